# ad-hoc: how does the exact kernel's time scale with the number of candidates?
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd import mitofilter as mf
from tests.util_data import make_bait
bait = make_bait()
ks = mf.KmerSet.from_text(bait, 31)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 33_333_334
for ppm in (0, 500, 5000, 20000, 100000):
    reads = mf.Reads.synth(n, 150, 1, bait, mito_ppm=ppm, n_read_ppm=0)
    mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 3)
    st = mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 10)
    print(f"mito_ppm {ppm:6d}: cand {st.n_candidates:8d} pass {st.n_pass:8d} screen {st.ms_screen*1e3:7.1f} us exact {st.ms_exact*1e3:7.1f} us total {st.ms_total*1e3:7.1f} us")
    reads.close()
