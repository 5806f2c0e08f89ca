# How the pass time moves with the fraction of bait-derived reads (the benchmark plants 0.5 %).
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd import mitofilter as mf
from mitoflex_amd.utility.synth_bait import make_bait
bait = make_bait()
ks = mf.KmerSet.from_text(bait, 31)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 33_333_334
ppms = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 500, 5000, 20000, 100000)
for ppm in ppms:
    reads = mf.Reads.synth(n, 150, 1, bait, mito_ppm=ppm, n_read_ppm=0)
    for _ in range(3):                       # (the pass kind follows what the previous calls of this read set saw)
        mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 3)
    st = mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 10)
    print(f"bait reads {ppm / 1e4:6.2f} %: candidates {st.n_candidates:8d} pass {st.n_pass:8d} | screen {st.ms_screen*1e3:6.1f} mark {st.ms_mark*1e3:6.1f} "
          f"exact {st.ms_exact*1e3:6.1f} pass {st.ms_total*1e3:6.1f} us | {n / st.ms_total / 1e6:6.1f} G reads/s")
    reads.close()
