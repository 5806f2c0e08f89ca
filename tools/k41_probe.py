# configs[2], k = 41: what hides the two-word finish kernels under the next screen?  Pass time with the finish kernels of consecutive
# passes on one stream and on two (mf_set_option finish_streams), and with one / two screen streams, same resident reads, same box.
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd import mitofilter as mf
from mitoflex_amd.utility.synth_bait import make_bait
bait = make_bait()
n = 33_333_334
reads = mf.Reads.synth(n, 150, 20261003, bait, mito_ppm=5000, sub_ppm=10000, n_read_ppm=10000, n_base_ppm=1000)
os.environ["MF_EVENT_STRIDE"] = "1000000000"
for k in (41, 31, 63):
    ks = mf.KmerSet.from_text(bait, k)
    for fs in ("0", "1", "2"):
        mf.set_option("finish_streams", fs)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.1:
            mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 20)
        best = 1e9
        for _ in range(5):
            mf.device_synchronize(0)
            t0 = time.perf_counter()
            st = mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 20)
            mf.device_synchronize(0)
            best = min(best, (time.perf_counter() - t0) / 20)
        print(f"k {k} finish_streams {fs}: {best * 1e3:.4f} ms per step, {n / best / 1e9:.2f} G reads/s, whole pass {1254166692 / best / 8e12:.3f} of HBM peak, pass {st.n_pass}", flush=True)
    mf.set_option("finish_streams", "0")
    ks.close()
