import os, sys, time
sys.path.insert(0, '/root/repo')
from mitoflex_amd import mitofilter as mf
from mitoflex_amd.utility.synth_bait import make_bait
bait = make_bait()
for k in (31, 41):
    ks = mf.KmerSet.from_text(bait, k)
    for sub in (10000, 0):
        reads = mf.Reads.synth(33_333_334, 150, 20261003, bait, sub_ppm=sub)
        os.environ["MF_EVENT_STRIDE"] = "1000000000"
        for _ in range(3): mf.filter_resident(ks, reads, 1, 0, 10)
        mf.device_synchronize(0); t0 = time.perf_counter(); st = mf.filter_resident(ks, reads, 1, 0, 30); mf.device_synchronize(0); dt = (time.perf_counter() - t0) / 30
        os.environ["MF_EVENT_STRIDE"] = "1"
        sp0 = mf.filter_resident(ks, reads, 1, 0, 10)
        os.environ["MF_TIME_PHASE1"] = "1"
        print(f"k={k} sub_ppm={sub}: {dt*1e3:.4f} ms/pass, screen {sp0.ms_screen*1e3:.1f} us, phase0 {sp0.ms_exact*1e3:.1f} us, items {st.n_candidates}, pass {st.n_pass}", flush=True)
        reads.close()
