"""The ORCHESTRATION of the device ingest path (mitoflex_amd/csrc/mf_devingest.cpp: producer, uploader, consumers, writers, ring,
text-buffer pool, carry hand-off; the quality filter's decide / gather turns) on the CPU: tests/native/ingest_check.cpp runs the real
file against a stand-in HIP runtime whose streams are in-order queues on threads of their own (tests/native/hipstub) and stand-in
kernels that are obvious loops (tests/native/ingest_stub.cpp), with knobs so small that a seam falls inside everything -- plain and
under ThreadSanitizer.  The HIP kernels themselves are not what is tested here (tests/test_gpu_devingest.py does that on the GPU).

Round 5 found three defects of the product with it on its first runs: paired files of unequal length could starve the longer mate's
producer of text buffers (a hang), a failed run destroyed a condition variable its producers still used, and a named pipe as the
quality filter's output was opened twice (its reader saw an end of file in between)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mitoflex_amd", "csrc")
NATIVE = os.path.join(ROOT, "tests", "native")
SOURCES = [os.path.join(NATIVE, "ingest_check.cpp"), os.path.join(NATIVE, "ingest_stub.cpp"), os.path.join(NATIVE, "hipstub", "hipstub.cpp"),
           os.path.join(CSRC, "mf_devingest.cpp"), os.path.join(CSRC, "mf_pinflate.cpp"), os.path.join(CSRC, "mf_inflate.cpp"),
           os.path.join(CSRC, "mf_pipeline.cpp"), os.path.join(CSRC, "mf_host.cpp")]
VARIANTS = {"plain": [], "tsan": ["-fsanitize=thread"], "without_partial_decisions": ["-DMF_TEST_WITHOUT_PARTIAL_DECISIONS"]}
CASES = ["bait_se_gz", "bait_pe_gz", "bait_pe_gz_nocarry", "bait_plain", "bait_two_devices", "bait_members_flush", "bait_margin",
         "bait_default_knobs", "bait_long_records", "bait_damaged", "qual_pe", "qual_se", "qual_pe_budget", "qual_pipes"]


def _build(tmp, name, flags):
    """objects in parallel (mf_devingest.cpp alone takes 15 s), one link"""
    objs = []

    def cc(src):
        obj = os.path.join(tmp, name + "_" + os.path.basename(src) + ".o")
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", *flags, "-I", os.path.join(NATIVE, "hipstub"), "-I", CSRC, "-c", src, "-o", obj])
        return obj
    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, SOURCES))
    exe = os.path.join(tmp, "ingest_check_" + name)
    subprocess.check_call(["g++", *flags, *objs, "-lz", "-lpthread", "-o", exe])
    return exe


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("ingest_check"))
    return {name: _build(tmp, name, flags) for name, flags in VARIANTS.items()}


def _run(exe, case, tmp_path, records, timeout, **extra):
    env = dict(os.environ, INGEST_CHECK_RECORDS=str(records), INGEST_CHECK_TIMEOUT=str(timeout), TSAN_OPTIONS="report_thread_leaks=0 halt_on_error=0", **extra)
    for k in list(env):
        if k.startswith("MF_") and k not in extra:
            del env[k]
    return subprocess.run([exe, case, str(tmp_path)], env=env, capture_output=True, text=True, timeout=timeout + 60)


@pytest.mark.parametrize("case", CASES)
def test_ingest_orchestration(exes, tmp_path, case):
    r = _run(exes["plain"], case, tmp_path, 3000, 120)
    assert r.returncode == 0 and "all checks of" in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("case", CASES)
def test_ingest_orchestration_under_tsan(exes, tmp_path, case):
    r = _run(exes["tsan"], case, tmp_path, 1500, 240)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "all checks of" in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("case", ["bait_se_gz", "bait_plain", "bait_two_devices", "qual_pe"])
def test_uploads_through_staging_buffers(exes, tmp_path, case):
    """The uploads read the file's pages where the page cache holds them (the mapping registered with the runtime); where that cannot be done
    (the stand-in runtime refuses under STUB_NO_REGISTER) they go through pinned staging buffers as before -- same bytes, under TSan."""
    r = _run(exes["tsan"], case, tmp_path, 1500, 240, STUB_NO_REGISTER="1")
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "all checks of" in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("case", ["bait_se_gz", "bait_pe_gz", "bait_two_devices", "qual_pe"])
def test_bodies_and_crc_on_a_stream_of_their_own(exes, tmp_path, case):
    """On an input of many slabs the chunks' bodies and the CRC run on a second stream behind the link step they belong to (GzStream::lane_post,
    b_behind_a); MF_GZDEV_RESOLVE_STREAM=1 asks for that on the small files of this check -- same bytes, same CRC verdicts, nothing for TSan."""
    r = _run(exes["tsan"], case, tmp_path, 1500, 240, MF_GZDEV_RESOLVE_STREAM="1")
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "all checks of" in r.stderr, r.stderr[-3000:]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("case", ["bait_pe_gz", "bait_two_devices", "qual_pe"])
def test_other_sizes_of_everything_under_tsan(exes, tmp_path, case, seed):
    """The cases run with one set of tiny sizes; INGEST_CHECK_KNOB_SEED picks others (chunk, slab, ring, margin, text piece and buffers, consumers,
    upload buffers, one or two post streams, decode streams, slabs in flight) from a small generator: same bytes, nothing for TSan."""
    r = _run(exes["tsan"], case, tmp_path, 1200, 240, INGEST_CHECK_KNOB_SEED=str(seed))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0 and "all checks of" in r.stderr, r.stderr[-3000:]


def test_the_check_hangs_without_partial_decisions(exes, tmp_path):
    """Round 4 fixed a hang of the paired quality filter: a piece of mate 1 waited for decisions that waited for text buffers the other mate
    held (mf_devingest.cpp, q_progress: `ready`).  With that rule compiled out the check must hang (its watchdog exits with 3) -- i.e. this
    suite would have found it."""
    r = _run(exes["without_partial_decisions"], "qual_pe", tmp_path, 3000, 12)
    assert r.returncode == 3 and "HANG" in r.stderr, r.stderr[-2000:]
