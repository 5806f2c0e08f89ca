"""Multi-GPU path: reads shard by contiguous chunks of whole pairs, no collective;
the only cross-rank step is a host-side sum of kept counts.  Covered here on CPU with
world_size-2 gloo: each rank filters its shard with the oracle standing in for its GPU, and the
concatenation of shard results equals the single-device result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_whole_pairs():
    from mitoflex_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1000, 16_666_667):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from mitoflex_amd.sharding import reduce_counts, shard_range
    from oracle import oracle_lib as ol
    from tests.util_data import make_bait, make_reads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bait = make_bait()
    n_pairs = 501
    m1, m2 = make_reads(bait, n_pairs, seed=21, uniform=True), make_reads(bait, n_pairs, seed=22, uniform=True)
    t = ol.OracleTable(bait, 31)
    lo, hi = shard_range(n_pairs, rank, world)
    b1, _ = ol.filter_reads(t, ol.OracleReads.from_seqs(m1[lo:hi]), 1)
    b2, _ = ol.filter_reads(t, ol.OracleReads.from_seqs(m2[lo:hi]), 1)
    keep = np.unpackbits((b1 | b2).view(np.uint8), bitorder="little")[:hi - lo]
    kept, total = reduce_counts(int(keep.sum()), hi - lo, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, keep.tolist())
    if rank == 0:
        q.put((kept, total, [x for part in gathered for x in part]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single():
    import torch.multiprocessing as mp
    from oracle import oracle_lib as ol
    from tests.util_data import make_bait, make_reads
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    kept, total, keep = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    bait = make_bait()
    m1, m2 = make_reads(bait, 501, seed=21, uniform=True), make_reads(bait, 501, seed=22, uniform=True)
    t = ol.OracleTable(bait, 31)
    b1, _ = ol.filter_reads(t, ol.OracleReads.from_seqs(m1), 1)
    b2, _ = ol.filter_reads(t, ol.OracleReads.from_seqs(m2), 1)
    single = np.unpackbits((b1 | b2).view(np.uint8), bitorder="little")[:501].tolist()
    assert keep == single and total == 501 and kept == sum(single)
