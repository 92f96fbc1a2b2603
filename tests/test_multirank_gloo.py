"""N>1 path on CPU: two and eight ranks over gloo exercise the same sharding + reduction code bench.py runs over RCCL.
Each rank owns an independent image stream (no data-path collective); only timing / counters are reduced."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    from orb_slam_amd import dist_util, synth
    import oracle_lib as orc
    dist = dist_util.init("gloo", world, rank)
    ring = 4
    first = dist_util.stream_first_index(rank, ring)
    frames = synth.frames(160, 120, synth.BLOCKS, first, ring)          # this rank's stream
    k, d = orc.OracleExtractor(nfeatures=100, nlevels=3)(frames[0])    # stands in for the GPU step on CPU
    elapsed = 1.0 + rank                                                # rank 1 is the slow one
    tmax, total, rows = dist_util.reduce_run(dist, elapsed, [ring, len(k), int(frames[0].sum()) % 1000003], torch.device("cpu"))
    dist.barrier()
    q.put((rank, first, tmax, total, rows))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, t0, tot0, rows0), (r1, f1, t1, tot1, rows1) = res
    assert (f0, f1) == (0, 4)                              # disjoint streams
    assert t0 == t1 == 2.0                                 # MAX over ranks
    assert tot0 == tot1 and tot0[0] == 8.0                 # whole-job frame count
    assert rows0 == rows1 and rows0[0][2] != rows0[1][2]   # ranks really processed different frames
    assert tot0[1] == rows0[0][1] + rows0[1][1]


def test_eight_rank_sharding_and_reduction():
    """the world size the driver's scaling run ends at: eight disjoint streams, MAX of the elapsed times, whole-job sums"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [4 * r for r in range(world)]            # rank r owns frames [4 r, 4 r + 4)
    assert all(r[2] == float(world) for r in res)                          # MAX over ranks = the slowest rank's 1 + 7 s
    tot, rows = res[0][3], res[0][4]
    assert all(r[3] == tot and r[4] == rows for r in res)                  # every rank holds the same reduction
    assert tot[0] == 4.0 * world and tot[1] == sum(row[1] for row in rows)
    assert len({row[2] for row in rows}) == world                          # eight different streams


def test_single_process_path():
    from orb_slam_amd import dist_util
    t, tot, rows = dist_util.reduce_run(None, 0.5, [3, 4], torch.device("cpu"))
    assert t == 0.5 and tot == [3.0, 4.0] and rows == [[3.0, 4.0]]
