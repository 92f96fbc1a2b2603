"""GPU parity of the Hamming top-2 matcher against the sequential-scan oracle (integer exact)."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

PATHS = {"mfma": 1, "mfma_fp4": 2, "popcount": 0}


@pytest.fixture(params=sorted(PATHS))
def match_path(request):
    """Every dense top-2 case runs through the three kernel families the library ships: the FP4 and the int8 MFMA forms and the
    xor + popcount form `north_star` describes (ORBX_MATCH_MFMA=4 / 8 / 0) — selected at run time through the C ABI's test hook."""
    capi.set_match_path(PATHS[request.param])
    yield request.param
    capi.set_match_path(-1)


def _check(Q, T):
    gi, gb, gs = capi.match_top2(Q, T)
    ri, rb, rs = orc.match_top2(Q, T)
    np.testing.assert_array_equal(gb, rb)
    np.testing.assert_array_equal(gs, rs)
    np.testing.assert_array_equal(gi, ri)


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (5, 3), (64, 64), (513, 1000), (1000, 1000), (2000, 2000), (300, 4097), (1025, 70001)])
def test_random(nq, nt, match_path):
    _check(synth.descriptors(nq, 10 + nq), synth.descriptors(nt, 20 + nt))


def test_empty_train_and_empty_query(match_path):
    gi, gb, gs = capi.match_top2(synth.descriptors(7, 1), np.zeros((0, 32), np.uint8))
    assert (gi == -1).all() and (gb == 2**31 - 1).all() and (gs == 2**31 - 1).all()
    gi, gb, gs = capi.match_top2(np.zeros((0, 32), np.uint8), synth.descriptors(7, 1))
    assert len(gi) == 0
    gi, gb, gs = capi.match_top2(synth.descriptors(3, 1), synth.descriptors(1, 2))   # one train: second = INT_MAX
    assert (gs == 2**31 - 1).all() and (gi == 0).all()


def test_ties_duplicates_and_extremes(match_path):
    rng = np.random.default_rng(7)
    T = synth.descriptors(3000, 5)
    T[100:2000:7] = T[50]                 # many exact duplicates -> best==second, first index must win
    T[1234] = 0
    T[2345] = 255
    Q = np.concatenate([T[50:51], T[:200], np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8), synth.descriptors(800, 9)])
    few = rng.integers(0, 4, size=(4000, 32)).astype(np.uint8)   # low-entropy descriptors: dense distance ties
    _check(Q, T)
    _check(few[:1500], few[1500:])


@pytest.mark.parametrize("nq,nt,keep", [(1, 1, 1.0), (5, 3, 0.5), (1000, 1000, 0.7), (300, 4097, 0.1), (513, 70001, 0.9), (64, 2000, 0.0), (2000, 100, 0.5), (40000, 1000, 0.5), (9000, 9000, 0.8)])
def test_train_mask(nq, nt, keep, match_path):
    """SURVEY 8b's optional t_valid mask (src/ORBmatcher.cc:205-206: already matched train features are skipped): the scan over the
    valid descriptors only = the oracle's sequential scan of the order-preserving compaction, indices mapped back to the train array"""
    rng = np.random.default_rng(nq * 7 + nt)
    Q, T = synth.descriptors(nq, 31 + nq), synth.descriptors(nt, 41 + nt)
    if nt > 50:
        T[10:nt:5] = T[3]                                   # duplicates: the first VALID index attaining the best distance must win
    valid = (rng.random(nt) < keep).astype(np.uint8)
    gi, gb, gs = capi.match_top2_masked(Q, T, valid)
    kept = np.flatnonzero(valid)
    ri, rb, rs = orc.match_top2(Q, T[kept])
    ri = np.where(ri >= 0, kept[np.maximum(ri, 0)] if len(kept) else -1, -1).astype(np.int32)
    np.testing.assert_array_equal(gb, rb)
    np.testing.assert_array_equal(gs, rs)
    np.testing.assert_array_equal(gi, ri)
    if keep == 1.0:
        for a, b in zip((gi, gb, gs), capi.match_top2(Q, T)):
            np.testing.assert_array_equal(a, b)


def test_batch_device(match_path):
    torch = pytest.importorskip("torch")
    B, cap = 9, 1000
    nq = np.array([1000, 999, 0, 1, 513, 1000, 37, 512, 1000], np.int32)
    nt = np.array([1000, 5, 1000, 0, 1000, 1, 999, 1000, 1000], np.int32)
    Q = np.stack([synth.descriptors(cap, 100 + i) for i in range(B)])
    T = np.stack([synth.descriptors(cap, 200 + i) for i in range(B)])
    T[4, 10:900:3] = T[4, 7]
    dQ, dT = torch.from_numpy(Q).cuda(), torch.from_numpy(T).cuda()
    dnq, dnt = torch.from_numpy(nq).cuda(), torch.from_numpy(nt).cuda()
    out = torch.full((3, B, cap), -7, dtype=torch.int32, device="cuda")
    capi.match_top2_batch_device(dQ.data_ptr(), dnq.data_ptr(), dT.data_ptr(), dnt.data_ptr(), B, cap,
                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for i in range(B):
        ri, rb, rs = orc.match_top2(Q[i, :nq[i]], T[i, :nt[i]])
        np.testing.assert_array_equal(o[0, i, :nq[i]], ri)
        np.testing.assert_array_equal(o[1, i, :nq[i]], rb)
        np.testing.assert_array_equal(o[2, i, :nq[i]], rs)
        assert (o[:, i, nq[i]:] == -7).all()          # nothing written past nq


@pytest.mark.parametrize("cap,nq,nt", [(20000, 300, 19997), (32768, 40, 32768), (40000, 33, 39999)])
def test_batch_long_train_lists(cap, nq, nt, match_path):
    """One problem with a long train list.  The FP4 form's keys are hamming * 2^S + index in an f32 accumulator: S = 15 (the largest:
    sums up to 2^23 + 2^15) at 20000 and 32768 entries, and past 2^15 entries the call takes the int8 kernels.  Extremes planted: exact
    matches (hamming 0), complements (hamming 256), repeated rows at the far end of the list (first index wins)."""
    torch = pytest.importorskip("torch")
    Q, T = synth.descriptors(cap, 31 + cap), synth.descriptors(cap, 32 + cap)
    Q[0] = T[nt - 1]                                     # distance 0 at the last index
    Q[1] = ~T[nt - 2]                                    # distance 256 to one entry
    T[nt - 40:nt:3] = T[5]                               # repeated rows: the first index attains the best distance
    Q[2] = T[5]
    Q[3] = 0; Q[4] = 255
    dQ, dT = torch.from_numpy(Q).cuda(), torch.from_numpy(T).cuda()
    dnq, dnt = torch.tensor([nq], dtype=torch.int32, device="cuda"), torch.tensor([nt], dtype=torch.int32, device="cuda")
    out = torch.full((3, cap), -7, dtype=torch.int32, device="cuda")
    capi.match_top2_batch_device(dQ.data_ptr(), dnq.data_ptr(), dT.data_ptr(), dnt.data_ptr(), 1, cap,
                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    ri, rb, rs = orc.match_top2(Q[:nq], T[:nt])
    np.testing.assert_array_equal(o[0, :nq], ri)
    np.testing.assert_array_equal(o[1, :nq], rb)
    np.testing.assert_array_equal(o[2, :nq], rs)
    assert (o[:, nq:] == -7).all()


def test_many_queries_long_train_list(match_path):
    """So many query blocks (> 2048) that the split heuristic would take the whole train list as ONE chunk — longer than the 2^15 entries the FP4
    form's keys can index: the host splits it (`floor_ns`), S = 15.  Sampled rows against the oracle + the self-match property on planted rows."""
    nq, nt = 1_060_000, 40_000
    Q, T = synth.descriptors(nq, 91), synth.descriptors(nt, 92)
    plant = np.arange(0, nq, 9973)
    Q[plant] = T[(plant * 7) % nt]                       # exact matches, some of them in the second chunk
    gi, gb, gs = capi.match_top2(Q, T)
    assert (gb[plant] == 0).all()
    sample = np.unique(np.concatenate([plant[:40], np.arange(0, nq, 26501), [nq - 1]]))
    ri, rb, rs = orc.match_top2(Q[sample], T)
    np.testing.assert_array_equal(gi[sample], ri)
    np.testing.assert_array_equal(gb[sample], rb)
    np.testing.assert_array_equal(gs[sample], rs)


def test_full_size_properties(match_path):
    """100k x 100k (BASELINE config 5) is too slow for the scalar oracle; check size-independent properties:
    self-match (best=0 at own index when descriptors are unique) and agreement with the oracle on a query sample."""
    n = 100_000
    D = synth.descriptors(n, 77)
    gi, gb, gs = capi.match_top2(D, D)
    assert (gb == 0).all()
    assert (gi == np.arange(n)).all()                 # PRNG descriptors are unique
    assert (gs > 0).all() and (gs <= 256).all()
    sample = np.arange(0, n, 997)
    ri, rb, rs = orc.match_top2(D[sample], D)
    np.testing.assert_array_equal(gi[sample], ri)
    np.testing.assert_array_equal(gs[sample], rs)
    assert capi.count_accepted(gb, gs, 50, 0.6) == orc.lib().orc_count_accepted(gb.ctypes.data, gs.ctypes.data, n, 50, 0.6)


def test_full_size_every_row_against_the_oracle():
    """100k x 100k (BASELINE config 5), EVERY query row: the scalar oracle scan runs on all host cores (ctypes releases the
    GIL; 1e10 pairs at ~1.6e8 pairs/s per core), with planted duplicates so that first-index ties and the multiplicity of the
    second-best distance occur at this size too."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = 100_000
    Q, T = synth.descriptors(n, 78), synth.descriptors(n, 79)
    T[5000:90000:1777] = T[123]                        # repeated train descriptors (ties on the best distance)
    Q[::4999] = T[40000:40000 + len(Q[::4999])]        # exact matches (distance 0)
    Q[7::9973] = T[123]                                 # best = 0 with multiplicity: second = 0
    gi, gb, gs = capi.match_top2(Q, T)
    for other in (0, 1, 2):                             # the xor + popcount, int8 MFMA and FP4 MFMA kernels on the same problem (the oracle scan runs once)
        capi.set_match_path(other)
        try:
            pi, pb, ps = capi.match_top2(Q, T)
        finally:
            capi.set_match_path(-1)
        np.testing.assert_array_equal(pi, gi)
        np.testing.assert_array_equal(pb, gb)
        np.testing.assert_array_equal(ps, gs)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        cores = int(float(q) / float(p)) if q != "max" else len(os.sched_getaffinity(0))
    except Exception:
        cores = len(os.sched_getaffinity(0))
    cores = max(1, min(cores, 64))
    blocks = [(a, min(a + 500, n)) for a in range(0, n, 500)]
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(lambda ab: orc.match_top2(Q[ab[0]:ab[1]], T), blocks))
    ri = np.concatenate([r[0] for r in res]); rb = np.concatenate([r[1] for r in res]); rs = np.concatenate([r[2] for r in res])
    np.testing.assert_array_equal(gi, ri)
    np.testing.assert_array_equal(gb, rb)
    np.testing.assert_array_equal(gs, rs)
    assert (gs[7::9973] == 0).all() and (gb[::4999] == 0).all()


def _random_segments(rng, nq, nt, max_len):
    lens = rng.integers(0, max_len + 1, nq)
    lens[rng.integers(0, nq, max(nq // 10, 1))] = 0                       # empty candidate sets
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = rng.integers(0, nt, int(seg[-1])).astype(np.int32)            # unsorted, with repeats
    return seg, cand


@pytest.mark.parametrize("nq,nt,max_len", [(1, 10, 5), (100, 1000, 40), (1000, 1000, 130), (257, 5000, 300), (2000, 2000, 64)])
def test_candidate_lists(nq, nt, max_len):
    """window / vocabulary-node candidate sets: scan in LIST order, first listed candidate wins ties"""
    rng = np.random.default_rng(nq + nt)
    Q, T = synth.descriptors(nq, 3), synth.descriptors(nt, 4)
    T[: nt // 3] = T[0]                                                   # heavy duplication -> ties decided by list order
    seg, cand = _random_segments(rng, nq, nt, max_len)
    gi, gb, gs = capi.match_top2_segments(Q, T, seg, cand)
    ri, rb, rs = orc.match_top2_segments(Q, T, seg, cand)
    np.testing.assert_array_equal(gb, rb)
    np.testing.assert_array_equal(gs, rs)
    np.testing.assert_array_equal(gi, ri)


def test_candidate_lists_equal_dense_when_lists_are_full():
    nq, nt = 300, 500
    Q, T = synth.descriptors(nq, 8), synth.descriptors(nt, 9)
    seg = (np.arange(nq + 1) * nt).astype(np.int32)
    cand = np.tile(np.arange(nt, dtype=np.int32), nq)
    a = capi.match_top2_segments(Q, T, seg, cand)
    b = capi.match_top2(Q, T)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_distinctive_descriptors():
    """MapPoint::ComputeDistinctiveDescriptors batched over map points: least-median row, first on ties"""
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 4, 5, 7, 8, 16, 33, 63, 64, 65, 100, 130, 0, 257, 2, 2, 6] + list(rng.integers(1, 40, 300))
    segs, descs = [0], []
    for i, n in enumerate(sizes):
        d = synth.descriptors(n, 500 + i) if n else np.zeros((0, 32), np.uint8)
        if n >= 4 and i % 3 == 0:
            base = d[0].copy()                       # a tight cluster (observations of one point): few bits apart, many ties
            for r in range(n):
                d[r] = base
                for bit in rng.integers(0, 256, int(rng.integers(0, 12))):
                    d[r, bit // 8] ^= np.uint8(1 << (bit % 8))
        if n >= 2 and i % 5 == 1:
            d[1:] = d[0]                             # all identical: every median 0 -> index 0
        descs.append(d)
        segs.append(segs[-1] + n)
    desc = np.concatenate(descs)
    gi, gm = capi.distinctive(desc, np.array(segs, np.int32))
    for p, n in enumerate(sizes):
        wi, wm = orc.distinctive(desc[segs[p]:segs[p + 1]])
        assert (gi[p], gm[p]) == (wi, wm), (p, n, gi[p], gm[p], wi, wm)
    gi, gm = capi.distinctive(np.zeros((0, 32), np.uint8), np.array([0], np.int32))
    assert len(gi) == 0


def test_top2_property_hypothesis(match_path):
    """for ANY query / train multiset (hypothesis explores tiny alphabets, duplicates, empty sets) the GPU top-2 equals
    the sequential-scan semantics: two smallest distances with multiplicity, first index on ties"""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    palette = np.concatenate([np.zeros((1, 32), np.uint8), np.full((1, 32), 255, np.uint8), synth.descriptors(6, 3)])
    palette[3] = palette[2]; palette[3, 0] ^= 1          # distance-1 neighbours
    palette[5] = palette[4]; palette[5, 31] ^= 0x81

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.integers(0, 7), min_size=0, max_size=40), st.lists(st.integers(0, 7), min_size=0, max_size=70))
    def prop(qi, ti):
        Q = palette[np.array(qi, dtype=np.int64)] if qi else np.zeros((0, 32), np.uint8)
        T = palette[np.array(ti, dtype=np.int64)] if ti else np.zeros((0, 32), np.uint8)
        gi, gb, gs = capi.match_top2(Q, T)
        ri, rb, rs = orc.match_top2(Q, T)
        assert np.array_equal(gi, ri) and np.array_equal(gb, rb) and np.array_equal(gs, rs)

    prop()


def test_descriptor_arrays_need_only_dword_alignment(match_path):
    """device arrays that start 4 bytes into an allocation — all the alignment the reference's DescriptorDistance needs
    (8 x int32 reads, src/ORBmatcher.cc:1796-1801) — through the dense, the batched and the candidate-list kernels"""
    torch = pytest.importorskip("torch")
    nq, nt = 700, 1300
    Q, T = synth.descriptors(nq, 91), synth.descriptors(nt, 92)
    T[100:900:5] = T[7]
    bq = torch.zeros(nq * 32 + 4, dtype=torch.uint8, device="cuda"); bq[4:] = torch.from_numpy(Q.reshape(-1)).cuda()
    bt = torch.zeros(nt * 32 + 4, dtype=torch.uint8, device="cuda"); bt[4:] = torch.from_numpy(T.reshape(-1)).cuda()
    out = torch.zeros((3, nq), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert (bq.data_ptr() + 4) % 16 == 4
    ri, rb, rs = orc.match_top2(Q, T)
    capi.match_top2_device(bq.data_ptr() + 4, nq, bt.data_ptr() + 4, nt, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    np.testing.assert_array_equal(o[0], ri); np.testing.assert_array_equal(o[1], rb); np.testing.assert_array_equal(o[2], rs)
    dn = torch.tensor([nq, nt], dtype=torch.int32, device="cuda")
    out.zero_()
    capi.match_top2_batch_device(bq.data_ptr() + 4, dn[0:1].data_ptr(), bt.data_ptr() + 4, dn[1:2].data_ptr(), 1, nt, out[0].data_ptr(), out[1].data_ptr(),
                                 out[2].data_ptr(), st)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    np.testing.assert_array_equal(o[0], ri); np.testing.assert_array_equal(o[1], rb); np.testing.assert_array_equal(o[2], rs)
    rng = np.random.default_rng(5)
    seg, cand = _random_segments(rng, nq, nt, 50)
    dseg, dcand = torch.from_numpy(seg).cuda(), torch.from_numpy(cand).cuda()
    out.zero_()
    rc = capi.lib().orbm_match_top2_segments_device(bq.data_ptr() + 4, nq, bt.data_ptr() + 4, nt, dseg.data_ptr(), dcand.data_ptr(), out[0].data_ptr(),
                                                    out[1].data_ptr(), out[2].data_ptr(), st)
    assert rc == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    si, sb, ss = orc.match_top2_segments(Q, T, seg, cand)
    np.testing.assert_array_equal(o[0], si); np.testing.assert_array_equal(o[1], sb); np.testing.assert_array_equal(o[2], ss)
    # 2-byte alignment is refused
    assert capi.lib().orbm_match_top2_device(bq.data_ptr() + 2, nq, bt.data_ptr() + 4, nt, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st) == capi.ORBX_ERR_ARG


def test_randomised_sizes_and_entry_points():
    """a slice of tools/fuzz_match.py: the dense, per-frame batch and candidate-list entry points on both kernel families, sizes around the
    kernels' tile / block / split boundaries, duplicates, low-entropy descriptors, 4-byte-aligned arrays — integer exact vs the oracle"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_match.py"), "800", "11"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["mismatches"] == 0 and out["ok"] == 800 and min(out["by_entry_point"].values()) >= 100
