"""GPU parity of the throughput configuration (orb_slam_amd/pipeline.py): a step cut into concurrent lanes with the border
frame's descriptors handed over between streams gives, for EVERY frame of consecutive steps, exactly the keypoints,
descriptors and top-2 matches against the previous frame that one handle on one stream — and the oracle — give."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _run(frames, B, lanes, nf, steps):
    torch = pytest.importorskip("torch")
    from orb_slam_amd.pipeline import LanePipeline
    _, h, w = frames.shape
    d_img = torch.from_numpy(frames).cuda()
    pipe = LanePipeline(w, h, B, lanes=lanes, nfeatures=nf)
    assert pipe.G == lanes and pipe.b * lanes == B
    out = []
    for i in range(steps):
        pipe.step(d_img.data_ptr() + i * B * w * h)
        if i % 2 == 1 or i == steps - 1:
            torch.cuda.synchronize()           # (results are only read after a synchronise; steps 0->1 run back to back)
        if i % 2 == 0 and i != steps - 1:
            continue
        out.append((i, pipe.counts().cpu().numpy(), pipe.keypoints().cpu().numpy().view(np.uint8).reshape(B, pipe.cap, 28),
                    pipe.descriptors().cpu().numpy(), pipe.matches().cpu().numpy()))
    assert (pipe.status().cpu().numpy() == 0).all()
    pipe.close()
    return out


@pytest.mark.parametrize("lanes", [4, 3, 8])
def test_lanes_pipeline_equals_one_stream_and_oracle(lanes):
    B, w, h, nf, steps = 24, 320, 240, 300, 4
    frames = np.concatenate([synth.frames(w, h, synth.BLOCKS, 40, 60), synth.frames(w, h, synth.NOISE, 90, 20), synth.frames(w, h, synth.LOWTEX, 7, 16)])
    assert len(frames) == B * steps
    one = _run(frames, B, 1, nf, steps)
    many = _run(frames, B, lanes, nf, steps)
    assert [o[0] for o in one] == [o[0] for o in many] and len(one) >= 2
    o = ol.OracleExtractor(nfeatures=nf)
    checked = 0
    for (i, n1, k1, d1, m1), (_, n2, k2, d2, m2) in zip(one, many):
        np.testing.assert_array_equal(n1, n2)
        for f in range(B):
            n = n1[f]
            assert k1[f, :n].tobytes() == k2[f, :n].tobytes(), (i, f)
            assert d1[f, :n].tobytes() == d2[f, :n].tobytes(), (i, f)
            np.testing.assert_array_equal(m1[:, f, :n], m2[:, f, :n], err_msg="step %d frame %d" % (i, f))
        # the frames at the lane borders (and the step border) against the oracle: extraction and match vs the previous frame
        b = B // lanes
        for f in sorted({0, b - 1, b, 2 * b, B - 1}):
            g = i * B + f
            ok, od = o(frames[g])
            assert n2[f] == len(ok)
            np.testing.assert_array_equal(d2[f, :len(ok)], od)
            _, pd = o(frames[g - 1])
            if len(od) and len(pd):
                idx, best, sec = ol.match_top2(od, pd)
                np.testing.assert_array_equal(m2[0, f, :len(od)], idx)
                np.testing.assert_array_equal(m2[1, f, :len(od)], best)
                np.testing.assert_array_equal(m2[2, f, :len(od)], sec)
                checked += 1
    assert checked >= 8


def test_cpp_lane_pipeline(tmp_path):
    """orb_slam_amd/cpp/LanePipeline.h (plain C++ over the C ABI, no HIP headers): the example runs the same frames through 4 lanes
    and through one, checks them byte-identical itself, and its dump of the last step equals the Python pipeline's"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "orb_slam_amd", "cpp", "example_lanes")
    assert os.path.exists(exe), "run make"
    B, w, h, nf, steps = 24, 320, 240, 300, 4
    frames = np.concatenate([synth.frames(w, h, synth.BLOCKS, 40, 60), synth.frames(w, h, synth.NOISE, 90, 20), synth.frames(w, h, synth.LOWTEX, 7, 16)])
    raw, out = tmp_path / "frames.raw", tmp_path / "out.bin"
    raw.write_bytes(frames.tobytes())
    res = subprocess.run([exe, str(w), str(h), str(B), str(steps), "4", str(raw), str(out)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, EXAMPLE_NFEATURES=str(nf)))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "IDENTICAL" in res.stdout and "lanes 4 x 6 frames" in res.stdout
    blob = out.read_bytes()
    hdr = np.frombuffer(blob, np.int32, 4)
    assert hdr[0] == B and hdr[2] == 28
    cap = int(hdr[1])
    o = 16
    n = np.frombuffer(blob, np.int32, B, o); o += 4 * B
    kps = np.frombuffer(blob, np.uint8, B * cap * 28, o).reshape(B, cap, 28); o += B * cap * 28
    desc = np.frombuffer(blob, np.uint8, B * cap * 32, o).reshape(B, cap, 32); o += B * cap * 32
    match = np.frombuffer(blob, np.int32, 3 * B * cap, o).reshape(3, B, cap)
    i, n2, k2, d2, m2 = _run(frames, B, 4, nf, steps)[-1]
    assert i == steps - 1 and cap == k2.shape[1]
    np.testing.assert_array_equal(n, n2)
    for f in range(B):
        assert kps[f, :n[f]].tobytes() == k2[f, :n[f]].tobytes(), f
        assert desc[f, :n[f]].tobytes() == d2[f, :n[f]].tobytes(), f
        np.testing.assert_array_equal(match[:, f, :n[f]], m2[:, f, :n[f]], err_msg="frame %d" % f)


def test_lane_placement_is_probed_and_survives_foreign_streams():
    """Streams created by other libraries before the pipeline shift the runtime's stream -> hardware-queue placement; the pipeline
    times its candidate stream sets in an explicit tune() call (step() itself never probes or blocks), reports the choice, and the
    probes leave no trace in the results; a fixed placement (argument or ORBX_LANE_PLACEMENT) skips the probe."""
    torch = pytest.importorskip("torch")
    foreign = [capi.stream_create(0) for _ in range(6)]
    try:
        B, w, h, nf, steps = 24, 320, 240, 300, 2
        frames = synth.frames(w, h, synth.BLOCKS, 40, B * steps)
        one = _run(frames, B, 1, nf, steps)
        many = _run(frames, B, 4, nf, steps)
        for (i, n1, k1, d1, m1), (_, n2, k2, d2, m2) in zip(one, many):
            np.testing.assert_array_equal(n1, n2)
            for f in range(B):
                n = n1[f]
                assert k1[f, :n].tobytes() == k2[f, :n].tobytes() and d1[f, :n].tobytes() == d2[f, :n].tobytes()
                np.testing.assert_array_equal(m1[:, f, :n], m2[:, f, :n])
        from orb_slam_amd.pipeline import LanePipeline
        d_img = torch.from_numpy(frames).cuda()
        pipe = LanePipeline(w, h, B, lanes=4, nfeatures=nf)
        pipe.step(d_img.data_ptr())
        torch.cuda.synchronize()
        assert pipe.placement["probe_ms_per_step"] is None          # step() alone never probes
        pipe.tune(d_img.data_ptr())
        p = pipe.placement
        assert p["candidates"] == 3 and len(p["probe_ms_per_step"]) == 3 and 0 <= p["chosen"] < 3
        assert p["probe_ms_per_step"][p["chosen"]] == min(p["probe_ms_per_step"])
        assert pipe.steps_done == 0 and int(pipe.counts().abs().sum().item()) == 0      # the probes left no state behind
        pipe.close()
        fixed = LanePipeline(w, h, B, lanes=4, nfeatures=nf, placement=2)
        assert fixed.tune(d_img.data_ptr())["probe_ms_per_step"] is None and fixed.placement["chosen"] == 2
        fixed.close()
    finally:
        for s in foreign:
            capi.stream_destroy(0, s)
