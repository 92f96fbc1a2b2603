"""k_describe_od (round 6): the GaussianBlur computed per keypoint window on the matrix cores inside the description kernel — no blurred plane,
no blur kernel.  Every frame of full launch groups against the oracle (key points + descriptors byte-equal), on the geometries, families, blur
roundings and float modes the blur kernels are tested on, border-heavy small levels and pitched / unaligned inputs included; and the same
launch groups through the blur-kernel path must give the same bytes."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _run(ex, frames, cap, row_stride=None, frame_stride=None, offset=0):
    import torch
    F, h, w = frames.shape
    row_stride = row_stride or w
    frame_stride = frame_stride or row_stride * h
    buf = np.full(offset + F * frame_stride + 64, 0xA5, np.uint8)
    for f in range(F):
        for r in range(h):
            o = offset + f * frame_stride + r * row_stride
            buf[o:o + w] = frames[f, r]
    d = torch.from_numpy(buf).cuda()
    kps = torch.zeros((F, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(F, dtype=torch.int32, device="cuda")
    st = torch.zeros(F, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d.data_ptr() + offset, F, w, h, row_stride, frame_stride, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), cap, st.data_ptr())
    torch.cuda.synchronize()
    assert (st.cpu().numpy() == 0).all()
    return kps.cpu().numpy(), desc.cpu().numpy(), n.cpu().numpy()


CASES = [
    dict(w=640, h=480, nfeatures=1000),
    dict(w=640, h=480, nfeatures=1000, blur_rounding=capi.BLUR_HALF_UP),
    dict(w=640, h=480, nfeatures=1000, fp_contract=True),
    dict(w=640, h=480, nfeatures=2000),
    dict(w=640, h=480, nfeatures=1000, scoreType=capi.HARRIS_SCORE),
    dict(w=752, h=480, nfeatures=1000),                          # strides that are no multiple of 64
    dict(w=641, h=479, nfeatures=500, fastTh=12),                # odd sizes: every alignment phase of the windows, w & ~3 != w
    dict(w=1920, h=1080, nfeatures=2000),
    dict(w=1280, h=720, nfeatures=1000),
    dict(w=320, h=240, nfeatures=600, nlevels=5),                # small levels: a large share of border windows
    dict(w=160, h=120, nfeatures=200, nlevels=3),
    dict(w=200, h=600, nfeatures=400, nlevels=4),
    dict(w=640, h=480, nfeatures=1000, scaleFactor=1.5, nlevels=4),
]


@pytest.mark.parametrize("cfg", CASES, ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items()))
def test_on_demand_blur_equals_the_oracle_every_frame(gpu_extractor_factory, cfg):
    cfg = dict(cfg)
    w, h = cfg.pop("w"), cfg.pop("h")
    F = 33 if w >= 1280 else 40
    fams = [synth.BLOCKS, synth.NOISE, synth.LOWTEX, synth.MIDTEX, synth.BLOCKS]
    frames = np.stack([synth.frame(w, h, fams[i % 5], 300 + i) for i in range(F)])
    ex = gpu_extractor_factory(max_batch=F, **cfg)
    cap = ex.max_keypoints
    ex.set_blur_on_demand(1)
    k1, d1, n1 = _run(ex, frames, cap)
    ex.set_blur_on_demand(0)
    k0, d0, n0 = _run(ex, frames, cap)
    okw = dict(cfg)
    if "blur_rounding" in okw:
        okw["blur_mode"] = okw.pop("blur_rounding")
    o = orc.OracleExtractor(**okw)
    border = 0
    for f in range(F):
        ok, od = o(frames[f])
        assert n1[f] == len(ok) == n0[f], (f, n1[f], len(ok))
        got = k1[f, :n1[f]].reshape(-1).view(capi.KP_DTYPE)
        bad = np.flatnonzero((d1[f, :n1[f]] != od).any(axis=1))
        assert got.tobytes() == ok.tobytes(), f
        assert bad.size == 0, (f, bad[:10], ok[bad[:10]])
        assert k0[f, :n0[f]].tobytes() == k1[f, :n1[f]].tobytes() and np.array_equal(d0[f, :n0[f]], d1[f, :n1[f]])
    assert n1.sum() > 0


def test_on_demand_blur_pitched_and_unaligned_inputs(gpu_extractor_factory):
    """rows 704 bytes apart, the buffer starting at an odd address, the last frame ending exactly where the promise of include/orbx.h ends"""
    w, h, F = 640, 480, 36
    frames = np.stack([synth.frame(w, h, [synth.BLOCKS, synth.MIDTEX][i % 2], 500 + i) for i in range(F)])
    o = orc.OracleExtractor()
    want = [o(fr) for fr in frames]
    for row_stride, offset in ((704, 0), (704, 3), (641, 1), (644, 0), (640, 2)):
        ex = gpu_extractor_factory(max_batch=F)
        ex.set_blur_on_demand(1)
        k, d, n = _run(ex, frames, ex.max_keypoints, row_stride=row_stride, offset=offset)
        for f in range(F):
            ok, od = want[f]
            assert n[f] == len(ok)
            assert k[f, :n[f]].tobytes() == ok.tobytes() and np.array_equal(d[f, :n[f]], od), (row_stride, offset, f)


def test_on_demand_blur_on_adversarial_frames(gpu_extractor_factory):
    """saturated / binary / checkerboard / step frames (the rounding ties and the saturation of the filter), both rounding modes"""
    w, h = 640, 480
    yy, xx = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(9)
    fr = [(((yy // 5 + xx // 7) & 1) * 255).astype(np.uint8), (((yy // 9 + xx // 4) & 1) * 254 + 1).astype(np.uint8),
          np.where((xx // 16 + yy // 12) & 1, 255, 0).astype(np.uint8), (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8),
          np.where(rng.random((h, w)) < 0.02, 255, 0).astype(np.uint8), np.where(rng.random((h, w)) < 0.02, 0, 255).astype(np.uint8),
          (((yy * 3 + xx * 5) // 11) & 255).astype(np.uint8), np.full((h, w), 255, np.uint8)]
    fr += [synth.frame(w, h, synth.NOISE, 700 + i) for i in range(32 - len(fr))]
    frames = np.stack(fr)
    for mode in (capi.BLUR_X86_SSE2, capi.BLUR_HALF_UP):
        ex = gpu_extractor_factory(max_batch=32, blur_rounding=mode)
        ex.set_blur_on_demand(1)
        k, d, n = _run(ex, frames, ex.max_keypoints)
        o = orc.OracleExtractor(blur_mode=mode)
        for f in range(len(frames)):
            ok, od = o(frames[f])
            assert n[f] == len(ok), (mode, f)
            assert k[f, :n[f]].tobytes() == ok.tobytes() and np.array_equal(d[f, :n[f]], od), (mode, f)
    assert n[:7].sum() > 1000


@pytest.mark.parametrize("w,row_stride", [(640, 704), (644, 704), (640, 640), (641, 641)])
@pytest.mark.parametrize("mode", [1, 0], ids=["on_demand", "blur_kernels"])
def test_pitched_input_ending_at_the_documented_readable_limit(gpu_extractor_factory, w, row_stride, mode):
    """VERDICT r05 W7 / ADVICE r04: include/orbx.h promises the kernels only min(row_stride, w rounded up to 16) readable bytes per row, the LAST row of
    the LAST frame included.  The frames sit in a larger allocation so that they END exactly there; what follows is poison.  Nothing may be written
    behind the limit, and the outputs may not depend on what the poison is (a read past the limit that reached a result would show) — through the
    on-demand description kernel and through both blur kernels (k_blur_mfma: 16-byte pitches; k_blur: the others)."""
    import torch
    h, F = 480, 33
    frames = np.stack([synth.frame(w, h, [synth.BLOCKS, synth.MIDTEX, synth.NOISE][i % 3], 900 + i) for i in range(F)])
    readable = min(row_stride, (w + 15) & ~15)
    frame_stride = row_stride * h
    used = (F - 1) * frame_stride + (h - 1) * row_stride + readable
    tail = 8192
    o = orc.OracleExtractor()
    want = [o(fr) for fr in frames[[0, 1, F - 2, F - 1]]]
    outs = []
    for poison in (0x00, 0xFF, 0x5A):
        buf = np.full(used + tail, poison, np.uint8)
        for f in range(F):
            for r in range(h):
                o0 = f * frame_stride + r * row_stride
                buf[o0:o0 + w] = frames[f, r]
        ex = gpu_extractor_factory(max_batch=F)
        ex.set_blur_on_demand(mode)
        cap = ex.max_keypoints
        d = torch.from_numpy(buf).cuda()
        kps = torch.zeros((F, cap, 28), dtype=torch.uint8, device="cuda")
        desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros(F, dtype=torch.int32, device="cuda")
        ex.extract_batch_device(d.data_ptr(), F, w, h, row_stride, frame_stride, kps.data_ptr(), desc.data_ptr(), n.data_ptr(), cap)
        torch.cuda.synchronize()
        back = d.cpu().numpy()
        assert (back[used:] == poison).all() and np.array_equal(back[:used], buf[:used])           # the input is read-only, the tail untouched
        outs.append((kps.cpu().numpy(), desc.cpu().numpy(), n.cpu().numpy()))
    for k, dd, nn in outs[1:]:
        assert np.array_equal(nn, outs[0][2]) and np.array_equal(k, outs[0][0]) and np.array_equal(dd, outs[0][1])
    k, dd, nn = outs[0]
    for (ok, od), f in zip(want, (0, 1, F - 2, F - 1)):
        assert nn[f] == len(ok) and k[f, :nn[f]].tobytes() == ok.tobytes() and np.array_equal(dd[f, :nn[f]], od), f
