#!/usr/bin/env python3
"""Integer model of k_blur_mfma (orb_slam_amd/csrc/k_blur.hip): the per-lane data flow of the MFMA formulation of the 7x7
Gaussian blur — operand slots, the +128 constant slot, the 16-bit -> (hi, lo) byte split, the previous / current row-tile pair, the
column permutation that leaves every lane 12 contiguous output pixels — on numpy, checked against the oracle's gaussian_blur7
(both rounding modes).  v_mfma_i32_32x32x32_i8 is modelled by what the layout probe measured (profiles/r02_mfma_layout.txt):
D[m][n] = sum over the two lane groups g and the 16 byte slots s of A[lane (m, g)][s] * B[lane (n, g)][s]; lane (n, g) of D
holds rows m = 8 (r / 4) + 4 g + r % 4 in register r.  Development aid (CPU only); the kernel is tested on the GPU by tests/."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_lib as orc

G7 = [18, 34, 49, 55, 49, 34, 18]
LANES = np.arange(64)
LM, LG = LANES % 32, LANES // 32


def mfma(A, B, C):
    """A, B: [64][16] int8 slots; C: [64][16] int32 -> D in the measured layout"""
    A = A.astype(np.int64); B = B.astype(np.int64)
    M = np.zeros((32, 32), np.int64)                      # M[m][n]
    for g in (0, 1):
        M += A[32 * g:32 * g + 32] @ B[32 * g:32 * g + 32].T
    D = C.astype(np.int64).copy()
    for l in range(64):
        n, g = l % 32, l // 32
        for r in range(16):
            D[l, r] += M[8 * (r // 4) + 4 * g + r % 4, n]
    return D


def reflect101(p, n):
    if p < 0: p = -p
    if p >= n: p = 2 * n - 2 - p
    return min(max(p, 0), n - 1)


def pi_col(colidx):
    """D2 register (i, j) of a lane of group g holds colidx = 8 i + 4 g + j; it is made image column 12 g + 4 i + j of the tile"""
    i, g, j = colidx // 8, (colidx // 4) % 2, colidx % 4
    return 12 * g + 4 * i + j if i < 3 else -1


def blur_plane(img, blur_mode):
    h, w = img.shape
    wvec = (w & ~3) if blur_mode == 0 else 0
    w4 = (w + 3) & ~3
    stride = (w + 63) // 64 * 64
    src = np.zeros((h, stride), np.uint8); src[:, :w] = img; src[:, w:] = 0xAB       # padding bytes must not matter
    out = np.zeros((h, stride), np.uint8)
    ntc = (w4 + 23) // 24
    # column-pass B operands: lane (n = out row m', g), slot (v, b) <-> in-row rho = 8 v + 4 g + b
    Wcur = np.zeros((64, 16), np.int8); Wprev = np.zeros((64, 16), np.int8)
    for l in range(64):
        for s in range(16):
            rho = 8 * (s // 4) + 4 * LG[l] + s % 4
            t = rho - LM[l] + 6
            if 0 <= t <= 6: Wcur[l, s] = G7[t]
            t = rho - LM[l] - 26
            if 0 <= t <= 6: Wprev[l, s] = G7[t]
    for tc in range(ntc):
        last = tc == ntc - 1 and tc > 0
        xs0 = w4 - 24 if last else 24 * tc
        c0 = 0 if tc == 0 else (w4 - 32 if last else xs0 - 4)
        trick = tc != 0                                   # slot k = 0 is free: constant 64 there, weight 2 -> + 128
        Tr = np.zeros((64, 16), np.int64)
        for l in range(64):
            o = pi_col(LM[l])
            if o >= 0:
                for t in range(7):
                    k = reflect101(xs0 + o - 3 + t, w) - c0
                    assert 0 <= k < 32, (w, tc, o, t, k)
                    if k // 16 == LG[l]: Tr[l, k % 16] += G7[t]
            if trick and LG[l] == 0: Tr[l, 0] += 2
        assert Tr.max() <= 127
        Tr = Tr.astype(np.int8)
        # per-lane output columns and rounding constants
        xcol = xs0 + 12 * LG                              # first of the lane's 12 columns
        def load_tile(R):
            A = np.zeros((64, 16), np.int8)
            for l in range(64):
                row = reflect101(R + LM[l], h)
                px = src[row, c0 + 16 * LG[l]: c0 + 16 * LG[l] + 16].astype(np.int16)
                a = (px - 128).astype(np.int8)
                if trick and LG[l] == 0: a[0] = 64
                A[l] = a
            return A
        def row_pass(A):
            H = mfma(A, Tr, np.zeros((64, 16), np.int32))                 # lane (colidx, g): rows 8 i + 4 g + j in reg 4 i + j
            if not trick: H = H + 128
            assert H.min() >= -32768 and H.max() <= 32767
            Z = H & 0xFFFF
            lo = ((Z & 255) ^ 0x80).astype(np.uint8).view(np.int8)       # slot (v = i, b = j) <- reg 4 i + j
            hi = (Z >> 8).astype(np.uint8).view(np.int8)
            return hi, lo
        prev = row_pass(load_tile(3 - 32))
        for Y0 in range(0, h, 32):
            cur = row_pass(load_tile(Y0 + 3))
            HI = mfma(prev[0], Wprev, np.zeros((64, 16), np.int32)); HI = mfma(cur[0], Wcur, HI)     # A = H^T (lane = colidx), B = weights (lane = out row)
            K = 257 * 32896 + 0x7FFF
            T = HI * 256 + K
            T = mfma(prev[1], Wprev, T); T = mfma(cur[1], Wcur, T)
            for l in range(64):
                y = Y0 + LM[l]
                if y >= h: continue
                for i in range(3):
                    for j in range(4):
                        x = xcol[l] + 4 * i + j
                        tew = 1 if (x & ~3) < wvec else 0
                        t = int(T[l, 4 * i + j]) + (1 - tew)
                        q = t + (((t >> 16) & 1) if tew else 0)
                        if x < stride: out[y, x] = min(q >> 16, 255)
            prev = cur
    return out[:, :w]


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for (w, h) in ((97, 83), (640, 75), (41, 40), (533, 70), (214, 161)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img[: h // 3] = 255; img[h // 3: h // 2, : w // 2] = 0           # saturated regions: 255 * 257 * 257 >> 16 = 256 -> clamp
        for mode in (0, 1):
            got = blur_plane(img, mode)
            ref = orc.gaussian_blur7(img, mode)
            bad = np.argwhere(got != ref)
            print(w, h, "mode", mode, "mismatches", len(bad), bad[:3].tolist() if len(bad) else "")
