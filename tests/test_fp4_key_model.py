"""CPU model of the FP4 matcher's key arithmetic (orb_slam_amd/csrc/orbm_match.hip, mfma4_scan; NOTES.md 10.8): the f32 accumulator that starts
at 2^(S+7) + i and takes four K = 64 partial sums of +-2^(S-1) ends as hamming * 2^S + i EXACTLY, float bit patterns order like (hamming, index),
rows past the chunk's end decode to "none".  No GPU: numpy float32 does what the matrix instruction's f32 accumulate does for exact operands."""
import numpy as np
import pytest

KEY_SHIFT, KEY_NONE = 22, 0xFFFFFFFF


def keys_f32(q_bits, t_bits, S, idx, penalised=False):
    """q_bits, t_bits: (256,) arrays of 0/1.  The kernel's arithmetic: +-1.0 nibbles, trains negated and scaled by 2^(S-1), accumulator start
    2^(S+7) + idx (+ 2^(S+9) for a row past the chunk's end), four chunks of K = 64 accumulated one after the other in float32."""
    q = np.where(q_bits, -1.0, 1.0).astype(np.float32)
    t = (-np.where(t_bits, -1.0, 1.0) * np.float32(2.0 ** (S - 1))).astype(np.float32)
    acc = np.float32(2.0 ** (S + 7) + idx) + (np.float32(2.0 ** (S + 9)) if penalised else np.float32(0))
    for c in range(4):
        part = np.float32(0)
        for k in range(64 * c, 64 * c + 64):          # any summation order inside the instruction gives the same: every term is +-2^(S-1)
            part = np.float32(part + q[k] * t[k])
        acc = np.float32(acc + part)
    return acc


def widen(bits, S, t0=0):
    none_from = np.float32(2.0 ** (S + 9)).view(np.uint32)
    if bits >= none_from:
        return KEY_NONE
    u = int(np.uint32(bits).view(np.float32))
    return ((u >> S) << KEY_SHIFT) + (u & ((1 << S) - 1)) + t0


@pytest.mark.parametrize("S", [5, 10, 13, 15])
def test_key_is_exact_and_ordered(S):
    rng = np.random.default_rng(S)
    cases = []
    for h in (0, 1, 2, 127, 128, 129, 255, 256):
        for idx in (0, 1, (1 << S) - 1, (1 << S) // 2):
            q = rng.integers(0, 2, 256)
            t = q.copy()
            flip = rng.choice(256, h, replace=False)
            t[flip] ^= 1
            k = keys_f32(q, t, S, idx)
            assert float(k) == h * 2.0 ** S + idx                      # exact in f32: below 2^24
            assert widen(k.view(np.uint32), S, t0=7000) == (h << KEY_SHIFT) + idx + 7000
            cases.append(((h, idx), int(k.view(np.uint32))))
    by_key = sorted(cases, key=lambda c: c[1])
    assert [c[0] for c in by_key] == sorted(c[0] for c in cases)       # float bit patterns order like (hamming, index)


@pytest.mark.parametrize("S", [5, 15])
def test_rows_past_the_chunk_decode_to_none(S):
    rng = np.random.default_rng(100 + S)
    q = rng.integers(0, 2, 256)
    for t in (q, 1 - q, rng.integers(0, 2, 256)):                       # hamming 0 (the strongest competitor), 256, random
        k = keys_f32(q, t, S, (1 << S) - 1, penalised=True)
        assert float(k) >= 2.0 ** (S + 9)                               # above every real key (< 2^(S+8) + 2^S) ...
        assert widen(k.view(np.uint32), S) == KEY_NONE                  # ... and decoded as "no train descriptor"
    real_max = keys_f32(q, 1 - q, S, (1 << S) - 1)
    assert float(real_max) < 2.0 ** (S + 9) and widen(real_max.view(np.uint32), S) == (256 << KEY_SHIFT) + (1 << S) - 1


def test_nibble_table_is_plus_minus_one():
    """bit b -> 0x2 | b << 3: E2M1 (sign, 2 exponent bits, 1 mantissa bit) 0b0010 = +1.0 (bit clear), 0b1010 = -1.0 (bit set); which sign a set bit
    takes does not matter: both operands use the same map and the trains are looked up with ~byte"""
    def e2m1(n):
        sign, e, m = n >> 3, (n >> 1) & 3, n & 1
        v = (m * 0.5) if e == 0 else (1 + m * 0.5) * 2.0 ** (e - 1)
        return -v if sign else v
    assert e2m1(0x2) == 1.0 and e2m1(0xA) == -1.0
    for v in (0, 1, 0x80, 0xA5, 0xFF):
        w = 0
        for j in range(8):
            w |= (0x2 | (((v >> j) & 1) << 3)) << (4 * j)
        assert [e2m1((w >> (4 * j)) & 15) for j in range(8)] == [-1.0 if (v >> j) & 1 else 1.0 for j in range(8)]
