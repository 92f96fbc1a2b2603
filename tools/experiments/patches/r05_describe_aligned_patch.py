"""Round 5, call 15 (NOTES.md 10.5, VERDICT r04 #4 ii): k_describe's IC_Angle patch of levels >= 1 from DWORD-ALIGNED addresses — 36 bytes
per row from the dword at or left of x - 15, three lanes x 12 bytes (global_load_dwordx3), six loads per lane; the data stays where it lands,
the circle masks (one v_perm of two dwords of a 12-dword-pitch table) and the (u + 15) weights are shifted by the misalignment instead.
Bit-exact (115 GPU tests, 40 fuzz launch groups; the arithmetic also emulated on the CPU).  Measured, every kernel alone: k_describe 0.6815 /
0.6848 -> 0.6902 / 0.6971 ms per 1024 VGA frames (+1.5 %), 1080p 0.5215 -> 0.5236, four-lane line 365.6k -> 364.6k frames/s: the ~150
texture-addresser cycles per wave it saves are paid back by two more vector loads, 20 more LDS reads and ~95 more VALU instructions per
lane.  NOT in the product.  Applies to the sources of commit 0372782 (python tools/experiments/patches/r05_describe_aligned_patch.py from the
repository root, then build with -DORBX_DESC_ALIGNED_PATCH=1); a record of what was measured, not maintained against later edits."""
p='orb_slam_amd/csrc/orbx_kernels.hip'
s=open(p).read()
def rep(old,new):
    global s
    assert s.count(old)==1,(s.count(old),old[:80])
    s=s.replace(old,new)
rep('''    __shared__ __attribute__((aligned(16))) uint32_t s_mask[256];       // circle byte masks of the 31 x 8 patch dwords (slots 248.. = 0)''',
'''#ifndef ORBX_DESC_ALIGNED_PATCH
#define ORBX_DESC_ALIGNED_PATCH 0
#endif
    constexpr int MASK_PITCH = ORBX_DESC_ALIGNED_PATCH ? 12 : 8;        // dwords per row of the circle mask table (aligned patch: 8 mask dwords + 4 zero dwords)
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[32 * MASK_PITCH];       // circle byte masks of the 31 x 8 patch dwords (row 31 and the pad dwords = 0)''')
# table build: generalise to the pitch
rep('''    for (int t = tid; t < 256; t += DESC_WAVES * 64) {
        const uint32_t pk = (ORBX_DESC_EARLY_PATTERN && t == tid) ? pk_first : c_pattern[t];
#if ORBX_DESC_PACKED_PATTERN
        s_pat[t] = pk;
#else
        reinterpret_cast<float4*>(s_pat)[t] = make_float4((float)(int)(int8_t)pk, (float)(int)(int8_t)(pk >> 8), (float)(int)(int8_t)(pk >> 16), (float)(int)(int8_t)(pk >> 24));
#endif
        // umax[] (reference :495-510) depends only on HALF_PATCH_SIZE = 15: nibble v of UMAX_NIBBLES (the host checks it against the computed table)
        const int r = t >> 3, c = t & 7;
        const int v = r - HALF_PATCH, av = v < 0 ? -v : v;
        const int um = r < 31 ? (int)((UMAX_NIBBLES >> (4 * (av & 15))) & 15ull) : -1;
        uint32_t mask = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int u = 4 * c + kk - HALF_PATCH;
            if ((u < 0 ? -u : u) <= um) mask |= 0xFFu << (8 * kk);
        }
        s_mask[t] = mask;
    }''','''    for (int t = tid; t < 32 * MASK_PITCH; t += DESC_WAVES * 64) {
        if (t < 256) {
            const uint32_t pk = (ORBX_DESC_EARLY_PATTERN && t == tid) ? pk_first : c_pattern[t];
#if ORBX_DESC_PACKED_PATTERN
            s_pat[t] = pk;
#else
            reinterpret_cast<float4*>(s_pat)[t] = make_float4((float)(int)(int8_t)pk, (float)(int)(int8_t)(pk >> 8), (float)(int)(int8_t)(pk >> 16), (float)(int)(int8_t)(pk >> 24));
#endif
        }
        // umax[] (reference :495-510) depends only on HALF_PATCH_SIZE = 15: nibble v of UMAX_NIBBLES (the host checks it against the computed table)
        const int r = MASK_PITCH == 8 ? t >> 3 : (t * 171) >> 11, c = t - MASK_PITCH * r;      // t / 12 for t < 384
        const int v = r - HALF_PATCH, av = v < 0 ? -v : v;
        const int um = r < 31 && c < 8 ? (int)((UMAX_NIBBLES >> (4 * (av & 15))) & 15ull) : -1;
        uint32_t mask = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int u = 4 * c + kk - HALF_PATCH;
            if ((u < 0 ? -u : u) <= um) mask |= 0xFFu << (8 * kk);
        }
        s_mask[t] = mask;
    }''')
# IC_Angle: aligned path for levels >= 1
rep('''        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        const int rsub = li >> 1, hf = li & 1;''','''        if (ORBX_DESC_ALIGNED_PATCH && level != 0) {
            // (round 5) Levels >= 1 (planes of our own, rows padded to 64 bytes): the patch from DWORD-ALIGNED addresses — 36 bytes per row from the
            // dword at or left of x - 15, three lanes x 12 bytes, 6 loads per lane over the 31 rows — because the texture addresser prices a
            // byte-aligned 16-byte row piece at 4.0 cycles and an aligned one at 2.4 (profiles/r04_ta_shapes.txt).  The data stays where it
            // lands; the circle masks and the (u + 15) weights are shifted instead: mask dword i of the shifted stream = bytes 4 i - sh .. of the
            // row's mask stream (one v_perm of two table dwords), weights minus sh per byte.  Level 0 keeps the byte-aligned loads: its rows
            // are the caller's, and 36 bytes from an aligned start can end 3 bytes past what include/orbx.h promises readable.
            typedef uint32_t u32x3_a __attribute__((ext_vector_type(3), aligned(4)));
            const int xa = (x - HALF_PATCH) & ~3, sh = (x - HALF_PATCH) & 3;
            const uint32_t psel = 0x07060504u - (uint32_t)sh * 0x01010101u;
            int rr[6], cc[6];
            u32x3_a Q[6];
#pragma unroll
            for (int n = 0; n < 6; n++) {
                const int t = li + 16 * n;
                rr[n] = (t * 171) >> 9;                       // t / 3 for t < 96: rows 0 .. 31 (row 31 is masked, still inside the level)
                cc[n] = t - 3 * rr[n];
                Q[n] = *reinterpret_cast<const u32x3_a*>(plain + ((unsigned)(xa + 12 * cc[n]) + __umul24((unsigned)(y - HALF_PATCH + rr[n]), pstride)));
            }
            uint32_t a_su = 0, a_si = 0, a_r = 0;
#pragma unroll
            for (int n = 0; n < 6; n++) {
                const uint32_t* mrow = s_mask + MASK_PITCH * rr[n] + 3 * cc[n];
                uint32_t mprev = mrow[cc[n] ? -1 : 0];
                mprev = cc[n] ? mprev : 0u;
                const uint32_t m0 = mrow[0], m1 = mrow[1], m2 = mrow[2];
                const uint32_t mm[3] = {__builtin_amdgcn_perm(m0, mprev, psel), __builtin_amdgcn_perm(m1, m0, psel), __builtin_amdgcn_perm(m2, m1, psel)};
                // u + 15 = 12 c + 4 d + j - sh of the four pixels of dword d: bytes without carries except in the very first dword of a row
                // (12 c + 4 d - sh < 0), whose bytes j < sh are masked: there the weights j - sh are 0x03020100 shifted up by sh bytes
                const uint32_t wfirst = 0x03020100u << (8 * sh);
                const uint32_t w1 = (uint32_t)(12 * cc[n] + 4 - sh) * 0x01010101u + 0x03020100u;
                const uint32_t ww[3] = {cc[n] ? w1 - 0x04040404u : wfirst, w1, w1 + 0x04040404u};
                uint32_t srow = 0;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const uint32_t Im = Q[n][d] & mm[d];
                    srow = __builtin_amdgcn_udot4(Im, 0x01010101u, srow, false);
                    a_su = __builtin_amdgcn_udot4(Im, ww[d], a_su, false);
                }
                a_si += srow;
                a_r = __umul24(srow, (uint32_t)rr[n]) + a_r;
            }
            const int p10 = (int)a_su - HALF_PATCH * (int)a_si;
            const int p01 = (int)a_r - HALF_PATCH * (int)a_si;
            m10 = row16_sum(p10); m01 = row16_sum(p01);
        } else {
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        const int rsub = li >> 1, hf = li & 1;''')
rep('''            const uint4 mk = *reinterpret_cast<const uint4*>(s_mask + (8 * (8 * n + rsub) + 4 * hf));''','''            const uint4 mk = *reinterpret_cast<const uint4*>(s_mask + (MASK_PITCH * (8 * n + rsub) + 4 * hf));''')
rep('''        const int p01 = (rsub - HALF_PATCH) * (int)a_si + (int)a_r;
        m10 = row16_sum(p10); m01 = row16_sum(p01);
    }''','''        const int p01 = (rsub - HALF_PATCH) * (int)a_si + (int)a_r;
        m10 = row16_sum(p10); m01 = row16_sum(p01);
        }
    }''')
open(p,'w').write(s)
