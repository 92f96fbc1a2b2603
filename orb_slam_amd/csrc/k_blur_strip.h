// blur_strip: one wave's strip of the register-resident GaussianBlur (shared by k_blur in k_blur.hip and by k_fast_blur in k_fast.hip, the one-frame
// call's fused FAST + blur launch).
#pragma once
#include "orbx_device.h"

namespace orbx {

// ------------------------------------------------------------------------------------ blur
// GaussianBlur 7x7 sigma 2 (8U fixed point, taps [18,34,49,55,49,34,18]/256 twice, 16 fractional bits).
// Register-resident separable filter, no LDS: one wave owns a 248-px wide column strip and streams down
// ROWS output rows.  Lane j holds one dword (4 pixels) of the current row; the neighbouring dwords
// come from lanes j-1 / j+1 by DPP wave shifts; the 7 taps of each of the lane's pixels are byte-weight dwords over the
// three aligned dwords (v_dot4_u32_u8, no shifted copies).  The last row sums live in registers as row pairs (loop fully
// unrolled), so the column pass is three v_dot2_u32_u16 + one multiply-add per pixel; each lane stores its 4 output pixels as
// one dword.  Details at blur_strip below and in NOTES.md 4.2.
// Reads the UNBLURRED plane and writes a separate blurred plane, which is what the reference's in-place
// filter computes (its border taps read the unblurred reflect-101 border; here: reflect-101 index math).
constexpr int BLUR_STRIP_DW = 62;   // useful dwords per wave (lanes 1..62; lanes 0 and 63 are halo)

__device__ __forceinline__ uint32_t load_px4_reflect(const uint8_t* row, int x, int w) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint32_t)row[reflect101(x + i, w)] << (8 * i);
    return v;
}

template <bool ALIGNED, int ROWS>
__device__ __forceinline__ void blur_strip(const Batch& b, int frame, int t) {   // t = strip index within the frame (wave-uniform)
    const DevGeom& g = b.g;
    constexpr bool SHORT = ROWS != BLUR_ROWS;       // the tiling with short strips (launch_extract picks it for small launch groups)
    const int level = find_level(SHORT ? g.btile_bases_s : g.btile_bases, t);
    const LevelGeom& L = g.lv[level];
    const int tl = t - (SHORT ? L.btile_base_s : L.btile_base);
    const int band = tl / L.btiles_x, strip = tl - band * L.btiles_x;
    const int lane = threadIdx.x & 63;
    const int x = (strip * BLUR_STRIP_DW + lane - 1) * 4;      // first pixel of this lane's dword (may be < 0)
    const int y0 = band * ROWS;
    const int w = L.w, h = L.h;
    long long stride;
    const uint8_t* src = plain_plane(b, L, level, frame, stride);
    uint8_t* dst = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    const bool fetch = x >= -4 && x < w + 4;                   // halo lanes beyond that are never consumed
    const int xq = (w - 1) & ~3;                               // first pixel of the last (possibly partial) dword of a row
    const int xl = x < 0 ? 0 : (x > xq ? xq : x);
    const bool is_left = x == -4, is_last = x == xq, is_halo = x == xq + 4;
    const bool writer = lane >= 1 && lane <= BLUR_STRIP_DW && x < w;
    const uint32_t tew = x < L.blur_wvec ? 1u : 0u;           // ties-to-even columns (blur_wvec is a multiple of 4); others round half-up
    const uint32_t nte = 1u - tew;
    // horizontal taps [18,34,49,55,49,34,18] of output pixel i (0..3 of the lane's dword C) as byte weights over the three aligned
    // dwords L | C | R: v_dot4_u32_u8 needs no byte-shifted copies of the data (10 dot4 per row; cutting the tap windows out with
    // v_alignbyte first took 6 + 8 instructions)
    constexpr uint32_t WL0 = 0x31221200u, WC0 = 0x12223137u;
    constexpr uint32_t WL1 = 0x22120000u, WC1 = 0x22313731u, WR1 = 0x00000012u;
    constexpr uint32_t WL2 = 0x12000000u, WC2 = 0x31373122u, WR2 = 0x00001222u;
    constexpr uint32_t WC3 = 0x37312212u, WR3 = 0x00122231u;
    // the three kinds of border lanes build their reflect-101 bytes from their own dword and one a lane or two to the left: one
    // ds_bpermute (the LDS crossbar, idle in this kernel) + one v_perm with per-lane source lane and selector, for every lane alike
    const int bp_addr = (lane - (is_last ? 1 : is_halo ? 2 : 0)) * 4;
    const uint32_t bsel = is_left ? 0x01020300u                          // px -3..-1 <- px 3,2,1 of dword 0
                        : is_last ? (uint32_t)L.blur_sel_last            // (D_last, D_prev)
                        : is_halo ? (uint32_t)L.blur_sel_halo            // the lane's own load is D_last (clamped)
                                  : 0x07060504u;                         // every other lane: its own dword
    uint32_t pp[6][4];            // pp[r % 6] = (row r-1 | row r << 16) of the lane's 4 pixels
    uint32_t prev[4] = {0, 0, 0, 0};
    // the row loads run BLUR_AHEAD rows ahead of their use (the loop is fully unrolled, but each row's store sits in its own
    // basic block and the compiler issues a row's load right before its first use otherwise: one exposed round trip per row)
    constexpr int BLUR_AHEAD = 3;
    auto load_row = [&](int r) -> uint32_t {
        const int yy = reflect101(y0 + r - 3, h);
        const uint8_t* row = src + (long long)yy * stride;
        if (ALIGNED) return *reinterpret_cast<const uint32_t*>(row + xl);          // every lane loads an aligned dword (x clamped into the row)
        return fetch ? load_px4_reflect(row, x, w) : 0u;
    };
    uint32_t ahead[BLUR_AHEAD];
#pragma unroll
    for (int r = 0; r < BLUR_AHEAD; r++) ahead[r] = load_row(r);
#pragma unroll
    for (int r = 0; r < ROWS + 6; r++) {
        const uint32_t Craw = ahead[r % BLUR_AHEAD];
        if (r + BLUR_AHEAD < ROWS + 6) ahead[r % BLUR_AHEAD] = load_row(r + BLUR_AHEAD);
        uint32_t C = Craw;
        if (ALIGNED) {
            const uint32_t X = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_addr, (int)Craw);
            C = __builtin_amdgcn_perm(Craw, X, bsel);
        }
        const uint32_t Lw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x138, 0xf, 0xf, true);   // wave_shr:1  <- lane-1
        const uint32_t R = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C, 0x130, 0xf, 0xf, true);    // wave_shl:1  <- lane+1
        uint32_t cur[4];
        cur[0] = __builtin_amdgcn_udot4(C, WC0, __builtin_amdgcn_udot4(Lw, WL0, 0u, false), false);
        cur[1] = __builtin_amdgcn_udot4(R, WR1, __builtin_amdgcn_udot4(C, WC1, __builtin_amdgcn_udot4(Lw, WL1, 0u, false), false), false);
        cur[2] = __builtin_amdgcn_udot4(R, WR2, __builtin_amdgcn_udot4(C, WC2, __builtin_amdgcn_udot4(Lw, WL2, 0u, false), false), false);
        cur[3] = __builtin_amdgcn_udot4(R, WR3, __builtin_amdgcn_udot4(C, WC3, 0u, false), false);
        // Vertical pass.  A row sum is at most 255 * 257 = 65535, so two consecutive rows of one pixel fit one dword and
        // v_dot2_u32_u16 takes two taps per instruction: with pair(r) = (row r-1 | row r << 16) the output of rows r-6 .. r is
        //   dot2(pair(r-5), 18|34) + dot2(pair(r-3), 49|55) + dot2(pair(r-1), 49|34) + 18 * row r        (4 ops + 1 pack per pixel)
        // Rounding (orb_math.h blur_round): with t = sum + 0x7FFF both modes are (t + bit) >> 16, bit = bit 16 of t in the
        // ties-to-even columns (the same carry behaviour as the parity of sum's integer part: when they differ, the low half
        // of t cannot carry) and 1 in the half-up ones: v_bfe with a per-lane width of 1 or 0, then one v_add3.
        if (r >= 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) pp[r % 6][i] = prev[i] | (cur[i] << 16);
        }
        if (r >= 6) {
            const int oy = y0 + r - 6;
            uint32_t q[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t t = __umul24(cur[i], (uint32_t)ORBX_G0) + 0x7FFFu;
                t = __builtin_amdgcn_udot2(as_us2v(pp[(r - 5) % 6][i]), as_us2v((uint32_t)ORBX_G0 | ((uint32_t)ORBX_G1 << 16)), t, false);
                t = __builtin_amdgcn_udot2(as_us2v(pp[(r - 3) % 6][i]), as_us2v((uint32_t)ORBX_G2 | ((uint32_t)ORBX_G3 << 16)), t, false);
                t = __builtin_amdgcn_udot2(as_us2v(pp[(r - 1) % 6][i]), as_us2v((uint32_t)ORBX_G2 | ((uint32_t)ORBX_G1 << 16)), t, false);
                q[i] = t + __builtin_amdgcn_ubfe(t, 16u, tew) + nte;
            }
            // (q >> 16) of two pixels per dword, saturated to 255 as packed 16-bit (the taps sum to 257 per pass: 254 and 255 overshoot)
            const us2v lo2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(q[1], q[0], 0x07060302u)), as_us2v(0x00FF00FFu));
            const us2v hi2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(q[3], q[2], 0x07060302u)), as_us2v(0x00FF00FFu));
            const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi2), __builtin_bit_cast(uint32_t, lo2), 0x06040200u);
            if (writer && oy < h) *reinterpret_cast<uint32_t*>(dst + (long long)oy * L.stride + x) = packed;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) prev[i] = cur[i];
    }
}


}  // namespace orbx
