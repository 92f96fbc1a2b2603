// GaussianBlur 7x7 sigma 2 over whole level planes for gfx950 (reference src/ORBextractor.cc:760; OpenCV 2.4's 8-bit fixed-point filter): k_blur (register-
// resident separable filter on the VALU) and k_blur_mfma (both passes as exact int8 matrix products).  Since round 6 full launch groups blur per keypoint
// window instead (k_describe_od.hip); these kernels serve the one-frame call, ORBX_BLUR_ON_DEMAND=0 and the stage dumps.
#include <algorithm>
#include <type_traits>

#include "orbx_device.h"
#include "orbx_launch.h"
#include "k_blur_strip.h"

namespace orbx {

template <bool ALIGNED, int ROWS>
__global__ __launch_bounds__(BLUR_WAVES * 64) void k_blur(Batch b) {
    const DevGeom& g = b.g;
    const int ntiles = ROWS != BLUR_ROWS ? g.nbtiles_total_s : g.nbtiles_total;
    int frame, wgi;
    if (!frame_item(b, blockIdx.x, (ntiles + BLUR_WAVES - 1) / BLUR_WAVES, frame, wgi)) return;
    const int t = wgi * BLUR_WAVES + wave_id();
    if (t < ntiles) blur_strip<ALIGNED, ROWS>(b, frame, t);
}

// ------------------------------------------------------------------------------------ blur on the matrix cores (round 4)
// The same filter as exact int8 matrix products per 32 x 32 tile (v_mfma_i32_32x32x32_i8, i32 accumulate).  k_blur spends 19.7
// lane-operations per pixel on the taps (v_dot4 / v_dot2) and is VALU-issue bound at 0.55 ms per 1024 VGA frames while the matrix
// pipe idles; here the taps are banded 0 / 18 / 34 / 49 / 55 matrices held in registers, the VALU only converts between the passes
// and rounds (about 7 lane-operations per pixel), and the floor becomes the HBM time of reading and writing the pyramid once
// (2 P_total bytes: 0.31 ms per 1024 VGA frames at 6.3 TB/s).  tools/proto/blur_mfma_emulation.py is the integer model of this data
// flow, checked against the oracle.
//
// One wave = one 64-pixel strip (two 32-pixel tiles) of a level, streamed down in steps of 32 rows.  Operand slots (probe:
// profiles/r02_mfma_layout.txt): lane (i, g) = i + 32 g holds row i of A (column i of B) and 16 of the 32 k-values; which k-value a
// byte slot stands for is ours to choose as long as A and B agree; lane (n, g) of D holds rows 8 (r / 4) + 4 g + r % 4 in register r.
//   row pass     D[row][c] = sum_k I[row][k] T[k][c] over the 64 input columns X - 16 .. X + 47 of a tile (two MFMAs): A = pixels
//                minus 128, lane = row, slot = column: whole aligned 16-byte chunks.  B = the horizontal taps of output column pi(c);
//                reflect-101 at the level's edges is folded into this matrix (a reflected tap adds its weight to the column it
//                lands on).  Column X - 16 is never tapped: its slot carries the constant 64 with weight 2, so D = S - 32896 + 128
//                = Z with S the 16-bit row sum of the reference and Z in [-32768, 32767].
//   split        Z = 256 hi + lo + 128 with hi = Z >> 8 and lo = (Z & 255) - 128 both in int8: register r of a lane (rows 8 i +
//                4 g + j of ITS column) becomes byte j of operand dword i — the column pass contracts over rows, and the slots a
//                lane holds after the row pass are exactly the k-slots its lane group needs: no data moves between lanes.
//   column pass  D2[c][y] = sum_rho H[rho][c] W[rho][y] over the previous and the current row tile (rows Y0 - 29 .. Y0 + 34 cover
//                the taps of output rows Y0 .. Y0 + 31): A = hi (then lo) bytes, lane = column index c, B = vertical taps, lane =
//                output row.  256 HI + LO + 257 * 32896 is the reference's 32-bit column sum; the shifted HI accumulator plus the
//                rounding constant seeds the LO products, so the epilogue is one v_bfe + v_add (ties-to-even columns) per pixel.
//   pi           column index 8 i + 4 g + j <-> tile column 16 g + 4 i + j: the 16 registers of a lane of D2 are 16 CONTIGUOUS
//                pixels of its row.
// Memory side.  An MFMA operand wants a ROW per lane, i.e. 32 (or 64) scattered 16-byte accesses per wave instruction: the first
// form of this kernel loaded and stored that way and was bound by the address coalescer at 0.9 ms (0.50 ms with the stores and
// the row scatter taken out, NOTES.md 9.2).  So both directions go through a wave-private LDS area in row-major order: the input
// rows arrive by LDS-DMA in runs of 96 contiguous bytes (a step ahead, two buffers; no VGPR-destination load anywhere, so the one
// wait per step is the explicit vmcnt(0) at its top, which also covers the previous step's stores — gfx9-family stores count on
// vmcnt and complete out of order with loads), the operands are ds_read_b128, the output tiles are written to LDS as 16 bytes per
// lane and leave as 64 contiguous bytes per row.  No workgroup barrier: the waves of a workgroup only share the launch.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int MB_IN_CHUNKS = 2 * MB_TILES + 2;                       // 16-byte chunks per staged input row: the strip + 16 columns either side
constexpr int MB_IN_BYTES = 32 * MB_IN_CHUNKS * 16;                  // one input buffer: 32 rows
constexpr int MB_OUT_PITCH = MB_TILES * 32 + 16;                     // bytes per row of the staged output (80: conflict-free ds_write_b128 of a row per lane)
static_assert(MB_TILES == 2, "k_blur_mfma's lane maps are written for two tiles per strip");

__global__ __launch_bounds__(MB_WAVES * 64) void k_blur_mfma(Batch b) {      // (134 VGPRs: three waves per SIMD, each with two independent chains)
    // three LDS objects on purpose: hipcc orders a ds_read behind an outstanding LDS-DMA (s_waitcnt vmcnt(0)) unless it can prove that
    // the two do not alias, which it can for distinct objects only — with one array the rows requested for the NEXT step were drained
    // in front of the first operand read of THIS step
    __shared__ __attribute__((aligned(16))) uint8_t mb_in0[MB_WAVES * MB_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_in1[MB_WAVES * MB_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_out[MB_WAVES * 32 * MB_OUT_PITCH];
    typedef const void __attribute__((address_space(1))) * gptr_t;
    typedef void __attribute__((address_space(3))) * lptr_t;
    const DevGeom& g = b.g;
    int frame, wgi;
    if (!frame_item(b, blockIdx.x, (g.nmb_total + MB_WAVES - 1) / MB_WAVES, frame, wgi)) return;
    const int item = wgi * MB_WAVES + wave_id();
    if (item >= g.nmb_total) return;
    const int level = __builtin_amdgcn_readfirstlane(find_level(g.mb_bases, item));
    const LevelGeom& L = g.lv[level];
    const int lane = threadIdx.x & 63, m = lane & 31, gg = lane >> 5;
    const int w = L.w, h = L.h;
    const int band = (item - L.mb_base) / L.mb_strips;           // (wave-uniform)
    const int X0 = 64 * ((item - L.mb_base) - band * L.mb_strips);       // first column of the strip
    const int Ybeg = 32 * band * L.mb_band_steps, Yend = min(h, Ybeg + 32 * L.mb_band_steps);   // the band's output rows
    long long sstride64;
    const uint8_t* src = plain_plane(b, L, level, frame, sstride64);
    const unsigned sstride = (unsigned)sstride64;                // rows < 2^24 bytes, planes < 2^31 (host-checked); a multiple of 16 on this path
    uint8_t* dst = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    uint8_t* const in0 = mb_in0 + wave_id() * MB_IN_BYTES;
    uint8_t* const in1 = mb_in1 + wave_id() * MB_IN_BYTES;
    uint8_t* const obuf = mb_out + wave_id() * 32 * MB_OUT_PITCH;

    // B operands of the row pass, per tile: lane (c, g), slot (v, bb) <-> input column X - 16 + 16 g + 4 v + bb (first MFMA) / + 32 (second)
    v4i Ta[MB_TILES], Tb[MB_TILES];
    {
        const int ci = m >> 3, cg = (m >> 2) & 1, cj = m & 3;
        const int o = 16 * cg + 4 * ci + cj;                     // pi(c)
#pragma unroll
        for (int j = 0; j < MB_TILES; j++) {
            const int X = X0 + 32 * j;
            int ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
            if (X >= 3 && X + 32 + 3 <= w) {                     // inner tile (wave-uniform): tap t of column o sits on input column k = o + 13 + t,
#pragma unroll                                                   //  so every operand dword is a 4-byte window of the tap string
                for (int v = 0; v < 4; v++) {
                    ta[v] = gauss7_taps4(16 * gg + 4 * v - o - 13);
                    tb[v] = gauss7_taps4(32 + 16 * gg + 4 * v - o - 13);
                }
            } else if (X + o < w) {                              // edge tile (output columns beyond the level get no taps: never stored past the row padding)
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    int x = X + o - 3 + t;                       // one reflection suffices: |offset| <= 3 < w
                    x = x < 0 ? -x : (x >= w ? 2 * w - 2 - x : x);
                    const int k = x - (X - 16);                  // 0 .. 63 by construction
                    const int wgt = ((k >> 4) & 1) == gg ? gauss7_tap(t) << (8 * (k & 3)) : 0;
                    const int v = (k >> 2) & 3;
#pragma unroll
                    for (int vv = 0; vv < 4; vv++) {
                        ta[vv] += (k < 32 && v == vv) ? wgt : 0;
                        tb[vv] += (k >= 32 && v == vv) ? wgt : 0;
                    }
                }
            }
            if (gg == 0) ta[0] += 2;                             // x the constant 64 in slot 0 of the first operand
            Ta[j] = (v4i){ta[0], ta[1], ta[2], ta[3]};
            Tb[j] = (v4i){tb[0], tb[1], tb[2], tb[3]};
        }
    }
    // B operands of the column pass: lane (y, g), slot (v, bb) <-> input row rho = 8 v + 4 g + bb of the current (previous) row tile;
    // output row m takes tap rho - m + 6 of the current tile and tap rho - m - 26 of the previous one
    v4i Wc, Wp;
    {
        int wc[4], wp[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            wc[v] = gauss7_taps4(8 * v + 4 * gg - m + 6);
            wp[v] = gauss7_taps4(8 * v + 4 * gg - m - 26);
        }
        Wc = (v4i){wc[0], wc[1], wc[2], wc[3]};
        Wp = (v4i){wp[0], wp[1], wp[2], wp[3]};
    }
    // ties-to-even flags of the lane's output dwords (orb_math.h blur_round; blur_wvec is a multiple of 4): bit 4 j + i
    uint32_t tewmask = 0;
#pragma unroll
    for (int j = 0; j < MB_TILES; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) tewmask |= (X0 + 32 * j + 16 * gg + 4 * i < L.blur_wvec ? 1u : 0u) << (4 * j + i);
    // LDS-DMA of one row tile: 32 rows x MB_IN_CHUNKS chunks, chunk q = 64 n + lane of the buffer = row q / 6, chunk q % 6 of the row
    int dma_c[3];
#pragma unroll
    for (int n = 0; n < 3; n++) {
        const int q = 64 * n + lane, r = (q * 171) >> 10;        // q / 6 for q < 192
        const int ca = (X0 >> 4) - 1 + (q - 6 * r);              // absolute chunk of the row; chunks outside it are clamped (no tap reaches them)
        // level 0 is the caller's frame: only min(row_stride, w rounded up to 16) bytes of a row are promised readable (include/orbx.h), so the
        // clamp stops there (with the clamp at row_stride an ROI at the right edge of a wider image was read past its last row)
        dma_c[n] = 16 * min(max(ca, 0), min((int)(sstride >> 4), (w + 15) >> 4) - 1);
    }
    auto dma_tile = [&](int R, uint8_t* ibuf) {                  // rows R .. R + 31 (reflect-101; rows no tap reaches are clamped into the level)
#pragma unroll
        for (int n = 0; n < 3; n++) {
            int row = R + (((64 * n + lane) * 171) >> 10);       // (recomputed: the kernel sits at its 128-register budget)
            row = row < 0 ? -row : row;
            row = row >= h ? 2 * h - 2 - row : row;
            row = min(max(row, 0), h - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (__umul24((unsigned)row, sstride) + (unsigned)dma_c[n])), (lptr_t)(ibuf + 1024 * n), 16, 0, 0);
        }
    };
    // row pass + split of tile j of the staged buffer: hi / lo operand dwords of the column pass
    auto row_pass = [&](const uint8_t* ibuf, int j, v4i& hi, v4i& lo, const v4i* pre = nullptr) {
        // The operand reads are inline assembly: hipcc orders its own ds_read / ds_write behind every outstanding LDS-DMA (vmcnt(0)) where it
        // cannot prove that they do not alias — here that drained the rows requested for the NEXT step in front of this step's reads.
        // What these reads depend on (this step's DMA) is covered by the explicit wait at the top of the step.
        const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(ibuf + (m * MB_IN_CHUNKS + 2 * j + gg) * 16);
        v4i p1, p2;
        if (pre) { p1 = pre[0]; p2 = pre[1]; }                   // (already read: the step fetches the operands of both tiles in one go)
        else asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p1), "=&v"(p2) : "v"(ra) : "memory");
        v4i a1, a2;
        a1.x = (int)((uint32_t)p1.x ^ 0x80808080u);
        a1.x = gg == 0 ? (int)(((uint32_t)a1.x & 0xFFFFFF00u) | 0x40u) : a1.x;      // the constant slot
        a1.y = (int)((uint32_t)p1.y ^ 0x80808080u); a1.z = (int)((uint32_t)p1.z ^ 0x80808080u); a1.w = (int)((uint32_t)p1.w ^ 0x80808080u);
        a2.x = (int)((uint32_t)p2.x ^ 0x80808080u); a2.y = (int)((uint32_t)p2.y ^ 0x80808080u); a2.z = (int)((uint32_t)p2.z ^ 0x80808080u); a2.w = (int)((uint32_t)p2.w ^ 0x80808080u);
        v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, Ta[j], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, Tb[j], z, 0, 0, 0);
        int h4[4], l4[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)z[4 * i + 1], (uint32_t)z[4 * i], 0x05010400u);      // lo0 lo1 hi0 hi1
            const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)z[4 * i + 3], (uint32_t)z[4 * i + 2], 0x05010400u);
            l4[i] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
            h4[i] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
        }
        hi = (v4i){h4[0], h4[1], h4[2], h4[3]};
        lo = (v4i){l4[0], l4[1], l4[2], l4[3]};
    };
    const bool tile1 = X0 + 32 < w;                              // the strip's second tile exists (wave-uniform)
    v4i phi[MB_TILES], plo[MB_TILES];
    dma_tile(Ybeg + 3 - 32, in1);                                // rows Ybeg - 29 .. Ybeg + 2: the taps above the band's first output rows
    dma_tile(Ybeg + 3, in0);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");             // LDS-DMA returns in order: the first three instructions (buffer 1) have landed
    row_pass(in1, 0, phi[0], plo[0]);
    if (tile1) row_pass(in1, 1, phi[1], plo[1]);
    // flush of the staged output block (32 rows x 64 bytes): chunk q = 64 n + lane = row q / 4, chunk q % 4
    const int fl_row = lane >> 2, fl_c = 16 * (lane & 3);
    const bool fl_on = X0 + fl_c < (int)L.stride && X0 + fl_c < ((w + 15) & ~15);
    auto flush = [&](int Yb) {                                   // the block of output rows Yb .. Yb + 31 leaves as 64 contiguous bytes per row
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int row = 16 * n + fl_row, oy = Yb + row;
            const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(obuf + row * MB_OUT_PITCH + fl_c, 16));
            if (fl_on && oy < Yend) *reinterpret_cast<uint4*>(__builtin_assume_aligned(dst + (__umul24((unsigned)oy, (unsigned)L.stride) + (unsigned)(X0 + fl_c)), 16)) = v;
        }
    };
    auto step = [&](int Y0, const uint8_t* cur, uint8_t* nxt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this step's rows have landed (requested a step ago), the stores of the step before
        if (Y0 > Ybeg) flush(Y0 - 32);                                   // are done, the output tiles of the step before are in LDS
        uint32_t tm = tewmask;
        asm volatile("" : "+v"(tm));                             // (opaque per step: the 16 flag values derived from it are not worth 16 registers held across the loop)
        dma_tile(Y0 + 32 + 3, nxt);                              // the next step's rows (unconditional: behind the last step it re-reads clamped rows nobody uses —
                                                                 //  a branch here makes hipcc drain the DMA at the join, in front of this step's operand reads)
        // the operands of both tiles in one LDS round trip: chunks g, 2 + g, 4 + g of the lane's row — tile 0 takes the first two, tile 1 the
        // last two (the strip's second tile starts where the first one's second operand does)
        v4i pre[3];
        {
            const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(cur + (m * MB_IN_CHUNKS + gg) * 16);
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:32\n\tds_read_b128 %2, %3 offset:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(pre[0]), "=&v"(pre[1]), "=&v"(pre[2]) : "v"(ra) : "memory");
        }
        if (tile1) {
            // both tiles stage by stage: the two chains (row pass -> split -> HI products -> shift -> LO products -> rounding) are independent, and
            // a wave that walks them one after the other leaves the matrix pipe and the VALU waiting on each other's results — 0.52 -> 0.48 ms
            // per 1024 VGA frames although the 32 accumulator registers of two chains cost the fourth wave per SIMD (NOTES.md 9.2)
            auto centre = [&](v4i pv, bool first) -> v4i {
                v4i a;
                a.x = (int)((uint32_t)pv.x ^ 0x80808080u);
                if (first) a.x = gg == 0 ? (int)(((uint32_t)a.x & 0xFFFFFF00u) | 0x40u) : a.x;      // the constant slot
                a.y = (int)((uint32_t)pv.y ^ 0x80808080u); a.z = (int)((uint32_t)pv.z ^ 0x80808080u); a.w = (int)((uint32_t)pv.w ^ 0x80808080u);
                return a;
            };
            auto split = [&](const v16i& z, v4i& hi, v4i& lo) {
                int h4[4], l4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)z[4 * i + 1], (uint32_t)z[4 * i], 0x05010400u);
                    const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)z[4 * i + 3], (uint32_t)z[4 * i + 2], 0x05010400u);
                    l4[i] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
                    h4[i] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
                }
                hi = (v4i){h4[0], h4[1], h4[2], h4[3]};
                lo = (v4i){l4[0], l4[1], l4[2], l4[3]};
            };
            auto finish = [&](const v16i& acc, int j) {
                uint32_t o4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t tw = (tm >> (4 * j + i)) & 1u;
                    uint32_t q[4];
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const uint32_t t = (uint32_t)acc[4 * i + jj];
                        q[jj] = t + __builtin_amdgcn_ubfe(t, 16u, tw) + (tw ^ 1u);
                    }
                    const us2v lo2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(q[1], q[0], 0x07060302u)), as_us2v(0x00FF00FFu));
                    const us2v hi2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(q[3], q[2], 0x07060302u)), as_us2v(0x00FF00FFu));
                    o4[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi2), __builtin_bit_cast(uint32_t, lo2), 0x06040200u);
                }
                const unsigned wa = (unsigned)(uintptr_t)(lptr_t)(obuf + m * MB_OUT_PITCH + 32 * j + 16 * gg);
                const v4i ov = {(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(wa), "v"(ov) : "memory");
            };
            const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const v4i a10 = centre(pre[0], true), a20 = centre(pre[1], false), a11 = centre(pre[1], true), a21 = centre(pre[2], false);
            v16i z0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a10, Ta[0], zero, 0, 0, 0);
            v16i z1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a11, Ta[1], zero, 0, 0, 0);
            z0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a20, Tb[0], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a21, Tb[1], z1, 0, 0, 0);
            v16i acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi[0], Wp, zero, 0, 0, 0);      // (the previous tiles' halves of the HI products need nothing of this step)
            v16i acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi[1], Wp, zero, 0, 0, 0);
            v4i chi0, clo0, chi1, clo1;
            split(z0, chi0, clo0);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(chi0, Wc, acc0, 0, 0, 0);
            split(z1, chi1, clo1);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(chi1, Wc, acc1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) acc0[r] = (int)(((uint32_t)acc0[r] << 8) + (uint32_t)(257 * 32896 + 0x7FFF));
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo[0], Wp, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(clo0, Wc, acc0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) acc1[r] = (int)(((uint32_t)acc1[r] << 8) + (uint32_t)(257 * 32896 + 0x7FFF));
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo[1], Wp, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(clo1, Wc, acc1, 0, 0, 0);
            phi[0] = chi0; plo[0] = clo0; phi[1] = chi1; plo[1] = clo1;
            finish(acc0, 0);
            finish(acc1, 1);
            return;
        }
        // a strip whose second tile lies beyond the level: one chain
#pragma unroll
        for (int j = 0; j < 1; j++) {
            v4i chi, clo;
            row_pass(cur, j, chi, clo, pre + j);
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(phi[j], Wp, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(chi, Wc, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)(257 * 32896 + 0x7FFF));     // the reference's offset + the rounding constant
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(plo[j], Wp, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(clo, Wc, acc, 0, 0, 0);
            phi[j] = chi; plo[j] = clo;
            uint32_t o4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t tw = (tm >> (4 * j + i)) & 1u;
                uint32_t q[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const uint32_t t = (uint32_t)acc[4 * i + jj];
                    q[jj] = t + __builtin_amdgcn_ubfe(t, 16u, tw) + (tw ^ 1u);      // + bit 16 (ties to even) or + 1 (half up): one v_add3
                }
                // (q >> 16) of two pixels per dword, saturated to 255 as packed 16-bit (the taps sum to 257 per pass: 254 and 255 overshoot)
                const us2v lo2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(q[1], q[0], 0x07060302u)), as_us2v(0x00FF00FFu));
                const us2v hi2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(q[3], q[2], 0x07060302u)), as_us2v(0x00FF00FFu));
                o4[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi2), __builtin_bit_cast(uint32_t, lo2), 0x06040200u);
            }
            {
                const unsigned wa = (unsigned)(uintptr_t)(lptr_t)(obuf + m * MB_OUT_PITCH + 32 * j + 16 * gg);
                const v4i ov = {(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(wa), "v"(ov) : "memory");      // (read back by flush() behind the lgkmcnt(0) at the top of the next step)
            }
        }
    };
    for (int Y0 = Ybeg; Y0 < Yend; Y0 += 64) {                   // two steps per trip: the buffer of each step is a named LDS object
        step(Y0, in0, in1);
        if (Y0 + 32 < Yend) step(Y0 + 32, in1, in0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    flush(Ybeg + (((Yend - 1 - Ybeg) >> 5) << 5));               // the last step's block
}


int launch_blur(const Batch& b, const HostGeom& hg, hipStream_t st) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
    const bool aligned = (((uintptr_t)b.img | (uintptr_t)b.img_row_stride | (uintptr_t)b.img_frame_stride) & 3) == 0;
    if (F < PYR_FUSED_MAX_FRAMES) {   // short strips: 4x the waves, a quarter of the serial row chain each
        const int nblk = frame_item_blocks(b, (g.nbtiles_total_s + BLUR_WAVES - 1) / BLUR_WAVES);
        if (aligned) hipLaunchKernelGGL((k_blur<true, BLUR_ROWS_SMALL>), dim3(nblk), dim3(BLUR_WAVES * 64), 0, st, b);
        else hipLaunchKernelGGL((k_blur<false, BLUR_ROWS_SMALL>), dim3(nblk), dim3(BLUR_WAVES * 64), 0, st, b);
    } else if (ORBX_BLUR_MFMA && aligned && ((b.img_row_stride | b.img_frame_stride) & 15) == 0 && (g.lv[0].w <= MB_MAX_WIDTH || ORBX_BLUR_MFMA > 1)) {
        // full launch groups whose frames can be staged in whole 16-byte chunks: the filter as int8 matrix products.  Measured
        // (NOTES.md 9.2): 0.50 against 0.55 ms per 1024 VGA frames and half the VALU instructions, which the lanes next to it pick
        // up (+1.5 % frames/s); on 1920-byte rows its 96-byte row pieces lose to k_blur's 256-byte ones (0.88 against 0.78 ms per
        // 256 1080p frames), so wide levels keep k_blur
        hipLaunchKernelGGL(k_blur_mfma, dim3(frame_item_blocks(b, (g.nmb_total + MB_WAVES - 1) / MB_WAVES)), dim3(MB_WAVES * 64), 0, st, b);
    } else {
        const int nblk = frame_item_blocks(b, (g.nbtiles_total + BLUR_WAVES - 1) / BLUR_WAVES);
        if (aligned) hipLaunchKernelGGL((k_blur<true, BLUR_ROWS>), dim3(nblk), dim3(BLUR_WAVES * 64), 0, st, b);
        else hipLaunchKernelGGL((k_blur<false, BLUR_ROWS>), dim3(nblk), dim3(BLUR_WAVES * 64), 0, st, b);
    }
    ORBX_LAUNCH_CHECK();
    return ORBX_OK;
}

}  // namespace orbx
