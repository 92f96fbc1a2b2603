// HIP kernels of the ORB extractor for gfx950 (MI355X, wave64).  One launch group processes a whole
// batch of frames: every kernel's grid spans frames x (levels x tiles | cells | keypoint slots), so
// launch cost is amortised over the batch and the 256 CUs always see >> 256 workgroups.
//
// Stage            reference (under /root/reference/src/ORBextractor.cc)        kernel
//   pyramid        ComputePyramid :781-822 (cv::resize INTER_LINEAR)             k_resize (per level, 7 launches) | k_pyramid (cones of levels, 2 launches; < 32 frames)
//   FAST + NMS     cv::FAST(cell, th, true) :607/:613 + raster-ordered cell lists  k_fast_cells (one workgroup per grid-cell row band)
//   quotas         :622-670                                                      k_quota      (one wave per level; a pass of the rule = one sweep of the lanes + two reductions)
//   retainBest     :683-685 (per cell), :697-701 (per level)                     k_cell_select (+ _long) / k_level_select (wave-parallel, permutation-exact introselect)
//   blur           GaussianBlur 7x7 s=2 :760                                     k_blur       (inside k_fast_blur for < 32 frames)
//   orientation    IC_Angle :124-151, descriptor :155-194, scaling :769-775      k_describe   (one wave per four keypoints)
// The one-frame drop-in call also has k_ingest (the staged frame fetched from pinned host memory by a kernel).
//
// No 16-px border planes exist on the device: the only out-of-image reads of the reference (blur
// taps <= 3 px, rotated BRIEF taps <= 2 px outside the ROI) are served by reflect-101 index math,
// which is what copyMakeBorder(BORDER_REFLECT_101) materialises.
#include <algorithm>
#include <type_traits>

#include "orb_math.h"
#include "orbx_internal.h"
#include "orbx_device.h"

namespace orbx {

__device__ __forceinline__ int fast_pair_test(const uint8_t* c, int S, int v, int t) {
    // ring offsets k=0..15 (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    const int x0 = c[3 * S], x1 = c[3 * S + 1], x2 = c[2 * S + 2], x3 = c[S + 3], x4 = c[3], x5 = c[-S + 3], x6 = c[-2 * S + 2], x7 = c[-3 * S + 1];
    const int x8 = c[-3 * S], x9 = c[-3 * S - 1], x10 = c[-2 * S - 2], x11 = c[-S - 3], x12 = c[-3], x13 = c[S - 3], x14 = c[2 * S - 2], x15 = c[3 * S - 1];
    // a dark 9-arc contains one pixel of every opposite pair, so max_k min(pair) < v - t is necessary (bright: mirrored)
    const int a = imax3(imax3(imin(x0, x8), imin(x1, x9), imin(x2, x10)), imax3(imin(x3, x11), imin(x4, x12), imin(x5, x13)), imax(imin(x6, x14), imin(x7, x15)));
    const int bq = imin3(imin3(imax(x0, x8), imax(x1, x9), imax(x2, x10)), imin3(imax(x3, x11), imax(x4, x12), imax(x5, x13)), imin(imax(x6, x14), imax(x7, x15)));
    return (int)(v - a > t) | (int)(bq - v > t);
}

__device__ __forceinline__ int fast_score_raw(const uint8_t* c, int S, int v, int tmin) {
    int x[16];
    x[0] = c[3 * S]; x[1] = c[3 * S + 1]; x[2] = c[2 * S + 2]; x[3] = c[S + 3]; x[4] = c[3]; x[5] = c[-S + 3]; x[6] = c[-2 * S + 2]; x[7] = c[-3 * S + 1];
    x[8] = c[-3 * S]; x[9] = c[-3 * S - 1]; x[10] = c[-2 * S - 2]; x[11] = c[-S - 3]; x[12] = c[-3]; x[13] = c[S - 3]; x[14] = c[2 * S - 2]; x[15] = c[3 * S - 1];
    int hi3[16], lo3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hi3[k] = imax3(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
        lo3[k] = imin3(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
    }
    int hi9[16], lo9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        hi9[k] = imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
        lo9[k] = imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    }
    // 16 -> 1 with three-input ops (8 instead of 16 two-input ones)
    const int min_hi9 = imin3(imin3(imin3(hi9[0], hi9[1], hi9[2]), imin3(hi9[3], hi9[4], hi9[5]), imin3(hi9[6], hi9[7], hi9[8])),
                              imin3(imin3(hi9[9], hi9[10], hi9[11]), imin3(hi9[12], hi9[13], hi9[14]), hi9[15]), 255);
    const int max_lo9 = imax3(imax3(imax3(lo9[0], lo9[1], lo9[2]), imax3(lo9[3], lo9[4], lo9[5]), imax3(lo9[6], lo9[7], lo9[8])),
                              imax3(imax3(lo9[9], lo9[10], lo9[11]), imax3(lo9[12], lo9[13], lo9[14]), lo9[15]), 0);
    const int s = imax(v - min_hi9, max_lo9 - v) - 1;   // == OpenCV cornerScore for every corner
    return s >= tmin ? s : 0;
}

// ------------------------------------------------------------------------------------ pyramid
// cv::resize INTER_LINEAR 8U, level-1 -> level.  A workgroup produces a 256 x RZ_ROWS output tile: the source rectangle it
// needs (<= ~60 rows x ~310 px at scale 1.2) is staged in LDS (LDS-DMA for aligned planes), each lane then produces 4 horizontally
// adjacent output pixels per row of its wave's RZ_ROWS / 4 consecutive rows and stores one dword per row.
// (Byte gathers straight from global memory made this kernel texture-addresser bound.)
// LDS source tile: L.rz_rows x L.rz_pitch bytes, the exact maximum over the level's tiles (a fixed worst-case array for scale
// 2.5 capped the kernel's occupancy)

template <bool ALIGNED, bool WINDOW>
__global__ __launch_bounds__(256) void k_resize(Batch b, int level) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_src[];
    const DevGeom& g = b.g;
    const LevelGeom& L = g.lv[level];
    const int RZ_SRC_W = L.rz_pitch;
    const LevelGeom& P = g.lv[level - 1];
    const int tiles_x = (L.w + 255) / 256, tiles_y = (L.h + RZ_ROWS - 1) / RZ_ROWS;
    int frame, tile;
    if (!frame_item(b, blockIdx.x, tiles_x * tiles_y, frame, tile)) return;
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int bx0 = tile_x * 256, by0 = tile_y * RZ_ROWS;
    const int bx1 = min(bx0 + 255, L.w - 1), by1 = min(by0 + RZ_ROWS - 1, L.h - 1);
    long long sstride;
    const uint8_t* src = plain_plane(b, P, level - 1, frame, sstride);
    const ResizeX* tx = b.tabx + L.tabx_off;
    const ResizeY* ty = b.taby + L.taby_off;
    // this lane's 4 output columns (independent of the staging below: issued first)
    const int dx0 = bx0 + lane * 4;
    ResizeX rx[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rx[k] = tx[min(dx0 + k, L.w - 1)];
    // the row-table entries of this wave's output rows, one per lane, fetched before the staging: read inside the row loop they
    // were a global-load round trip per output row (a uniform address, but not provably read-only, so no scalar load), and that
    // chain of dependent loads - not the VALU work - set the kernel's pace
    constexpr int RPW = RZ_ROWS / 4;                               // consecutive output rows per wave (windowed form)
    const uint2 ryl = *reinterpret_cast<const uint2*>(ty + min(by0 + wave * RPW + min(lane, RPW - 1), L.h - 1));
    // source rectangle of this tile (tables are monotone)
    const int r0 = ty[by0].sy0, r1 = ty[by1].sy1;
    const int c0 = tx[bx0].sx & ~3, c1 = tx[bx1].sx1;
    const int nd = ((c1 - c0) >> 2) + 1, nr = r1 - r0 + 1;
    if (ALIGNED) {
        // LDS-DMA staging (as in k_fast_cells): one global_load_lds_dword per (row, 64-dword piece), lane i's dword lands at
        // M0 + 4 i.  Row bases are scalars, so the staging costs a wave ~2 instructions per row instead of ~20 VALU instructions
        // per dword (flattened index -> row / column, bounds, address, ds_write): that loop was 45 % of this kernel's instructions.
        typedef const void __attribute__((address_space(1))) * gptr_t;
        typedef void __attribute__((address_space(3))) * lptr_t;
        const uint8_t* base = src + (long long)r0 * sstride + c0;
        for (int p0 = 0; p0 < nd; p0 += 64) {
            const bool on = p0 + lane < nd;
            for (int r = wave; r < nr; r += 4) {
                const uint8_t* grow = base + (long long)r * sstride + 4 * p0;     // wave-uniform
                if (on) __builtin_amdgcn_global_load_lds((gptr_t)(grow + 4 * lane), (lptr_t)(s_src + r * RZ_SRC_W + 4 * p0), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA writes have landed; the barrier below covers the other waves
    } else {
        // unaligned frames (level 0 -> 1 only): flattened (row, dword) items, 8 independent loads in flight per lane
        const int total = nr * nd;
        const float inv_nd = 1.0f / (float)nd;
        const uint8_t* base = src + (long long)r0 * sstride + c0;
        const int xm = P.w - 1 - c0;
        for (int i0 = 0; i0 < total; i0 += 256 * 8) {
            uint32_t v4[8];
            int off[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * 256 + tid;
                v4[k] = 0;
                off[k] = -1;
                if (i < total) {
                    int r, d;
                    split_px(i, nd, inv_nd, r, d);
                    const uint8_t* row = base + (long long)r * sstride;
                    off[k] = r * RZ_SRC_W + 4 * d;
                    if (ALIGNED) v4[k] = *reinterpret_cast<const uint32_t*>(row + 4 * d);
                    else v4[k] = (uint32_t)row[min(4 * d, xm)] | (uint32_t)row[min(4 * d + 1, xm)] << 8 | (uint32_t)row[min(4 * d + 2, xm)] << 16 |
                                 (uint32_t)row[min(4 * d + 3, xm)] << 24;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (off[k] >= 0) *reinterpret_cast<uint32_t*>(s_src + off[k]) = v4[k];
        }
    }
    __syncthreads();
    if (dx0 >= L.w) return;
    uint8_t* dplane = b.pyr + (long long)frame * g.frame_plane_bytes + L.plane_off;
    if (WINDOW) {
        // Windowed form (scale factors up to ~1.7): the lane's 8 source bytes of a row lie within 8 bytes from its first tap.  Three
        // aligned LDS dwords + two v_alignbyte bring that window into a register pair; a per-lane v_perm selector (row-invariant)
        // builds (s[sx] | s[sx1] << 16) of each output pixel and v_dot2_u32_u16 with (a0 | a1 << 16) is the horizontal pass
        // (D = S[sx]*a0 + S[sx+1]*a1).  A wave works on CONSECUTIVE output rows, so the lower source row of one output row is
        // usually the upper one of the next and its horizontal results are reused (1.2 instead of 2 source rows per output row).
        const int w0 = rx[0].sx - c0;                              // the lane's first tap inside the staged row
        const int wa = w0 & ~3, sh = w0 & 3;
        uint32_t sel[4], apair[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sel[k] = (uint32_t)(rx[k].sx - c0 - w0) | 0x0C000C00u | ((uint32_t)(rx[k].sx1 - c0 - w0) << 16);     // bytes: s[sx], 0, s[sx1], 0
            apair[k] = (uint32_t)(uint16_t)rx[k].a0 | ((uint32_t)(uint16_t)rx[k].a1 << 16);
        }
        // hrow leaves the horizontal results already shifted (D >> 4, the form the vertical pass consumes: once per source row, not
        // once per use).  Vertical pass: (b * (D >> 4)) >> 16 is v_mul_hi_u32 with the weight pre-shifted to the high half (a scalar
        // per row) - one instruction instead of multiply + shift; the result is < 256 by construction (weights sum to 2048), so the
        // four pixels are packed with shift-or, no masks.
        auto hrow = [&](int sy, uint32_t (&d)[4]) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(s_src + (sy - r0) * RZ_SRC_W + wa);
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)sh), hi = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)sh);
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = __builtin_amdgcn_udot2(as_us2v(__builtin_amdgcn_perm(hi, lo, sel[k])), as_us2v(apair[k]), 0u, false) >> 4;
        };
        int have = -1;
        uint32_t dprev[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int dy = by0 + wave * RPW + j;
            if (dy >= L.h) break;
            const uint32_t ry_rows = (uint32_t)__builtin_amdgcn_readlane((int)ryl.x, j), ry_w = (uint32_t)__builtin_amdgcn_readlane((int)ryl.y, j);
            const int sy0 = (int16_t)(ry_rows & 0xffffu), sy1 = (int16_t)(ry_rows >> 16);
            const uint32_t b0s = ry_w << 16, b1s = ry_w & 0xffff0000u;      // the weights (0 .. 2048) in the high halves
            uint32_t da[4], db[4];
            if (sy0 == have) {
#pragma unroll
                for (int k = 0; k < 4; k++) da[k] = dprev[k];
            } else hrow(sy0, da);
            if (sy1 == sy0) {
#pragma unroll
                for (int k = 0; k < 4; k++) db[k] = da[k];
            } else hrow(sy1, db);
            have = sy1;
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                dprev[k] = db[k];
                const uint32_t px = (__umulhi(da[k], b0s) + __umulhi(db[k], b1s) + 2u) >> 2;
                packed |= px << (8 * k);
            }
            // columns past L.w (dx0+k clamped above) land in the row padding: stride is a multiple of 64
            *reinterpret_cast<uint32_t*>(dplane + (long long)dy * L.stride + dx0) = packed;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < RZ_ROWS / 4; j++) {
        const int dy = by0 + wave + 4 * j;
        if (dy >= L.h) break;
        const ResizeY ry = ty[dy];
        const uint8_t* q0 = s_src + (ry.sy0 - r0) * RZ_SRC_W - c0;
        const uint8_t* q1 = s_src + (ry.sy1 - r0) * RZ_SRC_W - c0;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int px = resize_px(q0[rx[k].sx], q0[rx[k].sx1], q1[rx[k].sx], q1[rx[k].sx1], rx[k].a0, rx[k].a1, ry.b0, ry.b1);
            packed |= (uint32_t)(px & 255) << (8 * k);
        }
        // columns past L.w (dx0+k clamped above) land in the row padding: stride is a multiple of 64
        *reinterpret_cast<uint32_t*>(dplane + (long long)dy * L.stride + dx0) = packed;
    }
}

// Fused pyramid: one launch produces the levels l0+1 .. l0+depth of a PyrGroup (round 2: 2 launches for 8 levels instead of 7
// dependent ones — on one frame each k_resize launch cost ~9 us of latency, 65 of the ~150 us of a frame's kernel chain).
// A workgroup owns a tile of the deepest level and the cone above it.  The region of the source level it needs is staged in
// LDS; every further level is computed from the LDS copy of the level above (the same fixed-point cv::resize arithmetic and
// tables as k_resize), kept in LDS for the next one and written to HBM where the tile OWNS it (regions of neighbouring tiles
// overlap by the bilinear footprint; ownership — region start to the next tile's region start — partitions each level).
// Threads: 32 dword columns x 8 row phases; a thread keeps its four resize-table entries across its rows.
constexpr int PYR_FUSED_MAX_FRAMES = 32;
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_pyramid(Batch b, int group) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lv[];
    const DevGeom& g = b.g;
    const PyrGroup& pg = g.pyr[group];
    const int frame = blockIdx.z, ix = blockIdx.x, iy = blockIdx.y;
    const int tid = threadIdx.x;
    const int l0 = pg.l0, depth = pg.depth;
    if (group == 0 && ix == 0 && iy == 0 && tid == 0) b.status[frame] = ORBX_OK;     // (the selection stage may set an error later)
    const int* xt = b.pyr_tab + pg.xtab;
    const int* yt = b.pyr_tab + pg.ytab;
    auto xr = [&](int k, int i, int e) { return xt[(k * (pg.ntx + 1) + i) * 2 + e]; };
    auto yr = [&](int k, int i, int e) { return yt[(k * (pg.nty + 1) + i) * 2 + e]; };
    // stage the source region (level l0): rows ys..ye, dwords from xs (a multiple of 4)
    {
        const LevelGeom& P = g.lv[l0];
        const int xs = xr(0, ix, 0), xe = xr(0, ix, 1), ys = yr(0, iy, 0), ye = yr(0, iy, 1);
        const int nd = ((xe - xs) >> 2) + 1, nr = ye - ys + 1;
        long long sstride;
        const uint8_t* src = plain_plane(b, P, l0, frame, sstride);
        const uint8_t* base = src + (long long)ys * sstride + xs;
        const int xm = P.w - 1 - xs;
        const int total = nr * nd;
        const float inv_nd = 1.0f / (float)nd;
        const int pitch = pg.pitch[0];
        for (int i0 = 0; i0 < total; i0 += 256 * 8) {
            uint32_t v4[8];
            int off[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * 256 + tid;
                v4[k] = 0;
                off[k] = -1;
                if (i < total) {
                    int r, d;
                    split_px(i, nd, inv_nd, r, d);
                    const uint8_t* row = base + (long long)r * sstride;
                    off[k] = r * pitch + 4 * d;
                    if (ALIGNED) v4[k] = *reinterpret_cast<const uint32_t*>(row + 4 * d);
                    else v4[k] = (uint32_t)row[min(4 * d, xm)] | (uint32_t)row[min(4 * d + 1, xm)] << 8 | (uint32_t)row[min(4 * d + 2, xm)] << 16 |
                                 (uint32_t)row[min(4 * d + 3, xm)] << 24;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (off[k] >= 0) *reinterpret_cast<uint32_t*>(s_lv + pg.lds_off[0] + off[k]) = v4[k];
        }
    }
    __syncthreads();
    for (int k = 1; k <= depth; k++) {
        const int level = l0 + k;
        const LevelGeom& L = g.lv[level];
        const ResizeX* tx = b.tabx + L.tabx_off;
        const ResizeY* ty = b.taby + L.taby_off;
        const int xs = xr(k, ix, 0), xe = xr(k, ix, 1), ys = yr(k, iy, 0), ye = yr(k, iy, 1);
        const int own_x1 = xr(k, ix + 1, 0), own_y1 = yr(k, iy + 1, 0);          // owned: [xs, own_x1) x [ys, own_y1)
        const int sxs = xr(k - 1, ix, 0), sys = yr(k - 1, iy, 0);
        const uint8_t* sbuf = s_lv + pg.lds_off[k - 1];
        const int spitch = pg.pitch[k - 1];
        uint8_t* dbuf = k < depth ? s_lv + pg.lds_off[k] : nullptr;
        const int dpitch = k < depth ? pg.pitch[k] : 0;
        uint8_t* dplane = b.pyr + (long long)frame * g.frame_plane_bytes + L.plane_off;
        const int ngroups = ((xe - xs) >> 2) + 1;
        // threads: dword columns x row phases, the split chosen per level (the regions narrow towards the deepest level)
        const int lc = ngroups > 16 ? 5 : ngroups > 8 ? 4 : 3;
        const int gx = tid & ((1 << lc) - 1), gy = tid >> lc, ystep = 256 >> lc;
        for (int G0 = 0; G0 < ngroups; G0 += 1 << lc) {
            const int G = G0 + gx;
            if (G >= ngroups) continue;
            const int X = xs + 4 * G;
            ResizeX rx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) rx[j] = tx[min(X + j, L.w - 1)];
            for (int y = ys + gy; y <= ye; y += ystep) {
                const ResizeY ry = ty[y];
                const uint8_t* q0 = sbuf + (ry.sy0 - sys) * spitch - sxs;
                const uint8_t* q1 = sbuf + (ry.sy1 - sys) * spitch - sxs;
                uint32_t packed = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int px = resize_px(q0[rx[j].sx], q0[rx[j].sx1], q1[rx[j].sx], q1[rx[j].sx1], rx[j].a0, rx[j].a1, ry.b0, ry.b1);
                    packed |= (uint32_t)(px & 255) << (8 * j);
                }
                if (dbuf) *reinterpret_cast<uint32_t*>(dbuf + (y - ys) * dpitch + 4 * G) = packed;
                // columns past L.w (clamped above) land in the row padding: stride is a multiple of 64
                if (X < own_x1 && y < own_y1) *reinterpret_cast<uint32_t*>(dplane + (long long)y * L.stride + X) = packed;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------ FAST + NMS + cell lists
// One workgroup = one grid cell of one level of one frame (or one row band of a big cell) — the unit the reference
// calls cv::FAST on (src/ORBextractor.cc:599-614).  Because the NMS of cv::FAST never looks outside the cell view, a
// cell-native workgroup needs no score halo towards other cells, no survivor plane in HBM and no compaction pass.
// One score pass at min(fastTh, 7) serves the normal threshold and the reference's threshold-7 fallback (score >= t <=> corner at t).
// Phases: stage the band (+3 halo) in LDS -> A1 dense compass test -> A2 opposite-pair test (sparse) -> B exact score (sparse)
// -> N 3x3 strict NMS -> raster-ordered list with its counts at fastTh and at 7.  (Round 2 form; the round-1 kernel — flat pixel
// index per lane, block-wide queues with a barrier per phase — took 1.51 ms per 1024 VGA frames, this one 1.11.)
//   * everything is addressed by the pixel's BYTE OFFSET q inside the staged LDS image (pitch S): the score plane has the
//     image's layout, so ring / neighbour addresses are q +- dy*S + dx with no division anywhere in the dense or sparse phases;
//   * A1 is SWAR: a lane tests 4 horizontally adjacent pixels (one aligned dword) per step.  Bytes are unpacked to two
//     16-bit-field dwords (even / odd pixels); with the bias K = 0x8000 - t - 1 per field, "x < v - t" is bit 15 of (v + K) - x
//     and "x > v + t" is bit 15 of x + (K - v) — plain v_add / v_sub / v_and / v_or / v_bitop3, which issue at twice the rate of
//     v_min / v_max / v_cmp on gfx950 (profiles/r01_valu_issue_rates.txt, r02_valu_issue_rates2.txt).  Rule: a 9-arc of the
//     16-ring contains ring 0 or 8 AND ring 4 or 12, so a corner needs (N | S) & (E | W) beyond the threshold with one polarity —
//     an exact necessary condition.  (gfx950 serves unaligned ds_read_b32, but slowly: reading the E / W dwords that way instead
//     of two v_alignbyte cost +44 % on the kernel);
//   * flagged dwords are queued per WAVE (ballot + mbcnt, no atomics), expanded to pixels, pair-tested and scored by the same
//     wave in full-wave slices: no workgroup barrier between staging and the NMS;
//   * every wave remembers the pixels it gave a score; the NMS visits those (a dense sweep over the score plane only where a
//     wave's list overflowed: noise-like bands), survivors set bits in a q-space bitmask, the raster-ordered list comes from
//     the same chunk scan as before.
// Round 3 (1.12 -> 0.99 ms per 1024 VGA frames).  Cut short phase by phase the kernel costs staging 0.33 + dense phase 0.34 + drain
// 0.28 + NMS 0.06 + list 0.07 ms, and with its dynamic LDS padded 1.06 / 1.15 / 1.30 / 1.56 ms at 6 / 5 / 4 / 3 workgroups per CU: it is
// bound by latency at the occupancy its LDS allows, not by instruction issue alone.  Hence: the queues are as small as their
// invariants allow (7 workgroups per CU on VGA grids), the staging moves 16 bytes per lane, the drain scores both remainders in
// one pass without the pair test, the list output scans with DPP adds instead of ds_bpermute and counts by ballot.  (NOTES.md 8.2b;
// what did NOT help: pooling the waves' remainders behind extra barriers, an L2 prefetch of a later band, smaller bands.)
struct FastHdr { int n_hi, n_lo, overflow, pad1; int wsum[8]; int pad2[4]; };
static_assert(sizeof(FastHdr) == 64, "LDS carve");

__device__ __forceinline__ int lane_rank(unsigned long long m) {   // number of set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
template <bool ALIGNED, int NT, int PPT>
__device__ __forceinline__ void fast_band_task(const Batch& b, int frame, int item, uint8_t* smem) {
    constexpr int NW = NT / 64;
    static_assert((NW & (NW - 1)) == 0, "wave roles rotate modulo NW");
    constexpr int Q0CAP = fast_q0cap(PPT), Q1CAP = FAST_Q1CAP, Q2CAP = FAST_Q2CAP, Q3CAP = FAST_Q3CAP;
    constexpr int WQ_BYTES = fast_wave_queue_bytes(PPT);
    const DevGeom& g = b.g;
    const BandGeom bg = b.bands[item];
    const int level = bg.level;
    const LevelGeom& L = g.lv[level];
    // Wave roles rotate with the band index.  The tails of the task are wave-0 heavy (a VGA cell's survivor list is <= 70 chunks: wave 0
    // writes all of it; tid 0 closes the band), and the waves of a 4-wave workgroup land on the CU's four SIMDs in order, so without
    // the rotation one SIMD of every CU carried all of that.  Every use of wave / tid below is work assignment only.
    const int lane = threadIdx.x & 63, wave = ORBX_FAST_ROTATE ? (wave_id() + item) & (NW - 1) : wave_id(), tid = wave * 64 + lane;
    const int cw = bg.x1 - bg.x0 + 1, ch = bg.ey1 - bg.ey0 + 1;      // scored rectangle: own rows + halo rows towards sibling bands
    CellState* cst = b.cstate + (long long)frame * g.nbands_total + item;
    // Fallback hint (round 6: per FRAME SLOT): how many launch groups in a row THIS band of THIS slot of the launch group ended with <= 3
    // survivors@fastTh.  It lives in the upper bits of CellState::thr, which the same work item of the previous launch group left behind: one
    // load in front of the band, no store of its own (round 5 kept one table per handle, written by frame 0 and read by every frame: a launch
    // group that mixes streams — several cameras, a handle reused across sequences — inherited frame 0's texture class).
    const int hint = b.fallback_hint ? (int)((uint32_t)cst->thr >> 8) : 0;
    if (cw <= 0 || ch <= 0) {
        if (tid == 0) { CellState st; st.n_all = 0; st.n_hi = 0; st.n_lo = 0; st.thr = g.tmin; *cst = st; }
        return;
    }
    const int own_lo = bg.y0 - bg.ey0, own_hi = bg.y1 - bg.ey0;
    // staged image: rows ey0-3 .. ey1+3, columns from the dword-aligned start at or left of x0-3; pixel (x, y) of the band
    // (relative to (x0, ey0)) lives at byte offset q = (y + 3) * S + x + xoff + 3
    const int gxb = (bg.x0 - 3) & ~3;
    const int xoff = (bg.x0 - 3) - gxb;
    const int nd = fast_row_dwords(xoff, cw);
    const int S = nd * 4;
    const int x_first = xoff + 3;
    const int nrows = ch + 6;
    // LDS carve: header | survivor bit masks | per-wave queues | image | scores (image layout)
    FastHdr* hdr = reinterpret_cast<FastHdr*>(smem);
    unsigned long long* cmask = reinterpret_cast<unsigned long long*>(smem + sizeof(FastHdr));
    uint8_t* wq = smem + sizeof(FastHdr) + g.fast_max_chunks * 8 + wave * WQ_BYTES;
    uint32_t* q0 = reinterpret_cast<uint32_t*>(wq);
    uint16_t* q1 = reinterpret_cast<uint16_t*>(wq + Q0CAP * 4);
    uint16_t* q2 = q1 + Q1CAP;
    uint16_t* q3 = q2 + Q2CAP;
    uint8_t* s_img = smem + sizeof(FastHdr) + g.fast_max_chunks * 8 + NW * WQ_BYTES;
    uint8_t* s_sc = s_img + g.fast_max_img;
    const int q_own_lo = (3 + own_lo) * S, q_own_hi = (3 + own_hi + 1) * S;      // byte offsets of the band's own rows
    const int nchunks = (q_own_hi - q_own_lo + 63) >> 6;
    long long stride64;
    const uint8_t* src = plain_plane(b, L, level, frame, stride64);
    auto clear_lds = [&]() {
        for (int i = tid; i < ((nrows * S + 15) >> 4); i += NT) reinterpret_cast<uint4*>(s_sc)[i] = make_uint4(0, 0, 0, 0);
        if (tid < (int)(sizeof(FastHdr) / 4)) reinterpret_cast<int*>(hdr)[tid] = 0;
        for (int i = tid; i < nchunks; i += NT) cmask[i] = 0ull;
    };
    if (ALIGNED) {
        // LDS-DMA staging (global_load_lds_dwordx4: lane i's 16 bytes land at M0 + 16 i).  No VGPRs for the data, no ds_write; a wave
        // issues its few instructions back to back and waits once.  (Round 2 moved a dword per lane, one instruction per row piece:
        // 17 instructions per wave for a VGA band against 3 now, 1.00 -> 0.97 ms per 1024 frames.)
        typedef const void __attribute__((address_space(1))) * gptr_t;
        typedef void __attribute__((address_space(3))) * lptr_t;
        const uint8_t* src0 = src + (long long)(bg.ey0 - 3) * stride64 + gxb;
        clear_lds();      // first: the compiler orders every LDS write behind outstanding LDS-DMA loads (vmcnt(0)), so behind them it would wait for the band
        // The band's rows are nd / 4 chunks of 16 bytes each, the chunks of all rows one flat list q (LDS offset 16 q, row q / cpr,
        // chunk q % cpr); a wave instruction moves 64 consecutive chunks, i.e. several whole rows of a VGA-class band.  Global
        // addresses are dword-aligned only (the band starts at the dword at or left of x0 - 3), which the x4 load accepts.
        {
            const int cpr = nd >> 2, nchunks16 = nrows * cpr;
            const float inv_cpr = bg.inv_cpr;
            for (int q0 = 64 * wave; q0 < nchunks16; q0 += 64 * NW) {
                const int q = q0 + lane;
                int r, c;
                split_px(imin(q, nchunks16 - 1), cpr, inv_cpr, r, c);
                const uint8_t* ga = src0 + (long long)r * stride64 + 16 * c;
                if (q < nchunks16) __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)(s_img + 16 * q0), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA writes of THIS wave have landed; the barrier below covers the others
    } else {
        // unaligned frames (level 0 only): flattened (row, dword) items assembled from byte loads, 8 in flight per lane
        const int total = nrows * nd;
        const float inv_nd0 = bg.inv_nd;
        const uint8_t* src0 = src + (long long)(bg.ey0 - 3) * stride64 + gxb;
        const int xm = L.w - 1 - gxb;   // never read past the row end
        for (int i0 = 0; i0 < total; i0 += NT * 8) {
            uint32_t v4[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * NT + tid;
                v4[k] = 0;
                if (i < total) {
                    int r, d;
                    split_px(i, nd, inv_nd0, r, d);
                    const uint8_t* row = src0 + (long long)r * stride64;
                    v4[k] = (uint32_t)row[imin(4 * d, xm)] | (uint32_t)row[imin(4 * d + 1, xm)] << 8 | (uint32_t)row[imin(4 * d + 2, xm)] << 16 |
                            (uint32_t)row[imin(4 * d + 3, xm)] << 24;
                }
            }
            if (i0 == 0) clear_lds();
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * NT + tid;
                if (i < total) reinterpret_cast<uint32_t*>(s_img)[i] = v4[k];
            }
        }
    }
    __syncthreads();

    // Threshold of the current pass (round 4).  The band is first scored at fastTh: survivors@fastTh = survivors@7 intersected with
    // {score >= fastTh} (a neighbour scoring below fastTh can never block a pixel scoring at least fastTh), so a band that keeps
    // more than 3 of them proves that its cell has more than 3 and never takes the reference's threshold-7 fallback (:609-614): its
    // list at fastTh is all the later stages read.  Only a band with <= 3 survivors@fastTh is scored again at 7 (below).  On textured
    // input (corners@7 several times corners@fastTh) the pair test, score, NMS and list phases shrink by that factor; on the S-blocks
    // stream 3 % of the bands take the second pass.  fastTh <= 7: one pass at fastTh serves both (g.tmin = fastTh).
    // Fallback hint (round 5; per frame slot since round 6, see above): a band that needed the second pass in FAST_HINT_RUN launch groups in a row starts at 7
    // (low-texture streams: every band would otherwise run twice, S-lowtex +35 % on this kernel).  A wrong hint costs one pass at 7
    // instead of one at fastTh, never a wrong result: CellState::thr tells the later stages what the list was made at.
    constexpr int FAST_HINT_RUN = 6;
    int tmin = ORBX_FAST_TWO_PASS ? (hint >= FAST_HINT_RUN && g.fast_th > 7 ? 7 : g.fast_th) : g.tmin;
    const float inv_nd = bg.inv_nd;                // (1 / nd, 1 / S, 1 / cpr come with the band: BandGeom)
    int n0 = 0, n1 = 0, n2 = 0, n3 = 0;          // fill of this wave's queues (wave-uniform)

    // B: exact FAST score of <= 64 queued pixels (`on` lanes hold one each); scored corners are remembered in q3
    auto score_vals = [&](bool on, int p) {
        int sc = 0;
        if (on) {
            const uint8_t* c = s_img + p;
            sc = fast_score_raw(c, S, c[0], tmin);
            s_sc[p] = (uint8_t)sc;
        }
        const unsigned long long mk = __ballot(sc != 0);
        if (mk) {
            const int add = __popcll(mk);
            if (n3 + add <= Q3CAP) { if (sc) q3[n3 + lane_rank(mk)] = (uint16_t)p; }
            n3 += add;                              // beyond Q3CAP: the band takes the dense NMS sweep
        }
    };
    auto score_step = [&](const uint16_t* q, int m) { score_vals(lane < m, lane < m ? (int)q[lane] : 0); };
    // A2: OpenCV's opposite-pair pre-test of m <= 64 queued pixels; survivors go to q2, which is scored whenever it holds a full wave
    auto pair_step = [&](const uint16_t* q, int m) {
        int pass = 0, p = 0;
        if (lane < m) {
            p = q[lane];
            const uint8_t* c = s_img + p;
            pass = fast_pair_test(c, S, c[0], tmin);
        }
        const unsigned long long mk = __ballot(pass);
        if (mk) {
            if (pass) q2[n2 + lane_rank(mk)] = (uint16_t)p;
            n2 += __popcll(mk);
            if (n2 >= 64) { n2 -= 64; score_step(q2 + n2, 64); }
        }
    };
    // expansion of m <= 64 flagged dwords into pixel offsets (pixels of the alignment / halo columns are dropped here)
    auto expand_step = [&](const uint32_t* q, int m) {
        uint32_t e = 0;
        if (lane < m) e = q[lane];
        const int idx = (int)(e & 0x3FFFu);
        int r, d;
        split_px(idx, nd, inv_nd, r, d);
        const int col0 = 4 * d - x_first;                                   // band column of the dword's first pixel
        // flag bits: pixel 0 -> bit 15, 1 -> bit 14, 2 -> bit 31, 3 -> bit 30
        const int f[4] = {(int)((e >> 15) & 1u) & (int)((unsigned)col0 < (unsigned)cw), (int)((e >> 14) & 1u) & (int)((unsigned)(col0 + 1) < (unsigned)cw),
                          (int)((e >> 31) & 1u) & (int)((unsigned)(col0 + 2) < (unsigned)cw), (int)((e >> 30) & 1u) & (int)((unsigned)(col0 + 3) < (unsigned)cw)};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long mk = __ballot(f[j]);
            if (mk) {
                if (f[j]) q1[n1 + lane_rank(mk)] = (uint16_t)(4 * idx + j);
                n1 += __popcll(mk);
            }
            if (j & 1) while (n1 >= 64) { n1 -= 64; pair_step(q1 + n1, 64); }      // after two pixels per dword: q1 never holds more than 63 + 128
        }
    };

    // A1: SWAR compass test, 4 pixels per lane and step (see the header of this section)
    const uint32_t M8 = 0x00FF00FFu, HH = 0x80008000u;
    uint32_t KD = (uint32_t)(0x8000 - tmin - 1) * 0x00010001u;
    auto compass4 = [&](uint32_t C, uint32_t E, uint32_t W, uint32_t Nn, uint32_t Ss) -> uint32_t {
        uint32_t P[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const uint32_t c = hlf ? (C >> 8) & M8 : C & M8, n = hlf ? (Nn >> 8) & M8 : Nn & M8, s = hlf ? (Ss >> 8) & M8 : Ss & M8,
                           e = hlf ? (E >> 8) & M8 : E & M8, w = hlf ? (W >> 8) & M8 : W & M8;
            const uint32_t vd = c + KD, vb = KD - c;
            const uint32_t dk = ((vd - n) | (vd - s)) & ((vd - e) | (vd - w));
            const uint32_t br = ((n + vb) | (s + vb)) & ((e + vb) | (w + vb));
            P[hlf] = dk | br;
        }
        return (P[0] & HH) | ((P[1] & HH) >> 1);
    };
    const int i_begin = 3 * nd, i_end = (ch + 3) * nd;          // dwords of the scored rows (all columns of the staged image)
    constexpr int RG = NW * 64 * PPT;                            // dwords per round of the workgroup
    auto round = [&](auto full_c, int base) {
        constexpr bool FULL = decltype(full_c)::value;          // every lane's PPT dwords lie below i_end: constant LDS offsets, no masking
        const int i0 = base + wave * 64 + lane;
        uint32_t C[PPT], E[PPT], W[PPT], Nn[PPT], Ss[PPT];
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int ik = FULL ? i0 + k * NW * 64 : imin(i0 + k * NW * 64, i_end - 1);
            const uint8_t* pk = s_img + 4 * ik;
            C[k] = *reinterpret_cast<const uint32_t*>(pk);
            E[k] = __builtin_amdgcn_alignbyte(*reinterpret_cast<const uint32_t*>(pk + 4), C[k], 3);   // pixels x+3 .. x+6
            W[k] = __builtin_amdgcn_alignbyte(C[k], *reinterpret_cast<const uint32_t*>(pk - 4), 1);   // pixels x-3 .. x
            Nn[k] = *reinterpret_cast<const uint32_t*>(pk - 3 * S);
            Ss[k] = *reinterpret_cast<const uint32_t*>(pk + 3 * S);
        }
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int i = i0 + k * NW * 64;
            uint32_t Q = compass4(C[k], E[k], W[k], Nn[k], Ss[k]);
            if (!FULL && i >= i_end) Q = 0;
            const unsigned long long mk = __ballot(Q != 0);
            if (mk) {
                if (Q) q0[n0 + lane_rank(mk)] = Q | (uint32_t)i;
                n0 += __popcll(mk);
                if (n0 >= 64) { n0 -= 64; expand_step(q0 + n0, 64); }          // q0 never holds more than 63 + 64
            }
        }
    };
    for (;;) {      // one pass at fastTh; a second one at 7 for a band with <= 3 survivors@fastTh
    {
        int base = i_begin;
        for (; base + RG <= i_end; base += RG) round(std::true_type{}, base);
        if (base < i_end) round(std::false_type{}, base);
    }
    // Drain this wave's queues.  The remainders (< 64 each) run with a fraction of the lanes whatever is done, and this tail is a chain
    // of dependent LDS round trips (measured by cutting the kernel short: the drain costs 0.28 of the 1.08 ms per 1024 VGA frames,
    // the whole dense phase 0.34).  So the pair test is skipped here: it is only a filter (a pixel that fails it scores below tmin,
    // i.e. 0) and costs a queue round trip plus the same 16 ring reads the score needs; the pixel remainder and the pair-tested
    // remainder are scored together in one pass (two when they exceed a wave).
    if (n0) expand_step(q0, n0);
    for (int base = 0; base < n1 + n2; base += 64) {
        const int i = base + lane;
        const bool on = i < n1 + n2;
        score_vals(on, on ? (int)(i < n1 ? q1[i] : q2[i - n1]) : 0);
    }
    if (n3 > Q3CAP && lane == 0) hdr->overflow = 1;
    __syncthreads();

    // N: 3x3 strict NMS of the scored pixels of the band's own rows.  Scores of the halo columns / rows and of every non-corner
    // are 0, which is what cv::FAST's NMS sees outside the cell view.
    auto nms_test = [&](int p, int s) -> bool {       // strict maximum of its 3 x 3 neighbourhood: the survivor's bit is set
        const uint8_t* sp = s_sc + p;
        const int mx = imax3(imax3(sp[-1], sp[1], sp[-S]), imax3(sp[S], sp[-S - 1], sp[-S + 1]), imax(sp[S - 1], sp[S + 1]));
        if (s > mx) {
            const int bit = p - q_own_lo;
            atomicOr(&cmask[bit >> 6], 1ull << (bit & 63));
        }
        return s > mx;
    };
    auto nms_px = [&](int p, int s) {                 // ... counted per lane (the dense sweep's lanes are out of step)
        if (nms_test(p, s)) {
            if (s >= g.fast_th) atomicAdd(&hdr->n_hi, 1);
            if (s >= 7) atomicAdd(&hdr->n_lo, 1);
        }
    };
    if (!hdr->overflow) {
        int c_hi = 0, c_lo = 0;                       // counted per wave: two ballots per pass instead of two LDS atomics per survivor
        for (int i0 = 0; i0 < n3; i0 += 64) {
            const int i = i0 + lane;
            int s = 0;
            bool keep = false;
            if (i < n3) {
                const int p = q3[i];
                s = s_sc[p];
                keep = p >= q_own_lo && p < q_own_hi && nms_test(p, s);
            }
            c_hi += __popcll(__ballot(keep && s >= g.fast_th));
            c_lo += __popcll(__ballot(keep && s >= 7));
        }
        if (lane == 0 && (c_hi | c_lo)) { atomicAdd(&hdr->n_hi, c_hi); atomicAdd(&hdr->n_lo, c_lo); }      // (fastTh < 7: scores of 5 and 6 count in n_hi only)
    } else {
        const int d_lo = q_own_lo >> 2, d_hi = q_own_hi >> 2;
        for (int i = d_lo + tid; i < d_hi; i += NT) {
            uint32_t sc4 = reinterpret_cast<const uint32_t*>(s_sc)[i];
            while (sc4) {
                const int j = (__ffs((int)sc4) - 1) >> 3;
                const int s = (int)((sc4 >> (8 * j)) & 255u);
                sc4 &= ~(255u << (8 * j));
                nms_px(4 * i + j, s);
            }
        }
    }
    __syncthreads();
    if (tmin <= 7 || __builtin_amdgcn_readfirstlane(hdr->n_hi) > 3) break;          // workgroup-uniform
    // second pass at the fallback threshold: score plane, survivor masks and counts start over (the staged image stays)
    __syncthreads();                                 // every wave has read n_hi
    clear_lds();
    tmin = 7;
    KD = (uint32_t)(0x8000 - 7 - 1) * 0x00010001u;
    n0 = n1 = n2 = n3 = 0;
    __syncthreads();
    }
    // the band's keypoint list in raster order (cv::FAST's order): one lane per 64-byte chunk of the survivor bitmask
    Cand* out = b.cand + (long long)frame * g.frame_cands + L.cand_base + bg.cand_off;
    const float inv_S = bg.inv_s;
    int run_base = 0;
    for (int c0 = 0; c0 < nchunks; c0 += NT) {
        // a wave whose 64 chunks lie beyond the band (waves 2 and 3 of a VGA level-0 band: 81 chunks) only keeps the barriers company:
        // it reports a count of 0 and skips the scan, the count exchange and the output loop (wave 0 is never idle: it carries run_base)
        const bool busy = c0 + 64 * wave < nchunks;
        unsigned long long m = 0ull;
        int cnt = 0, incl = 0;
        if (busy) {
            if (c0 + tid < nchunks) m = cmask[c0 + tid];
            cnt = __popcll(m);
            incl = wave_scan_inclusive(cnt);
        }
        if (NW > 1) {
            if (lane == 63) hdr->wsum[wave] = incl;
            __syncthreads();
        }
        int run = run_base + incl - cnt, total = NW > 1 ? 0 : __builtin_amdgcn_readlane(incl, 63);
        if (NW > 1 && busy) {
#pragma unroll
            for (int wv = 0; wv < NW; wv++) { const int t = hdr->wsum[wv]; if (wv < wave) run += t; total += t; }
        }
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int p = q_own_lo + (c0 + tid) * 64 + bit;
            int r, xr;
            split_px(p, S, inv_S, r, xr);
            Cand e;
            e.pos = (uint32_t)(bg.x0 + xr - x_first) | ((uint32_t)(bg.ey0 + r - 3) << 16);
            e.resp = (float)s_sc[p];
            out[run++] = e;
        }
        run_base += total;
        if (NW > 1 && c0 + NT < nchunks) __syncthreads();
    }
    if (tid == 0) {
        CellState st;
        st.n_all = run_base; st.n_hi = hdr->n_hi; st.n_lo = hdr->n_lo;
        st.thr = tmin | ((st.n_hi <= 3 && g.fast_th > 7 ? imin(hint + 1, FAST_HINT_RUN) : 0) << 8);      // list threshold | the slot's run of fallbacks
        *cst = st;
    }
}

// (round 5: workgroups walking 2 / 4 / 16 bands grid-stride — one launch of long-lived workgroups instead of 729 k short ones — took
//  1.02 / 1.02 / 1.05 ms against 0.845 per 1024 VGA frames: the dispatcher's interleaving of fresh workgroups is what hides a band's
//  serial phases, a resident workgroup exposes them)
template <bool ALIGNED, int NT, int PPT>
__global__ __launch_bounds__(NT) void k_fast_cells(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int frame, item;
    if (!frame_item_magic(b, blockIdx.x, (unsigned)b.g.nbands_total, b.g.nbands_magic, frame, item)) return;
    fast_band_task<ALIGNED, NT, PPT>(b, frame, item, smem);
}

// ------------------------------------------------------------------------------------ quotas
// reference :609-670.  One wave per (frame, level).  The redistribution loop of the reference looks sequential, but within one of
// its passes the new per-cell allowance is fixed before the pass starts and every cell decides on its own; only the two totals
// (features left to distribute, cells that cannot take more) couple the cells, and they are sums.  So a pass is one sweep of the
// lanes over their cells plus two wave reductions, and the output offsets are a wave prefix sum in cell order.  Lane i owns the
// cells i, i + 64, ...; nothing a lane writes is read by another lane, so the LDS arrays need no barriers.  (A single lane walking
// the cells one by one took 10 us per level: nothing for a full batch, 7 % of the one-frame call.)
// LDS: four per-cell arrays sized by the level with the most cells (DevGeom::quota_cells, a multiple of 64; the host bounds it by
// QUOTA_MAX_CELLS = what 160 KiB hold)
struct QuotaLds { int *nkeys, *nret; uint8_t *thr, *done; };

__global__ __launch_bounds__(64) void k_quota(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DevGeom& g = b.g;
    QuotaLds q;
    q.nkeys = reinterpret_cast<int*>(smem);
    q.nret = q.nkeys + g.quota_cells;
    q.thr = reinterpret_cast<uint8_t*>(q.nret + g.quota_cells);
    q.done = q.thr + g.quota_cells;
    const int frame = blockIdx.x / g.nlevels, level = blockIdx.x - frame * g.nlevels;
    const int lane = (int)threadIdx.x;
    const LevelGeom& L = g.lv[level];
    const CellGeom* cg = b.cells + L.cell_base;
    const CellState* cs = b.cstate + (long long)frame * g.nbands_total;   // per band; a cell sums its bands
    CellSel* sel = b.csel + (long long)frame * g.ncells_total + L.cell_base;
    const int nCells = L.ncells, nfc = L.nfeat_cell;
    // first pass of the reference (:622-641), fused with the gather of the cells' counts
    int dist = 0, nomore = 0;
    for (int c = lane; c < nCells; c += 64) {
        const CellGeom cgc = cg[c];
        int n_hi = 0, n_lo = 0;
        for (int k = 0; k < cgc.nbands; k++) { const CellState t = cs[cgc.band0 + k]; n_hi += t.n_hi; n_lo += t.n_lo; }
        const bool fallback = n_hi <= 3;                      // :609  size()<=3 -> FAST(...,7,...)
        const int nk = cgc.skipped ? 0 : (fallback ? n_lo : n_hi);
        q.thr[c] = (uint8_t)(cgc.skipped || !fallback ? g.fast_th : 7);
        q.nkeys[c] = nk;
        // a skipped cell takes the reference's `continue`: it never reaches the bookkeeping of this pass (and is then treated as an
        // open cell with no keypoints by the passes below, exactly like there)
        if (cgc.skipped) { q.nret[c] = 0; q.done[c] = 0; }
        else if (nk > nfc) { q.nret[c] = nfc; q.done[c] = 0; }
        else { q.nret[c] = nk; dist += nfc - nk; q.done[c] = 1; nomore++; }
    }
    int nToDistribute = wave_sum(dist), nNoMore = wave_sum(nomore);
    while (nToDistribute > 0 && nNoMore < nCells) {           // :645-668
        const int nNew = nfc + (int)ceilf((float)nToDistribute / (float)(nCells - nNoMore));
        dist = 0; nomore = 0;
        for (int c = lane; c < nCells; c += 64) {
            if (q.done[c]) continue;
            const int nk = q.nkeys[c];
            if (nk > nNew) q.nret[c] = nNew;
            else { q.nret[c] = nk; dist += nNew - nk; q.done[c] = 1; nomore++; }
        }
        nToDistribute = wave_sum(dist);
        nNoMore += wave_sum(nomore);
    }
    // output offsets in cell order; the level's total decides whether the lists fit
    int total = 0;
    for (int c0 = 0; c0 < nCells; c0 += 64) total += (c0 + lane < nCells) ? q.nret[c0 + lane] : 0;
    total = wave_sum(total);
    const bool bad = total > L.sel_cap;
    int base = 0;
    for (int c0 = 0; c0 < nCells; c0 += 64) {
        const int c = c0 + lane;
        const int v = c < nCells ? q.nret[c] : 0;
        const int incl = wave_scan_inclusive(v);
        if (c < nCells) {
            CellSel r;
            r.thr = q.thr[c]; r.nkeys = q.nkeys[c];
            r.nretain = bad ? 0 : v;
            r.out_off = bad ? 0 : base + incl - v;
            sel[c] = r;
        }
        base += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) {
        if (bad) b.status[frame] = ORBX_ERR_CAPACITY;
        b.level_total[frame * MAX_LEVELS + level] = bad ? 0 : total;
    }
}

// ------------------------------------------------------------------------------------ retainBest per cell
// KeyPointsFilter::retainBest(keysCell, n) followed by resize(n) keeps exactly the first n elements
// that std::nth_element leaves in front (the std::partition of boundary ties is truncated away again by
// the resize).  Which tied keypoints survive, and their ORDER, is libstdc++'s introselect.
//
// wave_nth_element reproduces libstdc++'s std::nth_element(first, nth, last, greater-by-response) — the exact
// permutation, not just the set — with the wave working in parallel on the Hoare partition passes:
//   __introselect:   while (last-first > 3) { depth check; cut = __unguarded_partition_pivot; narrow } + insertion sort
//   pivot:           __move_median_to_first(first, first+1, mid, last-1)            (lane 0, 3 compares)
//   partition:       i scans right over elements > pivot, j scans left over elements < pivot, swap, repeat.
// Within one pass the scans only ever stop at "left stoppers" (value <= pivot) resp. "right stoppers" (value >= pivot)
// of the ORIGINAL array — elements between the pointers are untouched — so swap k exchanges the k-th left stopper
// L[k] with the k-th right stopper from the top R[k] while L[k] < R[k]; after S swaps the left scan stops at
// min(L[S], R[S-1]) (R[S-1] now holds a value <= pivot), which is the returned cut.  L and R are built with ordered
// __ballot compaction, the swaps are disjoint and run in parallel.  The depth-limit fallback (heap select) and the
// final <= 3-element insertion sort call libstdc++'s own constexpr internals on lane 0.
// The list `a` and the scratch `lpos`/`rpos` (n uint16 each) live in LDS.
struct RespGreater {   // KeypointResponseGreater (OpenCV keypoint.cpp)
    __host__ __device__ constexpr bool operator()(const Cand& x, const Cand& y) const { return x.resp > y.resp; }
};


__device__ __forceinline__ int mask_rank(unsigned long long m) {     // number of set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// Ranges of at most 64 entries are finished IN REGISTERS (round 5): lane i holds entry first + i, and a partition pass is a few ballots
// and two hops through the LDS crossbar instead of ~25 dependent LDS round trips (most passes of any list, and every pass of the 20- to
// 40-entry lists of an ordinary cell, are over such ranges; a wave's selection is a chain of latencies, ~1 us per LDS pass).
//   pivot      __move_median_to_first reads three entries (v_readlane at wave-uniform lanes) and swaps two lanes;
//   stoppers   the ballots mL (entry <= pivot) and mR (entry >= pivot) over the lanes (f, l).  Swap k of the Hoare partition exchanges the
//              k-th left stopper from below with the k-th right stopper from above while the former lies below the latter: a left
//              stopper with kL left stoppers below it and aR right stoppers above it is swapped iff aR > kL, a right stopper (rank kR
//              = aR from the top) iff more than kR left stoppers lie below it — no lists, two mbcnt per lane; S = popcount of either;
//   swaps      left swappers push (pos, resp, lane) to lane kL, right ones to lane 32 + kR (ds_permute; S <= 31), partners look at each
//              other's origin (ds_bpermute lane ^ 32) and push the entries on to it;
//   cut        position of the left stopper of rank S / the right stopper of rank S - 1 (ballot + ffs), as in the LDS form;
//   <= 3 left  __insertion_sort of at most three entries = their stable descending order: ranks from three readlanes.
// The depth-limit fallback writes the window back and calls libstdc++'s heap select on lane 0 like the LDS form.
__device__ __forceinline__ void wave_nth_small(Cand* a, int first, int nth, int last, int depth, int lane) {
    const int n = last - first;                        // 4 .. 64
    uint32_t pos = 0;
    float resp = 0.f;
    if (lane < n) { const Cand e = a[first + lane]; pos = e.pos; resp = e.resp; }
    int f = 0, l = n;
    const int k = nth - first;
    bool heap = false;
    while (l - f > 3) {
        if (depth == 0) { heap = true; break; }
        --depth;
        const int mid = f + (l - f) / 2;
        const float ra = readlane_f(resp, f + 1), rb = readlane_f(resp, mid), rc = readlane_f(resp, l - 1);
        // std::__move_median_to_first(result = f, a = f + 1, b = mid, c = l - 1) with comp = greater
        int sl;
        if (ra > rb) sl = rb > rc ? mid : (ra > rc ? l - 1 : f + 1);
        else sl = ra > rc ? f + 1 : (rb > rc ? l - 1 : mid);
        {
            const uint32_t pf = (uint32_t)__builtin_amdgcn_readlane((int)pos, f), ps = (uint32_t)__builtin_amdgcn_readlane((int)pos, sl);
            const float rf = readlane_f(resp, f), rs = readlane_f(resp, sl);
            if (lane == f) { pos = ps; resp = rs; }
            if (lane == sl) { pos = pf; resp = rf; }
        }
        const float P = readlane_f(resp, f);
        const bool inr = lane > f && lane < l;
        const bool stL = inr && !(resp > P), stR = inr && !(P > resp);
        const unsigned long long mL = __ballot(stL), mR = __ballot(stR);
        const int nL = __popcll(mL), nR = __popcll(mR);
        const int kL = mask_rank(mL);                                  // left stoppers below this lane
        const int aR = nR - mask_rank(mR) - (stR ? 1 : 0);             // right stoppers above this lane
        const bool doL = stL && aR > kL, doR = stR && kL > aR;
        const int S = __popcll(__ballot(doL));
        int cut;
        {
            const unsigned long long cl = __ballot(stL && kL == S), cr = __ballot(stR && aR == S - 1);
            if (S < nL) { cut = __ffsll((long long)cl) - 1; if (S > 0) { const int r = __ffsll((long long)cr) - 1; if (r < cut) cut = r; } }
            else cut = __ffsll((long long)cr) - 1;
        }
        if (S > 0) {
            const int d1 = 4 * (doL ? kL : (doR ? 32 + aR : 63));      // (lane 63 is no rank lane: S <= 31)
            const uint32_t h_pos = (uint32_t)__builtin_amdgcn_ds_permute(d1, (int)pos);
            const float h_resp = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(d1, __builtin_bit_cast(int, resp)));
            const int h_src = __builtin_amdgcn_ds_permute(d1, lane);
            const int partner = __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), h_src);
            const bool holder = (lane & 31) < S;
            const int d2 = 4 * (holder ? partner : f);                 // (lane f holds the pivot: never a swap position)
            const uint32_t n_pos = (uint32_t)__builtin_amdgcn_ds_permute(d2, (int)h_pos);
            const float n_resp = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(d2, __builtin_bit_cast(int, h_resp)));
            if (doL || doR) { pos = n_pos; resp = n_resp; }
        }
        if (cut <= k) f = cut; else l = cut;
    }
    if (!heap && l - f >= 2) {
        // std::__insertion_sort of the 2 or 3 entries left = their stable order by descending response
        const int m = l - f;
        const float r0 = readlane_f(resp, f), r1 = readlane_f(resp, f + 1), r2 = readlane_f(resp, m == 3 ? f + 2 : f);
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)pos, f), p1 = (uint32_t)__builtin_amdgcn_readlane((int)pos, f + 1),
                       p2 = (uint32_t)__builtin_amdgcn_readlane((int)pos, m == 3 ? f + 2 : f);
        const bool three = m == 3;
        const int k0 = (r1 > r0 ? 1 : 0) + (three && r2 > r0 ? 1 : 0);                       // entries that end up in front of entry 0
        const int k1 = (r0 >= r1 ? 1 : 0) + (three && r2 > r1 ? 1 : 0);
        const int k2 = (r0 >= r2 ? 1 : 0) + (r1 >= r2 ? 1 : 0);
        const int t = lane - f;
        if (t >= 0 && t < m) {
            if (k0 == t) { pos = p0; resp = r0; }
            else if (k1 == t) { pos = p1; resp = r1; }
            else if (three && k2 == t) { pos = p2; resp = r2; }
        }
    }
    if (lane < n) { Cand e; e.pos = pos; e.resp = resp; a[first + lane] = e; }
    wave_lds_fence();
    if (heap) {
        if (lane == 0) std::__introselect(a + first + f, a + first + k, a + first + l, 0, __gnu_cxx::__ops::__iter_comp_iter(RespGreater()));
        wave_lds_fence();
    }
}

// (inlined on purpose: as a called function its arguments are VGPRs — every branch an EXEC mask, every LDS access a FLAT instruction,
//  and a FLAT access past a small workgroup's LDS allocation is an aperture violation where a ds_read is not)
__device__ __forceinline__ void wave_nth_element(Cand* a, int first, int nth, int last, uint16_t* lpos, uint16_t* rpos, int lane) {
    if (first == last || nth == last) return;
    int depth = 2 * (31 - __clz(last - first));   // std::__lg(n) * 2
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (last - first > 3) {
        if (last - first <= 64) { wave_nth_small(a, first, nth, last, depth, lane); return; }
        if (depth == 0) {
            if (lane == 0) std::__introselect(a + first, a + nth, a + last, 0, __gnu_cxx::__ops::__iter_comp_iter(RespGreater()));
            wave_lds_fence();
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        if (lane == 0) std::__move_median_to_first(a + first, a + first + 1, a + mid, a + last - 1, __gnu_cxx::__ops::__iter_comp_iter(RespGreater()));
        wave_lds_fence();
        const float P = a[first].resp;
        const int f = first + 1, l = last;
        // left stoppers !(value > P) and right stoppers !(P > value), both in ASCENDING positions, in one sweep that reads every entry once,
        // four independent reads in flight (the k-th right stopper from the top is rpos[nR - 1 - k])
        int nL = 0, nR = 0;
        for (int base = f; base < l; base += 256) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const int p = base + 64 * j + lane; v[j] = p < l ? a[p].resp : 0.f; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int p = base + 64 * j + lane;
                const bool in = p < l;
                const bool sL = in && !(v[j] > P), sR = in && !(P > v[j]);
                const unsigned long long mL = __ballot(sL), mR = __ballot(sR);
                if (sL) lpos[nL + __popcll(mL & lt)] = (uint16_t)p;
                if (sR) rpos[nR + __popcll(mR & lt)] = (uint16_t)p;
                nL += __popcll(mL);
                nR += __popcll(mR);
            }
        }
        wave_lds_fence();
        // S = number of leading k with L[k] < R[k]  (L ascending, R descending: a prefix)
        const int nmin = nL < nR ? nL : nR;
        int S = 0;
        for (int kb = 0; kb < nmin; kb += 64) {
            const int k = kb + lane;
            const bool ok = k < nmin && lpos[k] < rpos[nR - 1 - k];
            const unsigned long long m = __ballot(ok);
            const int c = __popcll(m);
            S += c;
            if (c < 64) break;
        }
        for (int kb = 0; kb < S; kb += 64) {
            const int k = kb + lane;
            if (k < S) {
                const int pl = lpos[k], pr = rpos[nR - 1 - k];
                const Cand t = a[pl];
                a[pl] = a[pr];
                a[pr] = t;
            }
        }
        int cut;
        if (S < nL) { cut = lpos[S]; if (S > 0 && (int)rpos[nR - S] < cut) cut = rpos[nR - S]; }
        else cut = rpos[nR - S];
        wave_lds_fence();
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0) std::__insertion_sort(a + first, a + last, __gnu_cxx::__ops::__iter_comp_iter(RespGreater()));
    wave_lds_fence();
}

// reference :79-120 (HarrisResponses, blockSize 7) on the unblurred level; x,y = level coords of the corner
// fp_contract: the last expression as the reference's own build flags fuse it (orbx_params::fp_contract):
// t = fma(a, b, -(c*c)); response = fma(-(a+b), k*(a+b), t) * scale^4
__device__ float harris_response(const uint8_t* img, long long step, int x, int y, int fp_contract) {
    const float scale = 1.0f / ((1 << 2) * 7 * 255.0f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const uint8_t* p0 = img + (long long)(y - 3) * step + (x - 3);
    int a = 0, bb = 0, c = 0;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) {
            const uint8_t* p = p0 + i * step + j;
            const int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
            const int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
            a += Ix * Ix;
            bb += Iy * Iy;
            c += Ix * Iy;
        }
    if (fp_contract) {
        const float cc = (float)c * (float)c, sum = (float)a + (float)bb;
        return __builtin_fmaf(-sum, 0.04f * sum, __builtin_fmaf((float)a, (float)bb, -cc)) * scale_sq_sq;
    }
    return ((float)a * (float)bb - (float)c * (float)c - 0.04f * ((float)a + (float)bb) * ((float)a + (float)bb)) * scale_sq_sq;
}

// One wave per (frame, cell): ordered __ballot filter of the cell's list at its threshold into LDS, Harris responses
// in parallel when selected, wave_nth_element, first nToRetain entries out.
// k_cell_select: four cells per workgroup (one-wave workgroups made the launch dispatch-bound: 151 k workgroups per 1024 VGA frames,
// ~95 us even when every wave exits at once), each wave with a staging area of SEL_SMALL entries (the common case; small LDS
// footprint, many waves per CU).  A cell whose list is longer is flagged in Batch::long_cells and taken by k_cell_select_long
// (one-wave workgroups with the full staging area, each looking after 8 consecutive cells).
constexpr int SEL_SMALL = 384;
__host__ __device__ constexpr int sel_wave_bytes(int entries) { return (entries * ((int)sizeof(Cand) + 4) + 16 + 15) & ~15; }
// one wave; returns false when the cell's list belongs to the other length class (it did nothing)
__device__ __forceinline__ bool cell_select_body(const Batch& b, int frame, int cell, int level, uint8_t* smem, int lds_entries, int min_entries, int lane) {
    const DevGeom& g = b.g;
    const LevelGeom& L = g.lv[level];
    const CellGeom cgeo = b.cells[cell];
    const CellSel s = b.csel[(long long)frame * g.ncells_total + cell];
    if (s.nretain <= 0) return true;
    // the cell's list = its bands' sub-lists in band (= raster) order
    const CellState* bst = b.cstate + (long long)frame * g.nbands_total + cgeo.band0;
    const BandGeom* bgs = b.bands + cgeo.band0;
    Cand* lbase = b.cand + (long long)frame * g.frame_cands + L.cand_base;
    Cand* c = lbase + cgeo.cand_off;
    int n_all = 0;
    for (int k = 0; k < cgeo.nbands; k++) n_all += bst[k].n_all;
    if (n_all < min_entries || (n_all > lds_entries && lds_entries < g.sel_lds_entries)) return false;   // the other class
    Cand* out = b.sel + (long long)frame * g.frame_sel + L.sel_base + s.out_off;
    const float thr = (float)s.thr;
    long long stride;
    const uint8_t* img = plain_plane(b, L, level, frame, stride);
    if (n_all > lds_entries) {
        // rare: list longer than the LDS staging area -> the plain sequential algorithm in global memory
        // (filtered entries are compacted to the front of the cell's area; the write index never passes the read index)
        if (lane == 0) {
            int m = 0;
            for (int k = 0; k < cgeo.nbands; k++) {
                const Cand* bc = lbase + bgs[k].cand_off;
                const int nb = bst[k].n_all;
                for (int i = 0; i < nb; i++) { const Cand e = bc[i]; if (e.resp >= thr) c[m++] = e; }
            }
            if (g.score_type == ORBX_HARRIS_SCORE)
                for (int i = 0; i < m; i++) c[i].resp = harris_response(img, stride, c[i].pos & 0xFFFF, c[i].pos >> 16, g.fp_contract);
            if (m > s.nretain) std::nth_element(c, c + s.nretain, c + m, RespGreater());
            const int keep = min(m, s.nretain);
            for (int i = 0; i < keep; i++) out[i] = c[i];
        }
        return true;
    }
    Cand* lst = reinterpret_cast<Cand*>(smem);
    uint16_t* lpos = reinterpret_cast<uint16_t*>(smem + (size_t)lds_entries * sizeof(Cand));
    uint16_t* rpos = lpos + lds_entries;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int m = 0;
    for (int k = 0; k < cgeo.nbands; k++) {
        const Cand* bc = lbase + bgs[k].cand_off;
        const int nb = bst[k].n_all;
        for (int base = 0; base < nb; base += 256) {           // four loads in flight per lane, then the ordered filter chunk by chunk
            Cand e4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = base + 64 * k + lane;
                e4[k].pos = 0; e4[k].resp = -1.f;
                if (i < nb) e4[k] = bc[i];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (base + 64 * k >= nb) break;
                const int i = base + 64 * k + lane;
                const bool pass = i < nb && e4[k].resp >= thr;
                const unsigned long long mk = __ballot(pass);
                if (pass) lst[m + __popcll(mk & lt)] = e4[k];
                m += __popcll(mk);
            }
        }
    }
    wave_lds_fence();
    if (g.score_type == ORBX_HARRIS_SCORE) {
        for (int i = lane; i < m; i += 64) lst[i].resp = harris_response(img, stride, lst[i].pos & 0xFFFF, lst[i].pos >> 16, g.fp_contract);
        wave_lds_fence();
    }
    if (m > s.nretain) wave_nth_element(lst, 0, s.nretain, m, lpos, rpos, lane);
    const int keep = min(m, s.nretain);
    for (int i = lane; i < keep; i += 64) out[i] = lst[i];
    return true;
}

__global__ __launch_bounds__(256) void k_cell_select(Batch b, int lds_entries) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DevGeom& g = b.g;
    const int wave = wave_id(), lane = (int)threadIdx.x & 63;
    const int id = (int)blockIdx.x * 4 + wave;
    if (id >= b.nframes * g.ncells_total) return;
    const int frame = id / g.ncells_total, cell = id - frame * g.ncells_total;
    const bool done = cell_select_body(b, frame, cell, find_level(g.cell_bases, cell), smem + wave * sel_wave_bytes(lds_entries), lds_entries, 0, lane);
    if (lane == 0) b.long_cells[id] = done ? 0 : 1;          // a flag per cell: no list, no atomics (one counter for ~150 k long cells of a
}                                                            // noise-like batch serialised for 1.3 ms, 64 sharded ones still for 0.5)

// The cells k_cell_select left over (lists beyond its staging area).  A workgroup of four waves looks at the flags of SEL_LONG_CHUNK
// consecutive cells and shares ONE full staging area (sel_lds_entries entries) by list length: lists that fit a quarter of it are taken
// four at a time (one wave each), lists that fit a third three at a time, half two at a time, the rest one at a time with all of it.  A wave's selection
// is latency-bound (~20 us per cell whatever its length: ten partition passes of a few dependent LDS round trips each), so what counts
// is the number of cells in flight per CU, and that is set by the LDS a cell holds.  (Rounds 2-4: one-wave workgroups with the full area
// each, six cells per CU.  S-lowtex lists 440-480 corners per level-0 cell, S-noise 910 / 570 / 520 on levels 0 / 1 / 2: 0.42 and 1.1 ms
// per 1024 frames.  Round 5 first tried the opposite, four waves on ONE list — block-wide stopper scans, three barriers per pass — and
// lost: 1.37 -> 1.67 ms on S-noise, the passes over short ranges dominate and stay serial.)  Normally no cell is flagged and the launch
// is a few thousand workgroups that exit.
#ifndef ORBX_SEL_LONG_CHUNK
#define ORBX_SEL_LONG_CHUNK 8
#endif
constexpr int SEL_LONG_CHUNK = ORBX_SEL_LONG_CHUNK, SEL_LONG_WAVES = 4;
__host__ __device__ constexpr int sel_long_bytes(int entries) {      // the shared area: four quarter areas, three thirds, two halves or one full one
    int m = 0;
    for (int share = 1; share <= 4; share++) { const int v = share * sel_wave_bytes((entries + share - 1) / share); m = v > m ? v : m; }
    return m;
}
__global__ __launch_bounds__(SEL_LONG_WAVES * 64) void k_cell_select_long(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DevGeom& g = b.g;
    const int lane = (int)threadIdx.x & 63, wave = wave_id(), total = b.nframes * g.ncells_total;
    const int id0 = (int)blockIdx.x * SEL_LONG_CHUNK;
    static_assert(SEL_LONG_CHUNK <= 64, "one flag per lane");
    const int quarter = (g.sel_lds_entries + 3) / 4, third = (g.sel_lds_entries + 2) / 3, half = (g.sel_lds_entries + 1) / 2;
    // every wave reads the same flags and list lengths (lane i: cell id0 + i)
    int n_all = 0;
    const bool mine = lane < SEL_LONG_CHUNK && id0 + lane < total && b.long_cells[id0 + lane] != 0;
    if (mine) {
        const int id = id0 + lane, frame = id / g.ncells_total, cell = id - frame * g.ncells_total;
        const CellGeom cgeo = b.cells[cell];
        const CellState* bst = b.cstate + (long long)frame * g.nbands_total + cgeo.band0;
        for (int k = 0; k < cgeo.nbands; k++) n_all += bst[k].n_all;
    }
    const unsigned long long mQ = __ballot(mine && n_all <= quarter), mT = __ballot(mine && n_all > quarter && n_all <= third),
                             mH = __ballot(mine && n_all > third && n_all <= half), mF = __ballot(mine && n_all > half);
    if (!(mQ | mT | mH | mF)) return;
    // class c: `share` waves work side by side, wave w on the class's cells of rank w, w + share, ... in its own part of the area
    const unsigned long long cls_m[4] = {mQ, mT, mH, mF};
    const int cls_share[4] = {4, 3, 2, 1}, cls_entries[4] = {quarter, third, half, g.sel_lds_entries};
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
        unsigned long long m = cls_m[c];
        const int share = cls_share[c], entries = cls_entries[c];
        if (!m) continue;                                      // (workgroup-uniform)
        if (wave < share) {
            uint8_t* area = smem + wave * sel_wave_bytes(entries);
            for (int r = 0; m; r++) {
                const int id = id0 + __ffsll((long long)m) - 1;
                m &= m - 1;
                if (r % share != wave) continue;
                const int frame = id / g.ncells_total, cell = id - frame * g.ncells_total;
                (void)cell_select_body(b, frame, cell, find_level(g.cell_bases, cell), area, entries, 0, lane);
                wave_lds_fence();
            }
        }
        __syncthreads();                                       // the parts change owners
    }
}

// reference :697-701 (per-level cap), same scheme; one wave
__device__ __forceinline__ void level_select_body(const Batch& b, int frame, int level, uint8_t* smem, int lane) {
    const DevGeom& g = b.g;
    const LevelGeom& L = g.lv[level];
    const int total = b.level_total[frame * MAX_LEVELS + level];
    int n = total;
    if (total > L.ndesired) {
        n = L.ndesired;
        if (n > 0) {
            Cand* v = b.sel + (long long)frame * g.frame_sel + L.sel_base;
            if (total > g.sel_lds_entries) {
                if (lane == 0) std::nth_element(v, v + n, v + total, RespGreater());
            } else {
                Cand* lst = reinterpret_cast<Cand*>(smem);
                uint16_t* lpos = reinterpret_cast<uint16_t*>(smem + (size_t)g.sel_lds_entries * sizeof(Cand));
                uint16_t* rpos = lpos + g.sel_lds_entries;
                // four loads in flight per lane (one per iteration made the gather a chain of round trips: 7 for a VGA level 0)
                for (int i0 = 0; i0 < total; i0 += 256) {
                    Cand e[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int i = i0 + 64 * k + lane; if (i < total) e[k] = v[i]; }
#pragma unroll
                    for (int k = 0; k < 4; k++) { const int i = i0 + 64 * k + lane; if (i < total) lst[i] = e[k]; }
                }
                wave_lds_fence();
                wave_nth_element(lst, 0, n, total, lpos, rpos, lane);
                for (int i = lane; i < n; i += 64) v[i] = lst[i];
            }
        }
    }
    if (lane == 0) b.level_count[frame * MAX_LEVELS + level] = n;
}

__global__ __launch_bounds__(64) void k_level_select(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DevGeom& g = b.g;
    const int frame = blockIdx.x / g.nlevels, level = blockIdx.x - frame * g.nlevels;
    level_select_body(b, frame, level, smem, (int)threadIdx.x);
}

// diagnostics: wave_nth_element on a caller-supplied response list (pos carries the original index)
__global__ __launch_bounds__(64) void k_debug_nth(const float* resp, int n, int nth, int* out_idx) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Cand* lst = reinterpret_cast<Cand*>(smem);
    uint16_t* lpos = reinterpret_cast<uint16_t*>(smem + (size_t)n * sizeof(Cand));
    uint16_t* rpos = lpos + n;
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) { Cand e; e.pos = (uint32_t)i; e.resp = resp[i]; lst[i] = e; }
    wave_lds_fence();
    wave_nth_element(lst, 0, nth, n, lpos, rpos, lane);
    for (int i = lane; i < n; i += 64) out_idx[i] = (int)lst[i].pos;
}
int launch_debug_nth(const float* d_resp, int n, int nth, int* d_out) {
    const size_t lds = (size_t)n * (sizeof(Cand) + 4) + 16;
    if (lds > 160 * 1024) return ORBX_ERR_ARG;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_debug_nth), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
    hipLaunchKernelGGL(k_debug_nth, dim3(1), dim3(64), lds, 0, d_resp, n, nth, d_out);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

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

// FAST and the blur both depend on the pyramid only.  A launch group that cannot fill the chip (the one-frame drop-in call) runs
// them side by side in ONE launch: the first blocks of a frame blur short strips (4 waves = 4 strips), the rest are cell bands.
// (Two streams would do the same for a full batch - launch_extract forks there - but a fork / join across hardware queues costs
// ~8 us each way, as much as either kernel takes on one frame.)
template <bool ALIGNED, bool SMALL>
__global__ __launch_bounds__(SMALL ? FAST_SMALL.threads : FAST_LARGE.threads) void k_fast_blur(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr FastShape A = SMALL ? FAST_SMALL : FAST_LARGE;
    constexpr int NW = A.threads / 64;
    const DevGeom& g = b.g;
    const int nblur = (g.nbtiles_total_s + NW - 1) / NW;
    const int per_frame = nblur + g.nbands_total;
    const int frame = blockIdx.x / per_frame, item = blockIdx.x - frame * per_frame;
    if (item < nblur) {
        const int t = item * NW + wave_id();
        if (t < g.nbtiles_total_s) blur_strip<ALIGNED, BLUR_ROWS_SMALL>(b, frame, t);
    } else fast_band_task<ALIGNED, A.threads, A.ppt>(b, frame, item - nblur, smem);
}

// ------------------------------------------------------------------------------------ orientation + rBRIEF + output
// Four keypoints per wave, 16 lanes each (round 2; one wave per keypoint before: every lane of a wave then repeated the same
// fastAtan2 + double-precision sin / cos, ~28 % of the kernel's instructions, and 47 v_readfirstlane + 64-bit tap addresses).
//   IC_Angle (reference :124-151): the 31 x 31 box is cut into 31 rows x 8 dwords (u = -15 .. 16); a lane owns dword column
//     lane & 7 and the rows of one parity: 16 (unaligned) dword loads, the circle as byte masks from an LDS table (constant LDS
//     offsets), two v_dot4_u32_u8 per dword (sum of (u + 15) I and sum of I), the row weight as a multiply-add; 4-step reduction.
//   rBRIEF (:154-194): lane i of a group evaluates tests i, i + 16, ..., i + 240.  The 37 x 37 window the rotated pattern can reach
//     (|offset| <= 18) is first copied from the blurred level into LDS (LDS-DMA, 6 wave instructions per keypoint): 512 scattered
//     byte gathers per keypoint straight from global memory kept the kernel bound by the L1's cache-line rate (one wave-load
//     touched 40-64 lines), the LDS serves them at bank speed.  The pattern comes from an LDS table of floats (one 16-byte read
//     per test, no unpacking); the rotated coordinates are rounded with v_rndne and the tap offset iy * pitch + ix is formed in
//     float (exact) and converted once.  The 16 x 16 test bits of a group are transposed into descriptor halfwords by ds_swizzle.
//   Keypoints closer than 19 px to an edge may read the level's UNBLURRED reflect-101 border: their group of lanes
//     takes its taps from global memory with the reflection in the index math.
// Waves are formed per level (slots padded to multiples of 4), so the level is wave-uniform and its geometry scalar.
constexpr int DESC_KPW = 4;
constexpr int DESC_WIN_PITCH = 40, DESC_WIN_ROWS = 37, DESC_WIN_BYTES = DESC_WIN_PITCH * DESC_WIN_ROWS;   // 37 px + up to 3 px of dword alignment per row

// FMA: the two rotation expressions of computeOrbDescriptor as the reference's own build flags contract them (orbx_params::fp_contract):
// `x*b + y*a` -> fma(x, b, y*a), `x*a - y*b` -> fma(x, a, -(y*b)); false: unfused (ISO evaluation, the default).
template <bool FMA>
__global__ __launch_bounds__(DESC_WAVES * 64) void k_describe(Batch b) {
#if ORBX_DESC_PACKED_PATTERN
    __shared__ uint32_t s_pat[256];                                    // test t: x0, y0, x1, y1 as the four int8 of c_pattern[t] (1 KB: six workgroups per CU; as floats, 4 KB: five)
#else
    __shared__ __attribute__((aligned(16))) float s_pat[256 * 4];     // test t: x0, y0, x1, y1
#endif
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[256];       // circle byte masks of the 31 x 8 patch dwords (slots 248.. = 0)
    __shared__ __attribute__((aligned(16))) uint8_t s_win[DESC_WAVES * DESC_KPW * DESC_WIN_BYTES];   // per keypoint: 37 rows x 40 bytes of the blurred level
    const DevGeom& g = b.g;
    int frame, wgi;
    if (!frame_item(b, blockIdx.x, (g.nquads + DESC_WAVES - 1) / DESC_WAVES, frame, wgi)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = lane >> 4, li = lane & 15;
    // The wave's chain of dependent memory round trips sets this kernel's pace as much as its arithmetic, so everything is requested
    // as early as its address is known: the per-level counts (one load, lane l holds level l) and the frame status first, then the
    // wave's keypoints, and only then the LDS tables are built (their barrier rides on those loads); the 37 x 37 windows of the
    // blurred level follow by LDS-DMA as soon as the keypoints are there, in flight during IC_Angle and the angle arithmetic.
#ifndef ORBX_DESC_EARLY_PATTERN
#define ORBX_DESC_EARLY_PATTERN 1      // (0.701 -> 0.692 ms per 1024 VGA frames)
#endif
    // (round 5) the thread's word of the BRIEF pattern is requested FIRST: loads return in order, so the tables can be built while the
    // counts and the keypoint are still on their way — before, the pattern load started only after the keypoint had arrived: one
    // dependent round trip more in front of the tables' barrier
    const uint32_t pk_first = ORBX_DESC_EARLY_PATTERN ? c_pattern[tid & 255] : 0u;
#ifndef ORBX_DESC_SCALAR_LOADS
#define ORBX_DESC_SCALAR_LOADS 1       // (0.692 -> 0.686)
#endif
    const int32_t* counts = b.level_count + frame * MAX_LEVELS;
    int cl = 0, st0 = 0;
    if (!ORBX_DESC_SCALAR_LOADS) {
        cl = lane < g.nlevels ? counts[lane] : 0;
        st0 = b.status[frame];
    }
    const int quad = wgi * DESC_WAVES + wave_id();
    const bool live = quad < g.nquads;
    const int level = __builtin_amdgcn_readfirstlane(find_level(g.quad_bases, live ? quad : 0));
    // the level's geometry as scalars (wave-uniform by construction; pinned so that nothing is re-read through per-lane addresses)
    const LevelGeom& LG = g.lv[level];
    struct { int w, h, stride, plane_off, sel_base, quad_base; float scale, kp_size; } L = {
        __builtin_amdgcn_readfirstlane(LG.w), __builtin_amdgcn_readfirstlane(LG.h), __builtin_amdgcn_readfirstlane(LG.stride),
        __builtin_amdgcn_readfirstlane(LG.plane_off), __builtin_amdgcn_readfirstlane(LG.sel_base), __builtin_amdgcn_readfirstlane(LG.quad_base),
        __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, LG.scale))),
        __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, LG.kp_size)))};
    // The group's keypoint is requested before the counts are there (round 4: its address depends on the quad only; the counts decide
    // whether it is one — the slot index is clamped into the level's list, what lies behind the list's end is never used).  One
    // dependent memory round trip less in front of the two gathers.
    const int k0 = (quad - L.quad_base) * DESC_KPW;
    Cand kp;
    typedef int v8i_s __attribute__((ext_vector_type(8)));
    v8i_s s_cnt0 = {0, 0, 0, 0, 0, 0, 0, 0}, s_cnt1 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ORBX_DESC_SCALAR_LOADS) {
        // (round 5) The wave's four keypoints, the per-level counts and the frame status are wave-uniform data: SCALAR loads.  This kernel
        // keeps the vector-memory front end 0.7-0.8 busy with its gathers, and a small vector load queues behind the gathers of the ~20
        // other waves of the CU (the keypoint wait was 23 % of a wave's life, profiles/r04_describe_wave_phases.txt); the scalar cache path
        // does not.  (Written by the selection kernels of earlier launches: coherent at the kernel boundary.)
        const int sel_cap = __builtin_amdgcn_readfirstlane(LG.sel_cap);
        const int ks = __builtin_amdgcn_readfirstlane(max(min(k0, sel_cap - DESC_KPW), 0));        // (the sel block carries 4 slots of padding)
        const Cand* kp4 = b.sel + ((long long)frame * g.frame_sel + L.sel_base + ks);
        const int32_t* stp = b.status + frame;
        // REQUIRES of Batch::sel: 4 readable Cand slots behind every level's list (the 32-byte load below may start up to 3 entries in front of the list's
        // last slot: ensure_geometry pads d_sel by DESC_KPW entries) and level_count rows of MAX_LEVELS >= 16 ints (two x8 loads).
        static_assert(MAX_LEVELS >= 16, "the per-level counts are fetched as two s_load_dwordx8");
        v8i_s kq;
        int sst;
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %5, 0x0\n\ts_load_dwordx8 %2, %5, 0x20\n\ts_load_dword %3, %6, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(kq), "=&s"(s_cnt0), "=&s"(s_cnt1), "=&s"(sst) : "s"(kp4), "s"(counts), "s"(stp) : "memory");
        st0 = sst;
        const int e = min(max(k0 + grp, 0), sel_cap - 1) - ks;       // 0 .. 3
        kp.pos = (uint32_t)(e == 0 ? kq[0] : e == 1 ? kq[2] : e == 2 ? kq[4] : kq[6]);
        kp.resp = __builtin_bit_cast(float, e == 0 ? kq[1] : e == 1 ? kq[3] : e == 2 ? kq[5] : kq[7]);
    } else kp = b.sel[(long long)frame * g.frame_sel + L.sel_base + min(max(k0 + grp, 0), __builtin_amdgcn_readfirstlane(LG.sel_cap) - 1)];
    auto build_tables = [&]() {
    for (int t = tid; t < 256; t += DESC_WAVES * 64) {
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
    }
    };
    if (ORBX_DESC_EARLY_PATTERN) build_tables();
    int out_base = 0, total = 0, cnt = 0;
    for (int l = 0; l < g.nlevels; l++) {
        int c;
        if (ORBX_DESC_SCALAR_LOADS) {
            c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { if (l == i) c = s_cnt0[i]; if (l == 8 + i) c = s_cnt1[i]; }
        } else c = __builtin_amdgcn_readlane(cl, l);
        if (l < level) out_base += c;
        if (l == level) cnt = c;
        total += c;
    }
    const bool work = live && k0 < cnt && total <= b.cap && __builtin_amdgcn_readfirstlane(st0) == ORBX_OK;
    const bool valid = work && k0 + grp < cnt;
    const int k = valid ? k0 + grp : (work ? k0 : 0);       // idle groups shadow the wave's first keypoint (results dropped)
    if (!valid) {
        kp.pos = (uint32_t)__builtin_amdgcn_readlane((int)kp.pos, 0);
        kp.resp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kp.resp), 0));
    }
    if (!ORBX_DESC_EARLY_PATTERN) build_tables();
#ifndef ORBX_DESC_LATE_BARRIER
#define ORBX_DESC_LATE_BARRIER 1     // the tables' barrier behind the window DMA issue (round 5: 0.727 -> 0.701 ms; 0 = in front of it, rounds 2-4; behind the
                                     // patch loads' issue as well: 0.714)
#endif
    if (!ORBX_DESC_LATE_BARRIER) __syncthreads();
    if (quad == 0 && lane == 0) {
        int st = st0, tot = total;
        if (tot > b.cap) { st = ORBX_ERR_CAPACITY; tot = 0; }
        b.out_n[frame] = st == ORBX_OK ? tot : 0;
        if (b.out_status) b.out_status[frame] = st;
    }
    if (!work) return;
    const int x = kp.pos & 0xFFFF, y = kp.pos >> 16;
    const uint8_t* plain;
    unsigned pstride;                                        // rows < 2^24 bytes, planes < 2^31 bytes (host-checked)
    if (level == 0) { pstride = (unsigned)b.img_row_stride; plain = b.img + (long long)frame * b.img_frame_stride; }
    else { pstride = (unsigned)L.stride; plain = b.pyr + (long long)frame * g.frame_plane_bytes + L.plane_off; }
    const uint8_t* blur = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    // rounded pattern offsets never exceed 18 px (|(-13,-13)| = 18.4): keypoints at least 19 px from every edge — all but the
    // outermost ring of candidates — take the branch-free path with the window in LDS
    const bool interior = x >= 19 && y >= 19 && x < L.w - 19 && y < L.h - 19;
    {
        // window of keypoint q: rows y-18 .. y+18, 40 bytes from the aligned start at or left of x-18, row after row (pitch 40 = 10
        // dwords), i.e. 370 consecutive LDS dwords: 6 global_load_lds_dword of the whole wave per keypoint (lane i of instruction n
        // fetches dword e = 64 n + i: row e / 10, column e % 10; the lane offsets are the same for the four keypoints).  A row's last
        // dword may reach past the level's last pixel: it stays inside the blurred plane (rows are padded to 64, a next row exists)
        // and those bytes are never tapped.
        typedef const void __attribute__((address_space(1))) * gptr_t;
        typedef void __attribute__((address_space(3))) * lptr_t;
        unsigned eoff[6];
#pragma unroll
        for (int n = 0; n < 6; n++) {
            const unsigned e = 64u * n + (unsigned)lane, er = (e * 205u) >> 11;           // e / 10 for e < 1029
            eoff[n] = __umul24(er, (unsigned)L.stride) + 4u * (e - 10u * er);
        }
        uint8_t* win0 = s_win + wave_id() * DESC_KPW * DESC_WIN_BYTES;
#pragma unroll
        for (int q = 0; q < DESC_KPW; q++) {
            const unsigned posq = (unsigned)__builtin_amdgcn_readlane((int)kp.pos, 16 * q);
            const bool inq = __builtin_amdgcn_readlane((int)interior, 16 * q) != 0;
            if (!inq || (q > 0 && k0 + q >= cnt)) continue;                              // wave-uniform
            const int xq = posq & 0xFFFF, yq = posq >> 16;
            const uint8_t* srcq = blur + __umul24((unsigned)(yq - 18), (unsigned)L.stride) + (unsigned)((xq - 18) & ~3);
#pragma unroll
            for (int n = 0; n < 6; n++)
                if (n < 5 || lane < DESC_WIN_ROWS * 10 - 320)
                    __builtin_amdgcn_global_load_lds((gptr_t)(srcq + eoff[n]), (lptr_t)(win0 + q * DESC_WIN_BYTES + 256 * n), 4, 0, 0);
        }
    }

    if (ORBX_DESC_LATE_BARRIER) {
        // the tables' barrier behind the window DMA issue: only the LDS writes have to be complete (no vmcnt wait: the DMA stays in flight);
        // a wave that returned above has left the workgroup's barrier count with its s_endpgm
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // IC_Angle on the unblurred level (:705-706 run before the blur)
    int m10, m01;
    {
        // The 31 x 32-byte patch as FOUR 16-byte loads per lane (lane = row 8 n + li / 2, half li % 2 of the row; rounds 1-3: sixteen dword
        // loads per lane, a row parity and a dword column each) and the circle masks of a lane's four dwords as one ds_read_b128: 12 vector-memory
        // and 12 LDS instructions less per wave for the same bytes and the same sums (-1 % on the VGA stream, -6 % on the 1080p one).
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        const int rsub = li >> 1, hf = li & 1;
        const unsigned off0 = (unsigned)(x - HALF_PATCH + 16 * hf) + __umul24((unsigned)(y - HALF_PATCH + rsub), pstride);
        u32x4_u P[4];
#pragma unroll
        for (int n = 0; n < 4; n++) P[n] = *reinterpret_cast<const u32x4_u*>(plain + (off0 + (unsigned)(8 * n) * pstride));     // row 31 (n = 3, rsub = 7) is masked, still inside the level
        uint32_t uw[4];
#pragma unroll
        for (int d = 0; d < 4; d++) uw[d] = (uint32_t)(16 * hf + 4 * d) * 0x01010101u + 0x03020100u;      // u + 15 of the dword's four pixels
        uint32_t a_su = 0, a_si = 0, a_r = 0;
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const uint4 mk = *reinterpret_cast<const uint4*>(s_mask + (8 * (8 * n + rsub) + 4 * hf));
            const uint32_t mm[4] = {mk.x, mk.y, mk.z, mk.w};
            uint32_t srow = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t Im = P[n][d] & mm[d];
                srow = __builtin_amdgcn_udot4(Im, 0x01010101u, srow, false);
                a_su = __builtin_amdgcn_udot4(Im, uw[d], a_su, false);
            }
            a_si += srow;
            a_r = __umul24(srow, (uint32_t)(8 * n)) + a_r;     // sum of (row - rsub) * rowsum
        }
        const int p10 = (int)a_su - HALF_PATCH * (int)a_si;
        const int p01 = (rsub - HALF_PATCH) * (int)a_si + (int)a_r;
        m10 = row16_sum(p10); m01 = row16_sum(p01);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // rotated BRIEF on the blurred level (:154-194)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    sincosf_orb(angle * factorPI, &sn, &cs);
#if ORBX_DESC_PACKED_PATTERN
    struct PatRow {          // one ds_read_b32 and four v_cvt_f32_i32 with a sign-extending byte select (SDWA) per test
        const uint32_t* p;
        __device__ __forceinline__ float4 operator[](int i) const {
            const uint32_t pk = p[i];
            return make_float4((float)(int)(int8_t)pk, (float)(int)(int8_t)(pk >> 8), (float)(int)(int8_t)(pk >> 16), (float)(int)(int8_t)(pk >> 24));
        }
    } pat{s_pat + li};
#else
    const float4* pat = reinterpret_cast<const float4*>(s_pat) + li;
#endif
    uint32_t mybits = 0;                                        // bit j: test li + 16 j
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the wave's window DMA has landed (issued before IC_Angle)
    wave_lds_fence();                                           // the windows are private to this wave: no workgroup barrier
    if (interior) {
        const uint8_t* win = s_win + (wave_id() * DESC_KPW + grp) * DESC_WIN_BYTES;
        const int xa = (x - 18) & ~3;
        const uint8_t* ctr = win + 18 * DESC_WIN_PITCH + (x - xa);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float4 P = pat[16 * j];
            const float fy0 = FMA ? __builtin_fmaf(P.x, sn, P.y * cs) : P.x * sn + P.y * cs, fx0 = FMA ? __builtin_fmaf(P.x, cs, -(P.y * sn)) : P.x * cs - P.y * sn;
            const float fy1 = FMA ? __builtin_fmaf(P.z, sn, P.w * cs) : P.z * sn + P.w * cs, fx1 = FMA ? __builtin_fmaf(P.z, cs, -(P.w * sn)) : P.z * cs - P.w * sn;
            // cvRound (ties to even) of both coordinates, then iy * pitch + ix exactly in float (the fused multiply-add rounds nothing here)
            const int o0 = (int)__builtin_fmaf(__builtin_rintf(fy0), (float)DESC_WIN_PITCH, __builtin_rintf(fx0));
            const int o1 = (int)__builtin_fmaf(__builtin_rintf(fy1), (float)DESC_WIN_PITCH, __builtin_rintf(fx1));
            const int v0 = ctr[o0], v1 = ctr[o1];
            mybits |= (uint32_t)(v0 < v1) << j;
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < 16; j++) {
            const float4 P = pat[16 * j];
            int val[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float px = e ? P.z : P.x, py = e ? P.w : P.y;
                const int iy = cv_round_f(FMA ? __builtin_fmaf(px, sn, py * cs) : px * sn + py * cs);
                const int ix = cv_round_f(FMA ? __builtin_fmaf(px, cs, -(py * sn)) : px * cs - py * sn);
                int X = x + ix, Y = y + iy;
                // inside the level: blurred pixel.  Outside (<= 2 px, only for keypoints 16..17 px from the edge): the
                // reference reads the level's UNBLURRED reflect-101 border; one reflection suffices.
                const bool inside = (unsigned)X < (unsigned)L.w && (unsigned)Y < (unsigned)L.h;
                X = X < 0 ? -X : (X >= L.w ? 2 * L.w - 2 - X : X);
                Y = Y < 0 ? -Y : (Y >= L.h ? 2 * L.h - 2 - Y : Y);
                const uint8_t* base = inside ? blur : plain;
                const unsigned st = inside ? (unsigned)L.stride : pstride;
                val[e] = base[__umul24((unsigned)Y, st) + (unsigned)X];
            }
            mybits |= (uint32_t)(val[0] < val[1]) << j;
        }
    }
    // Lane li of a group holds the outcomes of the tests li, li + 16, ..., li + 240 in bits 0..15; descriptor halfword j is bit j of
    // the 16 lanes: a 16 x 16 bit-matrix transpose inside the group, four butterfly stages (partner lane ^ s by ds_swizzle, the LDS
    // crossbar; keep half of the own bits, take the other half from the partner shifted by s).  (16 ballots and a 16-way select of
    // SGPR pairs took ~130 instructions; this takes ~30.)
    uint32_t half = mybits;
    auto stage = [&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr uint32_t M0 = s == 8 ? 0x00FFu : s == 4 ? 0x0F0Fu : s == 2 ? 0x3333u : 0x5555u;     // bit positions with (pos & s) == 0
        const bool hi = (li & s) != 0;
        const uint32_t y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)half, (s << 10) | 0x1F);        // lane ^ s
        const uint32_t ysh = hi ? (y >> s) : (y << s);
        const uint32_t mk = hi ? (~M0 & 0xFFFFu) : M0;
        half = (half & mk) | (ysh & ~mk & 0xFFFFu);
    };
    stage(std::integral_constant<int, 8>{}); stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 2>{}); stage(std::integral_constant<int, 1>{});
    if (!valid) return;
    const int out_idx = out_base + k;
    reinterpret_cast<uint16_t*>(b.out_desc + ((long long)frame * b.cap + out_idx) * 32)[li] = (uint16_t)half;   // lane li stores halfword li
    if (li == 0) {
        orbx_keypoint o;
        o.x = (float)x; o.y = (float)y;
        if (level != 0) { o.x = o.x * L.scale; o.y = o.y * L.scale; }   // :769-775
        o.size = L.kp_size;
        o.angle = angle;
        o.response = kp.resp;
        o.octave = level;
        o.class_id = -1;
        b.out_kps[(long long)frame * b.cap + out_idx] = o;
    }
}

// ------------------------------------------------------------------------------------ launcher
#define ORBX_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e_ = hipGetLastError();                    \
        if (e_ != hipSuccess) return ORBX_ERR_DEVICE;         \
    } while (0)

struct StageScope {   // records (start, stop) events around one stage when timing is on
    StageTimer* t; hipStream_t s; int stage;
    StageScope(StageTimer* t_, hipStream_t s_, int stage_) : t(t_ && t_->enabled ? t_ : nullptr), s(s_), stage(stage_) {
        if (t) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); t->pool.push_back(e); } else t = nullptr; }
    }
    ~StageScope() {
        if (t) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); t->pool.push_back(e); t->pool_stage.push_back(stage); } else { (void)hipEventDestroy(t->pool.back()); t->pool.pop_back(); } }
    }
};

static bool use_on_demand(const Batch& b, const HostGeom& hg, int stop_after) {
    return b.blur_on_demand && b.nframes >= PYR_FUSED_MAX_FRAMES && stop_after < 0 && describe_od_supported(b, hg);
}

int launch_extract(const Batch& b, const HostGeom& hg, hipStream_t stream, int stop_after, StageTimer* timer, const SideStream* side, int phases) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
    if (F <= 0) return ORBX_OK;
    const bool fused_pyramid = g.npyr_groups > 0 && F < PYR_FUSED_MAX_FRAMES;
    // per-frame status starts at ORBX_OK: a fill launch for full batches, folded into the first k_pyramid launch otherwise
    if ((phases & ORBX_PHASE_PYRAMID) && !fused_pyramid && hipMemsetAsync(b.status, 0, sizeof(int32_t) * F, stream) != hipSuccess) return ORBX_ERR_DEVICE;
    if (phases & ORBX_PHASE_PYRAMID) {
        StageScope sc(timer, stream, ST_PYRAMID);
        const bool al0 = (((uintptr_t)b.img | (uintptr_t)b.img_row_stride | (uintptr_t)b.img_frame_stride) & 3) == 0;
        // Fused launches when the batch is too small to fill the chip (the drop-in call: one frame): there the chain of dependent
        // launches is the cost (189 vs 236 us per VGA frame); a full batch prefers the leaner per-level kernels (0.72 vs 0.81 ms per 1024
        // frames: the cones recompute their overlap and synchronise per level).
        if (fused_pyramid) {
            for (int gi = 0; gi < g.npyr_groups; gi++) {
                const PyrGroup& pg = g.pyr[gi];
                const bool al = pg.l0 > 0 || al0;
                dim3 grid(pg.ntx, pg.nty, F);
                if (al) hipLaunchKernelGGL(k_pyramid<true>, grid, dim3(256), (size_t)pg.lds_bytes, stream, b, gi);
                else hipLaunchKernelGGL(k_pyramid<false>, grid, dim3(256), (size_t)pg.lds_bytes, stream, b, gi);
                ORBX_LAUNCH_CHECK();
            }
        } else {
            for (int l = 1; l < g.nlevels; l++) {
                const LevelGeom& L = g.lv[l];
                dim3 grid(frame_item_blocks(b, ((L.w + 255) / 256) * ((L.h + RZ_ROWS - 1) / RZ_ROWS)));
                const bool al = l > 1 || al0;
                const size_t lds = (size_t)L.rz_pitch * L.rz_rows;
                if (L.rz_window) {
                    if (al) hipLaunchKernelGGL((k_resize<true, true>), grid, dim3(256), lds, stream, b, l);
                    else hipLaunchKernelGGL((k_resize<false, true>), grid, dim3(256), lds, stream, b, l);
                } else {
                    if (al) hipLaunchKernelGGL((k_resize<true, false>), grid, dim3(256), lds, stream, b, l);
                    else hipLaunchKernelGGL((k_resize<false, false>), grid, dim3(256), lds, stream, b, l);
                }
                ORBX_LAUNCH_CHECK();
            }
        }
    }
    if (stop_after == ST_PYRAMID) return ORBX_OK;
    if (!(phases & ORBX_PHASE_DETECT)) {
        if (stop_after >= 0 && stop_after < ST_DESCRIBE) return ORBX_OK;      // the diagnostics' early stop lies inside the part this call skips
        goto describe;
    }
    {
    auto launch_blur = [&](hipStream_t st) -> int {
        StageScope sc(timer, st, ST_BLUR);
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
    };
    // (a launch group that cannot fill the chip keeps the blur in line: its short strips take ~6 us, the fork and the join across
    //  two hardware queues cost 8 us each)
    // blur on demand (k_describe_od.hip): full launch groups only (the one-frame call keeps the blur inside the FAST launch), never under the
    // diagnostics' early stops (they fetch the blurred plane)
    const bool on_demand = use_on_demand(b, hg, stop_after);
    const bool overlap = side && side->aux && stop_after < 0 && F >= PYR_FUSED_MAX_FRAMES && !on_demand;
    const bool fuse_blur = F < PYR_FUSED_MAX_FRAMES && !b.xcd_affinity;    // k_fast_blur
    {
        StageScope sc(timer, stream, ST_FAST_CELLS);
        const bool aligned = (((uintptr_t)b.img | (uintptr_t)b.img_row_stride | (uintptr_t)b.img_frame_stride) & 3) == 0;
        const size_t lds = (size_t)g.fast_lds_bytes;
        auto launch = [&](auto kern, int threads) -> bool {
            if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
            hipLaunchKernelGGL(kern, dim3(frame_item_blocks(b, g.nbands_total)), dim3(threads), lds, stream, b);
            return true;
        };
        constexpr FastShape A = FAST_SMALL, B = FAST_LARGE;
        bool ok;
        if (fuse_blur) {
            const int threads = g.fast_small ? A.threads : B.threads;
            const int per_frame = (g.nbtiles_total_s + threads / 64 - 1) / (threads / 64) + g.nbands_total;
            auto launch2 = [&](auto kern) -> bool {
                if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
                hipLaunchKernelGGL(kern, dim3(F * per_frame), dim3(threads), lds, stream, b);
                return true;
            };
            if (g.fast_small) ok = aligned ? launch2(k_fast_blur<true, true>) : launch2(k_fast_blur<false, true>);
            else ok = aligned ? launch2(k_fast_blur<true, false>) : launch2(k_fast_blur<false, false>);
        } else if (g.fast_small) ok = aligned ? launch(k_fast_cells<true, A.threads, A.ppt>, A.threads) : launch(k_fast_cells<false, A.threads, A.ppt>, A.threads);
        else ok = aligned ? launch(k_fast_cells<true, B.threads, B.ppt>, B.threads) : launch(k_fast_cells<false, B.threads, B.ppt>, B.threads);
        if (!ok) return ORBX_ERR_DEVICE;
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_FAST_CELLS) return ORBX_OK;
    if (overlap) {
        // fork: the VALU-bound blur runs on the side stream next to the latency-bound quota / retainBest kernels
        if (hipEventRecord(side->fork, stream) != hipSuccess || hipStreamWaitEvent(side->aux, side->fork, 0) != hipSuccess) return ORBX_ERR_DEVICE;
        if (launch_blur(side->aux) != ORBX_OK) return ORBX_ERR_DEVICE;
        if (hipEventRecord(side->join, side->aux) != hipSuccess) return ORBX_ERR_DEVICE;
    }
    {
        StageScope sc(timer, stream, ST_QUOTA);
        const size_t lds = (size_t)g.quota_cells * QUOTA_LDS_PER_CELL;
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_quota), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
        hipLaunchKernelGGL(k_quota, dim3(F * g.nlevels), dim3(64), lds, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_QUOTA) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_CELL_SELECT);
        // a launch group too small to fill the chip takes ONE launch with the full staging area per wave (a dependent launch costs
        // more than the occupancy gains); otherwise short lists first, then the (usually empty) list of long cells
        const int small = F < PYR_FUSED_MAX_FRAMES ? g.sel_lds_entries : std::min(SEL_SMALL, g.sel_lds_entries);
        const size_t lds = (size_t)4 * sel_wave_bytes(small);
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cell_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
        hipLaunchKernelGGL(k_cell_select, dim3((F * g.ncells_total + 3) / 4), dim3(256), lds, stream, b, small);
        ORBX_LAUNCH_CHECK();
        if (small < g.sel_lds_entries) {
            const size_t ldsl = (size_t)sel_long_bytes(g.sel_lds_entries);
            if (ldsl > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cell_select_long), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsl) != hipSuccess) return ORBX_ERR_DEVICE;
            hipLaunchKernelGGL(k_cell_select_long, dim3((F * g.ncells_total + SEL_LONG_CHUNK - 1) / SEL_LONG_CHUNK), dim3(SEL_LONG_WAVES * 64), ldsl, stream, b);
            ORBX_LAUNCH_CHECK();
        }
    }
    if (stop_after == ST_CELL_SELECT) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_LEVEL_SELECT);
        const size_t lds = (size_t)g.sel_lds_level;
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
        hipLaunchKernelGGL(k_level_select, dim3(F * g.nlevels), dim3(64), lds, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_LEVEL_SELECT) return ORBX_OK;
    if (overlap) {
        if (hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) return ORBX_ERR_DEVICE;
    } else if (!fuse_blur && !on_demand && launch_blur(stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    if (stop_after == ST_BLUR) return ORBX_OK;
    }
describe:
    if (phases & ORBX_PHASE_DESCRIBE) {
        StageScope sc(timer, stream, ST_DESCRIBE);
        if (use_on_demand(b, hg, stop_after)) { if (launch_describe_od(b, hg, stream) != ORBX_OK) return ORBX_ERR_DEVICE; }
        else if (g.fp_contract) hipLaunchKernelGGL(k_describe<true>, dim3(frame_item_blocks(b, (g.nquads + DESC_WAVES - 1) / DESC_WAVES)), dim3(DESC_WAVES * 64), 0, stream, b);
        else hipLaunchKernelGGL(k_describe<false>, dim3(frame_item_blocks(b, (g.nquads + DESC_WAVES - 1) / DESC_WAVES)), dim3(DESC_WAVES * 64), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    return ORBX_OK;
}

// The one-frame drop-in call (orbx_extract): the frame is fetched from the pinned, device-mapped staging buffer by a kernel instead
// of a DMA copy (a copy -> kernel dependency costs ~9 us on top of the copy's 15 us for a VGA frame; 300 waves with one 16-byte
// load each in flight pull the 300 KB over PCIe in less, and the next kernel follows without a queue switch).
__global__ __launch_bounds__(256) void k_ingest(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
int launch_ingest(uint8_t* d_dst, const uint8_t* mapped_src, size_t bytes, hipStream_t stream) {
    const int n16 = (int)(bytes / 16);
    hipLaunchKernelGGL(k_ingest, dim3((n16 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const uint4*>(mapped_src), reinterpret_cast<uint4*>(d_dst), n16);
    ORBX_LAUNCH_CHECK();
    return ORBX_OK;
}

// Fold finished event pairs into the per-stage totals (caller has synchronised the stream).
void stage_timer_collect(StageTimer& t) {
    for (size_t i = 0; i + 1 < t.pool.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.pool[i], t.pool[i + 1]) == hipSuccess) {
            const int st = t.pool_stage[i / 2];
            t.ms[st] += ms;
            t.launches[st] += 1;
        }
    }
    for (hipEvent_t e : t.pool) (void)hipEventDestroy(e);
    t.pool.clear();
    t.pool_stage.clear();
}

// ------------------------------------------------------------------------------------ math probe (diagnostics)
__global__ void k_eval_math(int kind, const float* in0, const float* in1, float* out0, float* out1, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (kind == 0) out0[i] = fast_atan2_deg(in0[i], in1[i]);
    else { float s, c; sincosf_orb(in0[i], &s, &c); out0[i] = s; out1[i] = c; }
}
int launch_eval_math(int kind, const float* in0, const float* in1, float* out0, float* out1, int n) {
    hipLaunchKernelGGL(k_eval_math, dim3((n + 255) / 256), dim3(256), 0, 0, kind, in0, in1, out0, out1, n);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

}  // namespace orbx
