// Pyramid kernels of the ORB extractor for gfx950 (ComputePyramid, reference src/ORBextractor.cc:781-822: cv::resize INTER_LINEAR 8U level after level):
// k_resize (one launch per level: full launch groups) and k_pyramid (cones of levels in one launch: launch groups too small to fill the chip).
#include <algorithm>
#include <type_traits>

#include "orbx_device.h"
#include "orbx_launch.h"

namespace orbx {

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


// (launch_extract's pyramid stage; the per-frame status fill that rides on it stays with the caller)
int launch_pyramid(const Batch& b, const HostGeom& hg, hipStream_t stream) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
    const bool fused_pyramid = g.npyr_groups > 0 && F < PYR_FUSED_MAX_FRAMES;
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
    return ORBX_OK;
}

}  // namespace orbx
