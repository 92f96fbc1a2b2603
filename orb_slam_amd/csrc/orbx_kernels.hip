// HIP kernels of the ORB extractor for gfx950 (MI355X, wave64).  One launch group processes a whole
// batch of frames: every kernel's grid spans frames x (levels x tiles | cells | keypoint slots), so
// launch cost is amortised over the batch and the 256 CUs always see >> 256 workgroups.
//
// Stage            reference (under /root/reference/src/ORBextractor.cc)        kernel
//   pyramid        ComputePyramid :781-822 (cv::resize INTER_LINEAR)             k_resize (per level, 7 launches)
//   FAST + NMS     cv::FAST(cell, th, true) :607/:613                            k_fast_nms   (dense score, cell-local NMS)
//   cell lists     raster-ordered keypoints of each cell                         k_compact    (wave ballot/popcount compaction)
//   quotas         :622-670                                                      k_quota      (sequential, one lane per level)
//   retainBest     :683-685 (per cell), :697-701 (per level)                     k_cell_select / k_level_select (libstdc++ introselect)
//   blur           GaussianBlur 7x7 s=2 :760                                     k_blur
//   orientation    IC_Angle :124-151, descriptor :155-194, scaling :769-775      k_describe   (one wave per keypoint)
//
// No 16-px border planes exist on the device: the only out-of-image reads of the reference (blur
// taps <= 3 px, rotated BRIEF taps <= 2 px outside the ROI) are served by reflect-101 index math,
// which is what copyMakeBorder(BORDER_REFLECT_101) materialises.
#include <algorithm>

#include "orb_math.h"
#include "orbx_internal.h"

namespace orbx {

__device__ __constant__ uint32_t c_pattern[256] = {
#include "orb_pattern_packed.inc"
};

__device__ __forceinline__ const uint8_t* plain_plane(const Batch& b, const LevelGeom& L, int level, int frame, long long& stride) {
    if (level == 0) {
        stride = b.img_row_stride;
        return b.img + (long long)frame * b.img_frame_stride;
    }
    stride = L.stride;
    return b.pyr + (long long)frame * b.g->frame_plane_bytes + L.plane_off;
}

template <typename T>
__device__ __forceinline__ int find_level(const DevGeom& g, int idx, T base_of) {
    int level = 0;
    while (level + 1 < g.nlevels && idx >= base_of(g.lv[level + 1])) level++;
    return level;
}

// ------------------------------------------------------------------------------------ pyramid
// cv::resize INTER_LINEAR 8U, level-1 -> level.  Each lane produces 4 horizontally adjacent output
// pixels and stores them as one dword (coalesced 256 B per wave); source taps are byte gathers that
// hit L1/L2 (each source row segment is re-read by the neighbouring output row).
__global__ __launch_bounds__(256) void k_resize(Batch b, int level) {
    const DevGeom& g = *b.g;
    const LevelGeom& L = g.lv[level];
    const LevelGeom& P = g.lv[level - 1];
    const int frame = blockIdx.z;
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dy >= L.h || dx0 >= L.w) return;
    long long sstride;
    const uint8_t* src = plain_plane(b, P, level - 1, frame, sstride);
    const ResizeY ry = b.taby[L.taby_off + dy];
    const uint8_t* r0 = src + ry.sy0 * sstride;
    const uint8_t* r1 = src + ry.sy1 * sstride;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = dx0 + k;
        if (dx < L.w) {
            const ResizeX rx = b.tabx[L.tabx_off + dx];
            const int px = resize_px(r0[rx.sx], r0[rx.sx1], r1[rx.sx], r1[rx.sx1], rx.a0, rx.a1, ry.b0, ry.b1);
            packed |= (uint32_t)(px & 255) << (8 * k);
        }
    }
    uint8_t* dst = b.pyr + (long long)frame * g.frame_plane_bytes + L.plane_off + (long long)dy * L.stride;
    *reinterpret_cast<uint32_t*>(dst + dx0) = packed;   // stride is a multiple of 64: in-bounds and aligned
}

// ------------------------------------------------------------------------------------ FAST + NMS
// One workgroup = one 64x32 tile of a level's scan area.  The image tile (+4 halo) is staged in LDS,
// every pixel of the tile +1 halo gets its exact FAST-9 score (fast9_score: two rounds of 3-input
// min/max give the 9-arc extrema), then the 3x3 strict NMS runs from LDS.  Neighbours that belong to
// a different grid cell count as 0, exactly as when cv::FAST is called per cell view.
constexpr int FI_W = TILE_W + 8, FI_H = TILE_H + 8;   // image tile
constexpr int FS_W = TILE_W + 2, FS_H = TILE_H + 2;   // score tile
constexpr int FI_S = FI_W;                            // LDS row stride of the image tile (72 B)
constexpr int FS_S = FS_W + 2;                        // 68 B

__global__ __launch_bounds__(256) void k_fast_nms(Batch b) {
    __shared__ uint8_t s_img[FI_H * FI_S];
    __shared__ uint8_t s_sc[FS_H * FS_S];
    const DevGeom& g = *b.g;
    const int frame = blockIdx.x / g.ntiles_total;
    const int t = blockIdx.x - frame * g.ntiles_total;
    const int level = find_level(g, t, [](const LevelGeom& l) { return l.tile_base; });
    const LevelGeom& L = g.lv[level];
    const int tl = t - L.tile_base;
    const int ty = tl / L.tiles_x, tx = tl - ty * L.tiles_x;
    const int x0 = EDGE + tx * TILE_W, y0 = EDGE + ty * TILE_H;
    const int tid = threadIdx.x;
    long long stride;
    const uint8_t* src = plain_plane(b, L, level, frame, stride);

    for (int i = tid; i < FI_H * FI_W; i += 256) {
        const int ly = i / FI_W, lx = i - ly * FI_W;
        const int gx = min(x0 - 4 + lx, L.w - 1), gy = min(y0 - 4 + ly, L.h - 1);
        s_img[ly * FI_S + lx] = src[gy * stride + gx];
    }
    __syncthreads();

    const int tmin = g.tmin;
    for (int p = tid; p < FS_H * FS_W; p += 256) {
        const int sy = p / FS_W, sx = p - sy * FS_W;
        const int gx = x0 - 1 + sx, gy = y0 - 1 + sy;
        const bool in_scan = gx >= EDGE && gx < L.w - EDGE && gy >= EDGE && gy < L.h - EDGE;
        const uint8_t* c = &s_img[(sy + 3) * FI_S + sx + 3];
        const int v = c[0];
        int d[16];
        d[0] = v - c[3 * FI_S + 0];   d[1] = v - c[3 * FI_S + 1];   d[2] = v - c[2 * FI_S + 2];   d[3] = v - c[1 * FI_S + 3];
        d[4] = v - c[3];              d[5] = v - c[-1 * FI_S + 3];  d[6] = v - c[-2 * FI_S + 2];  d[7] = v - c[-3 * FI_S + 1];
        d[8] = v - c[-3 * FI_S + 0];  d[9] = v - c[-3 * FI_S - 1];  d[10] = v - c[-2 * FI_S - 2]; d[11] = v - c[-1 * FI_S - 3];
        d[12] = v - c[-3];            d[13] = v - c[1 * FI_S - 3];  d[14] = v - c[2 * FI_S - 2];  d[15] = v - c[3 * FI_S - 1];
        // cheap necessary condition, wave-uniform skip over flat regions
        int mx = imax3(d[0], d[1], d[2]), mn = imin3(d[0], d[1], d[2]);
#pragma unroll
        for (int k = 3; k < 15; k += 2) { mx = imax3(mx, d[k], d[k + 1]); mn = imin3(mn, d[k], d[k + 1]); }
        mx = imax(mx, d[15]); mn = imin(mn, d[15]);
        const bool maybe = in_scan && (mx > tmin || mn < -tmin);
        int score = 0;
        if (__any(maybe)) score = maybe ? fast9_score(d, tmin) : 0;
        s_sc[sy * FS_S + sx] = (uint8_t)score;
    }
    __syncthreads();

    uint8_t* nms = b.nms + (long long)frame * g.frame_plane_bytes + L.plane_off;
    const uint8_t* fxs = b.flagx + L.flag_off_x;
    const uint8_t* fys = b.flagy + L.flag_off_y;
    const int lx = tid & 63;
    const int gx = x0 + lx;
    if (gx >= L.w - EDGE) return;
    const int fx = fxs[gx];
#pragma unroll
    for (int k = 0; k < TILE_H / 4; k++) {
        const int ly = (tid >> 6) + 4 * k;
        const int gy = y0 + ly;
        if (gy >= L.h - EDGE) break;
        const uint8_t* sc = &s_sc[(ly + 1) * FS_S + lx + 1];
        const int s = sc[0];
        int keep = 0;
        if (s) {
            const int fy = fys[gy];
            const bool hasL = !(fx & 1), hasR = !(fx & 2), hasU = !(fy & 1), hasD = !(fy & 2);
            int m = 0;   // max over valid neighbours
            if (hasL) m = imax(m, sc[-1]);
            if (hasR) m = imax(m, sc[1]);
            if (hasU) {
                m = imax(m, sc[-FS_S]);
                if (hasL) m = imax(m, sc[-FS_S - 1]);
                if (hasR) m = imax(m, sc[-FS_S + 1]);
            }
            if (hasD) {
                m = imax(m, sc[FS_S]);
                if (hasL) m = imax(m, sc[FS_S - 1]);
                if (hasR) m = imax(m, sc[FS_S + 1]);
            }
            keep = s > m ? s : 0;
        }
        nms[(long long)gy * L.stride + gx] = (uint8_t)keep;
    }
}

// ------------------------------------------------------------------------------------ cell lists
// One wave per (frame, cell): walk the cell's scan rectangle in raster order, 64 pixels per step;
// __ballot + popcount give each survivor its rank, so the list comes out in cv::FAST's raster order.
__global__ __launch_bounds__(64) void k_compact(Batch b) {
    const DevGeom& g = *b.g;
    const int frame = blockIdx.x / g.ncells_total;
    const int cell = blockIdx.x - frame * g.ncells_total;
    const int level = find_level(g, cell, [](const LevelGeom& l) { return l.cell_base; });
    const LevelGeom& L = g.lv[level];
    const CellGeom c = b.cells[cell];
    const int lane = threadIdx.x;
    Cand* out = b.cand + (long long)frame * g.frame_cands + L.cand_base + c.cand_off;
    const uint8_t* nms = b.nms + (long long)frame * g.frame_plane_bytes + L.plane_off;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int n = 0, nhi = 0, nlo = 0;
    for (int y = c.y0; y <= c.y1; y++) {
        const uint8_t* row = nms + (long long)y * L.stride;
        for (int xb = c.x0; xb <= c.x1; xb += 64) {
            const int x = xb + lane;
            const int s = (x <= c.x1) ? row[x] : 0;
            const unsigned long long m = __ballot(s != 0);
            if (m == 0) continue;
            if (s) {
                Cand e;
                e.pos = (uint32_t)x | ((uint32_t)y << 16);
                e.resp = (float)s;
                out[n + __popcll(m & lt)] = e;
            }
            n += __popcll(m);
            nhi += __popcll(__ballot(s >= g.fast_th && s != 0));
            nlo += __popcll(__ballot(s >= 7));
        }
    }
    if (lane == 0) {
        CellState st;
        st.n_all = n; st.n_hi = nhi; st.n_lo = nlo;
        b.cstate[(long long)frame * g.ncells_total + cell] = st;
    }
}

// ------------------------------------------------------------------------------------ quotas
// reference :609-670, one lane per (frame, level); sequential by definition.
__global__ __launch_bounds__(64) void k_quota(Batch b) {
    const DevGeom& g = *b.g;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= b.nframes * g.nlevels) return;
    const int frame = idx / g.nlevels, level = idx - frame * g.nlevels;
    const LevelGeom& L = g.lv[level];
    const CellGeom* cg = b.cells + L.cell_base;
    const CellState* cs = b.cstate + (long long)frame * g.ncells_total + L.cell_base;
    CellSel* sel = b.csel + (long long)frame * g.ncells_total + L.cell_base;
    const int nCells = L.ncells, nfc = L.nfeat_cell;
    int nToDistribute = 0, nNoMore = 0;
    // csel.out_off doubles as the bNoMore flag during this pass
    for (int c = 0; c < nCells; c++) {
        CellSel s;
        if (cg[c].skipped) {
            s.thr = g.fast_th; s.nkeys = 0; s.nretain = 0; s.out_off = 0;
        } else {
            const CellState st = cs[c];
            const bool fallback = st.n_hi <= 3;               // :609  size()<=3 -> FAST(...,7,...)
            s.thr = fallback ? 7 : g.fast_th;
            s.nkeys = fallback ? st.n_lo : st.n_hi;
            if (s.nkeys > nfc) { s.nretain = nfc; s.out_off = 0; }
            else { s.nretain = s.nkeys; nToDistribute += nfc - s.nkeys; s.out_off = 1; nNoMore++; }
        }
        sel[c] = s;
    }
    while (nToDistribute > 0 && nNoMore < nCells) {
        const int nNew = nfc + (int)ceilf((float)nToDistribute / (float)(nCells - nNoMore));
        nToDistribute = 0;
        for (int c = 0; c < nCells; c++) {
            CellSel s = sel[c];
            if (!s.out_off) {
                if (s.nkeys > nNew) { s.nretain = nNew; }
                else { s.nretain = s.nkeys; nToDistribute += nNew - s.nkeys; s.out_off = 1; nNoMore++; }
                sel[c] = s;
            }
        }
    }
    int off = 0;
    for (int c = 0; c < nCells; c++) {
        sel[c].out_off = off;
        off += sel[c].nretain;
    }
    if (off > L.sel_cap) { b.status[frame] = ORBX_ERR_CAPACITY; off = 0; for (int c = 0; c < nCells; c++) { sel[c].nretain = 0; sel[c].out_off = 0; } }
    b.level_total[frame * MAX_LEVELS + level] = off;
}

struct RespGreater {   // KeypointResponseGreater (OpenCV keypoint.cpp)
    __host__ __device__ constexpr bool operator()(const Cand& a, const Cand& b) const { return a.resp > b.resp; }
};

// reference :79-120 (HarrisResponses, blockSize 7) on the unblurred level; x,y = level coords of the corner
__device__ float harris_response(const uint8_t* img, long long step, int x, int y) {
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
    return ((float)a * (float)bb - (float)c * (float)c - 0.04f * ((float)a + (float)bb) * ((float)a + (float)bb)) * scale_sq_sq;
}

// ------------------------------------------------------------------------------------ retainBest per cell
// KeyPointsFilter::retainBest(keysCell, n) followed by resize(n) keeps exactly the first n elements
// that std::nth_element leaves in front (the std::partition of boundary ties is truncated away again by
// the resize).  Which tied keypoints survive, and their ORDER, is libstdc++'s introselect;
// std::nth_element is constexpr in C++20, so the very same library code is compiled for the device.
__global__ __launch_bounds__(64) void k_cell_select(Batch b) {
    const DevGeom& g = *b.g;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= b.nframes * g.ncells_total) return;
    const int frame = idx / g.ncells_total, cell = idx - frame * g.ncells_total;
    const int level = find_level(g, cell, [](const LevelGeom& l) { return l.cell_base; });
    const LevelGeom& L = g.lv[level];
    const CellGeom cgeo = b.cells[cell];
    const CellSel s = b.csel[(long long)frame * g.ncells_total + cell];
    if (s.nretain <= 0) return;
    const int n_all = b.cstate[(long long)frame * g.ncells_total + cell].n_all;
    Cand* c = b.cand + (long long)frame * g.frame_cands + L.cand_base + cgeo.cand_off;
    int m = 0;
    const float thr = (float)s.thr;
    for (int i = 0; i < n_all; i++) {
        const Cand e = c[i];
        if (e.resp >= thr) c[m++] = e;
    }
    if (g.score_type == ORBX_HARRIS_SCORE) {
        long long stride;
        const uint8_t* img = plain_plane(b, L, level, frame, stride);
        for (int i = 0; i < m; i++) c[i].resp = harris_response(img, stride, c[i].pos & 0xFFFF, c[i].pos >> 16);
    }
    if (m > s.nretain) std::nth_element(c, c + s.nretain, c + m, RespGreater());
    Cand* out = b.sel + (long long)frame * g.frame_sel + L.sel_base + s.out_off;
    const int keep = min(m, s.nretain);
    for (int i = 0; i < keep; i++) out[i] = c[i];
}

// reference :697-701 (per-level cap)
__global__ __launch_bounds__(64) void k_level_select(Batch b) {
    const DevGeom& g = *b.g;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= b.nframes * g.nlevels) return;
    const int frame = idx / g.nlevels, level = idx - frame * g.nlevels;
    const LevelGeom& L = g.lv[level];
    const int total = b.level_total[frame * MAX_LEVELS + level];
    int n = total;
    if (total > L.ndesired) {
        n = L.ndesired;
        if (n > 0) {
            Cand* v = b.sel + (long long)frame * g.frame_sel + L.sel_base;
            std::nth_element(v, v + n, v + total, RespGreater());
        }
    }
    b.level_count[frame * MAX_LEVELS + level] = n;
}

// ------------------------------------------------------------------------------------ blur
// GaussianBlur 7x7 sigma 2 (8U fixed point, taps [18,34,49,55,49,34,18]/256 twice, 16 fractional bits).
// Separable inside one workgroup: image tile (+3 halo, reflect-101 at the image edge) -> LDS, row sums
// -> LDS, column pass -> global.  Reads the UNBLURRED plane, writes a separate blurred plane, which is
// what the reference's in-place filter computes (its border taps read the unblurred reflect border).
constexpr int BI_W = TILE_W + 6, BI_H = TILE_H + 6;
constexpr int BI_S = TILE_W + 8;   // 72

__global__ __launch_bounds__(256) void k_blur(Batch b) {
    __shared__ uint8_t s_in[BI_H * BI_S];
    __shared__ uint32_t s_row[BI_H * TILE_W];
    const DevGeom& g = *b.g;
    const int frame = blockIdx.x / g.nbtiles_total;
    const int t = blockIdx.x - frame * g.nbtiles_total;
    const int level = find_level(g, t, [](const LevelGeom& l) { return l.btile_base; });
    const LevelGeom& L = g.lv[level];
    const int tl = t - L.btile_base;
    const int ty = tl / L.btiles_x, tx = tl - ty * L.btiles_x;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;
    const int tid = threadIdx.x;
    long long stride;
    const uint8_t* src = plain_plane(b, L, level, frame, stride);
    for (int i = tid; i < BI_H * BI_W; i += 256) {
        const int ly = i / BI_W, lx = i - ly * BI_W;
        const int gx = reflect101(x0 - 3 + lx, L.w), gy = reflect101(y0 - 3 + ly, L.h);
        s_in[ly * BI_S + lx] = src[gy * stride + gx];
    }
    __syncthreads();
    for (int p = tid; p < BI_H * TILE_W; p += 256) {
        const int r = p >> 6, x = p & 63;
        const uint8_t* q = &s_in[r * BI_S + x];
        s_row[p] = (uint32_t)blur_taps7(q[0], q[1], q[2], q[3], q[4], q[5], q[6]);
    }
    __syncthreads();
    const int lx = tid & 63, gx = x0 + lx;
    if (gx >= L.w) return;
    uint8_t* dst = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    const int te = gx < L.blur_wvec;
#pragma unroll
    for (int k = 0; k < TILE_H / 4; k++) {
        const int ly = (tid >> 6) + 4 * k, gy = y0 + ly;
        if (gy >= L.h) break;
        const uint32_t* q = &s_row[ly * TILE_W + lx];
        const int sum = blur_taps7(q[0], q[TILE_W], q[2 * TILE_W], q[3 * TILE_W], q[4 * TILE_W], q[5 * TILE_W], q[6 * TILE_W]);
        dst[(long long)gy * L.stride + gx] = (uint8_t)blur_round(sum, te);
    }
}

// ------------------------------------------------------------------------------------ orientation + rBRIEF + output
// One wave per output keypoint.  IC_Angle: 2 patch rows per step, lanes over u; wave reduction of the
// integer moments.  Descriptor: lane i evaluates tests i, i+64, i+128, i+192; each __ballot is 8
// descriptor bytes (test t is bit t%8 of byte t/8, LSB first — the reference's packing).
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_describe(Batch b) {
    const DevGeom& g = *b.g;
    const int frame = blockIdx.y;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int32_t* counts = b.level_count + frame * MAX_LEVELS;
    if (slot == 0 && lane == 0) {
        int total = 0;
        for (int l = 0; l < g.nlevels; l++) total += counts[l];
        int st = b.status[frame];
        if (total > b.cap) { st = ORBX_ERR_CAPACITY; total = 0; }
        b.out_n[frame] = st == ORBX_OK ? total : 0;
        if (b.out_status) b.out_status[frame] = st;
    }
    if (slot >= g.nslots) return;
    const int level = find_level(g, slot, [](const LevelGeom& l) { return l.slot_base; });
    const LevelGeom& L = g.lv[level];
    const int k = slot - L.slot_base;
    if (k >= counts[level]) return;
    int out_idx = k, total = 0;
    for (int l = 0; l < g.nlevels; l++) { if (l < level) out_idx += counts[l]; total += counts[l]; }
    if (total > b.cap || b.status[frame] != ORBX_OK) return;

    const Cand kp = b.sel[(long long)frame * g.frame_sel + L.sel_base + k];
    const int x = kp.pos & 0xFFFF, y = kp.pos >> 16;
    long long pstride;
    const uint8_t* plain = plain_plane(b, L, level, frame, pstride);

    // IC_Angle on the unblurred level (:705-706 run before the blur)
    int m10 = 0, m01 = 0;
    {
        const int u = (lane & 31) - HALF_PATCH;
        const int au = u < 0 ? -u : u;
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int r = it * 2 + (lane >> 5);
            const int v = r - HALF_PATCH;
            const int av = v < 0 ? -v : v;
            if (r <= 2 * HALF_PATCH && au <= HALF_PATCH && au <= g.umax[av]) {
                const int I = plain[(long long)(y + v) * pstride + x + u];
                m10 += u * I;
                m01 += v * I;
            }
        }
        m10 = wave_sum(m10);
        m01 = wave_sum(m01);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // rotated BRIEF on the blurred level (:154-194)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    sincosf_orb(angle * factorPI, &sn, &cs);
    const uint8_t* blur = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    unsigned long long words[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t pat = c_pattern[r * 64 + lane];
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float px = (float)(int)(int8_t)(pat >> (16 * e));
            const float py = (float)(int)(int8_t)(pat >> (16 * e + 8));
            const int iy = cv_round_f(px * sn + py * cs);
            const int ix = cv_round_f(px * cs - py * sn);
            const int X = x + ix, Y = y + iy;
            if ((unsigned)X < (unsigned)L.w && (unsigned)Y < (unsigned)L.h)
                val[e] = blur[(long long)Y * L.stride + X];
            else   // the reference reads the level's unblurred reflect-101 border here
                val[e] = plain[(long long)reflect101(Y, L.h) * pstride + reflect101(X, L.w)];
        }
        words[r] = __ballot(val[0] < val[1]);
    }
    if (lane < 4) {
        unsigned long long w = lane == 0 ? words[0] : lane == 1 ? words[1] : lane == 2 ? words[2] : words[3];
        uint8_t* d = b.out_desc + ((long long)frame * b.cap + out_idx) * 32 + lane * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = (uint8_t)(w >> (8 * i));
    }
    if (lane == 0) {
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

int launch_extract(const Batch& b, const HostGeom& hg, hipStream_t stream, int stop_after, StageTimer* timer) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
    if (F <= 0) return ORBX_OK;
    if (hipMemsetAsync(b.status, 0, sizeof(int32_t) * F, stream) != hipSuccess) return ORBX_ERR_DEVICE;
    {
        StageScope sc(timer, stream, ST_PYRAMID);
        for (int l = 1; l < g.nlevels; l++) {
            const LevelGeom& L = g.lv[l];
            dim3 grid((L.w + 255) / 256, (L.h + 3) / 4, F);
            hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, stream, b, l);
            ORBX_LAUNCH_CHECK();
        }
    }
    if (stop_after == ST_PYRAMID) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_FAST_NMS);
        hipLaunchKernelGGL(k_fast_nms, dim3(F * g.ntiles_total), dim3(256), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_FAST_NMS) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_COMPACT);
        hipLaunchKernelGGL(k_compact, dim3(F * g.ncells_total), dim3(64), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_COMPACT) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_QUOTA);
        hipLaunchKernelGGL(k_quota, dim3((F * g.nlevels + 63) / 64), dim3(64), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_QUOTA) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_CELL_SELECT);
        hipLaunchKernelGGL(k_cell_select, dim3((F * g.ncells_total + 63) / 64), dim3(64), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_CELL_SELECT) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_LEVEL_SELECT);
        hipLaunchKernelGGL(k_level_select, dim3((F * g.nlevels + 63) / 64), dim3(64), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_LEVEL_SELECT) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_BLUR);
        hipLaunchKernelGGL(k_blur, dim3(F * g.nbtiles_total), dim3(256), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_BLUR) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_DESCRIBE);
        hipLaunchKernelGGL(k_describe, dim3((g.nslots + 3) / 4, F), dim3(256), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
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
