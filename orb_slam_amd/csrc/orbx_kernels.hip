// HIP kernels of the ORB extractor for gfx950 (MI355X, wave64).  One launch group processes a whole
// batch of frames: every kernel's grid spans frames x (levels x tiles | cells | keypoint slots), so
// launch cost is amortised over the batch and the 256 CUs always see >> 256 workgroups.
//
// Stage            reference (under /root/reference/src/ORBextractor.cc)        kernel
//   pyramid        ComputePyramid :781-822 (cv::resize INTER_LINEAR)             k_resize (per level, 7 launches)
//   FAST + NMS     cv::FAST(cell, th, true) :607/:613 + raster-ordered cell lists  k_fast_cells (one workgroup per grid cell)
//   quotas         :622-670                                                      k_quota      (one wave per level; lane 0 runs the sequential rule)
//   retainBest     :683-685 (per cell), :697-701 (per level)                     k_cell_select / k_level_select (wave-parallel, permutation-exact introselect)
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
    return b.pyr + (long long)frame * b.g.frame_plane_bytes + L.plane_off;
}

// threadIdx.x >> 6 is wave-uniform but the compiler cannot know it: pin it in an SGPR so that everything derived
// from it (task -> level -> LevelGeom fields) is fetched with scalar loads instead of per-lane vector loads.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// level of a flat task index: number of levels l >= 1 whose base is <= idx.  `bases` is a compact kernarg array
// (INT_MAX beyond nlevels), so all compares are independent: one scalar-load round trip, no dependent chain.
__device__ __forceinline__ int find_level(const int (&bases)[MAX_LEVELS], int idx) {
    int level = 0;
#pragma unroll
    for (int l = 1; l < MAX_LEVELS; l++) level += idx >= bases[l] ? 1 : 0;
    return level;
}

constexpr unsigned long long UMAX_NIBBLES = 0x3689ABCDDEEEFFFFull;   // umax[v] for v = 0..15 (15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3)

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}


// p -> (p / cw, p % cw) for p < 65536, cw <= 2048 with full-rate VALU ops only (v_mul_lo/hi_u32 are quarter
// rate): q = trunc((p + 0.5) * (1/cw)), exact for every (p, cw) in that range (exhaustively checked offline).
__device__ __forceinline__ void split_px(int p, int cw, float inv_cw, int& y, int& x) {
    y = (int)(((float)p + 0.5f) * inv_cw);
    x = p - (int)__umul24((unsigned)y, (unsigned)cw);
}

// ------------------------------------------------------------------------------------ pyramid
// cv::resize INTER_LINEAR 8U, level-1 -> level.  A workgroup produces a 256x4 output tile: the source
// rectangle it needs (<= 7 rows x ~310 px) is staged in LDS with coalesced dword loads, each lane then
// reads its 16 taps as LDS bytes, produces 4 horizontally adjacent output pixels and stores one dword.
// (Byte gathers straight from global memory made this kernel texture-addresser bound.)
// LDS source tile: L.rz_rows x L.rz_pitch bytes, the exact maximum over the level's tiles (7 KB at scale 1.2; a fixed
// worst-case array for scale 2.5 was 31 KB and capped the kernel at 5 workgroups per CU)

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_resize(Batch b, int level) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_src[];
    const DevGeom& g = b.g;
    const LevelGeom& L = g.lv[level];
    const int RZ_SRC_W = L.rz_pitch;
    const LevelGeom& P = g.lv[level - 1];
    const int frame = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int bx0 = blockIdx.x * 256, by0 = blockIdx.y * RZ_ROWS;
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
    // source rectangle of this tile (tables are monotone)
    const int r0 = ty[by0].sy0, r1 = ty[by1].sy1;
    const int c0 = tx[bx0].sx & ~3, c1 = tx[bx1].sx1;
    const int nd = ((c1 - c0) >> 2) + 1, nr = r1 - r0 + 1;
    {
        // flattened (row, dword) items, 8 independent loads in flight per lane (a row-per-iteration loop serialises
        // ~6 dependent global-load round trips per workgroup and made this kernel latency-bound)
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

// ------------------------------------------------------------------------------------ FAST + NMS + cell lists
// One workgroup = one grid cell of one level of one frame (or one row band of a big cell) — the unit the reference
// calls cv::FAST on (src/ORBextractor.cc:599-614).  Because the NMS of cv::FAST never looks outside the cell view, a
// cell-native workgroup needs no score halo towards other cells, no survivor plane in HBM and no compaction pass:
//   1. stage the cell's pixels (+3 halo) in LDS with pipelined dword loads;
//   2. rounds of 2048 pixels —
//      A1: every lane tests 4 pixels against the 4 compass ring pixels (two of them, with one polarity, must be beyond the
//          threshold: exact necessary condition); survivors are queued in LDS by __ballot/popcount;
//      A2: dense over that queue, OpenCV's opposite-pair pre-test on the raw ring bytes; survivors queued again;
//      B : dense over those, the exact FAST-9 score (9-arc extrema of the RAW ring bytes by two rounds of
//          v_min3/v_max3: dark = v - min_arcs(max9), bright = max_arcs(min9) - v);
//      N : 3x3 strict NMS, dense over the PREVIOUS round's scored pixels (all their neighbours are scored by then);
//          survivors of the band's own rows set their bit in an LDS bitmask;
//   3. a wave-level scan over the bitmask popcounts gives list offsets and the keypoint list comes out in cv::FAST's
//      raster order together with its counts at fastTh and at 7.
constexpr int FAST_MAX_ROUNDS = 32;   // cells hold < 65536 pixels (checked on the host)

typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2v as_us2v(uint32_t v) { return __builtin_bit_cast(us2v, v); }

struct FastLds {
    int n1[FAST_MAX_ROUNDS];   // per round: pixels that passed the compass test
    int n2[FAST_MAX_ROUNDS];   // per round: pixels that passed the opposite-pair test (get an exact score)
    int n_hi, n_lo, n_all, pad;
};

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
    int min_hi9 = 255, max_lo9 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        min_hi9 = imin(min_hi9, imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
        max_lo9 = imax(max_lo9, imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
    }
    const int s = imax(v - min_hi9, max_lo9 - v) - 1;   // == OpenCV cornerScore for every corner
    return s >= tmin ? s : 0;
}

// wave-aggregated append of the lanes with `pass` to an LDS queue; returns nothing, order inside the queue is irrelevant
__device__ __forceinline__ void queue_push(uint16_t* q, int* counter, int pass, int value, int lane, unsigned long long lt) {
    const unsigned long long m = __ballot(pass);
    if (m) {
        int qb = 0;
        if (lane == 0) qb = atomicAdd(counter, __popcll(m));
        qb = __shfl(qb, 0, 64);
        if (pass) q[qb + __popcll(m & lt)] = (uint16_t)value;
    }
}

// FAST_THREADS: 512 (8 waves: 1080p-class grids, ~40 KB LDS per work item, 4 items per CU) or 256 (VGA-class grids: smaller bands,
// ~28 KB, 5 items per CU — more independent latency chains in flight)
template <bool ALIGNED, int FAST_THREADS, int FAST_PPT, int FAST_SLACK>
__device__ __forceinline__ void fast_cell_task(const Batch& b, int task, uint8_t* smem) {
    constexpr int FAST_ROUND = FAST_THREADS * FAST_PPT;   // pixels per round
    constexpr int FAST_QCAP = FAST_ROUND + FAST_SLACK;
    const DevGeom& g = b.g;
    const int frame = task / g.nbands_total;
    const int item = task - frame * g.nbands_total;
    const BandGeom bg = b.bands[item];
    const int level = bg.level;
    const LevelGeom& L = g.lv[level];
    // the band scores rows ey0..ey1 (its own rows plus one halo row towards neighbouring bands of the same cell) and
    // emits survivors of its own rows y0..y1 only; below, "cell" coordinates are relative to (x0, ey0)
    struct { int x0, y0, x1, y1; } cg = {bg.x0, bg.ey0, bg.x1, bg.ey1};
    const int own_lo = bg.y0 - bg.ey0, own_hi = bg.y1 - bg.ey0;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int cw = cg.x1 - cg.x0 + 1, ch = cg.y1 - cg.y0 + 1;
    CellState* cst = b.cstate + (long long)frame * g.nbands_total + item;
    if (cw <= 0 || ch <= 0) {
        if (tid == 0) { CellState st; st.n_all = 0; st.n_hi = 0; st.n_lo = 0; *cst = st; }
        return;
    }
    const int npx = cw * ch;
    const int nchunks = (npx + 63) >> 6;
    // LDS carve (all offsets multiples of 16): header | survivor bit masks | chunk offsets | queues | scores | image
    FastLds* hdr = reinterpret_cast<FastLds*>(smem);
    unsigned long long* cmask = reinterpret_cast<unsigned long long*>(smem + sizeof(FastLds));
    int* coffs = reinterpret_cast<int*>(smem + sizeof(FastLds) + g.fast_max_chunks * 8);
    uint16_t* q1 = reinterpret_cast<uint16_t*>(smem + sizeof(FastLds) + g.fast_max_chunks * 12);
    uint16_t* q2 = q1 + FAST_QCAP;                  // two buffers: the NMS of batch i-1 runs after batch i has been scored
    uint8_t* s_sc = reinterpret_cast<uint8_t*>(q2 + 2 * FAST_QCAP);
    uint8_t* s_img = s_sc + g.fast_max_px;
    // image region: rows y0-3..y1+3, columns from the dword-aligned start at or left of x0-3
    const int gxb = (cg.x0 - 3) & ~3;
    const int xoff = (cg.x0 - 3) - gxb;
    const int nd = (xoff + cw + 6 + 3) >> 2;
    const int S = nd * 4;
    long long stride;
    const uint8_t* src = plain_plane(b, L, level, frame, stride);
    {
        // flattened (row, dword) items, 8 independent loads in flight per lane
        const int total = (ch + 6) * nd;
        const float inv_nd = 1.0f / (float)nd;
        const uint8_t* src0 = src + (long long)(cg.y0 - 3) * stride + gxb;
        const int xm = L.w - 1 - gxb;   // unaligned path: never read past the row end
        for (int i0 = 0; i0 < ((b.dbg & 8) ? 0 : total); i0 += FAST_THREADS * 8) {
            uint32_t v4[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * FAST_THREADS + tid;
                v4[k] = 0;
                if (i < total) {
                    int r, d;
                    split_px(i, nd, inv_nd, r, d);
                    const uint8_t* row = src0 + (long long)r * stride;
                    if (ALIGNED) v4[k] = *reinterpret_cast<const uint32_t*>(row + 4 * d);
                    else v4[k] = (uint32_t)row[imin(4 * d, xm)] | (uint32_t)row[imin(4 * d + 1, xm)] << 8 | (uint32_t)row[imin(4 * d + 2, xm)] << 16 |
                                 (uint32_t)row[imin(4 * d + 3, xm)] << 24;
                }
            }
            if (i0 == 0) {
                // the clears of the other LDS regions ride on the latency of the loads just issued
                for (int i = tid; i < (g.fast_max_px >> 4); i += FAST_THREADS) reinterpret_cast<uint4*>(s_sc)[i] = make_uint4(0, 0, 0, 0);
                for (int i = tid; i < (int)(sizeof(FastLds) / 4); i += FAST_THREADS) reinterpret_cast<int*>(hdr)[i] = 0;
                for (int i = tid; i < nchunks; i += FAST_THREADS) cmask[i] = 0ull;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * FAST_THREADS + tid;
                if (i < total) reinterpret_cast<uint32_t*>(s_img)[i] = v4[k];   // row r, dword d  ==  r*nd + d  (S = 4*nd)
            }
        }
        if (b.dbg & 8) {   // development switch "no image staging": the clears still have to happen
            for (int i = tid; i < (g.fast_max_px >> 4); i += FAST_THREADS) reinterpret_cast<uint4*>(s_sc)[i] = make_uint4(0, 0, 0, 0);
            for (int i = tid; i < (int)(sizeof(FastLds) / 4); i += FAST_THREADS) reinterpret_cast<int*>(hdr)[i] = 0;
            for (int i = tid; i < nchunks; i += FAST_THREADS) cmask[i] = 0ull;
        }
    }
    __syncthreads();

    const float inv_cw = 1.0f / (float)cw;
    const int tmin = g.tmin;
    const uint8_t* img0 = s_img + 3 * S + xoff + 3;         // pixel (0,0) of the cell
    const unsigned long long lt = (1ull << lane) - 1ull;

    // 3x3 strict NMS of the scored pixels of one finished round (dense over its queue); survivors set their bit
    auto nms_round = [&](const uint16_t* q, int n) {
        for (int i = tid; i < n; i += FAST_THREADS) {
            const int p = q[i];
            const int s = s_sc[p];
            if (s) {
                int y, x;
                split_px(p, cw, inv_cw, y, x);
                const uint8_t* sp = s_sc + p;
                // neighbours outside the cell count as 0 (cv::FAST on the cell view): offsets are clamped to stay inside
                // the score array and the values masked, so the 8 LDS loads are independent and branch-free
                const int dl = x > 0 ? -1 : 0, dr = x < cw - 1 ? 1 : 0, du = y > 0 ? -cw : 0, dd = y < ch - 1 ? cw : 0;
                const int nl = sp[dl], nr = sp[dr], nu = sp[du], nd_ = sp[dd];
                const int nul = sp[du + dl], nur = sp[du + dr], ndl = sp[dd + dl], ndr = sp[dd + dr];
                int mx = imax3(dl ? nl : 0, dr ? nr : 0, du ? nu : 0);
                mx = imax3(mx, dd ? nd_ : 0, (du && dl) ? nul : 0);
                mx = imax3(mx, (du && dr) ? nur : 0, (dd && dl) ? ndl : 0);
                mx = imax(mx, (dd && dr) ? ndr : 0);
                if (s > mx && y >= own_lo && y <= own_hi) {
                    atomicOr(&cmask[p >> 6], 1ull << (p & 63));
                    if (s >= g.fast_th) atomicAdd(&hdr->n_hi, 1);   // survivors are rare: LDS atomics beat a wave reduction
                    if (s >= 7) atomicAdd(&hdr->n_lo, 1);
                }
            }
        }
    };

    // Rounds of 2048 pixels run A1 and append to q1; the later phases run per BATCH of rounds, flushed once q1 holds more
    // than FAST_QCAP - FAST_ROUND entries (or at the end): on ordinary images one or two flushes per cell instead of one
    // per round, i.e. fewer barriers and full waves in A2 / B / N.  Batches cover whole rounds (> 1 pixel row each), so the
    // neighbours of a batch's pixels lie in the previous, the same or the next batch: N lags by one batch.
    int batch = 0, rnd = 0, qfill = 0;   // qfill: entries in q1 (block-uniform; every lane adds the final per-round counts)
    const int npx_scan = (b.dbg & 4) ? imin(npx, 1) : npx;
    for (int base = 0; base < npx_scan; base += FAST_ROUND, rnd++) {
        uint16_t* q2cur = q2 + (batch & 1) * FAST_QCAP;
        // A1: compass test on every pixel
        {
            int pass[FAST_PPT];
            unsigned long long pm[FAST_PPT];
            int cnt = 0;
            // all 4 pixels' loads are issued before any test (addresses clamped into the cell, results masked)
            // Any 9 contiguous ring positions contain at least TWO of the 4 compass positions (0,4,8,12), so a corner needs
            // two compass pixels beyond the threshold with the same polarity: the 2nd smallest must be < v - t or the 2nd
            // largest > v + t.  (With only ">= 1 compass pixel" 42 % of the S-blocks pixels passed — every pixel within
            // 3 px of an edge; the pair rule rejects straight axis-aligned edges: 4x fewer pixels reach the pair test.)
            int vv[FAST_PPT], s2[FAST_PPT], s3[FAST_PPT];
#pragma unroll
            for (int k = 0; k < FAST_PPT; k++) {
                const int p = imin(base + k * FAST_THREADS + tid, npx - 1);
                int y, x;
                split_px(p, cw, inv_cw, y, x);
                const uint8_t* c = img0 + __umul24((unsigned)y, (unsigned)S) + x;
                const int x0 = c[3 * S], x4 = c[3], x8 = c[-3 * S], x12 = c[-3];
                vv[k] = c[0];
                const int lo1 = imin(x0, x4), hi1 = imax(x0, x4), lo2 = imin(x8, x12), hi2 = imax(x8, x12);
                const int a = imax(lo1, lo2), bq = imin(hi1, hi2);
                s2[k] = imin(a, bq);     // 2nd smallest of the four
                s3[k] = imax(a, bq);     // 2nd largest
            }
#pragma unroll
            for (int k = 0; k < FAST_PPT; k++) {
                const int p = base + k * FAST_THREADS + tid;
                pass[k] = p < npx && ((vv[k] - s2[k] > tmin) | (s3[k] - vv[k] > tmin));
                pm[k] = __ballot(pass[k]);
                cnt += __popcll(pm[k]);
            }
            if (cnt) {
                int qb = 0;
                if (lane == 0) qb = qfill + atomicAdd(&hdr->n1[rnd], cnt);   // per-round counter: final once the barrier is passed
                qb = __shfl(qb, 0, 64);
#pragma unroll
                for (int k = 0; k < FAST_PPT; k++) {
                    if (pass[k]) q1[qb + __popcll(pm[k] & lt)] = (uint16_t)(base + k * FAST_THREADS + tid);
                    qb += __popcll(pm[k]);
                }
            }
        }
        __syncthreads();
        // A2: opposite-pair test, dense over the compass survivors.  n1 / n2 are block-uniform after the barriers, so
        // rounds without candidates (flat image regions) skip the remaining phases and their barriers altogether.
        qfill += hdr->n1[rnd];
        const bool last_round = base + FAST_ROUND >= npx_scan;
        if (qfill <= FAST_QCAP - FAST_ROUND && !last_round) continue;   // keep filling q1 (block-uniform decision)
        const int n1 = (b.dbg & 1) ? 0 : qfill;
        qfill = 0;
        if (n1 > 0) {
            for (int i0 = 0; i0 < n1; i0 += FAST_THREADS) {
                const int i = i0 + tid;
                int pass = 0, p = 0;
                if (i < n1) {
                    p = q1[i];
                    int y, x;
                    split_px(p, cw, inv_cw, y, x);
                    const uint8_t* c = img0 + __umul24((unsigned)y, (unsigned)S) + x;
                    pass = fast_pair_test(c, S, c[0], tmin);
                }
                queue_push(q2cur, &hdr->n2[batch], pass, p, lane, lt);
            }
            __syncthreads();
            // B: exact FAST score, dense over the pair-test survivors
            const int n2 = (b.dbg & 2) ? 0 : hdr->n2[batch];
            if (n2 > 0) {
                for (int i = tid; i < n2; i += FAST_THREADS) {
                    const int p = q2cur[i];
                    int y, x;
                    split_px(p, cw, inv_cw, y, x);
                    const uint8_t* c = img0 + __umul24((unsigned)y, (unsigned)S) + x;
                    s_sc[p] = (uint8_t)fast_score_raw(c, S, c[0], tmin);
                }
                __syncthreads();
            }
        }
        // N: every neighbour of the previous batch's pixels is scored now
        if (batch > 0) nms_round(q2 + ((batch - 1) & 1) * FAST_QCAP, hdr->n2[batch - 1]);
        batch++;
    }
    if (b.dbg & 16) return;
    if (batch > 0) nms_round(q2 + ((batch - 1) & 1) * FAST_QCAP, hdr->n2[batch - 1]);
    __syncthreads();
    // The cell's keypoint list in raster order (cv::FAST's order): one LANE per 64-pixel chunk of the survivor bitmask.
    // Block-wide exclusive scan of the chunk popcounts (wave scan + per-wave totals), then every lane walks the few set bits
    // of its own chunk.  (A wave-per-chunk loop spent most of its time on empty chunks: 18 dependent LDS reads per wave.)
    int* wsum = coffs;                                  // reuse: FAST_THREADS / 64 wave totals
    unsigned long long m = 0ull;
    if (tid < nchunks) m = cmask[tid];
    int cnt = __popcll(m), incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - cnt, total = 0;
#pragma unroll
    for (int wv = 0; wv < FAST_THREADS / 64; wv++) { const int t = wsum[wv]; if (wv < wave) run += t; total += t; }
    Cand* out = b.cand + (long long)frame * g.frame_cands + L.cand_base + bg.cand_off;
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int p = tid * 64 + bit;
        int y, x;
        split_px(p, cw, inv_cw, y, x);
        Cand e;
        e.pos = (uint32_t)(cg.x0 + x) | ((uint32_t)(cg.y0 + y) << 16);
        e.resp = (float)s_sc[p];
        out[run++] = e;
    }
    if (tid == 0) {
        CellState st;
        st.n_all = total; st.n_hi = hdr->n_hi; st.n_lo = hdr->n_lo;
        *cst = st;
    }
}

// One workgroup per (frame, cell).  (A persistent variant — 4 workgroups per CU walking the cells with a static stride —
// measured 35 % slower: the hardware dispatcher balances the very uneven cell sizes better than a static schedule.)
template <bool ALIGNED, int FAST_THREADS, int FAST_PPT, int FAST_SLACK>
__global__ __launch_bounds__(FAST_THREADS) void k_fast_cells(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    fast_cell_task<ALIGNED, FAST_THREADS, FAST_PPT, FAST_SLACK>(b, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------ quotas
// reference :609-670.  One wave per (frame, level): the lanes gather the cells' counts in parallel into LDS, lane 0
// runs the reference's (inherently sequential) redistribution loop on the LDS copy, the lanes write the result back.
constexpr int QUOTA_MAX_CELLS = 1024;   // checked on the host

__global__ __launch_bounds__(64) void k_quota(Batch b) {
    __shared__ int s_nkeys[QUOTA_MAX_CELLS], s_nret[QUOTA_MAX_CELLS], s_off[QUOTA_MAX_CELLS];
    __shared__ uint8_t s_thr[QUOTA_MAX_CELLS], s_done[QUOTA_MAX_CELLS], s_skip[QUOTA_MAX_CELLS];
    __shared__ int s_total;
    const DevGeom& g = b.g;
    const int frame = blockIdx.x / g.nlevels, level = blockIdx.x - frame * g.nlevels;
    const int lane = threadIdx.x;
    const LevelGeom& L = g.lv[level];
    const CellGeom* cg = b.cells + L.cell_base;
    const CellState* cs = b.cstate + (long long)frame * g.nbands_total;   // per band; a cell sums its bands
    CellSel* sel = b.csel + (long long)frame * g.ncells_total + L.cell_base;
    const int nCells = L.ncells, nfc = L.nfeat_cell;
    for (int c = lane; c < nCells; c += 64) {
        const CellGeom cgc = cg[c];
        int n_hi = 0, n_lo = 0;
        for (int k = 0; k < cgc.nbands; k++) { const CellState t = cs[cgc.band0 + k]; n_hi += t.n_hi; n_lo += t.n_lo; }
        const bool fallback = n_hi <= 3;                      // :609  size()<=3 -> FAST(...,7,...)
        s_skip[c] = (uint8_t)cgc.skipped;
        s_thr[c] = (uint8_t)(cgc.skipped || !fallback ? g.fast_th : 7);
        s_nkeys[c] = cgc.skipped ? 0 : (fallback ? n_lo : n_hi);
    }
    __syncthreads();
    if (lane == 0) {
        int nToDistribute = 0, nNoMore = 0;
        for (int c = 0; c < nCells; c++) {
            if (s_skip[c]) { s_nret[c] = 0; s_done[c] = 0; continue; }   // reference `continue`: never reaches the bookkeeping
            const int nk = s_nkeys[c];
            if (nk > nfc) { s_nret[c] = nfc; s_done[c] = 0; }
            else { s_nret[c] = nk; nToDistribute += nfc - nk; s_done[c] = 1; nNoMore++; }
        }
        while (nToDistribute > 0 && nNoMore < nCells) {
            const int nNew = nfc + (int)ceilf((float)nToDistribute / (float)(nCells - nNoMore));
            nToDistribute = 0;
            for (int c = 0; c < nCells; c++) {
                if (!s_done[c]) {
                    const int nk = s_nkeys[c];
                    if (nk > nNew) { s_nret[c] = nNew; }
                    else { s_nret[c] = nk; nToDistribute += nNew - nk; s_done[c] = 1; nNoMore++; }
                }
            }
        }
        int off = 0;
        for (int c = 0; c < nCells; c++) { s_off[c] = off; off += s_nret[c]; }
        if (off > L.sel_cap) { b.status[frame] = ORBX_ERR_CAPACITY; off = -1; }
        s_total = off;
        b.level_total[frame * MAX_LEVELS + level] = off < 0 ? 0 : off;
    }
    __syncthreads();
    const bool bad = s_total < 0;
    for (int c = lane; c < nCells; c += 64) {
        CellSel r;
        r.thr = s_thr[c]; r.nkeys = s_nkeys[c];
        r.nretain = bad ? 0 : s_nret[c];
        r.out_off = bad ? 0 : s_off[c];
        sel[c] = r;
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

__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

__device__ void wave_nth_element(Cand* a, int first, int nth, int last, uint16_t* lpos, uint16_t* rpos, int lane) {
    if (first == last || nth == last) return;
    int depth = 2 * (31 - __clz(last - first));   // std::__lg(n) * 2
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (last - first > 3) {
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
        // left stoppers (ascending positions): !(value > P)
        int nL = 0;
        for (int base = f; base < l; base += 64) {
            const int p = base + lane;
            const bool st = p < l && !(a[p].resp > P);
            const unsigned long long m = __ballot(st);
            if (st) lpos[nL + __popcll(m & lt)] = (uint16_t)p;
            nL += __popcll(m);
        }
        // right stoppers (descending positions): !(P > value)
        int nR = 0;
        for (int top = l; top > f; top -= 64) {
            const int p = top - 1 - lane;
            const bool st = p >= f && !(P > a[p].resp);
            const unsigned long long m = __ballot(st);
            if (st) rpos[nR + __popcll(m & lt)] = (uint16_t)p;
            nR += __popcll(m);
        }
        wave_lds_fence();
        // S = number of leading k with L[k] < R[k]  (L ascending, R descending: a prefix)
        const int nmin = nL < nR ? nL : nR;
        int S = 0;
        for (int kb = 0; kb < nmin; kb += 64) {
            const int k = kb + lane;
            const bool ok = k < nmin && lpos[k] < rpos[k];
            const unsigned long long m = __ballot(ok);
            const int c = __popcll(m);
            S += c;
            if (c < 64) break;
        }
        for (int kb = 0; kb < S; kb += 64) {
            const int k = kb + lane;
            if (k < S) {
                const int pl = lpos[k], pr = rpos[k];
                const Cand t = a[pl];
                a[pl] = a[pr];
                a[pr] = t;
            }
        }
        int cut;
        if (S < nL) { cut = lpos[S]; if (S > 0 && (int)rpos[S - 1] < cut) cut = rpos[S - 1]; }
        else cut = rpos[S - 1];
        wave_lds_fence();
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0) std::__insertion_sort(a + first, a + last, __gnu_cxx::__ops::__iter_comp_iter(RespGreater()));
    wave_lds_fence();
}

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

// One wave per (frame, cell): ordered __ballot filter of the cell's list at its threshold into LDS, Harris responses
// in parallel when selected, wave_nth_element, first nToRetain entries out.
// Launched twice: lists of up to SEL_SMALL entries with a small LDS footprint (many waves per CU — the common case),
// longer ones with the full staging area; each launch skips the cells of the other class.
constexpr int SEL_SMALL = 384;
__global__ __launch_bounds__(64) void k_cell_select(Batch b, int lds_entries, int min_entries) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DevGeom& g = b.g;
    const int frame = blockIdx.x / g.ncells_total, cell = blockIdx.x - frame * g.ncells_total;
    const int level = find_level(g.cell_bases, cell);
    const LevelGeom& L = g.lv[level];
    const CellGeom cgeo = b.cells[cell];
    const CellSel s = b.csel[(long long)frame * g.ncells_total + cell];
    if (s.nretain <= 0) return;
    const int lane = threadIdx.x;
    // the cell's list = its bands' sub-lists in band (= raster) order
    const CellState* bst = b.cstate + (long long)frame * g.nbands_total + cgeo.band0;
    const BandGeom* bgs = b.bands + cgeo.band0;
    Cand* lbase = b.cand + (long long)frame * g.frame_cands + L.cand_base;
    Cand* c = lbase + cgeo.cand_off;
    int n_all = 0;
    for (int k = 0; k < cgeo.nbands; k++) n_all += bst[k].n_all;
    if (n_all < min_entries || (n_all > lds_entries && lds_entries < g.sel_lds_entries)) return;   // the other launch's class
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
                for (int i = 0; i < m; i++) c[i].resp = harris_response(img, stride, c[i].pos & 0xFFFF, c[i].pos >> 16);
            if (m > s.nretain) std::nth_element(c, c + s.nretain, c + m, RespGreater());
            const int keep = min(m, s.nretain);
            for (int i = 0; i < keep; i++) out[i] = c[i];
        }
        return;
    }
    Cand* lst = reinterpret_cast<Cand*>(smem);
    uint16_t* lpos = reinterpret_cast<uint16_t*>(smem + (size_t)lds_entries * sizeof(Cand));
    uint16_t* rpos = lpos + lds_entries;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int m = 0;
    for (int k = 0; k < cgeo.nbands; k++) {
        const Cand* bc = lbase + bgs[k].cand_off;
        const int nb = bst[k].n_all;
        for (int base = 0; base < nb; base += 64) {
            const int i = base + lane;
            Cand e;
            e.pos = 0; e.resp = -1.f;
            if (i < nb) e = bc[i];
            const bool pass = i < nb && e.resp >= thr;
            const unsigned long long mk = __ballot(pass);
            if (pass) lst[m + __popcll(mk & lt)] = e;
            m += __popcll(mk);
        }
    }
    wave_lds_fence();
    if (g.score_type == ORBX_HARRIS_SCORE) {
        for (int i = lane; i < m; i += 64) lst[i].resp = harris_response(img, stride, lst[i].pos & 0xFFFF, lst[i].pos >> 16);
        wave_lds_fence();
    }
    if (m > s.nretain) wave_nth_element(lst, 0, s.nretain, m, lpos, rpos, lane);
    const int keep = min(m, s.nretain);
    for (int i = lane; i < keep; i += 64) out[i] = lst[i];
}

// reference :697-701 (per-level cap), same scheme
__global__ __launch_bounds__(64) void k_level_select(Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const DevGeom& g = b.g;
    const int frame = blockIdx.x / g.nlevels, level = blockIdx.x - frame * g.nlevels;
    const LevelGeom& L = g.lv[level];
    const int lane = threadIdx.x;
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
                for (int i = lane; i < total; i += 64) lst[i] = v[i];
                wave_lds_fence();
                wave_nth_element(lst, 0, n, total, lpos, rpos, lane);
                for (int i = lane; i < n; i += 64) v[i] = lst[i];
            }
        }
    }
    if (lane == 0) b.level_count[frame * MAX_LEVELS + level] = n;
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
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_debug_nth), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_debug_nth, dim3(1), dim3(64), lds, 0, d_resp, n, nth, d_out);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

// ------------------------------------------------------------------------------------ blur
// GaussianBlur 7x7 sigma 2 (8U fixed point, taps [18,34,49,55,49,34,18]/256 twice, 16 fractional bits).
// Register-resident separable filter, no LDS: one wave owns a 248-px wide column strip and streams down
// BLUR_ROWS output rows.  Lane j holds one dword (4 pixels) of the current row; the neighbouring dwords
// come from lanes j-1 / j+1 by DPP wave shifts, v_alignbyte_b32 cuts the 4-byte tap windows and two
// v_dot4_u32_u8 give a pixel's 7-tap row sum.  The last 7 row sums live in registers (loop fully unrolled),
// so the column pass is 7 multiply-adds per pixel; each lane stores its 4 output pixels as one dword.
// Reads the UNBLURRED plane and writes a separate blurred plane, which is what the reference's in-place
// filter computes (its border taps read the unblurred reflect-101 border; here: reflect-101 index math).
constexpr int BLUR_STRIP_DW = 62;   // useful dwords per wave (lanes 1..62; lanes 0 and 63 are halo)

__device__ __forceinline__ uint32_t load_px4_reflect(const uint8_t* row, int x, int w) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v |= (uint32_t)row[reflect101(x + i, w)] << (8 * i);
    return v;
}

template <bool ALIGNED>
__global__ __launch_bounds__(BLUR_WAVES * 64) void k_blur(Batch b) {
    const DevGeom& g = b.g;
    const int task_all = blockIdx.x * BLUR_WAVES + wave_id();
    const int frame = task_all / g.nbtiles_total;
    if (frame >= b.nframes) return;
    const int t = task_all - frame * g.nbtiles_total;
    const int level = find_level(g.btile_bases, t);
    const LevelGeom& L = g.lv[level];
    const int tl = t - L.btile_base;
    const int band = tl / L.btiles_x, strip = tl - band * L.btiles_x;
    const int lane = threadIdx.x & 63;
    const int x = (strip * BLUR_STRIP_DW + lane - 1) * 4;      // first pixel of this lane's dword (may be < 0)
    const int y0 = band * BLUR_ROWS;
    const int w = L.w, h = L.h;
    long long stride;
    const uint8_t* src = plain_plane(b, L, level, frame, stride);
    uint8_t* dst = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    const bool fetch = x >= -4 && x < w + 4;                   // halo lanes beyond that are never consumed
    const int xq = (w - 1) & ~3;                               // first pixel of the last (possibly partial) dword of a row
    const int xl = x < 0 ? 0 : (x > xq ? xq : x);
    const bool is_left = x == -4, is_last = x == xq, is_halo = x == xq + 4;
    const bool writer = lane >= 1 && lane <= BLUR_STRIP_DW && x < w;
    const int te = x < L.blur_wvec;                            // blur_wvec is a multiple of 4
    const uint32_t WA = 0x37312212u;                           // taps 18,34,49,55 (little-endian bytes)
    const uint32_t WB = 0x00122231u;                           // taps 49,34,18,0
    uint32_t pp[6][4];            // pp[r % 6] = (row r-1 | row r << 16) of the lane's 4 pixels
    uint32_t prev[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < BLUR_ROWS + 6; r++) {
        const int yy = reflect101(y0 + r - 3, h);
        const uint8_t* row = src + (long long)yy * stride;
        uint32_t C = 0;
        if (ALIGNED) {
            // every lane loads an aligned dword (x clamped into the row); the three kinds of border lanes then build
            // their reflect-101 bytes from dwords already in the wave (DPP + v_perm): no divergent byte-load path
            const uint32_t Craw = *reinterpret_cast<const uint32_t*>(row + xl);
            const uint32_t L1 = __builtin_amdgcn_update_dpp(0u, Craw, 0x138, 0xf, 0xf, false);   // lane-1
            const uint32_t L2 = __builtin_amdgcn_update_dpp(0u, L1, 0x138, 0xf, 0xf, false);     // lane-2
            C = Craw;
            if (is_left) C = __builtin_amdgcn_perm(Craw, Craw, 0x01020300u);       // px -3..-1 <- px 3,2,1 of dword 0
            if (is_last) C = __builtin_amdgcn_perm(Craw, L1, (uint32_t)L.blur_sel_last);   // (D_last, D_prev)
            if (is_halo) C = __builtin_amdgcn_perm(Craw, L2, (uint32_t)L.blur_sel_halo);   // Craw == D_last (clamped load)
        } else if (fetch) {
            C = load_px4_reflect(row, x, w);
        }
        const uint32_t Lw = __builtin_amdgcn_update_dpp(0u, C, 0x138, 0xf, 0xf, false);   // wave_shr:1  <- lane-1
        const uint32_t R = __builtin_amdgcn_update_dpp(0u, C, 0x130, 0xf, 0xf, false);    // wave_shl:1  <- lane+1
        const uint32_t wa0 = __builtin_amdgcn_alignbyte(C, Lw, 1), wa1 = __builtin_amdgcn_alignbyte(C, Lw, 2),
                       wa2 = __builtin_amdgcn_alignbyte(C, Lw, 3), wa3 = C;
        const uint32_t wb0 = __builtin_amdgcn_alignbyte(R, C, 1), wb1 = __builtin_amdgcn_alignbyte(R, C, 2),
                       wb2 = __builtin_amdgcn_alignbyte(R, C, 3), wb3 = R;
        uint32_t cur[4];
        cur[0] = __builtin_amdgcn_udot4(wa0, WA, __builtin_amdgcn_udot4(wb0, WB, 0u, false), false);
        cur[1] = __builtin_amdgcn_udot4(wa1, WA, __builtin_amdgcn_udot4(wb1, WB, 0u, false), false);
        cur[2] = __builtin_amdgcn_udot4(wa2, WA, __builtin_amdgcn_udot4(wb2, WB, 0u, false), false);
        cur[3] = __builtin_amdgcn_udot4(wa3, WA, __builtin_amdgcn_udot4(wb3, WB, 0u, false), false);
        // Vertical pass.  A row sum is at most 255 * 257 = 65535, so two consecutive rows of one pixel fit one dword and
        // v_dot2_u32_u16 takes two taps per instruction: with pair(r) = (row r-1 | row r << 16) the output of rows r-6 .. r is
        //   dot2(pair(r-5), 18|34) + dot2(pair(r-3), 49|55) + dot2(pair(r-1), 49|34) + 18 * row r        (4 ops + 1 pack per pixel)
        if (r >= 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) pp[r % 6][i] = prev[i] | (cur[i] << 16);
        }
        if (r >= 6) {
            const int oy = y0 + r - 6;
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t sum = cur[i] * (uint32_t)ORBX_G0;
                sum = __builtin_amdgcn_udot2(as_us2v(pp[(r - 5) % 6][i]), as_us2v((uint32_t)ORBX_G0 | ((uint32_t)ORBX_G1 << 16)), sum, false);
                sum = __builtin_amdgcn_udot2(as_us2v(pp[(r - 3) % 6][i]), as_us2v((uint32_t)ORBX_G2 | ((uint32_t)ORBX_G3 << 16)), sum, false);
                sum = __builtin_amdgcn_udot2(as_us2v(pp[(r - 1) % 6][i]), as_us2v((uint32_t)ORBX_G2 | ((uint32_t)ORBX_G1 << 16)), sum, false);
                packed |= (uint32_t)blur_round((int)sum, te) << (8 * i);
            }
            if (writer && oy < h) *reinterpret_cast<uint32_t*>(dst + (long long)oy * L.stride + x) = packed;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) prev[i] = cur[i];
    }
}

// ------------------------------------------------------------------------------------ orientation + rBRIEF + output
// One wave per output keypoint.  IC_Angle: 2 patch rows per step, lanes over u; wave reduction of the
// integer moments.  Descriptor: lane i evaluates tests i, i+64, i+128, i+192; each __ballot is 8
// descriptor bytes (test t is bit t%8 of byte t/8, LSB first — the reference's packing).
__global__ __launch_bounds__(DESC_WAVES * 64) void k_describe(Batch b) {
    const DevGeom& g = b.g;
    const int frame = blockIdx.y;
    const int slot = blockIdx.x * DESC_WAVES + wave_id();
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
    const int level = find_level(g.slot_bases, slot);
    const LevelGeom& L = g.lv[level];
    const int k = slot - L.slot_base;
    if (k >= counts[level]) return;
    int out_idx = k, total = 0;
    for (int l = 0; l < g.nlevels; l++) { if (l < level) out_idx += counts[l]; total += counts[l]; }
    if (total > b.cap || b.status[frame] != ORBX_OK) return;

    const Cand kp = b.sel[(long long)frame * g.frame_sel + L.sel_base + k];
    const int x = kp.pos & 0xFFFF, y = kp.pos >> 16;
    long long pstride64;
    const uint8_t* plain = plain_plane(b, L, level, frame, pstride64);
    const unsigned pstride = (unsigned)pstride64;   // rows < 2^24 bytes, planes < 2^31 bytes (host-checked): 24-bit multiplies

    // the 4 BRIEF tests of this lane (independent of everything below: issue the loads now)
    uint32_t pat[4];
#pragma unroll
    for (int r = 0; r < 4; r++) pat[r] = c_pattern[r * 64 + lane];

    // IC_Angle on the unblurred level (:705-706 run before the blur).  umax[] (reference :495-510) depends only on
    // HALF_PATCH_SIZE = 15, so it is a constant: nibble v of UMAX_NIBBLES (the host checks it against the computed table).
    int m10 = 0, m01 = 0;
    {
        // The 31 x 31 box is cut into 31 rows x 8 dwords (u = -15 .. 16; the 32nd byte is masked off): 248 slots, 4 per lane.
        // Per slot ONE (unaligned) dword load, the circle mask as a byte mask, and two v_dot4_u32_u8: sum of (u + 15) * I and
        // sum of I — m10 = sum((u + 15) I) - 15 sum(I), m01 = sum(v * rowsum(I)).  The masks and weights depend only on the lane.
        int a_su = 0, a_si = 0, a_v = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sl = lane + 64 * i;
            const int r = sl >> 3, c = sl & 7;                   // row 0..30 (31 = idle), dword column
            const int v = r - HALF_PATCH, av = v < 0 ? -v : v;
            const int um = r < 31 ? (int)((UMAX_NIBBLES >> (4 * (av & 15))) & 15ull) : -1;
            uint32_t mask = 0, uw = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int u = 4 * c + k - HALF_PATCH;
                const int au = u < 0 ? -u : u;
                if (au <= um) mask |= 0xFFu << (8 * k);
                uw |= (uint32_t)(4 * c + k) << (8 * k);          // u + 15
            }
            const int rr = r < 31 ? r : 30;                       // idle slots re-read the last row (mask = 0)
            uint32_t I;
            __builtin_memcpy(&I, plain + (unsigned)(x + 4 * c - HALF_PATCH) + __umul24((unsigned)(y - HALF_PATCH + rr), pstride), 4);
            I &= mask;
            const int si = (int)__builtin_amdgcn_udot4(I, 0x01010101u, 0u, false);
            a_su = (int)__builtin_amdgcn_udot4(I, uw, (uint32_t)a_su, false);
            a_si += si;
            a_v += v * si;
        }
        m10 = wave_sum(a_su - HALF_PATCH * a_si);
        m01 = wave_sum(a_v);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // rotated BRIEF on the blurred level (:154-194)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    sincosf_orb(angle * factorPI, &sn, &cs);
    const uint8_t* blur = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    unsigned long long words[4];
    // rounded pattern offsets never exceed 18 px (|(-13,-13)| = 18.4): keypoints at least 19 px from every edge — all
    // but the outermost ring of candidates — take a branch-free path; the test is wave-uniform (scalar branch)
    const bool interior = x >= 19 && y >= 19 && x < L.w - 19 && y < L.h - 19;
    if (interior) {
        const uint8_t* ctr = blur + __umul24((unsigned)y, (unsigned)L.stride) + (unsigned)x;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int val[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float px = (float)(int)(int8_t)(pat[r] >> (16 * e));
                const float py = (float)(int)(int8_t)(pat[r] >> (16 * e + 8));
                const int iy = cv_round_f(px * sn + py * cs);
                const int ix = cv_round_f(px * cs - py * sn);
                val[e] = ctr[__mul24(iy, L.stride) + ix];
            }
            words[r] = __ballot(val[0] < val[1]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int val[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float px = (float)(int)(int8_t)(pat[r] >> (16 * e));
                const float py = (float)(int)(int8_t)(pat[r] >> (16 * e + 8));
                const int iy = cv_round_f(px * sn + py * cs);
                const int ix = cv_round_f(px * cs - py * sn);
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
            words[r] = __ballot(val[0] < val[1]);
        }
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

int launch_extract(const Batch& b, const HostGeom& hg, hipStream_t stream, int stop_after, StageTimer* timer, const SideStream* side) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
    if (F <= 0) return ORBX_OK;
    if (hipMemsetAsync(b.status, 0, sizeof(int32_t) * F, stream) != hipSuccess) return ORBX_ERR_DEVICE;
    {
        StageScope sc(timer, stream, ST_PYRAMID);
        for (int l = 1; l < g.nlevels; l++) {
            const LevelGeom& L = g.lv[l];
            dim3 grid((L.w + 255) / 256, (L.h + RZ_ROWS - 1) / RZ_ROWS, F);
            const bool al = l > 1 || (((uintptr_t)b.img | (uintptr_t)b.img_row_stride | (uintptr_t)b.img_frame_stride) & 3) == 0;
            const size_t lds = (size_t)L.rz_pitch * L.rz_rows;
            if (al) hipLaunchKernelGGL(k_resize<true>, grid, dim3(256), lds, stream, b, l);
            else hipLaunchKernelGGL(k_resize<false>, grid, dim3(256), lds, stream, b, l);
            ORBX_LAUNCH_CHECK();
        }
    }
    if (stop_after == ST_PYRAMID) return ORBX_OK;
    auto launch_blur = [&](hipStream_t st) -> int {
        StageScope sc(timer, st, ST_BLUR);
        const bool aligned = (((uintptr_t)b.img | (uintptr_t)b.img_row_stride | (uintptr_t)b.img_frame_stride) & 3) == 0;
        const int nblk = (F * g.nbtiles_total + BLUR_WAVES - 1) / BLUR_WAVES;
        if (aligned) hipLaunchKernelGGL(k_blur<true>, dim3(nblk), dim3(BLUR_WAVES * 64), 0, st, b);
        else hipLaunchKernelGGL(k_blur<false>, dim3(nblk), dim3(BLUR_WAVES * 64), 0, st, b);
        ORBX_LAUNCH_CHECK();
        return ORBX_OK;
    };
    const bool overlap = side && side->aux && stop_after < 0;
    {
        StageScope sc(timer, stream, ST_FAST_CELLS);
        const bool aligned = (((uintptr_t)b.img | (uintptr_t)b.img_row_stride | (uintptr_t)b.img_frame_stride) & 3) == 0;
        const size_t lds = (size_t)g.fast_lds_bytes;
        auto launch = [&](auto kern, int threads) {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(F * g.nbands_total), dim3(threads), lds, stream, b);
        };
        constexpr FastShape A = FAST_SMALL, B = FAST_LARGE;
        if (g.fast_threads == A.threads) {
            if (aligned) launch(k_fast_cells<true, A.threads, A.ppt, A.slack>, A.threads); else launch(k_fast_cells<false, A.threads, A.ppt, A.slack>, A.threads);
        } else {
            if (aligned) launch(k_fast_cells<true, B.threads, B.ppt, B.slack>, B.threads); else launch(k_fast_cells<false, B.threads, B.ppt, B.slack>, B.threads);
        }
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
        hipLaunchKernelGGL(k_quota, dim3(F * g.nlevels), dim3(64), 0, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_QUOTA) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_CELL_SELECT);
        const size_t lds = (size_t)g.sel_lds_cell;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cell_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int small = std::min(SEL_SMALL, g.sel_lds_entries);
        hipLaunchKernelGGL(k_cell_select, dim3(F * g.ncells_total), dim3(64), (size_t)small * (sizeof(Cand) + 4) + 16, stream, b, small, 0);
        ORBX_LAUNCH_CHECK();
        if (small < g.sel_lds_entries) hipLaunchKernelGGL(k_cell_select, dim3(F * g.ncells_total), dim3(64), lds, stream, b, g.sel_lds_entries, small + 1);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_CELL_SELECT) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_LEVEL_SELECT);
        const size_t lds = (size_t)g.sel_lds_level;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_level_select, dim3(F * g.nlevels), dim3(64), lds, stream, b);
        ORBX_LAUNCH_CHECK();
    }
    if (stop_after == ST_LEVEL_SELECT) return ORBX_OK;
    if (overlap) {
        if (hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) return ORBX_ERR_DEVICE;
    } else if (launch_blur(stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    if (stop_after == ST_BLUR) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_DESCRIBE);
        hipLaunchKernelGGL(k_describe, dim3((g.nslots + DESC_WAVES - 1) / DESC_WAVES, F), dim3(DESC_WAVES * 64), 0, stream, b);
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
