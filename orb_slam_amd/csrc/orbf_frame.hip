// Frame-side steps on gfx950 (include/orbf.h): undistortion of the extractor's keypoints,
// the 64x48 search grid, and the grid window query, all on device-resident data.
//
//   k_undistort_grid   one workgroup per frame.  Lane i: cvUndistortPoints on keypoint i (f64, 5 iterations — ~150
//                      f64 ops), cell = PosInGrid.  The grid is a stable counting sort by cell in LDS: counts by atomics, one
//                      scan over the 3073 offsets, an unordered placement and a rank pass (ascending i inside a cell = the
//                      reference's push_back order).  (Rounds 2-5 bitonic-sorted keys (cell << 13 | i): 55 barrier-separated
//                      passes, 90 us a frame's workgroup; this form: NOTES 11.7.)  HBM traffic: 28 B in, 28 + 4 B out per
//                      keypoint + 12 KB of offsets per frame.
//   k_area<FILL>       one lane per window query: pass 1 counts, a single-block scan turns counts into CSR offsets,
//                      pass 2 writes the indices in the reference's order (cells x-major / y / cell order).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "orbf.h"
#include "orbf_math.h"
#include "orbx.h"

namespace orbf {

constexpr int FR_BLOCK = 256;
constexpr int IDX_BITS = 13;                      // ORBF_MAX_FEATURES = 8192
static_assert((1 << IDX_BITS) == ORBF_MAX_FEATURES, "index bits");

// LDS of k_undistort_grid for a frame capacity: cell of each feature + the unordered placement (u16 each), cell offsets + fill cursors (u32 each)
__host__ __device__ constexpr size_t undistort_grid_lds(int cap) { return ((size_t)cap * 2 * 2 + 15) / 16 * 16 + (size_t)(ORBF_GRID_CELLS + 1) * 4 * 2; }

__global__ __launch_bounds__(FR_BLOCK) void k_undistort_grid(orbf_camera cam, orbf_bounds b, const orbx_keypoint* __restrict__ kps,
                                                            const int32_t* __restrict__ d_n, int n_or_cap,
                                                            orbx_keypoint* __restrict__ kps_un, int32_t* __restrict__ cell_off,
                                                            int32_t* __restrict__ cell_feat) {
    extern __shared__ __align__(16) uint8_t lds[];
    __shared__ int s_part[FR_BLOCK / 64];
    uint16_t* cell_of = reinterpret_cast<uint16_t*>(lds);                         // [cap]  0xFFFF: in no cell
    uint16_t* placed = cell_of + n_or_cap;                                        // [cap]  features grouped by cell, order inside a cell arbitrary
    uint32_t* off = reinterpret_cast<uint32_t*>(lds + ((size_t)n_or_cap * 4 + 15) / 16 * 16);      // [CELLS + 1]
    uint32_t* fill = off + (ORBF_GRID_CELLS + 1);                                 // [CELLS + 1]
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = d_n ? min(d_n[frame], n_or_cap) : n_or_cap;
    kps += (size_t)frame * n_or_cap;
    kps_un += (size_t)frame * n_or_cap;
    cell_feat += (size_t)frame * n_or_cap;
    cell_off += (size_t)frame * (ORBF_GRID_CELLS + 1);
    for (int c = tid; c <= ORBF_GRID_CELLS; c += FR_BLOCK) fill[c] = 0;
    __syncthreads();
    // ---- every feature: undistort, its cell, the cell's count (fill[c + 1])
    const bool distorted = cam.dist[0] != 0.0f;    // src/Frame.cc:291: `if(mDistCoef.at<float>(0)==0.0) mvKeysUn=mvKeys`
    for (int i = tid; i < n; i += FR_BLOCK) {
        orbx_keypoint kp = kps[i];
        if (distorted) undistort_point(cam, kp.x, kp.y, &kp.x, &kp.y);
        kps_un[i] = kp;
        const int c = grid_cell(b, kp.x, kp.y);
        cell_of[i] = (uint16_t)(c >= 0 ? c : 0xFFFF);
        if (c >= 0) atomicAdd(&fill[c + 1], 1u);
    }
    __syncthreads();
    // ---- inclusive scan of fill[0 .. CELLS] (fill[0] = 0): entry c becomes the start of cell c; consecutive entries per thread, wave scan, wave totals
    {
        constexpr int CH = (ORBF_GRID_CELLS + 1 + FR_BLOCK - 1) / FR_BLOCK;
        int sum = 0;
        for (int k = 0; k < CH; ++k) { const int c = tid * CH + k; if (c <= ORBF_GRID_CELLS) sum += (int)fill[c]; }
        int incl = sum;
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        int run = incl - sum;
        for (int w = 0; w < wave; ++w) run += s_part[w];
        for (int k = 0; k < CH; ++k) {
            const int c = tid * CH + k;
            if (c <= ORBF_GRID_CELLS) { run += (int)fill[c]; off[c] = (uint32_t)run; fill[c] = (uint32_t)run; cell_off[c] = run; }
        }
    }
    __syncthreads();
    // ---- group by cell (whoever comes first takes the next slot of its cell) ...
    for (int i = tid; i < n; i += FR_BLOCK) {
        const uint32_t c = cell_of[i];
        if (c != 0xFFFFu) placed[atomicAdd(&fill[c], 1u)] = (uint16_t)i;
    }
    __syncthreads();
    // ... and put every feature at its rank inside the cell: the number of smaller indices there (ascending i inside a cell = the reference's
    // push_back order, src/Frame.cc:116-123).  A cell holds a handful of features; the walk is per feature, so a crowded cell costs its square spread
    // over the workgroup, never one thread's serial sort.
    for (int i = tid; i < n; i += FR_BLOCK) {
        const uint32_t c = cell_of[i];
        if (c == 0xFFFFu) continue;
        const uint32_t s0 = off[c], s1 = off[c + 1];
        uint32_t rank = 0;
        for (uint32_t k = s0; k < s1; ++k) rank += placed[k] < (uint32_t)i;
        cell_feat[s0 + rank] = i;
    }
}

template <bool FILL>
__global__ __launch_bounds__(FR_BLOCK) void k_area(orbf_bounds b, const orbx_keypoint* __restrict__ kps_un, const int32_t* __restrict__ cell_off,
                                                  const int32_t* __restrict__ cell_feat, const float* __restrict__ qxyr,
                                                  const int32_t* __restrict__ qlev, int nq, int32_t* __restrict__ seg, int32_t* __restrict__ cand,
                                                  int cand_cap) {
    const int q = blockIdx.x * FR_BLOCK + threadIdx.x;
    if (q >= nq) return;
    const float x = qxyr[3 * q], y = qxyr[3 * q + 1], r = qxyr[3 * q + 2];
    const int minLevel = qlev[2 * q], maxLevel = qlev[2 * q + 1];
    int x0, x1, y0, y1, cnt = 0;
    int out = FILL ? seg[q] : 0;
    if (window_cells(b, x, y, r, &x0, &x1, &y0, &y1)) {
        for (int ix = x0; ix <= x1; ix++) {
            // the cells (ix, y0..y1) are consecutive in the CSR: one contiguous run per grid column
            const int j0 = cell_off[ix * ORBF_GRID_ROWS + y0], j1 = cell_off[ix * ORBF_GRID_ROWS + y1 + 1];
            for (int j = j0; j < j1; j++) {
                const int f = cell_feat[j];
                const orbx_keypoint kp = kps_un[f];
                if (!in_window(kp.x, kp.y, kp.octave, x, y, r, minLevel, maxLevel)) continue;
                if (FILL) { if (out < cand_cap) cand[out] = f; out++; }
                else cnt++;
            }
        }
    }
    if (!FILL) seg[q + 1] = cnt;
}

// seg[1..nq] hold counts: exclusive scan in place into seg[0..nq]; one block
__global__ __launch_bounds__(1024) void k_scan_counts(int32_t* __restrict__ seg, int nq, int cand_cap, int32_t* __restrict__ status) {
    __shared__ int part[1024];
    const int C = (nq + 1023) / 1024;
    const int lo = threadIdx.x * C, hi = min(lo + C, nq);
    int s = 0;
    for (int i = lo; i < hi; i++) s += seg[i + 1];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < 1024; t++) { const int v = part[t]; part[t] = run; run += v; }
        seg[0] = 0;
        if (status) status[0] = run > cand_cap ? ORBX_ERR_CAPACITY : ORBX_OK;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { run += seg[i + 1]; seg[i + 1] = run; }
}

}  // namespace orbf

extern "C" {

int orbf_image_bounds(const orbf_camera* cam, orbf_bounds* out) {
    if (!cam || !out || cam->ndist < 0 || cam->ndist > 8 || cam->width <= 0 || cam->height <= 0) return ORBX_ERR_ARG;
    if (cam->dist[0] != 0.0f) {      // src/Frame.cc:323-342
        float m[4][2] = {{0.f, 0.f}, {(float)cam->width, 0.f}, {0.f, (float)cam->height}, {(float)cam->width, (float)cam->height}};
        for (int i = 0; i < 4; i++) orbf::undistort_point(*cam, m[i][0], m[i][1], &m[i][0], &m[i][1]);
        out->min_x = (int32_t)fminf(floorf(m[0][0]), floorf(m[2][0]));
        out->max_x = (int32_t)fmaxf(ceilf(m[1][0]), ceilf(m[3][0]));
        out->min_y = (int32_t)fminf(floorf(m[0][1]), floorf(m[1][1]));
        out->max_y = (int32_t)fmaxf(ceilf(m[2][1]), ceilf(m[3][1]));
    } else {
        out->min_x = 0; out->max_x = cam->width; out->min_y = 0; out->max_y = cam->height;
    }
    if (out->max_x <= out->min_x || out->max_y <= out->min_y) return ORBX_ERR_GEOMETRY;
    out->inv_w = (float)ORBF_GRID_COLS / (float)(out->max_x - out->min_x);      // src/Frame.cc:75-76
    out->inv_h = (float)ORBF_GRID_ROWS / (float)(out->max_y - out->min_y);
    return ORBX_OK;
}

int orbf_undistort_grid_batch_device(const orbf_camera* cam, const orbf_bounds* b, const orbx_keypoint* d_kps, const int32_t* d_n,
                                     int nframes, int cap, orbx_keypoint* d_kps_un, int32_t* d_cell_off, int32_t* d_cell_feat, void* stream) {
    if (!cam || !b || nframes < 0 || cap < 1 || cap > ORBF_MAX_FEATURES || cam->ndist < 0 || cam->ndist > 8) return ORBX_ERR_ARG;
    if (nframes == 0) return ORBX_OK;
    if (!d_kps || !d_kps_un || !d_cell_off || !d_cell_feat) return ORBX_ERR_ARG;
    hipLaunchKernelGGL(orbf::k_undistort_grid, dim3(nframes), dim3(orbf::FR_BLOCK), orbf::undistort_grid_lds(cap), (hipStream_t)stream, *cam, *b, d_kps, d_n, cap,
                       d_kps_un, d_cell_off, d_cell_feat);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbf_undistort_grid(const orbf_camera* cam, const orbf_bounds* b, const orbx_keypoint* kps, int n, orbx_keypoint* kps_un,
                        int32_t* cell_off, int32_t* cell_feat, int device) {
    if (!cam || !b || n < 0 || n > ORBF_MAX_FEATURES || !cell_off || (n > 0 && (!kps || !kps_un || !cell_feat))) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    const size_t N = (size_t)std::max(n, 1);
    const size_t o_un = N * 28, o_off = ((2 * N * 28 + 15) & ~(size_t)15), o_feat = o_off + (ORBF_GRID_CELLS + 1) * 4, o_n = o_feat + N * 4,
                 total = o_n + 4;
    uint8_t* d = nullptr;
    int rc = ORBX_ERR_DEVICE;
    const int32_t count = n;
    if (hipMalloc(&d, total) == hipSuccess && (n == 0 || hipMemcpy(d, kps, (size_t)n * 28, hipMemcpyHostToDevice) == hipSuccess) &&
        hipMemcpy(d + o_n, &count, 4, hipMemcpyHostToDevice) == hipSuccess) {
        rc = orbf_undistort_grid_batch_device(cam, b, (const orbx_keypoint*)d, (const int32_t*)(d + o_n), 1, (int)N, (orbx_keypoint*)(d + o_un),
                                              (int32_t*)(d + o_off), (int32_t*)(d + o_feat), nullptr);
        if (rc == ORBX_OK && (hipMemcpy(cell_off, d + o_off, (ORBF_GRID_CELLS + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              (n > 0 && (hipMemcpy(kps_un, d + o_un, (size_t)n * 28, hipMemcpyDeviceToHost) != hipSuccess ||
                                         hipMemcpy(cell_feat, d + o_feat, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess))))
            rc = ORBX_ERR_DEVICE;
    }
    if (d) (void)hipFree(d);
    return rc;
}

int orbf_features_in_area_device(const orbf_bounds* b, const orbx_keypoint* d_kps_un, int n, const int32_t* d_cell_off, const int32_t* d_cell_feat,
                                 const float* d_qxyr, const int32_t* d_qlev, int nq, int32_t* d_seg_off, int32_t* d_cand, int cand_cap,
                                 int32_t* d_status, void* stream) {
    if (!b || n < 0 || nq < 0 || cand_cap < 0 || !d_seg_off) return ORBX_ERR_ARG;
    if (nq > 0 && (!d_kps_un || !d_cell_off || !d_cell_feat || !d_qxyr || !d_qlev || (cand_cap > 0 && !d_cand))) return ORBX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (nq + orbf::FR_BLOCK - 1) / orbf::FR_BLOCK;
    if (nq > 0)
        hipLaunchKernelGGL(orbf::k_area<false>, dim3(blocks), dim3(orbf::FR_BLOCK), 0, st, *b, d_kps_un, d_cell_off, d_cell_feat, d_qxyr, d_qlev, nq,
                           d_seg_off, d_cand, cand_cap);
    hipLaunchKernelGGL(orbf::k_scan_counts, dim3(1), dim3(1024), 0, st, d_seg_off, nq, cand_cap, d_status);
    if (nq > 0)
        hipLaunchKernelGGL(orbf::k_area<true>, dim3(blocks), dim3(orbf::FR_BLOCK), 0, st, *b, d_kps_un, d_cell_off, d_cell_feat, d_qxyr, d_qlev, nq,
                           d_seg_off, d_cand, cand_cap);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbf_features_in_area(const orbf_bounds* b, const orbx_keypoint* kps_un, int n, const int32_t* cell_off, const int32_t* cell_feat,
                          const float* qxyr, const int32_t* qlev, int nq, int32_t* seg_off, int32_t* cand, int cand_cap, int device) {
    if (!b || n < 0 || nq < 0 || cand_cap < 0 || !seg_off || !cell_off) return ORBX_ERR_ARG;
    if (nq > 0 && (!qxyr || !qlev)) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    const size_t N = (size_t)std::max(n, 1), Q = (size_t)std::max(nq, 1), C = (size_t)std::max(cand_cap, 1);
    const size_t o_off = (N * 28 + 15) & ~(size_t)15, o_feat = o_off + (ORBF_GRID_CELLS + 1) * 4, o_q = o_feat + N * 4, o_l = o_q + Q * 12,
                 o_seg = o_l + Q * 8, o_cand = o_seg + (Q + 1) * 4, o_st = o_cand + C * 4, total = o_st + 4;
    uint8_t* d = nullptr;
    int rc = ORBX_ERR_DEVICE;
    if (hipMalloc(&d, total) == hipSuccess && (n == 0 || (hipMemcpy(d, kps_un, (size_t)n * 28, hipMemcpyHostToDevice) == hipSuccess &&
                                                            hipMemcpy(d + o_feat, cell_feat, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess)) &&
        hipMemcpy(d + o_off, cell_off, (ORBF_GRID_CELLS + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
        (nq == 0 || (hipMemcpy(d + o_q, qxyr, (size_t)nq * 12, hipMemcpyHostToDevice) == hipSuccess &&
                     hipMemcpy(d + o_l, qlev, (size_t)nq * 8, hipMemcpyHostToDevice) == hipSuccess))) {
        rc = orbf_features_in_area_device(b, (const orbx_keypoint*)d, n, (const int32_t*)(d + o_off), (const int32_t*)(d + o_feat), (const float*)(d + o_q),
                                          (const int32_t*)(d + o_l), nq, (int32_t*)(d + o_seg), (int32_t*)(d + o_cand), cand_cap, (int32_t*)(d + o_st), nullptr);
        int32_t st = 0;
        if (rc == ORBX_OK && (hipMemcpy(&st, d + o_st, 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(seg_off, d + o_seg, ((size_t)nq + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = ORBX_ERR_DEVICE;
        if (rc == ORBX_OK) {
            const int total_c = seg_off[nq];
            const int ncopy = std::min(total_c, cand_cap);
            if (ncopy > 0 && hipMemcpy(cand, d + o_cand, (size_t)ncopy * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = ORBX_ERR_DEVICE;
            else if (st != ORBX_OK) rc = st;
        }
    }
    if (d) (void)hipFree(d);
    return rc;
}

}  // extern "C"
