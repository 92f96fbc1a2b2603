// Brute-force 256-bit Hamming matching for gfx950: the inner loop of every ORBmatcher search
// (reference src/ORBmatcher.cc:201-222 and siblings: DescriptorDistance + running best/second-best
// with strict '<') lifted to a dense N x M kernel.
//
// Mapping (integer/bitwise work, no MFMA): a lane owns QPL query descriptors in VGPRs (8 dwords
// each).  The train descriptor of the current step is the same for the whole wave, so it is read
// with wave-uniform (scalar, SGPR) loads and costs no VGPRs, no LDS and no per-lane memory traffic.
// Per pair: 8 v_xor + 8 v_bcnt_u32_b32 (accumulating popcount) + 3 ops for the top-2 update.
//
// Top-2 with the reference's tie rules as ONE associative reduction: key = (distance << 22) | index.
// Keys are unique, min(key) is the smallest distance at its FIRST index, and the second-smallest
// key carries the second-smallest distance counted with multiplicity — exactly what the
// `if(d<best){best2=best;best=d;idx=i}else if(d<best2)best2=d` scan produces.  Because it is a
// plain two-smallest reduction, the train set can be split across workgroups and merged in any order.
#include <climits>
#include <cstring>

#include "orb_math.h"
#include "orbx_internal.h"

namespace orbx {

constexpr int KEY_SHIFT = 22;                       // index bits; distance (<=256) sits above
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
#ifndef ORBX_MATCH_BLOCK
#define ORBX_MATCH_BLOCK 256
#endif
constexpr int MATCH_BLOCK = ORBX_MATCH_BLOCK;

// k1 <= k2 are the two smallest keys so far: the new second-smallest is the median of (k1, k2, key)
__device__ __forceinline__ void top2_update(uint32_t& k1, uint32_t& k2, uint32_t key) {
    uint32_t m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k1), "v"(k2), "v"(key));
    k2 = m;
    k1 = min(k1, key);
}

// popcount(x) + acc in ONE VALU op (v_bcnt_u32_b32 accumulates); the compiler otherwise builds an add tree
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// Scan train descriptors [t0, t1) for the block's queries.  T must be wave-uniform readable.
template <int QPL>
__device__ __forceinline__ void scan_range(const uint32_t* __restrict__ T, int t0, int t1, const uint32_t (&q)[QPL][8],
                                           uint32_t (&k1)[QPL], uint32_t (&k2)[QPL]) {
    auto one = [&](const uint32_t (&tw)[8], int t) {
#pragma unroll
        for (int j = 0; j < QPL; j++) {
            uint32_t d = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) d = bcnt_acc(q[j][i] ^ tw[i], d);
            top2_update(k1[j], k2[j], (d << KEY_SHIFT) | (uint32_t)t);
        }
    };
    int t = t0;
    // 4 train descriptors per trip: their scalar loads are issued together, so the s_load latency is paid once per 4
    for (; t + 4 <= t1; t += 4) {
        uint32_t tw[4][8];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t* tp = T + (long long)(t + u) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) tw[u][i] = tp[i];   // uniform address -> s_load_dwordx8
        }
#pragma unroll
        for (int u = 0; u < 4; u++) one(tw[u], t + u);
    }
    for (; t < t1; t++) {
        const uint32_t* tp = T + (long long)t * 8;
        uint32_t tw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) tw[i] = tp[i];
        one(tw, t);
    }
}

template <int QPL>
__device__ __forceinline__ void load_queries(const uint32_t* __restrict__ Q, int nq, int qbase, uint32_t (&q)[QPL][8]) {
#pragma unroll
    for (int j = 0; j < QPL; j++) {
        const int qi = qbase + j * MATCH_BLOCK;
        if (qi < nq) {
            const uint4* p = reinterpret_cast<const uint4*>(Q + (long long)qi * 8);
            const uint4 a = p[0], c = p[1];
            q[j][0] = a.x; q[j][1] = a.y; q[j][2] = a.z; q[j][3] = a.w;
            q[j][4] = c.x; q[j][5] = c.y; q[j][6] = c.z; q[j][7] = c.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) q[j][i] = 0;
        }
    }
}

__device__ __forceinline__ void write_result(uint32_t k1, uint32_t k2, int32_t* idx, int32_t* best, int32_t* second) {
    *idx = (k1 == KEY_NONE) ? -1 : (int32_t)(k1 & ((1u << KEY_SHIFT) - 1));
    *best = (k1 == KEY_NONE) ? INT_MAX : (int32_t)(k1 >> KEY_SHIFT);
    *second = (k2 == KEY_NONE) ? INT_MAX : (int32_t)(k2 >> KEY_SHIFT);
}

// Large single problem: grid = (query blocks, train splits); partial (k1,k2) per (split, query).
template <int QPL>
__global__ __launch_bounds__(MATCH_BLOCK) void k_match_split(const uint32_t* __restrict__ Q, int nq, const uint32_t* __restrict__ T, int nt,
                                                             int chunk, uint32_t* __restrict__ pk1, uint32_t* __restrict__ pk2) {
    const int qbase = blockIdx.x * (MATCH_BLOCK * QPL) + threadIdx.x;
    const int t0 = blockIdx.y * chunk, t1 = min(nt, t0 + chunk);
    uint32_t q[QPL][8], k1[QPL], k2[QPL];
    load_queries<QPL>(Q, nq, qbase, q);
#pragma unroll
    for (int j = 0; j < QPL; j++) { k1[j] = KEY_NONE; k2[j] = KEY_NONE; }
    scan_range<QPL>(T, t0, t1, q, k1, k2);
#pragma unroll
    for (int j = 0; j < QPL; j++) {
        const int qi = qbase + j * MATCH_BLOCK;
        if (qi < nq) {
            pk1[(long long)blockIdx.y * nq + qi] = k1[j];
            pk2[(long long)blockIdx.y * nq + qi] = k2[j];
        }
    }
}

__global__ __launch_bounds__(256) void k_match_merge(const uint32_t* __restrict__ pk1, const uint32_t* __restrict__ pk2, int nq, int nsplit,
                                                     int32_t* __restrict__ idx, int32_t* __restrict__ best, int32_t* __restrict__ second) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t k1 = KEY_NONE, k2 = KEY_NONE;
    for (int s = 0; s < nsplit; s++) {
        top2_update(k1, k2, pk1[(long long)s * nq + qi]);
        top2_update(k1, k2, pk2[(long long)s * nq + qi]);
    }
    write_result(k1, k2, idx + qi, best + qi, second + qi);
}

// Many small problems (frame-to-frame matching): blockIdx.y = problem, sizes read on the device.
template <int QPL>
__global__ __launch_bounds__(MATCH_BLOCK) void k_match_batch(const uint32_t* __restrict__ Q, const int32_t* __restrict__ nqs,
                                                             const uint32_t* __restrict__ T, const int32_t* __restrict__ nts, int cap,
                                                             int32_t* __restrict__ idx, int32_t* __restrict__ best, int32_t* __restrict__ second) {
    const int prob = blockIdx.y;
    const int nq = min(nqs[prob], cap), nt = min(nts[prob], cap);
    const int qbase = blockIdx.x * (MATCH_BLOCK * QPL) + threadIdx.x;
    if (blockIdx.x * (MATCH_BLOCK * QPL) >= nq) return;
    const uint32_t* Qp = Q + (long long)prob * cap * 8;
    const uint32_t* Tp = T + (long long)prob * cap * 8;
    uint32_t q[QPL][8], k1[QPL], k2[QPL];
    load_queries<QPL>(Qp, nq, qbase, q);
#pragma unroll
    for (int j = 0; j < QPL; j++) { k1[j] = KEY_NONE; k2[j] = KEY_NONE; }
    scan_range<QPL>(Tp, 0, nt, q, k1, k2);
#pragma unroll
    for (int j = 0; j < QPL; j++) {
        const int qi = qbase + j * MATCH_BLOCK;
        if (qi < nq) {
            const long long o = (long long)prob * cap + qi;
            write_result(k1[j], k2[j], idx + o, best + o, second + o);
        }
    }
}

// Candidate-set form (what every ORBmatcher search really scans: the grid window of GetFeaturesInArea or the features of
// one vocabulary node): query q scans the train descriptors cand[seg_off[q] .. seg_off[q+1]) IN LIST ORDER.  One wave per
// query, one candidate per lane and step; key = (distance << 22) | position-in-list keeps the reference's "first candidate
// attaining the best distance" rule; a butterfly of min / med3 steps reduces the 64 lanes' (k1,k2) pairs.
__device__ __forceinline__ void top2_merge(uint32_t& k1, uint32_t& k2, uint32_t o1, uint32_t o2) {
    // two sorted pairs -> the two smallest of the four keys
    const uint32_t lo = min(k1, o1), hi = max(k1, o1);
    k2 = min(hi, min(k2, o2));
    k1 = lo;
}

__global__ __launch_bounds__(256) void k_match_segments(const uint32_t* __restrict__ Q, int nq, const uint32_t* __restrict__ T, int nt,
                                                        const int32_t* __restrict__ seg_off, const int32_t* __restrict__ cand,
                                                        int32_t* __restrict__ idx, int32_t* __restrict__ best, int32_t* __restrict__ second) {
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (q >= nq) return;
    const int lane = threadIdx.x & 63;
    const int s0 = seg_off[q], s1 = seg_off[q + 1];
    uint32_t qw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) qw[i] = Q[(long long)q * 8 + i];   // wave-uniform: scalar loads
    uint32_t k1 = KEY_NONE, k2 = KEY_NONE;
    for (int base = s0; base < s1; base += 64) {
        const int p = base + lane;
        if (p < s1) {
            const int t = cand[p];
            uint32_t key = KEY_NONE - 1;   // invalid candidate index: never wins, never reported
            if ((unsigned)t < (unsigned)nt) {
                const uint4* tp = reinterpret_cast<const uint4*>(T + (long long)t * 8);
                const uint4 a = tp[0], c = tp[1];
                uint32_t d = 0;
                d = bcnt_acc(qw[0] ^ a.x, d); d = bcnt_acc(qw[1] ^ a.y, d); d = bcnt_acc(qw[2] ^ a.z, d); d = bcnt_acc(qw[3] ^ a.w, d);
                d = bcnt_acc(qw[4] ^ c.x, d); d = bcnt_acc(qw[5] ^ c.y, d); d = bcnt_acc(qw[6] ^ c.z, d); d = bcnt_acc(qw[7] ^ c.w, d);
                key = (d << KEY_SHIFT) | (uint32_t)(p - s0);
                top2_update(k1, k2, key);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)k1, off, 64), o2 = (uint32_t)__shfl_xor((int)k2, off, 64);
        top2_merge(k1, k2, o1, o2);
    }
    if (lane == 0) {
        int32_t bi, bd, sd;
        write_result(k1, k2, &bi, &bd, &sd);
        idx[q] = bi < 0 ? -1 : cand[s0 + bi];   // list position -> train index
        best[q] = bd;
        second[q] = sd;
    }
}

struct MatchScratch {
    uint32_t* buf = nullptr;
    size_t bytes = 0;
    int device = -1;
};
static thread_local MatchScratch t_scratch;

static int ensure_scratch(size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ORBX_ERR_DEVICE;
    if (t_scratch.buf && t_scratch.bytes >= bytes && t_scratch.device == dev) return ORBX_OK;
    if (t_scratch.buf) (void)hipFree(t_scratch.buf);
    t_scratch.buf = nullptr;
    if (hipMalloc(&t_scratch.buf, bytes) != hipSuccess) return ORBX_ERR_DEVICE;
    t_scratch.bytes = bytes;
    t_scratch.device = dev;
    return ORBX_OK;
}


// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:216-244), one wave per map point.  Lane i owns row i of
// the N x N distance matrix: it never stores the row — the median (element (int)(0.5*(N-1)) of the sorted row) is found by
// bisection on the value range 0..256, each step counting the row's distances <= mid by recomputing them (8 xor + 8 popcount
// per pair; the other descriptors arrive by wave-uniform scalar loads).  The winner is the smallest (median << 16 | i).
__global__ __launch_bounds__(MATCH_BLOCK) void k_distinctive(const uint32_t* __restrict__ desc, const int32_t* __restrict__ seg_off, int npoints,
                                                            int32_t* __restrict__ best_idx, int32_t* __restrict__ best_median) {
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (MATCH_BLOCK / 64) + (threadIdx.x >> 6)));
    if (p >= npoints) return;
    const int s0 = seg_off[p], N = seg_off[p + 1] - s0;
    if (N <= 0) { if (lane == 0) { best_idx[p] = -1; best_median[p] = INT_MAX; } return; }
    const int m = (int)(0.5 * (double)(N - 1));              // `vDists[0.5*(N-1)]`
    const uint32_t* D = desc + (size_t)s0 * 8;
    uint32_t bestkey = 0xFFFFFFFFu;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; w++) q[w] = i < N ? D[(size_t)i * 8 + w] : 0u;
        int lo = 0, hi = 256;                                // smallest v with #{j : d(i,j) <= v} >= m + 1
        while (__any(lo < hi)) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; j++) {
                const uint32_t* t = D + (size_t)j * 8;       // wave-uniform address: scalar loads
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) d = bcnt_acc(q[w] ^ t[w], d);
                cnt += (int)d <= mid;
            }
            if (lo < hi) { if (cnt >= m + 1) hi = mid; else lo = mid + 1; }
        }
        if (i < N) bestkey = min(bestkey, ((uint32_t)lo << 16) | (uint32_t)i);
    }
    for (int s = 32; s > 0; s >>= 1) bestkey = min(bestkey, (uint32_t)__shfl_xor((int)bestkey, s, 64));
    if (lane == 0) { best_idx[p] = (int32_t)(bestkey & 0xFFFFu); best_median[p] = (int32_t)(bestkey >> 16); }
}

}  // namespace orbx

using namespace orbx;

extern "C" {

int orbm_hamming256(const uint8_t* a, const uint8_t* b) {
    uint32_t wa[8], wb[8];
    memcpy(wa, a, 32);
    memcpy(wb, b, 32);
    return hamming256_words(wa, wb);
}

int orbm_count_accepted(const int32_t* best, const int32_t* second, int nq, int th, float ratio) {
    int n = 0;
    for (int q = 0; q < nq; q++)
        if (best[q] <= th && (float)best[q] < ratio * (float)second[q]) n++;
    return n;
}

int orbm_match_top2_device(const uint8_t* dQ, int nq, const uint8_t* dT, int nt, int32_t* d_best_idx, int32_t* d_best,
                           int32_t* d_second, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nq < 0 || nt < 0 || nt >= (1 << KEY_SHIFT)) return ORBX_ERR_ARG;
    if (nq == 0) return ORBX_OK;
    if (((uintptr_t)dQ & 15) || ((uintptr_t)dT & 3)) return ORBX_ERR_ARG;
    constexpr int QPL = 2, Q_PER_BLOCK = MATCH_BLOCK * QPL;   // big problems: 2 queries per lane halve the scalar T traffic
    const int qblocks = (nq + Q_PER_BLOCK - 1) / Q_PER_BLOCK;
    // enough workgroups to fill 256 CUs several times over, but chunks of >= 256 train descriptors
    int nsplit = std::max(1, std::min((nt + 255) / 256, (4096 + qblocks - 1) / qblocks));
    const int chunk = nt > 0 ? (nt + nsplit - 1) / nsplit : 1;
    nsplit = nt > 0 ? (nt + chunk - 1) / chunk : 1;
    const size_t need = (size_t)2 * nsplit * nq * sizeof(uint32_t);
    if (ensure_scratch(need) != ORBX_OK) return ORBX_ERR_DEVICE;
    uint32_t* pk1 = t_scratch.buf;
    uint32_t* pk2 = pk1 + (size_t)nsplit * nq;
    hipLaunchKernelGGL(k_match_split<QPL>, dim3(qblocks, nsplit), dim3(MATCH_BLOCK), 0, stream, (const uint32_t*)dQ, nq, (const uint32_t*)dT, nt,
                       chunk, pk1, pk2);
    if (hipGetLastError() != hipSuccess) return ORBX_ERR_DEVICE;
    hipLaunchKernelGGL(k_match_merge, dim3((nq + 255) / 256), dim3(256), 0, stream, pk1, pk2, nq, nsplit, d_best_idx, d_best, d_second);
    if (hipGetLastError() != hipSuccess) return ORBX_ERR_DEVICE;
    return ORBX_OK;
}

int orbm_match_top2_batch_device(const uint8_t* dQ, const int32_t* d_nq, const uint8_t* dT, const int32_t* d_nt, int nbatch, int cap,
                                 int32_t* d_best_idx, int32_t* d_best, int32_t* d_second, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nbatch < 0 || cap < 1 || cap >= (1 << KEY_SHIFT)) return ORBX_ERR_ARG;
    if (nbatch == 0) return ORBX_OK;
    if (((uintptr_t)dQ & 15) || ((uintptr_t)dT & 3)) return ORBX_ERR_ARG;
    // frame-sized problems (~1000 x 1000): 1 query per lane so that a batch of 256 still fills the chip (4 waves per SIMD)
#ifndef ORBX_MATCH_QPL
#define ORBX_MATCH_QPL 1
#endif
    constexpr int BQPL = ORBX_MATCH_QPL;
    hipLaunchKernelGGL(k_match_batch<BQPL>, dim3((cap + MATCH_BLOCK * BQPL - 1) / (MATCH_BLOCK * BQPL), nbatch), dim3(MATCH_BLOCK), 0, stream, (const uint32_t*)dQ,
                       d_nq, (const uint32_t*)dT, d_nt, cap, d_best_idx, d_best, d_second);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbm_match_top2_segments_device(const uint8_t* dQ, int nq, const uint8_t* dT, int nt, const int32_t* d_seg_off, const int32_t* d_cand,
                                    int32_t* d_best_idx, int32_t* d_best, int32_t* d_second, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nq < 0 || nt < 0 || !d_seg_off || !d_cand) return ORBX_ERR_ARG;
    if (nq == 0) return ORBX_OK;
    if (((uintptr_t)dQ & 3) || ((uintptr_t)dT & 15)) return ORBX_ERR_ARG;
    hipLaunchKernelGGL(k_match_segments, dim3((nq + 3) / 4), dim3(256), 0, stream, (const uint32_t*)dQ, nq, (const uint32_t*)dT, nt, d_seg_off,
                       d_cand, d_best_idx, d_best, d_second);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbm_match_top2_segments(const uint8_t* Q, int nq, const uint8_t* T, int nt, const int32_t* seg_off, const int32_t* cand, int32_t* best_idx,
                             int32_t* best, int32_t* second, int device) {
    if (nq < 0 || nt < 0 || !seg_off || (nq > 0 && seg_off[nq] > 0 && !cand)) return ORBX_ERR_ARG;
    if (nq == 0) return ORBX_OK;
    const int ncand = seg_off[nq];
    if (ncand < 0 || ncand >= (1 << KEY_SHIFT)) return ORBX_ERR_ARG;
    for (int q = 0; q < nq; q++)
        if (seg_off[q] > seg_off[q + 1] || seg_off[q] < 0) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    uint8_t *dQ = nullptr, *dT = nullptr;
    int32_t *dseg = nullptr, *dcand = nullptr, *dout = nullptr;
    int rc = ORBX_ERR_DEVICE;
    if (hipMalloc(&dQ, (size_t)nq * 32) == hipSuccess && hipMalloc(&dT, (size_t)std::max(nt, 1) * 32) == hipSuccess &&
        hipMalloc(&dseg, (size_t)(nq + 1) * 4) == hipSuccess && hipMalloc(&dcand, (size_t)std::max(ncand, 1) * 4) == hipSuccess &&
        hipMalloc(&dout, (size_t)nq * 12) == hipSuccess && hipMemcpy(dQ, Q, (size_t)nq * 32, hipMemcpyHostToDevice) == hipSuccess &&
        (nt == 0 || hipMemcpy(dT, T, (size_t)nt * 32, hipMemcpyHostToDevice) == hipSuccess) &&
        hipMemcpy(dseg, seg_off, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
        (ncand == 0 || hipMemcpy(dcand, cand, (size_t)ncand * 4, hipMemcpyHostToDevice) == hipSuccess)) {
        rc = orbm_match_top2_segments_device(dQ, nq, dT, nt, dseg, dcand, dout, dout + nq, dout + 2 * (size_t)nq, nullptr);
        if (rc == ORBX_OK && (hipMemcpy(best_idx, dout, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(best, dout + nq, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(second, dout + 2 * (size_t)nq, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = ORBX_ERR_DEVICE;
    }
    if (dQ) (void)hipFree(dQ);
    if (dT) (void)hipFree(dT);
    if (dseg) (void)hipFree(dseg);
    if (dcand) (void)hipFree(dcand);
    if (dout) (void)hipFree(dout);
    return rc;
}

int orbm_match_top2(const uint8_t* Q, int nq, const uint8_t* T, int nt, int32_t* best_idx, int32_t* best, int32_t* second, int device) {
    if (nq < 0 || nt < 0) return ORBX_ERR_ARG;
    if (nq == 0) return ORBX_OK;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    uint8_t *dQ = nullptr, *dT = nullptr;
    int32_t* dout = nullptr;
    int rc = ORBX_ERR_DEVICE;
    if (hipMalloc(&dQ, (size_t)nq * 32) == hipSuccess && hipMalloc(&dT, (size_t)std::max(nt, 1) * 32) == hipSuccess &&
        hipMalloc(&dout, (size_t)nq * 3 * sizeof(int32_t)) == hipSuccess &&
        hipMemcpy(dQ, Q, (size_t)nq * 32, hipMemcpyHostToDevice) == hipSuccess &&
        (nt == 0 || hipMemcpy(dT, T, (size_t)nt * 32, hipMemcpyHostToDevice) == hipSuccess)) {
        rc = orbm_match_top2_device(dQ, nq, dT, nt, dout, dout + nq, dout + 2 * (size_t)nq, nullptr);
        if (rc == ORBX_OK) {
            if (hipMemcpy(best_idx, dout, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(best, dout + nq, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(second, dout + 2 * (size_t)nq, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess)
                rc = ORBX_ERR_DEVICE;
        }
    }
    if (dQ) (void)hipFree(dQ);
    if (dT) (void)hipFree(dT);
    if (dout) (void)hipFree(dout);
    return rc;
}


int orbm_distinctive_device(const uint8_t* d_desc, const int32_t* d_seg_off, int npoints, int32_t* d_best_idx, int32_t* d_best_median, void* stream) {
    if (npoints < 0) return ORBX_ERR_ARG;
    if (npoints == 0) return ORBX_OK;
    if (!d_desc || !d_seg_off || !d_best_idx || !d_best_median) return ORBX_ERR_ARG;
    const int per_block = orbx::MATCH_BLOCK / 64;
    hipLaunchKernelGGL(orbx::k_distinctive, dim3((npoints + per_block - 1) / per_block), dim3(orbx::MATCH_BLOCK), 0, (hipStream_t)stream,
                       (const uint32_t*)d_desc, d_seg_off, npoints, d_best_idx, d_best_median);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbm_distinctive(const uint8_t* desc, const int32_t* seg_off, int npoints, int32_t* best_idx, int32_t* best_median, int device) {
    if (npoints < 0 || (npoints > 0 && (!seg_off || !best_idx || !best_median))) return ORBX_ERR_ARG;
    if (npoints == 0) return ORBX_OK;
    const int total = seg_off[npoints];
    if (total < 0 || total >= 65536 * 64 || (total > 0 && !desc)) return ORBX_ERR_ARG;
    for (int p = 0; p < npoints; p++) if (seg_off[p + 1] < seg_off[p] || seg_off[p + 1] - seg_off[p] > 65535) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    uint8_t* d = nullptr;
    const size_t o_seg = (size_t)std::max(total, 1) * 32, o_out = o_seg + ((size_t)npoints + 1) * 4, bytes = o_out + (size_t)npoints * 8;
    int rc = ORBX_ERR_DEVICE;
    if (hipMalloc(&d, bytes) == hipSuccess && (total == 0 || hipMemcpy(d, desc, (size_t)total * 32, hipMemcpyHostToDevice) == hipSuccess) &&
        hipMemcpy(d + o_seg, seg_off, ((size_t)npoints + 1) * 4, hipMemcpyHostToDevice) == hipSuccess) {
        int32_t* out = (int32_t*)(d + o_out);
        rc = orbm_distinctive_device(d, (const int32_t*)(d + o_seg), npoints, out, out + npoints, nullptr);
        if (rc == ORBX_OK && (hipMemcpy(best_idx, out, (size_t)npoints * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                              hipMemcpy(best_median, out + npoints, (size_t)npoints * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = ORBX_ERR_DEVICE;
    }
    if (d) (void)hipFree(d);
    return rc;
}

}  // extern "C"
