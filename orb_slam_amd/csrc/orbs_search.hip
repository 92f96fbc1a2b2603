// Greedy grid-window searches on gfx950 (include/orbs.h): exact ORBmatcher semantics —
// queries resolved IN ORDER, each skipping the train features claimed by earlier queries — batched over independent
// problems (one per frame / frame pair).
//
// Mapping.  The sequential dependence runs through ~2 bytes of state per train feature, so one problem = one wavefront
// and the whole TRAIN FRAME IS STAGED IN LDS in grid (CSR) order: x, y, (index | octave << 16), the 32-byte descriptor,
// the 3073 cell offsets (u16) and the claim state.  A query then touches only LDS:
//   * its window is a few grid columns; the cells (col, y0..y1) of one column are one contiguous CSR run, so the wave
//     takes 8 columns at a time, 8 lanes per column, each lane striding its column's run;
//   * a candidate's key is (distance << 16 | CSR position): CSR position order IS the reference's candidate order
//     (cells x-major, then y, then push_back order), so the two smallest keys are exactly the best / second-best the
//     reference's `if(d<best)... else if(d<best2)` scan ends with, first-listed candidate on ties;
//   * two DPP min-reductions give best and second; the accept rule and the state update are wave-uniform.
// 64 queries at a time are preloaded (one per lane) and broadcast with v_readlane, so the in-order loop has no global
// loads on its critical path except the matched keypoint's angle when the rotation histogram is on.
// Per problem the cost is ~nq x (LDS latency chain + 2 reductions); problems run concurrently, 1-2 per CU.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstring>

#include "orbf_math.h"
#include "orbs.h"

namespace orbs {

constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
constexpr int HISTO_LENGTH = 30;             // src/ORBmatcher.cc:42
constexpr int CHUNK = 64;

struct Layout { uint32_t off16, tx, ty, tmeta, tdesc, state, t2q, q2t, binv, qdesc, hist, total; };

__host__ __device__ inline Layout make_layout(int cap, int qcap) {
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    Layout L;
    uint32_t o = 0;
    L.tdesc = o; o += al((uint32_t)cap * 32);
    L.qdesc = o; o += CHUNK * 32;
    L.tx = o; o += al((uint32_t)cap * 4);
    L.ty = o; o += al((uint32_t)cap * 4);
    L.tmeta = o; o += al((uint32_t)cap * 4);
    L.off16 = o; o += al((ORBF_GRID_CELLS + 1) * 2);
    L.state = o; o += al((uint32_t)cap * 2);
    L.t2q = o; o += al((uint32_t)cap * 2);
    L.q2t = o; o += al((uint32_t)qcap * 2);
    L.binv = o; o += al((uint32_t)(cap > qcap ? cap : qcap));
    L.hist = o; o += 32 * 4;
    L.total = o;
    return L;
}

struct Args {
    const orbx_keypoint* kps_un;
    const uint8_t* desc;
    const int32_t* cell_off;
    const int32_t* cell_feat;
    const int32_t* nt;
    const uint8_t* claimed;
    const float* qxyr;
    const int32_t* qlev;
    const uint8_t* qdesc;
    const float* qangle;
    const uint8_t* qvalid;
    const int32_t* nq;
    int32_t* q2t;
    int32_t* t2q;
    int32_t* best;
    int32_t* second;
    int32_t* nmatches;
    int cap, qcap;
};

// minimum over the 64 lanes, returned wave-uniform: 4 DPP steps inside each row of 16, then the 4 rows via readlane
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}

__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// src/ORBmatcher.cc:234-241 and siblings: rot in [0,360), bin = round(rot/30) (only bins 0..12 are ever hit)
__device__ __forceinline__ int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1748-1789) on bin sizes
__host__ __device__ inline void three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
}

__global__ __launch_bounds__(64) void k_window_search(orbf_bounds b, orbs_params prm, Args a) {
    extern __shared__ __align__(16) uint8_t lds[];
    const Layout L = make_layout(a.cap, a.qcap);
    uint16_t* off16 = (uint16_t*)(lds + L.off16);
    float* tx = (float*)(lds + L.tx);
    float* ty = (float*)(lds + L.ty);
    uint32_t* tmeta = (uint32_t*)(lds + L.tmeta);
    uint4* tdesc = (uint4*)(lds + L.tdesc);
    uint16_t* state = (uint16_t*)(lds + L.state);
    int16_t* t2q = (int16_t*)(lds + L.t2q);
    int16_t* q2t = (int16_t*)(lds + L.q2t);
    uint8_t* binv = lds + L.binv;
    uint4* qdl = (uint4*)(lds + L.qdesc);
    int* hist = (int*)(lds + L.hist);

    const int p = blockIdx.x, lane = threadIdx.x;
    const int nt = min(a.nt[p], a.cap), nq = min(a.nq[p], a.qcap);
    const int rule = prm.rule;
    const size_t tb = (size_t)p * a.cap, qb = (size_t)p * a.qcap;
    const int32_t* coff = a.cell_off + (size_t)p * (ORBF_GRID_CELLS + 1);
    const int32_t* cfeat = a.cell_feat + tb;
    const orbx_keypoint* kps = a.kps_un + tb;

    // ---- stage the train frame in LDS, in grid order
    for (int i = lane; i <= ORBF_GRID_CELLS; i += 64) off16[i] = (uint16_t)min(coff[i], a.cap);
    const int m = min(coff[ORBF_GRID_CELLS], a.cap);
    for (int j = lane; j < m; j += 64) {
        const int f = cfeat[j];
        const orbx_keypoint kp = kps[f];
        tx[j] = kp.x;
        ty[j] = kp.y;
        tmeta[j] = (uint32_t)f | ((uint32_t)kp.octave << 16);
        const uint4* d = (const uint4*)(a.desc + (tb + f) * 32);
        tdesc[2 * j] = d[0];
        tdesc[2 * j + 1] = d[1];
    }
    for (int i = lane; i < nt; i += 64) {
        state[i] = rule == ORBS_RULE_INIT ? (uint16_t)0xFFFF : (uint16_t)((a.claimed && a.claimed[tb + i]) ? 1 : 0);
        t2q[i] = -1;
    }
    for (int i = lane; i < nq; i += 64) q2t[i] = -1;
    const int nbin = rule == ORBS_RULE_INIT ? nq : nt;
    for (int i = lane; i < nbin; i += 64) binv[i] = 255;
    if (lane < 32) hist[lane] = 0;
    __syncthreads();

    const bool rot_on = prm.check_orientation != 0 && rule != ORBS_RULE_MAPPOINTS;

    for (int q0 = 0; q0 < nq; q0 += CHUNK) {
        // ---- one query per lane: parameters into registers, descriptors into LDS
        const int qi = q0 + lane;
        float qx = 0.f, qy = 0.f, qr = 0.f, qa = 0.f;
        int ql0 = 0, ql1 = 0, qv = 0;
        if (qi < nq) {
            qx = a.qxyr[(qb + qi) * 3]; qy = a.qxyr[(qb + qi) * 3 + 1]; qr = a.qxyr[(qb + qi) * 3 + 2];
            ql0 = a.qlev[(qb + qi) * 2]; ql1 = a.qlev[(qb + qi) * 2 + 1];
            qv = a.qvalid ? (a.qvalid[qb + qi] != 0) : 1;
            if (a.qangle) qa = a.qangle[qb + qi];
            const uint4* d = (const uint4*)(a.qdesc + (qb + qi) * 32);
            qdl[2 * lane] = d[0];
            qdl[2 * lane + 1] = d[1];
        }
        __syncthreads();
        int my_best = -1, my_second = -1;
        const int nchunk = min(CHUNK, nq - q0);
        for (int jq = 0; jq < nchunk; jq++) {
            if (!__builtin_amdgcn_readlane(qv, jq)) continue;
            const float x = lane_f(qx, jq), y = lane_f(qy, jq), r = lane_f(qr, jq);
            const int minLevel = __builtin_amdgcn_readlane(ql0, jq), maxLevel = __builtin_amdgcn_readlane(ql1, jq);
            int x0, x1, y0, y1;
            if (!orbf::window_cells(b, x, y, r, &x0, &x1, &y0, &y1)) continue;
            x0 = uni(x0); x1 = uni(x1); y0 = uni(y0); y1 = uni(y1);
            const uint4 qd0 = qdl[2 * jq], qd1 = qdl[2 * jq + 1];
            uint32_t a1 = KEY_NONE, a2 = KEY_NONE;
            int any = 0;
            for (int cx = x0; cx <= x1; cx += 8) {
                const int col = cx + (lane >> 3);
                int j = 0, jend = 0;
                if (col <= x1) {
                    j = off16[col * ORBF_GRID_ROWS + y0] + (lane & 7);
                    jend = off16[col * ORBF_GRID_ROWS + y1 + 1];
                }
                for (; j < jend; j += 8) {
                    const uint32_t meta = tmeta[j];
                    if (!orbf::in_window(tx[j], ty[j], (int)(meta >> 16), x, y, r, minLevel, maxLevel)) continue;
                    any = 1;
                    const uint32_t st = state[meta & 0xFFFFu];
                    if (rule != ORBS_RULE_INIT && st) continue;                         // `if(F.mvpMapPoints[idx]) continue;`
                    const uint4 t0 = tdesc[2 * j], t1 = tdesc[2 * j + 1];
                    const uint32_t dist = __popc(t0.x ^ qd0.x) + __popc(t0.y ^ qd0.y) + __popc(t0.z ^ qd0.z) + __popc(t0.w ^ qd0.w) +
                                          __popc(t1.x ^ qd1.x) + __popc(t1.y ^ qd1.y) + __popc(t1.z ^ qd1.z) + __popc(t1.w ^ qd1.w);
                    if (rule == ORBS_RULE_INIT && st <= dist) continue;                 // `if(vMatchedDistance[i2]<=dist) continue;`
                    const uint32_t key = (dist << 16) | (uint32_t)j;
                    if (key < a1) { a2 = a1; a1 = key; }
                    else if (key < a2) a2 = key;
                }
            }
            if (__ballot(any) == 0ull) continue;                  // empty window: the reference `continue`s before the scan
            const uint32_t k1 = wave_min_u32(a1);
            if (a1 == k1) a1 = a2;
            const uint32_t k2 = wave_min_u32(a1);
            int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1, bestLevel = -1, bestLevel2 = -1;
            if (k1 != KEY_NONE) { const uint32_t mt = tmeta[k1 & 0xFFFFu]; bestDist = (int)(k1 >> 16); bestIdx = (int)(mt & 0xFFFFu); bestLevel = (int)(mt >> 16); }
            if (k2 != KEY_NONE) { const uint32_t mt = tmeta[k2 & 0xFFFFu]; bestDist2 = (int)(k2 >> 16); bestLevel2 = (int)(mt >> 16); }
            if (lane == jq) { my_best = bestDist; my_second = bestDist2; }
            bool accept;
            if (rule == ORBS_RULE_MAPPOINTS)
                accept = bestDist <= prm.th && !(bestLevel == bestLevel2 && (float)bestDist > prm.ratio * (float)bestDist2);
            else if (rule == ORBS_RULE_WINDOW)
                accept = (float)bestDist <= (float)bestDist2 * prm.ratio && bestDist <= prm.th;
            else if (rule == ORBS_RULE_BEST)
                accept = bestDist <= prm.th;
            else
                accept = bestDist <= prm.th && (float)bestDist < (float)bestDist2 * prm.ratio;
            if (!accept) continue;
            const int q = q0 + jq;
            int bin = 255;
            if (rot_on) bin = rot_bin(lane_f(qa, jq), kps[bestIdx].angle);
            if (lane == 0) {
                if (rule == ORBS_RULE_INIT) {
                    const int prev = t2q[bestIdx];
                    if (prev >= 0) q2t[prev] = -1;                 // vnMatches12[vnMatches21[bestIdx2]] = -1
                    state[bestIdx] = (uint16_t)bestDist;           // vMatchedDistance[bestIdx2] = bestDist
                    binv[q] = (uint8_t)bin;                        // rotHist[bin].push_back(i1)
                } else {
                    state[bestIdx] = 1;
                    binv[bestIdx] = (uint8_t)bin;                  // rotHist[bin].push_back(bestIdx2)
                }
                q2t[q] = (int16_t)bestIdx;
                t2q[bestIdx] = (int16_t)q;
            }
        }
        if (qi < nq) {
            if (a.best) a.best[qb + qi] = my_best;
            if (a.second) a.second[qb + qi] = my_second;
        }
        __syncthreads();
    }

    // ---- rotation consistency (the rotHist blocks + ComputeThreeMaxima)
    if (rot_on) {
        for (int i = lane; i < nbin; i += 64) { const int bn = binv[i]; if (bn != 255) atomicAdd(&hist[bn], 1); }
        __syncthreads();
        int i1, i2, i3;
        three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
        for (int i = lane; i < nbin; i += 64) {
            const int bn = binv[i];
            if (bn == 255 || bn == i1 || bn == i2 || bn == i3) continue;
            if (rule == ORBS_RULE_INIT) {
                const int t = q2t[i];
                if (t >= 0) { t2q[t] = -1; q2t[i] = -1; }
            } else {
                const int q = t2q[i];
                if (q >= 0) { q2t[q] = -1; t2q[i] = -1; }
            }
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = lane; i < nq; i += 64) { const int t = q2t[i]; a.q2t[qb + i] = t; cnt += t >= 0; }
    for (int i = lane; i < nt; i += 64) a.t2q[tb + i] = t2q[i];
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
    if (lane == 0) a.nmatches[p] = cnt;
}

}  // namespace orbs

extern "C" {

size_t orbs_lds_bytes(int cap, int qcap) {
    if (cap < 1 || qcap < 1) return 0;
    return orbs::make_layout(cap, qcap).total;
}

void orbs_three_maxima(const int32_t* sizes, int L, int32_t* ind) {
    int a, b, c;
    orbs::three_maxima(sizes, L, a, b, c);
    ind[0] = a; ind[1] = b; ind[2] = c;
}

int orbs_window_search_batch_device(const orbf_bounds* b, const orbs_params* prm, const orbx_keypoint* d_kps_un, const uint8_t* d_desc,
                                    const int32_t* d_cell_off, const int32_t* d_cell_feat, const int32_t* d_nt, int cap, const uint8_t* d_claimed,
                                    const float* d_qxyr, const int32_t* d_qlev, const uint8_t* d_qdesc, const float* d_qangle,
                                    const uint8_t* d_qvalid, const int32_t* d_nq, int qcap, int nproblems, int32_t* d_q2t, int32_t* d_t2q,
                                    int32_t* d_best, int32_t* d_second, int32_t* d_nmatches, void* stream) {
    if (!b || !prm || nproblems < 0 || cap < 1 || cap > ORBF_MAX_FEATURES || qcap < 1 || qcap > ORBF_MAX_FEATURES) return ORBX_ERR_ARG;
    if (prm->rule < ORBS_RULE_MAPPOINTS || prm->rule > ORBS_RULE_INIT) return ORBX_ERR_ARG;
    if (nproblems == 0) return ORBX_OK;
    if (!d_kps_un || !d_desc || !d_cell_off || !d_cell_feat || !d_nt || !d_qxyr || !d_qlev || !d_qdesc || !d_nq || !d_q2t || !d_t2q || !d_nmatches)
        return ORBX_ERR_ARG;
    if (prm->check_orientation && prm->rule != ORBS_RULE_MAPPOINTS && !d_qangle) return ORBX_ERR_ARG;
    const size_t lds = orbs::make_layout(cap, qcap).total;
    if (lds > 160 * 1024) return ORBX_ERR_CAPACITY;
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        if (hipFuncSetAttribute((const void*)orbs::k_window_search, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
        attr_bytes = lds;
    }
    orbs::Args a{d_kps_un, d_desc, d_cell_off, d_cell_feat, d_nt, d_claimed, d_qxyr, d_qlev, d_qdesc, d_qangle, d_qvalid, d_nq,
                 d_q2t, d_t2q, d_best, d_second, d_nmatches, cap, qcap};
    hipLaunchKernelGGL(orbs::k_window_search, dim3(nproblems), dim3(64), lds, (hipStream_t)stream, *b, *prm, a);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

}  // extern "C"
