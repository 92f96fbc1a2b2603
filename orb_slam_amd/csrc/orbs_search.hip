// Greedy grid-window searches on gfx950 (include/orbs.h): exact ORBmatcher semantics —
// queries resolved IN ORDER, each skipping the train features claimed by earlier queries — batched over independent
// problems (one per frame / frame pair).
//
// Mapping.  The sequential dependence runs through ~2 bytes of state per train feature, so one problem = one wavefront
// and the whole TRAIN FRAME IS STAGED IN LDS in grid (CSR) order: x, y, (index | octave << 16), the 32-byte descriptor,
// the 3073 cell offsets (u16) and the claim state.  A query then touches only LDS:
//   * its window is a few grid columns; the cells (col, y0..y1) of one column are one contiguous CSR run;
//   * a candidate's key is (distance << 16 | CSR position): CSR position order IS the reference's candidate order
//     (cells x-major, then y, then push_back order), so the two smallest keys are exactly the best / second-best the
//     reference's `if(d<best)... else if(d<best2)` scan ends with, first-listed candidate on ties.
// Queries go 256 at a time, one per thread (four waves): every lane scans its own window SPECULATIVELY against the claim
// state of the group start and keeps its six smallest keys; the group is then committed in query order, wave after wave,
// in vectorised rounds (see k_window_search).  Claims only ever remove candidates, so a query's exact best / second are
// the first two still-admissible entries of its list; only a list that runs dry makes the wave rescan that one window
// (8 columns x 8 lanes, two DPP min-reductions).  Nothing on the in-order path touches global memory.
#include <atomic>
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "orbf_math.h"
#include "orbs.h"

namespace orbs {

constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;
constexpr int HISTO_LENGTH = 30;             // src/ORBmatcher.cc:42

struct Layout { uint32_t off16, tx, ty, tang, tmeta, tdesc, state, claim, t2q, q2t, binv, hist, epi, rank16, r2s, total; };

// Level-bucketed index of the grid searches (round 2).  A query asks for the features of a window AND of a level range
// (Frame::GetFeaturesInArea's minLevel / maxLevel): with the plain 64 x 48 CSR a same-level query at r = 100 walks ~140 entries
// to find the ~18 of its level, and every one costs three scattered LDS reads before the octave test rejects it.  The staged frame
// is therefore ordered by (level bucket, coarse column, coarse row) - coarse cell = 2 x 2 grid cells - with one offset per
// bucket, so a query walks, per level of its range and per coarse column of its window, one contiguous run.  Which entries it
// VISITS changes; what it accepts does not: the exact window / level test still runs on every visited entry, and a candidate's
// key carries its position in the reference's CSR order (rank16), not its LDS slot, so ties break as before.
constexpr int BK_LEVELS = 8;                           // levels >= 7 share the last bucket (the exact octave test sorts them out)
constexpr int BK_COLS = ORBF_GRID_COLS / 2, BK_ROWS = ORBF_GRID_ROWS / 2, BK_CELLS = BK_COLS * BK_ROWS;
constexpr int BK_N = BK_LEVELS * BK_CELLS;             // buckets; the offset table has BK_N + 2 entries (see stage_bucketed)

// desc_in_lds: the train descriptors (32 of the ~57 bytes a staged feature costs) are staged too; frames too large for that
// (beyond ~2850 features) leave them in global memory — the candidates of a window then gather 32 bytes each through L2 — which
// carries the capacity to ~6500 features per problem (an initialisation extractor of 4000 features fits).
__host__ __device__ inline Layout make_layout(int cap, int qcap, bool desc_in_lds, bool bucketed = false) {
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    Layout L;
    uint32_t o = 0;
    L.tdesc = o; o += desc_in_lds ? al((uint32_t)cap * 32) : 0u;
    L.tx = o; o += al((uint32_t)cap * 4);
    L.ty = o; o += al((uint32_t)cap * 4);
    L.tang = o; o += al((uint32_t)cap * 4);
    L.tmeta = o; o += al((uint32_t)cap * 4);
    L.off16 = o; o += al((uint32_t)((bucketed ? BK_N + 2 : ORBF_GRID_CELLS + 1) * 2));   // (BK_N + 2 > ORBF_GRID_CELLS + 1: the fine offsets are staged here first)
    L.state = o; o += al((uint32_t)cap * 2);
    L.claim = o; o += al((uint32_t)cap * 4);
    L.t2q = o; o += al((uint32_t)cap * 2);
    L.q2t = o; o += al((uint32_t)qcap * 2);
    L.binv = o; o += al((uint32_t)(cap > qcap ? cap : qcap));
    L.hist = o; o += 32 * 4;
    L.epi = o; o += ORBS_MAX_LEVELS * 4;
    L.rank16 = o; o += bucketed ? al((uint32_t)cap * 2) : 0u;
    L.r2s = o; o += bucketed ? al((uint32_t)cap * 2) : 0u;
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
    int desc_in_lds;
    int bucketed;              // grid mode with the level-bucketed index (make_layout(.., true))
    // list mode (orbs_list_search_batch_device): cell_feat is the candidate list, nlist its length, qrange the per-query runs
    const int32_t* nlist;
    const int32_t* qrange;
    const int32_t* qindex;
    // ORBS_RULE_TRIANGULATION: query keypoints (slot-indexed like qdesc), F12 per problem, the per-octave bound on dsqr
    const orbx_keypoint* qkps;
    const float* F12;
    float epi_thr[ORBS_MAX_LEVELS];
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
    // angles outside [0, 360) (e.g. -1 = "no orientation") or NaN would leave [0, HISTO_LENGTH): the reference asserts here
    // (ROS_ASSERT(bin >= 0 && bin < HISTO_LENGTH)); such a match simply takes no part in the rotation vote (255 = no bin)
    return (unsigned)bin < (unsigned)HISTO_LENGTH ? bin : 255;
}

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1748-1789) on bin sizes
__host__ __device__ inline void three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {           // the reference's if / else-if chain as selects (keeps everything in registers)
        const int s = sizes[i];
        const bool g1 = s > max1, g2 = !g1 && s > max2, g3 = !g1 && !g2 && s > max3;
        const int n3 = (g1 || g2) ? max2 : (g3 ? s : max3), j3 = (g1 || g2) ? ind2 : (g3 ? i : ind3);
        const int n2 = g1 ? max1 : (g2 ? s : max2), j2 = g1 ? ind1 : (g2 ? i : ind2);
        max3 = n3; ind3 = j3; max2 = n2; ind2 = j2;
        if (g1) { max1 = s; ind1 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// LDS views of one staged problem
struct Staged {
    const uint16_t* off16;
    const float* tx;
    const float* ty;
    const uint32_t* tmeta;
    const uint4* tdesc;        // staged descriptors in grid / list order (only when desc_lds)
    const uint16_t* state;
    const float* epi_thr;
    const uint8_t* gdesc;      // the problem's train descriptors in global memory (read by feature index when !desc_lds)
    bool desc_lds;
    const uint16_t* rank16;    // bucketed index: the slot's position in the reference's CSR order (the low half of its key)
};
__device__ __forceinline__ void staged_desc(const Staged& S, int j, uint32_t meta, uint4& t0, uint4& t1) {
    if (S.desc_lds) { t0 = S.tdesc[2 * j]; t1 = S.tdesc[2 * j + 1]; }
    else { const uint4* d = (const uint4*)(S.gdesc + (size_t)(meta & 0xFFFFu) * 32); t0 = d[0]; t1 = d[1]; }
}

// ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:136-153) split in two: the query's line l = x1' F12 once per query ...
struct EpiLine { float a, b, c, den; int th; };

__device__ __forceinline__ EpiLine epi_line(float x, float y, const float* F, int th) {
    EpiLine e;          // every product and sum rounded on its own, as the reference's float expressions are (no fma)
    e.a = __fadd_rn(__fadd_rn(__fmul_rn(x, F[0]), __fmul_rn(y, F[3])), F[6]);
    e.b = __fadd_rn(__fadd_rn(__fmul_rn(x, F[1]), __fmul_rn(y, F[4])), F[7]);
    e.c = __fadd_rn(__fadd_rn(__fmul_rn(x, F[2]), __fmul_rn(y, F[5])), F[8]);
    e.den = __fadd_rn(__fmul_rn(e.a, e.a), __fmul_rn(e.b, e.b));
    e.th = th;
    return e;
}
// ... and the distance test per candidate.  thr = the smallest float t with (double)t >= 3.84 * sigma2(octave), so that
// `dsqr < thr` in float IS the reference's `(double)dsqr < 3.84*sigma2` (host: epi_bound()).
__device__ __forceinline__ bool epi_ok(const EpiLine& e, float x2, float y2, float thr) {
    const float num = __fadd_rn(__fadd_rn(__fmul_rn(e.a, x2), __fmul_rn(e.b, y2)), e.c);
    if (e.den == 0.0f) return false;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), e.den);
    return dsqr < thr;
}

__device__ __forceinline__ uint32_t hamming_key(const uint4& t0, const uint4& t1, const uint4& q0, const uint4& q1) {
    return __popc(t0.x ^ q0.x) + __popc(t0.y ^ q0.y) + __popc(t0.z ^ q0.z) + __popc(t0.w ^ q0.w) +
           __popc(t1.x ^ q1.x) + __popc(t1.y ^ q1.y) + __popc(t1.z ^ q1.z) + __popc(t1.w ^ q1.w);
}

// One candidate of a window: its key (distance << 16 | CSR position) when it is admissible under the current claim state,
// KEY_NONE when it is in the window but claimed; `inwin` says whether it was in the window at all.
// ORBS_RULE_TRIANGULATION: admissible = unclaimed && distance <= th; bit 15 of the key says whether the candidate also lies on
// the query's epipolar line (list positions stay below 2^15: the staged frame has to fit the LDS).
template <bool BK>
__device__ __forceinline__ uint32_t candidate_key(const Staged& S, int rule, int j, float x, float y, float r, int minLevel, int maxLevel,
                                                  const uint4& q0, const uint4& q1, const EpiLine& E, bool& inwin) {
    const uint32_t meta = S.tmeta[j];
    if (rule == ORBS_RULE_TRIANGULATION) {
        inwin = true;
        if (S.state[meta & 0xFFFFu]) return KEY_NONE;                               // `if(vbMatched2[idx2] || pMP2) continue;`
        uint4 t0, t1;
        staged_desc(S, j, meta, t0, t1);
        const uint32_t dist = hamming_key(t0, t1, q0, q1);
        if ((int)dist > E.th) return KEY_NONE;                                      // `if(dist>TH_LOW) continue;`
        const uint32_t on_line = epi_ok(E, S.tx[j], S.ty[j], S.epi_thr[min((int)(meta >> 16), ORBS_MAX_LEVELS - 1)]) ? 0x8000u : 0u;
        return (dist << 16) | on_line | (uint32_t)j;
    }
    inwin = orbf::in_window(S.tx[j], S.ty[j], (int)(meta >> 16), x, y, r, minLevel, maxLevel);
    if (!inwin) return KEY_NONE;
    const uint32_t st = S.state[meta & 0xFFFFu];
    if (rule != ORBS_RULE_INIT && st) return KEY_NONE;                              // `if(F.mvpMapPoints[idx]) continue;`
    uint4 t0, t1;
    staged_desc(S, j, meta, t0, t1);
    const uint32_t dist = hamming_key(t0, t1, q0, q1);
    if (rule == ORBS_RULE_INIT && st <= dist) return KEY_NONE;                      // `if(vMatchedDistance[i2]<=dist) continue;`
    return (dist << 16) | (BK ? (uint32_t)S.rank16[j] : (uint32_t)j);
}

// level buckets a query's level range can hold entries in (Frame::GetFeaturesInArea: no check when both are -1; with a check and
// maxLevel < minLevel nothing passes)
// Derived from orbf::level_passes itself (the test in_window applies), so the two cannot drift: bucket k files the octaves
// clamp(octave, 0, BK_LEVELS - 1) == k, i.e. bucket 0 everything <= 0 and the last bucket everything >= BK_LEVELS - 1; it has to be
// visited iff some octave it may hold passes, and for the two open-ended buckets the octave most likely to pass is the range end
// itself.  The passing octaves form an interval, so the visited buckets are lb0 .. lb1 (empty: lb1 < lb0).
__device__ __forceinline__ void bk_levels(int minLevel, int maxLevel, int& lb0, int& lb1) {
    lb0 = BK_LEVELS;
    lb1 = -1;
#pragma unroll
    for (int k = 0; k < BK_LEVELS; ++k) {
        const int rep = k == 0 ? min(0, minLevel) : k == BK_LEVELS - 1 ? max(BK_LEVELS - 1, maxLevel) : k;
        if (orbf::level_passes(rep, minLevel, maxLevel)) { lb0 = min(lb0, k); lb1 = max(lb1, k); }
    }
    if (lb1 < lb0) { lb0 = 0; lb1 = -1; }
}

// packed 16-bit counters in LDS (two per dword): add v to entry idx, return its old value (entries stay below 65536)
__device__ __forceinline__ uint32_t add16(uint16_t* base, int idx, uint32_t v) {
    const int sh = (idx & 1) * 16;
    const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(base) + (idx >> 1), v << sh);
    return (old >> sh) & 0xFFFFu;
}

// The whole wave scans ONE query's window (8 grid columns at a time, 8 lanes per column) under the current claim state and
// reduces to the two smallest keys: the in-order fallback for queries whose speculative result was overtaken by a claim.
template <bool BK>
__device__ __forceinline__ void scan_wave(const Staged& S, int rule, bool list_mode, int lane, int x0, int x1, int y0, int y1, float x, float y, float r,
                                          int minLevel, int maxLevel, const uint4& q0, const uint4& q1, const EpiLine& E, uint32_t& k1, uint32_t& k2) {
    uint32_t a1 = KEY_NONE, a2 = KEY_NONE;
    if (BK) {                                          // (level bucket, coarse column) pairs, 8 at a time, 8 lanes per run
        int lb0, lb1;
        bk_levels(minLevel, maxLevel, lb0, lb1);
        const int cx0 = x0 >> 1, cy0 = y0 >> 1, cy1 = y1 >> 1, ncol = (x1 >> 1) - cx0 + 1;
        const int npair = x1 >= x0 ? (lb1 - lb0 + 1) * ncol : 0;
        for (int p0 = 0; p0 < npair; p0 += 8) {
            const int pi = p0 + (lane >> 3);
            int j = 0, jend = 0;
            if (pi < npair) {
                const int lv = pi / ncol, base = (lb0 + lv) * BK_CELLS + (cx0 + pi - lv * ncol) * BK_ROWS;
                j = S.off16[base + cy0] + (lane & 7);
                jend = S.off16[base + cy1 + 1];
            }
            for (; j < jend; j += 8) {
                bool inwin;
                const uint32_t key = candidate_key<true>(S, rule, j, x, y, r, minLevel, maxLevel, q0, q1, E, inwin);
                a2 = min(a2, max(a1, key));
                a1 = min(a1, key);
            }
        }
        x1 = x0 - 1;
    }
    if (list_mode) {                                   // one explicit run [y0, y1): all 64 lanes stride it
        for (int j = y0 + lane; j < y1; j += 64) {
            bool inwin;
            const uint32_t key = candidate_key<false>(S, rule, j, x, y, r, minLevel, maxLevel, q0, q1, E, inwin);
            if (rule == ORBS_RULE_TRIANGULATION) a2 = min(a2, (key & 0x8000u) ? key : KEY_NONE);      // a2 = best candidate ON the line
            else a2 = min(a2, max(a1, key));
            a1 = min(a1, key);
        }
        x1 = x0 - 1;
    }
    for (int cx = x0; cx <= x1; cx += 8) {
        const int col = cx + (lane >> 3);
        int j = 0, jend = 0;
        if (col <= x1) {
            j = S.off16[col * ORBF_GRID_ROWS + y0] + (lane & 7);
            jend = S.off16[col * ORBF_GRID_ROWS + y1 + 1];
        }
        for (; j < jend; j += 8) {
            bool inwin;
            const uint32_t key = candidate_key<false>(S, rule, j, x, y, r, minLevel, maxLevel, q0, q1, E, inwin);
            a2 = min(a2, max(a1, key));          // branch-free two-smallest update (a1 <= a2)
            a1 = min(a1, key);
        }
    }
    k1 = wave_min_u32(a1);
    if (rule == ORBS_RULE_TRIANGULATION) { k2 = wave_min_u32(a2); return; }
    if (a1 == k1) a1 = a2;
    k2 = wave_min_u32(a1);
}

__device__ __forceinline__ uint4 lane_u4(const uint4& v, int l) {
    uint4 r;
    r.x = (uint32_t)__builtin_amdgcn_readlane((int)v.x, l); r.y = (uint32_t)__builtin_amdgcn_readlane((int)v.y, l);
    r.z = (uint32_t)__builtin_amdgcn_readlane((int)v.z, l); r.w = (uint32_t)__builtin_amdgcn_readlane((int)v.w, l);
    return r;
}

// accept rule of the four searches (wave-uniform or per lane alike)
__device__ __forceinline__ bool accept_rule(const orbs_params& prm, int bestDist, int bestDist2, int bestLevel, int bestLevel2) {
    if (prm.rule == ORBS_RULE_TRIANGULATION)   // src/ORBmatcher.cc:927-936: bestDist2 = the best candidate ON the epipolar line (INT_MAX none),
        return bestDist2 != INT_MAX && bestDist2 <= 2 * bestDist;      // taken when it is within DistTh = round(2*BestDist)
    if (prm.rule == ORBS_RULE_MAPPOINTS)       // src/ORBmatcher.cc:114-121
        return bestDist <= prm.th && !(bestLevel == bestLevel2 && (float)bestDist > prm.ratio * (float)bestDist2);
    if (prm.rule == ORBS_RULE_WINDOW)          // :476, :585
        return (float)bestDist <= (float)bestDist2 * prm.ratio && bestDist <= prm.th;
    if (prm.rule == ORBS_RULE_BEST || prm.rule == ORBS_RULE_FREE)            // :1583, :1113
        return bestDist <= prm.th;
    return bestDist <= prm.th && (float)bestDist < prm.ratio * (float)bestDist2;      // :652-654 (INIT), :224-226 (BOW)
}

// Per group of 256 queries (one per thread, four waves):
//  (1) speculative, parallel: every lane scans its own query's window against the claim state as of the group start and
//      keeps its NK (six) smallest keys with the train index / octave of each.
//  (2) commit, wave after wave, in query order.  A claim can only REMOVE candidates (claims are never released during the
//      scan; SearchForInitialization's matched distance only decreases), so at any moment a query's exact best / second are
//      the first two STILL-ADMISSIBLE entries of its list — as long as two survive or the list was never full.  A commit
//      round therefore refreshes every lane's alive mask (one bit per key) from the state array, lets every accepting lane post its
//      claim (LDS atomicMax of stamp | lane: the earliest lane wins), and finalises ALL lanes up to the first one that an
//      earlier lane of the same round would affect (its best or second was just claimed) or whose list ran dry.  Affected
//      lanes simply go again next round under the refreshed state; a dry lane has the whole wave rescan its window.
//      With little contention a wave commits its 64 queries in two or three rounds.
// GROUP = queries scanned speculatively at a time = threads of the workgroup.  256 (four waves) for launches with many problems; 1024 for launches
// with FEW problems (round 6: the one-problem calls of orb_slam_amd/cpp/ORBmatcher.cc) — a lone four-wave workgroup has one wave per SIMD and nothing to
// hide its LDS round trips behind; sixteen waves scan a 1000-query frame in one pass.  Results do not depend on it (speculation + in-order commit).
constexpr int GROUP_BATCH = 256, GROUP_WIDE = 1024;

// NK = keys a query keeps from its speculative scan.  A list that runs dry (fewer than two of its entries still admissible) makes the committing wave
// rescan that query's window, alone, while the other waves wait: with four keys the 1000 x 1000, r = 100 problems of the one-problem calls paid 27-44
// rescans and 59-98 commit rounds (121-260 us of a 176-373 us kernel); with six, 4-24 and 31-49 (WindowSearch 176 -> 98 us, SearchForInitialization
// 373 -> 213, SearchByProjection(last frame) 212 -> 103; eight: no further gain but for the initialisation search).  The batched form (512 problems) pays the longer
// insertion where windows hold few real candidates (independent S-blocks frames, r = 15: 0.478 -> 0.497 ms per 512 frames) and gains where they hold many:
// r = 100 1.457 -> 1.345 ms, the correlated S-warp stream 1.105 -> 1.055 (r = 15) and 2.21 -> 1.70 (r = 40) - a camera's case: six there as well.
#ifndef ORBS_LIST_KEYS_WIDE
#define ORBS_LIST_KEYS_WIDE 6
#endif
#ifndef ORBS_LIST_KEYS_BATCH
#define ORBS_LIST_KEYS_BATCH 6
#endif
template <bool BK, int GROUP>
__global__ __launch_bounds__(GROUP) void k_window_search(orbf_bounds b, orbs_params prm, Args a) {
    constexpr int NK = GROUP == GROUP_WIDE ? ORBS_LIST_KEYS_WIDE : ORBS_LIST_KEYS_BATCH;
    static_assert(NK >= 2 && NK <= 15, "alive / on_line are bit masks; TRIANGULATION keeps bit 15 of a key for the line test");
    extern __shared__ __align__(16) uint8_t lds[];
    const Layout L = make_layout(a.cap, a.qcap, a.desc_in_lds != 0, BK);
    uint16_t* off16 = (uint16_t*)(lds + L.off16);
    float* tx = (float*)(lds + L.tx);
    float* ty = (float*)(lds + L.ty);
    float* tang = (float*)(lds + L.tang);
    uint32_t* tmeta = (uint32_t*)(lds + L.tmeta);
    uint4* tdesc = (uint4*)(lds + L.tdesc);
    const bool desc_lds = a.desc_in_lds != 0;
    uint16_t* state = (uint16_t*)(lds + L.state);
    uint32_t* claim_by = (uint32_t*)(lds + L.claim);
    int16_t* t2q = (int16_t*)(lds + L.t2q);
    int16_t* q2t = (int16_t*)(lds + L.q2t);
    uint8_t* binv = lds + L.binv;
    int* hist = (int*)(lds + L.hist);
    float* epi_thr = (float*)(lds + L.epi);
    uint16_t* rank16 = (uint16_t*)(lds + L.rank16);      // BK only
    uint16_t* r2s = (uint16_t*)(lds + L.r2s);            // BK only: CSR position -> LDS slot
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int nt = min(a.nt[p], a.cap), nq = min(a.nq[p], a.qcap);
    const int rule = prm.rule;
    const size_t tb = (size_t)p * a.cap, qb = (size_t)p * a.qcap;
    const Staged S{off16, tx, ty, tmeta, tdesc, state, epi_thr, a.desc + tb * 32, desc_lds, rank16};
    const bool list_mode = !BK && a.qrange != nullptr;
    const int32_t* coff = a.cell_off + (size_t)p * (ORBF_GRID_CELLS + 1);
    const int32_t* cfeat = a.cell_feat + tb;
    const orbx_keypoint* kps = a.kps_un + tb;

    // ---- stage the train frame in LDS, in grid (or list) order
    if (!list_mode) for (int i = tid; i <= ORBF_GRID_CELLS; i += GROUP) off16[i] = (uint16_t)min(coff[i], a.cap);
    const int m = list_mode ? max(min(a.nlist[p], a.cap), 0) : min(coff[ORBF_GRID_CELLS], a.cap);
    if (BK) {
        // counting sort of the CSR entries into (level bucket, coarse column, coarse row) order, all in LDS:
        //  (a) every entry's bucket, from the cell the caller's CSR files it under (r2s[] holds it for now);
        //  (b) counts: entry b + 2 of the table; an inclusive prefix sum turns entry b + 1 into bucket b's start;
        //  (c) scatter with a fetch-add on entry b + 1, which ends as bucket b's end = bucket b + 1's start: the table is then
        //      T[x] = start of bucket x for x = 0 .. BK_N (T[BK_N] = m), what the scans index.  Slots inside a bucket are handed out
        //      in atomic order; nothing depends on it (keys carry the CSR position, rank16).
        __syncthreads();
        for (int c = tid; c < ORBF_GRID_CELLS; c += GROUP) {
            const int cx = c / ORBF_GRID_ROWS, cy = c - cx * ORBF_GRID_ROWS;
            const int cc = (cx >> 1) * BK_ROWS + (cy >> 1);
            for (int j = off16[c]; j < min((int)off16[c + 1], m); ++j) {
                const int f = min(max(cfeat[j], 0), max(nt - 1, 0));
                r2s[j] = (uint16_t)(min(max(kps[f].octave, 0), BK_LEVELS - 1) * BK_CELLS + cc);
            }
        }
        __syncthreads();
        for (int i = tid; i < (BK_N + 2 + 1) / 2; i += GROUP) reinterpret_cast<uint32_t*>(off16)[i] = 0u;
        __syncthreads();
        for (int j = tid; j < m; j += GROUP) (void)add16(off16, r2s[j] + 2, 1u);
        __syncthreads();
        {
            constexpr int CH = (BK_N + 2 + GROUP - 1) / GROUP;
            int sum = 0;
            for (int k = 0; k < CH; ++k) { const int i = tid * CH + k; if (i < BK_N + 2) sum += off16[i]; }
            int incl = sum;
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
            if (lane == 63) hist[wave] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wave; ++w) run += hist[w];
            for (int k = 0; k < CH; ++k) { const int i = tid * CH + k; if (i < BK_N + 2) { run += off16[i]; off16[i] = (uint16_t)run; } }
        }
        __syncthreads();
        for (int j = tid; j < m; j += GROUP) {
            const int slot = (int)add16(off16, r2s[j] + 1, 1u);
            const int f = min(max(cfeat[j], 0), max(nt - 1, 0));
            const orbx_keypoint kp = kps[f];
            tx[slot] = kp.x;
            ty[slot] = kp.y;
            tang[slot] = kp.angle;
            tmeta[slot] = (uint32_t)f | ((uint32_t)kp.octave << 16);
            rank16[slot] = (uint16_t)j;
            r2s[j] = (uint16_t)slot;
        }
        __syncthreads();
        if (desc_lds)
            for (int j = tid; j < m; j += GROUP) {
                const uint4* d = (const uint4*)(a.desc + (tb + (tmeta[j] & 0xFFFFu)) * 32);
                tdesc[2 * j] = d[0];
                tdesc[2 * j + 1] = d[1];
            }
        __syncthreads();                                       // hist[] (scan scratch above) is zeroed below
    } else
    for (int j = tid; j < m; j += GROUP) {
        const int f = min(max(cfeat[j], 0), max(nt - 1, 0));          // a malformed list must not index outside the frame
        const orbx_keypoint kp = kps[f];
        tx[j] = kp.x;
        ty[j] = kp.y;
        tang[j] = kp.angle;
        tmeta[j] = (uint32_t)f | ((uint32_t)kp.octave << 16);
        if (desc_lds) {
            const uint4* d = (const uint4*)(a.desc + (tb + f) * 32);
            tdesc[2 * j] = d[0];
            tdesc[2 * j + 1] = d[1];
        }
    }
    for (int i = tid; i < nt; i += GROUP) {
        state[i] = rule == ORBS_RULE_INIT ? (uint16_t)0xFFFF : (uint16_t)((a.claimed && a.claimed[tb + i]) ? 1 : 0);
        t2q[i] = -1;
        claim_by[i] = 0;
    }
    for (int i = tid; i < nq; i += GROUP) q2t[i] = -1;
    const bool tri = rule == ORBS_RULE_TRIANGULATION;
    const bool bin_by_query = rule == ORBS_RULE_INIT || tri;       // rotHist holds i1 / idx1 there, the train index elsewhere
    const uint32_t pos_mask = tri ? 0x7FFFu : 0xFFFFu;             // list position inside a key
    const int nbin = bin_by_query ? nq : nt;
    for (int i = tid; i < nbin; i += GROUP) binv[i] = 255;
    if (tid < 32) hist[tid] = 0;
#pragma unroll
    for (int i = 0; i < ORBS_MAX_LEVELS; ++i) if (tid == i) epi_thr[i] = a.epi_thr[i];
    float F[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tri) for (int i = 0; i < 9; ++i) F[i] = a.F12[(size_t)p * 9 + i];
    __syncthreads();

    const bool rot_on = (prm.check_orientation & 1) != 0 && rule != ORBS_RULE_MAPPOINTS && rule != ORBS_RULE_FREE;
    const bool claims = rule != ORBS_RULE_FREE;              // FREE: every query independent, nothing is ever claimed

    for (int q0 = 0, group = 0; q0 < nq; q0 += GROUP, ++group) {
        // ---- one query per thread
        const int qi = q0 + tid;
        float qx = 0.f, qy = 0.f, qr = 0.f, qa = 0.f;
        int ql0 = 0, ql1 = 0, qv = 0;
        uint4 qd0 = make_uint4(0, 0, 0, 0), qd1 = qd0;
        if (qi < nq) {
            const size_t qs = qb + (a.qindex ? (size_t)a.qindex[qb + qi] : (size_t)qi);      // slot of the query-side arrays
            if (!list_mode) {
                qx = a.qxyr[(qb + qi) * 3]; qy = a.qxyr[(qb + qi) * 3 + 1]; qr = a.qxyr[(qb + qi) * 3 + 2];
                ql0 = a.qlev[(qb + qi) * 2]; ql1 = a.qlev[(qb + qi) * 2 + 1];
            }
            qv = a.qvalid ? (a.qvalid[qs] != 0) : 1;
            if (a.qangle) qa = a.qangle[qs];
            if (tri) { const orbx_keypoint kp1 = a.qkps[qs]; qx = kp1.x; qy = kp1.y; qa = kp1.angle; }
            const uint4* d = (const uint4*)(a.qdesc + qs * 32);
            qd0 = d[0];
            qd1 = d[1];
        }
        // ---- (1) speculative scan of the lane's own query: the NK smallest keys e[0] <= e[1] <= ...
        int wx0 = 0, wx1 = -1, wy0 = 0, wy1 = 0;
        const EpiLine E = epi_line(qx, qy, F, prm.th);
        if (list_mode) {
            // one run of list positions; no geometric test (an infinite box on every level)
            qr = INFINITY; ql0 = -1; ql1 = -1;
            if (qi < nq && qv) { wy0 = min(max(a.qrange[(qb + qi) * 2], 0), m); wy1 = min(max(a.qrange[(qb + qi) * 2 + 1], wy0), m); wx1 = 0; }
        } else if (qv && !orbf::window_cells(b, qx, qy, qr, &wx0, &wx1, &wy0, &wy1)) { wx0 = 0; wx1 = -1; }
        if (!qv) wx1 = -1;
        uint32_t e[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) e[i] = KEY_NONE;
        bool any = false;
#ifdef ORBS_PROBE_NO_SCAN                                 // (timing probe, never in the product: results are wrong) no speculative scan
        wx1 = -1;
#endif
        auto take = [&](uint32_t t) {                    // insertion into the NK smallest keys
#pragma unroll
            for (int i = 0; i < NK - 1; ++i) { const uint32_t lo = min(e[i], t); t = max(e[i], t); e[i] = lo; }
            e[NK - 1] = min(e[NK - 1], t);
        };
        if (BK) {
            int lb0, lb1;
            bk_levels(ql0, ql1, lb0, lb1);
            const int cy0 = wy0 >> 1, cy1 = wy1 >> 1;
            for (int lv = lb0; lv <= lb1; ++lv)
                for (int cc = wx0 >> 1; cc <= (wx1 >> 1) && wx1 >= wx0; ++cc) {
                    const int base = lv * BK_CELLS + cc * BK_ROWS;
                    const int jend = off16[base + cy1 + 1];
                    for (int j = off16[base + cy0]; j < jend; ++j) {
                        bool inwin;
                        const uint32_t t = candidate_key<true>(S, rule, j, qx, qy, qr, ql0, ql1, qd0, qd1, E, inwin);
                        any |= inwin;
                        take(t);
                    }
                }
        } else
        for (int col = wx0; col <= wx1; ++col) {
            const int jend = list_mode ? wy1 : off16[col * ORBF_GRID_ROWS + wy1 + 1];
            for (int j = list_mode ? wy0 : off16[col * ORBF_GRID_ROWS + wy0]; j < jend; ++j) {
                bool inwin;
                const uint32_t t = candidate_key<false>(S, rule, j, qx, qy, qr, ql0, ql1, qd0, qd1, E, inwin);
                any |= inwin;
                take(t);
            }
        }
        // train index | octave << 16 of the four entries
        auto slot_of = [&](uint32_t key) -> uint32_t { return BK ? (uint32_t)r2s[key & 0xFFFFu] : (key & pos_mask); };   // the staged entry a key names
        uint32_t f[NK], on_line = 0u;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            f[i] = e[i] != KEY_NONE ? tmeta[slot_of(e[i])] : 0u;
            if (tri) on_line |= ((e[i] >> 15) & 1u) << i;
        }
        const bool full = e[NK - 1] != KEY_NONE;       // one more candidate may exist
        int my_best = -1, my_second = -1;
        __syncthreads();

        // ---- (2) commit: wave after wave, rounds inside a wave
#ifdef ORBS_PROBE_NO_COMMIT                               // (timing probe, never in the product: results are wrong) no in-order commit
        { uint32_t x = 0; for (int i = 0; i < NK; ++i) x ^= e[i] ^ f[i]; if (x == 0xDEADBEEFu) hist[30] = 1; }      // keeps the scan alive
        any = false;
#endif
        for (int w = 0; w < GROUP / 64; ++w) {
            if (wave == w) {
                int cursor = 0;
                for (int round = 0;; ++round) {
                    // alive = entries still admissible under the current claim state
                    uint32_t alive = 0;
#pragma unroll
                    for (int i = 0; i < NK; ++i) {
                        const uint32_t st = state[f[i] & 0xFFFFu];
                        if (e[i] != KEY_NONE && (rule == ORBS_RULE_INIT ? st > (e[i] >> 16) : st == 0)) alive |= 1u << i;
                    }
                    // current best / second = first two alive entries
                    // (TRIANGULATION: "second" = the first alive entry ON the epipolar line — possibly the best itself)
                    const int ib = alive ? (__ffs((int)alive) - 1) : NK;
                    const uint32_t rest = tri ? (alive & on_line) : (alive & (alive - 1u));      // (without its lowest set bit)
                    const int is = rest ? (__ffs((int)rest) - 1) : NK;
                    uint32_t kb = KEY_NONE, mb = 0u, ks = KEY_NONE, ms = 0u;
#pragma unroll
                    for (int i = 0; i < NK; ++i) {
                        if (ib == i) { kb = e[i]; mb = f[i]; }
                        if (is == i) { ks = e[i]; ms = f[i]; }
                    }
                    const int vBest = kb != KEY_NONE ? (int)(kb >> 16) : INT_MAX, vBest2 = ks != KEY_NONE ? (int)(ks >> 16) : INT_MAX;
                    const int vLev = kb != KEY_NONE ? (int)(mb >> 16) : -1, vLev2 = ks != KEY_NONE ? (int)(ms >> 16) : -1;
                    const bool active = any && lane >= cursor;
                    // list exhausted: the exact answer needs a rescan.  TRIANGULATION: no alive on-line entry left, and a fifth
                    // candidate could still lie within DistTh (it has distance >= the last entry's)
                    const bool dry = tri ? (full && is == NK && (ib == NK || (int)(e[NK - 1] >> 16) <= 2 * vBest)) : (full && __popc(alive) < 2);
                    const bool vAccept = !dry && accept_rule(prm, vBest, vBest2, vLev, vLev2);
                    // accepting lanes post their claim; the earliest lane of this round wins the slot
                    const uint32_t stamp = (uint32_t)((group * (GROUP / 64) + w) * 128 + round + 1);
                    const uint32_t bIdx = mb & 0xFFFFu, sIdx = ms & 0xFFFFu;
                    const uint32_t cIdx = tri ? sIdx : bIdx, kc = tri ? ks : kb;        // what an accepting query takes
                    if (claims && active && vAccept) atomicMax(&claim_by[cIdx], (stamp << 6) | (uint32_t)(63 - lane));
                    bool affected = false;
                    if (claims && active && kb != KEY_NONE) { const uint32_t c = claim_by[bIdx]; affected |= (c >> 6) == stamp && (int)(63u - (c & 63u)) < lane; }
                    // (measured and dropped, NOTES 11.6: letting an ACCEPTING query keep going when only its second was taken — WINDOW / BOW / INIT accept
                    //  monotonically in the second — saved 3-7 rounds of 28-49 and no measurable time)
                    if (claims && active && ks != KEY_NONE) { const uint32_t c = claim_by[sIdx]; affected |= (c >> 6) == stamp && (int)(63u - (c & 63u)) < lane; }
                    const unsigned long long stop = __ballot(active && (affected || dry));
                    const int F = stop ? (__ffsll((long long)stop) - 1) : 64;
                    // every lane before F is final: record, and commit its own claim
                    if (active && lane < F) {
                        my_best = vBest;
                        my_second = tri ? (vAccept ? vBest2 : INT_MAX) : vBest2;
                        if (vAccept) {
                            const int q = q0 + w * 64 + lane;
                            int bin = 255;
                            if (rot_on) bin = rot_bin(qa, tang[slot_of(kc)]);
                            if (rule == ORBS_RULE_INIT) {
                                const int prev = t2q[bIdx];
                                if (prev >= 0) q2t[prev] = -1;                 // vnMatches12[vnMatches21[bestIdx2]] = -1
                                state[bIdx] = (uint16_t)vBest;                 // vMatchedDistance[bestIdx2] = bestDist
                                binv[q] = (uint8_t)bin;                        // rotHist[bin].push_back(i1)
                            } else if (claims) {
                                state[cIdx] = 1;
                                binv[tri ? (uint32_t)q : cIdx] = (uint8_t)bin; // rotHist[bin].push_back(bestIdx2)  (TRIANGULATION: idx1)
                            }
                            q2t[q] = (int16_t)cIdx;
                            if (claims) t2q[cIdx] = (int16_t)q;
                        }
                    }
                    cursor = F;
#ifdef ORBS_PROBE_ROUNDS                                  // (counting probe, never in the product) nmatches = rounds + 1000 * rescans
                    if (lane == 0) hist[30] += 1 + (F < 64 && __builtin_amdgcn_readlane((int)dry, F) ? 1000 : 0);
#endif
                    if (F >= 64) break;
                    if (__builtin_amdgcn_readlane((int)dry, F)) {
                        // the whole wave rescans query F under the current state
                        uint32_t k1, k2;
                        scan_wave<BK>(S, rule, list_mode, lane, __builtin_amdgcn_readlane(wx0, F), __builtin_amdgcn_readlane(wx1, F), __builtin_amdgcn_readlane(wy0, F),
                                  __builtin_amdgcn_readlane(wy1, F), lane_f(qx, F), lane_f(qy, F), lane_f(qr, F), __builtin_amdgcn_readlane(ql0, F),
                                  __builtin_amdgcn_readlane(ql1, F), lane_u4(qd0, F), lane_u4(qd1, F),
                                  EpiLine{lane_f(E.a, F), lane_f(E.b, F), lane_f(E.c, F), lane_f(E.den, F), prm.th}, k1, k2);
                        const uint32_t m1 = k1 != KEY_NONE ? tmeta[slot_of(k1)] : 0u, m2 = k2 != KEY_NONE ? tmeta[slot_of(k2)] : 0u;
                        const int bestDist = k1 != KEY_NONE ? (int)(k1 >> 16) : INT_MAX, bestDist2 = k2 != KEY_NONE ? (int)(k2 >> 16) : INT_MAX;
                        const int bestIdx = (int)((tri ? m2 : m1) & 0xFFFFu);
                        const bool accept = accept_rule(prm, bestDist, bestDist2, k1 != KEY_NONE ? (int)(m1 >> 16) : -1, k2 != KEY_NONE ? (int)(m2 >> 16) : -1);
                        if (lane == F) { my_best = bestDist; my_second = tri ? (accept ? bestDist2 : INT_MAX) : bestDist2; }
                        if (accept) {
                            const int q = q0 + w * 64 + F;
                            int bin = 255;
                            if (rot_on) bin = rot_bin(lane_f(qa, F), tang[slot_of(tri ? k2 : k1)]);
                            if (lane == 0) {
                                if (rule == ORBS_RULE_INIT) {
                                    const int prev = t2q[bestIdx];
                                    if (prev >= 0) q2t[prev] = -1;
                                    state[bestIdx] = (uint16_t)bestDist;
                                    binv[q] = (uint8_t)bin;
                                } else {
                                    state[bestIdx] = 1;
                                    binv[tri ? q : bestIdx] = (uint8_t)bin;
                                }
                                q2t[q] = (int16_t)bestIdx;
                                t2q[bestIdx] = (int16_t)q;
                            }
                        }
                        cursor = F + 1;
                    }
                }
            }
            __syncthreads();
        }
        if (qi < nq) {
            if (a.best) a.best[qb + qi] = my_best;
            if (a.second) a.second[qb + qi] = my_second;
        }
    }

    // ---- rotation consistency (the rotHist blocks + ComputeThreeMaxima)
    if (rot_on) {
        for (int i = tid; i < nbin; i += GROUP) { const int bn = binv[i]; if (bn != 255) atomicAdd(&hist[bn], 1); }
        __syncthreads();
        int i1, i2, i3;
        three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
        for (int i = tid; i < nbin; i += GROUP) {
            const int bn = binv[i];
            if (bn == 255 || bn == i1 || bn == i2 || bn == i3) continue;
            if (bin_by_query) {
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
    for (int i = tid; i < nq; i += GROUP) { const int t = q2t[i]; a.q2t[qb + i] = t; cnt += t >= 0; }
    for (int i = tid; i < nt; i += GROUP) a.t2q[tb + i] = t2q[i];
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
    __syncthreads();
    if (lane == 0) atomicAdd(&hist[31], cnt);           // hist[31] is never a rotation bin (HISTO_LENGTH = 30): zero since the start
    __syncthreads();
#ifdef ORBS_PROBE_ROUNDS
    if (tid == 0) a.nmatches[p] = hist[30];
#else
    if (tid == 0) a.nmatches[p] = hist[31];
#endif
}


// SearchByBoW's merge walk as data: one thread per node of the query frame's FeatureVector, binary search of the node id in
// the train frame's (both ascending), then the node's query positions get the train run.
__global__ __launch_bounds__(256) void k_bow_ranges(const uint32_t* __restrict__ fvq_node, const int32_t* __restrict__ fvq_off, const int32_t* __restrict__ nfv_q,
                                                   const uint32_t* __restrict__ fvt_node, const int32_t* __restrict__ fvt_off, const int32_t* __restrict__ nfv_t,
                                                   int cap, int32_t* __restrict__ qrange, int32_t* __restrict__ nq) {
    const int p = blockIdx.y;
    const int nn = min(nfv_q[p], cap), nt = min(nfv_t[p], cap);
    const uint32_t* qn = fvq_node + (size_t)p * cap;
    const int32_t* qo = fvq_off + (size_t)p * (cap + 1);
    const uint32_t* tn = fvt_node + (size_t)p * cap;
    const int32_t* to = fvt_off + (size_t)p * (cap + 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) nq[p] = nn > 0 ? qo[nn] : 0;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < nn; s += gridDim.x * 256) {
        const uint32_t node = qn[s];
        int lo = 0, hi = nt;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (tn[mid] < node) lo = mid + 1; else hi = mid; }
        int r0 = 0, r1 = 0;
        if (lo < nt && tn[lo] == node) { r0 = to[lo]; r1 = to[lo + 1]; }
        for (int j = qo[s]; j < qo[s + 1]; j++) { qrange[((size_t)p * cap + j) * 2] = r0; qrange[((size_t)p * cap + j) * 2 + 1] = r1; }
    }
}

// The "check agreement" tail of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1486-1505): keep i1 -> idx2 only when idx2 -> i1.
__global__ __launch_bounds__(256) void k_agreement(const int32_t* __restrict__ m12, const int32_t* __restrict__ n1, int cap1,
                                                  const int32_t* __restrict__ m21, const int32_t* __restrict__ n2, int cap2,
                                                  int32_t* __restrict__ out12, int32_t* __restrict__ nfound) {
    __shared__ int total;
    const int p = blockIdx.x, N1 = min(n1[p], cap1), N2 = min(n2[p], cap2);
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int cnt = 0;
    for (int i1 = threadIdx.x; i1 < N1; i1 += 256) {
        const int idx2 = m12[(size_t)p * cap1 + i1];
        const bool ok = idx2 >= 0 && idx2 < N2 && m21[(size_t)p * cap2 + idx2] == i1;
        out12[(size_t)p * cap1 + i1] = ok ? idx2 : -1;
        cnt += ok;
    }
    for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&total, cnt);
    __syncthreads();
    if (threadIdx.x == 0) nfound[p] = total;
}

}  // namespace orbs

// the kernel may use up to the whole 160 KiB of LDS: raise the dynamic-LDS limit of the current device
static std::atomic<int> g_wide_max{-2};          // launches of up to this many problems take the 1024-thread form (-2: ORBS_WIDE_MAX in the environment, else 64)
static bool orbs_wide(int nproblems) {
    int m = g_wide_max.load(std::memory_order_relaxed);
    if (m == -2) { const char* e = getenv("ORBS_WIDE_MAX"); m = e ? atoi(e) : 64; g_wide_max.store(m, std::memory_order_relaxed); }
    return nproblems <= m;
}
template <bool BK>
static int orbs_launch(int nproblems, size_t lds, hipStream_t stream, const orbf_bounds& b, const orbs_params& prm, const orbs::Args& a) {
    // the kernel may use up to the whole 160 KiB of LDS: raise the dynamic-LDS limit (the attribute belongs to the CURRENT device: set on every launch
    // path — a cheap runtime call —, not once per process)
    if (orbs_wide(nproblems)) {
        if (hipFuncSetAttribute((const void*)orbs::k_window_search<BK, orbs::GROUP_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ORBX_ERR_DEVICE;
        hipLaunchKernelGGL((orbs::k_window_search<BK, orbs::GROUP_WIDE>), dim3(nproblems), dim3(orbs::GROUP_WIDE), lds, stream, b, prm, a);
    } else {
        if (hipFuncSetAttribute((const void*)orbs::k_window_search<BK, orbs::GROUP_BATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ORBX_ERR_DEVICE;
        hipLaunchKernelGGL((orbs::k_window_search<BK, orbs::GROUP_BATCH>), dim3(nproblems), dim3(orbs::GROUP_BATCH), lds, stream, b, prm, a);
    }
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

extern "C" {

// ORBS_BUCKETS=0 in the environment (the process default): the plain 64 x 48 CSR scan for the grid searches too;
// orbs_debug_set_buckets() overrides it at run time (the parity tests run the same problems through both forms).
static std::atomic<int> g_buckets{-1};           // -1: environment default, 0: plain CSR scan, 1: bucketed index where it fits
static bool orbs_use_buckets() {
    static const bool env_default = [] { const char* e = getenv("ORBS_BUCKETS"); return !(e && e[0] == '0'); }();
    const int f = g_buckets.load(std::memory_order_relaxed);
    return f < 0 ? env_default : f != 0;
}

int orbs_debug_set_wide_max(int nproblems) {
    if (nproblems < -2) return ORBX_ERR_ARG;
    g_wide_max.store(nproblems, std::memory_order_relaxed);      // -2: back to the process default
    return ORBX_OK;
}

int orbs_debug_set_buckets(int mode) {
    if (mode < -1 || mode > 1) return ORBX_ERR_ARG;
    g_buckets.store(mode, std::memory_order_relaxed);
    return ORBX_OK;
}

// the layout a problem of this size gets: descriptors staged when that fits the 160 KiB of a workgroup
static size_t orbs_choose_layout(int cap, int qcap, int& desc_in_lds) {
    const size_t with = orbs::make_layout(cap, qcap, true).total;
    desc_in_lds = with <= 160 * 1024;
    return desc_in_lds ? with : orbs::make_layout(cap, qcap, false).total;
}

// what a grid search of this size actually launches with: the bucketed layout (descriptors + level buckets) where it fits
size_t orbs_lds_bytes(int cap, int qcap) {
    if (cap < 1 || qcap < 1) return 0;
    int d;
    const size_t plain = orbs_choose_layout(cap, qcap, d);
    const size_t bk = orbs::make_layout(cap, qcap, true, true).total;
    return d && bk <= 160 * 1024 && orbs_use_buckets() ? bk : plain;
}

float orbs_epipolar_bound(float sigma2) {
    const double T = 3.84 * (double)sigma2;          // `dsqr<3.84*pKF2->GetSigma2(kp2.octave)`: float < double, compared in double
    float t = (float)T;
    if ((double)t < T) t = nextafterf(t, INFINITY);   // the smallest float >= T: for a float d, d < t  <=>  (double)d < T
    return t;
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
    if (prm->rule < ORBS_RULE_MAPPOINTS || prm->rule > ORBS_RULE_FREE) return ORBX_ERR_ARG;
    if (nproblems == 0) return ORBX_OK;
    if (!d_kps_un || !d_desc || !d_cell_off || !d_cell_feat || !d_nt || !d_qxyr || !d_qlev || !d_qdesc || !d_nq || !d_q2t || !d_t2q || !d_nmatches)
        return ORBX_ERR_ARG;
    if (prm->check_orientation && prm->rule != ORBS_RULE_MAPPOINTS && !d_qangle) return ORBX_ERR_ARG;
    int desc_in_lds = 1;
    size_t lds = orbs_choose_layout(cap, qcap, desc_in_lds);
    if (lds > 160 * 1024) return ORBX_ERR_CAPACITY;
    // the level-bucketed index (see BK_LEVELS) wherever the frame still fits with its descriptors staged
    const size_t lds_bk = orbs::make_layout(cap, qcap, true, true).total;
    const bool bucketed = desc_in_lds && lds_bk <= 160 * 1024 && orbs_use_buckets();
    if (bucketed) lds = lds_bk;
    orbs_params prm2 = *prm;
    prm2.check_orientation = prm->check_orientation ? 1 : 0;
    orbs::Args a{d_kps_un, d_desc, d_cell_off, d_cell_feat, d_nt, d_claimed, d_qxyr, d_qlev, d_qdesc, d_qangle, d_qvalid, d_nq,
                 d_q2t, d_t2q, d_best, d_second, d_nmatches, cap, qcap, desc_in_lds, bucketed ? 1 : 0, nullptr, nullptr, nullptr, nullptr, nullptr, {}};
    return bucketed ? orbs_launch<true>(nproblems, lds, (hipStream_t)stream, *b, prm2, a) : orbs_launch<false>(nproblems, lds, (hipStream_t)stream, *b, prm2, a);
}


int orbs_list_search_batch_device(const orbs_params* prm, const orbx_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_list, const int32_t* d_nlist,
                                  const int32_t* d_nt, int cap, const uint8_t* d_claimed, const int32_t* d_qrange, const int32_t* d_qindex,
                                  const uint8_t* d_qdesc, const float* d_qangle, const uint8_t* d_qvalid, const int32_t* d_nq, int qcap, int nproblems,
                                  int32_t* d_q2t, int32_t* d_t2q, int32_t* d_best, int32_t* d_second, int32_t* d_nmatches, void* stream) {
    if (!prm || nproblems < 0 || cap < 1 || cap > ORBF_MAX_FEATURES || qcap < 1 || qcap > ORBF_MAX_FEATURES) return ORBX_ERR_ARG;
    if (prm->rule < ORBS_RULE_MAPPOINTS || prm->rule > ORBS_RULE_FREE) return ORBX_ERR_ARG;
    if (nproblems == 0) return ORBX_OK;
    if (!d_kps || !d_desc || !d_list || !d_nlist || !d_nt || !d_qrange || !d_qdesc || !d_nq || !d_q2t || !d_t2q || !d_nmatches) return ORBX_ERR_ARG;
    if (prm->check_orientation && prm->rule != ORBS_RULE_MAPPOINTS && !d_qangle) return ORBX_ERR_ARG;
    int desc_in_lds = 1;
    const size_t lds = orbs_choose_layout(cap, qcap, desc_in_lds);
    if (lds > 160 * 1024) return ORBX_ERR_CAPACITY;
    orbs_params prm2 = *prm;
    prm2.check_orientation = prm->check_orientation ? 1 : 0;
    orbs::Args a{d_kps, d_desc, nullptr, d_list, d_nt, d_claimed, nullptr, nullptr, d_qdesc, d_qangle, d_qvalid, d_nq,
                 d_q2t, d_t2q, d_best, d_second, d_nmatches, cap, qcap, desc_in_lds, 0, d_nlist, d_qrange, d_qindex, nullptr, nullptr, {}};
    orbf_bounds nob{};
    return orbs_launch<false>(nproblems, lds, (hipStream_t)stream, nob, prm2, a);
}

int orbs_triangulation_search_batch_device(const orbs_params* prm, const float* d_F12, const float* level_sigma2, int nlevels,
                                           const orbx_keypoint* d_kps2, const uint8_t* d_desc2, const int32_t* d_list, const int32_t* d_nlist,
                                           const int32_t* d_nt, int cap, const uint8_t* d_claimed, const int32_t* d_qrange, const int32_t* d_qindex,
                                           const orbx_keypoint* d_kps1, const uint8_t* d_qdesc, const uint8_t* d_qvalid, const int32_t* d_nq, int qcap,
                                           int nproblems, int32_t* d_q2t, int32_t* d_t2q, int32_t* d_best, int32_t* d_second, int32_t* d_nmatches,
                                           void* stream) {
    if (!prm || nproblems < 0 || cap < 1 || cap > ORBF_MAX_FEATURES || qcap < 1 || qcap > ORBF_MAX_FEATURES) return ORBX_ERR_ARG;
    if (prm->rule != ORBS_RULE_TRIANGULATION || !level_sigma2 || nlevels < 1 || nlevels > ORBS_MAX_LEVELS || prm->th < 0 || prm->th > 256) return ORBX_ERR_ARG;
    if (nproblems == 0) return ORBX_OK;
    if (!d_F12 || !d_kps2 || !d_desc2 || !d_list || !d_nlist || !d_nt || !d_qrange || !d_kps1 || !d_qdesc || !d_nq || !d_q2t || !d_t2q || !d_nmatches)
        return ORBX_ERR_ARG;
    int desc_in_lds = 1;
    const size_t lds = orbs_choose_layout(cap, qcap, desc_in_lds);
    if (lds > 160 * 1024) return ORBX_ERR_CAPACITY;
    orbs_params prm2 = *prm;
    prm2.check_orientation = prm->check_orientation ? 1 : 0;
    orbs::Args a{d_kps2, d_desc2, nullptr, d_list, d_nt, d_claimed, nullptr, nullptr, d_qdesc, nullptr, d_qvalid, d_nq,
                 d_q2t, d_t2q, d_best, d_second, d_nmatches, cap, qcap, desc_in_lds, 0, d_nlist, d_qrange, d_qindex, d_kps1, d_F12, {}};
    for (int i = 0; i < ORBS_MAX_LEVELS; ++i) a.epi_thr[i] = orbs_epipolar_bound(level_sigma2[i < nlevels ? i : nlevels - 1]);
    orbf_bounds nob{};
    return orbs_launch<false>(nproblems, lds, (hipStream_t)stream, nob, prm2, a);
}

int orbs_agreement_batch_device(const int32_t* d_match12, const int32_t* d_n1, int cap1, const int32_t* d_match21, const int32_t* d_n2, int cap2,
                                int nproblems, int32_t* d_out12, int32_t* d_nfound, void* stream) {
    if (nproblems < 0 || cap1 < 1 || cap2 < 1) return ORBX_ERR_ARG;
    if (nproblems == 0) return ORBX_OK;
    if (!d_match12 || !d_n1 || !d_match21 || !d_n2 || !d_out12 || !d_nfound) return ORBX_ERR_ARG;
    hipLaunchKernelGGL(orbs::k_agreement, dim3(nproblems), dim3(256), 0, (hipStream_t)stream, d_match12, d_n1, cap1, d_match21, d_n2, cap2, d_out12, d_nfound);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbs_bow_ranges_batch_device(const uint32_t* d_fvq_node, const int32_t* d_fvq_off, const int32_t* d_nfv_q, const uint32_t* d_fvt_node,
                                 const int32_t* d_fvt_off, const int32_t* d_nfv_t, int cap, int nproblems, int32_t* d_qrange, int32_t* d_nq, void* stream) {
    if (nproblems < 0 || cap < 1 || cap > ORBF_MAX_FEATURES) return ORBX_ERR_ARG;
    if (nproblems == 0) return ORBX_OK;
    if (!d_fvq_node || !d_fvq_off || !d_nfv_q || !d_fvt_node || !d_fvt_off || !d_nfv_t || !d_qrange || !d_nq) return ORBX_ERR_ARG;
    hipLaunchKernelGGL(orbs::k_bow_ranges, dim3(4, nproblems), dim3(256), 0, (hipStream_t)stream, d_fvq_node, d_fvq_off, d_nfv_q, d_fvt_node, d_fvt_off,
                       d_nfv_t, cap, d_qrange, d_nq);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

}  // extern "C"
