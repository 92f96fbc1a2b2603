// Quotas and retainBest for gfx950 (reference src/ORBextractor.cc:622-701: quota redistribution over the cells of a level, KeyPointsFilter::retainBest per
// cell and per level in libstdc++'s introselect order, HarrisResponses :79-120): k_quota, k_cell_select (+ _long), k_level_select.
#include <algorithm>
#include <type_traits>

#include "orbx_device.h"
#include "orbx_launch.h"

namespace orbx {

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


int launch_quota(const Batch& b, const HostGeom& hg, hipStream_t stream) {
    const DevGeom& g = hg.g;
    const size_t lds = (size_t)g.quota_cells * QUOTA_LDS_PER_CELL;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_quota), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
    hipLaunchKernelGGL(k_quota, dim3(b.nframes * g.nlevels), dim3(64), lds, stream, b);
    ORBX_LAUNCH_CHECK();
    return ORBX_OK;
}

int launch_cell_select(const Batch& b, const HostGeom& hg, hipStream_t stream) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
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
    return ORBX_OK;
}

int launch_level_select(const Batch& b, const HostGeom& hg, hipStream_t stream) {
    const DevGeom& g = hg.g;
    const size_t lds = (size_t)g.sel_lds_level;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ORBX_ERR_DEVICE;
    hipLaunchKernelGGL(k_level_select, dim3(b.nframes * g.nlevels), dim3(64), lds, stream, b);
    ORBX_LAUNCH_CHECK();
    return ORBX_OK;
}

}  // namespace orbx
