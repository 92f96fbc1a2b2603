// FAST-9 + cell-local NMS + raster-ordered cell lists for gfx950 (cv::FAST(cell, th, true) per grid cell with the threshold-7 fallback,
// reference src/ORBextractor.cc:599-614): k_fast_cells (one workgroup per row band of a grid cell) and k_fast_blur (the same beside the blur's
// strips in one launch: launch groups too small to fill the chip).
#include <algorithm>
#include <type_traits>

#include "orbx_device.h"
#include "orbx_launch.h"
#include "k_blur_strip.h"

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


// (launch_extract's FAST stage.  fuse_blur: the blur's short strips ride in the same launch)
int launch_fast(const Batch& b, const HostGeom& hg, hipStream_t stream, bool fuse_blur) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
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
    return ORBX_OK;
}

}  // namespace orbx
