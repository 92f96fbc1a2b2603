// k_describe_od — orientation + rBRIEF with the GaussianBlur computed ON DEMAND, per keypoint, on the matrix cores (round 6).
//
// reference: IC_Angle src/ORBextractor.cc:124-151, GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) :760, computeOrbDescriptor :154-194,
//            scaling / output :769-775.
//
// k_describe reads two things per keypoint: the 31 x 31 patch of the PLAIN level (IC_Angle) and the 37 x 37 window of the BLURRED level
// the rotated pattern can reach — the only consumer the blurred plane has.  Here a wave stages ONE 43 x 48-byte window of the plain
// level per keypoint (rows y-21 .. y+21, 48 bytes from the 4-byte aligned column at or left of x-22: the patch AND everything the 7 x 7
// filter needs around the 37 x 37 tap window), computes IC_Angle from it, then blurs it IN PLACE with exact int8 matrix products
// (v_mfma_i32_16x16x64_i8; the arithmetic of k_blur_mfma on window-shaped operands) and takes the 512 taps from the result.  The blurred
// plane, the blur kernel (VGA: 0.49 ms and 2.2 GB per 1024 frames; 1080p: 0.80 ms per 256 frames on the VALU) and the second gather
// of every keypoint disappear; the windows of a frame cover 1.4 x the pyramid's pixels at VGA / 1000 and 0.58 x at 1080p / 2000.
//
//   window      row r <-> level row y-21+r, reflected (BORDER_REFLECT_101) by the loader's row index; byte c <-> level column xs+c with
//               xs = (x-22) & ~3 (window column cx = x-xs in 22..25 is the keypoint).  Loaded by 16-byte LDS-DMA, three chunks per row:
//               129 chunks = three wave instructions per keypoint (one of them a single lane).  Chunks that would leave the row's readable
//               bytes are fetched from a clamped start and put right afterwards, together with the reflected columns of windows that
//               reach over the level's left / right edge, by LDS-to-LDS byte moves (border keypoints only, a few % of a frame).
//               The window is then the level's reflect-101 extension at DISTINCT positions: every window takes the same arithmetic.
//   row pass    Mid[r][c'] = sum_t tap[t] * In[r][c'+1+t]  (c' <-> window column c'+4: outputs start on a dword): per 16-column tile
//               ct and 16-row tile t one MFMA, A = the window rows (a lane's 16 bytes are one ds_read_b128), B = the banded tap matrix
//               (a lane constant).  Pixels go in centred (p - 128) and the accumulator starts at 128, so the result is S - 32768 in
//               signed 16 bits: HI byte signed, LO byte unsigned (centred again) — two int8 operands.
//   column pass Out[ro][c'] = sum_t tap[t] * Mid[ro+t][c']: the row pass leaves lane (c' % 16, g) with Mid rows 16t+4g+i of column c',
//               which IS an A operand of this shape when the summed index K is numbered 16g+4t+i (the tap matrix is permuted to match,
//               a lane constant again): no data moves between the passes.  Two products (HI, LO) per 16 x 16 tile; the result has four
//               consecutive COLUMNS of one output row per lane: one aligned ds_write_b32 into the window, in place.
//   rounding    (S + 0x8000) >> 16 half-up or ties-to-even by ABSOLUTE column (x < w & ~3: OpenCV's SSE2 column filter), saturated —
//               orb_math.h blur_round, as k_blur / k_blur_mfma.
//   H4          taps up to 2 px outside the level read the UNBLURRED reflect-101 border in the reference.  Outputs at
//               out-of-level positions are simply not written: the window keeps the plain reflected pixel there.
// 27 MFMAs (432 matrix-pipe cycles) and ~250 VALU instructions per keypoint for the blur; a wave owns four keypoints.
#include <algorithm>
#include <type_traits>

#include "orbx_device.h"

namespace orbx {

constexpr int OD_KPW = 4;                                        // keypoints per wave (16 lanes each for IC_Angle and the taps)
constexpr int OD_PITCH = 48, OD_ROWS = 43;
constexpr int OD_WIN_BYTES = OD_PITCH * OD_ROWS;                 // 2064 (a multiple of 16)
constexpr int OD_CHUNKS = OD_ROWS * 3;                           // 129 16-byte chunks
constexpr int OD_WAVES = DESC_WAVES;
constexpr int OD_TAIL = 512;                                     // the operand reads of row tile 2 run 5 rows past a window (garbage in, unused out)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int float_bits(float v) { return __builtin_bit_cast(int, v); }
__host__ __device__ constexpr int od_col0(int ct) { return ct == 0 ? 0 : ct == 1 ? 16 : 24; }      // first output column c' of column tile ct
__host__ __device__ constexpr int od_row0(int rt) { return rt == 0 ? 0 : rt == 1 ? 16 : 21; }      // first output row ro of row tile rt
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

// Everything the kernel's lanes need that does not depend on the key point, computed by the COMPILER (constant memory, one load each instead of
// ~170 VALU instructions per wave): the circle masks of the IC_Angle patch, the BRIEF pattern as floats, and the tap operands of the two passes.
constexpr int od_tap(int t) { return t >= 0 && t <= 6 ? (int)((0x12223137312212ull >> (8 * t)) & 255ull) : 0; }      // [18, 34, 49, 55, 49, 34, 18][t], 0 outside
constexpr uint32_t od_taps4(int t0) { return (uint32_t)od_tap(t0) | (uint32_t)od_tap(t0 + 1) << 8 | (uint32_t)od_tap(t0 + 2) << 16 | (uint32_t)od_tap(t0 + 3) << 24; }
struct OdTables {
    uint32_t mask[256];            // circle byte masks of the 31 x 8 patch dwords (umax[] of reference :495-510; slots 248.. = 0)
    float pat[256][4];             // test t: x0, y0, x1, y1
    uint32_t trow[3][64][4];       // row pass, column tile ct, lane: K = window column 16 g + 4 v + byte, output column c' = col0(ct) + n
    uint32_t tcol[3][64][4];       // column pass, row tile rt, lane: K = 16 g + 4 v + byte <-> Mid row 16 v + 4 g + byte, output row ro = row0(rt) + n
};
constexpr uint32_t c_pattern_host[256] = {
#include "orb_pattern_packed.inc"
};
constexpr OdTables make_od_tables() {
    OdTables T{};
    for (int t = 0; t < 256; t++) {
        const int r = t >> 3, c = t & 7;
        const int v = r - HALF_PATCH, av = v < 0 ? -v : v;
        const int um = r < 31 ? (int)((UMAX_NIBBLES >> (4 * (av & 15))) & 15ull) : -1;
        uint32_t mask = 0;
        for (int kk = 0; kk < 4; kk++) {
            const int u = 4 * c + kk - HALF_PATCH;
            if ((u < 0 ? -u : u) <= um) mask |= 0xFFu << (8 * kk);
        }
        T.mask[t] = mask;
        const uint32_t pk = c_pattern_host[t];
        for (int e = 0; e < 4; e++) T.pat[t][e] = (float)(int)(int8_t)(uint8_t)(pk >> (8 * e));
    }
    for (int c = 0; c < 3; c++)
        for (int l = 0; l < 64; l++)
            for (int v = 0; v < 4; v++) {
                const int n = l & 15, g = l >> 4;
                T.trow[c][l][v] = od_taps4(16 * g + 4 * v - (od_col0(c) + n) - 1);
                T.tcol[c][l][v] = od_taps4(16 * v + 4 * g - (od_row0(c) + n));
            }
    return T;
}
static __device__ __constant__ OdTables c_od = make_od_tables();

template <bool FMA>
__global__ __launch_bounds__(OD_WAVES * 64) void k_describe_od(Batch b) {
    __shared__ __attribute__((aligned(16))) float s_pat[256 * 4];       // test t: x0, y0, x1, y1 as floats (one ds_read_b128 = the two points as register pairs)
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[256];       // circle byte masks of the 31 x 8 patch dwords (slots 248.. = 0)
    __shared__ __attribute__((aligned(16))) uint8_t s_win[OD_WAVES * OD_KPW * OD_WIN_BYTES + OD_TAIL];
    const DevGeom& g = b.g;
    int frame, wgi;
    if (!frame_item(b, blockIdx.x, (g.nquads + OD_WAVES - 1) / OD_WAVES, frame, wgi)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = lane >> 4, li = lane & 15;
    const float4 pat_first = *reinterpret_cast<const float4*>(c_od.pat[tid & 255]);      // requested first: loads return in order, the tables are built while the key points are on their way
    const uint32_t mask_first = c_od.mask[tid & 255];
    const int32_t* counts = b.level_count + frame * MAX_LEVELS;
    const int quad = wgi * OD_WAVES + wave_id();
    const bool live = quad < g.nquads;
    const int level = __builtin_amdgcn_readfirstlane(find_level(g.quad_bases, live ? quad : 0));
    const LevelGeom& LG = g.lv[level];
    struct { int w, h, stride, plane_off, sel_base, quad_base, wvec; float scale, kp_size; } L = {
        __builtin_amdgcn_readfirstlane(LG.w), __builtin_amdgcn_readfirstlane(LG.h), __builtin_amdgcn_readfirstlane(LG.stride),
        __builtin_amdgcn_readfirstlane(LG.plane_off), __builtin_amdgcn_readfirstlane(LG.sel_base), __builtin_amdgcn_readfirstlane(LG.quad_base),
        __builtin_amdgcn_readfirstlane(LG.blur_wvec),
        __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, LG.scale))),
        __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, LG.kp_size)))};
    // the wave's four keypoints, the per-level counts and the frame status: wave-uniform data by scalar loads (as k_describe)
    const int k0 = (quad - L.quad_base) * OD_KPW;
    Cand kp;
    typedef int v8i_s __attribute__((ext_vector_type(8)));
    v8i_s s_cnt0, s_cnt1;
    int st0;
    {
        const int sel_cap = __builtin_amdgcn_readfirstlane(LG.sel_cap);
        const int ks = __builtin_amdgcn_readfirstlane(max(min(k0, sel_cap - OD_KPW), 0));        // (the sel block carries 4 slots of padding)
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
    }
    static_assert(OD_WAVES * 64 == 256, "one table entry per thread");
    reinterpret_cast<float4*>(s_pat)[tid] = pat_first;
    s_mask[tid] = mask_first;
    int out_base = 0, total = 0, cnt = 0;
    for (int l = 0; l < g.nlevels; l++) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { if (l == i) c = s_cnt0[i]; if (l == 8 + i) c = s_cnt1[i]; }
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
    int wlim;                                                // bytes of a row that may be read (>= w and >= 64: host-checked)
    if (level == 0) {
        pstride = (unsigned)b.img_row_stride; plain = b.img + (long long)frame * b.img_frame_stride;
        wlim = (int)min((long long)pstride, (long long)((L.w + 15) & ~15));               // include/orbx.h: what a pitched caller buffer promises (any value: the 16-byte DMA takes byte-aligned sources)
    } else { pstride = (unsigned)L.stride; plain = b.pyr + (long long)frame * g.frame_plane_bytes + L.plane_off; wlim = L.stride; }
    uint8_t* const win0 = s_win + wave_id() * OD_KPW * OD_WIN_BYTES;

    // ---- the four windows by LDS-DMA: chunk e = 64 n + lane <-> row e / 3, chunk e % 3 of the row
    int erow[3], ecol[3];
    unsigned eoff[3];                                        // the chunk's offset from the window's first byte in the level
#pragma unroll
    for (int n = 0; n < 3; n++) {
        const int e = 64 * n + lane;
        erow[n] = (e * 171) >> 9;                            // e / 3 for e < 193
        ecol[n] = 16 * (e - 3 * erow[n]);
        eoff[n] = __umul24((unsigned)erow[n], pstride) + (unsigned)ecol[n];
    }
    uint32_t fixmask = 0;                                    // (wave-uniform) bit q: window q has columns to put right; bit 4 + q: it reaches outside the level
#pragma unroll
    for (int q = 0; q < OD_KPW; q++) {
        if (q > 0 && k0 + q >= cnt) continue;                // wave-uniform
        const unsigned posq = (unsigned)__builtin_amdgcn_readlane((int)kp.pos, 16 * q);
        const int xq = posq & 0xFFFF, yq = posq >> 16;
        const int xs = (xq - 22) & ~3;
        if (xs < 0 || xs + 48 > wlim || xq + 21 >= L.w) fixmask |= 1u << q;
        if (xq < 18 || xq + 18 >= L.w || yq < 18 || yq + 21 >= L.h) fixmask |= 16u << q;
        if (xs >= 0 && xs + 48 <= wlim && yq >= 21 && yq + 21 < L.h) {       // (wave-uniform) the whole window lies inside the level's readable rows:
            const uint8_t* srcq = plain + (__umul24((unsigned)(yq - 21), pstride) + (unsigned)xs);      // one scalar base, the lanes' constant offsets
#pragma unroll
            for (int n = 0; n < 3; n++)
                if (n < 2 || lane < OD_CHUNKS - 128)
                    __builtin_amdgcn_global_load_lds((gptr_t)(srcq + eoff[n]), (lptr_t)(win0 + q * OD_WIN_BYTES + 1024 * n), 16, 0, 0);
            continue;
        }
#pragma unroll
        for (int n = 0; n < 3; n++) {
            if (n == 2 && lane >= OD_CHUNKS - 128) continue;
            int Y = yq - 21 + erow[n];
            Y = Y < 0 ? -Y : Y;
            Y = Y >= L.h ? 2 * L.h - 2 - Y : Y;
            Y = min(max(Y, 0), L.h - 1);
            const int X = min(max(xs + ecol[n], 0), wlim - 16);
            __builtin_amdgcn_global_load_lds((gptr_t)(plain + (__umul24((unsigned)Y, pstride) + (unsigned)X)), (lptr_t)(win0 + q * OD_WIN_BYTES + 1024 * n), 16, 0, 0);
        }
    }
    fixmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)fixmask);
    // the tables' barrier behind the DMA issue (only the LDS writes above have to be complete; a wave that returned has left the barrier count)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- lane constants of the two passes (independent of the keypoint)
    const int n16 = li, g4 = grp;                            // the MFMA's view of the lane: row / column lane % 16, K block lane / 16
    // Tile origins: output columns c' = 0 / 16 / 24 + n, output rows ro = 0 / 16 / 21 + n.  The third tile of either axis OVERLAPS the second instead
    // of running past what is needed (c' <= 39, ro <= 36): every output of every tile is one the taps can reach or a second, identical copy
    // of one, every store lands inside the window — no predicate anywhere (round 6: 374 -> 2xx VALU instructions per keypoint).
    v4i Trow[3], Tcol[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        Trow[c] = *reinterpret_cast<const v4i*>(c_od.trow[c][lane]);
        Tcol[c] = *reinterpret_cast<const v4i*>(c_od.tcol[c][lane]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the windows have landed
    wave_lds_fence();

    // ---- border windows: put the clamped chunks and the reflected columns right (LDS to LDS; chunk 1 is never affected: x <= w - 17)
    if (fixmask & 15u) {
#pragma unroll 1
        for (int q = 0; q < OD_KPW; q++) {
            if (!((fixmask >> q) & 1u)) continue;
            const unsigned posq = (unsigned)__builtin_amdgcn_readlane((int)kp.pos, 16 * q);
            const int xq = posq & 0xFFFF;
            const int xs = (xq - 22) & ~3;
            uint8_t* W = win0 + q * OD_WIN_BYTES;
#pragma unroll 1
            for (int j = 0; j < 3; j += 2) {
                const int X0 = xs + 16 * j, Xc = min(max(X0, 0), wlim - 16);
                if (X0 >= 0 && X0 + 16 <= wlim && X0 + 15 < L.w) continue;          // this chunk is what it should be
                // item = (row, dword of the chunk): 172 items, <= 3 per lane; every byte read first, then every dword written
                uint32_t nv[3];
#pragma unroll
                for (int it = 0; it < 3; it++) {
                    const int item = 64 * it + lane;
                    const int r = item >> 2, d = item & 3;
                    uint32_t v = 0;
                    if (item < 4 * OD_ROWS) {
#pragma unroll
                        for (int bb = 0; bb < 4; bb++) {
                            const int X = X0 + 4 * d + bb;
                            int Xr = X < 0 ? -X : (X >= L.w ? 2 * L.w - 2 - X : X);
                            Xr = min(max(Xr, 0), L.w - 1);
                            const int src = (Xr >= Xc && Xr < Xc + 16) ? 16 * j + (Xr - Xc) : min(max(Xr - xs, 0), OD_PITCH - 1);
                            v |= (uint32_t)W[r * OD_PITCH + src] << (8 * bb);
                        }
                    }
                    nv[it] = v;
                }
                wave_lds_fence();
#pragma unroll
                for (int it = 0; it < 3; it++) {
                    const int item = 64 * it + lane;
                    if (item < 4 * OD_ROWS) *reinterpret_cast<uint32_t*>(W + (item >> 2) * OD_PITCH + 16 * j + 4 * (item & 3)) = nv[it];
                }
                wave_lds_fence();
            }
        }
    }

    // ---- IC_Angle on the plain window (reference :124-151; the group's own window, rows 6 .. 36, columns cx-15 .. cx+15)
    const int xs_own = (x - 22) & ~3, cx = x - xs_own;
    const uint8_t* Wown = win0 + grp * OD_WIN_BYTES;
    int m10, m01;
    {
        const int rsub = li >> 1, hf = li & 1;
        const int boff = cx - HALF_PATCH + 16 * hf;          // first byte of the lane's 16 in its row
        const int sh = boff & 3;
        const uint32_t* rowp = reinterpret_cast<const uint32_t*>(Wown + (6 + rsub) * OD_PITCH + (boff & ~3));
        uint32_t uw[4];
#pragma unroll
        for (int d = 0; d < 4; d++) uw[d] = (uint32_t)(16 * hf + 4 * d) * 0x01010101u + 0x03020100u;      // u + 15 of the dword's four pixels
        uint32_t a_su = 0, a_si = 0, a_r = 0;
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const uint32_t* p = rowp + (8 * n) * (OD_PITCH / 4);     // row 31 (n = 3, rsub = 7) is masked, still inside the window
            uint32_t dw[5];
#pragma unroll
            for (int d = 0; d < 5; d++) dw[d] = p[d];
            const uint4 mk = *reinterpret_cast<const uint4*>(s_mask + (8 * (8 * n + rsub) + 4 * hf));
            const uint32_t mm[4] = {mk.x, mk.y, mk.z, mk.w};
            uint32_t srow = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t Im = __builtin_amdgcn_alignbyte(dw[d + 1], dw[d], (uint32_t)sh) & mm[d];
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
    wave_lds_fence();                                          // every patch read is done before the windows are overwritten

    // ---- the blur, window after window, the whole wave on each
    const unsigned a_off = (unsigned)(n16 * OD_PITCH + 16 * g4);                 // the lane's 16 operand bytes in a row tile
    const unsigned o_off = (unsigned)((n16 + 3) * OD_PITCH + 4 + 4 * g4);        // the lane's output dword in tile (0, 0)
    auto blur_window = [&](auto EDGE, uint8_t* W, int xs, int yq) {
        constexpr bool edge = decltype(EDGE)::value;
        v4i A[3];
        {
            const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(W + a_off);
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:768\n\tds_read_b128 %2, %3 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]) : "v"(ra) : "memory");
        }
#pragma unroll
        for (int t = 0; t < 3; t++) {
            A[t].x = (int)((uint32_t)A[t].x ^ 0x80808080u); A[t].y = (int)((uint32_t)A[t].y ^ 0x80808080u);
            A[t].z = (int)((uint32_t)A[t].z ^ 0x80808080u); A[t].w = (int)((uint32_t)A[t].w ^ 0x80808080u);
        }
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
            // row pass of column tile ct: z[t] = S - 32768 for Mid rows 16 t + 4 g + i, column c' = col0(ct) + n
            const v4i c128 = {128, 128, 128, 128};
            v4i z[3];
#pragma unroll
            for (int t = 0; t < 3; t++) z[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], Trow[ct], c128, 0, 0, 0);
            int h4[4] = {0, 0, 0, 0}, l4[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)z[t][1], (uint32_t)z[t][0], 0x05010400u);      // lo0 lo1 hi0 hi1
                const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)z[t][3], (uint32_t)z[t][2], 0x05010400u);
                l4[t] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
                h4[t] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
            }
            const v4i HI = {h4[0], h4[1], h4[2], h4[3]}, LO = {l4[0], l4[1], l4[2], l4[3]};
            const int X0 = xs + 4 + od_col0(ct) + 4 * g4;    // level column of the lane's first output byte (a multiple of 4)
            const uint32_t tw = X0 < L.wvec ? 1u : 0u;       // ties-to-even columns (blur_wvec is a multiple of 4)
            const uint32_t cadd = (uint32_t)(257 * 32896 + 0x7FFF) + (tw ^ 1u);      // the centring offsets + the rounding constant (+ 1 more: half up)
            uint32_t keepc = 0;                              // (edge windows) bytes of the dword whose column lies outside the level
            if (edge) {
#pragma unroll
                for (int bb = 0; bb < 4; bb++) keepc |= ((unsigned)(X0 + bb) < (unsigned)L.w ? 0u : 0xFFu) << (8 * bb);
            }
#pragma unroll
            for (int rt = 0; rt < 3; rt++) {
                const v4i zero = {0, 0, 0, 0};
                v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(HI, Tcol[rt], zero, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = (int)(((uint32_t)acc[i] << 8) + cadd);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(LO, Tcol[rt], acc, 0, 0, 0);
                uint32_t qv[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // + bit 16 (ties to even; the half-up columns carry their + 1 in `cadd`): shift, and, add — three 2-cycle instructions instead of a
                    // bit-field extract and a three-operand add at 4 cycles each (profiles/r02_valu_issue_rates2.txt)
                    const uint32_t t = (uint32_t)acc[i];
                    qv[i] = t + ((t >> 16) & tw);
                }
                const us2v lo2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(qv[1], qv[0], 0x07060302u)), as_us2v(0x00FF00FFu));
                const us2v hi2 = __builtin_elementwise_min(as_us2v(__builtin_amdgcn_perm(qv[3], qv[2], 0x07060302u)), as_us2v(0x00FF00FFu));
                uint32_t o = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi2), __builtin_bit_cast(uint32_t, lo2), 0x06040200u);
                uint32_t* dst = reinterpret_cast<uint32_t*>(W + o_off + (od_row0(rt) * OD_PITCH + od_col0(ct)));
                if (edge) {
                    // out-of-level positions keep the plain reflected pixel (H4): per-byte merge with what the window holds
                    const int Y = yq - 18 + od_row0(rt) + n16;
                    const uint32_t keep = (unsigned)Y < (unsigned)L.h ? keepc : 0xFFFFFFFFu;
                    o = (o & ~keep) | (*dst & keep);
                }
                *dst = o;
            }
        }
    };
#pragma unroll 1
    for (int q = 0; q < OD_KPW; q++) {
        if (q > 0 && k0 + q >= cnt) continue;                // wave-uniform
        const unsigned posq = (unsigned)__builtin_amdgcn_readlane((int)kp.pos, 16 * q);
        const int xq = posq & 0xFFFF, yq = posq >> 16;
        const int xs = (xq - 22) & ~3;
        uint8_t* W = win0 + q * OD_WIN_BYTES;
        if ((fixmask >> (4 + q)) & 1u) blur_window(std::true_type{}, W, xs, yq);      // (wave-uniform)
        else blur_window(std::false_type{}, W, xs, yq);
    }
    wave_lds_fence();

    // ---- rotated BRIEF on the blurred window (:154-194)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    sincosf_orb(angle * factorPI, &sn, &cs);
    const float4* pat = reinterpret_cast<const float4*>(s_pat) + li;
    uint32_t mybits = 0;                                        // bit j: test li + 16 j
    {
        // Both coordinates of a point in ONE packed-fp32 instruction each step (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: IEEE single
        // precision per half, the same roundings as the scalar forms):  (x sn, x cs), (y cs, y sn), then (x sn + y cs, x cs - y sn).
        const uint8_t* ctr = Wown + 21 * OD_PITCH + cx;
        const f2v SC = {sn, cs};
        auto rotate = [&](f2v pt) -> f2v {                      // (x, y) -> (row offset, column offset), unrounded
            f2v m1, r;
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(m1) : "v"(pt), "v"(SC));                  // (y cs, y sn)
            if (FMA) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(pt), "v"(SC), "v"(m1));   // fma(x, sn, y cs), fma(x, cs, -(y sn))
            else {
                f2v m0;
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(m0) : "v"(pt), "v"(SC));              // (x sn, x cs)
                asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(m0), "v"(m1));                               // (x sn + y cs, x cs - y sn)
            }
            return r;
        };
        // cvRound (ties to even) by the magic constant: for |f| < 2^22, f + 1.5 * 2^23 rounds to the integer nearest f (ties to even, the
        // default mode) and its bit pattern is 0x4B400000 + that integer.  iy * pitch + ix is then ONE 24-bit multiply-add on the bit patterns
        // (the low 24 bits of the first are 0x400000 + iy); the constant it drags along is taken off the base address once.
        f2v MAGIC = {12582912.0f, 12582912.0f};
        asm("" : "+v"(MAGIC));                                  // in a VGPR pair: as an SGPR pair with op_sel_hi:[1,0] (what hipcc 7.2 emits for the splat) the packed add gave wrong sums
        const uint32_t ctr_a = (uint32_t)(uintptr_t)(lptr_t)ctr - (uint32_t)(OD_PITCH * 0x400000 + 0x4B400000);
        typedef const uint8_t __attribute__((address_space(3))) * lbyte_t;
#pragma unroll
        for (int j = 15; j >= 0; j--) {                       // downwards: the bits are shifted in from the bottom, test li + 16 j ends at bit j
            const float4 P = pat[16 * j];
            f2v r0 = rotate((f2v){P.x, P.y}), r1 = rotate((f2v){P.z, P.w});
            r0 = r0 + MAGIC;                                    // (v_pk_add_f32)
            r1 = r1 + MAGIC;
            // (the elements go through scalar temporaries: hipcc 7.2 folds __builtin_bit_cast(int, vec.y) of a vector ELEMENT expression to element 0 —
            //  `mad24(bits(r.x), 48, bits(r.x))` came out of the direct form; tools/microbench/bitcast_vector_element.hip)
            const float r0y = r0.x, r0x = r0.y, r1y = r1.x, r1x = r1.y;
            const uint32_t o0 = (uint32_t)(__mul24(float_bits(r0y), OD_PITCH) + float_bits(r0x));
            const uint32_t o1 = (uint32_t)(__mul24(float_bits(r1y), OD_PITCH) + float_bits(r1x));
            const uint32_t v0 = *(lbyte_t)(uintptr_t)(ctr_a + o0), v1 = *(lbyte_t)(uintptr_t)(ctr_a + o1);
            // mybits = 2 mybits + (v0 < v1): the comparison's carry straight into the add (two instructions instead of compare, select, or)
            asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mybits) : "v"(v0), "v"(v1) : "vcc");
        }
    }
    // 16 x 16 bit-matrix transpose inside the group (ds_swizzle butterflies, as k_describe)
    uint32_t half = mybits;
    auto stage = [&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr uint32_t M0 = s == 8 ? 0x00FFu : s == 4 ? 0x0F0Fu : s == 2 ? 0x3333u : 0x5555u;
        const bool hi = (li & s) != 0;
        const uint32_t yv = (uint32_t)__builtin_amdgcn_ds_swizzle((int)half, (s << 10) | 0x1F);        // lane ^ s
        const uint32_t ysh = hi ? (yv >> s) : (yv << s);
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

// every level must offer 48 readable bytes per row and be at least 64 px wide (the border fix-up's case analysis), rows single-reflect
bool describe_od_supported(const Batch& b, const HostGeom& hg) {
    const DevGeom& g = hg.g;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].w < 64 || g.lv[l].h < 44) return false;
    const long long wlim0 = std::min<long long>(b.img_row_stride, (g.lv[0].w + 15) & ~15);
    return wlim0 >= 64 && wlim0 >= g.lv[0].w;
}

int launch_describe_od(const Batch& b, const HostGeom& hg, hipStream_t stream) {
    const DevGeom& g = hg.g;
    const dim3 grid(frame_item_blocks(b, (g.nquads + OD_WAVES - 1) / OD_WAVES)), block(OD_WAVES * 64);
    if (g.fp_contract) hipLaunchKernelGGL(k_describe_od<true>, grid, block, 0, stream, b);
    else hipLaunchKernelGGL(k_describe_od<false>, grid, block, 0, stream, b);
    return hipGetLastError() == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

}  // namespace orbx
