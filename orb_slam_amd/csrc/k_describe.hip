// Orientation + rBRIEF + output from the blurred plane for gfx950 (reference src/ORBextractor.cc: IC_Angle :124-151, computeOrbDescriptor :154-194,
// scaling :769-775): k_describe (four keypoints per wave).  The one-frame call and ORBX_BLUR_ON_DEMAND=0 take it; full launch groups take
// k_describe_od.hip, which computes the blur it needs itself.
#include <algorithm>
#include <type_traits>

#include "orbx_device.h"
#include "orbx_launch.h"

namespace orbx {

// ------------------------------------------------------------------------------------ orientation + rBRIEF + output
// Four keypoints per wave, 16 lanes each (round 2; one wave per keypoint before: every lane of a wave then repeated the same
// fastAtan2 + double-precision sin / cos, ~28 % of the kernel's instructions, and 47 v_readfirstlane + 64-bit tap addresses).
//   IC_Angle (reference :124-151): the 31 x 31 box is cut into 31 rows x 8 dwords (u = -15 .. 16); a lane owns dword column
//     lane & 7 and the rows of one parity: 16 (unaligned) dword loads, the circle as byte masks from an LDS table (constant LDS
//     offsets), two v_dot4_u32_u8 per dword (sum of (u + 15) I and sum of I), the row weight as a multiply-add; 4-step reduction.
//   rBRIEF (:154-194): lane i of a group evaluates tests i, i + 16, ..., i + 240.  The 37 x 37 window the rotated pattern can reach
//     (|offset| <= 18) is first copied from the blurred level into LDS (LDS-DMA, 6 wave instructions per keypoint): 512 scattered
//     byte gathers per keypoint straight from global memory kept the kernel bound by the L1's cache-line rate (one wave-load
//     touched 40-64 lines), the LDS serves them at bank speed.  The pattern comes from an LDS table of floats (one 16-byte read
//     per test, no unpacking); the rotated coordinates are rounded with v_rndne and the tap offset iy * pitch + ix is formed in
//     float (exact) and converted once.  The 16 x 16 test bits of a group are transposed into descriptor halfwords by ds_swizzle.
//   Keypoints closer than 19 px to an edge may read the level's UNBLURRED reflect-101 border: their group of lanes
//     takes its taps from global memory with the reflection in the index math.
// Waves are formed per level (slots padded to multiples of 4), so the level is wave-uniform and its geometry scalar.
constexpr int DESC_KPW = 4;
constexpr int DESC_WIN_PITCH = 40, DESC_WIN_ROWS = 37, DESC_WIN_BYTES = DESC_WIN_PITCH * DESC_WIN_ROWS;   // 37 px + up to 3 px of dword alignment per row

// FMA: the two rotation expressions of computeOrbDescriptor as the reference's own build flags contract them (orbx_params::fp_contract):
// `x*b + y*a` -> fma(x, b, y*a), `x*a - y*b` -> fma(x, a, -(y*b)); false: unfused (ISO evaluation, the default).
template <bool FMA>
__global__ __launch_bounds__(DESC_WAVES * 64) void k_describe(Batch b) {
#if ORBX_DESC_PACKED_PATTERN
    __shared__ uint32_t s_pat[256];                                    // test t: x0, y0, x1, y1 as the four int8 of c_pattern[t] (1 KB: six workgroups per CU; as floats, 4 KB: five)
#else
    __shared__ __attribute__((aligned(16))) float s_pat[256 * 4];     // test t: x0, y0, x1, y1
#endif
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[256];       // circle byte masks of the 31 x 8 patch dwords (slots 248.. = 0)
    __shared__ __attribute__((aligned(16))) uint8_t s_win[DESC_WAVES * DESC_KPW * DESC_WIN_BYTES];   // per keypoint: 37 rows x 40 bytes of the blurred level
    const DevGeom& g = b.g;
    int frame, wgi;
    if (!frame_item(b, blockIdx.x, (g.nquads + DESC_WAVES - 1) / DESC_WAVES, frame, wgi)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = lane >> 4, li = lane & 15;
    // The wave's chain of dependent memory round trips sets this kernel's pace as much as its arithmetic, so everything is requested
    // as early as its address is known: the per-level counts (one load, lane l holds level l) and the frame status first, then the
    // wave's keypoints, and only then the LDS tables are built (their barrier rides on those loads); the 37 x 37 windows of the
    // blurred level follow by LDS-DMA as soon as the keypoints are there, in flight during IC_Angle and the angle arithmetic.
#ifndef ORBX_DESC_EARLY_PATTERN
#define ORBX_DESC_EARLY_PATTERN 1      // (0.701 -> 0.692 ms per 1024 VGA frames)
#endif
    // (round 5) the thread's word of the BRIEF pattern is requested FIRST: loads return in order, so the tables can be built while the
    // counts and the keypoint are still on their way — before, the pattern load started only after the keypoint had arrived: one
    // dependent round trip more in front of the tables' barrier
    const uint32_t pk_first = ORBX_DESC_EARLY_PATTERN ? c_pattern[tid & 255] : 0u;
#ifndef ORBX_DESC_SCALAR_LOADS
#define ORBX_DESC_SCALAR_LOADS 1       // (0.692 -> 0.686)
#endif
    const int32_t* counts = b.level_count + frame * MAX_LEVELS;
    int cl = 0, st0 = 0;
    if (!ORBX_DESC_SCALAR_LOADS) {
        cl = lane < g.nlevels ? counts[lane] : 0;
        st0 = b.status[frame];
    }
    const int quad = wgi * DESC_WAVES + wave_id();
    const bool live = quad < g.nquads;
    const int level = __builtin_amdgcn_readfirstlane(find_level(g.quad_bases, live ? quad : 0));
    // the level's geometry as scalars (wave-uniform by construction; pinned so that nothing is re-read through per-lane addresses)
    const LevelGeom& LG = g.lv[level];
    struct { int w, h, stride, plane_off, sel_base, quad_base; float scale, kp_size; } L = {
        __builtin_amdgcn_readfirstlane(LG.w), __builtin_amdgcn_readfirstlane(LG.h), __builtin_amdgcn_readfirstlane(LG.stride),
        __builtin_amdgcn_readfirstlane(LG.plane_off), __builtin_amdgcn_readfirstlane(LG.sel_base), __builtin_amdgcn_readfirstlane(LG.quad_base),
        __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, LG.scale))),
        __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, LG.kp_size)))};
    // The group's keypoint is requested before the counts are there (round 4: its address depends on the quad only; the counts decide
    // whether it is one — the slot index is clamped into the level's list, what lies behind the list's end is never used).  One
    // dependent memory round trip less in front of the two gathers.
    const int k0 = (quad - L.quad_base) * DESC_KPW;
    Cand kp;
    typedef int v8i_s __attribute__((ext_vector_type(8)));
    v8i_s s_cnt0 = {0, 0, 0, 0, 0, 0, 0, 0}, s_cnt1 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ORBX_DESC_SCALAR_LOADS) {
        // (round 5) The wave's four keypoints, the per-level counts and the frame status are wave-uniform data: SCALAR loads.  This kernel
        // keeps the vector-memory front end 0.7-0.8 busy with its gathers, and a small vector load queues behind the gathers of the ~20
        // other waves of the CU (the keypoint wait was 23 % of a wave's life, profiles/r04_describe_wave_phases.txt); the scalar cache path
        // does not.  (Written by the selection kernels of earlier launches: coherent at the kernel boundary.)
        const int sel_cap = __builtin_amdgcn_readfirstlane(LG.sel_cap);
        const int ks = __builtin_amdgcn_readfirstlane(max(min(k0, sel_cap - DESC_KPW), 0));        // (the sel block carries 4 slots of padding)
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
    } else kp = b.sel[(long long)frame * g.frame_sel + L.sel_base + min(max(k0 + grp, 0), __builtin_amdgcn_readfirstlane(LG.sel_cap) - 1)];
    auto build_tables = [&]() {
    for (int t = tid; t < 256; t += DESC_WAVES * 64) {
        const uint32_t pk = (ORBX_DESC_EARLY_PATTERN && t == tid) ? pk_first : c_pattern[t];
#if ORBX_DESC_PACKED_PATTERN
        s_pat[t] = pk;
#else
        reinterpret_cast<float4*>(s_pat)[t] = make_float4((float)(int)(int8_t)pk, (float)(int)(int8_t)(pk >> 8), (float)(int)(int8_t)(pk >> 16), (float)(int)(int8_t)(pk >> 24));
#endif
        // umax[] (reference :495-510) depends only on HALF_PATCH_SIZE = 15: nibble v of UMAX_NIBBLES (the host checks it against the computed table)
        const int r = t >> 3, c = t & 7;
        const int v = r - HALF_PATCH, av = v < 0 ? -v : v;
        const int um = r < 31 ? (int)((UMAX_NIBBLES >> (4 * (av & 15))) & 15ull) : -1;
        uint32_t mask = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int u = 4 * c + kk - HALF_PATCH;
            if ((u < 0 ? -u : u) <= um) mask |= 0xFFu << (8 * kk);
        }
        s_mask[t] = mask;
    }
    };
    if (ORBX_DESC_EARLY_PATTERN) build_tables();
    int out_base = 0, total = 0, cnt = 0;
    for (int l = 0; l < g.nlevels; l++) {
        int c;
        if (ORBX_DESC_SCALAR_LOADS) {
            c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) { if (l == i) c = s_cnt0[i]; if (l == 8 + i) c = s_cnt1[i]; }
        } else c = __builtin_amdgcn_readlane(cl, l);
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
    if (!ORBX_DESC_EARLY_PATTERN) build_tables();
#ifndef ORBX_DESC_LATE_BARRIER
#define ORBX_DESC_LATE_BARRIER 1     // the tables' barrier behind the window DMA issue (round 5: 0.727 -> 0.701 ms; 0 = in front of it, rounds 2-4; behind the
                                     // patch loads' issue as well: 0.714)
#endif
    if (!ORBX_DESC_LATE_BARRIER) __syncthreads();
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
    if (level == 0) { pstride = (unsigned)b.img_row_stride; plain = b.img + (long long)frame * b.img_frame_stride; }
    else { pstride = (unsigned)L.stride; plain = b.pyr + (long long)frame * g.frame_plane_bytes + L.plane_off; }
    const uint8_t* blur = b.blur + (long long)frame * g.frame_plane_bytes + L.plane_off;
    // rounded pattern offsets never exceed 18 px (|(-13,-13)| = 18.4): keypoints at least 19 px from every edge — all but the
    // outermost ring of candidates — take the branch-free path with the window in LDS
    const bool interior = x >= 19 && y >= 19 && x < L.w - 19 && y < L.h - 19;
    {
        // window of keypoint q: rows y-18 .. y+18, 40 bytes from the aligned start at or left of x-18, row after row (pitch 40 = 10
        // dwords), i.e. 370 consecutive LDS dwords: 6 global_load_lds_dword of the whole wave per keypoint (lane i of instruction n
        // fetches dword e = 64 n + i: row e / 10, column e % 10; the lane offsets are the same for the four keypoints).  A row's last
        // dword may reach past the level's last pixel: it stays inside the blurred plane (rows are padded to 64, a next row exists)
        // and those bytes are never tapped.
        typedef const void __attribute__((address_space(1))) * gptr_t;
        typedef void __attribute__((address_space(3))) * lptr_t;
        unsigned eoff[6];
#pragma unroll
        for (int n = 0; n < 6; n++) {
            const unsigned e = 64u * n + (unsigned)lane, er = (e * 205u) >> 11;           // e / 10 for e < 1029
            eoff[n] = __umul24(er, (unsigned)L.stride) + 4u * (e - 10u * er);
        }
        uint8_t* win0 = s_win + wave_id() * DESC_KPW * DESC_WIN_BYTES;
#pragma unroll
        for (int q = 0; q < DESC_KPW; q++) {
            const unsigned posq = (unsigned)__builtin_amdgcn_readlane((int)kp.pos, 16 * q);
            const bool inq = __builtin_amdgcn_readlane((int)interior, 16 * q) != 0;
            if (!inq || (q > 0 && k0 + q >= cnt)) continue;                              // wave-uniform
            const int xq = posq & 0xFFFF, yq = posq >> 16;
            const uint8_t* srcq = blur + __umul24((unsigned)(yq - 18), (unsigned)L.stride) + (unsigned)((xq - 18) & ~3);
#pragma unroll
            for (int n = 0; n < 6; n++)
                if (n < 5 || lane < DESC_WIN_ROWS * 10 - 320)
                    __builtin_amdgcn_global_load_lds((gptr_t)(srcq + eoff[n]), (lptr_t)(win0 + q * DESC_WIN_BYTES + 256 * n), 4, 0, 0);
        }
    }

    if (ORBX_DESC_LATE_BARRIER) {
        // the tables' barrier behind the window DMA issue: only the LDS writes have to be complete (no vmcnt wait: the DMA stays in flight);
        // a wave that returned above has left the workgroup's barrier count with its s_endpgm
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // IC_Angle on the unblurred level (:705-706 run before the blur)
    int m10, m01;
    {
        // The 31 x 32-byte patch as FOUR 16-byte loads per lane (lane = row 8 n + li / 2, half li % 2 of the row; rounds 1-3: sixteen dword
        // loads per lane, a row parity and a dword column each) and the circle masks of a lane's four dwords as one ds_read_b128: 12 vector-memory
        // and 12 LDS instructions less per wave for the same bytes and the same sums (-1 % on the VGA stream, -6 % on the 1080p one).
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
        const int rsub = li >> 1, hf = li & 1;
        const unsigned off0 = (unsigned)(x - HALF_PATCH + 16 * hf) + __umul24((unsigned)(y - HALF_PATCH + rsub), pstride);
        u32x4_u P[4];
#pragma unroll
        for (int n = 0; n < 4; n++) P[n] = *reinterpret_cast<const u32x4_u*>(plain + (off0 + (unsigned)(8 * n) * pstride));     // row 31 (n = 3, rsub = 7) is masked, still inside the level
        uint32_t uw[4];
#pragma unroll
        for (int d = 0; d < 4; d++) uw[d] = (uint32_t)(16 * hf + 4 * d) * 0x01010101u + 0x03020100u;      // u + 15 of the dword's four pixels
        uint32_t a_su = 0, a_si = 0, a_r = 0;
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const uint4 mk = *reinterpret_cast<const uint4*>(s_mask + (8 * (8 * n + rsub) + 4 * hf));
            const uint32_t mm[4] = {mk.x, mk.y, mk.z, mk.w};
            uint32_t srow = 0;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t Im = P[n][d] & mm[d];
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

    // rotated BRIEF on the blurred level (:154-194)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float sn, cs;
    sincosf_orb(angle * factorPI, &sn, &cs);
#if ORBX_DESC_PACKED_PATTERN
    struct PatRow {          // one ds_read_b32 and four v_cvt_f32_i32 with a sign-extending byte select (SDWA) per test
        const uint32_t* p;
        __device__ __forceinline__ float4 operator[](int i) const {
            const uint32_t pk = p[i];
            return make_float4((float)(int)(int8_t)pk, (float)(int)(int8_t)(pk >> 8), (float)(int)(int8_t)(pk >> 16), (float)(int)(int8_t)(pk >> 24));
        }
    } pat{s_pat + li};
#else
    const float4* pat = reinterpret_cast<const float4*>(s_pat) + li;
#endif
    uint32_t mybits = 0;                                        // bit j: test li + 16 j
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the wave's window DMA has landed (issued before IC_Angle)
    wave_lds_fence();                                           // the windows are private to this wave: no workgroup barrier
    if (interior) {
        const uint8_t* win = s_win + (wave_id() * DESC_KPW + grp) * DESC_WIN_BYTES;
        const int xa = (x - 18) & ~3;
        const uint8_t* ctr = win + 18 * DESC_WIN_PITCH + (x - xa);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float4 P = pat[16 * j];
            const float fy0 = FMA ? __builtin_fmaf(P.x, sn, P.y * cs) : P.x * sn + P.y * cs, fx0 = FMA ? __builtin_fmaf(P.x, cs, -(P.y * sn)) : P.x * cs - P.y * sn;
            const float fy1 = FMA ? __builtin_fmaf(P.z, sn, P.w * cs) : P.z * sn + P.w * cs, fx1 = FMA ? __builtin_fmaf(P.z, cs, -(P.w * sn)) : P.z * cs - P.w * sn;
            // cvRound (ties to even) of both coordinates, then iy * pitch + ix exactly in float (the fused multiply-add rounds nothing here)
            const int o0 = (int)__builtin_fmaf(__builtin_rintf(fy0), (float)DESC_WIN_PITCH, __builtin_rintf(fx0));
            const int o1 = (int)__builtin_fmaf(__builtin_rintf(fy1), (float)DESC_WIN_PITCH, __builtin_rintf(fx1));
            const int v0 = ctr[o0], v1 = ctr[o1];
            mybits |= (uint32_t)(v0 < v1) << j;
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < 16; j++) {
            const float4 P = pat[16 * j];
            int val[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float px = e ? P.z : P.x, py = e ? P.w : P.y;
                const int iy = cv_round_f(FMA ? __builtin_fmaf(px, sn, py * cs) : px * sn + py * cs);
                const int ix = cv_round_f(FMA ? __builtin_fmaf(px, cs, -(py * sn)) : px * cs - py * sn);
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
            mybits |= (uint32_t)(val[0] < val[1]) << j;
        }
    }
    // Lane li of a group holds the outcomes of the tests li, li + 16, ..., li + 240 in bits 0..15; descriptor halfword j is bit j of
    // the 16 lanes: a 16 x 16 bit-matrix transpose inside the group, four butterfly stages (partner lane ^ s by ds_swizzle, the LDS
    // crossbar; keep half of the own bits, take the other half from the partner shifted by s).  (16 ballots and a 16-way select of
    // SGPR pairs took ~130 instructions; this takes ~30.)
    uint32_t half = mybits;
    auto stage = [&](auto S) {
        constexpr int s = decltype(S)::value;
        constexpr uint32_t M0 = s == 8 ? 0x00FFu : s == 4 ? 0x0F0Fu : s == 2 ? 0x3333u : 0x5555u;     // bit positions with (pos & s) == 0
        const bool hi = (li & s) != 0;
        const uint32_t y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)half, (s << 10) | 0x1F);        // lane ^ s
        const uint32_t ysh = hi ? (y >> s) : (y << s);
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


int launch_describe(const Batch& b, const HostGeom& hg, hipStream_t stream) {
    const DevGeom& g = hg.g;
    const dim3 grid(frame_item_blocks(b, (g.nquads + DESC_WAVES - 1) / DESC_WAVES)), block(DESC_WAVES * 64);
    if (g.fp_contract) hipLaunchKernelGGL(k_describe<true>, grid, block, 0, stream, b);
    else hipLaunchKernelGGL(k_describe<false>, grid, block, 0, stream, b);
    ORBX_LAUNCH_CHECK();
    return ORBX_OK;
}

}  // namespace orbx
