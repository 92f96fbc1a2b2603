// Device-side helpers shared by the kernel translation units (k_pyramid.hip, k_fast.hip, k_select.hip, k_blur.hip, k_describe.hip, k_describe_od.hip): block -> (frame, item) maps, wave-level
// reductions, the BRIEF pattern, the Gaussian tap strings of the matrix-core blur.  Everything is inline / per-TU static.
#pragma once
#include <hip/hip_runtime.h>

#include "orb_math.h"
#include "orbx_internal.h"

namespace orbx {

static __device__ __constant__ uint32_t c_pattern[256] = {
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

// Frame -> XCD affinity.  A launch deals its workgroups round-robin to the 8 XCDs (block b runs on XCD b % 8: observed dispatch
// rule, used for speed only), each with its own 4 MiB L2.  In frame-major block order the workgroups of ONE frame would be spread
// over all eight L2s, and every L2 would fetch its own copy of the 128-byte sectors that neighbouring cells, strips or keypoint
// windows share.  With `xcd_affinity` a launch's blocks are renumbered so that all work items of frame f run on XCD f % 8:
// block b -> (frame, item) = ((b >> 3) / per_frame * 8 + (b & 7), (b >> 3) % per_frame); the grid is rounded up to whole groups of
// 8 frames and blocks of frames >= nframes exit.  Used from XCD_AFFINITY_MIN_FRAMES frames per launch (small launches want every
// CU, not locality).
__device__ __forceinline__ bool frame_item(const Batch& b, int block, int per_frame, int& frame, int& item) {
    if (b.xcd_affinity) {
        const int slot = block >> 3;
        const int fr = slot / per_frame;
        frame = fr * 8 + (block & 7);
        item = slot - fr * per_frame;
    } else {
        frame = block / per_frame;
        item = block - frame * per_frame;
    }
    return frame < b.nframes;
}
// The same split with the division replaced by a multiply with magic = floor(2^32 / per_frame) + 1 (host: DevGeom::nbands_magic; 0 = one
// item per frame): umulhi(n, magic) is n / per_frame or one more for every 32-bit n (the excess n * (magic * per_frame - 2^32) / (per_frame * 2^32)
// is below n / 2^32 < 1), so one compare corrects it.  Everything is wave-uniform: scalar multiplies instead of the ~20 vector instructions of
// an integer division in front of every wave of a kernel whose waves are short (k_fast_cells).
__device__ __forceinline__ bool frame_item_magic(const Batch& b, unsigned block, unsigned per_frame, unsigned magic, int& frame, int& item) {
    const unsigned slot = b.xcd_affinity ? block >> 3 : block;
    unsigned fr = magic ? __umulhi(slot, magic) : slot;
    if (fr * per_frame > slot) --fr;
    frame = (int)(b.xcd_affinity ? fr * 8u + (block & 7u) : fr);
    item = (int)(slot - fr * per_frame);
    return frame < b.nframes;
}
static inline int frame_item_blocks(const Batch& b, int per_frame) {
    return (b.xcd_affinity ? (b.nframes + 7) / 8 * 8 : b.nframes) * per_frame;
}

constexpr unsigned long long UMAX_NIBBLES = 0x3689ABCDDEEEFFFFull;   // umax[v] for v = 0..15 (15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3)

// inclusive prefix sum over the 64 lanes in six DPP adds (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast15 / row_bcast31
// across them) — no LDS round trips (__shfl_up is a ds_bpermute per step).  Needs all 64 lanes active.
__device__ __forceinline__ int wave_scan_inclusive(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_inclusive(v), 63); }   // all 64 lanes active

// sum over each row of 16 lanes, left in every lane of the row: four rotate-and-add steps (DPP row_ror 8, 4, 2, 1)
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    return v;
}


// p -> (p / cw, p % cw) for p < 65536, cw <= 8192 with full-rate VALU ops only (v_mul_lo/hi_u32 are quarter
// rate): q = trunc((p + 0.5) * (1/cw)), exact for every (p, cw) in that range (exhaustively checked offline).
__device__ __forceinline__ void split_px(int p, int cw, float inv_cw, int& y, int& x) {
    y = (int)(((float)p + 0.5f) * inv_cw);
    x = p - (int)__umul24((unsigned)y, (unsigned)cw);
}

typedef unsigned short us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2v as_us2v(uint32_t v) { return __builtin_bit_cast(us2v, v); }



// a wave's own LDS traffic in program order (the areas it guards are private to one wave: no workgroup barrier)
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ int gauss7_tap(int t) {      // [18, 34, 49, 55, 49, 34, 18][t], 0 outside
    return (unsigned)t <= 6u ? (int)((0x12223137312212ull >> (8 * t)) & 255ull) : 0;
}
__device__ __forceinline__ int gauss7_taps4(int t0) {   // the bytes tap(t0), tap(t0 + 1), tap(t0 + 2), tap(t0 + 3): a window of the tap string
    const unsigned long long taps = 0x12223137312212ull;
    const int sh = 8 * min(max(t0, -4), 7);             // |shift| <= 56 bits; beyond that the window is empty anyway
    return (int)(uint32_t)(t0 >= 0 ? taps >> sh : taps << -sh);
}


}  // namespace orbx
