// HIP kernels of the ORB extractor for gfx950 (MI355X, wave64).  One launch group processes a whole
// batch of frames: every kernel's grid spans frames x (levels x tiles | cells | keypoint slots), so
// launch cost is amortised over the batch and the 256 CUs always see >> 256 workgroups.
//
// Stage            reference (under /root/reference/src/ORBextractor.cc)        kernel
//   pyramid        ComputePyramid :781-822 (cv::resize INTER_LINEAR)             k_resize (per level, 7 launches) | k_pyramid (cones of levels, 2 launches; < 32 frames)
//   FAST + NMS     cv::FAST(cell, th, true) :607/:613 + raster-ordered cell lists  k_fast_cells (one workgroup per grid-cell row band)
//   quotas         :622-670                                                      k_quota      (one wave per level; a pass of the rule = one sweep of the lanes + two reductions)
//   retainBest     :683-685 (per cell), :697-701 (per level)                     k_cell_select (+ _long) / k_level_select (wave-parallel, permutation-exact introselect)
//   blur           GaussianBlur 7x7 s=2 :760                                     k_blur       (inside k_fast_blur for < 32 frames)
//   orientation    IC_Angle :124-151, descriptor :155-194, scaling :769-775      k_describe   (one wave per four keypoints)
// The one-frame drop-in call also has k_ingest (the staged frame fetched from pinned host memory by a kernel).
//
// No 16-px border planes exist on the device: the only out-of-image reads of the reference (blur
// taps <= 3 px, rotated BRIEF taps <= 2 px outside the ROI) are served by reflect-101 index math,
// which is what copyMakeBorder(BORDER_REFLECT_101) materialises.
#include <algorithm>

#include "orbx_device.h"
#include "orbx_launch.h"

namespace orbx {

static bool use_on_demand(const Batch& b, const HostGeom& hg, int stop_after) {
    return b.blur_on_demand && b.nframes >= b.od_min_frames && stop_after < 0 && describe_od_supported(b, hg);
}

int launch_extract(const Batch& b, const HostGeom& hg, hipStream_t stream, int stop_after, StageTimer* timer, const SideStream* side, int phases) {
    const DevGeom& g = hg.g;
    const int F = b.nframes;
    if (F <= 0) return ORBX_OK;
    const bool fused_pyramid = g.npyr_groups > 0 && F < PYR_FUSED_MAX_FRAMES;
    // per-frame status starts at ORBX_OK: a fill launch for full batches, folded into the first k_pyramid launch otherwise
    if ((phases & ORBX_PHASE_PYRAMID) && !fused_pyramid && hipMemsetAsync(b.status, 0, sizeof(int32_t) * F, stream) != hipSuccess) return ORBX_ERR_DEVICE;
    if (phases & ORBX_PHASE_PYRAMID) {
        StageScope sc(timer, stream, ST_PYRAMID);
        if (launch_pyramid(b, hg, stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    }
    if (stop_after == ST_PYRAMID) return ORBX_OK;
    if (!(phases & ORBX_PHASE_DETECT)) {
        if (stop_after >= 0 && stop_after < ST_DESCRIBE) return ORBX_OK;      // the diagnostics' early stop lies inside the part this call skips
        goto describe;
    }
    {
    auto blur_stage = [&](hipStream_t st) -> int {
        StageScope sc(timer, st, ST_BLUR);
        return launch_blur(b, hg, st);
    };
    // blur on demand (k_describe_od.hip): full launch groups only (the one-frame call keeps the blur inside the FAST launch), never under the
    // diagnostics' early stops (they fetch the blurred plane)
    const bool on_demand = use_on_demand(b, hg, stop_after);
    // (a launch group that cannot fill the chip keeps the blur in line: its short strips take ~6 us, the fork and the join across
    //  two hardware queues cost 8 us each)
    const bool overlap = side && side->aux && stop_after < 0 && F >= PYR_FUSED_MAX_FRAMES && !on_demand;
    const bool fuse_blur = F < PYR_FUSED_MAX_FRAMES && !b.xcd_affinity && !on_demand;    // k_fast_blur
    {
        StageScope sc(timer, stream, ST_FAST_CELLS);
        if (launch_fast(b, hg, stream, fuse_blur) != ORBX_OK) return ORBX_ERR_DEVICE;
    }
    if (stop_after == ST_FAST_CELLS) return ORBX_OK;
    if (overlap) {
        // fork: the VALU-bound blur runs on the side stream next to the latency-bound quota / retainBest kernels
        if (hipEventRecord(side->fork, stream) != hipSuccess || hipStreamWaitEvent(side->aux, side->fork, 0) != hipSuccess) return ORBX_ERR_DEVICE;
        if (blur_stage(side->aux) != ORBX_OK) return ORBX_ERR_DEVICE;
        if (hipEventRecord(side->join, side->aux) != hipSuccess) return ORBX_ERR_DEVICE;
    }
    {
        StageScope sc(timer, stream, ST_QUOTA);
        if (launch_quota(b, hg, stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    }
    if (stop_after == ST_QUOTA) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_CELL_SELECT);
        if (launch_cell_select(b, hg, stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    }
    if (stop_after == ST_CELL_SELECT) return ORBX_OK;
    {
        StageScope sc(timer, stream, ST_LEVEL_SELECT);
        if (launch_level_select(b, hg, stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    }
    if (stop_after == ST_LEVEL_SELECT) return ORBX_OK;
    if (overlap) {
        if (hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) return ORBX_ERR_DEVICE;
    } else if (!fuse_blur && !on_demand && blur_stage(stream) != ORBX_OK) return ORBX_ERR_DEVICE;
    if (stop_after == ST_BLUR) return ORBX_OK;
    }
describe:
    if (phases & ORBX_PHASE_DESCRIBE) {
        StageScope sc(timer, stream, ST_DESCRIBE);
        if ((use_on_demand(b, hg, stop_after) ? launch_describe_od(b, hg, stream) : launch_describe(b, hg, stream)) != ORBX_OK) return ORBX_ERR_DEVICE;
    }
    return ORBX_OK;
}

// The one-frame drop-in call (orbx_extract): the frame is fetched from the pinned, device-mapped staging buffer by a kernel instead
// of a DMA copy (a copy -> kernel dependency costs ~9 us on top of the copy's 15 us for a VGA frame; 300 waves with one 16-byte
// load each in flight pull the 300 KB over PCIe in less, and the next kernel follows without a queue switch).
__global__ __launch_bounds__(256) void k_ingest(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
int launch_ingest(uint8_t* d_dst, const uint8_t* mapped_src, size_t bytes, hipStream_t stream) {
    const int n16 = (int)(bytes / 16);
    hipLaunchKernelGGL(k_ingest, dim3((n16 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const uint4*>(mapped_src), reinterpret_cast<uint4*>(d_dst), n16);
    ORBX_LAUNCH_CHECK();
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

