// Host-side pieces the kernel translation units share: the launch check, the per-stage timers' scope, and the per-stage launchers that
// launch_extract (orbx_launch.hip) strings together.  One translation unit per stage since round 6 (k_pyramid.hip, k_fast.hip, k_select.hip,
// k_blur.hip, k_describe.hip, k_describe_od.hip); orbx_build_id() hashes all of them.
#pragma once
#include <hip/hip_runtime.h>

#include "orbx_internal.h"

namespace orbx {

#define ORBX_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e_ = hipGetLastError();                    \
        if (e_ != hipSuccess) return ORBX_ERR_DEVICE;         \
    } while (0)

constexpr int PYR_FUSED_MAX_FRAMES = 32;      // launch groups below this take the latency-oriented sequence: fused pyramid cones, FAST + blur in one launch, one selection class

struct StageScope {   // records (start, stop) events around one stage when timing is on
    StageTimer* t; hipStream_t s; int stage;
    StageScope(StageTimer* t_, hipStream_t s_, int stage_) : t(t_ && t_->enabled ? t_ : nullptr), s(s_), stage(stage_) {
        if (t) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); t->pool.push_back(e); } else t = nullptr; }
    }
    ~StageScope() {
        if (t) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); t->pool.push_back(e); t->pool_stage.push_back(stage); } else { (void)hipEventDestroy(t->pool.back()); t->pool.pop_back(); } }
    }
};

int launch_pyramid(const Batch& b, const HostGeom& hg, hipStream_t stream);                    // k_pyramid.hip
int launch_fast(const Batch& b, const HostGeom& hg, hipStream_t stream, bool fuse_blur);       // k_fast.hip
int launch_quota(const Batch& b, const HostGeom& hg, hipStream_t stream);                      // k_select.hip
int launch_cell_select(const Batch& b, const HostGeom& hg, hipStream_t stream);
int launch_level_select(const Batch& b, const HostGeom& hg, hipStream_t stream);
int launch_blur(const Batch& b, const HostGeom& hg, hipStream_t stream);                       // k_blur.hip
int launch_describe(const Batch& b, const HostGeom& hg, hipStream_t stream);                   // k_describe.hip
// (k_describe_od.hip: describe_od_supported, launch_describe_od — orbx_internal.h)

}  // namespace orbx
