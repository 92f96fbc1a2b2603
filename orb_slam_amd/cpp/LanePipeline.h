// LanePipeline — the throughput configuration of the ORB front-end for C++ hosts (NOTES.md §4.5), written against the C ABI
// only (include/orbx.h): no HIP headers, no torch.  The Python twin is orb_slam_amd/pipeline.py.
//
// A step's B consecutive frames are cut into G lanes of B/G consecutive frames.  Every lane owns an extractor handle
// (orbx_create with max_batch = B/G), a stream and its output buffers, extracts its slice (orbx_extract_batch_device) and
// matches every frame against its predecessor (orbm_match_top2_batch_device — the best/second-best scan all ORBmatcher
// searches share, src/ORBmatcher.cc:201-222).  The one frame per lane whose predecessor lies in the lane to its left (lane 0:
// in the last lane's slice of the previous step) receives that frame's descriptors through a two-slot hand-off buffer ordered
// by events; otherwise lanes never wait for each other, so the latency-bound kernels of one lane run next to the VALU-bound
// kernels of another and consecutive steps overlap.
#ifndef ORB_SLAM_AMD_LANE_PIPELINE_H
#define ORB_SLAM_AMD_LANE_PIPELINE_H

#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbx.h"

namespace ORB_SLAM {

class LanePipeline {
public:
    struct Lane {
        orbx_extractor* ex = nullptr;
        void* stream = nullptr;
        orbx_keypoint* kps = nullptr;      // [b][cap]
        uint8_t* desc = nullptr;           // [b + 1][cap][32]; slot 0 = the frame before the slice
        int32_t* n = nullptr;              // [b + 1]
        int32_t* status = nullptr;         // [b]
        int32_t* match = nullptr;          // [3][b][cap]: train index in the previous frame, best, second-best distance
        uint8_t* h_desc = nullptr;         // hand-off of the slice's last frame: [2][cap][32] ...
        int32_t* h_n = nullptr;            // ... and its count [2]
        void* h_written[2] = {nullptr, nullptr};
        void* h_consumed[2] = {nullptr, nullptr};
        bool consumed_valid[2] = {false, false};
    };

    LanePipeline(int width, int height, int batch, int lanes, const orbx_params& params, bool do_match = true, bool autotune = true,
                 int placement = -1)
        : w_(width), h_(height), B_(batch), device_(params.device), do_match_(do_match) {
        if (placement < 0) {
            const char* e = std::getenv("ORBX_LANE_PLACEMENT");
            if (e && *e) {                                  // a candidate index 0..2, parsed as orb_slam_amd/pipeline.py parses it: anything else is an error
                char* end = nullptr;
                const long v = std::strtol(e, &end, 10);
                if (end == e || *end != '\0' || v < 0 || v > 2) throw std::invalid_argument(std::string("ORBX_LANE_PLACEMENT=") + e + ": expected a candidate index 0..2");
                placement = (int)v;
            }
        }
        if (placement > 2) throw std::invalid_argument("lane placement: expected a candidate index 0..2");
        G_ = lanes < 1 ? 1 : (lanes > batch ? batch : lanes);
        while (B_ % G_) --G_;
        b_ = B_ / G_;
        lanes_.resize(G_);
        // Stream placement.  The HIP runtime binds a stream to one of its hardware queues (GPU_MAX_HW_QUEUES, 4 by default) when
        // the stream is created: new queues until 4 exist, then the least-loaded one.  Streams on one hardware queue are
        // launched in order.  Measured best (NOTES.md §4.5, rocprofv3 Queue_Id column): every lane stream on a hardware queue of
        // its own, the blur side streams (created inside the extractor handles) sharing those queues.  Creating the G handles
        // first and the G lane streams after them gives that placement in a fresh process, but streams other libraries created
        // earlier shift it: three candidate sets of lane streams are created (behind 0, 1 and 2 spacer streams); tune() — an
        // explicit, blocking call — times them and keeps the fastest (placement_ms(), placement_chosen()); step() never probes.
        // `placement` >= 0 (or ORBX_LANE_PLACEMENT=k in the environment) fixes candidate k: no probe (e.g. for 8 ranks tuned once).
        orbx_params p = params;
        p.max_batch = b_;
        for (Lane& L : lanes_) check(orbx_create(&p, &L.ex), "orbx_create");
        const int ncand = ((autotune || placement >= 0) && G_ > 1) ? 3 : 1;
        sets_.resize(ncand);
        for (int k = 0; k < ncand; ++k) {
            for (int sp = 0; sp < k; ++sp) { void* st = nullptr; check(orbx_stream_create(device_, &st), "orbx_stream_create"); spacers_.push_back(st); }
            sets_[k].resize(G_);
            for (int g = 0; g < G_; ++g) check(orbx_stream_create(device_, &sets_[k][g]), "orbx_stream_create");
        }
        chosen_ = placement >= 0 ? (placement < ncand ? placement : ncand - 1) : 0;
        for (int g = 0; g < G_; ++g) lanes_[g].stream = sets_[chosen_][g];
        tuned_ = ncand == 1 || placement >= 0;
        for (Lane& L : lanes_) {
            cap_ = orbx_max_keypoints(L.ex);
            alloc(L.kps, (size_t)b_ * cap_);
            alloc(L.desc, (size_t)(b_ + 1) * cap_ * 32);
            alloc(L.n, (size_t)b_ + 1);
            alloc(L.status, (size_t)b_);
            alloc(L.match, (size_t)3 * b_ * cap_);
            alloc(L.h_desc, (size_t)2 * cap_ * 32);
            alloc(L.h_n, 2);
            for (int s = 0; s < 2; ++s) {
                check(orbx_event_create(device_, &L.h_written[s]), "orbx_event_create");
                check(orbx_event_create(device_, &L.h_consumed[s]), "orbx_event_create");
            }
        }
        check(orbx_stream_synchronize(device_, nullptr), "orbx_stream_synchronize");   // the zero-fills of the allocations are done
    }

    ~LanePipeline() {
        (void)orbx_stream_synchronize(device_, nullptr);
        for (Lane& L : lanes_) {
            for (int s = 0; s < 2; ++s) { (void)orbx_event_destroy(device_, L.h_written[s]); (void)orbx_event_destroy(device_, L.h_consumed[s]); }
            (void)orbx_device_free(device_, L.kps); (void)orbx_device_free(device_, L.desc); (void)orbx_device_free(device_, L.n);
            (void)orbx_device_free(device_, L.status); (void)orbx_device_free(device_, L.match); (void)orbx_device_free(device_, L.h_desc);
            (void)orbx_device_free(device_, L.h_n);
            orbx_destroy(L.ex);
        }
        for (auto& set : sets_) for (void* st : set) (void)orbx_stream_destroy(device_, st);
        for (void* st : spacers_) (void)orbx_stream_destroy(device_, st);
        if (zeros_) (void)orbx_device_free(device_, zeros_);
    }
    LanePipeline(const LanePipeline&) = delete;
    LanePipeline& operator=(const LanePipeline&) = delete;

    // Explicit, BLOCKING placement probe: one untimed + `timed_steps` timed steps over the B frames at d_frames on every candidate
    // stream set, the fastest stays.  Call once after construction, before the real stream starts.  The state the probes touch (step
    // counter, hand-off slots, counts) is reset; only this pipeline's own streams are synchronised (other streams and handles of the
    // process keep running).  No-op with a fixed placement or one lane.
    void tune(const uint8_t* d_frames, ptrdiff_t frame_stride = 0, ptrdiff_t row_stride = 0, int timed_steps = 3) {
        if (tuned_) return;
        if (row_stride == 0) row_stride = w_;
        if (frame_stride == 0) frame_stride = row_stride * h_;
        tuned_ = true;
        probe_ms_.clear();
        for (size_t k = 0; k < sets_.size(); ++k) {
            for (int g = 0; g < G_; ++g) lanes_[g].stream = sets_[k][g];
            reset_handoff();
            step(d_frames, frame_stride, row_stride);
            synchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < timed_steps; ++r) step(d_frames, frame_stride, row_stride);
            synchronize();
            probe_ms_.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / timed_steps);
        }
        chosen_ = 0;
        for (size_t k = 1; k < probe_ms_.size(); ++k) if (probe_ms_[k] < probe_ms_[chosen_]) chosen_ = (int)k;
        for (int g = 0; g < G_; ++g) lanes_[g].stream = sets_[chosen_][g];
        reset_handoff();
    }

    // d_frames: device address of the step's first frame (B frames, frame_stride bytes apart).  Asynchronous; never probes or blocks.
    void step(const uint8_t* d_frames, ptrdiff_t frame_stride = 0, ptrdiff_t row_stride = 0) {
        if (row_stride == 0) row_stride = w_;
        if (frame_stride == 0) frame_stride = row_stride * h_;
        const long i = steps_done_;
        const int par = (int)(i & 1);
        for (int g = 0; g < G_; ++g) {
            Lane& L = lanes_[g];
            check(orbx_extract_batch_device(L.ex, d_frames + (ptrdiff_t)g * b_ * frame_stride, b_, w_, h_, row_stride, frame_stride, L.kps,
                                            L.desc + (size_t)cap_ * 32, L.n + 1, cap_, L.status, L.stream), "orbx_extract_batch_device");
            if (!do_match_) continue;
            // publish the slice's last frame (the slot's reader of step i-2 must be done with it)
            if (L.consumed_valid[par]) check(orbx_stream_wait_event(L.stream, L.h_consumed[par]), "wait consumed");
            check(orbx_device_copy_async(L.h_desc + (size_t)par * cap_ * 32, L.desc + (size_t)b_ * cap_ * 32, (size_t)cap_ * 32, L.stream), "copy");
            check(orbx_device_copy_async(L.h_n + par, L.n + b_, sizeof(int32_t), L.stream), "copy");
            check(orbx_event_record(L.h_written[par], L.stream), "record");
            // take the frame before the slice from the lane on the left (lane 0: the last lane's previous step)
            if (g > 0 || i > 0) {
                Lane& S = g > 0 ? lanes_[g - 1] : lanes_[G_ - 1];
                const int sp = g > 0 ? par : par ^ 1;
                check(orbx_stream_wait_event(L.stream, S.h_written[sp]), "wait written");
                check(orbx_device_copy_async(L.desc, S.h_desc + (size_t)sp * cap_ * 32, (size_t)cap_ * 32, L.stream), "copy");
                check(orbx_device_copy_async(L.n, S.h_n + sp, sizeof(int32_t), L.stream), "copy");
                check(orbx_event_record(S.h_consumed[sp], L.stream), "record");
                S.consumed_valid[sp] = true;
            }
            check(orbm_match_top2_batch_device(L.desc + (size_t)cap_ * 32, L.n + 1, L.desc, L.n, b_, cap_, L.match, L.match + (size_t)b_ * cap_,
                                               L.match + (size_t)2 * b_ * cap_, L.stream), "orbm_match_top2_batch_device");
        }
        ++steps_done_;
    }

    // waits for this pipeline's lane streams (every side stream joins its lane stream before the lane's last kernel), nothing else
    void synchronize() { for (Lane& L : lanes_) check(orbx_stream_synchronize(device_, L.stream), "orbx_stream_synchronize"); }

    // Results of the last step as host arrays in frame order: n[B], kps[B][cap], desc[B][cap][32], match[3][B][cap].
    void download(std::vector<int32_t>& n, std::vector<orbx_keypoint>& kps, std::vector<uint8_t>& desc, std::vector<int32_t>& match) {
        n.assign(B_, 0); kps.assign((size_t)B_ * cap_, orbx_keypoint()); desc.assign((size_t)B_ * cap_ * 32, 0); match.assign((size_t)3 * B_ * cap_, 0);
        for (int g = 0; g < G_; ++g) {
            Lane& L = lanes_[g];
            const size_t f0 = (size_t)g * b_;
            check(orbx_device_download(device_, n.data() + f0, L.n + 1, (size_t)b_ * 4), "download");
            check(orbx_device_download(device_, kps.data() + f0 * cap_, L.kps, (size_t)b_ * cap_ * sizeof(orbx_keypoint)), "download");
            check(orbx_device_download(device_, desc.data() + f0 * cap_ * 32, L.desc + (size_t)cap_ * 32, (size_t)b_ * cap_ * 32), "download");
            for (int k = 0; k < 3; ++k)
                check(orbx_device_download(device_, match.data() + ((size_t)k * B_ + f0) * cap_, L.match + (size_t)k * b_ * cap_, (size_t)b_ * cap_ * 4), "download");
        }
    }

    int lanes() const { return G_; }
    int placement_chosen() const { return chosen_; }
    const std::vector<double>& placement_ms() const { return probe_ms_; }      // per candidate stream set: ms per step of the probe
    int frames_per_lane() const { return b_; }
    int frames_per_step() const { return B_; }
    int cap() const { return cap_; }
    const Lane& lane(int g) const { return lanes_[g]; }

private:
    template <typename T>
    void alloc(T*& p, size_t count) {
        void* v = nullptr;
        check(orbx_device_alloc(device_, count * sizeof(T), &v), "orbx_device_alloc");
        p = static_cast<T*>(v);
    }
    // counts of the slots in front of every slice and the hand-off slots back to "no previous frame" (queued on the lane's own stream)
    void reset_handoff() {
        synchronize();
        steps_done_ = 0;
        if (zeros_ == nullptr) alloc(zeros_, (size_t)b_ + 1);       // device zeros (orbx_device_alloc zero-fills)
        for (Lane& L : lanes_) {
            L.consumed_valid[0] = L.consumed_valid[1] = false;
            check(orbx_device_copy_async(L.n, zeros_, ((size_t)b_ + 1) * 4, L.stream), "copy");
            check(orbx_device_copy_async(L.h_n, zeros_, 2 * 4, L.stream), "copy");
        }
        synchronize();
    }
    static void check(int rc, const char* what) {
        if (rc != ORBX_OK) throw std::runtime_error(std::string(what) + " failed with orbx status " + std::to_string(rc));
    }

    int w_, h_, B_, G_ = 1, b_ = 1, cap_ = 0, device_;
    bool do_match_;
    bool tuned_ = true;
    int chosen_ = 0;
    long steps_done_ = 0;
    std::vector<Lane> lanes_;
    std::vector<std::vector<void*>> sets_;      // candidate lane-stream sets
    std::vector<void*> spacers_;
    std::vector<double> probe_ms_;
    int32_t* zeros_ = nullptr;
};

}  // namespace ORB_SLAM

#endif
