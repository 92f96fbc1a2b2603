/* orbx.h — C ABI of the MI355X-native ORB front-end (liborbx.so).
 *
 * Drop-in boundary for ORB_SLAM's per-frame hot path.  Every entry point uses plain pointers and
 * sizes only (no C++/torch types) so the reference's C++ classes — or any FFI — can bind it.
 * The reference-side shim that keeps `ORB_SLAM::ORBextractor` / `ORB_SLAM::ORBmatcher` source
 * compatible lives in orb_slam_amd/cpp/ (see INTEGRATION.md).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   orbx_create / orbx_destroy      <- ORBextractor::ORBextractor(int,float,int,int,int)   include/ORBextractor.h:38, src/ORBextractor.cc:457-511
 *   orbx_extract                    <- ORBextractor::operator()(InputArray,InputArray,vector<KeyPoint>&,OutputArray)
 *                                                                                           include/ORBextractor.h:43-45, src/ORBextractor.cc:718-779
 *   orbx_get_levels                 <- ORBextractor::GetLevels()                           include/ORBextractor.h:47-48
 *   orbx_get_scale_factor           <- ORBextractor::GetScaleFactor()                      include/ORBextractor.h:50-51
 *   orbx_extract_batch_device       <- the same operator(), throughput form (device-resident frames, many per call)
 *   orbm_hamming256                 <- ORBmatcher::DescriptorDistance(const Mat&,const Mat&) include/ORBmatcher.h:44, src/ORBmatcher.cc:1794-1810
 *   orbm_match_top2[_masked][_device|_batch_device]
 *                                   <- the best / second-best scan shared by every ORBmatcher search
 *                                      (src/ORBmatcher.cc:87-111, :201-222, :454-474, :629-650, ...)
 *   orbm_match_top2_segments[_device]
 *                                   <- the same scan over a per-query candidate list (window / vocabulary node): "next" row N2
 *   orbm_distinctive[_device]       <- MapPoint::ComputeDistinctiveDescriptors(): N x N Hamming distances of a map point's observed
 *                                      descriptors, the one with the least median distance to the rest   src/MapPoint.cc:185-250 ("next" row N4)
 *   orbm_count_accepted             <- accept rule `best<=TH && (float)best < mfNNratio*(float)second` (src/ORBmatcher.cc:224-226)
 *
 * All compute runs in hand-written HIP kernels for gfx950.  There is NO CPU fallback: every
 * compute entry point returns ORBX_ERR_DEVICE when no usable GPU / kernel image is present.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define ORBX_OK             0
#define ORBX_EMPTY          1   /* empty image: nothing done, outputs untouched (reference src/ORBextractor.cc:721-722) */
#define ORBX_ERR_ARG       -1
#define ORBX_ERR_DEVICE    -2   /* no GPU, HIP error, or kernel image missing */
#define ORBX_ERR_CAPACITY  -3   /* caller buffer or internal list too small; or an implementation limit the reference does not have
                                 * (a grid cell wider than 6500 px, > 16384 cells on a level — far outside ORB_SLAM's settings) */
#define ORBX_ERR_GEOMETRY  -4   /* image/grid geometry the reference itself cannot process (cv::Exception / div-by-zero there) */

/* ORBextractor::{HARRIS_SCORE, FAST_SCORE} (include/ORBextractor.h:36) */
#define ORBX_HARRIS_SCORE 0
#define ORBX_FAST_SCORE   1

/* GaussianBlur column rounding: what an x86-64 OpenCV 2.4 does (SSE2 column filter:
 * ties-to-even for x < (w & ~3), half-up for the scalar tail) or half-up everywhere (non-SIMD build) */
#define ORBX_BLUR_X86_SSE2 0
#define ORBX_BLUR_HALF_UP  1

/* How the two float expressions of src/ORBextractor.cc that a compiler may contract are evaluated:
 * `x*b + y*a`, `x*a - y*b` in computeOrbDescriptor (:165-166) and the Harris response (:118-119).
 *   ORBX_FP_ISO           unfused, one rounding per operation: the source text under ISO C++ rules = the reference built with
 *                         -ffp-contract=off, or for a CPU without FMA instructions (default; oracle/_ref, the golden fixtures)
 *   ORBX_FP_GCC_CONTRACT  fused where GCC fuses them under the reference's own flags (CMakeLists.txt:12-13: -O3 -march=native with
 *                         GCC's default -ffp-contract=fast, any x86 since 2013): fma(x,b,y*a), fma(x,a,-(y*b)),
 *                         fma(-(a+b), k*(a+b), fma(a,b,-(c*c))) — read off that build's object code (oracle/_ref_native).
 * The two differ in ~1 descriptor bit per 10^5 key points and in the last bits of ~30 % of the Harris responses
 * (tests/test_ref_pin_native.py).  orbx_default_params takes ORBX_FP_CONTRACT=1 from the environment. */
#define ORBX_FP_ISO          0
#define ORBX_FP_GCC_CONTRACT 1

/* identical to OpenCV 2.4's cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct orbx_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} orbx_keypoint;

typedef struct orbx_params {
    /* the five reference constructor arguments, same meaning and defaults (1000, 1.2f, 8, FAST_SCORE, 20) */
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;
    int32_t score_type;
    int32_t fast_th;
    /* device-side additions */
    int32_t device;         /* HIP device ordinal */
    int32_t max_batch;      /* frames processed per launch group by orbx_extract_batch_device (>=1) */
    int32_t blur_rounding;  /* ORBX_BLUR_* */
    int32_t fp_contract;    /* ORBX_FP_* (took the first reserved slot: a zeroed struct of an older caller means ORBX_FP_ISO) */
    int32_t reserved[7];    /* must be zero */
} orbx_params;

typedef struct orbx_extractor orbx_extractor;   /* opaque; not thread-safe (same as the reference instance) */

void  orbx_default_params(orbx_params* p);
int   orbx_create(const orbx_params* p, orbx_extractor** out);
void  orbx_destroy(orbx_extractor* h);
int   orbx_get_levels(const orbx_extractor* h);
float orbx_get_scale_factor(const orbx_extractor* h);
/* upper bound on keypoints per frame (= sum of the per-level quotas; == nfeatures for sane parameters) */
int   orbx_max_keypoints(const orbx_extractor* h);
const char* orbx_last_error(const orbx_extractor* h);
/* 16 hex digits: hash of the kernel sources this library was built from (the Makefile passes it in).  bench.py compares it with
 * the hash recorded in profiles/traffic.json / valu_mix.json and refuses to replay counters measured on other kernels. */
const char* orbx_build_id(void);

/* One frame, host buffers (the operator() drop-in).  img: 8-bit single channel, `stride` bytes per row.
 * kps[cap], desc[cap*32] are caller buffers, cap >= orbx_max_keypoints().  *n_out = number of features.
 * Keypoint order, coordinates, angle, response, octave and descriptors are bit-identical to the
 * reference algorithm (see DESIGN.md "Parity").  Synchronous. */
int orbx_extract(orbx_extractor* h, const uint8_t* img, int w, int hgt, ptrdiff_t stride,
                 orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Throughput form.  d_imgs: DEVICE pointer to nframes frames, frame f at d_imgs + f*frame_stride,
 * rows `row_stride` bytes apart.  d_kps[nframes*cap], d_desc[nframes*cap*32], d_n[nframes]: DEVICE
 * buffers.  Work is enqueued on `stream` (a hipStream_t, may be NULL) and NOT synchronised.
 * d_status[nframes] (optional DEVICE int32 buffer) receives ORBX_OK / ORBX_ERR_CAPACITY per frame.
 * Pitched frames (row_stride > w): every row, the last row of the last frame included, must be readable up to
 * min(row_stride, w rounded up to 16) bytes — the kernels stage rows in whole 16-byte chunks (as a hipMallocPitch-style
 * buffer of row_stride x hgt bytes per frame guarantees).  Fastest when d_imgs, row_stride and frame_stride are multiples of
 * 16 (any alignment is accepted: bytes are then assembled on the fly). */
/* Temporal coherence is used for SPEED only, per frame slot: slot f of a launch group (frame f0 + f of a call, f < max_batch) remembers which
 * of its grid-cell bands took the reference's threshold-7 fallback (src/ORBextractor.cc:609-614) in the last launch groups and starts those at 7
 * right away.  Outputs never depend on it; a caller that keeps each camera / sequence in the same slots of consecutive calls gets the fast pass,
 * a slot whose content changes class pays one slower pass per band for a few calls.  ORBX_FALLBACK_HINT=0 at orbx_create switches it off. */
int orbx_extract_batch_device(orbx_extractor* h, const uint8_t* d_imgs, int nframes, int w, int hgt,
                              ptrdiff_t row_stride, ptrdiff_t frame_stride,
                              orbx_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap,
                              int32_t* d_status, void* stream);

/* The same call cut into the three parts a host can interleave with other work on other streams (tools/corun_probe.py times
 * them alone and in pairs; the bench's lanes queue whole calls — NOTES.md 8.1):
 *   ORBX_PHASE_PYRAMID   ComputePyramid                                              (src/ORBextractor.cc:781-822)
 *   ORBX_PHASE_DETECT    FAST + NMS per cell, quotas, retainBest, GaussianBlur       (:527-707, :760)
 *   ORBX_PHASE_DESCRIBE  IC_Angle, rBRIEF, scaling, outputs                          (:124-194, :709-779)
 * `phases` is a bit mask of CONSECUTIVE parts; the parts of one batch must be queued in this order on ONE stream with identical
 * arguments, and nframes <= max_batch (the handle's scratch holds one launch group).  A call whose parts are not consecutive
 * (PYRAMID | DESCRIBE) or that queues a part before the parts in front of it were queued for the same batch returns
 * ORBX_ERR_ARG; repeating a part is allowed.  ORBX_PHASE_ALL == orbx_extract_batch_device. */
#define ORBX_PHASE_PYRAMID  1
#define ORBX_PHASE_DETECT   2
#define ORBX_PHASE_DESCRIBE 4
#define ORBX_PHASE_ALL      7
int orbx_extract_batch_device_phases(orbx_extractor* h, const uint8_t* d_imgs, int nframes, int w, int hgt,
                                     ptrdiff_t row_stride, ptrdiff_t frame_stride,
                                     orbx_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap,
                                     int32_t* d_status, void* stream, int phases);

/* ---- matcher ---------------------------------------------------------------------------------- */
/* Hamming distance of two 256-bit descriptors (pure, re-entrant, host). */
int orbm_hamming256(const uint8_t* a, const uint8_t* b);

/* For each of nq query descriptors (32 B each) scan all nt train descriptors:
 *   best[q], second[q] = the two smallest distances WITH multiplicity; best_idx[q] = FIRST index
 *   attaining best (strict '<' update order of the reference loops); nt==0 -> -1, INT_MAX, INT_MAX.
 * Host-pointer form (copies in/out, synchronous) and device-pointer form (async on `stream`: its scratch — the partial
 * results of the train splits — is allocated and freed in stream order from a memory pool the library creates for itself
 * (hipMallocFromPoolAsync; the device's default pool is left as the host set it), so concurrent calls on different streams,
 * also from one host thread, never share a buffer).  nt < 2^22.
 * The kernels compute the distances on the matrix cores, exactly: +-1 encoded bits through CDNA4's block-scaled FP4 MFMA
 * (v_mfma_scale_f32_32x32x64_f8f6f4, f32 accumulate; the process default since round 5) or through the int8 MFMA
 * (ORBX_MATCH_MFMA=8 in the environment); ORBX_MATCH_MFMA=0 selects the xor + popcount kernels instead, ORBX_MATCH_MFMA=4 names
 * the default; orbm_debug_set_match_path() switches at run time (same results; tests/test_gpu_matcher.py runs every case
 * through all three).  The FP4 form keeps train indices in 15 bits per scan: batch calls with cap > 32768 take the int8 form.
 * Alignment: 4 bytes for every descriptor array of every matcher entry point — what the reference's DescriptorDistance needs, which
 * reads a descriptor as 8 x int32 (src/ORBmatcher.cc:1796-1801).  ORBX_ERR_ARG otherwise. */
int orbm_match_top2(const uint8_t* Q, int nq, const uint8_t* T, int nt,
                    int32_t* best_idx, int32_t* best, int32_t* second, int device);
int orbm_match_top2_device(const uint8_t* dQ, int nq, const uint8_t* dT, int nt,
                           int32_t* d_best_idx, int32_t* d_best, int32_t* d_second, void* stream);
/* The same scan over the train descriptors with t_valid[t] != 0 only (one byte per train descriptor; NULL = all): the
 * `if(vpMapPointMatches[realIdxF]) continue;` of the reference's scans (src/ORBmatcher.cc:205-206).  best_idx is an index into the
 * ORIGINAL train array; no valid descriptor -> -1, INT_MAX, INT_MAX.  nq, nt < 2^22. */
int orbm_match_top2_masked(const uint8_t* Q, int nq, const uint8_t* T, int nt, const uint8_t* t_valid,
                           int32_t* best_idx, int32_t* best, int32_t* second, int device);
int orbm_match_top2_masked_device(const uint8_t* dQ, int nq, const uint8_t* dT, int nt, const uint8_t* d_t_valid,
                                  int32_t* d_best_idx, int32_t* d_best, int32_t* d_second, void* stream);
/* nbatch independent problems with per-problem sizes read on the device:
 * problem i: queries dQ + i*cap*32 (d_nq[i] of them), train dT + i*cap*32 (d_nt[i]); outputs at i*cap. */
int orbm_match_top2_batch_device(const uint8_t* dQ, const int32_t* d_nq, const uint8_t* dT, const int32_t* d_nt,
                                 int nbatch, int cap, int32_t* d_best_idx, int32_t* d_best, int32_t* d_second,
                                 void* stream);
/* Candidate-set form — the shape every ORBmatcher search actually scans (GetFeaturesInArea windows,
 * src/Frame.cc:200-265; the features of one vocabulary node, src/ORBmatcher.cc:171-260): query q scans the train
 * descriptors with indices cand[seg_off[q] .. seg_off[q+1]) in LIST order; best_idx = the first listed candidate
 * attaining the best distance (a train index), -1 / INT_MAX / INT_MAX for an empty list.  seg_off has nq+1 entries,
 * non-decreasing, every segment shorter than 2^22 candidates (the key keeps the list position in 22 bits); candidate
 * indices outside [0, nt) are skipped. */
int orbm_match_top2_segments(const uint8_t* Q, int nq, const uint8_t* T, int nt, const int32_t* seg_off, const int32_t* cand,
                             int32_t* best_idx, int32_t* best, int32_t* second, int device);
int orbm_match_top2_segments_device(const uint8_t* dQ, int nq, const uint8_t* dT, int nt, const int32_t* d_seg_off,
                                    const int32_t* d_cand, int32_t* d_best_idx, int32_t* d_best, int32_t* d_second, void* stream);
/* number of queries passing  best <= th && (float)best < ratio*(float)second  (host arrays) */
int orbm_count_accepted(const int32_t* best, const int32_t* second, int nq, int th, float ratio);
/* test hook: which kernels the dense top-2 calls use from now on in this process: -1 = default (environment), 0 = xor + popcount,
 * 1 = int8 MFMA, 2 = FP4 MFMA; orbm_debug_get_match_path() = the path in effect (0 / 1 / 2) */
int orbm_debug_set_match_path(int path);
int orbm_debug_get_match_path(void);

/* Device-memory helpers for hosts that do not want to include the HIP headers (a C or C++ translation unit of ORB_SLAM can
 * keep frames, keypoints, descriptors and the search structures resident on the GPU with these four calls and chain the
 * *_device entry points of orbx.h / orbf.h / orbv.h / orbs.h; see orb_slam_amd/cpp/example_pipeline.cpp).  Synchronous. */
int orbx_device_alloc(int device, size_t bytes, void** d_ptr);       /* zero-initialised */
int orbx_device_free(int device, void* d_ptr);
int orbx_device_upload(int device, void* d_dst, const void* src, size_t bytes);
int orbx_device_download(int device, void* dst, const void* d_src, size_t bytes);      /* waits for all queued work of the device */

/* Stream / event helpers in the same spirit (opaque HIP handles): what a host needs to run several extractor handles
 * concurrently and order them against each other — the lanes of orb_slam_amd/cpp/LanePipeline.h — without HIP headers. */
int orbx_stream_create(int device, void** stream);                  /* a non-blocking stream */
int orbx_stream_create_priority(int device, int priority, void** stream);   /* hipStreamCreateWithPriority(non-blocking, priority) */
int orbx_stream_destroy(int device, void* stream);
int orbx_stream_synchronize(int device, void* stream);              /* stream == NULL: everything queued on the device */
int orbx_event_create(int device, void** event);                    /* ordering only (no timing) */
int orbx_event_destroy(int device, void* event);
int orbx_event_record(void* event, void* stream);
int orbx_stream_wait_event(void* stream, void* event);
int orbx_device_copy_async(void* d_dst, const void* d_src, size_t bytes, void* stream);    /* device to device */
/* Pinned host memory and stream-ordered copies: what a host-side caller with small per-call inputs needs to keep one search at
 * two copies + its launches + ONE synchronisation (orb_slam_amd/cpp/ORBmatcher.cc stages every search through one pinned block).
 * h_src / h_dst should come from orbx_host_alloc (pageable memory works but the copy then blocks the calling thread). */
int orbx_host_alloc(int device, size_t bytes, void** h_ptr);        /* hipHostMalloc */
int orbx_host_free(int device, void* h_ptr);
int orbx_device_upload_async(void* d_dst, const void* h_src, size_t bytes, void* stream);
int orbx_device_download_async(void* h_dst, const void* d_src, size_t bytes, void* stream);

/* MapPoint::ComputeDistinctiveDescriptors for M map points at once (src/MapPoint.cc:216-244): point p owns the descriptors
 * [seg_off[p], seg_off[p+1]) of `desc`; best_idx[p] = the index INSIDE its segment of the descriptor whose sorted row of
 * distances (self included) has the smallest element at position (int)(0.5*(N-1)) — first such row on ties —, best_median[p]
 * that value; -1 / INT_MAX for an empty segment. */
int orbm_distinctive(const uint8_t* desc, const int32_t* seg_off, int npoints, int32_t* best_idx, int32_t* best_median, int device);
int orbm_distinctive_device(const uint8_t* d_desc, const int32_t* d_seg_off, int npoints, int32_t* d_best_idx,
                            int32_t* d_best_median, void* stream);

/* ---- diagnostics (stage dumps for the parity tests; not part of the drop-in surface) ------------ */
#define ORBX_DBG_PLANE      0   /* unblurred level plane, tight w*h bytes */
#define ORBX_DBG_BLUR       1   /* blurred level plane, tight w*h bytes */
#define ORBX_DBG_NMS        2   /* per-pixel FAST score of NMS survivors (0 elsewhere), tight w*h bytes */
#define ORBX_DBG_LEVEL_KPS  3   /* selected keypoints of a level before orientation: int32 triples (x,y,response bits) */
#define ORBX_DBG_BANDS      4   /* the FAST work items (row bands of grid cells) of a level: int32[8] each = x0, x1 (the cell's columns), y0, y1
                                 * (the rows the band owns), survivors listed, of them with score >= fastTh, with score >= 7, and the threshold
                                 * the band's list was made at (fastTh, or 7 for a band that kept <= 3 survivors at fastTh) */
/* run the kernel sequence only up to `stage` (0 pyramid, 1 FAST + NMS + cell lists, 2 quotas, 3 per-cell retainBest,
 * 4 per-level cap, 5 blur, 6 describe); <0 = everything (default) */
int orbx_debug_set_stop_after(orbx_extractor* h, int stage);
/* 1: full launch groups compute the GaussianBlur per keypoint window inside the description kernel (k_describe_od: no blurred plane, no blur
 * kernel); 0: the blur kernels + k_describe.  Same outputs.  Default: ORBX_BLUR_ON_DEMAND in the environment at orbx_create, else the build's
 * choice.  ORBX_DBG_BLUR planes exist only in mode 0. */
int orbx_debug_set_blur_on_demand(orbx_extractor* h, int mode);
/* per-stage GPU time from HIP events recorded on the launch stream: enable = 0 off, 1 on, 2 on + reset totals.
 * orbx_debug_stage_time synchronises the device and returns the accumulated ms / launch-group count of a stage. */
int orbx_debug_stage_timing(orbx_extractor* h, int enable);
int orbx_debug_stage_time(orbx_extractor* h, int stage, double* total_ms, long* launches);
int orbx_debug_level_size(const orbx_extractor* h, int level, int* w, int* hgt);
/* copies stage data of `frame` (index inside the last batch) / `level` into host memory; returns bytes or <0 */
long orbx_debug_fetch(orbx_extractor* h, int what, int frame, int level, void* host_out, long cap_bytes);
/* evaluate the device arithmetic on arrays (kind 0: fast_atan2(in0,in1)->out0; kind 1: sincos(in0)->out0,out1) */
/* host-only (needs no GPU): the geometry the library derives for a w x h image — per level 8 ints
 * {w, h, quota, grid_cols, grid_rows, cell_w, cell_h, n_bands}; returns the number of levels or an ORBX_ERR_* code */
int orbx_debug_geometry(const orbx_params* p, int w, int hgt, int32_t* out, int cap_levels);
/* the wave-parallel std::nth_element (greater-by-response) used by the retainBest kernels, on one list:
 * out_perm[i] = original index of the element that ends up at position i (n <= 13000) */
int orbx_debug_nth_element(const float* resp, int n, int nth, int32_t* out_perm, int device);
int orbx_debug_eval_math(int kind, const float* in0, const float* in1, float* out0, float* out1, int n, int device);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
