// C ABI of the extractor (include/orbx.h): handle, device memory, launch orchestration.
// All pixel work happens in the k_*.hip kernels (strung together by orbx_launch.hip); there is no host fallback.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "orbx_internal.h"

namespace orbx {
int launch_eval_math(int kind, const float* in0, const float* in1, float* out0, float* out1, int n);
void stage_timer_collect(StageTimer& t);
int launch_debug_nth(const float* d_resp, int n, int nth, int* d_out);
int launch_ingest(uint8_t* d_dst, const uint8_t* mapped_src, size_t bytes, hipStream_t stream);
}
using namespace orbx;

struct orbx_extractor {
    orbx_params p;
    std::string err;
    // geometry for the current image size
    int gw = 0, gh = 0;
    HostGeom hg;
    CellGeom* d_cells = nullptr;
    BandGeom* d_bands = nullptr;
    ResizeX* d_tabx = nullptr;
    ResizeY* d_taby = nullptr;
    int* d_pyr_tab = nullptr;
    // per-batch working set (max_batch frames)
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    Cand *d_cand = nullptr, *d_sel = nullptr;
    CellState* d_cstate = nullptr;
    bool fallback_hint = true;        // ORBX_FALLBACK_HINT=0 at orbx_create
    CellSel* d_csel = nullptr;
    int32_t *d_level_total = nullptr, *d_level_count = nullptr, *d_status = nullptr, *d_long_cells = nullptr;
    // single-frame staging for orbx_extract
    uint8_t* d_img1 = nullptr;
    size_t img1_bytes = 0;
    int img1_stride = 0;
    uint8_t* d_out1 = nullptr;   // one block: [n, status, pad to 64 B][kps: cap x 28 B, padded to 64][desc: cap x 32 B]
    uint8_t* h_out1 = nullptr;   // the same block in pinned host memory, mapped into the device: k_describe writes the results there
    uint8_t* m_out1 = nullptr;   //   (its device address); d_out1 + one D2H copy only with ORBX_ZERO_COPY=0
    uint8_t* h_img1 = nullptr;   // pinned staging of the input frame, mapped into the device (m_img1): fetched by k_ingest
    uint8_t* m_img1 = nullptr;
    bool zero_copy = true;       // ORBX_ZERO_COPY=0 at orbx_create: DMA copies both ways instead (A/B measurements)
    hipStream_t s1 = nullptr;    // stream of the single-frame path
    size_t out1_bytes = 0, kps1_off = 0, desc1_off = 0;
    int out1_cap = 0;
    // diagnostics
    int stop_after = -1;
    bool no_xcd_affinity = false;     // ORBX_XCD_AFFINITY=0 at orbx_create (A/B measurements)
    int od_min_frames = ORBX_OD_MIN_FRAMES_DEFAULT;    // ORBX_OD_MIN_FRAMES at orbx_create (A/B measurements)
    int blur_on_demand = ORBX_BLUR_ON_DEMAND_DEFAULT;   // ORBX_BLUR_ON_DEMAND=0/1 at orbx_create, orbx_debug_set_blur_on_demand
    StageTimer timer;
    SideStream side;
    Batch last;
    bool have_last = false;
    // phased calls (orbx_extract_batch_device_phases): the parts already queued for the batch described by ph_key
    int ph_done = 0;
    struct PhaseKey {
        const void *img, *kps, *desc, *n, *stream;
        int nframes, w, hgt, cap;
        ptrdiff_t row_stride, frame_stride;
        bool operator==(const PhaseKey& o) const {
            return img == o.img && kps == o.kps && desc == o.desc && n == o.n && stream == o.stream && nframes == o.nframes && w == o.w && hgt == o.hgt &&
                   cap == o.cap && row_stride == o.row_stride && frame_stride == o.frame_stride;
        }
    } ph_key = {};
};

#define HIPCHK(h, call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return ORBX_ERR_DEVICE;                                                          \
        }                                                                                    \
    } while (0)

template <typename T>
static void dev_free(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

static void free_geometry(orbx_extractor* h) {
    dev_free(h->d_cells); dev_free(h->d_bands); dev_free(h->d_tabx); dev_free(h->d_taby); dev_free(h->d_pyr_tab);
    dev_free(h->d_pyr); dev_free(h->d_blur);
    dev_free(h->d_cand); dev_free(h->d_sel); dev_free(h->d_cstate); dev_free(h->d_csel);
    dev_free(h->d_level_total); dev_free(h->d_level_count); dev_free(h->d_status); dev_free(h->d_long_cells);
    h->gw = h->gh = 0;
    h->have_last = false;
}

template <typename T>
static int upload(orbx_extractor* h, T*& dptr, const std::vector<T>& v) {
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIPCHK(h, hipMalloc(&dptr, bytes));
    if (!v.empty()) HIPCHK(h, hipMemcpy(dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return ORBX_OK;
}

static int ensure_geometry(orbx_extractor* h, int w, int hgt) {
    if (h->gw == w && h->gh == hgt) return ORBX_OK;
    HIPCHK(h, hipDeviceSynchronize());
    free_geometry(h);
    HostGeom hg;
    int rc = build_geometry(h->p, w, hgt, hg, h->err);
    if (rc != ORBX_OK) return rc;
    h->hg = hg;
    const DevGeom& g = h->hg.g;
    const size_t B = (size_t)h->p.max_batch;
    if ((rc = upload(h, h->d_cells, h->hg.cells)) != ORBX_OK) return rc;
    if ((rc = upload(h, h->d_bands, h->hg.bands)) != ORBX_OK) return rc;
    if ((rc = upload(h, h->d_tabx, h->hg.tabx)) != ORBX_OK) return rc;
    if ((rc = upload(h, h->d_taby, h->hg.taby)) != ORBX_OK) return rc;
    if ((rc = upload(h, h->d_pyr_tab, h->hg.pyr_tab)) != ORBX_OK) return rc;
    HIPCHK(h, hipMalloc(&h->d_pyr, B * g.frame_plane_bytes));
    HIPCHK(h, hipMalloc(&h->d_blur, B * g.frame_plane_bytes));
    HIPCHK(h, hipMalloc(&h->d_cand, B * std::max(g.frame_cands, 1) * sizeof(Cand)));
    HIPCHK(h, hipMalloc(&h->d_sel, (B * std::max(g.frame_sel, 1) + 4) * sizeof(Cand)));      // (+ 4: k_describe reads a wave's four keypoints as one 32-byte scalar load)
    HIPCHK(h, hipMalloc(&h->d_cstate, B * g.nbands_total * sizeof(CellState)));
    HIPCHK(h, hipMemset(h->d_cstate, 0, B * g.nbands_total * sizeof(CellState)));      // (the fallback runs start at 0: k_fast_cells reads its slot's state before it writes it)
    HIPCHK(h, hipMalloc(&h->d_csel, B * g.ncells_total * sizeof(CellSel)));
    HIPCHK(h, hipMalloc(&h->d_level_total, B * MAX_LEVELS * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&h->d_level_count, B * MAX_LEVELS * sizeof(int32_t)));
    HIPCHK(h, hipMemset(h->d_level_total, 0, B * MAX_LEVELS * sizeof(int32_t)));
    HIPCHK(h, hipMemset(h->d_level_count, 0, B * MAX_LEVELS * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&h->d_status, B * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&h->d_long_cells, (B * g.ncells_total + 64) * sizeof(int32_t)));
    HIPCHK(h, hipDeviceSynchronize());
    h->gw = w;
    h->gh = hgt;
    return ORBX_OK;
}

static void fill_batch(orbx_extractor* h, Batch& b) {
    b.g = h->hg.g; b.cells = h->d_cells; b.bands = h->d_bands; b.tabx = h->d_tabx; b.taby = h->d_taby; b.pyr_tab = h->d_pyr_tab;
    b.pyr = h->d_pyr; b.blur = h->d_blur;
    b.cand = h->d_cand; b.sel = h->d_sel; b.cstate = h->d_cstate; b.csel = h->d_csel;
    b.level_total = h->d_level_total; b.level_count = h->d_level_count; b.status = h->d_status; b.long_cells = h->d_long_cells;
}

extern "C" {

void orbx_default_params(orbx_params* p) {
    memset(p, 0, sizeof(*p));
    p->nfeatures = 1000;       // include/ORBextractor.h:38 defaults
    p->scale_factor = 1.2f;
    p->nlevels = 8;
    p->score_type = ORBX_FAST_SCORE;
    p->fast_th = 20;
    p->device = 0;
    p->max_batch = 1;
    p->blur_rounding = ORBX_BLUR_X86_SSE2;
    const char* fc = getenv("ORBX_FP_CONTRACT");           // setup time only (never on a launch path)
    p->fp_contract = (fc && fc[0] == '1' && fc[1] == 0) ? ORBX_FP_GCC_CONTRACT : ORBX_FP_ISO;
}

int orbx_create(const orbx_params* p, orbx_extractor** out) {
    if (!p || !out) return ORBX_ERR_ARG;
    *out = nullptr;
    if (p->max_batch < 1 || p->nlevels < 1 || p->nlevels > MAX_LEVELS || p->nfeatures < 1) return ORBX_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device < 0 || p->device >= ndev) return ORBX_ERR_DEVICE;
    if (hipSetDevice(p->device) != hipSuccess) return ORBX_ERR_DEVICE;
    orbx_extractor* h = new orbx_extractor();
    h->p = *p;
    // The blur runs on a side stream next to the latency-bound selection kernels (see launch_extract);
    // ORBX_OVERLAP=0 keeps everything on one stream (cleaner per-kernel timings when profiling).
    { const char* xa = getenv("ORBX_XCD_AFFINITY"); h->no_xcd_affinity = xa && xa[0] == '0'; }
    { const char* fh = getenv("ORBX_FALLBACK_HINT"); h->fallback_hint = !(fh && fh[0] == '0'); }
    { const char* om = getenv("ORBX_OD_MIN_FRAMES"); if (om && atoi(om) >= 1) h->od_min_frames = atoi(om); }
    { const char* od = getenv("ORBX_BLUR_ON_DEMAND"); if (od && (od[0] == '0' || od[0] == '1') && od[1] == 0) h->blur_on_demand = od[0] - '0'; }
    { const char* zc = getenv("ORBX_ZERO_COPY"); h->zero_copy = !(zc && zc[0] == '0'); }
    const char* ovl = getenv("ORBX_OVERLAP");
    if (ovl && ovl[0] == '0') { *out = h; return ORBX_OK; }
    if (hipStreamCreateWithFlags(&h->side.aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->side.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->side.join, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return ORBX_ERR_DEVICE;
    }
    *out = h;
    return ORBX_OK;
}

void orbx_destroy(orbx_extractor* h) {
    if (!h) return;
    (void)hipSetDevice(h->p.device);
    (void)hipDeviceSynchronize();
    free_geometry(h);
    if (h->side.fork) (void)hipEventDestroy(h->side.fork);
    if (h->side.join) (void)hipEventDestroy(h->side.join);
    if (h->side.aux) (void)hipStreamDestroy(h->side.aux);
    dev_free(h->d_img1); dev_free(h->d_out1);
    if (h->h_img1) (void)hipHostFree(h->h_img1);
    if (h->h_out1) (void)hipHostFree(h->h_out1);
    if (h->s1) (void)hipStreamDestroy(h->s1);
    delete h;
}

int orbx_get_levels(const orbx_extractor* h) { return h ? h->p.nlevels : 0; }
float orbx_get_scale_factor(const orbx_extractor* h) { return h ? (float)(double)h->p.scale_factor : 0.f; }
const char* orbx_last_error(const orbx_extractor* h) { return h ? h->err.c_str() : "null handle"; }
#ifndef ORBX_SRC_HASH
#define ORBX_SRC_HASH "unknown"
#endif
const char* orbx_build_id(void) { return ORBX_SRC_HASH; }

int orbx_max_keypoints(const orbx_extractor* h) {
    if (!h) return 0;
    HostGeom tmp;
    std::string e;
    orbx_params p = h->p;
    // quotas do not depend on the image size; use any valid size
    if (build_geometry(p, 4096, 4096, tmp, e) != ORBX_OK) return std::max(p.nfeatures, 0) + p.nlevels;
    return tmp.g.nslots;
}

int orbx_extract_batch_device(orbx_extractor* h, const uint8_t* d_imgs, int nframes, int w, int hgt, ptrdiff_t row_stride,
                              ptrdiff_t frame_stride, orbx_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap,
                              int32_t* d_status, void* stream_) {
    return orbx_extract_batch_device_phases(h, d_imgs, nframes, w, hgt, row_stride, frame_stride, d_kps, d_desc, d_n, cap, d_status, stream_, ORBX_PHASE_ALL);
}

int orbx_extract_batch_device_phases(orbx_extractor* h, const uint8_t* d_imgs, int nframes, int w, int hgt, ptrdiff_t row_stride,
                                     ptrdiff_t frame_stride, orbx_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap,
                                     int32_t* d_status, void* stream_, int phases) {
    if (!h) return ORBX_ERR_ARG;
    if (!d_imgs || nframes <= 0 || w <= 0 || hgt <= 0) return ORBX_EMPTY;
    if (!d_kps || !d_desc || !d_n || cap < 1 || row_stride < w) { h->err = "bad argument"; return ORBX_ERR_ARG; }
    if (phases < 1 || phases > ORBX_PHASE_ALL || phases == (ORBX_PHASE_PYRAMID | ORBX_PHASE_DESCRIBE)) {
        h->err = "bad phase mask (the parts of one call must be consecutive)";
        return ORBX_ERR_ARG;
    }
    if (phases != ORBX_PHASE_ALL && nframes > h->p.max_batch) { h->err = "a phased call covers one launch group: nframes <= max_batch"; return ORBX_ERR_ARG; }
    // a part may only follow the parts in front of it, queued for the SAME batch (same arguments, same stream): the detection reads the
    // pyramid of this batch from the handle's scratch, the description reads its selections and blurred planes.  Repeating a part is fine.
    // The bookkeeping is committed only when the call has queued its launches (a failed PYRAMID call must not let a DETECT pass the check).
    const orbx_extractor::PhaseKey key = {d_imgs, d_kps, d_desc, d_n, stream_, nframes, w, hgt, cap, row_stride, frame_stride};
    {
        const int first = phases & -phases;                       // lowest part of this call
        const int before = first - 1;                             // every part in front of it
        if (!(phases & ORBX_PHASE_PYRAMID) && (!(key == h->ph_key) || (h->ph_done & before) != before)) {
            h->err = "phase queued out of order: the parts in front of it were not queued for this batch (same arguments, same stream)";
            return ORBX_ERR_ARG;
        }
    }
    if (phases & ORBX_PHASE_PYRAMID) h->ph_done = 0;              // a new batch starts: whatever was queued before no longer counts
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = ensure_geometry(h, w, hgt);
    if (rc != ORBX_OK) return rc;
    if (cap < h->hg.g.nslots) { h->err = "cap < orbx_max_keypoints()"; return ORBX_ERR_CAPACITY; }
    hipStream_t stream = (hipStream_t)stream_;
    for (int f0 = 0; f0 < nframes; f0 += h->p.max_batch) {
        Batch b;
        memset(&b, 0, sizeof(b));
        fill_batch(h, b);
        b.fallback_hint = h->fallback_hint ? 1 : 0;
        b.nframes = std::min(h->p.max_batch, nframes - f0);
        b.xcd_affinity = (b.nframes >= XCD_AFFINITY_MIN_FRAMES && !h->no_xcd_affinity) ? 1 : 0;
        b.blur_on_demand = h->blur_on_demand;
        b.od_min_frames = h->od_min_frames;
        b.img = d_imgs + (ptrdiff_t)f0 * frame_stride;
        b.img_row_stride = row_stride;
        b.img_frame_stride = frame_stride;
        b.out_kps = d_kps + (size_t)f0 * cap;
        b.out_desc = d_desc + (size_t)f0 * cap * 32;
        b.out_n = d_n + f0;
        b.out_status = d_status ? d_status + f0 : nullptr;
        b.cap = cap;
        rc = launch_extract(b, h->hg, stream, h->stop_after, &h->timer, &h->side, phases);
        if (rc != ORBX_OK) { h->err = "kernel launch failed (no gfx950 code object for this device?)"; return rc; }
        h->last = b;
        h->have_last = true;
    }
    if (phases & ORBX_PHASE_PYRAMID) h->ph_key = key;
    h->ph_done |= phases;
    return ORBX_OK;
}

int orbx_extract(orbx_extractor* h, const uint8_t* img, int w, int hgt, ptrdiff_t stride, orbx_keypoint* kps, uint8_t* desc, int cap,
                 int* n_out) {
    if (!h) return ORBX_ERR_ARG;
    if (!img || w <= 0 || hgt <= 0) return ORBX_EMPTY;   // reference: silent return, outputs untouched
    if (!kps || !desc || !n_out || stride < w) { h->err = "bad argument"; return ORBX_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->p.device));
    int rc = ensure_geometry(h, w, hgt);
    if (rc != ORBX_OK) return rc;
    const int need = h->hg.g.nslots;
    if (cap < need) { h->err = "cap < orbx_max_keypoints()"; return ORBX_ERR_CAPACITY; }
    const int dstride = (w + 63) / 64 * 64;
    const size_t bytes = (size_t)dstride * hgt;
    if (h->img1_bytes < bytes) {
        dev_free(h->d_img1);
        if (h->h_img1) (void)hipHostFree(h->h_img1);
        h->h_img1 = nullptr;
        HIPCHK(h, hipMalloc(&h->d_img1, bytes));
        HIPCHK(h, hipHostMalloc(&h->h_img1, bytes, hipHostMallocMapped));
        HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->m_img1), h->h_img1, 0));
        h->img1_bytes = bytes;
    }
    if (h->out1_cap < need) {
        dev_free(h->d_out1);
        if (h->h_out1) (void)hipHostFree(h->h_out1);
        h->h_out1 = nullptr;
        h->kps1_off = 64;
        h->desc1_off = h->kps1_off + ((size_t)need * sizeof(orbx_keypoint) + 63) / 64 * 64;
        h->out1_bytes = h->desc1_off + (size_t)need * 32;
        HIPCHK(h, hipMalloc(&h->d_out1, h->out1_bytes));
        HIPCHK(h, hipHostMalloc(&h->h_out1, h->out1_bytes, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->m_out1), h->h_out1, 0));
        h->out1_cap = need;
    }
    if (!h->s1) HIPCHK(h, hipStreamCreateWithFlags(&h->s1, hipStreamNonBlocking));
    // In: small frames are staged in pinned memory and fetched from there by a kernel (k_ingest: no copy engine, no queue switch before
    // the first pyramid launch); large ones go through the runtime's pipelined copy.  Out: k_describe, the only writer of the results,
    // stores n / status / keypoints / descriptors straight into the pinned, device-mapped block (posted PCIe writes, visible to the
    // host once the stream has drained) - the D2H copy and the kernel -> copy dependency in front of it (~15 us) are gone.
    if (bytes <= (size_t)512 << 10) {
        for (int y = 0; y < hgt; y++) memcpy(h->h_img1 + (size_t)y * dstride, img + (ptrdiff_t)y * stride, (size_t)w);
        if (h->zero_copy) { if ((rc = launch_ingest(h->d_img1, h->m_img1, bytes, h->s1)) != ORBX_OK) { h->err = "kernel launch failed (no gfx950 code object for this device?)"; return rc; } }
        else HIPCHK(h, hipMemcpyAsync(h->d_img1, h->h_img1, bytes, hipMemcpyHostToDevice, h->s1));
    } else {
        HIPCHK(h, hipMemcpy2DAsync(h->d_img1, dstride, img, stride, w, hgt, hipMemcpyHostToDevice, h->s1));
    }
    uint8_t* out = h->zero_copy ? h->m_out1 : h->d_out1;
    int32_t* d_n = reinterpret_cast<int32_t*>(out);
    // (Replaying the launch group from a HIP graph was measured and does not help: 188 vs 181 us - the latency is the
    //  chain of dependent small kernels, not the launch calls.)
    rc = orbx_extract_batch_device(h, h->d_img1, 1, w, hgt, dstride, (ptrdiff_t)bytes, reinterpret_cast<orbx_keypoint*>(out + h->kps1_off),
                                   out + h->desc1_off, d_n, need, d_n + 1, h->s1);
    if (rc != ORBX_OK) return rc;
    if (!h->zero_copy) HIPCHK(h, hipMemcpyAsync(h->h_out1, h->d_out1, h->out1_bytes, hipMemcpyDeviceToHost, h->s1));
    HIPCHK(h, hipStreamSynchronize(h->s1));
    const int32_t* res = reinterpret_cast<const int32_t*>(h->h_out1);
    if (h->stop_after >= 0) { *n_out = 0; return ORBX_OK; }
    if (res[1] != ORBX_OK) { h->err = "internal list capacity exceeded"; return res[1]; }
    const int n = res[0];
    if (n > 0) {
        memcpy(kps, h->h_out1 + h->kps1_off, (size_t)n * sizeof(orbx_keypoint));
        memcpy(desc, h->h_out1 + h->desc1_off, (size_t)n * 32);
    }
    *n_out = n;
    return ORBX_OK;
}

// ---- diagnostics ---------------------------------------------------------------------------------
int orbx_debug_set_blur_on_demand(orbx_extractor* h, int mode) {
    if (!h || mode < 0 || mode > 1) return ORBX_ERR_ARG;
    h->blur_on_demand = mode;
    return ORBX_OK;
}

int orbx_debug_set_stop_after(orbx_extractor* h, int stage) {
    if (!h) return ORBX_ERR_ARG;
    h->stop_after = stage;
    return ORBX_OK;
}

int orbx_debug_stage_timing(orbx_extractor* h, int enable) {
    if (!h) return ORBX_ERR_ARG;
    if (hipSetDevice(h->p.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ORBX_ERR_DEVICE;
    stage_timer_collect(h->timer);
    h->timer.enabled = enable != 0;
    if (enable == 2) { memset(h->timer.ms, 0, sizeof(h->timer.ms)); memset(h->timer.launches, 0, sizeof(h->timer.launches)); }
    return ORBX_OK;
}

int orbx_debug_stage_time(orbx_extractor* h, int stage, double* total_ms, long* launches) {
    if (!h || stage < 0 || stage >= ST_COUNT) return ORBX_ERR_ARG;
    if (hipSetDevice(h->p.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ORBX_ERR_DEVICE;
    stage_timer_collect(h->timer);
    *total_ms = h->timer.ms[stage];
    *launches = h->timer.launches[stage];
    return ORBX_OK;
}

int orbx_debug_level_size(const orbx_extractor* h, int level, int* w, int* hgt) {
    if (!h || h->gw == 0 || level < 0 || level >= h->hg.g.nlevels) return ORBX_ERR_ARG;
    *w = h->hg.g.lv[level].w;
    *hgt = h->hg.g.lv[level].h;
    return ORBX_OK;
}

long orbx_debug_fetch(orbx_extractor* h, int what, int frame, int level, void* host_out, long cap_bytes) {
    if (!h || !h->have_last || frame < 0 || frame >= h->last.nframes || level < 0 || level >= h->hg.g.nlevels) return ORBX_ERR_ARG;
    if (hipSetDevice(h->p.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ORBX_ERR_DEVICE;
    const DevGeom& g = h->hg.g;
    const LevelGeom& L = g.lv[level];
    const Batch& b = h->last;
    if (what == ORBX_DBG_NMS) {
        // rebuilt on the host from the per-cell survivor lists (valid while they are untouched, i.e. when the
        // last run stopped after stage 1: the retainBest kernels filter and permute the lists in place)
        const long need = (long)L.w * L.h;
        if (cap_bytes < need) return ORBX_ERR_CAPACITY;
        memset(host_out, 0, need);
        std::vector<CellState> st(g.nbands_total);
        if (hipMemcpy(st.data(), b.cstate + (size_t)frame * g.nbands_total, g.nbands_total * sizeof(CellState), hipMemcpyDeviceToHost) != hipSuccess)
            return ORBX_ERR_DEVICE;
        std::vector<Cand> tmp;
        for (int it = 0; it < g.nbands_total; it++) {
            const BandGeom& bgm = h->hg.bands[it];
            if (bgm.level != level) continue;
            const int n = st[it].n_all;
            if (n <= 0) continue;
            if (n > bgm.cand_cap) return ORBX_ERR_CAPACITY;
            tmp.resize(n);
            if (hipMemcpy(tmp.data(), b.cand + (size_t)frame * g.frame_cands + L.cand_base + bgm.cand_off, (size_t)n * sizeof(Cand), hipMemcpyDeviceToHost) != hipSuccess)
                return ORBX_ERR_DEVICE;
            for (int i = 0; i < n; i++) ((uint8_t*)host_out)[(size_t)(tmp[i].pos >> 16) * L.w + (tmp[i].pos & 0xFFFF)] = (uint8_t)tmp[i].resp;
        }
        return need;
    }
    if (what == ORBX_DBG_BANDS) {
        std::vector<CellState> st(g.nbands_total);
        if (hipMemcpy(st.data(), b.cstate + (size_t)frame * g.nbands_total, g.nbands_total * sizeof(CellState), hipMemcpyDeviceToHost) != hipSuccess)
            return ORBX_ERR_DEVICE;
        long n = 0;
        for (int it = 0; it < g.nbands_total; it++) {
            const BandGeom& bgm = h->hg.bands[it];
            if (bgm.level != level) continue;
            if ((n + 1) * 32 > cap_bytes) return ORBX_ERR_CAPACITY;
            int32_t* o = (int32_t*)host_out + 8 * n++;
            o[0] = bgm.x0; o[1] = bgm.x1; o[2] = bgm.y0; o[3] = bgm.y1;
            o[4] = st[it].n_all; o[5] = st[it].n_hi; o[6] = st[it].n_lo;
            o[7] = st[it].thr & 0xFF;          // the threshold the band's list was made at (fastTh; 7 after the second pass or on the fallback hint)
        }
        return n * 32;
    }
    if (what == ORBX_DBG_PLANE || what == ORBX_DBG_BLUR) {
        const long need = (long)L.w * L.h;
        if (cap_bytes < need) return ORBX_ERR_CAPACITY;
        const uint8_t* src;
        size_t spitch;
        if (what == ORBX_DBG_PLANE && level == 0) {
            src = b.img + (ptrdiff_t)frame * b.img_frame_stride;
            spitch = (size_t)b.img_row_stride;
        } else {
            const uint8_t* base = what == ORBX_DBG_PLANE ? b.pyr : b.blur;
            src = base + (size_t)frame * g.frame_plane_bytes + L.plane_off;
            spitch = (size_t)L.stride;
        }
        if (hipMemcpy2D(host_out, L.w, src, spitch, L.w, L.h, hipMemcpyDeviceToHost) != hipSuccess) return ORBX_ERR_DEVICE;
        return need;
    }
    if (what == ORBX_DBG_LEVEL_KPS) {
        int32_t n = 0;
        if (hipMemcpy(&n, b.level_count + frame * MAX_LEVELS + level, 4, hipMemcpyDeviceToHost) != hipSuccess) return ORBX_ERR_DEVICE;
        const long need = (long)n * 12;
        if (cap_bytes < need) return ORBX_ERR_CAPACITY;
        std::vector<Cand> tmp(std::max(n, 1));
        if (n > 0 && hipMemcpy(tmp.data(), b.sel + (size_t)frame * g.frame_sel + L.sel_base, (size_t)n * sizeof(Cand), hipMemcpyDeviceToHost) != hipSuccess)
            return ORBX_ERR_DEVICE;
        int32_t* o = (int32_t*)host_out;
        for (int i = 0; i < n; i++) {
            o[3 * i] = tmp[i].pos & 0xFFFF;
            o[3 * i + 1] = tmp[i].pos >> 16;
            memcpy(&o[3 * i + 2], &tmp[i].resp, 4);
        }
        return need;
    }
    return ORBX_ERR_ARG;
}

int orbx_debug_geometry(const orbx_params* p, int w, int hgt, int32_t* out, int cap_levels) {
    if (!p || !out) return ORBX_ERR_ARG;
    HostGeom hg;
    std::string err;
    const int rc = build_geometry(*p, w, hgt, hg, err);
    if (rc != ORBX_OK) {
        if (getenv("ORBX_DBG_GEOM")) fprintf(stderr, "orbx geometry: %s\n", err.c_str());
        return rc;
    }
    if (cap_levels < hg.g.nlevels) return ORBX_ERR_CAPACITY;
    if (getenv("ORBX_DBG_GEOM"))
        fprintf(stderr, "orbx geometry: fast_lds_bytes %d (band image %d B, chunks %d), sel_lds %d, bands %d, cells %d\n", hg.g.fast_lds_bytes, hg.g.fast_max_img,
                hg.g.fast_max_chunks, hg.g.sel_lds_cell, (int)hg.bands.size(), (int)hg.cells.size());
    for (int l = 0; l < hg.g.nlevels; l++) {
        const LevelGeom& L = hg.g.lv[l];
        int nb = 0;
        for (const BandGeom& b : hg.bands) nb += b.level == l;
        int32_t* o = out + 8 * l;
        o[0] = L.w; o[1] = L.h; o[2] = L.ndesired; o[3] = L.gcols; o[4] = L.grows; o[5] = L.cellW; o[6] = L.cellH; o[7] = nb;
    }
    return hg.g.nlevels;
}

int orbx_debug_nth_element(const float* resp, int n, int nth, int32_t* out_perm, int device) {
    if (!resp || !out_perm || n < 1 || nth < 0 || nth > n || n > 13000) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    float* d_r = nullptr;
    int* d_o = nullptr;
    int rc = ORBX_ERR_DEVICE;
    if (hipMalloc(&d_r, (size_t)n * 4) == hipSuccess && hipMalloc(&d_o, (size_t)n * 4) == hipSuccess &&
        hipMemcpy(d_r, resp, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess) {
        rc = launch_debug_nth(d_r, n, nth, d_o);
        if (rc == ORBX_OK && hipMemcpy(out_perm, d_o, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = ORBX_ERR_DEVICE;
    }
    if (d_r) (void)hipFree(d_r);
    if (d_o) (void)hipFree(d_o);
    return rc;
}

int orbx_debug_eval_math(int kind, const float* in0, const float* in1, float* out0, float* out1, int n, int device) {
    if (n <= 0) return ORBX_OK;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    float* d = nullptr;
    const size_t nb = (size_t)n * sizeof(float);
    if (hipMalloc(&d, 4 * nb) != hipSuccess) return ORBX_ERR_DEVICE;
    int rc = ORBX_ERR_DEVICE;
    if (hipMemcpy(d, in0, nb, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d + n, in1 ? in1 : in0, nb, hipMemcpyHostToDevice) == hipSuccess) {
        rc = launch_eval_math(kind, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n);
        if (rc == ORBX_OK && hipMemcpy(out0, d + 2 * (size_t)n, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = ORBX_ERR_DEVICE;
        if (rc == ORBX_OK && out1 && hipMemcpy(out1, d + 3 * (size_t)n, nb, hipMemcpyDeviceToHost) != hipSuccess) rc = ORBX_ERR_DEVICE;
    }
    (void)hipFree(d);
    return rc;
}


int orbx_device_alloc(int device, size_t bytes, void** d_ptr) {
    if (!d_ptr || bytes == 0) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return ORBX_ERR_DEVICE;
    if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return ORBX_ERR_DEVICE; }
    *d_ptr = p;
    return ORBX_OK;
}

int orbx_device_free(int device, void* d_ptr) {
    if (!d_ptr) return ORBX_OK;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    return hipFree(d_ptr) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_device_upload(int device, void* d_dst, const void* src, size_t bytes) {
    if (bytes == 0) return ORBX_OK;
    if (!d_dst || !src) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    return hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_device_download(int device, void* dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return ORBX_OK;
    if (!dst || !d_src) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ORBX_ERR_DEVICE;
    return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_stream_create(int device, void** stream) {
    if (!stream) return ORBX_ERR_ARG;
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return ORBX_ERR_DEVICE;
    *stream = s;
    return ORBX_OK;
}

int orbx_stream_create_priority(int device, int priority, void** stream) {
    if (!stream) return ORBX_ERR_ARG;
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) return ORBX_ERR_DEVICE;
    *stream = s;
    return ORBX_OK;
}

int orbx_stream_destroy(int device, void* stream) {
    if (!stream) return ORBX_OK;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_stream_synchronize(int device, void* stream) {
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    const hipError_t e = stream ? hipStreamSynchronize((hipStream_t)stream) : hipDeviceSynchronize();
    return e == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_event_create(int device, void** event) {
    if (!event) return ORBX_ERR_ARG;
    hipEvent_t e = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return ORBX_ERR_DEVICE;
    *event = e;
    return ORBX_OK;
}

int orbx_event_destroy(int device, void* event) {
    if (!event) return ORBX_OK;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    return hipEventDestroy((hipEvent_t)event) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_event_record(void* event, void* stream) {
    if (!event) return ORBX_ERR_ARG;
    return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_stream_wait_event(void* stream, void* event) {
    if (!event) return ORBX_ERR_ARG;
    return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_device_copy_async(void* d_dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes == 0) return ORBX_OK;
    if (!d_dst || !d_src) return ORBX_ERR_ARG;
    return hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_host_alloc(int device, size_t bytes, void** h_ptr) {
    if (!h_ptr || bytes == 0) return ORBX_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return ORBX_ERR_DEVICE;
    *h_ptr = p;
    return ORBX_OK;
}

int orbx_host_free(int device, void* h_ptr) {
    if (!h_ptr) return ORBX_OK;
    if (hipSetDevice(device) != hipSuccess) return ORBX_ERR_DEVICE;
    return hipHostFree(h_ptr) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_device_upload_async(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    if (bytes == 0) return ORBX_OK;
    if (!d_dst || !h_src) return ORBX_ERR_ARG;
    return hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

int orbx_device_download_async(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes == 0) return ORBX_OK;
    if (!h_dst || !d_src) return ORBX_ERR_ARG;
    return hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? ORBX_OK : ORBX_ERR_DEVICE;
}

}  // extern "C"
