// Internal data layout shared by the host side (geometry, launches) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "orbx.h"

namespace orbx {

constexpr int MAX_LEVELS = 16;
constexpr int EDGE = 16;        // EDGE_THRESHOLD, reference src/ORBextractor.cc:77
constexpr int HALF_PATCH = 15;  // HALF_PATCH_SIZE, :76

// One detected corner that survived NMS.  resp is the float response the reference sorts by
// (FAST score, or Harris when score_type==HARRIS_SCORE).
struct Cand {
    uint32_t pos;   // x | y<<16, level coordinates
    float resp;
};

// resize tables (cv::resize INTER_LINEAR 8U; host-computed with the exact float/double sequence)
struct ResizeX { int16_t sx, sx1, a0, a1; };   // sx1 = min(sx+1, sw-1) (weight a1 is 0 when clamped)
struct ResizeY { int16_t sy0, sy1, b0, b1; };  // rows already clamped to [0, sh-1]

struct LevelGeom {
    int w, h;              // level size
    int stride;            // row stride (bytes) of this level in the pyramid / blur / nms blocks
    int plane_off;         // byte offset of the level plane inside one frame's block (same for pyr, blur, nms)
    int gcols, grows;      // cell grid (reference ComputeKeyPoints :534-535)
    int cellW, cellH;
    int ncells;
    int nfeat_cell;        // nfeaturesCell (:547)
    int ndesired;          // mnFeaturesPerLevel[level]
    int cell_base;         // first cell index of this level in per-frame cell arrays
    int cand_base;         // first Cand slot of this level in one frame's candidate block
    int sel_base, sel_cap; // selected-keypoint list of this level in one frame's sel block
    int slot_base;         // prefix of ndesired over levels (output slot of the level's first keypoint when every level is full)
    int quad_base;         // prefix of ceil(ndesired / 4) over levels: k_describe forms its waves (4 keypoints each) per level
    int tabx_off, taby_off;// offsets into the ResizeX / ResizeY tables (level >= 1)
    int rz_window;         // k_resize: 1 = the four output pixels of a lane read at most 8 adjacent source bytes per row (windowed form)
    int rz_pitch, rz_rows; // k_resize: LDS source tile of one 256x16 output tile (bytes per row, rows), maxima over the level's tiles
    int btile_base, btiles_x, btiles_y;// blur wave tasks: 248-px strips x 32-row bands of the whole plane
    int btile_base_s, btiles_y_s;      // the same with BLUR_ROWS_SMALL-row bands (launch groups too small to fill the chip: shorter waves)
    int mb_base, mb_n;     // k_blur_mfma work items of the level: mb_strips 64-px strips x mb_bands row bands of mb_band_steps 32-row steps, band-major
    int mb_strips, mb_bands, mb_band_steps;
    int blur_wvec;         // columns x < blur_wvec round ties-to-even (SSE2 emulation), others half-up
    int blur_sel_last, blur_sel_halo;   // v_perm selectors building the reflect-101 bytes of the last / right-halo dword of a row
    float scale;           // mvScaleFactor[level]
    float kp_size;         // (float)(int)(PATCH_SIZE*mvScaleFactor[level])  (:675,:692)
};

struct CellGeom {
    int16_t x0, y0, x1, y1;   // inclusive scan rectangle in level coords; empty when x1<x0 or y1<y0
    int32_t skipped;          // reference `continue` cells (hX<=0 / hY<=0): never reach the quota else-branch
    int32_t cand_off;         // first Cand slot relative to the level's cand_base (= its first band's sub-list)
    int32_t cand_cap;         // sum of its bands' capacities
    int32_t band0, nbands;    // its k_fast_cells work items
};

// One k_fast_cells work item: a band of rows of a grid cell.  Cells above BAND_PX pixels (1080p grids) are cut into
// row bands so that a work item's LDS footprint stays ~35 KB (4 workgroups per CU); bands concatenate in raster order.
#ifndef ORBX_BLUR_WAVES
#define ORBX_BLUR_WAVES 2
#endif
#ifndef ORBX_BLUR_ON_DEMAND_DEFAULT
#define ORBX_BLUR_ON_DEMAND_DEFAULT 1     // (round 6) what a handle does when ORBX_BLUR_ON_DEMAND is not in the environment: full launch groups blur per keypoint
                                          // window (k_describe_od: VGA 381.5k -> 404.1k frames/s, 1080p 75.2k -> 90.4k on the first build; profiles/r06_ab_on_demand.txt)
#endif
#ifndef ORBX_OD_MIN_FRAMES_DEFAULT
#define ORBX_OD_MIN_FRAMES_DEFAULT 32    // launch groups below this keep the blur kernels (the one-frame call: FAST + blur in one launch)
#endif
#ifndef ORBX_DESC_PACKED_PATTERN
#define ORBX_DESC_PACKED_PATTERN 1      // k_describe: the BRIEF pattern in LDS as packed int8 (1 KB) instead of floats (4 KB)
#endif
#ifndef ORBX_DESC_WAVES
#define ORBX_DESC_WAVES 4
#endif
constexpr int BLUR_WAVES = ORBX_BLUR_WAVES;   // k_blur: strips (waves) per workgroup
constexpr int DESC_WAVES = ORBX_DESC_WAVES;   // k_describe: keypoints (waves) per workgroup
#ifndef ORBX_BLUR_ROWS
#define ORBX_BLUR_ROWS 32
#endif
#ifndef ORBX_RZ_ROWS
#define ORBX_RZ_ROWS 48
#endif
#ifndef ORBX_BLUR_MFMA
#define ORBX_BLUR_MFMA 1                    // 1: full launch groups of aligned VGA-class frames take k_blur_mfma (round 4); 0: k_blur everywhere; 2: k_blur_mfma at every width
#endif
#ifndef ORBX_MB_BAND_STEPS
#define ORBX_MB_BAND_STEPS 8
#endif
constexpr int MB_BAND_STEPS = ORBX_MB_BAND_STEPS;   // k_blur_mfma: a wave streams at most this many 32-row steps (taller levels are cut into row bands: the
                                                    // launch ends with its longest waves, and a VGA level 0 is 15 steps against 5 on level 7)
constexpr int MB_MAX_WIDTH = 1024;           // ... and only levels-0 up to this width (see launch_extract)
constexpr int MB_TILES = 2;                 // k_blur_mfma: 32-column tiles per strip (one wave streams a 64-pixel strip down all rows)
#ifndef ORBX_MB_WAVES
#define ORBX_MB_WAVES 4
#endif
constexpr int MB_WAVES = ORBX_MB_WAVES;     // ... strips (waves) per workgroup
constexpr int BLUR_ROWS = ORBX_BLUR_ROWS;   // k_blur: output rows per wave strip
constexpr int BLUR_ROWS_SMALL = 8;          // ... when the launch group cannot fill the chip anyway (a wave's strip is a serial chain of rows)
constexpr int RZ_ROWS = ORBX_RZ_ROWS;       // k_resize: output rows per workgroup (a tall tile amortises the table -> source -> LDS latency chain)
// The two launch shapes of k_fast_cells: threads per work item, dwords (4 pixels) per lane and round, band size in pixels, widest
// cell.  Per WAVE the kernel keeps a queue of flagged dwords (one step of 64 on top of a remainder < 64), of expanded pixel offsets
// (two of a dword's four pixels per 64 dwords = up to 128 on top of a remainder < 64), of pair-test survivors (one step of 64 on top
// of a remainder < 64) and of scored corners (the NMS work list; a band that overflows it takes a dense sweep instead).
struct FastShape { int threads, ppt, band_px, max_cw; };
#ifndef ORBX_FS
#define ORBX_FS 256, 1, 8192, 500
#endif
#ifndef ORBX_FL
#define ORBX_FL 256, 2, 7168, 6500
#endif
constexpr FastShape FAST_SMALL = {ORBX_FS};   // VGA-class grids (round 4: ONE dword per lane and round — with the bands scored at fastTh first the sparse phases shrank and the
                                              // shorter dense rounds win: 0.883 -> 0.833 ms per 1024 VGA frames, S-midtex 1.02 -> 0.95; rounds 2-3 ran two, then a wash)
constexpr FastShape FAST_LARGE = {ORBX_FL};   // 720p / 1080p-class grids (round 3: 256 threads over 7168-px bands — per 256 1080p frames 1.84 ms
                                              // against 2.06 for the 512 threads over 10240-px bands of rounds 1-2; 256 threads at 5120 / 6144 / 8192 /
                                              // 12288 px: 1.91 / 1.91 / 1.90 / 1.95).  max_cw 6500: two own rows + 2 halo rows + the 6 ring rows of a staged band
                                              // (pitch <= 6512) stay below 64 KiB, the range of the 16-bit pixel offsets; the kernel's float division
                                              // p -> (p / S, p % S) is exact for every S <= 8192, p < 65536 (checked exhaustively)
// (round 3: the kernel is latency-bound at the occupancy its LDS allows — 1.06 / 1.15 / 1.30 / 1.56 ms per 1024 VGA frames at
//  6 / 5 / 4 / 3 workgroups per CU — so the queues are kept as small as their invariants allow: every producer drains as soon as
//  a slice of 64 is full)
// 1: k_fast_cells scores a band at fastTh first and again at 7 only when it keeps <= 3 survivors (round 4); 0: one pass at min(fastTh, 7)
#ifndef ORBX_FAST_TWO_PASS
#define ORBX_FAST_TWO_PASS 1
#endif
#ifndef ORBX_FAST_ROTATE
#define ORBX_FAST_ROTATE 1      // k_fast_cells: wave roles rotate with the band index (see fast_band_task)
#endif
constexpr int FAST_Q1CAP = 192, FAST_Q2CAP = 128, FAST_Q3CAP = 192;
constexpr int fast_q0cap(int) { return 128; }
constexpr int fast_wave_queue_bytes(int ppt) { return fast_q0cap(ppt) * 4 + FAST_Q1CAP * 2 + FAST_Q2CAP * 2 + FAST_Q3CAP * 2; }
// dwords per row of a staged band (k_fast_cells): the cell's columns + 3 ring columns either side from the dword-aligned start
// `xoff` bytes to the left, rounded up to whole 16-byte chunks (the LDS-DMA staging moves 16 bytes per lane)
__host__ __device__ constexpr int fast_row_dwords(int xoff, int cw) { return (((xoff + cw + 6 + 3) >> 2) + 3) & ~3; }
struct BandGeom {
    int16_t x0, x1;           // the cell's column range (inclusive)
    int16_t y0, y1;           // rows this band owns (inclusive)
    int16_t ey0, ey1;         // rows it scores: own rows + 1 halo row towards neighbouring bands of the same cell
    int32_t level;            // (a whole dword: as int16 the compiler fetched it with a VECTOR load — a memory round trip in front of every k_fast_cells workgroup)
    int32_t cand_off;         // first Cand slot of its sub-list relative to the level's cand_base
    int32_t cand_cap;
    // 1 / (dwords, bytes, 16-byte chunks per row of the staged band): the divisors of k_fast_cells' p -> (row, column) splits.  Computed
    // on the host (IEEE single division, the bits the device's own 1.0f / x gives): three division sequences less in front of every wave
    float inv_nd, inv_s, inv_cpr;
};

// Fused pyramid (k_pyramid): one launch produces `depth` consecutive levels l0+1 .. l0+depth from level l0.  A workgroup owns a
// tw x th tile of the DEEPEST level and everything above it: the region of every intermediate level it needs lives in LDS.
// Regions and ownership come from host tables (pyr_tab): per level k = 0 .. depth (k = 0: the source level) and tile index,
// the first / last column (row) of the tile's region; a tile OWNS (= writes to HBM) the columns from its region start up to the
// next tile's region start, which partitions every level.  Region starts are multiples of 4 (dword stores never straddle owners).
constexpr int XCD_AFFINITY_MIN_FRAMES = 64;      // launch groups of at least this many frames keep a frame's workgroups on one XCD (orbx_device.h: frame_item)
#ifndef ORBX_PYR_DEPTH
#define ORBX_PYR_DEPTH 4, 4
#endif
constexpr int PYR_MAX_GROUPS = 4, PYR_MAX_DEPTH = 8;
struct PyrGroup {
    int l0, depth;               // source level, number of levels produced
    int ntx, nty;                // tiles of the deepest level
    int xtab, ytab;              // offsets into pyr_tab: x regions [(depth + 1)][ntx + 1][2], then y regions [(depth + 1)][nty + 1][2]
    int lds_off[PYR_MAX_DEPTH];  // LDS byte offset of the buffer of level l0 + k (k = 0 .. depth-1; the deepest level goes to HBM only)
    int pitch[PYR_MAX_DEPTH];    // ... its row pitch (multiple of 4)
    int lds_bytes;
};
#ifndef ORBX_PYR_TILE
#define ORBX_PYR_TILE 16, 16
#endif

constexpr int QUOTA_LDS_PER_CELL = 10;                              // k_quota: two ints + two bytes of LDS per cell of a level
constexpr int QUOTA_MAX_CELLS = 160 * 1024 / QUOTA_LDS_PER_CELL / 64 * 64;   // 16384: what one workgroup's LDS holds
// Per FAST work item (row band of a grid cell).  `thr` is the threshold the band's list and counters were made at: fastTh when the
// band kept more than 3 survivors@fastTh (or fastTh <= 7), else 7 — the second pass, or a pass that STARTED at 7 on the band's
// fallback hint (CellState::thr >> 8), in which case n_hi may exceed 3.  INVARIANT the later stages rely on: in a list made at fastTh,
// n_lo counts only survivors@fastTh (not survivors@7), so n_lo may be consumed only for a cell whose bands ALL have n_hi <= 3 (then every
// band of it was listed at 7): k_quota reads n_lo exactly when sum(n_hi) <= 3.  A list made at 7 is filtered by score >= the cell's
// threshold in k_cell_select (survivors@fastTh = survivors@7 with score >= fastTh).
struct CellState { int32_t n_all, n_hi, n_lo, thr; };     // survivors listed, of them with score >= fastTh, with score >= 7; list threshold (low byte) | run of fallbacks of this frame slot << 8
struct CellSel { int32_t thr, nkeys, nretain, out_off; };

struct DevGeom {
    int nlevels;
    int ncells_total;
    int nbands_total;        // k_fast_cells work items per frame (>= ncells_total)
    uint32_t nbands_magic;   // floor(2^32 / nbands_total) + 1 (0 for one band): block -> (frame, band) with a scalar multiply instead of a division
    int nbtiles_total, nbtiles_total_s;
    int nmb_total;           // k_blur_mfma: work items (strip x band) per frame (all levels)
    int nslots;              // sum of ndesired (max keypoints per frame)
    int nquads;              // sum of ceil(ndesired / 4): k_describe waves per frame
    int quota_cells;         // cells of the level with the most cells, rounded up to 64 (k_quota LDS arrays)
    int score_type, fast_th, tmin;
    int fp_contract;         // orbx_params::fp_contract (Harris response, descriptor rotation)
    int frame_plane_bytes;   // bytes of one frame's pyramid block (== blur block == nms block)
    int frame_cands;         // Cand slots per frame
    int frame_sel;           // sel slots per frame
    int sel_lds_cell, sel_lds_level, sel_lds_entries;   // LDS bytes of k_cell_select / k_level_select and the list entries they stage
    int fast_max_img, fast_max_chunks, fast_lds_bytes;   // k_fast_cells LDS carve: staged image bytes (= score plane bytes) of the largest band, 64-byte chunks of a band's own rows
    int fast_threads;        // k_fast_cells workgroup size chosen for this geometry
    int fast_small;          // 1: the small launch shape (VGA-class grids), 0: the large one
    int npyr_groups;         // 0: one k_resize launch per level (scale factors > 2 or regions that do not fit the LDS)
    PyrGroup pyr[PYR_MAX_GROUPS];
    int umax[HALF_PATCH + 1];
    // per-level bases as compact arrays: one scalar load each, so a wave finds its level in a single round trip
    // (walking lv[l].xxx_base level by level was a chain of up to nlevels dependent scalar loads, ~1.5 us per wave)
    int cell_bases[MAX_LEVELS], quad_bases[MAX_LEVELS], btile_bases[MAX_LEVELS], btile_bases_s[MAX_LEVELS], mb_bases[MAX_LEVELS];
    LevelGeom lv[MAX_LEVELS];
};

// Arguments of one batch launch group; passed to kernels BY VALUE so that the geometry lives in the kernarg
// segment: every LevelGeom field is then fetched with scalar loads (wave-uniform, no vector-memory round trip).
// (A pointer to the same struct in global memory makes the compiler issue per-lane global loads for each field.)
struct Batch {
    DevGeom g;
    const CellGeom* cells;
    const BandGeom* bands;
    const ResizeX* tabx;
    const ResizeY* taby;
    const int* pyr_tab;       // region tables of the fused pyramid launches (PyrGroup)
    const uint8_t* img;       // level 0 (caller's frames)
    long long img_row_stride, img_frame_stride;
    uint8_t* pyr;             // [frame][frame_plane_bytes]  levels >= 1 (level-0 slot unused)
    uint8_t* blur;            // [frame][frame_plane_bytes]
    Cand* cand;               // [frame][frame_cands]
    Cand* sel;                // [frame][frame_sel]
    CellState* cstate;        // [frame][nbands_total]  (per band: survivors, and how many reach fastTh / 7)
    int fallback_hint;        // 1: a band starts at threshold 7 once its slot fell back FAST_HINT_RUN launch groups in a row (the run rides in CellState::thr);
                              // 0 (ORBX_FALLBACK_HINT=0): every band is scored at fastTh first
    CellSel* csel;            // [frame][ncells_total]
    int32_t* level_total;     // [frame][MAX_LEVELS]  keypoints gathered from the cells
    int32_t* level_count;     // [frame][MAX_LEVELS]  after the per-level cap
    int32_t* status;          // [frame]
    int32_t* long_cells;      // [frame][ncells_total]  1: the cell's list exceeds k_cell_select's short staging area (k_cell_select_long takes it)
    orbx_keypoint* out_kps;   // [frame][cap]
    uint8_t* out_desc;        // [frame][cap][32]
    int32_t* out_n;           // [frame]
    int32_t* out_status;      // optional [frame]
    int cap;
    int nframes;
    int xcd_affinity;         // 1: launches renumber their blocks so that a frame's work items share one XCD (its L2)
    int od_min_frames;        // ... from this many frames per launch group on
    int blur_on_demand;       // 1: no blurred plane — k_describe_od blurs each keypoint's window itself (full launch groups of supported geometries)
};

// Host-side geometry builder result.
struct HostGeom {
    DevGeom g;
    std::vector<CellGeom> cells;
    std::vector<BandGeom> bands;
    std::vector<ResizeX> tabx;
    std::vector<ResizeY> taby;
    std::vector<int> pyr_tab;
    std::vector<int> features_per_level;
    std::vector<float> scale, inv_scale;
};

// Fills `out` for a w x h input; returns ORBX_OK / ORBX_ERR_GEOMETRY / ORBX_ERR_ARG.
int build_geometry(const orbx_params& p, int w, int h, HostGeom& out, std::string& err);

enum Stage { ST_PYRAMID = 0, ST_FAST_CELLS, ST_QUOTA, ST_CELL_SELECT, ST_LEVEL_SELECT, ST_BLUR, ST_DESCRIBE, ST_COUNT };

// Optional per-stage timing with HIP events recorded on the launch stream (diagnostics / bench roofline).
struct StageTimer {
    bool enabled = false;
    std::vector<hipEvent_t> pool;      // events of the launch groups not yet folded into `ms`
    std::vector<int> pool_stage;       // stage id per (start, stop) pair
    double ms[ST_COUNT] = {0};
    long launches[ST_COUNT] = {0};
};

// Launch the whole per-batch kernel sequence on `stream`; stop_after < 0 runs everything.
// Side stream: the blur depends only on the pyramid.  It is forked after the (VALU-bound) FAST kernel so that it
// runs concurrently with the quota / retainBest kernels, which are latency-bound and leave the chip mostly idle.
struct SideStream {
    hipStream_t aux = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
// `phases`: ORBX_PHASE_* bits of include/orbx.h (which parts of the sequence to queue; ORBX_PHASE_ALL = everything)
int launch_extract(const Batch& b, const HostGeom& hg, hipStream_t stream, int stop_after, StageTimer* timer, const SideStream* side,
                   int phases = ORBX_PHASE_ALL);
// k_describe_od.hip: the description stage with the Gaussian blur computed per keypoint window (no blurred plane)
bool describe_od_supported(const Batch& b, const HostGeom& hg);
int launch_describe_od(const Batch& b, const HostGeom& hg, hipStream_t stream);

}  // namespace orbx
