/* orbf.h — C ABI of the Frame-side steps around the extractor / matcher boundary (part of liborbx.so).
 *
 * SURVEY.md §8f row N3 (what Frame::Frame does with the keypoints as soon as the extractor returns) and §8a row M3
 * (the grid window query every ORBmatcher search starts from), so that keypoints, descriptors and the search grid
 * stay in HBM between orbx_extract_batch_device and the matcher kernels.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   orbf_image_bounds        <- Frame::ComputeImageBounds()  src/Frame.cc:321-351  and the two inverse cell sizes :75-76
 *                               (host, once per camera: four points)
 *   orbf_undistort_grid[_batch_device]
 *                            <- Frame::UndistortKeyPoints()  src/Frame.cc:289-319  (cv::undistortPoints(mat, mat, mK, mDistCoef,
 *                               cv::Mat(), mK): OpenCV 2.4 cvUndistortPoints, doubles, 5 iterations) and the mGrid fill
 *                               src/Frame.cc:108-123 with Frame::PosInGrid :267-277
 *   orbf_features_in_area[_batch_device]
 *                            <- Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel)  src/Frame.cc:200-265
 *
 * Grid form.  `std::vector<size_t> mGrid[64][48]` (include/Frame.h:35-36,:90) becomes CSR per frame: cell c = x*48 + y,
 * cell_off[c]..cell_off[c+1] delimits the feature indices of the cell in cell_feat, ascending (push_back order).
 * Keypoints are orbx_keypoint (= cv::KeyPoint, 28 bytes).  Status codes are orbx.h's; no CPU fallback for the
 * device entry points.
 */
#ifndef ORBF_H
#define ORBF_H

#include <stddef.h>
#include <stdint.h>

#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBF_GRID_COLS 64                      /* FRAME_GRID_COLS include/Frame.h:36 */
#define ORBF_GRID_ROWS 48                      /* FRAME_GRID_ROWS include/Frame.h:35 */
#define ORBF_GRID_CELLS (ORBF_GRID_COLS * ORBF_GRID_ROWS)
#define ORBF_MAX_FEATURES 8192                 /* features of one frame */

/* mK (3x3 CV_32F, row major) and mDistCoef (CV_32F: k1 k2 p1 p2 [k3 [k4 k5 k6]]; the reference passes 4,
 * src/Tracking.cc:58-70); width/height = im.cols / im.rows */
typedef struct orbf_camera {
    float K[9];
    float dist[8];
    int32_t ndist;
    int32_t width, height;
} orbf_camera;

/* mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv (static members of Frame) */
typedef struct orbf_bounds {
    int32_t min_x, max_x, min_y, max_y;
    float inv_w, inv_h;
} orbf_bounds;

int orbf_image_bounds(const orbf_camera* cam, orbf_bounds* out);

/* One frame, host pointers: kps_un[n], cell_off[ORBF_GRID_CELLS + 1], cell_feat[n]. */
int orbf_undistort_grid(const orbf_camera* cam, const orbf_bounds* b, const orbx_keypoint* kps, int n,
                        orbx_keypoint* kps_un, int32_t* cell_off, int32_t* cell_feat, int device);
/* Throughput form in the extractor's batch layout: frame f has d_n[f] keypoints at d_kps + f*cap.  Outputs:
 * d_kps_un + f*cap, d_cell_off + f*(ORBF_GRID_CELLS+1), d_cell_feat + f*cap.  cap <= ORBF_MAX_FEATURES.
 * Asynchronous on `stream` (hipStream_t). */
int orbf_undistort_grid_batch_device(const orbf_camera* cam, const orbf_bounds* b, const orbx_keypoint* d_kps,
                                     const int32_t* d_n, int nframes, int cap, orbx_keypoint* d_kps_un,
                                     int32_t* d_cell_off, int32_t* d_cell_feat, void* stream);

/* Window queries against ONE frame's grid: query q = (x, y, r, minLevel, maxLevel) as five floats/ints in
 * qxyr[3*q..] and qlev[2*q..].  Results as CSR: seg_off[nq+1], cand[] in the reference's push_back order
 * (cells x-major, then y, then cell order).  Returns ORBX_ERR_CAPACITY when more than cand_cap indices result
 * (seg_off is still complete, so the caller can size the buffer and call again). */
int orbf_features_in_area(const orbf_bounds* b, const orbx_keypoint* kps_un, int n, const int32_t* cell_off,
                          const int32_t* cell_feat, const float* qxyr, const int32_t* qlev, int nq,
                          int32_t* seg_off, int32_t* cand, int cand_cap, int device);
int orbf_features_in_area_device(const orbf_bounds* b, const orbx_keypoint* d_kps_un, int n, const int32_t* d_cell_off,
                                 const int32_t* d_cell_feat, const float* d_qxyr, const int32_t* d_qlev, int nq,
                                 int32_t* d_seg_off, int32_t* d_cand, int cand_cap, int32_t* d_status, void* stream);

#ifdef __cplusplus
}
#endif
#endif
