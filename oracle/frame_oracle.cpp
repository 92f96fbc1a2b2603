// =====================================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp's header; the same rules apply).
//
// CPU restatement of what Frame::Frame does with the keypoints right after the extractor returns
//  and of the grid window query every ORBmatcher search starts from (§8a M3):
//   /root/reference/src/Frame.cc:289-319   Frame::UndistortKeyPoints   (cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK))
//   /root/reference/src/Frame.cc:321-351   Frame::ComputeImageBounds
//   /root/reference/src/Frame.cc:75-76     mfGridElementWidthInv / mfGridElementHeightInv
//   /root/reference/src/Frame.cc:108-123   the mGrid fill;  :267-277 Frame::PosInGrid
//   /root/reference/src/Frame.cc:200-265   Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel)
//
// PARITY PINNED for the ORB_SLAM parts above: the reference's own src/Frame.cc + include/Frame.h compile where they lie against
// plain-data stand-ins of their heavy includes (oracle/matcherstub with -DORB_ORACLE_REAL_FRAME, oracle/ref_frame_wrap.cpp ->
// _ref/libref_frame.so; the "extractor" hands the constructor preset key points) and tests/test_ref_pin_frame.py compares bounds,
// inverse cell sizes, the grid, window queries (every level-argument form) and the scale tables on 4 cameras.
// cv::undistortPoints is an OpenCV primitive (absent): restated from OpenCV 2.4 modules/imgproc/src/undistort.cpp
// (cvUndistortPoints: all arithmetic in double, 5 fixed-point iterations when distortion coefficients are given, RR = P·I = K,
// float stores).  PARITY UNPINNED for that primitive (behind the stand-in header Frame.cc calls this restatement).
// =====================================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
const int GRID_ROWS = 48, GRID_COLS = 64;     // include/Frame.h:35-36

struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

struct Camera {
    float K[9];          // row-major 3x3, CV_32F (src/Tracking.cc:58-63)
    float dist[8];       // k1 k2 p1 p2 [k3 [k4 k5 k6]] CV_32F (src/Tracking.cc:65-70 passes 4)
    int32_t ndist;
    int32_t width, height;
};
struct Bounds { int32_t min_x, max_x, min_y, max_y; float inv_w, inv_h; };

// cvUndistortPoints for one CV_32FC2 point with R = I, P = cameraMatrix
void undistort_point(const Camera& c, float xin, float yin, float* xo, float* yo) {
    double A[3][3], k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 9; i++) A[i / 3][i % 3] = c.K[i];
    for (int i = 0; i < c.ndist && i < 8; i++) k[i] = c.dist[i];
    const int iters = c.ndist > 0 ? 5 : 1;
    const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
    double x = xin, y = yin, x0, y0;
    x0 = x = (x - cx) * ifx;
    y0 = y = (y - cy) * ify;
    for (int j = 0; j < iters; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    double xx = A[0][0] * x + A[0][1] * y + A[0][2];
    double yy = A[1][0] * x + A[1][1] * y + A[1][2];
    double ww = 1. / (A[2][0] * x + A[2][1] * y + A[2][2]);
    x = xx * ww;
    y = yy * ww;
    *xo = (float)x;
    *yo = (float)y;
}

// Frame.cc:267-277
bool pos_in_grid(const Bounds& b, const KeyPoint& kp, int& posX, int& posY) {
    posX = (int)roundf((kp.x - b.min_x) * b.inv_w);
    posY = (int)roundf((kp.y - b.min_y) * b.inv_h);
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return false;
    return true;
}
}  // namespace

extern "C" {
// Frame.cc:321-351 and :75-76
void orc_frame_bounds(const Camera* c, Bounds* b) {
    if (c->dist[0] != 0.0) {
        float m[4][2] = {{0.f, 0.f}, {(float)c->width, 0.f}, {0.f, (float)c->height}, {(float)c->width, (float)c->height}};
        for (int i = 0; i < 4; i++) undistort_point(*c, m[i][0], m[i][1], &m[i][0], &m[i][1]);
        b->min_x = (int32_t)fmin(floor(m[0][0]), floor(m[2][0]));
        b->max_x = (int32_t)fmax(ceil(m[1][0]), ceil(m[3][0]));
        b->min_y = (int32_t)fmin(floor(m[0][1]), floor(m[1][1]));
        b->max_y = (int32_t)fmax(ceil(m[2][1]), ceil(m[3][1]));
    } else {
        b->min_x = 0; b->max_x = c->width; b->min_y = 0; b->max_y = c->height;
    }
    b->inv_w = (float)GRID_COLS / (float)(b->max_x - b->min_x);
    b->inv_h = (float)GRID_ROWS / (float)(b->max_y - b->min_y);
}

// Frame.cc:289-319: kps_un = kps with pt replaced (plain copy when dist[0] == 0)
void orc_frame_undistort(const Camera* c, const KeyPoint* kps, int n, KeyPoint* out) {
    for (int i = 0; i < n; i++) {
        out[i] = kps[i];
        if (c->dist[0] != 0.0) undistort_point(*c, kps[i].x, kps[i].y, &out[i].x, &out[i].y);
    }
}

// Frame.cc:108-123: mGrid[x][y] as CSR over cell = x*48 + y; features in push_back (index) order
void orc_frame_grid(const Bounds* b, const KeyPoint* kps_un, int n, int32_t* cell_off /*3073*/, int32_t* cell_feat /*n*/) {
    std::vector<std::vector<int> > grid(GRID_COLS * GRID_ROWS);
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(*b, kps_un[i], px, py)) grid[px * GRID_ROWS + py].push_back(i);
    }
    int o = 0;
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) {
        cell_off[c] = o;
        for (size_t j = 0; j < grid[c].size(); j++) cell_feat[o++] = grid[c][j];
    }
    cell_off[GRID_COLS * GRID_ROWS] = o;
}

// Frame.cc:200-265; returns the number of indices written (in the reference's push_back order)
int orc_frame_features_in_area(const Bounds* b, const KeyPoint* kps_un, const int32_t* cell_off, const int32_t* cell_feat,
                               float x, float y, float r, int minLevel, int maxLevel, int32_t* out) {
    int n = 0;
    int nMinCellX = (int)floor((x - b->min_x - r) * b->inv_w);
    nMinCellX = nMinCellX > 0 ? nMinCellX : 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceil((x - b->min_x + r) * b->inv_w);
    nMaxCellX = nMaxCellX < GRID_COLS - 1 ? nMaxCellX : GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floor((y - b->min_y - r) * b->inv_h);
    nMinCellY = nMinCellY > 0 ? nMinCellY : 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceil((y - b->min_y + r) * b->inv_h);
    nMaxCellY = nMaxCellY < GRID_ROWS - 1 ? nMaxCellY : GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    bool bCheckLevels = true, bSameLevel = false;
    if (minLevel == -1 && maxLevel == -1) bCheckLevels = false;
    else if (minLevel == maxLevel) bSameLevel = true;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = cell_off[c]; j < cell_off[c + 1]; j++) {
                const KeyPoint& kp = kps_un[cell_feat[j]];
                if (bCheckLevels && !bSameLevel) {
                    if (kp.octave < minLevel || kp.octave > maxLevel) continue;
                } else if (bSameLevel) {
                    if (kp.octave != minLevel) continue;
                }
                if (fabsf(kp.x - x) > r || fabsf(kp.y - y) > r) continue;
                out[n++] = cell_feat[j];
            }
        }
    return n;
}
}
