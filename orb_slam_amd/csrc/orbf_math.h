// Scalar arithmetic of the Frame-side steps (include/orbf.h), written once for host and device like orb_math.h:
// IEEE +,-,*,/ in double / float only, no FMA contraction (-ffp-contract=off on every build of this header).
//
// Reference call sites (in /root/reference):
//   cv::undistortPoints   src/Frame.cc:303 (keypoints), :335 (image corners)   — OpenCV 2.4 imgproc/src/undistort.cpp
//   Frame::PosInGrid      src/Frame.cc:267-277
//   window cell range     src/Frame.cc:205-223 (Frame::GetFeaturesInArea)
#pragma once
#include <math.h>
#include <stdint.h>

#include "orb_math.h"
#include "orbf.h"

namespace orbf {

// cvUndistortPoints for one CV_32FC2 point, R = identity, P = the camera matrix: everything in double, five
// fixed-point iterations when distortion coefficients are present, result stored as float.
ORBX_HD void undistort_point(const orbf_camera& c, float xin, float yin, float* xo, float* yo) {
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 8; i++) if (i < c.ndist) k[i] = (double)c.dist[i];
    const int iters = c.ndist > 0 ? 5 : 1;
    const double fx = (double)c.K[0], fy = (double)c.K[4], cx = (double)c.K[2], cy = (double)c.K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = (double)xin, y = (double)yin;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < iters; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = (double)c.K[0] * x + (double)c.K[1] * y + (double)c.K[2];
    const double yy = (double)c.K[3] * x + (double)c.K[4] * y + (double)c.K[5];
    const double ww = 1. / ((double)c.K[6] * x + (double)c.K[7] * y + (double)c.K[8]);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

// Frame::PosInGrid: the cell x*48 + y, or -1 when the (undistorted) point rounds outside the grid
ORBX_HD int grid_cell(const orbf_bounds& b, float x, float y) {
    const int px = (int)roundf((x - (float)b.min_x) * b.inv_w);
    const int py = (int)roundf((y - (float)b.min_y) * b.inv_h);
    if (px < 0 || px >= ORBF_GRID_COLS || py < 0 || py >= ORBF_GRID_ROWS) return -1;
    return px * ORBF_GRID_ROWS + py;
}

// Frame::GetFeaturesInArea's cell window; false = the early `return vIndices` exits
ORBX_HD bool window_cells(const orbf_bounds& b, float x, float y, float r, int* x0, int* x1, int* y0, int* y1) {
    int a = (int)floorf((x - (float)b.min_x - r) * b.inv_w);
    a = a > 0 ? a : 0;
    if (a >= ORBF_GRID_COLS) return false;
    int c = (int)ceilf((x - (float)b.min_x + r) * b.inv_w);
    c = c < ORBF_GRID_COLS - 1 ? c : ORBF_GRID_COLS - 1;
    if (c < 0) return false;
    int d = (int)floorf((y - (float)b.min_y - r) * b.inv_h);
    d = d > 0 ? d : 0;
    if (d >= ORBF_GRID_ROWS) return false;
    int e = (int)ceilf((y - (float)b.min_y + r) * b.inv_h);
    e = e < ORBF_GRID_ROWS - 1 ? e : ORBF_GRID_ROWS - 1;
    if (e < 0) return false;
    *x0 = a; *x1 = c; *y0 = d; *y1 = e;
    return true;
}

// the octave filter and the |dx|,|dy| <= r box of GetFeaturesInArea (src/Frame.cc:241-256)
// the octave filter alone (src/Frame.cc:214-221, :241-249): no check when both arguments are -1, equality when they are equal,
// otherwise the closed range — which is empty for maxLevel < minLevel (e.g. the callers' (level, -1))
ORBX_HD bool level_passes(int octave, int minLevel, int maxLevel) {
    const bool check = !(minLevel == -1 && maxLevel == -1);
    const bool same = check && minLevel == maxLevel;
    if (check && !same) { if (octave < minLevel || octave > maxLevel) return false; }
    else if (same) { if (octave != minLevel) return false; }
    return true;
}

ORBX_HD bool in_window(float kx, float ky, int octave, float x, float y, float r, int minLevel, int maxLevel) {
    if (!level_passes(octave, minLevel, maxLevel)) return false;
    if (fabsf(kx - x) > r || fabsf(ky - y) > r) return false;
    return true;
}

}  // namespace orbf
