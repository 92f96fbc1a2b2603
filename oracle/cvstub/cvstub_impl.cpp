// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY.  Definitions for oracle/cvstub/opencv2/core/core.hpp: the OpenCV pixel
// primitives the reference calls, forwarded to the oracle's restatements (liborb_oracle.so).
// The container-level behaviour of copyMakeBorder (parent peeking unless BORDER_ISOLATED; early-out when nothing is
// left to pad) follows OpenCV 2.4 core/src/copy.cpp.
#include <opencv2/core/core.hpp>

extern "C" {
struct orc_keypoint { float x, y, size, angle, response; int octave, class_id; };
void orc_resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
int orc_fast(const uint8_t* img, int w, int hh, int stride, int threshold, orc_keypoint* out, int cap, uint8_t* scores);
void orc_gaussian_blur7_level(uint8_t* whole, int w, int hh, int blur_mode);
int orc_retain_best_kps(orc_keypoint* kps, int count, int n);
float orc_fastAtan2(float y, float x);
int orc_reflect101(int p, int len);
}

static int g_blur_mode = 0;
extern "C" void cvstub_set_blur_mode(int m) { g_blur_mode = m; }

namespace cv {
static_assert(sizeof(KeyPoint) == 28 && sizeof(orc_keypoint) == 28, "KeyPoint layout");

float fastAtan2(float y, float x) { return orc_fastAtan2(y, x); }

void resize(InputArray _src, OutputArray _dst, Size dsize, double fx, double fy, int interpolation) {
    Mat src = _src.getMat();
    CV_Assert(src.type() == CV_8UC1 && interpolation == INTER_LINEAR && fx == 0 && fy == 0 && dsize.width > 0 && dsize.height > 0);
    _dst.create(dsize, src.type());
    Mat dst = _dst.getMat();
    orc_resize_linear_8u(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
}

void copyMakeBorder(InputArray _src, OutputArray _dst, int top, int bottom, int left, int right, int borderType, const Scalar& value) {
    Mat src = _src.getMat();
    CV_Assert(top >= 0 && bottom >= 0 && left >= 0 && right >= 0 && src.type() == CV_8UC1);
    if (src.isSubmatrix() && (borderType & BORDER_ISOLATED) == 0) {
        Size whole; Point ofs;
        src.locateROI(whole, ofs);
        int dtop = std::min(ofs.y, top), dbottom = std::min(whole.height - src.rows - ofs.y, bottom);
        int dleft = std::min(ofs.x, left), dright = std::min(whole.width - src.cols - ofs.x, right);
        src.adjustROI(dtop, dbottom, dleft, dright);
        top -= dtop; left -= dleft; bottom -= dbottom; right -= dright;
    }
    _dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
    Mat dst = _dst.getMat();
    if (top == 0 && left == 0 && bottom == 0 && right == 0) {
        if (src.data != dst.data || src.step != dst.step) src.copyTo(dst);
        return;
    }
    borderType &= ~BORDER_ISOLATED;
    CV_Assert(borderType == BORDER_REFLECT_101 || borderType == BORDER_CONSTANT);
    // interior first (memmove: src may be the centred view of dst), then the frame from the interior
    for (int y = 0; y < src.rows; y++) {
        uchar* d = dst.data + (size_t)(y + top) * dst.step + left;
        const uchar* s = src.data + (size_t)y * src.step;
        if (d != s) memmove(d, s, src.cols);
    }
    for (int y = 0; y < src.rows; y++) {
        uchar* row = dst.data + (size_t)(y + top) * dst.step + left;
        for (int x = -left; x < 0; x++) row[x] = borderType == BORDER_CONSTANT ? (uchar)value.val[0] : row[orc_reflect101(x, src.cols)];
        for (int x = src.cols; x < src.cols + right; x++) row[x] = borderType == BORDER_CONSTANT ? (uchar)value.val[0] : row[orc_reflect101(x, src.cols)];
    }
    for (int y = -top; y < src.rows + bottom; y++) {
        if (y >= 0 && y < src.rows) continue;
        uchar* d = dst.data + (size_t)(y + top) * dst.step;
        if (borderType == BORDER_CONSTANT) memset(d, (int)value.val[0], dst.cols);
        else memcpy(d, dst.data + (size_t)(orc_reflect101(y, src.rows) + top) * dst.step, dst.cols);
    }
}

void GaussianBlur(InputArray _src, OutputArray _dst, Size ksize, double sigmaX, double sigmaY, int borderType) {
    Mat src = _src.getMat();
    CV_Assert(src.type() == CV_8UC1 && ksize.width == 7 && ksize.height == 7 && sigmaX == 2 && sigmaY == 2);
    CV_Assert((borderType & ~BORDER_ISOLATED) == BORDER_REFLECT_101);
    _dst.create(src.size(), src.type());
    Mat dst = _dst.getMat();
    // the filter engine reads real parent pixels around a view (no BORDER_ISOLATED) and extrapolates beyond the parent
    Size whole(src.cols, src.rows); Point ofs(0, 0);
    const uchar* base = src.data;
    if (src.isSubmatrix() && (borderType & BORDER_ISOLATED) == 0) {
        src.locateROI(whole, ofs);
        base = src.data - (size_t)ofs.y * src.step - ofs.x;
    }
    const int B = 16, w = src.cols, h = src.rows, st = w + 2 * B;
    std::vector<uchar> buf((size_t)st * (h + 2 * B));
    for (int y = -B; y < h + B; y++) {
        const int py = orc_reflect101(y + ofs.y, whole.height);
        for (int x = -B; x < w + B; x++) {
            const int px = orc_reflect101(x + ofs.x, whole.width);
            buf[(size_t)(y + B) * st + (x + B)] = base[(size_t)py * src.step + px];
        }
    }
    orc_gaussian_blur7_level(buf.data(), w, h, g_blur_mode);
    for (int y = 0; y < h; y++) memcpy(dst.data + (size_t)y * dst.step, buf.data() + (size_t)(y + B) * st + B, w);
}

void FAST(InputArray _image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression) {
    Mat img = _image.getMat();
    CV_Assert(img.type() == CV_8UC1 && nonmaxSuppression);
    std::vector<orc_keypoint> out((size_t)img.rows * img.cols + 1);
    const int n = orc_fast(img.data, img.cols, img.rows, (int)img.step, threshold, out.data(), (int)out.size(), 0);
    CV_Assert(n >= 0);
    keypoints.clear();
    for (int i = 0; i < n; i++)
        keypoints.push_back(KeyPoint(out[i].x, out[i].y, out[i].size, out[i].angle, out[i].response, out[i].octave, out[i].class_id));
}

void KeyPointsFilter::retainBest(std::vector<KeyPoint>& keypoints, int npoints) {
    if (keypoints.empty()) return;
    const int n = orc_retain_best_kps((orc_keypoint*)&keypoints[0], (int)keypoints.size(), npoints);
    keypoints.resize(n);
}
}  // namespace cv
