// =====================================================================================
// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// Stand-in for <opencv2/core/core.hpp> as /root/reference/src/ORBmatcher.cc uses it, so that the reference's OWN
// translation unit can be compiled where it lies (oracle/Makefile -> oracle/_ref/libref_orbmatcher.so) and its search
// functions executed beside the oracle restatements of oracle/search_oracle.cpp.  Separate from oracle/cvstub (the
// extractor's stand-in): ORBmatcher.cc needs value-semantics float matrices (R*x + t, .t(), rowRange ...), not images.
// The matrix algebra is plain float arithmetic; it only runs in the projection-based functions, which are compiled but
// NOT used for pinning (see ref_orbmatcher_wrap.cpp for the list of functions that are).
// =====================================================================================
#ifndef ORB_ORACLE_MATCHERSTUB_CORE_HPP
#define ORB_ORACLE_MATCHERSTUB_CORE_HPP
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5

namespace cv {
using std::vector;

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;

struct KeyPoint {               // 28 bytes, the layout of OpenCV 2.4's cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};

class Mat {
public:
    int rows, cols;
    uchar* data;
    size_t step;

    Mat() : rows(0), cols(0), data(0), step(0), type_(CV_8U) {}
    Mat(int r, int c, int type) : rows(r), cols(c), data(0), step(0), type_(type) {
        step = (size_t)c * elemSize();
        store_.reset(new std::vector<uchar>((size_t)r * step + 16, 0));
        data = store_->data();
    }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    int type() const { return type_; }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }

    Mat view(int r0, int r1, int c0, int c1) const {
        Mat m(*this);
        m.data = data + (size_t)r0 * step + (size_t)c0 * elemSize();
        m.rows = r1 - r0; m.cols = c1 - c0;
        return m;
    }
    Mat rowRange(int a, int b) const { return view(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return view(0, rows, a, b); }
    Mat row(int y) const { return view(y, y + 1, 0, cols); }
    Mat col(int x) const { return view(0, rows, x, x + 1); }
    void copyTo(Mat& dst) const { dst = clone(); }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    Mat reshape(int /*channels*/) const { return *this; }      // N x 2 CV_32F <-> N x 1 CV_32FC2: the same bytes (no channel model here)
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
        return m;
    }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
    template <typename T> T& at(int y, int x) { return ((T*)(data + (size_t)y * step))[x]; }
    template <typename T> const T& at(int y, int x) const { return ((const T*)(data + (size_t)y * step))[x]; }
    // single index: element i of a vector (column vector: row i; row vector: column i)
    template <typename T> T& at(int i) { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }
    template <typename T> const T& at(int i) const { return cols == 1 ? at<T>(i, 0) : at<T>(0, i); }

    // container semantics of cv::Mat::create / release (as far as the extractor shim and Frame use them)
    void create(int r, int c, int type) {
        if (data && r == rows && c == cols && type == type_) return;
        *this = Mat(r, c, type);
    }
    void release() { store_.reset(); data = 0; rows = cols = 0; step = 0; }
    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }

    Mat t() const {
        Mat m(cols, rows, CV_32F);
        for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) m.at<float>(x, y) = at<float>(y, x);
        return m;
    }
    double dot(const Mat& o) const {
        double s = 0;
        for (int y = 0; y < rows; y++) for (int x = 0; x < cols; x++) s += (double)at<float>(y, x) * o.at<float>(y, x);
        return s;
    }

private:
    int type_;
    std::shared_ptr<std::vector<uchar> > store_;
};

// OpenCV 2.4's proxy classes (core.hpp: `typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray;`) as far as
// the ORBextractor signature and body use them — for the Frame pin that runs the reference's constructor against the PRODUCT's
// orb_slam_amd/cpp/ORBextractor.h (oracle/ref_frame_product_wrap.cpp)
class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }
protected:
    const Mat* m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) : _InputArray(m) {}
    void create(int r, int c, int type) const { const_cast<Mat*>(m_)->create(r, c, type); }
    void release() const { const_cast<Mat*>(m_)->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

inline Mat operator*(const Mat& a, const Mat& b) {
    Mat m(a.rows, b.cols, CV_32F);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < b.cols; x++) {
            float s = 0;
            for (int k = 0; k < a.cols; k++) s += a.at<float>(y, k) * b.at<float>(k, x);
            m.at<float>(y, x) = s;
        }
    return m;
}
template <typename F> inline Mat map2(const Mat& a, const Mat& b, F f) {
    Mat m(a.rows, a.cols, CV_32F);
    for (int y = 0; y < a.rows; y++) for (int x = 0; x < a.cols; x++) m.at<float>(y, x) = f(a.at<float>(y, x), b.at<float>(y, x));
    return m;
}
inline Mat operator+(const Mat& a, const Mat& b) { return map2(a, b, [](float p, float q) { return p + q; }); }
inline Mat operator-(const Mat& a, const Mat& b) { return map2(a, b, [](float p, float q) { return p - q; }); }
inline Mat operator*(const Mat& a, double s) { return map2(a, a, [s](float p, float) { return (float)(p * s); }); }
inline Mat operator*(double s, const Mat& a) { return a * s; }
inline Mat operator/(const Mat& a, double s) { return map2(a, a, [s](float p, float) { return (float)(p / s); }); }
inline Mat operator-(const Mat& a) { return a * -1.0; }
inline double norm(const Mat& a) { return std::sqrt(a.dot(a)); }

}  // namespace cv
#endif
