// =====================================================================================
// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// A minimal stand-in for the OpenCV 2.4 headers the reference sources include
// (<opencv2/core/core.hpp>, <opencv2/highgui/highgui.hpp>, <opencv/cv.h>), so that the reference's OWN
// translation units — /root/reference/src/ORBextractor.cc and Thirdparty/DBoW2/DBoW2/*.cpp — can be compiled where
// they lie (oracle/Makefile → oracle/_ref/) and executed beside the oracle restatement.  OpenCV itself is a
// third-party dependency that is neither vendored in the reference tree nor installed in this image.
//
// What this pins and what it does not:
//  * everything ORB_SLAM / DBoW2 compute THEMSELVES (grid, quotas, fallback, retain order, IC_Angle, rBRIEF,
//    scale chains, octave bookkeeping; FORB::distance, the vocabulary descent, BowVector / FeatureVector, scores)
//    runs from the reference's source text → the oracle restatement of those parts is pinned by it;
//  * the OpenCV PIXEL primitives (resize, FAST, GaussianBlur, retainBest, fastAtan2) are NOT OpenCV here: the
//    functions below forward to the same Appendix-A restatements the oracle uses (liborb_oracle.so), so they stay
//    unpinned.  Container semantics that matter to the reference (ROI views sharing a parent buffer, create() as a
//    no-op on a matching view, `Mat = Mat::zeros()` filling in place, copyMakeBorder's parent-peeking without
//    BORDER_ISOLATED) are modelled after OpenCV 2.4's core/src/matrix.cpp and copy.cpp behaviour.
// =====================================================================================
#ifndef ORB_ORACLE_CVSTUB_CORE_HPP
#define ORB_ORACLE_CVSTUB_CORE_HPP
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_PI 3.1415926535897932384626433832795
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_32FC1 5
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error(std::string("CV_Assert failed: ") + #expr); } while (0)

// ties-to-even under the default rounding mode, as cvRound's SSE2 / lrint paths
inline int cvRound(double v) { return (int)lrint(v); }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

namespace cv {
using std::vector;
using std::string;

enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4,
       BORDER_REFLECT101 = 4, BORDER_DEFAULT = 4, BORDER_ISOLATED = 16 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
// core/operations.hpp: a.x = saturate_cast<T>(a.x*b) — for T = float, b = float a plain float product
template <typename T> static inline Point_<T>& operator*=(Point_<T>& a, float b) { a.x = (T)(a.x * b); a.y = (T)(a.y * b); return a; }
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
struct Scalar {
    double val[4];
    Scalar(double v0 = 0, double v1 = 0, double v2 = 0, double v3 = 0) { val[0] = v0; val[1] = v1; val[2] = v2; val[3] = v3; }
};

struct KeyPoint {               // 28 bytes, the layout of OpenCV 2.4's cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
};

template <typename T, size_t N = 4096 / sizeof(T) + 8> class AutoBuffer {
public:
    explicit AutoBuffer(size_t n) : v_(n) {}
    operator T*() { return v_.data(); }
    operator const T*() const { return v_.data(); }
private:
    std::vector<T> v_;
};

class Mat;
struct MatExpr {                // only Mat::zeros is needed
    int rows, cols, type;
};

class Mat {
public:
    int rows, cols;
    uchar* data;
    size_t step;                // bytes per row

    Mat() : rows(0), cols(0), data(0), step(0), type_(0), wrows_(0), wcols_(0), base_(0) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(Size sz, int type) : Mat() { create(sz.height, sz.width, type); }
    Mat(int r, int c, int type, void* ext, size_t st = 0) : Mat() {      // user data, not owned
        rows = r; cols = c; type_ = type; data = (uchar*)ext; step = st ? st : (size_t)c * elemSize();
        base_ = data; wrows_ = r; wcols_ = c;
    }
    Mat(const Mat& m, const Rect& roi) : Mat(m) {
        CV_Assert(roi.x >= 0 && roi.y >= 0 && roi.width >= 0 && roi.height >= 0 && roi.x + roi.width <= m.cols && roi.y + roi.height <= m.rows);
        data = m.data + (size_t)roi.y * m.step + (size_t)roi.x * m.elemSize();
        rows = roi.height; cols = roi.width;
    }
    Mat(const MatExpr& e) : Mat() { *this = e; }
    Mat& operator=(const MatExpr& e) {          // MatExpr::assign for zeros: create() (no-op on a matching view) then fill
        create(e.rows, e.cols, e.type);
        for (int y = 0; y < rows; y++) memset(data + (size_t)y * step, 0, (size_t)cols * elemSize());
        return *this;
    }
    static MatExpr zeros(int r, int c, int type) { MatExpr e = {r, c, type}; return e; }

    void create(int r, int c, int type) {
        if (data && rows == r && cols == c && type_ == type) return;
        release();
        rows = r; cols = c; type_ = type;
        step = (size_t)c * elemSize();
        if ((size_t)r * step > 0) {
            store_.reset(new std::vector<uchar>((size_t)r * step));
            data = store_->data();
        }
        base_ = data; wrows_ = r; wcols_ = c;
    }
    void create(Size sz, int type) { create(sz.height, sz.width, type); }
    void release() { store_.reset(); data = 0; base_ = 0; rows = cols = 0; step = 0; wrows_ = wcols_ = 0; }

    int type() const { return type_; }
    int depth() const { return type_; }
    int channels() const { return 1; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    size_t elemSize1() const { return elemSize(); }
    size_t step1() const { return step / elemSize1(); }
    bool empty() const { return data == 0 || rows == 0 || cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    Size size() const { return Size(cols, rows); }
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * elemSize(); }
    bool isSubmatrix() const { return rows != wrows_ || cols != wcols_; }

    Mat operator()(const Rect& roi) const { return Mat(*this, roi); }
    Mat rowRange(int a, int b) const { return Mat(*this, Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return Mat(*this, Rect(a, 0, b - a, rows)); }
    Mat row(int y) const { return rowRange(y, y + 1); }

    void locateROI(Size& whole, Point& ofs) const {
        const size_t d = (size_t)(data - base_);
        const size_t wstep = step ? step : 1;
        ofs.y = (int)(d / wstep);
        ofs.x = (int)((d - (size_t)ofs.y * wstep) / elemSize());
        whole = Size(wcols_, wrows_);
    }
    Mat& adjustROI(int dtop, int dbottom, int dleft, int dright) {
        Size whole; Point ofs;
        locateROI(whole, ofs);
        int row1 = std::max(ofs.y - dtop, 0), row2 = std::min(ofs.y + rows + dbottom, whole.height);
        int col1 = std::max(ofs.x - dleft, 0), col2 = std::min(ofs.x + cols + dright, whole.width);
        data += (ptrdiff_t)(row1 - ofs.y) * (ptrdiff_t)step + (ptrdiff_t)(col1 - ofs.x) * (ptrdiff_t)elemSize();
        rows = row2 - row1; cols = col2 - col1;
        return *this;
    }

    void copyTo(Mat& dst) const {
        dst.create(rows, cols, type_);
        if (dst.data == data) return;
        for (int y = 0; y < rows; y++) memmove(dst.data + (size_t)y * dst.step, data + (size_t)y * step, (size_t)cols * elemSize());
    }
    Mat clone() const { Mat m; copyTo(m); return m; }
    Mat& setTo(const Scalar& s) {
        CV_Assert(type_ == CV_8U);
        for (int y = 0; y < rows; y++) memset(data + (size_t)y * step, (int)s.val[0], cols);
        return *this;
    }

    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
    template <typename T> T& at(int y, int x) { return ((T*)(data + (size_t)y * step))[x]; }
    template <typename T> const T& at(int y, int x) const { return ((const T*)(data + (size_t)y * step))[x]; }

private:
    int type_;
    int wrows_, wcols_;          // the parent ("whole") matrix this header is a view of
    uchar* base_;                // its first byte
    std::shared_ptr<std::vector<uchar> > store_;
};

class _InputArray {
public:
    _InputArray() : m_(0) {}
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int rows, int cols, int type) const { m_->create(rows, cols, type); }
    void create(Size sz, int type) const { m_->create(sz, type); }
    Mat getMat() const { return *m_; }
    void release() const { m_->release(); }
private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

// ---- FileStorage: syntax only (DBoW2's virtual save()/load() must compile); using it throws.
class FileNode {
public:
    FileNode operator[](const char*) const { throw std::logic_error("cvstub: FileStorage not available"); }
    FileNode operator[](const std::string&) const { throw std::logic_error("cvstub: FileStorage not available"); }
    FileNode operator[](int) const { throw std::logic_error("cvstub: FileStorage not available"); }
    size_t size() const { throw std::logic_error("cvstub: FileStorage not available"); }
    operator int() const { throw std::logic_error("cvstub: FileStorage not available"); }
    operator float() const { throw std::logic_error("cvstub: FileStorage not available"); }
    operator double() const { throw std::logic_error("cvstub: FileStorage not available"); }
    operator std::string() const { throw std::logic_error("cvstub: FileStorage not available"); }
};
class FileStorage {
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string&, int) {}
    bool isOpened() const { return false; }
    FileNode operator[](const char*) const { throw std::logic_error("cvstub: FileStorage not available"); }
    FileNode operator[](const std::string&) const { throw std::logic_error("cvstub: FileStorage not available"); }
    void release() {}
};
template <typename T> inline FileStorage& operator<<(FileStorage&, const T&) { throw std::logic_error("cvstub: FileStorage not available"); }

// ---- pixel primitives: forwarded to the oracle's Appendix-A restatements (cvstub_impl.cpp)
float fastAtan2(float y, float x);
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void copyMakeBorder(InputArray src, OutputArray dst, int top, int bottom, int left, int right, int borderType,
                    const Scalar& value = Scalar());
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_DEFAULT);
void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression = true);
struct ORB { enum { kBytes = 32, HARRIS_SCORE = 0, FAST_SCORE = 1 }; };   // features2d.hpp: only the enum is used
struct KeyPointsFilter {
    static void retainBest(std::vector<KeyPoint>& keypoints, int npoints);
};
}  // namespace cv
#endif
