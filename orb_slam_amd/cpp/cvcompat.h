// cvcompat.h — the few OpenCV 2.4 types the ORBextractor / ORBmatcher signatures mention, for builds
// without OpenCV (this container has none).  Field layouts match OpenCV 2.4 so that code written against
// the real headers compiles unchanged: define ORBX_WITH_OPENCV to use the real <opencv2/core/core.hpp>.
#pragma once
#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

typedef unsigned char uchar;

template <typename T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;
typedef Point_<int> Point;

// cv::KeyPoint (OpenCV 2.4 features2d.hpp): 28 bytes, same field order as orbx_keypoint
struct KeyPoint {
    Point2f pt;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint must be 28 bytes");

// 8-bit single-channel 2-D matrix: enough of cv::Mat for image in / descriptor out
class Mat {
public:
    int rows, cols;
    size_t step;
    uchar* data;
    Mat() : rows(0), cols(0), step(0), data(nullptr) {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, void* ext, size_t step_ = 0) : rows(r), cols(c), step(step_ ? step_ : (size_t)c), data((uchar*)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && owner_ && step == (size_t)c) return;
        owner_ = std::make_shared<std::vector<uchar>>((size_t)r * c);
        rows = r; cols = c; step = (size_t)c; data = owner_->data();
    }
    void release() { owner_.reset(); rows = cols = 0; step = 0; data = nullptr; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    Mat row(int r) const { Mat m(1, cols, CV_8UC1, data + (size_t)r * step, step); m.owner_ = owner_; return m; }
    bool isContinuous() const { return step == (size_t)cols; }
private:
    std::shared_ptr<std::vector<uchar>> owner_;
};

// the proxy classes of OpenCV 2.4's core.hpp (`typedef const _InputArray& InputArray; typedef const _OutputArray& OutputArray;`),
// reduced to what the ORBextractor signature and body use: a Mat (lvalue or temporary) converts implicitly, as there
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

}  // namespace cv
#endif
