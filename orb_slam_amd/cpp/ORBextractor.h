// ORB_SLAM::ORBextractor with the reference's class surface (reference include/ORBextractor.h:31-79),
// implemented on the MI355X C ABI (include/orbx.h).  Frame / Tracking call it unchanged:
//     (*mpORBextractor)(im, cv::Mat(), mvKeys, mDescriptors);          // reference src/Frame.cc:60
// Differences from the reference, all documented in INTEGRATION.md:
//   * device errors throw std::runtime_error (the reference cannot fail there);
//   * the mask argument is accepted and ignored — it has no effect in the reference either
//     (cellMask is built but never passed to cv::FAST, reference src/ORBextractor.cc:601-607).
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "cvcompat.h"
#include "orbx.h"

namespace ORB_SLAM {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures = 1000, float scaleFactor = 1.2f, int nlevels = 8, int scoreType = FAST_SCORE, int fastTh = 20,
                 int device = 0)
        : h_(nullptr), nlevels_(nlevels), scaleFactor_(scaleFactor) {
        orbx_params p;
        orbx_default_params(&p);
        p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels; p.score_type = scoreType; p.fast_th = fastTh;
        p.device = device;
        const int rc = orbx_create(&p, &h_);
        if (rc != ORBX_OK) throw std::runtime_error("orbx_create failed (" + std::to_string(rc) + "): no usable MI355X / HIP runtime");
        cap_ = orbx_max_keypoints(h_);
    }
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image (reference include/ORBextractor.h:43-45).  Written against the proxy
    // classes the way the reference's own body is (src/ORBextractor.cc:718-742: _image.empty(), _image.getMat(),
    // _descriptors.release() / create() / getMat()), so it compiles against a real OpenCV 2.4 (-DORBX_WITH_OPENCV) as well as
    // against cvcompat.h.
    void operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors) {
        if (_image.empty()) return;                                  // reference :721-722: outputs untouched
        cv::Mat image = _image.getMat();
        kps_.resize(cap_);
        desc_.resize((size_t)cap_ * 32);
        int n = 0;
        const int rc = orbx_extract(h_, image.data, image.cols, image.rows, (ptrdiff_t)image.step,
                                    reinterpret_cast<orbx_keypoint*>(kps_.data()), desc_.data(), cap_, &n);
        if (rc == ORBX_EMPTY) return;
        if (rc != ORBX_OK) throw std::runtime_error(std::string("orbx_extract: ") + orbx_last_error(h_));
        if (n == 0) _descriptors.release();                           // reference :738-739
        else {
            _descriptors.create(n, 32, CV_8U);                        // reference :742
            cv::Mat descriptors = _descriptors.getMat();
            for (int i = 0; i < n; i++) std::memcpy(descriptors.ptr(i), desc_.data() + (size_t)i * 32, 32);
        }
        _keypoints.assign(kps_.begin(), kps_.begin() + n);            // reference :746-747,:777
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return (float)scaleFactor_; }

private:
    orbx_extractor* h_;
    int nlevels_;
    double scaleFactor_;   // the reference keeps a double member initialised from the float argument
    int cap_;
    std::vector<cv::KeyPoint> kps_;
    std::vector<unsigned char> desc_;
};
static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "KeyPoint layout");

}  // namespace ORB_SLAM
