// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY.  C entry points around the reference's own ORB_SLAM::ORBextractor
// (/root/reference/src/ORBextractor.cc, compiled where it lies against oracle/cvstub; see oracle/Makefile).
// Used by tests/test_ref_pin.py to check the oracle restatement against the reference's source text.
#include <opencv2/core/core.hpp>
#include "ORBextractor.h"

extern "C" {
void* ref_orb_create(int nfeatures, float scaleFactor, int nlevels, int scoreType, int fastTh) {
    return new ORB_SLAM::ORBextractor(nfeatures, scaleFactor, nlevels, scoreType, fastTh);
}
void ref_orb_destroy(void* h) { delete (ORB_SLAM::ORBextractor*)h; }
int ref_orb_levels(void* h) { return ((ORB_SLAM::ORBextractor*)h)->GetLevels(); }
float ref_orb_scale_factor(void* h) { return ((ORB_SLAM::ORBextractor*)h)->GetScaleFactor(); }
// the call of src/Frame.cc:60; returns the number of keypoints, -1 when cap is too small, -2 on a cv assertion
int ref_orb_extract(void* h, const unsigned char* img, int w, int hh, int stride, cv::KeyPoint* kps, unsigned char* desc, int cap) {
    try {
        cv::Mat im(hh, w, CV_8UC1, (void*)img, (size_t)stride);
        std::vector<cv::KeyPoint> keys;
        cv::Mat descriptors;
        (*(ORB_SLAM::ORBextractor*)h)(im, cv::Mat(), keys, descriptors);
        const int n = (int)keys.size();
        if (n > cap) return -1;
        if (n != descriptors.rows && !(n == 0 && descriptors.empty())) return -3;
        for (int i = 0; i < n; i++) { kps[i] = keys[i]; memcpy(desc + (size_t)i * 32, descriptors.ptr(i), 32); }
        return n;
    } catch (const std::exception&) { return -2; }
}
}
