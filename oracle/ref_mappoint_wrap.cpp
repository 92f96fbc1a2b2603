// =====================================================================================
// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// The reference's OWN src/MapPoint.cc + include/MapPoint.h (and src/ORBmatcher.cc for DescriptorDistance), compiled where they lie
// against oracle/matcherstub with -DORB_ORACLE_REAL_MAPPOINT (KeyFrame / Frame / Map stand-ins, a no-op boost::mutex) ->
// oracle/_ref/libref_mappoint.so.  Pins orc_distinctive (oracle/search_oracle.cpp) to MapPoint::ComputeDistinctiveDescriptors
// (src/MapPoint.cc:185-250); tests/test_ref_pin_matcher.py::test_distinctive_descriptor.
// =====================================================================================
#include <vector>

#include "MapPoint.h"

// the stand-in Frame's statics (ORBmatcher.cc, compiled into this library for DescriptorDistance, refers to them)
float ORB_SLAM::Frame::fx = 0, ORB_SLAM::Frame::fy = 0, ORB_SLAM::Frame::cx = 0, ORB_SLAM::Frame::cy = 0;
int ORB_SLAM::Frame::mnMinX = 0, ORB_SLAM::Frame::mnMaxX = 0, ORB_SLAM::Frame::mnMinY = 0, ORB_SLAM::Frame::mnMaxY = 0;

extern "C" {

// N observations, observation i = descriptor row i seen by key frame i (key frames in ascending address order = the std::map order the
// reference iterates in); kf_bad[i] marks key frames that are skipped.  Writes the chosen descriptor (32 bytes); returns 1 if one was chosen.
int ref_distinctive(const uint8_t* desc, int N, const uint8_t* kf_bad, uint8_t* out) {
    using namespace ORB_SLAM;
    std::vector<KeyFrame> kfs(N > 0 ? N : 1);
    for (int i = 0; i < N; i++) {
        kfs[i].descriptors = cv::Mat(1, 32, CV_8U);
        memcpy(kfs[i].descriptors.ptr<uchar>(0), desc + (size_t)i * 32, 32);
        kfs[i].bad = kf_bad && kf_bad[i];
        kfs[i].mapPoints.assign(1, (MapPoint*)0);
    }
    Map map;
    cv::Mat pos(3, 1, CV_32F);
    MapPoint mp(pos, &kfs[0], &map);
    for (int i = 0; i < N; i++) mp.AddObservation(&kfs[i], 0);
    mp.ComputeDistinctiveDescriptors();
    cv::Mat d = mp.GetDescriptor();
    if (d.empty()) return 0;
    memcpy(out, d.ptr<uchar>(0), 32);
    return 1;
}

}  // extern "C"
