// oracle/matcherstub: <opencv2/opencv.hpp> as /root/reference/include/Frame.h and src/Frame.cc use it (TEST INFRASTRUCTURE ONLY).
#ifndef ORB_ORACLE_MATCHERSTUB_OPENCV_HPP
#define ORB_ORACLE_MATCHERSTUB_OPENCV_HPP
#include "core/core.hpp"
namespace cv {
// OpenCV primitive, not part of the reference: forwards to the oracle's restatement of cvUndistortPoints (oracle/frame_oracle.cpp);
// defined in oracle/ref_frame_wrap.cpp.  src / dst: N x 2 CV_32F (see Mat::reshape), R empty, P = cameraMatrix.
void undistortPoints(const Mat& src, Mat& dst, const Mat& cameraMatrix, const Mat& distCoeffs, const Mat& R, const Mat& P);
}
#endif
