// =====================================================================================
// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// C entry points in front of the reference's OWN src/Frame.cc + include/Frame.h, compiled where they lie (oracle/Makefile ->
// oracle/_ref/libref_frame.so) against oracle/matcherstub with -DORB_ORACLE_REAL_FRAME: MapPoint.h / KeyFrame.h / Converter.h /
// ORBVocabulary.h / ORBextractor.h are cut by their include guards and replaced by plain-data stand-ins, the "extractor" hands the
// constructor preset key points.  What runs from the reference's source text and is pinned by tests/test_ref_pin_frame.py:
//   Frame::Frame(...)            src/Frame.cc:55-128  the static grid constants, the scale tables, the mGrid fill
//   Frame::PosInGrid             :267-277
//   Frame::ComputeImageBounds    :321-351  (the corner logic; the undistortion primitive below is the oracle's)
//   Frame::UndistortKeyPoints    :289-319  (the packing around the primitive)
//   Frame::GetFeaturesInArea     :200-265
// cv::undistortPoints is an OpenCV primitive (absent): it forwards to the oracle's restatement, so IT stays unpinned.
// =====================================================================================
#include <memory>

#include "Frame.h"

namespace {
struct OKeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };
struct OCamera { float K[9]; float dist[8]; int32_t ndist; int32_t width, height; };
struct OBounds { int32_t min_x, max_x, min_y, max_y; float inv_w, inv_h; };
std::unique_ptr<ORB_SLAM::Frame> g_frame;
ORB_SLAM::ORBextractor g_extractor;
ORB_SLAM::ORBVocabulary g_voc;
}
extern "C" void orc_frame_undistort(const void* cam, const void* kps, int n, void* out);

namespace cv {
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& dist, const Mat&, const Mat&) {
    OCamera c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < 9; i++) c.K[i] = K.at<float>(i / 3, i % 3);
    const int nd = dist.rows * dist.cols;
    for (int i = 0; i < nd && i < 8; i++) c.dist[i] = dist.at<float>(i);
    c.ndist = nd;
    const int n = src.rows;
    std::vector<OKeyPoint> in(n > 0 ? n : 1), out(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) { in[i].x = src.at<float>(i, 0); in[i].y = src.at<float>(i, 1); }
    orc_frame_undistort(&c, in.data(), n, out.data());
    for (int i = 0; i < n; i++) { dst.at<float>(i, 0) = out[i].x; dst.at<float>(i, 1) = out[i].y; }
}
}  // namespace cv

extern "C" {

// Builds a Frame the reference's way (first frame of a run: the static bounds / grid constants are recomputed).  Outputs: bounds (incl. the
// two inverse cell sizes), undistorted key points, the grid as CSR (cell = x * 48 + y, push_back order).  Returns mvKeysUn.size().
int ref_frame_build(const void* cam_, const void* kps_, int n, void* bounds_out, void* kps_un_out, int32_t* cell_off, int32_t* cell_feat) {
    using namespace ORB_SLAM;
    const OCamera& c = *(const OCamera*)cam_;
    const OKeyPoint* kps = (const OKeyPoint*)kps_;
    g_extractor.preset.assign(n, cv::KeyPoint());
    for (int i = 0; i < n; i++) memcpy(&g_extractor.preset[i], &kps[i], sizeof(OKeyPoint));
    cv::Mat im(c.height, c.width, CV_8U), K(3, 3, CV_32F), D(c.ndist > 0 ? c.ndist : 4, 1, CV_32F);
    for (int i = 0; i < 9; i++) K.at<float>(i / 3, i % 3) = c.K[i];
    for (int i = 0; i < c.ndist; i++) D.at<float>(i) = c.dist[i];
    Frame::mbInitialComputations = true;
    g_frame.reset(new Frame(im, 0.0, &g_extractor, &g_voc, K, D));
    Frame& F = *g_frame;
    OBounds b = {Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv};
    memcpy(bounds_out, &b, sizeof(b));
    for (size_t i = 0; i < F.mvKeysUn.size(); i++) memcpy((OKeyPoint*)kps_un_out + i, &F.mvKeysUn[i], sizeof(OKeyPoint));
    int pos = 0;
    for (int x = 0; x < FRAME_GRID_COLS; x++)
        for (int y = 0; y < FRAME_GRID_ROWS; y++) {
            cell_off[x * FRAME_GRID_ROWS + y] = pos;
            for (size_t k = 0; k < F.mGrid[x][y].size(); k++) cell_feat[pos++] = (int32_t)F.mGrid[x][y][k];
        }
    cell_off[FRAME_GRID_COLS * FRAME_GRID_ROWS] = pos;
    return (int)F.mvKeysUn.size();
}

// Frame::GetFeaturesInArea on the frame built last
int ref_frame_features_in_area(float x, float y, float r, int minLevel, int maxLevel, int32_t* out) {
    if (!g_frame) return -1;
    const std::vector<size_t> v = g_frame->GetFeaturesInArea(x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size(); i++) out[i] = (int32_t)v[i];
    return (int)v.size();
}

// the scale tables the constructor derives from GetLevels() / GetScaleFactor()
int ref_frame_scale_tables(float* factors, float* sigma2, float* inv_sigma2) {
    if (!g_frame) return -1;
    for (int i = 0; i < g_frame->mnScaleLevels; i++) { factors[i] = g_frame->mvScaleFactors[i]; sigma2[i] = g_frame->mvLevelSigma2[i]; inv_sigma2[i] = g_frame->mvInvLevelSigma2[i]; }
    return g_frame->mnScaleLevels;
}

}  // extern "C"
