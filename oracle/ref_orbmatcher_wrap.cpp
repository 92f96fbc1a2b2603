// =====================================================================================
// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY (never part of the product path).
//
// C entry points in front of the reference's OWN src/ORBmatcher.cc, compiled where it lies against oracle/matcherstub
// (oracle/Makefile -> oracle/_ref/libref_orbmatcher.so; only where /root/reference exists).  Each function rebuilds the few
// objects the reference function reads (Frame / KeyFrame / MapPoint stand-ins, plain data) from flattened arrays, calls the
// reference function and flattens what it wrote.  tests/test_ref_pin_matcher.py compares the result with the oracle
// restatements of oracle/search_oracle.cpp on seeded problems — that is what pins them.
//
// Pinned through here (reference file:line = src/ORBmatcher.cc):
//   SearchByProjection(Frame&, const vector<MapPoint*>&, float)   :49-125     (oracle rule 0)
//   WindowSearch(Frame&, Frame&, int, vector<MapPoint*>&, int, int) :409-516  (oracle rule 1)
//   SearchByProjection(Frame& Current, const Frame& Last, float)    :1507-1619 (oracle rule 2; identity pose, see below)
//   SearchByProjection(Frame& F1, Frame& F2, int windowSize, ...)   :519-596  (oracle rule 1 behind a projection; identity pose)
//   SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist) :1622-1746 (oracle rule 2 from a key frame; identity pose)
//   SearchForInitialization(...)                                    :598-713  (oracle rule 3)
//   SearchByBoW(KeyFrame*, Frame&, ...)                             :155-281
//   SearchByBoW(KeyFrame*, KeyFrame*, ...)                          :715-850
//   SearchForTriangulation(...) + CheckDistEpipolarLine             :852-1014, :136-153
//   ComputeThreeMaxima :1748-1789, DescriptorDistance :1794-1810
//   SearchBySim3 :1267-1505 (identity poses: its two scans = oracle rule 5, its agreement check)
//   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) :286-407 (oracle rule 2 without rotation check; identity Scw)
//   Fuse(KeyFrame*, vector<MapPoint*>&, th) :1016-1134 and Fuse(KeyFrame*, Scw, ...) :1136-1265 (oracle rule 5; identity pose)
// i.e. every function of ORBmatcher.cc.  (Non-identity poses only change where the windows lie: that arithmetic is the caller's side of the
// boundary and is not restated.)
// =====================================================================================
#include <deque>

#include "ORBmatcher.h"
#include <chrono>

using namespace ORB_SLAM;

namespace {
struct Open : ORBmatcher {       // the protected helpers
    Open(float r, bool c) : ORBmatcher(r, c) {}
    using ORBmatcher::CheckDistEpipolarLine;
    using ORBmatcher::ComputeThreeMaxima;
};

cv::Mat desc_mat(const uint8_t* d, int n) {
    cv::Mat m(n > 0 ? n : 1, 32, CV_8U);
    for (int i = 0; i < n; i++) memcpy(m.ptr<uchar>(i), d + (size_t)i * 32, 32);
    m.rows = n;
    return m;
}
cv::Mat desc_row(const uint8_t* d) {
    cv::Mat m(1, 32, CV_8U);
    memcpy(m.ptr<uchar>(0), d, 32);
    return m;
}
std::vector<cv::KeyPoint> kp_vec(const void* kps, int n) {
    const cv::KeyPoint* p = (const cv::KeyPoint*)kps;
    return std::vector<cv::KeyPoint>(p, p + n);
}
void fill_grid(GridView& g, const void* bounds, const int32_t* cell_off, const int32_t* cell_feat, int n) {
    memcpy(&g.bounds, bounds, sizeof(g.bounds));
#ifdef ORBM_STUB_REAL_SHAPES
    // the camera's statics ORB_SLAM's Frame::Frame computes once (src/Frame.cc:88-101): ORBmatcherAccess.h reads the bounds from there
    Frame::mnMinX = g.bounds.min_x; Frame::mnMaxX = g.bounds.max_x; Frame::mnMinY = g.bounds.min_y; Frame::mnMaxY = g.bounds.max_y;
    Frame::mfGridElementWidthInv = g.bounds.inv_w; Frame::mfGridElementHeightInv = g.bounds.inv_h;
#endif
    g.cell_off.assign(cell_off, cell_off + FRAME_GRID_COLS * FRAME_GRID_ROWS + 1);
    (void)n;
    g.cell_feat.assign(cell_feat, cell_feat + g.cell_off.back());      // only the features inside the image bounds are in the grid
    g.cell_feat.push_back(0);
}
DBoW2::FeatureVector fv_from_csr(const uint32_t* node, const int32_t* off, const uint32_t* feat, int nnodes) {
    DBoW2::FeatureVector fv;
    for (int a = 0; a < nnodes; a++)
        for (int j = off[a]; j < off[a + 1]; j++) fv.addFeature(node[a], feat[j]);
    return fv;
}
// state 0 = NULL, 1 = good map point, 2 = bad map point
std::vector<MapPoint*> map_points(std::deque<MapPoint>& pool, const uint8_t* state, int n) {
    std::vector<MapPoint*> v(n, (MapPoint*)0);
    for (int i = 0; i < n; i++)
        if (state[i]) { pool.emplace_back(); pool.back().bad = state[i] == 2; v[i] = &pool.back(); }
    return v;
}
int index_of(const std::vector<MapPoint*>& owner, MapPoint* p) {
    if (!p) return -1;
    for (size_t i = 0; i < owner.size(); i++) if (owner[i] == p) return (int)i;
    return -2;
}

// The pose every projection-based entry point below gives its frame / key frame (default: identity, which is what the CPU pin tests
// reproduce in plain float arithmetic).  tests/test_gpu_orbmatcher_dropin.py moves it (ref_set_pose / ref_set_sim3) to run the
// reference's and the product's projection code at a general pose; the two libraries are then compared with each other only.
float g_Rt[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, g_scale = 1.0f;        // R (row major), t; Scw = g_scale * [R|t]
float g_simRt[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}, g_sim_scale = 1.0f;  // SearchBySim3's R12, t12, s12
cv::Mat pose_R(const float* Rt = g_Rt) {
    cv::Mat R(3, 3, CV_32F);
    for (int i = 0; i < 9; i++) R.at<float>(i / 3, i % 3) = Rt[i];
    return R;
}
cv::Mat pose_t(const float* Rt = g_Rt) {
    cv::Mat t(3, 1, CV_32F);
    for (int i = 0; i < 3; i++) t.at<float>(i) = Rt[9 + i];
    return t;
}
cv::Mat pose_centre() {                        // Ow = -R' t
    cv::Mat o(3, 1, CV_32F);
    for (int c = 0; c < 3; c++) {
        double acc = 0;
        for (int r = 0; r < 3; r++) acc += (double)g_Rt[3 * r + c] * g_Rt[9 + r];
        o.at<float>(c) = (float)(0.0 - acc);
    }
    return o;
}
cv::Mat pose_44(float scale) {                 // [scale*R | scale*t; 0 0 0 1]
    cv::Mat T(4, 4, CV_32F);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T.at<float>(r, c) = scale * g_Rt[3 * r + c];
        T.at<float>(r, 3) = scale * g_Rt[9 + r];
    }
    T.at<float>(3, 3) = 1.0f;
    return T;
}
}  // namespace

float Frame::fx = 0, Frame::fy = 0, Frame::cx = 0, Frame::cy = 0;
int Frame::mnMinX = 0, Frame::mnMaxX = 0, Frame::mnMinY = 0, Frame::mnMaxY = 0;
#ifdef ORBM_STUB_REAL_SHAPES
float Frame::mfGridElementWidthInv = 0, Frame::mfGridElementHeightInv = 0;
#endif

// the duration of the LAST ORBmatcher method call made through this library, without the harness around it (tools/bench_orbmatcher_dropin.py)
static thread_local double g_last_call_ms = 0.0;
struct CallTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~CallTimer() { g_last_call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
#define TIMED(call) ([&] { CallTimer timer_; return (call); }())

extern "C" {

void ref_set_pose(const float* Rt, float scale) {
    static const float I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    memcpy(g_Rt, Rt ? Rt : I, sizeof(g_Rt));
    g_scale = Rt ? scale : 1.0f;
}
void ref_set_sim3(const float* Rt, float scale) {
    static const float I[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    memcpy(g_simRt, Rt ? Rt : I, sizeof(g_simRt));
    g_sim_scale = Rt ? scale : 1.0f;
}

void ref_matcher_three_maxima(const int32_t* sizes, int L, int32_t* ind) {
    std::vector<std::vector<int> > h(L);
    for (int i = 0; i < L; i++) h[i].assign(sizes[i], 0);
    int a = -1, b = -1, c = -1;
    Open(0.6f, true).ComputeThreeMaxima(h.data(), L, a, b, c);
    ind[0] = a; ind[1] = b; ind[2] = c;
}

int ref_matcher_descriptor_distance(const uint8_t* a, const uint8_t* b) { return ORBmatcher::DescriptorDistance(desc_row(a), desc_row(b)); }

int ref_matcher_check_epipolar(float x1, float y1, float x2, float y2, int octave2, const float* F12, const float* sigma2, int nlevels) {
    cv::KeyPoint k1, k2;
    k1.pt.x = x1; k1.pt.y = y1; k2.pt.x = x2; k2.pt.y = y2; k2.octave = octave2;
    cv::Mat F(3, 3, CV_32F);
    for (int i = 0; i < 9; i++) F.at<float>(i / 3, i % 3) = F12[i];
    KeyFrame kf;
    kf.levelSigma2.assign(sigma2, sigma2 + nlevels);
    return Open(0.6f, true).CheckDistEpipolarLine(k1, k2, F, &kf) ? 1 : 0;
}

// rule 0.  Query q = a MapPoint: qstate 0 = not in view, 1 = in view and good, 2 = in view but bad.  t2q[nt]: the query whose map point the
// frame feature holds afterwards (-1 none, -2 = held a map point on entry).  Returns the function's return value.
int ref_search_by_projection_mappoints(const void* bounds, float ratio, float th, const void* kps_un, const uint8_t* desc, const int32_t* cell_off,
                                       const int32_t* cell_feat, int nt, const uint8_t* claimed, const float* scale_factors, int nlevels,
                                       const float* qxy, const int32_t* qlevel, const float* qviewcos, const uint8_t* qdesc, const uint8_t* qstate, int nq,
                                       int32_t* t2q) {
    Frame F;
    F.mvKeysUn = kp_vec(kps_un, nt);
    F.mDescriptors = desc_mat(desc, nt);
    F.mvScaleFactors.assign(scale_factors, scale_factors + nlevels);
    F.mnScaleLevels = nlevels;
    fill_grid(F.grid, bounds, cell_off, cell_feat, nt);
    std::deque<MapPoint> pool;
    MapPoint old;                                    // what the already-claimed features point to
    F.mvpMapPoints.assign(nt, (MapPoint*)0);
    for (int i = 0; i < nt; i++) if (claimed && claimed[i]) F.mvpMapPoints[i] = &old;
    std::vector<MapPoint*> q(nq);
    for (int i = 0; i < nq; i++) {
        pool.emplace_back();
        MapPoint& m = pool.back();
        m.mbTrackInView = qstate[i] != 0; m.bad = qstate[i] == 2;
        m.mTrackProjX = qxy[2 * i]; m.mTrackProjY = qxy[2 * i + 1];
        m.mnTrackScaleLevel = qlevel[i]; m.mTrackViewCos = qviewcos[i];
        m.descriptor = desc_row(qdesc + (size_t)i * 32);
        q[i] = &m;
    }
    ORBmatcher matcher(ratio, true);
    const int n = TIMED(matcher.SearchByProjection(F, q, th));
    for (int i = 0; i < nt; i++) t2q[i] = F.mvpMapPoints[i] == &old ? -2 : index_of(q, F.mvpMapPoints[i]);
    return n;
}

// rule 1.  state1[i1]: 0 = no map point, 1 = good, 2 = bad.  t2q[n2] = the F1 feature whose map point vpMapPointMatches2[i2] is.
int ref_window_search(const void* bounds, float ratio, int check, const void* kps_un1, const uint8_t* desc1, const uint8_t* state1, int n1,
                      const void* kps_un2, const uint8_t* desc2, const int32_t* cell_off2, const int32_t* cell_feat2, int n2, int windowSize,
                      int minLevel, int maxLevel, int32_t* t2q) {
    Frame F1, F2;
    F1.mvKeysUn = kp_vec(kps_un1, n1); F1.mDescriptors = desc_mat(desc1, n1);
    F2.mvKeysUn = kp_vec(kps_un2, n2); F2.mDescriptors = desc_mat(desc2, n2);
    F2.mvpMapPoints.assign(n2, (MapPoint*)0);
    fill_grid(F2.grid, bounds, cell_off2, cell_feat2, n2);
    std::deque<MapPoint> pool;
    F1.mvpMapPoints = map_points(pool, state1, n1);
    std::vector<MapPoint*> m2;
    ORBmatcher matcher(ratio, check != 0);
    const int n = TIMED(matcher.WindowSearch(F1, F2, windowSize, m2, minLevel, maxLevel));
    for (int i = 0; i < n2; i++) t2q[i] = index_of(F1.mvpMapPoints, m2[i]);
    return n;
}

// rule 3.  prev[2*n1] = vbPrevMatched (updated in place as the reference does), q2t[n1] = vnMatches12.
int ref_search_for_initialization(const void* bounds, float ratio, int check, const void* kps_un1, const uint8_t* desc1, int n1, const void* kps_un2,
                                  const uint8_t* desc2, const int32_t* cell_off2, const int32_t* cell_feat2, int n2, float* prev, int windowSize,
                                  int32_t* q2t) {
    Frame F1, F2;
    F1.mvKeysUn = kp_vec(kps_un1, n1); F1.mDescriptors = desc_mat(desc1, n1);
    F2.mvKeysUn = kp_vec(kps_un2, n2); F2.mDescriptors = desc_mat(desc2, n2);
    fill_grid(F2.grid, bounds, cell_off2, cell_feat2, n2);
    std::vector<cv::Point2f> pm(n1);
    for (int i = 0; i < n1; i++) pm[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);
    std::vector<int> m12;
    ORBmatcher matcher(ratio, check != 0);
    const int n = TIMED(matcher.SearchForInitialization(F1, F2, pm, m12, windowSize));
    for (int i = 0; i < n1; i++) { q2t[i] = m12[i]; prev[2 * i] = pm[i].x; prev[2 * i + 1] = pm[i].y; }
    return n;
}

// rule 2: SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th) with CurrentFrame.mTcw = identity, so that the
// reference's own projection code maps the world point (X, Y, 1) to (fx*X + cx, fy*Y + cy) in plain float arithmetic the test
// reproduces.  state1: 0 = no map point, 1 = map point; t2q[n2] = the last-frame feature whose map point the current feature holds
// afterwards (-1 none, -2 held one on entry).
int ref_search_by_projection_last_frame(const void* bounds, float ratio, int check, float th, const float* cam /* fx fy cx cy */, const void* kps_un2,
                                        const uint8_t* desc2, const int32_t* cell_off2, const int32_t* cell_feat2, int n2, const uint8_t* claimed2,
                                        const float* scale_factors, int nlevels, const void* kps1, const uint8_t* desc1, const float* world1,
                                        const uint8_t* state1, const uint8_t* outlier1, int n1, int32_t* t2q) {
    Frame C, L;
    Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
    GridView::Bounds bb;
    memcpy(&bb, bounds, sizeof(bb));
    Frame::mnMinX = bb.min_x; Frame::mnMaxX = bb.max_x; Frame::mnMinY = bb.min_y; Frame::mnMaxY = bb.max_y;
    C.mvKeysUn = kp_vec(kps_un2, n2); C.mDescriptors = desc_mat(desc2, n2);
    C.mvScaleFactors.assign(scale_factors, scale_factors + nlevels); C.mnScaleLevels = nlevels;
    fill_grid(C.grid, bounds, cell_off2, cell_feat2, n2);
    C.mTcw = pose_44(1.0f);
    MapPoint old;
    C.mvpMapPoints.assign(n2, (MapPoint*)0);
    for (int i = 0; i < n2; i++) if (claimed2 && claimed2[i]) C.mvpMapPoints[i] = &old;
    L.mvKeys = kp_vec(kps1, n1); L.mvKeysUn = L.mvKeys; L.mDescriptors = desc_mat(desc1, n1);
    std::deque<MapPoint> pool;
    L.mvpMapPoints = map_points(pool, state1, n1);
    L.mvbOutlier.assign(n1, false);
    for (int i = 0; i < n1; i++) {
        L.mvbOutlier[i] = outlier1[i] != 0;
        if (L.mvpMapPoints[i]) {
            cv::Mat w(3, 1, CV_32F);
            for (int k = 0; k < 3; k++) w.at<float>(k) = world1[3 * i + k];
            L.mvpMapPoints[i]->worldPos = w;
        }
    }
    ORBmatcher matcher(ratio, check != 0);
    const int n = TIMED(matcher.SearchByProjection(C, L, th));
    for (int i = 0; i < n2; i++) t2q[i] = C.mvpMapPoints[i] == &old ? -2 : index_of(L.mvpMapPoints, C.mvpMapPoints[i]);
    return n;
}

// rule 1 behind a projection: SearchByProjection(Frame& F1, Frame& F2, int windowSize, vector<MapPoint*>&) with F2.mTcw = identity.  state1 as in
// ref_window_search, world1 = the map points' positions (X, Y, 1); claimed2 = F2 features that hold a map point on entry.  t2q[n2]: -2 = held on entry.
int ref_search_by_projection_two_frames(const void* bounds, float ratio, const float* cam, const void* kps_un1, const uint8_t* desc1, const uint8_t* state1,
                                        const float* world1, int n1, const void* kps_un2, const uint8_t* desc2, const int32_t* cell_off2,
                                        const int32_t* cell_feat2, int n2, const uint8_t* claimed2, int windowSize, int32_t* t2q) {
    Frame F1, F2;
    Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
    F1.mvKeysUn = kp_vec(kps_un1, n1); F1.mDescriptors = desc_mat(desc1, n1);
    F2.mvKeysUn = kp_vec(kps_un2, n2); F2.mDescriptors = desc_mat(desc2, n2);
    fill_grid(F2.grid, bounds, cell_off2, cell_feat2, n2);
    F2.mTcw = pose_44(1.0f);
    MapPoint old;
    F2.mvpMapPoints.assign(n2, (MapPoint*)0);
    for (int i = 0; i < n2; i++) if (claimed2 && claimed2[i]) F2.mvpMapPoints[i] = &old;
    std::deque<MapPoint> pool;
    F1.mvpMapPoints = map_points(pool, state1, n1);
    for (int i = 0; i < n1; i++) if (F1.mvpMapPoints[i]) {
        cv::Mat w(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) w.at<float>(k) = world1[3 * i + k];
        F1.mvpMapPoints[i]->worldPos = w;
    }
    std::vector<MapPoint*> m2;
    ORBmatcher matcher(ratio, true);
    const int n = TIMED(matcher.SearchByProjection(F1, F2, windowSize, m2));
    for (int i = 0; i < n2; i++) t2q[i] = m2[i] == &old ? -2 : index_of(F1.mvpMapPoints, m2[i]);
    return n;
}

// rule 2 from a key frame: SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, float th, int ORBdist) with
// CurrentFrame.mTcw = identity.  kf_state: 0 none / 1 good / 2 bad / 3 good but in sAlreadyFound; world / mindist per key-frame feature.
int ref_search_by_projection_keyframe(const void* bounds, int check, float th, int orbdist, const float* cam, const float* scale_factors, int nlevels,
                                      const void* kps_un2, const uint8_t* desc2, const int32_t* cell_off2, const int32_t* cell_feat2, int n2,
                                      const uint8_t* claimed2, const void* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_state, const float* world,
                                      const float* mindist, int nKF, int32_t* t2q) {
    Frame C;
    Frame::fx = cam[0]; Frame::fy = cam[1]; Frame::cx = cam[2]; Frame::cy = cam[3];
    GridView::Bounds bb;
    memcpy(&bb, bounds, sizeof(bb));
    Frame::mnMinX = bb.min_x; Frame::mnMaxX = bb.max_x; Frame::mnMinY = bb.min_y; Frame::mnMaxY = bb.max_y;
    C.mvKeysUn = kp_vec(kps_un2, n2); C.mDescriptors = desc_mat(desc2, n2);
    C.mvScaleFactors.assign(scale_factors, scale_factors + nlevels); C.mnScaleLevels = nlevels;
    fill_grid(C.grid, bounds, cell_off2, cell_feat2, n2);
    C.mTcw = pose_44(1.0f);
    MapPoint old;
    C.mvpMapPoints.assign(n2, (MapPoint*)0);
    for (int i = 0; i < n2; i++) if (claimed2 && claimed2[i]) C.mvpMapPoints[i] = &old;
    KeyFrame kf;
    kf.keysUn = kp_vec(kf_kps, nKF);
    std::deque<MapPoint> pool;
    std::vector<uint8_t> st(kf_state, kf_state + nKF);
    for (auto& v : st) if (v == 3) v = 1;
    kf.mapPoints = map_points(pool, st.data(), nKF);
    std::set<MapPoint*> found;
    for (int i = 0; i < nKF; i++) if (kf.mapPoints[i]) {
        MapPoint& m = *kf.mapPoints[i];
        cv::Mat w(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) w.at<float>(k) = world[3 * i + k];
        m.worldPos = w; m.minDistance = mindist[i]; m.maxDistance = 1e9f;
        m.descriptor = desc_row(kf_desc + (size_t)i * 32);
        if (kf_state[i] == 3) found.insert(&m);
    }
    ORBmatcher matcher(0.75f, check != 0);
    const int n = TIMED(matcher.SearchByProjection(C, &kf, found, th, orbdist));
    for (int i = 0; i < n2; i++) t2q[i] = C.mvpMapPoints[i] == &old ? -2 : index_of(kf.mapPoints, C.mvpMapPoints[i]);
    return n;
}

// SearchByBoW(KeyFrame*, Frame&, ...).  kf_state: 0 / 1 / 2 as above.  t2q[nF] = the key-frame feature whose map point vpMapPointMatches[iF] is.
int ref_search_by_bow(float ratio, int check, const uint32_t* kf_node, const int32_t* kf_off, const uint32_t* kf_feat, int kf_nnodes, const uint8_t* kf_desc,
                      const float* kf_angle, const uint8_t* kf_state, int nKF, const uint32_t* f_node, const int32_t* f_off, const uint32_t* f_feat,
                      int f_nnodes, const uint8_t* f_desc, const float* f_angle, int nF, int32_t* t2q) {
    KeyFrame kf;
    kf.featVec = fv_from_csr(kf_node, kf_off, kf_feat, kf_nnodes);
    kf.descriptors = desc_mat(kf_desc, nKF);
    kf.keysUn.resize(nKF);
    for (int i = 0; i < nKF; i++) kf.keysUn[i].angle = kf_angle[i];
    std::deque<MapPoint> pool;
    kf.mapPoints = map_points(pool, kf_state, nKF);
    Frame F;
    F.mFeatVec = fv_from_csr(f_node, f_off, f_feat, f_nnodes);
    F.mDescriptors = desc_mat(f_desc, nF);
    F.mvKeys.resize(nF);
    for (int i = 0; i < nF; i++) F.mvKeys[i].angle = f_angle[i];
    F.mvKeysUn = F.mvKeys;
    F.mvpMapPoints.assign(nF, (MapPoint*)0);
    std::vector<MapPoint*> out;
    ORBmatcher matcher(ratio, check != 0);
    const int n = TIMED(matcher.SearchByBoW(&kf, F, out));
    for (int i = 0; i < nF; i++) t2q[i] = index_of(kf.mapPoints, out[i]);
    return n;
}

// SearchByBoW(KeyFrame*, KeyFrame*, ...).  q2t[n1] = the pKF2 feature whose map point vpMatches12[idx1] is.
int ref_search_by_bow_kf(float ratio, int check, const uint32_t* node1, const int32_t* off1, const uint32_t* feat1, int nnodes1, const uint8_t* desc1,
                         const float* angle1, const uint8_t* state1, int n1, const uint32_t* node2, const int32_t* off2, const uint32_t* feat2,
                         int nnodes2, const uint8_t* desc2, const float* angle2, const uint8_t* state2, int n2, int32_t* q2t) {
    KeyFrame k1, k2;
    std::deque<MapPoint> pool;
    k1.featVec = fv_from_csr(node1, off1, feat1, nnodes1); k1.descriptors = desc_mat(desc1, n1); k1.keysUn.resize(n1);
    k2.featVec = fv_from_csr(node2, off2, feat2, nnodes2); k2.descriptors = desc_mat(desc2, n2); k2.keysUn.resize(n2);
    for (int i = 0; i < n1; i++) k1.keysUn[i].angle = angle1[i];
    for (int i = 0; i < n2; i++) k2.keysUn[i].angle = angle2[i];
    k1.mapPoints = map_points(pool, state1, n1);
    k2.mapPoints = map_points(pool, state2, n2);
    std::vector<MapPoint*> out;
    ORBmatcher matcher(ratio, check != 0);
    const int n = TIMED(matcher.SearchByBoW(&k1, &k2, out));
    for (int i = 0; i < n1; i++) q2t[i] = index_of(k2.mapPoints, out[i]);
    return n;
}

// SearchForTriangulation.  has_mp: 0 / 1.  q2t[n1] = vMatches12 rebuilt from vMatchedPairs; also checks that the three output vectors agree.
int ref_search_for_triangulation(float ratio, int check, const float* F12, const float* sigma2, int nlevels, const uint32_t* node1, const int32_t* off1,
                                 const uint32_t* feat1, int nnodes1, const void* kps1, const uint8_t* desc1, const uint8_t* has_mp1, int n1,
                                 const uint32_t* node2, const int32_t* off2, const uint32_t* feat2, int nnodes2, const void* kps2, const uint8_t* desc2,
                                 const uint8_t* has_mp2, int n2, int32_t* q2t) {
    KeyFrame k1, k2;
    std::deque<MapPoint> pool;
    k1.featVec = fv_from_csr(node1, off1, feat1, nnodes1); k1.descriptors = desc_mat(desc1, n1); k1.keysUn = kp_vec(kps1, n1);
    k2.featVec = fv_from_csr(node2, off2, feat2, nnodes2); k2.descriptors = desc_mat(desc2, n2); k2.keysUn = kp_vec(kps2, n2);
    k1.mapPoints = map_points(pool, has_mp1, n1);
    k2.mapPoints = map_points(pool, has_mp2, n2);
    k2.levelSigma2.assign(sigma2, sigma2 + nlevels);
    cv::Mat F(3, 3, CV_32F);
    for (int i = 0; i < 9; i++) F.at<float>(i / 3, i % 3) = F12[i];
    std::vector<cv::KeyPoint> mk1, mk2;
    std::vector<std::pair<size_t, size_t> > pairs;
    ORBmatcher matcher(ratio, check != 0);
    const int n = TIMED(matcher.SearchForTriangulation(&k1, &k2, F, mk1, mk2, pairs));
    for (int i = 0; i < n1; i++) q2t[i] = -1;
    if ((int)pairs.size() != n || mk1.size() != pairs.size() || mk2.size() != pairs.size()) return -1000;
    for (size_t j = 0; j < pairs.size(); j++) {
        q2t[pairs[j].first] = (int)pairs[j].second;
        if (mk1[j].pt.x != k1.keysUn[pairs[j].first].pt.x || mk2[j].pt.y != k2.keysUn[pairs[j].second].pt.y) return -1001;
    }
    return n;
}

namespace {
// a key frame at the identity pose with camera `cam`, and map points at (X, Y, 1) looking straight at it
void identity_keyframe(KeyFrame& k, const void* bounds, const float* cam, const float* scale_factors, int nlevels, const void* kps, const uint8_t* desc,
                       const int32_t* off, const int32_t* feat, int n) {
    GridView::Bounds bb;
    memcpy(&bb, bounds, sizeof(bb));
    k.keysUn = kp_vec(kps, n); k.descriptors = desc_mat(desc, n);
    fill_grid(k.grid, bounds, off, feat, n);
    k.scaleFactors.assign(scale_factors, scale_factors + nlevels);
    k.fx = cam[0]; k.fy = cam[1]; k.cx = cam[2]; k.cy = cam[3];
    k.minX = bb.min_x; k.maxX = bb.max_x; k.minY = bb.min_y; k.maxY = bb.max_y;
    k.Rcw = pose_R(); k.tcw = pose_t(); k.Ow = pose_centre();
}
std::vector<MapPoint*> query_points(std::deque<MapPoint>& pool, const uint8_t* state, const float* world, const float* mindist, const uint8_t* desc, int nq) {
    std::vector<MapPoint*> q(nq, (MapPoint*)0);
    for (int i = 0; i < nq; i++) {
        if (!state[i]) continue;
        pool.emplace_back();
        MapPoint& m = pool.back();
        m.bad = state[i] == 2;
        cv::Mat w(3, 1, CV_32F), nrm(3, 1, CV_32F);
        double len = 0;
        for (int c = 0; c < 3; c++) { w.at<float>(c) = world[3 * i + c]; len += (double)world[3 * i + c] * world[3 * i + c]; }
        for (int c = 0; c < 3; c++) nrm.at<float>(c) = (float)(world[3 * i + c] / std::sqrt(len));
        m.worldPos = w; m.normal = nrm; m.minDistance = mindist[i]; m.maxDistance = 1e9f;
        m.descriptor = desc_row(desc + (size_t)i * 32);
        q[i] = &m;
    }
    return q;
}
}  // namespace

// SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) with Scw = identity.
// qstate: 1 good, 2 bad, 3 good but already in vpMatched (it is put into the first claimed slot).  t2q[nKF]: -2 = matched on entry.
int ref_search_by_projection_scw(const void* bounds, int th, const float* cam, const float* scale_factors, int nlevels, const void* kps, const uint8_t* desc,
                                 const int32_t* off, const int32_t* feat, int nKF, const uint8_t* claimed, const uint8_t* qstate, const float* world,
                                 const float* mindist, const uint8_t* qdesc, int nq, int32_t* t2q) {
    KeyFrame kf;
    identity_keyframe(kf, bounds, cam, scale_factors, nlevels, kps, desc, off, feat, nKF);
    std::deque<MapPoint> pool;
    std::vector<uint8_t> st(qstate, qstate + nq);
    for (auto& v : st) if (v == 3) v = 1;
    std::vector<MapPoint*> q = query_points(pool, st.data(), world, mindist, qdesc, nq);
    MapPoint old;
    std::vector<MapPoint*> matched(nKF, (MapPoint*)0);
    for (int i = 0; i < nKF; i++) if (claimed[i]) matched[i] = &old;
    int slot = 0;
    for (int i = 0; i < nq; i++) if (qstate[i] == 3) { while (slot < nKF && !claimed[slot]) slot++; if (slot < nKF) matched[slot++] = q[i]; }
    std::vector<MapPoint*> before = matched;
    cv::Mat Scw = pose_44(g_scale);
    ORBmatcher matcher(0.75f, true);
    const int n = TIMED(matcher.SearchByProjection(&kf, Scw, q, matched, th));
    for (int i = 0; i < nKF; i++) t2q[i] = before[i] ? -2 : index_of(q, matched[i]);
    return n;
}

// Fuse.  which = 0: Fuse(KeyFrame*, vector<MapPoint*>&, float th) with the key frame at the identity pose; which = 1: Fuse(KeyFrame*, cv::Mat Scw,
// const vector<MapPoint*>&, float th) with Scw = identity.  kf_state: map points the key frame already holds (0 none / 1 good / 2 bad);
// qstate: 0 NULL (variant 0 only) / 1 good / 2 bad.  log[]: the feature index of every fused point in query order (the reference asks the key frame
// for exactly that index once per fused point); returns nFused.
int ref_fuse(int which, const void* bounds, float th, const float* cam, const float* scale_factors, int nlevels, const void* kps, const uint8_t* desc,
             const int32_t* off, const int32_t* feat, int nKF, const uint8_t* kf_state, const uint8_t* qstate, const float* world, const float* mindist,
             const uint8_t* qdesc, int nq, int32_t* log, int* nlog) {
    KeyFrame kf;
    identity_keyframe(kf, bounds, cam, scale_factors, nlevels, kps, desc, off, feat, nKF);
    std::deque<MapPoint> pool;
    kf.mapPoints = map_points(pool, kf_state, nKF);
    std::vector<MapPoint*> q = query_points(pool, qstate, world, mindist, qdesc, nq);
    ORBmatcher matcher(0.75f, true);
    int n;
    if (which == 0) n = TIMED(matcher.Fuse(&kf, q, th));
    else {
        cv::Mat Scw = pose_44(g_scale);
        std::vector<MapPoint*> q2;
        for (MapPoint* p : q) if (p) q2.push_back(p);               // this overload does not accept NULL entries
        n = TIMED(matcher.Fuse(&kf, Scw, q2, th));
    }
    *nlog = (int)kf.getMapPointLog.size();
    for (size_t i = 0; i < kf.getMapPointLog.size(); i++) log[i] = kf.getMapPointLog[i];
    return n;
}

// SearchBySim3 with identity poses and the identity similarity (s12 = 1, R12 = I, t12 = 0): every map point (X, Y, 1) is seen at
// (fx*X + cx, fy*Y + cy) by both key frames through the reference's own projection code.  state: 0 = none, 1 = good, 2 = bad map
// point; world / mindist per feature (its map point).  match12[n1] = the pKF2 feature whose map point vpMatches12[i1] is (-1 none).
int ref_search_by_sim3(const void* bounds, float th, const float* cam, const float* scale_factors, int nlevels,
                       const void* kps1, const uint8_t* desc1, const int32_t* cell_off1, const int32_t* cell_feat1, const uint8_t* state1,
                       const float* world1, const float* mindist1, int n1,
                       const void* kps2, const uint8_t* desc2, const int32_t* cell_off2, const int32_t* cell_feat2, const uint8_t* state2,
                       const float* world2, const float* mindist2, int n2, int32_t* match12) {
    std::deque<MapPoint> pool;
    GridView::Bounds bb;
    memcpy(&bb, bounds, sizeof(bb));
    auto make = [&](KeyFrame& k, const void* kps, const uint8_t* desc, const int32_t* off, const int32_t* feat, const uint8_t* state, const float* world,
                    const float* mind, int n) {
        k.keysUn = kp_vec(kps, n); k.descriptors = desc_mat(desc, n);
        fill_grid(k.grid, bounds, off, feat, n);
        k.scaleFactors.assign(scale_factors, scale_factors + nlevels);
        k.fx = cam[0]; k.fy = cam[1]; k.cx = cam[2]; k.cy = cam[3];
        k.minX = bb.min_x; k.maxX = bb.max_x; k.minY = bb.min_y; k.maxY = bb.max_y;
        k.Rcw = pose_R(); k.tcw = pose_t();
        k.mapPoints = map_points(pool, state, n);
        for (int i = 0; i < n; i++) if (k.mapPoints[i]) {
            MapPoint& m = *k.mapPoints[i];
            cv::Mat w(3, 1, CV_32F);
            for (int c = 0; c < 3; c++) w.at<float>(c) = world[3 * i + c];
            m.worldPos = w; m.minDistance = mind[i]; m.maxDistance = 1e9f;
            m.descriptor = desc_row(desc + (size_t)i * 32);
        }
    };
    KeyFrame k1, k2;
    make(k1, kps1, desc1, cell_off1, cell_feat1, state1, world1, mindist1, n1);
    make(k2, kps2, desc2, cell_off2, cell_feat2, state2, world2, mindist2, n2);
    cv::Mat R12 = pose_R(g_simRt), t12 = pose_t(g_simRt);
    std::vector<MapPoint*> m12(n1, (MapPoint*)0);
    ORBmatcher matcher(0.75f, true);
    const float s12 = g_sim_scale;
    const int n = TIMED(matcher.SearchBySim3(&k1, &k2, m12, s12, R12, t12, th));
    for (int i = 0; i < n1; i++) match12[i] = index_of(k2.mapPoints, m12[i]);
    return n;
}

double ref_last_call_ms() { return g_last_call_ms; }

}  // extern "C"
