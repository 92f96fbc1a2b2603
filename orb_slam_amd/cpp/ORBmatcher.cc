// ORB_SLAM::ORBmatcher — the thirteen searches of reference include/ORBmatcher.h:48-88 on the MI355X (drop-in for src/ORBmatcher.cc).
//
// Every search of the reference is "for each query in order: collect candidates (grid window / vocabulary node), scan them for the
// best / second-best Hamming distance skipping what earlier queries took, accept by a rule, afterwards drop the matches outside the
// three dominant rotation bins".  Here that whole inner part is ONE kernel launch per search (include/orbs.h: the scanned frame is
// staged in LDS, 256 queries scan speculatively, a wave commits them in query order); this file is the part that cannot move:
//
//   1. walk the caller's objects ONCE and turn them into flat arrays — which queries take part (NULL / bad / already found /
//      not in view / outside the image / outside the scale range / seen too obliquely), where their window lies (the reference's
//      own cv::Mat expressions for the projection, so the floats are the reference's), radius and level range, descriptor, angle;
//      the scanned frame's key points, descriptors, 64 x 48 grid (or vocabulary-node lists) and which features are taken on entry;
//   2. one pinned block up, the launch, one pinned block down (a private stream per host thread: Tracking, LocalMapping and
//      LoopClosing call in concurrently, as they call the reference);
//   3. write the result where the reference writes it (F.mvpMapPoints, vpMatched, vnMatches12, Replace / AddObservation ...).
//
// Each function cites the reference lines it replaces.  tests/test_gpu_orbmatcher_dropin.py runs this file (compiled against the
// stand-in Frame / KeyFrame / MapPoint of oracle/matcherstub, recipe oracle/Makefile) and the reference's own src/ORBmatcher.cc
// through the same harness on the same problems and compares every output.  There is no CPU fallback: without a usable GPU the
// searches throw std::runtime_error.
#include "ORBmatcher.h"

#ifndef ORBMATCHER_HAS_SLAM_TYPES
#error "ORBmatcher.cc needs ORB_SLAM's MapPoint.h, KeyFrame.h and Frame.h on the include path (see INTEGRATION.md section 2)"
#endif

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>

#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "orbf.h"
#include "orbs.h"

#ifndef ORBMATCHER_ACCESS_HEADER
#define ORBMATCHER_ACCESS_HEADER "ORBmatcherAccess.h"
#endif
#include ORBMATCHER_ACCESS_HEADER

namespace ORB_SLAM {

const int ORBmatcher::TH_HIGH;
const int ORBmatcher::TH_LOW;
const int ORBmatcher::HISTO_LENGTH;

namespace {

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must have OpenCV 2.4's 28-byte layout");

void require(int rc, const char* what) {
    if (rc == ORBX_OK) return;
    throw std::runtime_error(std::string("ORB_SLAM::ORBmatcher: ") + what + " failed with status " + std::to_string(rc) +
                             (rc == ORBX_ERR_CAPACITY ? " (frame beyond the search capacity, include/orbs.h)" : " (no usable MI355X / HIP runtime)"));
}

// One pinned host block mirrored by one device block: a search lays its inputs, then its outputs, into the host block by bumping a
// cursor; the same offsets address the device block.  [0, inputs) goes up, [inputs, cursor) comes down, one synchronisation.
class Workspace {
public:
    explicit Workspace(int device) : device_(device) { require(orbx_stream_create(device, &stream_), "orbx_stream_create"); }
    ~Workspace() {
        (void)orbx_host_free(device_, host_);
        (void)orbx_device_free(device_, dev_);
        (void)orbx_stream_destroy(device_, stream_);
    }
    Workspace(const Workspace&) = delete;
    Workspace& operator=(const Workspace&) = delete;

    void begin(size_t bytes) {
        if (bytes > cap_) {
            (void)orbx_host_free(device_, host_);
            (void)orbx_device_free(device_, dev_);
            host_ = dev_ = nullptr; cap_ = 0;
            const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 20);
            require(orbx_host_alloc(device_, want, (void**)&host_), "orbx_host_alloc");
            require(orbx_device_alloc(device_, want, (void**)&dev_), "orbx_device_alloc");
            cap_ = want;
        }
        cursor_ = inputs_ = 0;
    }
    template <class T> struct Span { T* h; T* d; };
    template <class T> Span<T> take(size_t count) {
        const size_t off = cursor_;
        cursor_ = (cursor_ + count * sizeof(T) + 255) & ~(size_t)255;
        if (cursor_ > cap_) throw std::logic_error("ORBmatcher workspace sized too small");
        return Span<T>{(T*)(host_ + off), (T*)(dev_ + off)};
    }
    template <class T> Span<T> put(const T* src, size_t count) {
        Span<T> s = take<T>(count);
        if (count) std::memcpy(s.h, src, count * sizeof(T));
        return s;
    }
    template <class T> Span<T> put(const std::vector<T>& v) { return put(v.data(), v.size()); }
    template <class T> Span<T> put1(T v) { return put(&v, 1); }
    void inputs_done() { inputs_ = cursor_; require(orbx_device_upload_async(dev_, host_, inputs_, stream_), "upload"); }
    void fetch() {
        require(orbx_device_download_async(host_ + inputs_, dev_ + inputs_, cursor_ - inputs_, stream_), "download");
        require(orbx_stream_synchronize(device_, stream_), "the search kernel");
    }
    void* stream() const { return stream_; }

private:
    int device_;
    void* stream_ = nullptr;
    unsigned char *host_ = nullptr, *dev_ = nullptr;
    size_t cap_ = 0, cursor_ = 0, inputs_ = 0;
};

Workspace& workspace(int device) {
    thread_local std::map<int, std::unique_ptr<Workspace> > per_thread;
    std::unique_ptr<Workspace>& w = per_thread[device];
    if (!w) w.reset(new Workspace(device));
    return *w;
}

size_t workspace_bytes(int cap, int qcap) { return (size_t)cap * 96 + (size_t)qcap * 160 + (ORBF_GRID_CELLS + 1) * 4 + 64 * 256; }

// ---- the scanned ("train") side of a search -----------------------------------------------------------------------------------
struct Scanned {
    const cv::KeyPoint* kps = nullptr;       // undistorted key points (position, octave, angle)
    cv::Mat desc;                            // n x 32
    int n = 0;
    orbf_bounds bounds;                      // grid searches only
    std::vector<int32_t> cell_off, cell_feat;
};

void pack_rows(unsigned char* dst, const cv::Mat& desc, int n) {
    for (int i = 0; i < n; i++) std::memcpy(dst + (size_t)i * 32, desc.ptr<unsigned char>(i), 32);
}

// ---- the query side: one slot per reference loop iteration, invalid slots skipped by the kernel ---------------------------------
struct Queries {
    int n;
    std::vector<float> xyr, angle;
    std::vector<int32_t> lev;
    std::vector<unsigned char> desc, valid;
    explicit Queries(size_t n_) : n((int)n_), xyr(3 * n_, 0.f), angle(n_, 0.f), lev(2 * n_, 0), desc(32 * n_, 0), valid(n_, 0) {}
    void set(size_t i, float x, float y, float r, int minLevel, int maxLevel, const cv::Mat& d, float a = 0.f) {
        xyr[3 * i] = x; xyr[3 * i + 1] = y; xyr[3 * i + 2] = r;
        lev[2 * i] = minLevel; lev[2 * i + 1] = maxLevel;
        std::memcpy(&desc[32 * i], d.ptr<unsigned char>(0), 32);
        angle[i] = a;
        valid[i] = 1;
    }
};

struct Found {
    std::vector<int32_t> q2t, t2q;       // per query the feature it ends on, per feature the query that ends on it (-1 none)
    int n = 0;                           // the reference function's return value
};

struct Uploaded {                        // device views of a Scanned frame's grid form
    const orbx_keypoint* kps; const uint8_t* desc; const int32_t *cell_off, *cell_feat, *nt; const uint8_t* claimed;
};
Uploaded upload_grid(Workspace& w, const Scanned& T, const std::vector<unsigned char>& claimed) {
    Uploaded u;
    u.kps = w.put((const orbx_keypoint*)T.kps, (size_t)T.n).d;
    Workspace::Span<uint8_t> d = w.take<uint8_t>((size_t)T.n * 32);
    pack_rows(d.h, T.desc, T.n);
    u.desc = d.d;
    u.cell_off = w.put(T.cell_off).d;
    Workspace::Span<int32_t> cf = w.take<int32_t>((size_t)T.n);
    if (!T.cell_feat.empty()) std::memcpy(cf.h, T.cell_feat.data(), T.cell_feat.size() * sizeof(int32_t));
    u.cell_feat = cf.d;
    u.nt = w.put1<int32_t>(T.n).d;
    u.claimed = claimed.empty() ? nullptr : w.put(claimed).d;
    return u;
}

// One grid-window search (include/orbs.h: orbs_window_search_batch_device, one problem)
Found grid_search(int device, int rule, int th, float ratio, bool check, const Scanned& T, const std::vector<unsigned char>& claimed, const Queries& Q) {
    Found f;
    f.q2t.assign(Q.n, -1); f.t2q.assign(T.n, -1);
    if (T.n == 0 || Q.n == 0) return f;
    if ((int)T.cell_off.size() != ORBF_GRID_CELLS + 1) throw std::logic_error("ORBmatcher: the scanned frame has no 64 x 48 grid");
    Workspace& w = workspace(device);
    w.begin(workspace_bytes(T.n, Q.n));
    const Uploaded u = upload_grid(w, T, claimed);
    const float* qxyr = w.put(Q.xyr).d;
    const int32_t* qlev = w.put(Q.lev).d;
    const uint8_t* qdesc = w.put(Q.desc).d;
    const float* qangle = w.put(Q.angle).d;
    const uint8_t* qvalid = w.put(Q.valid).d;
    const int32_t* nq = w.put1<int32_t>(Q.n).d;
    w.inputs_done();
    Workspace::Span<int32_t> q2t = w.take<int32_t>(Q.n), t2q = w.take<int32_t>(T.n), nm = w.take<int32_t>(1);
    orbs_params prm;
    prm.rule = rule; prm.th = th; prm.ratio = ratio; prm.check_orientation = check ? 1 : 0;
    require(orbs_window_search_batch_device(&T.bounds, &prm, u.kps, u.desc, u.cell_off, u.cell_feat, u.nt, T.n, u.claimed, qxyr, qlev, qdesc, qangle, qvalid,
                                            nq, Q.n, 1, q2t.d, t2q.d, nullptr, nullptr, nm.d, w.stream()),
            "orbs_window_search_batch_device");
    w.fetch();
    f.q2t.assign(q2t.h, q2t.h + Q.n); f.t2q.assign(t2q.h, t2q.h + T.n); f.n = nm.h[0];
    return f;
}

// ---- vocabulary-node searches: the merge walk over two FeatureVectors as data (src/ORBmatcher.cc:171-260, :738-819, :886-972) -----
// list      = the scanned frame's features of the nodes BOTH vectors hold, node after node, in the vectors' own order;
// qslot[q]  = the query frame's feature at query position q (positions run through the same nodes in the same order);
// qrange    = for each position the run of `list` its node owns.
struct NodeWalk {
    std::vector<int32_t> list, qslot, qrange;
};
NodeWalk walk_nodes(const DBoW2::FeatureVector& fvQ, const DBoW2::FeatureVector& fvT) {
    NodeWalk nw;
    DBoW2::FeatureVector::const_iterator q = fvQ.begin(), t = fvT.begin();
    while (q != fvQ.end() && t != fvT.end()) {
        if (q->first < t->first) { q = fvQ.lower_bound(t->first); continue; }
        if (t->first < q->first) { t = fvT.lower_bound(q->first); continue; }
        const int32_t r0 = (int32_t)nw.list.size();
        for (size_t j = 0; j < t->second.size(); j++) nw.list.push_back((int32_t)t->second[j]);
        const int32_t r1 = (int32_t)nw.list.size();
        for (size_t j = 0; j < q->second.size(); j++) {
            nw.qslot.push_back((int32_t)q->second[j]);
            nw.qrange.push_back(r0); nw.qrange.push_back(r1);
        }
        ++q; ++t;
    }
    return nw;
}

struct ListProblem {                         // what SearchByBoW (both) and SearchForTriangulation share
    const cv::KeyPoint* kpsT; cv::Mat descT; int nT;                 // scanned key frame / frame, by feature index
    const cv::KeyPoint* kpsQ; cv::Mat descQ; int nQ;                 // query key frame, by feature index
    std::vector<unsigned char> claimedT, validQ;                     // by feature index
};

// One list search: rule BOW (orbs_list_search_batch_device) or, with F12 != NULL, TRIANGULATION (orbs_triangulation_search_batch_device).
// Returns per QUERY FEATURE the scanned feature it ends on and per scanned feature the query feature.
Found list_search(int device, int rule, int th, float ratio, bool check, const ListProblem& P, const NodeWalk& nw, const float* F12 = nullptr,
                  const std::vector<float>* sigma2 = nullptr) {
    Found f;
    f.q2t.assign(P.nQ, -1); f.t2q.assign(P.nT, -1);
    const int npos = (int)nw.qslot.size();
    if (P.nT == 0 || P.nQ == 0 || npos == 0 || nw.list.empty()) return f;
    const int cap = P.nT, qcap = std::max(P.nQ, npos);
    // a FeatureVector holds every feature of its frame at most once (DBoW2 FeatureVector::addFeature): anything else is not one of this frame
    if ((int)nw.list.size() > cap) throw std::logic_error("ORBmatcher: the scanned frame's FeatureVector lists more features than the frame has");
    for (size_t j = 0; j < nw.list.size(); j++)
        if ((unsigned)nw.list[j] >= (unsigned)P.nT) throw std::logic_error("ORBmatcher: a FeatureVector entry of the scanned frame is not one of its features");
    for (size_t j = 0; j < nw.qslot.size(); j++)
        if ((unsigned)nw.qslot[j] >= (unsigned)P.nQ) throw std::logic_error("ORBmatcher: a FeatureVector entry of the query key frame is not one of its features");
    Workspace& w = workspace(device);
    w.begin(workspace_bytes(cap, qcap) + (size_t)P.nQ * 64);
    const orbx_keypoint* kpsT = w.put((const orbx_keypoint*)P.kpsT, (size_t)P.nT).d;
    Workspace::Span<uint8_t> dT = w.take<uint8_t>((size_t)P.nT * 32);
    pack_rows(dT.h, P.descT, P.nT);
    Workspace::Span<int32_t> list = w.take<int32_t>((size_t)cap);
    std::memcpy(list.h, nw.list.data(), nw.list.size() * sizeof(int32_t));
    const int32_t* nlist = w.put1<int32_t>((int32_t)nw.list.size()).d;
    const int32_t* nt = w.put1<int32_t>(P.nT).d;
    const uint8_t* claimed = w.put(P.claimedT).d;
    const int32_t* qrange = w.put(nw.qrange).d;
    const int32_t* qslot = w.put(nw.qslot).d;
    Workspace::Span<uint8_t> dQ = w.take<uint8_t>((size_t)qcap * 32);
    pack_rows(dQ.h, P.descQ, P.nQ);
    Workspace::Span<uint8_t> vQ = w.take<uint8_t>((size_t)qcap);
    std::memcpy(vQ.h, P.validQ.data(), (size_t)P.nQ);
    Workspace::Span<float> aQ = w.take<float>((size_t)qcap);
    for (int i = 0; i < P.nQ; i++) aQ.h[i] = P.kpsQ[i].angle;
    Workspace::Span<orbx_keypoint> kQ = w.take<orbx_keypoint>((size_t)qcap);
    std::memcpy(kQ.h, P.kpsQ, (size_t)P.nQ * sizeof(orbx_keypoint));
    const int32_t* nq = w.put1<int32_t>(npos).d;
    const float* dF = F12 ? w.put(F12, 9).d : nullptr;
    w.inputs_done();
    Workspace::Span<int32_t> q2t = w.take<int32_t>(qcap), t2q = w.take<int32_t>(cap), nm = w.take<int32_t>(1);
    orbs_params prm;
    prm.rule = rule; prm.th = th; prm.ratio = ratio; prm.check_orientation = check ? 1 : 0;
    if (F12)
        require(orbs_triangulation_search_batch_device(&prm, dF, sigma2->data(), (int)sigma2->size(), kpsT, dT.d, list.d, nlist, nt, cap, claimed, qrange, qslot,
                                                       kQ.d, dQ.d, vQ.d, nq, qcap, 1, q2t.d, t2q.d, nullptr, nullptr, nm.d, w.stream()),
                "orbs_triangulation_search_batch_device");
    else
        require(orbs_list_search_batch_device(&prm, kpsT, dT.d, list.d, nlist, nt, cap, claimed, qrange, qslot, dQ.d, aQ.d, vQ.d, nq, qcap, 1, q2t.d, t2q.d,
                                              nullptr, nullptr, nm.d, w.stream()),
                "orbs_list_search_batch_device");
    w.fetch();
    for (int pos = 0; pos < npos; pos++) if (q2t.h[pos] >= 0) f.q2t[nw.qslot[pos]] = q2t.h[pos];
    for (int t = 0; t < P.nT; t++) if (t2q.h[t] >= 0) f.t2q[t] = nw.qslot[t2q.h[t]];
    f.n = nm.h[0];
    return f;
}

// ---- projection pieces (each in the float form the reference function it serves uses) ------------------------------------------------
struct Pinhole { float fx, fy, cx, cy; };

// `u = fx*xc*invzc+cx` with `invzc = 1.0/z` (src/ORBmatcher.cc:548-553, :1531-1536, :1654-1659)
inline void project_scaled(const cv::Mat& p3Dc, const Pinhole& c, float& u, float& v) {
    const float xc = p3Dc.at<float>(0);
    const float yc = p3Dc.at<float>(1);
    const float invzc = 1.0 / p3Dc.at<float>(2);
    u = c.fx * xc * invzc + c.cx;
    v = c.fy * yc * invzc + c.cy;
}
// `x = X*invz; u = fx*x+cx` (:329-334, :1048-1053, :1188-1193, :1340-1345, :1418-1423)
inline void project_normalised(const cv::Mat& p3Dc, const Pinhole& c, float& u, float& v) {
    const float invz = 1.0 / p3Dc.at<float>(2);
    const float x = p3Dc.at<float>(0) * invz;
    const float y = p3Dc.at<float>(1) * invz;
    u = c.fx * x + c.cx;
    v = c.fy * y + c.cy;
}
// `lower_bound(factors, ratio)` clipped to the last level (:359-360, :1073-1074, :1213-1214, :1360-1361, :1438-1439, :1672-1673)
inline int predicted_level(const std::vector<float>& factors, float ratio, int nMaxLevel) {
    return std::min((int)(std::lower_bound(factors.begin(), factors.end(), ratio) - factors.begin()), nMaxLevel);
}

// A map point seen from a key frame with pose (Rcw, tcw, Ow): in front, inside the image, inside its scale-invariance range and
// less than 60 degrees off its mean viewing direction; then window centre, predicted level and radius th * factor[level]
// (the common body of :313-363, :1040-1077, :1172-1217)
struct KeyFrameView {
    KeyFrame* pKF; cv::Mat Rcw, tcw, Ow; Pinhole cam; std::vector<float> factors; int nMaxLevel;
};
bool window_in_keyframe(const KeyFrameView& V, MapPoint* pMP, float th, float& u, float& v, int& level, float& radius) {
    cv::Mat p3Dw = pMP->GetWorldPos();
    cv::Mat p3Dc = V.Rcw * p3Dw + V.tcw;
    if (p3Dc.at<float>(2) < 0.0f) return false;
    project_normalised(p3Dc, V.cam, u, v);
    if (!V.pKF->IsInImage(u, v)) return false;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    cv::Mat PO = p3Dw - V.Ow;
    const float dist3D = cv::norm(PO);
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    cv::Mat Pn = pMP->GetNormal();
    if (PO.dot(Pn) < 0.5 * dist3D) return false;
    level = predicted_level(V.factors, dist3D / minDistance, V.nMaxLevel);
    radius = th * V.factors[level];
    return true;
}

KeyFrameView view_of(KeyFrame* pKF, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow) {
    KeyFrameView V;
    V.pKF = pKF; V.Rcw = Rcw; V.tcw = tcw; V.Ow = Ow;
    V.cam.fx = pKF->fx; V.cam.fy = pKF->fy; V.cam.cx = pKF->cx; V.cam.cy = pKF->cy;
    V.factors = pKF->GetScaleFactors();
    V.nMaxLevel = pKF->GetScaleLevels() - 1;
    return V;
}
// `Scw` -> rotation, translation, camera centre (:296-303, :1145-1150)
KeyFrameView view_through_similarity(KeyFrame* pKF, const cv::Mat& Scw) {
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    return view_of(pKF, Rcw, tcw, Ow);
}

struct KeyFrameData {                        // a key frame's copies (its accessors lock and copy: taken once per search)
    std::vector<cv::KeyPoint> keys;
    Scanned scan;
};
void scan_keyframe(KeyFrame* pKF, KeyFrameData& K) {
    K.keys = pKF->GetKeyPointsUn();
    K.scan.kps = K.keys.data();
    K.scan.n = (int)K.keys.size();
    K.scan.desc = pKF->GetDescriptors();
    orbm_access::GridOf(pKF, K.scan.bounds, K.scan.cell_off, K.scan.cell_feat);
}
void scan_frame(const Frame& F, Scanned& S) {
    S.kps = F.mvKeysUn.data();
    S.n = (int)F.mvKeysUn.size();
    S.desc = F.mDescriptors;
    orbm_access::GridOf(F, S.bounds, S.cell_off, S.cell_feat);
}
std::vector<unsigned char> held(const std::vector<MapPoint*>& v) {
    std::vector<unsigned char> c(v.size());
    for (size_t i = 0; i < v.size(); i++) c[i] = v[i] ? 1 : 0;
    return c;
}

}  // namespace

// ---- Tracking: the local map's points into the frame (replaces src/ORBmatcher.cc:48-125) ----------------------------------------------
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    const bool bFactor = th != 1.0;
    Queries Q(vpMapPoints.size());
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView || pMP->isBad()) continue;
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);      // the window grows with the viewing angle
        if (bFactor) r *= th;
        Q.set(iMP, pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, pMP->GetDescriptor());
    }
    Scanned T;
    scan_frame(F, T);
    // best / second with their levels; the ratio only rejects when both lie on one level (:113-121): ORBS_RULE_MAPPOINTS
    const Found f = grid_search(device_, ORBS_RULE_MAPPOINTS, TH_HIGH, mfNNratio, false, T, held(F.mvpMapPoints), Q);
    for (size_t idx = 0; idx < f.t2q.size(); idx++)
        if (f.t2q[idx] >= 0) F.mvpMapPoints[idx] = vpMapPoints[f.t2q[idx]];
    return f.n;
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }      // :127-133

// :136-153.  `dsqr < 3.84*sigma2` is a float-against-double comparison; orbs_epipolar_bound is the float it is equivalent to, the
// same one the triangulation kernel tests against.
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF2) {
    const float a = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);      // l = x1' F12
    const float b = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
    const float c = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < orbs_epipolar_bound(pKF2->GetSigma2(kp2.octave));
}

// ---- Relocalisation / loop detection: features of the same vocabulary node, key frame -> frame (replaces :155-281) -------------------------
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.mvpMapPoints.size(), static_cast<MapPoint*>(NULL));
    const DBoW2::FeatureVector vFeatVecKF = pKF->GetFeatureVector();
    const std::vector<cv::KeyPoint> keysKF = pKF->GetKeyPointsUn();

    ListProblem P;
    P.kpsT = F.mvKeys.data(); P.descT = F.mDescriptors; P.nT = (int)vpMapPointMatches.size();        // `F.mvKeys[bestIdxF].angle` :232
    P.kpsQ = keysKF.data(); P.descQ = pKF->GetDescriptors(); P.nQ = (int)vpMapPointsKF.size();
    P.claimedT.assign(P.nT, 0);
    P.validQ.resize(P.nQ);
    for (int i = 0; i < P.nQ; i++) P.validQ[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
    // accept `best <= TH_LOW && best < ratio*second` (:224-226), rotation bins hold the frame feature (:239): ORBS_RULE_BOW
    const Found f = list_search(device_, ORBS_RULE_BOW, TH_LOW, mfNNratio, mbCheckOrientation, P, walk_nodes(vFeatVecKF, F.mFeatVec));
    for (int idxF = 0; idxF < P.nT; idxF++)
        if (f.t2q[idxF] >= 0) vpMapPointMatches[idxF] = vpMapPointsKF[f.t2q[idxF]];
    return f.n;
}

// ---- Loop closing: candidate points through a similarity into the key frame (replaces :286-407) --------------------------------------------
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) {
    const KeyFrameView V = view_through_similarity(pKF, Scw);
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));

    Queries Q(vpPoints.size());
    for (size_t iMP = 0; iMP < vpPoints.size(); iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        float u, v, radius;
        int level;
        if (!window_in_keyframe(V, pMP, (float)th, u, v, level, radius)) continue;
        Q.set(iMP, u, v, radius, level - 1, level, pMP->GetDescriptor());
    }
    KeyFrameData K;
    scan_keyframe(pKF, K);
    // best only, `<= TH_LOW`, a match takes its feature (:386-396): ORBS_RULE_BEST without the rotation check
    const Found f = grid_search(device_, ORBS_RULE_BEST, TH_LOW, 0.f, false, K.scan, held(vpMatched), Q);
    for (size_t idx = 0; idx < f.t2q.size() && idx < vpMatched.size(); idx++)
        if (f.t2q[idx] >= 0) vpMatched[idx] = vpPoints[f.t2q[idx]];
    return f.n;
}

// ---- Frame 1's tracked points in a window of frame 2 (replaces :408-516) -------------------------------------------------------------------
int ORBmatcher::WindowSearch(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2, int minScaleLevel, int maxScaleLevel) {
    vpMapPointMatches2 = std::vector<MapPoint*>(F2.mvpMapPoints.size(), static_cast<MapPoint*>(NULL));
    const bool bMinLevel = minScaleLevel > 0;
    const bool bMaxLevel = maxScaleLevel < INT_MAX;

    Queries Q(F1.mvpMapPoints.size());
    for (size_t i1 = 0; i1 < F1.mvpMapPoints.size(); i1++) {
        MapPoint* pMP1 = F1.mvpMapPoints[i1];
        if (!pMP1 || pMP1->isBad()) continue;
        const cv::KeyPoint& kp1 = F1.mvKeysUn[i1];
        const int level1 = kp1.octave;
        if ((bMinLevel && level1 < minScaleLevel) || (bMaxLevel && level1 > maxScaleLevel)) continue;
        Q.set(i1, kp1.pt.x, kp1.pt.y, windowSize, level1, level1, F1.mDescriptors.row(i1), kp1.angle);
    }
    Scanned T;
    scan_frame(F2, T);
    // accept `best <= second*ratio && best <= TH_HIGH` (:476); the rotation bins are applied only with mbCheckOrientation (:492): ORBS_RULE_WINDOW
    const Found f = grid_search(device_, ORBS_RULE_WINDOW, TH_HIGH, mfNNratio, mbCheckOrientation, T, std::vector<unsigned char>(), Q);
    for (size_t i2 = 0; i2 < f.t2q.size() && i2 < vpMapPointMatches2.size(); i2++)
        if (f.t2q[i2] >= 0) vpMapPointMatches2[i2] = F1.mvpMapPoints[f.t2q[i2]];
    return f.n;
}

// ---- The same behind a guess of frame 2's pose (replaces :519-594) -------------------------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2) {
    vpMapPointMatches2 = F2.mvpMapPoints;
    const std::set<MapPoint*> spMapPointsAlreadyFound(vpMapPointMatches2.begin(), vpMapPointMatches2.end());
    const cv::Mat Rc2w = F2.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tc2w = F2.mTcw.rowRange(0, 3).col(3);
    const Pinhole cam = {F2.fx, F2.fy, F2.cx, F2.cy};

    Queries Q(F1.mvpMapPoints.size());
    for (size_t i1 = 0; i1 < F1.mvpMapPoints.size(); i1++) {
        MapPoint* pMP1 = F1.mvpMapPoints[i1];
        if (!pMP1 || pMP1->isBad() || spMapPointsAlreadyFound.count(pMP1)) continue;
        const int level1 = F1.mvKeysUn[i1].octave;
        cv::Mat x3Dw = pMP1->GetWorldPos();
        cv::Mat x3Dc2 = Rc2w * x3Dw + tc2w;
        float u2, v2;
        project_scaled(x3Dc2, cam, u2, v2);
        Q.set(i1, u2, v2, windowSize, level1, level1, F1.mDescriptors.row(i1));
    }
    Scanned T;
    scan_frame(F2, T);
    const Found f = grid_search(device_, ORBS_RULE_WINDOW, TH_HIGH, mfNNratio, false, T, held(F2.mvpMapPoints), Q);      // :585, no rotation check
    for (size_t i2 = 0; i2 < f.t2q.size() && i2 < vpMapPointMatches2.size(); i2++)
        if (f.t2q[i2] >= 0) vpMapPointMatches2[i2] = F1.mvpMapPoints[f.t2q[i2]];
    return f.n;
}

// ---- Map initialisation (replaces :596-713) ------------------------------------------------------------------------------------------------
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) {
    const size_t n1 = F1.mvKeysUn.size();
    Queries Q(n1);
    for (size_t i1 = 0; i1 < n1; i1++) {
        const cv::KeyPoint& kp1 = F1.mvKeysUn[i1];
        if (kp1.octave > 0) continue;                                        // only the finest level takes part (:619-621)
        Q.set(i1, vbPrevMatched[i1].x, vbPrevMatched[i1].y, windowSize, kp1.octave, kp1.octave, F1.mDescriptors.row(i1), kp1.angle);
    }
    Scanned T;
    scan_frame(F2, T);
    // a feature already matched is taken over by a later query with a strictly smaller distance (:640-641, :658-666): ORBS_RULE_INIT
    const Found f = grid_search(device_, ORBS_RULE_INIT, TH_LOW, mfNNratio, mbCheckOrientation, T, std::vector<unsigned char>(), Q);
    vnMatches12.assign(f.q2t.begin(), f.q2t.end());
    for (size_t i1 = 0; i1 < vnMatches12.size(); i1++)
        if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;      // :707-710
    return f.n;
}

// ---- Loop closing: two key frames, same vocabulary node, both features holding good map points (replaces :715-850) -----------------------------
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    const std::vector<cv::KeyPoint> vKeysUn1 = pKF1->GetKeyPointsUn(), vKeysUn2 = pKF2->GetKeyPointsUn();
    const DBoW2::FeatureVector vFeatVec1 = pKF1->GetFeatureVector(), vFeatVec2 = pKF2->GetFeatureVector();
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));

    ListProblem P;
    P.kpsT = vKeysUn2.data(); P.descT = pKF2->GetDescriptors(); P.nT = (int)vpMapPoints2.size();
    P.kpsQ = vKeysUn1.data(); P.descQ = pKF1->GetDescriptors(); P.nQ = (int)vpMapPoints1.size();
    P.claimedT.resize(P.nT); P.validQ.resize(P.nQ);
    for (int i = 0; i < P.nT; i++) P.claimedT[i] = !(vpMapPoints2[i] && !vpMapPoints2[i]->isBad());      // :768-773: not a candidate
    for (int i = 0; i < P.nQ; i++) P.validQ[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
    // `bestDist1 < TH_LOW` here, not `<=` (:791)
    const Found f = list_search(device_, ORBS_RULE_BOW, TH_LOW - 1, mfNNratio, mbCheckOrientation, P, walk_nodes(vFeatVec1, vFeatVec2));
    for (int idx1 = 0; idx1 < P.nQ; idx1++)
        if (f.q2t[idx1] >= 0) vpMatches12[idx1] = vpMapPoints2[f.q2t[idx1]];
    return f.n;
}

// ---- LocalMapping: untracked features of two key frames along the epipolar line (replaces :852-1014) -----------------------------------------
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<cv::KeyPoint>& vMatchedKeys1,
                                       std::vector<cv::KeyPoint>& vMatchedKeys2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const std::vector<cv::KeyPoint> vKeysUn1 = pKF1->GetKeyPointsUn(), vKeysUn2 = pKF2->GetKeyPointsUn();
    const DBoW2::FeatureVector vFeatVec1 = pKF1->GetFeatureVector(), vFeatVec2 = pKF2->GetFeatureVector();

    ListProblem P;
    P.kpsT = vKeysUn2.data(); P.descT = pKF2->GetDescriptors(); P.nT = (int)vKeysUn2.size();
    P.kpsQ = vKeysUn1.data(); P.descQ = pKF1->GetDescriptors(); P.nQ = (int)vKeysUn1.size();
    P.claimedT = held(vpMapPoints2);                                        // features with a map point are no candidates (:914-916)
    P.validQ.resize(P.nQ);
    for (int i = 0; i < P.nQ; i++) P.validQ[i] = vpMapPoints1[i] ? 0 : 1;    // ... and do not ask (:896-898)
    float F[9];
    for (int i = 0; i < 9; i++) F[i] = F12.at<float>(i / 3, i % 3);
    std::vector<float> sigma2 = orbm_access::LevelSigma2Of(pKF2);
    if (sigma2.size() > ORBS_MAX_LEVELS) require(ORBX_ERR_CAPACITY, "SearchForTriangulation (more pyramid levels than ORBS_MAX_LEVELS)");
    Found f;
    f.q2t.assign(P.nQ, -1);
    if (!sigma2.empty())
        // candidates with distance <= TH_LOW sorted by (distance, index), the first within 2 x the best on the epipolar line (:920-958)
        f = list_search(device_, ORBS_RULE_TRIANGULATION, TH_LOW, 0.f, mbCheckOrientation, P, walk_nodes(vFeatVec1, vFeatVec2), F, &sigma2);

    vMatchedKeys1.clear(); vMatchedKeys1.reserve(f.n);
    vMatchedKeys2.clear(); vMatchedKeys2.reserve(f.n);
    vMatchedPairs.clear(); vMatchedPairs.reserve(f.n);
    for (size_t i = 0; i < f.q2t.size(); i++) {
        if (f.q2t[i] < 0) continue;
        vMatchedKeys1.push_back(vKeysUn1[i]);
        vMatchedKeys2.push_back(vKeysUn2[f.q2t[i]]);
        vMatchedPairs.push_back(std::make_pair(i, (size_t)f.q2t[i]));
    }
    return f.n;
}

namespace {
// The part of both Fuse overloads that is a search: every candidate point scans its window on its own — nothing is taken, several
// points may end on one feature (ORBS_RULE_FREE, best <= TH_LOW, levels [predicted-1, predicted]).  Validity that the loop itself
// can change (isBad after a Replace, membership after an AddMapPoint) is left to the caller's in-order pass.
Found fuse_scan(int device, const KeyFrameView& V, const std::vector<MapPoint*>& points, float th) {
    Queries Q(points.size());
    for (size_t i = 0; i < points.size(); i++) {
        MapPoint* pMP = points[i];
        if (!pMP) continue;
        float u, v, radius;
        int level;
        if (!window_in_keyframe(V, pMP, th, u, v, level, radius)) continue;
        Q.set(i, u, v, radius, level - 1, level, pMP->GetDescriptor());
    }
    KeyFrameData K;
    scan_keyframe(V.pKF, K);
    return grid_search(device, ORBS_RULE_FREE, ORBS_TH_LOW, 0.f, false, K.scan, std::vector<unsigned char>(), Q);
}
}  // namespace

// ---- LocalMapping: fuse map points into a key frame (replaces :1016-1134) ----------------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame* pKF, std::vector<MapPoint*>& vpMapPoints, float th) {
    const Found f = fuse_scan(device_, view_of(pKF, pKF->GetRotation(), pKF->GetTranslation(), pKF->GetCameraCenter()), vpMapPoints, th);
    int nFused = 0;
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;       // as of NOW: an earlier iteration may have replaced it (:1037-1038)
        if (f.q2t[i] < 0) continue;
        const int bestIdx = f.q2t[i];
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);                      // a map point already there: keep that one (:1114-1126)
        if (pMPinKF) {
            if (!pMPinKF->isBad()) pMP->Replace(pMPinKF);
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ---- Loop closing: the same through a corrected similarity (replaces :1136-1265) ---------------------------------------------------------
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th) {
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const Found f = fuse_scan(device_, view_through_similarity(pKF, Scw), vpPoints, th);
    int nFused = 0;
    for (size_t iMP = 0; iMP < vpPoints.size(); iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        if (f.q2t[iMP] < 0) continue;
        const int bestIdx = f.q2t[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);                      // here the candidate wins (:1245-1257)
        if (pMPinKF) {
            if (!pMPinKF->isBad()) pMPinKF->Replace(pMP);
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ---- Loop closing: both key frames' points through [s12*R12|t12] and back, kept where the two directions agree (replaces :1267-1505) ----------
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12,
                             float th) {
    const Pinhole cam = {pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy};
    cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;

    const std::vector<float> vfScaleFactors1 = pKF1->GetScaleFactors(), vfScaleFactors2 = pKF2->GetScaleFactors();
    const int nMaxLevel1 = pKF1->GetScaleLevels() - 1, nMaxLevel2 = pKF2->GetScaleLevels() - 1;
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();

    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        MapPoint* pMP = vpMatches12[i];
        if (!pMP) continue;
        vbAlreadyMatched1[i] = true;
        const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
        if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
    }

    // one direction: the points of `from` (not matched yet, good) into `into`'s image; no viewing-angle test here, the distance is
    // the camera-frame norm (:1325-1363, :1403-1443)
    struct Direction {
        static void build(Queries& Q, const std::vector<MapPoint*>& points, const std::vector<bool>& done, const cv::Mat& Rw, const cv::Mat& tw, const cv::Mat& sR,
                          const cv::Mat& t, const Pinhole& cam, KeyFrame* into, const std::vector<float>& factors, int nMaxLevel, float th) {
            for (size_t i = 0; i < points.size(); i++) {
                MapPoint* pMP = points[i];
                if (!pMP || done[i] || pMP->isBad()) continue;
                cv::Mat p3Dw = pMP->GetWorldPos();
                cv::Mat p3Dfrom = Rw * p3Dw + tw;
                cv::Mat p3Dinto = sR * p3Dfrom + t;
                if (p3Dinto.at<float>(2) < 0.0) continue;
                float u, v;
                project_normalised(p3Dinto, cam, u, v);
                if (!into->IsInImage(u, v)) continue;
                const float maxDistance = pMP->GetMaxDistanceInvariance();
                const float minDistance = pMP->GetMinDistanceInvariance();
                const float dist3D = cv::norm(p3Dinto);
                if (dist3D < minDistance || dist3D > maxDistance) continue;
                const int level = predicted_level(factors, dist3D / minDistance, nMaxLevel);
                Q.set(i, u, v, th * factors[level], level - 1, level, pMP->GetDescriptor());
            }
        }
    };
    Queries Q1(N1), Q2(N2);
    Direction::build(Q1, vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, cam, pKF2, vfScaleFactors2, nMaxLevel2, th);
    Direction::build(Q2, vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12, cam, pKF1, vfScaleFactors1, nMaxLevel1, th);
    if (N1 == 0 || N2 == 0) return 0;

    // both scans (ORBS_RULE_FREE, best <= TH_HIGH :1394, :1474) and the agreement check (:1480-1502) in one submission: only the
    // agreed matches come back
    KeyFrameData K1, K2;
    scan_keyframe(pKF1, K1);
    scan_keyframe(pKF2, K2);
    Workspace& w = workspace(device_);
    w.begin(workspace_bytes(N1, N1) + workspace_bytes(N2, N2));
    const std::vector<unsigned char> none;
    const Uploaded u1 = upload_grid(w, K1.scan, none), u2 = upload_grid(w, K2.scan, none);
    struct DevQ { const float* xyr; const int32_t* lev; const uint8_t *desc, *valid; const int32_t* n; };
    auto upq = [&w](const Queries& Q) {
        DevQ d;
        d.xyr = w.put(Q.xyr).d; d.lev = w.put(Q.lev).d; d.desc = w.put(Q.desc).d; d.valid = w.put(Q.valid).d; d.n = w.put1<int32_t>(Q.n).d;
        return d;
    };
    const DevQ q1 = upq(Q1), q2 = upq(Q2);
    w.inputs_done();
    Workspace::Span<int32_t> out12 = w.take<int32_t>(N1), nFoundDev = w.take<int32_t>(1);
    Workspace::Span<int32_t> m12 = w.take<int32_t>(N1), m21 = w.take<int32_t>(N2), t2q1 = w.take<int32_t>(N1), t2q2 = w.take<int32_t>(N2), nm = w.take<int32_t>(2);
    orbs_params prm;
    prm.rule = ORBS_RULE_FREE; prm.th = TH_HIGH; prm.ratio = 0.f; prm.check_orientation = 0;
    require(orbs_window_search_batch_device(&K2.scan.bounds, &prm, u2.kps, u2.desc, u2.cell_off, u2.cell_feat, u2.nt, N2, nullptr, q1.xyr, q1.lev, q1.desc, nullptr,
                                            q1.valid, q1.n, N1, 1, m12.d, t2q2.d, nullptr, nullptr, nm.d, w.stream()),
            "orbs_window_search_batch_device (1 -> 2)");
    require(orbs_window_search_batch_device(&K1.scan.bounds, &prm, u1.kps, u1.desc, u1.cell_off, u1.cell_feat, u1.nt, N1, nullptr, q2.xyr, q2.lev, q2.desc, nullptr,
                                            q2.valid, q2.n, N2, 1, m21.d, t2q1.d, nullptr, nullptr, nm.d + 1, w.stream()),
            "orbs_window_search_batch_device (2 -> 1)");
    require(orbs_agreement_batch_device(m12.d, q1.n, N1, m21.d, q2.n, N2, 1, out12.d, nFoundDev.d, w.stream()), "orbs_agreement_batch_device");
    w.fetch();
    for (int i1 = 0; i1 < N1; i1++)
        if (out12.h[i1] >= 0) vpMatches12[i1] = vpMapPoints2[out12.h[i1]];
    return nFoundDev.h[0];
}

// ---- Tracking: the last frame's points into the current frame (replaces :1507-1619) --------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th) {
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const Pinhole cam = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy};

    Queries Q(LastFrame.mvpMapPoints.size());
    for (size_t i = 0; i < LastFrame.mvpMapPoints.size(); i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        float u, v;
        project_scaled(x3Dc, cam, u, v);
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX || v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const int nPredictedOctave = LastFrame.mvKeys[i].octave;           // the window follows the scale it was seen at
        Q.set(i, u, v, th * CurrentFrame.mvScaleFactors[nPredictedOctave], nPredictedOctave - 1, nPredictedOctave + 1, LastFrame.mDescriptors.row(i),
              LastFrame.mvKeysUn[i].angle);
    }
    Scanned T;
    scan_frame(CurrentFrame, T);
    const Found f = grid_search(device_, ORBS_RULE_BEST, TH_HIGH, 0.f, mbCheckOrientation, T, held(CurrentFrame.mvpMapPoints), Q);      // :1574
    for (size_t i2 = 0; i2 < f.t2q.size(); i2++)
        if (f.t2q[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[f.t2q[i2]];
    return f.n;
}

// ---- Relocalisation: a key frame's points into the current frame (replaces :1622-1746) -------------------------------------------------------
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, float th, int ORBdist) {
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const Pinhole cam = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy};
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();

    Queries Q(vpMPs.size());
    for (size_t i = 0; i < vpMPs.size(); i++) {
        MapPoint* pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        float u, v;
        project_scaled(x3Dc, cam, u, v);
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX || v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const float minDistance = pMP->GetMinDistanceInvariance();       // the level is predicted from the distance (:1666-1673)
        cv::Mat PO = x3Dw - Ow;
        const float dist3D = cv::norm(PO);
        const int level = predicted_level(CurrentFrame.mvScaleFactors, dist3D / minDistance, CurrentFrame.mnScaleLevels - 1);
        Q.set(i, u, v, th * CurrentFrame.mvScaleFactors[level], level - 1, level + 1, pMP->GetDescriptor(), pKF->GetKeyPointUn(i).angle);
    }
    Scanned T;
    scan_frame(CurrentFrame, T);
    const Found f = grid_search(device_, ORBS_RULE_BEST, ORBdist, 0.f, mbCheckOrientation, T, held(CurrentFrame.mvpMapPoints), Q);      // :1703
    for (size_t i2 = 0; i2 < f.t2q.size(); i2++)
        if (f.t2q[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[f.t2q[i2]];
    return f.n;
}

// :1748-1789 on bin sizes (orbs_three_maxima is the host form of what the kernels apply)
void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    std::vector<int32_t> sizes(L);
    for (int i = 0; i < L; i++) sizes[i] = (int32_t)histo[i].size();
    int32_t ind[3];
    orbs_three_maxima(sizes.data(), L, ind);
    ind1 = ind[0]; ind2 = ind[1]; ind3 = ind[2];
}

}  // namespace ORB_SLAM
