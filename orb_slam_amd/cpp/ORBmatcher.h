// ORB_SLAM::ORBmatcher on the MI355X C ABI — the reference's class surface (reference include/ORBmatcher.h:41-103): the
// constructor, DescriptorDistance, the thirteen Frame / KeyFrame / MapPoint searches with the reference's signatures, the three
// thresholds.  The searches are DEFINED in ORBmatcher.cc: each does on the host what only the host can do — walk the MapPoint /
// Frame / KeyFrame objects, project, test visibility, pick radius and level range (the reference's own cv::Mat expressions, so the
// floats are the reference's) — and hands the scan (best / second-best over the grid window or the vocabulary node, the in-order
// "already matched" masking, the accept rule, the rotation histogram) to the kernels behind include/orbs.h, then writes the
// result back where the reference writes it.
//
// Two ways to use this header:
//   * inside ORB_SLAM (drop-in for include/ORBmatcher.h + src/ORBmatcher.cc): the SLAM headers are on the include path and are
//     pulled in below exactly as the reference header does; compile ORBmatcher.cc into the project, link liborbx.so;
//   * stand-alone (examples, the dense / candidate-list forms MatchTop2*): without MapPoint.h / KeyFrame.h / Frame.h on the include
//     path the three classes are only forward-declared; the thirteen searches are then declared but not linkable.
#pragma once
#include <climits>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

#include "cvcompat.h"
#include "orbx.h"

#if defined(__has_include)
#if __has_include("MapPoint.h") && __has_include("KeyFrame.h") && __has_include("Frame.h")
#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
#define ORBMATCHER_HAS_SLAM_TYPES 1
#endif
#endif

namespace ORB_SLAM {

class MapPoint;
class KeyFrame;
class Frame;

class ORBmatcher {
public:
    // reference include/ORBmatcher.h:41; `device` = the GPU the searches run on (one process per GPU: 0)
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}

    // Computes the Hamming distance between two ORB descriptors (reference :44, src/ORBmatcher.cc:1794-1810)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbm_hamming256(a.ptr<unsigned char>(), b.ptr<unsigned char>()); }

    // ---- the thirteen searches, signatures of reference include/ORBmatcher.h:48-88 (defined in ORBmatcher.cc) ----
    // Tracking: local map -> frame (src/ORBmatcher.cc:48-125)
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    // Tracking: last frame -> current frame (:1507-1619)
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th);
    // Relocalisation: key frame -> frame (:1622-1746)
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, float th, int ORBdist);
    // Loop closing: map points through a similarity into a key frame (:286-407)
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);
    // Same vocabulary node, key frame -> frame (:155-281) and key frame -> key frame (:715-850)
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    // Window around the position in frame 1 (:408-516); the same behind a pose guess (:519-594)
    int WindowSearch(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2, int minOctave = -1, int maxOctave = INT_MAX);
    int SearchByProjection(Frame& F1, Frame& F2, int windowSize, std::vector<MapPoint*>& vpMapPointMatches2);
    // Map initialisation (:596-713)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
    // New map points: epipolar constraint (:852-1014)
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<cv::KeyPoint>& vMatchedKeys1,
                               std::vector<cv::KeyPoint>& vMatchedKeys2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs);
    // Loop closing: both directions through [s12*R12|t12] (:1267-1505)
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, float th);
    // Duplicate map points (:1016-1134, :1136-1265)
    int Fuse(KeyFrame* pKF, std::vector<MapPoint*>& vpMapPoints, float th = 2.5);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th = 2.5);

    // ---- dense / candidate-list forms of the scan every search shares (not in the reference class) ----
    // For each row of Q (N x 32, CV_8U) the two smallest distances over all rows of T, first index on ties (e.g. src/ORBmatcher.cc:201-222).
    void MatchTop2(const cv::Mat& Q, const cv::Mat& T, std::vector<int>& bestIdx, std::vector<int>& bestDist, std::vector<int>& bestDist2) const {
        if (!Q.empty() && (!Q.isContinuous() || Q.cols != 32)) throw std::runtime_error("MatchTop2: Q must be N x 32 continuous");
        if (!T.empty() && (!T.isContinuous() || T.cols != 32)) throw std::runtime_error("MatchTop2: T must be M x 32 continuous");
        const int nq = Q.rows, nt = T.rows;
        bestIdx.assign(nq, -1); bestDist.assign(nq, INT_MAX); bestDist2.assign(nq, INT_MAX);
        if (nq == 0) return;
        const int rc = orbm_match_top2(Q.data, nq, T.data, nt, bestIdx.data(), bestDist.data(), bestDist2.data(), device_);
        if (rc != ORBX_OK) throw std::runtime_error("orbm_match_top2 failed: no usable MI355X / HIP runtime");
    }

    // The same scan skipping train rows that are already taken — `if(vpMapPointMatches[realIdxF]) continue;` (reference
    // src/ORBmatcher.cc:205-206): tValid[t] != 0 marks the rows still in play; bestIdx holds rows of T.
    void MatchTop2(const cv::Mat& Q, const cv::Mat& T, const std::vector<unsigned char>& tValid, std::vector<int>& bestIdx, std::vector<int>& bestDist,
                   std::vector<int>& bestDist2) const {
        if (!Q.empty() && (!Q.isContinuous() || Q.cols != 32)) throw std::runtime_error("MatchTop2: Q must be N x 32 continuous");
        if (!T.empty() && (!T.isContinuous() || T.cols != 32)) throw std::runtime_error("MatchTop2: T must be M x 32 continuous");
        if ((int)tValid.size() != T.rows) throw std::runtime_error("MatchTop2: tValid must have T.rows entries");
        const int nq = Q.rows, nt = T.rows;
        bestIdx.assign(nq, -1); bestDist.assign(nq, INT_MAX); bestDist2.assign(nq, INT_MAX);
        if (nq == 0 || nt == 0) return;
        const int rc = orbm_match_top2_masked(Q.data, nq, T.data, nt, tValid.data(), bestIdx.data(), bestDist.data(), bestDist2.data(), device_);
        if (rc != ORBX_OK) throw std::runtime_error("orbm_match_top2_masked failed: no usable MI355X / HIP runtime");
    }

    // Candidate-set form of the same scan (GetFeaturesInArea windows, vocabulary-node feature lists): query q scans
    // T rows cand[segOff[q] .. segOff[q+1]) in list order; bestIdx holds train row indices.
    void MatchTop2Candidates(const cv::Mat& Q, const cv::Mat& T, const std::vector<int>& segOff, const std::vector<int>& cand,
                             std::vector<int>& bestIdx, std::vector<int>& bestDist, std::vector<int>& bestDist2) const {
        const int nq = Q.rows;
        if ((int)segOff.size() != nq + 1) throw std::runtime_error("MatchTop2Candidates: segOff must have Q.rows+1 entries");
        bestIdx.assign(nq, -1); bestDist.assign(nq, INT_MAX); bestDist2.assign(nq, INT_MAX);
        if (nq == 0) return;
        const int rc = orbm_match_top2_segments(Q.data, nq, T.data, T.rows, segOff.data(), cand.empty() ? nullptr : cand.data(),
                                                bestIdx.data(), bestDist.data(), bestDist2.data(), device_);
        if (rc != ORBX_OK) throw std::runtime_error("orbm_match_top2_segments failed");
    }

    // Accept rule of SearchByBoW (reference src/ORBmatcher.cc:224-226) applied to MatchTop2's output
    int CountAccepted(const std::vector<int>& bestDist, const std::vector<int>& bestDist2, int th = TH_LOW) const {
        return orbm_count_accepted(bestDist.data(), bestDist2.data(), (int)bestDist.size(), th, mfNNratio);
    }

public:
    static const int TH_LOW = 50;        // reference include/ORBmatcher.h:92-94, src/ORBmatcher.cc:40-42
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;

protected:
    // reference include/ORBmatcher.h:99-103 (host helpers; the device searches carry their own copies of these rules)
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

    float mfNNratio;
    bool mbCheckOrientation;
    int device_;
};

}  // namespace ORB_SLAM
