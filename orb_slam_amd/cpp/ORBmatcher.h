// ORB_SLAM::ORBmatcher — the part of the reference class that is on the hot path (reference
// include/ORBmatcher.h:41-44,:90-92 and the scan loop shared by all its searches), on the MI355X C ABI.
// The 11 Frame/KeyFrame/MapPoint search methods keep living in ORB-SLAM's host code (they walk Map objects
// under mutexes: out of scope); INTEGRATION.md shows how their inner loops call MatchTop2.
#pragma once
#include <climits>
#include <stdexcept>
#include <vector>

#include "cvcompat.h"
#include "orbx.h"

namespace ORB_SLAM {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}

    // Computes the Hamming distance between two ORB descriptors (reference src/ORBmatcher.cc:1794-1810)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbm_hamming256(a.ptr<unsigned char>(), b.ptr<unsigned char>()); }

    // Dense form of the best / second-best scan every search shares (e.g. reference src/ORBmatcher.cc:201-222):
    // for each row of Q (N x 32, CV_8U) the two smallest distances over all rows of T, first index on ties.
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

    static const int TH_LOW = 50;        // reference src/ORBmatcher.cc:40-42
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    int device_;
};

}  // namespace ORB_SLAM
