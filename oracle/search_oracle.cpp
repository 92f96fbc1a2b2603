// =====================================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp's header; the same rules apply).
//
// CPU restatement of the greedy grid-window searches of ORBmatcher on flattened arrays:
// one loop over the queries IN ORDER, each scanning Frame::GetFeaturesInArea's candidates while skipping train
// features claimed by earlier queries, then the accept rule, then (optionally) the rotation-consistency filter.
//   rule 0  ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, float)      src/ORBmatcher.cc:48-125
//   rule 1  ORBmatcher::WindowSearch(Frame&, Frame&, int, vector<MapPoint*>&, int, int)   :408-516
//           ORBmatcher::SearchByProjection(Frame&, Frame&, int, vector<MapPoint*>&)       :519-594 (no rotation check)
//   rule 2  ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, float)      :1508-1619
//   rule 3  ORBmatcher::SearchForInitialization(Frame&, Frame&, vector<Point2f>&, vector<int>&, int)   :596-716
//   rule 2  also SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist) :1622-1746 and, without rotation check,
//           SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) :286-407 (vpMatched = the claim state)
//   rule 5  the scans of ORBmatcher::Fuse :1016-1134, :1136-1265 and of SearchBySim3 :1267-1505 (no claims)
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)                        :155-281   (orc_search_by_bow)
//   ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)                      :715-850   (orc_search_by_bow_kf)
//   ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, ...)             :852-1014  (orc_search_for_triangulation)
//   ORBmatcher::CheckDistEpipolarLine                                                     :136-153   (orc_check_dist_epipolar_line)
//   ORBmatcher::SearchBySim3 "check agreement"                                            :1486-1505 (orc_sim3_agreement)
//   ORBmatcher::ComputeThreeMaxima                                                        :1748-1789
// What stays with the caller (pointer-graph work): which queries are valid (pMP != NULL, !isBad(), mbTrackInView, level
// bounds), their window centres / radii (projection, RadiusByViewingCos, scale factors) and what a match means
// (vpMapPointMatches2[i2] = pMP1 ...).  Candidate windows come from frame_oracle.cpp (Frame::GetFeaturesInArea).
// PARITY PINNED to the reference's own src/ORBmatcher.cc for: rules 0, 1 (WindowSearch), 2, 3, 5 (through SearchBySim3, with its
// agreement check), both SearchByBoW, SearchForTriangulation + CheckDistEpipolarLine, ComputeThreeMaxima, DescriptorDistance —
// ORBmatcher.cc and the reference's ORBmatcher.h compile unmodified against plain-data stand-ins of Frame / KeyFrame / MapPoint
// (oracle/matcherstub, oracle/ref_orbmatcher_wrap.cpp -> _ref/libref_orbmatcher.so) and tests/test_ref_pin_matcher.py runs the same
// seeded problems through both — every function of ORBmatcher.cc, the projection-based ones at identity poses (so that the
// reference's own projection code runs and the test reproduces its float arithmetic).  orc_distinctive is pinned to the reference's own
// src/MapPoint.cc (MapPoint::ComputeDistinctiveDescriptors :185-250) through _ref/libref_mappoint.so.
// =====================================================================================
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" int orc_frame_features_in_area(const void* b, const void* kps_un, const int32_t* cell_off, const int32_t* cell_feat,
                                          float x, float y, float r, int minLevel, int maxLevel, int32_t* out);
extern "C" int orc_hamming256(const uint8_t* a, const uint8_t* b);

namespace {
struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };
const int HISTO_LENGTH = 30;      // src/ORBmatcher.cc:42

// src/ORBmatcher.cc:1748-1789
void ComputeThreeMaxima(const std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

int rot_bin(float a1, float a2) {
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}
}  // namespace

extern "C" {
void orc_three_maxima(const int32_t* sizes, int L, int32_t* ind) {
    std::vector<std::vector<int> > h(L);
    for (int i = 0; i < L; i++) h[i].resize(sizes[i]);
    int a = -1, b = -1, c = -1;
    ComputeThreeMaxima(h.data(), L, a, b, c);
    ind[0] = a; ind[1] = b; ind[2] = c;
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:216-244) for one map point's observed descriptors: returns BestIdx,
// *best_median = BestMedian; -1 / INT_MAX when there are none (the reference returns before touching mDescriptor).
int orc_distinctive(const uint8_t* desc, int N, int32_t* best_median) {
    *best_median = INT_MAX;
    if (N <= 0) return -1;
    std::vector<std::vector<float> > Distances(N, std::vector<float>(N));
    for (int i = 0; i < N; i++) {
        Distances[i][i] = 0;
        for (int j = i + 1; j < N; j++) {
            int distij = orc_hamming256(desc + (size_t)i * 32, desc + (size_t)j * 32);
            Distances[i][j] = distij;
            Distances[j][i] = distij;
        }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; i++) {
        std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
        std::sort(vDists.begin(), vDists.end());
        int median = vDists[0.5 * (N - 1)];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    *best_median = BestMedian;
    return BestIdx;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:155-281) on flattened arrays: the two
// FeatureVectors as CSR (std::map order = ascending node id), kf_valid[i] = (pMP != NULL && !pMP->isBad()) of key-frame feature i.
// Outputs: t2q[nF] = the key-frame feature whose map point vpMapPointMatches[iF] ends up holding (-1 none), q2t[nKF] its inverse,
// best / second[nKF] = the two distances the scan of key-frame feature i left (-1 when it was never visited).
int orc_search_by_bow(int th, float ratio, int check_orientation,
                      const uint32_t* kf_node, const int32_t* kf_off, const uint32_t* kf_feat, int kf_nnodes,
                      const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_valid, int nKF,
                      const uint32_t* f_node, const int32_t* f_off, const uint32_t* f_feat, int f_nnodes,
                      const uint8_t* f_desc, const float* f_angle, int nF,
                      int32_t* q2t, int32_t* t2q, int32_t* best_out, int32_t* second_out) {
    int nmatches = 0;
    for (int i = 0; i < nKF; i++) { q2t[i] = -1; best_out[i] = -1; second_out[i] = -1; }
    for (int i = 0; i < nF; i++) t2q[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a < kf_nnodes && b < f_nnodes) {
        if (kf_node[a] == f_node[b]) {
            for (int iKF = kf_off[a]; iKF < kf_off[a + 1]; iKF++) {
                const unsigned realIdxKF = kf_feat[iKF];
                if (!kf_valid[realIdxKF]) continue;
                int bestDist1 = INT_MAX, bestIdxF = -1, bestDist2 = INT_MAX;
                for (int iF = f_off[b]; iF < f_off[b + 1]; iF++) {
                    const unsigned realIdxF = f_feat[iF];
                    if (t2q[realIdxF] >= 0) continue;                    // `if(vpMapPointMatches[realIdxF]) continue;`
                    const int dist = orc_hamming256(kf_desc + (size_t)realIdxKF * 32, f_desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                best_out[realIdxKF] = bestDist1;
                second_out[realIdxKF] = bestDist2;
                if (bestDist1 <= th) {
                    if ((float)bestDist1 < ratio * (float)bestDist2) {
                        t2q[bestIdxF] = (int)realIdxKF;
                        q2t[realIdxKF] = bestIdxF;
                        if (check_orientation) rotHist[rot_bin(kf_angle[realIdxKF], f_angle[bestIdxF])].push_back(bestIdxF);
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (kf_node[a] < f_node[b]) {
            while (a < kf_nnodes && kf_node[a] < f_node[b]) a++;        // lower_bound(Fit->first)
        } else {
            while (b < f_nnodes && f_node[b] < kf_node[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { q2t[t2q[rotHist[i][j]]] = -1; t2q[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:715-850).
// valid1[i] / valid2[i] = (pMP != NULL && !pMP->isBad()).  q2t[n1]: the pKF2 feature whose map point vpMatches12[idx1] holds.
int orc_search_by_bow_kf(int th_low, float ratio, int check_orientation,
                         const uint32_t* node1, const int32_t* off1, const uint32_t* feat1, int nnodes1,
                         const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                         const uint32_t* node2, const int32_t* off2, const uint32_t* feat2, int nnodes2,
                         const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, int32_t* q2t, int32_t* t2q) {
    for (int i = 0; i < n1; i++) q2t[i] = -1;
    for (int i = 0; i < n2; i++) t2q[i] = -1;
    std::vector<char> vbMatched2(n2, 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    int nmatches = 0, a = 0, b = 0;
    while (a < nnodes1 && b < nnodes2) {
        if (node1[a] == node2[b]) {
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const unsigned idx1 = feat1[i1];
                if (!valid1[idx1]) continue;
                int bestDist1 = INT_MAX, bestIdx2 = -1, bestDist2 = INT_MAX;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const unsigned idx2 = feat2[i2];
                    if (vbMatched2[idx2] || !valid2[idx2]) continue;
                    const int dist = orc_hamming256(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < th_low) {                                             // strict here (:793)
                    if ((float)bestDist1 < ratio * (float)bestDist2) {
                        q2t[idx1] = bestIdx2;
                        t2q[bestIdx2] = (int)idx1;
                        vbMatched2[bestIdx2] = 1;
                        if (check_orientation) rotHist[rot_bin(angle1[idx1], angle2[bestIdx2])].push_back(idx1);
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) {
            while (a < nnodes1 && node1[a] < node2[b]) a++;
        } else {
            while (b < nnodes2 && node2[b] < node1[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { t2q[q2t[rotHist[i][j]]] = -1; q2t[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:136-153); F12 row major 3x3 floats, sigma2 = pKF2->GetSigma2(kp2.octave)
int orc_check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* F12, float sigma2) {
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:852-1014) up to vMatches12 (the three output vectors are its compaction).
// has_mp1 / has_mp2[i] = (vpMapPoints[i] != NULL).  q2t[n1] = vMatches12, t2q[n2] its inverse; best[n1] = BestDist of the
// query's vDistIndex (INT_MAX empty, -1 never visited), second[n1] = the distance of the match taken (INT_MAX none).
int orc_search_for_triangulation(int th_low, int check_orientation, const float* F12, const float* level_sigma2,
                                 const uint32_t* node1, const int32_t* off1, const uint32_t* feat1, int nnodes1,
                                 const void* kps1_, const uint8_t* desc1, const uint8_t* has_mp1, int n1,
                                 const uint32_t* node2, const int32_t* off2, const uint32_t* feat2, int nnodes2,
                                 const void* kps2_, const uint8_t* desc2, const uint8_t* has_mp2, int n2,
                                 int32_t* q2t, int32_t* t2q, int32_t* best_out, int32_t* second_out) {
    const KeyPoint* vKeysUn1 = (const KeyPoint*)kps1_;
    const KeyPoint* vKeysUn2 = (const KeyPoint*)kps2_;
    int nmatches = 0;
    std::vector<char> vbMatched2(n2, 0);
    for (int i = 0; i < n1; i++) { q2t[i] = -1; best_out[i] = -1; second_out[i] = -1; }
    for (int i = 0; i < n2; i++) t2q[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a < nnodes1 && b < nnodes2) {
        if (node1[a] == node2[b]) {
            for (int i1 = off1[a]; i1 < off1[a + 1]; i1++) {
                const unsigned idx1 = feat1[i1];
                if (has_mp1[idx1]) continue;                                          // "If there is already a MapPoint skip"
                const KeyPoint& kp1 = vKeysUn1[idx1];
                std::vector<std::pair<int, size_t> > vDistIndex;
                for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                    const unsigned idx2 = feat2[i2];
                    if (vbMatched2[idx2] || has_mp2[idx2]) continue;
                    const int dist = orc_hamming256(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
                    if (dist > th_low) continue;
                    vDistIndex.push_back(std::make_pair(dist, (size_t)idx2));
                }
                best_out[idx1] = INT_MAX;
                second_out[idx1] = INT_MAX;
                if (vDistIndex.empty()) continue;
                std::sort(vDistIndex.begin(), vDistIndex.end());
                const int BestDist = vDistIndex.front().first;
                const int DistTh = (int)round(2 * BestDist);
                best_out[idx1] = BestDist;
                for (size_t id = 0; id < vDistIndex.size(); id++) {
                    if (vDistIndex[id].first > DistTh) break;
                    const int currentIdx2 = (int)vDistIndex[id].second;
                    const KeyPoint& kp2 = vKeysUn2[currentIdx2];
                    if (orc_check_dist_epipolar_line(kp1.x, kp1.y, kp2.x, kp2.y, F12, level_sigma2[kp2.octave])) {
                        vbMatched2[currentIdx2] = 1;
                        q2t[idx1] = currentIdx2;
                        t2q[currentIdx2] = (int)idx1;
                        second_out[idx1] = vDistIndex[id].first;
                        nmatches++;
                        if (check_orientation) rotHist[rot_bin(kp1.angle, kp2.angle)].push_back(idx1);
                        break;
                    }
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) {
            while (a < nnodes1 && node1[a] < node2[b]) a++;
        } else {
            while (b < nnodes2 && node2[b] < node1[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { t2q[q2t[rotHist[i][j]]] = -1; q2t[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// ORBmatcher::SearchBySim3's agreement check (src/ORBmatcher.cc:1486-1505)
int orc_sim3_agreement(const int32_t* vnMatch1, int N1, const int32_t* vnMatch2, int N2, int32_t* out12) {
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        out12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { out12[i1] = idx2; nFound++; }
        }
    }
    (void)N2;
    return nFound;
}

// One search problem.  q2t[nq]: the train feature a query ended up matched to (-1 none); t2q[nt]: the query a train feature
// ended up matched to (-1 none; features with claimed[] set on entry keep -1).  best/second[nq]: the two distances the scan
// left for the query (INT_MAX where the reference had INT_MAX; untouched (-1) for skipped queries).  Returns nmatches.
int orc_window_search(const void* bounds, int rule, int th, float ratio, int check_orientation,
                      const KeyPoint* kps_un, const uint8_t* desc, const int32_t* cell_off, const int32_t* cell_feat, int nt,
                      const uint8_t* claimed_in,
                      const float* qxyr, const int32_t* qlev, const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq,
                      int32_t* q2t, int32_t* t2q, int32_t* best_out, int32_t* second_out) {
    int nmatches = 0;
    std::vector<uint8_t> claimed(nt, 0);
    if (claimed_in) memcpy(claimed.data(), claimed_in, nt);
    std::vector<int> vMatchedDistance(nt, INT_MAX);           // rule 3 (:608)
    for (int i = 0; i < nq; i++) { q2t[i] = -1; best_out[i] = -1; second_out[i] = -1; }
    for (int i = 0; i < nt; i++) t2q[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int32_t> near(nt > 0 ? nt : 1);
    for (int q = 0; q < nq; q++) {
        if (qvalid && !qvalid[q]) continue;
        const int nn = orc_frame_features_in_area(bounds, kps_un, cell_off, cell_feat, qxyr[3 * q], qxyr[3 * q + 1], qxyr[3 * q + 2],
                                                  qlev[2 * q], qlev[2 * q + 1], near.data());
        if (nn == 0) continue;
        int bestDist = INT_MAX, bestLevel = -1, bestDist2 = INT_MAX, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nn; c++) {
            const int idx = near[c];
            if (rule != 3 && rule != 5 && claimed[idx]) continue;
            const int dist = orc_hamming256(qdesc + (size_t)q * 32, desc + (size_t)idx * 32);
            if (rule == 3 && vMatchedDistance[idx] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps_un[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = kps_un[idx].octave; bestDist2 = dist; }
        }
        best_out[q] = bestDist;
        second_out[q] = bestDist2;
        bool accept = false;
        if (rule == 0) {                    // :114-121
            if (bestDist <= th) accept = !(bestLevel == bestLevel2 && bestDist > ratio * bestDist2);
        } else if (rule == 1) {             // :476 / :585
            accept = bestDist <= bestDist2 * ratio && bestDist <= th;
        } else if (rule == 2 || rule == 5) { // :1583; rule 5: Fuse :1113 / SearchByProjection(KeyFrame*, Scw, ...) :391 — no claims
            accept = bestDist <= th;
        } else {                            // :652-654
            accept = bestDist <= th && bestDist < (float)bestDist2 * ratio;
        }
        if (!accept) continue;
        if (rule == 3) {
            if (t2q[bestIdx] >= 0) { q2t[t2q[bestIdx]] = -1; nmatches--; }
            q2t[q] = bestIdx; t2q[bestIdx] = q; vMatchedDistance[bestIdx] = bestDist; nmatches++;
            if (check_orientation) rotHist[rot_bin(qangle[q], kps_un[bestIdx].angle)].push_back(q);
        } else if (rule == 5) {
            q2t[q] = bestIdx; nmatches++;                        // independent queries: nothing is claimed, t2q stays -1
        } else {
            claimed[bestIdx] = 1; q2t[q] = bestIdx; t2q[bestIdx] = q; nmatches++;
            if (check_orientation && rule != 0 && rule != 5) rotHist[rot_bin(qangle[q], kps_un[bestIdx].angle)].push_back(bestIdx);
        }
    }
    if (check_orientation && rule != 0 && rule != 5) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                if (rule == 3) {            // :696-703
                    const int idx1 = rotHist[i][j];
                    if (q2t[idx1] >= 0) { t2q[q2t[idx1]] = -1; q2t[idx1] = -1; nmatches--; }
                } else {                    // :503-507 / :1609-1613
                    const int i2 = rotHist[i][j];
                    q2t[t2q[i2]] = -1; t2q[i2] = -1; nmatches--;
                }
            }
        }
    }
    return nmatches;
}
}
