/* orbs.h — C ABI of the greedy grid-window searches (part of liborbx.so).
 *
 * SURVEY.md §8f row N2 with §8a rows M2–M4: the per-frame searches of ORBmatcher that scan a Frame's 64x48 grid window
 * around every query, skip train features claimed by EARLIER queries (the sequential "already matched" masking), apply
 * the accept rule and, where the reference does, the rotation-consistency histogram.  Exact reference semantics, batched
 * over many independent search problems (one per frame / frame pair), all inputs and outputs device resident.
 *
 * Reference interfaces replaced (paths relative to /root/reference), selected by `rule`:
 *   ORBS_RULE_MAPPOINTS  <- int ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, float th)   src/ORBmatcher.cc:48-125
 *                           (best / second with their octaves; reject only when both lie on one level and best > ratio*second)
 *   ORBS_RULE_WINDOW     <- int ORBmatcher::WindowSearch(Frame&, Frame&, int, vector<MapPoint*>&, int, int)   :408-516
 *                           int ORBmatcher::SearchByProjection(Frame&, Frame&, int, vector<MapPoint*>&)       :519-594
 *                           (accept best <= second*ratio && best <= th)
 *   ORBS_RULE_BEST       <- int ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, float th)   :1507-1619
 *                           int ORBmatcher::SearchByProjection(Frame& Current, KeyFrame*, const set<MapPoint*>&, float th, int ORBdist)  :1622-1746
 *                           (accept best <= th; th = TH_HIGH resp. ORBdist)
 *                           int ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>& vpMatched, int th)  :286-407
 *                           (the same with th = TH_LOW, d_claimed = "vpMatched[idx] is set on entry" and no rotation check:
 *                           a match claims its feature, later map points skip it)
 *   ORBS_RULE_INIT       <- int ORBmatcher::SearchForInitialization(Frame&, Frame&, vector<Point2f>&, vector<int>&, int)   :596-716
 *                           (a train feature may be re-matched by a later query with a strictly smaller distance)
 *   ORBS_RULE_BOW        <- int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)                 :155-281
 *                           (candidates = the features of the same vocabulary node instead of a grid window:
 *                           orbs_bow_ranges_batch_device + orbs_list_search_batch_device; accept best <= th && best < ratio*second)
 *                           int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)              :715-850
 *                           (the same with d_claimed = "pMP2 is NULL or bad" and th = TH_LOW - 1: that function tests `bestDist1<TH_LOW`)
 *   ORBS_RULE_TRIANGULATION <- int ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12, ...)  :852-1014
 *                           with ORBmatcher::CheckDistEpipolarLine :136-153 (orbs_triangulation_search_batch_device: candidates of the
 *                           same vocabulary node with distance <= th, sorted by (distance, index); the first one within
 *                           2 x the best distance that lies on the query's epipolar line is taken)
 *   ORBS_RULE_FREE       <- the searches WITHOUT the "already matched" masking, every query independent: the scan of
 *                           int ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>&, float th)                       :1016-1134
 *                           int ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float th)    :1136-1265
 *                           and the two scans of SearchBySim3                                                   :1267-1505
 *                           (window + level range, best distance only, accept best <= th; several queries may end on the same
 *                           feature, so d_t2q is not produced (all -1) and there is no rotation check)
 *   orbs_agreement_batch_device <- the two scans of int ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, vector<MapPoint*>&, s12, R12, t12, th)
 *                           :1267-1505 are two ORBS_RULE_FREE searches (window + levels [predicted-1, predicted], best <= TH_HIGH);
 *                           this is its "check agreement" tail :1486-1505
 *   rotation filter      <- the rotHist blocks of those functions + ORBmatcher::ComputeThreeMaxima            :1748-1789
 *   candidate windows    <- Frame::GetFeaturesInArea                                                          src/Frame.cc:200-265
 *
 * What stays with the caller (pointer-graph work on MapPoint / Frame objects): which queries are valid (pMP != NULL,
 * !isBad(), mbTrackInView, level bounds → d_qvalid), their window centre / radius / level range (projection,
 * RadiusByViewingCos, scale factors → d_qxyr, d_qlev) and what a match means (vpMapPointMatches2[i2] = F1.mvpMapPoints[q]).
 *
 * Layout: problem p uses train slots [p*cap, p*cap + d_nt[p]) of d_kps_un / d_desc / d_claimed / d_t2q, its grid at
 * d_cell_off + p*(ORBF_GRID_CELLS+1), d_cell_feat + p*cap (exactly what orbf_undistort_grid_batch_device writes), and query
 * slots [p*qcap, p*qcap + d_nq[p]).  Status codes are orbx.h's; there is no CPU fallback.
 */
#ifndef ORBS_H
#define ORBS_H

#include <stddef.h>
#include <stdint.h>

#include "orbf.h"
#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBS_RULE_MAPPOINTS 0
#define ORBS_RULE_WINDOW    1
#define ORBS_RULE_BEST      2
#define ORBS_RULE_INIT      3
#define ORBS_RULE_BOW       4
#define ORBS_RULE_FREE      5
#define ORBS_RULE_TRIANGULATION 6

#define ORBS_MAX_LEVELS 16    /* pyramid levels the epipolar test of ORBS_RULE_TRIANGULATION can address */

#define ORBS_TH_HIGH 100      /* ORBmatcher::TH_HIGH src/ORBmatcher.cc:40 */
#define ORBS_TH_LOW  50       /* ORBmatcher::TH_LOW  src/ORBmatcher.cc:41 */

typedef struct orbs_params {
    int32_t rule;               /* ORBS_RULE_* */
    int32_t th;                 /* TH_HIGH / TH_LOW / ORBdist */
    float ratio;                /* mfNNratio */
    int32_t check_orientation;  /* mbCheckOrientation (ignored by ORBS_RULE_MAPPOINTS, which has no rotation check) */
} orbs_params;

/* LDS bytes one problem needs (the train frame is staged in LDS); ORBX_ERR_CAPACITY from the search when this exceeds
 * what a gfx950 workgroup can have (160 KiB).  cap = 1000 / qcap = 1000 needs ~60 KiB.  Up to ~2850 train features the
 * descriptors are staged too; larger frames keep them in global memory (25 instead of 57 bytes of LDS per feature), which
 * carries a problem to ~6500 train features — e.g. the 4000-feature initialisation extractor of an nFeatures = 2000 setup.
 * The grid searches of frames up to ~2400 features add a level-bucketed index (16 KiB more; included in the figure returned for
 * such sizes — the function reports what the launch really uses); larger frames are searched without it.
 *
 * PRECONDITION on the caller's grid CSR (d_cell_off / d_cell_feat): every feature is filed in exactly the cell the reference's
 * Frame::PosInGrid gives it (`round((x - mnMinX) * mfGridElementWidthInv)`, src/Frame.cc:267-277) — what orbf_undistort_grid
 * builds.  The plain CSR scan visits the reference's fine cell window; the bucketed index visits a superset of it (2 x 2 cells per
 * bucket) and relies on the exact |dx|,|dy| <= r test, so the two forms agree — and agree with the reference — only for a grid
 * filed by that rule.  A CSR built differently (floor instead of round, features filed twice, ...) is outside the contract. */
size_t orbs_lds_bytes(int cap, int qcap);
/* test hook: -1 = process default (ORBS_BUCKETS in the environment, on unless "0"), 0 = plain CSR scan for every size,
 * 1 = bucketed index where it fits.  Same results either way under the precondition above (tests/test_gpu_search.py runs both). */
int orbs_debug_set_buckets(int mode);
/* Launches of up to `nproblems` problems take the 1024-thread form of the search kernel (sixteen waves per problem: the latency form, what a
 * one-problem call of ORB_SLAM::ORBmatcher gets), larger ones the 256-thread form.  Same results.  Default: ORBS_WIDE_MAX in the environment, else 64;
 * -2 restores it; 0 = never, a large number = always (tests/test_gpu_search.py runs both). */
int orbs_debug_set_wide_max(int nproblems);

/* Outputs per problem: d_q2t[qcap] the train feature each query is finally matched to (-1 none), d_t2q[cap] the query each
 * train feature is finally matched to (-1 none), d_best / d_second[qcap] the two distances the scan left (INT_MAX as in the
 * reference; -1 for queries that were skipped or had an empty window; may be NULL), d_nmatches[p] the function's return
 * value.  d_claimed (may be NULL): train features that already hold a map point on entry (F.mvpMapPoints[idx] != NULL).
 * d_qangle: the query keypoints' angles (needed when check_orientation; may be NULL otherwise); d_qvalid may be NULL. */
int orbs_window_search_batch_device(const orbf_bounds* b, const orbs_params* prm,
                                    const orbx_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_cell_off,
                                    const int32_t* d_cell_feat, const int32_t* d_nt, int cap, const uint8_t* d_claimed,
                                    const float* d_qxyr, const int32_t* d_qlev, const uint8_t* d_qdesc, const float* d_qangle,
                                    const uint8_t* d_qvalid, const int32_t* d_nq, int qcap, int nproblems,
                                    int32_t* d_q2t, int32_t* d_t2q, int32_t* d_best, int32_t* d_second, int32_t* d_nmatches,
                                    void* stream);

/* The same in-order search over an explicit candidate LIST instead of a grid window: problem p's train features are listed in
 * d_list + p*cap (d_nlist[p] entries; a feature appears at most once), query q scans the list positions
 * [d_qrange[2q], d_qrange[2q+1]) in list order.  With the FeatureVector CSR of orbv_transform_batch_device as the list
 * (d_list = fv_feat, d_nlist = fv_off[n_fv]) and ranges from orbs_bow_ranges_batch_device this is SearchByBoW.
 * d_qindex (may be NULL): query q takes its descriptor / angle / valid flag from slot d_qindex[q] of the query-side arrays
 * instead of slot q (so the query frame's own FeatureVector order can be used without gathering).  d_q2t is indexed by query
 * position q.  Every rule is accepted; ORBS_RULE_BOW is the one the reference uses here. */
int orbs_list_search_batch_device(const orbs_params* prm, const orbx_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_list,
                                  const int32_t* d_nlist, const int32_t* d_nt, int cap, const uint8_t* d_claimed,
                                  const int32_t* d_qrange, const int32_t* d_qindex, const uint8_t* d_qdesc, const float* d_qangle,
                                  const uint8_t* d_qvalid, const int32_t* d_nq, int qcap, int nproblems,
                                  int32_t* d_q2t, int32_t* d_t2q, int32_t* d_best, int32_t* d_second, int32_t* d_nmatches, void* stream);

/* SearchForTriangulation over the same list form (train = pKF2 staged in its FeatureVector order, query ranges from
 * orbs_bow_ranges_batch_device, d_qindex = pKF1's fv_feat).  prm->rule must be ORBS_RULE_TRIANGULATION, prm->th = TH_LOW,
 * prm->ratio unused.  d_F12: 9 floats per problem (row major, `F12.at<float>(r,c)`); level_sigma2: HOST pointer to pKF2's
 * mvLevelSigma2 (nlevels <= ORBS_MAX_LEVELS).  d_kps1 / d_qdesc / d_qvalid are pKF1's undistorted keypoints, descriptors and
 * "has no MapPoint yet" flags in feature order (query q uses slot d_qindex[q]); d_claimed marks pKF2's features that already
 * hold a MapPoint.  Outputs as in orbs_list_search_batch_device: d_q2t[q] = vMatches12[idx1] for query position q; d_best = the
 * BestDist of the query's candidate list (INT_MAX when empty), d_second = the distance of the match taken (INT_MAX none).
 * Precondition: within one vocabulary node the train list ascends in feature index (true for the fv_feat of
 * orbv_transform_batch_device): distance ties are broken by list position here, by (distance, idx2) in the reference's sorted
 * vDistIndex — the two agree exactly when the list is in index order. */
int orbs_triangulation_search_batch_device(const orbs_params* prm, const float* d_F12, const float* level_sigma2, int nlevels,
                                           const orbx_keypoint* d_kps2, const uint8_t* d_desc2, const int32_t* d_list, const int32_t* d_nlist,
                                           const int32_t* d_nt, int cap, const uint8_t* d_claimed, const int32_t* d_qrange, const int32_t* d_qindex,
                                           const orbx_keypoint* d_kps1, const uint8_t* d_qdesc, const uint8_t* d_qvalid, const int32_t* d_nq, int qcap,
                                           int nproblems, int32_t* d_q2t, int32_t* d_t2q, int32_t* d_best, int32_t* d_second, int32_t* d_nmatches,
                                           void* stream);
/* The float bound the kernel compares CheckDistEpipolarLine's dsqr with: the smallest float >= 3.84 * (double)sigma2 (host). */
float orbs_epipolar_bound(float sigma2);

/* SearchBySim3's agreement check: d_out12[i1] = d_match12[i1] when d_match21[d_match12[i1]] == i1, else -1; d_nfound[p] = the
 * function's return value.  Problem p: d_match12 + p*cap1 (d_n1[p] entries), d_match21 + p*cap2 (d_n2[p] entries). */
int orbs_agreement_batch_device(const int32_t* d_match12, const int32_t* d_n1, int cap1, const int32_t* d_match21, const int32_t* d_n2, int cap2,
                                int nproblems, int32_t* d_out12, int32_t* d_nfound, void* stream);

/* SearchByBoW's merge walk (src/ORBmatcher.cc:171-260) as data: for problem p, query position j of the QUERY frame's
 * FeatureVector CSR (node ids d_fvq_node + p*cap, offsets d_fvq_off + p*(cap+1), d_nfv_q[p] nodes) gets the list range of
 * the SAME node in the TRAIN frame's FeatureVector (empty when the train frame has no feature under that node).  Also
 * writes d_nq[p] = the number of query positions (fvq_off[n_fv]). */
int orbs_bow_ranges_batch_device(const uint32_t* d_fvq_node, const int32_t* d_fvq_off, const int32_t* d_nfv_q,
                                 const uint32_t* d_fvt_node, const int32_t* d_fvt_off, const int32_t* d_nfv_t, int cap,
                                 int nproblems, int32_t* d_qrange, int32_t* d_nq, void* stream);

/* ORBmatcher::ComputeThreeMaxima on a histogram of bin sizes (host, re-entrant): ind[3], -1 = none */
void orbs_three_maxima(const int32_t* sizes, int L, int32_t* ind);

#ifdef __cplusplus
}
#endif
#endif
