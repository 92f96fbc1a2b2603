// ORACLE SUPPORT — TEST INFRASTRUCTURE ONLY.  C entry points around the reference's own vocabulary code
// (ORB_SLAM::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, /root/reference/include/
// ORBVocabulary.h and Thirdparty/DBoW2/DBoW2/*), compiled where it lies against oracle/cvstub (cv::Mat is only a
// 32-byte container there: all arithmetic on this path is DBoW2's own).  Used by tests/test_ref_pin.py.
#include <opencv2/core/core.hpp>
#include "ORBVocabulary.h"

namespace {
struct Voc : public ORB_SLAM::ORBVocabulary {
    // the per-feature descent (TemplatedVocabulary.h:1218-1259) is protected in the reference class
    void one(const cv::Mat& f, DBoW2::WordId& id, DBoW2::WordValue& w, DBoW2::NodeId* nid, int levelsup) const {
        transform(f, id, w, nid, levelsup);
    }
};
std::vector<cv::Mat> rows_of(const unsigned char* desc, int n) {     // Converter::toDescriptorVector: one 1x32 Mat per row
    std::vector<cv::Mat> v;
    v.reserve(n);
    for (int i = 0; i < n; i++) v.push_back(cv::Mat(1, 32, CV_8U, (void*)(desc + (size_t)i * 32)));
    return v;
}
}  // namespace

extern "C" {
void* ref_voc_load_text(const char* path) {
    Voc* v = new Voc();
    if (!v->loadFromTextFile(path)) { delete v; return 0; }
    return v;
}
void ref_voc_destroy(void* h) { delete (Voc*)h; }
void ref_voc_info(void* h, int* k, int* L, int* scoring, int* weighting, int* nwords) {
    Voc* v = (Voc*)h;
    *k = v->getBranchingFactor(); *L = v->getDepthLevels(); *scoring = (int)v->getScoringType();
    *weighting = (int)v->getWeightingType(); *nwords = (int)v->size();
}
int ref_forb_distance(const unsigned char* a, const unsigned char* b) {
    cv::Mat ma(1, 32, CV_8U, (void*)a), mb(1, 32, CV_8U, (void*)b);
    return DBoW2::FORB::distance(ma, mb);
}
// per-feature word / weight / node at (L - levelsup)
void ref_voc_descend(void* h, const unsigned char* desc, int n, int levelsup, unsigned* word, double* weight, unsigned* node) {
    Voc* v = (Voc*)h;
    for (int i = 0; i < n; i++) {
        cv::Mat f(1, 32, CV_8U, (void*)(desc + (size_t)i * 32));
        DBoW2::WordId id; DBoW2::WordValue w; DBoW2::NodeId nid = 0;
        v->one(f, id, w, &nid, levelsup);
        word[i] = id; weight[i] = w; node[i] = nid;
    }
}
// Frame::ComputeBoW's call (src/Frame.cc:285): BowVector in map order; FeatureVector as CSR in map order
void ref_voc_transform(void* h, const unsigned char* desc, int n, int levelsup, unsigned* bow_id, double* bow_val, int* n_bow,
                       unsigned* fv_node, int* fv_off, unsigned* fv_feat, int* n_fv) {
    Voc* v = (Voc*)h;
    DBoW2::BowVector bv;
    DBoW2::FeatureVector fv;
    v->transform(rows_of(desc, n), bv, fv, levelsup);
    int i = 0;
    for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it, ++i) { bow_id[i] = it->first; bow_val[i] = it->second; }
    *n_bow = i;
    int j = 0, o = 0;
    fv_off[0] = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++j) {
        fv_node[j] = it->first;
        for (size_t q = 0; q < it->second.size(); q++) fv_feat[o++] = it->second[q];
        fv_off[j + 1] = o;
    }
    *n_fv = j;
}
double ref_voc_score(void* h, const unsigned* id1, const double* v1, int n1, const unsigned* id2, const double* v2, int n2) {
    DBoW2::BowVector a, b;
    for (int i = 0; i < n1; i++) a.insert(a.end(), std::make_pair(id1[i], v1[i]));
    for (int i = 0; i < n2; i++) b.insert(b.end(), std::make_pair(id2[i], v2[i]));
    return ((Voc*)h)->score(a, b);
}
}
