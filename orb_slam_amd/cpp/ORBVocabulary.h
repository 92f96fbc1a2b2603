// ORB_SLAM::ORBVocabulary — the per-frame part of the reference vocabulary class (reference include/ORBVocabulary.h:31-32
// = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>), kept source compatible for its callers
//   src/main.cc:97-98            ORBVocabulary Vocabulary; Vocabulary.loadFromTextFile(path)
//   src/Frame.cc:284-285         mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)
//   src/KeyFrame.cc:62-63        the same for key frames
//   src/KeyFrameDatabase.cc:132  mpVoc->score(bowA, bowB);   :71  mpVoc->size()
// on top of the C ABI in include/orbv.h.  The tree lives in HBM; transform() runs on the MI355X (no CPU fallback).
// DBoW2::BowVector / FeatureVector keep the reference's container types (Thirdparty/DBoW2/DBoW2/BowVector.h:56-57,
// FeatureVector.h:21-22) so Frame / KeyFrame / ORBmatcher code that walks them compiles unchanged.  With the real DBoW2
// headers on the include path define ORBX_WITH_DBOW2 to reuse its types instead.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "cvcompat.h"
#include "orbv.h"
#include "orbx.h"

#ifdef ORBX_WITH_DBOW2
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#else
namespace DBoW2 {
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}  // namespace DBoW2
#endif

namespace ORB_SLAM {

class ORBVocabulary {
public:
    explicit ORBVocabulary(int device = 0) : v_(nullptr), device_(device) {}
    ~ORBVocabulary() { orbv_destroy(v_); }
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;

    // TemplatedVocabulary::loadFromTextFile: false on a missing / malformed file
    bool loadFromTextFile(const std::string& filename) {
        orbv_destroy(v_);
        v_ = nullptr;
        const int rc = orbv_load_text(filename.c_str(), device_, &v_);
        if (rc == ORBX_ERR_DEVICE) throw std::runtime_error("ORBVocabulary: no usable MI355X / HIP runtime");
        return rc == ORBX_OK;
    }
    bool empty() const { return size() == 0; }
    unsigned int size() const { int n = 0; if (v_) orbv_info(v_, nullptr, nullptr, nullptr, nullptr, &n, nullptr); return (unsigned)n; }
    int getBranchingFactor() const { int k = 0; if (v_) orbv_info(v_, &k, nullptr, nullptr, nullptr, nullptr, nullptr); return k; }
    int getDepthLevels() const { int L = 0; if (v_) orbv_info(v_, nullptr, &L, nullptr, nullptr, nullptr, nullptr); return L; }

    // the call of Frame::ComputeBoW: one 1x32 CV_8U Mat per feature (Converter::toDescriptorVector)
    void transform(const std::vector<cv::Mat>& features, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
        std::vector<unsigned char> flat(features.size() * 32);
        for (size_t i = 0; i < features.size(); i++) memcpy(&flat[i * 32], features[i].ptr<unsigned char>(), 32);
        transformRows(flat.data(), (int)features.size(), v, fv, levelsup);
    }
    // the same without the per-row Mat headers: descriptors as the N x 32 matrix the extractor returns
    void transform(const cv::Mat& descriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
        if (!descriptors.empty() && (!descriptors.isContinuous() || descriptors.cols != 32)) throw std::runtime_error("transform: N x 32 continuous");
        transformRows(descriptors.empty() ? nullptr : descriptors.ptr<unsigned char>(), descriptors.rows, v, fv, levelsup);
    }
    double score(const DBoW2::BowVector& a, const DBoW2::BowVector& b) const {
        std::vector<uint32_t> ia, ib;
        std::vector<double> va, vb;
        for (DBoW2::BowVector::const_iterator it = a.begin(); it != a.end(); ++it) { ia.push_back(it->first); va.push_back(it->second); }
        for (DBoW2::BowVector::const_iterator it = b.begin(); it != b.end(); ++it) { ib.push_back(it->first); vb.push_back(it->second); }
        return orbv_score(v_, ia.data(), va.data(), (int)ia.size(), ib.data(), vb.data(), (int)ib.size());
    }
    orbv_vocabulary* handle() const { return v_; }

private:
    void transformRows(const unsigned char* desc, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
        v.clear();
        fv.clear();
        if (!v_ || n == 0 || empty()) return;               // TemplatedVocabulary.h:1134-1137
        std::vector<uint32_t> bid(n), fnode(n), ffeat(n);
        std::vector<double> bval(n);
        std::vector<int32_t> foff(n + 1);
        int nb = 0, nf = 0;
        const int rc = orbv_transform(v_, desc, n, levelsup, bid.data(), bval.data(), &nb, fnode.data(), foff.data(), ffeat.data(), &nf);
        if (rc != ORBX_OK) throw std::runtime_error("orbv_transform failed");
        for (int i = 0; i < nb; i++) v.insert(v.end(), std::make_pair(bid[i], bval[i]));
        for (int j = 0; j < nf; j++) {
            std::vector<unsigned int>& dst = fv.insert(fv.end(), std::make_pair(fnode[j], std::vector<unsigned int>()))->second;
            dst.assign(ffeat.begin() + foff[j], ffeat.begin() + foff[j + 1]);
        }
    }
    orbv_vocabulary* v_;
    int device_;
};

}  // namespace ORB_SLAM
