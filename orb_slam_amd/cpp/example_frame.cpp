// What Frame::Frame does at the drop-in boundary (reference src/Frame.cc:56-65), against the shim classes.
// usage: example_frame <w> <h> <raw 8-bit image file> <out file> [vocabulary.txt bow_out]
//   writes N, keypoints (28 B each), descriptors; with a vocabulary also Frame::ComputeBoW (src/Frame.cc:280-287):
//   bow_out = nBow, (u32 word, f64 value)*, nNodes, (u32 node, u32 count, u32 feature*)*
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s w h image.raw out.bin\n", argv[0]); return 2; }
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    std::vector<unsigned char> buf((size_t)w * h);
    FILE* f = std::fopen(argv[3], "rb");
    if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fprintf(stderr, "cannot read image\n"); return 2; }
    std::fclose(f);
    cv::Mat im(h, w, CV_8UC1, buf.data());

    ORB_SLAM::ORBextractor* mpORBextractor = new ORB_SLAM::ORBextractor(1000, 1.2f, 8, ORB_SLAM::ORBextractor::FAST_SCORE, 20);
    std::vector<cv::KeyPoint> mvKeys;
    cv::Mat mDescriptors;
    (*mpORBextractor)(im, cv::Mat(), mvKeys, mDescriptors);        // the reference call, verbatim
    const int N = (int)mvKeys.size();

    ORB_SLAM::ORBmatcher matcher(0.6, true);
    std::vector<int> idx, best, second;
    matcher.MatchTop2(mDescriptors, mDescriptors, idx, best, second);
    int self = 0;
    for (int i = 0; i < N; i++) self += (best[i] == 0);
    const int d01 = N >= 2 ? ORB_SLAM::ORBmatcher::DescriptorDistance(mDescriptors.row(0), mDescriptors.row(1)) : -1;

    FILE* o = std::fopen(argv[4], "wb");
    std::fwrite(&N, 4, 1, o);
    std::fwrite(mvKeys.data(), sizeof(cv::KeyPoint), N, o);
    for (int i = 0; i < N; i++) std::fwrite(mDescriptors.ptr(i), 1, 32, o);
    std::fclose(o);
    std::printf("N=%d levels=%d scale=%.3f self_matches=%d d01=%d\n", N, mpORBextractor->GetLevels(), mpORBextractor->GetScaleFactor(), self, d01);
    if (argc >= 7) {
        ORB_SLAM::ORBVocabulary Vocabulary;
        if (!Vocabulary.loadFromTextFile(argv[5])) { std::fprintf(stderr, "Wrong path to vocabulary\n"); return 3; }   // src/main.cc:98-104
        ORB_SLAM::ORBVocabulary* mpORBvocabulary = &Vocabulary;
        DBoW2::BowVector mBowVec;
        DBoW2::FeatureVector mFeatVec;
        std::vector<cv::Mat> vCurrentDesc;                           // Converter::toDescriptorVector
        for (int i = 0; i < N; i++) vCurrentDesc.push_back(mDescriptors.row(i));
        mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);      // the reference call, verbatim
        DBoW2::BowVector again;
        DBoW2::FeatureVector fvAgain;
        mpORBvocabulary->transform(mDescriptors, again, fvAgain, 4);
        FILE* b = std::fopen(argv[6], "wb");
        const unsigned nb = (unsigned)mBowVec.size(), nn = (unsigned)mFeatVec.size();
        std::fwrite(&nb, 4, 1, b);
        for (DBoW2::BowVector::const_iterator it = mBowVec.begin(); it != mBowVec.end(); ++it) { std::fwrite(&it->first, 4, 1, b); std::fwrite(&it->second, 8, 1, b); }
        std::fwrite(&nn, 4, 1, b);
        for (DBoW2::FeatureVector::const_iterator it = mFeatVec.begin(); it != mFeatVec.end(); ++it) {
            const unsigned c = (unsigned)it->second.size();
            std::fwrite(&it->first, 4, 1, b);
            std::fwrite(&c, 4, 1, b);
            std::fwrite(it->second.data(), 4, c, b);
        }
        std::fclose(b);
        std::printf("words=%u bow=%u nodes=%u self_score=%.6f same=%d\n", Vocabulary.size(), nb, nn, Vocabulary.score(mBowVec, mBowVec),
                    (int)(again == mBowVec && fvAgain == mFeatVec));
    }
    delete mpORBextractor;
    return 0;
}
