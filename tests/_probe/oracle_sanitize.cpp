// ASan/UBSan run of the CPU oracle ("ASan/UBSan for host oracle tests").  Built by tests/test_oracle_sanitize.py with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all -O1 -g oracle/*.cpp orb_slam_amd/csrc/synth_frames.c this file
// Drives every oracle entry point on synthetic data; any out-of-bounds access, signed overflow, misaligned access or invalid
// shift inside the oracle aborts the process.  Exit code 0 = clean.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
void synth_frame(uint8_t* out, int w, int h, std::ptrdiff_t stride, int family, uint64_t frame_index);
void synth_descriptors(uint8_t* out, int n, uint64_t desc_seed);
struct orc_keypoint { float x, y, size, angle, response; int32_t octave, class_id; };
void* orc_create(int nfeatures, float scaleFactor, int nlevels, int scoreType, int fastTh, int blur_mode);
void orc_destroy(void* h);
int orc_extract(void* h, const uint8_t* img, int w, int hh, int stride, orc_keypoint* kps, uint8_t* desc, int cap);
void orc_match_top2(const uint8_t* Q, int nq, const uint8_t* T, int nt, int32_t* best_idx, int32_t* best, int32_t* second);
void* orc_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight);
void orc_voc_destroy(void* h);
void orc_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, unsigned* bow_id, double* bow_val, int* n_bow, unsigned* fv_node, int* fv_off, unsigned* fv_feat, int* n_fv);
struct Camera { float K[9]; float dist[8]; int32_t ndist, width, height; };
struct Bounds { int32_t min_x, max_x, min_y, max_y; float inv_w, inv_h; };
void orc_frame_bounds(const Camera* c, Bounds* b);
void orc_frame_undistort(const Camera* c, const orc_keypoint* kps, int n, orc_keypoint* out);
void orc_frame_grid(const Bounds* b, const orc_keypoint* kps_un, int n, int32_t* cell_off, int32_t* cell_feat);
int orc_window_search(const void* bounds, int rule, int th, float ratio, int check_orientation, const orc_keypoint* kps_un, const uint8_t* desc,
                      const int32_t* cell_off, const int32_t* cell_feat, int nt, const uint8_t* claimed_in, const float* qxyr, const int32_t* qlev,
                      const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq, int32_t* q2t, int32_t* t2q, int32_t* best_out, int32_t* second_out);
int orc_distinctive(const uint8_t* desc, int N, int32_t* best_median);
int orc_search_by_bow(int th, float ratio, int check_orientation, const uint32_t* kf_node, const int32_t* kf_off, const uint32_t* kf_feat, int kf_nnodes,
                      const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_valid, int nKF, const uint32_t* f_node, const int32_t* f_off,
                      const uint32_t* f_feat, int f_nnodes, const uint8_t* f_desc, const float* f_angle, int nF, int32_t* q2t, int32_t* t2q, int32_t* best_out,
                      int32_t* second_out);
int orc_search_by_bow_kf(int th_low, float ratio, int check_orientation, const uint32_t* node1, const int32_t* off1, const uint32_t* feat1, int nnodes1,
                         const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const uint32_t* node2, const int32_t* off2,
                         const uint32_t* feat2, int nnodes2, const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, int32_t* q2t, int32_t* t2q);
int orc_search_for_triangulation(int th_low, int check_orientation, const float* F12, const float* level_sigma2, const uint32_t* node1, const int32_t* off1,
                                 const uint32_t* feat1, int nnodes1, const void* kps1, const uint8_t* desc1, const uint8_t* has_mp1, int n1,
                                 const uint32_t* node2, const int32_t* off2, const uint32_t* feat2, int nnodes2, const void* kps2, const uint8_t* desc2,
                                 const uint8_t* has_mp2, int n2, int32_t* q2t, int32_t* t2q, int32_t* best_out, int32_t* second_out);
int orc_sim3_agreement(const int32_t* vnMatch1, int N1, const int32_t* vnMatch2, int N2, int32_t* out12);
int orc_check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* F12, float sigma2);
}

int main() {
    long checksum = 0;
    const int cfgs[][7] = {{640, 480, 1000, 8, 1, 20, 1}, {321, 243, 500, 5, 0, 9, 0}, {97, 83, 50, 4, 1, 20, 3}, {200, 600, 400, 4, 1, 20, 1}};
    std::vector<orc_keypoint> kps, last_kps;
    std::vector<uint8_t> desc, last_desc;
    for (const auto& c : cfgs) {
        const int w = c[0], h = c[1], nf = c[2];
        std::vector<uint8_t> img((size_t)(w + 5) * h);                        // strided rows
        synth_frame(img.data(), w, h, w + 5, c[6], 3);
        void* ex = orc_create(nf, c[3] == 5 ? 1.5f : 1.2f, c[3], c[4], c[5], 0);
        kps.assign(2 * nf, orc_keypoint());
        desc.assign((size_t)2 * nf * 32, 0);
        const int n = orc_extract(ex, img.data(), w, h, w + 5, kps.data(), desc.data(), 2 * nf);
        orc_destroy(ex);
        if (n < 0) return 10;
        kps.resize(n); desc.resize((size_t)n * 32);
        checksum += n;
        if (w == 640) { last_kps = kps; last_desc = desc; }
    }
    const int n = (int)last_kps.size();
    // matcher
    std::vector<int32_t> bi(n), bb(n), bs(n);
    std::vector<uint8_t> T((size_t)777 * 32);
    synth_descriptors(T.data(), 777, 5);
    orc_match_top2(last_desc.data(), n, T.data(), 777, bi.data(), bb.data(), bs.data());
    orc_match_top2(last_desc.data(), n, T.data(), 0, bi.data(), bb.data(), bs.data());
    // bag of words on a small full tree (k = 5, L = 3)
    {
        const int k = 5, L = 3;
        std::vector<int32_t> parent(1, 0);
        std::vector<uint8_t> leaf(1, 0);
        int lo = 0, hi = 1;
        for (int lev = 1; lev <= L; lev++) {
            const int start = (int)parent.size();
            for (int p = lo; p < hi; p++) for (int j = 0; j < k; j++) { parent.push_back(p); leaf.push_back(lev == L); }
            lo = start; hi = (int)parent.size();
        }
        const int nn = (int)parent.size();
        std::vector<uint8_t> nd((size_t)nn * 32);
        synth_descriptors(nd.data(), nn, 9);
        std::vector<double> wt(nn, 0.0);
        for (int i = 0; i < nn; i++) if (leaf[i]) wt[i] = 0.5 + (i % 7);
        void* v = orc_voc_create(k, L, 0, 0, nn, parent.data(), leaf.data(), nd.data(), wt.data());
        std::vector<unsigned> bid(n), fn(n), ff(n);
        std::vector<double> bv(n);
        std::vector<int> fo(n + 1);
        int nb = 0, nfv = 0;
        orc_voc_transform(v, last_desc.data(), n, 2, bid.data(), bv.data(), &nb, fn.data(), fo.data(), ff.data(), &nfv);
        orc_voc_transform(v, last_desc.data(), 0, 2, bid.data(), bv.data(), &nb, fn.data(), fo.data(), ff.data(), &nfv);
        orc_voc_destroy(v);
        checksum += nb + nfv;
    }
    // frame steps + the four window-search rules
    {
        Camera cam = {{517.3f, 0, 318.6f, 0, 516.5f, 255.3f, 0, 0, 1}, {0.2624f, -0.9531f, -0.0054f, 0.0026f, 0, 0, 0, 0}, 4, 640, 480};
        Bounds b;
        orc_frame_bounds(&cam, &b);
        std::vector<orc_keypoint> un(n);
        orc_frame_undistort(&cam, last_kps.data(), n, un.data());
        std::vector<int32_t> off(64 * 48 + 1), feat(n);
        orc_frame_grid(&b, un.data(), n, off.data(), feat.data());
        std::vector<float> qxyr((size_t)n * 3), qa(n);
        std::vector<int32_t> ql((size_t)n * 2), q2t(n), t2q(n), be(n), se(n);
        std::vector<uint8_t> qv(n, 1), cl(n, 0);
        for (int i = 0; i < n; i++) {
            const int s = (i * 7 + 3) % n;
            qxyr[3 * i] = un[s].x + (i % 5) - 2; qxyr[3 * i + 1] = un[s].y - (i % 3); qxyr[3 * i + 2] = i % 11 == 0 ? 500.f : 14.f;
            ql[2 * i] = un[s].octave - 1; ql[2 * i + 1] = un[s].octave + 1;
            qa[i] = un[s].angle; qv[i] = i % 9 != 0; cl[i] = i % 6 == 0;
        }
        for (int rule = 0; rule < 4; rule++)
            checksum += orc_window_search(&b, rule, rule == 3 ? 50 : 100, 0.8f, 1, un.data(), last_desc.data(), off.data(), feat.data(), n, rule == 0 ? cl.data() : nullptr,
                                          qxyr.data(), ql.data(), last_desc.data(), qa.data(), qv.data(), n, q2t.data(), t2q.data(), be.data(), se.data());
        checksum += orc_window_search(&b, 5, 50, 0.0f, 0, un.data(), last_desc.data(), off.data(), feat.data(), n, nullptr, qxyr.data(), ql.data(),
                                      last_desc.data(), nullptr, qv.data(), n, q2t.data(), t2q.data(), be.data(), se.data());
        // the vocabulary-node searches on a FeatureVector of the frame against itself (every third feature dropped on one side), the
        // agreement check and the epipolar test
        std::vector<unsigned> node, featv, node2, featv2;
        std::vector<int32_t> offv(1, 0), offv2(1, 0);
        for (int i = 0; i < n; i += 16) {
            node.push_back(100 + i); node2.push_back(100 + i + (i % 64 == 0 ? 1 : 0));
            for (int j = i; j < i + 16 && j < n; j++) { featv.push_back(j); if (j % 3) featv2.push_back(j); }
            offv.push_back((int)featv.size()); offv2.push_back((int)featv2.size());
        }
        std::vector<uint8_t> mp1(n), mp2(n);
        for (int i = 0; i < n; i++) { mp1[i] = i % 4 == 0; mp2[i] = i % 5 == 0; }
        const float F12[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
        const float sig[8] = {1.f, 1.44f, 2.0736f, 2.98598f, 4.29982f, 6.19174f, 8.9161f, 12.8392f};
        for (int chk = 0; chk < 2; chk++) {
            checksum += orc_search_by_bow(50, 0.75f, chk, node.data(), offv.data(), featv.data(), (int)node.size(), last_desc.data(), qa.data(), qv.data(), n,
                                          node2.data(), offv2.data(), featv2.data(), (int)node2.size(), last_desc.data(), qa.data(), n, q2t.data(), t2q.data(),
                                          be.data(), se.data());
            checksum += orc_search_by_bow_kf(50, 0.9f, chk, node.data(), offv.data(), featv.data(), (int)node.size(), last_desc.data(), qa.data(), qv.data(), n,
                                             node2.data(), offv2.data(), featv2.data(), (int)node2.size(), last_desc.data(), qa.data(), cl.data(), n, q2t.data(),
                                             t2q.data());
            checksum += orc_search_for_triangulation(50, chk, F12, sig, node.data(), offv.data(), featv.data(), (int)node.size(), un.data(), last_desc.data(),
                                                     mp1.data(), n, node2.data(), offv2.data(), featv2.data(), (int)node2.size(), un.data(), last_desc.data(),
                                                     mp2.data(), n, q2t.data(), t2q.data(), be.data(), se.data());
        }
        checksum += orc_search_for_triangulation(50, 1, F12, sig, node.data(), offv.data(), featv.data(), 0, un.data(), last_desc.data(), mp1.data(), 0,
                                                 node2.data(), offv2.data(), featv2.data(), 0, un.data(), last_desc.data(), mp2.data(), 0, q2t.data(), t2q.data(),
                                                 be.data(), se.data());
        checksum += orc_sim3_agreement(q2t.data(), n, t2q.data(), n, be.data()) + orc_sim3_agreement(q2t.data(), 0, t2q.data(), 0, be.data());
        checksum += orc_check_dist_epipolar_line(1.f, 2.f, 3.f, 4.f, F12, 1.f) + orc_check_dist_epipolar_line(0.f, 0.f, 0.f, 0.f, std::vector<float>(9, 0.f).data(), 1.f);
        int32_t med;
        checksum += orc_distinctive(last_desc.data(), 40, &med) + orc_distinctive(last_desc.data(), 1, &med) + orc_distinctive(last_desc.data(), 0, &med);
    }
    std::printf("oracle sanitize run clean, checksum %ld\n", checksum);
    return 0;
}
