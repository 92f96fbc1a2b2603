// The per-frame front-end of ORB_SLAM's Tracking thread with everything between the camera image and the matches resident
// on the GPU, driven from plain C++ through the C ABI only (no HIP headers, no OpenCV):
//   Frame::Frame            src/Frame.cc:56-127   extract -> UndistortKeyPoints -> grid
//   Frame::ComputeBoW       src/Frame.cc:280-287  bag-of-words transform
//   ORBmatcher::WindowSearch(last, current, window, ...)   src/ORBmatcher.cc:408-516   (as Tracking.cc:497-502 calls it)
// usage: example_pipeline <w> <h> <frameA.raw> <frameB.raw> <vocabulary.txt> <out.bin>
// out.bin: for frame B: nB, keypoints_un (28 B each), cell_off[3073], n_bow, (word u32, value f64)*, then the search of A's keypoints
// in B: nmatches, q2t[nA]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbf.h"
#include "orbs.h"
#include "orbv.h"
#include "orbx.h"

#define CHECK(x) do { const int rc_ = (x); if (rc_ != ORBX_OK) { std::fprintf(stderr, "%s -> %d\n", #x, rc_); return 3; } } while (0)

template <typename T> static T* dalloc(size_t n) { void* p = nullptr; if (orbx_device_alloc(0, n * sizeof(T), &p) != ORBX_OK) std::exit(4); return (T*)p; }

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: %s w h a.raw b.raw voc.txt out.bin\n", argv[0]); return 2; }
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    std::vector<unsigned char> img((size_t)2 * w * h);
    for (int f = 0; f < 2; f++) {
        FILE* fp = std::fopen(argv[3 + f], "rb");
        if (!fp || std::fread(img.data() + (size_t)f * w * h, 1, (size_t)w * h, fp) != (size_t)w * h) { std::fprintf(stderr, "cannot read frame\n"); return 2; }
        std::fclose(fp);
    }
    // ---- set-up (once per run): extractor, vocabulary, camera
    orbx_params p;
    orbx_default_params(&p);
    p.max_batch = 2;
    orbx_extractor* ex = nullptr;
    CHECK(orbx_create(&p, &ex));
    orbv_vocabulary* voc = nullptr;
    CHECK(orbv_load_text(argv[5], 0, &voc));
    orbf_camera cam;
    std::memset(&cam, 0, sizeof cam);
    cam.K[0] = 517.3f; cam.K[2] = 318.6f; cam.K[4] = 516.5f; cam.K[5] = 255.3f; cam.K[8] = 1.f;       // TUM fr1
    cam.dist[0] = 0.2624f; cam.dist[1] = -0.9531f; cam.dist[2] = -0.0054f; cam.dist[3] = 0.0026f;
    cam.ndist = 4; cam.width = w; cam.height = h;
    orbf_bounds bounds;
    CHECK(orbf_image_bounds(&cam, &bounds));
    const int cap = orbx_max_keypoints(ex), F = 2;

    // ---- device buffers for two frames (A = last, B = current)
    unsigned char* d_img = dalloc<unsigned char>((size_t)F * w * h);
    orbx_keypoint* d_kps = dalloc<orbx_keypoint>((size_t)F * cap);
    orbx_keypoint* d_un = dalloc<orbx_keypoint>((size_t)F * cap);
    unsigned char* d_desc = dalloc<unsigned char>((size_t)F * cap * 32);
    int32_t* d_n = dalloc<int32_t>(F);
    int32_t* d_off = dalloc<int32_t>((size_t)F * (ORBF_GRID_CELLS + 1));
    int32_t* d_feat = dalloc<int32_t>((size_t)F * cap);
    uint32_t* d_bow_id = dalloc<uint32_t>((size_t)F * cap);
    double* d_bow_val = dalloc<double>((size_t)F * cap);
    uint32_t* d_fv_node = dalloc<uint32_t>((size_t)F * cap);
    int32_t* d_fv_off = dalloc<int32_t>((size_t)F * (cap + 1));
    uint32_t* d_fv_feat = dalloc<uint32_t>((size_t)F * cap);
    int32_t* d_cnt = dalloc<int32_t>(2 * F);
    CHECK(orbx_device_upload(0, d_img, img.data(), img.size()));

    // ---- Frame::Frame + ComputeBoW for both frames, nothing leaves the device
    CHECK(orbx_extract_batch_device(ex, d_img, F, w, h, w, (ptrdiff_t)w * h, d_kps, d_desc, d_n, cap, nullptr, nullptr));
    CHECK(orbf_undistort_grid_batch_device(&cam, &bounds, d_kps, d_n, F, cap, d_un, d_off, d_feat, nullptr));
    CHECK(orbv_transform_batch_device(voc, d_desc, d_n, F, cap, 4, d_bow_id, d_bow_val, d_cnt, d_fv_node, d_fv_off, d_fv_feat, d_cnt + F, nullptr));

    // ---- WindowSearch(last = A, current = B, 100, ...): queries are A's undistorted keypoints at their own level
    std::vector<int32_t> n(F);
    CHECK(orbx_device_download(0, n.data(), d_n, F * sizeof(int32_t)));
    const int nA = n[0], nB = n[1];
    std::vector<orbx_keypoint> unA(cap);
    CHECK(orbx_device_download(0, unA.data(), d_un, (size_t)cap * sizeof(orbx_keypoint)));
    std::vector<float> qxyr((size_t)cap * 3, 0.f), qang(cap, 0.f);
    std::vector<int32_t> qlev((size_t)cap * 2, 0);
    for (int i = 0; i < nA; i++) {
        qxyr[3 * i] = unA[i].x; qxyr[3 * i + 1] = unA[i].y; qxyr[3 * i + 2] = 100.f;
        qlev[2 * i] = qlev[2 * i + 1] = unA[i].octave;
        qang[i] = unA[i].angle;
    }
    float* d_qxyr = dalloc<float>((size_t)cap * 3);
    int32_t* d_qlev = dalloc<int32_t>((size_t)cap * 2);
    float* d_qang = dalloc<float>(cap);
    int32_t* d_q2t = dalloc<int32_t>(cap);
    int32_t* d_t2q = dalloc<int32_t>(cap);
    int32_t* d_nm = dalloc<int32_t>(1);
    CHECK(orbx_device_upload(0, d_qxyr, qxyr.data(), qxyr.size() * 4));
    CHECK(orbx_device_upload(0, d_qlev, qlev.data(), qlev.size() * 4));
    CHECK(orbx_device_upload(0, d_qang, qang.data(), qang.size() * 4));
    const orbs_params prm = {ORBS_RULE_WINDOW, ORBS_TH_HIGH, 0.8f, 1};
    // train = frame B (slot 1 of every per-frame array), queries = frame A (slot 0): pointers offset by one frame
    CHECK(orbs_window_search_batch_device(&bounds, &prm, d_un + cap, d_desc + (size_t)cap * 32, d_off + (ORBF_GRID_CELLS + 1), d_feat + cap, d_n + 1, cap,
                                          nullptr, d_qxyr, d_qlev, d_desc, d_qang, nullptr, d_n, cap, 1, d_q2t, d_t2q, nullptr, nullptr, d_nm, nullptr));

    // ---- results
    std::vector<orbx_keypoint> unB(nB > 0 ? nB : 1);
    std::vector<int32_t> off(ORBF_GRID_CELLS + 1), q2t(nA > 0 ? nA : 1), cnt(2 * F);
    int32_t nm = 0;
    CHECK(orbx_device_download(0, unB.data(), d_un + cap, (size_t)nB * sizeof(orbx_keypoint)));
    CHECK(orbx_device_download(0, off.data(), d_off + (ORBF_GRID_CELLS + 1), off.size() * 4));
    CHECK(orbx_device_download(0, cnt.data(), d_cnt, cnt.size() * 4));
    CHECK(orbx_device_download(0, q2t.data(), d_q2t, (size_t)nA * 4));
    CHECK(orbx_device_download(0, &nm, d_nm, 4));
    const int nbow = cnt[1];
    std::vector<uint32_t> bid(nbow > 0 ? nbow : 1);
    std::vector<double> bval(nbow > 0 ? nbow : 1);
    CHECK(orbx_device_download(0, bid.data(), d_bow_id + cap, (size_t)nbow * 4));
    CHECK(orbx_device_download(0, bval.data(), d_bow_val + cap, (size_t)nbow * 8));
    FILE* o = std::fopen(argv[6], "wb");
    std::fwrite(&nB, 4, 1, o);
    std::fwrite(unB.data(), sizeof(orbx_keypoint), nB, o);
    std::fwrite(off.data(), 4, off.size(), o);
    std::fwrite(&nbow, 4, 1, o);
    for (int i = 0; i < nbow; i++) { std::fwrite(&bid[i], 4, 1, o); std::fwrite(&bval[i], 8, 1, o); }
    std::fwrite(&nm, 4, 1, o);
    std::fwrite(&nA, 4, 1, o);
    std::fwrite(q2t.data(), 4, nA, o);
    std::fclose(o);
    std::printf("nA=%d nB=%d words_B=%d window_matches=%d\n", nA, nB, nbow, nm);
    void* bufs[] = {d_img, d_kps, d_un, d_desc, d_n, d_off, d_feat, d_bow_id, d_bow_val, d_fv_node, d_fv_off, d_fv_feat, d_cnt, d_qxyr, d_qlev, d_qang, d_q2t, d_t2q, d_nm};
    for (void* b : bufs) orbx_device_free(0, b);
    orbv_destroy(voc);
    orbx_destroy(ex);
    return 0;
}
