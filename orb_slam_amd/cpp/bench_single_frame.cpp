// Latency of the drop-in call from plain C++ (the C ABI of include/orbx.h, host buffers in and out, synchronous) — what
// ORB_SLAM's Tracking thread sees at src/Frame.cc:60.  usage: bench_single_frame [w h nfeatures [calls]]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx.h"

// the exact-integer S-blocks generator lives in libsynthframes.so (orb_slam_amd/csrc/synth_frames.c); a plain LCG pattern is enough here
static void make_frame(std::vector<uint8_t>& f, int w, int h, unsigned seed) {
    unsigned s = seed * 2654435761u + 12345u;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) f[(size_t)y * w + x] = (uint8_t)(((x + 2 * y) / 8) & 255);
    for (int r = 0; r < 200; r++) {
        s = s * 1664525u + 1013904223u; const int x0 = (s >> 8) % (w - 100);
        s = s * 1664525u + 1013904223u; const int y0 = (s >> 8) % (h - 100);
        s = s * 1664525u + 1013904223u; const int rw = 8 + (s >> 8) % 88;
        s = s * 1664525u + 1013904223u; const int rh = 8 + (s >> 8) % 88;
        s = s * 1664525u + 1013904223u; const uint8_t g = (uint8_t)(s >> 24);
        for (int y = y0; y < y0 + rh; y++) for (int x = x0; x < x0 + rw; x++) f[(size_t)y * w + x] = g;
    }
}

int main(int argc, char** argv) {
    const int w = argc > 3 ? std::atoi(argv[1]) : 640, h = argc > 3 ? std::atoi(argv[2]) : 480, nf = argc > 3 ? std::atoi(argv[3]) : 1000;
    const int calls = argc > 4 ? std::atoi(argv[4]) : 400;
    orbx_params p;
    orbx_default_params(&p);
    p.nfeatures = nf;
    orbx_extractor* ex = nullptr;
    if (orbx_create(&p, &ex) != ORBX_OK) { std::fprintf(stderr, "orbx_create failed (no gfx950 GPU?)\n"); return 1; }
    const int cap = orbx_max_keypoints(ex);
    std::vector<std::vector<uint8_t>> frames(16, std::vector<uint8_t>((size_t)w * h));
    for (int i = 0; i < 16; i++) make_frame(frames[i], w, h, 7 + i);
    std::vector<orbx_keypoint> kps(cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    for (int i = 0; i < 10; i++) if (orbx_extract(ex, frames[i].data(), w, h, w, kps.data(), desc.data(), cap, &n) != ORBX_OK) { std::fprintf(stderr, "%s\n", orbx_last_error(ex)); return 1; }
    std::vector<double> us(calls);
    for (int i = 0; i < calls; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        (void)orbx_extract(ex, frames[i & 15].data(), w, h, w, kps.data(), desc.data(), cap, &n);
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(us.begin(), us.end());
    std::printf("orbx_extract from C++ %dx%d nf=%d: median %.0f us  p10 %.0f  p90 %.0f  -> %.0f frames/s single stream (host buffers, H2D+D2H included), N=%d\n", w, h, nf,
                us[calls / 2], us[calls / 10], us[calls * 9 / 10], 1e6 / us[calls / 2], n);
    orbx_destroy(ex);
    return 0;
}
