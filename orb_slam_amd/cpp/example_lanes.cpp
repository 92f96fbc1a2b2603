// The lanes configuration from plain C++ (orb_slam_amd/cpp/LanePipeline.h over the C ABI; no HIP headers on the host side).
// usage: example_lanes <width> <height> <frames_per_step> <steps> <lanes> <frames.raw> [out.bin] [bench_steps]
//   frames.raw = frames_per_step * steps grayscale frames.  Runs the sequence through `lanes` lanes and through ONE lane, checks that
//   every output of the last step is byte-identical, optionally dumps it ([n][kps][desc][match] as in LanePipeline::download) and,
//   with bench_steps > 0, times that many further steps over the same frames.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "LanePipeline.h"

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: %s w h frames_per_step steps lanes frames.raw [out.bin] [bench_steps]\n", argv[0]); return 2; }
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), B = std::atoi(argv[3]), steps = std::atoi(argv[4]), lanes = std::atoi(argv[5]);
    const int bench_steps = argc > 8 ? std::atoi(argv[8]) : 0;
    const size_t fbytes = (size_t)w * h, total = fbytes * B * steps;
    std::vector<uint8_t> frames(total);
    { std::ifstream f(argv[6], std::ios::binary); if (!f.read(reinterpret_cast<char*>(frames.data()), (std::streamsize)total)) { std::fprintf(stderr, "short read\n"); return 2; } }
    try {
        orbx_params p;
        orbx_default_params(&p);
        if (const char* nf = std::getenv("EXAMPLE_NFEATURES")) p.nfeatures = std::atoi(nf);
        void* d_frames = nullptr;
        if (orbx_device_alloc(p.device, total, &d_frames) != ORBX_OK || orbx_device_upload(p.device, d_frames, frames.data(), total) != ORBX_OK) {
            std::fprintf(stderr, "no device memory\n");
            return 1;
        }
        std::vector<int32_t> n[2], match[2];
        std::vector<orbx_keypoint> kps[2];
        std::vector<uint8_t> desc[2];
        int cap = 0;
        for (int run = 0; run < 2; ++run) {
            ORB_SLAM::LanePipeline pipe(w, h, B, run == 0 ? lanes : 1, p);
            cap = pipe.cap();
            pipe.tune(static_cast<const uint8_t*>(d_frames));       // explicit, blocking placement probe (no-op for one lane)
            for (int i = 0; i < steps; ++i) pipe.step(static_cast<const uint8_t*>(d_frames) + (size_t)i * B * fbytes);
            pipe.synchronize();
            pipe.download(n[run], kps[run], desc[run], match[run]);
            if (run == 0) {
                std::printf("lanes %d x %d frames, cap %d\n", pipe.lanes(), pipe.frames_per_lane(), cap);
                for (size_t k = 0; k < pipe.placement_ms().size(); ++k) std::printf("  stream set %zu: %.3f ms per step%s\n", k, pipe.placement_ms()[k], (int)k == pipe.placement_chosen() ? "  <- chosen" : "");
            }
            if (run == 0 && bench_steps > 0) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < bench_steps; ++i) pipe.step(static_cast<const uint8_t*>(d_frames) + (size_t)(i % steps) * B * fbytes);
                pipe.synchronize();
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("%d steps of %d frames in %.3f ms: %.0f frames/s extract + match\n", bench_steps, B, s * 1e3, (double)bench_steps * B / s);
            }
        }
        // every valid output byte of the last step must agree between the two runs
        long bad = 0;
        for (int f = 0; f < B; ++f) {
            if (n[0][f] != n[1][f]) { ++bad; continue; }
            const size_t k = (size_t)n[0][f];
            bad += std::memcmp(&kps[0][(size_t)f * cap], &kps[1][(size_t)f * cap], k * sizeof(orbx_keypoint)) != 0;
            bad += std::memcmp(&desc[0][(size_t)f * cap * 32], &desc[1][(size_t)f * cap * 32], k * 32) != 0;
            for (int c = 0; c < 3; ++c)
                bad += std::memcmp(&match[0][((size_t)c * B + f) * cap], &match[1][((size_t)c * B + f) * cap], k * 4) != 0;
        }
        std::printf("%s: last step, %d frames, lanes vs one stream\n", bad ? "DIFFERENT" : "IDENTICAL", B);
        if (argc > 7 && std::strlen(argv[7])) {
            std::ofstream o(argv[7], std::ios::binary);
            const int32_t hdr[4] = {B, cap, (int32_t)sizeof(orbx_keypoint), 0};
            o.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
            o.write(reinterpret_cast<const char*>(n[0].data()), (std::streamsize)(n[0].size() * 4));
            o.write(reinterpret_cast<const char*>(kps[0].data()), (std::streamsize)(kps[0].size() * sizeof(orbx_keypoint)));
            o.write(reinterpret_cast<const char*>(desc[0].data()), (std::streamsize)desc[0].size());
            o.write(reinterpret_cast<const char*>(match[0].data()), (std::streamsize)(match[0].size() * 4));
        }
        (void)orbx_device_free(p.device, d_frames);
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
