// Test-only: exposes orb_slam_amd/csrc/orb_math.h (host instantiation) to ctypes so the CPU test
// suite can check the product's scalar arithmetic against the oracle / glibc without a GPU.
#include "orb_math.h"
#include <math.h>
#include <string.h>
#include <stdint.h>
extern "C" {
int probe_cv_round_f(float v) { return orbx::cv_round_f(v); }
float probe_fast_atan2(float y, float x) { return orbx::fast_atan2_deg(y, x); }
void probe_sincos(float a, float* s, float* c) { orbx::sincosf_orb(a, s, c); }
int probe_fast9_score(const int* d, int tmin) { return orbx::fast9_score(d, tmin); }
int probe_resize_px(int s00, int s01, int s10, int s11, int a0, int a1, int b0, int b1) { return orbx::resize_px(s00, s01, s10, s11, a0, a1, b0, b1); }
int probe_blur_round(int sum, int te) { return orbx::blur_round(sum, te); }
int probe_blur_taps7(const int* v) { return orbx::blur_taps7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]); }
int probe_reflect101(int p, int len) { return orbx::reflect101(p, len); }
// sweep all floats in [lo_bits, hi_bits] (positive floats: bit order == value order) against glibc
// sinf/cosf; returns the number of inputs where either differs, first few offenders in bad[]
long probe_sincos_sweep(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride, uint32_t* bad, int bad_cap) {
    long nbad = 0;
    for (uint64_t b = lo_bits; b <= hi_bits; b += stride) {
        uint32_t u = (uint32_t)b;
        float y; memcpy(&y, &u, 4);
        float s, c;
        orbx::sincosf_orb(y, &s, &c);
        float gs = sinf(y), gc = cosf(y);
        if (memcmp(&s, &gs, 4) || memcmp(&c, &gc, 4)) { if (nbad < bad_cap) bad[nbad] = u; nbad++; }
    }
    return nbad;
}
}
