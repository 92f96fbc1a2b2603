// Scalar arithmetic of the ORB front-end, written once for host and device (ORBX_HD) so the
// same expressions can be unit-tested on the CPU (tests/test_device_math.py builds this header
// with g++) and run inside the HIP kernels.  Everything here must evaluate identically under
// g++ and hipcc: no FMA contraction (-ffp-contract=off on both sides), IEEE +,-,*,/ only.
//
// Reference call sites (in /root/reference):
//   cvRound            src/ORBextractor.cc:102-103,:128,:166-167,:786  (OpenCV: round-half-to-even)
//   fastAtan2          src/ORBextractor.cc:150                         (OpenCV 2.4 mathfuncs.cpp)
//   cos/sin(float)     src/ORBextractor.cc:160                         (libm cosf/sinf)
//   cv::FAST score     src/ORBextractor.cc:607,:613                    (OpenCV fast_score.cpp cornerScore<16>)
//   cv::resize         src/ORBextractor.cc:800                         (OpenCV imgwarp.cpp, 8U fixed point)
//   cv::GaussianBlur   src/ORBextractor.cc:760                         (OpenCV smooth.cpp/filter.cpp, 8U fixed point)
//   DescriptorDistance src/ORBmatcher.cc:1794-1810
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ inline
#else
#define ORBX_HD inline
#endif

namespace orbx {

// cvRound(double(float v)): ties-to-even.  float->double is exact, so rintf suffices.
ORBX_HD int cv_round_f(float v) { return (int)rintf(v); }

ORBX_HD int imin(int a, int b) { return a < b ? a : b; }
ORBX_HD int imax(int a, int b) { return a > b ? a : b; }
ORBX_HD int imin3(int a, int b, int c) { return imin(imin(a, b), c); }
ORBX_HD int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

// BORDER_REFLECT_101 index map (OpenCV borderInterpolate).
ORBX_HD int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// ---- cv::fastAtan2 (degrees, [0,360)) -------------------------------------------------------
// Coefficient products are formed in float exactly as the static initialisers of OpenCV do.
ORBX_HD float fast_atan2_deg(float y, float x) {
    const float k = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = 2.2204460492503131e-16f;   // (float)DBL_EPSILON
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- sinf/cosf for the descriptor rotation ---------------------------------------------------
// The reference calls libm cosf/sinf (glibc).  A GPU cannot call glibc, so this is the published
// algorithm glibc >= 2.28 uses (Arm Optimized Routines sincosf: reduce by pi/2 in double, 8th/7th
// order minimax polynomials in double, one final rounding to float), written with plain double
// +,* so that host and device agree bit-for-bit.  tests/test_device_math.py measures it against
// this machine's glibc over the whole input range the path can produce ([0, 2*pi]).
// Valid for |x| < 120 (the path only produces [0, 6.2832]).
ORBX_HD void sincosf_orb(float y, float* sinp, float* cosp) {
    const double hpi_inv = 0x1.45F306DC9C883p+23;   // 2/pi * 2^24
    const double hpi = 0x1.921FB54442D18p0;         // pi/2
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5;
    const double c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    double x = (double)y;
    int n = 0;
    const float ay = fabsf(y);
    if (!(ay < 0x1.921fb6p-1f)) {            // |y| >= pi/4 : reduce
        double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
    }
    const double x2 = x * x;
    // sine polynomial (odd) and cosine polynomial (even) on the reduced argument
    double ps, pc;
    {
        double x3 = x * x2;
        double t1 = s2 + x2 * s3;
        double x7 = x3 * x2;
        double s = x + x3 * s1;
        ps = s + x7 * t1;
    }
    {
        double x4 = x2 * x2;
        double t2 = c3 + x2 * c4;
        double t1 = c0 + x2 * c1;
        double x6 = x4 * x2;
        double c = t1 + x4 * c2;
        pc = c + x6 * t2;
    }
    // quadrant: n&1 swaps, signs by n&2 / (n+1)&2
    double sv = (n & 1) ? pc : ps;
    double cv = (n & 1) ? ps : pc;
    if (n & 2) sv = -sv;
    if ((n + 1) & 2) cv = -cv;
    if (n == 0 && ay < 0x1p-12f) sv = (double)y;     // glibc returns y itself for tiny |y|
    *sinp = (float)sv;
    *cosp = (float)cv;
}

// ---- cv::FAST 9/16 corner score --------------------------------------------------------------
// d[k] = v - ring[k], k = 0..15 around the Bresenham circle.  Returns the OpenCV corner score
//   max over the 16 nine-pixel arcs of min(arc of d) (darker) / min(arc of -d) (brighter), minus 1,
// i.e. the largest threshold at which the pixel is still a FAST-9 corner, or 0 when that is < tmin
// (not a corner at threshold tmin).  corner@t <=> score >= t, independent of t.
// Window-9 circular min/max built from two rounds of 3-input min/max.
ORBX_HD int fast9_score(const int d[16], int tmin) {
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        hi3[k] = imax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int dark = -1000, bright = 1000;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int lo9 = imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
        int hi9 = imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]);
        dark = imax(dark, lo9);
        bright = imin(bright, hi9);
    }
    int s = imax(dark, -bright) - 1;
    return s >= tmin ? s : 0;
}

// ---- cv::resize INTER_LINEAR 8U, one output pixel ----------------------------------------------
// s00,s01: source row sy0 at sx, sx+1; s10,s11: row sy1.  a0,a1 / b0,b1: 11-bit fixed-point weights.
ORBX_HD int resize_px(int s00, int s01, int s10, int s11, int a0, int a1, int b0, int b1) {
    int d0 = s00 * a0 + s01 * a1;
    int d1 = s10 * a0 + s11 * a1;
    return (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
}

// ---- GaussianBlur 7x7 sigma 2, 8U: fixed-point taps and final rounding --------------------------
// Taps x256: [18,34,49,55,49,34,18] (sum 257, not renormalised; KAT in tests/test_oracle_kat.py).
#define ORBX_G0 18
#define ORBX_G1 34
#define ORBX_G2 49
#define ORBX_G3 55
ORBX_HD int blur_taps7(int a, int b, int c, int d, int e, int f, int g) {
    return ORBX_G0 * (a + g) + ORBX_G1 * (b + f) + ORBX_G2 * (c + e) + ORBX_G3 * d;
}
// sum has 16 fractional bits.  ties_even = 1 for the columns OpenCV's SSE2 column filter handles
// (float accumulate + cvtps2dq), 0 for its scalar tail / non-SIMD build (half-up).  SURVEY.md A.5.
ORBX_HD int blur_round(int sum, int ties_even) {
    // half-up: (sum + 0x8000) >> 16;  ties-to-even: add 0x7FFF plus the parity of the integer part
    int q = (sum + 0x7FFF + (ties_even ? ((sum >> 16) & 1) : 1)) >> 16;
    return q > 255 ? 255 : q;
}

// ---- 256-bit Hamming distance (host form; kernels use v_bcnt directly) -------------------------
ORBX_HD int hamming256_words(const uint32_t* a, const uint32_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        dist += __builtin_popcount(a[i] ^ b[i]);
    }
    return dist;
}

}  // namespace orbx
