// =====================================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing in the product path (orb_slam_amd/, include/)
// may include, link, import or execute this file.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg use it, and only as the checker / CPU baseline.
//
// What it is: a scalar CPU restatement of the reference's per-frame ORB front-end,
//   /root/reference/src/ORBextractor.cc   (operator(), ComputePyramid, ComputeKeyPoints,
//                                          HarrisResponses, IC_Angle, computeOrbDescriptor)
//   /root/reference/src/ORBmatcher.cc     (DescriptorDistance + the best/second-best scan)
// plus a restatement of the OpenCV-2.4 primitives the reference delegates its pixel
// arithmetic to (cv::resize INTER_LINEAR 8U, copyMakeBorder REFLECT_101, cv::FAST 9/16 with
// NMS, KeyPointsFilter::retainBest, GaussianBlur 7x7 sigma 2 for 8U, fastAtan2, cvRound).
// OpenCV "tested 2.4" (reference README.md:56) is NOT vendored in the reference tree and is
// not installed here, so those primitives are restated from the published OpenCV 2.4.x
// algorithms (modules/imgproc/src/imgwarp.cpp, smooth.cpp, filter.cpp; modules/features2d/
// src/fast.cpp, fast_score.cpp, keypoint.cpp; modules/core/src/copy.cpp, mathfuncs.cpp).
//
// PARITY — what is pinned and what is not:
//  * PINNED (by executing the reference's source): everything ORB_SLAM computes itself.  oracle/Makefile compiles
//    /root/reference/src/ORBextractor.cc where it lies against oracle/cvstub (stand-in OpenCV headers) into
//    oracle/_ref/libref_orbextractor.so; tests/test_ref_pin.py requires this restatement to reproduce its
//    keypoints (order, coordinates, size, angle bits, response, octave) and descriptors byte for byte.
//  * PARITY UNPINNED for the OpenCV 2.4 pixel primitives (resize, FAST+NMS, GaussianBlur, retainBest,
//    fastAtan2): OpenCV is not vendored in the reference and not installed here, the reference ships no
//    tests, golden vectors or fixtures, and behind the stand-in headers those calls
//    resolve to the restatements below.  They are pinned only by known-answer tests
//    (tests/test_oracle_kat.py) and this oracle's own committed golden fixtures (tests/golden/).
// Each function cites the reference lines it follows.
//
// Float discipline: built with -ffp-contract=off (ISO evaluation, no FMA fusion); the
// reference's own -O3 -march=native build may fuse differently per CPU.
// cos/sin are glibc cosf/sinf, as `cos(float)` resolves in the reference (using namespace std).
// =====================================================================================
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

using std::ptrdiff_t;
using std::size_t;

namespace {

// ----------------------------------------------------------------------------- OpenCV scalars
// cvRound: round-half-to-even (SSE2 cvtsd2si / lrint under the default rounding mode).
inline int cvRound(double v) { return (int)lrint(v); }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
inline short saturate_short(float v) {
    int iv = cvRound(v);
    return (short)((unsigned)(iv - SHRT_MIN) <= (unsigned)USHRT_MAX ? iv : iv > 0 ? SHRT_MAX : SHRT_MIN);
}
inline uint8_t saturate_u8(int v) { return (uint8_t)((unsigned)v <= 255u ? v : v > 0 ? 255 : 0); }

// cv::KeyPoint, OpenCV 2.4 field layout (28 bytes).
struct KeyPoint {
    float x, y;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
};

// cv::fastAtan2 (OpenCV 2.4.x mathfuncs.cpp), degrees in [0,360).  SURVEY.md A.6.
const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);
float fastAtan2(float y, float x) {
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cv::borderInterpolate for BORDER_REFLECT_101.  SURVEY.md A.4.
inline int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

const int EDGE_THRESHOLD = 16;   // reference src/ORBextractor.cc:77
const int PATCH_SIZE = 31;       // :75
const int HALF_PATCH_SIZE = 15;  // :76
const float HARRIS_K = 0.04f;    // :73

// One pyramid level: a (w+32)x(h+32) buffer with the level as the centred ROI, exactly the
// `temp(Rect(16,16,w,h))` construction of reference ComputePyramid (:786-789).
struct Level {
    int w = 0, h = 0, stride = 0;
    std::vector<uint8_t> buf;
    uint8_t* roi() { return buf.data() + EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
    const uint8_t* roi() const { return buf.data() + EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
    void alloc(int w_, int h_) {
        w = w_; h = h_; stride = w + 2 * EDGE_THRESHOLD;
        buf.assign((size_t)stride * (h + 2 * EDGE_THRESHOLD), 0);
    }
};

// cv::copyMakeBorder(roi, whole, 16,16,16,16, BORDER_REFLECT_101[+ISOLATED]) with roi inside whole.
void make_border_reflect101(Level& L) {
    uint8_t* r = L.roi();
    const int B = EDGE_THRESHOLD;
    for (int y = 0; y < L.h; y++) {
        uint8_t* row = r + (ptrdiff_t)y * L.stride;
        for (int x = -B; x < 0; x++) row[x] = row[reflect101(x, L.w)];
        for (int x = L.w; x < L.w + B; x++) row[x] = row[reflect101(x, L.w)];
    }
    for (int y = -B; y < L.h + B; y++) {
        if (y >= 0 && y < L.h) continue;
        int sy = reflect101(y, L.h);
        memcpy(r + (ptrdiff_t)y * L.stride - B, r + (ptrdiff_t)sy * L.stride - B, L.w + 2 * B);
    }
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 (imgwarp.cpp: fixed-point
// path, INTER_RESIZE_COEF_BITS=11).  SURVEY.md A.2, with scale = 1./((double)dsize/ssize)
// as the 2.4 source computes it (inv_scale first, then its reciprocal).
void resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = saturate_short((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = saturate_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = saturate_short((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = saturate_short(fy * 2048);
    }
    std::vector<int> rows[2];
    rows[0].resize(dw); rows[1].resize(dw);
    for (int dy = 0; dy < dh; dy++) {
        for (int k = 0; k < 2; k++) {
            int sy = yofs[dy] + k;
            sy = sy < 0 ? 0 : sy >= sh ? sh - 1 : sy;   // clip(sy, 0, ssize.height)
            const uint8_t* S = src + (ptrdiff_t)sy * sstride;
            int* D = rows[k].data();
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
                // for sx == sw-1 the weight a1 is 0 and OpenCV never reads S[sx+1]
                D[dx] = S[sx] * a0 + (a1 ? S[sx + 1] * a1 : 0);
            }
        }
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* out = dst + (ptrdiff_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            out[x] = (uint8_t)((((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2);
    }
}

// ------------------------------------------------------------------------------- cv::FAST
// Ring offsets of the 16-pixel Bresenham circle, OpenCV fast_score.cpp makeOffsets(16).
const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

// cornerScore<16> (fast_score.cpp, scalar form).
int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int N = 25;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// cv::FAST(img, keypoints, threshold, nonmaxSuppression=true) on an ROI view (fast.cpp FAST_t<16>).
// Scans rows 3..rows-4, cols 3..cols-4 of the given view; NMS neighbours outside the scanned
// area count as 0; output in raster order.  SURVEY.md A.3.  `scores_out` (optional, rows*cols)
// receives the stored score map for stage dumps.
void cv_FAST(const uint8_t* img, int cols, int rows, int step, std::vector<KeyPoint>& kps, int threshold,
             std::vector<uint8_t>* scores_out = nullptr) {
    kps.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = RING_DX[k] + RING_DY[k] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    std::vector<uint8_t> score((size_t)std::max(rows, 0) * std::max(cols, 0), 0), iscorner(score.size(), 0);
    const int K = 8, N = 25;
    for (int i = 3; i < rows - 3; i++) {
        for (int j = 3; j < cols - 3; j++) {
            const uint8_t* ptr = img + (ptrdiff_t)i * step + j;
            int v = ptr[0];
            // OpenCV's table-driven quick reject (fast.cpp): a 9-arc must contain one pixel of every
            // opposite pair {k,k+8}; d bit0 = "darker than v-t", bit1 = "brighter than v+t".
            auto cls = [&](int k) { int x = ptr[pixel[k]]; return (x < v - threshold ? 1 : 0) | (x > v + threshold ? 2 : 0); };
            int d = cls(0) | cls(8);
            if (d == 0) continue;
            d &= cls(2) | cls(10);
            d &= cls(4) | cls(12);
            d &= cls(6) | cls(14);
            if (d == 0) continue;
            d &= cls(1) | cls(9);
            d &= cls(3) | cls(11);
            d &= cls(5) | cls(13);
            d &= cls(7) | cls(15);
            bool corner = false;
            if (d & 1) {   // darker arc
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] < vt) { if (++count > K) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (!corner && (d & 2)) {  // brighter arc
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (corner) {
                iscorner[(size_t)i * cols + j] = 1;
                score[(size_t)i * cols + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
            }
        }
    }
    for (int i = 3; i < rows - 3; i++) {
        for (int j = 3; j < cols - 3; j++) {
            size_t o = (size_t)i * cols + j;
            if (!iscorner[o]) continue;
            int s = score[o];
            // rows i-1/i+1 and cols j-1/j+1 are always inside the buffer (i,j >= 3)
            if (s > score[o - 1] && s > score[o + 1] && s > score[o - cols - 1] && s > score[o - cols] &&
                s > score[o - cols + 1] && s > score[o + cols - 1] && s > score[o + cols] && s > score[o + cols + 1]) {
                KeyPoint kp = {(float)j, (float)i, 7.f, -1.f, (float)s, 0, -1};
                kps.push_back(kp);
            }
        }
    }
    if (scores_out) *scores_out = score;
}

// KeyPointsFilter::retainBest (features2d/src/keypoint.cpp).  SURVEY.md A.7.
struct ResponseGreater {
    bool operator()(const KeyPoint& a, const KeyPoint& b) const { return a.response > b.response; }
};
void retainBest(std::vector<KeyPoint>& v, int n) {
    if (n >= 0 && v.size() > (size_t)n) {
        if (n == 0) { v.clear(); return; }
        std::nth_element(v.begin(), v.begin() + n, v.end(), ResponseGreater());
        float amb = v[n - 1].response;
        std::vector<KeyPoint>::iterator new_end =
            std::partition(v.begin() + n, v.end(), [amb](const KeyPoint& k) { return k.response >= amb; });
        v.resize(new_end - v.begin());
    }
}

// reference src/ORBextractor.cc:79-120 (HarrisResponses), `img` = the cell view inside the level.
// fp_contract: the reference's own build (CMakeLists.txt:12-13: -O3 -march=native, GCC's default -ffp-contract=fast) fuses two
// multiply-adds of the last expression (read off its object code: oracle/Makefile _ref_native):
//   t = fma(a, b, -(c*c));  response = fma(-(a+b), k*(a+b), t) * scale^4
void HarrisResponses(const uint8_t* ptr00, int step, std::vector<KeyPoint>& pts, int blockSize, float harris_k, bool fp_contract = false) {
    int r = blockSize / 2;
    float scale = (1 << 2) * blockSize * 255.0f;
    scale = 1.0f / scale;
    float scale_sq_sq = scale * scale * scale * scale;
    for (size_t p = 0; p < pts.size(); p++) {
        int x0 = cvRound(pts[p].x - r);
        int y0 = cvRound(pts[p].y - r);
        const uint8_t* ptr0 = ptr00 + (ptrdiff_t)y0 * step + x0;
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < blockSize; i++)
            for (int j = 0; j < blockSize; j++) {
                const uint8_t* ptr = ptr0 + i * step + j;
                int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
                int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
        if (fp_contract) {
            const float cc = (float)c * c, sum = (float)a + b;
            const float t = fmaf((float)a, (float)b, -cc);
            pts[p].response = fmaf(-sum, harris_k * sum, t) * scale_sq_sq;
        } else
        pts[p].response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }
}

// reference :124-151 (IC_Angle)
float IC_Angle(const uint8_t* roi, int step, float ptx, float pty, const std::vector<int>& u_max) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = roi + (ptrdiff_t)cvRound(pty) * step + cvRound(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
}

struct Point { int x, y; };
const Point bit_pattern_31[512] = {
#include "orb_pattern_points.inc"
};

// reference :154-194 (computeOrbDescriptor).  cos/sin of a float resolve to cosf/sinf.
const float factorPI = (float)(M_PI / 180.f);
// fp_contract: the reference's own build fuses `x*b + y*a` into fma(x, b, y*a) and `x*a - y*b` into fma(x, a, -(y*b)) (see HarrisResponses).
void computeOrbDescriptor(const KeyPoint& kpt, const uint8_t* roi, int step, const Point* pattern, uint8_t* desc, bool fp_contract = false) {
    float angle = (float)kpt.angle * factorPI;
    float a = (float)cosf(angle), b = (float)sinf(angle);
    const uint8_t* center = roi + (ptrdiff_t)cvRound(kpt.y) * step + cvRound(kpt.x);
#define GET_VALUE(idx) \
    (fp_contract ? center[cvRound(fmaf((float)pattern[idx].x, b, pattern[idx].y * a)) * step + cvRound(fmaf((float)pattern[idx].x, a, -(pattern[idx].y * b)))] \
                 : center[cvRound(pattern[idx].x * b + pattern[idx].y * a) * step + cvRound(pattern[idx].x * a - pattern[idx].y * b)])
    for (int i = 0; i < 32; ++i, pattern += 16) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = GET_VALUE(2 * k), t1 = GET_VALUE(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
}

// GaussianBlur(roi, roi, Size(7,7), 2, 2, BORDER_REFLECT_101) for 8U, in place on an ROI whose
// 16-px surround already holds its reflect-101 border.  SURVEY.md A.5.
//   kernel: getGaussianKernel(7, 2, CV_32F) -> x256, cvRound -> int (sum 257, not renormalised)
//   row pass 8U->32S exact; column pass 32S->8U with 16 fractional bits:
//   blur_mode 0 (default, "x86 SSE2 emulation"): columns x < (w & ~3) are rounded to nearest-EVEN
//       (OpenCV's SymmColumnVec_32s8u: float accumulate + cvtps2dq), the last w%4 columns half-up
//       (scalar FixedPtCastEx tail);  blur_mode 1: half-up everywhere (non-SIMD build).
void gaussian_kernel_q8(int k[7]) {
    const int n = 7;
    const double sigma = 2.0;
    double scale2X = -0.5 / (sigma * sigma);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = cvRound(cf[i] * 256.f);
    }
}
void gaussian_blur7_inplace(Level& L, int blur_mode) {
    int k[7];
    gaussian_kernel_q8(k);
    const int w = L.w, h = L.h, st = L.stride;
    std::vector<int> rowsum((size_t)(h + 6) * w);
    const uint8_t* r = L.roi();
    for (int y = -3; y < h + 3; y++) {
        const uint8_t* p = r + (ptrdiff_t)y * st;
        int* o = rowsum.data() + (size_t)(y + 3) * w;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * p[x + i - 3];
            o[x] = s;
        }
    }
    std::vector<uint8_t> out((size_t)w * h);
    const int wvec = (blur_mode == 0) ? (w & ~3) : 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int j = 0; j < 7; j++) s += k[j] * rowsum[(size_t)(y + j) * w + x];
            int q;
            if (x < wvec) {
                q = s >> 16;
                int rem = s & 0xFFFF;
                if (rem > 0x8000 || (rem == 0x8000 && (q & 1))) q++;
            } else {
                q = (s + 0x8000) >> 16;
            }
            out[(size_t)y * w + x] = saturate_u8(q);
        }
    uint8_t* wr = L.roi();
    for (int y = 0; y < h; y++) memcpy(wr + (ptrdiff_t)y * st, out.data() + (size_t)y * w, w);
}

// ---------------------------------------------------------------------------- ORBextractor
struct CellDump {           // stage dump: what cv::FAST returned for one grid cell (cell-local coords)
    int level, row, col, iniX, iniY, used_fallback, cw, ch;
    std::vector<KeyPoint> kps;
};

struct Extractor {
    // reference include/ORBextractor.h:59-75
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int scoreType;
    int fastTh;
    int blur_mode;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor;
    std::vector<Level> pyr;          // mvImagePyramid (unblurred until operator() blurs in place)
    std::vector<Level> pyr_plain;    // stage dump: copy of the unblurred pyramid
    std::vector<std::vector<KeyPoint>> level_kps;   // stage dump: allKeypoints (level coords, with angle)
    std::vector<CellDump> cells;     // stage dump
    bool keep_dumps = false;
    bool fp_contract = false;        // orc_set_fp_contract
    int error = 0;

    // reference :457-511
    Extractor(int nf, float sf, int nl, int st, int ft, int bm)
        : nfeatures(nf), scaleFactor(sf), nlevels(nl), scoreType(st), fastTh(ft), blur_mode(bm) {
        mvScaleFactor.resize(nlevels);
        mvScaleFactor[0] = 1;
        for (int i = 1; i < nlevels; i++) mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
        float invScaleFactor = 1.0f / scaleFactor;
        mvInvScaleFactor.resize(nlevels);
        mvInvScaleFactor[0] = 1;
        for (int i = 1; i < nlevels; i++) mvInvScaleFactor[i] = mvInvScaleFactor[i - 1] * invScaleFactor;
        pyr.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0 / scaleFactor);
        float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    // reference :781-822 (mask pyramid omitted: it is built but never consumed)
    void ComputePyramid(const uint8_t* image, int cols, int rows, int step) {
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            int sw = cvRound((float)cols * scale), sh = cvRound((float)rows * scale);
            Level& L = pyr[level];
            L.alloc(sw, sh);
            if (level != 0) {
                const Level& P = pyr[level - 1];
                resize_linear_8u(P.roi(), P.w, P.h, P.stride, L.roi(), sw, sh, L.stride);
            } else {
                for (int y = 0; y < rows; y++) memcpy(L.roi() + (ptrdiff_t)y * L.stride, image + (ptrdiff_t)y * step, cols);
            }
            make_border_reflect101(L);
        }
    }

    // reference :522-707
    void ComputeKeyPoints(std::vector<std::vector<KeyPoint>>& allKeypoints) {
        allKeypoints.assign(nlevels, std::vector<KeyPoint>());
        float imageRatio = (float)pyr[0].w / pyr[0].h;
        for (int level = 0; level < nlevels; ++level) {
            const int nDesiredFeatures = mnFeaturesPerLevel[level];
            const int levelCols = std::sqrt((float)nDesiredFeatures / (5 * imageRatio));
            const int levelRows = imageRatio * levelCols;
            if (levelCols <= 0 || levelRows <= 0) { error = -3; return; }   // reference divides by zero here
            const int minBorderX = EDGE_THRESHOLD;
            const int minBorderY = minBorderX;
            const int maxBorderX = pyr[level].w - EDGE_THRESHOLD;
            const int maxBorderY = pyr[level].h - EDGE_THRESHOLD;
            const int W = maxBorderX - minBorderX;
            const int H = maxBorderY - minBorderY;
            const int cellW = std::ceil((float)W / levelCols);
            const int cellH = std::ceil((float)H / levelRows);
            const int nCells = levelRows * levelCols;
            const int nfeaturesCell = std::ceil((float)nDesiredFeatures / nCells);

            std::vector<std::vector<std::vector<KeyPoint>>> cellKeyPoints(levelRows, std::vector<std::vector<KeyPoint>>(levelCols));
            std::vector<std::vector<int>> nToRetain(levelRows, std::vector<int>(levelCols));
            std::vector<std::vector<int>> nTotal(levelRows, std::vector<int>(levelCols));
            std::vector<std::vector<bool>> bNoMore(levelRows, std::vector<bool>(levelCols, false));
            std::vector<int> iniXCol(levelCols);
            std::vector<int> iniYRow(levelRows);
            int nNoMore = 0;
            int nToDistribute = 0;

            float hY = cellH + 6;
            for (int i = 0; i < levelRows; i++) {
                const float iniY = minBorderY + i * cellH - 3;
                iniYRow[i] = iniY;
                if (i == levelRows - 1) {
                    hY = maxBorderY + 3 - iniY;
                    if (hY <= 0) continue;
                }
                float hX = cellW + 6;
                for (int j = 0; j < levelCols; j++) {
                    float iniX;
                    if (i == 0) {
                        iniX = minBorderX + j * cellW - 3;
                        iniXCol[j] = iniX;
                    } else {
                        iniX = iniXCol[j];
                    }
                    if (j == levelCols - 1) {
                        hX = maxBorderX + 3 - iniX;
                        if (hX <= 0) continue;
                    }
                    // Mat::rowRange/colRange assert 0 <= start <= end <= size (cv::Exception in the reference)
                    if (iniY < 0 || iniX < 0 || iniY + hY > pyr[level].h || iniX + hX > pyr[level].w) { error = -3; return; }
                    const uint8_t* cellImage = pyr[level].roi() + (ptrdiff_t)(int)iniY * pyr[level].stride + (int)iniX;
                    const int cw = (int)(iniX + hX) - (int)iniX, ch = (int)(iniY + hY) - (int)iniY;
                    std::vector<KeyPoint>& ck = cellKeyPoints[i][j];
                    cv_FAST(cellImage, cw, ch, pyr[level].stride, ck, fastTh);
                    int fb = 0;
                    if (ck.size() <= 3) {
                        ck.clear();
                        cv_FAST(cellImage, cw, ch, pyr[level].stride, ck, 7);
                        fb = 1;
                    }
                    if (scoreType == 0 /*HARRIS_SCORE*/) HarrisResponses(cellImage, pyr[level].stride, ck, 7, HARRIS_K, fp_contract);
                    if (keep_dumps) {
                        CellDump cd = {level, i, j, (int)iniX, (int)iniY, fb, cw, ch, ck};
                        cells.push_back(cd);
                    }
                    const int nKeys = ck.size();
                    nTotal[i][j] = nKeys;
                    if (nKeys > nfeaturesCell) {
                        nToRetain[i][j] = nfeaturesCell;
                        bNoMore[i][j] = false;
                    } else {
                        nToRetain[i][j] = nKeys;
                        nToDistribute += nfeaturesCell - nKeys;
                        bNoMore[i][j] = true;
                        nNoMore++;
                    }
                }
            }
            while (nToDistribute > 0 && nNoMore < nCells) {
                int nNewFeaturesCell = nfeaturesCell + std::ceil((float)nToDistribute / (nCells - nNoMore));
                nToDistribute = 0;
                for (int i = 0; i < levelRows; i++)
                    for (int j = 0; j < levelCols; j++)
                        if (!bNoMore[i][j]) {
                            if (nTotal[i][j] > nNewFeaturesCell) {
                                nToRetain[i][j] = nNewFeaturesCell;
                                bNoMore[i][j] = false;
                            } else {
                                nToRetain[i][j] = nTotal[i][j];
                                nToDistribute += nNewFeaturesCell - nTotal[i][j];
                                bNoMore[i][j] = true;
                                nNoMore++;
                            }
                        }
            }
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            keypoints.reserve(nDesiredFeatures * 2);
            const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level];
            for (int i = 0; i < levelRows; i++)
                for (int j = 0; j < levelCols; j++) {
                    std::vector<KeyPoint>& keysCell = cellKeyPoints[i][j];
                    retainBest(keysCell, nToRetain[i][j]);
                    if ((int)keysCell.size() > nToRetain[i][j]) keysCell.resize(nToRetain[i][j]);
                    for (size_t k = 0, kend = keysCell.size(); k < kend; k++) {
                        keysCell[k].x += iniXCol[j];
                        keysCell[k].y += iniYRow[i];
                        keysCell[k].octave = level;
                        keysCell[k].size = scaledPatchSize;
                        keypoints.push_back(keysCell[k]);
                    }
                }
            if ((int)keypoints.size() > nDesiredFeatures) {
                retainBest(keypoints, nDesiredFeatures);
                keypoints.resize(nDesiredFeatures);
            }
        }
        for (int level = 0; level < nlevels; ++level)
            for (KeyPoint& kp : allKeypoints[level])
                kp.angle = IC_Angle(pyr[level].roi(), pyr[level].stride, kp.x, kp.y, umax);
    }

    // reference :718-779.  Returns N (>=0) or <0 on error; -1 = empty image (outputs untouched).
    int extract(const uint8_t* image, int cols, int rows, int step, KeyPoint* out_kps, uint8_t* out_desc, int cap) {
        error = 0;
        if (!image || cols <= 0 || rows <= 0) return -1;
        cells.clear();
        ComputePyramid(image, cols, rows, step);
        if (keep_dumps) pyr_plain = pyr;
        std::vector<std::vector<KeyPoint>> allKeypoints;
        ComputeKeyPoints(allKeypoints);
        if (error) return error;
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
        if (nkeypoints > cap) return -2;
        if (keep_dumps) level_kps = allKeypoints;
        int offset = 0;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            int n = (int)keypoints.size();
            if (n == 0) continue;
            gaussian_blur7_inplace(pyr[level], blur_mode);
            for (int i = 0; i < n; i++)
                computeOrbDescriptor(keypoints[i], pyr[level].roi(), pyr[level].stride, bit_pattern_31, out_desc + (size_t)(offset + i) * 32, fp_contract);
            if (level != 0) {
                float scale = mvScaleFactor[level];
                for (KeyPoint& kp : keypoints) { kp.x *= scale; kp.y *= scale; }
            }
            memcpy(out_kps + offset, keypoints.data(), sizeof(KeyPoint) * n);
            offset += n;
        }
        return nkeypoints;
    }
};

// reference src/ORBmatcher.cc:1794-1810 (DescriptorDistance), bithack form kept verbatim in spirit.
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

}  // namespace

// =================================================================================== C API
extern "C" {

typedef KeyPoint orc_keypoint;

void* orc_create(int nfeatures, float scaleFactor, int nlevels, int scoreType, int fastTh, int blur_mode) {
    if (nlevels < 1 || nfeatures < 0) return nullptr;
    return new Extractor(nfeatures, scaleFactor, nlevels, scoreType, fastTh, blur_mode);
}
void orc_destroy(void* h) { delete (Extractor*)h; }
void orc_keep_dumps(void* h, int on) { ((Extractor*)h)->keep_dumps = on != 0; }
// 1: the float expressions of HarrisResponses / computeOrbDescriptor as the reference's own build flags contract them (default 0: ISO evaluation)
void orc_set_fp_contract(void* h, int on) { ((Extractor*)h)->fp_contract = on != 0; }

int orc_extract(void* h, const uint8_t* img, int w, int hh, int stride, orc_keypoint* kps, uint8_t* desc, int cap) {
    return ((Extractor*)h)->extract(img, w, hh, stride, kps, desc, cap);
}

// constructor tables (E1)
int orc_features_per_level(void* h, int level) { return ((Extractor*)h)->mnFeaturesPerLevel[level]; }
float orc_scale_factor(void* h, int level) { return ((Extractor*)h)->mvScaleFactor[level]; }
float orc_inv_scale_factor(void* h, int level) { return ((Extractor*)h)->mvInvScaleFactor[level]; }
int orc_umax(void* h, int v) { return ((Extractor*)h)->umax[v]; }

// stage dumps (valid after orc_extract with keep_dumps on)
int orc_level_size(void* h, int level, int* w, int* hh) {
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= (int)e->pyr.size()) return -1;
    *w = e->pyr[level].w; *hh = e->pyr[level].h;
    return 0;
}
// which: 0 = unblurred ROI (w*h), 1 = ROI after operator() (blurred where the level had keypoints),
//        2 = unblurred padded plane ((w+32)*(h+32))
int orc_level_plane(void* h, int level, int which, uint8_t* out) {
    Extractor* e = (Extractor*)h;
    const Level& L = (which == 1) ? e->pyr[level] : e->pyr_plain[level];
    if (which == 2) { memcpy(out, L.buf.data(), L.buf.size()); return 0; }
    for (int y = 0; y < L.h; y++) memcpy(out + (size_t)y * L.w, L.roi() + (ptrdiff_t)y * L.stride, L.w);
    return 0;
}
int orc_level_keypoints(void* h, int level, orc_keypoint* out, int cap) {
    Extractor* e = (Extractor*)h;
    int n = (int)e->level_kps[level].size();
    if (n > cap) return -2;
    memcpy(out, e->level_kps[level].data(), sizeof(KeyPoint) * n);
    return n;
}
int orc_num_cells(void* h) { return (int)((Extractor*)h)->cells.size(); }
// info[8] = {level,row,col,iniX,iniY,used_fallback,view_w,view_h}; returns count (cell-local coords, raster order)
int orc_cell(void* h, int idx, int* info, orc_keypoint* out, int cap) {
    Extractor* e = (Extractor*)h;
    const CellDump& c = e->cells[idx];
    info[0] = c.level; info[1] = c.row; info[2] = c.col; info[3] = c.iniX; info[4] = c.iniY; info[5] = c.used_fallback; info[6] = c.cw; info[7] = c.ch;
    int n = (int)c.kps.size();
    if (out) { if (n > cap) return -2; memcpy(out, c.kps.data(), sizeof(KeyPoint) * n); }
    return n;
}

// primitives for known-answer tests
int orc_cvRound(double v) { return cvRound(v); }
int orc_cvFloor(double v) { return cvFloor(v); }
int orc_cvCeil(double v) { return cvCeil(v); }
float orc_fastAtan2(float y, float x) { return fastAtan2(y, x); }
int orc_reflect101(int p, int len) { return reflect101(p, len); }
void orc_gaussian_kernel_q8(int* k7) { gaussian_kernel_q8(k7); }
void orc_resize_linear_8u(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
    resize_linear_8u(src, sw, sh, sstride, dst, dw, dh, dstride);
}
// whole-view cv::FAST: returns count; kps in view coords; scores = stored score map (w*h) or NULL
int orc_fast(const uint8_t* img, int w, int hh, int stride, int threshold, orc_keypoint* out, int cap, uint8_t* scores) {
    std::vector<KeyPoint> k;
    std::vector<uint8_t> sc;
    cv_FAST(img, w, hh, stride, k, threshold, scores ? &sc : nullptr);
    if (scores) memcpy(scores, sc.data(), sc.size());
    if ((int)k.size() > cap) return -2;
    if (out && !k.empty()) memcpy(out, k.data(), sizeof(KeyPoint) * k.size());
    return (int)k.size();
}
// in-place blur of a tight w*h image (border synthesised by reflect-101, as in the pipeline)
void orc_gaussian_blur7(uint8_t* img, int w, int hh, int blur_mode) {
    Level L;
    L.alloc(w, hh);
    for (int y = 0; y < hh; y++) memcpy(L.roi() + (ptrdiff_t)y * L.stride, img + (size_t)y * w, w);
    make_border_reflect101(L);
    gaussian_blur7_inplace(L, blur_mode);
    for (int y = 0; y < hh; y++) memcpy(img + (size_t)y * w, L.roi() + (ptrdiff_t)y * L.stride, w);
}
// retainBest + resize(n) on a response list: returns the surviving ORIGINAL indices in output order
int orc_retain_best(const float* responses, int count, int n, int* out_idx) {
    std::vector<KeyPoint> v(count);
    for (int i = 0; i < count; i++) { v[i] = KeyPoint{0, 0, 0, 0, responses[i], 0, i}; }
    retainBest(v, n);
    if ((int)v.size() > n) v.resize(n);
    for (size_t i = 0; i < v.size(); i++) out_idx[i] = v[i].class_id;
    return (int)v.size();
}
// std::nth_element(v, v+nth, v+count, response greater): the full resulting permutation (original indices)
void orc_nth_element_perm(const float* responses, int count, int nth, int* out_perm) {
    std::vector<KeyPoint> v(count);
    for (int i = 0; i < count; i++) { v[i] = KeyPoint{0, 0, 0, 0, responses[i], 0, i}; }
    std::nth_element(v.begin(), v.begin() + nth, v.end(), ResponseGreater());
    for (int i = 0; i < count; i++) out_perm[i] = v[i].class_id;
}
// --- entry points used only by oracle/cvstub (the reference's own sources compiled against stand-in cv headers)
// KeyPointsFilter::retainBest in place on a keypoint array (ties at the boundary kept): returns the new size
int orc_retain_best_kps(orc_keypoint* kps, int count, int n) {
    std::vector<KeyPoint> v(count);
    if (count) memcpy(v.data(), kps, sizeof(KeyPoint) * count);
    retainBest(v, n);
    if (!v.empty()) memcpy(kps, v.data(), sizeof(KeyPoint) * v.size());
    return (int)v.size();
}
// in-place 7x7 blur of the centred w*h view of a tight (w+32)x(h+32) buffer; the 16-px frame is read, not written
void orc_gaussian_blur7_level(uint8_t* whole, int w, int hh, int blur_mode) {
    Level L;
    L.alloc(w, hh);
    memcpy(L.buf.data(), whole, L.buf.size());
    gaussian_blur7_inplace(L, blur_mode);
    memcpy(whole, L.buf.data(), L.buf.size());
}
void orc_sincosf(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }

// matcher (M1, M2)
int orc_hamming256(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }
// The scan shared by every ORBmatcher search (e.g. src/ORBmatcher.cc:201-222): strict '<' updates,
// so (best,second) are the two smallest distances with multiplicity and idx is the FIRST index
// attaining best.  nt==0 -> idx=-1, best=second=INT_MAX.
void orc_match_top2(const uint8_t* Q, int nq, const uint8_t* T, int nt, int32_t* best_idx, int32_t* best, int32_t* second) {
    for (int q = 0; q < nq; q++) {
        int bestDist1 = INT_MAX, bestIdx = -1, bestDist2 = INT_MAX;
        for (int t = 0; t < nt; t++) {
            const int dist = descriptor_distance(Q + (size_t)q * 32, T + (size_t)t * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = t; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        best_idx[q] = bestIdx; best[q] = bestDist1; second[q] = bestDist2;
    }
}
// CPU-baseline variant of the same scan ("a __builtin_popcountll variant"): the descriptor as four 64-bit words, one
// popcount instruction per word instead of the reference's bit-trick sum (src/ORBmatcher.cc:1794-1810).  Same integers out.
void orc_match_top2_popcountll(const uint8_t* Q, int nq, const uint8_t* T, int nt, int32_t* best_idx, int32_t* best, int32_t* second) {
    for (int q = 0; q < nq; q++) {
        uint64_t a[4];
        memcpy(a, Q + (size_t)q * 32, 32);
        int bestDist1 = INT_MAX, bestIdx = -1, bestDist2 = INT_MAX;
        for (int t = 0; t < nt; t++) {
            uint64_t b[4];
            memcpy(b, T + (size_t)t * 32, 32);
            const int dist = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) + __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = t; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        best_idx[q] = bestIdx; best[q] = bestDist1; second[q] = bestDist2;
    }
}
// the same scan over a per-query candidate list (e.g. src/ORBmatcher.cc:87-111 over vNearIndices, :201-222 over vIndicesF)
void orc_match_top2_segments(const uint8_t* Q, int nq, const uint8_t* T, int nt, const int32_t* seg_off, const int32_t* cand,
                             int32_t* best_idx, int32_t* best, int32_t* second) {
    for (int q = 0; q < nq; q++) {
        int bestDist1 = INT_MAX, bestIdx = -1, bestDist2 = INT_MAX;
        for (int p = seg_off[q]; p < seg_off[q + 1]; p++) {
            const int t = cand[p];
            if (t < 0 || t >= nt) continue;
            const int dist = descriptor_distance(Q + (size_t)q * 32, T + (size_t)t * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx = t; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        best_idx[q] = bestIdx; best[q] = bestDist1; second[q] = bestDist2;
    }
}
// accept rule of SearchByBoW (src/ORBmatcher.cc:224-226): best<=th && (float)best < ratio*(float)second
int orc_count_accepted(const int32_t* best, const int32_t* second, int nq, int th, float ratio) {
    int n = 0;
    for (int q = 0; q < nq; q++)
        if (best[q] <= th && static_cast<float>(best[q]) < ratio * static_cast<float>(second[q])) n++;
    return n;
}

}  // extern "C"
