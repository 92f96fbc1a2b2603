/* Synthetic grayscale frame generator (bench/test input only; pure host C, no HIP).
 *
 * Exact-integer PRNG so that every machine produces identical bytes:
 *   xorshift64*:  s ^= s>>12; s ^= s<<25; s ^= s>>27; out = s * 0x2545F4914F6CDD1D
 *   seed = 0x9E3779B97F4A7C15 ^ (frame_index + 1)
 * Families:
 *   0 S-noise  : iid bytes (out>>56)               -> saturates every quota
 *   1 S-blocks : ramp background (x+2y)/8 mod 256, (w*h)/1536 random axis-aligned rectangles
 *                (8..96 px, gray from PRNG, drawn in order), then noise (out>>61)-4 clamped
 *                -> realistic corner density, threshold-7 fallback, empty cells      (PRIMARY)
 *   2 S-flat   : constant 128                      -> N = 0
 *   3 S-lowtex : ramp + noise (out>>59)-16 in [-16,15] -> few corners at th 20, many at 7 (fallback cells)
 *   4 S-midtex : ramp background, (w*h)/256 small rectangles (6..37 px) drawn in order, three of four as a LOW-contrast step of
 *                +-(8..19) gray levels on what lies under them, one of four with a PRNG gray (high contrast), then noise
 *                (out>>60)-8 scaled to +-6 -> several times more corners at th 7 than at th 20 while most cells keep more than 3
 *                corners at th 20 (no fallback): the regime of textured real imagery
 *   5 S-warp   : CORRELATED stream (round 6).  Frames come in sequences of 64 (sequence = index / 64, t = index % 64).  A sequence has one
 *                base texture — the S-blocks recipe without its noise, seeded by the sequence, on a canvas 128 px wider on every side —
 *                and frame t is that canvas seen through an exact-integer affine map (nearest neighbour): the camera pans (t, t/2) px,
 *                rolls 23 t / 16384 rad (0.08 deg per frame) and zooms 1 + t / 1024 about the image centre, plus per-frame noise
 *                (out>>61)-4.  Consecutive frames show the same corners a pixel or two apart: what frame-to-frame matching, the accept
 *                rule and the rotation histogram of the reference see from a real camera (S-blocks frames are independent images).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

static inline uint64_t xs_next(uint64_t* s) {
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

#define WARP_SEQ 64
#define WARP_MARGIN 128
static void blocks_texture(uint8_t* out, int w, int h, ptrdiff_t stride, uint64_t* s) {      /* S-blocks without its noise */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[y * stride + x] = (uint8_t)(((x + 2 * y) / 8) & 255);
    int nrect = (int)(((int64_t)w * h) / 1536);
    if (nrect < 1) nrect = 1;
    for (int r = 0; r < nrect; r++) {
        uint64_t v = xs_next(s);
        int x0 = (int)((v & 0xFFFF) % (uint64_t)w);
        int y0 = (int)(((v >> 16) & 0xFFFF) % (uint64_t)h);
        int rw = 8 + (int)(((v >> 32) & 0xFF) % 89);
        int rh = 8 + (int)(((v >> 40) & 0xFF) % 89);
        uint8_t g = (uint8_t)(v >> 56);
        int x1 = x0 + rw > w ? w : x0 + rw, y1 = y0 + rh > h ? h : y0 + rh;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) out[y * stride + x] = g;
    }
}
static void warp_frame(uint8_t* out, int w, int h, ptrdiff_t stride, uint64_t frame_index) {
    /* the sequence's canvas is kept between calls (a caller walks a sequence frame after frame); one cache per thread */
    static _Thread_local uint8_t* canvas = 0;
    static _Thread_local int cw = 0, chh = 0;
    static _Thread_local uint64_t cseq = ~0ULL;
    const uint64_t seq = frame_index / WARP_SEQ;
    const int t = (int)(frame_index % WARP_SEQ);
    const int W = w + 2 * WARP_MARGIN, H = h + 2 * WARP_MARGIN;
    if (!canvas || cw != W || chh != H || cseq != seq) {
        if (cw != W || chh != H) { free(canvas); canvas = (uint8_t*)malloc((size_t)W * H); cw = W; chh = H; }
        uint64_t sb = 0x9E3779B97F4A7C15ULL ^ (0x5EC0000000000000ULL + seq + 1);
        blocks_texture(canvas, W, H, W, &sb);
        cseq = seq;
    }
    /* source = centre + M (p - centre) + pan, M = zoom * [A -B; B A] in 2^-14 units: B = 23 t, A = 16384 - B^2 / 32768 (cos to second order) */
    const int64_t B0 = 23 * t, A0 = 16384 - (B0 * B0) / 32768, Z = 1024 + t;
    const int64_t A = A0 * Z / 1024, Bq = B0 * Z / 1024;
    const int64_t BIAS = (int64_t)1 << 40;                       /* keeps the shifted sums positive: the same floor on every machine */
    uint64_t s = 0x9E3779B97F4A7C15ULL ^ (frame_index + 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int64_t dx = 2 * x - w + 1, dy = 2 * y - h + 1;          /* twice the offset from the image centre */
            int64_t u = ((A * dx - Bq * dy + BIAS) >> 15) - (BIAS >> 15) + (W / 2) + t;
            int64_t v = ((Bq * dx + A * dy + BIAS) >> 15) - (BIAS >> 15) + (H / 2) + t / 2;
            u = u < 0 ? 0 : u >= W ? W - 1 : u;
            v = v < 0 ? 0 : v >= H ? H - 1 : v;
            const int n = (int)(xs_next(&s) >> 61) - 4;
            out[y * stride + x] = clamp_u8(canvas[v * W + u] + n);
        }
}

void synth_frame(uint8_t* out, int w, int h, ptrdiff_t stride, int family, uint64_t frame_index) {
    uint64_t s = 0x9E3779B97F4A7C15ULL ^ (frame_index + 1);
    if (family == 5) { warp_frame(out, w, h, stride, frame_index); return; }
    if (family == 0) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) out[y * stride + x] = (uint8_t)(xs_next(&s) >> 56);
        return;
    }
    if (family == 2) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) out[y * stride + x] = 128;
        return;
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[y * stride + x] = (uint8_t)(((x + 2 * y) / 8) & 255);
    if (family == 4) {
        int nrect = (int)(((int64_t)w * h) / 256);
        if (nrect < 1) nrect = 1;
        for (int r = 0; r < nrect; r++) {
            uint64_t v = xs_next(&s);
            int x0 = (int)((v & 0xFFFF) % (uint64_t)w);
            int y0 = (int)(((v >> 16) & 0xFFFF) % (uint64_t)h);
            int rw = 6 + (int)(((v >> 32) & 0xFF) % 32);
            int rh = 6 + (int)(((v >> 40) & 0xFF) % 32);
            int hi = ((v >> 48) & 3) == 0;                                  /* one rectangle in four: a PRNG gray */
            int step = 8 + (int)(((v >> 50) & 0x1F) % 12);                  /* the others: +-(8..19) on what is there */
            if ((v >> 55) & 1) step = -step;
            uint8_t g = (uint8_t)(v >> 56);
            int x1 = x0 + rw > w ? w : x0 + rw, y1 = y0 + rh > h ? h : y0 + rh;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) out[y * stride + x] = hi ? g : clamp_u8(out[y * stride + x] + step);
        }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int n = ((int)(xs_next(&s) >> 60) - 8) * 3 / 4;              /* -6 .. 5 */
                out[y * stride + x] = clamp_u8(out[y * stride + x] + n);
            }
    } else if (family == 1) {
        int nrect = (int)(((int64_t)w * h) / 1536);
        if (nrect < 1) nrect = 1;
        for (int r = 0; r < nrect; r++) {
            uint64_t v = xs_next(&s);
            int x0 = (int)((v & 0xFFFF) % (uint64_t)w);
            int y0 = (int)(((v >> 16) & 0xFFFF) % (uint64_t)h);
            int rw = 8 + (int)(((v >> 32) & 0xFF) % 89);
            int rh = 8 + (int)(((v >> 40) & 0xFF) % 89);
            uint8_t g = (uint8_t)(v >> 56);
            int x1 = x0 + rw > w ? w : x0 + rw, y1 = y0 + rh > h ? h : y0 + rh;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) out[y * stride + x] = g;
        }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int n = (int)(xs_next(&s) >> 61) - 4;
                out[y * stride + x] = clamp_u8(out[y * stride + x] + n);
            }
    } else {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int n = (int)(xs_next(&s) >> 59) - 16;
                out[y * stride + x] = clamp_u8(out[y * stride + x] + n);
            }
    }
}

/* nframes frames, tightly packed (stride = w), frame i uses index first_index + i */
void synth_frames(uint8_t* out, int w, int h, int family, uint64_t first_index, int nframes) {
    for (int i = 0; i < nframes; i++) synth_frame(out + (size_t)i * w * h, w, h, w, family, first_index + (uint64_t)i);
}

/* 256-bit descriptors: 32 PRNG bytes each (seed as above with index = desc_seed) */
void synth_descriptors(uint8_t* out, int n, uint64_t desc_seed) {
    uint64_t s = 0x9E3779B97F4A7C15ULL ^ (desc_seed + 1);
    for (size_t i = 0; i < (size_t)n * 4; i++) {
        uint64_t v = xs_next(&s);
        for (int b = 0; b < 8; b++) out[i * 8 + b] = (uint8_t)(v >> (8 * b));
    }
}
