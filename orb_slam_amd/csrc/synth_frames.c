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
 */
#include <stdint.h>
#include <stddef.h>

static inline uint64_t xs_next(uint64_t* s) {
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static inline uint8_t clamp_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void synth_frame(uint8_t* out, int w, int h, ptrdiff_t stride, int family, uint64_t frame_index) {
    uint64_t s = 0x9E3779B97F4A7C15ULL ^ (frame_index + 1);
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
