// Probe: operand / result / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (E2M1) operands on gfx950 — what the FP4 matcher
// (orbm_match.hip, mfma4_scan) relies on.  FP4 magnitudes cannot carry a row number, the block scale can: with every nibble +1.0 and the E8M0
// scale of lane l set to 2^(l % 32), D[i][j] = 64 * 2^i tells which A row a result register holds (mode 0; mode 1: the same through B for the
// columns).  Mode 2: scale 1 in lanes 0-31, 2 in lanes 32-63 -> 32 * 1 + 32 * 2 = 96 everywhere iff a lane's scale applies to ITS 32 K values
// (lane half = K half).  Mode 3: A = +1.0 in the low lane half only, B = +1.0 in the high lane half only -> 0 iff A's and B's lane halves
// pair up by the same index (32 if they were crossed).  build: hipcc --offload-arch=gfx950 -O3 mfma_layout_fp4.hip -o mfma_layout_fp4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, int mode) {
    const int l = threadIdx.x, id = l % 32, hi = l / 32;
    const int ones = 0x22222222;                                   // eight +1.0 nibbles
    int av = ones, bv = ones, sa = 0x7F7F7F7F, sb = 0x7F7F7F7F;    // E8M0 127 = 2^0 in every byte
    if (mode == 0) sa = (127 + id) * 0x01010101;
    if (mode == 1) sb = (127 + id) * 0x01010101;
    if (mode == 2) sa = (127 + hi) * 0x01010101;
    if (mode == 3) { av = hi ? 0 : ones; bv = hi ? ones : 0; }
    const i32x8 a = {av, av, av, av, 0, 0, 0, 0}, b = {bv, bv, bv, bv, 0, 0, 0, 0};
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, sa, 0, sb);
    for (int r = 0; r < 16; r++) out[(mode * 64 + l) * 16 + r] = acc[r];
}
int main() {
    float* d; hipMalloc(&d, 4 * 64 * 16 * 4);
    for (int m = 0; m < 4; m++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, m);
    static float h[4 * 64 * 16];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("no device\n"); return 1; }
    int ok_row = 1, ok_col = 1, ok_half = 1, ok_pair = 1;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 16; r++) {
            const int row = (int)lround(log2(h[(0 * 64 + l) * 16 + r] / 64.0)), col = (int)lround(log2(h[(1 * 64 + l) * 16 + r] / 64.0));
            if (row != 8 * (r / 4) + 4 * (l / 32) + (r % 4) || h[(0 * 64 + l) * 16 + r] != 64.0f * exp2f((float)row)) ok_row = 0;
            if (col != l % 32 || h[(1 * 64 + l) * 16 + r] != 64.0f * exp2f((float)col)) ok_col = 0;
            if (h[(2 * 64 + l) * 16 + r] != 96.0f) ok_half = 0;
            if (h[(3 * 64 + l) * 16 + r] != 0.0f) ok_pair = 0;
        }
    printf("v_mfma_scale_f32_32x32x64_f8f6f4, FP4 x FP4:\n");
    printf("  D layout: row(l, r) == 8*(r/4) + 4*(l/32) + r%%4 (D = 64 * 2^row exactly): %s;  col(l) == l %% 32: %s\n", ok_row ? "yes" : "NO", ok_col ? "yes" : "NO");
    printf("  a lane's E8M0 scale applies to its own 32 K values (lane half = K half; 32*1 + 32*2 = 96 everywhere): %s  (sample %.1f)\n", ok_half ? "yes" : "NO", h[2 * 64 * 16]);
    printf("  A's and B's lane halves pair up by the same index (low-half-only A x high-half-only B = 0): %s  (sample %.1f)\n", ok_pair ? "yes" : "NO", h[3 * 64 * 16]);
    for (int l = 0; l < 64; l += 31) { printf("  lane %2d rows:", l); for (int r = 0; r < 16; r++) printf(" %d", (int)lround(log2(h[(0 * 64 + l) * 16 + r] / 64.0))); printf("\n"); }
    return ok_row && ok_col && ok_half && ok_pair ? 0 : 2;
}
