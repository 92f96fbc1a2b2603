// Probe: operand / result layout of v_mfma_i32_32x32x32_i8 on gfx950 (the matcher's top-2 epilogue indexes the accumulator registers by it).
// A: every byte of lane l = l % 32 (+ 32 for lanes >= 32 in the second run), B: every byte = 1  ->  D[i][j] = 32 * rowid(i): register r of lane l tells which A-row it holds.
// Swapped for the columns.  build: hipcc --offload-arch=gfx950 -O3 mfma_layout.hip -o mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ void k(int* out, int mode) {
    const int l = threadIdx.x;
    const int id = l % 32, one = 0x01010101;
    const int rep = id * 0x01010101;
    i32x4 a, b;
    if (mode == 0) { a = (i32x4){rep, rep, rep, rep}; b = (i32x4){one, one, one, one}; }
    else { b = (i32x4){rep, rep, rep, rep}; a = (i32x4){one, one, one, one}; }
    i32x16 acc = {};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) out[(mode * 64 + l) * 16 + r] = acc[r];
}
int main() {
    int* d; hipMalloc(&d, 2 * 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1);
    static int h[2 * 64 * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_row = 1, ok_col = 1;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 16; r++) {
            const int row = h[(0 * 64 + l) * 16 + r] / 32, col = h[(1 * 64 + l) * 16 + r] / 32;
            if (row != 8 * (r / 4) + 4 * (l / 32) + (r % 4)) ok_row = 0;
            if (col != l % 32) ok_col = 0;
        }
    printf("D layout: row(l, r) == 8*(r/4) + 4*(l/32) + r%%4 : %s;  col(l) == l %% 32 : %s\n", ok_row ? "yes" : "NO", ok_col ? "yes" : "NO");
    for (int l = 0; l < 64; l += 31) { printf("lane %2d rows:", l); for (int r = 0; r < 16; r++) printf(" %d", h[(0 * 64 + l) * 16 + r] / 32); printf("  cols:"); for (int r = 0; r < 16; r++) printf(" %d", h[(64 + l) * 16 + r] / 32); printf("\n"); }
    return 0;
}
