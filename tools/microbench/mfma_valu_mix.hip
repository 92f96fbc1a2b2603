// Microbenchmark (round 2): does VALU work issued between MFMAs of the SAME wave hide in the matrix pipe's shadow on gfx950?
// Each wave loops over { 1 x v_mfma_i32_32x32x32_i8 ; N x one VALU opcode } with 2 independent accumulators, 8 waves per SIMD
// (and a second run with 2 waves per SIMD).  Reports cycles per MFMA per SIMD: 36 = matrix-pipe bound, more = the VALU work shows.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_mix.hip -o mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define V6(OP) OP OP OP OP OP OP
#define KERN(NAME, OPS)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(int* out, int iters) {                                   \
        i32x4 a = {(int)threadIdx.x, 1, 2, 3}, bq = {(int)blockIdx.x, 5, 6, 7};                           \
        i32x16 acc0 = {}, acc1 = {};                                                                      \
        unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, y = blockIdx.x | 1u;                        \
        for (int i = 0; i < iters; i++) {                                                                 \
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq, acc0, 0, 0, 0);                           \
            asm volatile(OPS : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(y));                                    \
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq, acc1, 0, 0, 0);                           \
            asm volatile(OPS : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(y));                                    \
        }                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = acc0[0] ^ acc1[5] ^ (int)(x0 ^ x1 ^ x2);                    \
    }
#define MIN2 "v_min_u32 %0, %0, %3\n v_min_u32 %1, %1, %3\n"
#define MIN6 "v_min_u32 %0, %0, %3\n v_min_u32 %1, %1, %3\n v_min_u32 %2, %2, %3\n v_min_u32 %0, %0, %3\n v_min_u32 %1, %1, %3\n v_min_u32 %2, %2, %3\n"
#define ADD6 "v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %3\n v_add_u32 %2, %2, %3\n v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %3\n v_add_u32 %2, %2, %3\n"
#define MED6 "v_med3_u32 %0, %0, %3, %1\n v_med3_u32 %1, %1, %3, %2\n v_med3_u32 %2, %2, %3, %0\n v_med3_u32 %0, %0, %3, %1\n v_med3_u32 %1, %1, %3, %2\n v_med3_u32 %2, %2, %3, %0\n"
#define MIN4 "v_min_u32 %0, %0, %3\n v_min_u32 %1, %1, %3\n v_min_u32 %2, %2, %3\n v_min_u32 %0, %0, %3\n"
#define MIN3 "v_min3_u32 %0, %0, %3, %1\n v_min3_u32 %1, %1, %3, %2\n v_min3_u32 %2, %2, %3, %0\n"
#define ADD12 ADD6 ADD6
KERN(k_none, "")
KERN(k_min2, MIN2) KERN(k_min4, MIN4) KERN(k_min6, MIN6) KERN(k_add6, ADD6) KERN(k_add12, ADD12) KERN(k_med6, MED6) KERN(k_min3x3, MIN3)
typedef void (*kern_t)(int*, int);
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    int* out; hipMalloc(&out, (size_t)cus * 32 * 256 * 4);
    struct { const char* name; kern_t k; } K[] = {{"MFMA only", k_none}, {"+2 v_min_u32", k_min2}, {"+4 v_min_u32", k_min4}, {"+6 v_min_u32", k_min6}, {"+6 v_med3_u32", k_med6},
                                                  {"+3 v_min3_u32", k_min3x3}, {"+6 v_add_u32", k_add6}, {"+12 v_add_u32", k_add12}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {8, 2, 1}) {
        const int blocks = cus * wps * 4;   // wps waves per SIMD, 4 rounds
        printf("-- %d wave(s) per SIMD\n", wps);
        for (auto& k : K) {
            const int it = 2000;
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 20);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, it);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double n_mfma = (double)blocks * 4 * it * 2;
            printf("%-16s %8.3f ms  = %.1f cycles per MFMA per SIMD (2.4 GHz)\n", k.name, ms, (double)cus * 4 * 2.4e9 / (n_mfma / (ms * 1e-3)));
        }
    }
    return 0;
}
