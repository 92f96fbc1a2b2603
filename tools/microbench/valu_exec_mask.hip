// Microbenchmark (round 4): does a wave64 VALU instruction cost fewer issue cycles when only part of EXEC is set?  (The queue tails of
// k_fast_cells run ~100-instruction passes with a handful of active lanes packed at the low lane indices; whether those are priced
// like full passes decides whether pooling them across waves can pay.)  Each kernel runs the same unrolled chain with EXEC restricted
// to lanes [0, n) or to a strided pattern.
// build: hipcc --offload-arch=gfx950 -O3 valu_exec_mask.hip -o valu_exec_mask      run: ./valu_exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(OP)                                                                                                  \
    asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)                                        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                 : "v"(b), "v"(c));

#define KERNEL(NAME, OP)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, unsigned long long mask, unsigned long long mask2) {             \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned b = blockIdx.x | 1u, c = threadIdx.x * 2654435761u;                                             \
        if (mask2 && ((threadIdx.x >> 6) & 1)) mask = mask2;          /* mixed workgroups: odd waves take the second mask */   \
        if ((mask >> (threadIdx.x & 63)) & 1ull) {                                                              \
            for (int i = 0; i < iters; ++i) {                                                                    \
                REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)                          \
                REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)                          \
            }                                                                                                    \
        }                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                            \
    }

#define OP_MIN3(x) "v_min3_u32 " #x ", " #x ", %8, %9\n"
#define OP_ADD(x) "v_add_u32 " #x ", " #x ", %8\n"
#define OP_FMA(x) "v_fma_f32 " #x ", " #x ", %8, %9\n"
#define OP_CND(x) "v_cndmask_b32 " #x ", " #x ", %8, vcc\n"
#define OP_MULLO(x) "v_mul_lo_u32 " #x ", " #x ", %8\n"
#define OP_LSHLADD(x) "v_lshl_add_u32 " #x ", " #x ", 1, %8\n"
KERNEL(k_min3, OP_MIN3) KERNEL(k_add, OP_ADD) KERNEL(k_fma, OP_FMA) KERNEL(k_cnd, OP_CND) KERNEL(k_mullo, OP_MULLO) KERNEL(k_lshladd, OP_LSHLADD)
typedef void (*kern_t)(unsigned*, int, unsigned long long, unsigned long long);

int main() {
    const int blocks = 8192, iters = 1000;
    unsigned* out;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; kern_t k; } ks[] = {{"v_min3_u32", k_min3}, {"v_add_u32", k_add}, {"v_fma_f32", k_fma}, {"v_cndmask_b32", k_cnd}, {"v_mul_lo_u32", k_mullo}, {"v_lshl_add_u32", k_lshladd}};
    struct { const char* name; unsigned long long m; } ms[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0..31", 0xFFFFFFFFull}, {"lanes 0..15", 0xFFFFull}, {"lanes 0..13", 0x3FFFull}, {"lanes 0..11", 0xFFFull}, {"lanes 0..9", 0x3FFull}, {"lanes 0..8", 0x1FFull}, {"lanes 0..7", 0xFFull}, {"lane 0", 1ull},
        {"lanes 16..31", 0xFFFF0000ull}, {"lanes 48..63", 0xFFFF000000000000ull}, {"lanes 0 and 32", 0x100000001ull}, {"every 16th lane", 0x0001000100010001ull}, {"every 4th lane", 0x1111111111111111ull}};
    for (auto& k : ks)
        for (auto& m : ms) {
            hipLaunchKernelGGL(k.k, blocks, 256, 0, 0, out, 10, m.m, 0ull);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, blocks, 256, 0, 0, out, iters, m.m, 0ull);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms_ = 0;
            hipEventElapsedTime(&ms_, e0, e1);
            const double insts = (double)blocks * 4 * iters * 128;
            printf("%-12s %-18s %8.3f ms  = %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", k.name, m.name, ms_, ms_ * 1e-3 * 2.4e9 * 1024 / insts);
        }
    // Is the sparse-EXEC cost ISSUE TIME of the SIMD (other waves wait) or only the sparse wave's own interval (other waves fill it)?
    // Half the waves of every workgroup full, half with one lane: if the 12.5 cycles were SIMD time the kernel would take the mean of the
    // two pure runs (4.4 and 12.5 cycles per instruction), if they are the wave's own business it takes about as long as the slower half alone.
    for (auto& k : ks) {
        float t[3];
        const unsigned long long m1[3] = {~0ull, 1ull, ~0ull}, m2[3] = {0ull, 0ull, 1ull};
        for (int c = 0; c < 3; c++) {
            hipLaunchKernelGGL(k.k, blocks, 256, 0, 0, out, 10, m1[c], m2[c]);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, blocks, 256, 0, 0, out, iters, m1[c], m2[c]);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t[c], e0, e1);
        }
        printf("%-14s all waves full %.3f ms, all waves one lane %.3f ms, even waves full + odd waves one lane %.3f ms (mean of the pure runs %.3f)\n", k.name, t[0], t[1], t[2], 0.5f * (t[0] + t[1]));
    }
    return 0;
}
