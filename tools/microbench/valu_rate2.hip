// Microbenchmark, second set (round 2): issue cost of the opcodes the SWAR forms of the FAST kernels rely on (v_bitop3, packed 16-bit
// integer ops, v_lerp_u8, float min/max, mbcnt, wave shuffles through ds_bpermute / DPP), plus two functional probes:
//   * does ds_read_b32 accept an address that is not a multiple of 4 on this device (SH_MEM_CONFIG alignment mode)?
//   * integer-MFMA issue rate next to VALU work (the Hamming-by-MFMA matcher).
// build: hipcc --offload-arch=gfx950 -O3 valu_rate2.hip -o valu_rate2      run: ./valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(OP)                                                                                                  \
    asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)                                        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                 : "v"(b), "v"(c), "s"(m)                                                                         \
                 : "vcc", "s20", "s21");

#define KERNEL(NAME, OP)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters) {                                      \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned b = blockIdx.x | 1u, c = threadIdx.x * 2654435761u;                                             \
        unsigned long long m = 0x5555555555555555ull ^ (unsigned long long)iters;                               \
        for (int i = 0; i < iters; ++i) {                                                                        \
            REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)                              \
            REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)                              \
        }                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                            \
    }

#define OP_BITOP3(x) "v_bitop3_b32 " #x ", " #x ", %8, %9 bitop3:0xc8\n"
#define OP_PKADD(x) "v_pk_add_u16 " #x ", " #x ", %8\n"
#define OP_PKSUB(x) "v_pk_sub_i16 " #x ", " #x ", %8\n"
#define OP_PKMIN(x) "v_pk_min_u16 " #x ", " #x ", %8\n"
#define OP_PKMAX(x) "v_pk_max_u16 " #x ", " #x ", %8\n"
#define OP_PKLSHR(x) "v_pk_lshrrev_b16 " #x ", 1, " #x "\n"
#define OP_LERP(x) "v_lerp_u8 " #x ", " #x ", %8, %9\n"
#define OP_MINF(x) "v_min_f32 " #x ", " #x ", %8\n"
#define OP_MAXF(x) "v_max_f32 " #x ", " #x ", %8\n"
#define OP_MBCNT(x) "v_mbcnt_lo_u32_b32 " #x ", %8, " #x "\n"
#define OP_SUBREV(x) "v_subrev_u32 " #x ", " #x ", %8\n"
#define OP_ADDCO(x) "v_add_co_u32 " #x ", vcc, " #x ", %8\n"
#define OP_PKFMA(x) "v_pk_fma_f32 " #x ", " #x ", %8, %9\n"
#define OP_MAD_I24(x) "v_mad_i32_i24 " #x ", " #x ", %8, %9\n"
#define OP_MIN3(x) "v_min3_u32 " #x ", " #x ", %8, %9\n"
#define OP_BFI(x) "v_bfi_b32 " #x ", " #x ", %8, %9\n"
#define OP_ALIGNBIT(x) "v_alignbit_b32 " #x ", " #x ", %8, 8\n"
#define OP_CVTPKU8(x) "v_cvt_pk_u8_f32 " #x ", " #x ", 1, %8\n"
#define OP_XNOR(x) "v_xnor_b32 " #x ", " #x ", %8\n"
#define OP_MAXU16(x) "v_max_u16 " #x ", " #x ", %8\n"
#define OP_ADDU16(x) "v_add_u16 " #x ", " #x ", %8\n"
#define OP_BCNT0(x) "v_bcnt_u32_b32 " #x ", " #x ", 0\n"
#define OP_DOT4I8(x) "v_dot4_i32_i8 " #x ", %8, %9, " #x "\n"
#define OP_DOT8I4(x) "v_dot8_i32_i4 " #x ", %8, %9, " #x "\n"
#define OP_FMAMIX(x) "v_fma_mix_f32 " #x ", " #x ", %8, %9\n"
KERNEL(k_bitop3, OP_BITOP3) KERNEL(k_pkadd, OP_PKADD) KERNEL(k_pksub, OP_PKSUB) KERNEL(k_pkmin, OP_PKMIN) KERNEL(k_pkmax, OP_PKMAX) KERNEL(k_pklshr, OP_PKLSHR)
KERNEL(k_lerp, OP_LERP) KERNEL(k_minf, OP_MINF) KERNEL(k_maxf, OP_MAXF) KERNEL(k_mbcnt, OP_MBCNT) KERNEL(k_subrev, OP_SUBREV) KERNEL(k_addco, OP_ADDCO)
KERNEL(k_madi24, OP_MAD_I24) KERNEL(k_min3, OP_MIN3) KERNEL(k_bfi, OP_BFI) KERNEL(k_alignbit, OP_ALIGNBIT) KERNEL(k_cvtpku8, OP_CVTPKU8)
KERNEL(k_xnor, OP_XNOR) KERNEL(k_maxu16, OP_MAXU16) KERNEL(k_addu16, OP_ADDU16) KERNEL(k_bcnt0, OP_BCNT0) KERNEL(k_dot4i8, OP_DOT4I8) KERNEL(k_dot8i4, OP_DOT8I4)

typedef void (*kern_t)(unsigned*, int);

// ---- unaligned LDS dword reads: lane i reads 4 bytes at byte address base + i*5 + 1 (never a multiple of 4 for most lanes)
__global__ void k_lds_unaligned(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned char s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(s) + threadIdx.x * 5 + 1;      // LDS address (low 32 bits of the generic shared pointer)
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[threadIdx.x] = v;
}

// ---- integer MFMA issue rate: v_mfma_i32_16x16x64_i8 / 32x32x32_i8, one accumulator chain per wave group of 4 independent accumulators
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_mfma16(int* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, bq = {(int)blockIdx.x, 5, 6, 7};
    i32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bq, acc3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] ^ acc1[1] ^ acc2[2] ^ acc3[3];
}
__global__ __launch_bounds__(256) void k_mfma32(int* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, bq = {(int)blockIdx.x, 5, 6, 7};
    i32x16 acc0 = {}, acc1 = {};
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq, acc1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] ^ acc1[5];
}

int main() {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = pr.multiProcessorCount, blocks = cus * 8 * 4;
    unsigned* out;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
    struct { const char* name; kern_t k; int per; } K[] = {
        {"v_bitop3_b32", k_bitop3, 1}, {"v_pk_add_u16", k_pkadd, 1}, {"v_pk_sub_i16", k_pksub, 1}, {"v_pk_min_u16", k_pkmin, 1}, {"v_pk_max_u16", k_pkmax, 1},
        {"v_pk_lshrrev_b16", k_pklshr, 1}, {"v_lerp_u8", k_lerp, 1}, {"v_min_f32", k_minf, 1}, {"v_max_f32", k_maxf, 1}, {"v_mbcnt_lo_u32_b32", k_mbcnt, 1},
        {"v_subrev_u32", k_subrev, 1}, {"v_add_co_u32", k_addco, 1}, {"v_mad_i32_i24", k_madi24, 1}, {"v_min3_u32", k_min3, 1},
        {"v_bfi_b32", k_bfi, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_cvt_pk_u8_f32", k_cvtpku8, 1}, {"v_xnor_b32", k_xnor, 1}, {"v_max_u16", k_maxu16, 1},
        {"v_add_u16", k_addu16, 1}, {"v_bcnt_u32_b32 x,0", k_bcnt0, 1}, {"v_dot4_i32_i8", k_dot4i8, 1}, {"v_dot8_i32_i4", k_dot8i4, 1}};
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%d CUs, clock %d MHz (reported), %d blocks x 256 threads, %d x 128 ops per wave\n", cus, pr.clockRate / 1000, blocks, iters);
    for (auto& k : K) {
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 50);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 * iters * 128 * k.per;
        const double rate = insts / (ms * 1e-3);
        printf("%-22s %8.3f ms  %.3e wave-insts/s  = %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", k.name, ms, rate,
               (double)cus * 4 * 2.4e9 / rate);
    }
    {   // unaligned LDS read probe
        hipMemset(out, 0, 64 * 4);
        hipLaunchKernelGGL(k_lds_unaligned, dim3(1), dim3(64), 0, 0, out);
        unsigned h[64];
        hipError_t e = hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        int good = 0, aligned_down = 0;
        for (int i = 0; i < 64; i++) {
            const int a = i * 5 + 1;
            unsigned want = 0, down = 0;
            for (int k = 0; k < 4; k++) { want |= (unsigned)((unsigned char)((a + k) * 7 + 3)) << (8 * k); down |= (unsigned)((unsigned char)(((a & ~3) + k) * 7 + 3)) << (8 * k); }
            good += h[i] == want;
            aligned_down += h[i] == down;
        }
        printf("ds_read_b32 at unaligned addresses: %s (%d/64 lanes exact, %d/64 equal the aligned-down dword, status %d)\n",
               good == 64 ? "SUPPORTED" : "NOT byte-exact", good, aligned_down, (int)e);
    }
    {   // integer MFMA rates
        const int it = 4000;
        for (int which = 0; which < 2; which++) {
            auto run = [&](int n) { if (which == 0) hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(256), 0, 0, (int*)out, n); else hipLaunchKernelGGL(k_mfma32, dim3(blocks), dim3(256), 0, 0, (int*)out, n); };
            run(50);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            run(it);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double n_mfma = (double)blocks * 4 * it * (which == 0 ? 4 : 2);
            const double macs = n_mfma * (which == 0 ? 16.0 * 16 * 64 : 32.0 * 32 * 32);
            printf("%-26s %8.3f ms  %.3e MFMA/s  %.1f TOPS (2 ops per MAC)  = %.1f cycles per MFMA per SIMD\n",
                   which == 0 ? "v_mfma_i32_16x16x64_i8" : "v_mfma_i32_32x32x32_i8", ms, n_mfma / (ms * 1e-3), 2 * macs / (ms * 1e-3) / 1e12,
                   (double)cus * 4 * 2.4e9 / (n_mfma / (ms * 1e-3)));
        }
    }
    return 0;
}
