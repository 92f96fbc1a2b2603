// Microbenchmark: issue rate of the integer VALU instructions the ORB kernels are made of, on gfx950.
// Every wave runs a loop of 128 instructions of ONE opcode over 8 independent accumulators (no memory, no LDS); 8 waves per SIMD.
// Prints wave-level instructions per second and the cycles one instruction occupies a SIMD at 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate      run: ./valu_rate
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

#define OP_XOR(x) "v_xor_b32 " #x ", " #x ", %8\n"
#define OP_ADD(x) "v_add_u32 " #x ", " #x ", %8\n"
#define OP_AND_OR(x) "v_and_or_b32 " #x ", " #x ", %8, %9\n"
#define OP_MIN(x) "v_min_u32 " #x ", " #x ", %8\n"
#define OP_SHL(x) "v_lshlrev_b32 " #x ", 1, " #x "\n"
#define OP_BCNT(x) "v_bcnt_u32_b32 " #x ", " #x ", %8\n"
#define OP_PERM(x) "v_perm_b32 " #x ", " #x ", %8, %9\n"
#define OP_DOT4(x) "v_dot4_u32_u8 " #x ", " #x ", %8, %9\n"
#define OP_ALIGN(x) "v_alignbyte_b32 " #x ", " #x ", %8, 1\n"
#define OP_MULLO(x) "v_mul_lo_u32 " #x ", " #x ", %8\n"
#define OP_MAD24(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9\n"
#define OP_SAD(x) "v_sad_u8 " #x ", " #x ", %8, %9\n"
#define OP_FMA(x) "v_fma_f32 " #x ", " #x ", %8, %9\n"
#define OP_CMPCND(x) "v_cmp_gt_u32 vcc, " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc\n"
#define OP_MAX3(x) "v_max3_u32 " #x ", " #x ", %8, %9\n"
#define OP_XAD(x) "v_xad_u32 " #x ", " #x ", %8, %9\n"

#define OP_MOV(x) "v_mov_b32 " #x ", %8\n"
#define OP_CND(x) "v_cndmask_b32 " #x ", " #x ", %8, %10\n"
#define OP_DOT2(x) "v_dot2_u32_u16 " #x ", %8, %9, " #x "\n"
#define OP_LSHLOR(x) "v_lshl_or_b32 " #x ", " #x ", 1, %8\n"
#define OP_ASHR(x) "v_ashrrev_i32 " #x ", 1, " #x "\n"
#define OP_ADD3(x) "v_add3_u32 " #x ", " #x ", %8, %9\n"
#define OP_MINI(x) "v_min_i32 " #x ", " #x ", %8\n"
#define OP_SUB(x) "v_sub_u32 " #x ", " #x ", %8\n"
#define OP_BFE(x) "v_bfe_u32 " #x ", " #x ", 3, 8\n"
#define OP_AND(x) "v_and_b32 " #x ", " #x ", %8\n"
#define OP_OR(x) "v_or_b32 " #x ", " #x ", %8\n"
#define OP_LSHLADD(x) "v_lshl_add_u32 " #x ", " #x ", 1, %8\n"
#define OP_CMP(x) "v_cmp_lt_i32 vcc, " #x ", %8\n"
#define OP_MUL24(x) "v_mul_i32_i24 " #x ", " #x ", %8\n"
#define OP_OR3(x) "v_or3_b32 " #x ", " #x ", %8, %9\n"
KERNEL(k_mov, OP_MOV) KERNEL(k_cnd, OP_CND) KERNEL(k_dot2, OP_DOT2) KERNEL(k_lshlor, OP_LSHLOR) KERNEL(k_ashr, OP_ASHR) KERNEL(k_add3, OP_ADD3)
KERNEL(k_mini, OP_MINI) KERNEL(k_sub, OP_SUB) KERNEL(k_bfe, OP_BFE) KERNEL(k_and, OP_AND) KERNEL(k_or, OP_OR) KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_cmp, OP_CMP) KERNEL(k_mul24, OP_MUL24) KERNEL(k_or3, OP_OR3)
#define OP_LSHR(x) "v_lshrrev_b32 " #x ", 1, " #x "\n"
#define OP_MULF(x) "v_mul_f32 " #x ", " #x ", %8\n"
#define OP_ADDF(x) "v_add_f32 " #x ", " #x ", %8\n"
#define OP_CVTFI(x) "v_cvt_f32_i32 " #x ", " #x "\n"
#define OP_CVTIF(x) "v_cvt_i32_f32 " #x ", " #x "\n"
#define OP_RNDNE(x) "v_rndne_f32 " #x ", " #x "\n"
#define OP_UBYTE(x) "v_cvt_f32_ubyte0 " #x ", " #x "\n"
#define OP_DPP(x) "v_mov_b32_dpp " #x ", " #x " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ANDSDWA(x) "v_and_b32_sdwa " #x ", " #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_MINSDWA(x) "v_min_u32_sdwa " #x ", " #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define OP_ADDSDWA(x) "v_add_u32_sdwa " #x ", " #x ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OP_MED3(x) "v_med3_u32 " #x ", " #x ", %8, %9\n"
#define OP_BFEI(x) "v_bfe_i32 " #x ", " #x ", 3, 8\n"
#define OP_RFL(x) "v_readfirstlane_b32 s20, " #x "\n"
#define OP_CMPS(x) "v_cmp_lt_i32_e64 s[20:21], " #x ", %8\n"
#define OP_MULHI(x) "v_mul_hi_u32 " #x ", " #x ", %8\n"
#define OP_NOT(x) "v_not_b32 " #x ", " #x "\n"
KERNEL(k_lshr, OP_LSHR) KERNEL(k_mulf, OP_MULF) KERNEL(k_addf, OP_ADDF) KERNEL(k_cvtfi, OP_CVTFI) KERNEL(k_cvtif, OP_CVTIF) KERNEL(k_rndne, OP_RNDNE)
KERNEL(k_ubyte, OP_UBYTE) KERNEL(k_dpp, OP_DPP) KERNEL(k_andsdwa, OP_ANDSDWA) KERNEL(k_minsdwa, OP_MINSDWA) KERNEL(k_addsdwa, OP_ADDSDWA) KERNEL(k_med3, OP_MED3)
KERNEL(k_bfei, OP_BFEI) KERNEL(k_rfl, OP_RFL) KERNEL(k_cmps, OP_CMPS) KERNEL(k_mulhi, OP_MULHI) KERNEL(k_not, OP_NOT)
KERNEL(k_xor, OP_XOR) KERNEL(k_add, OP_ADD) KERNEL(k_and_or, OP_AND_OR) KERNEL(k_min, OP_MIN) KERNEL(k_shl, OP_SHL)
KERNEL(k_bcnt, OP_BCNT) KERNEL(k_perm, OP_PERM) KERNEL(k_dot4, OP_DOT4) KERNEL(k_align, OP_ALIGN) KERNEL(k_mullo, OP_MULLO)
KERNEL(k_mad24, OP_MAD24) KERNEL(k_sad, OP_SAD) KERNEL(k_fma, OP_FMA) KERNEL(k_cmpcnd, OP_CMPCND) KERNEL(k_max3, OP_MAX3) KERNEL(k_xad, OP_XAD)

typedef void (*kern_t)(unsigned*, int);

int main() {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = pr.multiProcessorCount, blocks = cus * 8 * 4;     // 8 workgroups of 4 waves per CU = 8 waves per SIMD, 4 rounds
    unsigned* out;
    if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
    struct { const char* name; kern_t k; int per; } K[] = {
        {"v_xor_b32", k_xor, 1}, {"v_add_u32", k_add, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_min_u32", k_min, 1}, {"v_lshlrev_b32", k_shl, 1},
        {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_perm_b32", k_perm, 1}, {"v_dot4_u32_u8", k_dot4, 1}, {"v_alignbyte_b32", k_align, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_sad_u8", k_sad, 1}, {"v_fma_f32", k_fma, 1},
        {"v_cmp+v_cndmask", k_cmpcnd, 2}, {"v_max3_u32", k_max3, 1}, {"v_xad_u32", k_xad, 1},
        {"v_mov_b32", k_mov, 1}, {"v_cndmask_b32", k_cnd, 1}, {"v_dot2_u32_u16", k_dot2, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_ashrrev_i32", k_ashr, 1},
        {"v_add3_u32", k_add3, 1}, {"v_min_i32", k_mini, 1}, {"v_sub_u32", k_sub, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1},
        {"v_lshl_add_u32", k_lshladd, 1}, {"v_cmp_lt_i32", k_cmp, 1}, {"v_mul_i32_i24", k_mul24, 1}, {"v_or3_b32", k_or3, 1},
        {"v_lshrrev_b32", k_lshr, 1}, {"v_mul_f32", k_mulf, 1}, {"v_add_f32", k_addf, 1}, {"v_cvt_f32_i32", k_cvtfi, 1}, {"v_cvt_i32_f32", k_cvtif, 1},
        {"v_rndne_f32", k_rndne, 1}, {"v_cvt_f32_ubyte0", k_ubyte, 1}, {"v_mov_b32_dpp row_shr", k_dpp, 1}, {"v_and_b32_sdwa WORD_1", k_andsdwa, 1},
        {"v_min_u32_sdwa WORD_1", k_minsdwa, 1}, {"v_add_u32_sdwa BYTE_1", k_addsdwa, 1}, {"v_med3_u32", k_med3, 1}, {"v_bfe_i32", k_bfei, 1},
        {"v_readfirstlane_b32", k_rfl, 1}, {"v_cmp_lt_i32 -> sgpr pair", k_cmps, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_not_b32", k_not, 1}};
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
        printf("%-18s %8.3f ms  %.3e wave-insts/s  = %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", k.name, ms, rate,
               (double)cus * 4 * 2.4e9 / rate);
    }
    return 0;
}
