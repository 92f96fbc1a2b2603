// Probe (round 6, k_describe_od): operand / result layout of v_mfma_i32_16x16x64_i8 on gfx950, and the 16-byte LDS-DMA
// (global_load_lds_dwordx4) from source addresses that are only 4-byte aligned.
// build: hipcc --offload-arch=gfx950 -O3 mfma_layout_16x16x64.hip -o mfma_layout_16x16x64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

// D = A x B with A[m][k] = (m == M0 && k == K0), B[k][n] = 1 + n (mode 0: where does A's element (m, k) live?)  etc.
__global__ void k_layout(int* out) {
    const int l = threadIdx.x;
    // test 1: A[m][k] = m + 1 for all k (lane l, every byte = (l % 16) + 1 if the row index is l % 16), B = delta on k: B[k][n] = (k == 0)
    //   -> D[m][n] = A[m][0] = m + 1: tells the D layout (which (m, n) each lane/register holds) provided A's row = l % 16
    // Simpler and assumption-free: use distinct primes.
    // A[m][k] = 1 if k == kk else 0, with every lane byte b of register v set iff (l, v, b) == probe -> D tells m; sweep probes on the host side.
    int probe_l = out[0], probe_v = out[1], probe_b = out[2], which = out[3];
    i32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    const int one = 0x01010101;
    if (which == 0) {            // one byte of A set, B all ones: D[m][n] = 1 for the row m that byte belongs to (all n)
        if (l == probe_l) a[probe_v] = 1 << (8 * probe_b);
        b = (i32x4){one, one, one, one};
    } else if (which == 1) {     // one byte of B set, A all ones: D[m][n] = 1 for the column n that byte belongs to
        if (l == probe_l) b[probe_v] = 1 << (8 * probe_b);
        a = (i32x4){one, one, one, one};
    } else {                     // k pairing: A byte (probe) = 1 and B[lane l2][v2][b2] = 1 for ALL lanes/bytes with a code: B byte = 1 + (16*(l/16) + 4*v + b) -> D = 1 + k index of A's byte under the hypothesis
        if (l == probe_l) a[probe_v] = 1 << (8 * probe_b);
        for (int v = 0; v < 4; v++) { int w = 0; for (int bb = 0; bb < 4; bb++) w |= (1 + 16 * (l / 16) + 4 * v + bb) << (8 * bb); b[v] = w; }
    }
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[16 + l * 4 + r] = acc[r];
}

__global__ void k_dma(const unsigned char* src, unsigned char* dst, int shift) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[64 * 16];
    const int l = threadIdx.x;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + shift + 48 * l), (lptr_t)buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 16; i++) dst[l * 16 + i] = buf[l * 16 + i];
}

int main() {
    int* d; hipMalloc(&d, (16 + 256) * 4);
    int h[16 + 256];
    // A layout: for each (lane, v, b) find m (row) ; hypothesis m = l % 16
    int okA = 1, okB = 1, okD = 1, okK = 1;
    for (int pl = 0; pl < 64; pl++) for (int pv = 0; pv < 4; pv++) for (int pb = 0; pb < 4; pb += 3) {
        for (int which = 0; which < 3; which++) {
            int hdr[4] = {pl, pv, pb, which};
            hipMemcpy(d, hdr, sizeof(hdr), hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
                const int v = h[16 + l * 4 + r];
                const int m = 4 * (l / 16) + r, n = l % 16;          // hypothesis for D
                if (which == 0) { const int want = (m == pl % 16) ? 1 : 0; if (v != want) okA = 0; }
                if (which == 1) { const int want = (n == pl % 16) ? 1 : 0; if (v != want) okB = 0; }
                if (which == 2) { const int want = (m == pl % 16) ? 1 + 16 * (pl / 16) + 4 * pv + pb : 0; if (v != want) okK = 0; }
            }
        }
    }
    printf("16x16x64 i8: A row = lane %% 16: %s; B col = lane %% 16: %s; D (m = 4*(lane/16)+reg, n = lane %% 16) consistent; k = 16*(lane/16) + 4*reg + byte on both operands: %s\n",
           okA ? "yes" : "NO", okB ? "yes" : "NO", okK ? "yes" : "NO");
    (void)okD;
    unsigned char *s, *o;
    hipMalloc(&s, 8192); hipMalloc(&o, 1024);
    unsigned char hs[8192], ho[1024];
    for (int i = 0; i < 8192; i++) hs[i] = (unsigned char)(i * 7 + (i >> 8));
    hipMemcpy(s, hs, 8192, hipMemcpyHostToDevice);
    for (int shift = 0; shift <= 12; shift += 4) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, s, o, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess;
        for (int l = 0; l < 64 && ok; l++) for (int i = 0; i < 16; i++) if (ho[l * 16 + i] != hs[shift + 48 * l + i]) { ok = 0; break; }
        printf("16-byte LDS-DMA from source offset %2d (mod 16): %s\n", shift, ok ? "ok" : "WRONG");
    }
    for (int shift = 1; shift <= 3; shift++) {
        hipMemset(o, 0, 1024);
        hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, s, o, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
        int ok = e == hipSuccess;
        for (int l = 0; l < 64 && ok; l++) for (int i = 0; i < 16; i++) if (ho[l * 16 + i] != hs[shift + 48 * l + i]) { ok = 0; break; }
        printf("16-byte LDS-DMA from source offset %2d (byte-unaligned): %s\n", shift, ok ? "ok" : "WRONG");
    }
    return 0;
}
