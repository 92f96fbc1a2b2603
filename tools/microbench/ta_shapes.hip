// Microbenchmark (round 4): what does the vector-memory front end (TA / L1) charge for the access SHAPES of k_describe's gathers?
// Counters put that kernel's texture addresser at 71-80 % busy, and ablation builds price its two gathers (37 x 40-byte window rows by
// LDS-DMA dwords, 31 x 32-byte patch rows by byte-aligned 16-byte loads) at 0.49 of its 0.73 ms.  Every case: all waves of a full
// chip issue the same kind of load over a small L2-resident region (4 MB), 64 loads per lane and iteration, no dependent use of the
// data inside the loop (one xor at the end); reported: cycles per wave instruction per CU and bytes per cycle per CU.
// build: hipcc --offload-arch=gfx950 -O3 ta_shapes.hip -o ta_shapes      run: ./ta_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

// lane -> byte offset inside a wave's region for shape s; `piece` bytes contiguous per row, rows `pitch` apart, `misalign` added
struct Shape { const char* name; int bytes_per_lane; int piece; int pitch; int misalign; int dma; };

template <int BPL, bool DMA>
__global__ __launch_bounds__(256) void k_load(const uint8_t* __restrict__ buf, uint32_t* out, int iters, int piece, int pitch, int misalign, unsigned region_mask) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 1024 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lanes_per_row = piece / BPL;
    const int row = lane / lanes_per_row, col = lane - row * lanes_per_row;
    const unsigned lane_off = (unsigned)(row * pitch + col * BPL + misalign);
    unsigned base = (blockIdx.x * 4 + wave) * 9973u * 64u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned a = ((base + (unsigned)k * 40960u) & region_mask) + lane_off;
            if constexpr (DMA) {
                if constexpr (BPL == 4) __builtin_amdgcn_global_load_lds((gptr_t)(buf + a), (lptr_t)(lds + wave * 4096 + (k & 3) * 1024), 4, 0, 0);
                else __builtin_amdgcn_global_load_lds((gptr_t)(buf + a), (lptr_t)(lds + wave * 4096 + (k & 3) * 1024), 16, 0, 0);
            } else if constexpr (BPL == 4) {
                uint32_t v;
                asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(buf + a) : "memory");
                asm volatile("" :: "v"(v));
            } else {
                u32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(buf + a) : "memory");
                asm volatile("" :: "v"(v));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        base += 7919u * 64u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int BPL>
__global__ __launch_bounds__(256) void k_store(uint8_t* __restrict__ buf, uint32_t* out, int iters, int piece, int pitch, int misalign, unsigned region_mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lanes_per_row = piece / BPL;
    const int row = lane / lanes_per_row, col = lane - row * lanes_per_row;
    const unsigned lane_off = (unsigned)(row * pitch + col * BPL + misalign);
    unsigned base = (blockIdx.x * 4 + wave) * 9973u * 64u;
    const u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned a = ((base + (unsigned)k * 40960u) & region_mask) + lane_off;
            if constexpr (BPL == 4) asm volatile("global_store_dword %0, %1, off" :: "v"(buf + a), "v"(v.x) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(buf + a), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        base += 7919u * 64u;
    }
    if (iters < 0) out[threadIdx.x] = 0;
}

// the 37 x 40-byte window of k_describe out of a STRIP-MAJOR plane (column strips of SW bytes, the rows of a strip back to back):
// dword lane e of instruction n -> (row e / 10, column e % 10); byte x = xa + 4 col; address = (x / SW) * strip_bytes + row * SW + x % SW
__global__ __launch_bounds__(256) void k_window_strips(const uint8_t* __restrict__ buf, uint32_t* out, int iters, int SW, int xa, unsigned region_mask) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 1024 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned strip_bytes = 480u * (unsigned)SW;
    unsigned lane_off[6];
    for (int n = 0; n < 6; n++) {
        const int e = 64 * n + lane, r = e / 10, c = e - 10 * r, x = xa + 4 * c;
        lane_off[n] = SW ? (unsigned)(x / SW) * strip_bytes + (unsigned)(r * SW + x % SW) : (unsigned)(r * 640 + x);
    }
    unsigned base = (blockIdx.x * 4 + wave) * 9973u * 64u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const unsigned a = ((base + (unsigned)(k / 6) * 40960u) & region_mask) + lane_off[k % 6];
            if (k % 6 < 5 || lane < 50) __builtin_amdgcn_global_load_lds((gptr_t)(buf + a), (lptr_t)(lds + wave * 4096 + (k & 3) * 1024), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        base += 7919u * 64u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}

int main() {
    const size_t region = 4u << 20;
    uint8_t* buf; uint32_t* out;
    if (hipMalloc(&buf, region + (4 << 20)) != hipSuccess || hipMalloc(&out, 4096 * 256 * 4) != hipSuccess) return 1;
    hipMemset(buf, 1, region + (4 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const Shape shapes[] = {
        {"dword, 256 B contiguous per wave", 4, 256, 256, 0, 0},
        {"dword, 64-byte rows (pitch 640)", 4, 64, 640, 0, 0},
        {"dword, 40-byte rows (pitch 640)  [window, VGPR]", 4, 40, 640, 0, 0},
        {"dword DMA, 256 B contiguous", 4, 256, 256, 0, 1},
        {"dword DMA, 64-byte rows", 4, 64, 640, 0, 1},
        {"dword DMA, 40-byte rows (pitch 640)  [window]", 4, 40, 640, 0, 1},
        {"dword DMA, 40-byte rows, +4 B misaligned", 4, 40, 640, 4, 1},
        {"x4, 1024 B contiguous per wave", 16, 1024, 1024, 0, 0},
        {"x4, 64-byte rows aligned (pitch 640)", 16, 64, 640, 0, 0},
        {"x4, 48-byte rows aligned", 16, 48, 640, 0, 0},
        {"x4, 32-byte rows aligned", 16, 32, 640, 0, 0},
        {"x4, 32-byte rows, +1 B  [patch]", 16, 32, 640, 1, 0},
        {"x4, 32-byte rows, +4 B", 16, 32, 640, 4, 0},
        {"x4, 32-byte rows, +8 B", 16, 32, 640, 8, 0},
        {"x4 DMA, 1024 B contiguous", 16, 1024, 1024, 0, 1},
        {"x4 DMA, 64-byte rows aligned", 16, 64, 640, 0, 1},
        {"x4 DMA, 48-byte rows aligned", 16, 48, 640, 0, 1},
        {"x4 DMA, 48-byte rows, +4 B  [window x4]", 16, 48, 640, 4, 1},
        {"x4 DMA, 96-byte rows aligned  [k_blur_mfma in]", 16, 96, 640, 0, 1},
        {"x4 DMA, 128-byte rows aligned", 16, 128, 640, 0, 1},
        {"x4 DMA, 160-byte rows aligned", 16, 160, 640, 0, 1},
        {"x4 DMA, 256-byte rows aligned", 16, 256, 640, 0, 1},
        {"STORE dword, 256 B contiguous per wave", 4, 256, 256, 0, 2},
        {"STORE x4, 1024 B contiguous per wave", 16, 1024, 1024, 0, 2},
        {"STORE x4, 64-byte rows aligned (pitch 640)  [k_blur_mfma out]", 16, 64, 640, 0, 2},
        {"STORE x4, 128-byte rows aligned", 16, 128, 640, 0, 2},
        {"STORE x4, 256-byte rows aligned", 16, 256, 640, 0, 2},
        {"STORE dword, 256-byte rows (one row per wave)  [k_resize out]", 4, 256, 640, 0, 2},
    };
    const int blocks = 2048, iters = 200;
    for (const Shape& s : shapes) {
        for (int rep = 0; rep < 2; rep++) {
            const int it = rep ? iters : 5;
            hipEventRecord(e0);
            const unsigned mask = (unsigned)(region - 1) & ~63u;
            if (s.dma == 2 && s.bytes_per_lane == 4) hipLaunchKernelGGL((k_store<4>), blocks, 256, 0, 0, buf, out, it, s.piece, s.pitch, s.misalign, mask);
            if (s.dma == 2 && s.bytes_per_lane == 16) hipLaunchKernelGGL((k_store<16>), blocks, 256, 0, 0, buf, out, it, s.piece, s.pitch, s.misalign, mask);
            if (s.bytes_per_lane == 4 && !s.dma) hipLaunchKernelGGL((k_load<4, false>), blocks, 256, 0, 0, buf, out, it, s.piece, s.pitch, s.misalign, mask);
            if (s.bytes_per_lane == 4 && s.dma == 1) hipLaunchKernelGGL((k_load<4, true>), blocks, 256, 0, 0, buf, out, it, s.piece, s.pitch, s.misalign, mask);
            if (s.bytes_per_lane == 16 && !s.dma) hipLaunchKernelGGL((k_load<16, false>), blocks, 256, 0, 0, buf, out, it, s.piece, s.pitch, s.misalign, mask);
            if (s.bytes_per_lane == 16 && s.dma == 1) hipLaunchKernelGGL((k_load<16, true>), blocks, 256, 0, 0, buf, out, it, s.piece, s.pitch, s.misalign, mask);
            hipEventRecord(e1); hipEventSynchronize(e1);
            if (rep) {
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                const double winsts = (double)blocks * 4 * iters * 16;
                const double cyc_per_inst_cu = ms * 1e-3 * 2.4e9 * 256 / winsts;
                printf("%-66s %8.3f ms  %7.1f cycles per wave instruction per CU   %6.1f B per cycle per CU\n", s.name, ms, cyc_per_inst_cu, 64.0 * s.bytes_per_lane / cyc_per_inst_cu);
            }
        }
    }
    for (int SW : {0, 32, 64, 128})
        for (int xa : {0, 12, 28, 44, 60}) {
            const unsigned mask = (unsigned)(region - 1) & ~63u;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_window_strips, blocks, 256, 0, 0, buf, out, rep ? iters : 5, SW, xa, mask);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double windows = (double)blocks * 4 * iters * 2;
            printf("window 37 x 40 B by six dword DMA instructions, %s, x0 %% 64 = %2d: %8.3f ms  %6.1f cycles per window per CU\n",
                   SW == 0 ? "row-major pitch 640      " : SW == 32 ? "strip-major, 32-B strips " : SW == 64 ? "strip-major, 64-B strips " : "strip-major, 128-B strips", xa, ms, ms * 1e-3 * 2.4e9 * 256 / windows);
        }
    return 0;
}
