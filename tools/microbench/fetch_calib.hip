// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the ORB kernels (round 2).
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly half the bytes of a 16 B/lane streaming read; other widths are uncalibrated.
// Each kernel reads (or writes) a KNOWN number of unique bytes from a 1 GiB buffer (far beyond L2 + Infinity Cache), once:
//   k_read4   aligned dword per lane, coalesced            (staging loads of k_fast_cells / k_resize / k_blur)
//   k_read16  aligned dwordx4 per lane, coalesced          (the guide's calibrated case)
//   k_read1   one byte per lane, coalesced                 (byte-load kernel variants)
//   k_patch   unaligned dword gathers: every group of 8 lanes reads a 32-byte run at an arbitrary byte offset, runs 640 bytes
//             apart (a keypoint patch row of k_describe); unique bytes = 32 per run, but the run touches 1-2 64-byte lines
//   k_write4  aligned dword store per lane
// run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetch_calib   and   --pmc WRITE_SIZE ...; tools/fetch_calib_table.py prints the ratios
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_read4(const uint32_t* p, uint32_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_read16(const uint4* p, uint32_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_read1(const uint8_t* p, uint32_t* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_patch(const uint8_t* p, uint32_t* out, size_t nruns, int misalign) {
    // run r starts at byte 640 * r + (r * 7 + misalign) % 29: lanes 8 j .. 8 j + 7 of a wave read its 8 dwords
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (size_t r = t >> 3; r < nruns; r += ((size_t)gridDim.x * blockDim.x) >> 3) {
        const uint8_t* q = p + 640 * r + (r * 7 + misalign) % 29 + 4 * (t & 7);
        uint32_t v;
        __builtin_memcpy(&v, q, 4);
        acc ^= v;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_write4(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    uint8_t* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes + 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes + 4096);
    hipDeviceSynchronize();
    const int blocks = 256 * 16, threads = 256;
    hipLaunchKernelGGL(k_read4, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)buf, out, bytes / 4);
    hipLaunchKernelGGL(k_read16, dim3(blocks), dim3(threads), 0, 0, (const uint4*)buf, out, bytes / 16);
    hipLaunchKernelGGL(k_read1, dim3(blocks), dim3(threads), 0, 0, (const uint8_t*)buf, out, bytes / 4);      // a quarter of the buffer
    hipLaunchKernelGGL(k_patch, dim3(blocks), dim3(threads), 0, 0, (const uint8_t*)buf, out, bytes / 640, 3);
    hipLaunchKernelGGL(k_write4, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, bytes / 4);
    hipDeviceSynchronize();
    printf("known bytes: k_read4 %zu  k_read16 %zu  k_read1 %zu  k_patch unique %zu (runs %zu x 32 B; 64-B lines touched: see table)  k_write4 %zu\n",
           bytes, bytes, bytes / 4, (bytes / 640) * 32, bytes / 640, bytes);
    return 0;
}
