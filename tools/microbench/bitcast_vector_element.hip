// hipcc 7.2 (AMD clang 22) at -O3, gfx950: k1 below — __builtin_bit_cast(int, r.y) on an ELEMENT of an ext_vector_type(2) float after a vector add —
// compiles to v_mad_i32_i24 v1, v1, 48, v1 (element 0 twice); k2 (the same through scalar floats) is correct.  Found while writing k_describe_od's
// tap addressing (round 6); build with -save-temps and read the .s.
#include <hip/hip_runtime.h>
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void k1(const float* in, unsigned* out) {     // no asm
    f2v r = {in[threadIdx.x], in[threadIdx.x + 64]};
    const f2v M = {12582912.0f, 12582912.0f};
    r = r + M;
    out[threadIdx.x] = (unsigned)(__mul24(__builtin_bit_cast(int, r.x), 48) + __builtin_bit_cast(int, r.y));
}
__global__ void k2(const float* in, unsigned* out) {     // scalar adds
    float rx = in[threadIdx.x] + 12582912.0f, ry = in[threadIdx.x + 64] + 12582912.0f;
    out[threadIdx.x] = (unsigned)(__mul24(__builtin_bit_cast(int, rx), 48) + __builtin_bit_cast(int, ry));
}
__global__ void k3(const float* in, unsigned* out) {     // plain multiply
    float rx = in[threadIdx.x] + 12582912.0f, ry = in[threadIdx.x + 64] + 12582912.0f;
    out[threadIdx.x] = (unsigned)(__builtin_bit_cast(int, rx) * 48 + __builtin_bit_cast(int, ry));
}
