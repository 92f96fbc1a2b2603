"""Round 5, call 7 (NOTES.md 10.3): k_blur_mfma with the MB_WAVES waves of a workgroup on ADJACENT strips, staging one shared input tile
(160- / 288-byte row pieces for 2 / 4 waves) and one shared, double-buffered output block (128- / 256-byte row pieces), one barrier per
step.  Bit-exact (tests/test_gpu_parity.py, both group sizes, also with ORBX_BLUR_MFMA=2 on 1920-px rows).  Measured, every kernel alone:
VGA 0.510 -> 0.499 (2 waves) / 0.513 ms (4 waves) per 1024 frames; 1080p k_blur 0.797, MFMA form 0.825 (2) / 0.772 (4) per 256 frames.
The texture-addresser model (profiles/r04_ta_shapes.txt) priced it at -27 %: the kernel is not bound there.  NOT in the product.
Applies to the sources of commit 5b6d869 (run from the repository root: python tools/experiments/patches/r05_blur_mfma_group_staging.py);
a record of what was measured, not maintained against later edits."""
p='orb_slam_amd/csrc/orbx_kernels.hip'
s=open(p).read()

def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (s.count(old), old[:90])
    s = s.replace(old, new)

# --- constants
rep('''constexpr int MB_IN_CHUNKS = 2 * MB_TILES + 2;                       // 16-byte chunks per staged input row: the strip + 16 columns either side
constexpr int MB_IN_BYTES = 32 * MB_IN_CHUNKS * 16;                  // one input buffer: 32 rows
constexpr int MB_OUT_PITCH = MB_TILES * 32 + 16;                     // bytes per row of the staged output (80: conflict-free ds_write_b128 of a row per lane)
static_assert(MB_TILES == 2, "k_blur_mfma's lane maps are written for two tiles per strip");''',
'''// Round 5: the MB_WAVES waves of a workgroup take ADJACENT strips and stage TOGETHER — the texture addresser prices a vector-memory
// instruction by the row pieces it touches (profiles/r04_ta_shapes.txt: 96-byte input rows 27.6 B/clk/CU, 64-byte output rows 11.4), and
// this kernel sat at 0.66-0.75 of it.  One shared input tile per step (32 rows x (64 MB_WAVES + 32) bytes: 160- / 288-byte pieces for 2 / 4
// waves) and one shared output block (32 rows x 64 MB_WAVES bytes, two of them: the block of step s - 1 leaves while step s fills the
// other).  Price: one workgroup barrier per step (it replaces the wave's own vmcnt(0) wait) and idle waves where a level's strips do not
// fill its last group.
constexpr int MB_IN_CHUNKS = 4 * MB_WAVES + 2;                       // 16-byte chunks per staged input row: the group's strips + 16 columns either side
constexpr int MB_IN_INSTR = (32 * MB_IN_CHUNKS + 64 * MB_WAVES - 1) / (64 * MB_WAVES);   // LDS-DMA instructions per wave and row tile
constexpr int MB_IN_BYTES = MB_IN_INSTR * MB_WAVES * 1024;           // one input buffer: 32 rows, rounded up to whole instructions (64 chunks each)
constexpr int MB_OUT_PITCH = MB_WAVES * 64 + 16;                     // bytes per row of the staged output block
constexpr int MB_OUT_BYTES = 32 * MB_OUT_PITCH;
static_assert(MB_TILES == 2, "k_blur_mfma's lane maps are written for two tiles per strip");
static_assert(MB_WAVES == 2 || MB_WAVES == 4, "chunk -> (row, chunk) divisions below are written for 10 or 18 chunks per row");
__device__ __forceinline__ int mb_chunk_row(int q) { return MB_WAVES == 2 ? (q * 205) >> 11 : (q * 3641) >> 16; }   // q / MB_IN_CHUNKS (exact for every q staged)''')

rep('''    __shared__ __attribute__((aligned(16))) uint8_t mb_in0[MB_WAVES * MB_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_in1[MB_WAVES * MB_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_out[MB_WAVES * 32 * MB_OUT_PITCH];''',
'''    __shared__ __attribute__((aligned(16))) uint8_t mb_in0[MB_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_in1[MB_IN_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_out0[MB_OUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t mb_out1[MB_OUT_BYTES];''')

rep('''    int frame, wgi;
    if (!frame_item(b, blockIdx.x, (g.nmb_total + MB_WAVES - 1) / MB_WAVES, frame, wgi)) return;
    const int item = wgi * MB_WAVES + wave_id();
    if (item >= g.nmb_total) return;
    const int level = __builtin_amdgcn_readfirstlane(find_level(g.mb_bases, item));''',
'''    int frame, item;
    if (!frame_item(b, blockIdx.x, g.nmb_total, frame, item)) return;          // one workgroup per (band, group of MB_WAVES strips)
    const int wv = wave_id();
    const int level = __builtin_amdgcn_readfirstlane(find_level(g.mb_bases, item));''')

rep('''    const int band = (item - L.mb_base) / L.mb_strips;           // (wave-uniform)
    const int X0 = 64 * ((item - L.mb_base) - band * L.mb_strips);       // first column of the strip''',
'''    const int band = (item - L.mb_base) / L.mb_strips;           // (workgroup-uniform; mb_strips counts GROUPS of MB_WAVES strips)
    const int XG = 64 * MB_WAVES * ((item - L.mb_base) - band * L.mb_strips);       // first column of the group
    const int X0 = XG + 64 * wv;                                 // first column of this wave's strip
    const bool strip_on = X0 < w;                                // (a wave beyond the level stages and flushes with the others, nothing else)''')

rep('''    uint8_t* const in0 = mb_in0 + wave_id() * MB_IN_BYTES;
    uint8_t* const in1 = mb_in1 + wave_id() * MB_IN_BYTES;
    uint8_t* const obuf = mb_out + wave_id() * 32 * MB_OUT_PITCH;
''','''    uint8_t* const in0 = mb_in0;
    uint8_t* const in1 = mb_in1;
''')

# DMA
rep('''    // LDS-DMA of one row tile: 32 rows x MB_IN_CHUNKS chunks, chunk q = 64 n + lane of the buffer = row q / 6, chunk q % 6 of the row
    int dma_c[3];
#pragma unroll
    for (int n = 0; n < 3; n++) {
        const int q = 64 * n + lane, r = (q * 171) >> 10;        // q / 6 for q < 192
        const int ca = (X0 >> 4) - 1 + (q - 6 * r);              // absolute chunk of the row; chunks outside it are clamped (no tap reaches them)''',
'''    // LDS-DMA of one row tile: 32 rows x MB_IN_CHUNKS chunks; instruction n of wave wv carries the chunks q = 64 (MB_WAVES n + wv) + lane
    // of the buffer = row q / MB_IN_CHUNKS, chunk q % MB_IN_CHUNKS of the row (the last instructions run past row 31: clamped rows into the
    // buffer's padding)
    int dma_c[MB_IN_INSTR];
#pragma unroll
    for (int n = 0; n < MB_IN_INSTR; n++) {
        const int q = 64 * (MB_WAVES * n + wv) + lane, r = mb_chunk_row(q);
        const int ca = (XG >> 4) - 1 + (q - MB_IN_CHUNKS * r);   // absolute chunk of the row; chunks outside it are clamped (no tap reaches them)''')
rep('''    auto dma_tile = [&](int R, uint8_t* ibuf) {                  // rows R .. R + 31 (reflect-101; rows no tap reaches are clamped into the level)
#pragma unroll
        for (int n = 0; n < 3; n++) {
            int row = R + (((64 * n + lane) * 171) >> 10);       // (recomputed: the kernel sits at its 128-register budget)
            row = row < 0 ? -row : row;
            row = row >= h ? 2 * h - 2 - row : row;
            row = min(max(row, 0), h - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (__umul24((unsigned)row, sstride) + (unsigned)dma_c[n])), (lptr_t)(ibuf + 1024 * n), 16, 0, 0);
        }
    };''','''    auto dma_tile = [&](int R, uint8_t* ibuf) {                  // rows R .. R + 31 (reflect-101; rows no tap reaches are clamped into the level)
#pragma unroll
        for (int n = 0; n < MB_IN_INSTR; n++) {
            int row = R + mb_chunk_row(64 * (MB_WAVES * n + wv) + lane);       // (recomputed: the kernel sits at its 128-register budget)
            row = row < 0 ? -row : row;
            row = row >= h ? 2 * h - 2 - row : row;
            row = min(max(row, 0), h - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (__umul24((unsigned)row, sstride) + (unsigned)dma_c[n])), (lptr_t)(ibuf + 1024 * (MB_WAVES * n + wv)), 16, 0, 0);
        }
    };''')
# operand reads
rep('''        const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(ibuf + (m * MB_IN_CHUNKS + 2 * j + gg) * 16);
        v4i p1, p2;''','''        const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(ibuf + (m * MB_IN_CHUNKS + 4 * wv + 2 * j + gg) * 16);
        v4i p1, p2;''')
# prologue
rep('''    dma_tile(Ybeg + 3 - 32, in1);                                // rows Ybeg - 29 .. Ybeg + 2: the taps above the band's first output rows
    dma_tile(Ybeg + 3, in0);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");             // LDS-DMA returns in order: the first three instructions (buffer 1) have landed
    row_pass(in1, 0, phi[0], plo[0]);
    if (tile1) row_pass(in1, 1, phi[1], plo[1]);
    // flush of the staged output block (32 rows x 64 bytes): chunk q = 64 n + lane = row q / 4, chunk q % 4
    const int fl_row = lane >> 2, fl_c = 16 * (lane & 3);
    const bool fl_on = X0 + fl_c < (int)L.stride && X0 + fl_c < ((w + 15) & ~15);
    auto flush = [&](int Yb) {                                   // the block of output rows Yb .. Yb + 31 leaves as 64 contiguous bytes per row
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int row = 16 * n + fl_row, oy = Yb + row;
            const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(obuf + row * MB_OUT_PITCH + fl_c, 16));
            if (fl_on && oy < Yend) *reinterpret_cast<uint4*>(__builtin_assume_aligned(dst + (__umul24((unsigned)oy, (unsigned)L.stride) + (unsigned)(X0 + fl_c)), 16)) = v;
        }
    };
    auto step = [&](int Y0, const uint8_t* cur, uint8_t* nxt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this step's rows have landed (requested a step ago), the stores of the step before
        if (Y0 > Ybeg) flush(Y0 - 32);                                   // are done, the output tiles of the step before are in LDS''',
'''    dma_tile(Ybeg + 3 - 32, in1);                                // rows Ybeg - 29 .. Ybeg + 2: the taps above the band's first output rows
    dma_tile(Ybeg + 3, in0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MB_IN_INSTR) : "memory");      // LDS-DMA returns in order: this wave's instructions into buffer 1 have landed,
    __builtin_amdgcn_s_barrier();                                //  and behind the barrier everybody's
    if (strip_on) {
        row_pass(in1, 0, phi[0], plo[0]);
        if (tile1) row_pass(in1, 1, phi[1], plo[1]);
    }
    // flush of a staged output block (32 rows x 64 MB_WAVES bytes): instruction n of wave wv carries the chunks q = 64 (MB_WAVES n + wv) + lane
    // = row q / (4 MB_WAVES), chunk q % (4 MB_WAVES): whole rows of 128 / 256 bytes
    const int fl_c = 16 * (lane & (4 * MB_WAVES - 1));
    const bool fl_on = XG + fl_c < (int)L.stride && XG + fl_c < ((w + 15) & ~15);
    auto flush = [&](int Yb, const uint8_t* obuf) {              // the block of output rows Yb .. Yb + 31
#pragma unroll
        for (int n = 0; n < 2; n++) {
            const int row = (64 * (MB_WAVES * n + wv) + lane) / (4 * MB_WAVES), oy = Yb + row;
            const uint4 v = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(obuf + row * MB_OUT_PITCH + fl_c, 16));
            if (fl_on && oy < Yend) *reinterpret_cast<uint4*>(__builtin_assume_aligned(dst + (__umul24((unsigned)oy, (unsigned)L.stride) + (unsigned)(XG + fl_c)), 16)) = v;
        }
    };
    auto step = [&](int Y0, const uint8_t* cur, uint8_t* nxt, uint8_t* obuf, const uint8_t* oprev) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's share of this step's rows has landed (requested a step ago), its stores of
        __builtin_amdgcn_s_barrier();                                    // the step before are done, its output tiles of the step before are in LDS — and, behind
        if (Y0 > Ybeg) flush(Y0 - 32, oprev);                            // the barrier, everybody's: the block of the step before leaves''')
# after dma next: skip compute for idle waves
rep('''        // the operands of both tiles in one LDS round trip: chunks g, 2 + g, 4 + g of the lane's row — tile 0 takes the first two, tile 1 the
        // last two (the strip's second tile starts where the first one's second operand does)
        v4i pre[3];
        {
            const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(cur + (m * MB_IN_CHUNKS + gg) * 16);''',
'''        if (!strip_on) return;
        // the operands of both tiles in one LDS round trip: chunks g, 2 + g, 4 + g of the lane's row in this wave's strip — tile 0 takes the first
        // two, tile 1 the last two (the strip's second tile starts where the first one's second operand does)
        v4i pre[3];
        {
            const unsigned ra = (unsigned)(uintptr_t)(lptr_t)(cur + (m * MB_IN_CHUNKS + 4 * wv + gg) * 16);''')
rep('''                const unsigned wa = (unsigned)(uintptr_t)(lptr_t)(obuf + m * MB_OUT_PITCH + 32 * j + 16 * gg);
                const v4i ov = {(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(wa), "v"(ov) : "memory");
            };''','''                const unsigned wa = (unsigned)(uintptr_t)(lptr_t)(obuf + m * MB_OUT_PITCH + 64 * wv + 32 * j + 16 * gg);
                const v4i ov = {(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(wa), "v"(ov) : "memory");
            };''')
rep('''                const unsigned wa = (unsigned)(uintptr_t)(lptr_t)(obuf + m * MB_OUT_PITCH + 32 * j + 16 * gg);
                const v4i ov = {(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(wa), "v"(ov) : "memory");      // (read back by flush() behind the lgkmcnt(0) at the top of the next step)''',
'''                const unsigned wa = (unsigned)(uintptr_t)(lptr_t)(obuf + m * MB_OUT_PITCH + 64 * wv + 32 * j + 16 * gg);
                const v4i ov = {(int)o4[0], (int)o4[1], (int)o4[2], (int)o4[3]};
                asm volatile("ds_write_b128 %0, %1" :: "v"(wa), "v"(ov) : "memory");      // (read back by flush() behind the barrier at the top of the next step)''')
rep('''    for (int Y0 = Ybeg; Y0 < Yend; Y0 += 64) {                   // two steps per trip: the buffer of each step is a named LDS object
        step(Y0, in0, in1);
        if (Y0 + 32 < Yend) step(Y0 + 32, in1, in0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    flush(Ybeg + (((Yend - 1 - Ybeg) >> 5) << 5));               // the last step's block
}''','''    for (int Y0 = Ybeg; Y0 < Yend; Y0 += 64) {                   // two steps per trip: the buffers of each step are named LDS objects
        step(Y0, in0, in1, mb_out0, mb_out1);
        if (Y0 + 32 < Yend) step(Y0 + 32, in1, in0, mb_out1, mb_out0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int last = (Yend - 1 - Ybeg) >> 5;                     // the last step's block
    flush(Ybeg + (last << 5), (last & 1) ? mb_out1 : mb_out0);
}''')
rep('''hipLaunchKernelGGL(k_blur_mfma, dim3(frame_item_blocks(b, (g.nmb_total + MB_WAVES - 1) / MB_WAVES)), dim3(MB_WAVES * 64), 0, st, b);''',
    '''hipLaunchKernelGGL(k_blur_mfma, dim3(frame_item_blocks(b, g.nmb_total)), dim3(MB_WAVES * 64), 0, st, b);''')
open(p,'w').write(s)

p='orb_slam_amd/csrc/orbx_geometry.hip'
s=open(p).read()
rep('''            L.mb_strips = (L.w + 32 * MB_TILES - 1) / (32 * MB_TILES);''','''            L.mb_strips = (L.w + 32 * MB_TILES * MB_WAVES - 1) / (32 * MB_TILES * MB_WAVES);      // groups of MB_WAVES adjacent 64-px strips: one workgroup each''')
open(p,'w').write(s)
p='orb_slam_amd/csrc/orbx_internal.h'
s=open(p).read()
rep('''    int mb_base, mb_n;     // k_blur_mfma work items of the level: mb_strips 64-px strips x mb_bands row bands of mb_band_steps 32-row steps, band-major''',
    '''    int mb_base, mb_n;     // k_blur_mfma work items (workgroups) of the level: mb_strips groups of MB_WAVES 64-px strips x mb_bands row bands of mb_band_steps 32-row steps, band-major''')
rep('''#ifndef ORBX_MB_WAVES
#define ORBX_MB_WAVES 4
#endif
constexpr int MB_WAVES = ORBX_MB_WAVES;     // ... strips (waves) per workgroup''','''#ifndef ORBX_MB_WAVES
#define ORBX_MB_WAVES 2
#endif
constexpr int MB_WAVES = ORBX_MB_WAVES;     // ... adjacent strips (waves) per workgroup, staged together (2 or 4)''')
open(p,'w').write(s)
