#!/usr/bin/env python3
"""Builds the INSTRUMENTED library tools/fast_prof.py reads (not part of the product): a scratch copy of the sources with s_memtime marks
around the phases of k_fast_cells (-> build_variants/prof) or k_describe (-> build_variants/profd).  Every wave accumulates the
deltas in registers and writes ONE 12-dword record at its end (slot = workgroup * waves + wave); a global atomic per mark measured
only its own contention.  usage: python tools/build_prof_variant.py fast|describe"""
import os, shutil, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_FILES = {"fast": os.path.join("orb_slam_amd", "csrc", "k_fast.hip"),            # (round 6: one translation unit per stage)
                "describe": os.path.join("orb_slam_amd", "csrc", "k_describe.hip")}    # the blurred-plane form (ORBX_BLUR_ON_DEMAND=0); k_describe_od has no marks yet

DECL = ("constexpr int PROF_SLOTS = 1 << 19;\n__device__ unsigned g_fast_prof[PROF_SLOTS * 12];\n"
        "#define PROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); pacc[i] += (unsigned)(t_ - tprev); tprev = t_; } while (0)\n")
INIT = "    unsigned long long tprev = __builtin_readcyclecounter();\n    unsigned pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};\n"
RECORD = ("    { const unsigned slot = (blockIdx.x * %s + %s) & (PROF_SLOTS - 1); if (lane < 10) { unsigned v = 0;\n#pragma unroll\n"
          "      for (int kq = 0; kq < 10; kq++) if (lane == kq) v = pacc[kq];\n      g_fast_prof[slot * 12 + lane] = v; }"
          " if (lane == 10) g_fast_prof[slot * 12 + 10] = 1u; if (lane == 11) g_fast_prof[slot * 12 + 11] = (unsigned)level; }\n")
FETCH = ("\nextern \"C\" int orbx_debug_fast_prof(unsigned* out, int reset) {\n"
         "    if (reset) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(orbx::g_fast_prof)) != hipSuccess) return -1; return (int)hipMemset(p, 0, sizeof(unsigned) * orbx::PROF_SLOTS * 12); }\n"
         "    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(orbx::g_fast_prof), sizeof(unsigned) * orbx::PROF_SLOTS * 12);\n}\n")


def patch_source(s, which):
    """-> the kernel source with the marks of `which` ("fast" | "describe"); every anchor must occur exactly once (tests/test_tools.py)"""
    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, (s.count(old), old[:80])
        s = s.replace(old, new)
    if which == "fast":
        rep("struct FastHdr {", DECL + "struct FastHdr {")
        rep("    const int cw = bg.x1 - bg.x0 + 1, ch = bg.ey1 - bg.ey0 + 1;      // scored rectangle", INIT + "    const int cw = bg.x1 - bg.x0 + 1, ch = bg.ey1 - bg.ey0 + 1;      // scored rectangle")
        rep("        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // the DMA writes of THIS wave have landed; the barrier below covers the others",
            "        PROF(0);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        PROF(1);")
        rep("    __syncthreads();\n\n    // Threshold of the current pass", "    __syncthreads();\n    PROF(2);\n\n    // Threshold of the current pass")
        rep("    // Drain this wave's queues.", "    PROF(3);\n    // Drain this wave's queues.")
        rep("    if (n3 > Q3CAP && lane == 0) hdr->overflow = 1;\n    __syncthreads();", "    if (n3 > Q3CAP && lane == 0) hdr->overflow = 1;\n    PROF(4);\n    __syncthreads();\n    PROF(5);")
        rep("    __syncthreads();\n    if (tmin <= 7 || __builtin_amdgcn_readfirstlane(hdr->n_hi) > 3) break;",
            "    PROF(6);\n    __syncthreads();\n    PROF(7);\n    if (tmin <= 7 || __builtin_amdgcn_readfirstlane(hdr->n_hi) > 3) break;")
        rep("    if (tid == 0) {\n        CellState st;\n        st.n_all = run_base;", "    PROF(8);\n" + RECORD % ("NW", "wave") + "    if (tid == 0) {\n        CellState st;\n        st.n_all = run_base;")
    elif which == "describe":
        rep("constexpr int DESC_KPW = 4;", DECL + "constexpr int DESC_KPW = 4;")
        rep("    const int32_t* counts = b.level_count + frame * MAX_LEVELS;", INIT + "    const int32_t* counts = b.level_count + frame * MAX_LEVELS;")
        rep("    if (!ORBX_DESC_LATE_BARRIER) __syncthreads();\n    if (quad == 0 && lane == 0) {", "    PROF(0);\n    if (!ORBX_DESC_LATE_BARRIER) __syncthreads();\n    if (quad == 0 && lane == 0) {")
        rep("    if (ORBX_DESC_LATE_BARRIER) {\n", "    PROF(1);\n    if (ORBX_DESC_LATE_BARRIER) {\n")
        rep("    // IC_Angle on the unblurred level (:705-706 run before the blur)\n", "    PROF(2);\n    // IC_Angle on the unblurred level (:705-706 run before the blur)\n")
        rep("    const float angle = fast_atan2_deg((float)m01, (float)m10);", "    PROF(3);\n    const float angle = fast_atan2_deg((float)m01, (float)m10);")
        rep("    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");            // the wave's window DMA has landed (issued before IC_Angle)\n",
            "    asm volatile(\"\" :: \"v\"(sn), \"v\"(cs));\n    PROF(4);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    PROF(5);\n")
        rep("    uint32_t half = mybits;\n", "    asm volatile(\"\" :: \"v\"(mybits));\n    PROF(6);\n    uint32_t half = mybits;\n")
        rep("    if (!valid) return;\n    const int out_idx = out_base + k;", "    asm volatile(\"\" :: \"v\"(half));\n    PROF(7);\n" + RECORD % ("DESC_WAVES", "wave_id()") + "    if (!valid) return;\n    const int out_idx = out_base + k;")
    else:
        raise ValueError(which)
    return s + FETCH


def main(which):
    tmp = tempfile.mkdtemp(prefix="orbx_prof_")
    for d in ("Makefile", "include", "orb_slam_amd", "oracle"):
        src = os.path.join(R, d)
        (shutil.copytree if os.path.isdir(src) else shutil.copy)(src, os.path.join(tmp, d))
    p = os.path.join(tmp, KERNEL_FILES[which])
    patched = patch_source(open(p).read(), which)        # (read BEFORE the file is opened for writing)
    open(p, "w").write(patched)
    lib = os.path.join(tmp, "orb_slam_amd/liborbx.so")
    if os.path.exists(lib):
        os.remove(lib)
    subprocess.check_call(["make", "-C", tmp, "orb_slam_amd/liborbx.so"], stdout=subprocess.DEVNULL)
    dst = os.path.join(R, "build_variants", "prof" if which == "fast" else "profd")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(lib, os.path.join(dst, "liborbx.so"))
    shutil.rmtree(tmp)
    print(os.path.join(dst, "liborbx.so"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "fast")
