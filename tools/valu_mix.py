#!/usr/bin/env python3
"""VALU issue cost of the hot kernels' instruction mix (no GPU needed): compiles the kernels to gfx950 assembly, takes the static
histogram of VALU opcodes per kernel and prices every opcode with the issue cost measured by tools/microbench/valu_rate on an
MI355X (profiles/r01_valu_issue_rates.txt): 2 cycles per wave64 instruction for the plain VOP1/VOP2 ops (mov, not, add, sub, and,
or, xor, right shifts, f32 add/mul/fma), 4 cycles for everything else measured (min/max/med3, compares, cndmask, left shift, bfe,
perm, alignbyte, bcnt, dot2/dot4, sad, integer multiplies, conversions, readfirstlane, every three-operand fused op and every
SDWA / DPP form).  Unmeasured opcodes are priced at both
ends (2 and 4) and give the [lo, hi] interval.  Writes profiles/valu_mix.json, read by bench.py for `roofline_valu`.
The histogram is static (all code of the kernel counts once), so the result is an estimate of the dynamic mix."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam_amd import capi
FAST = {"v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_ashrrev_i32", "v_lshrrev_b32",
        "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_bitop3_b32", "v_add_u16", "v_max_u16"}   # + round 2: tools/microbench/valu_rate2
SLOW = {"v_and_or_b32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_lshlrev_b32", "v_bcnt_u32_b32", "v_perm_b32", "v_dot4_u32_u8",
        "v_dot2_u32_u16", "v_alignbyte_b32", "v_mul_lo_u32", "v_mad_u32_u24", "v_mad_i32_i24", "v_sad_u8", "v_cndmask_b32", "v_max3_u32", "v_min3_u32",
        "v_xad_u32", "v_lshl_or_b32", "v_add3_u32", "v_bfe_u32", "v_bfe_i32", "v_lshl_add_u32", "v_mul_i32_i24", "v_mul_u32_u24", "v_or3_b32", "v_med3_u32",
        "v_med3_i32", "v_mul_hi_u32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_rndne_f32", "v_cvt_f32_ubyte0",
        "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3", "v_readfirstlane_b32",
        "v_pk_add_u16", "v_pk_sub_i16", "v_pk_min_u16", "v_pk_max_u16", "v_pk_lshrrev_b16", "v_lerp_u8", "v_min_f32", "v_max_f32", "v_mbcnt_lo_u32_b32",
        "v_mbcnt_hi_u32_b32", "v_add_co_u32", "v_bfi_b32", "v_alignbit_b32", "v_cvt_pk_u8_f32", "v_xnor_b32", "v_dot4_i32_i8", "v_dot8_i32_i4"}   # round 2 measurements
STAGE = {"k_resize<true, true>": "pyramid", "k_fast_cells<true, 256, 1>": "fast_cells", "k_fast_cells<true, 256, 2>": "fast_cells_large",
         "k_quota": "quota", "k_cell_select": "cell_select", "k_level_select": "level_select", "k_blur_mfma": "blur", "k_blur<true, 32>": "blur_valu", "k_describe<false>": "describe_plane", "k_describe_od<false>": "describe",
         "k_match_batch_mfma4<4>": "match", "k_match_batch_mfma<4>": "match_int8"}
hist = {}
with tempfile.TemporaryDirectory() as td:
    for src in ("k_pyramid.hip", "k_fast.hip", "k_select.hip", "k_blur.hip", "k_describe.hip", "k_describe_od.hip", "orbm_match.hip"):
        asm = os.path.join(td, src + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-I" + ROOT + "/include",
                        "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + ROOT + "/orb_slam_amd/csrc", "--cuda-device-only", "-S", ROOT + "/orb_slam_amd/csrc/" + src, "-o", asm],
                       check=True, stderr=subprocess.DEVNULL)
        cur = None
        for line in open(asm):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                cur = next((st for key, st in STAGE.items() if re.search(r"::" + re.escape(key) + r"\(", dem)), None)
                continue
            if line.startswith(".Lfunc_end"):
                cur = None
            m = re.match(r"^\s+(v_[a-z0-9_]+)", line)
            if m and cur and not m.group(1).startswith("v_mfma"):          # (matrix-core instructions are not VALU work: SQ_INSTS_MFMA counts them)
                op = re.sub(r"_(e32|e64)$", "", m.group(1))
                if op.endswith("_sdwa") or op.endswith("_dpp"):
                    op = "sdwa/dpp form"            # measured: 4 cycles whatever the base opcode (v_mov_b32_dpp, v_and_b32_sdwa, v_add_u32_sdwa ...)
                if op.startswith("v_cmp"):
                    op = "v_cmp"
                hist.setdefault(cur, collections.Counter())[op] += 1
out = {"src_hash": capi.build_id(), "issue_cycles": {"fast_class": 2, "slow_class": 4, "measured_by": "tools/microbench/valu_rate, valu_rate2 (profiles/r01_valu_issue_rates.txt, r02_valu_issue_rates2.txt)"}, "kernels": {}}
for st, h in hist.items():
    n = sum(h.values())
    fast = sum(c for o, c in h.items() if o in FAST)
    slow = sum(c for o, c in h.items() if o in SLOW or o in ("v_cmp", "sdwa/dpp form"))
    unk = n - fast - slow
    out["kernels"][st] = {"static_valu_insts": n, "fast_frac": round(fast / n, 3), "slow_frac": round(slow / n, 3), "unmeasured_frac": round(unk / n, 3),
                          "cycles_per_inst_lo": round((2 * fast + 4 * slow + 2 * unk) / n, 3), "cycles_per_inst_hi": round((2 * fast + 4 * slow + 4 * unk) / n, 3),
                          "top": dict(h.most_common(12))}
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)
for st, k in out["kernels"].items():
    print("%-13s %5d insts  fast %.2f slow %.2f unmeasured %.2f  -> %.2f .. %.2f cycles per wave64 inst" % (
        st, k["static_valu_insts"], k["fast_frac"], k["slow_frac"], k["unmeasured_frac"], k["cycles_per_inst_lo"], k["cycles_per_inst_hi"]))
