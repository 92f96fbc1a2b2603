"""Tooling that patches the kernel sources must keep finding its anchors (no GPU, no compile)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_prof_variant_anchors_exist_exactly_once():
    import build_prof_variant as b
    for which, marks in (("fast", 9), ("describe", 8)):
        src = open(os.path.join(ROOT, b.KERNEL_FILES[which])).read()
        out = b.patch_source(src, which)                 # asserts every anchor occurs exactly once
        assert out.count("PROF(") == marks + 1           # the marks + the macro's definition
        assert "orbx_debug_fast_prof" in out and "orbx_debug_fast_prof" not in src      # the product exports no such symbol


def test_experiment_index_lists_every_call():
    """tools/experiments/r0N_calls.sh hold one `n)` case per GPU call; README.md indexes every one of round 4's"""
    import re
    d = os.path.join(ROOT, "tools", "experiments")
    calls = sorted(int(m) for m in re.findall(r"^(\d+)\)", open(os.path.join(d, "r04_calls.sh")).read(), re.M))
    assert calls and calls[-1] == 51
    readme = open(os.path.join(d, "README.md")).read()
    listed = set()
    for line in readme.splitlines():
        if not line.startswith("| ") or line.startswith("| call") or line.startswith("|---"):
            continue
        for part in line.split("|")[1].replace(" ", "").split(","):
            if "–" in part:
                a, z = part.split("–")
                listed.update(range(int(a), int(z) + 1))
            elif part.isdigit():
                listed.add(int(part))
    assert set(calls) <= listed, sorted(set(calls) - listed)
    for f in os.listdir(d):                      # no script hard-codes the author's checkout
        if f.endswith(".sh"):
            assert "/root/repo" not in open(os.path.join(d, f)).read(), f
