"""Tooling that patches the kernel sources must keep finding its anchors (no GPU, no compile)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_prof_variant_anchors_exist_exactly_once():
    import build_prof_variant as b
    src = open(os.path.join(ROOT, b.KERNELS)).read()
    for which, marks in (("fast", 9), ("describe", 8)):
        out = b.patch_source(src, which)                 # asserts every anchor occurs exactly once
        assert out.count("PROF(") == marks + 1           # the marks + the macro's definition
        assert "orbx_debug_fast_prof" in out and "orbx_debug_fast_prof" not in src      # the product exports no such symbol


def test_experiment_index_lists_every_call_script():
    d = os.path.join(ROOT, "tools", "experiments")
    calls = sorted(int(f[4:-3]) for f in os.listdir(d) if f.startswith("call") and f.endswith(".sh"))
    readme = open(os.path.join(d, "README.md")).read()
    listed = set()
    for line in readme.splitlines():
        if not line.startswith("| ") or line.startswith("| call") or line.startswith("|---"):
            continue
        for part in line.split("|")[1].replace(" ", "").split(","):
            if "–" in part:
                a, z = part.split("–")
                listed.update(range(int(a), int(z) + 1))
            elif part:
                listed.add(int(part))
    assert set(calls) <= listed, sorted(set(calls) - listed)


def test_band_hint_patch_still_applies():
    """tools/experiments/band_hint_frame0.patch (NOTES.md 9.6: measured, waiting for a round with GPU budget to collect counters) must keep applying"""
    import subprocess
    r = subprocess.run(["git", "apply", "--check", "-p0", "tools/experiments/band_hint_frame0.patch"], cwd=ROOT, capture_output=True, text=True)
    if "not a git repository" in (r.stderr or "").lower():
        import pytest
        pytest.skip("no git metadata here")
    assert r.returncode == 0, r.stderr
