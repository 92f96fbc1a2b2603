#!/usr/bin/env python3
"""Randomised differential test of the matcher drop-in: the PRODUCT's ORB_SLAM::ORBmatcher (orb_slam_amd/cpp/ORBmatcher.cc, GPU scans) against the
REFERENCE's own src/ORBmatcher.cc, both behind the harness of oracle/ref_orbmatcher_wrap.cpp (oracle/_ref/lib{prod,ref}_orbmatcher.so).

Every case is one of the thirteen search signatures on a random problem (sizes from one feature to 1300, random radii / thresholds / level bands /
orientation check, crowded and sparse frames) at a random camera pose and similarity; each harness call goes to the reference first (on copies of
every array argument) and to the product second; the return value and every array the call could have written must be equal.  The reference is the only judge here (the oracle
restatement is pinned against the same reference library by tests/test_ref_pin_matcher.py on the CPU).

    python tools/fuzz_orbmatcher.py [cases] [seed] [real]  (GPU box; test infrastructure, not product; `real`: the product library built with
                                                            orb_slam_amd/cpp/ORBmatcherAccess.h, libprod_orbmatcher_realaccess.so)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import test_gpu_orbmatcher_dropin as drop  # noqa: E402
import test_ref_pin_matcher as trm  # noqa: E402

BIG = 2**31 - 1


def size(rng, lo=1):
    """mostly frame-sized, sometimes tiny"""
    r = rng.random()
    if r < 0.08:
        return int(rng.integers(lo, lo + 4))
    if r < 0.25:
        return int(rng.integers(lo, 200))
    return int(rng.integers(200, 1301))


def band(rng):
    if rng.random() < 0.5:
        return -1, BIG
    lo = int(rng.integers(0, 6))
    return lo, int(rng.integers(lo, 8))


def tri_kw(rng):
    kw = {}
    if rng.random() < 0.4:
        kw["line_noise"] = float(rng.choice([0.5, 1.0, 3.0]))
    if rng.random() < 0.3:
        kw["max_flips"] = int(rng.integers(10, 60))
    if rng.random() < 0.2:
        kw["p_mp1"] = kw["p_mp2"] = 0.0
    return kw


# name -> (argument maker, runs at general poses too)
CASES = {
    "test_search_by_projection_of_map_points": (lambda r, s: dict(seed=s, nt=size(r), nq=size(r, 0), th=float(r.choice([1.0, 3.0, 5.0, r.uniform(0.5, 6)])), crowd=bool(r.integers(2))), False),
    "test_window_search": (lambda r, s: dict(zip(("lo", "hi"), band(r)), seed=s, n1=size(r), n2=size(r), win=int(r.integers(5, 201)), check=bool(r.integers(2)), crowd=bool(r.integers(2))), False),
    "test_search_for_initialization": (lambda r, s: dict(seed=s, n1=size(r), n2=size(r), win=int(r.integers(5, 201)), check=bool(r.integers(2)), crowd=bool(r.integers(2))), False),
    "test_search_by_projection_from_last_frame": (lambda r, s: dict(seed=s, n1=size(r, 24), n2=size(r, 24), th=float(r.uniform(2, 20)), check=bool(r.integers(2)), crowd=bool(r.integers(2))), True),
    "test_search_by_sim3": (lambda r, s: dict(seed=s, n1=size(r), n2=size(r), th=float(r.uniform(2, 12)), crowd=bool(r.integers(2))), True),
    "test_search_by_projection_between_two_frames": (lambda r, s: dict(seed=s, n1=size(r), n2=size(r), win=int(r.integers(4, 80)), crowd=bool(r.integers(2))), True),
    "test_search_by_projection_from_keyframe": (lambda r, s: dict(seed=s, nkf=size(r, 24), n2=size(r, 24), th=float(r.uniform(2, 12)), orbdist=int(r.choice([50, 64, 100])), check=bool(r.integers(2)), crowd=bool(r.integers(2))), True),
    "test_search_by_projection_with_sim3_pose": (lambda r, s: dict(seed=s, nkf=size(r, 24), nq=size(r, 24), th=int(r.integers(2, 14)), crowd=bool(r.integers(2))), True),
    "test_fuse": (lambda r, s: dict(which=int(r.integers(2)), seed=s, nkf=size(r, 24), nq=size(r, 24), th=float(r.uniform(1.5, 5)), crowd=bool(r.integers(2))), True),
    "test_search_by_bow_keyframe_frame": (lambda r, s: dict(seed=s, n1=size(r), n2=size(r), check=bool(r.integers(2))), False),
    "test_search_by_bow_keyframe_keyframe": (lambda r, s: dict(seed=s, n1=size(r), n2=size(r), check=bool(r.integers(2))), False),
    "test_search_for_triangulation": (lambda r, s: dict(seed=s, n1=size(r), n2=size(r), check=bool(r.integers(2)), kw=tri_kw(r)), False),
}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6106
    import torch
    assert torch.cuda.is_available(), "the product side runs on the GPU"
    from orb_slam_amd import capi
    rng = np.random.default_rng(seed)
    real = len(sys.argv) > 3 and sys.argv[3] == "real"
    both = drop.Both(drop.PROD_REAL_ACCESS_PATH if real else drop.PROD_PATH)
    trm.ref = lambda: both
    trm.P = both.P
    names = sorted(CASES)
    per, posed, calls, matches, bad, unbuilt = {}, 0, 0, 0, [], []
    t0 = time.time()
    for i in range(cases):
        name = names[int(rng.integers(len(names)))]
        make, poses = CASES[name]
        kw = make(rng, int(rng.integers(1, 2**31 - 1)))
        general = poses and rng.random() < 0.7
        if general:
            angle, shift, scale = float(rng.uniform(0.02, 0.8)), float(rng.uniform(0.0005, 0.006)), float(rng.uniform(0.9, 1.1))
            both.set_pose(drop._pose(int(rng.integers(1 << 30)), angle, shift), scale)
            both.set_sim3(drop._pose(int(rng.integers(1 << 30)), angle / 2, shift / 2), 2.0 - scale)
            posed += 1
        else:
            both.set_pose(None)
            both.set_sim3(None)
        both.calls.clear()
        both.arrays.clear()
        try:
            getattr(trm, name)(**kw)
        except drop.DropInMismatch as e:
            bad.append({"case": i, "name": name, "args": {k: (v if not isinstance(v, (np.generic,)) else v.item()) for k, v in kw.items()}, "general_pose": general, "what": str(e)})
        except (ValueError, IndexError) as e:
            if both.calls:
                raise
            unbuilt.append({"name": name, "args": {k: str(v) for k, v in kw.items()}, "what": str(e)[:120]})      # the case's generator cannot build this size
        except AssertionError:
            pass        # the case's own assertions (oracle at the identity pose, sanity thresholds on match counts random sizes need not meet): the reference is the judge here
        per[name] = per.get(name, 0) + 1
        calls += len(both.calls)
        matches += sum(int(r) for c, r in both.calls if c.startswith("ref_search") or c in ("ref_window_search", "ref_fuse"))
    out = {"cases": cases, "seed": seed, "access_header": "orb_slam_amd/cpp/ORBmatcherAccess.h" if real else "oracle/matcherstub/access.h", "general_pose_cases": posed, "harness_calls_compared": calls, "matches_returned": matches, "by_case": per,
           "cases_the_generator_could_not_build": len(unbuilt), "unbuilt": unbuilt[:10], "mismatches": bad, "seconds": round(time.time() - t0, 1), "build": capi.build_id() if hasattr(capi, "build_id") else None}
    print(json.dumps(out))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
