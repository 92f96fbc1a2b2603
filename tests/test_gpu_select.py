"""The wave-parallel introselect of the retainBest kernels must produce libstdc++'s EXACT permutation
(std::nth_element with the response-greater comparator), not merely a valid partition."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam_amd import capi

pytestmark = pytest.mark.gpu


def _check(resp, nth):
    got = capi.nth_element_perm(resp, nth)
    ref = orc.nth_element_perm(resp, nth)
    assert np.array_equal(got, ref), "n=%d nth=%d first diff at %d" % (len(resp), nth, int(np.argmax(got != ref)))


def test_random_lists_with_ties():
    rng = np.random.default_rng(0)
    for it in range(400):
        n = int(rng.integers(2, 2600))
        kind = it % 5
        if kind == 0:
            r = rng.integers(7, 60, n).astype(np.float32)          # FAST scores: small ints, dense ties
        elif kind == 1:
            r = rng.integers(7, 12, n).astype(np.float32)          # almost everything tied
        elif kind == 2:
            r = rng.random(n).astype(np.float32)                   # Harris-like, no ties
        elif kind == 3:
            r = rng.integers(0, 256, n).astype(np.float32)
        else:
            r = np.round(rng.normal(30, 8, n)).clip(7, 254).astype(np.float32)
        for nth in {1, n // 2, max(n - 1, 1), int(rng.integers(1, n)), min(n, max(1, n // 7))}:
            if 0 < nth < n:
                _check(r, nth)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 7, 63, 64, 65, 128, 1000, 2049, 5000])
def test_structured_lists(n):
    base = np.arange(n, dtype=np.float32)
    pats = [base, base[::-1].copy(), np.full(n, 20, np.float32), np.minimum(base, base[::-1]),     # sorted, reversed, constant, organ pipe
            np.where(np.arange(n) % 2 == 0, 50, 10).astype(np.float32), np.concatenate([base[: n // 2], base[: n - n // 2]]),
            (np.arange(n) % 3).astype(np.float32)]
    for r in pats:
        for nth in sorted({1, n // 3, n // 2, n - 1}):
            if 0 < nth < n:
                _check(r, nth)


def test_median_of_three_killer_hits_the_depth_limit():
    """Musser's median-of-3 killer drives introselect into its heap-select fallback (depth limit 2*lg n)."""
    def killer(n):
        a = np.zeros(n, np.float32)
        k = n // 2
        for i in range(1, k + 1):
            if i % 2 == 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        return a
    for n in (64, 256, 1024, 4096):
        for nth in (n // 2, n - 2, 1):
            _check(killer(n), nth)
            _check(-killer(n), nth)
