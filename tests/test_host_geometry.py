"""Host logic of the product without a GPU: the geometry liborbx derives from (constructor args, image size) must equal
the oracle's (which follows the reference's float/double arithmetic) and the numbers SURVEY.md §8d derived independently."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam_amd import capi, synth


def test_vga_geometry_matches_survey():
    g = capi.geometry(640, 480, nfeatures=1000)
    assert [(l["w"], l["h"]) for l in g] == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert [l["quota"] for l in g] == [217, 181, 151, 126, 105, 87, 73, 60]
    assert [(l["grid_cols"], l["grid_rows"]) for l in g] == [(5, 6), (5, 6), (4, 5), (4, 5), (3, 4), (3, 4), (3, 4), (3, 4)]
    assert [(l["cell_w"], l["cell_h"]) for l in g] == [(122, 75), (101, 62), (103, 61), (85, 50), (93, 50), (75, 41), (61, 33), (49, 26)]
    # VGA-class grids use the 256-thread shape with 8192-pixel bands: only level 0 (128 x 81 = 10368-pixel cell views) is cut in two
    assert [l["n_bands"] for l in g] == [60, 30, 20, 20, 12, 12, 12, 12]


def test_hd_and_init_extractor_geometry():
    g = capi.geometry(1920, 1080, nfeatures=2000)
    assert [l["quota"] for l in g] == [434, 362, 302, 251, 209, 175, 145, 122]
    assert [(l["grid_cols"], l["grid_rows"]) for l in g] == [(6, 10), (6, 10), (5, 8), (5, 8), (4, 7), (4, 7), (4, 7), (3, 5)]
    assert g[0]["n_bands"] > g[0]["grid_cols"] * g[0]["grid_rows"]                 # 33k-px cells are cut into row bands
    gi = capi.geometry(640, 480, nfeatures=2000)                                  # the reference's init extractor
    assert [(l["grid_cols"], l["grid_rows"]) for l in gi] == [(8, 10), (7, 9), (6, 8), (6, 8), (5, 6), (5, 6), (4, 5), (4, 5)]


@pytest.mark.parametrize("w,h,kw", [(752, 480, dict(nfeatures=1000)), (641, 479, dict(nfeatures=500)), (320, 240, dict(nfeatures=300, nlevels=5)),
                                    (640, 480, dict(nfeatures=1000, scaleFactor=1.5, nlevels=4)), (3840, 2160, dict(nfeatures=4000)),
                                    (200, 600, dict(nfeatures=400, nlevels=4))])
def test_geometry_equals_oracle(w, h, kw):
    g = capi.geometry(w, h, **kw)
    o = orc.OracleExtractor(dumps=True, **kw)
    o(synth.frame(w, h, synth.FLAT, 0))
    assert [(l["w"], l["h"]) for l in g] == [o.level_size(i) for i in range(len(g))]
    assert [l["quota"] for l in g] == o.features_per_level()
    # the oracle's cell dump gives the grid: cells per level and their view sizes (cell + 6 px, clipped at the last row/col)
    cells = o.cells()
    for lvl, l in enumerate(g):
        mine = [c for c, _ in cells if c[0] == lvl]
        assert len(mine) <= l["grid_cols"] * l["grid_rows"]
        first = [c for c in mine if c[1] == 0 and c[2] == 0][0]
        if l["grid_cols"] > 1:
            assert first[6] == l["cell_w"] + 6
        if l["grid_rows"] > 1:
            assert first[7] == l["cell_h"] + 6


def test_geometry_errors():
    with pytest.raises(capi.OrbxError) as e:
        capi.geometry(100, 80)                       # top levels smaller than the 16-px border + FAST ring
    assert e.value.code == capi.ORBX_ERR_GEOMETRY
    with pytest.raises(capi.OrbxError) as e:
        capi.geometry(640, 480, nlevels=40)
    assert e.value.code == capi.ORBX_ERR_ARG


def test_rejections_are_classified():
    """ORBX_ERR_GEOMETRY only for inputs the reference itself cannot process; implementation limits are ORBX_ERR_CAPACITY"""
    for args in [dict(w=640, h=480, nfeatures=50, nlevels=8),          # level quota 3..11 -> levelCols = 0: the reference divides by zero
                 dict(w=100, h=80, nfeatures=500, nlevels=8),          # level 7 is 28x22: cell views with negative extent
                 dict(w=320, h=240, nfeatures=1000, scaleFactor=2.0, nlevels=5)]:
        with pytest.raises(capi.OrbxError) as e:
            capi.geometry(**args)
        assert e.value.code == capi.ORBX_ERR_GEOMETRY, args
    # cells 2084 / 4168 px wide and levels of 1170 / 2310 cells were refused until round 3 (limits of 2000 px and 1024 cells): accepted now
    assert capi.geometry(w=4200, h=900, nfeatures=40, nlevels=1)[0]["cell_w"] == 4168
    assert capi.geometry(w=4200, h=3000, nfeatures=60, nlevels=1)[0]["cell_w"] == 2084
    assert capi.geometry(w=4200, h=3000, nfeatures=40, nlevels=2)[0]["cell_w"] == 4168
    assert [l["grid_cols"] * l["grid_rows"] for l in capi.geometry(w=2080, h=1568, nfeatures=12000, nlevels=1)] == [2310]
    # the boundary of what is left: a staged band is two own rows + 2 halo + 6 ring rows of the cell's width and must stay below
    # 64 KiB (16-bit pixel offsets): 6500 px is accepted, 6501 is ORBX_ERR_CAPACITY (the reference has no such limit)
    assert capi.geometry(w=6532, h=4600, nfeatures=20, nlevels=1)[0]["cell_w"] == 6500
    with pytest.raises(capi.OrbxError) as e:
        capi.geometry(w=6533, h=4600, nfeatures=20, nlevels=1)
    assert e.value.code == capi.ORBX_ERR_CAPACITY


def test_magic_multiply_block_mapping_is_exact():
    """k_fast_cells maps a block index to (frame, band) with umulhi(n, floor(2^32 / d) + 1) and one correcting compare
    (orbx_device.h: frame_item_magic; the constant is DevGeom::nbands_magic, made by orbx_geometry.hip).  The arithmetic restated on
    32-bit integers: exact for every divisor and every block index a launch can have (boundaries of the quotient, powers of two, the
    largest grids), including the products staying below 2^32 as the device computes them."""
    rng = np.random.default_rng(3)
    U32 = 1 << 32
    ds = list(range(1, 300)) + [511, 512, 513, 1023, 1024, 1025, 4095, 4096, 16383, 16384, 65535, 65536, 262143] + [int(v) for v in rng.integers(2, 1 << 18, 300)]
    for d in ds:
        magic = (U32 // d + 1) if d > 1 else 0
        assert magic < U32
        ns = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 24) - 1, 1 << 24, (1 << 31) - d - 1} | {int(v) for v in rng.integers(0, (1 << 31) - d, 200)}
        ns |= {k * d + e for k in (1, 7, 1000, ((1 << 31) - d) // d - 1) for e in (-1, 0, 1) if 0 <= k * d + e < (1 << 31) - d}
        for n in ns:
            q = (n * magic) >> 32 if magic else n
            assert q in (n // d, n // d + 1)
            assert q * d < U32                       # the device's 32-bit product does not wrap
            if q * d > n:
                q -= 1
            assert q == n // d and n - q * d == n % d
