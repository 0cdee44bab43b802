"""GPU parity: HIP engine (through the C-ABI) vs the CPU oracle on identical SoA batches."""
import numpy as np
import pytest

import oracle as O
import util
from xcltk_amd import capi
from xcltk_amd.synth import soa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    regions, snps, names = soa.make_tables(600, 20000, soa.HG38_LENGTHS[:2], seed=11, max_len=200000)
    bs = soa.gen_reads(regions, names, 300000, 200, seed=12, max_batch=70000)
    return regions, snps, names, [util.batch_from_dict(b) for b in bs]


@pytest.mark.parametrize("min_include", [0.9, 0.5, 30, 0, 1.0, 91])
def test_basefc_include_modes(small, min_include):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, snps, 200, batches,
                                         min_include=min_include)
    assert len(exp["count"][0]) > 1000
    util.assert_coo_equal(got, exp, ["count"])


@pytest.mark.parametrize("filt", [dict(), dict(min_mapq=0), dict(excl_flag=1796), dict(incl_flag=16),
                                  dict(min_len=60), dict(excl_flag=0, min_mapq=2)])
def test_basefc_filters(small, filt):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, snps, 200, batches, **filt)
    util.assert_coo_equal(got, exp, ["count"])


def test_basefc_key128(small):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, snps, 200, batches,
                                         flags=capi.XCK_F_FORCE_KEY128)
    assert st["key_bits"] == 128
    util.assert_coo_equal(got, exp, ["count"])


@pytest.mark.parametrize("opts", [dict(), dict(no_dup_hap=False), dict(min_count=3, min_maf=0.1),
                                  dict(min_count=11, min_maf=0.1), dict(min_mapq=0, excl_flag=0)])
def test_baf(small, opts):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 200, batches, **opts)
    if not opts.get("min_count", 0) > 5:
        assert len(exp["dp"][0]) > 500
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_baf_key128(small):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 200, batches,
                                         flags=capi.XCK_F_FORCE_KEY128)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_frac_divide_matches_host():
    """m/float(n) < v on the GPU (IEEE double divide) vs the host for every m <= n <= 160 at
    awkward thresholds: single-base reads with crafted CIGARs."""
    lib = O.load_oracle()
    for v in (0.9, 0.1, 1.0 / 3.0, 0.7, 0.5, 0.30000000000000004):
        rows = []
        for n in range(1, 161):
            for m in range(0, n + 1):
                rows.append((m, n))
        # region [1001, 2000] 1-based; read with n aligned bases of which m inside: start at 1000 - (n - m)
        regions = [("1", 1001, 2000, "r")]
        names = ["1"]
        pos = np.array([1000 - (n - m) for m, n in rows], dtype=np.int32)
        order = np.argsort(pos, kind="stable")
        pos = pos[order]
        nn = np.array([rows[i][1] for i in order], dtype=np.uint32)
        mm = np.array([rows[i][0] for i in order], dtype=np.int32)
        keep = mm > 0                       # reads with m = 0 do not overlap the region at all
        pos, nn, mm = pos[keep], nn[keep], mm[keep]
        k = len(pos)
        b = capi.make_batch(0, 0, pos, np.zeros(k, np.uint16), np.full(k, 60, np.uint8), np.zeros(k, np.int32),
                            np.arange(k, dtype=np.uint64) + 1000, np.arange(k + 1, dtype=np.uint32),
                            (nn << 4).astype(np.uint32))
        got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, [b], min_include=v,
                                             min_len=0)
        want = sum(1 for m, n in zip(mm, nn) if not lib.xo_frac_drop(int(m), int(n), v))
        assert int(exp["count"][2].sum()) == want
        util.assert_coo_equal(got, exp, ["count"])


@pytest.fixture(scope="module")
def large():
    from xcltk_amd.synth import soa as _soa
    regions, snps, names = _soa.make_tables(4000, 60000, _soa.HG38_LENGTHS[:4], seed=21, max_len=300000)
    bs = _soa.gen_reads(regions, names, 3000000, 1500, seed=22, max_batch=700000)
    return regions, snps, names, [util.batch_from_dict(b) for b in bs]


@pytest.mark.parametrize("mode,mats", [(capi.XCK_MODE_BASEFC, ["count"]), (capi.XCK_MODE_BAF, ["ad", "dp", "oth"])])
def test_large_scale_parity(large, mode, mats):
    """3 M reads / thousands of tiles: catches races and tile-boundary bugs that tiny inputs cannot
    (every LDS flush / spill path and the sharded cursors are exercised)."""
    regions, snps, names, batches = large
    got, exp, st = util.engine_vs_oracle(mode, names, regions, snps, 1500, batches)
    assert st["n_hits"] > 100000
    util.assert_coo_equal(got, exp, mats)
    # determinism: a second run gives the identical result and the identical accepted-pair count
    got2, _, st2 = util.engine_vs_oracle(mode, names, regions, snps, 1500, batches)
    util.assert_coo_equal(got2, exp, mats)
    assert st2["n_hits"] == st["n_hits"]
