"""The basefc fold (keys -> count matrix) on the GPU: the partition fold (csrc/fold_partition.h: two-level partition + one LDS
pass per work item, no sort) against the oracle and against the radix-sort fold it replaces, with the knobs that force every
path on small inputs: XCK_FOLD_C (page size: small pages -> many work items, "big" cells, level 2, hand-overs to the sort fold),
XCK_FOLD_LGG (cell groups per row), XCK_FOLD=sort (the radix-sort fold).  Reference semantics: one set per (region, cell),
xcltk/rdr/fc/mcount.py:34-54."""
import os

import numpy as np
import pytest

import util
from fuzz_cases import make_case
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa

pytestmark = pytest.mark.gpu


@pytest.fixture
def fold_env():
    saved = {k: os.environ.get(k) for k in ("XCK_FOLD", "XCK_FOLD_C", "XCK_FOLD_LGG", "XCK_PILEUP_SORT", "XCK_PILEUP_ITEM_SORT", "XCK_PILEUP_HAP")}
    yield os.environ
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _run(names, regions, n_cells, batches, **filt):
    kw = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9)
    kw.update(filt)
    eng = Engine(capi.XCK_MODE_BASEFC, names, regions, n_cells, **kw)
    try:
        for b, _ in batches:
            eng.push(b)
        got = eng.finish()
        got = {"count": tuple(np.array(a) for a in got["count"])}
        st = eng.stats()
    finally:
        eng.close()
    return got, st


@pytest.mark.parametrize("page,lgg", [(1024, 8), (64, 8), (16, 3), (4, 0), (2, 8)])
@pytest.mark.parametrize("seed", [1000, 1003, 1007, 1011, 1016, 1052, 1101])
def test_partition_fold_matches_oracle_on_random_cases(seed, page, lgg, fold_env):
    names, regions, snps, n_cells, batches, fc, baf, flags = make_case(seed)
    fold_env["XCK_FOLD_C"], fold_env["XCK_FOLD_LGG"] = str(page), str(lgg)
    fold_env.pop("XCK_FOLD", None)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], n_cells, batches, flags=flags, **fc)
    util.assert_coo_equal(got, exp, ["count"])
    if st["key_bits"] == 64 and len(exp["count"][0]):
        assert st["fold_path"] in (1, 2)              # (2: a (region, cell) with more keys than a work item of this page size holds)


def _workload(n_reads, n_cells, n_genes, seed, span=3000000):
    regions, snps, names = soa.make_tables(n_genes, 0, [span], seed=seed, max_len=200000)
    bs = soa.gen_reads(regions, names, n_reads, n_cells, seed=seed + 1)
    return names, regions, [util.batch_from_dict(b) for b in bs]


def test_partition_fold_is_the_default_and_equals_the_sort_fold(fold_env):
    """300 k reads over 300 genes x 2000 cells: several chunks per shard slice, merged work items, no big cell at the default page
    size; the two folds must give the same matrix, and the default must be the partition fold."""
    names, regions, batches = _workload(300000, 2000, 300, seed=21)
    for k in ("XCK_FOLD", "XCK_FOLD_C", "XCK_FOLD_LGG"):
        fold_env.pop(k, None)
    got, st = _run(names, regions, 2000, batches)
    assert st["fold_path"] == 1 and st["fold_fallbacks"] == 0 and len(got["count"][0]) > 10000
    fold_env["XCK_FOLD"] = "sort"
    ref, st2 = _run(names, regions, 2000, batches)
    assert st2["fold_path"] == 2
    util.assert_coo_equal(got, ref, ["count"])
    fold_env.pop("XCK_FOLD")
    for page, lgg in ((256, 8), (32, 5), (8, 11)):    # big cells, level 2 (sub-cells by the low cell bits)
        fold_env["XCK_FOLD_C"], fold_env["XCK_FOLD_LGG"] = str(page), str(lgg)
        alt, st3 = _run(names, regions, 2000, batches)
        util.assert_coo_equal(alt, ref, ["count"])
        assert st3["fold_path"] in (1, 2)


def test_hot_gene_goes_through_level_two(fold_env):
    """Three genes hold all reads: at a page size of 64 keys its cell groups are "big" and are partitioned once
    more by the low cell bits; with 4096 cells and 8 groups per row a sub-cell is one (gene, cell) of ~40 keys."""
    names, regions, batches = _workload(200000, 4096, 3, seed=33, span=400000)
    for k in ("XCK_FOLD",):
        fold_env.pop(k, None)
    fold_env["XCK_FOLD_C"], fold_env["XCK_FOLD_LGG"] = "64", "3"
    got, st = _run(names, regions, 4096, batches)
    fold_env["XCK_FOLD"] = "sort"
    ref, _ = _run(names, regions, 4096, batches)
    util.assert_coo_equal(got, ref, ["count"])
    assert st["fold_path"] == 1


def test_deep_cell_is_cut_by_umi_hash(fold_env):
    """Bulk-like input: ONE cell, so a (gene, cell) holds ~20 k distinct keys - more than a work item (2 x 1024 keys) holds: level 2 cuts
    such a cell into UMI-hash parts, the parts' counts are added into one entry (continuation entries across work items)."""
    import oracle as O
    names, regions, batches = _workload(120000, 1, 5, seed=44, span=300000)
    for k in ("XCK_FOLD", "XCK_FOLD_C", "XCK_FOLD_LGG"):
        fold_env.pop(k, None)
    cfg_kw = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9, min_count=1, min_maf=0, no_dup_hap=True)
    cfg, keep = O.make_config(capi.XCK_MODE_BASEFC, names, regions, [], 1, **cfg_kw)
    exp = O.run_oracle(cfg, [b for b, _ in batches])
    assert exp["count"][2].max() > 4096
    got, st = _run(names, regions, 1, batches)
    assert st["fold_path"] == 1 and st["fold_fallbacks"] == 0
    util.assert_coo_equal(got, exp, ["count"])
    for page in (256, 16):                            # more parts per cell, parts spread over many work items
        fold_env["XCK_FOLD_C"] = str(page)
        got, st = _run(names, regions, 1, batches)
        util.assert_coo_equal(got, exp, ["count"])
    # 3 cells, small pages: several deep cells inside one big cell group, continuation at nearly every item
    names, regions, batches = _workload(60000, 3, 4, seed=45, span=200000)
    cfg, keep = O.make_config(capi.XCK_MODE_BASEFC, names, regions, [], 3, **cfg_kw)
    exp = O.run_oracle(cfg, [b for b, _ in batches])
    for page, lgg in ((8, 0), (8, 2), (64, 1)):
        fold_env["XCK_FOLD_C"], fold_env["XCK_FOLD_LGG"] = str(page), str(lgg)
        got, st = _run(names, regions, 3, batches)
        util.assert_coo_equal(got, exp, ["count"])
        assert st["fold_path"] in (1, 2)


def test_unplaceable_input_hands_over_to_the_sort_fold(fold_env):
    """A page size of ONE key: the hash parts of a deep cell collide, some part holds more than a work item may (2 keys), the partition
    fold reports it before it has touched the shard slices and the radix-sort fold produces the matrix."""
    import oracle as O
    names, regions, batches = _workload(120000, 1, 5, seed=44, span=300000)
    fold_env.pop("XCK_FOLD", None); fold_env.pop("XCK_FOLD_LGG", None)
    fold_env["XCK_FOLD_C"] = "1"
    got, st = _run(names, regions, 1, batches)
    assert st["fold_path"] == 2 and st["fold_fallbacks"] == 1
    cfg_kw = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9, min_count=1, min_maf=0, no_dup_hap=True)
    cfg, keep = O.make_config(capi.XCK_MODE_BASEFC, names, regions, [], 1, **cfg_kw)
    exp = O.run_oracle(cfg, [b for b, _ in batches])
    util.assert_coo_equal(got, exp, ["count"])


def test_pileup_hits_sorted_by_partition_equal_the_radix_sort(fold_env):
    """The pileup's (key, value) hits: row partition + one LDS sort per item (the default) against the library radix sort
    (XCK_PILEUP_SORT=radix) and the oracle, at the default page size and at small ones, with both item sorts."""
    regions, snps, names = soa.make_tables(120, 4000, [1500000], seed=51, max_len=150000)
    bs = soa.gen_reads(regions, names, 150000, 300, seed=52)
    batches = [util.batch_from_dict(b) for b in bs]
    for k in ("XCK_FOLD", "XCK_FOLD_C", "XCK_FOLD_LGG", "XCK_PILEUP_SORT", "XCK_PILEUP_ITEM_SORT", "XCK_PILEUP_HAP"):
        fold_env.pop(k, None)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 300, batches)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    assert st["pileup_sort_path"] == 1 and st["pileup_sort2_path"] == 1 and len(exp["dp"][0]) > 1000
    for item_sort in ("radix", "bitonic"):              # the LDS radix sort of an item (default) and the bitonic network
        fold_env["XCK_PILEUP_ITEM_SORT"] = item_sort
        for page in ("1024", "64", "8"):
            fold_env["XCK_FOLD_C"] = page
            got, _, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 300, batches)
            util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
            assert st["pileup_sort_path"] in (1, 2)
    fold_env.pop("XCK_PILEUP_ITEM_SORT")
    # region-level items: "values" = the haplotype class in a value word beside the key (default: in two free bits of the UMI field);
    # "sorted" = items sorted completely + k_hap_class / k_hap_sum, instead of k_hap_items
    for hap, paths in (("values", (1, 2)), ("sorted", (3, 2))):
        fold_env["XCK_PILEUP_HAP"] = hap
        for page in ("1024", "8"):
            fold_env["XCK_FOLD_C"] = page
            got, _, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 300, batches)
            util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
            assert st["pileup_sort2_path"] in paths
    fold_env.pop("XCK_FOLD_C"); fold_env.pop("XCK_PILEUP_HAP")
    fold_env["XCK_PILEUP_SORT"] = "radix"
    got, _, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 300, batches)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    assert st["pileup_sort_path"] == 2 and st["pileup_sort2_path"] == 2


def test_uneven_cells_refine_the_level_two_geometry(fold_env):
    """Well-based data: every cell has its own hot genes, so the cells of a group are far from even - here 70 % of the reads of 64 cells
    sit in cell 5.  With 4 cell groups per row and pages of 64 keys the group that holds cell 5 comes out "big", its sub-cells (sized for
    even cells) overflow, and the fold asks for a finer geometry of that group instead of handing over to the radix fold."""
    import oracle as O
    regions, snps, names = soa.make_tables(6, 0, [400000], seed=61, max_len=150000)
    bs = soa.gen_reads(regions, names, 120000, 64, seed=62)
    rng = np.random.default_rng(63)
    for b in bs:
        c = b["cell"]
        hot = (rng.random(len(c)) < 0.7) & (c >= 0)
        c[hot] = 5
    batches = [util.batch_from_dict(b) for b in bs]
    cfg_kw = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9, min_count=1, min_maf=0, no_dup_hap=True)
    cfg, keep = O.make_config(capi.XCK_MODE_BASEFC, names, regions, [], 64, **cfg_kw)
    exp = O.run_oracle(cfg, [b for b, _ in batches])
    fold_env.pop("XCK_FOLD", None)
    fold_env["XCK_FOLD_C"], fold_env["XCK_FOLD_LGG"] = "64", "2"
    got, st = _run(names, regions, 64, batches)
    util.assert_coo_equal(got, exp, ["count"])
    assert st["fold_path"] == 1 and st["fold_refinements"] >= 1
