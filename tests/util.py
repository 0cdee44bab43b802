"""Shared helpers for the parity tests (tests only; may import oracle/)."""
import ctypes as C
import os

import numpy as np

import oracle as O
from xcltk_amd import capi
from xcltk_amd.engine import Engine

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tables(d):
    regions = O.load_regions(os.path.join(d, "regions.tsv"))
    snp_fn = os.path.join(d, "snps.tsv")
    snps = O.load_snps(snp_fn) if os.path.isfile(snp_fn) else []
    return regions, snps


def batch_from_dict(g):
    return capi.make_batch(g["contig"], g["ordinal_base"], g["pos"], g["flag"], g["mapq"], g["cell"],
                           g["umi"], g["cig_off"], g["cigar"], g.get("seq_off"), g.get("seq"))


def engine_vs_oracle(mode, names, regions, snps, n_cells, batches, flags=0, **filt):
    """Run the HIP engine and the C oracle on the same SoA batches; return both COO dicts."""
    kw = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9,
              min_count=1, min_maf=0, no_dup_hap=True)
    kw.update(filt)
    eng = Engine(mode, names, regions, n_cells, snps=snps if mode == capi.XCK_MODE_BAF else (),
                 flags=flags, **kw)
    try:
        for b, _ in batches:
            eng.push(b)
        got = eng.finish()
        stats = eng.stats()
    finally:
        eng.close()
    cfg, keep = O.make_config(mode, names, regions, snps if mode == capi.XCK_MODE_BAF else [], n_cells,
                              flags=flags, **kw)
    exp = O.run_oracle(cfg, [b for b, _ in batches])
    return got, exp, stats


def assert_coo_equal(got, exp, names):
    for k in names:
        g, e = got[k], exp[k]
        assert len(g[0]) == len(e[0]), "%s: nnz %d != %d" % (k, len(g[0]), len(e[0]))
        for j, what in enumerate(("row", "col", "val")):
            assert np.array_equal(g[j], e[j]), "%s.%s differs" % (k, what)
