#!/usr/bin/env python3
"""Randomised engine-vs-oracle parity (kernel level, through the C-ABI): random tables, read mixes, batch cuts, filters and
key widths.  usage: fuzz_parity.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from xcltk_amd import capi

from fuzz_cases import make_case


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + n_cases):
        names, regions, snps, n_cells, batches, fc, baf, flags = make_case(seed)
        for mode, kw, mats, tabs in ((capi.XCK_MODE_BASEFC, fc, ["count"], []), (capi.XCK_MODE_BAF, baf, ["ad", "dp", "oth"], snps)):
            try:
                got, exp, st = util.engine_vs_oracle(mode, names, regions, tabs, n_cells, batches, flags=flags, **kw)
                util.assert_coo_equal(got, exp, mats)
                print("seed %d mode %d ok  reads=%d hits=%d nnz=%s key_bits=%d" % (seed, mode, st["n_reads"], st["n_hits"], [len(got[m][0]) for m in mats], st["key_bits"]), flush=True)
            except Exception as e:
                bad += 1
                print("seed %d mode %d FAIL %s %s (opts %s, flags %d)" % (seed, mode, type(e).__name__, e, kw, flags), flush=True)
    print("%d cases, %d failures, %.1fs" % (n_cases, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
