import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import util
from fuzz_cases import make_case
from xcltk_amd import capi
for seed in [int(x) for x in sys.argv[1:]]:
    for label, kw in (("all", {}), ("no odd regions", dict(odd_regions=False)), ("no 70k cells", dict(many_cells=False)), ("no long reads", dict(long=False)), ("no pos -1", dict(neg_pos=False))):
        names, regions, snps, n_cells, batches, fc, baf, flags = make_case(seed, **kw)
        res = []
        for mode, o, mats, tabs in ((1, fc, ["count"], []), (2, baf, ["ad", "dp", "oth"], snps)):
            got, exp, st = util.engine_vs_oracle(mode, names, regions, tabs, n_cells, batches, flags=flags, **o)
            try:
                util.assert_coo_equal(got, exp, mats); res.append("ok")
            except AssertionError as e:
                res.append("FAIL(%s)" % str(e)[:40])
        print(seed, "%-16s" % label, res, "cells", n_cells, "regions", len(regions), flush=True)
