"""Randomised GPU parity: 24 seeded cases of tests/fuzz_cases.py, both modes, against the oracle (tools/fuzz_parity.py runs
more).  Seeds 1003 and 1052 once exposed an include-test shortcut that trusted the fetch span of unmapped-flagged reads."""
import pytest

import util
from fuzz_cases import make_case
from xcltk_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(1000, 1020)) + [1052, 1077, 1101, 1133])
def test_random_case_matches_oracle(seed):
    names, regions, snps, n_cells, batches, fc, baf, flags = make_case(seed)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], n_cells, batches, flags=flags, **fc)
    util.assert_coo_equal(got, exp, ["count"])
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, n_cells, batches, flags=flags, **baf)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
