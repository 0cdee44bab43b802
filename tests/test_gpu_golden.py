"""End-to-end drop-in parity on the GPU: this package's fc_wrapper / afc_wrapper / `basefc` CLI
(C++ BAM decoder -> HIP engine -> writers) must reproduce, byte for byte, the files the
unmodified reference wrote for the same inputs (tests/golden/)."""
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", util.list_cases())
def test_frontend_matches_reference_outputs(name, tmp_path):
    from xcltk_amd.baf.fc.main import afc_wrapper
    from xcltk_amd.rdr.fc.main import fc_main, fc_wrapper
    case, ddir, odir, exp = util.load_case(name, tmp_path)
    if "argv" in case:
        ret = fc_main(["xcltk", "basefc"] + case["argv"])
    elif case["kind"] == "basefc":
        ret = fc_wrapper(**case["kwargs"])
    else:
        ret = afc_wrapper(**case["kwargs"])
    assert ret == 0
    util.assert_dirs_equal(odir, exp)


def _read(d):
    import os
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}


@pytest.mark.parametrize("pair", [("c1_basefc_default", "c1_baf_allreg"), ("dense_basefc_default", "dense_baf_allreg_dup"),
                                  ("multibam_basefc", None), ("special_basefc", "special_baf")])
def test_fused_single_decode_matches_both_references(pair, tmp_path):
    """XCK_MODE_BOTH: one decode of the BAM feeds both pipelines; each output set must equal
    the reference's output for the corresponding separate command."""
    import os
    from xcltk_amd.fused import fused_wrapper
    fc_name, baf_name = pair
    case, ddir, odir, exp_fc = util.load_case(fc_name, tmp_path)
    kw = case["kwargs"]
    extra = {}
    if baf_name:
        bcase, _, _, exp_baf = util.load_case(baf_name, tmp_path)
        extra = {k: bcase["kwargs"][k] for k in ("no_dup_hap", "min_count", "min_maf") if k in bcase["kwargs"]}
        snp = bcase["kwargs"]["phased_snp_fn"]
    else:
        snp = os.path.join(ddir, "snps.tsv")
    out = str(tmp_path / "fused")
    ret = fused_wrapper(kw["sam_fn"], kw["barcode_fn"], kw["region_fn"], snp, out, ncores=2, **extra)
    assert ret == 0
    assert _read(os.path.join(out, "basefc")) == _read(exp_fc)
    if baf_name:
        assert _read(os.path.join(out, "baf")) == _read(exp_baf)
