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


@pytest.mark.parametrize("name,world", [("well_baf", 1), ("c1_basefc_noumi", 1), ("c1_baf_noumi", 2)])
def test_key_id_overflow_is_retried_with_128bit_keys(name, world, tmp_path):
    """Read names used as keys are interned; XCK_TEST_INTERN_LIMIT makes the id space of the 64-bit key layout overflow after
    40 ids (really: 2^25 or more).  The decoder then stops with XCK_E_CAPACITY and the front-end must count again with
    128-bit keys - on all ranks of a multi-rank run - and still write the reference's bytes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XCK_TEST_INTERN_LIMIT="40", XCK_DIST_BACKEND="gloo", XCK_DEVICE="0", MASTER_ADDR="127.0.0.1")
    worker = os.path.join(root, "tests", "dist_worker.py")
    if world == 1:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, worker, name, str(tmp_path)]
    else:
        from test_gpu_multirank import _free_port
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
               "--master-port", _free_port(), worker, name, str(tmp_path)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert "MULTIRANK_OK " + name in r.stdout, r.stdout[-3000:]
    assert "counting again with 128-bit keys" in r.stdout, r.stdout[-3000:]
