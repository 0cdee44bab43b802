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
