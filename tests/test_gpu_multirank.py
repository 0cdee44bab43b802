"""Contig-sharded multi-rank run of the real front-ends (two ranks sharing the one GPU of the
test box, gloo for the exchange): rank 0's gathered output must equal the reference output."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


@pytest.mark.parametrize("name", ["multibam_basefc", "multibam_baf", "special_baf", "c1_basefc_default"])
def test_two_ranks_match_reference(name, tmp_path):
    env = dict(os.environ, XCK_DIST_BACKEND="gloo", XCK_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(),
                        os.path.join(ROOT, "tests", "dist_worker.py"), name, str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    if "MULTIRANK_OK " + name not in r.stdout:
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "multirank_fail_%s.log" % name), "w") as fp:
            fp.write(r.stdout)
    tb = [ln for ln in r.stdout.splitlines() if "Error" in ln or "error" in ln or "File \"/" in ln]
    assert "MULTIRANK_OK " + name in r.stdout, "\n".join(tb[-40:])
