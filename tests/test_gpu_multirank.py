"""Contig-sharded multi-rank runs of the real front-ends (ranks sharing the one GPU of the test box over gloo; one GPU per rank over
RCCL where several are visible): the output files must equal the reference's, through both exchange forms."""
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


def _n_gpus():
    import torch
    return torch.cuda.device_count()                    # (counting devices does not initialise the GPU in this process)


@pytest.mark.parametrize("backend,world,exchange", [("gloo", 2, "sharded"), ("gloo", 2, "gather"), ("gloo", 4, "sharded"), ("nccl", 2, "sharded"), ("nccl", 2, "gather")])
@pytest.mark.parametrize("name", ["multibam_basefc", "multibam_baf", "special_baf", "c1_basefc_default", "phasing_baf_refcells"])
def test_ranks_match_reference(name, backend, world, exchange, tmp_path):
    """gloo: the ranks share the one GPU of the test box; nccl: one GPU per rank, collectives over RCCL (runs only where at least
    two GPUs are visible - the driver's multi-GPU node).  exchange "sharded" (default of the front-ends): every rank writes the lines
    of its own rows into the output files, the only collectives are all-reduces of text sizes / row presence; "gather"
    (XCK_DIST_GATHER=1): the per-rank sparse blocks travel to rank 0 (GPU to GPU with nccl), which merges and writes."""
    if backend == "nccl" and _n_gpus() < world:
        pytest.skip("the RCCL exchange needs %d visible GPUs" % world)
    if world == 4 and name not in ("multibam_basefc", "special_baf"):
        pytest.skip("world size 4 is rehearsed on two cases")
    if exchange == "gather" and name not in ("multibam_baf", "c1_basefc_default", "phasing_baf_refcells"):
        pytest.skip("the gather form is rehearsed on three cases")
    env = dict(os.environ, XCK_DIST_BACKEND=backend, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               XCK_DIST_GATHER="1" if exchange == "gather" else "0")
    if backend == "gloo":
        env["XCK_DEVICE"] = "0"
    else:
        env.pop("XCK_DEVICE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(),
                        os.path.join(ROOT, "tests", "dist_worker.py"), name, str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    if "MULTIRANK_OK " + name not in r.stdout:
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "multirank_fail_%s_%s.log" % (name, exchange)), "w") as fp:
            fp.write(r.stdout)
    tb = [ln for ln in r.stdout.splitlines() if "Error" in ln or "error" in ln or "File \"/" in ln]
    assert "MULTIRANK_OK " + name in r.stdout, "\n".join(tb[-40:])


def test_pipeline_pileup_then_counting_two_ranks(tmp_path):
    """`xcltk baf --snpvcf ... --phasedSNP ...` on two ranks (gloo, shared GPU): rank 0 writes the pileup directory of step 1, both
    ranks read it in step 3 - after waiting for the writer (xcltk_amd/baf/genotype.py pileup(): status all-reduce)."""
    env = dict(os.environ, XCK_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", XCK_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(),
                        os.path.join(ROOT, "tests", "dist_worker.py"), "pipeline_steps_1_3", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    tb = [ln for ln in r.stdout.splitlines() if "Error" in ln or "error" in ln or "File \"/" in ln]
    assert "MULTIRANK_OK pipeline_steps_1_3" in r.stdout, "\n".join(tb[-40:])


@pytest.mark.parametrize("exchange", ["sharded", "gather"])
@pytest.mark.parametrize("name", ["multibam_baf", "c1_basefc_default"])
def test_rccl_backend_with_a_world_of_one(name, exchange, tmp_path):
    """The RCCL (`nccl`) backend on the ONE GPU of the test box: XCK_DIST_FORCE=1 sends a single rank down the multi-rank path of the
    front-ends - fc_common.Dist brings the communicator up with `device_id`, plans the units, and the results meet through the
    collectives of the sharded writer (all-reduces of device tensors) or through BlockGatherer (all-gather of sizes + gather of the
    blocks still resident in HBM).  Covers what a world of one can: librccl next to the pre-loaded HIP runtime, communicator
    init, every collective call of the path, and output files equal to the reference's.  The process is a plain child (no torchrun,
    no exec after a GPU call)."""
    env = dict(os.environ, XCK_DIST_BACKEND="nccl", XCK_DIST_FORCE="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0", XCK_DIST_GATHER="1" if exchange == "gather" else "0")
    env.pop("XCK_DEVICE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), name, str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    tb = [ln for ln in r.stdout.splitlines() if "Error" in ln or "error" in ln or "File \"/" in ln]
    assert "MULTIRANK_OK " + name in r.stdout, "\n".join(tb[-40:]) or r.stdout[-2000:]
    assert "MULTIRANK_BACKEND nccl WORLD 1" in r.stdout
