"""bench.py's contract with the driver, at a size that runs in seconds: the JSON line carries the keys the driver and the judge read
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / config.workload / roofline / cpu_baseline), the
in-run comparison with the oracle holds, and a two-rank run (sharing the one GPU of the test box, gloo) writes byte-identical files
through both exchange forms (every rank writes its rows; --gather: blocks gathered on rank 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = ["--reads", "3000000", "--cells", "2000", "--snps", "100000", "--genes", "8000"]
FILES = ("basefc/matrix.mtx", "basefc/features.tsv", "basefc/barcodes.tsv", "baf/xcltk.AD.mtx", "baf/xcltk.DP.mtx", "baf/xcltk.OTH.mtx",
         "baf/xcltk.region.tsv", "baf/xcltk.samples.tsv")


def _run(args, env, timeout=600):
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _read(work, n):
    return {f: open(os.path.join(work, "out_n%d" % n, f), "rb").read() for f in FILES}


def test_bench_line_and_two_rank_outputs(tmp_path):
    from test_gpu_multirank import _free_port
    work = str(tmp_path / "work")
    env = dict(os.environ, XCK_BENCH_DIR=work, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    d = _run([sys.executable, "bench.py", "--steps", "4", "--warmup", "1", "--cpu-sample", "400000", "--resident-passes", "1", "--sub-reads", "1000000",
              "--well-bams", "12", "--well-sub-reads", "20000"] + SIZE, env)
    assert d["metric"] == "reads/sec into AD/DP+basefc matrices" and d["unit"] == "reads/s" and d["n_gpus"] == 1
    assert d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "strong"                               # the same file at every N: total work fixed
    assert d["dtype"] == "int64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["steps"] / 1e3 - d["end_to_end"]["seconds"]) < 0.01
    assert d["end_to_end"]["records_decoded"] == 3000000 and d["end_to_end"]["records_timed"] < 3000000   # the warm-up slice is not counted
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and set(rf["kernels"]) == {"k_join<basefc>", "k_join<pileup>"}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["gpu_rows_vs_oracle"].startswith("ok")
    assert 0 < cb["value_with_decode"] < cb["value"] and cb["decode_seconds"] > 0       # decode + counting: the headline's boundary on the CPU
    assert "deflate_fast" in d["end_to_end"]["bgzf_writer"] and d["end_to_end"]["inflated_bytes_per_record"] > 150
    z6, cr = d["end_to_end_zlib6"], d["cellranger_shape"]
    assert z6["records"] == 1000000 and cr["records"] == 1000000 and "zlib level 6" in z6["bgzf_writer"]
    assert cr["inflated_bytes_per_record"] > 1.7 * z6["inflated_bytes_per_record"] and 0 < cr["ratio_to_headline"] and 0 < z6["ratio_to_headline"]
    assert z6["passes"] == 3 and len(z6["values_all_passes"]) == 3 and z6["value_min"] <= z6["value"] <= z6["value_max"] and "zlib level 6" in cr["bgzf_writer"]
    w4 = d["configs4_well"]                                        # configs[4] at a reduced size inside the default run
    assert w4["bams"] == 12 and w4["value"] > 0 and w4["gpu_rows_vs_oracle"].startswith("ok") and w4["nnz"]["count"] > 100 and "configs[4]" in w4["workload"]
    one = _read(work, 1)
    assert one["basefc/matrix.mtx"].startswith(b"%%MatrixMarket matrix coordinate integer general\n%%\n8000\t2000\t")
    for extra in (["--selfcheck", "--selfcheck-reads", "1500000"], ["--gather"]):
        d2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                   "--master-port", _free_port(), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"] + SIZE + extra, env)
        if "--selfcheck" in extra:                                    # N ranks together == rank 0 alone, checked inside the run before the timed pass
            assert d2["selfcheck"]["identical"] is True and d2["selfcheck"]["ranks"] == 2 and d2["selfcheck"]["reads"] == 1500000
        assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["end_to_end"]["records_in_bam"] == 3000000
        assert 3000000 <= d2["end_to_end"]["records_decoded"] < 3100000 and d2["end_to_end"]["records_timed"] < 3000000   # (blocks shared by two ranks' contigs are walked twice, counted once)
        assert d2["config"]["nnz"] == d["config"]["nnz"]
        mg = d2["multi_gpu"]                                          # what a SCALE record shows of the exchange
        assert mg["exchange"] == ("gather" if "--gather" in extra else "sharded-write") and mg["world_size"] == 2 and mg["ranks_in_collective"] == 2
        assert mg["backend"].startswith("gloo") and "plan_units" in mg["planner"] and mg["units"] >= 24 and mg["bytes_exchanged"] > 0
        assert [e["rank"] for e in mg["per_rank"]] == [0, 1] and all(e["decode_threads"] >= 1 and e["ingest_seconds"] > 0 and e["records_decoded"] > 0 for e in mg["per_rank"])
        assert sum(e["regions"] for e in mg["per_rank"]) == 8000 and sum(e["units"] for e in mg["per_rank"]) == mg["units"]
        two = _read(work, 2)
        for f in FILES:
            assert two[f] == one[f], "%s differs between N=1 and N=2 (%s)" % (f, extra or "sharded write")
        os.remove(os.path.join(work, "out_n2", "basefc", "matrix.mtx"))


def test_bench_well_workload_line(tmp_path):
    """`bench.py --workload well` (BASELINE configs[4] shape) at 24 BAMs x 40 k records: multi-BAM ingest without CB / UB tags, the matrices'
    rows of the sampled contigs equal the oracle's inside the run (it exits non-zero otherwise)."""
    work = str(tmp_path / "work")
    env = dict(os.environ, XCK_BENCH_DIR=work)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    d = _run([sys.executable, "bench.py", "--workload", "well", "--well-bams", "24", "--well-reads", "40000", "--genes", "8000", "--snps", "50000",
              "--cpu-sample", "200000"], env)
    assert d["metric"] == "reads/sec into AD/DP+basefc matrices" and d["n_gpus"] == 1 and "configs[4]" in d["config"]["workload"]
    assert 0.95 * 24 * 40000 < d["end_to_end"]["records"] <= 24 * 40000 and d["value"] > 0 and d["config"]["nnz"]["count"] > 1000   # (the generator draws 1-8 reads per molecule: a small file can fall a little short)
    assert d["cpu_baseline"]["gpu_rows_vs_oracle"].startswith("ok") and d["cpu_baseline"]["value_with_decode"] > 0
    hdr = open(os.path.join(work, "well", "out", "basefc", "matrix.mtx")).read().split("\n")[2].split("\t")
    assert hdr[0] == "8000" and hdr[1] == "24"
    # the same list on two ranks (sharing the test box's GPU, gloo): every rank streams its contigs of every BAM, the ranks write the
    # files together - byte-identical to the single-rank files; each rank checks the sampled contigs it owns against the oracle
    from test_gpu_multirank import _free_port
    env2 = dict(env, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in ([], ["--gather"]):
        d2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", _free_port(),
                   "bench.py", "--gpus", "2", "--workload", "well", "--well-bams", "24", "--well-reads", "40000", "--genes", "8000", "--snps", "50000", "--cpu-sample", "200000"] + extra, env2)
        assert d2["n_gpus"] == 2 and d2["config"]["nnz"] == d["config"]["nnz"] and d2["gpu_rows_vs_oracle"].startswith("ok")
        assert d2["end_to_end"]["records"] == d["end_to_end"]["records"]       # (a BGZF block shared by two ranks' contigs is walked by both, its records are counted once)
        assert sum(e["records_decoded"] for e in d2["multi_gpu"]["per_rank"]) >= d["end_to_end"]["records"]
        mg = d2["multi_gpu"]
        assert mg["exchange"] == ("gather" if extra else "sharded-write") and mg["ranks_in_collective"] == 2 and len(mg["per_rank"]) == 2
        for f in FILES:
            a, b = os.path.join(work, "well", "out", f), os.path.join(work, "well", "out_n2", f)
            assert open(a, "rb").read() == open(b, "rb").read(), "%s differs between N=1 and N=2 (%s)" % (f, extra or "sharded write")
        os.remove(os.path.join(work, "well", "out_n2", "basefc", "matrix.mtx"))



def test_config4_at_size_384_bams_x_100k_records(tmp_path):
    """BASELINE configs[4] at 384 per-cell BAMs x 100 k records (38.4 M records, paired-end, no CB / UB): the multi-BAM ingest path at
    the file count of the config; every row of the sampled contigs of the four matrices equals the oracle's (checked inside the
    run: bench.py exits non-zero on a difference) - reference loop over the BAM list: xcltk/rdr/fc/core.py:153-170."""
    work = str(tmp_path / "work")
    env = dict(os.environ, XCK_BENCH_DIR=work)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    d = _run([sys.executable, "bench.py", "--workload", "well", "--well-bams", "384", "--well-reads", "100000", "--cpu-sample", "3000000"], env, timeout=900)
    assert 0.98 * 384 * 100000 < d["end_to_end"]["records"] <= 384 * 100000 and "384 per-cell BAMs x 100000 records" in d["config"]["workload"]
    cb = d["cpu_baseline"]
    assert cb["gpu_rows_vs_oracle"].startswith("ok") and int(cb["gpu_rows_vs_oracle"].split("(")[1].split()[0]) > 100000
    hdr = open(os.path.join(work, "well", "out", "basefc", "matrix.mtx")).read().split("\n")[2].split("\t")
    assert hdr[0] == "33472" and hdr[1] == "384" and int(hdr[2]) > 1000000
