#!/usr/bin/env python3
"""bench.py - reads/sec into the basefc count matrix AND the AD/DP/OTH matrices, from a BAM file (BASELINE.json).

Headline (`value`): END TO END.  A synthetic coordinate-sorted 10x BAM of BASELINE.json configs[2] (500 M reads,
10 k barcodes, 1 M het SNPs, 33,472 genes on the 24 hg38 contigs; csrc/xck_synth_bam, page cache) goes through the
C-ABI exactly as `xcltk basefc` + `xcltk baf` step 3 would drive it, from ONE decode (XCK_MODE_BOTH):
    xck_bam_open -> xck_ingest_bam (host BGZF/BAM decode on --threads cores, pinned SoA batches, hipMemcpyAsync,
    k_join<basefc> + k_join<pileup>) -> xck_finish (folds on the GPU, copy-out) -> xck_write_mtx x 4 (+ the .tsv files).
The pass is cut into warmup + steps equal slices of the file's records (xck_ingest_opts.pause_records): the warmup
slices are streamed untimed (buffers reach their sizes), the `steps` slices are timed, and the LAST timed step also
carries xck_finish and the writers for the whole file.  value = records of the timed slices / that time; PCIe and
host decode included; every BAM record counts, filtered or not.

N > 1 (one rank per GPU, launched by torch.distributed.run): STRONG scaling of the same file - contigs are assigned to
ranks by longest-processing-time on the .bai record counts, every rank inflates only the byte ranges of its contigs with
its share of the host cores, ranks own disjoint matrix rows, and every rank writes the lines of its own rows into the
four .mtx files at offsets derived from ONE all-reduce of text sizes per file (shard.write_mtx_sharded; the ranks of a
node share the output directory).  --gather restores the exchange of the sparse blocks instead: one all-gather of sizes +
one padded gather of the blocks still resident in HBM (RCCL over xGMI) to rank 0, which merges and writes.

Sub-records of the same JSON line:
  roofline         dominant hand-written kernel on the HBM-resident form of the same workload shape (500 M synthetic
                   reads generated on the device; one fused launch per pass), HIP-event time per launch against SURVEY
                   section 8d's algorithmic bytes;  `device_resident` holds that pass's rate and stage times.
  cpu_baseline     oracle/xck_oracle.c (CPU restatement of the reference's per-region / per-SNP loops) on whole
                   contigs of the SAME BAM, and the GPU rows of those contigs compared with it bit for bit.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.shard import BlockGatherer, contig_owner, merge_row_blocks, write_mtx_sharded
from xcltk_amd.synth import soa, soa_torch

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FILT = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True)


def host_cores():
    """(CPUs' worth of time this process may use, decode threads that spend it): affinity and cgroup quota.  A GPU box shows 256
    hardware threads behind a 16-CPU quota; there 1.5 threads per quota CPU keep the quota spent (csrc/bam.cpp default_threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    threads = n
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            c = max(1, -(-int(q) // int(p)))
            if c < n:
                n, threads = c, min(n, c + c // 2)
    except Exception:
        pass
    return max(1, n), max(1, threads)


def make_inputs(args, work, threads, log):
    """Tables + the synthetic BAM (generated once per work dir and size; reused by later runs / other ranks)."""
    os.makedirs(work, exist_ok=True)
    regions, snps, names = soa.make_tables(args.genes, args.snps, soa.HG38_LENGTHS, seed=2)
    rng = np.random.default_rng(7)
    bcs = sorted({"".join("ACGT"[i] for i in rng.integers(0, 4, 16)) + "-1" for _ in range(args.cells)})
    while len(bcs) < args.cells:                            # (a duplicate draw: top up)
        bcs = sorted(set(bcs) | {"".join("ACGT"[i] for i in rng.integers(0, 4, 16)) + "-1"})
    bam = os.path.join(work, "synth_%d_%d_l%d.bam" % (args.reads, args.cells, args.level))
    ok = bam + ".ok"
    fresh = False
    if not (os.path.isfile(ok) and os.path.isfile(bam) and os.path.isfile(bam + ".bai")):
        t0 = time.time()
        with open(work + "/contigs.tsv", "w") as fp:
            fp.write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names, soa.HG38_LENGTHS)))
        with open(work + "/regions.tsv", "w") as fp:
            fp.write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions))
        with open(work + "/barcodes.tsv", "w") as fp:
            fp.write("".join(b + "\n" for b in bcs))
        subprocess.check_call([os.path.join(ROOT, "xcltk_amd", "csrc", "xck_synth_bam"), bam, work + "/contigs.tsv",
                               work + "/regions.tsv", work + "/barcodes.tsv", str(args.reads), "11", str(threads), str(args.level)],
                              stderr=subprocess.DEVNULL if not args.verbose else None)
        open(ok, "w").write("ok\n")
        fresh = True
        log("BAM generated in %.1f s: %.1f GB" % (time.time() - t0, os.path.getsize(bam) / 1e9))
    return regions, snps, names, bcs, bam, fresh


def warm_page_cache(path):
    """The metric reads the BAM from the page cache: touch every page once, untimed (a no-op cost right after generation)."""
    buf = bytearray(64 << 20)
    with open(path, "rb", buffering=0) as fp:
        while fp.readinto(buf):
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=500_000_000, help="records in the BAM (BASELINE.json configs[2]; configs[1] = --reads 50000000 --cells 5000 --snps 100000)")
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--genes", type=int, default=33472)
    ap.add_argument("--threads", type=int, default=0, help="host decode threads per rank (0 = this process's CPU share / ranks)")
    ap.add_argument("--level", type=int, default=0, help="BGZF compression of the synthetic BAM: 0 = csrc/deflate_fast.h, 1-9 = zlib")
    ap.add_argument("--work", default=os.environ.get("XCK_BENCH_DIR", "/tmp/xck_bench"))
    ap.add_argument("--cpu-sample", type=int, default=40_000_000, help="records in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--resident-passes", type=int, default=5, help="timed passes of the HBM-resident sub-record (0 = skip it and the roofline)")
    ap.add_argument("--gather", action="store_true", help="N > 1: gather the per-rank sparse blocks on rank 0 (RCCL) and let it write the files, "
                    "instead of every rank writing the lines of its own rows (default)")
    ap.add_argument("--resident-only", action="store_true", help="run only the HBM-resident sub-record (the command profiles/ traces with rocprofv3: "
                    "its k_join launches are the 500 M-read launches the roofline is computed from)")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback of the hot path)")

    def log(msg):
        if args.verbose and rank == 0:
            sys.stderr.write("[bench] %s\n" % msg)
            sys.stderr.flush()

    # One rank per GPU (RCCL).  Test mode only: if fewer GPUs than ranks are visible (a 1-GPU box rehearsing the N>1
    # code path) the ranks share device 0 and the exchange runs over gloo.
    shared_gpu = torch.cuda.device_count() < world
    dev_idx = 0 if shared_gpu else local
    torch.cuda.set_device(dev_idx)
    device = torch.device("cuda", dev_idx)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=30))
    gather_device = "cpu" if shared_gpu else device

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cores, pool_threads = host_cores()
    threads = args.threads if args.threads > 0 else max(1, pool_threads // world)

    if args.resident_only:                                     # profiling aid, N = 1: no BAM, no files
        if world != 1:
            sys.exit("--resident-only is a single-GPU run")
        regions, snps, names = soa.make_tables(args.genes, args.snps, soa.HG38_LENGTHS, seed=2)
        resident, roofline = device_resident(args, names, regions, snps, dev_idx, device, log)
        print(json.dumps(dict(metric="reads/sec, HBM-resident sub-record only (NOT the headline metric)", value=resident["reads_per_s"], unit="reads/s",
                              n_gpus=1, sub_record_only=True, dtype="int64", data="synthetic", device_resident=resident, roofline=roofline)))
        return

    # ---- inputs: generated by local rank 0 with all of the box's cores, the others wait ----
    if rank == 0:
        regions, snps, names, bcs, bam, fresh = make_inputs(args, args.work, cores, log)
        if not fresh:
            warm_page_cache(bam)
    barrier()
    if rank != 0:
        regions, snps, names, bcs, bam, fresh = make_inputs(args, args.work, cores, log)
    bam_bytes = os.path.getsize(bam)

    # ---- engine (both pipelines behind one handle) and this rank's contigs ----
    t_setup = time.perf_counter()
    eng = Engine(capi.XCK_MODE_BOTH, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB",
                 device=dev_idx, min_include=0.9, min_count=1, min_maf=0, no_dup_hap=True, n_threads=threads, **FILT)
    counts = eng.contig_record_counts(bam)
    if counts is None:
        sys.exit("bench.py: the synthetic BAM has no usable .bai")
    n_total = int(counts.sum())
    mask = None
    if world > 1:
        mask = contig_owner(names, counts.astype(np.float64) + 1e-9, world) == rank
    n_mine = int(counts[mask].sum()) if mask is not None else n_total
    setup_s = time.perf_counter() - t_setup
    cidx = {n: i for i, n in enumerate(names)}
    row_contig = np.array([cidx[r[0]] for r in regions], dtype=np.int32)
    out_dir = os.path.join(args.work, "out_n%d" % world)
    if rank == 0:
        os.makedirs(os.path.join(out_dir, "basefc"), exist_ok=True)
        os.makedirs(os.path.join(out_dir, "baf"), exist_ok=True)
    gatherer = BlockGatherer(world, rank, device, backend_is_nccl=not shared_gpu) if world > 1 and args.gather else None

    # ---- the pass: warmup + steps slices of this rank's records ----
    n_slices = max(1, args.warmup + args.steps)
    slice_records = max(1, -(-n_mine // n_slices))
    stream = eng.open_stream(bam, sample=0, n_threads=threads, contig_mask=mask, use_index=mask is not None)
    done_records = 0
    for i in range(args.warmup):
        done_records, _ = stream.advance(slice_records)
    eng.flush()
    barrier()
    warm_records = done_records
    marks = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = i + 1 == args.steps
        done_records, _ = stream.advance(0 if last else slice_records)
    marks["ingest"] = time.perf_counter() - t0
    coo = eng.finish(copy=False)                                # folds on the GPU + copy-out to pinned host memory
    stats = eng.stats()
    marks["finish"] = time.perf_counter() - t0
    n = len(regions)
    rm = np.arange(1, n + 1, dtype=np.int32)                   # output_all_reg: row = input line (rdr/fc/config.py:103, baf/pipeline.py:356)
    files = (("count", os.path.join(out_dir, "basefc", "matrix.mtx")), ("ad", os.path.join(out_dir, "baf", "xcltk.AD.mtx")),
             ("dp", os.path.join(out_dir, "baf", "xcltk.DP.mtx")), ("oth", os.path.join(out_dir, "baf", "xcltk.OTH.mtx")))
    nnz = None
    if world > 1 and not args.gather:
        # every rank writes the lines of its own rows into the four files (one all-reduce of text sizes per file): no triplets
        # travel, and the ~10^8 lines are formatted by all ranks' host cores instead of rank 0's
        row_owner = contig_owner(names, counts.astype(np.float64) + 1e-9, world)[row_contig]

        def all_reduce_sum(x):
            t_ = torch.from_numpy(np.ascontiguousarray(x)).to(gather_device)
            dist.all_reduce(t_)
            return t_.cpu().numpy()
        nnz = {k: write_mtx_sharded(fn, coo[k], rm, row_owner, n, len(bcs), rank, all_reduce_sum, dist.barrier, eng.lib) for k, fn in files}
        marks["write"] = time.perf_counter() - t0
    elif world > 1:
        gatherer.start(eng.result_device())
        blocks = gatherer.wait()                               # rank 0: {name: [per-rank int32 [row|col|val] tensors]}
        marks["gather"] = time.perf_counter() - t0
        if rank == 0:
            owner = contig_owner(names, counts.astype(np.float64) + 1e-9, world)
            coo = {k: merge_row_blocks([b.cpu().numpy() for b in blocks[k]], owner[row_contig]) for k in blocks}
            marks["merge"] = time.perf_counter() - t0
    if rank == 0:
        with open(os.path.join(out_dir, "basefc", "features.tsv"), "w") as fp:
            fp.write("".join("%s\t%d\t%d\t%s\n" % r for r in regions))
        with open(os.path.join(out_dir, "basefc", "barcodes.tsv"), "w") as fp:
            fp.write("".join(b + "\n" for b in bcs))
        with open(os.path.join(out_dir, "baf", "xcltk.region.tsv"), "w") as fp:
            fp.write("".join("%s\t%d\t%d\t%s\n" % r for r in regions))
        with open(os.path.join(out_dir, "baf", "xcltk.samples.tsv"), "w") as fp:
            fp.write("".join(b + "\n" for b in bcs))
        if nnz is None:
            for k, fn in files:
                eng.write_mtx_arrays(fn, coo[k], rm, n)
            nnz = {k: int(len(coo[k][0])) for k, _ in files}
        marks["write"] = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    stream.close()
    timed_records = done_records - warm_records
    t = torch.tensor([dt, float(timed_records), float(done_records)], dtype=torch.float64, device=gather_device)
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0].item())
    timed_all, decoded_all = int(t[1].item()), int(t[2].item())
    if rank != 0:
        eng.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    if decoded_all > n_total:                                  # N > 1: ranks walk whole BGZF blocks, so the records of a block that spans
        timed_all = int(timed_all * (n_total / decoded_all))   # two ranks' contigs are walked twice - count every BAM record once
    value = timed_all / dt
    ms_step = dt / args.steps * 1e3
    mtx_bytes = sum(os.path.getsize(os.path.join(out_dir, d, f)) for d, f in (("basefc", "matrix.mtx"), ("baf", "xcltk.AD.mtx"), ("baf", "xcltk.DP.mtx"), ("baf", "xcltk.OTH.mtx")))
    e2e = dict(records_in_bam=n_total, records_decoded=decoded_all, records_timed=timed_all, seconds=round(dt, 3),
               phase_seconds={k: round(v, 3) for k, v in marks.items()}, host_threads_per_rank=threads, host_cores=cores,
               bam_gb=round(bam_bytes / 1e9, 2), bam_bytes_per_record=round(bam_bytes / max(n_total, 1), 1), bgzf_level=args.level,
               mtx_output_mb=round(mtx_bytes / 1e6, 1), engine_setup_s=round(setup_s, 2),
               engine_ms=dict(h2d=round(stats["ms_h2d"], 1), join=round(stats["ms_join"], 1), fold=round(stats["ms_sort"], 1), d2h=round(stats["ms_d2h"], 1)),
               hits=dict(accepted=int(stats["n_hits"]), after_lds_dedup=int(stats["n_hits_unique"])))
    log("e2e: %.2f M reads/s (%s)" % (value / 1e6, e2e["phase_seconds"]))

    # ---- CPU baseline + parity of this run's matrices: whole contigs of the same BAM through the oracle ----
    cpu = None
    if args.cpu_sample > 0 and world == 1:
        cpu = cpu_baseline(args, eng, bam, names, regions, snps, bcs, counts, coo, row_contig, threads, cores, log)
    eng.close()

    # ---- HBM-resident sub-record + roofline of the dominant kernel ----
    resident, roofline = None, None
    if args.resident_passes > 0 and world == 1:
        resident, roofline = device_resident(args, names, regions, snps, dev_idx, device, log)

    cfg_name = {(500_000_000, 10000, 1_000_000): "configs[2]", (50_000_000, 5000, 100_000): "configs[1]"}.get((args.reads, args.cells, args.snps), "configs[2] shape at a custom size")
    line = dict(metric="reads/sec into AD/DP+basefc matrices", value=round(value, 1), unit="reads/s",
                n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_step, 3),
                higher_is_better=True, scaling="strong", vs_baseline=None, dtype="int64", data="synthetic",
                config=dict(workload="BASELINE.json %s: ONE synthetic 10x BAM of %d records (%.1f GB BGZF, page cache), %d barcodes, %d het SNPs, %d genes, "
                                     "24 hg38 contigs; end to end BAM -> basefc matrix.mtx + AD/DP/OTH.mtx from one decode; a step = 1/%d of the file's records, "
                                     "the last step also folds and writes" % (cfg_name, n_total, bam_bytes / 1e9, len(bcs), len(snps), len(regions), n_slices),
                            parallelism="contig-shard x%d (LPT on .bai counts)" % world, host_threads_per_rank=threads, nnz=nnz),
                end_to_end=e2e, device_resident=resident, roofline=roofline, cpu_baseline=cpu)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, eng, bam, names, regions, snps, bcs, counts, coo, row_contig, threads, cores, log):
    """The oracle on whole contigs of the same BAM (smallest first, up to --cpu-sample records), timed; the GPU rows of
    those contigs must equal it bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    import util
    take, tot = [], 0
    for c in sorted(range(len(names)), key=lambda c: (counts[c], c)):
        if counts[c] == 0:
            continue
        if tot >= args.cpu_sample or (take and tot + counts[c] > 2 * args.cpu_sample):
            break
        take.append(c)
        tot += int(counts[c])
    mask = np.zeros(len(names), dtype=bool)
    mask[take] = True
    dec = Engine(capi.XCK_MODE_BOTH, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB",
                 decode_only=True, n_threads=threads)
    hb = [util.batch_from_dict(d) for d in dec.decode_bam(bam, contig_mask=mask, use_index=True)]
    dec.close()
    n_dec = sum(b.n_reads for b, _ in hb)
    cpu_threads = args.cpu_threads if args.cpu_threads > 0 else max(1, min(cores, 16))
    sample_rows = mask[row_contig]
    parity, n_cmp, tc = "ok", 0, 0.0
    for mode, sn, mats in ((capi.XCK_MODE_BASEFC, [], ["count"]), (capi.XCK_MODE_BAF, snps, ["ad", "dp", "oth"])):
        cfg, keep = O.make_config(mode, names, regions, sn, len(bcs), min_include=0.9, min_count=1, min_maf=0, no_dup_hap=True, **FILT)
        t1 = time.perf_counter()
        exp = O.run_oracle(cfg, [b for b, _ in hb], n_threads=cpu_threads)
        tc += time.perf_counter() - t1
        for m in mats:
            g = coo[m]
            sel = sample_rows[g[0]]
            n_cmp += int(sel.sum())
            if not all(np.array_equal(g[j][sel], exp[m][j]) for j in range(3)):
                parity = "MISMATCH in %s" % m
    log("cpu baseline: %d records in %.1f s; parity %s" % (n_dec, tc, parity))
    if parity != "ok":
        sys.exit("bench.py: GPU result differs from the oracle on the sampled contigs: " + parity)
    return dict(value=round(n_dec / tc, 1), unit="reads/s", cores=cpu_threads, kind="port",
                sample="the %d records of contig(s) %s of the same BAM (decoded to SoA batches before the timed call), basefc + pileup, "
                       "oracle/xck_oracle.c (xo_run_mt: region chunks over %d threads)" % (n_dec, ",".join(names[c] for c in take), cpu_threads),
                seconds=round(tc, 2), gpu_rows_vs_oracle="%s (%d non-zeros of the end-to-end matrices compared bit for bit)" % (parity, n_cmp))


def device_resident(args, names, regions, snps, dev_idx, device, log):
    """The same workload shape as SoA batches already in HBM (generated on the device): one basefc pass + one pileup pass
    per step through xck_push_batch_device / xck_finish, serial (no overlap between engines) so that every stage time is
    the stage's own.  Feeds the roofline of the dominant hand-written kernel."""
    arrays, batches = soa_torch.gen_reads_device(regions, names, args.reads, args.cells, seed=100, device=device)
    torch.cuda.synchronize()
    n_reads = arrays["n_reads"]
    eng_fc = Engine(capi.XCK_MODE_BASEFC, names, regions, args.cells, device=dev_idx, min_include=0.9, **FILT)
    eng_baf = Engine(capi.XCK_MODE_BAF, names, regions, args.cells, snps=snps, device=dev_idx, min_count=1, min_maf=0, no_dup_hap=True, **FILT)
    b_fc = [soa_torch.device_batch(capi, arrays, c, s, e, False) for c, s, e in batches]
    b_baf = [soa_torch.device_batch(capi, arrays, c, s, e, True) for c, s, e in batches]
    acc = dict(ms_join_fc=0.0, ms_join_baf=0.0, ms_fin_fc=0.0, ms_fin_baf=0.0, ms_d2h=0.0)
    res = {}

    def one(eng, bs, join_key, fin_key, collect):
        eng.reset()
        for b in bs:
            eng.push(b, device_resident=True)              # queued; fused into one launch
        eng.flush()
        out = eng.finish(copy=False)
        st = eng.stats()
        if collect:
            acc[join_key] += st["ms_join"]; acc[fin_key] += st["ms_sort"]; acc["ms_d2h"] += st["ms_d2h"]
            res.update(out); res[join_key] = st
    for _ in range(2):                                      # buffers reach their sizes (hit buffers grow by replay)
        one(eng_fc, b_fc, "ms_join_fc", "ms_fin_fc", False)
        one(eng_baf, b_baf, "ms_join_baf", "ms_fin_baf", False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.resident_passes):
        one(eng_fc, b_fc, "ms_join_fc", "ms_fin_fc", True)
        one(eng_baf, b_baf, "ms_join_baf", "ms_fin_baf", True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.resident_passes
    sfc, sbaf = res["ms_join_fc"], res["ms_join_baf"]
    k = {a: b / args.resident_passes for a, b in acc.items()}
    L = arrays["read_len"]
    hits_fc, hits_baf = sfc["n_hits"], sbaf["n_hits"]
    # SURVEY 8d per-unit figures: 20 B/read + 4 B/CIGAR op (+ ceil(L/4) B bases for pileup); 16 B per basefc hit written, 24 B per pileup hit written
    A_fc = n_reads * 20 + arrays["n_cig"] * 4 + hits_fc * 16
    A_baf = n_reads * (20 + (L + 3) // 4) + arrays["n_cig"] * 4 + hits_baf * 24
    dom = max(("k_join<basefc>", k["ms_join_fc"], A_fc, sfc), ("k_join<pileup>", k["ms_join_baf"], A_baf, sbaf), key=lambda x: x[1])
    n_launch = max(1, int(dom[3]["n_join_launches"]))
    avg_ms = dom[1] / n_launch
    achieved = (dom[2] / n_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None                                          # PMC bytes were collected at the default workload only (profiles/pmc_traffic.json)
    if os.path.isfile(args.pmc_json) and (args.reads, args.cells, args.snps) == (500_000_000, 10000, 1_000_000):
        try:
            traffic = json.load(open(args.pmc_json)).get(dom[0])
        except Exception:
            traffic = None
    both = {name: dict(avg_launch_ms=round(ms / max(1, int(stt["n_join_launches"])), 4), algorithmic_bytes_per_launch=int(A / max(1, int(stt["n_join_launches"]))),
                       frac=round((A / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0, 4))
            for name, ms, A, stt in (("k_join<basefc>", k["ms_join_fc"], A_fc, sfc), ("k_join<pileup>", k["ms_join_baf"], A_baf, sbaf))}
    roofline = dict(bound="hbm", kernel=dom[0], achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, launches_per_pass=n_launch, avg_launch_ms=round(avg_ms, 4),
                    algorithmic_bytes_per_launch=int(dom[2] / n_launch), workload="HBM-resident sub-record (device_resident)",
                    note="dominant = the join kernel with the longer launch; both join kernels are listed under 'kernels'", kernels=both)
    resident = dict(reads=n_reads, passes=args.resident_passes, ms_per_pass=round(dt * 1e3, 3), reads_per_s=round(n_reads / dt, 1),
                    pipeline="serial: basefc pass then pileup pass, matrices copied out to pinned host memory inside the pass",
                    stage_ms_per_pass={a: round(b, 3) for a, b in k.items()},
                    nnz={m: int(len(res[m][0])) for m in ("count", "ad", "dp", "oth")},
                    hits=dict(basefc=int(hits_fc), pileup=int(hits_baf), basefc_after_lds_dedup=int(sfc["n_hits_unique"]), pileup_after_lds_dedup=int(sbaf["n_hits_unique"])))
    log("resident: %.2f ms/pass %s" % (dt * 1e3, resident["stage_ms_per_pass"]))
    eng_fc.close(); eng_baf.close()
    return resident, roofline


if __name__ == "__main__":
    main()
