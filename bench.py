#!/usr/bin/env python3
"""bench.py - reads/sec into the basefc count matrix AND the AD/DP/OTH matrices (BASELINE.json).

One "step" = one full pass of the hot path over one synthetic coordinate-sorted record set
that is already resident in HBM: read x region join + UMI de-duplication -> count matrix, and
read x SNP pileup + first-read-per-UMI + haplotype algebra -> AD/DP/OTH matrices, through the
C-ABI (xck_push_batch_device / xck_finish).  Workload at N=1: BASELINE.json configs[1]
(50 M reads, 5 k barcodes, 100 k het SNPs, 33,472 genes on the 24 hg38 contigs).

Multi-GPU (N>1, one rank per GPU, launched by torch.distributed.run): contigs are assigned to
ranks by longest-processing-time; every rank processes --reads reads of ITS contigs (weak
scaling) with no data-path collective; the per-rank sparse blocks are concatenated on rank 0
with one RCCL all-gather of sizes plus one padded all-gather of triplets per matrix.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.shard import BlockGatherer, linear_partition
from xcltk_amd.synth import soa, soa_torch

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=500_000_000, help="reads per GPU (BASELINE.json configs[2]; configs[1] = --reads 50000000 --cells 5000 --snps 100000)")
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--snps", type=int, default=1_000_000)
    ap.add_argument("--genes", type=int, default=33472)
    ap.add_argument("--cpu-sample", type=int, default=40_000_000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0 = this process's cores, at most 16)")
    ap.add_argument("--serial", action="store_true", help="no overlap between the basefc and pileup engines")
    ap.add_argument("--depth", type=int, default=2, help="basefc engines used in rotation: the copy-out of pass i drains while pass i+1 computes (1 = overlap with the pileup pass only)")
    ap.add_argument("--contig-subset", default="", help="N=1 only: draw the reads from these contig indices only (comma separated), i.e. the shard one rank of a larger run would own")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback of the hot path)")
    # One rank per GPU (RCCL).  Test mode only: if fewer GPUs than ranks are visible (a 1-GPU box
    # rehearsing the N>1 code path) the ranks share device 0 and the exchange runs over gloo.
    shared_gpu = torch.cuda.device_count() < world
    dev_idx = 0 if shared_gpu else local
    torch.cuda.set_device(dev_idx)
    device = torch.device("cuda", dev_idx)
    gather_device = "cpu" if shared_gpu else device
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    # ---- tables (identical on every rank) and this rank's shard of contigs --------------------
    regions, snps, names = soa.make_tables(args.genes, args.snps, soa.HG38_LENGTHS, seed=2)
    # contiguous contig ranges per rank: rank-order concatenation of the blocks is already (row, col) order
    shard = linear_partition(soa.HG38_LENGTHS, world)[rank]
    arrays, batches = soa_torch.gen_reads_device(regions, names, args.reads, args.cells, seed=100 + rank,
                                                device=device, contig_subset=shard if world > 1 else ([int(x) for x in args.contig_subset.split(",")] if args.contig_subset else None))
    torch.cuda.synchronize()
    n_reads = arrays["n_reads"]
    data_checksum = [int(arrays["pos"].to(torch.int64).sum().item()), int((arrays["umi"] & 0xFFFFFF).sum().item()),
                     int(arrays["cell"].to(torch.int64).sum().item())]
    filt = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True)
    depth = 1 if args.serial else max(1, args.depth)
    engs_fc = [Engine(capi.XCK_MODE_BASEFC, names, regions, args.cells, device=dev_idx, min_include=0.9, **filt) for _ in range(depth)]
    engs_baf = [Engine(capi.XCK_MODE_BAF, names, regions, args.cells, snps=snps, device=dev_idx,
                       min_count=1, min_maf=0, no_dup_hap=True, **filt) for _ in range(depth)]
    b_fc = [soa_torch.device_batch(capi, arrays, c, s, e, False) for c, s, e in batches]
    b_baf = [soa_torch.device_batch(capi, arrays, c, s, e, True) for c, s, e in batches]

    host = dict(reset=0.0, push=0.0, flush=0.0, fin_async=0.0, fin_baf=0.0, collect=0.0)   # host wall time per call site

    def push_all(eng, bs):
        t_a = time.perf_counter()
        eng.reset()
        t_b = time.perf_counter()
        for b in bs:
            eng.push(b, device_resident=True)             # queued; fused into one launch
        t_c = time.perf_counter()
        eng.flush()                                        # join kernel done (timed on its own stream)
        t_d = time.perf_counter()
        host["reset"] += t_b - t_a; host["push"] += t_c - t_b; host["flush"] += t_d - t_c

    acc = dict(ms_join_fc=0.0, ms_join_baf=0.0, ms_fin_fc=0.0, ms_fin_baf=0.0, ms_d2h_fc=0.0)
    last = {}                                              # latest host views of the four matrices + stats
    pending = []                                           # basefc engines whose copy-out is still in flight
    pending_baf = []                                       # pileup engines, likewise (AD / DP / OTH are 0.36 GB at configs[2])

    def collect_fc(eng):
        t_a = time.perf_counter()
        last.update(eng.finish(copy=False))                # waits for the copy stream; views of the pinned result buffers
        host["collect"] += time.perf_counter() - t_a
        st = eng.stats(); last["sfc"] = st
        acc["ms_join_fc"] += st["ms_join"]; acc["ms_fin_fc"] += st["ms_sort"]; acc["ms_d2h_fc"] += st["ms_d2h"]

    def pass_fc(i):
        eng_fc = engs_fc[i % depth]
        push_all(eng_fc, b_fc)
        if args.serial:
            collect_fc(eng_fc)
        else:
            t_a = time.perf_counter()
            eng_fc.finish_async(); pending.append(eng_fc)
            host["fin_async"] += time.perf_counter() - t_a
        return eng_fc

    def collect_baf(eng):
        t_a = time.perf_counter()
        last.update(eng.finish(copy=False))
        host["collect"] += time.perf_counter() - t_a
        st = eng.stats(); last["sbaf"] = st
        acc["ms_join_baf"] += st["ms_join"]; acc["ms_fin_baf"] += st["ms_sort"]

    def pass_baf(i):
        eng_baf = engs_baf[i % depth]
        push_all(eng_baf, b_baf)
        if args.serial:
            collect_baf(eng_baf)
        else:
            t_a = time.perf_counter()
            eng_baf.finish_async(); pending_baf.append(eng_baf)
            host["fin_baf"] += time.perf_counter() - t_a
        return eng_baf

    def step(i):
        # basefc and pileup are independent engines (own handles, streams and accumulators).  Each fold ends with the
        # copy-out of its matrices over PCIe (1.2 GB + 0.4 GB at configs[2], ~27 ms): it is only ENQUEUED
        # (xck_finish_async) and drains on the copy stream while the CUs run the other pass and - with --depth 2, two
        # engines per mode used in rotation, as a multi-sample run does - the next pass's join and fold.  Every pass's
        # matrices are collected (xck_finish) inside the timed region, at the latest in sync().
        # (driving the two passes from two host threads was measured: the kernels of both passes then share the CUs
        # and the step gets slower, 5.2 vs 4.7 ms)
        eng_fc = pass_fc(i)
        eng_baf = pass_baf(i)
        if world > 1:
            # all-gatherv of the per-contig sparse blocks to the writer rank, GPU to GPU over xGMI (RCCL): sizes
            # first, then ONE gather of the padded [row|col|val] blocks that are still resident in HBM.  It is only
            # enqueued here and overlaps the next pass (collected before the next exchange and at the end of the
            # run).  Every rank has also delivered its own row range to pinned host memory over its own PCIe link.
            blocks = dict(eng_fc.result_device()); blocks.update(eng_baf.result_device())
            last["_gathered_sizes"] = gatherer.start(blocks)
        while len(pending) > depth - 1:
            collect_fc(pending.pop(0))
        while len(pending_baf) > depth - 1:
            collect_baf(pending_baf.pop(0))

    # setup, not a step: every engine sizes its device buffers on first use (hit buffers grow by replay, workspaces
    # and pinned result buffers are allocated) - do that once per engine before the warmup passes
    for e in engs_fc:
        push_all(e, b_fc); e.finish(copy=False)
    for e in engs_baf:
        push_all(e, b_baf); e.finish(copy=False)

    gatherer = BlockGatherer(world, rank, device, backend_is_nccl=not shared_gpu) if world > 1 else None

    def sync():
        while pending:
            collect_fc(pending.pop(0))
        while pending_baf:
            collect_baf(pending_baf.pop(0))
        if gatherer is not None:
            gatherer.wait()                               # the last exchange is inside the timed region
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    for kk in acc:
        acc[kk] = 0.0                                      # HIP-event stage times of the timed passes only
    for kk in host:
        host[kk] = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    sync()                                                 # collects the last pass's matrices: inside the timed region
    dt = time.perf_counter() - t0
    res, sfc, sbaf = last, last["sfc"], last["sbaf"]
    t = torch.tensor([dt], dtype=torch.float64, device=gather_device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    ms_step = dt / args.steps * 1e3
    value = n_reads * world / (dt / args.steps)

    # ---- roofline of the dominant hand-written kernel: k_join (one launch per contig batch) ----
    n_launch = max(1, int(sfc["n_join_launches"]))        # device-resident contig batches are fused into one launch
    L = arrays["read_len"]
    hits_fc, hits_baf = sfc["n_hits"], sbaf["n_hits"]
    # SURVEY 8d per-unit figures: 20 B/read + 4 B/CIGAR op (+ ceil(L/4) B bases for pileup);
    # 16 B per basefc hit written, 24 B per pileup hit written.
    A_fc = n_reads * 20 + arrays["n_cig"] * 4 + hits_fc * 16
    A_baf = n_reads * (20 + (L + 3) // 4) + arrays["n_cig"] * 4 + hits_baf * 24
    k = {name: acc[name] / args.steps for name in acc}
    dom = max(("k_join<basefc>", k["ms_join_fc"], A_fc), ("k_join<pileup>", k["ms_join_baf"], A_baf), key=lambda x: x[1])
    if dom[0].endswith("<pileup>"):
        n_launch = max(1, int(sbaf["n_join_launches"]))
    avg_ms = dom[1] / n_launch
    achieved = (dom[2] / n_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None                                       # PMC bytes were collected at the default workload only (profiles/pmc_traffic.json)
    if os.path.isfile(args.pmc_json) and (args.reads, args.cells, args.snps) == (500_000_000, 10000, 1_000_000) and not args.contig_subset:
        try:
            traffic = json.load(open(args.pmc_json)).get(dom[0])
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel=dom[0], achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    launches_per_step=n_launch, avg_launch_ms=round(avg_ms, 4),
                    algorithmic_bytes_per_launch=int(dom[2] / n_launch),
                    stage_ms_per_step={a: round(b, 3) for a, b in k.items()},
                    host_ms_per_step={a: round(b / args.steps * 1e3, 3) for a, b in host.items()})

    # ---- CPU baseline: the oracle (C restatement of the reference's per-region loops) on the host cores, region chunks per
    # thread like the reference's worker pool (oracle/xck_oracle.c xo_run_mt) ----
    cpu = None
    if args.cpu_sample > 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle as O
        import util
        take, tot = [], 0
        per_contig = {}
        for c, s, e in batches:                        # a long contig may come as several batches
            per_contig.setdefault(c, []).append((c, s, e))
        for c in sorted(per_contig, key=lambda c: sum(e - s for _, s, e in per_contig[c])):   # whole contigs only (their matrix rows are complete), smallest first
            if tot >= args.cpu_sample:
                break
            take += per_contig[c]; tot += sum(e - s for _, s, e in per_contig[c])
        hb = [util.batch_from_dict(soa_torch.host_batch_dict(arrays, c, s, e, True)) for c, s, e in take]
        tc = 0.0
        try:
            n_cores = len(os.sched_getaffinity(0))
        except AttributeError:
            n_cores = os.cpu_count() or 1
        cpu_threads = args.cpu_threads if args.cpu_threads > 0 else max(1, min(n_cores, 16))
        sample_contigs = {names[c] for c, _, _ in take}
        row_in_sample = np.array([r[0] in sample_contigs for r in regions])
        parity = "ok"
        n_cmp = 0
        for mode, sn, mats in ((capi.XCK_MODE_BASEFC, [], ["count"]), (capi.XCK_MODE_BAF, snps, ["ad", "dp", "oth"])):
            cfg, keep = O.make_config(mode, names, regions, sn, args.cells)
            t1 = time.perf_counter()
            exp = O.run_oracle(cfg, [b for b, _ in hb], n_threads=cpu_threads)
            tc += time.perf_counter() - t1
            for m in mats:                              # the GPU result of the last timed step, restricted to those rows
                g = res[m]
                sel = row_in_sample[g[0]]
                ok = all(np.array_equal(g[j][sel], exp[m][j]) for j in range(3))
                n_cmp += int(sel.sum())
                if not ok:
                    parity = "MISMATCH in %s" % m
        cpu = dict(value=round(tot / tc, 1), unit="reads/s", cores=cpu_threads, kind="port",
                   sample="the %d reads of contig(s) %s of the same workload, basefc + pileup, oracle/xck_oracle.c (xo_run_mt: region chunks over %d threads)" % (tot, ",".join(sorted(sample_contigs)), cpu_threads),
                   seconds=round(tc, 2), gpu_rows_vs_oracle="%s (%d non-zeros compared bit for bit)" % (parity, n_cmp))
        if parity != "ok":
            sys.exit("bench.py: GPU result differs from the oracle on the sampled contigs: " + parity)

    cfg_name = {(500_000_000, 10000, 1_000_000): "configs[2]", (50_000_000, 5000, 100_000): "configs[1]"}.get((args.reads, args.cells, args.snps), "configs[2] shape at a custom size")
    line = dict(metric="reads/sec into AD/DP+basefc matrices", value=round(value, 1), unit="reads/s",
                n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_step, 3),
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="int64", data="synthetic",
                config=dict(workload="BASELINE.json %s: %d reads/GPU, %d barcodes, %d het SNPs, %d genes, 24 hg38 contigs; "
                                     "basefc + pileup per step, SoA resident in HBM" % (cfg_name, n_reads, args.cells, len(snps), len(regions)),
                            reads_per_gpu=n_reads, data_checksum=data_checksum, parallelism="contig-shard x%d" % world, pipeline="serial" if args.serial else "copy-out overlapped, %d engine(s) per mode in rotation" % depth,
                            nnz={kk: (int(sum(sz[j] for sz in res["_gathered_sizes"])) if world > 1 else int(len(res[kk][0]))) for j, kk in enumerate(("count", "ad", "dp", "oth"))},
                            hits=dict(basefc=int(hits_fc), pileup=int(hits_baf),
                                      basefc_after_lds_dedup=int(sfc["n_hits_unique"]), pileup_after_lds_dedup=int(sbaf["n_hits_unique"]))),
                roofline=roofline, cpu_baseline=cpu)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
