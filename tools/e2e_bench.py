#!/usr/bin/env python3
"""End-to-end (BAM file -> .mtx files) throughput of the drop-in front-ends on one GPU.
Generates a synthetic 10x BAM with csrc/xck_synth_bam, then times fc_wrapper and afc_wrapper.
usage: e2e_bench.py [N_READS] [THREADS] [--decode-only] [--snps N] [--cells C]"""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xcltk_amd import capi
from xcltk_amd.synth import soa

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
decode_only = "--decode-only" in sys.argv
n_snps = int(sys.argv[sys.argv.index("--snps") + 1]) if "--snps" in sys.argv else 100000
n_cells = int(sys.argv[sys.argv.index("--cells") + 1]) if "--cells" in sys.argv else 5000
work = os.environ.get("XCK_E2E_DIR", "/tmp/xck_e2e")
os.makedirs(work, exist_ok=True)
regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
rng = np.random.default_rng(7)
bcs = sorted({"".join("ACGT"[i] for i in rng.integers(0, 4, 16)) + "-1" for _ in range(n_cells)})
with open(work + "/contigs.tsv", "w") as fp:
    fp.write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names, soa.HG38_LENGTHS)))
with open(work + "/regions.tsv", "w") as fp:
    fp.write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions))
with open(work + "/barcodes.tsv", "w") as fp:
    fp.write("".join(b + "\n" for b in bcs))
with open(work + "/snps.tsv", "w") as fp:
    fp.write("chrom\tpos\tref\talt\tref_hap\talt_hap\n" + "".join("chr%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
bam = work + "/synth_%d_%d.bam" % (n_reads, n_cells)
t0 = time.time()
if not os.path.isfile(bam):
    subprocess.check_call([os.path.join(ROOT, "xcltk_amd", "csrc", "xck_synth_bam"), bam, work + "/contigs.tsv",
                           work + "/regions.tsv", work + "/barcodes.tsv", str(n_reads), "11", str(threads), "6"])
print("BAM: %.1f MB, generated in %.1fs" % (os.path.getsize(bam) / 1e6, time.time() - t0), flush=True)
out = {"n_reads": n_reads, "threads": threads, "n_snps": len(snps), "n_cells": len(bcs), "bam_mb": os.path.getsize(bam) / 1e6}
from xcltk_amd.engine import Engine
for mode, name in ((capi.XCK_MODE_BASEFC, "decode_basefc"), (capi.XCK_MODE_BAF, "decode_baf")):
    eng = Engine(mode, names, regions, len(bcs), snps=snps if mode == 2 else (), barcodes=bcs, cell_tag="CB", umi_tag="UB",
                 decode_only=True, n_threads=threads)
    t0 = time.time(); n = 0
    for b in eng.decode_bam(bam):
        n += b["n_reads"]
    dt = time.time() - t0
    out[name + "_reads_per_s"] = n / dt
    print("%s (host decoder alone, incl. numpy copies): %d reads in %.2fs = %.2f M reads/s" % (name, n, dt, n / dt / 1e6), flush=True)
    eng.close()
if not decode_only:
    import logging
    logging.disable(logging.CRITICAL)
    from xcltk_amd.rdr.fc.main import fc_wrapper
    from xcltk_amd.baf.fc.main import afc_wrapper
    for rep in range(2):
        t0 = time.time()
        assert fc_wrapper(bam, work + "/barcodes.tsv", work + "/regions.tsv", work + "/out_fc", ncores=threads) == 0
        t1 = time.time()
        assert afc_wrapper(bam, work + "/barcodes.tsv", work + "/regions.tsv", work + "/snps.tsv", work + "/out_baf", ncores=threads, output_all_reg=True) == 0
        t2 = time.time()
        out["basefc_e2e_reads_per_s"] = n_reads / (t1 - t0); out["baf_e2e_reads_per_s"] = n_reads / (t2 - t1)
        out["both_e2e_reads_per_s"] = n_reads / (t2 - t0)
        print("rep %d: basefc %.2fs (%.2f M reads/s)  baf %.2fs (%.2f M reads/s)  both: %.2f M reads/s" %
              (rep, t1 - t0, n_reads / (t1 - t0) / 1e6, t2 - t1, n_reads / (t2 - t1) / 1e6, n_reads / (t2 - t0) / 1e6), flush=True)
if not decode_only:
    from xcltk_amd.fused import fused_wrapper
    for rep in range(2):
        t0 = time.time()
        assert fused_wrapper(bam, work + "/barcodes.tsv", work + "/regions.tsv", work + "/snps.tsv", work + "/out_fused", ncores=threads) == 0
        dt = time.time() - t0
        out["fused_e2e_reads_per_s"] = n_reads / dt
        print("rep %d: fused basefc+baf from one decode: %.2fs (%.2f M reads/s into all 4 matrices)" % (rep, dt, n_reads / dt / 1e6), flush=True)
print(json.dumps(out))
