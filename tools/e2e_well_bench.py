#!/usr/bin/env python3
"""End-to-end throughput of the multi-BAM well-based mode (BASELINE configs[4] shape: one BAM per cell, no CB / UB tags,
`--cellTAG None --UMItag None`, counting read names) on one GPU.
usage: e2e_well_bench.py [N_BAMS] [READS_PER_BAM] [THREADS]"""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xcltk_amd.synth import soa

n_bams = int(sys.argv[1]) if len(sys.argv) > 1 else 48
per = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
work = os.environ.get("XCK_E2E_DIR", "/tmp/xck_e2e_well")
os.makedirs(work, exist_ok=True)
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
with open(work + "/contigs.tsv", "w") as fp:
    fp.write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names, soa.HG38_LENGTHS)))
with open(work + "/regions.tsv", "w") as fp:
    fp.write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions))
with open(work + "/barcodes.tsv", "w") as fp:
    fp.write("AAAA-1\n")
with open(work + "/snps.tsv", "w") as fp:
    fp.write("chrom\tpos\tref\talt\tref_hap\talt_hap\n" + "".join("chr%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
t0 = time.time()
bams = []
env = dict(os.environ, XCK_SYNTH_NOTAGS="1")
for i in range(n_bams):
    fn = work + "/cell_%03d_%d.bam" % (i, per)
    if not os.path.isfile(fn):
        subprocess.check_call([os.path.join(ROOT, "xcltk_amd", "csrc", "xck_synth_bam"), fn, work + "/contigs.tsv", work + "/regions.tsv",
                               work + "/barcodes.tsv", str(per), str(100 + i), str(threads), os.environ.get("XCK_E2E_LEVEL", "6")], env=env)
    bams.append(fn)
with open(work + "/bams.txt", "w") as fp:
    fp.write("".join(b + "\n" for b in bams))
with open(work + "/samples.txt", "w") as fp:
    fp.write("".join("cell%03d\n" % i for i in range(n_bams)))
print("%d BAMs x %d reads, %.1f MB, generated in %.1fs" % (n_bams, per, sum(os.path.getsize(b) for b in bams) / 1e6, time.time() - t0), flush=True)
import logging
logging.disable(logging.CRITICAL)
from xcltk_amd.rdr.fc.main import fc_wrapper
from xcltk_amd.baf.fc.main import afc_wrapper
n = n_bams * per
out = dict(n_bams=n_bams, reads_per_bam=per, threads=threads)
for rep in range(2):
    t0 = time.time()
    assert fc_wrapper(None, None, work + "/regions.tsv", work + "/out_fc", sam_list_fn=work + "/bams.txt", sample_id_fn=work + "/samples.txt",
                      cell_tag=None, umi_tag=None, ncores=threads) == 0
    t1 = time.time()
    assert afc_wrapper(None, None, work + "/regions.tsv", work + "/snps.tsv", work + "/out_baf", sam_list_fn=work + "/bams.txt",
                       sample_id_fn=work + "/samples.txt", cell_tag=None, umi_tag=None, ncores=threads, output_all_reg=True) == 0
    t2 = time.time()
    out["basefc_e2e_reads_per_s"] = n / (t1 - t0); out["baf_e2e_reads_per_s"] = n / (t2 - t1)
    print("rep %d: basefc %.2fs (%.2f M reads/s)  baf %.2fs (%.2f M reads/s)" % (rep, t1 - t0, n / (t1 - t0) / 1e6, t2 - t1, n / (t2 - t1) / 1e6), flush=True)
print(json.dumps(out))
