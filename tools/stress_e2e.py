#!/usr/bin/env python3
"""Repeat bench.py's end-to-end pass (paused ingest in slices -> finish) on an existing synthetic BAM, to shake out rare
host-side failures.  usage: stress_e2e.py BAM BARCODES.tsv [reps] [threads] [slices]   (tables as in bench.py / ingest_scaling.py)"""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa

bam, bc_fn = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
slices = int(sys.argv[5]) if len(sys.argv) > 5 else 25
regions, snps, names = soa.make_tables(33472, 1000000, soa.HG38_LENGTHS, seed=2)
bcs = [l.strip() for l in open(bc_fn)]
for rep in range(reps):
    t0 = time.time()
    eng = Engine(capi.XCK_MODE_BOTH, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB", n_threads=threads,
                 min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9, min_count=1, min_maf=0, no_dup_hap=True)
    counts = eng.contig_record_counts(bam)
    n_total = int(counts.sum())
    st = eng.open_stream(bam, sample=0, n_threads=threads)
    for i in range(slices):
        n, done = st.advance(0 if i + 1 == slices else -(-n_total // slices))
    coo = eng.finish(copy=False)
    nnz = {k: len(v[0]) for k, v in coo.items()}
    st.close()
    eng.close()
    print("rep %d: %d records, %.1f s, nnz %s" % (rep, n, time.time() - t0, nnz), flush=True)
