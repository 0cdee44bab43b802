#!/usr/bin/env python3
"""scratch: cProfile of fused_wrapper on the e2e dataset left by tools/e2e_bench.py in /tmp/xck_e2e (run that first)."""
import cProfile, pstats, sys, os, glob, logging
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
logging.disable(logging.CRITICAL)
from xcltk_amd.fused import fused_wrapper
w = "/tmp/xck_e2e"
bam = sorted(glob.glob(w + "/synth_*.bam"))[-1]
fused_wrapper(bam, w + "/barcodes.tsv", w + "/regions.tsv", w + "/snps.tsv", w + "/out_prof0", ncores=16)      # warm
pr = cProfile.Profile()
pr.enable()
fused_wrapper(bam, w + "/barcodes.tsv", w + "/regions.tsv", w + "/snps.tsv", w + "/out_prof", ncores=16)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
