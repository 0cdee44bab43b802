#!/usr/bin/env python3
"""Where does the time of a multi-BAM (well-based) ingest go?  Per-BAM wall time of Engine.ingest_bam on the BAMs that
tools/e2e_well_bench.py left in XCK_E2E_DIR, against a decode-only handle on the same files."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcltk_amd.synth import soa
from xcltk_amd.engine import Engine
work = os.environ.get("XCK_E2E_DIR", "/tmp/xck_e2e_well")
bams = sorted(glob.glob(work + "/cell_*.bam"))[:32]
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for label, kw in (("decode-only", dict(decode_only=True)), ("basefc", dict()), ("basefc again", dict())):
    eng = Engine(1, names, regions, len(bams), n_threads=threads, **kw)
    t0 = time.time(); n = 0; per = []
    for i, fn in enumerate(bams):
        t1 = time.time()
        if kw.get("decode_only"):
            for b in eng.decode_bam(fn, sample=i):
                n += b["n_reads"]
        else:
            n += eng.ingest_bam(fn, sample=i)
        per.append(time.time() - t1)
    if not kw.get("decode_only"):
        t1 = time.time(); eng.finish(); tf = time.time() - t1
    else:
        tf = 0.0
    dt = time.time() - t0
    print("%-14s %d BAMs %d reads %.2fs = %.2f M reads/s; per BAM min %.1f med %.1f max %.1f ms; finish %.1f ms; stats %s" % (
        label, len(bams), n, dt, n / dt / 1e6, min(per) * 1e3, sorted(per)[len(per) // 2] * 1e3, max(per) * 1e3, tf * 1e3,
        {k: round(v, 1) for k, v in eng.stats().items() if k.startswith("ms_")} if not kw.get("decode_only") else ""), flush=True)
    eng.close()
