#!/usr/bin/env python3
"""Host-ingest thread scaling: the BAM decoder alone (decode-only handle: BGZF inflate + record walk + field / tag parse
into pinned SoA, batches discarded) and, with a GPU, the fused end-to-end ingest (decode + H2D + both join kernels).
XCK_DEBUG_TIMING=1 adds the decoder's own time accounting per run.
usage: ingest_scaling.py BAM BARCODES.tsv [--threads 1,2,4,8,16] [--snps N] [--gpu]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa

bam, bc_fn = sys.argv[1], sys.argv[2]
if "--gen" in sys.argv:                      # generate BAM + barcode list first: --gen N_READS [--level L]
    import subprocess
    import numpy as np
    n_gen = int(sys.argv[sys.argv.index("--gen") + 1])
    level = sys.argv[sys.argv.index("--level") + 1] if "--level" in sys.argv else "0"
    work = os.path.dirname(os.path.abspath(bam))
    os.makedirs(work, exist_ok=True)
    regions0, _, names0 = soa.make_tables(33472, 0, soa.HG38_LENGTHS, seed=2)
    rng = np.random.default_rng(7)
    bcs0 = sorted({"".join("ACGT"[i] for i in rng.integers(0, 4, 16)) + "-1" for _ in range(10000)})
    open(work + "/contigs.tsv", "w").write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names0, soa.HG38_LENGTHS)))
    open(work + "/regions.tsv", "w").write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions0))
    open(bc_fn, "w").write("".join(b + "\n" for b in bcs0))
    if not os.path.isfile(bam):
        subprocess.check_call([os.path.join(ROOT, "xcltk_amd", "csrc", "xck_synth_bam"), bam, work + "/contigs.tsv", work + "/regions.tsv", bc_fn, str(n_gen), "11", "16", level])
threads = [int(x) for x in (sys.argv[sys.argv.index("--threads") + 1] if "--threads" in sys.argv else "1,2,4,8,16").split(",")]
n_snps = int(sys.argv[sys.argv.index("--snps") + 1]) if "--snps" in sys.argv else 1000000
regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
bcs = [l.strip() for l in open(bc_fn)]
out = {}
for mode_name, decode_only in (("decode_only", True),) + ((("fused_e2e_ingest", False),) if "--gpu" in sys.argv else ()):
    for t in threads:
        eng = Engine(capi.XCK_MODE_BOTH, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB",
                     decode_only=decode_only, n_threads=t)
        best = 0.0
        for rep in range(2):
            if not decode_only:
                eng.reset()
            t0 = time.time()
            st = eng.open_stream(bam)
            n, _ = st.advance(0)
            if not decode_only:
                eng.flush()
            st.close()
            dt = time.time() - t0
            best = max(best, n / dt)
        eng.close()
        out["%s_t%d" % (mode_name, t)] = round(best / 1e6, 2)
        print("%-18s threads=%-3d %8.2f M reads/s (%.2f per thread)" % (mode_name, t, best / 1e6, best / 1e6 / t), flush=True)
print(json.dumps(out))
