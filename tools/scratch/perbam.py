import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from xcltk_amd.synth import soa
from xcltk_amd.engine import Engine
work = "/tmp/xck_pb"; os.makedirs(work, exist_ok=True)
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
open(work + "/contigs.tsv", "w").write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names, soa.HG38_LENGTHS)))
open(work + "/regions.tsv", "w").write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions))
open(work + "/barcodes.tsv", "w").write("AAAA-1\n")
for tag, env in (("t", {}), ("n", {"XCK_SYNTH_NOTAGS": "1"})):
    for i in range(8):
        subprocess.check_call(["xcltk_amd/csrc/xck_synth_bam", work + "/%s_%d.bam" % (tag, i), work + "/contigs.tsv", work + "/regions.tsv", work + "/barcodes.tsv", "500000", str(100 + i), "16", "1"], env=dict(os.environ, **env), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
for label, tag, kw in (("10x tags", "t", dict(barcodes=["AAAA-1"], cell_tag="CB", umi_tag="UB")), ("well names", "n", dict()), ("well names 4thr", "n", dict())):
    thr = 4 if "4thr" in label else 16
    eng = Engine(1, names, regions, 1 if tag == "t" else 8, decode_only=True, n_threads=thr, **kw)
    for rep in range(2):
        per = []
        for i in range(8):
            t1 = time.time(); n = 0
            for b in eng.decode_bam(work + "/%s_%d.bam" % (tag, i), sample=0 if tag == "t" else i):
                n += b["n_reads"]
            per.append(time.time() - t1)
        print("%-16s rep %d per BAM ms: %s" % (label, rep, " ".join("%.0f" % (x * 1e3) for x in per)), flush=True)
    eng.close()
