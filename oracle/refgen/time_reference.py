#!/usr/bin/env python3
"""time_reference.py - CONTAINER-ONLY (CPU baseline item 1): wall time of the UNMODIFIED reference
(run_reference.py: /opt/conda/bin/python3.9, pysam -> oracle/pybam.py) on BASELINE configs[0] (C1: 10 k reads, 1 k
barcodes, 500 SNPs, 200 genes) and on a C2-shaped slice (1 M reads over 24 hg38-length contigs, 33,472 regions, 5 k
barcodes, 100 k SNPs), ncores=8 on this container's 8 vCPUs.

What the figure is: "reference (Python, pysam stand-in)".  The BAM is parsed into the stand-in's memory before the timed
call and its fetch() bisects an in-memory index, so the time is the reference's own per-region / per-read Python work
plus multiprocessing and merge - without BGZF inflate or record decoding.  That flatters the reference (real pysam
would pay htslib's inflate and decode); it is reported beside, never instead of, bench.py's cpu_baseline.

usage: time_reference.py [--slice-reads N] [--skip-slice]   -> prints a JSON summary (copy into profiles/)"""
import json, os, shutil, subprocess, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT]
import numpy as np

RUNNER = os.path.join(HERE, "run_reference.py")


def run(kind, kwargs, bams):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fp:
        json.dump(dict(kind=kind, out_dir=kwargs["out_dir"], kwargs=kwargs, prewarm=bams), fp)
    t0 = time.time()
    r = subprocess.run(["/opt/conda/bin/python3.9", RUNNER, fp.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    os.unlink(fp.name)
    if r.returncode != 0:
        raise SystemExit("reference failed: " + r.stderr[-2000:])
    res = json.loads(r.stdout.strip().split("\n")[-1])
    res["process_wall_s"] = time.time() - t0
    return res


def case(name, n_reads, bam, barcodes, regions, snps, work, ncores=8):
    out = {}
    r = run("basefc", dict(sam_fn=bam, barcode_fn=barcodes, region_fn=regions, out_dir=work + "/fc_" + name, ncores=ncores), [bam])
    out["basefc"] = dict(r, reads_per_s=n_reads / r["elapsed_s"])
    r = run("baf", dict(sam_fn=bam, barcode_fn=barcodes, region_fn=regions, phased_snp_fn=snps, out_dir=work + "/baf_" + name,
                        ncores=ncores, output_all_reg=True), [bam])
    out["baf"] = dict(r, reads_per_s=n_reads / r["elapsed_s"])
    both = out["basefc"]["elapsed_s"] + out["baf"]["elapsed_s"]
    out["both_reads_per_s"] = n_reads / both
    out["n_reads"] = n_reads
    print("%s: basefc %.1fs (%.0f reads/s)  baf %.1fs (%.0f reads/s)  both %.0f reads/s   [stand-in parse %.1fs, not counted]" %
          (name, out["basefc"]["elapsed_s"], out["basefc"]["reads_per_s"], out["baf"]["elapsed_s"], out["baf"]["reads_per_s"],
           out["both_reads_per_s"], out["basefc"]["standin_load_s"]), file=sys.stderr, flush=True)
    return out


def main():
    slice_reads = int(sys.argv[sys.argv.index("--slice-reads") + 1]) if "--slice-reads" in sys.argv else 1_000_000
    work = tempfile.mkdtemp(prefix="xck_timeref_")
    res = {"host": "build container, %d vCPU" % (os.cpu_count() or 0), "ncores": 8,
           "label": "reference (Python, pysam stand-in; BAM pre-parsed in memory, inflate/decode not counted)"}
    from xcltk_amd.synth.generate import make_10x_dataset
    d = make_10x_dataset(work + "/c1", n_reads=10000, n_barcodes=1000, n_snps=500, n_genes=200, contigs=(("chr1", 2000000),), seed=1)
    res["C1"] = case("C1", d["n_reads"], d["bam"], d["barcodes"], d["regions"], d["snps_tsv"], work)
    if "--skip-slice" not in sys.argv:
        from xcltk_amd.synth import soa
        regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
        rng = np.random.default_rng(7)
        bcs = sorted({"".join("ACGT"[i] for i in rng.integers(0, 4, 16)) + "-1" for _ in range(5000)})
        w = work + "/c2"
        os.makedirs(w)
        open(w + "/contigs.tsv", "w").write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names, soa.HG38_LENGTHS)))
        open(w + "/regions.tsv", "w").write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions))
        open(w + "/barcodes.tsv", "w").write("".join(b + "\n" for b in bcs))
        open(w + "/snps.tsv", "w").write("chrom\tpos\tref\talt\tref_hap\talt_hap\n" + "".join("chr%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
        bam = w + "/slice.bam"
        subprocess.check_call([os.path.join(ROOT, "xcltk_amd", "csrc", "xck_synth_bam"), bam, w + "/contigs.tsv", w + "/regions.tsv",
                               w + "/barcodes.tsv", str(slice_reads), "11", "8", "6"])
        res["C2_slice"] = case("C2 slice", slice_reads, bam, w + "/barcodes.tsv", w + "/regions.tsv", w + "/snps.tsv", work)
    shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
