#!/usr/bin/env python3
"""make_genotype_goldens.py - CONTAINER-ONLY: pins what CAN be pinned of row f1 (step 1 of `xcltk baf`) with the reference's own code.

The counts of step 1 come from the external cellsnp-lite binary in the reference (xcltk/baf/genotype.py:144-162) and stay PARITY
UNPINNED.  What the reference itself does with such a directory is pure Python and runs here through run_reference.py (pysam ->
oracle/pybam.py, anndata -> anndata_standin.py):

  1. tests/golden/genotype/raw/           a raw pileup directory of the `phasing` dataset: the ORACLE's per-SNP x cell counts (one
                                           feature per candidate SNP, REF on haplotype 0 / ALT on 1) written by this repo's writer
                                           (xcltk_amd.baf.genotype.write_cellsnp_dir) - the fixture INPUT;
  2. tests/golden/genotype/ref_load.json   what the reference's loader (utils/csp_io.load_data, :16-63) reads from that directory:
                                           cells, sites, layer sums per SNP - "the reference's consumers read our files";
  3. tests/golden/genotype/filtered_*/     the reference's filter_snps (baf/genotype.py:200-229: load_data, DP >= minCOUNT,
                                           minMAF <= AD / DP <= 1 - minMAF on the VCF's INFO sums, save_data) on that directory,
                                           for three (minCOUNT, minMAF) pairs - the fixture OUTPUT that
                                           xcltk_amd.baf.genotype.filter_snps is compared with (tests/test_genotype.py, tests/test_gpu_genotype.py).

Nothing of the reference is copied: it is imported in place, only its outputs are kept.  usage: python3 oracle/refgen/make_genotype_goldens.py
"""
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p_ in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402

import oracle as O  # noqa: E402
from xcltk_amd import capi  # noqa: E402
from xcltk_amd.baf import genotype as G  # noqa: E402

PY39 = "/opt/conda/bin/python3.9"
RUNNER = os.path.join(HERE, "run_reference.py")
GOLD = os.path.join(ROOT, "tests", "golden")
DS = os.path.join(GOLD, "datasets", "phasing")
OUT = os.path.join(GOLD, "genotype")
PARAMS = ((20, 0.1), (10, 0.25), (1, 0))


def run_ref(job):
    with tempfile.TemporaryDirectory() as tmp:
        jf = os.path.join(tmp, "job.json")
        with open(jf, "w") as fp:
            json.dump(job, fp)
        r = subprocess.run([PY39, RUNNER, jf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr[-3000:])
            raise SystemExit("reference failed on job %s" % job["kind"])
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    O.build_oracle()
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    # ---- 1. the oracle's pileup of the candidate SNPs -> raw/ through the product's writer
    cand = G.load_candidate_snps(os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"))
    regions = [(G.format_chrom(c), p, p, "%s_%d" % (c, p)) for c, p, _, _ in cand]
    snps = [(G.format_chrom(c), p, r, a, 0, 1) for c, p, r, a in cand]
    cells = [x.strip() for x in open(os.path.join(DS, "barcodes.tsv")) if x.strip()]
    with tempfile.TemporaryDirectory() as tmp:
        rfn, sfn = os.path.join(tmp, "r.tsv"), os.path.join(tmp, "s.tsv")
        open(rfn, "w").write("".join("%s\t%d\t%d\t%s\n" % r for r in regions))
        open(sfn, "w").write("chrom\tpos\tref\talt\tref_hap\talt_hap\n" + "".join("%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
        coo = O.run_files(capi.XCK_MODE_BAF, [os.path.join(DS, "possorted.bam")], rfn, barcode_fn=os.path.join(DS, "barcodes.tsv"), snp_fn=sfn,
                          output_all_reg=True, min_count=1, min_maf=0, no_dup_hap=True)
    raw = os.path.join(OUT, "raw")
    G._write_raw_dir(raw, cand, sorted(cells), {k: tuple(np.asarray(a) for a in v) for k, v in coo.items()})
    # ---- 2. the reference's loader on it
    seen = run_ref(dict(kind="csp_load", kwargs=dict(data_dir=raw)))
    with open(os.path.join(OUT, "ref_load.json"), "w") as fp:
        json.dump(seen, fp, indent=1, sort_keys=True)
    # ---- 3. the reference's filter_snps
    meta = {}
    for mc, mm in PARAMS:
        name = "filtered_c%d_m%s" % (mc, str(mm).replace(".", "p"))
        res = run_ref(dict(kind="filter_snps", kwargs=dict(in_dir=raw, out_dir=os.path.join(OUT, name), min_count=mc, min_maf=mm)))
        meta[name] = dict(min_count=mc, min_maf=mm, p_raw=res["p_raw"], p_new=res["p_new"])
        print(name, meta[name])
    with open(os.path.join(OUT, "cases.json"), "w") as fp:
        json.dump(dict(reference="hxj5/xcltk v0.5.2 filter_snps / csp_io via oracle/refgen/run_reference.py (pysam / anndata stand-ins)",
                       input="raw/ = oracle counts of tests/golden/datasets/phasing written by xcltk_amd.baf.genotype (parity of the COUNTS with cellsnp-lite: unpinned)",
                       cases=meta), fp, indent=1, sort_keys=True)
    print("reference loader saw %d cells x %d SNPs, DP sum %d" % (seen["n_cells"], seen["n_snps"], seen["sum_DP"]))


if __name__ == "__main__":
    main()
