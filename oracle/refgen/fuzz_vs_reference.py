#!/usr/bin/env python3
"""fuzz_vs_reference.py - CONTAINER-ONLY: random datasets and option sets through the UNMODIFIED reference
(oracle/refgen/run_reference.py: /opt/conda/bin/python3.9, pysam -> oracle/pybam.py) and through the oracle's whole-run
path (oracle.run_files); the output directories must be identical byte for byte.  Widens the pin of the oracle beyond
the 25 committed golden cases.  Test infrastructure - nothing here is imported by the product.

usage: fuzz_vs_reference.py [n_cases] [first_seed]"""
import json, os, shutil, subprocess, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import oracle as O
from xcltk_amd import capi
from xcltk_amd.synth.generate import make_10x_dataset, make_smartseq_dataset

RUNNER = os.path.join(HERE, "run_reference.py")


class RefRaises(Exception):
    pass


def run_ref(kind, kwargs, out_dir, argv=None):
    job = dict(kind=kind, out_dir=out_dir, kwargs=kwargs)
    if argv is not None:
        job["argv"] = argv
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fp:
        json.dump(job, fp)
    try:
        r = subprocess.run(["/opt/conda/bin/python3.9", RUNNER, fp.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    finally:
        os.unlink(fp.name)
    if r.returncode != 0:
        # a CIGAR-less read reaching the fractional include test makes the reference compare None < float
        # (rdr/fc/core.py:35,161); the oracle and the engine drop such a read instead (oracle/xck_oracle.c run_basefc)
        if "'<' not supported between instances of 'NoneType' and 'float'" in r.stderr:
            raise RefRaises()
        raise RuntimeError("reference failed: " + r.stderr[-800:])


def same(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    if fa != fb:
        return "file lists differ: %s vs %s" % (fa, fb)
    for f in fa:
        if open(os.path.join(a, f), "rb").read() != open(os.path.join(b, f), "rb").read():
            return f + " differs"
    return None


def one(seed, work):
    rng = np.random.default_rng(77000 + seed)
    bad = []
    if seed % 4 == 3:                                           # well-based, UMI-less
        d = make_smartseq_dataset(work + "/ds", n_cells=int(rng.integers(2, 7)), reads_per_cell=int(rng.integers(200, 700)),
                                  n_snps=int(rng.integers(50, 300)), n_genes=int(rng.integers(5, 40)), contigs=(("1", 300000), ("2", 200000)), seed=seed)
        common = dict(no_orphan=bool(rng.integers(0, 2)), min_mapq=int(rng.choice([0, 20])), min_len=int(rng.choice([0, 30])))
        kw = dict(sam_fn=None, barcode_fn=None, region_fn=d["regions"], sam_list_fn=d["sam_list"], sample_id_fn=d["sample_list"], cell_tag="None", umi_tag="None", ncores=1, **common)
        run_ref("basefc", dict(kw, out_dir=work + "/ref_fc"), work + "/ref_fc")
        O.run_files(capi.XCK_MODE_BASEFC, d["bams"], d["regions"], out_dir=work + "/ora_fc", cell_tag=None, umi_tag=None, sample_ids=d["sample_ids"], **common)
        bad.append(("basefc", same(work + "/ref_fc", work + "/ora_fc")))
        bopts = dict(min_count=int(rng.choice([1, 3])), min_maf=float(rng.choice([0, 0.1])), no_dup_hap=bool(rng.integers(0, 2)), output_all_reg=bool(rng.integers(0, 2)))
        run_ref("baf", dict(kw, phased_snp_fn=d["snps_tsv"], out_dir=work + "/ref_baf", **bopts), work + "/ref_baf")
        O.run_files(capi.XCK_MODE_BAF, d["bams"], d["regions"], out_dir=work + "/ora_baf", snp_fn=d["snps_tsv"], cell_tag=None, umi_tag=None,
                    sample_ids=d["sample_ids"], **bopts, **common)
        bad.append(("baf", same(work + "/ref_baf", work + "/ora_baf")))
        return bad
    n_contigs = int(rng.integers(1, 4))
    contigs = tuple(("chr%d" % (i + 1), int(rng.integers(150000, 600000))) for i in range(n_contigs))
    d = make_10x_dataset(work + "/ds", n_reads=int(rng.integers(1000, 4000)), n_barcodes=int(rng.integers(5, 80)),
                         n_snps=int(rng.integers(50, 500)), n_genes=int(rng.integers(5, 50)), contigs=contigs, seed=seed,
                         bam_contig_prefix=[None, "", "chr"][int(rng.integers(0, 3))], align_records=bool(rng.integers(0, 2)),
                         n_bams=int(rng.integers(1, 3)), paired=bool(rng.integers(0, 2)), umi_len=int(rng.integers(6, 15)),
                         frac_cb_outside=float(rng.uniform(0, 0.1)), iupac_frac=float(rng.uniform(0, 0.03)),
                         odd_frac=float(rng.choice([0, 0.05, 0.2])))
    d_odd = d["odd_frac"]
    if rng.integers(0, 4) == 0:                                 # fixed-size bins as features (utils/gregion.py)
        from xcltk_amd.utils import gregion as G
        bins = G.get_fixsize_reg_from_input_len(dict(contigs), int(rng.choice([1, 5, 20, 100])))
        d["regions"] = work + "/ds/bins.tsv"
        G.output_feature_table(bins, d["regions"])
    sam = ",".join(d["bams"])
    umi_tag = ["UB", "UB", "None"][int(rng.integers(0, 3))]
    common = dict(min_mapq=int(rng.choice([0, 2, 20, 30])), min_len=int(rng.choice([0, 30, 60, 91])),
                  incl_flag=int(rng.choice([0, 0, 16])), no_orphan=bool(rng.integers(0, 2)))
    min_include = [0.9, 0.5, 0.1, 0, 1, 30, 91, 0.999][int(rng.integers(0, 8))]
    all_reg = bool(rng.integers(0, 2))
    fc_excl = [None, 0, 256][int(rng.integers(0, 3))]
    kw = dict(sam_fn=sam, barcode_fn=d["barcodes"], region_fn=d["regions"], umi_tag=umi_tag, ncores=int(rng.integers(1, 3)), **common)
    try:
        if fc_excl is None:
            run_ref("basefc", dict(kw, out_dir=work + "/ref_fc", output_all_reg=all_reg, min_include=min_include), work + "/ref_fc")
        else:
            # fc_wrapper() never copies a non-None excl_flag into its config (rdr/fc/main.py:177-178): the command line does
            all_reg = True
            argv = ["-s", sam, "-b", d["barcodes"], "-R", d["regions"], "-O", work + "/ref_fc", "-p", str(kw["ncores"]), "--UMItag", umi_tag,
                    "--inclFLAG", str(common["incl_flag"]), "--exclFLAG", str(fc_excl), "--minLEN", str(common["min_len"]),
                    "--minMAPQ", str(common["min_mapq"]), "--minINCLUDE", repr(min_include)]
            if not common["no_orphan"]:
                argv.append("--countORPHAN")
            run_ref("basefc", {}, work + "/ref_fc", argv=argv)
        O.run_files(capi.XCK_MODE_BASEFC, d["bams"], d["regions"], out_dir=work + "/ora_fc", barcode_fn=d["barcodes"], umi_tag=umi_tag,
                    output_all_reg=all_reg, min_include=min_include, excl_flag=fc_excl, **common)
        bad.append(("basefc", same(work + "/ref_fc", work + "/ora_fc")))
    except RefRaises:
        if not (0 < min_include < 1 and common["min_len"] == 0 and d_odd > 0):
            raise
        print("seed %d basefc: reference raises on a CIGAR-less read in fraction mode (expected)" % seed, flush=True)
    excl = [None, 0, 1024, 772][int(rng.integers(0, 4))]
    bopts = dict(min_count=int(rng.choice([0, 1, 3, 11])), min_maf=float(rng.choice([0, 0.05, 0.1, 0.3])), no_dup_hap=bool(rng.integers(0, 2)))
    snp_fn = d["snps_vcf"] if rng.integers(0, 2) else d["snps_tsv"]
    run_ref("baf", dict(kw, phased_snp_fn=snp_fn, out_dir=work + "/ref_baf", output_all_reg=all_reg, excl_flag=excl, **bopts), work + "/ref_baf")
    O.run_files(capi.XCK_MODE_BAF, d["bams"], d["regions"], out_dir=work + "/ora_baf", barcode_fn=d["barcodes"], snp_fn=snp_fn, umi_tag=umi_tag,
                output_all_reg=all_reg, excl_flag=excl, **bopts, **common)
    bad.append(("baf", same(work + "/ref_baf", work + "/ora_baf")))
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fails = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        work = tempfile.mkdtemp(prefix="xck_fuzzref_")
        try:
            res = one(seed, work)
            for kind, msg in res:
                if msg:
                    fails += 1
                    print("seed %d %s MISMATCH: %s (kept in %s)" % (seed, kind, msg, work), flush=True)
            if not any(m for _, m in res):
                print("seed %d ok" % seed, flush=True)
                shutil.rmtree(work, ignore_errors=True)
        except Exception as e:
            fails += 1
            print("seed %d ERROR %s: %s" % (seed, type(e).__name__, str(e)[-400:]), flush=True)
    print("%d cases, %d problems, %.0fs" % (n, fails, time.time() - t0))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
