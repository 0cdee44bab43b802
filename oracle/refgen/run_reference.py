#!/opt/conda/bin/python3.9
"""run_reference.py - CONTAINER-ONLY golden-vector generator (test infrastructure).

Runs the *unmodified* reference (hxj5/xcltk, mounted read-only at /root/reference)
on inputs produced by this repo's synthetic generator and prints/stores its outputs.
The reference is pure Python but imports two third-party packages that do not exist
in this image: `pysam` (BAM access) and `anndata` (only touched when cellsnp_dir is
given).  As recorded in SURVEY.md section 8c they are replaced by stand-ins:

  * pysam   -> oracle/pybam.py (AlignmentFile.fetch + AlignedSegment accessors restated
               from the SAM spec / pysam documentation);
  * anndata -> oracle/refgen/anndata_standin.py (a minimal AnnData; touched only when cellsnp_dir is given).

`intervaltree` (xcltk/utils/grange.py:4) exists only for /opt/conda/bin/python3.9, hence
the interpreter.  Nothing from the reference is copied: it is imported in place with
sys.dont_write_bytecode = True (the tree is read-only) and only its *outputs* are kept
as fixtures under tests/golden/.

usage: run_reference.py <job.json>
  job = {"kind": "basefc"|"baf"|"convert"|"filter_snps"|"csp_load", "out_dir": ..., "kwargs": {...}}   (kwargs of fc_wrapper /
  afc_wrapper, xcltk/rdr/fc/main.py:142 and xcltk/baf/fc/main.py:32); optional "argv" (basefc command line
  instead of kwargs) and "prewarm" (BAMs parsed before the timed call)
"""
import json
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # oracle/ -> pybam
sys.path.insert(0, HERE)                           # oracle/refgen -> anndata_standin
REF = os.environ.get("XCLTK_REFERENCE", "/root/reference")


def install_standins():
    import pybam
    m = types.ModuleType("pysam")
    m.AlignmentFile = pybam.AlignmentFile
    m.BGZFile = pybam.BGZFile
    m.__version__ = "0.0-standin"
    sys.modules["pysam"] = m
    import anndata_standin                                    # minimal AnnData: only the local-phasing path touches it
    a = types.ModuleType("anndata")
    a.AnnData = anndata_standin.AnnData
    sys.modules["anndata"] = a


def main():
    with open(sys.argv[1]) as fp:
        job = json.load(fp)
    install_standins()
    sys.path.insert(0, REF)
    import logging
    import time
    logging.disable(logging.CRITICAL)
    # "prewarm": BAMs to parse into the stand-in's cache before the timed call; the reference's forked workers then
    # open them for free, like pysam opening an indexed file (used by time_reference.py only)
    t_load = time.time()
    for fn in job.get("prewarm", ()):
        sys.modules["pysam"].AlignmentFile(fn, "r")
    t_load = time.time() - t_load
    t_run = time.time()
    if job["kind"] == "basefc":
        from xcltk.rdr.fc.main import fc_wrapper
        # NOTE reference quirk (rdr/fc/main.py:177-178): a non-None excl_flag is never
        # copied into conf.  To exercise explicit exclude flags we go through the CLI
        # entry point instead when "argv" is given.
        if "argv" in job:
            from xcltk.rdr.fc.main import fc_main
            ret = fc_main(["xcltk", "basefc"] + job["argv"])
        else:
            ret = fc_wrapper(**job["kwargs"])
    elif job["kind"] == "baf":
        from xcltk.baf.fc.main import afc_wrapper
        ret = afc_wrapper(**job["kwargs"])
    elif job["kind"] == "convert":
        # xcltk/tools/convert.py:15 - returns None, exits non-zero on errors
        from xcltk.tools.convert import convert_main
        convert_main(["xcltk", "convert"] + job["argv"])
        ret = 0
    elif job["kind"] == "filter_snps":
        # the reference's own post-filter of a cellsnp-lite style pileup directory (xcltk/baf/genotype.py:200-229), which loads the
        # directory with utils/csp_io.load_data (:16-63) and writes the kept SNPs with save_data (:66-100)
        from xcltk.baf.genotype import filter_snps
        vcf, p_raw, p_new = filter_snps(**job["kwargs"])
        sys.stdout.write(json.dumps({"ret": 0, "vcf": vcf, "p_raw": int(p_raw), "p_new": int(p_new)}) + "\n")
        return 0
    elif job["kind"] == "csp_load":
        # the reference's loader on a directory written by this repo's writer: shapes and layer sums as the reference sees them
        from xcltk.utils.csp_io import load_data
        ad = load_data(job["kwargs"]["data_dir"])
        n, p = ad.shape
        out = dict(ret=0, n_cells=int(n), n_snps=int(p), cells=[str(x) for x in ad.obs["cell"]], pos=[int(x) for x in ad.var["pos"]],
                   chrom=[str(x) for x in ad.var["chrom"]], ref=[str(x) for x in ad.var["ref"]], alt=[str(x) for x in ad.var["alt"]])
        for k in ("AD", "DP", "OTH"):
            m = ad.layers[k]
            out["sum_" + k] = int(m.sum()); out["colsum_" + k] = [int(x) for x in m.sum(axis=0)]
        sys.stdout.write(json.dumps(out) + "\n")
        return 0
    elif job["kind"] == "fet1":
        # direct per-region call used for the known-answer tests of SURVEY 8c
        raise SystemExit("fet1 jobs are handled by kat.py")
    else:
        raise SystemExit("unknown job kind")
    sys.stdout.write(json.dumps({"ret": ret, "standin_load_s": t_load, "elapsed_s": time.time() - t_run}) + "\n")
    return 0 if ret == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
