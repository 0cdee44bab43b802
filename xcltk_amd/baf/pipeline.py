"""`xcltk baf` command line.

The reference pipeline (xcltk/baf/pipeline.py:73-363) has three steps: (1) pileup with the
external `cellsnp-lite` binary, (2) reference phasing with the external `eagle` binary and
bcftools, (3) allele-specific feature counting (`afc_wrapper`).  Only step 3 is compute that
lives in the reference repository, and it is the part this package accelerates.  The command
keeps the reference's options; in addition `--phasedSNP FILE` supplies an already phased
SNP file (VCF or TSV) and runs step 3 directly.  Without it, steps 1-2 would have to shell
out to binaries that are not part of either code base, so the command stops with an
explanatory error (`xcltk baf` row).
"""

import getopt
import os
import sys
from logging import error, info

from ..config import APP, VERSION
from ..utils.xlog import init_logging
from .fc.main import afc_wrapper as baf_fc

COMMAND = "baf"
CELL_TAG, UMI_TAG = "CB", "UB"
MIN_COUNT, MIN_MAF, N_CORES = 11, 0.1, 1


def usage(fp=sys.stdout):
    s = "\n"
    s += "Version: %s\n" % VERSION
    s += "Usage:   %s %s [options]\n" % (APP, COMMAND)
    s += "\n"
    s += "Options:\n"
    s += "  --label STR        Task label.\n"
    s += "  --sam FILE         Comma separated indexed BAM file(s) (CRAM: convert first).\n"
    s += "  --samList FILE     A file listing BAM files, each per line.\n"
    s += "  --barcode FILE     A plain file listing all effective cell barcodes, for\n"
    s += "                     droplet-based data, e.g., 10x Genomics.\n"
    s += "  --sampleList FILE  A plain file listing sample IDs, one ID per BAM, for\n"
    s += "                     well-based data, e.g., SMART-seq.\n"
    s += "  --snpvcf FILE      A VCF file listing all candidate SNPs.\n"
    s += "  --region FILE      A TSV file listing target features. The first 4 columns are:\n"
    s += "                     chrom, start, end (both 1-based and inclusive), name.\n"
    s += "  --outdir DIR       Output dir.\n"
    s += "  --gmap FILE        Path to genetic map provided by Eagle2\n"
    s += "                     (e.g. Eagle_v2.4.1/tables/genetic_map_hg38_withX.txt.gz).\n"
    s += "  --eagle FILE       Path to Eagle2 binary file.\n"
    s += "  --paneldir DIR     Directory to phasing reference panel (BCF files).\n"
    s += "  --version          Print version and exit.\n"
    s += "  --help             Print this message and exit.\n"
    s += "\n"
    s += "Optional arguments:\n"
    s += "  --refCell FILE     A plain file listing reference cells, one per line.\n"
    s += "  --cellTAG STR      Cell barcode tag; Set to None if not available [%s]\n" % CELL_TAG
    s += "  --UMItag STR       UMI tag; Set to None if not available [%s]\n" % UMI_TAG
    s += "  --minCOUNT INT     Mininum aggragated count for SNP [%d]\n" % MIN_COUNT
    s += "  --minMAF FLOAT     Mininum minor allele fraction for SNP [%f]\n" % MIN_MAF
    s += "  --ncores INT       Number of threads [%d]\n" % N_CORES
    s += "  --phasedSNP FILE   (this engine) phased SNP VCF/TSV: skip steps 1-2 and run the\n"
    s += "                     allele-specific feature counting on the GPU directly.\n"
    s += "\n"
    s += "Notes:\n"
    s += "1. One and only one of `--sam` and `--samlist` should be specified.\n"
    s += "2. For well-based data, the order of the BAM files (in `--sam` or `--samlist`)\n"
    s += "   and the sample IDs (in `--sampleList`) should match each other.\n"
    s += "3. For bulk data, the label (`--label`) will be used as the sample ID.\n"
    s += "\n"
    fp.write(s)


def pipeline_main(argv):
    if len(argv) <= 2:
        usage()
        sys.exit(0)
    init_logging(stream=sys.stdout)
    o = dict(label=None, sam=None, samlist=None, barcode=None, samplelist=None, snpvcf=None,
             region=None, outdir=None, gmap=None, eagle=None, paneldir=None, refcell=None,
             celltag=CELL_TAG, umitag=UMI_TAG, mincount=MIN_COUNT, minmaf=MIN_MAF, ncores=N_CORES,
             phasedsnp=None)
    conv = dict(mincount=int, minmaf=float, ncores=int)
    opts, _ = getopt.getopt(argv[2:], "", [
        "label=", "sam=", "samList=", "barcode=", "sampleList=", "snpvcf=", "region=", "outdir=",
        "gmap=", "eagle=", "paneldir=", "version", "help", "refCell=", "cellTAG=", "UMItag=",
        "minCOUNT=", "minMAF=", "ncores=", "phasedSNP="])
    for op, val in opts:
        key = op.lower().lstrip("-")
        if key == "version":
            sys.stdout.write(VERSION + "\n")
            sys.exit(0)
        if key == "help":
            usage()
            sys.exit(0)
        if key not in o:
            error("invalid option: '%s'." % op)
            return -1
        o[key] = conv.get(key, str)(val)
    ret = pipeline_wrapper(
        label=o["label"], sam_fn=o["sam"], sam_list_fn=o["samlist"], barcode_fn=o["barcode"],
        sample_id_fn=o["samplelist"], snp_vcf_fn=o["snpvcf"], region_fn=o["region"], out_dir=o["outdir"],
        gmap_fn=o["gmap"], eagle_fn=o["eagle"], panel_dir=o["paneldir"], ref_cell_fn=o["refcell"],
        cell_tag=o["celltag"], umi_tag=o["umitag"], min_count=o["mincount"], min_maf=o["minmaf"],
        ncores=o["ncores"], phased_snp_fn=o["phasedsnp"])
    info("All Done!")
    return ret


def pipeline_wrapper(label, sam_fn=None, sam_list_fn=None, barcode_fn=None, sample_id_fn=None,
                     snp_vcf_fn=None, region_fn=None, out_dir=None, gmap_fn=None, eagle_fn=None,
                     panel_dir=None, ref_cell_fn=None, cell_tag="CB", umi_tag="UB", min_count=11,
                     min_maf=0.1, ncores=1, phased_snp_fn=None):
    """Reference signature (baf/pipeline.py:155-170) plus `phased_snp_fn`."""
    if not out_dir:
        error("out dir needed!")
        return -1
    os.makedirs(out_dir, exist_ok=True)
    sample_id = None
    mode = "droplet" if barcode_fn is not None else ("well" if sample_id_fn is not None else "bulk")
    if mode == "bulk":
        sample_id = label                      # bulk: the label is the sample id
    # step 1: per-SNP x cell pileup of the candidate SNPs.  The reference runs the external cellsnp-lite binary here
    # (baf/genotype.py:144-187); this engine produces the same directory itself (baf/genotype.py of this package).
    pileup_dir = None
    if snp_vcf_fn is not None:
        from .genotype import pileup
        pileup_dir = os.path.join(out_dir, "1_pileup")
        info("start genotyping ...")
        try:
            vcf, p_raw, p_new = pileup(sam_fn=sam_fn, sam_list_fn=sam_list_fn, barcode_fn=barcode_fn, sample_id_fn=sample_id_fn,
                                       sample_id=sample_id, snp_vcf_fn=snp_vcf_fn, out_dir=pileup_dir, mode=mode,
                                       cell_tag=cell_tag, umi_tag=umi_tag, ncores=ncores, min_count=min_count, min_maf=min_maf)
        except Exception as e:                 # noqa: BLE001 - every rank of a multi-GPU run leaves step 1 the same way (pileup() spreads a failure to all ranks)
            error("pileup failed: %s%s" % (type(e).__name__ + ": " if not isinstance(e, (ValueError, IOError, OSError)) else "", e))
            return -1
        if vcf is not None:
            info("pileup #SNP raw=%d; post-filtering=%d." % (p_raw, p_new))
            info("pileup VCF is '%s'." % vcf)
    # step 2: reference phasing with Eagle2 - an external binary plus a phasing panel (baf/refphase.py:112-148) that this
    # engine does not replace: the phased SNP list has to be given
    if phased_snp_fn is None:
        error("step 2 of `%s baf` calls the external binary eagle with a phasing panel (baf/refphase.py:112-148), which this "
              "engine does not replace; phase the pileup VCF%s separately and pass the result with --phasedSNP."
              % (APP, " ('%s')" % os.path.join(pileup_dir, "cellSNP.base.vcf.gz") if pileup_dir else ""))
        return -1
    fc_dir = os.path.join(out_dir, "3_baf_fc")
    os.makedirs(fc_dir, exist_ok=True)
    info("BAF feature counting ...")
    # same arguments as the reference's call of baf_fc (baf/pipeline.py:341-360): the pileup directory drives the local phasing
    ret = baf_fc(sam_fn=sam_fn, barcode_fn=barcode_fn, region_fn=region_fn, phased_snp_fn=phased_snp_fn,
                 out_dir=fc_dir, sam_list_fn=sam_list_fn, sample_ids=sample_id, sample_id_fn=sample_id_fn,
                 debug_level=0, ncores=ncores, cellsnp_dir=pileup_dir, ref_cell_fn=ref_cell_fn,
                 cell_tag=cell_tag, umi_tag=umi_tag, min_count=1, min_maf=0, output_all_reg=True,
                 no_dup_hap=True, min_mapq=20, min_len=30, incl_flag=0, excl_flag=None, no_orphan=True)
    info("feature BAFs are at '%s'." % fc_dir)
    return ret
