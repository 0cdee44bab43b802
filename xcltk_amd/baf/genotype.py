"""genotype.py - step 1 of `xcltk baf`: per-SNP x cell pileup in cellsnp-lite's output layout.

The reference shells out to the external `cellsnp-lite` binary here (xcltk/baf/genotype.py:144-187) and then re-filters
its output (`filter_snps`, :190-229).  This module produces the same directory with the MI355X engine instead: a candidate
SNP is a one-base feature whose REF allele sits on haplotype 0 and whose ALT allele on haplotype 1, so the allele-specific
feature counting of the hot path (k_join<pileup> ... k_hap_counts) yields exactly AD = UMIs showing ALT, DP = UMIs showing
REF or ALT, OTH = UMIs showing another base, per SNP and cell - no new kernel.  Read filters are the ones cellsnp-lite and
xcltk share (MAPQ >= 20, aligned length >= 30, exclude UNMAP / SECONDARY / QCFAIL (+ DUP without UMIs), no orphans).

PARITY UNPINNED: the binary is absent from the reference tree and from this image, so nothing pins the numbers to
cellsnp-lite's.  Known difference by construction: a (cell, UMI) takes the base of its FIRST read in fetch order
(baf/fc/mcount.py:118-119), cellsnp-lite's own choice inside a UMI group is not restated.  What IS pinned, with the reference's
own code (oracle/refgen/make_genotype_goldens.py, container-only; fixtures under tests/golden/genotype/): a directory written by
write_cellsnp_dir is read by the reference's loader (utils/csp_io.load_data, :16-63) to the same cells, sites and per-SNP sums,
and filter_snps below keeps the same SNPs and matrices as the reference's (baf/genotype.py:200-229) on it, for three
(minCOUNT, minMAF) pairs - tests/test_genotype.py (CPU) and tests/test_gpu_genotype.py (engine pileup -> same fixtures).

Output (`<out_dir>/raw` = every candidate SNP with at least one counted UMI, `<out_dir>` = after `filter_snps`):
cellSNP.base.vcf.gz (bgzip; INFO = AD=..;DP=..;OTH=..), cellSNP.samples.tsv, cellSNP.tag.{AD,DP,OTH}.mtx (SNP x cell).
"""
import os
from logging import info

import numpy as np

from .. import fc_common as fcc
from ..capi import XCK_MODE_BAF
from ..utils.grange import format_chrom
from ..utils.zfile import ZF_F_BGZIP, zopen

VCF_HEADER = ("##fileformat=VCFv4.2\n##source=xcltk_amd pileup (cellsnp-lite layout)\n"
              "##FILTER=<ID=PASS,Description=\"All filters passed\">\n"
              "##INFO=<ID=AD,Number=1,Type=Integer,Description=\"total counts for ALT\">\n"
              "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"total counts for ALT and REF\">\n"
              "##INFO=<ID=OTH,Number=1,Type=Integer,Description=\"total counts for other bases from REF and ALT\">\n")


def load_candidate_snps(vcf_fn):
    """Candidate SNP list: (chrom as written, pos, ref, alt) of every VCF line with single-base REF / ALT in ACGTN."""
    out = []
    with zopen(vcf_fn, "rt") as fp:
        for line in fp:
            if not line or line[0] == "#":
                continue
            p = line.rstrip("\n").split("\t")
            if len(p) < 5:
                continue
            ref, alt = p[3].upper(), p[4].upper()
            if len(ref) != 1 or len(alt) != 1 or ref not in "ACGTN" or alt not in "ACGTN":
                continue
            out.append((p[0], int(p[1]), ref, alt))
    return out


def write_cellsnp_dir(out_dir, snps, cells, mats, is_gzip=True):
    """snps: [(chrom, pos, ref, alt)]; mats: {"AD" | "DP" | "OTH": (row, col, val)} with row = index into snps, col = cell."""
    os.makedirs(out_dir, exist_ok=True)
    sums = {k: np.bincount(np.asarray(m[0], dtype=np.int64), weights=np.asarray(m[2], dtype=np.float64), minlength=len(snps)).astype(np.int64)
            for k, m in mats.items()}
    fn = os.path.join(out_dir, "cellSNP.base.vcf" + (".gz" if is_gzip else ""))
    with zopen(fn, "wb" if is_gzip else "w", ZF_F_BGZIP if is_gzip else None, is_bytes=is_gzip) as fp:
        text = VCF_HEADER + "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n" + "".join(
            "%s\t%d\t.\t%s\t%s\t.\tPASS\tAD=%d;DP=%d;OTH=%d\n" % (s[0], s[1], s[2], s[3], sums["AD"][i], sums["DP"][i], sums["OTH"][i])
            for i, s in enumerate(snps))
        fp.write(text.encode("utf8") if is_gzip else text)
    with open(os.path.join(out_dir, "cellSNP.samples.tsv"), "w") as fp:
        fp.write("".join(c + "\n" for c in cells))
    for k, (row, col, val) in mats.items():
        with open(os.path.join(out_dir, "cellSNP.tag.%s.mtx" % k), "w") as fp:
            fp.write("%%MatrixMarket matrix coordinate integer general\n%\n")
            fp.write("%d\t%d\t%d\n" % (len(snps), len(cells), len(row)))
            fp.write("".join("%d\t%d\t%d\n" % (r + 1, c + 1, v) for r, c, v in zip(np.asarray(row).tolist(), np.asarray(col).tolist(), np.asarray(val).tolist())))
    return fn


def filter_snps(in_dir, out_dir, min_count, min_maf):
    """The reference's second filter (baf/genotype.py:190-229): keep SNPs with DP >= min_count and min_maf <= AD / DP <=
    1 - min_maf, REF and ALT counts only.  -> (output VCF, #SNPs before, #SNPs after)."""
    from ..utils.csp_io import load_data
    d = load_data(in_dir)
    AD = np.asarray(d.AD.sum(axis=0)).reshape(-1).astype(np.int64)
    DP = np.asarray(d.DP.sum(axis=0)).reshape(-1).astype(np.int64)
    with np.errstate(all="ignore"):
        baf = AD / DP
    keep = np.flatnonzero((DP >= min_count) & (baf >= min_maf) & (baf <= 1 - min_maf))
    sub = d.subset_snps(keep)
    mats = {}
    for k, m in (("AD", sub.AD), ("DP", sub.DP), ("OTH", sub.OTH)):
        coo = m.T.tocsr().tocoo()                                 # SNP x cell, row-major order like the raw files
        mats[k] = (coo.row, coo.col, coo.data)
    snps = [(str(c), int(p), str(r), str(a)) for c, p, r, a in zip(sub.chrom, sub.pos, sub.ref, sub.alt)]
    fn = write_cellsnp_dir(out_dir, snps, sub.cells, mats)
    return fn, d.shape[1], sub.shape[1]


def pileup(sam_fn=None, sam_list_fn=None, barcode_fn=None, sample_id_fn=None, sample_id=None, snp_vcf_fn=None,
           out_dir=None, mode="droplet", cell_tag="CB", umi_tag="UB", ncores=1, min_count=20, min_maf=0.1,
           script_fn=None, log_fn=None):
    """Signature of the reference's pileup() (baf/genotype.py:20-187); the engine replaces the cellsnp-lite call.
    Returns (output VCF, #SNPs in <out_dir>/raw, #SNPs after filtering)."""
    if mode not in ("droplet", "well", "bulk"):
        raise ValueError("mode must be droplet, well or bulk")
    conf = type("PileupConf", (), {})()
    conf.sam_fn, conf.sam_list_fn, conf.barcode_fn = sam_fn, sam_list_fn, (barcode_fn if mode == "droplet" else None)
    conf.sample_id_str = sample_id if mode == "bulk" else None
    conf.sample_id_fn = sample_id_fn if mode == "well" else None
    conf.cell_tag = None if str(cell_tag) == "None" else cell_tag
    conf.umi_tag = None if str(umi_tag) == "None" else umi_tag
    conf.excl_flag, conf.debug, conf.nproc = -1, 0, ncores
    conf.min_mapq, conf.min_len, conf.incl_flag, conf.no_orphan = 20, 30, 0, True
    conf.min_count, conf.min_maf, conf.no_dup_hap = 1, 0, True      # per-SNP filters are applied to the sums below
    conf.use_barcodes = lambda: conf.cell_tag is not None
    conf.use_umi = lambda: conf.umi_tag is not None
    conf.defaults = type("D", (), dict(UMI_TAG_BC="UB"))()
    if fcc.resolve_inputs(conf) < 0 or fcc.resolve_tags(conf) < 0:
        raise ValueError("invalid pileup inputs")
    if not snp_vcf_fn or not os.path.isfile(snp_vcf_fn):
        raise ValueError("SNP vcf '%s' does not exist." % snp_vcf_fn)
    cand = load_candidate_snps(snp_vcf_fn)
    if not cand:
        raise ValueError("no candidate SNPs in '%s'." % snp_vcf_fn)
    info("pileup of %d candidate SNPs in %d cells ..." % (len(cand), len(conf.samples)))
    # one feature per SNP: rows of the engine's AD / DP / OTH matrices are then SNPs; REF on haplotype 0, ALT on 1
    regions = [(format_chrom(c), p, p, "%s_%d" % (c, p)) for c, p, _, _ in cand]
    snps = [(format_chrom(c), p, r, a, 0, 1) for c, p, r, a in cand]
    conf.reg_list = regions
    eng, coo, dist = fcc.make_and_count(conf, XCK_MODE_BAF, regions, snps, log_prefix="[pileup]", gather=True)   # rank 0 writes the directory
    # Multi-GPU: only rank 0 holds the gathered matrices and writes the directory; every other rank WAITS for it (step 3 of
    # the pipeline reads this directory on every rank) and learns whether the writer failed, so that all ranks leave together.
    # Whatever goes wrong between here and the status all-reduce - MemoryError while the .mtx text is built, an index error, a
    # full disk - is held back until the all-reduce has run on this rank too: a rank that skips it leaves the others waiting
    # for the backend's timeout.
    result, failure = (None, 0, 0), None
    try:
        try:
            if coo is not None:                                   # the matrices are views of the engine's pinned buffers: keep copies past close()
                coo = {k: tuple(np.array(a) for a in v) for k, v in coo.items()}
        finally:
            eng.close()
        if coo is not None:
            result = _write_pileup_dirs(out_dir, cand, conf.samples, coo, min_count, min_maf)
    except Exception as e:                                        # noqa: BLE001 - re-raised below, after the collective
        failure = e
    if dist.active:
        failed = int(dist.all_reduce_np(np.array([int(failure is not None)], dtype=np.int64), op="max")[0])   # also the barrier
        if failed and failure is None:
            failure = ValueError("the pileup directory could not be written on rank 0")
    if failure is not None:
        raise failure
    return result


def _write_pileup_dirs(out_dir, cand, samples, coo, min_count, min_maf):
    """raw/ (every covered SNP) and the filtered directory, from the gathered AD / DP / OTH matrices (writer rank)."""
    raw_dir = os.path.join(out_dir, "raw")
    _write_raw_dir(raw_dir, cand, samples, coo)
    return filter_snps(raw_dir, out_dir, min_count, min_maf)


def _write_raw_dir(raw_dir, cand, samples, coo):
    """The raw pileup directory: every candidate SNP with at least one counted UMI (coo = engine / oracle matrices, rows = candidates)."""
    covered = np.zeros(len(cand), dtype=bool)
    covered[coo["dp"][0]] = True
    covered[coo["oth"][0]] = True
    idx = np.flatnonzero(covered)
    remap = np.full(len(cand), -1, dtype=np.int64)
    remap[idx] = np.arange(len(idx))
    mats = {k: (remap[np.asarray(coo[m][0], dtype=np.int64)], coo[m][1], coo[m][2]) for k, m in (("AD", "ad"), ("DP", "dp"), ("OTH", "oth"))}
    write_cellsnp_dir(raw_dir, [cand[i] for i in idx.tolist()], samples, mats)
