"""csp_io.py - reading (and writing) the output directory of a cellsnp-lite style pileup.

Files (reference xcltk/utils/csp_io.py:16-63): cellSNP.base.vcf[.gz] (one line per SNP: CHROM POS ID REF ALT ...),
cellSNP.samples.tsv (one cell per line), cellSNP.tag.{AD,DP,OTH}.mtx (SNP x cell, MatrixMarket).  The reference wraps
them in an AnnData (cell x SNP, dense layers); here they stay a small container with scipy sparse matrices - anndata is
not required, and a dense 1 M SNP x 10 k cell layer would not fit in memory.  `to_anndata()` gives the reference's object
when anndata is installed.
"""
import gzip
import os

import numpy as np
from scipy import io as spio
from scipy import sparse


class CellSnpData(object):
    """cell x SNP matrices AD / DP / OTH (CSC), SNP columns (chrom, pos, ref, alt as in the VCF) and cell names."""

    def __init__(self, chrom, pos, ref, alt, cells, AD, DP, OTH, vcf_comment="", vcf_rest=None):
        self.chrom = np.asarray(chrom, dtype=object)
        self.pos = np.asarray(pos, dtype=np.int64)
        self.ref, self.alt = np.asarray(ref, dtype=object), np.asarray(alt, dtype=object)
        self.cells = list(cells)
        self.AD, self.DP, self.OTH = (sparse.csc_matrix(m) for m in (AD, DP, OTH))
        self.vcf_comment, self.vcf_rest = vcf_comment, vcf_rest
        n_cell, n_snp = len(self.cells), len(self.pos)
        for m in (self.AD, self.DP, self.OTH):
            if m.shape != (n_cell, n_snp):
                raise ValueError("matrix shape %s does not match %d cells x %d SNPs" % (m.shape, n_cell, n_snp))

    @property
    def shape(self):
        return (len(self.cells), len(self.pos))

    def subset_snps(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        return CellSnpData(self.chrom[idx], self.pos[idx], self.ref[idx], self.alt[idx], self.cells,
                           self.AD[:, idx], self.DP[:, idx], self.OTH[:, idx], self.vcf_comment,
                           None if self.vcf_rest is None else [self.vcf_rest[i] for i in idx.tolist()])

    def subset_cells(self, keep):
        keep = np.asarray(keep, dtype=bool)
        rows = np.flatnonzero(keep)
        return CellSnpData(self.chrom, self.pos, self.ref, self.alt, [self.cells[i] for i in rows.tolist()],
                           self.AD.tocsr()[rows, :], self.DP.tocsr()[rows, :], self.OTH.tocsr()[rows, :],
                           self.vcf_comment, self.vcf_rest)

    def to_anndata(self):
        """The object xcltk.utils.csp_io.load_data() returns (cell x SNP AnnData with dense layers)."""
        import anndata as ad
        import pandas as pd
        var = pd.DataFrame(dict(chrom=self.chrom.astype(str), pos=self.pos, ref=self.ref.astype(str), alt=self.alt.astype(str)))
        obs = pd.DataFrame(dict(cell=self.cells))
        adata = ad.AnnData(X=None, obs=obs, var=var)
        for k, m in (("AD", self.AD), ("DP", self.DP), ("OTH", self.OTH)):
            adata.layers[k] = m.toarray()
        return adata


def _open_text(fn):
    return gzip.open(fn, "rt") if fn.lower().endswith(".gz") else open(fn, "r")


def load_vcf_sites(fn):
    """-> (comment text, chrom[], pos[], ref[], alt[], rest-of-line[]) of a VCF (header lines start with '#')."""
    comment, chrom, pos, ref, alt, rest = [], [], [], [], [], []
    last = None
    with _open_text(fn) as fp:
        for line in fp:
            if line.startswith("#"):
                comment.append(line)
                last = line
                continue
            p = line.rstrip("\n").split("\t")
            if len(p) < 5:
                continue
            chrom.append(p[0]); pos.append(int(p[1])); ref.append(p[3]); alt.append(p[4]); rest.append(p)
    if last is None or not last.startswith("#CHROM"):
        raise IOError("'%s': no #CHROM header line" % fn)
    return "".join(comment), chrom, pos, ref, alt, rest


def load_data(data_dir, is_gzip=True):
    """Read a cellsnp-lite output directory (reference csp_io.load_data, :16-63)."""
    vcf = os.path.join(data_dir, "cellSNP.base.vcf" + (".gz" if is_gzip else ""))
    if not os.path.isfile(vcf) and is_gzip and os.path.isfile(vcf[:-3]):
        vcf = vcf[:-3]
    comment, chrom, pos, ref, alt, rest = load_vcf_sites(vcf)
    with open(os.path.join(data_dir, "cellSNP.samples.tsv")) as fp:
        cells = [x.rstrip("\n").split("\t")[0] for x in fp if x.strip()]
    mats = [sparse.csc_matrix(spio.mmread(os.path.join(data_dir, "cellSNP.tag.%s.mtx" % k))).T.tocsc() for k in ("AD", "DP", "OTH")]
    return CellSnpData(chrom, pos, ref, alt, cells, mats[0], mats[1], mats[2], comment, rest)
