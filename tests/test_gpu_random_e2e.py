"""Randomised end-to-end parity: seeded random datasets and option sets, product front-ends on the GPU vs
the oracle's whole-run path (independent BAM reader, encoder, C restatement, writers), byte for byte."""
import os

import numpy as np
import pytest

import oracle as O
import util
from xcltk_amd import capi
from xcltk_amd.synth.generate import make_10x_dataset, make_smartseq_dataset

pytestmark = pytest.mark.gpu


def _cmp_dirs(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    assert fa == fb
    for f in fa:
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f


_SEEDS = list(range(1, 13)) if not os.environ.get("XCK_E2E_SEEDS") else list(range(1, 1 + int(os.environ["XCK_E2E_SEEDS"])))   # bigger one-off sweeps


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_10x_dataset_and_options(seed, tmp_path):
    from xcltk_amd.baf.fc.main import afc_wrapper
    from xcltk_amd.rdr.fc.main import fc_main, fc_wrapper
    rng = np.random.default_rng(1000 + seed)
    n_contigs = int(rng.integers(1, 4))
    contigs = tuple(("chr%d" % (i + 1), int(rng.integers(150000, 600000))) for i in range(n_contigs))
    n_bams = int(rng.integers(1, 3))
    d = make_10x_dataset(str(tmp_path / "ds"), n_reads=int(rng.integers(1500, 6000)), n_barcodes=int(rng.integers(5, 120)),
                         n_snps=int(rng.integers(50, 600)), n_genes=int(rng.integers(5, 60)), contigs=contigs, seed=seed,
                         bam_contig_prefix=[None, "", "chr"][int(rng.integers(0, 3))], align_records=bool(rng.integers(0, 2)),
                         n_bams=n_bams, paired=bool(rng.integers(0, 2)), umi_len=int(rng.integers(6, 15)),
                         frac_cb_outside=float(rng.uniform(0, 0.1)), iupac_frac=float(rng.uniform(0, 0.03)),
                         odd_frac=float(rng.choice([0, 0.05, 0.2])))          # unmapped flag with a CIGAR, CIGAR-less reads
    sam = ",".join(d["bams"])
    umi_tag = ["UB", "UB", "None"][int(rng.integers(0, 3))]
    common = dict(min_mapq=int(rng.choice([0, 2, 20, 30])), min_len=int(rng.choice([0, 30, 60, 91])),
                  incl_flag=int(rng.choice([0, 0, 16])), no_orphan=bool(rng.integers(0, 2)))
    min_include = [0.9, 0.5, 0.1, 0, 1, 30, 91, 0.999][int(rng.integers(0, 8))]
    all_reg = bool(rng.integers(0, 2))
    # ---- basefc
    out = str(tmp_path / "fc"); ref = str(tmp_path / "fc_ref")
    fc_excl = [None, 0, 256][int(rng.integers(0, 3))]
    ncores = int(rng.integers(1, 5))
    if fc_excl is None:
        assert fc_wrapper(sam, d["barcodes"], d["regions"], out, umi_tag=umi_tag, output_all_reg=all_reg,
                          min_include=min_include, ncores=ncores, **common) == 0
    else:
        # like the reference's (rdr/fc/main.py:177-178), fc_wrapper() ignores a non-None excl_flag: only the command line sets it
        all_reg = True
        argv = ["xcltk", "basefc", "-s", sam, "-b", d["barcodes"], "-R", d["regions"], "-O", out, "-p", str(ncores), "--UMItag", umi_tag,
                "--inclFLAG", str(common["incl_flag"]), "--exclFLAG", str(fc_excl), "--minLEN", str(common["min_len"]),
                "--minMAPQ", str(common["min_mapq"]), "--minINCLUDE", repr(min_include)] + ([] if common["no_orphan"] else ["--countORPHAN"])
        assert fc_main(argv) == 0
    O.run_files(capi.XCK_MODE_BASEFC, d["bams"], d["regions"], out_dir=ref, barcode_fn=d["barcodes"], umi_tag=umi_tag,
                output_all_reg=all_reg, min_include=min_include, excl_flag=fc_excl, **common)
    _cmp_dirs(out, ref)
    # ---- baf
    excl = [None, 0, 1024, 772][int(rng.integers(0, 4))]
    bopts = dict(min_count=int(rng.choice([0, 1, 3, 11])), min_maf=float(rng.choice([0, 0.05, 0.1, 0.3])),
                 no_dup_hap=bool(rng.integers(0, 2)))
    snp_fn = d["snps_vcf"] if rng.integers(0, 2) else d["snps_tsv"]
    out = str(tmp_path / "baf"); ref = str(tmp_path / "baf_ref")
    assert afc_wrapper(sam, d["barcodes"], d["regions"], snp_fn, out, umi_tag=umi_tag, output_all_reg=all_reg,
                       excl_flag=excl, ncores=2, **bopts, **common) == 0
    O.run_files(capi.XCK_MODE_BAF, d["bams"], d["regions"], out_dir=ref, barcode_fn=d["barcodes"], snp_fn=snp_fn, umi_tag=umi_tag,
                output_all_reg=all_reg, excl_flag=excl, **bopts, **common)
    _cmp_dirs(out, ref)


@pytest.mark.parametrize("seed", [3, 4])
def test_random_well_mode(seed, tmp_path):
    from xcltk_amd.fused import fused_wrapper
    rng = np.random.default_rng(2000 + seed)
    d = make_smartseq_dataset(str(tmp_path / "ds"), n_cells=int(rng.integers(3, 9)), reads_per_cell=int(rng.integers(300, 900)),
                              n_snps=200, n_genes=30, contigs=(("1", 300000), ("2", 200000)), seed=seed)
    out = str(tmp_path / "fused")
    assert fused_wrapper(None, None, d["regions"], d["snps_tsv"], out, sam_list_fn=d["sam_list"], sample_id_fn=d["sample_list"],
                         cell_tag="None", umi_tag="None", ncores=2, no_orphan=bool(seed & 1)) == 0
    common = dict(cell_tag=None, umi_tag=None, sample_ids=d["sample_ids"], no_orphan=bool(seed & 1))
    O.run_files(capi.XCK_MODE_BASEFC, d["bams"], d["regions"], out_dir=str(tmp_path / "r1"), **common)
    O.run_files(capi.XCK_MODE_BAF, d["bams"], d["regions"], out_dir=str(tmp_path / "r2"), snp_fn=d["snps_tsv"], **common)
    _cmp_dirs(os.path.join(out, "basefc"), str(tmp_path / "r1"))
    _cmp_dirs(os.path.join(out, "baf"), str(tmp_path / "r2"))


@pytest.mark.parametrize("bin_kb,min_include", [(1, 0.9), (10, 0.5), (50, 0), (500, 30)])
def test_fixed_size_bins_as_features(bin_kb, min_include, tmp_path):
    """Features = fixed-size bins over the BAM header's contigs (utils/gregion.py): adjacent, non-overlapping
    regions, the last one reaching past the end of its contig; reads straddling a bin boundary fail the include test in
    both bins unless min_include is small."""
    from xcltk_amd.baf.fc.main import afc_wrapper
    from xcltk_amd.rdr.fc.main import fc_wrapper
    from xcltk_amd.utils import gregion as G
    d = make_10x_dataset(str(tmp_path / "ds"), n_reads=6000, n_barcodes=40, n_snps=400, n_genes=30,
                         contigs=(("chr1", 400000), ("chr2", 250000)), seed=90 + bin_kb, bam_contig_prefix="chr")
    bins = G.get_fixsize_reg_from_sam_header(["1", "chr2"], bin_kb, d["bam"])          # "1" resolves to chr1
    assert bins[0].chrom == "chr1" and bins[-1].end >= 250000
    feat = str(tmp_path / "bins.tsv")
    G.output_feature_table(bins, feat)
    out, ref = str(tmp_path / "fc"), str(tmp_path / "fc_ref")
    assert fc_wrapper(d["bam"], d["barcodes"], feat, out, min_include=min_include, ncores=2) == 0
    O.run_files(capi.XCK_MODE_BASEFC, d["bams"], feat, out_dir=ref, barcode_fn=d["barcodes"], min_include=min_include)
    _cmp_dirs(out, ref)
    out, ref = str(tmp_path / "baf"), str(tmp_path / "baf_ref")
    assert afc_wrapper(d["bam"], d["barcodes"], feat, d["snps_tsv"], out, ncores=2, output_all_reg=bool(bin_kb & 1)) == 0
    O.run_files(capi.XCK_MODE_BAF, d["bams"], feat, out_dir=ref, barcode_fn=d["barcodes"], snp_fn=d["snps_tsv"], output_all_reg=bool(bin_kb & 1))
    _cmp_dirs(out, ref)
