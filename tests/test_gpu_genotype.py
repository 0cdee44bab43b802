"""f1 on the GPU: baf.genotype.pileup() (step 1 of `xcltk baf`, the engine instead of the cellsnp-lite binary) against the
read-by-read counted pileup directory of the `phasing` golden dataset, and the chain pileup -> local phasing -> counting."""
import os

import numpy as np
import pytest

import util
from xcltk_amd.utils import csp_io

pytestmark = pytest.mark.gpu
DS = os.path.join(util.GOLDEN, "datasets", "phasing")


def test_engine_pileup_equals_counted_known_answer(tmp_path):
    from xcltk_amd.baf.genotype import pileup
    out = str(tmp_path / "pileup")
    vcf, p_raw, p_new = pileup(sam_fn=os.path.join(DS, "possorted.bam"), barcode_fn=os.path.join(DS, "barcodes.tsv"),
                               snp_vcf_fn=os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"), out_dir=out, mode="droplet", ncores=2,
                               min_count=20, min_maf=0.1)
    want = csp_io.load_data(os.path.join(DS, "cellsnp"))
    got = csp_io.load_data(os.path.join(out, "raw"))
    covered = np.asarray((want.DP + want.OTH).sum(axis=0)).reshape(-1) > 0     # raw/ lists the SNPs with at least one counted UMI
    assert p_raw == int(covered.sum()) and np.array_equal(got.pos, want.pos[covered]) and got.cells == want.cells
    for a, b in ((got.AD, want.AD), (got.DP, want.DP), (got.OTH, want.OTH)):
        assert (a != b[:, np.flatnonzero(covered)]).nnz == 0
    assert 0 < p_new < p_raw and vcf == os.path.join(out, "cellSNP.base.vcf.gz")
    flt = csp_io.load_data(out)
    assert flt.shape[1] == p_new


def test_pipeline_steps_1_and_3_with_engine_pileup(tmp_path):
    """`xcltk baf` with a given phased SNP list: the engine's own pileup directory (step 1) feeds the local phasing of step 3,
    like cellsnp-lite's does in the reference (baf/pipeline.py:341-360).  Checked against the oracle driven with the same
    pileup directory and the pipeline's fixed step-3 arguments."""
    import oracle as O
    from xcltk_amd import capi
    from xcltk_amd.baf.pipeline import pipeline_wrapper
    want = csp_io.load_data(os.path.join(DS, "cellsnp"))
    covered = set(want.pos[np.asarray((want.DP + want.OTH).sum(axis=0)).reshape(-1) > 0].tolist())
    lines = open(os.path.join(DS, "snps.tsv")).read().splitlines()
    snp_fn = str(tmp_path / "phased.tsv")                       # the phased list derives from the pileup VCF: covered SNPs only
    open(snp_fn, "w").write("\n".join([lines[0]] + [l for l in lines[1:] if int(l.split("\t")[1]) in covered]) + "\n")
    out = str(tmp_path / "pipe")
    ret = pipeline_wrapper("smp", sam_fn=os.path.join(DS, "possorted.bam"), barcode_fn=os.path.join(DS, "barcodes.tsv"),
                           snp_vcf_fn=os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"), region_fn=os.path.join(DS, "regions.tsv"),
                           out_dir=out, phased_snp_fn=snp_fn, ref_cell_fn=os.path.join(DS, "ref_cells.tsv"), min_count=1, min_maf=0, ncores=2)
    assert ret == 0
    pdir = os.path.join(out, "1_pileup")
    exp = str(tmp_path / "oracle")
    O.run_files(capi.XCK_MODE_BAF, [os.path.join(DS, "possorted.bam")], os.path.join(DS, "regions.tsv"), out_dir=exp,
                barcode_fn=os.path.join(DS, "barcodes.tsv"), snp_fn=snp_fn, output_all_reg=True, min_count=1, min_maf=0, no_dup_hap=True,
                phase=util.phase_from_cellsnp(pdir, os.path.join(DS, "ref_cells.tsv"), True))
    util.assert_dirs_equal(os.path.join(out, "3_baf_fc"), exp)


@pytest.mark.parametrize("name,case", __import__("test_genotype")._genotype_cases())
def test_engine_pileup_directories_equal_the_reference_filtered_fixtures(name, case, tmp_path):
    """pileup() through the engine: raw/ equals the fixture input (oracle counts through the same writer) and the filtered directory
    equals what the REFERENCE's filter_snps made of that input (tests/golden/genotype, oracle/refgen/make_genotype_goldens.py).
    The counts themselves stay unpinned against cellsnp-lite (binary absent)."""
    from test_genotype import GT, assert_cellsnp_dirs_equal
    from xcltk_amd.baf.genotype import pileup
    out = str(tmp_path / "pileup")
    vcf, p_raw, p_new = pileup(sam_fn=os.path.join(DS, "possorted.bam"), barcode_fn=os.path.join(DS, "barcodes.tsv"),
                               snp_vcf_fn=os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"), out_dir=out, mode="droplet", ncores=2,
                               min_count=case["min_count"], min_maf=case["min_maf"])
    assert (p_raw, p_new) == (case["p_raw"], case["p_new"])
    assert_cellsnp_dirs_equal(os.path.join(out, "raw"), os.path.join(GT, "raw"))
    assert_cellsnp_dirs_equal(out, os.path.join(GT, name))
