"""f1 / f4 host logic without a GPU: the cellsnp-lite style pileup is the allele-specific counting with one feature per
SNP (REF on haplotype 0, ALT on 1).  The `phasing` golden dataset carries a pileup directory that its generator counted
read by read (every read has its own UMI), i.e. a known answer that is independent of engine and oracle."""
import gzip
import os

import numpy as np
import pytest
from scipy import io as spio

import oracle as O
import util
from xcltk_amd import capi
from xcltk_amd.baf import genotype as G
from xcltk_amd.utils import csp_io
from xcltk_amd.utils.zfile import ZF_F_BGZIP, BGZFile, zopen

DS = os.path.join(util.GOLDEN, "datasets", "phasing")


def _known_answer():
    d = csp_io.load_data(os.path.join(DS, "cellsnp"))
    out = {}
    for k, m in (("ad", d.AD), ("dp", d.DP), ("oth", d.OTH)):
        coo = m.T.tocsr().tocoo()
        out[k] = (coo.row.astype(np.int32), coo.col.astype(np.int32), coo.data.astype(np.int32))
    return d, out


def snp_feature_tables(cand):
    regions = [(G.format_chrom(c), p, p, "%s_%d" % (c, p)) for c, p, _, _ in cand]
    snps = [(G.format_chrom(c), p, r, a, 0, 1) for c, p, r, a in cand]
    return regions, snps


def test_pileup_construction_equals_counted_known_answer(oracle_lib, tmp_path):
    d, want = _known_answer()
    cand = G.load_candidate_snps(os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"))
    assert len(cand) == d.shape[1] > 50
    regions, snps = snp_feature_tables(cand)
    rfn, sfn = str(tmp_path / "r.tsv"), str(tmp_path / "s.tsv")
    open(rfn, "w").write("".join("%s\t%d\t%d\t%s\n" % r for r in regions))
    open(sfn, "w").write("chrom\tpos\tref\talt\tref_hap\talt_hap\n" + "".join("%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
    coo = O.run_files(capi.XCK_MODE_BAF, [os.path.join(DS, "possorted.bam")], rfn, barcode_fn=os.path.join(DS, "barcodes.tsv"), snp_fn=sfn,
                      output_all_reg=True, min_count=1, min_maf=0, no_dup_hap=True)
    for k in ("ad", "dp", "oth"):
        for j in range(3):
            assert np.array_equal(coo[k][j], want[k][j]), (k, j)


def test_cellsnp_dir_roundtrip_and_second_filter(tmp_path):
    d, want = _known_answer()
    snps = [(str(c), int(p), str(r), str(a)) for c, p, r, a in zip(d.chrom, d.pos, d.ref, d.alt)]
    raw = str(tmp_path / "raw")
    G.write_cellsnp_dir(raw, snps, d.cells, {"AD": want["ad"], "DP": want["dp"], "OTH": want["oth"]})
    back = csp_io.load_data(raw)
    assert back.cells == d.cells and np.array_equal(back.pos, d.pos)
    for a, b in ((back.AD, d.AD), (back.DP, d.DP), (back.OTH, d.OTH)):
        assert (a != b).nnz == 0
    # the files are what scipy / the reference's loaders expect (csp_io.py:93-100): SNP x cell, integer
    m = spio.mmread(os.path.join(raw, "cellSNP.tag.DP.mtx"))
    assert m.shape == (d.shape[1], d.shape[0])
    # filter_snps (baf/genotype.py:190-229): DP >= min_count and min_maf <= AD / DP <= 1 - min_maf on REF + ALT counts
    fn, p_raw, p_new = G.filter_snps(raw, str(tmp_path / "flt"), min_count=20, min_maf=0.1)
    AD = np.asarray(d.AD.sum(axis=0)).reshape(-1); DP = np.asarray(d.DP.sum(axis=0)).reshape(-1)
    with np.errstate(all="ignore"):
        keep = (DP >= 20) & (AD / DP >= 0.1) & (AD / DP <= 0.9)
    assert p_raw == d.shape[1] and p_new == int(keep.sum()) and 0 < p_new < p_raw
    flt = csp_io.load_data(str(tmp_path / "flt"))
    assert np.array_equal(flt.pos, d.pos[keep])
    info = [l.rstrip("\n").split("\t")[7] for l in gzip.open(fn, "rt") if not l.startswith("#")]
    assert info == ["AD=%d;DP=%d;OTH=%d" % (a, b, c) for a, b, c in zip(AD[keep], DP[keep], np.asarray(d.OTH.sum(axis=0)).reshape(-1)[keep])]


def test_bgzf_writer_blocks_and_eof(tmp_path):
    fn = str(tmp_path / "t.vcf.gz")
    rng = np.random.default_rng(1)
    text = "".join("chr1\t%d\t.\tA\tC\t.\tPASS\tAD=%d\n" % (i, int(rng.integers(0, 99))) for i in range(40000)).encode()
    noise = bytes(rng.integers(0, 256, 200000, dtype=np.uint8))               # incompressible: stored blocks
    with zopen(fn, "wb", ZF_F_BGZIP, is_bytes=True) as fp:
        fp.write(text)
        fp.write(noise)
    raw = open(fn, "rb").read()
    assert raw.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    off, n_blocks = 0, 0
    while off < len(raw):                                                     # every block: gzip member with the BC subfield, <= 64 KiB
        assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 14] == b"BC"
        bsize = int.from_bytes(raw[off + 16:off + 18], "little") + 1
        assert bsize <= 0x10000
        off += bsize; n_blocks += 1
    assert off == len(raw) and n_blocks > 10
    assert gzip.open(fn, "rb").read() == text + noise
    with pytest.raises(ValueError):
        BGZFile(fn, "r")


def test_output_directory_adapters_read_the_reference_files():
    """f4: baf/io.py and rdr/io.py read what the reference wrote (golden expected dirs) - without anndata."""
    from xcltk_amd.baf import io as bio
    from xcltk_amd.rdr import io as rio
    cdir = os.path.join(util.GOLDEN, "cases")
    f, c, m = rio.load_matrix_data(os.path.join(cdir, "c1_basefc_default", "expected"))
    assert list(f.columns) == ["chrom", "start", "end", "feature"] and m.shape == (len(c), len(f)) == (1000, 200)
    hdr = open(os.path.join(cdir, "c1_basefc_default", "expected", "matrix.mtx")).read().split("\n")[2].split("\t")
    assert m.nnz == int(hdr[2]) and f["chrom"].dtype == object
    f, c, mats = bio.load_matrix_data(os.path.join(cdir, "c1_baf_allreg", "expected"))
    assert set(mats) == {"AD", "DP", "OTH"} and all(x.shape == (1000, 200) for x in mats.values())
    assert (mats["AD"] > mats["DP"]).nnz == 0                              # AD is part of DP


@pytest.mark.parametrize("exc", [MemoryError, IndexError, OSError])
def test_pileup_writer_failure_reaches_the_status_collective(exc, monkeypatch, tmp_path):
    """Multi-rank pileup(): whatever the writer rank's failure is (MemoryError while the .mtx text is built, an index error, a full
    disk), the rank still takes part in the status all-reduce the other ranks are waiting in, and only then raises."""
    calls = []

    class FakeDist(object):
        active = True

        def all_reduce_np(self, x, op="sum"):
            calls.append((np.asarray(x).tolist(), op))
            return np.asarray(x)

    class FakeEngine(object):
        closed = False

        def close(self):
            self.closed = True
    eng = FakeEngine()
    z = np.zeros(0, dtype=np.int32)
    monkeypatch.setattr(G.fcc, "make_and_count", lambda *a, **k: (eng, {m: (z, z, z) for m in ("ad", "dp", "oth")}, FakeDist()))

    def boom(*a, **k):
        raise exc("writer failed")
    monkeypatch.setattr(G, "_write_pileup_dirs", boom)
    with pytest.raises(exc):
        G.pileup(sam_fn=os.path.join(DS, "possorted.bam"), barcode_fn=os.path.join(DS, "barcodes.tsv"),
                 snp_vcf_fn=os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"), out_dir=str(tmp_path / "p"))
    assert calls == [([1], "max")] and eng.closed


# ---- f1 against the reference's own code (oracle/refgen/make_genotype_goldens.py; fixtures under tests/golden/genotype) -------------
GT = os.path.join(util.GOLDEN, "genotype")


def assert_cellsnp_dirs_equal(got, exp):
    """Same directory content: VCF text after decompression, cell list bytes, the three matrices entry by entry."""
    assert gzip.open(os.path.join(got, "cellSNP.base.vcf.gz"), "rt").read() == gzip.open(os.path.join(exp, "cellSNP.base.vcf.gz"), "rt").read()
    assert open(os.path.join(got, "cellSNP.samples.tsv"), "rb").read() == open(os.path.join(exp, "cellSNP.samples.tsv"), "rb").read()
    for k in ("AD", "DP", "OTH"):
        a, b = (spio.mmread(os.path.join(d, "cellSNP.tag.%s.mtx" % k)).tocsr() for d in (got, exp))
        assert a.shape == b.shape and (a != b).nnz == 0, k


def test_raw_fixture_is_the_oracle_pileup_through_our_writer(oracle_lib, tmp_path):
    """The fixture INPUT: the oracle's per-SNP x cell counts of the `phasing` dataset, written by write_cellsnp_dir."""
    cand = G.load_candidate_snps(os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"))
    regions, snps = snp_feature_tables(cand)
    rfn, sfn = str(tmp_path / "r.tsv"), str(tmp_path / "s.tsv")
    open(rfn, "w").write("".join("%s\t%d\t%d\t%s\n" % r for r in regions))
    open(sfn, "w").write("chrom\tpos\tref\talt\tref_hap\talt_hap\n" + "".join("%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
    coo = O.run_files(capi.XCK_MODE_BAF, [os.path.join(DS, "possorted.bam")], rfn, barcode_fn=os.path.join(DS, "barcodes.tsv"), snp_fn=sfn,
                      output_all_reg=True, min_count=1, min_maf=0, no_dup_hap=True)
    cells = sorted(x.strip() for x in open(os.path.join(DS, "barcodes.tsv")) if x.strip())
    G._write_raw_dir(str(tmp_path / "raw"), cand, cells, coo)
    assert_cellsnp_dirs_equal(str(tmp_path / "raw"), os.path.join(GT, "raw"))


def test_reference_loader_reads_our_directory_as_we_do():
    """utils/csp_io.load_data of the REFERENCE on a directory of our writer (recorded by the generator) == our own loader."""
    import json
    seen = json.load(open(os.path.join(GT, "ref_load.json")))
    d = csp_io.load_data(os.path.join(GT, "raw"))
    assert seen["n_cells"] == d.shape[0] and seen["n_snps"] == d.shape[1] and seen["cells"] == d.cells
    assert seen["pos"] == d.pos.tolist() and seen["chrom"] == [str(c) for c in d.chrom] and seen["ref"] == list(d.ref) and seen["alt"] == list(d.alt)
    for k, m in (("AD", d.AD), ("DP", d.DP), ("OTH", d.OTH)):
        assert seen["colsum_" + k] == np.asarray(m.sum(axis=0)).reshape(-1).astype(int).tolist() and seen["sum_" + k] == int(m.sum())


def _genotype_cases():
    import json
    return sorted(json.load(open(os.path.join(GT, "cases.json")))["cases"].items())


@pytest.mark.parametrize("name,case", _genotype_cases())
def test_filter_snps_equals_the_reference_filter(name, case, tmp_path):
    """filter_snps against the output of the reference's filter_snps (baf/genotype.py:200-229) on the same raw directory."""
    fn, p_raw, p_new = G.filter_snps(os.path.join(GT, "raw"), str(tmp_path / "flt"), case["min_count"], case["min_maf"])
    assert (p_raw, p_new) == (case["p_raw"], case["p_new"]) and fn == str(tmp_path / "flt" / "cellSNP.base.vcf.gz")
    assert_cellsnp_dirs_equal(str(tmp_path / "flt"), os.path.join(GT, name))
