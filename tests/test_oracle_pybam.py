"""The pysam/htslib boundary on a REAL, htslib-written BAM (container-only: the file lives in the read-only reference
tree, /root/reference/preprocess/deprecated/merge_smartseq/BCH869.output.bam, `samtools merge` of 492 SMART-seq cells,
and never travels to the GPU box).

Every other BAM in the test suite was produced by this repo's own writers; here the two independent readers of the repo -
oracle/pybam.py (pure Python, zlib) and csrc/bam.cpp (own inflate, speculative record walk; decoder-only handle, no GPU) -
meet a third party's file and its samtools-made .bai:
  * the oracle reproduces the md5 that SURVEY.md section 8c recorded from the UNMODIFIED reference run on this BAM
    (hg19 gene list, 492 read-group ids as barcodes, --cellTAG RG --UMItag None);
  * the C++ decoder delivers the same records as pybam, field by field;
  * the .bai gives the per-reference record counts and byte ranges the contig sharding uses.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle as O
import pybam
from xcltk_amd import capi
from xcltk_amd.engine import Engine

D = "/root/reference/preprocess/deprecated/merge_smartseq/"
BAM = D + "BCH869.output.bam"
BARCODES = D + "BCH869.output.492.RG.barcodes.tsv"
GENES = "/root/reference/data/anno/annotate_genes_hg19_update_20230126.txt"
SURVEY_MD5 = "308c881da3d4ae0434c576258b78c70f"          # reference fc_wrapper on this input

pytestmark = [pytest.mark.container,
              pytest.mark.skipif(not os.path.isfile(BAM), reason="needs /root/reference (build container only)")]


def test_real_bam_reference_md5_through_oracle(tmp_path):
    out = str(tmp_path / "out")
    O.run_files(capi.XCK_MODE_BASEFC, [BAM], GENES, out_dir=out, barcode_fn=BARCODES, cell_tag="RG", umi_tag=None)
    m = open(os.path.join(out, "matrix.mtx"), "rb").read()
    assert m.split(b"\n")[2] == b"32696\t492\t8945"
    assert hashlib.md5(m).hexdigest() == SURVEY_MD5


def _decoder(regions, names, barcodes, **kw):
    return Engine(capi.XCK_MODE_BAF, names, regions, len(barcodes), snps=[(names[0], 1, "A", "C", 0, 1)], barcodes=barcodes,
                  cell_tag="RG", umi_tag=None, decode_only=True, n_threads=3, **kw)


def test_real_bam_decoder_vs_pybam():
    refs, recs = pybam.read_bam(BAM)
    names = [O.format_chrom(n) for n, _ in refs]
    names = list(dict.fromkeys(names))
    regions = [(n, 1, 1000, "r%d" % i) for i, n in enumerate(names)]
    with open(BARCODES) as fp:
        barcodes = sorted(x.strip() for x in fp)
    cell_of = {b: i for i, b in enumerate(barcodes)}
    eng = _decoder(regions, names, barcodes)
    try:
        got = list(eng.decode_bam(BAM))
    finally:
        eng.close()
    t2c = O.resolve_contigs([n for n, _ in refs], names)
    want = [r for r in recs if 0 <= r.tid < len(t2c) and t2c[r.tid] >= 0]
    assert sum(b["n_reads"] for b in got) == len(want) > 10000
    i = 0
    name_code = {}
    for b in got:
        n = b["n_reads"]
        assert (b["ordinal_base"] >> 40) == 0
        for k in range(n):
            r = want[i]
            assert b["contig"] == t2c[r.tid]
            assert b["pos"][k] == r.pos and b["flag"][k] == r.flag and b["mapq"][k] == r.mapq
            cig = b["cigar"][b["cig_off"][k]:b["cig_off"][k + 1]].tolist()
            assert cig == [(l << 4) | op for op, l in (r.cigartuples or [])]
            seq = bytes(b["seq"][b["seq_off"][k]:b["seq_off"][k + 1]])
            assert seq == bytes(r.seq_nibbles)
            rg = r.get_tag("RG") if r.has_tag("RG") else None
            assert b["cell"][k] == (cell_of.get(rg, -1) if isinstance(rg, str) else -1)
            # UMI-less mode: the key is the read name - equal names <=> equal codes (mates share a name)
            code = int(b["umi"][k])
            assert name_code.setdefault(r.query_name, code) == code
            i += 1
    assert len(set(name_code.values())) == len(name_code)


def test_real_bai_counts_and_ranges():
    refs, recs = pybam.read_bam(BAM)
    names = list(dict.fromkeys(O.format_chrom(n) for n, _ in refs))
    regions = [(n, 1, 1000, "r%d" % i) for i, n in enumerate(names)]
    with open(BARCODES) as fp:
        barcodes = sorted(x.strip() for x in fp)
    eng = _decoder(regions, names, barcodes)
    try:
        counts = eng.contig_record_counts(BAM)
        t2c = O.resolve_contigs([n for n, _ in refs], names)
        want = np.zeros(len(names), dtype=np.int64)
        for r in recs:
            if 0 <= r.tid < len(t2c) and t2c[r.tid] >= 0:
                want[t2c[r.tid]] += 1
        assert counts is not None and np.array_equal(counts, want)
        # decode through the samtools index: only the byte ranges of three references, same records as the filtered full decode
        order = np.argsort(-want)
        keep = np.zeros(len(names), dtype=bool)
        keep[order[[0, 3, 7]]] = True
        part = list(eng.decode_bam(BAM, contig_mask=keep, use_index=True))
        full = [b for b in eng.decode_bam(BAM) if keep[b["contig"]]]
        assert sum(b["n_reads"] for b in part) == int(want[keep].sum())
        cat = lambda bs, k: np.concatenate([b[k] for b in bs])
        for k in ("pos", "flag", "mapq", "cell"):
            assert np.array_equal(cat(part, k), cat(full, k))
    finally:
        eng.close()
