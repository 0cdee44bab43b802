"""Known-answer tests that were executed through the reference's own fc_fet1() during the
survey (KAT-BAF-1 and KAT-RDR-1); inputs are fully specified there."""
import os

import numpy as np
import pytest

import oracle as O
from xcltk_amd import capi
from xcltk_amd.synth.bamwriter import BamWriter


def _write(tmp, reads, cells, regions, snps=None):
    d = str(tmp)
    bw = BamWriter(os.path.join(d, "kat.bam"), [("1", 100000)])
    for i, (pos, cig, seq, cb, ub) in enumerate(sorted(reads, key=lambda r: r[0])):
        tags = [("CB", cb), ("UB", ub)]
        bw.write(0, pos, "r%d" % i, 0, 255, cig, seq, tags)
    bw.close()
    with open(os.path.join(d, "barcodes.tsv"), "w") as fp:
        fp.write("".join(c + "\n" for c in cells))
    with open(os.path.join(d, "regions.tsv"), "w") as fp:
        fp.write("".join("%s\t%d\t%d\t%s\n" % r for r in regions))
    if snps:
        with open(os.path.join(d, "snps.tsv"), "w") as fp:
            fp.write("chrom\tpos\tref\talt\tref_hap\talt_hap\n")
            fp.write("".join("%s\t%d\t%s\t%s\t%d\t%d\n" % s for s in snps))
    return d


KAT_BAF_READS = [
    (99, "30M40N30M", "C" * 60, "c1", "u1"),
    (120, "60M", "G" * 60, "c1", "u1"),
    (140, "70M", "A" * 59 + "C" + "A" * 10, "c1", "u2"),
    (140, "40M", "T" * 9 + "G" + "T" * 30, "c1", "u3"),
    (140, "40M", "T" * 40, "c2", "u4"),
    (140, "40M", "G" * 40, "zz", "u5"),
]
KAT_BAF_SNPS = [("1", 150, "A", "G", 0, 1), ("1", 200, "C", "T", 1, 0)]


def dense(coo, n_cells):
    out = np.zeros(n_cells, dtype=int)
    for r, c, v in zip(*coo):
        out[c] += v
    return out.tolist()


@pytest.mark.parametrize("no_dup_hap,exp", [
    (True, dict(ad=[1, 0], dp=[1, 0], oth=[0, 1])),
    (False, dict(ad=[2, 0], dp=[3, 0], oth=[0, 1])),
])
def test_kat_baf_1(tmp_path, oracle_lib, no_dup_hap, exp):
    d = _write(tmp_path, KAT_BAF_READS, ["c1", "c2"], [("1", 100, 300, "g")], KAT_BAF_SNPS)
    coo = O.run_files(capi.XCK_MODE_BAF, [d + "/kat.bam"], d + "/regions.tsv", barcode_fn=d + "/barcodes.tsv",
                      snp_fn=d + "/snps.tsv", no_dup_hap=no_dup_hap)
    for k, v in exp.items():
        assert dense(coo[k], 2) == v, k


@pytest.mark.parametrize("min_include,exp", [(0.9, [1, 1, 0, 1, 0, 0]), (30, [1, 1, 1, 0, 0, 1]), (0, [1] * 6)])
def test_kat_rdr_1(tmp_path, oracle_lib, min_include, exp):
    specs = [(100, 100), (100, 90), (100, 89), (30, 97), (30, 96), (200, 50)]      # (nM, 0-based pos)
    got = []
    for i, (n, p) in enumerate(specs):
        sub = tmp_path / ("k%d" % i)
        sub.mkdir()
        d = _write(sub, [(p, "%dM" % n, "A" * n, "c1", "ACGTACGTAC")], ["c1"], [("1", 101, 200, "g")])
        coo = O.run_files(capi.XCK_MODE_BASEFC, [d + "/kat.bam"], d + "/regions.tsv", barcode_fn=d + "/barcodes.tsv",
                          min_include=min_include)
        got.append(int(coo["count"][2].sum()))
    assert got == exp


@pytest.mark.parametrize("threads", [2, 5])
def test_threaded_oracle_gives_the_same_matrices(oracle_lib, threads):
    """xo_run_mt (region chunks on host threads, the CPU baseline of bench.py) == xo_run, entry for entry."""
    import util
    from xcltk_amd.synth import soa
    regions, snps, names = soa.make_tables(300, 5000, soa.HG38_LENGTHS[:3], seed=5, max_len=150000)
    batches = [util.batch_from_dict(b) for b in soa.gen_reads(regions, names, 120000, 150, seed=6, max_batch=50000)]
    for mode, sn in ((capi.XCK_MODE_BASEFC, []), (capi.XCK_MODE_BAF, snps)):
        cfg, keep = O.make_config(mode, names, regions, sn, 150, min_count=2 if sn else 1)
        one = O.run_oracle(cfg, [b for b, _ in batches])
        many = O.run_oracle(cfg, [b for b, _ in batches], n_threads=threads)
        assert sum(len(v[0]) for v in one.values()) > 1000
        for k in one:
            assert all(np.array_equal(x, y) for x, y in zip(one[k], many[k])), k
