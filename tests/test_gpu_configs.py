"""Parity at the sizes BASELINE.json names ("configs run at <= 3 M reads inside -m gpu"):
  configs[1]  50 M reads / 5 k barcodes / 100 k SNPs: the WHOLE count and AD / DP / OTH matrices against the oracle (xo_run_mt);
  configs[2]  500 M reads / 10 k barcodes / 1 M SNPs: every row of the smallest contigs (~40 M reads) against the oracle, and
              size-independent properties of the full result (sortedness, AD <= DP, totals bounded by the accepted pairs);
  configs[4]  shape: 384 per-cell BAMs, no CB / UB tags, read names as keys, through the drop-in front-ends against the oracle's
              independent BAM reader.
Reads are synthetic (xcltk_amd/synth), generated on the device for the two large cases."""
import os
import sys

import numpy as np
import pytest

import oracle as O
import util
from xcltk_amd import capi
from xcltk_amd.engine import Engine

pytestmark = pytest.mark.gpu
FILT = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True)


def _device_workload(n_reads, n_cells, n_snps):
    import torch
    from xcltk_amd.synth import soa, soa_torch
    regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
    arrays, batches = soa_torch.gen_reads_device(regions, names, n_reads, n_cells, seed=100, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    return regions, snps, names, arrays, batches


def _run_engine(mode, names, regions, snps, n_cells, arrays, batches):
    from xcltk_amd.synth import soa_torch
    kw = dict(min_include=0.9) if mode == capi.XCK_MODE_BASEFC else dict(min_count=1, min_maf=0, no_dup_hap=True)
    eng = Engine(mode, names, regions, n_cells, snps=snps if mode == capi.XCK_MODE_BAF else (), device=0, **kw, **FILT)
    try:
        bs = [soa_torch.device_batch(capi, arrays, c, s, e, mode == capi.XCK_MODE_BAF) for c, s, e in batches]
        for b in bs:
            eng.push(b, device_resident=True)
        return eng.finish(), eng.stats()
    finally:
        eng.close()


def _oracle(mode, names, regions, snps, n_cells, arrays, take, threads=16):
    from xcltk_amd.synth import soa_torch
    hb = [util.batch_from_dict(soa_torch.host_batch_dict(arrays, c, s, e, True)) for c, s, e in take]
    kw = dict(min_include=0.9) if mode == capi.XCK_MODE_BASEFC else dict(min_count=1, min_maf=0, no_dup_hap=True)
    cfg, keep = O.make_config(mode, names, regions, snps if mode == capi.XCK_MODE_BAF else [], n_cells, **kw, **FILT)
    return O.run_oracle(cfg, [b for b, _ in hb], n_threads=threads)


def _check_properties(res, mats, stats):
    for m in mats:
        r, c, v = res[m]
        key = r.astype(np.int64) * (1 << 31) + c
        assert np.all(np.diff(key) > 0), m + ": not strictly sorted by (row, col)"      # sorted, no duplicate entries
        assert v.min() > 0
    if "ad" in res:
        ad = dict(zip(zip(res["ad"][0].tolist()[:200000], res["ad"][1].tolist()[:200000]), res["ad"][2].tolist()[:200000]))
        dp = dict(zip(zip(res["dp"][0].tolist(), res["dp"][1].tolist()), res["dp"][2].tolist())) if len(res["dp"][0]) < 30_000_000 else None
        if dp is not None:
            assert all(dp.get(k, 0) >= v for k, v in ad.items())                        # AD is part of DP


def test_config1_whole_matrices_vs_oracle():
    regions, snps, names, arrays, batches = _device_workload(50_000_000, 5000, 100_000)
    for mode, mats in ((capi.XCK_MODE_BASEFC, ["count"]), (capi.XCK_MODE_BAF, ["ad", "dp", "oth"])):
        got, st = _run_engine(mode, names, regions, snps, 5000, arrays, batches)
        exp = _oracle(mode, names, regions, snps, 5000, arrays, batches)
        util.assert_coo_equal(got, exp, mats)
        assert st["n_reads"] == arrays["n_reads"] and int(sum(got[m][2].sum() for m in mats[:1])) <= st["n_hits"]
        _check_properties(got, mats, st)


def test_config2_sampled_contigs_vs_oracle_and_properties():
    regions, snps, names, arrays, batches = _device_workload(500_000_000, 10000, 1_000_000)
    per = {}
    for c, s, e in batches:
        per.setdefault(c, []).append((c, s, e))
    take, tot = [], 0
    for c in sorted(per, key=lambda c: sum(e - s for _, s, e in per[c])):                # whole contigs, smallest first, ~40 M reads
        if tot >= 40_000_000:
            break
        take += per[c]; tot += sum(e - s for _, s, e in per[c])
    in_sample = np.array([r[0] in {names[c] for c, _, _ in take} for r in regions])
    for mode, mats in ((capi.XCK_MODE_BASEFC, ["count"]), (capi.XCK_MODE_BAF, ["ad", "dp", "oth"])):
        got, st = _run_engine(mode, names, regions, snps, 10000, arrays, batches)
        exp = _oracle(mode, names, regions, snps, 10000, arrays, take)
        n_cmp = 0
        for m in mats:
            sel = in_sample[got[m][0]]
            n_cmp += int(sel.sum())
            for j in range(3):
                assert np.array_equal(got[m][j][sel], exp[m][j]), "%s[%d] differs on the sampled contigs" % (m, j)
        assert n_cmp > 1_000_000
        assert st["n_reads"] == 500_000_000
        _check_properties(got, mats, st)


def test_config4_shape_384_per_cell_bams(tmp_path):
    """Well-based run: one BAM per cell, `--cellTAG None --UMItag None`, the sample list gives the columns."""
    from xcltk_amd.baf.fc.main import afc_wrapper
    from xcltk_amd.rdr.fc.main import fc_wrapper
    from xcltk_amd.synth.generate import make_smartseq_dataset
    d = str(tmp_path / "ss")
    p = make_smartseq_dataset(d, n_cells=384, reads_per_cell=400, n_snps=600, n_genes=60, contigs=(("1", 500000), ("2", 300000)), seed=15)
    lst = os.path.join(d, "bam_list.txt")
    with open(lst, "w") as fp:
        fp.write("".join(b + "\n" for b in p["bams"]))
    ids = os.path.join(d, "sample_ids.txt")
    common = dict(sam_fn=None, sam_list_fn=lst, barcode_fn=None, sample_id_fn=ids, region_fn=os.path.join(d, "regions.tsv"),
                  cell_tag="None", umi_tag="None", ncores=4)
    assert fc_wrapper(out_dir=str(tmp_path / "fc"), **common) == 0
    O.run_files(capi.XCK_MODE_BASEFC, p["bams"], os.path.join(d, "regions.tsv"), out_dir=str(tmp_path / "fc_o"), sample_ids=p["sample_ids"],
                cell_tag=None, umi_tag=None, output_all_reg=True, min_include=0.9)
    util.assert_dirs_equal(str(tmp_path / "fc"), str(tmp_path / "fc_o"))
    assert afc_wrapper(phased_snp_fn=os.path.join(d, "snps.tsv"), out_dir=str(tmp_path / "baf"), output_all_reg=True, **common) == 0
    O.run_files(capi.XCK_MODE_BAF, p["bams"], os.path.join(d, "regions.tsv"), out_dir=str(tmp_path / "baf_o"), sample_ids=p["sample_ids"],
                snp_fn=os.path.join(d, "snps.tsv"), cell_tag=None, umi_tag=None, output_all_reg=True, min_count=1, min_maf=0, no_dup_hap=True)
    util.assert_dirs_equal(str(tmp_path / "baf"), str(tmp_path / "baf_o"))
    hdr = open(str(tmp_path / "fc" / "matrix.mtx")).read().split("\n")[2].split("\t")
    assert hdr[1] == "384" and int(hdr[2]) > 5000


def test_hot_gene_cell_run_folds_without_run_walks():
    """One (gene, cell) pair holding ~10^6 (UMI, SNP) entries - a SMART-seq cell's most expressed gene: the haplotype fold must
    not depend on a thread walking the run (k_hap_sum: block scans + cross-tile atomics)."""
    from test_gpu_parity import _dense_pileup_case
    regions, snps, names, batches = _dense_pileup_case(seed=31, n_reads=120000, n_cells=2, n_umis=100000, snp_step=7, span=60000, max_batch=40000, gap_max=400)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 2, batches, min_len=10)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    assert st["n_hits"] > 1_000_000 and len(exp["dp"][0]) <= 6           # a million (read, SNP) pairs folded into at most 3 regions x 2 cells


@pytest.mark.parametrize("flags", [0, capi.XCK_F_FORCE_KEY128])
def test_deep_molecule_runs_are_not_walked_by_one_lane(flags):
    """Three UMIs in two cells over a dense SNP panel: every (SNP, cell, UMI) key holds hundreds to thousands of reads, far
    beyond what a run head walks itself (RUN_WALK = 64) - k_first_long finishes those runs, one block each.  64-bit keys take
    the split path (k_first_base), 128-bit keys the single sorted stream (k_first_read)."""
    from test_gpu_parity import _dense_pileup_case
    regions, snps, names, batches = _dense_pileup_case(seed=47, n_reads=60000, n_cells=2, n_umis=3, snp_step=11, span=40000, max_batch=20000, gap_max=300)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 2, batches, flags=flags, min_len=10)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    assert st["n_hits"] > 200_000


def test_resident_generator_draws_the_committed_workload():
    """The HBM-resident generator (xcltk_amd/synth/soa_torch.py) is reproducible: two draws in this process are identical array by
    array, and equal to the checksums committed in tests/golden/gen_check.json (drawn on another box by tools/gen_check.py) - so the
    `device_resident` numbers of bench.py mean the same workload wherever they are measured (they did not)."""
    import json
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(util.GOLDEN), os.pardir, "tools"))
    import gen_check
    from xcltk_amd.synth import soa, soa_torch
    want = json.load(open(os.path.join(util.GOLDEN, "gen_check.json")))
    regions, snps, names = soa.make_tables(want["genes"], 100000, soa.HG38_LENGTHS, seed=2)
    got = []
    for rep in range(2):
        arrays, batches = soa_torch.gen_reads_device(regions, names, want["reads"], want["cells"], seed=want["seed"], device=torch.device("cuda", 0))
        got.append(gen_check.checksums(arrays))
        del arrays
    assert got[0] == got[1]
    assert got[0] == want["checksums"]
