"""GPU parity: HIP engine (through the C-ABI) vs the CPU oracle on identical SoA batches."""
import numpy as np
import pytest

import oracle as O
import util
from xcltk_amd import capi
from xcltk_amd.synth import soa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    regions, snps, names = soa.make_tables(600, 20000, soa.HG38_LENGTHS[:2], seed=11, max_len=200000)
    bs = soa.gen_reads(regions, names, 300000, 200, seed=12, max_batch=70000)
    return regions, snps, names, [util.batch_from_dict(b) for b in bs]


@pytest.mark.parametrize("min_include", [0.9, 0.5, 30, 0, 1.0, 91])
def test_basefc_include_modes(small, min_include):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, snps, 200, batches,
                                         min_include=min_include)
    assert len(exp["count"][0]) > 1000
    util.assert_coo_equal(got, exp, ["count"])


@pytest.mark.parametrize("filt", [dict(), dict(min_mapq=0), dict(excl_flag=1796), dict(incl_flag=16),
                                  dict(min_len=60), dict(excl_flag=0, min_mapq=2)])
def test_basefc_filters(small, filt):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, snps, 200, batches, **filt)
    util.assert_coo_equal(got, exp, ["count"])


def test_basefc_key128(small):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, snps, 200, batches,
                                         flags=capi.XCK_F_FORCE_KEY128)
    assert st["key_bits"] == 128
    util.assert_coo_equal(got, exp, ["count"])


@pytest.mark.parametrize("opts", [dict(), dict(no_dup_hap=False), dict(min_count=3, min_maf=0.1),
                                  dict(min_count=11, min_maf=0.1), dict(min_mapq=0, excl_flag=0)])
def test_baf(small, opts):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 200, batches, **opts)
    if not opts.get("min_count", 0) > 5:
        assert len(exp["dp"][0]) > 500
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_baf_key128(small):
    regions, snps, names, batches = small
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 200, batches,
                                         flags=capi.XCK_F_FORCE_KEY128)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_frac_divide_matches_host():
    """m/float(n) < v on the GPU (IEEE double divide) vs the host for every m <= n <= 160 at
    awkward thresholds: single-base reads with crafted CIGARs."""
    lib = O.load_oracle()
    for v in (0.9, 0.1, 1.0 / 3.0, 0.7, 0.5, 0.30000000000000004):
        rows = []
        for n in range(1, 161):
            for m in range(0, n + 1):
                rows.append((m, n))
        # region [1001, 2000] 1-based; read with n aligned bases of which m inside: start at 1000 - (n - m)
        regions = [("1", 1001, 2000, "r")]
        names = ["1"]
        pos = np.array([1000 - (n - m) for m, n in rows], dtype=np.int32)
        order = np.argsort(pos, kind="stable")
        pos = pos[order]
        nn = np.array([rows[i][1] for i in order], dtype=np.uint32)
        mm = np.array([rows[i][0] for i in order], dtype=np.int32)
        keep = mm > 0                       # reads with m = 0 do not overlap the region at all
        pos, nn, mm = pos[keep], nn[keep], mm[keep]
        k = len(pos)
        b = capi.make_batch(0, 0, pos, np.zeros(k, np.uint16), np.full(k, 60, np.uint8), np.zeros(k, np.int32),
                            np.arange(k, dtype=np.uint64) + 1000, np.arange(k + 1, dtype=np.uint32),
                            (nn << 4).astype(np.uint32))
        got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, [b], min_include=v,
                                             min_len=0)
        want = sum(1 for m, n in zip(mm, nn) if not lib.xo_frac_drop(int(m), int(n), v))
        assert int(exp["count"][2].sum()) == want
        util.assert_coo_equal(got, exp, ["count"])


@pytest.fixture(scope="module")
def large():
    from xcltk_amd.synth import soa as _soa
    regions, snps, names = _soa.make_tables(4000, 60000, _soa.HG38_LENGTHS[:4], seed=21, max_len=300000)
    bs = _soa.gen_reads(regions, names, 3000000, 1500, seed=22, max_batch=700000)
    return regions, snps, names, [util.batch_from_dict(b) for b in bs]


@pytest.mark.parametrize("mode,mats", [(capi.XCK_MODE_BASEFC, ["count"]), (capi.XCK_MODE_BAF, ["ad", "dp", "oth"])])
def test_large_scale_parity(large, mode, mats):
    """3 M reads / thousands of tiles: catches races and tile-boundary bugs that tiny inputs cannot
    (every LDS flush / spill path and the sharded cursors are exercised)."""
    regions, snps, names, batches = large
    got, exp, st = util.engine_vs_oracle(mode, names, regions, snps, 1500, batches)
    assert st["n_hits"] > 100000
    util.assert_coo_equal(got, exp, mats)
    # determinism: a second run gives the identical result and the identical accepted-pair count
    got2, _, st2 = util.engine_vs_oracle(mode, names, regions, snps, 1500, batches)
    util.assert_coo_equal(got2, exp, mats)
    assert st2["n_hits"] == st["n_hits"]


def test_deep_nesting_spill_and_overflow_replay():
    """12 identical + 6 nested regions over every read: >12 accepted pairs per read saturates the LDS
    set (spill through emit_global) and exceeds the 5-hits-per-read capacity estimate (overflow flag,
    buffer growth, cursor rewind and replay).  Result must still equal the oracle."""
    rng = np.random.default_rng(5)
    regions = [("1", 1000, 900000, "big%d" % i) for i in range(12)] + [("1", 1000 + 50000 * i, 400000 + 50000 * i, "n%d" % i) for i in range(6)]
    names = ["1"]
    bs = soa.gen_reads(regions, names, 400000, 50, seed=9, max_batch=150000)
    batches = [util.batch_from_dict(b) for b in bs]
    for inc in (0, 0.9):
        got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 50, batches, min_include=inc)
        assert st["n_hits"] > 12 * 250000
        util.assert_coo_equal(got, exp, ["count"])


def test_many_batches_fused_queue_and_empty_batches(small):
    """More device-resident-style pushes than one fused launch holds (host path here: each push is its own
    launch), empty batches and a contig without regions in between."""
    regions, snps, names, batches = small
    tiny = []
    for b, keep in batches:
        n = b.n_reads
        pos, flag, mapq, cell, umi, cig_off, cigar = keep[:7]
        step = max(1, n // 30)
        for s0 in range(0, n, step):
            e0 = min(n, s0 + step)
            c_lo, c_hi = int(cig_off[s0]), int(cig_off[e0])
            args = [b.contig, int(b.ordinal_base) + s0, pos[s0:e0], flag[s0:e0], mapq[s0:e0], cell[s0:e0], umi[s0:e0],
                    (cig_off[s0:e0 + 1] - cig_off[s0]).astype(np.uint32), cigar[c_lo:c_hi] if c_hi > c_lo else np.zeros(1, np.uint32)]
            if len(keep) > 7:
                so, sq = keep[7], keep[8]
                q_lo, q_hi = int(so[s0]), int(so[e0])
                args += [(so[s0:e0 + 1] - so[s0]).astype(np.uint32), sq[q_lo:q_hi] if q_hi > q_lo else np.zeros(1, np.uint8)]
            tiny.append(capi.make_batch(*args))
        tiny.append(capi.make_batch(b.contig, 0, np.zeros(0, np.int32), [], [], [], [], np.zeros(1, np.uint32), np.zeros(1, np.uint32),
                                    np.zeros(1, np.uint32), np.zeros(1, np.uint8)))
    assert len(tiny) > 60
    for mode, mats in ((capi.XCK_MODE_BASEFC, ["count"]), (capi.XCK_MODE_BAF, ["ad", "dp", "oth"])):
        got, exp, st = util.engine_vs_oracle(mode, names, regions, snps, 200, tiny)
        got0, _, _ = util.engine_vs_oracle(mode, names, regions, snps, 200, batches)
        util.assert_coo_equal(got, exp, mats)
        util.assert_coo_equal(got0, exp, mats)


def test_device_resident_batches_more_than_one_fused_launch():
    """xck_push_batch_device: 70 HBM-resident batches (slices of contigs) need three fused launches of <= 24
    batches; both modes must equal the oracle run on host copies of the same arrays."""
    import torch
    from xcltk_amd.engine import Engine
    from xcltk_amd.synth import soa_torch
    regions, snps, names = soa.make_tables(1500, 30000, soa.HG38_LENGTHS[:5], seed=31, max_len=200000)
    arrays, contig_batches = soa_torch.gen_reads_device(regions, names, 1200000, 300, seed=32, device=torch.device("cuda", 0))
    pieces = []
    for c, s, e in contig_batches:
        step = max(1, (e - s) // 14)
        pieces += [(c, a, min(e, a + step)) for a in range(s, e, step)]
    assert len(pieces) > 48
    hb = [util.batch_from_dict(soa_torch.host_batch_dict(arrays, c, s, e, True)) for c, s, e in pieces]
    for mode, mats in ((capi.XCK_MODE_BASEFC, ["count"]), (capi.XCK_MODE_BAF, ["ad", "dp", "oth"])):
        eng = Engine(mode, names, regions, 300, snps=snps if mode == 2 else ())
        for rep in range(2):                                   # second pass re-uses the engine after reset()
            eng.reset()
            for c, s, e in pieces:
                eng.push(soa_torch.device_batch(capi, arrays, c, s, e, mode == 2), device_resident=True)
            got = eng.finish()
            st = eng.stats()
            assert st["n_join_launches"] >= 3
            cfg, keep = O.make_config(mode, names, regions, snps if mode == 2 else [], 300)
            exp = O.run_oracle(cfg, [b for b, _ in hb])
            util.assert_coo_equal(got, exp, mats)
        eng.close()


def _dense_pileup_case(seed, n_reads, n_cells, n_umis, snp_step=37, span=120000, max_batch=5000, gap_max=25000):
    """Hand-rolled reads over ONE region with a SNP every `snp_step` bp: long N gaps (hundreds of SNPs per gap: many
    32-SNP gap records), D gaps, insertions / clips, reads without sequence, unmapped-flag reads with a CIGAR, and few
    (cell, UMI) pairs, so that 'the first read of a (SNP, cell, UMI) wins - even without a base' decides most keys."""
    rng = np.random.default_rng(seed)
    names = ["1"]
    regions = [("1", 1, span, "g0"), ("1", 2000, 50000, "g1"), ("1", 60000, 61000, "g2")]
    snps = [("1", p, "ACGT"[(p // snp_step) % 4], "ACGT"[(p // snp_step + 1 + (p % 3)) % 4], p % 2, 1 - p % 2)
            for p in range(10, span - 10, snp_step)]
    M, I, D, N, S = 0, 1, 2, 3, 4
    recs = []
    for _ in range(n_reads):
        pos = int(rng.integers(0, span - 30000))
        kind = rng.integers(0, 10)
        if kind < 3:
            cig = [(M, 91)]
        elif kind < 6:
            a = int(rng.integers(10, 80)); cig = [(M, a), (N, int(rng.integers(50, gap_max))), (M, 91 - a)]
        elif kind == 6:
            a = int(rng.integers(10, 60)); cig = [(M, a), (D, int(rng.integers(1, 300))), (M, 91 - a)]
        elif kind == 7:
            a = int(rng.integers(10, 60)); cig = [(S, 5), (M, a), (I, 3), (M, 20), (N, int(rng.integers(100, 3000))), (M, 63 - a)]
        elif kind == 8:
            a = int(rng.integers(5, 40)); cig = [(M, a), (N, 2000), (M, 20), (N, 1500), (M, 71 - a)]
        else:
            cig = [(M, 40), (N, min(9000, gap_max)), (M, 51)]
        qlen = sum(l for op, l in cig if op in (M, I, S))
        noseq = rng.random() < 0.04
        flag = 0
        if rng.random() < 0.03:
            flag |= 4                                      # unmapped flag with a CIGAR: reference span collapses to 1
        if rng.random() < 0.3:
            flag |= 16
        nib = (1 << rng.integers(0, 4, qlen)).astype(np.uint8)
        nib[rng.random(qlen) < 0.01] = 15
        if qlen % 2:
            nib = np.append(nib, 0)
        seq = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8) if not noseq else np.zeros(0, np.uint8)
        cell = int(rng.integers(-1, n_cells))
        umi = np.uint64((1 << 24) | int(rng.integers(0, n_umis)))
        recs.append((pos, cig, seq, flag, cell, umi))
    recs.sort(key=lambda r: r[0])
    batches = []
    for s in range(0, len(recs), max_batch):
        part = recs[s:s + max_batch]
        cig_off = np.zeros(len(part) + 1, np.uint32); seq_off = np.zeros(len(part) + 1, np.uint32)
        cw, sq = [], []
        for j, (pos, cig, seq, flag, cell, umi) in enumerate(part):
            cw += [(l << 4) | op for op, l in cig]; sq.append(seq)
            cig_off[j + 1] = len(cw); seq_off[j + 1] = seq_off[j] + len(seq)
        d = dict(contig=0, ordinal_base=s, pos=np.array([r[0] for r in part], np.int32), flag=np.array([r[3] for r in part], np.uint16),
                 mapq=np.full(len(part), 60, np.uint8), cell=np.array([r[4] for r in part], np.int32),
                 umi=np.array([r[5] for r in part], np.uint64), cig_off=cig_off, cigar=np.array(cw, np.uint32),
                 seq_off=seq_off, seq=np.concatenate(sq) if sq else np.zeros(0, np.uint8))
        batches.append(util.batch_from_dict(d))
    return regions, snps, names, batches


@pytest.mark.parametrize("flags", [0, capi.XCK_F_FORCE_KEY128])
@pytest.mark.parametrize("opts", [dict(excl_flag=0), dict(), dict(no_dup_hap=False, min_count=2, min_maf=0.05)])
def test_pileup_gap_records_and_claims(flags, opts):
    """Split pileup path (gap records, Bloom filter, claims) against the oracle where gaps cover hundreds of SNPs and
    (cell, UMI) pairs are reused by many reads; the 128-bit path (single sorted stream) must agree too."""
    regions, snps, names, batches = _dense_pileup_case(seed=5, n_reads=30000, n_cells=6, n_umis=40)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 6, batches, flags=flags, min_len=10, **opts)
    assert len(exp["dp"][0]) > 5 and int(exp["dp"][2].sum()) > 20
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_pileup_gap_records_many_cells():
    regions, snps, names, batches = _dense_pileup_case(seed=9, n_reads=60000, n_cells=500, n_umis=3000, snp_step=101, max_batch=20000)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 500, batches, min_len=10)
    assert len(exp["dp"][0]) > 100
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_pileup_deep_snps_are_tallied_by_blocks():
    """All reads start inside 1 kb: every SNP there holds thousands of (cell, UMI) keys - more than the eight lanes of k_tally_rows
    take (TALLY_LONG = 2048), so k_tally_long counts them; the SNPs further out (reached through N gaps) stay on the short path.
    min_count / min_maf make the tallies decide which SNPs survive."""
    regions, snps, names, batches = _dense_pileup_case(seed=77, n_reads=120000, n_cells=3, n_umis=1 << 22, snp_step=53, span=31000, max_batch=40000, gap_max=9000)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 3, batches, min_len=10, min_count=3000, min_maf=0.2)
    got2, exp2, _ = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 3, batches, min_len=10, min_count=20, min_maf=0.0)
    assert int(exp["dp"][2].max()) > 2048 and len(exp2["dp"][0]) > len(exp["dp"][0]) > 0
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    util.assert_coo_equal(got2, exp2, ["ad", "dp", "oth"])


def test_pileup_overflow_replay_both_streams(monkeypatch):
    """A SNP every 3 bp: ~30 hits with a base and tens of gap records per read overrun the first capacity guess
    (1.25 keys per read) of BOTH pileup streams: overflow flag, growth of all four buffers, cursor rewind, replay."""
    monkeypatch.setenv("XCK_HIT_CAP0", "1024"); monkeypatch.setenv("XCK_HIT_SLACK", "0")    # read by the library at xck_create / first use
    regions, snps, names, batches = _dense_pileup_case(seed=21, n_reads=200000, n_cells=50, n_umis=5000, snp_step=3,
                                                       span=60000, max_batch=200000, gap_max=900)
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, regions, snps, 50, batches, min_len=10)
    assert st["n_hits_unique"] > 2 * 16 * (200000 * 5 // 4 // 16)          # more than the first guess can hold
    assert st["n_join_launches"] > len(batches)                          # at least one launch was replayed
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])


def test_degenerate_tables_and_lifecycle(small):
    """Empty region / SNP tables, finish() without any push, reset() and reuse of a handle, regions that pysam's fetch
    rejects (start < 1, start - 1 > end), SNPs outside every region, a contig without targets."""
    regions, snps, names, batches = small
    # no reads at all
    for mode, tabs in ((capi.XCK_MODE_BASEFC, ()), (capi.XCK_MODE_BAF, snps)):
        from xcltk_amd.engine import Engine
        eng = Engine(mode, names, regions, 200, snps=tabs)
        try:
            out = eng.finish()
            assert all(len(v[0]) == 0 for v in out.values())
            eng.reset()
            for b, _ in batches[:2]:
                eng.push(b)
            first = {k: [a.copy() for a in v] for k, v in eng.finish().items()}
            eng.reset()                                       # same handle, same input again: identical result
            for b, _ in batches[:2]:
                eng.push(b)
            again = eng.finish()
            for k in first:
                assert all(np.array_equal(x, y) for x, y in zip(first[k], again[k]))
        finally:
            eng.close()
    # odd tables: empty SNP list; regions outside fetch's domain; SNPs on a contig / at positions no region covers
    odd_regions = [("1", 0, 5000, "starts_at_0"), ("1", 9000, 100, "inverted"), ("2", 1, 10, "tiny")] + list(regions[:50])
    odd_snps = [("2", 5, "A", "C", 0, 1), ("1", 248000000, "G", "T", 1, 0)] + list(snps[:2000])
    odd_snps.sort(key=lambda s: (names.index(s[0]), s[1]))
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, odd_regions, [], 200, batches)
    util.assert_coo_equal(got, exp, ["count"])
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, odd_regions, odd_snps, 200, batches)
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BAF, names, odd_regions, [], 200, batches)
    assert all(len(got[k][0]) == 0 for k in ("ad", "dp", "oth"))
    util.assert_coo_equal(got, exp, ["ad", "dp", "oth"])
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, [], [], 200, batches)
    assert len(got["count"][0]) == 0


@pytest.mark.parametrize("n,hot,n_umis,label", [(150000, 0.7, 60000, "giant"), (20000, 0.2, 3000, "lead-in"), (60000, 0.1, 2500, "lead-in-dups")])
def test_basefc_long_runs_hash_fold_and_fallback(n, hot, n_umis, label):
    """Long (region, cell) runs for the hash fold (keys sorted by (row, cell) only): 'giant' = one pair with tens of
    thousands of distinct UMIs, far beyond the fold's lead-in window - finish() must notice (FOLD_GIANT) and redo the fold
    on fully sorted keys; 'lead-in*' = runs of a few thousand keys that cross tile borders inside the window, with the
    same UMI on both sides of a border.  A second region / other cells keep ordinary runs around them."""
    rng = np.random.default_rng(3)
    names = ["1"]
    regions = [("1", 1, 200000, "hot"), ("1", 50000, 60000, "inner")]
    pos = np.sort(rng.integers(0, 190000, n)).astype(np.int32)
    cell = np.where(rng.random(n) < hot, 0, rng.integers(0, 40, n)).astype(np.int32)
    umi = ((1 << 24) | rng.integers(0, n_umis, n)).astype(np.uint64)
    d = dict(contig=0, ordinal_base=0, pos=pos, flag=np.zeros(n, np.uint16), mapq=np.full(n, 60, np.uint8), cell=cell, umi=umi,
             cig_off=np.arange(n + 1, dtype=np.uint32), cigar=np.full(n, (91 << 4) | 0, np.uint32))
    batches = [util.batch_from_dict(d)]
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 40, batches)
    assert int(exp["count"][2].max()) > (30000 if label == "giant" else 1500)
    util.assert_coo_equal(got, exp, ["count"])


def test_unmapped_flag_with_cigar_include_test():
    """A read with the UNMAP flag but a CIGAR is fetched by its first base only (bam_endpos = pos + 1), while
    read.positions - and with it the include fraction - follows the whole CIGAR (rdr/fc/core.py:32-43).  With
    --exclFLAG 0 such reads reach the include test: 91M starting 30 bases before a region's end has m/n = 31/91."""
    names = ["1"]
    regions = [("1", 101, 150, "ends_at_150"), ("1", 101, 400, "covers_all"), ("1", 121, 121, "one_base")]
    pos = np.array([119, 119, 119, 50, 140], np.int32)              # 0-based
    flag = np.array([4, 0, 4 | 16, 4, 4], np.uint16)
    n = len(pos)
    d = dict(contig=0, ordinal_base=0, pos=pos, flag=flag, mapq=np.full(n, 60, np.uint8), cell=np.zeros(n, np.int32),
             umi=((1 << 24) | np.arange(n)).astype(np.uint64), cig_off=np.arange(n + 1, dtype=np.uint32), cigar=np.full(n, (91 << 4) | 0, np.uint32))
    batches = [util.batch_from_dict(d)]
    for inc in (0.5, 0.9, 0.3, 0, 31, 32):
        got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, batches, excl_flag=0, min_include=inc)
        util.assert_coo_equal(got, exp, ["count"])
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, batches, excl_flag=0, min_include=0.5)
    assert exp["count"][0].tolist() == [1] and exp["count"][2].tolist() == [4]   # 31/91 fails 0.5 in region 0 for mapped and unmapped alike; region 1 holds 4 whole reads
    got, exp, st = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, batches, excl_flag=0, min_include=0.3)
    assert exp["count"][0].tolist() == [0, 1] and exp["count"][2].tolist() == [3, 4]
    util.assert_coo_equal(got, exp, ["count"])


def test_documented_divergences_from_the_reference_are_the_chosen_behaviour():
    """Two inputs on which this implementation does NOT do what the reference would, asserted here so
    that the choice cannot drift silently:
    (1) a read without a CIGAR that reaches the FRACTIONAL include test: the reference compares `None < float` and the whole
        run dies (rdr/fc/core.py:35,161); engine and oracle drop that read - the result equals the run without it.  With an
        integer threshold or min_include = 0 the reference has an answer (the read has no aligned base: it fails `>= 1` and
        passes `0`) and parity holds as everywhere else;
    (2) a region beyond pysam's MAX_POS (2^29 in some 0.15 builds): the reference's fetch raises -> all-zero row; engine and
        oracle take any int32 coordinate and count the reads there."""
    names = ["1"]
    far = (1 << 29) + 1000
    regions = [("1", 101, 400, "g"), ("1", far, far + 500, "beyond_max_pos")]
    M = 0
    def mk(with_cigarless):
        pos = [150, 160, far + 10] + ([170] if with_cigarless else [])
        cig = [[(M, 91)], [(M, 91)], [(M, 91)]] + ([[]] if with_cigarless else [])
        order = np.argsort(np.array(pos), kind="stable")
        cw, off = [], [0]
        for i in order:
            cw += [(l << 4) | op for op, l in cig[i]]; off.append(len(cw))
        n = len(pos)
        d = dict(contig=0, ordinal_base=0, pos=np.array(pos, np.int32)[order], flag=np.zeros(n, np.uint16), mapq=np.full(n, 60, np.uint8),
                 cell=np.zeros(n, np.int32), umi=((1 << 24) | np.arange(n)).astype(np.uint64)[order], cig_off=np.array(off, np.uint32),
                 cigar=np.array(cw, np.uint32))
        return [util.batch_from_dict(d)]
    for inc in (0.9, 0.5):
        got, exp, _ = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, mk(True), min_include=inc, min_len=0)
        util.assert_coo_equal(got, exp, ["count"])
        got0, _, _ = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, mk(False), min_include=inc, min_len=0)
        util.assert_coo_equal(got, got0, ["count"])                      # (1): the CIGAR-less read is dropped
        assert got["count"][0].tolist() == [0, 1] and got["count"][2].tolist() == [2, 1]   # (2): the far region counts its read
    for inc in (0, 1):                                                   # the reference has an answer here: plain parity
        got, exp, _ = util.engine_vs_oracle(capi.XCK_MODE_BASEFC, names, regions, [], 1, mk(True), min_include=inc, min_len=0)
        util.assert_coo_equal(got, exp, ["count"])


def test_push_batch_rejects_inconsistent_host_arrays(small):
    """xck_push_batch checks caller-supplied host arrays before a kernel sees them (api.cpp check_host_batch): offsets that run
    backwards, a cell index outside the cell table, a contig outside the configured ones; the handle stays usable."""
    from xcltk_amd.engine import Engine, XckError
    regions, snps, names, _ = small
    g = soa.gen_reads(regions, names, 20000, 200, seed=77, max_batch=70000)[0]
    for mode in (capi.XCK_MODE_BASEFC, capi.XCK_MODE_BAF):
        eng = Engine(mode, names, regions, 200, snps=snps if mode == capi.XCK_MODE_BAF else ())
        try:
            def broken(**kw):
                d = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in g.items()}
                for k, f in kw.items():
                    d[k] = f(d[k])
                return util.batch_from_dict(d)

            def swap(a):
                a[100], a[101] = a[101] + 5, a[100]
                return a

            def big_cell(a):
                a[7] = 200
                return a
            cases = [broken(cig_off=swap), broken(cell=big_cell), broken(contig=lambda c: len(names))]
            if mode == capi.XCK_MODE_BAF:
                cases.append(broken(seq_off=swap))
            for b, _ in cases:
                with pytest.raises(XckError):
                    eng.push(b)
            good, _ = util.batch_from_dict(g)
            eng.push(good)
            out = eng.finish()
            assert sum(len(v[0]) for v in out.values()) > 0
        finally:
            eng.close()
