"""Random engine-vs-oracle cases (tables, read mixes incl. unmapped-flagged / CIGAR-less / sequence-less reads, batch cuts,
filters, key widths) shared by tests/test_gpu_fuzz.py and tools/fuzz_parity.py."""
import numpy as np

import util
from xcltk_amd import capi

M, I, D, N, S, H, P, EQ, X = 0, 1, 2, 3, 4, 5, 6, 7, 8


def make_case(seed, odd_regions=True, many_cells=True, long=True, neg_pos=True):
    rng = np.random.default_rng(seed)
    n_contigs = int(rng.integers(1, 4))
    names = [str(i + 1) for i in range(n_contigs)]
    span = int(rng.choice([20000, 150000, 2000000]))
    n_reg = int(rng.integers(1, 60))
    regions = []
    for g in range(n_reg):
        c = names[int(rng.integers(0, n_contigs))]
        s = int(rng.integers(1, span - 100)); ln = int(rng.choice([50, 500, 5000, 60000]))
        regions.append((c, s, min(span, s + int(rng.integers(1, ln + 1))), "g%d" % g))
    if rng.random() < 0.3:
        regions.append(regions[0][:3] + ("dup",))                       # identical interval twice
    if (rng.random() < 0.2) and odd_regions:
        regions += [(names[0], 0, 500, "starts_at_0"), (names[0], 900, 100, "inverted"), (names[-1], span - 3, span + 5000, "past_the_end")]
    snp_step = int(rng.choice([3, 11, 37, 200, 3000]))
    snps = []
    for c in names:
        for p in range(int(rng.integers(1, 50)), span - 5, snp_step):
            if rng.random() < 0.8:
                r = int(rng.integers(0, 4)); a = (r + int(rng.integers(1, 4))) % 4; h = int(rng.integers(0, 2))
                snps.append((c, p, "ACGT"[r], "ACGT"[a], h, 1 - h))
                if rng.random() < 0.01:
                    snps.append((c, p, "ACGT"[a], "ACGT"[r], 1 - h, h))  # duplicate position
    if len(snps) > 60000:
        snps = snps[:60000]
    n_cells = int(rng.choice([1, 3, 40, 700, 70000], p=[.24, .24, .24, .24, .04]))
    if not many_cells and n_cells > 700:
        n_cells = 700
    n_umis = int(rng.choice([5, 200, 50000]))
    # key codes as the host decoder would produce them: 2-bit coded UMIs of one length, and (for some cases) interned ids,
    # which carry the top bit of the UMI field - those keys cannot be squeezed and take the classic sort / fold
    from xcltk_amd.engine import Engine
    probe = Engine(capi.XCK_MODE_BAF, names, regions, n_cells, snps=snps, decode_only=True)
    umi_bits = probe.umi_bits
    probe.close()
    umi_len = int(rng.choice([4, 12, min(16, (umi_bits - 2) // 2)]))
    interned_frac = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
    n_reads = int(rng.choice([300, 5000, 40000]))
    gap_max = int(rng.choice([100, 3000, 30000]))
    L = int(rng.choice([30, 91, 150]))
    long_reads = (rng.random() < 0.15) and long
    if long_reads:
        n_reads = min(n_reads, 3000)                                    # the oracle walks every CIGAR per region / SNP                                    # hundreds of CIGAR ops per read: beyond the per-tile LDS staging
    recs = []
    for _ in range(n_reads):
        ci = int(rng.integers(0, n_contigs))
        pos = int(rng.integers(0, span - 10))
        k = rng.integers(0, 12)
        a = int(rng.integers(1, L))
        if k < 4: cig = [(M, L)]
        elif k < 6: cig = [(M, a), (N, int(rng.integers(1, gap_max))), (M, L - a)]
        elif k == 6: cig = [(M, a), (D, int(rng.integers(1, 400))), (M, L - a)]
        elif k == 7: cig = [(S, 3), (M, a), (I, 2), (M, L - a)] if L - a > 0 else [(M, L)]
        elif k == 8: cig = [(EQ, a), (X, 1), (M, max(1, L - a - 1))]
        elif k == 9: cig = [(H, 5), (M, a), (N, int(rng.integers(1, gap_max))), (M, 5), (N, int(rng.integers(1, gap_max))), (M, L), (S, 2)]
        elif k == 10: cig = [(S, L)] if rng.random() < 0.3 else [(M, 1)]
        else: cig = []                                                   # no CIGAR at all
        if long_reads and rng.random() < 0.5:
            cig = []
            for _ in range(int(rng.integers(20, 300))):
                cig.append((int(rng.choice([M, M, M, EQ, X])), int(rng.integers(1, 40))))
                cig.append((int(rng.choice([I, D, N, S if not cig else I])), int(rng.integers(1, 30))))
        if (rng.random() < 0.01) and neg_pos:
            pos = -1
        cig = [(op, l) for op, l in cig if l > 0]
        qlen = sum(l for op, l in cig if op in (M, I, S, EQ, X))
        noseq = rng.random() < 0.05 or not cig
        nib = (1 << rng.integers(0, 4, max(qlen, 1))).astype(np.uint8)[:qlen]
        nib[rng.random(qlen) < 0.02] = 15
        if qlen % 2: nib = np.append(nib, 0)
        seq = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8) if (qlen and not noseq) else np.zeros(0, np.uint8)
        flag = int(rng.choice([0, 16, 256, 1024, 4, 1, 3, 2048], p=[.55, .25, .04, .04, .03, .03, .03, .03]))
        mapq = int(rng.choice([255, 60, 20, 19, 0], p=[.5, .3, .08, .06, .06]))
        cell = int(rng.integers(-1, n_cells))
        if rng.random() < 0.03:
            umi = np.uint64(capi.XCK_UMI_NONE)
        elif rng.random() < interned_frac:
            umi = np.uint64((1 << (umi_bits - 1)) | int(rng.integers(0, n_umis)))
        else:
            umi = np.uint64((1 << (2 * umi_len)) | (int(rng.integers(0, n_umis)) % (1 << (2 * umi_len))))
        recs.append((ci, pos, cig, seq, flag, mapq, cell, umi))
    recs.sort(key=lambda r: (r[0], r[1]))
    batches, ordinal = [], 0
    i = 0
    while i < len(recs):
        ci = recs[i][0]
        j = i
        cut = int(rng.choice([len(recs), 1000, 177]))
        while j < len(recs) and recs[j][0] == ci and j - i < cut:
            j += 1
        part = recs[i:j]
        cig_off = np.zeros(len(part) + 1, np.uint32); seq_off = np.zeros(len(part) + 1, np.uint32); cw, sq = [], []
        for t, r in enumerate(part):
            cw += [(l << 4) | op for op, l in r[2]]; sq.append(r[3])
            cig_off[t + 1] = len(cw); seq_off[t + 1] = seq_off[t] + len(r[3])
        d = dict(contig=ci, ordinal_base=ordinal, pos=np.array([r[1] for r in part], np.int32), flag=np.array([r[4] for r in part], np.uint16),
                 mapq=np.array([r[5] for r in part], np.uint8), cell=np.array([r[6] for r in part], np.int32), umi=np.array([r[7] for r in part], np.uint64),
                 cig_off=cig_off, cigar=np.array(cw if cw else [0], np.uint32)[:len(cw)] if cw else np.zeros(0, np.uint32),
                 seq_off=seq_off, seq=np.concatenate(sq) if sq else np.zeros(0, np.uint8))
        if len(d["cigar"]) == 0: d["cigar"] = np.zeros(1, np.uint32)
        if len(d["seq"]) == 0: d["seq"] = np.zeros(1, np.uint8)
        batches.append(util.batch_from_dict(d)); ordinal += len(part); i = j
    opts = dict(min_mapq=int(rng.choice([0, 20, 30])), min_len=int(rng.choice([0, 10, 30])), excl_flag=int(rng.choice([772, 1796, 0, 4])),
                incl_flag=int(rng.choice([0, 0, 0, 16])), no_orphan=bool(rng.random() < 0.7))
    fc = dict(opts, min_include=float(rng.choice([0.9, 0.5, 0.0, 1.0, 20, 0.3333])))
    baf = dict(opts, min_count=int(rng.choice([1, 1, 2, 11])), min_maf=float(rng.choice([0, 0, 0.1])), no_dup_hap=bool(rng.random() < 0.6))
    flags = capi.XCK_F_FORCE_KEY128 if rng.random() < 0.2 else 0
    return names, regions, snps, n_cells, batches, fc, baf, flags


