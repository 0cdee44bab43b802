"""soa.py - vectorised synthetic record batches (structure-of-arrays) for benchmarks and
large parity tests, bypassing BAM files.

Shapes follow SURVEY.md section 8d: droplet data with 1-8 reads per UMI, read length 91,
CIGAR mix 80 % 91M / 12 % spliced / 3 % insertion / 3 % deletion / 2 % soft clip, MAPQ 255
(90 %) or {0,1,3}, flags 0/16 (+256 for 3 %, +1024 for 5 %), 3 % missing CB, 3 % missing UB,
genes with nested/overlapping structure, Zipf-like expression.  Everything is seeded.
"""

import numpy as np

HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
                145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
                101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
                156040895, 57227415]
HG38_NAMES = [str(i) for i in range(1, 23)] + ["X", "Y"]


def make_tables(n_genes, n_snps, contig_lengths, seed, min_len=2000, max_len=400000,
                nested_frac=0.15):
    """-> regions [(chrom, start1, end1, name)], snps [(chrom, pos1, ref, alt, rh, ah)],
    names.  Gene lengths are log-uniform; genes are emitted sorted by (contig, start)."""
    rng = np.random.default_rng(seed)
    lens = np.asarray(contig_lengths, dtype=np.int64)
    names = HG38_NAMES[:len(lens)] if len(lens) <= 24 else [str(i) for i in range(len(lens))]
    per = np.floor(n_genes * lens / lens.sum()).astype(np.int64)
    per[0] += n_genes - per.sum()
    regions = []
    g_contig, g_s, g_e = [], [], []
    for ci, (L, k) in enumerate(zip(lens, per)):
        ln = np.exp(rng.uniform(np.log(min_len), np.log(max_len), k)).astype(np.int64)
        ln = np.minimum(ln, L - 4000)
        st = rng.integers(1000, np.maximum(1001, L - ln - 1000))
        nest = rng.random(k) < nested_frac
        if k > 1:
            host = rng.integers(0, k, k)
            st = np.where(nest, np.clip(st[host] + rng.integers(-2000, 2000, k), 1, L - ln - 1), st)
        order = np.argsort(st, kind="stable")
        st, ln = st[order], ln[order]
        g_contig.append(np.full(k, ci, dtype=np.int64)); g_s.append(st); g_e.append(st + ln - 1)
    g_contig = np.concatenate(g_contig); g_s = np.concatenate(g_s); g_e = np.concatenate(g_e)
    regions = [(names[c], int(s), int(e), "G%06d" % (i + 1)) for i, (c, s, e) in enumerate(zip(g_contig, g_s, g_e))]
    snps = []
    if n_snps:
        w = (g_e - g_s + 1).astype(np.float64); w /= w.sum()
        gi = rng.choice(len(g_s), size=n_snps, p=w)
        pos = g_s[gi] + (rng.random(n_snps) * (g_e[gi] - g_s[gi] + 1)).astype(np.int64)
        ref = rng.integers(0, 4, n_snps); alt = (ref + rng.integers(1, 4, n_snps)) % 4
        rh = rng.integers(0, 2, n_snps)
        key = g_contig[gi] * (1 << 32) + pos
        _, first = np.unique(key, return_index=True)
        first.sort()
        for j in first:
            snps.append((names[g_contig[gi[j]]], int(pos[j]), "ACGT"[ref[j]], "ACGT"[alt[j]], int(rh[j]), int(1 - rh[j])))
        snps.sort(key=lambda s: (names.index(s[0]), s[1]))
    return regions, snps, names


def gen_reads(regions, names, n_reads, n_cells, seed, umi_len=12, read_len=91, with_seq=True,
              bam_index=0, frac_nocell=0.05, frac_noumi=0.03, max_batch=None, seq_pool=None):
    """Vectorised reads over the given regions. Returns a list of batch dicts (one or more per
    contig, split at max_batch reads) with the fields of xck_batch as numpy arrays."""
    rng = np.random.default_rng(seed)
    cidx = {n: i for i, n in enumerate(names)}
    g_c = np.array([cidx[r[0]] for r in regions], dtype=np.int64)
    g_s = np.array([r[1] for r in regions], dtype=np.int64)
    g_e = np.array([r[2] for r in regions], dtype=np.int64)
    w = 1.0 / np.arange(1, len(regions) + 1) ** 0.8
    rng.shuffle(w); w /= w.sum()
    n_mol = max(1, int(n_reads / 4.5) + 1)
    per = rng.integers(1, 9, n_mol)
    while per.sum() < n_reads:
        per = np.concatenate([per, rng.integers(1, 9, max(16, n_mol // 10))])
    cs = np.cumsum(per)
    n_mol = int(np.searchsorted(cs, n_reads) + 1)
    per = per[:n_mol]; per[-1] -= int(per.sum() - n_reads)
    mg = rng.choice(len(regions), size=n_mol, p=w)
    m_cell = rng.integers(0, n_cells, n_mol).astype(np.int32)
    m_cell[rng.random(n_mol) < frac_nocell] = -1
    m_umi = (np.uint64(1) << np.uint64(2 * umi_len)) | rng.integers(0, 1 << (2 * umi_len), n_mol, dtype=np.uint64)
    m_umi[rng.random(n_mol) < frac_noumi] = np.uint64(0xFFFFFFFFFFFFFFFF)
    m_anchor = g_s[mg] - 1 - 40 + (rng.random(n_mol) * (g_e[mg] - g_s[mg] + 40)).astype(np.int64)
    rep = np.repeat(np.arange(n_mol), per)
    n = len(rep)
    contig = g_c[mg][rep]
    pos = np.maximum(m_anchor[rep] + rng.integers(0, 200, n), 0)
    cell = m_cell[rep]; umi = m_umi[rep]
    u = rng.random(n)
    kind = np.select([u < 0.80, u < 0.92, u < 0.95, u < 0.98], [0, 1, 2, 3], 4).astype(np.int8)
    mapq = np.where(rng.random(n) < 0.9, 255, rng.choice([0, 1, 3], n)).astype(np.uint8)
    flag = np.where(rng.random(n) < 0.5, 16, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.03, 256, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.05, 1024, 0).astype(np.uint16)
    # CIGAR words: up to 3 ops per read
    L = read_len
    a = rng.integers(10, L - 15, n); x = rng.integers(1, 4, n)
    gap = rng.integers(50, 20000, n); dl = rng.integers(1, 6, n); sc = rng.integers(1, 40, n)
    left_clip = rng.random(n) < 0.5
    M, I, D, N, S = 0, 1, 2, 3, 4
    w0 = np.zeros(n, dtype=np.uint32); w1 = np.zeros(n, dtype=np.uint32); w2 = np.zeros(n, dtype=np.uint32)
    ncig = np.ones(n, dtype=np.uint32)
    k0 = kind == 0; w0[k0] = (L << 4) | M
    k1 = kind == 1; w0[k1] = (a[k1] << 4) | M; w1[k1] = (gap[k1] << 4) | N; w2[k1] = ((L - a[k1]) << 4) | M; ncig[k1] = 3
    k2 = kind == 2; w0[k2] = (a[k2] << 4) | M; w1[k2] = (x[k2] << 4) | I; w2[k2] = ((L - a[k2] - x[k2]) << 4) | M; ncig[k2] = 3
    k3 = kind == 3; w0[k3] = (a[k3] << 4) | M; w1[k3] = (dl[k3] << 4) | D; w2[k3] = ((L - a[k3]) << 4) | M; ncig[k3] = 3
    k4 = (kind == 4) & left_clip; w0[k4] = (sc[k4] << 4) | S; w1[k4] = ((L - sc[k4]) << 4) | M; ncig[k4] = 2
    k5 = (kind == 4) & ~left_clip; w0[k5] = ((L - sc[k5]) << 4) | M; w1[k5] = (sc[k5] << 4) | S; ncig[k5] = 2
    # coordinate sort within contig (stable, like a sorted BAM)
    order = np.lexsort((pos, contig))
    contig, pos, cell, umi, mapq, flag = contig[order], pos[order], cell[order], umi[order], mapq[order], flag[order]
    w0, w1, w2, ncig = w0[order], w1[order], w2[order], ncig[order]
    nb = (L + 1) // 2
    if with_seq and seq_pool is None:
        pool_n = 1 << 22
        codes = rng.integers(0, 4, pool_n * 2)
        nibs = (1 << codes).astype(np.uint8)
        nibs[rng.random(pool_n * 2) < 0.002] = 15
        seq_pool = (nibs[0::2] << 4) | nibs[1::2]
    out = []
    bounds = np.flatnonzero(np.diff(contig)) + 1
    starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [n]])
    rec = 0
    for s0, e0 in zip(starts, ends):
        step = max_batch or (e0 - s0)
        for s in range(s0, e0, step):
            e = min(e0, s + step)
            m = e - s
            nc = ncig[s:e]
            cig_off = np.zeros(m + 1, dtype=np.uint32); np.cumsum(nc, out=cig_off[1:])
            cigar = np.zeros(int(cig_off[-1]), dtype=np.uint32)
            base = cig_off[:-1]
            cigar[base] = w0[s:e]
            m2 = nc >= 2; cigar[base[m2] + 1] = w1[s:e][m2]
            m3 = nc >= 3; cigar[base[m3] + 2] = w2[s:e][m3]
            b = dict(contig=int(contig[s]), n_reads=int(m), ordinal_base=(bam_index << 40) | rec,
                     pos=pos[s:e].astype(np.int32), flag=flag[s:e].copy(), mapq=mapq[s:e].copy(),
                     cell=cell[s:e].copy(), umi=umi[s:e].copy(), cig_off=cig_off, cigar=cigar)
            if with_seq:
                b["seq_off"] = (np.arange(m + 1, dtype=np.uint64) * nb).astype(np.uint32)
                start = int(rng.integers(0, len(seq_pool) - 1))
                idx = (start + np.arange(m * nb, dtype=np.int64)) % len(seq_pool)
                b["seq"] = seq_pool[idx]
            out.append(b)
            rec += m
    return out
