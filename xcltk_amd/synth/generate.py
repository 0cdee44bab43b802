"""generate.py - seeded synthetic single-cell datasets (BAM + barcodes + regions + phased SNPs).

Shapes follow SURVEY.md section 8d / BASELINE.json configs: droplet ("10x") mode with CB/UB
tags and well ("SMART-seq") mode with one BAM per cell and no tags.  All files are
produced by this repo's own writer (bamwriter.py); nothing is read from the reference.
"""

import bisect
import os

import numpy as np

from .bamwriter import BamWriter, parse_cigar

_ACGT = "ACGT"


def _rand_seq(rng, n):
    return "".join(_ACGT[i] for i in rng.integers(0, 4, n))


def make_genes(rng, contigs, n_genes, min_len=2000, max_len=100000, nested_frac=0.15):
    """Return list of (contig_idx, start1, end1_incl, name), sorted by contig/start."""
    tot = float(sum(l for _, l in contigs))
    genes = []
    k = 0
    for ci, (cname, clen) in enumerate(contigs):
        n_c = int(round(n_genes * clen / tot)) if ci < len(contigs) - 1 else n_genes - len(genes)
        mine = []
        for _ in range(max(n_c, 0)):
            ln = int(np.exp(rng.uniform(np.log(min_len), np.log(max_len))))
            ln = min(ln, clen - 2000)
            if mine and rng.random() < nested_frac:
                ps, pe = mine[int(rng.integers(0, len(mine)))][:2]
                s = int(rng.integers(max(1, ps - ln // 2), pe + 1))
            else:
                s = int(rng.integers(1000, max(1001, clen - ln - 1000)))
            e = min(s + ln - 1, clen)
            mine.append((s, e))
        mine.sort()
        for s, e in mine:
            k += 1
            genes.append((ci, s, e, "G%06d" % k))
    return genes


def make_snps(rng, genes, n_snps, dup_frac=0.0):
    """Het SNPs inside genes (prob ~ gene length). Returns sorted list of
    (contig_idx, pos1, ref, alt, ref_hap, alt_hap)."""
    lens = np.array([e - s + 1 for _, s, e, _ in genes], dtype=np.float64)
    p = lens / lens.sum()
    seen = set()
    snps = []
    tries = 0
    while len(snps) < n_snps and tries < 50 * n_snps:
        tries += 1
        g = genes[int(rng.choice(len(genes), p=p))]
        pos = int(rng.integers(g[1], g[2] + 1))
        if (g[0], pos) in seen and rng.random() >= dup_frac:
            continue
        seen.add((g[0], pos))
        r = int(rng.integers(0, 4))
        a = (r + int(rng.integers(1, 4))) % 4
        h = int(rng.integers(0, 2))
        snps.append((g[0], pos, _ACGT[r], _ACGT[a], h, 1 - h))
    snps.sort(key=lambda x: (x[0], x[1]))
    return snps


def _cigar_for(rng, kind, L):
    if kind == 0:
        return "%dM" % L
    if kind == 1:                                    # spliced
        a = int(rng.integers(10, L - 10))
        gap = int(rng.integers(50, 20000))
        return "%dM%dN%dM" % (a, gap, L - a)
    if kind == 2:                                    # insertion
        a = int(rng.integers(10, L - 15))
        i = int(rng.integers(1, 4))
        return "%dM%dI%dM" % (a, i, L - a - i)
    if kind == 3:                                    # deletion
        a = int(rng.integers(10, L - 10))
        dl = int(rng.integers(1, 6))
        return "%dM%dD%dM" % (a, dl, L - a)
    s = int(rng.integers(1, 40))                     # soft clip (either side)
    if rng.random() < 0.5:
        return "%dS%dM" % (s, L - s)
    return "%dM%dS" % (L - s, s)


def _aligned_query_offsets(pos, cig):
    """[(ref_pos0, query_off)] blocks as (ref_start, q_start, len) for M/=/X."""
    blocks = []
    r, q = pos, 0
    for op, l in cig:
        if op in (0, 7, 8):
            blocks.append((r, q, l))
            r += l
            q += l
        elif op in (2, 3):
            r += l
        elif op in (1, 4):
            q += l
    return blocks, r


def write_tables(out_dir, contigs, genes, snps, barcodes, chr_prefix_regions=None):
    """Write barcodes.tsv, regions.tsv, snps.tsv, snps.vcf. Returns dict of paths."""
    os.makedirs(out_dir, exist_ok=True)
    p = {}
    p["barcodes"] = os.path.join(out_dir, "barcodes.tsv")
    with open(p["barcodes"], "w") as fp:
        fp.write("".join(b + "\n" for b in barcodes))
    p["regions"] = os.path.join(out_dir, "regions.tsv")
    with open(p["regions"], "w") as fp:
        for ci, s, e, name in genes:
            fp.write("%s\t%d\t%d\t%s\n" % (contigs[ci][0], s, e, name))
    p["snps_tsv"] = os.path.join(out_dir, "snps.tsv")
    with open(p["snps_tsv"], "w") as fp:
        fp.write("chrom\tpos\tref\talt\tref_hap\talt_hap\n")
        for ci, pos, r, a, rh, ah in snps:
            fp.write("%s\t%d\t%s\t%s\t%d\t%d\n" % (contigs[ci][0], pos, r, a, rh, ah))
    p["snps_vcf"] = os.path.join(out_dir, "snps.vcf")
    with open(p["snps_vcf"], "w") as fp:
        fp.write("##fileformat=VCFv4.2\n")
        for n, l in contigs:
            fp.write("##contig=<ID=%s,length=%d>\n" % (n, l))
        fp.write("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        fp.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n")
        for ci, pos, r, a, rh, ah in snps:
            fp.write("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%d|%d\n" % (contigs[ci][0], pos, r, a, rh, ah))
    return p


def make_10x_dataset(out_dir, n_reads=10000, n_barcodes=1000, n_snps=500, n_genes=200,
                     contigs=(("chr1", 2000000),), seed=1, read_len=91, umi_len=12,
                     bam_contig_prefix=None, align_records=True, n_bams=1,
                     frac_missing_cb=0.03, frac_missing_ub=0.03, frac_cb_outside=0.02,
                     iupac_frac=0.002, paired=False, level=6, with_index=True,
                     lowercase_none=False, odd_frac=0.0):
    """Droplet-style dataset. Returns dict with paths and the python-side tables.

    odd_frac > 0 marks that share of the reads as oddities a fetch still returns: the unmapped
    flag on a read that has a CIGAR, a read without a CIGAR, or both (no rng draws when 0).

    n_bams > 1 splits molecules across BAMs that share the barcode list
    (multi-BAM barcode mode of the reference, rdr/fc/core.py:153).
    """
    rng = np.random.default_rng(seed)
    contigs = [tuple(c) for c in contigs]
    os.makedirs(out_dir, exist_ok=True)
    genes = make_genes(rng, contigs, n_genes)
    snps = make_snps(rng, genes, n_snps)
    barcodes = set()
    while len(barcodes) < n_barcodes:
        barcodes.add(_rand_seq(rng, 16) + "-1")
    barcodes = sorted(barcodes)
    outside = [_rand_seq(rng, 16) + "-1" for _ in range(max(8, n_barcodes // 50))]
    outside = [b for b in outside if b not in set(barcodes)]

    snp_pos = [[] for _ in contigs]
    snp_rec = [[] for _ in contigs]
    for s in snps:
        snp_pos[s[0]].append(s[1] - 1)
        snp_rec[s[0]].append(s)

    # gene expression weights ~ Zipf
    w = 1.0 / np.arange(1, len(genes) + 1) ** 0.8
    rng.shuffle(w)
    w /= w.sum()

    reads = []          # (bam_idx, tid, pos, qname, flag, mapq, cigar, seq, tags)
    rid = 0
    kinds = np.array([0, 1, 2, 3, 4])
    kind_p = np.array([0.80, 0.12, 0.03, 0.03, 0.02])
    while len(reads) < n_reads:
        g = genes[int(rng.choice(len(genes), p=w))]
        ci, gs, ge, _ = g
        clen = contigs[ci][1]
        hap = int(rng.integers(0, 2))
        u = rng.random()
        if u < frac_missing_cb:
            cb = None
        elif u < frac_missing_cb + frac_cb_outside:
            cb = outside[int(rng.integers(0, len(outside)))]
        else:
            cb = barcodes[int(rng.integers(0, len(barcodes)))]
        ub = None if rng.random() < frac_missing_ub else _rand_seq(rng, umi_len)
        if ub is not None and rng.random() < 0.002:
            ub = ub[:3] + "N" + ub[4:]                 # IUPAC UMI -> interned path
        anchor = int(rng.integers(max(0, gs - 1 - read_len + 5), ge))
        bam_idx = int(rng.integers(0, n_bams))
        for _ in range(int(rng.integers(1, 9))):
            if len(reads) >= n_reads:
                break
            pos = min(max(0, anchor + int(rng.integers(0, 200))), clen - read_len - 25000)
            pos = max(pos, 0)
            cig_s = _cigar_for(rng, int(rng.choice(kinds, p=kind_p)), read_len)
            cig = parse_cigar(cig_s)
            blocks, _ = _aligned_query_offsets(pos, cig)
            seq = list(_rand_seq(rng, read_len))
            # impose haplotype alleles at covered SNPs
            sp = snp_pos[ci]
            for (r0, q0, l) in blocks:
                lo = bisect.bisect_left(sp, r0)
                hi = bisect.bisect_left(sp, r0 + l)
                for k in range(lo, hi):
                    _, p1, ref, alt, rh, ah = snp_rec[ci][k]
                    base = ref if rh == hap else alt
                    e = rng.random()
                    if e < iupac_frac:
                        base = "N" if rng.random() < 0.7 else "R"
                    elif e < iupac_frac + 0.01:
                        base = _ACGT[int(rng.integers(0, 4))]
                    seq[q0 + (sp[k] - r0)] = base
            mapq = 255 if rng.random() < 0.9 else int(rng.choice([0, 1, 3]))
            flag = 16 if rng.random() < 0.5 else 0
            if rng.random() < 0.03:
                flag |= 256
            if rng.random() < 0.05:
                flag |= 1024
            if paired:
                flag |= 1 | (2 if rng.random() < 0.9 else 0) | (64 if rng.random() < 0.5 else 128)
            if odd_frac > 0 and rng.random() < odd_frac:
                odd = int(rng.integers(0, 3))
                if odd != 1:
                    flag |= 4
                if odd != 0:
                    cig_s = "*"
            tags = [("NH", 1, "C")]
            if cb is not None:
                tags.append(("CB", cb))
            if ub is not None:
                tags.append(("UB", ub))
            rid += 1
            reads.append((bam_idx, ci, pos, "r%09d" % rid, flag, mapq, cig_s, "".join(seq), tags))

    refs = [((bam_contig_prefix + n[3:] if n.lower().startswith("chr") else bam_contig_prefix + n)
             if bam_contig_prefix is not None else n, l) for n, l in contigs]
    bam_paths = []
    for b in range(n_bams):
        mine = [r for r in reads if r[0] == b]
        mine.sort(key=lambda r: (r[1], r[2]))          # stable: coordinate order
        path = os.path.join(out_dir, "possorted_%d.bam" % b if n_bams > 1 else "possorted.bam")
        bw = BamWriter(path, refs, align_records=align_records, level=level)
        for (_, tid, pos, qn, flag, mapq, cig_s, seq, tags) in mine:
            bw.write(tid, pos, qn, flag, mapq, cig_s, seq, tags)
        bw.close()
        if with_index:
            bw.write_index()
        bam_paths.append(path)

    paths = write_tables(out_dir, contigs, genes, snps, barcodes)
    paths["bams"] = bam_paths
    paths["bam"] = bam_paths[0]
    paths.update(dict(contigs=contigs, genes=genes, snps=snps, barcode_list=barcodes,
                      n_reads=len(reads), odd_frac=odd_frac))
    return paths


def make_smartseq_dataset(out_dir, n_cells=8, reads_per_cell=2000, n_snps=300, n_genes=100,
                          contigs=(("1", 1000000),), seed=5, read_len=75, level=6,
                          with_index=True):
    """Well-based dataset: one BAM per cell, paired-end, no CB/UB; mates share qname
    (reference mode --cellTAG None --UMItag None with --samList / --sampleList)."""
    rng = np.random.default_rng(seed)
    contigs = [tuple(c) for c in contigs]
    os.makedirs(out_dir, exist_ok=True)
    genes = make_genes(rng, contigs, n_genes)
    snps = make_snps(rng, genes, n_snps)
    snp_pos = [[] for _ in contigs]
    snp_rec = [[] for _ in contigs]
    for s in snps:
        snp_pos[s[0]].append(s[1] - 1)
        snp_rec[s[0]].append(s)
    w = 1.0 / np.arange(1, len(genes) + 1) ** 0.8
    rng.shuffle(w)
    w /= w.sum()
    sample_ids = ["cell%03d" % i for i in range(n_cells)]
    bam_paths = []
    for c in range(n_cells):
        reads = []
        frag = 0
        while len(reads) < reads_per_cell:
            ci, gs, ge, _ = genes[int(rng.choice(len(genes), p=w))]
            clen = contigs[ci][1]
            hap = int(rng.integers(0, 2))
            frag += 1
            qn = "c%03d_f%07d" % (c, frag)
            p1 = int(rng.integers(max(0, gs - 60), ge))
            p1 = min(p1, clen - 2 * read_len - 25000)
            p2 = p1 + int(rng.integers(20, 300))
            proper = rng.random() < 0.92
            for mate, pos in ((0, p1), (1, p2)):
                kind = int(rng.choice([0, 1, 3, 4], p=[0.85, 0.1, 0.03, 0.02]))
                cig_s = _cigar_for(rng, kind, read_len)
                cig = parse_cigar(cig_s)
                blocks, _ = _aligned_query_offsets(pos, cig)
                seq = list(_rand_seq(rng, read_len))
                sp = snp_pos[ci]
                for (r0, q0, l) in blocks:
                    lo = bisect.bisect_left(sp, r0)
                    hi = bisect.bisect_left(sp, r0 + l)
                    for k in range(lo, hi):
                        _, pp, ref, alt, rh, ah = snp_rec[ci][k]
                        base = ref if rh == hap else alt
                        if rng.random() < 0.01:
                            base = _ACGT[int(rng.integers(0, 4))]
                        seq[q0 + (sp[k] - r0)] = base
                flag = 1 | (2 if proper else 0) | (64 if mate == 0 else 128) | (16 if mate else 32)
                if rng.random() < 0.04:
                    flag |= 1024
                mapq = 255 if rng.random() < 0.92 else int(rng.choice([0, 3]))
                reads.append((ci, pos, qn, flag, mapq, cig_s, "".join(seq), [("NH", 1, "C")]))
        reads.sort(key=lambda r: (r[0], r[1]))
        path = os.path.join(out_dir, "cell%03d.bam" % c)
        bw = BamWriter(path, contigs, level=level)
        for (tid, pos, qn, flag, mapq, cig_s, seq, tags) in reads:
            bw.write(tid, pos, qn, flag, mapq, cig_s, seq, tags)
        bw.close()
        if with_index:
            bw.write_index()
        bam_paths.append(path)
    paths = write_tables(out_dir, contigs, genes, snps, sample_ids)
    os.remove(paths["barcodes"])
    del paths["barcodes"]
    paths["sam_list"] = os.path.join(out_dir, "bam_list.txt")
    with open(paths["sam_list"], "w") as fp:
        fp.write("".join(p + "\n" for p in bam_paths))
    paths["sample_list"] = os.path.join(out_dir, "sample_ids.txt")
    with open(paths["sample_list"], "w") as fp:
        fp.write("".join(s + "\n" for s in sample_ids))
    paths.update(dict(bams=bam_paths, contigs=contigs, genes=genes, snps=snps,
                      sample_ids=sample_ids, n_reads=n_cells * reads_per_cell))
    return paths
