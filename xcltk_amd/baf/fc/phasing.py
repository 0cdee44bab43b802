"""phasing.py - region-wise local phasing before the allele-specific counting.

Restates the host logic of xcltk/baf/fc/main.py:107-153 and xcltk/baf/fc/phasing.py:13-78: after Eagle2's reference
phasing, the SNPs of every long enough region are re-phased against each other with the per-cell allele counts of the
cellsnp-lite pileup (baf/localphase.py).  The result is (1) the final haplotype index of every SNP allele - one state per
SNP, shared by all regions that contain it and updated region after region in region-file order, exactly like the SNP
objects the reference shares between its region lists - and (2) the (region, SNP) pairs that drop out of a region's SNP
list because the SNP has no coverage in the (non-reference) cells.  Both go to the engine (xck_config.snps,
xck_config.excl_*): the counting itself is unchanged.
"""
from logging import debug, info
from logging import warning as warn

import numpy as np

from ..localphase import snp_local_phasing
from ...utils.grange import format_chrom

RLP_MIN_LEN = 50000        # baf/fc/config.py:66-68
RLP_MIN_N_SNPS = 2
RLP_MIN_GAP = 50000


def reg_local_phasing(ref_idx, AD, DP, positions):
    """One region (baf/fc/phasing.py:13-78).  ref_idx[j] = current haplotype index of SNP j's REF allele; AD, DP: cell x
    SNP (dense).  -> (kept: bool per SNP, flip: int per KEPT SNP or None when the phasing failed)."""
    cell_idx = DP.sum(axis=1) > 0
    snp_idx = DP.sum(axis=0) > 0
    AD, DP = AD[np.ix_(cell_idx, snp_idx)], DP[np.ix_(cell_idx, snp_idx)]
    kept = np.asarray(snp_idx, dtype=bool).reshape(-1)
    BD = DP - AD
    flip0 = np.array([r == 1 for r in np.asarray(ref_idx)[kept[:len(ref_idx)]]])
    AD_ref = AD * (1 - flip0.T) + BD * flip0.T
    flip = snp_local_phasing(AD_ref, DP, positions=np.asarray(positions)[kept[:len(positions)]])
    if flip is None:
        return kept, None
    flip = 1 - flip if np.mean(flip) > 0.5 else flip + 0              # keep the reference phasing of the majority
    return kept, flip.astype(int)


def local_phasing(regions, snps, csp, ref_cells=None, debug_level=0):
    """regions: [(chrom, start, end_incl, name)] in file order; snps: [(chrom, pos, ref, alt, ref_hap, alt_hap)] (list or
    SnpTable); csp: utils.csp_io.CellSnpData.  -> (ref_hap, alt_hap, excl_region, excl_snp, stats)."""
    n_snp = len(snps)
    s_chrom = [s[0] for s in snps]
    s_pos = np.array([s[1] for s in snps], dtype=np.int64)
    ref_hap = np.array([s[4] for s in snps], dtype=np.int64)
    alt_hap = np.array([s[5] for s in snps], dtype=np.int64)
    by_chrom = {}
    for j, ch in enumerate(s_chrom):
        by_chrom.setdefault(ch, []).append(j)
    for ch, idx in by_chrom.items():
        idx = np.array(idx, dtype=np.int64)
        by_chrom[ch] = idx[np.argsort(s_pos[idx], kind="stable")]       # the region's SNP list is sorted by position
    if ref_cells is not None:
        csp = csp.subset_cells(~np.isin(np.array(csp.cells, dtype=object), np.array(list(ref_cells), dtype=object)))
    c_chrom = np.array([format_chrom(str(c)) for c in csp.chrom], dtype=object)
    AD_all, DP_all = csp.AD.tocsc(), csp.DP.tocsc()
    excl_region, excl_snp = [], []
    n_rlp = n_failed = n_slp = n_flipped = 0
    for g, (ch, start, end, name) in enumerate(regions):
        if end + 1 - start < RLP_MIN_LEN:
            continue
        cand = by_chrom.get(ch)
        if cand is None:
            continue
        lo, hi = np.searchsorted(s_pos[cand], start, "left"), np.searchsorted(s_pos[cand], end, "right")
        lst = cand[lo:hi]                                                # start <= pos <= end (baf/fc/main.py:93)
        if len(lst) < max(1, RLP_MIN_N_SNPS):
            continue
        if s_pos[lst[-1]] - s_pos[lst[0]] + 1 < RLP_MIN_GAP:
            continue
        cols = np.flatnonzero((c_chrom == ch) & (csp.pos >= start) & (csp.pos < end + 1))
        AD, DP = AD_all[:, cols].toarray(), DP_all[:, cols].toarray()
        if AD.shape[1] > len(lst):
            # more pileup columns than phased SNPs in the region: the reference takes its column mask (depth > 0) over ALL columns,
            # pairs it with the SNP list by zip() - the list's length - and multiplies the filtered matrices with a vector as long
            # as the filtered list (baf/fc/phasing.py:43-52): that only works when both filters keep the same number, i.e. when
            # the columns beyond the list's length are empty (golden case phasing_baf_surplus_pileup); otherwise numpy stops it
            covered = np.asarray(DP.sum(axis=0) > 0).reshape(-1)
            if int(covered.sum()) != int(covered[:len(lst)].sum()):
                raise ValueError("region '%s': %d SNPs in the phased list but %d in the cellsnp pileup, %d of them covered beyond the list's length"
                                 % (name, len(lst), AD.shape[1], int(covered[len(lst):].sum())))
        if AD.shape[1] < len(lst):
            # fewer columns: zip() pairs them with the FIRST SNPs of the list and the rest of the list leaves this region
            # (baf/fc/phasing.py:47; golden case phasing_baf_short_pileup)
            for j in lst[AD.shape[1]:].tolist():
                excl_region.append(g); excl_snp.append(j)
            lst = lst[:AD.shape[1]]
        kept, flip = reg_local_phasing(ref_hap[lst], AD, DP, s_pos[lst])
        kept = kept[:len(lst)]                                          # (a column mask longer than the list: zip() stops at the list's end)
        for j in lst[~kept].tolist():                                   # dropped from THIS region's list, whatever the phasing says
            excl_region.append(g); excl_snp.append(j)
        lst_kept = lst[kept]
        if flip is None:
            warn("local phasing for region '%s' failed!" % name)
            n_failed += 1
        else:
            sel = lst_kept[flip == 1]
            ref_hap[sel], alt_hap[sel] = 1 - ref_hap[sel], 1 - alt_hap[sel]
            n_flipped += int(np.sum(flip))
            if debug_level > 1:
                debug("region '%s': #SNPs - total=%d; flipped=%d" % (name, len(lst_kept), int(np.sum(flip))))
        n_slp += len(lst_kept)
        n_rlp += 1
    info("#regions: total=%d; local_phasing=%d; local_phasing_failed=%d." % (len(regions), n_rlp, n_failed))
    info("#SNPs: local_phasing=%d; local_phasing_flipped=%d." % (n_slp, n_flipped))
    return (ref_hap, alt_hap, np.array(excl_region, dtype=np.int32), np.array(excl_snp, dtype=np.int32),
            dict(n_rlp=n_rlp, n_failed=n_failed, n_slp=n_slp, n_flipped=n_flipped))
