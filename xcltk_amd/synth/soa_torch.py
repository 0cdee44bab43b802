"""soa_torch.py - the synthetic record generator of soa.py, on the GPU (torch), for benchmark
sized inputs (tens of millions of reads are produced in seconds and are already resident
in HBM, which is what bench.py times).  Same distributions as soa.gen_reads; seeded, and the
same arrays for the same seed in every process and on every box (tests/test_gpu_configs.py
checks the checksums of a 2 M-read draw against committed values): every draw is a
torch.rand / torch.randint of the seeded generator, every scatter has unique indices, the sort is stable."""

import numpy as np
import torch


def gen_reads_device(regions, names, n_reads, n_cells, seed, device, umi_len=12, read_len=91,
                     with_seq=True, bam_index=0, frac_nocell=0.05, frac_noumi=0.03,
                     contig_subset=None):
    """-> (arrays, batches): arrays = dict of whole-run device tensors sorted by (contig, pos);
    batches = list of (contig_id, start, end) slices, one per contig present.
    contig_subset: optional iterable of contig ids; reads are drawn only from genes on them."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cidx = {n: i for i, n in enumerate(names)}
    g_c = torch.tensor([cidx[r[0]] for r in regions], dtype=torch.int64, device=device)
    g_s = torch.tensor([r[1] for r in regions], dtype=torch.int64, device=device)
    g_e = torch.tensor([r[2] for r in regions], dtype=torch.int64, device=device)
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(regions) + 1) ** 0.8
    rng.shuffle(w)
    if contig_subset is not None:
        keep = np.zeros(len(names), dtype=bool)
        keep[sorted(contig_subset)] = True
        w = w * keep[np.array([cidx[r[0]] for r in regions])]
    # gene of a molecule by inverse CDF (float64, the table summed on the host): torch.multinomial gave a different draw on the
    # first call of a process than on later ones (same seed, same generator offset), so that no two boxes agreed on the workload
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    cdf = torch.tensor(cdf, dtype=torch.float64, device=device)

    def rint(lo, hi, n, dtype=torch.int64):
        return torch.randint(lo, hi, (n,), generator=g, device=device, dtype=dtype)

    def rnd(n):
        return torch.rand((n,), generator=g, device=device)

    n_mol = int(n_reads / 4.5 * 1.05) + 16
    per = rint(1, 9, n_mol)
    cs = torch.cumsum(per, 0)
    k = int(torch.searchsorted(cs, torch.tensor([n_reads], device=device)).item()) + 1
    if k > n_mol:
        raise RuntimeError("molecule pool too small")
    per = per[:k].clone()
    per[-1] -= int(per.sum().item()) - n_reads
    n_mol = k
    mg = torch.clamp(torch.searchsorted(cdf, torch.rand((n_mol,), generator=g, device=device, dtype=torch.float64), right=True), max=len(regions) - 1)
    m_cell = rint(0, n_cells, n_mol, torch.int32)
    m_cell[rnd(n_mol) < frac_nocell] = -1
    m_umi = rint(0, 1 << (2 * umi_len), n_mol) | (1 << (2 * umi_len))
    m_umi[rnd(n_mol) < frac_noumi] = -1                    # int64 -1 == XCK_UMI_NONE as uint64
    span = (g_e[mg] - g_s[mg] + 40).to(torch.float64)
    m_anchor = g_s[mg] - 41 + (torch.rand((n_mol,), generator=g, device=device, dtype=torch.float64) * span).to(torch.int64)
    rep = torch.repeat_interleave(torch.arange(n_mol, device=device), per)
    n = rep.numel()
    contig = g_c[mg][rep]
    pos = torch.clamp(m_anchor[rep] + rint(0, 200, n), min=0)
    cell = m_cell[rep]
    umi = m_umi[rep]
    u = rnd(n)
    kind = torch.full((n,), 4, dtype=torch.int8, device=device)
    kind[u < 0.98] = 3
    kind[u < 0.95] = 2
    kind[u < 0.92] = 1
    kind[u < 0.80] = 0
    mapq = torch.where(rnd(n) < 0.9, torch.tensor(255, device=device), torch.tensor([0, 1, 3], device=device)[rint(0, 3, n)]).to(torch.uint8)
    flag = torch.where(rnd(n) < 0.5, 16, 0) | torch.where(rnd(n) < 0.03, 256, 0) | torch.where(rnd(n) < 0.05, 1024, 0)
    flag = flag.to(torch.int16)                             # bit patterns < 2^15: same bytes as uint16
    L = read_len
    a = rint(10, L - 15, n)
    x = rint(1, 4, n)
    gap = rint(50, 20000, n)
    dl = rint(1, 6, n)
    sc = rint(1, 40, n)
    left = rnd(n) < 0.5
    M, I, D, N, S = 0, 1, 2, 3, 4
    w0 = torch.full((n,), (L << 4) | M, dtype=torch.int64, device=device)
    w1 = torch.zeros(n, dtype=torch.int64, device=device)
    w2 = torch.zeros(n, dtype=torch.int64, device=device)
    ncig = torch.ones(n, dtype=torch.int64, device=device)
    k1 = kind == 1; w0[k1] = (a[k1] << 4) | M; w1[k1] = (gap[k1] << 4) | N; w2[k1] = ((L - a[k1]) << 4) | M; ncig[k1] = 3
    k2 = kind == 2; w0[k2] = (a[k2] << 4) | M; w1[k2] = (x[k2] << 4) | I; w2[k2] = ((L - a[k2] - x[k2]) << 4) | M; ncig[k2] = 3
    k3 = kind == 3; w0[k3] = (a[k3] << 4) | M; w1[k3] = (dl[k3] << 4) | D; w2[k3] = ((L - a[k3]) << 4) | M; ncig[k3] = 3
    k4 = (kind == 4) & left; w0[k4] = (sc[k4] << 4) | S; w1[k4] = ((L - sc[k4]) << 4) | M; ncig[k4] = 2
    k5 = (kind == 4) & ~left; w0[k5] = ((L - sc[k5]) << 4) | M; w1[k5] = (sc[k5] << 4) | S; ncig[k5] = 2
    del a, x, gap, dl, sc, left, k1, k2, k3, k4, k5, u, kind
    order = torch.argsort(contig * (1 << 32) + pos, stable=True)
    contig, pos, cell, umi, mapq, flag = contig[order], pos[order], cell[order], umi[order], mapq[order], flag[order]
    w0, w1, w2, ncig = w0[order], w1[order], w2[order], ncig[order]
    del order, rep
    cig_off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(ncig, 0, out=cig_off[1:])
    n_cig = int(cig_off[-1].item())
    cigar = torch.zeros(n_cig, dtype=torch.int32, device=device)
    base = cig_off[:-1]
    cigar[base] = w0.to(torch.int32)
    m2 = ncig >= 2; cigar[base[m2] + 1] = w1[m2].to(torch.int32)
    m3 = ncig >= 3; cigar[base[m3] + 2] = w2[m3].to(torch.int32)
    # offsets are kept as absolute int64; device_batch() turns a slice into the u32 offsets of the C-ABI, relative to the
    # slice's own first CIGAR word / sequence byte (one batch may not exceed 4 GiB of either, see max_batch below)
    arrays = dict(pos=pos.to(torch.int32).contiguous(), flag=flag.contiguous(), mapq=mapq.contiguous(),
                  cell=cell.contiguous(), umi=umi.contiguous(), cig_off=cig_off.contiguous(),
                  cigar=cigar, n_cig=n_cig, n_reads=n, read_len=L, _rel={})
    if with_seq:
        nb = (L + 1) // 2
        seq = torch.empty(n * nb, dtype=torch.uint8, device=device)
        step = 1 << 28
        for s in range(0, n * nb, step):
            e = min(n * nb, s + step)
            b = torch.randint(0, 256, (e - s,), generator=g, device=device, dtype=torch.uint8)
            hi = torch.bitwise_left_shift(torch.ones_like(b), (b >> 2) & 3)
            lo = torch.bitwise_left_shift(torch.ones_like(b), b & 3)
            seq[s:e] = (hi << 4) | lo
        arrays["seq"] = seq
        arrays["seq_off"] = torch.arange(n + 1, dtype=torch.int64, device=device) * nb
    bounds = torch.nonzero(contig[1:] != contig[:-1]).flatten() + 1
    starts = [0] + bounds.tolist()
    ends = bounds.tolist() + [n]
    cids = contig[torch.tensor(starts, device=device)].tolist()
    max_batch = max(1, (7 << 29) // max((L + 1) // 2, 12))      # 3.5 GiB of packed bases / 12-byte CIGAR budget per batch
    batches = []
    for c, s, e in zip(cids, starts, ends):
        for s2 in range(int(s), int(e), max_batch):               # a contig's reads may be split: batches stay coordinate sorted
            batches.append((int(c), s2, min(int(e), s2 + max_batch)))
    arrays["bam_index"] = bam_index
    return arrays, batches


def device_batch(capi, arrays, contig, s, e, with_seq):
    """capi.Batch whose pointers are DEVICE addresses of the slice [s, e) (for xck_push_batch_device)."""
    import ctypes as C
    b = capi.Batch()
    b.contig = contig
    b.n_reads = e - s
    b.ordinal_base = (arrays["bam_index"] << 40) | s

    def ptr(t, off, ctype):
        return C.cast(t.data_ptr() + off * t.element_size(), C.POINTER(ctype))

    def rel(name):                                           # u32 offsets relative to the slice (cached: the tensors must outlive the batch)
        key = (name, s, e)
        if key not in arrays["_rel"]:
            o = arrays[name][s:e + 1]
            base = int(o[0].item())
            if int(o[-1].item()) - base >= (1 << 32):
                raise ValueError("batch [%d, %d) exceeds the 4 GiB u32 offset range of xck_batch.%s" % (s, e, name))
            arrays["_rel"][key] = ((o - base).to(torch.int32).contiguous(), base)      # values < 2^32: same bits as uint32
        return arrays["_rel"][key]
    b.pos = ptr(arrays["pos"], s, C.c_int32)
    b.flag = ptr(arrays["flag"], s, C.c_uint16)
    b.mapq = ptr(arrays["mapq"], s, C.c_uint8)
    b.cell = ptr(arrays["cell"], s, C.c_int32)
    b.umi = ptr(arrays["umi"], s, C.c_uint64)
    co, cbase = rel("cig_off")
    b.cig_off = ptr(co, 0, C.c_uint32)
    b.cigar = ptr(arrays["cigar"], cbase, C.c_uint32)
    if with_seq:
        so, sbase = rel("seq_off")
        b.seq_off = ptr(so, 0, C.c_uint32)
        b.seq = ptr(arrays["seq"], sbase, C.c_uint8)
    return b


def host_batch_dict(arrays, contig, s, e, with_seq):
    """Copy one slice to host numpy in the layout tests/util.batch_from_dict expects."""
    d = dict(contig=contig, n_reads=e - s, ordinal_base=(arrays["bam_index"] << 40) | s)
    for k in ("pos", "mapq", "cell"):
        d[k] = arrays[k][s:e].cpu().numpy()
    d["flag"] = arrays["flag"][s:e].cpu().numpy().view(np.uint16)
    d["umi"] = arrays["umi"][s:e].cpu().numpy().view(np.uint64)
    co = arrays["cig_off"][s:e + 1].cpu().numpy().astype(np.int64)
    d["cigar"] = arrays["cigar"][int(co[0]):int(co[-1])].cpu().numpy().view(np.uint32)
    d["cig_off"] = (co - co[0]).astype(np.uint32)
    if with_seq:
        so = arrays["seq_off"][s:e + 1].cpu().numpy().astype(np.int64)
        d["seq"] = arrays["seq"][int(so[0]):int(so[-1])].cpu().numpy()
        d["seq_off"] = (so - so[0]).astype(np.uint32)
    return d
