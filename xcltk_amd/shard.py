"""shard.py - multi-GPU sharding of the path.

Every output row depends only on reads overlapping that region and regions never span
contigs, so the unit of work is the contig: ranks own disjoint sets of contigs (hence
disjoint matrix rows), run the whole path independently, and the only exchange is one
all-gatherv of the per-rank COO blocks (sizes first, then padded triplets) - RCCL over xGMI
on GPUs (backend "nccl"), gloo in the CPU tests.  No reduction is needed.
"""

import numpy as np


def lpt_assign(weights, n_bins):
    """Longest-processing-time assignment of items to bins; returns list of index lists."""
    bins = [[] for _ in range(n_bins)]
    load = [0.0] * n_bins
    for i in sorted(range(len(weights)), key=lambda i: (-weights[i], i)):
        b = min(range(n_bins), key=lambda j: (load[j], j))
        bins[b].append(i)
        load[b] += weights[i]
    return [sorted(b) for b in bins]


def contig_owner(contig_names, weights, world):
    """contig id -> owning rank (LPT on weights such as contig length or BAI read counts)."""
    owner = np.zeros(len(contig_names), dtype=np.int32)
    for r, items in enumerate(lpt_assign(list(weights), world)):
        for i in items:
            owner[i] = r
    return owner


def gather_coo(coo, world, device="cpu", group=None):
    """All-gatherv of one sparse block.  coo = (row, col, val) int32 numpy arrays of this
    rank; returns the concatenation over ranks sorted by (row, col) - identical to what a
    single process would have produced because ranks own disjoint rows."""
    import torch
    import torch.distributed as dist
    n = torch.tensor([len(coo[0])], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros((3, mx), dtype=torch.int32, device=device)
    if len(coo[0]):
        buf[:, :len(coo[0])] = torch.from_numpy(np.stack([np.asarray(c, dtype=np.int32) for c in coo])).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    cat = torch.cat([o[:, :s] for o, s in zip(out, sizes)], dim=1)
    key = cat[0].to(torch.int64) * (1 << 31) + cat[1].to(torch.int64)
    cat = cat[:, torch.argsort(key)].cpu().numpy()
    return cat[0].copy(), cat[1].copy(), cat[2].copy()
