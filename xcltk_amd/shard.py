"""shard.py - multi-GPU sharding of the path.

Every output row depends only on reads overlapping that region and regions never span
contigs, so the unit of work is the contig: ranks own disjoint sets of contigs (hence
disjoint matrix rows), run the whole path independently, and the only exchange is one
all-gatherv of the per-rank COO blocks (sizes first, then padded triplets) - RCCL over xGMI
on GPUs (backend "nccl"), gloo in the CPU tests.  No reduction is needed.  On one node the
ranks can also write the output file together (write_mtx_sharded: one all-reduce of text sizes).
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


def plan_units(contig_weights, world, regions, contig_index, byte_profiles=None, slack=1.02):
    """Work units for `world` ranks: whole contigs, except that a contig heavier than 1 / world of all
    reads is cut at region boundaries into pieces of about equal compressed bytes (= reads; `byte_profiles[c]` = file offset
    of the first record overlapping every 16 kb window of contig c, from the .bai linear index).  A region belongs to exactly
    one piece; a piece reads the contig from the first record overlapping its first region start up to the last record that
    starts before the largest end of its regions, so reads that straddle a cut are seen by both sides.
    regions: [(chrom, start1, end1_incl, ...)]; contig_index: {chrom: contig id}.
    -> (units, owner): units = [dict(contig, regions (index array), window (beg0, end0) or None, weight)],
       owner[unit] = rank (longest-processing-time on the weights)."""
    w = np.asarray(contig_weights, dtype=np.float64)
    total = float(w.sum())
    by_c = {}
    for g, r in enumerate(regions):
        c = contig_index.get(r[0])
        if c is not None:
            by_c.setdefault(c, []).append(g)
    units = []
    for c in range(len(w)):
        idx = np.array(by_c.get(c, []), dtype=np.int64)
        prof = None if byte_profiles is None else byte_profiles.get(c)
        k = int(np.ceil(w[c] / (total / world * slack))) if total > 0 and world > 1 else 1
        if k <= 1 or prof is None or len(prof) < 2 or len(idx) < 2 or prof[-1] <= prof[0]:
            units.append(dict(contig=c, regions=idx, window=None, weight=float(w[c])))
            continue
        starts = np.array([regions[g][1] for g in idx], dtype=np.int64)
        ends = np.array([regions[g][2] for g in idx], dtype=np.int64)
        order = np.argsort(starts, kind="stable")
        idx, starts, ends = idx[order], starts[order], ends[order]
        frac = (prof.astype(np.float64) - prof[0]) / float(prof[-1] - prof[0])       # share of the contig's bytes left of every window
        f_at = lambda p: float(frac[min(max(int(p) >> 14, 0), len(frac) - 1)])
        cuts = [0]
        for j in range(1, k):
            nxt = next((i for i in range(cuts[-1] + 1, len(idx)) if f_at(starts[i] - 1) >= j / k), None)
            if nxt is None:
                break
            cuts.append(nxt)
        cuts.append(len(idx))
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b <= a:
                continue
            beg0 = 0 if a == 0 else int(starts[a] - 1)
            end0 = 0 if b == len(idx) else int(ends[a:b].max())                     # 0 = open end
            share = (f_at(end0) if end0 else 1.0) - f_at(beg0)
            units.append(dict(contig=c, regions=idx[a:b], window=(beg0, end0), weight=float(w[c]) * max(share, 1e-6)))
    bins = lpt_assign([u["weight"] for u in units], world)
    owner = np.zeros(len(units), dtype=np.int32)
    for r, items in enumerate(bins):
        for i in items:
            owner[i] = r
    return units, owner


def merge_row_blocks(blocks, row_owner):
    """Concatenate per-rank sparse blocks into one (row, col, val) sorted by (row, col) without sorting.
    blocks[r] = int32 array [row | col | val] (3 * nnz_r) of rank r, itself sorted by (row, col) and holding only rows
    owned by r; row_owner[row] = owning rank.  Ranks own disjoint rows, so the result is the sequence of maximal runs of
    rows with one owner, each cut out of its owner's block with two binary searches (regions of a contig are normally
    adjacent in the region file: a few dozen runs)."""
    row_owner = np.asarray(row_owner)
    n = len(row_owner)
    views = []
    for b in blocks:
        b = np.asarray(b, dtype=np.int32)
        z = len(b) // 3
        views.append((b[:z], b[z:2 * z], b[2 * z:3 * z]))
    if n == 0:
        z = np.zeros(0, dtype=np.int32)
        return z, z.copy(), z.copy()
    change = np.flatnonzero(np.diff(row_owner)) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [n]])
    owners = row_owner[starts]
    cuts = {}
    for r in set(int(x) for x in owners):
        if 0 <= r < len(views):
            sel = owners == r
            cuts[r] = (np.searchsorted(views[r][0], starts[sel], "left"), np.searchsorted(views[r][0], ends[sel], "left"))
    seq, pos = [], {r: 0 for r in cuts}
    for r in owners.tolist():
        if r in cuts:
            i = pos[r]; pos[r] += 1
            lo, hi = int(cuts[r][0][i]), int(cuts[r][1][i])
            if hi > lo:
                seq.append((r, lo, hi))
    total = sum(hi - lo for _, lo, hi in seq)
    assert total == sum(len(v[0]) for v in views), "a rank holds rows it does not own"
    out = tuple(np.empty(total, dtype=np.int32) for _ in range(3))
    o = 0
    for r, lo, hi in seq:
        for j in range(3):
            out[j][o:o + hi - lo] = views[r][j][lo:hi]
        o += hi - lo
    return out


class _DevArray(object):
    """Minimal __cuda_array_interface__ holder so torch can wrap an engine-owned device buffer."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def owner_runs(row_owner):
    """Maximal runs of rows with one owner: (starts, ends, owners)."""
    row_owner = np.asarray(row_owner)
    n = len(row_owner)
    if n == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z, np.zeros(0, dtype=row_owner.dtype)
    change = np.flatnonzero(np.diff(row_owner)) + 1
    starts = np.concatenate([[0], change]).astype(np.int64)
    ends = np.concatenate([change, [n]]).astype(np.int64)
    return starts, ends, row_owner[starts]


def write_mtx_sharded(path, coo, row_map, row_owner, n_rows_out, n_cols, rank, all_reduce_sum, barrier, lib=None):
    """ONE MatrixMarket file written by all ranks of a node together, each rank the lines of the rows it owns - instead of
    gathering the triplets on rank 0 and formatting ~10^8 lines there (the serial tail of a multi-GPU run: gather + merge +
    write were ~1 s of a 2 s pass at 8 ranks).  The only exchange is ONE all-reduce of the text sizes.
    coo: this rank's (row, col, val), sorted by (row, col), rows owned by this rank only; row_owner[row] = rank;
    all_reduce_sum(int64 numpy array) -> summed array on every rank; barrier() -> None.  Every rank must see `path`
    (same node / shared file system).  Byte-identical to xck_write_mtx of the merged matrix; returns the total line count."""
    import ctypes as C
    from . import capi
    lib = lib or capi.load()
    row, col, val = (np.ascontiguousarray(a, dtype=np.int32) for a in coo)
    rm = np.ascontiguousarray(row_map, dtype=np.int32)
    starts, ends, owners = owner_runs(row_owner)
    n_runs = len(starts)
    mine = np.flatnonzero(owners == rank)
    lo = np.searchsorted(row, starts[mine], "left")
    hi = np.searchsorted(row, ends[mine], "left")
    if int((hi - lo).sum()) != len(row):
        raise ValueError("write_mtx_sharded: this rank holds rows it does not own")

    def piece(a, b):
        c = capi.Coo()
        c.nnz = int(b - a)
        c.row, c.col, c.val = (capi.np_ptr(x[a:b], C.c_int32) for x in (row, col, val))
        return c
    sizes = np.zeros(2 * n_runs, dtype=np.int64)                    # [bytes per run | lines per run], zero for other ranks' runs
    for i, a, b in zip(mine.tolist(), lo.tolist(), hi.tolist()):
        nb, nl = C.c_int64(0), C.c_int64(0)
        if lib.xck_mtx_part_size(C.byref(piece(a, b)), capi.np_ptr(rm, C.c_int32), C.byref(nb), C.byref(nl)) != 0:
            raise RuntimeError("xck_mtx_part_size failed")
        sizes[i], sizes[n_runs + i] = nb.value, nl.value
    sizes = np.asarray(all_reduce_sum(sizes), dtype=np.int64)
    header = ("%%%%MatrixMarket matrix coordinate integer general\n%%%%\n%d\t%d\t%d\n" % (n_rows_out, n_cols, int(sizes[n_runs:].sum()))).encode()
    offs = len(header) + np.concatenate([[0], np.cumsum(sizes[:n_runs])])
    if rank == 0:
        with open(path, "wb") as fp:
            fp.write(header)
            fp.truncate(int(offs[-1]))
    barrier()                                                        # the file exists with its final size
    for i, a, b in zip(mine.tolist(), lo.tolist(), hi.tolist()):
        if b > a and lib.xck_write_mtx_part(path.encode(), int(offs[i]), C.byref(piece(a, b)), capi.np_ptr(rm, C.c_int32)) != 0:
            raise RuntimeError("xck_write_mtx_part failed for %s" % path)
    barrier()
    return int(sizes[n_runs:].sum())


class BlockGatherer(object):
    """all-gatherv of the per-rank sparse blocks to the writer rank (rank 0), GPU to GPU.

    Every rank contributes, for each matrix, its [row|col|val] int32 block that is still resident in
    HBM after xck_finish (Engine.result_device()).  One tiny all-gather exchanges the sizes, then ONE
    gather moves all matrices (each padded to its largest block) to rank 0 - RCCL over xGMI with the
    nccl backend.  start() only enqueues the exchange (async_op) so that the caller can overlap it with
    the next pass; wait() returns, on rank 0, {name: [per-rank int32 tensors in rank order]}.
    With contiguous per-rank row ranges the rank-order concatenation is already the (row, col) order.
    """

    def __init__(self, world, rank, device, names=("count", "ad", "dp", "oth"), backend_is_nccl=True):
        self.world, self.rank, self.device, self.names = world, rank, device, tuple(names)
        self.nccl = backend_is_nccl
        self._pending = None

    def start(self, blocks):
        """blocks: {name: (device_ptr, nnz)}."""
        import torch
        import torch.distributed as dist
        self.wait()
        dev = self.device if self.nccl else "cpu"
        mine = torch.tensor([blocks[k][1] for k in self.names], dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(all_sizes, mine)
        sizes = torch.stack(all_sizes).cpu().tolist()                 # [rank][matrix]
        pad = [max(max(sizes[r][j] for r in range(self.world)), 1) * 3 for j in range(len(self.names))]
        buf = torch.zeros(sum(pad), dtype=torch.int32, device=self.device)
        off = 0
        for j, k in enumerate(self.names):
            ptr, nnz = blocks[k]
            if nnz:                                                   # device-to-device copy out of the engine workspace
                buf[off:off + 3 * nnz] = torch.as_tensor(_DevArray(ptr, 3 * nnz), device=self.device)
            off += pad[j]
        # the engine re-uses its workspace on the next pass: make sure the copy out of it has really run
        torch.cuda.current_stream(self.device).synchronize()
        if not self.nccl:
            buf = buf.cpu()
        outs = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
        work = dist.gather(buf, outs, dst=0, async_op=True)
        self._pending = (work, outs, sizes, pad, buf)
        return sizes

    def wait(self):
        if self._pending is None:
            return None
        work, outs, sizes, pad, buf = self._pending
        self._pending = None
        work.wait()
        if self.rank != 0:
            return None
        res, off = {}, 0
        for j, k in enumerate(self.names):
            res[k] = [outs[r][off:off + 3 * sizes[r][j]] for r in range(self.world)]
            off += pad[j]
        return res
