"""fc_common.py - host logic shared by `basefc` and the BAF feature counting front-ends.

The reference duplicates this logic in xcltk/rdr/fc/main.py:305-431 and
xcltk/baf/fc/main.py:298-499 (prepare_config) and in the two utils.py (loaders, merge_mtx).
Here it lives once; behaviour (defaults, error messages, file formats, return codes) follows
those lines.  The compute itself is delegated to the HIP engine (engine.Engine).
"""

import ctypes as C
import os
import sys
import time
from logging import error, info
from logging import warning as warn

import numpy as np

from .capi import XCK_MODE_BASEFC
from .engine import Engine
from .snptable import SnpTable
from .utils.grange import format_chrom
from .utils.zfile import zopen

EXCL_FLAG_UMI = 772       # reference DefaultConfig, rdr/fc/config.py:108-109
EXCL_FLAG_XUMI = 1796


# ----------------------------------------------------------------------------- input resolution
def resolve_inputs(conf):
    """BAM list, barcodes / sample ids, tags, default exclude flag.  Returns 0 or -1 and
    logs the reference's messages (rdr/fc/main.py:323-373,401-429)."""
    if conf.sam_fn:
        if conf.sam_list_fn:
            error("should not specify 'sam_fn' and 'sam_list_fn' together.")
            return -1
        conf.sam_fn_list = conf.sam_fn.split(",")
    else:
        if not conf.sam_list_fn:
            error("one of 'sam_fn' and 'sam_list_fn' should be specified.")
            return -1
        with open(conf.sam_list_fn, "r") as fp:
            conf.sam_fn_list = [x.rstrip() for x in fp.readlines()]
    for fn in conf.sam_fn_list:
        if not os.path.isfile(fn):
            error("sam file '%s' does not exist." % fn)
            return -1

    if conf.barcode_fn:
        conf.sample_ids = None
        if conf.sample_id_str or conf.sample_id_fn:
            error("should not specify barcodes and sample IDs together.")
            return -1
        if not os.path.isfile(conf.barcode_fn):
            error("barcode file '%s' does not exist." % conf.barcode_fn)
            return -1
        with zopen(conf.barcode_fn, "rt") as fp:
            conf.barcodes = sorted(x.strip() for x in fp)       # column j = j-th SORTED barcode
        if len(set(conf.barcodes)) != len(conf.barcodes):
            error("duplicate barcodes!")
            return -1
    else:
        conf.barcodes = None
        if conf.sample_id_str and conf.sample_id_fn:
            error("should not specify 'sample_id_str' and 'sample_fn' together.")
            return -1
        if conf.sample_id_str:
            conf.sample_ids = conf.sample_id_str.split(",")
        elif conf.sample_id_fn:
            with zopen(conf.sample_id_fn, "rt") as fp:
                conf.sample_ids = [x.strip() for x in fp]
        else:
            warn("use default sample IDs ...")
            conf.sample_ids = ["Sample%d" % i for i in range(len(conf.sam_fn_list))]
        if len(conf.sample_ids) != len(conf.sam_fn_list):
            error("numbers of sam files and sample IDs are different.")
            return -1
    conf.samples = conf.barcodes if conf.barcodes else conf.sample_ids
    return 0


def resolve_tags(conf):
    """'None' strings, barcode/tag consistency, UMI 'Auto', default exclude flag."""
    if conf.cell_tag and conf.cell_tag.upper() == "NONE":
        conf.cell_tag = None
    if (not conf.cell_tag) ^ (not conf.barcodes):
        error("should not specify cell_tag or barcodes alone.")
        return -1
    if conf.umi_tag:
        up = conf.umi_tag.upper()
        if up == "AUTO":
            conf.umi_tag = None if conf.barcodes is None else conf.defaults.UMI_TAG_BC
        elif up == "NONE":
            conf.umi_tag = None
    if conf.excl_flag < 0:
        conf.excl_flag = EXCL_FLAG_UMI if conf.use_umi() else EXCL_FLAG_XUMI
    for tag, what in ((conf.cell_tag, "cell"), (conf.umi_tag, "UMI")):
        if tag and len(tag) != 2:
            error("%s tag '%s' is not a 2-character BAM tag." % (what, tag))
            return -1
    return 0


def load_region_from_txt(fn, sep="\t", verbose=False):
    """Header-free TSV: chrom, start, end (1-based inclusive), name -> list of
    (chrom_stripped, start, end_incl, name) in file order, or None (rdr/fc/utils.py:10-45)."""
    func = "load_region_from_txt"
    out = []
    if verbose:
        sys.stderr.write("[I::%s] start to load regions from file '%s' ...\n" % (func, fn))
    with zopen(fn, "rt") as fp:
        for nl, line in enumerate(fp, 1):
            parts = line.rstrip().split(sep)
            if len(parts) < 4:
                if verbose:
                    sys.stderr.write("[E::%s] too few columns of line %d.\n" % (func, nl))
                return None
            out.append((format_chrom(parts[0]), int(parts[1]), int(parts[2]), parts[3]))
    return out


_BASES = frozenset("ACGTN")


def _snp_from_fields(chrom, pos, ref, alt, a1, a2):
    ref, alt = ref.upper(), alt.upper()
    if len(ref) != 1 or ref not in "ACGTN":
        return "invalid REF base"
    if len(alt) != 1 or alt not in "ACGTN":
        return "invalid ALT base"
    if not ((a1 == "0" and a2 == "1") or (a1 == "1" and a2 == "0")):
        return "invalid GT"
    return (format_chrom(chrom), int(pos), ref, alt, int(a1), int(a2))


_REJ_TSV = {1: "too few columns of", 2: "invalid REF base of", 3: "invalid ALT base of", 7: "invalid GT of"}
_REJ_VCF = {1: "too few columns of", 2: "invalid REF base", 3: "invalid ALT base", 4: "GT not in", 5: "len(fields) != len(values) in",
            6: "invalid delimiter of", 7: "invalid GT of"}


def _load_snps_native(fn, is_vcf, verbose=False, func=""):
    """The library's parser (csrc/snptext.cpp): same list as the loops below for plain-ASCII files, about 4x sooner for a
    million SNPs; None when the file is outside what it reproduces exactly (or XCK_PY_LOADERS=1): the caller then loops.
    verbose: the per-line warnings of the loops, printed from the parser's list of rejected lines."""
    if os.environ.get("XCK_PY_LOADERS") == "1":
        return None
    from . import capi
    lib = capi.load()
    t = C.POINTER(capi.SnpText)()
    rc = lib.xck_parse_snp_text(fn.encode(), 1 if is_vcf else 0, C.byref(t))
    if rc != 0:
        return None                              # 1 = not eligible; I/O errors surface from the generic loader with Python's own message
    try:
        v = t.contents
        n, nr = int(v.n), int(v.n_rejected)
        if verbose and nr:
            text = _REJ_VCF if is_vcf else _REJ_TSV
            lines = np.ctypeslib.as_array(v.rej_line, shape=(nr,)).tolist()
            codes = np.ctypeslib.as_array(v.rej_code, shape=(nr,)).tolist()
            sys.stderr.write("".join("[W::%s] %s line %d.\n" % (func, text[c], l) for l, c in zip(lines, codes)))
        if n == 0:
            return []
        names = [v.chroms[i].decode("ascii") for i in range(v.n_chroms)]
        col = lambda p, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True)     # the library frees its own arrays
        return SnpTable(names, col(v.chrom_id, np.int32), col(v.pos, np.int64),
                        np.frombuffer(C.string_at(v.ref, n), dtype=np.uint8), np.frombuffer(C.string_at(v.alt, n), dtype=np.uint8),
                        col(v.ref_hap, np.int8), col(v.alt_hap, np.int8))
    finally:
        lib.xck_free_snp_text(t)


def load_snp_from_tsv(fn, verbose=False):
    """TSV with header: chrom pos ref alt ref_hap alt_hap (baf/fc/utils.py:51-110)."""
    func = "load_snp_from_tsv"
    snps = []
    if verbose:
        sys.stderr.write("[I::%s] start to load SNPs from tsv '%s' ...\n" % (func, fn))
    native = _load_snps_native(fn, False, verbose, func)
    if native is not None:
        return native
    with zopen(fn, "rt") as fp:
        for nl, line in enumerate(fp, 1):
            if nl == 1:
                continue
            parts = line.rstrip().split("\t")
            if len(parts) < 6:
                if verbose:
                    sys.stderr.write("[W::%s] too few columns of line %d.\n" % (func, nl))
                continue
            # inlined common case of _snp_from_fields (a million-SNP panel is a million calls)
            ref, alt, a1, a2 = parts[2], parts[3], parts[4], parts[5]
            if ref in _BASES and alt in _BASES and ((a1 == "0" and a2 == "1") or (a1 == "1" and a2 == "0")):
                snps.append((format_chrom(parts[0]), int(parts[1]), ref, alt, int(a1), int(a2)))
                continue
            s = _snp_from_fields(parts[0], parts[1], parts[2], parts[3], parts[4], parts[5])
            if isinstance(s, str):
                if verbose:
                    sys.stderr.write("[W::%s] %s of line %d.\n" % (func, s, nl))
                continue
            snps.append(s)
    return snps


def load_snp_from_vcf(fn, verbose=False):
    """Phased VCF, first sample column; GT must be 0|1, 1|0, 0/1 or 1/0 (baf/fc/utils.py:114-193)."""
    func = "load_snp_from_vcf"
    snps = []
    if verbose:
        sys.stderr.write("[I::%s] start to load SNPs from vcf '%s' ...\n" % (func, fn))
    native = _load_snps_native(fn, True, verbose, func)
    if native is not None:
        return native
    with zopen(fn, "rt") as fp:
        for nl, line in enumerate(fp, 1):
            if line[0] in ("#", "\n"):
                continue
            parts = line.rstrip().split("\t")
            if len(parts) < 10:
                if verbose:
                    sys.stderr.write("[W::%s] too few columns of line %d.\n" % (func, nl))
                continue
            msg = None
            fields = parts[8].split(":")
            values = parts[9].split(":")
            ref, alt = parts[3].upper(), parts[4].upper()
            if len(ref) != 1 or ref not in "ACGTN":
                msg = "invalid REF base"
            elif len(alt) != 1 or alt not in "ACGTN":
                msg = "invalid ALT base"
            elif "GT" not in fields:
                msg = "GT not in"
            elif len(values) != len(fields):
                msg = "len(fields) != len(values) in"
            else:
                gt = values[fields.index("GT")]
                sepc = "|" if "|" in gt else ("/" if "/" in gt else None)
                if sepc is None:
                    msg = "invalid delimiter of"
                else:
                    a1, a2 = gt.split(sepc)[:2]
                    s = _snp_from_fields(parts[0], parts[1], ref, alt, a1, a2)
                    if isinstance(s, str):
                        msg = s + " of"
                    else:
                        snps.append(s)
            if msg and verbose:
                sys.stderr.write("[W::%s] %s line %d.\n" % (func, msg, nl))
    return snps


def is_vcf_name(fn):
    return fn.endswith(".vcf") or fn.endswith(".vcf.gz") or fn.endswith(".vcf.bgz")


# ----------------------------------------------------------------------------- outputs
def contig_table(regions, snps=()):
    names, seen = [], set()
    for ch in [r[0] for r in regions] + (snps.chroms() if isinstance(snps, SnpTable) else [s[0] for s in snps]):
        if ch not in seen:
            seen.add(ch)
            names.append(ch)
    return names


def row_map_all(n):
    return np.arange(1, n + 1, dtype=np.int32)


def row_map_from_rows(n, *row_arrays):
    """Rows are numbered over the regions that produced at least one output line
    (k_reg bookkeeping, rdr/fc/core.py:118-124 and baf/fc/core.py:101-113)."""
    keep = np.zeros(n, dtype=bool)
    for r in row_arrays:
        keep[r] = True
    rm = np.zeros(n, dtype=np.int32)
    rm[keep] = np.arange(1, int(keep.sum()) + 1, dtype=np.int32)
    return rm


def write_region_tsv(path, regions, rm):
    with open(path, "w") as fp:
        fp.write("".join("%s\t%d\t%d\t%s\n" % (ch, s, e, name)
                         for (ch, s, e, name), r in zip(regions, rm) if r > 0))


def is_writer_rank():
    """True in a single-process run and on rank 0 of a multi-GPU run (the only rank that writes files)."""
    return int(os.environ.get("RANK", "0")) == 0


def write_samples(path, samples):
    with open(path, "w") as fp:
        fp.write("".join(smp + "\n" for smp in samples))


def output_row_map(dist, n, all_reg, *row_arrays):
    """Output row of every region (0 = no row).  With sharded output the regions that wrote a line are known only jointly:
    one all-reduce of a presence vector (collective call)."""
    if all_reg:
        return row_map_all(n)
    if dist is not None and dist.active and dist.sharded_output:
        keep = np.zeros(n, dtype=np.int32)
        for r in row_arrays:
            keep[r] = 1
        keep = dist.all_reduce_np(keep, "max") > 0
        rm = np.zeros(n, dtype=np.int32)
        rm[keep] = np.arange(1, int(keep.sum()) + 1, dtype=np.int32)
        return rm
    return row_map_from_rows(n, *row_arrays)


def write_mtx(eng, dist, path, coo_k, rm, n_rows_out):
    """One matrix file: written by this process, or - sharded output - by all ranks together, each the lines of its own rows
    (shard.write_mtx_sharded; collective call)."""
    if dist is not None and dist.active and dist.sharded_output:
        from .shard import write_mtx_sharded
        write_mtx_sharded(path, coo_k, rm, dist.row_owner, n_rows_out, eng.n_cells, dist.rank, dist.all_reduce_np, dist.barrier, eng.lib)
    else:
        eng.write_mtx_arrays(path, coo_k, rm, n_rows_out)


# ----------------------------------------------------------------------------- engine driver
def make_engine(conf, mode, regions, snps=(), device=None, **extra):
    """Build the per-GPU engine from a resolved Config."""
    names = contig_table(regions, snps)
    if device is None:
        device = int(os.environ.get("XCK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    kw = dict(min_mapq=conf.min_mapq, min_len=conf.min_len, incl_flag=conf.incl_flag,
              excl_flag=conf.excl_flag, no_orphan=conf.no_orphan, n_threads=max(0, int(conf.nproc)))
    if mode == XCK_MODE_BASEFC:
        kw["min_include"] = conf.min_include
    else:
        kw.update(min_count=conf.min_count, min_maf=conf.min_maf, no_dup_hap=conf.no_dup_hap)
    kw.update(extra)
    dist = Dist()
    if dist.active:                                   # multi-GPU: this rank counts the regions of its contigs (or contig pieces)
        dist.plan(conf, regions, snps)
        kw["region_mask"] = dist.region_mask
    eng = Engine(mode, names, regions, len(conf.samples), snps=snps,
                 barcodes=conf.barcodes if conf.use_barcodes() else None,
                 cell_tag=conf.cell_tag, umi_tag=conf.umi_tag, device=device, **kw)
    eng.dist = dist
    return eng


class Dist(object):
    """Multi-GPU context (one process per GPU).  With WORLD_SIZE > 1 every
    rank builds the full tables, streams only the contigs it owns (LPT on .bai record counts,
    contig lengths as fallback; an over-weight contig is cut at region boundaries).  The results meet in one of two ways:
    every rank writes the lines of its own rows into the shared output files (default: one all-reduce of text sizes per
    file), or - XCK_DIST_GATHER=1 - the per-rank sparse blocks are concatenated on rank 0 by one all-gatherv.  RCCL over
    xGMI with the nccl backend; gloo when XCK_DIST_BACKEND=gloo."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        self.device = "cpu"
        # XCK_DIST_FORCE=1: take the multi-rank code path with ONE rank as well (communicator, collectives, gather and the sharded
        # writer all run, over a world of one) - how the one-GPU test box exercises the RCCL backend
        self.forced = self.world == 1 and os.environ.get("XCK_DIST_FORCE", "0") not in ("", "0")
        # how the ranks' results meet: every rank writes the lines of its own rows into the shared output files (default; the
        # ranks of one node see the same directory), or XCK_DIST_GATHER=1: the sparse blocks are gathered on rank 0, which writes
        self.sharded_output = False
        if self.world > 1 or self.forced:
            import torch
            import torch.distributed as dist
            self.backend = os.environ.get("XCK_DIST_BACKEND", "nccl")
            if not dist.is_initialized():
                if self.forced:
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", "29533")
                kw = dict(rank=self.rank, world_size=self.world) if self.forced else {}
                if self.backend == "nccl":
                    torch.cuda.set_device(self.local_rank)
                    dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank), **kw)
                else:
                    dist.init_process_group(self.backend, **kw)
            if self.backend == "nccl":
                self.device = "cuda:%d" % self.local_rank
            self.sharded_output = os.environ.get("XCK_DIST_GATHER", "0") in ("", "0")
            # Ranks that SHARE a GPU (fewer devices than ranks on this node: test rehearsals, oversubscribed nodes) keep the BGZF inflate on the
            # host: the device decoder's launches run for tens of milliseconds, and the GPU time-slices between processes at that grain - the
            # other rank's join kernels then wait behind them (measured on one GPU with two ranks: 40 M reads/s host-only, 4.8 M with the
            # device share).  One GPU per rank - the deployment - is not affected.  (The engine reads the knob at xck_create.)
            try:
                # (ranks on THIS node: torchrun says so; without it, a rendezvous on this host means all ranks are here - otherwise unknown, no guess)
                one_node = os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1")
                local_world = int(os.environ["LOCAL_WORLD_SIZE"]) if "LOCAL_WORLD_SIZE" in os.environ else (self.world if one_node else 0)
                if self.world > 1 and torch.cuda.device_count() < local_world:
                    os.environ.setdefault("XCK_GPU_INFLATE", "0")
            except Exception:
                pass

    @property
    def active(self):
        return self.world > 1 or self.forced

    def plan(self, conf, regions, snps=(), names=None):
        """Who counts what (identical on every rank): contigs by longest-processing-time on the .bai record counts; a contig
        that outweighs 1 / world of the reads is cut at region boundaries (shard.plan_units).  Sets contig_mask (contigs this
        rank streams), windows ({contig: (beg0, end0)} for the cut ones), region_mask (regions this rank counts) and
        row_owner (rank per region, for stitching the gathered blocks)."""
        from .shard import plan_units
        if names is None:
            names = contig_table(regions, snps)
        probe = Engine(XCK_MODE_BASEFC, names, [], 1, decode_only=True)
        try:
            weights = np.zeros(len(names), dtype=np.float64)
            for fn in conf.sam_fn_list:
                c = probe.contig_record_counts(fn)
                if c is None:                           # no index: every contig weighs the same
                    weights[:] = 1.0
                    break
                weights += c
            heavy = [c for c in range(len(names)) if weights.sum() > 0 and weights[c] > weights.sum() / self.world * 1.02]
            profiles = {}
            for c in heavy:                            # byte profile of the first BAM that has one (cuts are positions, valid for every BAM)
                for fn in conf.sam_fn_list:
                    pr = probe.contig_byte_profile(fn, c)
                    if pr is not None:
                        profiles[c] = pr
                        break
        finally:
            probe.close()
        cidx = {n: i for i, n in enumerate(names)}
        units, owner = plan_units(weights + 1e-9, self.world, regions, cidx, profiles)
        self.units, self.unit_owner, self.contig_weights = units, owner, weights
        self.contig_mask = np.zeros(len(names), dtype=bool)
        self.region_mask = np.zeros(len(regions), dtype=bool)
        self.row_owner = np.full(len(regions), -1, dtype=np.int32)
        self.windows = {}
        for u, r in zip(units, owner.tolist()):
            self.row_owner[u["regions"]] = r
            if r != self.rank:
                continue
            self.contig_mask[u["contig"]] = True
            self.region_mask[u["regions"]] = True
            if u["window"] is not None:
                if u["contig"] in self.windows:          # two pieces of one contig on this rank: their hull
                    b0, e0 = self.windows[u["contig"]]
                    nb, ne = u["window"]
                    self.windows[u["contig"]] = (min(b0, nb), 0 if (e0 == 0 or ne == 0) else max(e0, ne))
                else:
                    self.windows[u["contig"]] = u["window"]
        # a rank that holds pieces next to whole contigs: whole contigs have no window (decoded entirely)
        return self

    # -- the collectives the sharded writer needs (tiny tensors; RCCL with the nccl backend, gloo in the shared-GPU tests)
    def all_reduce_np(self, x, op="sum"):
        import torch
        import torch.distributed as td
        t = torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        td.all_reduce(t, op=td.ReduceOp.MAX if op == "max" else td.ReduceOp.SUM)
        return t.cpu().numpy()

    def barrier(self):
        import torch.distributed as td
        td.barrier()

    def gather(self, eng, regions):
        """all-gatherv of the per-rank sparse blocks to rank 0, straight from HBM (xck_get_result_device): one
        all-gather of sizes + one padded gather (RCCL over xGMI; host tensors over gloo).  Ranks own whole contigs,
        hence disjoint rows: rank 0 stitches the row runs together without sorting.  -> coo dict on rank 0, else None."""
        import torch
        from .shard import BlockGatherer, merge_row_blocks
        nccl = self.backend == "nccl"
        dev = torch.device(self.device) if nccl else torch.device("cuda", int(os.environ.get("XCK_DEVICE", self.local_rank if torch.cuda.device_count() > self.local_rank else 0)))
        blocks = eng.result_device()
        g = BlockGatherer(self.world, self.rank, dev, names=tuple(blocks), backend_is_nccl=nccl)
        g.start(blocks)
        res = g.wait()
        if self.rank != 0:
            return None
        return {k: merge_row_blocks([b.cpu().numpy() for b in res[k]], self.row_owner) for k in res}


def stream_bams(eng, conf, log_prefix="[engine]", contig_mask=None, windows=None):
    """One streaming pass over every BAM, in list order (= the reference's fetch order)."""
    t0 = time.time()
    n_tot = 0
    def per_file(i, fn, n):
        if conf.debug > 0:
            info("%s %s: %d records" % (log_prefix, fn, n))
    # (the next file reads ahead while this one is parsed and joined: Engine.ingest_bams)
    n_tot = eng.ingest_bams(list(conf.sam_fn_list), contig_mask=contig_mask, use_index=contig_mask is not None, windows=windows or None, on_file=per_file)
    dt = max(time.time() - t0, 1e-9)
    info("%s %d BAM record(s) decoded and joined in %.2fs (%.0f reads/s)" % (log_prefix, n_tot, dt, n_tot / dt))
    return n_tot


def make_and_count(conf, mode, regions, snps=(), log_prefix="[engine]", gather=None, **extra):
    """Build the engine, stream every BAM (this rank's contigs when running multi-GPU), fold, gather.  Keys that are not ACGT strings (IUPAC UMIs, integer tags, read names in UMI-less runs) are
    interned on the host and take one id each; when the ids outgrow the UMI field of a 64-bit key the decoder stops with
    XCK_E_CAPACITY - the run is then repeated once with 128-bit keys (XCK_F_FORCE_KEY128), on every rank of a multi-GPU run.
    Multi-GPU: by default every rank gets ITS OWN rows back (dist.sharded_output) and the ranks write the files together
    (output_row_map / write_mtx below are then collective calls); gather=True, or XCK_DIST_GATHER=1, gathers the blocks on
    rank 0 instead (coo is None on the other ranks).
    Returns (engine - the caller closes it, coo or None, Dist)."""
    from .capi import XCK_E_CAPACITY, XCK_F_FORCE_KEY128
    from .engine import XckError
    flags = int(extra.pop("flags", 0))
    while True:
        eng = make_engine(conf, mode, regions, snps, flags=flags, **extra)
        dist, overflow, failure = eng.dist, 0, None
        try:
            stream_bams(eng, conf, log_prefix, dist.contig_mask if dist.active else None, dist.windows if dist.active else None)
        except XckError as e:
            failure = e
            overflow = int(getattr(e, "code", 0) == XCK_E_CAPACITY and not flags & XCK_F_FORCE_KEY128)
        except BaseException:
            eng.close()
            raise
        if dist.active:                                   # the ranks agree: one overflow sends all of them round again
            import torch
            import torch.distributed as td
            t = torch.tensor([overflow, int(failure is not None)], dtype=torch.int64, device=dist.device)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            overflow, any_failure = int(t[0]), int(t[1])
            if any_failure and not overflow and failure is None:
                failure = XckError("another rank failed")
        if overflow:
            eng.close()
            warn("%s key ids exceed the 64-bit key layout; counting again with 128-bit keys." % log_prefix)
            flags |= XCK_F_FORCE_KEY128
            continue
        if failure is not None:
            eng.close()
            raise failure
        try:
            coo = eng.finish(copy=False)
            if gather is not None:
                dist.sharded_output = dist.active and not gather
            if dist.active and not dist.sharded_output:
                coo = dist.gather(eng, conf.reg_list)              # rank 0: merged matrices; other ranks: None
        except BaseException:
            eng.close()
            raise
        return eng, coo, dist
