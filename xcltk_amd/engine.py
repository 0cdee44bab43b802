"""engine.py - Python handle on the HIP counting engine (libxck.so) through the C-ABI.

One `Engine` drives one GPU.  It owns the region / SNP tables, the decoder settings
(barcode list, tag names) and the device-side hit accumulators; BAM files are streamed
through it once (`ingest_bam`) or record batches are pushed directly (`push`), and
`finish()` returns the sparse matrices as COO triplets.

There is no CPU compute path here: constructing an Engine without a usable HIP device
raises (the only exception is `decode_only=True`, which gives access to the host BAM
decoder alone and refuses push/finish).
"""

import ctypes as C

import numpy as np

from . import capi
from .snptable import SnpTable
from .capi import XCK_MODE_BAF, XCK_MODE_BASEFC


class XckError(RuntimeError):
    """Engine / decoder failure; `code` holds the C-ABI error code (include/xck.h XCK_E_*) when there is one."""
    def __init__(self, msg, code=0):
        RuntimeError.__init__(self, msg)
        self.code = code


def resolve_contigs(bam_refs, contig_names):
    """tid -> engine contig id, with the reference's chr-prefix tolerance
    (sam_fetch, xcltk/utils/sam.py:105-118: try the name, then the name with 'chr'
    added / removed)."""
    tid_of = {}
    for i, n in enumerate(bam_refs):
        tid_of.setdefault(n, i)
    t2c = np.full(len(bam_refs), -1, dtype=np.int32)
    for ci, x in enumerate(contig_names):
        if x in tid_of:
            t2c[tid_of[x]] = ci
        else:
            y = x[3:] if x.startswith("chr") else "chr" + x
            if y in tid_of:
                t2c[tid_of[y]] = ci
    return t2c


class Engine(object):
    def __init__(self, mode, contig_names, regions, n_cells, snps=(), barcodes=None,
                 cell_tag=None, umi_tag=None, device=0, min_mapq=20, min_len=30,
                 incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9, min_count=1,
                 min_maf=0, no_dup_hap=True, n_threads=0, flags=0, decode_only=False, excl_pairs=None, region_mask=None):
        """regions: iterable of (chrom, start1, end1_incl[, name]); snps: iterable of
        (chrom, pos1, ref, alt, ref_hap, alt_hap); chrom names must already be stripped of
        'chr' and present in contig_names.  excl_pairs = (region indices, snp indices): pairs left out of the SNP -> region
        join (region-wise local phasing, xck_config.excl_*).  region_mask (bool per region): regions that are False keep
        their row index but are not counted by this engine (multi-GPU: another rank owns them)."""
        self.lib = capi.load()
        self.mode = mode
        self.contig_names = list(contig_names)
        cidx = {n: i for i, n in enumerate(self.contig_names)}
        regions = list(regions)
        if not isinstance(snps, SnpTable):
            snps = list(snps)
        # column-wise fills (a per-row structured assignment costs ~1.5 us: 1.5 s for 1 M SNPs)
        self._reg = np.zeros(len(regions), dtype=capi.REGION_DTYPE)
        if regions:
            self._reg["contig"] = [cidx[r[0]] for r in regions]
            self._reg["start"] = [r[1] for r in regions]
            self._reg["end"] = [r[2] for r in regions]
            if region_mask is not None:
                self._reg["contig"][~np.asarray(region_mask, dtype=bool)] = -1
        self._snp = np.zeros(len(snps), dtype=capi.SNP_DTYPE)
        if isinstance(snps, SnpTable):
            if len(snps):
                if int(snps.pos.max()) > 2 ** 31 - 1 or int(snps.pos.min()) < -2 ** 31:
                    raise OverflowError("SNP position does not fit the engine's int32 coordinate")
                self._snp["contig"] = np.array([cidx[n] for n in snps.names], dtype=np.int32)[snps.chrom_id]
                self._snp["pos"] = snps.pos
                self._snp["ref"] = snps.ref
                self._snp["alt"] = snps.alt
                self._snp["ref_hap"] = snps.ref_hap
                self._snp["alt_hap"] = snps.alt_hap
        elif snps:
            self._snp["contig"] = [cidx[s[0]] for s in snps]
            self._snp["pos"] = [s[1] for s in snps]
            self._snp["ref"] = [ord(s[2]) for s in snps]
            self._snp["alt"] = [ord(s[3]) for s in snps]
            self._snp["ref_hap"] = [s[4] for s in snps]
            self._snp["alt_hap"] = [s[5] for s in snps]
        cfg = capi.Config()
        cfg.struct_size = C.sizeof(capi.Config)
        cfg.mode = mode
        cfg.device = device
        cfg.min_mapq = float(min_mapq)
        cfg.min_len = int(min_len)
        cfg.incl_flag = int(incl_flag)
        cfg.excl_flag = int(excl_flag)
        cfg.no_orphan = 1 if no_orphan else 0
        cfg.min_include = float(min_include)
        cfg.min_count = float(min_count)
        cfg.min_maf = float(min_maf)
        cfg.no_dup_hap = 1 if no_dup_hap else 0
        cfg.n_cells = int(n_cells)
        cfg.n_contigs = len(self.contig_names)
        cfg.n_regions = len(regions)
        cfg.regions = self._reg.ctypes.data_as(C.POINTER(capi.Region))
        cfg.n_snps = len(snps)
        cfg.snps = self._snp.ctypes.data_as(C.POINTER(capi.Snp))
        self._bc = None
        if barcodes is not None:
            assert len(barcodes) == n_cells
            self._bc = (C.c_char_p * len(barcodes))(*[b.encode("ascii") for b in barcodes])
            cfg.barcodes = C.cast(self._bc, C.POINTER(C.c_char_p))
            if not cell_tag or len(cell_tag) != 2:
                raise ValueError("cell_tag must be a 2-character tag when barcodes are given")
            cfg.cell_tag = cell_tag.encode("ascii")
        if umi_tag:
            if len(umi_tag) != 2:
                raise ValueError("umi_tag must be a 2-character tag")
            cfg.umi_tag = umi_tag.encode("ascii")
        self._excl = None
        if excl_pairs is not None and len(excl_pairs[0]):
            er = np.ascontiguousarray(excl_pairs[0], dtype=np.int32)
            es = np.ascontiguousarray(excl_pairs[1], dtype=np.int32)
            assert len(er) == len(es)
            self._excl = (er, es)
            cfg.n_excl_pairs = len(er)
            cfg.excl_region = er.ctypes.data_as(C.POINTER(C.c_int32))
            cfg.excl_snp = es.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.n_threads = int(n_threads)
        cfg.flags = int(flags) | (capi.XCK_F_DECODE_ONLY if decode_only else 0)
        self.cfg = cfg
        self.n_cells = int(n_cells)
        self.n_regions = len(regions)
        h = C.c_void_p()
        rc = self.lib.xck_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise XckError("xck_create failed (%d): %s" % (rc, self.lib.xck_last_error(None).decode()), rc)
        self.h = h
        self.umi_bits = self.lib.xck_umi_bits(self.h)
        self._result = None

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.lib.xck_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise XckError("%s failed (%d): %s" % (what, rc, self.lib.xck_last_error(self.h).decode()), rc)

    # ------------------------------------------------------------------ data path
    def push(self, batch, device_resident=False):
        fn = self.lib.xck_push_batch_device if device_resident else self.lib.xck_push_batch
        self._check(fn(self.h, C.byref(batch)), "xck_push_batch")

    def flush(self):
        self._check(self.lib.xck_flush(self.h), "xck_flush")

    def reset(self):
        self._check(self.lib.xck_reset(self.h), "xck_reset")
        self._result = None

    def _open(self, path, n_threads=0):
        b = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.lib.xck_bam_open(path.encode(), n_threads, C.byref(b), err, 512)
        if rc != 0:
            raise XckError("xck_bam_open failed (%d): %s" % (rc, err.value.decode()), rc)
        refs = [self.lib.xck_bam_ref_name(b, i).decode() for i in range(self.lib.xck_bam_n_refs(b))]
        return b, refs

    def _opts(self, refs, sample, max_records=0, contig_mask=None, use_index=False, windows=None):
        """windows: {engine contig id: (beg0, end0)} - only the records of that contig from the first one overlapping beg0 up to
        the last one starting before end0 (0 / None = open end) are decoded (needs use_index and a .bai with a linear index)."""
        t2c = resolve_contigs(refs, self.contig_names)
        if contig_mask is not None:                     # multi-GPU: this rank only owns some contigs
            keep = np.asarray(contig_mask, dtype=bool)
            t2c = np.where((t2c >= 0) & keep[np.maximum(t2c, 0)], t2c, -1).astype(np.int32)
        o = capi.IngestOpts()
        o.struct_size = C.sizeof(capi.IngestOpts)
        o.sample = sample
        o.tid_to_contig = t2c.ctypes.data_as(C.POINTER(C.c_int32))
        o.max_records = max_records
        o.use_index = 1 if use_index else 0
        if windows:
            beg = np.zeros(len(refs), dtype=np.int32); end = np.zeros(len(refs), dtype=np.int32)
            for tid, c in enumerate(t2c.tolist()):
                if c >= 0 and c in windows:
                    beg[tid], end[tid] = int(windows[c][0] or 0), int(windows[c][1] or 0)
            o.tid_beg = beg.ctypes.data_as(C.POINTER(C.c_int32)); o.tid_end = end.ctypes.data_as(C.POINTER(C.c_int32))
            o._keep = (beg, end)
        return o, t2c

    def contig_record_counts(self, path):
        """Records per engine contig from the BAM's .bai (None if there is no usable index)."""
        b, refs = self._open(path, 1)
        try:
            t2c = resolve_contigs(refs, self.contig_names)
            out = np.zeros(len(self.contig_names), dtype=np.int64)
            m, u = C.c_int64(0), C.c_int64(0)
            for tid, c in enumerate(t2c.tolist()):
                if c < 0:
                    continue
                if self.lib.xck_bam_ref_records(b, tid, C.byref(m), C.byref(u)) != 0:
                    return None
                out[c] += m.value + u.value
            return out
        finally:
            self.lib.xck_bam_close(b)

    def contig_byte_profile(self, path, contig):
        """Compressed file offset of the first record overlapping every 16 kb window of one engine contig (from the .bai linear
        index; None without one): differences weigh the reads between two positions."""
        b, refs = self._open(path, 1)
        try:
            t2c = resolve_contigs(refs, self.contig_names)
            tids = [t for t, c in enumerate(t2c.tolist()) if c == contig]
            if not tids:
                return None
            n, ptr = C.c_int64(0), C.POINTER(C.c_uint64)()
            if self.lib.xck_bam_linear_index(b, tids[0], C.byref(n), C.byref(ptr)) != 0 or n.value == 0:
                return None
            v = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
            return (v >> np.uint64(16)).astype(np.int64)
        finally:
            self.lib.xck_bam_close(b)

    def ingest_bam(self, path, sample=0, n_threads=0, max_records=0, contig_mask=None, use_index=False, windows=None):
        """Decode one BAM file and run the join kernels on every batch. Returns #records decoded.
        contig_mask (bool per engine contig) restricts the pass to the contigs this rank owns;
        with use_index the .bai is used to inflate only their byte ranges."""
        b, refs = self._open(path, n_threads or self.cfg.n_threads)
        try:
            o, keep = self._opts(refs, sample, max_records, contig_mask, use_index, windows)
            n = C.c_int64(0)
            self._check(self.lib.xck_ingest_bam(self.h, b, C.byref(o), C.byref(n)), "xck_ingest_bam")
            return int(n.value)
        finally:
            self.lib.xck_bam_close(b)

    def ingest_bams(self, paths, n_threads=0, contig_mask=None, use_index=False, windows=None, on_file=None, ahead=2):
        """ingest_bam() over a list of files, sample = position in the list, with the next `ahead` files already open and reading
        ahead (xck_bam_prefetch: scanner + inflate, on the host pool or the GPU) while this one is parsed and joined - what pays with
        many small files (per-cell BAMs).  on_file(i, path, n_records) is called after each file.  Returns the total number of records."""
        total, queue, nxt_i = 0, [], 0                             # queue: [(b, opts, keep-alive)] of files nxt_i - len(queue) .. nxt_i - 1

        def prepare(i):
            b, refs = self._open(paths[i], n_threads or self.cfg.n_threads)
            try:
                o, keep = self._opts(refs, i, 0, contig_mask, use_index, windows)
            except Exception:
                self.lib.xck_bam_close(b)
                raise
            return (b, o, keep)

        try:
            for i, path in enumerate(paths):
                while nxt_i < len(paths) and nxt_i <= i + max(0, int(ahead)):
                    queue.append(prepare(nxt_i))
                    if nxt_i > i:
                        self.lib.xck_bam_prefetch(self.h, queue[-1][0], C.byref(queue[-1][1]))   # (best effort: an error shows up when the file is ingested)
                    nxt_i += 1
                cur = queue.pop(0)
                try:
                    n = C.c_int64(0)
                    self._check(self.lib.xck_ingest_bam(self.h, cur[0], C.byref(cur[1]), C.byref(n)), "xck_ingest_bam")
                finally:
                    self.lib.xck_bam_close(cur[0])
                total += int(n.value)
                if on_file:
                    on_file(i, path, int(n.value))
        finally:
            for q in queue:
                self.lib.xck_bam_close(q[0])
        return total

    def open_stream(self, path, sample=0, n_threads=0, contig_mask=None, use_index=False, windows=None):
        """Resumable ingest of one BAM: -> BamStream whose advance(n) decodes and joins about n further
        records (whole decode chunks) and returns (records so far, done)."""
        return BamStream(self, path, sample, n_threads, contig_mask, use_index, windows)

    def decode_bam(self, path, sample=0, n_threads=0, max_records=0, contig_mask=None, use_index=False, windows=None):
        """Pull-style decode (tests / inspection): yields dicts of numpy copies per batch."""
        b, refs = self._open(path, n_threads or self.cfg.n_threads)
        try:
            o, keep = self._opts(refs, sample, max_records, contig_mask, use_index, windows)
            bt = capi.Batch()
            while True:
                rc = self.lib.xck_bam_next_batch(self.h, b, C.byref(o), C.byref(bt))
                if rc == 0:
                    break
                if rc < 0:
                    self._check(rc, "xck_bam_next_batch")
                n = bt.n_reads
                as_np = np.ctypeslib.as_array
                d = dict(contig=bt.contig, n_reads=n, ordinal_base=int(bt.ordinal_base),
                         pos=as_np(bt.pos, (n,)).copy(), flag=as_np(bt.flag, (n,)).copy(),
                         mapq=as_np(bt.mapq, (n,)).copy(), cell=as_np(bt.cell, (n,)).copy(),
                         umi=as_np(bt.umi, (n,)).copy(), cig_off=as_np(bt.cig_off, (n + 1,)).copy())
                c_hi = int(d["cig_off"][-1])
                d["cigar"] = as_np(bt.cigar, (max(c_hi, 1),)).copy()
                if bt.seq_off:
                    d["seq_off"] = as_np(bt.seq_off, (n + 1,)).copy()
                    d["seq"] = as_np(bt.seq, (max(int(d["seq_off"][-1]), 1),)).copy()
                yield d
        finally:
            self.lib.xck_bam_close(b)

    def finish_async(self):
        """Run the fold on the GPU and enqueue the copy-out of the matrices without waiting for it."""
        self._check(self.lib.xck_finish_async(self.h), "xck_finish_async")

    def finish(self, copy=True):
        """-> {"count": (row, col, val)} or {"ad": .., "dp": .., "oth": ..}; rows are 0-based
        region indices (input order), cols 0-based cell indices, sorted by (row, col).
        copy=False returns views of the engine's pinned result buffers."""
        res = capi.Result()
        self._check(self.lib.xck_finish(self.h, C.byref(res)), "xck_finish")
        self._result = res
        out = {}
        if self.mode & XCK_MODE_BASEFC:
            out["count"] = res.count.to_numpy(copy)
        if self.mode & XCK_MODE_BAF:
            out.update(ad=res.ad.to_numpy(copy), dp=res.dp.to_numpy(copy), oth=res.oth.to_numpy(copy))
        return out

    def result_device(self):
        """Device-resident copy of the last finish(): {name: (device_ptr_of_[row|col|val], nnz)}."""
        res = capi.Result()
        self._check(self.lib.xck_get_result_device(self.h, C.byref(res)), "xck_get_result_device")
        names = (["count"] if self.mode & XCK_MODE_BASEFC else []) + (["ad", "dp", "oth"] if self.mode & XCK_MODE_BAF else [])
        out = {}
        for k in names:
            coo = getattr(res, k)
            out[k] = (C.cast(coo.row, C.c_void_p).value or 0, int(coo.nnz))
        return out

    def write_mtx(self, path, name, row_map, n_rows_out):
        """Write matrix `name` of the last finish() in the reference's exact text format."""
        if self._result is None:
            raise XckError("finish() first")
        coo = getattr(self._result, name)
        rm = np.ascontiguousarray(row_map, dtype=np.int32)
        rc = self.lib.xck_write_mtx(path.encode(), C.byref(coo), rm.ctypes.data_as(C.POINTER(C.c_int32)),
                                    int(n_rows_out), self.n_cells)
        if rc != 0:
            raise XckError("xck_write_mtx failed (%d)" % rc)

    def write_mtx_arrays(self, path, coo, row_map, n_rows_out):
        """Same writer for (row, col, val) arrays that did not come from this engine's last
        finish() - e.g. the blocks gathered from all ranks."""
        row, col, val = (np.ascontiguousarray(a, dtype=np.int32) for a in coo)
        c = capi.Coo()
        c.nnz = len(row)
        c.row, c.col, c.val = (capi.np_ptr(a, C.c_int32) for a in (row, col, val))
        rm = np.ascontiguousarray(row_map, dtype=np.int32)
        rc = self.lib.xck_write_mtx(path.encode(), C.byref(c), rm.ctypes.data_as(C.POINTER(C.c_int32)),
                                    int(n_rows_out), self.n_cells)
        if rc != 0:
            raise XckError("xck_write_mtx failed (%d)" % rc)

    def stats(self):
        st = capi.Stats()
        self._check(self.lib.xck_get_stats(self.h, C.byref(st)), "xck_get_stats")
        return {k: getattr(st, k) for k, _ in capi.Stats._fields_}


class BamStream(object):
    """One open BAM being streamed through an Engine in slices (xck_ingest_opts.pause_records)."""

    def __init__(self, eng, path, sample=0, n_threads=0, contig_mask=None, use_index=False, windows=None):
        self.eng = eng
        self.b, refs = eng._open(path, n_threads or eng.cfg.n_threads)
        self.opts, self._keep = eng._opts(refs, sample, 0, contig_mask, use_index, windows)
        self.n_records = 0
        self.done = False

    def advance(self, n_records=0):
        """Decode and join at least n_records further records (0 = the rest of the file)."""
        if self.done:
            return self.n_records, True
        self.opts.pause_records = int(n_records)
        n = C.c_int64(0)
        rc = self.eng.lib.xck_ingest_bam(self.eng.h, self.b, C.byref(self.opts), C.byref(n))
        if rc < 0:
            self.eng._check(rc, "xck_ingest_bam")
        self.n_records = int(n.value)
        self.done = rc == 0
        return self.n_records, self.done

    def close(self):
        if self.b:
            self.eng.lib.xck_bam_close(self.b)
            self.b = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
