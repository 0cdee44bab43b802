"""capi.py - ctypes mirror of include/xck.h and loader of libxck.so (the HIP engine).

There is no fallback: if the shared library is missing or cannot be loaded, every product
entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""

import ctypes as C
import os

import numpy as np

XCK_MODE_BASEFC = 1
XCK_MODE_BAF = 2
XCK_MODE_BOTH = 3
XCK_UMI_NONE = 0xFFFFFFFFFFFFFFFF
XCK_F_FORCE_KEY128 = 1
XCK_F_VERIFY_CRC = 2
XCK_F_DECODE_ONLY = 4
XCK_F_LOW_PRIORITY = 8
XCK_E_ARG, XCK_E_DEVICE, XCK_E_NOMEM, XCK_E_IO, XCK_E_STATE, XCK_E_CAPACITY = -1, -2, -3, -4, -5, -6

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libxck.so")


class Region(C.Structure):
    _fields_ = [("contig", C.c_int32), ("start", C.c_int32), ("end", C.c_int32)]


class Snp(C.Structure):
    _fields_ = [("contig", C.c_int32), ("pos", C.c_int32),
                ("ref", C.c_uint8), ("alt", C.c_uint8),
                ("ref_hap", C.c_uint8), ("alt_hap", C.c_uint8)]


REGION_DTYPE = np.dtype([("contig", "<i4"), ("start", "<i4"), ("end", "<i4")])
SNP_DTYPE = np.dtype([("contig", "<i4"), ("pos", "<i4"), ("ref", "u1"), ("alt", "u1"),
                      ("ref_hap", "u1"), ("alt_hap", "u1")])
assert REGION_DTYPE.itemsize == C.sizeof(Region) and SNP_DTYPE.itemsize == C.sizeof(Snp)


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("mode", C.c_int32), ("device", C.c_int32),
        ("min_mapq", C.c_double), ("min_len", C.c_int32),
        ("incl_flag", C.c_uint32), ("excl_flag", C.c_uint32), ("no_orphan", C.c_int32),
        ("min_include", C.c_double),
        ("min_count", C.c_double), ("min_maf", C.c_double), ("no_dup_hap", C.c_int32),
        ("n_cells", C.c_int32), ("n_contigs", C.c_int32),
        ("n_regions", C.c_int32), ("regions", C.POINTER(Region)),
        ("n_snps", C.c_int32), ("snps", C.POINTER(Snp)),
        ("barcodes", C.POINTER(C.c_char_p)),
        ("cell_tag", C.c_char * 4), ("umi_tag", C.c_char * 4),
        ("max_batch_reads", C.c_int64), ("n_threads", C.c_int32), ("flags", C.c_int32),
        ("n_excl_pairs", C.c_int32), ("excl_region", C.POINTER(C.c_int32)), ("excl_snp", C.POINTER(C.c_int32)),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("contig", C.c_int32), ("n_reads", C.c_int32), ("ordinal_base", C.c_uint64),
        ("pos", C.POINTER(C.c_int32)), ("flag", C.POINTER(C.c_uint16)),
        ("mapq", C.POINTER(C.c_uint8)), ("cell", C.POINTER(C.c_int32)),
        ("umi", C.POINTER(C.c_uint64)), ("cig_off", C.POINTER(C.c_uint32)),
        ("cigar", C.POINTER(C.c_uint32)), ("seq_off", C.POINTER(C.c_uint32)),
        ("seq", C.POINTER(C.c_uint8)),
    ]


class Coo(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("row", C.POINTER(C.c_int32)),
                ("col", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_int32))]

    def to_numpy(self, copy=True):
        """(row, col, val) int32 arrays; copy=False returns views of the engine-owned pinned
        buffers (valid until the next finish / reset / destroy of that engine)."""
        n = int(self.nnz)
        if n == 0:
            z = np.zeros(0, dtype=np.int32)
            return z, z.copy(), z.copy()
        out = tuple(np.ctypeslib.as_array(p, shape=(n,)) for p in (self.row, self.col, self.val))
        return tuple(a.copy() for a in out) if copy else out


class Result(C.Structure):
    _fields_ = [("count", Coo), ("ad", Coo), ("dp", Coo), ("oth", Coo)]


class Stats(C.Structure):
    _fields_ = [("n_batches", C.c_int64), ("n_reads", C.c_int64), ("n_hits", C.c_int64),
                ("n_hits_unique", C.c_int64), ("ms_h2d", C.c_double), ("ms_device", C.c_double),
                ("ms_join", C.c_double), ("ms_sort", C.c_double), ("ms_d2h", C.c_double),
                ("algo_bytes_join", C.c_int64), ("n_join_launches", C.c_int64), ("key_bits", C.c_int32), ("umi_bits", C.c_int32),
                ("fold_path", C.c_int32), ("fold_fallbacks", C.c_int32), ("pileup_sort_path", C.c_int32), ("fold_refinements", C.c_int32),
                ("pileup_sort2_path", C.c_int32), ("gpu_inflate_chunks", C.c_int32)]


class IngestOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("sample", C.c_int32),
                ("tid_to_contig", C.POINTER(C.c_int32)), ("use_index", C.c_int32),
                ("max_records", C.c_int64), ("pause_records", C.c_int64),
                ("tid_beg", C.POINTER(C.c_int32)), ("tid_end", C.POINTER(C.c_int32))]


# every symbol include/xck.h declares: (name, restype, argtypes)
_P = C.POINTER
class SnpText(C.Structure):
    _fields_ = [("n", C.c_int64), ("chrom_id", _P(C.c_int32)), ("pos", _P(C.c_int64)), ("ref", _P(C.c_char)), ("alt", _P(C.c_char)),
                ("ref_hap", _P(C.c_int8)), ("alt_hap", _P(C.c_int8)), ("n_chroms", C.c_int32), ("chroms", _P(C.c_char_p)),
                ("n_rejected", C.c_int64), ("rej_line", _P(C.c_int64)), ("rej_code", _P(C.c_int8))]


SYMBOLS = [
    ("xck_version", C.c_char_p, []),
    ("xck_abi_version", C.c_int, []),
    ("xck_device_count", C.c_int, []),
    ("xck_last_error", C.c_char_p, [C.c_void_p]),
    ("xck_create", C.c_int, [_P(Config), _P(C.c_void_p)]),
    ("xck_destroy", None, [C.c_void_p]),
    ("xck_umi_bits", C.c_int, [C.c_void_p]),
    ("xck_push_batch", C.c_int, [C.c_void_p, _P(Batch)]),
    ("xck_push_batch_device", C.c_int, [C.c_void_p, _P(Batch)]),
    ("xck_flush", C.c_int, [C.c_void_p]),
    ("xck_finish", C.c_int, [C.c_void_p, _P(Result)]),
    ("xck_finish_async", C.c_int, [C.c_void_p]),
    ("xck_get_result_device", C.c_int, [C.c_void_p, _P(Result)]),
    ("xck_reset", C.c_int, [C.c_void_p]),
    ("xck_get_stats", C.c_int, [C.c_void_p, _P(Stats)]),
    ("xck_bam_open", C.c_int, [C.c_char_p, C.c_int, _P(C.c_void_p), C.c_char_p, C.c_size_t]),
    ("xck_bam_close", None, [C.c_void_p]),
    ("xck_bam_n_refs", C.c_int, [C.c_void_p]),
    ("xck_bam_ref_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("xck_bam_ref_len", C.c_int64, [C.c_void_p, C.c_int]),
    ("xck_bam_ref_records", C.c_int, [C.c_void_p, C.c_int, _P(C.c_int64), _P(C.c_int64)]),
    ("xck_bam_linear_index", C.c_int, [C.c_void_p, C.c_int, _P(C.c_int64), _P(_P(C.c_uint64))]),
    ("xck_ingest_bam", C.c_int, [C.c_void_p, C.c_void_p, _P(IngestOpts), _P(C.c_int64)]),
    ("xck_bam_prefetch", C.c_int, [C.c_void_p, C.c_void_p, _P(IngestOpts)]),
    ("xck_bam_next_batch", C.c_int, [C.c_void_p, C.c_void_p, _P(IngestOpts), _P(Batch)]),
    ("xck_write_mtx", C.c_int, [C.c_char_p, _P(Coo), _P(C.c_int32), C.c_int32, C.c_int32]),
    ("xck_mtx_part_size", C.c_int, [_P(Coo), _P(C.c_int32), _P(C.c_int64), _P(C.c_int64)]),
    ("xck_write_mtx_part", C.c_int, [C.c_char_p, C.c_int64, _P(Coo), _P(C.c_int32)]),
    ("xck_parse_snp_text", C.c_int, [C.c_char_p, C.c_int, _P(_P(SnpText))]),
    ("xck_free_snp_text", None, [_P(SnpText)]),
]

_lib = None


class XckLibraryError(RuntimeError):
    pass


def _preload_torch_hip():
    """PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same sonames as /opt/rocm).
    Two HIP runtimes cannot share a process: whichever is loaded first serves both libxck.so and torch.
    If libxck.so pulled in the system runtime first, a later `import torch` finds "no ROCm-capable device".
    So when torch is installed, load ITS runtime before libxck.so (no `import torch` needed); the dynamic
    linker then binds libxck.so to the already loaded sonames."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.isfile(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(path=None):
    """Load libxck.so (built by `make -C xcltk_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    _preload_torch_hip()
    p = path or os.environ.get("XCK_LIB", LIB_PATH)
    if not os.path.isfile(p):
        raise XckLibraryError(
            "HIP engine library not found at %s - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C xcltk_amd/csrc`. "
            "There is no CPU fallback." % p)
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise XckLibraryError("cannot load %s: %s (no CPU fallback)" % (p, e))
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.xck_abi_version() != 3:
        raise XckLibraryError("ABI mismatch")
    if path is None:
        _lib = lib
    return lib


def bam_references(path):
    """[(name, length)] of a BAM header, read by the engine's own decoder (xck_bam_open; no device needed)."""
    lib = load()
    b = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.xck_bam_open(path.encode(), 1, C.byref(b), err, 512)
    if rc != 0:
        raise XckLibraryError("xck_bam_open failed (%d): %s" % (rc, err.value.decode()))
    try:
        return [(lib.xck_bam_ref_name(b, i).decode(), int(lib.xck_bam_ref_len(b, i))) for i in range(lib.xck_bam_n_refs(b))]
    finally:
        lib.xck_bam_close(b)


def np_ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def make_batch(contig, ordinal_base, pos, flag, mapq, cell, umi, cig_off, cigar,
               seq_off=None, seq=None):
    """Build a Batch struct over numpy arrays (kept alive by the returned holder tuple)."""
    pos = np.ascontiguousarray(pos, dtype=np.int32)
    flag = np.ascontiguousarray(flag, dtype=np.uint16)
    mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
    cell = np.ascontiguousarray(cell, dtype=np.int32)
    umi = np.ascontiguousarray(umi, dtype=np.uint64)
    cig_off = np.ascontiguousarray(cig_off, dtype=np.uint32)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    n = len(pos)
    assert len(flag) == n and len(mapq) == n and len(cell) == n and len(umi) == n
    assert len(cig_off) == n + 1 and (n == 0 or int(cig_off[-1]) <= len(cigar))
    b = Batch()
    b.contig = contig
    b.n_reads = n
    b.ordinal_base = ordinal_base
    b.pos = np_ptr(pos, C.c_int32)
    b.flag = np_ptr(flag, C.c_uint16)
    b.mapq = np_ptr(mapq, C.c_uint8)
    b.cell = np_ptr(cell, C.c_int32)
    b.umi = np_ptr(umi, C.c_uint64)
    b.cig_off = np_ptr(cig_off, C.c_uint32)
    b.cigar = np_ptr(cigar, C.c_uint32)
    keep = [pos, flag, mapq, cell, umi, cig_off, cigar]
    if seq_off is not None:
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint32)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        assert len(seq_off) == n + 1 and (n == 0 or int(seq_off[-1]) <= len(seq))
        b.seq_off = np_ptr(seq_off, C.c_uint32)
        b.seq = np_ptr(seq, C.c_uint8)
        keep += [seq_off, seq]
    return b, keep
