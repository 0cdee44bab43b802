"""oracle/oracle.py - TEST INFRASTRUCTURE ONLY: Python side of the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module.  It (1) loads oracle/libxck_oracle.so (C restatement, xck_oracle.c), (2) turns BAM
records read by oracle/pybam.py into the engine's structure-of-arrays batches with an
independent, pure-Python encoder, and (3) restates the reference's file-level rules
(input loaders, row numbering, .mtx/.tsv text) so that whole runs can be compared
byte for byte with golden outputs of the real reference.

Reference lines restated here:
  load_region_from_txt     xcltk/rdr/fc/utils.py:10-45 (== baf/fc/utils.py:12-47)
  format_chrom / Region    xcltk/utils/grange.py:8-27,263-264
  load_snp_from_tsv/_vcf   xcltk/baf/fc/utils.py:51-193
  prepare_config           xcltk/rdr/fc/main.py:305-431, xcltk/baf/fc/main.py:298-499
  sam_fetch chr fallback   xcltk/utils/sam.py:85-118
  fc_features row rules    xcltk/rdr/fc/core.py:96-124, xcltk/baf/fc/core.py:70-113
  afc_core region filter   xcltk/baf/fc/main.py:92-104
  merge_mtx header         xcltk/rdr/fc/utils.py:54-93
"""

import ctypes as C
import gzip
import math
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from xcltk_amd import capi          # struct layouts of include/xck.h only (no engine code)
import pybam

ORACLE_LIB = os.path.join(_HERE, "libxck_oracle.so")


class XoResult(C.Structure):
    _fields_ = [("nnz", C.c_int64 * 4),
                ("row", C.POINTER(C.c_int32) * 4), ("col", C.POINTER(C.c_int32) * 4),
                ("val", C.POINTER(C.c_int32) * 4), ("cap", C.c_int64 * 4),
                ("n_reads", C.c_int64), ("n_pass", C.c_int64)]


_olib = None


def build_oracle(force=False):
    src = os.path.join(_HERE, "xck_oracle.c")
    if force or not os.path.isfile(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", ORACLE_LIB, src, "-lm"])
    return ORACLE_LIB


def load_oracle():
    global _olib
    if _olib is None:
        if not os.path.isfile(ORACLE_LIB):
            build_oracle()
        lib = C.CDLL(ORACLE_LIB)
        lib.xo_run.restype = C.c_int
        lib.xo_run.argtypes = [C.POINTER(capi.Config), C.POINTER(capi.Batch), C.c_int, C.POINTER(XoResult)]
        lib.xo_run_mt.restype = C.c_int
        lib.xo_run_mt.argtypes = [C.POINTER(capi.Config), C.POINTER(capi.Batch), C.c_int, C.POINTER(XoResult), C.c_int]
        lib.xo_free.restype = None
        lib.xo_free.argtypes = [C.POINTER(XoResult)]
        lib.xo_frac_drop.restype = C.c_int
        lib.xo_frac_drop.argtypes = [C.c_int32, C.c_int32, C.c_double]
        _olib = lib
    return _olib


MATS = ("count", "ad", "dp", "oth")


def run_oracle(cfg, batches, n_threads=1):
    """cfg: capi.Config, batches: list of capi.Batch. Returns {name: (row, col, val)}.
    n_threads > 1: region chunks on host threads (xo_run_mt), same output."""
    lib = load_oracle()
    arr = (capi.Batch * max(1, len(batches)))(*batches)
    res = XoResult()
    rc = lib.xo_run_mt(C.byref(cfg), arr, len(batches), C.byref(res), int(n_threads)) if n_threads > 1 else \
        lib.xo_run(C.byref(cfg), arr, len(batches), C.byref(res))
    if rc != 0:
        raise RuntimeError("oracle failed: %d" % rc)
    out = {}
    for m, name in enumerate(MATS):
        n = int(res.nnz[m])
        if n:
            out[name] = tuple(np.ctypeslib.as_array(p[m], shape=(n,)).copy()
                              for p in (res.row, res.col, res.val))
        else:
            z = np.zeros(0, dtype=np.int32)
            out[name] = (z, z.copy(), z.copy())
    lib.xo_free(C.byref(res))
    return out


# --------------------------------------------------------------------------- table loading
def format_chrom(chrom):
    return chrom[3:] if chrom.lower().startswith("chr") else chrom


def _zopen(fn):
    if fn.lower().endswith((".gz", ".gzip", ".bgz")):
        return gzip.open(fn, "rt")
    return open(fn, "rt")


def load_regions(fn):
    """-> list of (chrom_stripped, start1, end1_incl, name)"""
    out = []
    with _zopen(fn) as fp:
        for line in fp:
            parts = line.rstrip().split("\t")
            if len(parts) < 4:
                raise ValueError("too few columns")
            out.append((format_chrom(parts[0]), int(parts[1]), int(parts[2]), parts[3]))
    return out


def load_snps(fn):
    """-> list of (chrom_stripped, pos1, ref, alt, ref_hap, alt_hap)"""
    is_vcf = fn.endswith(".vcf") or fn.endswith(".vcf.gz") or fn.endswith(".vcf.bgz")
    out = []
    with _zopen(fn) as fp:
        for nl, line in enumerate(fp, 1):
            if is_vcf:
                if line[0] in ("#", "\n"):
                    continue
                parts = line.rstrip().split("\t")
                if len(parts) < 10:
                    continue
                ref, alt = parts[3].upper(), parts[4].upper()
            else:
                if nl == 1:
                    continue
                parts = line.rstrip().split("\t")
                if len(parts) < 6:
                    continue
                ref, alt = parts[2].upper(), parts[3].upper()
            if len(ref) != 1 or ref not in "ACGTN" or len(alt) != 1 or alt not in "ACGTN":
                continue
            if is_vcf:
                fields = parts[8].split(":")
                if "GT" not in fields:
                    continue
                values = parts[9].split(":")
                if len(values) != len(fields):
                    continue
                gt = values[fields.index("GT")]
                if "|" in gt:
                    sep = "|"
                elif "/" in gt:
                    sep = "/"
                else:
                    continue
                a1, a2 = gt.split(sep)[:2]
            else:
                a1, a2 = parts[4], parts[5]
            if (a1 == "0" and a2 == "1") or (a1 == "1" and a2 == "0"):
                out.append((format_chrom(parts[0]), int(parts[1]), ref, alt, int(a1), int(a2)))
    return out


# --------------------------------------------------------------------------- SoA encoding
def encode_umi(s, umi_bits, intern):
    """Key code of a UMI / read-name string (see include/xck.h, xck_umi_bits)."""
    if not s:
        return capi.XCK_UMI_NONE
    L = len(s)
    if 2 * L + 1 <= umi_bits - 1 and all(c in "ACGT" for c in s):
        v = 1
        for c in s:
            v = (v << 2) | "ACGT".index(c)
        return v
    if s not in intern:
        intern[s] = len(intern)
    i = intern[s]
    if i >= (1 << (umi_bits - 1)) - 1:
        raise OverflowError("too many interned keys")
    return (1 << (umi_bits - 1)) | i


def resolve_contigs(bam_refs, contig_names):
    """sam_fetch() chr-tolerance (utils/sam.py:105-118): for every engine contig X (already
    stripped) the BAM contig is X if present, else X with 'chr' toggled. -> tid -> contig id"""
    tid_of = {n: i for i, n in enumerate(bam_refs)}
    t2c = [-1] * len(bam_refs)
    for ci, x in enumerate(contig_names):
        if x in tid_of:
            t2c[tid_of[x]] = ci
        else:
            y = x[3:] if x.startswith("chr") else "chr" + x
            if y in tid_of:
                t2c[tid_of[y]] = ci
    return t2c


def encode_bam(records, tid_to_contig, sample, cell_index, cell_tag, umi_tag, umi_bits,
               intern, with_seq=True, max_batch=None):
    """pybam records (file order) -> list of (capi.Batch, keepalive). One batch per run of
    records on the same contig (split further at max_batch)."""
    batches = []
    cur = None

    def flush():
        nonlocal cur
        if cur is None or not cur["pos"]:
            cur = None
            return
        b = capi.make_batch(cur["contig"], cur["ord"], cur["pos"], cur["flag"], cur["mapq"],
                            cur["cell"], np.array(cur["umi"], dtype=np.uint64),
                            np.array(cur["cig_off"], dtype=np.uint32),
                            np.array(cur["cigar"] or [0], dtype=np.uint32),
                            np.array(cur["seq_off"], dtype=np.uint32) if with_seq else None,
                            np.frombuffer(bytes(cur["seq"]) or b"\0", dtype=np.uint8) if with_seq else None)
        batches.append(b)
        cur = None

    for rec_idx, r in enumerate(records):
        c = tid_to_contig[r.tid] if 0 <= r.tid < len(tid_to_contig) else -1
        if c < 0:
            flush()
            continue
        if cur is None or cur["contig"] != c or (max_batch and len(cur["pos"]) >= max_batch):
            flush()
            cur = dict(contig=c, ord=(sample << 40) | rec_idx, pos=[], flag=[], mapq=[], cell=[],
                       umi=[], cig_off=[0], cigar=[], seq_off=[0], seq=bytearray())
        elif (sample << 40) | rec_idx != cur["ord"] + len(cur["pos"]):
            flush()
            cur = dict(contig=c, ord=(sample << 40) | rec_idx, pos=[], flag=[], mapq=[], cell=[],
                       umi=[], cig_off=[0], cigar=[], seq_off=[0], seq=bytearray())
        cur["pos"].append(r.pos)
        cur["flag"].append(r.flag)
        cur["mapq"].append(r.mapq)
        if cell_tag:
            cell = -1
            if r.has_tag(cell_tag):
                v = r.get_tag(cell_tag)
                cell = cell_index.get(v, -1) if isinstance(v, str) else -1
        else:
            cell = sample
        cur["cell"].append(cell)
        if umi_tag:
            key = None
            if r.has_tag(umi_tag):
                v = r.get_tag(umi_tag)
                # a numeric tag is its VALUE; 0 / 0.0 is falsy and the read is skipped (`if not umi`, rdr/fc/mcount.py:41, baf/fc/mcount.py:116-117)
                key = v if isinstance(v, str) else (None if not v else "\x01int:%r" % (v,))
        else:
            key = r.query_name
        cur["umi"].append(encode_umi(key, umi_bits, intern))
        if r.cigartuples:
            cur["cigar"].extend((l << 4) | op for op, l in r.cigartuples)
        cur["cig_off"].append(len(cur["cigar"]))
        cur["seq"] += r.seq_nibbles
        cur["seq_off"].append(len(cur["seq"]))
    flush()
    return batches


# --------------------------------------------------------------------------- whole runs
def make_config(mode, contig_names, regions, snps, n_cells, min_mapq=20, min_len=30,
                incl_flag=0, excl_flag=772, no_orphan=True, min_include=0.9, min_count=1,
                min_maf=0, no_dup_hap=True, flags=0, device=0):
    """-> (capi.Config, keepalive)"""
    cidx = {n: i for i, n in enumerate(contig_names)}
    reg = np.zeros(len(regions), dtype=capi.REGION_DTYPE)
    for i, (ch, s, e, _) in enumerate(regions):
        reg[i] = (cidx[ch], s, e)
    sn = np.zeros(len(snps), dtype=capi.SNP_DTYPE)
    for i, (ch, p, r, a, rh, ah) in enumerate(snps):
        sn[i] = (cidx[ch], p, ord(r), ord(a), rh, ah)
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
    cfg.n_cells = n_cells
    cfg.n_contigs = len(contig_names)
    cfg.n_regions = len(regions)
    cfg.regions = reg.ctypes.data_as(C.POINTER(capi.Region))
    cfg.n_snps = len(snps)
    cfg.snps = sn.ctypes.data_as(C.POINTER(capi.Snp))
    cfg.flags = flags
    return cfg, (reg, sn)


def default_umi_bits(mode, n_regions, n_snps, n_cells, force128=False):
    """Mirror of the engine's key layout rule: 64-bit keys when the UMI code
    gets at least 26 bits, else 128-bit keys with a 64-bit UMI code."""
    def nb(n):
        return max(1, int(math.ceil(math.log2(max(n, 2)))))
    rb = nb(n_regions)
    if mode == capi.XCK_MODE_BAF:
        rb = max(rb, nb(n_snps))
    ub = 64 - rb - nb(n_cells)
    if force128 or ub < 26:
        return 64
    return ub


def contig_table(regions, snps=()):
    names = []
    seen = set()
    for ch in [r[0] for r in regions] + [s[0] for s in snps]:
        if ch not in seen:
            seen.add(ch)
            names.append(ch)
    return names


def run_files(mode, bam_fns, region_fn, out_dir=None, barcode_fn=None, sample_ids=None,
              snp_fn=None, cell_tag="CB", umi_tag="UB", excl_flag=None,
              output_all_reg=True, umi_bits=None, phase=None, **kw):
    """Whole-run oracle: files in -> COO dict (and reference-format files if out_dir).
    phase: optional callable (regions, snps) -> (snps with their final haplotype indices, (excl_region, excl_snp)): the
    outcome of region-wise local phasing, which is host logic in front of the counting (baf/fc/main.py:107-153)."""
    regions = load_regions(region_fn)
    snps = load_snps(snp_fn) if snp_fn else []
    excl = None
    if phase is not None:
        snps, excl = phase(regions, snps)
    if barcode_fn:
        with _zopen(barcode_fn) as fp:
            samples = sorted(x.strip() for x in fp)
    else:
        samples = list(sample_ids)
    if cell_tag and cell_tag.upper() == "NONE":
        cell_tag = None
    if umi_tag and umi_tag.upper() == "NONE":
        umi_tag = None
    if excl_flag is None or excl_flag < 0:
        excl_flag = 772 if umi_tag else 1796
    names = contig_table(regions, snps)
    cfg, keep = make_config(mode, names, regions, snps, len(samples), excl_flag=excl_flag, **kw)
    if excl is not None and len(excl[0]):
        er, es = (np.ascontiguousarray(x, dtype=np.int32) for x in excl)
        cfg.n_excl_pairs = len(er)
        cfg.excl_region = er.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.excl_snp = es.ctypes.data_as(C.POINTER(C.c_int32))
        keep = keep + (er, es)
    if umi_bits is None:
        umi_bits = default_umi_bits(mode, len(regions), len(snps), len(samples))
    cell_index = {s: i for i, s in enumerate(samples)}
    intern = {}
    batches = []
    for bi, fn in enumerate(bam_fns):
        refs, recs = pybam.read_bam(fn)
        t2c = resolve_contigs([n for n, _ in refs], names)
        batches += encode_bam(recs, t2c, bi, cell_index, cell_tag, umi_tag, umi_bits, intern,
                              with_seq=(mode == capi.XCK_MODE_BAF))
    coo = run_oracle(cfg, [b for b, _ in batches])
    if out_dir:
        write_outputs(mode, out_dir, regions, snps, samples, coo, output_all_reg)
    return coo


def row_map(mode, regions, snps, coo, output_all_reg):
    """1-based output row of every region, 0 if the region is not written
    (rdr/fc/core.py:118-124; baf/fc/main.py:103-104 + baf/fc/core.py:101-113)."""
    n = len(regions)
    if output_all_reg:
        return np.arange(1, n + 1, dtype=np.int32)
    keep = np.zeros(n, dtype=bool)
    if mode == capi.XCK_MODE_BASEFC:
        keep[coo["count"][0]] = True
    else:
        keep[coo["dp"][0]] = True
        keep[coo["oth"][0]] = True
    rm = np.zeros(n, dtype=np.int32)
    rm[keep] = np.arange(1, int(keep.sum()) + 1, dtype=np.int32)
    return rm


def mtx_text(coo_m, rm, n_rows, n_cols):
    row, col, val = coo_m
    s = ["%%MatrixMarket matrix coordinate integer general\n", "%%\n",
         "%d\t%d\t%d\n" % (n_rows, n_cols, len(row))]
    s += ["%d\t%d\t%d\n" % (rm[r], c + 1, v) for r, c, v in zip(row.tolist(), col.tolist(), val.tolist())]
    return "".join(s)


def write_outputs(mode, out_dir, regions, snps, samples, coo, output_all_reg):
    os.makedirs(out_dir, exist_ok=True)
    rm = row_map(mode, regions, snps, coo, output_all_reg)
    n_rows = int(rm.max()) if len(rm) else 0
    reg_txt = "".join("%s\t%d\t%d\t%s\n" % (ch, s, e, name)
                      for (ch, s, e, name), r in zip(regions, rm) if r > 0)
    smp_txt = "".join(s + "\n" for s in samples)
    if mode == capi.XCK_MODE_BASEFC:
        files = {"features.tsv": reg_txt, "barcodes.tsv": smp_txt,
                 "matrix.mtx": mtx_text(coo["count"], rm, n_rows, len(samples))}
    else:
        files = {"xcltk.region.tsv": reg_txt, "xcltk.samples.tsv": smp_txt,
                 "xcltk.AD.mtx": mtx_text(coo["ad"], rm, n_rows, len(samples)),
                 "xcltk.DP.mtx": mtx_text(coo["dp"], rm, n_rows, len(samples)),
                 "xcltk.OTH.mtx": mtx_text(coo["oth"], rm, n_rows, len(samples))}
    for k, v in files.items():
        with open(os.path.join(out_dir, k), "w") as fp:
            fp.write(v)
    return files
