"""CPU-side tests (no GPU): C-ABI surface, BAM decoder vs the oracle's independent reader,
writers, front-end configuration logic, multi-rank gather (gloo, world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
import pybam
import util
from xcltk_amd import capi
from xcltk_amd.engine import Engine, XckError, resolve_contigs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "xck.h")).read()
    declared = set(re.findall(r"\b(xck_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert {n for n, _, _ in capi.SYMBOLS} == declared
    assert lib.xck_abi_version() == 3 and b"gfx950" in lib.xck_version()


def test_struct_layouts_match_header_sizes(lib):
    # a wrong layout would be rejected by xck_create's struct_size check
    cfg = capi.Config()
    cfg.struct_size = C.sizeof(capi.Config) - 4
    cfg.mode = 1; cfg.n_cells = 1
    h = C.c_void_p()
    assert lib.xck_create(C.byref(cfg), C.byref(h)) == -1
    assert b"struct_size" in lib.xck_last_error(None)


def test_ctypes_stats_mirror_matches_the_header(tmp_path):
    """capi.Stats is filled by xck_get_stats through a plain pointer: its size and field offsets must be the C header's."""
    import subprocess
    src = tmp_path / "sz.c"
    fields = [n for n, _ in capi.Stats._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "xck.h"\nint main(void) { printf("%zu", sizeof(xck_stats));\n'
                   + "".join('printf(" %%zu", offsetof(xck_stats, %s));\n' % n for n in fields) + "return 0; }\n")
    exe = tmp_path / "sz"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-I", inc, "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got[0] == C.sizeof(capi.Stats)
    assert got[1:] == [getattr(capi.Stats, n).offset for n in fields]


def test_no_cpu_fallback(lib):
    """Without a HIP device the engine must refuse to exist (no silent CPU path)."""
    if lib.xck_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(XckError) as ei:
        Engine(capi.XCK_MODE_BASEFC, ["1"], [("1", 1, 100, "g")], 1)
    assert "no HIP device" in str(ei.value) or "fallback" in str(ei.value)
    with pytest.raises(capi.XckLibraryError):
        capi.load("/nonexistent/libxck.so")


def test_decode_only_handle_refuses_compute(lib):
    eng = Engine(capi.XCK_MODE_BASEFC, ["1"], [("1", 1, 100, "g")], 1, decode_only=True)
    b, keep = capi.make_batch(0, 0, [5], [0], [60], [0], [7], [0, 1], [16])
    with pytest.raises(XckError):
        eng.push(b)
    with pytest.raises(XckError):
        eng.finish()
    eng.close()


def _decode_compare(ds, mode, n_threads, env_chunk=None, cell_tag="CB", umi_tag="UB", force128=False):
    ddir = ds if os.path.isabs(ds) else os.path.join(util.GOLDEN, "datasets", ds)
    regions, snps = util.load_tables(ddir)
    names = O.contig_table(regions, snps)
    import json
    info = json.load(open(os.path.join(ddir, "dataset.json")))
    if "barcodes" in info:
        samples = sorted(x.strip() for x in open(os.path.join(ddir, "barcodes.tsv")))
    else:
        samples = info["sample_ids"]; cell_tag = None; umi_tag = None
    if env_chunk:
        os.environ["XCK_CHUNK_BYTES"] = str(env_chunk)
    try:
        eng = Engine(mode, names, regions, len(samples), snps=snps if mode == 2 else (),
                     barcodes=samples if cell_tag else None, cell_tag=cell_tag, umi_tag=umi_tag,
                     decode_only=True, n_threads=n_threads, flags=capi.XCK_F_FORCE_KEY128 if force128 else 0)
        intern = {}
        cell_index = {s: i for i, s in enumerate(samples)}
        tot = 0
        for bi, bam in enumerate(info["bams"]):
            fn = os.path.join(ddir, bam)
            got = list(eng.decode_bam(fn, sample=bi))
            refs, recs = pybam.read_bam(fn)
            t2c = O.resolve_contigs([n for n, _ in refs], names)
            exp = O.encode_bam(recs, t2c, bi, cell_index, cell_tag, umi_tag, eng.umi_bits, intern, with_seq=(mode == 2))
            # the decoder may cut a contig run into several batches (chunk boundaries): concatenate both sides
            def cat(batches, key, is_dict):
                out = []
                for b in batches:
                    d = b if is_dict else None
                    out.append(d)
                return out
            g_pos = np.concatenate([g["pos"] for g in got]) if got else np.zeros(0, np.int32)
            e_pos = np.concatenate([k[0] for _, k in exp]) if exp else np.zeros(0, np.int32)
            assert np.array_equal(g_pos, e_pos)
            for gi, ki in (("flag", 1), ("mapq", 2), ("cell", 3)):
                assert np.array_equal(np.concatenate([g[gi] for g in got]), np.concatenate([k[ki] for _, k in exp])), gi
            # ordinals
            g_ord = np.concatenate([g["ordinal_base"] + np.arange(g["n_reads"], dtype=np.uint64) for g in got])
            e_ord = np.concatenate([np.uint64(b.ordinal_base) + np.arange(b.n_reads, dtype=np.uint64) for b, _ in exp])
            assert np.array_equal(g_ord, e_ord)
            assert [g["contig"] for g in got if g["n_reads"]][0] == exp[0][0].contig
            # CIGAR words and sequence bytes per read
            def per_read(off, data):
                return [bytes(data[off[i]:off[i + 1]].tobytes()) for i in range(len(off) - 1)]
            g_c = sum((per_read(g["cig_off"], g["cigar"]) for g in got), [])
            e_c = sum((per_read(k[5], k[6]) for _, k in exp), [])
            assert g_c == e_c
            if mode == 2:
                g_s = sum((per_read(g["seq_off"], g["seq"]) for g in got), [])
                e_s = sum((per_read(k[7], k[8]) for _, k in exp), [])
                assert g_s == e_s
            # key codes: 2-bit coded ones are equal; interned ids equal up to renaming
            gu = np.concatenate([g["umi"] for g in got]); eu = np.concatenate([k[4] for _, k in exp])
            hi = np.uint64(1 << (eng.umi_bits - 1))
            coded = ((eu & hi) == 0) | (eu == np.uint64(capi.XCK_UMI_NONE))
            assert np.array_equal(gu[coded], eu[coded])
            fwd, bwd = {}, {}
            for a, b in zip(gu[~coded].tolist(), eu[~coded].tolist()):
                assert fwd.setdefault(a, b) == b and bwd.setdefault(b, a) == a
            tot += len(g_pos)
        eng.close()
        return tot
    finally:
        os.environ.pop("XCK_CHUNK_BYTES", None)


@pytest.mark.parametrize("ds", ["c1", "dense", "multibam", "well", "special"])
@pytest.mark.parametrize("mode", [1, 2])
def test_decoder_matches_independent_reader(ds, mode):
    assert _decode_compare(ds, mode, n_threads=3) > 0


def test_decoder_many_chunks_and_record_carry_over():
    # dense / special were written with records straddling BGZF blocks; tiny chunks force the
    # carry-over path between chunks as well
    n1 = _decode_compare("dense", 2, n_threads=4, env_chunk=4096)
    n2 = _decode_compare("special", 2, n_threads=2, env_chunk=1024)
    n3 = _decode_compare("c1", 1, n_threads=8, env_chunk=70000, force128=True)
    assert n1 == 6000 and n2 > 30 and n3 == 10000


def test_decoder_rejects_garbage(tmp_path, lib):
    bad = tmp_path / "bad.bam"
    bad.write_bytes(b"this is not a bam file at all, but it is long enough........")
    b = C.c_void_p(); err = C.create_string_buffer(256)
    assert lib.xck_bam_open(str(bad).encode(), 1, C.byref(b), err, 256) == -4
    assert b"BGZF" in err.value or b"BAM" in err.value
    # the reference's pysam would read these two; the decoder names the format instead of reporting a bad gzip magic
    for name, head, word in (("x.cram", b"CRAM\x03\x00" + bytes(40), b"CRAM"), ("x.sam", b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:1\tLN:1000\n", b"SAM")):
        f = tmp_path / name
        f.write_bytes(head)
        assert lib.xck_bam_open(str(f).encode(), 1, C.byref(b), err, 256) == -4
        assert word in err.value and b"BAM only" in err.value
    # truncated real file
    src = open(os.path.join(util.GOLDEN, "datasets", "c1", "possorted.bam"), "rb").read()
    tr = tmp_path / "trunc.bam"
    tr.write_bytes(src[:len(src) // 2])
    eng = Engine(capi.XCK_MODE_BASEFC, ["1"], [("1", 1, 100, "g")], 1, decode_only=True)
    with pytest.raises(XckError):
        list(eng.decode_bam(str(tr)))
    eng.close()


def test_write_mtx_text(tmp_path, lib):
    row = np.array([0, 0, 2, 5], dtype=np.int32); col = np.array([1, 3, 0, 2], dtype=np.int32); val = np.array([1, 22, 333, 4], dtype=np.int32)
    coo = capi.Coo(); coo.nnz = 4
    coo.row = capi.np_ptr(row, C.c_int32); coo.col = capi.np_ptr(col, C.c_int32); coo.val = capi.np_ptr(val, C.c_int32)
    rm = np.array([1, 0, 2, 0, 0, 3], dtype=np.int32)
    fn = str(tmp_path / "m.mtx")
    assert lib.xck_write_mtx(fn.encode(), C.byref(coo), capi.np_ptr(rm, C.c_int32), 3, 4) == 0
    assert open(fn).read() == O.mtx_text((row, col, val), rm, 3, 4)
    assert open(fn).read().startswith("%%MatrixMarket matrix coordinate integer general\n%%\n3\t4\t4\n1\t2\t1\n")


def test_resolve_contigs_chr_fallback():
    assert resolve_contigs(["chr1", "chr2", "3"], ["1", "2", "3", "4"]).tolist() == [0, 1, 2]
    assert resolve_contigs(["1", "chr1"], ["1"]).tolist() == [0, -1]          # exact name wins
    assert resolve_contigs(["Chr1"], ["1"]).tolist() == [-1]                  # case-sensitive like sam_fetch
    assert resolve_contigs(["1"], ["chr1"]).tolist() == [0]                   # 'chr' removed on retry


def test_frontend_config_errors(tmp_path, caplog):
    from xcltk_amd.baf.fc.main import afc_wrapper
    from xcltk_amd.rdr.fc.main import fc_wrapper
    d = os.path.join(util.GOLDEN, "datasets", "c1")
    out = str(tmp_path / "o")
    assert fc_wrapper(d + "/nope.bam", d + "/barcodes.tsv", d + "/regions.tsv", out) == -1
    assert fc_wrapper(d + "/possorted.bam", d + "/barcodes.tsv", d + "/nope.tsv", out) == -1
    assert fc_wrapper(d + "/possorted.bam", None, d + "/regions.tsv", out) == -1              # cell tag without barcodes
    assert fc_wrapper(d + "/possorted.bam", d + "/barcodes.tsv", d + "/regions.tsv", None) == -1
    assert fc_wrapper(d + "/possorted.bam", d + "/barcodes.tsv", d + "/regions.tsv", out, sample_ids="a") == -1
    assert afc_wrapper(d + "/possorted.bam", d + "/barcodes.tsv", d + "/regions.tsv", d + "/nope.tsv", out) == -1
    dup = tmp_path / "dup.tsv"; dup.write_text("AAA-1\nAAA-1\n")
    assert fc_wrapper(d + "/possorted.bam", str(dup), d + "/regions.tsv", out) == -1


def test_cli_usage_and_dispatch(capsys):
    from xcltk_amd.rdr.fc.main import fc_main
    from xcltk_amd.xcltk import main
    with pytest.raises(SystemExit) as e:
        fc_main(["xcltk", "basefc"])
    assert e.value.code == 0
    txt = capsys.readouterr().out
    assert "Usage:   xcltk basefc <options>" in txt and "--minINCLUDE FLOAT|INT" in txt and "[0.900000]" in txt
    with pytest.raises(SystemExit) as e:
        main(["xcltk", "nonsense"])
    assert e.value.code == 1
    with pytest.raises(SystemExit) as e:
        main(["xcltk", "baf"])
    assert e.value.code == 0 and "--phasedSNP" in capsys.readouterr().out


def test_snp_loaders_agree_with_oracle_loaders():
    from xcltk_amd import fc_common as fcc
    for ds in ("c1", "special"):
        d = os.path.join(util.GOLDEN, "datasets", ds)
        assert fcc.load_snp_from_tsv(d + "/snps.tsv") == O.load_snps(d + "/snps.tsv")
        assert fcc.load_region_from_txt(d + "/regions.tsv") == O.load_regions(d + "/regions.tsv")
    d = os.path.join(util.GOLDEN, "datasets", "c1")
    assert fcc.load_snp_from_vcf(d + "/snps.vcf") == O.load_snps(d + "/snps.vcf") == fcc.load_snp_from_tsv(d + "/snps.tsv")


def test_lpt_assign_balances():
    from xcltk_amd.shard import contig_owner, lpt_assign
    from xcltk_amd.synth.soa import HG38_LENGTHS
    bins = lpt_assign(HG38_LENGTHS, 8)
    assert sorted(sum(bins, [])) == list(range(24))
    loads = [sum(HG38_LENGTHS[i] for i in b) for b in bins]
    assert max(loads) / (sum(loads) / 8) < 1.08
    assert set(contig_owner(["a"] * 24, HG38_LENGTHS, 8).tolist()) == set(range(8))


_SHARDED_WRITER_WORKER = r"""
import os, sys, ctypes as C
import numpy as np
import torch, torch.distributed as dist
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from xcltk_amd import capi, shard
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(11)                              # the same matrix on every rank; each rank keeps the rows it owns
n_rows, n_cols, n = 900, 123, 60000
row_owner = np.repeat(rng.integers(0, world, 45), 20).astype(np.int32)
row = rng.integers(0, n_rows, n).astype(np.int32); col = rng.integers(0, n_cols, n).astype(np.int32)
val = rng.choice(np.array([1, 7, 10, 99, 100, 12345, 2147483647], dtype=np.int32), n).astype(np.int32)
o = np.lexsort((col, row)); row, col, val = row[o], col[o], val[o]
rm = np.arange(1, n_rows + 1, dtype=np.int32); rm[::5] = 0; rm[rm > 0] = np.arange(1, int((rm > 0).sum()) + 1)
def allred(x):
    t = torch.from_numpy(np.ascontiguousarray(x)); dist.all_reduce(t); return t.numpy()
sel = row_owner[row] == rank
lines = shard.write_mtx_sharded(os.path.join(out, "sharded.mtx"), (row[sel], col[sel], val[sel]), rm, row_owner, int(rm.max()), n_cols,
                                rank, allred, dist.barrier)
if rank == 0:
    lib = capi.load()
    c = capi.Coo(); c.nnz = n; c.row, c.col, c.val = (capi.np_ptr(x, C.c_int32) for x in (row, col, val))
    assert lib.xck_write_mtx(os.path.join(out, "single.mtx").encode(), C.byref(c), capi.np_ptr(rm, C.c_int32), int(rm.max()), n_cols) == 0
    a = open(os.path.join(out, "single.mtx"), "rb").read(); b = open(os.path.join(out, "sharded.mtx"), "rb").read()
    assert a == b and lines == int((rm[row] > 0).sum()), (len(a), len(b), lines)
    print("SHARDED_WRITE_OK", len(a), lines)
dist.barrier()
dist.destroy_process_group()
"""


def test_mtx_written_by_all_ranks_equals_single_writer(tmp_path):
    """shard.write_mtx_sharded (three gloo ranks, every rank writes the lines of its rows at offsets derived from ONE all-reduce
    of text sizes) must produce the bytes xck_write_mtx writes for the merged matrix - row map with dropped rows included."""
    script = tmp_path / "w.py"
    script.write_text(_SHARDED_WRITER_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script), ROOT, str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert "SHARDED_WRITE_OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("ds,mask", [("multibam", [True, False]), ("multibam", [False, True]), ("special", [False, True, False])])
def test_indexed_contig_subset_decode(ds, mask):
    """use_index + contig_mask (multi-GPU sharding) must yield exactly the records of the owned
    contigs, in file order, with the same ordinals' relative order."""
    import json
    ddir = os.path.join(util.GOLDEN, "datasets", ds)
    regions, snps = util.load_tables(ddir)
    names = O.contig_table(regions, snps)
    info = json.load(open(os.path.join(ddir, "dataset.json")))
    samples = sorted(x.strip() for x in open(os.path.join(ddir, "barcodes.tsv")))
    mask = (mask + [False] * len(names))[:len(names)]
    eng = Engine(capi.XCK_MODE_BAF, names, regions, len(samples), snps=snps, barcodes=samples, cell_tag="CB", umi_tag="UB",
                 decode_only=True, n_threads=2)
    fn = os.path.join(ddir, info["bams"][0])
    counts = eng.contig_record_counts(fn)
    full = list(eng.decode_bam(fn))
    sub = list(eng.decode_bam(fn, contig_mask=mask, use_index=True))
    sub_noidx = list(eng.decode_bam(fn, contig_mask=mask, use_index=False))
    eng.close()
    assert counts is not None
    want = [b for b in full if mask[b["contig"]]]
    for got in (sub, sub_noidx):
        assert [b["contig"] for b in got] == [b["contig"] for b in want]
        for g, w in zip(got, want):
            for k in ("pos", "flag", "mapq", "cell", "umi"):
                assert np.array_equal(g[k], w[k]), k
    assert sum(b["n_reads"] for b in full if b["contig"] >= 0) <= int(counts.sum())
    assert sum(b["n_reads"] for b in want) == int(counts[np.array(mask)].sum())


def test_prefetched_reader_decodes_the_same_batches():
    """xck_bam_prefetch (csrc/bam.cpp): a reader whose scanner and inflate were started ahead of time hands out exactly the batches
    of a reader that was not; the engine's intern table (well mode: read names as keys, restarted per file) is untouched by the
    prefetch of the NEXT file while this one is decoded; Engine.ingest_bams counts what the per-file calls count."""
    import ctypes as C
    import json
    ddir = os.path.join(util.GOLDEN, "datasets", "multibam")
    regions, snps = util.load_tables(ddir)
    names = O.contig_table(regions, snps)
    info = json.load(open(os.path.join(ddir, "dataset.json")))
    bams = [os.path.join(ddir, b) for b in info["bams"]]
    assert len(bams) >= 2
    eng = Engine(capi.XCK_MODE_BAF, names, regions, len(bams), snps=snps, decode_only=True, n_threads=2)   # well mode: column = file, key = read name
    try:
        plain = [list(eng.decode_bam(fn, sample=i)) for i, fn in enumerate(bams)]
        per_file = [eng.ingest_bam(fn, sample=i) for i, fn in enumerate(bams)]
        assert eng.ingest_bams(bams) == sum(per_file) and eng.ingest_bams(bams, ahead=0) == sum(per_file)
        # file 1 is prefetched BEFORE file 0 is decoded, then both are pulled batch by batch
        b1, refs1 = eng._open(bams[1], 2)
        o1, keep1 = eng._opts(refs1, 1, 0, None, False, None)
        assert eng.lib.xck_bam_prefetch(eng.h, b1, C.byref(o1)) == 0
        assert eng.lib.xck_bam_prefetch(eng.h, b1, C.byref(o1)) == 0          # (a second call is harmless)
        first = list(eng.decode_bam(bams[0], sample=0))
        bt, second = capi.Batch(), []
        while eng.lib.xck_bam_next_batch(eng.h, b1, C.byref(o1), C.byref(bt)) > 0:
            n = bt.n_reads
            second.append(dict(contig=bt.contig, n_reads=n, ordinal_base=int(bt.ordinal_base), pos=np.ctypeslib.as_array(bt.pos, (n,)).copy(),
                               cell=np.ctypeslib.as_array(bt.cell, (n,)).copy(), umi=np.ctypeslib.as_array(bt.umi, (n,)).copy()))
        eng.lib.xck_bam_close(b1)
        for got, want in ((first, plain[0]), (second, plain[1])):
            assert [(g["contig"], g["n_reads"], g["ordinal_base"]) for g in got] == [(w["contig"], w["n_reads"], w["ordinal_base"]) for w in want]
            for g, w in zip(got, want):
                assert np.array_equal(g["pos"], w["pos"]) and np.array_equal(g["cell"], w["cell"])
                # (interned read names: ids are per file and per decode - compare their equality pattern, not the ids)
                first_seen = lambda a: [{v: i for i, v in reversed(list(enumerate(a.tolist())))}[v] for v in a.tolist()]
                assert first_seen(g["umi"]) == first_seen(w["umi"])
    finally:
        eng.close()


def test_fast_inflate_matches_zlib_on_every_block():
    """inflate_fast.h vs zlib: every BGZF block of the golden BAMs plus 600 synthetic streams (stored, fixed and
    dynamic blocks, all levels / strategies, corrupted and truncated inputs must not write out of bounds)."""
    exe = os.path.join(ROOT, "xcltk_amd", "csrc", "xck_inflate_test")
    if not os.path.isfile(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "xcltk_amd", "csrc"), "xck_inflate_test"])
    import glob
    bams = sorted(glob.glob(os.path.join(util.GOLDEN, "datasets", "*", "*.bam")))
    r = subprocess.run([exe] + bams, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatches 0 fast-decoder-declined 0" in r.stdout, r.stdout[-500:]


def test_decoder_same_output_with_zlib_and_fast_inflate():
    """XCK_INFLATE=zlib forces the library decoder: both must give identical SoA batches (one decoder thread:
    interned UMI ids are labels handed out in arrival order, which is only reproducible without a thread pool)."""
    code = ("import sys, hashlib, numpy as np; sys.path[:0]=[%r, %r, %r]\n"
            "import util, oracle as O\nfrom xcltk_amd import capi\nfrom xcltk_amd.engine import Engine\n"
            "import os\nd=os.path.join(util.GOLDEN,'datasets','dense')\nregions,snps=util.load_tables(d)\nnames=O.contig_table(regions,snps)\n"
            "bc=sorted(x.strip() for x in open(d+'/barcodes.tsv'))\n"
            "e=Engine(2,names,regions,len(bc),snps=snps,barcodes=bc,cell_tag='CB',umi_tag='UB',decode_only=True,n_threads=1)\n"
            "h=hashlib.md5()\n"
            "for b in e.decode_bam(d+'/possorted.bam'):\n"
            "    [h.update(np.ascontiguousarray(b[k]).tobytes()) for k in ('pos','flag','mapq','cell','umi','cig_off','cigar','seq_off','seq')]\n"
            "print(h.hexdigest())\n") % (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"))
    outs = []
    for mode in ("fast", "zlib"):
        env = dict(os.environ, XCK_INFLATE=mode)
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 32


def test_fraction_bounds():
    """engine.hip frac_bounds/frac_below: integer bounds around RN(f*n) decide `m / float(n) < f`
    (rdr/fc/core.py:160-165) exactly like the IEEE divide for every (m, n) - checked exhaustively for
    n <= 1500 and for thresholds that make f*n land on / next to integers."""
    fs = [0.9, 0.5, 0.25, 0.75, 0.1, 1.0 / 3.0, 2.0 / 3.0, 0.999999, 1e-9, 0.3, 0.7, np.nextafter(0.5, 0), np.nextafter(0.5, 1)]
    n = np.arange(1, 1501, dtype=np.int64)
    for f in fs:
        p = f * n.astype(np.float64)
        m_rej = np.ceil(p * (1.0 - 2.0 ** -50)).astype(np.int64)
        m_acc = np.floor(p * (1.0 + 2.0 ** -50)).astype(np.int64) + 1
        assert np.all(m_acc - m_rej <= 1)                   # at most one m per n needs the divide
        for m in range(0, 1501):
            ok = n >= m
            exact = (m / n.astype(np.float64)) < f
            fast = np.where(m < m_rej, True, np.where(m >= m_acc, False, exact))
            assert np.array_equal(fast[ok], exact[ok])
            # and the bounds alone are never wrong where they decide
            assert np.all(exact[ok & (m < m_rej)]) and not np.any(exact[ok & (m >= m_acc)])


def test_write_mtx_threads_give_identical_files(tmp_path, monkeypatch):
    """xck_write_mtx formats 1 M-entry chunks on several threads: same bytes as the single-threaded path, header counts
    only the rows that survive row_map (merge_mtx format, rdr/fc/utils.py:54-93)."""
    import ctypes as C
    lib = capi.load()
    rng = np.random.default_rng(4)
    n = 2_300_000                                              # three chunks, the last one partial
    row = np.sort(rng.integers(0, 5000, n)).astype(np.int32); col = rng.integers(0, 300, n).astype(np.int32)
    val = rng.integers(1, 100000, n).astype(np.int32)
    coo = capi.Coo(); coo.nnz = n
    coo.row = row.ctypes.data_as(C.POINTER(C.c_int32)); coo.col = col.ctypes.data_as(C.POINTER(C.c_int32)); coo.val = val.ctypes.data_as(C.POINTER(C.c_int32))
    rm = np.zeros(5000, np.int32); keep = rng.random(5000) < 0.8; rm[keep] = np.arange(1, int(keep.sum()) + 1)
    outs = []
    for nt in ("1", "5"):
        monkeypatch.setenv("XCK_WRITE_THREADS", nt)
        fn = str(tmp_path / ("m%s.mtx" % nt))
        assert lib.xck_write_mtx(fn.encode(), C.byref(coo), rm.ctypes.data_as(C.POINTER(C.c_int32)), int(rm.max()), 300) == 0
        outs.append(open(fn, "rb").read())
    assert outs[0] == outs[1]
    lines = outs[0].split(b"\n")
    assert lines[0] == b"%%MatrixMarket matrix coordinate integer general" and lines[1] == b"%%"
    kept = rm[row] > 0
    assert lines[2] == b"%d\t300\t%d" % (int(rm.max()), int(kept.sum()))
    i = int(np.flatnonzero(kept)[0])
    assert lines[3] == b"%d\t%d\t%d" % (rm[row[i]], col[i] + 1, val[i])
    assert len(lines) == 3 + int(kept.sum()) + 1 and lines[-1] == b""


@pytest.mark.parametrize("seed", range(int(os.environ.get("XCK_DECODER_FUZZ_SEEDS", "12"))))
def test_decoder_fuzz_random_bam_layouts(seed, tmp_path):
    """Random BAMs from the Python writer against the independent reader: records packed across BGZF blocks or not, random
    compression levels, aux tags of every scalar type before / between / after CB and UB (first occurrence wins), numeric
    and IUPAC UMIs, barcodes outside the list, empty and very long read names, records without sequence or CIGAR, long
    CIGARs - decoded with random thread counts and chunk sizes, 10x keys or read-name keys."""
    import json
    from xcltk_amd.synth.bamwriter import BamWriter
    rng = np.random.default_rng(500 + seed)
    d = tmp_path / ("fz%d" % seed); d.mkdir()
    refs = [("chr1", 300000), ("2", 200000), ("chrM", 16000)][:int(rng.integers(1, 4))]
    bcs = sorted({"".join("ACGT"[i] for i in rng.integers(0, 4, 10)) + "-1" for _ in range(int(rng.integers(1, 40)))})
    well = seed % 3 == 2
    n_bams = int(rng.integers(1, 4)) if well else int(rng.integers(1, 3))
    bams = []
    for bi in range(n_bams):
        fn = "b%d.bam" % bi
        w = BamWriter(str(d / fn), refs, align_records=bool(rng.integers(0, 2)), level=int(rng.choice([0, 1, 6, 9])))
        recs = []
        for _ in range(int(rng.integers(50, 3000))):
            tid = int(rng.integers(0, len(refs))); pos = int(rng.integers(0, refs[tid][1] - 200))
            k = rng.integers(0, 8)
            L = int(rng.integers(20, 151))
            if k < 4: cig = "%dM" % L
            elif k == 4: cig = "%dM%dN%dM" % (L // 2, int(rng.integers(1, 5000)), L - L // 2)
            elif k == 5: cig = "3S%dM2I%dM1D%dM" % (10, L - 20, 5)
            elif k == 6: cig = "".join("%dM1I" % int(rng.integers(1, 4)) for _ in range(int(rng.integers(30, 200)))); L = None
            else: cig = ""
            if L is None:
                from xcltk_amd.synth.bamwriter import parse_cigar
                L = sum(l for op, l in parse_cigar(cig) if op in (0, 1, 4, 7, 8))
            elif k == 5:
                L = 3 + 10 + 2 + (L - 20) + 5
            seq = "" if (rng.random() < 0.05 or not cig) else "".join("ACGTNRY"[i] for i in rng.choice(7, L, p=[.24, .24, .24, .24, .02, .01, .01]))
            tags = []
            def junk():
                t = int(rng.integers(0, 7))
                name = "X%s" % "abcdefg"[t]
                return [(name, int(rng.integers(-100, 100)), "c"), (name, int(rng.integers(0, 60000)), "S"), (name, int(rng.integers(-2**31, 2**31)), "i"),
                        (name, 1.5, "f"), (name, "q", "A"), (name, "some text", "Z"), (name, int(rng.integers(0, 200)), "C"),
                        ("YB", ("S", [int(x) for x in rng.integers(0, 60000, int(rng.integers(0, 9)))]), "B"), ("YH", "1AE301", "H"),
                        ("YF", ("f", [0.5, 2.0]), "B")][int(rng.integers(0, 10))]
            for _j in range(int(rng.integers(0, 3))): tags.append(junk())
            if not well:
                r = rng.random()
                if r < 0.85: tags.append(("CB", bcs[int(rng.integers(0, len(bcs)))]))
                elif r < 0.92: tags.append(("CB", "TTTTTTTTTT-9"))
                if rng.random() < 0.1: tags.append(("CB", bcs[0]))              # second occurrence: ignored
                for _j in range(int(rng.integers(0, 2))): tags.append(junk())
                r = rng.random()
                if r < 0.8: tags.append(("UB", "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.choice([8, 12, 30])))))),
                elif r < 0.86: tags.append(("UB", "ACGTNACGTN"))
                elif r < 0.9: tags.append(("UB", int(rng.integers(0, 50)), "i"))
                elif r < 0.93: tags.append(("UB", ""))
            for _j in range(int(rng.integers(0, 2))): tags.append(junk())
            qname = ["", "r%d" % int(rng.integers(0, 400)), "a_very_long_read_name_" * 9 + str(int(rng.integers(0, 50)))][int(rng.choice(3, p=[.02, .9, .08]))]
            flag = int(rng.choice([0, 16, 99, 147, 4, 256, 1024, 2048]))
            recs.append((tid, pos, qname, flag, int(rng.choice([0, 3, 20, 60, 255])), cig, seq, tuple(t for t in tags if not isinstance(t, tuple) or len(t) in (2, 3))))
        recs.sort(key=lambda r: (r[0], r[1]))
        for r in recs:
            w.write(r[0], r[1], r[2], r[3], r[4], r[5], r[6], tags=[t[0] if isinstance(t[0], tuple) else t for t in r[7]])
        w.close()
        bams.append(fn)
    (d / "regions.tsv").write_text("chr1\t100\t5000\tg1\n2\t1\t90000\tg2\nM\t1\t16000\tg3\n")
    (d / "snps.tsv").write_text("chrom\tpos\tref\talt\tref_hap\talt_hap\nchr1\t150\tA\tC\t0\t1\n2\t500\tG\tT\t1\t0\n")
    info = dict(bams=bams)
    if well:
        info["sample_ids"] = ["s%d" % i for i in range(n_bams)]
    else:
        info["barcodes"] = "barcodes.tsv"
        (d / "barcodes.tsv").write_text("".join(b + "\n" for b in bcs))
    (d / "dataset.json").write_text(json.dumps(info))
    n = _decode_compare(str(d), int(rng.choice([1, 2])), n_threads=int(rng.integers(1, 7)), env_chunk=int(rng.choice([0, 3000, 20000])) or None,
                        force128=bool(rng.random() < 0.2))
    assert n > 0


# ---------------------------------------------------------------- xcltk convert: bins as features
_CONVERT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convert")


def _convert_cases():
    import json
    with open(os.path.join(_CONVERT, "cases.json")) as fp:
        return [(c["name"], c["argv"]) for c in json.load(fp)["cases"]]


@pytest.mark.parametrize("name,argv", _convert_cases())
def test_convert_matches_reference_outputs(name, argv, tmp_path):
    """Fixtures = outputs of the reference's `xcltk convert` (oracle/refgen/make_convert_goldens.py)."""
    from xcltk_amd.xcltk import main
    out = str(tmp_path / name)
    av = [a.replace("$BED", os.path.join(_CONVERT, "in.bed")).replace("$TSV", os.path.join(_CONVERT, "in.tsv")).replace("$GTF", os.path.join(_CONVERT, "in.gtf")) for a in argv]
    main(["xcltk", "convert"] + av + ["-o", out])
    assert open(out, "rb").read() == open(os.path.join(_CONVERT, name), "rb").read()


def test_convert_gff_gene_ids_as_feature_names(tmp_path):
    """gff branch (utils/gregion.py:60-62): gene lines only; the id is the LAST ID / gene_id attribute, '*' without one."""
    from xcltk_amd.utils.gregion import load_regions
    regs = load_regions(os.path.join(_CONVERT, "in.gtf"), "gff")
    assert [(r.chrom, r.start, r.end, r.id) for r in regs] == [
        ("chr1", 11869, 14409, "ENSG00000223972.5"), ("chr1", 14404, 29570, "ENSG00000227232.5"), ("2", 100, 2000, "gene:G3"),
        ("X", 5, 9, "*"), ("X", 50, 90, "second")]


def test_convert_errors_and_feature_tables(tmp_path, capsys):
    from xcltk_amd.tools.convert import convert_main
    from xcltk_amd.utils import gregion as G
    for av in (["-B", "0"], ["-B", "50", "-H", "37"], ["-i", str(tmp_path / "missing.bed"), "-I", "bed"], ["-B", "50", "-O", "gff"],
               ["-i", os.path.join(_CONVERT, "in.bed"), "-I", "vcf"]):
        with pytest.raises(SystemExit) as e:
            convert_main(["xcltk", "convert"] + av)
        assert e.value.code == 1
    bad = tmp_path / "nonl.tsv"
    bad.write_text("1\t1\t1000\nMT\t7\t9")                      # no final newline: the reference eats the "9" and fails, so do we
    with pytest.raises(SystemExit) as e:
        convert_main(["xcltk", "convert", "-i", str(bad), "-I", "tsv"])
    assert e.value.code == 1
    capsys.readouterr()
    # stdout when no -o
    convert_main(["xcltk", "convert", "-B", "100000", "-H", "19"])
    txt = capsys.readouterr().out.split("\n")
    assert txt[0] == "1\t1\t100000000" and txt[2] == "1\t200000001\t300000000"      # last bin not clipped to 249,250,621
    assert G.chr2reg("1", 0, 10) is None and G.chr2reg("1", "x", 10) is None and G.get_fixsize_regions(50, "hg38") is None
    assert [(r.start, r.end) for r in G.chr2reg("c", 25, 10)] == [(1, 10), (11, 20), (21, 30)]
    assert len(G.chr2reg("c", 30, 10)) == 3
    # bins over the contigs of a BAM header, with the chr prefix retried either way; four-column feature table loads back
    case, _, _, _ = util.load_case("c1_basefc_default", str(tmp_path))
    bam = case["kwargs"]["sam_fn"]
    refs = capi.bam_references(bam)
    name0, len0 = refs[0]
    other = name0[3:] if name0.startswith("chr") else "chr" + name0
    bins = G.get_fixsize_reg_from_sam_header([other], 100, bam)
    assert bins[0].chrom == name0 and len(bins) == -(-len0 // 100000) and bins[-1].end >= len0
    assert G.get_fixsize_reg_from_sam_header(["nope"], 100, bam) is None
    fn = str(tmp_path / "bins.tsv")
    G.output_feature_table(bins, fn)
    from xcltk_amd.fc_common import load_region_from_txt
    regs = load_region_from_txt(fn)
    assert len(regs) == len(bins) and regs[0][3] == bins[0].id and regs[1][1] == bins[1].start


# ---------------------------------------------------------------- untrusted input: the decoder under ASAN + UBSan (host-only build)
def test_decoder_survives_corrupt_bams_under_sanitizers(tmp_path):
    """tools/asan: csrc/bam.cpp + csrc/api.cpp built by g++ with -fsanitize=address,undefined (device side stubbed), run
    over BAMs whose record stream, compressed bytes, length or index were damaged: an error code or a decode, never a
    bad access (a sanitizer report aborts the harness).  Larger campaigns: tools/asan/mutate_bams.py with more seeds."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asan = os.path.join(root, "tools", "asan")
    r = subprocess.run(["make", "-C", asan], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 and ("cannot find -lasan" in r.stdout or "cannot find -lubsan" in r.stdout or "libasan" in r.stdout):
        pytest.skip("no sanitizer runtime for g++ here")
    assert r.returncode == 0, r.stdout[-2000:]
    gold = os.path.join(root, "tests", "golden", "datasets")
    src = [os.path.join(gold, "special", f) for f in sorted(os.listdir(os.path.join(gold, "special"))) if f.endswith(".bam")][0]
    multi = [os.path.join(gold, "multibam", f) for f in sorted(os.listdir(os.path.join(gold, "multibam"))) if f.endswith(".bam")][0]
    mut = os.path.join(asan, "mutate_bams.py")
    subprocess.check_call([sys.executable, mut, src, str(tmp_path / "a"), "160", "5"])
    subprocess.check_call([sys.executable, mut, multi, str(tmp_path / "i"), "40", "6", "--bai"])
    rnd = np.random.default_rng(9)                                 # SNP lists with random byte damage: the text parser (csrc/snptext.cpp)
    os.makedirs(str(tmp_path / "s"))
    for k in range(120):
        lines = (_VCF_LINES if k & 1 else _TSV_LINES) * 3
        b = bytearray(("\n".join(lines) + "\n").encode())
        for _ in range(int(rnd.integers(1, 12))):
            b[int(rnd.integers(0, len(b)))] = int(rnd.choice([9, 10, 58, 124, 47, 48, 49, 0, 255, 65, 13, int(rnd.integers(0, 256))]))
        open(str(tmp_path / "s" / ("m%03d.%s" % (k, "vcf" if k & 1 else "tsv"))), "wb").write(bytes(b))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")          # the reader parks its buffers in a process-wide pool on purpose
    for sub, extra in (("a", {}), ("i", {"XCK_ASAN_INDEX": "1"}), ("s", {})):
        files = sorted(str(p) for p in (tmp_path / sub).glob("*.bam" if sub != "s" else "*.*"))
        r = subprocess.run([os.path.join(asan, "decoder_asan")] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(env, **extra), timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        m = re.search(r"(\d+) files: (\d+) clean decodes, (\d+) rejected", r.stdout)
        assert m and int(m.group(1)) == len(files) and int(m.group(2)) + int(m.group(3)) == (2 if sub == "s" else 4) * len(files)
        if sub == "a":
            assert int(m.group(3)) > len(files)                      # most damage is detected and reported as an error


# ---------------------------------------------------------------- native SNP text parser (csrc/snptext.cpp) == the generic loaders
_TSV_LINES = [
    "chrom\tpos\tref\talt\tref_hap\talt_hap",
    "chr1\t100\tA\tG\t0\t1", "1\t101\tc\tt\t1\t0", "CHRX\t7\tN\tA\t0\t1\textra\tcolumns", "chr2\t5\tA\tG\t0\t1\t\t ",
    "chr2\t6\tAC\tG\t0\t1", "chr2\t7\tA\tG,T\t0\t1", "chr2\t8\tA\tG\t0\t0", "chr2\t9\tA\tG\t1\t1", "chr2\t10\tA\tG\t0", "",
    "chr2\t11\tR\tG\t0\t1", "chr2\t12\tA\tG\t0 \t1", "chr2\t0012\tA\tG\t1\t0", "chrM\t13\tt\tn\t0\t1", "2\t13\tA\tG\t0\t1", "\t14\tA\tG\t0\t1",
    "chr\t15\tA\tG\t0\t1", " chr3\t16\tA\tG\t0\t1", "chr2\t17\tA\tG\t0\t1 ", "chr2\t18\t\tG\t0\t1", "chr2\t19\tA\tG\t01\t0",
]
_VCF_LINES = [
    "##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2", "",
    "chr1\t100\t.\tA\tG\t.\tPASS\t.\tGT\t0|1", "1\t101\trs1\tc\tt\t50\t.\tAC=1\tGT:PS\t1|0:77", "chr1\t102\t.\tA\tG\t.\t.\t.\tDP:GT\t9:0/1\t1|1",
    "chr1\t103\t.\tA\tG\t.\t.\t.\tGT\t1/0 ", "chr1\t104\t.\tA\tG\t.\t.\t.\tGT\t0|0", "chr1\t105\t.\tA\tG,T\t.\t.\t.\tGT\t0|1",
    "chr1\t106\t.\tAT\tG\t.\t.\t.\tGT\t0|1", "chr1\t107\t.\tA\tG\t.\t.\t.\tDP\t9", "chr1\t108\t.\tA\tG\t.\t.\t.\tGT:DP\t0|1",
    "chr1\t109\t.\tA\tG\t.\t.\t.\tGT\t01", "chr1\t110\t.\tA\tG\t.\t.\t.\tGT\t0|1|1", "chr1\t111\t.\tA\tG\t.\t.\t.\tGT\t.|1", "chr1\t112\t.\tA\tG\t.\t.\t.\tGT",
    "chr1\t113\t.\tn\tA\t.\t.\t.\tGT\t1|0", "chrY\t114\t.\tA\tG\t.\t.\t.\tGQ:GT:DP\t3:0|1:4\t0|1", "chr1\t115\t.\tA\tG\t.\t.\t.\tGT\t0/1|x", "chr1\t116\t.\tA\tG\t.\t.\t.\tgt\t0|1",
    "chr1\t117\t.\tA\tG\t.\t.\t.\tGT:GT\t1|0:0|1", "#late comment", "chr1\t118\t.\tA\t*\t.\t.\t.\tGT\t0|1",
]


@pytest.mark.parametrize("kind", ["tsv", "vcf", "tsv.gz", "vcf.gz", "tsv_nonl"])
def test_native_snp_parser_equals_generic_loaders(kind, tmp_path, monkeypatch, capsys):
    import gzip
    from xcltk_amd import fc_common as F
    from xcltk_amd.snptable import SnpTable
    vcf = kind.startswith("vcf")
    text = "\n".join(_VCF_LINES if vcf else _TSV_LINES) + ("" if kind == "tsv_nonl" else "\n")
    fn = str(tmp_path / ("snps." + kind.replace("_nonl", "")))
    if kind.endswith(".gz"):
        with gzip.open(fn, "wt") as fp:
            fp.write(text)
    else:
        open(fn, "w").write(text)
    loader = F.load_snp_from_vcf if vcf else F.load_snp_from_tsv
    got = loader(fn)
    assert isinstance(got, SnpTable)
    capsys.readouterr()
    assert loader(fn, verbose=True) == got                      # what the front-ends call
    said = capsys.readouterr().err
    monkeypatch.setenv("XCK_PY_LOADERS", "1")
    exp = loader(fn, verbose=True)
    assert capsys.readouterr().err == said and said.count("[W::") >= 8          # the same warning for the same lines
    assert isinstance(exp, list) and len(exp) >= 8
    assert got == exp and list(got) == exp and len(got) == len(exp) and got[3] == exp[3] and got[-1] == exp[-1] and got[2:5] == exp[2:5]
    assert got.chroms() == list(dict.fromkeys(s[0] for s in exp))
    regs = [("1", 1, 200, "a"), ("2", 14, 20, "b"), ("X", 8, 9, "c"), ("M", 13, 13, "d"), ("nope", 1, 5, "e"), ("1", 300, 200, "f")]
    from xcltk_amd.baf.fc.main import regions_with_snps
    assert regions_with_snps(regs, got) == regions_with_snps(regs, exp)
    assert F.contig_table(regs, got) == F.contig_table(regs, exp)


def test_native_snp_parser_declines_what_it_cannot_reproduce(tmp_path, monkeypatch):
    """Carriage returns, non-ASCII bytes and positions that are not plain digits go to the generic loader, whose result
    (or exception) is the behaviour of record."""
    from xcltk_amd import fc_common as F
    head = "chrom\tpos\tref\talt\tref_hap\talt_hap\n"
    for name, body in (("crlf", "chr1\t5\tA\tG\t0\t1\r\nchr1\t6\tA\tG\t1\t0\r\n"), ("plus", "chr1\t+5\tA\tG\t0\t1\n"), ("blank", "chr1\t 5\tA\tG\t0\t1\n"),
                       ("latin", "chr1\t5\tA\tG\t0\t1\n# café\t1\tA\tG\t0\t1\n"), ("big", "chr1\t1234567890123456789012\tA\tG\t0\t1\n")):
        fn = str(tmp_path / (name + ".tsv"))
        open(fn, "w", encoding="utf-8").write(head + body)
        got = F.load_snp_from_tsv(fn)
        assert isinstance(got, list), name                     # the generic loop ran
        monkeypatch.setenv("XCK_PY_LOADERS", "1")
        assert got == F.load_snp_from_tsv(fn)
        monkeypatch.delenv("XCK_PY_LOADERS")
    bad = str(tmp_path / "bad.tsv")
    open(bad, "w").write(head + "chr1\tfive\tA\tG\t0\t1\n")
    with pytest.raises(ValueError):                            # int("five"), exactly as before
        F.load_snp_from_tsv(bad)
    assert F.load_snp_from_tsv(str(tmp_path / "crlf.tsv"))[1][:2] == ("1", 6)
    empty = str(tmp_path / "empty.tsv")
    open(empty, "w").write(head)
    assert len(F.load_snp_from_tsv(empty)) == 0


def test_plan_units_cuts_an_overweight_contig_at_region_boundaries():
    """a contig heavier than 1 / world of the reads is split at region boundaries into pieces of equal bytes."""
    from xcltk_amd.shard import plan_units
    regions = [("1", 1 + 10000 * i, 9000 + 10000 * i, "g%d" % i) for i in range(100)] + [("2", 100, 5000, "h0"), ("3", 1, 50, "k0")]
    regions[10] = ("1", 100001, 400000, "long")                      # a long gene across many others
    prof = {0: (np.arange(0, 64) * 1000).astype(np.int64)}             # bytes grow linearly along contig 1 (1 Mb = 62 windows)
    units, owner = plan_units([1000.0, 30.0, 5.0], 4, regions, {"1": 0, "2": 1, "3": 2}, prof)
    pieces = [u for u in units if u["contig"] == 0]
    assert len(pieces) == 4 and all(u["window"] is not None for u in pieces)
    seen = np.concatenate([u["regions"] for u in pieces])
    assert sorted(seen.tolist()) == list(range(100))                   # every region of the contig in exactly one piece
    for a, b in zip(pieces[:-1], pieces[1:]):
        assert max(regions[g][1] for g in a["regions"]) <= min(regions[g][1] for g in b["regions"])   # cut in start order
        assert a["window"][1] >= max(regions[g][2] for g in a["regions"]) and b["window"][0] == min(regions[g][1] for g in b["regions"]) - 1
    assert pieces[0]["window"][0] == 0 and pieces[-1]["window"][1] == 0
    loads = np.bincount(owner, weights=[u["weight"] for u in units], minlength=4)
    assert loads.max() / loads.sum() < 0.4                             # 1000 of 1035 on one rank without the split
    # a contig that is not over-weight, or has no byte profile, stays whole
    units2, _ = plan_units([10.0, 9.0, 8.0], 2, regions, {"1": 0, "2": 1, "3": 2}, prof)
    assert all(u["window"] is None for u in units2)
    units3, _ = plan_units([1000.0, 30.0, 5.0], 4, regions, {"1": 0, "2": 1, "3": 2}, {})
    assert len(units3) == 3


def test_position_windows_decode_through_the_linear_index():
    """xck_ingest_opts.tid_beg / tid_end: [0, X) gives exactly the records that start before X; [X, open) starts at the first
    record the .bai linear index reports for X's window - a suffix of the file that holds every record reaching X or beyond."""
    ddir = os.path.join(util.GOLDEN, "datasets", "c1")
    regions, snps = util.load_tables(ddir)
    names = O.contig_table(regions, snps)
    with open(os.path.join(ddir, "barcodes.tsv")) as fp:
        bcs = sorted(x.strip() for x in fp)
    eng = Engine(capi.XCK_MODE_BAF, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB", decode_only=True, n_threads=3)
    bam = os.path.join(ddir, "possorted.bam")
    try:
        cat = lambda it, k: np.concatenate([b[k] for b in it] or [np.zeros(0, np.int64)])
        full = list(eng.decode_bam(bam))
        pos, umi = cat(full, "pos"), cat(full, "umi")
        X = int(np.median(pos))
        left = list(eng.decode_bam(bam, use_index=True, windows={0: (0, X)}))
        assert np.array_equal(cat(left, "pos"), pos[pos < X]) and np.array_equal(cat(left, "umi"), umi[pos < X])
        right = list(eng.decode_bam(bam, use_index=True, windows={0: (X, 0)}))
        rp = cat(right, "pos")
        k = len(pos) - len(rp)
        assert 0 < k < len(pos) and np.array_equal(rp, pos[k:]) and np.array_equal(cat(right, "umi"), umi[k:])
        assert pos[k:].min() <= X and (pos[:k] < X).all()              # starts left of X (the 16 kb window), nothing at / beyond X is missing
        assert (X >> 14) << 14 <= pos[k] + 20000                       # ... and not at the start of the file: within reach of X's window
        prof = eng.contig_byte_profile(bam, 0)
        assert prof is not None and len(prof) >= (int(pos.max()) >> 14) and np.all(np.diff(prof[prof > 0]) >= 0)
    finally:
        eng.close()


def test_io_adapters_build_the_reference_anndata_objects(tmp_path, monkeypatch):
    """rdr/io.py / baf/io.py load_data(): the cell x feature AnnData the reference returns (xcltk/rdr/io.py:14-26, baf/io.py:14-33) -
    built here with the minimal AnnData of oracle/refgen/anndata_standin.py standing in for the `anndata` package (absent from the
    image), on two reference-generated output directories; save_data() writes directories that load back to the same object."""
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(util.GOLDEN), os.pardir, "oracle", "refgen"))
    import anndata_standin
    m = types.ModuleType("anndata")
    m.AnnData = anndata_standin.AnnData
    monkeypatch.setitem(sys.modules, "anndata", m)
    from scipy import io as spio
    from xcltk_amd.baf import io as bio
    from xcltk_amd.rdr import io as rio
    d = os.path.join(util.GOLDEN, "cases", "c1_basefc_default", "expected")
    a = rio.load_data(d)
    feats, cells, mtx = rio.load_matrix_data(d)
    assert a.shape == (len(cells), len(feats)) == mtx.shape and list(a.obs["cell"]) == list(cells["cell"]) and list(a.var["feature"]) == list(feats["feature"])
    assert np.array_equal(np.asarray(a.X), spio.mmread(os.path.join(d, "matrix.mtx")).toarray().T) and int(np.asarray(a.X).sum()) == int(mtx.sum()) > 0
    # save_data writes the object as it stands - cell x feature, like the reference's (rdr/io.py:29-37: its files are the transpose of
    # what load_data reads; kept, it is the reference's contract)
    rio.save_data(a, str(tmp_path / "rdr"))
    assert np.array_equal(spio.mmread(str(tmp_path / "rdr" / "matrix.mtx")).toarray(), np.asarray(a.X))
    assert open(str(tmp_path / "rdr" / "barcodes.tsv")).read() == open(os.path.join(d, "barcodes.tsv")).read()
    assert open(str(tmp_path / "rdr" / "features.tsv")).read() == open(os.path.join(d, "features.tsv")).read()
    d = os.path.join(util.GOLDEN, "cases", "c1_baf_allreg", "expected")
    a = bio.load_data(d)
    feats, cells, mats = bio.load_matrix_data(d)
    assert a.shape == (len(cells), len(feats)) and set(a.layers) >= {"AD", "DP", "OTH"}
    for k in bio.LAYERS:
        assert np.array_equal(np.asarray(a.layers[k]), mats[k].toarray()) and a.layers[k].shape == a.shape
    assert int(np.asarray(a.layers["DP"]).sum()) > 0
    bio.save_data(a, str(tmp_path / "baf"))
    for k in bio.LAYERS:
        assert np.array_equal(spio.mmread(str(tmp_path / "baf" / ("xcltk.%s.mtx" % k))).toarray(), np.asarray(a.layers[k]))
    assert open(str(tmp_path / "baf" / "xcltk.region.tsv")).read() == open(os.path.join(d, "xcltk.region.tsv")).read()
