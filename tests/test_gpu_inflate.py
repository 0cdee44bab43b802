"""The GPU share of the BGZF inflate (csrc/inflate_dev.hip, csrc/bam.cpp; XCK_GPU_INFLATE): the matrices of a run whose chunks are
partly inflated on the device equal those of the host-only run bit for bit - for blocks written by zlib and by this repo's fast
compressor -, the statistics say that the device took part, parked slots serve the next reader, and asking for CRC checks keeps the
inflate on the host.  Reference boundary: the inflate inside pysam / htslib's fetch (xcltk/rdr/fc/core.py:73-76)."""
import os
import subprocess

import numpy as np
import pytest

import util
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOBS = ("XCK_GPU_INFLATE", "XCK_GPU_INFLATE_DEPTH", "XCK_GPU_INFLATE_MIN_MB", "XCK_CHUNK_BYTES", "XCK_GPU_INFLATE_LDS_RING")


@pytest.fixture
def knob_env():
    saved = {k: os.environ.get(k) for k in KNOBS}
    yield os.environ
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _make_bam(tmp, n_reads, level):
    regions, snps, names = soa.make_tables(4000, 40000, soa.HG38_LENGTHS, seed=2)
    rng = np.random.default_rng(7)
    bcs = sorted({"".join("ACGT"[i] for i in rng.integers(0, 4, 16)) + "-1" for _ in range(500)})
    open(os.path.join(tmp, "contigs.tsv"), "w").write("".join("chr%s\t%d\n" % (n, l) for n, l in zip(names, soa.HG38_LENGTHS)))
    open(os.path.join(tmp, "regions.tsv"), "w").write("".join("chr%s\t%d\t%d\t%s\n" % r for r in regions))
    open(os.path.join(tmp, "barcodes.tsv"), "w").write("".join(b + "\n" for b in bcs))
    bam = os.path.join(tmp, "l%d.bam" % level)
    subprocess.check_call([os.path.join(ROOT, "xcltk_amd", "csrc", "xck_synth_bam"), bam, os.path.join(tmp, "contigs.tsv"), os.path.join(tmp, "regions.tsv"),
                           os.path.join(tmp, "barcodes.tsv"), str(n_reads), "11", "8", str(level)], stderr=subprocess.DEVNULL)
    return bam, regions, snps, names, bcs


def _count(bam, regions, snps, names, bcs, flags=0, passes=1):
    eng = Engine(capi.XCK_MODE_BOTH, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB", min_include=0.9,
                 min_count=1, min_maf=0, no_dup_hap=True, min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, n_threads=8, flags=flags)
    try:
        out = None
        for _ in range(passes):                                   # (a second pass opens a second reader: parked slots)
            eng.reset()
            n = eng.ingest_bam(bam)
            got = eng.finish()
            got = {k: tuple(np.array(a) for a in v) for k, v in got.items()}
            st = eng.stats()
            assert out is None or all(np.array_equal(a, b) for k in got for a, b in zip(got[k], out[k]))
            out = got
    finally:
        eng.close()
    return n, out, st


@pytest.mark.parametrize("level", [6, 0])
def test_device_share_of_the_inflate_changes_nothing(level, knob_env, tmp_path):
    bam, regions, snps, names, bcs = _make_bam(str(tmp_path), 1500000, level)
    knob_env["XCK_CHUNK_BYTES"] = str(6 << 20)                    # ~90 blocks per chunk: dozens of chunks, each large enough for the device
    knob_env["XCK_GPU_INFLATE_MIN_MB"] = "0"                      # (auto mode leaves files below 96 MB to the host)
    knob_env["XCK_GPU_INFLATE"] = "0"
    n0, host, st0 = _count(bam, regions, snps, names, bcs)
    assert n0 == 1500000 and st0["gpu_inflate_chunks"] == 0 and len(host["count"][0]) > 10000 and len(host["dp"][0]) > 100
    # (the last two: the kernel's optional LDS ring of the last 4 / 8 KB of output, csrc/inflate_dev.hip)
    for share, depth, ring in (("50", "4", "0"), ("auto", "3", "0"), ("100", "4", "0"), ("100", "4", "1"), ("50", "4", "2")):
        knob_env["XCK_GPU_INFLATE"], knob_env["XCK_GPU_INFLATE_DEPTH"], knob_env["XCK_GPU_INFLATE_LDS_RING"] = share, depth, ring
        n1, dev, st1 = _count(bam, regions, snps, names, bcs, passes=2)
        assert n1 == n0 and st1["gpu_inflate_chunks"] >= 5, (share, ring, st1["gpu_inflate_chunks"])
        for k in host:
            for a, b in zip(host[k], dev[k]):
                assert np.array_equal(a, b), (share, ring, k)
    # CRC verification wanted: the inflate stays on the host (the device does not compute the checksum)
    knob_env["XCK_GPU_INFLATE"] = "50"
    n2, crc, st2 = _count(bam, regions, snps, names, bcs, flags=capi.XCK_F_VERIFY_CRC)
    assert n2 == n0 and st2["gpu_inflate_chunks"] == 0
    for k in host:
        for a, b in zip(host[k], crc[k]):
            assert np.array_equal(a, b)


def _engine(regions, snps, names, bcs):
    return Engine(capi.XCK_MODE_BOTH, names, regions, len(bcs), snps=snps, barcodes=bcs, cell_tag="CB", umi_tag="UB", min_include=0.9,
                  min_count=1, min_maf=0, no_dup_hap=True, min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True, n_threads=8)


@pytest.mark.parametrize("share", ["0", "100"])
def test_sliced_ingest_equals_one_call(share, knob_env, tmp_path):
    """xck_ingest_opts.pause_records: the parse and the push of a chunk run behind the decoder's coordinator (csrc/bam.cpp: IngestJob,
    Pusher); a call that returns at a chunk boundary has drained them, so slices of any size give the matrices of one call."""
    bam, regions, snps, names, bcs = _make_bam(str(tmp_path), 600000, 6)
    knob_env["XCK_CHUNK_BYTES"] = str(3 << 20)
    knob_env["XCK_GPU_INFLATE_MIN_MB"] = "0"
    knob_env["XCK_GPU_INFLATE"] = share
    n0, whole, _ = _count(bam, regions, snps, names, bcs)
    eng = _engine(regions, snps, names, bcs)
    try:
        st = eng.open_stream(bam)
        seen = []
        while not st.done:
            n, done = st.advance(70000)
            assert n > (seen[-1] if seen else 0) or done
            seen.append(n)
        st.close()
        assert seen[-1] == n0 and len(seen) >= 5
        got = eng.finish()
        for k in whole:
            for a, b in zip(whole[k], got[k]):
                assert np.array_equal(np.array(b), a), k
    finally:
        eng.close()


def test_damaged_file_ends_the_ingest_with_an_error_and_the_handle_lives_on(knob_env, tmp_path):
    """A BGZF block whose DEFLATE stream is damaged in the middle of the file: the device leaves it to the host (non-zero status), the
    host decoder reports it, xck_ingest_bam returns XCK_E_IO after draining what was in flight - and after xck_reset the same handle
    counts the intact file correctly."""
    bam, regions, snps, names, bcs = _make_bam(str(tmp_path), 600000, 6)
    knob_env["XCK_CHUNK_BYTES"] = str(3 << 20)
    knob_env["XCK_GPU_INFLATE_MIN_MB"] = "0"
    knob_env["XCK_GPU_INFLATE"] = "100"
    n0, whole, _ = _count(bam, regions, snps, names, bcs)
    raw = bytearray(open(bam, "rb").read())
    at = len(raw) * 2 // 3
    raw[at:at + 64] = bytes(64)                                   # zeros in the middle of some block's compressed stream
    bad = os.path.join(str(tmp_path), "damaged.bam")
    open(bad, "wb").write(raw)
    eng = _engine(regions, snps, names, bcs)
    try:
        with pytest.raises(Exception) as ei:
            eng.ingest_bam(bad)
        assert getattr(ei.value, "code", capi.XCK_E_IO) in (capi.XCK_E_IO, capi.XCK_E_ARG), ei.value
        eng.reset()
        assert eng.ingest_bam(bam) == n0
        got = eng.finish()
        for k in whole:
            for a, b in zip(whole[k], got[k]):
                assert np.array_equal(np.array(b), a), k
    finally:
        eng.close()
