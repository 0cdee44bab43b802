"""oracle/pybam.py - TEST INFRASTRUCTURE ONLY (never imported by the product path).

A tiny, obviously-correct, pure-Python BAM reader that restates the pysam/htslib
behaviour the reference's hot path relies on.  pysam (setup.py:18 `pysam>=0.15.2`)
is a third-party dependency that is absent from /root/reference and from this
image, so what is restated here is the *published* SAM/BAM specification plus
the documented pysam accessor semantics listed in SURVEY.md section 8c:

* ``fetch(contig, start, stop)``: 0-based half-open; yields in file order the
  records on that contig with ``pos < stop`` and ``endpos > start`` where
  ``endpos = pos + reflen(M/D/N/=/X)`` or ``pos + 1`` if the read is unmapped,
  has no CIGAR or that length is 0 (htslib ``bam_endpos``); unknown contig or
  invalid coordinates raise ``ValueError`` (pysam ``parse_region``).
* ``positions``: reference coordinates of M/=/X bases only
  (pysam ``get_reference_positions()``).
* ``cigartuples``: ``[(op, len)]`` or ``None``; ``query_sequence``: decoded from
  the 4-bit codes ``=ACMGRSVTWYHKDBN`` or ``None``.
* ``get_tag``: ``str`` for Z/A/H, ``int`` for integer types, ``float`` for f.

Call sites in the reference that this serves: xcltk/rdr/fc/core.py:47-60,75,
xcltk/rdr/fc/mcount.py:38-40,120, xcltk/baf/fc/core.py:19-32,48,
xcltk/baf/fc/mcount.py:54,113-115,224, xcltk/utils/sam.py:21-39,106,114.

Parity status of this boundary: *unpinned by the reference* (it ships no tests);
pinned here by the SAM spec and by round-tripping the real BAM that ships in
the reference tree (see tests/test_oracle_pybam.py, container-only part).
"""

import bisect
import gzip
import struct

SEQ_NT16 = "=ACMGRSVTWYHKDBN"
# op codes as in xcltk/utils/sam.py:125-134
CIGAR_OPS = "MIDNSHP=XB"
_CONSUMES_REF = (True, False, True, True, False, False, False, True, True, False)
_ALIGNED = (True, False, False, False, False, False, False, True, True, False)

BAM_FUNMAP = 4
_NT16_PAIR = [SEQ_NT16[b >> 4] + SEQ_NT16[b & 0xF] for b in range(256)]


class Record(object):
    """Duck-typed stand-in for pysam.AlignedSegment (only what the hot path touches)."""

    __slots__ = ("tid", "pos", "mapq", "flag", "cigartuples", "query_sequence",
                 "query_name", "tags", "l_seq", "seq_nibbles", "ordinal",
                 "_positions", "_endpos", "next_tid", "next_pos", "tlen", "bin")

    def __init__(self):
        self._positions = None
        self._endpos = None

    # -- pysam accessors ---------------------------------------------------
    @property
    def mapping_quality(self):
        return self.mapq

    def has_tag(self, tag):
        return tag in self.tags

    def get_tag(self, tag):
        if tag not in self.tags:
            raise KeyError("tag '%s' not present" % tag)
        return self.tags[tag][1]

    @property
    def positions(self):
        if self._positions is None:
            out = []
            p = self.pos
            if self.cigartuples:
                for op, l in self.cigartuples:
                    if _ALIGNED[op]:
                        out.extend(range(p, p + l))
                        p += l
                    elif _CONSUMES_REF[op]:
                        p += l
            self._positions = out
        return self._positions

    def get_reference_positions(self):
        return self.positions

    @property
    def reference_end(self):
        return self.endpos

    @property
    def endpos(self):
        """htslib bam_endpos()."""
        if self._endpos is None:
            rlen = 0
            if not (self.flag & BAM_FUNMAP) and self.cigartuples:
                for op, l in self.cigartuples:
                    if _CONSUMES_REF[op]:
                        rlen += l
            else:
                rlen = 1
            if rlen == 0:
                rlen = 1
            self._endpos = self.pos + rlen
        return self._endpos


def _parse_tags(buf, off, end):
    tags = {}
    while off < end:
        tag = buf[off:off + 2].decode("ascii")
        typ = chr(buf[off + 2])
        off += 3
        if typ == "A":
            val = chr(buf[off]); off += 1
        elif typ == "c":
            val, = struct.unpack_from("<b", buf, off); off += 1
        elif typ == "C":
            val = buf[off]; off += 1
        elif typ == "s":
            val, = struct.unpack_from("<h", buf, off); off += 2
        elif typ == "S":
            val, = struct.unpack_from("<H", buf, off); off += 2
        elif typ == "i":
            val, = struct.unpack_from("<i", buf, off); off += 4
        elif typ == "I":
            val, = struct.unpack_from("<I", buf, off); off += 4
        elif typ == "f":
            val, = struct.unpack_from("<f", buf, off); off += 4
        elif typ in "ZH":
            e = buf.index(b"\0", off)
            val = buf[off:e].decode("ascii", errors="replace")
            off = e + 1
        elif typ == "B":
            sub = chr(buf[off])
            n, = struct.unpack_from("<i", buf, off + 1)
            off += 5
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            sz = struct.calcsize(fmt)
            val = list(struct.unpack_from("<%d%s" % (n, fmt), buf, off))
            off += n * sz
        else:
            raise ValueError("unknown tag type %r" % typ)
        if tag not in tags:      # htslib bam_aux_get returns the first occurrence
            tags[tag] = (typ, val)
    return tags


def read_bam(path):
    """Return (refs, records): refs = [(name, length)], records in file order."""
    with gzip.open(path, "rb") as fp:      # BGZF is a series of gzip members
        d = fp.read()
    if d[:4] != b"BAM\x01":
        raise ValueError("not a BAM file: %s" % path)
    l_text, = struct.unpack_from("<i", d, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", d, off)
    off += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", d, off)
        off += 4
        name = d[off:off + l_name - 1].decode("ascii")
        off += l_name
        l_ref, = struct.unpack_from("<i", d, off)
        off += 4
        refs.append((name, l_ref))
    records = []
    n = len(d)
    ordinal = 0
    while off < n:
        block_size, = struct.unpack_from("<i", d, off)
        off += 4
        end = off + block_size
        (tid, pos, l_read_name, mapq, bin_, n_cig, flag, l_seq,
         next_tid, next_pos, tlen) = struct.unpack_from("<iiBBHHHiiii", d, off)
        p = off + 32
        r = Record()
        r.tid, r.pos, r.mapq, r.flag, r.l_seq = tid, pos, mapq, flag, l_seq
        r.next_tid, r.next_pos, r.tlen, r.bin = next_tid, next_pos, tlen, bin_
        r.query_name = d[p:p + l_read_name - 1].decode("ascii")
        p += l_read_name
        if n_cig:
            raw = struct.unpack_from("<%dI" % n_cig, d, p)
            r.cigartuples = [(v & 0xF, v >> 4) for v in raw]
        else:
            r.cigartuples = None
        p += 4 * n_cig
        nb = (l_seq + 1) // 2
        r.seq_nibbles = d[p:p + nb]
        if l_seq:
            r.query_sequence = "".join([_NT16_PAIR[b] for b in r.seq_nibbles])[:l_seq]
        else:
            r.query_sequence = None
        p += nb + l_seq              # skip qualities
        r.tags = _parse_tags(d, p, end)
        r.ordinal = ordinal
        ordinal += 1
        records.append(r)
        off = end
    return refs, records


class BGZFile(object):
    """Stand-in for pysam.BGZFile(fn, "w") (utils/zfile.py:58, utils/csp_io.py:159): write() of bytes, close().  BGZF per the SAM
    spec section 4.1: gzip members of at most 64 KB with the BC extra field, closed by the empty EOF block.  Only the writer
    is needed (the reference reads .gz files with gzip.open); consumers compare the DECOMPRESSED text."""
    _EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

    def __init__(self, fn, mode="w", *a, **k):
        if "w" not in mode:
            raise NotImplementedError("the stand-in BGZFile only writes")
        self._fp = open(fn, "wb")
        self._buf = bytearray()

    def _block(self, data):
        import zlib
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(bytes(data)) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        self._fp.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body
                       + struct.pack("<II", zlib.crc32(bytes(data)) & 0xffffffff, len(data)))

    def write(self, data):
        self._buf += data
        while len(self._buf) >= 0xff00:
            self._block(self._buf[:0xff00]); del self._buf[:0xff00]
        return len(data)

    def close(self):
        if self._fp is None:
            return
        if self._buf:
            self._block(self._buf)
        self._fp.write(self._EOF)
        self._fp.close()
        self._fp = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class AlignmentFile(object):
    """Stand-in for pysam.AlignmentFile(fn, "r") restricted to fetch()."""

    _cache = {}

    def __init__(self, fn, mode="r", **kw):
        key = fn
        if key not in AlignmentFile._cache:
            refs, recs = read_bam(fn)
            by_tid = {}
            for r in recs:
                by_tid.setdefault(r.tid, []).append(r)
            # what the .bai gives htslib: per contig the start positions (coordinate-sorted file) and the running
            # maximum of endpos, so that fetch() can bisect to the first record that can still overlap the window
            index = {}
            for tid, rs in by_tid.items():
                pos = [r.pos for r in rs]
                if any(pos[i] > pos[i + 1] for i in range(len(pos) - 1)):
                    continue                                   # not sorted: fetch() scans
                run, top = [], -1 << 62
                for r in rs:
                    top = max(top, r.endpos)
                    run.append(top)
                index[tid] = (pos, run)
            AlignmentFile._cache[key] = (refs, recs, by_tid, index)
        self.refs, self.records, self.by_tid, self._index = AlignmentFile._cache[key]
        self.references = tuple(n for n, _ in self.refs)
        self.lengths = tuple(l for _, l in self.refs)
        self._tid = {n: i for i, n in enumerate(self.references)}
        self.filename = fn

    def get_tid(self, contig):
        return self._tid.get(contig, -1)

    def fetch(self, contig=None, start=None, stop=None):
        if contig is None:
            return iter(self.records)
        tid = self.get_tid(contig)
        if tid < 0:
            raise ValueError("invalid contig `%s`" % contig)
        # pysam parse_region() rejects coordinates beyond MAX_POS: (1 << 31) - 1 in current pysam (libchtslib.pyx), smaller in
        # some 0.15.x builds (BAI's 2^29); version dependent and beyond any real contig - the oracle and the engine accept int32
        max_pos = (1 << 31) - 1
        rstart = 0 if start is None else int(start)
        rstop = max_pos if stop is None else int(stop)
        if rstart > rstop:
            raise ValueError("invalid coordinates: start (%i) > stop (%i)" % (rstart, rstop))
        if not 0 <= rstart < max_pos:
            raise ValueError("start out of range (%i)" % rstart)
        if not 0 <= rstop <= max_pos:
            raise ValueError("stop out of range (%i)" % rstop)
        recs = self.by_tid.get(tid, [])
        if tid in self._index:
            pos, run = self._index[tid]
            recs = recs[bisect.bisect_right(run, rstart):bisect.bisect_left(pos, rstop)]
        return iter([r for r in recs if r.pos < rstop and r.endpos > rstart])

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
