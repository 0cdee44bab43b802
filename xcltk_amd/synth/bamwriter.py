"""bamwriter.py - minimal BGZF / BAM / BAI writer used by the synthetic-input generator.

Written from the SAM/BAM specification (SAMv1 section 4, "The BAM format"); no htslib.
Only what the counting engine's inputs need: header with @SQ lines, alignment
records with CIGAR, 4-bit sequence, qualities and Z/i tags, BGZF blocks with
either record-aligned flushing (what htslib's writer does) or fully packed
64 KiB blocks (records straddle blocks; exercises the decoder's carry-over).
"""

import struct
import zlib

import numpy as np

SEQ_NT16 = "=ACMGRSVTWYHKDBN"
_NT16_CODE = {c: i for i, c in enumerate(SEQ_NT16)}
_NT16_CODE.update({c.lower(): i for i, c in enumerate(SEQ_NT16) if c != "="})
CIGAR_OPS = "MIDNSHP=XB"
_CIG_CODE = {c: i for i, c in enumerate(CIGAR_OPS)}
_CONSUMES_REF = (1, 0, 1, 1, 0, 0, 0, 1, 1, 0)

BGZF_MAX_PAYLOAD = 0xff00        # htslib BGZF_BLOCK_SIZE
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg, end):
    """SAMv1 section 5.3 (C code in the spec), 0-based half-open [beg, end)."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(cigar):
    """'30M40N30M' or [(op, len)] -> list of (op_code, len)."""
    if cigar is None or cigar == "*" or cigar == "":
        return []
    if not isinstance(cigar, str):
        return [(int(op), int(l)) for op, l in cigar]
    out = []
    num = ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            out.append((_CIG_CODE[ch], int(num)))
            num = ""
    return out


def encode_seq(seq):
    """str -> BAM 4-bit packed bytes (high nibble first)."""
    n = len(seq)
    codes = np.fromiter((_NT16_CODE.get(c, 15) for c in seq), dtype=np.uint8, count=n)
    if n & 1:
        codes = np.append(codes, np.uint8(0))
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8).tobytes()


def pack_tag(tag, value, typ=None):
    t = tag.encode("ascii")
    if typ is None:
        typ = "Z" if isinstance(value, str) else "i"
    if typ == "Z":
        return t + b"Z" + value.encode("ascii") + b"\0"
    if typ == "A":
        return t + b"A" + value.encode("ascii")[:1]
    if typ == "H":
        return t + b"H" + value.encode("ascii") + b"\0"
    if typ == "B":                                            # value = (subtype, [numbers])
        sub, arr = value
        f = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        return t + b"B" + sub.encode("ascii") + struct.pack("<I", len(arr)) + struct.pack("<%d%s" % (len(arr), f), *arr)
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}[typ]
    return t + typ.encode("ascii") + struct.pack(fmt, value)


def pack_record(tid, pos, qname, flag, mapq, cigar, seq, tags=(), qual=None,
                next_tid=-1, next_pos=-1, tlen=0):
    """One BAM alignment record including its leading block_size field.

    cigar: string or [(op, len)]; seq: str ('' for '*'); tags: iterable of
    (tag, value) or (tag, value, type).
    """
    cig = parse_cigar(cigar)
    rlen = sum(l for op, l in cig if _CONSUMES_REF[op])
    unmapped = bool(flag & 4)
    end = pos + (rlen if (rlen and not unmapped) else 1)
    bin_ = reg2bin(pos if pos >= 0 else 0, end if pos >= 0 else 1)
    name = qname.encode("ascii") + b"\0"
    l_seq = len(seq)
    body = [struct.pack("<iiBBHHHiiii", tid, pos, len(name), mapq, bin_, len(cig), flag,
                        l_seq, next_tid, next_pos, tlen), name]
    if cig:
        body.append(struct.pack("<%dI" % len(cig), *[(l << 4) | op for op, l in cig]))
    body.append(encode_seq(seq))
    if qual is None:
        body.append(b"\xff" * l_seq)
    else:
        body.append(bytes(qual))
    for t in tags:
        body.append(pack_tag(*t))
    blob = b"".join(body)
    return struct.pack("<i", len(blob)) + blob, end


def bgzf_block(payload, level=6):
    assert len(payload) <= 0x10000
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(payload) + co.flush()
    bsize = len(cdata) + 25        # total block size - 1
    if bsize > 0xffff:             # incompressible: store
        co = zlib.compressobj(0, zlib.DEFLATED, -15)
        cdata = co.compress(payload) + co.flush()
        bsize = len(cdata) + 25
        assert bsize <= 0xffff
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, bsize)
    return hdr + cdata + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))


class BamWriter(object):
    """Coordinate-sorted BAM writer (caller supplies records already sorted).

    align_records=True flushes a BGZF block when the next record would not fit
    (htslib behaviour); False packs records across block boundaries.
    Also accumulates a BAI index (bins, linear index, per-contig counts) that
    write_index() emits.
    """

    def __init__(self, path, refs, header_text=None, align_records=True, level=6,
                 block_payload=BGZF_MAX_PAYLOAD):
        self.path = path
        self.refs = list(refs)
        self.fp = open(path, "wb")
        self.align = align_records
        self.level = level
        self.block_payload = block_payload
        self.buf = bytearray()
        self.coff = 0                 # compressed offset of the block being filled
        self.n_records = 0
        if header_text is None:
            header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
                "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in self.refs)
        ht = header_text.encode("ascii")
        hdr = [b"BAM\x01", struct.pack("<i", len(ht)), ht, struct.pack("<i", len(self.refs))]
        for n, l in self.refs:
            nb = n.encode("ascii") + b"\0"
            hdr.append(struct.pack("<i", len(nb)) + nb + struct.pack("<i", l))
        self._write_bytes(b"".join(hdr))
        self._flush()                 # header in its own block(s), as samtools does
        # index state
        self._bins = [dict() for _ in self.refs]      # tid -> bin -> [[beg_voff, end_voff], ...]
        self._lin = [dict() for _ in self.refs]       # tid -> window -> min voff
        self._meta = [[None, None, 0, 0] for _ in self.refs]  # off_beg, off_end, n_mapped, n_unmapped
        self._n_no_coor = 0

    # -- BGZF --------------------------------------------------------------
    def _voff(self):
        return (self.coff << 16) | len(self.buf)

    def _flush(self):
        while self.buf:
            chunk = bytes(self.buf[:self.block_payload])
            del self.buf[:self.block_payload]
            blk = bgzf_block(chunk, self.level)
            self.fp.write(blk)
            self.coff += len(blk)

    def _write_bytes(self, b):
        self.buf += b
        while len(self.buf) >= self.block_payload:
            chunk = bytes(self.buf[:self.block_payload])
            del self.buf[:self.block_payload]
            blk = bgzf_block(chunk, self.level)
            self.fp.write(blk)
            self.coff += len(blk)

    # -- records -----------------------------------------------------------
    def write(self, tid, pos, qname, flag, mapq, cigar, seq, tags=(), qual=None,
              next_tid=-1, next_pos=-1, tlen=0):
        blob, end = pack_record(tid, pos, qname, flag, mapq, cigar, seq, tags, qual,
                                next_tid, next_pos, tlen)
        self.write_raw(blob, tid, pos, end, flag)

    def write_raw(self, blob, tid, pos, end, flag):
        if self.align and self.buf and len(self.buf) + len(blob) > self.block_payload:
            self._flush()
        v0 = self._voff()
        self._write_bytes(blob)
        v1 = self._voff()
        self.n_records += 1
        if tid < 0:
            self._n_no_coor += 1
            return
        b = reg2bin(max(pos, 0), max(end, 1))
        chunks = self._bins[tid].setdefault(b, [])
        if chunks and chunks[-1][1] == v0:
            chunks[-1][1] = v1
        else:
            chunks.append([v0, v1])
        lin = self._lin[tid]
        for w in range(max(pos, 0) >> 14, (max(end, 1) - 1 >> 14) + 1):
            if w not in lin:
                lin[w] = v0
        m = self._meta[tid]
        if m[0] is None:
            m[0] = v0
        m[1] = v1
        if flag & 4:
            m[3] += 1
        else:
            m[2] += 1

    def close(self):
        self._flush()
        self.fp.write(BGZF_EOF)
        self.fp.close()

    def write_index(self, path=None):
        path = path or (self.path + ".bai")
        out = [b"BAI\x01", struct.pack("<i", len(self.refs))]
        for tid in range(len(self.refs)):
            bins = self._bins[tid]
            m = self._meta[tid]
            n_bin = len(bins) + (1 if m[0] is not None else 0)
            out.append(struct.pack("<i", n_bin))
            for b in sorted(bins):
                ch = bins[b]
                out.append(struct.pack("<Ii", b, len(ch)))
                for v0, v1 in ch:
                    out.append(struct.pack("<QQ", v0, v1))
            if m[0] is not None:     # samtools pseudo-bin 37450
                out.append(struct.pack("<Ii", 37450, 2))
                out.append(struct.pack("<QQ", m[0], m[1]))
                out.append(struct.pack("<QQ", m[2], m[3]))
            lin = self._lin[tid]
            n_intv = (max(lin) + 1) if lin else 0
            out.append(struct.pack("<i", n_intv))
            last = 0
            vals = []
            for w in range(n_intv):   # htslib fills gaps with the next/previous offset
                if w in lin:
                    last = lin[w]
                vals.append(last)
            # back-fill leading zeros like htslib (offset of the first following window)
            nxt = 0
            for w in range(n_intv - 1, -1, -1):
                if w in lin:
                    nxt = lin[w]
                elif vals[w] == 0:
                    vals[w] = nxt
            out.append(struct.pack("<%dQ" % n_intv, *vals) if n_intv else b"")
        out.append(struct.pack("<Q", self._n_no_coor))
        with open(path, "wb") as fp:
            fp.write(b"".join(out))
