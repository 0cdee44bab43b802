"""Plain / gzip / bgzip text files (the reference's ZFile, xcltk/utils/zfile.py:14-133).

Reading: plain or gzip (a BGZF file is a series of gzip members, so gzip reads it - the reference reads BGZF through gzip
for the same reason, zfile.py:3-8).  Writing: ZF_F_BGZIP goes through this module's own BGZF writer (`BGZFile`: 64 KiB
blocks with the BC extra field, CRC32, the 28-byte EOF block - SAMv1 section 4.1) where the reference uses pysam.BGZFile
(zfile.py:58); the files are readable by bgzip / tabix / htslib.
"""
import gzip
import struct
import zlib

ZF_F_PLAIN = 0
ZF_F_GZIP = 1
ZF_F_BGZIP = 2
ZF_F_AUTO = 3
ZF_BUFSIZE = 1048576   # 1M

_BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
_BGZF_PAYLOAD = 0xff00


class BGZFile(object):
    """Write-only BGZF stream (bytes in, blocked gzip out)."""

    def __init__(self, file_name, mode="w", level=6):
        if "r" in mode:
            raise ValueError("BGZFile is write-only here; read BGZF files with gzip")
        self.fp = open(file_name, "wb")
        self.level = level
        self.buf = bytearray()

    def _block(self, payload):
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        data = co.compress(payload) + co.flush()
        if len(data) + 26 > 0x10000:                         # incompressible: store
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            data = co.compress(payload) + co.flush()
        hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(data) + 25)
        self.fp.write(hdr + data + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))

    def write(self, data):
        if isinstance(data, str):
            data = data.encode("utf8")
        self.buf += data
        while len(self.buf) >= _BGZF_PAYLOAD:
            self._block(bytes(self.buf[:_BGZF_PAYLOAD]))
            del self.buf[:_BGZF_PAYLOAD]
        return len(data)

    def close(self):
        if self.fp:
            if self.buf:
                self._block(bytes(self.buf))
                self.buf = bytearray()
            self.fp.write(_BGZF_EOF)
            self.fp.close()
            self.fp = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ZFile(object):
    """File object wrapper for plain / gzip / bgzip (same surface as the reference's ZFile)."""

    def __init__(self, file_name, mode, file_type, is_bytes=False, encoding=None):
        self.file_name, self.mode, self.file_type = file_name, mode, file_type
        self.is_bytes = is_bytes
        self.encoding = encoding if encoding else "utf8"
        self.buf = b"" if is_bytes else ""
        if file_type == ZF_F_AUTO:
            low = file_name.lower()
            file_type = ZF_F_GZIP if (low.endswith(".gz") or low.endswith(".gzip") or low.endswith(".bgz")) else ZF_F_PLAIN
        if file_type == ZF_F_PLAIN:
            self.fp = open(file_name, mode)
        elif file_type == ZF_F_GZIP or (file_type == ZF_F_BGZIP and "r" in mode):
            self.fp = gzip.open(file_name, mode)
        elif file_type == ZF_F_BGZIP:
            self.fp = BGZFile(file_name, mode)
        else:
            raise ValueError("invalid file type")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __iter__(self):
        return iter(self.fp)

    def __next__(self):
        line = self.readline()
        if not line:
            raise StopIteration()
        return line

    def close(self):
        if self.fp:
            if self.buf:
                self.fp.write(self.buf)
                self.buf = None
            self.fp.close()
            self.fp = None

    def read(self, size=None):
        return self.fp.read() if size is None else self.fp.read(size)

    def readline(self, size=None):
        return self.fp.readline() if size is None else self.fp.readline(size)

    def readlines(self, size=None):
        return self.fp.readlines() if size is None else self.fp.readlines(size)

    def write(self, data):
        self.buf += data
        if len(self.buf) >= ZF_BUFSIZE:
            ret = self.fp.write(self.buf)
            self.buf = b"" if self.is_bytes else ""
            return ret
        return len(data)


def zopen(file_name, mode="rt", file_type=None, is_bytes=False, encoding=None):
    if not file_name:
        raise OSError()
    if file_type is None:
        file_type = ZF_F_AUTO
    return ZFile(file_name, mode, file_type, is_bytes, encoding)
