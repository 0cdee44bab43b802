#!/usr/bin/env python3
"""Corrupt BAM files for the sanitizer harness: mutations of the INFLATED record stream (re-compressed into valid BGZF
blocks of random sizes, so the damage reaches the record parser), mutations of the compressed bytes, truncations.
usage: mutate_bams.py SRC.bam OUT_DIR N [seed] [--bai]   (--bai: keep the BAM, corrupt a copy of SRC.bam.bai instead)"""
import gzip, os, random, struct, sys, zlib


def bgzf(raw, rnd):
    out = bytearray()
    i = 0
    while i < len(raw):
        n = rnd.choice([200, 1500, 9000, 30000, 65280])
        chunk = raw[i:i + n]
        i += n
        c = zlib.compressobj(rnd.choice([0, 1, 6]), zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def main():
    src, out_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    seed = int(sys.argv[4]) if len(sys.argv) > 4 and not sys.argv[4].startswith("-") else 1
    os.makedirs(out_dir, exist_ok=True)
    raw = gzip.open(src, "rb").read()
    comp = open(src, "rb").read()
    l_text, = struct.unpack_from("<i", raw, 4)
    n_ref, = struct.unpack_from("<i", raw, 8 + l_text)
    off = 12 + l_text
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, off)
        off += 8 + l_name
    first_rec = off
    if "--bai" in sys.argv:
        bai = open(src + ".bai", "rb").read()
        for k in range(n):
            rnd = random.Random(seed * 7919 + k)
            b = bytearray(bai)
            for _ in range(rnd.choice([1, 2, 8])):
                where = rnd.randrange(len(b))
                if rnd.randrange(2) and where + 4 <= len(b): struct.pack_into("<i", b, where, rnd.choice([-1, 0, 1, 2**31 - 1, 1 << 20, 37450, 5]))
                else: b[where] = rnd.randrange(256)
            if k % 5 == 4:
                b = b[:rnd.randrange(len(b))]
            fn = os.path.join(out_dir, "i%05d.bam" % k)
            open(fn, "wb").write(comp)
            open(fn + ".bai", "wb").write(bytes(b))
        return
    for k in range(n):
        rnd = random.Random(seed * 100003 + k)
        kind = k % 8
        if kind < 5:                                    # damage in the record stream
            b = bytearray(raw)
            for _ in range(rnd.choice([1, 1, 2, 5, 20])):
                where = rnd.randrange(first_rec, len(b)) if kind < 4 else rnd.randrange(0, first_rec)
                how = rnd.randrange(5)
                if how == 0: b[where] ^= 1 << rnd.randrange(8)
                elif how == 1: b[where] = 0xFF
                elif how == 2: b[where] = 0
                elif how == 3 and where + 4 <= len(b): struct.pack_into("<i", b, where, rnd.choice([-1, 0, 1, 2**31 - 1, -2**31, 70000, 1 << 20]))
                else: b[where] = rnd.randrange(256)
            if kind == 3:
                b = b[:rnd.randrange(first_rec, len(b))]   # cut inside a record
            data = bgzf(bytes(b), rnd)
        elif kind == 5:                                 # damage in the compressed bytes / block headers
            c = bytearray(comp)
            for _ in range(rnd.choice([1, 3, 10])):
                c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
            data = bytes(c)
        elif kind == 6:                                 # truncated file
            data = comp[:rnd.randrange(0, len(comp))]
        else:                                           # garbage after a valid prefix
            data = comp[:rnd.randrange(0, len(comp))] + bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 3000)))
        open(os.path.join(out_dir, "m%05d.bam" % k), "wb").write(data)


if __name__ == "__main__":
    main()
