"""Plain / gzip text readers (the reference's ZFile, xcltk/utils/zfile.py:14-133, reduced to
what the hot path reads: region, barcode, sample and SNP files)."""
import gzip


def zopen(fn, mode="rt"):
    low = fn.lower()
    if low.endswith(".gz") or low.endswith(".gzip") or low.endswith(".bgz"):
        return gzip.open(fn, mode)
    return open(fn, mode)
