"""Columnar phased-SNP list.  The library's text parser (csrc/snptext.cpp) returns columns; keeping them as numpy arrays
saves building (and later taking apart) a million Python tuples on the way to the engine's SNP table.  Everywhere else
the object reads like the list of (chrom, pos, ref, alt, ref_hap, alt_hap) tuples the generic loaders return."""
from collections.abc import Sequence

import numpy as np


class SnpTable(Sequence):
    __slots__ = ("names", "chrom_id", "pos", "ref", "alt", "ref_hap", "alt_hap")

    def __init__(self, names, chrom_id, pos, ref, alt, ref_hap, alt_hap):
        self.names = list(names)                         # chromosome names ('chr' stripped), in order of first appearance
        self.chrom_id = np.ascontiguousarray(chrom_id, dtype=np.int32)
        self.pos = np.ascontiguousarray(pos, dtype=np.int64)
        self.ref = np.ascontiguousarray(ref, dtype=np.uint8)          # ASCII codes
        self.alt = np.ascontiguousarray(alt, dtype=np.uint8)
        self.ref_hap = np.ascontiguousarray(ref_hap, dtype=np.int8)
        self.alt_hap = np.ascontiguousarray(alt_hap, dtype=np.int8)

    def __len__(self):
        return int(self.pos.shape[0])

    def _rows(self, sl):
        return list(zip([self.names[i] for i in self.chrom_id[sl].tolist()], self.pos[sl].tolist(),
                        [chr(c) for c in self.ref[sl].tolist()], [chr(c) for c in self.alt[sl].tolist()],
                        self.ref_hap[sl].tolist(), self.alt_hap[sl].tolist()))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._rows(i)
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("SNP index out of range")
        return self._rows(slice(i, i + 1))[0]

    def __iter__(self):
        return iter(self._rows(slice(None)))

    def __eq__(self, other):
        if isinstance(other, SnpTable):
            other = list(other)
        return list(self) == other

    __hash__ = None

    def chroms(self):
        """Chromosome names in order of first appearance (what contig_table() needs)."""
        return list(self.names)

    def positions_by_chrom(self):
        """{chrom: sorted int64 positions} for the SNP -> region pre-join of the front-end."""
        out = {}
        order = np.argsort(self.chrom_id, kind="stable")
        ids = self.chrom_id[order]
        cuts = np.flatnonzero(np.diff(ids)) + 1
        for seg in np.split(order, cuts):
            if len(seg):
                out[self.names[int(self.chrom_id[seg[0]])]] = np.sort(self.pos[seg])
        return out
