"""Genomic-range helpers the hot path needs (reference xcltk/utils/grange.py:8-27,263-264)."""


def format_chrom(chrom):
    """Strip a leading 'chr' (any case), as the reference does for every region / SNP."""
    return chrom[3:] if chrom.lower().startswith("chr") else chrom


class Region(object):
    """1-based region, `end` exclusive internally (file end + 1), chrom stripped of 'chr'."""
    __slots__ = ("chrom", "start", "end", "name")

    def __init__(self, chrom, start, end, name=None):
        self.chrom = format_chrom(chrom)
        self.start = start
        self.end = end
        self.name = name

    def get_id(self):
        if self.name is None:
            self.name = "%s_%d_%d" % (self.chrom, self.start, self.end)
        return self.name
