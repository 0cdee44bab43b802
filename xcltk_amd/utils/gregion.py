"""Fixed-size genome bins and bed/tsv region lists: the feature tables that feed `basefc` / `baf` when
the features are bins instead of genes.

Same results as the reference's xcltk/utils/gregion.py: `chr2reg` :104-139, `get_fixsize_reg_from_input_len`
:142-165, `get_fixsize_reg_from_sam_header` :168-197, `get_fixsize_regions` :200-227, `load_regions` :40-86
(bed and tsv; the gff branch goes through the GTF gene parser, which is outside this path), `output_regions`
:230-271.  Bins are 1-based, inclusive, `bin_size` wide, and the last bin of a contig is NOT clipped to the
contig length (:133-136), exactly like the reference.  Contig lengths for the header variant come from the
engine's own BAM decoder (xck_bam_open) instead of pysam."""
import gzip
import os
import sys

# GRCh37 / GRCh38 primary assembly lengths, chr1..22, X, Y (public assembly reports; the reference keeps the
# same two tables at utils/gregion.py:12-20)
CHROM_LEN_HG19 = (249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022,
                  141213431, 135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753,
                  81195210, 78077248, 59128983, 63025520, 48129895, 51304566, 155270560, 59373566)
CHROM_LEN_HG38 = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                  138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                  83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415)
CHROM_NAMES = tuple(str(i) for i in range(1, 23)) + ("X", "Y")


class Region(object):
    """chrom, start, end (1-based, both inclusive), id."""
    __slots__ = ("chrom", "start", "end", "id")

    def __init__(self, chrom=None, start=None, end=None, _id=None):
        self.chrom, self.start, self.end, self.id = chrom, start, end, _id


def _is_gz(fn):
    return os.path.splitext(fn)[1] in (".gz", ".gzip")


def load_regions(reg_file, reg_type):
    """bed (0-based start), tsv (1-based start) -> [Region] with id "chrom:start-end", or gff (gene lines, id = gene id); None on any error."""
    if not reg_file or not os.path.isfile(reg_file) or not reg_type:
        return None
    reg_type = reg_type.lower()
    if reg_type == "gff":                               # gene records of a GTF / GFF3 annotation (utils/gregion.py:60-62)
        from .gtf import load_gene_regions
        return [Region(c, s, e, gid) for c, s, e, gid in load_gene_regions(reg_file)]
    if reg_type not in ("bed", "tsv"):
        return None
    with (gzip.open(reg_file, "rt") if _is_gz(reg_file) else open(reg_file, "r")) as fp:
        lines = fp.readlines()
    shift = 1 if reg_type == "bed" else 0
    out = []
    for no, ln in enumerate(lines, 1):
        cols = ln[:-1].split("\t")               # like the reference: the last character goes, newline or not
        try:
            start, end = int(cols[1]) + shift, int(cols[2])
        except (IndexError, ValueError) as e:
            print("Error: invalid bed record in No.%d line: %s" % (no, str(e)))
            return None
        out.append(Region(cols[0], start, end, "%s:%d-%d" % (cols[0], start, end)))
    return out


def bed2reg(bed_file):
    return load_regions(bed_file, "bed")


def tsv2reg(tsv_file):
    return load_regions(tsv_file, "tsv")


def chr2reg(chrom_name, chrom_len, bin_size):
    """ceil(chrom_len / bin_size) bins of bin_size bp; None for non-numeric or non-positive input."""
    try:
        chrom_len, bin_size = int(chrom_len), int(bin_size)
    except (TypeError, ValueError):
        return None
    if chrom_len <= 0 or bin_size <= 0:
        return None
    n_bins = -(-chrom_len // bin_size)
    return [Region(chrom_name, k * bin_size + 1, (k + 1) * bin_size, "%s:%d-%d" % (chrom_name, k * bin_size + 1, (k + 1) * bin_size))
            for k in range(n_bins)]


def get_fixsize_reg_from_input_len(chroms, bin_size):
    """chroms: {name: length} in the order to emit; bin_size in kb."""
    out = []
    for name, length in chroms.items():
        bins = chr2reg(name, length, bin_size * 1000)
        if bins is None:
            print("Error: cannot split chromsome %s to bins!" % name)
            return None
        out.extend(bins)
    return out


def get_fixsize_reg_from_sam_header(chr_names, bin_size, sam_file):
    """Bins over the named contigs with the lengths of the BAM header; a name is retried with "chr" added or
    removed, and an unknown contig gives None (reference :189-193)."""
    from .. import capi
    refs = dict(capi.bam_references(sam_file))
    chroms = {}
    for name in chr_names:
        if name not in refs:
            name = name[3:] if name.startswith("chr") else "chr" + name
            if name not in refs:
                return None
        chroms[name] = refs[name]
    return get_fixsize_reg_from_input_len(chroms, bin_size)


def get_fixsize_regions(bin_size, hg_ver):
    """Whole-genome bins (bin_size in kb) for hg19 or hg38, contigs named 1..22, X, Y."""
    try:
        hg_ver = int(hg_ver)
    except (TypeError, ValueError):
        return None
    lengths = {19: CHROM_LEN_HG19, 38: CHROM_LEN_HG38}.get(hg_ver)
    if lengths is None:
        return None
    return get_fixsize_reg_from_input_len(dict(zip(CHROM_NAMES, lengths)), bin_size)


def output_regions(reg_list, fname, ftype):
    """bed: chrom, start-1, end, id; tsv: chrom, start, end (three columns, as the reference writes them).
    fname None -> stdout.  Returns 0, or -3 for an unknown type."""
    if ftype not in ("bed", "tsv"):
        return -3
    if ftype == "bed":
        text = "".join("%s\t%d\t%d\t%s\n" % (r.chrom, r.start - 1, r.end, r.id) for r in reg_list)
    else:
        text = "".join("%s\t%d\t%d\n" % (r.chrom, r.start, r.end) for r in reg_list)
    if not fname:
        sys.stdout.write(text)
    elif _is_gz(fname):
        with gzip.open(fname, "wt") as fp:       # the reference opens "wb" and writes str, which raises: text mode here
            fp.write(text)
    else:
        with open(fname, "w") as fp:
            fp.write(text)
    return 0


def reg2bed(reg_list, bed_file):
    return output_regions(reg_list, bed_file, "bed")


def reg2tsv(reg_list, tsv_file):
    return output_regions(reg_list, tsv_file, "tsv")


def output_feature_table(reg_list, fname):
    """Four columns - chrom, start, end, id - which is what the feature loaders of basefc / baf take
    (rdr/fc/utils.py:10-45 needs a name column; `output_regions(..., "tsv")` writes none)."""
    with (gzip.open(fname, "wt") if _is_gz(fname) else open(fname, "w")) as fp:
        fp.write("".join("%s\t%d\t%d\t%s\n" % (r.chrom, r.start, r.end, r.id) for r in reg_list))
    return 0
