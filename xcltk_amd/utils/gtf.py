"""gtf.py - gene records of a GTF / GFF3 annotation as features (the gff branch of `xcltk convert`).

The reference's convert reads genes with load_genes(file, tranTag="", exonTag="") and keeps (chrom, start, stop, gene id)
of every `gene` line (xcltk/utils/gregion.py:60-62, utils/gtf.py:174-259).  Only that is restated here: one pass over the
lines, no transcript / exon model.  Attribute rules as in parse_attribute (utils/gtf.py:101-171): `key=value` (GFF3) or
`key "value"` (GTF) items separated by ';', the gene id is the value of the LAST `ID` or `gene_id` item, '*' when absent."""
import gzip


def gene_id_of(att_str, default="*"):
    gid = default
    for att in att_str.rstrip().split(";"):
        att = att.lstrip(" ")
        if not att:
            continue
        kv = att.split("=") if "=" in att else att.split(" ")
        if len(kv) < 2:
            print("Can't pase this attribute: %s" % att)
            continue
        val = kv[1]
        if val[:1] == '"':
            val = val.split('"')[1]
        if kv[0] in ("ID", "gene_id"):
            gid = val
    return gid


def load_gene_regions(anno_file, comments="#>", gene_tag="gene"):
    """-> [(chrom, start, stop, gene_id)] of every `gene` line, in file order (1-based, both ends inclusive)."""
    opener = gzip.open if (anno_file.endswith(".gz") or anno_file.endswith(".gzip")) else open
    out = []
    with opener(anno_file, "rt") as fp:
        for line in fp:
            if not line or line[0] in comments:
                continue
            f = line.split("\t")
            if len(f) < 8:
                continue
            if f[2] == gene_tag:
                out.append((f[0], int(f[3]), int(f[4]), gene_id_of(f[8]) if len(f) > 8 else "*"))
    return out
