#!/usr/bin/env python3
"""make_convert_goldens.py - CONTAINER-ONLY: outputs of the unmodified reference's `xcltk convert`
(xcltk/tools/convert.py, through run_reference.py) for fixed-size bins and bed/tsv re-typing, kept as fixtures under
tests/golden/convert/ (inputs written here + the reference's outputs; no reference source)."""
import json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GOLD = os.path.join(ROOT, "tests", "golden", "convert")
RUNNER = os.path.join(HERE, "run_reference.py")

BED_IN = "chr1\t0\t1000\tx\nchr1\t999\t2500\ty\n2\t10\t20\nchrX\t5\t5\n"      # last line keeps its newline
TSV_IN = "1\t1\t1000\n1\t1000\t2500\textra\nMT\t7\t9\n"
GTF_IN = ("#!genome-build GRCh38\n"
          'chr1\tHAVANA\tgene\t11869\t14409\t.\t+\t.\tgene_id "ENSG00000223972.5"; gene_type "transcribed_unprocessed_pseudogene"; gene_name "DDX11L1"; level 2;\n'
          'chr1\tHAVANA\ttranscript\t11869\t14409\t.\t+\t.\tgene_id "ENSG00000223972.5"; transcript_id "ENST00000456328.2"; gene_name "DDX11L1";\n'
          'chr1\tHAVANA\texon\t11869\t12227\t.\t+\t.\tgene_id "ENSG00000223972.5"; transcript_id "ENST00000456328.2"; exon_number 1;\n'
          'chr1\tHAVANA\tgene\t14404\t29570\t.\t-\t.\tgene_id "ENSG00000227232.5"; gene_name "WASH7P"\n'
          '2\tsrc\tgene\t100\t2000\t.\t+\t.\tID=gene:G3;Name=three;biotype=protein_coding\n'
          '2\tsrc\tmRNA\t100\t2000\t.\t+\t.\tID=t3;Parent=gene:G3\n'
          'X\tsrc\tgene\t5\t9\t.\t+\t.\tName=noid\n'
          '>a fasta-like comment line\n'
          'X\tsrc\tgene\t50\t90\t.\t+\t.\tID=first;gene_id=second\n'
          'short\tline\n')

CASES = [
    ("bins_1000kb_hg38.tsv", ["-B", "1000", "-H", "38"]),
    ("bins_1000kb_hg19.bed", ["-B", "1000", "-H", "19", "-O", "bed"]),
    ("bins_50000kb_hg38.bed", ["-B", "50000", "-O", "bed"]),
    ("bed_to_tsv.tsv", ["-i", "$BED", "-I", "bed", "-O", "tsv"]),
    ("bed_to_bed.bed", ["-i", "$BED", "-I", "BED", "-O", "bed"]),
    ("tsv_to_bed.bed", ["-i", "$TSV", "-I", "tsv", "-O", "bed"]),
    ("gff_to_tsv.tsv", ["-i", "$GTF", "-I", "gff", "-O", "tsv"]),
    ("gff_to_bed.bed", ["-i", "$GTF", "-I", "GFF", "-O", "bed"]),
]


def main():
    os.makedirs(GOLD, exist_ok=True)
    open(os.path.join(GOLD, "in.bed"), "w").write(BED_IN)
    open(os.path.join(GOLD, "in.tsv"), "w").write(TSV_IN)
    open(os.path.join(GOLD, "in.gtf"), "w").write(GTF_IN)
    meta = []
    for name, argv in CASES:
        out = os.path.join(GOLD, name)
        av = [a.replace("$BED", os.path.join(GOLD, "in.bed")).replace("$TSV", os.path.join(GOLD, "in.tsv")).replace("$GTF", os.path.join(GOLD, "in.gtf")) for a in argv] + ["-o", out]
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fp:
            json.dump(dict(kind="convert", argv=av), fp)
        r = subprocess.run(["/opt/conda/bin/python3.9", RUNNER, fp.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        os.unlink(fp.name)
        if r.returncode != 0:
            raise SystemExit("reference convert failed on %s: %s" % (name, r.stderr[-1500:]))
        meta.append(dict(name=name, argv=argv, lines=sum(1 for _ in open(out))))
        print(name, meta[-1]["lines"], "lines")
    with open(os.path.join(GOLD, "cases.json"), "w") as fp:
        json.dump(dict(reference="hxj5/xcltk v0.5.2 xcltk/tools/convert.py via oracle/refgen/run_reference.py", cases=meta), fp, indent=1)


if __name__ == "__main__":
    main()
