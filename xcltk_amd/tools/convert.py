"""`xcltk convert`: write a feature list - fixed-size whole-genome bins, or a bed/tsv list re-typed - for
`basefc` / `baf` to count over.  Same options, checks, messages and exit codes as the reference's
xcltk/tools/convert.py:15-86; `--outType features` is an addition (four columns with the bin id as name, the
form the feature loaders of basefc / baf accept; the reference's tsv output has three)."""
import os
import sys
from optparse import OptionParser

from ..config import APP, VERSION
from ..utils.gregion import get_fixsize_regions, load_regions, output_feature_table, output_regions

COMMAND = "convert"


def _die(msg):
    sys.stderr.write(msg)
    sys.exit(1)


def convert_main(argv):
    if len(argv) < 3:
        _die("Welcome to %s %s v%s!\n\nuse -h or --help for help on argument.\n" % (APP, COMMAND, VERSION))
    parser = OptionParser(usage="Usage: %s %s [options]" % (APP, COMMAND))
    parser.add_option("--input", "-i", dest="in_file", default=None, help="Path to input region file.")
    parser.add_option("--inType", "-I", dest="in_type", default=None, help="Input region type, one of bed|tsv.")
    parser.add_option("--output", "-o", dest="out_file", default=None, help="Path to output file; if not set, use stdout.")
    parser.add_option("--outType", "-O", dest="out_type", default="tsv",
                      help="Output region type, one of bed|tsv|features [default: %default]")
    parser.add_option("--binsize", "-B", type="int", dest="bin_size", default=None,
                      help="Fixed size of bin in kb. it will be used when no input file.")
    parser.add_option("--hgver", "-H", type="int", dest="hg_ver", default=38,
                      help="Version of human genome, one of 19|38; set together with @p binsize [default: %default]")
    opt, _ = parser.parse_args(args=argv[2:])
    in_file, in_type, bin_size, hg_ver = opt.in_file, opt.in_type, opt.bin_size, opt.hg_ver
    if not in_file or not in_type:
        in_file = in_type = None
    elif not os.path.isfile(in_file):
        _die("Error: input region file not exist: %s\n" % in_file)
    elif in_type.lower() not in ("bed", "gff", "tsv"):
        _die("Error: input region type should be one of bed|gff|tsv.\n")
    else:
        bin_size = hg_ver = None
        in_type = in_type.lower()
    if in_file is None and (not bin_size or bin_size <= 0 or not hg_ver or hg_ver not in (19, 38)):
        _die("Error: either region file & type or a valid bin size & hg ver should be provided!\n")
    if not opt.out_type:
        _die("Error: out region type should be provided!\n")
    if opt.out_type not in ("bed", "tsv", "features"):
        _die("Error: out region type should be one of bed|tsv.\n")
    if in_type == "gff":
        _die("Error: gff input is not provided by this build (it needs the GTF gene parser).\n")
    regs = get_fixsize_regions(bin_size, hg_ver) if in_file is None else load_regions(in_file, in_type)
    if not regs:
        _die("Error: empty region file or failed to parse regions.\n")
    if opt.out_type == "features":
        if not opt.out_file:
            _die("Error: --outType features needs --output.\n")
        output_feature_table(regs, opt.out_file)
    else:
        output_regions(regs, opt.out_file or None, opt.out_type)
