"""`xcltk convert`: write a feature list for `basefc` / `baf` to count over - fixed-size whole-genome bins, or an
existing bed / tsv list in the other form.  Option letters, defaults, checks, messages and exit codes follow the
reference's command (xcltk/tools/convert.py:15-86); `--outType features` is an addition: four columns with the bin id
as the name, which is what the feature loaders accept (the reference's own tsv output has three)."""
import os
import sys
from optparse import OptionParser

from ..config import APP, VERSION
from ..utils import gregion

COMMAND = "convert"
IN_TYPES = ("bed", "gff", "tsv")
OUT_TYPES = ("bed", "tsv", "features")

#          flags                 dest        extra keyword arguments
OPTIONS = (
    (("--input", "-i"),   "in_file",  dict(default=None, help="Path to input region file.")),
    (("--inType", "-I"),  "in_type",  dict(default=None, help="Input region type, one of bed|gff|tsv.")),
    (("--output", "-o"),  "out_file", dict(default=None, help="Path to output file; if not set, use stdout.")),
    (("--outType", "-O"), "out_type", dict(default="tsv", help="Output region type, one of bed|tsv|features [default: %default]")),
    (("--binsize", "-B"), "bin_size", dict(default=None, type="int", help="Fixed size of bin in kb. it will be used when no input file.")),
    (("--hgver", "-H"),   "hg_ver",   dict(default=38, type="int",
                                           help="Version of human genome, one of 19|38; set together with @p binsize [default: %default]")),
)


class _Fail(Exception):
    pass


def _source(opt):
    """-> ("file", path, type) or ("bins", kb, hg); raises _Fail with the message to print."""
    if opt.in_file and opt.in_type:
        if not os.path.isfile(opt.in_file):
            raise _Fail("Error: input region file not exist: %s\n" % opt.in_file)
        kind = opt.in_type.lower()
        if kind not in IN_TYPES:
            raise _Fail("Error: input region type should be one of bed|gff|tsv.\n")
        return "file", opt.in_file, kind
    kb, hg = opt.bin_size, opt.hg_ver
    if not kb or kb <= 0 or hg not in (19, 38):
        raise _Fail("Error: either region file & type or a valid bin size & hg ver should be provided!\n")
    return "bins", kb, hg


def _run(argv):
    parser = OptionParser(usage="Usage: %s %s [options]" % (APP, COMMAND))
    for flags, dest, kw in OPTIONS:
        parser.add_option(*flags, dest=dest, **kw)
    opt, _ = parser.parse_args(args=argv[2:])
    how, x, y = _source(opt)
    if not opt.out_type:
        raise _Fail("Error: out region type should be provided!\n")
    if opt.out_type not in OUT_TYPES:
        raise _Fail("Error: out region type should be one of bed|tsv.\n")
    regs = gregion.get_fixsize_regions(x, y) if how == "bins" else gregion.load_regions(x, y)
    if not regs:
        raise _Fail("Error: empty region file or failed to parse regions.\n")
    if opt.out_type != "features":
        gregion.output_regions(regs, opt.out_file or None, opt.out_type)
    elif opt.out_file:
        gregion.output_feature_table(regs, opt.out_file)
    else:
        raise _Fail("Error: --outType features needs --output.\n")


def convert_main(argv):
    if len(argv) < 3:
        sys.stderr.write("Welcome to %s %s v%s!\n\nuse -h or --help for help on argument.\n" % (APP, COMMAND, VERSION))
        sys.exit(1)
    try:
        _run(argv)
    except _Fail as e:
        sys.stderr.write(str(e))
        sys.exit(1)
