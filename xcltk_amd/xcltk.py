"""Command-line dispatcher, same shape as the reference's (xcltk/xcltk.py:16-53) for the two
commands on the accelerated hot path, plus `convert` (fixed-size bins as features).  `fixref` is a
VCF utility outside that path and is not provided."""

import sys

from .config import APP, ENGINE, VERSION


def _usage(fp=sys.stdout):
    fp.write("\n"
             "Program: %s (Toolkit for XClone Preprocessing; %s)\n" % (APP, ENGINE) +
             "Version: %s\n" % VERSION +
             "\n"
             "Usage:   %s <command> [options]\n" % APP +
             "\n"
             "Commands:\n"
             "  -- BAF calculation\n"
             "     baf              Preprocessing pipeline for XClone BAF (GPU feature counting).\n"
             "\n"
             "  -- RDR calculation\n"
             "     basefc           Basic feature counting (GPU).\n"
             "\n"
             "  -- Tools\n"
             "     convert          Convert between different formats of genomic features (fixed-size bins).\n"
             "\n"
             "  -- Others\n"
             "     -h, --help       Print this message and exit.\n"
             "     -V, --version    Print version and exit.\n"
             "\n")


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < 2:
        _usage()
        sys.exit(0)
    command = argv[1]
    if command == "basefc":
        from .rdr.fc.main import fc_main
        return fc_main(argv)
    if command == "baf":
        from .baf.pipeline import pipeline_main
        return pipeline_main(argv)
    if command in ("-h", "--help"):
        _usage()
        sys.exit(0)
    if command in ("-V", "--version"):
        sys.stderr.write("%s\n" % VERSION)
        sys.exit(0)
    if command == "convert":
        from .tools.convert import convert_main
        return convert_main(argv)
    if command in ("fixref",):
        sys.stderr.write("Error: command '%s' is outside the accelerated hot path and not provided by %s\n" % (command, ENGINE))
        sys.exit(1)
    sys.stderr.write("Error: wrong command '%s'\n" % command)
    sys.exit(1)
