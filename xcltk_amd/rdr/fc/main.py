"""`xcltk basefc` - feature x cell UMI / read counting on the MI355X engine.

Drop-in for the reference front-end xcltk/rdr/fc/main.py: same command line (fc_main, :61-139),
same Python wrapper signature (fc_wrapper, :142-182), same outputs (features.tsv, barcodes.tsv,
matrix.mtx, :378-383) and return codes (0 / -1).  The per-region pysam loops of
fc_features()/fc_fet1() (rdr/fc/core.py:69-178) are replaced by one streaming pass through
the HIP engine (csrc/engine.hip) behind the C-ABI of include/xck.h.
"""

import getopt
import os
import sys
import time
from logging import error, info

from ... import fc_common as fcc
from ...capi import XCK_MODE_BASEFC
from ...engine import XckError
from ...config import APP, VERSION
from ...utils.xlog import init_logging
from .config import Config

COMMAND = "basefc"


def usage(fp=sys.stdout, conf=None):
    conf = conf or Config().defaults
    s = "\n"
    s += "Version: %s\n" % VERSION
    s += "Usage:   %s %s <options>\n" % (APP, COMMAND)
    s += "\n"
    s += "Options:\n"
    s += "  -s, --sam FILE         Comma separated indexed BAM file(s) (SAM / CRAM: convert first).\n"
    s += "  -S, --samList FILE     A list file containing bam files, each per line.\n"
    s += "  -b, --barcode FILE     A plain file listing all effective cell barcode.\n"
    s += "  -R, --region FILE      A TSV file listing target regions. The first 4 columns shoud be:\n"
    s += "                         chrom, start, end (both 1-based and inclusive), name.\n"
    s += "  -i, --sampleList FILE  A list file containing sample IDs, each per line.\n"
    s += "  -I, --sampleIDs STR    Comma separated sample IDs.\n"
    s += "  -O, --outdir DIR       Output directory for sparse matrices.\n"
    s += "  -h, --help             Print this message and exit.\n"
    s += "\n"
    s += "Optional arguments:\n"
    s += "  -p, --ncores INT       Number of processes [%d]\n" % conf.NPROC
    s += "      --cellTAG STR      Tag for cell barcodes, set to None when using sample IDs [%s]\n" % conf.CELL_TAG
    s += "      --UMItag STR       Tag for UMI, set to None when reads only [%s]\n" % conf.UMI_TAG
    s += "  -D, --debug INT        Used by developer for debugging [%d]\n" % conf.DEBUG
    s += "\n"
    s += "Read filtering:\n"
    s += "  --inclFLAG INT          Required flags: skip reads with all mask bits unset [%d]\n" % conf.INCL_FLAG
    s += "  --exclFLAG INT          Filter flags: skip reads with any mask bits set [%d\n" % conf.EXCL_FLAG_UMI
    s += "                          (when use UMI) or %d (otherwise)]\n" % conf.EXCL_FLAG_XUMI
    s += "  --minLEN INT            Minimum mapped length for read filtering [%d]\n" % conf.MIN_LEN
    s += "  --minMAPQ INT           Minimum MAPQ for read filtering [%d]\n" % conf.MIN_MAPQ
    s += "  --minINCLUDE FLOAT|INT  Minimum fraction or length of included part within specific feature [%f]\n" % conf.MIN_INCLUDE
    s += "  --countORPHAN           If use, do not skip anomalous read pairs.\n"
    s += "\n"
    fp.write(s)


_VALUE_OPTS = {   # option -> (attribute, converter)
    "-s": ("sam_fn", str), "--sam": ("sam_fn", str),
    "-S": ("sam_list_fn", str), "--samlist": ("sam_list_fn", str),
    "-b": ("barcode_fn", str), "--barcode": ("barcode_fn", str),
    "-R": ("region_fn", str), "--region": ("region_fn", str),
    "-i": ("sample_id_fn", str), "--samplelist": ("sample_id_fn", str),
    "-I": ("sample_id_str", str), "--sampleids": ("sample_id_str", str),
    "-O": ("out_dir", str), "--outdir": ("out_dir", str),
    "-p": ("nproc", int), "--ncores": ("nproc", int),
    "--celltag": ("cell_tag", str), "--umitag": ("umi_tag", str),
    "-D": ("debug", int), "--debug": ("debug", int),
    "--inclflag": ("incl_flag", int), "--exclflag": ("excl_flag", int),
    "--minlen": ("min_len", int), "--minmapq": ("min_mapq", float),
    "--mininclude": ("min_include", lambda v: float(v) if "." in v else int(v)),
}


def fc_main(argv, conf=None):
    """Command-line entry: argv = ["xcltk", "basefc", ...]. Returns 0 / -1."""
    if conf is None:
        conf = Config()
    if len(argv) <= 2:
        usage(sys.stdout, conf.defaults)
        sys.exit(0)
    conf.argv = list(argv)
    init_logging(stream=sys.stderr)
    opts, _ = getopt.getopt(argv[2:], "-s:-S:-b:-R:-i:-I:-O:-h-p:-D:", [
        "sam=", "samList=", "barcode=", "region=", "sampleList=", "sampleIDs=", "outdir=", "help",
        "ncores=", "cellTAG=", "UMItag=", "debug=",
        "inclFLAG=", "exclFLAG=", "minLEN=", "minMAPQ=", "minINCLUDE=", "countORPHAN"])
    for op, val in opts:
        if len(op) > 2:
            op = op.lower()                     # long options are case-insensitive
        if op in _VALUE_OPTS:
            attr, conv = _VALUE_OPTS[op]
            setattr(conf, attr, conv(val))
        elif op in ("-h", "--help"):
            usage(sys.stdout, conf.defaults)
            sys.exit(0)
        elif op == "--countorphan":
            conf.no_orphan = False
        else:
            error("invalid option: '%s'." % op)
            return -1
    return fc_run(conf)


def fc_wrapper(sam_fn, barcode_fn, region_fn, out_dir, sam_list_fn=None, sample_ids=None,
               sample_id_fn=None, debug_level=0, ncores=1, cell_tag="CB", umi_tag="UB",
               output_all_reg=True, min_mapq=20, min_len=30, min_include=0.9, incl_flag=0,
               excl_flag=None, no_orphan=True):
    """Python API, same signature as the reference (rdr/fc/main.py:142-182).

    Note the reference never copies a non-None `excl_flag` into its config (:177-178), so the
    tag-dependent default always applies there; that quirk is kept for output parity."""
    conf = Config()
    conf.sam_fn, conf.sam_list_fn = sam_fn, sam_list_fn
    conf.barcode_fn, conf.region_fn = barcode_fn, region_fn
    conf.sample_id_str, conf.sample_id_fn = sample_ids, sample_id_fn
    conf.out_dir, conf.debug = out_dir, debug_level
    conf.cell_tag, conf.umi_tag = cell_tag, umi_tag
    conf.nproc, conf.output_all_reg = ncores, output_all_reg
    conf.min_mapq, conf.min_len, conf.min_include = min_mapq, min_len, min_include
    conf.incl_flag, conf.no_orphan = incl_flag, no_orphan
    if excl_flag is None:
        conf.excl_flag = -1
    return fc_run(conf)


def prepare_config(conf):
    """Resolve inputs and outputs; 0 on success, -1 otherwise (rdr/fc/main.py:305-431)."""
    if fcc.resolve_inputs(conf) < 0:
        return -1
    if not conf.out_dir:
        error("out dir needed!")
        return -1
    os.makedirs(conf.out_dir, exist_ok=True)          # every rank may get here first in a multi-GPU run
    conf.out_region_fn = os.path.join(conf.out_dir, conf.out_prefix + "features.tsv")
    conf.out_sample_fn = os.path.join(conf.out_dir, conf.out_prefix + "barcodes.tsv")
    conf.out_mtx_fn = os.path.join(conf.out_dir, conf.out_prefix + "matrix.mtx")
    if not conf.region_fn:
        error("region file needed!")
        return -1
    if not os.path.isfile(conf.region_fn):
        error("region file '%s' does not exist." % conf.region_fn)
        return -1
    conf.reg_list = fcc.load_region_from_txt(conf.region_fn, verbose=True)
    if not conf.reg_list:
        error("failed to load region file.")
        return -1
    info("count %d regions in %d single cells." % (len(conf.reg_list), len(conf.samples)))
    if fcc.resolve_tags(conf) < 0:
        return -1
    if fcc.is_writer_rank():
        fcc.write_samples(conf.out_sample_fn, conf.samples)
    return 0


def fc_core(conf):
    if prepare_config(conf) < 0:
        raise ValueError("errcode -2")
    info("program configuration:")
    conf.show(fp=sys.stderr, prefix="\t")
    regions = conf.reg_list
    eng, coo, dist = fcc.make_and_count(conf, XCK_MODE_BASEFC, regions)
    try:
        if coo is not None:                               # the only process / rank 0 after a gather / every rank (sharded output)
            rm = fcc.output_row_map(dist, len(regions), conf.output_all_reg, coo["count"][0])   # all_reg: row = input line number
            if fcc.is_writer_rank():
                fcc.write_region_tsv(conf.out_region_fn, regions, rm)
            fcc.write_mtx(eng, dist, conf.out_mtx_fn, coo["count"], rm, int(rm.max()) if len(rm) else 0)
        if conf.debug > 0:
            info("engine stats: %s" % eng.stats())
    finally:
        eng.close()


def fc_run(conf):
    ret = -1
    cmdline = None
    start_time = time.time()
    info("start time: %s." % time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(start_time)))
    if conf.argv is not None:
        cmdline = " ".join(conf.argv)
        info("CMD: %s" % cmdline)
    try:
        fc_core(conf)
    except (ValueError, XckError) as e:
        error(str(e))
        error("Running program failed.")
        error("Quiting ...")
        ret = -1
    else:
        info("All Done!")
        ret = 0
    finally:
        if conf.argv is not None:
            info("CMD: %s" % cmdline)
        end_time = time.time()
        info("end time: %s" % time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(end_time)))
        info("time spent: %.2fs" % (end_time - start_time))
    return ret
