"""Configuration of `basefc` - same fields and defaults as the reference
(xcltk/rdr/fc/config.py:5-111) so user scripts that poke at Config keep working."""
import sys


class DefaultConfig(object):
    def __init__(self):
        self.DEBUG = 0
        self.CELL_TAG = "CB"
        self.UMI_TAG = "UB"
        self.UMI_TAG_BC = "UB"
        self.NPROC = 1
        self.OUTPUT_ALL_REG = True
        self.MIN_MAPQ = 20
        self.MIN_LEN = 30
        self.MIN_INCLUDE = 0.9
        self.INCL_FLAG = 0
        self.EXCL_FLAG_UMI = 772
        self.EXCL_FLAG_XUMI = 1796
        self.NO_ORPHAN = True


class Config(object):
    _SHOW = (
        ("sam_file", "sam_fn", "%s"), ("sam_list_file", "sam_list_fn", "%s"),
        ("barcode_file", "barcode_fn", "%s"), ("sample_id_str", "sample_id_str", "%s"),
        ("sample_id_file", "sample_id_fn", "%s"), ("region_file", "region_fn", "%s"),
        ("out_dir", "out_dir", "%s"), ("debug_level", "debug", "%d"), None,
        ("cell_tag", "cell_tag", "%s"), ("umi_tag", "umi_tag", "%s"),
        ("number_of_processes", "nproc", "%d"), ("output_all_reg", "output_all_reg", "%s"), None,
        ("min_mapq", "min_mapq", "%d"), ("min_len", "min_len", "%d"),
        ("min_include", "min_include", "%f"), ("include_flag", "incl_flag", "%d"),
        ("exclude_flag", "excl_flag", "%d"), ("no_orphan", "no_orphan", "%s"), None,
    )

    def __init__(self):
        d = self.defaults = DefaultConfig()
        self.argv = None
        self.sam_fn = self.sam_list_fn = self.barcode_fn = None
        self.sample_id_str = self.sample_id_fn = self.region_fn = self.out_dir = None
        self.debug = d.DEBUG
        self.cell_tag, self.umi_tag = d.CELL_TAG, d.UMI_TAG
        self.nproc = d.NPROC
        self.output_all_reg = d.OUTPUT_ALL_REG
        self.min_mapq, self.min_len, self.min_include = d.MIN_MAPQ, d.MIN_LEN, d.MIN_INCLUDE
        self.incl_flag, self.excl_flag, self.no_orphan = d.INCL_FLAG, -1, d.NO_ORPHAN
        self.barcodes = self.sample_ids = self.reg_list = None
        self.sam_fn_list = self.samples = None
        self.out_prefix = ""
        self.out_region_fn = self.out_sample_fn = self.out_mtx_fn = None

    def show(self, fp=None, prefix=""):
        fp = fp or sys.stderr
        lines = [""]
        for item in self._SHOW:
            lines.append("" if item is None else ("%s = " + item[2]) % (item[0], getattr(self, item[1])))

        def n_of(x):
            return len(x) if x is not None else -1
        lines += ["number_of_BAMs = %d" % n_of(self.sam_fn_list), "number_of_barcodes = %d" % n_of(self.barcodes),
                  "number_of_sample_IDs = %d" % n_of(self.sample_ids), "number_of_regions = %d" % n_of(self.reg_list), "",
                  "output_region_file = %s" % self.out_region_fn, "output_sample_file = %s" % self.out_sample_fn,
                  "output_mtx_file = %s" % self.out_mtx_fn, ""]
        fp.write("".join("%s%s\n" % (prefix, ln) for ln in lines))

    def use_barcodes(self):
        return self.cell_tag is not None

    def use_umi(self):
        return self.umi_tag is not None
