"""Configuration of the allele-specific feature counting (step 3 of `xcltk baf`) - fields and
defaults of the reference (xcltk/baf/fc/config.py:8-164)."""
import sys

from ...config import APP


class DefaultConfig(object):
    def __init__(self):
        self.DEBUG = 0
        self.CELL_TAG = "CB"
        self.UMI_TAG = "UB"
        self.UMI_TAG_BC = "UB"
        self.NPROC = 1
        self.MIN_COUNT = 1
        self.MIN_MAF = 0
        self.OUTPUT_ALL_REG = False
        self.NO_DUP_HAP = True
        self.MIN_MAPQ = 20
        self.MIN_LEN = 30
        self.INCL_FLAG = 0
        self.EXCL_FLAG_UMI = 772
        self.EXCL_FLAG_XUMI = 1796
        self.NO_ORPHAN = True


class Config(object):
    def __init__(self):
        d = self.defaults = DefaultConfig()
        self.argv = None
        self.sam_fn = self.sam_list_fn = self.barcode_fn = None
        self.sample_id_str = self.sample_id_fn = None
        self.region_fn = self.snp_fn = self.out_dir = None
        self.debug = d.DEBUG
        self.cellsnp_dir = self.ref_cell_fn = None
        self.cell_tag, self.umi_tag = d.CELL_TAG, d.UMI_TAG
        self.nproc = d.NPROC
        self.min_count, self.min_maf = d.MIN_COUNT, d.MIN_MAF
        self.output_all_reg, self.no_dup_hap = d.OUTPUT_ALL_REG, d.NO_DUP_HAP
        self.min_mapq, self.min_len = d.MIN_MAPQ, d.MIN_LEN
        self.incl_flag, self.excl_flag, self.no_orphan = d.INCL_FLAG, -1, d.NO_ORPHAN
        self.barcodes = self.sample_ids = self.reg_list = self.snp_list = None
        self.sam_fn_list = self.samples = None
        self.out_prefix = APP + "."
        self.out_region_fn = self.out_sample_fn = None
        self.out_ad_fn = self.out_dp_fn = self.out_oth_fn = None

    def show(self, fp=None, prefix=""):
        fp = fp or sys.stderr

        def n_of(x):
            return len(x) if x is not None else -1
        lines = ["",
                 "sam_file = %s" % self.sam_fn, "sam_list_file = %s" % self.sam_list_fn,
                 "barcode_file = %s" % self.barcode_fn, "sample_id_str = %s" % self.sample_id_str,
                 "sample_id_file = %s" % self.sample_id_fn, "region_file = %s" % self.region_fn,
                 "snp_file = %s" % self.snp_fn, "out_dir = %s" % self.out_dir, "debug_level = %d" % self.debug, "",
                 "cellsnp_dir = %s" % self.cellsnp_dir, "ref_cell_fn = %s" % self.ref_cell_fn,
                 "cell_tag = %s" % self.cell_tag, "umi_tag = %s" % self.umi_tag,
                 "number_of_processes = %d" % self.nproc, "min_count = %d" % self.min_count,
                 "min_maf = %f" % self.min_maf, "output_all_reg = %s" % self.output_all_reg,
                 "no_dup_hap = %s" % self.no_dup_hap, "",
                 "min_mapq = %d" % self.min_mapq, "min_len = %d" % self.min_len,
                 "include_flag = %d" % self.incl_flag, "exclude_flag = %d" % self.excl_flag,
                 "no_orphan = %s" % self.no_orphan, "",
                 "#BAMs = %d" % n_of(self.sam_fn_list), "#barcodes = %d" % n_of(self.barcodes),
                 "#sample IDs = %d" % n_of(self.sample_ids), "#regions = %d" % n_of(self.reg_list),
                 "#snps = %d" % n_of(self.snp_list), "",
                 "output_region_file = %s" % self.out_region_fn, "output_sample_file = %s" % self.out_sample_fn,
                 "output_ad_file = %s" % self.out_ad_fn, "output_dp_file = %s" % self.out_dp_fn,
                 "output_oth_file = %s" % self.out_oth_fn, ""]
        fp.write("".join("%s%s\n" % (prefix, ln) for ln in lines))

    def use_barcodes(self):
        return self.cell_tag is not None

    def use_local_phasing(self):
        return self.cellsnp_dir is not None

    def use_umi(self):
        return self.umi_tag is not None
