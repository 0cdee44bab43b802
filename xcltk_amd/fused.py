"""fused.py - `basefc` AND the BAF feature counting from ONE decode of the BAM(s).

Not a reference command: the reference runs `xcltk basefc` and `xcltk baf` as two programs
that each fetch the BAM per region / per SNP.  The hot path here is one streaming pass, so
both matrices can be fed from the same decoded record batches (SURVEY.md section 8f, row f2;
C-ABI mode XCK_MODE_BOTH).  Outputs are the two reference output sets, byte-identical to what
the separate commands write: <out_dir>/basefc/{features.tsv,barcodes.tsv,matrix.mtx} and
<out_dir>/baf/xcltk.{region.tsv,samples.tsv,AD.mtx,DP.mtx,OTH.mtx}.
"""
import os
import time
from logging import error, info

from . import fc_common as fcc
from .baf.fc.config import Config as BafConfig
from .baf.fc.main import prepare_config as baf_prepare
from .capi import XCK_MODE_BOTH
from .engine import XckError


def fused_wrapper(sam_fn, barcode_fn, region_fn, phased_snp_fn, out_dir, sam_list_fn=None,
                  sample_ids=None, sample_id_fn=None, ncores=1, cell_tag="CB", umi_tag="UB",
                  min_mapq=20, min_len=30, min_include=0.9, incl_flag=0, excl_flag=None, no_orphan=True,
                  min_count=1, min_maf=0, no_dup_hap=True, rdr_output_all_reg=True,
                  baf_output_all_reg=True, debug_level=0):
    """Returns 0 / -1 like fc_wrapper / afc_wrapper."""
    t0 = time.time()
    if not out_dir:
        error("out dir needed!")
        return -1
    os.makedirs(out_dir, exist_ok=True)
    conf = BafConfig()
    conf.sam_fn, conf.sam_list_fn = sam_fn, sam_list_fn
    conf.barcode_fn, conf.region_fn, conf.snp_fn = barcode_fn, region_fn, phased_snp_fn
    conf.sample_id_str, conf.sample_id_fn = sample_ids, sample_id_fn
    conf.out_dir = os.path.join(out_dir, "baf")
    conf.debug, conf.nproc = debug_level, ncores
    conf.cell_tag, conf.umi_tag = cell_tag, umi_tag
    conf.min_count, conf.min_maf, conf.no_dup_hap = min_count, min_maf, no_dup_hap
    conf.min_mapq, conf.min_len, conf.incl_flag, conf.no_orphan = min_mapq, min_len, incl_flag, no_orphan
    conf.excl_flag = -1 if excl_flag is None else excl_flag
    if baf_prepare(conf) < 0:
        error("errcode -2")
        return -1
    regions, snps = conf.reg_list, conf.snp_list
    fc_dir = os.path.join(out_dir, "basefc")
    os.makedirs(fc_dir, exist_ok=True)
    if fcc.is_writer_rank():
        fcc.write_samples(os.path.join(fc_dir, "barcodes.tsv"), conf.samples)
    conf.min_include = min_include
    try:
        eng, coo, dist = fcc.make_and_count(conf, XCK_MODE_BOTH, regions, snps, log_prefix="[fused]", min_include=min_include,
                                            min_count=min_count, min_maf=min_maf, no_dup_hap=no_dup_hap)
    except XckError as e:                                # truncated / non-BAM input, no GPU, ...: logged, -1 like the reference's errors
        error(str(e))
        return -1
    try:
        if coo is None:
            return 0
        n = len(regions)                                  # (sharded output: every rank is here and the calls below are collective)
        rm = fcc.output_row_map(dist, n, rdr_output_all_reg, coo["count"][0])
        if fcc.is_writer_rank():
            fcc.write_region_tsv(os.path.join(fc_dir, "features.tsv"), regions, rm)
        fcc.write_mtx(eng, dist, os.path.join(fc_dir, "matrix.mtx"), coo["count"], rm, int(rm.max()) if n else 0)
        rm = fcc.output_row_map(dist, n, baf_output_all_reg, coo["dp"][0], coo["oth"][0])
        nr = int(rm.max()) if n else 0
        if fcc.is_writer_rank():
            fcc.write_region_tsv(conf.out_region_fn, regions, rm)
        fcc.write_mtx(eng, dist, conf.out_ad_fn, coo["ad"], rm, nr)
        fcc.write_mtx(eng, dist, conf.out_dp_fn, coo["dp"], rm, nr)
        fcc.write_mtx(eng, dist, conf.out_oth_fn, coo["oth"], rm, nr)
    finally:
        eng.close()
    info("fused basefc + baf done in %.2fs" % (time.time() - t0))
    return 0
