"""Allele-specific feature counting (pileup at phased het SNPs) on the MI355X engine.

Drop-in for xcltk/baf/fc/main.py: `afc_wrapper` keeps the reference signature (:32-77) and
produces xcltk.region.tsv, xcltk.samples.tsv, xcltk.{AD,DP,OTH}.mtx (:373-379).  The per-SNP
pysam pileup of plp_snp()/fc_fet1() (baf/fc/core.py:143-247) runs as HIP kernels
(csrc/engine.hip) behind the C-ABI of include/xck.h.

Region-wise local phasing from a cellsnp-lite pileup (`cellsnp_dir` / `ref_cell_fn`,
baf/fc/main.py:107-153) runs on the host before the counting (baf/fc/phasing.py, baf/localphase.py):
it changes the haplotype index of SNP alleles and drops uncovered SNPs from region lists; the
engine then counts with that phase.
"""

import os
import sys
import time
from logging import error, info
from logging import warning as warn

import numpy as np

from ... import fc_common as fcc
from ...capi import XCK_MODE_BAF
from ...engine import XckError
from .config import Config


def afc_wrapper(sam_fn, barcode_fn, region_fn, phased_snp_fn, out_dir, sam_list_fn=None,
                sample_ids=None, sample_id_fn=None, debug_level=0, ncores=1, cellsnp_dir=None,
                ref_cell_fn=None, cell_tag="CB", umi_tag="UB", min_count=1, min_maf=0,
                output_all_reg=False, no_dup_hap=True, min_mapq=20, min_len=30, incl_flag=0,
                excl_flag=None, no_orphan=True):
    conf = Config()
    conf.sam_fn, conf.sam_list_fn = sam_fn, sam_list_fn
    conf.barcode_fn, conf.region_fn, conf.snp_fn = barcode_fn, region_fn, phased_snp_fn
    conf.sample_id_str, conf.sample_id_fn = sample_ids, sample_id_fn
    conf.out_dir, conf.debug = out_dir, debug_level
    conf.cellsnp_dir, conf.ref_cell_fn = cellsnp_dir, ref_cell_fn
    conf.cell_tag, conf.umi_tag = cell_tag, umi_tag
    conf.nproc = ncores
    conf.min_count, conf.min_maf = min_count, min_maf
    conf.output_all_reg, conf.no_dup_hap = output_all_reg, no_dup_hap
    conf.min_mapq, conf.min_len = min_mapq, min_len
    conf.incl_flag = incl_flag
    conf.excl_flag = -1 if excl_flag is None else excl_flag
    conf.no_orphan = no_orphan
    return afc_run(conf)


def prepare_config(conf):
    if fcc.resolve_inputs(conf) < 0:
        return -1
    if not conf.out_dir:
        error("out dir needed!")
        return -1
    os.makedirs(conf.out_dir, exist_ok=True)          # every rank may get here first in a multi-GPU run
    pre = os.path.join(conf.out_dir, conf.out_prefix)
    conf.out_region_fn, conf.out_sample_fn = pre + "region.tsv", pre + "samples.tsv"
    conf.out_ad_fn, conf.out_dp_fn, conf.out_oth_fn = pre + "AD.mtx", pre + "DP.mtx", pre + "OTH.mtx"
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
    if not conf.snp_fn:
        error("SNP file needed!")
        return -1
    if not os.path.isfile(conf.snp_fn):
        error("snp file '%s' does not exist." % conf.snp_fn)
        return -1
    loader = fcc.load_snp_from_vcf if fcc.is_vcf_name(conf.snp_fn) else fcc.load_snp_from_tsv
    conf.snp_list = loader(conf.snp_fn, verbose=True)
    if not conf.snp_list:
        error("failed to load snp file.")
        return -1
    info("%d SNPs loaded." % len(conf.snp_list))
    conf.snp_csp, conf.ref_cells = None, None
    if conf.cellsnp_dir is not None:
        # the per-SNP x cell pileup that local phasing works on (baf/fc/main.py:419-454)
        from ...utils.csp_io import load_data as csp_load_data
        from ...utils.grange import format_chrom
        if not os.path.isdir(conf.cellsnp_dir):
            error("cellsnp dir '%s' does not exist." % conf.cellsnp_dir)
            return -1
        try:
            csp = csp_load_data(conf.cellsnp_dir)
        except (IOError, OSError, ValueError) as e:
            error("failed to load cellsnp dir '%s': %s" % (conf.cellsnp_dir, e))
            return -1
        info("cellsnp SNP adata shape = %s." % str(csp.shape))
        if len(conf.samples) != csp.shape[0] or not set(conf.samples) <= set(csp.cells):
            error("cells of the cellsnp pileup do not match the barcodes / sample IDs.")
            return -1
        if csp.shape[1] != len(conf.snp_list):
            warn("n_snp: snp_adata=%d; snp_set=%d!" % (csp.shape[1], len(conf.snp_list)))
            if csp.shape[1] < len(conf.snp_list):
                error("the cellsnp pileup holds fewer SNPs than the phased SNP list.")
                return -1
        known = set((s[0], s[1]) for s in conf.snp_list)
        keep = [i for i in range(csp.shape[1]) if (format_chrom(str(csp.chrom[i])), int(csp.pos[i])) in known]
        if len(keep) < csp.shape[1]:                          # SNPs that were filtered in phasing
            csp = csp.subset_snps(keep)
            info("SNP adata shape after subset: %s." % str(csp.shape))
        conf.snp_csp = csp
    if conf.ref_cell_fn is not None:
        if not os.path.isfile(conf.ref_cell_fn):
            error("ref cell file '%s' does not exist." % conf.ref_cell_fn)
            return -1
        with open(conf.ref_cell_fn) as fp:
            conf.ref_cells = [x.rstrip("\n").split("\t")[0] for x in fp if x.strip()]
        if len(conf.ref_cells) > len(conf.samples) or not set(conf.ref_cells) <= set(conf.samples):
            error("reference cells are not a subset of the barcodes / sample IDs.")
            return -1
    if fcc.resolve_tags(conf) < 0:
        return -1
    if fcc.is_writer_rank():
        fcc.write_samples(conf.out_sample_fn, conf.samples)
    return 0


def regions_with_snps(regions, snps):
    """Boolean per region: at least one SNP with start <= pos <= end on the same (stripped)
    chromosome - the IntervalTree fetch of baf/fc/main.py:92-101."""
    import bisect
    from ...snptable import SnpTable
    if isinstance(snps, SnpTable):
        by_chrom = snps.positions_by_chrom()
    else:
        by_chrom = {}
        for s in snps:
            by_chrom.setdefault(s[0], []).append(s[1])
        for v in by_chrom.values():
            v.sort()
    out = []
    for ch, s, e, _ in regions:
        pos = by_chrom.get(ch)
        if pos is None or len(pos) == 0 or e < s:
            out.append(False)
            continue
        k = bisect.bisect_left(pos, s)
        out.append(bool(k < len(pos) and pos[k] <= e))
    return out


def afc_core(conf):
    if prepare_config(conf) < 0:
        raise ValueError("errcode -2")
    info("program configuration:")
    conf.show(fp=sys.stderr, prefix="\t")
    regions, snps = conf.reg_list, conf.snp_list
    has_snp = regions_with_snps(regions, snps)
    info("#regions: total=%d; with_snps=%d." % (len(regions), sum(has_snp)))
    excl = None
    if conf.use_local_phasing():
        # region-wise local phasing (baf/fc/main.py:107-153): new haplotype indices for flipped SNPs, and the SNPs without
        # coverage in the pileup leave the SNP list of the region that was phased
        from .phasing import local_phasing
        from ...snptable import SnpTable
        phase_regions = regions if conf.output_all_reg else [r for r, h in zip(regions, has_snp) if h]
        phase_index = list(range(len(regions))) if conf.output_all_reg else [i for i, h in enumerate(has_snp) if h]
        rh, ah, ex_r, ex_s, _ = local_phasing(phase_regions, snps, conf.snp_csp, conf.ref_cells, conf.debug)
        ex_r = np.array([phase_index[i] for i in ex_r.tolist()], dtype=np.int32)
        if isinstance(snps, SnpTable):
            snps = SnpTable(snps.names, snps.chrom_id, snps.pos, snps.ref, snps.alt, rh, ah)
        else:
            snps = [(s[0], s[1], s[2], s[3], int(r), int(a)) for s, r, a in zip(snps, rh.tolist(), ah.tolist())]
        excl = (ex_r, ex_s)
        conf.snp_csp = conf.ref_cells = None
    eng, coo, dist = fcc.make_and_count(conf, XCK_MODE_BAF, regions, snps, excl_pairs=excl)
    try:
        if coo is not None:                               # the only process / rank 0 after a gather / every rank (sharded output)
            # only regions that wrote a DP or OTH line keep a row (baf/fc/core.py:101-113)
            rm = fcc.output_row_map(dist, len(regions), conf.output_all_reg, coo["dp"][0], coo["oth"][0])
            n_rows = int(rm.max()) if len(rm) else 0
            if fcc.is_writer_rank():
                fcc.write_region_tsv(conf.out_region_fn, regions, rm)
            fcc.write_mtx(eng, dist, conf.out_ad_fn, coo["ad"], rm, n_rows)
            fcc.write_mtx(eng, dist, conf.out_dp_fn, coo["dp"], rm, n_rows)
            fcc.write_mtx(eng, dist, conf.out_oth_fn, coo["oth"], rm, n_rows)
        if conf.debug > 0:
            info("engine stats: %s" % eng.stats())
    finally:
        eng.close()


def afc_run(conf):
    ret = -1
    cmdline = None
    start_time = time.time()
    info("start time: %s." % time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(start_time)))
    if conf.argv is not None:
        cmdline = " ".join(conf.argv)
        info("CMD: %s" % cmdline)
    try:
        afc_core(conf)
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
