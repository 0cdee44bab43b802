"""Worker run under torch.distributed.run by test_gpu_multirank.py: every rank executes the same
front-end call; ranks own disjoint contigs, rank 0 gathers and writes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import util  # noqa: E402

name, tmp = sys.argv[1], sys.argv[2]
if name == "pipeline_steps_1_3":
    # `xcltk baf` with --snpvcf on every rank: step 1 (pileup) is written by rank 0 alone, step 3 reads it on EVERY rank -
    # the ranks must wait for the writer in between (they did not, and raced with a half-written directory)
    import numpy as np
    import oracle as O
    import torch.distributed as dist
    from xcltk_amd import capi
    from xcltk_amd.baf.pipeline import pipeline_wrapper
    from xcltk_amd.utils import csp_io
    DS = os.path.join(util.GOLDEN, "datasets", "phasing")
    want = csp_io.load_data(os.path.join(DS, "cellsnp"))
    covered = set(want.pos[np.asarray((want.DP + want.OTH).sum(axis=0)).reshape(-1) > 0].tolist())
    lines = open(os.path.join(DS, "snps.tsv")).read().splitlines()
    rank = int(os.environ.get("RANK", "0"))
    snp_fn = os.path.join(tmp, "phased.%d.tsv" % rank)
    open(snp_fn, "w").write("\n".join([lines[0]] + [l for l in lines[1:] if int(l.split("\t")[1]) in covered]) + "\n")
    out = os.path.join(tmp, "pipe")
    ret = pipeline_wrapper("smp", sam_fn=os.path.join(DS, "possorted.bam"), barcode_fn=os.path.join(DS, "barcodes.tsv"),
                           snp_vcf_fn=os.path.join(DS, "cellsnp", "cellSNP.base.vcf.gz"), region_fn=os.path.join(DS, "regions.tsv"),
                           out_dir=out, phased_snp_fn=snp_fn, ref_cell_fn=os.path.join(DS, "ref_cells.tsv"), min_count=1, min_maf=0, ncores=2)
    assert ret == 0
    dist.barrier()
    if rank == 0:
        exp = os.path.join(tmp, "oracle")
        O.run_files(capi.XCK_MODE_BAF, [os.path.join(DS, "possorted.bam")], os.path.join(DS, "regions.tsv"), out_dir=exp,
                    barcode_fn=os.path.join(DS, "barcodes.tsv"), snp_fn=snp_fn, output_all_reg=True, min_count=1, min_maf=0, no_dup_hap=True,
                    phase=util.phase_from_cellsnp(os.path.join(out, "1_pileup"), os.path.join(DS, "ref_cells.tsv"), True))
        util.assert_dirs_equal(os.path.join(out, "3_baf_fc"), exp)
        print("MULTIRANK_OK", name)
    dist.destroy_process_group()
    sys.exit(0)
case, ddir, odir, exp = util.load_case(name, tmp)
from xcltk_amd.baf.fc.main import afc_wrapper  # noqa: E402
from xcltk_amd.rdr.fc.main import fc_wrapper  # noqa: E402
ret = fc_wrapper(**case["kwargs"]) if case["kind"] == "basefc" else afc_wrapper(**case["kwargs"])
assert ret == 0
if int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("XCK_DIST_FORCE", "0") in ("", "0"):   # plain single-process run (test_gpu_golden.py's key-overflow case)
    util.assert_dirs_equal(odir, exp)
    print("MULTIRANK_OK", name)
    sys.exit(0)
import torch.distributed as dist  # noqa: E402
dist.barrier()
if dist.get_rank() == 0:
    util.assert_dirs_equal(odir, exp)
    print("MULTIRANK_BACKEND %s WORLD %d" % (dist.get_backend(), dist.get_world_size()))
    print("MULTIRANK_OK", name)
dist.destroy_process_group()
