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
case, ddir, odir, exp = util.load_case(name, tmp)
from xcltk_amd.baf.fc.main import afc_wrapper  # noqa: E402
from xcltk_amd.rdr.fc.main import fc_wrapper  # noqa: E402
ret = fc_wrapper(**case["kwargs"]) if case["kind"] == "basefc" else afc_wrapper(**case["kwargs"])
assert ret == 0
if int(os.environ.get("WORLD_SIZE", "1")) == 1:           # plain single-process run (test_gpu_golden.py's key-overflow case)
    util.assert_dirs_equal(odir, exp)
    print("MULTIRANK_OK", name)
    sys.exit(0)
import torch.distributed as dist  # noqa: E402
dist.barrier()
if dist.get_rank() == 0:
    util.assert_dirs_equal(odir, exp)
    print("MULTIRANK_OK", name)
dist.destroy_process_group()
