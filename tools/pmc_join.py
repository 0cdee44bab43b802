#!/usr/bin/env python3
"""Run k_join on the bench workload (BASELINE configs[2] by default) for ONE configuration, three passes, so that a
rocprofv3 --pmc run can attribute HBM counters to the LAST k_join dispatch (buffers are sized by then, no replay).
usage: pmc_join.py {fc|fc_filtered|baf|baf_filtered} [reads] [cells] [snps]
fc_filtered rejects every read (min_mapq above any MAPQ): its byte count is known, which calibrates FETCH_SIZE."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa, soa_torch
label = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000_000
cells = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
n_snps = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
arrays, batches = soa_torch.gen_reads_device(regions, names, n, cells, seed=100, device=torch.device("cuda", 0))
print("n_reads", arrays["n_reads"], "n_cig", arrays["n_cig"])
mode, kw = {"fc": (1, dict(min_mapq=20)), "fc_filtered": (1, dict(min_mapq=256)), "baf": (2, dict(min_mapq=20)), "baf_filtered": (2, dict(min_mapq=256))}[label]
eng = Engine(mode, names, regions, cells, snps=snps if mode == 2 else (), device=0, min_len=30, excl_flag=772, **kw)
for rep in range(3):
    eng.reset()
    for c, s, e in batches:
        eng.push(soa_torch.device_batch(capi, arrays, c, s, e, mode == 2), device_resident=True)
    eng.flush()
st = eng.stats()
print(label, "accepted", st["n_hits"], "unique", st["n_hits_unique"], "ms_join", st["ms_join"], "launches", st["n_join_launches"], flush=True)
eng.close()
