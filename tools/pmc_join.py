#!/usr/bin/env python3
"""Run k_join on the bench workload twice per configuration so that rocprofv3 --pmc can attribute HBM
counters: [basefc normal, basefc all-filtered (known byte count -> calibration), pileup normal]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa, soa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
arrays, batches = soa_torch.gen_reads_device(regions, names, n, 5000, seed=100, device=torch.device("cuda", 0))
print("n_reads", arrays["n_reads"], "n_cig", arrays["n_cig"])
for label, mode, kw in (("fc_normal", 1, dict(min_mapq=20)), ("fc_filtered", 1, dict(min_mapq=256)), ("baf_normal", 2, dict(min_mapq=20))):
    eng = Engine(mode, names, regions, 5000, snps=snps if mode == 2 else (), device=0, min_len=30, excl_flag=772, **kw)
    for rep in range(2):
        eng.reset()
        for c, s, e in batches:
            eng.push(soa_torch.device_batch(capi, arrays, c, s, e, mode == 2), device_resident=True)
        eng.flush()
    st = eng.stats()
    print(label, "accepted", st["n_hits"], "ms_join", st["ms_join"], flush=True)
    eng.close()
