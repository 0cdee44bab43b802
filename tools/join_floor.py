#!/usr/bin/env python3
"""How fast is k_join when every read is filtered out (pure streaming + prologue + empty flush)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa, soa_torch
n = 50_000_000
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
arrays, batches = soa_torch.gen_reads_device(regions, names, n, 5000, seed=100, device=torch.device("cuda", 0))
for label, kw in (("normal", dict(min_mapq=20)), ("all filtered (mapq)", dict(min_mapq=256)), ("no include test (min_include=0)", dict(min_mapq=20, min_include=0)),
                  ("len mode (min_include=30)", dict(min_mapq=20, min_include=30))):
    for mode in (1, 2):
        eng = Engine(mode, names, regions, 5000, snps=snps if mode == 2 else (), device=0, min_len=30, excl_flag=772, **kw)
        ts = []
        for rep in range(4):
            eng.reset()
            for c, s, e in batches:
                eng.push(soa_torch.device_batch(capi, arrays, c, s, e, mode == 2), device_resident=True)
            eng.flush()
            ts.append(eng.stats()["ms_join"])
        print("%-34s mode %d  ms_join %s" % (label, mode, ["%.3f" % t for t in ts]), flush=True)
        eng.close()
