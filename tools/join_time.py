#!/usr/bin/env python3
"""HBM-resident join / fold stage times of one engine build (XCK_LIB selects the .so): 500 M synthetic reads generated on the
device, N passes per mode.  usage: join_time.py [reads] [passes] [modes: fc,baf]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa, soa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
modes = (sys.argv[3] if len(sys.argv) > 3 else "fc,baf").split(",")
cells, n_snps = 10000, 1_000_000
regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
arrays, batches = soa_torch.gen_reads_device(regions, names, n, cells, seed=100, device=torch.device("cuda", 0))
for label in modes:
    mode = 1 if label == "fc" else 2
    eng = Engine(mode, names, regions, cells, snps=snps if mode == 2 else (), device=0, min_len=30, excl_flag=772, min_mapq=20)
    bs = [soa_torch.device_batch(capi, arrays, c, s, e, mode == 2) for c, s, e in batches]
    js, fs = [], []
    for rep in range(passes + 2):
        eng.reset()
        for b in bs:
            eng.push(b, device_resident=True)
        eng.flush()
        res = eng.finish(copy=False)
        st = eng.stats()
        if rep >= 2:
            js.append(st["ms_join"]); fs.append(st["ms_sort"])
    print("%s %-4s join %.3f ms  fold %.3f ms  (hits %d unique %d nnz %s)" % (os.environ.get("XCK_LIB", "libxck.so").split("/")[-1], label, sum(js) / len(js), sum(fs) / len(fs),
          st["n_hits"], st["n_hits_unique"], [len(v[0]) for v in res.values()]), flush=True)
    eng.close()
