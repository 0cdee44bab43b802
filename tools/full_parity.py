#!/usr/bin/env python3
"""Engine vs oracle on a full bench-shaped workload (default 50 M reads): bit-exact comparison of EVERY non-zero.
usage: full_parity.py [reads] [modes 1,2] [cells] [snps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle as O, util
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa, soa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
cells = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
n_snps = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
dev = torch.device("cuda", 0)
regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
arrays, batches = soa_torch.gen_reads_device(regions, names, n, cells, seed=100, device=dev)
print("data checksum", int(arrays["pos"].to(torch.int64).sum().item()), int((arrays["umi"] & 0xFFFFFF).sum().item()), int(arrays["cell"].to(torch.int64).sum().item()), flush=True)
filt = dict(min_mapq=20, min_len=30, incl_flag=0, excl_flag=772, no_orphan=True)
hb = None
for mode in modes:
    eng = Engine(mode, names, regions, cells, snps=snps if mode == 2 else (), device=0, **filt)
    outs = []
    for rep in range(3):
        eng.reset()
        for c, s, e in batches:
            eng.push(soa_torch.device_batch(capi, arrays, c, s, e, mode == 2), device_resident=True)
        got = eng.finish()
        st = eng.stats()
        outs.append((got, st["n_hits"], st["n_hits_unique"]))
        print("mode", mode, "rep", rep, "accepted", st["n_hits"], "to_hbm", st["n_hits_unique"], {k: len(v[0]) for k, v in got.items()}, flush=True)
    eng.close()
    if hb is None:
        t0 = time.time()
        hb = [util.batch_from_dict(soa_torch.host_batch_dict(arrays, c, s, e, True)) for c, s, e in batches]
        print("host copy %.1fs" % (time.time() - t0), flush=True)
    cfg, keep = O.make_config(mode, names, regions, snps if mode == 2 else [], cells, **filt)
    t0 = time.time()
    exp = O.run_oracle(cfg, [b for b, _ in hb])
    print("oracle %.1fs" % (time.time() - t0), {k: len(v[0]) for k, v in exp.items() if len(v[0])}, flush=True)
    for rep, (got, a, u) in enumerate(outs):
        try:
            util.assert_coo_equal(got, exp, ["count"] if mode == 1 else ["ad", "dp", "oth"])
            print("mode", mode, "rep", rep, "PARITY OK")
        except AssertionError as e:
            print("mode", mode, "rep", rep, "PARITY FAIL", e)
