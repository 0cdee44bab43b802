#!/usr/bin/env python3
"""Capacity / scale check on one GPU: basefc over N reads (default 300 M, config C3-like: 10 k barcodes),
growth of the hit buffers, fused launches, and parity of the chr1 rows against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    import numpy as np, torch
    import oracle as O, util
    from xcltk_amd import capi
    from xcltk_amd.engine import Engine
    from xcltk_amd.synth import soa, soa_torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
    cells = 10000
    regions, snps, names = soa.make_tables(33472, 0, soa.HG38_LENGTHS, seed=3)
    t0 = time.time()
    arrays, batches = soa_torch.gen_reads_device(regions, names, n, cells, seed=300, device=torch.device("cuda", 0), with_seq=False)
    torch.cuda.synchronize()
    print("generated %d reads in %.1fs; torch mem %.1f GB" % (arrays["n_reads"], time.time() - t0, torch.cuda.memory_allocated() / 1e9), flush=True)
    eng = Engine(capi.XCK_MODE_BASEFC, names, regions, cells, device=0)
    for rep in range(2):
        eng.reset()
        t0 = time.time()
        for c, s, e in batches:
            eng.push(soa_torch.device_batch(capi, arrays, c, s, e, False), device_resident=True)
        got = eng.finish(copy=False)
        dt = time.time() - t0
        st = eng.stats()
        print("rep %d: %.1f ms (%.2f G reads/s) accepted=%d to_hbm=%d nnz=%d sum=%d join=%.2fms fold=%.2fms d2h=%.2fms" % (
            rep, dt * 1e3, arrays["n_reads"] / dt / 1e9, st["n_hits"], st["n_hits_unique"], len(got["count"][0]), int(got["count"][2].sum()),
            st["ms_join"], st["ms_sort"], st["ms_d2h"]), flush=True)
    c, s, e = batches[0]
    hb = [util.batch_from_dict(soa_torch.host_batch_dict(arrays, c, s, e, False))]
    cfg, keep = O.make_config(capi.XCK_MODE_BASEFC, names, regions, [], cells)
    t0 = time.time()
    exp = O.run_oracle(cfg, [b for b, _ in hb])
    sel = np.array([r[0] == names[c] for r in regions])[got["count"][0]]
    ok = all(np.array_equal(got["count"][j][sel], exp["count"][j]) for j in range(3))
    print("oracle on contig %s (%d reads) %.1fs: %s (%d non-zeros)" % (names[c], e - s, time.time() - t0, "PARITY OK" if ok else "MISMATCH", int(sel.sum())))


if __name__ == "__main__":
    main()
