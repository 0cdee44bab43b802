#!/usr/bin/env python3
"""PCIe-inclusive rate of the C-ABI boundary: host SoA batches (pinned) -> xck_push_batch (H2D copy + join) -> xck_finish
(results in host memory).  No BAM decoding, no file output.  usage: h2d_bench.py [reads] [cells] [snps] [reads_per_batch]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import util
from xcltk_amd import capi
from xcltk_amd.engine import Engine
from xcltk_amd.synth import soa, soa_torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
n_snps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
per = int(sys.argv[4]) if len(sys.argv) > 4 else 4_000_000
pinned = (sys.argv[5] if len(sys.argv) > 5 else "pinned") == "pinned"      # "pageable": plain numpy arrays, what a ctypes binding holds
dev = torch.device("cuda", 0)
regions, snps, names = soa.make_tables(33472, n_snps, soa.HG38_LENGTHS, seed=2)
arrays, batches = soa_torch.gen_reads_device(regions, names, n, cells, seed=100, device=dev)
host = []
for c, s, e in batches:                                   # decoder-sized pieces, in pinned memory like the decoder's own
    for a in range(s, e, per):
        d = soa_torch.host_batch_dict(arrays, c, a, min(a + per, e), True)
        for k, v in list(d.items()):
            if isinstance(v, np.ndarray) and not pinned:
                d[k] = np.ascontiguousarray(v).copy()
            elif isinstance(v, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).reshape(-1)).pin_memory()
                d[k] = t.numpy().view(v.dtype)
                d.setdefault("_keep", []).append(t)
        host.append(d)
del arrays
torch.cuda.empty_cache()
out = {"reads": n, "batches": len(host), "caller_arrays": "pinned" if pinned else "pageable", "XCK_PUSH_STAGE": os.environ.get("XCK_PUSH_STAGE", "auto")}
for mode, name in ((capi.XCK_MODE_BASEFC, "basefc"), (capi.XCK_MODE_BAF, "baf")):
    eng = Engine(mode, names, regions, cells, snps=snps if mode == 2 else (), device=0)
    hb = [util.batch_from_dict({k: v for k, v in d.items() if k != "_keep" and (mode == 2 or k not in ("seq", "seq_off"))}) for d in host]
    best = 1e9
    for rep in range(3):
        eng.reset()
        t0 = time.time()
        for b, _ in hb:
            eng.push(b)
        res = eng.finish()
        best = min(best, time.time() - t0)
    st = eng.stats()
    byt = sum(sum(v.nbytes for k, v in d.items() if isinstance(v, np.ndarray) and (mode == 2 or k not in ("seq", "seq_off"))) for d in host)
    out[name] = {"seconds": best, "reads_per_s": n / best, "h2d_bytes_per_read": byt / n, "h2d_GBps_if_alone": byt / best / 1e9}
    print("%s: %.3f s, %.2f G reads/s, %.1f B/read over PCIe (%.1f GB/s averaged over the whole pass)" % (name, best, n / best / 1e9, byt / n, byt / best / 1e9), flush=True)
    eng.close()
print(json.dumps(out))
