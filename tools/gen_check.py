"""Is the HBM-resident generator reproducible?  Three draws in this process: checksums (sum, and a position-weighted sum that sees
the order) of every array must be equal; run it twice to compare processes.  usage: gen_check.py [reads] [cells] [json to write]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xcltk_amd.synth import soa, soa_torch


def checksums(arrays):
    out = {}
    for k, v in arrays.items():
        if not hasattr(v, "sum"):
            continue
        x = v.to(torch.int64)
        out[k] = int(x.sum().item())
        if k in ("pos", "umi", "cell", "cigar"):
            out[k + "_w"] = int((x * (torch.arange(x.numel(), device=x.device, dtype=torch.int64) % 1000003 + 1)).sum().item())
    out["n_cig"] = arrays["n_cig"]
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    cells = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
    got = []
    for rep in range(3):
        arrays, batches = soa_torch.gen_reads_device(regions, names, n, cells, seed=100, device=torch.device("cuda", 0))
        got.append(checksums(arrays))
        print(rep, got[-1], len(batches), flush=True)
        del arrays
    print("identical draws:", got[0] == got[1] == got[2])
    if len(sys.argv) > 3:                                      # write the checksums (tests/golden/gen_check.json: what every box must draw)
        import json
        json.dump(dict(reads=n, cells=cells, seed=100, genes=33472, checksums=got[0]), open(sys.argv[3], "w"), indent=1, sort_keys=True)
