import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xcltk_amd.synth import soa, soa_torch
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
for rep in range(3):
    arrays, batches = soa_torch.gen_reads_device(regions, names, 50_000_000, 5000, seed=100, device=torch.device("cuda", 0))
    print({k: int(v.to(torch.int64).sum().item()) for k, v in arrays.items() if hasattr(v, "sum") and k != "seq"}, arrays["n_cig"], len(batches))
