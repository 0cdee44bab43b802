import sys, torch
sys.path.insert(0, '/root/repo')
from xcltk_amd.synth import soa, soa_torch
dev = torch.device("cuda", 0)
regions, snps, names = soa.make_tables(33472, 100000, soa.HG38_LENGTHS, seed=2)
sig = []
for rep in range(3):
    arrays, batches = soa_torch.gen_reads_device(regions, names, 100_000_000, 5000, seed=100, device=dev)
    s = {k: int(v.to(torch.int64).sum().item()) for k, v in arrays.items() if torch.is_tensor(v)}
    w = torch.arange(arrays["pos"].numel(), device=dev, dtype=torch.int64) % 1000003
    s["pos_w"] = int((arrays["pos"].to(torch.int64) * w).sum().item()); s["umi_w"] = int(((arrays["umi"] & 0xFFFFFF) * w).sum().item())
    sig.append(s); print(rep, s, flush=True)
    del arrays
print("identical multiset sums:", all(all(sig[0][k] == x[k] for k in sig[0] if not k.endswith("_w")) for x in sig))
print("identical order:", all(sig[0] == x for x in sig))
