import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import util
from xcltk_amd import capi
from xcltk_amd.synth import soa
regions, snps, names = soa.make_tables(200, 5000, [2000000], seed=1, max_len=100000)
bs = soa.gen_reads(regions, names, 20000, 100, seed=2)
batches = [util.batch_from_dict(b) for b in bs]
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
got, exp, st = util.engine_vs_oracle(mode, names, regions, snps, 100, batches)
print(st)
for k in (["count"] if mode == 1 else ["ad", "dp", "oth"]):
    g = {(r, c): v for r, c, v in zip(*[x.tolist() for x in got[k]])}
    e = {(r, c): v for r, c, v in zip(*[x.tolist() for x in exp[k]])}
    print(k, "got", len(got[k][0]), "uniq", len(g), "exp", len(e), "sum", sum(g.values()), sum(e.values()))
    extra = sorted(set(g) - set(e))[:10]; miss = sorted(set(e) - set(g))[:10]
    print(" extra", extra, "\n miss", miss)
    diff = [(x, g[x], e[x]) for x in sorted(set(g) & set(e)) if g[x] != e[x]][:10]
    print(" diff", diff)
    rows = got[k][0]; cols = got[k][1]
    key = rows.astype(np.int64) * 100000 + cols
    print(" sorted:", bool(np.all(np.diff(key) > 0)), "first", list(zip(rows[:8].tolist(), cols[:8].tolist(), got[k][2][:8].tolist())))
    print(" exp first", list(zip(exp[k][0][:8].tolist(), exp[k][1][:8].tolist(), exp[k][2][:8].tolist())))
