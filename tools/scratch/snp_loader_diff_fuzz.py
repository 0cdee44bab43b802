#!/usr/bin/env python3
"""scratch: differential fuzz of the native SNP text parser against the generic Python loaders (mutated TSV / VCF text)."""
import io, sys, os, numpy as np, tempfile, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
T = importlib.import_module("test_host_logic")
from xcltk_amd import fc_common as F
from xcltk_amd.snptable import SnpTable
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rnd = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 123)
d = tempfile.mkdtemp()
nat = fb = 0
for k in range(n):
    vcf = k & 1
    b = bytearray(("\n".join(T._VCF_LINES if vcf else T._TSV_LINES) + "\n").encode())
    for _ in range(int(rnd.integers(1, 6))):
        b[int(rnd.integers(0, len(b)))] = int(rnd.choice([9, 10, 58, 124, 47, 48, 49, 32, 65, 97, 71, 84, 46, 11, 12, 28, int(rnd.integers(32, 127))]))
    fn = os.path.join(d, "m.vcf" if vcf else "m.tsv")
    open(fn, "wb").write(bytes(b))
    loader = F.load_snp_from_vcf if vcf else F.load_snp_from_tsv
    res = []
    for force in (False, True):
        os.environ.pop("XCK_PY_LOADERS", None)
        if force:
            os.environ["XCK_PY_LOADERS"] = "1"
        err = io.StringIO()
        old, sys.stderr = sys.stderr, err
        try:
            res.append((loader(fn, verbose=True), None, err))
        except Exception as e:
            res.append((None, type(e).__name__, err))
        finally:
            sys.stderr = old
    (got, gerr, e1), (exp, eerr, e2) = res
    if gerr is None and e1.getvalue() != e2.getvalue():
        print("STDERR DIFF at", k, bytes(b)); print(e1.getvalue()); print(e2.getvalue()); sys.exit(1)
    if gerr != eerr or (got is not None and not (got == exp)):
        print("DIFF at", k, gerr, eerr, bytes(b)); print(list(got) if got is not None else None); print(exp)
        sys.exit(1)
    nat += isinstance(got, SnpTable); fb += not isinstance(got, SnpTable)
print("%d mutated lists: native %d, fallback %d, no differences" % (n, nat, fb))
