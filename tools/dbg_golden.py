#!/usr/bin/env python3
"""Run one golden basefc case through fc_wrapper and diff matrix.mtx against the expected file (debug aid)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import util
from xcltk_amd.rdr.fc.main import fc_wrapper
name = sys.argv[1] if len(sys.argv) > 1 else "c1_basefc_default"
tmp = tempfile.mkdtemp()
case, ddir, odir, exp = util.load_case(name, tmp)
ret = fc_wrapper(**case["kwargs"])
a = open(os.path.join(odir, "matrix.mtx")).read().split("\n")
b = open(os.path.join(exp, "matrix.mtx")).read().split("\n")
print("ret", ret, "lines", len(a), len(b))
bad = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
print("first differing lines:", bad[:10], "n_diff", len(bad))
