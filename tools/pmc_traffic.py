#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc counter csvs of tools/pmc_join.py runs into profiles/pmc_traffic.json.
usage: pmc_traffic.py <dir with {fc,fc_filtered,baf}_{FETCH_SIZE,WRITE_SIZE}/**/*counter_collection.csv> <out.json> <note>"""
import csv, glob, json, os, sys
root, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
raw = {}
for label in ("fc", "fc_filtered", "baf"):
    raw[label] = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        last = None
        for fn in glob.glob(os.path.join(root, "%s_%s" % (label, ctr), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                if "k_join" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    d = int(r["Dispatch_Id"])
                    if last is None or d > last[0]:
                        last = (d, float(r["Counter_Value"]))
        raw[label][ctr] = last[1] if last else None
def hbm(label):                      # counters are in KB; FETCH_SIZE x2 = gfx950 correction (MI355X_MICROARCH.md, HBM section)
    f, w = raw[label]["FETCH_SIZE"], raw[label]["WRITE_SIZE"]
    return None if f is None or w is None else int((2 * f + w) * 1024)
import hashlib
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xcltk_amd", "csrc")
stamp = hashlib.sha256(b"".join(open(os.path.join(src, f), "rb").read() for f in ("engine.hip", "fold_partition.h"))).hexdigest()[:16]
res = {"k_join<basefc>": hbm("fc"), "k_join<pileup>": hbm("baf"), "k_join<basefc, every read filtered>": hbm("fc_filtered"),
       "_note": note, "_raw_kb": raw,
       # the kernels these bytes were counted on: bench.py reports `traffic` only while engine.hip + fold_partition.h still hash to this
       "_kernel_source_sha256_16": stamp}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
