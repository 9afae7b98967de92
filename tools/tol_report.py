"""Summarise a BCBF_TOL_REPORT file (tests/_tolreport.py): per (test function, what, dtype id) the worst measured
error / scale next to the asserted tolerance.  Usage: python tools/tol_report.py gpurun_out/tol.jsonl [min_rtol]"""
import collections
import json
import re
import sys

rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    t = r["test"].split("::")[-1]
    fn = t.split("[")[0]
    ids = t[len(fn):]
    f32 = "float32" in ids or "f32" in ids
    what = re.sub(r"[\[ ]?\d+\]?$", "", re.sub(r" N=\d+", "", r["what"]))
    k = (fn, what, r["rtol"])
    a = rows.setdefault(k, [0.0, 0, ids])
    a[0] = max(a[0], r["err_over_scale"])
    a[1] += 1
min_rtol = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
print("%-78s %-28s %9s %9s %6s" % ("test", "what", "rtol", "worst", "n"))
for (fn, what, rtol), (worst, cnt, ids) in rows.items():
    if rtol >= min_rtol:
        print("%-78s %-28s %9.1e %9.2e %6d" % (fn[:78], what[:28], rtol, worst, cnt))
