"""Summarise rocprofv3 --pmc output (counter_collection.csv files) per kernel: mean counter value per dispatch.

usage: python tests/tools/pmc_summary.py <dir-with-csv> [<dir> ...]  -> JSON on stdout
Kernel names are shortened to the template head + first template argument.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(?:aleppo::)?(\w+)(?:<(?:aleppo::)?([\w:]+))?", name)
    if not m:
        return name[:40]
    return m.group(1) + ("<" + m.group(2) + ">" if m.group(2) else "")


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
out = {k: {c: round(v[0] / v[1], 1) for c, v in cs.items()} | {"dispatches": max(v[1] for v in cs.values())}
       for k, cs in acc.items()}
print(json.dumps(out, indent=1))
