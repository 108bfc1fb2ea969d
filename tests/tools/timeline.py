"""print the kernel timeline (start / end in us, queue) of one minibatch in a rocprofv3 kernel-trace CSV:
    timeline.py <dir> [k]   (k: index of the Adam launch that precedes it; default: the last complete minibatch)"""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n); n = n.replace("aleppo::", "")
    return n[:70]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 3  # which minibatch (index of the Adam launch BEFORE it)
a, b = adam[k], adam[k + 1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f  %6.1f us  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
