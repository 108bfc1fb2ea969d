"""print the kernel timeline of a few consecutive rollout slots from a rocprofv3 kernel-trace CSV of tests/tools/roll_time.py:
    slot_timeline.py <dir> [first_head_index] [slots]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "infer_head_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) - 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a, b = heads[k], heads[k + n]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("aleppo::", "")[:60]
    print("%8.1f %8.1f  %6.1f us  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
