"""Kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...`
writes DIR/NAME_results.db on this image).  Prints / writes the per-kernel summary (calls, total, average, share) and,
for the update kernels, the average over the launches of the LAST update separately: bench.py's final profiling pass
runs that update with every kernel alone on the main stream (ALEPPO_OPT_SERIAL_UPDATE), all earlier updates use the
timed region's two-stream schedule.

  python tests/tools/rocpd_stats.py gpurun_out/prof_final/bench_results.db profiles/r01_bench_kernel_stats_final
"""
import csv
import re
import sqlite3
import subprocess
import sys


def short(name):
    if name.startswith("_Z"):
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    name = re.sub(r"\baleppo::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def main(db, out, per_update=16):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    by = {}
    for name, s, e in rows:
        by.setdefault(short(name), []).append((e - s) / 1e3)
    total = sum(sum(v) for v in by.values())
    table = []
    updates = max((len(v) for k, v in by.items() if "adam_kernel" in k), default=0) // per_update  # one Adam step per minibatch
    for k, v in by.items():
        iso = co = None
        if updates >= 2 and len(v) == updates * per_update:  # launched once per minibatch: an update kernel
            iso = sum(v[-per_update:]) / per_update
            co = sum(v[:-per_update]) / (len(v) - per_update)
        table.append((k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, co, iso))
    table.sort(key=lambda r: -r[2])
    with open(out + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage", "AvgUsTwoStreamUpdates", "AvgUsLastSerialUpdate"])
        for r in table:
            w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.2f}", f"{r[4]:.2f}",
                        "" if r[5] is None else f"{r[5]:.2f}", "" if r[6] is None else f"{r[6]:.2f}"])
    return table


if __name__ == "__main__":
    t = main(sys.argv[1], sys.argv[2])
    for r in t[:30]:
        print(f"{r[0][:60]:60s} {r[1]:6d} {r[3]:8.2f} {r[4]:6.2f}", "" if r[5] is None else f"co {r[5]:.1f} iso {r[6]:.1f}")
