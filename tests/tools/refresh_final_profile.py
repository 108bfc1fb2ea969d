"""Regenerate profiles/r01_bench_kernel_stats_final.{csv,md} and the two bench-line logs from a traced bench run
(developer tool):  gpurun_out/prof_final/bench_results.db, gpurun_out/b_prof.log (traced run), gpurun_out/b_final.log."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import rocpd_stats  # noqa: E402

G = os.path.join(ROOT, "gpurun_out")
out = os.path.join(ROOT, "profiles", "r01_bench_kernel_stats_final")
rocpd_stats.main(os.path.join(G, "prof_final", "bench_results.db"), out)
rows = list(csv.DictReader(open(out + ".csv")))
r = json.loads([l for l in open(os.path.join(G, "b_prof.log")) if l.startswith("{")][-1])["roofline"]
md = ["# rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o bench -- python3 bench.py --no-cpu-baseline --no-v1 "
      "(MI355X, round 1 final build)", "",
      "Summary exported from the rocpd database with `tests/tools/rocpd_stats.py`. The update runs 16 times: 15 updates on "
      "the timed", "region's two-stream schedule (wgrad kernels + slab reduce co-scheduled on a second stream) and, last, "
      "bench.py's", "`ALEPPO_OPT_SERIAL_UPDATE` pass with every kernel alone on the main stream. The bench line of the "
      "same run", f"(`r01_bench_n1_rocprof.log`) reports conv1 wgrad {r['avg_launch_ms'] * 1e3:.1f} us in the timed-region "
      f"schedule and {r['isolated']['avg_launch_ms'] * 1e3:.1f} us isolated (HIP events).", "",
      "| kernel | calls | avg us | % of GPU time | avg us, two-stream updates | avg us, last (serial) update |",
      "|---|---|---|---|---|---|"]
for x in rows[:24]:
    md.append(f"| `{x['Name'][:70]}` | {x['Calls']} | {float(x['AverageUs']):.1f} | {x['Percentage']} | "
              f"{x['AvgUsTwoStreamUpdates']} | {x['AvgUsLastSerialUpdate']} |")
open(out + ".md", "w").write("\n".join(md) + "\n")
open(os.path.join(ROOT, "profiles", "r01_bench_n1_final.log"), "w").writelines(
    l for l in open(os.path.join(G, "b_final.log")) if l.startswith("{"))
open(os.path.join(ROOT, "profiles", "r01_bench_n1_rocprof.log"), "w").writelines(
    l for l in open(os.path.join(G, "b_prof.log")) if l.startswith("{") or l.startswith("[bench"))
j = json.loads(open(os.path.join(ROOT, "profiles", "r01_bench_n1_final.log")).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["isolated"]["frac"],
      (j.get("cpu_baseline") or {}).get("value"), (j.get("v1_shape") or {}).get("value"))
