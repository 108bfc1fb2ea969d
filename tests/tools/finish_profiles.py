"""Turn gpurun_out/<tag>/ (written on the GPU box by tests/tools/collect_profiles.sh) into the committed evidence under
profiles/<tag>_* (developer tool):

    python tests/tools/finish_profiles.py r02

  <tag>_bench_n1.log              the un-profiled bench line (+ stderr progress)
  <tag>_bench_n1_rocprof.log      the bench line of the rocprofv3-traced run
  <tag>_bench_kernel_stats.{csv,md}  per-kernel summary of that trace (rocprofv3 --kernel-trace --stats)
  <tag>_pmc_traffic.json          HBM bytes per update kernel (FETCH_SIZE / WRITE_SIZE passes over kbench), stamped with
                                  the kernel-source hash bench.py checks before quoting it
  <tag>_pmc_sq_update_kernels.json   SQ wave-state / LDS / MFMA-busy counters per update kernel
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")


def lines(path, keep):
    return [l for l in open(path) if keep(l)]


open(os.path.join(P, f"{tag}_bench_n1.log"), "w").writelines(
    lines(os.path.join(G, "bench_final.err"), lambda l: l.startswith("[bench")) +
    lines(os.path.join(G, "bench_final.log"), lambda l: l.startswith("{")))
open(os.path.join(P, f"{tag}_bench_n1_rocprof.log"), "w").writelines(
    lines(os.path.join(G, "bench_prof.err"), lambda l: l.startswith("[bench")) +
    lines(os.path.join(G, "bench_prof.log"), lambda l: l.startswith("{")))
shutil.copy(os.path.join(G, "bench_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(G, "pmc_traffic.json"), os.path.join(P, f"{tag}_pmc_traffic.json"))
sq = json.load(open(os.path.join(G, "pmc_sq.json")))
sq = {k: v for k, v in sq.items() if not k.startswith(("__amd", "at::", "void at::"))}
json.dump(sq, open(os.path.join(P, f"{tag}_pmc_sq_update_kernels.json"), "w"), indent=1)

rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"))))
r = json.loads(lines(os.path.join(G, "bench_prof.log"), lambda l: l.startswith("{"))[-1])["roofline"]
md = [f"# rocprofv3 --kernel-trace --stats -d gpurun_out/{tag}/prof_bench -o bench -- python3 bench.py --steps 10 --warmup 3 "
      "--no-cpu-baseline --no-v1 --no-host-legs (MI355X)", "",
      "Summary exported from the rocpd database with `tests/tools/rocpd_stats.py`. The update runs 16 times: 15 updates on "
      "the timed", "region's two-stream schedule (weight-gradient kernels + slab reduce co-scheduled on a second stream) and, "
      "last, bench.py's", "`ALEPPO_OPT_SERIAL_UPDATE` pass with every kernel alone on the main stream. The bench line of the "
      f"same run (`{tag}_bench_n1_rocprof.log`)", f"reports {r['kernel']} {r['avg_launch_ms'] * 1e3:.1f} us in the "
      f"timed-region schedule and {r['isolated']['avg_launch_ms'] * 1e3:.1f} us isolated (HIP events in bench.py).", "",
      "| kernel | calls | avg us | % of GPU time | avg us, two-stream updates | avg us, last (serial) update |",
      "|---|---|---|---|---|---|"]
for x in rows[:26]:
    md.append(f"| `{x['Name'][:70]}` | {x['Calls']} | {float(x['AverageUs']):.1f} | {x['Percentage']} | "
              f"{x['AvgUsTwoStreamUpdates']} | {x['AvgUsLastSerialUpdate']} |")
open(os.path.join(P, f"{tag}_bench_kernel_stats.md"), "w").write("\n".join(md) + "\n")
j = json.loads(lines(os.path.join(G, "bench_final.log"), lambda l: l.startswith("{"))[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["isolated"]["frac"],
      j["roofline"]["traffic"], (j.get("cpu_baseline") or {}).get("value"), (j.get("v1_shape") or {}).get("value"))
