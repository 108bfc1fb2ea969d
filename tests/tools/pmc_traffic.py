"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes over tests/tools/kbench.py (developer tool).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p --output-format csv -- python3 tests/tools/kbench.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p --output-format csv -- python3 tests/tools/kbench.py
    rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o p --output-format csv -- python3 tests/tools/kbench.py
    python tests/tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/kt > profiles/r02_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB per the guide's HBM section; FETCH_SIZE is doubled on gfx950 (its correction).
Per-launch means; kernels are mapped to the bench's kernel classes by name.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

CLASS = [("fwd_fused_kernel", "conv_fwd"), ("conv_bwd_fused_kernel", "conv_bwd"), ("conv3_dgrad_tile_kernel", "conv3_dgrad"), ("LConv1Fwd", "conv1_fwd"), ("LConv2Fwd", "conv2_fwd"), ("LConv3Fwd", "conv3_fwd"), ("LConv3Dgrad", "conv3_dgrad"),
         ("LConv2Dgrad", "conv2_dgrad"), ("conv1_wgrad_shift_kernel", "conv1_wgrad"), ("LConv1Wgrad", "conv1_wgrad"), ("LConv2Wgrad", "conv2_wgrad"),
         ("LConv3Wgrad", "conv3_wgrad"), ("gemm_pipe_kernel<0", "fc_fwd"), ("gemm_pipe_kernelILi0", "fc_fwd"),
         ("gemm_pipe_kernel<1", "fc_dgrad"), ("gemm_pipe_kernelILi1", "fc_dgrad"), ("gemm_pipe_kernel<2", "fc_wgrad"),
         ("gemm_pipe_kernelILi2", "fc_wgrad"), ("gemm_tn_kernel", "fc_wgrad"), ("head_train_kernel", "head"),
         ("adam_kernel", "adam"), ("reduce_slabs_kernel", "reduce"), ("sumsq_kernel", "sumsq")]
# algorithmic bytes per launch at the C1 minibatch (4096 samples, H = 512): DESIGN.md section 5
ALGO_MB = {"conv_fwd": 287.3, "conv_bwd": 262.9, "conv1_fwd": 220.5, "conv2_fwd": 147.3, "conv3_fwd": 68.2, "fc_fwd": 34.1, "fc_dgrad": 55.6, "fc_wgrad": 29.9,
           "conv3_dgrad": 110.6, "conv3_wgrad": 68.2, "conv2_dgrad": 252.2, "conv2_wgrad": 147.3, "conv1_wgrad": 220.5}


def cls(name):
    for key, c in CLASS:
        if key in name:
            return c
    return None


def counter(d, cname):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = cls(r["Kernel_Name"])
            if c and r["Counter_Name"] == cname:
                acc[c][0] += float(r["Counter_Value"])
                acc[c][1] += 1
    return {c: v[0] / v[1] for c, v in acc.items()}, {c: v[1] for c, v in acc.items()}


fetch, n = counter(sys.argv[1], "FETCH_SIZE")
write, _ = counter(sys.argv[2], "WRITE_SIZE")
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[3], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        c = cls(r["Kernel_Name"])
        if c:
            dur[c][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dur[c][1] += 1
out = {}
for c in fetch:
    fm, wm = fetch[c] * 2 * 1024 / 1e6, write.get(c, 0.0) * 1024 / 1e6
    us = dur[c][0] / dur[c][1] if dur[c][1] else None
    out[c] = {"launches": n[c], "avg_us": round(us, 1) if us else None, "fetch_MB": round(fm, 1), "write_MB": round(wm, 1),
              "traffic_MB": round(fm + wm, 1), "algorithmic_MB": ALGO_MB.get(c),
              "GBps": round((fm + wm) * 1e6 / (us * 1e-6) / 1e9, 1) if us else None}
# stamp with the kernel sources the counters were taken on: bench.py quotes the record only for the same sources
import hashlib
_h = hashlib.sha256()
_d = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "ale-libtorch-ppo_amd", "csrc")
for _f in sorted(os.listdir(_d)):
    if _f.endswith((".hip", ".hpp")):
        _h.update(open(os.path.join(_d, _f), "rb").read())
out["kernel_source_sha16"] = _h.hexdigest()[:16]
print(json.dumps(out, indent=1))
