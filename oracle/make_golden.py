#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: regenerate tests/golden/ref_golden.npz.

Builds oracle/_ref/ref_harness (oracle/build_ref.sh: our harness + the reference's own
gae.cc / buffer.cc / ppo/losses.cc / ppo/train.{h,cc} compiled where they lie under
/root/reference), runs `ref_harness gen`, and packs the OUTPUT arrays into one compressed npz.
Inputs are never stored: they are closed-form hash fills (oracle/hashfill.h == tests/hashfill.py).
Run in the build container only (the reference does not exist on the GPU box).
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])
    tmp = os.path.join(HERE, "_golden_tmp")
    os.makedirs(tmp, exist_ok=True)
    subprocess.check_call([os.path.join(HERE, "_ref", "ref_harness"), "gen", tmp])
    arrays = {}
    for f in sorted(os.listdir(tmp)):
        if f.endswith(".npy"):
            arrays[f[:-4]] = np.load(os.path.join(tmp, f), allow_pickle=False)
    out = os.path.join(ROOT, "tests", "golden", "ref_golden.npz")
    np.savez_compressed(out, **arrays)
    print(f"wrote {out}: {len(arrays)} arrays, {os.path.getsize(out)} bytes")


if __name__ == "__main__":
    sys.exit(main())
