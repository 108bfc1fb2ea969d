#!/usr/bin/env bash
# TEST INFRASTRUCTURE: build oracle/_ref/ref_harness from our harness (oracle/ref_harness.cc) plus
# the reference's own hot-path sources compiled WHERE THEY LIE under /root/reference (never copied):
#   src/ai/gae.cc src/ai/buffer.cc src/ai/ppo/losses.cc src/ai/ppo/train.cc (+ headers)
# against the pip libtorch (CPU).  The reference's own build system (bazel) is not used.
# -U__linux__ drops the reference's CUDA-graph block (src/ai/ppo/train.h:4-8,159-198 includes
# ATen/cuda/CUDAGraph.h -> cuda.h, absent on ROCm) and leaves its CPU train() intact.
# Outputs only into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${ALEPPO_REFERENCE:-/root/reference}"
if [ ! -d "$REF/src/ai" ]; then
  echo "build_ref: $REF not present (GPU box) - keeping prebuilt oracle/_ref" >&2
  exit 0
fi
T="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
mkdir -p "$HERE/_ref"
OUT="$HERE/_ref/ref_harness"
if [ "$OUT" -nt "$HERE/ref_harness.cc" ] && [ "$OUT" -nt "$HERE/hashfill.h" ]; then
  echo "build_ref: up to date"
  exit 0
fi
g++ -std=c++20 -O2 -U__linux__ -D_GLIBCXX_USE_CXX11_ABI=1 \
  -I"$REF/src" -I"$HERE" -I"$T/include" -I"$T/include/torch/csrc/api/include" \
  "$HERE/ref_harness.cc" "$REF/src/ai/gae.cc" "$REF/src/ai/buffer.cc" \
  "$REF/src/ai/ppo/losses.cc" "$REF/src/ai/ppo/train.cc" \
  -L"$T/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$T/lib" -o "$OUT"
echo "build_ref: built $OUT"
