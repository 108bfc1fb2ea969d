#!/bin/bash
# Build the library of another git revision next to the current one for same-box A/B runs (tests/tools/ab.sh):
#   tests/tools/build_prev.sh <rev>   ->  ale-libtorch-ppo_amd/libaleppo_prev.so
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d /tmp/aleppo_prev.XXXX)
git -C "$root" archive "$rev" ale-libtorch-ppo_amd/csrc include | tar -x -C "$tmp"
make -j4 -C "$tmp/ale-libtorch-ppo_amd/csrc" > "$tmp/build.log" 2>&1 || { tail -20 "$tmp/build.log"; exit 1; }
cp "$tmp/ale-libtorch-ppo_amd/libaleppo.so" "$root/ale-libtorch-ppo_amd/libaleppo_prev.so"
rm -rf "$tmp"
echo "built $rev -> ale-libtorch-ppo_amd/libaleppo_prev.so"
