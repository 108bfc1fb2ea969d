#!/bin/bash
# Round-end evidence run on the MI355X box (developer tool; run through gpurun from the repo root):
#   bash tests/tools/collect_profiles.sh r02
# writes under gpurun_out/<tag>/ : kernel trace + FETCH_SIZE / WRITE_SIZE / SQ counter passes over
# tests/tools/kbench.py (one update at the BASELINE configs[1] minibatch shape; counters in their own runs, never
# combined with a trace), then the un-profiled bench line (it quotes the traffic record just taken: same kernel
# sources) and a rocprofv3 kernel trace of the same bench command.  tests/tools/finish_profiles.py turns them
# into profiles/<tag>_*.
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "[profiles] kbench kernel trace"; date
rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python3 tests/tools/kbench.py > $OUT/kbench_kt.log 2>&1 || exit 1
echo "[profiles] kbench FETCH_SIZE"; date
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 tests/tools/kbench.py > $OUT/kbench_fetch.log 2>&1 || exit 1
echo "[profiles] kbench WRITE_SIZE"; date
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 tests/tools/kbench.py > $OUT/kbench_write.log 2>&1 || exit 1
echo "[profiles] kbench SQ counters"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_sq -o p --output-format csv -- python3 tests/tools/kbench.py > $OUT/kbench_sq.log 2>&1 || exit 1
python3 tests/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/kt > $OUT/pmc_traffic.json || exit 1
python3 tests/tools/pmc_summary.py $OUT/pmc_sq > $OUT/pmc_sq.json || exit 1
cp $OUT/pmc_traffic.json profiles/${TAG}_pmc_traffic.json   # (on the box: lets the bench runs below quote it)
echo "[profiles] bench (un-profiled)"; date
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_final.log 2> $OUT/bench_final.err || exit 1
echo "[profiles] bench under rocprofv3 --kernel-trace --stats"; date
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-v1 --no-host-legs > $OUT/bench_prof.log 2> $OUT/bench_prof.err || exit 1
# keep the merge-back small: the raw rocpd database is summarised here, only the summary travels
python3 tests/tools/rocpd_stats.py $OUT/prof_bench/bench_results.db $OUT/bench_kernel_stats > $OUT/bench_kernel_stats.txt || exit 1
rm -rf $OUT/prof_bench $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
echo "[profiles] done"; date
