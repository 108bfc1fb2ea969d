# developer tool: re-take only the rocprofv3 kernel trace of the bench (tests/tools/collect_profiles.sh's last step)
set -o pipefail
TAG=${1:-r03}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-v1 --no-host-legs > $OUT/bench_prof.log 2> $OUT/bench_prof.err || exit 1
python3 tests/tools/rocpd_stats.py $OUT/prof_bench/bench_results.db $OUT/bench_kernel_stats > $OUT/bench_kernel_stats.txt || exit 1
rm -rf $OUT/prof_bench
head -8 $OUT/bench_kernel_stats.txt
