for v in cur prev cur prev; do if [ $v = prev ]; then export ALEPPO_LIB_PATH=$PWD/ale-libtorch-ppo_amd/libaleppo_prev.so; else unset ALEPPO_LIB_PATH; fi; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-v1 --no-host-legs > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err; echo $v $(grep -E "timed" gpurun_out/ab_$v.err | sed 's/.*-> //'); python - <<PY
import json
j=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1])
r=j['roofline']
print('   iso', {k:round(v['ms']*1e3,1) for k,v in r['update_kernels_isolated'].items()})
print('   co ', {k:round(v['ms']*1e3,1) for k,v in r['update_kernels'].items()}, r['update_other_kernel_ms'])
PY
done
