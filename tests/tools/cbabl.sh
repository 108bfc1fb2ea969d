# developer tool: timing-only ablations of conv_bwd_fused_kernel (ALEPPO_CB_ABLATE), kernel alone on one stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 8 16 31; do
  ALEPPO_CB_ABLATE=$a ALEPPO_BWD_STREAMS=1 ALEPPO_BWD_FUSED=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cba_$a -- python3 tests/tools/upd_time.py 3 > gpurun_out/cba_$a.log 2>&1 || exit 1
  python3 - $a <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/cba_%s/**/*kernel_trace.csv"%sys.argv[1],recursive=True)[0]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "conv_bwd_fused" in r["Kernel_Name"]]
d.sort(); print("ablate",sys.argv[1],"median us (alone, one stream)",d[len(d)//2],"n",len(d))
PY
done
