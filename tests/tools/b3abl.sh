cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "0 256" "1 256" "2 256" "4 256" "7 256" "0 128"; do
  set -- $a
  ALEPPO_B3_ABLATE=$1 ALEPPO_B3_GRID=$2 ALEPPO_BWD_STREAMS=1 ALEPPO_BWD3_FUSED=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b3a_$1_$2 -- python3 tests/tools/upd_time.py 3 > gpurun_out/b3a_$1_$2.log 2>&1 || exit 1
  python3 - $1 $2 <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/b3a_%s_%s/**/*kernel_trace.csv"%(sys.argv[1],sys.argv[2]),recursive=True)[0]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "conv3_bwd_fused" in r["Kernel_Name"]]
d.sort(); print("ablate",sys.argv[1],"grid",sys.argv[2],"median us (alone, one stream)",d[len(d)//2],"n",len(d))
PY
done
