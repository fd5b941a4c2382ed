#!/bin/bash
# Kernel trace of the Swin step replayed as ONE HIP graph (no host between the launches: what is left idle on a
# stream is waiting for the other stream), compact CSV as tools/trace_step.sh:  bash tools/trace_step_graph.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-tsg}_swin
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --workload swin --graph on --steps 6 --warmup 2 --settle-seconds 0 --no-cpu-baseline --no-small-batch --no-roofline --no-e2e > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
cd $R && python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
with open(O + "/trace.csv", "w") as out:
    out.write("start,end,queue,stream,grid,wg,kernel\n")
    for r in rows:
        out.write(f'{r["Start_Timestamp"]},{r["End_Timestamp"]},{r.get("Queue_Id","")},{r.get("Stream_Id","")},{r["Grid_Size_X"]},{r["Workgroup_Size_X"]},"{r["Kernel_Name"][:90]}"\n')
PY
tail -1 $O/bench.log | cut -c1-200
rm -rf $O/trace
