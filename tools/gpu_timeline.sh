#!/bin/bash
# kernel timeline of the timed loop as it really runs (side streams, no --serial-launches): start / end of every
# kernel of the last iteration relative to its first kernel
TAG=${1:-tl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/trace -o t -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $REPO/$OUT/bench.json 2> $REPO/$OUT/bench.err); echo "rc=$?"
python - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# iterations start with the gramian of the item half: find the starts of gramian_partial kernels
g=[i for i,r in enumerate(rows) if "gramian_partial" in r[2]]
# two gramians per iteration (items half, users half); take the last full iteration
start=g[-2] if len(g)>=2 else 0
t0=rows[start][0]
out=open("$OUT/timeline.txt","w")
for s,e,n in rows[start:]:
    n=n.replace("rsparse_hip::(anonymous namespace)::","").replace("void ","")
    line="%9.3f %9.3f %8.3f ms  %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n[:110])
    out.write(line+"\n")
    if (e-s)/1e6>0.3: print(line)
PY
find $OUT/trace -name "*kernel_trace.csv" -size +8M -delete
