#!/bin/bash
# rocprofv3 kernel durations of tools/bench_polar.py --quick (K1 variants); usage: tools/k1_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_k1_$1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_k1_$1 -- python3 $GRAFT_REPO_ROOT/tools/bench_polar.py --quick > $GRAFT_REPO_ROOT/gpurun_out/prof_k1_$1.log 2>&1
python3 - <<PY
import csv,glob,itertools
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_k1_$1/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'polar_kernel' in r['Kernel_Name']]
seq=[(r['Kernel_Name'].split('(')[1][-30:] if False else r['Kernel_Name'][28:62], r['Grid_Size_X'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
for k,g in itertools.groupby(seq,key=lambda x:(x[0],x[1])):
    d=sorted(x[2] for x in g)
    print(k, len(d), "median us %.1f min %.1f"%(d[len(d)//2], d[0]))
PY
