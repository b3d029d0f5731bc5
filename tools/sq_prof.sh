#!/bin/bash
# SQ counters of the implicit-GEMM kernels in tools/bench_fwd.py: matrix-pipe busy cycles vs wave cycles vs clock.
#   usage (GPU box): tools/sq_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sq_$1
rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/bench_fwd.py > $O.log 2>&1
python3 - <<PY
import csv, glob, collections
d = '$O'
kt = glob.glob(d + '/*/*kernel_trace.csv')[0]
ct = glob.glob(d + '/*/*counter_collection.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = collections.defaultdict(dict)
name = {}
for r in csv.DictReader(open(ct)):
    rows[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    name[r['Dispatch_Id']] = (r['Kernel_Name'], r['Grid_Size'])
agg = collections.OrderedDict()
for k, c in rows.items():
    n = name[k]
    if 'conv_' not in n[0]:
        continue
    key = (n[0].replace('(anonymous namespace)::', '').split('(')[0][:60], n[1])
    a = agg.setdefault(key, collections.Counter())
    for cn, v in c.items():
        a[cn] += v
    a['n'] += 1
    a['us'] += dur.get(k, 0)
print("kernel grid n us/launch clockGHz mfma_busy%(of 1024 SIMD x active cycles) wait_any% wait_inst% active% valu_insts/wave_kcyc")
for key, a in agg.items():
    n = a['n']
    act = a['GRBM_GUI_ACTIVE'] / 8            # cycles per XCD summed -> average active cycles
    clk = act / n / (a['us'] / n) / 1e3 if a['us'] else 0
    busy = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 1024) * 100 if act else 0
    wc = a['SQ_WAVE_CYCLES']
    print(key[0], key[1], n, "%.1f" % (a['us'] / n), "%.3f" % clk, "%.1f" % busy, "%.1f %.1f %.1f" % (100 * a['SQ_WAIT_ANY'] / wc, 100 * a['SQ_WAIT_INST_ANY'] / wc, 100 * a['SQ_ACTIVE_INST_ANY'] / wc), "%.2f" % (a['SQ_INSTS_VALU'] / wc * 1000))
PY
