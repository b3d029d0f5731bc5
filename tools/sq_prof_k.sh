#!/bin/bash
# SQ counters of the bf16 attention kernels (tools/$2): matrix-pipe busy, wait / issue split, LDS activity and conflicts
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sq_$1
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $GRAFT_REPO_ROOT/tools/$2 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $O/p2 -- python3 $GRAFT_REPO_ROOT/tools/$2 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --output-format csv -d $O/p3 -- python3 $GRAFT_REPO_ROOT/tools/$2 > $O/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    try:
        ct = glob.glob('$O/' + p + '/*/*counter_collection.csv')[0]
    except IndexError:
        print(p, "no counters (see", '$O/' + p + '.log)'); continue
    kt = glob.glob('$O/' + p + '/*/*kernel_trace.csv')[0]
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt))}
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(ct)):
        if '$1' not in r['Kernel_Name']: continue
        key = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        a = agg.setdefault(key, collections.Counter())
        a[r['Counter_Name']] += float(r['Counter_Value'])
        a['_ids', r['Dispatch_Id']] = 1
    for key, a in agg.items():
        ids = [k[1] for k in a if isinstance(k, tuple)]
        us = sum(dur.get(i, 0) for i in ids) / max(len(ids), 1)
        vals = {k: v for k, v in a.items() if not isinstance(k, tuple)}
        wc = vals.get('SQ_WAVE_CYCLES', 1)
        line = " ".join(f"{k}={v / wc * 100:.1f}%wc" if k != 'SQ_WAVE_CYCLES' else f"wavecyc={v:.3g}" for k, v in vals.items())
        if 'GRBM_GUI_ACTIVE' in vals:
            act = vals['GRBM_GUI_ACTIVE'] / 8
            line += f" | mfma_busy={vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 1024) * 100:.1f}% of SIMD cycles"
        print(p, key, f"n={len(ids)} us={us:.1f}", line)
PY
