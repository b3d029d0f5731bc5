#!/bin/bash
# second SQ pass: LDS / scalar / VMEM issue activity of the conv kernels in tools/bench_fwd.py
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sq2_$1
rm -rf $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/bench_fwd.py > $O.log 2>&1
python3 - <<PY
import csv, glob, collections
d = '$O'
ct = glob.glob(d + '/*/*counter_collection.csv')[0]
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
print("kernel grid | % of wave cycles: lds_inst sca_inst vmem_inst valu_inst misc_inst | lds_idx_active/wavecyc% bank_conflict/idx_active%")
for key, a in agg.items():
    wc = a['SQ_WAVE_CYCLES']
    print(key[0], key[1], "| %.1f %.1f %.1f %.1f %.1f | %.2f %.1f" % (100 * a['SQ_ACTIVE_INST_LDS'] / wc, 100 * a['SQ_ACTIVE_INST_SCA'] / wc,
          100 * a['SQ_ACTIVE_INST_VMEM'] / wc, 100 * a['SQ_ACTIVE_INST_VALU'] / wc, 100 * a['SQ_ACTIVE_INST_MISC'] / wc,
          100 * a['SQ_LDS_IDX_ACTIVE'] / wc, 100 * a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1)))
PY
