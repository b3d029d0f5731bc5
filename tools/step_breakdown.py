"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py: launches and ms between two Adam launches."""
import collections
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + '/*/*kernel_trace.csv')[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    a, b = idx[1], idx[2]
    step = rows[a + 1:b + 1]

    def short(n):
        return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    c, t = collections.Counter(), collections.Counter()
    for r in step:
        c[short(r['Kernel_Name'])] += 1
        t[short(r['Kernel_Name'])] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    conv = sum(v for k, v in t.items() if 'conv_' in k or k.startswith('conv16_'))      # (conv16_* = the 16-channel tail kernels)
    print('launches/step', len(step), 'span ms %.2f' % ((int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6),
          'conv ms %.2f' % conv, 'non-conv ms %.2f' % (sum(t.values()) - conv))
    for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        print("%.3f ms %4d  %s" % (v, c[k], k))


if __name__ == "__main__":
    main(sys.argv[1])
