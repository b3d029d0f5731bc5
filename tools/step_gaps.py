"""Idle time of the GPU inside one training step of a rocprofv3 --kernel-trace CSV (default streams: kernels overlap):
union of the kernel intervals vs the step span, and the largest gaps with the kernels around them."""
import csv, glob, sys


def main(d):
    f = glob.glob(d + '/*/*kernel_trace.csv')[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    a, b = idx[-3], idx[-2]
    step = rows[a + 1:b + 1]
    t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
    cur_end, busy, gaps = t0, 0, []
    prev = None
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s > cur_end:
            gaps.append((s - cur_end, prev, r['Kernel_Name']))
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
        prev = r['Kernel_Name']
    short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50]
    print("span %.2f ms  busy(union) %.2f ms  idle %.2f ms in %d gaps; sum of kernel durations %.2f ms" % (
        (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps),
        sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6))
    for g, p, n in sorted(gaps, key=lambda x: -x[0])[:12]:
        print("  gap %.1f us  after %s  before %s" % (g / 1e3, short(p or ''), short(n)))


if __name__ == "__main__":
    main(sys.argv[1])
