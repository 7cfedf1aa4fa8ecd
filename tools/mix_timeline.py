"""What runs beside what in the timed loop: analysis of a rocprofv3 --kernel-trace of `bench.py --quick`.

    python3 tools/mix_timeline.py <rocprof output dir> [skip_fraction=0.35]

Reads *kernel_trace.csv, drops the first `skip_fraction` of the wall span (warm-up, clocks), and prints
  1. per kernel class: launches, average duration IN THE MIX, share of the analysed span in which at least one
     launch of the class is running, and the time it runs with no kernel of another class beside it;
  2. how many kernels run concurrently (time-weighted histogram) and per hardware queue occupancy;
  3. class-by-class co-residency: share of class A's running time during which class B also runs;
  4. a table of 50-us buckets (the first 40 after the skip): launches of each class resident in the bucket.
Classes: desc (k_descriptor*), ori (k_orientation), detect, refine, scan, blur0 (64-row tiles: octave 0),
blur1 (32-row tiles with 256 lanes, duo<.,1>, and the strip-march level launches k_blur_march of octaves 0 .. 2 in a batch),
blurS (1024-lane small-octave launches and the one-launch tail), other.
"""
import collections
import csv
import glob
import sys


def klass(name):
    n = name.replace('popsift_hip::(anonymous namespace)::', '').replace('void ', '')
    if n.startswith('k_descriptor'):
        return 'desc'
    if n.startswith('k_orientation'):
        return 'ori'
    if n.startswith('k_detect'):
        return 'detect'
    if n.startswith('k_refine'):
        return 'refine'
    if n.startswith('k_scan'):
        return 'scan'
    if n.startswith('k_blur_tile64') or (n.startswith('k_blur_tile<') and ', 64,' in n) or n.startswith('k_blur_batch64'):
        return 'blur0'
    if n.startswith('k_blur_small') or n.startswith('k_pyr_tail') or (n.startswith('k_blur_duo') and n.rstrip('>').endswith(', 4')):
        return 'blurS'
    if n.startswith('k_blur'):
        return 'blur1'
    return 'other'


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), klass(r['Kernel_Name']), r.get('Queue_Id', '?')))
    rows.sort()
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    t0 = t_first + int((t_last - t_first) * skip)
    t1 = t_last - int((t_last - t_first) * 0.05)
    span = (t1 - t0) / 1e3
    classes = ['desc', 'ori', 'detect', 'refine', 'scan', 'blur0', 'blur1', 'blurS', 'other']
    print('# %s' % f)
    print('# analysed span %.1f us of %.1f us (skip %.2f), %d launches' % (span, (t_last - t_first) / 1e3, skip, len(rows)))

    # sweep
    ev = []
    for s, e, k, q in rows:
        if e <= t0 or s >= t1:
            continue
        ev.append((max(s, t0), 1, k, q))
        ev.append((min(e, t1), -1, k, q))
    ev.sort(key=lambda x: (x[0], x[1]))
    run = collections.Counter()
    qrun = collections.Counter()
    busy = collections.Counter()      # class -> ns with >= 1 running
    alone = collections.Counter()     # class -> ns with only this class running
    co = collections.Counter()        # (a, b) -> ns both running
    conc = collections.Counter()      # number of kernels running -> ns
    qbusy = collections.Counter()
    prev = t0
    for t, dlt, k, q in ev:
        dt = t - prev
        if dt > 0:
            live = [c for c in classes if run[c] > 0]
            conc[sum(run.values())] += dt
            for c in live:
                busy[c] += dt
                for c2 in live:
                    co[(c, c2)] += dt
            if len(live) == 1:
                alone[live[0]] += dt
            for qq, n in qrun.items():
                if n > 0:
                    qbusy[qq] += dt
        run[k] += dlt
        qrun[q] += dlt
        prev = t
    cnt = collections.Counter()
    dur = collections.Counter()
    for s, e, k, q in rows:
        if s >= t0 and e <= t1:
            cnt[k] += 1
            dur[k] += e - s
    print('\nclass     launches   avg_us_in_mix   sum_us    running_share   alone_share')
    for c in classes:
        if cnt[c]:
            print('%-8s %9d %15.2f %9.0f %14.3f %13.3f' % (c, cnt[c], dur[c] / cnt[c] / 1e3, dur[c] / 1e3,
                                                       busy[c] / 1e3 / span, alone[c] / 1e3 / span))
    print('\nkernels running at once (share of the span): ' +
          '  '.join('%d: %.3f' % (n, conc[n] / 1e3 / span) for n in sorted(conc)))
    print('hardware queues busy (share of the span):    ' +
          '  '.join('q%s: %.3f' % (q, qbusy[q] / 1e3 / span) for q in sorted(qbusy)))
    print('\nshare of the ROW class running time during which the COLUMN class also runs')
    print('%-8s' % '' + ''.join('%8s' % c for c in classes if cnt[c]))
    for a in classes:
        if not cnt[a]:
            continue
        print('%-8s' % a + ''.join('%8.2f' % (co[(a, b)] / busy[a] if busy[a] else 0.0) for b in classes if cnt[b]))
    # 50-us buckets
    print('\n50-us buckets after the skip: launches resident in the bucket, by class')
    print('%8s' % 't_us' + ''.join('%7s' % c for c in classes if cnt[c]))
    B = 50000
    for b in range(40):
        lo, hi = t0 + b * B, t0 + (b + 1) * B
        c = collections.Counter()
        for s, e, k, q in rows:
            if s < hi and e > lo:
                c[k] += 1
        print('%8d' % (b * 50) + ''.join('%7d' % c[k] for k in classes if cnt[k]))


if __name__ == '__main__':
    main()
