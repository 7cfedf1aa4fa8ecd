"""Per-launch timeline of the LAST image in a rocprofv3 kernel_trace.csv (compact)."""
import csv, glob, sys
d = sys.argv[1]; nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // nimg
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
tot = 0
for r in last:
    name = r['Kernel_Name'].replace('popsift_hip::(anonymous namespace)::', '').replace('void ', '')
    name = name.split('(')[0][:26]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    print("%-26s @%8.1f  %8.1f us  ends %8.1f  grid %8s  queue %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r['Grid_Size_X'], r.get('Queue_Id', '?')))
print("sum of kernel durations %.1f us, span %.1f us" % (tot / 1e3, (int(last[-1]['End_Timestamp']) - t0) / 1e3))
