"""Print a rocprofv3 kernel_stats.csv compactly; optional per-launch timeline of the last image."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('popsift_hip::(anonymous namespace)::', '').replace('void ', '')[:48]
    print("%-50s calls %5s tot %10.1f us avg %9.1f us %6s%%" % (n, r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, r['Percentage'][:5]))
