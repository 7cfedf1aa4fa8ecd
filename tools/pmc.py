"""Summarise a rocprofv3 counter_collection.csv: per kernel, sum of each counter over its dispatches (last image)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].replace('popsift_hip::(anonymous namespace)::', '').replace('void ', '')[:28]
    k = agg.setdefault(n, collections.OrderedDict())
    k[r['Counter_Name']] = k.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    k['_n'] = k.get('_n', 0) + 1
names = []
for k in agg.values():
    for c in k:
        if c not in names and c != '_n': names.append(c)
print("%-30s" % "kernel", " ".join("%14s" % c[-14:] for c in names))
for n, k in agg.items():
    print("%-30s" % n, " ".join("%14.4g" % k.get(c, 0) for c in names))
