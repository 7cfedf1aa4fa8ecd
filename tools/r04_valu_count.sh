#!/bin/bash
# GPU box: dynamic vector / scalar / LDS instruction counts of k_descriptor for several builds of the library (one rocprofv3 --pmc pass each,
# tools/prof_run.py 3 = three config-2 images).   [KPAT=k_orientation] tools/r04_valu_count.sh <lib.so> ...   (default kernel: k_descriptor)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/vc; mkdir -p /tmp/vc
  POPSIFT_HIP_LIB=$R/popsift_amd/$lib timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d /tmp/vc -- python3 $R/tools/prof_run.py 3 > /tmp/vc.log 2>&1 || { tail -5 /tmp/vc.log; exit 1; }
  KPAT=${KPAT:-k_descriptor} python3 - "$lib" <<'P'
import csv, glob, os, sys, collections
KPAT = os.environ.get('KPAT', 'k_descriptor') + '('
tot = collections.Counter()
for f in glob.glob('/tmp/vc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if (KPAT in r['Kernel_Name'] and 'grid' not in r['Kernel_Name'] and 'notile' not in r['Kernel_Name']):
            tot[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[1], {k: round(v / 3) for k, v in sorted(tot.items())})
P
done
