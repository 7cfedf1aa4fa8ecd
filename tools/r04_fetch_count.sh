#!/bin/bash
# GPU box: HBM-side fetch of the keypoint kernels for several builds of the library (one rocprofv3 --pmc FETCH_SIZE pass each, 3 images).
#   tools/r04_fetch_count.sh <lib.so> ...     MB per image = 2 x FETCH_SIZE KiB x 1024 / 1e6 (gfx950 counts half of a wide read)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/fc; mkdir -p /tmp/fc
  POPSIFT_HIP_LIB=$R/popsift_amd/$lib timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc -- python3 $R/tools/prof_run.py 3 > /tmp/fc.log 2>&1 || { tail -5 /tmp/fc.log; exit 1; }
  python3 - "$lib" <<'P'
import csv, glob, sys, collections
tot = collections.Counter()
for f in glob.glob('/tmp/fc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        for k in ('k_descriptor(', 'k_orientation(', 'k_refine<', 'k_detect<'):
            if k in n: tot[k] += float(r['Counter_Value'])
print(sys.argv[1], {k: '%.0f MB' % (2 * v * 1024 / 1e6 / 3) for k, v in sorted(tot.items())})
P
done
