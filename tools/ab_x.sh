R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 tools/ab_compare.py $R/build_variants/v1.so $R/popsift_amd/libpopsift_hip.so --big 2>&1 | tail -2
echo "== product"; python3 tools/sparse_stages.py 2>&1 | grep default
echo "== v1"; POPSIFT_HIP_LIB=$R/build_variants/v1.so python3 tools/sparse_stages.py 2>&1 | grep default
for i in 1 2 3; do
echo "== product"; timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 2>/dev/null | tail -1 | cut -c1-60
echo "== v1"; POPSIFT_HIP_LIB=$R/build_variants/v1.so timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 2>/dev/null | tail -1 | cut -c1-60
done
