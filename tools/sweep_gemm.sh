export MS=${MS:-480,960,1440}
for cfg in "X=1" "SBL_SKINNY_MAX_M=1500 SBL_SKINNY_MAX_TILES=4096"; do
echo "== $cfg"; env $cfg timeout -k 10 120 python tools/bench_gemm2.py 2>&1 | grep "M=" | cut -c1-107 || exit 1
done
