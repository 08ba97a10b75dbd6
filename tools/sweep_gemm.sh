export MS=${MS:-480,960,1440,2112,2880}
for cfg in "SBL_TILED_KU=4" "SBL_TILED_KU=2" "SBL_TILED_KU=1"; do
echo "== $cfg"; env $cfg timeout -k 10 120 python tools/bench_gemm2.py 2>&1 | grep "M=" | cut -c1-107 || exit 1
done
