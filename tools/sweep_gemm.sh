export MS=${MS:-960,1440,2112}
for cfg in "X=1" "SBL_SPLIT_TARGET=512 SBL_SPLIT_TILES=384" "SBL_SPLIT_TARGET=768 SBL_SPLIT_TILES=512"; do
echo "== $cfg"; env $cfg timeout -k 10 120 python tools/bench_gemm2.py 2>&1 | grep "M=" | cut -c1-107 || exit 1
done
