export MS=4352
for cfg in "SBL_SEG_TARGET=256" "SBL_SEG_TARGET=512" "SBL_SEG_TARGET=768" "SBL_SEG_TARGET=1024" "SBL_SEG_TARGET=1536" "SBL_SEG_TARGET=768 SBL_SEG_KU=4" "SBL_SEG_TARGET=1536 SBL_SEG_KU=4"; do
echo "== $cfg"; env $cfg timeout -k 10 120 python tools/bench_gemm2.py 2>&1 | grep "M=" | cut -c1-22,108- || exit 1
done
