B="timeout 600 python bench.py --no-cpu-baseline --no-other-configs"
NP=$PWD/ctransformers_amd/lib_nopre/libctransformers.so
P8=$PWD/ctransformers_amd/lib_pre8/libctransformers.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden" 2>&1 | tail -2
for i in 1 2 3; do
$B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused, preload 13 dwords"
CT_AMD_FUSE_QA=0 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " plain, preload 13 dwords"
CTRANSFORMERS_AMD_LIB=$P8 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused, preload 6 dwords"
CTRANSFORMERS_AMD_LIB=$NP $B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused, no preload"
done
