B="timeout 600 python bench.py --no-cpu-baseline --no-other-configs"
BF=$PWD/ctransformers_amd/lib_barrier_first/libctransformers.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or bit_identical_to_reference_build" 2>&1 | tail -2
for i in 1 2 3; do
$B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused, barrier behind the first requests"
CTRANSFORMERS_AMD_LIB=$BF $B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused, barrier first"
CT_AMD_FUSE_QA=0 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " plain, barrier behind the first requests"
CTRANSFORMERS_AMD_LIB=$BF CT_AMD_FUSE_QA=0 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " plain, barrier first"
done
