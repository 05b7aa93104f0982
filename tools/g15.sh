B="timeout 600 python bench.py --no-cpu-baseline --no-other-configs"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or bit_identical_to_reference_build" 2>&1 | tail -2
for i in 1 2; do
$B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused"
CT_AMD_QA_PHASE1=1 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused kernel, phase 1 + attention launch"
CT_AMD_FUSE_QA=0 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " plain"
done
