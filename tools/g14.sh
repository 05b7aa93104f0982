for i in 1 2; do
CT_AMD_QA_PHASE1=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " group-mapped qkv (fused kernel, phase 1) + attention launch"
CT_AMD_QA_PHASE1=5 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " plain-mapped qkv (fused kernel, phase 1) + attention launch"
CT_AMD_FUSE_QA=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " plain"
done
