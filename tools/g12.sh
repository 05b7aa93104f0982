python tools/qa_trace.py 2>&1 | grep -v amdgpu.ids
SITES_PROMPT=200 CT_AMD_FUSE_QA=0 python tools/gpu_trace.py 2>&1 | grep -v amdgpu.ids | grep -A8 "^qkv"
for i in 1 2; do
CT_AMD_FUSE_QA=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fused"
CT_AMD_QA_PHASE1=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " fused kernel, phase 1 + attention launch"
CT_AMD_FUSE_QA=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | head -c 120 | cut -c40-120; echo " plain"
done
