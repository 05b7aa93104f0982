python tools/qa_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
B="timeout 600 python bench.py --no-cpu-baseline --no-other-configs"
$B 2>/dev/null | head -c 120 | cut -c40-120; echo " fused"
CT_AMD_FUSE_QA=0 $B 2>/dev/null | head -c 120 | cut -c40-120; echo " plain"
