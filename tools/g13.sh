SITES_PROMPT=200 CT_AMD_FUSE_QA=0 python tools/gpu_trace.py 2>&1 | grep -v amdgpu.ids | grep -A8 "^qkv\|^attn"
