python tools/stamps.py 2>&1 | grep -v amdgpu.ids
CT_AMD_SPEC=0 python tools/stamps.py 2>&1 | grep -v amdgpu.ids
CT_AMD_HEAD_FOLD=0 python tools/stamps.py 2>&1 | grep -v amdgpu.ids
