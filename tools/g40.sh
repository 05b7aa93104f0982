cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/g40; mkdir -p $O
python tools/qa_trace.py 2>&1 | grep -v amdgpu | cut -c1-250 | tee $O/qa_trace.txt
