# round 3: per-kernel times of a Q4_1 7B decode loop (matvec_raw32_kernel<Q4_1, 512>)
cd /root/repo
O=gpurun_out/r3raw4; rm -rf $O; mkdir -p $O
python - <<'PY'
from ctransformers_amd import synth
synth.write_llama_gguf("/tmp/q41_7b.gguf", "llama-2-7b", "Q4_1", seed=1)
PY
cd /tmp && export TMPDIR=/tmp
CT_AMD_RAW_NT=512 CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o q41 -- python /root/repo/tools/decode_loop.py --model /tmp/q41_7b.gguf --prompt 8 --decode 24 > /root/repo/$O/prof.log 2>&1
cd /root/repo
python tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
head -16 $O/kernel_stats.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
