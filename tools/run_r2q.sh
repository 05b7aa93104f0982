#!/bin/bash
# Round-2 measurement of the Q8_0 / Q4_0 prompt-chunk kernels (BASELINE config 3): per-site kernel durations of a 128-token prompt
# for the matrix-core form at 8 / 32 tokens per workgroup and the dot4 form.  Run on the GPU box:  bash tools/run_r2q.sh
O=gpurun_out/r2q; mkdir -p $O
python -c "
import bench
bench.SHAPE, bench.FTYPE, bench.MODEL = bench.CONFIGS[3]
bench.GEN_VERSION = 'synth-r2:%s:%s:seed1234' % (bench.SHAPE, bench.FTYPE)
bench.ensure_model(); print(bench.MODEL)" > $O/model.txt 2>&1
M=$(tail -1 $O/model.txt)
cd /tmp && export TMPDIR=/tmp
for v in "1 8" "1 32" "0 8"; do set -- $v
  CT_AMD_PF_MFMA=$1 CT_AMD_PF_TB=$2 CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_$1_$2 -o pf -- python /root/repo/tools/decode_loop.py --model $M --prompt 128 --decode 2 > /root/repo/$O/prof_$1_$2.log 2>&1
  python /root/repo/tools/pf_sites.py /root/repo/$O/prof_$1_$2 > /root/repo/$O/sites_mfma$1_tb$2.txt 2>&1
done
cd /root/repo
for f in $O/sites_*.txt; do echo "== $f"; head -12 $f; done
