#!/bin/bash
# ring slots of the 32-block (Q8_0 / Q4_0) decode launches: 3 (lib_b3) / 4 — config 3 (Llama-2-7B Q8_0) through bench.py, alternating
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4K; mkdir -p $O
for v in b3 base b3 base; do
  L=""; [ $v != base ] && L=$PWD/ctransformers_amd/lib_$v/libctransformers.so
  ( CTRANSFORMERS_AMD_LIB=$L timeout 400 python bench.py --config 3 --no-cpu-baseline --no-other-configs --steps 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" ) >> $O/bench3.txt
done
cat $O/bench3.txt
