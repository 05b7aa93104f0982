#!/bin/bash
# the QKV launch touches the layer's K / V rows of the earlier positions (CT_AMD_TOUCH_KV=1) / not (0): token rate, alternating on one box;
# bench window (128-token prompt, 64 steps) through bench.py too; parity
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4F; mkdir -p $O
for v in 1 0 1 0; do
  ( CT_AMD_TOUCH_KV=$v timeout 300 python tools/gpu_sites.py touch$v 2>&1 | tail -1 ) >> $O/sites.txt
done
for v in 1 0 1 0; do
  ( CT_AMD_TOUCH_KV=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --steps 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('touch$v', d['value'], d['ms_per_step'])" ) >> $O/bench.txt
done
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not big_config and not eight_stages" 2>&1 | tail -3 ) > $O/pytest.txt
cat $O/sites.txt | cut -c1-330; cat $O/bench.txt; cat $O/pytest.txt
