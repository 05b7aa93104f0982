#!/bin/bash
# prologue rewrite (per-block partial sums + one LDS read per lane, signed max / min instead of the first-maximum search): A/B against the
# previous build on one box (sites, alternating), in-kernel stamps, GPU parity of the changed code
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r4D; mkdir -p $O
for v in old new old new; do
  L=""; [ $v = old ] && L=$PWD/ctransformers_amd/lib_old/libctransformers.so
  ( SITES_LIB=$L timeout 300 python tools/gpu_sites.py $v 2>&1 | tail -1 ) >> $O/sites.txt
done
( SITES_LIB=$PWD/ctransformers_amd/lib_trace/libctransformers.so timeout 300 python tools/gpu_trace.py 2>&1 | grep -A5 "^qkv\|^wo\|^gate_up\|^down" | grep -v "^--" ) > $O/trace.txt
( timeout 1500 python -m pytest tests/test_weight_population.py tests/test_gpu_parity.py -m gpu -x -q -k "not big_config and not eight_stages and not full_size" 2>&1 | tail -5 ) > $O/pytest.txt
cat $O/sites.txt; cat $O/trace.txt; cat $O/pytest.txt
