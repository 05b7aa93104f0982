mkdir -p gpurun_out/g4; R=$PWD; M=/tmp/ctamd_llama2_7b_q4km_r2.gguf
timeout 900 python -m pytest tests/test_greedy_chain.py -m gpu -x -q > gpurun_out/g4/pytest_chain.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g4/bench.json 2> gpurun_out/g4/bench.err
CT_AMD_SPEC=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g4/bench_nospec.json 2> gpurun_out/g4/bench_nospec.err
CT_AMD_SPEC=0 CT_AMD_HEAD_FOLD=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g4/bench_nofold.json 2> gpurun_out/g4/bench_nofold.err
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g4/bench2.json 2> gpurun_out/g4/bench2.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/g4/tr_spec -o t -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 40 > $R/gpurun_out/g4/tr_spec.log 2>&1
python $R/tools/timeline.py $R/gpurun_out/g4/tr_spec > $R/gpurun_out/g4/timeline_spec.txt 2>&1
cd $R; find gpurun_out/g4 -name "*.csv" -size +3M -delete
tail -5 gpurun_out/g4/pytest_chain.txt; for f in bench bench_nospec bench_nofold bench2; do head -c 220 gpurun_out/g4/$f.json | cut -c40-220; echo; done; grep -E "^1[56][0-9] |per token" gpurun_out/g4/timeline_spec.txt
