# round 6: kernel shares of a Falcon-40B Q4_K_M token step (BASELINE config 4) with the four-launch form, eager launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-long-context --no-fast-prefill > /dev/null 2>&1   # writes the file
CT_AMD_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f40 -o f40 -- python $R/bench.py --config 4 --steps 32 --warmup 4 --no-cpu-baseline --no-other-configs --no-long-context --no-fast-prefill > $R/gpurun_out/prof_f40.log 2>&1
cd $R && python tools/prof_summary.py gpurun_out/prof_f40 > gpurun_out/kernel_stats_falcon40b_q4km.txt 2>&1; head -16 gpurun_out/kernel_stats_falcon40b_q4km.txt
