mkdir -p gpurun_out/g1
( cd tools/experiments && timeout 120 ./handoff_probe 8 ) > gpurun_out/g1/handoff.txt 2>&1
timeout 300 python tools/host_gap.py > gpurun_out/g1/host_gap.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/g1/bench.json 2> gpurun_out/g1/bench.err
cat gpurun_out/g1/handoff.txt gpurun_out/g1/host_gap.txt; head -c 1500 gpurun_out/g1/bench.json
