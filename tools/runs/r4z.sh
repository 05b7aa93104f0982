#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4z
( timeout 300 tools/experiments/stream_probe2.bin 2>&1 ) > gpurun_out/r4z/stream_probe2.txt; cat gpurun_out/r4z/stream_probe2.txt
