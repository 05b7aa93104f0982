#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4u
( timeout 120 tools/experiments/mfma_rate.bin 2>&1 ) > gpurun_out/r4u/mfma_rate.txt; cat gpurun_out/r4u/mfma_rate.txt
