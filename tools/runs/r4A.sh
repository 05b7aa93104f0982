#!/bin/bash
# persistent-grid probe: one launch per phase in a graph against one resident grid with flag-array / one-counter barriers
cd /root/repo/tools/experiments
mkdir -p /root/repo/gpurun_out
timeout 120 ./persist_probe.bin 32 > /root/repo/gpurun_out/persist_probe.txt 2>&1
echo "rc=$?" >> /root/repo/gpurun_out/persist_probe.txt
timeout 120 ./persist_probe.bin 8 >> /root/repo/gpurun_out/persist_probe.txt 2>&1
echo "rc=$?" >> /root/repo/gpurun_out/persist_probe.txt
cat /root/repo/gpurun_out/persist_probe.txt
