# round 6: chunk sizes of the order-free form (CT_AMD_PF_CAP) and the other file types
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/mm8_caps.log
for cap in 512 1024 2048; do
  echo "== 7B Q4_K_M 2048 tokens CT_AMD_PF_CAP=$cap" >> gpurun_out/mm8_caps.log
  CT_AMD_PF_CAP=$cap timeout 300 python tools/mm8_check.py llama-2-7b Q4_K_M 2048 4 2304 >> gpurun_out/mm8_caps.log 2>&1
done
for n in 128 512 2048; do
  echo "== 7B Q8_0 $n tokens" >> gpurun_out/mm8_caps.log
  timeout 400 python tools/mm8_check.py llama-2-7b Q8_0 $n 4 2304 >> gpurun_out/mm8_caps.log 2>&1
done
echo "== 70B-2l Q5_K_M 2048 tokens" >> gpurun_out/mm8_caps.log
timeout 400 python tools/mm8_check.py llama-70b-2l Q5_K_M 2048 4 2304 >> gpurun_out/mm8_caps.log 2>&1
echo "== 70B-2l Q4_K_M 2048 tokens cap 1024" >> gpurun_out/mm8_caps.log
CT_AMD_PF_CAP=1024 timeout 400 python tools/mm8_check.py llama-70b-2l Q4_K_M 2048 4 2304 >> gpurun_out/mm8_caps.log 2>&1
cat gpurun_out/mm8_caps.log
