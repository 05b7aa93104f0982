# per-site sweep of the launch shapes of the order-free prompt kernels: prompt rate with ONE site on the order-free kernels (the others exact) per forced shape
python tools/mm8_check.py llama-2-7b Q4_K_M 128 0 | tail -1
for site in qkv wo gate_up down; do
  for sh in 2,1 2,2 2,4 1,8; do
    r=$(CT_AMD_MM8_SITES=$site CT_AMD_MM8_SHAPE=$sh python tools/mm8_check.py llama-2-7b Q4_K_M 128 0 2>&1 | tail -1)
    echo "$site $sh: $r"
  done
done
