#!/bin/bash
# First contact with a node of more than one MI355X (no such node was available to rounds 1-6: every multi-stage result in profiles/ is N stages on ONE GPU).
# One pass over everything that has never crossed a device boundary, cheapest first, each step with its own time limit; results under profiles/<tag>_scale_*.
#   usage (on the node, from the repo root):  bash tools/first_contact_multigpu.sh [tag]        e.g. r07
# What to read afterwards:
#   <tag>_scale_devices.txt        visible devices, peer access matrix
#   <tag>_scale_tests.txt          the two-device tests (bit-identical to the reference build / goldens), the hand-off self-check's verdict
#   <tag>_scale_handoff_probe.txt  hop latency of the in-stream hand-off (peer stores + stream wait) against copy + event, across two devices
#   <tag>_scale_gpusN_<form>.json  bench.py --gpus N lines: in-process pipeline, hand-off forms flag (default on distinct devices) and event
#   <tag>_scale_torchrun_gpusN.json  the driver's launch form (one rank per GPU; rank 0 drives the stages when it sees every device)
TAG=${1:-r07}; R=$PWD; O=$R/profiles; mkdir -p $O gpurun_out
N=$(python - <<'PY'
import ctypes
try:
    h = ctypes.CDLL("libamdhip64.so"); n = ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); print(n.value)
except Exception:
    print(0)
PY
)
echo "visible HIP devices: $N" | tee $O/${TAG}_scale_devices.txt
if [ "$N" -lt 2 ]; then echo "fewer than two devices: nothing to do"; exit 0; fi
rocm-smi --showtopo >> $O/${TAG}_scale_devices.txt 2>&1 || true

# 1. the two-device tests: the flag form's first real run starts with the load-time self-check (pipeline.cc:handoff_self_check) — its verdict is on stderr
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "two_real_devices or inprocess_pipeline" -s 2>&1 | tail -40 > $O/${TAG}_scale_tests.txt
grep -h "self-check" $O/${TAG}_scale_tests.txt || echo "(hand-off self-check: passed silently)" >> $O/${TAG}_scale_tests.txt

# 2. hop latency across two devices
if [ -f tools/experiments/handoff_probe.cpp ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/handoff_probe tools/experiments/handoff_probe.cpp 2>/dev/null && timeout 120 /tmp/handoff_probe 8 0 1 > $O/${TAG}_scale_handoff_probe.txt 2>&1
fi

# 3. the scaling curve, in-process pipeline (one process drives the stages): both hand-off forms
for G in 2 4 8; do
  [ "$G" -gt "$N" ] && continue
  for FORM in flag event; do
    CT_AMD_HANDOFF=$FORM timeout 900 python bench.py --gpus $G --no-cpu-baseline --no-other-configs --no-long-context --steps 64 --warmup 8 2> gpurun_out/scale_${G}_${FORM}.err | grep '^{' | tail -1 > $O/${TAG}_scale_gpus${G}_${FORM}.json
  done
done

# 4. the driver's launch form
for G in 2 4 8; do
  [ "$G" -gt "$N" ] && continue
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port $((29500 + G)) bench.py --gpus $G --no-cpu-baseline --no-other-configs --no-long-context --steps 64 --warmup 8 2> gpurun_out/scale_torchrun_${G}.err | grep '^{' | tail -1 > $O/${TAG}_scale_torchrun_gpus${G}.json
done
python - <<PY
import glob, json
for f in sorted(glob.glob("$O/${TAG}_scale_*gpus*.json")):
    try:
        d = json.loads(open(f).read())
        print("%-48s %8.1f tok/s  prefill %8.1f  handoff %s" % (f.split("/")[-1], d["value"], d.get("prefill_tok_s", 0), d["config"].get("handoff")))
    except Exception as e:
        print(f.split("/")[-1], "no line:", e)
PY
