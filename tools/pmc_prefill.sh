# HBM fetch per launch of the prompt-chunk kernels for a 128-token prompt (8 token groups re-reading every weight tile:
# FETCH_SIZE shows how much of that the L2s absorb).  usage (GPU box): bash tools/pmc_prefill.sh
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/pmc_prefill; rm -rf $O; mkdir -p $O
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o pf -- python /root/repo/tools/decode_loop.py --model /tmp/l7b.gguf --shape llama-2-7b --prompt 128 --decode 1 > $O/fetch.log 2>&1
cd /root/repo
python tools/pmc_traffic.py $O/fetch/pf_counter_collection.csv > $O/pmc_traffic.json 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_prefill/pmc_traffic.json"))
for k, v in d["kernels"].items():
    if "pf" in k or "attn" in k:
        print("%-60s dispatches %4d  fetch %8.1f MB per dispatch" % (k[:60], v["dispatches"], v["fetch_bytes_per_dispatch"] / 1e6))
PY
find $O -name "*.csv" -size +1M -delete
