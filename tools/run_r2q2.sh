#!/bin/bash
# Config 3 prompt path: eager vs hipGraph replay (bench numbers), and the per-site kernel durations INSIDE the graph replay.
O=gpurun_out/r2q; mkdir -p $O
(CT_AMD_GRAPH=0 timeout 600 python bench.py --config 3 --no-cpu-baseline --steps 32 2>/dev/null | tail -1) > $O/bench3_eager.json
(timeout 600 python bench.py --config 3 --no-cpu-baseline --steps 32 2>/dev/null | tail -1) > $O/bench3_graph.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_graph -o pf -- python /root/repo/bench.py --config 3 --no-cpu-baseline --steps 8 > /root/repo/$O/prof_graph.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_graph > $O/sites_graph.txt 2>&1
python -c "
import json
for f in ('eager','graph'):
    d=json.load(open('$O/bench3_%s.json' % f)); print(f, d['value'], d['prefill_tok_s'], d['prefill_cold_tok_s'])"
head -12 $O/sites_graph.txt
