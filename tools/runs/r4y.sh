#!/bin/bash
# full -m gpu suite + smoke + default bench on one box (the driver's round-end sequence)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest_gpu.txt
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > $O/smoke.txt
( timeout 2400 python bench.py 2>&1 | tail -1 ) > $O/bench_default.json
cat $O/pytest_gpu.txt $O/smoke.txt; python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "prefill_tok_s", "load_s")}, d["roofline"]["frac"], d["token_roofline"]["frac_of_8TBps"])
print(d.get("other_configs")); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["threads"])
PY
