#!/bin/bash
# The round's measurement run on the GPU box (gpurun -- 'bash tools/measure_round.sh r05'): GPU suite, smoke, default bench (cpu_baseline, other_configs
# with configs 3 / 4 / 5, 2k-context fields), rocprofv3 kernel stats of the bench command, separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters /
# MFMA-busy of the prompt chunks), in-kernel traces (mat-vec sites, fused QKV + attention launch), in-stream stamps (token-step gaps, pipeline hops),
# per-site prefill timings, attention context scaling, the in-process pipeline on the one GPU (2 / 4 / 8 stages), the hand-off probe, legacy architectures.
# Summaries land in gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
# the file bench.py times: block pools from the reference's quantizer where oracle/_ref travelled with the snapshot (bench.py:_ref_quantizer)
if [ -f oracle/_ref/libctransformers_ref.so ]; then M=/tmp/ctamd_llama2_7b_q4km_refq.gguf; else M=/tmp/ctamd_llama2_7b_q4km_r2.gguf; fi
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1
export CTAMD_BENCH_MODEL=$M   # (behind the suite: its fixtures write the files themselves)
timeout 2400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for S in 2 4 8; do
  D=$(python -c "print(','.join(['0'] * $S))")
  CTAMD_BENCH_DEVICES=$D timeout 600 python bench.py --gpus $S --steps 64 --no-cpu-baseline --no-other-configs > $O/bench_gpus${S}_inprocess_one_gpu.json 2> $O/bench_$S.err
done
# the cross-stream forms on the one device (one stream per stage): events, and the flag form stages on distinct devices take (two-launch decode form)
CT_AMD_PP_SHARED_STREAM=0 CTAMD_BENCH_DEVICES=0,0,0,0,0,0,0,0 timeout 600 python bench.py --gpus 8 --steps 64 --no-cpu-baseline --no-other-configs --no-long-context > $O/bench_gpus8_events_one_gpu.json 2> $O/bench_8e.err
CT_AMD_PP_SHARED_STREAM=0 CT_AMD_HANDOFF=flag CT_AMD_FUSE_QA=0 CTAMD_BENCH_DEVICES=0,0,0,0 timeout 600 python bench.py --gpus 4 --steps 64 --no-cpu-baseline --no-other-configs --no-long-context > $O/bench_gpus4_flag_one_gpu.json 2> $O/bench_4f.err
( cd tools/experiments && hipcc -O2 --offload-arch=gfx950 -o handoff_probe handoff_probe.cpp 2>/dev/null; timeout 120 ./handoff_probe 8 ) > $O/handoff_probe.txt 2>&1
( python tools/pp_stamps.py; CT_AMD_DEVICES=0,0 python tools/pp_stamps.py; CT_AMD_DEVICES=0,0,0,0 python tools/pp_stamps.py; CT_AMD_PP_SHARED_STREAM=0 CT_AMD_DEVICES=0,0,0,0 python tools/pp_stamps.py; CT_AMD_PP_SHARED_STREAM=0 CT_AMD_HANDOFF=flag CT_AMD_FUSE_QA=0 CT_AMD_DEVICES=0,0,0,0 python tools/pp_stamps.py ) 2>&1 | grep -v amdgpu.ids > $O/pipeline_stamps.txt
( python tools/stamps.py; CT_AMD_SPEC=0 python tools/stamps.py; CT_AMD_SPEC=0 CT_AMD_HEAD_FOLD=0 python tools/stamps.py ) 2>&1 | grep -v amdgpu.ids > $O/token_step_stamps.txt
cd /tmp
CT_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o v9 -- python $R/bench.py --no-cpu-baseline --no-other-configs --no-long-context --steps 64 > $R/$O/prof.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o v9 -- python $R/tools/decode_loop.py --model $M --prompt 8 --decode 8 > $R/$O/pmc_fetch.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o v9 -- python $R/tools/decode_loop.py --model $M --prompt 8 --decode 8 > $R/$O/pmc_write.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $R/$O/pmc_sq -o p -- python $R/tools/decode_loop.py --model $M --prompt 8 --decode 6 > $R/$O/pmc_sq.log 2>&1
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_prefill -o pf -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 2 > $R/$O/prof_prefill.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_graph -o t -- python $R/tools/decode_loop.py --model $M --prompt 128 --decode 40 > $R/$O/tr_graph.log 2>&1
cd $R
python tools/prof_summary.py $O/prof > $O/kernel_stats_7b_q4km.txt 2>&1
python tools/timeline.py $O/tr_graph > $O/token_step_timeline.txt 2>&1
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) > $O/v9_pmc_traffic.json 2>&1
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); ( python tools/pmc_sq.py $f matvec_v9; python tools/pmc_sq.py $f qkv_attn9 ) > $O/v9_sq_counters_7b_q4km.txt 2>&1
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites_7b_q4km.txt 2>&1
CTAMD_BENCH_MODEL=$M bash tools/pmc_mfma.sh $O/pmc_mfma > $O/pmc_mfma.log 2>&1; cp $O/pmc_mfma/prefill_mfma_pmc.txt $O/prefill_mfma_pmc.txt 2>/dev/null
timeout 300 python tools/gpu_sites.py final > $O/v9_sites_7b_q4km.json 2> $O/sites.err
( timeout 300 python tools/gpu_trace.py; timeout 300 python tools/qa_trace.py ) > $O/v9_inkernel_trace_7b_q4km.txt 2> $O/trace.err
( timeout 300 python tools/ctx_scaling.py llama-7b-2l; timeout 300 python tools/ctx_scaling.py; echo "CT_AMD_ATTN_SHARE=0"; CT_AMD_ATTN_SHARE=0 timeout 300 python tools/ctx_scaling.py llama-7b-2l; CT_AMD_ATTN_SHARE=0 timeout 300 python tools/ctx_scaling.py ) > $O/ctx_scaling.txt 2>&1
timeout 300 python tools/prefill_sweep.py $M 8 16 32 64 128 > $O/prefill_sweep_7b_q4km.txt 2>&1
( timeout 600 python tools/ctx_scaling_free.py; timeout 300 python tools/attn_trace_free.py ) 2>&1 | grep -v amdgpu.ids > $O/decode_attn_free.txt   # order-free decode attention (CT_AMD_DECODE_ATTN=fast)
timeout 600 python tools/legacy_speed.py > $O/legacy_arch_speed.txt 2>&1
# the order-free prompt form (CT_AMD_PREFILL=fast): rates against the bit-identical form, kernel shares, SQ counters at 128- and 512-token prompts
( for n in 128 512 2048; do timeout 300 python tools/mm8_check.py llama-2-7b Q4_K_M $n 8 2304; done
  for n in 128 512 2048; do timeout 400 python tools/mm8_check.py llama-2-7b Q8_0 $n 8 2304; done
  timeout 400 python tools/mm8_check.py llama-70b-2l Q5_K_M 2048 8 2304; timeout 400 python tools/mm8_check.py falcon-40b-2l Q4_K_M 512 8 2304 ) > $O/prefill_fast_rates.txt 2>&1
( cd /tmp; for n in 128 512; do CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fast$n -o fast -- python $R/tools/mm8_check.py --worker fast llama-2-7b Q4_K_M $n 2 2304 > $R/$O/prof_fast$n.log 2>&1; done )
for n in 128 512; do python tools/prof_summary.py $O/prof_fast$n > $O/kernel_stats_fast_${n}tok_7b_q4km.txt 2>&1; done
timeout 900 bash tools/pmc_mm8.sh $O/pmc_mm8_128 llama-2-7b Q4_K_M 128 > $O/pmc_mm8_128.log 2>&1; cp $O/pmc_mm8_128/mm8_pmc.txt $O/mm8_pmc_128tok.txt 2>/dev/null
timeout 900 bash tools/pmc_mm8.sh $O/pmc_mm8_512 llama-2-7b Q4_K_M 512 > $O/pmc_mm8_512.log 2>&1; cp $O/pmc_mm8_512/mm8_pmc.txt $O/mm8_pmc_512tok.txt 2>/dev/null
python - <<PY
import json
for n in ("bench_default", "bench_gpus2_inprocess_one_gpu", "bench_gpus4_inprocess_one_gpu", "bench_gpus8_inprocess_one_gpu", "bench_gpus8_events_one_gpu", "bench_gpus4_flag_one_gpu"):
    try:
        d = json.loads([l for l in open("$O/%s.json" % n) if l.startswith("{")][-1])
        print(n, d["value"], "tok/s prefill", d["prefill_tok_s"], "2k", d.get("prefill_2k_tok_s"), d.get("decode_tok_s_at_2k"), d["config"]["parallelism"], "load", d["load_s"],
              "frac", (d.get("roofline") or {}).get("frac"), "tokfrac", d["token_roofline"]["frac_of_8TBps"],
              "cpu", (d.get("cpu_baseline") or {}).get("value"),
              "other", [(o.get("config"), o.get("decode_tok_s"), o.get("prefill_tok_s"), o.get("prefill_2k_tok_s"), o.get("decode_tok_s_at_2k"), o.get("frac_of_8TBps_per_token")) for o in d.get("other_configs") or []],
              "issue", (d["config"].get("host_issue") or {}).get("us_per_eval_total"))
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/pytest_gpu.txt; tail -1 $O/smoke.txt; head -14 $O/kernel_stats_7b_q4km.txt; head -6 $O/token_step_timeline.txt; cat $O/prefill_sweep_7b_q4km.txt; head -c 600 $O/v9_pmc_traffic.json; echo; cat $O/ctx_scaling.txt; cat $O/pipeline_stamps.txt $O/token_step_stamps.txt; cat $O/prefill_fast_rates.txt; head -12 $O/kernel_stats_fast_128tok_7b_q4km.txt; cat $O/mm8_pmc_512tok.txt | grep -E 'dispatches'
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete; rm -rf $O/pmc_mfma/p1 $O/pmc_mfma/p2 $O/pmc_mfma/p3 $O/pmc_mm8_128/p1 $O/pmc_mm8_128/p2 $O/pmc_mm8_512/p1 $O/pmc_mm8_512/p2
