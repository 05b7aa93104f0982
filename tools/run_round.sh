cd /root/repo
O=gpurun_out/r10; rm -rf $O; mkdir -p $O
export CTAMD_BENCH_MODEL=/tmp/l7b.gguf
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch -o v6 -- python /root/repo/tools/decode_loop.py --model /tmp/l7b.gguf --shape llama-2-7b --prompt 8 --decode 8 > /root/repo/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write -o v6 -- python /root/repo/tools/decode_loop.py --model /tmp/l7b.gguf --prompt 8 --decode 8 > /root/repo/$O/pmc_write.log 2>&1
cd /root/repo
python tools/pmc_traffic.py $O/pmc_fetch/v6_counter_collection.csv $O/pmc_write/v6_counter_collection.csv > $O/pmc_traffic.json 2>&1
cp $O/pmc_traffic.json profiles/r01_v6_pmc_traffic.json
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
cd /tmp
CT_AMD_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -o v6 -- python /root/repo/bench.py --no-cpu-baseline --steps 64 > /root/repo/$O/prof.log 2>&1
cd /root/repo
python tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1
cd /tmp
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_prefill -o pf -- python /root/repo/tools/decode_loop.py --model /tmp/l7b.gguf --prompt 128 --decode 2 > /root/repo/$O/prof_prefill.log 2>&1
cd /root/repo
python tools/pf_sites.py $O/prof_prefill > $O/prefill_sites.txt 2>&1
timeout 300 python tools/prefill_sweep.py /tmp/l7b.gguf 8 16 32 64 128 > $O/prefill_sweep.txt 2>&1
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
head -12 $O/kernel_stats.txt; head -14 $O/prefill_sites.txt; cat $O/prefill_sweep.txt
