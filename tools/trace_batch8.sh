cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/t8; rm -rf $O; mkdir -p $O
CT_AMD_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o pf -- python /root/repo/tools/decode_loop.py --model /tmp/l7b.gguf --prompt 64 --batch 8 --decode 1 > $O/prof.log 2>&1
cd /root/repo; python tools/pf_sites.py $O/prof | head -14
find $O -name "*.csv" -size +1M -delete
