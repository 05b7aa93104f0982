cd /root/repo
O=gpurun_out/r3C; rm -rf $O; mkdir -p $O
timeout 900 python tools/scratch/bisect_mpt112.py > $O/bisect.txt 2>&1
cat $O/bisect.txt | grep -v "^ctransformers_amd\|deprecated" | tail -20
