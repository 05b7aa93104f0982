# round 3: decode speed of full-size 7B files of the file-layout ftypes on the final tree
cd /root/repo
O=gpurun_out/r3raw6; rm -rf $O; mkdir -p $O
timeout 600 python tools/scratch/raw32_speed.py Q4_1 Q5_0 Q5_1 F16 > $O/speed.txt 2>&1; cat $O/speed.txt
