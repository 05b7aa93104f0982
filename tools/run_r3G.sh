cd /root/repo
O=gpurun_out/r3G; rm -rf $O; mkdir -p $O
run() { echo "## $*" >> $O/env_ab.txt; env "$@" python tools/scratch/attn_ab.py 192 2>/dev/null | tail -1 >> $O/env_ab.txt; }
run TAG=default
run TAG=kernarg1 HIP_FORCE_DEV_KERNARG=1
run TAG=kernarg0 HIP_FORCE_DEV_KERNARG=0
run TAG=pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run TAG=pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run TAG=barrier_value0 DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run TAG=default2
cat $O/env_ab.txt
