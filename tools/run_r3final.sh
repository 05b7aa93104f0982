# round 3, closing validation of the final tree: GPU suite, smoke(), default bench line
cd /root/repo
O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --no-other-configs > $O/bench_1.json 2> $O/bench_1.err; tail -1 $O/bench_1.json | head -c 400
