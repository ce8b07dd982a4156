cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05i
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05i/test.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05i/smoke.txt 2>&1
cat gpurun_out/r05i/test.txt; tail -3 gpurun_out/r05i/smoke.txt
