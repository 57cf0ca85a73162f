cd $GRAFT_REPO_ROOT
O=gpurun_out/r2z; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_final.log 2>&1; tail -3 $O/pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
