cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r3q_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err
echo "bench rc=$? seconds=$SECONDS"
cat gpurun_out/r3q_pytest.log
