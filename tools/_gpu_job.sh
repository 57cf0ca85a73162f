cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x -s 2>&1 | tail -40 > gpurun_out/r3a_pytest.log
echo "pytest rc=$?" >> gpurun_out/r3a_pytest.log
for n in 2 4 8; do
  timeout 600 python bench.py --gpus $n --dist-backend gloo --steps 2 --warmup 1 --batch 1000 --no-pmc --no-secondary --no-cpu-baseline > gpurun_out/r3_dryrun_n$n.json 2> gpurun_out/r3_dryrun_n$n.err
  echo "dryrun $n rc=$?"
done
timeout 300 python bench.py --emulate-world 8 --steps 5 --warmup 1 > gpurun_out/r3a_emulate_w8.json 2> gpurun_out/r3a_emulate_w8.err
timeout 300 python bench.py --emulate-world 2 --steps 5 --warmup 1 > gpurun_out/r3a_emulate_w2.json 2> gpurun_out/r3a_emulate_w2.err
timeout 600 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r3a_pytest.log
