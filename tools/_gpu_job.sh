cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final3
timeout 150 python bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1 --batch 1000 --no-pmc --no-secondary --no-cpu-baseline > gpurun_out/final3/r3z_dryrun_n2.json 2> gpurun_out/final3/err.txt; echo "rc $?"
head -c 700 gpurun_out/final3/r3z_dryrun_n2.json; echo; tail -2 gpurun_out/final3/err.txt
