cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
export TMPDIR=/tmp
for n in 2 8; do
timeout 900 python bench.py --gpus $n --dist-backend gloo --steps 2 --warmup 1 --batch 2000 --no-secondary --no-cpu-baseline --no-pmc > gpurun_out/r4l/dry_n$n.json 2> gpurun_out/r4l/dry_n$n.err
echo "dry run n=$n rc $?"; tail -c 1500 gpurun_out/r4l/dry_n$n.json; echo; tail -3 gpurun_out/r4l/dry_n$n.err | cut -c1-300
done
timeout 900 python bench.py --gpus 2 --dist-backend gloo --sharding frames --steps 2 --warmup 1 --batch 2000 --no-secondary --no-cpu-baseline --no-pmc > gpurun_out/r4l/dry_frames_n2.json 2> gpurun_out/r4l/dry_frames_n2.err
echo "dry run frames n=2 rc $?"; tail -c 600 gpurun_out/r4l/dry_frames_n2.json
