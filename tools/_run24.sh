cd $GRAFT_REPO_ROOT
O=gpurun_out/r24; mkdir -p $O
T0=$(date +%s)
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
T1=$(date +%s); echo "bench seconds: $((T1-T0))" | tee $O/bench_seconds.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/final_profiles.sh r6f > $O/final.log 2>&1
tail -3 $O/final.log
