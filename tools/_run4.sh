cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cp bench_detail.json $O/bench_detail.json 2>/dev/null
tail -c 3000 $O/bench.json
echo; tail -3 $O/bench.err
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-configs4"
for w in 2 4 8; do
  for raw in 0 1; do
    O3DMI_SLICED_RAW=$raw timeout 200 $B --emulate-world $w --emulate-rank all > $O/emu_blocks_w${w}_raw${raw}.json 2> $O/emu_blocks_w${w}_raw${raw}.err
  done
  timeout 200 $B --emulate-world $w --emulate-rank all --sharding frames > $O/emu_frames_w${w}.json 2> $O/emu_frames_w${w}.err
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/emu_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d["config"]
        print("%-28s value %9d  ranks ms/step %s  kernel %.4f ms" % (
            os.path.basename(f)[4:-5], round(d["value"]),
            [round(x, 2) for x in c.get("emulated_ranks_ms_per_step") or []],
            d["roofline"].get("avg_kernel_ms") or 0))
    except Exception as e:
        print(os.path.basename(f), "failed", e, open(f[:-5] + ".err").read()[-400:])
PY
