cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final4
cd /tmp
timeout 100 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p1 -o p -- python $R/tools/bench_raycast.py --repeat 10 > /dev/null 2>&1
timeout 100 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/p2 -o p -- python $R/tools/bench_raycast.py --repeat 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(list)
for d in ("/tmp/p1", "/tmp/p2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "RayCastKernel" in r["Kernel_Name"]:
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for disp, c in per.items():
            for k, v in c.items():
                out[k].append(v)
res = {k: sum(v) / len(v) for k, v in out.items()}
res["dispatches"] = {k: len(v) for k, v in out.items()}
json.dump(res, open("/root/repo/gpurun_out/final4/r3z_raycast_counters_raw.json", "w"), indent=1)
print(json.dumps(res))
PY
