cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for hd in "" "--hd"; do
O3DMI_RAYCAST_XCD_BANDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rcA$hd -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 20 $hd > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rcB$hd -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 20 $hd > /dev/null 2>&1
done
python - <<'PY'
import csv
for d in ("/tmp/rcA","/tmp/rcB","/tmp/rcA--hd","/tmp/rcB--hd"):
    for r in csv.DictReader(open(d+"/rc_kernel_stats.csv")):
        if "RayCastKernel" in r["Name"]: print(d, r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
