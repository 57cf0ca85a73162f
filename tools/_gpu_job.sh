cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/bench_raycast.py
python tools/bench_raycast.py --attrs depth
python tools/bench_raycast.py --hd
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rc_pmc -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 10 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rc_pmc2 -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 10 > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rc_pmc3 -o rc -- python $GRAFT_REPO_ROOT/tools/bench_raycast.py --repeat 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
for d in ("rc_pmc","rc_pmc2","rc_pmc3"):
    acc={}; disp=set()
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "RayCastKernel" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n=max(1,len(disp))
    print(d, n, {k:round(v/n) for k,v in acc.items()})
PY
