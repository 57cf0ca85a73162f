cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2z; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o v -- $R/examples/icp_slam 60 640 480 > /tmp/ex.log 2>&1
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1); cp "$f" $O/icp_slam_vga_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph -o h -- $R/examples/icp_slam 60 1280 720 > /tmp/ex.log 2>&1
f=$(find /tmp/ph -name "*kernel_stats.csv" | head -1); cp "$f" $O/icp_slam_720p_kernel_stats.csv
python - <<'PY'
import csv
for tag in ("vga","720p"):
    rows=list(csv.reader(open("/root/repo/gpurun_out/r2z/icp_slam_%s_kernel_stats.csv"%tag)))
    tot=sum(float(r[2]) for r in rows[1:])/59/1e3
    print(tag, "GPU kernel time per frame us", round(tot,1))
    for r in rows[1:9]:
        print("   ", r[0].replace("void o3dmi::(anonymous namespace)::","")[:60], r[1], round(float(r[3])/1e3,1), "us; per frame", round(float(r[2])/59/1e3,1))
PY
