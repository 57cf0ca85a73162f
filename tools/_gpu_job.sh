cd /tmp && export TMPDIR=/tmp
cat > /tmp/ks.py <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'RayCastKernel' in r[0] or 'EstimateRange' in r[0]:
        print("   ", r[0].replace('void o3dmi::(anonymous namespace)::','')[:70], "calls", r[1], "avg us", round(float(r[3])/1e3,1), "max", round(float(r[6])/1e3,1))
PY
for S in 0 12 16 24 32; do
  rm -rf /tmp/prc; O3DMI_RAYCAST_SPLIT=$S timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prc -o r -- $GRAFT_REPO_ROOT/examples/icp_slam 40 640 480 > /tmp/ex.log 2>&1
  f=$(find /tmp/prc -name "*kernel_stats.csv" | head -1); echo "VGA split $S"; python /tmp/ks.py "$f"
done
for S in 0 16 32; do
  rm -rf /tmp/prc; O3DMI_RAYCAST_SPLIT=$S timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prc -o r -- $GRAFT_REPO_ROOT/examples/icp_slam 40 1280 720 > /tmp/ex.log 2>&1
  f=$(find /tmp/prc -name "*kernel_stats.csv" | head -1); echo "720p split $S"; python /tmp/ks.py "$f"
done
cd $GRAFT_REPO_ROOT
for S in 0 16; do for i in 1 2 3; do echo -n "split $S: "; O3DMI_RAYCAST_SPLIT=$S examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*'; done; done
