cd $GRAFT_REPO_ROOT
lscpu | grep -i "numa\|socket\|^CPU(s)\|model name" | head -12
for d in /sys/class/drm/card*/device; do echo $d $(cat $d/local_cpulist 2>/dev/null) node $(cat $d/numa_node 2>/dev/null); done | head
nproc; taskset -p $$ 
for i in 1 2 3; do examples/icp_slam 60 640 480 | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " free"
NODES=$(ls -d /sys/devices/system/node/node* | wc -l)
for n in $(seq 0 $((NODES-1))); do
  CPUS=$(cat /sys/devices/system/node/node$n/cpulist)
  for i in 1 2 3; do taskset -c $CPUS examples/icp_slam 60 640 480 2>/dev/null | grep -o '"frames_per_s": [0-9.]*' | tr '\n' ' '; done; echo " node $n ($CPUS)"
done
