#!/bin/bash
# rocprofv3 --kernel-trace --stats over the timed steps of one bench workload -> gpurun_out/<workload>_kernel_stats.csv
#   bash tools/prof_step.sh <workload> <steps>
W=$1; K=${2:-25}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
D=gpurun_out/prof_step_$W
rm -rf $D
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o s -- python3 bench.py --workload $W --steps $K --warmup 5 --inline-steps 0 --refstyle-steps 0 --cpu-clouds 0 --no-roofline > $D.log 2>&1
cp $D/s_kernel_stats.csv gpurun_out/${W}_kernel_stats.csv
python3 tools/kstats.py gpurun_out/${W}_kernel_stats.csv $((K + 5)) 12
rm -rf $D
