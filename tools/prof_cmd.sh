#!/bin/bash
# rocprofv3 --kernel-trace --stats over a python command; prints the kernel_stats rows matching a pattern.
#   bash tools/prof_cmd.sh "<egrep pattern>" <script.py> [args]
PAT=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
D=gpurun_out/prof_cmd
rm -rf $D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 "$@" > $D.log 2>&1
python3 tools/kstats_csv.py $D/p_kernel_stats.csv | egrep -i "$PAT"
rm -rf $D
