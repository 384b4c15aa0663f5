cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp FV2P_BENCH_LEG=boundary
rm -rf gpurun_out/prof_b
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -o b -- python3 bench.py --steps 2 --warmup 2 --inline-steps 20 --refstyle-steps 0 > gpurun_out/prof_b.log 2>&1
tail -3 gpurun_out/prof_b.log
f=$(find gpurun_out/prof_b -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/boundary_kernel_stats.csv
find gpurun_out/prof_b -name '*kernel_trace.csv' -size +60M -delete
find gpurun_out/prof_b -name '*.db' -delete
ls -la gpurun_out/prof_b/* | head
