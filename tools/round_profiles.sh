#!/bin/bash
# Every profiles/<tag>_* file of a round from ONE tree in ONE call (run on the GPU box; copy gpurun_out/<tag>_* into profiles/ afterwards).
#   bash tools/round_profiles.sh r06 [quick]
# Sections: bench lines (4 workloads), kernel stats of the FV2P step / its boundary leg / MGAF (rocprofv3 --kernel-trace --stats), the per-op
# tables of tools/microbench.py, the roofline kernel's counters (separate --pmc passes, counters only with --kernel-trace), the in-situ
# launch times of the roofline kernel, the one-rank DDP stream-budget runs, the step-noise calibration.  profiles/README.md quotes only
# numbers found in these files.
TAG=${1:-r06}; QUICK=$2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
make -s -C tools/ubench || exit 1
echo "== tree $(cat .git/HEAD 2>/dev/null) $(date -u +%FT%TZ)" > $O/${TAG}_manifest.txt
sha256sum from-voxel-to-point_amd/lib/libfv2p_ops.so bench.py >> $O/${TAG}_manifest.txt

echo "== bench lines"
for w in fv2p mgaf fv2p-waymo backbone; do
  timeout 900 python3 bench.py --workload $w > $O/${TAG}_bench_${w//-/_}.json 2> $O/bench_$w.err || echo "bench $w FAILED"
  tail -c 600 $O/${TAG}_bench_${w//-/_}.json | head -c 400; echo
done

echo "== kernel stats"
prof() {   # <name> <steps in the table> <bench args...>
  local name=$1 n=$2; shift 2
  rm -rf $O/prof_$name
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o s -- python3 bench.py --watchdog 0 "$@" > $O/prof_$name.log 2>&1
  local f=$(find $O/prof_$name -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/${TAG}_${name}_kernel_stats.csv && python3 tools/kstats.py $O/${TAG}_${name}_kernel_stats.csv $n 14
  if [ "$name" = fv2p ]; then
    local t=$(find $O/prof_$name -name '*kernel_trace.csv' | head -1)
    python3 tools/insitu_conv.py "$t" > $O/${TAG}_insitu_conv.txt; cat $O/${TAG}_insitu_conv.txt
    python3 tools/gaps.py "$t" fps_wave_k 30 > $O/${TAG}_fv2p_idle_gaps.txt 2>&1
  fi
  rm -rf $O/prof_$name
}
prof fv2p 25 --steps 20 --warmup 5 --cpu-clouds 0 --no-roofline --inline-steps 0 --refstyle-steps 0
FV2P_BENCH_LEG=boundary prof fv2p_boundary 20 --steps 2 --warmup 2 --inline-steps 20 --refstyle-steps 0 --cpu-clouds 0 --no-roofline
if [ -z "$QUICK" ]; then
  prof mgaf 13 --workload mgaf --steps 10 --warmup 3 --cpu-clouds 0 --no-roofline --refstyle-steps 0
  prof fv2p_waymo 14 --workload fv2p-waymo --steps 10 --warmup 4 --cpu-clouds 0 --no-roofline --inline-steps 0
fi

echo "== per-op tables"
FV2P_RES=1 python3 tools/microbench.py conv > $O/${TAG}_microbench_conv_kitti.txt 2>&1; tail -32 $O/${TAG}_microbench_conv_kitti.txt
python3 tools/microbench.py oproof > $O/${TAG}_op_roofline.txt 2>&1; tail -25 $O/${TAG}_op_roofline.txt
python3 tools/microbench.py fps > $O/${TAG}_microbench_fps.txt 2>&1; tail -12 $O/${TAG}_microbench_fps.txt
python3 tools/microbench.py dcn > $O/${TAG}_microbench_dcn.txt 2>&1
for shape in "4 128 200 176 1" "4 256 100 88 1" "4 256 50 44 1" "4 256 200 176 4"; do tools/ubench/dcn_bench $shape >> $O/${TAG}_microbench_dcn.txt 2>&1; done   # the C-ABI calls alone
tail -12 $O/${TAG}_microbench_dcn.txt
if [ -z "$QUICK" ]; then
  FV2P_WAYMO=1 python3 tools/microbench.py conv > $O/${TAG}_microbench_conv_waymo.txt 2>&1
  python3 tools/microbench.py nn > $O/${TAG}_microbench_nn.txt 2>&1
  python3 tools/microbench.py sa > $O/${TAG}_microbench_sa.txt 2>&1
fi

echo "== round 6: what the BatchNorm statistics cost a conv launch, one-launch BatchNorm passes, the completion counter, the sampler's walk over its touched buckets"
python3 tools/fin_time.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_fin_time.txt; cat $O/${TAG}_fin_time.txt
python3 tools/bn_time.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bn_time.txt; tail -8 $O/${TAG}_bn_time.txt
tools/ubench/atomic_rate > $O/${TAG}_atomic_rate.txt 2>&1; tail -17 $O/${TAG}_atomic_rate.txt
{ for t in 0 1 2; do echo "== development library, FV2P_FPS_IDX=$t (0: straight-line scan of compile-time slots, rounds 3-5; 1: run-time register indices up to 32 slots, the default; 2: at every slot count)"; FV2P_LIB_DIR=$PWD/from-voxel-to-point_amd/lib/dev FV2P_FPS_IDX=$t python3 tools/microbench.py fps 2>&1 | grep "^FPS" | head -5; done; } > $O/${TAG}_fps_idx.txt; cat $O/${TAG}_fps_idx.txt
python3 tools/microbench.py fpstrace 2>&1 | grep -v amdgpu.ids > $O/${TAG}_fpstrace.txt; grep -A10 "n = 16384" $O/${TAG}_fpstrace.txt | tail -2
for f in 1 0; do echo "FV2P_BN_FOLD=$f (1: residual blocks on conv_fin / bn_apply, 0: the round-5 arrangement): $(FV2P_BN_FOLD=$f python3 bench.py --workload backbone --backbone res8x --cpu-clouds 0 --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done > $O/${TAG}_res8x_fold.txt; cat $O/${TAG}_res8x_fold.txt

echo "== the isolated roofline probe inside the bench command, as the kernel trace has it"
rm -rf $O/prof_probe
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -o s -- python3 bench.py --watchdog 0 --steps 5 --warmup 5 --cpu-clouds 0 --inline-steps 0 --refstyle-steps 0 > $O/prof_probe.log 2>&1
{ echo "rocprofv3 --kernel-trace --stats -- python3 bench.py --watchdog 0 --steps 5 --warmup 5 --cpu-clouds 0 --inline-steps 0 --refstyle-steps 0 (roofline probe ON)"
  python3 tools/probe_from_trace.py "$(find $O/prof_probe -name '*kernel_trace.csv' | head -1)"
  python3 -c "import json,sys; d=json.loads([l for l in open('$O/prof_probe.log') if l.startswith('{\"metric')][-1]); r=d['roofline']; print('bench line of the same run: ms_per_step', d['ms_per_step'], '(under the profiler), roofline.avg_kernel_us', r['avg_kernel_us'], 'roofline.frac', r['frac'], 'in_step_us', r.get('in_step_us'))"; } > $O/${TAG}_roofline_probe_trace.txt 2>&1
cat $O/${TAG}_roofline_probe_trace.txt; rm -rf $O/prof_probe

echo "== counters of the roofline kernel"
bash tools/pmc_roofline.sh $TAG > $O/pmc_roofline.log 2>&1; tail -2 $O/pmc_roofline.log
bash tools/pmc_mfma.sh $TAG > $O/pmc_mfma.log 2>&1; tail -2 $O/pmc_mfma.log

echo "== one-rank DDP: RCCL's stream (and a stand-in for its traffic) beside the step's four streams"
for mode in plain solo standin; do   # bench.py defaults: flat gradient sync, six hardware queues for a DDP-like run (numbers taken by pattern: a c10d warning may sit inside the JSON line)
  case $mode in plain) E="";; solo) E="FV2P_DDP_SOLO=1";; standin) E="FV2P_DDP_SOLO=1 FV2P_DDP_COMM_STANDIN=1";; esac
  env $E timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-clouds 0 --no-roofline --inline-steps 0 --refstyle-steps 0 > $O/ddp_$mode.json 2> $O/ddp_$mode.err
  echo "{\"mode\": \"$mode\", \"env\": \"$E\", $(grep -o '"ms_per_step": [0-9.]*' $O/ddp_$mode.json | head -1), $(grep -o '"value": [0-9.]*' $O/ddp_$mode.json | head -1)}"
done > $O/${TAG}_ddp_stream_budget.jsonl
cat $O/${TAG}_ddp_stream_budget.jsonl

if [ -z "$QUICK" ]; then
  echo "== step-noise calibration (tests/f64_calibration.py)"
  python3 tools/step_noise.py 8 > $O/${TAG}_step_noise.txt 2>&1; tail -3 $O/${TAG}_step_noise.txt
  python3 tools/noise_locate.py 4 > $O/${TAG}_noise_locate.txt 2>&1
fi
ls -la $O/${TAG}_* | awk '{print $5, $9}'
