#!/bin/bash
# Stream budget of the multi-rank arrangement, measured on ONE GPU (no 8-GPU node is available to the builder): the FV2P step under a one-rank
# DistributedDataParallel over RCCL (FV2P_DDP_SOLO=1: DDP's hooks, bucket views and RCCL's call per bucket) plus a stand-in for the traffic of
# a real all-reduce on a communication stream of its own (FV2P_DDP_COMM_STANDIN=1: two device copies of every 25 MB bucket), for several
# stream arrangements and hardware-queue counts.  -> gpurun_out/<tag>_ddp_stream_matrix.txt
TAG=${1:-r05}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ddp_stream_matrix.txt
: > $OUT
run() { local tag=$1; shift; local envs=$1; shift
  env $envs timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-clouds 0 --no-roofline --inline-steps 0 --refstyle-steps 0 "$@" > gpurun_out/ddpm_$tag.json 2>gpurun_out/ddpm_$tag.err
  echo "$tag | $envs $* | $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ddpm_$tag.json | head -1)" | tee -a $OUT; }
S="FV2P_DDP_SOLO=1 FV2P_DDP_COMM_STANDIN=1"
W1="FV2P_WGRAD_OVERLAP=1"; W0="FV2P_WGRAD_OVERLAP=0"     # weight gradients on their own stream / on the calling stream (bench.py's default for FV2P)
Q4="GPU_MAX_HW_QUEUES=4"                                  # (bench.py itself sets 6 for DDP runs unless the variable is given)
run plain_wgradstream "$W1 $Q4"
run plain "$W0 $Q4"
run plain_hwq6 "$W0 GPU_MAX_HW_QUEUES=6"
run plain_hwq8 "$W0 GPU_MAX_HW_QUEUES=8"
run solo "FV2P_DDP_SOLO=1 $W0 $Q4" --grad-sync ddp
run solo_hwq6 "FV2P_DDP_SOLO=1 $W0 GPU_MAX_HW_QUEUES=6" --grad-sync ddp
run standin_wgradstream_hwq4 "$S $W1 $Q4" --grad-sync ddp
run standin_wgradstream_hwq5 "$S $W1 GPU_MAX_HW_QUEUES=5" --grad-sync ddp
run standin_wgradstream_hwq6 "$S $W1 GPU_MAX_HW_QUEUES=6" --grad-sync ddp
run standin_wgradstream_hwq8 "$S $W1 GPU_MAX_HW_QUEUES=8" --grad-sync ddp
run standin_hwq4 "$S $W0 $Q4" --grad-sync ddp
run standin_hwq5 "$S $W0 GPU_MAX_HW_QUEUES=5" --grad-sync ddp
run standin_hwq6 "$S $W0 GPU_MAX_HW_QUEUES=6" --grad-sync ddp
run standin_hwq8 "$S $W0 GPU_MAX_HW_QUEUES=8" --grad-sync ddp
run standin_hwq4_pointstream "$S $W0 $Q4" --dense-stream 0 --grad-sync ddp
run standin_hwq4_nobranchstream "$S $W0 $Q4" --dense-stream 0 --point-stream 0 --grad-sync ddp
run standin_hwq4_nofpsahead "$S $W0 $Q4" --fps-ahead 0 --grad-sync ddp
run standin_default_ddp "$S" --grad-sync ddp
run flat_solo "FV2P_DDP_SOLO=1" --grad-sync flat
run flat_standin_hwq4 "$S $Q4" --grad-sync flat
run flat_standin_hwq6 "$S GPU_MAX_HW_QUEUES=6" --grad-sync flat
run flat_standin_hwq8 "$S GPU_MAX_HW_QUEUES=8" --grad-sync flat
run flat_standin_default "$S"
