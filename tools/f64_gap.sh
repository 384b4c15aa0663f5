#!/bin/bash
# bisect of the float64 gap: one process per setting
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for s in "" "FV2P_FUSED_BN=0" "FV2P_BN_EPILOGUE=0" "FV2P_CONV_KSPLIT=0" "FV2P_CONV_IMPL=dense" "FV2P_CONV_PLAN=0" "FV2P_DEFER_WGRAD=0" "FV2P_WGRAD_OVERLAP=0" "FV2P_FUSED_BN=0 FV2P_CONV_KSPLIT=0"; do
  env $s python3 tools/f64_gap.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids"
done
