#!/bin/bash
# resource usage (VGPRs, spills, LDS) of one EKF step-kernel variant: tools/ekf_res.sh NMAX W KG UNR F32 PIPE [extra flags]
cd "$(dirname "$0")/../live_ekf_slam_amd/csrc" || exit 1
a=("$@"); extra=("${a[@]:6}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -x hip -mllvm -disable-machine-licm \
  -Rpass-analysis=kernel-resource-usage -DV_NMAX=$1 -DV_W=$2 -DV_KG=$3 -DV_UNR=$4 -DV_F32=$5 -DV_PIPE=$6 "${extra[@]}" -c ekf_inst.hip -o /tmp/ekf_res_$$.o 2>&1 \
  | grep -E "error|Function Name|VGPRs:|VGPRs Spill|LDS Size" | sed 's/.*remark: *//;s/\[-Rpass.*//;s/Function Name: _ZN4slam15ekf_step_kernelI//' | tr '\n' ' '
echo; rm -f /tmp/ekf_res_$$.o
