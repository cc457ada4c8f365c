# round 5, session k: EKF control wavefront - the next timestep's command requested a step ahead (A/B, same box, alternating)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
for rep in 1 2 3; do
  bash tools/gpu_ab_ekf.sh tools/lib_base.so:0 tools/lib_cmdpre.so:0 >> gpurun_out/r5k/cmdpre_ab.txt 2>&1
done
cat gpurun_out/r5k/cmdpre_ab.txt
