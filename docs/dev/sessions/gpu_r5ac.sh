# round 5, session ac: what the dense Cholesky's panel step waits for - ablations (wrong results, timing only) in an experimental library:
# SLAM_PGS_CHOL_LL = 1 + 16 * a, a & 1: no staging / tile formation by wavefronts 1 .. 15 beside the diagonal block, a & 2: no diagonal factor / inverse in wavefront 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ac
for a in 1 17 33 49; do
  echo "SLAM_PGS_CHOL_LL=$a"
  SLAM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libslam_hip_abl.so SLAM_PGS_CHOL_LL=$a SLAM_PGS_PROF=1 python3 tools/gpu_pgs_phases.py 2>&1 | tail -2
done | tee gpurun_out/r5ac/chol_ablation.txt
