# round 5, session ap: what wavefront 1 of the Cholesky spends in the staging and in the tile formation (experimental library with two more timers), full batch and 8 instances
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ap
for bsz in 256 8; do
W1_BATCH=$bsz SLAM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libslam_hip_w1.so python3 tools/gpu_pgs_chol_w1.py 2>&1 | tail -2
done | tee gpurun_out/r5ap/chol_w1.txt
