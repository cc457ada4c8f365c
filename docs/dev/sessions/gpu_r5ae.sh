# round 5, session ae: A/B on one box - Cholesky with the first A batch of the next panel's tiles requested before the half-phase barrier (156 B of scratch instead of 56)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in head pf2; do
SLAM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libslam_hip_$v.so python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v B256:', round(d['value'],1), d['config'].get('kernel_ms_per_solve')['chol'])"
done
done
for v in head pf2; do
SLAM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libslam_hip_$v.so SLAM_PGS_PROF=1 python3 tools/gpu_pgs_phases.py 2>&1 | tail -2 | head -1 | cut -c1-250
done
