# round 5, session ag: phase table of the EKF kernel for a 20-step launch against an 80-step launch from the same timestep (which phases a short launch inflates)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ag
PHASE_STEPS=20 python3 tools/gpu_phases.py f64 > gpurun_out/r5ag/phases_K20.txt 2>&1
PHASE_STEPS=80 python3 tools/gpu_phases.py f64 > gpurun_out/r5ag/phases_K80.txt 2>&1
paste gpurun_out/r5ag/phases_K20.txt gpurun_out/r5ag/phases_K80.txt | cut -c1-200
