# round 5, session ak: phase timers of the UKF step kernel at L = 20 (the PROF instantiation)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ak
python3 tools/gpu_ukf_phases.py 20 8 > gpurun_out/r5ak/step_phases_L20.txt 2>&1; cat gpurun_out/r5ak/step_phases_L20.txt | head -30
