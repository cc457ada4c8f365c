# round 5, session ah: adversarial soak extended with UKF_LOC (random maps, ids beyond the map, messages longer than the class holds) and the device-buffer entry
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ah
timeout 500 python3 tools/gpu_soak_adversarial.py 400 71 both > gpurun_out/r5ah/soak_adv.txt 2>&1; tail -5 gpurun_out/r5ah/soak_adv.txt | cut -c1-400
