mkdir -p gpurun_out/r06b/soak_long
python tools/gpu_soak_pgs.py 600 101 > gpurun_out/r06b/soak_long/soak_pgs.txt 2>&1; tail -2 gpurun_out/r06b/soak_long/soak_pgs.txt
python tools/gpu_soak_pgs.py 600 102 big > gpurun_out/r06b/soak_long/soak_pgs_big.txt 2>&1; tail -2 gpurun_out/r06b/soak_long/soak_pgs_big.txt
python tools/gpu_soak_pgs.py 400 103 wide > gpurun_out/r06b/soak_long/soak_pgs_wide.txt 2>&1; tail -2 gpurun_out/r06b/soak_long/soak_pgs_wide.txt
python tools/gpu_soak_pgs_api.py 300 104 > gpurun_out/r06b/soak_long/soak_pgs_api.txt 2>&1; tail -1 gpurun_out/r06b/soak_long/soak_pgs_api.txt
python tools/gpu_soak_ekf.py 600 105 both > gpurun_out/r06b/soak_long/soak_ekf.txt 2>&1; tail -1 gpurun_out/r06b/soak_long/soak_ekf.txt
python tools/gpu_soak_adversarial.py 300 106 > gpurun_out/r06b/soak_long/soak_adversarial.txt 2>&1; tail -1 gpurun_out/r06b/soak_long/soak_adversarial.txt
python tools/gpu_soak_api.py 300 107 > gpurun_out/r06b/soak_long/soak_api.txt 2>&1; tail -1 gpurun_out/r06b/soak_long/soak_api.txt
