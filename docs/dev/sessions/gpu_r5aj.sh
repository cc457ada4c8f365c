# round 5, session aj: long soaks of the final tree (other seeds than the artefact session's)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5aj; mkdir -p $OUT
timeout 400 python3 tools/gpu_soak_ekf.py 300 9301 both > $OUT/soak_ekf.txt 2>&1
timeout 300 python3 tools/gpu_soak_api.py 200 9302 > $OUT/soak_api.txt 2>&1
timeout 400 python3 tools/gpu_soak_adversarial.py 300 9303 both > $OUT/soak_adversarial.txt 2>&1
timeout 400 python3 tools/gpu_soak_pgs.py 300 9304 > $OUT/soak_pgs.txt 2>&1
timeout 300 python3 tools/gpu_soak_pgs.py 200 9305 big > $OUT/soak_pgs_big.txt 2>&1
timeout 250 python3 tools/gpu_soak_pgs_api.py 150 9306 > $OUT/soak_pgs_api.txt 2>&1
for f in soak_ekf soak_api soak_adversarial soak_pgs soak_pgs_big soak_pgs_api; do echo "$f: $(tail -n 1 $OUT/$f.txt | cut -c 1-300)"; done
