for v in "x=1" "slots=0 list=1"; do
  echo "=== $v"; timeout 120 python scratch/repro.py $v 2>&1 | grep -v amdgpu.ids | tail -2
done
python tools/gpu_soak_pgs.py 200 11 wide 2>&1 | tail -3
python tools/gpu_soak_pgs.py 100 13 wide 2>&1 | tail -3
