python scratch/two.py 2 0 999 256 2>&1 | tail -1
python scratch/two.py 4 0 999 256 2>&1 | tail -1
python scratch/two.py 3 0 999 256 2>&1 | tail -1
python scratch/two.py 4 0 999 128 2>&1 | tail -1
python scratch/two.py 4 0 999 512 2>&1 | tail -1
