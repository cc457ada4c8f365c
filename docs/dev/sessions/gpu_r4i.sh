#!/bin/bash
# round 4, session i: the HBM-streamed EKF size class (L_max > 200) against the oracle
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "400_landmarks or any_length or 200_landmarks" 2>&1 | tail -30
