#!/bin/bash
# round 4, run 1: determinism tests + re-posed trajectory test, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_determinism_gpu.py tests/test_bf16_trajectory_gpu.py -q -s -m gpu > gpurun_out/r04_run1_det.txt 2>&1
echo "det rc=$?" | tee -a gpurun_out/r04_run1_det.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_run1_suite.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04_run1_suite.txt
tail -5 gpurun_out/r04_run1_det.txt; tail -15 gpurun_out/r04_run1_suite.txt
