#!/bin/bash
# round 4: the full-size step twice, bit for bit (test tightened from 2e-3 / 2e-2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu > gpurun_out/r04_full_size_bits.txt 2>&1
echo "rc=$?" | tee -a gpurun_out/r04_full_size_bits.txt
tail -15 gpurun_out/r04_full_size_bits.txt | cut -c1-220
