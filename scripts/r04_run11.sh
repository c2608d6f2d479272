#!/bin/bash
# round 4: where the 9 minutes of the GPU suite go (--durations), on the final tree (one more green run on a fresh box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=40 > gpurun_out/r04_final_suite_durations.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04_final_suite_durations.txt
grep -E "passed|failed" gpurun_out/r04_final_suite_durations.txt | tail -2
grep -A45 "slowest" gpurun_out/r04_final_suite_durations.txt | cut -c1-160
