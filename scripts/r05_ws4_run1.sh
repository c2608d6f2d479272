#!/bin/bash
# round 5: conv_ws4_kernel (two four-wave workgroups per CU) against the 256x256 persistent kernels (parity) and against round 4's conv_ws_kernel
# (bit identity of the outputs, A/B times), race screen, start-delay sweep.  Usage: gpurun -- bash scripts/r05_ws4_run1.sh
O=gpurun_out/r05_ws4_run1.txt; : > $O
echo "== ws4 vs the persistent 256x256 kernels (parity), race screen of 20 launches ==" >> $O
WB_RACE=20 WB_REPS=10 timeout 600 ./build/ws_bench >> $O 2>&1
echo "== ws4 vs conv_ws_kernel (round 4): bit identity + A/B, two repetitions ==" >> $O
for r in 1 2; do WB_REF_WS=1 timeout 600 ./build/ws_bench >> $O 2>&1; done
echo "== start delay of the second workgroup (cycles), l3 256->1024 and l1 64->256 and l2 128->512 ==" >> $O
for st in 0 768 1536 3072 6144; do
  for sh in 0 1 2; do
    echo "-- stagger $st shape $sh" >> $O
    WB_REF_WS=1 WB_STAGGER=$st WB_ONLY=$sh timeout 300 ./build/ws_bench >> $O 2>&1
  done
done
grep -c MISMATCH $O
tail -5 $O
