#!/bin/bash
# Builds the timing ablations of conv_ws_kernel (scripts/ws_bench.hip) - run in the build container; the binaries travel in build/.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
F="-O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-result -w"
for v in "ws_bench:" "ws_pp:-DWS_PP=1" "ws_pp_nostore:-DWS_PP=1 -DWS_ABL_NOSTORE" "ws_pp_nomfma:-DWS_PP=1 -DWS_ABL_NOMFMA" "ws_nt:-DWS_STORE_AUX=2" "ws_pp_nt:-DWS_PP=1 -DWS_STORE_AUX=2" "ws_pxcd:-DWS_ABL_PANEL_XCD" "ws_nostore:-DWS_ABL_NOSTORE" "ws_nomfma:-DWS_ABL_NOMFMA" "ws_nodma:-DWS_ABL_NODMA"; do
  /opt/rocm/bin/hipcc $F ${v#*:} scripts/ws_bench.hip -o build/${v%%:*} &
done
wait
ls -la build/ws_*
