#!/bin/bash
# Developer tool (GPU box): A/B of two BUILDS of the library on the same box, interleaved (boxes differ by 10 % in clocks).
#   bash tools/ab_libs.sh ab_libs/old.so ab_libs/new.so [rounds]
cd $GRAFT_REPO_ROOT
R=${3:-3}
for r in $(seq $R); do
  for lib in "$1" "$2"; do
    cp "$lib" sig_sdp_mmw_amd/libmmw_hip.so
    echo "== [$lib]"
    bash tools/ab.sh "A=1" | tail -1
  done
done
