#!/bin/bash
# Developer tool (GPU box): A/B of several BUILDS of the library on the same box, interleaved (boxes differ by 5 % in clocks).
#   bash tools/ab_libs.sh <rounds> ab_libs/old.so ab_libs/new.so [...]
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    cp "$lib" sig_sdp_mmw_amd/libmmw_hip.so
    echo "== [$lib]"
    bash tools/ab.sh "A=1" | tail -1
  done
done
