#!/bin/bash
# Developer tool (GPU box, from the repo root): rocprofv3 kernel statistics of one bench.py command.
#   bash tools/profile_bench.sh <tag> [bench.py arguments...]   ->  gpurun_out/<tag>_kernel_stats.csv + gpurun_out/<tag>.json
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$TAG.json 2> $OUT/err.log
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
head -25 $f | cut -c1-160
rm -rf $OUT  # raw traces are large: only the summaries above travel back
