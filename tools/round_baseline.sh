#!/bin/bash
# Per-round evidence pass (was r05_baseline.sh) (run through gpurun from the repo root): single-pair kernel trace, batch sweep, workspace pipeline rate,
# KeyframeConv rate.  Everything lands under gpurun_out/<tag>_*; the summaries are copied to profiles/ by hand.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_sp -o sp -- python $R/tools/single_pair_trace.py > $O/${TAG}_sp.log 2>&1
tail -1 $O/${TAG}_sp.log
SP=$(find $O/${TAG}_sp -name '*.db' | head -1)
cd $R
python tools/rocpd_summary.py $SP > $O/${TAG}_single_pair_kernel_stats.txt
rm -rf $O/${TAG}_sp
python tools/batch_sweep.py > $O/${TAG}_batch_sweep.json 2> $O/${TAG}_batch_sweep.err; cat $O/${TAG}_batch_sweep.json
python tools/pipeline_rate.py 256 6 > $O/${TAG}_pipeline_rate.json 2> $O/${TAG}_pipeline_rate.err; cat $O/${TAG}_pipeline_rate.json
python tools/pipeline_rate.py 64 6 > $O/${TAG}_pipeline_rate_64.json 2>> $O/${TAG}_pipeline_rate.err; cat $O/${TAG}_pipeline_rate_64.json
python tools/keyframe_conv_rate.py > $O/${TAG}_keyframe_conv_rate.txt 2> $O/${TAG}_keyframe_conv_rate.err; cat $O/${TAG}_keyframe_conv_rate.txt
head -40 $O/${TAG}_single_pair_kernel_stats.txt
