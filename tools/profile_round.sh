#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01c
# 1) kernel trace + stats of the default bench workload, 2) + 3) FETCH_SIZE / WRITE_SIZE PMC passes (separate
# runs, kernel-trace only), then the text / JSON summaries under gpurun_out/<tag>_*.
set -u
TAG=${1:-r01c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-single --no-fast --no-prof --no-handoff --no-verify --no-sweep --no-pipeline"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o bench -- python $R/bench.py --steps 2 --warmup 1 $COMMON > $O/${TAG}_trace.log 2>&1
tail -1 $O/${TAG}_trace.log | cut -c1-300
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o pmc -- python $R/bench.py --steps 1 --warmup 0 $COMMON > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o pmc -- python $R/bench.py --steps 1 --warmup 0 $COMMON > $O/${TAG}_pmc_write.log 2>&1
TR=$(find $O/${TAG}_trace -name '*.db' | head -1)
PF=$(find $O/${TAG}_pmc_fetch -name '*.db' | head -1)
PW=$(find $O/${TAG}_pmc_write -name '*.db' | head -1)
cd $R
python tools/rocpd_summary.py $TR > $O/${TAG}_bench_b64_kernel_stats.txt
python tools/rocpd_summary.py --pmc $PF > $O/${TAG}_bench_b64_pmc_fetch.txt
python tools/rocpd_summary.py --pmc $PW > $O/${TAG}_bench_b64_pmc_write.txt
python tools/pmc_traffic.py $PF $PW $TAG > $O/${TAG}_pmc_traffic_b64.json
rm -rf $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
head -12 $O/${TAG}_bench_b64_kernel_stats.txt
cat $O/${TAG}_pmc_traffic_b64.json
