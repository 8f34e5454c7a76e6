#!/bin/bash
# LDS bank-conflict cycles per kernel of the fp32 step (gpurun from the repo root): bash tools/lds_conflicts.sh [tag]
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-single --no-fast --no-volsplit --no-prof --no-handoff --no-verify --no-sweep --no-pipeline --steps 1 --warmup 1"
rm -rf $O/ldsc
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/ldsc -o pmc -- python $R/bench.py $COMMON > $O/ldsc.log 2>&1
DB=$(find $O/ldsc -name '*.db' | head -1)
{ echo "# bench.py $COMMON under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; python $R/tools/pmc_ratio.py $DB SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE; } > $O/${TAG}_lds_conflicts.txt
cut -c1-220 $O/${TAG}_lds_conflicts.txt | head -24
rm -rf $O/ldsc
