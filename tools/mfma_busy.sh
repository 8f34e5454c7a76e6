#!/bin/bash
# Matrix-pipe occupancy and clock of the step's kernels, per arithmetic (gpurun from the repo root): bash tools/mfma_busy.sh r06
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-single --no-fast --no-volsplit --no-prof --no-handoff --no-verify --no-sweep --no-pipeline --steps 1 --warmup 1"
for prec in fp32 bf16x6 bf16x3; do
  rm -rf $O/mb
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mb -o pmc -- python $R/bench.py $COMMON --precision $prec > $O/mb_$prec.log 2>&1
  DB=$(find $O/mb -name '*.db' | head -1)
  { echo "# $prec: bench.py $COMMON --precision $prec under rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; python $R/tools/mfma_busy.py $DB; } > $O/${TAG}_mfma_busy_$prec.txt
  head -14 $O/${TAG}_mfma_busy_$prec.txt | cut -c1-140; tail -2 $O/${TAG}_mfma_busy_$prec.txt
done
rm -rf $O/mb
