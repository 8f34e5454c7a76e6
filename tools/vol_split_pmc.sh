#!/bin/bash
# SQ counters of the split volume GEMM (run through gpurun from the repo root): where the waves' cycles go with and without the stores.
#   bash tools/vol_split_pmc.sh  -> gpurun_out/volsplit_pmc_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 0 1; do
  rm -rf $O/vs_pmc
  OFX_VOLSPLIT_DBG=$d rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR -d $O/vs_pmc -o pmc -- python $R/tools/vol_split_time.py > $O/vs_pmc_$d.log 2>&1
  DB=$(find $O/vs_pmc -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py --pmc $DB | grep -E "corr_vol_split|kernel|counter" > $O/volsplit_pmc_dbg$d.txt
  tail -1 $O/vs_pmc_$d.log
  cat $O/volsplit_pmc_dbg$d.txt
done
rm -rf $O/vs_pmc
