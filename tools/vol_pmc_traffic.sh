#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the volume kernels alone (gpurun from the repo root): bash tools/vol_pmc_traffic.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/vp
  rocprofv3 --kernel-trace --pmc $c -d $O/vp -o pmc -- python $R/tools/vol_split_time.py > $O/vp_$c.log 2>&1
  DB=$(find $O/vp -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py --pmc $DB | grep -E "corr_vol_split|igemm_kernel|counter " 
done
rm -rf $O/vp
