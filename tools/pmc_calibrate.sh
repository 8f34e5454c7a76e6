#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per access width on this box (gpurun from the repo root): bash tools/pmc_calibrate.sh r02
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/pmc_calib $R/tools/pmc_calib.hip || exit 1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_cal_trace -o cal -- /tmp/pmc_calib > $O/${TAG}_cal.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_cal_fetch -o pmc -- /tmp/pmc_calib >> $O/${TAG}_cal.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_cal_write -o pmc -- /tmp/pmc_calib >> $O/${TAG}_cal.log 2>&1
cd $R
{
  echo "# access-width calibration of FETCH_SIZE / WRITE_SIZE (tools/pmc_calib.hip: 1 GiB per launch, known bytes)"
  python tools/rocpd_summary.py $(find $O/${TAG}_cal_trace -name '*.db' | head -1) | head -16
  python tools/rocpd_summary.py --pmc $(find $O/${TAG}_cal_fetch -name '*.db' | head -1)
  python tools/rocpd_summary.py --pmc $(find $O/${TAG}_cal_write -name '*.db' | head -1)
} > $O/${TAG}_pmc_calibration.txt
rm -rf $O/${TAG}_cal_trace $O/${TAG}_cal_fetch $O/${TAG}_cal_write
cat $O/${TAG}_pmc_calibration.txt
