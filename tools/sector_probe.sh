#!/bin/bash
# HBM read granularity on this part (gpurun from the repo root): bash tools/sector_probe.sh -> gpurun_out/r06_sector_probe.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/sector_probe $R/tools/sector_probe.hip || exit 1
{
  echo "# tools/sector_probe.hip: a 4 GiB span walked line by line, only the first 128 / 64 / 32 bytes of every 128-byte line read (16-byte lane loads, non-temporal)"
  /tmp/sector_probe
  echo
  echo "# counters available for the L2's memory-side read requests:"
  rocprofv3 -L 2>/dev/null | grep -iE "TCC_EA0?_RDREQ|TCC_BUBBLE|FETCH_SIZE" | head -20
} > $O/r06_sector_probe.txt 2>&1
for c in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf $O/sp_pmc
  rocprofv3 --kernel-trace --pmc $c -d $O/sp_pmc -o pmc -- /tmp/sector_probe > /dev/null 2>>$O/r06_sector_probe.err
  DB=$(find $O/sp_pmc -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py --pmc $DB | grep -E "part_line|counter " >> $O/r06_sector_probe.txt
done
rm -rf $O/sp_pmc
cat $O/r06_sector_probe.txt
