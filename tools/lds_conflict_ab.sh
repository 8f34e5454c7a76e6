R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in base nobswz; do
  if [ $v = nobswz ]; then export OFX_LIB_PATH=$R/tools/variants/libofx_nobswz.so; fi
  SP_N=3 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/ldsc_$v -o l -f csv -- python $R/tools/single_pair_trace.py 2>&1 | grep "single pair"
  F=$(find $R/gpurun_out/ldsc_$v -name "*counter_collection.csv" | head -1)
  python - $F <<EOF
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:100]+" g"+r["Grid_Size"]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_LDS_IDX_ACTIVE",0))[:9]:
    a=v.get("SQ_LDS_IDX_ACTIVE",0); c=v.get("SQ_LDS_BANK_CONFLICT",0)
    print(f"{c/max(a,1):.4f}  {a:.3g}  {k}")
EOF
  rm -rf $R/gpurun_out/ldsc_$v
  python $R/tools/single_pair_time.py 2>&1 | grep "B="
done
