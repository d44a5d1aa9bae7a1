# per-kernel durations + wave-level wait counters of the stand-alone attention kernels (GPU box) -> gpurun_out/attn/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for v in ${VARIANTS-0 3}; do
  export VBX_ATTN_BWD_DMA=$v
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/st$v -o run -- python $R/tools/attn_bench.py 20 $COLD > $O/st$v.log 2>&1
  (cd $R; python tools/prof_summary.py $(find $O/st$v -name "*.db" | head -1) 1 | grep -i attn > $O/stats_v$v.txt; cat $O/stats_v$v.txt)
  rm -rf $O/st$v
done
if [ -n "$PMC" ]; then
  rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $O/sq_counters.txt
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA"; do
    n=$(echo $set | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$n -o run -- python $R/tools/attn_bench.py 3 $COLD > $O/pmc_$n.log 2>&1
    python - $O/pmc_$n <<PY
import csv,glob,sys,collections
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][:40]
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,d in agg.items():
    print(k, {c: round(v/cnt[(k,c)]) for c,v in d.items()})
PY
    rm -rf $O/pmc_$n
  done
fi
