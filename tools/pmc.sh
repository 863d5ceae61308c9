#!/bin/bash
# usage: tools_pmc.sh <outdir> <bench args...>   -- collects a few PMC passes for the LM kernel (own run each)
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $out/$name -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --check 0 --extras 0 "$@" > /dev/null 2> $out/$name.err
  f=$(find $out/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if "lm_solve" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, "%.4g"%(sum(v)/len(v)))
PY
done | tee $out/summary.txt
