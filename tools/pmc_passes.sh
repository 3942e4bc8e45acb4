#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <kernel-name-substring> -- <program and args>   (run on the GPU box from the repo root)
# One rocprofv3 --pmc pass per counter group (separate passes: TCC/SQ slots), per-kernel averages printed and saved as
# gpurun_out/pmc_<tag>.txt. Never combined with hip/hsa trace domains (the pool refuses that).
tag=$1; kern=$2; shift 3
root=$PWD
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum"
 "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum"
)
i=0
for g in "${groups[@]}"; do
  if [ -n "$PMC_MAX" ] && [ $i -ge $PMC_MAX ]; then break; fi
  timeout 150 rocprofv3 --kernel-trace --pmc $g -d $out/p$i --output-format csv -- "$@" > $out/p$i.log 2>&1
  i=$((i+1))
done
cd $root
python3 - "$out" "$kern" <<'PY' | tee gpurun_out/pmc_$tag.txt
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print("%-45s n=%3d avg=%.6g" % (k, len(v), sum(v) / len(v)))
PY
