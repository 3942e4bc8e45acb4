#!/bin/bash
# per-kernel durations of the proposal / detection chain, alone on the chip (GPU box, repo root); arg: batch size
root=$PWD; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/chain_ser
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/chain_ser -- python3 $root/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 5 --warmup 2 --streams 1 --no-graphs --batch ${1:-8} > $out/chain_ser.log 2>&1
cd $root
python3 - <<P
import csv,glob,collections,re
f=glob.glob("$out/chain_ser/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if any(k in n for k in ('nms_','rpn_','box_decode','roi_align')):
        short=re.sub(r'\(anonymous namespace\)::','',n).split('(')[0].replace('void ','')
        d[(short, r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()): print("%-28s grid %-6s %-4s %-3s n=%-3d avg %6.1f us  min %6.1f"%(k[0][:28],k[1],k[2],k[3],len(v),sum(v)/len(v),min(v)))
P
