#!/bin/bash
# usage (on the GPU box, repo root): tools/profile_round.sh r2   -> gpurun_out/<tag>_* (copy what is to be judged into profiles/)
tag=${1:-r2}
root=$PWD
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_ser -- $B --steps 5 --warmup 2 --streams 1 --no-graphs > $out/${tag}_prof_ser.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof_def -- $B --steps 5 --warmup 2 > $out/${tag}_prof_def.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- $B --steps 3 --warmup 1 --no-graphs --streams 1 --no-roofline > $out/${tag}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- $B --steps 3 --warmup 1 --no-graphs --streams 1 --no-roofline > $out/${tag}_pmc_write.log 2>&1
cd $root
cp $(ls $out/${tag}_prof_ser/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_streams1_nographs.csv
cp $(ls $out/${tag}_prof_def/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_default.csv
python3 tools/pmc_summary.py $(ls $out/${tag}_pmc_fetch/*/*counter_collection.csv | head -1) $(ls $out/${tag}_pmc_write/*/*counter_collection.csv | head -1) > $out/${tag}_hbm_traffic.json
python3 tools/prof_layers.py > $out/${tag}_layers.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 300 $out/${tag}_bench.json
