mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
bash tools/profile_round.sh r5 > gpurun_out/r5_profile_round.log 2>&1
echo "profile_round done"; tail -c 300 gpurun_out/r5_bench.json; echo
python3 tools/prof_layers.py bf16 1 > gpurun_out/r5_layers_batch1.txt 2>&1
export LAYERS="res4 conv1"
PMC_MAX=6 bash tools/pmc_passes.sh r5_pws_res4conv1 conv1x1_pws_kernel -- python3 $R/tools/pws_micro.py 8 > gpurun_out/r5_pmc_pws.log 2>&1
python3 - <<'PY' > gpurun_out/r5_sq_counters_ring_res4conv1.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_r5_pws_res4conv1/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_ring_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]; print("%-45s n=%3d avg=%.6g" % (k, len(v), sum(v) / len(v)))
PY
cp gpurun_out/pmc_r5_pws_res4conv1.txt gpurun_out/r5_sq_counters_pws_res4conv1.txt
unset LAYERS
timeout 300 python3 tools/pws_micro.py 8 > gpurun_out/r5_pws_micro_b8.txt 2>&1
timeout 300 python3 tools/pws_micro.py 1 > gpurun_out/r5_pws_micro_b1.txt 2>&1
bash tools/other_configs.sh > gpurun_out/r5_other_configs.txt 2>&1
STEPS=40 bash tools/ab_modes.sh "DP_CONV_PWS=1 DP_CONV_PWS=0 DP_FUSE_PAIR=0 DP_GROUP_DECONV=0 DP_CONV_PWS=1 DP_CONV_PWS=0 DP_FUSE_PAIR=0 DP_GROUP_DECONV=0" > gpurun_out/r5_ab_round5.txt 2>&1
cat gpurun_out/r5_ab_round5.txt
# keep what is merged back small: the raw rocprofv3 output directories stay on the box
rm -rf gpurun_out/r5_prof_ser gpurun_out/r5_prof_def gpurun_out/r5_pmc_fetch gpurun_out/r5_pmc_write gpurun_out/pmc_r5_pws_res4conv1
ls -la gpurun_out | head -40
