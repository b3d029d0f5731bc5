#!/bin/bash
# Evidence pass of round 4 (GPU box): bench line (eager + hipGraph), rocprofv3 kernel stats (serial + overlapped), step
# breakdown, HBM traffic PMC passes, per-layer conv table, SQ counters of the halo-tile kernels, K1 rates + ceiling.
# usage: collect_r04.sh a | b     (two gpurun calls)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
STAGE=${1:-a}
if [ "$STAGE" = a ]; then
python3 $R/bench.py > $O/r04_bench_b16.json 2> $O/bench.err || exit 1
rm -rf $O/serial $O/overlap $O/fetch $O/write
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_graph > $O/serial.log 2>&1 || exit 2
python3 $R/tools/step_breakdown.py $O/serial 70 > $O/r04_step_breakdown_serial.txt
cp $(ls $O/serial/*/*kernel_stats.csv | head -1) $O/r04_bench_b16_kernel_stats_serial.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_graph > $O/overlap.log 2>&1 || exit 3
cp $(ls $O/overlap/*/*kernel_stats.csv | head -1) $O/r04_bench_b16_kernel_stats_overlap.csv
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_graph > $O/fetch.log 2>&1 || exit 5
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_graph > $O/write.log 2>&1 || exit 6
python3 $R/tools/pmc_hbm_summary.py $O/fetch $O/write $O/r04_pmc_hbm_traffic.json > $O/pmc.log 2>&1
rm -rf $O/serial $O/overlap $O/fetch $O/write
echo done a; exit 0
fi
python3 $R/tools/profile_layers.py > $O/r04_conv_layers.log 2>&1 || exit 4
cd $R
OP=fwd KS=5 bash tools/sq_prof_k.sh conv_halo_x3 halo_run.py > $O/r04_halo_fwd5_sq_counters.txt 2>&1
OP=fwd KS=3 bash tools/sq_prof_k.sh conv_halo_x3 halo_run.py > $O/r04_halo_fwd3_sq_counters.txt 2>&1
OP=wgrad KS=3 bash tools/sq_prof_k.sh conv_wgrad_roll_x3 halo_run.py > $O/r04_wgrad_roll3_sq_counters.txt 2>&1
PD_WGRAD_ROLL=0 OP=wgrad KS=3 bash tools/sq_prof_k.sh conv_wgrad_halo_x3 halo_run.py > $O/r04_wgrad_halo3_sq_counters.txt 2>&1
OP=wgrad KS=5 bash tools/sq_prof_k.sh conv_wgrad_halo_x3 halo_run.py > $O/r04_wgrad_halo5_sq_counters.txt 2>&1
python3 tools/bench_polar.py --quick > $O/r04_polar_kernel_gbps.log 2>&1
bash tools/k1_variants.sh > $O/r04_polar_kernel_nt_loads.log 2>&1
tools/halo_ab.sh $O/r04_halo_vs_gather_layers.log
tools/tail_ab.sh $O/r04_conv_tail_layers.log
tools/ab_roll.sh > $O/r04_wgrad_roll_vs_rows.log 2>&1
echo done b
