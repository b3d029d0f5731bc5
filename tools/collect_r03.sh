#!/bin/bash
# Evidence pass of round 3 (GPU box): bench line (eager + hipGraph), rocprofv3 kernel stats (serial + overlapped), step
# breakdown, per-layer conv table, HBM traffic PMC passes, K1 timeline / read-schedule microbenchmarks, attention bench.
# usage: collect_r03.sh a | b | k1  (gpurun calls: a = bench + kernel stats + PMC traffic, b = per-layer / bf16x3 / attention, k1 = K1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
STAGE=${1:-a}
if [ "$STAGE" = a ]; then
python3 $R/bench.py > $O/r03_bench_b16.json 2> $O/bench.err || exit 1
rm -rf $O/serial $O/overlap $O/fetch $O/write
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_graph > $O/serial.log 2>&1 || exit 2
python3 $R/tools/step_breakdown.py $O/serial 70 > $O/r03_step_breakdown_serial.txt
cp $(ls $O/serial/*/*kernel_stats.csv | head -1) $O/r03_bench_b16_kernel_stats_serial.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_graph > $O/overlap.log 2>&1 || exit 3
cp $(ls $O/overlap/*/*kernel_stats.csv | head -1) $O/r03_bench_b16_kernel_stats_overlap.csv
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_graph > $O/fetch.log 2>&1 || exit 5
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_graph > $O/write.log 2>&1 || exit 6
python3 $R/tools/pmc_hbm_summary.py $O/fetch $O/write $O/r03_pmc_hbm_traffic.json > $O/pmc.log 2>&1
rm -rf $O/serial $O/overlap $O/fetch $O/write
echo done a; exit 0
fi
if [ "$STAGE" = k1 ]; then     # K1 evidence (unchanged by the convolution work: run once)
cd $R
python3 tools/k1_trace.py 2>&1 | grep event > $O/r03_k1_timeline.log
tools/bin/membench3 > $O/r03_membench3_read_schedules.log 2>&1
bash tools/k1_threads.sh 2>&1 | grep GBps > $O/r03_polar_kernel_gbps.log
echo done k1; exit 0
fi
python3 $R/tools/profile_layers.py > $O/r03_conv_layers.log 2>&1 || exit 4
cd $R
tools/bin/bf16x3_peak > $O/r03_bf16x3_peak.log 2>&1
bash tools/sq_prof_k.sh conv_igemm x3_run.py > $O/r03_x3_sq_counters.txt 2>&1
bash tools/sq_prof_k.sh conv_wgrad x3c_run.py > $O/r03_x3c_sq_counters.txt 2>&1
python3 bench.py --attention --bf16 --no_graph --steps 5 > $O/r03_bench_attention_bf16.json 2> $O/att.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/att -- python3 $R/bench.py --attention --bf16 --steps 2 --warmup 1 --no_cpu_baseline --no_graph > $O/att.log 2>&1
cp $(ls $O/att/*/*kernel_stats.csv | head -1) $O/r03_attention_bf16_kernel_stats.csv; rm -rf $O/att
echo done
