#!/bin/bash
# Evidence pass of round 2 (GPU box): bench line, rocprofv3 kernel stats (serial + overlapped), step breakdown,
# per-layer conv table, HBM traffic PMC passes, SQ counters of the conv kernels, fp32 MFMA/VALU microbench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/r02_bench_b16.json 2> $O/bench.err || exit 1
rm -rf $O/serial $O/overlap $O/fetch $O/write
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/serial.log 2>&1 || exit 2
python3 $R/tools/step_breakdown.py $O/serial 60 > $O/r02_step_breakdown_serial.txt
cp $(ls $O/serial/*/*kernel_stats.csv | head -1) $O/r02_bench_b16_kernel_stats_serial.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline > $O/overlap.log 2>&1 || exit 3
cp $(ls $O/overlap/*/*kernel_stats.csv | head -1) $O/r02_bench_b16_kernel_stats_overlap.csv
python3 $R/tools/profile_layers.py > $O/r02_conv_layers.log 2>&1 || exit 4
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline > $O/fetch.log 2>&1 || exit 5
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no_cpu_baseline > $O/write.log 2>&1 || exit 6
python3 $R/tools/pmc_hbm_summary.py $O/fetch $O/write $O/r02_pmc_hbm_traffic.json > $O/pmc.log 2>&1
cd $R && tools/sq_prof.sh ev > $O/r02_conv_sq_counters.txt 2>&1
tools/sq_prof2.sh ev > $O/r02_conv_sq_counters2.txt 2>&1
tools/bin/mfma_peak > $O/r02_mfma_peak.log 2>&1
python3 tools/bench_fwd.py > $O/r02_conv_bench_fwd.log 2>&1
python3 tools/bench_chain.py > $O/r02_chain_kernels.log 2>&1
python3 tools/bench_c16.py > $O/r02_conv16_kernels.log 2>&1
rm -rf $O/serial $O/overlap $O/fetch $O/write $R/gpurun_out/sq_ev $R/gpurun_out/sq2_ev
echo done
