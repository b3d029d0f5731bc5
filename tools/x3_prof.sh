#!/bin/bash
# kernel statistics of the serial step (GPU box): which convolution kernels the step runs, and for how long
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/x3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/serial
PD_WGRAD_STREAM=0 PD_ENCODER_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_graph > $O/serial.log 2>&1 || exit 2
python3 $R/tools/step_breakdown.py $O/serial 30 > $O/step_breakdown_serial.txt
cp $(ls $O/serial/*/*kernel_stats.csv | head -1) $O/kernel_stats_serial.csv
rm -rf $O/serial
head -32 $O/step_breakdown_serial.txt
