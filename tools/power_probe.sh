#!/bin/bash
# Average package power and shader clock while the training step runs, with the convolutions on the bf16-split kernels and
# on the fp32 MFMA (GPU box): rocm-smi sampled every 0.5 s next to `bench.py --steps 400`.
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  PD_CONV_X3=$v PD_WGRAD_X3C=$v python3 bench.py --no_graph --steps 400 --warmup 5 --no_cpu_baseline > /tmp/pp_$v.json 2>/dev/null &
  BP=$!
  sleep 9
  echo "== PD_CONV_X3=$v PD_WGRAD_X3C=$v"
  for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level" | tr '\n' ';'
    echo
    sleep 0.5
  done
  wait $BP
  python3 -c "import json; d=json.loads([l for l in open('/tmp/pp_$v.json') if l.startswith('{')][-1]); print('images/s', d['value'], 'ms/step', d['ms_per_step'])"
done
