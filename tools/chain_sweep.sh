#!/bin/bash
# One-GPU proxy for the overlap of the slab path's collectives: one RCCL rank (self send/recv, one-rank all-to-all: ~3.6 ms of
# RCCL copies per step at 512^3) with two interleaved trajectories, over pipeline offsets and with / without the compute token.
mkdir -p gpurun_out/chains
export MCPM_BENCH_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
p=29520
for q in ${QUEUES:-4 8}; do export GPU_MAX_HW_QUEUES=$q; for tok in ${TOKENS:-1 0}; do for off in ${OFFSETS:-0 2 4}; do
  p=$((p+1))
  MASTER_PORT=$p MCPM_CHAIN_TOKEN=$tok MCPM_CHAIN_OFFSET=$off timeout -k 10 300 python bench.py --chains 2 --no-cpu-baseline "$@" > gpurun_out/chains/t${tok}_o${off}.json 2> gpurun_out/chains/t${tok}_o${off}.err || { echo failed t$tok o$off; tail -3 gpurun_out/chains/t${tok}_o${off}.err; exit 1; }
  python - gpurun_out/chains/t${tok}_o${off}.json $tok $off <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ic = d["interleaved_chains"]
import os
print(f"queues {os.environ.get('GPU_MAX_HW_QUEUES')} token {sys.argv[2]} offset {sys.argv[3]}: single {d['ms_per_step']} ms/step, two chains {ic['ms_per_step']} ms per chain-step ({ic['vs_single_trajectory']}x)")
PY
done; done; done
