#!/bin/bash
# Chunked transposes on the one-GPU proxies: (a) the local-copy communicator (launch overhead of the extra windows only),
# (b) ONE RCCL rank (all-to-alls and ghost exchanges are real RCCL kernels: ~3 ms per step at 512^3 to hide).
mkdir -p gpurun_out/chunks
p=29700
for c in ${CHUNKS:-1 2 4 8}; do
  MCPM_SLAB_CHUNKS=$c python bench.py --force-slab --chains 1 --no-cpu-baseline --steps 10 "$@" > gpurun_out/chunks/local_$c.json 2> gpurun_out/chunks/local_$c.err || { echo failed local $c; tail -3 gpurun_out/chunks/local_$c.err; exit 1; }
  p=$((p+1))
  MCPM_SLAB_CHUNKS=$c MCPM_BENCH_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$p RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --chains 1 --no-cpu-baseline --steps 10 "$@" > gpurun_out/chunks/rccl_$c.json 2> gpurun_out/chunks/rccl_$c.err || { echo failed rccl $c; tail -3 gpurun_out/chunks/rccl_$c.err; exit 1; }
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
a = json.loads([l for l in open(f"gpurun_out/chunks/local_{c}.json") if l.startswith("{")][-1])
b = json.loads([l for l in open(f"gpurun_out/chunks/rccl_{c}.json") if l.startswith("{")][-1])
print(f"chunks {c}: local-copy communicator {a['ms_per_step']} ms/step; one RCCL rank {b['ms_per_step']} ms/step")
PY
done
