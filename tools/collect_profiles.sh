#!/bin/bash
# Round profile set (run on the GPU box through gpurun): tools/collect_profiles.sh r02b
#   bench lines (512^3 default with the 256^3 sub-record and the CPU baseline; 256^3; 128^3; forward-only; slab code path),
#   rocprofv3 --kernel-trace --stats of the default workload and of 256^3, PMC FETCH_SIZE / WRITE_SIZE passes (separate runs,
#   as MI355X_MICROARCH.md prescribes) for both meshes.  Outputs under gpurun_out/<tag>/; copy what is to be judged to profiles/.
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag && O=gpurun_out/$tag && \
python3 bench.py > $O/bench512.json 2> $O/bench512.err && echo "bench512 done" && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats512 -o k -- python3 bench.py --no-sub-record --no-cpu-baseline > $O/bench512_prof.json 2>> $O/prof.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats256 -o k -- python3 bench.py --no-sub-record --no-cpu-baseline --mesh 256 > $O/bench256_prof.json 2>> $O/prof.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats128 -o k -- python3 bench.py --no-sub-record --no-cpu-baseline --mesh 128 > $O/bench128_prof.json 2>> $O/prof.err && echo "stats done" && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch512 -- python3 bench.py --no-sub-record --no-cpu-baseline --warmup 0 > $O/pmc_fetch512.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write512 -- python3 bench.py --no-sub-record --no-cpu-baseline --warmup 0 > $O/pmc_write512.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch256 -- python3 bench.py --no-sub-record --no-cpu-baseline --warmup 0 --mesh 256 > $O/pmc_fetch256.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write256 -- python3 bench.py --no-sub-record --no-cpu-baseline --warmup 0 --mesh 256 > $O/pmc_write256.log 2>&1 && echo "pmc done" && \
python3 tools/make_traffic_json.py $O/pmc_fetch512 $O/pmc_write512 512 $O/pmc_traffic.json $O/pmc_per_kernel_512.txt > $O/traffic512.txt && \
python3 tools/make_traffic_json.py $O/pmc_fetch256 $O/pmc_write256 256 $O/pmc_traffic.json $O/pmc_per_kernel_256.txt > $O/traffic256.txt && \
python3 bench.py --mesh 256 --no-cpu-baseline > $O/bench256.json 2>> $O/bench.err && \
python3 bench.py --mesh 128 --no-cpu-baseline > $O/bench128.json 2>> $O/bench.err && \
python3 bench.py --mesh 256 --forward-only --no-cpu-baseline > $O/bench256_fwd.json 2>> $O/bench.err && \
python3 bench.py --mesh 128 --forward-only --no-cpu-baseline > $O/bench128_fwd.json 2>> $O/bench.err && \
python3 bench.py --force-slab --no-cpu-baseline > $O/bench512_slab1.json 2>> $O/bench.err && \
(python3 tools/halo_tiles.py 512 && python3 tools/halo_tiles.py 256) > $O/halo_tiles.txt 2>> $O/bench.err; echo rc=$?
# the raw counter csv files are large: keep the summaries only
rm -rf $O/pmc_fetch512 $O/pmc_write512 $O/pmc_fetch256 $O/pmc_write256 $O/stats512/*trace.csv $O/stats256/*trace.csv $O/stats128/*trace.csv
