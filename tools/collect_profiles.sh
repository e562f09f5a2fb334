cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r01g && \
python bench.py > gpurun_out/r01g/bench512.json 2> gpurun_out/r01g/bench512.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01g/stats -o r01g -- python bench.py --no-cpu-baseline > gpurun_out/r01g/bench512_prof.json 2>> gpurun_out/r01g/bench512.err && \
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01g/pmc_fetch -- python bench.py --no-cpu-baseline --warmup 0 > gpurun_out/r01g/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01g/pmc_write -- python bench.py --no-cpu-baseline --warmup 0 > gpurun_out/r01g/pmc_write.log 2>&1 && \
python bench.py --mesh 256 --no-cpu-baseline > gpurun_out/r01g/bench256.json 2>> gpurun_out/r01g/bench512.err && \
python bench.py --mesh 128 --no-cpu-baseline > gpurun_out/r01g/bench128.json 2>> gpurun_out/r01g/bench512.err && \
python bench.py --mesh 256 --forward-only --no-cpu-baseline > gpurun_out/r01g/bench256_fwd.json 2>> gpurun_out/r01g/bench512.err && \
python bench.py --mesh 128 --forward-only --no-cpu-baseline > gpurun_out/r01g/bench128_fwd.json 2>> gpurun_out/r01g/bench512.err; echo rc=$?
