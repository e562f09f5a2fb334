for pad in 16 8 12 24; do
  touch montecosmo_amd/csrc/fftpm.hip
  make -C montecosmo_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -DMCPM_NZPAD=$pad" > gpurun_out/build_$pad.log 2>&1
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']; print('pad', $pad, d['ms_per_step'], 'r2c', s['fft_r2c']['ms_per_launch'], 'c2r', s['fft_c2r']['ms_per_launch'], 'kspace', s['kspace']['ms_per_launch'], 'pm_forces', d['force_cycle']['pm_forces_ms'])"
done
