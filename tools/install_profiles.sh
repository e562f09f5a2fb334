#!/bin/bash
# Copies the summaries of a tools/collect_profiles.sh run (gpurun_out/<tag>/) to profiles/<round>_*: tools/install_profiles.sh r04b r04
tag=${1:-r04}; rnd=${2:-r04}; O=gpurun_out/$tag
for f in bench512 bench256 bench128 bench256_fwd bench128_fwd bench512_slab1 pmc_traffic; do cp $O/$f.json profiles/${rnd}_$f.json; done
for n in 128 256 512; do cp $O/stats$n/k_kernel_stats.csv profiles/${rnd}_kernel_stats_$n.csv; done
cp $O/pmc_per_kernel_256.txt profiles/${rnd}_pmc_per_kernel_256.txt
cp $O/pmc_per_kernel_512.txt profiles/${rnd}_pmc_per_kernel_512.txt
[ -f $O/halo_tiles.txt ] && cp $O/halo_tiles.txt profiles/${rnd}_halo_tiles.txt
ls -la profiles/${rnd}_* | wc -l
