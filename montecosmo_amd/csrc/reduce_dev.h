// Block / grid reductions in f64 shared by the model-side kernels (bias.hip, observe.hip): per-block sums are added to
// NSLOT spread slots (one address would serialise at the memory-side atomic unit), then folded by one block.
#pragma once
#include <hip/hip_runtime.h>

#define NSLOT 256

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// adds the block's sum of each of the K values to slots[k * NSLOT + blockIdx.x % NSLOT]
template <int K>
__device__ __forceinline__ void block_add(const double (&v)[K], double *slots) {
    __shared__ double sh[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const double t = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        if (t != 0.) atomicAdd(slots + threadIdx.x * NSLOT + (blockIdx.x % NSLOT), t);
    }
}

// out[k] = scale * sum of slot row k
__global__ __launch_bounds__(NSLOT) void fold_kernel(const double *__restrict__ slots, int K, double scale, double *out) {
    __shared__ double sh[NSLOT / 64];
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(slots[k * NSLOT + threadIdx.x]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.;
            for (int i = 0; i < NSLOT / 64; ++i) t += sh[i];
            out[k] = scale * t;
        }
        __syncthreads();
    }
}


}  // namespace
