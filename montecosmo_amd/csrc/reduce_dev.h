// Block / grid reductions in f64 shared by the model-side kernels (bias.hip, observe.hip): per-block sums are added to
// NSLOT spread slots (one address would serialise at the memory-side atomic unit), then folded by one block.
#pragma once
#include <hip/hip_runtime.h>

#define NSLOT 256

namespace {

// Sum over the 64 lanes, valid in LANE 63, by DPP (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31 on the two halves of the double:
// 18 VALU instructions) instead of __shfl_down, which is two ds_bpermute_b32 per step through the LDS crossbar (DESIGN finding
// 27).  Every lane of the wave must be active.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_shift_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);       // lanes without a source read +0.0
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_shift_d<0x111, 0xf>(v);
    v += dpp_shift_d<0x112, 0xf>(v);
    v += dpp_shift_d<0x114, 0xf>(v);
    v += dpp_shift_d<0x118, 0xf>(v);
    v += dpp_shift_d<0x142, 0xa>(v);
    v += dpp_shift_d<0x143, 0xc>(v);
    return v;
}

// adds the block's sum of each of the K values to slots[k * NSLOT + blockIdx.x % NSLOT]
template <int K>
__device__ __forceinline__ void block_add(const double (&v)[K], double *slots) {
    __shared__ double sh[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 63) sh[k][threadIdx.x >> 6] = s;
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
        if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = s;
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
