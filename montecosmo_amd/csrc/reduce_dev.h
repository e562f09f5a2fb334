// Block / grid reductions in f64 shared by the adjoint kernels (composite.hip, bias.hip, observe.hip): every workgroup writes its
// partial sums, det_fold_kernel adds them up in a fixed order (see DETERMINISTIC GRID SUMS below).
#pragma once
#include <hip/hip_runtime.h>

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

// DETERMINISTIC GRID SUMS (round 4).  Round 3 added every workgroup's float64 partial to one of NSLOT spread slots with atomicAdd: several
// workgroups per slot, in arrival order, so the last bit of a scalar cotangent moved from call to call (4e-16 .. 7e-16 relative; invisible at
// the float32 the samplers carry, visible to a float64 equality).  Now every workgroup WRITES its partials (fixed tree inside the
// workgroup) to P[k * nblk + block], and det_fold_kernel adds them up in a fixed order: R workgroups each sum a contiguous range of P into
// Q[k * R + r], the last one to finish (an integer ticket) sums Q with the same fixed tree.  No floating-point atomic is left on the gradient path.
template <int K>
__device__ __forceinline__ void block_partial(const double (&v)[K], double *__restrict__ P, unsigned nblk, unsigned blk) {
    __shared__ double sh[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 63) sh[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int nw = (blockDim.x + 63) >> 6;
        double t = 0.;
        for (int w = 0; w < nw; ++w) t += sh[threadIdx.x][w];
        P[(size_t)threadIdx.x * nblk + blk] = t;
    }
}

struct DetOuts {
    double *p[10];      // destination of value k (NULL: dropped)
    int accumulate;     // 1: *p[k] += scale sum, 0: *p[k] = scale sum
};
__host__ inline DetOuts det_outs(double *o0) {      // one value, stored (not accumulated)
    DetOuts o{};
    o.p[0] = o0;
    return o;
}
// scratch behind P: Q (K * R doubles) and the ticket (one unsigned, zero between launches).  K <= 10, R <= 1024.
// (The last workgroup's sum over Q is a fixed TREE over 256 lanes, not a serial loop: 256 dependent-latency loads by one lane cost
// 43 us at 256^3 and 55 us at 512^3 -- `profiles/r04_kernel_stats_*.csv` of the first version -- against 5 us for everything else.)
__global__ __launch_bounds__(256) void det_fold_kernel(const double *__restrict__ P, unsigned nblk, int K, double *Q, unsigned *ticket, double scale,
                                                       DetOuts o) {
    const unsigned R = gridDim.x, r = blockIdx.x, C = (nblk + R - 1) / R, lo = r * C, hi = min(lo + C, nblk);
    __shared__ double sh[10][4];
    __shared__ int last;
    for (int k = 0; k < K; ++k) {
        // eight independent loads in flight per lane, added in a fixed pattern (one load per iteration is a chain of memory latencies:
        // 29 us per fold at 512^3, 16 iterations x 3 rows)
        const double *Pk = P + (size_t)k * nblk;
        double t = 0.;
        for (unsigned i = lo + threadIdx.x; i < hi; i += 256 * 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = i + 256u * j < hi ? Pk[i + 256u * j] : 0.;
            t += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 63) sh[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < K) Q[(size_t)threadIdx.x * R + r] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(ticket, 1u) == R - 1u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    for (int k = 0; k < K; ++k) {      // lane i holds Q[k][i] (+ Q[k][i + 256] ..., in that order); the same DPP tree and wave order as above
        double t = 0.;
        for (unsigned i = threadIdx.x; i < R; i += 256) t += __hip_atomic_load(Q + (size_t)k * R + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 63) sh[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < K) {
        const double t = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
        double *dst = o.p[threadIdx.x];
        if (dst) *dst = (o.accumulate ? *dst : 0.) + scale * t;
    }
    if (threadIdx.x == 0) *ticket = 0u;
}

}  // namespace
