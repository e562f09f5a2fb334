// Microbenchmark: LDS atomic-add rate on gfx950 for f32 / u32 / u64 / f32-via-CAS-free alternatives.
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ __launch_bounds__(256) void k(int iters, int pattern, T *out) {
    __shared__ T tile[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tile[i] = T(0);
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        int idx;
        if (pattern == 0) idx = (threadIdx.x + it * 64) & 8191;                  // lane-linear, conflict free
        else { s = s * 1664525u + 1013904223u; idx = ((threadIdx.x + it * 64) + ((s >> 20) & 3)) & 8191; }  // jittered
        atomicAdd(&tile[idx], T(1));
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[5];
}

template <typename T>
void run(const char *name) {
    T *out;
    hipMalloc(&out, sizeof(T) * 4096);
    for (int pattern = 0; pattern < 2; ++pattern) {
        int iters = 4096, blocks = 256 * 8;
        k<T><<<blocks, 256>>>(16, pattern, out);
        hipDeviceSynchronize();
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        k<T><<<blocks, 256>>>(iters, pattern, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double lane_atomics = (double)blocks * 256 * iters;
        printf("%s pattern %d: %.3f ms, %.2f G lane-atomics/s, %.2f lane-atomics/clk/CU (2.4GHz,256CU)\n", name, pattern, ms,
               lane_atomics / ms / 1e6, lane_atomics / (ms * 1e-3) / 2.4e9 / 256);
    }
    hipFree(out);
}

int main() {
    run<float>("f32");
    run<int>("i32");
    run<unsigned long long>("u64");
    run<double>("f64");
    return 0;
}
