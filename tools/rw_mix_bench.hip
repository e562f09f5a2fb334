// Streaming ceilings by read : write mix on gfx950 (16 B per lane, 0.54 GB arrays = one 512^3 half-spectrum): what can a
// pass that reads R arrays and writes W arrays reach, with plain and with nontemporal stores?
//   hipcc -O3 --offload-arch=gfx950 tools/rw_mix_bench.hip -o tools/rw_mix_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void mix(const f4 *__restrict__ a, f4 *__restrict__ o, int64_t n, int64_t stride, float never) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f4 s = {1.f, 2.f, 3.f, (float)i};
#pragma unroll
    for (int r = 0; r < R; ++r) s += a[i + r * stride];
    if (W == 0) {
        if (s.x == never) o[i] = s;
        return;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (NT) __builtin_nontemporal_store(s, &o[i + w * stride]);
        else o[i + w * stride] = s;
    }
}
int main() {
    const int64_t n = (int64_t)512 * 512 * 272 * 8 / 16;   // f4 elements per array (0.57 GB)
    f4 *a, *o;
    CK(hipMalloc(&a, n * 16 * 3)); CK(hipMalloc(&o, n * 16 * 3));
    CK(hipMemset(a, 0, n * 16 * 3));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
#define RUN(R, W, NT) { auto L = [&] { mix<R, W, NT><<<(unsigned)((n + 255) / 256), 256>>>(a, o, n, n, -1.f); }; L(); CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) L(); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; \
        printf("reads %d writes %d %-3s  %.4f ms  total %.2f TB/s  (read %.2f, write %.2f)\n", R, W, NT ? "nt" : "", ms, (R + W) * n * 16.0 / ms / 1e9, R * n * 16.0 / ms / 1e9, W * n * 16.0 / ms / 1e9); }
    RUN(1, 0, false) RUN(3, 0, false)
    RUN(0, 1, false) RUN(0, 1, true) RUN(0, 3, false) RUN(0, 3, true)
    RUN(1, 1, false) RUN(1, 1, true)
    RUN(2, 1, false) RUN(2, 1, true)
    RUN(1, 2, false) RUN(1, 2, true)
    RUN(2, 3, false) RUN(2, 3, true)
    RUN(3, 3, false) RUN(3, 3, true)
    RUN(3, 1, false) RUN(3, 1, true)
    return 0;
}
