// Streaming ceiling by access width on gfx950: out = a + b over 1.6 GB arrays (the particle arrays of a 512^3 run),
// with 4-, 12- (AoS float3, what the particle kernels use) and 16-byte accesses per lane.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_bench.hip -o tools/stream_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__global__ __launch_bounds__(256) void k1(const float *a, const float *b, float *o, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void k3(const F3 *a, const F3 *b, F3 *o, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { F3 x = a[i], y = b[i]; o[i] = F3{x.x + y.x, x.y + y.y, x.z + y.z}; }
}
__global__ __launch_bounds__(256) void k4(const float4 *a, const float4 *b, float4 *o, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 x = a[i], y = b[i]; o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
}
// two reads + two writes of float3 (kick_drift's streaming part)
__global__ __launch_bounds__(256) void k3rw(const F3 *a, const F3 *b, F3 *o, F3 *o2, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { F3 x = a[i], y = b[i]; F3 s{x.x + y.x, x.y + y.y, x.z + y.z}; o[i] = s; o2[i] = F3{s.x * 2.f, s.y, s.z}; }
}
int main() {
    const int64_t nf = (int64_t)512 * 512 * 512 * 3;   // floats per array
    float *a, *b, *o, *o2;
    CK(hipMalloc(&a, nf * 4)); CK(hipMalloc(&b, nf * 4)); CK(hipMalloc(&o, nf * 4)); CK(hipMalloc(&o2, nf * 4));
    CK(hipMemset(a, 0, nf * 4)); CK(hipMemset(b, 0, nf * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
#define RUN(name, bytes, launch) { launch; CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) { launch; } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-34s %.3f ms  %.2f TB/s\n", name, ms / 5, (bytes) / (ms / 5 * 1e-3) / 1e12); }
    RUN("4 B / lane  (2 reads + 1 write)", 3.0 * nf * 4, (k1<<<(unsigned)((nf + 255) / 256), 256>>>(a, b, o, nf)));
    RUN("12 B / lane (2 reads + 1 write)", 3.0 * nf * 4, (k3<<<(unsigned)((nf / 3 + 255) / 256), 256>>>((F3 *)a, (F3 *)b, (F3 *)o, nf / 3)));
    RUN("16 B / lane (2 reads + 1 write)", 3.0 * nf * 4, (k4<<<(unsigned)((nf / 4 + 255) / 256), 256>>>((float4 *)a, (float4 *)b, (float4 *)o, nf / 4)));
    RUN("12 B / lane (2 reads + 2 writes)", 4.0 * nf * 4, (k3rw<<<(unsigned)((nf / 3 + 255) / 256), 256>>>((F3 *)a, (F3 *)b, (F3 *)o, (F3 *)o2, nf / 3)));
    return 0;
}
