// Is an aligned 16-byte gather cheaper than the 12-byte one the step kernels use?  Interleaved force mesh [cell][3] (12-byte
// corners, global_load_dwordx3) against [cell][4] (16-byte aligned corners, global_load_dwordx4 with the fourth float USED, so the
// compiler cannot shrink the load), 8 corner gathers per lane, lanes along z, 512^3, smooth synthetic displacements.
//   hipcc -O3 --offload-arch=gfx950 tools/gather4_bench.hip -o tools/gather4_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__constant__ float g_amp;
__device__ __forceinline__ void displaced(int n, int &x, int &y, int &z) {
    if (g_amp == 0.f) return;
    const float px = 0.049f * x, py = 0.037f * y, pz = 0.043f * z;
    const int ox = (int)floorf(g_amp * __sinf(py + 2.f * pz + 0.5f * px)), oy = (int)floorf(g_amp * __sinf(pz + 2.f * px + 0.5f * py)),
              oz = (int)floorf(g_amp * __sinf(px + 2.f * py + 0.5f * pz));
    x = (x + ox + n) % n; y = (y + oy + n) % n; z = (z + oz + n) % n;
}
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
template <int W, bool IO>
__global__ __launch_bounds__(256) void kaos(const float *__restrict__ m, const float *__restrict__ pos, float *__restrict__ out, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int z = (int)(i % n), y = (int)((i / n) % n), x = (int)(i / ((int64_t)n * n));
    displaced(n, x, y, z);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (IO) {   // a 12-byte streaming read per lane, as the real kernels have
        const F3 p = *(const F3 *)((const char *)pos + i * 12);
        a0 = p.a; a1 = p.b; a2 = p.c;
    }
    const char *mb = (const char *)m;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int dx = (r >> 2) & 1, dy = (r >> 1) & 1, dz = r & 1;
        const uint32_t cell = (uint32_t)((((x + dx) % n) * n + (y + dy) % n) * n + (z + dz) % n);
        if (W == 3) {
            const F3 v = *(const F3 *)(mb + (size_t)cell * 12u);
            a0 += v.a; a1 += v.b; a2 += v.c;
        } else {
            const float4 v = *(const float4 *)(mb + (size_t)cell * 16u);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
    if (IO) {
        F3 o; o.a = a0 + a3; o.b = a1; o.c = a2;
        *(F3 *)((char *)out + i * 12) = o;
    } else
        out[i] = a0 + 2.f * a1 + 3.f * a2 + 4.f * a3;
}
__global__ void fill(float *m, int64_t n) { const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) m[i] = (float)((i * 2654435761u) & 1023) * 0.001f; }
int main() {
    const int n = 512;
    const int64_t N = (int64_t)n * n * n;
    float *m, *out, *pos;
    CK(hipMalloc(&m, 4 * N * 4 + 64)); CK(hipMalloc(&out, N * 12)); CK(hipMalloc(&pos, N * 12));
    CK(hipMemset(pos, 0, N * 12));
    fill<<<(unsigned)((4 * N + 255) / 256), 256>>>(m, 4 * N);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
#define RUN(W, IO, name) { kaos<W, IO><<<(unsigned)(N / 256), 256>>>(m, pos, out, n); CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) kaos<W, IO><<<(unsigned)(N / 256), 256>>>(m, pos, out, n); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-52s %.3f ms\n", name, ms / 5); }
    for (float amp : {0.f, 1.f, 3.f}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_amp), &amp, sizeof(float)));
        printf("--- smooth displacement amplitude %.0f cells\n", amp);
        RUN(3, false, "[cell][3], 8 x dwordx3 gathers");
        RUN(4, false, "[cell][4], 8 x aligned dwordx4 gathers");
        RUN(3, true, "[cell][3] + 12 B in / 12 B out per lane");
        RUN(4, true, "[cell][4] + 12 B in / 12 B out per lane");
    }
    return 0;
}
