// What does a PARTIALLY active gather cost?  8 coalesced 12-byte corner gathers per lane (interleaved [cell][3] mesh, lanes along z,
// 512^3) as in the step kernels; the last four are executed only by a fraction of the lanes (scattered, by a hash of the lane), the
// others taking the value a neighbour lane already holds (here: a DPP move of their own).  If the texture-address path is paid per
// instruction, sharing z-adjacent corners between neighbouring lanes cannot pay; if per active lane (or quad), it can.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_mask_bench.hip -o tools/gather_mask_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
template <int MODE>   // 0: hash per lane; 1: every k-th lane (pct = 100 / k); 2: contiguous lanes [0, 64 pct / 100)
__global__ __launch_bounds__(256) void kaos(const float *__restrict__ m, const float *__restrict__ pos, float *__restrict__ out, int n, int pct) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int z = (int)(i % n), y = (int)((i / n) % n), x = (int)(i / ((int64_t)n * n));
    const F3 p = *(const F3 *)((const char *)pos + i * 12);
    float a0 = p.a, a1 = p.b, a2 = p.c;
    const char *mb = (const char *)m;
    const int lane = threadIdx.x & 63;
    bool act;
    if (MODE == 0) act = (int)(((unsigned)(lane * 2654435761u + blockIdx.x * 40503u) >> 8) % 100u) < pct;
    else if (MODE == 1) act = pct > 0 && (lane % (100 / pct)) == 0;
    else act = lane < (64 * pct) / 100;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int dx = (r >> 1) & 1, dy = r & 1, dz = (r >> 2) & 1;      // the four z+1 corners last
        const uint32_t cell = (uint32_t)((((x + dx) & (n - 1)) * n + ((y + dy) & (n - 1))) * n + ((z + dz) & (n - 1)));
        if (r < 4 || act) {
            const F3 v = *(const F3 *)(mb + (size_t)cell * 12u);
            a0 += v.a; a1 += v.b; a2 += v.c;
        } else {
            a0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a0), 0x101, 0xf, 0xf, true));   // row_shl:1
            a1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a1), 0x101, 0xf, 0xf, true));
            a2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), 0x101, 0xf, 0xf, true));
        }
    }
    F3 o; o.a = a0; o.b = a1; o.c = a2;
    *(F3 *)((char *)out + i * 12) = o;
}
__global__ void fill(float *m, int64_t n) { const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i < n) m[i] = (float)((i * 2654435761u) & 1023) * 0.001f; }
int main() {
    const int n = 512;
    const int64_t N = (int64_t)n * n * n;
    float *m, *out, *pos;
    CK(hipMalloc(&m, 3 * N * 4 + 64)); CK(hipMalloc(&out, N * 12)); CK(hipMalloc(&pos, N * 12));
    CK(hipMemset(pos, 0, N * 12));
    fill<<<(unsigned)((3 * N + 255) / 256), 256>>>(m, 3 * N);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
#define RUN(MODE, pct, name) { kaos<MODE><<<(unsigned)(N / 256), 256>>>(m, pos, out, n, pct); CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) kaos<MODE><<<(unsigned)(N / 256), 256>>>(m, pos, out, n, pct); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-36s %3d %% of the lanes run gathers 5-8: %.3f ms\n", name, pct, ms / 5); }
    for (int pct : {100, 50, 33, 25, 10, 0}) RUN(0, pct, "scattered lanes (hash)");
    for (int pct : {50, 25, 10}) RUN(1, pct, "every k-th lane");
    for (int pct : {50, 25, 10}) RUN(2, pct, "the first lanes of the wave");
    return 0;
}
