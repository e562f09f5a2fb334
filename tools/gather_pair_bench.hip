// The two z corners of a CIC stencil are adjacent in the interleaved [cell][3] force mesh: 24 contiguous bytes at a 12-byte
// aligned address.  Is one dwordx4 + one dwordx2 per (x, y) corner pair cheaper on the texture-address path than the two dwordx3
// gathers the step kernels issue?  512^3, lanes along z, 12 B in / 12 B out per lane as in read3.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_pair_bench.hip -o tools/gather_pair_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; };
struct __attribute__((packed, aligned(4))) F2 { float a, b; };
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v3 __attribute__((ext_vector_type(3)));
typedef float v4 __attribute__((ext_vector_type(4)));
// the instructions themselves (the compiler merges or splits adjacent struct loads as it sees fit); one wait after the last
#define LD(W, dst, ptr) asm volatile("global_load_dword" W " %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
template <int MODE>   // 0: 8 x dwordx3; 1: 4 x (dwordx4 + dwordx2); 2: 4 x (3 x dwordx2); 3: 4 x dwordx4 only; 4: 8 x dwordx2 only
__global__ __launch_bounds__(256) void kaos(const float *__restrict__ m, const float *__restrict__ pos, float *__restrict__ out, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int z = (int)(i % n), y = (int)((i / n) % n), x = (int)(i / ((int64_t)n * n));
    const F3 p = *(const F3 *)((const char *)pos + i * 12);
    float a0 = p.a, a1 = p.b, a2 = p.c;
    const char *mb = (const char *)m;
    const int zc = z < n - 1 ? z : n - 2;      // keep the pair inside the row (the bench does not model the wrap)
    const char *q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int dx = (r >> 1) & 1, dy = r & 1;
        q[r] = mb + (size_t)((uint32_t)((((x + dx) & (n - 1)) * n + ((y + dy) & (n - 1))) * n + zc)) * 12u;
    }
    v3 u3[4], w3[4];
    v4 u4[4];
    v2 u2[4], w2[4], x2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (MODE == 0) { LD("x3", u3[r], q[r]); LD("x3", w3[r], q[r] + 12); }
        if (MODE == 1) { LD("x4", u4[r], q[r]); LD("x2", u2[r], q[r] + 16); }
        if (MODE == 2) { LD("x2", u2[r], q[r]); LD("x2", w2[r], q[r] + 8); LD("x2", x2[r], q[r] + 16); }
        if (MODE == 3) { LD("x4", u4[r], q[r]); }
        if (MODE == 4) { LD("x2", u2[r], q[r]); LD("x2", w2[r], q[r] + 16); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (MODE == 0) { a0 += u3[r].x + w3[r].x; a1 += u3[r].y + w3[r].y; a2 += u3[r].z + w3[r].z; }
        if (MODE == 1) { a0 += u4[r].x + u4[r].w; a1 += u4[r].y + u2[r].x; a2 += u4[r].z + u2[r].y; }
        if (MODE == 2) { a0 += u2[r].x + w2[r].y; a1 += u2[r].y + x2[r].x; a2 += w2[r].x + x2[r].y; }
        if (MODE == 3) { a0 += u4[r].x + u4[r].w; a1 += u4[r].y; a2 += u4[r].z; }
        if (MODE == 4) { a0 += u2[r].x + w2[r].x; a1 += u2[r].y + w2[r].y; }
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
#define RUN(MODE, name) { kaos<MODE><<<(unsigned)(N / 256), 256>>>(m, pos, out, n); CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) kaos<MODE><<<(unsigned)(N / 256), 256>>>(m, pos, out, n); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-58s %.3f ms\n", name, ms / 5); }
    for (int rep = 0; rep < 2; ++rep) {
        RUN(0, "8 x dwordx3 (two per corner pair)");
        RUN(1, "4 x (dwordx4 + dwordx2): the pair as 24 contiguous bytes");
        RUN(2, "4 x (3 x dwordx2)");
        RUN(3, "4 x dwordx4 only");
        RUN(4, "8 x dwordx2 only");
    }
    return 0;
}
