// Takes the fused x pass of the Poisson solve (montecosmo_amd/csrc/fftpm.hip, xfused_kernel MODE 0) apart at 512^3:
// what do the barriers, the twiddle loads, the three FFTs, the global loads and the global stores each cost?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I montecosmo_amd/csrc -o tools/xpass_bench.bin tools/xpass_bench.hip
#include "fft_dev.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define TWO_PI 6.283185307179586f
enum { F_NOSYNC = 1, F_CONSTTW = 2, F_NOFFT = 4, F_NOLOAD = 8, F_NOSTORE = 16, F_BLOCKED = 32, F_PRETW = 64 };
// F_BLOCKED: spectrum stored [y / 16][x][y % 16][kz] instead of [x][y][kz]: the x stride drops from 1.1 MB to 35 KB (an x line
// then spans 9 two-megabyte pages instead of 285)

template <int N, int ML>
struct CS {
    static constexpr int T = N / 8;
    static constexpr int LINES = (1024 / T) < ML ? (1024 / T) : ML;
    static constexpr int THREADS = T * LINES;
};

// the library's kernel, single-GPU layout, with switches
template <int N, int ML, int FL>
__global__ __launch_bounds__((CS<N, ML>::THREADS)) void xf_kernel(int ny, int nz, int nzh, int nzp, const cf *__restrict__ in,
                                                                 cf *__restrict__ out, int64_t SC, float scale, const cf *__restrict__ W,
                                                                 int never) {
    constexpr int T = CS<N, ML>::T, LINES = CS<N, ML>::LINES;
    constexpr int DBG = FL & 3;
    typedef Tile<N, LINES, true> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int l = threadIdx.x % LINES, u = threadIdx.x / LINES;
    const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned vv = (nb % 8 == 0) ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
    const unsigned by = vv / gridDim.x, bx = vv - by * gridDim.x;
    const int kzi = bx * LINES + l, iy = by;
    const bool ok = kzi < nzh;
    const uint32_t off0 = (uint32_t)iy * nzp + kzi, xs = (uint32_t)ny * nzp;
    uint32_t o1[8];
    float sx[8], L[8];
    const int sy = iy < (ny + 1) / 2 ? iy : iy - ny;
    const float ky = TWO_PI * (float)sy / (float)ny, kz = TWO_PI * (float)kzi / (float)nz;
    const bool special = (kzi == 0) || (kzi == nz / 2);
    const float dkx = TWO_PI / (float)N;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        o1[m] = (FL & F_BLOCKED) ? (((uint32_t)(iy >> 4) * N + x) * 16 + (iy & 15)) * (uint32_t)nzp + kzi : (uint32_t)x * xs + off0;
        const float kx = dkx * (float)(x < N / 2 ? x : x - N);
        const float kk = kx * kx + ky * ky + kz * kz;
        L[m] = kk == 0.f ? 0.f : -scale * __frcp_rn(kk);
        sx[m] = (special && x == N / 2) ? 0.f : kx * L[m];
    }
    TL tile{l};
    cf tw[3];
    if (FL & F_PRETW) fft_twiddles<N>(W, u, tw);
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = (FL & F_NOLOAD) ? mkc((float)(u + m) * scale, (float)l) : (ok ? in[o1[m]] : mkc(0.f, 0.f));
    if (!(FL & F_NOFFT)) { if (FL & F_PRETW) fft_line_tw<N, -1, TL, DBG>(v, lds, tw, u, tile); else fft_line<N, -1, TL, DBG>(v, lds, W, u, tile); }
    cf w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = mkc(sx[m] * v[m].y, -sx[m] * v[m].x);
    if (!(FL & F_NOFFT)) { if (FL & F_PRETW) fft_line_tw<N, +1, TL, DBG>(w, lds, tw, u, tile); else fft_line<N, +1, TL, DBG>(w, lds, W, u, tile); }
    if (ok && (!(FL & F_NOSTORE) || w[0].x == (float)never)) {
#pragma unroll
        for (int m = 0; m < 8; ++m) __builtin_nontemporal_store(w[m], &out[o1[m]]);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = mkc(L[m] * v[m].y, -L[m] * v[m].x);
    if (!(FL & F_NOFFT)) { if (FL & F_PRETW) fft_line_tw<N, +1, TL, DBG>(w, lds, tw, u, tile); else fft_line<N, +1, TL, DBG>(w, lds, W, u, tile); }
    if (ok && (!(FL & F_NOSTORE) || w[0].x == (float)never)) {
        cf *og = out + SC;
#pragma unroll
        for (int m = 0; m < 8; ++m) __builtin_nontemporal_store(w[m], &og[o1[m]]);
    }
}

// Wave-per-line form: the workgroup loads its 64-byte row segments coalesced (line-fastest mapping) into an LDS tile
// [line][point], then every WAVE owns one line (64 lanes x 8 points) and runs the three transforms with exchanges through
// its own LDS line: no workgroup barrier inside the transforms, waves drift apart freely.  Results go back through the tile
// to coalesced stores.  Two barriers per output instead of twelve in all.
template <int N, int SIGN>
__device__ __forceinline__ void fft_wave(cf (&v)[8], cf *line, const cf *__restrict__ W, int u) {
    // N = 512: three radix-8 stages, T = 64 lanes; same index algebra as fft_line with a one-line tile
    constexpr int T = N / 8;
    int P = 1, S = N;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int S2 = S / 8;
        const int K = u / S2, n2 = u - K * S2;
        fft8<SIGN>(v);
        if (S2 > 1) {
            cf w1 = W[n2 * P];
            if (SIGN > 0) w1.y = -w1.y;
            const cf w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
            v[1] = cmul(v[1], w1);
            v[2] = cmul(v[2], w2);
            v[3] = cmul(v[3], w3);
            v[4] = cmul(v[4], w4);
            v[5] = cmul(v[5], cmul(w4, w1));
            v[6] = cmul(v[6], cmul(w4, w2));
            v[7] = cmul(v[7], cmul(w4, w3));
        }
        if (s < 2) {
            const int Snext = S2, S2next = S2 / 8;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = (K + P * k) * S2 + n2;
                line[e + (e / Snext) * S2next] = v[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int Kn = u / S2next, n2n = u - Kn * S2next;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int e = Kn * Snext + n1 * S2next + n2n;
                v[n1] = line[e + (e / Snext) * S2next];
            }
        }
        P *= 8;
        S = S2;
    }
    (void)T;
}

template <int N, int ML, int FL>
__global__ __launch_bounds__((CS<N, ML>::THREADS)) void xw_kernel(int ny, int nz, int nzh, int nzp, const cf *__restrict__ in,
                                                                 cf *__restrict__ out, int64_t SC, float scale, const cf *__restrict__ W,
                                                                 int never) {
    constexpr int T = CS<N, ML>::T, LINES = CS<N, ML>::LINES;     // T = 64 = one wave per line
    constexpr int NP = N + N / 8 + 8 + 1;                           // odd pitch: the transposing accesses spread over banks
    __shared__ cf lds[NP * LINES];
    const unsigned nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned vv = (nb % 8 == 0) ? (b & 7u) * (nb >> 3) + (b >> 3) : b;
    const unsigned by = vv / gridDim.x, bx = vv - by * gridDim.x;
    const int iy = by;
    const uint32_t xs = (uint32_t)ny * nzp;
    // phase 1: coalesced load, line-fastest mapping (thread -> (row x = tid / LINES + ..., line l = tid % LINES))
    {
        const int l = threadIdx.x % LINES, r = threadIdx.x / LINES;
        const int kzi = bx * LINES + l;
        const bool ok = kzi < nzh;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int x = r + T * m;
            const cf val = ok ? in[(uint32_t)x * xs + (uint32_t)iy * nzp + kzi] : mkc(0.f, 0.f);
            lds[l * NP + x + (x >> 6)] = val;     // natural order + one pad per 64 points
        }
    }
    __syncthreads();
    // phase 2: wave w owns line w
    const int wl = threadIdx.x / 64, u = threadIdx.x % 64;
    cf *line = lds + wl * NP;
    const int kzi = bx * LINES + wl;
    const int sy = iy < (ny + 1) / 2 ? iy : iy - ny;
    const float ky = TWO_PI * (float)sy / (float)ny, kz = TWO_PI * (float)kzi / (float)nz;
    const bool special = (kzi == 0) || (kzi == nz / 2);
    const float dkx = TWO_PI / (float)N;
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        v[m] = line[x + (x >> 6)];
    }
    __builtin_amdgcn_wave_barrier();
    fft_wave<N, -1>(v, line, W, u);
    cf w[8];
    float sx[8], L[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        const float kx = dkx * (float)(x < N / 2 ? x : x - N);
        const float kk = kx * kx + ky * ky + kz * kz;
        L[m] = kk == 0.f ? 0.f : -scale * __frcp_rn(kk);
        sx[m] = (special && x == N / 2) ? 0.f : kx * L[m];
        w[m] = mkc(sx[m] * v[m].y, -sx[m] * v[m].x);
    }
    fft_wave<N, +1>(w, line, W, u);
    // A back to the tile (natural order), whole workgroup stores it coalesced
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        line[x + (x >> 6)] = w[m];
    }
    __syncthreads();
    {
        const int l = threadIdx.x % LINES, r = threadIdx.x / LINES;
        const int kz2 = bx * LINES + l;
        if (kz2 < nzh && (!(FL & F_NOSTORE) || iy == never)) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int x = r + T * m;
                __builtin_nontemporal_store(lds[l * NP + x + (x >> 6)], &out[(uint32_t)x * xs + (uint32_t)iy * nzp + kz2]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = mkc(L[m] * v[m].y, -L[m] * v[m].x);
    fft_wave<N, +1>(w, line, W, u);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        line[x + (x >> 6)] = w[m];
    }
    __syncthreads();
    {
        const int l = threadIdx.x % LINES, r = threadIdx.x / LINES;
        const int kz2 = bx * LINES + l;
        if (kz2 < nzh && (!(FL & F_NOSTORE) || iy == never)) {
            cf *og = out + SC;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int x = r + T * m;
                __builtin_nontemporal_store(lds[l * NP + x + (x >> 6)], &og[(uint32_t)x * xs + (uint32_t)iy * nzp + kz2]);
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class K>
static float time_it(K launch, int reps = 20) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main() {
    constexpr int N = 512;
    const int ny = 512, nz = 512, nzh = 257, nzp = 272;
    const int64_t ss = (int64_t)N * ny * nzp;
    cf *in, *out, *W;
    CK(hipMalloc(&in, ss * sizeof(cf)));
    CK(hipMalloc(&out, 2 * ss * sizeof(cf)));
    CK(hipMalloc(&W, N * sizeof(cf)));
    std::vector<float> h(2 * N);
    for (int j = 0; j < N; ++j) {
        h[2 * j] = (float)cos(-2.0 * M_PI * j / N);
        h[2 * j + 1] = (float)sin(-2.0 * M_PI * j / N);
    }
    CK(hipMemcpy(W, h.data(), sizeof(float) * 2 * N, hipMemcpyHostToDevice));
    CK(hipMemset(in, 0, ss * sizeof(cf)));
    const float scale = 1.f / ((float)N * ny * nz);
    const double gb = (double)N * ny * nzh * 8 * 3 / 1e9;
#define RUN(ML, FL, label)                                                                                          \
    {                                                                                                               \
        dim3 grid((nzh + CS<N, ML>::LINES - 1) / CS<N, ML>::LINES, ny);                                             \
        float ms = time_it([&] { xf_kernel<N, ML, FL><<<grid, CS<N, ML>::THREADS>>>(ny, nz, nzh, nzp, in, out, ss, scale, W, -12345); }); \
        printf("%-58s lines %2d  %.4f ms  (%.2f TB/s of 1.6 GB)\n", label, ML, ms, gb / ms);                         \
    }
    RUN(8, 0, "library kernel");
    RUN(16, 0, "library kernel");
    RUN(8, F_PRETW, "twiddles preloaded into registers");
    RUN(16, F_PRETW, "twiddles preloaded into registers");
    RUN(8, F_NOSYNC, "no barriers");
    RUN(8, F_CONSTTW, "constant twiddles (no table loads)");
    RUN(8, F_NOSYNC | F_CONSTTW, "no barriers, constant twiddles");
    RUN(8, F_NOFFT, "no FFTs (load, multiply, 2 stores)");
    RUN(16, F_NOFFT, "no FFTs (load, multiply, 2 stores)");
    RUN(8, F_NOLOAD | F_NOSTORE, "FFTs only (no global memory)");
    RUN(8, F_NOLOAD | F_NOSTORE | F_NOSYNC, "FFTs only, no barriers");
    RUN(8, F_NOLOAD | F_NOSTORE | F_NOSYNC | F_CONSTTW, "FFTs only, no barriers, constant twiddles");
    RUN(8, F_NOLOAD | F_NOSTORE | F_CONSTTW, "FFTs only, constant twiddles");
    RUN(8, F_NOSTORE, "no stores");
    RUN(8, F_NOLOAD, "no loads");
    RUN(8, F_BLOCKED, "library kernel, blocked layout (x stride 35 KB)");
    RUN(16, F_BLOCKED, "library kernel, blocked layout (x stride 35 KB)");
    RUN(8, F_BLOCKED | F_NOFFT, "no FFTs, blocked layout");
    RUN(16, F_BLOCKED | F_NOFFT, "no FFTs, blocked layout");
    RUN(8, F_BLOCKED | F_NOLOAD, "no loads, blocked layout");
#define RUNW(ML, FL, label)                                                                                         \
    {                                                                                                               \
        dim3 grid((nzh + CS<N, ML>::LINES - 1) / CS<N, ML>::LINES, ny);                                             \
        float ms = time_it([&] { xw_kernel<N, ML, FL><<<grid, CS<N, ML>::THREADS>>>(ny, nz, nzh, nzp, in, out, ss, scale, W, -12345); }); \
        printf("%-58s lines %2d  %.4f ms  (%.2f TB/s of 1.6 GB)\n", label, ML, ms, gb / ms);                         \
    }
    RUNW(8, 0, "wave-per-line form");
    RUNW(16, 0, "wave-per-line form");
    RUNW(8, F_NOSTORE, "wave-per-line form, no stores");
    return 0;
}
