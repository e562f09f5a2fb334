// Column-pass access pattern in isolation: every workgroup copies a block of ROWS row segments of SEG bytes whose rows are S
// bytes apart (the x pass of the Poisson solve reads 512 segments of 64 B at S = 1.1 MB; the y passes 512 segments of 128 B
// at S = 2176 B).  The buffer is [outer][ROWS][S bytes]; same total size for every S.  Prints GB/s (read + write).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/stride_bench.bin tools/stride_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SEG, int ROWS>
__global__ __launch_bounds__(256) void colcopy(const float4 *__restrict__ src, float4 *__restrict__ dst, long S16, int cbs) {
    constexpr int TPR = SEG / 16, RPI = 256 / TPR, IT = ROWS / RPI;   // threads per row segment, rows per iteration
    const int nb = gridDim.x, b = blockIdx.x;
    const int vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;   // contiguous run of column blocks per XCD
    const long o = vb / cbs, cb = vb % cbs;
    const int r0 = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const long base = o * ROWS * S16 + cb * TPR + t;
    float4 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = src[base + (long)(r0 + i * RPI) * S16];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        v[i].x += 1.f;
        dst[base + (long)(r0 + i * RPI) * S16] = v[i];
    }
}

int main(int argc, char **argv) {
    const long total = 512L * 1114112L;   // bytes: 512 rows x 1.1 MB
    float4 *a, *b;
    hipMalloc(&a, total);
    hipMalloc(&b, total);
    hipMemset(a, 0, total);
    hipMemset(b, 0, total);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long strides[] = {2176, 8704, 34816, 139264, 557056, 1114112};
    for (int seg : {64, 128})
        for (long S : strides) {
            const int cbs = (int)(S / seg);
            const long outer = total / (512 * S);
            const unsigned grid = (unsigned)(outer * cbs);
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (seg == 64) colcopy<64, 512><<<grid, 256>>>(a, b, S / 16, cbs);
                else colcopy<128, 512><<<grid, 256>>>(a, b, S / 16, cbs);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("seg %3d B  stride %8ld B  grid %7u : %.3f ms  %.0f GB/s\n", seg, S, grid, best, 2.0 * total / best / 1e6);
        }
    return 0;
}
