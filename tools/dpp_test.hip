#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);          // wave_shr:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);     // wave_shl:1
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    out[192 + threadIdx.x] = __shfl_up(v, 1);
}
int main() {
    int *d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"wave_shr:1", "wave_shl:1", "row_shr:1", "__shfl_up"};
    for (int r = 0; r < 4; ++r) { printf("%-10s:", names[r]); for (int i = 0; i < 20; ++i) printf(" %d", h[r * 64 + i]); printf(" ... %d %d\n", h[r * 64 + 62], h[r * 64 + 63]); }
    return 0;
}
