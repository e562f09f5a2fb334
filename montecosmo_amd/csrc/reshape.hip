// chreshape: reshape a half-spectrum (nx, ny, nz/2+1) to another half-spectrum shape, truncating or zero-padding the
// centred wavevectors so that Hermitian symmetry and mean power are preserved (montecosmo/utils.py:924-1013), and its
// VJP.  Plain (unpadded) complex64 layout on both sides; no plan is involved, only a stream.
//
// Reference order of operations: truncated axes, last axis first, aggregate their Nyquist planes
//   last axis : z[..., s-1]  <- (z + Hsym(z)) / sqrt2 on that plane (Hsym: conj at (-ix, -iy))
//   other axes: z[-s/2]      <- (z[s/2] + z[-s/2]) / sqrt2
// then crop / pad and scale by the real cell-count ratio, then for padded axes, first axis first,
//   other axes: out[-ms/2] /= sqrt2 ; out[ms/2] = out[-ms/2]        last axis: out[..., ms-1] /= sqrt2.
// Here every output element gathers its (at most 8) sources directly; the VJP scatters the same terms.
#include "mcpm_internal.h"

namespace {

struct RS {
    int mx, my, mzc;  // input  (nx, ny, nz/2+1)
    int sx, sy, szc;  // output
    float scale;
};

#define R2 0.70710678118654752f

// source index (within the input) and weight of output index o along a full (non-halved) axis; false: zero
__device__ __forceinline__ bool axis_src(int o, int ms, int s, int &i, float &fac) {
    fac = 1.f;
    if (s > ms) {  // padded axis: the input Nyquist plane is split between +ms/2 and -ms/2
        const int neg = s - ms / 2, pos = ms / 2;
        if (o == pos || o == neg) {
            o = neg;
            fac = R2;
        }
    }
    const int f = o < s / 2 ? o : o - s;  // frequency index; o = s/2 is -s/2
    if (s > ms && (f < -ms / 2 || f > ms / 2 - 1)) return false;
    i = f < 0 ? f + ms : f;
    return true;
}

// enumerates the input elements one output element depends on: fn(ix, iy, k, weight, conjugated)
template <class F>
__device__ __forceinline__ void sources(const RS &r, int ix, int iy, int k, float w, F fn) {
    const bool ax = r.sx < r.mx && ix == r.mx - r.sx / 2;
    const bool ay = r.sy < r.my && iy == r.my - r.sy / 2;
    const bool az = r.szc < r.mzc && k == r.szc - 1;
    const float wz = az ? R2 : 1.f;
    for (int bx = 0; bx <= (ax ? 1 : 0); ++bx) {
        const int jx = bx ? r.sx / 2 : ix;
        const float wx = ax ? R2 : 1.f;
        for (int by = 0; by <= (ay ? 1 : 0); ++by) {
            const int jy = by ? r.sy / 2 : iy;
            const float wy = ay ? R2 : 1.f;
            fn(jx, jy, k, w * wx * wy * wz, false);
            if (az) fn(jx ? r.mx - jx : 0, jy ? r.my - jy : 0, k, w * wx * wy * wz, true);
        }
    }
}

template <bool ADJOINT>
__global__ __launch_bounds__(256) void chreshape_kernel(RS r, const float2 *__restrict__ in, float2 *__restrict__ out) {
    // forward: in = input spectrum, out = reshaped.  adjoint: in = cotangent of the reshaped spectrum, out = cotangent
    // of the input (zeroed by the caller; contributions are scattered with atomics)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)r.sx * r.sy * r.szc;
    if (idx >= n) return;
    const int ok = (int)(idx % r.szc);
    const int64_t t = idx / r.szc;
    const int oy = (int)(t % r.sy), ox = (int)(t / r.sy);
    int ix, iy;
    float fx, fy, fz = 1.f;
    bool valid = axis_src(ox, r.mx, r.sx, ix, fx) && axis_src(oy, r.my, r.sy, iy, fy);
    if (r.szc > r.mzc) {
        if (ok > r.mzc - 1) valid = false;
        if (ok == r.mzc - 1) fz = R2;
    }
    if (!ADJOINT) {
        float2 acc = make_float2(0.f, 0.f);
        if (valid)
            sources(r, ix, iy, ok, r.scale * fx * fy * fz, [&](int jx, int jy, int k, float w, bool cj) {
                const float2 v = in[((int64_t)jx * r.my + jy) * r.mzc + k];
                acc.x += w * v.x;
                acc.y += cj ? -w * v.y : w * v.y;
            });
        out[idx] = acc;
    } else {
        if (!valid) return;
        const float2 ob = in[idx];
        sources(r, ix, iy, ok, r.scale * fx * fy * fz, [&](int jx, int jy, int k, float w, bool cj) {
            float *dst = reinterpret_cast<float *>(out + ((int64_t)jx * r.my + jy) * r.mzc + k);
            atomicAdd(dst, w * ob.x);
            atomicAdd(dst + 1, cj ? -w * ob.y : w * ob.y);
        });
    }
}

int check(const void *a, const void *b, int mx, int my, int mz, int sx, int sy, int sz) {
    if (!a || !b) return MCPM_E_ARG;
    const int d[6] = {mx, my, mz, sx, sy, sz};
    for (int v : d)
        if (v < 2 || (v & 1)) return MCPM_E_SHAPE;  // the reference assumes even real sizes
    return MCPM_OK;
}

}  // namespace

extern "C" {

int mcpm_chreshape_c64(void *stream, const float *in, int in_nx, int in_ny, int in_nz, float *out, int out_nx, int out_ny,
                       int out_nz) {
    if (int rc = check(in, out, in_nx, in_ny, in_nz, out_nx, out_ny, out_nz)) return rc;
    const RS r{in_nx, in_ny, in_nz / 2 + 1, out_nx, out_ny, out_nz / 2 + 1,
               (float)(((double)out_nx * out_ny * out_nz) / ((double)in_nx * in_ny * in_nz))};
    const int64_t n = (int64_t)r.sx * r.sy * r.szc;
    chreshape_kernel<false><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(r, (const float2 *)in, (float2 *)out);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

int mcpm_chreshape_vjp_c64(void *stream, const float *out_bar, int out_nx, int out_ny, int out_nz, float *in_bar, int in_nx,
                           int in_ny, int in_nz) {
    if (int rc = check(out_bar, in_bar, in_nx, in_ny, in_nz, out_nx, out_ny, out_nz)) return rc;
    const RS r{in_nx, in_ny, in_nz / 2 + 1, out_nx, out_ny, out_nz / 2 + 1,
               (float)(((double)out_nx * out_ny * out_nz) / ((double)in_nx * in_ny * in_nz))};
    const int64_t n = (int64_t)r.sx * r.sy * r.szc, ni = (int64_t)r.mx * r.my * r.mzc;
    if (hipMemsetAsync(in_bar, 0, sizeof(float2) * ni, (hipStream_t)stream) != hipSuccess) return MCPM_E_HIP;
    chreshape_kernel<true><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(r, (const float2 *)out_bar, (float2 *)in_bar);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

}  // extern "C"
