// chreshape: reshape a half-spectrum (nx, ny, nz/2+1) to another half-spectrum shape, truncating or zero-padding the
// centred wavevectors so that Hermitian symmetry and mean power are preserved (montecosmo/utils.py:924-1013), and its
// VJP.  Plain (unpadded) complex64 layout on both sides; no plan is involved, only a stream.
//
// Reference order of operations: truncated axes, last axis first, aggregate their Nyquist planes
//   last axis : z[..., s-1]  <- (z + Hsym(z)) / sqrt2 on that plane (Hsym: conj at (-ix, -iy))
//   other axes: z[-s/2]      <- (z[s/2] + z[-s/2]) / sqrt2
// then crop / pad and scale by the real cell-count ratio, then for padded axes, first axis first,
//   other axes: out[-ms/2] /= sqrt2 ; out[ms/2] = out[-ms/2]        last axis: out[..., ms-1] /= sqrt2.
// Here every output element gathers its (at most 8) sources directly; the VJP is a gather too: every input element collects the terms of
// the outputs that list it (chreshape_vjp_gather_kernel).
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

__global__ __launch_bounds__(256) void chreshape_kernel(RS r, const float2 *__restrict__ in, float2 *__restrict__ out) {
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
    float2 acc = make_float2(0.f, 0.f);
    if (valid)
        sources(r, ix, iy, ok, r.scale * fx * fy * fz, [&](int jx, int jy, int k, float w, bool cj) {
            const float2 v = in[((int64_t)jx * r.my + jy) * r.mzc + k];
            acc.x += w * v.x;
            acc.y += cj ? -w * v.y : w * v.y;
        });
    out[idx] = acc;
}

// Outputs (along one full axis) that read input index i: the inverse of axis_src.  At most two (a padded axis splits the input
// Nyquist plane between +ms/2 and -ms/2).  Returns the count.
__device__ __forceinline__ int axis_dst(int i, int ms, int s, int (&o)[2]) {
    const int f = i < ms / 2 ? i : i - ms;      // frequency of the input index (i = ms/2 is -ms/2)
    if (s > ms) {
        if (i == ms / 2) {
            o[0] = s - ms / 2;
            o[1] = ms / 2;
            return 2;
        }
        o[0] = f < 0 ? f + s : f;
        return 1;
    }
    if (f < -s / 2 || f > s / 2 - 1) return 0;
    o[0] = f < 0 ? f + s : f;
    return 1;
}
// candidate outputs along one axis for input index j: through j itself, through its mirror -j (the conjugated Nyquist term of the last
// axis), and through the aggregated plane ms - s/2 that also reads +s/2 on a truncated axis; deduplicated.  A superset: the caller
// keeps a candidate's terms only where the forward enumeration really lists (jx, jy, k).
__device__ __forceinline__ int axis_candidates(int j, int ms, int s, int (&c)[6]) {
    const int nj = j ? ms - j : 0;
    int src[3] = {j, nj, (s < ms && (j == s / 2 || nj == s / 2)) ? ms - s / 2 : -1};
    int n = 0;
    for (int a = 0; a < 3; ++a) {
        if (src[a] < 0) continue;
        int o[2];
        const int m = axis_dst(src[a], ms, s, o);
        for (int q = 0; q < m; ++q) {
            bool dup = false;
            for (int t = 0; t < n; ++t) dup = dup || c[t] == o[q];
            if (!dup) c[n++] = o[q];
        }
    }
    return n;
}

// The adjoint as a GATHER (round 4): every element of the input's cotangent collects its terms itself, in a fixed order -- no atomics,
// no fixed-point accumulators, no maximum pass, no flush: one launch instead of three and two memsets, and bitwise reproducible by
// construction.  Which outputs list a given input is not guessed: the candidates above are a superset, and each candidate runs the
// FORWARD enumeration (`sources`), keeping the terms that name this input -- the adjoint cannot drift from the forward rule.
__global__ __launch_bounds__(256) void chreshape_vjp_gather_kernel(RS r, const float2 *__restrict__ ob, float2 *__restrict__ in_bar) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)r.mx * r.my * r.mzc;
    if (idx >= n) return;
    const int k = (int)(idx % r.mzc);
    const int64_t t = idx / r.mzc;
    const int jy = (int)(t % r.my), jx = (int)(t / r.my);
    float2 acc = make_float2(0.f, 0.f);
    if (k < r.szc) {      // the last axis maps k -> k (cropped above szc - 1, zero-padded above mzc - 1)
        float fz = 1.f;
        if (r.szc > r.mzc && k == r.mzc - 1) fz = R2;
        int cx[6], cy[6];
        const int nx = axis_candidates(jx, r.mx, r.sx, cx), ny = axis_candidates(jy, r.my, r.sy, cy);
        for (int a = 0; a < nx; ++a) {
            int ix;
            float fx;
            if (!axis_src(cx[a], r.mx, r.sx, ix, fx)) continue;
            for (int b = 0; b < ny; ++b) {
                int iy;
                float fy;
                if (!axis_src(cy[b], r.my, r.sy, iy, fy)) continue;
                bool have = false;
                float2 o = make_float2(0.f, 0.f);
                sources(r, ix, iy, k, r.scale * fx * fy * fz, [&](int sx_, int sy_, int sk, float w, bool cj) {
                    if (sx_ != jx || sy_ != jy || sk != k) return;
                    if (!have) {
                        o = ob[((int64_t)cx[a] * r.sy + cy[b]) * r.szc + k];
                        have = true;
                    }
                    acc.x += w * o.x;
                    acc.y += cj ? -w * o.y : w * o.y;
                });
            }
        }
    }
    in_bar[idx] = acc;
}

// ---- rg2cgh / cgh2rg (montecosmo/utils.py:785-921, norm = "backward") ----------------------------------------------
// A real Gaussian tensor (nx, ny, nz) is permuted and reweighted into a complex Hermitian one (nx, ny, nz/2+1) that is
// distributed as rfftn of a real Gaussian tensor.  Source of the real / imaginary part of output mode (i, j, k):
//   0 < k < hz          : re x[i, j, k]                 im  x[i, j, hz + k]
//   k in {0, hz}, 0<j<hy: re x[i, j, k]                 im  x[i, hy + j, k]
//                 j > hy: re x[-i, ny - j, k]           im -x[-i, hy + ny - j, k]        (-i = (nx - i) mod nx)
//     j in {0, hy}, 0<i<hx: re x[i, j, k]               im  x[hx + i, j, k]
//                   i > hx: re x[nx - i, j, k]          im -x[hx + nx - i, j, k]
//                   i in {0, hx}: re sqrt2 x[i, j, k]   im 0
// all times sqrt(M / 2).
struct CghSrc {
    int64_t re, im;   // flat indices into the real tensor (im < 0: none)
    float wre, wim;
};
__device__ __forceinline__ CghSrc cgh_source(int nx, int ny, int nz, int i, int j, int k) {
    const int hx = nx / 2, hy = ny / 2, hz = nz / 2;
    auto at = [&](int a, int b, int c) { return ((int64_t)a * ny + b) * nz + c; };
    CghSrc r;
    r.wre = 1.f;
    r.wim = 1.f;
    if (k > 0 && k < hz) {
        r.re = at(i, j, k);
        r.im = at(i, j, hz + k);
    } else if (j != 0 && j != hy) {
        if (j < hy) {
            r.re = at(i, j, k);
            r.im = at(i, hy + j, k);
        } else {
            const int mi = i ? nx - i : 0;
            r.re = at(mi, ny - j, k);
            r.im = at(mi, hy + ny - j, k);
            r.wim = -1.f;
        }
    } else if (i != 0 && i != hx) {
        if (i < hx) {
            r.re = at(i, j, k);
            r.im = at(hx + i, j, k);
        } else {
            r.re = at(nx - i, j, k);
            r.im = at(hx + nx - i, j, k);
            r.wim = -1.f;
        }
    } else {
        r.re = at(i, j, k);
        r.im = -1;
        r.wre = 1.41421356237309505f;
        r.wim = 0.f;
    }
    return r;
}

// MODE 0: out = rg2cgh(in real).  MODE 1 (VJP): in = cotangent of the complex output, out = cotangent of the real
// tensor (zeroed by the caller, scattered with atomics: face modes share their source with their Hermitian mirror)
template <int MODE>
__global__ __launch_bounds__(256) void rg2cgh_kernel(int nx, int ny, int nz, float scale, const float *__restrict__ in,
                                                     float *__restrict__ out) {
    const int nzc = nz / 2 + 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, n = (int64_t)nx * ny * nzc;
    if (idx >= n) return;
    const int k = (int)(idx % nzc);
    const int64_t t = idx / nzc;
    const int j = (int)(t % ny), i = (int)(t / ny);
    const CghSrc s = cgh_source(nx, ny, nz, i, j, k);
    if (MODE == 0) {
        out[2 * idx] = scale * s.wre * in[s.re];
        out[2 * idx + 1] = s.im >= 0 ? scale * s.wim * in[s.im] : 0.f;
    } else {
        atomicAdd(out + s.re, scale * s.wre * in[2 * idx]);
        if (s.im >= 0) atomicAdd(out + s.im, scale * s.wim * in[2 * idx + 1]);
    }
}

// out = cgh2rg(in complex): every real element takes the value of ONE stored mode (the mirrored one on the faces and
// edges, as the reference's last assignment does); utils.py:839-889.  AMP (norm="amp", utils.py:916-918): both the "real"
// and the "imaginary" element of a mode take the REAL part of `in` there, unsigned and unweighted -- the layout of a
// per-mode amplitude (model.py:1147).
template <bool AMP>
__global__ __launch_bounds__(256) void cgh2rg_kernel(int nx, int ny, int nz, float scale, const float *__restrict__ in,
                                                     float *__restrict__ out) {
    const int hx = nx / 2, hy = ny / 2, hz = nz / 2, nzc = hz + 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, n = (int64_t)nx * ny * nz;
    if (idx >= n) return;
    const int z = (int)(idx % nz);
    const int64_t t = idx / nz;
    const int y = (int)(t % ny), x = (int)(t / ny);
    auto at = [&](int a, int b, int c) { return 2 * (((int64_t)a * ny + b) * nzc + c); };
    float v;
    if (z != 0 && z != hz) {
        v = z < hz ? in[at(x, y, z)] : (AMP ? in[at(x, y, z - hz)] : in[at(x, y, z - hz) + 1]);
    } else if (y != 0 && y != hy) {
        const int mx_ = x ? nx - x : 0;
        v = y < hy ? in[at(mx_, ny - y, z)] : (AMP ? in[at(mx_, ny + hy - y, z)] : -in[at(mx_, ny + hy - y, z) + 1]);
    } else if (x != 0 && x != hx) {
        v = x < hx ? in[at(nx - x, y, z)] : (AMP ? in[at(nx + hx - x, y, z)] : -in[at(nx + hx - x, y, z) + 1]);
    } else {
        v = in[at(x, y, z)] * (AMP ? 1.f : 0.70710678118654752f);
    }
    out[idx] = scale * v;
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
    chreshape_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(r, (const float2 *)in, (float2 *)out);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

int mcpm_chreshape_vjp_c64(void *stream, const float *out_bar, int out_nx, int out_ny, int out_nz, float *in_bar, int in_nx,
                           int in_ny, int in_nz) {
    if (int rc = check(out_bar, in_bar, in_nx, in_ny, in_nz, out_nx, out_ny, out_nz)) return rc;
    const RS r{in_nx, in_ny, in_nz / 2 + 1, out_nx, out_ny, out_nz / 2 + 1,
               (float)(((double)out_nx * out_ny * out_nz) / ((double)in_nx * in_ny * in_nz))};
    const int64_t ni = (int64_t)r.mx * r.my * r.mzc;
    chreshape_vjp_gather_kernel<<<(unsigned)((ni + 255) / 256), 256, 0, (hipStream_t)stream>>>(r, (const float2 *)out_bar, (float2 *)in_bar);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

int mcpm_rg2cgh_f32(void *stream, const float *real, int nx, int ny, int nz, float *spec) {
    if (int rc = check(real, spec, nx, ny, nz, nx, ny, nz)) return rc;
    const int64_t n = (int64_t)nx * ny * (nz / 2 + 1);
    const float scale = (float)sqrt(0.5 * (double)nx * ny * nz);
    rg2cgh_kernel<0><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(nx, ny, nz, scale, real, spec);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

int mcpm_rg2cgh_vjp_f32(void *stream, const float *spec_bar, int nx, int ny, int nz, float *real_bar) {
    if (int rc = check(spec_bar, real_bar, nx, ny, nz, nx, ny, nz)) return rc;
    const int64_t n = (int64_t)nx * ny * (nz / 2 + 1);
    const float scale = (float)sqrt(0.5 * (double)nx * ny * nz);
    if (hipMemsetAsync(real_bar, 0, sizeof(float) * (size_t)nx * ny * nz, (hipStream_t)stream) != hipSuccess) return MCPM_E_HIP;
    rg2cgh_kernel<1><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(nx, ny, nz, scale, spec_bar, real_bar);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

int mcpm_cgh2rg_f32(void *stream, const float *spec, int nx, int ny, int nz, float *real) {
    if (int rc = check(spec, real, nx, ny, nz, nx, ny, nz)) return rc;
    const int64_t n = (int64_t)nx * ny * nz;
    const float scale = (float)sqrt(2.0 / ((double)nx * ny * nz));
    cgh2rg_kernel<false><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(nx, ny, nz, scale, spec, real);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

int mcpm_cgh2rg_amp_f32(void *stream, const float *spec, int nx, int ny, int nz, float *real) {
    if (int rc = check(spec, real, nx, ny, nz, nx, ny, nz)) return rc;
    const int64_t n = (int64_t)nx * ny * nz;
    cgh2rg_kernel<true><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(nx, ny, nz, 1.f, spec, real);
    return hipGetLastError() == hipSuccess ? MCPM_OK : MCPM_E_HIP;
}

}  // extern "C"
