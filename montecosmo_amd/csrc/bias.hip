// Lagrangian bias expansion (montecosmo/bricks.py:327-443, png_type = None) and its VJP:
//   weights = 1 + b1 d + b2 (d^2 - <d^2>)/2 + bs2 (s2 - 2/3 <d^2>) + b3 (d^3 - 3 <d^2> d)/6 + bds2 d (s2 - 2/3 <d^2>)
//             + bs3 s3 + bn2 lap(d),          dvel = bnpar grad(d) g
// with d = delta_L(q) g, s2 = shear^2(q) g^2, s3 = 3 det(shear)(q) g^3 read at the (Lagrangian) particle positions
// and g = a2g(a) a scalar or one value per particle (light cone).
// Mesh side (`mcpm_bias_fields*`): lin_mesh -> 7 real meshes {delta, shear^2, 3 det shear, laplacian, grad x3} through
// 10 C2R transforms (wavevectors in h/Mpc: k_cell * kphys[axis]); particle side (`mcpm_bias_weights*`): the
// per-particle polynomial, its moment <d^2> and the cotangents, reductions in f64.  The reads / their adjoints in
// between are mcpm_read_f32 / mcpm_paint_f32.
#include "mcpm_internal.h"
#include "reduce_dev.h"

#define TWO_PI 6.283185307179586f

namespace {

__device__ __forceinline__ float kfreq(int i, int n) {
    int s = (i < (n + 1) / 2) ? i : i - n;
    return TWO_PI * (float)s / (float)n;
}

struct BMode {
    float k[3];     // physical wavevector
    bool nyq[3], special;
    float zw;
};
__device__ __forceinline__ BMode bdecode(const Geom &g, float kx, float ky, float kz, uint32_t idx) {
    BMode m;
    const int iz = idx % (uint32_t)g.nzh;
    const uint32_t r = idx / (uint32_t)g.nzh;
    const int iy = r % (uint32_t)g.ny, ix = r / (uint32_t)g.ny;
    m.k[0] = kfreq(ix, g.nx) * kx;
    m.k[1] = kfreq(iy, g.ny) * ky;
    m.k[2] = TWO_PI * (float)iz / (float)g.nz * kz;
    m.nyq[0] = !(g.nx & 1) && ix == g.nx / 2;
    m.nyq[1] = !(g.ny & 1) && iy == g.ny / 2;
    m.nyq[2] = iz == g.nz / 2;
    m.special = iz == 0 || m.nyq[2];
    m.zw = m.special ? 1.f : 2.f;
    return m;
}

// the 10 multipliers y_s = irfftn(m_s X): real part re[s] (s = 0..6) or imaginary part im (s = 7..9)
//  0: 1   1: kx kx/k2   2: ky ky/k2   3: kx ky/k2   4: kx kz/k2   5: ky kz/k2   6: -k2   7..9: i k_c
// `herm`: apply numpy irfftn's projection on the kz = 0 / Nyquist planes (odd number of Nyquist factors -> 0)
__device__ __forceinline__ void multipliers(const BMode &m, bool herm, float (&re)[7], float (&im)[3]) {
    const float k2 = m.k[0] * m.k[0] + m.k[1] * m.k[1] + m.k[2] * m.k[2];
    const float ik2 = k2 == 0.f ? 0.f : 1.f / k2;
    const bool pr = herm && m.special;
    re[0] = 1.f;
    re[1] = m.k[0] * m.k[0] * ik2;
    re[2] = m.k[1] * m.k[1] * ik2;
    re[3] = (pr && m.nyq[0] != m.nyq[1]) ? 0.f : m.k[0] * m.k[1] * ik2;
    re[4] = (pr && m.nyq[0] != m.nyq[2]) ? 0.f : m.k[0] * m.k[2] * ik2;
    re[5] = (pr && m.nyq[1] != m.nyq[2]) ? 0.f : m.k[1] * m.k[2] * ik2;
    re[6] = -k2;
#pragma unroll
    for (int c = 0; c < 3; ++c) im[c] = (pr && m.nyq[c]) ? 0.f : m.k[c];
}

// GROUP 0: spectra 0..5 (6 outputs), GROUP 1: spectra 6..9 (4 outputs); out[s] = scale * m_s * in
template <int GROUP>
__global__ __launch_bounds__(256) void bias_spectra_kernel(Geom g, float kx, float ky, float kz, float scale,
                                                           const float2 *__restrict__ in, float2 *__restrict__ out, int64_t Mh) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    const BMode m = bdecode(g, kx, ky, kz, idx);
    float re[7], im[3];
    multipliers(m, true, re, im);
    const float2 v = in[idx];
    if (GROUP == 0) {
#pragma unroll
        for (int s = 0; s < 6; ++s) out[s * Mh + idx] = make_float2(scale * re[s] * v.x, scale * re[s] * v.y);
    } else {
        out[idx] = make_float2(scale * re[6] * v.x, scale * re[6] * v.y);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(1 + c) * Mh + idx] = make_float2(-scale * im[c] * v.y, scale * im[c] * v.x);  // (a+ib)(i s)
    }
}

// out (+)= scale * zw * sum_s conj(m_s) in[s]   (adjoint of irfftn o multiply; un-projected multipliers)
template <int GROUP>
__global__ __launch_bounds__(256) void bias_spectra_vjp_kernel(Geom g, float kx, float ky, float kz, float scale,
                                                               const float2 *__restrict__ in, float2 *__restrict__ out,
                                                               int64_t Mh, int accumulate) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    const BMode m = bdecode(g, kx, ky, kz, idx);
    float re[7], im[3];
    multipliers(m, false, re, im);
    float ar = 0.f, ai = 0.f;
    if (GROUP == 0) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const float2 v = in[s * Mh + idx];
            ar += re[s] * v.x;
            ai += re[s] * v.y;
        }
    } else {
        const float2 v = in[idx];
        ar += re[6] * v.x;
        ai += re[6] * v.y;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float2 w = in[(1 + c) * Mh + idx];
            ar += im[c] * w.y;   // (a+ib)(-i s)
            ai += -im[c] * w.x;
        }
    }
    ar *= scale * m.zw;
    ai *= scale * m.zw;
    if (accumulate) {
        const float2 o = out[idx];
        ar += o.x;
        ai += o.y;
    }
    out[idx] = make_float2(ar, ai);
}

// r6 = {delta, h00, h11, h01, h02, h12} (h_ij = d_i d_j laplace^-1 delta) -> shear^2, 3 det(shear)
__global__ __launch_bounds__(256) void shear_combine_kernel(const float *__restrict__ r6, int64_t M, float *__restrict__ s2,
                                                            float *__restrict__ s3) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float dl = r6[i], t = dl * (1.f / 3.f);
    const float a = r6[M + i] - t, b = r6[2 * M + i] - t, c = -(a + b);
    const float d = r6[3 * M + i], e = r6[4 * M + i], f = r6[5 * M + i];
    s2[i] = a * a + b * b + c * c + 2.f * (d * d + e * e + f * f);
    s3[i] = 3.f * (a * (b * c - f * f) - d * (d * c - e * f) + e * (d * f - b * e));
}

// src (forward values; may be r6 itself) -> r6 = cotangents of {delta, h00, h11, h01, h02, h12} given those of delta, shear^2, shear^3
__global__ __launch_bounds__(256) void shear_combine_vjp_kernel(const float *src, float *r6, int64_t M, const float *__restrict__ db,
                                                                const float *__restrict__ s2b, const float *__restrict__ s3b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float dl = src[i], t = dl * (1.f / 3.f);
    const float a = src[M + i] - t, b = src[2 * M + i] - t, c = -(a + b);
    const float d = src[3 * M + i], e = src[4 * M + i], f = src[5 * M + i];
    const float w2 = s2b[i], w3 = 3.f * s3b[i];
    const float ca = b * c - f * f, cb = a * c - e * e, cc = a * b - d * d;
    const float ab = w2 * (2.f * a - 2.f * c) + w3 * (ca - cc), bb = w2 * (2.f * b - 2.f * c) + w3 * (cb - cc);
    r6[i] = db[i] - (ab + bb) * (1.f / 3.f);
    r6[M + i] = ab;
    r6[2 * M + i] = bb;
    r6[3 * M + i] = w2 * 4.f * d + w3 * 2.f * (e * f - d * c);
    r6[4 * M + i] = w2 * 4.f * e + w3 * 2.f * (d * f - b * e);
    r6[5 * M + i] = w2 * 4.f * f + w3 * 2.f * (d * e - a * f);
}

// ---- particle side ------------------------------------------------------------------------------------------
struct Bias8 {
    float b1, b2, bs2, b3, bds2, bs3, bn2, bnpar;
};

__global__ __launch_bounds__(256) void bias_moment_kernel(const float *__restrict__ dr, const float *__restrict__ gp, float gs,
                                                          int64_t n, double *part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[1] = {0.};
    if (i < n) {
        const float d = dr[i] * (gp ? gp[i] : gs);
        v[0] = (double)d * (double)d;
    }
    block_partial<1>(v, part, gridDim.x, blockIdx.x);
}

__global__ __launch_bounds__(256) void bias_weights_kernel(const float *__restrict__ dr, const float *__restrict__ s2r,
                                                           const float *__restrict__ s3r, const float *__restrict__ lr,
                                                           const float *__restrict__ gr, int64_t ges, int64_t gcs,
                                                           const float *__restrict__ gp, float gs,
                                                           Bias8 B, const double *__restrict__ sigma2p, int64_t n,
                                                           float *__restrict__ w, float *__restrict__ dvel) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = gp ? gp[i] : gs, sig = (float)sigma2p[0];
    const float d = dr[i] * g, s2 = s2r[i] * g * g - (2.f / 3.f) * sig, s3 = s3r[i] * g * g * g, l = lr[i] * g;
    float wt = 1.f + B.b1 * d;
    wt += B.b2 * (d * d - sig) * 0.5f;
    wt += B.bs2 * s2;
    wt += B.b3 * (d * d * d - 3.f * sig * d) * (1.f / 6.f);
    wt += B.bds2 * d * s2;
    wt += B.bs3 * s3;
    wt += B.bn2 * l;
    w[i] = wt;
    const float c = B.bnpar * g;
    dvel[3 * i] = c * gr[ges * i];                 // gr[i][c] at gr + i * ges + c * gcs (particle-major or mesh-major)
    dvel[3 * i + 1] = c * gr[ges * i + gcs];
    dvel[3 * i + 2] = c * gr[ges * i + 2 * gcs];
}

// pass 1 of the VJP: the 8 bias cotangents and sigma2_bar (per-workgroup partials, rows 0..8; det_fold_kernel adds them up)
__global__ __launch_bounds__(256) void bias_vjp_reduce_kernel(const float *__restrict__ dr, const float *__restrict__ s2r,
                                                              const float *__restrict__ s3r, const float *__restrict__ lr,
                                                              const float *__restrict__ gr, int64_t ges, int64_t gcs,
                                                              const float *__restrict__ gp, float gs,
                                                              Bias8 B, const double *__restrict__ sigma2p,
                                                              const float *__restrict__ wb, const float *__restrict__ vb,
                                                              int64_t n, double *part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.};
    if (i < n) {
        const float g = gp ? gp[i] : gs, sig = (float)sigma2p[0];
        const float d = dr[i] * g, s2 = s2r[i] * g * g - (2.f / 3.f) * sig, s3 = s3r[i] * g * g * g, l = lr[i] * g;
        const float w = wb[i];
        v[0] = (double)(w * d);
        v[1] = (double)(w * (d * d - sig) * 0.5f);
        v[2] = (double)(w * s2);
        v[3] = (double)(w * (d * d * d - 3.f * sig * d) * (1.f / 6.f));
        v[4] = (double)(w * d * s2);
        v[5] = (double)(w * s3);
        v[6] = (double)(w * l);
        v[7] = (double)(g * (vb[3 * i] * gr[ges * i] + vb[3 * i + 1] * gr[ges * i + gcs] + vb[3 * i + 2] * gr[ges * i + 2 * gcs]));
        const float dw_ds2 = B.bs2 + B.bds2 * d;
        v[8] = (double)(w * (-0.5f * B.b2 - 0.5f * B.b3 * d - (2.f / 3.f) * dw_ds2));
    }
    block_partial<9>(v, part, gridDim.x, blockIdx.x);
}

// pass 2: per-particle cotangents of the raw reads and of g (g_bar per particle, and its per-workgroup partial sums)
__global__ __launch_bounds__(256) void bias_vjp_particles_kernel(const float *__restrict__ dr, const float *__restrict__ s2r,
                                                                 const float *__restrict__ s3r, const float *__restrict__ lr,
                                                                 const float *__restrict__ gr, int64_t ges, int64_t gcs,
                                                                 const float *__restrict__ gp, float gs,
                                                                 Bias8 B, const double *__restrict__ sigma2p,
                                                                 const double *__restrict__ sigbarp, const float *__restrict__ wb,
                                                                 const float *__restrict__ vb, int64_t n,
                                                                 float *__restrict__ drb, float *__restrict__ s2rb,
                                                                 float *__restrict__ s3rb, float *__restrict__ lrb,
                                                                 float *__restrict__ grb, float *__restrict__ gbar, double *part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[1] = {0.};
    if (i < n) {
        const float g = gp ? gp[i] : gs, sig = (float)sigma2p[0], sigbar = (float)sigbarp[0];
        const float draw = dr[i], s2raw = s2r[i], s3raw = s3r[i], lraw = lr[i];
        const float d = draw * g, s2 = s2raw * g * g - (2.f / 3.f) * sig;
        const float w = wb[i];
        const float dw_ds2 = B.bs2 + B.bds2 * d;
        const float dbar = w * (B.b1 + B.b2 * d + B.b3 * (d * d - sig) * 0.5f + B.bds2 * s2) + sigbar * 2.f * d / (float)n;
        drb[i] = dbar * g;
        s2rb[i] = w * dw_ds2 * g * g;
        s3rb[i] = w * B.bs3 * g * g * g;
        lrb[i] = w * B.bn2 * g;
        const float c = B.bnpar * g;
        const float v0 = vb[3 * i], v1 = vb[3 * i + 1], v2 = vb[3 * i + 2];
        grb[ges * i] = c * v0;
        grb[ges * i + gcs] = c * v1;
        grb[ges * i + 2 * gcs] = c * v2;
        const float gb = dbar * draw + w * dw_ds2 * 2.f * g * s2raw + w * B.bs3 * 3.f * g * g * s3raw + w * B.bn2 * lraw +
                         B.bnpar * (v0 * gr[ges * i] + v1 * gr[ges * i + gcs] + v2 * gr[ges * i + 2 * gcs]);
        if (gbar) gbar[i] = gb;
        v[0] = (double)gb;
    }
    block_partial<1>(v, part, gridDim.x, blockIdx.x);
}

// jnp.interp(x, xp, fp, left=0, right=0) on float64 device tables
__device__ __forceinline__ double interp_zero(double x, const double *xp, const double *fp, int n) {
    if (x < xp[0] || x > xp[n - 1]) return 0.;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    return fp[lo] + (fp[hi] - fp[lo]) / (xp[hi] - xp[lo]) * (x - xp[lo]);
}

// out = in * sqrt(amp * P(|k|)), P linearly interpolated from (ks, pows), zero outside the table (bricks.py:83-100, :149-154)
__global__ __launch_bounds__(256) void power_mult_kernel(Geom g, float kx, float ky, float kz, double amp,
                                                         const double *__restrict__ ks, const double *__restrict__ pows, int nt,
                                                         const float2 *__restrict__ in, float2 *__restrict__ out, int64_t Mh) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    const BMode m = bdecode(g, kx, ky, kz, idx);
    const double k = sqrt((double)m.k[0] * m.k[0] + (double)m.k[1] * m.k[1] + (double)m.k[2] * m.k[2]);
    const float t = (float)sqrt(amp * interp_zero(k, ks, pows, nt));
    const float2 v = in[idx];
    out[idx] = make_float2(t * v.x, t * v.y);
}

// np.interp / jnp.interp with clamped ends (what the growth and distance look-ups use, nbody.py:748-804, :862-884)
__global__ __launch_bounds__(256) void interp_kernel(const float *__restrict__ x, int64_t n, const double *__restrict__ xp,
                                                     const double *__restrict__ fp, int nt, float scale, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = (double)x[i];
    double r;
    if (v <= xp[0]) r = fp[0];
    else if (v >= xp[nt - 1]) r = fp[nt - 1];
    else {
        int lo = 0, hi = nt - 1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (xp[mid] <= v) lo = mid; else hi = mid;
        }
        r = fp[lo] + (fp[hi] - fp[lo]) / (xp[hi] - xp[lo]) * (v - xp[lo]);
    }
    out[i] = scale * (float)r;
}

// light-cone LPT (nbody.py:652-666 with a of shape (N,1)): dpos = g F1 - g2 F2, vel = F1 - c F2, per-particle (g, g2, c)
__global__ __launch_bounds__(256) void lpt_combine_kernel(const float *__restrict__ F1, const float *__restrict__ F2,
                                                          const float *__restrict__ gt, int64_t n, float *__restrict__ dpos,
                                                          float *__restrict__ vel) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = gt[3 * i], g2 = gt[3 * i + 1], c = gt[3 * i + 2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float f1 = F1[3 * i + a], f2 = F2 ? F2[3 * i + a] : 0.f;
        dpos[3 * i + a] = g * f1 - g2 * f2;
        vel[3 * i + a] = f1 - c * f2;
    }
}

// in place: (xb, vb) = cotangents of (dpos, vel) -> cotangents of (F2, F1) [so that they feed mcpm_lpt_vjp_f32 called
// with (g, g2, c) = (0, -1, 0)]; gtb = per-particle cotangents of (g, g2, c)
__global__ __launch_bounds__(256) void lpt_combine_vjp_kernel(const float *__restrict__ F1, const float *__restrict__ F2,
                                                              const float *__restrict__ gt, int64_t n, float *xb, float *vb,
                                                              float *__restrict__ gtb) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = gt[3 * i], g2 = gt[3 * i + 1], c = gt[3 * i + 2];
    float gb = 0.f, g2b = 0.f, cb = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = xb[3 * i + a], v = vb[3 * i + a];
        const float f1 = F1[3 * i + a], f2 = F2 ? F2[3 * i + a] : 0.f;
        gb += x * f1;
        g2b -= x * f2;
        cb -= v * f2;
        xb[3 * i + a] = -g2 * x - c * v;   // F2_bar
        vb[3 * i + a] = g * x + v;         // F1_bar
    }
    gtb[3 * i] = gb;
    gtb[3 * i + 1] = g2b;
    gtb[3 * i + 2] = cb;
}

int fields_group(mcpm_plan *p, const float *lin_mesh, float kx, float ky, float kz, int group, float *spec, float *real) {
    const unsigned nb = (unsigned)((p->Mh + 255) / 256);
    const float scale = 1.f / (float)p->M;
    if (group == 0)
        bias_spectra_kernel<0><<<nb, 256, 0, p->stream>>>(p->g, kx, ky, kz, scale, (const float2 *)lin_mesh, (float2 *)spec, p->Mh);
    else
        bias_spectra_kernel<1><<<nb, 256, 0, p->stream>>>(p->g, kx, ky, kz, scale, (const float2 *)lin_mesh, (float2 *)spec, p->Mh);
    MCPM_LAUNCH_CHECK(p, "bias_spectra_kernel");
    return mcpm_fft_c2r(p, spec, real, group == 0 ? 6 : 4);
}

}  // namespace

extern "C" {

// lin_mesh (plain half-spectrum) -> fields7 = {delta, shear^2, 3 det shear, laplacian delta, grad_x, grad_y, grad_z} (7 real
// meshes, M apart).  kphys = mesh_shape / box_size per axis (bricks.py:352: wavevectors in h/Mpc).
// hess6 (may be NULL): 6 M floats that receive delta and the five Hessian meshes the shear is built from, for mcpm_bias_fields_vjp_saved_f32
// (the adjoint then recomputes nothing: six transforms less per gradient, 400 MB at 256^3)
int mcpm_bias_fields_save_f32(mcpm_plan *p, const float *lin_mesh, float kpx, float kpy, float kpz, float *fields7, float *hess6) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, lin_mesh && fields7, MCPM_E_ARG, "mcpm_bias_fields_f32: null buffer");
    MCPM_REQUIRE(p, !p->g.xslab, MCPM_E_UNSUPPORTED, "mcpm_bias_fields_f32: not slab-decomposed");
    const int64_t M = p->M;
    float *spec = p->spec, *r6 = hess6 ? hess6 : p->fmesh;  // scratch: 6 plain spectra, 6 of the 9 real meshes
    MCPM_TRY(fields_group(p, lin_mesh, kpx, kpy, kpz, 0, spec, r6));
    {
        StageTimer st_(p, ST_LPT, 32.0 * M);
        shear_combine_kernel<<<(unsigned)((M + 255) / 256), 256, 0, p->stream>>>(r6, M, fields7 + M, fields7 + 2 * M);
        MCPM_LAUNCH_CHECK(p, "shear_combine_kernel");
    }
    MCPM_HIP(p, hipMemcpyAsync(fields7, r6, sizeof(float) * M, hipMemcpyDeviceToDevice, p->stream));
    MCPM_TRY(fields_group(p, lin_mesh, kpx, kpy, kpz, 1, spec, fields7 + 3 * M));
    return MCPM_OK;
}

int mcpm_bias_fields_f32(mcpm_plan *p, const float *lin_mesh, float kpx, float kpy, float kpz, float *fields7) {
    return mcpm_bias_fields_save_f32(p, lin_mesh, kpx, kpy, kpz, fields7, nullptr);
}

// cotangents of the 7 fields -> cotangent of lin_mesh (real-pair convention, irfftn multiplicity weights); hess6: what
// mcpm_bias_fields_save_f32 left (read only), or NULL: delta and the Hessian meshes are recomputed from lin_mesh
static int bias_fields_vjp(mcpm_plan *p, const float *lin_mesh, float kpx, float kpy, float kpz, const float *hess6, const float *fields7_bar,
                           float *lin_mesh_bar) {
    const int64_t M = p->M, Mh = p->Mh;
    const unsigned nb = (unsigned)((Mh + 255) / 256);
    const float scale = 1.f / (float)M;
    float *spec = p->spec, *r6 = p->fmesh;
    if (!hess6) MCPM_TRY(fields_group(p, lin_mesh, kpx, kpy, kpz, 0, spec, r6));   // recompute delta and the Hessian meshes
    shear_combine_vjp_kernel<<<(unsigned)((M + 255) / 256), 256, 0, p->stream>>>(hess6 ? hess6 : r6, r6, M, fields7_bar, fields7_bar + M, fields7_bar + 2 * M);
    MCPM_LAUNCH_CHECK(p, "shear_combine_vjp_kernel");
    MCPM_TRY(mcpm_fft_r2c(p, r6, spec, 6));
    bias_spectra_vjp_kernel<0><<<nb, 256, 0, p->stream>>>(p->g, kpx, kpy, kpz, scale, (const float2 *)spec, (float2 *)lin_mesh_bar, Mh, 0);
    MCPM_LAUNCH_CHECK(p, "bias_spectra_vjp_kernel");
    // rocFFT may overwrite a C2R / R2C input: work on a copy of the caller's cotangent meshes
    MCPM_HIP(p, hipMemcpyAsync(r6, fields7_bar + 3 * M, sizeof(float) * 4 * M, hipMemcpyDeviceToDevice, p->stream));
    MCPM_TRY(mcpm_fft_r2c(p, r6, spec, 4));
    bias_spectra_vjp_kernel<1><<<nb, 256, 0, p->stream>>>(p->g, kpx, kpy, kpz, scale, (const float2 *)spec, (float2 *)lin_mesh_bar, Mh, 1);
    MCPM_LAUNCH_CHECK(p, "bias_spectra_vjp_kernel");
    return MCPM_OK;
}

int mcpm_bias_fields_vjp_f32(mcpm_plan *p, const float *lin_mesh, float kpx, float kpy, float kpz, const float *fields7_bar,
                             float *lin_mesh_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, lin_mesh && fields7_bar && lin_mesh_bar, MCPM_E_ARG, "mcpm_bias_fields_vjp_f32: null buffer");
    MCPM_REQUIRE(p, !p->g.xslab, MCPM_E_UNSUPPORTED, "mcpm_bias_fields_vjp_f32: not slab-decomposed");
    return bias_fields_vjp(p, lin_mesh, kpx, kpy, kpz, nullptr, fields7_bar, lin_mesh_bar);
}

int mcpm_bias_fields_vjp_saved_f32(mcpm_plan *p, float kpx, float kpy, float kpz, const float *hess6, const float *fields7_bar,
                                   float *lin_mesh_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, hess6 && fields7_bar && lin_mesh_bar, MCPM_E_ARG, "mcpm_bias_fields_vjp_saved_f32: null buffer");
    MCPM_REQUIRE(p, !p->g.xslab, MCPM_E_UNSUPPORTED, "mcpm_bias_fields_vjp_saved_f32: not slab-decomposed");
    return bias_fields_vjp(p, nullptr, kpx, kpy, kpz, hess6, fields7_bar, lin_mesh_bar);
}

// reads (raw values of the 7 fields at the particles: dr, s2r, s3r, lr (n each), gr (n,3)) -> weights (n), dvel (n,3).
// growth: one value per particle, or NULL and growth_scalar.  bias8 = {b1, b2, bs2, b3, bds2, bs3, bn2, bnpar} (host).
// sigma2_out (device double, may be NULL) receives <d^2>.
int mcpm_bias_weights_f32(mcpm_plan *p, int64_t n, const float *dr, const float *s2r, const float *s3r, const float *lr,
                          const float *gr, int64_t gr_cstride, const float *growth, float growth_scalar, const float *bias8,
                          float *weights, float *dvel, double *sigma2_out) {
    const int64_t ges = gr_cstride ? 1 : 3, gcs = gr_cstride ? gr_cstride : 1;
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, n > 0 && dr && s2r && s3r && lr && gr && bias8 && weights && dvel, MCPM_E_ARG, "mcpm_bias_weights_f32: bad argument");
    const Bias8 B{bias8[0], bias8[1], bias8[2], bias8[3], bias8[4], bias8[5], bias8[6], bias8[7]};
    double *sig = p->reduce, *P, *Q;
    unsigned *ticket, R;
    const unsigned nb = (unsigned)((n + 255) / 256);
    StageTimer st_(p, ST_LPT, 60.0 * n);
    MCPM_TRY(mcpm_det_scratch(p, 1, nb, &P, &Q, &ticket, &R));
    bias_moment_kernel<<<nb, 256, 0, p->stream>>>(dr, growth, growth_scalar, n, P);
    det_fold_kernel<<<R, 256, 0, p->stream>>>(P, nb, 1, Q, ticket, 1.0 / (double)n, det_outs(sig));
    bias_weights_kernel<<<nb, 256, 0, p->stream>>>(dr, s2r, s3r, lr, gr, ges, gcs, growth, growth_scalar, B, sig, n, weights, dvel);
    MCPM_LAUNCH_CHECK(p, "bias_weights_kernel");
    if (sigma2_out) MCPM_HIP(p, hipMemcpyAsync(sigma2_out, sig, sizeof(double), hipMemcpyDeviceToDevice, p->stream));
    return MCPM_OK;
}

// VJP: (weights_bar (n), dvel_bar (n,3)) -> cotangents of the raw reads (drb, s2rb, s3rb, lrb (n), grb (n,3)), of growth
// (per particle into growth_bar if not NULL) and the scalars: scalars_out (device, 10 doubles) = 8 bias cotangents,
// the summed growth cotangent, <d^2>.
int mcpm_bias_weights_vjp_f32(mcpm_plan *p, int64_t n, const float *dr, const float *s2r, const float *s3r, const float *lr,
                              const float *gr, int64_t gr_cstride, const float *growth, float growth_scalar, const float *bias8,
                              const float *weights_bar, const float *dvel_bar, float *drb, float *s2rb, float *s3rb, float *lrb,
                              float *grb, float *growth_bar, double *scalars_out) {
    const int64_t ges = gr_cstride ? 1 : 3, gcs = gr_cstride ? gr_cstride : 1;
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, n > 0 && dr && s2r && s3r && lr && gr && bias8 && weights_bar && dvel_bar && drb && s2rb && s3rb && lrb && grb && scalars_out,
                 MCPM_E_ARG, "mcpm_bias_weights_vjp_f32: bad argument");
    const Bias8 B{bias8[0], bias8[1], bias8[2], bias8[3], bias8[4], bias8[5], bias8[6], bias8[7]};
    double *P, *Q;
    unsigned *ticket, R;
    const unsigned nb = (unsigned)((n + 255) / 256);
    StageTimer st_(p, ST_LPT, 120.0 * n);
    MCPM_TRY(mcpm_det_scratch(p, 9, nb, &P, &Q, &ticket, &R));      // 9 rows of per-workgroup partials, summed in a fixed order
    bias_moment_kernel<<<nb, 256, 0, p->stream>>>(dr, growth, growth_scalar, n, P);
    det_fold_kernel<<<R, 256, 0, p->stream>>>(P, nb, 1, Q, ticket, 1.0 / (double)n, det_outs(scalars_out + 9));
    bias_vjp_reduce_kernel<<<nb, 256, 0, p->stream>>>(dr, s2r, s3r, lr, gr, ges, gcs, growth, growth_scalar, B, scalars_out + 9, weights_bar,
                                                      dvel_bar, n, P);
    DetOuts o9{};
    for (int k = 0; k < 9; ++k) o9.p[k] = scalars_out + k;      // [8] = sigma2_bar for now
    det_fold_kernel<<<R, 256, 0, p->stream>>>(P, nb, 9, Q, ticket, 1.0, o9);
    bias_vjp_particles_kernel<<<nb, 256, 0, p->stream>>>(dr, s2r, s3r, lr, gr, ges, gcs, growth, growth_scalar, B, scalars_out + 9, scalars_out + 8,
                                                         weights_bar, dvel_bar, n, drb, s2rb, s3rb, lrb, grb, growth_bar, P);
    det_fold_kernel<<<R, 256, 0, p->stream>>>(P, nb, 1, Q, ticket, 1.0, det_outs(scalars_out + 8));   // summed growth cotangent
    MCPM_LAUNCH_CHECK(p, "bias_vjp_particles_kernel");
    return MCPM_OK;
}

// white2lin / lin2white multiplier (bricks.py:149-161): out = in * sqrt(amp * P(|k|)) on the plan's half-spectrum, |k| in h/Mpc
// (kphys = mesh_shape / box_size), P from the DEVICE float64 table (ks ascending, pows), zero outside it; amp = sigma8^2
// for a table normalised to sigma8 = 1.  The multiplier is real, so the same call is its own adjoint.
int mcpm_power_mult_f32(mcpm_plan *p, const float *in, float kpx, float kpy, float kpz, double amp, const double *ks,
                        const double *pows, int ntab, float *out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out && ks && pows && ntab >= 2, MCPM_E_ARG, "mcpm_power_mult_f32: bad argument");
    StageTimer st_(p, ST_KSPACE, 16.0 * p->Mh);
    power_mult_kernel<<<(unsigned)((p->Mh + 255) / 256), 256, 0, p->stream>>>(p->g, kpx, kpy, kpz, amp, ks, pows, ntab,
                                                                             (const float2 *)in, (float2 *)out, p->Mh);
    MCPM_LAUNCH_CHECK(p, "power_mult_kernel");
    return MCPM_OK;
}

// Light-cone LPT combination: F1, F2 (n,3) first / second order forces at the particles (F2 may be NULL), gtab (n,3) =
// per-particle (a2g, a2g2, a2dg2dg)(a_i) -> dpos = g F1 - g2 F2, vel = F1 - dg2dg F2 (nbody.py:652-666).
int mcpm_lpt_combine_f32(mcpm_plan *p, const float *F1, const float *F2, const float *gtab, int64_t n, float *dpos, float *vel) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, F1 && gtab && dpos && vel && n > 0, MCPM_E_ARG, "mcpm_lpt_combine_f32: bad argument");
    StageTimer st_(p, ST_LPT, 60.0 * n);
    lpt_combine_kernel<<<(unsigned)((n + 255) / 256), 256, 0, p->stream>>>(F1, F2, gtab, n, dpos, vel);
    MCPM_LAUNCH_CHECK(p, "lpt_combine_kernel");
    return MCPM_OK;
}

// VJP, in place: on entry (xb, vb) = cotangents of (dpos, vel); on exit xb = cotangent of F2, vb = cotangent of F1 (feed
// them to mcpm_lpt_vjp_f32 with (g, g2, dg2dg) = (0, -1, 0)); gtab_bar (n,3) = per-particle growth cotangents.
int mcpm_lpt_combine_vjp_f32(mcpm_plan *p, const float *F1, const float *F2, const float *gtab, int64_t n, float *xb, float *vb,
                             float *gtab_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, F1 && gtab && xb && vb && gtab_bar && n > 0, MCPM_E_ARG, "mcpm_lpt_combine_vjp_f32: bad argument");
    StageTimer st_(p, ST_LPT, 96.0 * n);
    lpt_combine_vjp_kernel<<<(unsigned)((n + 255) / 256), 256, 0, p->stream>>>(F1, F2, gtab, n, xb, vb, gtab_bar);
    MCPM_LAUNCH_CHECK(p, "lpt_combine_vjp_kernel");
    return MCPM_OK;
}

// out[i] = scale * interp(x[i]; xp, fp) with clamped ends (np.interp); xp ascending, tables float64 on the DEVICE.
int mcpm_interp_f32(mcpm_plan *p, const float *x, int64_t n, const double *xp, const double *fp, int ntab, float scale, float *out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, x && xp && fp && out && n > 0 && ntab >= 2, MCPM_E_ARG, "mcpm_interp_f32: bad argument");
    interp_kernel<<<(unsigned)((n + 255) / 256), 256, 0, p->stream>>>(x, n, xp, fp, ntab, scale, out);
    MCPM_LAUNCH_CHECK(p, "interp_kernel");
    return MCPM_OK;
}

}  // extern "C"
