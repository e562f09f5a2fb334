// k-space kernels of libmcpm.so: inverse Laplacian x gradient (forces), Hessian (2LPT source), their
// adjoints, and the real-space Hessian combine.  Reference: montecosmo/nbody.py:109-163 (kernels),
// :596-603 (pm_forces), :611-627 (pm_forces2).  Wavevectors are computed in-kernel from the mode
// indices (nbody.py:50-77 in cell units), never materialised.
//
// Hermitian note.  The reference multiplies by i*k_c without zeroing the Nyquist mode (nbody.py:163) and
// relies on irfftn (= complex ifft over x,y then c2r over z) discarding the non-Hermitian part on the
// kz = 0 and kz = nz/2 planes.  A C2R library is free to treat non-Hermitian input differently, so the
// forward kernels apply that projection explicitly: on those two planes a factor (i k_a) changes sign
// under k -> -k unless index a sits at its Nyquist, hence a multiplier with an odd number of Nyquist
// factors projects to 0 and everything else is unchanged.  Output is identical to numpy's.
#include "mcpm_internal.h"

#define TWO_PI 6.283185307179586f

__device__ __forceinline__ float kfreq(int i, int n) {  // 2*pi*fftfreq(n)[i]
    int s = (i < (n + 1) / 2) ? i : i - n;
    return TWO_PI * (float)s / (float)n;
}

template <int FD>
__device__ __forceinline__ float lap_term(float k) {
    if (FD == MCPM_FD_2) return (cosf(k) - 1.f) * 2.f;
    if (FD == MCPM_FD_4) return (cosf(2.f * k) - 16.f * cosf(k) + 15.f) / 6.f;
    return k * k;
}
template <int FD>
__device__ __forceinline__ float grad_term(float k) {
    if (FD == MCPM_FD_2) return sinf(k);
    if (FD == MCPM_FD_4) return (8.f * sinf(k) - sinf(2.f * k)) / 6.f;
    return k;
}
__device__ __forceinline__ float sincf_pi(float k) {  // np.sinc(k / (2 pi)) = sin(k/2)/(k/2)
    float h = 0.5f * k;
    return h == 0.f ? 1.f : sinf(h) / h;
}

struct KArgs {
    Geom g;
    float scale;
    float kcut;   // <= 0: no gaussian smoothing
    int deconv;   // 0: none, else divide by sinc^(2*deconv)
    int zweights, hermitian, accumulate;
};

struct Mode {
    int ix, iy, iz;
    float gk[3];   // gradient factors (real part of gradient_hat / i)
    float L;       // invlaplace_hat * gaussian / deconvolution * scale
    bool nyq[3];
    bool special;  // kz == 0 or kz == nz/2 plane
    float zw;
};

template <int LAP, int GRAD>
__device__ __forceinline__ Mode decode(const KArgs &a, uint32_t idx) {
    Mode m;
    const Geom &g = a.g;
    m.iz = idx % (uint32_t)g.nzh;
    uint32_t r = idx / (uint32_t)g.nzh;
    m.iy = r % (uint32_t)g.ny;
    m.ix = r / (uint32_t)g.ny;
    float kx = kfreq(m.ix, g.nx), ky = kfreq(m.iy, g.ny), kz = TWO_PI * (float)m.iz / (float)g.nz;
    float kk = lap_term<LAP>(kx) + lap_term<LAP>(ky) + lap_term<LAP>(kz);
    float L = kk == 0.f ? 0.f : -1.f / kk;  // - safe_div(1, kk)
    if (a.kcut > 0.f) {
        float k2 = kx * kx + ky * ky + kz * kz, rc = TWO_PI / a.kcut;
        L *= expf(-k2 * rc * rc * 0.5f);
    }
    if (a.deconv > 0) {
        float s = sincf_pi(kx) * sincf_pi(ky) * sincf_pi(kz);
        float s2 = s * s, d = 1.f;
        for (int i = 0; i < a.deconv; ++i) d *= s2;
        L /= d;
    }
    m.L = L * a.scale;
    m.gk[0] = grad_term<GRAD>(kx);
    m.gk[1] = grad_term<GRAD>(ky);
    m.gk[2] = grad_term<GRAD>(kz);
    m.nyq[0] = !(g.nx & 1) && m.ix == g.nx / 2;
    m.nyq[1] = !(g.ny & 1) && m.iy == g.ny / 2;
    m.nyq[2] = m.iz == g.nz / 2;
    m.special = (m.iz == 0) || m.nyq[2];
    m.zw = m.special ? 1.f : 2.f;
    return m;
}

// out[c] = scale * (-(i gk_c)) * L * in
template <int LAP, int GRAD>
__global__ __launch_bounds__(256) void kspace_force_kernel(KArgs a, const float2 *__restrict__ in,
                                                           float2 *__restrict__ out, int64_t Mh) {
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    Mode m = decode<LAP, GRAD>(a, idx);
    float2 v = in[idx];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = m.gk[c] * m.L;
        if (m.special && m.nyq[c]) s = 0.f;
        out[c * Mh + idx] = make_float2(s * v.y, -s * v.x);  // (a+ib)(-i s)
    }
}

// out (+)= scale * [zw] * sum_c conj(-(i gk_c) L) in[c]
template <int LAP, int GRAD>
__global__ __launch_bounds__(256) void kspace_force_vjp_kernel(KArgs a, const float2 *__restrict__ in,
                                                               float2 *__restrict__ out, int64_t Mh) {
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    Mode m = decode<LAP, GRAD>(a, idx);
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = m.gk[c] * m.L;
        if (a.hermitian && m.special && m.nyq[c]) s = 0.f;
        float2 v = in[c * Mh + idx];
        re += -s * v.y;  // (a+ib)(+i s)
        im += s * v.x;
    }
    if (a.zweights) {
        re *= m.zw;
        im *= m.zw;
    }
    if (a.accumulate) {
        float2 o = out[idx];
        re += o.x;
        im += o.y;
    }
    out[idx] = make_float2(re, im);
}

// out[ab] = scale * (i gk_a)(i gk_b) * L * in, ab = 00,01,02,11,12,22
template <int LAP, int GRAD>
__global__ __launch_bounds__(256) void kspace_hessian_kernel(KArgs a, const float2 *__restrict__ in,
                                                             float2 *__restrict__ out, int64_t Mh) {
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    Mode m = decode<LAP, GRAD>(a, idx);
    float2 v = in[idx];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            float s = -m.gk[i] * m.gk[j] * m.L;
            if (m.special && (m.nyq[i] != m.nyq[j])) s = 0.f;
            out[k * Mh + idx] = make_float2(s * v.x, s * v.y);
            ++k;
        }
}

template <int LAP, int GRAD>
__global__ __launch_bounds__(256) void kspace_hessian_vjp_kernel(KArgs a, const float2 *__restrict__ in,
                                                                 float2 *__restrict__ out, int64_t Mh) {
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    Mode m = decode<LAP, GRAD>(a, idx);
    float re = 0.f, im = 0.f;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) {
            float s = -m.gk[i] * m.gk[j] * m.L;
            if (a.hermitian && m.special && (m.nyq[i] != m.nyq[j])) s = 0.f;
            float2 v = in[k * Mh + idx];
            re += s * v.x;
            im += s * v.y;
            ++k;
        }
    if (a.zweights) {
        re *= m.zw;
        im *= m.zw;
    }
    if (a.accumulate) {
        float2 o = out[idx];
        re += o.x;
        im += o.y;
    }
    out[idx] = make_float2(re, im);
}

// out (+)= scale * exp(+-i shift (kx+ky+kz)) / prod_a sinc(k_a/2pi)^deconv [/ zw] * in
// (interlacing phase nbody.py:525 and paint-kernel deconvolution nbody.py:315-334; conj / inverse multiplicity
// weights give the adjoint used by nufft_vjp)
__global__ __launch_bounds__(256) void kspace_phase_kernel(Geom g, float scale, float shift, int deconv, int conj, int inv_zw,
                                                           int accumulate, const float2 *__restrict__ in,
                                                           float2 *__restrict__ out, int64_t Mh) {
    uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= Mh) return;
    const int iz = idx % (uint32_t)g.nzh;
    const uint32_t r = idx / (uint32_t)g.nzh;
    const int iy = r % (uint32_t)g.ny, ix = r / (uint32_t)g.ny;
    const float kx = kfreq(ix, g.nx), ky = kfreq(iy, g.ny), kz = TWO_PI * (float)iz / (float)g.nz;
    float m = scale;
    if (deconv > 0) {
        const float sc = sincf_pi(kx) * sincf_pi(ky) * sincf_pi(kz);
        float d = 1.f;
        for (int i = 0; i < deconv; ++i) d *= sc;
        m /= d;
    }
    if (inv_zw && !(iz == 0 || iz == g.nz / 2)) m *= 0.5f;
    float c = 1.f, sn = 0.f;
    if (shift != 0.f) {
        const float ph = shift * (kx + ky + kz);
        c = cosf(ph);
        sn = conj ? -sinf(ph) : sinf(ph);
    }
    const float2 v = in[idx];
    float re = m * (v.x * c - v.y * sn), im = m * (v.x * sn + v.y * c);
    if (accumulate) {
        const float2 o = out[idx];
        re += o.x;
        im += o.y;
    }
    out[idx] = make_float2(re, im);
}

// delta2 = h00*h11 + h22*(h00+h11) - h01^2 - h02^2 - h12^2 (running-sum order of nbody.py:615-627)
__global__ __launch_bounds__(256) void hessian_combine_kernel(const float *__restrict__ h, int64_t M,
                                                              float *__restrict__ d2) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float h00 = h[i], h01 = h[M + i], h02 = h[2 * M + i], h11 = h[3 * M + i], h12 = h[4 * M + i], h22 = h[5 * M + i];
    float d = 0.f;
    d -= h01 * h01;
    d -= h02 * h02;
    d += h11 * h00;
    d -= h12 * h12;
    d += h22 * (h00 + h11);
    d2[i] = d;
}

// hb may alias h (in-place): every input of element i is read before any output of element i is written.
__global__ __launch_bounds__(256) void hessian_combine_vjp_kernel(const float *h, const float *__restrict__ d2b, int64_t M,
                                                                  float *hb) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float h00 = h[i], h01 = h[M + i], h02 = h[2 * M + i], h11 = h[3 * M + i], h12 = h[4 * M + i], h22 = h[5 * M + i];
    float b = d2b[i];
    hb[i] = b * (h11 + h22);
    hb[M + i] = -2.f * b * h01;
    hb[2 * M + i] = -2.f * b * h02;
    hb[3 * M + i] = b * (h00 + h22);
    hb[4 * M + i] = -2.f * b * h12;
    hb[5 * M + i] = b * (h00 + h11);
}

static bool fd_ok(int fd) { return fd == MCPM_FD_INF || fd == MCPM_FD_2 || fd == MCPM_FD_4; }

#define DISPATCH_FD(lap, grad, KERNEL, ...)                                                   \
    do {                                                                                      \
        if (lap == MCPM_FD_INF && grad == MCPM_FD_INF) KERNEL<MCPM_FD_INF, MCPM_FD_INF> __VA_ARGS__; \
        else if (lap == MCPM_FD_INF && grad == MCPM_FD_2) KERNEL<MCPM_FD_INF, MCPM_FD_2> __VA_ARGS__; \
        else if (lap == MCPM_FD_INF && grad == MCPM_FD_4) KERNEL<MCPM_FD_INF, MCPM_FD_4> __VA_ARGS__; \
        else if (lap == MCPM_FD_2 && grad == MCPM_FD_INF) KERNEL<MCPM_FD_2, MCPM_FD_INF> __VA_ARGS__; \
        else if (lap == MCPM_FD_2 && grad == MCPM_FD_2) KERNEL<MCPM_FD_2, MCPM_FD_2> __VA_ARGS__;     \
        else if (lap == MCPM_FD_2 && grad == MCPM_FD_4) KERNEL<MCPM_FD_2, MCPM_FD_4> __VA_ARGS__;     \
        else if (lap == MCPM_FD_4 && grad == MCPM_FD_INF) KERNEL<MCPM_FD_4, MCPM_FD_INF> __VA_ARGS__; \
        else if (lap == MCPM_FD_4 && grad == MCPM_FD_2) KERNEL<MCPM_FD_4, MCPM_FD_2> __VA_ARGS__;     \
        else KERNEL<MCPM_FD_4, MCPM_FD_4> __VA_ARGS__;                                                \
    } while (0)

extern "C" {

int mcpm_kspace_force_f32(mcpm_plan *p, const float *in, float *out3, float scale, int lap_fd, int grad_fd, float kcut,
                          int deconv_order) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out3, MCPM_E_ARG, "mcpm_kspace_force_f32: null buffer");
    MCPM_REQUIRE(p, fd_ok(lap_fd) && fd_ok(grad_fd), MCPM_E_ORDER, "finite-difference order must be 0 (inf), 2 or 4");
    KArgs a{p->g, scale, kcut, deconv_order, 0, 1, 0};
    StageTimer st_(p, ST_KSPACE, 32.0 * p->Mh);
    unsigned nb = (unsigned)((p->Mh + 255) / 256);
    DISPATCH_FD(lap_fd, grad_fd, kspace_force_kernel, <<<nb, 256, 0, p->stream>>>(a, (const float2 *)in, (float2 *)out3, p->Mh));
    MCPM_LAUNCH_CHECK(p, "kspace_force_kernel");
    return MCPM_OK;
}

int mcpm_kspace_force_vjp_f32(mcpm_plan *p, const float *in3, float *out, float scale, int lap_fd, int grad_fd,
                              float kcut, int deconv_order, int zweights, int hermitian, int accumulate) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in3 && out, MCPM_E_ARG, "mcpm_kspace_force_vjp_f32: null buffer");
    MCPM_REQUIRE(p, fd_ok(lap_fd) && fd_ok(grad_fd), MCPM_E_ORDER, "finite-difference order must be 0 (inf), 2 or 4");
    KArgs a{p->g, scale, kcut, deconv_order, zweights, hermitian, accumulate};
    StageTimer st_(p, ST_KSPACE, (accumulate ? 40.0 : 32.0) * p->Mh);
    unsigned nb = (unsigned)((p->Mh + 255) / 256);
    DISPATCH_FD(lap_fd, grad_fd, kspace_force_vjp_kernel, <<<nb, 256, 0, p->stream>>>(a, (const float2 *)in3, (float2 *)out, p->Mh));
    MCPM_LAUNCH_CHECK(p, "kspace_force_vjp_kernel");
    return MCPM_OK;
}

int mcpm_kspace_hessian_f32(mcpm_plan *p, const float *in, float *out6, float scale, int lap_fd, int grad_fd) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out6, MCPM_E_ARG, "mcpm_kspace_hessian_f32: null buffer");
    MCPM_REQUIRE(p, fd_ok(lap_fd) && fd_ok(grad_fd), MCPM_E_ORDER, "finite-difference order must be 0 (inf), 2 or 4");
    KArgs a{p->g, scale, 0.f, 0, 0, 1, 0};
    StageTimer st_(p, ST_KSPACE, 56.0 * p->Mh);
    unsigned nb = (unsigned)((p->Mh + 255) / 256);
    DISPATCH_FD(lap_fd, grad_fd, kspace_hessian_kernel, <<<nb, 256, 0, p->stream>>>(a, (const float2 *)in, (float2 *)out6, p->Mh));
    MCPM_LAUNCH_CHECK(p, "kspace_hessian_kernel");
    return MCPM_OK;
}

int mcpm_kspace_hessian_vjp_f32(mcpm_plan *p, const float *in6, float *out, float scale, int lap_fd, int grad_fd,
                                int zweights, int accumulate) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in6 && out, MCPM_E_ARG, "mcpm_kspace_hessian_vjp_f32: null buffer");
    MCPM_REQUIRE(p, fd_ok(lap_fd) && fd_ok(grad_fd), MCPM_E_ORDER, "finite-difference order must be 0 (inf), 2 or 4");
    KArgs a{p->g, scale, 0.f, 0, zweights, 0, accumulate};
    StageTimer st_(p, ST_KSPACE, (accumulate ? 64.0 : 56.0) * p->Mh);
    unsigned nb = (unsigned)((p->Mh + 255) / 256);
    DISPATCH_FD(lap_fd, grad_fd, kspace_hessian_vjp_kernel, <<<nb, 256, 0, p->stream>>>(a, (const float2 *)in6, (float2 *)out, p->Mh));
    MCPM_LAUNCH_CHECK(p, "kspace_hessian_vjp_kernel");
    return MCPM_OK;
}

int mcpm_kspace_phase_f32(mcpm_plan *p, const float *in, float *out, float scale, float shift, int deconv_order, int conj,
                          int inv_zweights, int accumulate) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out, MCPM_E_ARG, "mcpm_kspace_phase_f32: null buffer");
    MCPM_REQUIRE(p, deconv_order >= 0 && deconv_order <= 8, MCPM_E_ORDER, "mcpm_kspace_phase_f32: bad deconvolution order");
    StageTimer st_(p, ST_KSPACE, (accumulate ? 24.0 : 16.0) * p->Mh);
    unsigned nb = (unsigned)((p->Mh + 255) / 256);
    kspace_phase_kernel<<<nb, 256, 0, p->stream>>>(p->g, scale, shift, deconv_order, conj, inv_zweights, accumulate,
                                                   (const float2 *)in, (float2 *)out, p->Mh);
    MCPM_LAUNCH_CHECK(p, "kspace_phase_kernel");
    return MCPM_OK;
}

int mcpm_hessian_combine_f32(mcpm_plan *p, const float *hess6, float *delta2) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, hess6 && delta2, MCPM_E_ARG, "mcpm_hessian_combine_f32: null buffer");
    StageTimer st_(p, ST_LPT, 28.0 * p->M);
    unsigned nb = (unsigned)((p->M + 255) / 256);
    hessian_combine_kernel<<<nb, 256, 0, p->stream>>>(hess6, p->M, delta2);
    MCPM_LAUNCH_CHECK(p, "hessian_combine_kernel");
    return MCPM_OK;
}

int mcpm_hessian_combine_vjp_f32(mcpm_plan *p, const float *hess6, const float *delta2_bar, float *hess6_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, hess6 && delta2_bar && hess6_bar, MCPM_E_ARG, "mcpm_hessian_combine_vjp_f32: null buffer");
    StageTimer st_(p, ST_LPT, 52.0 * p->M);
    unsigned nb = (unsigned)((p->M + 255) / 256);
    hessian_combine_vjp_kernel<<<nb, 256, 0, p->stream>>>(hess6, delta2_bar, p->M, hess6_bar);
    MCPM_LAUNCH_CHECK(p, "hessian_combine_vjp_kernel");
    return MCPM_OK;
}

}  // extern "C"
