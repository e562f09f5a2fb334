// Composite operators of libmcpm.so: pm_forces, pm_forces2, lpt, nbody_bf and the hand-written reverse
// sweep (VJP) of nbody_bf.  Reference: montecosmo/nbody.py:583-667 (forces, lpt), :902-1002 (BullFrog
// vector field + diffrax Euler driver).
//
// Stepping form.  The reference evaluates drift(dg/2) . kick . drift(dg/2) as a vector field and lets
// diffrax's Euler add it back (nbody.py:946-951, :999).  Consecutive half drifts commute with nothing in
// between, so the loop below runs the identical map as
//     x'_0 = x_0 + v_0 dg/2;   v_{i+1} = alpha_i v_i + beta_i F(x'_i);   x'_{i+1} = x'_i + v_{i+1} tau_i
// with tau_i = dg (dg/2 on the last step): ONE fused read+kick+drift particle kernel per step, whose output
// is written straight into the next checkpoint slot.  The three force meshes of each step are C2R-ed
// straight into the checkpoint as well, so the adjoint needs no force recomputation.
#include "particles_dev.h"
#include "reduce_dev.h"

// dpos = (init ? 0 : dpos) + ad * F(q), vel likewise with av; F from three contiguous meshes.
__global__ __launch_bounds__(256) void lpt_accum_kernel(Geom g, const float *__restrict__ meshes, int64_t M, float ad,
                                                        float av, int init, float *__restrict__ dpos,
                                                        float *__restrict__ vel) {
    PIdx pi = particle_index<MCPM_POS_LATTICE>(g, 0);
    if (!pi.valid) return;
    int64_t c = lattice_cell(g, pi);
    float F0 = meshes[c], F1 = meshes[M + c], F2 = meshes[2 * M + c];
    P3 d = {0.f, 0.f, 0.f}, v = {0.f, 0.f, 0.f};
    if (!init) {
        d = load3(dpos, pi.i);
        v = load3(vel, pi.i);
    }
    d.x += ad * F0; d.y += ad * F1; d.z += ad * F2;
    v.x += av * F0; v.y += av * F1; v.z += av * F2;
    store3(dpos, pi.i, d);
    store3(vel, pi.i, v);
}

// Adjoint of the NGP lattice read on the identity lattice: out_c[cell(i)] = a*xb[i][c] + b*vb[i][c], every cell is hit
// exactly once (plain store).  Other lattices: mcpm_lattice_scatter_fx (particles.hip), order-independent fixed-point sums.
__global__ __launch_bounds__(256) void lattice_scatter_kernel(Geom g, const float *__restrict__ xb,
                                                              const float *__restrict__ vb, float a, float b,
                                                              float *__restrict__ out, int64_t M) {
    PIdx pi = particle_index<MCPM_POS_LATTICE>(g, 0);
    if (!pi.valid) return;
    int64_t c = lattice_cell(g, pi);
    P3 x = load3(xb, pi.i), v = load3(vb, pi.i);
    out[c] = a * x.x + b * v.x;
    out[M + c] = a * x.y + b * v.y;
    out[2 * M + c] = a * x.z + b * v.z;
}

// out0 += sum_i a[i].F(q_i), out1 += sum_i b[i].F(q_i)  (growth-scalar cotangents of lpt)
__global__ __launch_bounds__(256) void lattice_dot_kernel(Geom g, const float *__restrict__ meshes, int64_t M,
                                                          const float *__restrict__ a, const float *__restrict__ b,
                                                          double *__restrict__ P) {
    PIdx pi = particle_index<MCPM_POS_LATTICE>(g, 0);
    double ra = 0., rb = 0.;
    if (pi.valid) {
        int64_t c = lattice_cell(g, pi);
        float F0 = meshes[c], F1 = meshes[M + c], F2 = meshes[2 * M + c];
        if (a) {
            P3 x = load3(a, pi.i);
            ra = (double)(x.x * F0 + x.y * F1 + x.z * F2);
        }
        if (b) {
            P3 x = load3(b, pi.i);
            rb = (double)(x.x * F0 + x.y * F1 + x.z * F2);
        }
    }
    const double v[2] = {ra, rb};
    block_partial<2>(v, P, gridDim.x, blockIdx.x);      // deterministic: one partial per workgroup, det_fold_kernel adds them up
}

// Sums of three per-lane values over the 256-thread workgroup (waves in f32 -- the lane values are f32 products already --,
// the four waves and everything beyond in f64), WRITTEN as this workgroup's partials P[k * nblk + block] (det_fold_kernel adds them
// up in a fixed order: reduce_dev.h); optionally the workgroup maximum of |a|, |b|, |c| (see absmax_commit).  One barrier.
__device__ __forceinline__ void block_add3_max(float a, float b, float c, double *__restrict__ P, unsigned nblk, float ma, float mb, float mc,
                                               unsigned *__restrict__ out_max) {
    __shared__ double sh[3][4];
    __shared__ unsigned shm[4];
    const float ta = wave_sum_dpp(a), tb = wave_sum_dpp(b), tc = wave_sum_dpp(c);
    unsigned m = 0u;
    if (out_max) m = wave_umax_dpp(max(max(__float_as_uint(ma) & 0x7fffffffu, __float_as_uint(mb) & 0x7fffffffu), __float_as_uint(mc) & 0x7fffffffu));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) {
        sh[0][w] = (double)ta;
        sh[1][w] = (double)tb;
        sh[2][w] = (double)tc;
        shm[w] = m;
    }
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;      // 1 .. 4 waves (lattice rows shorter than 256 run smaller workgroups)
    if (threadIdx.x < 3) {
        double t = 0.;
        for (int i = 0; i < nw; ++i) t += sh[threadIdx.x][i];
        P[(size_t)threadIdx.x * nblk + blockIdx.x] = t;
    } else if (threadIdx.x == 3 && out_max) {
        unsigned mm = 0u;
        for (int i = 0; i < nw; ++i) mm = max(mm, shm[i]);
        unsigned *slot = out_max + (blockIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE;
        if (mm > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, mm);
    }
}

// Adjoint of one fused step (see file header).  Inputs: x'_i, v_i (checkpoint), cotangents xb, vb of
// (x'_{i+1}, v_{i+1}) (updated in place to those of (x'_i, v_i)), the step's three force meshes and
// rho_bar = cotangent of the painted density.
// max over the workgroup of |a|, |b|, |c| as float bits (any NaN / Inf gives >= 0x7f800000): what the fixed-point
// three-component paint needs to choose its scale (particles.hip, paint3_fx_kernel).  One LDS atomic per wave, then ONE
// L2-coherent read of a slot per workgroup and a global atomic only if the slot is smaller -- after the first few
// workgroups none is: per-wave atomics on one address cost 20 ms per launch at 512^3.  Slots are a cache line apart.
__device__ __forceinline__ void absmax_commit(float a, float b, float c, unsigned *__restrict__ out) {
    __shared__ unsigned sh_max;
    if (threadIdx.x == 0) sh_max = 0u;
    __syncthreads();
    unsigned m = max(max(__float_as_uint(a) & 0x7fffffffu, __float_as_uint(b) & 0x7fffffffu), __float_as_uint(c) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&sh_max, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *slot = out + (blockIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE;
        if (sh_max > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, sh_max);
    }
}

template <int ORDER, bool IL>
__global__ __launch_bounds__(256) void step_adjoint_kernel(Geom g, const float *__restrict__ x, const float *__restrict__ v,
                                                           const float *xb_src, const float *vb_src, float *xb, float *vb,
                                                           const float *__restrict__ fm, const float *__restrict__ rho_bar,
                                                           int64_t M, float alpha, float beta, float tau,
                                                           double *__restrict__ part, float *__restrict__ fb_next, float beta_next,
                                                           float tau_next, float dtau_ddg, unsigned *__restrict__ fb_max, int nt) {
    PIdx pi = particle_index<MCPM_POS_LATTICE>(g, 0);
    float ra = 0.f, rb = 0.f, rc = 0.f;
    P3 fbn = {0.f, 0.f, 0.f};
    if (pi.valid) {
        P3 d, vi, xbi, vbi;
        if (nt & 1) load3_nt4(x, v, xb_src, vb_src, pi.i, d, vi, xbi, vbi);      // streaming: each is read once by this kernel
        else {
            d = load3(x, pi.i);
            vi = load3(v, pi.i);
            xbi = load3(xb_src, pi.i);
            vbi = load3(vb_src, pi.i);
        }
        const P3 xin = xbi;
        int c[3];
        float f[3];
        locate<MCPM_POS_LATTICE, ORDER>(g, pi, d, c, f);
        Stencil<ORDER> s(g, c);
        const P3 vt = {vbi.x + tau * xbi.x, vbi.y + tau * xbi.y, vbi.z + tau * xbi.z};
        const float Fb[3] = {beta * vt.x, beta * vt.y, beta * vt.z};
        float F[3], G[3][3];
        interp3<ORDER, true, IL>(fm, M, s, f, F, G);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            xbi.x += Fb[k] * G[k][0];
            xbi.y += Fb[k] * G[k][1];
            xbi.z += Fb[k] * G[k][2];
        }
        {
            float val, gx, gy, gz;
            interp<ORDER, true>(rho_bar, s, f, val, gx, gy, gz);
            xbi.x += gx;
            xbi.y += gy;
            xbi.z += gz;
        }
        ra = vt.x * vi.x + vt.y * vi.y + vt.z * vi.z;
        rb = vt.x * F[0] + vt.y * F[1] + vt.z * F[2];
        {   // explicit dependence of the drift x' += v_new tau on the step size: <x_bar_in, v_new> dtau/ddg
            const float vnx = alpha * vi.x + beta * F[0], vny = alpha * vi.y + beta * F[1], vnz = alpha * vi.z + beta * F[2];
            rc = dtau_ddg * (xin.x * vnx + xin.y * vny + xin.z * vnz);
        }
        const P3 vnew = {alpha * vt.x, alpha * vt.y, alpha * vt.z};
        if (nt & 2) {
            store3_nt(xb, pi.i, xbi.x, xbi.y, xbi.z);
            store3_nt(vb, pi.i, vnew.x, vnew.y, vnew.z);
        } else {
            store3(xb, pi.i, xbi);
            store3(vb, pi.i, vnew);
        }
        if (fb_next) {  // force cotangent of the PREVIOUS step, F_bar = beta' (v_bar + tau' x_bar)
            fbn = P3{beta_next * (vnew.x + tau_next * xbi.x), beta_next * (vnew.y + tau_next * xbi.y),
                     beta_next * (vnew.z + tau_next * xbi.z)};
            if (nt & 2) store3_nt(fb_next, pi.i, fbn.x, fbn.y, fbn.z);
            else store3(fb_next, pi.i, fbn);
        }
    }
    block_add3_max(ra, rb, rc, part, gridDim.x, fbn.x, fbn.y, fbn.z, fb_max);
}

// partials of sum_i a[i] b[i] (det_fold_kernel scales and adds them up)
__global__ __launch_bounds__(256) void dot_partial_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n, double *__restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double v[1] = {i < n ? (double)a[i] * (double)b[i] : 0.};
    block_partial<1>(v, P, gridDim.x, blockIdx.x);
}

__global__ void axpby_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t n, float a, float b,
                             float *__restrict__ out, unsigned *__restrict__ out_max) {
    unsigned r = 0u;      // max |out| as bits: a NaN / Inf stays the maximum
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float o = a * x[i] + b * y[i];
        out[i] = o;
        r = max(r, __float_as_uint(o) & 0x7fffffffu);
    }
    if (out_max) absmax_commit(__uint_as_float(r), 0.f, 0.f, out_max);
}


// adjoint of the NGP lattice read: a plain store per cell on the identity lattice, fixed-point sums otherwise
static int lattice_scatter(mcpm_plan *p, const float *xb, const float *vb, float a, float b, float *meshes3) {
    if (!p->g.same_lattice) return mcpm_lattice_scatter_fx(p, xb, vb, a, b, meshes3);
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    lattice_scatter_kernel<<<grid, block, 0, p->stream>>>(p->g, xb, vb, a, b, meshes3, p->M);
    MCPM_LAUNCH_CHECK(p, "lattice_scatter_kernel");
    return MCPM_OK;
}

// adds up the K per-workgroup partials a kernel with `nblk` workgroups left in P (reduce_dev.h): *outs[k] += scale sum (NULL: dropped)
static int det_fold(mcpm_plan *p, const double *P, double *Q, unsigned *ticket, unsigned R, unsigned nblk, int K, double scale,
                    double *o0, double *o1 = nullptr, double *o2 = nullptr) {
    DetOuts o{};
    o.p[0] = o0, o.p[1] = o1, o.p[2] = o2;
    o.accumulate = 1;
    det_fold_kernel<<<R, 256, 0, p->stream>>>(P, nblk, K, Q, ticket, scale, o);
    MCPM_LAUNCH_CHECK(p, "det_fold_kernel");
    return MCPM_OK;
}

// out0 += sum_i a[i].F(q_i), out1 += sum_i b[i].F(q_i), added up in a fixed order (deterministic grid sums, reduce_dev.h)
static int lattice_dot(mcpm_plan *p, const float *meshes3, const float *a, const float *b, double *out0, double *out1) {
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    double *P, *Q;
    unsigned *ticket, R;
    MCPM_TRY(mcpm_det_scratch(p, 2, grid.x, &P, &Q, &ticket, &R));
    lattice_dot_kernel<<<grid, block, 0, p->stream>>>(p->g, meshes3, p->M, a, b, P);
    MCPM_LAUNCH_CHECK(p, "lattice_dot_kernel");
    MCPM_TRY(det_fold(p, P, Q, ticket, R, grid.x, 2, 1.0, a ? out0 : nullptr, b ? out1 : nullptr));
    return MCPM_OK;
}

// out = a x + b y; with_max: also leaves max|out| for the fixed-point paint of `out` (plan->fx_wmax / fx_src)
static int axpby(mcpm_plan *p, const float *x, const float *y, int64_t n, float a, float b, float *out, bool with_max = false) {
    StageTimer st_(p, ST_AXPY, 12.0 * n);
    unsigned nb = (unsigned)((n + 255) / 256);
    with_max = with_max && p->paint3_variant == 4;
    if (with_max && nb > 16384) nb = 16384;      // grid-stride: few workgroups commit the maximum
    if (with_max) {
        if (!p->fx_clean) MCPM_HIP(p, hipMemsetAsync(p->fx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream));
        p->fx_clean = 0;
        p->fx_src = out;
    }
    axpby_kernel<<<nb, 256, 0, p->stream>>>(x, y, n, a, b, out, with_max ? p->fx_wmax : nullptr);
    MCPM_LAUNCH_CHECK(p, "axpby_kernel");
    return MCPM_OK;
}

// spectrum -> three force meshes (C2R output buffer `fm`), using plan->spec as spectral scratch
static int spec_to_force_meshes(mcpm_plan *p, const float *spec, int lap_fd, int grad_fd, float kcut, int deconv,
                                float *fm) {
    if (mcpm_fftpm_supported(p) && !p->g.xslab && lap_fd == MCPM_FD_INF && grad_fd == MCPM_FD_INF && kcut <= 0.f && !deconv &&
        spec != p->spec && spec != p->spec1)
        return mcpm_fftpm_spec_meshes(p, spec, fm, 3);
    MCPM_TRY(mcpm_kspace_force_f32(p, spec, p->spec, 1.f / (float)p->M, lap_fd, grad_fd, kcut, deconv));
    MCPM_TRY(mcpm_fft_c2r(p, p->spec, fm, 3));
    return MCPM_OK;
}

// half-spectrum -> delta2 spectrum in plan->spec1; leaves the six Hessian meshes in fmesh[0:6]
static bool spec_custom(const mcpm_plan *p, const float *spec, int lap_fd, int grad_fd) {
    return mcpm_fftpm_supported(p) && !p->g.xslab && lap_fd == MCPM_FD_INF && grad_fd == MCPM_FD_INF && spec != p->spec &&
           spec != p->spec1;
}

// half-spectrum -> delta2 (real, in plan->rho), the six Hessian meshes left in `h` (default: fmesh[0:6])
static int spec_to_delta2_real(mcpm_plan *p, const float *spec, int lap_fd, int grad_fd, float *h = nullptr) {
    if (!h) h = p->fmesh;
    if (spec_custom(p, spec, lap_fd, grad_fd)) {
        MCPM_TRY(mcpm_fftpm_spec_meshes(p, spec, h, 6));
    } else {
        MCPM_TRY(mcpm_kspace_hessian_f32(p, spec, p->spec, 1.f / (float)p->M, lap_fd, grad_fd));
        MCPM_TRY(mcpm_fft_c2r(p, p->spec, h, 6));
    }
    MCPM_TRY(mcpm_hessian_combine_f32(p, h, p->rho));
    return MCPM_OK;
}

// delta2 (plan->rho) -> its three force meshes
static int delta2_to_force_meshes(mcpm_plan *p, int lap_fd, int grad_fd, float *fm) {
    if (lap_fd == MCPM_FD_INF && grad_fd == MCPM_FD_INF) return mcpm_force_meshes_f32(p, p->rho, fm);
    MCPM_TRY(mcpm_fft_r2c(p, p->rho, p->spec1, 1));
    return spec_to_force_meshes(p, p->spec1, lap_fd, grad_fd, 0.f, 0, fm);
}

// variable-size float scratch (pm_forces_vjp with arbitrary particle counts)
static int ensure_pscratch_n(mcpm_plan *p, int64_t nfloats) {
    if (p->vscratch && p->vscratch_n >= nfloats) return MCPM_OK;
    if (p->vscratch) (void)hipFree(p->vscratch);
    p->vscratch = nullptr;
    if (hipMalloc((void **)&p->vscratch, sizeof(float) * nfloats) != hipSuccess) return mcpm_fail(p, MCPM_E_NOMEM, "particle scratch");
    p->vscratch_n = nfloats;
    return MCPM_OK;
}

// x_bar, v_bar, F_bar of the composite adjoint: array j of the plan's particle scratch, mcpm_pitch() floats apart
static inline float *ps_arr(const mcpm_plan *p, int j) { return p->pscratch + j * mcpm_pitch(p); }
static int ensure_pscratch(mcpm_plan *p) {
    if (!p->pscratch && hipMalloc((void **)&p->pscratch, sizeof(float) * 3 * mcpm_pitch_max(p)) != hipSuccess)
        return mcpm_fail(p, MCPM_E_NOMEM, "adjoint particle scratch");
    return MCPM_OK;
}

// rho_bar = adjoint of (rho -> R2C -> k-space force kernels with FD orders / deconvolution -> 3 C2R) at f_bar (3 meshes)
static int force_meshes_vjp_opts(mcpm_plan *p, const float *fbar3, float *rho_bar, int lap_fd, int grad_fd, int deconv) {
    if (lap_fd == MCPM_FD_INF && grad_fd == MCPM_FD_INF && !deconv) return mcpm_force_meshes_vjp_f32(p, fbar3, rho_bar);
    MCPM_TRY(mcpm_fft_r2c(p, fbar3, p->spec, 3));
    MCPM_TRY(mcpm_kspace_force_vjp_f32(p, p->spec, p->spec1, 1.f / (float)p->M, lap_fd, grad_fd, 0.f, deconv, 0, 1, 0));
    return mcpm_fft_c2r(p, p->spec1, rho_bar, 1);
}

// Adjoint of lpt at the lattice (read_order = 1): cotangents (xb, vb) of (dpos, vel) -> init_mesh_bar (real-pair
// convention) and three DEVICE double accumulators sb = {g_bar, -g2_bar, -dg2dg_bar} (added to, not reset).
// `saved` (may be NULL): what the forward pass left in the checkpoint -- the first-order force meshes (3 M floats), then for
// lpt_order = 2 the second-order ones (3 M) and the six Hessian meshes of the first-order potential (6 M): the adjoint then
// recomputes none of them (a third of its transforms; 805 MB at 256^3 that a 288 GB part does not miss), and reads them only.
static int lpt_vjp_device(mcpm_plan *p, const float *init_mesh, int lpt_order, const double *lpt_scalars, const float *xb,
                          const float *vb, float *init_mesh_bar, double *sb, int lap_fd = MCPM_FD_INF, int grad_fd = MCPM_FD_INF,
                          const float *saved = nullptr) {
    const int64_t M = p->M;
    const float invM = 1.f / (float)M;
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    const float g = (float)lpt_scalars[0], g2 = (float)lpt_scalars[1], c2 = (float)lpt_scalars[2];
    if (saved) MCPM_TRY(lattice_dot(p, saved, xb, nullptr, sb + 0, nullptr));
    else {
        MCPM_TRY(spec_to_force_meshes(p, init_mesh, lap_fd, grad_fd, 0.f, 0, p->fmesh));
        MCPM_TRY(lattice_dot(p, p->fmesh, xb, nullptr, sb + 0, nullptr));
    }
    MCPM_TRY(lattice_scatter(p, xb, vb, g, 1.f, p->fmesh));
    const bool custom = spec_custom(p, init_mesh, lap_fd, grad_fd);
    if (custom) {
        MCPM_TRY(mcpm_fftpm_spec_meshes_vjp(p, p->fmesh, init_mesh_bar, 3));
    } else {
        MCPM_TRY(mcpm_fft_r2c(p, p->fmesh, p->spec, 3));
        MCPM_TRY(mcpm_kspace_force_vjp_f32(p, p->spec, init_mesh_bar, invM, lap_fd, grad_fd, 0.f, 0, 1, 0, 0));
    }
    if (lpt_order == 2) {
        float *h = p->fmesh, *f2 = p->fmesh + 6 * M;
        const float *hsrc = h;
        if (saved) {
            MCPM_TRY(lattice_dot(p, saved + 3 * M, xb, vb, sb + 1, sb + 2));  // negated on the host
            hsrc = saved + 6 * M;
        } else {
            MCPM_TRY(spec_to_delta2_real(p, init_mesh, lap_fd, grad_fd));        // h in fmesh[0:6], delta2 in rho
            MCPM_TRY(delta2_to_force_meshes(p, lap_fd, grad_fd, f2));            // F2 meshes
            MCPM_TRY(lattice_dot(p, f2, xb, vb, sb + 1, sb + 2));  // negated on the host
        }
        MCPM_TRY(lattice_scatter(p, xb, vb, -g2, -c2, f2));
        MCPM_TRY(force_meshes_vjp_opts(p, f2, p->rho, lap_fd, grad_fd, 0));          // delta2_bar
        MCPM_TRY(mcpm_hessian_combine_vjp_f32(p, hsrc, p->rho, h));                   // h -> h_bar (in place without `saved`)
        if (custom) {
            MCPM_TRY(mcpm_fftpm_spec_meshes_vjp(p, h, init_mesh_bar, 6));
        } else {
            MCPM_TRY(mcpm_fft_r2c(p, h, p->spec, 6));
            MCPM_TRY(mcpm_kspace_hessian_vjp_f32(p, p->spec, init_mesh_bar, invM, lap_fd, grad_fd, 1, 1));
        }
    }
    return MCPM_OK;
}

static int step_adjoint_particles(mcpm_plan *p, const float *pos_in, const float *vel_in, const float *force_meshes, int layout,
                                  const float *rho_bar, double alpha, double beta, double tau, int paint_order, float *pos_bar,
                                  float *vel_bar, double *alpha_bar, double *beta_bar, double dtau_ddg, double *dg_bar,
                                  const float *pos_bar_src, const float *vel_bar_src);
// force-mesh layout of the steppers' checkpoints: 1 = interleaved [cell][3] (hand-written Poisson solve), 0 = three meshes
static inline int step_layout(const mcpm_plan *p) { return mcpm_fftpm_supported(p) ? 1 : 0; }

extern "C" {

int mcpm_force_meshes_f32(mcpm_plan *p, const float *rho, float *fm3) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, rho && fm3, MCPM_E_ARG, "mcpm_force_meshes_f32: null buffer");
    if (mcpm_fftpm_supported(p)) return mcpm_fftpm_force_meshes(p, rho, fm3);
    MCPM_TRY(mcpm_fft_r2c(p, rho, p->spec1, 1));
    return spec_to_force_meshes(p, p->spec1, MCPM_FD_INF, MCPM_FD_INF, 0.f, 0, fm3);
}

int mcpm_force_meshes_vjp_f32(mcpm_plan *p, const float *fbar3, float *rho_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, fbar3 && rho_bar, MCPM_E_ARG, "mcpm_force_meshes_vjp_f32: null buffer");
    if (mcpm_fftpm_supported(p)) return mcpm_fftpm_force_meshes_vjp(p, fbar3, rho_bar);
    // rho_bar = C2R( (1/M) sum_c conj(m_c) R2C(f_bar_c) )
    MCPM_TRY(mcpm_fft_r2c(p, fbar3, p->spec, 3));
    MCPM_TRY(mcpm_kspace_force_vjp_f32(p, p->spec, p->spec1, 1.f / (float)p->M, MCPM_FD_INF, MCPM_FD_INF, 0.f, 0, 0, 1, 0));
    return mcpm_fft_c2r(p, p->spec1, rho_bar, 1);
}

int mcpm_pm_forces_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, int order, int paint_deconv, int lap_fd,
                       int grad_fd, float kcut, float *forces) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, forces != nullptr, MCPM_E_ARG, "mcpm_pm_forces_f32: null output");
    MCPM_TRY(mcpm_paint_f32(p, pos, n, mode, nullptr, 1, 1.f, order, p->rho, 0));
    if (!paint_deconv && lap_fd == MCPM_FD_INF && grad_fd == MCPM_FD_INF && kcut <= 0.f) {
        if (step_layout(p)) {   // interleaved force mesh: one 12-byte gather per stencil corner
            static const int nt = [] { const char *e = getenv("MCPM_NT3"); return e ? atoi(e) : 1; }();
            MCPM_TRY(mcpm_fftpm_force_meshes(p, p->rho, p->fmesh, 1, (nt == 1 || nt == 2) && p->M >= ((int64_t)1 << 23)));
            return mcpm_read3_il(p, pos, n, mode, p->fmesh, order, forces);
        }
        MCPM_TRY(mcpm_force_meshes_f32(p, p->rho, p->fmesh));
    } else {
        MCPM_TRY(mcpm_fft_r2c(p, p->rho, p->spec1, 1));
        MCPM_TRY(spec_to_force_meshes(p, p->spec1, lap_fd, grad_fd, kcut, paint_deconv ? order : 0, p->fmesh));
    }
    MCPM_TRY(mcpm_read_f32(p, pos, n, mode, p->fmesh, 3, order, forces));
    return MCPM_OK;
}

int mcpm_pm_forces_spec_f32(mcpm_plan *p, const float *spec, const float *pos, int64_t n, int mode, int order,
                            int lap_fd, int grad_fd, float kcut, float *forces) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, spec && forces, MCPM_E_ARG, "mcpm_pm_forces_spec_f32: null buffer");
    MCPM_TRY(spec_to_force_meshes(p, spec, lap_fd, grad_fd, kcut, 0, p->fmesh));
    MCPM_TRY(mcpm_read_f32(p, pos, n, mode, p->fmesh, 3, order, forces));
    return MCPM_OK;
}

// VJP of pm_forces (fd_order = inf, no deconvolution / smoothing).  spec == NULL: the particles were painted
// (mesh = shape tuple) and pos_bar carries both the read and the paint dependence; otherwise spec_bar (plain
// half-spectrum, real-pair convention with the irfftn multiplicity weights) is written as well.
int mcpm_pm_forces_vjp_f32(mcpm_plan *p, const float *spec, const float *pos, int64_t n, int mode, int order,
                           const float *forces_bar, float *pos_bar, float *spec_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos && forces_bar && pos_bar, MCPM_E_ARG, "mcpm_pm_forces_vjp_f32: null buffer");
    MCPM_REQUIRE(p, spec == nullptr || spec_bar != nullptr, MCPM_E_ARG, "mcpm_pm_forces_vjp_f32: spec given without spec_bar");
    const int64_t M = p->M;
    float *fm = p->fmesh, *fb = p->fmesh + 3 * M, *tmp = p->fmesh + 6 * M;  // force meshes, their cotangents, scratch
    // forward force meshes
    if (spec) {
        MCPM_TRY(spec_to_force_meshes(p, spec, MCPM_FD_INF, MCPM_FD_INF, 0.f, 0, fm));
    } else {
        MCPM_TRY(mcpm_paint_f32(p, pos, n, mode, nullptr, 1, 1.f, order, p->rho, 0));
        MCPM_TRY(mcpm_force_meshes_f32(p, p->rho, fm));
    }
    // read: pos_bar = sum_c F_bar_c grad f_c ; mesh cotangents f_bar_c = paint(pos, F_bar_c)
    MCPM_TRY(mcpm_read_vjp_pos_f32(p, pos, n, mode, fm, 3, order, forces_bar, pos_bar));
    MCPM_TRY(mcpm_paint3_f32(p, pos, n, mode, forces_bar, order, fb, 0));
    if (spec) {
        if (spec_custom(p, spec, MCPM_FD_INF, MCPM_FD_INF)) return mcpm_fftpm_spec_meshes_vjp(p, fb, spec_bar, 3);
        MCPM_TRY(mcpm_fft_r2c(p, fb, p->spec, 3));
        return mcpm_kspace_force_vjp_f32(p, p->spec, spec_bar, 1.f / (float)M, MCPM_FD_INF, MCPM_FD_INF, 0.f, 0, 1, 0, 0);
    }
    // painted: rho_bar = adjoint Poisson solve, then the paint's dependence on pos
    MCPM_TRY(mcpm_force_meshes_vjp_f32(p, fb, p->rho));
    MCPM_TRY(ensure_pscratch_n(p, 3 * n));
    float *pb2 = p->vscratch;
    MCPM_TRY(mcpm_paint_vjp_f32(p, pos, n, mode, nullptr, 1, 1.f, order, p->rho, pb2, nullptr));
    (void)tmp;
    return axpby(p, pos_bar, pb2, 3 * n, 1.f, 1.f, pos_bar);
}

int mcpm_pm_forces2_f32(mcpm_plan *p, const float *spec, const float *pos, int64_t n, int mode, int order, int lap_fd,
                        int grad_fd, float *forces) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, spec && forces, MCPM_E_ARG, "mcpm_pm_forces2_f32: null buffer");
    MCPM_TRY(spec_to_delta2_real(p, spec, lap_fd, grad_fd));
    MCPM_TRY(delta2_to_force_meshes(p, lap_fd, grad_fd, p->fmesh));
    MCPM_TRY(mcpm_read_f32(p, pos, n, mode, p->fmesh, 3, order, forces));
    return MCPM_OK;
}

int mcpm_lpt_accum_f32(mcpm_plan *p, const float *meshes3, float ad, float av, int init, float *dpos, float *vel) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, meshes3 && dpos && vel, MCPM_E_ARG, "mcpm_lpt_accum_f32: null buffer");
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    StageTimer st_(p, ST_LPT, 12.0 * p->M + (init ? 24.0 : 48.0) * p->Np);
    lpt_accum_kernel<<<grid, block, 0, p->stream>>>(p->g, meshes3, p->M, ad, av, init, dpos, vel);
    MCPM_LAUNCH_CHECK(p, "lpt_accum_kernel");
    return MCPM_OK;
}

int mcpm_lattice_scatter_f32(mcpm_plan *p, const float *xb, const float *vb, float a, float b, float *meshes3) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, xb && vb && meshes3, MCPM_E_ARG, "mcpm_lattice_scatter_f32: null buffer");
    StageTimer st_(p, ST_LPT, 12.0 * p->M + 24.0 * p->Np);
    return lattice_scatter(p, xb, vb, a, b, meshes3);
}

int mcpm_lattice_dot_f32(mcpm_plan *p, const float *meshes3, const float *a, const float *b, double *out2) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, meshes3 && out2 && (a || b), MCPM_E_ARG, "mcpm_lattice_dot_f32: null buffer");
    StageTimer st_(p, ST_LPT, 12.0 * p->M + 24.0 * p->Np);
    return lattice_dot(p, meshes3, a, b, out2, out2 + 1);
}

// `save` (may be NULL): 3 M floats (lpt_order 1) or 12 M (lpt_order 2) that receive the first-order force meshes, the second-order
// ones and the six Hessian meshes instead of the plan's scratch -- what lpt_vjp_device(saved) reads (mcpm_nbody_bf_f32's checkpoint)
static int lpt_forward(mcpm_plan *p, const float *init_mesh, int lpt_order, float g, float g2, float dg2dg, int lap_fd, int grad_fd,
                       float *dpos, float *vel, float *save) {
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    const int64_t M = p->M;
    float *f1 = save ? save : p->fmesh;
    MCPM_TRY(spec_to_force_meshes(p, init_mesh, lap_fd, grad_fd, 0.f, 0, f1));
    {
        StageTimer st_(p, ST_LPT, 12.0 * p->M + 24.0 * p->Np);
        lpt_accum_kernel<<<grid, block, 0, p->stream>>>(p->g, f1, p->M, g, 1.f, 1, dpos, vel);
    }
    MCPM_LAUNCH_CHECK(p, "lpt_accum_kernel");
    if (lpt_order == 2) {
        float *f2 = save ? save + 3 * M : p->fmesh;
        MCPM_TRY(spec_to_delta2_real(p, init_mesh, lap_fd, grad_fd, save ? save + 6 * M : nullptr));
        MCPM_TRY(delta2_to_force_meshes(p, lap_fd, grad_fd, f2));
        StageTimer st_(p, ST_LPT, 12.0 * p->M + 48.0 * p->Np);
        lpt_accum_kernel<<<grid, block, 0, p->stream>>>(p->g, f2, p->M, -g2, -dg2dg, 0, dpos, vel);
        MCPM_LAUNCH_CHECK(p, "lpt_accum_kernel");
    }
    return MCPM_OK;
}

int mcpm_lpt_f32(mcpm_plan *p, const float *init_mesh, int lpt_order, float g, float g2, float dg2dg, int lap_fd,
                 int grad_fd, float *dpos, float *vel) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && dpos && vel, MCPM_E_ARG, "mcpm_lpt_f32: null buffer");
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_lpt_f32: lpt_order must be 1 or 2");
    return lpt_forward(p, init_mesh, lpt_order, g, g2, dg2dg, lap_fd, grad_fd, dpos, vel, nullptr);
}

int mcpm_bullfrog_step_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, double alpha, double beta, double tau,
                           int paint_order, float *force_meshes, float *pos_out, float *vel_out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos_in && vel_in && pos_out && vel_out, MCPM_E_ARG, "mcpm_bullfrog_step_f32: null buffer");
    float *fm = force_meshes ? force_meshes : p->fmesh;
    MCPM_TRY(mcpm_paint_f32(p, pos_in, p->Np, MCPM_POS_LATTICE, nullptr, 1, 1.f, paint_order, p->rho, 0));
    // On plans served by the hand-written Poisson solve the step's force meshes are interleaved [cell][3] (one 12-byte
    // gather per stencil corner in the particle kernels); the checkpoint handed back to the caller is opaque either way.
    const int il = step_layout(p);
    if (il) MCPM_TRY(mcpm_fftpm_force_meshes(p, p->rho, fm, 1));
    else MCPM_TRY(mcpm_force_meshes_f32(p, p->rho, fm));
    MCPM_TRY(mcpm_kick_drift_layout(p, pos_in, vel_in, p->Np, MCPM_POS_LATTICE, fm, il, paint_order, (float)alpha, (float)beta,
                                    (float)tau, pos_out, vel_out));
    return MCPM_OK;
}

int mcpm_bullfrog_step_vjp_from_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, const float *force_meshes,
                                    double alpha, double beta, double tau, int paint_order, const float *pos_bar_src,
                                    const float *vel_bar_src, float *pos_bar, float *vel_bar, double *alpha_bar, double *beta_bar,
                                    double dtau_ddg, double *dg_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos_in && vel_in && force_meshes && pos_bar && vel_bar && pos_bar_src && vel_bar_src, MCPM_E_ARG,
                 "mcpm_bullfrog_step_vjp_f32: null buffer");
    MCPM_REQUIRE(p, paint_order >= 1 && paint_order <= 4, MCPM_E_ORDER, "mcpm_bullfrog_step_vjp_f32: paint_order must be 1..4");
    MCPM_TRY(ensure_pscratch(p));
    const int64_t N = p->Np;
    float *Fb = ps_arr(p, 2);
    const float b = (float)beta, t = (float)tau;
    // force cotangent F_bar = beta (v_bar + tau x_bar), scattered onto three meshes (adjoint of read); already written
    // by the previous call's particle kernel when the caller chained the steps (mcpm_plan_hint_next_adjoint)
    if (!(p->fb_valid && p->fb_beta == b && p->fb_tau == t && p->fb_xb == pos_bar_src && p->fb_vb == vel_bar_src))
        MCPM_TRY(axpby(p, vel_bar_src, pos_bar_src, 3 * N, b, b * t, Fb, true));
    p->fb_valid = 0;
    MCPM_TRY(mcpm_paint3_f32(p, pos_in, N, MCPM_POS_LATTICE, Fb, paint_order, p->fmesh, 0));
    // adjoint of 3 C2R + k-space + R2C: rho_bar = C2R( (1/M) sum_c conj(m_c) R2C(f_bar_c) )
    MCPM_TRY(mcpm_force_meshes_vjp_f32(p, p->fmesh, p->rho));
    return step_adjoint_particles(p, pos_in, vel_in, force_meshes, step_layout(p), p->rho, alpha, beta, tau, paint_order, pos_bar,
                                  vel_bar, alpha_bar, beta_bar, dtau_ddg, dg_bar, pos_bar_src, vel_bar_src);
}

int mcpm_bullfrog_step_vjp_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, const float *force_meshes,
                               double alpha, double beta, double tau, int paint_order, float *pos_bar, float *vel_bar,
                               double *alpha_bar, double *beta_bar, double dtau_ddg, double *dg_bar) {
    return mcpm_bullfrog_step_vjp_from_f32(p, pos_in, vel_in, force_meshes, alpha, beta, tau, paint_order, pos_bar, vel_bar, pos_bar, vel_bar,
                                           alpha_bar, beta_bar, dtau_ddg, dg_bar);
}

// The force cotangent F_bar = beta (v_bar + tau x_bar) of the NEXT adjoint step, if the previous
// mcpm_step_adjoint_particles_f32 call was hinted (mcpm_plan_hint_next_adjoint) with exactly these (beta, tau) and
// cotangent arrays: *fb then points at plan-owned memory (Np x 3 floats, valid until the next hinted call) and the caller
// can skip forming it.  *fb = NULL otherwise.  Used by callers that compose the adjoint step themselves (slabs).
int mcpm_axpby_f32(mcpm_plan *p, const float *x, const float *y, int64_t n, float a, float b, float *out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, x && y && out && n >= 0, MCPM_E_ARG, "mcpm_axpby_f32: bad argument");
    if (n == 0) return MCPM_OK;
    return axpby(p, x, y, n, a, b, out);
}

int mcpm_plan_chained_fb(mcpm_plan *p, double beta, double tau, const float *pos_bar, const float *vel_bar, float **fb) {
    if (!p || !fb) return MCPM_E_ARG;
    *fb = nullptr;
    if (p->fb_valid && p->fb_beta == (float)beta && p->fb_tau == (float)tau && p->fb_xb == pos_bar && p->fb_vb == vel_bar && p->pscratch)
        *fb = ps_arr(p, 2);
    p->fb_valid = 0;
    return MCPM_OK;
}

int mcpm_step_adjoint_particles_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, const float *force_meshes,
                                    const float *rho_bar, double alpha, double beta, double tau, int paint_order,
                                    float *pos_bar, float *vel_bar, double *alpha_bar, double *beta_bar, double dtau_ddg,
                                    double *dg_bar) {
    return step_adjoint_particles(p, pos_in, vel_in, force_meshes, 0, rho_bar, alpha, beta, tau, paint_order, pos_bar, vel_bar,
                                  alpha_bar, beta_bar, dtau_ddg, dg_bar, nullptr, nullptr);
}

}  // extern "C"

// layout 0: three force meshes M apart (public entry point); 1: interleaved [cell][3] (what the steppers checkpoint on
// plans served by the hand-written Poisson solve)
extern "C" int mcpm_step_adjoint_particles_il_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, const float *force_mesh_il,
                                                  const float *rho_bar, double alpha, double beta, double tau, int paint_order,
                                                  float *pos_bar, float *vel_bar, double *alpha_bar, double *beta_bar, double dtau_ddg,
                                                  double *dg_bar) {
    return step_adjoint_particles(p, pos_in, vel_in, force_mesh_il, 1, rho_bar, alpha, beta, tau, paint_order, pos_bar, vel_bar,
                                  alpha_bar, beta_bar, dtau_ddg, dg_bar, nullptr, nullptr);
}

// pos_bar_src / vel_bar_src: where the incoming cotangents are read (NULL: pos_bar / vel_bar, in place)
static int step_adjoint_particles(mcpm_plan *p, const float *pos_in, const float *vel_in, const float *force_meshes, int layout,
                                  const float *rho_bar, double alpha, double beta, double tau, int paint_order, float *pos_bar,
                                  float *vel_bar, double *alpha_bar, double *beta_bar, double dtau_ddg, double *dg_bar,
                                  const float *pos_bar_src, const float *vel_bar_src) {
    if (!p) return MCPM_E_ARG;
    if (!pos_bar_src) pos_bar_src = pos_bar;
    if (!vel_bar_src) vel_bar_src = vel_bar;
    MCPM_REQUIRE(p, pos_in && vel_in && force_meshes && rho_bar && pos_bar && vel_bar, MCPM_E_ARG,
                 "mcpm_step_adjoint_particles_f32: null buffer");
    MCPM_REQUIRE(p, paint_order >= 1 && paint_order <= 4, MCPM_E_ORDER, "mcpm_step_adjoint_particles_f32: paint_order must be 1..4");
    const int64_t N = p->Np, M = p->M;
    const float a = (float)alpha, b = (float)beta, t = (float)tau;
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    // algorithmic bytes: x, v, x_bar, v_bar in (48 N), x_bar, v_bar out (24 N), three force meshes + rho_bar gathered once (16 M); in a
    // chained reverse sweep the kernel also WRITES the next step's force cotangent (12 N: the axpby pass it replaces)
    StageTimer st_(p, ST_STEPADJ, 72.0 * N + 16.0 * M + (p->hint_set ? 12.0 * N : 0.0));
    float *fb_next = nullptr;
    if (p->hint_set) {
        MCPM_TRY(ensure_pscratch(p));
        fb_next = ps_arr(p, 2);
        p->fb_valid = 1;
        p->fb_beta = p->hint_beta;
        p->fb_tau = p->hint_tau;
        p->fb_xb = pos_bar;
        p->fb_vb = vel_bar;
        p->hint_set = 0;
        if (p->paint3_variant == 4) {   // the fixed-point paint of fb_next needs max|fb_next| (slots zero after a tiled paint3)
            if (!p->fx_clean) MCPM_HIP(p, hipMemsetAsync(p->fx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream));
            p->fx_clean = 0;
            p->fx_src = fb_next;
        }
    }
    unsigned *fb_max = (fb_next && p->paint3_variant == 4) ? p->fx_wmax : nullptr;
    static const int ntp_env = [] { const char *e = getenv("MCPM_NT_PART"); return e ? atoi(e) : 3; }();     // streaming loads / stores: 2.80 -> 2.62 ms at 512^3
    const int ntp = N < ((int64_t)1 << 23) ? 0 : ntp_env;      // not for problems that live in the caches
    double *P, *Q;      // this launch's per-workgroup partials of (alpha_bar, beta_bar, dg_bar)
    unsigned *ticket, R;
    MCPM_TRY(mcpm_det_scratch(p, 3, grid.x, &P, &Q, &ticket, &R));
#define ADJ(OR)                                                                                                                   \
    if (layout) step_adjoint_kernel<OR, true><<<grid, block, 0, p->stream>>>(p->g, pos_in, vel_in, pos_bar_src, vel_bar_src, pos_bar, vel_bar, force_meshes, rho_bar, M, a, b, t, \
                                                           P, fb_next, p->hint_beta, p->hint_tau, (float)dtau_ddg, fb_max, ntp);  \
    else step_adjoint_kernel<OR, false><<<grid, block, 0, p->stream>>>(p->g, pos_in, vel_in, pos_bar_src, vel_bar_src, pos_bar, vel_bar, force_meshes, rho_bar, M, a, b, t, \
                                                           P, fb_next, p->hint_beta, p->hint_tau, (float)dtau_ddg, fb_max, ntp)
    if (paint_order == 2) ADJ(2);
    else if (paint_order == 1) ADJ(1);
    else if (paint_order == 3) ADJ(3);
    else ADJ(4);
#undef ADJ
    MCPM_LAUNCH_CHECK(p, "step_adjoint_kernel");
    if (alpha_bar || beta_bar || dg_bar) {
        MCPM_TRY(det_fold(p, P, Q, ticket, R, grid.x, 3, 1.0, alpha_bar, beta_bar, dg_bar));
    }
    return MCPM_OK;
}

extern "C" {

int mcpm_lpt_vjp_f32(mcpm_plan *p, const float *init_mesh, int lpt_order, const double *lpt_scalars, const float *dpos_bar,
                     const float *vel_bar, float *init_mesh_bar, double *scalar_bars) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && lpt_scalars && dpos_bar && vel_bar && init_mesh_bar, MCPM_E_ARG, "mcpm_lpt_vjp_f32: null argument");
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_lpt_vjp_f32: lpt_order must be 1 or 2");
    MCPM_HIP(p, hipMemsetAsync(p->reduce, 0, sizeof(double) * 3, p->stream));
    MCPM_TRY(lpt_vjp_device(p, init_mesh, lpt_order, lpt_scalars, dpos_bar, vel_bar, init_mesh_bar, p->reduce));
    if (scalar_bars) {
        MCPM_HIP(p, hipMemcpyAsync(scalar_bars, p->reduce, sizeof(double) * 3, hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipStreamSynchronize(p->stream));
        scalar_bars[1] = -scalar_bars[1];
        scalar_bars[2] = -scalar_bars[2];
    }
    return MCPM_OK;
}

// mcpm_lpt_f32 that also leaves its force and Hessian meshes in `save` (3 M floats for lpt_order 1, 12 M for 2), and the adjoint that
// reads them there instead of recomputing them (infinite-order kernels: what the model uses)
int mcpm_lpt_save_f32(mcpm_plan *p, const float *init_mesh, int lpt_order, float g, float g2, float dg2dg, float *dpos, float *vel,
                      float *save) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && dpos && vel && save, MCPM_E_ARG, "mcpm_lpt_save_f32: null buffer");
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_lpt_save_f32: lpt_order must be 1 or 2");
    return lpt_forward(p, init_mesh, lpt_order, g, g2, dg2dg, MCPM_FD_INF, MCPM_FD_INF, dpos, vel, save);
}

int mcpm_lpt_vjp_saved_f32(mcpm_plan *p, const float *init_mesh, int lpt_order, const double *lpt_scalars, const float *saved,
                           const float *dpos_bar, const float *vel_bar, float *init_mesh_bar, double *scalar_bars) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && lpt_scalars && saved && dpos_bar && vel_bar && init_mesh_bar, MCPM_E_ARG, "mcpm_lpt_vjp_saved_f32: null argument");
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_lpt_vjp_saved_f32: lpt_order must be 1 or 2");
    MCPM_HIP(p, hipMemsetAsync(p->reduce, 0, sizeof(double) * 3, p->stream));
    MCPM_TRY(lpt_vjp_device(p, init_mesh, lpt_order, lpt_scalars, dpos_bar, vel_bar, init_mesh_bar, p->reduce, MCPM_FD_INF, MCPM_FD_INF, saved));
    if (scalar_bars) {
        MCPM_HIP(p, hipMemcpyAsync(scalar_bars, p->reduce, sizeof(double) * 3, hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipStreamSynchronize(p->stream));
        scalar_bars[1] = -scalar_bars[1];
        scalar_bars[2] = -scalar_bars[2];
    }
    return MCPM_OK;
}

int mcpm_lpt_vjp_opts_f32(mcpm_plan *p, const float *init_mesh, int lpt_order, const double *lpt_scalars, int lap_fd, int grad_fd,
                          const float *dpos_bar, const float *vel_bar, float *init_mesh_bar, double *scalar_bars) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && lpt_scalars && dpos_bar && vel_bar && init_mesh_bar, MCPM_E_ARG, "mcpm_lpt_vjp_opts_f32: null argument");
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_lpt_vjp_opts_f32: lpt_order must be 1 or 2");
    MCPM_HIP(p, hipMemsetAsync(p->reduce, 0, sizeof(double) * 3, p->stream));
    MCPM_TRY(lpt_vjp_device(p, init_mesh, lpt_order, lpt_scalars, dpos_bar, vel_bar, init_mesh_bar, p->reduce, lap_fd, grad_fd));
    if (scalar_bars) {
        MCPM_HIP(p, hipMemcpyAsync(scalar_bars, p->reduce, sizeof(double) * 3, hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipStreamSynchronize(p->stream));
        scalar_bars[1] = -scalar_bars[1];
        scalar_bars[2] = -scalar_bars[2];
    }
    return MCPM_OK;
}

// VJP of pm_forces(pos, mesh_shape, order, paint_deconv, grad_fd, lap_fd) w.r.t. pos (painted case, nbody.py:583-604)
int mcpm_pm_forces_vjp_opts_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, int order, int paint_deconv, int lap_fd,
                                int grad_fd, const float *forces_bar, float *pos_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos && forces_bar && pos_bar, MCPM_E_ARG, "mcpm_pm_forces_vjp_opts_f32: null buffer");
    const int64_t M = p->M;
    const int deconv = paint_deconv ? order : 0;
    float *fm = p->fmesh, *fb = p->fmesh + 3 * M;
    MCPM_TRY(mcpm_paint_f32(p, pos, n, mode, nullptr, 1, 1.f, order, p->rho, 0));
    MCPM_TRY(mcpm_fft_r2c(p, p->rho, p->spec1, 1));
    MCPM_TRY(spec_to_force_meshes(p, p->spec1, lap_fd, grad_fd, 0.f, deconv, fm));
    MCPM_TRY(mcpm_read_vjp_pos_f32(p, pos, n, mode, fm, 3, order, forces_bar, pos_bar));
    MCPM_TRY(mcpm_paint3_f32(p, pos, n, mode, forces_bar, order, fb, 0));
    MCPM_TRY(force_meshes_vjp_opts(p, fb, p->rho, lap_fd, grad_fd, deconv));
    MCPM_TRY(ensure_pscratch_n(p, 3 * n));
    float *pb2 = p->vscratch;
    MCPM_TRY(mcpm_paint_vjp_f32(p, pos, n, mode, nullptr, 1, 1.f, order, p->rho, pb2, nullptr));
    return axpby(p, pos_bar, pb2, 3 * n, 1.f, 1.f, pos_bar);
}

// checkpoint layout: states (x'_i, v_i) as arrays 2 i and 2 i + 1, mcpm_pitch() floats apart, in a region sized for the largest
// candidate pitch (so that the buffer does not depend on the pitch chosen later); then the n_steps force meshes
// ... then what the LPT start leaves for its adjoint: the first-order force meshes (3 M), and with lpt_order = 2 the second-order ones
// (3 M) and the six Hessian meshes (6 M) -- lpt_forward(save) / lpt_vjp_device(saved)
static int64_t ckpt_lpt_offset(const mcpm_plan *p, int n_steps) { return (int64_t)n_steps * (2 * mcpm_pitch_max(p) + 3 * p->M); }
int64_t mcpm_nbody_ckpt_floats(const mcpm_plan *p, int n_steps, int lpt_order) {
    if (!p || n_steps < 1) return 0;
    return ckpt_lpt_offset(p, n_steps) + (lpt_order == 2 ? 12 : 3) * p->M;
}

int mcpm_plan_set_particle_pitch(mcpm_plan *p, int64_t pitch_floats) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pitch_floats == 0 || (pitch_floats >= 3 * p->Np && pitch_floats <= mcpm_pitch_max(p) && pitch_floats % 4 == 0), MCPM_E_ARG,
                 "mcpm_plan_set_particle_pitch: 0, or 3 N <= pitch <= 3 N + 17472, a multiple of 4 floats");
    p->ppitch = pitch_floats;
    p->fb_valid = 0;
    return MCPM_OK;
}

int mcpm_plan_particle_pitch(const mcpm_plan *p, int64_t *pitch_floats) {
    if (!p || !pitch_floats) return MCPM_E_ARG;
    *pitch_floats = mcpm_pitch(p);
    return MCPM_OK;
}

// Times the adjoint particle kernel (the one whose speed follows the placement: 2.45 or 2.75 ms at 512^3) with the checkpoint
// arrays of `flat` and the plan's own cotangent scratch laid out at each candidate pitch, and keeps the fastest.  The buffer is
// zeroed (zero displacements are valid particles); call it on the buffer that will hold the checkpoints, before filling it.
int mcpm_plan_probe_particle_pitch(mcpm_plan *p, float *flat, int64_t flat_floats, int64_t *pitch_floats) {
    if (!p || !flat) return MCPM_E_ARG;
    const int64_t N = p->Np, M = p->M;
    int64_t best = 3 * N;
    if (N >= ((int64_t)1 << 23) && p->g.same_lattice && flat_floats >= 2 * mcpm_pitch_max(p)) {
        MCPM_TRY(ensure_pscratch(p));
        static const int64_t shifts[3] = {0, 1088, MCPM_PITCH_MAX_SHIFT};      // in phase; 4 KB + 256 B; 64 KB + 4 KB + 256 B
        hipEvent_t e0, e1;
        MCPM_HIP(p, hipEventCreate(&e0));
        MCPM_HIP(p, hipEventCreate(&e1));
        const int64_t keep = p->ppitch;
        float best_ms = 0.f;
        int rc = MCPM_OK;
        // force mesh and rho_bar: the plan's scratch meshes, zeroed (the kernel's gathers then read finite numbers)
        if (hipMemsetAsync(p->fmesh, 0, sizeof(float) * 3 * M, p->stream) != hipSuccess || hipMemsetAsync(p->rho, 0, sizeof(float) * M, p->stream) != hipSuccess)
            rc = mcpm_fail(p, MCPM_E_HIP, "mcpm_plan_probe_particle_pitch: memset");
        for (int c = 0; c < 3 && rc == MCPM_OK; ++c) {
            p->ppitch = 3 * N + shifts[c];
            const int64_t pitch = p->ppitch;
            (void)hipMemsetAsync(flat, 0, sizeof(float) * 2 * pitch, p->stream);
            (void)hipMemsetAsync(p->pscratch, 0, sizeof(float) * 3 * pitch, p->stream);
            for (int rep = 0; rep < 4 && rc == MCPM_OK; ++rep) {      // the first call warms up
                if (rep == 1) (void)hipEventRecord(e0, p->stream);
                p->hint_set = 1, p->hint_beta = 0.f, p->hint_tau = 0.f;      // as inside a reverse sweep: the kernel also writes F_bar
                rc = step_adjoint_particles(p, flat, flat + pitch, p->fmesh, step_layout(p), p->rho, 1.0, 0.0, 0.0, 2, ps_arr(p, 0), ps_arr(p, 1),
                                            nullptr, nullptr, 1.0, nullptr, nullptr, nullptr);
            }
            if (rc != MCPM_OK) break;
            (void)hipEventRecord(e1, p->stream);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = mcpm_fail(p, MCPM_E_HIP, "mcpm_plan_probe_particle_pitch: event"); break; }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (c == 0 || ms < best_ms) best_ms = ms, best = pitch;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        p->fb_valid = 0;
        p->fx_src = nullptr;
        p->ppitch = keep;
        if (rc != MCPM_OK) return rc;
    }
    p->ppitch = best == 3 * N ? 0 : best;
    if (pitch_floats) *pitch_floats = best;
    return MCPM_OK;
}

int mcpm_nbody_bf_f32(mcpm_plan *p, const float *init_mesh, int n_steps, const double *alpha, const double *beta,
                      double dg, const double *lpt_scalars, int lpt_order, int paint_order, float *pos_out,
                      float *vel_out, float *ckpt) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && alpha && beta && lpt_scalars && pos_out && vel_out, MCPM_E_ARG, "mcpm_nbody_bf_f32: null argument");
    MCPM_REQUIRE(p, n_steps >= 1, MCPM_E_ARG, "mcpm_nbody_bf_f32: n_steps must be >= 1");
    MCPM_REQUIRE(p, paint_order >= 1 && paint_order <= 4, MCPM_E_ORDER, "mcpm_nbody_bf_f32: paint_order must be 1..4");
    const int64_t N = p->Np, M = p->M;
    const int64_t pitch = mcpm_pitch(p);
    auto state_x = [&](int i) { return ckpt + (int64_t)(2 * i) * pitch; };
    auto state_v = [&](int i) { return ckpt + (int64_t)(2 * i + 1) * pitch; };
    auto force_m = [&](int i) { return ckpt + (int64_t)n_steps * 2 * mcpm_pitch_max(p) + (int64_t)i * 3 * M; };
    float *x = ckpt ? state_x(0) : pos_out, *v = ckpt ? state_v(0) : vel_out;
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_nbody_bf_f32: lpt_order must be 1 or 2");
    MCPM_TRY(lpt_forward(p, init_mesh, lpt_order, (float)lpt_scalars[0], (float)lpt_scalars[1], (float)lpt_scalars[2], MCPM_FD_INF,
                         MCPM_FD_INF, x, v, ckpt ? ckpt + ckpt_lpt_offset(p, n_steps) : nullptr));
    MCPM_TRY(mcpm_drift_f32(p, x, v, N, (float)(dg / 2), x));
    for (int i = 0; i < n_steps; ++i) {
        const bool last = (i == n_steps - 1);
        float *xn = (ckpt && !last) ? state_x(i + 1) : pos_out, *vn = (ckpt && !last) ? state_v(i + 1) : vel_out;
        MCPM_TRY(mcpm_bullfrog_step_f32(p, x, v, alpha[i], beta[i], last ? dg / 2 : dg, paint_order,
                                        ckpt ? force_m(i) : nullptr, xn, vn));
        x = xn;
        v = vn;
    }
    return MCPM_OK;
}

int mcpm_nbody_bf_vjp_f32(mcpm_plan *p, const float *init_mesh, int n_steps, const double *alpha, const double *beta,
                          double dg, const double *lpt_scalars, int lpt_order, int paint_order, const float *ckpt,
                          const float *pos_bar, const float *vel_bar, float *init_mesh_bar, double *scalar_bars) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, init_mesh && alpha && beta && lpt_scalars && ckpt && pos_bar && vel_bar && init_mesh_bar, MCPM_E_ARG,
                 "mcpm_nbody_bf_vjp_f32: null argument");
    MCPM_REQUIRE(p, n_steps >= 1 && 2 * n_steps + 4 <= MCPM_NREDUCE, MCPM_E_ARG, "mcpm_nbody_bf_vjp_f32: bad n_steps");
    MCPM_REQUIRE(p, paint_order >= 1 && paint_order <= 4, MCPM_E_ORDER, "mcpm_nbody_bf_vjp_f32: paint_order must be 1..4");
    MCPM_REQUIRE(p, lpt_order == 1 || lpt_order == 2, MCPM_E_ORDER, "mcpm_nbody_bf_vjp_f32: lpt_order must be 1 or 2");
    const int64_t N = p->Np, M = p->M;
    MCPM_TRY(ensure_pscratch(p));
    float *xb = ps_arr(p, 0), *vb = ps_arr(p, 1);
    const int64_t pitch = mcpm_pitch(p);
    auto state_x = [&](int i) { return ckpt + (int64_t)(2 * i) * pitch; };
    auto state_v = [&](int i) { return ckpt + (int64_t)(2 * i + 1) * pitch; };
    auto force_m = [&](int i) { return ckpt + (int64_t)n_steps * 2 * mcpm_pitch_max(p) + (int64_t)i * 3 * M; };
    MCPM_HIP(p, hipMemsetAsync(p->reduce, 0, sizeof(double) * (2 * n_steps + 4 + 256), p->stream));
    p->fb_valid = 0;
    for (int i = n_steps - 1; i >= 0; --i) {
        if (i > 0) MCPM_TRY(mcpm_plan_hint_next_adjoint(p, beta[i - 1], dg));
        // the first reverse step READS the loss cotangents where the caller has them and writes the running ones (no copy of 24 N bytes)
        const bool first = i == n_steps - 1;
        MCPM_TRY(mcpm_bullfrog_step_vjp_from_f32(p, state_x(i), state_v(i), force_m(i), alpha[i], beta[i], first ? dg / 2 : dg, paint_order,
                                                 first ? pos_bar : xb, first ? vel_bar : vb, xb, vb, p->reduce + i, p->reduce + n_steps + i,
                                                 first ? 0.5 : 1.0, p->reduce + 2 * n_steps + 3));
    }
    // initial half drift x'_0 = x_0 + v_0 dg/2 (its explicit dg dependence: <x_bar, v_0> / 2)
    {
        unsigned nbk = (unsigned)((3 * N + 255) / 256);
        double *P, *Q;
        unsigned *ticket, R;
        MCPM_TRY(mcpm_det_scratch(p, 1, nbk, &P, &Q, &ticket, &R));
        dot_partial_kernel<<<nbk, 256, 0, p->stream>>>(xb, state_v(0), 3 * N, P);
        MCPM_LAUNCH_CHECK(p, "dot_partial_kernel");
        MCPM_TRY(det_fold(p, P, Q, ticket, R, nbk, 1, 0.5, p->reduce + 2 * n_steps + 3));
    }
    MCPM_TRY(axpby(p, vb, xb, 3 * N, 1.f, (float)(dg / 2), vb));

    // ---- adjoint of lpt (nbody.py:634-667) at the lattice, read_order = 1
    MCPM_TRY(lpt_vjp_device(p, init_mesh, lpt_order, lpt_scalars, xb, vb, init_mesh_bar, p->reduce + 2 * n_steps, MCPM_FD_INF, MCPM_FD_INF,
                            ckpt + ckpt_lpt_offset(p, n_steps)));
    if (scalar_bars) {
        MCPM_HIP(p, hipMemcpyAsync(scalar_bars, p->reduce, sizeof(double) * (2 * n_steps + 4), hipMemcpyDeviceToHost, p->stream));
        MCPM_HIP(p, hipStreamSynchronize(p->stream));
        scalar_bars[2 * n_steps + 1] = -scalar_bars[2 * n_steps + 1];
        scalar_bars[2 * n_steps + 2] = -scalar_bars[2 * n_steps + 2];
    }
    return MCPM_OK;
}

}  // extern "C"
