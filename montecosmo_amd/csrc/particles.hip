// Particle <-> mesh kernels of libmcpm.so for gfx950: paint, read, fused kick/drift and their VJPs.
//
// Reference semantics: montecosmo/nbody.py:365-396 (paint), :398-427 (read), :933-944 (kick, drift).
//
// Paint design (MI355X-first, not a translation of the reference's 8 scatter-add passes):
//   * fast path (paint_tiled.hip): particles are stored in Lagrangian (lattice) order as fp32 displacements from their
//     lattice point; one workgroup owns one Eulerian tile of the mesh in LDS and *pulls* the lattice particles that can land
//     in it, through a window centred on the local bulk displacement.
//   * generic path `paint_atomic_kernel` (here): arbitrary (absolute) positions, any order, any mesh; fixed-point integer
//     global atomics (order-independent).
#include "particles_dev.h"

// ------------------------------------------------------------------------------------------------
// cell index (integer part of the path, checked bit-exactly against the oracle)
template <int MODE, int ORDER>
__global__ __launch_bounds__(256) void cell_index_kernel(Geom g, const float *__restrict__ pos, int64_t n,
                                                         int16_t *__restrict__ idx) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    P3 d = load3(pos, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    idx[3 * pi.i + 0] = (int16_t)(g.xslab ? c[0] : wrapi(c[0], g.nx));
    idx[3 * pi.i + 1] = (int16_t)wrapi(c[1], g.ny);
    idx[3 * pi.i + 2] = (int16_t)wrapi(c[2], g.nz);
}

// ------------------------------------------------------------------------------------------------
// generic paint: one thread per particle, arbitrary positions / NGP / TSC / PCS / any mesh size.
// Deposits are ORDER-INDEPENDENT: every contribution w * kx * ky * kz (an f32 product, as in the tiled kernels) is
// rounded once to fixed point with the power-of-two scale S = 2^(q - e), 2^e <= max|w| < 2^(e+1), and added with a
// 64-bit INTEGER global atomic into the plan's accumulator mesh `acc` (int64 per cell, kept all-zero between calls);
// paint_fxg_flush_kernel then adds acc / S to the f32 mesh and zeroes acc again.  Integer sums are exact, so two
// launches give bitwise identical meshes whatever the arrival order (f32 float atomics did not: a last-bit change of
// the density moved a particle across a cell face a few steps later).  q = min(40, 61 - ceil(log2(n ORDER^3))) keeps
// the sum of |contribution| S below 2^62 even if every particle lands in one cell; one deposit is rounded by at most
// 2^-(q+1) max|w| <= 2^-25 max|w| (n < 2^31, PCS), below the f32 rounding of the product itself.
// Non-finite weights (max|w| = inf / NaN) fall back to f32 float atomics so that NaN / inf reach the mesh as before.
struct FxgScale {
    double S, Sinv;   // 0 / 0: all weights are zero, nothing to deposit
    bool flt;         // non-finite weights: f32 float atomics straight into the mesh
};
__device__ __forceinline__ FxgScale fxg_scale(const unsigned *__restrict__ wmax_bits, int q) {
    unsigned wb = wmax_bits[(threadIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE];   // maximum over the slots, in every wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wb = max(wb, (unsigned)__shfl_xor((int)wb, o));
    FxgScale r;
    int be = (int)(wb >> 23);
    r.flt = be >= 255;
    if (wb == 0u || r.flt) {
        r.S = r.Sinv = 0.;
        return r;
    }
    if (be == 0) be = 1;   // subnormal maximum
    const int e = be - 127;
    r.S = __longlong_as_double((long long)(1023 + q - e) << 52);
    r.Sinv = __longlong_as_double((long long)(1023 - q + e) << 52);
    return r;
}

__global__ __launch_bounds__(256) void absmax_strided_kernel(const float *__restrict__ w, int64_t stride, int64_t n,
                                                             unsigned *__restrict__ out) {
    float m = 0.f;
    unsigned bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned b = __float_as_uint(w[i * stride]) & 0x7fffffffu;
        bad |= b >= 0x7f800000u;
        m = fmaxf(m, __uint_as_float(b));
    }
    unsigned b = bad ? 0x7fc00000u : __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
    if ((threadIdx.x & 63) == 0 && b) atomicMax(out + (blockIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE, b);
}

__global__ __launch_bounds__(64) void fxg_set_unit_kernel(unsigned *__restrict__ out) {   // max|w| = 1 (unweighted paint)
    out[threadIdx.x * MCPM_FX_STRIDE] = 0x3f800000u;
}

template <int MODE, int ORDER>
__global__ __launch_bounds__(256) void paint_atomic_kernel(Geom g, const float *__restrict__ pos, int64_t n,
                                                           const float *__restrict__ w, int64_t wstride,
                                                           float *__restrict__ mesh, unsigned long long *__restrict__ acc,
                                                           const unsigned *__restrict__ wmax_bits, int q, int *__restrict__ oob) {
    const FxgScale sc = fxg_scale(wmax_bits, q);
    if (sc.S == 0. && !sc.flt) return;
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    P3 d = load3(pos, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    if (g.xslab && (c[0] < 0 || c[0] > g.nx - ORDER)) atomicAdd(oob, 1);  // beyond the ghost planes: clamped + counted
    const float wt = w ? w[pi.i * wstride] : 1.f;   // unweighted: the scalar weight is applied by the flush
    Stencil<ORDER> s(g, c);
    auto deposit = [&](int64_t cell, float v) {
        if (sc.flt) atomicAdd(mesh + cell, v);
        else atomicAdd(acc + cell, (unsigned long long)__double2ll_rn((double)v * sc.S));
    };
    if (ORDER == 1) {
        deposit(s.xo[0] + s.yo[0] + s.zo[0], wt);
        return;
    }
    if (ORDER >= 3) {
        constexpr int NPG = ORDER < 3 ? 3 : ORDER;
        float wx[NPG], wy[NPG], wz[NPG], dd[NPG];
        axis_weights<NPG, false>(f[0], wx, dd);
        axis_weights<NPG, false>(f[1], wy, dd);
        axis_weights<NPG, false>(f[2], wz, dd);
#pragma unroll
        for (int a = 0; a < NPG; ++a)
#pragma unroll
            for (int b = 0; b < NPG; ++b)
#pragma unroll
                for (int e = 0; e < NPG; ++e) deposit(s.xo[a] + s.yo[b] + s.zo[e], wt * wx[a] * wy[b] * wz[e]);
        return;
    }
    float kx[2] = {1.f - f[0], f[0]}, ky[2] = {1.f - f[1], f[1]}, kz[2] = {1.f - f[2], f[2]};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 2; ++e) deposit(s.xo[a] + s.yo[b] + s.zo[e], wt * kx[a] * ky[b] * kz[e]);
}

// mesh += wscalar * acc / S; acc = 0 (the accumulator is all-zero again for the next paint)
__global__ __launch_bounds__(256) void paint_fxg_flush_kernel(long long *__restrict__ acc, float *__restrict__ mesh, int64_t M,
                                                              const unsigned *__restrict__ wmax_bits, int q, float wscalar,
                                                              int vec) {
    const FxgScale sc = fxg_scale(wmax_bits, q);
    if (sc.S == 0.) return;   // nothing was deposited in fixed point
    const double s = sc.Sinv * (double)wscalar;
    if (!vec) {   // mesh pointer not 8-byte aligned
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
            const long long a = acc[i];
            if (a) {
                mesh[i] += (float)((double)a * s);
                acc[i] = 0;
            }
        }
        return;
    }
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < M; i += (int64_t)gridDim.x * 512) {
        if (i + 1 < M) {
            longlong2 a = *reinterpret_cast<longlong2 *>(acc + i);
            if (a.x | a.y) {
                float2 m = *reinterpret_cast<float2 *>(mesh + i);
                m.x += (float)((double)a.x * s);
                m.y += (float)((double)a.y * s);
                *reinterpret_cast<float2 *>(mesh + i) = m;
                *reinterpret_cast<longlong2 *>(acc + i) = make_longlong2(0, 0);
            }
        } else {
            const long long a = acc[i];
            if (a) {
                mesh[i] += (float)((double)a * s);
                acc[i] = 0;
            }
        }
    }
}

// Adjoint of the NGP lattice read on a lattice that is not the mesh (several lattice points per cell): component C of
// out[cell(i)] += a*xb[i] + b*vb[i], through the same fixed-point accumulator as the generic paint.
__global__ __launch_bounds__(256) void absmax_axpby3_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t n3,
                                                            float a, float b, unsigned *__restrict__ out) {
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n3; i += (int64_t)gridDim.x * 256) {
        const unsigned v = __float_as_uint(a * x[i] + b * y[i]) & 0x7fffffffu;
        m = max(m, v >= 0x7f800000u ? 0x7fc00000u : v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out + (blockIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE, m);
}

__global__ __launch_bounds__(256) void lattice_scatter_fx_kernel(Geom g, const float *__restrict__ xb, const float *__restrict__ vb,
                                                                 float a, float b, int comp, float *__restrict__ out,
                                                                 unsigned long long *__restrict__ acc,
                                                                 const unsigned *__restrict__ wmax_bits, int q) {
    const FxgScale sc = fxg_scale(wmax_bits, q);
    if (sc.S == 0. && !sc.flt) return;
    PIdx pi = particle_index<MCPM_POS_LATTICE>(g, 0);
    if (!pi.valid) return;
    const int64_t c = lattice_cell(g, pi);
    const float o = a * xb[3 * pi.i + comp] + b * vb[3 * pi.i + comp];
    if (sc.flt) atomicAdd(out + c, o);
    else atomicAdd(acc + c, (unsigned long long)__double2ll_rn((double)o * sc.S));
}

// ------------------------------------------------------------------------------------------------
// Kaiser-Bessel assignment (nbody.py:280-290 `kaiser_bessel`, selected by kernel_type in paint :381-382 and read :411-412):
//     K(s) = I0(kc sqrt(1 - (2 s / order)^2)) kc / (order sinh(kc)),   kc = kcut order / 2,
// on the same order^3 stencil and base cell (floor / round-half-even) as the rectangular kernels.  Any order 1..4 at run
// time, absolute or lattice positions; not a hot path (the model's default is kernel_type = 'rectangular', model.py:52).
__device__ __forceinline__ float kb_i0(float x) {          // sum_k (x^2/4)^k / (k!)^2: positive terms, no cancellation
    const float t = 0.25f * x * x;
    float term = 1.f, sum = 1.f;
    for (int k = 1; k < 64; ++k) {
        term *= t / (float)(k * k);
        sum += term;
        if (term < 1e-9f * sum) break;
    }
    return sum;
}
__device__ __forceinline__ float kb_i1_over_x(float x) {   // I1(x) / x = sum_k (x^2/4)^k / (2 k! (k+1)!)
    const float t = 0.25f * x * x;
    float term = 0.5f, sum = 0.5f;
    for (int k = 1; k < 64; ++k) {
        term *= t / (float)(k * (k + 1));
        sum += term;
        if (term < 1e-9f * sum) break;
    }
    return sum;
}
// per-axis weights w[j] = K(idx_j - pos) and dw[j] = d/dpos K(idx_j - pos) of the `order` stencil points, f = pos - id0
__device__ __forceinline__ void kb_axis(float f, int order, float kc, float ninv, float (&w)[4], float (&dw)[4]) {
    const int sh = -((order - 1) / 2);
    const float so = 2.f / (float)order;
    for (int j = 0; j < order; ++j) {
        const float sp = ((float)(sh + j) - f) * so;
        const float z = kc * sqrtf(fmaxf(1.f - sp * sp, 0.f));
        w[j] = kb_i0(z) * ninv;
        dw[j] = ninv * kc * kc * sp * so * kb_i1_over_x(z);      // -dK/ds,  I1(z) dz/ds = -kc^2 sp (2/order) I1(z)/z
    }
}
struct KbStencil {
    int64_t xo[4], yo[4];
    int zo[4];
};
template <int MODE>
__device__ __forceinline__ void kb_locate(const Geom &g, const PIdx &pi, P3 d, int order, KbStencil &s, float (&f)[3]) {
    int c[3];
    if (order & 1) locate<MODE, 3>(g, pi, d, c, f);      // round-half-even, f in [-1/2, 1/2]
    else locate<MODE, 2>(g, pi, d, c, f);                 // floor, f in [0, 1)
    const int sh = -((order - 1) / 2);
    for (int j = 0; j < order; ++j) {
        const int x = g.xslab ? min(max(c[0] + sh + j, 0), g.nx - 1) : wrapi(c[0] + sh + j, g.nx);
        s.xo[j] = (int64_t)x * g.ny * g.nz;
        s.yo[j] = (int64_t)wrapi(c[1] + sh + j, g.ny) * g.nz;
        s.zo[j] = wrapi(c[2] + sh + j, g.nz);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void paint_kb_kernel(Geom g, const float *__restrict__ pos, int64_t n, const float *__restrict__ w,
                                                       int64_t wstride, float *__restrict__ mesh, unsigned long long *__restrict__ acc,
                                                       const unsigned *__restrict__ wmax_bits, int q, int order, float kc, float ninv) {
    const FxgScale sc = fxg_scale(wmax_bits, q);
    if (sc.S == 0. && !sc.flt) return;
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    KbStencil s;
    float f[3], wx[4], wy[4], wz[4], dd[4];
    kb_locate<MODE>(g, pi, load3(pos, pi.i), order, s, f);
    kb_axis(f[0], order, kc, ninv, wx, dd);
    kb_axis(f[1], order, kc, ninv, wy, dd);
    kb_axis(f[2], order, kc, ninv, wz, dd);
    const float wt = w ? w[pi.i * wstride] : 1.f;
    for (int a = 0; a < order; ++a)
        for (int b = 0; b < order; ++b)
            for (int e = 0; e < order; ++e) {
                const float v = wt * wx[a] * wy[b] * wz[e];
                const int64_t cell = s.xo[a] + s.yo[b] + s.zo[e];
                if (sc.flt) atomicAdd(mesh + cell, v);
                else atomicAdd(acc + cell, (unsigned long long)__double2ll_rn((double)v * sc.S));
            }
}

// out[i] = read (out != NULL); pos_bar[i] = ob_i d read_i / d pos (pos_bar != NULL; ob == NULL: obscalar); val_out as out
template <int MODE>
__global__ __launch_bounds__(256) void read_kb_kernel(Geom g, const float *__restrict__ pos, int64_t n, const float *__restrict__ mesh,
                                                      int order, float kc, float ninv, float *__restrict__ out,
                                                      const float *__restrict__ ob, int64_t obstride, float obscalar,
                                                      float *__restrict__ pos_bar) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    KbStencil s;
    float f[3], wx[4], wy[4], wz[4], dx[4], dy[4], dz[4];
    kb_locate<MODE>(g, pi, load3(pos, pi.i), order, s, f);
    kb_axis(f[0], order, kc, ninv, wx, dx);
    kb_axis(f[1], order, kc, ninv, wy, dy);
    kb_axis(f[2], order, kc, ninv, wz, dz);
    float v = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    for (int a = 0; a < order; ++a)
        for (int b = 0; b < order; ++b) {
            float r0 = 0.f, r1 = 0.f;
            for (int e = 0; e < order; ++e) {
                const float m = mesh[s.xo[a] + s.yo[b] + s.zo[e]];
                r0 += m * wz[e];
                r1 += m * dz[e];
            }
            v += wx[a] * wy[b] * r0;
            gx += dx[a] * wy[b] * r0;
            gy += wx[a] * dy[b] * r0;
            gz += wx[a] * wy[b] * r1;
        }
    if (out) out[pi.i] = v;
    if (pos_bar) {
        const float o = ob ? ob[pi.i * obstride] : obscalar;
        store3(pos_bar, pi.i, P3{o * gx, o * gy, o * gz});
    }
}

// ------------------------------------------------------------------------------------------------
// read: gather NCOMP contiguous meshes
template <int MODE, int ORDER, int NCOMP>
__global__ __launch_bounds__(256) void read_kernel(Geom g, const float *__restrict__ pos, int64_t n,
                                                   const float *__restrict__ meshes, int64_t M,
                                                   float *__restrict__ out) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    P3 d = load3(pos, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    Stencil<ORDER> s(g, c);
    float v[NCOMP], gx, gy, gz;
#pragma unroll
    for (int k = 0; k < NCOMP; ++k) interp<ORDER, false>(meshes + k * M, s, f, v[k], gx, gy, gz);
#pragma unroll
    for (int k = 0; k < NCOMP; ++k) out[pi.i * NCOMP + k] = v[k];
}

// three components from one interleaved [cell][3] mesh
template <int MODE, int ORDER, bool NT = false, bool NTIN = false>
__global__ __launch_bounds__(256) void read3_il_kernel(Geom g, const float *__restrict__ pos, int64_t n,
                                                       const float *__restrict__ fm, float *__restrict__ out) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    P3 d = NTIN ? load3_nt(pos, pi.i) : load3(pos, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    Stencil<ORDER> s(g, c);
    float F[3], G[3][3];
    interp3<ORDER, false, true>(fm, 0, s, f, F, G);
    if (NT) store3_nt(out, pi.i, F[0], F[1], F[2]);
    else store3(out, pi.i, P3{F[0], F[1], F[2]});
}

// VJP of read w.r.t. pos (also the pos-VJP of paint with NCOMP = 1 and out_bar = weights).
// If val_out != nullptr also writes the read values (the weights-VJP of paint).
template <int MODE, int ORDER, int NCOMP>
__global__ __launch_bounds__(256) void read_vjp_pos_kernel(Geom g, const float *__restrict__ pos, int64_t n,
                                                           const float *__restrict__ meshes, int64_t M,
                                                           const float *__restrict__ ob, int64_t obstride, float obscalar,
                                                           float *__restrict__ pos_bar, float *__restrict__ val_out) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    P3 d = load3(pos, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    Stencil<ORDER> s(g, c);
    P3 acc = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCOMP; ++k) {
        float v, gx, gy, gz;
        interp<ORDER, true>(meshes + k * M, s, f, v, gx, gy, gz);
        const float o = ob ? ob[pi.i * obstride + k] : obscalar;
        acc.x += o * gx;
        acc.y += o * gy;
        acc.z += o * gz;
        if (val_out) val_out[pi.i * NCOMP + k] = v;
    }
    store3(pos_bar, pi.i, acc);
}

// ------------------------------------------------------------------------------------------------
// drift / kick / fused read+kick+drift  (nbody.py:933-944)
__global__ __launch_bounds__(256) void axpy_kernel(const float *__restrict__ x, const float *__restrict__ y, int64_t n,
                                                   float a, float b, float *__restrict__ out) {
    // out = a*x + b*y over n floats
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i];
}

// maximum over the wave, valid in lane 63 (call with every lane active)
__device__ __forceinline__ unsigned wave_umax(unsigned m) { return wave_umax_dpp(m); }

template <int MODE, int ORDER, bool IL>
__global__ __launch_bounds__(256) void kick_drift_kernel(Geom g, const float *__restrict__ pos_in,
                                                         const float *__restrict__ vel_in, int64_t n,
                                                         const float *__restrict__ meshes, int64_t M, float alpha,
                                                         float beta, float dt, float *__restrict__ pos_out,
                                                         float *__restrict__ vel_out, unsigned *__restrict__ dmax, int nt) {
    PIdx pi = particle_index<MODE>(g, n);
    unsigned mbits = 0u;
    if (pi.valid) {
        P3 d, v;
        if (nt & 1) load3_nt2(pos_in, vel_in, pi.i, d, v);     // streaming: both are read once here
        else {
            d = load3(pos_in, pi.i);
            v = load3(vel_in, pi.i);
        }
        int c[3];
        float f[3];
        locate<MODE, ORDER>(g, pi, d, c, f);
        Stencil<ORDER> s(g, c);
        float F[3], G[3][3];
        interp3<ORDER, false, IL>(meshes, M, s, f, F, G);
        P3 v1 = {alpha * v.x + beta * F[0], alpha * v.y + beta * F[1], alpha * v.z + beta * F[2]};
        P3 d1 = {d.x + v1.x * dt, d.y + v1.y * dt, d.z + v1.z * dt};
        if (nt & 2) {
            store3_nt(vel_out, pi.i, v1.x, v1.y, v1.z);
            store3_nt(pos_out, pi.i, d1.x, d1.y, d1.z);
        } else {
            store3(vel_out, pi.i, v1);
            store3(pos_out, pi.i, d1);
        }
        mbits = __float_as_uint(d1.x) & 0x7fffffffu;
    }
    if (dmax) {   // max |x displacement| of the new positions (slab plans: the ghost depth the next step needs); every lane
                  // of the wave is active again here; one L2-coherent read per wave, an atomic only while the slot is smaller
        const unsigned m = wave_umax(mbits);
        if ((threadIdx.x & 63) == 63) {
            unsigned *slot = dmax + (blockIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE;
            if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
static int fxg_prepare(mcpm_plan *p);
static inline int fxg_q(int64_t deposits);
static inline void flat_launch(int64_t n, dim3 &grid, dim3 &block) {
    block = dim3(256);
    grid = dim3((unsigned)((n + 255) / 256));
}

static int check_particles(mcpm_plan *p, const void *pos, int64_t n, int mode, int order, const char *who) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos != nullptr || n == 0, MCPM_E_ARG, std::string(who) + ": null particle array");   // empty input is valid
    MCPM_REQUIRE(p, n >= 0 && n < ((int64_t)1 << 31), MCPM_E_ARG, std::string(who) + ": bad particle count");
    MCPM_REQUIRE(p, mode == MCPM_POS_ABSOLUTE || mode == MCPM_POS_LATTICE, MCPM_E_ARG, std::string(who) + ": bad pos_mode");
    MCPM_REQUIRE(p, mode != MCPM_POS_LATTICE || n == p->Np, MCPM_E_SHAPE,
                 std::string(who) + ": MCPM_POS_LATTICE needs n == px*py*pz");
    MCPM_REQUIRE(p, order >= 1 && order <= 4, MCPM_E_ORDER, std::string(who) + ": assignment order must be 1 (NGP), 2 (CIC), 3 (TSC) or 4 (PCS)");
    return MCPM_OK;
}

#define DISPATCH_MODE_ORDER(mode, order, CALL)             \
    do {                                                   \
        if (mode == MCPM_POS_LATTICE) {                    \
            if (order == 2) {                              \
                CALL(MCPM_POS_LATTICE, 2);                 \
            } else if (order == 1) {                       \
                CALL(MCPM_POS_LATTICE, 1);                 \
            } else if (order == 3) {                       \
                CALL(MCPM_POS_LATTICE, 3);                 \
            } else {                                       \
                CALL(MCPM_POS_LATTICE, 4);                 \
            }                                              \
        } else {                                           \
            if (order == 2) {                              \
                CALL(MCPM_POS_ABSOLUTE, 2);                \
            } else if (order == 1) {                       \
                CALL(MCPM_POS_ABSOLUTE, 1);                \
            } else if (order == 3) {                       \
                CALL(MCPM_POS_ABSOLUTE, 3);                \
            } else {                                       \
                CALL(MCPM_POS_ABSOLUTE, 4);                \
            }                                              \
        }                                                  \
    } while (0)

// ---- self-test of the hand-written 12-byte streaming store ---------------------------------------------
// Record i receives the three floats (3i, 3i+1, 3i+2) mod 2^24 (exact in fp32).  MODE 0: plain compiler store (the twin);
// 1: the store3_nt sequence on pinned registers v[40:42] whose NEXT instructions overwrite all three data registers -- the
// situation the s_nop of MCPM_STORE_DATA_HAZARD_NOP exists for; 2: the same without the s_nop (informational: shows whether
// this part exhibits the hazard; never used by the product); 3: store3_nt itself followed by VALU writes of its source values.
template <int MODE>
__global__ void __launch_bounds__(256) store3_hazard_kernel(float *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = (float)(int)((3 * i) & 0xFFFFFF), b = (float)(int)((3 * i + 1) & 0xFFFFFF), c = (float)(int)((3 * i + 2) & 0xFFFFFF);
    float *q = out + 3 * i;
    if (MODE == 0) {
        q[0] = a, q[1] = b, q[2] = c;
    } else if (MODE == 1) {
        asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\t"
                     "global_store_dwordx3 %0, v[40:42], off nt\n\t" MCPM_STORE_DATA_HAZARD_NOP "\n\t"
                     "v_mov_b32 v40, -1.0\n\tv_mov_b32 v41, -1.0\n\tv_mov_b32 v42, -1.0"
                     : : "v"(q), "v"(a), "v"(b), "v"(c) : "v40", "v41", "v42", "memory");
    } else if (MODE == 2) {
        asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\t"
                     "global_store_dwordx3 %0, v[40:42], off nt\n\t"
                     "v_mov_b32 v40, -1.0\n\tv_mov_b32 v41, -1.0\n\tv_mov_b32 v42, -1.0"
                     : : "v"(q), "v"(a), "v"(b), "v"(c) : "v40", "v41", "v42", "memory");
    } else {
        store3_nt(out, i, a, b, c);
        asm volatile("v_mov_b32 %0, -1.0\n\tv_mov_b32 %1, -1.0\n\tv_mov_b32 %2, -1.0" : "+v"(a), "+v"(b), "+v"(c));
        if (a != -1.f || b != -1.f || c != -1.f) q[0] = a + b + c;      // keeps the overwrites alive; never taken
    }
}


extern "C" {

int mcpm_cell_index(mcpm_plan *p, const float *pos, int64_t n, int mode, int order, int16_t *idx) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_cell_index"));
    MCPM_REQUIRE(p, idx != nullptr, MCPM_E_ARG, "mcpm_cell_index: null output");
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
#define CALL(MO, OR) cell_index_kernel<MO, OR><<<grid, block, 0, p->stream>>>(p->g, pos, n, idx)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "cell_index_kernel");
    return MCPM_OK;
}

int mcpm_paint_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *weights, int64_t wstride,
                   float wscalar, int order, float *mesh, int accumulate) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_paint_f32"));
    MCPM_REQUIRE(p, mesh != nullptr, MCPM_E_ARG, "mcpm_paint_f32: null mesh");
    if (weights && wstride < 1) return mcpm_fail(p, MCPM_E_ARG, "mcpm_paint_f32: wstride must be >= 1");
    StageTimer st_(p, ST_PAINT, (weights ? 16.0 : 12.0) * n + (accumulate ? 8.0 : 4.0) * p->M);
    if (mode == MCPM_POS_LATTICE && order == 2 && n > 0 && mcpm_paint_tiled(p, pos, weights, wstride, wscalar, mesh, accumulate)) {
        MCPM_LAUNCH_CHECK(p, "paint_tile_kernel");
        return MCPM_OK;
    }
    if (!accumulate) MCPM_HIP(p, hipMemsetAsync(mesh, 0, sizeof(float) * p->M, p->stream));
    if (n == 0) return MCPM_OK;
    MCPM_TRY(fxg_prepare(p));
    if (weights) {
        MCPM_HIP(p, hipMemsetAsync(p->gx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream));
        absmax_strided_kernel<<<2048, 256, 0, p->stream>>>(weights, wstride, n, p->gx_wmax);
    } else {
        fxg_set_unit_kernel<<<1, MCPM_FX_SLOTS, 0, p->stream>>>(p->gx_wmax);
    }
    const int q = fxg_q(n * order * order * order);
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
#define CALL(MO, OR) paint_atomic_kernel<MO, OR><<<grid, block, 0, p->stream>>>(p->g, pos, n, weights, wstride, mesh, (unsigned long long *)p->gx_acc, p->gx_wmax, q, p->outlier_count + 2)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    paint_fxg_flush_kernel<<<2048, 256, 0, p->stream>>>(p->gx_acc, mesh, p->M, p->gx_wmax, q, weights ? 1.f : wscalar, (((uintptr_t)mesh) & 7) ? 0 : 1);
    MCPM_LAUNCH_CHECK(p, "paint_atomic_kernel");
    return MCPM_OK;
}

int mcpm_paint3_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *weights3, int order, float *meshes3,
                    int accumulate) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_paint3_f32"));
    MCPM_REQUIRE(p, weights3 && meshes3, MCPM_E_ARG, "mcpm_paint3_f32: null buffer");
    if (mode == MCPM_POS_LATTICE && order == 2 && n > 0) {
        StageTimer st_(p, ST_PAINT3, 24.0 * n + (accumulate ? 24.0 : 12.0) * p->M);
        if (mcpm_paint3_tiled(p, pos, weights3, meshes3, accumulate)) {
            MCPM_LAUNCH_CHECK(p, "paint3_tile_kernel");
            return MCPM_OK;
        }
    }
    for (int c = 0; c < 3; ++c) MCPM_TRY(mcpm_paint_f32(p, pos, n, mode, weights3 + c, 3, 0.f, order, meshes3 + c * p->M, accumulate));
    return MCPM_OK;
}

int mcpm_read_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *meshes, int ncomp, int order,
                  float *out) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_read_f32"));
    MCPM_REQUIRE(p, meshes && (out || n == 0), MCPM_E_ARG, "mcpm_read_f32: null buffer");
    MCPM_REQUIRE(p, ncomp == 1 || ncomp == 3, MCPM_E_ARG, "mcpm_read_f32: ncomp must be 1 or 3");
    StageTimer st_(p, ST_READ, 12.0 * n + 4.0 * ncomp * (p->M + n));
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
#define CALL(MO, OR)                                                                               \
    if (ncomp == 1) read_kernel<MO, OR, 1><<<grid, block, 0, p->stream>>>(p->g, pos, n, meshes, p->M, out); \
    else read_kernel<MO, OR, 3><<<grid, block, 0, p->stream>>>(p->g, pos, n, meshes, p->M, out)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "read_kernel");
    return MCPM_OK;
}

int mcpm_read_vjp_pos_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *meshes, int ncomp,
                          int order, const float *out_bar, float *pos_bar) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_read_vjp_pos_f32"));
    MCPM_REQUIRE(p, meshes && ((out_bar && pos_bar) || n == 0), MCPM_E_ARG, "mcpm_read_vjp_pos_f32: null buffer");
    MCPM_REQUIRE(p, ncomp == 1 || ncomp == 3, MCPM_E_ARG, "mcpm_read_vjp_pos_f32: ncomp must be 1 or 3");
    StageTimer st_(p, ST_READ, 24.0 * n + 4.0 * ncomp * (p->M + n));
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
#define CALL(MO, OR)                                                                                           \
    if (ncomp == 1)                                                                                            \
        read_vjp_pos_kernel<MO, OR, 1><<<grid, block, 0, p->stream>>>(p->g, pos, n, meshes, p->M, out_bar, 1, 0.f, pos_bar, nullptr); \
    else                                                                                                       \
        read_vjp_pos_kernel<MO, OR, 3><<<grid, block, 0, p->stream>>>(p->g, pos, n, meshes, p->M, out_bar, 3, 0.f, pos_bar, nullptr)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "read_vjp_pos_kernel");
    return MCPM_OK;
}

int mcpm_paint_vjp_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *weights, int64_t wstride,
                       float wscalar, int order, const float *mesh_bar, float *pos_bar, float *weights_bar) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_paint_vjp_f32"));
    MCPM_REQUIRE(p, mesh_bar && (pos_bar || n == 0), MCPM_E_ARG, "mcpm_paint_vjp_f32: null buffer");
    if (weights && wstride < 1) return mcpm_fail(p, MCPM_E_ARG, "mcpm_paint_vjp_f32: wstride must be >= 1");
    StageTimer st_(p, ST_READ, 28.0 * n + 4.0 * p->M);
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
#define CALL(MO, OR)                                                                                                   \
    read_vjp_pos_kernel<MO, OR, 1><<<grid, block, 0, p->stream>>>(p->g, pos, n, mesh_bar, p->M, weights, wstride, wscalar, \
                                                                  pos_bar, weights_bar)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "read_vjp_pos_kernel(paint_vjp)");
    return MCPM_OK;
}

static inline void kb_constants(int order, float kcut, float &kc, float &ninv) {
    kc = kcut * (float)order * 0.5f;
    ninv = (float)((double)kc / ((double)order * sinh((double)kc)));
}

int mcpm_paint_kb_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *weights, int64_t wstride, float wscalar,
                      int order, float kcut, float *mesh, int accumulate) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_paint_kb_f32"));
    MCPM_REQUIRE(p, mesh != nullptr && kcut > 0.f, MCPM_E_ARG, "mcpm_paint_kb_f32: null mesh or kcut <= 0");
    if (weights && wstride < 1) return mcpm_fail(p, MCPM_E_ARG, "mcpm_paint_kb_f32: wstride must be >= 1");
    StageTimer st_(p, ST_PAINT, (weights ? 16.0 : 12.0) * n + (accumulate ? 8.0 : 4.0) * p->M);
    if (!accumulate) MCPM_HIP(p, hipMemsetAsync(mesh, 0, sizeof(float) * p->M, p->stream));
    if (n == 0) return MCPM_OK;
    MCPM_TRY(fxg_prepare(p));
    if (weights) {
        MCPM_HIP(p, hipMemsetAsync(p->gx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream));
        absmax_strided_kernel<<<2048, 256, 0, p->stream>>>(weights, wstride, n, p->gx_wmax);
    } else {
        fxg_set_unit_kernel<<<1, MCPM_FX_SLOTS, 0, p->stream>>>(p->gx_wmax);
    }
    const int q = fxg_q(n * order * order * order) - 2;     // a Kaiser-Bessel weight can exceed 1 (1.6 at order 1): two bits of headroom
    float kc, ninv;
    kb_constants(order, kcut, kc, ninv);
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) {
        lattice_launch(p->g, grid, block);
        paint_kb_kernel<MCPM_POS_LATTICE><<<grid, block, 0, p->stream>>>(p->g, pos, n, weights, wstride, mesh, (unsigned long long *)p->gx_acc, p->gx_wmax, q, order, kc, ninv);
    } else {
        flat_launch(n, grid, block);
        paint_kb_kernel<MCPM_POS_ABSOLUTE><<<grid, block, 0, p->stream>>>(p->g, pos, n, weights, wstride, mesh, (unsigned long long *)p->gx_acc, p->gx_wmax, q, order, kc, ninv);
    }
    paint_fxg_flush_kernel<<<2048, 256, 0, p->stream>>>(p->gx_acc, mesh, p->M, p->gx_wmax, q, weights ? 1.f : wscalar, (((uintptr_t)mesh) & 7) ? 0 : 1);
    MCPM_LAUNCH_CHECK(p, "paint_kb_kernel");
    return MCPM_OK;
}

int mcpm_read_kb_f32(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *mesh, int order, float kcut, float *out,
                     const float *out_bar, int64_t obstride, float obscalar, float *pos_bar) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_read_kb_f32"));
    MCPM_REQUIRE(p, mesh && (out || pos_bar || n == 0) && kcut > 0.f, MCPM_E_ARG, "mcpm_read_kb_f32: null buffer or kcut <= 0");
    StageTimer st_(p, ST_READ, 24.0 * n + 4.0 * (p->M + n));
    if (n == 0) return MCPM_OK;
    float kc, ninv;
    kb_constants(order, kcut, kc, ninv);
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) {
        lattice_launch(p->g, grid, block);
        read_kb_kernel<MCPM_POS_LATTICE><<<grid, block, 0, p->stream>>>(p->g, pos, n, mesh, order, kc, ninv, out, out_bar, obstride, obscalar, pos_bar);
    } else {
        flat_launch(n, grid, block);
        read_kb_kernel<MCPM_POS_ABSOLUTE><<<grid, block, 0, p->stream>>>(p->g, pos, n, mesh, order, kc, ninv, out, out_bar, obstride, obscalar, pos_bar);
    }
    MCPM_LAUNCH_CHECK(p, "read_kb_kernel");
    return MCPM_OK;
}

int mcpm_drift_f32(mcpm_plan *p, const float *pos_in, const float *vel, int64_t n, float dt, float *pos_out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos_in && vel && pos_out && n >= 0, MCPM_E_ARG, "mcpm_drift_f32: bad argument");
    StageTimer st_(p, ST_AXPY, 36.0 * n);
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    flat_launch(3 * n, grid, block);
    axpy_kernel<<<grid, block, 0, p->stream>>>(pos_in, vel, 3 * n, 1.f, dt, pos_out);
    MCPM_LAUNCH_CHECK(p, "axpy_kernel(drift)");
    return MCPM_OK;
}

int mcpm_kick_f32(mcpm_plan *p, const float *vel_in, const float *forces, int64_t n, float alpha, float beta,
                  float *vel_out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, vel_in && forces && vel_out && n >= 0, MCPM_E_ARG, "mcpm_kick_f32: bad argument");
    StageTimer st_(p, ST_AXPY, 36.0 * n);
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    flat_launch(3 * n, grid, block);
    axpy_kernel<<<grid, block, 0, p->stream>>>(vel_in, forces, 3 * n, alpha, beta, vel_out);
    MCPM_LAUNCH_CHECK(p, "axpy_kernel(kick)");
    return MCPM_OK;
}

int mcpm_kick_drift_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, int64_t n, int mode,
                        const float *meshes3, int order, float alpha, float beta, float dt, float *pos_out,
                        float *vel_out) {
    return mcpm_kick_drift_layout(p, pos_in, vel_in, n, mode, meshes3, 0, order, alpha, beta, dt, pos_out, vel_out);
}

int mcpm_kick_drift_il_f32(mcpm_plan *p, const float *pos_in, const float *vel_in, int64_t n, int mode,
                           const float *mesh_il, int order, float alpha, float beta, float dt, float *pos_out,
                           float *vel_out) {
    return mcpm_kick_drift_layout(p, pos_in, vel_in, n, mode, mesh_il, 1, order, alpha, beta, dt, pos_out, vel_out);
}

int mcpm_plan_track_dmax(mcpm_plan *p, unsigned *slots) {
    if (!p) return MCPM_E_ARG;
    p->dmax = slots;
    return MCPM_OK;
}

int mcpm_selftest_store3_nt(void *stream, float *out, int64_t n, int mode) {
    if (!out || n < 0 || mode < 0 || mode > 3) return mcpm_fail(nullptr, MCPM_E_ARG, "mcpm_selftest_store3_nt: bad argument");
    if (n == 0) return MCPM_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) store3_hazard_kernel<0><<<grid, block, 0, st>>>(out, n);
    else if (mode == 1) store3_hazard_kernel<1><<<grid, block, 0, st>>>(out, n);
    else if (mode == 2) store3_hazard_kernel<2><<<grid, block, 0, st>>>(out, n);
    else store3_hazard_kernel<3><<<grid, block, 0, st>>>(out, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mcpm_fail(nullptr, MCPM_E_HIP, std::string("store3_hazard_kernel: ") + hipGetErrorString(e));
    return MCPM_OK;
}

}  // extern "C"

static int fxg_prepare(mcpm_plan *p) {   // the int64 accumulator mesh of the order-independent sums (all-zero between calls)
    if (!p->gx_acc) {
        MCPM_HIP(p, hipMalloc((void **)&p->gx_acc, sizeof(long long) * p->M));
        MCPM_HIP(p, hipMemsetAsync(p->gx_acc, 0, sizeof(long long) * p->M, p->stream));
    }
    return MCPM_OK;
}
static inline int fxg_q(int64_t deposits) {   // sum of |contribution| S < 2^62 whatever the collisions
    int nb = 1;
    while (nb < 62 && ((int64_t)1 << nb) < deposits) ++nb;
    return 61 - nb < 40 ? 61 - nb : 40;
}

int mcpm_lattice_scatter_fx(mcpm_plan *p, const float *xb, const float *vb, float a, float b, float *meshes3) {
    MCPM_TRY(fxg_prepare(p));
    MCPM_HIP(p, hipMemsetAsync(meshes3, 0, sizeof(float) * 3 * p->M, p->stream));
    MCPM_HIP(p, hipMemsetAsync(p->gx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream));
    absmax_axpby3_kernel<<<2048, 256, 0, p->stream>>>(xb, vb, 3 * p->Np, a, b, p->gx_wmax);
    const int q = fxg_q(p->Np);
    dim3 grid, block;
    lattice_launch(p->g, grid, block);
    for (int c = 0; c < 3; ++c) {
        float *out = meshes3 + c * p->M;
        lattice_scatter_fx_kernel<<<grid, block, 0, p->stream>>>(p->g, xb, vb, a, b, c, out, (unsigned long long *)p->gx_acc, p->gx_wmax, q);
        paint_fxg_flush_kernel<<<2048, 256, 0, p->stream>>>(p->gx_acc, out, p->M, p->gx_wmax, q, 1.f, (((uintptr_t)out) & 7) ? 0 : 1);
    }
    MCPM_LAUNCH_CHECK(p, "lattice_scatter_fx_kernel");
    return MCPM_OK;
}

// three-component read of an interleaved [cell][3] force mesh (internal: pm_forces)
int mcpm_read3_il(mcpm_plan *p, const float *pos, int64_t n, int mode, const float *fm_il, int order, float *out) {
    MCPM_TRY(check_particles(p, pos, n, mode, order, "mcpm_read3_il"));
    MCPM_REQUIRE(p, fm_il && out, MCPM_E_ARG, "mcpm_read3_il: null buffer");
    StageTimer st_(p, ST_READ, 24.0 * n + 12.0 * p->M);
    if (n == 0) return MCPM_OK;
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
    static const bool nt = [] { const char *e = getenv("MCPM_NT3"); const int v = e ? atoi(e) : 1; return v == 1 || v == 3; }();   // streaming output (A/B knob)
    static const bool ntin = [] { const char *e = getenv("MCPM_NT_POS"); return e ? atoi(e) != 0 : true; }();    // streaming position loads
    const bool big = n >= ((int64_t)1 << 23);      // streaming hints only when the arrays do not fit the caches anyway
#define CALL(MO, OR)                                                                                  \
    if (nt && ntin && big) read3_il_kernel<MO, OR, true, true><<<grid, block, 0, p->stream>>>(p->g, pos, n, fm_il, out);  \
    else if (nt && big) read3_il_kernel<MO, OR, true><<<grid, block, 0, p->stream>>>(p->g, pos, n, fm_il, out);  \
    else read3_il_kernel<MO, OR><<<grid, block, 0, p->stream>>>(p->g, pos, n, fm_il, out)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "read3_il_kernel");
    return MCPM_OK;
}

// layout 0: three meshes M apart; 1: interleaved [cell][3] (internal: the fused Poisson solve's output for the steppers)
int mcpm_kick_drift_layout(mcpm_plan *p, const float *pos_in, const float *vel_in, int64_t n, int mode, const float *meshes3,
                           int layout, int order, float alpha, float beta, float dt, float *pos_out, float *vel_out) {
    MCPM_TRY(check_particles(p, pos_in, n, mode, order, "mcpm_kick_drift_f32"));
    MCPM_REQUIRE(p, vel_in && meshes3 && pos_out && vel_out, MCPM_E_ARG, "mcpm_kick_drift_f32: null buffer");
    StageTimer st_(p, ST_KICKDRIFT, 48.0 * n + 12.0 * p->M);
    if (n == 0) return MCPM_OK;
    if (p->dmax) MCPM_HIP(p, hipMemsetAsync(p->dmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream));
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
    static const int ntp_env = [] { const char *e = getenv("MCPM_NT_PART"); return e ? atoi(e) : 3; }();   // streaming loads (1) and stores (2): step 11.95 -> 11.6 ms at 512^3
    // in-place updates keep ordinary accesses; so do small problems, whose arrays live in the caches (128^3: hints cost 1-10 %)
    const int ntp = (pos_in == pos_out || vel_in == vel_out || n < ((int64_t)1 << 23)) ? 0 : ntp_env;
#define CALL(MO, OR)                                                                                                             \
    if (layout) kick_drift_kernel<MO, OR, true><<<grid, block, 0, p->stream>>>(p->g, pos_in, vel_in, n, meshes3, p->M, alpha, beta, dt, pos_out, vel_out, p->dmax, ntp); \
    else kick_drift_kernel<MO, OR, false><<<grid, block, 0, p->stream>>>(p->g, pos_in, vel_in, n, meshes3, p->M, alpha, beta, dt, pos_out, vel_out, p->dmax, ntp)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "kick_drift_kernel");
    return MCPM_OK;
}

