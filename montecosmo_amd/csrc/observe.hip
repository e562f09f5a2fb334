// Evolved particles -> redshift-space positions on the paint mesh (montecosmo/model.py:780-797 without Alcock-Paczynski):
//   los, a   = los_scalefactor_pos(pos)                      bricks.py:750-768
//   pos_phys = cell2phys_pos(pos)                            bricks.py:628-636
//   pos_phys += rsd(vel, los, a, dvel)                       bricks.py:791-803
//   pos_out  = phys2cell_pos(pos_phys, paint_shape)          bricks.py:638-646
// fused into one pass with its VJP.  cell2phys then phys2cell with the same box cancel exactly, so the kernel evaluates
//   out = x * (cell_e / cell_p) + R^T [ (V . l) l ] / cell_p,   V = R (vel * cell_e) g(a) f(a) + dvel,
// P = R (x * cell_e - box/2) + centre, l = P/|P| (curved sky) or centre/|centre| (flat), a = chi2a(|P|) or |P . l|,
// which keeps the displacement-from-lattice encoding of the positions (no box-sized float32 round trip).
// Light cone: a and g(a) f(a) come from the same two linear-interpolation tables as the host (chi -> a, a -> g, f).
#include "mcpm_internal.h"
#include "reduce_dev.h"

namespace {

struct Obs {
    float R[9];                 // box_rot matrix, row major: apply(x) = R x
    float ce[3], cp[3];         // cell lengths (Mpc/h) of the evolution and paint meshes
    float hb[3], ctr[3];        // box_size / 2, box_center
    float lf[3];                // flat-sky line of sight
    int curved, lightcone;
    float gf;                   // g(a_obs) f(a_obs) when not on the light cone
    int nchi, ngrow;
};

struct Tables {                 // device, float64; chi ascending
    const double *chi, *a_of_chi, *a, *g, *f;
};

// np.interp (clamped) and its slope
__device__ __forceinline__ double interp1(double x, const double *xp, const double *fp, int n, double &slope) {
    if (x <= xp[0]) { slope = 0.; return fp[0]; }
    if (x >= xp[n - 1]) { slope = 0.; return fp[n - 1]; }
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    slope = (fp[hi] - fp[lo]) / (xp[hi] - xp[lo]);
    return fp[lo] + slope * (x - xp[lo]);
}

__device__ __forceinline__ void rot(const float (&R)[9], const float (&v)[3], float (&o)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
__device__ __forceinline__ void rot_t(const float (&R)[9], const float (&v)[3], float (&o)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = R[j] * v[0] + R[3 + j] * v[1] + R[6 + j] * v[2];
}

struct Fwd {
    float x[3], P[3], l[3], r, sgn, gf, dgf_dr, Vr[3], V[3], s;
};

// common forward evaluation of one particle; x = absolute cell coordinates on the evolution mesh
__device__ __forceinline__ void forward(const Obs &og, const Tables &tb, const float (&x)[3], const float (&vel)[3],
                                        const float (&dv)[3], Fwd &w) {
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a] = x[a] * og.ce[a] - og.hb[a];
    rot(og.R, t, w.P);
#pragma unroll
    for (int a = 0; a < 3; ++a) w.P[a] += og.ctr[a];
    w.sgn = 1.f;
    if (og.curved) {
        w.r = sqrtf(w.P[0] * w.P[0] + w.P[1] * w.P[1] + w.P[2] * w.P[2]);
        const float ir = w.r == 0.f ? 0.f : 1.f / w.r;   // safe_div
#pragma unroll
        for (int a = 0; a < 3; ++a) w.l[a] = w.P[a] * ir;
    } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) w.l[a] = og.lf[a];
        const float d = w.P[0] * w.l[0] + w.P[1] * w.l[1] + w.P[2] * w.l[2];
        w.sgn = d < 0.f ? -1.f : 1.f;
        w.r = fabsf(d);
    }
    w.gf = og.gf;
    w.dgf_dr = 0.f;
    if (og.lightcone) {
        double da_dr, dg_da, df_da;
        const double a = interp1((double)w.r, tb.chi, tb.a_of_chi, og.nchi, da_dr);
        const double g = interp1(a, tb.a, tb.g, og.ngrow, dg_da), f = interp1(a, tb.a, tb.f, og.ngrow, df_da);
        w.gf = (float)(g * f);
        w.dgf_dr = (float)((dg_da * f + g * df_da) * da_dr);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) t[a] = vel[a] * og.ce[a];
    rot(og.R, t, w.Vr);
#pragma unroll
    for (int a = 0; a < 3; ++a) w.V[a] = w.Vr[a] * w.gf + dv[a];
    w.s = w.V[0] * w.l[0] + w.V[1] * w.l[1] + w.V[2] * w.l[2];
}

__device__ __forceinline__ void lattice_point(const Geom &g, int64_t i, float (&q)[3]) {
    const int ipz = (int)(i % g.pz);
    const int64_t t = i / g.pz;
    const int ipy = (int)(t % g.py), ipx = (int)(t / g.py);
    q[0] = (float)((double)ipx * g.nx / g.px);
    q[1] = (float)((double)ipy * g.ny / g.py);
    q[2] = (float)((double)ipz * g.nz / g.pz);
}

template <int MODE>
__global__ __launch_bounds__(256) void observe_kernel(Geom g, Obs og, Tables tb, const float *__restrict__ pos,
                                                      const float *__restrict__ vel, const float *__restrict__ dvel, int64_t n,
                                                      float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float q[3] = {0.f, 0.f, 0.f}, d[3], x[3], v[3], dv[3] = {0.f, 0.f, 0.f};
    if (MODE == MCPM_POS_LATTICE) lattice_point(g, i, q);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d[a] = pos[3 * i + a];
        x[a] = q[a] + d[a];
        v[a] = vel[3 * i + a];
        if (dvel) dv[a] = dvel[3 * i + a];
    }
    Fwd w;
    forward(og, tb, x, v, dv, w);
    float D[3] = {w.s * w.l[0], w.s * w.l[1], w.s * w.l[2]}, Dr[3];
    rot_t(og.R, D, Dr);
#pragma unroll
    for (int a = 0; a < 3; ++a) out[3 * i + a] = d[a] * (og.ce[a] / og.cp[a]) + Dr[a] / og.cp[a];  // lattice mode: displacement from q * ce/cp
}

template <int MODE>
__global__ __launch_bounds__(256) void observe_vjp_kernel(Geom g, Obs og, Tables tb, const float *__restrict__ pos,
                                                          const float *__restrict__ vel, const float *__restrict__ dvel, int64_t n,
                                                          const float *__restrict__ ob, float *__restrict__ pos_bar,
                                                          float *__restrict__ vel_bar, float *__restrict__ dvel_bar, double *part) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double red[1] = {0.};
    if (i < n) {
        float q[3] = {0.f, 0.f, 0.f}, x[3], v[3], dv[3] = {0.f, 0.f, 0.f}, o[3];
        if (MODE == MCPM_POS_LATTICE) lattice_point(g, i, q);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x[a] = q[a] + pos[3 * i + a];
            v[a] = vel[3 * i + a];
            if (dvel) dv[a] = dvel[3 * i + a];
            o[a] = ob[3 * i + a];
        }
        Fwd w;
        forward(og, tb, x, v, dv, w);
        float t[3], Db[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a] = o[a] / og.cp[a];
        rot(og.R, t, Db);                                                   // D_bar = R (out_bar / cell_p)
        const float sb = Db[0] * w.l[0] + Db[1] * w.l[1] + Db[2] * w.l[2];  // s_bar
        float lb[3], Vb[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lb[a] = w.s * Db[a] + sb * w.V[a];
            Vb[a] = sb * w.l[a];
        }
        const float gfb = Vb[0] * w.Vr[0] + Vb[1] * w.Vr[1] + Vb[2] * w.Vr[2];
        float vt[3];
        rot_t(og.R, Vb, vt);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            vel_bar[3 * i + a] = vt[a] * og.ce[a] * w.gf;
            if (dvel_bar) dvel_bar[3 * i + a] = Vb[a];
        }
        float Pb[3];
        const float rb = gfb * w.dgf_dr;
        if (og.curved) {
            const float ir = w.r == 0.f ? 0.f : 1.f / w.r;
            const float ll = lb[0] * w.l[0] + lb[1] * w.l[1] + lb[2] * w.l[2];
#pragma unroll
            for (int a = 0; a < 3; ++a) Pb[a] = (lb[a] - ll * w.l[a]) * ir + rb * w.l[a];
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) Pb[a] = rb * w.sgn * w.l[a];
        }
        float xt[3];
        rot_t(og.R, Pb, xt);
#pragma unroll
        for (int a = 0; a < 3; ++a) pos_bar[3 * i + a] = o[a] * (og.ce[a] / og.cp[a]) + xt[a] * og.ce[a];
        red[0] = og.lightcone ? 0. : (double)gfb;
    }
    block_partial<1>(red, part, gridDim.x, blockIdx.x);
}


// ---- cotangents of the look-up TABLES (light cone: how the cosmology enters; model.py:740, :781, bricks.py:750-768) -----------
// y = np.interp(x, xp, fp) = fp[lo] + (fp[lo+1] - fp[lo]) t,  t = (x - xp[lo]) / (xp[lo+1] - xp[lo]):
//   dy/dfp[lo] = 1 - t, dy/dfp[lo+1] = t;   dy/dxp[lo] = -slope (1 - t), dy/dxp[lo+1] = -slope t;   dy/dx = slope
// (clamped ends: y = fp[0] or fp[n-1], slope 0).  The kernels below contract those with per-particle cotangents into small
// table cotangents (integer accumulators, see ORDER-INDEPENDENT SUMS below); the host then contracts them with the tables'
// finite-difference Jacobian w.r.t. the cosmological parameters (model.py cosmo_vjp).
struct Interp {
    int lo;
    bool clamped;
    double t, slope, y;
};
__device__ __forceinline__ Interp interp_idx(double x, const double *xp, const double *fp, int n) {
    Interp r;
    r.clamped = true;
    if (x <= xp[0]) { r.lo = 0, r.t = 0., r.slope = 0., r.y = fp[0]; return r; }
    if (x >= xp[n - 1]) { r.lo = n - 2, r.t = 1., r.slope = 0., r.y = fp[n - 1]; return r; }
    r.clamped = false;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xp[mid] <= x) lo = mid; else hi = mid;
    }
    const double dx = xp[hi] - xp[lo];
    r.lo = lo;
    r.t = (x - xp[lo]) / dx;
    r.slope = (fp[hi] - fp[lo]) / dx;
    r.y = fp[lo] + r.slope * (x - xp[lo]);
    return r;
}
// same bracket, another value table on the same nodes
__device__ __forceinline__ void interp_at(const Interp &b, const double *xp, const double *fp, double &y, double &slope) {
    const double dx = xp[b.lo + 1] - xp[b.lo];
    const double d = fp[b.lo + 1] - fp[b.lo];
    y = fp[b.lo] + d * b.t;
    slope = b.clamped ? 0. : d / dx;
}
// ORDER-INDEPENDENT SUMS.  The table cotangents end up in d logp / d Omega_m, and every gradient of this build is bitwise the same
// call after call; float64 atomics are not (the first version of these kernels changed the last bits of Omega_m's gradient between
// two calls).  So the contributions are summed as INTEGERS: pass 0 takes the maximum |contribution| of every table (order-independent
// by nature), pass 1 rounds each contribution to 2^(e - 30) units (2^e >= that maximum, so a contribution is below 2^30 and a table
// entry holds 2^32 of them) and adds it with 64-bit integer LDS / global atomics, and a last kernel scales the integers back.
// Resolution: 10^-9 of the largest contribution per term.  A non-finite maximum makes the whole table NaN.
#define LC_KINDS 5
struct Acc {      // PASS 0: per-thread maxima;  PASS 1: integer accumulators in LDS
    double mx[LC_KINDS];
    unsigned long long *sh;
    double scale[LC_KINDS];
};
template <int PASS>
__device__ __forceinline__ void acc_add(Acc &A, int kind, int off, int idx, double v) {
    if (PASS == 0) A.mx[kind] = fmax(A.mx[kind], fabs(v));      // (fmax drops a NaN operand: the callers' `bad` flag catches it)
    else if (v != 0.) atomicAdd(A.sh + off + idx, (unsigned long long)__double2ll_rn(v * A.scale[kind]));
}
template <int PASS>
__device__ __forceinline__ void scatter_fp(Acc &A, int kind, int off, const Interp &b, double ybar) {
    acc_add<PASS>(A, kind, off, b.lo, ybar * (1. - b.t));
    acc_add<PASS>(A, kind, off, b.lo + 1, ybar * b.t);
}
// ... and, for a look-up whose NODES move with the cosmology (chi -> a), into the node table's accumulator
template <int PASS>
__device__ __forceinline__ void scatter_xp(Acc &A, int kind, int off, const Interp &b, double ybar) {
    acc_add<PASS>(A, kind, off, b.lo, -ybar * b.slope * (1. - b.t));
    acc_add<PASS>(A, kind, off, b.lo + 1, -ybar * b.slope * b.t);
}
// mxbits: float bits (rounded up) of the maxima, one per kind; a NaN / inf contribution sets 0x7f800000 or above
__device__ __forceinline__ void acc_begin(Acc &A, unsigned long long *sh, int ntot, const unsigned *mxbits, int pass) {
    A.sh = sh;
    for (int k = 0; k < LC_KINDS; ++k) {
        A.mx[k] = 0.;
        const int be = pass ? (int)(mxbits[k] >> 23) : 0;                 // biased exponent of the maximum: max < 2^(be - 126)
        A.scale[k] = (be == 0 || be >= 255) ? 0. : __longlong_as_double((long long)(1023 + 30 - (be - 126)) << 52);      // 2^(30 - e)
    }
    if (pass) {
        for (int i = threadIdx.x; i < ntot; i += blockDim.x) sh[i] = 0ull;
        __syncthreads();
    }
}
template <int PASS>
__device__ __forceinline__ void acc_end(Acc &A, int ntot, unsigned *mxbits, unsigned long long *out, bool bad) {
    if (PASS == 0) {
        for (int k = 0; k < LC_KINDS; ++k) {
            unsigned b = bad ? 0x7fc00000u : __float_as_uint(__double2float_ru(A.mx[k]));
            if (!(A.mx[k] < 3.0e38)) b = 0x7fc00000u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
            if ((threadIdx.x & 63) == 0 && b) atomicMax(mxbits + k, b);
        }
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < ntot; i += blockDim.x)
            if (A.sh[i] != 0ull) atomicAdd(out + i, A.sh[i]);
    }
}
// out[i] = integer sum scaled back; kind_end[k]: one past the last entry of kind k
__global__ void lc_scale_kernel(const unsigned long long *__restrict__ acc, const unsigned *__restrict__ mxbits, int ntot, int e0, int e1,
                                int e2, int e3, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntot) return;
    const int k = i < e0 ? 0 : (i < e1 ? 1 : (i < e2 ? 2 : (i < e3 ? 3 : 4)));
    const int be = (int)(mxbits[k] >> 23);
    if (be >= 255) { out[i] = __longlong_as_double(0x7ff8000000000000ll); return; }
    out[i] = be == 0 ? 0. : (double)(long long)acc[i] * __longlong_as_double((long long)(1023 - 30 + (be - 126)) << 52);
}

// Lagrangian side (model.py:740-764): a_q = chi2a(r0_q) kept in float32, then a2g(a_q) (bias weights and lpt), a2g2(a_q),
// a2dg2dg(a_q) = safe_div(g2 f2, g f) (lpt).  Cotangents per particle: gB (of a2g), g2B (of a2g2 = -3/7 g2raw), dB (of a2dg2dg).
// tables: chi[nchi] ascending, a(chi)[nchi], a[ng], g[ng], g2raw[ng], f[ng], f2[ng];  accumulators: chi_bar[nchi], g_bar, g2raw_bar, f_bar, f2_bar.
template <int PASS>
__global__ __launch_bounds__(256) void lightcone_tables_vjp_kernel(const float *__restrict__ r0, int64_t n, const double *__restrict__ tb,
                                                                   int nchi, int ng, const float *__restrict__ gB,
                                                                   const float *__restrict__ g2B, const float *__restrict__ dB,
                                                                   unsigned *__restrict__ mxbits, unsigned long long *__restrict__ out) {
    extern __shared__ unsigned long long shl[];
    const int ntot = nchi + 4 * ng;
    Acc A;
    acc_begin(A, shl, ntot, mxbits, PASS);
    const double *chi = tb, *aoc = tb + nchi, *ag = tb + 2 * nchi, *tg = ag + ng, *tg2 = tg + ng, *tf = tg2 + ng, *tf2 = tf + ng;
    const int o_chi = 0, o_g = nchi, o_g2 = nchi + ng, o_f = nchi + 2 * ng, o_f2 = nchi + 3 * ng;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const Interp ba = interp_idx((double)r0[i], chi, aoc, nchi);
        const double a = (double)(float)ba.y;                 // the forward pass hands a on as float32 (mcpm_interp_f32)
        const Interp bg = interp_idx(a, ag, tg, ng);
        double g = bg.y, sg = bg.slope, g2r, sg2, f, sf, f2, sf2;
        interp_at(bg, ag, tg2, g2r, sg2);
        interp_at(bg, ag, tf, f, sf);
        interp_at(bg, ag, tf2, f2, sf2);
        double gb = (double)gB[i], g2rb = 0., fb = 0., f2b = 0.;
        if (g2B) g2rb += (-3. / 7.) * (double)g2B[i];
        if (dB) {
            const double den = g * f, db = (double)dB[i];
            if (den != 0.) {                                   // a2dg2dg = (-3/7 g2r f2) / (g f), safe_div
                const double q = (-3. / 7.) * g2r * f2 / den;
                gb += -db * q / g;
                fb += -db * q / f;
                g2rb += db * (-3. / 7.) * f2 / den;
                f2b += db * (-3. / 7.) * g2r / den;
            }
        }
        const double ab = gb * sg + g2rb * sg2 + fb * sf + f2b * sf2;
        bad = bad || !(gb == gb && g2rb == g2rb && fb == fb && f2b == f2b && ab == ab);
        scatter_fp<PASS>(A, 1, o_g, bg, gb);
        scatter_fp<PASS>(A, 2, o_g2, bg, g2rb);
        scatter_fp<PASS>(A, 3, o_f, bg, fb);
        scatter_fp<PASS>(A, 4, o_f2, bg, f2b);
        scatter_xp<PASS>(A, 0, o_chi, ba, ab);
    }
    acc_end<PASS>(A, ntot, mxbits, out, bad);
}

// Eulerian side (model.py:781-784): gf_p = a2g(a_p) a2f(a_p), a_p = chi2a(r_p) at the evolved particle's distance.
// tables as in observe_kernel (chi, a(chi), a, g, f);  accumulators: chi_bar[nchi], g_bar[ngrow], f_bar[ngrow] (kinds 0, 1, 2).
template <int MODE, int PASS>
__global__ __launch_bounds__(256) void observe_tables_vjp_kernel(Geom g, Obs og, Tables tb, const float *__restrict__ pos,
                                                                 const float *__restrict__ vel, const float *__restrict__ dvel, int64_t n,
                                                                 const float *__restrict__ ob, unsigned *__restrict__ mxbits,
                                                                 unsigned long long *__restrict__ out) {
    extern __shared__ unsigned long long shl[];
    const int ntot = og.nchi + 2 * og.ngrow;
    Acc A;
    acc_begin(A, shl, ntot, mxbits, PASS);
    const int o_chi = 0, o_g = og.nchi, o_f = og.nchi + og.ngrow;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float q[3] = {0.f, 0.f, 0.f}, x[3], v[3], dv[3] = {0.f, 0.f, 0.f}, o[3];
        if (MODE == MCPM_POS_LATTICE) lattice_point(g, i, q);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            x[a] = q[a] + pos[3 * i + a];
            v[a] = vel[3 * i + a];
            if (dvel) dv[a] = dvel[3 * i + a];
            o[a] = ob[3 * i + a];
        }
        Fwd w;
        forward(og, tb, x, v, dv, w);
        float t[3], Db[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a] = o[a] / og.cp[a];
        rot(og.R, t, Db);
        const float sb = Db[0] * w.l[0] + Db[1] * w.l[1] + Db[2] * w.l[2];
        const double gfb = (double)sb * ((double)w.l[0] * w.Vr[0] + (double)w.l[1] * w.Vr[1] + (double)w.l[2] * w.Vr[2]);
        const Interp ba = interp_idx((double)w.r, tb.chi, tb.a_of_chi, og.nchi);
        const Interp bg = interp_idx(ba.y, tb.a, tb.g, og.ngrow);
        double f, sf;
        interp_at(bg, tb.a, tb.f, f, sf);
        const double gb = gfb * f, fb = gfb * bg.y, ab = gb * bg.slope + fb * sf;
        bad = bad || !(gb == gb && fb == fb && ab == ab);
        scatter_fp<PASS>(A, 1, o_g, bg, gb);
        scatter_fp<PASS>(A, 2, o_f, bg, fb);
        scatter_xp<PASS>(A, 0, o_chi, ba, ab);
    }
    acc_end<PASS>(A, ntot, mxbits, out, bad);
}

Obs make_obs(const mcpm_plan *p, const float *geom, int flags, int nchi, int ngrow) {
    Obs og;
    for (int i = 0; i < 9; ++i) og.R[i] = geom[i];
    const int ms[3] = {p->g.nx, p->g.ny, p->g.nz};
    float cn = 0.f;
    for (int a = 0; a < 3; ++a) {
        og.ce[a] = geom[9 + a] / (float)ms[a];
        og.cp[a] = geom[9 + a] / geom[15 + a];
        og.hb[a] = 0.5f * geom[9 + a];
        og.ctr[a] = geom[12 + a];
        cn += geom[12 + a] * geom[12 + a];
    }
    cn = sqrtf(cn);
    for (int a = 0; a < 3; ++a) og.lf[a] = cn == 0.f ? 0.f : geom[12 + a] / cn;
    og.curved = flags & 1;
    og.lightcone = (flags >> 1) & 1;
    og.gf = geom[18];
    og.nchi = nchi;
    og.ngrow = ngrow;
    return og;
}

}  // namespace

extern "C" {

// geom (host, 19 floats) = R[9] row major, box_size[3], box_center[3], paint_shape[3] (as floats), g(a_obs) f(a_obs).
// flags: bit 0 = curved sky, bit 1 = light cone (then the four tables, float64 on the DEVICE: chi ascending [nchi],
// a(chi) [nchi], a [ngrow], g [ngrow], f [ngrow] -- the host's growth / distance tables).  pos / out follow pos_mode:
// MCPM_POS_LATTICE: displacements from the plan's particle lattice on the evolution mesh in, displacements from the same
// lattice scaled to the paint mesh out; MCPM_POS_ABSOLUTE: absolute cell coordinates in and out.
int mcpm_observe_pos_f32(mcpm_plan *p, const float *pos, const float *vel, const float *dvel, int64_t n, int mode,
                         const float *geom, int flags, const double *tables, int nchi, int ngrow, float *out) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos && vel && geom && out && n > 0, MCPM_E_ARG, "mcpm_observe_pos_f32: bad argument");
    MCPM_REQUIRE(p, mode == MCPM_POS_ABSOLUTE || (mode == MCPM_POS_LATTICE && n == p->Np), MCPM_E_ARG, "mcpm_observe_pos_f32: bad pos_mode / count");
    MCPM_REQUIRE(p, !(flags & 2) || (tables && nchi >= 2 && ngrow >= 2), MCPM_E_ARG, "mcpm_observe_pos_f32: light cone needs the tables");
    Obs og = make_obs(p, geom, flags, nchi, ngrow);
    Tables tb{tables, tables ? tables + nchi : nullptr, tables ? tables + 2 * nchi : nullptr,
              tables ? tables + 2 * nchi + ngrow : nullptr, tables ? tables + 2 * nchi + 2 * ngrow : nullptr};
    const unsigned nb = (unsigned)((n + 255) / 256);
    StageTimer st_(p, ST_LPT, (dvel ? 48.0 : 36.0) * n);
    if (mode == MCPM_POS_LATTICE) observe_kernel<MCPM_POS_LATTICE><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out);
    else observe_kernel<MCPM_POS_ABSOLUTE><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out);
    MCPM_LAUNCH_CHECK(p, "observe_kernel");
    return MCPM_OK;
}

// VJP: out_bar (n,3) -> pos_bar, vel_bar, dvel_bar (NULL if dvel was NULL) and gf_bar (device double; the cotangent of the
// scalar g(a_obs) f(a_obs); 0 on the light cone, where the growth dependence on the particle distance is already in pos_bar
// and the dependence of the tables on the cosmology is not propagated).
int mcpm_observe_pos_vjp_f32(mcpm_plan *p, const float *pos, const float *vel, const float *dvel, int64_t n, int mode,
                             const float *geom, int flags, const double *tables, int nchi, int ngrow, const float *out_bar,
                             float *pos_bar, float *vel_bar, float *dvel_bar, double *gf_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos && vel && geom && out_bar && pos_bar && vel_bar && gf_bar && n > 0, MCPM_E_ARG, "mcpm_observe_pos_vjp_f32: bad argument");
    MCPM_REQUIRE(p, mode == MCPM_POS_ABSOLUTE || (mode == MCPM_POS_LATTICE && n == p->Np), MCPM_E_ARG, "mcpm_observe_pos_vjp_f32: bad pos_mode / count");
    MCPM_REQUIRE(p, !(flags & 2) || (tables && nchi >= 2 && ngrow >= 2), MCPM_E_ARG, "mcpm_observe_pos_vjp_f32: light cone needs the tables");
    MCPM_REQUIRE(p, (dvel == nullptr) == (dvel_bar == nullptr), MCPM_E_ARG, "mcpm_observe_pos_vjp_f32: dvel and dvel_bar go together");
    Obs og = make_obs(p, geom, flags, nchi, ngrow);
    Tables tb{tables, tables ? tables + nchi : nullptr, tables ? tables + 2 * nchi : nullptr,
              tables ? tables + 2 * nchi + ngrow : nullptr, tables ? tables + 2 * nchi + 2 * ngrow : nullptr};
    double *P, *Q;
    unsigned *ticket, R;
    const unsigned nb = (unsigned)((n + 255) / 256);
    StageTimer st_(p, ST_LPT, (dvel ? 84.0 : 60.0) * n);
    MCPM_TRY(mcpm_det_scratch(p, 1, nb, &P, &Q, &ticket, &R));
    if (mode == MCPM_POS_LATTICE)
        observe_vjp_kernel<MCPM_POS_LATTICE><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out_bar, pos_bar, vel_bar, dvel_bar, P);
    else
        observe_vjp_kernel<MCPM_POS_ABSOLUTE><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out_bar, pos_bar, vel_bar, dvel_bar, P);
    det_fold_kernel<<<R, 256, 0, p->stream>>>(P, nb, 1, Q, ticket, 1.0, det_outs(gf_bar));
    MCPM_LAUNCH_CHECK(p, "observe_vjp_kernel");
    return MCPM_OK;
}

// Light cone, Lagrangian side: table cotangents of the look-ups a_q = chi2a(r0_q), a2g / a2g2 / a2dg2dg (a_q) (see
// lightcone_tables_vjp_kernel).  tables (device float64): chi[nchi] ascending, a(chi)[nchi], a[ngrow], g, g2 (raw table, without
// the -3/7), f, f2 [ngrow each]; g_bar (n) is required, g2_bar / dg2dg_bar may be NULL.  table_bar (device float64, OVERWRITTEN):
// chi_bar[nchi], g_bar[ngrow], g2_bar[ngrow], f_bar[ngrow], f2_bar[ngrow].
int mcpm_lightcone_tables_vjp_f32(mcpm_plan *p, const float *r0, int64_t n, const double *tables, int nchi, int ngrow,
                                  const float *g_bar, const float *g2_bar, const float *dg2dg_bar, double *table_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, r0 && tables && g_bar && table_bar && n > 0 && nchi >= 2 && ngrow >= 2, MCPM_E_ARG, "mcpm_lightcone_tables_vjp_f32: bad argument");
    const size_t ntot = (size_t)nchi + 4 * (size_t)ngrow;
    MCPM_REQUIRE(p, ntot + 8 <= 3072 && ntot * sizeof(double) <= 60 * 1024, MCPM_E_ARG, "mcpm_lightcone_tables_vjp_f32: tables exceed the accumulators");
    // integer accumulators and the maxima: the plan's reduction scratch (free between the model-side calls)
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(p->reduce);
    unsigned *mx = reinterpret_cast<unsigned *>(acc + ntot);
    MCPM_HIP(p, hipMemsetAsync(acc, 0, (ntot + 4) * sizeof(double), p->stream));
    const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    lightcone_tables_vjp_kernel<0><<<nb, 256, 0, p->stream>>>(r0, n, tables, nchi, ngrow, g_bar, g2_bar, dg2dg_bar, mx, acc);
    lightcone_tables_vjp_kernel<1><<<nb, 256, ntot * sizeof(double), p->stream>>>(r0, n, tables, nchi, ngrow, g_bar, g2_bar, dg2dg_bar, mx, acc);
    lc_scale_kernel<<<(unsigned)((ntot + 255) / 256), 256, 0, p->stream>>>(acc, mx, (int)ntot, nchi, nchi + ngrow, nchi + 2 * ngrow, nchi + 3 * ngrow, table_bar);
    MCPM_LAUNCH_CHECK(p, "lightcone_tables_vjp_kernel");
    return MCPM_OK;
}

// Light cone, observation side: table cotangents of gf_p = a2g(a_p) a2f(a_p), a_p = chi2a(|P_p|) inside mcpm_observe_pos_f32 (same
// arguments; flags must have the light-cone bit).  table_bar (device float64, OVERWRITTEN): chi_bar[nchi], g_bar[ngrow], f_bar[ngrow].
int mcpm_observe_pos_tables_vjp_f32(mcpm_plan *p, const float *pos, const float *vel, const float *dvel, int64_t n, int mode,
                                    const float *geom, int flags, const double *tables, int nchi, int ngrow, const float *out_bar,
                                    double *table_bar) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, pos && vel && geom && out_bar && table_bar && n > 0, MCPM_E_ARG, "mcpm_observe_pos_tables_vjp_f32: bad argument");
    MCPM_REQUIRE(p, mode == MCPM_POS_ABSOLUTE || (mode == MCPM_POS_LATTICE && n == p->Np), MCPM_E_ARG, "mcpm_observe_pos_tables_vjp_f32: bad pos_mode / count");
    MCPM_REQUIRE(p, (flags & 2) && tables && nchi >= 2 && ngrow >= 2, MCPM_E_ARG, "mcpm_observe_pos_tables_vjp_f32: light cone only (flags bit 1, tables)");
    const size_t ntot = (size_t)nchi + 2 * (size_t)ngrow;
    MCPM_REQUIRE(p, ntot + 8 <= 3072 && ntot * sizeof(double) <= 60 * 1024, MCPM_E_ARG, "mcpm_observe_pos_tables_vjp_f32: tables exceed the accumulators");
    Obs og = make_obs(p, geom, flags, nchi, ngrow);
    Tables tb{tables, tables + nchi, tables + 2 * nchi, tables + 2 * nchi + ngrow, tables + 2 * nchi + 2 * ngrow};
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(p->reduce);
    unsigned *mx = reinterpret_cast<unsigned *>(acc + ntot);
    MCPM_HIP(p, hipMemsetAsync(acc, 0, (ntot + 4) * sizeof(double), p->stream));
    const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
#define LAUNCH(MO)                                                                                                                              \
    observe_tables_vjp_kernel<MO, 0><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out_bar, mx, acc);                              \
    observe_tables_vjp_kernel<MO, 1><<<nb, 256, ntot * sizeof(double), p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out_bar, mx, acc)
    if (mode == MCPM_POS_LATTICE) { LAUNCH(MCPM_POS_LATTICE); } else { LAUNCH(MCPM_POS_ABSOLUTE); }
#undef LAUNCH
    // kinds 0, 1, 2 = chi, g, f (the scale kernel's last two boundaries coincide with the end)
    lc_scale_kernel<<<(unsigned)((ntot + 255) / 256), 256, 0, p->stream>>>(acc, mx, (int)ntot, nchi, nchi + ngrow, (int)ntot, (int)ntot, table_bar);
    MCPM_LAUNCH_CHECK(p, "observe_tables_vjp_kernel");
    return MCPM_OK;
}

}  // extern "C"
