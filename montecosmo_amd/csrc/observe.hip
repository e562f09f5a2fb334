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
                                                          float *__restrict__ vel_bar, float *__restrict__ dvel_bar, double *slots) {
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
    block_add<1>(red, slots);
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
    Tables tb{tables, tables ? tables + nchi : nullptr, tables ? tables + 2 * nchi : nullptr,
              tables ? tables + 2 * nchi + ngrow : nullptr, tables ? tables + 2 * nchi + 2 * ngrow : nullptr};
    double *slots = p->reduce;
    const unsigned nb = (unsigned)((n + 255) / 256);
    StageTimer st_(p, ST_LPT, (dvel ? 84.0 : 60.0) * n);
    MCPM_HIP(p, hipMemsetAsync(slots, 0, sizeof(double) * NSLOT, p->stream));
    if (mode == MCPM_POS_LATTICE)
        observe_vjp_kernel<MCPM_POS_LATTICE><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out_bar, pos_bar, vel_bar, dvel_bar, slots);
    else
        observe_vjp_kernel<MCPM_POS_ABSOLUTE><<<nb, 256, 0, p->stream>>>(p->g, og, tb, pos, vel, dvel, n, out_bar, pos_bar, vel_bar, dvel_bar, slots);
    fold_kernel<<<1, NSLOT, 0, p->stream>>>(slots, 1, 1.0, gf_bar);
    MCPM_LAUNCH_CHECK(p, "observe_vjp_kernel");
    return MCPM_OK;
}

}  // extern "C"
