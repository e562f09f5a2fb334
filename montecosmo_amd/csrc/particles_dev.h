// Device helpers shared by the particle kernels: lattice decoding, base cell + fraction, periodic wrap.
// Index semantics follow montecosmo/nbody.py:369-377 (floor for CIC, round-half-even for NGP, Python
// modulo wrap); the int16 range of the reference (|pos| < 32767) is a documented precondition.
#pragma once
#include "mcpm_internal.h"

struct P3 {
    float x, y, z;
};

__device__ __forceinline__ P3 load3(const float *__restrict__ p, int64_t i) {
    return reinterpret_cast<const P3 *>(p)[i];
}
__device__ __forceinline__ void store3(float *__restrict__ p, int64_t i, P3 v) { reinterpret_cast<P3 *>(p)[i] = v; }
// 12-byte streaming store (`nt`): for outputs nothing reads again before they have left every cache (pm_forces' force array,
// the force mesh it gathers from once): they then do not evict the lines the gathers are re-using.  The builtin nontemporal
// store has no 96-bit form, hence the instruction itself.
typedef float mcpm_f3v __attribute__((ext_vector_type(3)));
// two / four 12-byte streaming loads issued together (one wait): a particle kernel's first loads
// (MCPM_STORE_DATA_HAZARD_NOP: mcpm_internal.h)
__device__ __forceinline__ void load3_nt2(const float *a, const float *b, int64_t i, P3 &A, P3 &B) {
    mcpm_f3v va, vb;
    const float *qa = a + 3 * i, *qb = b + 3 * i;
    asm volatile("global_load_dwordx3 %0, %2, off nt\n\tglobal_load_dwordx3 %1, %3, off nt\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(va), "=&v"(vb) : "v"(qa), "v"(qb) : "memory");
    A = P3{va.x, va.y, va.z};
    B = P3{vb.x, vb.y, vb.z};
}
__device__ __forceinline__ void load3_nt4(const float *a, const float *b, const float *c, const float *d, int64_t i, P3 &A, P3 &B, P3 &Cc,
                                          P3 &D) {
    mcpm_f3v va, vb, vc, vd;
    const float *qa = a + 3 * i, *qb = b + 3 * i, *qc = c + 3 * i, *qd = d + 3 * i;
    asm volatile("global_load_dwordx3 %0, %4, off nt\n\tglobal_load_dwordx3 %1, %5, off nt\n\tglobal_load_dwordx3 %2, %6, off nt\n\t"
                 "global_load_dwordx3 %3, %7, off nt\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(va), "=&v"(vb), "=&v"(vc), "=&v"(vd) : "v"(qa), "v"(qb), "v"(qc), "v"(qd) : "memory");
    A = P3{va.x, va.y, va.z};
    B = P3{vb.x, vb.y, vb.z};
    Cc = P3{vc.x, vc.y, vc.z};
    D = P3{vd.x, vd.y, vd.z};
}
// 12-byte streaming load: for a kernel's FIRST load of a particle (nothing else of the wave is in flight: the wait is exact)
__device__ __forceinline__ P3 load3_nt(const float *p, int64_t i) {
    mcpm_f3v v;
    const float *q = p + 3 * i;
    asm volatile("global_load_dwordx3 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    return P3{v.x, v.y, v.z};
}
__device__ __forceinline__ void store3_nt(float *p, int64_t i, float a, float b, float c) {
    const mcpm_f3v v = {a, b, c};
    float *q = p + 3 * i;
    asm volatile("global_store_dwordx3 %0, %1, off nt\n\t" MCPM_STORE_DATA_HAZARD_NOP : : "v"(q), "v"(v) : "memory");
}

// Python-style modulo for |c| < 2^16, n < 2^15.
__device__ __forceinline__ int wrapi(int c, int n) {
    if ((n & (n - 1)) == 0) return c & (n - 1);
    int q = (int)floorf((float)c / (float)n);
    int r = c - q * n;
    r += (r < 0) ? n : 0;
    r -= (r >= n) ? n : 0;
    return r;
}

// Lattice point of particle-lattice index ip on an axis with n mesh cells and p lattice points:
// q = ip*n/p = qi + qf, qi integer, qf in [0,1).
__device__ __forceinline__ void lattice_q(int ip, int n, int p, int same, int &qi, float &qf) {
    if (same) {
        qi = ip;
        qf = 0.f;
    } else {
        int qn = ip * n;
        qi = qn / p;
        qf = (float)(qn - qi * p) / (float)p;
    }
}

// Base cell id0 (unwrapped, before the stencil shift) and offset f = pos - id0 along one axis (nbody.py:375).
// t = qf + d is the coordinate relative to qi.
// even ORDER (2 CIC, 4 PCS): id0 = floor, f in [0,1].   odd ORDER (1 NGP, 3 TSC): id0 = round-half-even, f in [-1/2,1/2].
template <int ORDER>
__device__ __forceinline__ void axis_cell(int qi, float t, int &c, float &f) {
    float fl = floorf(t);
    float fr = t - fl;
    int b = qi + (int)fl;
    if (ORDER % 2 == 0) {
        c = b;
        f = fr;
    } else {
        const int up = (fr > 0.5f || (fr == 0.5f && (b & 1))) ? 1 : 0;
        c = b + up;
        f = ORDER == 1 ? 0.f : fr - (float)up;
    }
}

// 1-D assignment kernels of nbody.py:239-246 at s = idx - pos, and d/dpos K(idx - pos) = -k'(|s|) sign(s)
// (sign(0) = 0, as jax differentiates |s|).
template <int ORDER>
__device__ __forceinline__ float kern(float s) {
    const float u = fabsf(s);
    if (ORDER == 3) return u <= 0.5f ? 0.75f - u * u : 0.5f * (1.5f - u) * (1.5f - u);
    return u <= 1.f ? (4.f - 6.f * u * u + 3.f * u * u * u) * (1.f / 6.f) : (2.f - u) * (2.f - u) * (2.f - u) * (1.f / 6.f);
}
template <int ORDER>
__device__ __forceinline__ float dkern(float s) {
    const float u = fabsf(s), sg = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    float kp;
    if (ORDER == 3) kp = u <= 0.5f ? -2.f * u : -(1.5f - u);
    else kp = u <= 1.f ? (-12.f * u + 9.f * u * u) * (1.f / 6.f) : -0.5f * (2.f - u) * (2.f - u);
    return -kp * sg;
}

// Thread -> particle mapping.
// LATTICE mode: one block row-chunk per lattice row (x,y), lanes along z, so that a wave touches
// consecutive z cells; ABSOLUTE mode: flat index.
struct PIdx {
    int64_t i;       // particle index
    int ipx, ipy, ipz;
    bool valid;
};

template <int MODE>
__device__ __forceinline__ PIdx particle_index(const Geom &g, int64_t n) {
    PIdx r;
    if (MODE == MCPM_POS_LATTICE && g.patch) {
        // A workgroup = four waves on a 2 x 2 (x, y) patch of lattice rows, 64 consecutive z each (lattice_launch: 256 threads).
        // What bounds the CIC gathers is the mesh rows a wave pulls into the CU's L1 -- 4 (x, y) rows of ~900 bytes per wave; a
        // second corner in a row already fetched is nearly free (tools/gather_pair_bench.hip) -- and the four waves of a patch
        // touch 3 x 3 rows between them instead of 4 x 4.  Workgroups in plain memory order (z chunk, then y pair, then x
        // pair), consecutive ones on consecutive XCDs as the hardware deals them: the kernels' four to seven particle streams
        // walk their arrays sequentially over the whole chip.  (Per-XCD contiguous runs of that order: pm_forces 4.14 against
        // 4.08 ms, step 11.15 against 11.10; runs with 2 / 4 / 8 / 16 x pairs grouped first, which re-read the shared x rows
        // from one L2: kick+drift 1.48 / 1.55 / 1.56 / 1.54 against 1.46 ms.)
        const unsigned b = blockIdx.x, nyp = g.py >> 1, nzc = g.pz >> 6;
        const unsigned zc = b % nzc, r2 = b / nzc;
        const unsigned yp = r2 % nyp, xp = r2 / nyp;
        const unsigned w = threadIdx.x >> 6;
        r.ipx = 2 * xp + (w & 1);
        r.ipy = 2 * yp + (w >> 1);
        r.ipz = zc * 64 + (threadIdx.x & 63);
        r.valid = true;
        r.i = ((int64_t)r.ipx * g.py + r.ipy) * g.pz + r.ipz;
    } else if (MODE == MCPM_POS_LATTICE) {
        // Block -> (lattice row, z chunk).  Two locality remaps, speed only:
        //  * blocks b, b+8, ... share an XCD (and its L2): give each XCD a contiguous run of virtual blocks;
        //  * within the run, 8 consecutive x planes of one (y, z chunk) come first, so that the mesh rows a
        //    CIC stencil shares between x and x+1 (and y, y+1 of the next blocks) are re-read from L2.
        const int cpr = (g.pz + blockDim.x - 1) / blockDim.x;
        const unsigned nb = gridDim.x, b = blockIdx.x;
        const unsigned vb = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
        const unsigned XB = (g.px % 8 == 0) ? 8 : 1;
        const unsigned xi = vb % XB, r1 = vb / XB;
        const unsigned chunk = r1 % cpr, r2 = r1 / cpr;
        r.ipy = r2 % g.py;
        r.ipx = (r2 / g.py) * XB + xi;
        r.ipz = chunk * blockDim.x + threadIdx.x;
        r.valid = r.ipz < g.pz;
        r.i = ((int64_t)r.ipx * g.py + r.ipy) * g.pz + r.ipz;
    } else {
        r.i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        r.ipx = r.ipy = r.ipz = 0;
        r.valid = r.i < n;
    }
    return r;
}

// Launch shape of the lattice-mode kernels (host side of particle_index)
static inline void lattice_launch(const Geom &g, dim3 &grid, dim3 &block) {
    if (g.patch) {
        block = dim3(256);
        grid = dim3((unsigned)((int64_t)(g.px / 2) * (g.py / 2) * (g.pz / 64)));
        return;
    }
    int bs = g.pz >= 256 ? 256 : ((g.pz + 63) / 64) * 64;
    int cpr = (g.pz + bs - 1) / bs;
    block = dim3(bs);
    grid = dim3((unsigned)((int64_t)g.px * g.py * cpr));
}

// Base cell + fractions of one particle.
template <int MODE, int ORDER>
__device__ __forceinline__ void locate(const Geom &g, const PIdx &pi, P3 d, int (&c)[3], float (&f)[3]) {
    int qx = 0, qy = 0, qz = 0;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (MODE == MCPM_POS_LATTICE) {
        lattice_q(pi.ipx, g.nx, g.px, g.same_lattice, qx, fx);
        qx += g.xoff;  // slab mode: lattice plane 0 sits at mesh plane `ghost`
        lattice_q(pi.ipy, g.ny, g.py, g.same_lattice, qy, fy);
        lattice_q(pi.ipz, g.nz, g.pz, g.same_lattice, qz, fz);
    }
    axis_cell<ORDER>(qx, fx + d.x, c[0], f[0]);
    axis_cell<ORDER>(qy, fy + d.y, c[1], f[1]);
    axis_cell<ORDER>(qz, fz + d.z, c[2], f[2]);
}

// ------------------------------------------------------------------------------------------------
// NGP cell of a lattice point (read_order = 1 at pos = regular_pos, nbody.py:984-985): exact integer
// round-half-even of ip*n/p.
__device__ __forceinline__ int lattice_ngp(int ip, int n, int p, int same) {
    if (same) return ip;
    int qn = ip * n, qi = qn / p, qr = qn - qi * p;
    int c = qi + ((2 * qr > p || (2 * qr == p && (qi & 1))) ? 1 : 0);
    return c >= n ? c - n : c;
}

__device__ __forceinline__ int64_t lattice_cell(const Geom &g, const PIdx &pi) {
    int cx = lattice_ngp(pi.ipx, g.nx, g.px, g.same_lattice) + g.xoff;
    int cy = lattice_ngp(pi.ipy, g.ny, g.py, g.same_lattice);
    int cz = lattice_ngp(pi.ipz, g.nz, g.pz, g.same_lattice);
    return ((int64_t)cx * g.ny + cy) * g.nz + cz;
}

// Flat mesh offsets of the ORDER^3 stencil (NGP 1, CIC 2, TSC 3, PCS 4 points per axis), periodic.  The stencil
// starts at id0 - (ORDER-1)/2 (nbody.py:376).
template <int ORDER>
struct Stencil {
    static constexpr int NP = ORDER < 2 ? 2 : ORDER;
    int64_t xo[NP], yo[NP];
    int zo[NP];
    __device__ __forceinline__ Stencil(const Geom &g, const int (&c)[3]) {
        if (ORDER <= 2) {
            // slab mode: x is not periodic on the ghost-extended mesh; clamp (only particles displaced beyond the
            // ghost width are affected, see DESIGN.md "Multi-GPU")
            int x0 = g.xslab ? min(max(c[0], 0), g.nx - ORDER) : wrapi(c[0], g.nx);
            int y0 = wrapi(c[1], g.ny), z0 = wrapi(c[2], g.nz);
            xo[0] = (int64_t)x0 * g.ny * g.nz;
            yo[0] = (int64_t)y0 * g.nz;
            zo[0] = z0;
            if (ORDER == 2) {
                int x1 = x0 + 1 == g.nx ? 0 : x0 + 1;
                int y1 = y0 + 1 == g.ny ? 0 : y0 + 1;
                int z1 = z0 + 1 == g.nz ? 0 : z0 + 1;
                xo[1] = (int64_t)x1 * g.ny * g.nz;
                yo[1] = (int64_t)y1 * g.nz;
                zo[1] = z1;
            } else {
                xo[1] = xo[0];
                yo[1] = yo[0];
                zo[1] = zo[0];
            }
        } else {
            constexpr int SH = -((ORDER - 1) / 2);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int x = g.xslab ? min(max(c[0] + SH + j, 0), g.nx - 1) : wrapi(c[0] + SH + j, g.nx);
                xo[j] = (int64_t)x * g.ny * g.nz;
                yo[j] = (int64_t)wrapi(c[1] + SH + j, g.ny) * g.nz;
                zo[j] = wrapi(c[2] + SH + j, g.nz);
            }
        }
    }
};

// per-axis weights (and d/dpos) of the ORDER >= 3 stencils from the offsets f = pos - id0
template <int ORDER, bool GRAD>
__device__ __forceinline__ void axis_weights(float f, float (&w)[ORDER], float (&dw)[ORDER]) {
    constexpr int SH = -((ORDER - 1) / 2);
#pragma unroll
    for (int j = 0; j < ORDER; ++j) {
        const float sj = (float)(SH + j) - f;
        w[j] = kern<ORDER>(sj);
        dw[j] = GRAD ? dkern<ORDER>(sj) : 0.f;
    }
}

// Trilinear combination of the 8 corner values v[xyz] (and its gradient w.r.t. the position).  d/dpos of
// K(c - x) = sign(c - x): -1 for the lower corner (0 when the fraction is exactly 0, jax's sign(0) = 0), +1 for the upper.
template <bool GRAD>
__device__ __forceinline__ void cic_combine(float v000, float v001, float v010, float v011, float v100, float v101, float v110,
                                            float v111, const float (&f)[3], float &val, float &gx, float &gy, float &gz) {
    float ax = 1.f - f[0], bx = f[0], ay = 1.f - f[1], by = f[1], az = 1.f - f[2], bz = f[2];
    // collapse z, then y, then x
    float v00 = az * v000 + bz * v001, v01 = az * v010 + bz * v011;
    float v10 = az * v100 + bz * v101, v11 = az * v110 + bz * v111;
    float v0 = ay * v00 + by * v01, v1 = ay * v10 + by * v11;
    val = ax * v0 + bx * v1;
    if (GRAD) {
        float lx = f[0] > 0.f ? 1.f : 0.f, ly = f[1] > 0.f ? 1.f : 0.f, lz = f[2] > 0.f ? 1.f : 0.f;
        gx = v1 - lx * v0;
        float d0 = v01 - ly * v00, d1 = v11 - ly * v10;
        gy = ax * d0 + bx * d1;
        float e00 = v001 - lz * v000, e01 = v011 - lz * v010, e10 = v101 - lz * v100, e11 = v111 - lz * v110;
        gz = ax * (ay * e00 + by * e01) + bx * (ay * e10 + by * e11);
    } else {
        gx = gy = gz = 0.f;
    }
}

// Trilinear value and gradient of one mesh at (cell, frac).  d/dpos of K(c - x) = sign(c - x):
// -1 for the lower corner (0 when the fraction is exactly 0, jax's sign(0) = 0), +1 for the upper.
template <int ORDER, bool GRAD>
__device__ __forceinline__ void interp(const float *__restrict__ m, const Stencil<ORDER> &s, const float (&f)[3],
                                       float &val, float &gx, float &gy, float &gz) {
    if (ORDER == 1) {
        val = m[s.xo[0] + s.yo[0] + s.zo[0]];
        gx = gy = gz = 0.f;
        return;
    }
    if (ORDER >= 3) {
        constexpr int NPG = ORDER < 3 ? 3 : ORDER;
        float wx[NPG], wy[NPG], wz[NPG], dx[NPG], dy[NPG], dz[NPG];
        axis_weights<NPG, GRAD>(f[0], wx, dx);
        axis_weights<NPG, GRAD>(f[1], wy, dy);
        axis_weights<NPG, GRAD>(f[2], wz, dz);
        float v = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int a = 0; a < NPG; ++a)
#pragma unroll
            for (int b = 0; b < NPG; ++b) {
                float r0 = 0.f, r1 = 0.f;  // sum over z of m*wz and m*dz
#pragma unroll
                for (int e = 0; e < NPG; ++e) {
                    const float mv = m[s.xo[a] + s.yo[b] + s.zo[e]];
                    r0 += mv * wz[e];
                    if (GRAD) r1 += mv * dz[e];
                }
                v += wx[a] * wy[b] * r0;
                if (GRAD) {
                    ax += dx[a] * wy[b] * r0;
                    ay += wx[a] * dy[b] * r0;
                    az += wx[a] * wy[b] * r1;
                }
            }
        val = v;
        gx = ax;
        gy = ay;
        gz = az;
        return;
    }
    // z and z+1 are adjacent in memory: one 8-byte gather per (x, y) corner instead of two 4-byte ones (the
    // gathers are bound by address processing, not bytes).  At the periodic wrap (z = nz-1, z+1 = 0) the pair
    // is read one cell lower and the z = 0 cell separately.
    struct __attribute__((packed, aligned(4))) F2 {
        float a, b;
    };
    const bool wrap = s.zo[1] != s.zo[0] + 1;
    const int zc = wrap ? s.zo[0] - 1 : s.zo[0];
    const F2 p00 = *reinterpret_cast<const F2 *>(m + s.xo[0] + s.yo[0] + zc);
    const F2 p01 = *reinterpret_cast<const F2 *>(m + s.xo[0] + s.yo[1] + zc);
    const F2 p10 = *reinterpret_cast<const F2 *>(m + s.xo[1] + s.yo[0] + zc);
    const F2 p11 = *reinterpret_cast<const F2 *>(m + s.xo[1] + s.yo[1] + zc);
    float v000 = wrap ? p00.b : p00.a, v001 = p00.b, v010 = wrap ? p01.b : p01.a, v011 = p01.b;
    float v100 = wrap ? p10.b : p10.a, v101 = p10.b, v110 = wrap ? p11.b : p11.a, v111 = p11.b;
    if (wrap) {
        v001 = m[s.xo[0] + s.yo[0] + s.zo[1]];
        v011 = m[s.xo[0] + s.yo[1] + s.zo[1]];
        v101 = m[s.xo[1] + s.yo[0] + s.zo[1]];
        v111 = m[s.xo[1] + s.yo[1] + s.zo[1]];
    }
    cic_combine<GRAD>(v000, v001, v010, v011, v100, v101, v110, v111, f, val, gx, gy, gz);
}

// Three force components at once.  IL = false: three meshes M apart (m + c*M); IL = true: one interleaved mesh
// [cell][3] (what the fused Poisson solve writes for the step kernels): a CIC corner is then ONE 12-byte gather for
// the three components, 8 gathers per particle instead of 12 z-pair gathers -- the gathers are bound by address
// processing per instruction, not by bytes (tools/gather_bench.hip).
template <int ORDER, bool GRAD, bool IL>
__device__ __forceinline__ void interp3(const float *__restrict__ m, int64_t M, const Stencil<ORDER> &s, const float (&f)[3],
                                        float (&F)[3], float (&G)[3][3]) {
    if (!IL) {
#pragma unroll
        for (int c = 0; c < 3; ++c) interp<ORDER, GRAD>(m + c * M, s, f, F[c], G[c][0], G[c][1], G[c][2]);
        return;
    }
    struct __attribute__((packed, aligned(4))) F3 {
        float a, b, c;
    };
    if (ORDER == 2) {
        F3 v[8];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    v[a * 4 + b * 2 + e] = *reinterpret_cast<const F3 *>(m + 3 * (int64_t)(s.xo[a] + s.yo[b] + s.zo[e]));
        cic_combine<GRAD>(v[0].a, v[1].a, v[2].a, v[3].a, v[4].a, v[5].a, v[6].a, v[7].a, f, F[0], G[0][0], G[0][1], G[0][2]);
        cic_combine<GRAD>(v[0].b, v[1].b, v[2].b, v[3].b, v[4].b, v[5].b, v[6].b, v[7].b, f, F[1], G[1][0], G[1][1], G[1][2]);
        cic_combine<GRAD>(v[0].c, v[1].c, v[2].c, v[3].c, v[4].c, v[5].c, v[6].c, v[7].c, f, F[2], G[2][0], G[2][1], G[2][2]);
        return;
    }
    if (ORDER == 1) {
        const F3 v = *reinterpret_cast<const F3 *>(m + 3 * (int64_t)(s.xo[0] + s.yo[0] + s.zo[0]));
        F[0] = v.a; F[1] = v.b; F[2] = v.c;
#pragma unroll
        for (int c = 0; c < 3; ++c) G[c][0] = G[c][1] = G[c][2] = 0.f;
        return;
    }
    constexpr int NPG = ORDER < 3 ? 3 : ORDER;
    float wx[NPG], wy[NPG], wz[NPG], dx[NPG], dy[NPG], dz[NPG];
    axis_weights<NPG, GRAD>(f[0], wx, dx);
    axis_weights<NPG, GRAD>(f[1], wy, dy);
    axis_weights<NPG, GRAD>(f[2], wz, dz);
#pragma unroll
    for (int c = 0; c < 3; ++c) F[c] = G[c][0] = G[c][1] = G[c][2] = 0.f;
#pragma unroll
    for (int a = 0; a < NPG; ++a)
#pragma unroll
        for (int b = 0; b < NPG; ++b) {
            float r0[3] = {0.f, 0.f, 0.f}, r1[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < NPG; ++e) {
                const F3 mv = *reinterpret_cast<const F3 *>(m + 3 * (int64_t)(s.xo[a] + s.yo[b] + s.zo[e]));
                const float t[3] = {mv.a, mv.b, mv.c};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    r0[c] += t[c] * wz[e];
                    if (GRAD) r1[c] += t[c] * dz[e];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                F[c] += wx[a] * wy[b] * r0[c];
                if (GRAD) {
                    G[c][0] += dx[a] * wy[b] * r0[c];
                    G[c][1] += wx[a] * dy[b] * r0[c];
                    G[c][2] += wx[a] * wy[b] * r1[c];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Wave reductions by DPP (row shifts within 16-lane rows, then row_bcast:15 / :31): 6 VALU instructions per value where the
// __shfl-based versions are 12 ds_bpermute (doubles) through the LDS crossbar -- the adjoint particle kernel issued 42 of them
// per wave next to its 30 vector-memory instructions.  Every lane of the wave must be active; the result is in lane 63.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xf, true); }
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __int_as_float(dpp_i<0x111, 0xf>(__float_as_int(v)));   // row_shr:1
    v += __int_as_float(dpp_i<0x112, 0xf>(__float_as_int(v)));   // row_shr:2
    v += __int_as_float(dpp_i<0x114, 0xf>(__float_as_int(v)));   // row_shr:4
    v += __int_as_float(dpp_i<0x118, 0xf>(__float_as_int(v)));   // row_shr:8   -> lane 15 of every row holds the row's sum
    v += __int_as_float(dpp_i<0x142, 0xa>(__float_as_int(v)));   // row_bcast:15 into rows 1 and 3
    v += __int_as_float(dpp_i<0x143, 0xc>(__float_as_int(v)));   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}
__device__ __forceinline__ unsigned wave_umax_dpp(unsigned m) {
    m = max(m, (unsigned)dpp_i<0x111, 0xf>((int)m));
    m = max(m, (unsigned)dpp_i<0x112, 0xf>((int)m));
    m = max(m, (unsigned)dpp_i<0x114, 0xf>((int)m));
    m = max(m, (unsigned)dpp_i<0x118, 0xf>((int)m));
    m = max(m, (unsigned)dpp_i<0x142, 0xa>((int)m));
    m = max(m, (unsigned)dpp_i<0x143, 0xc>((int)m));
    return m;
}
