// Particle <-> mesh kernels of libmcpm.so for gfx950: paint, read, fused kick/drift and their VJPs.
//
// Reference semantics: montecosmo/nbody.py:365-396 (paint), :398-427 (read), :933-944 (kick, drift).
//
// Paint design (MI355X-first, not a translation of the reference's 8 scatter-add passes):
//   * fast path `paint_tile_kernel`: particles are stored in Lagrangian (lattice) order as fp32
//     displacements from their lattice point.  One workgroup owns one Eulerian tile of the mesh in
//     LDS and *pulls* every lattice particle whose lattice point lies within `H` cells of the tile
//     (coalesced 12-byte loads, wave lanes along z), depositing with LDS float atomics only the
//     stencil points that fall inside its own tile.  The tile is then written to HBM with plain
//     16-byte stores: no global atomics, no sort.  Particles displaced by more than H cells are
//     appended to an outlier list by their home tile and deposited by a small global-atomic kernel.
//   * generic path `paint_atomic_kernel`: arbitrary (absolute) positions, global float atomics.
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
// tiled paint (lattice displacements, lattice == mesh, CIC)
// THREADS x U particle loads are issued before any is consumed: the pull loop is otherwise bound by load
// latency (one 12-byte load in flight per thread moves < 1 TB/s chip-wide).
// Tile of this workgroup.  Blocks b, b+8, ... run on the same XCD (and share its 4 MB L2), in dispatch order b/8.
// order 0 (default): each XCD works through one contiguous run of tiles (z fastest: a pencil of tiles whose flushes and
// particle reads are contiguous in memory).  order 1: compact bricks of tiles (2x4x4 ...) per XCD, meant to keep the halo
// particles neighbouring tiles re-read in that L2 -- measured SLOWER at 512^3 (density paint 1.34 vs 1.15 ms, three-component
// 3.10 vs 2.98 ms, tools/time_paint_halo.py): memory contiguity of the pencil wins over L2 reuse.  Kept as a knob.
__device__ __forceinline__ void tile_of_block(int order, int ntx, int nty, int ntz, int per, int &tx, int &ty, int &tz) {
    const int nb = gridDim.x, b = blockIdx.x;
    if (order == 1 && nb % 8 == 0 && ntx % 8 == 0) {
        const int sx = ntx / 8;
        const int bxk = (sx % 2 == 0) ? 2 : 1;
        int byk = 4, bzk = per / (bxk * 4);          // 32: 2x4x4 or 1x4x8;  128: 2x4x16 -> prefer 2x8x8 below
        if (per >= 128) { byk = 8; bzk = per / (bxk * 8); }
        if (bzk >= 1 && nty % byk == 0 && ntz % bzk == 0) {
            const int xcd = b % 8, v = b / 8, pb = bxk * byk * bzk;
            const int brick = v / pb, w = v % pb;
            const int nbz = ntz / bzk, nby = nty / byk;
            const int kz = brick % nbz, r = brick / nbz, ky = r % nby, kx = r / nby;
            const int wz = w % bzk, r2 = w / bzk, wy = r2 % byk, wx = r2 / byk;
            tx = xcd * sx + kx * bxk + wx;
            ty = ky * byk + wy;
            tz = kz * bzk + wz;
            return;
        }
    }
    const int t = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;   // contiguous run of tiles per XCD
    tz = t % ntz;
    const int tt = t / ntz;
    ty = tt % nty;
    tx = tt / nty;
}

// c in (-n, 2n) -> [0, n): a mask on power-of-two sizes (the condition is uniform, the compiler branches on it once)
__device__ __forceinline__ int wrap_once(int c, int n) {
    if ((n & (n - 1)) == 0) return c & (n - 1);
    c += c < 0 ? n : 0;
    c -= c >= n ? n : 0;
    return c;
}

__device__ __forceinline__ int cvt_rpi(float x) {   // floor(x + 0.5)
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int BX, int BY, int BZ, int H, bool WEIGHTED, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void paint_tile_kernel(Geom g, const float *__restrict__ disp,
                                                             const float *__restrict__ w, int64_t wstride, float wscalar,
                                                             float *__restrict__ mesh, int accumulate,
                                                             int *__restrict__ outliers, int *__restrict__ ocount) {
    constexpr int WX = BX + 2 * H + 1, WY = BY + 2 * H + 1, WZ = BZ + 2 * H + 1, NW = WX * WY * WZ;
    constexpr int NT = BX * BY * BZ;
    // double accumulators: on gfx950 LDS ds_add_f64 sustains ~4-5 lanes/clk/CU, ds_add_f32 only ~0.3
    // (tools/lds_atomic_bench.hip), and the sums become insensitive to arrival order at fp32 output precision.
    // Unweighted paint (the density of the force cycle): the stencil weights are non-negative and at most 1, so they are
    // accumulated as 2^-30 fixed point in 64-bit INTEGER atomics (1.6x the f64 rate under bank conflicts, same
    // instruction count): exact order-independent sums, no overflow below 2^34 deposits per cell; the scalar weight is
    // applied when the tile is written.
    constexpr bool FXU = !WEIGHTED;
    __shared__ double tile[NT];
    unsigned long long *utile = reinterpret_cast<unsigned long long *>(tile);

    int tx, ty, tz;
    tile_of_block(g.tile_order, g.nx / BX, g.ny / BY, g.nz / BZ, 32 * ((160 * 1024) / (int)(sizeof(double) * NT) < 2048 / THREADS ? (160 * 1024) / (int)(sizeof(double) * NT) : 2048 / THREADS), tx, ty, tz);
    const int x0 = tx * BX, y0 = ty * BY, z0 = tz * BZ;

    double2 *tile2 = reinterpret_cast<double2 *>(tile);
    for (int i = threadIdx.x; i < NT / 2; i += THREADS) tile2[i] = make_double2(0., 0.);
    __syncthreads();

    for (int j0 = threadIdx.x; j0 < NW; j0 += THREADS * U) {
        P3 d[U];
        float wt[U];
        int rxs[U], rys[U], rzs[U], gis[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * THREADS;
            bool ok = j < NW;
            const int jz = j % WZ, r = j / WZ, jy = r % WY, jx = r / WY;
            const int rx = jx - (H + 1), ry = jy - (H + 1), rz = jz - (H + 1);  // lattice point relative to the tile
            int gx = x0 + rx, gy = y0 + ry, gz = z0 + rz;
            bool inx = true;
            if (g.xslab) {  // ghost-extended slab: lattice planes are mesh planes [xoff, xoff + px), no wrap
                gx -= g.xoff;
                inx = (unsigned)gx < (unsigned)g.px;
            } else {
                gx = wrap_once(gx, g.nx);
            }
            gy = wrap_once(gy, g.ny);
            gz = wrap_once(gz, g.nz);
            ok = ok && inx;
            const int gi = ok ? (gx * g.ny + gy) * g.nz + gz : -1;
            rxs[u] = rx;
            rys[u] = ry;
            rzs[u] = rz;
            gis[u] = gi;
            if (ok) {
                d[u] = load3(disp, gi);
                wt[u] = WEIGHTED ? w[(int64_t)gi * wstride] : wscalar;
            } else {
                d[u] = P3{0.f, 0.f, 0.f};
                wt[u] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gis[u] < 0) continue;
            const int rx = rxs[u], ry = rys[u], rz = rzs[u];
            const float fx = floorf(d[u].x), fy = floorf(d[u].y), fz = floorf(d[u].z);
            // outliers: |floor(d)| > H on any axis (NaN compares false everywhere -> treated as outlier)
            const bool inl = fx >= (float)-H && fx <= (float)H && fy >= (float)-H && fy <= (float)H && fz >= (float)-H &&
                             fz <= (float)H;
            if (!inl) {
                const bool home = (unsigned)rx < (unsigned)BX && (unsigned)ry < (unsigned)BY && (unsigned)rz < (unsigned)BZ;
                if (home) {
                    int k = atomicAdd(ocount, 1);
                    outliers[k] = gis[u];
                }
                continue;
            }
            const int cx = rx + (int)fx, cy = ry + (int)fy, cz = rz + (int)fz;
            if (cx < -1 || cx >= BX || cy < -1 || cy >= BY || cz < -1 || cz >= BZ) continue;
            const float tx1 = d[u].x - fx, ty1 = d[u].y - fy, tz1 = d[u].z - fz;
            const float sc = FXU ? 1073741824.f : 1.f;      // 2^30: exact scaling of the x weights
            const float kx[2] = {(1.f - tx1) * sc, tx1 * sc}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
            // one LDS base address, corners at immediate offsets; a corner outside the tile is skipped
            const bool vx[2] = {cx >= 0, cx < BX - 1}, vy[2] = {cy >= 0, cy < BY - 1}, vz[2] = {cz >= 0, cz < BZ - 1};
            const int base = (cx * BY + cy) * BZ + cz;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const float wxy = FXU ? kx[a] * ky[bb] : wt[u] * kx[a] * ky[bb];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (vx[a] && vy[bb] && vz[e]) {
                            const int q = base + (a * BY + bb) * BZ + e;
                            if (FXU) atomicAdd(utile + q, (unsigned long long)(unsigned)cvt_rpi(wxy * kz[e]));
                            else atomicAdd(tile + q, (double)(wxy * kz[e]));
                        }
                }
        }
    }
    __syncthreads();

    for (int i = threadIdx.x; i < NT / 4; i += THREADS) {
        const int lz = (i % (BZ / 4)) * 4, r = i / (BZ / 4), ly = r % BY, lx = r / BY;
        float4 v;
        if (FXU) {
            const double s = (double)wscalar * 9.313225746154785e-10;      // 2^-30
            v = make_float4((float)((double)utile[4 * i] * s), (float)((double)utile[4 * i + 1] * s),
                            (float)((double)utile[4 * i + 2] * s), (float)((double)utile[4 * i + 3] * s));
        } else {
            const double2 lo = tile2[2 * i], hi = tile2[2 * i + 1];
            v = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
        }
        float4 *dst = reinterpret_cast<float4 *>(mesh + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz);
        if (accumulate) {
            float4 o = *dst;
            v.x += o.x;
            v.y += o.y;
            v.z += o.z;
            v.w += o.w;
        }
        *dst = v;
    }
}

// Three weighted paints at once (the adjoint of a three-component read: weights[N][3] -> three meshes M apart).
// Same pull scheme; the tile is 16x16x16 so that three f64 accumulators fit in LDS (96 KB).  One visit loads the
// displacement and the three weights (12 + 12 bytes) and does the index / fraction arithmetic once.
template <int B, int H, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void paint3_tile_kernel(Geom g, const float *__restrict__ disp,
                                                              const float *__restrict__ w3, float *__restrict__ mesh,
                                                              int64_t M, int accumulate, int *__restrict__ outliers,
                                                              int *__restrict__ ocount, const int *__restrict__ redo = nullptr) {
    constexpr int W = B + 2 * H + 1, NW = W * W * W, NT = B * B * B;
    __shared__ double tile[3 * NT];
    int tx, ty, tz;
    if (redo) {   // second pass of the fixed-point paint: only the tiles it flagged (redo[0] = count, then tile indices);
                  // their outliers are already on the list
        if ((int)blockIdx.x >= redo[0]) return;
        const int t = redo[1 + blockIdx.x], ntz = g.nz / B, nty = g.ny / B;
        tz = t % ntz;
        ty = (t / ntz) % nty;
        tx = t / (ntz * nty);
    } else
        tile_of_block(g.tile_order, g.nx / B, g.ny / B, g.nz / B, 32, tx, ty, tz);   // 96 KB of LDS: one workgroup per CU, 32 per XCD
    const int x0 = tx * B, y0 = ty * B, z0 = tz * B;
    double2 *tile2 = reinterpret_cast<double2 *>(tile);
    for (int i = threadIdx.x; i < 3 * NT / 2; i += THREADS) tile2[i] = make_double2(0., 0.);
    __syncthreads();

    for (int j0 = threadIdx.x; j0 < NW; j0 += THREADS * U) {
        P3 d[U], wt[U];
        int rxs[U], rys[U], rzs[U], gis[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * THREADS;
            bool ok = j < NW;
            const int jz = j % W, r = j / W, jy = r % W, jx = r / W;
            const int rx = jx - (H + 1), ry = jy - (H + 1), rz = jz - (H + 1);
            int gx = x0 + rx, gy = y0 + ry, gz = z0 + rz;
            bool inx = true;
            if (g.xslab) {
                gx -= g.xoff;
                inx = (unsigned)gx < (unsigned)g.px;
            } else {
                gx = wrap_once(gx, g.nx);
            }
            gy = wrap_once(gy, g.ny);
            gz = wrap_once(gz, g.nz);
            ok = ok && inx;
            const int gi = ok ? (gx * g.ny + gy) * g.nz + gz : -1;
            rxs[u] = rx;
            rys[u] = ry;
            rzs[u] = rz;
            gis[u] = gi;
            if (ok) {
                d[u] = load3(disp, gi);
                wt[u] = load3(w3, gi);
            } else {
                d[u] = P3{0.f, 0.f, 0.f};
                wt[u] = P3{0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gis[u] < 0) continue;
            const int rx = rxs[u], ry = rys[u], rz = rzs[u];
            const float fx = floorf(d[u].x), fy = floorf(d[u].y), fz = floorf(d[u].z);
            const bool inl = fx >= (float)-H && fx <= (float)H && fy >= (float)-H && fy <= (float)H && fz >= (float)-H &&
                             fz <= (float)H;
            if (!inl) {
                const bool home = (unsigned)rx < (unsigned)B && (unsigned)ry < (unsigned)B && (unsigned)rz < (unsigned)B;
                if (home && !redo) {
                    int k = atomicAdd(ocount, 1);
                    outliers[k] = gis[u];
                }
                continue;
            }
            const int cx = rx + (int)fx, cy = ry + (int)fy, cz = rz + (int)fz;
            if (cx < -1 || cx >= B || cy < -1 || cy >= B || cz < -1 || cz >= B) continue;
            const float tx1 = d[u].x - fx, ty1 = d[u].y - fy, tz1 = d[u].z - fz;
            const float kx[2] = {1.f - tx1, tx1}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int x = cx + a;
                if ((unsigned)x >= (unsigned)B) continue;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int y = cy + bb;
                    if ((unsigned)y >= (unsigned)B) continue;
                    const float kxy = kx[a] * ky[bb];
                    double *row = tile + (x * B + y) * B;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int z = cz + e;
                        if ((unsigned)z < (unsigned)B) {
                            const float k = kxy * kz[e];
                            atomicAdd(row + z, (double)(wt[u].x * k));
                            atomicAdd(row + NT + z, (double)(wt[u].y * k));
                            atomicAdd(row + 2 * NT + z, (double)(wt[u].z * k));
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    for (int i = threadIdx.x; i < 3 * NT / 4; i += THREADS) {
        const int c = i / (NT / 4), ii = i - c * (NT / 4);
        const int lz = (ii % (B / 4)) * 4, r = ii / (B / 4), ly = r % B, lx = r / B;
        const double2 lo = tile2[2 * i], hi = tile2[2 * i + 1];
        float4 v = make_float4((float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y);
        float4 *dst = reinterpret_cast<float4 *>(mesh + c * M + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz);
        if (accumulate) {
            float4 o = *dst;
            v.x += o.x;
            v.y += o.y;
            v.z += o.z;
            v.w += o.w;
        }
        *dst = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Fixed-point three-component paint.  An LDS f64 atomic costs about twice a 64-bit integer one under the bank
// conflicts of real deposits and 24 of them per particle bound the f64 kernel above (tools/lds_atomic_bench.hip).
// Here the three weighted corner contributions are rounded to 32-bit fixed point with a common power-of-two scale
// S = 2^24 / 2^e (2^e <= max|w| < 2^(e+1), so one contribution is below 2^25 and a cell holds 64 maximal ones) and
// travel in TWO 64-bit integer atomics per corner:
//     word A = c0 + 2^32 c1        word B = c2 + 2^32 bound,   bound += max_c |contribution_c| / 2^11 + 1 (rounded up)
// A signed low field added as a sign-extended 64-bit number leaves the high field exact as long as the low field's
// true sum fits 32 bits, and modular arithmetic makes intermediate wrap-arounds harmless, so the sums are exact
// integers and independent of the arrival order (bitwise reproducible).  `bound` proves it: a component field can only
// leave the int32 range if sum |contribution| >= 2^31, i.e. bound >= 2^20; the bound field itself cannot overflow
// (< 2^14 + 1 per deposit, < 2^17 deposits per cell).  A tile holding a cell with bound >= 2^19 is not written: it
// is appended to the redo list and painted by the f64 kernel.  Non-finite or tiny (< 2^-97) max|w| sends every tile
// there.  Rounding: half a unit per deposit = max|w| 2^-25, so the mesh differs from the exact sums by
// ~1e-8 max|w| per cell (the f32 conversion of the output costs 6e-8 relative).
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ w, int64_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    unsigned bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned b = __float_as_uint(w[i]) & 0x7fffffffu;
        bad |= b >= 0x7f800000u;
        m = fmaxf(m, __uint_as_float(b));
    }
    unsigned b = bad ? 0x7fc00000u : __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out + (blockIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE, b);
}

template <int B, int H, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void paint3_fx_kernel(Geom g, const float *__restrict__ disp,
                                                           const float *__restrict__ w3, float *__restrict__ mesh, int64_t M,
                                                           int accumulate, int *__restrict__ outliers, int *__restrict__ ocount,
                                                           const unsigned *__restrict__ wmax_bits, int *__restrict__ redo) {
    constexpr int W = B + 2 * H + 1, NW = W * W * W, NT = B * B * B;
    typedef unsigned long long u64;
    __shared__ u64 tile[2 * NT];   // 64 KB: two workgroups per CU
    __shared__ int flagged;
    int tx, ty, tz;
    tile_of_block(g.tile_order, g.nx / B, g.ny / B, g.nz / B, 64, tx, ty, tz);
    const int x0 = tx * B, y0 = ty * B, z0 = tz * B;
    unsigned wb = wmax_bits[(threadIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE];   // maximum over the slots, in every wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wb = max(wb, (unsigned)__shfl_xor((int)wb, o));
    const unsigned be = wb >> 23;
    if (be < 30u || be > 254u) {     // zero / tiny / non-finite weights: the f64 kernel paints this tile
        if (wb == 0u) {              // all weights are zero: the tile is zero (or unchanged)
            if (!accumulate)
                for (int i = threadIdx.x; i < 3 * NT / 4; i += THREADS) {
                    const int cc = i / (NT / 4), ii = i - cc * (NT / 4);
                    const int lz = (ii % (B / 4)) * 4, r = ii / (B / 4), ly = r % B, lx = r / B;
                    *reinterpret_cast<float4 *>(mesh + cc * M + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz) =
                        make_float4(0.f, 0.f, 0.f, 0.f);
                }
            return;
        }
        if (threadIdx.x == 0) redo[1 + atomicAdd(redo, 1)] = (tx * (g.ny / B) + ty) * (g.nz / B) + tz;
        return;
    }
    const float S = __uint_as_float((278u - be) << 23), Sinv = __uint_as_float((be - 24u) << 23);   // 2^(24-e), 2^(e-24)
    if (threadIdx.x == 0) flagged = 0;
    for (int i = threadIdx.x; i < 2 * NT; i += THREADS) tile[i] = 0ull;
    __syncthreads();

    for (int j0 = threadIdx.x; j0 < NW; j0 += THREADS * U) {
        P3 d[U], wt[U];
        int rxs[U], rys[U], rzs[U], gis[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * THREADS;
            bool ok = j < NW;
            const int jz = j % W, r = j / W, jy = r % W, jx = r / W;
            const int rx = jx - (H + 1), ry = jy - (H + 1), rz = jz - (H + 1);
            int gx = x0 + rx, gy = y0 + ry, gz = z0 + rz;
            bool inx = true;
            if (g.xslab) {
                gx -= g.xoff;
                inx = (unsigned)gx < (unsigned)g.px;
            } else {
                gx = wrap_once(gx, g.nx);
            }
            gy = wrap_once(gy, g.ny);
            gz = wrap_once(gz, g.nz);
            ok = ok && inx;
            const int gi = ok ? (gx * g.ny + gy) * g.nz + gz : -1;
            rxs[u] = rx;
            rys[u] = ry;
            rzs[u] = rz;
            gis[u] = gi;
            if (ok) {
                d[u] = load3(disp, gi);
                wt[u] = load3(w3, gi);
            } else {
                d[u] = P3{0.f, 0.f, 0.f};
                wt[u] = P3{0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gis[u] < 0) continue;
            const int rx = rxs[u], ry = rys[u], rz = rzs[u];
            const float fx = floorf(d[u].x), fy = floorf(d[u].y), fz = floorf(d[u].z);
            const bool inl = fx >= (float)-H && fx <= (float)H && fy >= (float)-H && fy <= (float)H && fz >= (float)-H &&
                             fz <= (float)H;
            if (!inl) {
                const bool home = (unsigned)rx < (unsigned)B && (unsigned)ry < (unsigned)B && (unsigned)rz < (unsigned)B;
                if (home) {
                    int k = atomicAdd(ocount, 1);
                    outliers[k] = gis[u];
                }
                continue;
            }
            const int cx = rx + (int)fx, cy = ry + (int)fy, cz = rz + (int)fz;
            if (cx < -1 || cx >= B || cy < -1 || cy >= B || cz < -1 || cz >= B) continue;
            const float tx1 = d[u].x - fx, ty1 = d[u].y - fy, tz1 = d[u].z - fz;
            const float kx[2] = {1.f - tx1, tx1}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
            typedef float v2f __attribute__((ext_vector_type(2)));
            const float sx = wt[u].x * S, sy = wt[u].y * S, sz = wt[u].z * S;
            const float mw = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)) * (1.f / 2048.f);
            const v2f s01 = {sx, sy}, s2m = {sz, mw}, c01 = {0.f, 1.f};     // packed f32 math: two products per instruction
            // one LDS base address, corners at immediate offsets; a corner outside the tile is skipped
            const bool vx[2] = {cx >= 0, cx < B - 1}, vy[2] = {cy >= 0, cy < B - 1}, vz[2] = {cz >= 0, cz < B - 1};
            u64 *base = tile + ((cx * B + cy) * B + cz);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const float kxy = kx[a] * ky[bb];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (vx[a] && vy[bb] && vz[e]) {
                            const float k = kxy * kz[e];
                            const v2f kk = {k, k};
                            const v2f p01 = s01 * kk, p2m = __builtin_elementwise_fma(s2m, kk, c01);
                            const int i0 = cvt_rpi(p01.x), i1 = cvt_rpi(p01.y), i2 = cvt_rpi(p2m.x);
                            const unsigned ib = (unsigned)p2m.y;
                            const u64 wa = ((u64)(unsigned)(i1 + (i0 >> 31)) << 32) | (unsigned)i0;
                            const u64 wbv = ((u64)(ib + (unsigned)(i2 >> 31)) << 32) | (unsigned)i2;
                            u64 *q = base + ((a * B + bb) * B + e);
                            atomicAdd(q, wa);
                            atomicAdd(q + NT, wbv);
                        }
                }
        }
    }
    __syncthreads();

    // overflow proof: bound field of every cell
    int over = 0;
    for (int i = threadIdx.x; i < NT; i += THREADS) {
        const long long bw = (long long)tile[NT + i];
        const int c2 = (int)(unsigned)bw;
        over |= (unsigned)((bw - (long long)c2) >> 32) >= (1u << 19);
    }
    if (over) flagged = 1;
    __syncthreads();
    if (flagged) {
        if (threadIdx.x == 0) redo[1 + atomicAdd(redo, 1)] = (tx * (g.ny / B) + ty) * (g.nz / B) + tz;
        return;
    }
    for (int i = threadIdx.x; i < NT / 4; i += THREADS) {
        const int lz = (i % (B / 4)) * 4, r = i / (B / 4), ly = r % B, lx = r / B;
        float v[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long aw = (long long)tile[4 * i + q], bw = (long long)tile[NT + 4 * i + q];
            const int c0 = (int)(unsigned)aw, c2 = (int)(unsigned)bw;
            const int c1 = (int)((aw - (long long)c0) >> 32);
            v[0][q] = (float)c0 * Sinv;
            v[1][q] = (float)c1 * Sinv;
            v[2][q] = (float)c2 * Sinv;
        }
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            float4 o = make_float4(v[cc][0], v[cc][1], v[cc][2], v[cc][3]);
            float4 *dst = reinterpret_cast<float4 *>(mesh + cc * M + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz);
            if (accumulate) {
                const float4 old = *dst;
                o.x += old.x;
                o.y += old.y;
                o.z += old.z;
                o.w += old.w;
            }
            *dst = o;
        }
    }
}

__global__ __launch_bounds__(256) void paint3_outlier_kernel(Geom g, const float *__restrict__ disp,
                                                             const float *__restrict__ w3, float *__restrict__ mesh,
                                                             int64_t M, const int *__restrict__ outliers,
                                                             int *__restrict__ ocount) {
    const int count = ocount[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) ocount[1] = count;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const int gi = outliers[k];
        PIdx pi;
        pi.i = gi;
        pi.ipz = gi % g.nz;
        const int r = gi / g.nz;
        pi.ipy = r % g.ny;
        pi.ipx = r / g.ny;
        pi.valid = true;
        const P3 d = load3(disp, gi), wt = load3(w3, gi);
        int c[3];
        float f[3];
        locate<MCPM_POS_LATTICE, 2>(g, pi, d, c, f);
        if (g.xslab && (c[0] < 0 || c[0] > g.nx - 2)) atomicAdd(ocount + 2, 1);
        Stencil<2> s(g, c);
        const float kx[2] = {1.f - f[0], f[0]}, ky[2] = {1.f - f[1], f[1]}, kz[2] = {1.f - f[2], f[2]};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float k = kx[a] * ky[b] * kz[e];
                    float *m = mesh + s.xo[a] + s.yo[b] + s.zo[e];
                    atomicAdd(m, wt.x * k);
                    atomicAdd(m + M, wt.y * k);
                    atomicAdd(m + 2 * M, wt.z * k);
                }
    }
}

// outliers of the tiled paint: global atomics, grid-stride over the device-side count
__global__ __launch_bounds__(256) void paint_outlier_kernel(Geom g, const float *__restrict__ disp,
                                                            const float *__restrict__ w, int64_t wstride, float wscalar,
                                                            float *__restrict__ mesh, const int *__restrict__ outliers,
                                                            int *__restrict__ ocount) {
    const int count = ocount[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) ocount[1] = count;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const int gi = outliers[k];
        PIdx pi;
        pi.i = gi;
        pi.ipz = gi % g.nz;
        const int r = gi / g.nz;
        pi.ipy = r % g.ny;
        pi.ipx = r / g.ny;
        pi.valid = true;
        const P3 d = load3(disp, gi);
        int c[3];
        float f[3];
        locate<MCPM_POS_LATTICE, 2>(g, pi, d, c, f);
        // slab mode: a particle displaced beyond the ghost planes cannot be deposited on this rank; it is clamped
        // to the edge and counted (mcpm_plan_slab_oob) so that the host can widen the ghost region
        if (g.xslab && (c[0] < 0 || c[0] > g.nx - 2)) atomicAdd(ocount + 2, 1);
        const float wt = w ? w[(int64_t)gi * wstride] : wscalar;
        Stencil<2> s(g, c);
        const float kx[2] = {1.f - f[0], f[0]}, ky[2] = {1.f - f[1], f[1]}, kz[2] = {1.f - f[2], f[2]};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 2; ++e) atomicAdd(mesh + s.xo[a] + s.yo[b] + s.zo[e], wt * kx[a] * ky[b] * kz[e]);
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
template <int MODE, int ORDER>
__global__ __launch_bounds__(256) void read3_il_kernel(Geom g, const float *__restrict__ pos, int64_t n,
                                                       const float *__restrict__ fm, float *__restrict__ out) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    P3 d = load3(pos, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    Stencil<ORDER> s(g, c);
    float F[3], G[3][3];
    interp3<ORDER, false, true>(fm, 0, s, f, F, G);
    store3(out, pi.i, P3{F[0], F[1], F[2]});
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

template <int MODE, int ORDER, bool IL>
__global__ __launch_bounds__(256) void kick_drift_kernel(Geom g, const float *__restrict__ pos_in,
                                                         const float *__restrict__ vel_in, int64_t n,
                                                         const float *__restrict__ meshes, int64_t M, float alpha,
                                                         float beta, float dt, float *__restrict__ pos_out,
                                                         float *__restrict__ vel_out) {
    PIdx pi = particle_index<MODE>(g, n);
    if (!pi.valid) return;
    const P3 d = load3(pos_in, pi.i);
    const P3 v = load3(vel_in, pi.i);
    int c[3];
    float f[3];
    locate<MODE, ORDER>(g, pi, d, c, f);
    Stencil<ORDER> s(g, c);
    float F[3], G[3][3];
    interp3<ORDER, false, IL>(meshes, M, s, f, F, G);
    P3 v1 = {alpha * v.x + beta * F[0], alpha * v.y + beta * F[1], alpha * v.z + beta * F[2]};
    P3 d1 = {d.x + v1.x * dt, d.y + v1.y * dt, d.z + v1.z * dt};
    store3(vel_out, pi.i, v1);
    store3(pos_out, pi.i, d1);
}

// ------------------------------------------------------------------------------------------------
// launch helpers
static inline void lattice_launch(const Geom &g, dim3 &grid, dim3 &block) {
    int bs = g.pz >= 256 ? 256 : ((g.pz + 63) / 64) * 64;
    int cpr = (g.pz + bs - 1) / bs;
    block = dim3(bs);
    grid = dim3((unsigned)((int64_t)g.px * g.py * cpr));
}
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

template <int BX, int BY, int BZ, int H, int THREADS, int U>
static void launch_tile(mcpm_plan *p, const float *pos, const float *w, int64_t wstride, float wscalar, float *mesh,
                        int accumulate) {
    const Geom &g = p->g;
    unsigned nb = (unsigned)((g.nx / BX) * (g.ny / BY) * (g.nz / BZ));
    if (w)
        paint_tile_kernel<BX, BY, BZ, H, true, THREADS, U><<<nb, THREADS, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate,
                                                                                          p->outliers, p->outlier_count);
    else
        paint_tile_kernel<BX, BY, BZ, H, false, THREADS, U><<<nb, THREADS, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate,
                                                                                           p->outliers, p->outlier_count);
}

template <int BZ, int H>
static void launch_tile_variant(mcpm_plan *p, const float *pos, const float *w, int64_t wstride, float wscalar, float *mesh,
                                int accumulate) {
    if (BZ == 64) {
        switch (p->paint_variant) {
            case 1: launch_tile<16, 16, 32, H, 512, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 2: launch_tile<16, 16, 32, H, 512, 2>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 3: launch_tile<16, 16, 32, H, 256, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 4: launch_tile<16, 16, 64, H, 1024, 2>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 5: launch_tile<16, 16, 64, H, 1024, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 6: launch_tile<16, 16, 64, H, 512, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 7: launch_tile<16, 16, 16, H, 256, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 8: launch_tile<16, 16, 32, H, 512, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 9: launch_tile<16, 16, 16, H, 512, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            default:
                // 16^3 tiles (32 KB of f64, several workgroups per CU): the kernel is bound by the LDS f64 atomic rate,
                // not by the 3.8x halo re-reads, so they tie the 16x16x64 tile at 512^3 and win on small meshes
                // (64^3: 0.030 vs 0.051 ms), where few large tiles cannot fill the chip
                launch_tile<16, 16, 16, H, 512, 4>(p, pos, w, wstride, wscalar, mesh, accumulate);
                break;
        }
    } else {
        launch_tile<16, 16, 16, H, 256, 4>(p, pos, w, wstride, wscalar, mesh, accumulate);
    }
}

// Tiled paint if the geometry allows; returns false if the caller must use the generic path.
static bool try_paint_tiled(mcpm_plan *p, const float *pos, const float *w, int64_t wstride, float wscalar, float *mesh,
                            int accumulate) {
    const Geom &g = p->g;
    if (!g.same_lattice) return false;
    if (g.nx % 16 || g.ny % 16 || g.nz % 16) return false;
    const int H = p->halo;
    if (g.nx < H + 1 || g.ny < H + 1 || g.nz < H + 1) return false;
    if (((uintptr_t)mesh) & 15) return false;
    (void)hipMemsetAsync(p->outlier_count, 0, sizeof(int), p->stream);
    const bool z64 = (g.nz % 64 == 0);
    if (z64) {
        switch (H) {
            case 1: launch_tile_variant<64, 1>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 2: launch_tile_variant<64, 2>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 3: launch_tile_variant<64, 3>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 4: launch_tile_variant<64, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            default: launch_tile_variant<64, 6>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
        }
    } else {
        switch (H) {
            case 1: launch_tile_variant<16, 1>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 2: launch_tile_variant<16, 2>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 3: launch_tile_variant<16, 3>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            case 4: launch_tile_variant<16, 4>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
            default: launch_tile_variant<16, 6>(p, pos, w, wstride, wscalar, mesh, accumulate); break;
        }
    }
    paint_outlier_kernel<<<256, 256, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, p->outliers, p->outlier_count);
    return true;
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
    if (mode == MCPM_POS_LATTICE && order == 2 && n > 0 && try_paint_tiled(p, pos, weights, wstride, wscalar, mesh, accumulate)) {
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
    const Geom &g = p->g;
    const int H = p->halo;
    const bool tiled = mode == MCPM_POS_LATTICE && order == 2 && n > 0 && g.same_lattice && !(g.nx % 16) && !(g.ny % 16) &&
                       !(g.nz % 16) && g.nx >= H + 1 && g.ny >= H + 1 && g.nz >= H + 1 && !(((uintptr_t)meshes3) & 15) &&
                       !((p->M * 4) & 15) && p->paint3_variant >= 0;
    if (!tiled) {
        for (int c = 0; c < 3; ++c) MCPM_TRY(mcpm_paint_f32(p, pos, n, mode, weights3 + c, 3, 0.f, order, meshes3 + c * p->M, accumulate));
        return MCPM_OK;
    }
    StageTimer st_(p, ST_PAINT3, 24.0 * n + (accumulate ? 24.0 : 12.0) * p->M);
    (void)hipMemsetAsync(p->outlier_count, 0, sizeof(int), p->stream);
    const unsigned nb = (unsigned)((g.nx / 16) * (g.ny / 16) * (g.nz / 16));
    if (p->paint3_variant == 4) {   // fixed-point tiles; the tiles they flag (and every tile if max|w| is unusable) in f64
        MCPM_REQUIRE(p, p->fx_tiles >= (int)nb, MCPM_E_ARG, "mcpm_paint3_f32: redo list too small");
        (void)hipMemsetAsync(p->fx_redo, 0, sizeof(int), p->stream);
        if (p->fx_src != weights3) {   // max|w| not left behind by the kernel that produced the weights
            (void)hipMemsetAsync(p->fx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream);
            absmax_kernel<<<2048, 256, 0, p->stream>>>(weights3, 3 * n, p->fx_wmax);
        }
        p->fx_src = nullptr;
#define CALLFX(HH)                                                                                                            \
    {                                                                                                                         \
        paint3_fx_kernel<16, HH, 512, 4><<<nb, 512, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, p->outliers, p->outlier_count, p->fx_wmax, p->fx_redo); \
        paint3_tile_kernel<16, HH, 1024, 4><<<nb, 1024, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, p->outliers, p->outlier_count, p->fx_redo); \
    }
        switch (H) {
            case 1: CALLFX(1) break;
            case 2: CALLFX(2) break;
            case 3: CALLFX(3) break;
            case 4: CALLFX(4) break;
            default: CALLFX(6) break;
        }
#undef CALLFX
        paint3_outlier_kernel<<<256, 256, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, p->outliers, p->outlier_count);
        MCPM_LAUNCH_CHECK(p, "paint3_fx_kernel");
        return MCPM_OK;
    }
#define CALL3(HH)                                                                                                             \
    {                                                                                                                         \
        if (p->paint3_variant == 1)                                                                                           \
            paint3_tile_kernel<16, HH, 512, 4><<<nb, 512, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, p->outliers, p->outlier_count); \
        else if (p->paint3_variant == 2)                                                                                      \
            paint3_tile_kernel<16, HH, 1024, 4><<<nb, 1024, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, p->outliers, p->outlier_count); \
        else                                                                                                                  \
            paint3_tile_kernel<16, HH, 1024, 2><<<nb, 1024, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, p->outliers, p->outlier_count); \
    }
    switch (H) {
        case 1: CALL3(1) break;
        case 2: CALL3(2) break;
        case 3: CALL3(3) break;
        case 4: CALL3(4) break;
        default: CALL3(6) break;
    }
#undef CALL3
    paint3_outlier_kernel<<<256, 256, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, p->outliers, p->outlier_count);
    MCPM_LAUNCH_CHECK(p, "paint3_tile_kernel");
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
#define CALL(MO, OR) read3_il_kernel<MO, OR><<<grid, block, 0, p->stream>>>(p->g, pos, n, fm_il, out)
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
    dim3 grid, block;
    if (mode == MCPM_POS_LATTICE) lattice_launch(p->g, grid, block); else flat_launch(n, grid, block);
#define CALL(MO, OR)                                                                                                             \
    if (layout) kick_drift_kernel<MO, OR, true><<<grid, block, 0, p->stream>>>(p->g, pos_in, vel_in, n, meshes3, p->M, alpha, beta, dt, pos_out, vel_out); \
    else kick_drift_kernel<MO, OR, false><<<grid, block, 0, p->stream>>>(p->g, pos_in, vel_in, n, meshes3, p->M, alpha, beta, dt, pos_out, vel_out)
    DISPATCH_MODE_ORDER(mode, order, CALL);
#undef CALL
    MCPM_LAUNCH_CHECK(p, "kick_drift_kernel");
    return MCPM_OK;
}

