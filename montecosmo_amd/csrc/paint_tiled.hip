// LDS-tiled paints of libmcpm.so for gfx950 (CIC, lattice displacements, particle lattice == mesh):
//   mcpm_paint_tiled   the density paint of the force cycle            (montecosmo/nbody.py:365-396 `paint`)
//   mcpm_paint3_tiled  three weighted paints at once, the adjoint of the three-component force read (`read` :398-427, x3)
//
// Design (MI355X-first; the reference runs 8 scatter-add passes with global atomics):
//   * PULL, not scatter.  One workgroup owns one 16^3 Eulerian tile of the mesh in LDS and pulls the lattice particles that
//     can land in it: coalesced 12-byte loads, lanes along z, 4 loads in flight per thread.  Only stencil points inside the
//     tile are deposited (LDS integer atomics); the tile is written with plain 16-byte stores.  No global atomics, no sort.
//   * WINDOWS THAT FOLLOW THE DISPLACEMENT FIELD.  Particles are stored in Lagrangian order as displacements from their lattice
//     point, and the displacement field is smooth: a particle with floor(d) = fd that lands in tile T sits at lattice point
//     T + c - fd (c in [-1, 15] its base cell in T), so T needs the lattice points [T - 1 - max fd, T + 15 - min fd] per axis over
//     the particles that land in it.  `tile_prologue_kernel` samples 64 particles per 16^3 Lagrangian block (25 MB of reads at
//     512^3) and `box_tile_kernel` gives every tile the box [lo, hi] = hull of the sampled floor(d) of the 27 blocks around it:
//     a window of prod_a (17 + hi_a - lo_a) lattice points, 1.4-1.9 window visits per particle on the benchmark's trajectory
//     (rms displacement 2 cells) where one symmetric halo H = 3 around the bulk offset, (16 + 2H + 1)^3 points for every tile,
//     cost 3.0 and an uncentred H = 4 window 3.8.  A fixed halo (mcpm_plan_set_halo; meshes below 2048 tiles) is the box
//     [o_T - H, o_T + H] around the block's rounded mean displacement o_T.
//   * BUCKETS for what the windows miss.  Every workgroup also watches the particles of its own Lagrangian block (they lie
//     in its window whenever |o_T| <= H; a second short loop covers them otherwise): a particle whose floor(d) leaves the
//     interval in which every neighbouring window is sure to contain it (six compares, the cost of the round-1 outlier
//     test) is appended to a SUSPECT list.  `coverage_duty_kernel` then runs the exact test on the suspects only: for each
//     of the (up to 8) tiles T' the stencil touches, with the same integer predicate the consumer uses, is the lattice
//     point in T's window?  If not the particle goes to T's bucket (a fixed-capacity list per tile).  A second kernel deposits every bucket through an LDS integer tile and adds it to the mesh with plain
//     read-modify-writes of the touched cells (one workgroup per tile: no races).  A bucket that overflows (displacement
//     fields with no bulk flow to follow: nothing a PM run produces) is dropped as a whole and a repair pass over the
//     particles deposits that tile's pairs with f32 global atomics: no input can lose mass.
//   * INTEGER ACCUMULATORS everywhere: 2^-30 fixed point (unweighted), max|w| 2^-28 fixed point (weighted; max|w| from a
//     reduction pass) in 64-bit LDS integer atomics; the three-component kernel packs 32-bit fields with an overflow-proof
//     bound (below).  Integer sums do not depend on arrival order, so every paint is bitwise reproducible.  Only particles
//     with non-finite / absurd displacements ("wild") and bucket overflows take f32 global atomics; both are counted
//     (mcpm_plan_last_outliers).
#include "particles_dev.h"

typedef unsigned long long u64;

#define MCPM_TILE 16
#define MCPM_TAME 16384.f   // |d| beyond this (or NaN) on any axis: "wild", deposited by the global-atomic kernel

// counters (device ints): [0] wild particles, [1] copy of last wild + overflow pairs (host query), [2] slab deposits beyond
// the ghost planes (cumulative), [3] appends that found their bucket full, [4] tiles with a non-empty bucket, [5] bucketed pairs, [6] suspects (particles handed to the exact coverage test)
enum { C_WILD = 0, C_LAST = 1, C_OOB = 2, C_PAIRS = 3, C_NTILES = 4, C_BUCKETED = 5, C_SUSPECTS = 6 };

__device__ __forceinline__ int pack_off(int ox, int oy, int oz) { return (ox & 0xff) | ((oy & 0xff) << 8) | ((oz & 0xff) << 16); }
__device__ __forceinline__ void unpack_off(int v, int &ox, int &oy, int &oz) {
    ox = (int)(signed char)(v & 0xff);
    oy = (int)(signed char)((v >> 8) & 0xff);
    oz = (int)(signed char)((v >> 16) & 0xff);
}
__device__ __forceinline__ int pymod(int a, int n) {
    int r = a % n;
    return r < 0 ? r + n : r;
}
__device__ __forceinline__ int wrap_once(int c, int n) {   // c in (-n, 2n) -> [0, n)
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

// Tile of this workgroup.  Blocks b, b+8, ... run on the same XCD (and share its 4 MB L2): each XCD works through one
// contiguous run of tiles -- a slab of ntx / 8 tile planes -- in pencils along z (flushes and particle reads contiguous in
// memory; compact bricks of tiles per XCD measured 4-15 % slower).  ORDER of the pencils within the slab (MCPM_TILE_ORDER):
// 0 = y fastest, then x: the ~128 tiles an XCD has in flight are four y-neighbouring pencils of ONE x plane, and the x halo
// of every window (7 of 23 planes) is fetched again from HBM when the next plane's turn comes, 1024 tiles later;
// 1 (default) = x fastest over the slab's planes, then y: the pencils in flight are x neighbours and share that halo in L2.
// Counters at 512^3 (profiles/r03_tile_order.txt): the three-component paint fetches 5.27 instead of 6.75 GB per launch, the
// density paint 2.97 / 2.89 GB (no change); neither kernel's time moves (they are bound by their LDS atomics).  Bricks of
// 4 x 4 x 8 tiles in flight fetch MORE (7.44 / 3.43 GB).
__device__ __forceinline__ void tile_of_block(int ntx, int nty, int ntz, int &tx, int &ty, int &tz, int order = 0) {
    const int nb = gridDim.x, b = blockIdx.x;
    const bool x8 = nb % 8 == 0;
    const int t = x8 ? (b % 8) * (nb / 8) + b / 8 : b;
    if (order == 1 && x8 && ntx % 8 == 0 && nb == ntx * nty * ntz) {
        const int nxp = ntx / 8, k = b % 8, l = b / 8;       // planes per XCD slab, XCD, index within the run
        tz = l % ntz;
        const int r = l / ntz;
        tx = k * nxp + r % nxp;
        ty = r / nxp;
        return;
    }
    tz = t % ntz;
    const int tt = t / ntz;
    ty = tt % nty;
    tx = tt / nty;
}

// ------------------------------------------------------------------------------------------------
// Prologue of every tiled paint, one launch: resets the bucket counts and the per-paint counters, and (toff != NULL) sets
// o_T = rounded mean displacement of 64 lattice points (four z rows) of the Lagrangian block at tile T.
__global__ __launch_bounds__(256) void tile_prologue_kernel(Geom g, const float *__restrict__ disp, int *__restrict__ toff,
                                                            int *__restrict__ thi, int *__restrict__ bcnt, int *__restrict__ cnts, int ntiles,
                                                            int maxoff, int *__restrict__ redo, int *__restrict__ rng, int hfix) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x < 8 && threadIdx.x != C_LAST && threadIdx.x != C_OOB) cnts[threadIdx.x] = 0;
    if (redo && blockIdx.x == 0 && threadIdx.x == 8) redo[0] = 0;   // empty list of tiles for the f64 repaint (paint3)
    if (tile >= ntiles) return;
    if (lane == 0) bcnt[tile] = 0;
    if (!toff) return;
    const int ntz = g.nz / MCPM_TILE, nty = g.ny / MCPM_TILE;
    const int tz = tile % ntz, ty = (tile / ntz) % nty, tx = tile / (ntz * nty);
    // four whole z rows of the tile ((x, y) = (4, 4), (4, 12), (12, 4), (12, 12); 16 consecutive particles = 192 bytes each):
    // 8 cache lines per tile instead of the 32 a 4 x 4 x 4 sub-grid touches (the kernel is bound by the lines it pulls)
    int gx = tx * MCPM_TILE + 4 + 8 * (lane >> 5) - g.xoff;
    const int gy = ty * MCPM_TILE + 4 + 8 * ((lane >> 4) & 1), gz = tz * MCPM_TILE + (lane & 15);
    if (g.xslab) gx = min(max(gx, 0), g.px - 1);   // ghost tiles: the nearest lattice plane
    const P3 d = load3(disp, ((int64_t)gx * g.ny + gy) * g.nz + gz);
    float sx = d.x, sy = d.y, sz = d.z;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        sz += __shfl_xor(sz, o);
    }
    if (lane == 0) {
        const float m = (float)maxoff;   // fmaxf / fminf return the non-NaN operand: a NaN mean gives a finite offset
        const int o0 = (int)rintf(fminf(fmaxf(sx * (1.f / 64.f), -m), m)), o1 = (int)rintf(fminf(fmaxf(sy * (1.f / 64.f), -m), m)),
                  o2 = (int)rintf(fminf(fmaxf(sz * (1.f / 64.f), -m), m));
        if (rng) toff[tile] = pack_off(o0, o1, o2);      // box_tile_kernel turns it into the window's lower corner
        else {      // a fixed halo: the window [o - H, o + H] per axis
            toff[tile] = pack_off(o0 - hfix, o1 - hfix, o2 - hfix);
            thi[tile] = pack_off(o0 + hfix, o1 + hfix, o2 + hfix);
        }
    }
    if (!rng) return;
    // range of floor(d) over the same 64 samples, per axis (clamped to +-100; a NaN sample widens it to the clamp): what
    // box_tile_kernel sizes the windows of the tiles around this block with
    const float f3[3] = {floorf(d.x), floorf(d.y), floorf(d.z)};
    float lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bool ok = f3[c] == f3[c];
        const float v = fminf(fmaxf(f3[c], -100.f), 100.f);
        lo[c] = ok ? v : -100.f;
        hi[c] = ok ? v : 100.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
        }
    if (lane == 0) {
        rng[tile] = pack_off((int)lo[0], (int)lo[1], (int)lo[2]);
        rng[ntiles + tile] = pack_off((int)hi[0], (int)hi[1], (int)hi[2]);
    }
}

// ------------------------------------------------------------------------------------------------
// The WINDOW of every tile, chosen on the device for every input (DESIGN findings 31, 43, 46).  A particle with floor(d) = fd that lands in
// tile T sits at the lattice point T + c - fd (c in [-1, 15] its base cell within T), so the lattice points T needs are, per axis,
// [T - 1 - max fd, T + 15 - min fd] over the particles that land in it -- and those come from the 27 Lagrangian blocks around it.  The window
// is that box for [lo_a, hi_a] = the hull of the SAMPLED floor(d) ranges of those blocks (tile_prologue_kernel: 64 samples each):
// 17 + hi_a - lo_a points along axis a.  Round 3 chose ONE symmetric halo H per input ((16 + 2H + 1)^3 points for every tile) and the first
// version of round 4 one H per tile around the block's mean offset o_T (H_T = max |fd - o_T|): a box per axis is 17-26 % smaller still
// (`tools/window_extents.py`: 1.86 instead of 2.46 window visits per particle on the evolved 512^3 state), because the ranges are neither
// centred on the mean nor equally wide along the three axes.  Extents above 8 (a symmetric halo of 4) are cut back around o_T; what the
// samples or the cut miss goes the way of everything a window misses (suspects -> exact test -> buckets).  Same input, same windows:
// results stay bitwise reproducible, and the host is not involved.  lo rides in the tile's packed offset word, hi in `thi`.
__global__ __launch_bounds__(256) void box_tile_kernel(Geom g, int *__restrict__ toff, int *__restrict__ thi, const int *__restrict__ rng,
                                                       int ntiles, int maxabs) {
    const int tile = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
    if (tile >= ntiles) return;      // (whole 16-lane groups leave together; the shuffles below stay within a group)
    const int ntz = g.nz / MCPM_TILE, nty = g.ny / MCPM_TILE, ntx = g.nx / MCPM_TILE;
    const int tz = tile % ntz, ty = (tile / ntz) % nty, tx = tile / (ntz * nty);
    int lo[3] = {127, 127, 127}, hi[3] = {-127, -127, -127};
#pragma unroll
    for (int h = 0; h < 2; ++h) {      // 27 neighbouring blocks over the tile's 16 lanes
        const int q = l16 + 16 * h;
        if (q < 27) {
            const int a = q / 9 - 1, b = (q / 3) % 3 - 1, e = q % 3 - 1;
            const int Tx = g.xslab ? min(max(tx + a, 0), ntx - 1) : pymod(tx + a, ntx);
            const int nb = (Tx * nty + pymod(ty + b, nty)) * ntz + pymod(tz + e, ntz);
            int l3[3], h3[3];
            unpack_off(rng[nb], l3[0], l3[1], l3[2]);
            unpack_off(rng[ntiles + nb], h3[0], h3[1], h3[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) lo[c] = min(lo[c], l3[c]), hi[c] = max(hi[c], h3[c]);
        }
    }
#pragma unroll
    for (int m = 8; m > 0; m >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = min(lo[c], __shfl_xor(lo[c], m));
            hi[c] = max(hi[c], __shfl_xor(hi[c], m));
        }
    if (l16 == 0) {
        int o[3];
        unpack_off(toff[tile], o[0], o[1], o[2]);      // the block's mean offset (tile_prologue_kernel)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (hi[c] - lo[c] > 8) lo[c] = max(lo[c], o[c] - 4), hi[c] = min(hi[c], o[c] + 4);
            lo[c] = min(max(lo[c], -maxabs), maxabs);
            hi[c] = max(min(max(hi[c], -maxabs), maxabs), lo[c]);
        }
        toff[tile] = pack_off(lo[0], lo[1], lo[2]);
        thi[tile] = pack_off(hi[0], hi[1], hi[2]);
    }
}

// ------------------------------------------------------------------------------------------------
// scale of the weighted fixed-point accumulators: S = 2^(28 - e), 2^e <= max|w| < 2^(e+1): one deposit is below 2^29, an
// int64 cell holds 2^34 of them (n < 2^31 particles x 8 corners).  mode: 0 = all weights zero, 1 = fixed point, 2 = non-finite
struct TScale {
    float S;
    double Sinv;
    int mode;
};
__device__ __forceinline__ TScale tile_scale(const unsigned *__restrict__ wmax_bits) {
    unsigned wb = wmax_bits[(threadIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wb = max(wb, (unsigned)__shfl_xor((int)wb, o));
    TScale r;
    int be = (int)(wb >> 23);
    r.mode = wb == 0u ? 0 : (be >= 255 ? 2 : 1);
    be = min(max(be, 30), 254);            // tiny maxima (< 2^-97): the scale is capped, absolute resolution 2^-125
    const int e = be - 127;
    r.S = __uint_as_float((unsigned)(127 + 28 - e) << 23);
    r.Sinv = __longlong_as_double((long long)(1023 - 28 + e) << 52);
    return r;
}

// ------------------------------------------------------------------------------------------------
// lists shared by the kernels below
struct TileLists {
    const int *toff;   // per tile: the lower corner (lo_x, lo_y, lo_z) of its window's floor(d) box, packed (NULL: every window is +-hfix)
    const int *thi;    // ... and the upper corner
    int *bcnt;         // bucket fill counts per tile (zeroed by the host before the paint)
    int *bucket;       // [tile][cap] particle indices
    int cap;
    int *nonempty;     // tiles whose bucket received something (cnts[C_NTILES] of them): what the bucket kernels walk
    int *list;         // suspects, from the back (capacity Np: at most one entry per particle)
    int *wild;         // wild particles (own Np ints: every wild particle found by the coverage kernel is also a suspect, so a
                       // shared list could be overrun while other threads still read suspects from it -- ADVICE r2)
    int listcap;
    int *cnts;
    int order;         // order of the tile pencils within an XCD's slab (tile_of_block)
    int hfix;          // toff == NULL (uncentred windows): the box is [-hfix, hfix] per axis for every tile
};

// The window of a tile as the box of floor(d) values it serves: a particle with floor(d) = fd landing in the tile lies in the window iff
// lo_a <= fd_a <= hi_a on every axis; the window's lattice points relative to the tile origin are r_a in [-1 - hi_a, 15 - lo_a].
struct Box {
    int lo[3], hi[3];
};
__device__ __forceinline__ Box tile_box(const TileLists &L, int tile_index) {
    Box b;
    if (L.toff) {
        unpack_off(L.toff[tile_index], b.lo[0], b.lo[1], b.lo[2]);
        unpack_off(L.thi[tile_index], b.hi[0], b.hi[1], b.hi[2]);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) b.lo[c] = -L.hfix, b.hi[c] = L.hfix;
    }
    return b;
}
// the same for the tile of a workgroup's window walk: wave-uniform, in scalar registers
__device__ __forceinline__ Box block_box(const TileLists &L, int tile_index) {
    Box b;
    if (L.toff) {
        unpack_off(__builtin_amdgcn_readfirstlane(L.toff[tile_index]), b.lo[0], b.lo[1], b.lo[2]);
        unpack_off(__builtin_amdgcn_readfirstlane(L.thi[tile_index]), b.hi[0], b.hi[1], b.hi[2]);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) b.lo[c] = -L.hfix, b.hi[c] = L.hfix;
    }
    return b;
}

__device__ __forceinline__ void append_wild(const TileLists &L, int gi) {
    const int k = atomicAdd(L.cnts + C_WILD, 1);
    if (k < L.listcap) L.wild[k] = gi;      // at most one entry per particle: cannot overflow
}

// Suspects are staged in LDS (one global atomic per workgroup instead of one per suspect: a single global counter serialised
// the kernel as soon as a few per cent of the particles were suspects); `sus`: MCPM_SUS ints + the count in sus[MCPM_SUS].
#define MCPM_SUS 960      // (with the z-padded density tile, 36 KB, four workgroups still fit the 160 KB of a CU)
__device__ __forceinline__ void append_suspect(const TileLists &L, int *sus, int gi) {
    const int k = atomicAdd(sus + MCPM_SUS, 1);
    if (k < MCPM_SUS) sus[k] = gi;
    else {
        const int kg = atomicAdd(L.cnts + C_SUSPECTS, 1);
        if (kg < L.listcap) L.list[L.listcap - 1 - kg] = gi;
    }
}
// after a barrier: moves the staged suspects to the global list (sus[MCPM_SUS + 1]: scratch word for the base index)
__device__ __forceinline__ void flush_suspects(const TileLists &L, int *sus) {
    const int n = min(sus[MCPM_SUS], MCPM_SUS);
    if (n == 0) return;
    if (threadIdx.x == 0) sus[MCPM_SUS + 1] = atomicAdd(L.cnts + C_SUSPECTS, n);
    __syncthreads();
    const int base = sus[MCPM_SUS + 1];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (base + i < L.listcap) L.list[L.listcap - 1 - (base + i)] = sus[i];
}

// The tiles the CIC stencil of one particle touches whose window does NOT contain its lattice point.  (tx, ty, tz): home
// tile of the lattice point, (r): lattice point relative to that tile's origin, (i): floor of the displacement.
// f(tile index, base cell relative to that tile's origin) is called for each such tile.
template <class F>
__device__ __forceinline__ void for_uncovered(const Geom &g, const TileLists &L, int tx, int ty, int tz, int ntx, int nty, int ntz,
                                              const Box &self, int rx, int ry, int rz, int ix, int iy, int iz, F f) {
    const int cx = rx + ix, cy = ry + iy, cz = rz + iz;
    const int d0x = cx >> 4, d0y = cy >> 4, d0z = cz >> 4, mx = cx & 15, my = cy & 15, mz = cz & 15;
    const int nax = mx == 15 ? 2 : 1, nay = my == 15 ? 2 : 1, naz = mz == 15 ? 2 : 1;
    for (int a = 0; a < nax; ++a) {
        const int dtx = d0x + a, rlx = a ? -1 - ix : mx - ix;
        int Tx = tx + dtx;
        if (g.xslab) {
            if (Tx < 0 || Tx >= ntx) continue;   // beyond the ghost planes: such particles are "wild" (see the callers)
        } else
            Tx = pymod(Tx, ntx);
        for (int b = 0; b < nay; ++b) {
            const int dty = d0y + b, rly = b ? -1 - iy : my - iy;
            const int Ty = pymod(ty + dty, nty);
            for (int e = 0; e < naz; ++e) {
                const int dtz = d0z + e, rlz = e ? -1 - iz : mz - iz;
                const int Tz = pymod(tz + dtz, ntz);
                const int tidx = (Tx * nty + Ty) * ntz + Tz;
                const Box bx = (dtx | dty | dtz) == 0 || !L.toff ? self : tile_box(L, tidx);      // the tile that would have to pull this particle
                const bool covered = (unsigned)(rlx + 1 + bx.hi[0]) < (unsigned)(MCPM_TILE + 1 + bx.hi[0] - bx.lo[0]) &&
                                     (unsigned)(rly + 1 + bx.hi[1]) < (unsigned)(MCPM_TILE + 1 + bx.hi[1] - bx.lo[1]) &&
                                     (unsigned)(rlz + 1 + bx.hi[2]) < (unsigned)(MCPM_TILE + 1 + bx.hi[2] - bx.lo[2]);
                if (!covered) f(tidx, a ? -1 : mx, b ? -1 : my, e ? -1 : mz);
            }
        }
    }
}

// particles the tiled kernels leave to the global-atomic kernel: non-finite / absurd displacements, and (slab mode) base
// cells beyond the ghost planes (clamped and counted there)
__device__ __forceinline__ bool is_wild(const Geom &g, const P3 &d, int x0, int rx) {
    const bool tame = fabsf(d.x) < MCPM_TAME && fabsf(d.y) < MCPM_TAME && fabsf(d.z) < MCPM_TAME;   // NaN compares false
    if (!tame) return true;
    if (g.xslab) {
        const int cxg = x0 + rx + (int)floorf(d.x);
        return cxg < 0 || cxg > g.nx - 2;
    }
    return false;
}

// Exact coverage test of the suspects: append each to the bucket of every tile whose window misses it.  A full bucket keeps
// counting (bcnt > cap marks the tile for the repair pass, which then deposits ALL of that tile's pairs).
__device__ __forceinline__ void coverage_duty_body(const Geom &g, const float *__restrict__ disp, const TileLists &L) {
    const int ns = min(L.cnts[C_SUSPECTS], L.listcap);
    const int ntx = g.nx / MCPM_TILE, nty = g.ny / MCPM_TILE, ntz = g.nz / MCPM_TILE;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ns; k += gridDim.x * blockDim.x) {
        const int gi = L.list[L.listcap - 1 - k];
        const int qz = gi % g.nz, r = gi / g.nz, qy = r % g.ny, qx = r / g.ny + g.xoff;   // lattice == mesh
        const int tx = qx >> 4, ty = qy >> 4, tz = qz >> 4;
        const P3 d = load3(disp, gi);
        if (is_wild(g, d, tx * MCPM_TILE, qx & 15)) {
            append_wild(L, gi);
            continue;
        }
        for_uncovered(g, L, tx, ty, tz, ntx, nty, ntz, tile_box(L, (tx * nty + ty) * ntz + tz), qx & 15, qy & 15, qz & 15,
                      (int)floorf(d.x), (int)floorf(d.y), (int)floorf(d.z), [&](int tidx, int, int, int) {
                             const int kb = atomicAdd(L.bcnt + tidx, 1);
                             if (kb == 0) L.nonempty[atomicAdd(L.cnts + C_NTILES, 1)] = tidx;
                             if (kb < L.cap) L.bucket[(int64_t)tidx * L.cap + kb] = gi;
                             else atomicAdd(L.cnts + C_PAIRS, 1);
                         });
    }
}

__global__ __launch_bounds__(256) void coverage_duty_kernel(Geom g, const float *__restrict__ disp, TileLists L) { coverage_duty_body(g, disp, L); }

// Conservative form of the coverage test, as cheap as the round-1 outlier test: a particle of the home block is covered by EVERY tile
// its stencil can touch if lo_T',a <= floor(d)_a <= hi_T',a on each axis for each of the 27 tiles T' around the block, i.e. if floor(d)_a
// lies in the INTERSECTION of their boxes.  (With per-tile windows that intersection contains the block's own sampled range by
// construction -- every T' sized its window with it -- whereas the symmetric interval of round 3, o_a +- (H - max |o_T',a - o_a|), has
// no slack left when each window is just large enough.)  slo / shi: the interval as floats, per axis
// (empty when slo > shi: every home particle is then a suspect).  Reduced by the first wave, broadcast through LDS.
__device__ __forceinline__ void sure_intervals(const Geom &g, const TileLists &L, int tx, int ty, int tz, int ntx, int nty, int ntz,
                                               int *sh27, float (&slo)[3], float (&shi)[3]) {
    if (threadIdx.x < 64) {
        int lo[3] = {-127, -127, -127}, hi[3] = {127, 127, 127};
        if (threadIdx.x < 27) {
            const int a = (int)threadIdx.x / 9 - 1, b = ((int)threadIdx.x / 3) % 3 - 1, e = (int)threadIdx.x % 3 - 1;
            const int Tx = g.xslab ? min(max(tx + a, 0), ntx - 1) : pymod(tx + a, ntx);
            const Box nb = tile_box(L, (Tx * nty + pymod(ty + b, nty)) * ntz + pymod(tz + e, ntz));
#pragma unroll
            for (int c = 0; c < 3; ++c) lo[c] = nb.lo[c], hi[c] = nb.hi[c];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                lo[c] = max(lo[c], __shfl_xor(lo[c], o));
                hi[c] = min(hi[c], __shfl_xor(hi[c], o));
            }
        if (threadIdx.x == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) sh27[c] = lo[c], sh27[3 + c] = hi[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        slo[c] = (float)sh27[c];
        shi[c] = (float)sh27[3 + c];
    }
}

// Window points of one thread, visited in the order j = tid, tid + THREADS, tid + 2 THREADS, ... of the flat index (z fastest) of a
// window of Wx x Wy x Wz points (run-time, wave-uniform: the tile's box): (jx, jy, jz) are carried along with add-and-carry (7
// instructions) instead of two divisions per visit, and -- FAST: power-of-two periodic meshes, every single-GPU bench configuration --
// the periodic wrap is one AND per axis.  (The generic walk of round 2 spent ~45 vector and ~10 scalar-branch instructions per visit on
// the run-time choice between the slab, power-of-two and general wraps: a quarter of the tile kernels' instruction stream.)
// The few divisions of the set-up (thread index and THREADS by the window widths) are exact in float: numerators below 1024 + 1/2,
// divisors below 1024, so the quotient's distance to the next integer (>= 1 / 2d) dwarfs the rounding of v_rcp_f32.
__device__ __forceinline__ int small_div(int n, int d) { return (int)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d)); }
template <int THREADS>
struct WinIter {
    int jx, jy, jz, Wx, Wy, Wz, DX, DY, DZ;
    __device__ __forceinline__ WinIter(int tid, int wx, int wy, int wz) : Wx(wx), Wy(wy), Wz(wz) {
        const int r = small_div(tid, wz);
        jz = tid - r * wz;
        jx = small_div(r, wy);
        jy = r - jx * wy;
        const int wyz = wy * wz;
        DX = small_div(THREADS, wyz);
        const int rem = THREADS - DX * wyz;
        DY = small_div(rem, wz);
        DZ = rem - DY * wz;
    }
    __device__ __forceinline__ bool valid() const { return jx < Wx; }
    __device__ __forceinline__ void next() {
        jz += DZ;
        const int cz = jz >= Wz ? 1 : 0;
        jz -= cz ? Wz : 0;
        jy += DY + cz;
        const int cy = jy >= Wy ? 1 : 0;
        jy -= cy ? Wy : 0;
        jx += DX + cy;
    }
    // lattice point relative to the tile (r) and flat lattice index of the current point; power-of-two periodic mesh: the index is
    // assembled with shifts (two v_lshl_or instead of two quarter-rate v_mad_u64_u32; the kernels are bound by VALU issue)
    __device__ __forceinline__ int point_fast(const Geom &g, int x0, int y0, int z0, const Box &b, int &rx, int &ry, int &rz) const {
        rx = jx - 1 - b.hi[0];
        ry = jy - 1 - b.hi[1];
        rz = jz - 1 - b.hi[2];
        const int gx = (x0 + rx) & (g.nx - 1), gy = (y0 + ry) & (g.ny - 1), gz = (z0 + rz) & (g.nz - 1);
        const int lz = __builtin_ctz((unsigned)g.nz), ly = __builtin_ctz((unsigned)g.ny);      // uniform: scalar registers
        return (((gx << ly) | gy) << lz) | gz;
    }
    // the same on a slab plan (FAST = 2): y and z periodic powers of two, x the ghost-extended, NON-periodic local planes whose
    // lattice planes are mesh planes [xoff, xoff + px): -1 where the window leaves them
    __device__ __forceinline__ int point_slab(const Geom &g, int x0, int y0, int z0, const Box &b, int &rx, int &ry, int &rz) const {
        rx = jx - 1 - b.hi[0];
        ry = jy - 1 - b.hi[1];
        rz = jz - 1 - b.hi[2];
        const int gx = x0 + rx - g.xoff, gy = (y0 + ry) & (g.ny - 1), gz = (z0 + rz) & (g.nz - 1);
        const int lz = __builtin_ctz((unsigned)g.nz), ly = __builtin_ctz((unsigned)g.ny);
        return (unsigned)gx < (unsigned)g.px ? ((((gx << ly) | gy) << lz) | gz) : -1;
    }
    // any tileable mesh (FAST = 0): one conditional wrap per axis (box_tile_kernel keeps |lo|, |hi| <= 12, so a window point is
    // less than one period away), the ghost-extended planes of a slab plan along x
    __device__ __forceinline__ int point_any(const Geom &g, int x0, int y0, int z0, const Box &b, int &rx, int &ry, int &rz) const {
        rx = jx - 1 - b.hi[0];
        ry = jy - 1 - b.hi[1];
        rz = jz - 1 - b.hi[2];
        int gx = x0 + rx;
        if (g.xslab) {  // ghost-extended slab: lattice planes are mesh planes [xoff, xoff + px), no wrap
            gx -= g.xoff;
            if ((unsigned)gx >= (unsigned)g.px) return -1;
        } else
            gx = wrap_once(gx, g.nx);
        const int gy = wrap_once(y0 + ry, g.ny), gz = wrap_once(z0 + rz, g.nz);
        return (gx * g.ny + gy) * g.nz + gz;
    }
    template <int FAST>
    __device__ __forceinline__ int point(const Geom &g, int x0, int y0, int z0, const Box &b, int &rx, int &ry, int &rz) const {
        if (!valid()) return -1;
        return FAST == 2 ? point_slab(g, x0, y0, z0, b, rx, ry, rz) : (FAST == 1 ? point_fast(g, x0, y0, z0, b, rx, ry, rz) : point_any(g, x0, y0, z0, b, rx, ry, rz));
    }
};
// 12-byte particle record i of an array of fewer than 2^32 / 12 records (what the FAST instantiations are launched for): uniform
// base + 32-bit byte offset, so the address costs two full-rate shifts / adds instead of a quarter-rate 64-bit multiply-add
// (round 4: density paint 0.81 -> 0.79 ms, three-component paint 1.93 -> 1.82 ms at 512^3, same box)
template <int FAST>
__device__ __forceinline__ P3 load3w(const float *__restrict__ p, int i) {
    if (FAST) return *reinterpret_cast<const P3 *>(reinterpret_cast<const char *>(p) + (((unsigned)i << 3) + ((unsigned)i << 2)));
    return load3(p, i);
}

// ------------------------------------------------------------------------------------------------
// density paint.  WMODE 0: unweighted (2^-30 fixed point, scalar weight applied at the flush); 1: weighted, fixed point with
// the max|w| scale; 2: weighted, f64 accumulators (non-finite weights only: runs when tile_scale().mode == 2, WMODE 1 otherwise)
// (amdgpu_num_sgpr: with more than 80 scalar registers a CU admits 7 waves per SIMD instead of 8, i.e. three of these
// 512-thread workgroups instead of four -- measured +45 % on the kernel; MI355X_MICROARCH.md "Residency")
template <int WMODE, int THREADS, int U, int FAST>
__device__ __forceinline__ void paint_tile_body(const Geom &g, const float *__restrict__ disp, const float *__restrict__ w,
                                                int64_t wstride, float wscalar, float *__restrict__ mesh, int accumulate, const TileLists &L,
                                                const unsigned *__restrict__ wmax_bits, int duty, u64 *tile, int *sh27, int *sus) {
    // The LDS tile is padded by one cell at either end of z (rows of BZ = 18): the two z corners of a deposit need no bounds test,
    // four masked blocks of two atomics instead of eight of one (the pads' sums are never read)
    constexpr int B = MCPM_TILE, BZ = B + 2, NT = B * B * B, NTP = B * B * BZ;
    double *dtile = reinterpret_cast<double *>(tile);
    TScale sc = {1073741824.f, 9.313225746154785e-10, 1};
    if (WMODE != 0) {
        sc = tile_scale(wmax_bits);
        if ((WMODE == 1 && sc.mode == 2) || (WMODE == 2 && sc.mode != 2)) return;   // the other instantiation paints
    }
    const int ntx = g.nx / B, nty = g.ny / B, ntz = g.nz / B;
    int tx, ty, tz;
    tile_of_block(ntx, nty, ntz, tx, ty, tz, L.order);
    const int x0 = tx * B, y0 = ty * B, z0 = tz * B;
    for (int i = threadIdx.x; i < NTP; i += THREADS) tile[i] = 0ull;
    if (threadIdx.x == 0) sus[MCPM_SUS] = 0;
    const Box bx = block_box(L, (tx * nty + ty) * ntz + tz);      // the tile's window (wave-uniform)
    const int Wx = B + 1 + bx.hi[0] - bx.lo[0], Wy = B + 1 + bx.hi[1] - bx.lo[1], Wz = B + 1 + bx.hi[2] - bx.lo[2], NW = Wx * Wy * Wz;
    float slo[3] = {(float)bx.lo[0], (float)bx.lo[1], (float)bx.lo[2]}, shi[3] = {(float)bx.hi[0], (float)bx.hi[1], (float)bx.hi[2]};
    if (L.toff) sure_intervals(g, L, tx, ty, tz, ntx, nty, ntz, sh27, slo, shi);   // windows that follow the bulk displacement (optional)
    __syncthreads();

    WinIter<THREADS> wi(threadIdx.x, Wx, Wy, Wz);
    for (int j0 = threadIdx.x; j0 < NW; j0 += THREADS * U) {
        P3 d[U];
        float wt[U];
        int rxs[U], rys[U], rzs[U], gis[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (one guard for index and load: on periodic plans a window point always has a lattice point)
            gis[u] = -1;
            d[u] = P3{0.f, 0.f, 0.f};
            wt[u] = 0.f;
            if (wi.valid()) {
                gis[u] = wi.template point<FAST>(g, x0, y0, z0, bx, rxs[u], rys[u], rzs[u]);
                if (FAST == 1 || gis[u] >= 0) {
                    d[u] = load3w<FAST>(disp, gis[u]);
                    wt[u] = WMODE ? w[(int64_t)gis[u] * wstride] : 1.f;
                }
            }
            wi.next();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gis[u] < 0) continue;
            const int rx = rxs[u], ry = rys[u], rz = rzs[u];
            const float fx = floorf(d[u].x), fy = floorf(d[u].y), fz = floorf(d[u].z);
            // Coverage duty: six compares (what the round-1 outlier test cost; NaN fails them).  What they cannot clear goes
            // to the suspect list of the home tile, for the exact test of coverage_duty_kernel (400 instructions: inlined in
            // this unrolled loop, and entered by most waves, they doubled the kernel's time).
            // Slab plans: a base cell beyond the ghost planes is nobody's to deposit; its HOME tile must hand it to the exact
            // test too (is_wild() sends it to the clamping, counting leftover kernel) -- with bulk-centred windows the sure
            // interval alone reaches |floor(d_x)| = 8 + H, past a ghost region of 8 planes, and such a particle used to vanish
            // uncounted.
            const bool unsure = !(fx >= slo[0] && fx <= shi[0] && fy >= slo[1] && fy <= shi[1] && fz >= slo[2] && fz <= shi[2]);
            const bool home = (unsigned)rx < (unsigned)B && (unsigned)ry < (unsigned)B && (unsigned)rz < (unsigned)B;
            // (one divergent region for the rare cases -- wild: unsure and not tame; beyond: slab plans only -- and one append site: the
            // tile kernels are bound by scalar and vector issue alike, and every divergent `if` costs four scalar instructions and a branch)
            const int cx = rx + (int)fx, cy = ry + (int)fy, cz = rz + (int)fz;      // (garbage for a wild particle: skipped below)
            bool skip = false;
            if (unsure || (FAST != 1 && g.xslab)) {
                const bool tame = fabsf(d[u].x) < MCPM_TAME && fabsf(d[u].y) < MCPM_TAME && fabsf(d[u].z) < MCPM_TAME;
                const bool beyond = FAST != 1 && g.xslab && tame && (x0 + cx < 0 || x0 + cx > g.nx - 2);
                if ((unsure || beyond) && duty && home) append_suspect(L, sus, gis[u]);
                skip = !tame || beyond;      // wild: the leftover kernel's; beyond: clamped + counted by paint_leftover_kernel
            }
            if (skip) continue;
            if (cx >= -1 && cx < B && cy >= -1 && cy < B && cz >= -1 && cz < B) {
                const float tx1 = d[u].x - fx, ty1 = d[u].y - fy, tz1 = d[u].z - fz;
                const float s0 = WMODE == 2 ? wt[u] : (WMODE == 1 ? wt[u] * sc.S : sc.S);   // exact power-of-two scaling
                const float kx[2] = {(1.f - tx1) * s0, tx1 * s0}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
                // one LDS base address, corners at immediate offsets; a corner outside the tile is skipped
                const bool vx[2] = {cx >= 0, cx < B - 1}, vy[2] = {cy >= 0, cy < B - 1};
                const int base = (cx * B + cy) * BZ + cz + 1;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const float wxy = kx[a] * ky[bb];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            // (depositing an outside corner into a per-lane trash cell instead, 8 unconditional atomics and no
                            // per-corner exec mask, is SLOWER: 0.956 vs 0.902 ms at 512^3 -- +90 vector instructions per 4 visits)
                            if (vx[a] && vy[bb]) {
                                const int q = base + (a * B + bb) * BZ + e;
                                if (WMODE == 2) atomicAdd(dtile + q, (double)(wxy * kz[e]));
                                else if (WMODE == 1) atomicAdd(tile + q, (u64)(long long)cvt_rpi(wxy * kz[e]));
                                else atomicAdd(tile + q, (u64)(unsigned)cvt_rpi(wxy * kz[e]));
                            }
                    }
            }
        }
    }
    // home lattice points outside the own window (the box does not contain floor(d) = 0 .. -1 on some axis): all of them are suspects
    if (duty && !(bx.hi[0] >= -1 && bx.lo[0] <= 0 && bx.hi[1] >= -1 && bx.lo[1] <= 0 && bx.hi[2] >= -1 && bx.lo[2] <= 0)) {
        for (int j = threadIdx.x; j < NT; j += THREADS) {
            const int rz = j % B, rr = j / B, ry = rr % B, rx = rr / B;
            const bool inwin = (unsigned)(rx + 1 + bx.hi[0]) < (unsigned)Wx && (unsigned)(ry + 1 + bx.hi[1]) < (unsigned)Wy &&
                               (unsigned)(rz + 1 + bx.hi[2]) < (unsigned)Wz;
            if (inwin) continue;
            int gx = x0 + rx;
            if (g.xslab) {
                gx -= g.xoff;
                if ((unsigned)gx >= (unsigned)g.px) continue;
            }
            const int gi = (gx * g.ny + y0 + ry) * g.nz + z0 + rz;
            append_suspect(L, sus, gi);
        }
    }
    __syncthreads();
    flush_suspects(L, sus);

    const double s = WMODE == 0 ? (double)wscalar * sc.Sinv : sc.Sinv;
    for (int i = threadIdx.x; i < NT / 4; i += THREADS) {
        const int lz = (i % (B / 4)) * 4, r = i / (B / 4), ly = r % B, lx = r / B;
        float4 v;
        const int t0 = r * BZ + 1 + lz;      // first of the four cells in the padded tile
        if (WMODE == 2) v = make_float4((float)dtile[t0], (float)dtile[t0 + 1], (float)dtile[t0 + 2], (float)dtile[t0 + 3]);
        else if (WMODE == 1)
            v = make_float4((float)((double)(long long)tile[t0] * s), (float)((double)(long long)tile[t0 + 1] * s),
                            (float)((double)(long long)tile[t0 + 2] * s), (float)((double)(long long)tile[t0 + 3] * s));
        else
            v = make_float4((float)((double)tile[t0] * s), (float)((double)tile[t0 + 1] * s), (float)((double)tile[t0 + 2] * s),
                            (float)((double)tile[t0 + 3] * s));
        float4 *dst = reinterpret_cast<float4 *>(mesh + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz);
        if (accumulate) {
            const float4 o = *dst;
            v.x += o.x;
            v.y += o.y;
            v.z += o.z;
            v.w += o.w;
        }
        *dst = v;      // (a streaming store here is slower: 0.915 vs 0.902 ms at 512^3)
    }
}

template <int WMODE, int THREADS, int U, int FAST = 0>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_num_sgpr(80))) void paint_tile_kernel(Geom g, const float *__restrict__ disp, const float *__restrict__ w,
                                                             int64_t wstride, float wscalar, float *__restrict__ mesh,
                                                             int accumulate, TileLists L, const unsigned *__restrict__ wmax_bits,
                                                             int duty) {
    __shared__ u64 tile[MCPM_TILE * MCPM_TILE * (MCPM_TILE + 2)];      // z-padded (paint_tile_body)
    __shared__ int sh27[27];
    __shared__ int sus[MCPM_SUS + 2];
    paint_tile_body<WMODE, THREADS, U, FAST>(g, disp, w, wstride, wscalar, mesh, accumulate, L, wmax_bits, duty, tile, sh27, sus);
}

// base cell of a bucketed particle relative to the tile origin, brought into [-1, n-1) (the tile sees it at c in [-1, 16))
__device__ __forceinline__ bool bucket_cell(const Geom &g, int gi, const P3 &d, int x0, int y0, int z0, int &cx, int &cy, int &cz) {
    const int qz = gi % g.nz, r = gi / g.nz, qy = r % g.ny, qx = r / g.ny;   // lattice == mesh
    cx = qx + g.xoff + (int)floorf(d.x) - x0;
    if (!g.xslab) cx = pymod(cx + 1, g.nx) - 1;
    cy = pymod(qy + (int)floorf(d.y) - y0 + 1, g.ny) - 1;
    cz = pymod(qz + (int)floorf(d.z) - z0 + 1, g.nz) - 1;
    return cx >= -1 && cx < MCPM_TILE && cy >= -1 && cy < MCPM_TILE && cz >= -1 && cz < MCPM_TILE;
}

// deposits the bucket of every tile through an LDS integer tile and adds the touched cells to the mesh
template <int WMODE>
__device__ __forceinline__ void paint_bucket_body(const Geom &g, const float *__restrict__ disp, const float *__restrict__ w,
                                                  int64_t wstride, float wscalar, float *__restrict__ mesh, const TileLists &L,
                                                  const unsigned *__restrict__ wmax_bits, u64 *tile, const int bid, const int nblk) {
    constexpr int B = MCPM_TILE, NT = B * B * B;
    const int ntl = L.cnts[C_NTILES];
    if (bid >= ntl) return;
    TScale sc = {1073741824.f, 9.313225746154785e-10, 1};
    if (WMODE != 0) sc = tile_scale(wmax_bits);
    const bool f64 = WMODE != 0 && sc.mode == 2;     // non-finite weights: doubles (uniform over the launch)
    double *dtile = reinterpret_cast<double *>(tile);
    const int ntz = g.nz / B, nty = g.ny / B;
    const double s = WMODE == 0 ? (double)wscalar * sc.Sinv : sc.Sinv;
    for (int it = bid; it < ntl; it += nblk) {
        const int t = L.nonempty[it];
        const int cnt = L.bcnt[t];
        if (cnt > L.cap) continue;     // an overflowed bucket is dropped: the repair pass deposits that tile's pairs
        const int x0 = (t / (ntz * nty)) * B, y0 = ((t / ntz) % nty) * B, z0 = (t % ntz) * B;
        __syncthreads();
        for (int i = threadIdx.x; i < NT; i += 256) tile[i] = 0ull;
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const int gi = L.bucket[(int64_t)t * L.cap + k];
            const P3 d = load3(disp, gi);
            int cx, cy, cz;
            if (!bucket_cell(g, gi, d, x0, y0, z0, cx, cy, cz)) continue;
            const float wt = WMODE ? w[(int64_t)gi * wstride] : 1.f;
            const float tx1 = d.x - floorf(d.x), ty1 = d.y - floorf(d.y), tz1 = d.z - floorf(d.z);
            const float s0 = f64 ? wt : (WMODE == 1 ? wt * sc.S : sc.S);
            const float kx[2] = {(1.f - tx1) * s0, tx1 * s0}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int x = cx + a, y = cy + bb, z = cz + e;
                        if ((unsigned)x < (unsigned)B && (unsigned)y < (unsigned)B && (unsigned)z < (unsigned)B) {
                            const int q = (x * B + y) * B + z;
                            const float v = kx[a] * ky[bb] * kz[e];
                            if (f64) atomicAdd(dtile + q, (double)v);
                            else if (WMODE == 1) atomicAdd(tile + q, (u64)(long long)cvt_rpi(v));
                            else atomicAdd(tile + q, (u64)(unsigned)cvt_rpi(v));
                        }
                    }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NT; i += 256) {
            const u64 a = tile[i];
            if (a == 0ull) continue;
            const int lz = i % B, r = i / B, ly = r % B, lx = r / B;
            const float v = f64 ? (float)dtile[i] : (WMODE == 1 ? (float)((double)(long long)a * s) : (float)((double)a * s));
            mesh[((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz] += v;   // this workgroup alone writes tile t here
        }
        if (threadIdx.x == 0) atomicAdd(L.cnts + C_BUCKETED, cnt);
    }
}

// What the integer tiles could not take, with f32 global atomics (counted): (1) wild particles, all eight corners;
// (2) repair pass, only when some bucket overflowed: every (particle, tile) pair whose tile is marked (bcnt > cap) and whose
// window misses the particle -- the same predicate as the coverage duty, re-evaluated over all particles.
template <int NC>
__device__ __forceinline__ void paint_leftover_body(const Geom &g, const float *__restrict__ disp, const float *__restrict__ w,
                                                    int64_t wstride, float wscalar, float *__restrict__ mesh, int64_t M,
                                                    const TileLists &L, const int bid, const int nblk) {
    const int nw = min(L.cnts[C_WILD], L.listcap);
    if (bid == 0 && threadIdx.x == 0) L.cnts[C_LAST] = L.cnts[C_WILD] + L.cnts[C_PAIRS];
    for (int k = bid * blockDim.x + threadIdx.x; k < nw; k += nblk * blockDim.x) {
        const int gi = L.wild[k];
        PIdx pi;
        pi.i = gi;
        pi.ipz = gi % g.nz;
        const int r = gi / g.nz;
        pi.ipy = r % g.ny;
        pi.ipx = r / g.ny;
        pi.valid = true;
        const P3 d = load3(disp, gi);
        float wt[3];
#pragma unroll
        for (int c = 0; c < NC; ++c) wt[c] = w ? w[(int64_t)gi * wstride + c] : wscalar;
        if (!(fabsf(d.x) < 3.0e4f && fabsf(d.y) < 3.0e4f && fabsf(d.z) < 3.0e4f)) {
            // non-finite position, or beyond the int16 index range of the reference (nbody.py:369): no cell can be named;
            // the first cell of every component is made non-finite so that the caller sees it
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) atomicAdd(mesh + cc * M, __int_as_float(0x7fc00000));
            continue;
        }
        int c[3];
        float f[3];
        locate<MCPM_POS_LATTICE, 2>(g, pi, d, c, f);
        // slab mode: a particle displaced beyond the ghost planes cannot be deposited on this rank; it is clamped to the
        // edge and counted (mcpm_plan_slab_oob) so that the host can widen the ghost region
        if (g.xslab && (c[0] < 0 || c[0] > g.nx - 2)) atomicAdd(L.cnts + C_OOB, 1);
        Stencil<2> s(g, c);
        const float kx[2] = {1.f - f[0], f[0]}, ky[2] = {1.f - f[1], f[1]}, kz[2] = {1.f - f[2], f[2]};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float k3 = kx[a] * ky[b] * kz[e];
                    float *m = mesh + s.xo[a] + s.yo[b] + s.zo[e];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) atomicAdd(m + cc * M, wt[cc] * k3);
                }
    }
    if (L.cnts[C_PAIRS] == 0) return;
    const int ntx = g.nx / MCPM_TILE, nty = g.ny / MCPM_TILE, ntz = g.nz / MCPM_TILE;
    const int64_t np = (int64_t)g.px * g.py * g.pz;
    for (int64_t gi = (int64_t)bid * blockDim.x + threadIdx.x; gi < np; gi += (int64_t)nblk * blockDim.x) {
        const int qz = (int)(gi % g.nz), r = (int)(gi / g.nz), qy = r % g.ny, qx = r / g.ny + g.xoff;
        const P3 d = load3(disp, gi);
        const int tx = qx >> 4, ty = qy >> 4, tz = qz >> 4;
        if (is_wild(g, d, tx * MCPM_TILE, qx & 15)) continue;
        const float fx = floorf(d.x), fy = floorf(d.y), fz = floorf(d.z);
        const float kx[2] = {1.f - (d.x - fx), d.x - fx}, ky[2] = {1.f - (d.y - fy), d.y - fy}, kz[2] = {1.f - (d.z - fz), d.z - fz};
        for_uncovered(g, L, tx, ty, tz, ntx, nty, ntz, tile_box(L, (tx * nty + ty) * ntz + tz), qx & 15, qy & 15, qz & 15, (int)fx, (int)fy,
                         (int)fz, [&](int tidx, int cx, int cy, int cz) {
                             if (L.bcnt[tidx] <= L.cap) return;      // that tile's bucket was deposited by the bucket kernel
                             const int x0 = (tidx / (ntz * nty)) * MCPM_TILE, y0 = ((tidx / ntz) % nty) * MCPM_TILE, z0 = (tidx % ntz) * MCPM_TILE;
                             float wt[3];
                             for (int c = 0; c < NC; ++c) wt[c] = w ? w[gi * wstride + c] : wscalar;
                             for (int a = 0; a < 2; ++a)
                                 for (int b = 0; b < 2; ++b)
                                     for (int e = 0; e < 2; ++e) {
                                         const int x = cx + a, y = cy + b, z = cz + e;
                                         if ((unsigned)x < (unsigned)MCPM_TILE && (unsigned)y < (unsigned)MCPM_TILE && (unsigned)z < (unsigned)MCPM_TILE) {
                                             const float k3 = kx[a] * ky[b] * kz[e];
                                             float *m = mesh + ((int64_t)(x0 + x) * g.ny + (y0 + y)) * g.nz + z0 + z;
                                             for (int cc = 0; cc < NC; ++cc) atomicAdd(m + cc * M, wt[cc] * k3);
                                         }
                                     }
                         });
    }
}

// Epilogue of a tiled density paint, ONE launch (it used to be two: at 128^3, where a kernel lasts a few microseconds, every
// launch costs about as much as the kernel): blocks [0, nbk) deposit the buckets, blocks [nbk, nbk + nlo) run the
// global-atomic leftovers (wild particles, overflowed buckets).  The two touch disjoint (particle, tile) pairs and both only
// read what the coverage kernel left, so they need no order between them.
template <int WMODE>
__global__ __launch_bounds__(256) void paint_epilogue_kernel(Geom g, const float *__restrict__ disp, const float *__restrict__ w,
                                                             int64_t wstride, float wscalar, float *__restrict__ mesh, int64_t M,
                                                             TileLists L, const unsigned *__restrict__ wmax_bits, int nbk) {
    __shared__ u64 tile[MCPM_TILE * MCPM_TILE * MCPM_TILE];
    if ((int)blockIdx.x < nbk) paint_bucket_body<WMODE>(g, disp, w, wstride, wscalar, mesh, L, wmax_bits, tile, (int)blockIdx.x, nbk);
    else {
        paint_leftover_body<1>(g, disp, w, wstride, wscalar, mesh, M, L, (int)blockIdx.x - nbk, (int)gridDim.x - nbk);
    }
}

// ------------------------------------------------------------------------------------------------
// Three weighted paints at once (the adjoint of a three-component read: weights[N][3] -> three meshes M apart).
//
// Fixed-point tiles.  An LDS f64 atomic costs about twice a 64-bit integer one under the bank conflicts of real deposits
// (tools/lds_atomic_bench.hip).  The three weighted corner contributions are rounded to 32-bit fixed point with a common
// power-of-two scale S = 2^24 / 2^e (2^e <= max|w| < 2^(e+1), so one contribution is below 2^25 and a cell holds 64 maximal
// ones) and travel in TWO 64-bit integer atomics per corner:
//     word A = c0 + 2^32 c1        word B = c2 + 2^32 bound,   bound += max_c |contribution_c| / 2^11 + 1 (rounded up)
// A signed low field added as a sign-extended 64-bit number leaves the high field exact as long as the low field's true sum
// fits 32 bits, and modular arithmetic makes intermediate wrap-arounds harmless, so the sums are exact integers and
// independent of the arrival order (bitwise reproducible).  `bound` proves it: a component field can only leave the int32
// range if sum |contribution| >= 2^31, i.e. bound >= 2^20; the bound field itself cannot overflow (< 2^14 + 1 per deposit,
// < 2^17 deposits per cell).  A tile holding a cell with bound >= 2^19 is not written: it is appended to the redo list and
// painted by the f64 kernel.  Non-finite or tiny (< 2^-97) max|w| sends every tile there.  Rounding: half a unit per deposit
// = max|w| 2^-25, so the mesh differs from the exact sums by ~1e-8 max|w| per cell.
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ w, int64_t stride, int64_t n, unsigned *__restrict__ out) {
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

// redo: nullptr = first (fixed-point) pass over all tiles, appending flagged tiles to `redo_out`; otherwise the f64 pass
// over the tiles listed in redo ([0] = count, then indices).  F64: accumulators are doubles (96 KB) instead of packed fields.
// One tile of the three-component paint (redo_tile < 0: the tile of this block; else the given tile, for the f64 repaint).
template <bool F64, int THREADS, int U, int FAST>
__device__ __forceinline__ void paint3_tile_body(const Geom &g, const float *__restrict__ disp, const float *__restrict__ w3,
                                                 float *__restrict__ mesh, int64_t M, int accumulate, const TileLists &L,
                                                 const unsigned *__restrict__ wmax_bits, int *__restrict__ redo_out, int redo_tile, int duty,
                                                 u64 *tile, int &flagged, int *sh27, int *sus) {
    constexpr int B = MCPM_TILE, BZ = B + 2, NT = B * B * B, NTP = B * B * BZ;      // z-padded LDS tiles: see paint_tile_body
    double *dtile = reinterpret_cast<double *>(tile);
    const int ntx = g.nx / B, nty = g.ny / B, ntz = g.nz / B;
    int tx, ty, tz;
    if (redo_tile >= 0) {
        tz = redo_tile % ntz;
        ty = (redo_tile / ntz) % nty;
        tx = redo_tile / (ntz * nty);
    } else
        tile_of_block(ntx, nty, ntz, tx, ty, tz, L.order);
    const int x0 = tx * B, y0 = ty * B, z0 = tz * B, tidx = (tx * nty + ty) * ntz + tz;
    const Box bx = block_box(L, tidx);      // the tile's window (wave-uniform)
    const int Wx = B + 1 + bx.hi[0] - bx.lo[0], Wy = B + 1 + bx.hi[1] - bx.lo[1], Wz = B + 1 + bx.hi[2] - bx.lo[2], NW = Wx * Wy * Wz;
    bool deposit = true;
    float S = 1.f, Sinv = 1.f;
    if (!F64) {
        unsigned wb = wmax_bits[(threadIdx.x & (MCPM_FX_SLOTS - 1)) * MCPM_FX_STRIDE];   // maximum over the slots, in every wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wb = max(wb, (unsigned)__shfl_xor((int)wb, o));
        const unsigned be = wb >> 23;
        if (wb == 0u) {              // all weights are zero: the tile is zero (or unchanged); nothing lands anywhere
            if (!accumulate)
                for (int i = threadIdx.x; i < 3 * NT / 4; i += THREADS) {
                    const int cc = i / (NT / 4), ii = i - cc * (NT / 4);
                    const int lz = (ii % (B / 4)) * 4, r = ii / (B / 4), ly = r % B, lx = r / B;
                    *reinterpret_cast<float4 *>(mesh + cc * M + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz) =
                        make_float4(0.f, 0.f, 0.f, 0.f);
                }
            return;
        }
        if (be < 30u || be > 254u) {     // tiny / non-finite weights: the f64 kernel paints this tile; the duty stays here
            deposit = false;
            if (threadIdx.x == 0) redo_out[1 + atomicAdd(redo_out, 1)] = tidx;
        } else {
            S = __uint_as_float((278u - be) << 23);      // 2^(24-e)
            Sinv = __uint_as_float((be - 24u) << 23);    // 2^(e-24)
        }
    }
    if (threadIdx.x == 0) {
        flagged = 0;
        sus[MCPM_SUS] = 0;
    }
    if (deposit)
        for (int i = threadIdx.x; i < (F64 ? 3 : 2) * NTP; i += THREADS) tile[i] = 0ull;
    float slo[3] = {(float)bx.lo[0], (float)bx.lo[1], (float)bx.lo[2]}, shi[3] = {(float)bx.hi[0], (float)bx.hi[1], (float)bx.hi[2]};
    if (L.toff) sure_intervals(g, L, tx, ty, tz, ntx, nty, ntz, sh27, slo, shi);
    __syncthreads();

    WinIter<THREADS> wi(threadIdx.x, Wx, Wy, Wz);
    for (int j0 = threadIdx.x; j0 < NW; j0 += THREADS * U) {
        P3 d[U], wt[U];
        int rxs[U], rys[U], rzs[U], gis[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gis[u] = -1;
            d[u] = P3{0.f, 0.f, 0.f};
            wt[u] = P3{0.f, 0.f, 0.f};
            if (wi.valid()) {
                gis[u] = wi.template point<FAST>(g, x0, y0, z0, bx, rxs[u], rys[u], rzs[u]);
                if (FAST == 1 || gis[u] >= 0) {
                    d[u] = load3w<FAST>(disp, gis[u]);
                    wt[u] = load3w<FAST>(w3, gis[u]);
                }
            }
            wi.next();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gis[u] < 0) continue;
            const int rx = rxs[u], ry = rys[u], rz = rzs[u];
            const float fx = floorf(d[u].x), fy = floorf(d[u].y), fz = floorf(d[u].z);
            // Coverage duty: six compares (what the round-1 outlier test cost; NaN fails them).  What they cannot clear goes
            // to the suspect list of the home tile, for the exact test of coverage_duty_kernel (400 instructions: inlined in
            // this unrolled loop, and entered by most waves, they doubled the kernel's time).
            // Slab plans: a base cell beyond the ghost planes is nobody's to deposit; its HOME tile must hand it to the exact
            // test too (is_wild() sends it to the clamping, counting leftover kernel) -- with bulk-centred windows the sure
            // interval alone reaches |floor(d_x)| = 8 + H, past a ghost region of 8 planes, and such a particle used to vanish
            // uncounted.
            const bool unsure = !(fx >= slo[0] && fx <= shi[0] && fy >= slo[1] && fy <= shi[1] && fz >= slo[2] && fz <= shi[2]);
            const bool home = (unsigned)rx < (unsigned)B && (unsigned)ry < (unsigned)B && (unsigned)rz < (unsigned)B;
            // (one divergent region for the rare cases -- wild: unsure and not tame; beyond: slab plans only -- and one append site: the
            // tile kernels are bound by scalar and vector issue alike, and every divergent `if` costs four scalar instructions and a branch)
            const int cx = rx + (int)fx, cy = ry + (int)fy, cz = rz + (int)fz;      // (garbage for a wild particle: skipped below)
            bool skip = false;
            if (unsure || (FAST != 1 && g.xslab)) {
                const bool tame = fabsf(d[u].x) < MCPM_TAME && fabsf(d[u].y) < MCPM_TAME && fabsf(d[u].z) < MCPM_TAME;
                const bool beyond = FAST != 1 && g.xslab && tame && (x0 + cx < 0 || x0 + cx > g.nx - 2);
                if ((unsure || beyond) && duty && home) append_suspect(L, sus, gis[u]);
                skip = !tame || beyond;      // wild: the leftover kernel's; beyond: clamped + counted by paint_leftover_kernel
            }
            if (skip) continue;
            if (deposit && cx >= -1 && cx < B && cy >= -1 && cy < B && cz >= -1 && cz < B) {
                const float tx1 = d[u].x - fx, ty1 = d[u].y - fy, tz1 = d[u].z - fz;
                const float kx[2] = {1.f - tx1, tx1}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
                const bool vx[2] = {cx >= 0, cx < B - 1}, vy[2] = {cy >= 0, cy < B - 1};
                const int base = (cx * B + cy) * BZ + cz + 1;
                if (F64) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            const float kxy = kx[a] * ky[bb];
#pragma unroll
                            for (int e = 0; e < 2; ++e)
                                if (vx[a] && vy[bb]) {
                                    const float k = kxy * kz[e];
                                    double *q = dtile + base + (a * B + bb) * BZ + e;
                                    atomicAdd(q, (double)(wt[u].x * k));
                                    atomicAdd(q + NTP, (double)(wt[u].y * k));
                                    atomicAdd(q + 2 * NTP, (double)(wt[u].z * k));
                                }
                        }
                } else {
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    const float sx = wt[u].x * S, sy = wt[u].y * S, sz = wt[u].z * S;
                    const float mw = fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)) * (1.f / 2048.f);
                    const v2f s01 = {sx, sy}, s2m = {sz, mw}, c01 = {0.f, 1.f};     // packed f32 math: two products per instruction
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            const float kxy = kx[a] * ky[bb];
#pragma unroll
                            for (int e = 0; e < 2; ++e)
                                if (vx[a] && vy[bb]) {
                                    const float k = kxy * kz[e];
                                    const v2f kk = {k, k};
                                    const v2f p01 = s01 * kk, p2m = __builtin_elementwise_fma(s2m, kk, c01);
                                    const int i0 = cvt_rpi(p01.x), i1 = cvt_rpi(p01.y), i2 = cvt_rpi(p2m.x);
                                    const unsigned ib = (unsigned)p2m.y;
                                    const u64 wa = ((u64)(unsigned)(i1 + (i0 >> 31)) << 32) | (unsigned)i0;
                                    const u64 wbv = ((u64)(ib + (unsigned)(i2 >> 31)) << 32) | (unsigned)i2;
                                    u64 *q = tile + base + (a * B + bb) * BZ + e;
                                    atomicAdd(q, wa);
                                    atomicAdd(q + NTP, wbv);
                                }
                        }
                }
            }
        }
    }
    if (duty && !(bx.hi[0] >= -1 && bx.lo[0] <= 0 && bx.hi[1] >= -1 && bx.lo[1] <= 0 && bx.hi[2] >= -1 && bx.lo[2] <= 0)) {
        for (int j = threadIdx.x; j < NT; j += THREADS) {
            const int rz = j % B, rr = j / B, ry = rr % B, rx = rr / B;
            const bool inwin = (unsigned)(rx + 1 + bx.hi[0]) < (unsigned)Wx && (unsigned)(ry + 1 + bx.hi[1]) < (unsigned)Wy &&
                               (unsigned)(rz + 1 + bx.hi[2]) < (unsigned)Wz;
            if (inwin) continue;
            int gx = x0 + rx;
            if (g.xslab) {
                gx -= g.xoff;
                if ((unsigned)gx >= (unsigned)g.px) continue;
            }
            const int gi = (gx * g.ny + y0 + ry) * g.nz + z0 + rz;
            append_suspect(L, sus, gi);
        }
    }
    __syncthreads();
    flush_suspects(L, sus);
    if (!deposit) return;

    if (!F64) {   // overflow proof: bound field of every cell
        int over = 0;
        for (int i = threadIdx.x; i < NT; i += THREADS) {
            const long long bw = (long long)tile[NTP + (i / B) * BZ + 1 + i % B];
            const int c2 = (int)(unsigned)bw;
            over |= (unsigned)((bw - (long long)c2) >> 32) >= (1u << 19);
        }
        if (over) flagged = 1;
        __syncthreads();
        if (flagged) {
            if (threadIdx.x == 0) redo_out[1 + atomicAdd(redo_out, 1)] = tidx;
            return;
        }
    }
    for (int i = threadIdx.x; i < NT / 4; i += THREADS) {
        const int lz = (i % (B / 4)) * 4, r = i / (B / 4), ly = r % B, lx = r / B;
        float v[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (F64) {
                v[0][q] = (float)dtile[r * BZ + 1 + lz + q];
                v[1][q] = (float)dtile[NTP + r * BZ + 1 + lz + q];
                v[2][q] = (float)dtile[2 * NTP + r * BZ + 1 + lz + q];
            } else {
                const long long aw = (long long)tile[r * BZ + 1 + lz + q], bw = (long long)tile[NTP + r * BZ + 1 + lz + q];
                const int c0 = (int)(unsigned)aw, c2 = (int)(unsigned)bw;
                const int c1 = (int)((aw - (long long)c0) >> 32);
                v[0][q] = (float)c0 * Sinv;
                v[1][q] = (float)c1 * Sinv;
                v[2][q] = (float)c2 * Sinv;
            }
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

// redo_in == nullptr: the first pass, one workgroup per tile.  Otherwise the f64 repaint of the tiles listed in redo_in
// ([0] = count, then indices): a small grid that walks the list (it is empty on every PM workload: a whole-mesh launch of
// workgroups that return at once cost 30 us per adjoint step at 512^3).
// (four waves per SIMD = two 512-thread workgroups per CU: with three candidate bodies the allocator would take 133 registers)
template <bool F64, int THREADS, int U, int FAST = 0>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4))) void paint3_tile_kernel(Geom g, const float *__restrict__ disp, const float *__restrict__ w3,
                                                              float *__restrict__ mesh, int64_t M, int accumulate, TileLists L,
                                                              const unsigned *__restrict__ wmax_bits, int *__restrict__ redo_out,
                                                              const int *__restrict__ redo_in, int duty) {
    constexpr int NT = MCPM_TILE * MCPM_TILE * (MCPM_TILE + 2);      // z-padded
    __shared__ u64 tile[(F64 ? 3 : 2) * NT];   // 72 KB (two workgroups per CU) / 108 KB
    __shared__ int flagged;
    __shared__ int sh27[27];
    __shared__ int sus[MCPM_SUS + 2];
    if (!redo_in) {
        paint3_tile_body<F64, THREADS, U, FAST>(g, disp, w3, mesh, M, accumulate, L, wmax_bits, redo_out, -1, duty, tile, flagged, sh27, sus);
        return;
    }
    if (!F64) return;      // only the f64 instance repaints (the fixed-point one would carry three more window-walk bodies, and their
                           // scalar-register pressure, for a branch it never takes)
    const int n = redo_in[0];
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        paint3_tile_body<F64, THREADS, U, FAST>(g, disp, w3, mesh, M, accumulate, L, wmax_bits, redo_out, redo_in[1 + k], duty, tile, flagged, sh27, sus);
        __syncthreads();
    }
}

// The fixed-point first pass with 1024 threads per tile, for meshes of at most 1024 tiles (128^3: 512 tiles, two per CU, and a tile's
// time is latency): two such workgroups per CU need eight waves per SIMD, i.e. at most 64 VGPRs (the 512-thread instance takes 75).
template <int FAST>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void paint3_tile_wide_kernel(Geom g, const float *__restrict__ disp,
                                                              const float *__restrict__ w3, float *__restrict__ mesh, int64_t M, int accumulate,
                                                              TileLists L, const unsigned *__restrict__ wmax_bits, int *__restrict__ redo_out, int duty) {
    constexpr int NT = MCPM_TILE * MCPM_TILE * (MCPM_TILE + 2);      // z-padded
    __shared__ u64 tile[2 * NT];
    __shared__ int flagged;
    __shared__ int sh27[27];
    __shared__ int sus[MCPM_SUS + 2];
    paint3_tile_body<false, 1024, 2, FAST>(g, disp, w3, mesh, M, accumulate, L, wmax_bits, redo_out, -1, duty, tile, flagged, sh27, sus);
}

// buckets of the three-component paint: int64 fixed point with the max|w| 2^-28 scale, doubles when the weights are non-finite
__device__ __forceinline__ void paint3_bucket_body(const Geom &g, const float *__restrict__ disp, const float *__restrict__ w3,
                                                   float *__restrict__ mesh, int64_t M, const TileLists &L,
                                                   const unsigned *__restrict__ wmax_bits, u64 *tile, const int bid, const int nblk) {
    constexpr int B = MCPM_TILE, NT = B * B * B;
    const int ntl = L.cnts[C_NTILES];
    if (bid >= ntl) return;
    const TScale sc = tile_scale(wmax_bits);
    if (sc.mode == 0) return;
    const bool F64 = sc.mode == 2;      // non-finite weights: doubles (uniform over the launch)
    double *dtile = reinterpret_cast<double *>(tile);
    const int ntz = g.nz / B, nty = g.ny / B;
    for (int it = bid; it < ntl; it += nblk) {
        const int t = L.nonempty[it];
        const int cnt = L.bcnt[t];
        if (cnt > L.cap) continue;
        const int x0 = (t / (ntz * nty)) * B, y0 = ((t / ntz) % nty) * B, z0 = (t % ntz) * B;
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * NT; i += 256) tile[i] = 0ull;
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += 256) {
            const int gi = L.bucket[(int64_t)t * L.cap + k];
            const P3 d = load3(disp, gi);
            int cx, cy, cz;
            if (!bucket_cell(g, gi, d, x0, y0, z0, cx, cy, cz)) continue;
            const P3 wt = load3(w3, gi);
            const float tx1 = d.x - floorf(d.x), ty1 = d.y - floorf(d.y), tz1 = d.z - floorf(d.z);
            const float kx[2] = {1.f - tx1, tx1}, ky[2] = {1.f - ty1, ty1}, kz[2] = {1.f - tz1, tz1};
            const float s = F64 ? 1.f : sc.S;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int x = cx + a, y = cy + bb, z = cz + e;
                        if ((unsigned)x < (unsigned)B && (unsigned)y < (unsigned)B && (unsigned)z < (unsigned)B) {
                            const int q = (x * B + y) * B + z;
                            const float k3 = kx[a] * ky[bb] * kz[e];
                            if (F64) {
                                atomicAdd(dtile + q, (double)(wt.x * k3));
                                atomicAdd(dtile + NT + q, (double)(wt.y * k3));
                                atomicAdd(dtile + 2 * NT + q, (double)(wt.z * k3));
                            } else {
                                atomicAdd(tile + q, (u64)(long long)cvt_rpi(wt.x * s * k3));
                                atomicAdd(tile + NT + q, (u64)(long long)cvt_rpi(wt.y * s * k3));
                                atomicAdd(tile + 2 * NT + q, (u64)(long long)cvt_rpi(wt.z * s * k3));
                            }
                        }
                    }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * NT; i += 256) {
            const u64 a = tile[i];
            if (a == 0ull) continue;
            const int cc = i / NT, ii = i - cc * NT;
            const int lz = ii % B, r = ii / B, ly = r % B, lx = r / B;
            const float v = F64 ? (float)dtile[i] : (float)((double)(long long)a * sc.Sinv);
            mesh[cc * M + ((int64_t)(x0 + lx) * g.ny + (y0 + ly)) * g.nz + z0 + lz] += v;
        }
        if (threadIdx.x == 0) atomicAdd(L.cnts + C_BUCKETED, cnt);
    }
}

// Epilogue of the three-component paint, one launch: buckets | leftovers (see paint_epilogue_kernel).  The last block to
// finish also clears the max|w| slots: they are zero between paints, so the kernel that produces the next weights (the
// adjoint particle kernel, axpby) commits its maximum without a memset launch in front of it.
__global__ __launch_bounds__(256) void paint3_epilogue_kernel(Geom g, const float *__restrict__ disp, const float *__restrict__ w3,
                                                              float *__restrict__ mesh, int64_t M, TileLists L,
                                                              unsigned *__restrict__ wmax_bits, int nbk, unsigned *__restrict__ done) {
    __shared__ u64 tile[3 * MCPM_TILE * MCPM_TILE * MCPM_TILE];
    __shared__ int last;
    if ((int)blockIdx.x < nbk) paint3_bucket_body(g, disp, w3, mesh, M, L, wmax_bits, tile, (int)blockIdx.x, nbk);
    else {
        const int bid = (int)blockIdx.x - nbk, nblk = (int)gridDim.x - nbk;
        paint_leftover_body<3>(g, disp, w3, 3, 0.f, mesh, M, L, bid, nblk);
    }
    // The bucket blocks that found work are the only readers of the max|w| slots in this launch (tile_scale at their top): the
    // last of THEM clears the slots (every block counting on one address cost 30 us: 1088 serialised atomics); with no reader
    // at all the first leftover block does it.
    const int nread = min(L.cnts[C_NTILES], nbk);
    const bool reader = (int)blockIdx.x < nread, lone = nread == 0 && (int)blockIdx.x == nbk;
    if (!reader && !lone) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = lone || atomicAdd(done, 1u) == (unsigned)nread - 1u;
    }
    __syncthreads();
    if (last) {
        for (int i = threadIdx.x; i < MCPM_FX_SLOTS; i += 256) wmax_bits[i * MCPM_FX_STRIDE] = 0u;
        if (threadIdx.x == 0) *done = 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
static bool tiled_geometry_ok(const mcpm_plan *p, const void *mesh) {
    const Geom &g = p->g;
    if (!g.same_lattice) return false;
    if (g.nx % MCPM_TILE || g.ny % MCPM_TILE || g.nz % MCPM_TILE) return false;
    // three tiles per axis at least: a tile and the window of its neighbour must not meet around the periodic box
    if (g.nx < 3 * MCPM_TILE || g.ny < 3 * MCPM_TILE || g.nz < 3 * MCPM_TILE) return false;
    if (((uintptr_t)mesh) & 15) return false;
    return p->bucket != nullptr;
}

// the FAST instantiations of the tile kernels (add-and-carry window walk, shift-built indices, 32-bit byte offsets): 1 = periodic
// power-of-two meshes, 2 = slab plans whose y and z axes are powers of two (x: the ghost-extended local planes, not periodic);
// 0 = the generic walk (MCPM_PAINT_FAST=0 forces it: A/B runs)
static int tiled_fast(const mcpm_plan *p) {
    static const int on = [] { const char *e = getenv("MCPM_PAINT_FAST"); return e ? atoi(e) : 1; }();
    const Geom &g = p->g;
    auto pow2 = [](int n) { return (n & (n - 1)) == 0; };
    // (12-byte records addressed with 32-bit byte offsets: load3w)
    if (!on || !pow2(g.ny) || !pow2(g.nz) || p->Np * 12 >= ((int64_t)1 << 32)) return 0;
    if (g.xslab) return 2;
    return pow2(g.nx) ? 1 : 0;
}

static unsigned wide_max_tiles() {   // meshes of at most this many tiles paint with 1024 threads per tile (MCPM_PAINT_WIDE_MAX_TILES)
    static const unsigned v = [] { const char *e = getenv("MCPM_PAINT_WIDE_MAX_TILES"); return e ? (unsigned)atoi(e) : 1024u; }();
    return v;
}

static int tile_order() {
    static const int o = [] { const char *e = getenv("MCPM_TILE_ORDER"); return e ? atoi(e) : 1; }();
    return o;
}

// The halo of the next tiled paint: p->halo if the caller fixed one (mcpm_plan_set_halo, MCPM_PAINT_HALO), else 0 = every tile's window
// chosen on the device for every input by box_tile_kernel (plans of 2048 tiles or more: MCPM_PAINT_ADAPT_MIN_TILES; with
// MCPM_PAINT_ADAPT=0 or on a smaller mesh: the static rule of plan.hip)
static int halo_of(const mcpm_plan *p) {
    static const int adapt = [] { const char *e = getenv("MCPM_PAINT_ADAPT"); return e ? atoi(e) : 1; }();
    if (p->halo > 0) return p->halo;
    // below ~2000 tiles every tile is resident at once and a tile's time is latency, not visits: the extra visits of a wide
    // halo are free there and the particles a narrow one misses are not (128^3: 3511 steps/s chosen per input, 3597 at H = 4)
    const int64_t ntiles = p->M / (MCPM_TILE * MCPM_TILE * MCPM_TILE);
    static const int slab_too = [] { const char *e = getenv("MCPM_PAINT_ADAPT_SLAB"); return e ? atoi(e) : 1; }();
    // (the per-tile halos ride in the offset words of centred windows: p->centre)
    static const int min_tiles = [] { const char *e = getenv("MCPM_PAINT_ADAPT_MIN_TILES"); return e ? atoi(e) : 2048; }();
    return (adapt && (slab_too || !p->g.xslab) && p->halo_sel && p->centre && ntiles >= min_tiles) ? 0 : mcpm_default_halo(p->M);
}

static TileLists tile_lists(const mcpm_plan *p) {
    const int64_t ntiles = p->M / (MCPM_TILE * MCPM_TILE * MCPM_TILE);
    const int h = halo_of(p);
    return TileLists{p->centre ? p->tile_off : nullptr, p->centre ? p->halo_sel + 2 * ntiles : nullptr, p->bucket_cnt, p->bucket, p->bucket_cap, p->bucket_tiles, p->outliers, p->outliers + p->Np, (int)(p->Np < (1 << 30) ? p->Np : (1 << 30)), p->outlier_count, tile_order(), h > 0 ? h : mcpm_default_halo(p->M)};
}

// p->halo_sel: [ntiles] minima and [ntiles] maxima of the blocks' sampled floor(d), then [ntiles] upper corners of the tiles' windows
static void tiled_prologue(mcpm_plan *p, const float *pos, int *redo = nullptr) {
    const Geom &g = p->g;
    const int ntiles = (g.nx / MCPM_TILE) * (g.ny / MCPM_TILE) * (g.nz / MCPM_TILE);
    const int h = halo_of(p);
    const bool adapt = h == 0;
    int *thi = p->centre ? p->halo_sel + 2 * (int64_t)ntiles : nullptr;
    tile_prologue_kernel<<<(ntiles + 3) / 4, 256, 0, p->stream>>>(g, pos, p->centre ? p->tile_off : nullptr, thi, p->bucket_cnt, p->outlier_count,
                                                                  ntiles, 8, redo, adapt ? p->halo_sel : nullptr, h);
    // (window corners stay within +-12, what an offset of 8 and a halo of 4 reached: the conditional wraps of the generic walk
    // assume a window point is less than one period away, and 48-cell meshes are tiled)
    if (adapt) box_tile_kernel<<<(ntiles + 15) / 16, 256, 0, p->stream>>>(g, p->tile_off, thi, p->halo_sel, ntiles, 12);
}

// Tiled density paint if the geometry allows; returns false if the caller must use the generic path.
bool mcpm_paint_tiled(mcpm_plan *p, const float *pos, const float *w, int64_t wstride, float wscalar, float *mesh, int accumulate) {
    if (!tiled_geometry_ok(p, mesh)) return false;
    const Geom &g = p->g;
    const unsigned nb = (unsigned)((g.nx / MCPM_TILE) * (g.ny / MCPM_TILE) * (g.nz / MCPM_TILE));
    tiled_prologue(p, pos);
    const TileLists L = tile_lists(p);
    const int fast = tiled_fast(p);
    // bucket blocks walk the list of non-empty buckets (grid-stride); every block of the epilogue launch is scheduled with the
    // bucket tile's LDS whether it finds work or not, so small meshes (few non-empty buckets) get few of them
    const unsigned nbk = nb <= 4096u ? (nb < 256u ? nb : 256u) : 1024u;
    // leftover blocks: grid-stride over the wild list (normally empty) and, only after a bucket overflow, over the particles.
    // Few: every block of the launch carries the bucket blocks' LDS tile, so 1024 of them cost more than the launch they save
    const unsigned nlo = 64u;
    if (w) {
        (void)hipMemsetAsync(p->gx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream);
        absmax_kernel<<<2048, 256, 0, p->stream>>>(w, wstride, p->Np, p->gx_wmax);
        // fixed point, and doubles for non-finite weights: the instantiation the weights do not call for returns at once
#define CALLW(FAST_)                                                                                                                  \
    {                                                                                                                                 \
        paint_tile_kernel<1, 512, 4, FAST_><<<nb, 512, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate, L, p->gx_wmax, 1); \
        paint_tile_kernel<2, 512, 4, FAST_><<<nb, 512, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate, L, p->gx_wmax, 1); \
    }
        if (fast == 1) CALLW(1) else if (fast == 2) CALLW(2) else CALLW(0)
#undef CALLW
    } else {
        // meshes of at most 1024 tiles (128^3: 512) leave half of the CUs' wave slots empty with 512 threads per tile, and a tile's
        // time is latency there: 1024 threads per tile halve it (two such workgroups still fit a CU: 49 VGPRs, 37 KB of LDS)
        const bool wide = fast == 1 && nb <= wide_max_tiles();
        if (wide) paint_tile_kernel<0, 1024, 4, 1><<<nb, 1024, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate, L, p->gx_wmax, 1);
        else if (fast == 1) paint_tile_kernel<0, 512, 4, 1><<<nb, 512, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate, L, p->gx_wmax, 1);
        else if (fast == 2) paint_tile_kernel<0, 512, 4, 2><<<nb, 512, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate, L, p->gx_wmax, 1);
        else paint_tile_kernel<0, 512, 4><<<nb, 512, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, accumulate, L, p->gx_wmax, 1);
    }
    coverage_duty_kernel<<<1024, 256, 0, p->stream>>>(g, pos, L);
    if (w) paint_epilogue_kernel<1><<<nbk + nlo, 256, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, p->M, L, p->gx_wmax, (int)nbk);
    else paint_epilogue_kernel<0><<<nbk + nlo, 256, 0, p->stream>>>(g, pos, w, wstride, wscalar, mesh, p->M, L, p->gx_wmax, (int)nbk);
    return true;
}

// Tiled three-component paint; returns false if the caller must paint the components one by one.
bool mcpm_paint3_tiled(mcpm_plan *p, const float *pos, const float *weights3, float *meshes3, int accumulate) {
    if (!tiled_geometry_ok(p, meshes3) || ((p->M * 4) & 15) || p->paint3_variant < 0) return false;
    const Geom &g = p->g;
    const unsigned nb = (unsigned)((g.nx / MCPM_TILE) * (g.ny / MCPM_TILE) * (g.nz / MCPM_TILE));
    if (p->fx_tiles < (int)nb) return false;
    tiled_prologue(p, pos, p->fx_redo);
    const TileLists L = tile_lists(p);
    const int fast = tiled_fast(p);
    const unsigned nbk = nb <= 4096u ? (nb < 256u ? nb : 256u) : 1024u;   // see mcpm_paint_tiled
    if (p->fx_src != weights3) {   // max|w| not left behind by the kernel that produced the weights
        if (!p->fx_clean) (void)hipMemsetAsync(p->fx_wmax, 0, sizeof(unsigned) * MCPM_FX_SLOTS * MCPM_FX_STRIDE, p->stream);
        absmax_kernel<<<2048, 256, 0, p->stream>>>(weights3, 1, 3 * p->Np, p->fx_wmax);
    }
    p->fx_src = nullptr;
    if (p->paint3_variant == 4) {   // fixed-point tiles; the tiles they flag (and every tile if max|w| is unusable) in f64
        if (fast == 1 && nb <= wide_max_tiles()) paint3_tile_wide_kernel<1><<<nb, 1024, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, L, p->fx_wmax, p->fx_redo, 1);
        else if (fast == 1) paint3_tile_kernel<false, 512, 4, 1><<<nb, 512, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, L, p->fx_wmax, p->fx_redo, nullptr, 1);
        else if (fast == 2) paint3_tile_kernel<false, 512, 4, 2><<<nb, 512, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, L, p->fx_wmax, p->fx_redo, nullptr, 1);
        else paint3_tile_kernel<false, 512, 4><<<nb, 512, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, L, p->fx_wmax, p->fx_redo, nullptr, 1);
        paint3_tile_kernel<true, 1024, 4><<<nb < 256u ? nb : 256u, 1024, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, L, p->fx_wmax, nullptr, p->fx_redo, 0);
    } else {   // f64 tiles everywhere (A/B and tests)
        paint3_tile_kernel<true, 1024, 4><<<nb, 1024, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, accumulate, L, p->fx_wmax, nullptr, nullptr, 1);
    }
    coverage_duty_kernel<<<1024, 256, 0, p->stream>>>(g, pos, L);
    const unsigned nlo = 64u;     // see mcpm_paint_tiled
    // buckets | leftovers in one launch; its last block leaves the max|w| slots zero for the next producer
    paint3_epilogue_kernel<<<nbk + nlo, 256, 0, p->stream>>>(g, pos, weights3, meshes3, p->M, L, p->fx_wmax, (int)nbk,
                                                             (unsigned *)(p->outlier_count + 7));
    p->fx_clean = 1;
    return true;
}
