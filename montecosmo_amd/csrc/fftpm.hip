// Hand-written FFT Poisson solve of libmcpm.so for gfx950: real density -> three real force meshes, and its
// adjoint, in five HBM passes instead of rocFFT's twelve plus a k-space kernel:
//
//     z R2C  ->  y column FFT  ->  [x column FFT . k-space multiply . inverse x column FFT] (1 -> 3 spectra)
//            ->  3 x inverse y column FFT  ->  3 x z C2R
//
// Replaces `rfftn` / `invlaplace_hat * gradient_hat` / 3 x `irfftn` of montecosmo/nbody.py:589-603 (and the
// transpose chain for the VJP).  Power-of-two axes 64..1024 only; other sizes use the rocFFT path (plan.hip).
//
// Building block `fft_line`: a length-N complex Stockham FFT held 8 points per thread (T = N/8 threads per
// line), radix-8 stages exchanged through LDS, an optional last radix-2/4 stage.  Input and output share one
// register layout (thread u holds points u + T*m), so a forward and an inverse transform chain without
// reshuffling -- that is what lets the k-space multiply sit between them in registers.
//   * z passes: lines are contiguous; two real lines ride one complex FFT (a + i b), split / merged with the
//     Hermitian mirror through LDS.
//   * y / x passes: a workgroup owns 16 adjacent kz columns of one (x or y) plane so every global access is a
//     128-byte row segment; lines sit "line-fastest" in LDS, which keeps the exchanges bank-conflict free.
// Spectra live in a padded internal layout [nx][ny][nzp], nzp = nz/2 + 16, so that those segments are aligned.
#include "mcpm_internal.h"

#include "fft_dev.h"

// ------------------------------------------------------------------------------------------------
// z passes: contiguous lines, two real lines per complex transform
struct FGeom {
    int nx, ny, nz, nzh, nzp;  // GLOBAL mesh; nzp: padded complex pitch of the internal spectra
};
// batched z lines: mesh b starts at real + b*real_bstride (floats) / spec + b*spec_bstride (complex); `lines` per mesh
struct ZBatch {
    int64_t lines, real_bstride, spec_bstride;
};
// y-line addressing of a column pass: point y of x-plane xl of spectrum c sits at
// c*BS + xl*PS + (y / YB)*SB + (y % YB)*nzp (complex).  Plain layout [c][xl][y][nzp]: YB = ny, SB = 0.
// All-to-all (transposed-order) layout [c][dest rank][xl][y_local][nzp]: YB = ny/ranks, SB = one rank block.
// Chunked all-to-all layout (mcpm_slab_set_chunks, C chunks of CW = nxl / C planes, so that a transpose is C all-to-alls
// that overlap the passes of the other chunks): [c][chunk w][dest rank][xlw][y_local][nzp]; plane xl then sits at
// (xl / CW) * WS + (xl % CW) * PS with WS = ranks * SB and SB = CW * YB * nzp.  C = 1 is the layout above.
struct YLayout {
    int64_t BS, PS, SB;
    int YB, NXL, lgYB;  // YB is a power of two
    int X0, NXW;        // the pass covers local planes [X0, X0 + NXW)
    int64_t WS;         // chunk stride (0 when not chunked)
    int lgCW, CW;       // planes per chunk (power of two); CW = NXL when not chunked
    __device__ __forceinline__ int64_t plane(int xl) const { return (int64_t)(xl >> lgCW) * WS + (int64_t)(xl & (CW - 1)) * PS; }
};
// x-line addressing of the fused pass.  One-spectrum side: [x][yl][nzp] (NYL rows per x).  Three-spectra side:
// c*SC + (x / XB)*SBx + ((x % XB)*NYL + yl)*nzp  (single GPU: XB = nx; slabs: [c][src/dest rank][xl][yl][nzp]).
// With C chunks (see YLayout) the multi-rank side is [c][chunk w][rank][xlw][yl][nzp]:
// c*SC + ((x % XB) / CW) * WS + (x / XB) * SBx + (((x % XB) % CW) * NYL + yl) * nzp, SBx = CW * NYL * nzp, WS = ranks * SBx.
struct XLayout {
    int NYL, iy0, XB, lgXB;  // XB is a power of two
    int64_t SBx, SC;
    int remap;               // XCD-aware block order (col_block)
    int lgCW, CW;            // planes per chunk (CW = XB when not chunked)
    uint32_t WS;
    // element offset of (x, yl) of one spectrum in the rank-blocked layout, without the kz term
    __device__ __forceinline__ uint32_t off(int x, uint32_t xs) const {
        const int xloc = x & (XB - 1);
        return (uint32_t)(xloc >> lgCW) * WS + (uint32_t)((x >> lgXB) * SBx) + (uint32_t)(xloc & (CW - 1)) * xs;
    }
};
// Workgroup -> (kz block, y row) of the x passes.  Hardware hands consecutive workgroups to the 8 XCDs in turn, and each XCD
// has its own L2: with 8 kz columns per workgroup a row segment is 64 bytes, HALF a 128-byte line, so in launch order the two
// workgroups that share every line of their columns run on different XCDs and the line is fetched from HBM twice
// (profiles/r01_pmc_traffic.json: 2.4 GB moved by the fused x pass for 1.6 GB of spectra).  Remapped, each XCD works a
// contiguous run of (y row, kz block) pairs, so the neighbour's half line is an L2 hit.
__device__ __forceinline__ void col_block(int remap, unsigned &bx, unsigned &by) {
    bx = blockIdx.x, by = blockIdx.y;
    if (remap) {
        const unsigned nb = gridDim.x * gridDim.y, b = by * gridDim.x + bx;
        const unsigned v = (b & 7u) * (nb >> 3) + (b >> 3);
        by = v / gridDim.x;
        bx = v - by * gridDim.x;
    }
}

template <int N>
__global__ __launch_bounds__(256) void zfwd_kernel(FGeom g, const float *__restrict__ real, cf *__restrict__ spec,
                                                   const cf *__restrict__ W, int64_t npairs, ZBatch zb) {
    constexpr int T = FftShape<N>::T, PAIRS = 256 / T;
    typedef Tile<N, PAIRS, false> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int pl = threadIdx.x / T, u = threadIdx.x - pl * T;
    const int64_t pair = (int64_t)blockIdx.x * PAIRS + pl;
    const bool ok = pair < npairs;
    const int64_t line = 2 * pair, bi = line / zb.lines, lm = line - bi * zb.lines;
    const float *a = real + bi * zb.real_bstride + lm * N, *b = a + N;
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = ok ? mkc(a[u + T * m], b[u + T * m]) : mkc(0.f, 0.f);
    TL tile{pl};
    fft_line<N, -1>(v, lds, W, u, tile);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) lds[tile(u + T * m)] = v[m];
    __syncthreads();
    if (!ok) return;
    cf *oa = spec + bi * zb.spec_bstride + lm * g.nzp, *ob = oa + g.nzp;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k = u + T * m;
        const cf z = v[m], zm = lds[tile((N - k) & (N - 1))];
        NTSTORE(mkc(0.5f * (z.x + zm.x), 0.5f * (z.y - zm.y)), &oa[k]);
        NTSTORE(mkc(0.5f * (z.y + zm.y), 0.5f * (zm.x - z.x)), &ob[k]);
    }
    if (u == 0) {  // Nyquist: Z[N/2] is its own mirror
        const cf z = v[4];
        oa[N / 2] = mkc(z.x, 0.f);
        ob[N / 2] = mkc(z.y, 0.f);
    }
}

template <int N>
__global__ __launch_bounds__(256) void zinv_kernel(FGeom g, const cf *__restrict__ spec, float *__restrict__ real,
                                                   const cf *__restrict__ W, int64_t npairs, ZBatch zb) {
    constexpr int T = FftShape<N>::T, PAIRS = 256 / T;
    typedef Tile<N, PAIRS, false> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int pl = threadIdx.x / T, u = threadIdx.x - pl * T;
    const int64_t pair = (int64_t)blockIdx.x * PAIRS + pl;
    const bool ok = pair < npairs;
    const int64_t line = 2 * pair, bi = line / zb.lines, lm = line - bi * zb.lines;
    const cf *ia = spec + bi * zb.spec_bstride + lm * g.nzp, *ib = ia + g.nzp;
    TL tile{pl};
    // Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k]); imaginary parts of k = 0 and N/2 are ignored (c2r)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k = u + T * m;
        cf A = ok ? ia[k] : mkc(0.f, 0.f), B = ok ? ib[k] : mkc(0.f, 0.f);
        if (k == 0) {
            lds[tile(0)] = mkc(A.x, B.x);
        } else {
            lds[tile(k)] = mkc(A.x - B.y, A.y + B.x);
            lds[tile(N - k)] = mkc(A.x + B.y, B.x - A.y);
        }
    }
    if (u == 0) {
        cf A = ok ? ia[N / 2] : mkc(0.f, 0.f), B = ok ? ib[N / 2] : mkc(0.f, 0.f);
        lds[tile(N / 2)] = mkc(A.x, B.x);
    }
    __syncthreads();
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = lds[tile(u + T * m)];
    fft_line<N, +1>(v, lds, W, u, tile);
    if (!ok) return;
    float *a = real + bi * zb.real_bstride + lm * N, *b = a + N;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        a[u + T * m] = v[m].x;
        b[u + T * m] = v[m].y;
    }
}

// z C2R of the three force spectra of one line pair, written interleaved [x][y][z][3] (one 12-byte store per cell):
// the layout the step kernels gather from (particles_dev.h interp3)
// NTOUT: streaming 12-byte stores, for callers whose consumer comes after the mesh has left the caches anyway (pm_forces: the
// gathers of the read start at the other end of a 1.6 GB mesh); the steppers keep plain stores (measured: no gain there)
template <int N, bool NTOUT = false>
__global__ __launch_bounds__(256) void zinv3_il_kernel(FGeom g, const cf *__restrict__ spec, float *__restrict__ real,
                                                      const cf *__restrict__ W, int64_t npairs, int64_t spec_cstride) {
    constexpr int T = FftShape<N>::T, PAIRS = 256 / T;
    typedef Tile<N, PAIRS, false> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int pl = threadIdx.x / T, u = threadIdx.x - pl * T;
    const int64_t pair = (int64_t)blockIdx.x * PAIRS + pl;
    const bool ok = pair < npairs;
    const int64_t line = 2 * pair;
    TL tile{pl};
    cf r[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const cf *ia = spec + c * spec_cstride + line * g.nzp, *ib = ia + g.nzp;
        __syncthreads();  // the previous component's last exchange has been read by every thread
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = u + T * m;
            cf A = ok ? ia[k] : mkc(0.f, 0.f), B = ok ? ib[k] : mkc(0.f, 0.f);
            if (k == 0) {
                lds[tile(0)] = mkc(A.x, B.x);
            } else {
                lds[tile(k)] = mkc(A.x - B.y, A.y + B.x);
                lds[tile(N - k)] = mkc(A.x + B.y, B.x - A.y);
            }
        }
        if (u == 0) {
            cf A = ok ? ia[N / 2] : mkc(0.f, 0.f), B = ok ? ib[N / 2] : mkc(0.f, 0.f);
            lds[tile(N / 2)] = mkc(A.x, B.x);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) r[c][m] = lds[tile(u + T * m)];
        fft_line<N, +1>(r[c], lds, W, u, tile);
    }
    if (!ok) return;
    struct __attribute__((packed, aligned(4))) F3 {
        float a, b, c;
    };
    F3 *a = reinterpret_cast<F3 *>(real) + line * N, *b = a + N;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (NTOUT) {
            typedef float f3v __attribute__((ext_vector_type(3)));
            const f3v va = {r[0][m].x, r[1][m].x, r[2][m].x}, vb = {r[0][m].y, r[1][m].y, r[2][m].y};
            // (the store-data hazard of > 8-byte stores, invisible to the compiler inside inline asm: the rule and its wait
            // states are at MCPM_STORE_DATA_HAZARD_NOP in particles_dev.h)
            asm volatile("global_store_dwordx3 %0, %1, off nt\n\t" MCPM_STORE_DATA_HAZARD_NOP : : "v"(&a[u + T * m]), "v"(va) : "memory");
            asm volatile("global_store_dwordx3 %0, %1, off nt\n\t" MCPM_STORE_DATA_HAZARD_NOP : : "v"(&b[u + T * m]), "v"(vb) : "memory");
        } else {
            a[u + T * m] = F3{r[0][m].x, r[1][m].x, r[2][m].x};
            b[u + T * m] = F3{r[0][m].y, r[1][m].y, r[2][m].y};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// column passes (y or x): 16 adjacent kz columns per workgroup
template <int N, int ML = 16>
struct ColShape {
    static constexpr int T = FftShape<N>::T;
    static constexpr int LINES = (1024 / T) < ML ? (1024 / T) : ML;
    static constexpr int THREADS = T * LINES;
};

// FFT along y for every (plane, kz); in == out allowed when both layouts are equal
template <int N, int SIGN>
__global__ __launch_bounds__(ColShape<N>::THREADS) void ycol_kernel(FGeom g, const cf *__restrict__ in, cf *__restrict__ out,
                                                                    YLayout li, YLayout lo, const cf *__restrict__ W) {
    constexpr int T = ColShape<N>::T, LINES = ColShape<N>::LINES;
    typedef Tile<N, LINES, true> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int l = threadIdx.x % LINES, u = threadIdx.x / LINES;
    const int kz = blockIdx.x * LINES + l;
    const bool ok = kz < g.nzh;
    const int pc = blockIdx.y / li.NXW, pxl = li.X0 + (blockIdx.y - pc * li.NXW);
    const cf *ib = in + pc * li.BS + li.plane(pxl) + kz;
    cf *ob = out + pc * lo.BS + lo.plane(pxl) + kz;
    cf v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int y = u + T * m, yb = y >> li.lgYB;
        v[m] = ok ? ib[yb * li.SB + (int64_t)(y & (li.YB - 1)) * g.nzp] : mkc(0.f, 0.f);
    }
    TL tile{l};
    fft_line<N, SIGN>(v, lds, W, u, tile);
    if (!ok) return;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int y = u + T * m, yb = y >> lo.lgYB;
        if (SIGN < 0) ob[yb * lo.SB + (int64_t)(y & (lo.YB - 1)) * g.nzp] = v[m];      // read next by the x pass
        else NTSTORE(v[m], &ob[yb * lo.SB + (int64_t)(y & (lo.YB - 1)) * g.nzp]);
    }
}

#define MCPM_TWO_PI 6.283185307179586f

// y pass that also applies the y / z force factors, so that only TWO spectra travel between the y and x passes:
// the x pass produces A = IFFTx(-i kx L X) and G = IFFTx(-i L X), and the force spectra are A, ky G, kz G
// (ky = 0 at its Nyquist on the kz = 0 / Nyquist planes, kz = 0 at its Nyquist: the Hermitian projection).
//   EXPAND  (inverse): in = {A, G} -> out = {IFFTy A, IFFTy(ky G), kz IFFTy G}
//   CONTRACT (forward, adjoint): in = {a, b, c} -> out = {FFTy a, ky FFTy b + kz FFTy c}
template <int N, bool EXPAND, int ML>
__global__ __launch_bounds__((ColShape<N, ML>::THREADS)) void ycol2_kernel(
    FGeom g, const cf *__restrict__ in, cf *__restrict__ out, YLayout li, YLayout lo, const cf *__restrict__ W, int parts) {
    constexpr int T = ColShape<N, ML>::T, LINES = ColShape<N, ML>::LINES;
    typedef Tile<N, LINES, true> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int l = threadIdx.x % LINES, u = threadIdx.x / LINES;
    const int kzi = blockIdx.x * LINES + l;
    const bool ok = kzi < g.nzh;
    const int plane = li.X0 + blockIdx.y;                    // local x plane; component c adds c * BS
    const cf *ib = in + li.plane(plane) + kzi;
    cf *ob = out + lo.plane(plane) + kzi;
    const bool special = (kzi == 0) || (kzi == g.nz / 2);
    const float fz = (kzi == g.nz / 2) ? 0.f : MCPM_TWO_PI * (float)kzi / (float)g.nz;
    // one transform at a time (load, FFT, store) keeps the kernel at two workgroups per CU; offsets are recomputed
    // (a shift, a mask, a multiply) rather than held in registers
#define ioff(m) ((uint32_t)((u + T * (m)) >> li.lgYB) * (uint32_t)li.SB + (uint32_t)((u + T * (m)) & (li.YB - 1)) * (uint32_t)g.nzp)
#define ooff(m) ((uint32_t)((u + T * (m)) >> lo.lgYB) * (uint32_t)lo.SB + (uint32_t)((u + T * (m)) & (lo.YB - 1)) * (uint32_t)g.nzp)
#define fy(m) ((special && (u + T * (m)) == N / 2) ? 0.f : MCPM_TWO_PI * (float)((u + T * (m)) < N / 2 ? (u + T * (m)) : (u + T * (m)) - N) / (float)N)
    const cf zero = mkc(0.f, 0.f);
    TL tile{l};
    cf v[8];
    // parts (uniform): bit 0 = the spectrum-0 transform, bit 1 = the other two (they travel in separate all-to-alls)
    if (EXPAND) {
        if (parts & 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = ok ? ib[ioff(m)] : zero;
            fft_line<N, +1>(v, lds, W, u, tile);
            if (ok) {
#pragma unroll
                for (int m = 0; m < 8; ++m) NTSTORE(v[m], &ob[ooff(m)]);
            }
        }
        if (!(parts & 2)) return;
        cf gq[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            gq[m] = ok ? ib[li.BS + ioff(m)] : zero;
            const float f = fy(m);
            v[m] = mkc(f * gq[m].x, f * gq[m].y);
        }
        fft_line<N, +1>(v, lds, W, u, tile);
        if (ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) NTSTORE(v[m], &ob[lo.BS + ooff(m)]);
        }
        fft_line<N, +1>(gq, lds, W, u, tile);
        if (ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) NTSTORE(mkc(fz * gq[m].x, fz * gq[m].y), &ob[2 * lo.BS + ooff(m)]);
        }
    } else {
        if (parts & 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = ok ? ib[ioff(m)] : zero;
            fft_line<N, -1>(v, lds, W, u, tile);
            if (ok) {
#pragma unroll
                for (int m = 0; m < 8; ++m) ob[ooff(m)] = v[m];
            }
        }
        if (!(parts & 2)) return;
        cf c[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            v[m] = ok ? ib[li.BS + ioff(m)] : zero;
            c[m] = ok ? ib[2 * li.BS + ioff(m)] : zero;
        }
        fft_line<N, -1>(v, lds, W, u, tile);
        fft_line<N, -1>(c, lds, W, u, tile);
        if (ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float f = fy(m);
                ob[lo.BS + ooff(m)] = mkc(f * v[m].x + fz * c[m].x, f * v[m].y + fz * c[m].y);
            }
        }
    }
#undef ioff
#undef ooff
#undef fy
}

// k-space multipliers (nbody.py:109-163 with fd_order = inf), wavevectors from indices
__device__ __forceinline__ float kfreq_i(int i, int n) {
    int s = (i < (n + 1) / 2) ? i : i - n;
    return MCPM_TWO_PI * (float)s / (float)n;
}

// MODE 0: in (1 spectrum) -> forward x FFT -> {-i kx L X, -i L X} -> inverse x FFT -> out (2 spectra: A and G)
// MODE 1: in (2 spectra a, b) -> forward x FFT -> i kx L FFTx(a) + i L FFTx(b) -> inverse x FFT -> out (1 spectrum)
// (L = -scale/k^2; the y pass applies ky, kz: ycol2_kernel.)
// The multiplier of component c at mode (kx, ky, kz) is s_c * (-i), s_c = k_c * (-scale/k^2), Hermitian-projected:
// on the kz = 0 / Nyquist planes a component whose own index sits at Nyquist is dropped (kspace.hip header).
template <int N, int MODE, int ML>
__global__ __launch_bounds__((ColShape<N, ML>::THREADS)) void xfused_kernel(FGeom g, const cf *__restrict__ in, cf *__restrict__ out,
                                                                          XLayout xl, float scale, const cf *__restrict__ W) {
    constexpr int T = ColShape<N, ML>::T, LINES = ColShape<N, ML>::LINES;
    typedef Tile<N, LINES, true> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int l = threadIdx.x % LINES, u = threadIdx.x / LINES;
    unsigned bx, by;
    col_block(xl.remap, bx, by);
    const int kzi = bx * LINES + l, yl = by, iy = xl.iy0 + yl;
    const bool ok = kzi < g.nzh;
    const uint32_t off0 = (uint32_t)yl * g.nzp + kzi, xs = (uint32_t)xl.NYL * g.nzp;
    // element offsets (complex units, < 2^31): one-spectrum side and three-spectra side (without the c*SC term)
    uint32_t o1[8], o3[8];
    float sx[8], L[8];
    const float ky = kfreq_i(iy, g.ny), kz = MCPM_TWO_PI * (float)kzi / (float)g.nz;
    const bool special = (kzi == 0) || (kzi == g.nz / 2);
    const float dkx = MCPM_TWO_PI / (float)N;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        o1[m] = xl.off(x, xs) + off0;      // the one-spectrum side is in the same rank-blocked (and chunked) layout
        o3[m] = o1[m];
        const float kx = dkx * (float)(x < N / 2 ? x : x - N);
        const float kk = kx * kx + ky * ky + kz * kz;
        L[m] = kk == 0.f ? 0.f : -scale * __frcp_rn(kk);
        sx[m] = (special && x == N / 2) ? 0.f : kx * L[m];
    }
    TL tile{l};
    // The y and z multipliers are (ky, kz) * L(kx): constant factors of ONE x-dependent spectrum, so they share one
    // x transform -- two inverse (MODE 0) or forward (MODE 1) transforms instead of three.
    if (MODE == 0) {
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = ok ? in[o1[m]] : mkc(0.f, 0.f);
        fft_line<N, -1>(v, lds, W, u, tile);
        cf w[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) w[m] = mkc(sx[m] * v[m].y, -sx[m] * v[m].x);  // (a + i b)(-i s_x)
        fft_line<N, +1>(w, lds, W, u, tile);
        if (ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) NTSTORE(w[m], &out[o3[m]]);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) w[m] = mkc(L[m] * v[m].y, -L[m] * v[m].x);  // (a + i b)(-i L)
        fft_line<N, +1>(w, lds, W, u, tile);
        if (ok) {  // G: the y pass turns it into the y and z components (ycol2_kernel)
            cf *og = out + xl.SC;
#pragma unroll
            for (int m = 0; m < 8; ++m) NTSTORE(w[m], &og[o3[m]]);
        }
    } else {
        cf a[8], b[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {  // b = ky FFTy(f_bar_y) + kz FFTy(f_bar_z), formed by the y pass (ycol2_kernel)
            a[m] = ok ? in[o3[m]] : mkc(0.f, 0.f);
            b[m] = ok ? in[xl.SC + o3[m]] : mkc(0.f, 0.f);
        }
        fft_line<N, -1>(a, lds, W, u, tile);
        fft_line<N, -1>(b, lds, W, u, tile);
        cf acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m)  // (a + i b)(+i s): conj of the forward multipliers
            acc[m] = mkc(-sx[m] * a[m].y - L[m] * b[m].y, sx[m] * a[m].x + L[m] * b[m].x);
        fft_line<N, +1>(acc, lds, W, u, tile);
        if (ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) NTSTORE(acc[m], &out[o1[m]]);
        }
    }
}

// Spectrum-side variants of the x pass, for lpt and its adjoint (nbody.py:611-667): the input (or output) is a
// half-spectrum in the caller's plain layout [x][ny][nz/2+1], already (or still) in k-space along x, so one of the
// two x transforms disappears.
//   MODE 2: spectrum -> x (-(i k_c))(-1/k^2) scale -> inverse x FFT -> 3 spectra          (pm_forces on a spectrum)
//   MODE 3: spectrum -> x (i k_a)(i k_b)(-1/k^2) scale -> inverse x FFT -> 6 spectra     (Hessian, ab = 00 01 02 11 12 22)
//   MODE 4: 3 spectra -> forward x FFT -> sum_c conj(mult_c) * zw -> spectrum            (VJP of MODE 2)
//   MODE 5: 6 spectra -> forward x FFT -> sum_ab mult_ab * zw -> spectrum += ...         (VJP of MODE 3, accumulates)
// MODE 2/3 feed a C2R, so their multipliers are Hermitian-projected; MODE 4/5 return the exact cotangent of the numpy
// function: un-projected multipliers times the irfftn multiplicity zw = (1,2,..,2,1) (kspace.hip header).
template <int N, int MODE>
__global__ __launch_bounds__(ColShape<N>::THREADS) void xspec_kernel(FGeom g, const cf *__restrict__ in, cf *__restrict__ out,
                                                                     XLayout xl, float scale, const cf *__restrict__ W) {
    constexpr int T = ColShape<N>::T, LINES = ColShape<N>::LINES;
    constexpr int NC = (MODE == 2 || MODE == 4) ? 3 : 6;
    constexpr bool PROJECT = (MODE == 2 || MODE == 3);
    typedef Tile<N, LINES, true> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int l = threadIdx.x % LINES, u = threadIdx.x / LINES;
    unsigned bx, by;
    col_block(xl.remap, bx, by);
    const int kzi = bx * LINES + l, yl = by, iy = xl.iy0 + yl;
    const bool ok = kzi < g.nzh;
    const uint32_t off0 = (uint32_t)yl * g.nzp + kzi, xs = (uint32_t)xl.NYL * g.nzp;
    uint32_t os[8], o3[8];
    float kx[8], L[8];
    const float ky = kfreq_i(iy, g.ny), kz = MCPM_TWO_PI * (float)kzi / (float)g.nz;
    const bool special = (kzi == 0) || (kzi == g.nz / 2);
    const bool nyq_y = iy == g.ny / 2, nyq_z = kzi == g.nz / 2;
    const float zw = special ? 1.f : 2.f;
    const float dkx = MCPM_TWO_PI / (float)N;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        os[m] = ((uint32_t)x * g.ny + iy) * g.nzh + kzi;  // caller's plain layout
        o3[m] = xl.off(x, xs) + off0;
        kx[m] = dkx * (float)(x < N / 2 ? x : x - N);
        const float kk = kx[m] * kx[m] + ky * ky + kz * kz;
        L[m] = kk == 0.f ? 0.f : -scale * __frcp_rn(kk);
    }
    // real factor of component c at point m: force: s_c = k_c L (multiplier -i s_c); hessian: h_ab = -k_a k_b L
    auto factor = [&](int c, int m) -> float {
        const bool nyq_x = (u + T * m) == N / 2;
        if (NC == 3) {
            const float k = c == 0 ? kx[m] : (c == 1 ? ky : kz);
            const bool nq = c == 0 ? nyq_x : (c == 1 ? nyq_y : nyq_z);
            if (PROJECT && nq && (special || c == 2)) return 0.f;
            return k * L[m];
        }
        const int a = c < 3 ? 0 : (c < 5 ? 1 : 2), b = c < 3 ? c : (c < 5 ? c - 2 : 2);
        const float ka = a == 0 ? kx[m] : (a == 1 ? ky : kz), kb = b == 0 ? kx[m] : (b == 1 ? ky : kz);
        const bool na = a == 0 ? nyq_x : (a == 1 ? nyq_y : nyq_z), nb2 = b == 0 ? nyq_x : (b == 1 ? nyq_y : nyq_z);
        if (PROJECT && special && (na != nb2)) return 0.f;
        return -ka * kb * L[m];
    };
    TL tile{l};
    if (MODE == 2 || MODE == 3) {
        cf v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = ok ? in[os[m]] : mkc(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cf w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float f = factor(c, m);
                w[m] = NC == 3 ? mkc(f * v[m].y, -f * v[m].x) : mkc(f * v[m].x, f * v[m].y);
            }
            fft_line<N, +1>(w, lds, W, u, tile);
            if (ok) {
                cf *oc = out + c * xl.SC;
#pragma unroll
                for (int m = 0; m < 8; ++m) oc[o3[m]] = w[m];
            }
        }
    } else {
        cf acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = mkc(0.f, 0.f);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cf v[8];
            const cf *ic = in + c * xl.SC;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = ok ? ic[o3[m]] : mkc(0.f, 0.f);
            fft_line<N, -1>(v, lds, W, u, tile);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float f = factor(c, m) * zw;
                if (NC == 3) {  // conj(-i s) = +i s
                    acc[m].x += -f * v[m].y;
                    acc[m].y += f * v[m].x;
                } else {
                    acc[m].x += f * v[m].x;
                    acc[m].y += f * v[m].y;
                }
            }
        }
        if (ok) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                cf r = acc[m];
                if (MODE == 5) {
                    const cf o = out[os[m]];
                    r.x += o.x;
                    r.y += o.y;
                }
                out[os[m]] = r;
            }
        }
    }
}

// Plain x transform between the caller's half-spectrum layout [x][ny][nz/2+1] and the internal padded one (single-GPU
// plans): the third pass of a generic R2C (DIR = -1: padded -> forward x FFT -> plain) or the first of a generic C2R
// (DIR = +1: plain -> inverse x FFT -> padded).  mcpm_fft_r2c / mcpm_fft_c2r on power-of-two meshes.
template <int N, int DIR>
__global__ __launch_bounds__(ColShape<N>::THREADS) void xplain_kernel(FGeom g, const cf *__restrict__ in, cf *__restrict__ out,
                                                                      const cf *__restrict__ W) {
    constexpr int T = ColShape<N>::T, LINES = ColShape<N>::LINES;
    typedef Tile<N, LINES, true> TL;
    __shared__ cf lds[TL::FLOATS2];
    const int l = threadIdx.x % LINES, u = threadIdx.x / LINES;
    const int kzi = blockIdx.x * LINES + l, iy = blockIdx.y;
    const bool ok = kzi < g.nzh;
    cf v[8];
    TL tile{l};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        const uint32_t op = ((uint32_t)x * g.ny + iy) * g.nzh + kzi, oq = ((uint32_t)x * g.ny + iy) * g.nzp + kzi;
        v[m] = ok ? in[DIR > 0 ? op : oq] : mkc(0.f, 0.f);
    }
    fft_line<N, DIR>(v, lds, W, u, tile);
    if (!ok) return;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int x = u + T * m;
        const uint32_t op = ((uint32_t)x * g.ny + iy) * g.nzh + kzi, oq = ((uint32_t)x * g.ny + iy) * g.nzp + kzi;
        out[DIR > 0 ? oq : op] = v[m];
    }
}

// ------------------------------------------------------------------------------------------------
// host side
static bool pow2_ok(int n) { return n == 64 || n == 128 || n == 256 || n == 512 || n == 1024; }

bool mcpm_fftpm_supported(const mcpm_plan *p) {
    if (getenv("MCPM_DISABLE_FFTPM")) return false;
    if (p->nranks & (p->nranks - 1)) return false;  // slab blocks must stay powers of two (shift / mask indexing)
    return pow2_ok(p->nx_global) && pow2_ok(p->g.ny) && pow2_ok(p->g.nz);
}

static int ensure_twiddles(mcpm_plan *p) {
    if (p->tw[0]) return MCPM_OK;
    const int dims[3] = {p->nx_global, p->g.ny, p->g.nz};
    for (int a = 0; a < 3; ++a) {
        const int n = dims[a];
        std::vector<float> h(2 * (size_t)n);
        for (int j = 0; j < n; ++j) {
            const double ang = -2.0 * 3.14159265358979323846 * j / n;
            h[2 * j] = (float)cos(ang);
            h[2 * j + 1] = (float)sin(ang);
        }
        if (hipMalloc((void **)&p->tw[a], sizeof(float) * 2 * n) != hipSuccess) return mcpm_fail(p, MCPM_E_NOMEM, "twiddle table");
        MCPM_HIP(p, hipMemcpy(p->tw[a], h.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    }
    return MCPM_OK;
}

static FGeom fgeom(const mcpm_plan *p) { return FGeom{p->nx_global, p->g.ny, p->g.nz, p->g.nzh, p->g.nz / 2 + MCPM_NZPAD}; }
// one local spectrum: nxl planes x ny x nzp complex
static int64_t spec_elems(const mcpm_plan *p) { return (int64_t)p->nxl * p->g.ny * (p->g.nz / 2 + MCPM_NZPAD); }
static double pass_bytes(const mcpm_plan *p, int batch) {  // algorithmic share of one pass of a 3-pass transform
    return (double)batch * (4.0 * p->xwn * p->g.ny * p->g.nz + 8.0 * p->xwn * p->g.ny * p->g.nzh) / 3.0;
}

#define DISPATCH_N(n, CALL)                  \
    switch (n) {                             \
        case 64: CALL(64); break;            \
        case 128: CALL(128); break;          \
        case 256: CALL(256); break;          \
        case 512: CALL(512); break;          \
        default: CALL(1024); break;          \
    }

static int z_forward(mcpm_plan *p, const float *real, int64_t real_bstride, cf *spec, int batch) {
    const FGeom g = fgeom(p);
    const ZBatch zb{(int64_t)p->xwn * g.ny, real_bstride, spec_elems(p)};
    real += (int64_t)p->xw0 * g.ny * g.nz;   // window of local planes (all of them unless mcpm_slab_set_window)
    spec += (int64_t)p->xw0 * g.ny * g.nzp;
    const int64_t npairs = (int64_t)batch * zb.lines / 2;
    StageTimer st_(p, ST_R2C, pass_bytes(p, batch));
#define CALL(NN)                                                                                                   \
    {                                                                                                              \
        constexpr int PAIRS = 256 / (NN / 8);                                                                      \
        zfwd_kernel<NN><<<(unsigned)((npairs + PAIRS - 1) / PAIRS), 256, 0, p->stream>>>(g, real, spec, (const cf *)p->tw[2], npairs, zb); \
    }
    DISPATCH_N(g.nz, CALL)
#undef CALL
    MCPM_LAUNCH_CHECK(p, "zfwd_kernel");
    return MCPM_OK;
}

static int z_inverse(mcpm_plan *p, const cf *spec, float *real, int64_t real_bstride, int batch) {
    const FGeom g = fgeom(p);
    const ZBatch zb{(int64_t)p->xwn * g.ny, real_bstride, spec_elems(p)};
    real += (int64_t)p->xw0 * g.ny * g.nz;
    spec += (int64_t)p->xw0 * g.ny * g.nzp;
    const int64_t npairs = (int64_t)batch * zb.lines / 2;
    StageTimer st_(p, ST_C2R, pass_bytes(p, batch));
#define CALL(NN)                                                                                                   \
    {                                                                                                              \
        constexpr int PAIRS = 256 / (NN / 8);                                                                      \
        zinv_kernel<NN><<<(unsigned)((npairs + PAIRS - 1) / PAIRS), 256, 0, p->stream>>>(g, spec, real, (const cf *)p->tw[2], npairs, zb); \
    }
    DISPATCH_N(g.nz, CALL)
#undef CALL
    MCPM_LAUNCH_CHECK(p, "zinv_kernel");
    return MCPM_OK;
}

static int z_inverse3_il(mcpm_plan *p, const cf *spec3, float *real_il, bool nt_out = false) {
    const FGeom g = fgeom(p);
    const int64_t npairs = (int64_t)p->xwn * g.ny / 2;
    real_il += (int64_t)p->xw0 * g.ny * g.nz * 3;
    spec3 += (int64_t)p->xw0 * g.ny * g.nzp;
    StageTimer st_(p, ST_C2R, pass_bytes(p, 3));
#define CALL(NN)                                                                                                   \
    {                                                                                                              \
        constexpr int PAIRS = 256 / (NN / 8);                                                                      \
        if (nt_out) zinv3_il_kernel<NN, true><<<(unsigned)((npairs + PAIRS - 1) / PAIRS), 256, 0, p->stream>>>(g, spec3, real_il, (const cf *)p->tw[2], npairs, spec_elems(p)); \
        else zinv3_il_kernel<NN><<<(unsigned)((npairs + PAIRS - 1) / PAIRS), 256, 0, p->stream>>>(g, spec3, real_il, (const cf *)p->tw[2], npairs, spec_elems(p)); \
    }
    DISPATCH_N(g.nz, CALL)
#undef CALL
    MCPM_LAUNCH_CHECK(p, "zinv3_il_kernel");
    return MCPM_OK;
}

// packed = all-to-all layout [c][dest rank][xl][y_local][nzp]; every spectrum is spec_elems() complex
static int lg2i(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
// planes per chunk of the all-to-all layouts (mcpm_slab_set_chunks; nxl when not chunked)
static int chunk_planes(const mcpm_plan *p) { return p->nxl / (p->chunks > 0 ? p->chunks : 1); }

static YLayout ylayout(const mcpm_plan *p, bool packed) {
    const int64_t nzp = p->g.nz / 2 + MCPM_NZPAD;
    if (!packed) return YLayout{spec_elems(p), (int64_t)p->g.ny * nzp, 0, p->g.ny, p->nxl, lg2i(p->g.ny), p->xw0, p->xwn, 0, lg2i(p->nxl), p->nxl};
    const int nyl = p->g.ny / p->nranks, cw = chunk_planes(p);
    const int64_t sb = (int64_t)cw * nyl * nzp;
    return YLayout{spec_elems(p), (int64_t)nyl * nzp, sb, nyl, p->nxl, lg2i(nyl), p->xw0, p->xwn, (int64_t)p->nranks * sb, lg2i(cw), cw};
}

static XLayout xlayout(const mcpm_plan *p) {
    const FGeom g = fgeom(p);
    const int nyl = g.ny / p->nranks, cw = chunk_planes(p);
    const int64_t sbx = (int64_t)cw * nyl * g.nzp;
    return XLayout{nyl, p->rank * nyl, p->nxl, lg2i(p->nxl), sbx, spec_elems(p), 0, lg2i(cw), cw, (uint32_t)(p->nranks * sbx)};
}

// kz columns per workgroup of the register-heavy passes (ycol2, xfused; 75-118 VGPRs, so a 1024-thread workgroup is
// alone on its CU).  Fused x pass: 8 columns (512 threads at N = 512, two independent workgroups per CU, 64-byte row
// segments) beat 16 (0.51 vs 0.53 ms); 4 columns (32-byte segments) are 2.3x SLOWER: a row segment must stay a whole
// 64-byte access.  Factor-carrying y pass: with streaming stores 16 columns (128-byte segments) win (stage 0.423 vs 0.440
// ms).  Knobs: MCPM_XCOL_LINES, MCPM_YCOL_LINES = 8 | 16 (MCPM_COL_LINES sets both).
static int col_lines(const char *name, int dflt) {
    const char *e = getenv(name);
    if (!e && dflt) e = getenv("MCPM_COL_LINES");
    const int v = e ? atoi(e) : dflt;
    return (v == 8 || v == 16) ? v : dflt;
}
static int col_lines_x() {
    static const int v = col_lines("MCPM_XCOL_LINES", 8);
    return v;
}
static int col_lines_y() {
    static const int v = col_lines("MCPM_YCOL_LINES", 16);
    return v;
}

static int y_columns(mcpm_plan *p, const cf *in, cf *out, int batch, int sign, bool in_packed, bool out_packed) {
    const FGeom g = fgeom(p);
    const YLayout li = ylayout(p, in_packed), lo = ylayout(p, out_packed);
    StageTimer st_(p, sign < 0 ? ST_R2C : ST_C2R, pass_bytes(p, batch));
#define CALL(NN)                                                                                       \
    {                                                                                                  \
        constexpr int LINES = ColShape<NN>::LINES, TH = ColShape<NN>::THREADS;                         \
        dim3 grid((unsigned)((g.nzh + LINES - 1) / LINES), (unsigned)(batch * p->xwn));                \
        if (sign < 0) ycol_kernel<NN, -1><<<grid, TH, 0, p->stream>>>(g, in, out, li, lo, (const cf *)p->tw[1]);  \
        else ycol_kernel<NN, +1><<<grid, TH, 0, p->stream>>>(g, in, out, li, lo, (const cf *)p->tw[1]);           \
    }
    DISPATCH_N(g.ny, CALL)
#undef CALL
    MCPM_LAUNCH_CHECK(p, "ycol_kernel");
    return MCPM_OK;
}

// y pass with the y / z force factors: expand (2 -> 3 spectra, inverse) or contract (3 -> 2, forward)
static int y_columns2(mcpm_plan *p, const cf *in, cf *out, bool expand, bool in_packed, bool out_packed, int parts = 3) {
    const FGeom g = fgeom(p);
    const YLayout li = ylayout(p, in_packed), lo = ylayout(p, out_packed);
    StageTimer st_(p, expand ? ST_C2R : ST_R2C, pass_bytes(p, parts == 3 ? 3 : (parts == 1 ? 1 : 2)));
#define CALLL(NN, ML)                                                                                  \
    {                                                                                                  \
        constexpr int LINES = ColShape<NN, ML>::LINES, TH = ColShape<NN, ML>::THREADS;                 \
        dim3 grid((unsigned)((g.nzh + LINES - 1) / LINES), (unsigned)p->xwn);                          \
        if (expand) ycol2_kernel<NN, true, ML><<<grid, TH, 0, p->stream>>>(g, in, out, li, lo, (const cf *)p->tw[1], parts);  \
        else ycol2_kernel<NN, false, ML><<<grid, TH, 0, p->stream>>>(g, in, out, li, lo, (const cf *)p->tw[1], parts);        \
    }
    static const int ysmall = col_lines("MCPM_YCOL_LINES_SMALL", 0);
    const int ylines_small = ysmall ? ysmall : 16;
#define CALL(NN)                                                                                       \
    if ((NN >= 512 ? col_lines_y() : ylines_small) == 8) CALLL(NN, 8) else CALLL(NN, 16)
    DISPATCH_N(g.ny, CALL)
#undef CALL
#undef CALLL
    MCPM_LAUNCH_CHECK(p, "ycol2_kernel");
    return MCPM_OK;
}

static size_t xf_lds_pad() {   // experiment: unused dynamic LDS (bytes) to lower the x pass's occupancy (MCPM_XF_LDS_PAD)
    static const size_t v = [] { const char *e = getenv("MCPM_XF_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
    return v;
}

static int xcd_remap() {   // MCPM_XCD_REMAP=0 restores launch order (A/B runs)
    static const int v = [] { const char *e = getenv("MCPM_XCD_REMAP"); return e ? atoi(e) : 1; }();
    return v;
}

// x pass over the y rows this rank holds after the transpose (all of them on one GPU)
static int x_fused(mcpm_plan *p, const cf *in, cf *out, int mode) {
    const FGeom g = fgeom(p);
    const int nyl = g.ny / p->nranks;
    XLayout xl = xlayout(p);
    const float scale = 1.f / ((float)g.nx * (float)g.ny * (float)g.nz);
    // one forward + one inverse x pass of (1 + 3) spectra and the k-space multiply
    StageTimer st_(p, ST_KSPACE, (32.0 * p->nxl * g.ny * g.nzh) + 4.0 * pass_bytes(p, 1));
#define CALLL(NN, ML)                                                                                         \
    {                                                                                                         \
        constexpr int LINES = ColShape<NN, ML>::LINES, TH = ColShape<NN, ML>::THREADS;                        \
        dim3 grid((unsigned)((g.nzh + LINES - 1) / LINES), (unsigned)nyl);                                    \
        xl.remap = xcd_remap() && (grid.x * grid.y) % 8 == 0;                                                 \
        if (mode == 0) xfused_kernel<NN, 0, ML><<<grid, TH, xf_lds_pad(), p->stream>>>(g, in, out, xl, scale, (const cf *)p->tw[0]); \
        else xfused_kernel<NN, 1, ML><<<grid, TH, xf_lds_pad(), p->stream>>>(g, in, out, xl, scale, (const cf *)p->tw[0]);           \
    }
    // columns per workgroup: 8 at N >= 512 (above) and below it too -- with 16 the 2 -> 1 (adjoint) form takes 146 VGPRs at
    // N = 256 (three waves per SIMD: 89 us at 256^3 against 61-67 us with 8) and the 1 -> 2 form 128 (56-60 vs 51 us).
    // MCPM_XCOL_LINES_SMALL=16: round 2's choice.
    static const int small_lines = col_lines("MCPM_XCOL_LINES_SMALL", 0);
    const int lines_small = small_lines ? small_lines : 8;
#define CALL(NN)                                                                                              \
    if ((NN >= 512 ? col_lines_x() : lines_small) == 8) CALLL(NN, 8) else CALLL(NN, 16)
    DISPATCH_N(g.nx, CALL)
#undef CALL
#undef CALLL
    MCPM_LAUNCH_CHECK(p, "xfused_kernel");
    return MCPM_OK;
}

// spectrum-side x pass (modes 2..5 of xspec_kernel); `spec` is the caller's plain half-spectrum, `multi` the 3 or 6
// internal spectra
static int x_spec(mcpm_plan *p, const cf *in, cf *out, int mode) {
    const FGeom g = fgeom(p);
    const int nyl = g.ny / p->nranks;
    XLayout xl = xlayout(p);
    const float scale = 1.f / ((float)g.nx * (float)g.ny * (float)g.nz);
    const int nc = (mode == 2 || mode == 4) ? 3 : 6;
    StageTimer st_(p, ST_KSPACE, 8.0 * (nc + 1) * p->nxl * g.ny * g.nzh + nc * pass_bytes(p, 1));
#define CALL(NN)                                                                                              \
    {                                                                                                         \
        constexpr int LINES = ColShape<NN>::LINES, TH = ColShape<NN>::THREADS;                                \
        dim3 grid((unsigned)((g.nzh + LINES - 1) / LINES), (unsigned)nyl);                                    \
        xl.remap = xcd_remap() && (grid.x * grid.y) % 8 == 0;                                                 \
        const cf *tw = (const cf *)p->tw[0];                                                                  \
        if (mode == 2) xspec_kernel<NN, 2><<<grid, TH, 0, p->stream>>>(g, in, out, xl, scale, tw);            \
        else if (mode == 3) xspec_kernel<NN, 3><<<grid, TH, 0, p->stream>>>(g, in, out, xl, scale, tw);       \
        else if (mode == 4) xspec_kernel<NN, 4><<<grid, TH, 0, p->stream>>>(g, in, out, xl, scale, tw);       \
        else xspec_kernel<NN, 5><<<grid, TH, 0, p->stream>>>(g, in, out, xl, scale, tw);                      \
    }
    DISPATCH_N(g.nx, CALL)
#undef CALL
    MCPM_LAUNCH_CHECK(p, "xspec_kernel");
    return MCPM_OK;
}

// half-spectrum (plain layout) -> three force meshes / six Hessian meshes, and the adjoints; single-GPU plans
int mcpm_fftpm_spec_meshes(mcpm_plan *p, const float *spec, float *meshes, int nc) {
    MCPM_TRY(ensure_twiddles(p));
    cf *s = (cf *)p->spec + spec_elems(p);
    MCPM_TRY(x_spec(p, (const cf *)spec, s, nc == 3 ? 2 : 3));
    MCPM_TRY(y_columns(p, s, s, nc, +1, false, false));
    MCPM_TRY(z_inverse(p, s, meshes, p->M, nc));
    return MCPM_OK;
}

int mcpm_fftpm_spec_meshes_vjp(mcpm_plan *p, const float *meshes_bar, float *spec_bar, int nc) {
    MCPM_TRY(ensure_twiddles(p));
    cf *s = (cf *)p->spec + spec_elems(p);
    MCPM_TRY(z_forward(p, meshes_bar, p->M, s, nc));
    MCPM_TRY(y_columns(p, s, s, nc, -1, false, false));
    MCPM_TRY(x_spec(p, s, (cf *)spec_bar, nc == 3 ? 4 : 5));
    return MCPM_OK;
}

// rho (real mesh) -> three force meshes irfftn(-(i k_c)(-1/k^2) rfftn(rho)); single-GPU plans
int mcpm_fftpm_force_meshes(mcpm_plan *p, const float *rho, float *fm3, int interleaved, int nt_out) {
    MCPM_TRY(ensure_twiddles(p));
    const int64_t ss = spec_elems(p);
    cf *s0 = (cf *)p->spec, *s123 = s0 + ss;
    cf *s45 = s123 + 3 * ss;  // A, G between the x and y passes
    MCPM_TRY(z_forward(p, rho, p->M, s0, 1));
    MCPM_TRY(y_columns(p, s0, s0, 1, -1, false, false));
    MCPM_TRY(x_fused(p, s0, s45, 0));
    MCPM_TRY(y_columns2(p, s45, s123, true, false, false));
    if (interleaved) return z_inverse3_il(p, s123, fm3, nt_out != 0);   // [cell][3] for the step kernels
    MCPM_TRY(z_inverse(p, s123, fm3, p->M, 3));
    return MCPM_OK;
}

// adjoint: three real cotangent meshes -> rho_bar = irfftn(sum_c conj(multiplier_c) rfftn(f_bar_c))
int mcpm_fftpm_force_meshes_vjp(mcpm_plan *p, const float *fbar3, float *rho_bar) {
    MCPM_TRY(ensure_twiddles(p));
    const int64_t ss = spec_elems(p);
    cf *s0 = (cf *)p->spec, *s123 = s0 + ss;
    cf *s45 = s123 + 3 * ss;
    MCPM_TRY(z_forward(p, fbar3, p->M, s123, 3));
    MCPM_TRY(y_columns2(p, s123, s45, false, false, false));
    MCPM_TRY(x_fused(p, s45, s0, 1));
    MCPM_TRY(y_columns(p, s0, s0, 1, +1, false, false));
    MCPM_TRY(z_inverse(p, s0, rho_bar, p->M, 1));
    return MCPM_OK;
}

// ---- pass-level entry points for the slab-decomposed solve (the all-to-all between them is the host's) ----------
static int x_plain(mcpm_plan *p, const cf *in, cf *out, int dir) {
    const FGeom g = fgeom(p);
    StageTimer st_(p, dir < 0 ? ST_R2C : ST_C2R, pass_bytes(p, 1));
#define CALL(NN)                                                                                              \
    {                                                                                                         \
        constexpr int LINES = ColShape<NN>::LINES, TH = ColShape<NN>::THREADS;                                \
        dim3 grid((unsigned)((g.nzh + LINES - 1) / LINES), (unsigned)g.ny);                                   \
        if (dir < 0) xplain_kernel<NN, -1><<<grid, TH, 0, p->stream>>>(g, in, out, (const cf *)p->tw[0]);     \
        else xplain_kernel<NN, +1><<<grid, TH, 0, p->stream>>>(g, in, out, (const cf *)p->tw[0]);             \
    }
    DISPATCH_N(g.nx, CALL)
#undef CALL
    MCPM_LAUNCH_CHECK(p, "xplain_kernel");
    return MCPM_OK;
}

// Generic unnormalised R2C / C2R of `batch` meshes (M floats apart) <-> plain half-spectra (Mh complex apart) with the
// hand-written passes; one padded spectrum of scratch (allocated on first use), so each mesh is three kernels.
// Like numpy's irfftn, the C2R ignores the imaginary parts of the kz = 0 and Nyquist modes after the x / y transforms.
static int fft_scratch(mcpm_plan *p, cf **s) {
    if (!p->fft_pad && hipMalloc((void **)&p->fft_pad, sizeof(cf) * spec_elems(p)) != hipSuccess)
        return mcpm_fail(p, MCPM_E_NOMEM, "padded spectrum scratch");
    *s = (cf *)p->fft_pad;
    return MCPM_OK;
}

int mcpm_fftpm_r2c(mcpm_plan *p, const float *real, float *spec, int batch) {
    MCPM_TRY(ensure_twiddles(p));
    cf *s;
    MCPM_TRY(fft_scratch(p, &s));
    for (int b = 0; b < batch; ++b) {
        MCPM_TRY(z_forward(p, real + (int64_t)b * p->M, p->M, s, 1));
        MCPM_TRY(y_columns(p, s, s, 1, -1, false, false));
        MCPM_TRY(x_plain(p, s, (cf *)spec + (int64_t)b * p->Mh, -1));
    }
    return MCPM_OK;
}

int mcpm_fftpm_c2r(mcpm_plan *p, const float *spec, float *real, int batch) {
    MCPM_TRY(ensure_twiddles(p));
    cf *s;
    MCPM_TRY(fft_scratch(p, &s));
    for (int b = 0; b < batch; ++b) {
        MCPM_TRY(x_plain(p, (const cf *)spec + (int64_t)b * p->Mh, s, +1));
        MCPM_TRY(y_columns(p, s, s, 1, +1, false, false));
        MCPM_TRY(z_inverse(p, s, real + (int64_t)b * p->M, p->M, 1));
    }
    return MCPM_OK;
}

extern "C" {

int64_t mcpm_slab_spec_elems(const mcpm_plan *p) { return p ? spec_elems(p) : 0; }

int mcpm_slab_zfwd(mcpm_plan *p, const float *real, int64_t real_bstride, float *spec, int batch) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, real && spec && batch >= 1, MCPM_E_ARG, "mcpm_slab_zfwd: bad argument");
    MCPM_REQUIRE(p, mcpm_fftpm_supported(p), MCPM_E_UNSUPPORTED, "slab FFT needs power-of-two axes in [64, 1024]");
    MCPM_TRY(ensure_twiddles(p));
    return z_forward(p, real, real_bstride, (cf *)spec, batch);
}

int mcpm_slab_zinv(mcpm_plan *p, const float *spec, float *real, int64_t real_bstride, int batch) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, real && spec && batch >= 1, MCPM_E_ARG, "mcpm_slab_zinv: bad argument");
    MCPM_REQUIRE(p, mcpm_fftpm_supported(p), MCPM_E_UNSUPPORTED, "slab FFT needs power-of-two axes in [64, 1024]");
    MCPM_TRY(ensure_twiddles(p));
    return z_inverse(p, (const cf *)spec, real, real_bstride, batch);
}

int mcpm_slab_zinv3_il(mcpm_plan *p, const float *spec3, float *real_il) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, real_il && spec3, MCPM_E_ARG, "mcpm_slab_zinv3_il: bad argument");
    MCPM_REQUIRE(p, mcpm_fftpm_supported(p), MCPM_E_UNSUPPORTED, "slab FFT needs power-of-two axes in [64, 1024]");
    MCPM_TRY(ensure_twiddles(p));
    return z_inverse3_il(p, (const cf *)spec3, real_il);
}

int mcpm_slab_ycol(mcpm_plan *p, const float *in, float *out, int batch, int sign, int in_packed, int out_packed) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out && batch >= 1 && (sign == 1 || sign == -1), MCPM_E_ARG, "mcpm_slab_ycol: bad argument");
    MCPM_REQUIRE(p, in != out || in_packed == out_packed, MCPM_E_ARG, "mcpm_slab_ycol: in-place needs equal layouts");
    MCPM_REQUIRE(p, mcpm_fftpm_supported(p), MCPM_E_UNSUPPORTED, "slab FFT needs power-of-two axes in [64, 1024]");
    MCPM_TRY(ensure_twiddles(p));
    return y_columns(p, (const cf *)in, (cf *)out, batch, sign, in_packed != 0, out_packed != 0);
}

int mcpm_slab_set_chunks(mcpm_plan *p, int chunks) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, chunks >= 1 && (chunks & (chunks - 1)) == 0 && p->nxl % chunks == 0 && p->nxl / chunks >= 1, MCPM_E_ARG,
                 "mcpm_slab_set_chunks: the chunk count must be a power of two dividing the local planes");
    p->chunks = chunks;
    return MCPM_OK;
}

int mcpm_slab_set_window(mcpm_plan *p, int x0, int count) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, x0 >= 0 && count >= 1 && x0 + count <= p->nxl, MCPM_E_ARG, "mcpm_slab_set_window: window outside the local planes");
    p->xw0 = x0;
    p->xwn = count;
    return MCPM_OK;
}

int mcpm_slab_ycol2(mcpm_plan *p, const float *in, float *out, int expand, int in_packed, int out_packed, int parts) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out && in != out && parts >= 1 && parts <= 3, MCPM_E_ARG, "mcpm_slab_ycol2: bad argument");
    MCPM_REQUIRE(p, mcpm_fftpm_supported(p), MCPM_E_UNSUPPORTED, "slab FFT needs power-of-two axes in [64, 1024]");
    MCPM_TRY(ensure_twiddles(p));
    return y_columns2(p, (const cf *)in, (cf *)out, expand != 0, in_packed != 0, out_packed != 0, parts);
}

int mcpm_slab_xfused(mcpm_plan *p, const float *in, float *out, int mode) {
    if (!p) return MCPM_E_ARG;
    MCPM_REQUIRE(p, in && out && in != out && mode >= 0 && mode <= 5, MCPM_E_ARG, "mcpm_slab_xfused: bad argument");
    MCPM_REQUIRE(p, mcpm_fftpm_supported(p), MCPM_E_UNSUPPORTED, "slab FFT needs power-of-two axes in [64, 1024]");
    MCPM_TRY(ensure_twiddles(p));
    if (mode >= 2) return x_spec(p, (const cf *)in, (cf *)out, mode);
    return x_fused(p, (const cf *)in, (cf *)out, mode);
}

}  // extern "C"
