// Device-side FFT building blocks of the hand-written Poisson solve (fftpm.hip) -- and of the micro-benchmarks under
// tools/ that take its passes apart (tools/xpass_bench.hip).  See fftpm.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// A complex number is a native two-float vector: the compiler then keeps (re, im) in one aligned register pair and
// maps complex adds to v_pk_add_f32 and complex multiplies to v_pk_mul_f32 + v_pk_fma_f32 (swaps and sign flips ride
// the op_sel / neg modifiers) instead of pairing unrelated scalars and shuffling them with v_mov.
typedef float cf __attribute__((ext_vector_type(2)));
// Streaming stores for pass outputs that the next kernel reads only after everything else has gone by (a 0.5-1.6 GB
// spectrum never survives in the 4 MB L2s): they keep the L2 for the twiddles and the read stream.  Measured at 512^3:
// z / y passes 7 % faster; outputs the fused x pass reads next stay plain stores (with streaming stores there it loses 2 %).
#define NTSTORE(v, p) __builtin_nontemporal_store((v), (p))

__device__ __forceinline__ cf mkc(float x, float y) {
    cf r = {x, y};
    return r;
}
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cmul(cf a, cf b) {
    const cf bs = {-b.y, b.x};
    return __builtin_elementwise_fma(a.yy, bs, a.xx * b);
}
// multiply by exp(SIGN * i * pi/2): SIGN = -1 (forward) -> -i, +1 (inverse) -> +i
template <int SIGN>
__device__ __forceinline__ cf mul_i(cf a) {
    return SIGN < 0 ? mkc(a.y, -a.x) : mkc(-a.y, a.x);
}

template <int SIGN>
__device__ __forceinline__ void fft4(cf &a0, cf &a1, cf &a2, cf &a3) {
    cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_i<SIGN>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

// in-place radix-8 DFT: v[k] <- sum_n v[n] exp(SIGN 2 pi i n k / 8)
template <int SIGN>
__device__ __forceinline__ void fft8(cf (&v)[8]) {
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4<SIGN>(e0, e1, e2, e3);
    fft4<SIGN>(o0, o1, o2, o3);
    const float h = 0.70710678118654752f;
    // w8^1 = (1 + SIGN i)/sqrt2, w8^2 = SIGN i, w8^3 = (-1 + SIGN i)/sqrt2
    cf w1 = SIGN < 0 ? mkc(h * (o1.x + o1.y), h * (o1.y - o1.x)) : mkc(h * (o1.x - o1.y), h * (o1.y + o1.x));
    cf w2 = mul_i<SIGN>(o2);
    cf w3 = SIGN < 0 ? mkc(h * (o3.y - o3.x), -h * (o3.x + o3.y)) : mkc(-h * (o3.x + o3.y), h * (o3.x - o3.y));
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, w1);
    v[5] = csub(e1, w1);
    v[2] = cadd(e2, w2);
    v[6] = csub(e2, w2);
    v[3] = cadd(e3, w3);
    v[7] = csub(e3, w3);
}

// LDS addressing of a tile of LINES lines of N points.  `e` is the point index; each exchange skews it by
// (e / S) * S2 (S = size of the sub-transform that READS the exchange) to spread the strided reads over banks.
template <int N, int LINES, bool LINE_FASTEST>
struct Tile {
    static constexpr int NP = N + N / 8 + 8;  // padded points per line
    static constexpr int FLOATS2 = NP * LINES;
    int l;
    __device__ __forceinline__ int operator()(int e_padded) const {
        return LINE_FASTEST ? e_padded * LINES + l : l * NP + e_padded;
    }
};

template <int N>
struct FftShape {
    static constexpr int T = N / 8;
    static constexpr int NST8 = (N == 64 || N == 128 || N == 256) ? 2 : 3;
    static constexpr int P8 = NST8 == 2 ? 64 : 512;
    static constexpr int RL = N / P8;  // last radix: 1 (none), 2 or 4
    static_assert(N == 64 || N == 128 || N == 256 || N == 512 || N == 1024, "unsupported FFT length");
};

// Length-N FFT of one line.  In: v[m] = x[u + T m].  Out: v[m] = X[u + T m].  W[j] = exp(-2 pi i j / N).
// Every thread of the workgroup must call it (it synchronises the workgroup).
// DBG (micro-benchmarks only; 0 in the library): bit 0 drops the workgroup barriers (wrong results, right instruction
// stream: what the barriers cost), bit 1 replaces the twiddle-table loads by a constant (what their latency costs).
// PRE (micro-benchmark only; the library passes false): the stage twiddles come from registers (`tw`, filled by fft_twiddles
// before the pass's global loads were issued) instead of a table load behind every exchange barrier.  Measured SLOWER
// (tools/xpass_bench.hip: fused x pass 0.545 vs 0.505 ms): the six extra registers cost more than the hidden latency gains.
template <int N, int SIGN, class TILE, int DBG, bool PRE>
__device__ __forceinline__ void fft_line_core(cf (&v)[8], cf *lds, const cf *__restrict__ W, const cf (&tw)[3], int u, const TILE &tile) {
    constexpr int T = FftShape<N>::T, NST8 = FftShape<N>::NST8, RL = FftShape<N>::RL;
    int P = 1, S = N;
#pragma unroll
    for (int s = 0; s < NST8; ++s) {
        const int S2 = S / 8;
        const int K = u / S2, n2 = u - K * S2;
        fft8<SIGN>(v);
        if (S2 > 1) {
            cf w1 = (DBG & 2) ? mkc(0.6f, 0.8f) : (PRE ? tw[s] : W[n2 * P]);
            // the powers of a register-held twiddle are recomputed by every transform of a kernel: left to common-subexpression
            // elimination they stay live across all of them (28 registers per stage; the fused x pass went from 108 to 152)
            if (PRE) asm volatile("" : "+v"(w1));
            if (SIGN > 0) w1.y = -w1.y;
            const cf w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
            v[1] = cmul(v[1], w1);
            v[2] = cmul(v[2], w2);
            v[3] = cmul(v[3], w3);
            v[4] = cmul(v[4], w4);
            v[5] = cmul(v[5], cmul(w4, w1));
            v[6] = cmul(v[6], cmul(w4, w2));
            v[7] = cmul(v[7], cmul(w4, w3));
        }
        const bool final_stage = (s == NST8 - 1) && (RL == 1);
        if (!final_stage) {
            // exchange: outputs at e = (K + P k) S2 + n2; next stage (sub-size S2) reads e = K' S2 + n1 (S2/R') + n2'
            const int Snext = S2;
            const int S2next = (s == NST8 - 1) ? 1 : S2 / 8;
            if (!(DBG & 1)) __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = (K + P * k) * S2 + n2;
                // the small last radix reads with stride RL: skew by one point per 32 instead
                lds[tile((s == NST8 - 1) ? e + (e >> 5) : e + (e / Snext) * S2next)] = v[k];
            }
            if (!(DBG & 1)) __syncthreads();
            if (s < NST8 - 1) {
                const int Kn = u / S2next, n2n = u - Kn * S2next;
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const int e = Kn * Snext + n1 * S2next + n2n;
                    v[n1] = lds[tile(e + (e / Snext) * S2next)];
                }
            }
        }
        P *= 8;
        S = S2;
    }
    if (RL > 1) {
        // last stage, radix RL: thread u owns butterflies K = u + T j; inputs e = K RL + n1; outputs X[K + P k]
        constexpr int NB = 8 / (RL > 1 ? RL : 8);
        cf x[8];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int n1 = 0; n1 < RL; ++n1) {
                const int e = (u + T * j) * RL + n1;
                x[j * RL + n1] = lds[tile(e + (e >> 5))];
            }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (RL == 2) {
                cf a = x[2 * j], b = x[2 * j + 1];
                v[j] = cadd(a, b);
                v[j + NB] = csub(a, b);
            } else {
                cf a0 = x[4 * j], a1 = x[4 * j + 1], a2 = x[4 * j + 2], a3 = x[4 * j + 3];
                fft4<SIGN>(a0, a1, a2, a3);
                v[j] = a0;
                v[j + NB] = a1;
                v[j + 2 * NB] = a2;
                v[j + 3 * NB] = a3;
            }
        }
    }
}


// first twiddle of every radix-8 stage of thread u: tw[s] = W[(u mod S2_s) P_s], S2_s = N / 8^(s+1), P_s = 8^s
template <int N>
__device__ __forceinline__ void fft_twiddles(const cf *__restrict__ W, int u, cf (&tw)[3]) {
    constexpr int NST8 = FftShape<N>::NST8;
    int P = 1, S = N;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int S2 = S / 8;
        tw[s] = (s < NST8 && S2 > 1) ? W[(u % S2) * P] : mkc(1.f, 0.f);
        P *= 8;
        S = S2 > 0 ? S2 : 1;
    }
}

template <int N, int SIGN, class TILE, int DBG = 0>
__device__ __forceinline__ void fft_line(cf (&v)[8], cf *lds, const cf *__restrict__ W, int u, const TILE &tile) {
    const cf none[3] = {mkc(1.f, 0.f), mkc(1.f, 0.f), mkc(1.f, 0.f)};
    fft_line_core<N, SIGN, TILE, DBG, false>(v, lds, W, none, u, tile);
}
template <int N, int SIGN, class TILE, int DBG = 0>
__device__ __forceinline__ void fft_line_tw(cf (&v)[8], cf *lds, const cf (&tw)[3], int u, const TILE &tile) {
    fft_line_core<N, SIGN, TILE, DBG, true>(v, lds, nullptr, tw, u, tile);
}
