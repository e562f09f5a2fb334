/*
 * TEST INFRASTRUCTURE (see oracle/__init__.py): OpenMP restatement, in float64, of the particle <-> mesh operators of
 * hsimonfroy/montecosmo, used by oracle/pm_oracle.py when `set_threads(n > 1)` is in force (large parity cases and the
 * reported CPU baseline of bench.py).  Never linked into, imported by or called from the product (montecosmo_amd/).
 *
 * Reference semantics restated (file:line relative to the reference checkout):
 *   montecosmo/nbody.py:220-246   rectangular(s, order): NGP / CIC / TSC / PCS weights of |s|
 *   montecosmo/nbody.py:365-396   paint: id0 = floor(pos) (even order) or round-half-even (odd) cast to int16; for each
 *                                 of the order^3 shifts (lexicographic): idx = id0 + shift, w = prod_ax K(idx - pos) on the
 *                                 UNWRAPPED index, mesh[idx mod shape] += weight * w
 *   montecosmo/nbody.py:398-427   read: the same stencil, out += mesh[idx mod shape] * w
 * and the derivatives jax.grad takes of them (d/dpos K(idx - pos) = -K'(|s|) sign(s), sign(0) = 0; id0 has no gradient).
 * Sums are accumulated in particle order per thread with `omp atomic` on doubles: the rounding differs from numpy's
 * bincount at the 1e-16 level only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline double kern(double s, int order) {
    const double u = fabs(s);
    switch (order) {
        case 1: return 1.0;
        case 2: return 1.0 - u;
        case 3: return u <= 0.5 ? 0.75 - u * u : 0.5 * (1.5 - u) * (1.5 - u);
        default: return u <= 1.0 ? (4.0 - 6.0 * u * u + 3.0 * u * u * u) / 6.0 : (2.0 - u) * (2.0 - u) * (2.0 - u) / 6.0;
    }
}
/* d/ds K(s) = K'(|s|) sign(s) */
static inline double dkern(double s, int order) {
    const double u = fabs(s), sg = (s > 0.0) - (s < 0.0);
    switch (order) {
        case 1: return 0.0;
        case 2: return -sg;
        case 3: return (u <= 0.5 ? -2.0 * u : -(1.5 - u)) * sg;
        default: return (u <= 1.0 ? (-12.0 * u + 9.0 * u * u) / 6.0 : -0.5 * (2.0 - u) * (2.0 - u)) * sg;
    }
}
static inline int pymod(int a, int n) {
    int r = a % n;
    return r < 0 ? r + n : r;
}
/* id0 of nbody.py:375: floor / round-half-even, then the int16 cast (two's complement wrap, as numpy / XLA do) */
static inline int id0_of(double x, int order) {
    const double r = (order & 1) ? nearbyint(x) : floor(x);   /* default rounding mode: half to even */
    return (int)(int16_t)(int64_t)r;
}

typedef struct {
    int64_t off[4];
    double w[4], dw[4];
} Axis;

static inline void axis_setup(double x, int order, int n, int64_t stride, Axis *a) {
    const int i0 = id0_of(x, order), sh = -((order - 1) / 2);
    for (int j = 0; j < order; ++j) {
        const int idx = (int)(int16_t)(i0 + sh + j);
        const double s = (double)idx - x;
        a->w[j] = kern(s, order);
        a->dw[j] = -dkern(s, order);      /* d/dpos K(idx - pos) */
        a->off[j] = (int64_t)pymod(idx, n) * stride;
    }
}

/* mesh (+)= paint(pos, weights); w == NULL: scalar weight wscalar.  The caller zeroes the mesh. */
void pmo_paint(const double *pos, int64_t n, const double *w, double wscalar, int order, int nx, int ny, int nz,
               double *mesh) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        Axis ax, ay, az;
        axis_setup(pos[3 * i], order, nx, (int64_t)ny * nz, &ax);
        axis_setup(pos[3 * i + 1], order, ny, nz, &ay);
        axis_setup(pos[3 * i + 2], order, nz, 1, &az);
        const double wt = w ? w[i] : wscalar;
        for (int a = 0; a < order; ++a)
            for (int b = 0; b < order; ++b) {
                const double wab = wt * (ax.w[a] * ay.w[b]);
                double *row = mesh + ax.off[a] + ay.off[b];
                for (int c = 0; c < order; ++c) {
                    const double v = wab * az.w[c];
#pragma omp atomic
                    row[az.off[c]] += v;
                }
            }
    }
}

/* out[i] = read(pos, mesh)[i];  if pos_bar != NULL also pos_bar[i] += ob_i * d read_i / d pos (ob == NULL: obscalar) */
void pmo_read(const double *pos, int64_t n, const double *mesh, int order, int nx, int ny, int nz, double *out,
              const double *ob, double obscalar, double *pos_bar) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        Axis ax, ay, az;
        axis_setup(pos[3 * i], order, nx, (int64_t)ny * nz, &ax);
        axis_setup(pos[3 * i + 1], order, ny, nz, &ay);
        axis_setup(pos[3 * i + 2], order, nz, 1, &az);
        double v = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
        for (int a = 0; a < order; ++a)
            for (int b = 0; b < order; ++b) {
                const double *row = mesh + ax.off[a] + ay.off[b];
                double r0 = 0.0, r1 = 0.0;
                for (int c = 0; c < order; ++c) {
                    const double m = row[az.off[c]];
                    r0 += m * az.w[c];
                    r1 += m * az.dw[c];
                }
                v += ax.w[a] * ay.w[b] * r0;
                gx += ax.dw[a] * ay.w[b] * r0;
                gy += ax.w[a] * ay.dw[b] * r0;
                gz += ax.w[a] * ay.w[b] * r1;
            }
        if (out) out[i] = v;
        if (pos_bar) {
            const double o = ob ? ob[i] : obscalar;
            pos_bar[3 * i] += o * gx;
            pos_bar[3 * i + 1] += o * gy;
            pos_bar[3 * i + 2] += o * gz;
        }
    }
}

/* wrapped base-cell index (N,3) int16: wrap(id0) of nbody.py:372-375 */
void pmo_cell_index(const double *pos, int64_t n, int order, int nx, int ny, int nz, int16_t *idx) {
    const int dims[3] = {nx, ny, nz};
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) idx[3 * i + a] = (int16_t)pymod(id0_of(pos[3 * i + a], order), dims[a]);
}
