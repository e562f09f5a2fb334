// Host-side float64 growth and distance tables of libmcpm.so (no GPU work).
// Reference: montecosmo/nbody.py:679-745 (_growth_factor_ODE) and :842-856 (a2chi), which integrate with
// jax_cosmo's fixed-step RK4 `odeint` over jax_cosmo.background (matter + curvature + w0-wa dark energy, no
// radiation; jax_cosmo==0.1.0 is not vendored in the reference, its published algorithm is restated here).
#include <cmath>
#include <vector>

#include "../../include/mcpm.h"

namespace {

struct Bg {
    double Om, Ode, Ok, w0, wa;
    double w(double a) const { return w0 + (1.0 - a) * wa; }
    double f_de(double a) const {
        const double eps = 1.1920928955078125e-07;  // float32 epsilon, as jax_cosmo guards log(a) at a = 1
        return -3.0 * (1.0 + w0) + 3.0 * wa * ((a - 1.0) / std::log(a - eps) - 1.0);
    }
    double Esqr(double a) const { return Om * std::pow(a, -3) + Ok * std::pow(a, -2) + Ode * std::pow(a, f_de(a)); }
    double Om_a(double a) const { return Om * std::pow(a, -3) / Esqr(a); }
    double Ode_a(double a) const { return Ode * std::pow(a, f_de(a)) / Esqr(a); }
};

struct Y {
    double g1, g2, f1, f2;
};
inline Y operator+(const Y &a, const Y &b) { return {a.g1 + b.g1, a.g2 + b.g2, a.f1 + b.f1, a.f2 + b.f2}; }
inline Y operator*(double s, const Y &a) { return {s * a.g1, s * a.g2, s * a.f1, s * a.f2}; }

Y derivs(const Bg &c, const Y &y, double x) {
    double q = (2.0 - (c.Om_a(x) + (1.0 + 3.0 * c.w(x)) * c.Ode_a(x)) / 2) / x;
    double r = 1.5 * c.Om_a(x) / (x * x);
    return {y.f1, y.f2, -q * y.f1 + r * y.g1, -q * y.f2 + r * y.g2 - r * y.g1 * y.g1};
}

std::vector<double> logspace(double lo, double hi, int n) {
    std::vector<double> t(n);
    for (int i = 0; i < n; ++i) t[i] = std::pow(10.0, n > 1 ? lo + (hi - lo) * i / (n - 1) : lo);
    return t;
}

}  // namespace

extern "C" {

int mcpm_growth_table(double Omega_m, double Omega_de, double Omega_k, double w0, double wa, double log10_amin,
                      int steps, double *a, double *g, double *f, double *h, double *g2, double *f2, double *h2) {
    if (steps < 2 || !a || !g || !f || !h || !g2 || !f2 || !h2) return MCPM_E_ARG;
    Bg c{Omega_m, Omega_de, Omega_k, w0, wa};
    std::vector<double> t = logspace(log10_amin, 0.0, steps);
    std::vector<Y> ys(steps);
    Y y{t[0], -3.0 / 7 * t[0] * t[0], 1.0, -6.0 / 7 * t[0]};
    double tp = t[0];
    for (int i = 0; i < steps; ++i) {
        double hh = t[i] - tp;
        Y k1 = derivs(c, y, tp);
        Y k2 = derivs(c, y + (hh / 2) * k1, tp + hh / 2);
        Y k3 = derivs(c, y + (hh / 2) * k2, tp + hh / 2);
        Y k4 = derivs(c, y + hh * k3, t[i]);
        y = y + (hh / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
        tp = t[i];
        ys[i] = y;
    }
    const double y1n = ys[steps - 1].g1, y2n = ys[steps - 1].g2;
    for (int i = 0; i < steps; ++i) {
        Y d = derivs(c, ys[i], t[i]);
        a[i] = t[i];
        g[i] = ys[i].g1 / y1n;
        g2[i] = ys[i].g2 / y2n;
        f[i] = ys[i].f1 / y1n * t[i] / g[i];
        f2[i] = ys[i].f2 / y2n * t[i] / g2[i];
        h[i] = d.f1 / y1n * t[i] / g[i];
        h2[i] = d.f2 / y2n * t[i] / g2[i];
    }
    return MCPM_OK;
}

int mcpm_distance_table(double Omega_m, double Omega_de, double Omega_k, double w0, double wa, double log10_amin,
                        int steps, double *a, double *chi) {
    if (steps < 2 || !a || !chi) return MCPM_E_ARG;
    Bg c{Omega_m, Omega_de, Omega_k, w0, wa};
    const double rh = 2997.92458;
    std::vector<double> t = logspace(log10_amin, 0.0, steps);
    auto fn = [&](double x) {  // d chi / d ln a
        double xa = std::exp(x);
        return rh / (xa * xa * std::sqrt(c.Esqr(xa))) * xa;
    };
    double y = 0.0, tp = std::log(t[0]);
    for (int i = 0; i < steps; ++i) {
        double ti = std::log(t[i]), hh = ti - tp;
        double k1 = fn(tp), k2 = fn(tp + hh / 2), k3 = fn(tp + hh / 2), k4 = fn(ti);
        y += hh / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4);
        tp = ti;
        a[i] = t[i];
        chi[i] = y;
    }
    const double last = chi[steps - 1];
    for (int i = 0; i < steps; ++i) chi[i] = last - chi[i];
    return MCPM_OK;
}

}  // extern "C"
