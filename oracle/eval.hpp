// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.hpp).
// Forward-mode dense AD scalar with 3 derivatives: restatement of DenseAd::Evaluation<double,3>
// (opm-material opm/material/densead/Evaluation3.hpp + Math.hpp — NOT in /root/reference).
// UNVERIFIED vs upstream: operation order inside * and /, and the tie rule of min/max, are recalled.
// In-tree evidence of the API: Toolbox::value / comparisons on .value() at ebos/eclfluxmodule.hh:257-355.
#pragma once
#include <cmath>

namespace orc {

struct Ev {
    double v = 0.0;
    double d[3] = {0.0, 0.0, 0.0};
    Ev() = default;
    Ev(double c) : v(c) {}  // constant
    static Ev variable(double x, int idx) {  // priVars.makeEvaluation(pvIdx, timeIdx = 0) on the focus cell
        Ev e(x);
        e.d[idx] = 1.0;
        return e;
    }
};
inline double value(const Ev& a) { return a.v; }
inline double value(double a) { return a; }

inline Ev operator-(const Ev& a) { Ev r; r.v = -a.v; for (int i = 0; i < 3; ++i) r.d[i] = -a.d[i]; return r; }
inline Ev operator+(const Ev& a, const Ev& b) { Ev r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
inline Ev operator-(const Ev& a, const Ev& b) { Ev r; r.v = a.v - b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
inline Ev operator+(const Ev& a, double b) { Ev r = a; r.v = a.v + b; return r; }
inline Ev operator+(double a, const Ev& b) { Ev r = b; r.v = a + b.v; return r; }
inline Ev operator-(const Ev& a, double b) { Ev r = a; r.v = a.v - b; return r; }
inline Ev operator-(double a, const Ev& b) { Ev r; r.v = a - b.v; for (int i = 0; i < 3; ++i) r.d[i] = -b.d[i]; return r; }
// (u v)' = u' v + v' u, value last
inline Ev operator*(const Ev& a, const Ev& b) {
    Ev r;
    const double u = a.v, w = b.v;
    for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] * w + b.d[i] * u;
    r.v = u * w;
    return r;
}
inline Ev operator*(const Ev& a, double b) { Ev r; r.v = a.v * b; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] * b; return r; }
inline Ev operator*(double a, const Ev& b) { return b * a; }
// (u/v)' = (v u' - u v') / v^2
inline Ev operator/(const Ev& a, const Ev& b) {
    Ev r;
    const double u = a.v, w = b.v;
    for (int i = 0; i < 3; ++i) r.d[i] = (w * a.d[i] - b.d[i] * u) / (w * w);
    r.v = u / w;
    return r;
}
inline Ev operator/(const Ev& a, double b) { Ev r; r.v = a.v / b; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] / b; return r; }
inline Ev operator/(double a, const Ev& b) {
    Ev r;
    const double t = -a / (b.v * b.v);
    r.v = a / b.v;
    for (int i = 0; i < 3; ++i) r.d[i] = t * b.d[i];
    return r;
}
inline Ev& operator+=(Ev& a, const Ev& b) { a = a + b; return a; }
inline Ev& operator-=(Ev& a, const Ev& b) { a = a - b; return a; }
inline Ev& operator*=(Ev& a, const Ev& b) { a = a * b; return a; }
inline Ev& operator*=(Ev& a, double b) { a = a * b; return a; }
inline Ev& operator/=(Ev& a, double b) { a = a / b; return a; }
// whole evaluation of the selected argument; on a tie the SECOND argument (UNVERIFIED)
inline Ev max(const Ev& a, const Ev& b) { return (a.v > b.v) ? a : b; }
inline Ev min(const Ev& a, const Ev& b) { return (a.v < b.v) ? a : b; }
inline double max(double a, double b) { return (a > b) ? a : b; }
// Opm::pow(Evaluation, Scalar): value std::pow(x, e); derivative e * x^(e-1) formed as pow_x / x * e (UNVERIFIED)
inline Ev pow(const Ev& a, double e) {
    Ev r;
    const double px = std::pow(a.v, e);
    r.v = px;
    const double df = (a.v == 0.0) ? 0.0 : px / a.v * e;
    for (int i = 0; i < 3; ++i) r.d[i] = df * a.d[i];
    return r;
}
inline double pow(double a, double e) { return std::pow(a, e); }
inline double min(double a, double b) { return (a < b) ? a : b; }

}  // namespace orc
