// The reference's solver, restated: MINPACK's hybrd / hybrj (Powell's hybrid method, the default method "hybr" of
// scipy.optimize.root, which the reference calls at raytracing/geometry.py:682, 744, 823) applied to the reference's own
// zero functions (geometry.py:635-657 reflections, :690-715 diffraction with its hand-written derivative, :751-797
// reflections + diffraction), with SciPy's defaults (xtol 1.49012e-8, maxfev 200 (n + 1) resp. 100 (n + 1), factor 100,
// forward differences with step sqrt(machine epsilon), diag from the column norms).
//
// Why it exists: the reference keeps a candidate path only when this solver reports success (ray_tracing.py:309, 345,
// 371), and the solver does not always report the root of these systems when one exists.  The library's kernels find
// the roots in closed form; in "reference-compatible" mode every path they found is put through this routine, started
// where the reference starts it, and is dropped when the reference's solver would not have reported it.
//
// The arithmetic follows the reference's as it runs on the machine its fixtures were made on, operation by operation:
// MINPACK's loops without fused multiply-adds (SciPy's build has none); NumPy's three-element dot product, norm and
// small matrix products as OpenBLAS computes them there (fused chains in the orders noted at dot3 / times_matrix /
// times_matrix_t).  The iterates then agree with SciPy's bit for bit (tests/golden/solvers.json: x, success, nfev).
//
// Compiles for the device (hipcc) and for the host (a plain C++ compiler, for the CPU test of the restatement).
//
// Every routine takes the size of the system as a template argument NC as well as at run time: with NC > 0 every loop
// bound is a compile-time constant, the loops unroll completely and the vectors and matrices of the method live in
// registers (a thread of a GPU kernel that indexes them at run time keeps them in scratch memory and waits for every
// element: 25 ms for one solve).  NC = 0 takes the size at run time.  The operations and their order are the same.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#pragma clang fp contract(off)
#else
#define RT_HD
#endif

#if defined(__clang__)
#define RT_UNROLL _Pragma("unroll")
#define RT_NOUNROLL _Pragma("nounroll")
#else
#define RT_UNROLL
#define RT_NOUNROLL
#endif

namespace rt_hybr {

#define RT_HYBR_MAX_N 13 /* 3 unknowns per reflection (RT_MAX_ORDER = 4) + 1 */

RT_HD inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// numpy.dot of two 3-vectors (OpenBLAS ddot, n = 3): the first product rounded, then two fused steps
RT_HD inline double dot3(const double *a, const double *b) { return fma_(a[2], b[2], fma_(a[1], b[1], a[0] * b[0])); }
// numpy.linalg.norm of a 3-vector: sqrt(x.dot(x))
RT_HD inline double norm3(const double *a) { return sqrt(dot3(a, a)); }
// (1,3) @ M for a C-contiguous 3x3 M (row-major m[3*k + j]): out[j] = sum_k a[k] M[k][j], fused in the order k = 0, 1, 2
RT_HD inline void times_matrix(const double *a, const double *m, double *out) {
    for (int j = 0; j < 3; ++j) out[j] = fma_(a[2], m[6 + j], fma_(a[1], m[3 + j], a[0] * m[j]));
}
// (1,3) @ M.T for the same M given as a transposed view: out[j] = sum_k a[k] M[j][k], fused in the order k = 1, 0, 2
RT_HD inline void times_matrix_t(const double *a, const double *m, double *out) {
    for (int j = 0; j < 3; ++j) out[j] = fma_(a[2], m[3 * j + 2], fma_(a[0], m[3 * j], a[1] * m[3 * j + 1]));
}

// projection_matrix_from_vector (geometry.py:414-435): columns x, y, z of the returned matrix, row-major m[3*i + j]
RT_HD inline void projection_matrix(const double *v, double *m) {
    const double nv = norm3(v);
    double z[3], y[3], x[3];
    for (int i = 0; i < 3; ++i) z[i] = v[i] / nv;
    const double ones[3] = {1.0, 1.0, 1.0};
    const double s = dot3(ones, z);
    for (int i = 0; i < 3; ++i) y[i] = 1.0 - s * z[i];
    const double ny = norm3(y);
    for (int i = 0; i < 3; ++i) y[i] = y[i] / ny;
    x[0] = y[1] * z[2] - y[2] * z[1];
    x[1] = y[2] * z[0] - y[0] * z[2];
    x[2] = y[0] * z[1] - y[1] * z[0];
    for (int i = 0; i < 3; ++i) {
        m[3 * i + 0] = x[i];
        m[3 * i + 1] = y[i];
        m[3 * i + 2] = z[i];
    }
}

// One of the reference's three systems.  n_refl planes (normal, d), the fixed end points, and for a diffracting edge the
// edge-aligned frame with what the reference precomputes in it.
struct System {
    int n_refl;        // reflections
    int has_edge;      // ends with a diffraction
    double normal[4][3], d[4];
    double points[6][3];  // origin, the reflection points being iterated, then the destination (or the edge's first end)
    double matrix[9];     // projection_matrix(edge[1] - edge[0]), row-major
    double ze, Xe[3], projected_A[3], projected_B[3];
    RT_HD int n_unknowns() const { return 3 * n_refl + (has_edge ? 1 : 0); }
};

// __generate__ (geometry.py:608-632) and the set-up of the three solvers (:676-687, :733-748, :802-833): fills the
// system and the first guess x0.
template <int NR = -1>
RT_HD inline void setup(System &s, const double *origin, const double *destination, const double *planes, int n_refl_rt,
                        const double *edge, double *x0) {
    const int n_refl = NR >= 0 ? NR : n_refl_rt;
    s.n_refl = n_refl;
    s.has_edge = edge != nullptr;
    for (int k = 0; k < 3; ++k) {
        s.points[0][k] = origin[k];
        s.points[n_refl + 1][k] = destination[k];
    }
    RT_UNROLL
    for (int i = 0; i < n_refl; ++i) {
        for (int k = 0; k < 3; ++k) s.normal[i][k] = planes[4 * i + k];
        s.d[i] = planes[4 * i + 3];
        const double t = 2 * (s.d[i] + dot3(s.points[i], s.normal[i]));
        for (int k = 0; k < 3; ++k) s.points[i + 1][k] = 0.5 * ((s.points[i][k] + destination[k]) - t * s.normal[i][k]);
    }
    if (edge) {
        double v[3] = {edge[3] - edge[0], edge[4] - edge[1], edge[5] - edge[2]};
        projection_matrix(v, s.matrix);
        times_matrix(edge, s.matrix, s.Xe);
        s.ze = s.Xe[2];
        times_matrix(destination, s.matrix, s.projected_B);
        if (n_refl == 0) {
            times_matrix(origin, s.matrix, s.projected_A);
            x0[0] = s.ze;
            return;
        }
        for (int k = 0; k < 3; ++k) s.points[n_refl + 1][k] = edge[k];  // geometry.py:821: stays there for the whole solve
    }
    RT_UNROLL
    for (int i = 0; i < n_refl; ++i)
        for (int k = 0; k < 3; ++k) x0[3 * i + k] = s.points[i + 1][k];
    if (edge) x0[3 * n_refl] = edge[0];  // geometry.py:824: points[1:].flat[:-2] ends with the world x of the edge's first end
}

// residual of the diffraction equation for the point of the edge at height z1 (in the edge frame) seen from pA
RT_HD inline double keller(const System &s, const double *pA, double x_last, double *jac) {
    const double t = x_last - s.ze;
    double xn[3] = {s.Xe[0], s.Xe[1], s.Xe[2] + t};
    double i_[3], d_[3];
    for (int k = 0; k < 3; ++k) {
        i_[k] = xn[k] - pA[k];
        d_[k] = s.projected_B[k] - xn[k];
    }
    const double num1 = (s.ze - pA[2]) + t;
    const double num2 = (s.projected_B[2] - s.ze) - t;
    const double den1 = norm3(i_), den2 = norm3(d_);
    if (jac) {
        const double den1_prime = num1 / den1, den2_prime = -num2 / den2;
        *jac = (-num1 - den1_prime) / (den1 * den1) - (num2 - den2_prime) / (den2 * den2);
    }
    return num1 / den1 - num2 / den2;
}

// the zero functions: f = F(x); jac only for the pure diffraction (n = 1), where the reference hands its own derivative
template <int NR = -1, int HE = -1>
RT_HD inline void evaluate(System &s, const double *x, double *f, double *jac) {
    const int n = NR >= 0 ? NR : s.n_refl;
    if (n == 0) {
        f[0] = keller(s, s.projected_A, x[0], jac);
        return;
    }
    RT_UNROLL
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) s.points[i + 1][k] = x[3 * i + k];
    double v[5][3], norms[5];
    RT_UNROLL
    for (int i = 0; i <= n; ++i) {
        for (int k = 0; k < 3; ++k) v[i][k] = s.points[i + 1][k] - s.points[i][k];
        norms[i] = norm3(v[i]);
    }
    RT_UNROLL
    for (int i = 0; i < n; ++i) {
        const double g = norms[i] / norms[i + 1];
        const double t = 2 * (s.d[i] + dot3(s.points[i], s.normal[i]));
        for (int k = 0; k < 3; ++k) f[3 * i + k] = (g * v[i + 1][k] - v[i][k]) - t * s.normal[i][k];
    }
    if (HE >= 0 ? HE != 0 : s.has_edge != 0) {
        double pA[3];
        times_matrix(s.points[n], s.matrix, pA);
        f[3 * n] = keller(s, pA, x[3 * n], nullptr);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MINPACK
// ---------------------------------------------------------------------------------------------------------------
#define RT_EPSMCH 2.220446049250313e-16
#define RT_GIANT 1.7976931348623157e308

// MINPACK's enorm: the Euclidean norm with its guards against overflow and underflow (three sums: large, mid-range and
// small components).
RT_HD inline double enorm_guarded(int n, const double *x, int stride = 1) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        const double xabs = fabs(x[i * stride]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double r = x3max / xabs;
                s3 = 1 + s3 * (r * r);
                x3max = xabs;
            } else if (xabs != 0) {
                const double r = xabs / x3max;
                s3 += r * r;
            }
        } else {
            if (xabs > x1max) {
                const double r = x1max / xabs;
                s1 = 1 + s1 * (r * r);
                x1max = xabs;
            } else {
                const double r = xabs / x1max;
                s1 += r * r;
            }
        }
    }
    if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

// The guarded form with its components passed by value, kept out of line on the device: a vector with a component
// outside the mid range is rare, and inlined at the thirty places that take a norm the guards were a third of the code
// of a solve (which has to stay within the instruction cache).  Components beyond n are zero and add nothing.
#if defined(__HIPCC__)
__device__ __noinline__
#endif
    inline double
    enorm_guarded_by_value(int n, double x0, double x1, double x2, double x3, double x4, double x5, double x6, double x7,
                           double x8, double x9, double x10, double x11, double x12) {
    const double x[RT_HYBR_MAX_N] = {x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12};
    return enorm_guarded(n, x);
}

// enorm as the method calls it.  When every component is zero or in the mid range, the guarded form adds their squares
// in order and takes the root of the sum (its other two sums stay zero and drop out of its last expression exactly):
// that is done here; anything else goes to the guarded form.
// (n is a compile-time constant wherever the caller's loops have unrolled)
RT_HD inline double enorm(int n, const double *x, int stride = 1) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    const double agiant = rgiant / (double)n;
    double s2 = 0;
    bool plain = true;
    RT_UNROLL
    for (int i = 0; i < n; ++i) {
        const double xabs = fabs(x[i * stride]);
        plain = plain && (xabs == 0 || (xabs > rdwarf && xabs < agiant));
        s2 += xabs * xabs;
    }
    if (plain) return sqrt(s2);
#if defined(__HIP_DEVICE_COMPILE__)
    double v[RT_HYBR_MAX_N];
    RT_UNROLL
    for (int i = 0; i < RT_HYBR_MAX_N; ++i) v[i] = 0;
    RT_UNROLL
    for (int i = 0; i < RT_HYBR_MAX_N; ++i)
        if (i < n) v[i] = x[i * stride];
    return enorm_guarded_by_value(n, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12]);
#else
    return enorm_guarded(n, x, stride);
#endif
}

// matrices are column-major with leading dimension RT_HYBR_MAX_N, as MINPACK holds them: a(i, j) = a[i + LD * j]
#define RT_LD RT_HYBR_MAX_N
#define A_(a, i, j) (a)[(i) + RT_LD * (j)]
// the size of the system: the template argument when there is one, else the run-time value
#define RT_N(NC, n_rt) ((NC) > 0 ? (NC) : (n_rt))

template <int NC>
RT_HD inline void qrfac(int n_rt, double *a, double *rdiag, double *acnorm, double *wa) {
    const int n = RT_N(NC, n_rt);
    RT_UNROLL
    for (int j = 0; j < n; ++j) {
        acnorm[j] = enorm(n, &A_(a, 0, j));
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
    }
    RT_UNROLL
    for (int j = 0; j < n; ++j) {
        double ajnorm = enorm(n - j, &A_(a, j, j));
        if (ajnorm != 0) {
            if (A_(a, j, j) < 0) ajnorm = -ajnorm;
            RT_UNROLL
            for (int i = j; i < n; ++i) A_(a, i, j) /= ajnorm;
            A_(a, j, j) += 1;
            RT_UNROLL
            for (int k = j + 1; k < n; ++k) {
                double sum = 0;
                RT_UNROLL
                for (int i = j; i < n; ++i) sum += A_(a, i, j) * A_(a, i, k);
                const double temp = sum / A_(a, j, j);
                RT_UNROLL
                for (int i = j; i < n; ++i) A_(a, i, k) -= temp * A_(a, i, j);
            }
        }
        rdiag[j] = -ajnorm;
    }
}

template <int NC>
RT_HD inline void qform(int n_rt, double *q, double *wa) {
    const int n = RT_N(NC, n_rt);
    RT_UNROLL
    for (int j = 1; j < n; ++j) {
        RT_UNROLL
        for (int i = 0; i < j; ++i) A_(q, i, j) = 0;
    }
    RT_UNROLL
    for (int l = 0; l < n; ++l) {
        const int k = n - l - 1;
        RT_UNROLL
        for (int i = k; i < n; ++i) {
            wa[i] = A_(q, i, k);
            A_(q, i, k) = 0;
        }
        A_(q, k, k) = 1;
        if (wa[k] != 0) {
            RT_UNROLL
            for (int j = k; j < n; ++j) {
                double sum = 0;
                RT_UNROLL
                for (int i = k; i < n; ++i) sum += A_(q, i, j) * wa[i];
                const double temp = sum / wa[k];
                RT_UNROLL
                for (int i = k; i < n; ++i) A_(q, i, j) -= temp * wa[i];
            }
        }
    }
}

// r: upper triangle stored by rows (r[0] = r11, r12 ... r1n, r22 ...); row j starts at j n - j (j - 1) / 2
template <int NC>
RT_HD inline void dogleg(int n_rt, const double *r, const double *diag, const double *qtb, double delta, double *x, double *wa1,
                         double *wa2) {
    const int n = RT_N(NC, n_rt);
    RT_UNROLL
    for (int k = 1; k <= n; ++k) {
        const int j = n - k;  // 0-based row
        const int jj = j * n - (j * (j - 1)) / 2;  // its diagonal element
        double sum = 0;
        RT_UNROLL
        for (int i = j + 1; i < n; ++i) sum += r[jj + (i - j)] * x[i];
        double temp = r[jj];
        if (temp == 0) {
            RT_UNROLL
            for (int i = 0; i <= j; ++i) temp = fmax(temp, fabs(r[i * n - (i * (i - 1)) / 2 + (j - i)]));  // column j
            temp = RT_EPSMCH * temp;
            if (temp == 0) temp = RT_EPSMCH;
        }
        x[j] = (qtb[j] - sum) / temp;
    }
    RT_UNROLL
    for (int j = 0; j < n; ++j) {
        wa1[j] = 0;
        wa2[j] = diag[j] * x[j];
    }
    const double qnorm = enorm(n, wa2);
    if (qnorm <= delta) return;
    RT_UNROLL
    for (int j = 0; j < n; ++j) {
        const double temp = qtb[j];
        const int row = j * n - (j * (j - 1)) / 2;
        RT_UNROLL
        for (int i = j; i < n; ++i) wa1[i] += r[row + (i - j)] * temp;
        wa1[j] /= diag[j];
    }
    const double gnorm = enorm(n, wa1);
    double sgnorm = 0;
    double alpha = delta / qnorm;
    if (gnorm != 0) {
        RT_UNROLL
        for (int j = 0; j < n; ++j) wa1[j] = (wa1[j] / gnorm) / diag[j];
        RT_UNROLL
        for (int j = 0; j < n; ++j) {
            double sum = 0;
            const int row = j * n - (j * (j - 1)) / 2;
            RT_UNROLL
            for (int i = j; i < n; ++i) sum += r[row + (i - j)] * wa1[i];
            wa2[j] = sum;
        }
        double temp = enorm(n, wa2);
        sgnorm = (gnorm / temp) / temp;
        alpha = 0;
        if (sgnorm < delta) {
            const double bnorm = enorm(n, qtb);
            temp = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
            const double d1 = sgnorm / delta, d2 = temp - delta / qnorm, d3 = delta / qnorm, d4 = sgnorm / delta;
            temp = temp - delta / qnorm * (d1 * d1) + sqrt(d2 * d2 + (1 - d3 * d3) * (1 - d4 * d4));
            const double d5 = sgnorm / delta;
            alpha = delta / qnorm * (1 - d5 * d5) / temp;
        }
    }
    const double temp = (1 - alpha) * fmin(sgnorm, delta);
    RT_UNROLL
    for (int j = 0; j < n; ++j) x[j] = temp * wa1[j] + alpha * x[j];
}

// The rotation of r1updt that eliminates b against a: cosine, sine and the number MINPACK stores for it.  MINPACK
// writes the two cases (|a| < |b| and the other) as two branches with the same three operations on swapped operands;
// here the operands are swapped first and the operations written once.
RT_HD inline void givens(double a, double b, double &cos_, double &sin_, double &tau) {
    const double p5 = 0.5, p25 = 0.25;
    const bool steep = fabs(a) < fabs(b);
    const double q = steep ? a / b : b / a;  // cotangent, or tangent
    const double first = p5 / sqrt(p25 + p25 * (q * q));
    const double second = first * q;
    if (steep) {
        sin_ = first;
        cos_ = second;
        tau = 1;
        if (fabs(cos_) * RT_GIANT > 1) tau = 1 / cos_;
    } else {
        cos_ = first;
        sin_ = second;
        tau = sin_;
    }
}

// the rotation r1mpyq reads back from such a stored number
RT_HD inline void givens_stored(double t, double &cos_, double &sin_) {
    const bool inverse = fabs(t) > 1;
    const double u = inverse ? 1 / t : t;
    const double other = sqrt(1 - u * u);
    cos_ = inverse ? u : other;
    sin_ = inverse ? other : u;
}

template <int NC>
RT_HD inline void r1updt(int n_rt, double *s, const double *u, double *v, double *w, bool *sing) {
    const int n = RT_N(NC, n_rt);
    w[n - 1] = s[(n * (n + 1)) / 2 - 1];  // the last diagonal element (m = n)
    RT_UNROLL
    for (int j = n - 2; j >= 0; --j) {
        const int jj = j * n - (j * (j - 1)) / 2;  // diagonal element of row j
        w[j] = 0;
        if (v[j] != 0) {
            double cos_, sin_, tau;
            givens(v[n - 1], v[j], cos_, sin_, tau);
            v[n - 1] = sin_ * v[j] + cos_ * v[n - 1];
            v[j] = tau;
            RT_UNROLL
            for (int i = j; i < n; ++i) {
                const int l = jj + (i - j);
                const double temp = cos_ * s[l] - sin_ * w[i];
                w[i] = sin_ * s[l] + cos_ * w[i];
                s[l] = temp;
            }
        }
    }
    RT_UNROLL
    for (int i = 0; i < n; ++i) w[i] += v[n - 1] * u[i];
    *sing = false;
    RT_UNROLL
    for (int j = 0; j < n - 1; ++j) {
        const int jj = j * n - (j * (j - 1)) / 2;
        if (w[j] != 0) {
            double cos_, sin_, tau;
            givens(s[jj], w[j], cos_, sin_, tau);
            RT_UNROLL
            for (int i = j; i < n; ++i) {
                const int l = jj + (i - j);
                const double temp = cos_ * s[l] + sin_ * w[i];
                w[i] = -sin_ * s[l] + cos_ * w[i];
                s[l] = temp;
            }
            w[j] = tau;
        }
        if (s[jj] == 0) *sing = true;
    }
    const int last = (n * (n + 1)) / 2 - 1;
    s[last] = w[n - 1];
    if (s[last] == 0) *sing = true;
}

// a: M rows (row stride rs, column stride cs), n columns
template <int NC, int MC>
RT_HD inline void r1mpyq(int m_rt, int n_rt, double *a, int rs, int cs, const double *v, const double *w) {
    const int n = RT_N(NC, n_rt);
    const int m = RT_N(MC, m_rt);
    RT_UNROLL
    for (int j = n - 2; j >= 0; --j) {
        double cos_, sin_;
        givens_stored(v[j], cos_, sin_);
        RT_UNROLL
        for (int i = 0; i < m; ++i) {
            const double temp = cos_ * a[i * rs + j * cs] - sin_ * a[i * rs + (n - 1) * cs];
            a[i * rs + (n - 1) * cs] = sin_ * a[i * rs + j * cs] + cos_ * a[i * rs + (n - 1) * cs];
            a[i * rs + j * cs] = temp;
        }
    }
    RT_UNROLL
    for (int j = 0; j < n - 1; ++j) {
        double cos_, sin_;
        givens_stored(w[j], cos_, sin_);
        RT_UNROLL
        for (int i = 0; i < m; ++i) {
            const double temp = cos_ * a[i * rs + j * cs] + sin_ * a[i * rs + (n - 1) * cs];
            a[i * rs + (n - 1) * cs] = -sin_ * a[i * rs + j * cs] + cos_ * a[i * rs + (n - 1) * cs];
            a[i * rs + j * cs] = temp;
        }
    }
}

// r1mpyq on the matrix q (n rows, column-major with leading dimension RT_LD) and on the row vector b, as the method
// calls it twice in a row with the same stored rotations: each rotation is read back once and applied to both.
template <int NC>
RT_HD inline void r1mpyq_both(int n_rt, double *q, double *b, const double *v, const double *w) {
    const int n = RT_N(NC, n_rt);
    RT_UNROLL
    for (int j = n - 2; j >= 0; --j) {
        double cos_, sin_;
        givens_stored(v[j], cos_, sin_);
        RT_UNROLL
        for (int i = 0; i < n; ++i) {
            const double temp = cos_ * A_(q, i, j) - sin_ * A_(q, i, n - 1);
            A_(q, i, n - 1) = sin_ * A_(q, i, j) + cos_ * A_(q, i, n - 1);
            A_(q, i, j) = temp;
        }
        const double temp = cos_ * b[j] - sin_ * b[n - 1];
        b[n - 1] = sin_ * b[j] + cos_ * b[n - 1];
        b[j] = temp;
    }
    RT_UNROLL
    for (int j = 0; j < n - 1; ++j) {
        double cos_, sin_;
        givens_stored(w[j], cos_, sin_);
        RT_UNROLL
        for (int i = 0; i < n; ++i) {
            const double temp = cos_ * A_(q, i, j) + sin_ * A_(q, i, n - 1);
            A_(q, i, n - 1) = -sin_ * A_(q, i, j) + cos_ * A_(q, i, n - 1);
            A_(q, i, j) = temp;
        }
        const double temp = cos_ * b[j] + sin_ * b[n - 1];
        b[n - 1] = -sin_ * b[j] + cos_ * b[n - 1];
        b[j] = temp;
    }
}

struct Result {
    int info;  // 1 = converged (what SciPy reports as success); -1 = stopped at the caller's evaluation limit, unfinished
    int nfev;
};

// hybrd (finite-difference Jacobian) or, with user_jac, hybrj (the system hands its own Jacobian; n = 1 here).
// NR reflections and HE (ends with a diffraction) as template arguments (then NC = 3 NR + HE unknowns), or -1, -1 for a
// system whose shape is known at run time only.
// nfev_limit: a run that has made this many evaluations without ending is given up (info -1); the caller then runs it
// again without a limit.  A GPU kernel uses it to set the few long runs apart from the many short ones.
template <int NR, int HE>
RT_HD inline Result hybr(System &sys, double *x, bool user_jac, int nfev_limit = 0x7fffffff) {
    constexpr int NC = NR >= 0 ? 3 * NR + HE : 0;
    const int n = RT_N(NC, sys.n_unknowns());
    const double xtol = 1.49012e-08, factor = 100.0;
    const double p1 = 0.1, p5 = 0.5, p001 = 1e-3, p0001 = 1e-4;
    const int maxfev = (user_jac ? 100 : 200) * (n + 1);
    double fvec[RT_HYBR_MAX_N], diag[RT_HYBR_MAX_N], qtf[RT_HYBR_MAX_N], wa1[RT_HYBR_MAX_N], wa2[RT_HYBR_MAX_N],
        wa3[RT_HYBR_MAX_N], wa4[RT_HYBR_MAX_N];
    double fjac[RT_HYBR_MAX_N * RT_HYBR_MAX_N], r[(RT_HYBR_MAX_N * (RT_HYBR_MAX_N + 1)) / 2];
    Result res{0, 0};
    double jac1 = 0;
    evaluate<NR, HE>(sys, x, fvec, user_jac ? &jac1 : nullptr);
    res.nfev = 1;
    double fnorm = enorm(n, fvec);
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double delta = 0, xnorm = 0;
    const double eps = sqrt(RT_EPSMCH);
    while (true) {
        bool jeval = true;
        if (user_jac) {
            evaluate<NR, HE>(sys, x, wa1, &jac1);  // n = 1: the Jacobian is the one number
            A_(fjac, 0, 0) = jac1;
        } else {
            // (the loop over the columns is not unrolled, so that the zero function is in the code once: column j is
            // picked out of x and put into the matrix with comparisons against j, which keep every index a constant)
            RT_NOUNROLL
            for (int j = 0; j < n; ++j) {
                double temp = 0;
                RT_UNROLL
                for (int i = 0; i < n; ++i)
                    if (i == j) temp = x[i];
                double h = eps * fabs(temp);
                if (h == 0) h = eps;
                RT_UNROLL
                for (int i = 0; i < n; ++i) wa2[i] = i == j ? temp + h : x[i];
                evaluate<NR, HE>(sys, wa2, wa1, nullptr);
                RT_UNROLL
                for (int i = 0; i < n; ++i) wa3[i] = (wa1[i] - fvec[i]) / h;
                RT_UNROLL
                for (int k = 0; k < n; ++k)
                    if (k == j) {
                        RT_UNROLL
                        for (int i = 0; i < n; ++i) A_(fjac, i, k) = wa3[i];
                    }
            }
            res.nfev += n;
        }
        qrfac<NC>(n, fjac, wa1, wa2, wa3);
        if (iter == 1) {
            RT_UNROLL
            for (int j = 0; j < n; ++j) {
                diag[j] = wa2[j];
                if (wa2[j] == 0) diag[j] = 1;
            }
            RT_UNROLL
            for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0) delta = factor;
        }
        RT_UNROLL
        for (int i = 0; i < n; ++i) qtf[i] = fvec[i];
        RT_UNROLL
        for (int j = 0; j < n; ++j) {
            if (A_(fjac, j, j) != 0) {
                double sum = 0;
                RT_UNROLL
                for (int i = j; i < n; ++i) sum += A_(fjac, i, j) * qtf[i];
                const double temp = -sum / A_(fjac, j, j);
                RT_UNROLL
                for (int i = j; i < n; ++i) qtf[i] += A_(fjac, i, j) * temp;
            }
        }
        bool sing = false;
        RT_UNROLL
        for (int j = 0; j < n; ++j) {
            RT_UNROLL
            for (int i = 0; i < j; ++i) r[i * n - (i * (i - 1)) / 2 + (j - i)] = A_(fjac, i, j);  // column j above the diagonal
            r[j * n - (j * (j - 1)) / 2] = wa1[j];
            if (wa1[j] == 0) sing = true;
        }
        qform<NC>(n, fjac, wa1);
        RT_UNROLL
        for (int j = 0; j < n; ++j) diag[j] = fmax(diag[j], wa2[j]);
        while (true) {
            dogleg<NC>(n, r, diag, qtf, delta, wa1, wa2, wa3);
            RT_UNROLL
            for (int j = 0; j < n; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(n, wa3);
            if (iter == 1) delta = fmin(delta, pnorm);
            evaluate<NR, HE>(sys, wa2, wa4, nullptr);
            ++res.nfev;
            const double fnorm1 = enorm(n, wa4);
            double actred = -1;
            if (fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1 - q * q;
            }
            RT_UNROLL
            for (int i = 0; i < n; ++i) {
                double sum = 0;
                const int row = i * n - (i * (i - 1)) / 2;
                RT_UNROLL
                for (int j = i; j < n; ++j) sum += r[row + (j - i)] * wa1[j];
                wa3[i] = qtf[i] + sum;
            }
            const double temp = enorm(n, wa3);
            double prered = 0;
            if (temp < fnorm) {
                const double q = temp / fnorm;
                prered = 1 - q * q;
            }
            double ratio = 0;
            if (prered > 0) ratio = actred / prered;
            if (ratio < p1) {
                ncsuc = 0;
                ++ncfail;
                delta = p5 * delta;
            } else {
                ncfail = 0;
                ++ncsuc;
                if (ratio >= p5 || ncsuc > 1) delta = fmax(delta, pnorm / p5);
                if (fabs(ratio - 1) <= p1) delta = pnorm / p5;
            }
            if (ratio >= p0001) {
                RT_UNROLL
                for (int j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                    fvec[j] = wa4[j];
                }
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                ++iter;
            }
            ++nslow1;
            if (actred >= p001) nslow1 = 0;
            if (jeval) ++nslow2;
            if (actred >= p1) nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0) res.info = 1;
            if (res.info != 0) return res;
            if (res.nfev >= nfev_limit) {
                res.info = -1;
                return res;
            }
            if (res.nfev >= maxfev) res.info = 2;
            if (p1 * fmax(p1 * delta, pnorm) <= RT_EPSMCH * xnorm) res.info = 3;
            if (nslow2 == 5) res.info = 4;
            if (nslow1 == 10) res.info = 5;
            if (res.info != 0) return res;
            if (ncfail == 2) break;
            RT_UNROLL
            for (int j = 0; j < n; ++j) {
                double sum = 0;
                RT_UNROLL
                for (int i = 0; i < n; ++i) sum += A_(fjac, i, j) * wa4[i];
                wa2[j] = (sum - wa3[j]) / pnorm;
                wa1[j] = diag[j] * ((diag[j] * wa1[j]) / pnorm);
                if (ratio >= p0001) qtf[j] = sum;
            }
            r1updt<NC>(n, r, wa1, wa2, wa3, &sing);
            r1mpyq_both<NC>(n, fjac, qtf, wa2, wa3);
            jeval = false;
        }
    }
}

// hybrd again, as ONE loop with one evaluation of the zero function per trip: the same operations in the same order on
// the same numbers as hybr above (finite-difference form, NC > 0), with the method's two nested loops turned into a
// state -- `col` >= 0: the trip evaluates column `col` of a finite-difference Jacobian; -1: it evaluates a trial point.
// For the lanes of a GPU wave that run different systems side by side: in the nested form a lane that needs a new
// Jacobian waits until every other lane has left its inner loop, and the evaluations of lanes in different places
// of the method go one after the other; here every trip evaluates for all lanes at once, and only the bookkeeping
// between two evaluations (a third of the work) is taken in turns.
// A run can be handed over: after trip_limit trips `claim()` is asked for room (RT_FLAT_STATE_DOUBLES(n) doubles, or null),
// the state of the method is left there and the run ends unfinished (info -1); a later call with resume_from pointing at
// that state goes on where this one stopped and gives, bit for bit, what one uninterrupted call gives.
#define RT_FLAT_STATE_DOUBLES(n) (7 * (n) + (n) * (n) + ((n) * ((n) + 1)) / 2 + 13)
struct NoHandOver {
    RT_HD double *operator()() const { return nullptr; }
};
template <int NR, int HE, class Claim = NoHandOver>
RT_HD inline Result hybr_flat(System &sys, double *x, int trip_limit = 0x7fffffff, const double *resume_from = nullptr,
                              Claim claim = Claim()) {
    constexpr int NC = 3 * NR + HE;
    static_assert(NR >= 1 && NC >= 2, "the finite-difference form of a system whose shape is known at compile time");
    const int n = NC;
    const double xtol = 1.49012e-08, factor = 100.0;
    const double p1 = 0.1, p5 = 0.5, p001 = 1e-3, p0001 = 1e-4;
    const int maxfev = 200 * (n + 1);
    double fvec[RT_HYBR_MAX_N], diag[RT_HYBR_MAX_N], qtf[RT_HYBR_MAX_N], wa1[RT_HYBR_MAX_N], wa2[RT_HYBR_MAX_N],
        wa3[RT_HYBR_MAX_N], wa4[RT_HYBR_MAX_N];
    double fjac[RT_HYBR_MAX_N * RT_HYBR_MAX_N], r[(RT_HYBR_MAX_N * (RT_HYBR_MAX_N + 1)) / 2];
    Result res{0, 0};
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double delta = 0, xnorm = 0, pnorm = 0, h = 0, fnorm = 0;
    const double eps = sqrt(RT_EPSMCH);
    bool jeval = true, sing = false;
    int col = 0;
    // the state in the order it is kept in: seven vectors, the matrix by columns, the packed triangle, four numbers,
    // nine counters and flags
    auto state_io = [&](auto &&io) {
        RT_UNROLL
        for (int i = 0; i < n; ++i) {
            io(x[i]); io(fvec[i]); io(diag[i]); io(qtf[i]); io(wa1[i]); io(wa2[i]); io(wa3[i]);
        }
        RT_UNROLL
        for (int j = 0; j < n; ++j) {
            RT_UNROLL
            for (int i = 0; i < n; ++i) io(A_(fjac, i, j));
        }
        RT_UNROLL
        for (int i = 0; i < (n * (n + 1)) / 2; ++i) io(r[i]);
        io(delta); io(xnorm); io(fnorm); io(pnorm);
    };
    if (resume_from) {
        const double *at = resume_from;
        state_io([&](double &v) { v = *at++; });
        iter = (int)at[0]; ncsuc = (int)at[1]; ncfail = (int)at[2]; nslow1 = (int)at[3]; nslow2 = (int)at[4];
        res.nfev = (int)at[5]; col = (int)at[6]; jeval = at[7] != 0; sing = at[8] != 0;
    } else {
        evaluate<NR, HE>(sys, x, fvec, nullptr);
        res.nfev = 1;
        fnorm = enorm(n, fvec);
    }
    int trips = 0;
    RT_NOUNROLL
    while (true) {
        if (trips == trip_limit) {
            double *at = claim();
            if (at) {
                state_io([&](double &v) { *at++ = v; });
                at[0] = iter; at[1] = ncsuc; at[2] = ncfail; at[3] = nslow1; at[4] = nslow2;
                at[5] = res.nfev; at[6] = col; at[7] = jeval ? 1 : 0; at[8] = sing ? 1 : 0;
            }
            res.info = -1;
            return res;
        }
        ++trips;
        if (col >= 0) {  // the point of column col: x with its col-th component stepped
            double temp = 0;
            RT_UNROLL
            for (int i = 0; i < n; ++i)
                if (i == col) temp = x[i];
            h = eps * fabs(temp);
            if (h == 0) h = eps;
            RT_UNROLL
            for (int i = 0; i < n; ++i) wa2[i] = i == col ? temp + h : x[i];
        }
        evaluate<NR, HE>(sys, wa2, wa4, nullptr);
        if (col >= 0) {
            RT_UNROLL
            for (int k = 0; k < n; ++k)
                if (k == col) {
                    RT_UNROLL
                    for (int i = 0; i < n; ++i) A_(fjac, i, k) = (wa4[i] - fvec[i]) / h;
                }
            ++col;
            if (col == n) {  // the Jacobian is complete: its QR factorisation and what hangs on it
                res.nfev += n;
                jeval = true;
                qrfac<NC>(n, fjac, wa1, wa2, wa3);
                if (iter == 1) {
                    RT_UNROLL
                    for (int j = 0; j < n; ++j) {
                        diag[j] = wa2[j];
                        if (wa2[j] == 0) diag[j] = 1;
                    }
                    RT_UNROLL
                    for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
                    xnorm = enorm(n, wa3);
                    delta = factor * xnorm;
                    if (delta == 0) delta = factor;
                }
                RT_UNROLL
                for (int i = 0; i < n; ++i) qtf[i] = fvec[i];
                RT_UNROLL
                for (int j = 0; j < n; ++j) {
                    if (A_(fjac, j, j) != 0) {
                        double sum = 0;
                        RT_UNROLL
                        for (int i = j; i < n; ++i) sum += A_(fjac, i, j) * qtf[i];
                        const double temp = -sum / A_(fjac, j, j);
                        RT_UNROLL
                        for (int i = j; i < n; ++i) qtf[i] += A_(fjac, i, j) * temp;
                    }
                }
                sing = false;
                RT_UNROLL
                for (int j = 0; j < n; ++j) {
                    RT_UNROLL
                    for (int i = 0; i < j; ++i) r[i * n - (i * (i - 1)) / 2 + (j - i)] = A_(fjac, i, j);
                    r[j * n - (j * (j - 1)) / 2] = wa1[j];
                    if (wa1[j] == 0) sing = true;
                }
                qform<NC>(n, fjac, wa1);
                RT_UNROLL
                for (int j = 0; j < n; ++j) diag[j] = fmax(diag[j], wa2[j]);
                col = -1;
            }
        } else {  // wa4 = F(x + p): is the step any good?
            ++res.nfev;
            const double fnorm1 = enorm(n, wa4);
            double actred = -1;
            if (fnorm1 < fnorm) {
                const double q = fnorm1 / fnorm;
                actred = 1 - q * q;
            }
            RT_UNROLL
            for (int i = 0; i < n; ++i) {
                double sum = 0;
                const int row = i * n - (i * (i - 1)) / 2;
                RT_UNROLL
                for (int j = i; j < n; ++j) sum += r[row + (j - i)] * wa1[j];
                wa3[i] = qtf[i] + sum;
            }
            const double temp = enorm(n, wa3);
            double prered = 0;
            if (temp < fnorm) {
                const double q = temp / fnorm;
                prered = 1 - q * q;
            }
            double ratio = 0;
            if (prered > 0) ratio = actred / prered;
            if (ratio < p1) {
                ncsuc = 0;
                ++ncfail;
                delta = p5 * delta;
            } else {
                ncfail = 0;
                ++ncsuc;
                if (ratio >= p5 || ncsuc > 1) delta = fmax(delta, pnorm / p5);
                if (fabs(ratio - 1) <= p1) delta = pnorm / p5;
            }
            if (ratio >= p0001) {
                RT_UNROLL
                for (int j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                    fvec[j] = wa4[j];
                }
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                ++iter;
            }
            ++nslow1;
            if (actred >= p001) nslow1 = 0;
            if (jeval) ++nslow2;
            if (actred >= p1) nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0) res.info = 1;
            if (res.info != 0) return res;
            if (res.nfev >= maxfev) res.info = 2;
            if (p1 * fmax(p1 * delta, pnorm) <= RT_EPSMCH * xnorm) res.info = 3;
            if (nslow2 == 5) res.info = 4;
            if (nslow1 == 10) res.info = 5;
            if (res.info != 0) return res;
            if (ncfail == 2) {
                col = 0;  // a new Jacobian
            } else {
                RT_UNROLL
                for (int j = 0; j < n; ++j) {
                    double sum = 0;
                    RT_UNROLL
                    for (int i = 0; i < n; ++i) sum += A_(fjac, i, j) * wa4[i];
                    wa2[j] = (sum - wa3[j]) / pnorm;
                    wa1[j] = diag[j] * ((diag[j] * wa1[j]) / pnorm);
                    if (ratio >= p0001) qtf[j] = sum;
                }
                r1updt<NC>(n, r, wa1, wa2, wa3, &sing);
                r1mpyq_both<NC>(n, fjac, qtf, wa2, wa3);
                jeval = false;
            }
        }
        if (col < 0) {  // the next trip tries a step: the dogleg step p, x + p in wa2
            dogleg<NC>(n, r, diag, qtf, delta, wa1, wa2, wa3);
            RT_UNROLL
            for (int j = 0; j < n; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            pnorm = enorm(n, wa3);
            if (iter == 1) delta = fmin(delta, pnorm);
        }
    }
}

// What the reference's solver call returns for one candidate: the solver's points (n_refl reflection points, then the
// diffraction point) and whether SciPy would report success.  NR, HE: the shape of the system as template arguments
// (n_refl and edge must agree with them), or -1, -1.
// nfev_limit / unfinished: see hybr.
// FLAT: through hybr_flat where the system has a finite-difference Jacobian and a shape known at compile time.
template <int NR, int HE, bool FLAT = false>
RT_HD inline bool solve_shape(const double *origin, const double *destination, const double *planes, int n_refl_rt,
                              const double *edge, double *out_points, int *nfev, int nfev_limit = 0x7fffffff,
                              bool *unfinished = nullptr) {
    const int n_refl = NR >= 0 ? NR : n_refl_rt;
    System sys;
    double x[RT_HYBR_MAX_N];
    setup<NR>(sys, origin, destination, planes, n_refl, edge, x);
    Result res;
    if constexpr (FLAT && NR >= 1)
        res = hybr_flat<NR, HE>(sys, x, nfev_limit);  // (here the limit counts trips, i.e. evaluations after the first)
    else
        res = hybr<NR, HE>(sys, x, n_refl == 0, nfev_limit);
    if (nfev) *nfev = res.nfev;
    if (unfinished) *unfinished = res.info < 0;
    RT_UNROLL
    for (int i = 0; i < n_refl; ++i)
        for (int k = 0; k < 3; ++k) out_points[3 * i + k] = x[3 * i + k];
    if (edge) {
        const double xe[3] = {sys.Xe[0], sys.Xe[1], x[3 * n_refl]};
        times_matrix_t(xe, sys.matrix, out_points + 3 * n_refl);
    }
    return res.info == 1;
}

// solve_shape through hybr_flat with a hand-over (see there): trip_limit trips at most, then the state goes where
// `claim` says; resume_from: go on from a state left by an earlier call for the same system.
template <int NR, int HE, class Claim>
RT_HD inline bool solve_shape_flat(const double *origin, const double *destination, const double *planes, const double *edge,
                                   double *out_points, int *nfev, int trip_limit, const double *resume_from, Claim claim,
                                   bool *unfinished) {
    System sys;
    double x[RT_HYBR_MAX_N];
    setup<NR>(sys, origin, destination, planes, NR, edge, x);
    const Result res = hybr_flat<NR, HE, Claim>(sys, x, trip_limit, resume_from, claim);
    if (nfev) *nfev = res.nfev;
    if (unfinished) *unfinished = res.info < 0;
    RT_UNROLL
    for (int i = 0; i < NR; ++i)
        for (int k = 0; k < 3; ++k) out_points[3 * i + k] = x[3 * i + k];
    if (edge) {
        const double xe[3] = {sys.Xe[0], sys.Xe[1], x[3 * NR]};
        times_matrix_t(xe, sys.matrix, out_points + 3 * NR);
    }
    return res.info == 1;
}

// the same for any shape: the shapes up to two reflections (and three without an edge) take their unrolled forms, the
// rest the run-time form
RT_HD inline bool solve_like_reference(const double *origin, const double *destination, const double *planes, int n_refl,
                                       const double *edge, double *out_points, int *nfev) {
    if (edge) {
        if (n_refl == 0) return solve_shape<0, 1>(origin, destination, planes, 0, edge, out_points, nfev);
        if (n_refl == 1) return solve_shape<1, 1>(origin, destination, planes, 1, edge, out_points, nfev);
        if (n_refl == 2) return solve_shape<2, 1>(origin, destination, planes, 2, edge, out_points, nfev);
    } else {
        if (n_refl == 1) return solve_shape<1, 0>(origin, destination, planes, 1, edge, out_points, nfev);
        if (n_refl == 2) return solve_shape<2, 0>(origin, destination, planes, 2, edge, out_points, nfev);
        if (n_refl == 3) return solve_shape<3, 0>(origin, destination, planes, 3, edge, out_points, nfev);
    }
    return solve_shape<-1, -1>(origin, destination, planes, n_refl, edge, out_points, nfev);
}

// the one-loop form of the unrolled shapes (held against SciPy in the CPU test like the others)
RT_HD inline bool solve_like_reference_flat(const double *origin, const double *destination, const double *planes, int n_refl,
                                            const double *edge, double *out_points, int *nfev) {
    const int shape = n_refl * 2 + (edge ? 1 : 0);
    switch (shape) {
        case 2: return solve_shape<1, 0, true>(origin, destination, planes, n_refl, edge, out_points, nfev);
        case 3: return solve_shape<1, 1, true>(origin, destination, planes, n_refl, edge, out_points, nfev);
        case 4: return solve_shape<2, 0, true>(origin, destination, planes, n_refl, edge, out_points, nfev);
        case 5: return solve_shape<2, 1, true>(origin, destination, planes, n_refl, edge, out_points, nfev);
        case 6: return solve_shape<3, 0, true>(origin, destination, planes, n_refl, edge, out_points, nfev);
        default: break;
    }
    return solve_like_reference(origin, destination, planes, n_refl, edge, out_points, nfev);
}

// the one-loop form run in pieces of `piece` trips, each piece going on from the state the one before it left
struct HostHandOver {
    double *room;
    RT_HD double *operator()() const { return room; }
};
template <int NR, int HE>
RT_HD inline bool solve_shape_in_pieces(const double *origin, const double *destination, const double *planes, const double *edge,
                                        double *out_points, int *nfev, int piece) {
    double state[2][RT_FLAT_STATE_DOUBLES(RT_HYBR_MAX_N)];
    bool unfinished = true, ok = false;
    int turn = 0;
    const double *from = nullptr;
    while (unfinished) {
        ok = solve_shape_flat<NR, HE>(origin, destination, planes, edge, out_points, nfev, piece, from, HostHandOver{state[turn]},
                                      &unfinished);
        from = state[turn];
        turn ^= 1;
    }
    return ok;
}
RT_HD inline bool solve_like_reference_in_pieces(const double *origin, const double *destination, const double *planes, int n_refl,
                                                 const double *edge, double *out_points, int *nfev, int piece) {
    const int shape = n_refl * 2 + (edge ? 1 : 0);
    switch (shape) {
        case 2: return solve_shape_in_pieces<1, 0>(origin, destination, planes, edge, out_points, nfev, piece);
        case 3: return solve_shape_in_pieces<1, 1>(origin, destination, planes, edge, out_points, nfev, piece);
        case 4: return solve_shape_in_pieces<2, 0>(origin, destination, planes, edge, out_points, nfev, piece);
        case 5: return solve_shape_in_pieces<2, 1>(origin, destination, planes, edge, out_points, nfev, piece);
        case 6: return solve_shape_in_pieces<3, 0>(origin, destination, planes, edge, out_points, nfev, piece);
        default: break;
    }
    return solve_like_reference(origin, destination, planes, n_refl, edge, out_points, nfev);
}

// the run-time form alone (what the unrolled forms are held against in the CPU test)
RT_HD inline bool solve_like_reference_generic(const double *origin, const double *destination, const double *planes,
                                               int n_refl, const double *edge, double *out_points, int *nfev) {
    return solve_shape<-1, -1>(origin, destination, planes, n_refl, edge, out_points, nfev);
}

#undef A_
#undef RT_LD
#undef RT_N

}  // namespace rt_hybr
