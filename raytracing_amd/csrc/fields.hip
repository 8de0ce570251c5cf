// Field step: received field and power of every path, per-receiver totals
// (reference electromagnetism.py:50-148 antenna frames, :154-215 source field and tree loss, :218-349 reflection,
//  :353-507 UTD wedge diffraction, :610-644 antenna alignment on the shortest path, :648-802 driver and power;
//  electromagnetism_utils.py:146-188; materials_properties.py:19-109).
//
// One thread per path, all carrier frequencies in a loop inside the thread (the path is read once, the geometry of
// its interactions is reused by every frequency).  The reference aborts with an AssertionError where the physics
// checks fail; here each path carries a status code instead (0 = fine) and the caller decides.
#include <cmath>
#include <vector>

#include "rt_candidates.h"
#include <cstring>
#include "rt_common.h"
#include "rt_fresnel.h"

#define N_MATERIALS RT_N_MATERIALS
#define MAX_FREQ 16

struct FreqTable {
    int n_freq;
    double freq[MAX_FREQ], K[MAX_FREQ], lambda[MAX_FREQ];
    double eps_re[MAX_FREQ][N_MATERIALS], eps_im[MAX_FREQ][N_MATERIALS];
    double roughness[N_MATERIALS];
};

struct PathArrays {
    const int32_t *rx;          // [n] receiver index
    const int32_t *kind;        // [n] number of reflections + 256 if the path ends with a diffraction; 0 = line of sight
    const int32_t *chain;       // [n * RT_MAX_ORDER] polygons of the reflections
    const int32_t *edge_polys;  // [n * 2] the two polygons of the diffracting edge (0-face first)
    const double *edge_pts;     // [n * 6] its endpoints
    const double *pts;          // [n * RT_MAX_ORDER * 3] interaction points
    int64_t n;
    // optional (null = default)
    const double *origin;       // [n * 3] first point of the path instead of the emitter
    const double *e_in;         // [n * 6] field arriving at the first interaction instead of the emitter's own field
    const double *aim_tx;       // [n_rx * 3] point the emitter's antenna is aimed at, per receiver
    const double *aim_rx;       // [n_rx * 3] point the receiver's antenna is aimed at
    int n_late;                 // the scene's last n_late polygons were added after the paths were found (tree crowns)
};

enum FieldStatus {
    ST_OK = 0,
    ST_INCIDENCE = 1,       // electromagnetism.py:294 incidence angle outside (0, 90 deg)
    ST_FRESNEL = 2,         // :268
    ST_REFLECTED_GAIN = 3,  // :342
    ST_WEDGE = 4,           // :408
    ST_BETA = 5,            // :413
    ST_INSIDE_WEDGE = 6,    // :425-426
    ST_EDGE_ORIENTATION = 7,// :470
    ST_DIFFRACTED_GAIN = 8, // :502
    ST_CONCAVE = 9,         // geometry.py:916
    ST_ORIGIN = 10          // electromagnetism_utils.py:151-152
};

// ---- small complex helpers -------------------------------------------------------------------------------------
struct cd {
    double re, im;
};
__device__ inline cd cmul(cd a, cd b) { return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ inline cd cscale(double s, cd a) { return cd{s * a.re, s * a.im}; }
__device__ inline cd cadd(cd a, cd b) { return cd{a.re + b.re, a.im + b.im}; }
__device__ inline cd cexpj(double phi) {
    double s, c;
    sincos(phi, &s, &c);
    return cd{c, s};
}
__device__ inline cd cdivc(cd a, cd b) {
    const double d = b.re * b.re + b.im * b.im;
    return cd{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ inline cd csqrt_(cd z) {
    // principal square root, as numpy.sqrt on a complex
    const double m = hypot(z.re, z.im);
    double re = sqrt(0.5 * (m + fabs(z.re)));
    double im = re > 0 ? 0.5 * fabs(z.im) / re : 0.0;
    if (z.re < 0) {
        const double t = re;
        re = im;
        im = t;
    }
    return cd{re, z.im < 0 ? -im : im};
}
struct c3 {
    cd x, y, z;
};
__device__ inline double cnorm3(c3 v) {
    return sqrt(v.x.re * v.x.re + v.x.im * v.x.im + v.y.re * v.y.re + v.y.im * v.y.im + v.z.re * v.z.re + v.z.im * v.z.im);
}
__device__ inline c3 cvec_scale(cd s, d3 v) { return c3{cscale(v.x, s), cscale(v.y, s), cscale(v.z, s)}; }
__device__ inline cd cdot_real(c3 a, d3 b) { return cadd(cadd(cscale(b.x, a.x), cscale(b.y, a.y)), cscale(b.z, a.z)); }

__device__ inline d3 vv_normalize(d3 v) {  // electromagnetism_utils.py:173-177
    const double n = norm3(v);
    if (n == 0) return mk3(0, 0, 0);
    return divs(v, n);
}

struct AntennaFrame {  // electromagnetism.py:66-89: rows of mat are the antenna axes
    d3 pos, ax, ay, az;
};
__device__ inline AntennaFrame align_antenna(d3 pos, d3 target) {
    AntennaFrame f;
    f.pos = pos;
    f.az = vv_normalize(target - pos);
    f.ax = vv_normalize(cross(f.az, mk3(0, 0, 1)));
    f.ay = vv_normalize(cross(f.ax, f.az));
    return f;
}
__device__ inline d3 w2a(const AntennaFrame &f, d3 v) { return mk3(dot(v, f.ax), dot(v, f.ay), dot(v, f.az)); }
// electromagnetism_utils.py:146-161
__device__ inline bool to_spherical(d3 p, double &r, double &el, double &az) {
    if (p.x == 0 && p.y == 0 && p.z == 0) return false;
    if (p.x == 0 && p.y == 0) {
        r = p.z;
        el = 0;
        az = 0;
        return true;
    }
    const double hxy = sqrt(p.x * p.x + p.y * p.y);
    r = sqrt(hxy * hxy + p.z * p.z);
    el = atan2(hxy, p.z);
    az = atan2(p.y, p.x);
    return true;
}
__device__ inline double radiation_pattern(double theta, const rt_antenna_params &ap) {  // electromagnetism.py:211-214
    if (ap.isotropic) return 1.0;
    return pow(cos(theta / 2), 2.0 * ap.alpha);
}

// Source field at `q` (electromagnetism.py:156-176), world frame.
struct SourceGeom {
    d3 v;  // unit polarisation vector in the antenna's frame
    double r, pattern;
    int status;
};
__device__ inline SourceGeom source_geometry(const AntennaFrame &tx, d3 q, const rt_antenna_params &ap) {
    SourceGeom g;
    g.status = ST_OK;
    const d3 pa = w2a(tx, q - tx.pos);
    double r = 1, theta = 0, phi;
    if (!to_spherical(pa, r, theta, phi)) g.status = ST_ORIGIN;
    const d3 r_vv = vv_normalize(pa);
    const d3 pol_a = w2a(tx, tx.ax);
    g.v = vv_normalize(cross(cross(r_vv, pol_a), r_vv));
    g.r = r;
    g.pattern = radiation_pattern(theta, ap);
    return g;
}
__device__ inline c3 source_apply(const SourceGeom &g, d3 ax, d3 ay, d3 az, double K, int &status) {
    if (g.status != ST_OK) {
        status = g.status;
        return c3{{0, 0}, {0, 0}, {0, 0}};
    }
    const double Z0 = 120.0 * M_PI;
    // -j K Z0 / (4 pi) exp(-j K r) / r F(theta)
    const double mag = K * Z0 / (4.0 * M_PI) / g.r * g.pattern;
    const cd ph = cexpj(-K * g.r);
    const cd amp = cmul(cd{0.0, -mag}, ph);
    const c3 ea = cvec_scale(amp, g.v);
    // back to world: E_A @ mat
    c3 ew;
    ew.x = cadd(cadd(cscale(ax.x, ea.x), cscale(ay.x, ea.y)), cscale(az.x, ea.z));
    ew.y = cadd(cadd(cscale(ax.y, ea.x), cscale(ay.y, ea.y)), cscale(az.y, ea.z));
    ew.z = cadd(cadd(cscale(ax.z, ea.x), cscale(ay.z, ea.y)), cscale(az.z, ea.z));
    return ew;
}

// The field functions come in two halves: what depends on the path alone (directions, angles, the vectors of the dyadics)
// and what depends on the carrier frequency as well (coefficients, phases).  A sweep over several frequencies computes
// the first half once per path; the arithmetic of each half is the same whether the two run in one thread or not.

// One reflection (electromagnetism.py:276-304, 247-273, 236-244, 308-311, 340): E <- (E·R) e^{-jK Sr} Si/(Si+Sr)
struct ReflGeom {
    d3 e_per, ei_par, er_par;
    double sin_i, cosi, spread, Sr;  // spread = Si / (Si + Sr)
    int material, status;
};
__device__ inline ReflGeom reflect_geometry(const SceneView &sc, d3 p0, d3 p1, d3 p2, int poly) {
    ReflGeom g;
    const d3 v0 = p1 - p0, v1 = p2 - p1;
    const double Si = norm3(v0), Sr = norm3(v1);
    const d3 si = divs(v0, Si), sr = divs(v1, Sr);
    const d3 n = ld3(sc.planes + 4 * (size_t)poly);
    const double theta_i = acos(dot(n, mk3(-si.x, -si.y, -si.z)));
    g.status = (0 < theta_i && theta_i < M_PI / 2) ? ST_OK : ST_INCIDENCE;
    g.material = sc.material[poly];
    g.sin_i = sin(theta_i);
    g.cosi = cos(theta_i);
    g.e_per = vv_normalize(cross(si, sr));
    g.ei_par = vv_normalize(cross(g.e_per, si));
    g.er_par = vv_normalize(cross(g.e_per, sr));
    g.spread = Si / (Si + Sr);
    g.Sr = Sr;
    return g;
}
__device__ inline c3 reflect_apply(c3 E, const ReflGeom &g, const FreqTable &ft, int f, int &status) {
    if (status == ST_OK) status = g.status;
    const int m = g.material;
    const cd eps{ft.eps_re[f][m], ft.eps_im[f][m]};
    const double sin_i = g.sin_i, cosi = g.cosi;
    const cd root = csqrt_(cd{eps.re - sin_i * sin_i, eps.im});
    const cd ec = cscale(cosi, eps);
    const cd rp = cdivc(cd{ec.re - root.re, ec.im - root.im}, cd{ec.re + root.re, ec.im + root.im});
    const cd rs = cdivc(cd{cosi - root.re, -root.im}, cd{cosi + root.re, root.im});
    double r_par = hypot(rp.re, rp.im), r_per = hypot(rs.re, rs.im);
    if (!(r_par <= 1 && r_per <= 1)) {
        if (status == ST_OK) status = ST_FRESNEL;
    }
    const double rough = ft.roughness[m];
    if (rough > ft.lambda[f] / (8 * cosi)) {
        const double chi = exp(-2 * (ft.K[f] * ft.K[f]) * (rough * rough) * cosi * cosi);
        r_par *= chi;
        r_per *= chi;
    }
    // E @ R with R = e_per e_per^T r_per + ei_par er_par^T r_par
    const cd a = cscale(r_per, cdot_real(E, g.e_per));
    const cd b = cscale(r_par, cdot_real(E, g.ei_par));
    const cd ph = cscale(g.spread, cexpj(-ft.K[f] * g.Sr));
    c3 out;
    out.x = cmul(cadd(cscale(g.e_per.x, a), cscale(g.er_par.x, b)), ph);
    out.y = cmul(cadd(cscale(g.e_per.y, a), cscale(g.er_par.y, b)), ph);
    out.z = cmul(cadd(cscale(g.e_per.z, a), cscale(g.er_par.z, b)), ph);
    return out;
}

// UTD transition function (electromagnetism.py:355-363): 2j sqrt(x) e^{jx} sqrt(pi/2) ((1 - j)/2 - C(t) + j S(t)) with
// t = sqrt(x) / sqrt(pi/2).  With the integrals written through their auxiliary functions (rt_fresnel.h) the bracket is
// (g - j f) e^{-jx}: the exponentials cancel and F = 2 sqrt(x) sqrt(pi/2) (f + j g), without sine, cosine or the
// cancellation of C and S against 1/2.  Below RT_FRESNEL_X0 the integrals themselves are used, as written.
__device__ inline cd transition_F(double x) {
    const double factor = sqrt(M_PI / 2);
    const double sqrtx = sqrt(x);
    const double t = sqrtx / factor;
    if (t >= RT_FRESNEL_X0) {
        double f, g;
        rt_fresnel_fg(t, &f, &g);
        const double w = 2.0 * sqrtx * factor;
        return cd{w * f, w * g};
    }
    double S, C;
    rt_fresnel_small(t, &S, &C);
    const cd inner{factor * (0.5 - C), factor * (-0.5 + S)};
    const cd ex = cexpj(x);
    return cmul(cmul(cd{0.0, 2.0 * sqrtx}, ex), inner);
}
__device__ inline double utd_A(double x, double n, bool plus) {  // electromagnetism.py:383-393, Python round = rint
    const double N = rint(((plus ? M_PI : -M_PI) + x) / (2 * n * M_PI));
    const double c = cos(n * M_PI * N - x / 2);
    return 2 * c * c;
}
__device__ inline double py_mod_2pi(double a) {
    double r = fmod(a, 2 * M_PI);
    if (r < 0) r += 2 * M_PI;
    return r;
}
__device__ inline double sgn(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }

// One wedge diffraction (electromagnetism.py:395-507)
struct DiffGeom {
    d3 b_i, phi_i_vv, b_d, phi_d_vv;
    double n, sin_b, L, A[4], cot[4], spread, Sd;  // spread = sqrt(Si / (Sd (Sd + Si)))
    int status;
};
__device__ inline DiffGeom diffract_geometry(const SceneView &sc, d3 p0, d3 p1, d3 p2, int poly0, int poly1, d3 c0, d3 c1) {
    DiffGeom g;
    int status = ST_OK;
    const d3 v0 = p1 - p0, v1 = p2 - p1;
    const double Si = norm3(v0), Sd = norm3(v1);
    const d3 si = divs(v0, Si), sd = divs(v1, Sd);
    const d3 n_0 = ld3(sc.planes + 4 * (size_t)poly0), n_1 = ld3(sc.planes + 4 * (size_t)poly1);
    d3 e = vv_normalize(c1 - c0);
    d3 t_0 = vv_normalize(cross(n_0, e));
    const d3 mid = mk3((c0.x + c1.x) / 2, (c0.y + c1.y) / 2, (c0.z + c1.z) / 2);
    if (!polygon_contains_point(sc, poly0, mid + 1e-6 * t_0, true, 1e-8)) {
        e = mk3(-e.x, -e.y, -e.z);
        t_0 = vv_normalize(cross(n_0, e));
    }
    if (!polygon_contains_point(sc, poly0, mid + 1e-6 * t_0, true, 1e-8)) {
        if (status == ST_OK) status = ST_EDGE_ORIENTATION;
    }
    const double alpha = M_PI - acos(dot(n_0, n_1));
    const double n = 2 - alpha / M_PI;
    if (!(0 < alpha && alpha < M_PI && 1 <= n && n <= 2)) {
        if (status == ST_OK) status = ST_WEDGE;
    }
    const double beta_0 = acos(dot(si, e));
    if (!(0 <= beta_0 && beta_0 <= M_PI)) {
        if (status == ST_OK) status = ST_BETA;
    }
    const d3 si_t = vv_normalize(si - dot(si, e) * e);
    const d3 sd_t = vv_normalize(sd - dot(sd, e) * e);
    const d3 msi_t = mk3(-si_t.x, -si_t.y, -si_t.z);
    double phi_i = M_PI - (M_PI - acos(dot(msi_t, t_0))) * sgn(dot(msi_t, n_0));
    double phi_d = M_PI - (M_PI - acos(dot(sd_t, t_0))) * sgn(dot(sd_t, n_0));
    phi_i = py_mod_2pi(phi_i);
    phi_d = py_mod_2pi(phi_d);
    if (!(0 <= phi_i && phi_i <= n * M_PI && 0 <= phi_d && phi_d <= n * M_PI)) {
        if (status == ST_OK) status = ST_INSIDE_WEDGE;
    }
    const double sb = sin(beta_0);
    g.n = n;
    g.sin_b = sb;
    g.L = sb * sb * Si * Sd / (Si + Sd);
    const double dm = phi_d - phi_i, dp = phi_d + phi_i;
    g.cot[0] = 1.0 / tan((M_PI + dm) * 1 / (2 * n));
    g.cot[1] = 1.0 / tan((M_PI - dm) * 1 / (2 * n));
    g.cot[2] = 1.0 / tan((M_PI + dp) * 1 / (2 * n));
    g.cot[3] = 1.0 / tan((M_PI - dp) * 1 / (2 * n));
    g.A[0] = utd_A(dm, n, true);
    g.A[1] = utd_A(dm, n, false);
    g.A[2] = utd_A(dp, n, true);
    g.A[3] = utd_A(dp, n, false);
    g.phi_i_vv = vv_normalize(cross(mk3(-e.x, -e.y, -e.z), si));
    g.phi_d_vv = vv_normalize(cross(e, sd));
    g.b_i = vv_normalize(cross(g.phi_i_vv, si));
    g.b_d = vv_normalize(cross(g.phi_d_vv, sd));
    g.spread = sqrt(Si / (Sd * (Sd + Si)));
    g.Sd = Sd;
    g.status = status;
    return g;
}
__device__ inline c3 diffract_apply(c3 E, const DiffGeom &g, double K, int &status) {
    if (status == ST_OK) status = g.status;
    // common_factor = -exp(-j pi/4) / (2 n sqrt(2 pi K) sin(beta_0))
    const double cf_mag = 1.0 / (2 * g.n * sqrt(2 * M_PI * K) * g.sin_b);
    const cd cf = cscale(-cf_mag, cexpj(-M_PI / 4));
    cd Dsum{0, 0};  // D_per = D_par (r_per = r_par = 1, :483-487); (D1 + D2) + (D3 + D4)
    cd D[4];
    for (int q = 0; q < 4; ++q) D[q] = cmul(cscale(g.cot[q], cf), transition_F(K * g.L * g.A[q]));
    Dsum = cadd(cadd(D[0], D[1]), cadd(D[2], D[3]));
    // E @ D, D = b_i b_d^T (-D_per) - phi_i phi_d^T D_par
    const cd a = cmul(cdot_real(E, g.b_i), cd{-Dsum.re, -Dsum.im});
    const cd b = cmul(cdot_real(E, g.phi_i_vv), cd{-Dsum.re, -Dsum.im});
    const cd ph = cscale(g.spread, cexpj(-K * g.Sd));
    c3 out;
    out.x = cmul(cadd(cscale(g.b_d.x, a), cscale(g.phi_d_vv.x, b)), ph);
    out.y = cmul(cadd(cscale(g.b_d.y, a), cscale(g.phi_d_vv.y, b)), ph);
    out.z = cmul(cadd(cscale(g.b_d.z, a), cscale(g.phi_d_vv.z, b)), ph);
    if (!(cnorm3(E) >= cnorm3(out))) {
        if (status == ST_OK) status = ST_DIFFRACTED_GAIN;
    }
    return out;
}

// Points of path i: P[0] = tx, interaction points, P[np-1] = receiver.  Returns the number of points.
__device__ inline int path_points(const PathArrays &pa, int64_t i, d3 tx, const double *rx, d3 *P) {
    const int kind = pa.kind[i];
    const int n_int = (kind & 255) + (kind >> 8);
    P[0] = pa.origin ? ld3(pa.origin + 3 * (size_t)i) : tx;
    for (int k = 0; k < n_int; ++k) P[1 + k] = ld3(pa.pts + (size_t)i * RT_MAX_ORDER * 3 + 3 * k);
    P[1 + n_int] = ld3(rx + 3 * (size_t)pa.rx[i]);
    return n_int + 2;
}

__device__ inline double path_key_length(const PathArrays &pa, int64_t i, d3 tx, const double *rx) {
    // a line of sight counts as length 0: it always wins (electromagnetism.py:615-617)
    if (pa.kind[i] == 0) return 0.0;
    d3 P[RT_MAX_ORDER + 2];
    const int np = path_points(pa, i, tx, rx, P);
    double len = 0;
    for (int k = 0; k + 1 < np; ++k) len += norm3(P[k + 1] - P[k]);
    return len;
}

// Loss the reference applies to a segment that crosses geometry: whatever it crosses is treated as tree crowns
// (electromagnetism.py:179-199, geometry.py:868-921).  Polygons p_begin.. are looked at: all of them for a line of
// sight (it was only checked against the polygons the emitter sees), and for the other segments the polygons added
// after the paths were found (the crowns put on the trunks, electromagnetism.py:745) — the solve has already found
// those segments clear of every other polygon with the same test.
__device__ inline double tree_factor(const SceneView &sc, d3 A, d3 B, double freq, int &status, int p_begin) {
    double factor = 1.0;
    int first = -1, second = -1, count = 0;
    auto flush = [&]() {
        if (count == 2) {
            // geometry.py:89-106 for both polygons, then the ITU tree loss
            d3 dir = vv_normalize(B - A);
            d3 pts[2];
            const int idx[2] = {first, second};
            for (int q = 0; q < 2; ++q) {
                const d3 n = ld3(sc.planes + 4 * (size_t)idx[q]);
                const d3 pp = ld3(sc.verts + 3 * ((size_t)sc.vert_off[idx[q]] + 1));
                const d3 w = A - pp;
                const double s = -dot(w, n) / dot(dir, n);
                pts[q] = (w + s * dir) + pp;
            }
            const double d = norm3(pts[0] - pts[1]);
            if (d > 0) {
                const double L_db = 0.39 * pow(freq / 1e6, 0.39) * pow(d, 0.25);
                factor *= pow(10.0, -L_db / 10);
            }
        } else if (count > 2) {
            if (status == ST_OK) status = ST_CONCAVE;
        }
    };
    const d3 V = B - A;
    const double pad = 1e-5 * fmax(fabs(V.x), fmax(fabs(V.y), fabs(V.z))) + 1e-6;
    for (int p = p_begin; p < sc.n; ++p) {
        // the cheap conservative test first; the exact one only for the few polygons it lets through
        if (!occ_pretest(sc.poly_recs + p, sc, A, V, pad) || !polygon_cuts_segment(sc, p, A, B)) continue;
        if (count == 0) {
            first = p;
            count = 1;
        } else if (sc.building[first] == sc.building[p]) {
            if (count == 1) second = p;
            ++count;
        } else {
            flush();  // geometry.py:890-892: the group is closed and the polygon that closed it is dropped
            count = 0;
        }
    }
    flush();
    return factor;
}

// What tree_factor finds for a segment, kept apart from the carrier frequency: the lengths of the stretches inside
// the groups of two crossed polygons of one building, in the order they are met, and whether a group of more than two
// turned up (the reference's "Cannot handle concave geometries").
#define TREE_MAX_GROUPS 4
struct TreeHits {
    double d[TREE_MAX_GROUPS];
    int32_t n;        // stretches recorded
    int32_t concave;  // 1 where the reference would raise
};

__device__ inline double tree_factor_of(const TreeHits &h, double freq) {
    double factor = 1.0;
    for (int q = 0; q < h.n; ++q) {
        const double L_db = 0.39 * pow(freq / 1e6, 0.39) * pow(h.d[q], 0.25);
        factor *= pow(10.0, -L_db / 10);
    }
    return factor;
}

// First pass over the paths of a batch, a thread per path:
//  - the paths by what the field kernel has to do for them: lists[0] lines of sight, [1] reflections only, [2] paths
//    ending with a diffraction (full waves of like work; a block reserves the room of its entries with one atomic per
//    class);
//  - the shortest length per receiver (the antennas are aimed along that path, electromagnetism.py:610-644);
//  - the number of paths per receiver.
__global__ __launch_bounds__(256) void k_fields_pass1(PathArrays pa, d3 tx, const double *__restrict__ rx,
                                                       unsigned long long *best_len, int32_t *__restrict__ lists,
                                                       unsigned long long *counts, int32_t *rx_counts) {
    __shared__ int s_n[3];
    __shared__ unsigned long long s_base[3];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 3) s_n[threadIdx.x] = 0;
    __syncthreads();
    int cls = -1, slot = 0;
    if (i < pa.n) {
        const int kind = pa.kind[i];
        cls = (kind >> 8) ? 2 : ((kind & 255) ? 1 : 0);
        slot = atomicAdd(&s_n[cls], 1);
        const int r = pa.rx[i];
        atomicMin(&best_len[r], (unsigned long long)__double_as_longlong(path_key_length(pa, i, tx, rx)));
        atomicAdd(&rx_counts[r], 1);
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_n[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], (unsigned long long)s_n[threadIdx.x]);
    __syncthreads();
    if (cls >= 0) lists[(size_t)cls * pa.n + s_base[cls] + slot] = (int32_t)i;
}

// Lines of sight are only known to be clear of the polygons the emitter sees (ray_tracing.py:232-237), so the reference's
// tree loss looks at every polygon of the scene for them (electromagnetism.py:669).  One workgroup per line of sight at
// a time (the workgroups stride over the list, whose length only the device knows): its four waves take a quarter of
// the polygons each, lanes = polygons, the crossed ones gathered per wave in ascending order -- so the waves' lists one
// after the other are in ascending order too --, then the grouping of geometry.py:886-895 by one lane.  (One wave per
// line of sight left two waves on a SIMD with nothing to hide the record reads behind: 100 us for lln's 2 000 lines.)
#define LOS_HIT_CAP 128
// The look-up itself, by a workgroup of WAVES waves (every thread of the workgroup calls it; `first`, `step`: which
// entries of the list this workgroup takes).
template <int WAVES>
__device__ __forceinline__ void los_tree_hits(const SceneView &sc, const PathArrays &pa, d3 tx, const double *__restrict__ rx,
                                              const int32_t *__restrict__ los_list, const unsigned long long *los_count,
                                              TreeHits *__restrict__ out, int64_t first, int64_t step) {
    __shared__ int32_t s_hits[WAVES][LOS_HIT_CAP];
    __shared__ int s_n[WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_los = (int64_t)*los_count;
    const int per_wave = ((sc.n + WAVES - 1) / WAVES + 63) & ~63;  // polygons per wave, whole trips of 64
    for (int64_t j = first; j < n_los; j += step) {
        const int64_t i = los_list[j];
        const d3 A = pa.origin ? ld3(pa.origin + 3 * (size_t)i) : tx;
        const d3 B = ld3(rx + 3 * (size_t)pa.rx[i]);
        const d3 V = B - A;
        const double pad = 1e-5 * fmax(fabs(V.x), fmax(fabs(V.y), fabs(V.z))) + 1e-6;
        int n_mine = 0;  // wave-uniform
        const int p_end = (wave + 1) * per_wave < sc.n ? (wave + 1) * per_wave : sc.n;
        for (int base = wave * per_wave; base < p_end; base += 64) {
            const int p = base + lane;
            const bool hit = p < p_end && occ_pretest(sc.poly_recs + p, sc, A, V, pad) && polygon_cuts_segment(sc, p, A, B);
            const unsigned long long mk = __ballot(hit);
            if (hit) {
                const int at = n_mine + __popcll(mk & ((1ull << lane) - 1));
                if (at < LOS_HIT_CAP) s_hits[wave][at] = p;
            }
            n_mine += __popcll(mk);
        }
        if (lane == 0) s_n[wave] = n_mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            int n_hits = 0;
            bool too_many = false;
            for (int w = 0; w < WAVES; ++w) {
                n_hits += s_n[w];
                too_many = too_many || s_n[w] > LOS_HIT_CAP;
            }
            // (a wave that finds more than LOS_HIT_CAP on its own has kept its first LOS_HIT_CAP; the path is then marked as
            // not to be trusted below, and what is grouped does not matter)
            too_many = too_many || n_hits > LOS_HIT_CAP;
            // the q-th crossed polygon in ascending order (q below LOS_HIT_CAP, which bounds what is looked at)
            auto hit_at = [&](int q) -> int {
                int w = 0;
                while (q >= s_n[w]) q -= s_n[w++];
                return s_hits[w][q];
            };
            TreeHits h;
            h.n = 0;
            h.concave = 0;
            int first_p = -1, second_p = -1, count = 0;
            auto flush = [&]() {
                if (count == 2) {
                    // geometry.py:89-106 for both polygons
                    const d3 dir = vv_normalize(B - A);
                    d3 pts[2];
                    const int idx[2] = {first_p, second_p};
                    for (int q = 0; q < 2; ++q) {
                        const d3 n = ld3(sc.planes + 4 * (size_t)idx[q]);
                        const d3 pp = ld3(sc.verts + 3 * ((size_t)sc.vert_off[idx[q]] + 1));
                        const d3 w = A - pp;
                        const double s = -dot(w, n) / dot(dir, n);
                        pts[q] = (w + s * dir) + pp;
                    }
                    const double d = norm3(pts[0] - pts[1]);
                    if (d > 0) {
                        if (h.n < TREE_MAX_GROUPS) h.d[h.n++] = d; else h.concave = 1;
                    }
                } else if (count > 2) {
                    h.concave = 1;
                }
            };
            const int m = too_many ? 0 : n_hits;
            for (int q = 0; q < m; ++q) {
                const int p = hit_at(q);
                if (count == 0) {
                    first_p = p;
                    count = 1;
                } else if (sc.building[first_p] == sc.building[p]) {
                    if (count == 1) second_p = p;
                    ++count;
                } else {
                    flush();  // geometry.py:890-892: the group is closed and the polygon that closed it is dropped
                    count = 0;
                }
            }
            flush();
            if (too_many) h.concave = 1;  // (a segment through more than 128 polygons: not a line of sight of any scene)
            out[j] = h;
        }
        __syncthreads();  // the lists are free for the next line of sight
    }
}
// Second pass, workgroups of 1 024 threads with three kinds of work side by side:
//  - blocks [0, path_blocks): among a receiver's equally short paths (exact ties only) the first in the order the paths
//    were given;
//  - block path_blocks: where each receiver's paths start in receiver order (running sum of the counts);
//  - the blocks after it: the tree look-up of the lines of sight (whose list the first pass has made), a line at a time.
__global__ __launch_bounds__(1024) void k_fields_pass2(SceneView sc, PathArrays pa, d3 tx, const double *__restrict__ rx,
                                                        const unsigned long long *__restrict__ best_len,
                                                        unsigned long long *best_idx, const int32_t *__restrict__ rx_counts,
                                                        int64_t n_rx, int64_t *__restrict__ rx_off,
                                                        const int32_t *__restrict__ los_list, const unsigned long long *los_count,
                                                        TreeHits *__restrict__ los_trees, int path_blocks) {
    if ((int)blockIdx.x < path_blocks) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= pa.n) return;
        if ((unsigned long long)__double_as_longlong(path_key_length(pa, i, tx, rx)) == best_len[pa.rx[i]])
            atomicMin(&best_idx[pa.rx[i]], (unsigned long long)i);
    } else if ((int)blockIdx.x == path_blocks) {
        __shared__ long long part[1024];
        __shared__ long long carry;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int64_t base = 0; base < n_rx; base += 1024) {
            const int64_t r = base + threadIdx.x;
            const long long v = r < n_rx ? rx_counts[r] : 0;
            part[threadIdx.x] = v;
            __syncthreads();
            for (int o = 1; o < 1024; o <<= 1) {
                const long long add = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
                __syncthreads();
                part[threadIdx.x] += add;
                __syncthreads();
            }
            if (r < n_rx) rx_off[r] = carry + part[threadIdx.x] - v;
            __syncthreads();
            if (threadIdx.x == 0) carry += part[1023];
            __syncthreads();
        }
        if (threadIdx.x == 0) rx_off[n_rx] = carry;
    } else if (!pa.e_in) {
        los_tree_hits<16>(sc, pa, tx, rx, los_list, los_count, los_trees, (int64_t)blockIdx.x - path_blocks - 1,
                          (int64_t)gridDim.x - path_blocks - 1);
    }
}

// Per-receiver power: the powers of a receiver's paths added in the order of the reference's table (line of sight,
// reflections by order and chain, diffractions by order, chain and edge; electromagnetism.py:776-778), whatever order
// the paths arrive in: the same map gives the same dB to the last bit on every run and every split over GPUs.
// What the reference's row order looks at, per path, in the order it looks at it: kind (0 line of sight, 1.. reflections,
// 256.. diffractions by the reflections before them), the reflecting polygons, the edge's two polygons, and the path's
// own index last.  Entries a kind does not have are zero, so that the order is the lexicographic order of the eight.
struct __attribute__((aligned(16))) PathKey {
    unsigned long long w[4];  // the eight in pairs, the earlier one in the upper half: (kind, chain 0), (chain 1, chain 2),
                              // (chain 3, edge 0), (edge 1, index); none of them is negative
    __device__ inline int32_t index() const { return (int32_t)(w[3] & 0xffffffffull); }
};
static_assert(RT_MAX_ORDER == 4, "PathKey holds kind, four chain entries, two edge polygons and the index");
// The keys of path i put among its receiver's (in the order the threads happen to arrive).
__device__ inline void rx_path_fill(const PathArrays &pa, int64_t i, const int64_t *__restrict__ off, int32_t *cursor,
                                    PathKey *__restrict__ keys) {
    if (i >= pa.n) return;
    const int r = pa.rx[i];
    const int64_t slot = off[r] + atomicAdd(&cursor[r], 1);
    const int kind = pa.kind[i], n_refl = kind & 255;
    unsigned v[8];
    v[0] = (unsigned)kind;
    for (int k = 0; k < RT_MAX_ORDER; ++k) v[1 + k] = k < n_refl ? (unsigned)pa.chain[(size_t)i * RT_MAX_ORDER + k] : 0u;
    for (int k = 0; k < 2; ++k) v[5 + k] = (kind >> 8) ? (unsigned)pa.edge_polys[2 * (size_t)i + k] : 0u;
    v[7] = (unsigned)i;
    PathKey key;
    for (int k = 0; k < 4; ++k) key.w[k] = ((unsigned long long)v[2 * k] << 32) | v[2 * k + 1];
    keys[slot] = key;
}
// lexicographic order of the eight, without branches (the index, last, makes it a strict order)
__device__ inline bool key_before(const PathKey &a, const PathKey &b) {
    bool before = a.w[3] < b.w[3];
    before = (a.w[2] < b.w[2]) | ((a.w[2] == b.w[2]) & before);
    before = (a.w[1] < b.w[1]) | ((a.w[1] == b.w[1]) & before);
    before = (a.w[0] < b.w[0]) | ((a.w[0] == b.w[0]) & before);
    return before;
}
struct FieldOut {
    double *field;    // [n_paths * n_freq * 6]
    double *power;    // [n_paths * n_freq]
    double *angles;   // [n_paths * 4] degrees: tx elevation, tx azimuth, rx elevation, rx azimuth
    double *time;     // [n_paths]
    int32_t *status;  // [n_paths]
    double *power_sum;// [n_rx * n_freq]
};

// What a path's geometry half leaves for its frequency half when the two run in different kernels: slot q of path i is
// geom[q * stride + i].  A value passes through `io` in the same order on both sides.
#define GEOM_COMMON 26
#define GEOM_PER_REFLECTION 14
#define GEOM_DIFFRACTION 26
#define GEOM_SLOTS (GEOM_COMMON + GEOM_PER_REFLECTION * RT_MAX_ORDER + GEOM_DIFFRACTION)
enum FieldMode {
    FIELDS_WHOLE = 0,     // a thread per (path, frequency) does both halves
    FIELDS_GEOMETRY = 1,  // a thread per path writes its geometry
    FIELDS_FREQUENCY = 2  // a thread per (path, frequency) reads it
};
template <int MODE>
struct GeomIO {
    double *at;  // slot 0 of this path
    int64_t stride;
    bool on;  // FIELDS_GEOMETRY: this thread has computed what passes now and writes it
    __device__ inline void io(double &v) {
        if (MODE == FIELDS_GEOMETRY && on) *at = v;
        if (MODE == FIELDS_FREQUENCY) v = *at;
        at += stride;
    }
    __device__ inline void io(d3 &v) {
        io(v.x);
        io(v.y);
        io(v.z);
    }
    __device__ inline void io(int &a, int &b) {  // two small non-negative integers in one slot
        double v = (double)(a * 256 + b);
        io(v);
        if (MODE == FIELDS_FREQUENCY) {
            a = (int)v >> 8;
            b = (int)v & 255;
        }
    }
    __device__ inline void skip(int slots) { at += (int64_t)slots * stride; }
};

// CLASS: 0 lines of sight, 1 reflections only, 2 paths ending with a diffraction (k_fields_pass1); a thread takes entry
// blockIdx.x * blockDim.x + threadIdx.x of its class's list, and in the modes with a thread per (path, frequency)
// blockIdx.y is the frequency.  In FIELDS_WHOLE what the path alone decides (antenna frames, angles, the geometry of
// every interaction) is redone for every frequency; a sweep over several frequencies goes through FIELDS_GEOMETRY once
// per path and FIELDS_FREQUENCY per frequency instead.  Segments through tree crowns (n_late) need FIELDS_WHOLE.
template <int CLASS, int MODE>
__device__ __forceinline__ void fields_of_path(const SceneView &sc, const PathArrays &pa, const FreqTable &ft,
                                               const rt_antenna_params &ap, d3 tx, const double *__restrict__ rx,
                                               const unsigned long long *__restrict__ best_idx, const int32_t *__restrict__ list,
                                               const unsigned long long *list_count, const TreeHits *__restrict__ los_trees,
                                               const FieldOut &out, double *geom, int64_t geom_stride, int64_t j,
                                               int f_or_part) {
    if (j >= (int64_t)*list_count) return;
    const int64_t i = list[j];
    const int f = MODE == FIELDS_GEOMETRY ? 0 : f_or_part;
    const int kind = pa.kind[i];
    const int n_refl = CLASS == 0 ? 0 : (kind & 255);
    const bool has_edge = CLASS == 2;
    // FIELDS_GEOMETRY: blockIdx.y is the part of the path's geometry this thread computes (the parts are independent
    // of each other, and a path's geometry is a chain of dependent operations that lasts as long as its longest part):
    // 0 antennas, angles and source, 1 + k the k-th reflection, RT_MAX_ORDER + 1 the diffraction
    const int part = MODE == FIELDS_GEOMETRY ? f_or_part : -1;
    if (MODE == FIELDS_GEOMETRY && ((part >= 1 && part <= RT_MAX_ORDER && part - 1 >= n_refl) || (part == RT_MAX_ORDER + 1 && !has_edge)))
        return;
    auto mine = [&](int what) { return MODE == FIELDS_WHOLE || (MODE == FIELDS_GEOMETRY && part == what); };
    GeomIO<MODE> g{geom + i, geom_stride, false};

    d3 P[RT_MAX_ORDER + 2];
    int np = 0;
    d3 tx_ax, tx_ay, tx_az, rx_ax, rx_ay, rx_az;
    double rx_pattern = 0;
    int status = ST_OK;
    SourceGeom src;
    src.status = ST_OK;
    if (MODE != FIELDS_FREQUENCY) np = path_points(pa, i, tx, rx, P);
    g.on = mine(0);
    if (mine(0)) {
        const int r = pa.rx[i];
        const d3 R = P[np - 1];
        // antennas aligned on the receiver's shortest path (electromagnetism.py:768-771)
        d3 aim_t, aim_r;
        if (pa.aim_tx && pa.aim_rx) {
            aim_t = ld3(pa.aim_tx + 3 * (size_t)r);
            aim_r = ld3(pa.aim_rx + 3 * (size_t)r);
        } else {
            d3 Q[RT_MAX_ORDER + 2];
            const int nq = path_points(pa, (int64_t)best_idx[r], tx, rx, Q);
            aim_t = Q[1];
            aim_r = Q[nq - 2];
        }
        const AntennaFrame txa = align_antenna(tx, aim_t);
        const AntennaFrame rxa = align_antenna(R, aim_r);
        double len = 0;
        for (int k = 0; k + 1 < np; ++k) len += norm3(P[k + 1] - P[k]);
        double rr, tx_el = 0, tx_az_deg = 0, rx_el = 0, rx_az_deg = 0;
        if (!to_spherical(w2a(txa, P[1] - tx), rr, tx_el, tx_az_deg)) status = ST_ORIGIN;
        if (!to_spherical(w2a(rxa, P[np - 2] - R), rr, rx_el, rx_az_deg)) status = ST_ORIGIN;
        if (f == 0) {
            const double deg = 180.0 / M_PI;
            out.time[i] = len / 299792458.0;
            out.angles[4 * i + 0] = tx_el * deg;
            out.angles[4 * i + 1] = tx_az_deg * deg;
            out.angles[4 * i + 2] = rx_el * deg;
            out.angles[4 * i + 3] = rx_az_deg * deg;
        }
        if (!pa.e_in) src = source_geometry(txa, P[1], ap);
        tx_ax = txa.ax; tx_ay = txa.ay; tx_az = txa.az;
        rx_ax = rxa.ax; rx_ay = rxa.ay; rx_az = rxa.az;
        rx_pattern = radiation_pattern(rx_el, ap);
    }
    g.io(src.v); g.io(src.r); g.io(src.pattern);
    g.io(status, src.status);
    g.io(tx_ax); g.io(tx_ay); g.io(tx_az);
    g.io(rx_ax); g.io(rx_ay); g.io(rx_az);
    g.io(rx_pattern);
    static_assert(GEOM_COMMON == 3 + 2 + 1 + 9 + 9 + 1 + 1, "one slot spare");

    const double K = ft.K[f];
    int st = status;
    c3 E{{0, 0}, {0, 0}, {0, 0}};
    double e0 = 0;
    if (MODE != FIELDS_GEOMETRY) {
        if (pa.e_in) {
            const double *q = pa.e_in + 6 * (size_t)i;
            E = c3{{q[0], q[1]}, {q[2], q[3]}, {q[4], q[5]}};
        } else {
            E = source_apply(src, tx_ax, tx_ay, tx_az, K, st);
        }
        e0 = cnorm3(E);
    }
    const int late_begin = sc.n - pa.n_late;
    auto through_trees = [&](int from, int to, int p_begin) {  // the segment between two of the path's points
        if (MODE != FIELDS_WHOLE || p_begin >= sc.n) return;
        const double fac = tree_factor(sc, P[from], P[to], ft.freq[f], st, p_begin);
        E = c3{cscale(fac, E.x), cscale(fac, E.y), cscale(fac, E.z)};
    };
    if (CLASS == 0 && !pa.e_in && MODE != FIELDS_GEOMETRY) {  // electromagnetism.py:669: every polygon of the scene, found by los_tree_hits
        const TreeHits h = los_trees[j];
        if (h.concave && st == ST_OK) st = ST_CONCAVE;
        const double fac = tree_factor_of(h, ft.freq[f]);
        E = c3{cscale(fac, E.x), cscale(fac, E.y), cscale(fac, E.z)};
    }
    for (int k = 0; k < n_refl; ++k) {
        through_trees(k, k + 1, late_begin);  // :338
        ReflGeom rg{};
        g.on = mine(1 + k);
        if (mine(1 + k)) rg = reflect_geometry(sc, P[k], P[k + 1], P[k + 2], pa.chain[(size_t)i * RT_MAX_ORDER + k]);
        g.io(rg.e_per); g.io(rg.ei_par); g.io(rg.er_par);
        g.io(rg.sin_i); g.io(rg.cosi); g.io(rg.spread); g.io(rg.Sr);
        g.io(rg.material, rg.status);
        if (MODE != FIELDS_GEOMETRY) {
            E = reflect_apply(E, rg, ft, f, st);
            if (!(e0 >= cnorm3(E)) && st == ST_OK) st = ST_REFLECTED_GAIN;  // electromagnetism.py:342 (E_i is the first field)
        }
    }
    if (n_refl > 0) through_trees(n_refl, n_refl + 1, late_begin);  // :346-348
    if (has_edge) {
        if (n_refl == 0 && !pa.e_in) through_trees(0, 1, late_begin);  // :724-725
        DiffGeom dg{};
        g.on = mine(RT_MAX_ORDER + 1);
        if (mine(RT_MAX_ORDER + 1)) {
            const d3 c0 = ld3(pa.edge_pts + 6 * (size_t)i), c1 = ld3(pa.edge_pts + 6 * (size_t)i + 3);
            dg = diffract_geometry(sc, P[np - 3], P[np - 2], P[np - 1], pa.edge_polys[2 * i], pa.edge_polys[2 * i + 1], c0, c1);
        }
        g.io(dg.b_i); g.io(dg.phi_i_vv); g.io(dg.b_d); g.io(dg.phi_d_vv);
        g.io(dg.n); g.io(dg.sin_b); g.io(dg.L);
        for (int q = 0; q < 4; ++q) g.io(dg.A[q]);
        for (int q = 0; q < 4; ++q) g.io(dg.cot[q]);
        g.io(dg.spread); g.io(dg.Sd);
        int none = 0;
        g.io(none, dg.status);
        if (MODE != FIELDS_GEOMETRY) E = diffract_apply(E, dg, K, st);
        through_trees(np - 2, np - 1, late_begin);  // :506
    }
    if (MODE == FIELDS_GEOMETRY) return;
    // received power (electromagnetism.py:779-787)
    const double Ae = (ft.lambda[f] * ft.lambda[f] / (4 * M_PI)) * ap.rx_gain * rx_pattern;
    c3 Ea;
    Ea.x = cdot_real(E, rx_ax);
    Ea.y = cdot_real(E, rx_ay);
    Ea.z = cdot_real(E, rx_az);
    const double en = cnorm3(Ea);
    const d3 pol_a = mk3(dot(rx_ax, rx_ax), dot(rx_ax, rx_ay), dot(rx_ax, rx_az));  // the x axis in the antenna's own frame
    cd pe{0, 0};
    if (en != 0) pe = cscale(1.0 / en, cdot_real(Ea, pol_a));
    const double pol_eff = pe.re * pe.re + pe.im * pe.im;
    const double re2 = Ea.x.re * Ea.x.re + Ea.y.re * Ea.y.re + Ea.z.re * Ea.z.re;
    const double Z0 = 120.0 * M_PI;
    double power = 1 / (2 * Z0) * Ae * re2;
    power = power * pol_eff * ap.tx_gain;
    double *fo = out.field + ((size_t)i * ft.n_freq + f) * 6;
    fo[0] = E.x.re; fo[1] = E.x.im; fo[2] = E.y.re; fo[3] = E.y.im; fo[4] = E.z.re; fo[5] = E.z.im;
    out.power[(size_t)i * ft.n_freq + f] = power;
    // the status of a path is that of the lowest frequency at which something went wrong (the order in which a
    // loop over the frequencies would have met it): keep the smallest (frequency, status) key
    if (st != ST_OK) atomicMin(&out.status[i], (f << 8) | st);
}

// The three classes in one launch, side by side: blockIdx.z is the class (lists[c * n] with counts[c]); the grid is sized
// for the whole batch per class and the blocks past the end of a list leave at once.
//   FIELDS_WHOLE      grid (entries / 128, frequencies, 3)
//   FIELDS_GEOMETRY   grid (entries / 128, parts of a path's geometry, 3); nothing to do for the lines of sight
//   FIELDS_FREQUENCY  grid (entries / (128 / frequencies), 1, 3): the frequencies of a path are neighbouring lanes, so a
//                     wave reads the geometry of its nine or so paths once instead of once per frequency and writes each
//                     path's fields as one run of bytes; the lines of sight go through FIELDS_WHOLE here
// With a fourth slice in z (FIELDS_WHOLE, FIELDS_FREQUENCY) the keys of the paths are put in receiver order beside the
// fields (rx_path_fill), a thread per path.
#define FIELDS_GEOMETRY_PARTS (RT_MAX_ORDER + 2)
template <int MODE>
__global__ __launch_bounds__(128) void k_fields(SceneView sc, PathArrays pa, FreqTable ft, rt_antenna_params ap, d3 tx,
                                                 const double *__restrict__ rx,
                                                 const unsigned long long *__restrict__ best_idx,
                                                 const int32_t *__restrict__ lists, const unsigned long long *counts,
                                                 TreeHits *__restrict__ los_trees, FieldOut out, double *geom,
                                                 int64_t geom_stride, const int64_t *__restrict__ rx_off, int32_t *rx_cursor,
                                                 PathKey *__restrict__ rx_keys) {
    const int cls = blockIdx.z;
    if (cls == 3) {  // (not in the FIELDS_GEOMETRY launch) beside the fields: the paths' keys in receiver order
        if (blockIdx.y == 0) rx_path_fill(pa, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, rx_off, rx_cursor, rx_keys);
        return;
    }
    const int32_t *list = lists + (size_t)cls * pa.n;
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int f_or_part = blockIdx.y;
    if (MODE == FIELDS_FREQUENCY) {
        const int per_block = (int)blockDim.x / ft.n_freq, local = (int)threadIdx.x / ft.n_freq;
        if (local >= per_block) return;
        f_or_part = (int)threadIdx.x - local * ft.n_freq;
        j = (int64_t)blockIdx.x * per_block + local;
    }
    if (cls == 0) {
        if (MODE != FIELDS_GEOMETRY) {
            fields_of_path<0, FIELDS_WHOLE>(sc, pa, ft, ap, tx, rx, best_idx, list, counts, los_trees, out, geom, geom_stride, j,
                                            f_or_part);
        }
    } else if (cls == 1) {
        fields_of_path<1, MODE>(sc, pa, ft, ap, tx, rx, best_idx, list, counts + 1, los_trees, out, geom, geom_stride, j, f_or_part);
    } else {
        fields_of_path<2, MODE>(sc, pa, ft, ap, tx, rx, best_idx, list, counts + 2, los_trees, out, geom, geom_stride, j, f_or_part);
    }
}

#define STATUS_UNSET 0x7f7f7f7f /* what hipMemset(0x7f) leaves: larger than any (frequency, status) key */
// Place of every path among its receiver's paths in the reference's row order: a thread per path counts the keys before
// its own (a receiver has a handful to a few hundred paths).  Threads go by the place in the filled list, so a workgroup
// holds runs of paths of the same receivers and the keys it has to look at are one stretch of the list, from the first
// key of its first thread's receiver to the last key of its last thread's: staged through LDS a tile at a time, every
// thread reading the part of the tile that belongs to its receiver.  (Each thread reading its receiver's keys from
// memory, one after the other, waited for memory at every key: 88 us for 72 000 paths.)
// The blocks after rank_blocks turn each path's smallest (frequency, status) key into its status.
#define RANK_TILE 256
__global__ __launch_bounds__(RANK_TILE) void k_rx_path_rank(PathArrays pa, const int64_t *__restrict__ off,
                                                             const PathKey *__restrict__ keys, int32_t *__restrict__ ordered,
                                                             int32_t *__restrict__ status, int rank_blocks) {
    if ((int)blockIdx.x >= rank_blocks) {
        const int64_t i = (int64_t)(blockIdx.x - rank_blocks) * blockDim.x + threadIdx.x;
        if (i >= pa.n) return;
        const int32_t key = status[i];
        status[i] = key == STATUS_UNSET ? ST_OK : (key & 255);
        return;
    }
    __shared__ PathKey s_keys[RANK_TILE];
    __shared__ int64_t s_span[2];
    const int64_t t = (int64_t)blockIdx.x * RANK_TILE + threadIdx.x;
    const bool live = t < pa.n;
    PathKey mine{};
    int64_t b = 0, e = 0;
    if (live) {
        mine = keys[t];
        const int r = pa.rx[mine.index()];
        b = off[r];
        e = off[r + 1];
    }
    const int64_t last = (pa.n - 1 < (int64_t)blockIdx.x * RANK_TILE + RANK_TILE - 1) ? pa.n - 1 : (int64_t)blockIdx.x * RANK_TILE + RANK_TILE - 1;
    if (threadIdx.x == 0) s_span[0] = b;
    if (t == last) s_span[1] = e;
    __syncthreads();
    const int64_t lo = s_span[0], hi = s_span[1];
    int rank = 0;
    for (int64_t base = lo; base < hi; base += RANK_TILE) {
        if (base + threadIdx.x < hi) s_keys[threadIdx.x] = keys[base + threadIdx.x];
        __syncthreads();
        const int64_t from = b > base ? b : base, to = e < base + RANK_TILE ? e : base + RANK_TILE;
        int64_t q = from;
        for (; q + 4 <= to; q += 4) {  // four keys fetched together
            const PathKey k0 = s_keys[q - base], k1 = s_keys[q - base + 1], k2 = s_keys[q - base + 2], k3 = s_keys[q - base + 3];
            rank += (int)key_before(k0, mine) + (int)key_before(k1, mine) + (int)key_before(k2, mine) + (int)key_before(k3, mine);
        }
        for (; q < to; ++q) rank += (int)key_before(s_keys[q - base], mine);
        __syncthreads();
    }
    if (live) ordered[b + rank] = mine.index();
}
// Sum of the powers of a receiver's paths in row order (the order fixes the rounding), and the receiver's dB per
// frequency (electromagnetism_utils.py:180-188): a wave per receiver, lanes = (path of a stretch, frequency), a stretch of
// 64 / n_freq paths fetched at once, the additions one after the other on values handed from lane to lane.
__global__ __launch_bounds__(256) void k_rx_power_db(const int64_t *__restrict__ off, const int32_t *__restrict__ ordered,
                                                      int64_t n_rx, int n_freq, const double *__restrict__ power, double p_in,
                                                      double *__restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rx) return;  // (whole waves leave)
    const int64_t b = off[r], e = off[r + 1];
    const int per_stretch = 64 / n_freq, slot = lane / n_freq, f = lane - slot * n_freq;
    double sum = 0.0;
    for (int64_t base = b; base < e; base += per_stretch) {
        const int cnt = (int)(e - base < per_stretch ? e - base : per_stretch);
        const double v = slot < cnt ? power[(size_t)ordered[base + slot] * n_freq + f] : 0.0;
        for (int q = 0; q < cnt; ++q) sum += __shfl(v, q * n_freq + f);
    }
    if (lane < n_freq) db[r * n_freq + lane] = sum != 0 ? 10 * log10(sum / p_in) : -INFINITY;
}

// Start values of the field step's accumulators: power sums, per-receiver counts and class counts zero, the shortest
// lengths and their indices all ones, every status "not computed" (bytes 0x7f).
__global__ void k_fields_start(double *__restrict__ power_sum, int64_t n_sum, unsigned long long *__restrict__ best_len,
                               unsigned long long *__restrict__ best_idx, int64_t n_rx, int32_t *__restrict__ rx_counts,
                               unsigned long long *__restrict__ cls_counts, int32_t *__restrict__ status, int64_t n_paths) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = first; i < n_sum; i += stride) power_sum[i] = 0.0;
    for (int64_t i = first; i < n_rx; i += stride) {
        best_len[i] = ~0ull;
        best_idx[i] = ~0ull;
    }
    for (int64_t i = first; i < 2 * n_rx; i += stride) rx_counts[i] = 0;
    for (int64_t i = first; i < n_paths; i += stride) status[i] = 0x7f7f7f7f;
    if (first < 3) cls_counts[first] = 0;
}

__global__ void k_power_db(const double *__restrict__ sum, int64_t n, double p_in, double *__restrict__ db) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double P = sum[i];
    db[i] = P != 0 ? 10 * log10(P / p_in) : -INFINITY;  // electromagnetism_utils.py:180-188
}

// ITU-R P.2040-2 table (materials_properties.py:47-109); index = material id used by the scene
static void fill_materials(FreqTable &ft, int f, double freq) {
    const double c0 = 299792458.0, eps0 = 8.8541878188e-12;
    const double g = freq / 1e9;
    const double eps[N_MATERIALS] = {1, 5.24, 3.91, 1.99, 6.31, 1, 15 * std::pow(g, -0.1)};
    const double sigma[N_MATERIALS] = {0, 0.0462 * std::pow(g, 0.7822), 0.0238 * std::pow(g, 0.16),
                                       0.0047 * std::pow(g, 1.0718), 0.0036 * std::pow(g, 1.3394), 1e7,
                                       0.035 * std::pow(g, 1.63)};
    ft.freq[f] = freq;
    ft.lambda[f] = c0 / freq;
    ft.K[f] = 2 * M_PI / ft.lambda[f];
    for (int m = 0; m < N_MATERIALS; ++m) {
        ft.eps_re[f][m] = eps[m];
        ft.eps_im[f][m] = -sigma[m] / (eps0 * 2 * M_PI * freq);
    }
}

// Per-path description arrays from the (group, candidate) pairs rt_solve_paths returns.
__global__ void k_describe(const int32_t *__restrict__ group, const int32_t *__restrict__ cand, int64_t n, int n_groups,
                           const int *__restrict__ g_refl, const int *__restrict__ g_edge,
                           const ChainSource *__restrict__ g_chains, const int32_t *const *__restrict__ g_edge_ids,
                           const int32_t *__restrict__ edge_pairs, const double *__restrict__ edges,
                           int32_t *__restrict__ kind, int32_t *__restrict__ chain, int32_t *__restrict__ edge_polys,
                           double *__restrict__ edge_pts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = rt_group_of(group[i]);
    const int64_t c = rt_cand_of(group[i], cand[i]);
    for (int k = 0; k < RT_MAX_ORDER; ++k) chain[i * RT_MAX_ORDER + k] = -1;
    edge_polys[2 * i] = edge_polys[2 * i + 1] = -1;
    for (int k = 0; k < 6; ++k) edge_pts[6 * i + k] = 0.0;
    if (g < 0 || g >= n_groups) {
        kind[i] = 0;
        return;
    }
    const int nr = g_refl[g];
    kind[i] = nr + (g_edge[g] ? 256 : 0);
    if (nr > 0) chain_load(g_chains[g], nr, c, chain + i * RT_MAX_ORDER);
    if (g_edge[g]) {
        const int e = g_edge_ids[g][c];
        edge_polys[2 * i] = edge_pairs[2 * e];
        edge_polys[2 * i + 1] = edge_pairs[2 * e + 1];
        for (int k = 0; k < 6; ++k) edge_pts[6 * i + k] = edges[6 * (size_t)e + k];
    }
}

extern "C" int rt_describe_paths(rt_candidates *cand, const int32_t *path_group, const int32_t *path_cand,
                                 int64_t n_paths, int32_t *out_kind, int32_t *out_chain, int32_t *out_edge_polys,
                                 double *out_edge_points, int on_device) {
    if (!cand) RT_FAIL("rt_describe_paths: null handle");
    if (n_paths < 0) RT_FAIL("rt_describe_paths: negative size");
    if (n_paths == 0) return 0;
    if (!path_group || !path_cand || !out_kind || !out_chain || !out_edge_polys || !out_edge_points)
        RT_FAIL("rt_describe_paths: null array");
    rt_scene *scene = cand->scene;
    RT_HIP(rt_use_device(scene->device));
    hipStream_t st = scene->stream;
    const int ng = (int)cand->groups.size();
    std::vector<int> h_refl(ng ? ng : 1), h_edge(ng ? ng : 1);
    std::vector<ChainSource> h_chains(ng ? ng : 1);
    std::vector<const int32_t *> h_eids(ng ? ng : 1);
    for (int g = 0; g < ng; ++g) {
        h_refl[g] = cand->groups[g].n_refl;
        h_edge[g] = cand->groups[g].has_edge;
        h_chains[g] = cand->groups[g].source;
        h_eids[g] = cand->groups[g].edge_ids;
    }
    if (!on_device) {
        for (int64_t i = 0; i < n_paths; ++i) {
            const int g = rt_group_of(path_group[i]);
            if (g >= ng) RT_FAIL("rt_describe_paths: group out of range");
            if (g >= 0 && (path_cand[i] < 0 || rt_cand_of(path_group[i], path_cand[i]) >= cand->groups[g].count))
                RT_FAIL("rt_describe_paths: candidate index out of range");
        }
    }
    auto dalloc = [&](size_t bytes) -> void * { return rt_scene_temp(scene, bytes); };
    const bool deferred = on_device && scene->defer_sync;
    int rc = -1;
    do {
        // the per-group tables: built and copied once per set of candidates (a set never changes after rt_enumerate)
        const size_t n_tab = h_refl.size();
        const size_t off_edge = sizeof(int) * n_tab, off_chains = (2 * sizeof(int) * n_tab + 15) & ~(size_t)15;
        const size_t off_eids = off_chains + sizeof(ChainSource) * n_tab, tab_bytes = off_eids + sizeof(void *) * n_tab;
        if (!cand->d_describe_tables) {
            std::vector<unsigned char> h_tab(tab_bytes, 0);
            std::memcpy(h_tab.data(), h_refl.data(), sizeof(int) * n_tab);
            std::memcpy(h_tab.data() + off_edge, h_edge.data(), sizeof(int) * n_tab);
            std::memcpy(h_tab.data() + off_chains, h_chains.data(), sizeof(ChainSource) * n_tab);
            std::memcpy(h_tab.data() + off_eids, h_eids.data(), sizeof(void *) * n_tab);
            void *d = nullptr;
            if (hipMalloc(&d, tab_bytes) != hipSuccess) break;
            cand->allocations.push_back(d);
            if (hipMemcpy(d, h_tab.data(), tab_bytes, hipMemcpyHostToDevice) != hipSuccess) break;
            cand->d_describe_tables = d;
        }
        unsigned char *tab = (unsigned char *)cand->d_describe_tables;
        int *d_refl = (int *)tab, *d_edge = (int *)(tab + off_edge);
        ChainSource *d_chains = (ChainSource *)(tab + off_chains);
        const int32_t **d_eids = (const int32_t **)(tab + off_eids);
        const int32_t *d_group = path_group, *d_cand = path_cand;
        int32_t *d_kind = out_kind, *d_chain = out_chain, *d_ep = out_edge_polys;
        double *d_pts = out_edge_points;
        const size_t n = (size_t)n_paths;
        if (!on_device) {
            int32_t *a = (int32_t *)dalloc(4 * n), *b = (int32_t *)dalloc(4 * n);
            d_kind = (int32_t *)dalloc(4 * n);
            d_chain = (int32_t *)dalloc(4 * n * RT_MAX_ORDER);
            d_ep = (int32_t *)dalloc(8 * n);
            d_pts = (double *)dalloc(48 * n);
            if (!a || !b || !d_kind || !d_chain || !d_ep || !d_pts) break;
            if (hipMemcpyAsync(a, path_group, 4 * n, hipMemcpyHostToDevice, st) != hipSuccess) break;
            if (hipMemcpyAsync(b, path_cand, 4 * n, hipMemcpyHostToDevice, st) != hipSuccess) break;
            d_group = a;
            d_cand = b;
        }
        hipLaunchKernelGGL(k_describe, dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, st, d_group, d_cand, n_paths,
                           ng, d_refl, d_edge, d_chains, d_eids, cand->d_edge_pairs, cand->d_edges, d_kind, d_chain, d_ep,
                           d_pts);
        if (hipGetLastError() != hipSuccess) break;
        if (!on_device) {
            if (hipMemcpyAsync(out_kind, d_kind, 4 * n, hipMemcpyDeviceToHost, st) != hipSuccess) break;
            if (hipMemcpyAsync(out_chain, d_chain, 4 * n * RT_MAX_ORDER, hipMemcpyDeviceToHost, st) != hipSuccess) break;
            if (hipMemcpyAsync(out_edge_polys, d_ep, 8 * n, hipMemcpyDeviceToHost, st) != hipSuccess) break;
            if (hipMemcpyAsync(out_edge_points, d_pts, 48 * n, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        }
        if (!deferred && hipStreamSynchronize(st) != hipSuccess) break;
        rc = 0;
    } while (0);
    if (rc) {
        rt_set_error(std::string("rt_describe_paths: ") + hipGetErrorString(hipGetLastError()));
        (void)hipStreamSynchronize(st);
    }
    rt_scene_temp_release_all(scene);
    return rc;
}

extern "C" int rt_fields(rt_scene *scene, const double *tx, const double *rx, int64_t n_rx, const int32_t *path_rx,
                         const int32_t *path_kind, const int32_t *path_chain, const int32_t *path_edge_polys,
                         const double *path_edge_points, const double *path_points, int64_t n_paths,
                         const double *freq_hz, int32_t n_freq, const rt_antenna_params *antenna,
                         const rt_field_options *options, double *out_field, double *out_power, double *out_angles,
                         double *out_time, int32_t *out_status, double *out_power_db, int on_device) {
    if (!scene || !tx || !freq_hz || !antenna) RT_FAIL("rt_fields: null argument");
    if (options && ((options->aim_tx == nullptr) != (options->aim_rx == nullptr)))
        RT_FAIL("rt_fields: aim_tx and aim_rx go together");
    if (n_freq < 1 || n_freq > MAX_FREQ) RT_FAIL("rt_fields: between 1 and 16 frequencies per call");
    if (n_rx < 0 || n_paths < 0) RT_FAIL("rt_fields: negative size");
    if (n_rx > 0 && (!rx || !out_power_db)) RT_FAIL("rt_fields: null receiver arrays");
    if (n_paths > 0 && (!path_rx || !path_kind || !path_chain || !path_edge_polys || !path_edge_points || !path_points ||
                        !out_field || !out_power || !out_angles || !out_time || !out_status))
        RT_FAIL("rt_fields: null path arrays");
    for (int f = 0; f < n_freq; ++f)
        if (!(freq_hz[f] > 0)) RT_FAIL("rt_fields: frequencies must be positive");
    if (n_rx == 0) return 0;
    if (!on_device) {
        // operand shapes are checked before a kernel dereferences them
        const int n_poly = scene->view.n;
        for (int64_t i = 0; i < n_paths; ++i) {
            const int nr = path_kind[i] & 255, he = path_kind[i] >> 8;
            if (path_rx[i] < 0 || path_rx[i] >= n_rx) RT_FAIL("rt_fields: receiver index out of range");
            if (path_kind[i] < 0 || he > 1 || nr + he > RT_MAX_ORDER) RT_FAIL("rt_fields: bad path kind");
            for (int k = 0; k < nr; ++k)
                if (path_chain[i * RT_MAX_ORDER + k] < 0 || path_chain[i * RT_MAX_ORDER + k] >= n_poly)
                    RT_FAIL("rt_fields: polygon index out of range");
            if (he && (path_edge_polys[2 * i] < 0 || path_edge_polys[2 * i] >= n_poly || path_edge_polys[2 * i + 1] < 0 ||
                       path_edge_polys[2 * i + 1] >= n_poly))
                RT_FAIL("rt_fields: edge polygon index out of range");
        }
    }
    RT_HIP(rt_use_device(scene->device));
    hipStream_t st = scene->stream;
    FreqTable ft{};
    ft.n_freq = n_freq;
    for (int f = 0; f < n_freq; ++f) fill_materials(ft, f, freq_hz[f]);
    const double rough[N_MATERIALS] = {NAN, 1e-4, 1e-3, 1e-4, 1e-6, 1e-6, 1e-2};
    for (int m = 0; m < N_MATERIALS; ++m) ft.roughness[m] = rough[m];

    auto dalloc = [&](size_t bytes) -> void * { return rt_scene_temp(scene, bytes); };
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        rt_scene_temp_release_all(scene);
    };
#define F_HIP(call)                                                                                 \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            rt_set_error(std::string("rt_fields: ") + #call + ": " + hipGetErrorString(e_));        \
            cleanup();                                                                              \
            return -1;                                                                              \
        }                                                                                           \
    } while (0)
#define F_PTR(p)                                                \
    if (!(p)) {                                                 \
        rt_set_error("rt_fields: out of device memory");        \
        cleanup();                                              \
        return -1;                                              \
    }

    PathArrays pa{};
    pa.n = n_paths;
    const double *d_rx = rx;
    FieldOut fo{};
    double *d_db = out_power_db;
    const size_t np_ = (size_t)n_paths, nf = (size_t)n_freq;
    if (on_device) {
        pa.rx = path_rx; pa.kind = path_kind; pa.chain = path_chain; pa.edge_polys = path_edge_polys;
        pa.edge_pts = path_edge_points; pa.pts = path_points;
        fo.field = out_field; fo.power = out_power; fo.angles = out_angles; fo.time = out_time; fo.status = out_status;
    } else {
        int32_t *a = (int32_t *)dalloc(4 * np_), *b = (int32_t *)dalloc(4 * np_), *c = (int32_t *)dalloc(4 * np_ * RT_MAX_ORDER),
                *d = (int32_t *)dalloc(8 * np_);
        double *e = (double *)dalloc(48 * np_), *p = (double *)dalloc(sizeof(double) * RT_MAX_ORDER * 3 * np_);
        double *r = (double *)dalloc(sizeof(double) * 3 * n_rx);
        F_PTR(a); F_PTR(b); F_PTR(c); F_PTR(d); F_PTR(e); F_PTR(p); F_PTR(r);
        if (n_paths) {
            F_HIP(hipMemcpyAsync(a, path_rx, 4 * np_, hipMemcpyHostToDevice, st));
            F_HIP(hipMemcpyAsync(b, path_kind, 4 * np_, hipMemcpyHostToDevice, st));
            F_HIP(hipMemcpyAsync(c, path_chain, 4 * np_ * RT_MAX_ORDER, hipMemcpyHostToDevice, st));
            F_HIP(hipMemcpyAsync(d, path_edge_polys, 8 * np_, hipMemcpyHostToDevice, st));
            F_HIP(hipMemcpyAsync(e, path_edge_points, 48 * np_, hipMemcpyHostToDevice, st));
            F_HIP(hipMemcpyAsync(p, path_points, sizeof(double) * RT_MAX_ORDER * 3 * np_, hipMemcpyHostToDevice, st));
        }
        F_HIP(hipMemcpyAsync(r, rx, sizeof(double) * 3 * n_rx, hipMemcpyHostToDevice, st));
        pa.rx = a; pa.kind = b; pa.chain = c; pa.edge_polys = d; pa.edge_pts = e; pa.pts = p;
        d_rx = r;
        fo.field = (double *)dalloc(sizeof(double) * 6 * np_ * nf);
        fo.power = (double *)dalloc(sizeof(double) * np_ * nf);
        fo.angles = (double *)dalloc(sizeof(double) * 4 * np_);
        fo.time = (double *)dalloc(sizeof(double) * np_);
        fo.status = (int32_t *)dalloc(sizeof(int32_t) * np_);
        d_db = (double *)dalloc(sizeof(double) * n_rx * nf);
        F_PTR(fo.field); F_PTR(fo.power); F_PTR(fo.angles); F_PTR(fo.time); F_PTR(fo.status); F_PTR(d_db);
    }
    if (options) {
        const double *opt_src[4] = {options->path_origin, options->path_e_in, options->aim_tx, options->aim_rx};
        const size_t opt_bytes[4] = {24 * np_, 48 * np_, 24 * (size_t)n_rx, 24 * (size_t)n_rx};
        const double *opt_dev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 4; ++k) {
            if (!opt_src[k]) continue;
            if (on_device) {
                opt_dev[k] = opt_src[k];
            } else {
                double *d = (double *)dalloc(opt_bytes[k]);
                F_PTR(d);
                if (opt_bytes[k]) F_HIP(hipMemcpyAsync(d, opt_src[k], opt_bytes[k], hipMemcpyHostToDevice, st));
                opt_dev[k] = d;
            }
        }
        pa.origin = opt_dev[0];
        pa.e_in = opt_dev[1];
        pa.aim_tx = opt_dev[2];
        pa.aim_rx = opt_dev[3];
        if (options->n_late_polygons < 0 || options->n_late_polygons > scene->view.n) {
            rt_set_error("rt_fields: n_late_polygons outside the scene");
            cleanup();
            return -1;
        }
        pa.n_late = options->n_late_polygons;
    }
    // device time of the call and of the field kernel alone (rt_fields_last_timing), while the scene's timing is on
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (scene->timing) {
        while (scene->event_pool.size() < 4) {
            hipEvent_t e = nullptr;
            F_HIP(hipEventCreate(&e));
            scene->event_pool.push_back(e);
        }
        for (int k = 0; k < 4; ++k) ev[k] = scene->event_pool[k];
        F_HIP(hipEventRecord(ev[0], st));
    }
    scene->last_fields_ms[0] = scene->last_fields_ms[1] = 0;
    fo.power_sum = (double *)dalloc(sizeof(double) * n_rx * nf);
    unsigned long long *best_len = (unsigned long long *)dalloc(sizeof(unsigned long long) * n_rx);
    unsigned long long *best_idx = (unsigned long long *)dalloc(sizeof(unsigned long long) * n_rx);
    // per-receiver path lists (counts and cursors, offsets, the paths in receiver order), class lists, tree hits
    int32_t *rx_counts = (int32_t *)dalloc(sizeof(int32_t) * 2 * n_rx);
    int64_t *rx_off = (int64_t *)dalloc(sizeof(int64_t) * (n_rx + 1));
    int32_t *rx_order = (int32_t *)dalloc(sizeof(int32_t) * np_);  // the paths in the reference's order (long lists only)
    PathKey *rx_keys = (PathKey *)dalloc(sizeof(PathKey) * np_);
    int32_t *cls_lists = (int32_t *)dalloc(sizeof(int32_t) * 3 * np_);
    TreeHits *los_trees = (TreeHits *)dalloc(sizeof(TreeHits) * np_);
    unsigned long long *cls_counts = scene->d_counters + 13;  // [13], [14], [15]
    F_PTR(fo.power_sum); F_PTR(best_len); F_PTR(best_idx); F_PTR(rx_counts); F_PTR(rx_off); F_PTR(rx_order);
    F_PTR(cls_lists); F_PTR(los_trees); F_PTR(rx_keys);
    // several frequencies and no segment through tree crowns: the paths' geometry goes through memory (k_fields)
    double *geom = nullptr;
    if (n_freq > 1 && pa.n_late == 0 && n_paths > 0) {
        geom = (double *)dalloc(sizeof(double) * GEOM_SLOTS * np_);
        F_PTR(geom);
    }
    // everything that has to start from a fixed value, in one launch (six fills used to go one by one)
    {
        const int64_t most = std::max<int64_t>((int64_t)(n_rx * nf), std::max<int64_t>(2 * n_rx, (int64_t)np_));
        hipLaunchKernelGGL(k_fields_start, dim3((unsigned)std::min<int64_t>(2048, (most + 255) / 256)), dim3(256), 0, st,
                           fo.power_sum, (int64_t)(n_rx * nf), best_len, best_idx, (int64_t)n_rx, rx_counts, cls_counts,
                           fo.status, (int64_t)np_);
    }
    const d3 TX = ld3(tx);
    if (n_paths > 0) {
        // six or seven launches: start values; classes, shortest lengths and counts; shortest paths, offsets and the
        // lines of sight's trees; (a sweep: geometry;) fields and keys; places and statuses; sums and dB
        hipLaunchKernelGGL(k_fields_pass1, dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, st, pa, TX, d_rx, best_len,
                           cls_lists, cls_counts, rx_counts);
        if (ev[1]) F_HIP(hipEventRecord(ev[1], st));
        const int path_blocks = (int)((n_paths + 1023) / 1024);
        const int los_blocks = pa.e_in ? 0 : (int)std::min<int64_t>(n_rx, 2048);  // (a receiver has one line of sight at most)
        hipLaunchKernelGGL(k_fields_pass2, dim3((unsigned)(path_blocks + 1 + los_blocks)), dim3(1024), 0, st, scene->view, pa, TX,
                           d_rx, best_len, best_idx, rx_counts, (int64_t)n_rx, rx_off, cls_lists, cls_counts, los_trees,
                           path_blocks);
        // the class kernels read their list's length on the device: the grids are sized for the whole batch and the
        // blocks past the end of a list leave at once
        const unsigned fblocks = (unsigned)((n_paths + 127) / 128);
        if (geom) {
            // a sweep: the geometry of every path once, then a thread per (path, frequency) for what the frequency decides
            hipLaunchKernelGGL(k_fields<FIELDS_GEOMETRY>, dim3(fblocks, FIELDS_GEOMETRY_PARTS, 3), dim3(128), 0, st, scene->view,
                               pa, ft, *antenna, TX, d_rx, best_idx, cls_lists, cls_counts, los_trees, fo, geom, (int64_t)np_,
                               rx_off, rx_counts + n_rx, rx_keys);
            const int64_t per_block = 128 / n_freq;
            hipLaunchKernelGGL(k_fields<FIELDS_FREQUENCY>, dim3((unsigned)((n_paths + per_block - 1) / per_block), 1, 4), dim3(128),
                               0, st, scene->view, pa, ft, *antenna, TX, d_rx, best_idx, cls_lists, cls_counts, los_trees, fo, geom,
                               (int64_t)np_, rx_off, rx_counts + n_rx, rx_keys);
        } else {
            hipLaunchKernelGGL(k_fields<FIELDS_WHOLE>, dim3(fblocks, (unsigned)n_freq, 4), dim3(128), 0, st, scene->view, pa, ft,
                               *antenna, TX, d_rx, best_idx, cls_lists, cls_counts, los_trees, fo, (double *)nullptr, (int64_t)0,
                               rx_off, rx_counts + n_rx, rx_keys);
        }
        if (ev[2]) F_HIP(hipEventRecord(ev[2], st));
        const int rank_blocks = (int)((n_paths + RANK_TILE - 1) / RANK_TILE);
        hipLaunchKernelGGL(k_rx_path_rank, dim3((unsigned)(2 * rank_blocks)), dim3(RANK_TILE), 0, st, pa, rx_off, rx_keys, rx_order,
                           fo.status, rank_blocks);
        hipLaunchKernelGGL(k_rx_power_db, dim3((unsigned)((n_rx + 3) / 4)), dim3(256), 0, st, rx_off, rx_order, (int64_t)n_rx,
                           n_freq, fo.power, antenna->p_in, d_db);
    } else {
        hipLaunchKernelGGL(k_power_db, dim3((unsigned)((n_rx * nf + 255) / 256)), dim3(256), 0, st, fo.power_sum,
                           (int64_t)(n_rx * nf), antenna->p_in, d_db);
    }
    F_HIP(hipGetLastError());
    if (!on_device) {
        if (n_paths) {
            F_HIP(hipMemcpyAsync(out_field, fo.field, sizeof(double) * 6 * np_ * nf, hipMemcpyDeviceToHost, st));
            F_HIP(hipMemcpyAsync(out_power, fo.power, sizeof(double) * np_ * nf, hipMemcpyDeviceToHost, st));
            F_HIP(hipMemcpyAsync(out_angles, fo.angles, sizeof(double) * 4 * np_, hipMemcpyDeviceToHost, st));
            F_HIP(hipMemcpyAsync(out_time, fo.time, sizeof(double) * np_, hipMemcpyDeviceToHost, st));
            F_HIP(hipMemcpyAsync(out_status, fo.status, sizeof(int32_t) * np_, hipMemcpyDeviceToHost, st));
        }
        F_HIP(hipMemcpyAsync(out_power_db, d_db, sizeof(double) * n_rx * nf, hipMemcpyDeviceToHost, st));
    }
    if (ev[3]) F_HIP(hipEventRecord(ev[3], st));
    if (on_device && scene->defer_sync) {
        // queued only: rt_scene_synchronize waits and reads the events.  The temporary blocks may be handed out again
        // at once: whoever gets them queues on the same stream, behind these kernels.
        scene->fields_events_pending = ev[0] != nullptr;
        scene->fields_events_with_kernels = n_paths > 0;
        rt_scene_temp_release_all(scene);
        return 0;
    }
    F_HIP(hipStreamSynchronize(st));
    if (ev[0]) {
        float t = 0;
        if (n_paths > 0 && hipEventElapsedTime(&t, ev[1], ev[2]) == hipSuccess) scene->last_fields_ms[0] = t;
        if (hipEventElapsedTime(&t, ev[0], ev[3]) == hipSuccess) scene->last_fields_ms[1] = t;
    }
    rt_scene_temp_release_all(scene);
    return 0;
}

extern "C" int rt_fields_last_timing(rt_scene *scene, double *ms2) {
    if (!scene || !ms2) RT_FAIL("rt_fields_last_timing: null argument");
    ms2[0] = scene->last_fields_ms[0];
    ms2[1] = scene->last_fields_ms[1];
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One fixed-size record per receiver: what the ranks of a multi-GPU run exchange instead of their paths (the
// device-side stand-in for merge_solved_problems of a coverage map, multithread_solve.py:45-60 of the reference).
// Record: n_freq doubles (power in dB per frequency), then RT_RECORD_INTS int32: paths per path type (line of sight;
// 1..RT_MAX_ORDER reflections; 0..RT_MAX_ORDER-1 reflections and a diffraction), paths in all, paths whose field status
// is not zero, largest status code; padded to a multiple of 8 bytes.
// ---------------------------------------------------------------------------------------------------------------
#define RT_RECORD_TYPES (1 + 2 * RT_MAX_ORDER)
#define RT_RECORD_INTS (RT_RECORD_TYPES + 3)

__global__ void k_records_start(int64_t n_rx, int n_freq, const double *__restrict__ power_db, unsigned char *__restrict__ rec,
                                int record_bytes) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rx) return;
    double *db = (double *)(rec + r * record_bytes);
    for (int f = 0; f < n_freq; ++f) db[f] = power_db[r * n_freq + f];
    int32_t *ints = (int32_t *)(rec + r * record_bytes + 8 * n_freq);
    for (int k = 0; k < (record_bytes - 8 * n_freq) / 4; ++k) ints[k] = 0;
}

__global__ void k_records_count(int64_t n_paths, int64_t n_rx, const int32_t *__restrict__ path_rx,
                                const int32_t *__restrict__ path_kind, const int32_t *__restrict__ path_status, int n_freq,
                                unsigned char *__restrict__ rec, int record_bytes) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_paths) return;
    const int64_t r = path_rx[i];
    if (r < 0 || r >= n_rx) return;
    int32_t *ints = (int32_t *)(rec + r * record_bytes + 8 * n_freq);
    const int kind = path_kind[i];
    const int type = (kind & 255) + (RT_MAX_ORDER + 1) * (kind >> 8);  // LOS, R.., then D, RD, ..
    if (type >= 0 && type < RT_RECORD_TYPES) atomicAdd(&ints[type], 1);
    atomicAdd(&ints[RT_RECORD_TYPES], 1);
    const int status = path_status ? path_status[i] : 0;
    if (status != 0) {
        atomicAdd(&ints[RT_RECORD_TYPES + 1], 1);
        atomicMax(&ints[RT_RECORD_TYPES + 2], status);
    }
}

extern "C" int32_t rt_receiver_record_bytes(int32_t n_freq) {
    const int raw = 8 * n_freq + 4 * RT_RECORD_INTS;
    return (raw + 7) & ~7;
}

extern "C" int rt_receiver_records(rt_scene *scene, int64_t n_rx, int32_t n_freq, const double *power_db, int64_t n_paths,
                                   const int32_t *path_rx, const int32_t *path_kind, const int32_t *path_status,
                                   unsigned char *out) {
    if (!scene) RT_FAIL("rt_receiver_records: null scene");
    if (n_rx < 0 || n_paths < 0 || n_freq < 1) RT_FAIL("rt_receiver_records: bad size");
    if (n_rx == 0) return 0;
    if (!power_db || !out || (n_paths > 0 && (!path_rx || !path_kind))) RT_FAIL("rt_receiver_records: null array");
    RT_HIP(rt_use_device(scene->device));
    hipStream_t st = scene->stream;
    const int bytes = rt_receiver_record_bytes(n_freq);
    hipLaunchKernelGGL(k_records_start, dim3((unsigned)((n_rx + 255) / 256)), dim3(256), 0, st, n_rx, n_freq, power_db, out, bytes);
    if (n_paths > 0)
        hipLaunchKernelGGL(k_records_count, dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, st, n_paths, n_rx, path_rx,
                           path_kind, path_status, n_freq, out, bytes);
    RT_HIP(hipGetLastError());
    if (!scene->defer_sync) RT_HIP(hipStreamSynchronize(st));
    return 0;
}
