// Host-side text encoding of the array columns of the reference's result table (its CSV holds a receiver position, two
// angle pairs and a complex field per row as strings: numpy.array2string(a, separator=",") with the spaces removed,
// electromagnetism_fun/electromagnetism_utils.py:104-117 of the reference).  A coverage map has a few hundred thousand
// rows; formatting them value by value through NumPy's array printer took seconds.  This does what the array printer
// does for a short 1-D array of finite numbers with NumPy's default print options:
//   - the array is printed in exponent form when its largest absolute value is >= 1e8, its smallest non-zero one
//     < 1e-4, or their ratio > 1e3; otherwise in positional form;
//   - positional: at most 8 fractional digits, correctly rounded, and no more than identify the number; trailing zeros
//     trimmed, the point kept ("1.", "0.25");
//   - exponent form: every element gets as many mantissa digits as the element that needs most (at most 8), the
//     exponent at least two digits ("1.50e-05");
//   - a complex array: real and imaginary parts are formatted as two arrays of their own, then joined "re+imj".
// No GPU code here.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rt_pathsolve.h"

namespace {

// one real array of n finite values -> n strings (without separators) appended to parts[]
void format_real_row(const double *v, int n, int stride, std::string *parts) {
    double lo = INFINITY, hi = 0.0;
    for (int i = 0; i < n; ++i) {
        const double a = std::fabs(v[i * stride]);
        if (a != 0.0) {
            lo = std::fmin(lo, a);
            hi = std::fmax(hi, a);
        }
    }
    const bool exp_format = hi > 0.0 && (hi >= 1e8 || lo < 1e-4 || hi / lo > 1e3);
    char buf[64];
    if (!exp_format) {
        for (int i = 0; i < n; ++i) {
            const double x = v[i * stride];
            int len = std::snprintf(buf, sizeof(buf), "%.8f", x);
            if (std::fabs(x) >= 1e7) {
                // 8 fractional digits would be 16 significant ones or more, past what identifies a double: the array
                // printer stops at the shortest string that reads back as the same number
                for (int p = 0; p < 8; ++p) {
                    char shorter[64];
                    const int l2 = std::snprintf(shorter, sizeof(shorter), "%.*f", p, x);
                    if (std::strtod(shorter, nullptr) == x) {
                        std::memcpy(buf, shorter, (size_t)l2 + 1);
                        len = l2;
                        if (p == 0) buf[len++] = '.';
                        break;
                    }
                }
            }
            while (len > 0 && buf[len - 1] == '0') --len;  // the point stays
            parts[i].assign(buf, (size_t)len);
        }
        return;
    }
    // exponent form: mantissa digits of the element that needs most, exponent digits likewise
    char rows[16][64];
    int mant_end[16], digits = 0, exp_digits = 2;
    for (int i = 0; i < n; ++i) {
        std::snprintf(rows[i], sizeof(rows[i]), "%.8e", v[i * stride]);
        char *e = std::strchr(rows[i], 'e');
        int end = (int)(e - rows[i]);  // one past the mantissa
        int last = end;
        while (last > 0 && rows[i][last - 1] == '0') --last;
        const char *point = std::strchr(rows[i], '.');
        const int d = last - (int)(point - rows[i]) - 1;
        digits = d > digits ? d : digits;
        mant_end[i] = end;
        const int ed = (int)std::strlen(e + 2);
        exp_digits = ed > exp_digits ? ed : exp_digits;
    }
    for (int i = 0; i < n; ++i) {
        const char *point = std::strchr(rows[i], '.');
        const int keep = (int)(point - rows[i]) + 1 + digits;  // sign, leading digit, point, digits
        std::string s(rows[i], (size_t)keep);
        const char *e = rows[i] + mant_end[i];
        s.push_back('e');
        s.push_back(e[1]);  // sign of the exponent
        const int ed = (int)std::strlen(e + 2);
        s.append((size_t)(exp_digits - ed), '0');
        s.append(e + 2);
        parts[i] = s;
    }
}

}  // namespace

// values[n_rows * n_cols] real, or [n_rows * n_cols * 2] complex (re, im interleaved).  Writes the rows' strings one
// after the other, each followed by '\n', into out[capacity]; a row that holds a non-finite value gets an empty string
// (the caller formats that one itself).  Returns the bytes written, or -1 when the buffer is too small (64 bytes per
// real value always suffice).
extern "C" int64_t rt_format_array_rows(const double *values, int64_t n_rows, int32_t n_cols, int32_t is_complex, char *out,
                                        int64_t capacity) {
    if (!values || !out || n_rows < 0 || n_cols < 1 || n_cols > 16) return -1;
    int64_t at = 0;
    std::string re[16], im[16];
    const int step = is_complex ? 2 : 1;
    for (int64_t r = 0; r < n_rows; ++r) {
        const double *row = values + (size_t)r * n_cols * step;
        bool finite = true;
        for (int i = 0; i < n_cols * step; ++i) finite = finite && std::isfinite(row[i]);
        std::string line;
        if (finite) {
            format_real_row(row, n_cols, step, re);
            if (is_complex) format_real_row(row + 1, n_cols, step, im);
            line.push_back('[');
            for (int i = 0; i < n_cols; ++i) {
                if (i) line.push_back(',');
                line += re[i];
                if (is_complex) {
                    if (im[i][0] != '-') line.push_back('+');
                    line += im[i];
                    line.push_back('j');
                }
            }
            line.push_back(']');
        }
        if (at + (int64_t)line.size() + 1 > capacity) return -1;
        std::memcpy(out + at, line.data(), line.size());
        at += (int64_t)line.size();
        out[at++] = '\n';
    }
    return at;
}

// ---------------------------------------------------------------------------------------------------------------
// JSON text of the solved problem's path containers (ray_tracing.py:66-132 of the reference writes them with
// json.dump: {receiver: {order: [[points, polygon chain(, edge key, edge points)], ...]}}).  A coverage map holds a
// few hundred thousand paths; building the nested Python lists and encoding them took seconds.  This writes the same
// characters json.dumps writes: ", " and ": " separators, integer keys as strings, floats as Python's repr (the
// shortest digits that read back as the same number; exponent form when the decimal point would lie more than 16
// places right or 4 places left of the first digit; ".0" after an integral value in positional form).
// ---------------------------------------------------------------------------------------------------------------
#include <charconv>

namespace {

// Python's float repr of a finite x, appended to out
inline void append_py_float(std::string &out, double x) {
    char sci[40];
    auto r = std::to_chars(sci, sci + sizeof(sci) - 1, x, std::chars_format::scientific);  // [-]d[.ddd]e[+-]dd
    *r.ptr = 0;
    const char *p = sci, *end = r.ptr;
    if (*p == '-') {
        out.push_back('-');
        ++p;
    }
    const char *e = p;
    while (e < end && *e != 'e') ++e;
    char digits[24];
    int nd = 0;
    for (const char *q = p; q < e; ++q)
        if (*q != '.') digits[nd++] = *q;
    const int exp10 = std::atoi(e + 1);
    const int decpt = exp10 + 1;  // value = 0.d1d2... * 10^decpt
    if (x == 0.0) {
        out.append("0.0");
        return;
    }
    if (decpt <= -4 || decpt > 16) {
        out.push_back(digits[0]);
        if (nd > 1) {
            out.push_back('.');
            out.append(digits + 1, (size_t)(nd - 1));
        }
        out.push_back('e');
        const int ex = decpt - 1;
        out.push_back(ex < 0 ? '-' : '+');
        const int ax = ex < 0 ? -ex : ex;
        if (ax < 10) out.push_back('0');
        out.append(std::to_string(ax));
        return;
    }
    if (decpt <= 0) {
        out.append("0.");
        out.append((size_t)(-decpt), '0');
        out.append(digits, (size_t)nd);
    } else if (decpt >= nd) {
        out.append(digits, (size_t)nd);
        out.append((size_t)(decpt - nd), '0');
        out.append(".0");
    } else {
        out.append(digits, (size_t)decpt);
        out.push_back('.');
        out.append(digits + decpt, (size_t)(nd - decpt));
    }
}

inline void append_point(std::string &out, const double *p) {
    out.push_back('[');
    append_py_float(out, p[0]);
    out.append(", ");
    append_py_float(out, p[1]);
    out.append(", ");
    append_py_float(out, p[2]);
    out.push_back(']');
}

}  // namespace

extern "C" int64_t rt_format_paths_json(int64_t n_paths, const int64_t *path_rx, const int32_t *path_order,
                                        const int32_t *n_points, const double *points, int32_t point_stride,
                                        const int32_t *n_chain, const int32_t *chain, int32_t chain_stride,
                                        const int32_t *edge_key, const double *edge_points, int64_t n_keys,
                                        const int64_t *keys, char *out, int64_t capacity) {
    if (n_paths < 0 || n_keys < 0 || (n_paths > 0 && (!path_rx || !path_order || !n_points || !points || !n_chain)) ||
        (n_keys > 0 && !keys) || !out)
        return -1;
    std::string s;
    s.reserve((size_t)std::max<int64_t>(1024, n_paths * 320 + n_keys * 12));
    s.push_back('{');
    int64_t i = 0;
    for (int64_t k = 0; k < n_keys; ++k) {
        if (k) s.append(", ");
        s.push_back('"');
        s.append(std::to_string(keys[k]));
        s.append("\": {");
        // the paths of this receiver follow each other, grouped by order in the order they came in
        bool first_order = true;
        while (i < n_paths && path_rx[i] == keys[k]) {
            const int order = path_order[i];
            if (!first_order) s.append(", ");
            first_order = false;
            s.push_back('"');
            s.append(std::to_string(order));
            s.append("\": [");
            bool first_item = true;
            while (i < n_paths && path_rx[i] == keys[k] && path_order[i] == order) {
                if (!first_item) s.append(", ");
                first_item = false;
                s.append("[[");
                for (int q = 0; q < n_points[i]; ++q) {
                    if (q) s.append(", ");
                    append_point(s, points + (size_t)i * point_stride + 3 * q);
                }
                s.append("], [");
                for (int q = 0; q < n_chain[i]; ++q) {
                    if (q) s.append(", ");
                    s.append(std::to_string(chain[(size_t)i * chain_stride + q]));
                }
                s.push_back(']');
                if (edge_key && edge_points) {
                    s.append(", [");
                    s.append(std::to_string(edge_key[2 * i]));
                    s.append(", ");
                    s.append(std::to_string(edge_key[2 * i + 1]));
                    s.append("], [");
                    append_point(s, edge_points + 6 * (size_t)i);
                    s.append(", ");
                    append_point(s, edge_points + 6 * (size_t)i + 3);
                    s.push_back(']');
                }
                s.push_back(']');
                ++i;
            }
            s.push_back(']');
        }
        s.push_back('}');
    }
    s.push_back('}');
    if (i != n_paths) return -2;  // paths not grouped by receiver in the order of the keys
    if ((int64_t)s.size() > capacity) return -(int64_t)s.size() - 16;  // the room it needs, negated (less 16)
    std::memcpy(out, s.data(), s.size());
    return (int64_t)s.size();
}
