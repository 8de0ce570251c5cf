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
