// Text artefacts (host only): the reference writes its consensus tables with DataFrame.to_csv(sep='\t') (cnmf.py:34-35),
// i.e. every float as Python's repr -- the shortest digit string that round-trips, fixed notation for exponents in
// [-4, 16), scientific with a two-digit exponent otherwise.  Formatting 450 000 usages in Python costs 0.25 s of a 0.6 s
// consensus() call at 50 000 cells; std::to_chars gives the same shortest digits, the notation rule is restated here.
#pragma once
#include <charconv>

namespace cnmf {

// repr(float) of a FINITE double into out (>= 32 bytes); returns the length
static inline int py_float_repr(double v, char* out)
{
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    *r.ptr = 0;                                               // [-]d[.ddd]e[+-]XX
    char* p = buf;
    int n = 0;
    if (*p == '-') { out[n++] = '-'; ++p; }
    char digits[32];
    int nd = 0;
    while (*p && *p != 'e') { if (*p != '.') digits[nd++] = *p; ++p; }
    const int ex = atoi(p + 1), decpt = ex + 1;
    if (decpt <= -4 || decpt > 16) {
        out[n++] = digits[0];
        if (nd > 1) { out[n++] = '.'; memcpy(out + n, digits + 1, nd - 1); n += nd - 1; }
        n += snprintf(out + n, 8, "e%c%02d", ex < 0 ? '-' : '+', abs(ex));
    } else if (decpt <= 0) {
        out[n++] = '0'; out[n++] = '.';
        for (int i = 0; i < -decpt; ++i) out[n++] = '0';
        memcpy(out + n, digits, nd); n += nd;
    } else if (decpt >= nd) {
        memcpy(out + n, digits, nd); n += nd;
        for (int i = nd; i < decpt; ++i) out[n++] = '0';
        out[n++] = '.'; out[n++] = '0';
    } else {
        memcpy(out + n, digits, decpt); n += decpt;
        out[n++] = '.';
        memcpy(out + n, digits + decpt, nd - decpt); n += nd - decpt;
    }
    return n;
}

}  // namespace cnmf

// rows x cols finite doubles (row-major) -> "[label<sep>]v<sep>v...<sep>v\n" per row, every v as Python's repr(float);
// row_labels (nullable): the rows' labels separated by '\n' (`labels_bytes` bytes in all).  Returns the bytes written, or
// -(bytes needed at most) when `cap` cannot hold the worst case (33 bytes per value + the labels + one separator per row).
extern "C" int64_t cnmf_format_rows_f64(const double* vals, int64_t rows, int64_t cols, char sep, const char* row_labels,
                                        int64_t labels_bytes, char* out, int64_t cap)
{
    if (!vals || !out || rows < 0 || cols < 1) return 0;
    const int64_t worst = rows * cols * 33 + (row_labels ? labels_bytes + rows + 1 : 0);
    if (cap < worst) return -worst;
    int64_t n = 0;
    const char* lp = row_labels;
    const char* lend = row_labels ? row_labels + labels_bytes : nullptr;
    for (int64_t i = 0; i < rows; ++i) {
        if (row_labels) {
            while (lp < lend && *lp != '\n') out[n++] = *lp++;
            if (lp < lend) ++lp;                                  // the '\n' between two labels
            out[n++] = sep;
        }
        for (int64_t j = 0; j < cols; ++j) {
            n += cnmf::py_float_repr(vals[i * cols + j], out + n);
            out[n++] = (j + 1 == cols) ? '\n' : sep;
        }
    }
    return n;
}
