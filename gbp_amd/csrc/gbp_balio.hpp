// gbp_balio.hpp -- native reader of the reference's BAL-style text files (host only, no GPU work).
//
// File layout: joeaortiz/gbp data/README.md:5-14; what is accepted mirrors utils/read_balfile.py:4-37 -- blank lines
// and lines whose first token is '#' are skipped in front of the header; then `C L F`, `fx fy cx cy`, F observation
// rows `cam lmk u v` (first four tokens), then 6*C + 3*L scalars, one per line (first token).  Numbers go through
// std::from_chars / strtol, which round exactly like Python's float() / int(), so the arrays equal the reference reader's.
// At 1M observations the Python reader takes seconds; this one ~0.1 s (set-up cost).
#pragma once
#include <cerrno>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace gbp {

struct BalText {
    std::vector<char> buf;          // whole file + trailing '\0'
    const char *cur = nullptr;      // start of the next unread line
    long line_no = 0;

    int open(const char *path, std::string &err)
    {
        FILE *f = std::fopen(path, "rb");
        if (!f) { err = std::string("cannot open ") + path; return -1; }
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize((size_t)std::max<long>(n, 0) + 1);
        const size_t got = n > 0 ? std::fread(buf.data(), 1, (size_t)n, f) : 0;
        std::fclose(f);
        if ((long)got != n) { err = std::string("short read on ") + path; return -1; }
        buf[(size_t)n] = '\0';
        cur = buf.data();
        return 0;
    }
    // [b, e) of the next line; false at end of file
    bool next_line(const char *&b, const char *&e)
    {
        if (!cur || *cur == '\0') return false;
        b = cur;
        const char *q = cur;
        while (*q != '\0' && *q != '\n') ++q;
        e = q;
        cur = (*q == '\n') ? q + 1 : q;
        ++line_no;
        return true;
    }
};

inline const char *skip_ws(const char *p, const char *e) { while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v')) ++p; return p; }

// next whitespace-separated token of [p, e) parsed as double / integer; false when there is none or it is malformed
inline bool tok_double(const char *&p, const char *e, double &v)
{
    p = skip_ws(p, e);
    if (p >= e) return false;
    // std::from_chars (Eisel-Lemire, correctly rounded like strtod and Python's float()) is ~20x faster than glibc's
    // strtod on 17-digit input; strtod remains for what from_chars rejects (a leading '+', hex floats)
    const char *q = p;
    while (q < e && !(*q == ' ' || *q == '\t' || *q == '\r')) ++q;
    const auto r = std::from_chars(p, q, v);
    if (r.ec == std::errc() && r.ptr == q) { p = q; return true; }
    char *end = nullptr;
    v = std::strtod(p, &end);
    if (end == p || end > e) return false;
    p = end;
    return true;
}
inline bool tok_int(const char *&p, const char *e, long &v)
{
    p = skip_ws(p, e);
    if (p >= e) return false;
    char *end = nullptr;
    v = std::strtol(p, &end, 10);
    if (end == p || end > e) return false;
    if (end < e && !(*end == ' ' || *end == '\t' || *end == '\r')) return false;      // "12.5" is not an id
    p = end;
    return true;
}

// Header only (sizes): 0 or -1 with `err` set.
inline int bal_header(BalText &t, long &C, long &L, long &F, std::string &err)
{
    const char *b, *e;
    for (;;) {
        if (!t.next_line(b, e)) { err = "no header line found"; return -1; }
        const char *p = skip_ws(b, e);
        if (p < e && *p != '#') break;
    }
    const char *p = b;
    if (!tok_int(p, e, C) || !tok_int(p, e, L) || !tok_int(p, e, F) || C < 0 || L < 0 || F < 0) {
        err = "line " + std::to_string(t.line_no) + ": expected `n_cams n_lmks n_obs`";
        return -1;
    }
    return 0;
}

inline int bal_body(BalText &t, long C, long L, long F, double *K4, double *cam_means, double *lmk_means, double *meas,
                    int32_t *cam_idx, int32_t *lmk_idx, std::string &err)
{
    const char *b, *e;
    auto bad = [&](const char *what) { err = "line " + std::to_string(t.line_no) + ": " + what; return -1; };
    if (!t.next_line(b, e)) return bad("missing intrinsics line");
    {
        const char *p = b;
        for (int k = 0; k < 4; ++k) if (!tok_double(p, e, K4[k])) return bad("expected `fx fy cx cy`");
    }
    for (long i = 0; i < F; ++i) {
        if (!t.next_line(b, e)) return bad("file ends inside the observation rows");
        const char *p = b;
        long c, l;
        double u, v;
        if (!tok_int(p, e, c) || !tok_int(p, e, l) || !tok_double(p, e, u) || !tok_double(p, e, v)) return bad("expected `cam lmk u v`");
        if (c < 0 || c >= C || l < 0 || l >= L) return bad("camera / landmark id out of range");
        cam_idx[i] = (int32_t)c; lmk_idx[i] = (int32_t)l;
        meas[2 * i] = u; meas[2 * i + 1] = v;
    }
    for (long i = 0; i < 6 * C + 3 * L; ++i) {
        if (!t.next_line(b, e)) return bad("file ends inside the initialisation scalars");
        const char *p = b;
        double v;
        if (!tok_double(p, e, v)) return bad("expected one scalar");
        if (i < 6 * C) cam_means[i] = v; else lmk_means[i - 6 * C] = v;
    }
    return 0;
}

}  // namespace gbp
