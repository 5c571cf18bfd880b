// Host-side file-format helpers of the inference CLI (no GPU work, no HIP calls).
//
// The reference does both in per-element Python loops:
//   - FASTA -> one-hot   phyloformer/data.py:11-31   (here: bytes -> uint8 residue indices [N][L])
//   - distances -> PHYLIP text   infer_alns.py:14-25   (here: float [P] -> "%.10f" square matrix)
// At 60 x 500 those loops cost ~3 ms per alignment in CPython, more than the MI355X forward pass
// (2.3 ms), and hold the GIL.  These two functions do the same work in ~0.2 ms and are called through
// ctypes (GIL released), so the CLI's loader / writer threads scale with host cores.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/phyloformer_amd.h"

namespace {

const char kAlphabet[] = "ARNDCQEGHILKMFPSTWYVX-";   // phyloformer/data.py:7

struct Lut {
    uint8_t v[256];
    Lut() {
        memset(v, 255, sizeof v);
        for (int i = 0; kAlphabet[i]; ++i) v[(unsigned char)kAlphabet[i]] = (uint8_t)i;
    }
};
const Lut kLut;

// "%.10f" of a float, exactly as printf / CPython format it (round-half-even on the exact binary value),
// without printf's arbitrary-precision machinery: a float is m * 2^e with m < 2^24, so m * 10^10 < 2^58
// fits an integer and the rounding is one shift and one comparison.
inline int format_f10(float x, char* out) {
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int w = 0;
    const uint32_t ex = (bits >> 23) & 0xff;
    uint64_t m = bits & 0x7fffffu;
    if (ex == 0xff) {
        if (m) { memcpy(out, "nan", 3); return 3; }
        if (bits >> 31) out[w++] = '-';
        memcpy(out + w, "inf", 3);
        return w + 3;
    }
    if (bits >> 31) out[w++] = '-';
    int e;                                   // value = m * 2^e
    if (ex == 0) e = -149; else { m |= 0x800000u; e = (int)ex - 150; }
    unsigned __int128 q;                     // round(value * 10^10)
    if (e >= 0) {
        if (e > 60)                          // > 2^84: far outside any distance; let printf do it
            return w + snprintf(out + w, 64, "%.10f", (double)(x < 0 ? -x : x));
        q = ((unsigned __int128)m << e) * 10000000000ull;   // < 2^(24 + 60 + 34)
    } else {
        const unsigned __int128 n = (unsigned __int128)m * 10000000000ull;
        const int s = -e;
        if (s >= 100) q = 0;                 // n < 2^58 < half an ulp of the last printed digit
        else {
            q = n >> s;
            const unsigned __int128 rem = n - (q << s), half = (unsigned __int128)1 << (s - 1);
            if (rem > half || (rem == half && (q & 1))) ++q;
        }
    }
    const unsigned __int128 ten10 = 10000000000ull;
    unsigned __int128 ip = q / ten10;
    uint64_t fp = (uint64_t)(q % ten10);
    char tmp[48];
    int k = 0;
    do { tmp[k++] = (char)('0' + (int)(ip % 10)); ip /= 10; } while (ip);
    while (k) out[w++] = tmp[--k];
    out[w++] = '.';
    for (int i = 9; i >= 0; --i) { out[w + i] = (char)('0' + fp % 10); fp /= 10; }
    return w + 10;
}

// bytes.strip() of CPython: space, \t, \n, \r, \x0b, \x0c
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

// Would bytes.decode("utf8") accept [p, p + n)?  (CPython's strict decoder: no overlong forms, no surrogates,
// nothing above U+10FFFF.)  The reference decodes every header where it meets it (data.py:22).
inline bool valid_utf8(const unsigned char* p, int64_t n) {
    int64_t i = 0;
    while (i < n) {
        const unsigned char c = p[i];
        if (c < 0x80) { ++i; continue; }
        int extra;
        unsigned char lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) extra = 1;
        else if (c == 0xE0) { extra = 2; lo = 0xA0; }
        else if (c >= 0xE1 && c <= 0xEC) extra = 2;
        else if (c == 0xED) { extra = 2; hi = 0x9F; }
        else if (c >= 0xEE && c <= 0xEF) extra = 2;
        else if (c == 0xF0) { extra = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) extra = 3;
        else if (c == 0xF4) { extra = 3; hi = 0x8F; }
        else return false;
        if (i + extra >= n) return false;                  // truncated sequence
        if (p[i + 1] < lo || p[i + 1] > hi) return false;
        for (int k = 2; k <= extra; ++k)
            if (p[i + k] < 0x80 || p[i + k] > 0xBF) return false;
        i += extra + 1;
    }
    return true;
}

}  // namespace

extern "C" {

int pf_parse_fasta(const char* data, int64_t len, uint8_t* idx, int64_t idx_cap, int64_t* id_spans,
                   int32_t max_seqs, int32_t* n_out, int32_t* l_out, int64_t* detail) {
    if (!data || len < 0 || !n_out || !l_out) return PF_EINVAL;
    int64_t dummy = 0;
    if (!detail) detail = &dummy;
    *detail = 0;
    int32_t n = 0;
    int64_t cur_len = 0;      // residues of the current record
    int64_t first_len = -1;   // residues of record 0
    int64_t written = 0;
    bool ragged = false;
    int64_t pos = 0;
    while (pos <= len) {
        // one line: [pos, eol), split on '\n' only (binary file iteration)
        const char* nl = pos < len ? (const char*)memchr(data + pos, '\n', (size_t)(len - pos)) : nullptr;
        int64_t eol = nl ? (int64_t)(nl - data) : len;
        int64_t a = pos, b = eol;
        while (a < b && is_space((unsigned char)data[a])) ++a;
        while (b > a && is_space((unsigned char)data[b - 1])) --b;
        if (a < b) {
            if (data[a] == '>') {
                if (n > 0) {
                    if (first_len < 0) first_len = cur_len;
                    else if (cur_len != first_len) ragged = true;
                }
                if (!valid_utf8((const unsigned char*)data + a + 1, b - (a + 1))) {
                    *detail = a + 1;                     // offset of the id: the caller decodes it to raise
                    *n_out = n;
                    return PF_FASTA_EUTF8;
                }
                if (id_spans) {
                    if (n >= max_seqs) return PF_FASTA_ECAP;
                    id_spans[2 * n] = a + 1;
                    id_spans[2 * n + 1] = b - (a + 1);
                }
                ++n;
                cur_len = 0;
            } else {
                if (n == 0) return PF_FASTA_ENOHEADER;
                for (int64_t i = a; i < b; ++i) {
                    uint8_t r = kLut.v[(unsigned char)data[i]];
                    if (r == 255) { *detail = (unsigned char)data[i]; return PF_FASTA_EBYTE; }
                    if (idx) {
                        if (written >= idx_cap) return PF_FASTA_ECAP;
                        idx[written] = r;
                    }
                    ++written;
                }
                cur_len += b - a;
            }
        }
        if (!nl) break;
        pos = eol + 1;
    }
    if (n == 0) return PF_FASTA_EEMPTY;
    if (first_len < 0) first_len = cur_len;
    else if (cur_len != first_len) ragged = true;
    *n_out = n;
    if (ragged) { *l_out = -1; return PF_FASTA_ERAGGED; }
    *l_out = (int32_t)first_len;
    return PF_OK;
}

// ids[i] has id_lens[i] bytes (id_lens == NULL: NUL-terminated)
static int64_t format_phylip_impl(const float* preds, int32_t n, const char* const* ids, const int64_t* id_lens,
                                  char* out, int64_t cap) {
    if (!preds || n < 1 || !ids || (!out && cap > 0)) return PF_EINVAL;
    int64_t w = 0;
    char num[96];
    auto put = [&](const char* s, int64_t k) {
        if (out && w + k <= cap) memcpy(out + w, s, (size_t)k);
        w += k;
    };
    int k = snprintf(num, sizeof num, "%d\n", n);
    put(num, k);
    // pair (i, j), i < j, lexicographic: offset of row i is i*n - i*(i+1)/2 - (i+1)
    auto at = [&](int64_t i, int64_t j) -> float {
        if (i == j) return 0.0f;
        if (i > j) { int64_t t = i; i = j; j = t; }
        // dm + dm.T of the reference (infer_alns.py:19-20) adds the zero of the other triangle: -0.0 becomes 0.0
        return preds[i * n - i * (i + 1) / 2 + (j - i - 1)] + 0.0f;
    };
    for (int32_t i = 0; i < n; ++i) {
        const char* id = ids[i] ? ids[i] : "";
        put(id, id_lens ? id_lens[i] : (int64_t)strlen(id));
        put(" ", 1);
        for (int32_t j = 0; j < n; ++j) {
            if (j) put(" ", 1);
            k = format_f10(at(i, j), num);
            put(num, k);
        }
        put("\n", 1);
    }
    return w;
}

int64_t pf_format_phylip(const float* preds, int32_t n, const char* const* ids, char* out, int64_t cap) {
    return format_phylip_impl(preds, n, ids, nullptr, out, cap);
}
int64_t pf_format_phylip_n(const float* preds, int32_t n, const char* const* ids, const int64_t* id_lens, char* out,
                           int64_t cap) {
    if (!id_lens) return PF_EINVAL;
    for (int32_t i = 0; i < n; ++i) if (id_lens[i] < 0) return PF_EINVAL;
    return format_phylip_impl(preds, n, ids, id_lens, out, cap);
}

}  // extern "C"

// ---- neighbour joining + Newick text (the CLI's --trees: infer_alns.py:62-64,120-123) -------------------------------
//
// The reference calls skbio.tree.nj (not installed here); phyloformer_amd/nj.py is this build's pinned statement of
// the algorithm (FastME -m N goldens, tests/test_treecmp.py) and THIS function is its native twin: the same float64
// operations in the same order - numpy's pairwise row sums included - and Python's repr() of the branch lengths, so
// that the .nj.nwk files are byte-identical to nj.py's (tests/test_host.py) while `-t` stays on the native file
// pipeline (with `-t` every file used to fall back to per-file Python under the GIL).
namespace {

// numpy's add.reduce over a contiguous row (DOUBLE_pairwise_sum): < 8 sequential, <= 128 eight accumulators, else halves
double np_pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// Python's repr(float): the shortest digits that round-trip, fixed notation for -4 < decimal point <= 16 (with ".0" for
// integers), else d.ddde+XX with at least two exponent digits
void append_repr(std::string& out, double x) {
    if (x != x) { out += "nan"; return; }
    if (x == INFINITY || x == -INFINITY) { out += x < 0 ? "-inf" : "inf"; return; }
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest
    const char* p = buf;
    if (*p == '-') { out += '-'; ++p; }
    char digits[32];
    int nd = 0;
    for (; p < r.ptr && *p != 'e'; ++p)
        if (*p != '.') digits[nd++] = *p;
    int e10 = 0;
    if (p < r.ptr && *p == 'e') {
        ++p;
        const bool neg = *p == '-';
        if (*p == '-' || *p == '+') ++p;
        for (; p < r.ptr; ++p) e10 = e10 * 10 + (*p - '0');
        if (neg) e10 = -e10;
    }
    const int decpt = e10 + 1;
    if (-4 < decpt && decpt <= 16) {
        if (decpt <= 0) {
            out += "0.";
            out.append((size_t)(-decpt), '0');
            out.append(digits, (size_t)nd);
        } else if (decpt >= nd) {
            out.append(digits, (size_t)nd);
            out.append((size_t)(decpt - nd), '0');
            out += ".0";
        } else {
            out.append(digits, (size_t)decpt);
            out += '.';
            out.append(digits + decpt, (size_t)(nd - decpt));
        }
    } else {
        out += digits[0];
        if (nd > 1) { out += '.'; out.append(digits + 1, (size_t)(nd - 1)); }
        char ex[16];
        const int e = decpt - 1;
        snprintf(ex, sizeof ex, "e%c%02d", e < 0 ? '-' : '+', e < 0 ? -e : e);
        out += ex;
    }
}

// preds [n(n-1)/2] (pairs i < j, lexicographic) -> Newick text of the neighbour-joining tree, as nj.py writes it
void nj_newick(const float* preds, int32_t n, const char* const* ids, const int64_t* id_lens, bool clamp, std::string& out) {
    std::vector<std::string> labels((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        const char* id = ids[i] ? ids[i] : "";
        labels[(size_t)i].assign(id, (size_t)(id_lens ? id_lens[i] : (int64_t)strlen(id)));
    }
    out.clear();
    if (n == 1) { out = "(" + labels[0] + ");\n"; return; }
    auto at = [&](int64_t i, int64_t j) -> double {      // the symmetric matrix vec_to_phylip builds (float32, dm + dm.T)
        if (i == j) return 0.0;
        if (i > j) { const int64_t t = i; i = j; j = t; }
        return (double)(preds[i * n - i * (i + 1) / 2 + (j - i - 1)] + 0.0f);
    };
    if (n == 2) {
        char num[64];
        snprintf(num, sizeof num, "%.6g", at(0, 1) / 2);
        out = "(" + labels[0] + ":" + num + "," + labels[1] + ":" + num + ");\n";
        return;
    }
    auto fmt = [&](std::string& dst, double x) {
        if (clamp && x < 0) x = 0.0;
        append_repr(dst, x);
    };
    const size_t N = (size_t)n;
    std::vector<double> d(N * N), sub, r, dn(N);
    for (size_t i = 0; i < N; ++i)
        for (size_t j = 0; j < N; ++j) d[i * N + j] = at((int64_t)i, (int64_t)j);
    std::vector<int32_t> active((size_t)n);
    for (int32_t i = 0; i < n; ++i) active[(size_t)i] = i;
    while (active.size() > 3) {
        const size_t m = active.size();
        sub.resize(m * m);
        r.resize(m);
        for (size_t a = 0; a < m; ++a)
            for (size_t b = 0; b < m; ++b) sub[a * m + b] = d[(size_t)active[a] * N + (size_t)active[b]];
        for (size_t a = 0; a < m; ++a) r[a] = np_pairwise_sum(&sub[a * m], (int64_t)m);
        // q = (m - 2) * sub - r[:, None] - r[None, :], diagonal = inf; the first minimum in row-major order (np.argmin)
        const double mm2 = (double)((int64_t)m - 2);
        double best = INFINITY;
        size_t ba = 0, bb = 0;
        bool have = false, saw_nan = false;
        for (size_t a = 0; a < m && !saw_nan; ++a)
            for (size_t b = 0; b < m; ++b) {
                const double q = a == b ? INFINITY : (mm2 * sub[a * m + b] - r[a]) - r[b];
                if (q != q) { ba = a; bb = b; saw_nan = true; have = true; break; }     // np.argmin: the first NaN wins
                if (!have || q < best) { best = q; ba = a; bb = b; have = true; }
            }
        if (ba > bb) { const size_t t = ba; ba = bb; bb = t; }
        const size_t ia = (size_t)active[ba], ib = (size_t)active[bb];
        const double dab = sub[ba * m + bb];
        const double la = 0.5 * dab + (r[ba] - r[bb]) / (double)(2 * ((int64_t)m - 2));
        const double lb = dab - la;
        std::string nl;
        nl.reserve(labels[ia].size() + labels[ib].size() + 64);
        nl += '('; nl += labels[ia]; nl += ':'; fmt(nl, la); nl += ','; nl += labels[ib]; nl += ':'; fmt(nl, lb); nl += ')';
        for (size_t k = 0; k < N; ++k) dn[k] = 0.5 * ((d[ia * N + k] + d[ib * N + k]) - dab);
        for (size_t k = 0; k < N; ++k) { d[ia * N + k] = dn[k]; d[k * N + ia] = dn[k]; }
        d[ia * N + ia] = 0.0;
        labels[ia].swap(nl);
        std::string().swap(labels[ib]);
        active.erase(active.begin() + (std::ptrdiff_t)bb);
    }
    const size_t i = (size_t)active[0], j = (size_t)active[1], k = (size_t)active[2];
    const double li = 0.5 * ((d[i * N + j] + d[i * N + k]) - d[j * N + k]);
    const double lj = 0.5 * ((d[i * N + j] + d[j * N + k]) - d[i * N + k]);
    const double lk = 0.5 * ((d[i * N + k] + d[j * N + k]) - d[i * N + j]);
    out.reserve(labels[i].size() + labels[j].size() + labels[k].size() + 96);
    out += '('; out += labels[i]; out += ':'; fmt(out, li); out += ','; out += labels[j]; out += ':'; fmt(out, lj);
    out += ','; out += labels[k]; out += ':'; fmt(out, lk); out += ");\n";
}

}  // namespace

extern "C" {

int64_t pf_nj_newick_n(const float* preds, int32_t n, const char* const* ids, const int64_t* id_lens, int32_t clamp_negative,
                       char* out, int64_t cap) {
    if (!preds || n < 1 || !ids || !id_lens || (!out && cap > 0)) return PF_EINVAL;
    for (int32_t i = 0; i < n; ++i) if (id_lens[i] < 0) return PF_EINVAL;
    try {
        std::string text;
        nj_newick(preds, n, ids, id_lens, clamp_negative != 0, text);
        if (out && (int64_t)text.size() <= cap) memcpy(out, text.data(), text.size());
        return (int64_t)text.size();
    } catch (...) { return PF_ENOMEM; }
}

}  // extern "C"

extern "C" {

// ---- many files per call, on native threads (no GIL anywhere near the file system) ---------------------------
//
// The CLI's loop over a directory (infer_alns.py:97-117) opens, parses and writes one small file per
// alignment; at 20 x 200 the MI355X finishes 11,000 alignments a second, and CPython - one open() / read() /
// parse / format / write() per file under the GIL - delivered 6,000.  These entry points take a
// LIST of paths: a pool of std::threads reads and parses (or formats and writes) them, the results stay in a
// library-owned batch object that the writer later takes the sequence ids from, so Python touches neither
// the residues nor the ids of an alignment on the fast path.

}  // extern "C"

struct pf_fasta_batch {
    struct File {
        int32_t status = PF_EINVAL, n = 0, l = 0;
        int64_t detail = 0;
        std::vector<uint8_t> idx;          // [n][l]
        std::string ids;                   // the ids back to back
        std::vector<int64_t> spans;        // (offset, length) per id, into `ids`
    };
    std::vector<File> files;
};

namespace {

template <typename F>
void run_pool(int32_t count, int32_t threads, F&& fn) {
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, count));
    if (nt == 1) { for (int32_t i = 0; i < count; ++i) { try { fn(i); } catch (...) {} } return; }
    std::atomic<int32_t> next{0};
    // (an exception that escaped a std::thread body would be std::terminate: the callers' lambdas record failures per
    // item, and whatever they did not foresee ends here - ADVICE r05)
    auto work = [&] { for (int32_t i = next.fetch_add(1); i < count; i = next.fetch_add(1)) { try { fn(i); } catch (...) {} } };
    std::vector<std::thread> pool;
    pool.reserve(nt);
    // the calling thread is one of the workers; a thread the system refuses to start (EAGAIN) only narrows the pool -
    // nothing may throw across the C ABI
    for (int t = 1; t < nt; ++t) {
        try { pool.emplace_back(work); } catch (...) { break; }
    }
    work();
    for (auto& th : pool) th.join();
}

// whole file -> buf; 0 or -errno
int read_file(const char* path, std::string& buf) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return -errno;
    struct stat st;
    if (fstat(fd, &st) != 0) { const int e = errno; close(fd); return -e; }
    if (S_ISDIR(st.st_mode)) { close(fd); return -EISDIR; }
    buf.resize(st.st_size > 0 ? (size_t)st.st_size : 0);
    size_t got = 0;
    for (;;) {
        if (got == buf.size()) buf.resize(buf.size() ? buf.size() * 2 : 65536);      // grew, or size unknown
        const ssize_t r = read(fd, &buf[got], buf.size() - got);
        if (r < 0) { if (errno == EINTR) continue; const int e = errno; close(fd); return -e; }
        if (r == 0) break;
        got += (size_t)r;
        if (got == (size_t)st.st_size && st.st_size > 0) {                           // the common case: one read
            char probe;
            const ssize_t more = read(fd, &probe, 1);
            if (more <= 0) break;
            buf.push_back(probe);
            got += 1;
        }
    }
    close(fd);
    buf.resize(got);
    return 0;
}

void load_one(const char* path, pf_fasta_batch::File& f) {
    std::string data;
    const int rc = read_file(path, data);
    if (rc) { f.status = PF_EIO; f.detail = -rc; return; }
    int64_t nrec = 1;
    for (char c : data) nrec += (c == '>');
    f.idx.resize(data.size() ? data.size() : 1);
    std::vector<int64_t> spans((size_t)2 * nrec);
    f.status = pf_parse_fasta(data.data(), (int64_t)data.size(), f.idx.data(), (int64_t)f.idx.size(), spans.data(),
                              (int32_t)std::min<int64_t>(nrec, INT32_MAX), &f.n, &f.l, &f.detail);
    if (f.status != PF_OK) { f.idx.clear(); f.idx.shrink_to_fit(); return; }
    f.idx.resize((size_t)f.n * (size_t)f.l);
    f.idx.shrink_to_fit();
    f.spans.resize((size_t)2 * f.n);
    for (int32_t i = 0; i < f.n; ++i) {
        f.spans[2 * i] = (int64_t)f.ids.size();
        f.spans[2 * i + 1] = spans[2 * i + 1];
        f.ids.append(data.data() + spans[2 * i], (size_t)spans[2 * i + 1]);
    }
}

// 0 or -errno
int write_file(const char* path, const char* data, size_t len) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
    if (fd < 0) return -errno;
    size_t done = 0;
    while (done < len) {
        const ssize_t r = write(fd, data + done, len - done);
        if (r < 0) { if (errno == EINTR) continue; const int e = errno; close(fd); return -e; }
        done += (size_t)r;
    }
    return close(fd) == 0 ? 0 : -errno;
}

}  // namespace

extern "C" {

int pf_fasta_batch_load(const char* const* paths, int32_t count, int32_t threads, pf_fasta_batch_t** out) {
    if (!paths || count < 0 || !out) return PF_EINVAL;
    for (int32_t i = 0; i < count; ++i) if (!paths[i]) return PF_EINVAL;
    pf_fasta_batch* b = new (std::nothrow) pf_fasta_batch();
    if (!b) return PF_ENOMEM;
    try {
        b->files.resize((size_t)count);
        run_pool(count, threads, [&](int32_t i) {
            try { load_one(paths[i], b->files[(size_t)i]); }
            catch (const std::bad_alloc&) { b->files[(size_t)i].status = PF_ENOMEM; }
        });
    } catch (...) { delete b; return PF_ENOMEM; }
    *out = b;
    return PF_OK;
}

void pf_fasta_batch_free(pf_fasta_batch_t* b) { delete b; }

int32_t pf_fasta_batch_count(const pf_fasta_batch_t* b) { return b ? (int32_t)b->files.size() : 0; }

int pf_fasta_batch_infos(const pf_fasta_batch_t* b, int32_t* status, int32_t* n, int32_t* l, int64_t* detail) {
    if (!b || !status || !n || !l || !detail) return PF_EINVAL;
    for (size_t i = 0; i < b->files.size(); ++i) {
        const auto& f = b->files[i];
        status[i] = f.status; n[i] = f.n; l[i] = f.l; detail[i] = f.detail;
    }
    return PF_OK;
}

int pf_fasta_batch_id(const pf_fasta_batch_t* b, int32_t file, int32_t seq, const char** id, int64_t* len) {
    if (!b || !id || !len || file < 0 || (size_t)file >= b->files.size()) return PF_EINVAL;
    const auto& f = b->files[(size_t)file];
    if (f.status != PF_OK || seq < 0 || seq >= f.n) return PF_EINVAL;
    *id = f.ids.data() + f.spans[2 * (size_t)seq];
    *len = f.spans[2 * (size_t)seq + 1];
    return PF_OK;
}

int pf_fasta_batch_gather(const pf_fasta_batch_t* const* batches, const int32_t* file_idx, int32_t count, int32_t n,
                          int32_t l, uint8_t* dst) {
    if (!batches || !file_idx || count < 0 || n < 1 || l < 0 || !dst) return PF_EINVAL;
    const size_t per = (size_t)n * (size_t)l;
    for (int32_t k = 0; k < count; ++k) {
        const pf_fasta_batch* b = batches[k];
        if (!b || file_idx[k] < 0 || (size_t)file_idx[k] >= b->files.size()) return PF_EINVAL;
        const auto& f = b->files[(size_t)file_idx[k]];
        if (f.status != PF_OK || f.n != n || f.l != l || f.idx.size() != per) return PF_EINVAL;
        if (per) memcpy(dst + (size_t)k * per, f.idx.data(), per);
    }
    return PF_OK;
}

int pf_phylip_write_batch(const pf_fasta_batch_t* const* batches, const int32_t* file_idx, int32_t count, int32_t n,
                          const float* preds, const char* const* out_paths, const char* const* tree_paths, int32_t threads,
                          int32_t* status) {
    if (!batches || !file_idx || count < 0 || n < 2 || !preds || !out_paths || !status) return PF_EINVAL;
    for (int32_t k = 0; k < count; ++k) {
        const pf_fasta_batch* b = batches[k];
        if (!b || !out_paths[k] || (tree_paths && !tree_paths[k]) || file_idx[k] < 0 || (size_t)file_idx[k] >= b->files.size())
            return PF_EINVAL;
        const auto& f = b->files[(size_t)file_idx[k]];
        if (f.status != PF_OK || f.n != n) return PF_EINVAL;
    }
    const size_t P = (size_t)n * (size_t)(n - 1) / 2;
    try {       // (run_pool's own bookkeeping allocates: nothing may leave through the C ABI)
        run_pool(count, threads, [&](int32_t k) {
            try {
                const auto& f = batches[k]->files[(size_t)file_idx[k]];
                std::vector<const char*> ids((size_t)n);
                std::vector<int64_t> lens((size_t)n);
                for (int32_t i = 0; i < n; ++i) { ids[(size_t)i] = f.ids.data() + f.spans[2 * (size_t)i]; lens[(size_t)i] = f.spans[2 * (size_t)i + 1]; }
                const float* p = preds + (size_t)k * P;
                const int64_t need = format_phylip_impl(p, n, ids.data(), lens.data(), nullptr, 0);
                if (need < 0) { status[k] = -EINVAL; return; }
                std::string text((size_t)need, '\0');
                format_phylip_impl(p, n, ids.data(), lens.data(), &text[0], need);
                status[k] = write_file(out_paths[k], text.data(), text.size());
                if (tree_paths && status[k] == 0) {             // <stem>.nj.nwk beside it (infer_alns.py:120-123)
                    nj_newick(p, n, ids.data(), lens.data(), true, text);
                    status[k] = write_file(tree_paths[k], text.data(), text.size());
                }
            } catch (const std::bad_alloc&) { status[k] = -ENOMEM; }
            catch (...) { status[k] = -EIO; }
        });
    } catch (...) { return PF_ENOMEM; }
    return PF_OK;
}

}  // extern "C"
