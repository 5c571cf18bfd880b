// Host-side file-format helpers of the inference CLI (no GPU work, no HIP calls).
//
// The reference does both in per-element Python loops:
//   - FASTA -> one-hot   phyloformer/data.py:11-31   (here: bytes -> uint8 residue indices [N][L])
//   - distances -> PHYLIP text   infer_alns.py:14-25   (here: float [P] -> "%.10f" square matrix)
// At 60 x 500 those loops cost ~3 ms per alignment in CPython, more than the MI355X forward pass
// (2.3 ms), and hold the GIL.  These two functions do the same work in ~0.2 ms and are called through
// ctypes (GIL released), so the CLI's loader / writer threads scale with host cores.
#include <cstdint>
#include <cstdio>
#include <cstring>

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

int64_t pf_format_phylip(const float* preds, int32_t n, const char* const* ids, char* out, int64_t cap) {
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
        return preds[i * n - i * (i + 1) / 2 + (j - i - 1)];
    };
    for (int32_t i = 0; i < n; ++i) {
        const char* id = ids[i] ? ids[i] : "";
        put(id, (int64_t)strlen(id));
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

}  // extern "C"
