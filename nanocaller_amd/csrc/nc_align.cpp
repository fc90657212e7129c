// Host-side global alignment + the reference's allele extraction.
// Restates parasail.nw_trace(...).cigar as used by generate_indel_pileups.py:77-127 (parasail itself is a third-party
// dependency that is absent from this image: tie-breaking documented in include/nanocaller_hip.h, parity unpinned).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/nanocaller_hip.h"

namespace {
constexpr int32_t NEG = -(1 << 29);
enum : uint8_t { H_DIAG = 0, H_DEL = 1, H_INS = 2, E_EXT = 4, F_EXT = 8 };   // E: gap consuming s2 (D), F: gap consuming s1 (I)

// -> CIGAR in alignment order
void nw_cigar(const char *s1, int n1, const char *s2, int n2, int open, int extend, int match, int mismatch,
              std::vector<int32_t> &ops, std::vector<int32_t> &cnts)
{
    const int W = n2 + 1;
    std::vector<int32_t> H((size_t)(n1 + 1) * W), E((size_t)(n1 + 1) * W, NEG), F((size_t)(n1 + 1) * W, NEG);
    std::vector<uint8_t> T((size_t)(n1 + 1) * W, 0);
    H[0] = 0;
    for (int j = 1; j <= n2; j++) { H[j] = -open - (j - 1) * extend; E[j] = H[j]; T[j] = (uint8_t)(H_DEL | (j > 1 ? E_EXT : 0)); }
    for (int i = 1; i <= n1; i++) {
        H[(size_t)i * W] = -open - (i - 1) * extend;
        F[(size_t)i * W] = H[(size_t)i * W];
        T[(size_t)i * W] = (uint8_t)(H_INS | (i > 1 ? F_EXT : 0));
        for (int j = 1; j <= n2; j++) {
            const size_t c = (size_t)i * W + j, up = c - W, left = c - 1, dg = c - W - 1;
            uint8_t t = 0;
            const int32_t e_open = H[left] - open, e_ext = E[left] - extend;
            int32_t e = e_open;
            if (e_ext >= e_open) { e = e_ext; t |= E_EXT; }
            const int32_t f_open = H[up] - open, f_ext = F[up] - extend;
            int32_t f = f_open;
            if (f_ext >= f_open) { f = f_ext; t |= F_EXT; }
            const int32_t d = H[dg] + (s1[i - 1] == s2[j - 1] ? match : mismatch);
            int32_t h = d;
            uint8_t w = H_DIAG;
            if (e > h) { h = e; w = H_DEL; }
            if (f > h) { h = f; w = H_INS; }
            H[c] = h; E[c] = e; F[c] = f;
            T[c] = (uint8_t)(t | w);
        }
    }
    // traceback
    std::vector<int32_t> rops;
    int i = n1, j = n2;
    int state = -1;                                   // -1: follow H, 1: inside a D gap (E), 2: inside an I gap (F)
    while (i > 0 || j > 0) {
        const size_t c = (size_t)i * W + j;
        if (state < 0) {
            const int w = T[c] & 3;
            if (w == H_DIAG) { rops.push_back(s1[i - 1] == s2[j - 1] ? 7 : 8); i--; j--; continue; }
            state = w == H_DEL ? 1 : 2;
        }
        if (state == 1) {                             // consume s2[j-1]
            rops.push_back(2);
            const bool ext = (T[c] & E_EXT) != 0;
            j--;
            if (!ext) state = -1;
        } else {                                      // consume s1[i-1]
            rops.push_back(1);
            const bool ext = (T[c] & F_EXT) != 0;
            i--;
            if (!ext) state = -1;
        }
    }
    ops.clear(); cnts.clear();
    for (size_t k = rops.size(); k-- > 0;) {
        if (!ops.empty() && ops.back() == rops[k]) cnts.back()++;
        else { ops.push_back(rops[k]); cnts.push_back(1); }
    }
}
}   // namespace

extern "C" int nc_nw_cigar(const char *s1, int32_t n1, const char *s2, int32_t n2, int32_t open, int32_t extend, int32_t match,
                           int32_t mismatch, int32_t *ops, int32_t *counts, int32_t cap, int32_t *n_ops)
{
    if (n1 < 0 || n2 < 0 || (n1 && !s1) || (n2 && !s2) || !n_ops || (cap && (!ops || !counts))) return NC_ERR_ARG;
    if ((int64_t)(n1 + 1) * (n2 + 1) > (int64_t)1 << 26) return NC_ERR_ARG;       // 64 M cells: far beyond the 260-base windows of this path
    std::vector<int32_t> o, c;
    nw_cigar(s1, n1, s2, n2, open, extend, match, mismatch, o, c);
    *n_ops = (int32_t)o.size();
    if ((int32_t)o.size() > cap) return NC_ERR_CAPACITY;
    for (size_t k = 0; k < o.size(); k++) { ops[k] = o[k]; counts[k] = c[k]; }
    return NC_OK;
}

// generate_indel_pileups.py:77-127, statement by statement (ref_cnt / alt_cnt are indexed by CIGAR op as there)
extern "C" int nc_allele_prediction(const char *alt, int32_t n_alt, const char *ref_seq, int32_t n_ref, int32_t max_range,
                                    int32_t *ref_len, int32_t *alt_len)
{
    if (!ref_len || !alt_len || n_alt < 0 || n_ref < 0 || (n_alt && !alt) || (n_ref && !ref_seq)) return NC_ERR_ARG;
    if ((int64_t)(n_alt + 1) * (n_ref + 1) > (int64_t)1 << 26) return NC_ERR_ARG;
    std::vector<int32_t> ops, cnts;
    nw_cigar(alt, n_alt, ref_seq, n_ref, 9, 1, 20, -10, ops, cnts);
    bool indel = false, mm_before = false;
    int64_t ref_cnt[10] = {0}, alt_cnt[10] = {0}, mm_after[2] = {0, 0};
    auto sum10 = [](const int64_t *a) { int64_t t = 0; for (int k = 0; k < 10; k++) t += a[k]; return t; };
    auto clamp = [](int64_t v, int32_t n) { return (int32_t)(v < 0 ? (v + n < 0 ? 0 : v + n) : (v > n ? n : v)); };   // Python slice s[:v]
    int op = 0;
    int64_t cnt = 0;
    for (size_t k = 0; k < ops.size(); k++) {
        op = ops[k];
        cnt = cnts[k];
        if (op == 8 || op == 7) {
            ref_cnt[op] += cnt;
            alt_cnt[op] += cnt;
            if (indel) mm_after[op - 7] += cnt;
            else mm_before = true;
        }
        if (op == 1) { alt_cnt[op] += cnt; mm_after[0] = mm_after[1] = 0; indel = true; }
        if (op == 2) { ref_cnt[op] += cnt; mm_after[0] = mm_after[1] = 0; indel = true; }
        if (!indel && sum10(ref_cnt) >= (int64_t)max_range + 10) {
            if (ref_cnt[8]) {
                const int64_t out_len = op == 8 ? sum10(ref_cnt) : sum10(ref_cnt) - cnt;
                *ref_len = clamp(out_len, n_ref);
                *alt_len = clamp(out_len, n_alt);
                return NC_OK;
            }
            *ref_len = -1;
            *alt_len = -1;
            return NC_OK;
        }
        if (indel && mm_after[0] + mm_after[1] > 20) break;
    }
    int64_t ro = op == 8 ? sum10(ref_cnt) : sum10(ref_cnt) - cnt;
    int64_t ao = op == 8 ? sum10(alt_cnt) : sum10(alt_cnt) - cnt;
    if (!mm_before) { ro += 1; ao += 1; }
    *ref_len = clamp(ro, n_ref);
    *alt_len = clamp(ao, n_alt);
    return NC_OK;
}
