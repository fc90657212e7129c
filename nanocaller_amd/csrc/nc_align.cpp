// Host-side global alignment + the reference's allele extraction.
// Restates parasail.nw_trace(...).cigar as used by generate_indel_pileups.py:77-127 (parasail itself is a third-party
// dependency that is absent from this image: tie-breaking documented in include/nanocaller_hip.h, parity unpinned).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/nanocaller_hip.h"

namespace {
constexpr int32_t NEG = -(1 << 29);
enum : uint8_t { H_DIAG = 0, H_DEL = 1, H_INS = 2, E_EXT = 4, F_EXT = 8 };   // E: gap consuming s2 (D), F: gap consuming s1 (I)

// -> CIGAR in alignment order
// free_tail: the alignment is anchored at the start only; it ends at the best cell of the last row or last column (ties:
// the cell closest to the corner along the last row first), and the unaligned tail follows as one trailing gap.
void nw_cigar(const char *s1, int n1, const char *s2, int n2, int open, int extend, int match, int mismatch,
              std::vector<int32_t> &ops, std::vector<int32_t> &cnts, bool free_tail = false)
{
    // Only the traceback codes are kept for every cell (1 byte each); H and F live in two rolling rows and E in a scalar: a
    // 160 x 161 alignment touches ~30 KB instead of 340 KB.  Buffers are per thread and reused between calls.
    const int W = n2 + 1;
    thread_local std::vector<int32_t> Hrow[2], Frow[2], lastcol;
    thread_local std::vector<uint8_t> T;
    for (int k = 0; k < 2; k++) { if ((int)Hrow[k].size() < W) { Hrow[k].resize(W); Frow[k].resize(W); } }
    if (T.size() < (size_t)(n1 + 1) * W) T.resize((size_t)(n1 + 1) * W);
    if ((int)lastcol.size() < n1 + 1) lastcol.resize(n1 + 1);
    int32_t *Hp = Hrow[0].data(), *Hc = Hrow[1].data(), *Fp = Frow[0].data(), *Fc = Frow[1].data();
    Hp[0] = 0;
    Fp[0] = NEG;
    T[0] = 0;
    for (int j = 1; j <= n2; j++) { Hp[j] = -open - (j - 1) * extend; Fp[j] = NEG; T[j] = (uint8_t)(H_DEL | (j > 1 ? E_EXT : 0)); }
    lastcol[0] = Hp[n2];
    for (int i = 1; i <= n1; i++) {
        uint8_t *Ti = T.data() + (size_t)i * W;
        Hc[0] = -open - (i - 1) * extend;
        Fc[0] = Hc[0];
        Ti[0] = (uint8_t)(H_INS | (i > 1 ? F_EXT : 0));
        int32_t e_left = NEG;                                       // E[i][0]
        const char a = s1[i - 1];
        for (int j = 1; j <= n2; j++) {
            uint8_t t = 0;
            const int32_t e_open = Hc[j - 1] - open, e_ext = e_left - extend;
            int32_t e = e_open;
            if (e_ext >= e_open) { e = e_ext; t |= E_EXT; }
            const int32_t f_open = Hp[j] - open, f_ext = Fp[j] - extend;
            int32_t f = f_open;
            if (f_ext >= f_open) { f = f_ext; t |= F_EXT; }
            const int32_t d = Hp[j - 1] + (a == s2[j - 1] ? match : mismatch);
            int32_t h = d;
            uint8_t w = H_DIAG;
            if (e > h) { h = e; w = H_DEL; }
            if (f > h) { h = f; w = H_INS; }
            Hc[j] = h; e_left = e; Fc[j] = f;
            Ti[j] = (uint8_t)(t | w);
        }
        lastcol[i] = Hc[n2];
        std::swap(Hp, Hc);
        std::swap(Fp, Fc);
    }
    // Hp now holds the last row (row n1; row 0 when n1 == 0)
    // traceback
    std::vector<int32_t> rops;
    rops.reserve((size_t)n1 + n2 + 2);
    int i = n1, j = n2;
    if (free_tail && n1 > 0 && n2 > 0) {
        int32_t best = Hp[n2];
        for (int jj = n2 - 1; jj >= 0; jj--)                       // last row: the rest of s2 is unaligned
            if (Hp[jj] > best) { best = Hp[jj]; i = n1; j = jj; }
        for (int ii = n1 - 1; ii >= 0; ii--)                       // last column: the rest of s1 is unaligned
            if (lastcol[ii] > best) { best = lastcol[ii]; i = ii; j = n2; }
        for (int t = n2; t > j; t--) rops.push_back(2);
        for (int t = n1; t > i; t--) rops.push_back(1);
    }
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
    if (n_alt == n_ref && n_alt > 0 && memcmp(alt, ref_seq, (size_t)n_alt) == 0) {
        ops.push_back(7);                                  // identical strings: the all-match diagonal is the unique optimum --
        cnts.push_back(n_alt);                             // most consensus sequences equal their reference window: skip the DP
    } else
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


// ---- Star alignment of a read set to its reference window (replaces the MUSCLE subprocess of
// generate_indel_pileups.py:24-44 -- not MUSCLE's algorithm, so not comparable row by row with it; judged by call concordance).
// Every read is aligned to the reference window on its own (Gotoh, anchored at the window start, free tail), then the
// pairwise alignments are merged in reference coordinates: a reference slot gets as many insertion columns as the longest
// insertion any read has there, shorter insertions are left-justified.  Symbols as in msa(): A=0 G=1 T=2 C=3 gap=4; any other
// read character is kept (5) so that the caller can reject it as the reference's `KeyError` does.
#include <thread>

#include "nc_host.h"

static inline uint8_t sym_code(char c)
{
    switch (c) { case 'A': return 0; case 'G': return 1; case 'T': return 2; case 'C': return 3; case '-': return 4; default: return 5; }
}

extern "C" int nc_star_msa(int32_t n_reads, const char *reads, const int32_t *read_off, const char *ref, int32_t n_ref, int32_t open,
                           int32_t extend, int32_t match, int32_t mismatch, int32_t col_cap, uint8_t *rows, uint8_t *ref_row,
                           int32_t *n_cols)
{
    if (n_reads < 0 || n_ref < 1 || !ref || !n_cols || (n_reads && (!reads || !read_off)) || col_cap < n_ref || !ref_row || (n_reads && !rows))
        return NC_ERR_ARG;
    for (int r = 0; r < n_reads; r++)
        if (read_off[r + 1] < read_off[r] || (int64_t)(read_off[r + 1] - read_off[r] + 1) * (n_ref + 1) > (int64_t)1 << 26) return NC_ERR_ARG;
    std::vector<std::vector<int32_t>> ops((size_t)n_reads), cnts((size_t)n_reads);
    int T = nc_host_cpus();
    if (T > 32) T = 32;
    if (T > n_reads) T = n_reads;
    auto work = [&](int t) {
        for (int r = t; r < n_reads; r += (T > 0 ? T : 1))
            nw_cigar(reads + read_off[r], read_off[r + 1] - read_off[r], ref, n_ref, open, extend, match, mismatch, ops[(size_t)r], cnts[(size_t)r], true);
    };
    if (T <= 1) { if (n_reads) work(0); }
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    // insertion columns per reference slot (slot j = before reference position j; slot n_ref = after the last)
    std::vector<int32_t> ins((size_t)n_ref + 1, 0);
    for (int r = 0; r < n_reads; r++) {
        int j = 0;
        for (size_t k = 0; k < ops[(size_t)r].size(); k++) {
            const int op = ops[(size_t)r][k], c = cnts[(size_t)r][k];
            if (op == 1) { if (c > ins[(size_t)j]) ins[(size_t)j] = c; }
            else j += c;
        }
    }
    std::vector<int32_t> col((size_t)n_ref + 1);                    // column of reference position j (col[n_ref] = end)
    int32_t acc = 0;
    for (int j = 0; j <= n_ref; j++) { acc += ins[(size_t)j]; col[(size_t)j] = acc + j; }
    const int32_t nc = col[(size_t)n_ref];
    *n_cols = nc;
    if (nc > col_cap) return NC_ERR_CAPACITY;
    memset(ref_row, 4, (size_t)nc);
    for (int j = 0; j < n_ref; j++) ref_row[col[(size_t)j]] = sym_code(ref[j]);
    for (int r = 0; r < n_reads; r++) {
        uint8_t *row = rows + (size_t)r * col_cap;
        memset(row, 4, (size_t)nc);
        const char *q = reads + read_off[r];
        int i = 0, j = 0;
        for (size_t k = 0; k < ops[(size_t)r].size(); k++) {
            const int op = ops[(size_t)r][k], c = cnts[(size_t)r][k];
            if (op == 7 || op == 8) { for (int t = 0; t < c; t++, i++, j++) row[col[(size_t)j]] = sym_code(q[i]); }
            else if (op == 2) j += c;
            else { const int32_t s0 = col[(size_t)j] - ins[(size_t)j]; for (int t = 0; t < c; t++, i++) row[s0 + t] = sym_code(q[i]); }
        }
    }
    return NC_OK;
}


// nc_allele_prediction for many (consensus, reference window) pairs at once, on the usable host cores (the three calls per
// indel anchor are independent).  alt i = alts[alt_off[i] .. alt_off[i+1]), ref i likewise.
extern "C" int nc_allele_prediction_batch(int32_t n, const char *alts, const int32_t *alt_off, const char *refs, const int32_t *ref_off,
                                          const int32_t *max_range, int32_t *ref_len, int32_t *alt_len)
{
    if (n < 0 || (n && (!alt_off || !ref_off || !max_range || !ref_len || !alt_len))) return NC_ERR_ARG;
    int T = nc_host_cpus();
    if (T > 32) T = 32;
    if (T > n / 8) T = n / 8;
    std::vector<int> rc((size_t)(T > 0 ? T : 1), NC_OK);
    auto work = [&](int t, int stride) {
        for (int i = t; i < n; i += stride) {
            const int r = nc_allele_prediction(alts + alt_off[i], alt_off[i + 1] - alt_off[i], refs + ref_off[i], ref_off[i + 1] - ref_off[i],
                                               max_range[i], &ref_len[i], &alt_len[i]);
            if (r != NC_OK) rc[(size_t)t] = r;
        }
    };
    if (T <= 1) work(0, 1);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t, T);
        for (auto &x : th) x.join();
    }
    for (int r : rc) if (r != NC_OK) return r;
    return NC_OK;
}

// Consensus rows of nc_star_msa_tensor (cns [n_sets][max_cols]: symbols 0..3 = AGTC, 4 = gap, 5.. = other, the first
// min(n_cols[s], max_cols) of a row valid) -> the strings msa() hands to allele_prediction (generate_indel_pileups.py:58-61: the
// consensus with the gap symbols removed), concatenated in `out` (capacity n_sets * max_cols suffices) with off[s] .. off[s + 1]
// delimiting set s.  On the usable host cores: a contig arm's consensus matrix is tens of megabytes.
extern "C" int nc_consensus_strings(const uint8_t *cns, int32_t n_sets, int32_t max_cols, const int32_t *n_cols, char *out, int64_t *off)
{
    if (n_sets < 0 || max_cols < 1 || (n_sets && (!cns || !n_cols || !out)) || !off) return NC_ERR_ARG;
    off[0] = 0;
    if (n_sets == 0) return NC_OK;
    int T = nc_host_cpus();
    if (T > 16) T = 16;
    if (T > n_sets / 256) T = n_sets / 256;
    if (T < 1) T = 1;
    auto each = [&](auto &&fn) {
        if (T == 1) { fn(0, n_sets); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] { fn((int32_t)((int64_t)n_sets * t / T), (int32_t)((int64_t)n_sets * (t + 1) / T)); });
        for (auto &x : th) x.join();
    };
    each([&](int32_t s0, int32_t s1) {
        for (int32_t s = s0; s < s1; s++) {
            const uint8_t *row = cns + (size_t)s * max_cols;
            const int32_t n = std::min(std::max(n_cols[s], 0), max_cols);
            int64_t k = 0;
            for (int32_t c = 0; c < n; c++) k += row[c] != 4;
            off[s + 1] = k;
        }
    });
    for (int32_t s = 0; s < n_sets; s++) off[s + 1] += off[s];
    static const char LET[8] = {'A', 'G', 'T', 'C', '-', 'N', 'N', 'N'};
    each([&](int32_t s0, int32_t s1) {
        for (int32_t s = s0; s < s1; s++) {
            const uint8_t *row = cns + (size_t)s * max_cols;
            const int32_t n = std::min(std::max(n_cols[s], 0), max_cols);
            char *d = out + off[s];
            for (int32_t c = 0; c < n; c++)
                if (row[c] != 4) *d++ = LET[row[c] & 7];
        }
    });
    return NC_OK;
}
