// Native SNP genotype rules + VCF record text (host).  Restates snpCaller.py:113-198 exactly
// as nanocaller_amd/snpCaller.py::snp_vcf_lines does, at ~0.2 us per record instead of ~10 us in Python.
// `order` is numpy's argsort of the four probabilities (ascending) computed by the caller, so ties resolve as in
// the reference on the same machine (quirk E15).  Number formatting: printf("%.Nf") and Python's '%.Nf' are both
// correctly rounded conversions of the same double.
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <thread>

#include "nc_host.h"
#include <vector>

#include "../../include/nanocaller_hip.h"

namespace {
const char B[4] = {'A', 'G', 'T', 'C'};                                   // snpCaller.py:14
inline double qual(float p, double cap, double mult)
{
    const double q = mult * std::log10(1e-10 + 1 - (double)p);            // float64, numpy<2 semantics (E7)
    return q < cap ? q : cap;
}

// ---- number formatting without printf.  '%.Nf' of a double is the correctly rounded (ties to even, on the EXACT binary
// value) decimal with N digits; glibc and CPython both do exactly that.  For |x| < 2^40 and N <= 4 the scaled value
// m * 10^N * 2^e fits a 128-bit integer, so the rounding is done exactly in integer arithmetic.
inline char *put_uint(char *o, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *o++ = tmp[--n];
    return o;
}
inline char *put_int(char *o, int64_t v)
{
    if (v < 0) { *o++ = '-'; return put_uint(o, (uint64_t)(-(v + 1)) + 1); }
    return put_uint(o, (uint64_t)v);
}
inline char *put_fixed(char *o, double x, int prec)                        // prec in 1..4; exact for |x| < 2^40
{
    static const uint32_t P10[5] = {1, 10, 100, 1000, 10000};
    uint64_t bits;
    memcpy(&bits, &x, 8);
    const int be = (int)((bits >> 52) & 0x7ff);
    // what Python's format() prints for values outside the exact range: a non-finite probability (a model that overflowed) gives 'nan' /
    // 'inf' in the reference's VCF too; the shifts below are defined for finite |x| < 2^40 only
    if (be == 0x7ff) {
        const bool nan = (bits & ((1ull << 52) - 1)) != 0;
        if (!nan && (bits >> 63)) *o++ = '-';
        memcpy(o, nan ? "nan" : "inf", 3);
        return o + 3;
    }
    if (be >= 1023 + 40) {                                                 // huge but finite: digits by the C library (never on a quality value)
        const int w = snprintf(o, 48, "%.*e", prec, x);                     // (bounded: a record keeps 64 spare bytes)
        return o + (w < 47 ? w : 47);
    }
    if (bits >> 63) *o++ = '-';                                            // also "-0.000" for a negative zero / tiny negative
    uint64_t m = bits & ((1ull << 52) - 1);
    int e;
    if (be == 0) e = -1074;                                                // subnormal
    else { m |= 1ull << 52; e = be - 1075; }
    unsigned __int128 N = (unsigned __int128)m * P10[prec];                // < 2^53 * 2^14
    unsigned __int128 q;
    if (e >= 0) q = N << e;                                                // |x| < 2^40 -> e <= -13 for nonzero m; kept for completeness
    else {
        const int sft = -e;
        if (sft >= 128) q = 0;                                             // N < 2^67 <= half
        else {
            q = N >> sft;
            const unsigned __int128 rem = N & ((((unsigned __int128)1) << sft) - 1), half = ((unsigned __int128)1) << (sft - 1);
            if (rem > half || (rem == half && (q & 1))) q++;
        }
    }
    const uint64_t qi = (uint64_t)q, ip = qi / P10[prec];
    uint32_t fp = (uint32_t)(qi % P10[prec]);
    o = put_uint(o, ip);
    *o++ = '.';
    for (int k = prec - 1; k >= 0; k--) { o[k] = (char)('0' + fp % 10); fp /= 10; }
    return o + prec;
}
inline char *put_str(char *o, const char *s, size_t n) { memcpy(o, s, n); return o + n; }
#define PUT_LIT(o, lit) put_str(o, lit, sizeof(lit) - 1)
}   // namespace

static int format_range(const char *chrom, int64_t n, const int32_t *pos, const int32_t *ref, const float *probs,
                        const int32_t *order, const int32_t *dp, const double *freq, const int32_t *fwd,
                        const int32_t *rev, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes);

// Host threads of the formatter / sorter: NC_VCF_THREADS, else the usable CPUs (nc_host.h: affinity and cgroup quota) minus two (the thread that feeds the
// GPU and the HIP runtime's own keep their cores while a worker formats the previous group), at most 32.
static int host_threads()
{
    if (const char *e = getenv("NC_VCF_THREADS")) {
        const int t = atoi(e);
        if (t >= 1) return t > 256 ? 256 : t;
    }
    const int hw = nc_host_cpus();
    const int t = hw > 4 ? hw - 2 : hw;
    return t > 32 ? 32 : t;
}

// Records are independent: ranges of sites are formatted by host threads into disjoint slices of `out` (400-byte
// budget per record) and compacted in order afterwards.
extern "C" int nc_snp_vcf_format(const char *chrom, int64_t n, const int32_t *pos, const int32_t *ref, const float *probs,
                                 const int32_t *order, const int32_t *dp, const double *freq, const int32_t *fwd,
                                 const int32_t *rev, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes)
{
    if (!chrom || n < 0 || !n_bytes) return NC_ERR_ARG;
    const int64_t per = 400 + (int64_t)strlen(chrom);
    int T = host_threads();
    if (n < 20000 || cap < n * per) T = 1;
    if (T == 1) return format_range(chrom, n, pos, ref, probs, order, dp, freq, fwd, rev, haploid, out, cap, n_bytes);
    // Two parallel phases: every thread formats its range of sites into its slice of `out` (400-byte budget per record),
    // then -- once all lengths are known -- the slices are closed up.  Slice t moves left to its final place only after the
    // slices whose bytes it would overwrite have moved (a slice depends on earlier slices only), so the compaction runs in
    // parallel as well instead of as one 70 MB memmove.
    std::vector<int64_t> nb((size_t)T, 0), src((size_t)T, 0), dst((size_t)T + 1, 0);
    std::vector<int> rc((size_t)T, NC_OK);
    const int64_t chunk = (n + T - 1) / T;
    int used = 0;
    for (int t = 0; t < T; t++) if ((int64_t)t * chunk < n) used = t + 1;
    {
        std::vector<std::thread> th;
        for (int t = 0; t < used; t++) {
            const int64_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
            src[(size_t)t] = a * per;
            th.emplace_back([=, &nb, &rc]() {
                rc[(size_t)t] = format_range(chrom, b - a, pos + a, ref + a, probs + 4 * a, order ? order + 4 * a : nullptr, dp + a, freq + a,
                                             fwd ? fwd + 4 * a : nullptr, rev ? rev + 4 * a : nullptr, haploid, out + a * per, (b - a) * per,
                                             &nb[(size_t)t]);
            });
        }
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < used; t++) {
        if (rc[(size_t)t] != NC_OK) return rc[(size_t)t];
        dst[(size_t)t + 1] = dst[(size_t)t] + nb[(size_t)t];
    }
    {
        // moved[t] = 1 once slice t sits at dst[t].  Slice t's destination [dst[t], dst[t] + nb[t]) may overlap the SOURCE of
        // earlier slices s < t only (dst[t] <= src[t]); it waits for exactly those.
        std::vector<std::atomic<int>> moved((size_t)used);
        for (auto &m : moved) m.store(0);
        std::vector<std::thread> th;
        for (int t = 0; t < used; t++)
            th.emplace_back([&, t]() {
                const int64_t d0 = dst[(size_t)t], d1 = d0 + nb[(size_t)t];
                for (int s2 = t - 1; s2 >= 0; s2--) {
                    const int64_t s0 = src[(size_t)s2], s1 = s0 + nb[(size_t)s2];
                    if (s1 <= d0) break;                                // earlier slices lie further left still
                    if (s0 < d1) while (moved[(size_t)s2].load(std::memory_order_acquire) == 0) std::this_thread::yield();
                }
                if (d0 != src[(size_t)t] && nb[(size_t)t]) memmove(out + d0, out + src[(size_t)t], (size_t)nb[(size_t)t]);
                moved[(size_t)t].store(1, std::memory_order_release);
            });
        for (auto &x : th) x.join();
    }
    *n_bytes = dst[(size_t)used];
    return NC_OK;
}

static int format_range(const char *chrom, int64_t n, const int32_t *pos, const int32_t *ref, const float *probs,
                        const int32_t *order, const int32_t *dp, const double *freq, const int32_t *fwd,
                        const int32_t *rev, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes)
{
    if (!chrom || n < 0 || !n_bytes || (n && (!pos || !ref || !probs || !dp || !freq || !out)) || (n && !haploid && (!order || !fwd || !rev)))
        return NC_ERR_ARG;
    int64_t w = 0;
    const size_t lc = strlen(chrom);
    for (int64_t j = 0; j < n; j++) {
        if (cap - w < 400 + (int64_t)lc) return NC_ERR_CAPACITY;
        char *o = out + w;
        const float *pr = probs + 4 * j;
        const int r = ref[j];
        const int d = dp[j];
        // genotype decision first (the record's ALT / QUAL / FILTER / sample column depend on it)
        int kind;                    // 0 haploid, 1 het 0/1, 2 het 1/2, 3 hom 1/1, 4 REF, 5 LOW
        int a1 = 0, a2 = 0;
        double q = 0.0;
        const int32_t *f = nullptr, *v = nullptr;
        if (haploid) {                                                                                     // :184-198
            int p = 0;
            for (int k = 1; k < 4; k++) if (pr[k] > pr[p]) p = k;                                          // np.argmax: first maximum
            kind = 0; a1 = p; q = qual(pr[p], 999.0, -100.0);
        } else {
            f = fwd + 4 * j; v = rev + 4 * j;
            const int p1 = order[4 * j + 3], p2 = order[4 * j + 2];
            int k = 0;
            for (int b = 0; b < 4; b++) k += pr[b] >= 0.5f;                                                // :122
            kind = -1;
            if (k >= 2) {
                if (p1 == r || (p2 == r && pr[p2] >= 0.5f)) { kind = 1; a1 = p1 == r ? p2 : p1; q = qual(pr[p2], 99.0, -10.0); }      // :132, :138
                else if (p2 != r && p1 != r && pr[p2] >= 0.5f) { kind = 2; a1 = p1; a2 = p2; q = qual(pr[p2], 99.0, -10.0); }        // :143
            } else if (k == 1 && r != p1 && pr[p1] >= 0.5f) { kind = 3; a1 = p1; q = qual(pr[p1], 99.0, -10.0); }                   // :150
            else if (k == 1 && r == p1) { kind = 4; q = qual(pr[p1], 99.0, -10.0); }                                                 // :157
            else kind = 5;                                                                                                           // :161
            if (kind < 0) continue;                          // k >= 2 with neither rule matching: the reference writes no record
        }
        o = put_str(o, chrom, lc); *o++ = '\t';
        o = put_int(o, pos[j]);
        o = PUT_LIT(o, "\t.\t"); *o++ = B[r]; *o++ = '\t';
        if (kind == 4 || kind == 5) *o++ = '.';
        else { *o++ = B[a1]; if (kind == 2) { *o++ = ','; *o++ = B[a2]; } }
        *o++ = '\t';
        if (kind == 5) o = PUT_LIT(o, "0.000"); else o = put_fixed(o, q, 3);
        if (kind == 0) o = (a1 != r) ? PUT_LIT(o, "\tPASS\t") : PUT_LIT(o, "\tREF\t");
        else if (kind == 4) o = PUT_LIT(o, "\tREF\t");
        else if (kind == 5) o = PUT_LIT(o, "\tLOW\t");
        else o = PUT_LIT(o, "\tPASS\t");
        o = PUT_LIT(o, "PR=");                                                                             // :127
        o = put_fixed(o, (double)pr[0], 4); *o++ = ',';
        o = put_fixed(o, (double)pr[3], 4); *o++ = ',';
        o = put_fixed(o, (double)pr[1], 4); *o++ = ',';
        o = put_fixed(o, (double)pr[2], 4);
        o = PUT_LIT(o, ";FQ=");
        o = put_fixed(o, freq[j], 4);
        o = PUT_LIT(o, "\tGT:DP:VF:AD:ADF:ADR\t");
        switch (kind) {
        case 0:
            o = PUT_LIT(o, "1/1:"); o = put_int(o, d); *o++ = ':'; o = put_fixed(o, freq[j], 4); o = PUT_LIT(o, ":.:.:.\n");
            break;
        case 1: case 3: {
            const int rf = f[r], rr = v[r];
            o = kind == 1 ? PUT_LIT(o, "0/1:") : PUT_LIT(o, "1/1:");
            o = put_int(o, d); *o++ = ':';
            o = put_fixed(o, (double)(f[a1] + v[a1]) / d, 4); *o++ = ':';
            o = put_int(o, rf + rr); *o++ = ','; o = put_int(o, f[a1] + v[a1]); *o++ = ':';
            o = put_int(o, rf); *o++ = ','; o = put_int(o, f[a1]); *o++ = ':';
            o = put_int(o, rr); *o++ = ','; o = put_int(o, v[a1]); *o++ = '\n';
            break;
        }
        case 2: {
            const int rf = f[r], rr = v[r];
            o = PUT_LIT(o, "1/2:"); o = put_int(o, d); *o++ = ':';
            o = put_fixed(o, (double)(f[a1] + v[a1]) / d, 4); *o++ = ','; o = put_fixed(o, (double)(f[a2] + v[a2]) / d, 4); *o++ = ':';
            o = put_int(o, rf + rr); *o++ = ','; o = put_int(o, f[a1] + v[a1]); *o++ = ','; o = put_int(o, f[a2] + v[a2]); *o++ = ':';
            o = put_int(o, rf); *o++ = ','; o = put_int(o, f[a1]); *o++ = ','; o = put_int(o, f[a2]); *o++ = ':';
            o = put_int(o, rr); *o++ = ','; o = put_int(o, v[a1]); *o++ = ','; o = put_int(o, v[a2]); *o++ = '\n';
            break;
        }
        default:
            o = PUT_LIT(o, "./.:"); o = put_int(o, d); o = PUT_LIT(o, ":.:.:.:.\n");
        }
        w = o - out;
    }
    *n_bytes = w;
    return NC_OK;
}

// Ascending argsort of n rows of 4 floats (insertion sort: stable, ties keep index order), multi-threaded.  Rows that
// contain two equal values are listed in tie_idx (ascending) so that the caller can re-sort exactly those rows with the
// reference's own sorter: numpy's tie order is implementation dependent (quirk E15) and must not be guessed here.
extern "C" int nc_argsort4(const float *probs, int64_t n, int32_t *order, int64_t *n_ties, int64_t *tie_idx, int64_t tie_cap)
{
    if (n < 0 || (n && (!probs || !order)) || !n_ties || (tie_cap && !tie_idx)) return NC_ERR_ARG;
    int T = host_threads();
    if (n < 50000) T = 1;
    const int64_t chunk = (n + T - 1) / T;
    std::vector<std::vector<int64_t>> ties((size_t)T);
    auto work = [&](int t) {
        const int64_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
        for (int64_t j = a; j < b; j++) {
            const float *p = probs + 4 * j;
            int o[4] = {0, 1, 2, 3};
            for (int i = 1; i < 4; i++) {
                const int k = o[i];
                int q = i - 1;
                while (q >= 0 && p[o[q]] > p[k]) { o[q + 1] = o[q]; q--; }
                o[q + 1] = k;
            }
            int32_t *dst = order + 4 * j;
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
            if (p[o[0]] == p[o[1]] || p[o[1]] == p[o[2]] || p[o[2]] == p[o[3]]) ties[(size_t)t].push_back(j);
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    int64_t w = 0;
    for (auto &v : ties)
        for (int64_t j : v) {
            if (w < tie_cap) tie_idx[w] = j;
            w++;
        }
    *n_ties = w;
    return w > tie_cap ? NC_ERR_CAPACITY : NC_OK;
}

// BGZF (SAMv1 4.1) compression of a byte stream on all host cores: independent blocks of 0xff00 payload bytes, each a
// raw-deflate gzip member with the 'BC' extra field; the 28-byte empty EOF block is appended.  block_coff[b] = compressed
// offset of block b (block_coff[n_blocks] = offset of the EOF block), the numbers virtual file offsets are made of.
// Replaces the `| bgzip >` of snpCaller.py:284-285.
extern "C" int nc_bgzf_compress(const uint8_t *data, int64_t n, int32_t level, uint8_t *out, int64_t cap, int64_t *n_out,
                                int64_t *block_coff, int64_t blk_cap, int64_t *n_blocks)
{
    if (n < 0 || (n && !data) || !out || !n_out || !n_blocks || level < 0 || level > 9) return NC_ERR_ARG;
    constexpr int64_t BLK = 0xff00;
    const int64_t nb = (n + BLK - 1) / BLK;
    *n_blocks = nb;
    if (block_coff && blk_cap < nb + 1) return NC_ERR_CAPACITY;
    std::vector<std::vector<uint8_t>> comp((size_t)nb);
    std::vector<int> rc((size_t)nb, 0);
    auto work = [&](int64_t b) {
        const uint8_t *src = data + b * BLK;
        const uInt len = (uInt)std::min<int64_t>(BLK, n - b * BLK);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { rc[(size_t)b] = 1; return; }
        std::vector<uint8_t> &o = comp[(size_t)b];
        o.resize(18 + deflateBound(&zs, len) + 8);
        zs.next_in = const_cast<Bytef *>(src);
        zs.avail_in = len;
        zs.next_out = o.data() + 18;
        zs.avail_out = (uInt)(o.size() - 26);
        const int r = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (r != Z_STREAM_END || clen + 26 > 65536) { rc[(size_t)b] = 1; return; }
        const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(o.data(), hdr, 16);
        const uint32_t bsize = (uint32_t)(clen + 25);
        o[16] = (uint8_t)(bsize & 0xff); o[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, len);
        uint8_t *t = o.data() + 18 + clen;
        for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)len >> (8 * k)); }
        o.resize(18 + clen + 8);
    };
    int T = host_threads();
    if (nb < 8) T = 1;
    if (T == 1) for (int64_t b = 0; b < nb; b++) work(b);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() { for (int64_t b = t; b < nb; b += T) work(b); });
        for (auto &x : th) x.join();
    }
    int64_t w = 0;
    static const uint8_t EOF_BLK[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t b = 0; b < nb; b++) {
        if (rc[(size_t)b]) return NC_ERR_ARG;
        if (block_coff) block_coff[b] = w;
        if (w + (int64_t)comp[(size_t)b].size() + 28 > cap) return NC_ERR_CAPACITY;
        memcpy(out + w, comp[(size_t)b].data(), comp[(size_t)b].size());
        w += (int64_t)comp[(size_t)b].size();
    }
    if (block_coff) block_coff[nb] = w;
    if (w + 28 > cap) return NC_ERR_CAPACITY;
    memcpy(out + w, EOF_BLK, 28);
    *n_out = w + 28;
    return NC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Indel genotype rules + VCF record text (indelCaller.py:87-152; haploid :173-179), as nanocaller_amd/indelCaller.py::
// indel_vcf_lines[_haploid] state them: float32 arithmetic for QUAL / GQ (`batch_prob_all` is a float32 tensor in the reference),
// '%.2f' of the float32 value widened to double.  Sites arrive chunk-major (nc_indel_sites_fetch); `prev` restarts per chunk.
extern "C" int nc_indel_vcf_format(const char *chrom, int64_t n, const int32_t *pos, const int32_t *chunk, int32_t n_chunks, const float *probs,
                                   int32_t sets, const int32_t *ref_len, const int32_t *alt_len, const uint8_t *alt_bases, const int32_t *phase,
                                   const char *contig, int64_t chrom_len, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes,
                                   int64_t *chunk_txt_off)
{
    if (!chrom || n < 0 || !n_bytes || (n && (!pos || !chunk || !probs || !ref_len || !alt_len || !contig || !out)) || (sets != 1 && sets != 3) ||
        (haploid ? sets != 1 : sets != 3) || n_chunks < 0)
        return NC_ERR_ARG;
    static const char LET[8] = {'A', 'G', 'T', 'C', 'N', 'N', 'N', 'N'};
    const size_t lc = strlen(chrom);
    // offsets of the ALT prefixes: (site, set) order, lengths max(alt_len, 0)
    std::vector<int64_t> aoff((size_t)n * sets + 1, 0);
    for (int64_t k = 0; k < n * sets; k++) aoff[(size_t)k + 1] = aoff[(size_t)k] + (alt_len[k] > 0 ? alt_len[k] : 0);
    if (aoff[(size_t)n * sets] && !alt_bases) return NC_ERR_ARG;
    char *o = out, *const end = out + cap;
    int32_t cur_chunk = -1, prev = 0;
    int next_chunk_mark = 0;
    auto mark = [&](int upto) { if (chunk_txt_off) for (; next_chunk_mark <= upto && next_chunk_mark <= n_chunks; next_chunk_mark++) chunk_txt_off[next_chunk_mark] = o - out; };
    auto q10 = [](float x) { return -10.0f * log10f(x); };
    struct Al { const char *ref; int32_t rl; const uint8_t *alt; int32_t al; bool ok; };
    for (int64_t j = 0; j < n; j++) {
        if (chunk[j] != cur_chunk) {
            if (chunk[j] < cur_chunk || chunk[j] >= (n_chunks ? n_chunks : INT32_MAX)) return NC_ERR_ARG;      // chunk-major order
            cur_chunk = chunk[j];
            prev = 0;
            mark(cur_chunk);
        }
        const int32_t pj = pos[j];
        if (!(pj > prev)) continue;                                                 // :93
        Al a[3];
        int64_t need = 64 + (int64_t)lc;
        for (int t = 0; t < sets; t++) {
            const int32_t rl = ref_len[j * sets + t], al = alt_len[j * sets + t];
            a[t].ok = rl > 0;                                                       // `if at[0]`: None and '' are both false
            a[t].rl = rl;
            a[t].al = al > 0 ? al : 0;
            a[t].ref = contig + (pj - 1);
            a[t].alt = alt_bases ? alt_bases + aoff[(size_t)(j * sets + t)] : nullptr;
            if (a[t].ok && (pj < 1 || (int64_t)pj - 1 + rl > chrom_len)) return NC_ERR_ARG;
            if (a[t].ok) need += 2 * ((int64_t)rl + a[t].al);
        }
        if (end - o < need + 64) return NC_ERR_CAPACITY;
        auto head = [&]() { o = put_str(o, chrom, lc); *o++ = '\t'; o = put_int(o, pj); o = PUT_LIT(o, "\t.\t"); };
        auto put_alt = [&](const uint8_t *s, int32_t len) { for (int32_t i = 0; i < len; i++) *o++ = LET[s[i] & 7]; };
        if (haploid) {
            const float p0 = probs[j];
            if (!(p0 >= 0.5f) || !a[0].ok) continue;                                // :173
            const float q = -100.0f * log10f((float)(1e-6 + 1) - p0);
            head();
            o = put_str(o, a[0].ref, (size_t)a[0].rl); *o++ = '\t';
            put_alt(a[0].alt, a[0].al); *o++ = '\t';
            o = put_fixed(o, (double)q, 2);
            o = PUT_LIT(o, "\tPASS\t.\tGT:GQ\t1/1:");
            o = put_fixed(o, (double)q, 2);
            *o++ = '\n';
            prev = pj + std::max(a[0].rl, a[0].al);
            continue;
        }
        const float *pr = probs + j * 4;
        if (!(pr[0] <= 0.95f)) continue;                                            // :95
        int pred = 0;
        for (int k = 1; k < 4; k++) if (pr[k] > pr[pred]) pred = k;                 // np.argmax: first maximum ...
        for (int k = 3; k >= 0; k--) if (pr[k] != pr[k]) pred = k;                  // ... and a NaN is its maximum (the first one)
        const float q = q10(1e-6f + pr[0]);                                         // :97
        const float one = (float)(1 + 1e-6);
        const Al &a0 = a[0], &a1 = a[1], &at = a[2];
        auto tail_simple = [&](const char *gt, float gq, bool with_ps) {
            *o++ = '\t';
            o = put_fixed(o, (double)q, 2);
            o = PUT_LIT(o, "\tPASS\t.\tGT:GQ");
            if (with_ps) o = PUT_LIT(o, ":PS");
            *o++ = '\t';
            o = put_str(o, gt, 3);
            *o++ = ':';
            o = put_fixed(o, (double)gq, 2);
            if (with_ps) { *o++ = ':'; o = put_int(o, phase[j]); }
            *o++ = '\n';
        };
        const bool ps = phase && phase[j] != 0;                                     // `if phase[j]:` (None and 0 are false)
        if (pred == 1 && at.ok) {                                                   // :100
            head();
            o = put_str(o, at.ref, (size_t)at.rl); *o++ = '\t';
            put_alt(at.alt, at.al);
            tail_simple("1/1", q10(one - pr[1]), false);
            prev = pj + std::max(at.rl, at.al);
        } else if (a0.ok && a1.ok) {
            if (a0.rl == a1.rl && a0.al == a1.al && memcmp(a0.alt, a1.alt, (size_t)a0.al) == 0) {       // :109
                head();
                o = put_str(o, a0.ref, (size_t)a0.rl); *o++ = '\t';
                put_alt(a0.alt, a0.al);
                tail_simple("1/1", q10(one - pr[1]), false);
                prev = pj + std::max(a0.rl, a0.al);
            } else {                                                                // :115-133 het-alt, alleles padded to one REF
                const int32_t ln = std::min(a0.rl, a1.rl);
                const bool first_longer = a0.rl > a1.rl;
                const int32_t rl = first_longer ? a0.rl : a1.rl, pad = rl - ln;
                const int32_t l1 = a0.al + (first_longer ? 0 : pad), l2 = a1.al + (first_longer ? pad : 0);
                head();
                o = put_str(o, contig + (pj - 1), (size_t)rl); *o++ = '\t';
                put_alt(a0.alt, a0.al);
                if (!first_longer) o = put_str(o, contig + (pj - 1) + ln, (size_t)pad);
                *o++ = ',';
                put_alt(a1.alt, a1.al);
                if (first_longer) o = put_str(o, contig + (pj - 1) + ln, (size_t)pad);
                tail_simple("1|2", q10(one - pr[3]), ps);
                prev = pj + std::max(rl, std::max(l1, l2));
            }
        } else if (a0.ok || a1.ok) {                                                // :135-151
            const Al &x = a0.ok ? a0 : a1;
            head();
            o = put_str(o, x.ref, (size_t)x.rl); *o++ = '\t';
            put_alt(x.alt, x.al);
            tail_simple(a0.ok ? "0|1" : "1|0", q10(one - pr[2]), ps);
            prev = pj + std::max(x.rl, x.al);
        }
    }
    mark(n_chunks);
    *n_bytes = o - out;
    return NC_OK;
}
