// Native SNP genotype rules + VCF record text (host).  Restates snpCaller.py:113-198 exactly
// as nanocaller_amd/snpCaller.py::snp_vcf_lines does, at ~0.2 us per record instead of ~10 us in Python.
// `order` is numpy's argsort of the four probabilities (ascending) computed by the caller, so ties resolve as in
// the reference on the same machine (quirk E15).  Number formatting: printf("%.Nf") and Python's '%.Nf' are both
// correctly rounded conversions of the same double.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/nanocaller_hip.h"

namespace {
const char B[4] = {'A', 'G', 'T', 'C'};                                   // snpCaller.py:14
inline double qual(float p, double cap, double mult)
{
    const double q = mult * std::log10(1e-10 + 1 - (double)p);            // float64, numpy<2 semantics (E7)
    return q < cap ? q : cap;
}
}   // namespace

static int format_range(const char *chrom, int64_t n, const int32_t *pos, const int32_t *ref, const float *probs,
                        const int32_t *order, const int32_t *dp, const double *freq, const int32_t *fwd,
                        const int32_t *rev, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes);

// Records are independent: ranges of sites are formatted by host threads into disjoint slices of `out` (400-byte
// budget per record) and compacted in order afterwards.
extern "C" int nc_snp_vcf_format(const char *chrom, int64_t n, const int32_t *pos, const int32_t *ref, const float *probs,
                                 const int32_t *order, const int32_t *dp, const double *freq, const int32_t *fwd,
                                 const int32_t *rev, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes)
{
    if (!chrom || n < 0 || !n_bytes) return NC_ERR_ARG;
    const int64_t per = 400 + (int64_t)strlen(chrom);
    unsigned hw = std::thread::hardware_concurrency();
    int T = (int)(hw ? (hw > 32 ? 32 : hw) : 1);
    if (n < 20000 || cap < n * per) T = 1;
    if (T == 1) return format_range(chrom, n, pos, ref, probs, order, dp, freq, fwd, rev, haploid, out, cap, n_bytes);
    std::vector<int64_t> nb((size_t)T, 0);
    std::vector<int> rc((size_t)T, NC_OK);
    std::vector<std::thread> th;
    const int64_t chunk = (n + T - 1) / T;
    for (int t = 0; t < T; t++) {
        const int64_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
        if (a >= b) break;
        th.emplace_back([=, &nb, &rc]() {
            rc[(size_t)t] = format_range(chrom, b - a, pos + a, ref + a, probs + 4 * a, order ? order + 4 * a : nullptr, dp + a, freq + a,
                                         fwd ? fwd + 4 * a : nullptr, rev ? rev + 4 * a : nullptr, haploid, out + a * per, (b - a) * per,
                                         &nb[(size_t)t]);
        });
    }
    for (auto &x : th) x.join();
    int64_t w = 0;
    for (size_t t = 0; t < th.size(); t++) {
        if (rc[t] != NC_OK) return rc[t];
        const int64_t a = (int64_t)t * chunk;
        if (w != a * per) memmove(out + w, out + a * per, (size_t)nb[t]);
        w += nb[t];
    }
    *n_bytes = w;
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
        char info[96];
        snprintf(info, sizeof info, "PR=%.4f,%.4f,%.4f,%.4f;FQ=%.4f", (double)pr[0], (double)pr[3], (double)pr[1], (double)pr[2], freq[j]);  // :127
        int len = 0;
        if (haploid) {                                                                                     // :184-198
            int p = 0;
            for (int k = 1; k < 4; k++) if (pr[k] > pr[p]) p = k;                                          // np.argmax: first maximum
            len = sprintf(o, "%s\t%d\t.\t%c\t%c\t%.3f\t%s\t%s\tGT:DP:VF:AD:ADF:ADR\t1/1:%d:%.4f:.:.:.\n", chrom, pos[j], B[r], B[p],
                          qual(pr[p], 999.0, -100.0), p != r ? "PASS" : "REF", info, d, freq[j]);
        } else {
            const int32_t *f = fwd + 4 * j, *v = rev + 4 * j;
            const int p1 = order[4 * j + 3], p2 = order[4 * j + 2];
            int k = 0;
            for (int b = 0; b < 4; b++) k += pr[b] >= 0.5f;                                                // :122
            const int rf = f[r], rr = v[r];
            if (k >= 2) {
                if (p1 == r || (p2 == r && pr[p2] >= 0.5f)) {                                              // :132, :138
                    const int a = p1 == r ? p2 : p1;
                    len = sprintf(o, "%s\t%d\t.\t%c\t%c\t%.3f\tPASS\t%s\tGT:DP:VF:AD:ADF:ADR\t0/1:%d:%.4f:%d,%d:%d,%d:%d,%d\n", chrom, pos[j], B[r],
                                  B[a], qual(pr[p2], 99.0, -10.0), info, d, (double)(f[a] + v[a]) / d, rf + rr, f[a] + v[a], rf, f[a], rr, v[a]);
                } else if (p2 != r && p1 != r && pr[p2] >= 0.5f) {                                         // :143
                    len = sprintf(o, "%s\t%d\t.\t%c\t%c,%c\t%.3f\tPASS\t%s\tGT:DP:VF:AD:ADF:ADR\t1/2:%d:%.4f,%.4f:%d,%d,%d:%d,%d,%d:%d,%d,%d\n", chrom,
                                  pos[j], B[r], B[p1], B[p2], qual(pr[p2], 99.0, -10.0), info, d, (double)(f[p1] + v[p1]) / d,
                                  (double)(f[p2] + v[p2]) / d, rf + rr, f[p1] + v[p1], f[p2] + v[p2], rf, f[p1], f[p2], rr, v[p1], v[p2]);
                }
            } else if (k == 1 && r != p1 && pr[p1] >= 0.5f) {                                              // :150
                len = sprintf(o, "%s\t%d\t.\t%c\t%c\t%.3f\tPASS\t%s\tGT:DP:VF:AD:ADF:ADR\t1/1:%d:%.4f:%d,%d:%d,%d:%d,%d\n", chrom, pos[j], B[r], B[p1],
                              qual(pr[p1], 99.0, -10.0), info, d, (double)(f[p1] + v[p1]) / d, rf + rr, f[p1] + v[p1], rf, f[p1], rr, v[p1]);
            } else if (k == 1 && r == p1) {                                                                // :157
                len = sprintf(o, "%s\t%d\t.\t%c\t.\t%.3f\tREF\t%s\tGT:DP:VF:AD:ADF:ADR\t./.:%d:.:.:.:.\n", chrom, pos[j], B[r],
                              qual(pr[p1], 99.0, -10.0), info, d);
            } else {                                                                                       // :161
                len = sprintf(o, "%s\t%d\t.\t%c\t.\t0.000\tLOW\t%s\tGT:DP:VF:AD:ADF:ADR\t./.:%d:.:.:.:.\n", chrom, pos[j], B[r], info, d);
            }
        }
        w += len;
    }
    *n_bytes = w;
    return NC_OK;
}
