/*
 * nc_oracle.c -- CPU restatement of NanoCaller's SNP hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  The product (nanocaller_amd/, include/) never calls into it.
 *
 * It restates, function by function, the algorithm of the reference (paths relative to
 * /root/reference):
 *   oracle_snp_scan        generate_SNP_pileups.py:137-186  column scan -> nbr sites / candidates
 *   oracle_get_cnd_pos     generate_SNP_pileups.py:6-101    bucketed neighbour pick
 *   oracle_snp_featurize   generate_SNP_pileups.py:200-277  5x41x5 tensor, strand depths, depth
 *   oracle_snp_forward     model_architect.py:36-64 (+ snpCaller.py:90-111 scaling / inputs)
 *   oracle_snp_hap_forward model_architect_SNP_haploid.py:33-53
 *   oracle_indel_forward   model_architect_indel.py:28-48 / model_architect_indels_haploid.py:29-48
 *   oracle_indel_tensor    generate_indel_pileups.py:57-71   aligned rows -> (5,128,2)
 * Parity pin: tests/golden/ hold outputs of the reference's own featuriser (run in the build
 * container through stub pysam, oracle/tools/make_goldens.py) for the same worlds; the CNN has
 * no TensorFlow golden (TF absent) -- "CNN-vs-TF parity unpinned", see DESIGN.md.
 *
 * Input boundary = decoded alignments, read-major: read r covers reference positions
 * [start[r], end[r]) (1-based) and codes[off[r] + p - start[r]] is its base code at p
 * (A=0 G=1 T=2 C=3, deletion/N=4: generate_SNP_pileups.py:104).  keep[r]=0 drops a read
 * (pileup flag filter, generate_SNP_pileups.py:151-157).  strand[r] = bit 0x10 of the record's flag (:143 on a primary record).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { MODE_ONT = 0, MODE_SHORT_ONT = 1, MODE_UL_ONT = 2, MODE_UL_ONT_EXTREME = 3, MODE_PACBIO = 4 };
#define NBR_SIZE 20
#define FLANK 50000
#define BIG 1000000000

typedef struct {
    int32_t n_reads;
    const int32_t *start, *end;
    const int64_t *off;
    const uint8_t *codes;
    const uint8_t *strand;
    const uint8_t *keep;
} reads_t;

/* ---------------------------------------------------------------- neighbour selection */
typedef struct { int lo, hi, k, first; } bucket_t;   /* left: p>=v-lo && p<v-hi ; right: p>v+lo && p<=v+hi */

static int mode_table(int mode, int *W, bucket_t *Lb, bucket_t *Rb)
{
    /* each row restates one list comprehension of get_cnd_pos; `first`=1 means [:k], 0 means [-k:] */
    switch (mode) {
    case MODE_ONT: {                                   /* generate_SNP_pileups.py:7-23 */
        *W = 50000;
        bucket_t l[] = {{2000, 0, 2, 1}, {5000, 2000, 3, 0}, {10000, 5000, 4, 0}, {20000, 10000, 5, 0}, {BIG, 20000, 6, 0}};
        bucket_t r[] = {{0, 2000, 2, 0}, {2000, 5000, 3, 1}, {5000, 10000, 4, 1}, {10000, 20000, 5, 1}, {20000, BIG, 6, 1}};
        memcpy(Lb, l, sizeof l); memcpy(Rb, r, sizeof r); return 5; }
    case MODE_SHORT_ONT: {                             /* :25-39 */
        *W = 50000;
        bucket_t l[] = {{2000, 0, 5, 0}, {5000, 2000, 10, 0}, {BIG, 5000, 5, 0}};
        bucket_t r[] = {{0, 2000, 5, 1}, {2000, 5000, 10, 1}, {5000, BIG, 5, 1}};
        memcpy(Lb, l, sizeof l); memcpy(Rb, r, sizeof r); return 3; }
    case MODE_UL_ONT: {                                /* :41-61 */
        *W = 100000;
        bucket_t l[] = {{2000, 0, 2, 1}, {5000, 2000, 2, 0}, {10000, 5000, 3, 0}, {20000, 10000, 3, 0},
                        {40000, 20000, 4, 0}, {50000, 40000, 3, 0}, {BIG, 50000, 3, 0}};
        bucket_t r[] = {{0, 2000, 2, 0}, {2000, 5000, 2, 1}, {5000, 10000, 3, 1}, {10000, 20000, 3, 1},
                        {20000, 40000, 4, 1}, {40000, 50000, 3, 1}, {50000, BIG, 3, 1}};
        memcpy(Lb, l, sizeof l); memcpy(Rb, r, sizeof r); return 7; }
    case MODE_UL_ONT_EXTREME: {                        /* :63-83 */
        *W = 300000;
        bucket_t l[] = {{10000, 0, 2, 1}, {20000, 10000, 2, 0}, {50000, 20000, 3, 0}, {75000, 50000, 3, 0},
                        {100000, 75000, 4, 0}, {200000, 100000, 4, 0}, {BIG, 200000, 2, 0}};
        bucket_t r[] = {{0, 10000, 2, 0}, {10000, 20000, 2, 1}, {20000, 50000, 3, 1}, {50000, 75000, 3, 1},
                        {75000, 100000, 4, 1}, {100000, 200000, 4, 1}, {200000, BIG, 2, 1}};
        memcpy(Lb, l, sizeof l); memcpy(Rb, r, sizeof r); return 7; }
    case MODE_PACBIO: {                                /* :85-99 */
        *W = 20000;
        bucket_t l[] = {{2000, 0, 4, 1}, {5000, 2000, 5, 0}, {10000, 5000, 5, 0}, {20000, 10000, 6, 0}};
        bucket_t r[] = {{0, 2000, 4, 0}, {2000, 5000, 5, 1}, {5000, 10000, 5, 1}, {10000, 20000, 6, 1}};
        memcpy(Lb, l, sizeof l); memcpy(Rb, r, sizeof r); return 4; }
    }
    return -1;
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* nbr ascending (positions appended in column order, :174).  Outputs ascending lists. */
int oracle_get_cnd_pos(int32_t v, const int32_t *nbr, int32_t n_nbr, int mode,
                       int32_t *left, int32_t *n_left, int32_t *right, int32_t *n_right)
{
    int W;
    bucket_t Lb[8], Rb[8];
    int nb = mode_table(mode, &W, Lb, Rb);
    if (nb < 0) return -1;
    int nl = 0, nr = 0;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_nbr > 0 ? n_nbr : 1));
    for (int b = 0; b < nb; b++) {
        for (int side = 0; side < 2; side++) {
            bucket_t B = side ? Rb[b] : Lb[b];
            int m = 0;
            for (int i = 0; i < n_nbr; i++) {
                int64_t p = nbr[i];
                int64_t d = p - v;
                if (!((d < 0 ? -d : d) < W)) continue;                    /* ls = cnd_pos[abs(cnd_pos-v_pos)<W] */
                int in;
                if (side == 0) in = (B.lo >= BIG || p >= (int64_t)v - B.lo) && p < (int64_t)v - B.hi;
                else           in = p > (int64_t)v + B.lo && (B.hi >= BIG || p <= (int64_t)v + B.hi);
                if (in) tmp[m++] = (int32_t)p;
            }
            int take = m < B.k ? m : B.k;
            const int32_t *src = B.first ? tmp : tmp + (m - take);
            for (int i = 0; i < take; i++) {
                if (side == 0) left[nl++] = src[i]; else right[nr++] = src[i];
            }
        }
    }
    free(tmp);
    qsort(left, (size_t)nl, sizeof(int32_t), cmp_i32);     /* sorted(ls1_0+ls1_1+...) */
    qsort(right, (size_t)nr, sizeof(int32_t), cmp_i32);
    *n_left = nl; *n_right = nr;
    return 0;
}

/* ---------------------------------------------------------------- column scan */
static inline int read_code(const reads_t *R, int r, int32_t p)
{
    if (p < R->start[r] || p >= R->end[r]) return -1;      /* read not in that column's pileup */
    return R->codes[R->off[r] + (p - R->start[r])];
}

/*
 * Scan columns [max(1,start-50000), min(L,end+50000)] (:137,156).  ref_code[p-1] in 0..3, or 4 when the
 * column is skipped (non-AGTC / soft-masked reference base, or exclude_bed hit; :137,161).
 * Outputs: nbr (ascending), candidates (pos, n, alt count).  Returns 0 / -2 on capacity overflow.
 */
int oracle_snp_scan(int32_t n_reads, const int32_t *rstart, const int32_t *rend, const int64_t *roff,
                    const uint8_t *codes, const uint8_t *keep,
                    const uint8_t *ref_code, int32_t L, int32_t start, int32_t end, int haploid,
                    int32_t mincov, double min_allele_freq, double t0, double t1,
                    int32_t cap, int32_t *nbr_pos, int32_t *n_nbr,
                    int32_t *cand_pos, int32_t *cand_n, int32_t *cand_alt, int32_t *n_cand)
{
    int64_t lo = (int64_t)start - FLANK; if (lo < 1) lo = 1;
    int64_t hi = (int64_t)end + FLANK;   if (hi > L) hi = L;
    int64_t ncol = hi - lo + 1;
    *n_nbr = 0; *n_cand = 0;
    if (ncol <= 0) return 0;
    /* per-column base counts by difference-free direct accumulation over reads */
    int32_t *cnt = (int32_t *)calloc((size_t)ncol * 5, sizeof(int32_t));
    if (!cnt) return -3;
    for (int r = 0; r < n_reads; r++) {
        if (keep && !keep[r]) continue;
        int64_t a = rstart[r] > lo ? rstart[r] : lo;
        int64_t b = (int64_t)rend[r] - 1 < hi ? (int64_t)rend[r] - 1 : hi;
        const uint8_t *c = codes + roff[r] - rstart[r];
        for (int64_t p = a; p <= b; p++) cnt[(p - lo) * 5 + c[p]]++;
    }
    int rc = 0;
    for (int64_t p = lo; p <= hi; p++) {
        int r = ref_code[p - 1];
        if (r > 3) continue;                                             /* `r in 'AGTC'` (:161) */
        const int32_t *c = cnt + (p - lo) * 5;
        int32_t n = c[0] + c[1] + c[2] + c[3] + c[4];                    /* get_num_aligned (:164) */
        if (n == 0) continue;                                            /* zero-depth columns are not yielded */
        int32_t alt = 0;
        for (int b = 0; b < 4; b++) if (b != r && c[b] > alt) alt = c[b];
        double alt_freq = (double)alt / (double)n;                       /* :166 */
        if (n < mincov) continue;                                        /* :170 */
        int is_nbr = haploid ? (t0 <= alt_freq) : (t0 <= alt_freq && alt_freq < t1);   /* :173,177 */
        if (is_nbr) {
            if (*n_nbr >= cap) { rc = -2; break; }
            nbr_pos[(*n_nbr)++] = (int32_t)p;
        }
        if (p >= start && p <= end && min_allele_freq <= alt_freq) {     /* :183 */
            if (*n_cand >= cap) { rc = -2; break; }
            cand_pos[*n_cand] = (int32_t)p; cand_n[*n_cand] = n; cand_alt[*n_cand] = alt;
            (*n_cand)++;
        }
    }
    free(cnt);
    return rc;
}

/*
 * Per-candidate tensor (Appendix A of SURVEY.md; :200-277).  `mat` is [n_cand][5][41][5] float32,
 * fwd/rev [n_cand][4], depth_each[n_cand] = |S| after the maxcov cut.  Returns number of sites
 * written (== n_cand unless min_nbr_sites filters), <0 on error.
 * Above maxcov the reference draws an UNSEEDED random.sample (:215-216); this build's documented
 * policy is "first maxcov reads in input order" (parity is only defined for depth <= maxcov).
 */
/* Alignments that share a read name: the reference's per-column pileup dict is keyed by NAME (:175,185), so where several alignments of one name
 * cover a column the LAST in file order is the column's entry for that name, and a site's row for a name reads every column from whichever of
 * the name's alignments covers it (:223,232).  mate_next[r] = the next alignment of r's name among the kept ones, circular (-1: the name is
 * r's alone).  named_code: the code of r's NAME at p (-1: no alignment of the name in that column's pileup); *who = the alignment that gives it. */
static inline int named_code(const reads_t *R, const int32_t *mate_next, int r, int32_t p, int *who)
{
    int best = (p >= R->start[r] && p < R->end[r]) ? r : -1;
    if (mate_next && mate_next[r] >= 0)
        for (int m = mate_next[r]; m != r; m = mate_next[m])
            if (p >= R->start[m] && p < R->end[m] && m > best) best = m;
    if (who) *who = best;
    return best < 0 ? -1 : R->codes[R->off[best] + (p - R->start[best])];
}

int oracle_snp_featurize(int32_t n_reads, const int32_t *rstart, const int32_t *rend, const int64_t *roff,
                         const uint8_t *codes, const uint8_t *strand, const uint8_t *keep,
                         const uint8_t *ref_code,
                         const int32_t *nbr_pos, int32_t n_nbr,
                         const int32_t *cand_pos, int32_t n_cand,
                         int mode, int32_t maxcov, int32_t min_nbr_sites,
                         int32_t *out_pos, int32_t *out_ref, float *mat, int32_t *fwd, int32_t *rev,
                         int32_t *depth_each, const int32_t *mate_next)
{
    reads_t R = {n_reads, rstart, rend, roff, codes, strand, keep};
    int32_t *S = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_reads > 0 ? n_reads : 1));
    int nout = 0;
    /* reads sorted or not: brute-force stabbing with a moving lower bound would assume order; keep it
       order-free but cheap: cand_pos ascending, so pre-filter by a sliding index when starts ascend */
    int sorted = 1;
    for (int r = 1; r < n_reads; r++) if (rstart[r] < rstart[r - 1]) { sorted = 0; break; }
    int first_live = 0;
    int32_t *pmax = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_reads > 0 ? n_reads : 1));   /* prefix max of ends */
    for (int r = 0; r < n_reads; r++) pmax[r] = (r && pmax[r - 1] > rend[r]) ? pmax[r - 1] : rend[r];
    for (int s = 0; s < n_cand; s++) {
        int32_t v = cand_pos[s];
        int32_t Lft[NBR_SIZE + 20], Rgt[NBR_SIZE + 20];
        int32_t nl, nr;
        if (oracle_get_cnd_pos(v, nbr_pos, n_nbr, mode, Lft, &nl, Rgt, &nr)) { free(S); free(pmax); return -1; }
        /* reads in the pileup at v (:208) */
        int ns = 0;
        int32_t f[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0};
        if (sorted) while (first_live < n_reads && pmax[first_live] <= v) first_live++;   /* reads 0..first_live-1 all end before v */
        for (int r = sorted ? first_live : 0; r < n_reads; r++) {
            if (sorted && rstart[r] > v) break;
            if (keep && !keep[r]) continue;
            int c = read_code(&R, r, v);
            if (c < 0) continue;
            if (mate_next && mate_next[r] >= 0) {                        /* a later alignment of the same name in this column replaces r (:175,185) */
                int who;
                named_code(&R, mate_next, r, v, &who);
                if (who != r) continue;
            }
            S[ns++] = r;
            if (c < 4) { if (strand[r]) rv[c]++; else f[c]++; }          /* :210-213, all reads */
        }
        if (ns > maxcov) ns = maxcov;                                    /* policy, see header */
        int ncols = nl + 1 + nr;
        if (ncols < min_nbr_sites) continue;                             /* :244 (list includes the candidate) */
        int32_t cols[2 * NBR_SIZE + 1];
        for (int j = 0; j < nl; j++) cols[j] = Lft[j];
        cols[nl] = v;
        for (int j = 0; j < nr; j++) cols[nl + 1 + j] = Rgt[j];
        int32_t cnt[4][2 * NBR_SIZE + 1][4];
        memset(cnt, 0, sizeof cnt);
        for (int i = 0; i < ns; i++) {
            int r = S[i];
            int c = read_code(&R, r, v);
            if (c > 3) continue;                                         /* centre code 4 contributes nowhere (:247) */
            for (int j = 0; j < ncols; j++) {
                int b = named_code(&R, mate_next, r, cols[j], NULL);      /* pileup_dict[nb_pos][name] (:223,232) */
                if (b >= 0 && b < 4) cnt[c][j][b]++;
            }
        }
        float *X = mat + (size_t)nout * 5 * 41 * 5;
        memset(X, 0, sizeof(float) * 5 * 41 * 5);
        int o = NBR_SIZE - nl;
        int rc_centre = ref_code[v - 1];
        for (int j = 0; j < ncols; j++) {
            int rcj = ref_code[cols[j] - 1];
            if (rcj < 4) X[(0 * 41 + o + j) * 5 + rcj] = 1.0f;           /* total_ref, channel 4 zeroed (:249-251) */
            for (int i = 0; i < 4; i++) {
                for (int b = 0; b < 4; b++) {
                    int32_t val = cnt[i][j][b];
                    X[((1 + i) * 41 + o + j) * 5 + b] = (float)(b == rcj ? -val : val);   /* :253 */
                }
                X[((1 + i) * 41 + o + j) * 5 + 4] = (i == rc_centre) ? 1.0f : 0.0f;       /* :252 */
            }
        }
        out_pos[nout] = v;
        out_ref[nout] = rc_centre;
        for (int b = 0; b < 4; b++) { fwd[nout * 4 + b] = f[b]; rev[nout * 4 + b] = rv[b]; }
        depth_each[nout] = ns;
        nout++;
    }
    free(S);
    free(pmax);
    return nout;
}

/* ---------------------------------------------------------------- CNN (Appendix C) */
#define SELU_L 1.0507009873554805
#define SELU_A 1.6732632423543772

#define DEF_CNN(T, SUF, EXP)                                                                              \
static inline T selu_##SUF(T x) { return x > 0 ? (T)SELU_L * x : (T)(SELU_L * SELU_A) * (EXP(x) - (T)1); } \
/* NHWC conv, HWIO kernel, cross-correlation; same => pad floor(k/2) (stride 1), valid otherwise */       \
static void conv_##SUF(const T *in, int H, int W, int Ci, const float *k, const float *b, int kh, int kw, \
                       int Co, int sh, int sw, int same, T *out, int Ho, int Wo, int co_off, int Ctot)    \
{                                                                                                         \
    int ph = same ? kh / 2 : 0, pw = same ? kw / 2 : 0;                                                   \
    for (int y = 0; y < Ho; y++) for (int x = 0; x < Wo; x++) for (int o = 0; o < Co; o++) {              \
        T acc = (T)b[o];                                                                                  \
        for (int dy = 0; dy < kh; dy++) { int iy = y * sh + dy - ph; if (iy < 0 || iy >= H) continue;     \
            for (int dx = 0; dx < kw; dx++) { int ix = x * sw + dx - pw; if (ix < 0 || ix >= W) continue; \
                const T *ip = in + ((size_t)iy * W + ix) * Ci;                                            \
                const float *kp = k + ((size_t)(dy * kw + dx) * Ci) * Co + o;                             \
                for (int c = 0; c < Ci; c++) acc += ip[c] * (T)kp[(size_t)c * Co]; } }                    \
        out[((size_t)y * Wo + x) * Ctot + co_off + o] = selu_##SUF(acc);                                  \
    }                                                                                                     \
}                                                                                                         \
static void dense_##SUF(const T *in, int n_in, const float *k, const float *b, int n_out, T *out, int act) \
{                                                                                                         \
    for (int o = 0; o < n_out; o++) { T acc = (T)b[o];                                                    \
        for (int i = 0; i < n_in; i++) acc += in[i] * (T)k[(size_t)i * n_out + o];                        \
        out[o] = act ? selu_##SUF(acc) : acc; }                                                           \
}                                                                                                         \
static void softmax_##SUF(T *x, int n)                                                                    \
{                                                                                                         \
    T m = x[0]; for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];                                       \
    T s = 0; for (int i = 0; i < n; i++) { x[i] = EXP(x[i] - m); s += x[i]; }                             \
    for (int i = 0; i < n; i++) x[i] /= s;                                                                \
}

DEF_CNN(float, f, expf)
DEF_CNN(double, d, exp)

/* canonical flat weight blob offsets (nanocaller_amd/weights.py LAYER_SPECS order: kernel then bias) */
typedef struct { const float *k, *b; } lw_t;
static const float *take(const float **p, size_t n) { const float *r = *p; *p += n; return r; }

#define DEF_TRUNK(T, SUF)                                                                                 \
/* x [H][W][Ci] already scaled; returns fc1 activations [F] */                                            \
static void trunk_##SUF(const T *x, int H, int W, int Ci, int C1, int C2, int C3, int F,                  \
                        const float **wp, T *fc1_out, T *scratch)                                         \
{                                                                                                         \
    lw_t c11, c12, c13, c2, c3, f1;                                                                       \
    c11.k = take(wp, (size_t)1 * 5 * Ci * C1); c11.b = take(wp, C1);                                      \
    c12.k = take(wp, (size_t)5 * 1 * Ci * C1); c12.b = take(wp, C1);                                      \
    c13.k = take(wp, (size_t)5 * 5 * Ci * C1); c13.b = take(wp, C1);                                      \
    c2.k = take(wp, (size_t)2 * 3 * 3 * C1 * C2); c2.b = take(wp, C2);                                    \
    c3.k = take(wp, (size_t)2 * 3 * C2 * C3); c3.b = take(wp, C3);                                        \
    int H2 = H - 1, W2 = (W - 3) / 2 + 1, H3 = H2 - 1, W3 = (W2 - 3) / 2 + 1;                              \
    size_t nflat = (size_t)H3 * W3 * C3;                                                                  \
    f1.k = take(wp, nflat * F); f1.b = take(wp, F);                                                       \
    T *a1 = scratch, *a2 = a1 + (size_t)H * W * 3 * C1, *a3 = a2 + (size_t)H2 * W2 * C2;                  \
    conv_##SUF(x, H, W, Ci, c11.k, c11.b, 1, 5, C1, 1, 1, 1, a1, H, W, 0, 3 * C1);                        \
    conv_##SUF(x, H, W, Ci, c12.k, c12.b, 5, 1, C1, 1, 1, 1, a1, H, W, C1, 3 * C1);                       \
    conv_##SUF(x, H, W, Ci, c13.k, c13.b, 5, 5, C1, 1, 1, 1, a1, H, W, 2 * C1, 3 * C1);                   \
    conv_##SUF(a1, H, W, 3 * C1, c2.k, c2.b, 2, 3, C2, 1, 2, 0, a2, H2, W2, 0, C2);                       \
    conv_##SUF(a2, H2, W2, C2, c3.k, c3.b, 2, 3, C3, 1, 2, 0, a3, H3, W3, 0, C3);                         \
    dense_##SUF(a3, (int)nflat, f1.k, f1.b, F, fc1_out, 1);                                               \
}

DEF_TRUNK(float, f)
DEF_TRUNK(double, d)

/*
 * Coverage scaling (snpCaller.py:93-96): rows 1..4, channels 0..3 are multiplied by `scale`.
 * scale_mode 0: numpy<2 value-based casting -- the float64 scalar is rounded to f32, product in f32.
 * scale_mode 1: --disable_coverage_normalization path, array*array -> product in f64, rounded to f32.
 */
static inline float scaled(float x, double s, int mode, int row, int ch)
{
    if (row == 0 || ch == 4) return x;
    if (mode == 0) return x * (float)s;
    return (float)((double)x * s);
}

#define DEF_SNP(T, SUF)                                                                                   \
int oracle_snp_forward_##SUF(const float *w, int64_t n, const float *x, const int32_t *ref_code,          \
                             const double *scale, int scale_mode, float *probs, float *gt)                \
{                                                                                                         \
    T *scratch = (T *)malloc(sizeof(T) * (5 * 41 * 48 + 4 * 20 * 32 + 3 * 9 * 64 + 1025));                \
    if (!scratch) return -3;                                                                              \
    T *xin = scratch + (5 * 41 * 48 + 4 * 20 * 32 + 3 * 9 * 64);                                          \
    for (int64_t s = 0; s < n; s++) {                                                                     \
        const float *xs = x + s * 1025;                                                                   \
        for (int h = 0; h < 5; h++) for (int c = 0; c < 41; c++) for (int ch = 0; ch < 5; ch++)           \
            xin[(h * 41 + c) * 5 + ch] = (T)scaled(xs[(h * 41 + c) * 5 + ch], scale[s], scale_mode, h, ch); \
        const float *wp = w;                                                                              \
        T fc1[48], fa[16], fc2[16], in17[17], heads[4][2], in24[24], fc3[8], g[2];                        \
        trunk_##SUF(xin, 5, 41, 5, 16, 32, 64, 48, &wp, fc1, scratch);                                    \
        const float *fa_k = take(&wp, 48 * 16), *fa_b = take(&wp, 16);                                    \
        dense_##SUF(fc1, 48, fa_k, fa_b, 16, fa, 1);                                                      \
        for (int i = 0; i < 16; i++) in17[i] = fa[i];                                                     \
        for (int hd = 0; hd < 4; hd++) {                           /* A,G,T,C heads (model_architect.py:55-58) */ \
            const float *hk = take(&wp, 17 * 2), *hb = take(&wp, 2);                                      \
            in17[16] = (ref_code[s] == hd) ? (T)1 : (T)0;                                                 \
            dense_##SUF(in17, 17, hk, hb, 2, heads[hd], 0);                                               \
            softmax_##SUF(heads[hd], 2);                                                                  \
            probs[s * 4 + hd] = (float)heads[hd][1];                                                      \
        }                                                                                                 \
        const float *f2k = take(&wp, 48 * 16), *f2b = take(&wp, 16);                                      \
        dense_##SUF(fc1, 48, f2k, f2b, 16, fc2, 1);                                                       \
        for (int i = 0; i < 16; i++) in24[i] = fc2[i];                                                    \
        for (int hd = 0; hd < 4; hd++) { in24[16 + 2 * hd] = heads[hd][0]; in24[17 + 2 * hd] = heads[hd][1]; } \
        const float *f3k = take(&wp, 24 * 8), *f3b = take(&wp, 8);                                        \
        dense_##SUF(in24, 24, f3k, f3b, 8, fc3, 1);                                                       \
        const float *gk = take(&wp, 8 * 2), *gb = take(&wp, 2);                                           \
        dense_##SUF(fc3, 8, gk, gb, 2, g, 0);                                                             \
        softmax_##SUF(g, 2);                                                                              \
        if (gt) { gt[s * 2] = (float)g[0]; gt[s * 2 + 1] = (float)g[1]; }                                 \
    }                                                                                                     \
    free(scratch);                                                                                        \
    return 0;                                                                                             \
}                                                                                                         \
int oracle_snp_hap_forward_##SUF(const float *w, int64_t n, const float *x, const int32_t *ref_code,      \
                                 const double *scale, int scale_mode, float *probs)                       \
{                                                                                                         \
    T *scratch = (T *)malloc(sizeof(T) * (5 * 41 * 48 + 4 * 20 * 32 + 3 * 9 * 64 + 1025));                \
    if (!scratch) return -3;                                                                              \
    T *xin = scratch + (5 * 41 * 48 + 4 * 20 * 32 + 3 * 9 * 64);                                          \
    for (int64_t s = 0; s < n; s++) {                                                                     \
        const float *xs = x + s * 1025;                                                                   \
        for (int h = 0; h < 5; h++) for (int c = 0; c < 41; c++) for (int ch = 0; ch < 5; ch++)           \
            xin[(h * 41 + c) * 5 + ch] = (T)scaled(xs[(h * 41 + c) * 5 + ch], scale[s], scale_mode, h, ch); \
        const float *wp = w;                                                                              \
        T fc1[48], fc2[16], in20[20], out[4];                                                             \
        trunk_##SUF(xin, 5, 41, 5, 16, 32, 64, 48, &wp, fc1, scratch);                                    \
        const float *f2k = take(&wp, 48 * 16), *f2b = take(&wp, 16);                                      \
        dense_##SUF(fc1, 48, f2k, f2b, 16, fc2, 1);                                                       \
        for (int i = 0; i < 16; i++) in20[i] = fc2[i];                                                    \
        for (int b = 0; b < 4; b++) in20[16 + b] = (ref_code[s] == b) ? (T)1 : (T)0;                      \
        const float *f3k = take(&wp, 20 * 4), *f3b = take(&wp, 4);                                        \
        dense_##SUF(in20, 20, f3k, f3b, 4, out, 1);                /* Dense(4, activation=selu) (:29,50) */ \
        softmax_##SUF(out, 4);                                                                            \
        for (int b = 0; b < 4; b++) probs[s * 4 + b] = (float)out[b];                                     \
    }                                                                                                     \
    free(scratch);                                                                                        \
    return 0;                                                                                             \
}                                                                                                         \
/* rows = 15 (diploid, 4-way softmax) or 5 (haploid, sigmoid); x [n][rows][128][2] */                     \
int oracle_indel_forward_##SUF(const float *w, int64_t n, int rows, const float *x, float *probs)         \
{                                                                                                         \
    int H = rows, W = 128;                                                                                \
    size_t na = (size_t)H * W * 24 + (size_t)(H - 1) * 63 * 32 + (size_t)(H - 2) * 31 * 48 + (size_t)H * W * 2; \
    T *scratch = (T *)malloc(sizeof(T) * na);                                                             \
    if (!scratch) return -3;                                                                              \
    T *xin = scratch + (na - (size_t)H * W * 2);                                                          \
    int nout = rows == 15 ? 4 : 1;                                                                        \
    for (int64_t s = 0; s < n; s++) {                                                                     \
        for (int i = 0; i < H * W * 2; i++) xin[i] = (T)x[s * H * W * 2 + i];                             \
        const float *wp = w;                                                                              \
        T fc1[32], fc2[24], out[4];                                                                       \
        trunk_##SUF(xin, H, W, 2, 8, 32, 48, 32, &wp, fc1, scratch);                                      \
        const float *f2k = take(&wp, 32 * 24), *f2b = take(&wp, 24);                                      \
        dense_##SUF(fc1, 32, f2k, f2b, 24, fc2, 1);                                                       \
        const float *f3k = take(&wp, 24 * nout), *f3b = take(&wp, nout);                                  \
        dense_##SUF(fc2, 24, f3k, f3b, nout, out, 0);                                                     \
        if (nout == 4) { softmax_##SUF(out, 4); for (int b = 0; b < 4; b++) probs[s * 4 + b] = (float)out[b]; } \
        else probs[s] = (float)((T)1 / ((T)1 + (sizeof(T) == 4 ? (T)expf((float)-out[0]) : (T)exp((double)-out[0])))); \
    }                                                                                                     \
    free(scratch);                                                                                        \
    return 0;                                                                                             \
}

DEF_SNP(float, f)
DEF_SNP(double, d)

/* ---------------------------------------------------------------- indel: aligned rows -> (5,128,2) */
/*
 * generate_indel_pileups.py:57-71.  rows [nrows][ncols] symbols 0..4 (A,G,T,C,'-'), ref_row [ncols].
 * out [5][128][2] float32: channel 0 = column frequency minus ref one-hot, channel 1 = ref one-hot.
 * cns: argmax symbol per column with the gap handicapped by 0.01 (:61-64), 255-terminated, gaps dropped.
 */
int oracle_indel_tensor(const uint8_t *rows, int32_t nrows, int32_t ncols, const uint8_t *ref_row,
                        float *out, uint8_t *cns, int32_t *cns_len)
{
    memset(out, 0, sizeof(float) * 5 * 128 * 2);
    int nc = 0;
    for (int c = 0; c < ncols; c++) {
        float h[5] = {0, 0, 0, 0, 0};
        for (int r = 0; r < nrows; r++) h[rows[(size_t)r * ncols + c]] += 1.0f;
        float tot = h[0] + h[1] + h[2] + h[3] + h[4];
        float alt[5], best = -1e30f; int arg = 0;
        for (int s = 0; s < 5; s++) {
            alt[s] = h[s] / tot;                                         /* f32 divide (:58-59) */
            float t = s == 4 ? alt[s] - 0.01f : alt[s];                  /* tmp_mat[:,4]-=0.01 (f32 array) */
            if (t > best) { best = t; arg = s; }                         /* np.argmax: first maximum */
        }
        if (arg != 4) cns[nc++] = (uint8_t)arg;
        if (c < 128)
            for (int s = 0; s < 5; s++) {
                float rf = ref_row[c] == s ? 1.0f : 0.0f;
                out[(s * 128 + c) * 2 + 0] = alt[s] - rf;                /* alt_mat -= ref (:67) */
                out[(s * 128 + c) * 2 + 1] = rf;
            }
    }
    *cns_len = nc;
    return 0;
}

/* ---------------------------------------------------------------- indel candidate window scan (pass 1) */
/*
 * generate_indel_pileups.py:197-304 (impute_indel_phase branch excluded).  Reads as above plus hap[r] (0 untagged,
 * 1/2 = HP tag, :178-188) and read-major indel events (ev_off[r]..ev_off[r+1]: ev_pos = column carrying the
 * '+n'/'-n' marker, ev_len signed: + insertion, - deletion).  excl[p-1] != 0 skips the column entirely (ex_bed, :217).
 * Sliding windows are over the last `win` / `small_win` YIELDED columns (deque maxlen, :197-204), a column is yielded
 * when at least one kept read covers it.  Output: variants (anchor position, type 0/1) in ascending anchor order;
 * a later detection overwrites an equal anchor (dict semantics, :268,274).
 */
int oracle_indel_scan(int32_t n_reads, const int32_t *rstart, const int32_t *rend, const uint8_t *keep, const uint8_t *hap,
                      const int32_t *ev_off, const int32_t *ev_pos, const int32_t *ev_len, const uint8_t *excl, int32_t L,
                      int32_t start, int32_t end, int32_t mincov, int32_t win, int32_t small_win, double ins_t, double del_t,
                      int32_t haploid, int32_t cap, int32_t *var_pos, int32_t *var_type, int32_t *n_var)
{
    int32_t lo = start < 1 ? 1 : start, hi = end > L ? L : end;
    *n_var = 0;
    if (hi < lo) return 0;
    int64_t ncol = (int64_t)hi - lo + 1;
    /* per-column depths by haplotype */
    int32_t *d = (int32_t *)calloc((size_t)(ncol + 1) * 3, sizeof(int32_t));
    /* per-column event lists: counting sort of the qualifying events */
    int32_t *ecnt = (int32_t *)calloc((size_t)ncol + 1, sizeof(int32_t));
    if (!d || !ecnt) return -3;
    int64_t n_ev = 0;
    for (int r = 0; r < n_reads; r++) {
        if (keep && !keep[r]) continue;
        int64_t a = rstart[r] > lo ? rstart[r] : lo, b = (int64_t)rend[r] - 1 < hi ? (int64_t)rend[r] - 1 : hi;
        if (a > b) continue;
        /* haploid (generate_indel_pileups_haploid.py:185-241): one read set, HP tags are not looked at */
        int h = haploid ? 0 : hap[r] == 1 ? 0 : hap[r] == 2 ? 1 : 2;
        d[(a - lo) * 3 + h]++;
        d[(b - lo + 1) * 3 + h]--;
        if (h < 2)
            for (int e = ev_off[r]; e < ev_off[r + 1]; e++)
                if (ev_pos[e] >= lo && ev_pos[e] <= hi) { ecnt[ev_pos[e] - lo + 1]++; n_ev++; }
    }
    for (int64_t c = 1; c <= ncol; c++) {
        for (int h = 0; h < 3; h++) d[c * 3 + h] += d[(c - 1) * 3 + h];
        ecnt[c] += ecnt[c - 1];
    }
    int32_t *eread = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ev + 1)), *elen = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ev + 1));
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ncol + 1));
    memcpy(fill, ecnt, sizeof(int32_t) * (size_t)(ncol + 1));
    for (int r = 0; r < n_reads; r++) {
        if ((keep && !keep[r]) || (!haploid && hap[r] != 1 && hap[r] != 2)) continue;
        for (int e = ev_off[r]; e < ev_off[r + 1]; e++)
            if (ev_pos[e] >= lo && ev_pos[e] <= hi && ev_pos[e] >= rstart[r] && ev_pos[e] < rend[r]) {
                int32_t k = fill[ev_pos[e] - lo]++;
                eread[k] = r; elen[k] = ev_len[e];
            } else if (ev_pos[e] >= lo && ev_pos[e] <= hi) {
                int32_t k = fill[ev_pos[e] - lo]++;      /* keep the counting sort consistent; mark unusable */
                eread[k] = -1; elen[k] = 0;
            }
    }
    /* window state: classes 0 del long, 1 ins long, 2 del small, 3 ins small; per class per hap: per-read multiplicity */
    int32_t *mult = (int32_t *)calloc((size_t)(n_reads > 0 ? n_reads : 1) * 4, sizeof(int32_t));
    int32_t distinct[4][2];
    memset(distinct, 0, sizeof distinct);
    int32_t *ycol = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncol);      /* yielded columns, in order */
    int64_t ny = 0;
    int64_t prev = 0;
    int rc = 0;
    for (int64_t v = lo; v <= hi; v++) {
        const int64_t c = v - lo;
        const int32_t n0 = d[c * 3 + 0], n1 = d[c * 3 + 1], ntot = n0 + n1 + d[c * 3 + 2];
        if (ntot == 0) continue;                                  /* not yielded */
        if (excl && excl[v - 1]) continue;                        /* :217 */
        /* append this column's sets; drop the column that leaves each deque */
        ycol[ny] = (int32_t)v;
        for (int pass = 0; pass < 2; pass++) {                    /* pass 0: leaving columns, pass 1: entering column */
            for (int cls = 0; cls < 4; cls++) {
                const int w = cls < 2 ? win : small_win;
                int64_t col;
                if (pass == 0) { if (ny - w < 0) continue; col = ycol[ny - w]; } else col = v;
                for (int32_t k = ecnt[col - lo]; k < ecnt[col - lo + 1]; k++) {
                    const int r = eread[k];
                    if (r < 0) continue;
                    const int32_t ln = elen[k] < 0 ? -elen[k] : elen[k];
                    const int is_ins = elen[k] > 0;
                    int q;
                    if (cls < 2) q = (ln > 2 && ln <= 50) && (is_ins == (cls == 1));      /* :225,228 */
                    else q = (ln <= 10) && (is_ins == (cls == 3));                         /* :226,229 */
                    if (!q) continue;
                    const int h = haploid ? 0 : hap[r] - 1;
                    int32_t *m = &mult[(size_t)r * 4 + cls];
                    if (pass == 1) { if ((*m)++ == 0) distinct[cls][h]++; }
                    else { if (--(*m) == 0) distinct[cls][h]--; }
                }
            }
        }
        ny++;
        if (v <= prev) continue;                                  /* :249 */
        if (haploid) {
            if (ntot >= mincov) {                                  /* haploid :224 */
                const double del = (double)distinct[0][0] / ntot, ins = (double)distinct[1][0] / ntot;
                const double dels = (double)distinct[2][0] / ntot, inss = (double)distinct[3][0] / ntot;
                int type = -1;
                int64_t anchor = 0;
                if (del >= del_t || ins >= ins_t) { prev = v + win; anchor = v - win; type = 0; }                  /* :232 */
                else if (dels >= del_t || inss >= ins_t || (dels + inss) >= 0.9) { prev = v + 10; anchor = v - 10; type = 1; }   /* :237 */
                if (type >= 0) {
                    if (anchor < 1) anchor = 1;
                    int found = 0;
                    for (int32_t q = 0; q < *n_var; q++)
                        if (var_pos[q] == (int32_t)anchor) { var_type[q] = type; found = 1; break; }
                    if (!found) {
                        if (*n_var >= cap) { rc = -2; break; }
                        var_pos[*n_var] = (int32_t)anchor; var_type[*n_var] = type; (*n_var)++;
                    }
                }
            }
            continue;
        }
        if (n0 >= mincov && n1 >= mincov) {                       /* :252 */
            const double del0 = n0 > 0 ? (double)distinct[0][0] / n0 : 0, del1 = n1 > 0 ? (double)distinct[0][1] / n1 : 0;
            const double ins0 = n0 > 0 ? (double)distinct[1][0] / n0 : 0, ins1 = n1 > 0 ? (double)distinct[1][1] / n1 : 0;
            const double dels0 = n0 > 0 ? (double)distinct[2][0] / n0 : 0, dels1 = n1 > 0 ? (double)distinct[2][1] / n1 : 0;
            const double inss0 = n0 > 0 ? (double)distinct[3][0] / n0 : 0, inss1 = n1 > 0 ? (double)distinct[3][1] / n1 : 0;
            int type = -1;
            int64_t anchor = 0;
            if ((del0 > del1 ? del0 : del1) >= del_t || (ins0 > ins1 ? ins0 : ins1) >= ins_t) {          /* :266 */
                prev = v + win; anchor = v - win; type = 0;
            } else if ((dels0 > dels1 ? dels0 : dels1) >= del_t || (inss0 > inss1 ? inss0 : inss1) >= ins_t ||
                       (dels0 + inss0) >= 0.9 || (dels1 + inss1) >= 0.9) {                               /* :271 */
                prev = v + 10; anchor = v - 10; type = 1;
            }
            if (type >= 0) {
                if (anchor < 1) anchor = 1;
                if (*n_var > 0 && var_pos[*n_var - 1] == anchor) var_type[*n_var - 1] = type;            /* dict overwrite */
                else {
                    if (*n_var >= cap) { rc = -2; break; }
                    var_pos[*n_var] = (int32_t)anchor; var_type[*n_var] = type; (*n_var)++;
                }
            }
        }
    }
    free(d); free(ecnt); free(eread); free(elen); free(fill); free(mult); free(ycol);
    return rc;
}
