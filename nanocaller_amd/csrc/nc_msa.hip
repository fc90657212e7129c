// Star alignment of read sets to their reference windows on the device (replaces the MUSCLE subprocess
// of generate_indel_pileups.py:24-44; the host statement of the same algorithm is nc_star_msa in nc_align.cpp, and the two
// agree bit for bit -- same recurrences, same tie rules, same end-point choice).
//
//   A  k_nw_fill       one lane per read: Gotoh DP of the read against its set's reference window, row by row; the H / F rows
//                      and the traceback bytes live in HBM in [cell][alignment] order, so the 64 lanes of a wave (64 reads)
//                      touch consecutive addresses on every access
//   B  k_nw_trace      one lane per read: end point (free tail), traceback -> the alignment in reference coordinates
//                      (read index aligned to every reference position, length / start of the insertion in every slot)
//   C1 k_set_columns   one workgroup per set: longest insertion per slot over the set's reads -> column of every reference
//                      position, number of columns
//   C2 k_set_rows      one workgroup per set: the aligned rows (symbols 0..4, other = 5) and the aligned reference row in the
//                      layout nc_indel_tensor reads
// then nc_indel_tensor's kernel (K8) on the rows.  Host round trips: the per-set column counts (row offsets are a prefix sum).
#include <cstdlib>
#include <vector>

#include "nc_common.h"

namespace {

constexpr int32_t NW_NEG = -(1 << 29);
enum : uint8_t { T_DIAG = 0, T_DEL = 1, T_INS = 2, T_EEXT = 4, T_FEXT = 8 };

struct NwArgs {
    const uint8_t *reads;          // concatenated read characters
    const int32_t *read_off;       // [A + 1]
    const int32_t *read_set;       // [A] set of every read
    const uint8_t *refs;           // concatenated reference windows
    const int32_t *ref_off;        // [n_sets + 1]
    int32_t A, Apad;               // alignments in this launch, padded to a multiple of 64 (the stride of the [cell][a] arrays)
    int32_t W;                     // max reference length + 1 (row pitch of the DP)
    int32_t a0;                    // first alignment of this launch (index into read_off / read_set)
    int32_t open, extend, match, mismatch;
    const int32_t *uniq;           // optional: the launch-local indices of the alignments to compute (duplicates skipped), n_uniq of them
    int32_t n_uniq;
    int32_t *Hrow, *Frow;          // [W][Apad]
    int32_t *hcol;                 // [N1 + 1][Apad]: H[i][n2]
    uint8_t *T;                    // [N1 + 1][W][Apad]
};

__global__ __launch_bounds__(64) void k_nw_fill(NwArgs p)
{
    const int al = blockIdx.x * 64 + threadIdx.x;
    if (al >= p.A) return;
    const int a = p.a0 + al;
    const uint8_t *s1 = p.reads + p.read_off[a];
    const int n1 = p.read_off[a + 1] - p.read_off[a];
    const int set = p.read_set[a];
    const uint8_t *s2 = p.refs + p.ref_off[set];
    const int n2 = p.ref_off[set + 1] - p.ref_off[set];
    const int64_t S = p.Apad;
    int32_t *H = p.Hrow + al, *F = p.Frow + al;
    uint8_t *T = p.T + al;
    H[0] = 0;
    F[0] = NW_NEG;
    for (int j = 1; j <= n2; j++) {
        H[j * S] = -p.open - (j - 1) * p.extend;
        F[j * S] = NW_NEG;
        T[j * S] = (uint8_t)(T_DEL | (j > 1 ? T_EEXT : 0));
    }
    p.hcol[al] = n2 > 0 ? -p.open - (n2 - 1) * p.extend : 0;
    for (int i = 1; i <= n1; i++) {
        const uint8_t c1 = s1[i - 1];
        int32_t hdiag = H[0];
        int32_t hleft = -p.open - (i - 1) * p.extend;
        H[0] = hleft;
        F[0] = hleft;
        uint8_t *Ti = T + (int64_t)i * p.W * S;
        Ti[0] = (uint8_t)(T_INS | (i > 1 ? T_FEXT : 0));
        int32_t e = NW_NEG;
        for (int j = 1; j <= n2; j++) {
            const int32_t hup = H[j * S], fup = F[j * S];
            uint8_t t = 0;
            const int32_t e_open = hleft - p.open, e_ext = e - p.extend;
            e = e_open;
            if (e_ext >= e_open) { e = e_ext; t |= T_EEXT; }
            const int32_t f_open = hup - p.open, f_ext = fup - p.extend;
            int32_t f = f_open;
            if (f_ext >= f_open) { f = f_ext; t |= T_FEXT; }
            const int32_t d = hdiag + (c1 == s2[j - 1] ? p.match : p.mismatch);
            int32_t h = d;
            uint8_t w = T_DIAG;
            if (e > h) { h = e; w = T_DEL; }
            if (f > h) { h = f; w = T_INS; }
            H[j * S] = h;
            F[j * S] = f;
            Ti[j * S] = (uint8_t)(t | w);
            hdiag = hup;
            hleft = h;
        }
        p.hcol[(int64_t)i * S + al] = hleft;                      // H[i][n2] (n2 = 0: H[i][0])
    }
}


// ---- A', B': the same DP with 16 lanes per alignment and the DP rows in REGISTERS ---------------------------------------------
// k_nw_fill keeps H / F and one traceback BYTE per cell in HBM (17 B of traffic per cell: 46 ms per 368 k alignments of 160 x 161,
// HBM-bound).  Here a group of 16 lanes owns one alignment, lane q the CPL consecutive reference columns q*CPL+1 .. q*CPL+CPL; the
// group sweeps the matrix in anti-diagonal order (at step t lane q works on read row t - q), H[i-1][.] / F[i-1][.] of a lane's own
// columns never leave its registers, and what crosses a column-block boundary -- H[i][j-1], E[i][j-1] -- moves one lane to the
// right with a DPP row shift (row_shr:1: a 16-lane DPP row IS the group).  Per cell only the 4-bit traceback code reaches HBM
// (8 cells per dword, 16 lanes = one 64-byte line per row and word): 0.5 B per cell instead of 17.  Same recurrences, same tie
// rules, same results as k_nw_fill / nc_star_msa.
__device__ __forceinline__ int32_t dpp_shr1(int32_t old, int32_t v)
{
    return __builtin_amdgcn_update_dpp(old, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}

template <int CPL>
__global__ __launch_bounds__(64) void k_nw_fill16(NwArgs p, int32_t N1, uint32_t *__restrict__ Tw, int32_t *__restrict__ Hlast,
                                                  int32_t *__restrict__ hcolA)
{
    constexpr int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;      // words per lane and row, padded to a vector store
    const int lane = threadIdx.x, g = lane >> 4, q = lane & 15;
    const int slot = blockIdx.x * 4 + g;
    const bool live = slot < (p.uniq ? p.n_uniq : p.A);
    const int al = live ? (p.uniq ? p.uniq[slot] : slot) : 0;
    int n1 = 0, n2 = 0;
    const uint8_t *s1 = p.reads, *s2 = p.refs;
    if (live) {
        const int a = p.a0 + al;
        s1 = p.reads + p.read_off[a];
        n1 = p.read_off[a + 1] - p.read_off[a];
        const int set = p.read_set[a];
        s2 = p.refs + p.ref_off[set];
        n2 = p.ref_off[set + 1] - p.ref_off[set];
    }
    int32_t H[CPL], F[CPL];
    int32_t rb[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int j = q * CPL + c + 1;
        H[c] = -p.open - (j - 1) * p.extend;                 // row 0
        F[c] = NW_NEG;
        rb[c] = j <= n2 ? (int32_t)s2[j - 1] : -1;
    }
    // the longest read of the wave's four alignments bounds the sweep
    int nmax = n1;
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    const int64_t arow = (int64_t)al * (N1 + 1);
    int32_t h_out = 0, e_out = NW_NEG;
    int32_t h_in_prev = q == 0 ? 0 : -p.open - (q * CPL - 1) * p.extend;      // H[0][q*CPL]: the diagonal of the lane's first row
    const int jn_lane = (n2 - 1) / CPL, jn_c = (n2 - 1) % CPL;                    // owner of column n2
    for (int t = 1; t <= nmax + 15; t++) {
        const int i = t - q;
        int32_t nh = dpp_shr1(0, h_out), ne = dpp_shr1(NW_NEG, e_out);
        if (q == 0) {
            nh = -p.open - (i - 1) * p.extend;                // H[i][0]
            ne = NW_NEG;
        }
        const bool active = live && i >= 1 && i <= n1 && q * CPL < n2;
        if (active) {
            const int32_t c1 = (int32_t)s1[i - 1];
            int32_t hdiag = h_in_prev, hleft = nh, e = ne;
            uint32_t words[NWD];
#pragma unroll
            for (int k = 0; k < NWD; k++) words[k] = 0;
            // straight-line: columns beyond n2 (rb = -1: they never match) are computed like real ones -- nothing reads them:
            // they lie to the right of everything valid, in this lane and in the lanes after it
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int32_t hup = H[c], fup = F[c];
                const int32_t e_open = hleft - p.open, e_ext = e - p.extend;
                const uint32_t te = e_ext >= e_open ? (uint32_t)T_EEXT : 0u;
                e = max(e_open, e_ext);
                const int32_t f_open = hup - p.open, f_ext = fup - p.extend;
                const uint32_t tf = f_ext >= f_open ? (uint32_t)T_FEXT : 0u;
                const int32_t f = max(f_open, f_ext);
                const int32_t d = hdiag + (c1 == rb[c] ? p.match : p.mismatch);
                const int32_t h1 = max(d, e);
                const uint32_t w1 = e > d ? (uint32_t)T_DEL : (uint32_t)T_DIAG;
                const int32_t h = max(h1, f);
                const uint32_t w = f > h1 ? (uint32_t)T_INS : w1;
                H[c] = h;
                F[c] = f;
                words[c >> 3] |= (te | tf | w) << ((c & 7) * 4);
                hdiag = hup;
                hleft = h;
            }
            h_out = hleft;
            e_out = e;
            // one vector store per lane: the 16 lanes of the group write one contiguous 64 / 128 / 256-byte run per row
            uint32_t *tp = Tw + ((arow + i) * 16 + q) * NWP;
            if (NWP == 1) tp[0] = words[0];
            else if (NWP == 2) *reinterpret_cast<uint2 *>(tp) = make_uint2(words[0], words[1]);
            else *reinterpret_cast<uint4 *>(tp) = make_uint4(words[0], words[1], NWD > 2 ? words[NWD > 2 ? 2 : 0] : 0u, 0u);
            if (q == jn_lane) {
                int32_t hv = H[0];
#pragma unroll
                for (int c = 1; c < CPL; c++) hv = c == jn_c ? H[c] : hv;
                hcolA[arow + i] = hv;                         // H[i][n2]
            }
            if (i == n1) {
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    if (rb[c] >= 0) Hlast[(int64_t)al * p.W + q * CPL + c + 1] = H[c];
            }
        }
        if (i >= 1) h_in_prev = nh;                           // H[i][q*CPL]: the diagonal of the next row (before row 1: row 0's value)
    }
}


struct TraceOut {
    int16_t *qidx;                 // [A][W]: read index aligned to reference position j, -1 = gap
    int16_t *ins_len, *ins_q;      // [A][W]: insertion in slot j (before reference position j; slot n2 = after the last)
};

__global__ __launch_bounds__(64) void k_nw_trace(NwArgs p, TraceOut o)
{
    const int al = blockIdx.x * 64 + threadIdx.x;
    if (al >= p.A) return;
    const int a = p.a0 + al;
    const int n1 = p.read_off[a + 1] - p.read_off[a];
    const int set = p.read_set[a];
    const int n2 = p.ref_off[set + 1] - p.ref_off[set];
    const int64_t S = p.Apad;
    int16_t *qidx = o.qidx + (int64_t)al * p.W, *il = o.ins_len + (int64_t)al * p.W, *iq = o.ins_q + (int64_t)al * p.W;
    for (int j = 0; j <= n2; j++) { qidx[j] = -1; il[j] = 0; iq[j] = 0; }
    int i = n1, j = n2;
    if (n1 > 0 && n2 > 0) {                                           // free tail: best cell of the last row / last column
        int32_t best = p.Hrow[(int64_t)n2 * S + al];
        for (int jj = n2 - 1; jj >= 0; jj--) {
            const int32_t v = p.Hrow[(int64_t)jj * S + al];
            if (v > best) { best = v; i = n1; j = jj; }
        }
        for (int ii = n1 - 1; ii >= 0; ii--) {
            const int32_t v = p.hcol[(int64_t)ii * S + al];
            if (v > best) { best = v; i = ii; j = n2; }
        }
        if (i < n1) { il[n2] = (int16_t)(n1 - i); iq[n2] = (int16_t)i; }   // the rest of the read: insertion after the window
    }
    const uint8_t *T = p.T + al;
    int state = -1;
    while (i > 0 || j > 0) {
        const uint8_t t = T[((int64_t)i * p.W + j) * S];
        if (state < 0) {
            const int w = t & 3;
            if (w == T_DIAG) { qidx[j - 1] = (int16_t)(i - 1); i--; j--; continue; }
            state = w == T_DEL ? 1 : 2;
        }
        if (state == 1) {
            const bool ext = (t & T_EEXT) != 0;
            j--;                                                       // reference position j stays a gap
            if (!ext) state = -1;
        } else {
            const bool ext = (t & T_FEXT) != 0;
            il[j]++;
            iq[j] = (int16_t)(i - 1);
            i--;
            if (!ext) state = -1;
        }
    }
}


// B': traceback over the packed codes of k_nw_fill16 (boundary row / column codes are implied: row 0 is a deletion run, column
// 0 an insertion run)
__global__ __launch_bounds__(64) void k_nw_trace16(NwArgs p, TraceOut o, int32_t N1, int32_t CPL, const uint32_t *__restrict__ Tw,
                                                   const int32_t *__restrict__ Hlast, const int32_t *__restrict__ hcolA)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= (p.uniq ? p.n_uniq : p.A)) return;
    const int al = p.uniq ? p.uniq[slot] : slot;
    const int a = p.a0 + al;
    const int n1 = p.read_off[a + 1] - p.read_off[a];
    const int set = p.read_set[a];
    const int n2 = p.ref_off[set + 1] - p.ref_off[set];
    const int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;
    const int64_t arow = (int64_t)al * (N1 + 1);
    int16_t *qidx = o.qidx + (int64_t)al * p.W, *il = o.ins_len + (int64_t)al * p.W, *iq = o.ins_q + (int64_t)al * p.W;
    for (int j = 0; j <= n2; j++) { qidx[j] = -1; il[j] = 0; iq[j] = 0; }
    int i = n1, j = n2;
    if (n1 > 0 && n2 > 0) {                                           // free tail: best cell of the last row / last column
        int32_t best = Hlast[(int64_t)al * p.W + n2];
        for (int jj = n2 - 1; jj >= 0; jj--) {
            const int32_t v = jj > 0 ? Hlast[(int64_t)al * p.W + jj] : -p.open - (n1 - 1) * p.extend;
            if (v > best) { best = v; i = n1; j = jj; }
        }
        for (int ii = n1 - 1; ii >= 0; ii--) {
            const int32_t v = ii > 0 ? hcolA[arow + ii] : -p.open - (n2 - 1) * p.extend;
            if (v > best) { best = v; i = ii; j = n2; }
        }
        if (i < n1) { il[n2] = (int16_t)(n1 - i); iq[n2] = (int16_t)i; }   // the rest of the read: insertion after the window
    }
    int state = -1;
    while (i > 0 || j > 0) {
        uint32_t t;
        if (i == 0) t = T_DEL | (j > 1 ? T_EEXT : 0);
        else if (j == 0) t = T_INS | (i > 1 ? T_FEXT : 0);
        else {
            const int q = (j - 1) / CPL, c = (j - 1) % CPL;
            t = (Tw[((arow + i) * 16 + q) * NWP + (c >> 3)] >> ((c & 7) * 4)) & 15u;
        }
        if (state < 0) {
            const int w = t & 3;
            if (w == T_DIAG) { qidx[j - 1] = (int16_t)(i - 1); i--; j--; continue; }
            state = w == T_DEL ? 1 : 2;
        }
        if (state == 1) {
            const bool ext = (t & T_EEXT) != 0;
            j--;                                                       // reference position j stays a gap
            if (!ext) state = -1;
        } else {
            const bool ext = (t & T_FEXT) != 0;
            il[j]++;
            iq[j] = (int16_t)(i - 1);
            i--;
            if (!ext) state = -1;
        }
    }
}

// allele_prediction (generate_indel_pileups.py:77-127) on the device: GLOBAL alignment of a consensus (s1) against its reference
// window (s2) filled by k_nw_fill16 with parasail's scoring, then one lane per alignment: traceback from (n1, n2) into runs of
// CIGAR operations (7 '=', 8 'X', 1 'I' consumes the consensus, 2 'D' consumes the reference), and the reference's allele
// extraction over the runs in alignment order -- nc_allele_prediction (nc_align.cpp) statement by statement.
__global__ __launch_bounds__(64) void k_allele_trace16(NwArgs p, int32_t N1, int32_t CPL, const uint32_t *__restrict__ Tw,
                                                       const int32_t *__restrict__ max_range, int16_t *__restrict__ runs, int32_t run_cap,
                                                       int32_t *__restrict__ ref_len, int32_t *__restrict__ alt_len)
{
    const int al = blockIdx.x * 64 + threadIdx.x;
    if (al >= p.A) return;
    const int a = p.a0 + al;
    const uint8_t *s1 = p.reads + p.read_off[a];
    const int n1 = p.read_off[a + 1] - p.read_off[a];
    const int set = p.read_set[a];
    const uint8_t *s2 = p.refs + p.ref_off[set];
    const int n2 = p.ref_off[set + 1] - p.ref_off[set];
    const int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;
    const int64_t arow = (int64_t)al * (N1 + 1);
    int16_t *rop = runs + (int64_t)al * run_cap * 2, *rcn = rop + run_cap;          // runs in REVERSE alignment order
    int nr = 0, last_op = -1;
    auto push = [&](int op) {
        if (op == last_op) rcn[nr - 1]++;
        else if (nr < run_cap) { rop[nr] = (int16_t)op; rcn[nr] = 1; nr++; last_op = op; }
    };
    int i = n1, j = n2, state = -1;
    while (i > 0 || j > 0) {
        uint32_t t;
        if (i == 0) t = T_DEL | (j > 1 ? T_EEXT : 0);
        else if (j == 0) t = T_INS | (i > 1 ? T_FEXT : 0);
        else {
            const int q = (j - 1) / CPL, c = (j - 1) % CPL;
            t = (Tw[((arow + i) * 16 + q) * NWP + (c >> 3)] >> ((c & 7) * 4)) & 15u;
        }
        if (state < 0) {
            const int w = t & 3;
            if (w == T_DIAG) { push(s1[i - 1] == s2[j - 1] ? 7 : 8); i--; j--; continue; }
            state = w == T_DEL ? 1 : 2;
        }
        if (state == 1) {
            push(2);
            const bool ext = (t & T_EEXT) != 0;
            j--;
            if (!ext) state = -1;
        } else {
            push(1);
            const bool ext = (t & T_FEXT) != 0;
            i--;
            if (!ext) state = -1;
        }
    }
    // the reference's loop over the CIGAR (ref_cnt / alt_cnt indexed by op: only 1, 2, 7, 8 occur)
    bool indel = false, mm_before = false;
    int32_t rc7 = 0, rc8 = 0, rc2 = 0, ac7 = 0, ac8 = 0, ac1 = 0, mm_after = 0;
    const int32_t mr = max_range[a];
    auto clampi = [](int32_t v, int32_t n) { return v < 0 ? (v + n < 0 ? 0 : v + n) : (v > n ? n : v); };       // Python slice s[:v]
    int op = 0, cnt = 0;
    bool done = false;
    int32_t out_r = 0, out_a = 0;
    for (int k = nr - 1; k >= 0 && !done; k--) {
        op = rop[k];
        cnt = rcn[k];
        if (op == 8 || op == 7) {
            if (op == 7) { rc7 += cnt; ac7 += cnt; } else { rc8 += cnt; ac8 += cnt; }
            if (indel) mm_after += cnt;
            else mm_before = true;
        }
        if (op == 1) { ac1 += cnt; mm_after = 0; indel = true; }
        if (op == 2) { rc2 += cnt; mm_after = 0; indel = true; }
        const int32_t rsum = rc7 + rc8 + rc2;
        if (!indel && rsum >= mr + 10) {
            if (rc8) {
                const int32_t ol = op == 8 ? rsum : rsum - cnt;
                out_r = clampi(ol, n2);
                out_a = clampi(ol, n1);
            } else {
                out_r = -1;
                out_a = -1;
            }
            done = true;
            break;
        }
        if (indel && mm_after > 20) break;
    }
    if (!done) {
        const int32_t rsum = rc7 + rc8 + rc2, asum = ac7 + ac8 + ac1;
        int32_t ro = op == 8 ? rsum : rsum - cnt, ao = op == 8 ? asum : asum - cnt;
        if (!mm_before) { ro += 1; ao += 1; }
        out_r = clampi(ro, n2);
        out_a = clampi(ao, n1);
    }
    ref_len[a] = out_r;
    alt_len[a] = out_a;
}

// per set: columns.  set_read0[s] .. set_read0[s+1]: the set's alignments (indices local to the launch)
__global__ __launch_bounds__(256) void k_set_columns(int32_t W, const int32_t *__restrict__ set_read0, const int32_t *__restrict__ ref_off,
                                                     int32_t set0, const int16_t *__restrict__ ins_len, int32_t *__restrict__ col /* [sets][W] */,
                                                     int32_t *__restrict__ n_cols, const int32_t *__restrict__ dup)
{
    // dup[r] >= 0: alignment r is the same read window against the same reference as alignment dup[r] (a read of the "all reads"
    // set that also sits in a haplotype set): it was not aligned again, its traceback is the other one's
    __shared__ int32_t mx[1024];
    const int sl = blockIdx.x, s = set0 + sl;
    const int n2 = ref_off[s + 1] - ref_off[s];
    const int r0 = set_read0[sl], r1 = set_read0[sl + 1];
    for (int j = threadIdx.x; j <= n2; j += 256) {
        int m = 0;
        for (int r = r0; r < r1; r++) {
            const int rs = dup && dup[r] >= 0 ? dup[r] : r;
            m = max(m, (int)ins_len[(int64_t)rs * W + j]);
        }
        mx[j] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int j = 0; j <= n2; j++) { acc += mx[j]; col[(int64_t)sl * W + j] = acc + j; }
        n_cols[sl] = acc + n2;
    }
}

__device__ __forceinline__ uint8_t sym_code(uint8_t c)
{
    // anything but AGTC (a read base N) counts as a gap at its column: the reference's symbol table raises KeyError there
    // (generate_indel_pileups.py:56), the host statement (generate_indel_pileups.msa) maps it the same way
    return c == 'A' ? 0 : c == 'G' ? 1 : c == 'T' ? 2 : c == 'C' ? 3 : 4;
}

__global__ __launch_bounds__(256) void k_set_rows(NwArgs p, TraceOut o, const int32_t *__restrict__ set_read0, int32_t set0,
                                                  const int32_t *__restrict__ col, const int32_t *__restrict__ n_cols,
                                                  const int64_t *__restrict__ row_off, const int64_t *__restrict__ refrow_off,
                                                  uint8_t *__restrict__ rows, uint8_t *__restrict__ ref_rows, const int32_t *__restrict__ dup)
{
    const int sl = blockIdx.x, s = set0 + sl;
    const uint8_t *s2 = p.refs + p.ref_off[s];
    const int n2 = p.ref_off[s + 1] - p.ref_off[s];
    const int nc = n_cols[sl];
    const int r0 = set_read0[sl], r1 = set_read0[sl + 1];
    const int32_t *C = col + (int64_t)sl * p.W;
    uint8_t *R = rows + row_off[sl], *RR = ref_rows + refrow_off[sl];
    for (int c = threadIdx.x; c < nc; c += 256) RR[c] = 4;
    for (int64_t k = threadIdx.x; k < (int64_t)(r1 - r0) * nc; k += 256) R[k] = 4;
    __syncthreads();
    for (int j = threadIdx.x; j < n2; j += 256) RR[C[j]] = sym_code(s2[j]);
    for (int r = r0; r < r1; r++) {
        const uint8_t *s1 = p.reads + p.read_off[p.a0 + r];
        const int rs = dup && dup[r] >= 0 ? dup[r] : r;
        const int16_t *qidx = o.qidx + (int64_t)rs * p.W, *il = o.ins_len + (int64_t)rs * p.W, *iq = o.ins_q + (int64_t)rs * p.W;
        uint8_t *row = R + (int64_t)(r - r0) * nc;
        for (int j = threadIdx.x; j <= n2; j += 256) {
            if (j < n2 && qidx[j] >= 0) row[C[j]] = sym_code(s1[qidx[j]]);
            const int L = il[j];
            if (L > 0) {
                const int slot_cols = j == 0 ? C[0] : C[j] - C[j - 1] - 1;         // longest insertion of the set in this slot
                const int c0 = C[j] - slot_cols;
                for (int t = 0; t < L; t++) row[c0 + t] = sym_code(s1[iq[j] + t]);
            }
        }
    }
}

}   // namespace

// from nc_indel.hip
extern "C" int nc_indel_tensor(nc_ctx *ctx, int32_t n_sets, const uint8_t *rows_dev, const int64_t *row_off_dev, const int32_t *n_rows_dev,
                               const int32_t *n_cols_dev, const uint8_t *ref_rows_dev, const int64_t *ref_off_dev, int32_t max_cols,
                               float *x_dev, uint8_t *cns_dev);

// Host arrays in, tensors out: reads of set s = reads read_set0[s] .. read_set0[s+1]; x_dev [n_sets][5][128][2] (device),
// cns_host [n_sets][max_cols] (NC_CODE_ABSENT-padded consensus symbols, gaps kept as 4), n_cols_host [n_sets].
// Optional rows_host / ref_rows_host (+ their offsets): the aligned rows themselves, for checks against nc_star_msa.
extern "C" int nc_star_msa_tensor_dup(nc_ctx *ctx, int32_t n_sets, const char *reads, const int32_t *read_off, const int32_t *set_read0,
                                      const char *refs, const int32_t *ref_off, int32_t open, int32_t extend, int32_t match, int32_t mismatch,
                                      int32_t max_cols, float *x_dev, uint8_t *cns_host, int32_t *n_cols_host, uint8_t *rows_host,
                                      const int64_t *rows_host_off, uint8_t *ref_rows_host, const int64_t *ref_rows_host_off, const int32_t *al_dup);

extern "C" int nc_star_msa_tensor(nc_ctx *ctx, int32_t n_sets, const char *reads, const int32_t *read_off, const int32_t *set_read0,
                                  const char *refs, const int32_t *ref_off, int32_t open, int32_t extend, int32_t match, int32_t mismatch,
                                  int32_t max_cols, float *x_dev, uint8_t *cns_host, int32_t *n_cols_host, uint8_t *rows_host,
                                  const int64_t *rows_host_off, uint8_t *ref_rows_host, const int64_t *ref_rows_host_off)
{
    return nc_star_msa_tensor_dup(ctx, n_sets, reads, read_off, set_read0, refs, ref_off, open, extend, match, mismatch, max_cols, x_dev, cns_host,
                                  n_cols_host, rows_host, rows_host_off, ref_rows_host, ref_rows_host_off, nullptr);
}

// al_dup (optional, [n alignments]): al_dup[a] = b < a when alignment a is the same read window against the same reference
// window as alignment b (nc_indel_pass2_sets: the reads of an anchor's "all reads" set that also sit in one of its haplotype
// sets), else -1.  Such an alignment is not computed again: the set kernels read b's traceback.  Same results.
extern "C" int nc_star_msa_tensor_dup(nc_ctx *ctx, int32_t n_sets, const char *reads, const int32_t *read_off, const int32_t *set_read0,
                                      const char *refs, const int32_t *ref_off, int32_t open, int32_t extend, int32_t match, int32_t mismatch,
                                      int32_t max_cols, float *x_dev, uint8_t *cns_host, int32_t *n_cols_host, uint8_t *rows_host,
                                      const int64_t *rows_host_off, uint8_t *ref_rows_host, const int64_t *ref_rows_host_off, const int32_t *al_dup)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_sets < 0 || (n_sets && (!read_off || !set_read0 || !refs || !ref_off || !x_dev || !cns_host || !n_cols_host)) || max_cols < 1)
        return nc_fail(ctx, NC_ERR_ARG, "nc_star_msa_tensor: bad argument");
    if (n_sets == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const int32_t A = set_read0[n_sets];
    int32_t N1 = 0, N2 = 0;
    for (int32_t a = 0; a < A; a++) N1 = std::max(N1, read_off[a + 1] - read_off[a]);
    for (int32_t s = 0; s < n_sets; s++) {
        if (set_read0[s + 1] < set_read0[s] || ref_off[s + 1] <= ref_off[s]) return nc_fail(ctx, NC_ERR_ARG, "nc_star_msa_tensor: set %d malformed", s);
        N2 = std::max(N2, ref_off[s + 1] - ref_off[s]);
    }
    if (N1 > 1000 || N2 > 1000) return nc_fail(ctx, NC_ERR_ARG, "nc_star_msa_tensor: windows longer than 1000 bases");
    const int32_t W = N2 + 1;
    std::vector<int32_t> read_set((size_t)std::max(A, 1));
    for (int32_t s = 0; s < n_sets; s++)
        for (int32_t a = set_read0[s]; a < set_read0[s + 1]; a++) read_set[(size_t)a] = s;
    // ---- device copies of the inputs
    const size_t n_read_bytes = (size_t)read_off[A], n_ref_bytes = (size_t)ref_off[n_sets];
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        NC_TRY(nc_ensure(ctx, b, bytes + 16));
        if (bytes) NC_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return NC_OK;
    };
    NC_TRY(up(ctx->msa_reads, reads, n_read_bytes));
    NC_TRY(up(ctx->msa_read_off, read_off, ((size_t)A + 1) * 4));
    NC_TRY(up(ctx->msa_read_set, read_set.data(), (size_t)std::max(A, 1) * 4));
    NC_TRY(up(ctx->msa_refs, refs, n_ref_bytes));
    NC_TRY(up(ctx->msa_ref_off, ref_off, ((size_t)n_sets + 1) * 4));
    // ---- groups of whole sets, at most GROUP alignments each (bounds the traceback matrix: (N1+1) * W bytes per alignment)
    const int64_t per_al = (int64_t)(N1 + 1) * W;
    // one lane per read is latency-bound until every SIMD holds several waves: up to 512 k alignments (8 waves per SIMD) and
    // 16 GB of traceback bytes per launch
    int32_t GROUP = (int32_t)std::min<int64_t>(524288, std::max<int64_t>(64, ((int64_t)16 << 30) / std::max<int64_t>(per_al, 1)));
    GROUP &= ~63;
    std::vector<int32_t> n_cols((size_t)n_sets);
    std::vector<int64_t> row_off((size_t)n_sets + 1, 0), refrow_off((size_t)n_sets + 1, 0);
    int32_t s0 = 0;
    while (s0 < n_sets) {
        int32_t s1 = s0;
        while (s1 < n_sets && (s1 == s0 || set_read0[s1 + 1] - set_read0[s0] <= GROUP)) s1++;
        const int32_t ng = s1 - s0, a0 = set_read0[s0], Ag = set_read0[s1] - a0, Apad = std::max(64, (Ag + 63) & ~63);
        // register / DPP kernel when 16 lanes x <= 17 columns cover the window (160 b ONT and 260 b PacBio windows do)
        const int CPL = N2 <= 64 ? 4 : N2 <= 128 ? 8 : N2 <= 176 ? 11 : N2 <= 272 ? 17 : 0;
        const bool fast = CPL > 0 && !getenv("NC_MSA_LANE_PER_READ");
        const int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;
        if (fast) {
            NC_TRY(nc_ensure(ctx, ctx->msa_rows_hf, (size_t)std::max(Ag, 1) * W * 4));                       // Hlast [A][W]
            NC_TRY(nc_ensure(ctx, ctx->msa_hcol, (size_t)std::max(Ag, 1) * (N1 + 1) * 4));                    // hcol  [A][N1 + 1]
            NC_TRY(nc_ensure(ctx, ctx->msa_tb, (size_t)std::max(Ag, 1) * (N1 + 1) * NWP * 64 + 64));          // 4-bit codes
        } else {
            NC_TRY(nc_ensure(ctx, ctx->msa_rows_hf, (size_t)2 * W * Apad * 4));
            NC_TRY(nc_ensure(ctx, ctx->msa_hcol, (size_t)(N1 + 1) * Apad * 4));
            NC_TRY(nc_ensure(ctx, ctx->msa_tb, (size_t)per_al * Apad));
        }
        NC_TRY(nc_ensure(ctx, ctx->msa_trace, (size_t)3 * std::max(Ag, 1) * W * 2));
        NC_TRY(nc_ensure(ctx, ctx->msa_cols, ((size_t)ng * W + (size_t)ng + (size_t)ng + 1) * 4 + 64));
        std::vector<int32_t> sr0((size_t)ng + 1);
        for (int32_t k = 0; k <= ng; k++) sr0[(size_t)k] = set_read0[s0 + k] - a0;
        int32_t *col = (int32_t *)ctx->msa_cols.p, *ncol_dev = col + (size_t)ng * W, *sr0_dev = ncol_dev + ng;
        NC_HIP(ctx, hipMemcpyAsync(sr0_dev, sr0.data(), ((size_t)ng + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        // duplicates (register kernel only): launch-local source index per alignment, and the list of alignments to compute
        const int32_t *dup_dev = nullptr, *uniq_dev = nullptr;
        int32_t n_uniq = Ag;
        std::vector<int32_t> dl, ul;                              // copy sources: alive until this group's closing synchronisation
        if (al_dup && fast && Ag > 0) {
            dl.resize((size_t)Ag);
            ul.reserve((size_t)Ag);
            for (int32_t r = 0; r < Ag; r++) {
                const int32_t b = al_dup[a0 + r];
                // only a source inside this launch that is itself computed counts
                dl[(size_t)r] = (b >= a0 && b < a0 + r && al_dup[b] < 0) ? b - a0 : -1;
                if (dl[(size_t)r] < 0) ul.push_back(r);
            }
            n_uniq = (int32_t)ul.size();
            NC_TRY(nc_ensure(ctx, ctx->msa_dup, ((size_t)Ag + (size_t)n_uniq) * 4 + 64));
            int32_t *dd = (int32_t *)ctx->msa_dup.p;
            NC_HIP(ctx, hipMemcpyAsync(dd, dl.data(), (size_t)Ag * 4, hipMemcpyHostToDevice, ctx->stream));
            if (n_uniq) NC_HIP(ctx, hipMemcpyAsync(dd + Ag, ul.data(), (size_t)n_uniq * 4, hipMemcpyHostToDevice, ctx->stream));
            dup_dev = dd;
            uniq_dev = dd + Ag;
        }
        NwArgs p;
        p.reads = (const uint8_t *)ctx->msa_reads.p; p.read_off = (const int32_t *)ctx->msa_read_off.p;
        p.read_set = (const int32_t *)ctx->msa_read_set.p; p.refs = (const uint8_t *)ctx->msa_refs.p; p.ref_off = (const int32_t *)ctx->msa_ref_off.p;
        p.A = Ag; p.Apad = Apad; p.W = W; p.a0 = a0; p.open = open; p.extend = extend; p.match = match; p.mismatch = mismatch;
        p.uniq = uniq_dev; p.n_uniq = n_uniq;
        p.Hrow = (int32_t *)ctx->msa_rows_hf.p; p.Frow = p.Hrow + (size_t)W * Apad; p.hcol = (int32_t *)ctx->msa_hcol.p; p.T = (uint8_t *)ctx->msa_tb.p;
        TraceOut o;
        o.qidx = (int16_t *)ctx->msa_trace.p; o.ins_len = o.qidx + (size_t)std::max(Ag, 1) * W; o.ins_q = o.ins_len + (size_t)std::max(Ag, 1) * W;
        if (Ag > 0 && fast) {
            uint32_t *Tw = (uint32_t *)ctx->msa_tb.p;
            int32_t *Hl = (int32_t *)ctx->msa_rows_hf.p, *hc = (int32_t *)ctx->msa_hcol.p;
            const int32_t nrun = std::max(n_uniq, 1);
            const dim3 gr((unsigned)((nrun + 3) / 4));
            if (CPL == 4) hipLaunchKernelGGL(k_nw_fill16<4>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
            else if (CPL == 8) hipLaunchKernelGGL(k_nw_fill16<8>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
            else if (CPL == 11) hipLaunchKernelGGL(k_nw_fill16<11>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
            else hipLaunchKernelGGL(k_nw_fill16<17>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
            hipLaunchKernelGGL(k_nw_trace16, dim3((nrun + 63) / 64), dim3(64), 0, ctx->stream, p, o, N1, CPL, (const uint32_t *)Tw, (const int32_t *)Hl,
                               (const int32_t *)hc);
        } else if (Ag > 0) {
            hipLaunchKernelGGL(k_nw_fill, dim3((Ag + 63) / 64), dim3(64), 0, ctx->stream, p);
            hipLaunchKernelGGL(k_nw_trace, dim3((Ag + 63) / 64), dim3(64), 0, ctx->stream, p, o);
        }
        hipLaunchKernelGGL(k_set_columns, dim3(ng), dim3(256), 0, ctx->stream, W, sr0_dev, p.ref_off, s0, o.ins_len, col, ncol_dev, dup_dev);
        NC_HIP(ctx, hipGetLastError());
        NC_HIP(ctx, hipMemcpyAsync(n_cols.data() + s0, ncol_dev, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream));
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // rows of the group in the K8 layout
        std::vector<int64_t> ro((size_t)ng + 1, 0), rro((size_t)ng + 1, 0);
        std::vector<int32_t> nrows((size_t)ng);
        int32_t mc = 1;
        for (int32_t k = 0; k < ng; k++) {
            nrows[(size_t)k] = sr0[(size_t)k + 1] - sr0[(size_t)k];
            ro[(size_t)k + 1] = ro[(size_t)k] + (int64_t)nrows[(size_t)k] * n_cols[(size_t)(s0 + k)];
            rro[(size_t)k + 1] = rro[(size_t)k] + n_cols[(size_t)(s0 + k)];
            mc = std::max(mc, n_cols[(size_t)(s0 + k)]);
        }
        const size_t o_rows = 0, o_ref = (size_t)((ro[(size_t)ng] + 15) & ~int64_t(15)), o_ro = (o_ref + (size_t)rro[(size_t)ng] + 15) & ~(size_t)15,
                     o_rro = o_ro + ((size_t)ng + 1) * 8, o_nr = o_rro + ((size_t)ng + 1) * 8, o_cns = o_nr + (size_t)ng * 4;
        const int32_t mcols = std::max(max_cols, 1);
        NC_TRY(nc_ensure(ctx, ctx->msa_out, o_cns + (size_t)ng * mcols + 64));
        uint8_t *ob = (uint8_t *)ctx->msa_out.p;
        NC_HIP(ctx, hipMemcpyAsync(ob + o_ro, ro.data(), ((size_t)ng + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        NC_HIP(ctx, hipMemcpyAsync(ob + o_rro, rro.data(), ((size_t)ng + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        NC_HIP(ctx, hipMemcpyAsync(ob + o_nr, nrows.data(), (size_t)ng * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_set_rows, dim3(ng), dim3(256), 0, ctx->stream, p, o, sr0_dev, s0, col, ncol_dev, (const int64_t *)(ob + o_ro),
                           (const int64_t *)(ob + o_rro), ob + o_rows, ob + o_ref, dup_dev);
        NC_HIP(ctx, hipGetLastError());
        NC_TRY(nc_indel_tensor(ctx, ng, ob + o_rows, (const int64_t *)(ob + o_ro), (const int32_t *)(ob + o_nr), ncol_dev, ob + o_ref,
                               (const int64_t *)(ob + o_rro), mcols, x_dev + (size_t)s0 * 5 * 128 * 2, ob + o_cns));
        NC_HIP(ctx, hipMemcpyAsync(cns_host + (size_t)s0 * mcols, ob + o_cns, (size_t)ng * mcols, hipMemcpyDeviceToHost, ctx->stream));
        if (rows_host && rows_host_off && ref_rows_host && ref_rows_host_off) {
            for (int32_t k = 0; k < ng; k++) {
                const size_t nb = (size_t)(ro[(size_t)k + 1] - ro[(size_t)k]);
                if (nb) NC_HIP(ctx, hipMemcpyAsync(rows_host + rows_host_off[s0 + k], ob + o_rows + ro[(size_t)k], nb, hipMemcpyDeviceToHost, ctx->stream));
                NC_HIP(ctx, hipMemcpyAsync(ref_rows_host + ref_rows_host_off[s0 + k], ob + o_ref + rro[(size_t)k], (size_t)n_cols[(size_t)(s0 + k)],
                                           hipMemcpyDeviceToHost, ctx->stream));
            }
        }
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        s0 = s1;
    }
    for (int32_t s = 0; s < n_sets; s++) n_cols_host[s] = n_cols[(size_t)s];
    return NC_OK;
}

// nc_allele_prediction_batch on the device (same results: the fill kernel's recurrences and tie rules are nc_nw_cigar's).
// Host arrays in and out.  NC_ERR_CAPACITY when a window is longer than the register kernel covers (272 reference bases, 1000
// consensus bases) or a string is empty: the caller then uses the host version.
extern "C" int nc_allele_prediction_device(nc_ctx *ctx, int32_t n, const char *alts, const int32_t *alt_off, const char *refs,
                                           const int32_t *ref_off, const int32_t *max_range, int32_t *ref_len, int32_t *alt_len)
{
    if (!ctx) return NC_ERR_ARG;
    if (n < 0 || (n && (!alts || !alt_off || !refs || !ref_off || !max_range || !ref_len || !alt_len))) return nc_fail(ctx, NC_ERR_ARG, "nc_allele_prediction_device: bad argument");
    if (n == 0) return NC_OK;
    int32_t N1 = 0, N2 = 0;
    for (int32_t a = 0; a < n; a++) {
        const int32_t l1 = alt_off[a + 1] - alt_off[a], l2 = ref_off[a + 1] - ref_off[a];
        if (l1 <= 0 || l2 <= 0) return NC_ERR_CAPACITY;
        N1 = std::max(N1, l1);
        N2 = std::max(N2, l2);
    }
    const int CPL = N2 <= 64 ? 4 : N2 <= 128 ? 8 : N2 <= 176 ? 11 : N2 <= 272 ? 17 : 0;
    if (CPL == 0 || N1 > 1000) return NC_ERR_CAPACITY;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const int32_t W = N2 + 1;
    const int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;
    std::vector<int32_t> ident((size_t)n);
    for (int32_t a = 0; a < n; a++) ident[(size_t)a] = a;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        NC_TRY(nc_ensure(ctx, b, bytes + 16));
        if (bytes) NC_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return NC_OK;
    };
    NC_TRY(up(ctx->msa_reads, alts, (size_t)alt_off[n]));
    NC_TRY(up(ctx->msa_read_off, alt_off, ((size_t)n + 1) * 4));
    NC_TRY(up(ctx->msa_read_set, ident.data(), (size_t)n * 4));
    NC_TRY(up(ctx->msa_refs, refs, (size_t)ref_off[n]));
    NC_TRY(up(ctx->msa_ref_off, ref_off, ((size_t)n + 1) * 4));
    const int32_t run_cap = N1 + N2 + 2;
    // outputs + max_range + runs share msa_out / msa_trace
    NC_TRY(nc_ensure(ctx, ctx->msa_out, (size_t)n * 12 + 64));
    int32_t *d_mr = (int32_t *)ctx->msa_out.p, *d_rl = d_mr + n, *d_al = d_rl + n;
    NC_HIP(ctx, hipMemcpyAsync(d_mr, max_range, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    NC_TRY(nc_ensure(ctx, ctx->msa_trace, (size_t)n * run_cap * 4 + 64));
    NC_TRY(nc_ensure(ctx, ctx->msa_rows_hf, (size_t)n * W * 4));
    NC_TRY(nc_ensure(ctx, ctx->msa_hcol, (size_t)n * (N1 + 1) * 4));
    NC_TRY(nc_ensure(ctx, ctx->msa_tb, (size_t)n * (N1 + 1) * NWP * 64 + 64));
    NwArgs p;
    p.reads = (const uint8_t *)ctx->msa_reads.p; p.read_off = (const int32_t *)ctx->msa_read_off.p;
    p.read_set = (const int32_t *)ctx->msa_read_set.p; p.refs = (const uint8_t *)ctx->msa_refs.p; p.ref_off = (const int32_t *)ctx->msa_ref_off.p;
    p.A = n; p.Apad = std::max(64, (n + 63) & ~63); p.W = W; p.a0 = 0;
    p.open = 9; p.extend = 1; p.match = 20; p.mismatch = -10;                      // parasail.nw_trace(alt, ref, 9, 1, matrix 20 / -10), :79
    p.uniq = nullptr; p.n_uniq = n;
    p.Hrow = nullptr; p.Frow = nullptr; p.hcol = nullptr; p.T = nullptr;
    uint32_t *Tw = (uint32_t *)ctx->msa_tb.p;
    int32_t *Hl = (int32_t *)ctx->msa_rows_hf.p, *hc = (int32_t *)ctx->msa_hcol.p;
    const dim3 gr((unsigned)((n + 3) / 4));
    if (CPL == 4) hipLaunchKernelGGL(k_nw_fill16<4>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
    else if (CPL == 8) hipLaunchKernelGGL(k_nw_fill16<8>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
    else if (CPL == 11) hipLaunchKernelGGL(k_nw_fill16<11>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
    else hipLaunchKernelGGL(k_nw_fill16<17>, gr, dim3(64), 0, ctx->stream, p, N1, Tw, Hl, hc);
    hipLaunchKernelGGL(k_allele_trace16, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, p, N1, CPL, (const uint32_t *)Tw, (const int32_t *)d_mr,
                       (int16_t *)ctx->msa_trace.p, run_cap, d_rl, d_al);
    NC_HIP(ctx, hipGetLastError());
    NC_HIP(ctx, hipMemcpyAsync(ref_len, d_rl, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    NC_HIP(ctx, hipMemcpyAsync(alt_len, d_al, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NC_OK;
}
