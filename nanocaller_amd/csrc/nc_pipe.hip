// Device-resident indel featuriser (gfx950): get_indel_testing_candidates (reference generate_indel_pileups.py:129-371; haploid
// generate_indel_pileups_haploid.py:118-277) for all chunks of a contig with no host code between the column decisions and the CNN input.
//
//   plan   K7 (nc_indel.hip's kernels; col_type stays in HBM)
//          k_pick        one wave per chunk: the order-dependent anchor selection `if v_pos <= prev: continue` (:249,266-275) with the
//                        dict semantics of `variants[anchor] = type`, then the pass-2 range test (:306)
//          k_sets        one wave per anchor: reference window test (:325-327), the pileup at the anchor from the read pack's tile
//                        index, the hap0 / hap1 / all read sets with the first-maxcov policy and the mincov tests (:333-348)
//   run    per group of sites (bounded by the traceback workspace):
//          k_windows     one lane per (site, read): query_sequence[q : q + window] (:331) rebuilt from the position-addressed codes, the
//                        read's indel events and the bases that have no reference column (inserted bases, trailing soft clip)
//          k_fill16p     Gotoh DP, 16 lanes per alignment, rows in registers (nc_msa.hip's k_nw_fill16: same recurrences and tie
//                        rules), read bases passed down the lanes by DPP instead of a byte load per step
//          k_trace16p    traceback -> alignment in reference coordinates
//          k_site_tensor one workgroup per site: per read set the longest insertion per slot -> columns, the per-column symbol histogram
//                        straight from the tracebacks (no row matrix in HBM), msa()'s frequencies / consensus / tensor (:57-71)
//          k_fill16p + k_allele_trace16p   allele_prediction (:77-127) of every consensus against its window
//          k_alt_gather  the ALT prefixes, back to back
// Results equal nc_indel_pass2_sets -> nc_star_msa_tensor_dup -> nc_allele_prediction_device (tests/test_indel_pipeline.py).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "nc_common.h"

int nc_indel_scan_group_launch(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const uint8_t *excl_dev, int32_t n_chunks,
                               const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *prm, int32_t *consumed,
                               std::vector<IndelChunk> &ck, const IndelChunk **ck_dev_out, const int8_t **ctype_out, const int64_t *slot_off_dev, int32_t *err_bits_dev,
                               const int32_t *rd_start_dev, const int32_t *rd_end_dev, bool reuse_tables);
int nc_indel_check(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const nc_indel_scan_params *prm, const char *who);

namespace {

constexpr int PICK_CAP = 12288;        // anchors of one chunk held in LDS by k_pick (a 100 kb chunk has at most 9,092), one packed word each: 48 KB, three waves per CU
constexpr int TWB_PITCH = 36;        // words of banded traceback codes per block of 8 anti-diagonals and alignment (C = 2: 32 cells + 4 empty slots a superblock)
constexpr int BAND_NBLK4 = 44;       // ... stored in whole superblocks of four blocks
constexpr int BAND_NBLK = 41;        // blocks of 8 anti-diagonals of a banded ALLELE alignment: n1 + n2 <= 328 (the star alignments size theirs by the window: stage_a)
constexpr int CNS_CAP = 1024;          // alignment columns of one read set (window + the longest insertion of every slot)
constexpr int32_t NW_NEG = -(1 << 29);
enum : uint32_t { T_DIAG = 0, T_DEL = 1, T_INS = 2, T_EEXT = 4, T_FEXT = 8 };

struct PipeChunk {
    int32_t lo, hi, ncol;      // columns lo .. hi (lo = max(1, start))
    int32_t a_lo;              // anchors with a_lo < v <= hi go to pass 2 (:306)
    int64_t coloff;            // offset of the chunk's col_type
    int32_t seg0;              // offset of the chunk's anchor segment
    int32_t id;                // index in the caller's chunk list
};

// ---------------------------------------------------------------------------------------------------------------- plan
// dct['impute_indel_phase'] on the device pipeline (generate_indel_pileups.py:278-304; round 6).  K7 marks the columns that meet the rule's column-level
// predicate (:278-284) with col_type 2; the rule then groups the column's reads by their pileup STRING (base letter + '+n<inserted bases>' /
// '-nN..', upper-cased: :279,287-289), takes the largest group against the runner-up (or against everybody else; or, when one string holds more
// than 80 % of the reads, its first half against its second: :291-297) and makes the column an anchor 10 bp upstream when both sides hold mincov
// reads (:298-303); pass 2 uses those two read-name sets instead of the haplotype tags (:310-312).  Here: k_impute_flags runs the grouping for
// every col_type-2 column and rewrites it to 3 (anchor) or -1; k_pick treats 3 as the small-window rule and notes the source column in the
// anchor's type byte (bit 1 = imputed, bits 2-5 = column - anchor: 10 unless the anchor was clipped to 1); k_sets<.., true> repeats the grouping
// at the source column of an imputed anchor and reads the members' sides where it read the HP tags.  A read's string is keyed by a 64-bit hash
// of (letter, event length, inserted bases); columns deeper than IMP_CAP reads raise the capacity bit (host-assembled route).
constexpr int IMP_CAP = 512;
struct ImpArgs {
    const int32_t *tile_off;
    const nc_tile_entry *tile_ent;
    int32_t tile_pos0, tile_size, n_tiles;
    const uint8_t *codes;
    const int64_t *slot_off;
    int32_t n_reads;
    const int32_t *ev_off, *ev_pos, *ev_len, *ins_off;
    const uint8_t *ins_bases;
    const int32_t *ent_read;           // the read of every tile entry (K7's table), or NULL: by search on the slot offsets
    int32_t mincov;
};
struct ImpLds {                        // one wave's scratch
    uint64_t key[IMP_CAP];
    int64_t slot[IMP_CAP];             // the read's slot offset: what identifies it in any tile's entries
    uint16_t first[IMP_CAP], cnt[IMP_CAP];
    uint8_t side[IMP_CAP];             // 0: in neither set, 1: read_names_0, 2: read_names_1
};
// One wave.  -> n = reads in the column's pileup (file order; -1: more than IMP_CAP), L.slot / L.side filled; pass = both sets hold mincov reads
__device__ __forceinline__ int impute_group(const ImpArgs &p, int32_t v, ImpLds &L, bool &pass)
{
    const int lane = threadIdx.x & 63;
    const uint64_t lt = (1ull << lane) - 1;
    pass = false;
    const int t = (v - p.tile_pos0) / p.tile_size;
    if (v < p.tile_pos0 || t >= p.n_tiles) return 0;
    const int e0 = p.tile_off[t], e1 = p.tile_off[t + 1];
    int n = 0;
    for (int eb = e0; eb < e1; eb += 64) {
        const int e = eb + lane;
        nc_tile_entry ent;
        ent.start = 0; ent.end = 0; ent.base_flag = 0;
        if (e < e1) ent = p.tile_ent[e];
        const bool cov = e < e1 && ent.start <= v && v < ent.end;
        const uint64_t m = __ballot(cov);
        const int idx = n + __popcll(m & lt);
        if (cov && idx < IMP_CAP) {
            const int64_t so = (ent.base_flag & ~int64_t(15)) + (ent.start & ~15);
            int r;
            if (p.ent_read) r = p.ent_read[e];
            else {
                int lo = 0, hi = p.n_reads;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.slot_off[mid] < so) lo = mid + 1; else hi = mid; }
                r = lo;
            }
            const int f0 = p.ev_off[r], f1 = p.ev_off[r + 1];
            int lo = f0, hi = f1;                                            // first event on a column >= v
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (p.ev_pos[mid] < v) lo = mid + 1; else hi = mid; }
            const int k = lo;
            bool deleted = false;
            if (k > f0) { const int32_t el = p.ev_len[k - 1]; deleted = el < 0 && p.ev_pos[k - 1] - el >= v; }
            const int code = p.codes[(ent.base_flag & ~int64_t(15)) + v];
            const uint32_t letter = deleted ? 5u : (code < 4 ? (uint32_t)code : 4u);       // A G T C, N (any other base), '*'
            int32_t len = 0;
            if (k < f1 && p.ev_pos[k] == v) len = p.ev_len[k];
            uint64_t h = 1469598103934665603ull;                                             // FNV-1a over letter, length, inserted bases
            h = (h ^ letter) * 1099511628211ull;
            h = (h ^ (uint64_t)(uint32_t)len) * 1099511628211ull;
            if (len > 0)
                for (int i = p.ins_off[k]; i < p.ins_off[k + 1]; i++) h = (h ^ p.ins_bases[i]) * 1099511628211ull;
            L.key[idx] = h;
            L.slot[idx] = so;
        }
        n += __popcll(m);
    }
    if (n > IMP_CAP) return -1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // groups: first member and size of every read's string
    for (int i = lane; i < n; i += 64) {
        const uint64_t ki = L.key[i];
        int first = -1, c = 0;
        for (int j = 0; j < n; j++)
            if (L.key[j] == ki) { if (first < 0) first = j; c++; }
        L.first[i] = (uint16_t)first;
        L.cnt[i] = (uint16_t)c;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // sorted(groups, key = size, reverse = True) is stable: the largest, ties to the string seen first; then the runner-up
    auto top = [&](int skip) -> uint32_t {
        uint32_t best = 0;
        for (int i = lane; i < n; i += 64)
            if (L.first[i] == i && i != skip) best = max(best, ((uint32_t)L.cnt[i] << 16) | (uint32_t)(0xffff - i));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o));
        return best;
    };
    if (n == 0) return 0;
    const uint32_t b0 = top(-1);
    const int g0 = 0xffff - (int)(b0 & 0xffffu), c0 = (int)(b0 >> 16);
    int n0 = 0, n1 = 0;
    if ((double)c0 <= 0.8 * (double)n) {                                                     // :291
        const uint32_t b1 = top(g0);
        const int g1 = 0xffff - (int)(b1 & 0xffffu), c1 = (int)(b1 >> 16);
        const bool second = c1 >= p.mincov;                                                  // :293: the runner-up, else everybody else
        for (int i = lane; i < n; i += 64) L.side[i] = L.first[i] == g0 ? 1 : ((second ? L.first[i] == g1 : true) ? 2 : 0);
        n0 = c0;
        n1 = second ? c1 : n - c0;
    } else {                                                                                 // :295-296: the string's first half against its second
        const int half = c0 / 2;
        int base = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool in = i < n && L.first[i] == g0;
            const uint64_t m = __ballot(in);
            const int rank = base + __popcll(m & lt);
            if (i < n) L.side[i] = in ? (rank < half ? 1 : 2) : 0;
            base += __popcll(m);
        }
        n0 = half;
        n1 = c0 - half;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    pass = n0 >= p.mincov && n1 >= p.mincov;                                                 // :298
    return n;
}

// every col_type-2 column of the chunks -> 3 (the grouping yields two sets of mincov reads: an anchor, :298-303) or -1.  One block per chunk and
// slab of 4096 columns, a wave per 1024 of them.
__global__ __launch_bounds__(256) void k_impute_flags(const PipeChunk *__restrict__ pc, int8_t *__restrict__ ctype, ImpArgs p, int32_t *__restrict__ err)
{
    __shared__ ImpLds lds[4];
    const PipeChunk c = pc[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int8_t *ct = ctype + c.coloff;
    const int col_lo = (int)blockIdx.y * 4096 + wv * 1024;
    for (int b = col_lo; b < min(col_lo + 1024, c.ncol); b += 64) {
        const int col = b + lane;
        uint64_t m = __ballot(col < c.ncol && ct[col] == 2);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            bool pass;
            const int n = impute_group(p, c.lo + b + l, lds[wv], pass);
            if (n < 0 && lane == 0) atomicOr(err, 16);
            if (lane == 0) ct[b + l] = pass ? 3 : -1;
        }
    }
}

__global__ __launch_bounds__(64) void k_pick(const PipeChunk *__restrict__ pc, const int8_t *__restrict__ ctype, int32_t win,
                                             int32_t *__restrict__ seg_pos, int8_t *__restrict__ seg_type, int32_t *__restrict__ cnt,
                                             int32_t *__restrict__ err)
{
    __shared__ uint32_t anc[PICK_CAP];                               // (anchor - aoff) << 8 | type: 4 bytes an anchor (5 in two arrays held two waves per CU)
    const PipeChunk c = pc[blockIdx.x];
    const int32_t aoff = c.lo - win - 16;                            // an anchor is >= max(1, column - win)
    auto apos = [&](int i) { return (int32_t)(anc[i] >> 8) + aoff; };
    auto apack = [&](int32_t an, int tb) { return ((uint32_t)(an - aoff) << 8) | (uint32_t)(tb & 0xff); };
    const int lane = threadIdx.x;
    const int8_t *ct = ctype + c.coloff;
    int n = 0;
    int base = 0;
    bool over = false;
    // 4096 columns a round: four steps of 1024 columns (16 per lane, one 16-byte load each), loaded together -- the walk is a chain of dependent
    // loads (a step of 64 columns was 0.95 ms per chr20-sized contig for ~1 % flagged columns, a step of 1024 columns 0.21 ms);
    // m16 = this lane's flagged columns of the step still to be visited
    while (base < c.ncol) {
        uint32_t W[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int col0 = base + 1024 * b + 16 * lane;
            W[b][0] = W[b][1] = W[b][2] = W[b][3] = 0xffffffffu;
            if (col0 + 16 <= c.ncol) __builtin_memcpy(W[b], ct + col0, 16);
            else
                for (int k = 0; k < 16 && col0 + k < c.ncol; k++) reinterpret_cast<int8_t *>(W[b])[k] = ct[col0 + k];
        }
        int64_t skip_to = 0;                                          // columns before it are skipped (carried from step to step)
        int next_base = base + 4096;
        bool jump = false;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int bb = base + 1024 * b, col0 = bb + 16 * lane;
            if (jump || bb >= c.ncol) continue;
            const uint32_t (&w)[4] = W[b];
            uint32_t m16 = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t t = (w[k >> 2] >> ((k & 3) * 8)) & 0xffu;
                m16 |= ((t <= 1u || t == 3u) ? 1u : 0u) << k;                 // (3: an imputed column, k_impute_flags)
            }
            {
                const int64_t sh = skip_to - col0;                    // (a skip that reaches into this step)
                if (sh >= 16) m16 = 0;
                else if (sh > 0) m16 &= ~((1u << (int)sh) - 1u);
            }
            for (;;) {
                const uint64_t lm = __ballot(m16 != 0);
                if (!lm) break;
                const int l = __ffsll((long long)lm) - 1;
                const uint32_t mm = (uint32_t)__shfl((int)m16, l);
                const int bq = __ffs((int)mm) - 1;
                const int32_t v = c.lo + bb + 16 * l + bq;
                const int tc = (int)((__shfl((int)w[0], l) * (bq < 4) + __shfl((int)w[1], l) * (bq >= 4 && bq < 8) + __shfl((int)w[2], l) * (bq >= 8 && bq < 12) +
                                      __shfl((int)w[3], l) * (bq >= 12)) >> ((bq & 3) * 8)) & 0xff;
                const int32_t prev = tc == 0 ? v + win : v + 10;                     // :267, :273, :301
                const int32_t an = tc == 0 ? max(1, v - win) : max(1, v - 10);       // :268, :274, :302
                const int tb = tc == 3 ? (1 | 2 | ((v - an) << 2)) : tc;             // imputed: the small-window type + where its read sets come from
                // variants[an] = tb: the anchors stay sorted; an equal key is overwritten (dict), a smaller one (a small-window
                // anchor followed by a long-window one less than 30 columns later) goes a few places back
                int i = n;
                while (i > 0 && apos(i - 1) > an) i--;
                if (i > 0 && apos(i - 1) == an) {
                    // variants[an] is overwritten; extra_variants[an] (an imputed column's read sets) stays unless an imputed column writes it again
                    if (lane == 0) {
                        const int old = (int)(anc[i - 1] & 0xffu);
                        anc[i - 1] = apack(an, ((tb & 2) || !(old & 2)) ? tb : ((old & ~1) | tb));
                    }
                } else if (n >= PICK_CAP) {
                    over = true;
                } else {
                    if (lane == 0) {
                        for (int k = n; k > i; k--) anc[k] = anc[k - 1];
                        anc[i] = apack(an, tb);
                    }
                    n++;
                }
                __syncthreads();
                // every column up to `prev` is skipped by `if v_pos <= prev: continue` (:249)
                skip_to = (int64_t)prev - c.lo + 1;
                if (skip_to >= base + 4096) {
                    next_base = (int)min((int64_t)c.ncol, skip_to);
                    jump = true;
                    break;
                }
                if (skip_to >= bb + 1024) break;                     // the rest of this step is skipped; the next one takes the mask
                const int sh = (int)(skip_to - col0);                              // this lane's columns before skip_to are done
                if (sh >= 16) m16 = 0;
                else if (sh > 0) m16 &= ~((1u << sh) - 1u);
            }
        }
        base = next_base;
    }
    if (over && lane == 0) atomicOr(err, 1);
    // pass-2 range (:306) and the copy to the chunk's segment, in order
    int m = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const bool ok = k < n && apos(k) > c.a_lo && apos(k) <= c.hi;
        const uint64_t bm = __ballot(ok);
        if (ok) {
            const int w = m + __popcll(bm & ((1ull << lane) - 1));
            seg_pos[c.seg0 + w] = apos(k);
            seg_type[c.seg0 + w] = (int8_t)(anc[k] & 0xffu);
        }
        m += __popcll(bm);
    }
    if (lane == 0) cnt[c.id] = m;
}

__device__ __forceinline__ int block_scan(int v, int *wsum, int &tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
    inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int wp = 0;
    tot = 0;
    for (int w = 0; w < nw; w++) {
        const int s = wsum[w];
        if (w < wv) wp += s;
        tot += s;
    }
    return wp + inc;
}

// exclusive scan of in[0..n) into out[0..n], out[n] = total (one workgroup, every thread a contiguous chunk); `add` is added to every input first
__global__ __launch_bounds__(1024) void k_scan_excl(const int32_t *__restrict__ in, int32_t n, int32_t add, int32_t *__restrict__ out)
{
    __shared__ int wsum[16];
    const int per = (n + 1023) / 1024, i0 = threadIdx.x * per, i1 = min(n, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; i++) local += in[i] + add;
    int tot;
    const int inc = block_scan(local, wsum, tot);
    int run = inc - local;
    for (int i = i0; i < i1; i++) {
        const int v = in[i] + add;
        out[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) out[n] = tot;
}

constexpr int TWB_LOG = 3, TWB = 1 << TWB_LOG;                  // steps per block
__host__ __device__ __forceinline__ int tw_blocks(int n1) { return ((n1 + 15) >> TWB_LOG) + 1; }      // (traceback storage: below)
// ---- the same scans in two launches of many workgroups.  (One workgroup walking the whole array, every thread a contiguous chunk, is a chain of
// uncoalesced loads on one CU: 0.21 ms for the 120 k consensus lengths of a chr20-sized pass, 0.19 for its ALT lengths, 3 x 0.04 in the plan.)
// k_scan_part: sum of every tile of 4096 inputs; k_scan_apply: a tile's offset = the sum of the tiles before it (a block reduction over <= a few
// hundred partial sums), then the scan of its own 4096 inputs, four consecutive ones per thread.
enum { SC_PLAIN = 0, SC_TWB = 1, SC_POS = 2 };
template <int F>
__device__ __forceinline__ int sc_val(int x) { return F == SC_TWB ? tw_blocks(x) : F == SC_POS ? max(x, 0) : x; }
constexpr int SC_TILE = 4096;

template <int F>
__global__ __launch_bounds__(1024) void k_scan_part(const int32_t *__restrict__ in, int32_t n, int32_t add, long long *__restrict__ part,
                                                    const long long *__restrict__ base_in)
{
    __shared__ long long wsum[16];
    const int i0 = blockIdx.x * SC_TILE + threadIdx.x * 4;
    long long local = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) if (i0 + u < n) local += sc_val<F>(in[i0 + u]) + add;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < 16; w++) t += wsum[w];
        part[blockIdx.x] = t;
        if (blockIdx.x == 0) part[gridDim.x] = base_in ? base_in[0] : 0;       // snapshot of the running base (k_scan_apply's last block advances it)
    }
}

// out[i] = base + exclusive prefix; OUT = int32_t or int64_t.  total_out (nullable): the grand total as {low 31 bits, 0} (the row mailbox's format) when
// mbox_fmt, else a plain OUT at out[n].  base_io (nullable): advanced by the total.
template <int F, class OUT>
__global__ __launch_bounds__(1024) void k_scan_apply(const int32_t *__restrict__ in, int32_t n, int32_t add, const long long *__restrict__ part,
                                                     OUT *__restrict__ out, int32_t write_total, int32_t *__restrict__ total_mbox, long long *__restrict__ base_io)
{
    __shared__ long long wsum2[16];
    __shared__ int wsum[16];
    __shared__ long long s_off;
    const int nb = gridDim.x;
    long long before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 1024) before += part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if ((threadIdx.x & 63) == 0) wsum2[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = part[nb];
        for (int w = 0; w < 16; w++) t += wsum2[w];
        s_off = t;
    }
    __syncthreads();
    const int i0 = blockIdx.x * SC_TILE + threadIdx.x * 4;
    int v[4], local = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { v[u] = i0 + u < n ? sc_val<F>(in[i0 + u]) + add : 0; local += v[u]; }
    int tot;
    const int inc = block_scan(local, wsum, tot);
    long long run = s_off + inc - local;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (i0 + u < n) out[i0 + u] = (OUT)run;
        run += v[u];
    }
    if ((int)blockIdx.x == nb - 1 && threadIdx.x == 0) {
        const long long total = s_off + tot;
        if (write_total) out[n] = (OUT)total;
        if (total_mbox) { total_mbox[0] = (int32_t)((total - part[nb]) & 0x7fffffff); total_mbox[1] = (int32_t)((total - part[nb]) >> 31); }
        if (base_io) base_io[0] = total;
    }
}

constexpr int SC_PARTS = 65536;
template <int F, class OUT>
static int scan_launch(nc_ctx *ctx, hipStream_t st, DevBuf &partbuf, const int32_t *in, int32_t n, int32_t add, OUT *out, bool write_total, int32_t *total_mbox,
                       long long *base_io)
{
    const int nb = std::max(1, (n + SC_TILE - 1) / SC_TILE);
    // the partial-sum buffers are sized ONCE per pass (nc_indel_sites_plan: SC_PARTS entries): launches queued on another stream may still read
    // them, so a scan never re-allocates -- an array too long for them is refused
    if (((size_t)nb + 2) * 8 > partbuf.cap)
        return nc_fail(ctx, NC_ERR_CAPACITY, "scan of %d elements needs %d partial sums, the pass holds %zu", n, nb + 2, partbuf.cap / 8);
    long long *part = (long long *)partbuf.p;
    hipLaunchKernelGGL((k_scan_part<F>), dim3(nb), dim3(1024), 0, st, in, n, add, part, (const long long *)base_io);
    hipLaunchKernelGGL((k_scan_apply<F, OUT>), dim3(nb), dim3(1024), 0, st, in, n, add, (const long long *)part, out, write_total ? 1 : 0, total_mbox, base_io);
    return NC_OK;
}

__global__ __launch_bounds__(256) void k_flatten(const PipeChunk *__restrict__ pc, const int32_t *__restrict__ seg_pos, const int8_t *__restrict__ seg_type,
                                                 const int32_t *__restrict__ cnt, const int32_t *__restrict__ off, int32_t *__restrict__ anc_pos,
                                                 int8_t *__restrict__ anc_type, int32_t *__restrict__ anc_chunk)
{
    const PipeChunk c = pc[blockIdx.x];
    const int n = cnt[c.id], o = off[c.id];
    for (int k = threadIdx.x; k < n; k += 256) {
        anc_pos[o + k] = seg_pos[c.seg0 + k];
        anc_type[o + k] = seg_type[c.seg0 + k];
        anc_chunk[o + k] = c.id;
    }
}

struct SetArgs {
    const int32_t *tile_off;
    const nc_tile_entry *tile_ent;
    int32_t tile_pos0, tile_size, n_tiles;
    const uint8_t *ref_code;
    int32_t ref_pos0, ref_len;
    int64_t chrom_len;
    int32_t window_after, maxcov, mincov, haploid;
    const int64_t *slot_off;
    const int32_t *read_ps;
    int32_t n_reads;
    int32_t n_anchor;
    const int32_t *anc_pos, *anc_chunk;
    const int8_t *anc_type;
    // count pass out
    int32_t *kept, *nuniq;
    // fill pass in / out
    const int32_t *site_of, *al_of;
    int32_t *site_pos, *site_chunk, *site_type, *site_phase, *site_al0, *site_nr, *site_n2;
    int32_t *al_read, *al_site;
    uint8_t *al_member;
    // K7's per-entry tables (k_entry_cursors; NULL when pass 1 ran without them): the read of every tile entry, and its first event at or after every
    // 1024-column block of the tile (less 64 columns) -> the read without a search, and for k_windows the short stretch of the read's events around the anchor
    const int32_t *ent_read, *ent_cur, *ev_off;
    int32_t spt;
    int2 *al_ev;
    ImpArgs imp;                       // (k_sets<.., true>) the grouping of an imputed anchor's source column
    int32_t *err;
};

template <bool FILL, bool IMP>
__global__ __launch_bounds__(256) void k_sets(SetArgs p)
{
    __shared__ ImpLds imp_lds[IMP ? 4 : 1];
    const int lane = threadIdx.x & 63;
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= p.n_anchor) return;
    if (FILL && !p.kept[a]) return;
    const int32_t v = p.anc_pos[a];
    // reference window [v, min(chrom_len, v + window_after + 1)): upper-case AGTC only (:325-327)
    const int64_t b = min(p.chrom_len, (int64_t)v + p.window_after + 1);
    bool ok = b > v && v >= 1;
    if (!FILL) {
        bool mine = true;
        for (int64_t x = v + lane; x < b; x += 64) {
            const int64_t i = x - p.ref_pos0;
            mine = mine && i >= 0 && i < p.ref_len && p.ref_code[i] < 4;
        }
        ok = ok && __all(mine);
    }
    int n_all = 0, n_1 = 0, n_2 = 0, n_u = 0, first0 = -1;
    // an imputed anchor (impute_indel_phase): the two read sets of its source column stand where the HP tags stand otherwise (:310-312)
    int n_imp = -1;
    if constexpr (IMP) {
        const int tb = (int)(uint8_t)p.anc_type[a];
        if (tb & 2) {
            bool pass_unused;
            n_imp = impute_group(p.imp, v + (tb >> 2), imp_lds[threadIdx.x >> 6], pass_unused);
            if (n_imp < 0) { if (lane == 0) atomicOr(p.err, 16); n_imp = 0; }
        }
    }
    const int t = (v - p.tile_pos0) / p.tile_size;
    const int site = FILL ? p.site_of[a] : 0;
    const int al0 = FILL ? p.al_of[a] : 0;
    if (ok && v >= p.tile_pos0 && t < p.n_tiles) {
        const int e0 = p.tile_off[t], e1 = p.tile_off[t + 1];
        const uint64_t lt = (1ull << lane) - 1;
        for (int eb = e0; eb < e1; eb += 64) {
            const int e = eb + lane;
            nc_tile_entry ent;
            ent.start = 0; ent.end = 0; ent.base_flag = 0;
            if (e < e1) ent = p.tile_ent[e];
            const bool cov = e < e1 && ent.start <= v && v < ent.end;            // the pileup at the anchor, in pack (= file) order
            int hp = (int)((ent.base_flag >> 1) & 3);
            if constexpr (IMP) {
                if (n_imp >= 0) {
                    hp = 0;
                    if (cov) {
                        const ImpLds &L = imp_lds[threadIdx.x >> 6];
                        const int64_t so = (ent.base_flag & ~int64_t(15)) + (ent.start & ~15);
                        for (int j = 0; j < n_imp; j++)
                            if (L.slot[j] == so) { hp = L.side[j]; break; }
                    }
                }
            }
            const uint64_t m_all = __ballot(cov), m_1 = __ballot(cov && hp == 1), m_2 = __ballot(cov && hp == 2);
            const int my_all = n_all + __popcll(m_all & lt), my_1 = n_1 + __popcll(m_1 & lt), my_2 = n_2 + __popcll(m_2 & lt);
            int member = 0;
            if (cov) {
                if (p.haploid) member = my_all < p.maxcov ? 1 : 0;
                else member = ((hp == 1 && my_1 < p.maxcov) ? 1 : 0) | ((hp == 2 && my_2 < p.maxcov) ? 2 : 0) | (my_all < p.maxcov ? 4 : 0);
            }
            const uint64_t m_u = __ballot(member != 0);
            if (FILL) {
                // the entry's read: its slot offset is unique
                int r = -1;
                if (member != 0 || (cov && hp == 1 && first0 < 0)) {
                    if (p.ent_read) r = p.ent_read[e];
                    else {
                        const int64_t so = (ent.base_flag & ~int64_t(15)) + (ent.start & ~15);
                        int lo = 0, hi = p.n_reads;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (p.slot_off[mid] < so) lo = mid + 1; else hi = mid;
                        }
                        r = lo;
                    }
                }
                if (member != 0) {
                    const int w = al0 + n_u + __popcll(m_u & lt);
                    p.al_read[w] = r;
                    p.al_site[w] = site;
                    p.al_member[w] = (uint8_t)member;
                    if (p.al_ev) {
                        // the read's events around the anchor: [first event at or after the anchor's 1024-column block less 64 columns, first one at or
                        // after the block after next less 64) -- the first event on a column >= v lies in that stretch or is its end
                        const int h = (v - (p.tile_pos0 + t * p.tile_size)) >> 10;
                        const int32_t *cur = p.ent_cur + (int64_t)e * NC_ENT_CUR_PITCH(p.spt);
                        p.al_ev[w] = make_int2(cur[h], h + 2 <= p.spt ? cur[h + 2] : p.ev_off[r + 1]);
                    }
                }
                if (first0 < 0) {
                    const uint64_t mf = p.haploid ? m_all : m_1;
                    if (mf) first0 = __shfl(r, __ffsll((long long)mf) - 1);
                }
            }
            n_all += __popcll(m_all);
            n_1 += __popcll(m_1);
            n_2 += __popcll(m_2);
            n_u += __popcll(m_u);
        }
    }
    const int s_all = min(n_all, p.maxcov), s_1 = min(n_1, p.maxcov), s_2 = min(n_2, p.maxcov);
    const bool pass = ok && (p.haploid ? s_all >= p.mincov : (s_1 >= 2 && s_2 >= 2 && s_all >= p.mincov));     // :48, :345
    if (!FILL) {
        if (lane == 0) {
            p.kept[a] = pass ? 1 : 0;
            p.nuniq[a] = pass ? n_u : 0;
        }
    } else if (lane == 0) {
        p.site_pos[site] = v;
        p.site_chunk[site] = p.anc_chunk[a];
        p.site_type[site] = p.anc_type[a] & 1;                                          // (the window rule; bits 1-5: an imputed anchor's source)
        p.site_phase[site] = (!p.haploid && first0 >= 0) ? p.read_ps[first0] : 0;       // :349 (set 0 holds HP-tagged reads only)
        p.site_al0[site] = al0;
        p.site_n2[site] = (int32_t)(b - v);
        if (p.haploid) p.site_nr[site] = s_all;
        else { p.site_nr[site * 3] = s_1; p.site_nr[site * 3 + 1] = s_2; p.site_nr[site * 3 + 2] = s_all; }
    }
}

// ---------------------------------------------------------------------------------------------------------------- run
struct WinArgs {
    const uint8_t *codes;
    const int64_t *slot_off;
    const int32_t *rd_start, *rd_end;
    const int32_t *ev_off, *ev_pos, *ev_len, *ins_off, *tail_off;
    const uint8_t *ins_bases, *tail_bases, *read_flag;
    const int32_t *al_read, *al_site, *site_pos, *site_n2;    // al_* offset to the group's first alignment
    const int2 *al_ev;          // (k_sets) the stretch of the read's events that holds the first one at or after the anchor, or NULL: search them all
    int32_t A, W, WS;
    uint8_t *win;           // [A][WS]
    int32_t *n1;            // [A]
    unsigned long long *cells;       // [0] += n1 x n2 of every alignment (the full matrices), [1] += the cells the banded route computes
    // banded alignment (k_fill_band): the diagonals j - i the read's own CIGAR visits inside the window bound the band
    int8_t *band_lo;        // [A] lowest diagonal of the alignment's band (even, <= 0), or NULL: no banding
    int32_t *list1, *list2, *listF;   // alignments whose band fits 32 / 64 diagonals; the rest (full matrix)
    int32_t *counts;        // [0] list1, [1] list2, [2] listF (k_trace_band appends the paths that touch a band edge), [3] class F by width alone
    int32_t band_margin;    // diagonals kept free on either side of the CIGAR's range
    int8_t *wcls;           // [A] (k_windows16) band class of the window, for k_window_lists
};

// one window by one lane (the round-3 form): the walk every other form of this kernel must reproduce, and the route of the few windows the 16-lane
// form leaves out (more events than its LDS arrays hold, three insertions inside one 16-column group)
__device__ __forceinline__ void window_serial(const WinArgs &p, int al, int r, int32_t v, int &n_out, int &dmin_out, int &dmax_out)
{
    uint32_t *out = reinterpret_cast<uint32_t *>(p.win + (int64_t)al * p.WS);       // rows are 16-byte aligned
    int n = 0;
    int dcur = 0, dmin = 0, dmax = 0;                                                 // diagonal (window column - read index) of the CIGAR's path
    uint32_t acc = 0;
    auto emit = [&](uint32_t b) {                                                     // bases leave as whole words
        acc |= b << ((n & 3) * 8);
        n++;
        if ((n & 3) == 0) { out[(n >> 2) - 1] = acc; acc = 0; }
    };
    if (!(p.read_flag[r] & 1)) {
        const int e0 = p.ev_off[r], e1 = p.ev_off[r + 1];
        int lo = e0, hi = e1;                                  // first event on a column >= v
        if (p.al_ev) { const int2 b2 = p.al_ev[al]; lo = b2.x; hi = b2.y; }      // (3 probes inside one or two sectors instead of 17 over the whole read)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (p.ev_pos[mid] < v) lo = mid + 1; else hi = mid;
        }
        int k = lo;
        int32_t del_until = 0;                                 // last position of the deletion that covers v, if any
        if (k > e0) {
            const int32_t el = p.ev_len[k - 1];
            if (el < 0) del_until = p.ev_pos[k - 1] - el;
        }
        if (del_until >= v) { dcur = del_until + 1 - v; dmax = dcur; }     // the window opens inside a deletion: its first base sits on column del_until + 1
        const int32_t rs = p.rd_start[r], re = p.rd_end[r];
        const uint8_t *cd = p.codes + (p.slot_off[r] - (rs & ~15));       // code of position x at cd[x]; 16-position groups are aligned
        int32_t next_ev = k < e1 ? p.ev_pos[k] : INT32_MAX;
        int32_t x = v;
        while (n < p.W && x < re) {
            const int32_t x0 = x & ~15;
            const uint4 g = *reinterpret_cast<const uint4 *>(cd + x0);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
            const int32_t xe = min(x0 + 16, re);
            for (; x < xe && n < p.W; x++) {
                const int o = x - x0;
                uint32_t code = 0;
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) code = (o >> 2) == q4 ? gw[q4] : code;
                code = (code >> ((o & 3) * 8)) & 0xffu;
                if (x > del_until) emit(code);
                while (next_ev == x) {
                    const int32_t el = p.ev_len[k];
                    if (el > 0) {
                        const int nb = n;
                        for (int i = p.ins_off[k]; i < p.ins_off[k + 1] && n < p.W; i++) emit(p.ins_bases[i]);
                        dcur -= n - nb;
                        dmin = min(dmin, dcur);
                    } else {
                        del_until = x - el;
                        dcur -= el;
                        dmax = max(dmax, dcur);
                    }
                    k++;
                    next_ev = k < e1 ? p.ev_pos[k] : INT32_MAX;
                }
            }
        }
        if (x >= re) {                                           // the soft-clipped tail has no column of its own: an insertion behind the last one
            const int nb = n;
            for (int i = p.tail_off[r]; i < p.tail_off[r + 1] && n < p.W; i++) emit(p.tail_bases[i]);
            dcur -= n - nb;
            dmin = min(dmin, dcur);
        }
    }
    if (n & 3) out[n >> 2] = acc;
    n_out = n; dmin_out = dmin; dmax_out = dmax;
}

// band class of a window from the diagonals its CIGAR visits; writes band_lo, returns the class (0: 32 diagonals, 1: 64, 2: full matrix) and the band's cells
__device__ __forceinline__ int window_band(const WinArgs &p, int al, int n, int n2, int dmin, int dmax, long long &bandcells)
{
    // band of B = 32 or 64 diagonals around [dmin, dmax] (0 is inside: the path starts at the origin), the slack split evenly, lowest
    // diagonal even (the anti-diagonal sweep alternates between the even and the odd diagonals of the band)
    // a read that ends inside the window leaves last-row cells to the right of its path: the free tail may jump there (a deletion, then
    // a few chance matches of the read's last bases), so the band reaches the last row's end: hi >= n2 - n1
    dmax = max(dmax, n2 - n - p.band_margin + 1);
    const int w = dmax - dmin;
    const int cls = w + 2 * p.band_margin <= 31 ? 0 : w + 2 * p.band_margin <= 63 ? 1 : 2;
    const int B = cls == 0 ? 32 : 64;
    int lo = dmin - ((B - 1 - w) >> 1);
    lo -= lo & 1;
    p.band_lo[al] = (int8_t)(cls == 2 ? 0 : lo);
    bandcells = cls == 2 ? 0 : (long long)(n + n2) * (B / 2);
    return cls;
}

// class lists (one atomic per wave and class) and the cell counters; cls = -1 for lanes without a window
__device__ __forceinline__ void window_lists(const WinArgs &p, int al, int cls, long long mycells, long long bandcells)
{
    if (p.band_lo) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (!m) continue;
            int base = 0;
            if (threadIdx.x == __ffsll((long long)m) - 1) {
                base = atomicAdd(p.counts + c, __popcll(m));
                if (c == 2) atomicAdd(p.counts + 3, __popcll(m));
            }
            base = __shfl(base, __ffsll((long long)m) - 1);
            int32_t *lst = c == 0 ? p.list1 : c == 1 ? p.list2 : p.listF;
            if (cls == c) lst[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = al;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mycells += __shfl_xor(mycells, o); bandcells += __shfl_xor(bandcells, o); }
    if (threadIdx.x == 0 && mycells) atomicAdd(p.cells, (unsigned long long)mycells);
    if (threadIdx.x == 0 && bandcells) atomicAdd(p.cells + 1, (unsigned long long)bandcells);
}

__global__ __launch_bounds__(64) void k_windows(WinArgs p)
{
    const int al = blockIdx.x * 64 + threadIdx.x;
    long long mycells = 0, bandcells = 0;
    int cls = -1;
    if (al < p.A) {
        const int r = p.al_read[al], site = p.al_site[al];
        int n, dmin, dmax;
        window_serial(p, al, r, p.site_pos[site], n, dmin, dmax);
        p.n1[al] = n;
        mycells = (long long)n * p.site_n2[site];
        if (p.band_lo) cls = window_band(p, al, n, p.site_n2[site], dmin, dmax, bandcells);
    }
    window_lists(p, al, cls, mycells, bandcells);
}

// ---- the 16-lane form: four windows per wave.  A window is the reference stretch behind the anchor with the read's events applied, so it is
// written by two passes over those events instead of a walk over its bases:
//   pass A, lane = event (rounds of 16, carried): prefix sums of the deleted columns and of the inserted bases before every event -> the output
//           index of every insertion (its bases are copied there), which events the walk would still have processed (those on a column it
//           reaches before the window is full), and the diagonal of the CIGAR's path behind each of them -> [dmin, dmax];
//   pass B, lane = one aligned 16-column group of the read's position-addressed codes (one dwordx4): the events before / inside the group from
//           LDS -> index of the group's first kept column, mask of its deleted columns, at most two insertions inside it; 16 byte stores into
//           the window's LDS row.
// The row leaves as 16-byte pieces.  Results are those of window_serial on every window (tests: NC_PIPE_WINDOWS=serial | force16).
constexpr int WIN_EV_CAP = 64;         // events of one window kept in LDS
constexpr int WIN_ROW = 288;           // bytes of a window's LDS row (>= WS = 272)

template <int CTRL>
__device__ __forceinline__ int dpp_row(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row_scan_add(int x)         // inclusive, over the 16 lanes of a DPP row
{
    x += dpp_row<0x111>(0, x); x += dpp_row<0x112>(0, x); x += dpp_row<0x114>(0, x); x += dpp_row<0x118>(0, x);
    return x;
}
__device__ __forceinline__ int row_scan_max(int x)
{
    x = max(x, dpp_row<0x111>(INT32_MIN, x)); x = max(x, dpp_row<0x112>(INT32_MIN, x));
    x = max(x, dpp_row<0x114>(INT32_MIN, x)); x = max(x, dpp_row<0x118>(INT32_MIN, x));
    return x;
}
__device__ __forceinline__ int row_last(int x) { return __shfl(x, (int)(threadIdx.x | 15)); }      // lane 15 of the row

#ifndef NC_WIN16_WAVES
#define NC_WIN16_WAVES 6
#endif
__global__ __launch_bounds__(64, NC_WIN16_WAVES) void k_windows16(WinArgs p, int32_t force_serial)
{
    __shared__ int32_t s_pos[4][WIN_EV_CAP], s_len[4][WIN_EV_CAP];       // s_len > 0: inserted bases, <= 0: minus the deleted columns
    __shared__ uint32_t s_row[4][WIN_ROW / 4];
    const int lane = threadIdx.x, g = lane >> 4, q = lane & 15;
    const int al = blockIdx.x * 4 + g;
    const bool live = al < p.A;
    uint8_t *rowb = reinterpret_cast<uint8_t *>(&s_row[g][0]);
    for (int i = q; i < WIN_ROW / 4; i += 16) s_row[g][i] = 0;
    int r = 0, site = 0;
    int32_t v = 0;
    bool act = false;
    if (live) {
        r = p.al_read[al]; site = p.al_site[al];
        v = p.site_pos[site];
        act = !(p.read_flag[r] & 1);
    }
    const int W = p.W;
    int n = 0, dmin = 0, dmax = 0;
    bool serial = act && force_serial;
    int k = 0, e1 = 0, ne = 0;
    int32_t del_until = 0, rs = 0, re = 0;
    const uint8_t *cd = p.codes;
    int32_t Dsum = 0, Isum = 0, dcur = 0;
    bool all_events = false;
    __syncthreads();                                                     // (one wave: orders the LDS accesses for the compiler)
    {
        // first event on a column >= v: a 16-ary search by the group's lanes (two dependent loads for the ~70 events of k_sets' stretch
        // instead of six), the last step also fetching the event before it (a deletion that covers the anchor)
        const bool srch = act && !serial;
        int e0 = 0, lo = 0, hi = 0;
        if (srch) {
            e0 = p.ev_off[r];
            e1 = p.ev_off[r + 1];
            lo = e0; hi = e1;
            if (p.al_ev) { const int2 b2 = p.al_ev[al]; lo = b2.x; hi = b2.y; }
            rs = p.rd_start[r]; re = p.rd_end[r];
            cd = p.codes + (p.slot_off[r] - (rs & ~15));
        }
        while (__any(srch && hi - lo > 15)) {
            const bool on = srch && hi - lo > 15;
            const int step = (hi - lo + 15) >> 4, idx = lo + q * step;
            const int32_t pv = on && idx < hi ? p.ev_pos[idx] : INT32_MAX;
            const int cnt = __popc((uint32_t)(__ballot(pv < v) >> (16 * g)) & 0xffffu);         // the probes ascend: those before v are a prefix
            if (on) {
                if (cnt == 0) hi = lo;
                else { const int nlo = lo + (cnt - 1) * step + 1; hi = min(hi, lo + cnt * step); lo = nlo; }
            }
        }
        {
            const int idx = lo - 1 + q;                                               // lane 0: the event before the range
            const bool ld = srch && idx >= e0 && idx < hi;
            const int32_t pv = ld ? p.ev_pos[idx] : INT32_MAX, lv = ld ? p.ev_len[idx] : 0;
            const int cnt = __popc((uint32_t)(__ballot(ld && q > 0 && pv < v) >> (16 * g)) & 0xffffu);
            k = lo + cnt;
            const int src = (lane & 48) | cnt;                                        // the lane that holds event k - 1
            const int32_t pp = __shfl(pv, src), pl = __shfl(lv, src);
            if (srch && k > e0 && pl < 0) del_until = pp - pl;
        }
        if (srch) {
            if (del_until >= v) { dcur = del_until + 1 - v; dmax = dcur; }
            Dsum = dcur;                                                 // deleted columns of [v, ...] so far
        }
    }
    // ---- pass A
    {
        bool more = act && !serial;
        int32_t c_dend = del_until, c_head = INT32_MIN, c_prev = v - 1;
        while (__any(more)) {
            const int i = k + ne + q;
            const bool valid = more && i < e1;
            int32_t pos = INT32_MAX, el = 0, io = 0, ilen = 0;
            if (valid) {                                                             // (four independent loads)
                pos = p.ev_pos[i]; el = p.ev_len[i];
                io = p.ins_off[i];
                const int32_t io1 = p.ins_off[i + 1];
                ilen = el > 0 ? io1 - io : 0;
            }
            const int32_t L = el < 0 ? -el : 0;
            const bool is_ins = valid && el > 0;
            const int32_t Iin = row_scan_add(ilen), Din = row_scan_add(L);
            const int32_t Iex = Isum + Iin - ilen, Dex = Dsum + Din - L;             // inserted bases / deleted columns of the events before this one
            const int32_t dend_in = row_scan_max(L > 0 ? pos + L : INT32_MIN);
            const int32_t dend_ex = max(c_dend, dpp_row<0x111>(INT32_MIN, dend_in));
            const bool deleted = valid && dend_ex >= pos;                            // the event's own column is a deleted one
            const int32_t prevpos = dpp_row<0x111>(c_prev, pos);
            const bool head = valid && pos != prevpos;                               // first event on its column
            const int32_t nh = (pos - v) - Dex + (deleted ? 1 : 0) + Iex;            // (heads) bases written when the walk reaches the event's column
            const int32_t hs = row_scan_max(head ? nh : INT32_MIN);
            const int32_t ncol = max(c_head, hs);
            const bool processed = valid && ncol < W;
            const int32_t nat = (pos - v + 1) - Dex + Iex;                           // bases written before this insertion's own
            const int32_t emitted = processed && is_ins ? max(0, min(W - nat, ilen)) : 0;
            const int32_t delta = processed ? (is_ins ? -emitted : L) : 0;
            const int32_t dc = dcur + row_scan_add(delta);
            const int32_t mn = row_scan_max(processed && is_ins ? -dc : INT32_MIN), mx = row_scan_max(processed && !is_ins ? dc : INT32_MIN);
            const int32_t mnl = row_last(mn), mxl = row_last(mx);
            if (mnl != INT32_MIN) dmin = min(dmin, -mnl);
            if (mxl != INT32_MIN) dmax = max(dmax, mxl);
            if (valid) { s_pos[g][ne + q] = pos; s_len[g][ne + q] = is_ins ? ilen : -L; }
            // the inserted bases: up to four by the event's own lane (independent loads), a longer run by the 16 lanes of the group together --
            // a byte loop per lane waits for one load after the other (a planted 50-base insertion: 50 latencies on every wave)
            {
                uint32_t b4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) b4[t] = t < emitted && emitted <= 4 ? (uint32_t)p.ins_bases[io + t] : 0u;
#pragma unroll
                for (int t = 0; t < 4; t++) if (t < emitted && emitted <= 4) rowb[nat + t] = (uint8_t)b4[t];
                uint32_t em = (uint32_t)(__ballot(emitted > 4) >> (16 * g)) & 0xffffu;
                while (__any(em != 0)) {
                    const int src = (lane & 48) | (em ? __builtin_ctz(em) : 0);
                    const int32_t io_b = __shfl(io, src), nat_b = __shfl(nat, src), cnt_b = __shfl(emitted, src);
                    if (em) for (int t = q; t < cnt_b; t += 16) rowb[nat_b + t] = p.ins_bases[io_b + t];
                    em &= em - 1;
                }
            }
            const int nv = __popc((uint32_t)(__ballot(valid) >> (16 * g)) & 0xffffu);
            Isum += row_last(Iin); Dsum += row_last(Din);
            c_dend = max(c_dend, row_last(dend_in));
            c_head = max(c_head, row_last(hs));
            c_prev = row_last(pos);
            dcur = row_last(dc);
            ne += nv;
            if (more) all_events = k + ne >= e1;
            more = more && !all_events && c_head < W;
            if (more && ne + 16 > WIN_EV_CAP) { serial = true; more = false; }
        }
    }
    // ---- the read's end: the soft-clipped tail is an insertion behind the last column
    int32_t n_end = 0, tail_emit = 0, t0 = 0;
    if (act && !serial) {
        n = W;
        if (all_events) {
            n_end = (re - v) - Dsum + Isum;
            if (n_end < W) {
                t0 = p.tail_off[r];
                tail_emit = min(W - n_end, p.tail_off[r + 1] - t0);
                n = n_end + tail_emit;
                dmin = min(dmin, dcur - tail_emit);
            }
        }
    }
    __syncthreads();
    // ---- pass B
    {
        const int32_t xb = v & ~15;
        int ne_w = act && !serial ? ne : 0;
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) ne_w = max(ne_w, __shfl_xor(ne_w, o));
        ne_w = __builtin_amdgcn_readfirstlane(ne_w);
        bool cont = act && !serial;
        for (int round = 0; __any(cont); round++) {
            const int32_t x0 = xb + 16 * (q + 16 * round);
            int32_t Ib = 0, Db = 0;
            uint32_t delm = 0;
            int c1 = 16, l1 = 0, c2 = 16, l2 = 0, nin = 0;
            if (del_until >= v) {                                                    // the deletion the anchor lies in: columns v .. del_until
                const int32_t b = min(del_until, x0 - 1);
                if (b >= v) Db += b - v + 1;
                const int lo2 = max(v, x0) - x0, hi2 = min(del_until, x0 + 15) - x0;
                if (hi2 >= lo2) delm |= ((2u << hi2) - 1u) & ~((1u << lo2) - 1u);
            }
            for (int j = 0; j < ne_w; j++) {
                if (!(cont && j < ne)) continue;
                const int32_t pj = s_pos[g][j], lj = s_len[g][j];
                if (lj > 0) {
                    if (pj < x0) Ib += lj;
                    else if (pj < x0 + 16) {
                        if (nin == 0) { c1 = pj - x0; l1 = lj; } else if (nin == 1) { c2 = pj - x0; l2 = lj; }
                        nin++;
                    }
                } else {
                    const int32_t a = pj + 1, bb = pj - lj;                         // deleted columns a .. bb
                    const int32_t b = min(bb, x0 - 1);
                    if (b >= a) Db += b - a + 1;
                    const int lo2 = max(a, x0) - x0, hi2 = min(bb, x0 + 15) - x0;
                    if (hi2 >= lo2) delm |= ((2u << hi2) - 1u) & ~((1u << lo2) - 1u);
                }
            }
            if (nin > 2) serial = true;
            // kept columns of the group: inside [v, re), not deleted
            const int klo = max(v, x0) - x0, khi = min(re - 1, x0 + 15) - x0;
            uint32_t keep = 0;
            if (cont && khi >= klo) keep = (((2u << khi) - 1u) & ~((1u << klo) - 1u)) & ~delm;
            const int32_t base = max(x0 - v, 0) - Db + Ib;                           // index of the group's first kept column
            uint4 cw = make_uint4(0, 0, 0, 0);
            if (keep) cw = *reinterpret_cast<const uint4 *>(cd + x0);
            const uint32_t gw[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int32_t idx = base + __popc(keep & ((1u << c) - 1u)) + (c > c1 ? l1 : 0) + (c > c2 ? l2 : 0);
                if (((keep >> c) & 1u) && idx < W) rowb[idx] = (uint8_t)(gw[c >> 2] >> ((c & 3) * 8));
            }
            // another round while the next group's first column is inside the read and before the window's end
            const int32_t nbase = base + __popc(keep) + l1 + l2;                      // (index behind this group)
            const int32_t nb15 = row_last(nbase);
            cont = cont && (xb + 256 * (round + 1) < re) && nb15 < W;
        }
    }
    {
        const uint32_t sm = (uint32_t)(__ballot(serial) >> (16 * g)) & 0xffffu;
        serial = sm != 0;
    }
    for (int t = q; t < tail_emit; t += 16) if (!serial) rowb[n_end + t] = p.tail_bases[t0 + t];
    __syncthreads();
    if (serial) {
        if (q == 0) window_serial(p, al, r, v, n, dmin, dmax);
    } else if (live) {
        uint4 *out = reinterpret_cast<uint4 *>(p.win + (int64_t)al * p.WS);
        const uint4 *src = reinterpret_cast<const uint4 *>(&s_row[g][0]);
        for (int w4 = q; w4 * 16 < n; w4 += 16) out[w4] = src[w4];
    }
    if (live && q == 0) {
        p.n1[al] = n;
        long long bandcells = 0;
        if (p.band_lo) p.wcls[al] = (int8_t)window_band(p, al, n, p.site_n2[site], dmin, dmax, bandcells);
    }
}

// class lists and cell counters of k_windows16's windows.  (One returning atomic per wave on the three list counters and two on the cell
// counters -- what k_windows does -- made the four-windows-per-wave kernel 1.1 ms slower than the arithmetic it saves: 260 k atomics on
// five addresses are served one after the other.  Here a workgroup of 1024 threads reserves list space for 4096 windows at once.)
__global__ __launch_bounds__(1024) void k_window_lists(WinArgs p)
{
    __shared__ int32_t l_cnt[3], g_base[3];
    __shared__ unsigned long long l_cells[2];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 3) l_cnt[tid] = 0;
    if (tid < 2) l_cells[tid] = 0;
    __syncthreads();
    int cls[4], at[4];
    long long mycells = 0, bandcells = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int al = blockIdx.x * 4096 + u * 1024 + tid;
        cls[u] = -1; at[u] = 0;
        if (al < p.A) {
            const int n = p.n1[al], n2 = p.site_n2[p.al_site[al]];
            mycells += (long long)n * n2;
            if (p.band_lo) {
                cls[u] = p.wcls[al];
                if (cls[u] != 2) bandcells += (long long)(n + n2) * (cls[u] == 0 ? 16 : 32);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned long long m = __ballot(cls[u] == c);
            if (!m) continue;
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&l_cnt[c], __popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (cls[u] == c) at[u] = base + __popcll(m & ((1ull << lane) - 1ull));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mycells += __shfl_xor(mycells, o); bandcells += __shfl_xor(bandcells, o); }
    if (lane == 0 && mycells) atomicAdd(&l_cells[0], (unsigned long long)mycells);
    if (lane == 0 && bandcells) atomicAdd(&l_cells[1], (unsigned long long)bandcells);
    __syncthreads();
    if (tid < 3 && l_cnt[tid]) {
        g_base[tid] = atomicAdd(p.counts + tid, l_cnt[tid]);
        if (tid == 2) atomicAdd(p.counts + 3, l_cnt[2]);
    }
    if (tid < 2 && l_cells[tid]) atomicAdd(p.cells + tid, l_cells[tid]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (cls[u] < 0) continue;
        int32_t *lst = cls[u] == 0 ? p.list1 : cls[u] == 1 ? p.list2 : p.listF;
        lst[g_base[cls[u]] + at[u]] = blockIdx.x * 4096 + u * 1024 + tid;
    }
}

struct FillArgs {
    const uint8_t *s1;           // read a = s1 + a * s1_stride, n1[a] bases (codes 0..4)
    int32_t s1_stride;
    const int32_t *n1;
    const uint8_t *ref_code;     // reference window of alignment a: ref_code + site_pos[site] - ref_pos0, site_n2[site] bases
    int32_t ref_pos0;
    const int32_t *site_pos, *site_n2;
    const int32_t *al_site;      // site of alignment a, or (NULL) site0 + a / site_div
    int32_t site0, site_div;
    int32_t A, W;                // alignments; row pitch of Hlast
    int32_t open, extend, match, mismatch;
    const int64_t *arow;         // first traceback BLOCK (8 steps) of alignment a, or (NULL) a * tw_blocks(N1)
    int32_t N1;                  // longest read of the launch (row pitch of hcol: hcol_pitch(N1), a multiple of four words)
    uint32_t *Tw;
    int32_t *Hlast, *hcol;       // free-tail end point inputs (NULL for a global alignment, and when `endcell` is set)
    int2 *endcell;               // free-tail end point (i, j) of alignment a (k_end_cells), read by k_trace16p instead of Hlast / hcol
    // list mode (the banded route's fallback): entry x < min(*count, A) of `list` is the alignment, x its slot in Tw / Hlast / hcol / endcell
    const int32_t *list, *count;
};

__device__ __forceinline__ int32_t dpp_shr1(int32_t old, int32_t v)
{
    return __builtin_amdgcn_update_dpp(old, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}

__host__ __device__ __forceinline__ int hcol_pitch(int N1) { return (N1 + 1 + 3) & ~3; }
__host__ __device__ __forceinline__ int hlast_pitch(int W) { return (W + 3) & ~3; }      // row pitch of Hlast (FillArgs::W columns)

__device__ __forceinline__ int fill_site(const FillArgs &p, int al) { return p.al_site ? p.al_site[al] : p.site0 + al / p.site_div; }

// Traceback storage.  The DP runs as a wavefront: at step t lane q of a 16-lane group works on read row t - q and produces CPL 4-bit
// codes.  The codes of the 8 steps lane q spends on block t >> 3 of its alignment are ONE contiguous run of CPL words:
//     run = ((first_block + (t >> 3)) * 16 + q) * CPL            [words]
//     words 0 .. 8 F - 1      the F = CPL / 8 full words (8 codes each) of step s = t & 7 at s * F + w
//     words 8 F .. CPL - 1    the R = CPL % 8 remaining codes of every step, 4 R bits a step, step s at bit 4 R s (8 steps = R words)
// so a traceback that climbs a diagonal (row - 1, column - 1: same lane, step - 1) stays inside one 44-byte run (CPL = 11) for 8 steps.
// [Rows stored one after the other made the fill write 16 partial lines per instruction (33 ms, 2.6x the arithmetic); steps stored one
// after the other fixed the writes (15 ms) but left the traceback one 64-byte sector per step (11 GB per chr20-sized contig); whole words
// per step (2 for 11 codes, 4 for 17) wrote 28.5 GB per pass for 15.3 GB of codes.]  The fill kernels collect 8 steps per lane in LDS (a
// lane reads back only what it wrote itself) and write whole runs.
__device__ __forceinline__ int64_t tw_run(int64_t first_block, int t, int q, int CPL) { return ((first_block + (t >> TWB_LOG)) * 16 + q) * (int64_t)CPL; }
struct __attribute__((packed, aligned(4))) U4 { uint32_t x, y, z, w; };      // four words at a 4-byte aligned address (one dwordx4 access)

template <int NWP>
__device__ __forceinline__ void tw_stage(uint32_t *lds, int k, int t, int lane, const uint32_t *wd)
{
    uint32_t *ls = lds + ((k * TWB + (t & (TWB - 1))) * 64 + lane) * NWP;
    if (NWP == 1) ls[0] = wd[0];
    else if (NWP == 2) *reinterpret_cast<uint2 *>(ls) = make_uint2(wd[0], wd[1]);
    else *reinterpret_cast<uint4 *>(ls) = make_uint4(wd[0], wd[1], wd[2], 0u);
}
// the 8 steps of block t >> 3 of this lane, LDS (NWP words a step) -> its run in HBM (CPL words)
template <int CPL, int NWP>
__device__ __forceinline__ void tw_flush(const uint32_t *lds, int k, int t, int lane, int q, uint32_t *Tw, int64_t first_block)
{
    constexpr int F = CPL / 8, R = CPL % 8;
    uint32_t *dst = Tw + tw_run(first_block, t, q, CPL);
    uint32_t out[CPL];
#pragma unroll
    for (int w = 0; w < CPL; w++) out[w] = 0;
#pragma unroll
    for (int ts = 0; ts < TWB; ts++) {
        const uint32_t *ls = lds + ((k * TWB + ts) * 64 + lane) * NWP;
#pragma unroll
        for (int w = 0; w < F; w++) out[ts * F + w] = ls[w];
        if (R > 0) {
            const uint32_t part = ls[F] & ((1u << (4 * R)) - 1u);     // the step's last R codes (a step the lane never staged holds anything)
            const int pos = 4 * R * ts, dw = 8 * F + (pos >> 5), sh = pos & 31;
            out[dw] |= part << sh;
            if (sh + 4 * R > 32) out[dw + 1] |= part >> (32 - sh);
        }
    }
#pragma unroll
    for (int x = 0; x + 4 <= CPL; x += 4) *reinterpret_cast<U4 *>(dst + x) = U4{out[x], out[x + 1], out[x + 2], out[x + 3]};
#pragma unroll
    for (int x = CPL & ~3; x < CPL; x++) dst[x] = out[x];
}

template <int CPL>
__global__ __launch_bounds__(64) void k_fill16p(FillArgs p)
{
    constexpr int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;
    const int lane = threadIdx.x, g = lane >> 4, q = lane & 15;
    const int al_raw = blockIdx.x * 4 + g;
    const bool live = al_raw < p.A;
    const int al = live ? al_raw : 0;
    int n1 = 0, n2 = 0;
    const uint8_t *s1 = p.s1, *s2 = p.ref_code;
    if (live) {
        s1 = p.s1 + (int64_t)al * p.s1_stride;
        n1 = p.n1[al];
        const int site = fill_site(p, al);
        s2 = p.ref_code + (p.site_pos[site] - p.ref_pos0);
        n2 = p.site_n2[site];
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
    int nmax = n1;
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    __shared__ uint32_t tw_lds[TWB * 64 * NWP];
    const int64_t arow = p.arow ? p.arow[al] : (int64_t)al * tw_blocks(p.N1);
    int32_t h_out = 0, e_out = NW_NEG;
    int32_t h_in_prev = q == 0 ? 0 : -p.open - (q * CPL - 1) * p.extend;      // H[0][q*CPL]
    const int jn_lane = (n2 - 1) / CPL, jn_c = (n2 - 1) % CPL;                    // owner of column n2
    // read bases: lane q of a group holds base 16*blk + q; lane 0 takes base t-1 from lane (t-1) & 15, the others get theirs
    // from the lane to their left one step later (lane q works on read row t - q)
    int32_t chunk = q < n1 ? (int32_t)s1[q] : 4;
    int32_t chunk_nxt = 16 + q < n1 ? (int32_t)s1[16 + q] : 4;
    int32_t c1 = 4;
    for (int t = 1; t <= nmax + 15; t++) {
        const int i = t - q;
        if (t > 1 && ((t - 1) & 15) == 0) {
            chunk = chunk_nxt;
            const int idx = t - 1 + 16 + q;
            chunk_nxt = idx < n1 ? (int32_t)s1[idx] : 4;
        }
        const int32_t c_new = __shfl(chunk, (lane & 48) | ((t - 1) & 15));
        c1 = dpp_shr1(4, c1);
        if (q == 0) c1 = c_new;
        int32_t nh = dpp_shr1(0, h_out), ne = dpp_shr1(NW_NEG, e_out);
        if (q == 0) {
            nh = -p.open - (i - 1) * p.extend;                // H[i][0]
            ne = NW_NEG;
        }
        const bool active = live && i >= 1 && i <= n1 && q * CPL < n2;
        if (active) {
            int32_t hdiag = h_in_prev, hleft = nh, e = ne;
            uint32_t words[NWD];
#pragma unroll
            for (int k = 0; k < NWD; k++) words[k] = 0;
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
            {
                uint32_t wd[4] = {words[0], NWD > 1 ? words[NWD > 1 ? 1 : 0] : 0u, NWD > 2 ? words[NWD > 2 ? 2 : 0] : 0u, 0u};
                tw_stage<NWP>(tw_lds, 0, t, lane, wd);
            }
            if (p.hcol && q == jn_lane) {
                int32_t hv = H[0];
#pragma unroll
                for (int c = 1; c < CPL; c++) hv = c == jn_c ? H[c] : hv;
                p.hcol[(int64_t)al * hcol_pitch(p.N1) + i] = hv;      // H[i][n2]
            }
            if (p.Hlast && i == n1) {
#pragma unroll
                for (int c = 0; c < CPL; c++)
                    if (rb[c] >= 0) p.Hlast[(int64_t)al * hlast_pitch(p.W) + q * CPL + c + 1] = H[c];
            }
        }
        if (i >= 1) h_in_prev = nh;
        if (((t & (TWB - 1)) == TWB - 1 || t == nmax + 15) && live && (t >> TWB_LOG) < tw_blocks(n1)) tw_flush<CPL, NWP>(tw_lds, 0, t, lane, q, p.Tw, arow);
    }
}


// ---- the same DP with TWO alignments per 16-lane group: every score is an exact small integer (|H| < 5,500 + 1,300 for windows of
// <= 272 bases, consensus rows <= 1,024), so a lane keeps alignment A in the low and alignment B in the high 16 bits of each
// register and every recurrence is ONE packed 16-bit instruction for both (v_pk_sub_i16, v_pk_max_i16, ...).  The kernel is bound
// by vector issue (k_fill16p: 23.5 VALU per cell, ~70 % of the issue rate): packing halves the instructions per cell.
// Traceback bits (format 1, decoded by tb_code): bit 0 = E beats the diagonal, bit 1 = F beats both, bit 2 = E opened, bit 3 = F opened
// -- the raw sign bits of four differences, gathered by 32-bit and-ors; the decisions are those of k_fill16p (same ties).  22.6 vector
// instructions per cell pair: the registers hold H - open (what E's and F's openings need; the diagonal's `open` is folded into the score),
// the score is match + (mismatch - match) * min(base xor base, 1).
constexpr int NEG16 = -20000;              // "minus infinity": never selected, and NEG16 - extend - (any score) stays inside int16

// packed 16-bit VALU (two alignments per register).  Inline assembly: written as vector C the compiler turns the sign-mask
// arithmetic back into per-half compares and selects (measured: no fewer instructions than the 32-bit kernel).
#define NC_PK2(name, op)                                                                                     \
    __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b)                                         \
    {                                                                                                        \
        uint32_t r;                                                                                          \
        asm(op " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                                                    \
        return r;                                                                                            \
    }
NC_PK2(pk_sub, "v_pk_sub_i16")
NC_PK2(pk_add, "v_pk_add_i16")
NC_PK2(pk_max, "v_pk_max_i16")
NC_PK2(pk_min_u, "v_pk_min_u16")
#undef NC_PK2
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)     // a * b + c per half (low 16 bits)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// full-rate 32-bit ops on the packed pair (the traceback bits are gathered with these: the packed 16-bit forms issue at half rate)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c)      // (a & b) | c
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t splat16(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }
__device__ __forceinline__ uint32_t dpp_shr1_u(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ int32_t half_of(uint32_t v, int k) { return k == 0 ? (int32_t)(int16_t)(v & 0xffffu) : (int32_t)(int16_t)(v >> 16); }

template <int CPL>
__global__ __launch_bounds__(64) void k_fill16q(FillArgs p)
{
    constexpr int NWD = (CPL + 7) / 8, NWP = NWD <= 1 ? 1 : NWD == 2 ? 2 : 4;
    constexpr int NH = (CPL + 3) / 4;                          // packed registers of 4 cells x 4 bits per alignment
    const int lane = threadIdx.x, g = lane >> 4, q = lane & 15;
    const int pair = blockIdx.x * 4 + g;
    const int A_live = p.count ? min(*p.count, p.A) : p.A;
    if ((int)blockIdx.x * 8 >= A_live) return;
    int al[2], n1[2], n2[2];                                          // al: the slot in Tw / Hlast / hcol (the alignment itself outside list mode)
    bool live[2];
    const uint8_t *s2[2];
    uint32_t s1o[2];                                                  // the read's bases at p.s1 + s1o (32-bit offsets and block indices: a register fewer each than 64-bit values)
    uint32_t arow[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int a = pair * 2 + k;
        live[k] = a < A_live;
        al[k] = live[k] ? a : 0;
        n1[k] = 0; n2[k] = 0;
        s1o[k] = 0; s2[k] = p.ref_code;
        arow[k] = 0;
        if (live[k]) {
            const int ain = p.list ? p.list[a] : a;
            arow[k] = p.arow ? (uint32_t)p.arow[ain] : (uint32_t)a * (uint32_t)tw_blocks(p.N1);      // (a row table is per alignment, a uniform pitch per slot)
            s1o[k] = (uint32_t)ain * (uint32_t)p.s1_stride;
            n1[k] = p.n1[ain];
            const int site = fill_site(p, ain);
            s2[k] = p.ref_code + (p.site_pos[site] - p.ref_pos0);
            n2[k] = p.site_n2[site];
        }
    }
    uint32_t H[CPL], F[CPL], rb[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int j = q * CPL + c + 1;
        H[c] = splat16(-2 * p.open - (j - 1) * p.extend);     // row 0.  H[] holds H - open throughout: that is what both the cell to the
        F[c] = splat16(NEG16);                                 // right (E's opening) and the cell below (F's) need; the diagonal adds `open` back inside the score
        const uint32_t r0 = j <= n2[0] ? (uint32_t)s2[0][j - 1] : 8u, r1 = j <= n2[1] ? (uint32_t)s2[1][j - 1] : 8u;     // 8: no read base equals it
        rb[c] = r0 | (r1 << 16);
    }
    int nmax = max(n1[0], n1[1]);
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    __shared__ uint32_t tw_lds[2 * TWB * 64 * NWP];
    const uint32_t k_open = splat16(p.open), k_ext = splat16(p.extend), k_match = splat16(p.match + p.open), k_dmis = splat16(p.mismatch - p.match);
    const uint32_t k_one = splat16(1);
    uint32_t h_out = 0, e_out = splat16(NEG16);
    uint32_t h_in_prev = splat16(q == 0 ? -p.open : -2 * p.open - (q * CPL - 1) * p.extend);     // H[0][q*CPL] - open
    int jn_lane[2], jn_c[2];
#pragma unroll
    for (int k = 0; k < 2; k++) { jn_lane[k] = (n2[k] - 1) / CPL; jn_c[k] = (n2[k] - 1) % CPL; }
    const int n2_first = __builtin_amdgcn_readfirstlane(n2[0]);
    const int jc_uni = __all(n2[0] == n2_first && n2[1] == n2_first && n2_first > 0) ? (n2_first - 1) % CPL : -1;      // wave-uniform (scalar)
    // read bases of both alignments: lane q holds base 16*blk + q (packed), see k_fill16p
    auto load_chunk = [&](int idx) {
        const uint32_t b0 = idx < n1[0] ? (uint32_t)p.s1[s1o[0] + (uint32_t)idx] : 4u, b1 = idx < n1[1] ? (uint32_t)p.s1[s1o[1] + (uint32_t)idx] : 4u;
        return b0 | (b1 << 16);
    };
    uint32_t chunk = load_chunk(q), chunk_nxt = load_chunk(16 + q), c1 = splat16(4);
    for (int t = 1; t <= nmax + 15; t++) {
        const int i = t - q;
        if (t > 1 && ((t - 1) & 15) == 0) {
            chunk = chunk_nxt;
            chunk_nxt = load_chunk(t - 1 + 16 + q);
        }
#ifdef NC_ABL_FILL_NOBPERM
        const uint32_t c_new = chunk;
#else
        const uint32_t c_new = (uint32_t)__shfl((int)chunk, (lane & 48) | ((t - 1) & 15));
#endif
        c1 = dpp_shr1_u(splat16(4), c1);
        if (q == 0) c1 = c_new;
        uint32_t nh = dpp_shr1_u(0u, h_out), ne = dpp_shr1_u(splat16(NEG16), e_out);
        if (q == 0) {
            nh = splat16(-2 * p.open - (i - 1) * p.extend);   // H[i][0] - open
            ne = splat16(NEG16);
        }
        if (i >= 1) {                                          // rows beyond a read's end compute values nothing reads
            uint32_t hdiag = h_in_prev, hleft = nh, e = ne;
            uint32_t words[NH + 1];
#pragma unroll
            for (int k = 0; k <= NH; k++) words[k] = 0;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const uint32_t hup = H[c], fup = F[c];
                const uint32_t e_ext = pk_sub(e, k_ext);                             // (E's opening = hleft, F's = hup: both already H - open)
                const uint32_t d_e = pk_sub(e_ext, hleft);                           // < 0: E opened
                e = pk_max(hleft, e_ext);
                const uint32_t f_ext = pk_sub(fup, k_ext);
                const uint32_t d_f = pk_sub(f_ext, hup);                             // < 0: F opened
                const uint32_t f = pk_max(hup, f_ext);
                const uint32_t ne_b = pk_min_u(c1 ^ rb[c], k_one);                   // 1: the bases differ
                const uint32_t d = pk_add(hdiag, pk_mad(ne_b, k_dmis, k_match));     // (H - open of the diagonal) + score + open
                const uint32_t h1 = pk_max(d, e);
                const uint32_t d_1 = pk_sub(d, e);                                   // < 0: E beats the diagonal
                const uint32_t hh = pk_max(h1, f);
                const uint32_t d_2 = pk_sub(h1, f);                                  // < 0: F beats both
                const uint32_t h = pk_sub(hh, k_open);
                H[c] = h;
                F[c] = f;
                // the four sign bits of each half -> its 4-bit code (d_1 bit 0, d_2 bit 1, d_e bit 2, d_f bit 3), with 32-bit shifts and
                // and-ors (full rate; the bits of the two halves never meet), then into the row's words
                uint32_t acc = d_1 & 0x80008000u;                                    // (the first one in ends lowest)
                acc = and_or(d_2, 0x80008000u, acc >> 1);
                acc = and_or(d_e, 0x80008000u, acc >> 1);
                acc = and_or(d_f, 0x80008000u, acc >> 1);
                words[c >> 2] |= (acc >> 12) << ((c & 3) * 4);
                hdiag = hup;
                hleft = h;
            }
            h_out = hleft;
            e_out = e;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (!(live[k] && i <= n1[k] && q * CPL < n2[k])) continue;
#ifdef NC_ABL_FILL_NOSTORE
                if (i != 100000) continue;
#endif
                uint32_t wd[4] = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < NWD; j++) {
                    const uint32_t lo = words[2 * j], hi = words[2 * j + 1 <= NH ? 2 * j + 1 : NH];
                    wd[j] = __builtin_amdgcn_perm(hi, lo, k == 0 ? 0x05040100u : 0x07060302u);      // this alignment's halves of the two registers: one v_perm_b32
                }
                tw_stage<NWP>(tw_lds, k, t, lane, wd);
                if (p.hcol && q == jn_lane[k]) {
                    uint32_t hv = H[0];
                    if (jc_uni >= 0) {                               // every window of the wave has the same length: the cell is picked by a scalar branch
                        switch (jc_uni) {
#define NC_HV(C) case C: hv = H[C < CPL ? C : 0]; break;
                            NC_HV(1) NC_HV(2) NC_HV(3) NC_HV(4) NC_HV(5) NC_HV(6) NC_HV(7) NC_HV(8) NC_HV(9) NC_HV(10) NC_HV(11) NC_HV(12) NC_HV(13)
                            NC_HV(14) NC_HV(15) NC_HV(16)
#undef NC_HV
                        default: break;
                        }
                    } else {
#pragma unroll
                        for (int c = 1; c < CPL; c++) hv = c == jn_c[k] ? H[c] : hv;
                    }
                    p.hcol[(int64_t)al[k] * hcol_pitch(p.N1) + i] = half_of(hv, k) + p.open;
                }
                if (p.Hlast && i == n1[k]) {
#pragma unroll
                    for (int c = 0; c < CPL; c++)
                        if (q * CPL + c + 1 <= n2[k]) p.Hlast[(int64_t)al[k] * hlast_pitch(p.W) + q * CPL + c + 1] = half_of(H[c], k) + p.open;
                }
            }
            h_in_prev = nh;
        }
        if ((t & (TWB - 1)) == TWB - 1 || t == nmax + 15) {
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (live[k] && (t >> TWB_LOG) < tw_blocks(n1[k])) tw_flush<CPL, NWP>(tw_lds, k, t, lane, q, p.Tw, arow[k]);
        }
    }
}

// ---- the banded form.  Every read window was rebuilt from the reference and the read's own CIGAR events (k_windows), so the diagonals
// d = j - i the optimal path can visit are known up front: the range the CIGAR's path covers inside the window plus a margin.  The band
// of B = 32 C diagonals [lo, lo + B) (lo even) is swept by ANTI-DIAGONALS a = i + j: on an even a the band's even diagonals hold a cell,
// on an odd a the odd ones, B / 2 cells either way -- one (C = 1) or two (C = 2) per lane of a 16-lane group, all independent:
//     lane q, cell c, x = q C + c:   a even: d = lo + 2 x        a odd: d = lo + 2 x + 1          i = (a - d) / 2, j = (a + d) / 2
//     left (i, j-1) = diagonal d - 1 of a - 1:   a odd: the same lane cell     a even: lane cell x - 1 (row_shr:1 across lanes)
//     up   (i-1, j) = diagonal d + 1 of a - 1:   a odd: lane cell x + 1 (row_shl:1)     a even: the same lane cell
//     diag (i-1, j-1) = diagonal d of a - 2:     the same lane cell
// so a lane cell walks a staircase down its pair of diagonals: j grows on odd steps (the reference bases move one lane cell down, a new one
// enters at the top lane), i on even steps (the read bases move one lane cell up, a new one enters at lane 0).  321 steps of one or two
// cells replace 175 steps of 11 (k_fill16q), and 4 bits per cell and step leave as ONE word per lane and 8 steps: 2.6 KB of traceback
// codes per alignment instead of 19 KB.  Cells outside the rectangle compute bounded garbage nothing reads: H(0,0) = 0 is planted in the
// registers of step 0, everything around it starts at "minus infinity", and the recurrence itself then produces row 0 and column 0
// (E / F chains from the origin).  Cells outside the band read as minus infinity (what a DPP shift hands the lanes at a row's end).  Two alignments per
// group in the halves of every register, arithmetic and tie rules exactly those of k_fill16q; a path that touches an edge diagonal of the
// band is re-run on the full matrix (k_trace_band -> listF).
struct BandArgs {
    FillArgs f;                  // windows, reference, scoring (Tw / Hlast / hcol / endcell unused)
    const int32_t *list;         // the alignments of this class (indices into the group), *count of them
    const int32_t *count;
    const int8_t *band_lo;       // [A] lowest diagonal (even, -B < lo <= 0)
    uint32_t *Twb;               // [A][NBLK * TWB_PITCH] words: see TbBand
    int16_t *hrow, *hcolb;       // [A][64] H of the band's cells in the last row / the last column, by diagonal index d - lo
    int32_t NBLK;                // blocks of 8 anti-diagonals per alignment
    int32_t *redo_list, *redo_count;      // k_trace_band: alignments whose path touched an edge of the band
    int32_t edge;                // ... = came within `edge` diagonals of it (0: the edge diagonals themselves)
};

__device__ __forceinline__ uint32_t dpp_shl1_u(uint32_t old, uint32_t v)          // lane q <- lane q + 1; the row's last lane keeps `old`
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x101, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_shl1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }   // ... gets 0
__device__ __forceinline__ uint32_t dpp_shr1_z(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_ror_u(uint32_t v, int n)                   // lane q <- lane (q - n) mod 16
{
    return n == 1 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, true)        // (a rotation has a source for every lane: with bound_ctrl
                  : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x12f, 0xf, 0xf, true);       // the `old` operand need not be initialised)
}

template <int C>
__global__ __launch_bounds__(64) void k_fill_band(BandArgs p)
{
    constexpr int B = 32 * C;
    __shared__ __attribute__((aligned(16))) uint32_t tws[2][64][4 * C];
    const int cnt = *p.count;
    if ((int)blockIdx.x * 8 >= cnt) return;
    const int lane = threadIdx.x, g = lane >> 4, q = lane & 15;
    const int pair = blockIdx.x * 4 + g;
    const FillArgs &f = p.f;
    int al[2], n1[2], n2[2], l0[2];
    bool live[2];
    const uint8_t *s1[2], *s2[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int a = pair * 2 + k;
        live[k] = a < cnt;
        al[k] = p.list[live[k] ? a : 0];
        s1[k] = f.s1 + (int64_t)al[k] * f.s1_stride;
        const int site = fill_site(f, al[k]);
        s2[k] = f.ref_code + (f.site_pos[site] - f.ref_pos0);
        n1[k] = live[k] ? f.n1[al[k]] : 0;
        n2[k] = live[k] ? f.site_n2[site] : 0;
        l0[k] = -(int)p.band_lo[al[k]] / 2;
    }
    // scores carry the bias -NEG16 (H' = H + 20000, likewise E and F): minus infinity is 0, which is what a DPP shift with bound_ctrl hands
    // the lanes at a row's end -- no `old` operand to load; every recurrence is linear in the bias
    constexpr int BIAS = -NEG16;
    const uint32_t k_open = splat16(f.open), k_ext = splat16(f.extend), k_match = splat16(f.match + f.open), k_dmis = splat16(f.mismatch - f.match);
    const uint32_t k_one = splat16(1);
    auto rd_base = [&](int k, int idx) -> uint32_t { return idx >= 0 && idx < n1[k] ? (uint32_t)s1[k][idx] : 4u; };      // string index -> code; 4 / 8 never match
    auto rf_base = [&](int k, int idx) -> uint32_t { return idx >= 0 && idx < n2[k] ? (uint32_t)s2[k][idx] : 8u; };
    // state of anti-diagonal 0 (H holds H - open, as in k_fill16q)
    uint32_t Hp1[C], Hp2[C], Ep1[C], Fp1[C], rd[C], rf[C];
    int di[2][C];                                                       // row of this lane cell at a = 0 (its column is the negative)
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int x = q * C + c;
        di[0][c] = l0[0] - x;
        di[1][c] = l0[1] - x;
        const uint32_t h0 = (uint32_t)(x == l0[0] ? BIAS - f.open : 0) & 0xffffu, h1 = (uint32_t)(x == l0[1] ? BIAS - f.open : 0) & 0xffffu;
        Hp1[c] = h0 | (h1 << 16);
        Hp2[c] = 0; Ep1[c] = 0; Fp1[c] = 0;
        rd[c] = rd_base(0, di[0][c] - 1) | (rd_base(1, di[1][c] - 1) << 16);
        rf[c] = rf_base(0, -di[0][c] - 1) | (rf_base(1, -di[1][c] - 1) << 16);
    }
    // the streams of bases that enter: read element e = string index l0 + e at lane 0 (lane q of a chunk holds element 16 blk + q, the chunk
    // rotates left after every entry); reference element e = string index 16 C - 1 - l0 + e at lane 15 (lane q holds 16 blk + 15 - q, rotates right)
    auto rd_chunk = [&](int blk) -> uint32_t { return rd_base(0, l0[0] + 16 * blk + q) | (rd_base(1, l0[1] + 16 * blk + q) << 16); };
    auto rf_chunk = [&](int blk) -> uint32_t {
        return rf_base(0, 16 * C - 1 - l0[0] + 16 * blk + 15 - q) | (rf_base(1, 16 * C - 1 - l0[1] + 16 * blk + 15 - q) << 16);
    };
    uint32_t ch_rd = rd_chunk(0), ch_rf = rf_chunk(0), ch_rd_n = rd_chunk(1), ch_rf_n = rf_chunk(1);
    int nb[2], nbw = 0, a_tail = 1 << 20;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        nb[k] = live[k] ? (n1[k] + n2[k] + 7) >> 3 : 0;
        nbw = max(nbw, nb[k]);
        if (live[k]) a_tail = min(a_tail, min(2 * n1[k] - 2 * l0[k], 2 * n2[k] + 2 * l0[k] - B + 1));      // first step with a cell in the last row / column
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        nbw = max(nbw, __shfl_xor(nbw, o));
        a_tail = min(a_tail, __shfl_xor(a_tail, o));
    }
    nbw = __builtin_amdgcn_readfirstlane(nbw);
    const int b_tail = p.hrow ? __builtin_amdgcn_readfirstlane(max(0, (a_tail - 1) >> 3)) : nbw;      // (a global alignment ends at the corner: no last row / column to keep)
    uint32_t P[4] = {0, 0, 0, 0};
    // one step.  ODD: the reference base moves (j grows); even: the read base (i grows).  TAIL: the cells of the last row / last column leave
    auto step = [&](auto odd_tag, auto tail_tag, int a, int s) {
        constexpr bool ODD = decltype(odd_tag)::value, TAIL = decltype(tail_tag)::value;
        uint32_t hl[C], el[C], hu[C], fu[C];
        if (ODD) {
            const uint32_t rot = dpp_ror_u(ch_rf, 1);                   // (rotated first: the entry below then overwrites the chunk register in place, no copy)
            const uint32_t top = dpp_shl1_u(ch_rf, rf[0]);
#pragma unroll
            for (int c = 0; c + 1 < C; c++) rf[c] = rf[c + 1];
            rf[C - 1] = top;
            ch_rf = rot;
            const uint32_t hn = dpp_shl1_z(Hp1[0]), fn = dpp_shl1_z(Fp1[0]);
#pragma unroll
            for (int c = 0; c < C; c++) {
                hl[c] = Hp1[c]; el[c] = Ep1[c];
                hu[c] = c + 1 < C ? Hp1[c + 1 < C ? c + 1 : 0] : hn;
                fu[c] = c + 1 < C ? Fp1[c + 1 < C ? c + 1 : 0] : fn;
            }
        } else {
            const uint32_t rot = dpp_ror_u(ch_rd, 15);
            const uint32_t bot = dpp_shr1_u(ch_rd, rd[C - 1]);
#pragma unroll
            for (int c = C - 1; c > 0; c--) rd[c] = rd[c - 1];
            rd[0] = bot;
            ch_rd = rot;
            const uint32_t hn = dpp_shr1_z(Hp1[C - 1]), en = dpp_shr1_z(Ep1[C - 1]);
#pragma unroll
            for (int c = 0; c < C; c++) {
                hu[c] = Hp1[c]; fu[c] = Fp1[c];
                hl[c] = c > 0 ? Hp1[c > 0 ? c - 1 : 0] : hn;
                el[c] = c > 0 ? Ep1[c > 0 ? c - 1 : 0] : en;
            }
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t e_ext = pk_sub(el[c], k_ext);
            const uint32_t d_e = pk_sub(e_ext, hl[c]);                           // < 0: E opened
            const uint32_t e = pk_max(hl[c], e_ext);
            const uint32_t f_ext = pk_sub(fu[c], k_ext);
            const uint32_t d_f = pk_sub(f_ext, hu[c]);                           // < 0: F opened
            const uint32_t ff = pk_max(hu[c], f_ext);
            const uint32_t ne_b = pk_min_u(rd[c] ^ rf[c], k_one);
            const uint32_t d = pk_add(Hp2[c], pk_mad(ne_b, k_dmis, k_match));
            const uint32_t h1 = pk_max(d, e);
            const uint32_t d_1 = pk_sub(d, e);                                   // < 0: E beats the diagonal
            const uint32_t hh = pk_max(h1, ff);
            const uint32_t d_2 = pk_sub(h1, ff);                                 // < 0: F beats both
            const uint32_t h = pk_sub(hh, k_open);
            // the cell's four sign bits join the register of its group of four cells (16 bits a half: the first cell in ends lowest)
            const int cell = s * C + c;                                          // cell of the block: 8 C of them, four to a register
            uint32_t &Pr = P[cell >> 2];
            Pr = (cell & 3) == 0 ? (d_1 & 0x80008000u) : and_or(d_1, 0x80008000u, Pr >> 1);
            Pr = and_or(d_2, 0x80008000u, Pr >> 1);
            Pr = and_or(d_e, 0x80008000u, Pr >> 1);
            Pr = and_or(d_f, 0x80008000u, Pr >> 1);
            Hp2[c] = Hp1[c];
            Hp1[c] = h; Ep1[c] = e; Fp1[c] = ff;
        }
        if (TAIL) {
#pragma unroll
            for (int c = 0; c < C; c++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int i = (a >> 1) + di[k][c], j = ((a + 1) >> 1) - di[k][c];
                    const int kk = 2 * (q * C + c) + (ODD ? 1 : 0);
                    const int32_t hv = half_of(Hp1[c], k) - BIAS + f.open;
                    if (live[k] && i == n1[k] && j >= 0 && j <= n2[k]) p.hrow[(int64_t)al[k] * 64 + kk] = (int16_t)hv;
                    if (live[k] && j == n2[k] && i >= 0 && i < n1[k]) p.hcolb[(int64_t)al[k] * 64 + kk] = (int16_t)hv;
                }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    for (int b = 0; b < nbw; b++) {
        if ((b & 3) == 0 && b > 0) {                                       // 16 bases of either stream are used up every four blocks
            ch_rd = ch_rd_n; ch_rf = ch_rf_n;
            ch_rd_n = rd_chunk((b >> 2) + 1); ch_rf_n = rf_chunk((b >> 2) + 1);
        }
        const int a0 = 8 * b + 1;
        if (b < b_tail) {
#pragma unroll
            for (int s = 0; s < 8; s += 2) { step(T_{}, F_{}, a0 + s, s); step(F_{}, F_{}, a0 + s + 1, s + 1); }
        } else {
#pragma unroll
            for (int s = 0; s < 8; s += 2) { step(T_{}, T_{}, a0 + s, s); step(F_{}, T_{}, a0 + s + 1, s + 1); }
        }
        // the block's codes wait in LDS (a lane reads back only what it wrote) until four blocks -- 32 anti-diagonals -- are together: they leave as
        // 16 C bytes per lane, so that a 64-byte line holds 8 diagonals x 32 steps (the traceback stays on a line for ~32 steps instead of 8)
        const int bb = b & 3;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t sel = k == 0 ? 0x05040100u : 0x07060302u;
            uint32_t *ls = &tws[k][lane][bb * C];
            if (C == 1) ls[0] = __builtin_amdgcn_perm(P[1], P[0], sel);
            else *reinterpret_cast<uint2 *>(ls) = make_uint2(__builtin_amdgcn_perm(P[1], P[0], sel), __builtin_amdgcn_perm(P[3], P[2], sel));
        }
        if (bb == 3 || b == nbw - 1) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (!(live[k] && (b & ~3) < nb[k])) continue;
                uint4 *dst = reinterpret_cast<uint4 *>(p.Twb + (int64_t)al[k] * p.NBLK * TWB_PITCH + ((b >> 2) * (16 * C + 4) + q * C + 2) * 4);
                const uint4 *src = reinterpret_cast<const uint4 *>(&tws[k][lane][0]);
#pragma unroll
                for (int u = 0; u < C; u++) dst[u] = src[u];
            }
        }
    }
}

// The line cache of a banded traceback (TbLine's role).  k_fill_band stores the codes of four blocks of 8 anti-diagonals -- a superblock of 32 -- as
// 4 C words per lane: the words of cell x = q C + c of superblock sb at ((sb * (16 C + 4) + x + 2) * 4), block bb = 0 .. 3 of the superblock at + bb
// (C = 2: a lane's two cells share their words: + bb * 2 + h, steps 0-3 and 4-7 of both).  A 64-byte line is therefore 4 cells = 8 diagonals x 32
// steps: the walk changes lines every ~32 steps (every 8 when a line was one block of all 32 diagonals: 41 lines per alignment instead of ~13, 4.3 GB
// read per pass for 0.2 GB of path codes).  The two empty cell slots in front shift the lines by half a line: the middle of the band -- where
// k_windows put the CIGAR's own diagonals -- is the middle of a line, not the border between two.
// One line per walking lane in LDS, re-fetched in epochs.
template <int C>
struct TbBand {
    static constexpr int B = 32 * C;
    uint32_t *slot;                                                    // this lane's 16 words in LDS (odd pitch)
    const uint32_t *tw;                                                // the alignment's codes
    int lo, ckey, edge;
    bool touched;
    U4 pre[4];                                                         // the line one superblock further down the path (same group of diagonals), in flight or arrived
    int pkey;
    __device__ __forceinline__ int key(int i, int j) const { return ((i + j - 1) >> 5) * 16 + ((((j - i - lo) >> 1) + 2) >> 2); }      // (superblock, line)
    __device__ __forceinline__ bool has(int i, int j) const { return key(i, j) == ckey; }
    // the same from the walk's running coordinates: a = i + j - 1 (anti-diagonal), kd = j - i - lo (diagonal of the band)
    __device__ __forceinline__ int key_akd(int a, int kd) const { return (a >> 5) * 16 + (((kd >> 1) + 2) >> 2); }
    __device__ __forceinline__ uint32_t raw(int a, int kd) const          // the cell's four sign bits (any a, kd: the index stays inside the slot)
    {
        const int s = a & 7, bb = (a >> 3) & 3, xx = kd >> 1;
        if (C == 1) return (slot[((xx + 2) & 3) * 4 + bb] >> (4 * s)) & 15u;
        return (slot[(((xx >> 1) + 1) & 1) * 8 + bb * 2 + (s >> 2)] >> (4 * ((s & 3) * 2 + (xx & 1)))) & 15u;
    }
    __device__ __forceinline__ const U4 *line(int k) const { return reinterpret_cast<const U4 *>(tw + (k >> 4) * (64 * C + 16) + (k & 15) * 16); }
    __device__ __forceinline__ void to_slot(const U4 *v)
    {
#pragma unroll
        for (int u = 0; u < 4; u++) { slot[4 * u] = v[u].x; slot[4 * u + 1] = v[u].y; slot[4 * u + 2] = v[u].z; slot[4 * u + 3] = v[u].w; }
    }
    __device__ __forceinline__ void prefetch(int k)
    {
        pkey = k >= 0 ? k : -1;
        if (k >= 0) {
            const U4 *src = line(k);
#pragma unroll
            for (int u = 0; u < 4; u++) pre[u] = src[u];
        }
    }
    // demand load of the line of (i, j) (the wave waits for it), and the request for the one the path most likely enters next
    __device__ __forceinline__ void load(int i, int j)
    {
        ckey = key(i, j);
        const U4 *src = line(ckey);
        U4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = src[u];
        prefetch(ckey - 16);
        to_slot(v);
    }
    // the line of (i, j) into the slot at an epoch's start: from the prefetch registers when the guess was right (no memory wait), else from memory.
    // (Taking a prefetched line inside the walk, lane by lane as each one leaves its line, was tried: the wave then runs the 40-instruction hand-over
    // ~800 times instead of 13 epochs -- 0.96 -> 1.99 ms.)
    __device__ __forceinline__ void fetch(int i, int j)
    {
        const int k = key(i, j);
        if (k == pkey) {
            to_slot(pre);
            ckey = k;
            prefetch(k - 16);
        } else load(i, j);
    }
    // cell (i, j), i, j > 0, of the cached line as a T_* code; notes a cell on (or within `edge` of) an edge diagonal of the band
    __device__ __forceinline__ uint32_t code(int i, int j)
    {
        const int k = j - i - lo, a = i + j - 1, s = a & 7, bb = (a >> 3) & 3, xx = k >> 1;
        touched |= k <= edge || k >= B - 1 - edge;
        uint32_t tc;
        if (C == 1) tc = (slot[((xx + 2) & 3) * 4 + bb] >> (4 * s)) & 15u;
        else tc = (slot[(((xx >> 1) + 1) & 1) * 8 + bb * 2 + (s >> 2)] >> (4 * ((s & 3) * 2 + (xx & 1)))) & 15u;
        return ((tc & 2u) ? (uint32_t)T_INS : (tc & 1u) ? (uint32_t)T_DEL : (uint32_t)T_DIAG) | ((tc & 4u) ? 0u : (uint32_t)T_EEXT) | ((tc & 8u) ? 0u : (uint32_t)T_FEXT);
    }
};
constexpr int TBB_PITCH = 17;

// traceback of a banded alignment: k_trace16p's walk and entries.  The end point (best cell of the last row, ties to the larger column, or a strictly
// better cell of the last column, ties to the larger row) comes from the band's 2 x B last-row / last-column values.
template <int C>
__device__ __forceinline__ void trace_band_body(const BandArgs &p, uint32_t *__restrict__ ent_all, int32_t EW, uint32_t *stage, uint32_t *tbl)
{
    constexpr int B = 32 * C;
    const int cnt = *p.count;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= cnt) return;
    const int lane = threadIdx.x;
    const FillArgs &f = p.f;
    const int al = p.list[idx];
    const int n1 = f.n1[al], n2 = f.site_n2[fill_site(f, al)], lo = p.band_lo[al];
    TbBand<C> tb;
    tb.slot = tbl + lane * TBB_PITCH; tb.tw = p.Twb + (int64_t)al * p.NBLK * TWB_PITCH; tb.lo = lo; tb.ckey = -1; tb.edge = p.edge; tb.touched = false; tb.pkey = -1;
    uint32_t *ent = ent_all + (int64_t)al * EW;
    int i = n1, j = n2;
    if (n1 > 0 && n2 > 0) {
        int32_t rv = INT32_MIN, cv = INT32_MIN;
        int rj = 0, ci = 0;
        const int16_t *hr = p.hrow + (int64_t)al * 64, *hc = p.hcolb + (int64_t)al * 64;
        for (int k0 = 0; k0 < B; k0 += 8) {
            const uint4 gr = *reinterpret_cast<const uint4 *>(hr + k0), gc = *reinterpret_cast<const uint4 *>(hc + k0);
            const uint32_t wr[4] = {gr.x, gr.y, gr.z, gr.w}, wc[4] = {gc.x, gc.y, gc.z, gc.w};
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + u;
                const int32_t vr = (int16_t)(wr[u >> 1] >> ((u & 1) * 16)), vc = (int16_t)(wc[u >> 1] >> ((u & 1) * 16));
                const int jr = n1 + lo + k, ic = n2 - lo - k;
                if (jr >= 0 && jr <= n2 && vr >= rv) { rv = vr; rj = jr; }
                if (ic >= 0 && ic < n1 && vc > cv) { cv = vc; ci = ic; }
            }
        }
        if (cv > rv) { i = ci; j = n2; } else { i = n1; j = rj; }
    }
    uint32_t cur = 0;
    if (i < n1) cur = ((uint32_t)(n1 - i) << 10) | ((uint32_t)i << 20);       // the rest of the read: insertion after the window
    int x = n2;
    auto put = [&](uint32_t e) {
        stage[(x & 15) * 64 + lane] = e;
        if ((x & 15) == 0) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                reinterpret_cast<uint4 *>(ent + (x & ~15))[u] = make_uint4(stage[(4 * u) * 64 + lane], stage[(4 * u + 1) * 64 + lane],
                                                                           stage[(4 * u + 2) * 64 + lane], stage[(4 * u + 3) * 64 + lane]);
        }
        x--;
    };
    while (x > j) put(0u);
    int state = -1;
    int a = i + j - 1, kd = j - i - lo;                                // running coordinates of the cell (anti-diagonal, diagonal of the band)
    // one step of the walk without branches but the one around put(): the round-4 form (a chain of if / else per state and border) ran ~200
    // instructions per step once the lanes of a wave sat in different states -- this kernel's time
    auto step = [&]() {
        const bool bi = i == 0, bj = j == 0;
        const uint32_t tc = tb.raw(a, kd);
        int w = (tc & 2u) ? 2 : (int)(tc & 1u);                        // 0 diagonal, 1 deletion (E), 2 insertion (F)
        bool eext = !(tc & 4u), fext = !(tc & 8u);
        w = bi ? 1 : bj ? 2 : w;                                       // row 0 / column 0: a gap to the origin
        eext = bi ? j > 1 : eext;
        fext = bj && !bi ? i > 1 : fext;
        tb.touched |= !bi && !bj && (kd <= tb.edge || kd >= B - 1 - tb.edge);
        const int wm = state < 0 ? w : state;
        const bool mv_d = wm == 0, mv_e = wm == 1;
        if (mv_d || mv_e) put(cur);
        const uint32_t cur_ins = (cur & 0x3ffu) | ((((cur >> 10) & 0x3ffu) + 1u) << 10) | ((uint32_t)(i - 1) << 20);
        cur = mv_d ? (uint32_t)i : mv_e ? 0u : cur_ins;
        const bool ext = mv_e ? eext : fext;
        state = (mv_d || !ext) ? -1 : wm;
        i -= mv_e ? 0 : 1;
        j -= (mv_d || mv_e) ? 1 : 0;
        a -= mv_d ? 2 : 1;
        kd += mv_d ? 0 : mv_e ? -1 : 1;
    };
    while (__any(i > 0 || j > 0)) {                                    // epochs: the lanes that left their line load the next one together
        if (i > 0 && j > 0 && tb.key_akd(a, kd) != tb.ckey) tb.fetch(i, j);
        for (;;) {
            const bool can = (i > 0 || j > 0) && (i == 0 || j == 0 || tb.key_akd(a, kd) == tb.ckey);
            if (!__any(can)) break;
            if (can) step();
        }
    }
    put(cur);
    if (tb.touched) p.redo_list[atomicAdd(p.redo_count, 1)] = al;
}
// both band widths in one launch (blockIdx.y): the 64-diagonal class is a fifteenth of the alignments, on its own a launch of one wave per SIMD
// whose time is the latency of a single traceback
__global__ __launch_bounds__(64) void k_trace_band12(BandArgs p1, BandArgs p2, uint32_t *__restrict__ ent_all, int32_t EW)
{
    __shared__ uint32_t stage[16 * 64];
    __shared__ uint32_t tbl[64 * TBB_PITCH];
    if (blockIdx.y == 0) trace_band_body<1>(p1, ent_all, EW, stage, tbl);
    else trace_band_body<2>(p2, ent_all, EW, stage, tbl);
}

// free-tail end point of every alignment: the best cell of the last row (ties: the larger column) or a cell of the last column that is
// strictly better (ties: the larger row) -- the order k_nw_trace16 scans them in.  16 lanes per alignment over Hlast / hcol (one lane per
// alignment inside the traceback kernel read its ~360 values one after the other: 2.4 of that kernel's 4.5 ms)
__global__ __launch_bounds__(256) void k_end_cells(FillArgs p)
{
    const int al = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
    const int A_live = p.count ? min(*p.count, p.A) : p.A;
    const bool live = al < A_live;
    const int a = live ? al : 0;                                       // slot
    const int ain = p.list ? p.list[a] : a;
    const int n1 = p.n1[ain], n2 = p.site_n2[fill_site(p, ain)];
    int32_t rv = INT32_MIN, rj = 0, cv = INT32_MIN, ci = 0;
    if (live && n1 > 0 && n2 > 0) {
        // four values a load (both rows are 16-byte aligned: hlast_pitch and hcol_pitch are multiples of four words); any split of the
        // indices over the lanes will do, the reduction below orders (value, index) pairs
        const int32_t *hl = p.Hlast + (int64_t)a * hlast_pitch(p.W), *hc = p.hcol + (int64_t)a * hcol_pitch(p.N1);
        for (int j0 = 4 * l; j0 <= n2; j0 += 64) {
            const int4 g = *reinterpret_cast<const int4 *>(hl + j0);
            const int32_t gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + u;
                const int32_t v = j > 0 ? gv[u] : -p.open - (n1 - 1) * p.extend;
                if (j <= n2 && v >= rv) { rv = v; rj = j; }
            }
        }
        for (int i0 = 4 * l; i0 < n1; i0 += 64) {
            const int4 g = *reinterpret_cast<const int4 *>(hc + i0);
            const int32_t gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u;
                const int32_t v = i > 0 ? gv[u] : -p.open - (n2 - 1) * p.extend;
                if (i < n1 && v >= cv) { cv = v; ci = i; }
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const int32_t ov = __shfl_xor(rv, o), oj = __shfl_xor(rj, o), pv = __shfl_xor(cv, o), pi = __shfl_xor(ci, o);
        if (ov > rv || (ov == rv && oj > rj)) { rv = ov; rj = oj; }
        if (pv > cv || (pv == cv && pi > ci)) { cv = pv; ci = pi; }
    }
    if (live && l == 0) p.endcell[al] = (n1 <= 0 || n2 <= 0) ? make_int2(n1, n2) : cv > rv ? make_int2(ci, n2) : make_int2(n1, rj);
}

// The walk of a lane reads one 4-bit code a step, each the end of a chain of dependent loads; the codes of the eight steps a lane q of
// the fill spent on one block are 8 * NWP consecutive words, and a path stays in such a line for several steps (i-- and j-- both lower
// t = i + q).  A lane therefore keeps the line it is in in LDS (odd pitch: no bank conflicts between lanes) and goes to HBM only when it leaves
// it -- in EPOCHS: the lanes that need a new line load it together, then every lane walks on inside its line until none can (a lane that
// fetched on its own whenever it left a line made the whole wave wait at nearly every step: some lane always does).
struct TbLine {
    uint32_t *slot;                                                    // this lane's run of CPL words in LDS (odd pitch)
    int cblk, cq;                                                      // block and fill lane of the cached line (-1: none)
    int q, c;                                                          // fill lane and cell-in-lane of column j, kept in step with j (no division per step)
    __device__ __forceinline__ void set_j(int j, int CPL) { q = j > 0 ? (j - 1) / CPL : 0; c = j > 0 ? (j - 1) % CPL : 0; }
    __device__ __forceinline__ void dec_j(int CPL) { if (--c < 0) { c = CPL - 1; q--; } }
    __device__ __forceinline__ bool has(int i) const { return ((i + q) >> TWB_LOG) == cblk && q == cq; }
    __device__ __forceinline__ void load(const uint32_t *__restrict__ Tw, int64_t arow, int i, int CPL)
    {
        cblk = (i + q) >> TWB_LOG;
        cq = q;
        const uint32_t *src = Tw + tw_run(arow, i + q, q, CPL);
        U4 v[5];                                                         // (whole dwordx4s: up to three words of the next run come along; the buffers end in a pad)
#pragma unroll
        for (int u = 0; u < 5; u++)                                      // CPL <= 17: at most five, all on their way before the first is used (left as an open
            if (4 * u < CPL) v[u] = *reinterpret_cast<const U4 *>(src + 4 * u);     // loop the compiler unrolled it sixteen times: 114 VGPRs instead of 42)
#pragma unroll
        for (int u = 0; u < 5; u++)
            if (4 * u < CPL) { slot[4 * u] = v[u].x; slot[4 * u + 1] = v[u].y; slot[4 * u + 2] = v[u].z; slot[4 * u + 3] = v[u].w; }
    }
    // cell (i, j) of the cached line (the caller checked has(i))
    __device__ __forceinline__ uint32_t code(int i, int CPL, int fmt) const
    {
        const int F = CPL >> 3, R = CPL & 7, st = (i + q) & (TWB - 1);
        int word = st * F + (c >> 3), sh = (c & 7) * 4;
        if (c >= 8 * F) {                                                // one of the step's last R codes
            const int bit = 4 * (R * st + c - 8 * F);
            word = 8 * F + (bit >> 5);
            sh = bit & 31;
        }
        const uint32_t tc = (slot[word] >> sh) & 15u;
        if (fmt == 0) return tc;
        return ((tc & 2u) ? (uint32_t)T_INS : (tc & 1u) ? (uint32_t)T_DEL : (uint32_t)T_DIAG) | ((tc & 4u) ? 0u : (uint32_t)T_EEXT) | ((tc & 8u) ? 0u : (uint32_t)T_FEXT);
    }
};
constexpr int TBL_PITCH = 33;                                          // words per lane (a run is CPL <= 32 words; odd pitch)

// traceback of a free-tail alignment into reference coordinates (nc_msa.hip k_nw_trace16): one lane per alignment.  Entry x of an
// alignment packs, for reference position x (0-based) and the slot BEFORE it (slot n2 = after the last position):
//     bits 0-9  read index aligned to position x, plus 1 (0 = gap)     bits 10-19  length of the insertion in slot x
//     bits 20-29 read index of the insertion's first base
// The walk visits the slots from n2 down to 0 and an entry is final when the walk leaves its slot, so every entry is written once
// (no initialisation pass, no read-modify-write); a lane collects 16 entries in LDS and writes 64-byte runs.
__global__ __launch_bounds__(64) void k_trace16p(FillArgs p, int32_t CPL, int32_t fmt, uint32_t *__restrict__ ent_all, int32_t EW)
{
    __shared__ uint32_t stage[16 * 64];
    __shared__ uint32_t tbl[64 * TBL_PITCH];
    const int al = blockIdx.x * 64 + threadIdx.x;                     // slot (the alignment itself outside list mode)
    if (al >= (p.count ? min(*p.count, p.A) : p.A)) return;
    const int lane = threadIdx.x;
    TbLine tb = {tbl + lane * TBL_PITCH, -1, -1, 0, 0};
    const int ain = p.list ? p.list[al] : al;
    const int n1 = p.n1[ain];
    const int n2 = p.site_n2[fill_site(p, ain)];
    const int64_t arow = (int64_t)al * tw_blocks(p.N1), hrow = (int64_t)al * hcol_pitch(p.N1);
    uint32_t *ent = ent_all + (int64_t)ain * EW;                      // EW: a multiple of 16 entries >= n2 + 1
    int i = n1, j = n2;
    uint32_t cur = 0;                                                  // the entry of slot j being built (position j's read index comes last)
    if (p.endcell) {
        const int2 ec = p.endcell[al];
        i = ec.x;
        j = ec.y;
        if (i < n1) cur = ((uint32_t)(n1 - i) << 10) | ((uint32_t)i << 20);
    } else if (n1 > 0 && n2 > 0) {                                    // free tail: best cell of the last row / last column
        int32_t best = p.Hlast[(int64_t)al * hlast_pitch(p.W) + n2];
        for (int jj = n2 - 1; jj >= 0; jj--) {
            const int32_t v = jj > 0 ? p.Hlast[(int64_t)al * hlast_pitch(p.W) + jj] : -p.open - (n1 - 1) * p.extend;
            if (v > best) { best = v; i = n1; j = jj; }
        }
        for (int ii = n1 - 1; ii >= 0; ii--) {
            const int32_t v = ii > 0 ? p.hcol[hrow + ii] : -p.open - (n2 - 1) * p.extend;
            if (v > best) { best = v; i = ii; j = n2; }
        }
        if (i < n1) cur = ((uint32_t)(n1 - i) << 10) | ((uint32_t)i << 20);       // the rest of the read: insertion after the window
    }
    // slots above the end point (free tail in the reference: j < n2) are empty
    int x = n2;                                                        // slot whose entry is being built
    auto put = [&](uint32_t e) {                                       // entry x is final
        stage[(x & 15) * 64 + lane] = e;
        if ((x & 15) == 0) {
            // entries x .. min(x | 15, n2) of this lane, 64 bytes
#pragma unroll
            for (int u = 0; u < 4; u++)
                reinterpret_cast<uint4 *>(ent + (x & ~15))[u] = make_uint4(stage[(4 * u) * 64 + lane], stage[(4 * u + 1) * 64 + lane],
                                                                           stage[(4 * u + 2) * 64 + lane], stage[(4 * u + 3) * 64 + lane]);
        }
        x--;
    };
    while (x > j) put(0u);                                            // (end point in the last ROW: i == n1, so cur is 0 and stays the entry of slot j)
    int state = -1;
    auto step = [&]() {
        uint32_t t;
        if (i == 0) t = T_DEL | (j > 1 ? T_EEXT : 0);
        else if (j == 0) t = T_INS | (i > 1 ? T_FEXT : 0);
        else t = tb.code(i, CPL, fmt);
        if (state < 0) {
            const int w = t & 3;
            if (w == T_DIAG) {                                        // position j-1 takes read base i-1; slot j is complete
                put(cur);
                cur = (uint32_t)i;                                    // (i - 1) + 1: the read index of position j - 1, entry j - 1
                i--; j--;
                tb.dec_j(CPL);
                return;
            }
            state = w == T_DEL ? 1 : 2;
        }
        if (state == 1) {
            const bool ext = (t & T_EEXT) != 0;
            put(cur);                                                  // reference position j-1 stays a gap
            cur = 0;
            j--;
            tb.dec_j(CPL);
            if (!ext) state = -1;
        } else {
            const bool ext = (t & T_FEXT) != 0;
            cur = (cur & 0x3ffu) | ((((cur >> 10) & 0x3ffu) + 1u) << 10) | ((uint32_t)(i - 1) << 20);     // il[j]++, iq[j] = i - 1
            i--;
            if (!ext) state = -1;
        }
    };
    tb.set_j(j, CPL);
    while (__any(i > 0 || j > 0)) {                                    // epochs
        if (i > 0 && j > 0 && !tb.has(i)) tb.load(p.Tw, arow, i, CPL);
        for (;;) {                                                     // every lane walks on inside its line
            const bool can = (i > 0 || j > 0) && (i == 0 || j == 0 || tb.has(i));
            if (!__any(can)) break;
            if (can) step();
        }
    }
    put(cur);                                                          // slot 0
}

struct TensorArgs {
    int32_t site0, n_sites_g, S, haploid, W, WS;
    int64_t A0;                                 // first alignment of the group
    const int32_t *site_al0, *site_nr, *site_pos, *site_n2;
    const uint8_t *al_member;                   // global
    const uint8_t *win;                         // group-local [A][WS]
    const uint32_t *ent;                        // group-local [A][EW] packed alignment entries (k_trace16p)
    int32_t EW;
    const uint8_t *ref_code;
    int32_t ref_pos0;
    float *x;                                   // global [n_sites][S*5][128][2]
    uint8_t *cns;                               // group-local [n_sites_g * S][CNS_CAP], gap-free consensus
    int32_t *ncns;                              // group-local [n_sites_g * S]
    int16_t *cband;                             // group-local [n_sites_g * S][2]: lowest / highest diagonal of the consensus against the window (or NULL)
    int32_t *err;
};

// HT: the histogram's counter type -- a read set holds at most maxcov reads, so bytes do for maxcov <= 255 (the reference's default is 160)
// and the kernel's LDS drops from 31 to 19 KB: eight workgroups per CU instead of five (3.1 -> 2.x ms; the kernel waits on memory)
template <class HT>
__global__ __launch_bounds__(256) void k_site_tensor(TensorArgs p)
{
    // the three read sets of a site share its alignments (member bits): ONE sweep over the packed entries for the insertion widths and
    // one for the histograms serve all sets (a sweep per set and pass read the site's entries six times: 4.5 GB per chr20-sized contig)
    __shared__ int32_t colv[3][288];
    __shared__ int16_t mxv[3][288];
    __shared__ HT hist[3][CNS_CAP * 4];
    __shared__ uint8_t refrow[CNS_CAP];
    __shared__ uint8_t cnsv[CNS_CAP];
    constexpr int TQ_CAP = 1024;
    __shared__ uint2 qitems[TQ_CAP];                               // insertions of the site's reads: alignment | slot | member bits, length | first base
    __shared__ int32_t s_nq;
    __shared__ int32_t s_ncols[3], s_run;
    // one more read with symbol `sym` in column c of set t: an atomic add on the 32-bit word that holds the counter (no carry: a counter stays <= maxcov)
    auto hist_add = [&](int t, int c, int sym) {
        if (sizeof(HT) == 1) atomicAdd(reinterpret_cast<uint32_t *>(&hist[t][0]) + c, 1u << (8 * sym));
        else atomicAdd(reinterpret_cast<uint32_t *>(&hist[t][0]) + 2 * c + (sym >> 1), 1u << (16 * (sym & 1)));
    };
    __shared__ int32_t wcnt[4], wcntr[4], s_runr, s_dmin, s_dmax;
    const int kl = blockIdx.x, site = p.site0 + kl;
    const int tid = threadIdx.x;
    const int n2 = p.site_n2[site];
    const int S = p.S;
    const int64_t a0 = p.site_al0[site] - p.A0, a1 = p.site_al0[site + 1] - p.A0;
    const uint8_t *mem = p.al_member + p.A0;
    const uint8_t *s2 = p.ref_code + (p.site_pos[site] - p.ref_pos0);
    // ---- ONE sweep over the site's packed entries (round 4 made two: the longest insertion per slot, then the histograms -- the entries are the
    // kernel's traffic, 0.69 GB a sweep per chr20-sized pass): per slot j (thread j; a second turn for the slots past 255 of the 260-base windows) the
    // longest insertion of every set (slot j = before reference position j; slot n2 = after the last), the four base counters of the position's own
    // column in registers, and the insertions onto the block's list.  Neither needs the columns, which come from the insertion widths afterwards.
    // Eight alignments a step, loads first: a load behind a test of the one before it costs a full memory latency each
    if (tid == 0) s_nq = 0;
    __syncthreads();
    uint64_t cntr[2][3] = {{0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int turn = 0; turn < 2; turn++) {
        const int j = tid + 256 * turn;
        if (j > n2) continue;
        int m[3] = {0, 0, 0};
        uint64_t cnt[3] = {0, 0, 0};
        for (int64_t ab = a0; ab < a1; ab += 8) {
            uint32_t en8[8];
            int sym8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) en8[u] = p.ent[min(ab + u, a1 - 1) * p.EW + j];
            // the eight member bytes as two unaligned words, with the other loads, not one by one inside the loop below (the array ends in a pad)
            typedef uint32_t __attribute__((aligned(1))) u32_u;
            const uint32_t mlo = *reinterpret_cast<const u32_u *>(mem + ab), mhi = *reinterpret_cast<const u32_u *>(mem + ab + 4);
#pragma unroll
            for (int u = 0; u < 8; u++) {                             // the base aligned to position j (index clamped: unused when there is none)
                const int qi = (int)(en8[u] & 0x3ffu) - 1;
                sym8[u] = p.win[min(ab + u, a1 - 1) * p.WS + max(qi, 0)];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t a = ab + u;
                if (a >= a1) continue;
                const int mb = (int)(((u < 4 ? mlo : mhi) >> (8 * (u & 3))) & 0xffu);
                const uint32_t en = en8[u];
                if (j < n2 && (en & 0x3ffu) != 0 && sym8[u] < 4) {    // anything else (a read base N) counts as a gap at its column
                    const uint64_t one = 1ull << (16 * sym8[u]);
#pragma unroll
                    for (int t = 0; t < 3; t++)
                        if (mb & (1 << t)) cnt[t] += one;
                }
                // an insertion goes on the block's list: walked here, the whole wave waited for one lane's loads at nearly every alignment
                // (some lane always has one) -- 2/3 of the kernel; from the list every thread takes one insertion
                const int L = (int)((en >> 10) & 0x3ffu);
                if (L > 0) {
#pragma unroll
                    for (int t = 0; t < 3; t++)
                        if (mb & (1 << t)) m[t] = max(m[t], L);        // haploid: one set, member bit 0
                    const int slot = atomicAdd(&s_nq, 1);
                    if (slot < TQ_CAP) qitems[slot] = make_uint2((uint32_t)(a - a0) | ((uint32_t)j << 16) | ((uint32_t)mb << 25), en >> 10);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; t++) { mxv[t][j] = (int16_t)m[t]; cntr[turn][t] = cnt[t]; }
    }
    __syncthreads();
    // ---- column of every slot's position = running sum of the insertion widths + j: scan over the block (2 slots per thread: n2 + 1 <= 288)
    for (int t = 0; t < S; t++) {
        const int j0 = 2 * tid, v0 = j0 <= n2 ? mxv[t][j0] : 0, v1 = j0 + 1 <= n2 ? mxv[t][j0 + 1] : 0;
        int inc = v0 + v1;
        inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
        if ((tid & 63) == 63) wcnt[tid >> 6] = inc;
        __syncthreads();
        int wp = 0;
        for (int w = 0; w < (tid >> 6); w++) wp += wcnt[w];
        const int before = wp + inc - v0 - v1;
        if (j0 <= n2) colv[t][j0] = before + v0 + j0;
        if (j0 + 1 <= n2) colv[t][j0 + 1] = before + v0 + v1 + j0 + 1;
        if (tid == 255) s_ncols[t] = wp + inc + n2;
        __syncthreads();
    }
    bool ok[3];
#pragma unroll
    for (int t = 0; t < 3; t++) ok[t] = t < S && s_ncols[t] <= CNS_CAP;      // a longer set: never with real windows; reported, the caller falls back
    for (int t = 0; t < S; t++)
        if (ok[t])
            for (int c = tid; c < s_ncols[t] * 4; c += 256) hist[t][c] = 0;
    __syncthreads();
    // ---- the position columns' counters go to their places
#pragma unroll
    for (int turn = 0; turn < 2; turn++) {
        const int j = tid + 256 * turn;
        if (j >= n2) continue;
#pragma unroll
        for (int t = 0; t < 3; t++)
            if (t < S && ok[t]) {
                const int cj = colv[t][j];
#pragma unroll
                for (int k = 0; k < 4; k++) hist[t][cj * 4 + k] = (HT)((cntr[turn][t] >> (16 * k)) & 0xffffu);
            }
    }
    // (more insertions than the block's list holds -- > 1024 at one site --: every insertion of the site the round-3 way, straight from the entries)
    if (s_nq > TQ_CAP) {
        for (int j = tid; j <= n2; j += 256) {
            int c0[3];
#pragma unroll
            for (int t = 0; t < 3; t++) c0[t] = t < S ? colv[t][j] - mxv[t][j] : 0;
            for (int64_t a = a0; a < a1; a++) {
                const uint32_t en = p.ent[a * p.EW + j];
                const int L = (int)((en >> 10) & 0x3ffu), mb = mem[a];
                const uint8_t *s1 = p.win + a * p.WS + (int)(en >> 20);
                for (int v = 0; v < L; v++) {
                    const int sym = s1[v];
                    if (sym < 4) {
#pragma unroll
                        for (int t = 0; t < 3; t++)
                            if ((mb & (1 << t)) && ok[t]) hist_add(t, c0[t] + v, sym);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- the inserted bases: one insertion per thread (columns c0 .. c0 + L - 1 of its slot, shared by the reads of a set: atomic adds)
    {
        const int nq = s_nq > TQ_CAP ? 0 : s_nq;
        for (int i = tid; i < nq; i += 256) {
            const uint2 it = qitems[i];
            const int a = (int)(it.x & 0xffffu), j = (int)((it.x >> 16) & 0x1ffu), mb = (int)(it.x >> 25), L = (int)(it.y & 0x3ffu), q0 = (int)(it.y >> 10);
            const uint8_t *s1 = p.win + (a0 + a) * p.WS + q0;
            int c0[3];
#pragma unroll
            for (int t = 0; t < 3; t++) c0[t] = t < S ? colv[t][j] - mxv[t][j] : 0;
            for (int v = 0; v < L; v++) {
                const int sym = s1[v];
                if (sym < 4) {
#pragma unroll
                    for (int t = 0; t < 3; t++)
                        if ((mb & (1 << t)) && ok[t]) hist_add(t, c0[t] + v, sym);
                }
            }
        }
    }
    __syncthreads();
    for (int t = 0; t < S; t++) {
        const int nr = p.site_nr[site * S + t];
        float *X = p.x + ((int64_t)site * S + t) * 5 * 128 * 2;
        const int ncols = s_ncols[t];
        if (!ok[t]) {
            if (tid == 0) { atomicOr(p.err, 2); p.ncns[kl * S + t] = 0; }
            for (int c = tid; c < 128 * 5; c += 256) { X[c * 2] = 0.0f; X[c * 2 + 1] = 0.0f; }
            continue;
        }
        for (int c = tid; c < ncols; c += 256) refrow[c] = 4;
        __syncthreads();
        for (int j = tid; j < n2; j += 256) refrow[colv[t][j]] = s2[j];
        __syncthreads();
        // frequencies, consensus symbol, tensor (:57-71)
        const float tot = (float)nr;
        for (int c = tid; c < max(ncols, 128); c += 256) {
            if (c < ncols) {
                int h[5];
                int sum = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { h[k] = hist[t][c * 4 + k]; sum += h[k]; }
                h[4] = nr - sum;
                float alt[5], best = -1e30f;
                int arg = 0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    alt[k] = (float)h[k] / tot;
                    const float tv = k == 4 ? alt[k] - 0.01f : alt[k];
                    if (tv > best) { best = tv; arg = k; }
                }
                cnsv[c] = (uint8_t)arg;
                if (c < 128) {
                    const int rc = refrow[c];
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const float rf = rc == k ? 1.0f : 0.0f;
                        *reinterpret_cast<float2 *>(X + (k * 128 + c) * 2) = make_float2(alt[k] - rf, rf);
                    }
                }
            } else if (c < 128) {
#pragma unroll
                for (int k = 0; k < 5; k++) *reinterpret_cast<float2 *>(X + (k * 128 + c) * 2) = make_float2(0.0f, 0.0f);
            }
        }
        if (tid == 0) { s_run = 0; s_runr = 0; s_dmin = 0; s_dmax = 0; }
        __syncthreads();
        // consensus with the gap symbols removed (:61-64).  On the way: the diagonals (window columns passed) - (consensus bases written) of the
        // consensus against its window, column by column -- the band of its global alignment in allele_prediction (k_allele_classes)
        uint8_t *out = p.cns + ((int64_t)kl * S + t) * CNS_CAP;
        int dlo = 0, dhi = 0;
        for (int base = 0; base < ncols; base += 256) {
            const int c = base + tid;
            const bool f = c < ncols && cnsv[c] != 4, isr = c < ncols && refrow[c] != 4;
            const uint64_t bm = __ballot(f), br = __ballot(isr);
            if ((tid & 63) == 0) { wcnt[tid >> 6] = __popcll(bm); wcntr[tid >> 6] = __popcll(br); }
            __syncthreads();
            int wp = s_run, totw = 0, wr = s_runr, totr = 0;
            for (int w = 0; w < 4; w++) {
                if (w < (tid >> 6)) { wp += wcnt[w]; wr += wcntr[w]; }
                totw += wcnt[w];
                totr += wcntr[w];
            }
            const uint64_t below = (1ull << (tid & 63)) - 1, upto = below | (1ull << (tid & 63));
            if (f) out[wp + __popcll(bm & below)] = cnsv[c];
            if (c < ncols) {
                const int d = (wr + __popcll(br & upto)) - (wp + __popcll(bm & upto));
                dlo = min(dlo, d);
                dhi = max(dhi, d);
            }
            __syncthreads();
            if (tid == 0) { s_run += totw; s_runr += totr; }
            __syncthreads();
        }
        if (p.cband) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { dlo = min(dlo, __shfl_xor(dlo, o)); dhi = max(dhi, __shfl_xor(dhi, o)); }
            if ((tid & 63) == 0) { atomicMin(&s_dmin, dlo); atomicMax(&s_dmax, dhi); }
            __syncthreads();
            if (tid == 0) { p.cband[(kl * S + t) * 2] = (int16_t)s_dmin; p.cband[(kl * S + t) * 2 + 1] = (int16_t)s_dmax; }
        }
        if (tid == 0) p.ncns[kl * S + t] = s_run;
        __syncthreads();
    }
}

// allele_prediction on the packed traceback of a GLOBAL alignment of the consensus (s1) against the window (nc_msa.hip k_allele_trace16).
// C = 0: the full-matrix codes of k_fill16q (all alignments, or bp.f.list's); C = 1 / 2: the banded codes of k_fill_band<C> over bp.list --
// an alignment whose path touches an edge diagonal of its band joins bp.redo_list (the caller runs those on the full matrix) and writes nothing
template <int C>
__device__ __forceinline__ void allele_trace_body(const BandArgs &bp, int32_t CPL, int32_t fmt, const int32_t *__restrict__ site_type, int32_t win_size,
                                                  int16_t *__restrict__ runs, int32_t *__restrict__ ref_len, int32_t *__restrict__ alt_len, uint32_t *tbl)
{
    const FillArgs &p = bp.f;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= (C ? *bp.count : p.count ? min(*p.count, p.A) : p.A)) return;
    const int al = C ? bp.list[idx] : p.list ? p.list[idx] : idx;
    TbLine tb = {tbl + threadIdx.x * TBL_PITCH, -1, -1, 0, 0};
    TbBand<(C ? C : 1)> tbb;
    tbb.slot = tbl + threadIdx.x * TBB_PITCH; tbb.tw = C ? bp.Twb + (int64_t)al * bp.NBLK * TWB_PITCH : nullptr; tbb.lo = C ? (int)bp.band_lo[al] : 0;
    tbb.ckey = -1; tbb.edge = bp.edge; tbb.touched = false; tbb.pkey = -1;
    const int site = fill_site(p, al);
    const uint8_t *s1 = p.s1 + (int64_t)al * p.s1_stride;
    const int n1 = p.n1[al];
    const uint8_t *s2 = p.ref_code + (p.site_pos[site] - p.ref_pos0);
    const int n2 = p.site_n2[site];
    if (n1 <= 0) {                                                   // (an empty consensus cannot happen: every column of a set has a symbol or a gap)
        ref_len[al] = -1;
        alt_len[al] = -1;
        return;
    }
    const int64_t arow = p.arow[al];
    const int run_cap = n1 + n2 + 2;
    int16_t *rop = runs + 2 * (TWB * arow + (int64_t)al * (p.W + 1)), *rcn = rop + run_cap;  // runs in REVERSE alignment order (TWB * blocks >= n1 + 1)
    int nr = 0, last_op = -1, last_cnt = 0;                            // the open run lives in registers: one store pair per run, no read-modify-write
    auto push = [&](int op) {
        if (op == last_op) last_cnt++;
        else {
            if (last_op >= 0 && nr < run_cap) { rop[nr] = (int16_t)last_op; rcn[nr] = (int16_t)last_cnt; nr++; }
            last_op = op;
            last_cnt = 1;
        }
    };
    int i = n1, j = n2, state = -1;
    int32_t path_score = 0;                                            // (banded route) the score of the path walked: the certificate below compares it with what any path outside the band can reach
    auto step = [&]() {
        uint32_t t;
        if (i == 0) t = T_DEL | (j > 1 ? T_EEXT : 0);
        else if (j == 0) t = T_INS | (i > 1 ? T_FEXT : 0);
        else t = C ? tbb.code(i, j) : tb.code(i, CPL, fmt);
        if (state < 0) {
            const int w = t & 3;
            if (w == T_DIAG) {
                const bool eq = s1[i - 1] == s2[j - 1];
                push(eq ? 7 : 8);
                path_score += eq ? p.match : p.mismatch;
                i--; j--; tb.dec_j(CPL);
                return;
            }
            state = w == T_DEL ? 1 : 2;
        }
        if (state == 1) {
            push(2);
            const bool ext = (t & T_EEXT) != 0;
            j--;
            tb.dec_j(CPL);
            path_score -= ext ? p.extend : p.open;                     // (walked backwards: the step that is not an extension is the gap's first base)
            if (!ext) state = -1;
        } else {
            push(1);
            const bool ext = (t & T_FEXT) != 0;
            i--;
            path_score -= ext ? p.extend : p.open;
            if (!ext) state = -1;
        }
    };
    tb.set_j(j, CPL);
    while (__any(i > 0 || j > 0)) {                                    // epochs: see TbLine
        if (C) { if (i > 0 && j > 0 && !tbb.has(i, j)) tbb.fetch(i, j); }
        else if (i > 0 && j > 0 && !tb.has(i)) tb.load(p.Tw, arow, i, CPL);
        for (;;) {
            const bool can = (i > 0 || j > 0) && (i == 0 || j == 0 || (C ? tbb.has(i, j) : tb.has(i)));
            if (!__any(can)) break;
            if (can) step();
        }
    }
    if (C) {
        // Is the banded optimum THE optimum?  A path that leaves the band [lo, lo + B) reaches diagonal d_out = lo - 1 or lo + B.  From diagonal 0 to d_out and
        // on to the corner's diagonal D = n2 - n1 it spends at least |d_out| gap bases on one string and |d_out - D| on the other -- two gap runs, and that
        // many bases of either string that pair with nothing -- so it scores at most
        //     match x min(n1 - gi, n2 - gj) - (open + (gj - 1) ext) - (open + (gi - 1) ext),     gj / gi = the gap bases in the window / the consensus.
        // A banded path that scores MORE is optimal over the full matrix, ties included (a co-optimal path through cells outside the band would be a
        // path that leaves the band and reaches the optimum).  A consensus is its window with a few indels applied: the bound holds for all but a few per
        // ten thousand sets; the rest, and the paths that touch an edge diagonal, go to the full matrix.  (For the star alignment of 8 % error reads
        // against the window the same bound proves nothing: that band stays part of the aligner's definition, section 11.4.)
        constexpr int Bw = 32 * (C ? C : 1);
        const int D = n2 - n1, lo = tbb.lo;
        // (0 and D are inside the band: k_allele_classes.)  Above the band: the diagonal rises by d_out window-only bases and falls d_out - D
        // consensus-only ones; below: it falls -d_out and rises D - d_out.
        auto ub = [&](int d_out) -> int32_t {
            const int rise = d_out > 0 ? d_out : D - d_out, fall = d_out > 0 ? d_out - D : -d_out;
            const int pairs = min(n1 - fall, n2 - rise);
            if (pairs < 0) return INT32_MIN;                               // no path gets there
            return p.match * pairs - (p.open + (rise - 1) * p.extend) - (p.open + (fall - 1) * p.extend);
        };
        const bool certified = path_score > ub(lo - 1) && path_score > ub(lo + Bw);
        if (tbb.touched || !certified) {
            bp.redo_list[atomicAdd(bp.redo_count, 1)] = al;
            return;
        }
    }
    if (last_op >= 0 && nr < run_cap) { rop[nr] = (int16_t)last_op; rcn[nr] = (int16_t)last_cnt; nr++; }
    bool indel = false, mm_before = false;
    int32_t rc7 = 0, rc8 = 0, rc2 = 0, ac7 = 0, ac8 = 0, ac1 = 0, mm_after = 0;
    const int32_t mr = site_type[site] == 0 ? max(10, win_size) : 10;       // max_range {0: max(10, win_size), 1: 10}
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
    ref_len[al] = out_r;
    alt_len[al] = out_a;
}
template <int C>
__global__ __launch_bounds__(64) void k_allele_trace16p(BandArgs bp, int32_t CPL, int32_t fmt, const int32_t *__restrict__ site_type, int32_t win_size,
                                                        int16_t *__restrict__ runs, int32_t *__restrict__ ref_len, int32_t *__restrict__ alt_len)
{
    __shared__ uint32_t tbl[64 * TBL_PITCH];
    allele_trace_body<C>(bp, CPL, fmt, site_type, win_size, runs, ref_len, alt_len, tbl);
}
__global__ __launch_bounds__(64) void k_allele_trace_b12(BandArgs b1, BandArgs b2, int32_t CPL, const int32_t *__restrict__ site_type, int32_t win_size,
                                                         int16_t *__restrict__ runs, int32_t *__restrict__ ref_len, int32_t *__restrict__ alt_len)
{
    __shared__ uint32_t tbl[64 * TBL_PITCH];
    if (blockIdx.y == 0) allele_trace_body<1>(b1, CPL, 1, site_type, win_size, runs, ref_len, alt_len, tbl);
    else allele_trace_body<2>(b2, CPL, 1, site_type, win_size, runs, ref_len, alt_len, tbl);
}

// band of a GLOBAL alignment of a consensus (n1 bases) against its window (n2): the consensus is the window with the set's indels applied, and
// k_site_tensor noted the diagonals its columns run on (cband); classes and lists as k_windows makes them for the star alignment
__global__ __launch_bounds__(256) void k_allele_classes(FillArgs p, const int16_t *__restrict__ cband, int32_t margin, int32_t max_sum, int8_t *__restrict__ band_lo, int32_t *__restrict__ list1,
                                                       int32_t *__restrict__ list2, int32_t *__restrict__ listF, int32_t *__restrict__ counts)
{
    const int al = blockIdx.x * 256 + threadIdx.x;
    int cls = -1;
    if (al < p.A) {
        const int n1 = p.n1[al], n2 = p.site_n2[fill_site(p, al)];
        const int dend = n2 - n1, dmin = min(min(0, dend), (int)cband[2 * al]), dmax = max(max(0, dend), (int)cband[2 * al + 1]), w = dmax - dmin;
        cls = (n1 <= 0 || n1 + n2 > max_sum) ? 2 : w + 2 * margin <= 31 ? 0 : w + 2 * margin <= 63 ? 1 : 2;      // (an empty consensus: the full route reports it)
        const int B = cls == 0 ? 32 : 64;
        int lo = dmin - ((B - 1 - w) >> 1);
        lo -= lo & 1;
        band_lo[al] = (int8_t)(cls == 2 ? 0 : lo);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const unsigned long long m = __ballot(cls == c);
        if (!m) continue;
        const int lead = __ffsll((long long)m) - 1, ln = threadIdx.x & 63;
        int base = 0;
        if (ln == lead) base = atomicAdd(counts + c, __popcll(m));
        base = __shfl(base, lead);
        int32_t *lst = c == 0 ? list1 : c == 1 ? list2 : listF;
        if (cls == c) lst[base + __popcll(m & ((1ull << ln) - 1ull))] = al;
    }
}

// One workgroup, every thread a contiguous chunk of ceil(n / 1024) items: local sum -> block scan of the 1,024 sums -> local scan.  (The
// strided form -- 1,024 items per round, two barriers per round -- took 0.3 ms for 30 k items on the critical path of every group.)
// traceback blocks of the allele alignments: arow[a] = sum_{b<a} tw_blocks(n1[b]); arow[n] = total
__global__ __launch_bounds__(1024) void k_scan_rows(const int32_t *__restrict__ n1, int32_t n, int64_t *__restrict__ arow, int32_t *__restrict__ total_mbox)
{
    __shared__ int wsum[16];
    const int per = (n + 1023) / 1024, i0 = threadIdx.x * per, i1 = min(n, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; i++) local += tw_blocks(n1[i]);
    int tot;
    const int inc = block_scan(local, wsum, tot);
    long long run = inc - local;
    for (int i = i0; i < i1; i++) {
        arow[i] = run;
        run += tw_blocks(n1[i]);
    }
    if (threadIdx.x == 0) {
        arow[n] = tot;
        total_mbox[0] = tot & 0x7fffffff;
        total_mbox[1] = 0;
    }
}

// ALT prefixes of the group's sets appended to the pool: pool_base[0] = bytes used so far (updated)
__global__ __launch_bounds__(1024) void k_alt_offsets(const int32_t *__restrict__ alt_len, int32_t n, long long *__restrict__ pool_base,
                                                      int64_t *__restrict__ off /* [n] */)
{
    __shared__ int wsum[16];
    const int per = (n + 1023) / 1024, i0 = threadIdx.x * per, i1 = min(n, i0 + per);
    int local = 0;
    for (int i = i0; i < i1; i++) local += max(alt_len[i], 0);
    int tot;
    const int inc = block_scan(local, wsum, tot);
    long long run = pool_base[0] + inc - local;
    __syncthreads();                                               // every thread has read pool_base before it is updated
    for (int i = i0; i < i1; i++) {
        off[i] = run;
        run += max(alt_len[i], 0);
    }
    if (threadIdx.x == 0) pool_base[0] += tot;
}

__global__ __launch_bounds__(256) void k_alt_copy(const uint8_t *__restrict__ cns, const int32_t *__restrict__ alt_len, const int64_t *__restrict__ off,
                                                  int32_t n, uint8_t *__restrict__ pool, int64_t pool_cap, int32_t *__restrict__ err)
{
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a >= n) return;
    const int L = alt_len[a];
    if (L <= 0) return;
    const int64_t o = off[a];
    if (o + L > pool_cap) { if (lane == 0) atomicOr(err, 4); return; }
    for (int i = lane; i < L; i += 64) pool[o + i] = cns[(int64_t)a * CNS_CAP + i];
}

}   // namespace

// ------------------------------------------------------------------------------------------------------------ host side
struct nc_pipe_state {
    bool planned = false, ran = false;
    nc_readpack pack;
    nc_indel_reads rd;
    const uint8_t *ref_code = nullptr;
    int32_t ref_pos0 = 0, ref_len = 0;
    int64_t chrom_len = 0;
    int32_t window_after = 0, maxcov = 0, mincov = 0, win_size = 0, haploid = 0, S = 3;
    int32_t n_chunks = 0, n_anchor = 0, n_sites = 0;
    int64_t n_al = 0;
    int32_t *al0_pin = nullptr;             // page-locked: first alignment of every site (+ total), read by the host to cut the groups
    int64_t tw_budget = 0;                  // bytes of traceback codes per group (set at the first run from the free device memory)
    size_t al0_cap = 0;
    DevBuf pc, seg_pos, seg_type, cnt, off, anc_pos, anc_type, anc_chunk, kept, nuniq, site_of, al_of;
    DevBuf site_pos, site_chunk, site_type, site_phase, site_al0, site_nr, site_n2, al_read, al_site, al_member, al_ev;
    bool have_al_ev = false;
    struct GroupBufs {
        DevBuf win, n1, tw, hlast, hcol, endc, trace, cns, ncns, arow, alt_off;
        DevBuf band_lo, lists, counts, twb, hrow, hcolb, cband;                // banded star alignment: per-alignment band, class lists, codes, last row / column
    } gb[2];                                // two sets: group g+1 is aligned while g is reduced
    int32_t band_mode = -1, band_margin_v = 0;   // nc_indel_sites_band: -1 = the environment's setting
    int64_t band_stats[6] = {0, 0, 0, 0, 0, 0};   // of the last run: alignments on 32 / 64 diagonals, on the full matrix by width, re-run after an edge touch
    DevBuf tw2, runs, rlen, alen, alt_pool, misc;
    DevBuf part_a, part_b;                   // partial sums of the two-launch scans (plan / stream A; stream B)
    DevBuf ab_lo, ab_lists, ab_counts, ab_twb;             // banded allele alignments (stage_b2; one group at a time on stream B)
    hipStream_t sB = nullptr;                // second stream: traceback / tensors / alleles of group g beside the alignment fill of g + 1
    hipEvent_t evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, ev_join = nullptr;
    int64_t alt_pool_cap = 0;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t scoring[4] = {25, 1, 20, -10};         // star alignment: gap open, gap extend, match, mismatch (nc_indel_sites_scoring)
    float stage_ms[6] = {0, 0, 0, 0, 0, 0};
    int64_t cells[2] = {0, 0};
};

void nc_pipe_destroy(nc_ctx *ctx)
{
    nc_pipe_state *s = ctx->pipe;
    if (!s) return;
    DevBuf *bufs[] = {&s->pc, &s->seg_pos, &s->seg_type, &s->cnt, &s->off, &s->anc_pos, &s->anc_type, &s->anc_chunk, &s->kept, &s->nuniq, &s->site_of,
                      &s->al_of, &s->site_pos, &s->site_chunk, &s->site_type, &s->site_phase, &s->site_al0, &s->site_nr, &s->site_n2, &s->al_read,
                      &s->al_site, &s->al_member, &s->al_ev, &s->tw2, &s->runs, &s->rlen, &s->alen, &s->alt_pool, &s->misc, &s->part_a, &s->part_b, &s->ab_lo, &s->ab_lists, &s->ab_counts, &s->ab_twb,
                      &s->gb[0].win, &s->gb[0].n1, &s->gb[0].tw, &s->gb[0].hlast, &s->gb[0].hcol, &s->gb[0].endc, &s->gb[0].trace, &s->gb[0].cns, &s->gb[0].ncns,
                      &s->gb[0].arow, &s->gb[0].alt_off, &s->gb[0].band_lo, &s->gb[0].lists, &s->gb[0].counts, &s->gb[0].twb, &s->gb[0].hrow, &s->gb[0].hcolb, &s->gb[0].cband, &s->gb[1].cband,
                      &s->gb[1].band_lo, &s->gb[1].lists, &s->gb[1].counts, &s->gb[1].twb, &s->gb[1].hrow, &s->gb[1].hcolb, &s->gb[1].win, &s->gb[1].n1, &s->gb[1].tw, &s->gb[1].hlast, &s->gb[1].hcol, &s->gb[1].endc, &s->gb[1].trace,
                      &s->gb[1].cns, &s->gb[1].ncns, &s->gb[1].arow, &s->gb[1].alt_off};
    for (DevBuf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
    for (auto &e : s->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : s->evA) if (e) (void)hipEventDestroy(e);
    for (auto &e : s->evB) if (e) (void)hipEventDestroy(e);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->sB) (void)hipStreamDestroy(s->sB);
    if (s->al0_pin) (void)hipHostFree(s->al0_pin);
    delete s;
    ctx->pipe = nullptr;
}

static int cpl_for(int n2) { return n2 <= 64 ? 4 : n2 <= 128 ? 8 : n2 <= 176 ? 11 : n2 <= 272 ? 17 : 0; }

static bool packed_fill() { static const bool on = !getenv("NC_PIPE_FILL32"); return on; }
// banded star alignment (k_fill_band): on unless NC_PIPE_BAND=0 or the 32-bit fill is forced; NC_PIPE_BAND_MARGIN = diagonals kept free on
// either side of the range the read's CIGAR covers (default 6)
static bool band_on() { const char *e = getenv("NC_PIPE_BAND"); return !(e && atoi(e) == 0); }                 // (read at every run: tests switch it)
static int band_margin() { const char *e = getenv("NC_PIPE_BAND_MARGIN"); return e ? std::max(1, std::min(15, atoi(e))) : 6; }

__global__ void k_band_stats(const int32_t *__restrict__ counts, long long *__restrict__ acc)
{
    acc[0] += counts[0]; acc[1] += counts[1]; acc[2] += counts[3]; acc[3] += counts[2] - counts[3];
}

static void launch_fill(nc_ctx *ctx, hipStream_t st, int CPL, const FillArgs &fa)
{
    (void)ctx;
    if (packed_fill()) {
        const dim3 gq((unsigned)((fa.A + 7) / 8));
        if (CPL == 4) hipLaunchKernelGGL(k_fill16q<4>, gq, dim3(64), 0, st, fa);
        else if (CPL == 8) hipLaunchKernelGGL(k_fill16q<8>, gq, dim3(64), 0, st, fa);
        else if (CPL == 11) hipLaunchKernelGGL(k_fill16q<11>, gq, dim3(64), 0, st, fa);
        else hipLaunchKernelGGL(k_fill16q<17>, gq, dim3(64), 0, st, fa);
        return;
    }
    const dim3 gr((unsigned)((fa.A + 3) / 4));
    if (CPL == 4) hipLaunchKernelGGL(k_fill16p<4>, gr, dim3(64), 0, st, fa);
    else if (CPL == 8) hipLaunchKernelGGL(k_fill16p<8>, gr, dim3(64), 0, st, fa);
    else if (CPL == 11) hipLaunchKernelGGL(k_fill16p<11>, gr, dim3(64), 0, st, fa);
    else hipLaunchKernelGGL(k_fill16p<17>, gr, dim3(64), 0, st, fa);
}

extern "C" int nc_indel_sites_plan(nc_ctx *ctx, const nc_readpack *pack, const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                                   int64_t chrom_len, const nc_indel_reads *reads, const uint8_t *excl_dev, int32_t n_chunks,
                                   const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *prm, int32_t window_after,
                                   int32_t maxcov, int32_t *n_sites, int64_t *n_alignments)
{
    if (!ctx) return NC_ERR_ARG;
    if (!pack || !ref_code_dev || !reads || !prm || !n_sites || !n_alignments || n_chunks < 0 || (n_chunks && (!starts || !ends)) || window_after < 1 ||
        maxcov < 1 || chrom_len < 1)
        return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_plan: bad argument");
    const bool impute = prm->impute && !prm->haploid;                 // (generate_indel_pileups.py:278: the diploid function only)
    if (impute && (!reads->ins_off || !reads->ins_bases)) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_plan: impute_indel_phase needs the inserted bases");
    if (cpl_for(window_after + 1) == 0) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites_plan: windows longer than 271 bases");
    nc_indel_events ev;
    ev.n_reads = reads->n_reads; ev.ev_off = reads->ev_off; ev.ev_pos = reads->ev_pos; ev.ev_len = reads->ev_len; ev.read_hap = reads->read_hap;
    NC_TRY(nc_indel_check(ctx, pack, &ev, prm, "nc_indel_sites_plan"));
    NC_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->pipe) ctx->pipe = new (std::nothrow) nc_pipe_state();
    nc_pipe_state *s = ctx->pipe;
    if (!s) return NC_ERR_NOMEM;
    s->planned = s->ran = false;
    s->pack = *pack; s->rd = *reads; s->ref_code = ref_code_dev; s->ref_pos0 = ref_pos0; s->ref_len = ref_len; s->chrom_len = chrom_len;
    s->window_after = window_after; s->maxcov = maxcov; s->mincov = prm->mincov; s->win_size = prm->win_size; s->haploid = prm->haploid ? 1 : 0;
    s->S = prm->haploid ? 1 : 3;
    s->n_chunks = n_chunks; s->n_sites = 0; s->n_al = 0; s->n_anchor = 0;
    *n_sites = 0;
    *n_alignments = 0;
    for (auto &m : s->stage_ms) m = 0;
    s->cells[0] = s->cells[1] = 0;
    if (n_chunks == 0) { s->planned = true; return NC_OK; }
    for (int32_t c = 0; c < n_chunks; c++) {
        if (ends[c] < starts[c]) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_plan: chunk %d has end < start", c);
        if (c && (starts[c] < starts[c - 1] || ends[c] < ends[c - 1])) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_plan: chunks must ascend");
    }
    const bool timing = ctx->timing == 1;
    if (timing)
        for (auto &e : s->ev) if (!e) NC_HIP(ctx, hipEventCreate(&e));
    if (timing) NC_HIP(ctx, hipEventRecord(s->ev[0], ctx->stream));
    // ---- anchors of every chunk
    std::vector<PipeChunk> pcs((size_t)n_chunks);
    int64_t seg_total = 0;
    for (int32_t c = 0; c < n_chunks; c++) {
        PipeChunk &k = pcs[(size_t)c];
        k.lo = starts[c] < 1 ? 1 : starts[c];
        k.hi = ends[c];
        k.ncol = k.hi - k.lo + 1;
        k.a_lo = std::max(0, starts[c] - 10 - prm->win_size);
        k.coloff = 0;
        k.seg0 = (int32_t)seg_total;
        k.id = c;
        const int64_t cap = k.ncol / 11 + 2;
        if (cap > PICK_CAP) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites_plan: chunk %d is longer than the anchor buffer covers", c);
        seg_total += cap;
        if (seg_total > INT32_MAX / 2) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites_plan: too many columns in one call");
    }
    NC_TRY(nc_ensure(ctx, s->pc, (size_t)n_chunks * sizeof(PipeChunk)));
    NC_TRY(nc_ensure(ctx, s->seg_pos, (size_t)seg_total * 4));
    NC_TRY(nc_ensure(ctx, s->seg_type, (size_t)seg_total));
    NC_TRY(nc_ensure(ctx, s->cnt, ((size_t)n_chunks + 1) * 4));
    NC_TRY(nc_ensure(ctx, s->off, ((size_t)n_chunks + 2) * 4));
    NC_TRY(nc_ensure(ctx, s->misc, 256));
    int32_t *err = (int32_t *)s->misc.p;                              // [0] error bits, [2..3] row total mailbox, [4..5] cells, [6..7] banded cells, [8..9] alt pool bytes, [16..23] band classes
    NC_HIP(ctx, hipMemsetAsync(s->misc.p, 0, 256, ctx->stream));
    ImpArgs imp;
    memset(&imp, 0, sizeof imp);
    imp.tile_off = pack->tile_off; imp.tile_ent = pack->tile_ent; imp.tile_pos0 = pack->tile_pos0; imp.tile_size = pack->tile_size; imp.n_tiles = pack->n_tiles;
    imp.codes = pack->codes; imp.slot_off = reads->slot_off; imp.n_reads = reads->n_reads;
    imp.ev_off = reads->ev_off; imp.ev_pos = reads->ev_pos; imp.ev_len = reads->ev_len; imp.ins_off = reads->ins_off; imp.ins_bases = reads->ins_bases;
    imp.mincov = prm->mincov;
    int32_t c0 = 0;
    std::vector<IndelChunk> ck;
    while (c0 < n_chunks) {
        int32_t used = 0;
        const IndelChunk *ck_dev = nullptr;
        const int8_t *ctype = nullptr;
        NC_TRY(nc_indel_scan_group_launch(ctx, pack, &ev, excl_dev, n_chunks - c0, starts + c0, ends + c0, prm, &used, ck, &ck_dev, &ctype,
                                          ev.n_reads == reads->n_reads ? reads->slot_off : nullptr, err, reads->rd_start, reads->rd_end, c0 > 0));
        for (int32_t k = 0; k < used; k++) pcs[(size_t)(c0 + k)].coloff = ck[(size_t)k].coloff;
        NC_TRY(nc_h2d_pieces(ctx, (PipeChunk *)s->pc.p + c0, pcs.data() + c0, (size_t)used * sizeof(PipeChunk), ctx->stream));
        if (impute) {                                                 // the read grouping of every col_type-2 column: 3 (an anchor) or -1
            int32_t maxcol = 1;
            for (int32_t k = 0; k < used; k++) maxcol = std::max(maxcol, pcs[(size_t)(c0 + k)].ncol);
            hipLaunchKernelGGL(k_impute_flags, dim3((unsigned)used, (unsigned)((maxcol + 4095) / 4096)), dim3(256), 0, ctx->stream, (const PipeChunk *)s->pc.p + c0,
                               const_cast<int8_t *>(ctype), imp, err);
        }
        hipLaunchKernelGGL(k_pick, dim3(used), dim3(64), 0, ctx->stream, (const PipeChunk *)s->pc.p + c0, ctype, prm->win_size, (int32_t *)s->seg_pos.p,
                           (int8_t *)s->seg_type.p, (int32_t *)s->cnt.p, err);
        NC_HIP(ctx, hipGetLastError());
        c0 += used;                                                   // (the next group's K7 reuses the workspace in stream order)
    }
    NC_TRY(nc_ensure(ctx, s->part_a, 8 * SC_PARTS));                  // (sized once: the scans never re-allocate between launches of a pass; SC_PARTS x SC_TILE
    NC_TRY(nc_ensure(ctx, s->part_b, 8 * SC_PARTS));                  // = 268 M elements per scan, beyond any array of a pass)
    NC_TRY((scan_launch<SC_PLAIN, int32_t>(ctx, ctx->stream, s->part_a, (const int32_t *)s->cnt.p, n_chunks, 0, (int32_t *)s->off.p, true, nullptr, nullptr)));
    // counts the host waits for travel by copy kernel into the context's page-locked mailbox (a hipMemcpyAsync of either direction
    // queues behind a contig's upload in flight on this platform: DESIGN.md section 2)
    volatile int32_t *mb = ctx->mbox + 32;
    NC_TRY(nc_d2h(ctx, ctx->mbox + 32, (int32_t *)s->off.p + n_chunks, 4, ctx->stream));
    NC_TRY(nc_d2h(ctx, ctx->mbox + 33, err, 4, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t na = mb[0], errh = mb[1];
    if (errh & 1) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites_plan: anchor buffer of a chunk overflowed");
    if (errh & 8) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites_plan: more than 16,000 alignments over one tile of the index");
    if (errh & 16) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites_plan: impute_indel_phase on a column deeper than %d reads", IMP_CAP);
    s->n_anchor = na;
    s->planned = true;
    if (na == 0) {
        if (timing) { NC_HIP(ctx, hipEventRecord(s->ev[1], ctx->stream)); }
        return NC_OK;
    }
    NC_TRY(nc_ensure(ctx, s->anc_pos, (size_t)na * 4));
    NC_TRY(nc_ensure(ctx, s->anc_type, (size_t)na));
    NC_TRY(nc_ensure(ctx, s->anc_chunk, (size_t)na * 4));
    NC_TRY(nc_ensure(ctx, s->kept, ((size_t)na + 1) * 4));
    NC_TRY(nc_ensure(ctx, s->nuniq, ((size_t)na + 1) * 4));
    NC_TRY(nc_ensure(ctx, s->site_of, ((size_t)na + 1) * 4));
    NC_TRY(nc_ensure(ctx, s->al_of, ((size_t)na + 1) * 4));
    hipLaunchKernelGGL(k_flatten, dim3(n_chunks), dim3(256), 0, ctx->stream, (const PipeChunk *)s->pc.p, (const int32_t *)s->seg_pos.p,
                       (const int8_t *)s->seg_type.p, (const int32_t *)s->cnt.p, (const int32_t *)s->off.p, (int32_t *)s->anc_pos.p,
                       (int8_t *)s->anc_type.p, (int32_t *)s->anc_chunk.p);
    SetArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.tile_off = pack->tile_off; sa.tile_ent = pack->tile_ent; sa.tile_pos0 = pack->tile_pos0; sa.tile_size = pack->tile_size; sa.n_tiles = pack->n_tiles;
    sa.ref_code = ref_code_dev; sa.ref_pos0 = ref_pos0; sa.ref_len = ref_len; sa.chrom_len = chrom_len;
    sa.window_after = window_after; sa.maxcov = maxcov; sa.mincov = prm->mincov; sa.haploid = s->haploid;
    sa.slot_off = reads->slot_off; sa.read_ps = reads->read_ps; sa.n_reads = reads->n_reads;
    const bool have_ent = ctx->indel_ent_of == (const void *)pack->tile_ent && ctx->indel_ent_read.p && !getenv("NC_PIPE_NO_EV_CURSORS");
    if (have_ent) {
        sa.ent_read = (const int32_t *)ctx->indel_ent_read.p; sa.ent_cur = sa.ent_read + pack->n_entries; sa.ev_off = reads->ev_off; sa.spt = ctx->indel_ent_spt;
    }
    sa.n_anchor = na; sa.anc_pos = (const int32_t *)s->anc_pos.p; sa.anc_chunk = (const int32_t *)s->anc_chunk.p; sa.anc_type = (const int8_t *)s->anc_type.p;
    sa.kept = (int32_t *)s->kept.p; sa.nuniq = (int32_t *)s->nuniq.p;
    sa.imp = imp; sa.imp.ent_read = sa.ent_read; sa.err = err;
    if (impute) hipLaunchKernelGGL((k_sets<false, true>), dim3((na + 3) / 4), dim3(256), 0, ctx->stream, sa);
    else hipLaunchKernelGGL((k_sets<false, false>), dim3((na + 3) / 4), dim3(256), 0, ctx->stream, sa);
    NC_TRY((scan_launch<SC_PLAIN, int32_t>(ctx, ctx->stream, s->part_a, (const int32_t *)s->kept.p, na, 0, (int32_t *)s->site_of.p, true, nullptr, nullptr)));
    NC_TRY((scan_launch<SC_PLAIN, int32_t>(ctx, ctx->stream, s->part_a, (const int32_t *)s->nuniq.p, na, 0, (int32_t *)s->al_of.p, true, nullptr, nullptr)));
    NC_HIP(ctx, hipGetLastError());
    NC_TRY(nc_d2h(ctx, ctx->mbox + 34, (int32_t *)s->site_of.p + na, 4, ctx->stream));
    NC_TRY(nc_d2h(ctx, ctx->mbox + 35, (int32_t *)s->al_of.p + na, 4, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t ns = mb[2], nal = mb[3];
    s->n_sites = ns;
    s->n_al = nal;
    *n_sites = ns;
    *n_alignments = nal;
    if (ns == 0) {
        if (timing) { NC_HIP(ctx, hipEventRecord(s->ev[1], ctx->stream)); }
        return NC_OK;
    }
    const size_t NS = (size_t)ns;
    NC_TRY(nc_ensure(ctx, s->site_pos, NS * 4));
    NC_TRY(nc_ensure(ctx, s->site_chunk, NS * 4));
    NC_TRY(nc_ensure(ctx, s->site_type, NS * 4));
    NC_TRY(nc_ensure(ctx, s->site_phase, NS * 4));
    NC_TRY(nc_ensure(ctx, s->site_al0, (NS + 1) * 4));
    NC_TRY(nc_ensure(ctx, s->site_nr, NS * 3 * 4));
    NC_TRY(nc_ensure(ctx, s->site_n2, NS * 4));
    NC_TRY(nc_ensure(ctx, s->al_read, (size_t)std::max(nal, 1) * 4));
    NC_TRY(nc_ensure(ctx, s->al_site, (size_t)std::max(nal, 1) * 4));
    NC_TRY(nc_ensure(ctx, s->al_member, (size_t)std::max(nal, 1) + 16));
    sa.site_of = (const int32_t *)s->site_of.p; sa.al_of = (const int32_t *)s->al_of.p;
    sa.site_pos = (int32_t *)s->site_pos.p; sa.site_chunk = (int32_t *)s->site_chunk.p; sa.site_type = (int32_t *)s->site_type.p;
    sa.site_phase = (int32_t *)s->site_phase.p; sa.site_al0 = (int32_t *)s->site_al0.p; sa.site_nr = (int32_t *)s->site_nr.p;
    sa.site_n2 = (int32_t *)s->site_n2.p; sa.al_read = (int32_t *)s->al_read.p; sa.al_site = (int32_t *)s->al_site.p; sa.al_member = (uint8_t *)s->al_member.p;
    s->have_al_ev = have_ent;
    if (have_ent) {
        NC_TRY(nc_ensure(ctx, s->al_ev, (size_t)std::max(nal, 1) * 8));
        sa.al_ev = (int2 *)s->al_ev.p;
    }
    if (impute) hipLaunchKernelGGL((k_sets<true, true>), dim3((na + 3) / 4), dim3(256), 0, ctx->stream, sa);
    else hipLaunchKernelGGL((k_sets<true, false>), dim3((na + 3) / 4), dim3(256), 0, ctx->stream, sa);
    NC_HIP(ctx, hipGetLastError());
    NC_TRY(nc_h2d_small(ctx, (int32_t *)s->site_al0.p + ns, &nal, 4, ctx->stream));
    if (s->al0_cap < NS + 1) {
        if (s->al0_pin) (void)hipHostFree(s->al0_pin);
        s->al0_pin = nullptr;
        s->al0_cap = 0;
        NC_HIP(ctx, hipHostMalloc((void **)&s->al0_pin, (NS + 1 + NS / 4) * 4, hipHostMallocDefault));
        s->al0_cap = NS + 1 + NS / 4;
    }
    NC_TRY(nc_d2h_pieces(ctx, s->al0_pin, s->site_al0.p, NS * 4, ctx->stream));
    if (timing) NC_HIP(ctx, hipEventRecord(s->ev[1], ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s->al0_pin[NS] = nal;
    if (timing) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, s->ev[0], s->ev[1]);
        s->stage_ms[0] = ms;
    }
    return NC_OK;
}

extern "C" int nc_indel_sites_scoring(nc_ctx *ctx, int32_t open, int32_t extend, int32_t match, int32_t mismatch)
{
    if (!ctx) return NC_ERR_ARG;
    // the packed 16-bit fill keeps |scores| below 2^15: 272 matches and 1,300 gap extensions must fit
    if (open < 0 || extend < 0 || open > 100 || extend > 10 || match < 0 || match > 100 || mismatch > 0 || mismatch < -100)
        return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_scoring: scores out of the range the 16-bit aligner covers");
    if (!ctx->pipe) ctx->pipe = new (std::nothrow) nc_pipe_state();
    if (!ctx->pipe) return NC_ERR_NOMEM;
    ctx->pipe->scoring[0] = open; ctx->pipe->scoring[1] = extend; ctx->pipe->scoring[2] = match; ctx->pipe->scoring[3] = mismatch;
    return NC_OK;
}

extern "C" int nc_indel_sites_run(nc_ctx *ctx, float *x_dev)
{
    if (!ctx) return NC_ERR_ARG;
    nc_pipe_state *s = ctx->pipe;
    if (!s || !s->planned) return nc_fail(ctx, NC_ERR_STATE, "nc_indel_sites_run: nc_indel_sites_plan first");
    s->ran = true;
    if (s->n_sites == 0) return NC_OK;
    if (!x_dev) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_run: x_dev");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const bool timing = ctx->timing == 1;
    const int S = s->S, ns = s->n_sites;
    const int W = s->window_after + 2;                             // n2 <= window_after + 1; row pitch n2 + 1
    const int EW = (W + 15) & ~15;
    const int WS = (s->window_after + 15) & ~15;
    const int N1 = WS;
    const int CPL = cpl_for(s->window_after + 1);
    // groups of whole sites, two in flight: the traceback codes of a group's alignments (16 KB each for 160-base windows) take a twelfth of
    // the device memory that is free when the context first runs, at most 24 GiB (a chr20-sized contig's 1.07 M alignments are then ONE group:
    // 24.4 -> 23.7 ms per pass against three groups of 6 GiB -- fewer launches and host waits, no allele stage with stream A idle); at least 1 GiB
    const int64_t tw_per_al = (int64_t)tw_blocks(N1) * 64 * CPL;          // bytes: blocks of 8 steps x 16 lanes x CPL words
    if (s->tw_budget == 0) {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess) mfree = (size_t)72 << 30;
        s->tw_budget = std::min<int64_t>((int64_t)24 << 30, std::max<int64_t>((int64_t)1 << 30, (int64_t)(mfree / 12)));
    }
    int64_t GROUP_AL = std::max<int64_t>(4096, s->tw_budget / tw_per_al);
    if (const char *g = getenv("NC_PIPE_GROUP_AL")) GROUP_AL = std::max<int64_t>(64, atoll(g));
    const int64_t GROUP_SITES = 65536;
    int32_t *err = (int32_t *)s->misc.p;
    unsigned long long *cells = (unsigned long long *)((int32_t *)s->misc.p + 4);
    long long *pool_base = (long long *)((int32_t *)s->misc.p + 8);
    // ALT pool: generous (most ALT alleles are a few dozen bases); an overflow is reported, not silent
    s->alt_pool_cap = std::max<int64_t>((int64_t)1 << 20, (int64_t)ns * S * 160);
    NC_TRY(nc_ensure(ctx, s->alt_pool, (size_t)s->alt_pool_cap));
    NC_TRY(nc_ensure(ctx, s->rlen, (size_t)ns * S * 4));
    NC_TRY(nc_ensure(ctx, s->alen, (size_t)ns * S * 4));
    const int32_t *al0h = s->al0_pin;
    std::vector<std::pair<int, int>> groups;
    {
        // as many groups as the bounds need, of about the same number of alignments each (filled greedily the last one is a remainder)
        const int64_t total_al = al0h[ns];
        const int64_t ng = std::max<int64_t>(std::max<int64_t>(1, (total_al + GROUP_AL - 1) / GROUP_AL), (ns + GROUP_SITES - 1) / GROUP_SITES);
        const int64_t target = std::min<int64_t>(GROUP_AL, (total_al + ng - 1) / ng + 64);
        for (int k0 = 0; k0 < ns;) {
            int k1 = k0 + 1;
            while (k1 < ns && k1 - k0 < GROUP_SITES && (int64_t)al0h[k1 + 1] - al0h[k0] <= target) k1++;
            groups.emplace_back(k0, k1);
            k0 = k1;
        }
    }
    const int G = (int)groups.size();
    // A contig with more alignments than one group holds is cut into groups of about the same size, each smaller than GROUP_AL; a later, slightly
    // shorter contig may then arrive as ONE group of nearly GROUP_AL alignments -- larger than any group before it.  Sized by their own group, the
    // per-alignment buffers then grew in the middle of a whole-genome pass (hipFree + hipMalloc of ~9 GB: a second of idle GPU at chr18 after
    // chr1 .. chr17).  A run of several groups therefore sizes them for GROUP_AL at once; a run of one group takes what it needs.
    const size_t Acap = G > 1 ? (size_t)std::min<int64_t>(GROUP_AL, al0h[ns]) + 64 : 0;
    // Stream A (the context's): query windows + alignment fill (bound by vector issue).  Stream B: traceback, tensors, allele_prediction
    // (bound by memory latency) of the previous group, beside it on the same CUs.  Stage timers (timing mode) need the stages one
    // after the other: one stream then.
    const bool two = !timing && G > 1 && !getenv("NC_PIPE_ONE_STREAM");
    if (two && !s->sB) {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        // high priority: its latency-bound kernels take the wave slots the issue-bound fill leaves free as soon as they open
        NC_HIP(ctx, hipStreamCreateWithPriority(&s->sB, hipStreamNonBlocking, getenv("NC_PIPE_B_PRIO_LOW") ? prio_lo : prio_hi));
        for (auto &e : s->evA) NC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : s->evB) NC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        NC_HIP(ctx, hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    hipStream_t sA = ctx->stream, sB = two ? s->sB : ctx->stream;
    if (two) {                                                         // B starts behind the plan's kernels
        NC_HIP(ctx, hipEventRecord(s->ev_join, sA));
        NC_HIP(ctx, hipStreamWaitEvent(sB, s->ev_join, 0));
    }
    FillArgs fa_of[2];
    BandArgs ba_of[2];
    bool band_of[2] = {false, false};
    auto stage_a = [&](int g) -> int {
        const int b = g & 1, k0 = groups[(size_t)g].first, k1 = groups[(size_t)g].second;
        nc_pipe_state::GroupBufs &B = s->gb[b];
        const int ng = k1 - k0;
        const int64_t A0 = al0h[k0];
        const int32_t Ag = al0h[k1] - al0h[k0];
        const size_t Agz = (size_t)std::max(Ag, 1);
        const size_t Asz = std::max(Agz, Acap);                          // what the buffers are sized for (the layout inside them goes by Agz)
        NC_TRY(nc_ensure(ctx, B.win, Asz * WS + 64));
        NC_TRY(nc_ensure(ctx, B.n1, Asz * 4));
        NC_TRY(nc_ensure(ctx, B.tw, Asz * (size_t)tw_per_al + 64));
        NC_TRY(nc_ensure(ctx, B.hlast, Asz * hlast_pitch(W) * 4 + 64));
        NC_TRY(nc_ensure(ctx, B.hcol, Asz * hcol_pitch(N1) * 4 + 64));
        NC_TRY(nc_ensure(ctx, B.endc, Asz * sizeof(int2)));
        NC_TRY(nc_ensure(ctx, B.trace, Asz * EW * 4 + 64));
        NC_TRY(nc_ensure(ctx, B.cns, (size_t)ng * S * CNS_CAP));
        NC_TRY(nc_ensure(ctx, B.ncns, (size_t)ng * S * 4));
        NC_TRY(nc_ensure(ctx, B.cband, (size_t)ng * S * 4));
        NC_TRY(nc_ensure(ctx, B.arow, ((size_t)ng * S + 1) * 8));
        NC_TRY(nc_ensure(ctx, B.alt_off, (size_t)ng * S * 8));
        const int nblk = (((N1 + W + 7) / 8) + 3) & ~3;                             // blocks of 8 anti-diagonals of a banded alignment: n1 + n2 <= N1 + W (41 for the 160-base windows, 66 for the 260-base ones)
        const bool band = packed_fill() && (s->band_mode < 0 ? band_on() : s->band_mode != 0) && nblk <= 80;
        if (band) {
            NC_TRY(nc_ensure(ctx, B.band_lo, 2 * Asz + 128));            // + the windows' classes (k_windows16 -> k_window_lists)
            NC_TRY(nc_ensure(ctx, B.lists, Asz * 3 * 4 + 64));
            NC_TRY(nc_ensure(ctx, B.counts, 64));
            NC_TRY(nc_ensure(ctx, B.twb, Asz * (size_t)nblk * (4 * TWB_PITCH) + 256));
            NC_TRY(nc_ensure(ctx, B.hrow, Asz * 128 + 64));
            NC_TRY(nc_ensure(ctx, B.hcolb, Asz * 128 + 64));
        }
        if (two && g >= 2) NC_HIP(ctx, hipStreamWaitEvent(sA, s->evB[b], 0));     // stream B is done with this buffer set (group g - 2)
        if (timing) NC_HIP(ctx, hipEventRecord(s->ev[0], sA));
        if (band) NC_HIP(ctx, hipMemsetAsync(B.counts.p, 0, 64, sA));
        // ---- query windows
        WinArgs wa;
        wa.codes = s->pack.codes; wa.slot_off = s->rd.slot_off; wa.rd_start = s->rd.rd_start; wa.rd_end = s->rd.rd_end;
        wa.ev_off = s->rd.ev_off; wa.ev_pos = s->rd.ev_pos; wa.ev_len = s->rd.ev_len; wa.ins_off = s->rd.ins_off; wa.tail_off = s->rd.tail_off;
        wa.ins_bases = s->rd.ins_bases; wa.tail_bases = s->rd.tail_bases; wa.read_flag = s->rd.read_flag;
        wa.al_read = (const int32_t *)s->al_read.p + A0; wa.al_site = (const int32_t *)s->al_site.p + A0;
        wa.site_pos = (const int32_t *)s->site_pos.p; wa.site_n2 = (const int32_t *)s->site_n2.p;
        wa.al_ev = s->have_al_ev ? (const int2 *)s->al_ev.p + A0 : nullptr;
        wa.A = Ag; wa.W = s->window_after; wa.WS = WS; wa.win = (uint8_t *)B.win.p; wa.n1 = (int32_t *)B.n1.p; wa.cells = cells;
        wa.band_lo = band ? (int8_t *)B.band_lo.p : nullptr;
        wa.wcls = band ? (int8_t *)B.band_lo.p + Agz + 64 : nullptr;
        wa.list1 = (int32_t *)B.lists.p; wa.list2 = wa.list1 + Agz; wa.listF = wa.list2 + Agz;
        wa.counts = (int32_t *)B.counts.p; wa.band_margin = s->band_margin_v > 0 ? s->band_margin_v : band_margin();
        if (Ag > 0) {
            // NC_PIPE_WINDOWS = serial: one lane per window (the round-3 kernel); force16: the 16-lane kernel with every window on its serial route
            const char *wm = getenv("NC_PIPE_WINDOWS");
            if (wm && !strcmp(wm, "serial")) hipLaunchKernelGGL(k_windows, dim3((Ag + 63) / 64), dim3(64), 0, sA, wa);
            else if (WS <= WIN_ROW - 16) {
                hipLaunchKernelGGL(k_windows16, dim3((Ag + 3) / 4), dim3(64), 0, sA, wa, (wm && !strcmp(wm, "force16")) ? 1 : 0);
                hipLaunchKernelGGL(k_window_lists, dim3((Ag + 4095) / 4096), dim3(1024), 0, sA, wa);
            }
            else hipLaunchKernelGGL(k_windows, dim3((Ag + 63) / 64), dim3(64), 0, sA, wa);
        }
        if (timing) NC_HIP(ctx, hipEventRecord(s->ev[1], sA));
        // ---- star alignment: every read window against its site's reference window (free tail)
        FillArgs &fa = fa_of[b];
        fa.s1 = (const uint8_t *)B.win.p; fa.s1_stride = WS; fa.n1 = (const int32_t *)B.n1.p;
        fa.ref_code = s->ref_code; fa.ref_pos0 = s->ref_pos0; fa.site_pos = (const int32_t *)s->site_pos.p; fa.site_n2 = (const int32_t *)s->site_n2.p;
        fa.al_site = (const int32_t *)s->al_site.p + A0; fa.site0 = 0; fa.site_div = 1;
        fa.A = Ag; fa.W = W;
        fa.open = s->scoring[0]; fa.extend = s->scoring[1]; fa.match = s->scoring[2]; fa.mismatch = s->scoring[3];
        fa.arow = nullptr; fa.N1 = N1;
        fa.Tw = (uint32_t *)B.tw.p;
        fa.Hlast = (int32_t *)B.hlast.p; fa.hcol = (int32_t *)B.hcol.p; fa.endcell = (int2 *)B.endc.p;
        fa.list = nullptr; fa.count = nullptr;
        band_of[b] = band;
        if (band && Ag > 0) {
            // every alignment on its band (the classes' sizes are known on the device only: the grids cover the group, blocks beyond a
            // class's count leave at once); the full matrix runs behind the banded traceback, over listF (stage_b1)
            BandArgs &ba = ba_of[b];
            ba.f = fa;
            ba.band_lo = (const int8_t *)B.band_lo.p; ba.Twb = (uint32_t *)B.twb.p; ba.hrow = (int16_t *)B.hrow.p; ba.hcolb = (int16_t *)B.hcolb.p;
            ba.NBLK = nblk; ba.redo_list = wa.listF; ba.redo_count = wa.counts + 2;
            ba.edge = getenv("NC_PIPE_BAND_EDGE") ? atoi(getenv("NC_PIPE_BAND_EDGE")) : 0;
            ba.list = wa.list1; ba.count = wa.counts;
            hipLaunchKernelGGL(k_fill_band<1>, dim3((Ag + 7) / 8), dim3(64), 0, sA, ba);
            ba.list = wa.list2; ba.count = wa.counts + 1;
            hipLaunchKernelGGL(k_fill_band<2>, dim3((Ag + 7) / 8), dim3(64), 0, sA, ba);
        } else if (Ag > 0) launch_fill(ctx, sA, CPL, fa);
        NC_HIP(ctx, hipGetLastError());
        if (timing) NC_HIP(ctx, hipEventRecord(s->ev[2], sA));
        if (two) NC_HIP(ctx, hipEventRecord(s->evA[b], sA));
        return NC_OK;
    };
    // traceback + tensors + the row count of the allele alignments (left in the mailbox)
    auto stage_b1 = [&](int g) -> int {
        const int b = g & 1, k0 = groups[(size_t)g].first, k1 = groups[(size_t)g].second;
        nc_pipe_state::GroupBufs &B = s->gb[b];
        const int ng = k1 - k0;
        const int64_t A0 = al0h[k0];
        const int32_t Ag = al0h[k1] - al0h[k0];
        const FillArgs &fa = fa_of[b];
        if (two) NC_HIP(ctx, hipStreamWaitEvent(sB, s->evA[b], 0));
        if (band_of[b] && Ag > 0) {
            BandArgs ba = ba_of[b];
            const int32_t *cnts = ba.count - 1;                         // (ba.count was left on the second class)
            ba.list = (const int32_t *)B.lists.p; ba.count = cnts;
            BandArgs ba2 = ba;
            ba2.list = (const int32_t *)B.lists.p + std::max(Ag, 1); ba2.count = cnts + 1;
            hipLaunchKernelGGL(k_trace_band12, dim3((Ag + 63) / 64, 2), dim3(64), 0, sB, ba, ba2, (uint32_t *)B.trace.p, EW);
            // the rest on the full matrix: too wide for a band, or a path that touched the edge of its band
            FillArgs fl = fa;
            fl.list = ba.redo_list; fl.count = ba.redo_count;
            launch_fill(ctx, sB, CPL, fl);
            hipLaunchKernelGGL(k_end_cells, dim3((Ag + 15) / 16), dim3(256), 0, sB, fl);
            hipLaunchKernelGGL(k_trace16p, dim3((Ag + 63) / 64), dim3(64), 0, sB, fl, CPL, 1, (uint32_t *)B.trace.p, EW);
            hipLaunchKernelGGL(k_band_stats, dim3(1), dim3(1), 0, sB, cnts, (long long *)((int32_t *)s->misc.p + 16));
        } else if (Ag > 0) {
            hipLaunchKernelGGL(k_end_cells, dim3((Ag + 15) / 16), dim3(256), 0, sB, fa);
            hipLaunchKernelGGL(k_trace16p, dim3((Ag + 63) / 64), dim3(64), 0, sB, fa, CPL, packed_fill() ? 1 : 0, (uint32_t *)B.trace.p, EW);
        }
        if (timing) NC_HIP(ctx, hipEventRecord(s->ev[3], sB));
        // ---- columns, histogram, tensor, consensus
        TensorArgs ta;
        ta.site0 = k0; ta.n_sites_g = ng; ta.S = S; ta.haploid = s->haploid; ta.W = W; ta.WS = WS; ta.A0 = A0;
        ta.site_al0 = (const int32_t *)s->site_al0.p; ta.site_nr = (const int32_t *)s->site_nr.p; ta.site_pos = (const int32_t *)s->site_pos.p;
        ta.site_n2 = (const int32_t *)s->site_n2.p; ta.al_member = (const uint8_t *)s->al_member.p; ta.win = (const uint8_t *)B.win.p;
        ta.ent = (const uint32_t *)B.trace.p; ta.EW = EW; ta.ref_code = s->ref_code; ta.ref_pos0 = s->ref_pos0; ta.x = x_dev;
        ta.cns = (uint8_t *)B.cns.p; ta.ncns = (int32_t *)B.ncns.p; ta.cband = (int16_t *)B.cband.p; ta.err = err;
        if (s->maxcov <= 255) hipLaunchKernelGGL(k_site_tensor<uint8_t>, dim3(ng), dim3(256), 0, sB, ta);
        else hipLaunchKernelGGL(k_site_tensor<uint16_t>, dim3(ng), dim3(256), 0, sB, ta);
        if (timing) NC_HIP(ctx, hipEventRecord(s->ev[4], sB));
        int32_t *mbox = (int32_t *)s->misc.p + 2;
        NC_TRY((scan_launch<SC_TWB, int64_t>(ctx, sB, s->part_b, (const int32_t *)B.ncns.p, ng * S, 0, (int64_t *)B.arow.p, true, mbox, nullptr)));
        NC_HIP(ctx, hipGetLastError());
        NC_TRY(nc_d2h(ctx, ctx->mbox + 36, mbox, 8, sB));
        return NC_OK;
    };
    // allele_prediction: global alignment of every consensus against its window (parasail scoring 9 / 1 / 20 / -10, :79)
    auto stage_b2 = [&](int g, int64_t rows) -> int {
        const int b = g & 1, k0 = groups[(size_t)g].first, k1 = groups[(size_t)g].second;
        nc_pipe_state::GroupBufs &B = s->gb[b];
        const int nset = (k1 - k0) * S;
        // (a run of several groups sizes these for the largest group its bounds allow, like stage_a's buffers: no growth in the middle of a genome)
        const int64_t nset_cap = G > 1 ? std::max<int64_t>(nset, std::min<int64_t>(GROUP_SITES, ns) * S) : nset;
        const int64_t rows_cap = nset > 0 ? (rows * nset_cap + nset - 1) / nset : rows;
        NC_TRY(nc_ensure(ctx, s->tw2, (size_t)(rows_cap + 1) * 64 * CPL + 64));                 // `rows` counts blocks of TWB steps
        NC_TRY(nc_ensure(ctx, s->runs, (size_t)(TWB * rows_cap + nset_cap * (W + 1) + 2) * 4 + 64));
        FillArgs fb = fa_of[b];
        fb.s1 = (const uint8_t *)B.cns.p; fb.s1_stride = CNS_CAP; fb.n1 = (const int32_t *)B.ncns.p;
        fb.al_site = nullptr; fb.site0 = k0; fb.site_div = S;
        fb.A = nset;
        fb.open = 9; fb.extend = 1; fb.match = 20; fb.mismatch = -10;
        fb.arow = (const int64_t *)B.arow.p; fb.N1 = 0;
        fb.Tw = (uint32_t *)s->tw2.p; fb.Hlast = nullptr; fb.hcol = nullptr; fb.endcell = nullptr;
        fb.list = nullptr; fb.count = nullptr;
        int32_t *rl = (int32_t *)s->rlen.p + (size_t)k0 * S, *al = (int32_t *)s->alen.p + (size_t)k0 * S;
        BandArgs bb;
        memset(&bb, 0, sizeof bb);
        bb.f = fb;
        // allele_prediction is an exact global alignment in the reference (parasail nw_trace, generate_indel_pileups.py:79), and REF / ALT strings are row
        // a13's bit-exact output.  The banded form is therefore kept only where it PROVES itself: k_allele_trace_b12 compares the score of the path it
        // walked with the most any path outside the band can reach (allele_trace_body's certificate) and sends every set it cannot certify -- and every
        // path that touches an edge diagonal -- to the full matrix.  NC_PIPE_BAND_ALLELES=0: every consensus on the full matrix (+1.0 ms per chr20 pass).
        const char *bae = getenv("NC_PIPE_BAND_ALLELES");                 // (read per run: the tests switch it)
        const bool band_alleles = !(bae && atoi(bae) == 0);
        if (band_of[b] && band_alleles) {
            // the consensus against its window on a band around the diagonals 0 .. n2 - n1; too long / too wide / edge-touching ones on the full matrix
            const size_t nz = (size_t)std::max(nset, 1), nzc = (size_t)std::max<int64_t>(nset_cap, 1);
            NC_TRY(nc_ensure(ctx, s->ab_lo, nzc + 64));
            NC_TRY(nc_ensure(ctx, s->ab_lists, nzc * 3 * 4 + 64));
            NC_TRY(nc_ensure(ctx, s->ab_counts, 64));
            NC_TRY(nc_ensure(ctx, s->ab_twb, nzc * (size_t)BAND_NBLK4 * (4 * TWB_PITCH) + 256));
            NC_HIP(ctx, hipMemsetAsync(s->ab_counts.p, 0, 64, sB));
            int32_t *l1 = (int32_t *)s->ab_lists.p, *l2 = l1 + nz, *lF = l2 + nz, *cn = (int32_t *)s->ab_counts.p;
            hipLaunchKernelGGL(k_allele_classes, dim3((nset + 255) / 256), dim3(256), 0, sB, fb, (const int16_t *)B.cband.p, s->band_margin_v > 0 ? s->band_margin_v : band_margin(),
                               8 * BAND_NBLK, (int8_t *)s->ab_lo.p, l1, l2, lF, cn);
            bb.band_lo = (const int8_t *)s->ab_lo.p; bb.Twb = (uint32_t *)s->ab_twb.p; bb.hrow = nullptr; bb.hcolb = nullptr; bb.NBLK = BAND_NBLK4;
            bb.redo_list = lF; bb.redo_count = cn + 2; bb.edge = 0;
            bb.list = l1; bb.count = cn;
            BandArgs bb2 = bb;
            bb2.list = l2; bb2.count = cn + 1;
            hipLaunchKernelGGL(k_fill_band<1>, dim3((nset + 7) / 8), dim3(64), 0, sB, bb);
            hipLaunchKernelGGL(k_fill_band<2>, dim3((nset + 7) / 8), dim3(64), 0, sB, bb2);
            hipLaunchKernelGGL(k_allele_trace_b12, dim3((nset + 63) / 64, 2), dim3(64), 0, sB, bb, bb2, CPL, (const int32_t *)s->site_type.p, s->win_size,
                               (int16_t *)s->runs.p, rl, al);
            fb.list = lF; fb.count = cn + 2;
            bb.f = fb;
        }
        launch_fill(ctx, sB, CPL, fb);
        hipLaunchKernelGGL(k_allele_trace16p<0>, dim3((nset + 63) / 64), dim3(64), 0, sB, bb, CPL, packed_fill() ? 1 : 0, (const int32_t *)s->site_type.p,
                           s->win_size, (int16_t *)s->runs.p, rl, al);
        NC_TRY((scan_launch<SC_POS, int64_t>(ctx, sB, s->part_b, (const int32_t *)al, nset, 0, (int64_t *)B.alt_off.p, false, nullptr, pool_base)));
        hipLaunchKernelGGL(k_alt_copy, dim3((nset + 3) / 4), dim3(256), 0, sB, (const uint8_t *)B.cns.p, (const int32_t *)al,
                           (const int64_t *)B.alt_off.p, nset, (uint8_t *)s->alt_pool.p, s->alt_pool_cap, err);
        NC_HIP(ctx, hipGetLastError());
        if (two) NC_HIP(ctx, hipEventRecord(s->evB[b], sB));
        if (timing) {
            NC_HIP(ctx, hipEventRecord(s->ev[5], sB));
            NC_HIP(ctx, hipEventSynchronize(s->ev[5]));
            for (int st = 0; st < 5; st++) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, s->ev[st], s->ev[st + 1]);
                s->stage_ms[st + 1] += ms;
            }
            s->cells[1] += TWB * rows * (int64_t)(s->window_after + 1);               // (upper estimate: whole blocks)
        }
        return NC_OK;
    };
    int dump_g = -1;
    NC_TRY(stage_a(0));
    for (int g = 0; g < G; g++) {
        if (!timing && g + 1 < G) NC_TRY(stage_a(g + 1));            // the next group's alignments are enqueued before the host waits for this one's row count
        NC_TRY(stage_b1(g));
        volatile int32_t *mb = ctx->mbox + 36;
        NC_HIP(ctx, hipStreamSynchronize(sB));
        if (const char *dump = getenv("NC_PIPE_DUMP")) {              // debugging aid: the group's alignments as files <dump>.<name>
            const int b = g & 1;
            const int32_t Ag = al0h[groups[(size_t)g].second] - al0h[groups[(size_t)g].first];
            nc_pipe_state::GroupBufs &B = s->gb[b];
            auto wr = [&](const char *name, const void *dev, size_t bytes) {
                std::vector<char> h(bytes);
                if (hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return;
                char path[512];
                snprintf(path, sizeof path, "%s.%s", dump, name);
                if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), 1, bytes, fp); fclose(fp); }
            };
            wr("trace", B.trace.p, (size_t)Ag * EW * 4);
            wr("win", B.win.p, (size_t)Ag * WS);
            wr("n1", B.n1.p, (size_t)Ag * 4);
            wr("al_site", (const int32_t *)s->al_site.p + al0h[groups[(size_t)g].first], (size_t)Ag * 4);
            wr("al_read", (const int32_t *)s->al_read.p + al0h[groups[(size_t)g].first], (size_t)Ag * 4);
            wr("site_pos", s->site_pos.p, (size_t)ns * 4);
            wr("site_n2", s->site_n2.p, (size_t)ns * 4);
            if (B.band_lo.p) wr("band_lo", B.band_lo.p, (size_t)Ag);
            dump_g = g;
        }
        const int64_t rows = ((int64_t)mb[1] << 31) | (int64_t)(mb[0] & 0x7fffffff);
        NC_TRY(stage_b2(g, rows));
        if (dump_g == g) {                                           // (debugging aid, continued: the allele stage's arrays of the group)
            const char *dump = getenv("NC_PIPE_DUMP");
            NC_HIP(ctx, hipStreamSynchronize(sB));
            const int k0 = groups[(size_t)g].first, nset = (groups[(size_t)g].second - k0) * S;
            nc_pipe_state::GroupBufs &B = s->gb[g & 1];
            auto wr = [&](const char *name, const void *dev, size_t bytes) {
                std::vector<char> h(bytes);
                if (!dev || hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return;
                char path[512];
                snprintf(path, sizeof path, "%s.%s", dump, name);
                if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), 1, bytes, fp); fclose(fp); }
            };
            wr("cns", B.cns.p, (size_t)nset * CNS_CAP);
            wr("ncns", B.ncns.p, (size_t)nset * 4);
            wr("rlen", (const int32_t *)s->rlen.p + (size_t)k0 * S, (size_t)nset * 4);
            wr("alen", (const int32_t *)s->alen.p + (size_t)k0 * S, (size_t)nset * 4);
            wr("ab_lo", s->ab_lo.p, (size_t)nset);
            wr("ab_counts", s->ab_counts.p, 16);
        }
        if (timing && g + 1 < G) NC_TRY(stage_a(g + 1));
    }
    if (two) {                                                         // the caller's stream continues behind stream B
        NC_HIP(ctx, hipEventRecord(s->ev_join, sB));
        NC_HIP(ctx, hipStreamWaitEvent(sA, s->ev_join, 0));
    }
    return NC_OK;
}

extern "C" int nc_indel_sites_fetch(nc_ctx *ctx, int32_t *pos, int32_t *chunk, int32_t *var_type, int32_t *phase, int32_t *ref_len,
                                    int32_t *alt_len, int64_t *n_alt_bytes)
{
    if (!ctx) return NC_ERR_ARG;
    nc_pipe_state *s = ctx->pipe;
    if (!s || !s->planned) return nc_fail(ctx, NC_ERR_STATE, "nc_indel_sites_fetch: nc_indel_sites_plan first");
    if (n_alt_bytes) *n_alt_bytes = 0;
    if (s->n_sites == 0) return NC_OK;
    const size_t NS = (size_t)s->n_sites;
    auto cp = [&](void *dst, const DevBuf &b, size_t bytes) -> int {
        if (dst) NC_HIP(ctx, hipMemcpyAsync(dst, b.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return NC_OK;
    };
    NC_TRY(cp(pos, s->site_pos, NS * 4));
    NC_TRY(cp(chunk, s->site_chunk, NS * 4));
    NC_TRY(cp(var_type, s->site_type, NS * 4));
    NC_TRY(cp(phase, s->site_phase, NS * 4));
    int32_t misc[32] = {0};
    if (s->ran) {
        NC_TRY(cp(ref_len, s->rlen, NS * s->S * 4));
        NC_TRY(cp(alt_len, s->alen, NS * s->S * 4));
    }
    NC_HIP(ctx, hipMemcpyAsync(misc, s->misc.p, sizeof misc, hipMemcpyDeviceToHost, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (misc[0] & 2) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites: a read set needs more than %d alignment columns", CNS_CAP);
    if (misc[0] & 4) return nc_fail(ctx, NC_ERR_CAPACITY, "nc_indel_sites: ALT allele pool overflow");
    long long cells = 0, pool = 0;
    memcpy(&cells, misc + 4, 8);
    memcpy(&pool, misc + 8, 8);
    memcpy(s->band_stats, misc + 16, 32);
    memcpy(s->band_stats + 4, misc + 6, 8);
    s->cells[0] = cells;
    if (n_alt_bytes) *n_alt_bytes = pool;
    return NC_OK;
}

extern "C" int nc_indel_sites_fetch_alt(nc_ctx *ctx, uint8_t *alt_bases, int64_t cap)
{
    if (!ctx) return NC_ERR_ARG;
    nc_pipe_state *s = ctx->pipe;
    if (!s || !s->ran) return nc_fail(ctx, NC_ERR_STATE, "nc_indel_sites_fetch_alt: nc_indel_sites_run first");
    if (cap <= 0 || s->n_sites == 0) return NC_OK;
    if (!alt_bases) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_fetch_alt: null buffer");
    NC_HIP(ctx, hipMemcpyAsync(alt_bases, s->alt_pool.p, (size_t)std::min<int64_t>(cap, s->alt_pool_cap), hipMemcpyDeviceToHost, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NC_OK;
}

extern "C" int nc_indel_sites_band(nc_ctx *ctx, int32_t mode, int32_t margin)
{
    if (!ctx || mode < -1 || mode > 1 || margin < 0 || margin > 15) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_sites_band: mode -1 / 0 / 1, margin 0 (default) .. 15");
    if (!ctx->pipe) ctx->pipe = new (std::nothrow) nc_pipe_state();
    if (!ctx->pipe) return NC_ERR_NOMEM;
    ctx->pipe->band_mode = mode;
    ctx->pipe->band_margin_v = margin;
    return NC_OK;
}

extern "C" int nc_indel_sites_band_stats(nc_ctx *ctx, int64_t *stats6)
{
    if (!ctx || !ctx->pipe || !stats6) return NC_ERR_ARG;
    for (int k = 0; k < 6; k++) stats6[k] = ctx->pipe->band_stats[k];
    return NC_OK;
}

extern "C" int nc_indel_sites_stage_ms(nc_ctx *ctx, float *ms6, int64_t *cells2)
{
    if (!ctx || !ctx->pipe) return NC_ERR_ARG;
    if (ms6) for (int k = 0; k < 6; k++) ms6[k] = ctx->pipe->stage_ms[k];
    if (cells2) { cells2[0] = ctx->pipe->cells[0]; cells2[1] = ctx->pipe->cells[1]; }
    return NC_OK;
}
