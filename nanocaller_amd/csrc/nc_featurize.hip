// K2-K4: per-candidate neighbour selection, read gather and (5,41,5) tensor build (gfx950).
//
// Restates get_cnd_pos + the per-candidate loop of get_snp_testing_candidates (reference
// generate_SNP_pileups.py:6-101, 200-263).  One 64-lane wavefront per candidate
// site: lanes first act as buckets (binary searches into the sorted neighbour-site list), then as reads
// (which tile entries cover the site; strand/base depth ballots), then as tensor columns (lane j gathers
// the code of each sampled read at column j and keeps a 4x4 histogram in packed 8/16-bit counters).  The
// walk over the sampled reads is driven by SCALAR code: the covering entries are a wave-uniform bit mask, split
// by the read's base at the candidate so that the histogram row is a compile-time register; the entry record
// of a read is fetched from the lane that loaded it in the coverage pass (v_readlane: no second trip to memory)
// and the gather is one global_load_ubyte with a scalar base and a 32-bit lane offset, 16 reads in flight per
// round (~6 vector instructions per read and 34 VGPRs: 8 waves per SIMD).  The
// site tensor is assembled in LDS and four sites leave the workgroup as one aligned, coalesced dwordx4 stream.
#include <type_traits>

#include "nc_common.h"

namespace {

constexpr int MAXCOV_CAP = 1024;   // LDS list of sampled reads per wave: largest supported maxcov
constexpr int MAXCOV_SMALL = 256;  // instantiation for maxcov <= 255 (default 160): 20.5 KB of LDS per block -> 7 waves per SIMD instead of 4,
                                   // and 8-bit histogram fields (32-bit instead of 64-bit counter updates)
constexpr int NBR = 20;

struct Bucket { int32_t dlo, dhi, k, far; };     // distance range (dlo, dhi], pick k, far=1: farthest k
struct ModeTab { int32_t nb, W; Bucket b[7]; };

// One row per list comprehension of get_cnd_pos (generate_SNP_pileups.py:6-101), expressed by distance |p-v|.
// Left and right sides are mirror images; only the innermost bucket may pick the FARTHEST sites (quirk E5).
__constant__ ModeTab MODES[5] = {
    {5, 50000, {{0, 2000, 2, 1}, {2000, 5000, 3, 0}, {5000, 10000, 4, 0}, {10000, 20000, 5, 0}, {20000, 1 << 30, 6, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}},
    {3, 50000, {{0, 2000, 5, 0}, {2000, 5000, 10, 0}, {5000, 1 << 30, 5, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}},
    {7, 100000, {{0, 2000, 2, 1}, {2000, 5000, 2, 0}, {5000, 10000, 3, 0}, {10000, 20000, 3, 0}, {20000, 40000, 4, 0}, {40000, 50000, 3, 0}, {50000, 1 << 30, 3, 0}}},
    {7, 300000, {{0, 10000, 2, 1}, {10000, 20000, 2, 0}, {20000, 50000, 3, 0}, {50000, 75000, 3, 0}, {75000, 100000, 4, 0}, {100000, 200000, 4, 0}, {200000, 1 << 30, 2, 0}}},
    {4, 20000, {{0, 2000, 4, 1}, {2000, 5000, 5, 0}, {5000, 10000, 5, 0}, {10000, 20000, 6, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}},
};

#ifndef NC_NBR_IDX_SHIFT
#define NC_NBR_IDX_SHIFT 10
#endif
constexpr int NBR_IDX_SHIFT = NC_NBR_IDX_SHIFT;   // coarse index granularity: first neighbour site >= every 2^shift-th position

__device__ __forceinline__ int lower_bound_i32(const int32_t *a, int n, int64_t key, const int32_t *cidx, int32_t cidx_pos0, int n_cidx)
{
    int lo = 0, hi = n;
    // narrow [lo, hi) with the coarse index: cidx[b] = lower_bound(a, cidx_pos0 + (b << 10))
    const int64_t rel = key - cidx_pos0;
    if (rel <= 0) hi = cidx[0];
    else {
        const int64_t b = rel >> NBR_IDX_SHIFT;
        if (b >= n_cidx - 1) lo = cidx[n_cidx - 1];
        else { lo = cidx[b]; hi = cidx[b + 1]; }
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct FeatArgs {
    const uint8_t *codes;
    const int32_t *tile_off;
    const nc_tile_entry *tile_ent;
    int32_t tile_pos0, tile_shift;
    const uint8_t *ref_code;
    int32_t ref_pos0;
    const int32_t *nbr_pos;
    int32_t n_nbr;
    const int32_t *cidx;
    int32_t cidx_pos0, n_cidx;
    const int32_t *site_pos, *site_chunk;
    const int32_t *chunk_start, *chunk_end;
    int32_t n_sites, mode, maxcov, min_nbr_sites;
    float *x;
    int32_t *ref_out, *fwd, *rev, *depth;
    uint8_t *valid;
    int32_t x_i16;                 // 1: the tensors leave as int16 (every entry is a small integer: exact), half the bytes
    // alignments that share a read name (nc_snp_set_mates; k_featurize_pairs<true>): slot offsets ascending, {start, end, next member of the name, 0}
    const int64_t *mate_key;
    const int4 *mate_rec;
    int32_t n_mates;
};

// The reference's pileup dicts are keyed by read NAME (generate_SNP_pileups.py:175,185): where several alignments of one name cover a column the
// last in file order is the column's entry, and a site's row for a name takes every column from whichever of the name's alignments covers it
// (:223,232).  named_member: among the alignments of the name of the read whose slot starts at byte `key`, the last one (= the largest table
// index: slots ascend in file order) that covers p; -1: none.  *self = the read's own index in the table.
__device__ __forceinline__ int named_member(const FeatArgs &a, int64_t key, int32_t p, int *self)
{
    int lo = 0, hi = a.n_mates;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.mate_key[mid] < key) lo = mid + 1; else hi = mid; }
    if (self) *self = lo;
    if (lo >= a.n_mates || a.mate_key[lo] != key) { if (self) *self = -2; return -1; }     // (an entry flagged without a row: a table of another pack)
    int best = -1, i = lo;
    for (int guard = 0; guard < 64; guard++) {                       // (a name's ring closes on itself; a damaged table must not hang the wave)
        const int4 r = a.mate_rec[i];
        if (r.x <= p && p < r.y && i > best) best = i;
        i = r.z;
        if (i == lo || i < 0 || i >= a.n_mates) break;
    }
    return best;
}
__device__ __forceinline__ int named_code(const FeatArgs &a, int64_t key, int32_t p)
{
    const int m = named_member(a, key, p, nullptr);
    if (m < 0) return 4;
    return a.codes[a.mate_key[m] - (a.mate_rec[m].x & ~15) + p];
}

// I16: the tensors leave as int16 (product path).  Then nothing is assembled in LDS: lane L < 41 takes tensor column L, fetches
// the counters of the lane that walked that column (ds_bpermute) and writes its five rows' 10 bytes straight to global memory
// (41 lanes x 10 B = one contiguous 410-byte row per store pair) -- no zero fill, no workgroup barrier, no LDS round trip.
template <int CAP, bool I16>
__global__ __launch_bounds__(256) void k_featurize(FeatArgs a)
{
    __shared__ __attribute__((aligned(16))) float sm[I16 ? 1 : 4][I16 ? 4 : NC_SNP_TENSOR + 3];
    __shared__ int32_t nlist[4][64];                              // per wave: indices (into nbr_pos) of the picked neighbour sites
    // the wave index is wave-uniform: telling the compiler so puts the site's scalars (position, tile, entry range) and
    // the entry records of the read loop into SGPRs / the scalar cache
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware mapping (workgroup b runs on XCD b % 8): each XCD walks ONE contiguous range of position-sorted
    // sites, so its private L2 holds one genomic neighbourhood instead of all eight sharing every line.
    const int nblk = (int)gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int s = blk * 4 + wv;
    float *X = sm[I16 ? 0 : wv];
    if constexpr (!I16)
        for (int i = lane; i < NC_SNP_TENSOR; i += 64) X[i] = 0.0f;

    if (s < a.n_sites) {
        const int32_t v = __builtin_amdgcn_readfirstlane(a.site_pos[s]);
        const int32_t ch = __builtin_amdgcn_readfirstlane(a.site_chunk[s]);
        // neighbour sites are those of the owning chunk's own scan window (quirk E9)
        const int64_t win_lo = max((int64_t)1, (int64_t)a.chunk_start[ch] - NC_FLANK);
        const int64_t win_hi = (int64_t)a.chunk_end[ch] + NC_FLANK;
        const ModeTab &M = MODES[a.mode];
        const int nb = M.nb;

        // ---- K2: lanes [0,nb) = left buckets far->near, lanes [nb,2nb) = right buckets near->far
        int take = 0, idx0 = 0;
        if (lane < 2 * nb) {
            const bool left = lane < nb;
            const Bucket B = M.b[left ? nb - 1 - lane : lane - nb];
            const int64_t dhi = min((int64_t)B.dhi, (int64_t)M.W - 1);       // abs(p - v) < W
            int64_t pmin = left ? (int64_t)v - dhi : (int64_t)v + B.dlo + 1;
            int64_t pmax = left ? (int64_t)v - B.dlo - 1 : (int64_t)v + dhi;
            pmin = max(pmin, win_lo);
            pmax = min(pmax, win_hi);
            if (pmin <= pmax) {
                const int lo = lower_bound_i32(a.nbr_pos, a.n_nbr, pmin, a.cidx, a.cidx_pos0, a.n_cidx);
                const int hi = lower_bound_i32(a.nbr_pos, a.n_nbr, pmax + 1, a.cidx, a.cidx_pos0, a.n_cidx);
                const int m = hi - lo;
                take = min(m, B.k);
                // ascending order: on the left the farthest are first, on the right the farthest are last
                const bool first = left ? (B.far != 0) : (B.far == 0);
                idx0 = first ? lo : hi - take;
            }
        }
        int incl = take;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        const int excl = incl - take;
        const int nl = __builtin_amdgcn_readlane(incl, nb - 1);      // wave-uniform (SGPRs)
        const int ntot = __builtin_amdgcn_readlane(incl, 2 * nb - 1);
        const int nr = ntot - nl;
        const int ncols = ntot + 1;

        // column position of lane j: every bucket lane writes the indices of its picks into the wave's scratch list
        // (at most k <= 10 per bucket), lane j reads entry j of the concatenated list
        int32_t col = v;
        {
            for (int i = 0; i < take; i++) nlist[wv][excl + i] = idx0 + i;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // other lanes' LDS writes are read back below (one wave: LDS is in order)
            __builtin_amdgcn_wave_barrier();
            const int jj = lane < nl ? lane : lane - 1;          // index into the concatenated neighbour list
            if (lane != nl && lane < ncols) col = a.nbr_pos[nlist[wv][jj]];
        }
        const bool active = lane < ncols;
        const int rc_col = active ? a.ref_code[(int64_t)col - a.ref_pos0] : 4;
        const int rc_centre = a.ref_code[(int64_t)v - a.ref_pos0];

        // ---- K4 + K3 in one pass over the tile's entries, 64 at a time.
        // Lanes as reads: which entries cover v (the pileup at v, generate_SNP_pileups.py:208), their base at v and strand
        // -> depth ballots.  Then the covering entries (a wave-uniform bit mask; the first maxcov in coordinate order)
        // are walked with SCALAR code: the entry record comes back from its lane with v_readlane, and lanes as columns
        // gather the code of that read at column j into a 4x4 histogram.  cnt[i] = per-lane counts of bases 0..3 at this column
        // among the reads whose centre base is i.  maxcov <= 255 (the small instantiation): four 8-bit fields in one
        // dword; otherwise four 16-bit fields in a qword.
        const int t = (v - a.tile_pos0) >> a.tile_shift;
        const int e0 = __builtin_amdgcn_readfirstlane(a.tile_off[t]), e1 = __builtin_amdgcn_readfirstlane(a.tile_off[t + 1]);
        const bool ok = ncols >= a.min_nbr_sites;               // :244, the list includes the candidate itself
        constexpr bool BYTE_CNT = CAP == MAXCOV_SMALL;
        using cnt_t = typename std::conditional<BYTE_CNT, uint32_t, unsigned long long>::type;
        constexpr int FIELD = BYTE_CNT ? 8 : 16;
        cnt_t cnt[4] = {0, 0, 0, 0};
        int n_all = 0;
        int fw[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0};
        for (int eb = e0; eb < e1; eb += 64) {
            const int e = eb + lane;
            bool cov = false;
            int code = 4, strand = 0;
            // this lane's read: first covered position, span, address of the code AT that position (read back by the
            // column phase below with v_readlane: no second trip to memory for the entry records)
            int32_t estart = 0;
            uint32_t elen = 0, erow_lo = 0, erow_hi = 0;
            if (e < e1) {
                const nc_tile_entry ent = a.tile_ent[e];
                cov = ent.start <= v && v < ent.end;
                const uint64_t row = (uint64_t)(uintptr_t)a.codes + (uint64_t)((ent.base_flag & ~int64_t(15)) + ent.start);
                estart = ent.start;
                elen = (uint32_t)(ent.end - ent.start);
                erow_lo = (uint32_t)row;
                erow_hi = (uint32_t)(row >> 32);
                if (cov) {
                    code = a.codes[(ent.base_flag & ~int64_t(15)) + v];
                    strand = (int)(ent.base_flag & 1);
                }
            }
            unsigned long long m = __ballot(cov);
            unsigned long long mb[4];                           // covering reads by their base at v (= the centre column)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                mb[b] = __ballot(cov && code == b);
                const unsigned long long f = __ballot(cov && code == b && strand == 0);
                fw[b] += __popcll(f);
                rv[b] += __popcll(mb[b] & ~f);
            }
            const int room = a.maxcov - n_all;                  // sampled reads still to take (wave-uniform)
            n_all += __popcll(m);
            if (!ok || room <= 0) continue;
            while (__popcll(m) > room) m &= ~(1ull << (63 - __builtin_clzll(m)));   // deeper than maxcov: the first maxcov in coordinate order
            // Reads are walked grouped by centre base k, so the histogram row is a compile-time register; reads deleted
            // at the centre (code 4) count nowhere and are skipped.
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned long long mk = mb[k] & m;
                while (mk != 0ull) {
                    // up to RND reads per round: every read's record is fetched from its lane (v_readlane) and its gather
                    // issued before the first loaded code is used
                    constexpr int RND = 16;
                    typedef const uint8_t __attribute__((address_space(1))) *gbyte_ptr;     // global (not flat) loads
                    int bcode[RND];
                    int nu = 0;
#pragma unroll
                    for (int u = 0; u < RND; u++) {
                        bcode[u] = 4;
                        if (mk != 0ull) {
                            const int idx = __builtin_ctzll(mk);
                            mk &= mk - 1ull;
                            nu = u + 1;
                            // wave-uniform: first covered position, span, and the address of the code AT that position
                            const int32_t rstart = __builtin_amdgcn_readlane(estart, idx);
                            const uint32_t rlen = (uint32_t)__builtin_amdgcn_readlane((int)elen, idx);
                            const gbyte_ptr rrow = (gbyte_ptr)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)erow_hi, idx) << 32) |
                                                                         (uint32_t)__builtin_amdgcn_readlane((int)erow_lo, idx));
                            const uint32_t off = (uint32_t)(col - rstart);                // covered <=> off < span (one unsigned compare)
                            if (active && off < rlen) bcode[u] = rrow[off];                 // scalar base + 32-bit lane offset
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RND; u++) {
                        // 64-bit shift: a code of 4 (not covered / deleted) shifts the one out of an 8-bit-field dword
                        const cnt_t inc = BYTE_CNT ? (cnt_t)(uint32_t)(1ull << (8 * bcode[u]))
                                                   : (bcode[u] < 4 ? ((cnt_t)1 << (16 * bcode[u])) : (cnt_t)0);
                        if (u < nu) cnt[k] += inc;
                    }
                }
            }
        }
        const int ns = min(n_all, a.maxcov);
        if constexpr (I16) {
            // ---- assemble (Appendix A step 5) in registers, one tensor column per lane
            const int o = NBR - nl;
            const int src = lane - o;                                        // the lane that walked tensor column `lane`
            const bool have = ok && lane < 41 && src >= 0 && src < ncols;
            const int sl = have ? src : lane;
            const int rc_s = __shfl(rc_col, sl, 64);
            cnt_t cs[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if constexpr (BYTE_CNT) cs[i] = (cnt_t)__shfl((int)cnt[i], sl, 64);
                else cs[i] = ((cnt_t)(uint32_t)__shfl((int)(cnt[i] >> 32), sl, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)cnt[i], sl, 64);
            }
            if (lane < 41) {
                int16_t *dst = reinterpret_cast<int16_t *>(a.x) + (int64_t)s * NC_SNP_TENSOR + lane * 5;     // 2-byte aligned
                typedef uint32_t __attribute__((aligned(2))) u32_a2;
                auto put_row = [&](int row, int v0, int v1, int v2, int v3, int v4) {
                    int16_t *d = dst + row * (41 * 5);
                    *reinterpret_cast<u32_a2 *>(d) = (uint32_t)(v0 & 0xffff) | ((uint32_t)v1 << 16);
                    *reinterpret_cast<u32_a2 *>(d + 2) = (uint32_t)(v2 & 0xffff) | ((uint32_t)v3 << 16);
                    d[4] = (int16_t)v4;
                };
                const int rcs = have ? rc_s : 4;
                put_row(0, rcs == 0, rcs == 1, rcs == 2, rcs == 3, 0);                   // row 0: reference one-hot
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int v[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int val = have ? (int)((cs[i] >> (FIELD * b)) & ((1 << FIELD) - 1)) : 0;
                        v[b] = b == rcs ? -val : val;
                    }
                    put_row(1 + i, v[0], v[1], v[2], v[3], (have && i == rc_centre) ? 1 : 0);
                }
            }
        } else if (ok) {
            // ---- assemble (Appendix A step 5)
            if (active) {
                const int o = NBR - nl;
                float *Xc = X + (o + lane) * 5;
                if (rc_col < 4) Xc[rc_col] = 1.0f;                                   // row 0: reference one-hot
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int val = (int)((cnt[i] >> (FIELD * b)) & ((1 << FIELD) - 1));
                        Xc[(1 + i) * 41 * 5 + b] = (float)(b == rc_col ? -val : val);
                    }
                    Xc[(1 + i) * 41 * 5 + 4] = (i == rc_centre) ? 1.0f : 0.0f;
                }
            }
        }
        if (lane == 0) {
            a.ref_out[s] = rc_centre;
            a.depth[s] = ns;
            a.valid[s] = ok ? 1 : 0;
        }
        if (lane < 4) {
            a.fwd[s * 4 + lane] = fw[lane];
            a.rev[s * 4 + lane] = rv[lane];
        }
        (void)nr;
    }
    if constexpr (I16) return;
    __syncthreads();
    // coalesced store of up to four site tensors
    const int s0 = blk * 4;
    const int nsite = min(4, a.n_sites - s0);
    if (a.x_i16) {
        // 4,100 int16 of up to four sites; 16-byte stores of 8 values (4 * 1025 = 512.5 groups of 8: the tail is scalar)
        int16_t *d16 = reinterpret_cast<int16_t *>(a.x) + (int64_t)s0 * NC_SNP_TENSOR;
        const int total = nsite * NC_SNP_TENSOR;
        for (int i = threadIdx.x * 8; i < total; i += 256 * 8) {
            if (i + 8 <= total) {
                union { int16_t h[8]; uint4 u; } o;
#pragma unroll
                for (int q = 0; q < 8; q++) o.h[q] = (int16_t)sm[(i + q) / NC_SNP_TENSOR][(i + q) % NC_SNP_TENSOR];
                *reinterpret_cast<uint4 *>(d16 + i) = o.u;
            } else {
                for (int q = i; q < total; q++) d16[q] = (int16_t)sm[q / NC_SNP_TENSOR][q % NC_SNP_TENSOR];
            }
        }
        return;
    }
    float *dst = a.x + (int64_t)s0 * NC_SNP_TENSOR;
    if (nsite == 4) {
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int i = threadIdx.x; i < NC_SNP_TENSOR; i += 256) {          // 4*1025 floats = 1025 float4
            const int f = i * 4;
            float4 o;
            o.x = sm[(f + 0) / NC_SNP_TENSOR][(f + 0) % NC_SNP_TENSOR];
            o.y = sm[(f + 1) / NC_SNP_TENSOR][(f + 1) % NC_SNP_TENSOR];
            o.z = sm[(f + 2) / NC_SNP_TENSOR][(f + 2) % NC_SNP_TENSOR];
            o.w = sm[(f + 3) / NC_SNP_TENSOR][(f + 3) % NC_SNP_TENSOR];
            d4[i] = o;
        }
    } else {
        for (int i = threadIdx.x; i < nsite * NC_SNP_TENSOR; i += 256) dst[i] = sm[i / NC_SNP_TENSOR][i % NC_SNP_TENSOR];
    }
}

// ---- [r5] the same tensors with lanes as (read, column) PAIRS.  k_featurize walks the sampled reads one after the other with scalar code (the
// entry record of a read by v_readlane, 41 of 64 lanes busy, ~45 vector + scalar instructions per read); here the covering reads' records go to an
// LDS list and the wave sweeps the reads x columns rectangle 64 pairs a step, four steps' gathers in flight: no scalar walk, every lane busy,
// counters by LDS atomic adds (at most the two reads that share a step meet on one counter).  int16 tensors, maxcov <= 255.
// MATES: the pack holds alignments that share read names (entries with bit 3 of base_flag; nc_snp_set_mates): such an entry is left out of a
// column's pileup when a later alignment of its name covers the column, and its row reads every column through named_code.
template <bool MATES>
__global__ __launch_bounds__(256) void k_featurize_pairs(FeatArgs a)
{
    __shared__ int32_t nlist[4][64];
    __shared__ uint4 rec[4][MAXCOV_SMALL];                         // sampled reads with a base at the centre: row address, first position, span | centre base << 28
    __shared__ int32_t colL[4][64];
    __shared__ uint32_t cntL[4][41 * 4 + 4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = (int)gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
    const int s = blk * 4 + wv;
    if (s >= a.n_sites) return;
    const int32_t v = __builtin_amdgcn_readfirstlane(a.site_pos[s]);
    const int32_t ch = __builtin_amdgcn_readfirstlane(a.site_chunk[s]);
    const int64_t win_lo = max((int64_t)1, (int64_t)a.chunk_start[ch] - NC_FLANK);
    const int64_t win_hi = (int64_t)a.chunk_end[ch] + NC_FLANK;
    const ModeTab &M = MODES[a.mode];
    const int nb = M.nb;
    // the pileup's first 64 entries and their bases at v are requested NOW: three dependent loads (tile range -> entry -> base) that run under
    // the neighbour search instead of behind it
    const int t = (v - a.tile_pos0) >> a.tile_shift;
    const int e0 = __builtin_amdgcn_readfirstlane(a.tile_off[t]), e1 = __builtin_amdgcn_readfirstlane(a.tile_off[t + 1]);
    nc_tile_entry pf_ent;
    pf_ent.start = 0; pf_ent.end = 0; pf_ent.base_flag = 0;
    int pf_code = 4;
    if (e0 + lane < e1) {
        pf_ent = a.tile_ent[e0 + lane];
        if (pf_ent.start <= v && v < pf_ent.end) pf_code = a.codes[(pf_ent.base_flag & ~int64_t(15)) + v];
    }
    const int rc_centre = a.ref_code[(int64_t)v - a.ref_pos0];
    // ---- K2 (as k_featurize)
    int take = 0, idx0 = 0;
    if (lane < 2 * nb) {
        const bool left = lane < nb;
        const Bucket B = M.b[left ? nb - 1 - lane : lane - nb];
        const int64_t dhi = min((int64_t)B.dhi, (int64_t)M.W - 1);
        int64_t pmin = left ? (int64_t)v - dhi : (int64_t)v + B.dlo + 1;
        int64_t pmax = left ? (int64_t)v - B.dlo - 1 : (int64_t)v + dhi;
        pmin = max(pmin, win_lo);
        pmax = min(pmax, win_hi);
        if (pmin <= pmax) {
            const int lo = lower_bound_i32(a.nbr_pos, a.n_nbr, pmin, a.cidx, a.cidx_pos0, a.n_cidx);
            const int hi = lower_bound_i32(a.nbr_pos, a.n_nbr, pmax + 1, a.cidx, a.cidx_pos0, a.n_cidx);
            take = min(hi - lo, B.k);
            const bool first = left ? (B.far != 0) : (B.far == 0);
            idx0 = first ? lo : hi - take;
        }
    }
    int incl = take;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const int y = __shfl_up(incl, o, 64);
        if (lane >= o) incl += y;
    }
    const int excl = incl - take;
    const int nl = __builtin_amdgcn_readlane(incl, nb - 1);
    const int ntot = __builtin_amdgcn_readlane(incl, 2 * nb - 1);
    const int ncols = ntot + 1;
    int32_t col = v;
    for (int i = 0; i < take; i++) nlist[wv][excl + i] = idx0 + i;
    for (int i = lane; i < 41 * 4; i += 64) cntL[wv][i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int jj = lane < nl ? lane : lane - 1;
        if (lane != nl && lane < ncols) col = a.nbr_pos[nlist[wv][jj]];
    }
    const bool active = lane < ncols;
    colL[wv][lane] = col;
    const int rc_col = active ? a.ref_code[(int64_t)col - a.ref_pos0] : 4;
    // ---- the pileup at v: depth ballots, and the sampled reads' records into the list
    const bool ok = ncols >= a.min_nbr_sites;
    int n_all = 0, nsel = 0;
    int fw[4] = {0, 0, 0, 0}, rv[4] = {0, 0, 0, 0};
    for (int eb = e0; eb < e1; eb += 64) {
        const int e = eb + lane;
        bool cov = false;
        int code = 4, strand = 0;
        uint4 rr = make_uint4(0, 0, 0, 0);
        if (e < e1) {
            nc_tile_entry ent = pf_ent;
            if (eb != e0) ent = a.tile_ent[e];
            cov = ent.start <= v && v < ent.end;
            const uint64_t row = (uint64_t)(uintptr_t)a.codes + (uint64_t)((ent.base_flag & ~int64_t(15)) + ent.start);
            rr = make_uint4((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)ent.start, (uint32_t)(ent.end - ent.start));
            if constexpr (MATES) {
                if (cov && (ent.base_flag & 8)) {
                    int self;
                    const int m = named_member(a, (ent.base_flag & ~int64_t(15)) + (ent.start & ~15), v, &self);
                    if (self >= 0) { cov = m == self; rr.w |= 0x80000000u; }                                      // m != self: replaced in this column
                }
            }
            if (cov) {
                code = eb == e0 ? pf_code : (int)a.codes[(ent.base_flag & ~int64_t(15)) + v];
                strand = (int)(ent.base_flag & 1);
            }
        }
        unsigned long long m = __ballot(cov);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned long long mb = __ballot(cov && code == b), f = __ballot(cov && code == b && strand == 0);
            fw[b] += __popcll(f);
            rv[b] += __popcll(mb & ~f);
        }
        const int room = a.maxcov - n_all;
        n_all += __popcll(m);
        if (!ok || room <= 0) continue;
        while (__popcll(m) > room) m &= ~(1ull << (63 - __builtin_clzll(m)));   // deeper than maxcov: the first maxcov in coordinate order
        const bool sel = ((m >> lane) & 1ull) && code < 4;                     // (a read deleted at the centre counts nowhere)
        const unsigned long long ms = __ballot(sel);
        if (sel) {
            rr.w |= (uint32_t)code << 28;
            rec[wv][nsel + __popcll(ms & ((1ull << lane) - 1ull))] = rr;
        }
        nsel += __popcll(ms);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- reads x columns, 64 pairs a step
    {
        typedef const uint8_t __attribute__((address_space(1))) *gbyte_ptr;
        const int P = nsel * ncols;
        const float inv = 1.0f / (float)ncols;
#ifndef NC_FEAT_U
#define NC_FEAT_U 8
#endif
        constexpr int U = NC_FEAT_U;                                   // steps whose gathers are in flight together
        for (int p0 = 0; p0 < P; p0 += 64 * U) {
            int jk[U];
            uint32_t bc[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int p = p0 + u * 64 + lane;
                const bool valid = p < P;
                const int r = (int)(((float)p + 0.5f) * inv), j = p - r * ncols;           // exact: p < 2^14, the product's error is far below 0.5 / 41
                const uint4 rr = rec[wv][valid ? r : 0];
                const uint32_t off = (uint32_t)(colL[wv][valid ? j : 0] - (int32_t)rr.z);
                bc[u] = 4;
                jk[u] = j * 4 + (int)((rr.w >> 28) & 7u);
                if (MATES && valid && (rr.w >> 31)) {
                    // (row address = codes + base + start; the slot starts at base + floor16(start))
                    const int64_t key = (int64_t)((((uint64_t)rr.y << 32) | rr.x) - (uint64_t)(uintptr_t)a.codes) - (int64_t)(int32_t)rr.z + ((int32_t)rr.z & ~15);
                    bc[u] = (uint32_t)named_code(a, key, colL[wv][j]);
                } else if (valid && off < (rr.w & 0x0fffffffu)) bc[u] = ((gbyte_ptr)(uintptr_t)(((uint64_t)rr.y << 32) | rr.x))[off];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (bc[u] < 4) atomicAdd(&cntL[wv][jk[u]], 1u << (8 * bc[u]));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int ns = min(n_all, a.maxcov);
    // ---- assemble (Appendix A step 5), one tensor column per lane
    {
        const int o = NBR - nl;
        const int src = lane - o;
        const bool have = ok && lane < 41 && src >= 0 && src < ncols;
        const int sl = have ? src : 0;
        const int rc_s = __shfl(rc_col, have ? src : lane, 64);
        if (lane < 41) {
            int16_t *dst = reinterpret_cast<int16_t *>(a.x) + (int64_t)s * NC_SNP_TENSOR + lane * 5;
            typedef uint32_t __attribute__((aligned(2))) u32_a2;
            auto put_row = [&](int row, int v0, int v1, int v2, int v3, int v4) {
                int16_t *d = dst + row * (41 * 5);
                *reinterpret_cast<u32_a2 *>(d) = (uint32_t)(v0 & 0xffff) | ((uint32_t)v1 << 16);
                *reinterpret_cast<u32_a2 *>(d + 2) = (uint32_t)(v2 & 0xffff) | ((uint32_t)v3 << 16);
                d[4] = (int16_t)v4;
            };
            const int rcs = have ? rc_s : 4;
            put_row(0, rcs == 0, rcs == 1, rcs == 2, rcs == 3, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t cs = have ? cntL[wv][sl * 4 + i] : 0u;
                int vv[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int val = (int)((cs >> (8 * b)) & 0xffu);
                    vv[b] = b == rcs ? -val : val;
                }
                put_row(1 + i, vv[0], vv[1], vv[2], vv[3], (have && i == rc_centre) ? 1 : 0);
            }
        }
    }
    if (lane == 0) {
        a.ref_out[s] = rc_centre;
        a.depth[s] = ns;
        a.valid[s] = ok ? 1 : 0;
    }
    if (lane < 4) {
        a.fwd[s * 4 + lane] = fw[lane];
        a.rev[s * 4 + lane] = rv[lane];
    }
}

// coarse index over the sorted neighbour sites: cidx[b] = lower_bound(nbr_pos, pos0 + b*1024)
__global__ void k_nbr_index(const int32_t *__restrict__ nbr_pos, int n_nbr, int32_t pos0, int n_cidx, int32_t *__restrict__ cidx)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_cidx) return;
    const int64_t key = (int64_t)pos0 + ((int64_t)b << NBR_IDX_SHIFT);
    int lo = 0, hi = n_nbr;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)nbr_pos[mid] < key) lo = mid + 1; else hi = mid;
    }
    cidx[b] = lo;
}

// per-chunk mean sampled depth (generate_SNP_pileups.py:274) and per-site scale (snpCaller.py:93-96)
__global__ __launch_bounds__(256) void k_chunk_scale(const int32_t *__restrict__ chunk_off, const int32_t *__restrict__ depth,
                                                     const uint8_t *__restrict__ valid, const int32_t *__restrict__ site_n,
                                                     double train_cov, int mode, double *__restrict__ scale,
                                                     double *__restrict__ chunk_depth)
{
    const int c = blockIdx.x;
    const int s0 = chunk_off[c], s1 = chunk_off[c + 1];
    long long sum = 0, cntv = 0;
    for (int s = s0 + threadIdx.x; s < s1; s += 256)
        if (valid[s]) { sum += depth[s]; cntv++; }
    __shared__ long long ssum[256], scnt[256];
    ssum[threadIdx.x] = sum;
    scnt[threadIdx.x] = cntv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
        __syncthreads();
    }
    const double mean = scnt[0] ? (double)ssum[0] / (double)scnt[0] : 0.0;       // np.mean of ints: exact sum / count
    if (threadIdx.x == 0) chunk_depth[c] = mean;
    for (int s = s0 + threadIdx.x; s < s1; s += 256)
        scale[s] = mode == 0 ? train_cov / mean : train_cov / (double)site_n[s];
}

}   // namespace

extern "C" {

int nc_snp_featurize(nc_ctx *ctx, const nc_readpack *pack, const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                     int32_t seq_mode, int32_t maxcov, int32_t min_nbr_sites, float *x_dev, int32_t *ref_code_out_dev,
                     int32_t *fwd_dp_dev, int32_t *rev_dp_dev, int32_t *site_depth_dev, uint8_t *valid_dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (!ctx->have_scan) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_featurize: call nc_snp_scan first");
    if (!pack || !ref_code_dev || seq_mode < 0 || seq_mode > 4)
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_featurize: bad argument");
    if (maxcov < 1 || maxcov > MAXCOV_CAP)
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_featurize: maxcov %d outside [1,%d]", maxcov, MAXCOV_CAP);
    if (ctx->n_sites == 0) return NC_OK;
    if (!x_dev || !ref_code_out_dev || !fwd_dp_dev || !rev_dp_dev || !site_depth_dev || !valid_dev)
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_featurize: null output");
    if (((uintptr_t)x_dev) & 15) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_featurize: x_dev must be 16-byte aligned");
    if (ref_pos0 != pack->tile_pos0 || (int64_t)ref_len < (int64_t)pack->n_tiles * pack->tile_size)
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_featurize: ref_code must cover the pack's tile grid");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    FeatArgs a;
    a.codes = pack->codes;
    a.tile_off = pack->tile_off;
    a.tile_ent = pack->tile_ent;
    a.tile_pos0 = pack->tile_pos0;
    a.tile_shift = pack->tile_size == 1024 ? 10 : pack->tile_size == 2048 ? 11 : 12;
    a.ref_code = ref_code_dev;
    a.ref_pos0 = ref_pos0;
    a.nbr_pos = (const int32_t *)ctx->nbr_pos.p;
    a.n_nbr = ctx->n_nbr;
    const int n_cidx = (int)(((int64_t)pack->n_tiles * pack->tile_size) >> NBR_IDX_SHIFT) + 2;
    NC_TRY(nc_ensure(ctx, ctx->nbr_idx, (size_t)n_cidx * 4));
    a.cidx = (const int32_t *)ctx->nbr_idx.p;
    a.cidx_pos0 = pack->tile_pos0;
    a.n_cidx = n_cidx;
    a.site_pos = (const int32_t *)ctx->site_pos.p;
    a.site_chunk = (const int32_t *)ctx->site_chunk.p;
    a.chunk_start = (const int32_t *)ctx->chunk_start.p;
    a.chunk_end = (const int32_t *)ctx->chunk_end.p;
    a.n_sites = ctx->n_sites;
    a.mode = seq_mode;
    a.maxcov = maxcov;
    a.min_nbr_sites = min_nbr_sites;
    a.x = x_dev;
    a.ref_out = ref_code_out_dev;
    a.fwd = fwd_dp_dev;
    a.rev = rev_dp_dev;
    a.depth = site_depth_dev;
    a.valid = valid_dev;
    a.x_i16 = ctx->x_i16 ? 1 : 0;
    a.mate_key = (const int64_t *)ctx->mate_key;
    a.mate_rec = (const int4 *)ctx->mate_rec;
    a.n_mates = ctx->n_mates;
    const char *fp = getenv("NC_FEAT_PAIRS");
    const bool pairs = maxcov < MAXCOV_SMALL && a.x_i16 && !(fp && atoi(fp) == 0);
    if (a.n_mates > 0 && !pairs)
        return nc_fail(ctx, NC_ERR_UNSUPPORTED, "nc_snp_featurize: alignments that share read names (nc_snp_set_mates) need the int16 tensor format and maxcov < %d", MAXCOV_SMALL);
    NcTimer tm(ctx, 1);
    hipLaunchKernelGGL(k_nbr_index, dim3((n_cidx + 255) / 256), dim3(256), 0, ctx->stream, a.nbr_pos, a.n_nbr, a.cidx_pos0, n_cidx,
                       (int32_t *)ctx->nbr_idx.p);
    const dim3 grid((ctx->n_sites + 3) / 4);
    if (maxcov < MAXCOV_SMALL) {                            // 8-bit counter fields: at most 255 sampled reads
        // int16 tensors (the product path): lanes as (read, column) pairs (k_featurize_pairs); NC_FEAT_PAIRS=0: the scalar-driven read walk
        // (read per call: A/B checks)
        if (pairs && a.n_mates > 0) hipLaunchKernelGGL(k_featurize_pairs<true>, grid, dim3(256), 0, ctx->stream, a);
        else if (pairs) hipLaunchKernelGGL(k_featurize_pairs<false>, grid, dim3(256), 0, ctx->stream, a);
        else if (a.x_i16) hipLaunchKernelGGL((k_featurize<MAXCOV_SMALL, true>), grid, dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_featurize<MAXCOV_SMALL, false>), grid, dim3(256), 0, ctx->stream, a);
    } else {
        if (a.x_i16) hipLaunchKernelGGL((k_featurize<MAXCOV_CAP, true>), grid, dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_featurize<MAXCOV_CAP, false>), grid, dim3(256), 0, ctx->stream, a);
    }
    NC_HIP(ctx, hipGetLastError());
    tm.stop();
    return NC_OK;
}

int nc_snp_set_mates(nc_ctx *ctx, int32_t n_mates, const int64_t *d_mate_key, const int32_t *d_mate_rec)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_mates < 0 || (n_mates > 0 && (!d_mate_key || !d_mate_rec || ((uintptr_t)d_mate_rec & 15))))
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_set_mates: bad argument");
    ctx->n_mates = n_mates;
    ctx->mate_key = n_mates ? d_mate_key : nullptr;
    ctx->mate_rec = n_mates ? d_mate_rec : nullptr;
    return NC_OK;
}

int nc_snp_scale(nc_ctx *ctx, const int32_t *site_depth_dev, const uint8_t *valid_dev, double train_coverage, int32_t mode,
                 double *scale_dev, double *chunk_depth_host)
{
    if (!ctx) return NC_ERR_ARG;
    if (!ctx->have_scan) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_scale: call nc_snp_scan first");
    if (mode != 0 && mode != 1) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_scale: mode must be 0 or 1");
    NC_TRY(nc_ensure(ctx, ctx->chunk_depth, (size_t)ctx->n_chunks * 8));
    if (ctx->n_sites > 0) {
        if (!site_depth_dev || !valid_dev || !scale_dev) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_scale: null argument");
        hipLaunchKernelGGL(k_chunk_scale, dim3(ctx->n_chunks), dim3(256), 0, ctx->stream, (const int32_t *)ctx->chunk_off.p,
                           site_depth_dev, valid_dev, (const int32_t *)ctx->site_n.p, train_coverage, mode, scale_dev,
                           (double *)ctx->chunk_depth.p);
        NC_HIP(ctx, hipGetLastError());
    } else {
        NC_HIP(ctx, hipMemsetAsync(ctx->chunk_depth.p, 0, (size_t)ctx->n_chunks * 8, ctx->stream));
    }
    if (chunk_depth_host) {
        NC_HIP(ctx, hipMemcpyAsync(chunk_depth_host, ctx->chunk_depth.p, (size_t)ctx->n_chunks * 8, hipMemcpyDeviceToHost,
                                   ctx->stream));
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NC_OK;
}

int nc_snp_chunk_depth_async(nc_ctx *ctx, void *copy_stream, double *chunk_depth_host_pinned)
{
    if (!ctx) return NC_ERR_ARG;
    if (!copy_stream || !chunk_depth_host_pinned) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_chunk_depth_async: null argument");
    if (!ctx->have_scan || !ctx->chunk_depth.p) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_chunk_depth_async: call nc_snp_scale first");
    if (!ctx->scale_ev) NC_HIP(ctx, hipEventCreateWithFlags(&ctx->scale_ev, hipEventDisableTiming));
    NC_HIP(ctx, hipEventRecord(ctx->scale_ev, ctx->stream));
    NC_HIP(ctx, hipStreamWaitEvent((hipStream_t)copy_stream, ctx->scale_ev, 0));
    NC_TRY(nc_d2h(ctx, chunk_depth_host_pinned, ctx->chunk_depth.p, (size_t)ctx->n_chunks * 8, (hipStream_t)copy_stream));
    return NC_OK;
}

}   // extern "C"
