// K8: MSA rows -> (5,128,2) indel tensor (gfx950).
//
// Restates the histogram half of msa() (reference generate_indel_pileups.py:57-71): per alignment column the
// symbol histogram over the read rows, normalised by the row count in f32, minus the aligned reference's
// one-hot; the consensus symbol is the arg-max with the gap handicapped by 0.01.  One workgroup per read set,
// one lane per alignment column, rows streamed with coalesced byte loads (row-major rows: lane = column).
#include <vector>

#include "nc_common.h"

namespace {

__global__ __launch_bounds__(256) void k_indel_tensor(const uint8_t *__restrict__ rows, const int64_t *__restrict__ row_off,
                                                      const int32_t *__restrict__ n_rows, const int32_t *__restrict__ n_cols,
                                                      const uint8_t *__restrict__ ref_rows, const int64_t *__restrict__ ref_off,
                                                      int max_cols, float *__restrict__ x, uint8_t *__restrict__ cns)
{
    const int s = blockIdx.x;
    const int nr = n_rows[s], nc = n_cols[s];
    const uint8_t *R = rows + row_off[s];
    const uint8_t *ref = ref_rows + ref_off[s];
    float *X = x + (int64_t)s * 5 * 128 * 2;
    for (int c = threadIdx.x; c < max_cols; c += 256) {
        if (c < nc) {
            int h[5] = {0, 0, 0, 0, 0};
            for (int r = 0; r < nr; r++) {
                const int sym = R[(int64_t)r * nc + c];
#pragma unroll
                for (int k = 0; k < 5; k++) h[k] += sym == k;
            }
            const float tot = (float)(h[0] + h[1] + h[2] + h[3] + h[4]);
            float alt[5], best = -1e30f;
            int arg = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                alt[k] = (float)h[k] / tot;                           // f32 divide (:58-59)
                const float tv = k == 4 ? alt[k] - 0.01f : alt[k];      // :62
                if (tv > best) { best = tv; arg = k; }                 // np.argmax: first maximum (:64)
            }
            cns[(int64_t)s * max_cols + c] = (uint8_t)arg;
            if (c < 128) {
                const int rc = ref[c];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const float rf = rc == k ? 1.0f : 0.0f;
                    X[(k * 128 + c) * 2 + 0] = alt[k] - rf;            // :67
                    X[(k * 128 + c) * 2 + 1] = rf;
                }
            }
        } else {
            cns[(int64_t)s * max_cols + c] = NC_CODE_ABSENT;
            if (c < 128) {
#pragma unroll
                for (int k = 0; k < 5; k++) {                           // zero padding (:70-71)
                    X[(k * 128 + c) * 2 + 0] = 0.0f;
                    X[(k * 128 + c) * 2 + 1] = 0.0f;
                }
            }
        }
    }
}

}   // namespace

extern "C" int nc_indel_tensor(nc_ctx *ctx, int32_t n_sets, const uint8_t *rows_dev, const int64_t *row_off_dev,
                               const int32_t *n_rows_dev, const int32_t *n_cols_dev, const uint8_t *ref_rows_dev,
                               const int64_t *ref_off_dev, int32_t max_cols, float *x_dev, uint8_t *cns_dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_sets < 0 || max_cols < 128) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_tensor: n_sets >= 0 and max_cols >= 128 required");
    if (n_sets == 0) return NC_OK;
    if (!rows_dev || !row_off_dev || !n_rows_dev || !n_cols_dev || !ref_rows_dev || !ref_off_dev || !x_dev || !cns_dev)
        return nc_fail(ctx, NC_ERR_ARG, "nc_indel_tensor: null argument");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    NcTimer tm(ctx, 3);
    hipLaunchKernelGGL(k_indel_tensor, dim3(n_sets), dim3(256), 0, ctx->stream, rows_dev, row_off_dev, n_rows_dev, n_cols_dev,
                       ref_rows_dev, ref_off_dev, max_cols, x_dev, cns_dev);
    NC_HIP(ctx, hipGetLastError());
    tm.stop();
    return NC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// K7: indel candidate window scan (pass 1 of get_indel_testing_candidates, reference generate_indel_pileups.py:197-276)
namespace {

__device__ __forceinline__ uint32_t lut8i(uint32_t x, uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, x); }

// All chunks of a call run in the same launches (chunk = a grid dimension), each chunk with its own workspace slice and
// the reference's per-chunk semantics (window deques start empty at the chunk's first column)
// IndelChunk: nc_common.h (shared with the device-resident pipeline, nc_pipe.hip)
__device__ __forceinline__ int32_t *ck_depth(char *ws, const IndelChunk &c) { return (int32_t *)(ws + c.ws); }
__device__ __forceinline__ int32_t *ck_rank(char *ws, const IndelChunk &c) { return (int32_t *)(ws + c.ws) + (int64_t)3 * c.ncol; }
__device__ __forceinline__ int32_t *ck_diff(char *ws, const IndelChunk &c) { return (int32_t *)(ws + c.ws) + (int64_t)3 * c.ncol + c.ncol + 1; }
// impute_indel_phase only: per column, over ALL kept reads: [0] reads deleted here ('*'), [1] insertions / [2] deletions that follow this column
__device__ __forceinline__ int32_t *ck_cnt(char *ws, const IndelChunk &c) { return ck_diff(ws, c) + (int64_t)8 * c.nd; }

// chunk of every tile-block (k_hap_depth_b's block index; k_event_tiles' / SPT): a workgroup finds its chunk by one load instead of a bisection of
// ten dependent scalar loads over the descriptors
__global__ void k_blk_chunks(const IndelChunk *__restrict__ ck, int32_t n_chunks, int32_t nblk, int32_t *__restrict__ blk_chunk)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int b0 = ck[c].blk0, b1 = c + 1 < n_chunks ? ck[c + 1].blk0 : nblk;
    for (int b = b0; b < b1; b++) blk_chunk[b] = c;
}

// per-column depth by haplotype tag; same access scheme as k_scan (one aligned dwordx4 of codes per read and lane).
// STAR: count the reads whose code is 4 (deleted at this column) of all haplotypes instead, into cnt[0]
template <int BLOCK, bool STAR>
__global__ __launch_bounds__(BLOCK) void k_hap_depth_b(const uint8_t *__restrict__ codes, const int32_t *__restrict__ tile_off,
                                                       const nc_tile_entry *__restrict__ tile_ent, int32_t tile_pos0,
                                                       const IndelChunk *__restrict__ ck, const int32_t *__restrict__ blk_chunk, char *__restrict__ ws, int32_t haploid,
                                                       const uint8_t *__restrict__ excl, int32_t grid_lo, int32_t *__restrict__ blk_yield)
{
    constexpr int TILE = BLOCK * 16;
    const IndelChunk c = ck[blk_chunk[blockIdx.x]];
    const int t = c.tile0 + ((int)blockIdx.x - c.blk0);
    const int32_t lo = c.lo, hi = c.hi, ncol = c.ncol;
    int32_t *depth = STAR ? ck_cnt(ws, c) : ck_depth(ws, c);
    const int32_t P0 = tile_pos0 + t * TILE + threadIdx.x * 16;
    uint32_t acc[3][4], wide[3][8];
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int d = 0; d < 4; d++) acc[q][d] = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) wide[q][d] = 0;
    }
    const int e0 = tile_off[t], e1 = tile_off[t + 1];
    int e = e0;
    while (e < e1) {
        const int lim = min(e1, e + 255);
#pragma unroll 8
        for (; e < lim; e++) {                                         // (eight reads' loads in flight, as in k_scan)
            const nc_tile_entry ent = tile_ent[e];
            const int32_t slo = ent.start & ~15, shi = (ent.end + 15) & ~15;
            uint4 v = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
            if (P0 >= slo && P0 < shi) v = *reinterpret_cast<const uint4 *>(codes + (ent.base_flag & ~int64_t(15)) + P0);
            const int hp = __builtin_amdgcn_readfirstlane((int)((ent.base_flag >> 1) & 3));           // wave-uniform: the plane is chosen by scalar branches
            const int plane = (STAR || haploid) ? 0 : hp == 1 ? 0 : hp == 2 ? 1 : 2;       // haploid: one read set, tags ignored
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t pres[4];
#pragma unroll
            for (int d = 0; d < 4; d++) pres[d] = STAR ? lut8i(w[d], 0u, 0x00000001u) : lut8i(w[d], 0x01010101u, 0x00000001u);   // code 4 | codes 0..4 -> 1
            if (STAR || plane == 0) {
#pragma unroll
                for (int d = 0; d < 4; d++) acc[0][d] += pres[d];
            } else if (plane == 1) {
#pragma unroll
                for (int d = 0; d < 4; d++) acc[1][d] += pres[d];
            } else {
#pragma unroll
                for (int d = 0; d < 4; d++) acc[2][d] += pres[d];
            }
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                wide[q][2 * d] += acc[q][d] & 0x00FF00FFu;
                wide[q][2 * d + 1] += (acc[q][d] >> 8) & 0x00FF00FFu;
                acc[q][d] = 0;
            }
    }
    // a lane holds 16 consecutive positions: through LDS (17-word pitch) so that a store instruction writes 64 consecutive positions
    // (a lane storing its own 16 made every instruction 64 four-byte pieces of 64 different lines: 1.26 GB written for 0.77)
    __shared__ int32_t tr[STAR ? 1 : 3][BLOCK * 17];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int d = i >> 2, k = i & 3, wi = 2 * d + (k & 1), sh = (k >> 1) * 16;
#pragma unroll
        for (int q = 0; q < (STAR ? 1 : 3); q++) tr[q][threadIdx.x * 17 + i] = (int32_t)((wide[q][wi] >> sh) & 0xFFFF);
    }
    __syncthreads();
    const int32_t T0 = tile_pos0 + t * TILE;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int j = r * BLOCK + threadIdx.x;                        // position T0 + j = lane j / 16, element j % 16
        const int32_t p = T0 + j;
        if (p < lo || p > hi) continue;
        if (!STAR && blk_yield) {                                     // the tiled route: 16-bit rows (the counters above are 16-bit fields already)
#pragma unroll
            for (int q = 0; q < 3; q++) reinterpret_cast<uint16_t *>(depth)[(int64_t)q * ncol + (p - lo)] = (uint16_t)tr[q][(j >> 4) * 17 + (j & 15)];
        } else {
#pragma unroll
            for (int q = 0; q < (STAR ? 1 : 3); q++) depth[(int64_t)q * ncol + (p - lo)] = tr[q][(j >> 4) * 17 + (j & 15)];
        }
    }
    if constexpr (!STAR) {
        if (!blk_yield) return;
        // The rank of a column among the chunk's yielded ones (depth > 0, not excluded), LOCAL to this tile-block: a lane has its 16 columns' depths
        // in registers.  The block's count goes to blk_yield, k_blk_base scans the counts of a chunk's blocks, and k_event_tiles adds the base of a
        // column's block (k_yield_rank_b walked a chunk in 98 dependent rounds of load -> scan: 0.30 ms per chr20-sized contig, all latency).
        int fl[16], sum = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int d = i >> 2, k = i & 3, wi = 2 * d + (k & 1), sh = (k >> 1) * 16;
            const int32_t p = P0 + i;
            const int tot = (int)(((wide[0][wi] >> sh) & 0xFFFF) + ((wide[1][wi] >> sh) & 0xFFFF) + ((wide[2][wi] >> sh) & 0xFFFF));
            bool y = p >= lo && p <= hi && tot > 0;
            if (y && excl) y = excl[(lo - grid_lo) + (p - lo)] == 0;
            fl[i] = y;
            sum += y;
        }
        __shared__ int wtot[BLOCK / 64];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int inc = sum;
        inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();                                              // (also: every depth row has left tr)
        int wp = 0, all = 0;
#pragma unroll
        for (int q = 0; q < BLOCK / 64; q++) {
            if (q < wv) wp += wtot[q];
            all += wtot[q];
        }
        int run = wp + inc - sum;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            tr[0][threadIdx.x * 17 + i] = fl[i] ? run : -1;
            run += fl[i];
        }
        if (threadIdx.x == 0) blk_yield[blockIdx.x] = all;
        __syncthreads();
        int16_t *rank = reinterpret_cast<int16_t *>(ck_rank(ws, c));   // (local to the block: < 4096; -1 = not yielded)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = r * BLOCK + threadIdx.x;
            const int32_t p = T0 + j;
            if (p < lo || p > hi) continue;
            rank[p - lo] = (int16_t)tr[0][(j >> 4) * 17 + (j & 15)];
        }
    }
}

// exclusive scan of the yielded-column counts of a chunk's tile-blocks (one wave per chunk) -> the rank of the first yielded column of every block
__global__ __launch_bounds__(256) void k_blk_base(const IndelChunk *__restrict__ ck, int32_t n_chunks, int32_t nblk, const int32_t *__restrict__ blk_yield,
                                                  int32_t *__restrict__ blk_base, char *__restrict__ ws)
{
    const int ci = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ci >= n_chunks) return;
    const IndelChunk c = ck[ci];
    const int b0 = c.blk0, b1 = ci + 1 < n_chunks ? ck[ci + 1].blk0 : nblk;
    int carry = 0;
    for (int b = b0; b < b1; b += 64) {
        const int v = b + lane < b1 ? blk_yield[b + lane] : 0;
        int inc = v;
        inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
        if (b + lane < b1) blk_base[b + lane] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) ck_rank(ws, c)[c.ncol] = carry;                    // ny, as k_yield_rank_b leaves it
}

// block-wide scan helper shared by the two per-chunk scans below: returns this thread's inclusive prefix inside the block
// and the block total (16 waves)
__device__ __forceinline__ int block_scan_1024(int v, int *wsum, int &tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
    inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int wp = 0;
    tot = 0;
    for (int w = 0; w < 16; w++) {
        const int s = wsum[w];
        if (w < wv) wp += s;
        tot += s;
    }
    return wp + inc;
}

// one workgroup per chunk: exclusive scan of the "column is yielded" flag -> rank among yielded columns; ny at rank[ncol]
__global__ __launch_bounds__(1024) void k_yield_rank_b(const IndelChunk *__restrict__ ck, char *__restrict__ ws, const uint8_t *__restrict__ excl,
                                                       int32_t grid_lo)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const IndelChunk c = ck[blockIdx.x];
    const int32_t ncol = c.ncol;
    const int32_t *depth = ck_depth(ws, c);
    int32_t *rank = ck_rank(ws, c);
    const int32_t excl_off = c.lo - grid_lo;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < ncol; base += 1024) {
        const int i = base + threadIdx.x;
        int v = 0;
        if (i < ncol) {
            const int tot = depth[i] + depth[ncol + i] + depth[2 * (int64_t)ncol + i];
            v = tot > 0 && !(excl && excl[excl_off + i]);
        }
        int tot;
        const int inc = block_scan_1024(v, wsum, tot);
        const int cc = carry;
        if (i < ncol) rank[i] = v ? cc + inc - v : -1;               // -1: not yielded
        __syncthreads();
        if (threadIdx.x == 0) carry = cc + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) rank[ncol] = carry;
}

// one thread per kept read: merge the window-end intervals [e, e+w-1] of its qualifying events (a read counts once per
// window, set-union semantics of :254-264) and add them to the per-(class, haplotype) difference arrays of every chunk
// its events fall into (chunks are ascending and may overlap)
__global__ void k_event_intervals_b(int32_t n_reads, const int32_t *__restrict__ ev_off, const int32_t *__restrict__ ev_pos,
                                    const int32_t *__restrict__ ev_len, const uint8_t *__restrict__ read_hap,
                                    const IndelChunk *__restrict__ ck, int32_t n_chunks, char *__restrict__ ws, int32_t win, int32_t small_win,
                                    int32_t haploid, int32_t impute)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int hp = read_hap[r];
    const bool tagged = haploid || hp == 1 || hp == 2;
    if (!tagged && !impute) return;
    const int h = haploid ? 0 : hp - 1;
    const int ea = ev_off[r], eb = ev_off[r + 1];
    if (ea >= eb) return;
    const int32_t p_first = ev_pos[ea], p_last = ev_pos[eb - 1];
    int a = 0, b = n_chunks;                                       // first chunk with hi >= p_first
    while (a < b) {
        const int m = (a + b) >> 1;
        if (ck[m].hi < p_first) a = m + 1; else b = m;
    }
    for (int ci = a; ci < n_chunks && ck[ci].lo <= p_last; ci++) {
        const IndelChunk c = ck[ci];
        if (impute) {                                                 // :279-284: '+' / '-' in the column's pileup strings, every read
            int32_t *cnt = ck_cnt(ws, c);
            for (int e = ea; e < eb; e++) {
                const int32_t p = ev_pos[e];
                if (p >= c.lo && p <= c.hi) atomicAdd(&cnt[(int64_t)(ev_len[e] > 0 ? 1 : 2) * c.ncol + (p - c.lo)], 1);
            }
        }
        if (!tagged) continue;
        const int32_t *rank = ck_rank(ws, c);
        int32_t *diff = ck_diff(ws, c);
        const int32_t nd = c.nd;
        int cur_lo[4] = {-1, -1, -1, -1}, cur_hi[4] = {-1, -1, -1, -1};
        for (int e = ea; e < eb; e++) {
            const int32_t p = ev_pos[e];
            if (p < c.lo || p > c.hi) continue;
            const int k = rank[p - c.lo];
            if (k < 0) continue;                                      // excluded column
            const int32_t sl = ev_len[e], ln = sl < 0 ? -sl : sl;
            const bool ins = sl > 0;
#pragma unroll
            for (int cls = 0; cls < 4; cls++) {
                const bool q = cls < 2 ? (ln > 2 && ln <= 50 && ins == (cls == 1)) : (ln <= 10 && ins == (cls == 3));
                if (!q) continue;
                const int w = cls < 2 ? win : small_win;
                if (cur_lo[cls] >= 0 && k <= cur_hi[cls]) cur_hi[cls] = k + w - 1;
                else {
                    if (cur_lo[cls] >= 0) {
                        atomicAdd(&diff[(int64_t)(cls * 2 + h) * nd + cur_lo[cls]], 1);
                        atomicAdd(&diff[(int64_t)(cls * 2 + h) * nd + cur_hi[cls] + 1], -1);
                    }
                    cur_lo[cls] = k;
                    cur_hi[cls] = k + w - 1;
                }
            }
        }
#pragma unroll
        for (int cls = 0; cls < 4; cls++)
            if (cur_lo[cls] >= 0) {
                atomicAdd(&diff[(int64_t)(cls * 2 + h) * nd + cur_lo[cls]], 1);
                atomicAdd(&diff[(int64_t)(cls * 2 + h) * nd + cur_hi[cls] + 1], -1);
            }
    }
}

// The same, one WAVE per kept read with the read's events across the lanes (k_event_intervals_b walks a read's ~400 events in one
// thread: 10 ms per chr20-sized contig, latency-bound).  The merged intervals of a (read, class, chunk) are the union of equal-length
// intervals [k, k + w - 1] over its qualifying events in rank order: an event OPENS an interval iff no qualifying event of the class
// precedes it within w - 1 ranks, and CLOSES one (at k + w) iff none follows within w - 1 ranks -- two local look-ups per event.
__global__ __launch_bounds__(256) void k_event_intervals_w(int32_t n_reads, const int32_t *__restrict__ ev_off, const int32_t *__restrict__ ev_pos,
                                                           const int32_t *__restrict__ ev_len, const uint8_t *__restrict__ read_hap,
                                                           const IndelChunk *__restrict__ ck, int32_t n_chunks, char *__restrict__ ws, int32_t win,
                                                           int32_t small_win, int32_t haploid, int32_t impute)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const int hp = read_hap[r];
    const bool tagged = haploid || hp == 1 || hp == 2;
    if (!tagged && !impute) return;
    const int h = haploid ? 0 : hp - 1;
    const int ea = ev_off[r], eb = ev_off[r + 1];
    auto qualifies = [](int32_t sl, int cls) {
        const int32_t ln = sl < 0 ? -sl : sl;
        const bool ins = sl > 0;
        return cls < 2 ? (ln > 2 && ln <= 50 && ins == (cls == 1)) : (ln <= 10 && ins == (cls == 3));
    };
    if (ea >= eb) return;
    // first chunk with hi >= the read's first event: one search per read on the scalar unit; an event's own first chunk is at most a
    // few steps further (a search per event was ten dependent loads, half of this kernel's time)
    int a0 = 0;
    {
        const int32_t p_first = __builtin_amdgcn_readfirstlane(ev_pos[ea]);
        int b = n_chunks;
        while (a0 < b) {
            const int m = (a0 + b) >> 1;
            if (ck[m].hi < p_first) a0 = m + 1; else b = m;
        }
    }
    for (int e = ea + lane; e < eb; e += 64) {
        const int32_t p = ev_pos[e], sl = ev_len[e];
        int a = a0;
        while (a < n_chunks && ck[a].hi < p) a++;
        for (int ci = a; ci < n_chunks && ck[ci].lo <= p; ci++) {
            const IndelChunk c = ck[ci];
            if (impute) atomicAdd(&ck_cnt(ws, c)[(int64_t)(sl > 0 ? 1 : 2) * c.ncol + (p - c.lo)], 1);      // :279-284, every read
            if (!tagged) continue;
            const int32_t *rank = ck_rank(ws, c);
            const int k = rank[p - c.lo];
            if (k < 0) continue;                                      // excluded column
            int32_t *diff = ck_diff(ws, c);
#pragma unroll
            for (int cls = 0; cls < 4; cls++) {
                if (!qualifies(sl, cls)) continue;
                const int w = cls < 2 ? win : small_win;
                bool has_prev = false, has_next = false;
                for (int e2 = e - 1; e2 >= ea; e2--) {
                    const int32_t p2 = ev_pos[e2];
                    if (p2 < c.lo) break;
                    const int k2 = rank[p2 - c.lo];
                    if (k2 < 0) continue;
                    if (k - k2 > w - 1) break;
                    if (qualifies(ev_len[e2], cls)) { has_prev = true; break; }
                }
                for (int e2 = e + 1; e2 < eb; e2++) {
                    const int32_t p2 = ev_pos[e2];
                    if (p2 > c.hi) break;
                    const int k2 = rank[p2 - c.lo];
                    if (k2 < 0) continue;
                    if (k2 - k > w - 1) break;
                    if (qualifies(ev_len[e2], cls)) { has_next = true; break; }
                }
#ifdef NC_ABL_EV_NOATOMIC
                if (!has_prev && !has_next && k == -12345) diff[0] = 1;
#else
                if (!has_prev) atomicAdd(&diff[(int64_t)(cls * 2 + h) * c.nd + k], 1);
                if (!has_next) atomicAdd(&diff[(int64_t)(cls * 2 + h) * c.nd + k + w], -1);
#endif
            }
        }
    }
}

// The same result without global atomics (k_event_intervals_w spends 4.8 of its 5.1 ms per chr20-sized contig in ~150 M scattered
// atomic adds): one workgroup per 1024 columns of a chunk.  It takes the reads of the tile index that overlap its columns (and the
// columns before them whose windows reach in: the last wmax - 1 yielded columns), finds each read's events there by bisection, gives
// every event to a lane, accumulates the interval ends in LDS (clipped to the block's ranks: what a read covers INSIDE the block does
// not depend on events outside the margin), scans the eight rows into the window counts U[class, haplotype][rank] and takes the
// columns' decisions from them: k_prefix_rows_b and k_indel_decide_b have nothing left to do, the counts never reach HBM (2 GB
// written and 2 GB read per chr20-sized contig) and the workspace needs no zeroing.  Needs the map tile entry -> read (slot_off of the wire pack): the device pipeline has
// it (nc_indel_sites_plan); the host-route API keeps k_event_intervals_w.
#ifndef NC_EV_NT
#define NC_EV_NT 512
#endif
// per-column decision of :252-275 (float64 divide-and-compare, as in the reference) from the depths n0 / n1 of the two haplotypes (haploid:
// n0 = all reads) and the window counts U(class, haplotype) at the column's rank
template <class UF>
__device__ __forceinline__ int8_t indel_decide(int k, int n0, int n1, UF U, int32_t mincov, double ins_t, double del_t, int32_t haploid)
{
    if (haploid) {
        if (k >= 0 && n0 >= mincov && n0 > 0) {
            double f[4];
#pragma unroll
            for (int cls = 0; cls < 4; cls++) f[cls] = (double)U(cls, 0) / (double)n0;
            if (f[0] >= del_t || f[1] >= ins_t) return 0;
            if (f[2] >= del_t || f[3] >= ins_t || (f[2] + f[3]) >= 0.9) return 1;
        }
    } else if (k >= 0 && n0 >= mincov && n1 >= mincov) {
        double f[4][2];
#pragma unroll
        for (int cls = 0; cls < 4; cls++) {
            f[cls][0] = n0 > 0 ? (double)U(cls, 0) / (double)n0 : 0.0;
            f[cls][1] = n1 > 0 ? (double)U(cls, 1) / (double)n1 : 0.0;
        }
        if (fmax(f[0][0], f[0][1]) >= del_t || fmax(f[1][0], f[1][1]) >= ins_t) return 0;
        if (fmax(f[2][0], f[2][1]) >= del_t || fmax(f[3][0], f[3][1]) >= ins_t || (f[2][0] + f[3][0]) >= 0.9 || (f[2][1] + f[3][1]) >= 0.9) return 1;
    }
    return -1;
}

// Decisions without divisions.  fl(U / n) >= t is monotone in the integer U, so for every depth n < DEC_N there is a smallest count that passes:
// k_decide_tables finds it with the reference's own float64 divide-and-compare (bisection over U; 65535 = none), once per call for del_t and
// ins_t, and a column's eight ratio tests become table look-ups.  The sum rule (f2 + f3 >= 0.9) is decided in integers when the exact sum is not
// 0.9 itself (then it is at least 1 / (10 n) > 1e-6 away, against rounding errors below 1e-12), else by the float64 expression.  indel_decide
// (the division form) stays for depths beyond the table and as the other routes' kernel: test_tiled_event_windows_equal_the_atomic_form compares them.
constexpr int DEC_N = 1024;
__global__ void k_decide_tables(double del_t, double ins_t, uint16_t *__restrict__ tab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * DEC_N) return;
    const int n = i % DEC_N;
    const double t = i < DEC_N ? del_t : ins_t;
    int v;
    if (n == 0) v = 0.0 >= t ? 0 : 65535;                            // (n == 0: the ratio is 0.0 by definition)
    else {
        int lo = 0, hi = 65535;                                       // smallest U in [0, 65535) with U / n >= t
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((double)mid / (double)n >= t) hi = mid; else lo = mid + 1;
        }
        v = lo;
    }
    tab[i] = (uint16_t)v;
}
template <class UF>
__device__ __forceinline__ int8_t indel_decide_tab(int k, int n0, int n1, UF U, int32_t mincov, double ins_t, double del_t, int32_t haploid,
                                                   const uint16_t *tdel, const uint16_t *tins, int dec_n)
{
    if (n0 >= dec_n || n1 >= dec_n) return indel_decide(k, n0, n1, U, mincov, ins_t, del_t, haploid);
    auto sum_rule = [&](int u2, int u3, int n) {
        if (n <= 0) return false;
        const int a = 10 * (u2 + u3), b = 9 * n;
        if (a != b) return a > b;
        return ((double)u2 / (double)n + (double)u3 / (double)n) >= 0.9;
    };
    if (haploid) {
        if (k >= 0 && n0 >= mincov && n0 > 0) {
            const int td = tdel[n0], ti = tins[n0];
            if (U(0, 0) >= td || U(1, 0) >= ti) return 0;
            if (U(2, 0) >= td || U(3, 0) >= ti || sum_rule(U(2, 0), U(3, 0), n0)) return 1;
        }
    } else if (k >= 0 && n0 >= mincov && n1 >= mincov) {
        const int td0 = tdel[n0], ti0 = tins[n0], td1 = tdel[n1], ti1 = tins[n1];
        if (U(0, 0) >= td0 || U(0, 1) >= td1 || U(1, 0) >= ti0 || U(1, 1) >= ti1) return 0;
        if (U(2, 0) >= td0 || U(2, 1) >= td1 || U(3, 0) >= ti0 || U(3, 1) >= ti1 || sum_rule(U(2, 0), U(3, 0), n0) || sum_rule(U(2, 1), U(3, 1), n1)) return 1;
    }
    return -1;
}

constexpr int EV_SUB = 1024, EV_MARGIN = 256, EV_CAP = 2048, EV_NT = NC_EV_NT;

// read index of every tile entry (its slot offset is unique) and its event cursors: for the tile's 1024-column blocks h = 0 .. SPT-1 (and the
// one after the tile) the first event of the read at or after (tile start + 1024 h - EV_BACK).  Once per call, one wave per tile, so that
// the blocks of k_event_tiles find an entry's events by a walk of a few steps instead of two bisections each (a third of that kernel).
// Row of an entry (NC_ENT_CUR_PITCH(SPT) words): [0 .. SPT] those cursors, [SPT + 1] / [SPT + 2] the read's event range, [SPT + 3 + h] the
// first event at or after the START of block h + 1 (exact: where block h's events end)
constexpr int EV_BACK = 64;
__global__ __launch_bounds__(64) void k_entry_cursors(const int32_t *__restrict__ tile_off, const nc_tile_entry *__restrict__ tile_ent, int32_t tile_pos0,
                                                      int32_t tile_size, const int64_t *__restrict__ slot_off, int32_t n_reads,
                                                      const int32_t *__restrict__ ev_off, const int32_t *__restrict__ ev_pos,
                                                      int32_t *__restrict__ ent_read, int32_t *__restrict__ ent_cur)
{
    const int t = blockIdx.x, SPT = tile_size / 1024;
    const int32_t t_lo = tile_pos0 + t * tile_size;
    for (int e = tile_off[t] + (int)threadIdx.x; e < tile_off[t + 1]; e += 64) {
        const nc_tile_entry ent = tile_ent[e];
        const int64_t so = (ent.base_flag & ~int64_t(15)) + (ent.start & ~15);
        int lo = 0, hi = n_reads;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (slot_off[mid] < so) lo = mid + 1; else hi = mid;
        }
        ent_read[e] = lo;
        const int eb = ev_off[lo + 1];
        int x = ev_off[lo];
        int32_t *row = ent_cur + (int64_t)e * NC_ENT_CUR_PITCH(SPT);
        for (int h = 0; h <= SPT; h++) {
            const int32_t want = t_lo + h * 1024 - EV_BACK;
            int y = eb;                                                // (the cursors ascend: each search starts at the one before)
            while (x < y) {
                const int mid = (x + y) >> 1;
                if (ev_pos[mid] < want) x = mid + 1; else y = mid;
            }
            row[h] = x;
            if (h > 0) {
                // the exact cursor of the block's start (= the end of block h - 1's events): a few events past the one EV_BACK columns before it
                const int32_t edge = t_lo + h * 1024;
                int z = x, zy = min(eb, x + 16);
                while (z < zy) {
                    const int mid = (z + zy) >> 1;
                    if (ev_pos[mid] < edge) z = mid + 1; else zy = mid;
                }
                if (z == x + 16 && z < eb && ev_pos[z] < edge) {
                    zy = eb;
                    while (z < zy) {
                        const int mid = (z + zy) >> 1;
                        if (ev_pos[mid] < edge) z = mid + 1; else zy = mid;
                    }
                }
                row[SPT + 3 + (h - 1)] = z;
            }
        }
        row[SPT + 1] = ev_off[lo];                                     // the read's event range rides along: k_event_tiles needs neither the read's index
        row[SPT + 2] = eb;                                             // nor ev_off (two levels of dependent loads less per workgroup)
    }
}

// The cursor table WITHOUT random probes (k_entry_cursors bisects every entry's read: ~35 probes into 276 MB of event positions, 2.2 GB fetched per
// chr20-sized contig, TA busy 0.82).  k_read_cursors streams the events once, one wave per read: event i is the first one at or after every
// block boundary B (and B - EV_BACK) that lies in (position of event i - 1, position of event i]; boundaries behind the last event get the
// read's event count.  A read's boundaries are those of the tiles it overlaps; its part of the table starts at rc_off(r) = slot offset / 1024
// + (2 SPT + 2) r (slots are at least as long as the reads: the parts do not overlap, no scan needed).  k_entry_rows then copies an entry's row.
__device__ __forceinline__ int64_t rc_off(int64_t slot_off, int r, int SPT) { return (slot_off >> 10) + (int64_t)(2 * SPT + 2) * r; }

__global__ __launch_bounds__(256) void k_read_cursors(int32_t n_reads, const int32_t *__restrict__ rd_start, const int32_t *__restrict__ rd_end,
                                                      const int64_t *__restrict__ slot_off, const int32_t *__restrict__ ev_off,
                                                      const int32_t *__restrict__ ev_pos, int32_t tile_pos0, int32_t tile_size,
                                                      int32_t *__restrict__ rc_lo, int32_t *__restrict__ rc_hi)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const int SPT = tile_size / 1024;
    const int32_t rs = rd_start[r], re = rd_end[r];
    const int ea = ev_off[r], eb = ev_off[r + 1];
    const int ta = (rs - tile_pos0) / tile_size, tb = (max(re - 1, rs) - tile_pos0) / tile_size;
    const int g0 = ta * SPT, cnt = (tb - ta + 1) * SPT + 1;         // boundaries g0 .. g0 + cnt - 1 at tile_pos0 + 1024 g
    int32_t *lo = rc_lo + rc_off(slot_off[r], r, SPT), *hi = rc_hi + rc_off(slot_off[r], r, SPT);
    constexpr int RU = 4;                                            // (a lane's loads of four rounds in flight together)
    for (int ib = ea + lane; ib <= eb; ib += 64 * RU) {              // (i == eb: the end of the list, behind every event)
        int32_t pp[RU], pc[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const int i = ib + 64 * u;
            pp[u] = i > ea && i <= eb ? ev_pos[i - 1] : 0;
            pc[u] = i < eb ? ev_pos[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const int i = ib + 64 * u;
            if (i > eb) break;
            const bool first = i == ea, last = i == eb;
            // boundaries B with pp < B - back <= pc  <=>  floor((pp + back - p0) / 1024) < k + g0 <= floor((pc + back - p0) / 1024)
#pragma unroll
            for (int w = 0; w < 2; w++) {
                const int back = w == 0 ? EV_BACK : 0;
                int k0 = first ? 0 : ((pp[u] + back - tile_pos0) >> 10) + 1 - g0;
                int k1 = last ? cnt - 1 : ((pc[u] + back - tile_pos0) >> 10) - g0;
                k0 = max(k0, 0);
                k1 = min(k1, cnt - 1);
                int32_t *dst = w == 0 ? lo : hi;
                for (int k = k0; k <= k1; k++) dst[k] = i;
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_entry_rows(const int32_t *__restrict__ tile_off, const nc_tile_entry *__restrict__ tile_ent, int32_t tile_pos0,
                                                   int32_t tile_size, const int64_t *__restrict__ slot_off, int32_t n_reads,
                                                   const int32_t *__restrict__ ev_off, const int32_t *__restrict__ rc_lo, const int32_t *__restrict__ rc_hi,
                                                   const int32_t *__restrict__ rd_start, const int32_t *__restrict__ rd_end,
                                                   int32_t *__restrict__ ent_read, int32_t *__restrict__ ent_cur)
{
    const int t = blockIdx.x, SPT = tile_size / 1024;
    for (int e = tile_off[t] + (int)threadIdx.x; e < tile_off[t + 1]; e += 64) {
        const nc_tile_entry ent = tile_ent[e];
        const int64_t so = (ent.base_flag & ~int64_t(15)) + (ent.start & ~15);
        int lo = 0, hi = n_reads;
        while (lo < hi) {                                              // (1.3 MB of slot offsets: the probes stay in L2)
            const int mid = (lo + hi) >> 1;
            if (slot_off[mid] < so) lo = mid + 1; else hi = mid;
        }
        const int r = lo;
        ent_read[e] = r;
        const int32_t rs = rd_start[r], re = rd_end[r];               // (the table's geometry is the read table's, as k_read_cursors took it)
        const int ta = (rs - tile_pos0) / tile_size, tb = (max(re - 1, rs) - tile_pos0) / tile_size;
        const int cnt = (tb - ta + 1) * SPT + 1;
        const int64_t off = rc_off(slot_off[r], r, SPT);
        int32_t *row = ent_cur + (int64_t)e * NC_ENT_CUR_PITCH(SPT);
        const int kb = (t - ta) * SPT;                                 // boundary index of the tile's first block (clamped: a tile before / behind the read)
        for (int h = 0; h <= SPT; h++) {
            const int k = min(max(kb + h, 0), cnt - 1);
            row[h] = rc_lo[off + k];
            if (h > 0) row[SPT + 3 + (h - 1)] = rc_hi[off + k];
        }
        row[SPT + 1] = ev_off[r];
        row[SPT + 2] = ev_off[r + 1];
    }
}

__global__ __launch_bounds__(EV_NT, 8) void k_event_tiles(const int32_t *__restrict__ tile_off, const nc_tile_entry *__restrict__ tile_ent, int32_t tile_pos0,
                                                     int32_t tile_size, const int32_t *__restrict__ ent_read, const int32_t *__restrict__ ent_cur,
                                                     const int32_t *__restrict__ ev_off, const int32_t *__restrict__ ev_pos,
                                                     const int32_t *__restrict__ ev_len, const uint8_t *__restrict__ read_hap,
                                                     const IndelChunk *__restrict__ ck, const int32_t *__restrict__ blk_chunk, char *__restrict__ ws, int32_t win,
                                                     int32_t small_win, int32_t haploid, int32_t mincov, double ins_t, double del_t,
                                                     int8_t *__restrict__ col_type_all, int32_t *__restrict__ err_bits, const uint16_t *__restrict__ dec_tab,
                                                     const int32_t *__restrict__ blk_base, int32_t dec_n)
{
    // interval ends per (class, haplotype) row and rank as 16-bit fields, two ranks per word, each biased by 0x4000: +1 is an atomic add and
    // -1 an atomic SUBTRACT of the field's unit, so neither carries into the neighbour field (16 KB instead of 32: a fourth workgroup per CU)
    __shared__ uint32_t difw[8][EV_SUB / 2];
    auto dif_add = [&](int row, int i) { atomicAdd(&difw[row][i >> 1], 1u << (16 * (i & 1))); };
    auto dif_sub = [&](int row, int i) { atomicSub(&difw[row][i >> 1], 1u << (16 * (i & 1))); };
    __shared__ int32_t rkw[EV_SUB + EV_MARGIN];
    __shared__ int32_t en_e0[256], en_pre[257], en_lim[256];
    __shared__ uint8_t en_h[256];
    __shared__ int32_t evk[EV_CAP];                                  // the batch's events: rank of the column (-1 excluded), classes they qualify for
    __shared__ uint8_t evq[EV_CAP];
    __shared__ uint8_t own[EV_CAP];                                  // entry (of the batch's 256) every event of the batch belongs to
    __shared__ int32_t sh_k0, sh_k1, sh_mlo, wsum[EV_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int SPT = tile_size / EV_SUB;
    const IndelChunk c = ck[blk_chunk[(int)blockIdx.x / SPT]];
    const int rel = (int)blockIdx.x - c.blk0 * SPT, t = c.tile0 + rel / SPT;
    const int32_t s_lo = tile_pos0 + t * tile_size + (rel % SPT) * EV_SUB;
    const int32_t b_lo = max(s_lo, c.lo), b_hi = min(s_lo + EV_SUB - 1, c.hi);
    if (b_lo > b_hi) return;
    const int16_t *rank = reinterpret_cast<const int16_t *>(ck_rank(ws, c));      // 16-bit rows, ranks local to their tile-block (k_hap_depth_b)
    const int32_t w_lo = max(c.lo, b_lo - EV_MARGIN);             // LDS window of the rank array
    if (tid == 0) { sh_k0 = INT32_MAX; sh_k1 = -1; sh_mlo = c.lo; }
    for (int i = tid; i < 8 * EV_SUB / 2; i += EV_NT) (&difw[0][0])[i] = 0x40004000u;
    // The entries of the block, FAST form: events from the cursor EV_BACK columns before the block to the exact cursor of the next block's start
    // -- no load depends on the margin, so the entry + cursor-row loads run beside the rank window's.  Events outside [c.lo, b_hi] are dropped one
    // by one (rkf below); events before the margin but inside the chunk are harmless: their intervals end before the block's first rank and are
    // clipped to it (+1 and -1 cancel), and a chain they extend to the left covers the same ranks of the block.  The fast form holds when the margin
    // starts at or after m_fix (block-uniform; else: excluded / empty stretches) and the tiles' entries fit one batch; otherwise the general walk.
    const int32_t m_fix = max(c.lo, b_lo - EV_BACK);
    const int t_fix = max(0, (m_fix - tile_pos0) / tile_size);    // t or t - 1
    const int fe0 = tile_off[t_fix], fe_t = tile_off[t], fe1 = tile_off[t + 1];
    const bool fast_ok = fe1 - fe0 <= 256;
    nc_tile_entry f_ent = {0, 0, 0};
    int f_x0 = 0, f_x1 = 0;
    const bool f_on = fast_ok && tid < 256 && fe0 + tid < fe1;
    const int f_tt = fe0 + tid >= fe_t ? t : t_fix;
    if (f_on) {
        const int e = fe0 + tid;
        f_ent = tile_ent[e];
        const int32_t *row = ent_cur + (int64_t)e * NC_ENT_CUR_PITCH(SPT);
        const int hq = rel % SPT;
        f_x0 = f_tt == t ? row[hq] : row[SPT];
        f_x1 = f_tt == t ? row[SPT + 3 + hq] : row[SPT + 2];
    }
    __syncthreads();
    {
        int32_t kmin = INT32_MAX, kmax = -1;                         // first / last yielded rank of the block's own columns
        constexpr int RKU = (EV_SUB + EV_MARGIN + EV_NT - 1) / EV_NT;     // (a thread's ranks in one round trip)
        int32_t rr[RKU];
#pragma unroll
        for (int u = 0; u < RKU; u++) {
            const int i = tid + u * EV_NT;
            rr[u] = i <= b_hi - w_lo ? (int32_t)rank[w_lo + i - c.lo] : -1;
        }
        // the ranks in HBM are local to their tile-block (k_hap_depth_b): the window lies in at most two blocks of the chunk
        const int wb0 = c.blk0 + (w_lo - tile_pos0) / tile_size - c.tile0;
        const int32_t edge = tile_pos0 + ((w_lo - tile_pos0) / tile_size + 1) * tile_size;      // first column of the second block
        const int32_t base0 = blk_base[wb0], base1 = edge <= b_hi ? blk_base[wb0 + 1] : 0;
#pragma unroll
        for (int u = 0; u < RKU; u++) {
            const int i = tid + u * EV_NT;
            if (rr[u] >= 0) rr[u] += w_lo + i >= edge ? base1 : base0;
        }
#pragma unroll
        for (int u = 0; u < RKU; u++) {
            const int i = tid + u * EV_NT;
            if (i > b_hi - w_lo) break;
            const int32_t r = rr[u];
            rkw[i] = r;
            if (r >= 0 && w_lo + i >= b_lo) { kmin = min(kmin, r); kmax = max(kmax, r); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmin = min(kmin, __shfl_xor(kmin, o, 64));
            kmax = max(kmax, __shfl_xor(kmax, o, 64));
        }
        if (lane == 0) { atomicMin(&sh_k0, kmin); atomicMax(&sh_k1, kmax); }
    }
    int f_cnt = 0;                                                   // (everything but `fast` itself is known here: one register lives on)
    {
        const int32_t ft_lo = tile_pos0 + f_tt * tile_size;
        const bool mine = f_tt == t || f_ent.end <= ft_lo + tile_size;
        const int hp = (int)((f_ent.base_flag >> 1) & 3);
        if (f_on && mine && f_ent.start <= b_hi && f_ent.end > m_fix && (haploid || hp == 1 || hp == 2)) {
            f_cnt = max(f_x1 - f_x0, 0);
            en_e0[tid] = f_x0;
            en_lim[tid] = f_x1;
            en_h[tid] = (uint8_t)(haploid ? 0 : hp - 1);
        }
    }
    __syncthreads();
#ifdef NC_ABL_EVT_A
    return;
#endif
    const int32_t k0 = sh_k0, nk = sh_k1 - k0 + 1;
    int8_t *col_type = col_type_all + c.coloff;
    if (sh_k1 < 0) {                                                 // no yielded column here
        for (int i = b_lo + tid; i <= b_hi; i += EV_NT) col_type[i - c.lo] = -1;
        return;
    }
    auto rk = [&](int32_t p) {                                       // (written so that the common case is a plain LDS read, not a flat load)
        int32_t k = rkw[max(p - w_lo, 0)];
        if (p < w_lo) {
            k = (int32_t)rank[p - c.lo];
            if (k >= 0) k += blk_base[c.blk0 + (p - tile_pos0) / tile_size - c.tile0];
        }
        return k;
    };
    const int wmax = max(win, small_win);
    if (wv == 0) {
        // margin: the columns before b_lo holding the ranks k0 - (wmax - 1) .. k0 - 1 -> its first column (walk back, 64 columns a step)
        const int32_t need = k0 - (wmax - 1);
        int32_t mlo = c.lo;
        for (int32_t base = b_lo - 1; base >= c.lo; base -= 64) {
            const int32_t pos = base - lane;
            const int32_t r = pos >= c.lo ? rk(pos) : -1;
            const uint64_t m = __ballot(r >= 0 && r < need);       // columns already outside the margin: the nearest one ends it
            if (m) { mlo = base - (int)__builtin_ctzll(m) + 1; break; }
        }
        if (lane == 0) sh_mlo = mlo;
    }
    __syncthreads();
    const int32_t m_lo = sh_mlo;
#ifdef NC_ABL_EVT_A2
    return;
#endif
    auto qualifies = [](int32_t sl, int cls) {
        const int32_t ln = sl < 0 ? -sl : sl;
        const bool ins = sl > 0;
        return cls < 2 ? (ln > 2 && ln <= 50 && ins == (cls == 1)) : (ln <= 10 && ins == (cls == 3));
    };
    auto rkf = [&](int32_t p) { return (p < c.lo || p > b_hi) ? -1 : rk(p); };      // (an event off the block's columns counts as on an excluded one)
    const bool fast = fast_ok && m_lo >= m_fix;
    if (fe1 - fe0 > 16000 && tid == 0) atomicOr(err_bits, 8);    // more reads than a 16-bit field counts interval ends for: the caller takes the other route
    const int t_first = fast ? t : max(0, (m_lo - tile_pos0) / tile_size);
    for (int tt = t_first; tt <= t; tt++) {
        const int32_t tt_lo = tile_pos0 + tt * tile_size;
        const int e0 = fast ? fe0 : tile_off[tt], e1 = fast ? fe0 + 1 : tile_off[tt + 1];          // (fast: one batch)
        if (e1 - e0 > 16000 && tid == 0) atomicOr(err_bits, 8);
        for (int eb0 = e0; eb0 < e1; eb0 += 256) {
            // ---- one entry per thread: its read, the read's events in [m_lo, b_hi]
            int cnt = 0;
            const int e = eb0 + tid;
            if (fast) cnt = f_cnt;
            else
            if (tid < 256 && e < e1) {
                // two levels of loads: the entry and its row of the cursor table (cursors of this block and the next, the read's event range); then the
                // events either side of both cursors, all at once.  The haplotype tag sits in the entry.  (Round 4 walked: entry -> read -> tag, event
                // range -> cursors -> one event per step: eight dependent loads per workgroup, a third of the kernel.)
                const nc_tile_entry ent = tile_ent[e];
                const int32_t *cur = ent_cur + (int64_t)e * NC_ENT_CUR_PITCH(SPT);
                const int hq = tt == t ? rel % SPT : SPT;
                int x0 = cur[hq], x1 = tt == t ? cur[hq + 1] : cur[SPT + 2];
                const int ea = cur[SPT + 1], eb = cur[SPT + 2];
                // a read is taken at the LAST of these tiles that lists it: tile t's entry knows where the block's events begin and end
                // (k_entry_cursors), an entry of an earlier tile is of a read that ends before tile t
                const bool mine = tt == t || ent.end <= tt_lo + tile_size;
                const int hp = (int)((ent.base_flag >> 1) & 3);
                if (mine && ent.start <= b_hi && ent.end > m_lo && (haploid || hp == 1 || hp == 2)) {
                    if (tt < t - 1) {                                         // (a margin longer than a tile: excluded stretch)
                        x0 = ea;
                        int y0 = eb;
                        while (x0 < y0) {
                            const int m0 = (x0 + y0) >> 1;
                            if (ev_pos[m0] < m_lo) x0 = m0 + 1; else y0 = m0;
                        }
                    }
                    // the cursors stand EV_BACK columns before their block
                    const int32_t p0m = x0 > ea ? ev_pos[x0 - 1] : INT32_MIN, p1m = x1 > ea ? ev_pos[x1 - 1] : INT32_MIN;
                    int32_t q0[4], q1[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { q0[u] = x0 + u < eb ? ev_pos[x0 + u] : INT32_MAX; q1[u] = x1 + u < eb ? ev_pos[x1 + u] : INT32_MAX; }
                    int adv0 = 0, adv1 = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) { adv0 += q0[u] < m_lo ? 1 : 0; adv1 += q1[u] <= b_hi ? 1 : 0; }       // (ascending: prefixes)
                    const int nx0 = x0 + adv0;
                    const bool slow = p0m >= m_lo || adv0 == 4 || adv1 == 4 || x1 < nx0 || (x1 > nx0 && p1m > b_hi);
                    if (!slow) { x0 = nx0; x1 += adv1; }
                    else {                                                   // the general walk
                        while (x0 > ea && ev_pos[x0 - 1] >= m_lo) x0--;
                        while (x0 < eb && ev_pos[x0] < m_lo) x0++;
                        x1 = max(x1, x0);
                        while (x1 > x0 && ev_pos[x1 - 1] > b_hi) x1--;
                        while (x1 < eb && ev_pos[x1] <= b_hi) x1++;
                    }
                    cnt = x1 - x0;
                    en_e0[tid] = x0;
                    en_lim[tid] = x1;
                    en_h[tid] = (uint8_t)(haploid ? 0 : hp - 1);
                }
            }
            // ---- exclusive prefix of the counts over the 256 entries
            int inc = cnt;
            inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
            if (lane == 63 && wv < 4) wsum[wv] = inc;
            __syncthreads();
            int wp = 0;
            for (int q = 0; q < wv && q < 4; q++) wp += wsum[q];
            if (tid < 256) en_pre[tid] = wp + inc - cnt;
            const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tid == 0) en_pre[256] = total;
            // (an event finds its entry by one LDS read: two bisections of eight dependent reads each per event were a third of this kernel)
            if (tid < 256 && total <= EV_CAP) for (int u = 0; u < cnt; u++) own[wp + inc - cnt + u] = (uint8_t)tid;
            __syncthreads();
            // ---- one event per thread
#ifdef NC_ABL_EVT_B
            if (total >= 0) { __syncthreads(); continue; }
#endif
            if (total <= EV_CAP) {
                // the batch's events into LDS (independent loads), then every look-up at a neighbour is an LDS read
                constexpr int EVU = 4;                                            // (a thread's events in one round trip to HBM)
                for (int base = 0; base < total; base += EVU * EV_NT) {
                    int32_t e_pos[EVU], e_len[EVU];
#pragma unroll
                    for (int u = 0; u < EVU; u++) {
                        const int idx = base + tid + u * EV_NT;
                        if (idx < total) {
                            const int lo = own[idx];
                            const int ev = en_e0[lo] + (idx - en_pre[lo]);
                            e_pos[u] = ev_pos[ev];
                            e_len[u] = ev_len[ev];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EVU; u++) {
                        const int idx = base + tid + u * EV_NT;
                        if (idx < total) {
                            const int32_t sl = e_len[u];
                            evk[idx] = rkf(e_pos[u]);
                            evq[idx] = (uint8_t)((qualifies(sl, 0) ? 1 : 0) | (qualifies(sl, 1) ? 2 : 0) | (qualifies(sl, 2) ? 4 : 0) | (qualifies(sl, 3) ? 8 : 0));
                        }
                    }
                }
                __syncthreads();
#ifdef NC_ABL_EVT_C
                if (total >= 0) { __syncthreads(); continue; }
#endif
                for (int idx = tid; idx < total; idx += EV_NT) {
                    const int k = evk[idx], qm = evq[idx];
                    if (k < 0 || qm == 0) continue;
                    const int lo = own[idx];
                    const int i0 = en_pre[lo], i1 = en_pre[lo + 1], h = en_h[lo];
                    // one walk back and one forward for all the classes the event qualifies for (a class leaves the search when the distance passes
                    // its window: the ranks only grow apart) instead of two loops per class: the per-class form was 0.9 of this kernel's 2.2 ms
                    uint32_t prevf = 0, nextf = 0, need = (uint32_t)qm;
                    for (int i2 = idx - 1; i2 >= i0 && need; i2--) {
                        const int k2 = evk[i2];
                        if (k2 < 0) continue;
                        const int d = k - k2;
                        if (d > win - 1) need &= ~3u;
                        if (d > small_win - 1) need &= ~12u;
                        const uint32_t f = (uint32_t)evq[i2] & need;
                        prevf |= f;
                        need &= ~f;
                    }
                    need = (uint32_t)qm;
                    for (int i2 = idx + 1; i2 < i1 && need; i2++) {
                        const int k2 = evk[i2];
                        if (k2 < 0) continue;
                        const int d = k2 - k;
                        if (d > win - 1) need &= ~3u;
                        if (d > small_win - 1) need &= ~12u;
                        const uint32_t f = (uint32_t)evq[i2] & need;
                        nextf |= f;
                        need &= ~f;
                    }
#pragma unroll
                    for (int cls = 0; cls < 4; cls++) {
                        if (!((qm >> cls) & 1)) continue;
                        const int w = cls < 2 ? win : small_win;
                        if (!((prevf >> cls) & 1)) dif_add(cls * 2 + h, max(k, k0) - k0);
                        if (!((nextf >> cls) & 1) && max(k + w, k0) - k0 < nk) dif_sub(cls * 2 + h, max(k + w, k0) - k0);
                    }
                }
            } else
            for (int idx = tid; idx < total; idx += EV_NT) {                        // (a batch of more events than the LDS arrays hold: straight from HBM)
                int lo = 0, hi = 255;                                             // last entry whose prefix is <= idx
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (en_pre[mid] <= idx) lo = mid; else hi = mid - 1;
                }
                const int j = lo, ef = en_e0[j], el = en_lim[j], h = en_h[j];
                const int ev = ef + (idx - en_pre[j]);
                const int32_t p = ev_pos[ev], sl = ev_len[ev];
                const int k = rkf(p);
                if (k < 0) continue;                                              // excluded column
#pragma unroll
                for (int cls = 0; cls < 4; cls++) {
                    if (!qualifies(sl, cls)) continue;
                    const int w = cls < 2 ? win : small_win;
                    bool has_prev = false, has_next = false;
                    for (int e2 = ev - 1; e2 >= ef; e2--) {
                        const int k2 = rkf(ev_pos[e2]);
                        if (k2 < 0) continue;
                        if (k - k2 > w - 1) break;
                        if (qualifies(ev_len[e2], cls)) { has_prev = true; break; }
                    }
                    for (int e2 = ev + 1; e2 < el; e2++) {
                        const int k2 = rkf(ev_pos[e2]);
                        if (k2 < 0) continue;
                        if (k2 - k > w - 1) break;
                        if (qualifies(ev_len[e2], cls)) { has_next = true; break; }
                    }
                    // ends clipped to the block's first rank: a margin event whose own window stops short of the block may still open the
                    // chain a later margin event continues into it (+1 and -1 on rank k0 cancel when nothing does)
                    if (!has_prev) dif_add(cls * 2 + h, max(k, k0) - k0);
                    if (!has_next && max(k + w, k0) - k0 < nk) dif_sub(cls * 2 + h, max(k + w, k0) - k0);
                }
            }
            __syncthreads();
        }
    }
#ifdef NC_ABL_EVT_D
    return;
#endif
    const uint16_t *depth = reinterpret_cast<const uint16_t *>(ck_depth(ws, c));
    constexpr int DCOL = EV_SUB / EV_NT;
    int dn0[DCOL], dn1[DCOL];
#pragma unroll
    for (int u = 0; u < DCOL; u++) {                                 // (all of a thread's depth loads in one round trip, under the scan)
        const int i = b_lo + tid + u * EV_NT;
        dn0[u] = i <= b_hi ? (int)depth[i - c.lo] : 0;
        dn1[u] = i <= b_hi ? (int)depth[c.ncol + (i - c.lo)] : 0;
    }
    // the decision tables over the batch's event ranks (free now); the barrier after the scan publishes them
    uint16_t *dtab = reinterpret_cast<uint16_t *>(evk);
    static_assert(sizeof(int32_t) * EV_CAP >= 2 * DEC_N * sizeof(uint16_t), "k_event_tiles: the decision tables fit the event ranks' array");
    for (int i = tid; i < 2 * DEC_N / 2; i += EV_NT) reinterpret_cast<uint32_t *>(dtab)[i] = reinterpret_cast<const uint32_t *>(dec_tab)[i];
    // ---- inclusive scan of the eight rows: dif becomes U[class, haplotype][rank - k0]
    // a wave per row: each lane sums 16 consecutive ranks, one scan over the 64 lane totals, then the lane's 16 window counts (in place)
    for (int row = wv; row < 8; row += EV_NT / 64) {
        int v[EV_SUB / 64], tot = 0;
#pragma unroll
        for (int q = 0; q < EV_SUB / 64; q++) {
            const int i = lane * (EV_SUB / 64) + q;
            v[q] = i < nk ? (int)((difw[row][i >> 1] >> (16 * (i & 1))) & 0xffffu) - 0x4000 : 0;
            tot += v[q];
        }
        int inc = tot;
        inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
        int run = inc - tot;
        uint16_t *U16 = reinterpret_cast<uint16_t *>(&difw[row][0]);        // the window counts (0 .. reads of the block) over the fields they came from:
#pragma unroll                                                        // a lane rewrites the 16 ranks (8 words) it has just read
        for (int q = 0; q < EV_SUB / 64; q++) {
            run += v[q];
            U16[lane * (EV_SUB / 64) + q] = (uint16_t)run;
        }
    }
    __syncthreads();
#ifdef NC_ABL_EVT_E
    return;
#endif
    // the columns' decisions (k_indel_decide_b's, without the window counts' trip through HBM)
#pragma unroll
    for (int u = 0; u < DCOL; u++) {
        const int i = b_lo + tid + u * EV_NT;
        if (i > b_hi) break;
        const int k = rkw[i - w_lo];
        const int n0 = dn0[u], n1 = dn1[u];
        col_type[i - c.lo] = indel_decide_tab(k, n0, n1, [&](int cls, int h) { return (int)reinterpret_cast<const uint16_t *>(&difw[cls * 2 + h][0])[k - k0]; },
                                              mincov, ins_t, del_t, haploid, dtab, dtab + DEC_N, dec_n);
    }
}

// in-place inclusive prefix sum of each of the 8 difference arrays of each chunk (one workgroup per array)
__global__ __launch_bounds__(1024) void k_prefix_rows_b(const IndelChunk *__restrict__ ck, char *__restrict__ ws)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const IndelChunk c = ck[blockIdx.y];
    const int32_t nd = c.nd;
    int32_t *row = ck_diff(ws, c) + (int64_t)blockIdx.x * nd;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nd; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nd ? row[i] : 0;
        int tot;
        const int inc = block_scan_1024(v, wsum, tot);
        const int cc = carry;
        if (i < nd) row[i] = cc + inc;
        __syncthreads();
        if (threadIdx.x == 0) carry = cc + tot;
        __syncthreads();
    }
}

// per-column decision of :252-275 (float64 divide-and-compare, as in the reference); with impute_indel_phase also the
// column-level part of :278-284 (type 2: the read grouping of :285-304 is left to the host for these few columns)
__global__ void k_indel_decide_b(const IndelChunk *__restrict__ ck, char *__restrict__ ws, int32_t mincov, double ins_t, double del_t,
                                 int32_t haploid, int32_t impute, int8_t *__restrict__ col_type_all)
{
    const IndelChunk c = ck[blockIdx.y];
    const int32_t ncol = c.ncol, nd = c.nd;
    const int32_t *depth = ck_depth(ws, c), *rank = ck_rank(ws, c), *U = ck_diff(ws, c);
    int8_t *col_type = col_type_all + c.coloff;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncol; i += gridDim.x * blockDim.x) {
        const int k = rank[i];
        const int n0 = depth[i], n1 = depth[ncol + i];
        int8_t type = indel_decide(k, n0, n1, [&](int cls, int h) { return U[(int64_t)(cls * 2 + h) * nd + k]; }, mincov, ins_t, del_t, haploid);
        const bool ruled = haploid || (k >= 0 && n0 >= mincov && n1 >= mincov);                         // :252-275 applied; otherwise (impute_indel_phase) :278-284
        if (!ruled && impute && k >= 0) {
            const int tot = n0 + n1 + depth[2 * (int64_t)ncol + i];
            if (tot >= 2 * mincov && tot > 0) {                                                          // :278
                const int32_t *cnt = ck_cnt(ws, c);
                const double del_f = (double)(cnt[i] + cnt[2 * (int64_t)ncol + i]) / (double)tot;        // '-' and '*' (:282)
                const double ins_f = (double)cnt[(int64_t)ncol + i] / (double)tot;                       // '+' (:283)
                if (del_t <= del_f || ins_t <= ins_f) type = 2;                                          // :284
            }
        }
        col_type[i] = type;
    }
}

}   // namespace

int nc_indel_check(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const nc_indel_scan_params *prm, const char *who)
{
    if (!pack || !ev || !prm) return nc_fail(ctx, NC_ERR_ARG, "%s: bad argument", who);
    const int tile = pack->tile_size;
    if (!(tile == 1024 || tile == 2048 || tile == 4096)) return nc_fail(ctx, NC_ERR_ARG, "%s: malformed read pack", who);
    if (prm->win_size < 1 || prm->small_win_size < 1 || prm->win_size > 4096) return nc_fail(ctx, NC_ERR_ARG, "%s: window sizes", who);
    return NC_OK;
}

// the launches of one group (up to and including the per-column decisions, which stay on the device at *ctype_out; chunk
// descriptors in `ck`, on the device at *ck_dev_out); no synchronisation
int nc_indel_scan_group_launch(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const uint8_t *excl_dev, int32_t n_chunks,
                               const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *prm, int32_t *consumed,
                               std::vector<IndelChunk> &ck, const IndelChunk **ck_dev_out, const int8_t **ctype_out, const int64_t *slot_off_dev,
                               int32_t *err_bits_dev, const int32_t *rd_start_dev, const int32_t *rd_end_dev, bool reuse_tables)
{
    // reuse_tables: a later group of chunks of the SAME pack and events (the device pipeline's plan): the cursor table of the first group stands
    if (!reuse_tables) ctx->indel_ent_of = nullptr;
    const int tile = pack->tile_size;
    const int32_t grid_lo = pack->tile_pos0, grid_hi = pack->tile_pos0 + pack->n_tiles * tile - 1;
    const int impute = prm->impute && !prm->haploid;
    // workspace per group of chunks: a twelfth of the device memory that is free at the first call, between 6 and 24 GiB (a chr1-sized contig's
    // columns need 12 GB: one group, one launch of every K7 kernel -- and ONE pass of k_entry_cursors over the tile index -- instead of two)
    // (per context, after hipSetDevice: a process that drives several GPUs sizes each one's workspace from that GPU; what this context's workspace
    // already holds counts as free, and a budget the device can no longer serve -- other buffers grew since -- shrinks to what is free now)
    NC_HIP(ctx, hipSetDevice(ctx->device));
    {
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess) mfree = (size_t)72 << 30;
        const size_t avail = mfree + ctx->indel_ws.cap;
        const size_t want = std::min<size_t>((size_t)24 << 30, std::max<size_t>((size_t)6 << 30, avail / 12));
        if (ctx->k7_budget == 0 || (ctx->k7_budget_shrunk && want <= avail / 2)) {      // first plan, or memory has been freed since a low moment
            ctx->k7_budget = want;
            ctx->k7_budget_shrunk = false;
        }
        if (ctx->k7_budget > avail / 2) {
            ctx->k7_budget = std::max<size_t>((size_t)256 << 20, avail / 2);
            ctx->k7_budget_shrunk = true;
        }
    }
    const size_t BUDGET = ctx->k7_budget;
    ck.clear();
    size_t wsb = 0;
    int64_t ncols = 0;
    int32_t nblk = 0, c1 = 0;
    bool clipped = false;                                            // a chunk reaches past the pack's tile grid (columns no kernel of the tile index visits)
    for (; c1 < n_chunks; c1++) {
        IndelChunk k;
        k.lo = starts[c1] < 1 ? 1 : starts[c1];
        k.hi = ends[c1];
        k.ncol = k.hi - k.lo + 1;
        k.nd = k.ncol + prm->win_size + 2;
        const size_t need = ((size_t)3 * k.ncol * 4 + ((size_t)k.ncol + 1) * 4 + (size_t)8 * k.nd * 4 + (impute ? (size_t)3 * k.ncol * 4 : 0) + 15) & ~(size_t)15;
        if (!ck.empty() && (wsb + need + (size_t)ncols + k.ncol > BUDGET || ck.size() >= 32768)) break;   // gridDim.y < 65536
        k.ws = (int64_t)wsb;
        k.coloff = ncols;
        const int32_t clo = k.lo > grid_lo ? k.lo : grid_lo, chi = k.hi < grid_hi ? k.hi : grid_hi;
        k.tile0 = chi >= clo ? (clo - grid_lo) / tile : 0;
        k.blk0 = nblk;
        nblk += chi >= clo ? (chi - grid_lo) / tile - k.tile0 + 1 : 0;
        if (k.lo < grid_lo || k.hi > grid_hi) clipped = true;
        wsb += need;
        ncols += k.ncol;
        ck.push_back(k);
    }
    *consumed = c1;
    const int32_t ng = (int32_t)ck.size();
    const size_t o_type = wsb, o_ck = (o_type + (size_t)ncols + 15) & ~(size_t)15, o_blk = o_ck + (((size_t)ng * sizeof(IndelChunk) + 15) & ~(size_t)15),
                 total = o_blk + (size_t)nblk * 4 * 3;                // blk_chunk, blk_yield, blk_base
    NC_TRY(nc_ensure(ctx, ctx->indel_ws, total));
    char *ws = (char *)ctx->indel_ws.p;
    const bool no_tiles = getenv("NC_K7_EVENT_ATOMICS") != nullptr;                         // k_event_intervals_w + k_prefix_rows_b + k_indel_decide_b, for A/B checks (read per call: tests flip it)
    const bool tiles = slot_off_dev && err_bits_dev && !impute && !no_tiles && !clipped && ev->n_reads > 0 && nblk > 0 && tile % EV_SUB == 0;
    // the tiled form writes every word it reads (depths, ranks, window counts in LDS, decisions); the other one accumulates into zeros
    if (!tiles) NC_HIP(ctx, hipMemsetAsync(ws, 0, o_type, ctx->stream));
    IndelChunk *ck_dev = (IndelChunk *)(ws + o_ck);
    NC_TRY(nc_h2d_pieces(ctx, ck_dev, ck.data(), (size_t)ng * sizeof(IndelChunk), ctx->stream));   // by copy kernel: never behind an upload in flight
    int8_t *ctype = (int8_t *)(ws + o_type);
    int32_t *blk_chunk = (int32_t *)(ws + o_blk);
    int32_t *blk_yield = tiles ? blk_chunk + nblk : nullptr, *blk_base = blk_chunk + 2 * (size_t)nblk;
    if (nblk > 0) hipLaunchKernelGGL(k_blk_chunks, dim3((ng + 255) / 256), dim3(256), 0, ctx->stream, ck_dev, ng, nblk, blk_chunk);
#define NC_HAP_DEPTH(B, STAR)                                                                                                        \
    hipLaunchKernelGGL((k_hap_depth_b<B, STAR>), dim3(nblk), dim3(B), 0, ctx->stream, pack->codes, pack->tile_off, pack->tile_ent, \
                       pack->tile_pos0, ck_dev, blk_chunk, ws, prm->haploid, excl_dev, grid_lo, (STAR) ? nullptr : blk_yield)
    if (nblk > 0) {
        if (tile == 1024) NC_HAP_DEPTH(64, false);
        else if (tile == 2048) NC_HAP_DEPTH(128, false);
        else NC_HAP_DEPTH(256, false);
        if (impute) {
            if (tile == 1024) NC_HAP_DEPTH(64, true);
            else if (tile == 2048) NC_HAP_DEPTH(128, true);
            else NC_HAP_DEPTH(256, true);
        }
    }
#undef NC_HAP_DEPTH
    if (tiles) hipLaunchKernelGGL(k_blk_base, dim3((ng + 3) / 4), dim3(256), 0, ctx->stream, ck_dev, ng, nblk, blk_yield, blk_base, ws);
    else hipLaunchKernelGGL(k_yield_rank_b, dim3(ng), dim3(1024), 0, ctx->stream, ck_dev, ws, excl_dev, grid_lo);
    if (tiles) {
        const int SPT = tile / EV_SUB;
        const size_t tab_words = (size_t)pack->n_entries * (size_t)(NC_ENT_CUR_PITCH(SPT) + 1) + 2 * DEC_N / 2;
        const size_t rc_words = (size_t)(pack->codes_len >> 10) + (size_t)(2 * SPT + 2) * (size_t)ev->n_reads + 8;      // per table (rc_off)
        const bool stream_tab = rd_start_dev && rd_end_dev && !getenv("NC_K7_CURSOR_PROBES");                       // (env: k_entry_cursors, for A/B checks)
        const bool have = reuse_tables && ctx->indel_ent_of == pack->tile_ent && ctx->indel_ent_read.p;
        if (!have) NC_TRY(nc_ensure(ctx, ctx->indel_ent_read, 4 * (tab_words + (stream_tab ? 2 * rc_words : 0))));
        int32_t *ent_read = (int32_t *)ctx->indel_ent_read.p, *ent_cur = ent_read + pack->n_entries;
        uint16_t *dec_tab = reinterpret_cast<uint16_t *>(ent_cur + (size_t)pack->n_entries * NC_ENT_CUR_PITCH(SPT));
        if (!have) {
            if (stream_tab) {
                int32_t *rc_lo = ent_read + tab_words, *rc_hi = rc_lo + rc_words;
                hipLaunchKernelGGL(k_read_cursors, dim3((unsigned)((ev->n_reads + 3) / 4)), dim3(256), 0, ctx->stream, ev->n_reads, rd_start_dev, rd_end_dev,
                                   slot_off_dev, ev->ev_off, ev->ev_pos, pack->tile_pos0, tile, rc_lo, rc_hi);
                hipLaunchKernelGGL(k_entry_rows, dim3((unsigned)pack->n_tiles), dim3(64), 0, ctx->stream, pack->tile_off, pack->tile_ent, pack->tile_pos0, tile,
                                   slot_off_dev, ev->n_reads, ev->ev_off, rc_lo, rc_hi, rd_start_dev, rd_end_dev, ent_read, ent_cur);
            } else
                hipLaunchKernelGGL(k_entry_cursors, dim3((unsigned)pack->n_tiles), dim3(64), 0, ctx->stream, pack->tile_off, pack->tile_ent, pack->tile_pos0, tile,
                                   slot_off_dev, ev->n_reads, ev->ev_off, ev->ev_pos, ent_read, ent_cur);
            hipLaunchKernelGGL(k_decide_tables, dim3(2 * DEC_N / 256), dim3(256), 0, ctx->stream, prm->del_t, prm->ins_t, dec_tab);
        }
        // depths the tables serve (NC_K7_DEC_N < 1024: tests send ordinary depths down the division form; read per call)
        const char *dn = getenv("NC_K7_DEC_N");
        const int32_t dec_n = dn ? std::max(0, std::min(DEC_N, atoi(dn))) : DEC_N;
        ctx->indel_ent_of = pack->tile_ent;                              // (the device pipeline's k_sets / k_windows use the tables too)
        ctx->indel_ent_spt = SPT;
        hipLaunchKernelGGL(k_event_tiles, dim3(nblk * (tile / EV_SUB)), dim3(EV_NT), 0, ctx->stream, pack->tile_off, pack->tile_ent, pack->tile_pos0, tile,
                           ent_read, ent_cur, ev->ev_off, ev->ev_pos, ev->ev_len, ev->read_hap, ck_dev, blk_chunk, ws, prm->win_size,
                           prm->small_win_size, prm->haploid, prm->mincov, prm->ins_t, prm->del_t, ctype, err_bits_dev, dec_tab, blk_base, dec_n);
    } else if (ev->n_reads > 0) {
        static const bool per_thread = getenv("NC_K7_THREAD_PER_READ") != nullptr;          // the round-1 form, kept for A/B checks
        if (per_thread)
            hipLaunchKernelGGL(k_event_intervals_b, dim3((ev->n_reads + 255) / 256), dim3(256), 0, ctx->stream, ev->n_reads, ev->ev_off, ev->ev_pos,
                               ev->ev_len, ev->read_hap, ck_dev, ng, ws, prm->win_size, prm->small_win_size, prm->haploid, impute);
        else
            hipLaunchKernelGGL(k_event_intervals_w, dim3((ev->n_reads + 3) / 4), dim3(256), 0, ctx->stream, ev->n_reads, ev->ev_off, ev->ev_pos,
                               ev->ev_len, ev->read_hap, ck_dev, ng, ws, prm->win_size, prm->small_win_size, prm->haploid, impute);
    }
    if (!tiles) hipLaunchKernelGGL(k_prefix_rows_b, dim3(8, ng), dim3(1024), 0, ctx->stream, ck_dev, ws);
    if (!tiles)
        hipLaunchKernelGGL(k_indel_decide_b, dim3(ng == 1 ? 512 : 64, ng), dim3(256), 0, ctx->stream, ck_dev, ws, prm->mincov, prm->ins_t, prm->del_t,
                           prm->haploid, impute, ctype);
    NC_HIP(ctx, hipGetLastError());
    *ck_dev_out = ck_dev;
    *ctype_out = ctype;
    return NC_OK;
}

// one group of ascending chunks: a single set of launches, one device-to-host copy per run of back-to-back outputs
static int indel_scan_group(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const uint8_t *excl_dev, int32_t n_chunks,
                            const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *prm, int8_t *col_type_host,
                            const int64_t *col_off, int32_t *consumed)
{
    std::vector<IndelChunk> ck;
    const IndelChunk *ck_dev = nullptr;
    const int8_t *ctype = nullptr;
    NC_TRY(nc_indel_scan_group_launch(ctx, pack, ev, excl_dev, n_chunks, starts, ends, prm, consumed, ck, &ck_dev, &ctype, nullptr, nullptr, nullptr, nullptr, false));
    const int32_t ng = (int32_t)ck.size();
    for (int32_t k = 0; k < ng;) {                                   // runs of chunks laid out back to back on the host as well
        int32_t j = k + 1;
        while (j < ng && col_off[j] - col_off[k] == ck[(size_t)j].coloff - ck[(size_t)k].coloff) j++;
        const int64_t nbytes = ck[(size_t)j - 1].coloff + ck[(size_t)j - 1].ncol - ck[(size_t)k].coloff;
        NC_HIP(ctx, hipMemcpyAsync(col_type_host + col_off[k], ctype + ck[(size_t)k].coloff, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
        k = j;
    }
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NC_OK;
}

// Many chunks of one contig per call: every chunk keeps the reference's per-chunk semantics (fresh window deques at the
// chunk start).  Runs of chunks ascending in start and end go into the same launches, the chunk being a grid dimension.
extern "C" int nc_indel_scan_batch(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const uint8_t *excl_dev,
                                   int32_t n_chunks, const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *prm,
                                   int8_t *col_type_host, const int64_t *col_off)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_chunks < 0 || (n_chunks && (!starts || !ends || !col_type_host || !col_off))) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_scan_batch: bad argument");
    NC_TRY(nc_indel_check(ctx, pack, ev, prm, "nc_indel_scan_batch"));
    if (n_chunks == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    for (int32_t c = 0; c < n_chunks; c++)
        if (ends[c] < starts[c]) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_scan_batch: chunk %d has end < start", c);
    NcTimer tm(ctx, 3);
    int32_t c0 = 0;
    while (c0 < n_chunks) {
        int32_t c1 = c0 + 1;                                         // maximal ascending run
        while (c1 < n_chunks && starts[c1] >= starts[c1 - 1] && ends[c1] >= ends[c1 - 1]) c1++;
        while (c0 < c1) {
            int32_t used = 0;
            NC_TRY(indel_scan_group(ctx, pack, ev, excl_dev, c1 - c0, starts + c0, ends + c0, prm, col_type_host, col_off + c0, &used));
            c0 += used;
        }
    }
    tm.stop();
    return NC_OK;
}

extern "C" int nc_indel_scan(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *ev, const uint8_t *excl_dev, int32_t start,
                             int32_t end, const nc_indel_scan_params *prm, int8_t *col_type_host)
{
    if (!ctx) return NC_ERR_ARG;
    if (!col_type_host || end < start) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_scan: bad argument");
    const int64_t off = 0;
    return nc_indel_scan_batch(ctx, pack, ev, excl_dev, 1, &start, &end, prm, col_type_host, &off);
}
