// K1: candidate / neighbour-site scan over packed alignments (gfx950).
//
// Restates the column loop of get_snp_testing_candidates (reference generate_SNP_pileups.py:156-186).
// HBM-bound design: one workgroup per tile of TILE = 16*BLOCK reference positions; every lane owns 16
// consecutive positions and, for each read overlapping the tile, loads the read's 16 codes as ONE aligned
// dwordx4 (position-aligned slots, see nanocaller_hip.h).  Codes are counted four positions at a time with
// v_perm_b32 used as an 8-entry byte LUT (code -> 0/1 per byte lane) into byte-lane accumulators that are
// widened every 255 reads.  Algorithmic traffic: one byte per pileup entry + one reference byte per column.
#include <algorithm>

#include "nc_common.h"

namespace {

// byte LUT: result byte i = table[x.byte i], table = {hi:lo} little endian, valid for byte values 0..7
__device__ __forceinline__ uint32_t lut8(uint32_t x, uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, x); }

constexpr uint32_t LUT_LO[5] = {0x00000001u, 0x00000100u, 0x00010000u, 0x01000000u, 0x01010101u};
constexpr uint32_t LUT_HI[5] = {0u, 0u, 0u, 0u, 0x00000001u};   // plane 4 = "present" (codes 0..4)

struct ScanP {
    int32_t mincov;
    int32_t haploid;
    double min_af, t0, t1;
    int32_t scan_lo, scan_hi;
};

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_scan(const uint8_t *__restrict__ codes, const int32_t *__restrict__ tile_off,
                                                const nc_tile_entry *__restrict__ tile_ent, int32_t tile_pos0,
                                                const uint8_t *__restrict__ ref_code, ScanP sp,
                                                int32_t *__restrict__ stage_nbr, int32_t *__restrict__ stage_cpos,
                                                int32_t *__restrict__ stage_cn, int32_t *__restrict__ stage_calt,
                                                int2 *__restrict__ tile_cnt)
{
    constexpr int TILE = BLOCK * 16;
    const int t = blockIdx.x;
    const int64_t tile_base = (int64_t)tile_pos0 + (int64_t)t * TILE;
    const int32_t P0 = (int32_t)(tile_base + threadIdx.x * 16);

    uint32_t acc[5][4];
    uint32_t wide[5][8];
#pragma unroll
    for (int c = 0; c < 5; c++) {
#pragma unroll
        for (int d = 0; d < 4; d++) acc[c][d] = 0;
#pragma unroll
        for (int d = 0; d < 8; d++) wide[c][d] = 0;
    }
    const int e0 = tile_off[t], e1 = tile_off[t + 1];
    int e = e0;
    while (e < e1) {
        const int lim = min(e1, e + 255);
#pragma unroll 8
        for (; e < lim; e++) {
            const nc_tile_entry ent = tile_ent[e];            // wave-uniform -> scalar loads
            const int32_t lo = ent.start & ~15, hi = (ent.end + 15) & ~15;
            uint4 v = make_uint4(0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u);
            if (P0 >= lo && P0 < hi) v = *reinterpret_cast<const uint4 *>(codes + (ent.base_flag & ~int64_t(15)) + P0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
#pragma unroll
                for (int c = 0; c < 5; c++) acc[c][d] += lut8(w[d], LUT_LO[c], LUT_HI[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 5; c++) {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                wide[c][2 * d] += acc[c][d] & 0x00FF00FFu;            // positions 4d+0 (lo16), 4d+2 (hi16)
                wide[c][2 * d + 1] += (acc[c][d] >> 8) & 0x00FF00FFu; // positions 4d+1, 4d+3
                acc[c][d] = 0;
            }
        }
    }

    // per-position decision (generate_SNP_pileups.py:161-186)
    const uint4 rv = *reinterpret_cast<const uint4 *>(ref_code + (int64_t)t * TILE + threadIdx.x * 16);
    const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
    uint32_t nbr_mask = 0, cand_mask = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int d = i >> 2, k = i & 3;
        const int wi = 2 * d + (k & 1), sh = (k >> 1) * 16;
        const int r = (rw[d] >> (8 * k)) & 0xFF;
        const int32_t p = P0 + i;
        const uint32_t n = (wide[4][wi] >> sh) & 0xFFFF;
        if (r < 4 && n > 0 && p >= sp.scan_lo && p <= sp.scan_hi && (int32_t)n >= sp.mincov) {
            uint32_t alt = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t cb = (wide[b][wi] >> sh) & 0xFFFF;
                if (b != r && cb > alt) alt = cb;
            }
            const double f = (double)alt / (double)n;                 // alt_freq, float64 (:166)
            const bool nb = sp.haploid ? (sp.t0 <= f) : (sp.t0 <= f && f < sp.t1);   // :173 / :177
            if (nb) nbr_mask |= 1u << i;
            if (sp.min_af <= f) cand_mask |= 1u << i;                 // :183 (chunk membership applied later)
        }
    }

    // ordered tile-local compaction: block exclusive scan of (nbr | cand << 16)
    const uint32_t mine = __popc(nbr_mask) | (__popc(cand_mask) << 16);
    uint32_t incl = mine;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    incl = (decltype(incl))nc_wave_incl_scan((int32_t)incl);
    __shared__ uint32_t wsum[BLOCK / 64 + 1];
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t wpre = 0, total = 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; i++) {
        const uint32_t s = wsum[i];
        if (i < wv) wpre += s;
        total += s;
    }
    const uint32_t excl = wpre + incl - mine;
    int on = excl & 0xFFFF, oc = excl >> 16;
    const int64_t sbase = (int64_t)t * TILE;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (nbr_mask & (1u << i)) stage_nbr[sbase + on++] = P0 + i;
        if (cand_mask & (1u << i)) {
            const int d = i >> 2, k = i & 3;
            const int wi = 2 * d + (k & 1), sh = (k >> 1) * 16;
            const int r = (rw[d] >> (8 * k)) & 0xFF;
            uint32_t alt = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t cb = (wide[b][wi] >> sh) & 0xFFFF;
                if (b != r && cb > alt) alt = cb;
            }
            stage_cpos[sbase + oc] = P0 + i;
            stage_cn[sbase + oc] = (int32_t)((wide[4][wi] >> sh) & 0xFFFF);
            stage_calt[sbase + oc] = (int32_t)alt;
            oc++;
        }
    }
    if (threadIdx.x == 0) tile_cnt[t] = make_int2((int)(total & 0xFFFF), (int)(total >> 16));
}

// single-workgroup exclusive scan of int2 counts (n_tiles is at most a few million).  A WAVE owns a contiguous sixteenth of the counts and walks it 64
// at a time (coalesced): first pass = the wave's total, one barrier for the 16 totals, second pass (the counts come from L2) = a DPP prefix sum per
// step with the running carry in SGPRs, prefixes written coalesced.  The round-1 form scanned 1024 counts per turn across the workgroup -- a load
// latency, twelve ds_bpermute and three barriers per turn, 31 turns for a chr20-sized contig: 52-56 us in front of the scan's totals, on the critical
// path of every contig (a thread-owns-a-run form is no faster: its strided loads are 64 lines per instruction on ONE CU's memory pipeline).
__global__ __launch_bounds__(1024) void k_tile_prefix(const int2 *__restrict__ cnt, int2 *__restrict__ pre, int n,
                                                      int32_t *__restrict__ totals)
{
    __shared__ int2 wsum[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = ((n + 15) / 16 + 63) & ~63;                          // counts per wave, a multiple of 64
    const int b0 = min(n, wv * per), b1 = min(n, b0 + per);
    int2 sum = make_int2(0, 0);
#pragma unroll 4
    for (int i = b0 + lane; i < b1; i += 64) {
        const int2 v = cnt[i];
        sum.x += v.x; sum.y += v.y;
    }
    const int2 winc = make_int2(nc_wave_incl_scan(sum.x), nc_wave_incl_scan(sum.y));
    if (lane == 63) wsum[wv] = winc;
    __syncthreads();
    int cx = 0, cy = 0, tx = 0, ty = 0;
    for (int w = 0; w < 16; w++) {
        const int2 t = wsum[w];
        if (w < wv) { cx += t.x; cy += t.y; }
        tx += t.x; ty += t.y;
    }
#pragma unroll 4
    for (int base = b0; base < b1; base += 64) {                           // (wave-uniform bounds; the loads do not depend on the carry: four in flight)
        const int i = base + lane;
        const int2 v = i < b1 ? cnt[i] : make_int2(0, 0);
        const int ix = nc_wave_incl_scan(v.x), iy = nc_wave_incl_scan(v.y);
        if (i < b1) pre[i] = make_int2(cx + ix - v.x, cy + iy - v.y);
        cx += __builtin_amdgcn_readlane(ix, 63);
        cy += __builtin_amdgcn_readlane(iy, 63);
    }
    if (threadIdx.x == 0) { totals[0] = tx; totals[1] = ty; }
}

__global__ __launch_bounds__(256) void k_compact(int tile, const int2 *__restrict__ cnt, const int2 *__restrict__ pre,
                                                 const int32_t *__restrict__ stage_nbr, const int32_t *__restrict__ stage_cpos,
                                                 const int32_t *__restrict__ stage_cn, const int32_t *__restrict__ stage_calt,
                                                 int32_t *__restrict__ nbr_pos, int32_t *__restrict__ cand_pos,
                                                 int32_t *__restrict__ cand_n, int32_t *__restrict__ cand_alt, int cap_nbr, int cap_cand)
{
    // cap_*: capacity of the outputs in elements.  The host launches this kernel BEFORE it knows the totals (the buffers of the
    // previous scan, a quarter larger than its totals, are reused) and repeats it with larger ones in the rare case they overflow.
    const int t = blockIdx.x;
    const int2 c = cnt[t], p = pre[t];
    const int64_t sb = (int64_t)t * tile;
    for (int i = threadIdx.x; i < c.x; i += 256)
        if (p.x + i < cap_nbr) nbr_pos[p.x + i] = stage_nbr[sb + i];
    for (int i = threadIdx.x; i < c.y; i += 256) {
        if (p.y + i >= cap_cand) continue;
        cand_pos[p.y + i] = stage_cpos[sb + i];
        cand_n[p.y + i] = stage_cn[sb + i];
        cand_alt[p.y + i] = stage_calt[sb + i];
    }
}

__device__ __forceinline__ int lower_bound_dev(const int32_t *a, int n, int32_t key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// candidates of chunk c = cand positions in [start, end] (both inclusive, generate_SNP_pileups.py:183)
// candidates of chunk c: [lo, lo + cnt) of the position-sorted candidate list (both chunk ends inclusive)
__device__ __forceinline__ int chunk_range(const int32_t *__restrict__ cand_pos, int n, int32_t cs, int32_t ce, int32_t *lo_out)
{
    const int lo = lower_bound_dev(cand_pos, n, cs);
    const int hi = ce == INT32_MAX ? n : lower_bound_dev(cand_pos, n, ce + 1);
    *lo_out = lo;
    return hi > lo ? hi - lo : 0;
}

// ranges of every chunk + their exclusive prefix in ONE single-workgroup kernel (the chunk list is a few hundred entries)
__global__ __launch_bounds__(1024) void k_chunk_prefix(const int32_t *__restrict__ cand_pos, int cap_cand, const int32_t *__restrict__ cs,
                                                       const int32_t *__restrict__ ce, int32_t *__restrict__ chunk_lo,
                                                       int32_t *__restrict__ chunk_cnt, int32_t *__restrict__ off, int n,
                                                       int32_t *__restrict__ totals)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_cand = totals[1] < cap_cand ? totals[1] : cap_cand;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        int v = 0;
        if (i < n) {
            int32_t lo;
            v = chunk_range(cand_pos, n_cand, cs[i], ce[i], &lo);
            chunk_lo[i] = lo;
            chunk_cnt[i] = v;
        }
        int inc = v;
        inc = (decltype(inc))nc_wave_incl_scan((int32_t)inc);
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int wp = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            const int s = wsum[w];
            if (w < wv) wp += s;
            tot += s;
        }
        const int c = carry;
        if (i < n) off[i] = c + wp + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { off[n] = carry; totals[2] = carry; }
}

__global__ void k_sites(int n_sites, int n_chunks, const int32_t *__restrict__ chunk_off, const int32_t *__restrict__ chunk_lo,
                        const int32_t *__restrict__ cand_pos, const int32_t *__restrict__ cand_n,
                        const int32_t *__restrict__ cand_alt, int32_t *__restrict__ site_pos,
                        int32_t *__restrict__ site_chunk, int32_t *__restrict__ site_n, int32_t *__restrict__ site_alt)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sites) return;
    // chunk = last c with chunk_off[c] <= s
    int lo = 0, hi = n_chunks;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (chunk_off[mid + 1] <= s) lo = mid + 1; else hi = mid;
    }
    const int c = lo;
    const int ci = chunk_lo[c] + (s - chunk_off[c]);
    site_pos[s] = cand_pos[ci];
    site_chunk[s] = c;
    site_n[s] = cand_n[ci];
    site_alt[s] = cand_alt[ci];
}

// ---- device self-test: the byte-LUT semantics of v_perm_b32 that k_scan relies on
__global__ void k_selftest(uint32_t *out)
{
    const uint32_t x = 0x07040300u | (threadIdx.x & 3);   // bytes: lane&3, 3, 4, 7
    uint32_t r = 0;
    for (int c = 0; c < 5; c++) r |= lut8(x, LUT_LO[c], LUT_HI[c]) << c;
    out[threadIdx.x] = r;
}

}   // namespace

int nc_selftest_device(nc_ctx *ctx)
{
    uint32_t *d = nullptr, h[64];
    NC_HIP(ctx, hipMalloc(&d, sizeof h));
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, ctx->stream, d);
    NC_HIP(ctx, hipGetLastError());
    NC_HIP(ctx, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    NC_HIP(ctx, hipFree(d));
    for (int l = 0; l < 64; l++) {
        // expected: byte0 = code l&3 -> plane (l&3) and present; byte1 = code 3 -> plane 3 + present;
        // byte2 = code 4 -> present only; byte3 = code 7 -> nothing
        uint32_t exp = 0;
        const int c0 = l & 3;
        exp |= (1u << c0) | (1u << 4);
        exp |= ((1u << 3) | (1u << 4)) << 8;
        exp |= (1u << 4) << 16;
        if (h[l] != exp)
            return nc_fail(ctx, NC_ERR_SELFTEST, "v_perm_b32 byte-LUT self-test: lane %d got %08x expected %08x", l, h[l], exp);
    }
    return NC_OK;
}

extern "C" {

// The scan in two halves (round 6): nc_snp_scan_begin enqueues every kernel up to the candidate compaction and the copy of the totals into the pinned
// mailbox and returns WITHOUT waiting; nc_snp_scan_end waits for that copy alone (an event, not the stream), sizes and launches what depends on the
// totals.  A caller may enqueue other work between the two (the previous contig's CNN: the host's round trip for the totals then runs under it
// instead of beside an idle GPU).  nc_snp_scan = begin + end.
static int scan_compact(nc_ctx *ctx, int cap_nbr, int cap_cand)
{
    const int tile = ctx->scan_tile, n_tiles = ctx->scan_n_tiles, n_chunks = ctx->scan_n_chunks;
    hipLaunchKernelGGL(k_compact, dim3(n_tiles), dim3(256), 0, ctx->stream, tile, (int2 *)ctx->tile_cnt.p, (const int2 *)ctx->tile_pre.p,
                       (int32_t *)ctx->stage_nbr.p, (int32_t *)ctx->stage_cpos.p, (int32_t *)ctx->stage_cn.p, (int32_t *)ctx->stage_calt.p,
                       (int32_t *)ctx->nbr_pos.p, (int32_t *)ctx->cand_pos.p, (int32_t *)ctx->cand_n.p, (int32_t *)ctx->cand_alt.p, cap_nbr, cap_cand);
    NC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_chunk_prefix, dim3(1), dim3(1024), 0, ctx->stream, (const int32_t *)ctx->cand_pos.p, cap_cand,
                       (const int32_t *)ctx->chunk_start.p, (const int32_t *)ctx->chunk_end.p, (int32_t *)ctx->chunk_lo.p,
                       (int32_t *)ctx->chunk_cnt.p, (int32_t *)ctx->chunk_off.p, n_chunks, (int32_t *)ctx->totals.p);
    NC_HIP(ctx, hipGetLastError());
    NC_TRY(nc_d2h(ctx, ctx->mbox, ctx->totals.p, 16, ctx->stream));
    if (!ctx->scan_tot_ev) NC_HIP(ctx, hipEventCreateWithFlags(&ctx->scan_tot_ev, hipEventDisableTiming));
    NC_HIP(ctx, hipEventRecord(ctx->scan_tot_ev, ctx->stream));
    return NC_OK;
}
static int cap_of(const DevBuf &b) { return (int)std::min<size_t>(b.cap / 4, (size_t)INT32_MAX); }

int nc_snp_scan_begin(nc_ctx *ctx, const nc_readpack *pack, const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                      int32_t scan_lo, int32_t scan_hi, const nc_scan_params *params, int32_t n_chunks,
                      const int32_t *chunk_start_host, const int32_t *chunk_end_host)
{
    if (!ctx) return NC_ERR_ARG;
    if (!pack || !ref_code_dev || !params || n_chunks <= 0 || !chunk_start_host || !chunk_end_host)
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_scan: null argument");
    const int tile = pack->tile_size;
    if (!(tile == 1024 || tile == 2048 || tile == 4096) || pack->n_tiles <= 0 || !pack->codes || !pack->tile_off ||
        (pack->n_entries && !pack->tile_ent) || (pack->tile_pos0 & 15))
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_scan: malformed read pack");
    // the reference codes must cover the pack's tile grid exactly (host pads with 4 = skip column)
    if (ref_pos0 != pack->tile_pos0 || (int64_t)ref_len < (int64_t)pack->n_tiles * tile)
        return nc_fail(ctx, NC_ERR_ARG, "nc_snp_scan: ref_code must start at tile_pos0 (%d) and cover %lld positions",
                       pack->tile_pos0, (long long)pack->n_tiles * tile);
    if (scan_hi < scan_lo) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_scan: empty scan range");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    ctx->have_scan = false;
    ctx->scan_begun = false;
    const int64_t npos = (int64_t)pack->n_tiles * tile;
    NC_TRY(nc_ensure(ctx, ctx->stage_nbr, npos * 4));
    NC_TRY(nc_ensure(ctx, ctx->stage_cpos, npos * 4));
    NC_TRY(nc_ensure(ctx, ctx->stage_cn, npos * 4));
    NC_TRY(nc_ensure(ctx, ctx->stage_calt, npos * 4));
    NC_TRY(nc_ensure(ctx, ctx->tile_cnt, (size_t)pack->n_tiles * 8));
    NC_TRY(nc_ensure(ctx, ctx->tile_pre, (size_t)pack->n_tiles * 8));
    NC_TRY(nc_ensure(ctx, ctx->totals, 16));
    NC_TRY(nc_ensure(ctx, ctx->chunk_start, (size_t)n_chunks * 4));
    NC_TRY(nc_ensure(ctx, ctx->chunk_end, (size_t)n_chunks * 4));
    NC_TRY(nc_ensure(ctx, ctx->chunk_lo, (size_t)n_chunks * 4));
    NC_TRY(nc_ensure(ctx, ctx->chunk_cnt, (size_t)n_chunks * 4));
    NC_TRY(nc_ensure(ctx, ctx->chunk_off, ((size_t)n_chunks + 1) * 4));
    NC_TRY(nc_h2d_small(ctx, ctx->chunk_start.p, chunk_start_host, (size_t)n_chunks * 4, ctx->stream));
    NC_TRY(nc_h2d_small(ctx, ctx->chunk_end.p, chunk_end_host, (size_t)n_chunks * 4, ctx->stream));

    ScanP sp;
    sp.mincov = params->mincov;
    sp.haploid = params->haploid;
    sp.min_af = params->min_allele_freq;
    sp.t0 = params->nbr_t0;
    sp.t1 = params->nbr_t1;
    sp.scan_lo = scan_lo;
    sp.scan_hi = scan_hi;

    NcTimer tm(ctx, 0);
    auto *sn = (int32_t *)ctx->stage_nbr.p;
    auto *sc = (int32_t *)ctx->stage_cpos.p;
    auto *scn = (int32_t *)ctx->stage_cn.p;
    auto *sca = (int32_t *)ctx->stage_calt.p;
    auto *tc = (int2 *)ctx->tile_cnt.p;
    if (tile == 1024)
        hipLaunchKernelGGL(k_scan<64>, dim3(pack->n_tiles), dim3(64), 0, ctx->stream, pack->codes, pack->tile_off,
                           pack->tile_ent, pack->tile_pos0, ref_code_dev, sp, sn, sc, scn, sca, tc);
    else if (tile == 2048)
        hipLaunchKernelGGL(k_scan<128>, dim3(pack->n_tiles), dim3(128), 0, ctx->stream, pack->codes, pack->tile_off,
                           pack->tile_ent, pack->tile_pos0, ref_code_dev, sp, sn, sc, scn, sca, tc);
    else
        hipLaunchKernelGGL(k_scan<256>, dim3(pack->n_tiles), dim3(256), 0, ctx->stream, pack->codes, pack->tile_off,
                           pack->tile_ent, pack->tile_pos0, ref_code_dev, sp, sn, sc, scn, sca, tc);
    NC_HIP(ctx, hipGetLastError());
    tm.stop();
    hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(1024), 0, ctx->stream, tc, (int2 *)ctx->tile_pre.p, pack->n_tiles,
                       (int32_t *)ctx->totals.p);
    NC_HIP(ctx, hipGetLastError());
    ctx->scan_tile = tile;
    ctx->scan_n_tiles = pack->n_tiles;
    ctx->scan_n_chunks = n_chunks;
    volatile int32_t *tot = ctx->mbox;                           // pinned mailbox: the totals arrive through a copy kernel
    ctx->scan_cap_nbr = cap_of(ctx->nbr_pos);
    ctx->scan_cap_cand = std::min(cap_of(ctx->cand_pos), std::min(cap_of(ctx->cand_n), cap_of(ctx->cand_alt)));
    if (ctx->scan_cap_nbr == 0 || ctx->scan_cap_cand == 0) {
        // first scan of this context: one round trip for the totals that size the outputs
        NC_TRY(nc_d2h(ctx, ctx->mbox, ctx->totals.p, 16, ctx->stream));
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NC_TRY(nc_ensure(ctx, ctx->nbr_pos, (size_t)(tot[0] + 1) * 4));
        NC_TRY(nc_ensure(ctx, ctx->cand_pos, (size_t)(tot[1] + 1) * 4));
        NC_TRY(nc_ensure(ctx, ctx->cand_n, (size_t)(tot[1] + 1) * 4));
        NC_TRY(nc_ensure(ctx, ctx->cand_alt, (size_t)(tot[1] + 1) * 4));
        ctx->scan_cap_nbr = cap_of(ctx->nbr_pos);
        ctx->scan_cap_cand = std::min(cap_of(ctx->cand_pos), std::min(cap_of(ctx->cand_n), cap_of(ctx->cand_alt)));
    }
    // otherwise: the outputs of the previous scan (a quarter larger than its totals) are reused without asking first -- ONE host round trip per scan,
    // taken in nc_snp_scan_end; the kernels never write past the capacities
    NC_TRY(scan_compact(ctx, ctx->scan_cap_nbr, ctx->scan_cap_cand));
    ctx->scan_begun = true;
    return NC_OK;
}

int nc_snp_scan_end(nc_ctx *ctx, int32_t *n_nbr, int32_t *n_cand, int32_t *n_sites)
{
    if (!ctx) return NC_ERR_ARG;
    if (!ctx->scan_begun) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_scan_end: no nc_snp_scan_begin on this context");
    ctx->scan_begun = false;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    volatile int32_t *tot = ctx->mbox;
    NC_HIP(ctx, hipEventSynchronize(ctx->scan_tot_ev));          // the totals' copy alone: work enqueued behind it keeps running
    if (!(tot[0] <= ctx->scan_cap_nbr && tot[1] <= ctx->scan_cap_cand)) {
        // they did not fit: grow and repeat the compaction (rare)
        NC_TRY(nc_ensure(ctx, ctx->nbr_pos, (size_t)(tot[0] + 1) * 4));
        NC_TRY(nc_ensure(ctx, ctx->cand_pos, (size_t)(tot[1] + 1) * 4));
        NC_TRY(nc_ensure(ctx, ctx->cand_n, (size_t)(tot[1] + 1) * 4));
        NC_TRY(nc_ensure(ctx, ctx->cand_alt, (size_t)(tot[1] + 1) * 4));
        ctx->scan_cap_nbr = cap_of(ctx->nbr_pos);
        ctx->scan_cap_cand = std::min(cap_of(ctx->cand_pos), std::min(cap_of(ctx->cand_n), cap_of(ctx->cand_alt)));
        NC_TRY(scan_compact(ctx, ctx->scan_cap_nbr, ctx->scan_cap_cand));
        NC_HIP(ctx, hipEventSynchronize(ctx->scan_tot_ev));
        if (!(tot[0] <= ctx->scan_cap_nbr && tot[1] <= ctx->scan_cap_cand)) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_scan: outputs still too small after regrowth");
    }
    const int32_t n_chunks = ctx->scan_n_chunks;
    ctx->n_nbr = tot[0];
    ctx->n_cand = tot[1];
    ctx->n_sites = tot[2];
    ctx->n_chunks = n_chunks;
    const size_t ns = (size_t)tot[2] + 1;
    NC_TRY(nc_ensure(ctx, ctx->site_pos, ns * 4));
    NC_TRY(nc_ensure(ctx, ctx->site_chunk, ns * 4));
    NC_TRY(nc_ensure(ctx, ctx->site_n, ns * 4));
    NC_TRY(nc_ensure(ctx, ctx->site_alt, ns * 4));
    if (tot[2] > 0) {
        hipLaunchKernelGGL(k_sites, dim3((tot[2] + 255) / 256), dim3(256), 0, ctx->stream, tot[2], n_chunks,
                           (const int32_t *)ctx->chunk_off.p, (const int32_t *)ctx->chunk_lo.p,
                           (const int32_t *)ctx->cand_pos.p, (const int32_t *)ctx->cand_n.p, (const int32_t *)ctx->cand_alt.p,
                           (int32_t *)ctx->site_pos.p, (int32_t *)ctx->site_chunk.p, (int32_t *)ctx->site_n.p,
                           (int32_t *)ctx->site_alt.p);
        NC_HIP(ctx, hipGetLastError());
    }
    // k_sites runs behind the last synchronisation: copies of the site arrays from another stream wait for this event
    if (!ctx->scan_ev) NC_HIP(ctx, hipEventCreateWithFlags(&ctx->scan_ev, hipEventDisableTiming));
    NC_HIP(ctx, hipEventRecord(ctx->scan_ev, ctx->stream));
    ctx->have_scan = true;
    if (n_nbr) *n_nbr = tot[0];
    if (n_cand) *n_cand = tot[1];
    if (n_sites) *n_sites = tot[2];
    return NC_OK;
}

int nc_snp_scan(nc_ctx *ctx, const nc_readpack *pack, const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                int32_t scan_lo, int32_t scan_hi, const nc_scan_params *params, int32_t n_chunks,
                const int32_t *chunk_start_host, const int32_t *chunk_end_host, int32_t *n_nbr, int32_t *n_cand,
                int32_t *n_sites)
{
    const int rc = nc_snp_scan_begin(ctx, pack, ref_code_dev, ref_pos0, ref_len, scan_lo, scan_hi, params, n_chunks, chunk_start_host, chunk_end_host);
    return rc != NC_OK ? rc : nc_snp_scan_end(ctx, n_nbr, n_cand, n_sites);
}

static int scan_fetch(nc_ctx *ctx, hipStream_t st, bool wait, int32_t *nbr_pos, int32_t *site_pos, int32_t *site_chunk, int32_t *site_n,
                      int32_t *site_alt)
{
    if (!ctx) return NC_ERR_ARG;
    if (!ctx->have_scan) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_scan_fetch: no scan on this context");
    const size_t nb = (size_t)ctx->n_nbr * 4, ns = (size_t)ctx->n_sites * 4;
    if (nbr_pos && nb) NC_TRY(nc_d2h(ctx, nbr_pos, ctx->nbr_pos.p, nb, st));
    if (site_pos && ns) NC_TRY(nc_d2h(ctx, site_pos, ctx->site_pos.p, ns, st));
    if (site_chunk && ns) NC_TRY(nc_d2h(ctx, site_chunk, ctx->site_chunk.p, ns, st));
    if (site_n && ns) NC_TRY(nc_d2h(ctx, site_n, ctx->site_n.p, ns, st));
    if (site_alt && ns) NC_TRY(nc_d2h(ctx, site_alt, ctx->site_alt.p, ns, st));
    if (wait) NC_HIP(ctx, hipStreamSynchronize(st));
    return NC_OK;
}

int nc_snp_scan_fetch(nc_ctx *ctx, int32_t *nbr_pos, int32_t *site_pos, int32_t *site_chunk, int32_t *site_n,
                      int32_t *site_alt)
{
    if (!ctx) return NC_ERR_ARG;
    return scan_fetch(ctx, ctx->stream, true, nbr_pos, site_pos, site_chunk, site_n, site_alt);
}

// The copies are ordered behind the scan's last kernel (scan_ev) on whichever stream they are put
int nc_snp_scan_fetch_async(nc_ctx *ctx, void *copy_stream, int32_t *nbr_pos, int32_t *site_pos, int32_t *site_chunk,
                            int32_t *site_n, int32_t *site_alt)
{
    if (!ctx) return NC_ERR_ARG;
    if (copy_stream && (hipStream_t)copy_stream != ctx->stream && ctx->have_scan && ctx->scan_ev)
        NC_HIP(ctx, hipStreamWaitEvent((hipStream_t)copy_stream, ctx->scan_ev, 0));
    return scan_fetch(ctx, copy_stream ? (hipStream_t)copy_stream : ctx->stream, false, nbr_pos, site_pos, site_chunk, site_n, site_alt);
}

}   // extern "C"
