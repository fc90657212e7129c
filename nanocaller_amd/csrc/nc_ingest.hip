// BAM records on the device: what nc_bam_decode (nc_bam.cpp) does on host threads between the inflated BGZF stream and the read pack, for
// the SNP route (generate_SNP_pileups.py:134-164's input: pysam's pileup over a coordinate-sorted BAM).  The stream is inflated in HBM by
// nc_inflate_device; from there
//   k_walk    record boundaries.  A record only says where the NEXT one starts, so the walk is a chain -- cut at the index: every entry
//             of the .bai linear index is the virtual offset of a record start, one lane walks from each to the next (about 16 kb of
//             reference, some hundred records), once to count and once to write the offsets;
//   k_meta    one lane per record: the fixed fields, the CIGAR's reference span and query length (the real CIGAR of the CG tag for
//             ultra-long reads, SAMv1 4.2.2), reference skips, the HP / PS tags, a hash of the read name;
//   k_codes   one wave per kept read: the CIGAR in steps of 64 operations, reference and query offsets by wave scans, every M / = / X run
//             written as base codes (BAM's 4-bit bases through a 16-entry table), D / N runs as code 4, into the read's 16-byte aligned
//             slot of the position-addressed pack -- byte for byte what nc_pack_fill writes from nc_bam_decode's arrays.
// Which reads are kept (flag filter, depth cap, unsupported inputs) and the tile index stay the host's decisions, on the arrays of k_meta.
#include "nc_common.h"

#include <cstring>

namespace {

__device__ __forceinline__ uint32_t ldu32(const uint8_t *p)          // (records are packed: nothing is aligned)
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
__device__ __forceinline__ int32_t ld32(const uint8_t *p) { return (int32_t)ldu32(p); }

// ---- record boundaries
template <bool FILL>
__global__ __launch_bounds__(64) void k_walk(const uint8_t *raw, int64_t raw_len, int32_t n_seeds, const int64_t *seed, const int32_t *seed_tid,
                                             const int64_t *first, int64_t *out, int32_t *status)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_seeds) return;
    int64_t p = seed[i];
    const int64_t stop = i + 1 < n_seeds ? seed[i + 1] : raw_len;
    const int32_t tid = seed_tid[i];
    const int64_t w = FILL ? first[i] : 0;
    int64_t n = 0;
    while (p < stop) {
        if (p + 36 > raw_len) break;                                  // the end of the data that was inflated
        const int32_t bs = ld32(raw + p);
        if (bs < 32) { atomicOr(status, 1); break; }
        if (p + 4 + bs > raw_len) break;
        if (ld32(raw + p + 4) != tid) break;                          // the next contig's records (they start at its own index entries)
        if (FILL) out[w + n] = p;
        n++;
        p += 4 + (int64_t)bs;
    }
    if (p > stop && i + 1 < n_seeds) atomicOr(status, 2);             // an index entry that is not a record start
    if (!FILL) out[i] = n;
}

// ---- per-record fields
static_assert(NC_BAM_META_COLS == 12, "k_meta writes twelve columns");
enum { M_REFID, M_POS, M_FLAG, M_RLEN, M_LSEQ, M_HASSEQ, M_HAP, M_PS, M_HASH_LO, M_HASH_HI, M_NCIG, M_CIGD };

__global__ __launch_bounds__(64) void k_meta(const uint8_t *raw, int64_t n_rec, const int64_t *rec_off, int32_t *meta, int32_t *status)
{
    const int64_t r = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n_rec) return;
    const uint8_t *p = raw + rec_off[r];
    const int32_t bs = ld32(p);
    p += 4;
    const int32_t refid = ld32(p), pos = ld32(p + 4);
    const int l_name = p[8];
    const int n_cig = p[12] | (p[13] << 8), flag = p[14] | (p[15] << 8);
    const int32_t l_seq = ld32(p + 16);
    const uint8_t *name = p + 32, *cig = name + l_name;
    int64_t rlen = 0, qlen = 0;
    int32_t ncr = n_cig, hp = 0, ps = 0, refskip = 0, bad = 0;
    uint64_t h = 1469598103934665603ull;                              // FNV-1a of the read name
    if (l_seq < 0 || 32 + (int64_t)l_name + 4 * (int64_t)n_cig + ((int64_t)l_seq + 1) / 2 + l_seq > bs) bad = 1;
    if (!bad) {
        for (int k = 0; k < l_name; k++) h = (h ^ name[k]) * 1099511628211ull;
        const uint8_t *seq = cig + 4 * (size_t)n_cig, *aux = seq + ((size_t)l_seq + 1) / 2 + l_seq, *aux_end = p + bs;
        const bool placeholder = n_cig == 2 && (ldu32(cig) & 15) == 4 && (int64_t)(ldu32(cig) >> 4) == l_seq && (ldu32(cig + 4) & 15) == 3;
        // one pass over the tags: HP, PS, and the real CIGAR behind the placeholder <l_seq>S<ref_len>N
        for (const uint8_t *a = aux; a + 3 <= aux_end;) {
            const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
            a += 3;
            int64_t iv = 0;
            bool is_int = true;
            switch (ty) {
            case 'c': iv = (int8_t)a[0]; a += 1; break;
            case 'C': iv = a[0]; a += 1; break;
            case 's': iv = (int16_t)(a[0] | (a[1] << 8)); a += 2; break;
            case 'S': iv = a[0] | (a[1] << 8); a += 2; break;
            case 'i': iv = ld32(a); a += 4; break;
            case 'I': iv = ldu32(a); a += 4; break;
            case 'A': a += 1; is_int = false; break;
            case 'f': a += 4; is_int = false; break;
            case 'Z': case 'H': while (a < aux_end && *a) a++; a++; is_int = false; break;
            case 'B': {
                if (a + 5 > aux_end) { a = aux_end; is_int = false; break; }
                const char st = (char)a[0];
                const uint32_t cnt = ldu32(a + 1);
                const int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                if (placeholder && t0 == 'C' && t1 == 'G' && st == 'I' && a + 5 + (size_t)cnt * 4 <= aux_end) { cig = a + 5; ncr = (int32_t)cnt; }
                a += 5 + (size_t)cnt * es;
                is_int = false;
                break; }
            default: a = aux_end; is_int = false; break;
            }
            if (is_int && t0 == 'H' && t1 == 'P') hp = (int)iv;
            if (is_int && t0 == 'P' && t1 == 'S') ps = (int)iv;
        }
        for (int64_t k = 0; k < ncr; k++) {
            const uint32_t c = ldu32(cig + 4 * k);
            const int op = c & 15, len = (int)(c >> 4);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += len;
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += len;
            if (op == 3) refskip = 1;
        }
    } else atomicOr(status, 4);
    // (nc_bam_decode: a span that is empty, or does not fit 32-bit positions, is not an alignment of the contig)
    const bool span_ok = !bad && rlen > 0 && rlen <= (int64_t)INT32_MAX - pos - 2;
    meta[M_REFID * n_rec + r] = refid;
    meta[M_POS * n_rec + r] = pos;
    meta[M_FLAG * n_rec + r] = flag | (refskip ? NC_FLAG_REFSKIP : 0);
    meta[M_RLEN * n_rec + r] = span_ok ? (int32_t)rlen : 0;
    meta[M_LSEQ * n_rec + r] = l_seq;
    meta[M_HASSEQ * n_rec + r] = qlen <= (int64_t)l_seq;
    meta[M_HAP * n_rec + r] = (hp == 1 || hp == 2) ? hp : 0;
    meta[M_PS * n_rec + r] = ps;
    meta[M_HASH_LO * n_rec + r] = (int32_t)(uint32_t)h;
    meta[M_HASH_HI * n_rec + r] = (int32_t)(uint32_t)(h >> 32);
    meta[M_NCIG * n_rec + r] = ncr;
    meta[M_CIGD * n_rec + r] = (int32_t)(cig - p);
}

// ---- the slots of the pack
__device__ __forceinline__ int wave_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    return v;
}

// BAM's 4-bit base "=ACMGRSVTWYHKDBN" -> A0 G1 T2 C3, everything else 4 (NT16_CODE of nc_bam.cpp), one nibble per entry
constexpr uint64_t NT16 = 0x4444444244414304ull;

__device__ __forceinline__ uint8_t base_code(const uint8_t *seq, int q)
{
    const uint32_t b = seq[q >> 1];
    const uint32_t nib = (q & 1) ? (b & 15u) : (b >> 4);
    return (uint8_t)((NT16 >> (4 * nib)) & 15u);
}

constexpr int WPB = 4;                                                 // waves (reads) per workgroup

__global__ __launch_bounds__(64 * WPB) void k_codes(const uint8_t *raw, int32_t n_reads, const int64_t *rec, const int64_t *slot, const int32_t *cigd,
                                                    const int32_t *ncig, const int32_t *start, uint8_t *codes)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const uint8_t *p = raw + rec[r] + 4;
    const int l_name = p[8], n_cig_field = p[12] | (p[13] << 8);
    const uint8_t *cig = p + cigd[r];
    const uint8_t *seq = p + 32 + l_name + 4 * (size_t)n_cig_field;
    const int nc = ncig[r] & 0x7fffffff;
    const bool has_seq = ncig[r] >= 0;
    uint8_t *out = codes + slot[r] + (start[r] & 15);                  // the slot starts at floor16(start)
    int rp = 0, qp = 0;
#pragma unroll 1
    for (int k0 = 0; k0 < nc; k0 += 64) {
        const int k = k0 + lane;
        const uint32_t c = k < nc ? ldu32(cig + 4 * (size_t)k) : 15u;
        const int op = c & 15, len = (int)(c >> 4);
        const bool m = op == 0 || op == 7 || op == 8, gap = op == 2 || op == 3;
        const int radv = (m || gap) ? len : 0, qadv = (m || op == 1 || op == 4) ? len : 0;
        const int ri = wave_scan(radv, lane), qi = wave_scan(qadv, lane);
        const int r0 = rp + ri - radv, q0 = qp + qi - qadv;
        const bool wr = (m || gap) && len > 0;
        const bool longrun = wr && len > 48;
        if (wr && !longrun) {
            if (m && has_seq) for (int i = 0; i < len; i++) out[r0 + i] = base_code(seq, q0 + i);
            else for (int i = 0; i < len; i++) out[r0 + i] = 4;
        }
        uint64_t lm = __ballot(longrun);                               // long runs (HiFi: thousands of bases between events): the whole wave
#pragma unroll 1
        while (lm) {
            const int j = __builtin_ctzll(lm);
            lm &= lm - 1;
            const int R0 = __builtin_amdgcn_readlane(r0, j), Q0 = __builtin_amdgcn_readlane(q0, j), L = __builtin_amdgcn_readlane(len, j);
            const bool M = __builtin_amdgcn_readlane((int)(m && has_seq), j) != 0;
            if (M) for (int i = lane; i < L; i += 64) out[R0 + i] = base_code(seq, Q0 + i);
            else for (int i = lane; i < L; i += 64) out[R0 + i] = 4;
        }
        rp += __builtin_amdgcn_readlane(ri, 63);
        qp += __builtin_amdgcn_readlane(qi, 63);
    }
}

// ---- the indel path's per-read sections (nc_bam_decode's events + nc_indel_pack_build, nc_bam.cpp): one lane per kept read walks its CIGAR
//      FILL == false: counts[0..3)[r] = insertion / deletion events ('+n' / '-n' markers on the previous reference column: an I or D with no
//      column before it has none), inserted bases of those events, query bases behind the last aligned one (at most tail_cap);
//      FILL == true: the events, where each one's inserted bases start, the bases themselves (codes), the tail -- at the offsets the counts'
//      prefix sums give
template <bool FILL>
__global__ __launch_bounds__(64) void k_indel_reads(const uint8_t *raw, int32_t n_reads, const int64_t *rec, const int32_t *cigd, const int32_t *ncig,
                                                    int32_t tail_cap, int32_t *counts, const int32_t *ev_off, const int32_t *ins_base, const int32_t *tail_off,
                                                    int32_t *ev_pos, int32_t *ev_len, int32_t *ins_off, uint8_t *ins_bases, uint8_t *tail_bases)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads) return;
    const uint8_t *p = raw + rec[r] + 4;
    const int32_t pos = ld32(p + 4);
    const int l_name = p[8], n_cig_field = p[12] | (p[13] << 8);
    const int64_t L = ld32(p + 16);                                    // bases the record carries (0: SEQ '*')
    const uint8_t *cig = p + cigd[r];
    const uint8_t *seq = p + 32 + l_name + 4 * (size_t)n_cig_field;
    const int nc = ncig[r] & 0x7fffffff;
    int64_t qp = 0, rp = 0;
    int32_t ne = 0, ni = 0;
    int32_t e = FILL ? ev_off[r] : 0, io = FILL ? ins_base[r] : 0;
    for (int k = 0; k < nc; k++) {
        const uint32_t c = ldu32(cig + 4 * (size_t)k);
        const int op = c & 15, len = (int)(c >> 4);
        switch (op) {
        case 0: case 7: case 8: rp += len; qp += len; break;
        case 1:
            if (rp > 0) {
                const int64_t a = qp < L ? qp : L, b = qp + len < L ? qp + len : L;
                if (FILL) {
                    ev_pos[e] = pos + (int32_t)rp;
                    ev_len[e] = len;
                    ins_off[e] = io;
                    for (int64_t q = a; q < b; q++) ins_bases[io++] = base_code(seq, (int)q);
                    e++;
                } else { ne++; ni += (int32_t)(b - a); }
            }
            qp += len;
            break;
        case 2:
            if (rp > 0) {
                if (FILL) { ev_pos[e] = pos + (int32_t)rp; ev_len[e] = -len; ins_off[e] = io; e++; }
                else ne++;
            }
            rp += len;
            break;
        case 3: rp += len; break;
        case 4: if (rp == 0) qp += len; break;                         // (a leading clip; a trailing one is the tail)
        default: break;
        }
    }
    const int64_t a = qp < L ? qp : L, b = qp + tail_cap < L ? qp + tail_cap : L;
    if (FILL) {
        int32_t t = tail_off[r];
        for (int64_t q = a; q < b; q++) tail_bases[t++] = base_code(seq, (int)q);
    } else {
        counts[r] = ne;
        counts[n_reads + r] = ni;
        counts[2 * n_reads + r] = (int32_t)(b - a);
    }
}

}   // namespace

extern "C" {

// members of a BGZF file image (SAMv1 4.1): the deflate payload of member k lies at data[coff[k] .. coff[k] + clen[k]) and inflates to isize[k]
// bytes.  Host side, no GPU.  nc_bgzf_scan walks from byte `start` as far as whole members lie inside data[0, n) and the outputs have room
// (`cap`): *n_members of them, *next = where the walk stopped (== n: the image ends with a whole member) -- a file that is still being read
// is scanned piece by piece.  nc_bgzf_members is the whole image at once: NC_ERR_CAPACITY when there are more than `cap` members
// (n_members still counts them all), NC_ERR_ARG when the image does not end with a whole member.
static int bgzf_member(const uint8_t *data, int64_t n, int64_t o, int64_t *bsize_out, int *xlen_out)
{
    // -> 0 a whole member at o, 1 not all of it inside [0, n), -1 not a BGZF member
    if (o + 18 > n) return 1;
    if (data[o] != 0x1f || data[o + 1] != 0x8b || data[o + 2] != 8 || !(data[o + 3] & 4)) return -1;
    const int xlen = data[o + 10] | (data[o + 11] << 8);
    if (o + 12 + xlen > n) return 1;
    int64_t bsize = -1;
    for (int64_t x = o + 12; x + 4 <= o + 12 + xlen;) {
        const int slen = data[x + 2] | (data[x + 3] << 8);
        if (data[x] == 'B' && data[x + 1] == 'C' && slen == 2 && x + 6 <= o + 12 + xlen) bsize = (int64_t)(data[x + 4] | (data[x + 5] << 8)) + 1;
        x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8) return -1;
    if (o + bsize > n) return 1;
    *bsize_out = bsize;
    *xlen_out = xlen;
    return 0;
}

int nc_bgzf_scan(const uint8_t *data, int64_t n, int64_t start, int64_t cap, int64_t *coff, int32_t *clen, int32_t *isize, int64_t *n_members,
                 int64_t *next)
{
    if (!data || n < 0 || start < 0 || start > n || cap < 0 || !n_members || !next || (cap && (!coff || !clen || !isize))) return NC_ERR_ARG;
    int64_t o = start, k = 0;
    while (o < n && k < cap) {
        int64_t bsize = 0;
        int xlen = 0;
        const int st = bgzf_member(data, n, o, &bsize, &xlen);
        if (st < 0) return NC_ERR_ARG;
        if (st > 0) break;
        coff[k] = o + 12 + xlen;
        clen[k] = (int32_t)(bsize - xlen - 20);
        uint32_t is;
        memcpy(&is, data + o + bsize - 4, 4);
        if (is > 65536u) return NC_ERR_ARG;
        isize[k] = (int32_t)is;
        k++;
        o += bsize;
    }
    *n_members = k;
    *next = o;
    return NC_OK;
}

int nc_bgzf_members(const uint8_t *data, int64_t n, int64_t cap, int64_t *coff, int32_t *clen, int32_t *isize, int64_t *n_members)
{
    if (!data || n < 0 || cap < 0 || !n_members || (cap && (!coff || !clen || !isize))) return NC_ERR_ARG;
    int64_t k = 0, next = 0;
    const int rc = nc_bgzf_scan(data, n, 0, cap, coff, clen, isize, &k, &next);
    if (rc != NC_OK) return rc;
    int64_t total = k;
    while (next < n) {                                               // out of room, or a cut member: count on without storing
        int64_t bsize = 0;
        int xlen = 0;
        if (bgzf_member(data, n, next, &bsize, &xlen) != 0) return NC_ERR_ARG;
        total++;
        next += bsize;
    }
    *n_members = total;
    return total > cap ? NC_ERR_CAPACITY : NC_OK;
}

// Record boundaries of the inflated stream d_raw[0, raw_len): n_seeds record starts in ascending order (the .bai linear index entries, as
// offsets into d_raw) with the contig each belongs to.  d_first == NULL: d_out[i] = records from seed i up to seed i + 1 (or the end of
// the contig's records); else d_out[d_first[i] + k] = offset of the k-th of them (d_first = the exclusive prefix sums of the counts).
// d_status (one int32, zeroed by the caller) collects: 1 a block_size below 32, 2 an index entry that is not a record start.
int nc_bam_walk(nc_ctx *ctx, const uint8_t *d_raw, int64_t raw_len, int32_t n_seeds, const int64_t *d_seed, const int32_t *d_seed_tid,
                const int64_t *d_first, int64_t *d_out, int32_t *d_status)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_seeds < 0 || raw_len < 0 || (n_seeds && (!d_raw || !d_seed || !d_seed_tid || !d_out || !d_status)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_bam_walk: bad argument");
    if (n_seeds == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid((n_seeds + 63) / 64), block(64);
    if (d_first) hipLaunchKernelGGL(k_walk<true>, grid, block, 0, ctx->stream, d_raw, raw_len, n_seeds, d_seed, d_seed_tid, d_first, d_out, d_status);
    else hipLaunchKernelGGL(k_walk<false>, grid, block, 0, ctx->stream, d_raw, raw_len, n_seeds, d_seed, d_seed_tid, d_first, d_out, d_status);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

// Per-record fields: d_meta is int32 [NC_BAM_META_COLS][n_rec] (column-major by field: include/nanocaller_hip.h).  d_status: 4 = a record
// whose fields do not fit its block_size.
int nc_bam_meta(nc_ctx *ctx, const uint8_t *d_raw, int64_t n_rec, const int64_t *d_rec_off, int32_t *d_meta, int32_t *d_status)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_rec < 0 || (n_rec && (!d_raw || !d_rec_off || !d_meta || !d_status))) return nc_fail(ctx, NC_ERR_ARG, "nc_bam_meta: bad argument");
    if (n_rec == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_meta, dim3((unsigned)((n_rec + 63) / 64)), dim3(64), 0, ctx->stream, d_raw, n_rec, d_rec_off, d_meta, d_status);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

// The slots of n_reads kept reads: d_rec[r] = the record's offset in d_raw, d_slot[r] = byte offset of its slot in d_codes (nc_pack_fill's
// layout: consecutive, [floor16(start), ceil16(end))), d_cigd / d_ncig = columns CIGD / NCIG of nc_bam_meta (bit 31 of d_ncig set: the
// record has fewer bases than its CIGAR consumes, every aligned position is code 4), d_start = 1-based first position.  d_codes must hold
// NC_CODE_ABSENT everywhere beforehand (hipMemset): only covered positions are written.
int nc_bam_codes(nc_ctx *ctx, const uint8_t *d_raw, int32_t n_reads, const int64_t *d_rec, const int64_t *d_slot, const int32_t *d_cigd,
                 const int32_t *d_ncig, const int32_t *d_start, uint8_t *d_codes)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_reads < 0 || (n_reads && (!d_raw || !d_rec || !d_slot || !d_cigd || !d_ncig || !d_start || !d_codes)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_bam_codes: bad argument");
    if (n_reads == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_codes, dim3((n_reads + WPB - 1) / WPB), dim3(64 * WPB), 0, ctx->stream, d_raw, n_reads, d_rec, d_slot, d_cigd, d_ncig, d_start,
                       d_codes);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

// The indel path's per-read sections of n_reads kept reads (d_rec / d_cigd / d_ncig as nc_bam_codes takes them).  Pass 1 (d_ev_off == NULL):
// d_counts = int32 [3][n_reads]: events, inserted bases, tail bases (at most tail_cap behind the last aligned base) of every read.  Pass 2:
// d_ev_off / d_ins_base / d_tail_off = the exclusive prefix sums of those; fills d_ev_pos / d_ev_len / d_ins_off (one entry per event; the
// caller sets d_ins_off[n_events] = total) / d_ins_bases / d_tail_bases -- the arrays nc_bam_decode + nc_indel_pack_build make on the host.
int nc_bam_indel_reads(nc_ctx *ctx, const uint8_t *d_raw, int32_t n_reads, const int64_t *d_rec, const int32_t *d_cigd, const int32_t *d_ncig,
                       int32_t tail_cap, int32_t *d_counts, const int32_t *d_ev_off, const int32_t *d_ins_base, const int32_t *d_tail_off,
                       int32_t *d_ev_pos, int32_t *d_ev_len, int32_t *d_ins_off, uint8_t *d_ins_bases, uint8_t *d_tail_bases)
{
    if (!ctx) return NC_ERR_ARG;
    const bool fill = d_ev_off != nullptr;
    if (n_reads < 0 || tail_cap < 0 || (n_reads && (!d_raw || !d_rec || !d_cigd || !d_ncig)) || (n_reads && !fill && !d_counts) ||
        (n_reads && fill && (!d_ins_base || !d_tail_off || !d_ev_pos || !d_ev_len || !d_ins_off || !d_ins_bases || !d_tail_bases)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_bam_indel_reads: bad argument");
    if (n_reads == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid((n_reads + 63) / 64), block(64);
    if (fill)
        hipLaunchKernelGGL(k_indel_reads<true>, grid, block, 0, ctx->stream, d_raw, n_reads, d_rec, d_cigd, d_ncig, tail_cap, d_counts, d_ev_off, d_ins_base,
                           d_tail_off, d_ev_pos, d_ev_len, d_ins_off, d_ins_bases, d_tail_bases);
    else
        hipLaunchKernelGGL(k_indel_reads<false>, grid, block, 0, ctx->stream, d_raw, n_reads, d_rec, d_cigd, d_ncig, tail_cap, d_counts, d_ev_off, d_ins_base,
                           d_tail_off, d_ev_pos, d_ev_len, d_ins_off, d_ins_bases, d_tail_bases);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

}   // extern "C"
