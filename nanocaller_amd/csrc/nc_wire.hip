// Wire pack: the host -> device transfer form of the read pack (the timed region starts at decoded alignments
// in pinned host memory; the reference feeds every chunk from the host, snpCaller.py:86, generate_SNP_pileups.py:156).
//
// The read pack costs 1 B per pileup entry -- 1.93 GB for a chr20-sized 30x contig, i.e. ~35 ms of PCIe Gen5 against
// ~11.5 ms of GPU work per contig.  But a decoded alignment is almost the reference: 92 % of an ONT read's positions (99.8 %
// of a HiFi read's) carry the reference base.  The transfer form therefore stores only the DIFFERENCES against the reference:
//   * the read table (start, end, slot offset) of the kept reads, 16 B per read;
//   * one byte per reference position (`ref_wire`: base code in bits 0-2, bit 3 = column skipped);
//   * per 1024-byte block of the codes array the index of the first read reaching into it and its difference events,
//     2 B each: (byte offset in block) | code << 12.
// nc_wire_expand rebuilds the position-addressed codes in HBM (one wave per block: reference bases through LDS, events
// scattered on top, one coalesced dwordx4 store per lane) -- byte for byte what nc_pack_fill writes.  ONT 30x chr20:
// 0.39 GB over PCIe instead of 1.93 GB; the expansion is an HBM-write-bound kernel (1 B written per pileup entry).
#include <algorithm>
#include <thread>
#include <vector>

#include <cstdlib>
#include <emmintrin.h>

#include "nc_common.h"
#include "nc_host.h"

#define WIRE_BLOCK 1024
#ifndef NC_WIRE_U
#define NC_WIRE_U 4
#endif

struct nc_wire {
    std::vector<int32_t> rd_start, rd_end;
    std::vector<int64_t> slot_off;
    std::vector<uint32_t> blk_off;
    std::vector<int32_t> blk_read;
    std::vector<uint16_t> events;
    std::vector<uint32_t> blk_ev;    // nc_wire_build_del: per block the cursor into the kept reads' indel events (0xffffffff: nothing implied in this block)
    std::vector<uint8_t> ev_bytes;   // nc_wire_build_bytes: the events one byte each (then `events` is empty and blk_off counts bytes)
    int64_t codes_len = 0;
};

// The difference events one byte each (round 6): bits 2-7 = columns skipped since the block's previous event (0 .. 62; 63: a filler that skips 63 columns and
// is no event), bits 0-1 = WHICH other code -- an event's code differs from the predicted one, so two bits tell it: against a predicted base b (0 .. 3)
// 0 / 1 / 2 = base (b + 1 + k) & 3, 3 = code 4; against a predicted 4 the base itself.  63 M events of a chr20-sized ONT contig: 126 -> 72 MB.
static inline unsigned wire_which(unsigned c, unsigned pred) { return pred < 4u ? (c == 4u ? 3u : ((c - pred - 1u) & 3u)) : c; }
// (the builder's intermediate events carry the predicted code in the three spare bits 10, 11, 15)
static inline uint16_t wire_stash(unsigned pred) { return (uint16_t)(((pred & 3u) << 10) | ((pred >> 2) << 15)); }
static inline unsigned wire_stashed(uint16_t e) { return ((e >> 10) & 3u) | (((unsigned)e >> 15) << 2); }

static inline int64_t floor16w(int64_t p) { return p & ~(int64_t)15; }
static inline int64_t ceil16w(int64_t p) { return (p + 15) & ~(int64_t)15; }

// ev_off / ev_pos / ev_len (optional; per read of the INPUT order, nc_indel_events' convention: a negative length -L at column c deletes
// columns c + 1 .. c + L): a deleted column's code (4) is implied by the read's own deletion event, which crosses PCIe anyway with the indel
// events -- it is left out of the difference events and written back in HBM by nc_wire_apply_deletions (round 6: 62.8 M of the 120.7 M
// events of a chr20-sized ONT contig).  A code 4 outside every deletion run (a read base N) stays an event.
static int wire_build(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in, const uint8_t *keep,
                      const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, const int32_t *ev_off, const int32_t *ev_pos, const int32_t *ev_len, nc_wire **out,
                      bool want_bytes = false)
{
    if (!out || n_reads < 0 || (n_reads && (!start || !end || !off || !codes_in)) || !ref_wire || ref_len < 0 || (ref_pos0 & 15))
        return NC_ERR_ARG;
    if (ev_off && n_reads && ev_off[n_reads] > 0 && (!ev_pos || !ev_len)) return NC_ERR_ARG;
    *out = nullptr;
    nc_wire *w = new (std::nothrow) nc_wire;
    if (!w) return NC_ERR_NOMEM;
    try {
        std::vector<int64_t> rd_off;
        std::vector<int32_t> orig;                                    // input index of every kept read (its events)
        w->slot_off.push_back(0);
        int32_t prev = INT32_MIN;
        for (int32_t r = 0; r < n_reads; r++) {
            if (keep && !keep[r]) continue;
            if (end[r] <= start[r] || start[r] < prev) { delete w; return NC_ERR_ARG; }      // coordinate order, as nc_pack_plan
            prev = start[r];
            w->rd_start.push_back(start[r]);
            w->rd_end.push_back(end[r]);
            rd_off.push_back(off[r]);
            orig.push_back(r);
            w->slot_off.push_back(w->slot_off.back() + (ceil16w(end[r]) - floor16w(start[r])));
        }
        const int64_t n = (int64_t)w->rd_start.size();
        const int64_t total = w->slot_off[(size_t)n];
        w->codes_len = total + 16;                                                          // nc_pack_plan's convention: never empty
        const int64_t n_blocks = (w->codes_len + WIRE_BLOCK - 1) / WIRE_BLOCK;
        w->blk_off.assign((size_t)n_blocks + 1, 0);
        w->blk_read.assign((size_t)n_blocks, 0);
        std::vector<int64_t> kept_ev_off;                             // running event count of the kept reads (the order the events are uploaded in)
        if (ev_off) {
            w->blk_ev.assign((size_t)n_blocks, 0xffffffffu);
            kept_ev_off.assign((size_t)n + 1, 0);
            for (int64_t q = 0; q < n; q++) kept_ev_off[(size_t)q + 1] = kept_ev_off[(size_t)q] + (ev_off[orig[(size_t)q] + 1] - ev_off[orig[(size_t)q]]);
            if (kept_ev_off[(size_t)n] >= 0xffffffffll) { delete w; return NC_ERR_CAPACITY; }
        }
        int T = std::min(nc_host_cpus(), 32);
        if (n_blocks < 64) T = 1;
        std::vector<std::vector<uint16_t>> part((size_t)T);
        std::vector<std::vector<uint8_t>> bpart((size_t)T);           // want_bytes: the same events one byte each
        std::vector<uint32_t> blk_bytes;
        if (want_bytes) blk_bytes.assign((size_t)n_blocks + 1, 0);
        std::vector<int> status((size_t)T, NC_OK), no_bytes((size_t)T, 0);
        auto work = [&](int t) {
            const int64_t b0 = n_blocks * t / T, b1 = n_blocks * (t + 1) / T;
            std::vector<uint16_t> &ev = part[(size_t)t];
            ev.reserve((size_t)((b1 - b0) * 400));
            int64_t r = std::upper_bound(w->slot_off.begin(), w->slot_off.end(), b0 * WIRE_BLOCK) - w->slot_off.begin() - 1;
            if (r < 0) r = 0;
            for (int64_t b = b0; b < b1; b++) {
                const int64_t byte0 = b * WIRE_BLOCK, byte1 = std::min<int64_t>(byte0 + WIRE_BLOCK, total);
                const size_t before = ev.size();
                while (r < n && w->slot_off[(size_t)r + 1] <= byte0) r++;
                w->blk_read[(size_t)b] = (int32_t)r;                                         // first read reaching into the block (n: none)
                for (int64_t q = r; q < n && w->slot_off[(size_t)q] < byte1; q++) {
                    const int64_t s = w->rd_start[(size_t)q], e = w->rd_end[(size_t)q];
                    const int64_t base = w->slot_off[(size_t)q] - floor16w(s);                // codes[base + p]
                    const int64_t p_lo = std::max<int64_t>(s, byte0 - base), p_hi = std::min<int64_t>(e, byte1 - base);
                    const uint8_t *src = codes_in + rd_off[(size_t)q] - s;                     // src[p]
                    const int64_t off0 = base - byte0;                                         // event offset of position p: off0 + p
                    // this read's deletion runs from p_lo on (events ascend): `covered(p)` for ascending p
                    int32_t de = 0, de1 = 0;
                    // deleted columns are left out only in blocks that lie wholly inside ONE read and inside the reference grid (k_wire_expand's short
                    // path: it applies that read's deletion events from the block's cursor); blocks with a read boundary keep them as events
                    const int64_t pb = floor16w(s) + (byte0 - w->slot_off[(size_t)q]);        // position of the block's first byte in this read's frame
                    const bool single = ev_off && q == r && pb >= s && pb + WIRE_BLOCK <= e && pb - ref_pos0 >= 0 && pb - ref_pos0 + WIRE_BLOCK <= ref_len;
                    if (single) {
                        de = ev_off[orig[(size_t)q]];
                        de1 = ev_off[orig[(size_t)q] + 1];
                        // first event whose run can reach p_lo: runs are short, so the first event with column >= p_lo - 65536 would do; bisect on the column
                        int32_t lo = de, hi = de1;
                        while (lo < hi) {
                            const int32_t mid = (lo + hi) >> 1;
                            if ((int64_t)ev_pos[mid] + (ev_len[mid] < 0 ? -(int64_t)ev_len[mid] : 0) < p_lo) lo = mid + 1; else hi = mid;
                        }
                        de = lo;
                        w->blk_ev[(size_t)b] = (uint32_t)(kept_ev_off[(size_t)q] + (de - ev_off[orig[(size_t)q]]));
                    }
                    auto covered = [&](int64_t p) -> bool {
                        if (!single) return false;
                        while (de < de1 && (ev_len[de] >= 0 || (int64_t)ev_pos[de] - (int64_t)ev_len[de] < p)) de++;
                        return de < de1 && (int64_t)ev_pos[de] < p;              // (de: first deletion whose last column c - len >= p)
                    };
                    // positions off the reference grid (none in a pack built by build_wire: the grid covers every kept read) compare with 'N'
                    const int64_t g_lo = std::min(std::max<int64_t>(p_lo, ref_pos0), p_hi), g_hi = std::max(std::min<int64_t>(p_hi, ref_pos0 + ref_len), g_lo);
                    unsigned worst = 0;
                    for (int64_t p = p_lo; p < g_lo; p++) {
                        const unsigned c = src[p];
                        worst = std::max(worst, c);
                        if (c != 4u) ev.push_back((uint16_t)((off0 + p) | (c << 12) | wire_stash(4)));
                    }
                    // on the grid: 16 positions a step (SSE2 is part of x86-64), the ~8 % that differ leave through the mask's set bits
                    const uint8_t *rf = ref_wire - ref_pos0;                                    // rf[p]
                    int64_t p = g_lo;
                    const __m128i seven = _mm_set1_epi8(7);
                    __m128i vmax = _mm_setzero_si128();
                    for (; p + 16 <= g_hi; p += 16) {
                        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + p));
                        const __m128i r = _mm_and_si128(_mm_loadu_si128(reinterpret_cast<const __m128i *>(rf + p)), seven);
                        vmax = _mm_max_epu8(vmax, c);
                        unsigned m = ~(unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(c, r)) & 0xffffu;
                        while (m) {
                            const int k = __builtin_ctz(m);
                            m &= m - 1;
                            if (src[p + k] == 4u && single && covered(p + k)) continue;
                            ev.push_back((uint16_t)((off0 + p + k) | ((unsigned)src[p + k] << 12) | wire_stash(rf[p + k] & 7u)));
                        }
                    }
                    vmax = _mm_max_epu8(vmax, _mm_srli_si128(vmax, 8));
                    vmax = _mm_max_epu8(vmax, _mm_srli_si128(vmax, 4));
                    vmax = _mm_max_epu8(vmax, _mm_srli_si128(vmax, 2));
                    vmax = _mm_max_epu8(vmax, _mm_srli_si128(vmax, 1));
                    worst = std::max(worst, (unsigned)_mm_cvtsi128_si32(vmax) & 0xffu);         // (a byte >= 7 shows in the maximum)
                    unsigned wmax = worst;
                    for (; p < g_hi; p++) {
                        const unsigned c = src[p];
                        wmax = std::max(wmax, c);
                        if (c != (rf[p] & 7u) && !(c == 4u && single && covered(p))) ev.push_back((uint16_t)((off0 + p) | (c << 12) | wire_stash(rf[p] & 7u)));
                    }
                    for (p = g_hi; p < p_hi; p++) {
                        const unsigned c = src[p];
                        wmax = std::max(wmax, c);
                        if (c != 4u) ev.push_back((uint16_t)((off0 + p) | (c << 12) | wire_stash(4)));
                    }
                    if (wmax >= NC_CODE_ABSENT) { status[(size_t)t] = NC_ERR_ARG; return; }   // codes are 0..6
                }
                w->blk_off[(size_t)b + 1] = (uint32_t)(ev.size() - before);
                // the block's events, in ascending offset, one byte each; the stash bits leave the two-byte form
                std::vector<uint8_t> &bv = bpart[(size_t)t];
                const size_t bbefore = bv.size();
                int prevo = -1;
                for (size_t i = before; i < ev.size(); i++) {
                    const uint16_t x = ev[i];
                    const unsigned pred = wire_stashed(x), c = (x >> 12) & 7u;
                    const int o = x & 0x3ff;
                    ev[i] = (uint16_t)(x & 0x73ffu);
                    if (!want_bytes) continue;
                    if (c > 4u || pred > 4u || o <= prevo) { no_bytes[(size_t)t] = 1; continue; }
                    int gap = o - prevo - 1;
                    while (gap >= 63) { bv.push_back((uint8_t)(63u << 2)); gap -= 63; }
                    bv.push_back((uint8_t)(((unsigned)gap << 2) | wire_which(c, pred)));
                    prevo = o;
                }
                if (want_bytes) blk_bytes[(size_t)b + 1] = (uint32_t)(bv.size() - bbefore);
            }
        };
        if (T == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < T; t++)
            if (status[(size_t)t] != NC_OK) { delete w; return status[(size_t)t]; }
        bool bytes = want_bytes;
        for (int t = 0; t < T; t++) bytes = bytes && !no_bytes[(size_t)t];     // (a code beyond 4: the two-byte form)
        if (bytes) {                                                            // sparse events (HiFi) are mostly fillers: the two-byte form when it is the shorter one
            uint64_t nb = 0, ne = 0;
            for (int t = 0; t < T; t++) { nb += bpart[(size_t)t].size(); ne += part[(size_t)t].size(); }
            bytes = nb < 2 * ne;
        }
        if (bytes) {
            for (int t = 0; t < T; t++) std::vector<uint16_t>().swap(part[(size_t)t]);
            w->blk_off.swap(blk_bytes);
        }
        {
            uint64_t acc = 0;
            for (int64_t b = 0; b < n_blocks; b++) {
                acc += w->blk_off[(size_t)b + 1];
                if (acc > 0xffffffffull) { delete w; return NC_ERR_CAPACITY; }               // 32-bit event offsets: < 4 G events per contig
                w->blk_off[(size_t)b + 1] = (uint32_t)acc;
            }
        }
        if (bytes) w->ev_bytes.resize((size_t)w->blk_off[(size_t)n_blocks] + 8);             // (+ 8: the expansion loads whole dwords)
        else w->events.resize((size_t)w->blk_off[(size_t)n_blocks]);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t]() {
                    const int64_t b0 = n_blocks * t / T;
                    if (bytes) {
                        if (!bpart[(size_t)t].empty()) memcpy(w->ev_bytes.data() + w->blk_off[(size_t)b0], bpart[(size_t)t].data(), bpart[(size_t)t].size());
                    } else if (!part[(size_t)t].empty())
                        memcpy(w->events.data() + w->blk_off[(size_t)b0], part[(size_t)t].data(), part[(size_t)t].size() * sizeof(uint16_t));
                });
            for (auto &x : th) x.join();
        }
    } catch (const std::bad_alloc &) {
        delete w;
        return NC_ERR_NOMEM;
    }
    *out = w;
    return NC_OK;
}

extern "C" int nc_wire_build(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                             const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, nc_wire **out)
{
    return wire_build(n_reads, start, end, off, codes_in, keep, ref_wire, ref_pos0, ref_len, nullptr, nullptr, nullptr, out);
}

static int wire_build_del(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                          const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, const int32_t *ev_off,
                          const int32_t *ev_pos, const int32_t *ev_len, nc_wire **out, bool want_bytes)
{
    if (!ev_off || n_reads < 0 || (n_reads && (!start || !end || !off || !codes_in)) || (n_reads && ev_off[n_reads] > 0 && (!ev_pos || !ev_len))) return NC_ERR_ARG;
    // the events must describe the codes: every deleted column inside its read carries code 4 (what a pileup's '*' decodes to,
    // generate_SNP_pileups.py:104).  Codes that say otherwise (a pack assembled from unrelated arrays) cannot leave them out: NC_ERR_UNSUPPORTED,
    // and the caller builds the plain form
    for (int32_t r = 0; r < n_reads; r++) {
        if (keep && !keep[r]) continue;
        const uint8_t *src = codes_in + off[r] - start[r];
        for (int32_t e = ev_off[r]; e < ev_off[r + 1]; e++) {
            if (ev_len[e] >= 0) continue;
            const int64_t a = std::max<int64_t>((int64_t)ev_pos[e] + 1, start[r]), b = std::min<int64_t>((int64_t)ev_pos[e] + 1 - ev_len[e], end[r]);
            for (int64_t p = a; p < b; p++)
                if (src[p] != 4u) return NC_ERR_UNSUPPORTED;
        }
    }
    return wire_build(n_reads, start, end, off, codes_in, keep, ref_wire, ref_pos0, ref_len, ev_off, ev_pos, ev_len, out, want_bytes);
}

extern "C" int nc_wire_build_del(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                                 const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, const int32_t *ev_off,
                                 const int32_t *ev_pos, const int32_t *ev_len, nc_wire **out)
{
    return wire_build_del(n_reads, start, end, off, codes_in, keep, ref_wire, ref_pos0, ref_len, ev_off, ev_pos, ev_len, out, false);
}

// flags: bit 0 = the events one byte each (nc_wire_arrays.ev_bytes; blk_off then counts bytes); ev_off == NULL: no implied deletions
extern "C" int nc_wire_build2(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                              const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, const int32_t *ev_off,
                              const int32_t *ev_pos, const int32_t *ev_len, int32_t flags, nc_wire **out)
{
    if (ev_off) return wire_build_del(n_reads, start, end, off, codes_in, keep, ref_wire, ref_pos0, ref_len, ev_off, ev_pos, ev_len, out, (flags & 1) != 0);
    return wire_build(n_reads, start, end, off, codes_in, keep, ref_wire, ref_pos0, ref_len, nullptr, nullptr, nullptr, out, (flags & 1) != 0);
}

extern "C" int nc_wire_view(const nc_wire *w, nc_wire_arrays *v)
{
    if (!w || !v) return NC_ERR_ARG;
    v->n_reads = (int32_t)w->rd_start.size();
    v->rd_start = w->rd_start.data();
    v->rd_end = w->rd_end.data();
    v->slot_off = w->slot_off.data();
    v->codes_len = w->codes_len;
    v->n_blocks = (int64_t)w->blk_off.size() - 1;
    v->blk_off = w->blk_off.data();
    v->blk_read = w->blk_read.data();
    v->events = w->events.data();
    v->n_events = (int64_t)w->events.size();
    v->blk_ev = w->blk_ev.empty() ? nullptr : w->blk_ev.data();
    v->ev_bytes = w->ev_bytes.empty() ? nullptr : w->ev_bytes.data();
    v->n_ev_bytes = w->ev_bytes.empty() ? 0 : (int64_t)w->ev_bytes.size() - 8;
    return NC_OK;
}

extern "C" int nc_wire_free(nc_wire *w)
{
    delete w;
    return NC_OK;
}

// ------------------------------------------------------------------------------------------------------------ device side
namespace {
// One WAVE per U consecutive 1024-byte blocks (four waves per workgroup), no workgroup barrier: lane t owns the aligned
// 16-byte group t of each block.  A 16-byte group never straddles two reads (slots are 16-byte aligned) and maps to 16
// consecutive, 16-aligned reference positions, so the predicted bytes are ONE aligned dwordx4 of ref_wire, masked to the read's
// span.  blk_read gives the first read reaching into the block (reads are kilobases long: usually THE read); the block's
// events are scattered over the wave's LDS image (no two events address the same byte), then every lane stores one dwordx4.
// The kernel is a chain of four dependent loads (block table -> read boundaries -> read record -> reference bytes): every
// level is issued for all U blocks before the first result is used, so a wave has U KiB in flight per memory latency.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the 16 bytes of group B for a lane whose block is NOT wholly inside one read (read boundaries, reference edges): the general
// per-lane form.  r = the block's first read.
__device__ __noinline__ uint4 wire_group_general(int64_t B, int64_t r, int32_t n_reads, const int32_t *__restrict__ rd_start,
                                                const int32_t *__restrict__ rd_end, const int64_t *__restrict__ slot_off,
                                                const uint8_t *__restrict__ ref_wire, int32_t ref_pos0, int64_t ref_len)
{
    while (r < n_reads && slot_off[r + 1] <= B) r++;             // at most 63 steps, almost always none or one
    uint32_t o[4] = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
    if (r < n_reads) {
        const int64_t s = rd_start[r], e = rd_end[r];
        const int64_t p0 = (s & ~(int64_t)15) + (B - slot_off[r]);          // position of this group's first byte
        if (p0 < e && p0 + 16 > s) {
            const int64_t ri = p0 - ref_pos0;
            uint32_t w[4] = {0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
            if (ri >= 0 && ri + 16 <= ref_len) {
                const uint4 v = *reinterpret_cast<const uint4 *>(ref_wire + ri);
                w[0] = v.x & 0x07070707u; w[1] = v.y & 0x07070707u; w[2] = v.z & 0x07070707u; w[3] = v.w & 0x07070707u;
            } else {
                for (int j = 0; j < 16; j++) {
                    const int64_t q = ri + j;
                    const uint32_t b = (q >= 0 && q < ref_len) ? (ref_wire[q] & 7u) : 4u;
                    w[j >> 2] = (w[j >> 2] & ~(0xffu << ((j & 3) * 8))) | (b << ((j & 3) * 8));
                }
            }
            // bytes [lo, hi) of the group belong to the read (a read's first and last group are partial): a 16-bit byte mask,
            // each nibble widened to a dword of 0x00 / 0xff bytes (the multiply spreads bit i to bit 8 i, no carries)
            const int64_t dl = s - p0, dh = e - p0;
            const int lo = dl < 0 ? 0 : (int)dl, hi = dh > 16 ? 16 : (int)dh;       // 0 <= lo < 16, 0 < hi <= 16 here
            const uint32_t bits = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint32_t m = (((bits >> (4 * d)) & 0xfu) * 0x00204081u) & 0x01010101u;
                m = (m << 8) - m;
                o[d] = (w[d] & m) | (0x07070707u & ~m);
            }
        }
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// The kernel is issue-bound, not bandwidth-bound (1 KiB per wave pass): everything that is the same for the 64 lanes of a block
// is SCALAR work.  The block's first read and its record are scalar loads; a block that lies wholly inside that read's span and
// inside the reference (8 of 10 blocks: reads are kilobases long) takes the short path -- one dwordx4 of ref_wire per lane from a
// scalar base, masked, into LDS.  Only blocks with a read boundary or a reference edge run the per-lane general form.  U blocks
// per wave: the scalar loads of all of them are issued before the first is used.
// DEL (round 6): the pack travels with its indel events and the builder left the deleted columns of single-read blocks out of the difference events
// (nc_wire_build_del): such a block writes them into its LDS image from the read's own deletion events (absolute ev_pos / ev_len, expanded before this
// kernel), starting at the block's cursor blk_ev -- one coalesced load of 64 events per ~1 KiB block, no second pass over the codes (a separate
// kernel writing 63 M scattered bytes re-reads and re-writes the whole 1.9 GB array: +0.6 ms per chr20-sized pass).
// BYTES (round 6): the events one byte each (nc_wire_build2 flag 1): `events` is that byte stream, blk_off counts bytes.  A lane takes four bytes (one
// dword), the columns they skip are summed across the wave, and an event's code follows from the predicted code already in the image (wire_which).
#ifndef NC_WIRE_UB
#define NC_WIRE_UB 4                  // blocks per wave of the byte-event form
#endif
#ifndef NC_WIRE_EVL
#define NC_WIRE_EVL 2                 // byte events per lane and round of k_wire_expand<.., true>: 4 (round 6's first form), 2 or 1
#endif
template <int U, bool DEL, bool BYTES>
__global__ __launch_bounds__(256) void k_wire_expand(int32_t n_reads, const int32_t *__restrict__ rd_start, const int32_t *__restrict__ rd_end,
                                                     const int64_t *__restrict__ slot_off, const uint8_t *__restrict__ ref_wire,
                                                     int32_t ref_pos0, int64_t ref_len, const uint32_t *__restrict__ blk_off,
                                                     const int32_t *__restrict__ blk_read, const uint16_t *__restrict__ events,
                                                     int64_t n_blocks, uint8_t *__restrict__ codes, int64_t codes_len,
                                                     const uint32_t *__restrict__ blk_ev, const int32_t *__restrict__ ev_off,
                                                     const int32_t *__restrict__ ev_pos, const int32_t *__restrict__ ev_len)
{
    __shared__ __attribute__((aligned(16))) uint8_t img_all[4 * U * WIRE_BLOCK];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t blk0 = ((int64_t)blockIdx.x * 4 + wv) * U;                    // wave-uniform
    if (blk0 >= n_blocks) return;
    uint8_t *img = img_all + wv * (U * WIRE_BLOCK);
    const int nu = (int)(n_blocks - blk0 < U ? n_blocks - blk0 : U);
    // ---- scalar: the blocks' first read, event range, and that read's record
    int32_t r[U];
    uint32_t e0[U], e1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t blk = u < nu ? blk0 + u : blk0;
        r[u] = blk_read[blk];
        e0[u] = blk_off[blk];
        e1[u] = blk_off[blk + 1];
    }
    int64_t ri0[U];                                                  // short path: reference index of the block's first byte, else -1
    uint32_t dcur[U], dend[U];                                       // DEL: the block's cursor into the events, the read's last event + 1
#pragma unroll
    for (int u = 0; u < U; u++) {
        ri0[u] = -1;
        dcur[u] = 0xffffffffu;
        dend[u] = 0;
        if (u < nu && r[u] < n_reads) {
            const int64_t s = rd_start[r[u]], e = rd_end[r[u]], so = slot_off[r[u]];
            const int64_t p0 = (s & ~(int64_t)15) + ((blk0 + u) * WIRE_BLOCK - so);
            const int64_t ri = p0 - ref_pos0;
            if (p0 >= s && p0 + WIRE_BLOCK <= e && ri >= 0 && ri + WIRE_BLOCK <= ref_len) {
                ri0[u] = ri;
                if constexpr (DEL) {
                    dcur[u] = blk_ev[blk0 + u];
                    dend[u] = (uint32_t)ev_off[r[u] + 1];
                }
            }
        }
    }
    // ---- vector: the first 128 events of every block (a PAIR of events per lane: one dword load -- sub-dword global loads run
    // at a fraction of the dword rate and were a third of this kernel), the predicted bytes
    // pair k, k + 1 (k even) as lo | hi << 16, halves outside [e0, e1) set to 0xffff (never an event: bits 10, 11 are clear)
    auto load_pair = [&](uint32_t k, uint32_t ea, uint32_t eb, bool last_block) -> uint32_t {
        if (k >= eb) return 0xffffffffu;
        uint32_t pr;
        if (!last_block || k + 1 < eb) pr = *reinterpret_cast<const uint32_t *>(events + k);
        else pr = (uint32_t)events[k] | 0xffff0000u;                 // the array's last element: no read past its end
        if (k < ea) pr |= 0x0000ffffu;
        if (k + 1 >= eb) pr |= 0xffff0000u;
        return pr;
    };
    const uint8_t *evb = reinterpret_cast<const uint8_t *>(events);
    auto load_quad = [&](uint32_t k, uint32_t eb) -> uint32_t {       // bytes k .. k + 3 (k a multiple of 4; the stream is readable 8 bytes past its end)
        return k < eb ? *reinterpret_cast<const uint32_t *>(evb + k) : 0u;
    };
    uint32_t pr0[U];
    uint4 rv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        if constexpr (BYTES) pr0[u] = (u < nu && lane < 16 * NC_WIRE_EVL) ? load_quad((e0[u] & ~3u) + 4 * lane, e1[u]) : 0u;
        else pr0[u] = u < nu ? load_pair((e0[u] & ~1u) + 2 * lane, e0[u], e1[u], blk0 + u == n_blocks - 1) : 0xffffffffu;
        if (ri0[u] >= 0) rv[u] = *reinterpret_cast<const uint4 *>(ref_wire + ri0[u] + lane * 16);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (u >= nu) continue;
        uint4 o;
        if (ri0[u] >= 0) o = make_uint4(rv[u].x & 0x07070707u, rv[u].y & 0x07070707u, rv[u].z & 0x07070707u, rv[u].w & 0x07070707u);
        else o = wire_group_general((blk0 + u) * WIRE_BLOCK + (int64_t)lane * 16, r[u], n_reads, rd_start, rd_end, slot_off, ref_wire, ref_pos0, ref_len);
        *reinterpret_cast<uint4 *>(img + u * WIRE_BLOCK + lane * 16) = o;
    }
    // LDS operations of one wave execute in program order: the byte stores below land on top of the 16-byte stores above
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto scatter = [&](uint8_t *im, uint32_t pr) {
        const uint32_t a = pr & 0xffffu, b = pr >> 16;
        if (a != 0xffffu) im[a & 0x3ffu] = (uint8_t)(a >> 12);
        if (b != 0xffffu) im[b & 0x3ffu] = (uint8_t)(b >> 12);
    };
    // BYTES: NC_WIRE_EVL events of a lane (a round of the wave = 64 NC_WIRE_EVL bytes, loaded as dwords by its first 16 NC_WIRE_EVL lanes; a block of an
    // ONT contig has ~61 events: with four per lane three lanes in four carried none and the wave still ran four scatter steps; with two a round of
    // 128 bytes covers nearly every block at half the steps); `carry` = columns covered by the block's earlier bytes
    auto scatter4 = [&](uint8_t *im, uint32_t qd, uint32_t kb, uint32_t ea, uint32_t eb, int32_t &carry) {
        constexpr int EVL = NC_WIRE_EVL;
        uint32_t q = qd;
        if constexpr (EVL == 2) q = ((uint32_t)__shfl((int)qd, lane >> 1) >> (16 * (lane & 1))) & 0xffffu;
        else if constexpr (EVL == 1) q = ((uint32_t)__shfl((int)qd, lane >> 2) >> (8 * (lane & 3))) & 0xffu;
        const uint32_t k = kb + EVL * lane;
        int32_t v[EVL], cum = 0;
        uint32_t which[EVL];
        bool isev[EVL];
#pragma unroll
        for (int i = 0; i < EVL; i++) {
            const uint32_t b = (q >> (8 * i)) & 0xffu, g = b >> 2;
            const bool in = k + i >= ea && k + i < eb;
            isev[i] = in && g != 63u;
            which[i] = b & 3u;
            cum += in ? (g == 63u ? 63 : (int32_t)g + 1) : 0;
            v[i] = cum;
        }
        const int32_t incl = nc_wave_incl_scan(cum);
        const int32_t base = carry + incl - cum;
#pragma unroll
        for (int i = 0; i < EVL; i++)
            if (isev[i]) {
                const int32_t o = base + v[i] - 1;
                const uint32_t pred = im[o & 0x3ff];
                im[o & 0x3ff] = (uint8_t)(pred < 4u ? (which[i] == 3u ? 4u : ((pred + 1u + which[i]) & 3u)) : which[i]);
            }
        carry += __builtin_amdgcn_readlane(incl, 63);
    };
#pragma unroll
    for (int u = 0; u < U; u++) {
        uint8_t *im = img + u * WIRE_BLOCK;
        if constexpr (BYTES) {
#ifdef NC_ABL_NOSCATTER
            if (u < 0) {
#else
            if (u < nu) {
#endif
                int32_t carry = 0;
                constexpr uint32_t RB = 64 * NC_WIRE_EVL;                 // bytes a round of the wave takes
                scatter4(im, pr0[u], e0[u] & ~3u, e0[u], e1[u], carry);
                for (uint32_t kb = (e0[u] & ~3u) + RB; kb < e1[u]; kb += RB)
                    scatter4(im, lane < 16 * NC_WIRE_EVL ? load_quad(kb + 4 * lane, e1[u]) : 0u, kb, e0[u], e1[u], carry);
            }
        } else {
            scatter(im, pr0[u]);
            if (u < nu)
                for (uint32_t k = (e0[u] & ~1u) + 128 + 2 * lane; k < e1[u]; k += 128) scatter(im, load_pair(k, e0[u], e1[u], blk0 + u == n_blocks - 1));
        }
    }
    if constexpr (DEL) {
        // the deleted columns of single-read blocks, from the read's deletion events (LDS stores of a wave land in program order: on top of the image,
        // never on a byte a difference event wrote -- a deleted column carries no base)
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (dcur[u] == 0xffffffffu) continue;                    // (wave-uniform)
            uint8_t *im = img + u * WIRE_BLOCK;
            const int32_t pbase = (int32_t)(ri0[u] + ref_pos0);      // position of the block's first byte
            for (uint32_t k0 = dcur[u]; k0 < dend[u]; k0 += 64) {
                const uint32_t k = k0 + lane;
                const bool act = k < dend[u];
                const int32_t pos = act ? ev_pos[k] : INT32_MAX, len = act ? ev_len[k] : 0;
                if (act && len < 0) {
                    const int32_t c0 = pos + 1 - pbase;
                    for (int32_t j = 0; j < -len; j++) {
                        const int32_t c = c0 + j;
                        if (c >= 0 && c < WIRE_BLOCK) im[c] = 4;
                    }
                }
                if (__shfl(pos, 63) >= pbase + WIRE_BLOCK) break;  // events ascend: nothing further reaches into this block
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t B = (blk0 + u) * WIRE_BLOCK + (int64_t)lane * 16;
#ifdef NC_ABL_NOSTORE
        if (u < nu && B + 16 <= codes_len && B == 12345) {
#else
        if (u < nu && B + 16 <= codes_len) {
#endif
            // streaming store: the expanded codes are read once by the scan, later, from HBM
            const u32x4 v = *reinterpret_cast<const u32x4 *>(img + u * WIRE_BLOCK + lane * 16);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(codes + B));
        }
    }
}

// the reference bytes cross PCIe two per byte (round 6: 4 bits say all there is -- base code + skip bit); unpacked into the byte array
// nc_wire_expand reads, 32 positions per thread
__global__ void k_ref_unpack(const uint8_t *__restrict__ nib, uint8_t *__restrict__ ref_wire, int64_t n)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 32;
    if (i + 32 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(nib + i / 2);
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // bytes b0 b1 b2 b3, each lo | hi << 4 -> positions b0.lo b0.hi b1.lo b1.hi | b2.lo b2.hi b3.lo b3.hi
            const uint32_t x = in[q], lo = x & 0x0f0f0f0fu, hi = (x >> 4) & 0x0f0f0f0fu;
            o[2 * q] = (lo & 0xffu) | ((hi & 0xffu) << 8) | ((lo & 0xff00u) << 8) | ((hi & 0xff00u) << 16);
            o[2 * q + 1] = ((lo >> 16) & 0xffu) | (((hi >> 16) & 0xffu) << 8) | (((lo >> 24) & 0xffu) << 16) | (((hi >> 24) & 0xffu) << 24);
        }
        *reinterpret_cast<uint4 *>(ref_wire + i) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(ref_wire + i + 16) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (int64_t j = i; j < n; j++) ref_wire[j] = (nib[j >> 1] >> ((j & 1) * 4)) & 0xfu;
    }
}

// reference codes of the column scan from the wire form: skipped columns (bit 3: soft-masked, non-AGTC, excluded) become 4
__global__ void k_ref_from_wire(const uint8_t *__restrict__ ref_wire, uint8_t *__restrict__ ref_code, int64_t n)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ref_wire + i);
        auto f = [](uint32_t x) {
            const uint32_t m = ((x >> 3) & 0x01010101u) * 0xffu;           // 0xff in the bytes whose skip bit is set
            return ((x & 0x07070707u) & ~m) | (0x04040404u & m);
        };
        *reinterpret_cast<uint4 *>(ref_code + i) = make_uint4(f(v.x), f(v.y), f(v.z), f(v.w));
    } else {
        for (int64_t j = i; j < n; j++) ref_code[j] = (ref_wire[j] & 8) ? 4 : (ref_wire[j] & 7);
    }
}

}   // namespace

static int wire_expand(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                       const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                       const int32_t *d_blk_read, const uint16_t *d_events, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                       uint8_t *d_ref_code, const uint32_t *d_blk_ev, const int32_t *d_ev_off, const int32_t *d_ev_pos, const int32_t *d_ev_len, bool bytes = false)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_reads < 0 || !d_slot_off || (n_reads && (!d_rd_start || !d_rd_end)) || !d_ref_wire || (ref_pos0 & 15) || ref_len < 0 ||
        !d_blk_off || !d_blk_read || n_blocks < 1 || !d_codes || codes_len < 16 || (codes_len & 15) || n_blocks != (codes_len + WIRE_BLOCK - 1) / WIRE_BLOCK ||
        (n_blocks + 3) / 4 > INT32_MAX || ((uintptr_t)d_codes & 15) || ((uintptr_t)d_events & 3) || ((uintptr_t)d_ref_wire & 15) || (d_ref_code && ((uintptr_t)d_ref_code & 15)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_wire_expand: bad argument");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    static const int U = [] { const char *e = getenv("NC_WIRE_U"); const int v = e ? atoi(e) : NC_WIRE_U; return (v == 1 || v == 2 || v == 8) ? v : 4; }();
#define NC_LAUNCH_EXPAND(UU, DD)                                                                                                                         \
    hipLaunchKernelGGL((k_wire_expand<UU, DD, false>), dim3((unsigned)((n_blocks + 4 * UU - 1) / (4 * UU))), dim3(256), 0, ctx->stream, n_reads, d_rd_start, \
                       d_rd_end, d_slot_off, d_ref_wire, ref_pos0, ref_len, d_blk_off, d_blk_read, d_events, n_blocks, d_codes, codes_len, d_blk_ev,         \
                       d_ev_off, d_ev_pos, d_ev_len)
#define NC_LAUNCH_EXPAND_B(DD)                                                                                                                          \
    hipLaunchKernelGGL((k_wire_expand<NC_WIRE_UB, DD, true>), dim3((unsigned)((n_blocks + 4 * NC_WIRE_UB - 1) / (4 * NC_WIRE_UB))), dim3(256), 0, ctx->stream, n_reads, d_rd_start, d_rd_end,  \
                       d_slot_off, d_ref_wire, ref_pos0, ref_len, d_blk_off, d_blk_read, d_events, n_blocks, d_codes, codes_len, d_blk_ev, d_ev_off,    \
                       d_ev_pos, d_ev_len)
    if (bytes) { if (d_blk_ev) NC_LAUNCH_EXPAND_B(true); else NC_LAUNCH_EXPAND_B(false); }
    else if (d_blk_ev) { if (U == 1) NC_LAUNCH_EXPAND(1, true); else if (U == 2) NC_LAUNCH_EXPAND(2, true); else if (U == 8) NC_LAUNCH_EXPAND(8, true); else NC_LAUNCH_EXPAND(4, true); }
    else { if (U == 1) NC_LAUNCH_EXPAND(1, false); else if (U == 2) NC_LAUNCH_EXPAND(2, false); else if (U == 8) NC_LAUNCH_EXPAND(8, false); else NC_LAUNCH_EXPAND(4, false); }
#undef NC_LAUNCH_EXPAND
#undef NC_LAUNCH_EXPAND_B
    if (d_ref_code && ref_len) {
        const int64_t groups = (ref_len + 15) / 16;
        hipLaunchKernelGGL(k_ref_from_wire, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, ctx->stream, d_ref_wire, d_ref_code, ref_len);
    }
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

// ---------------------------------------------------------------------------------------------------- indel events, transfer form
// The indel path's per-read events ('+n' / '-n' of the pileup: column, signed length, offset of the inserted bases) are 12 bytes each as the
// kernels read them -- 830 MB of a chr20-sized ONT contig's 1.2 GB transfer, more than its 21 ms pass can hide.  They cross PCIe as 3 bytes:
//     d16  uint16  column - column of the read's previous event (the first one: - the read's start); 0xFFFF = see the side table
//     l8   int8    signed length; the side table's entry when d16 is 0xFFFF.  l8 == NULL (round 6): the TWO-byte form, d16 = distance (bits 0-10,
//                  0x7ff -> side table) | signed length << 11 (-16 .. 15)
//     side table (big_idx ascending, big_pos, big_len): events with a distance >= 0xFFFF or |length| >= 128
//     read_ins_off [n_reads + 1]: offset of the read's first inserted base (the per-event offsets are its running sum of the positive lengths)
// nc_indel_events_pack makes them on the host; nc_indel_events_expand rebuilds ev_pos / ev_len / ins_off in HBM, one wave per read.
extern "C" int nc_indel_events_pack(int32_t n_reads, const int32_t *rd_start, const int32_t *ev_off, const int32_t *ev_pos, const int32_t *ev_len,
                                    uint16_t *d16, int8_t *l8, int32_t *read_ins_off, int64_t big_cap, int32_t *big_idx, int32_t *big_pos,
                                    int32_t *big_len, int64_t *n_big)
{
    if (n_reads < 0 || (n_reads && (!rd_start || !ev_off)) || !n_big) return NC_ERR_ARG;
    int64_t nb = 0, ins = 0;
    for (int32_t r = 0; r < n_reads; r++) {
        int32_t prev = rd_start[r];
        if (read_ins_off) read_ins_off[r] = (int32_t)ins;
        for (int32_t e = ev_off[r]; e < ev_off[r + 1]; e++) {
            const int32_t d = ev_pos[e] - prev, l = ev_len[e];
            if (d < 0) return NC_ERR_ARG;                            // events of a read ascend
            if (!l8) {
                // two-byte form (round 6): distance in bits 0-10 (0x7ff = side table), signed length in bits 11-15 (-16 .. 15)
                if (d >= 0x7ff || l > 15 || l < -16) {
                    if (nb < big_cap) { big_idx[nb] = e; big_pos[nb] = ev_pos[e]; big_len[nb] = l; }
                    nb++;
                    d16[e] = 0xFFFF;
                } else
                    d16[e] = (uint16_t)((unsigned)d | (((unsigned)l & 0x1fu) << 11));
            } else if (d >= 0xFFFF || l >= 128 || l <= -128) {
                if (nb < big_cap) { big_idx[nb] = e; big_pos[nb] = ev_pos[e]; big_len[nb] = l; }
                nb++;
                d16[e] = 0xFFFF;
                l8[e] = 0;
            } else {
                d16[e] = (uint16_t)d;
                l8[e] = (int8_t)l;
            }
            prev = ev_pos[e];
            if (l > 0) ins += l;
        }
    }
    if (read_ins_off) read_ins_off[n_reads] = (int32_t)ins;
    *n_big = nb;
    return nb > big_cap ? NC_ERR_CAPACITY : NC_OK;
}

// One byte per event (round 6): distance to the read's previous event in bits 2-7 (0 .. 62), length code in bits 0-1 (+1, -1, +2, -2) -- 86 % of an ONT
// read's events; 0xFF = the event is the next entry of the two-byte array d16x (the form above, its 0xFFFF pointing on into the side table), whose first
// entry per read is read_esc_off[r].  69 M events of a chr20-sized contig: 138 -> 88 MB.
extern "C" int nc_indel_events_pack8(int32_t n_reads, const int32_t *rd_start, const int32_t *ev_off, const int32_t *ev_pos, const int32_t *ev_len,
                                     uint8_t *b8, uint16_t *d16x, int64_t esc_cap, int32_t *read_esc_off, int32_t *read_ins_off, int64_t big_cap,
                                     int32_t *big_idx, int32_t *big_pos, int32_t *big_len, int64_t *n_esc, int64_t *n_big)
{
    if (n_reads < 0 || (n_reads && (!rd_start || !ev_off || !read_esc_off)) || !n_big || !n_esc) return NC_ERR_ARG;
    int64_t nb = 0, ne = 0, ins = 0;
    for (int32_t r = 0; r < n_reads; r++) {
        int32_t prev = rd_start[r];
        if (read_ins_off) read_ins_off[r] = (int32_t)ins;
        read_esc_off[r] = (int32_t)ne;
        for (int32_t e = ev_off[r]; e < ev_off[r + 1]; e++) {
            const int32_t d = ev_pos[e] - prev, l = ev_len[e];
            if (d < 0) return NC_ERR_ARG;                            // events of a read ascend
            const int code = l == 1 ? 0 : l == -1 ? 1 : l == 2 ? 2 : l == -2 ? 3 : -1;
            if (d <= 62 && code >= 0) b8[e] = (uint8_t)((d << 2) | code);
            else {
                b8[e] = 0xFF;
                uint16_t w;
                if (d >= 0x7ff || l > 15 || l < -16) {
                    if (nb < big_cap) { big_idx[nb] = e; big_pos[nb] = ev_pos[e]; big_len[nb] = l; }
                    nb++;
                    w = 0xFFFF;
                } else
                    w = (uint16_t)((unsigned)d | (((unsigned)l & 0x1fu) << 11));
                if (ne < esc_cap) d16x[ne] = w;
                ne++;
            }
            prev = ev_pos[e];
            if (l > 0) ins += l;
        }
    }
    read_esc_off[n_reads] = (int32_t)ne;
    if (read_ins_off) read_ins_off[n_reads] = (int32_t)ins;
    *n_big = nb;
    *n_esc = ne;
    return (nb > big_cap || ne > esc_cap || ne > INT32_MAX) ? NC_ERR_CAPACITY : NC_OK;
}

namespace {
__device__ __forceinline__ int32_t wave_incl_scan(int32_t v, int lane)
{
    (void)lane;
    return nc_wave_incl_scan(v);
}

// B8: the one-byte form (nc_indel_events_pack8): b8 per event, d16 = the two-byte array of the escapes, read_esc_off their start per read
template <bool B8>
__global__ __launch_bounds__(256) void k_events_expand(int32_t n_reads, const int32_t *__restrict__ rd_start, const int32_t *__restrict__ ev_off,
                                                       const uint16_t *__restrict__ d16, const int8_t *__restrict__ l8, int32_t n_big,
                                                       const int32_t *__restrict__ big_idx, const int32_t *__restrict__ big_pos,
                                                       const int32_t *__restrict__ big_len, const int32_t *__restrict__ read_ins_off,
                                                       int32_t *__restrict__ ev_pos, int32_t *__restrict__ ev_len, int32_t *__restrict__ ins_off,
                                                       const uint8_t *__restrict__ b8, const int32_t *__restrict__ read_esc_off)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const int e0 = ev_off[r], e1 = ev_off[r + 1];
    int32_t run_pos = rd_start[r], run_ins = read_ins_off ? read_ins_off[r] : 0;
    int32_t run_esc = 0;
    if constexpr (B8) run_esc = read_esc_off[r];
    for (int c = e0; c < e1; c += 64) {
        const int e = c + lane;
        const bool valid = e < e1;
        int32_t d = 0, l = 0, ab = 0;
        bool two = !B8;                                                  // this event is in the two-byte form
        if constexpr (B8) {
            const uint32_t b = valid ? (uint32_t)b8[e] : 0u;
            const bool esc = valid && b == 0xFFu;
            const unsigned long long em = __ballot(esc);
            if (esc) { d = (int32_t)d16[run_esc + __popcll(em & ((1ull << lane) - 1ull))]; two = true; }
            else { d = (int32_t)(b >> 2); l = ((b & 2u) ? 2 : 1) * ((b & 1u) ? -1 : 1); if (!valid) l = 0; }
            run_esc += __popcll(em);
        } else {
            d = valid ? (int32_t)d16[e] : 0;
            l = (valid && l8) ? (int32_t)l8[e] : 0;
        }
        const bool big = valid && two && d == 0xFFFF;
        if (!l8 && valid && two && !big) {                            // two-byte form: distance | signed 5-bit length << 11
            l = (int32_t)((uint32_t)d << 16) >> 27;
            d &= 0x7ff;
        }
        if (big) {
            int lo = 0, hi = n_big;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (big_idx[mid] < e) lo = mid + 1; else hi = mid;
            }
            ab = big_pos[lo];
            l = big_len[lo];
            d = 0;
        }
        const int32_t ip = l > 0 ? l : 0;
        const int32_t P = wave_incl_scan(d, lane), Q = wave_incl_scan(ip, lane);
        const unsigned long long bm = __ballot(big), mine = bm & ((lane == 63 ? 0ull : (1ull << (lane + 1))) - 1ull);
        int32_t pos = run_pos + P;
        const int lb = mine ? 63 - __builtin_clzll(mine) : 0;          // the last side-table event at or before this lane restarts the sum
        const int32_t ab_l = __shfl(ab, lb), P_l = __shfl(P, lb);
        if (mine) pos = ab_l + (P - P_l);
        if (valid) {
            ev_pos[e] = pos;
            ev_len[e] = l;
            if (ins_off) ins_off[e] = run_ins + Q - ip;
        }
        run_pos = __shfl(pos, 63);
        run_ins += __shfl(Q, 63);
    }
    if (ins_off && r == n_reads - 1 && lane == 0) ins_off[e1] = run_ins;
}
}   // namespace

extern "C" int nc_wire_expand(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                              const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                              const int32_t *d_blk_read, const uint16_t *d_events, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                              uint8_t *d_ref_code)
{
    return wire_expand(ctx, n_reads, d_rd_start, d_rd_end, d_slot_off, d_ref_wire, ref_pos0, ref_len, d_blk_off, d_blk_read, d_events, n_blocks, d_codes, codes_len,
                       d_ref_code, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int nc_wire_expand_del(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                                  const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                                  const int32_t *d_blk_read, const uint16_t *d_events, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                                  uint8_t *d_ref_code, const uint32_t *d_blk_ev, const int32_t *d_ev_off, const int32_t *d_ev_pos, const int32_t *d_ev_len)
{
    if (!ctx) return NC_ERR_ARG;
    if (!d_blk_ev || !d_ev_off || (n_reads && (!d_ev_pos || !d_ev_len))) return nc_fail(ctx, NC_ERR_ARG, "nc_wire_expand_del: bad argument");
    return wire_expand(ctx, n_reads, d_rd_start, d_rd_end, d_slot_off, d_ref_wire, ref_pos0, ref_len, d_blk_off, d_blk_read, d_events, n_blocks, d_codes, codes_len,
                       d_ref_code, d_blk_ev, d_ev_off, d_ev_pos, d_ev_len);
}

// Inserted bases as they cross PCIe since round 6 (the indel caller's pack): two bits a base (A0 G1 T2 C3, base i in bits 2 (i & 3) of byte i >> 2);
// the few other letters (code 4) travel as a list of indices.  nc_wire_ins_unpack rebuilds the byte array the kernels read.
namespace {
__global__ __launch_bounds__(256) void k_ins_unpack(const uint8_t *__restrict__ packed, int64_t n, uint8_t *__restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;                       // 16 bases = one dword in, one dwordx4 out
    if (i >= n) return;
    const uint32_t w = *reinterpret_cast<const uint32_t *>(packed + (i >> 2));
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t b = (w >> (8 * q)) & 0xffu;
        o[q] = (b & 3u) | ((b >> 2) & 3u) << 8 | ((b >> 4) & 3u) << 16 | ((b >> 6) & 3u) << 24;
    }
    *reinterpret_cast<uint4 *>(out + i) = make_uint4(o[0], o[1], o[2], o[3]);                 // (the buffers are padded to 16 bases)
}
__global__ __launch_bounds__(256) void k_ins_others(const int32_t *__restrict__ idx, int32_t n, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[idx[i]] = 4;
}
}   // namespace

extern "C" int nc_wire_ins_unpack(nc_ctx *ctx, const uint8_t *d_packed, int64_t n_bases, const int32_t *d_other_idx, int32_t n_other, uint8_t *d_ins_bases)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_bases < 0 || n_other < 0 || (n_bases && (!d_packed || !d_ins_bases)) || (n_other && !d_other_idx) || ((uintptr_t)d_packed & 3) || ((uintptr_t)d_ins_bases & 15))
        return nc_fail(ctx, NC_ERR_ARG, "nc_wire_ins_unpack: bad argument");
    if (n_bases == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t groups = (n_bases + 15) / 16;
    hipLaunchKernelGGL(k_ins_unpack, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, ctx->stream, d_packed, n_bases, d_ins_bases);
    if (n_other) hipLaunchKernelGGL(k_ins_others, dim3((unsigned)((n_other + 255) / 256)), dim3(256), 0, ctx->stream, d_other_idx, n_other, d_ins_bases);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

// the same with the events one byte each (nc_wire_build2, flag 1): d_ev_bytes (readable 8 bytes past its end), d_blk_off counting bytes; d_blk_ev etc. NULL
// when the pack does not leave deleted columns out
extern "C" int nc_wire_expand2(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                               const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                               const int32_t *d_blk_read, const uint8_t *d_ev_bytes, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                               uint8_t *d_ref_code, const uint32_t *d_blk_ev, const int32_t *d_ev_off, const int32_t *d_ev_pos, const int32_t *d_ev_len)
{
    if (!ctx) return NC_ERR_ARG;
    if (!d_ev_bytes || (d_blk_ev && (!d_ev_off || (n_reads && (!d_ev_pos || !d_ev_len))))) return nc_fail(ctx, NC_ERR_ARG, "nc_wire_expand2: bad argument");
    return wire_expand(ctx, n_reads, d_rd_start, d_rd_end, d_slot_off, d_ref_wire, ref_pos0, ref_len, d_blk_off, d_blk_read, reinterpret_cast<const uint16_t *>(d_ev_bytes),
                       n_blocks, d_codes, codes_len, d_ref_code, d_blk_ev, d_ev_off, d_ev_pos, d_ev_len, true);
}

extern "C" int nc_wire_ref_unpack(nc_ctx *ctx, const uint8_t *d_ref_nib, int64_t ref_len, uint8_t *d_ref_wire)
{
    if (!ctx) return NC_ERR_ARG;
    if (ref_len < 0 || (ref_len && (!d_ref_nib || !d_ref_wire))) return nc_fail(ctx, NC_ERR_ARG, "nc_wire_ref_unpack: bad argument");
    if (ref_len == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nt = (ref_len + 31) / 32;
    hipLaunchKernelGGL(k_ref_unpack, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, ctx->stream, d_ref_nib, d_ref_wire, ref_len);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

namespace {
// the deleted columns of every read written back into the expanded codes (nc_wire_build_del left them out of the difference events): one wave per
// read, a lane per event, a byte store per deleted column (runs are 1-2 columns long)
__global__ __launch_bounds__(256) void k_apply_deletions(int32_t n_reads, const int32_t *__restrict__ rd_start, const int32_t *__restrict__ rd_end,
                                                         const int64_t *__restrict__ slot_off, const int32_t *__restrict__ ev_off, const int32_t *__restrict__ ev_pos,
                                                         const int32_t *__restrict__ ev_len, uint8_t *__restrict__ codes)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const int32_t s = rd_start[r], e_ = rd_end[r];
    uint8_t *c = codes + (slot_off[r] - (int64_t)(s & ~15));             // c[p] = the read's code at column p
    for (int e = ev_off[r] + lane; e < ev_off[r + 1]; e += 64) {
        const int32_t l = ev_len[e];
        if (l >= 0) continue;
        const int32_t p0 = ev_pos[e] + 1, p1 = min(p0 - l, e_);
        for (int32_t p = max(p0, s); p < p1; p++) c[p] = 4;
    }
}
}   // namespace

extern "C" int nc_wire_apply_deletions(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                                       const int32_t *d_ev_off, const int32_t *d_ev_pos, const int32_t *d_ev_len, uint8_t *d_codes)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_reads < 0 || (n_reads && (!d_rd_start || !d_rd_end || !d_slot_off || !d_ev_off || !d_ev_pos || !d_ev_len || !d_codes)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_wire_apply_deletions: bad argument");
    if (n_reads == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_apply_deletions, dim3((n_reads + 3) / 4), dim3(256), 0, ctx->stream, n_reads, d_rd_start, d_rd_end, d_slot_off, d_ev_off, d_ev_pos, d_ev_len,
                       d_codes);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

extern "C" int nc_indel_events_expand(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_ev_off, const uint16_t *d_d16,
                                      const int8_t *d_l8, int32_t n_big, const int32_t *d_big_idx, const int32_t *d_big_pos,
                                      const int32_t *d_big_len, const int32_t *d_read_ins_off, int32_t *d_ev_pos, int32_t *d_ev_len,
                                      int32_t *d_ins_off)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_reads < 0 || n_big < 0 || (n_reads && (!d_rd_start || !d_ev_off || !d_d16 || !d_ev_pos || !d_ev_len)) ||
        (n_big && (!d_big_idx || !d_big_pos || !d_big_len)) || (d_ins_off && !d_read_ins_off))
        return nc_fail(ctx, NC_ERR_ARG, "nc_indel_events_expand: bad argument");
    if (n_reads == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_events_expand<false>, dim3((n_reads + 3) / 4), dim3(256), 0, ctx->stream, n_reads, d_rd_start, d_ev_off, d_d16, d_l8, n_big, d_big_idx,
                       d_big_pos, d_big_len, d_read_ins_off, d_ev_pos, d_ev_len, d_ins_off, nullptr, nullptr);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

extern "C" int nc_indel_events_expand8(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_ev_off, const uint8_t *d_b8,
                                       const uint16_t *d_d16x, const int32_t *d_read_esc_off, int32_t n_big, const int32_t *d_big_idx,
                                       const int32_t *d_big_pos, const int32_t *d_big_len, const int32_t *d_read_ins_off, int32_t *d_ev_pos,
                                       int32_t *d_ev_len, int32_t *d_ins_off)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_reads < 0 || n_big < 0 || (n_reads && (!d_rd_start || !d_ev_off || !d_b8 || !d_d16x || !d_read_esc_off || !d_ev_pos || !d_ev_len)) ||
        (n_big && (!d_big_idx || !d_big_pos || !d_big_len)) || (d_ins_off && !d_read_ins_off))
        return nc_fail(ctx, NC_ERR_ARG, "nc_indel_events_expand8: bad argument");
    if (n_reads == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_events_expand<true>, dim3((n_reads + 3) / 4), dim3(256), 0, ctx->stream, n_reads, d_rd_start, d_ev_off, d_d16x, nullptr, n_big, d_big_idx,
                       d_big_pos, d_big_len, d_read_ins_off, d_ev_pos, d_ev_len, d_ins_off, d_b8, d_read_esc_off);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}
