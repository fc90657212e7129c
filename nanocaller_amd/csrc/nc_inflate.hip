// DEFLATE (RFC 1951) on the device: the payloads of BGZF members (SAMv1 4.1: independent raw-deflate streams of at most 64 KB of data each)
// are inflated in HBM, ONE LANE PER MEMBER -- a BAM of a chr20-sized 30x contig is ~30,000 members, so the launch is as wide as the file is
// long and needs no cooperation between lanes; its duration is the time ONE lane takes for ONE member, whatever the file's size.  This
// replaces the host's inflate (libdeflate / zlib on worker threads: 73 % of the ingest of generate_SNP_pileups.py:134-164's input,
// DESIGN.md section 8) on the way from a BAM file to the read pack.
//
// Per lane: a 64-bit bit buffer refilled from aligned dwords; Huffman decoding through per-lane first-level tables in LDS (literal / length:
// 9 bits, distance: 8 bits; entry = symbol << 4 | code length), codes longer than the table's index by the canonical count / symbol walk;
// a lane is either decoding a symbol or copying up to eight bytes of a match per turn of the loop, so a wave never waits for its longest
// match.  Stored, fixed and dynamic blocks.  Checked: the stream ends exactly at ISIZE bytes, distances stay inside the output, the input
// is not overrun.  The CRC-32 of a member is the host's to check (nc_bam.cpp, when the bytes come back) -- the device path checks the lengths.
#include "nc_common.h"

namespace {

constexpr int LT_BITS = 9, DT_BITS = 6, LT_SZ = 1 << LT_BITS, DT_SZ = 1 << DT_BITS;
// one lane's LDS, in uint16 units: the two first-level tables, the canonical arrays of both codes (count[16] + sym[]), and the code lengths of
// the block being set up (bytes; reused as scratch).  A wave's 64 lanes walk these at different places all the time: in scratch memory (HBM
// latency per access, nothing to hide it behind) the set-up loops and the long-code walk were most of the kernel's 260 ms.
constexpr int L_LT = 0, L_DT = L_LT + LT_SZ, L_HL = L_DT + DT_SZ, L_HD = L_HL + 16 + 288, L_LENS = L_HD + 16 + 32, L_END = L_LENS + 320 / 2;
constexpr int TAB_WORDS = (L_END + 1) / 2 | 1;                      // odd pitch in words: lanes spread over the banks

struct InflateArgs {
    const uint8_t *comp;        // compressed payloads (the buffer is readable 8 bytes past the last payload)
    const int64_t *coff;        // byte offset of member b's deflate payload in comp
    const int32_t *clen;        // its length
    uint8_t *out;
    const int64_t *ooff;        // where member b's data goes
    const int32_t *isize;       // the length its trailer announces
    int32_t n;
    int32_t *status;            // 0 ok; 1 bad block type / stored length; 2 bad code lengths; 3 bad symbol / distance; 4 output overrun; 5 input
                                // overrun; 6 length differs from ISIZE
};

struct __attribute__((packed, aligned(4))) U4w { uint32_t x, y, z, w; };   // four dwords at a 4-byte aligned address

// canonical code of up to 288 symbols (RFC 1951 3.2.2) for the walk over the lengths: h[0..16) = count per length, h[16..) = symbols by code
// lens[0..n) -> h; returns false for an over-subscribed set (an incomplete one is accepted: single-symbol distance codes are legal)
__device__ bool huff_build(uint16_t *h, const uint8_t *lens, int n)
{
    for (int i = 0; i < 16; i++) h[i] = 0;
    for (int i = 0; i < n; i++) h[lens[i]]++;
    uint16_t offs[16];                                                 // first slot of every length (registers: every index below is a compile-time one)
    offs[0] = 0; offs[1] = 0;
#pragma unroll
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h[l]);
    int bad = 0, run = 1;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        run = (run << 1) - h[l];
        bad |= run < 0;
    }
    if (bad) return false;
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (!l) continue;
        uint16_t o = 0;
#pragma unroll
        for (int q = 1; q < 16; q++) o = l == q ? offs[q] : o;
#pragma unroll
        for (int q = 1; q < 16; q++) offs[q] = (uint16_t)(l == q ? offs[q] + 1 : offs[q]);
        h[16 + o] = (uint16_t)i;
    }
    return true;
}

// first-level table: index = the next `bits` bits of the stream (first bit lowest) -> symbol << 4 | length, 0 where the code is longer
__device__ void table_fill(uint16_t *tab, int bits, const uint16_t *h, const uint8_t *lens, int n)
{
    for (int i = 0; i < (1 << bits); i++) tab[i] = 0;
    uint32_t next[16];                                                 // first code of every length (h[0], the unused symbols, takes none)
    uint32_t code = 0;
    next[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        next[l] = code;
        code = (code + h[l]) << 1;
    }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t c = 0;
#pragma unroll
        for (int q = 1; q < 16; q++) c = l == q ? next[q] : c;
#pragma unroll
        for (int q = 1; q < 16; q++) next[q] = l == q ? next[q] + 1 : next[q];
        if (l > bits) continue;
        const uint32_t r = __brev(c) >> (32 - l);
        for (uint32_t k = r; k < (1u << bits); k += 1u << l) tab[k] = (uint16_t)(s << 4 | l);
    }
}

__global__ __launch_bounds__(64) void k_inflate(InflateArgs a)
{
    extern __shared__ uint32_t tabs[];
    const int lane = threadIdx.x, b = blockIdx.x * 64 + lane;
    if (b >= a.n) return;
    uint16_t *lbase = reinterpret_cast<uint16_t *>(tabs + lane * TAB_WORDS);
    uint16_t *lt = lbase + L_LT, *dt = lbase + L_DT, *hl = lbase + L_HL, *hd = lbase + L_HD;
    uint8_t *lens = reinterpret_cast<uint8_t *>(lbase + L_LENS);
    const int64_t c0 = a.coff[b];
    const int32_t clen = a.clen[b], isize = a.isize[b];
    uint8_t *out = a.out + a.ooff[b];
    // bit reader over aligned dwords, fetched FOUR at a time and one fetch ahead: a lane has nothing else to hide a load's latency behind (a member
    // is one lane's serial work), so the next 16 bytes are requested while the current ones are decoded
    const int skew = (int)(c0 & 3);
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(a.comp + (c0 & ~int64_t(3)));
    const int64_t w_end = (skew + clen + 3) / 4 + 2;                   // dwords that may be read (two of slack: the buffer is padded)
    auto fetch4 = [&](int64_t w) -> U4w {                              // dwords w .. w + 3 (zeros beyond the member's end)
        U4w v = {0u, 0u, 0u, 0u};
        if (w + 3 < w_end) v = *reinterpret_cast<const U4w *>(wp + w);
        else {
            if (w < w_end) v.x = wp[w];
            if (w + 1 < w_end) v.y = wp[w + 1];
            if (w + 2 < w_end) v.z = wp[w + 2];
        }
        return v;
    };
    U4w cur = fetch4(0), nxt = fetch4(4);
    int64_t wi = 1;                                                    // dwords consumed
    uint64_t bb = (uint64_t)(cur.x >> (8 * skew));
    int bc = 32 - 8 * skew;
    int err = 0;
    auto refill = [&]() {
        if (bc <= 32) {
            const int k = (int)(wi & 3);
            if (k == 0) {                                              // the queue turns over: the dwords fetched a while ago become current
                cur = nxt;
                nxt = fetch4(wi + 4);
            }
            const uint32_t w = k == 0 ? cur.x : k == 1 ? cur.y : k == 2 ? cur.z : cur.w;
            bb |= (uint64_t)w << bc;
            if (wi > w_end + 1) err = 5;
            wi++;
            bc += 32;
        }
    };
    auto take = [&](int n) -> uint32_t {                               // n <= 16 bits (the caller refilled)
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        bc -= n;
        return v;
    };
    auto slow = [&](const uint16_t *h) -> int {                        // RFC 1951 decoding, one bit at a time (codes longer than a table's index)
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)take(1);
            const int cnt = h[l];
            if (code - cnt < first) return h[16 + index + (code - first)];
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    };
    int op = 0;
    int copy_left = 0, copy_dist = 0;
    bool last = false, in_block = false, stored = false;
    int stored_left = 0;
    while (!err) {
        if (!in_block) {
            if (last) break;
            refill();
            last = take(1) != 0;
            const int type = (int)take(2);
            if (type == 0) {
                // stored: skip to the byte boundary, LEN, NLEN
                take(bc & 7);
                refill();
                const uint32_t len = take(16);
                refill();
                const uint32_t nlen = take(16);
                if ((len ^ 0xffffu) != nlen) { err = 1; break; }
                stored = true;
                stored_left = (int)len;
                in_block = true;
                if (op + stored_left > isize) { err = 4; break; }
            } else if (type == 1 || type == 2) {
                int nlen = 288, ndist = 30;
                if (type == 1) {
                    for (int i = 0; i < 144; i++) lens[i] = 8;
                    for (int i = 144; i < 256; i++) lens[i] = 9;
                    for (int i = 256; i < 280; i++) lens[i] = 7;
                    for (int i = 280; i < 288; i++) lens[i] = 8;
                    for (int i = 0; i < 30; i++) lens[288 + i] = 5;
                } else {
                    refill();
                    nlen = (int)take(5) + 257;
                    ndist = (int)take(5) + 1;
                    const int ncode = (int)take(4) + 4;
                    if (nlen > 286 || ndist > 30) { err = 2; break; }
                    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t cl[19];
                    for (int i = 0; i < 19; i++) cl[i] = 0;
                    for (int i = 0; i < ncode; i++) {
                        refill();
                        cl[order[i]] = (uint8_t)take(3);
                    }
                    if (!huff_build(hl, cl, 19)) { err = 2; break; }
                    int idx = 0;
                    while (idx < nlen + ndist && !err) {
                        refill();
                        int sym = slow(hl);
                        if (sym < 0) { err = 2; break; }
                        if (sym < 16) lens[idx++] = (uint8_t)sym;
                        else {
                            int prev = 0, rep;
                            refill();
                            if (sym == 16) {
                                if (idx == 0) { err = 2; break; }
                                prev = lens[idx - 1];
                                rep = 3 + (int)take(2);
                            } else if (sym == 17) rep = 3 + (int)take(3);
                            else rep = 11 + (int)take(7);
                            if (idx + rep > nlen + ndist) { err = 2; break; }
                            while (rep--) lens[idx++] = (uint8_t)prev;
                        }
                    }
                    if (err) break;
                    if (lens[256] == 0) { err = 2; break; }
                    // the distance lengths follow the literal / length ones: move them to their own place
                    for (int i = ndist - 1; i >= 0; i--) lens[288 + i] = lens[nlen + i];
                    for (int i = nlen; i < 288; i++) lens[i] = 0;
                    for (int i = ndist; i < 30; i++) lens[288 + i] = 0;
                }
                if (!huff_build(hl, lens, 288) || !huff_build(hd, lens + 288, 30)) { err = 2; break; }
                table_fill(lt, LT_BITS, hl, lens, 288);
                table_fill(dt, DT_BITS, hd, lens + 288, 30);
                stored = false;
                in_block = true;
            } else { err = 1; break; }
            continue;
        }
        if (stored) {
            // whole bytes from the bit buffer
            if (stored_left == 0) { in_block = false; continue; }
            refill();
            int n = min(stored_left, bc >> 3);
            n = min(n, 4);
            for (int i = 0; i < n; i++) out[op++] = (uint8_t)take(8);
            stored_left -= n;
            continue;
        }
        if (copy_left > 0) {                                           // a match in progress: up to eight bytes a turn
            const int n = min(copy_left, 8);
            if (copy_dist >= n) {                                      // the sources are all written: eight loads on their way at once, then the stores
                uint8_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = i < n ? out[op - copy_dist + i] : (uint8_t)0;
#pragma unroll
                for (int i = 0; i < 8; i++) if (i < n) out[op + i] = v[i];
            } else {                                                   // an overlapping match repeats its last copy_dist bytes
                uint8_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = i < copy_dist ? out[op - copy_dist + i] : (uint8_t)0;
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (i < n) {
                        const int j = i % copy_dist;                   // (copy_dist <= 7 here)
                        uint8_t b = v[0];
#pragma unroll
                        for (int q = 1; q < 8; q++) b = j == q ? v[q] : b;
                        out[op + i] = b;
                    }
            }
            op += n;
            copy_left -= n;
            continue;
        }
        refill();
        int sym;
        {
            const uint32_t e = lt[(uint32_t)bb & (LT_SZ - 1)];
            if (e) { take(e & 15); sym = (int)(e >> 4); }
            else sym = slow(hl);
        }
        if (sym < 0) { err = 3; break; }
        if (sym < 256) {
            if (op >= isize) { err = 4; break; }
            out[op++] = (uint8_t)sym;
        } else if (sym == 256) {
            in_block = false;
        } else {
            const int c = sym - 257;
            if (c > 28) { err = 3; break; }
            refill();
            int len;
            if (c < 8) len = 3 + c;
            else if (c == 28) len = 258;
            else {
                const int e = (c >> 2) - 1;
                len = 3 + ((4 + (c & 3)) << e) + (int)take(e);
            }
            refill();
            int dsym;
            {
                const uint32_t e = dt[(uint32_t)bb & (DT_SZ - 1)];
                if (e) { take(e & 15); dsym = (int)(e >> 4); }
                else dsym = slow(hd);
            }
            if (dsym < 0 || dsym > 29) { err = 3; break; }
            refill();
            int dist;
            if (dsym < 4) dist = 1 + dsym;
            else {
                const int e = (dsym >> 1) - 1;
                dist = 1 + ((2 + (dsym & 1)) << e) + (int)take(e);
            }
            if (dist > op) { err = 3; break; }
            if (op + len > isize) { err = 4; break; }
            copy_left = len;
            copy_dist = dist;
        }
    }
    if (!err && op != isize) err = 6;
    a.status[b] = err;
}

// ---- the cooperative form: SIXTEEN lanes per member, four members per wave.  Lane 0 of a group is the member's decoder: bit buffer, Huffman
// tables (the same per-member LDS tables as above), one symbol at a time -- but only FOUR lanes of a wave decode, so the wave executes the union of
// four decoders' paths, not of sixty-four (what bounded k_inflate).  The member's output lives in an 8 KB LDS ring: literals are single LDS writes of
// the decoder lane (up to eight per turn), a match is copied by all sixteen lanes (from the ring, or -- distances beyond the ring -- from the part
// of the output already flushed to HBM), and every 2 KB the sixteen lanes move a finished stretch of the ring to HBM.
constexpr int G_TAB = (2 * L_END + 255) & ~255;                     // bytes of a group's tables

template <int RING, int DBG = 0>
__global__ __launch_bounds__(64) void k_inflate16(InflateArgs a)
{
    int dbg = 0;
    constexpr int RMASK = RING - 1, FLUSH = RING / 4, G_LDS = G_TAB + RING;
    extern __shared__ uint32_t tabs[];
    const int lane = threadIdx.x, gi = lane >> 4, l16 = lane & 15, lead = lane & 48;
    const int b = blockIdx.x * 4 + gi;
    const bool live = b < a.n;
    uint8_t *gbase = reinterpret_cast<uint8_t *>(tabs) + gi * G_LDS;
    uint16_t *lbase = reinterpret_cast<uint16_t *>(gbase);
    uint16_t *lt = lbase + L_LT, *dt = lbase + L_DT, *hl = lbase + L_HL, *hd = lbase + L_HD;
    uint8_t *lens = reinterpret_cast<uint8_t *>(lbase + L_LENS);
    uint8_t *ring = gbase + G_TAB;
    const int bb_ = live ? b : 0;
    const int64_t c0 = a.coff[bb_];
    const int32_t clen = a.clen[bb_], isize = a.isize[bb_];
    uint8_t *out = a.out + a.ooff[bb_];
    const bool dec = live && l16 == 0;
    const int skew = (int)(c0 & 3);
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(a.comp + (c0 & ~int64_t(3)));
    const int64_t w_end = (skew + clen + 3) / 4 + 2;
    auto fetch4 = [&](int64_t w) -> U4w {
        U4w v = {0u, 0u, 0u, 0u};
        if (w + 3 < w_end) v = *reinterpret_cast<const U4w *>(wp + w);
        else {
            if (w < w_end) v.x = wp[w];
            if (w + 1 < w_end) v.y = wp[w + 1];
            if (w + 2 < w_end) v.z = wp[w + 2];
        }
        return v;
    };
    U4w cur = {0u, 0u, 0u, 0u}, nxt = {0u, 0u, 0u, 0u};
    int64_t wi = 1;
    uint64_t bb = 0;
    int bc = 0, err = 0;
    if (dec) {
        cur = fetch4(0);
        nxt = fetch4(4);
        bb = (uint64_t)(cur.x >> (8 * skew));
        bc = 32 - 8 * skew;
    }
    auto refill = [&]() {
        if (bc <= 32) {
            const int k = (int)(wi & 3);
            if (k == 0) {
                cur = nxt;
                nxt = fetch4(wi + 4);
            }
            const uint32_t w = k == 0 ? cur.x : k == 1 ? cur.y : k == 2 ? cur.z : cur.w;
            bb |= (uint64_t)w << bc;
            if (wi > w_end + 1) err = 5;
            wi++;
            bc += 32;
        }
    };
    auto take = [&](int n) -> uint32_t {
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        bc -= n;
        return v;
    };
    auto slow = [&](const uint16_t *h) -> int {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)take(1);
            const int cnt = h[l];
            if (code - cnt < first) return h[16 + index + (code - first)];
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    };
    int op = 0, flushed = 0;                                           // group-uniform: bytes produced / bytes already in HBM
    bool last = false, in_block = false, stored = false, done = !live;
    int stored_left = 0;
    while (!__all(done)) {
        if (DBG == 1) dbg++;
        // ---- the decoder lane: literals into the ring (up to eight), until a match, the stream's end or an error
        int nlit = 0, kind = 0, mlen = 0, mdist = 0;                   // kind 1: a match follows the literals; 2: the member is finished (or broken)
        if (dec && !done) {
            for (;;) {
                if (err) { kind = 2; break; }
                if (!in_block) {
                    if (last) { kind = 2; break; }
                    refill();
                    last = take(1) != 0;
                    const int type = (int)take(2);
                    if (type == 0) {
                        take(bc & 7);
                        refill();
                        const uint32_t len = take(16);
                        refill();
                        const uint32_t nl = take(16);
                        if ((len ^ 0xffffu) != nl) { err = 1; continue; }
                        stored = true;
                        stored_left = (int)len;
                        in_block = true;
                    } else if (type == 1 || type == 2) {
                        int nlen = 288, ndist = 30;
                        if (type == 1) {
                            for (int i = 0; i < 144; i++) lens[i] = 8;
                            for (int i = 144; i < 256; i++) lens[i] = 9;
                            for (int i = 256; i < 280; i++) lens[i] = 7;
                            for (int i = 280; i < 288; i++) lens[i] = 8;
                            for (int i = 0; i < 30; i++) lens[288 + i] = 5;
                        } else {
                            refill();
                            nlen = (int)take(5) + 257;
                            ndist = (int)take(5) + 1;
                            const int ncode = (int)take(4) + 4;
                            if (nlen > 286 || ndist > 30) { err = 2; continue; }
                            uint8_t *cl = reinterpret_cast<uint8_t *>(hd);  // 19 code-length-code lengths, in the distance code's area (built later)
                            for (int i = 0; i < 19; i++) cl[i] = 0;
                            for (int i = 0; i < ncode; i++) {
                                refill();
                                // order of the code length alphabet (RFC 1951 3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                                int sym;
                                if (i < 3) sym = 16 + i;
                                else if (i == 3) sym = 0;
                                else if ((i & 1) == 0) sym = 8 + ((i - 4) >> 1);      // i = 4, 6, 8, ... -> 8, 9, 10, ...
                                else sym = 7 - ((i - 5) >> 1);                         // i = 5, 7, 9, ... -> 7, 6, 5, ...
                                cl[sym] = (uint8_t)take(3);
                            }
                            if (!huff_build(hl, cl, 19)) { err = 2; continue; }
                            int idx = 0;
                            while (idx < nlen + ndist && !err) {
                                refill();
                                const int sym = slow(hl);
                                if (sym < 0) { err = 2; break; }
                                if (sym < 16) lens[idx++] = (uint8_t)sym;
                                else {
                                    int prev = 0, rep;
                                    refill();
                                    if (sym == 16) {
                                        if (idx == 0) { err = 2; break; }
                                        prev = lens[idx - 1];
                                        rep = 3 + (int)take(2);
                                    } else if (sym == 17) rep = 3 + (int)take(3);
                                    else rep = 11 + (int)take(7);
                                    if (idx + rep > nlen + ndist) { err = 2; break; }
                                    while (rep--) lens[idx++] = (uint8_t)prev;
                                }
                            }
                            if (err) continue;
                            if (lens[256] == 0) { err = 2; continue; }
                            for (int i = ndist - 1; i >= 0; i--) lens[288 + i] = lens[nlen + i];
                            for (int i = nlen; i < 288; i++) lens[i] = 0;
                            for (int i = ndist; i < 30; i++) lens[288 + i] = 0;
                        }
                        if (!huff_build(hl, lens, 288) || !huff_build(hd, lens + 288, 30)) { err = 2; continue; }
                        table_fill(lt, LT_BITS, hl, lens, 288);
                        table_fill(dt, DT_BITS, hd, lens + 288, 30);
                        stored = false;
                        in_block = true;
                    } else err = 1;
                    continue;
                }
                if (nlit == 8) break;
                if (stored) {
                    if (stored_left == 0) { in_block = false; continue; }
                    if (op + nlit >= isize) { err = 4; continue; }
                    refill();
                    ring[(op + nlit) & RMASK] = (uint8_t)take(8);
                    nlit++;
                    stored_left--;
                    continue;
                }
                refill();
                int sym;
                {
                    const uint32_t e = lt[(uint32_t)bb & (LT_SZ - 1)];
                    if (e) { take(e & 15); sym = (int)(e >> 4); }
                    else { sym = slow(hl); if (DBG == 4) dbg++; }
                }
                if (sym < 0) { err = 3; continue; }
                if (DBG == 2 && sym < 256) dbg++;
                if (DBG == 3 && sym > 256) dbg++;
                if (DBG == 5 && sym == 256) dbg++;
                if (sym < 256) {
                    if (op + nlit >= isize) { err = 4; continue; }
                    ring[(op + nlit) & RMASK] = (uint8_t)sym;
                    nlit++;
                    continue;
                }
                if (sym == 256) { in_block = false; continue; }
                const int c = sym - 257;
                if (c > 28) { err = 3; continue; }
                refill();
                if (c < 8) mlen = 3 + c;
                else if (c == 28) mlen = 258;
                else {
                    const int e = (c >> 2) - 1;
                    mlen = 3 + ((4 + (c & 3)) << e) + (int)take(e);
                }
                refill();
                int dsym;
                {
                    const uint32_t e = dt[(uint32_t)bb & (DT_SZ - 1)];
                    if (e) { take(e & 15); dsym = (int)(e >> 4); }
                    else dsym = slow(hd);
                }
                if (dsym < 0 || dsym > 29) { err = 3; continue; }
                refill();
                if (dsym < 4) mdist = 1 + dsym;
                else {
                    const int e = (dsym >> 1) - 1;
                    mdist = 1 + ((2 + (dsym & 1)) << e) + (int)take(e);
                }
                if (mdist > op + nlit) { err = 3; continue; }
                if (op + nlit + mlen > isize) { err = 4; continue; }
                kind = 1;
                break;
            }
        }
        // ---- the group: what the decoder found
        const int word = __shfl(nlit | (kind << 4) | (mlen << 8), lead);
        mdist = __shfl(mdist, lead);
        nlit = word & 15; kind = (word >> 4) & 3; mlen = word >> 8;
        op += nlit;
        if (kind == 1) {
            const bool near = mdist <= RING - 512;                     // every source is still in the ring (the copy overwrites at most 258 of its oldest bytes)
            const float rinv = 1.0f / (float)mdist;
            for (int k = l16; k < mlen; k += 16) {
                int j = k;
                if (mdist < mlen) {                                    // an overlapping match repeats its last mdist bytes: byte k = byte k mod mdist
                    int q = (int)((float)k * rinv);
                    j = k - q * mdist;
                    if (j < 0) j += mdist;
                    if (j >= mdist) j -= mdist;
                }
                const int src = op - mdist + j;
                const uint8_t v = near ? ring[src & RMASK] : out[src];
                ring[(op + k) & RMASK] = v;
            }
            op += mlen;
        }
        // ---- finished stretches of the ring go to HBM
        const int upto = kind == 2 ? op : op - (op - flushed) % FLUSH;
        if (upto - flushed >= FLUSH || (kind == 2 && upto > flushed)) {
            for (int i = flushed + l16; i < upto; i += 16) out[i] = ring[i & RMASK];
            flushed = upto;
        }
        if (kind == 2) {
            if (dec) a.status[b] = DBG ? dbg : err ? err : (op != isize ? 6 : 0);
            done = true;
        }
    }
}

bool g_lds_set[64] = {false};

}   // namespace

extern "C" int nc_inflate_device(nc_ctx *ctx, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen, uint8_t *d_out,
                                 const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_blocks < 0 || (n_blocks && (!d_comp || !d_coff || !d_clen || !d_out || !d_ooff || !d_isize || !d_status)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_inflate_device: bad argument");
    if (n_blocks == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = (size_t)64 * TAB_WORDS * 4;
    if (ctx->device >= 0 && ctx->device < 64 && !g_lds_set[ctx->device]) {
        NC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_inflate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        g_lds_set[ctx->device] = true;
    }
    InflateArgs a;
    a.comp = d_comp; a.coff = d_coff; a.clen = d_clen; a.out = d_out; a.ooff = d_ooff; a.isize = d_isize; a.n = n_blocks; a.status = d_status;
    if (!getenv("NC_INFLATE_LANE_PER_MEMBER")) {                    // default: sixteen lanes per member
        const char *rv = getenv("NC_INFLATE_RING");
        const int ring = rv ? atoi(rv) : 2048;
        const size_t lds16 = (size_t)4 * (G_TAB + ring);
        const dim3 grid((n_blocks + 3) / 4), block(64);
        const char *dv = getenv("NC_INFLATE_DEBUG");                   // experiment counters in status[]: 1 turns, 2 literals, 3 matches, 4 long codes, 5 blocks
        const int d = dv ? atoi(dv) : 0;
        if (d == 1) hipLaunchKernelGGL((k_inflate16<2048, 1>), grid, block, lds16, ctx->stream, a);
        else if (d == 2) hipLaunchKernelGGL((k_inflate16<2048, 2>), grid, block, lds16, ctx->stream, a);
        else if (d == 3) hipLaunchKernelGGL((k_inflate16<2048, 3>), grid, block, lds16, ctx->stream, a);
        else if (d == 4) hipLaunchKernelGGL((k_inflate16<2048, 4>), grid, block, lds16, ctx->stream, a);
        else if (d == 5) hipLaunchKernelGGL((k_inflate16<2048, 5>), grid, block, lds16, ctx->stream, a);
        else if (ring == 8192) hipLaunchKernelGGL(k_inflate16<8192>, grid, block, lds16, ctx->stream, a);
        else if (ring == 4096) hipLaunchKernelGGL(k_inflate16<4096>, grid, block, lds16, ctx->stream, a);
        else if (ring == 2048) hipLaunchKernelGGL(k_inflate16<2048>, grid, block, lds16, ctx->stream, a);
        else return nc_fail(ctx, NC_ERR_ARG, "NC_INFLATE_RING: 2048, 4096 or 8192");
        NC_HIP(ctx, hipGetLastError());
        return NC_OK;
    }
    hipLaunchKernelGGL(k_inflate, dim3((n_blocks + 63) / 64), dim3(64), lds, ctx->stream, a);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}
