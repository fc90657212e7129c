// DEFLATE (RFC 1951) on the device: the payloads of BGZF members (SAMv1 4.1: independent raw-deflate streams of at most 64 KB of data each)
// are inflated in HBM.  This replaces the host's inflate (libdeflate / zlib on worker threads: 73 % of the ingest of
// generate_SNP_pileups.py:134-164's input) on the way from a BAM file to the read pack.
//
// Two kernels, because the two halves of inflate want opposite shapes (DESIGN.md section 12 has the measurements that led here):
//   k_huff   ONE LANE PER MEMBER, eight members per workgroup.  Huffman decoding is a serial chain per stream but the SAME short loop for every stream:
//            64-bit bit buffer refilled from aligned dwords, per-lane first-level tables in LDS (literal / length 9 bits, distance 6 bits;
//            entry = symbol << 4 | code length), codes longer than the index by the canonical count / symbol walk.  It does NOT copy: a literal
//            leaves as the token 0x80000000 | byte (two literals in a row: | 0x10000 | second << 8), a match as length << 16 | distance, into
//            the member's token run.
//   k_lz     ONE WAVE PER MEMBER.  64 tokens per step: their output positions by a wave scan, all literals stored at once, the matches in
//            order with all 64 lanes copying (source and destination in a 16 KB LDS ring of the member's recent output, so overlapping and
//            chained matches are plain LDS traffic; older sources are read back from HBM); finished stretches leave LDS as aligned dwords.
// Stored, fixed and dynamic blocks.  Checked: the stream ends exactly at ISIZE bytes, distances stay inside the output, the input is not
// overrun; the CRC-32 of every member against its trailer by k_crc32 (nc_bgzf_crc_device) behind the resolution.
#include "nc_common.h"

namespace {

#ifndef NC_HUFF_DT_BITS
#define NC_HUFF_DT_BITS 5
#endif
constexpr int LT_BITS = 8, DT_BITS = NC_HUFF_DT_BITS, LT_SZ = 1 << LT_BITS, DT_SZ = 1 << DT_BITS;
// one lane's LDS, in BYTES: the two first-level tables (uint16), the canonical arrays of both codes -- count[16] (uint16) and the symbols by code as
// BYTES (the literal / length code's ninth bit in a bit mask: 30 of its 286 symbols need it) --, the code lengths of the block being set up as
// NIBBLES, the walk's start.  [r5] 1,164 + 144 bytes of window = 1,308 per member instead of 2,460: 16 members per workgroup, seven workgroups per
// CU.  The symbol loop is bound by instruction issue -- a wave instruction costs its four cycles whether 8 or 16 of the 64 lanes are live -- and the
// members in flight by LDS, so what a member's tables do not take, further lanes do.  (An 8-bit first-level table alone, at 8 lanes a wave, was
// within 4 % either way in round 4: more WAVES per SIMD do not help an issue-bound loop; more LANES per wave do.)
#ifndef NC_HUFF_WINPAD
#define NC_HUFF_WINPAD 4
#endif
constexpr int B_LT = 0, B_DT = B_LT + 2 * LT_SZ, B_HLC = B_DT + 2 * DT_SZ, B_HLS = B_HLC + 32, B_HLM = B_HLS + 288, B_HDC = B_HLM + 36, B_HDS = B_HDC + 32,
              B_LENS = B_HDS + 32, B_WALK = B_LENS + 160, B_END = B_WALK + 8;
static_assert(B_HLC % 2 == 0 && B_HDC % 2 == 0 && B_WALK % 2 == 0, "uint16 sections");
constexpr int TAB_WORDS = (B_END + 3) / 4 | 1;                      // odd pitch in words: lanes spread over the banks
#ifndef NC_HUFF_LPW_SH
#define NC_HUFF_LPW_SH 4
#endif
constexpr int LPW_SH = NC_HUFF_LPW_SH, LPW = 1 << LPW_SH;                                 // members (= lanes) per workgroup of k_huff, and its log2.  One workgroup's duration is its slowest member's
constexpr int WIN_DW = 32, WIN_PITCH = WIN_DW + NC_HUFF_WINPAD;     // a lane's window of the compressed stream: 32 dwords (+ pad: banks spread), topped up every 8 steps

struct InflateArgs {
    const uint8_t *comp;        // compressed payloads (the buffer is readable 8 bytes past the last payload)
    const int64_t *coff;        // byte offset of member b's deflate payload in comp
    const int32_t *clen;        // its length
    uint8_t *out;
    const int64_t *ooff;        // where member b's data goes
    const int32_t *isize;       // the length its trailer announces
    int32_t n;
    int32_t *status;            // 0 ok; 1 bad block type / stored length; 2 bad code lengths; 3 bad symbol / distance; 4 output overrun; 5 input
                                // overrun; 6 length differs from ISIZE; 7 (nc_bgzf_crc_device) CRC-32 differs from the trailer's
};

struct __attribute__((packed, aligned(4))) U4w { uint32_t x, y, z, w; };   // four dwords at a 4-byte aligned address

// canonical code of up to 288 symbols (RFC 1951 3.2.2) for the walk over the lengths: cnt[0..16) = count per length, sym[] = symbols by code (low
// byte; hi = bit mask of the ninth bits, NULL for a code of < 256 symbols)
struct Huff {
    uint16_t *cnt;
    uint8_t *sym, *hi;
    __device__ __forceinline__ int at(int pos) const { return sym[pos] | (hi ? ((hi[pos >> 3] >> (pos & 7)) & 1) << 8 : 0); }
};
// code lengths as nibbles (the literal / length + distance lengths of a block: 320 symbols in 160 bytes) or as plain bytes (the 19 of the code length code)
struct LensNib {
    uint8_t *p;
    int base;
    __device__ __forceinline__ int get(int i) const { const int k = base + i; return (p[k >> 1] >> ((k & 1) * 4)) & 15; }
    __device__ __forceinline__ void set(int i, int v) const { const int k = base + i, sh = (k & 1) * 4; p[k >> 1] = (uint8_t)((p[k >> 1] & ~(15 << sh)) | (v << sh)); }
};
struct LensByte {
    const uint8_t *p;
    __device__ __forceinline__ int get(int i) const { return p[i]; }
};
// lens[0..n) -> h; returns false for an over-subscribed set (an incomplete one is accepted: single-symbol distance codes are legal)
template <class L>
__device__ bool huff_build(const Huff &hf, const L &lens, int n)
{
    uint16_t *h = hf.cnt;
    for (int i = 0; i < 16; i++) h[i] = 0;
    if (hf.hi)
        for (int i = 0; i < 36; i++) hf.hi[i] = 0;
    for (int i = 0; i < n; i++) h[lens.get(i)]++;
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
        const int l = lens.get(i);
        if (!l) continue;
        uint16_t o = 0;
#pragma unroll
        for (int q = 1; q < 16; q++) o = l == q ? offs[q] : o;
#pragma unroll
        for (int q = 1; q < 16; q++) offs[q] = (uint16_t)(l == q ? offs[q] + 1 : offs[q]);
        hf.sym[o] = (uint8_t)i;
        if (hf.hi && (i & 256)) hf.hi[o >> 3] = (uint8_t)(hf.hi[o >> 3] | (1 << (o & 7)));
    }
    return true;
}

// what a literal / length symbol means, in 12 bits: kind x (0-5: a match length with x extra bits; 6: a literal; 7: end of block, or -- value
// 1 -- one of the two symbols that must not occur) | value << 3 (the byte; the length's base - 3).  The literal / length table holds this
// instead of the symbol: the symbol loop then has no length arithmetic left (RFC 1951 3.2.5's table, folded into the table build)
__device__ __forceinline__ uint32_t ll_enc(int s)
{
    if (s < 256) return 6u | (uint32_t)s << 3;
    if (s == 256) return 7u;
    const int c = s - 257;
    if (c > 28) return 7u | 1u << 3;
    const int e1 = (c < 8 || c == 28) ? 0 : (c >> 2) - 1;
    const int lbase = c < 8 ? 3 + c : c == 28 ? 258 : 3 + ((4 + (c & 3)) << e1);
    return (uint32_t)e1 | (uint32_t)(lbase - 3) << 3;
}

// the same for a distance symbol: extra-bit count (15: a symbol that must not occur) | symbol << 4
__device__ __forceinline__ uint32_t d_enc(int s)
{
    if (s > 29) return 15u | (uint32_t)s << 4;
    return (uint32_t)(s < 4 ? 0 : (s >> 1) - 1) | (uint32_t)s << 4;
}

// first-level table: index = the next `bits` bits of the stream (first bit lowest) -> symbol << 4 | length, 0 where the code is longer
// (MODE 1: ll_enc(symbol) << 4 | length; 2: d_enc(symbol) << 4 | length)
template <int MODE, class L>
__device__ void table_fill(uint16_t *tab, int bits, const uint16_t *h, const L &lens, int n)
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
        const int l = lens.get(s);
        if (!l) continue;
        uint32_t c = 0;
#pragma unroll
        for (int q = 1; q < 16; q++) c = l == q ? next[q] : c;
#pragma unroll
        for (int q = 1; q < 16; q++) next[q] = l == q ? next[q] + 1 : next[q];
        if (l > bits) continue;
        const uint32_t r = __brev(c) >> (32 - l);
        const uint16_t ent = (uint16_t)((MODE == 1 ? ll_enc(s) : MODE == 2 ? d_enc(s) : (uint32_t)s) << 4 | (uint32_t)l);
        for (uint32_t k = r; k < (1u << bits); k += 1u << l) tab[k] = ent;
    }
}

// where the canonical walk stands after `bits` lengths: w[0] = first code of length bits + 1 (before the shift), w[1] = symbols of length <= bits
__device__ void walk_start(uint16_t *w, int bits, const uint16_t *h)
{
    int first = 0, index = 0;
    for (int l = 1; l <= bits; l++) {
        first = (first + h[l]) << 1;
        index += h[l];
    }
    w[0] = (uint16_t)first;
    w[1] = (uint16_t)index;
}

__global__ __launch_bounds__(LPW) void k_huff(InflateArgs a, uint32_t *tok, int32_t *ntok)
{
    extern __shared__ uint32_t tabs[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x * LPW + lane;
    const bool live = b < a.n;
    uint8_t *lbase = reinterpret_cast<uint8_t *>(tabs + lane * TAB_WORDS);
    uint16_t *lt = reinterpret_cast<uint16_t *>(lbase + B_LT), *dt = reinterpret_cast<uint16_t *>(lbase + B_DT);
    const Huff hl = {reinterpret_cast<uint16_t *>(lbase + B_HLC), lbase + B_HLS, lbase + B_HLM};
    const Huff hd = {reinterpret_cast<uint16_t *>(lbase + B_HDC), lbase + B_HDS, nullptr};
    const Huff hcl = {reinterpret_cast<uint16_t *>(lbase + B_HLC), lbase + B_HLS, nullptr};      // the code length code (19 symbols) borrows the literal code's arrays
    const LensNib lens = {lbase + B_LENS, 0}, dlens = {lbase + B_LENS, 288};
    uint16_t *wk = reinterpret_cast<uint16_t *>(lbase + B_WALK);
    const int bb_ = live ? b : 0;
    const int64_t c0 = a.coff[bb_];
    const int32_t clen = a.clen[bb_], isize = a.isize[bb_];
    // token i of the workgroup's lane l is dword (i * LPW + l) of the workgroup's LPW x 65536 dwords: lanes decode in step, so a step's tokens
    // are one contiguous store
    uint32_t *tk = tok + ((size_t)blockIdx.x << (16 + LPW_SH)) + lane;
    // the compressed stream reaches the bit buffer through a per-lane LDS window of 64 dwords.  A global load inside the symbol loop costs
    // the WAVE a memory round trip (the s_waitcnt before its first use also waits for every token store in flight): with the loads of all
    // lanes issued together every 16 steps, and written to the window 16 steps later, that wait is paid once per 16 steps and is short.
    const int skew = (int)(c0 & 3);
    const uint32_t *wp = reinterpret_cast<const uint32_t *>(a.comp + (c0 & ~int64_t(3)));
    const int w_end = (skew + clen + 3) / 4 + 2;                       // dwords of the stream (+2: the last code may be looked up past its end)
    uint32_t *win = tabs + LPW * TAB_WORDS + lane * WIN_PITCH;
    auto fetch4 = [&](int w) -> U4w {
        U4w v = {0u, 0u, 0u, 0u};
        if (w + 3 < w_end) v = *reinterpret_cast<const U4w *>(wp + w);
        else {
            if (w < w_end) v.x = wp[w];
            if (w + 1 < w_end) v.y = wp[w + 1];
            if (w + 2 < w_end) v.z = wp[w + 2];
        }
        return v;
    };
    int rd = 0, wr = 0, pend = 0;                                      // dwords consumed / in the window / loaded but still in registers
    U4w pre[3];
    uint64_t bb = 0;
    int bc = 0, err = 0;
    if (live) {
#pragma unroll 1
        for (; wr < WIN_DW; wr += 4) *reinterpret_cast<U4w *>(win + wr) = fetch4(wr);
        bb = (uint64_t)(win[0] >> (8 * skew));
        bc = 32 - 8 * skew;
        rd = 1;
    }
    // every 8 iterations of a loop that consumes at most 48 bits per iteration (12 dwords): what the previous call loaded goes to the window, the
    // window's free part is loaded (at most 12 dwords).  After a call the window holds WIN_DW less what the 8 iterations before it consumed, >= 20
    // dwords: it never runs dry before the next call
    auto tick = [&]() {
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (q * 4 < pend) *reinterpret_cast<U4w *>(win + ((wr + q * 4) & (WIN_DW - 1))) = pre[q];
        wr += pend;
        int n4 = (WIN_DW - (wr - rd)) >> 2;
        n4 = n4 > 3 ? 3 : n4;
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (q < n4) pre[q] = fetch4(wr + q * 4);
        pend = n4 * 4;
    };
    auto refill = [&]() {                                              // at least 33 valid bits afterwards (no branch: the window word is read either way)
        const bool f = bc <= 32;
        const uint32_t w = win[rd & (WIN_DW - 1)];
        bb |= f ? (uint64_t)w << (f ? bc : 0) : 0ull;
        err = (f && rd > w_end + 1 && !err) ? 5 : err;
        rd += f;
        bc += f ? 32 : 0;
    };
    auto take = [&](int n) -> uint32_t {
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        bb >>= n;
        bc -= n;
        return v;
    };
    auto slow = [&](const Huff &h) -> int {                            // RFC 1951 decoding, one bit at a time (the code length code)
        int code = 0, first = 0, index = 0;
#pragma unroll 1
        for (int l = 1; l < 16; l++) {
            code |= (int)take(1);
            const int cnt = h.cnt[l];
            if (code - cnt < first) return h.at(index + (code - first));
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    };
    // the same walk for a code the first-level table has no entry for: it is longer than the table's index, so the walk starts behind those bits
    auto slow_from = [&](const Huff &h, const uint16_t *w, int bits) -> int {
        int code = (int)(__brev(take(bits)) >> (32 - bits)) << 1, first = w[0], index = w[1];
#pragma unroll 1
        for (int l = bits + 1; l < 16; l++) {
            code |= (int)take(1);
            const int cnt = h.cnt[l];
            if (code - cnt < first) return h.at(index + (code - first));
            index += cnt;
            first += cnt;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    };
    int op = 0, nt = 0;
    bool fin = !live;
#pragma unroll 1
    while (!fin) {                                                     // one turn per deflate block: the lanes of a wave set their tables up together
        tick();                                                        // (a block may consume its header only -- an empty stored block, 5 bytes -- and dozens of them may follow each other: the window is topped up per block, not only inside the symbol loops)
        refill();
        const bool last = take(1) != 0;
        const int type = (int)take(2);
        if (type == 0) {                                               // stored: skip to the byte boundary, LEN, NLEN, the bytes
            take(bc & 7);
            refill();
            const uint32_t len = take(16);
            refill();
            const uint32_t nlen = take(16);
            if ((len ^ 0xffffu) != nlen) err = 1;
            else if (op + (int)len > isize) err = 4;
            else {
#pragma unroll 1
                for (uint32_t i = 0; i < len; i++) {
                    if ((i & 7) == 0) tick();
                    refill();
                    tk[(size_t)(nt++) << LPW_SH] = 0x80000000u | take(8);
                }
                op += (int)len;
            }
        } else if (type == 3) err = 1;
        else {
            int nlen = 288, ndist = 30;
            if (type == 1) {
                uint8_t *lb = lbase + B_LENS;                           // (two lengths a byte)
                for (int i = 0; i < 72; i++) lb[i] = 0x88;              // 0 .. 143: 8
                for (int i = 72; i < 128; i++) lb[i] = 0x99;            // 144 .. 255: 9
                for (int i = 128; i < 140; i++) lb[i] = 0x77;           // 256 .. 279: 7
                for (int i = 140; i < 144; i++) lb[i] = 0x88;           // 280 .. 287: 8
                for (int i = 144; i < 159; i++) lb[i] = 0x55;           // 30 distance symbols: 5
                lb[159] = 0;
            } else {
                refill();
                nlen = (int)take(5) + 257;
                ndist = (int)take(5) + 1;
                const int ncode = (int)take(4) + 4;
                if (nlen > 286 || ndist > 30) err = 2;
                uint8_t *cl = lbase + B_HDS;                            // 19 code-length-code lengths, in the distance code's symbol area (built later)
                for (int i = 0; i < 19; i++) cl[i] = 0;
#pragma unroll 1
                for (int i = 0; i < ncode; i++) {
                    refill();
                    // order of the code length alphabet (RFC 1951 3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                    const int sym = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) == 0 ? 8 + ((i - 4) >> 1) : 7 - ((i - 5) >> 1);
                    cl[sym] = (uint8_t)take(3);
                }
                if (!err && !huff_build(hcl, LensByte{cl}, 19)) err = 2;
                int idx = 0;
#pragma unroll 1
                for (int it = 0; !err && idx < nlen + ndist; it++) {
                    if ((it & 7) == 0) tick();
                    refill();
                    const int sym = slow(hcl);
                    if (sym < 0) { err = 2; break; }
                    if (sym < 16) lens.set(idx++, sym);
                    else {
                        int prev = 0, rep;
                        refill();
                        if (sym == 16) {
                            if (idx == 0) { err = 2; break; }
                            prev = lens.get(idx - 1);
                            rep = 3 + (int)take(2);
                        } else if (sym == 17) rep = 3 + (int)take(3);
                        else rep = 11 + (int)take(7);
                        if (idx + rep > nlen + ndist) { err = 2; break; }
                        while (rep--) lens.set(idx++, prev);
                    }
                }
                if (!err && lens.get(256) == 0) err = 2;
                if (!err) {                                            // the distance lengths follow the literal / length ones: move them to their own place
                    for (int i = ndist - 1; i >= 0; i--) lens.set(288 + i, lens.get(nlen + i));
                    for (int i = nlen; i < 288; i++) lens.set(i, 0);
                    for (int i = ndist; i < 30; i++) lens.set(288 + i, 0);
                }
            }
            if (!err && (!huff_build(hl, lens, 288) || !huff_build(hd, dlens, 30))) err = 2;
            if (!err) {
                table_fill<1>(lt, LT_BITS, hl.cnt, lens, 288);
                table_fill<2>(dt, DT_BITS, hd.cnt, dlens, 30);
                walk_start(wk, LT_BITS, hl.cnt);
                walk_start(wk + 2, DT_BITS, hd.cnt);
            }
            // ---- the symbols of the block: the loop the kernel lives in.  ONE straight line of code for literals and matches alike: a wave
            // executes the union of its lanes' paths anyway, and as branches that union cost 210 instructions a step (a third of them exec-mask
            // bookkeeping); a literal lane simply takes zero-width length / distance fields.  Only the long-code walk is a real branch.
            bool run = !err;
#pragma unroll 1
            for (int it = 0; run; it++) {
                if ((it & 7) == 0) tick();
                refill();
                const uint32_t e = lt[(uint32_t)bb & (LT_SZ - 1)];
                uint32_t code = e >> 4;                                  // ll_enc of the symbol
                take((int)(e & 15));
                if (__builtin_expect(e == 0, 0)) {
                    const int sym = slow_from(hl, wk, LT_BITS);
                    code = sym < 0 ? (7u | 1u << 3) : ll_enc(sym);
                }
                const uint32_t x = code & 7u, val = code >> 3;
                const bool lit = x == 6, mat = x < 6, eob = code == 7u;
                // a literal often has a literal behind it: the step takes that one too when the table knows it (two codes: at most 30 of the 33 bits)
                const uint32_t e_2 = lt[(uint32_t)bb & (LT_SZ - 1)];
                const bool two = lit && e_2 != 0 && ((e_2 >> 4) & 7u) == 6 && op + 2 <= isize;
                take(two ? (int)(e_2 & 15) : 0);
                const int len = (int)val + 3 + (int)take(mat ? (int)x : 0);   // (a length code and its extra bits: at most 20 of the refill's 33 bits)
                refill();
                const uint32_t de = dt[(uint32_t)bb & (DT_SZ - 1)];
                uint32_t dcode = de >> 4;                                // d_enc of the distance symbol
                take(mat ? (int)(de & 15) : 0);
                if (__builtin_expect(mat && de == 0, 0)) {
                    const int dsym = slow_from(hd, wk + 2, DT_BITS);
                    dcode = dsym < 0 ? 15u : d_enc(dsym);
                }
                const int e2 = (int)(dcode & 15u), ds = (int)(dcode >> 4);
                const bool dbad = mat && e2 == 15;
                const int dist = (ds < 4 ? 1 + ds : 1 + ((2 + (ds & 1)) << e2)) + (int)take(mat && !dbad ? e2 : 0);   // (15 + 13 bits)
                const int add = lit ? (two ? 2 : 1) : mat ? len : 0;
                const int bad = ((x == 7 && !eob) || dbad || (mat && dist > op)) ? 3 : op + add > isize ? 4 : 0;
                const bool emit = (lit || mat) && !bad;
                if (emit) tk[(size_t)nt << LPW_SH] = lit ? 0x80000000u | (two ? 0x10000u | (e_2 >> 7) << 8 : 0u) | val : (uint32_t)len << 16 | (uint32_t)dist;
                nt += emit;
                op += emit ? add : 0;
                err = err ? err : bad;
                run = !err && !eob;
            }
        }
        if (err || last) fin = true;
    }
    if (live) {
        if (!err && op != isize) err = 6;
        ntok[b] = nt;
        a.status[b] = err;
    }
}

// a wave's inclusive prefix sum
__device__ __forceinline__ int wave_scan_incl(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    return v;
}

// One wave per member.  The member's recent output lives in an LDS ring of RING bytes (position p at ring[p & (RING - 1)]): a full 64 KB
// image per member allowed two waves per CU -- a wave alone on its SIMD, every LDS round trip exposed.  With a 16 KB ring nine members share
// a CU.  Finished stretches leave for HBM every FLUSH bytes; a match source older than the last flush is read back from there (the
// stores are waited for when they are issued and the loads pass the CU's cache by -- at WORKGROUP scope: reader and writer are one wave, the
// XCD's own L2 is where they meet; an agent-scope release made every flush write the whole L2 back, `buffer_wbl2`, 16 times per member).
template <int RING>
__global__ __launch_bounds__(64) void k_lz(int32_t n, const uint32_t *tok, const int32_t *ntok, uint8_t *out, const int64_t *ooff, const int32_t *isize,
                                           const int32_t *status)
{
    constexpr int M = RING - 1, SPAN = RING / 4, FLUSH = RING / 4;            // RING >= FLUSH + SPAN + 8
    extern __shared__ uint32_t ring_w[];
    uint8_t *ring = reinterpret_cast<uint8_t *>(ring_w);
    const int lane = threadIdx.x;
    // the LPW members of one k_huff workgroup share every 32-byte sector of their interleaved tokens: consecutive workgroups of ONE XCD (every
    // eighth of the grid) take them, so the sector comes from that XCD's L2 after its first use (member-order blocks spread the eight over the
    // eight L2s: 6.1 GB read per 14.5 k members for 0.8 GB of tokens, 2.3 GB this way)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = (((j >> LPW_SH) * 8 + xcd) << LPW_SH) + (j & (LPW - 1));
    if (b >= n || status[b]) return;
    const int nt = ntok[b], total = isize[b];
    const uint32_t *tk = tok + ((size_t)(b >> LPW_SH) << (16 + LPW_SH)) + (b & (LPW - 1));
    uint8_t *o = out + ooff[b];
    const int head = (int)((4 - (reinterpret_cast<uintptr_t>(o) & 3)) & 3);
    uint32_t *ow = reinterpret_cast<uint32_t *>(o + head);
    int base = 0, flushed = 0;                                         // bytes produced; bytes that are in HBM (0, or = head mod 4)
    auto flush = [&](int upto) {                                       // [flushed, upto rounded down to a destination dword) -> HBM
        int start = flushed;
        if (flushed == 0) {
            if (upto < head) return;
            if (lane < head) o[lane] = ring[lane];
            start = head;
        }
        const int nw = (upto - start) >> 2;
        for (int w = lane; w < nw; w += 64) {
            const int p = start + 4 * w, r = (p & M) >> 2;
            ow[(p - head) >> 2] = __builtin_amdgcn_alignbyte(ring_w[(r + 1) & (RING / 4 - 1)], ring_w[r], p & 3);
        }
        flushed = start + 4 * nw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };
    // the tokens of the next step are on their way while this one is resolved (a step normally takes all 64: the guess is seldom wrong)
    uint32_t t_next = lane < nt ? tk[(size_t)lane << LPW_SH] : 0u;
    int c_next = 0;
#pragma unroll 1
    for (int c = 0; c < nt;) {
        const int i = c + lane;
        const uint32_t t = c == c_next ? t_next : (i < nt ? tk[(size_t)i << LPW_SH] : 0u);
        c_next = c + 64;
        t_next = c_next + lane < nt ? tk[(size_t)(c_next + lane) << LPW_SH] : 0u;
        const bool lit = (t >> 31) != 0;
        int len = lit ? 1 + (int)((t >> 16) & 1u) : (int)(t >> 16);      // (a literal token carries one or two bytes)
        const int dist = (int)(t & 0xffffu);
        const int incl = wave_scan_incl(len, lane);
        // the step takes the tokens whose output ends within SPAN bytes (at least one: a token is at most 258 bytes); the rest wait for the next
        const uint64_t fit = __ballot(i < nt && incl <= SPAN);
        const int nv = __builtin_popcountll(fit);                      // (the scan is monotone: `fit` is a prefix of the lanes)
        if (lane >= nv) len = 0;
        const int pos = base + incl - len;
        if (lit && lane < nv) {
            ring[pos & M] = (uint8_t)t;
            if (len == 2) ring[(pos + 1) & M] = (uint8_t)(t >> 8);
        }
        // the ring holds the RING bytes before the end of this step's output (its literals are in already); what is older is in HBM:
        // flushed >= ring_lo, because a flush is due every FLUSH bytes and a step adds at most SPAN
        const int ring_lo = base + __builtin_amdgcn_readlane(incl, nv - 1) - RING;
        // matches whose whole source lies before this step's first byte depend on nothing the step produces: every lane copies its own (short
        // ones only -- the wave waits for the longest), sources beyond the ring read back from HBM side by side instead of one after the other
        const bool mat = !lit && len > 0;
        const bool indep = mat && len <= 32 && pos - dist + len <= base;
#pragma unroll 1
        for (int k = 0; __any(indep && k < len); k++)
            if (indep && k < len) {
                const int q = pos - dist + k;
                const uint8_t v = q >= ring_lo ? ring[q & M] : __hip_atomic_load(o + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ring[(pos + k) & M] = v;
            }
        // the others in token order, all lanes on one match
        uint64_t m = __ballot(mat && !indep);
#pragma unroll 1
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const int P = __builtin_amdgcn_readlane(pos, j), L = __builtin_amdgcn_readlane(len, j), D = __builtin_amdgcn_readlane(dist, j);
            const float rinv = 1.0f / (float)D;
            for (int k = lane; k < L; k += 64) {
                int r = k;
                if (D < L) {                                           // an overlapping match repeats its last D bytes: byte k = byte k mod D
                    const int q = (int)((float)k * rinv);
                    r = k - q * D;
                    if (r < 0) r += D;
                    if (r >= D) r -= D;
                }
                const int q = P - D + r;
                const uint8_t v = q >= ring_lo ? ring[q & M] : __hip_atomic_load(o + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ring[(P + k) & M] = v;
            }
        }
        base += __builtin_amdgcn_readlane(incl, nv - 1);
        c += nv;
        if (base - flushed >= FLUSH) flush(base);
    }
    flush(total);
    if (flushed + lane < total && total >= head) o[flushed + lane] = ring[(flushed + lane) & M];
    if (total < head && lane < total) o[lane] = ring[lane];
}


// ---- CRC-32 of the inflated members (RFC 1952 8; BGZF: the member's trailer holds it before ISIZE).  htslib checks it on every block it inflates
// (the reader behind generate_SNP_pileups.py:134); on the device route the bytes never reach the host, so the check runs here: a member whose
// bytes form a valid deflate stream of the announced length but are not the ones that were written is reported (status 7), not called from.
// One wave per member, four members per workgroup.  The member's <= 64 KB are cut from their END into 64 slices of 1024 bytes (the first
// non-empty slice is the short one and starts from the initial register 0xffffffff); a lane runs slice-by-4 over its slice (tables in LDS,
// dwords from 4-byte aligned addresses put in place by v_alignbyte); the 64 partial registers combine in a tree whose level l multiplies by
// x^(8 * 1024 * 2^l) mod P -- six constants, since every slice but the first has the same length (an empty lane's register is 0).
constexpr uint32_t CRC_POLY = 0xEDB88320u;
struct CrcOps { uint32_t x[6]; };                                      // x^(8 * 1024 * 2^l) mod P, reflected representation, l = 0 .. 5

__host__ __device__ inline uint32_t crc_multmodp(uint32_t a, uint32_t b)      // a * b mod P (zlib's crc32 combine arithmetic)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}

__global__ __launch_bounds__(256) void k_crc32(int32_t n_blocks, const uint8_t *__restrict__ comp, const int64_t *__restrict__ coff, const int32_t *__restrict__ clen,
                                               const uint8_t *__restrict__ out, const int64_t *__restrict__ ooff, const int32_t *__restrict__ isize,
                                               int32_t *__restrict__ status, CrcOps ops)
{
    __shared__ uint32_t T[4][256];
    const int tid = threadIdx.x, lane = tid & 63;
    {
        uint32_t c = (uint32_t)tid;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        T[0][tid] = c;
    }
    __syncthreads();
    {
        uint32_t c = T[0][tid];
#pragma unroll
        for (int t = 1; t < 4; t++) { c = (c >> 8) ^ T[0][c & 0xffu]; T[t][tid] = c; }
    }
    __syncthreads();
    const int b = blockIdx.x * 4 + (tid >> 6);
    if (b >= n_blocks) return;
    const int n = isize[b];
    const uint8_t *base = out + ooff[b];
    // slice `lane` covers bytes [n - (64 - lane) * 1024, n - (63 - lane) * 1024) of the member, clipped at 0
    const int hi = n - (63 - lane) * 1024, lo = max(hi - 1024, 0);
    int len = hi > 0 ? hi - lo : 0;
    uint32_t crc = (len > 0 && lo == 0) ? 0xffffffffu : 0u;            // the first non-empty slice carries the initial register
    const uint8_t *q = base + lo;
    for (; len > 0 && ((uintptr_t)q & 3); len--) crc = (crc >> 8) ^ T[0][(crc ^ *q++) & 0xffu];      // up to the next aligned dword (the short first slice only, or an unaligned member)
    const uint32_t *w = reinterpret_cast<const uint32_t *>(q);
    for (int i = 0; i + 4 <= len; i += 4) {
        const uint32_t x = crc ^ *w++;
        crc = T[3][x & 0xffu] ^ T[2][(x >> 8) & 0xffu] ^ T[1][(x >> 16) & 0xffu] ^ T[0][x >> 24];
    }
    q = reinterpret_cast<const uint8_t *>(w);
    for (int i = len & ~3; i < len; i++) crc = (crc >> 8) ^ T[0][(crc ^ *q++) & 0xffu];
    // tree: at level l the register of the left half moves 1024 * 2^l bytes forward
#pragma unroll
    for (int l = 0; l < 6; l++) {
        const uint32_t other = (uint32_t)__shfl_xor((int)crc, 1 << l);
        const bool right = (lane >> l) & 1;
        const uint32_t left_c = right ? other : crc, right_c = right ? crc : other;
        crc = crc_multmodp(ops.x[l], left_c) ^ right_c;               // (both lanes of a pair compute the same value)
    }
    if (lane == 0) {
        const uint32_t have = n > 0 ? crc ^ 0xffffffffu : 0u;
        const uint8_t *t = comp + coff[b] + clen[b];
        const uint32_t want = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        if (have != want && status[b] == 0) status[b] = 7;
    }
}
}   // namespace

// d_tok: workspace of ceil(n_blocks / 64) x 4,194,304 dwords (64 members x 65,536 tokens, interleaved); d_ntok: n_blocks counters.
// phase: 1 = tokens only (k_huff), 2 = resolution only (k_lz, of tokens made by an earlier phase-1 call), 3 = both.  The halves may run on
// different streams (nc_ctx_set_stream between the calls): k_huff takes a CU's whole LDS, so in a round that does not fill the GPU the
// match resolution of the previous batch runs on the CUs it leaves free.
static int inflate_phase(nc_ctx *ctx, int phase, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen, uint8_t *d_out,
                         const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status, uint32_t *d_tok, int32_t *d_ntok)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_blocks < 0 || (n_blocks && (!d_comp || !d_coff || !d_clen || !d_out || !d_ooff || !d_isize || !d_status || !d_tok || !d_ntok)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_inflate_device: bad argument");
    if (n_blocks == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds_h = (size_t)LPW * (TAB_WORDS + WIN_PITCH) * 4;
    if (!ctx->huff_lds_set) {                                            // per context (= per device, one thread): no process-wide table
        NC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_huff), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
        ctx->huff_lds_set = true;
    }
    if (phase & 1) {
        InflateArgs a;
        a.comp = d_comp; a.coff = d_coff; a.clen = d_clen; a.out = d_out; a.ooff = d_ooff; a.isize = d_isize; a.n = n_blocks; a.status = d_status;
        hipLaunchKernelGGL(k_huff, dim3((n_blocks + LPW - 1) / LPW), dim3(LPW), lds_h, ctx->stream, a, d_tok, d_ntok);
        NC_HIP(ctx, hipGetLastError());
    }
    if (phase & 2) {
        const int lz_grid = (n_blocks + 8 * LPW - 1) / (8 * LPW) * (8 * LPW);   // (k_lz's block -> member map covers whole groups of 8 x LPW)
        const char *rv = getenv("NC_INFLATE_RING");                    // (experiment switch: 16384 default, 32768)
        if (rv && atoi(rv) == 32768)
            hipLaunchKernelGGL(k_lz<32768>, dim3(lz_grid), dim3(64), 32768, ctx->stream, n_blocks, (const uint32_t *)d_tok, (const int32_t *)d_ntok, d_out,
                               d_ooff, d_isize, (const int32_t *)d_status);
        else
            hipLaunchKernelGGL(k_lz<16384>, dim3(lz_grid), dim3(64), 16384, ctx->stream, n_blocks, (const uint32_t *)d_tok, (const int32_t *)d_ntok, d_out,
                               d_ooff, d_isize, (const int32_t *)d_status);
        NC_HIP(ctx, hipGetLastError());
    }
    return NC_OK;
}

extern "C" int nc_inflate_device(nc_ctx *ctx, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen, uint8_t *d_out,
                                 const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status, uint32_t *d_tok, int32_t *d_ntok)
{
    return inflate_phase(ctx, 3, n_blocks, d_comp, d_coff, d_clen, d_out, d_ooff, d_isize, d_status, d_tok, d_ntok);
}

extern "C" int nc_inflate_device_phase(nc_ctx *ctx, int32_t phase, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen,
                                       uint8_t *d_out, const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status, uint32_t *d_tok, int32_t *d_ntok)
{
    if (phase < 1 || phase > 3) return ctx ? nc_fail(ctx, NC_ERR_ARG, "nc_inflate_device_phase: phase 1, 2 or 3") : NC_ERR_ARG;
    return inflate_phase(ctx, phase, n_blocks, d_comp, d_coff, d_clen, d_out, d_ooff, d_isize, d_status, d_tok, d_ntok);
}

// CRC-32 of every inflated member against its trailer; a mismatch sets d_status[b] = 7 (where the inflate itself left 0)
extern "C" int nc_bgzf_crc_device(nc_ctx *ctx, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen, const uint8_t *d_out,
                                  const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_blocks < 0 || (n_blocks && (!d_comp || !d_coff || !d_clen || !d_out || !d_ooff || !d_isize || !d_status)))
        return nc_fail(ctx, NC_ERR_ARG, "nc_bgzf_crc_device: bad argument");
    if (n_blocks == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    static const CrcOps ops = []() {
        CrcOps o;
        uint32_t p = 1u << 30;                                          // x^1
        for (int k = 0; k < 13; k++) p = crc_multmodp(p, p);           // x^(2^13) = x^(8 * 1024)
        for (int l = 0; l < 6; l++) { o.x[l] = p; p = crc_multmodp(p, p); }
        return o;
    }();
    hipLaunchKernelGGL(k_crc32, dim3((n_blocks + 3) / 4), dim3(256), 0, ctx->stream, n_blocks, d_comp, d_coff, d_clen, d_out, d_ooff, d_isize, d_status, ops);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}
