// Native BGZF / BAM / BAI ingest: decodes alignments straight into the read-major form the read pack is built from.
//
// Replaces the pysam/htslib objects of the reference (generate_SNP_pileups.py:134-164, generate_indel_pileups.py:147,
// 178-188, 213-235): per alignment the reference span [start, end), one base code per spanned reference position
// (A=0 G=1 T=2 C=3; deletion, reference skip and any other base = 4, i.e. the '*' / 'N' rows of base_to_num_map,
// generate_SNP_pileups.py:104), the insertion / deletion markers that pysam appends to the pileup string of the column
// BEFORE the event ('+n' / '-n'), the HP / PS tags, and the query sequence (for the indel pass-2 read slices).
// Host code only; file formats follow the SAM/BAM specification (SAMv1 section 4 and 5).
#include <dlfcn.h>
#include <sys/mman.h>
#include <zlib.h>
#include <tmmintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/nanocaller_hip.h"
#include "nc_host.h"

namespace {

// BGZF reader with parallel read-ahead: BGZF blocks (<= 64 KB of payload each) are independent deflate streams, so a window
// of upcoming blocks is read from the file in one go and inflated (+ CRC-checked) by a pool of host threads; the consumer
// walks the window in order.  The window starts small (region queries stop early) and doubles up to 1024 blocks (~64 MB).
struct Bgzf {
    FILE *f = nullptr;
    struct Blk {
        int64_t coff = 0, next = 0;
        std::vector<uint8_t> comp, data;
        uint32_t isize = 0, crc = 0;
        int clen = 0;
        bool ok = true;
    };
    std::vector<Blk> win;       // inflated window, consecutive blocks
    size_t wi = 0;              // next block of the window to hand out
    size_t grow = 16;           // blocks to read ahead next time
    int max_threads = 0;        // inflate threads (0 = all cores, capped at 32)
    size_t grow_cap = 1024;     // largest window of blocks inflated ahead of the parser (a region decode stops inside its last window: what lies past the region's end there was inflated for nothing)
    std::vector<uint8_t> block;
    int64_t block_coff = 0;     // compressed offset of the current block
    int64_t next_coff = 0;
    size_t upos = 0;            // read position inside `block`
    bool eof = false;

    // raw block at the current file position -> false on error; *at_eof when the file ended cleanly
    bool read_raw(Blk &k, int64_t coff, bool *at_eof)
    {
        uint8_t h[18];
        const size_t n = fread(h, 1, 18, f);
        if (n == 0) { *at_eof = true; return true; }
        if (n != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        const int xlen = h[10] | (h[11] << 8);
        // find the BC subfield (it is first in every writer in practice, but walk the extra field to be safe)
        std::vector<uint8_t> extra((size_t)xlen);
        memcpy(extra.data(), h + 12, (size_t)std::min(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, (size_t)xlen - 6, f) != (size_t)xlen - 6) return false;
        int bsize = -1;
        for (int p = 0; p + 4 <= xlen;) {
            const int slen = extra[p + 2] | (extra[p + 3] << 8);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2) bsize = (extra[p + 4] | (extra[p + 5] << 8)) + 1;
            p += 4 + slen;
        }
        if (bsize < 0) return false;
        const int clen = bsize - xlen - 12 - 8;
        if (clen < 0) return false;
        k.comp.resize((size_t)clen + 8);
        if (fread(k.comp.data(), 1, k.comp.size(), f) != k.comp.size()) return false;
        const uint8_t *t = k.comp.data() + clen;
        k.crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        k.isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
        k.clen = clen;
        k.coff = coff;
        k.next = coff + bsize;
        return true;
    }
    struct Deflate {
        bool ok = false;
        void *(*alloc)() = nullptr;
        int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
        uint32_t (*crc32)(uint32_t, const void *, size_t) = nullptr;
        void (*free_dec)(void *) = nullptr;
    };
    static const Deflate &deflate_lib()
    {
        static const Deflate D = []() {
            Deflate d;
            if (getenv("NC_BAM_ZLIB")) return d;
            void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return d;
            d.alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
            d.decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
            d.crc32 = (uint32_t(*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
            d.free_dec = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
            d.ok = d.alloc && d.decompress && d.crc32;
            return d;
        }();
        return D;
    }
    static void inflate_blk(Blk &k)
    {
        k.data.resize(k.isize);
        if (!k.isize) return;
        // libdeflate (2-3x zlib's inflate rate, vectorised CRC-32) when the shared library is present; it ships without headers
        // in this image, so its four entry points are bound at run time.  zlib otherwise.
        const Deflate &D = deflate_lib();
        if (D.ok) {
            // one decompressor per thread, released when the thread ends (decode threads live for one call)
            struct Dec {
                void *p = nullptr;
                void (*free_fn)(void *) = nullptr;
                ~Dec() { if (p && free_fn) free_fn(p); }
            };
            thread_local Dec tl;
            if (!tl.p) { tl.p = D.alloc(); tl.free_fn = D.free_dec; }
            void *dec = tl.p;
            size_t got = 0;
            if (dec && D.decompress(dec, k.comp.data(), (size_t)k.clen, k.data.data(), (size_t)k.isize, &got) == 0 && got == (size_t)k.isize) {
                if (D.crc32(0, k.data.data(), (size_t)k.isize) != k.crc) k.ok = false;
                std::vector<uint8_t>().swap(k.comp);
                return;
            }
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { k.ok = false; return; }
        zs.next_in = k.comp.data();
        zs.avail_in = (uInt)k.clen;
        zs.next_out = k.data.data();
        zs.avail_out = k.isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || (uint32_t)crc32(0L, k.data.data(), k.isize) != k.crc) k.ok = false;
        std::vector<uint8_t>().swap(k.comp);
    }
    bool fill(int64_t coff)
    {
        win.clear();
        wi = 0;
        if (fseeko(f, coff, SEEK_SET) != 0) return false;
        bool at_eof = false;
        int64_t c = coff;
        for (size_t i = 0; i < grow; i++) {
            Blk k;
            if (!read_raw(k, c, &at_eof)) return false;
            if (at_eof) break;
            c = k.next;
            win.push_back(std::move(k));
        }
        const size_t n = win.size();
        const int hw = nc_host_cpus();
        size_t T = (size_t)(hw > 32 ? 32 : hw);
        if (const char *e = getenv("NC_BAM_THREADS")) T = (size_t)std::max(1, atoi(e));       // 1 = sequential inflate
        if (max_threads > 0) T = (size_t)max_threads;
        if (T > n / 4) T = n / 4;                             // a thread per >= 4 blocks
        if (T <= 1) {
            for (auto &k : win) inflate_blk(k);
        } else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++)
                th.emplace_back([this, t, T, n]() { for (size_t i = t; i < n; i += T) inflate_blk(win[i]); });
            for (auto &x : th) x.join();
        }
        for (auto &k : win) if (!k.ok) return false;
        if (grow < grow_cap) grow *= 2;
        return true;
    }
    bool load_block(int64_t coff)
    {
        if (wi >= win.size() || win[wi].coff != coff) {
            if (!fill(coff)) return false;
            if (win.empty()) { eof = true; block.clear(); upos = 0; block_coff = coff; return true; }
        }
        Blk &k = win[wi++];
        block.swap(k.data);
        block_coff = k.coff;
        next_coff = k.next;
        upos = 0;
        return true;
    }

    bool seek(uint64_t voff)
    {
        eof = false;
        grow = 16;
        if (!load_block((int64_t)(voff >> 16))) return false;
        upos = (size_t)(voff & 0xffff);
        return upos <= block.size();
    }
    // returns false on EOF / error; *ok distinguishes
    bool read(void *dst, size_t n, bool *err)
    {
        uint8_t *d = (uint8_t *)dst;
        while (n) {
            if (upos >= block.size()) {
                if (eof) return false;
                if (!load_block(next_coff)) { *err = true; return false; }
                if (eof) return false;
                continue;
            }
            const size_t k = std::min(n, block.size() - upos);
            memcpy(d, block.data() + upos, k);
            d += k; upos += k; n -= k;
        }
        return true;
    }
};

inline int32_t rd32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }
inline uint32_t rdu32(const uint8_t *p) { return (uint32_t)rd32(p); }

// 4-bit BAM base "=ACMGRSVTWYHKDBN" -> reference code map A=0 G=1 T=2 C=3, everything else 4
const uint8_t NT16_CODE[16] = {4, 0, 3, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4};
// codes of the two bases of a byte of BAM's 4-bit sequence (first base = high nibble), as they lie in memory
struct Nt16Pair {
    uint16_t v[256];
    Nt16Pair()
    {
        for (int b = 0; b < 256; b++) {
            const uint8_t two[2] = {NT16_CODE[b >> 4], NT16_CODE[b & 15]};
            memcpy(&v[b], two, 2);
        }
    }
};
const Nt16Pair NT16_PAIR;

// out[q] = NT16_CODE[base q of BAM's 4-bit sequence], q < n (the buffer takes up to 31 bytes more): the whole SEQ of a record at once,
// so that every M run of its CIGAR is a plain copy.  32 bases a step through two 16-entry byte shuffles where the CPU has SSSE3.
__attribute__((target("ssse3"))) void seq_codes_ssse3(const uint8_t *seq, int32_t n, uint8_t *out)
{
    const __m128i lut = _mm_loadu_si128(reinterpret_cast<const __m128i *>(NT16_CODE)), nib = _mm_set1_epi8(15);
    const int32_t nb = (n + 1) / 2;
    int32_t b = 0;
    for (; b + 16 <= nb; b += 16) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(seq + b));
        const __m128i hi = _mm_shuffle_epi8(lut, _mm_and_si128(_mm_srli_epi16(v, 4), nib)), lo = _mm_shuffle_epi8(lut, _mm_and_si128(v, nib));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(out + 2 * b), _mm_unpacklo_epi8(hi, lo));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(out + 2 * b + 16), _mm_unpackhi_epi8(hi, lo));
    }
    for (; b < nb; b++) memcpy(out + 2 * b, &NT16_PAIR.v[seq[b]], 2);
}
void seq_codes_plain(const uint8_t *seq, int32_t n, uint8_t *out)
{
    for (int32_t b = 0, nb = (n + 1) / 2; b < nb; b++) memcpy(out + 2 * b, &NT16_PAIR.v[seq[b]], 2);
}
// (NC_BAM_PLAIN_SEQ=1: the table form on any CPU -- what tests/test_bam_ingest.py compares the shuffle form with)
void (*const SEQ_CODES)(const uint8_t *, int32_t, uint8_t *) =
    (__builtin_cpu_init(), __builtin_cpu_supports("ssse3")) && !getenv("NC_BAM_PLAIN_SEQ") ? seq_codes_ssse3 : seq_codes_plain;

}   // namespace

struct nc_bam {
    Bgzf z;
    std::string path;
    std::vector<std::string> ref_name;
    std::vector<int32_t> ref_len;
    uint64_t first_rec_voff = 0;
    // BAI linear index: per reference, smallest virtual offset of an alignment overlapping each 16 kb window
    std::vector<std::vector<uint64_t>> lin;
    bool have_bai = false;
    // CSI index (samtools index -c; needed for contigs longer than 2^29): per reference, bin -> (loffset, chunks)
    struct CsiBin { uint64_t loff; std::vector<std::pair<uint64_t, uint64_t>> chunks; };
    std::vector<std::unordered_map<uint32_t, CsiBin>> csi;
    int32_t csi_min_shift = 14, csi_depth = 5;
    bool have_csi = false;
    char err[256] = {0};
};

// allocator whose resize() leaves new elements uninitialised: the gigabyte arrays are filled right after they are sized,
// and their pages are first touched by whichever thread fills them (parallel page faults in the merged decode)
// Blocks of 4 MB and more are 2 MB-aligned and advised as transparent huge pages: first-touch page faults (one per 4 KB
// page, serialised across threads by the kernel's memory accounting) are what bounds a multi-threaded decode otherwise.
template <class T>
struct NoInit {
    typedef T value_type;
    NoInit() noexcept {}
    template <class U> NoInit(const NoInit<U> &) noexcept {}
    template <class U> struct rebind { typedef NoInit<U> other; };
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        void *p = nullptr;
        if (bytes >= ((size_t)4 << 20)) {
            const size_t huge = (size_t)2 << 20, rounded = (bytes + huge - 1) & ~(huge - 1);
            if (posix_memalign(&p, huge, rounded) != 0) throw std::bad_alloc();
            (void)madvise(p, rounded, MADV_HUGEPAGE);
        } else {
            p = malloc(bytes ? bytes : 1);
            if (!p) throw std::bad_alloc();
        }
        return (T *)p;
    }
    void deallocate(T *p, size_t) noexcept { free(p); }
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const NoInit<U> &) const noexcept { return true; }
    template <class U> bool operator!=(const NoInit<U> &) const noexcept { return false; }
};
template <class T> using bigvec = std::vector<T, NoInit<T>>;

struct nc_decoded {
    bigvec<int32_t> start, end, flag, ev_off, ev_pos, ev_len, ps, name_off, qstart;
    bigvec<int64_t> off, seq_off;
    bigvec<uint8_t> codes, hap, seq;
    bigvec<char> names;
};

namespace {

int bam_fail(nc_bam *b, int code, const char *msg)
{
    if (b) snprintf(b->err, sizeof b->err, "%s", msg);
    return code;
}

bool load_bai(nc_bam *b)
{
    FILE *f = fopen((b->path + ".bai").c_str(), "rb");
    if (!f) {
        std::string alt = b->path;
        if (alt.size() > 4 && alt.substr(alt.size() - 4) == ".bam") alt = alt.substr(0, alt.size() - 4) + ".bai";
        f = fopen(alt.c_str(), "rb");
    }
    if (!f) return false;
    auto r32 = [&](int32_t &v) { uint8_t t[4]; if (fread(t, 1, 4, f) != 4) return false; v = rd32(t); return true; };
    auto r64 = [&](uint64_t &v) { uint8_t t[8]; if (fread(t, 1, 8, f) != 8) return false; v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | t[i]; return true; };
    char magic[4];
    int32_t n_ref = 0;
    bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "BAI\1", 4) == 0 && r32(n_ref) && n_ref == (int32_t)b->ref_name.size();
    b->lin.assign(ok ? (size_t)n_ref : 0, {});
    for (int32_t r = 0; ok && r < n_ref; r++) {
        int32_t n_bin = 0;
        ok = r32(n_bin);
        for (int32_t k = 0; ok && k < n_bin; k++) {
            int32_t bin = 0, n_chunk = 0;
            ok = r32(bin) && r32(n_chunk) && fseeko(f, (off_t)n_chunk * 16, SEEK_CUR) == 0;
        }
        int32_t n_intv = 0;
        ok = ok && r32(n_intv);
        if (ok) b->lin[(size_t)r].resize((size_t)n_intv);
        for (int32_t k = 0; ok && k < n_intv; k++) ok = r64(b->lin[(size_t)r][(size_t)k]);
    }
    fclose(f);
    if (!ok) b->lin.clear();
    return ok;
}

// <bam>.csi: a BGZF-compressed CSIv1 file (hts-specs CSIv1.pdf); zlib's gz reader handles the multi-member stream
bool load_csi(nc_bam *b)
{
    gzFile g = gzopen((b->path + ".csi").c_str(), "rb");
    if (!g) return false;
    std::vector<uint8_t> raw;
    uint8_t buf[1 << 16];
    for (int k; (k = gzread(g, buf, sizeof buf)) > 0;) raw.insert(raw.end(), buf, buf + k);
    gzclose(g);
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= raw.size(); };
    auto r32 = [&]() { const int32_t v = rd32(raw.data() + p); p += 4; return v; };
    auto r64 = [&]() { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | raw[p + (size_t)i]; p += 8; return v; };
    if (!need(16) || memcmp(raw.data(), "CSI\1", 4) != 0) return false;
    p = 4;
    b->csi_min_shift = r32();
    b->csi_depth = r32();
    const int32_t l_aux = r32();
    if (l_aux < 0 || !need((size_t)l_aux + 4) || b->csi_min_shift < 1 || b->csi_min_shift > 30 || b->csi_depth < 1 || b->csi_depth > 9) return false;
    p += (size_t)l_aux;
    const int32_t n_ref = r32();
    if (n_ref != (int32_t)b->ref_name.size()) return false;
    b->csi.assign((size_t)n_ref, {});
    for (int32_t r = 0; r < n_ref; r++) {
        if (!need(4)) return false;
        const int32_t n_bin = r32();
        for (int32_t k = 0; k < n_bin; k++) {
            if (!need(16)) return false;
            const uint32_t bin = (uint32_t)r32();
            nc_bam::CsiBin cb;
            cb.loff = r64();
            const int32_t n_chunk = r32();
            if (n_chunk < 0 || !need((size_t)n_chunk * 16)) return false;
            for (int32_t c = 0; c < n_chunk; c++) { const uint64_t u = r64(), v = r64(); cb.chunks.emplace_back(u, v); }
            b->csi[(size_t)r].emplace(bin, std::move(cb));
        }
    }
    return true;
}

// htslib's rule (hts_itr_query): min_off = loffset of the finest existing bin that holds `beg` or lies to its left (walking up
// the hierarchy); the scan starts at the smallest chunk start among the bins overlapping [beg, end) whose chunk end exceeds it.
// -> virtual offset to start reading from, or 0 when the index holds nothing for the interval (nothing to read)
uint64_t csi_start(const nc_bam *b, int32_t tid, int64_t beg0, int64_t end0, bool *empty)
{
    const auto &bins = b->csi[(size_t)tid];
    const int ms = b->csi_min_shift, dp = b->csi_depth;
    *empty = false;
    auto first_of = [](int lvl) { return (uint32_t)(((1ull << (3 * lvl)) - 1) / 7); };
    const int64_t maxpos = (int64_t)1 << (ms + 3 * dp);
    if (beg0 < 0) beg0 = 0;
    if (end0 > maxpos) end0 = maxpos;
    if (beg0 >= end0) { *empty = true; return 0; }
    uint64_t min_off = 0;
    {
        uint32_t bin = first_of(dp) + (uint32_t)(beg0 >> ms);
        for (;;) {
            auto it = bins.find(bin);
            if (it != bins.end()) { min_off = it->second.loff; break; }
            if (bin == 0) break;
            const uint32_t parent = (bin - 1) >> 3, first = (parent << 3) + 1;
            if (bin > first) --bin; else bin = parent;
        }
    }
    uint64_t best = UINT64_MAX;
    int64_t e = end0 - 1;
    for (int lvl = 0, s = ms + 3 * dp; lvl <= dp; lvl++, s -= 3) {
        const uint32_t t = first_of(lvl);
        for (uint32_t k = t + (uint32_t)(beg0 >> s); k <= t + (uint32_t)(e >> s); k++) {
            auto it = bins.find(k);
            if (it == bins.end()) continue;
            for (const auto &c : it->second.chunks)
                if (c.second > min_off && c.first < best) best = c.first;
        }
    }
    if (best == UINT64_MAX) { *empty = true; return 0; }
    return best;
}

}   // namespace

extern "C" {

// Whole BGZF file -> bytes, through the reader the BAM path uses (block walk, inflate, CRC-32 and ISIZE checks).
int nc_bgzf_read_file(const char *path, uint8_t *out, int64_t cap, int64_t *n_out)
{
    if (!path || !n_out || cap < 0 || (cap && !out)) return NC_ERR_ARG;
    *n_out = 0;
    Bgzf z;
    z.f = fopen(path, "rb");
    if (!z.f) return NC_ERR_ARG;
    int rc = NC_OK;
    int64_t total = 0;
    if (!z.load_block(0)) rc = NC_ERR_ARG;
    while (rc == NC_OK && !z.eof) {
        const int64_t k = (int64_t)z.block.size();
        if (total + k <= cap && k) memcpy(out + total, z.block.data(), (size_t)k);
        total += k;
        if (!z.load_block(z.next_coff)) rc = NC_ERR_ARG;
    }
    fclose(z.f);
    *n_out = total;
    if (rc == NC_OK && total > cap) rc = NC_ERR_CAPACITY;
    return rc;
}

int nc_bam_open(const char *path, nc_bam **out)
{
    if (!path || !out) return NC_ERR_ARG;
    *out = nullptr;
    nc_bam *b = new nc_bam();
    b->path = path;
    b->z.f = fopen(path, "rb");
    if (!b->z.f) { delete b; return NC_ERR_ARG; }
    bool err = false;
    uint8_t h[12];
    auto fail = [&]() { fclose(b->z.f); delete b; return NC_ERR_ARG; };
    if (!b->z.load_block(0) || !b->z.read(h, 8, &err) || memcmp(h, "BAM\1", 4) != 0) return fail();
    const int32_t l_text = rd32(h + 4);
    std::vector<uint8_t> text((size_t)std::max(0, l_text));
    if (l_text > 0 && !b->z.read(text.data(), (size_t)l_text, &err)) return fail();
    if (!b->z.read(h, 4, &err)) return fail();
    const int32_t n_ref = rd32(h);
    for (int32_t r = 0; r < n_ref; r++) {
        if (!b->z.read(h, 4, &err)) return fail();
        const int32_t l_name = rd32(h);
        std::vector<char> nm((size_t)l_name);
        if (!b->z.read(nm.data(), (size_t)l_name, &err) || !b->z.read(h, 4, &err)) return fail();
        b->ref_name.emplace_back(nm.data());
        b->ref_len.push_back(rd32(h));
    }
    b->first_rec_voff = ((uint64_t)b->z.block_coff << 16) | (uint64_t)b->z.upos;
    if (b->z.upos >= b->z.block.size() && !b->z.eof) b->first_rec_voff = (uint64_t)b->z.next_coff << 16;
    b->have_bai = load_bai(b);
    if (!b->have_bai) b->have_csi = load_csi(b);
    *out = b;
    return NC_OK;
}

int nc_bam_close(nc_bam *b)
{
    if (!b) return NC_OK;
    if (b->z.f) fclose(b->z.f);
    delete b;
    return NC_OK;
}

int nc_bam_n_refs(nc_bam *b, int32_t *n, int32_t *has_index)
{
    if (!b || !n) return NC_ERR_ARG;
    *n = (int32_t)b->ref_name.size();
    if (has_index) *has_index = (b->have_bai || b->have_csi) ? 1 : 0;
    return NC_OK;
}

int nc_bam_ref(nc_bam *b, int32_t i, const char **name, int32_t *len)
{
    if (!b || i < 0 || i >= (int32_t)b->ref_name.size()) return NC_ERR_ARG;
    if (name) *name = b->ref_name[(size_t)i].c_str();
    if (len) *len = b->ref_len[(size_t)i];
    return NC_OK;
}

const char *nc_bam_error(const nc_bam *b) { return b ? b->err : "null handle"; }

// Decodes every mapped alignment of reference `tid` that overlaps [beg1, end1] (1-based, inclusive), in file
// (coordinate) order.  The BAM must be coordinate-sorted; with a .bai next to it the scan starts at the linear-index
// offset of beg1, otherwise at the first record.
int nc_bam_decode(nc_bam *b, int32_t tid, int32_t beg1, int32_t end1, int32_t keep_seq, nc_decoded **out)
{
    if (!b || !out || tid < 0 || tid >= (int32_t)b->ref_name.size() || end1 < beg1) return NC_ERR_ARG;
    *out = nullptr;
    uint64_t voff = b->first_rec_voff;
    if (b->have_bai && (size_t)tid < b->lin.size()) {
        const auto &L = b->lin[(size_t)tid];
        const size_t w = (size_t)std::max(0, beg1 - 1) >> 14;
        if (w < L.size() && L[w]) voff = L[w];
        else if (!L.empty() && w >= L.size()) { if (L.back()) voff = L.back(); }
    }
    bool nothing = false;
    if (!b->have_bai && b->have_csi && (size_t)tid < b->csi.size()) voff = csi_start(b, tid, (int64_t)beg1 - 1, end1, &nothing);
    if (!nothing && !b->z.seek(voff)) return bam_fail(b, NC_ERR_ARG, "BGZF seek failed");
    nc_decoded *d = new nc_decoded();
    // Address space for ~48x coverage of the interval up front (untouched pages cost nothing): a growing gigabyte vector
    // is re-mapped and copied again and again, and with many decoding threads those mmap / munmap calls serialise on the
    // process's address-space lock.  Deeper data simply grows the vectors as usual.
    try {
        const size_t span = (size_t)(end1 - beg1) + 1;
        d->codes.reserve(span * 48 + (1u << 20));
        if (keep_seq) d->seq.reserve(span * 48 + (1u << 20));
        const size_t nr = span / 64 + 1024;
        d->start.reserve(nr); d->end.reserve(nr); d->flag.reserve(nr); d->qstart.reserve(nr); d->ps.reserve(nr); d->hap.reserve(nr);
        d->off.reserve(nr + 1); d->ev_off.reserve(nr + 1); d->seq_off.reserve(nr + 1); d->name_off.reserve(nr + 1);
        d->ev_pos.reserve(span / 2 + 4096); d->ev_len.reserve(span / 2 + 4096);
        d->names.reserve(nr * 40);
    } catch (const std::bad_alloc &) {
    }
    d->off.push_back(0);
    d->ev_off.push_back(0);
    d->seq_off.push_back(0);
    d->name_off.push_back(0);
    std::vector<uint8_t> rec, qc;                    // the record; its bases as codes (SEQ_CODES)
    bool err = false;
    const int32_t beg0 = beg1 - 1, end0 = end1;      // 0-based half-open
    for (; !nothing;) {
        uint8_t lb[4];
        if (!b->z.read(lb, 4, &err)) break;
        const int32_t bs = rd32(lb);
        if (bs < 32) { err = true; break; }
        rec.resize((size_t)bs);
        if (!b->z.read(rec.data(), (size_t)bs, &err)) { err = true; break; }
        const uint8_t *p = rec.data();
        const int32_t refid = rd32(p), pos = rd32(p + 4);
        const int l_name = p[8];
        const int n_cig = p[12] | (p[13] << 8), flag = p[14] | (p[15] << 8);
        const int32_t l_seq = rd32(p + 16);
        if (refid < 0 || refid > tid) { if (refid > tid) break; else continue; }
        if (refid < tid) continue;
        if (pos >= end0) break;                       // sorted: nothing further can overlap
        if (flag & 0x4) continue;
        if (l_seq < 0) { err = true; break; }
        const uint8_t *name = p + 32, *cig = name + l_name, *seq = cig + 4 * (size_t)n_cig, *qual = seq + ((size_t)l_seq + 1) / 2;
        const uint8_t *aux = qual + l_seq, *aux_end = p + bs;
        if (aux > aux_end) { err = true; break; }
        int64_t n_cig_real = n_cig;
        // SAMv1 4.2.2: a CIGAR of more than 65535 operations (ultra-long reads) is stored in the CG:B,I tag and the record
        // carries the placeholder <l_seq>S<ref_len>N -- htslib swaps the real one in transparently, and so does this reader.
        if (n_cig == 2 && (rdu32(cig) & 15) == 4 && (int64_t)(rdu32(cig) >> 4) == l_seq && (rdu32(cig + 4) & 15) == 3) {
            for (const uint8_t *a = aux; a + 3 <= aux_end;) {
                const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
                a += 3;
                size_t adv = 0;
                switch (ty) {
                case 'c': case 'C': case 'A': adv = 1; break;
                case 's': case 'S': adv = 2; break;
                case 'i': case 'I': case 'f': adv = 4; break;
                case 'Z': case 'H': { const uint8_t *z = a; while (z < aux_end && *z) z++; adv = (size_t)(z - a) + 1; break; }
                case 'B': {
                    if (a + 5 > aux_end) { adv = (size_t)(aux_end - a); break; }
                    const char st = (char)a[0];
                    const uint32_t cnt = rdu32(a + 1);
                    const int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                    if (t0 == 'C' && t1 == 'G' && st == 'I' && a + 5 + (size_t)cnt * 4 <= aux_end) { cig = a + 5; n_cig_real = cnt; }
                    adv = 5 + (size_t)cnt * es;
                    break; }
                default: adv = (size_t)(aux_end - a); break;
                }
                if (adv > (size_t)(aux_end - a)) break;
                a += adv;
            }
        }
        // reference span and query length of the CIGAR
        int64_t rlen64 = 0, qlen = 0;
        for (int64_t k = 0; k < n_cig_real; k++) {
            const uint32_t c = rdu32(cig + 4 * k);
            const int op = c & 15, len = (int)(c >> 4);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen64 += len;
            if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += len;
        }
        if (rlen64 <= 0 || rlen64 > INT32_MAX - pos - 2 || pos + rlen64 <= beg0) continue;
        const int32_t rlen = (int32_t)rlen64;
        // A record without its bases (SEQ '*': l_seq 0, what minimap2 writes for secondary alignments) or with fewer bases
        // than its CIGAR consumes must not be read past its end: its aligned positions decode as 'N' (code 4).
        const bool has_seq = qlen <= (int64_t)l_seq;
        if (has_seq || keep_seq) {
            if (qc.size() < (size_t)l_seq + 34) qc.resize((size_t)l_seq + 34 + (size_t)l_seq / 4);
            SEQ_CODES(seq, l_seq, qc.data());
        }
        // codes + indel markers
        const size_t c0 = d->codes.size();
        d->codes.resize(c0 + (size_t)rlen);
        uint8_t *co = d->codes.data() + c0;
        int32_t rp = 0, qp = 0, q_first = -1;
        bool has_refskip = false;
        for (int64_t k = 0; k < n_cig_real; k++) {
            const uint32_t c = rdu32(cig + 4 * k);
            const int op = c & 15, len = (int)(c >> 4);
            switch (op) {
            case 0: case 7: case 8:                                   // M, =, X
                if (q_first < 0) q_first = qp;                        // query index of the first aligned base (leading S / I skipped)
                if (has_seq) memcpy(co + rp, qc.data() + qp, (size_t)len);
                else memset(co + rp, 4, (size_t)len);
                rp += len;
                qp += len;
                break;
            case 1:                                                   // I: '+n' on the previous reference column
                if (rp > 0) { d->ev_pos.push_back(pos + rp); d->ev_len.push_back(len); }
                qp += len;
                break;
            case 2:                                                   // D: '-n' on the previous column, then '*' columns
                if (rp > 0) { d->ev_pos.push_back(pos + rp); d->ev_len.push_back(-len); }
                for (int i = 0; i < len; i++, rp++) co[rp] = 4;
                break;
            case 3:                                                   // N (reference skip): the reference raises KeyError (E10); coded 4 here
                for (int i = 0; i < len; i++, rp++) co[rp] = 4;           // and the read is marked (NC_FLAG_REFSKIP): nc_decoded_check
                has_refskip = true;
                break;
            case 4: qp += len; break;                                 // S
            default: break;                                           // H, P
            }
        }
        d->start.push_back(pos + 1);
        d->end.push_back(pos + 1 + rlen);
        d->flag.push_back(flag | (has_refskip ? NC_FLAG_REFSKIP : 0));
        d->qstart.push_back(q_first < 0 ? qp : q_first);
        d->off.push_back((int64_t)d->codes.size());
        d->ev_off.push_back((int32_t)d->ev_pos.size());
        // tags HP / PS
        int hp = 0, ps = 0;
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
            case 'i': iv = rd32(a); a += 4; break;
            case 'I': iv = rdu32(a); a += 4; break;
            case 'A': a += 1; is_int = false; break;
            case 'f': a += 4; is_int = false; break;
            case 'Z': case 'H': while (a < aux_end && *a) a++; a++; is_int = false; break;
            case 'B': {
                const char st = (char)a[0];
                const uint32_t cnt = rdu32(a + 1);
                const int es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                a += 5 + (size_t)cnt * es;
                is_int = false;
                break; }
            default: a = aux_end; is_int = false; break;
            }
            if (is_int && t0 == 'H' && t1 == 'P') hp = (int)iv;
            if (is_int && t0 == 'P' && t1 == 'S') ps = (int)iv;
        }
        d->hap.push_back((uint8_t)((hp == 1 || hp == 2) ? hp : 0));
        d->ps.push_back(ps);
        d->names.insert(d->names.end(), (const char *)name, (const char *)name + l_name);   // includes the NUL
        d->name_off.push_back((int32_t)d->names.size());
        if (keep_seq) {
            const size_t s0 = d->seq.size();
            d->seq.resize(s0 + (size_t)l_seq);
            if (l_seq > 0) memcpy(d->seq.data() + s0, qc.data(), (size_t)l_seq);
        }
        d->seq_off.push_back((int64_t)d->seq.size());
    }
    if (err) { delete d; return bam_fail(b, NC_ERR_ARG, "truncated or corrupt BAM record / BGZF block"); }
    *out = d;
    return NC_OK;
}

// The same over `n_regions` equal sub-intervals at once: every host thread opens its own handle, seeks through the .bai
// linear index and decodes the alignments that START in its sub-interval (the first one also takes those that merely
// overlap its left edge); the parts are then copied -- again one thread per part, so that the pages of the merged arrays
// are faulted in in parallel -- into one nc_decoded identical to what a single nc_bam_decode call returns.
int nc_bam_decode_regions(const char *path, int32_t tid, int32_t beg1, int32_t end1, int32_t keep_seq, int32_t n_regions,
                          nc_decoded **out)
{
    if (!path || !out || end1 < beg1 || n_regions < 1) return NC_ERR_ARG;
    *out = nullptr;
    const int R = n_regions;
    const bool dbg = getenv("NC_BAM_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    std::vector<int32_t> edge((size_t)R + 1);
    for (int k = 0; k <= R; k++) edge[(size_t)k] = (int32_t)(beg1 + ((int64_t)end1 - beg1 + 1) * k / R);
    std::vector<nc_decoded *> part((size_t)R, nullptr);
    std::vector<int> rc((size_t)R, NC_OK);
    std::vector<int32_t> cut((size_t)R, 0);                       // first read of part k that starts inside its sub-interval
    {
        std::vector<std::thread> th;
        for (int k = 0; k < R; k++)
            th.emplace_back([&, k]() {
                if (edge[(size_t)k + 1] <= edge[(size_t)k]) return;            // empty sub-interval
                nc_bam *b = nullptr;
                const double t0 = now();
                rc[(size_t)k] = nc_bam_open(path, &b);
                if (rc[(size_t)k] != NC_OK) return;
                const double t1 = now();
                b->z.max_threads = 1;                                           // the regions are the parallelism
                b->z.grow_cap = 32;                                             // ... and short: at most 2 MB inflated past a region's end (up to 64 MB of a 1024-block window before)
                rc[(size_t)k] = nc_bam_decode(b, tid, edge[(size_t)k], edge[(size_t)k + 1] - 1, keep_seq, &part[(size_t)k]);
                const double t2 = now();
                nc_bam_close(b);
                if (dbg && (k < 8 || k == R - 1)) fprintf(stderr, "  region %d: start +%.3f open %.3f decode %.3f close %.3f\n", k, t0 - t_a, t1 - t0, t2 - t1, now() - t2);
                if (rc[(size_t)k] == NC_OK && k > 0) {
                    const auto &st = part[(size_t)k]->start;
                    cut[(size_t)k] = (int32_t)(std::lower_bound(st.begin(), st.end(), edge[(size_t)k]) - st.begin());
                }
            });
        for (auto &t : th) t.join();
    }
    const double t_b = now();
    int err = NC_OK;
    for (int k = 0; k < R; k++) if (rc[(size_t)k] != NC_OK) err = rc[(size_t)k];
    if (err != NC_OK) { for (auto *d : part) delete d; return err; }
    // offsets of every part in the merged arrays
    std::vector<int64_t> r0((size_t)R + 1, 0), c0((size_t)R + 1, 0), e0((size_t)R + 1, 0), s0((size_t)R + 1, 0), n0((size_t)R + 1, 0);
    for (int k = 0; k < R; k++) {
        const nc_decoded *d = part[(size_t)k];
        int64_t nr = 0, nc = 0, ne = 0, ns = 0, nn = 0;
        if (d) {
            const size_t c = (size_t)cut[(size_t)k], n = d->start.size();
            nr = (int64_t)(n - c);
            nc = d->off[n] - d->off[c];
            ne = d->ev_off[n] - d->ev_off[c];
            ns = d->seq_off[n] - d->seq_off[c];
            nn = d->name_off[n] - d->name_off[c];
        }
        r0[(size_t)k + 1] = r0[(size_t)k] + nr; c0[(size_t)k + 1] = c0[(size_t)k] + nc; e0[(size_t)k + 1] = e0[(size_t)k] + ne;
        s0[(size_t)k + 1] = s0[(size_t)k] + ns; n0[(size_t)k + 1] = n0[(size_t)k] + nn;
    }
    if (e0[(size_t)R] > INT32_MAX || n0[(size_t)R] > INT32_MAX) { for (auto *d : part) delete d; return NC_ERR_CAPACITY; }
    nc_decoded *m = new nc_decoded();
    const size_t NR = (size_t)r0[(size_t)R];
    m->start.resize(NR); m->end.resize(NR); m->flag.resize(NR); m->ps.resize(NR); m->qstart.resize(NR); m->hap.resize(NR);
    m->off.resize(NR + 1); m->ev_off.resize(NR + 1); m->seq_off.resize(NR + 1); m->name_off.resize(NR + 1);
    m->codes.resize((size_t)c0[(size_t)R]); m->ev_pos.resize((size_t)e0[(size_t)R]); m->ev_len.resize((size_t)e0[(size_t)R]);
    m->seq.resize((size_t)s0[(size_t)R]); m->names.resize((size_t)n0[(size_t)R]);
    m->off[0] = 0; m->ev_off[0] = 0; m->seq_off[0] = 0; m->name_off[0] = 0;
    const double t_c = now();
    {
        std::vector<std::thread> th;
        for (int k = 0; k < R; k++)
            th.emplace_back([&, k]() {
                nc_decoded *d = part[(size_t)k];
                if (!d) return;
                const size_t c = (size_t)cut[(size_t)k], n = d->start.size(), nr = n - c, r = (size_t)r0[(size_t)k];
                auto cp = [](void *dst, const void *src, size_t bytes) { if (bytes) memcpy(dst, src, bytes); };
                cp(m->start.data() + r, d->start.data() + c, nr * 4); cp(m->end.data() + r, d->end.data() + c, nr * 4);
                cp(m->flag.data() + r, d->flag.data() + c, nr * 4); cp(m->ps.data() + r, d->ps.data() + c, nr * 4);
                cp(m->qstart.data() + r, d->qstart.data() + c, nr * 4); cp(m->hap.data() + r, d->hap.data() + c, nr);
                const int64_t dc = c0[(size_t)k] - d->off[c], ds = s0[(size_t)k] - d->seq_off[c];
                const int32_t de = (int32_t)(e0[(size_t)k] - d->ev_off[c]), dn = (int32_t)(n0[(size_t)k] - d->name_off[c]);
                for (size_t i = 1; i <= nr; i++) {
                    m->off[r + i] = d->off[c + i] + dc;
                    m->seq_off[r + i] = d->seq_off[c + i] + ds;
                    m->ev_off[r + i] = d->ev_off[c + i] + de;
                    m->name_off[r + i] = d->name_off[c + i] + dn;
                }
                cp(m->codes.data() + c0[(size_t)k], d->codes.data() + d->off[c], (size_t)(d->off[n] - d->off[c]));
                cp(m->ev_pos.data() + e0[(size_t)k], d->ev_pos.data() + d->ev_off[c], (size_t)(d->ev_off[n] - d->ev_off[c]) * 4);
                cp(m->ev_len.data() + e0[(size_t)k], d->ev_len.data() + d->ev_off[c], (size_t)(d->ev_off[n] - d->ev_off[c]) * 4);
                cp(m->seq.data() + s0[(size_t)k], d->seq.data() + d->seq_off[c], (size_t)(d->seq_off[n] - d->seq_off[c]));
                cp(m->names.data() + n0[(size_t)k], d->names.data() + d->name_off[c], (size_t)(d->name_off[n] - d->name_off[c]));
                delete d;                                                       // frees this part's pages in parallel as well
                part[(size_t)k] = nullptr;
            });
        for (auto &t : th) t.join();
    }
    if (dbg) fprintf(stderr, "nc_bam_decode_regions: %d regions: decode %.3f s, alloc %.3f s, merge %.3f s\n", R, t_b - t_a, t_c - t_b, now() - t_c);
    *out = m;
    return NC_OK;
}

int nc_bam_set_threads(nc_bam *b, int32_t n)
{
    if (!b || n < 0) return NC_ERR_ARG;
    b->z.max_threads = n;
    return NC_OK;
}

int nc_decoded_view(const nc_decoded *d, nc_decoded_arrays *v)
{
    if (!d || !v) return NC_ERR_ARG;
    v->n_reads = (int32_t)d->start.size();
    v->start = d->start.data(); v->end = d->end.data(); v->flag = d->flag.data();
    v->off = d->off.data(); v->codes = d->codes.data(); v->n_codes = (int64_t)d->codes.size();
    v->ev_off = d->ev_off.data(); v->ev_pos = d->ev_pos.data(); v->ev_len = d->ev_len.data();
    v->n_events = (int64_t)d->ev_pos.size();
    v->hap = d->hap.data(); v->ps = d->ps.data();
    v->seq_off = d->seq_off.data(); v->seq = d->seq.data(); v->n_seq = (int64_t)d->seq.size();
    v->name_off = d->name_off.data(); v->names = d->names.data();
    v->qstart = d->qstart.data();
    return NC_OK;
}

int nc_decoded_free(nc_decoded *d)
{
    delete d;
    return NC_OK;
}


// ---------------------------------------------------------------------------------- a11: pass-2 read windows
struct nc_slices {
    std::vector<int32_t> anchor_off, read_idx;
    std::vector<int64_t> seq_off;
    std::vector<uint8_t> seq;
};

// pysam's PileupRead.query_position_or_next at 1-based reference position p (start <= p < end): the query index aligned
// to p, or -- when p lies in a deletion -- the index of the next aligned query base.  Events are the '+n' / '-n' markers
// on the column BEFORE the insertion / deletion, in reference order; qstart = query index of the first aligned base.
static int32_t qpos_or_next(int32_t start, int32_t qstart, const int32_t *ev_pos, const int32_t *ev_len, int32_t n_ev, int32_t p)
{
    int32_t q = qstart, r = start;
    for (int32_t e = 0; e < n_ev; e++) {
        const int32_t ep = ev_pos[e], el = ev_len[e];
        if (p <= ep) return q + (p - r);
        q += ep - r + 1;
        r = ep + 1;
        if (el > 0) q += el;                          // insertion: query bases with no reference column
        else {
            if (p <= ep - el) return q;               // inside the deletion: next aligned base
            r += -el;
        }
    }
    return q + (p - r);
}

int nc_indel_slices(const nc_decoded *d, int32_t n_anchor, const int32_t *anchor_pos, int32_t window_before, int32_t window_after,
                    const uint8_t *keep, nc_slices **out)
{
    if (!d || !out || n_anchor < 0 || (n_anchor && !anchor_pos) || window_before < 0 || window_after < 0) return NC_ERR_ARG;
    const int32_t n = (int32_t)d->start.size();
    if (n && d->seq.empty()) return NC_ERR_STATE;                          // decoded without keep_seq
    nc_slices *s = new (std::nothrow) nc_slices();
    if (!s) return NC_ERR_NOMEM;
    s->anchor_off.push_back(0);
    s->seq_off.push_back(0);
    // anchors ascending and reads coordinate-sorted: `first` = first read that can still cover an anchor >= p
    int32_t first = 0;
    for (int32_t a = 0; a < n_anchor; a++) {
        const int32_t p = anchor_pos[a];
        if (a && p < anchor_pos[a - 1]) first = 0;                         // not ascending: restart
        while (first < n && d->end[first] <= p) first++;
        for (int32_t r = first; r < n; r++) {
            if (d->start[r] > p) break;                                    // coordinate order
            if (d->end[r] <= p) continue;
            if (keep && !keep[r]) continue;
            const int32_t e0 = d->ev_off[r], e1 = d->ev_off[r + 1];
            const int32_t q = qpos_or_next(d->start[r], d->qstart[r], d->ev_pos.data() + e0, d->ev_len.data() + e0, e1 - e0, p);
            const int64_t s0 = d->seq_off[r], s1 = d->seq_off[r + 1], L = s1 - s0;
            int64_t a0 = (int64_t)q - window_before, a1 = (int64_t)q + window_after;
            if (a0 < 0) a0 = 0;
            if (a0 > L) a0 = L;                                            // a record without bases (SEQ '*')
            if (a1 > L) a1 = L;
            if (a1 < a0) a1 = a0;
            s->read_idx.push_back(r);
            s->seq.insert(s->seq.end(), d->seq.begin() + (s0 + a0), d->seq.begin() + (s0 + a1));
            s->seq_off.push_back((int64_t)s->seq.size());
        }
        s->anchor_off.push_back((int32_t)s->read_idx.size());
    }
    *out = s;
    return NC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Pass 2 of get_indel_testing_candidates (generate_indel_pileups.py:306-348, haploid :243-262) up to the aligner call, for
// ALL anchors of a chunk at once and without a Python object per read: the reference window of every anchor (skipped when it
// holds anything but upper-case AGTC, :325-327), the read windows (:329-331, as nc_indel_slices), the split into hap0 / hap1 /
// all reads by HP tag or by the imputed read collections (:310-318,333-337), the down-sampling to maxcov (first maxcov reads
// in pileup order: the deterministic stand-in for the unseeded random.sample of :19-20), the mincov tests (:48, :345), and the
// flat arrays nc_star_msa_tensor takes.
struct nc_pass2 {
    bigvec<int32_t> anchor_idx, first0, set_read0, read_off, ref_off, al_dup;
    bigvec<char> reads, refs;                                              // (no zero fill when the merged arrays are sized)
    int32_t sets_per_anchor = 3, max_cols = 1;
};

int nc_indel_pass2_sets(const nc_decoded *d, const uint8_t *keep, int32_t n_anchor, const int32_t *anchors, const char *contig,
                        int64_t chrom_len, int32_t ref_lo, int32_t ref_hi, int32_t window_after, int32_t mincov, int32_t maxcov,
                        int32_t haploid, const int32_t *imp_idx, const int32_t *imp_off, const int32_t *imp_reads, nc_pass2 **out)
{
    if (!d || !out || n_anchor < 0 || (n_anchor && (!anchors || !contig)) || window_after < 1 || maxcov < 1 || chrom_len < 0) return NC_ERR_ARG;
    const int32_t n = (int32_t)d->start.size();
    if (n && d->seq.empty()) return NC_ERR_STATE;                          // decoded without keep_seq
    nc_pass2 *o = new (std::nothrow) nc_pass2();
    if (!o) return NC_ERR_NOMEM;
    const int S = haploid ? 1 : 3;
    o->sets_per_anchor = S;
    static const char LET[8] = {'A', 'G', 'T', 'C', 'N', 'N', 'N', 'N'};
    // anchors [a_lo, a_hi) into `po` (offsets local to po); anchors are independent: contiguous ranges go to worker threads and
    // the partial results are concatenated in anchor order
    auto run = [&](int32_t a_lo, int32_t a_hi, nc_pass2 *po) -> int {
        po->set_read0.push_back(0);
        po->read_off.push_back(0);
        po->ref_off.push_back(0);
        // address space for the usual case (~60 windows per anchor): the vector does not move while it grows, untouched pages cost nothing
        try { po->reads.reserve(std::min<size_t>((size_t)(a_hi - a_lo) * 64 * (size_t)window_after, (size_t)1 << 30)); } catch (const std::bad_alloc &) {}
        int32_t first = 0;
        std::vector<int32_t> cov;                                          // reads in the pileup at the anchor
        std::vector<int32_t> side0, side1;
        std::vector<int32_t> sets[3];
        try {
            for (int32_t a = a_lo; a < a_hi; a++) {
                const int32_t p = anchors[a];
                if (a > a_lo && p < anchors[a - 1]) first = 0;
                while (first < n && d->end[first] <= p) first++;
                // reference window [p, min(chrom_len, p + window_after + 1)) from the bases kept for the chunk ([ref_lo, ref_hi])
                const int64_t b = std::min<int64_t>(chrom_len, (int64_t)p + window_after + 1);
                bool ok = b > p && p >= ref_lo && b - 1 <= ref_hi && p >= 1;
                for (int64_t x = p; ok && x < b; x++) {
                    const char c = contig[x - 1];
                    ok = c == 'A' || c == 'G' || c == 'T' || c == 'C';
                }
                if (!ok) continue;
                cov.clear();
                for (int32_t r = first; r < n; r++) {
                    if (d->start[r] > p) break;
                    if (d->end[r] <= p) continue;
                    if (keep && !keep[r]) continue;
                    cov.push_back(r);
                }
                for (auto &v : sets) v.clear();
                const int k = imp_idx ? imp_idx[a] : -1;
                if (k >= 0) {
                    side0.assign(imp_reads + imp_off[2 * k], imp_reads + imp_off[2 * k + 1]);
                    side1.assign(imp_reads + imp_off[2 * k + 1], imp_reads + imp_off[2 * k + 2]);
                    std::sort(side0.begin(), side0.end());
                    std::sort(side1.begin(), side1.end());
                }
                for (int32_t r : cov) {
                    if (!haploid) {
                        int h;
                        if (k >= 0) h = std::binary_search(side0.begin(), side0.end(), r) ? 1 : std::binary_search(side1.begin(), side1.end(), r) ? 2 : 0;
                        else h = d->hap[r];
                        if (h == 1) sets[0].push_back(r);
                        else if (h == 2) sets[1].push_back(r);
                        sets[2].push_back(r);
                    } else sets[0].push_back(r);
                }
                bool pass = true;
                for (int t = 0; t < S && pass; t++) {
                    if ((int32_t)sets[t].size() > maxcov) sets[t].resize((size_t)maxcov);
                    const int32_t need = haploid ? mincov : (t < 2 ? 2 : mincov);
                    pass = (int32_t)sets[t].size() >= need;
                }
                if (!pass) continue;
                po->anchor_idx.push_back(a);
                po->first0.push_back(sets[0].empty() ? -1 : sets[0][0]);
                const int32_t al0 = (int32_t)po->read_off.size() - 1;                 // the anchor's first alignment
                for (int t = 0; t < S; t++) {
                    int64_t cols = b - p;
                    for (int32_t r : sets[t]) {
                        // a read of the "all reads" set that is in a haplotype set too: same window, same reference -> same alignment
                        int32_t same = -1;
                        if (t == 2) {
                            auto i0 = std::lower_bound(sets[0].begin(), sets[0].end(), r);
                            if (i0 != sets[0].end() && *i0 == r) same = al0 + (int32_t)(i0 - sets[0].begin());
                            else {
                                auto i1 = std::lower_bound(sets[1].begin(), sets[1].end(), r);
                                if (i1 != sets[1].end() && *i1 == r) same = al0 + (int32_t)sets[0].size() + (int32_t)(i1 - sets[1].begin());
                            }
                        }
                        po->al_dup.push_back(same);
                        const int32_t e0 = d->ev_off[r], e1 = d->ev_off[r + 1];
                        const int32_t q = qpos_or_next(d->start[r], d->qstart[r], d->ev_pos.data() + e0, d->ev_len.data() + e0, e1 - e0, p);
                        const int64_t s0 = d->seq_off[r], L = d->seq_off[r + 1] - s0;
                        int64_t a0 = std::min<int64_t>(std::max<int64_t>(q, 0), L), a1 = std::min<int64_t>((int64_t)q + window_after, L);
                        if (a1 < a0) a1 = a0;
                        const size_t w0 = po->reads.size();
                        po->reads.resize(w0 + (size_t)(a1 - a0));
                        const uint8_t *src = d->seq.data() + s0 + a0;
                        char *dst = po->reads.data() + w0;
                        for (int64_t x = 0; x < a1 - a0; x++) dst[x] = LET[src[x] & 7];
                        po->read_off.push_back((int32_t)po->reads.size());
                        cols += a1 - a0;
                    }
                    po->set_read0.push_back((int32_t)po->read_off.size() - 1);
                    po->refs.insert(po->refs.end(), contig + (p - 1), contig + (b - 1));
                    po->ref_off.push_back((int32_t)po->refs.size());
                    po->max_cols = (int32_t)std::max<int64_t>(po->max_cols, cols);     // every read base can add at most one column
                }
                if (po->reads.size() > ((size_t)1 << 30) || po->refs.size() > ((size_t)1 << 30)) return NC_ERR_CAPACITY;
            }
        } catch (const std::bad_alloc &) {
            return NC_ERR_NOMEM;
        }
        return NC_OK;
    };
    int T = std::min(nc_host_cpus(), 16);
    if (T > n_anchor / 64) T = n_anchor / 64;
    if (T <= 1) {
        const int rc = run(0, n_anchor, o);
        if (rc != NC_OK) { delete o; return rc; }
    } else {
        std::vector<nc_pass2> part((size_t)T);
        std::vector<int> rcs((size_t)T, NC_OK);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] { rcs[(size_t)t] = run((int32_t)((int64_t)n_anchor * t / T), (int32_t)((int64_t)n_anchor * (t + 1) / T), &part[(size_t)t]); });
        for (auto &x : th) x.join();
        for (int rc : rcs)
            if (rc != NC_OK) { delete o; return rc; }
        // the merged arrays are sized once and every worker copies its own part to its place (the read windows of a chromosome
        // arm are ~100 MB: one thread appending them costs more than assembling them did)
        try {
            std::vector<size_t> b_k((size_t)T + 1, 0), b_s((size_t)T + 1, 0), b_al((size_t)T + 1, 0), b_rb((size_t)T + 1, 0), b_fb((size_t)T + 1, 0);
            for (int t = 0; t < T; t++) {
                const nc_pass2 &pp = part[(size_t)t];
                b_k[(size_t)t + 1] = b_k[(size_t)t] + pp.anchor_idx.size();
                b_s[(size_t)t + 1] = b_s[(size_t)t] + pp.set_read0.size() - 1;
                b_al[(size_t)t + 1] = b_al[(size_t)t] + pp.read_off.size() - 1;
                b_rb[(size_t)t + 1] = b_rb[(size_t)t] + pp.reads.size();
                b_fb[(size_t)t + 1] = b_fb[(size_t)t] + pp.refs.size();
                o->max_cols = std::max(o->max_cols, pp.max_cols);
            }
            if (b_rb[(size_t)T] > ((size_t)1 << 30) || b_fb[(size_t)T] > ((size_t)1 << 30)) { delete o; return NC_ERR_CAPACITY; }
            o->anchor_idx.resize(b_k[(size_t)T]);
            o->first0.resize(b_k[(size_t)T]);
            o->set_read0.resize(b_s[(size_t)T] + 1);
            o->ref_off.resize(b_s[(size_t)T] + 1);
            o->read_off.resize(b_al[(size_t)T] + 1);
            o->al_dup.resize(b_al[(size_t)T]);
            o->reads.resize(b_rb[(size_t)T]);
            o->refs.resize(b_fb[(size_t)T]);
            o->set_read0[0] = 0;
            o->read_off[0] = 0;
            o->ref_off[0] = 0;
            std::vector<std::thread> tc;
            for (int t = 0; t < T; t++)
                tc.emplace_back([&, t] {
                    const nc_pass2 &pp = part[(size_t)t];
                    const size_t k0 = b_k[(size_t)t], s0 = b_s[(size_t)t], a0 = b_al[(size_t)t], r0 = b_rb[(size_t)t], f0 = b_fb[(size_t)t];
                    std::copy(pp.anchor_idx.begin(), pp.anchor_idx.end(), o->anchor_idx.begin() + (ptrdiff_t)k0);
                    std::copy(pp.first0.begin(), pp.first0.end(), o->first0.begin() + (ptrdiff_t)k0);
                    for (size_t k = 1; k < pp.set_read0.size(); k++) o->set_read0[s0 + k] = pp.set_read0[k] + (int32_t)a0;
                    for (size_t k = 1; k < pp.ref_off.size(); k++) o->ref_off[s0 + k] = pp.ref_off[k] + (int32_t)f0;
                    for (size_t k = 1; k < pp.read_off.size(); k++) o->read_off[a0 + k] = pp.read_off[k] + (int32_t)r0;
                    for (size_t k = 0; k < pp.al_dup.size(); k++) o->al_dup[a0 + k] = pp.al_dup[k] < 0 ? -1 : pp.al_dup[k] + (int32_t)a0;
                    if (!pp.reads.empty()) memcpy(o->reads.data() + r0, pp.reads.data(), pp.reads.size());
                    if (!pp.refs.empty()) memcpy(o->refs.data() + f0, pp.refs.data(), pp.refs.size());
                });
            for (auto &x : tc) x.join();
        } catch (const std::bad_alloc &) {
            delete o;
            return NC_ERR_NOMEM;
        }
    }
    *out = o;
    return NC_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Inputs this library does not reproduce, found instead of silently accepted:
//   * a kept alignment whose CIGAR holds a reference skip (N): the reference's code table raises KeyError on the '>' / '<'
//     pileup symbols (generate_SNP_pileups.py:104,175);
//   * two kept alignments with the same read name that overlap on the reference: the reference's per-column dicts are keyed by
//     name, so the later one replaces the earlier in a column's pileup (generate_SNP_pileups.py:175,185,208) while both count
//     in `n`; the read-major pack keeps them apart.
int nc_decoded_check(const nc_decoded *d, const uint8_t *keep, int64_t *n_refskip, int64_t *n_dup_overlap)
{
    if (!d) return NC_ERR_ARG;
    const int32_t n = (int32_t)d->start.size();
    int64_t nskip = 0, ndup = 0;
    std::vector<std::pair<uint64_t, int32_t>> h;
    try { h.reserve((size_t)n); } catch (const std::bad_alloc &) { return NC_ERR_NOMEM; }
    for (int32_t r = 0; r < n; r++) {
        if (keep && !keep[r]) continue;
        if (d->flag[r] & NC_FLAG_REFSKIP) nskip++;
        uint64_t x = 1469598103934665603ull;                                   // FNV-1a of the name
        for (const char *c = d->names.data() + d->name_off[r]; *c; c++) x = (x ^ (uint8_t)*c) * 1099511628211ull;
        h.emplace_back(x, r);
    }
    std::sort(h.begin(), h.end());
    for (size_t i = 0; i < h.size();) {
        size_t j = i + 1;
        while (j < h.size() && h[j].first == h[i].first) j++;
        for (size_t a = i; a < j; a++)
            for (size_t b = a + 1; b < j; b++) {
                const int32_t ra = h[a].second, rb = h[b].second;
                if (strcmp(d->names.data() + d->name_off[ra], d->names.data() + d->name_off[rb]) != 0) continue;
                if (d->start[ra] < d->end[rb] && d->start[rb] < d->end[ra]) ndup++;
            }
        i = j;
    }
    if (n_refskip) *n_refskip = nskip;
    if (n_dup_overlap) *n_dup_overlap = ndup;
    return (nskip || ndup) ? NC_ERR_UNSUPPORTED : NC_OK;
}

// Alignments that share a read name among the kept ones (a split read's primary + supplementary records; paired-end mates): gid[r] = index of
// the FIRST kept alignment with r's name when the name occurs on more than one kept alignment, else -1.  The reference keys a column's pileup,
// the strand table and the neighbour lookups by name (generate_SNP_pileups.py:141-143,175,185,223,232): pack.name_groups builds the
// featuriser's table from this (nc_snp_set_mates).
int nc_decoded_name_groups(const nc_decoded *d, const uint8_t *keep, int32_t *gid, int64_t *n_shared)
{
    if (!d || !gid) return NC_ERR_ARG;
    const int32_t n = (int32_t)d->start.size();
    std::vector<std::pair<uint64_t, int32_t>> h;
    try { h.reserve((size_t)n); } catch (const std::bad_alloc &) { return NC_ERR_NOMEM; }
    for (int32_t r = 0; r < n; r++) {
        gid[r] = -1;
        if (keep && !keep[r]) continue;
        uint64_t x = 1469598103934665603ull;                                   // FNV-1a of the name
        for (const char *c = d->names.data() + d->name_off[r]; *c; c++) x = (x ^ (uint8_t)*c) * 1099511628211ull;
        h.emplace_back(x, r);
    }
    std::sort(h.begin(), h.end());                                             // (hash, then index: members of a name ascend)
    int64_t shared = 0;
    for (size_t i = 0; i < h.size();) {
        size_t j = i + 1;
        while (j < h.size() && h[j].first == h[i].first) j++;
        for (size_t a = i; a < j; a++) {                                       // hash-equal runs are confirmed on the names themselves
            const int32_t ra = h[a].second;
            if (gid[ra] >= 0) continue;
            bool any = false;
            for (size_t b = a + 1; b < j; b++) {
                const int32_t rb = h[b].second;
                if (gid[rb] < 0 && strcmp(d->names.data() + d->name_off[ra], d->names.data() + d->name_off[rb]) == 0) { gid[rb] = ra; any = true; shared++; }
            }
            if (any) { gid[ra] = ra; shared++; }
        }
        i = j;
    }
    if (n_shared) *n_shared = shared;
    return NC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The per-read inputs of the DEVICE pass 2 (nc_indel_sites_*), for the kept reads in pack order.  The position-addressed
// codes in HBM hold every aligned base of a read; what a query window (query_sequence[q : q + window], generate_indel_pileups.py:
// 331) needs beyond them are the bases WITHOUT a reference column: the inserted bases of every insertion event and the
// query bases that follow the last aligned one (trailing soft clip), at most `tail_cap` of them.  Also the events / HP / PS
// of the kept reads (what pack.pack_reads gathers for nc_indel_scan).
struct nc_indel_pack_h {
    bigvec<int32_t> ev_off, ev_pos, ev_len, ins_off, tail_off, ps;
    bigvec<uint8_t> hap, ins_bases, tail_bases, rflag;
};

int nc_indel_pack_build(const nc_decoded *d, const uint8_t *keep, int32_t tail_cap, nc_indel_pack_h **out)
{
    if (!d || !out || tail_cap < 0) return NC_ERR_ARG;
    *out = nullptr;
    const int32_t n = (int32_t)d->start.size();
    const bool have_seq = !d->seq.empty();
    nc_indel_pack_h *o = new (std::nothrow) nc_indel_pack_h();
    if (!o) return NC_ERR_NOMEM;
    try {
        std::vector<int32_t> kept;
        kept.reserve((size_t)n);
        for (int32_t r = 0; r < n; r++)
            if (!keep || keep[r]) kept.push_back(r);
        const size_t K = kept.size();
        // pass 1: per read, the number of events, inserted bases and tail bases
        std::vector<int64_t> c_ev(K + 1, 0), c_ins(K + 1, 0), c_tail(K + 1, 0);
        int T = std::min(nc_host_cpus(), 16);
        if ((size_t)T > K / 1024 + 1) T = (int)(K / 1024 + 1);
        auto each = [&](auto fn) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] { for (size_t k = K * (size_t)t / (size_t)T; k < K * ((size_t)t + 1) / (size_t)T; k++) fn(k); });
            for (auto &x : th) x.join();
        };
        // the query walk of one read: calls ins(e, q0, len) for every insertion event and returns the query index that follows
        // the last aligned base
        auto walk = [&](int32_t r, auto on_ins) -> int64_t {
            int64_t q = d->qstart[r];
            int32_t rp = d->start[r];
            for (int32_t e = d->ev_off[r]; e < d->ev_off[r + 1]; e++) {
                const int32_t ep = d->ev_pos[e], el = d->ev_len[e];
                q += ep - rp + 1;
                rp = ep + 1;
                if (el > 0) { on_ins(e, q, el); q += el; }
                else rp += -el;
            }
            return q + (d->end[r] - rp);
        };
        each([&](size_t k) {
            const int32_t r = kept[k];
            c_ev[k + 1] = d->ev_off[r + 1] - d->ev_off[r];
            const int64_t L = have_seq ? d->seq_off[r + 1] - d->seq_off[r] : 0;
            int64_t ni = 0;
            const int64_t qe = walk(r, [&](int32_t, int64_t q0, int32_t len) { ni += std::max<int64_t>(0, std::min<int64_t>(L, q0 + len) - std::min<int64_t>(L, q0)); });
            c_ins[k + 1] = ni;
            c_tail[k + 1] = std::max<int64_t>(0, std::min<int64_t>(L, qe + tail_cap) - std::min<int64_t>(L, qe));
        });
        for (size_t k = 0; k < K; k++) { c_ev[k + 1] += c_ev[k]; c_ins[k + 1] += c_ins[k]; c_tail[k + 1] += c_tail[k]; }
        if (c_ev[K] > INT32_MAX - 1 || c_ins[K] > INT32_MAX - 1 || c_tail[K] > INT32_MAX - 1) { delete o; return NC_ERR_CAPACITY; }
        const size_t NE = (size_t)c_ev[K];
        o->ev_off.resize(K + 1); o->tail_off.resize(K + 1); o->ps.resize(K); o->hap.resize(K); o->rflag.resize(K);
        o->ev_pos.resize(NE); o->ev_len.resize(NE); o->ins_off.resize(NE + 1);
        o->ins_bases.resize((size_t)c_ins[K]); o->tail_bases.resize((size_t)c_tail[K]);
        o->ev_off[K] = (int32_t)NE;
        o->tail_off[K] = (int32_t)c_tail[K];
        o->ins_off[NE] = (int32_t)c_ins[K];
        each([&](size_t k) {
            const int32_t r = kept[k];
            const int32_t e0 = d->ev_off[r], ne = d->ev_off[r + 1] - e0, w0 = (int32_t)c_ev[k];
            o->ev_off[k] = w0;
            o->tail_off[k] = (int32_t)c_tail[k];
            o->ps[k] = d->ps[r];
            o->hap[k] = d->hap[r];
            const int64_t s0 = have_seq ? d->seq_off[r] : 0, L = have_seq ? d->seq_off[r + 1] - s0 : 0;
            o->rflag[k] = (uint8_t)(L == 0 ? 1 : 0);                              // bit 0: a record without bases (SEQ '*'): its windows are empty
            if (ne) {
                memcpy(o->ev_pos.data() + w0, d->ev_pos.data() + e0, (size_t)ne * 4);
                memcpy(o->ev_len.data() + w0, d->ev_len.data() + e0, (size_t)ne * 4);
            }
            int64_t io = c_ins[k];
            int32_t e_next = e0;
            const int64_t qe = walk(r, [&](int32_t e, int64_t q0, int32_t len) {
                for (; e_next <= e; e_next++) o->ins_off[(size_t)(w0 + (e_next - e0))] = (int32_t)io;      // deletions before it: empty ranges
                const int64_t a = std::min<int64_t>(L, q0), b = std::min<int64_t>(L, q0 + len);
                if (b > a) { memcpy(o->ins_bases.data() + io, d->seq.data() + s0 + a, (size_t)(b - a)); io += b - a; }
            });
            for (; e_next < e0 + ne; e_next++) o->ins_off[(size_t)(w0 + (e_next - e0))] = (int32_t)io;
            const int64_t a = std::min<int64_t>(L, qe), b = std::min<int64_t>(L, qe + tail_cap);
            if (b > a) memcpy(o->tail_bases.data() + c_tail[k], d->seq.data() + s0 + a, (size_t)(b - a));
        });
    } catch (const std::bad_alloc &) {
        delete o;
        return NC_ERR_NOMEM;
    }
    *out = o;
    return NC_OK;
}

int nc_indel_pack_view(const nc_indel_pack_h *o, nc_indel_pack_arrays *v)
{
    if (!o || !v) return NC_ERR_ARG;
    v->n_reads = (int32_t)o->ps.size();
    v->ev_off = o->ev_off.data(); v->ev_pos = o->ev_pos.data(); v->ev_len = o->ev_len.data();
    v->n_events = (int64_t)o->ev_pos.size();
    v->ins_off = o->ins_off.data(); v->ins_bases = o->ins_bases.data(); v->n_ins_bases = (int64_t)o->ins_bases.size();
    v->tail_off = o->tail_off.data(); v->tail_bases = o->tail_bases.data(); v->n_tail_bases = (int64_t)o->tail_bases.size();
    v->read_ps = o->ps.data(); v->read_hap = o->hap.data(); v->read_flag = o->rflag.data();
    return NC_OK;
}

int nc_indel_pack_free(nc_indel_pack_h *o)
{
    delete o;
    return NC_OK;
}

int nc_pass2_view(const nc_pass2 *o, nc_pass2_arrays *v)
{
    if (!o || !v) return NC_ERR_ARG;
    v->n_kept = (int32_t)o->anchor_idx.size();
    v->anchor_idx = o->anchor_idx.data();
    v->first0 = o->first0.data();
    v->sets_per_anchor = o->sets_per_anchor;
    v->n_sets = (int32_t)o->set_read0.size() - 1;
    v->set_read0 = o->set_read0.data();
    v->n_alignments = (int32_t)o->read_off.size() - 1;
    v->read_off = o->read_off.data();
    v->reads = o->reads.data();
    v->ref_off = o->ref_off.data();
    v->refs = o->refs.data();
    v->max_cols = o->max_cols;
    v->al_dup = o->al_dup.data();
    return NC_OK;
}

int nc_pass2_free(nc_pass2 *o)
{
    delete o;
    return NC_OK;
}

int nc_slices_view(const nc_slices *s, nc_slices_arrays *v)
{
    if (!s || !v) return NC_ERR_ARG;
    v->n_anchor = (int32_t)s->anchor_off.size() - 1;
    v->anchor_off = s->anchor_off.data();
    v->read_idx = s->read_idx.data();
    v->seq_off = s->seq_off.data();
    v->seq = s->seq.data();
    v->n_slices = (int64_t)s->read_idx.size();
    return NC_OK;
}

int nc_slices_free(nc_slices *s)
{
    delete s;
    return NC_OK;
}

}   // extern "C"
