// Synthetic indel workload generated in HBM (bench / test tooling, NOT part of the product path): a contig with planted
// heterozygous / homozygous indels and SNPs, and ONT-like reads over it -- position-addressed codes in the read pack's slot
// layout, the '+n' / '-n' events of every read, the inserted bases.  SURVEY.md 8d's generator for the indel configs ("per-read
// insertion / deletion events of length 1-50 at het indel sites 1/5,000 bp"), with sequencing-noise indels as a decoded ONT BAM
// has them (a deletion is an event, not an 'N').  Everything is a pure function of (seed, read, position): the counting pass
// and the filling pass of a read see the same events.
#include "nc_common.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t rnd(uint64_t seed, uint64_t a, uint64_t b) { return (uint32_t)(mix64(seed ^ mix64(a * 0x100000001B3ull + b)) >> 32); }
__device__ __forceinline__ bool chance(uint32_t r, double p) { return r < (uint32_t)(p * 4294967296.0); }

struct SynthParams {
    int64_t L;
    uint64_t seed;
    double het_snp, hom_snp, het_indel, hom_indel;     // per position
    double p_sub, p_del, p_ins, carry;                 // per read base: substitution, noise deletion / insertion; share of reads that carry a planted indel
    int32_t max_len;                                   // planted indel lengths 1 .. max_len
};

// truth per position p (1-based, index p): ref base, the two haplotype bases, the planted indel that FOLLOWS p on each haplotype
__global__ void k_truth(SynthParams P, uint8_t *__restrict__ ref, uint8_t *__restrict__ hapb /* [2][L+1] */, int8_t *__restrict__ hapi /* [2][L+1] */)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (p > P.L) return;
    const uint32_t r0 = rnd(P.seed, 1, p);
    const uint8_t rb = (uint8_t)(r0 & 3);
    ref[p] = rb;
    uint8_t b0 = rb, b1 = rb;
    const uint32_t rs = rnd(P.seed, 2, p);
    const uint8_t alt = (uint8_t)((rb + 1 + (rnd(P.seed, 3, p) % 3)) & 3);
    if (chance(rs, P.hom_snp)) { b0 = alt; b1 = alt; }
    else if (chance(rs, P.hom_snp + P.het_snp)) { if (rnd(P.seed, 4, p) & 1) b0 = alt; else b1 = alt; }
    hapb[p] = b0;
    hapb[P.L + 1 + p] = b1;
    // a planted indel after p, unless one was planted within the 64 positions before (they must not overlap)
    auto planted = [&](int64_t q) { return q >= 2 && chance(rnd(P.seed, 5, q), P.het_indel + P.hom_indel); };
    int8_t i0 = 0, i1 = 0;
    if (planted(p) && p + 70 < P.L) {
        bool clear = true;
        for (int64_t q = p - 1; q >= p - 64 && clear; q--) clear = !planted(q);
        if (clear) {
            const uint32_t rl = rnd(P.seed, 6, p);
            // short indels dominate: 1-5 in 60 % of the cases, else up to max_len
            int len = (rl & 0xff) < 154 ? 1 + (int)((rl >> 8) % 5) : 1 + (int)((rl >> 8) % (uint32_t)P.max_len);
            const int8_t sl = (int8_t)(((rl >> 28) & 1) ? len : -len);
            const bool hom = chance(rnd(P.seed, 5, p), P.hom_indel);
            const bool first = (rnd(P.seed, 7, p) & 1) != 0;
            if (hom || first) i0 = sl;
            if (hom || !first) i1 = sl;
        }
    }
    hapi[p] = i0;
    hapi[P.L + 1 + p] = i1;
}

struct ReadArgs {
    int32_t n_reads;
    const int32_t *start, *end;
    const int64_t *slot_off;
    const uint8_t *hap;                 // 0 untagged / 1 / 2
    const uint8_t *hapb;
    const int8_t *hapi;
    int32_t *ev_cnt, *ins_cnt;          // count pass out
    const int32_t *ev_off, *ins_off_read;     // fill pass in: first event / first inserted base of the read
    uint8_t *codes;
    int32_t *ev_pos, *ev_len, *ins_off; // fill pass out (ins_off per event, [n_events + 1] closed by the caller)
    uint8_t *ins_bases;
};

template <bool FILL>
__global__ void k_reads(SynthParams P, ReadArgs A)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n_reads) return;
    const int32_t s = A.start[r], e = A.end[r];
    const int tag = A.hap[r];
    const int h = tag ? tag - 1 : (int)(rnd(P.seed, 8, r) & 1);                // an untagged read still comes from one haplotype
    const uint8_t *hb = A.hapb + (int64_t)h * (P.L + 1);
    const int8_t *hi = A.hapi + (int64_t)h * (P.L + 1);
    uint8_t *slot = FILL ? A.codes + A.slot_off[r] : nullptr;
    const int32_t lo16 = s & ~15;
    int nev = 0, nins = 0;
    int32_t ew = FILL ? A.ev_off[r] : 0, iw = FILL ? A.ins_off_read[r] : 0;
    int32_t del_left = 0;
    uint32_t word[4] = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
    const int32_t hi16 = (e + 15) & ~15;
    for (int32_t x = lo16; x < hi16; x++) {
        uint8_t code = 7;
        if (x >= s && x < e) {
            const uint32_t rr = rnd(P.seed ^ 0xA5A5A5A5ull, (uint64_t)r, (uint64_t)x);
            const bool was_del = del_left > 0;
            if (was_del) { code = 4; del_left--; }
            else {
                code = hb[x];
                if (chance(rr, P.p_sub)) code = (uint8_t)((code + 1 + ((rr >> 3) % 3)) & 3);
            }
            // an event on this column (the marker sits on the column BEFORE the insertion / deletion)
            if (!was_del && x > s && x + 60 < e) {                  // on a column where the read has a base, away from its ends
                int len = 0;
                const int8_t pl = hi[x];
                const uint32_t r2 = rnd(P.seed ^ 0x5A5A5A5Aull, (uint64_t)r, (uint64_t)x);
                bool from_truth = false;
                if (pl != 0 && chance(r2, P.carry)) { len = pl; from_truth = true; }
                else if (pl == 0) {
                    const uint32_t r3 = r2 * 2654435761u + 12345u;
                    if (chance(r2, P.p_del)) len = -(1 + ((r3 >> 5) % 16 == 0 ? 2 : (r3 >> 5) % 5 == 0 ? 1 : 0));
                    else if (chance(r2, P.p_del + P.p_ins)) len = 1 + ((r3 >> 5) % 6 == 0 ? 1 : 0);
                }
                if (len != 0) {
                    if (FILL) {
                        A.ev_pos[ew] = x;
                        A.ev_len[ew] = len;
                        A.ins_off[ew] = iw;
                        if (len > 0)
                            for (int i = 0; i < len; i++) {
                                uint8_t b = from_truth ? (uint8_t)(rnd(P.seed, 9, (uint64_t)x * 64 + i) & 3) : (uint8_t)(rnd(P.seed, 10 + r, (uint64_t)x * 64 + i) & 3);
                                const uint32_t r4 = rnd(P.seed ^ 0x77ull, (uint64_t)r * 64 + i, (uint64_t)x);
                                if (chance(r4, P.p_sub)) b = (uint8_t)((b + 1 + ((r4 >> 3) % 3)) & 3);
                                A.ins_bases[iw++] = b;
                            }
                        ew++;
                    }
                    nev++;
                    if (len > 0) nins += len; else del_left = -len;
                }
            }
        }
        if (FILL) {
            word[(x & 15) >> 2] = (word[(x & 15) >> 2] & ~(0xffu << ((x & 3) * 8))) | ((uint32_t)code << ((x & 3) * 8));
            if ((x & 15) == 15) {
                *reinterpret_cast<uint4 *>(slot + (x - 15 - lo16)) = make_uint4(word[0], word[1], word[2], word[3]);
                word[0] = word[1] = word[2] = word[3] = 0x07070707u;
            }
        }
    }
    if (!FILL) { A.ev_cnt[r] = nev; A.ins_cnt[r] = nins; }
}

}   // namespace

// Step 1: truth arrays (device, caller-allocated): ref_code [L + 1] (index p), hap_base [2][L + 1], hap_indel int8 [2][L + 1].
extern "C" int nc_synth_indel_truth(nc_ctx *ctx, int64_t L, uint64_t seed, double het_snp, double hom_snp, double het_indel, double hom_indel,
                                    int32_t max_len, uint8_t *ref_dev, uint8_t *hap_base_dev, int8_t *hap_indel_dev)
{
    if (!ctx || L < 100 || !ref_dev || !hap_base_dev || !hap_indel_dev || max_len < 1 || max_len > 50) return NC_ERR_ARG;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    SynthParams P;
    P.L = L; P.seed = seed; P.het_snp = het_snp; P.hom_snp = hom_snp; P.het_indel = het_indel; P.hom_indel = hom_indel;
    P.p_sub = P.p_del = P.p_ins = P.carry = 0; P.max_len = max_len;
    hipLaunchKernelGGL(k_truth, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream, P, ref_dev, hap_base_dev, hap_indel_dev);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

// Step 2 (fill = 0): per read the number of events / inserted bases -> ev_cnt, ins_cnt [n_reads].  Step 3 (fill = 1), with the
// exclusive prefix sums of those counts in ev_off / ins_off_read: codes (slot layout of nc_pack_fill, slot_off per read),
// ev_pos / ev_len / ins_off [n_events] and ins_bases.  All pointers device.
extern "C" int nc_synth_indel_reads(nc_ctx *ctx, int64_t L, uint64_t seed, double p_sub, double p_del, double p_ins, double carry, int32_t n_reads,
                                    const int32_t *start_dev, const int32_t *end_dev, const int64_t *slot_off_dev, const uint8_t *hap_dev,
                                    const uint8_t *hap_base_dev, const int8_t *hap_indel_dev, int32_t fill, int32_t *ev_cnt_dev,
                                    int32_t *ins_cnt_dev, const int32_t *ev_off_dev, const int32_t *ins_off_read_dev, uint8_t *codes_dev,
                                    int32_t *ev_pos_dev, int32_t *ev_len_dev, int32_t *ins_off_dev, uint8_t *ins_bases_dev)
{
    if (!ctx || n_reads < 0 || !start_dev || !end_dev || !slot_off_dev || !hap_dev || !hap_base_dev || !hap_indel_dev) return NC_ERR_ARG;
    if (n_reads == 0) return NC_OK;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    SynthParams P;
    P.L = L; P.seed = seed; P.het_snp = P.hom_snp = P.het_indel = P.hom_indel = 0; P.max_len = 50;
    P.p_sub = p_sub; P.p_del = p_del; P.p_ins = p_ins; P.carry = carry;
    ReadArgs A;
    A.n_reads = n_reads; A.start = start_dev; A.end = end_dev; A.slot_off = slot_off_dev; A.hap = hap_dev; A.hapb = hap_base_dev; A.hapi = hap_indel_dev;
    A.ev_cnt = ev_cnt_dev; A.ins_cnt = ins_cnt_dev; A.ev_off = ev_off_dev; A.ins_off_read = ins_off_read_dev; A.codes = codes_dev;
    A.ev_pos = ev_pos_dev; A.ev_len = ev_len_dev; A.ins_off = ins_off_dev; A.ins_bases = ins_bases_dev;
    const dim3 gr((unsigned)((n_reads + 63) / 64));
    if (fill) {
        if (!ev_off_dev || !ins_off_read_dev || !codes_dev || !ev_pos_dev || !ev_len_dev || !ins_off_dev || !ins_bases_dev) return NC_ERR_ARG;
        hipLaunchKernelGGL(k_reads<true>, gr, dim3(64), 0, ctx->stream, P, A);
    } else {
        if (!ev_cnt_dev || !ins_cnt_dev) return NC_ERR_ARG;
        hipLaunchKernelGGL(k_reads<false>, gr, dim3(64), 0, ctx->stream, P, A);
    }
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}
