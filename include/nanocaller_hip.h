/*
 * nanocaller_hip.h -- C ABI of libnanocaller_hip.so: NanoCaller's candidate-site pileup featurisation
 * and CNN inference path on MI355X (gfx950).
 *
 * The reference (WGLab/NanoCaller, pure Python) has no FFI; the boundary it exposes for this path is a
 * set of Python call signatures.  Each entry point below names the reference code it
 * replaces (paths relative to the reference repository); INTEGRATION.md shows the ctypes stub a
 * maintainer would add to nanocaller_src/.
 *
 * Conventions: every function returns 0 (NC_OK) or a negative nc_status; nothing throws across the
 * boundary; nc_last_error() gives a message for the last failure on that context.  One context per
 * (device, stream); a context is not thread-safe; no global state.  All arrays are row-major, little
 * endian.  "dev" pointers are device (HBM) addresses, "host" pointers are ordinary host memory.
 * Base codes everywhere: A=0 G=1 T=2 C=3, deletion/N=4 (generate_SNP_pileups.py:104).
 */
#ifndef NANOCALLER_HIP_H
#define NANOCALLER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NC_ABI_VERSION 11  /* 2: nc_decoded_arrays.qstart, nc_indel_scan_params.haploid, drain / async / pass-2 entry points;
                              3: nc_indel_scan_params.impute; 4: nc_timing_sums, nc_enable_timing(2), nc_snp_chunk_depth_async;
                              5: nc_wire_* (reference-difference transfer form of the read pack), nc_d2h_async, nc_indel_pass2_sets;
                              6: nc_allele_prediction_device; 7: nc_star_msa_tensor_dup + nc_pass2_arrays.al_dup, nc_bgzf_read_file, nc_consensus_strings;
                              8: device-resident indel pipeline (nc_indel_pack_*, nc_indel_sites_*, nc_indel_vcf_format), NC_ERR_UNSUPPORTED +
                                 nc_decoded_check, nc_cnn_x_limit + nc_cnn_range_watch (range guard of the fp16x3 trunk), nc_synth_indel_*;
                              9: nc_indel_sites_band + nc_indel_sites_band_stats (banded star alignment), nc_indel_events_pack / _expand (3-byte transfer form of the indel events), nc_inflate_device, nc_bgzf_members / _scan + nc_bam_walk / _meta / _codes / _indel_reads (BAM ingest on the device);
                              10: nc_bgzf_crc_device (CRC-32 of the device-inflated members);
                              11: nc_snp_trunk_info (which SNP trunk kernel the next nc_snp_forward runs, its MFMA count per site), nc_wire_build_del +
                                  nc_wire_apply_deletions (deleted columns implied by the indel events), nc_wire_ref_unpack (reference bytes two per byte), two-byte
                                  indel events (nc_indel_events_pack / _expand with l8 = NULL), nc_snp_scan_begin / _end, nc_wire_expand_del + nc_wire_arrays.blk_ev,
                                  nc_decoded_name_groups + nc_snp_set_mates (alignments that share a read name, keyed by name in the SNP featuriser) */

typedef struct nc_ctx nc_ctx;

enum nc_status {
    NC_OK = 0,
    NC_ERR_ARG = -1,       /* bad argument */
    NC_ERR_CAPACITY = -2,  /* caller-provided capacity too small */
    NC_ERR_NOMEM = -3,     /* host or device allocation failed */
    NC_ERR_HIP = -4,       /* HIP runtime error (message in nc_last_error) */
    NC_ERR_STATE = -5,     /* call order violated (e.g. featurize before scan, weights not loaded) */
    NC_ERR_SELFTEST = -6,  /* device self-test failed at context creation */
    NC_ERR_UNSUPPORTED = -7 /* input the library does not reproduce (reference skips, same-name reads in one column): nc_decoded_check */
};

enum nc_model_kind { NC_MODEL_SNP = 0, NC_MODEL_SNP_HAP = 1, NC_MODEL_INDEL = 2, NC_MODEL_INDEL_HAP = 3 };

/* neighbour-selection mode = the `seq` argument of get_cnd_pos (generate_SNP_pileups.py:6-101) */
enum nc_seq_mode { NC_SEQ_ONT = 0, NC_SEQ_SHORT_ONT = 1, NC_SEQ_UL_ONT = 2, NC_SEQ_UL_ONT_EXTREME = 3, NC_SEQ_PACBIO = 4 };

#define NC_CODE_ABSENT 7      /* padding byte in packed code slots */
#define NC_FLANK 50000        /* scan flank around a chunk (generate_SNP_pileups.py:137,156) */
#define NC_SNP_TENSOR 1025    /* 5*41*5 values per site */

/* ------------------------------------------------------------------ context / device memory */
int nc_abi_version(void);
int nc_device_count(int *n);
/* Creates a context on `device_id`, with its own HIP stream, and runs a device self-test. */
int nc_ctx_create(int device_id, nc_ctx **out);
int nc_ctx_destroy(nc_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream); NULL restores the context's own stream. */
int nc_ctx_set_stream(nc_ctx *ctx, void *hip_stream);
int nc_ctx_sync(nc_ctx *ctx);
const char *nc_last_error(const nc_ctx *ctx);
/* Plain device-memory helpers so a host without any GPU framework can drive the library. */
int nc_malloc(nc_ctx *ctx, size_t bytes, void **dev);
int nc_free(nc_ctx *ctx, void *dev);
int nc_memcpy_h2d(nc_ctx *ctx, void *dev, const void *host, size_t bytes);
int nc_memcpy_d2h(nc_ctx *ctx, void *host, const void *dev, size_t bytes);
/* Asynchronous device -> host transfer on `stream` (NULL: the context's stream).  Small transfers (<= 64 KB) into page-locked,
 * device-accessible memory are written by a copy kernel instead of hipMemcpyAsync: every hipMemcpyAsync, of either direction,
 * queues in order behind a large host -> device copy in flight (a contig's wire pack: ~7 ms), a kernel does not -- the scan
 * totals the host waits for mid-step go this way.  Bulk results use the copy engine. */
int nc_d2h_async(nc_ctx *ctx, void *stream, void *host, const void *dev, size_t bytes);
/* Wall-clock of the last timed call on this context's stream, measured with HIP events (ms);
 * `which`: 0 scan kernel, 1 featurize kernel, 2 CNN forward (all kernels), 3 indel tensor / scan,
 * 4 sum over the launches of the fused SNP trunk kernel in the last forward, 5 number of those launches. */
int nc_last_kernel_ms(nc_ctx *ctx, int which, float *ms);
/* on = 1: all six; on = 2: only the trunk kernel's launches (4, 5) -- their start / stop events ride on the kernel's own
 * dispatch packets and do not perturb the stream, whereas each stage timer puts two barrier packets on it; 0: off. */
int nc_enable_timing(nc_ctx *ctx, int on);
/* Totals of the same six quantities over all calls since nc_enable_timing(ctx, 1) (sum_ms[6]; count[6] = calls folded in,
 * may be NULL).  A call's timers are folded in when its events are re-used by the next call of the same stage or here, so
 * calls may be left in flight while the next one is enqueued (snpCaller.call_chunks(defer=True)). */
int nc_timing_sums(nc_ctx *ctx, double *sum_ms, int64_t *count);

/* ------------------------------------------------------------------ packed alignments ("read pack")
 * Replaces the pysam pileup objects of generate_SNP_pileups.py:134-164: alignments decoded once on the
 * host are laid out read-major for HBM.  A read covering reference positions [start,end) (1-based) owns
 * the 16-byte aligned slot of position-addressed bytes [floor16(start), ceil16(end)); its code at position
 * p is codes[base + p]; bytes of the slot outside [start,end) hold NC_CODE_ABSENT.  `base` is a multiple
 * of 16, so 16-position groups load as one aligned dwordx4.  Reads are in coordinate (start) order.
 * The tile index lists, for every tile of `tile_size` consecutive positions, the reads overlapping it.
 */
typedef struct {
    int32_t start;      /* first covered position (1-based) */
    int32_t end;        /* one past the last covered position */
    int64_t base_flag;  /* (base & ~15) | flags; bit0 = reverse strand ((flag & 0x910)/16, :143);
                           bits 1-2 = haplotype tag HP (0 untagged, 1, 2; generate_indel_pileups.py:180-185);
                           bit 3 = the read name is shared with another kept alignment (nc_snp_set_mates) */
} nc_tile_entry;

typedef struct {
    int64_t codes_len;            /* bytes in `codes` (multiple of 16) */
    const uint8_t *codes;         /* dev */
    int32_t tile_size;            /* 1024, 2048 or 4096 positions */
    int32_t tile_pos0;            /* first position of tile 0 (multiple of 16; may be <= 0) */
    int32_t n_tiles;
    const int32_t *tile_off;      /* dev [n_tiles+1] */
    const nc_tile_entry *tile_ent;/* dev [tile_off[n_tiles]], coordinate order within a tile */
    int64_t n_entries;
} nc_readpack;

/* Host-side packer.  Inputs (host): n reads in coordinate order, read r covers [start[r], end[r]) and
 * codes_in[off[r] + p - start[r]] is its code (0..4) at p; keep[r]==0 drops the read (pileup flag filter
 * 0x4|0x100|0x200|0x400|0x800, generate_SNP_pileups.py:151-157); strand[r] bit0 = reverse strand, bits 1-2 = HP tag, bit 3 = shared read name.
 * Step 1 sizes the outputs; step 2 fills caller-allocated host buffers.  With codes_in == codes_out == NULL
 * nc_pack_fill builds the tile index only (slot r starts at byte sum_{q<r} slot_size(q), kept reads only). */
int nc_pack_plan(int32_t n_reads, const int32_t *start, const int32_t *end, const uint8_t *keep,
                 int32_t tile_size, int32_t pos_lo, int32_t pos_hi,
                 int64_t *codes_len, int32_t *tile_pos0, int32_t *n_tiles, int64_t *n_entries);
int nc_pack_fill(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off,
                 const uint8_t *codes_in, const uint8_t *strand, const uint8_t *keep,
                 int32_t tile_size, int32_t tile_pos0, int32_t n_tiles,
                 uint8_t *codes_out, int64_t codes_len, int32_t *tile_off, nc_tile_entry *tile_ent,
                 int64_t n_entries);

/* ------------------------------------------------------------------ wire pack: host -> device transfer form of the read pack
 * The reference feeds every chunk from host memory (snpCaller.py:86, generate_SNP_pileups.py:156); here a contig's decoded
 * alignments cross PCIe once.  At 1 B per pileup entry that is 1.9 GB for a chr20-sized 30x contig (~35 ms of PCIe Gen5
 * against ~11.5 ms of GPU work), but 92 % of an ONT read's positions (99.8 % of a HiFi read's) carry the reference base.
 * The transfer form therefore holds only the differences against the reference:
 *   rd_start / rd_end / slot_off   the kept reads in slot order (slot_off[r] = byte offset of read r's slot in `codes`);
 *   ref_wire                       one byte per reference position of the tile grid: bits 0-2 base code (A0 G1 T2 C3, 4 =
 *                                  unknown), bit 3 = column skipped by the scan (soft-masked / non-AGTC, quirk E4, or
 *                                  excluded, :161) -- so ref_code[i] = (ref_wire[i] & 8) ? 4 : (ref_wire[i] & 7);
 *   blk_read / blk_off / events    per 1024-byte block of `codes`: the first read whose slot reaches into it, and its
 *                                  difference events, 2 B each: (byte offset in the block) | code << 12, for every read
 *                                  position whose code differs from the reference base (deleted positions: code 4).
 * nc_wire_build makes it on the host cores from the same inputs as nc_pack_fill; nc_wire_expand rebuilds the
 * position-addressed `codes` in HBM byte for byte as nc_pack_fill writes them (and the scan's ref_code array).
 * The tile index still comes from nc_pack_plan / nc_pack_fill (index-only mode). */
typedef struct nc_wire nc_wire;
typedef struct {
    int32_t n_reads;            /* kept reads */
    const int32_t *rd_start;    /* host [n_reads] */
    const int32_t *rd_end;      /* host [n_reads] */
    const int64_t *slot_off;    /* host [n_reads + 1] */
    int64_t codes_len;          /* bytes of the expanded codes array (= nc_pack_plan's codes_len) */
    int64_t n_blocks;           /* ceil(codes_len / 1024) */
    const uint32_t *blk_off;    /* host [n_blocks + 1]; NC_ERR_CAPACITY from nc_wire_build at >= 2^32 events */
    const int32_t *blk_read;    /* host [n_blocks] */
    const uint16_t *events;     /* host [n_events] */
    int64_t n_events;
    const uint32_t *blk_ev;     /* host [n_blocks], nc_wire_build_del only (else NULL): see nc_wire_expand_del */
    const uint8_t *ev_bytes;    /* host [n_ev_bytes (+ 8 readable)], nc_wire_build2 with flag 1 (else NULL): the events one byte each; blk_off then counts bytes */
    int64_t n_ev_bytes;
} nc_wire_arrays;
/* ref_wire[i] describes position ref_pos0 + i (ref_pos0 a multiple of 16: the tile grid's tile_pos0), i < ref_len;
 * positions outside predict code 4.  Other arguments as nc_pack_fill.  The result is owned by the library. */
int nc_wire_build(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                  const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, nc_wire **out);
int nc_wire_view(const nc_wire *w, nc_wire_arrays *view);
int nc_wire_free(nc_wire *w);
/* The same with a read pack that travels beside its indel events (nc_indel_events': ev_off [n_reads + 1] per read of the INPUT order, a negative
 * length -L at column c deletes columns c + 1 .. c + L): in every block that lies inside ONE read and inside the reference grid the deleted columns'
 * code (4) is implied by the read's events and left out of the difference events; blk_ev[b] = index, among the KEPT reads' events in pack order, of
 * the read's first event whose run reaches the block (0xffffffff: not such a block).  nc_wire_expand_del = nc_wire_expand that writes those columns
 * from d_ev_off / d_ev_pos / d_ev_len (dev, the kept reads' events, absolute: after nc_indel_events_expand on the same stream) in the block's LDS
 * image, so the codes array is written once.  nc_wire_apply_deletions (all pointers dev) writes code 4 over EVERY deleted column of an expanded
 * array: a separate pass with the same result (idempotent), kept for callers that expand without the events at hand.  NC_ERR_UNSUPPORTED from
 * nc_wire_build_del: a deleted column of the input does not carry code 4 (build the plain form).  This library's own transfer form: no reference
 * counterpart (the reference decodes per chunk with pysam, generate_indel_pileups.py:213-264). */
/* ref_wire as it crosses PCIe since ABI 11: two positions per byte (position 2 i in the low nibble of byte i); nc_wire_ref_unpack (pointers dev, 16-byte
 * aligned) rebuilds the byte array nc_wire_expand reads, on the context's stream. */
int nc_wire_ref_unpack(nc_ctx *ctx, const uint8_t *d_ref_nib, int64_t ref_len, uint8_t *d_ref_wire);
/* The inserted bases of the indel caller's pack (nc_indel_reads.ins_bases: codes 0..4) as they cross PCIe since round 6: two bits a base (base i in
 * bits 2 (i & 3) of byte i >> 2), the few code-4 bases as a list of their indices.  nc_wire_ins_unpack (pointers dev; d_packed 4-byte,
 * d_ins_bases 16-byte aligned, both padded to a multiple of 16 bases) rebuilds the byte array on the context's stream.  No reference
 * counterpart (pysam hands the reference query_sequence strings, generate_indel_pileups.py:331). */
int nc_wire_ins_unpack(nc_ctx *ctx, const uint8_t *d_packed, int64_t n_bases, const int32_t *d_other_idx, int32_t n_other, uint8_t *d_ins_bases);
int nc_wire_build_del(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                      const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, const int32_t *ev_off,
                      const int32_t *ev_pos, const int32_t *ev_len, nc_wire **out);
int nc_wire_apply_deletions(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                            const int32_t *d_ev_off, const int32_t *d_ev_pos, const int32_t *d_ev_len, uint8_t *d_codes);
/* All pointers dev, 16-byte aligned; d_codes [codes_len] and d_ref_code [ref_len] (may be NULL) are written on the
 * context's stream.  HBM-write-bound: 1 B per pileup entry. */
int nc_wire_expand(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                   const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                   const int32_t *d_blk_read, const uint16_t *d_events, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                   uint8_t *d_ref_code);
/* The difference events one byte each (round 6; 126 -> 72 MB per chr20-sized ONT contig): bits 2-7 = columns skipped since the block's previous event
 * (0 .. 62; 63 = a filler that skips 63 columns and is no event), bits 0-1 = which of the four codes other than the predicted one (against a predicted
 * base b: 0 / 1 / 2 = base (b + 1 + k) & 3, 3 = code 4; against a predicted 4: the base).  nc_wire_build2: flags bit 0 asks for them (a pack with a
 * code beyond 4 keeps the two-byte events: view.ev_bytes == NULL); ev_off != NULL = nc_wire_build_del's implied deletions.  nc_wire_expand2 = the
 * expansion of that form (d_blk_ev .. d_ev_len NULL without implied deletions). */
int nc_wire_build2(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off, const uint8_t *codes_in,
                   const uint8_t *keep, const uint8_t *ref_wire, int32_t ref_pos0, int64_t ref_len, const int32_t *ev_off,
                   const int32_t *ev_pos, const int32_t *ev_len, int32_t flags, nc_wire **out);
int nc_wire_expand2(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                    const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                    const int32_t *d_blk_read, const uint8_t *d_ev_bytes, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                    uint8_t *d_ref_code, const uint32_t *d_blk_ev, const int32_t *d_ev_off, const int32_t *d_ev_pos,
                    const int32_t *d_ev_len);
int nc_wire_expand_del(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_rd_end, const int64_t *d_slot_off,
                       const uint8_t *d_ref_wire, int32_t ref_pos0, int64_t ref_len, const uint32_t *d_blk_off,
                       const int32_t *d_blk_read, const uint16_t *d_events, int64_t n_blocks, uint8_t *d_codes, int64_t codes_len,
                       uint8_t *d_ref_code, const uint32_t *d_blk_ev, const int32_t *d_ev_off, const int32_t *d_ev_pos,
                       const int32_t *d_ev_len);

/* Transfer form of the indel path's per-read events (csrc/nc_wire.hip): 3 bytes per event instead of the 12 the kernels read.
 *   d16 [n_events]  column - column of the read's previous event (the first one: - rd_start[r]); 0xFFFF: the event is in the side table
 *   l8  [n_events]  signed length ('+n' / '-n' of the pileup, generate_indel_pileups.py:216-231); the side table's when d16 is 0xFFFF
 *   big_idx (ascending event indices) / big_pos / big_len: events with a distance >= 0xFFFF or |length| >= 128
 *   read_ins_off [n_reads + 1] (may be NULL): offset of each read's first inserted base; the per-event ins_off is its running sum
 * nc_indel_events_pack: host; returns NC_ERR_CAPACITY with *n_big = the needed size when big_cap is too small.
 * nc_indel_events_expand: all pointers dev; writes ev_pos / ev_len [n_events] and (if given) ins_off [n_events + 1] on the context's stream. */
int nc_indel_events_pack(int32_t n_reads, const int32_t *rd_start, const int32_t *ev_off, const int32_t *ev_pos, const int32_t *ev_len,
                         uint16_t *d16, int8_t *l8, int32_t *read_ins_off, int64_t big_cap, int32_t *big_idx, int32_t *big_pos,
                         int32_t *big_len, int64_t *n_big);
int nc_indel_events_expand(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_ev_off, const uint16_t *d_d16,
                           const int8_t *d_l8, int32_t n_big, const int32_t *d_big_idx, const int32_t *d_big_pos, const int32_t *d_big_len,
                           const int32_t *d_read_ins_off, int32_t *d_ev_pos, int32_t *d_ev_len, int32_t *d_ins_off);
/* One byte per event (round 6): b8 [n_events] = distance to the read's previous event in bits 2-7 (0 .. 62) | length code in bits 0-1 (+1, -1, +2, -2);
 * 0xFF = the event is the next entry of d16x (the two-byte form above: distance | signed 5-bit length << 11, 0xFFFF -> side table), whose entries
 * start at read_esc_off[r] [n_reads + 1] for read r.  NC_ERR_CAPACITY from the packer when esc_cap / big_cap are too small (*n_esc / *n_big say
 * what is needed). */
int nc_indel_events_pack8(int32_t n_reads, const int32_t *rd_start, const int32_t *ev_off, const int32_t *ev_pos, const int32_t *ev_len,
                          uint8_t *b8, uint16_t *d16x, int64_t esc_cap, int32_t *read_esc_off, int32_t *read_ins_off, int64_t big_cap,
                          int32_t *big_idx, int32_t *big_pos, int32_t *big_len, int64_t *n_esc, int64_t *n_big);
int nc_indel_events_expand8(nc_ctx *ctx, int32_t n_reads, const int32_t *d_rd_start, const int32_t *d_ev_off, const uint8_t *d_b8,
                            const uint16_t *d_d16x, const int32_t *d_read_esc_off, int32_t n_big, const int32_t *d_big_idx,
                            const int32_t *d_big_pos, const int32_t *d_big_len, const int32_t *d_read_ins_off, int32_t *d_ev_pos,
                            int32_t *d_ev_len, int32_t *d_ins_off);

/* DEFLATE on the device (csrc/nc_inflate.hip): the raw-deflate payloads of n BGZF members (SAMv1 4.1) in two launches -- Huffman decoding, one
 * lane per member, into tokens; match resolution, one wave per member.  All pointers dev: d_comp = the compressed bytes (readable 8 bytes
 * past the last payload), d_coff / d_clen = byte offset and length of member b's payload in it, d_out + d_ooff[b] = where its d_isize[b]
 * (<= 65536) bytes go, d_status[b] = 0 or why the member is not a valid stream of that length; workspace: d_tok = ceil(n_blocks / 64)
 * x 4,194,304 dwords (64 members x 65,536 tokens), d_ntok = n_blocks counters.  Runs on the context's stream.  CRC-32s: nc_bgzf_crc_device.  Replaces the host inflate behind generate_SNP_pileups.py:134-164's input. */
int nc_inflate_device(nc_ctx *ctx, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen, uint8_t *d_out,
                      const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status, uint32_t *d_tok, int32_t *d_ntok);
/* The two launches apart: phase 1 = tokens only, 2 = resolution only (of the tokens an earlier phase-1 call left in d_tok / d_ntok), 3 = both.
 * With nc_ctx_set_stream between the calls the halves of consecutive batches overlap (the Huffman kernel takes whole CUs; where a round
 * leaves CUs free the previous batch's resolution runs on them). */
int nc_inflate_device_phase(nc_ctx *ctx, int32_t phase, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen,
                            uint8_t *d_out, const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status, uint32_t *d_tok, int32_t *d_ntok);
/* CRC-32 (RFC 1952) of every inflated member against the value in its BGZF trailer (the four bytes behind the payload), on the context's stream
 * behind the inflate: d_status[b] = 7 where they differ and the inflate left 0.  htslib verifies the CRC of every block it inflates (the reader
 * behind generate_SNP_pileups.py:134); on the device route the bytes never reach the host, so the check runs where they are. */
int nc_bgzf_crc_device(nc_ctx *ctx, int32_t n_blocks, const uint8_t *d_comp, const int64_t *d_coff, const int32_t *d_clen, const uint8_t *d_out,
                       const int64_t *d_ooff, const int32_t *d_isize, int32_t *d_status);

/* BAM records on the device (csrc/nc_ingest.hip): from the inflated BGZF stream in HBM to the slots of the read pack, for the SNP route --
 * what nc_bam_decode + nc_pack_fill do on host threads (generate_SNP_pileups.py:134-164's input).
 * nc_bgzf_members (host): the members of a BGZF file image: payload offset / length and inflated size of each (the arguments of
 *   nc_inflate_device).  NC_ERR_CAPACITY when there are more than `cap` (n_members counts them all), NC_ERR_ARG for a malformed member.
 * nc_bgzf_scan (host): the same from byte `start`, as far as whole members lie inside data[0, n) and the outputs have room; *next = where
 *   it stopped (a file that is still being read is scanned piece by piece).
 * nc_bam_walk: record boundaries.  d_seed = n_seeds record starts, ascending offsets into d_raw (the entries of the .bai linear index),
 *   d_seed_tid = the contig of each.  d_first == NULL: d_out[i] = records from seed i up to seed i + 1 / the end of the contig's records;
 *   else d_out[d_first[i] + k] = offset of the k-th of them (d_first = exclusive prefix sums of the counts).  d_status (one int32, zeroed
 *   by the caller): bit 0 a block_size below 32, bit 1 an index entry that is not a record start, bit 2 (nc_bam_meta) a record whose
 *   fields do not fit its block_size.
 * nc_bam_meta: d_meta = int32 [NC_BAM_META_COLS][n_rec], field-major: refID, pos (0-based), flag (| NC_FLAG_REFSKIP), reference span of
 *   the CIGAR (0: not an alignment nc_bam_decode would return), l_seq, 1 if the record carries the bases its CIGAR consumes, HP (0/1/2),
 *   PS, FNV-1a hash of the read name (low, high), operations of the real CIGAR (the CG tag's for SAMv1 4.2.2 placeholders), its byte
 *   offset from the record's refID field.
 * nc_bam_codes: the slots of n_reads kept reads (d_rec = record offsets, d_slot = byte offset of each slot in d_codes, nc_pack_fill's
 *   layout; d_cigd / d_ncig = the last two meta columns, bit 31 of d_ncig set = no usable bases: code 4 everywhere; d_start = 1-based
 *   first position).  d_codes holds NC_CODE_ABSENT everywhere beforehand; the result is byte for byte what nc_pack_fill writes. */
#define NC_BAM_META_COLS 12
int nc_bgzf_members(const uint8_t *data, int64_t n, int64_t cap, int64_t *coff, int32_t *clen, int32_t *isize, int64_t *n_members);
int nc_bgzf_scan(const uint8_t *data, int64_t n, int64_t start, int64_t cap, int64_t *coff, int32_t *clen, int32_t *isize, int64_t *n_members,
                 int64_t *next);
int nc_bam_walk(nc_ctx *ctx, const uint8_t *d_raw, int64_t raw_len, int32_t n_seeds, const int64_t *d_seed, const int32_t *d_seed_tid,
                const int64_t *d_first, int64_t *d_out, int32_t *d_status);
int nc_bam_meta(nc_ctx *ctx, const uint8_t *d_raw, int64_t n_rec, const int64_t *d_rec_off, int32_t *d_meta, int32_t *d_status);
int nc_bam_codes(nc_ctx *ctx, const uint8_t *d_raw, int32_t n_reads, const int64_t *d_rec, const int64_t *d_slot, const int32_t *d_cigd,
                 const int32_t *d_ncig, const int32_t *d_start, uint8_t *d_codes);
/* The indel path's per-read sections of the same kept reads -- what nc_bam_decode's events and nc_indel_pack_build make on the host
 * (generate_indel_pileups.py:216-231's '+n' / '-n' markers, the inserted bases, the query bases behind the last aligned one).  Pass 1
 * (d_ev_off == NULL): d_counts = int32 [3][n_reads]: events, inserted bases, tail bases (at most tail_cap) per read.  Pass 2: d_ev_off /
 * d_ins_base / d_tail_off = exclusive prefix sums of those; fills d_ev_pos / d_ev_len / d_ins_off (per event) / d_ins_bases / d_tail_bases. */
int nc_bam_indel_reads(nc_ctx *ctx, const uint8_t *d_raw, int32_t n_reads, const int64_t *d_rec, const int32_t *d_cigd, const int32_t *d_ncig,
                       int32_t tail_cap, int32_t *d_counts, const int32_t *d_ev_off, const int32_t *d_ins_base, const int32_t *d_tail_off,
                       int32_t *d_ev_pos, int32_t *d_ev_len, int32_t *d_ins_off, uint8_t *d_ins_bases, uint8_t *d_tail_bases);

/* ------------------------------------------------------------------ SNP candidate scan (K1)
 * Replaces the column loop of get_snp_testing_candidates (generate_SNP_pileups.py:156-186) for a batch
 * of chunks of ONE contig: every position of [scan_lo, scan_hi] is scanned once, per-position base counts
 * give alt_freq = max_{b != ref} count(b) / n in float64, neighbour sites (t0 <= alt_freq [< t1]) and
 * candidates (min_allele_freq <= alt_freq) are emitted in ascending position order; each chunk
 * [chunk_start[c], chunk_end[c]] (both inclusive, utils.py:79-80) then owns the candidates inside it --
 * a position shared by two adjacent chunks is emitted once per chunk (quirk E3).
 * ref_code[p - ref_pos0] is the reference code at p: 0..3, or 4 to skip the column (non-AGTC or
 * soft-masked base, generate_SNP_pileups.py:137,161, or an exclude_bed hit).  It must lie on the pack's tile
 * grid: ref_pos0 == pack->tile_pos0 and ref_len >= n_tiles*tile_size (pad with 4).
 * Results stay in the context (device); counts are returned.  Synchronises once.
 */
typedef struct {
    int32_t mincov;             /* dct['mincov'] */
    double min_allele_freq;     /* dct['min_allele_freq'] */
    double nbr_t0, nbr_t1;      /* dct['threshold'] */
    int32_t haploid;            /* region['ploidy']=='haploid': neighbour test is t0 <= alt_freq only (:177) */
} nc_scan_params;

int nc_snp_scan(nc_ctx *ctx, const nc_readpack *pack,
                const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                int32_t scan_lo, int32_t scan_hi, const nc_scan_params *params,
                int32_t n_chunks, const int32_t *chunk_start_host, const int32_t *chunk_end_host,
                int32_t *n_nbr, int32_t *n_cand, int32_t *n_sites);

/* The same scan in two halves (ABI 11): nc_snp_scan_begin enqueues every kernel up to the candidate compaction plus the copy of the three totals into the
 * context's pinned mailbox and returns without waiting; nc_snp_scan_end waits for that copy alone (an event, not the stream), then sizes and
 * launches what depends on the totals and returns the counts.  Work enqueued between the two runs on: snpCaller.call_chunks puts the PREVIOUS
 * contig's CNN there, so the host's round trip for the totals is hidden under it (the reference's per-chunk loop has no such step:
 * snpCaller.py:83-87).  One scan per context may be open; nc_snp_scan = begin + end. */
int nc_snp_scan_begin(nc_ctx *ctx, const nc_readpack *pack,
                      const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                      int32_t scan_lo, int32_t scan_hi, const nc_scan_params *params,
                      int32_t n_chunks, const int32_t *chunk_start_host, const int32_t *chunk_end_host);
int nc_snp_scan_end(nc_ctx *ctx, int32_t *n_nbr, int32_t *n_cand, int32_t *n_sites);

/* Copies the scan results of the context to host arrays (any may be NULL):
 * nbr_pos[n_nbr]; per site (n_sites, chunk-major then ascending position): pos, chunk id,
 * n (= dp, pileup entries incl. deletions, :164,186) and alt count (freq = alt/n in float64, :166). */
int nc_snp_scan_fetch(nc_ctx *ctx, int32_t *nbr_pos, int32_t *site_pos, int32_t *site_chunk,
                      int32_t *site_n, int32_t *site_alt);

/* Same copies, asynchronous: enqueued on `copy_stream` (a hipStream_t; NULL = the context's stream) and NOT waited for.
 * The host buffers should be pinned; the caller synchronises `copy_stream` before reading them and before the next
 * nc_snp_scan on this context (the scan results live in the context).  Lets the fetch overlap featurisation / CNN. */
int nc_snp_scan_fetch_async(nc_ctx *ctx, void *copy_stream, int32_t *nbr_pos, int32_t *site_pos, int32_t *site_chunk,
                            int32_t *site_n, int32_t *site_alt);

/* ------------------------------------------------------------------ SNP tensor build (K2-K4)
 * Replaces get_cnd_pos + the per-candidate loop (generate_SNP_pileups.py:6-101, 200-263) for the sites of
 * the last nc_snp_scan on this context: one wavefront per site picks <= 20+20 neighbour sites, gathers the
 * codes of the site's reads at those columns and writes the (5,41,5) tensor.
 * Neighbour sites are limited to the owning chunk's scan window [max(1,start-50000), end+50000] (quirk E9).
 * Above maxcov the reference draws an unseeded random.sample (:215-216); this library keeps the first
 * maxcov reads in coordinate order (documented policy; parity is defined for depth <= maxcov).
 * Outputs (dev, caller-allocated): x f32 [n_sites][5][41][5]; ref_code i32 [n_sites]; fwd_dp, rev_dp
 * i32 [n_sites][4] (AGTC counts over all reads by strand, :210-213); site_depth i32 [n_sites] (|S| after the
 * maxcov cut, :263).  Sites failing `len(cols) < min_nbr_sites` (:244) get valid[s]=0 and a zero tensor.
 */
int nc_snp_featurize(nc_ctx *ctx, const nc_readpack *pack,
                     const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                     int32_t seq_mode, int32_t maxcov, int32_t min_nbr_sites,
                     float *x_dev, int32_t *ref_code_out_dev, int32_t *fwd_dp_dev, int32_t *rev_dp_dev,
                     int32_t *site_depth_dev, uint8_t *valid_dev);

/* Alignments that share a read name (a split read's primary + supplementary records under dct['supplementary']; ABI 11).  The reference's
 * pileup is a dict keyed by NAME per column (generate_SNP_pileups.py:175,185): where several alignments of one name cover a column the LAST in
 * file order is the column's entry (all of them count in `n` and the allele frequency, :164-166: the scan is per alignment), a site's row for a
 * name takes each neighbour column from whichever of the name's alignments covers it (:223,232), the sampled depth counts names (:208,263), and
 * the strand is the name's (:141-143: the caller writes it into every member's tile entries).  The pack's tile entries of such alignments carry
 * bit 3 of base_flag; this call hands the featuriser their table for the following nc_snp_featurize calls (n_mates = 0 clears it; the pointers
 * are borrowed): d_mate_key [n_mates] = byte offset of the alignment's slot in `codes`, ascending (= file order); d_mate_rec [n_mates][4] int32,
 * 16-byte aligned = {start, end, table index of the next alignment of the same name (circular), 0}.  Needs the int16 tensor format and
 * maxcov < 256 (NC_ERR_UNSUPPORTED from nc_snp_featurize otherwise). */
int nc_snp_set_mates(nc_ctx *ctx, int32_t n_mates, const int64_t *d_mate_key, const int32_t *d_mate_rec);

/* Per-site coverage scale (snpCaller.py:93-96, 170-173) for the sites of the last scan:
 * mode 0: scale[s] = train_coverage / mean(site_depth over the site's chunk) (chunk constant, quirk E2)
 * mode 1: scale[s] = train_coverage / dp[s] (--disable_coverage_normalization).
 * chunk_depth_host (may be NULL) receives the per-chunk mean depth (float64, :274). */
int nc_snp_scale(nc_ctx *ctx, const int32_t *site_depth_dev, const uint8_t *valid_dev, double train_coverage,
                 int32_t mode, double *scale_dev, double *chunk_depth_host);
/* The per-chunk mean depths of the last nc_snp_scale, copied on `copy_stream` (ordered behind the scale kernel) without
 * synchronising: with chunk_depth_host = NULL above, the caller's stream is never drained between the featuriser and the
 * CNN.  `chunk_depth_host_pinned` [n_chunks] must be page-locked and is valid once the copy stream has passed. */
int nc_snp_chunk_depth_async(nc_ctx *ctx, void *copy_stream, double *chunk_depth_host_pinned);

/* ------------------------------------------------------------------ CNN forward (K5 / K9)
 * nc_load_weights: canonical flat f32 blob (nanocaller_amd/weights.py LAYER_SPECS order, Keras layouts),
 * replaces Model.load_weights (snpCaller.py:70-78, indelCaller.py:51-57).
 * nc_snp_forward replaces SNP_model.call / haploid_SNP_model.call (model_architect.py:36-64,
 * model_architect_SNP_haploid.py:33-53) including the coverage scaling of rows 1..4 / channels 0..3:
 * scale_mode 0 multiplies in f32 by (float)scale[s] (numpy<2 scalar semantics), 1 multiplies in f64.
 * probs f32 [n][4]: diploid = class-1 probability of the A,G,T,C heads; haploid = 4-way softmax.
 * gt f32 [n][2] (diploid only, may be NULL).
 */
int nc_load_weights(nc_ctx *ctx, int32_t model_kind, const float *blob_host, size_t n_floats);
/* Arithmetic of the SNP convolution trunk: 0 (default) = fp16x3 split precision on the 16x-rate matrix pipe (every fp32
 * operand as hi + lo halves, hi*hi + hi*lo + lo*hi with fp32 accumulation: fp32-rounding-level error, measured
 * max |dp| ~1e-6); 1 = exact fp32 MFMA (bit-for-bit an fmaf chain). */
int nc_set_cnn_precision(nc_ctx *ctx, int exact_fp32);
/* Format of the SNP tensors between nc_snp_featurize and nc_snp_forward: 0 (default) = float32 [n][5][41][5], the reference's
 * array; 1 = int16 with the same shape -- every entry is a small integer (a count of at most maxcov <= 1024 reads, +-, or a
 * 0/1 flag), so the conversion is exact, and the 4,100 B per site that the featuriser writes and the CNN reads become 2,050.
 * x_dev of both calls is then an int16 buffer.  Only the split-precision trunk reads it (not nc_set_cnn_precision(ctx, 1)). */
int nc_set_tensor_format(nc_ctx *ctx, int fmt);
int nc_snp_forward(nc_ctx *ctx, int32_t model_kind, int64_t n, const float *x_dev, const int32_t *ref_code_dev,
                   const double *scale_dev, int32_t scale_mode, float *probs_dev, float *gt_dev);

/* nc_snp_forward with an asynchronous result drain: as soon as a batch of sites is finished its rows of probs (and gt)
 * are copied to the pinned host arrays probs_host[n][4] / gt_host[n][2] on `copy_stream`, overlapping the next batch's
 * compute (the reference moves every batch's predictions to numpy, snpCaller.py:111-113,183-185).  The caller
 * synchronises `copy_stream` before reading the host arrays.  gt_host / gt_dev may be NULL. */
int nc_snp_forward_drain(nc_ctx *ctx, int32_t model_kind, int64_t n, const float *x_dev, const int32_t *ref_code_dev,
                         const double *scale_dev, int32_t scale_mode, float *probs_dev, float *gt_dev,
                         void *copy_stream, float *probs_host, float *gt_host);
/* Range guard of the split-precision kernels.  Their epilogues clamp activations to the fp16 range (6e4); a clamp that fired
 * would be a wrong probability.  nc_load_weights derives, from the L1 norms of the three convolutions and |selu(v)| <= lambda |v|,
 * x_limit = the largest |input value| (after the coverage scaling) for which NO activation of that model can reach the clamp
 * (ONT-HG002: 171; a 30x site scaled by 48/30 has |x| <= 48).  nc_cnn_range_watch(ctx, flags): the following nc_snp_forward[_drain]
 * calls on the split-precision trunk set flags[s] = 1 (dev, one byte per site of the call, zeroed by the caller) for every site
 * whose scaled tensor exceeds x_limit; the caller re-runs exactly those sites with nc_set_cnn_precision(ctx, 1) (what
 * nanocaller_amd.snpCaller.call_chunks does), so a result is either proven in range or computed by the exact fp32 trunk.  NULL
 * stops watching.  The indel CNN's inputs are msa() frequencies (|x| <= 1): a model with x_limit < 1 (none of the shipped ones)
 * runs on the exact fp32 kernels. */
int nc_cnn_x_limit(nc_ctx *ctx, int32_t model_kind, float *x_limit);
int nc_cnn_range_watch(nc_ctx *ctx, uint8_t *site_flags_dev);
/* The split-precision SNP trunk the next nc_snp_forward[_drain] of this context launches (it depends on the tensor format and the precision
 * mode): kernel_id 0 = k4_conv12 (exact fp32), 1 = k5_trunk_h3, 2 = k5_trunk_p3, 3 = k5_trunk_lin (int16 tensors: conv1 by linearity, two f16
 * products on the integer entries instead of three); mfma_per_site = v_mfma_f32_16x16x32_f16 instructions it executes per site (bench.py's
 * executed-vs-algorithmic figure).  No reference counterpart (measurement support for model_architect.py:36-64's conv1-3). */
int nc_snp_trunk_info(nc_ctx *ctx, int32_t *mfma_per_site, int32_t *kernel_id);
/* Indel CNN (model_architect_indel.py:28-48 rows=15 -> [n][4]; haploid rows=5 -> [n][1] sigmoid). */
int nc_indel_forward(nc_ctx *ctx, int32_t model_kind, int64_t n, const float *x_dev, float *probs_dev);

/* ------------------------------------------------------------------ indel MSA rows -> tensor (K8)
 * Replaces the histogram part of msa() (generate_indel_pileups.py:57-71): for each of n_sets aligned read
 * sets (rows[s] is [n_rows[s]][n_cols[s]] symbols 0..4 = A,G,T,C,'-', stored at rows_dev + row_off[s]) and its
 * aligned reference row, writes x [n_sets][5][128][2] (column frequency minus ref one-hot; ref one-hot), and
 * the gap-handicapped consensus symbols cns [n_sets][max_cols] (NC_CODE_ABSENT-padded, gaps kept as 4).
 */
int nc_indel_tensor(nc_ctx *ctx, int32_t n_sets, const uint8_t *rows_dev, const int64_t *row_off_dev,
                    const int32_t *n_rows_dev, const int32_t *n_cols_dev, const uint8_t *ref_rows_dev,
                    const int64_t *ref_off_dev, int32_t max_cols, float *x_dev, uint8_t *cns_dev);

/* ------------------------------------------------------------------ indel candidate window scan (K7)
 * Replaces pass 1 of get_indel_testing_candidates (generate_indel_pileups.py:197-276; the impute_indel_phase
 * branch :278-304 is not covered): for every column of [start, end] the number of haplotype-1 / haplotype-2 reads
 * and, per haplotype, the number of DISTINCT reads carrying a long (2 < L <= 50) / small (L <= 10) insertion /
 * deletion marker within the last win_size / small_win_size yielded columns; col_type[v - start] receives the
 * decision of :266-275 for that column taken in isolation: 0 (long-window rule), 1 (small-window rule), -1 (none,
 * or the column is not evaluated: zero depth, excluded, or a haplotype below mincov).  The order-dependent
 * suppression `v <= prev` (:249) is a scalar recurrence over the few flagged columns and stays with the caller.
 * Events are read-major for the KEPT reads in pack order: read r owns ev_pos/ev_len[ev_off[r] .. ev_off[r+1]),
 * ascending positions; ev_pos = the column whose pileup string carries the '+n' / '-n' marker, ev_len > 0 insertion,
 * < 0 deletion.  excl (optional, on the pack's tile grid like ref_code) != 0 skips the column (ex_bed, :217).
 */
typedef struct {
    int32_t n_reads;
    const int32_t *ev_off;   /* dev [n_reads+1] */
    const int32_t *ev_pos;   /* dev */
    const int32_t *ev_len;   /* dev */
    const uint8_t *read_hap; /* dev [n_reads] 0/1/2 */
} nc_indel_events;

typedef struct {
    int32_t mincov, win_size, small_win_size;   /* dct['mincov'], dct['win_size'], dct['small_win_size'] */
    double ins_t, del_t;                        /* dct['ins_t'], dct['del_t'] */
    int32_t haploid;                            /* 1: get_indel_testing_candidates_haploid (generate_indel_pileups_haploid.py:185-241):
                                                   one read set, HP tags ignored, frequencies over all reads of the column */
    int32_t impute;                             /* 1: dct['impute_indel_phase'] (generate_indel_pileups.py:278-284, diploid only): columns
                                                   without mincov reads on both haplotypes but >= 2*mincov reads in total whose share
                                                   of '-'/'*' or '+' pileup strings reaches del_t / ins_t get col_type 2; the caller
                                                   groups the reads of these columns (:285-304) */
} nc_indel_scan_params;

int nc_indel_scan(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *events, const uint8_t *excl_dev,
                  int32_t start, int32_t end, const nc_indel_scan_params *params, int8_t *col_type_host);

/* The same for many chunks of one contig in one call (chunks keep their per-chunk semantics: the window deques start
 * empty at each chunk's first column, as in the reference, which calls the function once per chunk): col_type of chunk c
 * is written at col_type_host + col_off[c] (max(1,start_c) .. end_c).  Chunk lists ascending in start and end (chunks may
 * overlap or abut) run in the same kernel launches, the chunk being a grid dimension, in groups bounded by a 6 GiB
 * workspace; a list in any other order is cut into its ascending runs.  nc_indel_scan is the one-chunk form.  col_type:
 * -1 none, 0 long-window rule (:266), 1 small-window rule (:271), 2 (params.impute) impute_indel_phase candidate. */
int nc_indel_scan_batch(nc_ctx *ctx, const nc_readpack *pack, const nc_indel_events *events, const uint8_t *excl_dev,
                        int32_t n_chunks, const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *params,
                        int8_t *col_type_host, const int64_t *col_off);

/* ------------------------------------------------------------------ BGZF / BAM (+ .bai linear index) ingest, host side
 * Replaces the pysam/htslib objects of the reference (pysam.Samfile(...).fetch / .pileup, generate_SNP_pileups.py:
 * 134-164; generate_indel_pileups.py:147,178-188,213-235): a coordinate-sorted BAM is decoded straight into the
 * read-major arrays nc_pack_plan / nc_pack_fill and nc_indel_scan take.  Per alignment: reference span
 * [start, end) (1-based), BAM flag, one code per spanned reference position (deletions and reference skips = 4),
 * the '+n' / '-n' pileup markers as events on the column BEFORE the insertion / deletion, HP and PS tags (0 if
 * absent), the read name and (keep_seq != 0) the query bases as codes.  Unmapped reads are skipped; every other flag
 * is returned so that the caller applies the pileup flag filter.  All pointers stay valid until nc_decoded_free.
 */
typedef struct nc_bam nc_bam;
typedef struct nc_decoded nc_decoded;
typedef struct {
    int32_t n_reads;
    const int32_t *start, *end, *flag;   /* [n_reads] */
    const int64_t *off;                  /* [n_reads+1] into codes */
    const uint8_t *codes;
    int64_t n_codes;
    const int32_t *ev_off;               /* [n_reads+1] */
    const int32_t *ev_pos, *ev_len;      /* ev_len > 0 insertion, < 0 deletion */
    int64_t n_events;
    const uint8_t *hap;                  /* HP tag: 0 / 1 / 2 */
    const int32_t *ps;                   /* PS tag or 0 */
    const int64_t *seq_off;              /* [n_reads+1] into seq (all zero when keep_seq == 0) */
    const uint8_t *seq;
    int64_t n_seq;
    const int32_t *name_off;             /* [n_reads+1] into names (NUL-terminated strings) */
    const char *names;
    const int32_t *qstart;               /* [n_reads] query index of the first aligned base (leading soft clip / insertion) */
} nc_decoded_arrays;

int nc_bam_open(const char *path, nc_bam **out);
int nc_bam_close(nc_bam *bam);
int nc_bam_n_refs(nc_bam *bam, int32_t *n_refs, int32_t *has_index);
int nc_bam_ref(nc_bam *bam, int32_t i, const char **name, int32_t *length);
const char *nc_bam_error(const nc_bam *bam);
/* host threads that inflate BGZF blocks for this handle (0 = all cores up to 32; 1 when many handles decode regions in parallel) */
int nc_bam_set_threads(nc_bam *bam, int32_t n);
/* alignments of reference `tid` overlapping [beg1, end1] (1-based, inclusive), in coordinate order */
int nc_bam_decode(nc_bam *bam, int32_t tid, int32_t beg1, int32_t end1, int32_t keep_seq, nc_decoded **out);
/* The same interval decoded as `n_regions` equal sub-intervals by as many host threads (own file handle each, seek through
 * the .bai linear index, a read belongs to the sub-interval it starts in) and merged in parallel: identical result to one
 * nc_bam_decode call, several times the throughput on long intervals.  Needs the .bai for the seeks to pay. */
int nc_bam_decode_regions(const char *path, int32_t tid, int32_t beg1, int32_t end1, int32_t keep_seq, int32_t n_regions,
                          nc_decoded **out);
int nc_decoded_view(const nc_decoded *d, nc_decoded_arrays *view);
int nc_decoded_free(nc_decoded *d);
/* Bit set in nc_decoded_arrays.flag (above the 16 BAM flag bits) for an alignment whose CIGAR holds a reference skip (N). */
#define NC_FLAG_REFSKIP 0x10000
/* Inputs the library does not reproduce, reported instead of silently accepted: kept (keep[r] != 0,
 * NULL = all) alignments with a reference skip -- the reference's code table raises KeyError on their '>' / '<' pileup symbols
 * (generate_SNP_pileups.py:104,175) -- and pairs of kept alignments with the same read name that overlap on the reference -- the
 * reference's per-column dicts are keyed by name (:175,185,208), the read-major pack keeps them apart.  Returns NC_ERR_UNSUPPORTED
 * when either count is non-zero, NC_OK otherwise. */
int nc_decoded_check(const nc_decoded *d, const uint8_t *keep, int64_t *n_refskip, int64_t *n_dup_overlap);
/* Kept alignments that share a read name (ABI 11): gid[r] (host [n_reads]) = index of the first kept alignment with r's name when more than one
 * kept alignment carries it, else -1; *n_shared = how many alignments that is.  What nc_snp_set_mates' table is built from (the reference keys
 * its per-column pileups, strand table and neighbour lookups by name: generate_SNP_pileups.py:141-143,175,185,223,232). */
int nc_decoded_name_groups(const nc_decoded *d, const uint8_t *keep, int32_t *gid, int64_t *n_shared);

/* ------------------------------------------------------------------ indel pass 2, host side
 * nc_indel_slices replaces the per-read loop of generate_indel_pileups.py:329-338 at the anchor columns chosen by pass 1:
 * for every alignment of `d` (decoded with keep_seq) covering anchor_pos[a] (1-based; reads in coordinate order, like
 * pcol.pileups) the window query_sequence[max(0, q - window_before) : q + window_after], q = pysam's
 * query_position_or_next at that column (inside a deletion: the next aligned query base).  `keep` (per read, may be NULL)
 * applies the pileup flag filter.  Bases are codes A=0 G=1 T=2 C=3 other=4.  Parity with pysam/htslib is unpinned (absent
 * from this image): pinned against an independent CIGAR walk in tests/. */
typedef struct nc_slices nc_slices;
typedef struct {
    int32_t n_anchor;
    const int32_t *anchor_off;           /* [n_anchor+1] into read_idx / seq_off */
    const int32_t *read_idx;             /* [n_slices] index of the alignment in the decoded set */
    const int64_t *seq_off;              /* [n_slices+1] into seq */
    const uint8_t *seq;
    int64_t n_slices;
} nc_slices_arrays;
int nc_indel_slices(const nc_decoded *d, int32_t n_anchor, const int32_t *anchor_pos, int32_t window_before, int32_t window_after,
                    const uint8_t *keep, nc_slices **out);
int nc_slices_view(const nc_slices *s, nc_slices_arrays *view);
int nc_slices_free(nc_slices *s);

/* Pass 2 of get_indel_testing_candidates up to the aligner call (generate_indel_pileups.py:306-348; haploid :243-262) for all
 * anchors of a chunk at once: reference windows from `contig` (1-based position p = contig[p-1], only positions in
 * [ref_lo, ref_hi] count, :174; an anchor whose window holds anything but upper-case AGTC is skipped, :325-327), read windows
 * of `window_after` query bases from query_position_or_next (:331), the split into hap0 / hap1 / all reads by HP tag -- or,
 * where imp_idx[a] = k >= 0, by the imputed read collections imp_reads[imp_off[2k] .. imp_off[2k+1]) and
 * [imp_off[2k+1] .. imp_off[2k+2]) (read indices; :310-318) --, at most `maxcov` reads per set (the first in pileup order;
 * the reference samples at random, unseeded, :19-20), at least 2 / 2 / mincov reads (:48, :345; haploid: one set of >= mincov).
 * The result holds the flat arrays nc_star_msa_tensor takes (sets_per_anchor consecutive sets per kept anchor). */
typedef struct nc_pass2 nc_pass2;
typedef struct {
    int32_t n_kept;
    const int32_t *anchor_idx;    /* [n_kept] index into `anchors` */
    const int32_t *first0;        /* [n_kept] first read of set 0 in pileup order (phase lookup, :349) */
    int32_t sets_per_anchor;      /* 3 (hap0, hap1, all) or 1 (haploid) */
    int32_t n_sets;
    const int32_t *set_read0;     /* [n_sets + 1] */
    int32_t n_alignments;
    const int32_t *read_off;      /* [n_alignments + 1] */
    const char *reads;            /* ASCII, AGTC or N */
    const int32_t *ref_off;       /* [n_sets + 1] */
    const char *refs;
    int32_t max_cols;             /* bound on the alignment columns of any set (max_cols argument of nc_star_msa_tensor) */
    const int32_t *al_dup;        /* [n_alignments] an earlier alignment of the same read window to the same reference window (a read
                                     of the third set that is in a haplotype set too), or -1: nc_star_msa_tensor_dup's al_dup */
} nc_pass2_arrays;
int nc_indel_pass2_sets(const nc_decoded *d, const uint8_t *keep, int32_t n_anchor, const int32_t *anchors, const char *contig,
                        int64_t chrom_len, int32_t ref_lo, int32_t ref_hi, int32_t window_after, int32_t mincov, int32_t maxcov,
                        int32_t haploid, const int32_t *imp_idx, const int32_t *imp_off, const int32_t *imp_reads, nc_pass2 **out);
int nc_pass2_view(const nc_pass2 *p, nc_pass2_arrays *view);
int nc_pass2_free(nc_pass2 *p);


/* ------------------------------------------------------------------ indel path, device resident
 * The whole of get_indel_testing_candidates (generate_indel_pileups.py:129-371; haploid generate_indel_pileups_haploid.py) for
 * every chunk of a contig without the host in the loop: pass 1 (nc_indel_scan_batch's kernels), the order-dependent anchor
 * selection (:249,266-275), the read sets of pass 2 (:306-348) taken from the read pack's tile index, the query windows
 * (:331) rebuilt from the position-addressed codes + the indel events + the bases that have no reference column, the star
 * alignment (nc_star_msa_tensor_dup's algorithm: same recurrences and tie rules), msa()'s histogram / consensus / tensor
 * (:57-71) and allele_prediction (:77-127).  Results equal nc_indel_pass2_sets -> nc_star_msa_tensor_dup ->
 * nc_allele_prediction_device on the same inputs.  dct['impute_indel_phase'] (params.impute, diploid; :278-304) since round 6: the reads of
 * every col_type-2 column are grouped by pileup string in HBM (letter + event length + inserted bases, keyed by a 64-bit hash), the
 * column becomes a small-window anchor 10 bp upstream when both sides of the rule hold mincov reads, and the anchor's pass-2 sets are
 * those two groups instead of the HP tags; needs reads->ins_off / ins_bases; NC_ERR_CAPACITY on a col_type-2 column deeper than 512 reads.
 *
 * nc_indel_pack_build (host): the per-read arrays of the KEPT reads in pack order, from a decoded contig (keep_seq != 0):
 * events / HP / PS, the inserted bases of every insertion event (ins_off [n_events + 1] into ins_bases; deletions own empty
 * ranges), and up to `tail_cap` query bases following the last aligned one (trailing soft clip; tail_off [n_reads + 1]):
 * query_sequence[q : q + window] runs into them near a read's end.  Bases are codes A0 G1 T2 C3 other 4.  read_flag bit 0: a
 * record without bases (SEQ '*'): its windows are empty. */
typedef struct nc_indel_pack_h nc_indel_pack_h;
typedef struct {
    int32_t n_reads;
    const int32_t *ev_off;     /* [n_reads + 1] */
    const int32_t *ev_pos, *ev_len;
    int64_t n_events;
    const int32_t *ins_off;    /* [n_events + 1] */
    const uint8_t *ins_bases;
    int64_t n_ins_bases;
    const int32_t *tail_off;   /* [n_reads + 1] */
    const uint8_t *tail_bases;
    int64_t n_tail_bases;
    const int32_t *read_ps;    /* [n_reads] PS tag or 0 */
    const uint8_t *read_hap;   /* [n_reads] HP tag 0 / 1 / 2 */
    const uint8_t *read_flag;  /* [n_reads] */
} nc_indel_pack_arrays;
int nc_indel_pack_build(const nc_decoded *d, const uint8_t *keep, int32_t tail_cap, nc_indel_pack_h **out);
int nc_indel_pack_view(const nc_indel_pack_h *p, nc_indel_pack_arrays *view);
int nc_indel_pack_free(nc_indel_pack_h *p);

/* The same arrays in HBM (+ what maps a tile entry to its read: slot_off / rd_start of the wire pack, nc_wire_arrays) */
typedef struct {
    int32_t n_reads;
    const int64_t *slot_off;   /* dev [n_reads + 1] */
    const int32_t *rd_start;   /* dev [n_reads] */
    const int32_t *rd_end;     /* dev [n_reads] */
    const int32_t *ev_off, *ev_pos, *ev_len;
    const int32_t *ins_off;
    const uint8_t *ins_bases;
    const int32_t *tail_off;
    const uint8_t *tail_bases;
    const int32_t *read_ps;
    const uint8_t *read_hap, *read_flag;
} nc_indel_reads;

/* Step 1: pass 1 + anchor selection + read sets of all chunks (ascending, one contig) -> *n_sites candidate sites that reach
 * the CNN (chunk-major, ascending position inside a chunk; an anchor shared by two chunks appears once per chunk, like
 * the reference's per-chunk calls), *n_alignments read windows to align.  ref_code as nc_snp_scan (4 = not an upper-case
 * AGTC: an anchor whose window [p, p + window_after] touches one is skipped, :325-327; exclusions must NOT be folded in);
 * chrom_len = length of the contig.  The state stays in the context for step 2.  Synchronises. */
int nc_indel_sites_plan(nc_ctx *ctx, const nc_readpack *pack, const uint8_t *ref_code_dev, int32_t ref_pos0, int32_t ref_len,
                        int64_t chrom_len, const nc_indel_reads *reads, const uint8_t *excl_dev, int32_t n_chunks,
                        const int32_t *starts, const int32_t *ends, const nc_indel_scan_params *params, int32_t window_after,
                        int32_t maxcov, int32_t *n_sites, int64_t *n_alignments);
/* Scoring of the star alignment (gap open, gap extend, match, mismatch; default 25, 1, 20, -10 = the product's _lib.STAR_SCORING;
 * allele_prediction always uses the reference's call-site values 9, 1, 20, -10).  Persists on the context. */
int nc_indel_sites_scoring(nc_ctx *ctx, int32_t open, int32_t extend, int32_t match, int32_t mismatch);
/* Step 2: windows -> alignment -> tensors + consensus -> alleles, in groups of sites bounded by the traceback workspace.
 * x_dev f32 [n_sites][sets * 5][128][2] (sets = 3: hap0 | hap1 | all reads stacked as indelCaller.py:83 does; haploid 1) is
 * the input of nc_indel_forward.  Does not synchronise. */
int nc_indel_sites_run(nc_ctx *ctx, float *x_dev);
/* Step 3: per-site results to host arrays (any may be NULL): pos, chunk (index into the plan's chunk list), var_type (0 long
 * window / 1 small window rule), phase (PS tag of the first read of the first set, :349; 0 = none), ref_len / alt_len
 * [n_sites][sets] (allele_prediction's REF = window[:ref_len], ALT = consensus[:alt_len]; ref_len < 0: (None, None)), and
 * *n_alt_bytes = sum of max(alt_len, 0).  Synchronises.  NC_ERR_CAPACITY: a read set needed more than 1024 alignment
 * columns or a window exceeded the register aligner (the caller falls back to the host route). */
int nc_indel_sites_fetch(nc_ctx *ctx, int32_t *pos, int32_t *chunk, int32_t *var_type, int32_t *phase, int32_t *ref_len,
                         int32_t *alt_len, int64_t *n_alt_bytes);
/* the ALT prefixes back to back in (site, set) order, codes A0 G1 T2 C3 */
int nc_indel_sites_fetch_alt(nc_ctx *ctx, uint8_t *alt_bases, int64_t cap);
/* per-stage HIP-event milliseconds of the last plan + run on this context (timing mode 1): [0] pass 1 + anchors + sets,
 * [1] query windows, [2] alignment fill, [3] traceback, [4] tensors + consensus, [5] allele alignment + extraction */
int nc_indel_sites_stage_ms(nc_ctx *ctx, float *ms6, int64_t *cells2 /* [0] DP cells of the star alignments, [1] of the allele alignments */);
/* The star alignment behind msa() (generate_indel_pileups.py:24-44; MUSCLE in the reference) runs on a BAND of 32 or 64 diagonals around
 * the diagonals the read's own CIGAR visits inside the window; an alignment whose band would be wider, or whose banded path touches an edge
 * diagonal, runs on the full matrix.  Counts of the last plan + run + fetch: [0] alignments on 32 diagonals, [1] on 64, [2] on the full
 * matrix because of their width, [3] re-run on the full matrix after an edge touch, [4] DP cells of the banded alignments [0] + [1]
 * ((n1 + n2) x 16 or 32 each; nc_indel_sites_stage_ms' cells2[0] keeps counting n1 x n2 for every alignment), [5] 0.  NC_PIPE_BAND=0 in the
 * environment turns the band off. */
int nc_indel_sites_band_stats(nc_ctx *ctx, int64_t *stats6);
/* mode 1 / 0: band on / off for this context (-1: the environment's setting, the default = on); margin = diagonals kept free on either side
 * of the CIGAR's range, 1 .. 15 (0: the default, 6; NC_PIPE_BAND_MARGIN) */
int nc_indel_sites_band(nc_ctx *ctx, int32_t mode, int32_t margin);

/* Indel genotype rules + VCF record text (indelCaller.py:87-152, haploid :173-179), host, printf-free: sites in the order of
 * nc_indel_sites_fetch; `prev` (overlap suppression, :93) restarts at every chunk, as each chunk is one call of indel_run's
 * loop body.  probs f32 [n][4] (hom-ref, hom-alt, het-ref, het-alt) or, haploid, [n][1]; contig = the reference bases
 * (position p = contig[p-1]).  out receives the records back to back; chunk_txt_off [n_chunks + 1] (may be NULL) the byte
 * offset of every chunk's first record. */
int nc_indel_vcf_format(const char *chrom, int64_t n, const int32_t *pos, const int32_t *chunk, int32_t n_chunks, const float *probs,
                        int32_t sets, const int32_t *ref_len, const int32_t *alt_len, const uint8_t *alt_bases, const int32_t *phase,
                        const char *contig, int64_t chrom_len, int32_t haploid, char *out, int64_t cap, int64_t *n_bytes,
                        int64_t *chunk_txt_off);

/* Global alignment with affine gaps, the call parasail.nw_trace(alt, ref, 9, 1, matrix_create('AGTC', 20, -10)) of
 * generate_indel_pileups.py:10,79: a gap of length k costs open + (k-1)*extend.  Writes the CIGAR as (op, count) pairs
 * with parasail's op codes ('=' 7, 'X' 8, 'I' 1 = base of s1 only, 'D' 2 = base of s2 only).  Tie-breaking (parity with
 * parasail unpinned: the library is absent from this image): a cell prefers the diagonal, then D, then I; a gap prefers
 * extension over opening on equal scores.  NC_ERR_CAPACITY if `cap` pairs do not suffice (n1 + n2 always does). */
int nc_nw_cigar(const char *s1, int32_t n1, const char *s2, int32_t n2, int32_t open, int32_t extend, int32_t match, int32_t mismatch,
                int32_t *ops, int32_t *counts, int32_t cap, int32_t *n_ops);

/* allele_prediction(alt, ref_seq, max_range) of generate_indel_pileups.py:77-127: aligns with nc_nw_cigar (9, 1, 20, -10) and
 * walks the CIGAR exactly as the reference does.  Returns the lengths of the (REF, ALT) prefixes; *ref_len = -1 means the
 * reference returns (None, None). */
int nc_allele_prediction(const char *alt, int32_t n_alt, const char *ref_seq, int32_t n_ref, int32_t max_range, int32_t *ref_len,
                         int32_t *alt_len);
/* the same for n independent (alt, ref) pairs, on the usable host cores: alt i = alts[alt_off[i] .. alt_off[i+1]), ref i likewise */
int nc_allele_prediction_batch(int32_t n, const char *alts, const int32_t *alt_off, const char *refs, const int32_t *ref_off,
                               const int32_t *max_range, int32_t *ref_len, int32_t *alt_len);
/* The same on the device (the 16-lane register aligner of nc_star_msa_tensor with parasail's scoring, one lane per alignment for
 * the traceback and the allele extraction): identical results.  Host arrays in and out.  NC_ERR_CAPACITY when a reference window
 * is longer than 272 bases, a consensus longer than 1000, or a string empty: use nc_allele_prediction_batch then. */
int nc_allele_prediction_device(nc_ctx *ctx, int32_t n, const char *alts, const int32_t *alt_off, const char *refs, const int32_t *ref_off,
                                const int32_t *max_range, int32_t *ref_len, int32_t *alt_len);

/* Star alignment of a read set to its reference window: replaces the MUSCLE subprocess of
 * generate_indel_pileups.py:24-44 with pairwise Gotoh alignments (anchored at the window start, free tail; scoring as above)
 * merged in reference coordinates -- longest insertion per reference slot, shorter ones left-justified.  NOT MUSCLE's
 * algorithm: rows are not comparable with its output, only the resulting calls are.
 * reads: concatenated characters, read r = reads[read_off[r] .. read_off[r+1]); rows: [n_reads][col_cap] symbols
 * A=0 G=1 T=2 C=3 gap=4 other=5, the first *n_cols columns of each row valid; ref_row [col_cap] likewise.
 * NC_ERR_CAPACITY (with *n_cols set) when col_cap is too small. */
int nc_star_msa(int32_t n_reads, const char *reads, const int32_t *read_off, const char *ref, int32_t n_ref, int32_t open,
                int32_t extend, int32_t match, int32_t mismatch, int32_t col_cap, uint8_t *rows, uint8_t *ref_row, int32_t *n_cols);

/* The same star alignment on the device, for many read sets at once, followed by the rows -> tensor kernel (K8): one lane per
 * read fills the Gotoh DP in HBM and walks it back, one workgroup per set merges the alignments into rows.  Bit-identical rows
 * to nc_star_msa.  Host arrays in: reads (concatenated characters, read a = [read_off[a], read_off[a+1])), the reads of set s
 * are set_read0[s] .. set_read0[s+1]; refs / ref_off: the reference window of every set.  Out: x_dev [n_sets][5][128][2] (device),
 * cns_host [n_sets][max_cols] (consensus symbols, gaps kept as 4, NC_CODE_ABSENT padding), n_cols_host [n_sets].  rows_host /
 * ref_rows_host (optional, with per-set byte offsets; set s needs n_reads[s] * n_cols[s] resp. n_cols[s] bytes): the rows. */
int nc_star_msa_tensor(nc_ctx *ctx, int32_t n_sets, const char *reads, const int32_t *read_off, const int32_t *set_read0,
                       const char *refs, const int32_t *ref_off, int32_t open, int32_t extend, int32_t match, int32_t mismatch,
                       int32_t max_cols, float *x_dev, uint8_t *cns_host, int32_t *n_cols_host, uint8_t *rows_host,
                       const int64_t *rows_host_off, uint8_t *ref_rows_host, const int64_t *ref_rows_host_off);
/* The same with a duplicate map: al_dup[a] = b < a when alignment a is the same read window against the same reference window
 * as alignment b (what nc_pass2_arrays.al_dup holds), else -1; NULL = none.  A duplicate is not aligned again -- the set
 * kernels read b's traceback -- so the results are those of nc_star_msa_tensor on the same arrays. */
int nc_star_msa_tensor_dup(nc_ctx *ctx, int32_t n_sets, const char *reads, const int32_t *read_off, const int32_t *set_read0,
                           const char *refs, const int32_t *ref_off, int32_t open, int32_t extend, int32_t match, int32_t mismatch,
                           int32_t max_cols, float *x_dev, uint8_t *cns_host, int32_t *n_cols_host, uint8_t *rows_host,
                           const int64_t *rows_host_off, uint8_t *ref_rows_host, const int64_t *ref_rows_host_off, const int32_t *al_dup);

/* The consensus rows of nc_star_msa_tensor[_dup] as the strings msa() returns (generate_indel_pileups.py:58-61: gap symbols
 * removed; 0..3 -> AGTC, 5.. -> N): set s = out[off[s] .. off[s+1]), concatenated; `out` needs at most n_sets * max_cols bytes.
 * Host code on the usable cores. */
int nc_consensus_strings(const uint8_t *cns, int32_t n_sets, int32_t max_cols, const int32_t *n_cols, char *out, int64_t *off);

/* ------------------------------------------------------------------ SNP genotype rules + VCF record text (host)
 * Replaces the per-site Python loop of snpCaller.caller (snpCaller.py:113-198).  probs f32 [n][4]
 * (diploid: class-1 probability of the A,G,T,C heads; haploid: 4-way softmax); order i32 [n][4] = np.argsort(probs,
 * axis=1) computed by the caller (tie behaviour of the reference, quirk E15; unused when haploid); ref = reference
 * base code; dp, freq, fwd / rev i32 [n][4] as produced by the scan / featurize calls.  Writes the records
 * back-to-back into `out`; NC_ERR_CAPACITY if `cap` (>= 400 bytes per record recommended) is too small. */
int nc_snp_vcf_format(const char *chrom, int64_t n, const int32_t *pos, const int32_t *ref, const float *probs,
                      const int32_t *order, const int32_t *dp, const double *freq, const int32_t *fwd, const int32_t *rev,
                      int32_t haploid, char *out, int64_t cap, int64_t *n_bytes);

/* Ascending argsort of n rows of 4 probabilities (np.argsort(..., axis=1), snpCaller.py:120), multi-threaded.  Rows that
 * contain equal values are listed in tie_idx[0..*n_ties) (NC_ERR_CAPACITY, with *n_ties set, if tie_cap is too small):
 * numpy's order of tied elements is implementation dependent, so a caller that needs the
 * reference's exact records re-sorts those rows with numpy before passing `order` to nc_snp_vcf_format. */
int nc_argsort4(const float *probs, int64_t n, int32_t *order, int64_t *n_ties, int64_t *tie_idx, int64_t tie_cap);

/* BGZF compression of a byte stream on all host cores (the `| bgzip >` of snpCaller.py:284-285): blocks of 0xff00 payload
 * bytes + the EOF block.  block_coff[b] (b = 0..*n_blocks, may be NULL) = compressed offset of block b, from which virtual
 * file offsets (coffset << 16 | offset in block) for an index are formed.  cap >= n + n/100 + 64*(*n_blocks) + 64 suffices. */
int nc_bgzf_compress(const uint8_t *data, int64_t n, int32_t level, uint8_t *out, int64_t cap, int64_t *n_out,
                     int64_t *block_coff, int64_t blk_cap, int64_t *n_blocks);

/* A whole BGZF file (bgzip output: a BED, a VCF, ...) inflated into `out` by the BGZF layer of the BAM reader: every block's
 * CRC-32 and ISIZE are checked.  *n_out = the inflated size, also when NC_ERR_CAPACITY says `cap` was too small (cap 0 / out
 * NULL: size query).  NC_ERR_ARG: not readable or not BGZF.  Reads the `exclude_bed` file the reference opens with
 * pysam.TabixFile (generate_SNP_pileups.py:113-116). */
int nc_bgzf_read_file(const char *path, uint8_t *out, int64_t cap, int64_t *n_out);

/* ------------------------------------------------------------------ synthetic indel workload (bench.py / tests tooling, not product)
 * SURVEY.md 8d's generator for the indel configs, in HBM: a contig with planted het / hom SNPs and indels (lengths 1..max_len, never
 * overlapping) and ONT-like reads that carry their haplotype's variants plus sequencing noise (substitutions; noise deletions
 * and insertions as EVENTS, as a decoded ONT BAM has them).  Pure functions of (seed, read, position).
 * nc_synth_indel_truth: ref_dev [L + 1] base code of position p at index p; hap_base_dev [2][L + 1]; hap_indel_dev int8 [2][L + 1]: the
 * indel that FOLLOWS position p on each haplotype (+ insertion / - deletion length, 0 none).
 * nc_synth_indel_reads: fill = 0 counts events / inserted bases per read; fill = 1 (with the exclusive prefix sums ev_off /
 * ins_off_read [n_reads]) writes the position-addressed codes in nc_pack_fill's slot layout (slot_off per read), ev_pos / ev_len /
 * ins_off per event and the inserted bases.  All pointers dev. */
int nc_synth_indel_truth(nc_ctx *ctx, int64_t L, uint64_t seed, double het_snp, double hom_snp, double het_indel, double hom_indel,
                         int32_t max_len, uint8_t *ref_dev, uint8_t *hap_base_dev, int8_t *hap_indel_dev);
int nc_synth_indel_reads(nc_ctx *ctx, int64_t L, uint64_t seed, double p_sub, double p_del, double p_ins, double carry, int32_t n_reads,
                         const int32_t *start_dev, const int32_t *end_dev, const int64_t *slot_off_dev, const uint8_t *hap_dev,
                         const uint8_t *hap_base_dev, const int8_t *hap_indel_dev, int32_t fill, int32_t *ev_cnt_dev, int32_t *ins_cnt_dev,
                         const int32_t *ev_off_dev, const int32_t *ins_off_read_dev, uint8_t *codes_dev, int32_t *ev_pos_dev,
                         int32_t *ev_len_dev, int32_t *ins_off_dev, uint8_t *ins_bases_dev);

#ifdef __cplusplus
}
#endif
#endif /* NANOCALLER_HIP_H */
