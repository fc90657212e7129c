// Internal header of libnanocaller_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/nanocaller_hip.h"

// Inclusive prefix sum over the 64 lanes of a wave in six DPP additions (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then lane 15 of rows 0 and 2
// into rows 1 and 3, then lane 31 into rows 2 and 3): no LDS traffic, no per-step compare -- the __shfl_up form is six ds_bpermute_b32 round trips
// with a compare and a select each.
__device__ __forceinline__ int32_t nc_wave_incl_scan(int32_t v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
    return v;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct nc_weights {
    float *dev = nullptr;      // canonical flat blob
    size_t n = 0;
    float *packed = nullptr;   // kernel-specific repack (see nc_cnn.hip)
    size_t n_packed = 0;
    void *packed_h = nullptr;  // fp16x3 fragments of the trunk kernel
    float x_limit = 0.0f;      // largest |input value| for which the L1 norms of conv1-3 prove that no activation reaches the fp16 clamp
};

// words per tile entry of the K7 cursor table (k_entry_cursors, nc_indel.hip) for spt 1024-column blocks per tile
#define NC_ENT_CUR_PITCH(spt) (2 * (spt) + 3)

struct nc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    int timing = 0;                           // 0 off; 1 stage timers + trunk launches; 2 trunk launches only (events on the dispatch packets, no barrier packets)
    bool x_i16 = false;            // SNP tensors between featuriser and CNN as int16 instead of fp32 (nc_set_tensor_format)
    bool cnn_exact_fp32 = false;   // false: fp16x3 split-precision trunk (default); true: exact fp32 MFMA trunk
    bool k10_lds_set[2] = {false, false};   // k10_indel_trunk_h3<15 / 5>: dynamic LDS limit raised on this device
    bool huff_lds_set = false;              // k_huff: the same
    size_t k7_budget = 0;                   // bytes of K7 workspace per group of chunks (set from this context's device at its first plan)
    // alignments that share a read name (nc_snp_set_mates): borrowed device pointers, read by the next nc_snp_featurize calls
    int32_t n_mates = 0;
    const int64_t *mate_key = nullptr;
    const int32_t *mate_rec = nullptr;
    bool k7_budget_shrunk = false;          // the budget was cut to half of what was free at some plan: restored when memory allows again
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms[6] = {0, 0, 0, 0, 0, 0};   // 0 scan, 1 featurize, 2 cnn stage, 3 indel, 4 trunk kernel total, 5 trunk launches
    hipEvent_t kev[128] = {nullptr};          // per-launch event pairs of the trunk kernel (timing mode)
    int n_kev = 0;
    hipEvent_t tev[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // stage timers (scan, featurize, CNN, indel)
    bool tev_pending[4] = {false, false, false, false};
    bool kev_pending = false;
    double sum_ms[6] = {0, 0, 0, 0, 0, 0};   // running totals of last_ms[] since nc_enable_timing(1): calls may be left in flight
    int64_t sum_n[6] = {0, 0, 0, 0, 0, 0};   // while the next one is enqueued, their timers are folded in before the events are re-used
    hipEvent_t drain_ev[4] = {nullptr};       // batch-complete events of nc_snp_forward_drain
    hipEvent_t scale_ev = nullptr;            // recorded behind nc_snp_scale's kernel for nc_snp_chunk_depth_async
    hipEvent_t scan_ev = nullptr;             // recorded behind the last kernel of nc_snp_scan (its site arrays are complete)

    // scan results (device)
    DevBuf stage_nbr, stage_cpos, stage_cn, stage_calt;   // per-tile staging
    DevBuf tile_cnt, tile_pre;                            // int2 per tile
    DevBuf nbr_pos, cand_pos, cand_n, cand_alt;
    DevBuf chunk_start, chunk_end, chunk_lo, chunk_cnt, chunk_off;
    DevBuf site_pos, site_chunk, site_n, site_alt;
    DevBuf totals;                                        // int32[4]: n_nbr, n_cand, n_sites
    DevBuf cnn_a, cnn_b, cnn_c;                           // CNN intermediates
    DevBuf chunk_depth;                                   // double per chunk
    DevBuf nbr_idx;                                       // coarse index over nbr_pos
    DevBuf indel_ws;                                      // indel window-scan workspace
    DevBuf indel_ent_read;                                // read index of every tile entry + its event cursors per 1024-column block (k_entry_cursors)
    const void *indel_ent_of = nullptr;                   // the tile index (tile_ent) those tables were last made for, NULL: none (pass 1 ran in another form)
    int indel_ent_spt = 0;                                // 1024-column blocks per tile of that index
    DevBuf msa_reads, msa_read_off, msa_read_set, msa_refs, msa_ref_off;   // device star alignment (nc_msa.hip): inputs,
    DevBuf msa_dup;                                        // duplicate map + list of alignments to compute (nc_star_msa_tensor_dup)
    DevBuf msa_rows_hf, msa_hcol, msa_tb, msa_trace, msa_cols, msa_out;    // DP rows / last column / traceback bytes / alignments / columns / rows
    int32_t n_nbr = 0, n_cand = 0, n_sites = 0, n_chunks = 0;
    bool have_scan = false;

    nc_weights w[4];
    struct nc_pipe_state *pipe = nullptr;   // device-resident indel pipeline (nc_pipe.hip): plan state, workspaces, results
    uint8_t *range_sites = nullptr;         // nc_cnn_range_watch: device byte per site of the next nc_snp_forward, set when the site's
                                            // scaled tensor exceeds the model's x_limit (the caller re-runs those sites on the exact trunk)

    // The small transfers the host waits for in the middle of a step go through kernels that read / write page-locked host
    // memory directly, NOT through hipMemcpyAsync: on this platform every hipMemcpyAsync of either direction queues in order
    // behind a large H2D copy in flight (7 ms for a contig's wire pack), measured with tools/exp_sdma.py.
    int32_t *mbox = nullptr;       // pinned host, 64 int32: scan totals
    // nc_snp_scan_begin .. nc_snp_scan_end
    bool scan_begun = false;
    int scan_tile = 0, scan_n_tiles = 0, scan_n_chunks = 0, scan_cap_nbr = 0, scan_cap_cand = 0;
    hipEvent_t scan_tot_ev = nullptr;          // recorded behind the copy of the totals into the mailbox
    uint8_t *stage_h = nullptr;    // pinned host ring for small host -> device arrays (chunk bounds)
    int stage_turn = 0;
};

// device -> host: a copy kernel on `st` when `host` is device-accessible page-locked memory, hipMemcpyAsync otherwise
int nc_d2h(nc_ctx *ctx, void *host, const void *dev, size_t bytes, hipStream_t st);
// host -> device for small arrays (<= NC_STAGE_SLOT bytes): through the pinned ring and a copy kernel
int nc_h2d_small(nc_ctx *ctx, void *dev, const void *host, size_t bytes, hipStream_t st);
#define NC_D2H_KERNEL_MAX 65536
#define NC_STAGE_SLOT 16384
#define NC_STAGE_SLOTS 16

// the same for arrays of any size, in ring-slot pieces (descriptor tables of a few tens of kilobytes: a hipMemcpyAsync would wait for
// a contig's upload in flight).  The source may be released on return; the stream is drained before the ring would wrap.
inline int nc_h2d_pieces(nc_ctx *ctx, void *dev, const void *host, size_t bytes, hipStream_t st)
{
    int in_flight = 0;
    for (size_t o = 0; o < bytes; o += NC_STAGE_SLOT) {
        if (++in_flight >= NC_STAGE_SLOTS - 2) {
            if (hipStreamSynchronize(st) != hipSuccess) return NC_ERR_HIP;
            in_flight = 1;
        }
        const size_t n = bytes - o < (size_t)NC_STAGE_SLOT ? bytes - o : (size_t)NC_STAGE_SLOT;
        const int rc = nc_h2d_small(ctx, (char *)dev + o, (const char *)host + o, n, st);
        if (rc != NC_OK) return rc;
    }
    return NC_OK;
}
// device -> page-locked host in copy-kernel pieces (see nc_d2h): for arrays the host waits for while an upload is in flight
inline int nc_d2h_pieces(nc_ctx *ctx, void *host_pinned, const void *dev, size_t bytes, hipStream_t st)
{
    for (size_t o = 0; o < bytes; o += NC_D2H_KERNEL_MAX) {
        const size_t n = bytes - o < (size_t)NC_D2H_KERNEL_MAX ? bytes - o : (size_t)NC_D2H_KERNEL_MAX;
        const int rc = nc_d2h(ctx, (char *)host_pinned + o, (const char *)dev + o, n, st);
        if (rc != NC_OK) return rc;
    }
    return NC_OK;
}

inline int nc_fail(nc_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define NC_HIP(ctx, call)                                                                           \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return nc_fail(ctx, NC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                           __FILE__, __LINE__);                                                     \
    } while (0)

inline int nc_ensure(nc_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return NC_OK;
    if (b.p) {
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NC_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return nc_fail(ctx, NC_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    b.cap = want;
    return NC_OK;
}

#define NC_TRY(x)                \
    do {                         \
        int rc_ = (x);           \
        if (rc_ != NC_OK) return rc_; \
    } while (0)

// resolves pending event pairs of stage `which` (0..3) or of the trunk launches (4) into last_ms / sum_ms (nc_ctx.hip)
extern "C" __attribute__((visibility("hidden"))) void nc_timing_resolve(nc_ctx *ctx, int which);

// Stage timer (timing mode only): records an event pair on the launch stream and does NOT wait; the elapsed time is
// resolved lazily by nc_last_kernel_ms, so a timed step is not perturbed by host synchronisations.
struct NcTimer {
    nc_ctx *ctx;
    int which;
    NcTimer(nc_ctx *c, int w) : ctx(c), which(w)
    {
        if (ctx->timing == 1) {
            nc_timing_resolve(ctx, which);                   // an earlier call's pair (complete by now or soon): fold it in first
            for (int e = 0; e < 2; e++)
                if (!ctx->tev[which][e]) (void)hipEventCreate(&ctx->tev[which][e]);
            (void)hipEventRecord(ctx->tev[which][0], ctx->stream);
        }
    }
    void stop()
    {
        if (ctx->timing == 1) {
            (void)hipEventRecord(ctx->tev[which][1], ctx->stream);
            ctx->tev_pending[which] = true;
        }
    }
};

// implemented in the kernel translation units
int nc_selftest_device(nc_ctx *ctx);
void nc_pipe_destroy(nc_ctx *ctx);          // nc_pipe.hip

// K7 (nc_indel.hip): one chunk of a batched window scan
struct IndelChunk {
    int32_t lo, hi, ncol, nd;
    int64_t ws;          // byte offset of depth[3][ncol] | rank[ncol+1] | diff[8][nd] | (impute) cnt[3][ncol] in the workspace
    int64_t coloff;      // offset of this chunk's col_type in the concatenated output
    int32_t tile0, blk0; // first tile of the chunk on the pack's grid, first k_hap_depth_b block of the chunk
};
