// Context, device-memory helpers and the host-side read packer of libnanocaller_hip.so.
#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "nc_common.h"
#include "nc_host.h"

extern "C" {

int nc_abi_version(void) { return NC_ABI_VERSION; }

int nc_device_count(int *n)
{
    if (!n) return NC_ERR_ARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return NC_ERR_HIP; }
    *n = c;
    return NC_OK;
}

}   // extern "C" (the copy helpers below have C++ linkage)

// ------------------------------------------------------------------ small transfers without the SDMA queue
namespace {
__global__ void k_copy16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}
__global__ void k_copy4(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}
__global__ void k_copy1(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    __threadfence_system();
}

void launch_copy(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    const uintptr_t a = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    // few, small workgroups: they must fit beside the persistent CNN workgroups (<= 16 VGPRs, no LDS) and a PCIe link is
    // saturated by a few hundred 16-byte stores in flight
    if ((a & 15) == 0) {
        const size_t n = bytes / 16;
        hipLaunchKernelGGL(k_copy16, dim3((unsigned)std::min<size_t>((n + 255) / 256, 128)), dim3(256), 0, st, (uint4 *)dst, (const uint4 *)src, n);
    } else if ((a & 3) == 0) {
        const size_t n = bytes / 4;
        hipLaunchKernelGGL(k_copy4, dim3((unsigned)std::min<size_t>((n + 255) / 256, 128)), dim3(256), 0, st, (uint32_t *)dst, (const uint32_t *)src, n);
    } else {
        hipLaunchKernelGGL(k_copy1, dim3((unsigned)std::min<size_t>((bytes + 255) / 256, 128)), dim3(256), 0, st, (uint8_t *)dst, (const uint8_t *)src, bytes);
    }
}
}   // namespace

int nc_d2h(nc_ctx *ctx, void *host, const void *dev, size_t bytes, hipStream_t st)
{
    if (!bytes) return NC_OK;
    hipPointerAttribute_t at;
    void *alias = nullptr;
    if (hipPointerGetAttributes(&at, host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) alias = at.devicePointer;
    else (void)hipGetLastError();                                  // an ordinary host pointer is not an error
    // Bulk results stay on the copy engine (no CU time, full PCIe rate): the caller enqueues a contig's upload BEFORE the
    // result copies of the step running under it, so they queue behind it and still land before the step ends.  Only the
    // small transfers the HOST WAITS FOR mid-step (scan totals) must not queue behind 7 ms of upload: those go by kernel.
    // NC_D2H_KERNEL_BYTES (experiment / tuning): largest transfer that goes by copy kernel
    static const size_t kernel_max = [] { const char *e = getenv("NC_D2H_KERNEL_BYTES"); return e ? (size_t)atoll(e) : (size_t)NC_D2H_KERNEL_MAX; }();
    if (!alias || bytes > kernel_max) {
        NC_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
        return NC_OK;
    }
    launch_copy(alias, dev, bytes, st);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

int nc_h2d_small(nc_ctx *ctx, void *dev, const void *host, size_t bytes, hipStream_t st)
{
    if (!bytes) return NC_OK;
    if (bytes > NC_STAGE_SLOT || !ctx->stage_h) {
        NC_HIP(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
        return NC_OK;
    }
    uint8_t *slot = ctx->stage_h + (size_t)(ctx->stage_turn++ % NC_STAGE_SLOTS) * NC_STAGE_SLOT;
    memcpy(slot, host, bytes);
    const size_t padded = (bytes + 15) & ~(size_t)15;              // slots and device buffers are 16-byte granular
    launch_copy(dev, slot, ((uintptr_t)dev & 15) ? bytes : padded, st);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

extern "C" {

int nc_d2h_async(nc_ctx *ctx, void *stream, void *host, const void *dev, size_t bytes)
{
    if (!ctx || (bytes && (!host || !dev))) return NC_ERR_ARG;
    return nc_d2h(ctx, host, dev, bytes, stream ? (hipStream_t)stream : ctx->stream);
}

int nc_ctx_create(int device_id, nc_ctx **out)
{
    if (!out) return NC_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return NC_ERR_HIP;   // fail loudly: no CPU fallback exists
    if (device_id < 0 || device_id >= n) return NC_ERR_ARG;
    nc_ctx *ctx = new (std::nothrow) nc_ctx();
    if (!ctx) return NC_ERR_NOMEM;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return NC_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipHostMalloc((void **)&ctx->mbox, 256, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&ctx->stage_h, (size_t)NC_STAGE_SLOT * NC_STAGE_SLOTS, hipHostMallocDefault) != hipSuccess) {
        nc_ctx_destroy(ctx);
        return NC_ERR_NOMEM;
    }
    int rc = nc_selftest_device(ctx);
    if (rc != NC_OK) {
        fprintf(stderr, "nanocaller_hip: device self-test failed: %s\n", ctx->err);
        nc_ctx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return NC_OK;
}

static void freebuf(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int nc_ctx_destroy(nc_ctx *ctx)
{
    if (!ctx) return NC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->stage_nbr, &ctx->stage_cpos, &ctx->stage_cn, &ctx->stage_calt, &ctx->tile_cnt,
                      &ctx->tile_pre, &ctx->nbr_pos, &ctx->cand_pos, &ctx->cand_n, &ctx->cand_alt,
                      &ctx->chunk_start, &ctx->chunk_end, &ctx->chunk_lo, &ctx->chunk_cnt, &ctx->chunk_off,
                      &ctx->site_pos, &ctx->site_chunk, &ctx->site_n, &ctx->site_alt, &ctx->totals,
                      &ctx->cnn_a, &ctx->cnn_b, &ctx->cnn_c, &ctx->chunk_depth, &ctx->nbr_idx, &ctx->indel_ws, &ctx->indel_ent_read,
                      &ctx->msa_reads, &ctx->msa_read_off, &ctx->msa_read_set, &ctx->msa_refs, &ctx->msa_ref_off, &ctx->msa_rows_hf,
                      &ctx->msa_hcol, &ctx->msa_tb, &ctx->msa_trace, &ctx->msa_cols, &ctx->msa_out, &ctx->msa_dup};
    for (DevBuf *b : bufs) freebuf(*b);
    nc_pipe_destroy(ctx);
    for (auto &w : ctx->w) {
        if (w.dev) (void)hipFree(w.dev);
        if (w.packed) (void)hipFree(w.packed);
        if (w.packed_h) (void)hipFree(w.packed_h);
    }
    for (auto &e : ctx->kev) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->drain_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->scan_ev) (void)hipEventDestroy(ctx->scan_ev);
    if (ctx->scale_ev) (void)hipEventDestroy(ctx->scale_ev);
    for (auto &p : ctx->tev) for (auto &e : p) if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->mbox) (void)hipHostFree(ctx->mbox);
    if (ctx->stage_h) (void)hipHostFree(ctx->stage_h);
    delete ctx;
    return NC_OK;
}

int nc_ctx_set_stream(nc_ctx *ctx, void *hip_stream)
{
    if (!ctx) return NC_ERR_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return NC_OK;
}

int nc_ctx_sync(nc_ctx *ctx)
{
    if (!ctx) return NC_ERR_ARG;
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NC_OK;
}

const char *nc_last_error(const nc_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int nc_malloc(nc_ctx *ctx, size_t bytes, void **dev)
{
    if (!ctx || !dev) return NC_ERR_ARG;
    NC_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dev, bytes ? bytes : 16);
    if (e != hipSuccess) return nc_fail(ctx, NC_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return NC_OK;
}

int nc_free(nc_ctx *ctx, void *dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (dev) NC_HIP(ctx, hipFree(dev));
    return NC_OK;
}

int nc_memcpy_h2d(nc_ctx *ctx, void *dev, const void *host, size_t bytes)
{
    if (!ctx || (bytes && (!dev || !host))) return NC_ERR_ARG;
    if (bytes) NC_HIP(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NC_OK;
}

int nc_memcpy_d2h(nc_ctx *ctx, void *host, const void *dev, size_t bytes)
{
    if (!ctx || (bytes && (!dev || !host))) return NC_ERR_ARG;
    if (bytes) NC_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NC_OK;
}

int nc_set_cnn_precision(nc_ctx *ctx, int exact_fp32)
{
    if (!ctx) return NC_ERR_ARG;
    ctx->cnn_exact_fp32 = exact_fp32 != 0;
    return NC_OK;
}

int nc_set_tensor_format(nc_ctx *ctx, int fmt)
{
    if (!ctx || (fmt != 0 && fmt != 1)) return NC_ERR_ARG;
    ctx->x_i16 = fmt == 1;
    return NC_OK;
}

int nc_enable_timing(nc_ctx *ctx, int on)
{
    if (!ctx) return NC_ERR_ARG;
    if (on && !ctx->timing)
        for (int w = 0; w < 6; w++) { ctx->sum_ms[w] = 0.0; ctx->sum_n[w] = 0; }
    ctx->timing = on == 2 ? 2 : on != 0;
    return NC_OK;
}

void nc_timing_resolve(nc_ctx *ctx, int which)
{
    if (which < 4) {
        if (!ctx->tev_pending[which]) return;
        if (hipEventSynchronize(ctx->tev[which][1]) == hipSuccess &&
            hipEventElapsedTime(&ctx->last_ms[which], ctx->tev[which][0], ctx->tev[which][1]) == hipSuccess) {
            ctx->sum_ms[which] += ctx->last_ms[which];
            ctx->sum_n[which]++;
        }
        ctx->tev_pending[which] = false;
        return;
    }
    if (!ctx->kev_pending) return;                            // per-launch durations of the trunk kernel, on the launch stream
    float tot = 0.0f;
    for (int e = 0; e + 1 < ctx->n_kev; e += 2) {
        float one = 0.0f;
        if (hipEventSynchronize(ctx->kev[e + 1]) == hipSuccess && hipEventElapsedTime(&one, ctx->kev[e], ctx->kev[e + 1]) == hipSuccess) tot += one;
    }
    ctx->last_ms[4] = tot;
    ctx->last_ms[5] = (float)(ctx->n_kev / 2);
    ctx->sum_ms[4] += tot;
    ctx->sum_ms[5] += ctx->n_kev / 2;
    ctx->sum_n[4]++;
    ctx->sum_n[5]++;
    ctx->kev_pending = false;
}

int nc_last_kernel_ms(nc_ctx *ctx, int which, float *ms)
{
    if (!ctx || !ms || which < 0 || which > 5) return NC_ERR_ARG;
    nc_timing_resolve(ctx, which < 4 ? which : 4);
    *ms = ctx->last_ms[which];
    return NC_OK;
}

int nc_timing_sums(nc_ctx *ctx, double *sum_ms, int64_t *count)
{
    if (!ctx || !sum_ms) return NC_ERR_ARG;
    for (int w = 0; w < 5; w++) nc_timing_resolve(ctx, w);
    for (int w = 0; w < 6; w++) {
        sum_ms[w] = ctx->sum_ms[w];
        if (count) count[w] = ctx->sum_n[w];
    }
    return NC_OK;
}

// ---------------------------------------------------------------------------------- host packer
static inline int64_t floor16(int64_t x) { return x & ~int64_t(15); }   // two's complement: works for x <= 0
static inline int64_t ceil16(int64_t x) { return (x + 15) & ~int64_t(15); }

static bool tile_size_ok(int32_t t) { return t == 1024 || t == 2048 || t == 4096; }

int nc_pack_plan(int32_t n_reads, const int32_t *start, const int32_t *end, const uint8_t *keep,
                 int32_t tile_size, int32_t pos_lo, int32_t pos_hi,
                 int64_t *codes_len, int32_t *tile_pos0, int32_t *n_tiles, int64_t *n_entries)
{
    if (n_reads < 0 || (n_reads && (!start || !end)) || !tile_size_ok(tile_size) || pos_hi < pos_lo ||
        !codes_len || !tile_pos0 || !n_tiles || !n_entries)
        return NC_ERR_ARG;
    int64_t t0 = (int64_t)pos_lo - (((int64_t)pos_lo % tile_size) + tile_size) % tile_size;   // floor to tile multiple
    int64_t nt = ((int64_t)pos_hi - t0) / tile_size + 1;
    if (nt > INT32_MAX) return NC_ERR_ARG;
    int64_t bytes = 0, ents = 0;
    int32_t prev = INT32_MIN;
    for (int32_t r = 0; r < n_reads; r++) {
        if (keep && !keep[r]) continue;
        if (end[r] <= start[r]) return NC_ERR_ARG;
        if (start[r] < prev) return NC_ERR_ARG;       // coordinate order required (as in a sorted BAM)
        prev = start[r];
        bytes += ceil16(end[r]) - floor16(start[r]);
        int64_t a = std::max<int64_t>(start[r], t0), b = std::min<int64_t>((int64_t)end[r] - 1, t0 + nt * tile_size - 1);
        if (a <= b) ents += (b - t0) / tile_size - (a - t0) / tile_size + 1;
    }
    *codes_len = bytes + 16;   // never empty
    *tile_pos0 = (int32_t)t0;
    *n_tiles = (int32_t)nt;
    *n_entries = ents;
    return NC_OK;
}

int nc_pack_fill(int32_t n_reads, const int32_t *start, const int32_t *end, const int64_t *off,
                 const uint8_t *codes_in, const uint8_t *strand, const uint8_t *keep,
                 int32_t tile_size, int32_t tile_pos0, int32_t n_tiles,
                 uint8_t *codes_out, int64_t codes_len, int32_t *tile_off, nc_tile_entry *tile_ent,
                 int64_t n_entries)
{
    // codes_in == NULL && codes_out == NULL: build the tile index only (the caller fills the slots itself)
    const bool index_only = !codes_in && !codes_out;
    if (n_reads < 0 || !tile_size_ok(tile_size) || n_tiles <= 0 || !tile_off || (n_entries && !tile_ent) ||
        (n_reads && (!start || !end)) || (!index_only && (!codes_out || (n_reads && (!off || !codes_in)))))
        return NC_ERR_ARG;
    const int64_t t0 = tile_pos0;
    std::vector<int64_t> cnt((size_t)n_tiles + 1, 0);
    // pass 1: counts per tile
    for (int32_t r = 0; r < n_reads; r++) {
        if (keep && !keep[r]) continue;
        int64_t a = std::max<int64_t>(start[r], t0), b = std::min<int64_t>((int64_t)end[r] - 1, t0 + (int64_t)n_tiles * tile_size - 1);
        if (a > b) continue;
        for (int64_t t = (a - t0) / tile_size; t <= (b - t0) / tile_size; t++) cnt[(size_t)t + 1]++;
    }
    for (int32_t t = 0; t < n_tiles; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
    if (cnt[(size_t)n_tiles] != n_entries || n_entries > INT32_MAX) return NC_ERR_CAPACITY;
    for (int32_t t = 0; t <= n_tiles; t++) tile_off[t] = (int32_t)cnt[(size_t)t];
    std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
    // slot bases (sequential, cheap), then the tile entries
    std::vector<int64_t> base((size_t)(n_reads > 0 ? n_reads : 1), 0);
    int64_t w = 0;   // write cursor (multiple of 16)
    for (int32_t r = 0; r < n_reads; r++) {
        if (keep && !keep[r]) { base[(size_t)r] = -1; continue; }
        const int64_t lo = floor16(start[r]), hi = ceil16(end[r]);
        if (w + (hi - lo) > codes_len) return NC_ERR_CAPACITY;
        base[(size_t)r] = w - lo;                               // codes_out[base + p], multiple of 16
        w += hi - lo;
        nc_tile_entry e;
        e.start = start[r];
        e.end = end[r];
        e.base_flag = base[(size_t)r] | (strand ? (strand[r] & 15) : 0); // bit0 reverse strand, bits 1-2 HP tag, bit 3: the read name is shared (nc_snp_set_mates)
        int64_t a = std::max<int64_t>(start[r], t0), b = std::min<int64_t>((int64_t)end[r] - 1, t0 + (int64_t)n_tiles * tile_size - 1);
        if (a > b) continue;
        for (int64_t t = (a - t0) / tile_size; t <= (b - t0) / tile_size; t++) tile_ent[cur[(size_t)t]++] = e;
    }
    if (!index_only) {
        // the slots of consecutive reads are consecutive in codes_out: host threads fill disjoint, contiguous ranges
        const int hw = nc_host_cpus();
        int T = hw > 32 ? 32 : hw;
        if (n_reads < 4096) T = 1;
        auto fill = [&](int32_t r0, int32_t r1, int64_t out0, int64_t out1) {
            memset(codes_out + out0, NC_CODE_ABSENT, (size_t)(out1 - out0));
            for (int32_t r = r0; r < r1; r++)
                if (!(keep && !keep[r]))
                    memcpy(codes_out + base[(size_t)r] + start[r], codes_in + off[r], (size_t)(end[r] - start[r]));
        };
        // first output byte of read r's slot = base + floor16(start); kept reads only
        auto slot0 = [&](int32_t r) { return base[(size_t)r] + floor16(start[r]); };
        std::vector<int32_t> cut((size_t)T + 1, n_reads);
        cut[0] = 0;
        for (int t = 1; t < T; t++) {
            int32_t r = (int32_t)((int64_t)n_reads * t / T);
            while (r < n_reads && keep && !keep[r]) r++;        // a range starts at a kept read
            cut[(size_t)t] = r;
        }
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) {
            const int32_t r0 = cut[(size_t)t], r1 = cut[(size_t)t + 1];
            const int64_t out0 = t == 0 ? 0 : (r0 < n_reads ? slot0(r0) : w);
            const int64_t out1 = t + 1 == T ? codes_len : (r1 < n_reads ? slot0(r1) : w);
            if (out1 <= out0 && r1 <= r0) continue;
            if (T == 1) fill(r0, r1, out0, out1);
            else th.emplace_back(fill, r0, r1, out0, out1);
        }
        for (auto &x : th) x.join();
    }
    return NC_OK;
}

}   // extern "C"
