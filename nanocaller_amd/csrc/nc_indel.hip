// K8: MSA rows -> (5,128,2) indel tensor (gfx950).
//
// Restates the histogram half of msa() (reference generate_indel_pileups.py:57-71): per alignment column the
// symbol histogram over the read rows, normalised by the row count in f32, minus the aligned reference's
// one-hot; the consensus symbol is the arg-max with the gap handicapped by 0.01.  One workgroup per read set,
// one lane per alignment column, rows streamed with coalesced byte loads (row-major rows: lane = column).
#include "nc_common.h"

namespace {

__global__ __launch_bounds__(256) void k_indel_tensor(const uint8_t *__restrict__ rows, const int64_t *__restrict__ row_off,
                                                      const int32_t *__restrict__ n_rows, const int32_t *__restrict__ n_cols,
                                                      const uint8_t *__restrict__ ref_rows, const int64_t *__restrict__ ref_off,
                                                      int max_cols, float *__restrict__ x, uint8_t *__restrict__ cns)
{
    const int s = blockIdx.x;
    const int nr = n_rows[s], nc = n_cols[s];
    const uint8_t *R = rows + row_off[s];
    const uint8_t *ref = ref_rows + ref_off[s];
    float *X = x + (int64_t)s * 5 * 128 * 2;
    for (int c = threadIdx.x; c < max_cols; c += 256) {
        if (c < nc) {
            int h[5] = {0, 0, 0, 0, 0};
            for (int r = 0; r < nr; r++) {
                const int sym = R[(int64_t)r * nc + c];
#pragma unroll
                for (int k = 0; k < 5; k++) h[k] += sym == k;
            }
            const float tot = (float)(h[0] + h[1] + h[2] + h[3] + h[4]);
            float alt[5], best = -1e30f;
            int arg = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                alt[k] = (float)h[k] / tot;                           // f32 divide (:58-59)
                const float tv = k == 4 ? alt[k] - 0.01f : alt[k];      // :62
                if (tv > best) { best = tv; arg = k; }                 // np.argmax: first maximum (:64)
            }
            cns[(int64_t)s * max_cols + c] = (uint8_t)arg;
            if (c < 128) {
                const int rc = ref[c];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const float rf = rc == k ? 1.0f : 0.0f;
                    X[(k * 128 + c) * 2 + 0] = alt[k] - rf;            // :67
                    X[(k * 128 + c) * 2 + 1] = rf;
                }
            }
        } else {
            cns[(int64_t)s * max_cols + c] = NC_CODE_ABSENT;
            if (c < 128) {
#pragma unroll
                for (int k = 0; k < 5; k++) {                           // zero padding (:70-71)
                    X[(k * 128 + c) * 2 + 0] = 0.0f;
                    X[(k * 128 + c) * 2 + 1] = 0.0f;
                }
            }
        }
    }
}

}   // namespace

extern "C" int nc_indel_tensor(nc_ctx *ctx, int32_t n_sets, const uint8_t *rows_dev, const int64_t *row_off_dev,
                               const int32_t *n_rows_dev, const int32_t *n_cols_dev, const uint8_t *ref_rows_dev,
                               const int64_t *ref_off_dev, int32_t max_cols, float *x_dev, uint8_t *cns_dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (n_sets < 0 || max_cols < 128) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_tensor: n_sets >= 0 and max_cols >= 128 required");
    if (n_sets == 0) return NC_OK;
    if (!rows_dev || !row_off_dev || !n_rows_dev || !n_cols_dev || !ref_rows_dev || !ref_off_dev || !x_dev || !cns_dev)
        return nc_fail(ctx, NC_ERR_ARG, "nc_indel_tensor: null argument");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    NcTimer tm(ctx, 3);
    hipLaunchKernelGGL(k_indel_tensor, dim3(n_sets), dim3(256), 0, ctx->stream, rows_dev, row_off_dev, n_rows_dev, n_cols_dev,
                       ref_rows_dev, ref_off_dev, max_cols, x_dev, cns_dev);
    NC_HIP(ctx, hipGetLastError());
    tm.stop();
    return NC_OK;
}
