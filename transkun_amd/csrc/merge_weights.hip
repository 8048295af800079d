// merge_weights.hip -- the merged projection's weights and their gradient (transkun_amd.fused.merged_weights).
//
// The reference's two projections q = x Wq^T + bq, k = x Wk^T + bk (LayersTransformer.py:392-397, :406-410) enter the score only
// through <q_e, k_b> = <x_e A + v, x_b> + c_e with A = Wq^T Wk, v = bq Wk, c_e = <x_e, Wq^T bk> + <bq, bk>.  With Wk1 = [Wk | bk]
// (D x (size + 1)) the single GEMM [z | c | diag | 0 ...] = x Wm^T + bm has
//     Wm[i][j] = sum_r Wk1[r][i] Wq[r][j]   (i <= size),   Wm[size + 1] = wd,   zero rows behind;
//     bm[i]    = sum_r Wk1[r][i] bq[r]      (i <= size),   bm[size + 1] = bd.
// As torch operations with autograd this was ~10 small kernels forward and ~25 backward per training step (slices, cats, two
// products that hipBLASLt runs as single tiles, fill + copy + add per slice gradient): 0.13 ms of a 1.3 ms step.  Here: one kernel
// each way, a block per output row, fp32 fmaf chains in index order (the result does not depend on timing).
//   forward   W [2 D + 1][size], bias [2 D + 1]  ->  Wm [size + pad][size], bm [size + pad]
//   backward  dWm, dbm  ->  dW [2 D + 1][size], dbias [2 D + 1]:
//     dWq[r][j] = sum_{i <= size} Wk1[r][i] dWm[i][j]            dbq[r] = sum_{i <= size} Wk1[r][i] dbm[i]
//     dWk[r][i] = sum_j Wq[r][j] dWm[i][j] + bq[r] dbm[i]        dbk[r] = sum_j Wq[r][j] dWm[size][j] + bq[r] dbm[size]
//     dwd = dWm[size + 1],  dbd = dbm[size + 1]
#include "common.h"

namespace semicrf {

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh)
{
    // fixed-order tree over the 256 threads of a block
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
}

// out[j] = sum_{r < n} col[r] * Mat[r][j] for j = thread (< width), col in LDS; four partial chains (r mod 4), combined in a fixed order
__device__ __forceinline__ float col_times_rows(const float* __restrict__ col, const float* __restrict__ Mat, int n, int width, int j)
{
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < width) {
        int r = 0;
        for (; r + 4 <= n; r += 4) {
            const float m0 = Mat[(size_t)r * width + j], m1 = Mat[(size_t)(r + 1) * width + j];
            const float m2 = Mat[(size_t)(r + 2) * width + j], m3 = Mat[(size_t)(r + 3) * width + j];
            a0 = fmaf(col[r], m0, a0); a1 = fmaf(col[r + 1], m1, a1); a2 = fmaf(col[r + 2], m2, a2); a3 = fmaf(col[r + 3], m3, a3);
        }
        for (; r < n; ++r) a0 = fmaf(col[r], Mat[(size_t)r * width + j], a0);
    }
    return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void merge_weights_fwd_kernel(const float* __restrict__ W, const float* __restrict__ bias, int D, int size,
                                                                int rows, float* __restrict__ Wm, float* __restrict__ bm, float* __restrict__ WmT)
{
    extern __shared__ float col[];                        // [D]: column i of Wk1 = [Wk | bk]
    __shared__ float sh[256];
    const int i = blockIdx.x, j = threadIdx.x;
    const float* Wq = W;
    const float* Wk = W + (size_t)D * size;
    const float* bq = bias;
    const float* bk = bias + D;
    if (i <= size) {
        for (int r = j; r < D; r += 256) col[r] = i < size ? Wk[(size_t)r * size + i] : bk[r];
        __syncthreads();
        const float acc = col_times_rows(col, Wq, D, size, j);
        if (j < size) Wm[(size_t)i * size + j] = acc;
        if (WmT && i < size && j < size) WmT[(size_t)j * size + i] = acc;      // the main rows transposed: the forward GEMM's B operand
        float bacc = 0.0f;
        for (int r = j; r < D; r += 256) bacc = fmaf(col[r], bq[r], bacc);
        const float b = block_sum_256(bacc, sh);
        if (j == 0) bm[i] = b;
    } else if (i == size + 1) {
        if (j < size) Wm[(size_t)i * size + j] = W[(size_t)(2 * D) * size + j];
        if (j == 0) bm[i] = bias[2 * D];
    } else if (i < rows) {
        if (j < size) Wm[(size_t)i * size + j] = 0.0f;
        if (j == 0) bm[i] = 0.0f;
    }
}

// out[j][i] = in[i][j] for i < nrows, j < ncols (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void merge_transpose_kernel(const float* __restrict__ in, int nrows, int ncols, float* __restrict__ out)
{
    __shared__ float tile[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int k = ty; k < 32; k += 8)
        if (i0 + k < nrows && j0 + tx < ncols) tile[k][tx] = in[(size_t)(i0 + k) * ncols + j0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (j0 + k < ncols && i0 + tx < nrows) out[(size_t)(j0 + k) * nrows + i0 + tx] = tile[tx][k];
}

__global__ __launch_bounds__(256) void merge_weights_bwd_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                const float* __restrict__ dWm, const float* __restrict__ dbm,
                                                                const float* __restrict__ dWmT, int D, int size,
                                                                float* __restrict__ dW, float* __restrict__ dbias)
{
    extern __shared__ float col[];                        // [size + 1]: row r of Wk1 (dWq) or row r of Wq (dWk)
    __shared__ float sh[256];
    const int row = blockIdx.x, t = threadIdx.x;          // row of dW / entry of dbias
    const float* Wq = W;
    const float* Wk = W + (size_t)D * size;
    const float* bq = bias;
    const float* bk = bias + D;
    if (row < D) {                                        // dWq[r] = Wk1[r] dWm[:size + 1],  dbq[r] = <Wk1[r], dbm[:size + 1]>
        const int r = row;
        for (int i = t; i <= size; i += 256) col[i] = i < size ? Wk[(size_t)r * size + i] : bk[r];
        __syncthreads();
        const float acc = col_times_rows(col, dWm, size + 1, size, t);
        if (t < size) dW[(size_t)r * size + t] = acc;
        float bacc = 0.0f;
        for (int i = t; i <= size; i += 256) bacc = fmaf(col[i], dbm[i], bacc);
        const float b = block_sum_256(bacc, sh);
        if (t == 0) dbias[r] = b;
    } else if (row < 2 * D) {                             // dWk[r][i] = <Wq[r], dWm[i]> + bq[r] dbm[i]; dbk[r]: the same with i = size
        // through the TRANSPOSE of dWm's first size + 1 rows (dWmT [size][size + 1], written by merge_transpose_kernel): thread i
        // walks column i of it, consecutive threads consecutive addresses
        const int r = row - D;
        for (int j = t; j < size; j += 256) col[j] = Wq[(size_t)r * size + j];
        __syncthreads();
        const float bqr = bq[r];
        for (int i = t; i <= size; i += 256) {
            const float acc = fmaf(bqr, dbm[i], col_times_rows(col, dWmT, size, size + 1, i));
            if (i < size) dW[(size_t)(D + r) * size + i] = acc;
            else dbias[D + r] = acc;
        }
    } else {                                              // dwd, dbd
        if (t < size) dW[(size_t)(2 * D) * size + t] = dWm[(size_t)(size + 1) * size + t];
        if (t == 0) dbias[2 * D] = dbm[size + 1];
    }
}

// The reference Linear's parameters in the layouts the projection kernels read (transkun_amd.scorer._ScorerLinearPacked), one launch:
//   BT  [size][2 D]      BT[k][n] = W[n][k] for the q and k rows: scorer_proj_nn's B for q (columns 0 .. D-1, ldb = 2 D) and k (D ..)
//   Wqd [rows_pad][size] the rows [Wq; diag row; zeros]: B of the input gradient through [q | diag | 0 ..]
//   w2  [2][size], b2 [2] the diagonal row and a zero row (the two extra columns of the forward), their biases
__global__ __launch_bounds__(256) void stage_linear_kernel(const float* __restrict__ W, const float* __restrict__ bias, int D, int size,
                                                           int rows_pad, int ntx, int nty, float* __restrict__ BT, float* __restrict__ Wqd,
                                                           float* __restrict__ w2, float* __restrict__ b2)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.x;
    const int ntrans = ntx * nty;
    if (b < ntrans) {                                     // 32 x 32 tile of the transpose
        const int j0 = (b % ntx) * 32, i0 = (b / ntx) * 32;   // i: row of W (n), j: column of W (k)
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int k = ty; k < 32; k += 8)
            if (i0 + k < 2 * D && j0 + tx < size) tile[k][tx] = W[(size_t)(i0 + k) * size + j0 + tx];
        __syncthreads();
        for (int k = ty; k < 32; k += 8)
            if (j0 + k < size && i0 + tx < 2 * D) BT[(size_t)(j0 + k) * (2 * D) + i0 + tx] = tile[tx][k];
    } else if (b < ntrans + rows_pad) {                   // a row of Wqd
        const int r = b - ntrans;
        const float* src = r < D ? W + (size_t)r * size : (r == D ? W + (size_t)(2 * D) * size : nullptr);
        for (int k = threadIdx.x; k < size; k += 256) Wqd[(size_t)r * size + k] = src ? src[k] : 0.0f;
    } else {                                              // w2, b2
        for (int k = threadIdx.x; k < size; k += 256) { w2[k] = W[(size_t)(2 * D) * size + k]; w2[size + k] = 0.0f; }
        if (threadIdx.x == 0) { b2[0] = bias[2 * D]; b2[1] = 0.0f; }
    }
}

}  // namespace

void launch_stage_linear(const float* W, const float* bias, int D, int size, int rows_pad, float* BT, float* Wqd, float* w2, float* b2,
                         hipStream_t stream)
{
    const int ntx = (size + 31) / 32, nty = (2 * D + 31) / 32;
    hipLaunchKernelGGL(stage_linear_kernel, dim3(ntx * nty + rows_pad + 1), dim3(256), 0, stream, W, bias, D, size, rows_pad, ntx, nty, BT, Wqd,
                       w2, b2);
}

void launch_merge_weights_fwd(const float* W, const float* bias, int D, int size, int rows, float* Wm, float* bm, float* WmT, hipStream_t stream)
{
    hipLaunchKernelGGL(merge_weights_fwd_kernel, dim3(rows), dim3(256), (size_t)D * sizeof(float), stream, W, bias, D, size, rows, Wm, bm, WmT);
}
size_t merge_weights_bwd_workspace_bytes(int size) { return (size_t)size * (size + 1) * sizeof(float); }

void launch_merge_weights_bwd(const float* W, const float* bias, const float* dWm, const float* dbm, int D, int size, float* dW, float* dbias,
                              float* ws, hipStream_t stream)
{
    hipLaunchKernelGGL(merge_transpose_kernel, dim3((size + 31) / 32, (size + 1 + 31) / 32), dim3(256), 0, stream, dWm, size + 1, size, ws);
    hipLaunchKernelGGL(merge_weights_bwd_kernel, dim3(2 * D + 1), dim3(256), (size_t)(size + 1) * sizeof(float), stream, W, bias, dWm, dbm, ws, D,
                       size, dW, dbias);
}

}  // namespace semicrf
