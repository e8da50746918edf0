// HBM-bound helper kernels of the encoder: conv1 (3-tap, VALU), BatchNorm statistics
// finalisation, BN folding into the next layer's weights, dropout materialisation,
// BN+ReLU backward, conv1 backward, weight-gradient slab reduction.
// Activation layout inside the encoder: conv activations are stored position-major,
// [window][w = 0..11][channel = 0..63]  (feature index k' = w*64 + c), so that a row
// of the (window,position) x channel matrix is one contiguous 64-channel line.  The
// reference's Flatten order k = c*12 + w (code/models.py:263) is restored where weights
// of `emg_net.linear.0` are folded / their gradient is scattered.
#pragma once
#include "common.cuh"

// ------------------------------------------------------------------------------------
// gather: X[b][t][v][:] from the resident, mode-sliced table (code/utils.py:51-64,
// code/load.py:256-273).  train: src row = emg_rand[t][perm[b]];  eval: 25 consecutive
// rows of tensor[emg_rand[t][perm[b]]].  One thread per output float4 (3 per window).
// ------------------------------------------------------------------------------------
// g_gather_oob counts source rows that fell outside the table (a wrong emg_rand / V / table combination): such a row
// is read from row 0 so the launch stays memory-safe, and the count is there for cp_gather_oob_count to report.
__device__ unsigned int g_gather_oob;
__global__ void gather_groups_kernel(const float* __restrict__ table, const int64_t* __restrict__ emg_rand,
                                     const int64_t* __restrict__ perm, float* __restrict__ out, int64_t B, int T,
                                     int V, int64_t D, int64_t table_rows) {
    const int64_t total = B * T * V * 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % 3);
        const int64_t win = i / 3;
        const int v = (int)(win % V);
        const int64_t bt = win / V;
        const int t = (int)(bt % T);
        const int64_t b = bt / T;
        int64_t src = emg_rand[(int64_t)t * D + perm[b]] * V + v;      // row of the (rows,12) table
        if (src < 0 || src >= table_rows) {
            if (q == 0) atomicAdd(&g_gather_oob, 1u);
            src = 0;
        }
        *(float4*)(out + win * 12 + q * 4) = *(const float4*)(table + src * 12 + q * 4);
    }
}

// ------------------------------------------------------------------------------------
// BatchNorm statistics -> affine (train: batch stats, biased variance, eps 1e-5; running
// stats updated with momentum and the unbiased variance, as nn.BatchNorm does).
// partials: [nrows][2][C] (sum, sum of squares).  use_running: eval with stock BN.
// stats out: [4][C] = mean, invstd, scale = gamma*invstd, shift = beta - mean*scale
// ------------------------------------------------------------------------------------
// (finalize kernels: FIN_COLS columns x FIN_LANES row lanes per block of FIN_THREADS, FIN_GRID(C) blocks: up to
//  FIN_DIRECT_ROWS partial rows -- 32 per lane -- are folded here directly, without a reduce_rows_kernel launch first.
//  With 64 columns x 16 row lanes the 512-column layers ran on 8 blocks and 656 rows took 9-16 us.)
#define FIN_THREADS 1024
#define FIN_COLS 16
#define FIN_LANES (FIN_THREADS / FIN_COLS)
#define FIN_GRID(C) (((C) + FIN_COLS - 1) / FIN_COLS)
#define FIN_DIRECT_ROWS 2048
// sum over the block's row lanes of one value per thread (thread = column cl + FIN_COLS * row lane): the four row
// lanes of a wave by shuffles, the 16 waves through red; valid in the threads tid < FIN_COLS
__device__ __forceinline__ double fin_block_sum(double s, double (*red)[FIN_COLS], int tid) {
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if ((tid & 63) < FIN_COLS) red[tid >> 6][tid & 63] = s;
    __syncthreads();
    double t = 0;
    if (tid < FIN_COLS)
        for (int q = 0; q < FIN_THREADS / 64; ++q) t += red[q][tid];
    return t;
}
// s += p[r * stride] for r = g, g + L, ... < nrows, in that order; the loads go out eight rows at a time (one row per
// round trip made 656 rows cost 16 us instead of 6)
__device__ __forceinline__ void fin_fold_rows(const float* __restrict__ p, int nrows, int g, int L, int64_t stride, double& s) {
    int r = g;
    for (; r + 7 * L < nrows; r += 8 * L) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(r + u * L) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; r < nrows; r += L) s += (double)p[(int64_t)r * stride];
}
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_kernel(const float* __restrict__ partials, int nrows, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, int update_running,
                                                          int use_running, float momentum, float eps,
                                                          float* __restrict__ stats, int C, const int* __restrict__ unscale_exp = nullptr) {
    __shared__ double red[2][FIN_THREADS / 64][FIN_COLS];
    const int tid = threadIdx.x, cl = tid % FIN_COLS, g = tid / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + cl;
    double s1 = 0, s2 = 0;
    if (!use_running && c < C) {
        fin_fold_rows(partials + c, nrows, g, FIN_LANES, 2 * (int64_t)C, s1);
        fin_fold_rows(partials + C + c, nrows, g, FIN_LANES, 2 * (int64_t)C, s2);
    }
    s1 = fin_block_sum(s1, red[0], tid);
    s2 = fin_block_sum(s2, red[1], tid);
    if (unscale_exp != nullptr) {          // CP_FP8: the sums are of values stored with the scale 2^e (exact to undo)
        const double d = (double)f8_exp2i(-*unscale_exp);
        s1 *= d;
        s2 *= d * d;
    }
    if (g == 0 && c < C) {
        float mean, var;
        if (use_running) {
            mean = running_mean[c];
            var = running_var[c];
        } else {
            const double mu = s1 / count;
            double vb = s2 / count - mu * mu;
            if (vb < 0) vb = 0;
            mean = (float)mu;
            var = (float)vb;
            if (update_running) {
                const double unb = count > 1 ? vb * count / (count - 1) : vb;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
        const float invstd = 1.0f / sqrtf(var + eps);
        const float sc = gamma[c] * invstd;
        stats[0 * C + c] = mean;
        stats[1 * C + c] = invstd;
        stats[2 * C + c] = sc;
        stats[3 * C + c] = beta[c] - mean * sc;
    }
}

// ------------------------------------------------------------------------------------
// Evaluation with stock BatchNorm (running statistics, code/models.py:238-243 in eval mode): the affine of EVERY layer is known
// before the pass starts, so one launch writes all nine `stats` tables (round 4: nine bn_finalize launches -> one; the statistics
// the GEMM epilogues still sum are not read).  grid CP_N_BN blocks of 512 threads.
// ------------------------------------------------------------------------------------
struct BnRunningAll {
    const float* gamma[9];
    const float* beta[9];
    const float* mean[9];
    const float* var[9];
    float* stats[9];
    int C[9];
    float eps;
};
__global__ __launch_bounds__(512) void bn_running_stats_kernel(BnRunningAll a) {
    const int l = blockIdx.x, c = threadIdx.x, C = a.C[l];
    if (c >= C) return;
    const float mean = a.mean[l][c], invstd = 1.0f / sqrtf(a.var[l][c] + a.eps), sc = a.gamma[l][c] * invstd;
    a.stats[l][0 * C + c] = mean;
    a.stats[l][1 * C + c] = invstd;
    a.stats[l][2 * C + c] = sc;
    a.stats[l][3 * C + c] = a.beta[l][c] - mean * sc;
}

// ------------------------------------------------------------------------------------
// fold the previous BN's affine into a Linear:  y = (r*s + t) W^T + b = r (W diag s)^T + (b + W t)
//   mode 0: k' = k, channel = k               (512-wide inputs)
//   mode 1: input is the conv stack: source k = c*12 + w, stored k' = w*64 + c, channel = c
// out_w: [rows_out][K] T (rows >= F zero-filled, used to pad the 16-row projection to 32),
// out_b: [F] f32 (nullable when the layer has no bias and no fold),  s/t nullable -> plain copy.
// ------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void fold_linear_row(const float* __restrict__ W, const float* __restrict__ b,
                                                const float* __restrict__ s, const float* __restrict__ t,
                                                T* __restrict__ out_w, float* __restrict__ out_b, int F, int K,
                                                int mode, int j) {
    using D = DT<T>;
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int k = tid; k < K; k += 256) {
        int kp = k, ch = k;
        if (mode == 1) { ch = k / 12; kp = (k % 12) * 64 + ch; }
        float wv = 0.f;
        if (j < F) {
            wv = W[(int64_t)j * K + k];
            if (s != nullptr) { acc = fmaf(wv, t[ch], acc); wv *= s[ch]; }
        }
        D::store(out_w + (int64_t)j * K + kp, wv);
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && out_b != nullptr && j < F) out_b[j] = (b ? b[j] : 0.f) + red[0] + red[1] + red[2] + red[3];
}
template <typename T>
__global__ __launch_bounds__(256) void fold_linear_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                          const float* __restrict__ s, const float* __restrict__ t,
                                                          T* __restrict__ out_w, float* __restrict__ out_b, int F, int K,
                                                          int mode) {
    fold_linear_row<T>(W, b, s, t, out_w, out_b, F, K, mode, blockIdx.x);
}
// every fold of a forward pass whose statistics exist up front (evaluation with the running statistics: bn_running_stats_kernel wrote all
// nine tables) in ONE launch: blockIdx.y = job, blockIdx.x = output row (rows >= rows_out of a job: nothing to do)
struct FoldBnJob { const float* W; const float* b; const float* s; const float* t; void* out_w; float* out_b; int F, K, mode, rows_out; };
struct FoldBnBatch { FoldBnJob job[8]; };
template <typename T>
__global__ __launch_bounds__(256) void fold_linear_batch_kernel(FoldBnBatch fb) {
    const FoldBnJob& jb = fb.job[blockIdx.y];
    if ((int)blockIdx.x >= jb.rows_out) return;
    fold_linear_row<T>(jb.W, jb.b, jb.s, jb.t, (T*)jb.out_w, jb.out_b, jb.F, jb.K, jb.mode, blockIdx.x);
}

// plain copies (no BatchNorm fold: the layers behind a dropout, whose input already is dropout(BN(.))) of several layers in ONE
// launch at the start of the forward pass (blockIdx.y = job, blockIdx.x = output row; rows >= rows_out of a job: nothing to do) --
// they do not wait for any statistics, so they need not sit between the GEMMs as four ~5 us launches
struct FoldJob { const float* W; const float* b; void* out_w; float* out_b; int F, K, rows_out; };
struct FoldBatch { FoldJob job[4]; };
template <typename T>
__global__ __launch_bounds__(256) void fold_copy_batch_kernel(FoldBatch fb) {
    using D = DT<T>;
    const FoldJob& jb = fb.job[blockIdx.y];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (j >= jb.rows_out) return;
    for (int k = tid; k < jb.K; k += 256) D::store((T*)jb.out_w + (int64_t)j * jb.K + k, j < jb.F ? jb.W[(int64_t)j * jb.K + k] : 0.f);
    if (tid == 0 && jb.out_b != nullptr && j < jb.F) jb.out_b[j] = jb.b ? jb.b[j] : 0.f;
}

// transposed copy for the data-gradient GEMM:  out[k'][j] = W[j][k]   (ld_out >= F, zero padded)
template <typename T>
__global__ void transpose_w_kernel(const float* __restrict__ W, T* __restrict__ out, int F, int K, int ld_out, int mode) {
    using D = DT<T>;
    const int64_t total = (int64_t)K * ld_out;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % ld_out);
        const int kp = (int)(i / ld_out);
        int k = kp;
        if (mode == 1) k = (kp & 63) * 12 + (kp >> 6);
        D::store(out + i, j < F ? W[(int64_t)j * K + k] : 0.f);
    }
}

// the same for several weight matrices in ONE launch (blockIdx.y = job): the 7 fc layers + the projection at the start
// of the backward pass (8 launches of ~5 us each otherwise)
struct TransposeJob { const float* W; void* out; int F, K, ld_out, mode; };
struct TransposeBatch { TransposeJob job[8]; };
// 64 x 64 tiles through LDS: rows of W are read along k (coalesced f32), rows of the output are written along j
// (coalesced T).  grid (max tiles of a job, jobs), 256 threads.  (The first version read W with a stride of K floats
// between neighbouring threads and divided a 64-bit index per element: 16.6 us for the eight matrices, now 11.5.)
template <typename T>
__device__ __forceinline__ void transpose_w_job(const TransposeJob& j, int bx, int gx, float (*tile)[65]) {
    using D = DT<T>;
    const int tiles_j = j.ld_out / 64, tiles_k = j.K / 64;
    T* out = (T*)j.out;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int t = bx; t < tiles_j * tiles_k; t += gx) {
        const int j0 = (t % tiles_j) * 64, k0 = (t / tiles_j) * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = grp + 4 * r;
            tile[jj][lane] = (j0 + jj) < j.F ? j.W[(int64_t)(j0 + jj) * j.K + k0 + lane] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = grp + 4 * r, k = k0 + kk;
            const int kp = j.mode == 1 ? (k % 12) * 64 + k / 12 : k;       // fc1: internal order k' = w*64 + c of k = c*12 + w
            D::store(out + (int64_t)kp * j.ld_out + j0 + lane, tile[lane][kk]);
        }
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256) void transpose_w_batch_kernel(TransposeBatch b) {
    __shared__ float tile[64][65];
    transpose_w_job<T>(b.job[blockIdx.y], blockIdx.x, gridDim.x, tile);
}

// conv2 weights (64,64,3,3): kernel row 1 only.
//   fwd[o][tap*64 + i]  = W[o][i][1][tap]
//   dgr[i][tap*64 + o]  = W[o][i][1][2 - tap]     (flipped taps for the data gradient)
template <typename T>
__device__ __forceinline__ void prep_conv2_body(const float* __restrict__ W, T* __restrict__ fwd, T* __restrict__ dgr, int first, int stride) {
    using D = DT<T>;
    for (int idx = first; idx < 64 * 192; idx += stride) {
        const int a = idx / 192, rem = idx % 192, tap = rem / 64, b = rem % 64;
        D::store(fwd + idx, W[((a * 64 + b) * 3 + 1) * 3 + tap]);
        D::store(dgr + idx, W[((b * 64 + a) * 3 + 1) * 3 + (2 - tap)]);
    }
}
template <typename T>
__global__ void prep_conv2_kernel(const float* __restrict__ W, T* __restrict__ fwd, T* __restrict__ dgr) {
    prep_conv2_body<T>(W, fwd, dgr, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ------------------------------------------------------------------------------------
// u = dropout(r*s + t)  (code/models.py:282,287,292,297: Dropout after BN of fc4..fc7)
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_dropout_apply_kernel(const T* __restrict__ r, const float* __restrict__ stats,
                                                               T* __restrict__ u, int64_t rows, int C, uint32_t thresh,
                                                               uint32_t key, float inv_keep, const uint32_t* __restrict__ salt) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    if (salt) key ^= *salt;                        // graph replay: the per-step part of the key lives in device memory
    // A thread keeps one 16-byte column chunk and walks rows: scale / shift sit in registers as channel pairs and there
    // is no index arithmetic per chunk.  (The first version mapped a flat chunk index to (row, column) with a 64-bit
    // divide and re-read its 16 coefficients for every chunk: ~160 VALU instructions per 16 bytes, VALU-bound at 68 us
    // per 167,936 x 512 layer.)  Two rows per pass keep two loads in flight per thread.
    const int cpr = C / EPC, rpp = blockDim.x / cpr;
    const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
    const int f = cc * EPC;
    f32x2_t sc[EPC / 2], sh[EPC / 2];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e / 2][e & 1] = stats[2 * C + f + e]; sh[e / 2][e & 1] = stats[3 * C + f + e]; }
    auto apply = [&](const uint4& in, int64_t m) {
        float v[EPC];
        D::unpack(in, v);
#pragma unroll
        for (int e = 0; e < EPC; e += 2) {
            const uint32_t pr = dropout_pair(key, (uint32_t)m, (uint32_t)C, (uint32_t)(f + e));
            const f32x2_t keep = {dropout_scale(pr, 0, thresh, inv_keep), dropout_scale(pr, 1, thresh, inv_keep)};
            const f32x2_t y = __builtin_elementwise_fma((f32x2_t){v[e], v[e + 1]}, sc[e / 2], sh[e / 2]) * keep;
            v[e] = y.x;
            v[e + 1] = y.y;
        }
        *(uint4*)(u + m * C + f) = D::pack(v);
    };
    const int64_t step = (int64_t)gridDim.x * rpp;
    int64_t m = (int64_t)blockIdx.x * rpp + rr;
    for (; m + step < rows; m += 2 * step) {
        const uint4 a0 = *(const uint4*)(r + m * C + f);
        const uint4 a1 = *(const uint4*)(r + (m + step) * C + f);
        apply(a0, m);
        apply(a1, m + step);
    }
    if (m < rows) apply(*(const uint4*)(r + m * C + f), m);
}

// ------------------------------------------------------------------------------------
// BN backward, step 1: reduce the data-gradient GEMM's per-block sums S1 = sum g, S2 = sum g*r
// into the coefficients of   g_y = [r>0] * (ca*g + cb*r + cc)   and the gamma/beta gradients.
//   x_hat = (r - mean)*invstd;  dgamma = sum g*x_hat;  dbeta = sum g
//   g_r = s * (g - mean(g) - x_hat * mean(g*x_hat))
// ------------------------------------------------------------------------------------
// local (synchronised BatchNorm only, else nullptr): ONE row [2][nfold*C] of this rank's own sums.  `partials` then holds
// the sums over all ranks (they enter the data-gradient coefficients with the global count), while dgamma / dbeta are
// formed from the local row: the gradient all-reduce adds the ranks' parts, as torch.nn.SyncBatchNorm's backward does.
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nrows, double count,
                                                              const float* __restrict__ stats, float* __restrict__ coef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                                              int nfold, const float* __restrict__ local = nullptr) {
    // nfold > 1: the partial rows are nfold*C wide (feature = w*C + channel, conv stack seen
    // through the first Linear); the BatchNorm2d channel statistic sums over w.
    __shared__ double red[2][FIN_THREADS / 64][FIN_COLS];
    const int tid = threadIdx.x, cl = tid % FIN_COLS, g = tid / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + cl;
    const int W = C * nfold;
    double s1 = 0, s2 = 0;
    if (c < C) {
        if (nfold == 1) {
            fin_fold_rows(partials + c, nrows, g, FIN_LANES, 2 * (int64_t)W, s1);
            fin_fold_rows(partials + W + c, nrows, g, FIN_LANES, 2 * (int64_t)W, s2);
        } else {
            for (int r = g; r < nrows; r += FIN_LANES)
                for (int f = 0; f < nfold; ++f) {
                    s1 += (double)partials[((int64_t)r * 2 + 0) * W + f * C + c];
                    s2 += (double)partials[((int64_t)r * 2 + 1) * W + f * C + c];
                }
        }
    }
    s1 = fin_block_sum(s1, red[0], tid);
    s2 = fin_block_sum(s2, red[1], tid);
    if (g == 0 && c < C) {
        const double mean = stats[c], invstd = stats[C + c], sc = stats[2 * C + c];
        const double dot = (s2 - mean * s1) * invstd;          // sum g * x_hat
        const double c1 = s1 / count, c2 = dot / count;
        coef[0 * C + c] = (float)sc;
        coef[1 * C + c] = (float)(-sc * invstd * c2);
        coef[2 * C + c] = (float)(-sc * (c1 - mean * invstd * c2));
        if (local != nullptr) {
            s1 = s2 = 0;
            for (int f = 0; f < nfold; ++f) {
                s1 += (double)local[f * C + c];
                s2 += (double)local[W + f * C + c];
            }
            dgamma[c] = (float)((s2 - mean * s1) * invstd);
            dbeta[c] = (float)s1;
        } else {
            dgamma[c] = (float)dot;
            dbeta[c] = (float)s1;
        }
    }
}

// BN backward, step 2 + ReLU backward, in place; per-block column sums of g_y (bias gradient).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(T* __restrict__ g, const T* __restrict__ r,
                                                          const float* __restrict__ coef, float* __restrict__ partials,
                                                          int64_t rows, int C) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    extern __shared__ float dyn_red[];                  // [RPP][C]
    const int cpr = C / EPC, rpp = 256 / cpr;
    const int tid = threadIdx.x, cc = tid % cpr, rr = tid / cpr;
    float ca[EPC], cb[EPC], cz[EPC], sum[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        ca[e] = coef[cc * EPC + e];
        cb[e] = coef[C + cc * EPC + e];
        cz[e] = coef[2 * C + cc * EPC + e];
        sum[e] = 0.f;
    }
    auto apply = [&](const uint4& gq, const uint4& rq, int64_t m) {
        float gv[EPC], rv[EPC];
        D::unpack(gq, gv);
        D::unpack(rq, rv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float y = rv[e] > 0.f ? fmaf(ca[e], gv[e], fmaf(cb[e], rv[e], cz[e])) : 0.f;
            y = D::round(y);
            gv[e] = y;
            sum[e] += y;
        }
        *(uint4*)(g + m * C + cc * EPC) = D::pack(gv);
    };
    // two rows per pass: four 16-byte loads in flight per thread before the first store
    const int64_t step = (int64_t)gridDim.x * rpp;
    int64_t m = (int64_t)blockIdx.x * rpp + rr;
    for (; m + step < rows; m += 2 * step) {
        const uint4 g0 = *(const uint4*)(g + m * C + cc * EPC), r0 = *(const uint4*)(r + m * C + cc * EPC);
        const uint4 g1 = *(const uint4*)(g + (m + step) * C + cc * EPC), r1 = *(const uint4*)(r + (m + step) * C + cc * EPC);
        apply(g0, r0, m);
        apply(g1, r1, m + step);
    }
    if (m < rows) apply(*(const uint4*)(g + m * C + cc * EPC), *(const uint4*)(r + m * C + cc * EPC), m);
#pragma unroll
    for (int e = 0; e < EPC; ++e) dyn_red[rr * C + cc * EPC + e] = sum[e];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int q = 0; q < rpp; ++q) s += dyn_red[q * C + c];
        partials[(int64_t)blockIdx.x * C + c] = s;
    }
}

// column sums of an [rows][ld] T matrix (first C columns) -> partials[block][C]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ partials, int64_t rows,
                                                     int ld, int C) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    extern __shared__ float dyn_red[];
    const int cpr = C / EPC, rpp = 256 / cpr;
    const int tid = threadIdx.x, cc = tid % cpr, rr = tid / cpr;
    float sum[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) sum[e] = 0.f;
    // (C = 768: 96 or 192 chunks per row do not divide 256 -- the threads past the last whole row idle)
    for (int64_t m = (int64_t)blockIdx.x * rpp + rr; rr < rpp && m < rows; m += (int64_t)gridDim.x * rpp) {
        float v[EPC];
        D::unpack(*(const uint4*)(x + m * ld + cc * EPC), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) sum[e] += v[e];
    }
    if (rr < rpp) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) dyn_red[rr * C + cc * EPC + e] = sum[e];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int q = 0; q < rpp; ++q) s += dyn_red[q * C + c];
        partials[(int64_t)blockIdx.x * C + c] = s;
    }
}

// first stage of every partial-sum reduction: fold nrows partial rows of width W into
// REDUCE_SLICES rows (row r goes to slice r % REDUCE_SLICES), so that the single-block finalize
// kernels never walk more than REDUCE_SLICES rows.  grid (W/64, REDUCE_SLICES), 256 threads.
#define REDUCE_SLICES 32
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, int nrows, int W,
                                                          float* __restrict__ out) {
    __shared__ float red[4][64];
    const int tid = threadIdx.x, cl = tid & 63, g = tid >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    // same order of additions as the plain loop, loads eight rows deep (7,872 rows of 64 columns -- the conv2 bias
    // gradient under the dynamic tile schedule -- took 25 us one load at a time)
    constexpr int STEP = REDUCE_SLICES * 4;
    int r = blockIdx.y + REDUCE_SLICES * g;
    for (; r + 7 * STEP < nrows; r += 8 * STEP) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = in[(int64_t)(r + u * STEP) * W + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrows; r += STEP) s += in[(int64_t)r * W + c];
    red[g][cl] = s;
    __syncthreads();
    if (g == 0) out[(int64_t)blockIdx.y * W + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

// out[c] = sum_rows partials[row][c]  (f64 accumulation; one thread per column)
// grid FIN_GRID(C) blocks of FIN_THREADS
// unscale_exp (CP_FP8 statistics rows [sum | sum of squares] of values stored with the scale 2^e, else nullptr): columns below
// `half` are multiplied by 2^-e, the others by 2^-2e -- the row in true units (synchronised BatchNorm: every rank has its own e)
__global__ __launch_bounds__(FIN_THREADS) void colsum_finalize_kernel(const float* __restrict__ partials, int nrows, int C,
                                                                      float* __restrict__ out, const int* __restrict__ unscale_exp = nullptr,
                                                                      int half = 0) {
    __shared__ double red[FIN_THREADS / 64][FIN_COLS];
    const int tid = threadIdx.x, cl = tid % FIN_COLS, g = tid / FIN_COLS;
    const int c = blockIdx.x * FIN_COLS + cl;
    double s = 0;
    if (c < C) fin_fold_rows(partials + c, nrows, g, FIN_LANES, C, s);
    s = fin_block_sum(s, red, tid);
    if (unscale_exp != nullptr) {
        const double d = (double)f8_exp2i(-*unscale_exp);
        s *= c < half ? d : d * d;
    }
    if (g == 0 && c < C) out[c] = (float)s;
}

// dW1[c][0][1][tap] and db1[c] from the partials; the other kernel rows get zero data gradient.
__global__ void conv1_bwd_finalize_kernel(const float* __restrict__ partials, int nrows, float* __restrict__ dW,
                                          float* __restrict__ db) {
    const int tid = threadIdx.x;                       // 256 threads: (k, c)
    const int k = tid >> 6, c = tid & 63;
    // rows in order, loads eight deep (one row per round trip: 9.3 us for <= 512 rows)
    double s = 0;
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partials[((int64_t)(r + u) * 4 + k) * 64 + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; r < nrows; ++r) s += (double)partials[((int64_t)r * 4 + k) * 64 + c];
    if (k < 3) {
        dW[c * 9 + 3 + k] = (float)s;
        dW[c * 9 + 0 + k] = 0.f;
        dW[c * 9 + 6 + k] = 0.f;
    } else {
        db[c] = (float)s;
    }
}

// ------------------------------------------------------------------------------------
// sum weight-gradient slabs, undo the BN fold and scatter into the reference's layout.
//   mode 0 (fc):   grad[p*Q + q]      = s[q]*P + t[q]*dbsum[p]
//   mode 1 (fc1):  q = w*64 + c  ->   grad[p*768 + c*12 + w] = s[c]*P + t[c]*dbsum[p]
//   (conv2's slabs: conv2_wgrad_finish_kernel)
//   mode 3 (transposed): grad[q*p_valid + p] = P   (slab rows are the operand that was padded)
//   rows p >= p_valid are dropped (projection padded 16 -> 64)
// ------------------------------------------------------------------------------------
template <typename ST = float>         // ST: element type of the slabs (bf16_t: the 8-bit path's gemm_tn8_kernel)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs_, int S, int P, int Q, int p_valid,
                                    const float* __restrict__ s, const float* __restrict__ t,
                                    const float* __restrict__ dbsum, float* __restrict__ grad, int mode,
                                    float* __restrict__ raw, const int* __restrict__ exp_x = nullptr, const int* __restrict__ exp_y = nullptr) {
    // exp_x / exp_y (CP_FP8): scale exponents of the two 8-bit operands of the product; the slabs hold it in stored units
    const float unscale = exp_x != nullptr ? f8_exp2i(-*exp_x - *exp_y) : 1.f;
    // FOUR consecutive columns per thread and load (round 4, third part: one column per thread was 64 four-byte loads per output and
    // bound by their number, not by the bytes -- halving the bytes with bf16 slabs changed nothing: 11.6 -> 11.3 us); Q is a multiple of 4
    const int64_t total4 = (int64_t)p_valid * Q / 4;
    const int64_t sstride = (int64_t)P * Q;
    for (int64_t i4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i4 * 4;
        const int p = (int)(i / Q), q0 = (int)(i % Q);
        const ST* src = (const ST*)slabs_ + (int64_t)p * Q + q0;
        auto ld = [&](int kk, float (&v)[4]) {
            if constexpr (sizeof(ST) == 2) {
                const uint2 w = *(const uint2*)(src + kk * sstride);
                v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
                v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
            } else {
                const float4 w = *(const float4*)(src + kk * sstride);
                v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
            }
        };
        // four independent chains per column keep several slab loads in flight (a single chain is latency-bound); the order of the
        // additions per column is what it was with one column per thread
        float a[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[c][e] = 0.f;
        int k = 0;
        for (; k + 4 <= S; k += 4) {
            float v0[4], v1[4], v2[4], v3[4];
            ld(k + 0, v0); ld(k + 1, v1); ld(k + 2, v2); ld(k + 3, v3);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[0][e] += v0[e]; a[1][e] += v1[e]; a[2][e] += v2[e]; a[3][e] += v3[e]; }
        }
        for (; k < S; ++k) {
            float v0[4];
            ld(k, v0);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[0][e] += v0[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = q0 + e;
            float acc = ((a[0][e] + a[1][e]) + (a[2][e] + a[3][e])) * unscale;
            if (raw != nullptr) raw[i + e] = acc;          // un-fixed product g_y^T r, input of bn_bwd_sums_from_wgrad_kernel
            int ch = q, dst = p * Q + q;
            if (mode == 1) { ch = q & 63; dst = p * Q + ch * 12 + (q >> 6); }
            if (mode == 3) dst = q * p_valid + p;
            if (s != nullptr) acc = fmaf(s[ch], acc, t[ch] * dbsum[p]);
            grad[dst] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------
// conv2's weight gradient finished, and BatchNorm1's backward sums with it (round 4): no N-sized tensor is read.
// conv2_wgrad_kernel left slabs of the RAW product  P[o][tap][c] = sum_{n,q} g[n][q][o] r1[n][q + tap - 1][c]  (r1 = conv1's
// rounded ReLU output, zero outside the 12 positions).  With u1 = s r1 + t (BatchNorm1, zero-padded AFTER the affine map) and
// G[o][tap] = sum of g[n][q][o] over the positions q whose tap lands inside the window (tap 0: q >= 1, tap 2: q <= 10):
//     dW2[o][c][tap] = s[c] P + t[c] G[o][tap]
//     S1[c] = sum_{n,p} g_v1[n][p][c]           = sum_{o,tap} W2[o][c][tap] G[o][tap]
//     S2[c] = sum_{n,p} g_v1[n][p][c] r1[n][p][c] = sum_{o,tap} W2[o][c][tap] P[o][tap][c]
// (g_v1 = conv2's data gradient; W2 rounded as the data-gradient kernel's operand is).  gcols: nr partial rows [768] of the column
// sums of g seen as [N][q * 64 + o] -- the bias-gradient rows fc1's data-gradient launch wrote, or colsum_kernel's; or (gcols3, small
// batches) nr rows [3][64] of the sums over all positions / position 0 / position 11 from conv2_wgrad_kernel<T, false, true>, in which case
// conv2's bias gradient db2 is written here too.
// grid CONV2_FINISH_ROWS = 64 blocks (one output channel each) x 256 threads; out_rows[64][2][64] in bn_bwd_finalize_kernel's layout.
// ------------------------------------------------------------------------------------
#define CONV2_FINISH_ROWS 64
template <typename T>
__global__ __launch_bounds__(256) void conv2_wgrad_finish_kernel(const float* __restrict__ slabs, int S, const float* __restrict__ gcols, int nr,
                                                                 const float* __restrict__ W2, const float* __restrict__ stats1,
                                                                 float* __restrict__ dW2, float* __restrict__ out_rows,
                                                                 const float* __restrict__ gcols3 = nullptr, float* __restrict__ db2 = nullptr) {
    using D = DT<T>;
    __shared__ double part[21][12];
    __shared__ float Gs[3];
    __shared__ float red[2][3][64];
    const int tid = threadIdx.x, o = blockIdx.x;
    const int ncol = gcols3 ? 3 : 12;
    if (tid < 21 * ncol) {
        const int q = tid % ncol, ls = tid / ncol;
        double s = 0;
        if (gcols3) { for (int rw = ls; rw < nr; rw += 21) s += (double)gcols3[(int64_t)rw * 192 + q * 64 + o]; }
        else { for (int rw = ls; rw < nr; rw += 21) s += (double)gcols[(int64_t)rw * 768 + q * 64 + o]; }
        part[ls][q] = s;
    }
    __syncthreads();
    if (tid < ncol) {
        double s = 0;
        for (int ls = 0; ls < 21; ++ls) s += part[ls][tid];
        part[0][tid] = s;                              // (column tid: only this thread read it)
    }
    __syncthreads();
    if (tid == 0) {
        double all = 0, first, last;
        if (gcols3) { all = part[0][0]; first = part[0][1]; last = part[0][2]; }
        else { for (int q = 0; q < 12; ++q) all += part[0][q]; first = part[0][0]; last = part[0][11]; }
        Gs[0] = (float)(all - first);
        Gs[1] = (float)all;
        Gs[2] = (float)(all - last);
        if (db2 != nullptr) db2[o] = (float)all;
    }
    __syncthreads();
    if (tid < 192) {
        const int tap = tid >> 6, c = tid & 63;
        const float* src = slabs + o * 192 + tid;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = 0;
        for (; k + 8 <= S; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(k + u) * 64 * 192];
            a0 += v[0] + v[4]; a1 += v[1] + v[5]; a2 += v[2] + v[6]; a3 += v[3] + v[7];
        }
        for (; k < S; ++k) a0 += src[(int64_t)k * 64 * 192];
        const float P = (a0 + a1) + (a2 + a3);
        const float G = Gs[tap];
        const int base = (o * 64 + c) * 9;
        dW2[base + 3 + tap] = fmaf(stats1[2 * 64 + c], P, stats1[3 * 64 + c] * G);
        dW2[base + 0 + tap] = 0.f;
        dW2[base + 6 + tap] = 0.f;
        const float w = D::round(W2[base + 3 + tap]);
        red[0][tap][c] = w * G;
        red[1][tap][c] = w * P;
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        out_rows[((int64_t)o * 2 + which) * 64 + c] = (red[which][0][c] + red[which][1][c]) + red[which][2][c];
    }
}

// ------------------------------------------------------------------------------------
// BN-backward sums of the PREVIOUS layer without touching any N-sized tensor.  With g_v = g_y W
// (no dropout in between) and P = g_y^T r the raw weight-gradient product of this layer:
//     S1[k] = sum_n g_v[n][k]          = sum_j W[j][k] * db[j]          (db = column sums of g_y)
//     S2[k] = sum_n g_v[n][k] r[n][k]  = sum_j W[j][k] * P[j][k]
// out: gridDim.y "partial rows" [2][Kp] in the layout bn_bwd_finalize_kernel consumes (row y sums the
// j-slice y), so the reduction over j runs on Kp/64 x gridDim.y blocks instead of a handful.
//   mode 1 (fc1): P is in the internal k' = w*64 + c order, W in the reference order k = c*12 + w.
// grid (Kp/64, slices) blocks of 256 threads (64 columns x 4 row lanes).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_sums_from_wgrad_kernel(const float* __restrict__ P, const float* __restrict__ W,
                                                                     const float* __restrict__ db, float* __restrict__ out,
                                                                     int F, int Kp, int mode) {
    __shared__ double red[2][4][64];
    const int tid = threadIdx.x, cl = tid & 63, g = tid >> 6;
    const int kp = blockIdx.x * 64 + cl;
    const int k = mode == 1 ? (kp & 63) * 12 + (kp >> 6) : kp;
    const int per = (F + gridDim.y - 1) / gridDim.y;
    const int j0 = blockIdx.y * per, j1 = (j0 + per < F) ? j0 + per : F;
    double s1 = 0, s2 = 0;
#pragma unroll 4
    for (int j = j0 + g; j < j1; j += 4) {
        const double w = (double)W[(int64_t)j * Kp + k];
        s1 += w * (double)db[j];
        s2 += w * (double)P[(int64_t)j * Kp + kp];
    }
    red[0][g][cl] = s1;
    red[1][g][cl] = s2;
    __syncthreads();
    if (g == 0) {
        float* row = out + (int64_t)blockIdx.y * 2 * Kp;
        row[kp] = (float)(red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
        row[Kp + kp] = (float)(red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
    }
}
