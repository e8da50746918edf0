// Contrastive head: one wavefront per group of 41 windows.
//   z_hat = z/|z|, E_hat = E/|E| with E[c] = W_easy[:,c] + b_easy            (code/models.py:123-125, 457-458)
//   logits[g][i][j] = z_hat[g,i] . E_hat[label(g,j)]                          (code/models.py:127-129)
//   loss = 1/(2*G*41) * sum_g sum_i [ -log softmax_j(l[i,:])[y_i] - log softmax_i(l[:,i'])[y_i'] ]
//                                                                             (code/models.py:146-147, 204-207)
//   pred[g][i] = argmax_j l[i][j]                                             (code/models.py:149)
// and, when grads are requested, d loss / d z (through the normalisation) and d loss / d E_hat.
// 41 <= 64 lanes: lane i owns row i (and column i in the column pass); the 41x41 tile moves
// between the two views through a per-wave LDS tile with an odd pitch (conflict-free both ways);
// row/column softmax reductions are register loops per lane, group reductions are wave shuffles.
#pragma once
#include "common.cuh"

#define HEAD_T 41
#define HEAD_D 16
#define HEAD_WAVES 4

struct HeadArgs {
    const float* z;          // [N][16] encoder output, window order (b, t, v)
    const float* easy_w;     // (16,41)
    const float* easy_b;     // (16)
    const int64_t* labels;   // [B*41]
    int64_t G;               // groups = B*V
    int V;
    int want_grad;
    int dz_ld;               // row pitch of dz (elements)
    void* dz;                // [N][dz_ld] T (cols 0..15 written)
    float* logits;           // optional [G][41][41]
    int32_t* pred;           // [G][41]
    float* partials;         // [blocks][HEAD_PART] : loss sum, correct count, dE_hat[41][16]
    // GLOVE = true: the class embedding of position j of group b is row (b*41 + j) of zg
    // (the glove-angle encoder's output, one row per (group, class)) instead of row labels[..] of a shared table
    const float* zg;         // [B*41][16]
    void* dzg;               // [B*41][dzg_ld] T: dL/dzg through the normalisation (want_grad needs V == 1)
    int dzg_ld;
    // global negatives (SURVEY 8e extension, one-hot class table only): gneg = [2][41] device floats {G, H} from
    // gneg_* below, or nullptr = the reference's per-group column softmax
    const float* gneg;
};
#define GNEG_PART 64         // stride of the {G, H} table of the global-negatives extension
#define HEAD_PART 704        // 2 + 41*16 = 658 used, padded to a multiple of 64 for reduce_rows_kernel

template <typename T, bool GLOVE = false>
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
    using D = DT<T>;
    __shared__ float Eh[HEAD_T][HEAD_D];                 // normalised class table
    __shared__ float EhW[GLOVE ? HEAD_WAVES : 1][HEAD_T][HEAD_D];   // GLOVE: the group's own normalised class rows
    __shared__ float EnW[GLOVE ? HEAD_WAVES : 1][HEAD_T + 3];       //        and their norms
    __shared__ float Ls[HEAD_WAVES][HEAD_T][HEAD_T + 2]; // logits / dlogits tile per wave (odd pitch 43)
    __shared__ float Zs[HEAD_WAVES][HEAD_T][HEAD_D];     // z_hat rows per wave
    __shared__ float Cl[HEAD_WAVES][HEAD_T + 3];         // column log-sum-exp per wave
    __shared__ int Cls[HEAD_WAVES][HEAD_T + 3];          // class of position j in this group
    __shared__ float dE[HEAD_WAVES][HEAD_T][HEAD_D];     // per-wave accumulator of d/dE_hat (by class)
    __shared__ float wl[HEAD_WAVES], wc[HEAD_WAVES];
    __shared__ int Tg[HEAD_T + 3];                       // CE target column of row i = labels[i]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < HEAD_T; i += 256) {
        if constexpr (!GLOVE) {
            float e[HEAD_D], n = 0.f;
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) { e[d] = a.easy_w[d * HEAD_T + i] + a.easy_b[d]; n = fmaf(e[d], e[d], n); }
            n = sqrtf(n);
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) Eh[i][d] = e[d] / n;
        }
        Tg[i] = (int)a.labels[i];
    }
    for (int i = tid; i < HEAD_WAVES * HEAD_T * HEAD_D; i += 256) (&dE[0][0][0])[i] = 0.f;
    __syncthreads();

    const bool act = lane < HEAD_T;
    const int li = act ? lane : 0;
    const int tgt = Tg[li];
    const float cscale = 1.0f / (2.0f * (float)a.G * (float)HEAD_T);
    float loss_acc = 0.f, corr_acc = 0.f;

    for (int64_t g = (int64_t)blockIdx.x * HEAD_WAVES + wave; g < a.G; g += (int64_t)gridDim.x * HEAD_WAVES) {
        const int64_t b = g / a.V;
        const int v = (int)(g % a.V);
        const int64_t zrow = (b * HEAD_T + li) * a.V + v;
        if (act) Cls[wave][lane] = GLOVE ? lane : (int)a.labels[b * HEAD_T + lane];
        if constexpr (GLOVE) {
            if (act) {
                const float4* gp = (const float4*)(a.zg + (b * HEAD_T + lane) * HEAD_D);
                float e[HEAD_D], n = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t4 = gp[q];
                    e[4 * q] = t4.x; e[4 * q + 1] = t4.y; e[4 * q + 2] = t4.z; e[4 * q + 3] = t4.w;
                }
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) n = fmaf(e[d], e[d], n);
                n = sqrtf(n);
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) EhW[wave][lane][d] = e[d] / n;
                EnW[wave][lane] = n;
            }
        }
        float zh[HEAD_D], nz = 0.f;
        {
            const float4* zp = (const float4*)(a.z + zrow * HEAD_D);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = zp[q];
                zh[4 * q] = t4.x; zh[4 * q + 1] = t4.y; zh[4 * q + 2] = t4.z; zh[4 * q + 3] = t4.w;
            }
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) nz = fmaf(zh[d], zh[d], nz);
            nz = sqrtf(nz);
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) zh[d] = zh[d] / nz;
        }
        if (act) {
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) Zs[wave][lane][d] = zh[d];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- row pass: lane i owns logits[i][:] --------------------------------------
        // (the 41 logits of the row live in the wave's LDS tile, not in 41 registers: lanes >= 41
        //  shadow row 0 and never write)
        float mx = -INFINITY, lt = 0.f;
        int arg = 0;
#pragma unroll 4
        for (int j = 0; j < HEAD_T; ++j) {
            const float* e = GLOVE ? EhW[wave][j] : Eh[Cls[wave][j]];
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) s = fmaf(zh[d], e[d], s);
            if (act) Ls[wave][lane][j] = s;
            if (s > mx) { mx = s; arg = j; }
            if (j == tgt) lt = s;
        }
        __builtin_amdgcn_wave_barrier();
        float se = 0.f;
#pragma unroll 4
        // (__expf = one v_exp_f32: the arguments are differences of cosines, in [-2, 0]; the loss and the gradients stay
        //  within the tolerances of tests/test_gpu_parity.py, 2e-6 on the loss; 164 exponentials per lane and group: 72 -> 62 us)
        for (int j = 0; j < HEAD_T; ++j) se += __expf(Ls[wave][li][j] - mx);
        const float lse = mx + logf(se);
        if (act) {
            loss_acc += lse - lt;
            corr_acc += (arg == tgt) ? 1.f : 0.f;
            a.pred[g * HEAD_T + lane] = arg;
        }
        __builtin_amdgcn_wave_barrier();
        if (a.logits != nullptr) {
            float* lo = a.logits + g * (HEAD_T * HEAD_T);
            for (int i = lane; i < HEAD_T * HEAD_T; i += 64) lo[i] = Ls[wave][i / HEAD_T][i % HEAD_T];
        }
        // ---- column pass: lane j owns logits[:][j] -----------------------------------
        if (a.gneg == nullptr) {
            float cm = -INFINITY;
#pragma unroll 4
            for (int i = 0; i < HEAD_T; ++i) cm = fmaxf(cm, Ls[wave][i][li]);
            float cs = 0.f;
#pragma unroll 4
            for (int i = 0; i < HEAD_T; ++i) cs += __expf(Ls[wave][i][li] - cm);
            const float clse = cm + logf(cs);
            if (act) {
                Cl[wave][lane] = clse;
                loss_acc += clse - Ls[wave][tgt][lane];      // column j's target row is labels[j]
            }
        } else if (act) {
            // global negatives: the column of class c = Cls[j] sees its positive (row tgt of this group) and, through
            // G[c], every window of another class in the GLOBAL batch:  -pos + log(exp(pos) + G[c])
            const float pos = Ls[wave][tgt][lane];
            const float den = __expf(pos) + a.gneg[Cls[wave][lane]];
            Cl[wave][lane] = logf(den);                      // log of the column's denominator
            loss_acc += logf(den) - pos;
        }
        __builtin_amdgcn_wave_barrier();
        if (a.want_grad) {
            // dl[i][j] = c * (P_row + P_col - [j == y_i] - [i == y_j])
            float dzh[HEAD_D];
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) dzh[d] = 0.f;
            const float inv_se = 1.0f / se;
#pragma unroll 4
            for (int j = 0; j < HEAD_T; ++j) {
                const float lj = Ls[wave][li][j];
                float dl = __expf(lj - mx) * inv_se - ((j == tgt) ? 1.f : 0.f);
                if (a.gneg == nullptr) {
                    dl += __expf(lj - Cl[wave][j]) - ((Tg[j] == li) ? 1.f : 0.f);
                } else {
                    // positive of its column: exp(pos)/den - 1; a negative of column class c: exp(l) * H[c], H[c] = the sum over
                    // ALL groups of the global batch of 1/den (gneg_h_kernel); a row of the same class as the column that is
                    // not its positive cannot occur (one window per class and group)
                    dl += (Tg[j] == li) ? __expf(lj - Cl[wave][j]) - 1.f : __expf(lj) * a.gneg[GNEG_PART + Cls[wave][j]];
                }
                dl *= cscale;
                const float* e = GLOVE ? EhW[wave][j] : Eh[Cls[wave][j]];
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) dzh[d] = fmaf(dl, e[d], dzh[d]);
                if (act) Ls[wave][lane][j] = dl;      // own row, read above: the tile now holds dlogits
            }
            __builtin_amdgcn_wave_barrier();
            if (act) {
                // through the normalisation: dz = (dzh - zh (zh.dzh)) / |z|
                float dot = 0.f;
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) dot = fmaf(zh[d], dzh[d], dot);
                float o[HEAD_D];
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) o[d] = (dzh[d] - zh[d] * dot) / nz;
                T* dst = (T*)a.dz + zrow * a.dz_ld;
#pragma unroll
                for (int c = 0; c < HEAD_D / D::EPC; ++c) *(uint4*)(dst + c * D::EPC) = D::pack(o + c * D::EPC);
            }
            __builtin_amdgcn_wave_barrier();
            // d/dE_hat[class of j] += sum_i dl[i][j] * z_hat[i]      (lane j owns column j)
            if (act) {
                float accd[HEAD_D];
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) accd[d] = 0.f;
#pragma unroll 4
                for (int i = 0; i < HEAD_T; ++i) {
                    const float dl = Ls[wave][i][lane];
#pragma unroll
                    for (int d = 0; d < HEAD_D; ++d) accd[d] = fmaf(dl, Zs[wave][i][d], accd[d]);
                }
                if constexpr (GLOVE) {
                    // position j's embedding belongs to this group alone: through its normalisation, straight out
                    float dot = 0.f;
#pragma unroll
                    for (int d = 0; d < HEAD_D; ++d) dot = fmaf(EhW[wave][lane][d], accd[d], dot);
                    float o[HEAD_D];
#pragma unroll
                    for (int d = 0; d < HEAD_D; ++d) o[d] = (accd[d] - EhW[wave][lane][d] * dot) / EnW[wave][lane];
                    T* dst = (T*)a.dzg + (b * HEAD_T + lane) * a.dzg_ld;
#pragma unroll
                    for (int c = 0; c < HEAD_D / D::EPC; ++c) *(uint4*)(dst + c * D::EPC) = D::pack(o + c * D::EPC);
                } else {
                    const int c = Cls[wave][lane];
#pragma unroll
                    for (int d = 0; d < HEAD_D; ++d) atomicAdd(&dE[wave][c][d], accd[d]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    loss_acc = wave_sum(act ? loss_acc : 0.f);
    corr_acc = wave_sum(act ? corr_acc : 0.f);
    if (lane == 0) { wl[wave] = loss_acc; wc[wave] = corr_acc; }
    __syncthreads();
    float* part = a.partials + (int64_t)blockIdx.x * HEAD_PART;
    if (tid == 0) {
        part[0] = wl[0] + wl[1] + wl[2] + wl[3];
        part[1] = wc[0] + wc[1] + wc[2] + wc[3];
    }
    for (int i = tid; i < HEAD_T * HEAD_D; i += 256) {
        const float* p = &dE[0][0][0];
        part[2 + i] = p[i] + p[HEAD_T * HEAD_D + i] + p[2 * HEAD_T * HEAD_D + i] + p[3 * HEAD_T * HEAD_D + i];
    }
}

// one block: loss, correct count, class-table gradient (through E/|E| and E = W[:,c] + b)
__global__ __launch_bounds__(256) void head_finalize_kernel(const float* __restrict__ partials, int nblocks, int64_t G,
                                                            const float* __restrict__ easy_w, const float* __restrict__ easy_b,
                                                            int want_grad, float* __restrict__ out /* loss, correct */,
                                                            float* __restrict__ d_easy_w, float* __restrict__ d_easy_b) {
    __shared__ double dEh[HEAD_T][HEAD_D];
    __shared__ float dEc[HEAD_T][HEAD_D];
    const int tid = threadIdx.x;
    // column sums over the partial rows, rows in order, loads eight deep (one row per round trip made this kernel 33 us:
    // threads 0 and 1 walked 4 x 64 rows one load at a time); the two scalars go to the otherwise idle last threads
    auto colsum = [&](int col) {
        double s = 0;
        int r = 0;
        for (; r + 8 <= nblocks; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(int64_t)(r + u) * HEAD_PART + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; r < nblocks; ++r) s += (double)partials[(int64_t)r * HEAD_PART + col];
        return s;
    };
    if (tid >= 254) {
        const int k = tid - 254;
        const double s = colsum(k);
        out[k] = k == 0 ? (float)(s / (2.0 * (double)G * HEAD_T)) : (float)s;
    }
    if (!want_grad) return;
    for (int i = tid; i < HEAD_T * HEAD_D; i += 256) (&dEh[0][0])[i] = colsum(2 + i);
    __syncthreads();
    if (tid < HEAD_T) {
        float e[HEAD_D], n = 0.f;
        for (int d = 0; d < HEAD_D; ++d) { e[d] = easy_w[d * HEAD_T + tid] + easy_b[d]; n = fmaf(e[d], e[d], n); }
        n = sqrtf(n);
        float dot = 0.f;
        for (int d = 0; d < HEAD_D; ++d) dot += (e[d] / n) * (float)dEh[tid][d];
        for (int d = 0; d < HEAD_D; ++d) {
            const float v = ((float)dEh[tid][d] - (e[d] / n) * dot) / n;
            dEc[tid][d] = v;
            d_easy_w[d * HEAD_T + tid] = v;
        }
    }
    __syncthreads();
    if (tid < HEAD_D) {
        float s = 0.f;
        for (int c = 0; c < HEAD_T; ++c) s += dEc[c][tid];
        d_easy_b[tid] = s;
    }
}

// ------------------------------------------------------------------------------------
// Global negatives (no reference counterpart: the reference's column softmax of code/models.py:136-147 ranges
// over the 41 windows of ONE group).  With a shared class table the logit of window n against class k is s[n][k] =
// z_hat[n] . E_hat[k] whichever group n belongs to, so "all windows of other classes in the global batch" enter the
// column loss of class k only through
//     G[k] = sum over all windows n of the gathered batch with class(n) != k of exp(s[n][k])
// and its gradient through
//     H[k] = sum over all groups b of 1 / (exp(s[pos(b,k)][k]) + G[k]).
// gneg_g_kernel: one thread per window of the gathered z (rank-major rows, 41 per group, position t has class labels[t]);
// block partial rows [41]; also stores the window's positive logit.  gneg_h_kernel: the second sum, from the positives.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gneg_g_kernel(const float* __restrict__ z_all, int64_t n_all, const float* __restrict__ easy_w,
                                                      const float* __restrict__ easy_b, const int64_t* __restrict__ labels,
                                                      float* __restrict__ partials, float* __restrict__ pos) {
    __shared__ float Eh[HEAD_T][HEAD_D];
    __shared__ float acc[HEAD_T];
    __shared__ int cls[HEAD_T];
    const int tid = threadIdx.x;
    if (tid < HEAD_T) {
        float e[HEAD_D], n = 0.f;
#pragma unroll
        for (int d = 0; d < HEAD_D; ++d) { e[d] = easy_w[d * HEAD_T + tid] + easy_b[d]; n = fmaf(e[d], e[d], n); }
        n = sqrtf(n);
#pragma unroll
        for (int d = 0; d < HEAD_D; ++d) Eh[tid][d] = e[d] / n;
        acc[tid] = 0.f;
        cls[tid] = (int)labels[tid];
    }
    __syncthreads();
    float g[HEAD_T];
#pragma unroll
    for (int k = 0; k < HEAD_T; ++k) g[k] = 0.f;
    for (int64_t n = (int64_t)blockIdx.x * 256 + tid; n < n_all; n += (int64_t)gridDim.x * 256) {
        const float4* zp = (const float4*)(z_all + n * HEAD_D);
        float zh[HEAD_D], nz = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t4 = zp[q];
            zh[4 * q] = t4.x; zh[4 * q + 1] = t4.y; zh[4 * q + 2] = t4.z; zh[4 * q + 3] = t4.w;
        }
#pragma unroll
        for (int d = 0; d < HEAD_D; ++d) nz = fmaf(zh[d], zh[d], nz);
        nz = sqrtf(nz);
#pragma unroll
        for (int d = 0; d < HEAD_D; ++d) zh[d] = zh[d] / nz;
        const int c = cls[(int)(n % HEAD_T)];
#pragma unroll
        for (int k = 0; k < HEAD_T; ++k) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HEAD_D; ++d) s = fmaf(zh[d], Eh[k][d], s);
            if (k == c) pos[n] = s;
            else g[k] += __expf(s);
        }
    }
#pragma unroll
    for (int k = 0; k < HEAD_T; ++k) {
        const float v = wave_sum(g[k]);
        if ((tid & 63) == 0) atomicAdd(&acc[k], v);
    }
    __syncthreads();
    if (tid < GNEG_PART) partials[(int64_t)blockIdx.x * GNEG_PART + tid] = tid < HEAD_T ? acc[tid] : 0.f;
}

__global__ __launch_bounds__(256) void gneg_h_kernel(const float* __restrict__ pos, int64_t n_all, const float* __restrict__ G,
                                                      const int64_t* __restrict__ labels, float* __restrict__ partials) {
    __shared__ float acc[HEAD_T];
    __shared__ float Gs[HEAD_T];
    __shared__ int cls[HEAD_T];
    const int tid = threadIdx.x;
    if (tid < HEAD_T) { acc[tid] = 0.f; Gs[tid] = G[tid]; cls[tid] = (int)labels[tid]; }
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * 256 + tid; n < n_all; n += (int64_t)gridDim.x * 256) {
        const int c = cls[(int)(n % HEAD_T)];
        atomicAdd(&acc[c], 1.0f / (__expf(pos[n]) + Gs[c]));
    }
    __syncthreads();
    if (tid < GNEG_PART) partials[(int64_t)blockIdx.x * GNEG_PART + tid] = tid < HEAD_T ? acc[tid] : 0.f;
}

// ------------------------------------------------------------------------------------
// eval majority vote (code/models.py:151-163): pred (B, V=25, 41) -> for every prefix length
// win = 1..V the per-class mode over the first `win` samples; curve[b][win-1] = mean_t[mode == y_t].
// torch.mode returns the smallest of the most frequent values.  One thread per (b, t).
// ------------------------------------------------------------------------------------
__global__ void vote_kernel(const int32_t* __restrict__ pred, const int64_t* __restrict__ labels, int64_t B, int V,
                            float* __restrict__ curve /* [B][V] */, int32_t* __restrict__ y_pred /* [B][41] */) {
    const int64_t b = blockIdx.x;
    const int t = threadIdx.x;
    __shared__ float hit[32][64];
    unsigned char cnt[HEAD_T];
    for (int c = 0; c < HEAD_T; ++c) cnt[c] = 0;
    if (t < HEAD_T) {
        const int y = (int)labels[t];
        int best = 0, bestc = 0;
        for (int w = 0; w < V; ++w) {
            const int p = pred[(b * V + w) * HEAD_T + t];
            const int c = ++cnt[p];
            if (c > bestc || (c == bestc && p < best)) { best = p; bestc = c; }
            hit[w][t] = (best == y) ? 1.f : 0.f;
        }
        y_pred[b * HEAD_T + t] = best;
    }
    __syncthreads();
    if (t < V) {
        float s = 0.f;
        for (int i = 0; i < HEAD_T; ++i) s += hit[t][i];
        curve[b * V + t] = s / HEAD_T;
    }
}
