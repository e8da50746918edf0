// Contrastive head: one wavefront per group of 41 windows.
//   z_hat = z/|z|, E_hat = E/|E| with E[c] = W_easy[:,c] + b_easy            (code/models.py:123-125, 457-458)
//   logits[g][i][j] = z_hat[g,i] . E_hat[label(g,j)]                          (code/models.py:127-129)
//   loss = 1/(2*G*41) * sum_g sum_i [ -log softmax_j(l[i,:])[y_i] - log softmax_i(l[:,i'])[y_i'] ]
//                                                                             (code/models.py:146-147, 204-207)
//   pred[g][i] = argmax_j l[i][j]                                             (code/models.py:149)
// and, when grads are requested, d loss / d z (through the normalisation) and d loss / d E_hat.
// (the kernel's lane layout: above head_kernel)
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
    int dz_ld;               // row pitch of dz (elements): HEAD_LD
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

// ---- the kernel ------------------------------------------------------------------------------------------------------
// One wavefront per group, the 41x41 tile on the matrix cores (v_mfma_f32_16x16x4_f32, full fp32): lane l = (c = l & 15, q = l >> 4).
//   A operand: lane supplies A[m = c][k = q];  B operand: B[k = q][n = c];  result: lane holds D[m = 4q + r][n = c], r = 0..3.
// The tile is padded to 48x48 = 3x3 MFMA tiles and kept in registers TWICE, as L = Z E^T (rows i on (q, r), columns j on c) and as
// its transpose Lt = E Z^T (rows j on (q, r), columns i on c): a reduction "down the registers and across q" (two xor-shuffles) is the
// column softmax on L and the row softmax on Lt.  The gradient tile dl is formed ONCE, in L's layout -- where it is, as it stands, the
// A operand of dE_hat = dl^T Z_hat -- and crosses to the other layout through an LDS tile for dz_hat = dl E_hat.  Logits are carried
// in log2 units (z_hat is scaled by log2 e on its way into the MFMAs): every exponential is one v_exp_f32 of one subtraction, the
// loss is converted once per block.  The one-hot terms of dl, [j == y_i] + [i == y_j], are the same for every group: a register tile
// built once per kernel.
// Until round 3 every lane walked its row of 41 logits and its column of 41 through LDS with 16 FMAs per logit: 62 us at 4096 groups,
// 27 us at ONE group per wave (8 groups).
#define HEAD_LD 64           // row pitch (elements) of dz and dzg: the projection backward kernels contract over 64 columns
#define HEAD_P 20            // LDS row pitch (floats) of the 48x16 operand tiles: rows 4q + r of the four q land in disjoint bank sets
#define HEAD_TP 52           // LDS row pitch of the 41x48 dl tile (b128 reads along a row)
#define HEAD_LOG2E 1.4426950408889634f
#define HEAD_LN2 0.6931471805599453f
#define HEAD_PAD (-1e30f)    // a padded logit: finite (0 * pad = 0), its exponentials exactly 0

__device__ __forceinline__ f32x4 head_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float head_qsum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float head_qmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float head_exp2(float x) { return __builtin_amdgcn_exp2f(x); }      // v_exp_f32 (arguments <= 0 here, or tiny)
__device__ __forceinline__ float head_log2(float x) { return __builtin_amdgcn_logf(x); }       // v_log_f32 (arguments in [1, 48])

// F8L (BASELINE config 4, "fp8 MFMA logits GEMM"; CP_FP8 workspaces, one-hot class table): the logits come from ONE block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 per 16x16 tile on e4m3 copies of z_hat and E_hat (the 16 dims in the first 16 of the 128 k slots
// of lane group 0, scale 2^0) instead of four f32 MFMAs; loss, predictions and dl are those of the quantised logits, the two gradient
// products keep the f32 operands (straight-through).  e4m3 resolves [0.5, 1) in steps of 1/16: a logit moves by ~0.025 rms.
typedef __attribute__((ext_vector_type(8))) int head_i32x8;
template <typename T, bool GLOVE = false, bool GNEG = false, bool F8L = false>
__global__ __launch_bounds__(256, 2) void head_kernel(HeadArgs a) {
    using D = DT<T>;
    __shared__ float Eh[HEAD_T][HEAD_D];                               // normalised class table (one-hot path)
    __shared__ __attribute__((aligned(16))) float Zs[HEAD_WAVES][48][HEAD_P];   // z_hat rows of the wave's group (rows >= 41: zero)
    __shared__ __attribute__((aligned(16))) float Es[HEAD_WAVES][48][HEAD_P];   // E_hat row of position j of the group
    __shared__ __attribute__((aligned(16))) float Ts[HEAD_WAVES][HEAD_T][HEAD_TP];   // dl[i][j]
    __shared__ float Rl[HEAD_WAVES][48];                               // row log2-sum-exp2
    __shared__ float Nz[HEAD_WAVES][48], En[HEAD_WAVES][48];           // 1/|z_i|, 1/|E_j| (GLOVE)
    __shared__ float Dz[HEAD_WAVES][48], De[HEAD_WAVES][48];           // z_hat_i . dz_hat_i,  E_hat_j . dE_hat_j
    __shared__ int Cls[HEAD_WAVES][48];                                // class of position j in this group
    __shared__ float dE[HEAD_WAVES][HEAD_T][HEAD_D];                   // per-wave accumulator of d/dE_hat (by class): a fixed summation order
    __shared__ float wl[HEAD_WAVES], wc[HEAD_WAVES];
    __shared__ int Tg[48];                                             // CE target column of row i = labels[i]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    for (int i = tid; i < 48; i += 256) {
        if (i < HEAD_T) {
            if constexpr (!GLOVE) {
                float e[HEAD_D], n = 0.f;
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) { e[d] = a.easy_w[d * HEAD_T + i] + a.easy_b[d]; n = fmaf(e[d], e[d], n); }
                n = sqrtf(n);
#pragma unroll
                for (int d = 0; d < HEAD_D; ++d) Eh[i][d] = e[d] / n;
            }
            Tg[i] = (int)a.labels[i];
        } else {
            Tg[i] = -1;
        }
    }
    for (int i = tid; i < HEAD_WAVES * HEAD_T * HEAD_D; i += 256) (&dE[0][0][0])[i] = 0.f;
    __syncthreads();

    // group-invariant register tiles in L's layout (row i = 16ti + 4q + r, column j = 16tj + c):
    //   t1 = [j == y_i] (the row loss's target), t2 = [i == y_j] (the column loss's); padded rows and columns carry the target -1
    // and the padding of the third tile as an additive bias: index 32 + 4q + r past 40 -> HEAD_PAD
    f32x4 t1[3][3], t2[3][3];
    float tgc[3], padr[4];
#pragma unroll
    for (int t = 0; t < 3; ++t) tgc[t] = (float)Tg[16 * t + c];
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + 4 * q + r, yi = Tg[i];
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) {
                const int j = 16 * tj + c;
                t1[ti][tj][r] = (j == yi) ? 1.f : 0.f;
                t2[ti][tj][r] = (Tg[j] == i) ? 1.f : 0.f;
                if constexpr (!GNEG) t1[ti][tj][r] += t2[ti][tj][r];    // (one tile: both one-hot terms enter the loss and dl alike)
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) padr[r] = (32 + 4 * q + r < HEAD_T) ? 0.f : HEAD_PAD;
    const float fq4 = (float)(4 * q);
    const bool lane_ok2 = 32 + c < HEAD_T;                             // lane-side index of the third tile is a real row / column
    const float cscale = 1.0f / (2.0f * (float)a.G * (float)HEAD_T);
    float loss_acc = 0.f, corr_acc = 0.f;                              // (loss in log2 units)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int64_t g = (int64_t)blockIdx.x * HEAD_WAVES + wave; g < a.G; g += (int64_t)gridDim.x * HEAD_WAVES) {
        const int64_t b = g / a.V;
        const int v = (int)(g % a.V);
        // ---- operands: lane (c, q) holds dims 4q..4q+3 of rows 16t + c of Z_hat (times log2 e for the MFMAs) and of E_hat ----------
        f32x4 zA[3], eA[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int i = 16 * t + c;
            const bool ok = t < 2 || lane_ok2;
            const int ic = ok ? i : 0;
            f32x4 z4 = *(const f32x4*)(a.z + ((b * HEAD_T + ic) * a.V + v) * HEAD_D + 4 * q), e4;
            int cls;
            if constexpr (GLOVE) { e4 = *(const f32x4*)(a.zg + (b * HEAD_T + ic) * HEAD_D + 4 * q); cls = ic; }
            else { cls = (int)a.labels[b * HEAD_T + ic]; e4 = *(const f32x4*)&Eh[cls][4 * q]; }
            const float keep = ok ? 1.f : 0.f;
            const float iz = 1.0f / sqrtf(head_qsum(fmaf(z4[0], z4[0], fmaf(z4[1], z4[1], fmaf(z4[2], z4[2], z4[3] * z4[3])))));
            z4 *= iz * keep;
            if constexpr (GLOVE) {
                const float ie = 1.0f / sqrtf(head_qsum(fmaf(e4[0], e4[0], fmaf(e4[1], e4[1], fmaf(e4[2], e4[2], e4[3] * e4[3])))));
                e4 *= ie;
                if (q == 0) En[wave][i] = ie;
            }
            e4 *= keep;
            zA[t] = F8L ? z4 : z4 * HEAD_LOG2E; eA[t] = e4;
            *(f32x4*)&Zs[wave][i][4 * q] = z4;
            *(f32x4*)&Es[wave][i][4 * q] = e4;
            if (q == 0) { Nz[wave][i] = iz; Cls[wave][i] = cls; }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- L[ti][tj][r] = l2[16ti + 4q + r][16tj + c],  Lt[tj][ti][r] = l2[16ti + c][16tj + 4q + r],  l2 = logits * log2 e ------
        f32x4 L[3][3], Lt[3][3];
        if constexpr (F8L) {
            // lane group 0 carries row 16t + c of the operand: its 16 dims as 16 e4m3 bytes, k slots 0..15; every other slot is zero
            head_i32x8 zq[3], eq[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                int zd[4], ed[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 zv = *(const f32x4*)&Zs[wave][16 * t + c][4 * k], ev = *(const f32x4*)&Es[wave][16 * t + c][4 * k];
                    zd[k] = __builtin_amdgcn_cvt_pk_fp8_f32(zv[2], zv[3], __builtin_amdgcn_cvt_pk_fp8_f32(zv[0], zv[1], 0, false), true);
                    ed[k] = __builtin_amdgcn_cvt_pk_fp8_f32(ev[2], ev[3], __builtin_amdgcn_cvt_pk_fp8_f32(ev[0], ev[1], 0, false), true);
                    if (q != 0) { zd[k] = 0; ed[k] = 0; }
                }
                zq[t] = (head_i32x8){zd[0], zd[1], zd[2], zd[3], 0, 0, 0, 0};
                eq[t] = (head_i32x8){ed[0], ed[1], ed[2], ed[3], 0, 0, 0, 0};
            }
#pragma unroll
            for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) {
                    L[ti][tj] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(zq[ti], eq[tj], zero4, 0, 0, 0, 127, 0, 127) * HEAD_LOG2E;
                    Lt[tj][ti] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(eq[tj], zq[ti], zero4, 0, 0, 0, 127, 0, 127) * HEAD_LOG2E;
                }
        } else {
#pragma unroll
            for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) {
                    f32x4 x = zero4, y = zero4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { x = head_mfma(zA[ti][k], eA[tj][k], x); y = head_mfma(eA[tj][k], zA[ti][k], y); }
                    L[ti][tj] = x; Lt[tj][ti] = y;
                }
        }
        // ---- row softmax on Lt (row i = 16ti + c of l lives down the registers (tj, r) and across q) --------------------------------
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {
            float mx = -INFINITY, arg = 0.f;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (tj == 2) Lt[tj][ti][r] += padr[r];
                    const float s = Lt[tj][ti][r];
                    const bool gt = s > mx;
                    mx = gt ? s : mx;
                    arg = gt ? (float)(16 * tj + r) + fq4 : arg;
                }
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {                       // first maximum: ties go to the smaller column
                const float om = __shfl_xor(mx, m, 64), oa = __shfl_xor(arg, m, 64);
                const bool take = om > mx || (om == mx && oa < arg);
                mx = take ? om : mx;
                arg = take ? oa : arg;
            }
            float se = 0.f;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) se += head_exp2(Lt[tj][ti][r] - mx);
            const float rl = mx + head_log2(head_qsum(se));
            if (q == 0) {
                const int i = 16 * ti + c;
                Rl[wave][i] = rl;
                if (ti < 2 || lane_ok2) {
                    loss_acc += rl;
                    corr_acc += (arg == tgc[ti]) ? 1.f : 0.f;
                    a.pred[g * HEAD_T + i] = (int)arg;
                }
            }
        }
        if (a.logits != nullptr) {
            float* lo = a.logits + g * (HEAD_T * HEAD_T);
#pragma unroll
            for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ti + 4 * q + r, j = 16 * tj + c;
                        if (i < HEAD_T && j < HEAD_T) lo[i * HEAD_T + j] = L[ti][tj][r] * HEAD_LN2;
                    }
        }
        // ---- column pass on L (column j = 16tj + c); the padded rows of the third row tile go to HEAD_PAD for good ----------------------
#pragma unroll
        for (int tj = 0; tj < 3; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) L[2][tj][r] += padr[r];
        float ccl[3], cch[3];
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) {
            const bool jok = tj < 2 || lane_ok2;
            float cl, ch = 0.f;
            if constexpr (!GNEG) {
                float cm = -INFINITY;
#pragma unroll
                for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cm = fmaxf(cm, L[ti][tj][r]);
                cm = head_qmax(cm);
                float cs = 0.f;
#pragma unroll
                for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs += head_exp2(L[ti][tj][r] - cm);
                cl = cm + head_log2(head_qsum(cs));
                loss_acc += (q == 0 && jok) ? cl : 0.f;                // (minus the target logit: the element-wise pass below)
            } else {
                // global negatives: the column of class k = Cls[j] sees its positive (row labels[j] of this group) and, through
                // G[k], every window of another class in the GLOBAL batch:  -pos + log(exp(pos) + G[k])
                float pos = 0.f;
#pragma unroll
                for (int ti = 0; ti < 2 + 1; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pos = fmaf(t2[ti][tj][r], L[ti][tj][r], pos);
                pos = head_qsum(pos);
                const int k = Cls[wave][jok ? 16 * tj + c : 0];
                cl = head_log2(head_exp2(pos) + a.gneg[k]);
                ch = a.gneg[GNEG_PART + k];
                loss_acc += (q == 0 && jok) ? cl - pos : 0.f;
            }
            ccl[tj] = cl; cch[tj] = ch;
        }
        // the target logits of the row loss (l[i][labels[i]]) and of the per-group column loss (l[labels[j]][j])
#pragma unroll
        for (int ti = 0; ti < 3; ++ti)
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    loss_acc = fmaf(-t1[ti][tj][r], L[ti][tj][r], loss_acc);
                }
        __builtin_amdgcn_wave_barrier();
        if (!a.want_grad) continue;
        // ---- dl = c * (P_row + P_col - [j == y_i] - [i == y_j]); a padded element's exponentials are exactly 0 -----------------------
        f32x4 dzh[3] = {zero4, zero4, zero4}, dEp[3] = {zero4, zero4, zero4};
        float dote[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < 3; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + 4 * q + r;                     // row i on (q, r), column j = 16tj + c
                const bool iok = ti < 2 || 4 * q + r < HEAD_T - 32;
                const float rl = Rl[wave][i];
                const float zB = Zs[wave][i][c];
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) {
                    const float s = L[ti][tj][r];
                    float dl = head_exp2(s - rl) - t1[ti][tj][r];
                    if constexpr (!GNEG) dl += head_exp2(s - ccl[tj]);
                    // positive of its column: exp(pos)/den - 1; a negative of column class k: exp(l) * H[k], H[k] = the sum over ALL groups
                    // of the global batch of 1/den (gneg_h_kernel); a row of the column's class that is not its positive cannot occur
                    else dl += t2[ti][tj][r] != 0.f ? head_exp2(s - ccl[tj]) - 1.f : head_exp2(s) * cch[tj];
                    dl *= cscale;
                    if (iok) Ts[wave][iok ? i : 0][16 * tj + c] = dl;
                    if constexpr (GLOVE) dote[tj] = fmaf(dl, s, dote[tj]);
                    dEp[tj] = head_mfma(dl, zB, dEp[tj]);              // A[m = c][k = q] = dl[i][16tj + c],  B[k = q][n = c] = z_hat[i][c]
                }
            }
        __builtin_amdgcn_wave_barrier();
        // dz_hat = dl E_hat: A[m = c][k = q] = dl[16ti + c][16tj + 4q + r] (four consecutive columns of the LDS tile),
        //                    B[k = q][n = c] = E_hat[16tj + 4q + r][c];  z_hat_i . dz_hat_i = sum_j dl[i][j] l[i][j]
        float dotz[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) {
            f32x4 dlt[3];
#pragma unroll
            for (int ti = 0; ti < 3; ++ti) dlt[ti] = *(const f32x4*)&Ts[wave][(ti < 2 || lane_ok2) ? 16 * ti + c : 0][16 * tj + 4 * q];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool jok = tj < 2 || 4 * q + r < HEAD_T - 32;
                const float eB = Es[wave][16 * tj + 4 * q + r][c];
#pragma unroll
                for (int ti = 0; ti < 3; ++ti) {
                    const float dl = (tj == 2 && !jok) ? 0.f : dlt[ti][r];      // (columns 41..47 of the tile hold the padded columns' dl)
                    dotz[ti] = fmaf(dl, Lt[tj][ti][r], dotz[ti]);
                    dzh[ti] = head_mfma(dl, eB, dzh[ti]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float dz = head_qsum(dotz[t]) * HEAD_LN2;
            if (q == 0) Dz[wave][16 * t + c] = dz;
            if constexpr (GLOVE) {
                const float de = head_qsum(dote[t]) * HEAD_LN2;
                if (q == 0) De[wave][16 * t + c] = de;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- out: dz through the normalisation, dz = (dz_hat - z_hat (z_hat . dz_hat)) / |z|; rows are 64 wide for the projection's
        //      backward kernels, which contract over all 64: columns 16..63 are written as zeros here (no memset launch)
        const T zero_t = (T)0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * t + 4 * q + r;
                if (t == 2 && i >= HEAD_T) continue;
                {
                    const float o = (dzh[t][r] - Zs[wave][i][c] * Dz[wave][i]) * Nz[wave][i];
                    T* dst = (T*)a.dz + ((b * HEAD_T + i) * a.V + v) * HEAD_LD + c;
                    D::store(dst, o);
                    dst[16] = zero_t; dst[32] = zero_t; dst[48] = zero_t;
                }
                if constexpr (GLOVE) {
                    // position i's embedding belongs to this group alone: through its normalisation, straight out
                    const float o = (dEp[t][r] - Es[wave][i][c] * De[wave][i]) * En[wave][i];
                    T* dst = (T*)a.dzg + (b * HEAD_T + i) * HEAD_LD + c;
                    D::store(dst, o);
                    dst[16] = zero_t; dst[32] = zero_t; dst[48] = zero_t;
                } else {
                    atomicAdd(&dE[wave][Cls[wave][i]][c], dEp[t][r]);  // d/dE_hat[class of position i]
                }
            }
        __builtin_amdgcn_wave_barrier();
    }
    loss_acc = wave_sum(loss_acc) * HEAD_LN2;
    corr_acc = wave_sum(corr_acc);
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
