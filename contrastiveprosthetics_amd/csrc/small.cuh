// Small-batch form of the fc stack (the reference's own operating point: code/go.sh:6 trains at --batch_size 8, code/train.py:253
// defaults to 32, and the random search of code/train.py:140-166 runs 150 one-epoch trainings at that size: N = 41 B = 328 .. 2,624
// windows).  At these sizes a step is a chain of DEPENDENT launches -- train-mode BatchNorm makes every layer a grid-wide barrier --
// and its time is (number of launches) x (launch boundary + a kernel's fixed cost), not bytes or FLOPs: the large-batch kernels
// (one persistent workgroup per CU, every workgroup loading 1/256 of a weight matrix, separate fold / finalize / reduce launches
// around every GEMM) took ~100 launches x 7.6 us = 0.84 ms for 328 windows.  Here ONE launch does everything between two BatchNorm
// barriers:
//   * BatchNorm sums are accumulated by 64-bit INTEGER atomics (each workgroup's f32 partial sum, exactly representable, times a
//     power of two): integer addition is associative, so the totals do not depend on the order the workgroups arrive in -- no
//     partial rows, no finalize launch, run-to-run reproducible.  A consumer turns the two totals per column into scale / shift
//     (or the backward coefficients) itself, 2 loads per column.  (Tried first: every consumer workgroup folding the producer's
//     partial rows -- 45 KB per workgroup at 328 rows, 336 KB at 2,624: 15 us of a 21 us launch, and 130-320 us at 64 groups.)
//   * the BatchNorm affine and the dropout mask are applied to the A operand while it is staged (no folded weight copy per layer,
//     no dropout pass); backward: BatchNorm + ReLU backward of the layer is applied to the incoming gradient while it is staged, in
//     the data-gradient AND in the weight-gradient role of the same launch (nothing is written in between);
//   * weight gradients: 64 x 64 tiles, whole-batch sums up to 256 rows, otherwise 2..8 row splits into slabs that ONE launch sums at the
//     end of the backward pass (one split at 328 rows was tried: the tiles then walk all rows in a latency-bound loop, slower);
//   * one weight-preparation launch per step (sm_prep_kernel: copies, transposes, conv2's images, the zeroed totals); the conv stage's
//     BatchNorm totals are the same fixed-point atomics (conv_kernels.cuh: bn1 / acc_out);
//   * up to 24 groups the NT launches split their contraction over wave pairs (SmTile<true>: 64-feature tiles).
// Tiles are small (32 rows x 128 or 64 features; 64 x 64 for weight gradients) so that 328 rows still give 44 - 88 workgroups per role.
// T = float (parity path: the same f32 MFMA chain as the large-batch kernels, different summation grouping) or bf16.
// A grid barrier inside one persistent launch was priced first (guide, price list): 4-7 us per barrier against 1.5-2 us per launch
// boundary -- the boundaries are cheaper, and a chain of launches cannot hang.
#pragma once
#include "gemm_nt.cuh"
#include "gemm_tn.cuh"

#define SM_BM 32
#define SM_BN 128
#define SM_MAX_WINDOWS 2624          // 64 groups

// ---- BatchNorm statistics of the producer layer, finalised by the consumer ------------------------------------------------
// partials: [nrows][2][C] (sum, sum of squares).  All threads of the block take part; s_lds / t_lds: scale = gamma * invstd,
// shift = beta - mean * scale.  The `writer` block also stores [mean, invstd, scale, shift] and updates the running statistics
// (momentum, unbiased variance), exactly as bn_finalize_kernel does.
// (SM_ACT_SHIFT, SM_GRAD_SHIFT, sm_acc_add / sm_acc_get, SmBN, sm_bn_channel: common.cuh -- the conv kernels use them too)
template <int NT>
__device__ __forceinline__ void sm_finalize_stats(const SmBN& b, float* s_lds, float* t_lds, bool writer) {
    for (int c = threadIdx.x; c < b.C; c += NT) {
        if (b.acc == nullptr) {
            s_lds[c] = b.stats[2 * b.C + c];
            t_lds[c] = b.stats[3 * b.C + c];
            continue;
        }
        const double s1 = sm_acc_get(b.acc + c, SM_ACT_SHIFT), s2 = sm_acc_get(b.acc + b.C + c, SM_ACT_SHIFT);
        const double mu = s1 / b.count;
        double vb = s2 / b.count - mu * mu;
        if (vb < 0) vb = 0;
        const float mean = (float)mu, var = (float)vb;
        const float invstd = 1.0f / sqrtf(var + b.eps);
        const float sc = b.gamma[c] * invstd;
        const float sh = b.beta[c] - mean * sc;
        s_lds[c] = sc;
        t_lds[c] = sh;
        if (writer) {
            b.stats[0 * b.C + c] = mean;
            b.stats[1 * b.C + c] = invstd;
            b.stats[2 * b.C + c] = sc;
            b.stats[3 * b.C + c] = sh;
            if (b.update_running && b.running_mean) {
                const double unb = b.count > 1 ? vb * b.count / (b.count - 1) : vb;
                b.running_mean[c] = (1.f - b.momentum) * b.running_mean[c] + b.momentum * mean;
                b.running_var[c] = (1.f - b.momentum) * b.running_var[c] + b.momentum * (float)unb;
            }
        }
    }
    __syncthreads();
}

// one chunk of A' = dropout(BatchNorm(r)) from the saved activation, scale / shift in LDS indexed by (column & smask) (channel counts
// are powers of two: 64 for the conv stack, else 512; the first version's `% smod` and 16 scalar LDS reads per chunk were a third of
// a k-loop step's 1.6 us: with one workgroup per CU nothing hides a staging thread's own instructions)
template <typename T>
__device__ __forceinline__ uint4 sm_bn_drop_chunk(const uint4& in, const float* s_lds, const float* t_lds, int f, int smask, uint32_t key,
                                                  uint32_t row, uint32_t C, uint32_t thresh, float inv_keep) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    float v[EPC], sc[EPC], sh[EPC];
    D::unpack(in, v);
    const int c0 = f & smask;                                  // chunks never straddle a channel block (EPC divides 64)
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) {
        const float4 a = *(const float4*)(s_lds + c0 + 4 * q), b = *(const float4*)(t_lds + c0 + 4 * q);
        sc[4 * q] = a.x; sc[4 * q + 1] = a.y; sc[4 * q + 2] = a.z; sc[4 * q + 3] = a.w;
        sh[4 * q] = b.x; sh[4 * q + 1] = b.y; sh[4 * q + 2] = b.z; sh[4 * q + 3] = b.w;
    }
    if (thresh != 0) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
            const uint2 q = dropout_quad(key, row, C, (uint32_t)(f + e));
            v[e] = fmaf(v[e], sc[e], sh[e]) * dropout_scale(q.x, 0, thresh, inv_keep);
            v[e + 1] = fmaf(v[e + 1], sc[e + 1], sh[e + 1]) * dropout_scale(q.x, 1, thresh, inv_keep);
            v[e + 2] = fmaf(v[e + 2], sc[e + 2], sh[e + 2]) * dropout_scale(q.y, 0, thresh, inv_keep);
            v[e + 3] = fmaf(v[e + 3], sc[e + 3], sh[e + 3]) * dropout_scale(q.y, 1, thresh, inv_keep);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
    }
    return D::pack(v);
}

// ---- the two main loops, latency first ----------------------------------------------------------------------------------------
// With 44-170 workgroups on 256 CUs there is ONE workgroup per CU and nothing to overlap its memory latency with, so the loops keep as
// many loads in flight as the registers hold: a step moves NSUB sub-tiles of BK per operand (16-bit: 4 x 64 k -- a K = 512 contraction is
// TWO steps, both issued before anything else happens in the kernel; f32: 2 x 32 k), into one of two register sets; the loads of step
// i + 2 are issued as soon as step i's registers have gone to LDS.  `mid()` runs between the first two steps' loads and their first use:
// the kernels pass their statistics prologue (a round trip of its own, and a barrier) there, so it hides under the operand fetch.
// One LDS stage: store, barrier, MFMAs, barrier -- with every load already in flight the second buffer bought nothing.
// History (tools/small_stamps.py, k loop of one fc forward launch at 328 rows, bf16): one 64-deep sub-tile per step, loads one step
// ahead, 21 us per layer; two sub-tiles, loads two steps ahead but TRANSFORMED in the load phase (every step waited for the load it had
// just issued) 6.5 us; raw loads, transforms at store time 4.6 us; this form: see DESIGN.md 7g.
template <typename T> struct SmNsub { static constexpr int v = sizeof(T) == 2 ? 4 : 2; };
// f32: a wave's k loop is a chain of 64-cycle v_mfma_f32_32x32x2_f32 on one accumulator (256 of them for K = 512: 7.8 us at best, 13 us
// measured, of a 17 us launch), so the contraction is split over wave PAIRS: 64-feature tiles, wave w = feature group w & 1, k half
// w >> 1 of every 32-deep sub-tile; the two partial tiles meet in LDS in the epilogue (0.45 -> 0.37 ms per step at 8 groups).
// bf16, and f32 from 25 groups up: 128-feature tiles, no split (at 32 groups the split's 328 workgroups need a second round: 0.55 -> 0.60 ms).
// (the split doubles the workgroups: it is used while they still fit one round of the 256 CUs -- sm_ksplit())
template <bool KS> struct SmTile { static constexpr int BN = KS ? 64 : SM_BN; static constexpr bool KSPLIT = KS; };
template <typename T> static inline bool sm_ksplit(int64_t n_windows) { return (n_windows + SM_BM - 1) / SM_BM * 8 <= 256; }
// NT: acc += A'[32 rows][K] x W[BN rows][K]^T for the wave's 32 x 32 piece (W rows wrow .. wrow+31).  loadA(sub) / loadW(sub, i) return the
// thread's 16-byte chunk (row tid >> 3 [+ 32 i], chunk tid & 7) of sub-tile `sub`; xformA(raw, sub) turns the raw A chunk into the operand.
template <typename T, int BN, typename RawA, bool KSPLIT = false, typename FA, typename XA, typename FW, typename Mid>
__device__ __forceinline__ void sm_nt_loop(f32x16& acc, unsigned char* smem, int nsub, FA&& loadA, XA&& xformA, FW&& loadW, int wrow, bool do_mma,
                                           Mid&& mid, long long* dbg = nullptr) {
    constexpr int NSUB = SmNsub<T>::v;
    constexpr int A_BYTES = SM_BM * 128, W_BYTES = BN * 128, SUB = A_BYTES + W_BYTES, W_IT = BN / 32;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, sc = tid & 7, sr = tid >> 3;
    const int nit = (nsub + NSUB - 1) / NSUB;
    RawA ar[2][NSUB];
    uint4 wr[2][NSUB][W_IT];
    auto load = [&](int it, int set) {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int sub = it * NSUB + u;
            const int subc = sub < nsub ? sub : nsub - 1;       // (a contraction shorter than a step: loaded again, zeroed below)
#pragma unroll
            for (int i = 0; i < W_IT; ++i) wr[set][u][i] = loadW(subc, i);
            ar[set][u] = loadA(subc);
        }
    };
    auto store = [&](int it, int set) {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int sub = it * NSUB + u;
            const bool ok = sub < nsub;
            unsigned char* As = smem + u * SUB;
            unsigned char* Ws = As + A_BYTES;
            const uint4 av = xformA(ar[set][u], ok ? sub : nsub - 1);
            *(uint4*)(As + lds_tile_off(sr, sc)) = ok ? av : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < W_IT; ++i) *(uint4*)(Ws + lds_tile_off(sr + 32 * i, sc)) = ok ? wr[set][u][i] : make_uint4(0, 0, 0, 0);
        }
    };
    auto compute = [&]() {
        if (!do_mma) return;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const unsigned char* As = smem + u * SUB;
            const unsigned char* Ws = As + A_BYTES;
#pragma unroll
            for (int kq = 0; kq < (KSPLIT ? 2 : 4); ++kq) {
                const int ks = KSPLIT ? 2 * (int)(threadIdx.x >> 7) + kq : kq;      // KSPLIT: waves 0, 1 the first half of the sub-tile, 2, 3 the second
                const uint4 fw = *(const uint4*)(Ws + lds_tile_off(wrow + r, 2 * ks + h));
                const uint4 fs = *(const uint4*)(As + lds_tile_off(r, 2 * ks + h));
                mma_chunk<T>(fw, fs, acc);
            }
        }
    };
#ifdef SM_STAMP
    auto lstamp = [&](int slot) {
        __builtin_amdgcn_sched_barrier(0);
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[slot] = (long long)__builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
    };
#else
    auto lstamp = [&](int) {};
#endif
    load(0, 0);
    if (nit > 1) load(1, 1);
    lstamp(0);
    mid();
    lstamp(1);
    for (int it = 0; it < nit; it += 2) {
        store(it, 0);
        lstamp(2 + 4 * it);
        __syncthreads();
        if (it + 2 < nit) load(it + 2, 0);
        lstamp(3 + 4 * it);
        compute();
        lstamp(4 + 4 * it);
        __syncthreads();
        lstamp(5 + 4 * it);
        if (it + 1 >= nit) break;
        store(it + 1, 1);
        lstamp(6 + 4 * it);
        __syncthreads();
        if (it + 3 < nit) load(it + 3, 1);
        lstamp(7 + 4 * it);
        compute();
        lstamp(8 + 4 * it);
        __syncthreads();
        lstamp(9 + 4 * it);
    }
}
template <typename T, int BN> struct SmNT {
    static constexpr int LOOP_BYTES = SmNsub<T>::v * (SM_BM * 128 + BN * 128);
    static constexpr int C_PITCH = BN * (int)sizeof(T) + 16, C_BYTES = SM_BM * C_PITCH, RED_BYTES = 2 * 256 * DT<T>::EPC * 4;
    static constexpr int F_PITCH = BN * 4 + 16, F_BYTES = SM_BM * F_PITCH;   // an f32 image of the tile: one per k half of a split contraction
    static constexpr int EPI_BYTES = 2 * F_BYTES + RED_BYTES;
    static constexpr int BYTES = LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES;
};

// TN: acc += X'[rows][64 cols]^T x Y'[rows][64 cols] over the rows m_begin .. m_end of both operands (sub-steps of 32 rows), for the wave's
// 32 x 32 piece (X columns wp*32.., Y columns wq*32..).  loadX(m, ch) / loadY(m, ch): the raw 16-byte chunk ch of row m;
// xformX(raw, m, ch, ok) / xformY(raw, m, ch): the operands.
template <typename T, typename RawX, typename FX, typename XX, typename FY, typename XY, typename Mid>
__device__ __forceinline__ void sm_tn_loop(f32x16& acc, unsigned char* smem, int64_t m_begin, int64_t m_end, FX&& loadX, XX&& xformX, FY&& loadY,
                                           XY&& xformY, int wp, int wq, Mid&& mid) {
    using D = DT<T>;
    constexpr int NSUB = SmNsub<T>::v;
    constexpr int EPC = D::EPC, KSTEP = D::KSTEP, CPRX = 64 / EPC, RS = 256 / CPRX, IT = 32 / RS;
    constexpr int PX = TNPitch<T, 64>::value, XB = 32 * PX, SUB = 2 * XB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int rs = tid / CPRX, ch = tid % CPRX;
    const int nit = (int)((m_end - m_begin + 32 * NSUB - 1) / (32 * NSUB));
    RawX xr[2][NSUB][IT];
    uint4 yr[2][NSUB][IT];
    auto row_of = [&](int it, int u, int q) -> int64_t { return m_begin + ((int64_t)it * NSUB + u) * 32 + rs + RS * q; };
    auto load = [&](int it, int set) {              // (raw loads only, rows clamped: the transforms run at store time, see sm_nt_loop)
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int q = 0; q < IT; ++q) {
                const int64_t m = row_of(it, u, q);
                const int64_t mc = m < m_end ? m : m_end - 1;
                xr[set][u][q] = loadX(mc, ch);
                yr[set][u][q] = loadY(mc, ch);
            }
    };
    auto store = [&](int it, int set) {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            unsigned char* Xs = smem + u * SUB;
            unsigned char* Ys = Xs + XB;
#pragma unroll
            for (int q = 0; q < IT; ++q) {
                const int64_t m = row_of(it, u, q);
                const bool ok = m < m_end;
                const int64_t mc = ok ? m : m_end - 1;
                const uint4 vx = xformX(xr[set][u][q], mc, ch, ok), vy = xformY(yr[set][u][q], mc, ch);
                *(uint4*)(Xs + (rs + RS * q) * PX + ch * 16) = ok ? vx : make_uint4(0, 0, 0, 0);
                *(uint4*)(Ys + (rs + RS * q) * PX + ch * 16) = ok ? vy : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto compute = [&]() {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const unsigned char* Xs = smem + u * SUB;
            const unsigned char* Ys = Xs + XB;
#pragma unroll
            for (int ks = 0; ks < 32 / KSTEP; ++ks) {
                uint4 fx, fy;
                if constexpr (sizeof(T) == 2) {
                    fx = tn_frag_bf16<PX>(Xs, ks * KSTEP, wp * 32, lane);
                    fy = tn_frag_bf16<PX>(Ys, ks * KSTEP, wq * 32, lane);
                } else {
                    fx = tn_frag_f32<PX>(Xs, ks * KSTEP, wp * 32, lane);
                    fy = tn_frag_f32<PX>(Ys, ks * KSTEP, wq * 32, lane);
                }
                mma_chunk<T>(fx, fy, acc);
            }
        }
    };
    if (nit > 0) load(0, 0);
    if (nit > 1) load(1, 1);
    mid();
    for (int it = 0; it < nit; it += 2) {
        store(it, 0);
        __syncthreads();
        if (it + 2 < nit) load(it + 2, 0);
        compute();
        __syncthreads();
        if (it + 1 >= nit) break;
        store(it + 1, 1);
        __syncthreads();
        if (it + 3 < nit) load(it + 3, 1);
        compute();
        __syncthreads();
    }
}
template <typename T> struct SmTN {
    static constexpr int BYTES = SmNsub<T>::v * 2 * 32 * TNPitch<T, 64>::value;
};

// ---------------------------------------------------------------------------------------------------------------------------
// forward of one fc layer (MODE 0):  r_out = relu(dropout(BN_in(r_in)) W^T + b), partial BatchNorm sums of r_out per row tile;
// MODE 1 (projection): z = dropout(BN_in(r_in)) W_last^T, f32, 16 valid features of a 32-row padded weight, no sums.
// grid = tiles_m * (F / SmTile<KS>::BN) (MODE 1: tiles_m); 256 threads = 4 waves, bf16: wave w = features f0 + 32 w .. +31 of the 32-row tile;
// f32: wave w = features f0 + 32 (w & 1) .. +31, k half w >> 1 (SmTile).
// ---------------------------------------------------------------------------------------------------------------------------
struct SmFwdArgs {
    const void* A;           // [N][K] T: the previous layer's stored post-ReLU output
    const void* W;           // [F or 32][K] T: this step's copy of the weights (fc1: in the internal column order)
    const float* bias;       // [F] (MODE 0)
    void* C;                 // MODE 0: [N][512] T;  MODE 1: [N][16] f32
    long long* out_acc;      // MODE 0: [2][512] fixed-point totals of the output (zeroed at the start of the step)
    SmBN bn_in;              // the input's BatchNorm, finalised here
    int smod;                // channels of bn_in (64 for fc1's input, the conv stack; else K)
    int64_t N;
    int K;
    uint32_t dp_thresh, dp_key;
    const uint32_t* dp_salt;
    float dp_inv_keep;
};

// epilogue shared by the forward and the data-gradient role: the 32 x BN tile (already in the LDS image Cs, T) leaves as 16-byte row
// segments; per column sums of v and of v * w (w = v itself, or the matching element of a second tensor) -> out_partials
template <typename T, int BN, bool KS = false, typename FV>
__device__ __forceinline__ void sm_store_tile(unsigned char* smem, void* Cout, int64_t ldc, int64_t m0, int64_t N, int c0, float* out_partials,
                                              int64_t tile_m, int out_ld, long long* out_acc, int acc_shift, FV&& per_chunk) {
    // The tile sits in LDS as ONE image in T (row pitch C_PITCH) or, KS, as TWO f32 images of partial sums (the k halves, row pitch
    // F_PITCH, F_BYTES apart) that are added here.  per_chunk(v[EPC], m, column, s1, s2) finishes the values in f32 (bias, ReLU,
    // dropout mask, ...) and accumulates the two column sums; what it leaves in v is rounded to T and stored.
    using D = DT<T>;
    using G = SmNT<T, BN>;
    constexpr int EPC = D::EPC, C_PITCH = G::C_PITCH, F_PITCH = G::F_PITCH, F_BYTES = G::F_BYTES;
    constexpr int CPR = BN / EPC, RPP = 256 / CPR;
    const int tid = threadIdx.x;
    float* red = (float*)(smem + 2 * F_BYTES);
    const int cc = tid % CPR, rr = tid / CPR;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;
    T* Cg = (T*)Cout;
#pragma unroll
    for (int p = 0; p < SM_BM / RPP; ++p) {
        const int row = rr + p * RPP;
        const int64_t m = m0 + row;
        if (m < N) {
            float v[EPC];
            if constexpr (KS) {
#pragma unroll
                for (int q = 0; q < EPC / 4; ++q) {
                    const float4 x = *(const float4*)(smem + row * F_PITCH + (cc * EPC + 4 * q) * 4);
                    const float4 y = *(const float4*)(smem + F_BYTES + row * F_PITCH + (cc * EPC + 4 * q) * 4);
                    v[4 * q] = x.x + y.x; v[4 * q + 1] = x.y + y.y; v[4 * q + 2] = x.z + y.z; v[4 * q + 3] = x.w + y.w;
                }
            } else {
                D::unpack(*(const uint4*)(smem + row * C_PITCH + cc * 16), v);
            }
            per_chunk(v, m, c0 + cc * EPC, s1, s2);
            *(uint4*)(Cg + m * ldc + c0 + cc * EPC) = D::pack(v);
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        red[(0 * RPP + rr) * BN + cc * EPC + e] = s1[e];
        red[(1 * RPP + rr) * BN + cc * EPC + e] = s2[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
        const int which = tid / BN, col = tid % BN;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < RPP; ++q) s += red[(which * RPP + q) * BN + col];
        if (out_acc != nullptr) sm_acc_add(out_acc + (int64_t)which * out_ld + c0 + col, s, acc_shift);
        else out_partials[(tile_m * 2 + which) * out_ld + c0 + col] = s;      // (fc1's data gradient: rows for the conv tail's finalize launch)
    }
}

template <typename T, int MODE, bool KS = false>
__global__ __launch_bounds__(256) void sm_fc_fwd_kernel(SmFwdArgs a) {
    using D = DT<T>;
    constexpr int EPC = D::EPC, BK = D::BK, BM = SM_BM, BN = MODE == 0 ? SmTile<KS>::BN : 32;
    constexpr bool KSPLIT = MODE == 0 && KS;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SmNT<T, BN>::BYTES];
    __shared__ __attribute__((aligned(16))) float s_in[512], t_in[512];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    constexpr int tiles_f = MODE == 0 ? 512 / BN : 1;
    const int tile_f = blockIdx.x % tiles_f;
    const int64_t tile_m = blockIdx.x / tiles_f;
    const int64_t m0 = tile_m * BM;
    const int f0 = tile_f * BN;

#ifdef SM_STAMP
    // diagnostic build (tools/small_stamps.py): 100 MHz timestamps of workgroup 0 into the unused tail of the output accumulator block
    auto stamp = [&](int slot) {
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0) a.out_acc[1100 + slot] = (long long)__builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
    };
#else
    auto stamp = [&](int) {};
#endif
    stamp(0);
    const uint32_t key = a.dp_thresh != 0 ? (a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key) : 0u;

    const T* __restrict__ Ag = (const T*)a.A;
    const T* __restrict__ Wg = (const T*)a.W;
    const int sc = tid & 7, sr = tid >> 3;
    const int64_t m_st = m0 + sr, mc_st = m_st < a.N ? m_st : a.N - 1;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    sm_nt_loop<T, BN, uint4, KSPLIT>(
        acc, smem, a.K / BK,
        [&](int sub) -> uint4 { return *(const uint4*)(Ag + mc_st * a.K + sub * BK + sc * EPC); },
        [&](const uint4& raw, int sub) -> uint4 {
            const uint4 v = sm_bn_drop_chunk<T>(raw, s_in, t_in, sub * BK + sc * EPC, a.smod - 1, key, (uint32_t)mc_st, (uint32_t)a.K, a.dp_thresh,
                                                a.dp_inv_keep);
            return m_st < a.N ? v : make_uint4(0, 0, 0, 0);
        },
        [&](int sub, int i) -> uint4 { return *(const uint4*)(Wg + (int64_t)(f0 + sr + 32 * i) * a.K + sub * BK + sc * EPC); },
        MODE == 0 ? (KSPLIT ? (wave & 1) * 32 : wave * 32) : 0, MODE == 0 || wave == 0,
        [&]() {                                           // (under the first two steps' loads)
            sm_finalize_stats<256>(a.bn_in, s_in, t_in, blockIdx.x == 0);
            stamp(1);
        }
#ifdef SM_STAMP
        , MODE == 0 ? a.out_acc + 1110 : nullptr
#endif
        );
    stamp(2);
    // accumulator register g: feature (wave*32 +) (g&3) + 8*(g>>2) + 4*h of sample row r
    if constexpr (MODE == 1) {
        if (wave == 0) {
            float* Cg = (float*)a.C;
            const int64_t m = m0 + r;
#pragma unroll
            for (int q = 0; q < 2; ++q) {                 // features 8q + 4h .. +3 < 16
                const int f = 8 * q + 4 * h;
                if (m < a.N) *(float4*)(Cg + m * 16 + f) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        return;
    } else {
        constexpr int C_PITCH = SmNT<T, BN>::C_PITCH, F_PITCH = SmNT<T, BN>::F_PITCH, F_BYTES = SmNT<T, BN>::F_BYTES;
        if constexpr (KSPLIT) {
            // raw f32 partial sums of this wave's k half into image (wave >> 1); bias, ReLU and the statistics when the halves meet below
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fl = (wave & 1) * 32 + 8 * q + 4 * h;
                *(float4*)(smem + (wave >> 1) * F_BYTES + r * F_PITCH + fl * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fl = wave * 32 + 8 * q + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * q + e] + a.bias[f0 + fl + e], 0.f);
                unsigned char* dst = smem + r * C_PITCH + fl * (int)sizeof(T);
                if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                else *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        sm_store_tile<T, BN, KSPLIT>(smem, a.C, 512, m0, a.N, f0, nullptr, tile_m, 512, a.out_acc, SM_ACT_SHIFT, [&](float* v, int64_t, int f, float* s1, float* s2) {
            if constexpr (KSPLIT) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = D::round(fmaxf(v[e] + a.bias[f + e], 0.f));
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
        });
        stamp(3);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward of one fc layer L (PROJ = false) or of the projection (PROJ = true), one launch, two roles.
// fc layer: both roles form  A'[m][f] = dL/d(pre-activation_L) = [r_L > 0](ca g + cb r_L + cz)  while staging, from the incoming gradient
// g = Gin (masked dL/d(BN_L output)), the saved activation r_L and coefficients that every workgroup derives itself from the producer's
// partial sums (sum g, sum g r_L per row tile) and the layer's stored statistics.  Projection: A' = dz as the head kernel stored it
// ([N][64] T, 16 live columns, zero padded): no transform.
//   role 0 (blocks 0 .. n_dgrad-1): data gradient  Gout[m][k] = mask_{L-1}[m][k] / (1-p) * sum_f A'[m][f] Wt[k][f]  (Wt = W^T in T, [K][KC]),
//          partial sums (sum Gout, sum Gout r_{L-1}) per row tile -> the next launch's coefficients (fc1: the conv tail's);
//   role 1: weight gradient  dW[f][k] = sum_m A'[m][f] u_{L-1}[m][k]  per 64 x 64 tile and row split, u_{L-1} = dropout(BN(r_{L-1})) formed
//          while staging from the stored statistics.  One split (N <= 256): written straight into the gradient buffer (fc1: k' -> k =
//          c*12 + w); more: into the split's slab, summed by sm_reduce_grads_kernel at the end of the backward pass (nothing else reads a
//          weight gradient before the optimiser).  The first tile column also sums the bias gradient  db[f] = sum_m A'[m][f].
// ---------------------------------------------------------------------------------------------------------------------------
struct SmBwdArgs {
    const void* Gin;         // [N][KC] T
    const void* R;           // [N][512] T saved activation of layer L (fc layers)
    const long long* gsum;   // [2][512] fixed-point totals of (g, g r_L) over the batch, accumulated by the producer launch
    const float* stats;      // [4][512] of layer L (written by the forward pass)
    float* dgamma;           // [512]
    float* dbeta;
    const void* Wt;          // [K][KC] T = W^T
    const void* Rp;          // [N][K] T saved activation of layer L-1
    const float* stats_p;    // [4][smod] of layer L-1
    void* Gout;              // [N][K] T
    float* out_partials;     // [tiles_m][2][K] (fc1: the conv tail's finalize launch reads rows) or nullptr
    long long* out_acc;      // [2][K] fixed-point totals for the next launch (zeroed at the start of the step) or nullptr
    float* dW;               // [P][K] f32 (split 0 of `splits` slabs `slab_stride` floats apart when splits > 1)
    float* db;               // [P]       (likewise)
    int64_t slab_stride;
    int64_t rows_per_split;  // multiple of 64
    int splits;
    int p_valid;             // rows of dW that exist (512; projection: 16)
    int64_t N;
    int K, smod, wmode;      // wmode 1: fc1 (dW column k = (k' & 63) * 12 + (k' >> 6))
    int n_dgrad;
    uint32_t dp_thresh, dp_key;       // dropout of layer L-1's output (0: none)
    const uint32_t* dp_salt;
    float dp_inv_keep;
};

template <typename T>
__device__ __forceinline__ uint4 sm_bnrelu_bwd_chunk(const uint4& gq, const uint4& rq, const float* ca, const float* cb, const float* cz, int f) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    float g[EPC], rv[EPC];
    D::unpack(gq, g);
    D::unpack(rq, rv);
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) {
        const float4 a = *(const float4*)(ca + f + 4 * q), b = *(const float4*)(cb + f + 4 * q), z = *(const float4*)(cz + f + 4 * q);
        g[4 * q] = rv[4 * q] > 0.f ? fmaf(a.x, g[4 * q], fmaf(b.x, rv[4 * q], z.x)) : 0.f;
        g[4 * q + 1] = rv[4 * q + 1] > 0.f ? fmaf(a.y, g[4 * q + 1], fmaf(b.y, rv[4 * q + 1], z.y)) : 0.f;
        g[4 * q + 2] = rv[4 * q + 2] > 0.f ? fmaf(a.z, g[4 * q + 2], fmaf(b.z, rv[4 * q + 2], z.z)) : 0.f;
        g[4 * q + 3] = rv[4 * q + 3] > 0.f ? fmaf(a.w, g[4 * q + 3], fmaf(b.w, rv[4 * q + 3], z.w)) : 0.f;
    }
    return D::pack(g);
}

template <typename T, bool PROJ, bool KS = false>
__global__ __launch_bounds__(256) void sm_fc_bwd_kernel(SmBwdArgs a) {
    using D = DT<T>;
    constexpr int EPC = D::EPC, BK = D::BK, BM = SM_BM, BN = SmTile<KS>::BN;
    constexpr bool KSPLIT = KS;
    constexpr int KC = PROJ ? 64 : 512;                    // contraction of the data gradient = width of Gin
    constexpr int LDS_BYTES = SmNT<T, BN>::BYTES > SmTN<T>::BYTES ? SmNT<T, BN>::BYTES : SmTN<T>::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    __shared__ __attribute__((aligned(16))) float ca[PROJ ? 4 : 512], cb[PROJ ? 4 : 512], cz[PROJ ? 4 : 512];
    __shared__ __attribute__((aligned(16))) float s_p[512], t_p[512];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // coefficients of layer L and the scale / shift of layer L-1: run as the loops' `mid()`, under the first operand loads
    auto prologue = [&]() {
        if constexpr (!PROJ) {
            // coefficients of layer L (bn_bwd_finalize_kernel's arithmetic, the partial rows folded in a fixed order)
            for (int c = tid; c < 512; c += 256) {
                const double s1 = sm_acc_get(a.gsum + c, SM_GRAD_SHIFT), s2 = sm_acc_get(a.gsum + 512 + c, SM_GRAD_SHIFT);
                const double mean = a.stats[c], invstd = a.stats[512 + c], scl = a.stats[1024 + c];
                const double dot = (s2 - mean * s1) * invstd;
                const double c1 = s1 / (double)a.N, c2 = dot / (double)a.N;
                ca[c] = (float)scl;
                cb[c] = (float)(-scl * invstd * c2);
                cz[c] = (float)(-scl * (c1 - mean * invstd * c2));
                if (blockIdx.x == 0) { a.dgamma[c] = (float)dot; a.dbeta[c] = (float)s1; }
            }
        }
        for (int c = tid; c < a.smod; c += 256) { s_p[c] = a.stats_p[2 * a.smod + c]; t_p[c] = a.stats_p[3 * a.smod + c]; }
        __syncthreads();
    };
    const T* __restrict__ Gg = (const T*)a.Gin;
    const T* __restrict__ Rg = (const T*)a.R;
    const uint32_t key = a.dp_thresh != 0 ? (a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key) : 0u;
    struct RawA { uint4 g, r; };
    auto a_load = [&](int64_t m, int f) -> RawA {          // the raw 16-byte chunks behind A' at row m, column f
        RawA x;
        x.g = *(const uint4*)(Gg + m * KC + f);
        if constexpr (PROJ) x.r = make_uint4(0, 0, 0, 0);
        else x.r = *(const uint4*)(Rg + m * KC + f);
        return x;
    };
    auto a_xform = [&](const RawA& x, int f) -> uint4 {
        if constexpr (PROJ) return x.g;
        else return sm_bnrelu_bwd_chunk<T>(x.g, x.r, ca, cb, cz, f);
    };

    if ((int)blockIdx.x < a.n_dgrad) {
        // ---------------- role 0: data gradient ----------------------------------------------------------------------------
        const int tiles_k = a.K / BN;
        const int tile_k = blockIdx.x % tiles_k;
        const int64_t tile_m = blockIdx.x / tiles_k;
        const int64_t m0 = tile_m * BM;
        const int k0o = tile_k * BN;                       // output columns
        const T* __restrict__ Wg = (const T*)a.Wt;
        const int sc = tid & 7, sr = tid >> 3;
        const int64_t m_st = m0 + sr, mc_st = m_st < a.N ? m_st : a.N - 1;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        sm_nt_loop<T, BN, RawA, KSPLIT>(
            acc, smem, KC / BK,
            [&](int sub) -> RawA { return a_load(mc_st, sub * BK + sc * EPC); },
            [&](const RawA& raw, int sub) -> uint4 {
                const uint4 v = a_xform(raw, sub * BK + sc * EPC);
                return m_st < a.N ? v : make_uint4(0, 0, 0, 0);
            },
            [&](int sub, int i) -> uint4 { return *(const uint4*)(Wg + (int64_t)(k0o + sr + 32 * i) * KC + sub * BK + sc * EPC); },
            KSPLIT ? (wave & 1) * 32 : wave * 32, true, prologue);
        constexpr int C_PITCH = SmNT<T, BN>::C_PITCH, F_PITCH = SmNT<T, BN>::F_PITCH, F_BYTES = SmNT<T, BN>::F_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (KSPLIT) {                         // raw f32 partial sums, image = k half
                const int fl = (wave & 1) * 32 + 8 * q + 4 * h;
                *(float4*)(smem + (wave >> 1) * F_BYTES + r * F_PITCH + fl * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            } else {
                const int fl = wave * 32 + 8 * q + 4 * h;
                unsigned char* dst = smem + r * C_PITCH + fl * (int)sizeof(T);
                if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2(pack2bf(acc[4 * q], acc[4 * q + 1]), pack2bf(acc[4 * q + 2], acc[4 * q + 3]));
                else *(float4*)dst = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        __syncthreads();
        const T* Rpg = (const T*)a.Rp;
        sm_store_tile<T, BN, KSPLIT>(smem, a.Gout, a.K, m0, a.N, k0o, a.out_partials, tile_m, a.K, a.out_acc, SM_GRAD_SHIFT, [&](float* v, int64_t m, int k, float* s1, float* s2) {
            float rv[EPC];
            D::unpack(*(const uint4*)(Rpg + m * a.K + k), rv);
            if (a.dp_thresh != 0) {
#pragma unroll
                for (int e = 0; e < EPC; e += 2) {
                    const uint32_t pr = dropout_pair(key, (uint32_t)m, (uint32_t)a.K, (uint32_t)(k + e));
                    v[e] *= dropout_scale(pr, 0, a.dp_thresh, a.dp_inv_keep);
                    v[e + 1] *= dropout_scale(pr, 1, a.dp_thresh, a.dp_inv_keep);
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float g = D::round(v[e]);
                s1[e] += g;
                s2[e] = fmaf(g, rv[e], s2[e]);
            }
        });
        return;
    }
    // ---------------- role 1: weight gradient (64 x 64 tile of one row split) + bias gradient ------------------------------------
    {
        int j = blockIdx.x - a.n_dgrad;
        const int tiles_q = a.K / 64, tiles_p = PROJ ? 1 : 8;
        const int split = j / (tiles_p * tiles_q);
        j -= split * tiles_p * tiles_q;
        const int tp = j / tiles_q, tq = j % tiles_q;
        const int p0 = tp * 64, q0 = tq * 64;
        const int wp = wave >> 1, wq = wave & 1;
        const T* __restrict__ Yg = (const T*)a.Rp;
        const int64_t m_begin = (int64_t)split * a.rows_per_split;
        const int64_t m_end = m_begin + a.rows_per_split < a.N ? m_begin + a.rows_per_split : a.N;
        constexpr int CPRX = 64 / EPC, RS = 256 / CPRX;
        float bsum[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) bsum[e] = 0.f;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        sm_tn_loop<T, RawA>(
            acc, smem, m_begin, m_end,
            [&](int64_t m, int ch) -> RawA { return a_load(m, p0 + ch * EPC); },
            [&](const RawA& raw, int64_t, int ch, bool ok) -> uint4 {
                const uint4 x = a_xform(raw, p0 + ch * EPC);
                if (!PROJ && tq == 0) {
                    float xv[EPC];
                    D::unpack(x, xv);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) bsum[e] += ok ? xv[e] : 0.f;
                }
                return x;
            },
            [&](int64_t m, int ch) -> uint4 { return *(const uint4*)(Yg + m * a.K + q0 + ch * EPC); },
            [&](const uint4& raw, int64_t m, int ch) -> uint4 {
                return sm_bn_drop_chunk<T>(raw, s_p, t_p, q0 + ch * EPC, a.smod - 1, key, (uint32_t)m, (uint32_t)a.K, a.dp_thresh, a.dp_inv_keep);
            },
            wp, wq, prologue);
        // accumulator register g: row p = wp*32 + (g&3) + 8*(g>>2) + 4*h, column q = wq*32 + r
        float* dW = a.dW + (int64_t)split * a.slab_stride;
        {
            const int qk = q0 + wq * 32 + r;
            const int kdst = a.wmode == 1 ? (qk & 63) * 12 + (qk >> 6) : qk;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int p = p0 + wp * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                if (p < a.p_valid) dW[(int64_t)p * a.K + kdst] = acc[g];
            }
        }
        if (!PROJ && tq == 0) {                             // bias gradient of features p0 .. p0+63: sum over the row slots
            float* red = (float*)smem;                      // (the staging buffers are dead: the loop ended with a barrier)
            const int rs = tid / CPRX, ch = tid % CPRX;
#pragma unroll
            for (int e = 0; e < EPC; ++e) red[rs * 64 + ch * EPC + e] = bsum[e];
            __syncthreads();
            if (tid < 64) {
                float s = 0.f;
                for (int q = 0; q < RS; ++q) s += red[q * 64 + tid];
                (a.db + (int64_t)split * a.slab_stride)[p0 + tid] = s;
            }
        }
    }
}

// weight-gradient slabs of a step with more than one row split -> the gradient buffers: jobs of {slab 0, numel, destination}
struct SmReduceJob { const float* slab; float* dst; int numel; };
struct SmReduceBatch { SmReduceJob job[16]; int njobs, splits; int64_t slab_stride; };
__global__ __launch_bounds__(256) void sm_reduce_grads_kernel(SmReduceBatch b) {
    const SmReduceJob jb = b.job[blockIdx.y];
    // 16-byte loads and stores where the job allows (every tensor of the model does; round 4, third part: one element per thread and load
    // was 12.6 us of a 260 us step); per element the same additions in the same order
    if ((((uintptr_t)jb.slab | (uintptr_t)jb.dst) & 15) == 0 && ((jb.numel | b.slab_stride) & 3) == 0) {
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < jb.numel; i += gridDim.x * 1024) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < b.splits; ++q) {
                const float4 v = *(const float4*)(jb.slab + (int64_t)q * b.slab_stride + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *(float4*)(jb.dst + i) = s;
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.numel; i += gridDim.x * 256) {
        float s = 0.f;
        for (int q = 0; q < b.splits; ++q) s += jb.slab[(int64_t)q * b.slab_stride + i];
        jb.dst[i] = s;
    }
}

// this step's copies of the fc weights in the compute dtype, one launch: job j = {W f32 [F][K], out T [rows_out][K]}; mode 1 (fc1):
// columns in the internal order k' = w*64 + c of the reference's k = c*12 + w.  grid (512, jobs).
// ONE weight-preparation launch per step (grid (SM_PREP_GX, njobs + 1 + ntrans + 1)): the T copies of the fc weights and the projection
// (forward operands, fc1 in the internal column order), the step's BatchNorm accumulators zeroed, the transposed copies the backward
// pass's data gradients read (they depend on the parameters only, so they can be made before the forward pass needs nothing of them),
// and conv2's two tap-major weight images.  (Three launches until round 3: 20 us of a 300 us step.)
struct SmCopyJob { const float* W; void* out; int F, K, rows_out, mode; };
struct SmPrepBatch {
    SmCopyJob job[8]; int njobs; long long* zero; int nzero;
    TransposeJob tr[8]; int ntrans;
    const float* conv2_w; void* wc2_f; void* wc2_d;
};
// grid (128, jobs): a copy job's workgroups walk its rows with stride 128, a transpose job's its
// 64 x 64 tiles (at most 96 per job: one each), conv2's images 48 workgroups' worth of elements.  (The first merged form launched
// 512 x 18 workgroups, most of which returned at once: 13.6 us, more than the three launches' longest.)
#define SM_PREP_GX 128
template <typename T>
__global__ __launch_bounds__(256) void sm_prep_kernel(SmPrepBatch b) {
    using D = DT<T>;
    __shared__ float tile[64][65];
    const int y = blockIdx.y;
    if (y < b.njobs) {
        const SmCopyJob jb = b.job[y];
        // one 16-byte chunk of the output (8 bf16 / 4 f32) per thread and pass (round 4, third part: one element per thread -- 2-byte stores
        // in bf16 -- was most of this launch's 10 us); mode 1 gathers its chunk's elements from the reference's column order
        constexpr int EPC = D::EPC;
        const int cpr = jb.K / EPC;                        // chunks per row (K is a multiple of 64)
        const int total = (((uintptr_t)jb.W | (uintptr_t)jb.out) & 15) == 0 ? jb.rows_out * cpr : 0;      // (a caller's unaligned tensors: element by element below)
        if (total == 0)
            for (int j = blockIdx.x; j < jb.rows_out; j += SM_PREP_GX)
                for (int kp = threadIdx.x; kp < jb.K; kp += 256) {
                    const int k = jb.mode == 1 ? (kp & 63) * 12 + (kp >> 6) : kp;
                    D::store((T*)jb.out + (int64_t)j * jb.K + kp, j < jb.F ? jb.W[(int64_t)j * jb.K + k] : 0.f);
                }
        for (int c = blockIdx.x * 256 + threadIdx.x; c < total; c += SM_PREP_GX * 256) {
            const int j = c / cpr, kp = (c % cpr) * EPC;
            float v[EPC];
            if (j >= jb.F) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = 0.f;
            } else if (jb.mode == 1) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = jb.W[(int64_t)j * jb.K + ((kp + e) & 63) * 12 + ((kp + e) >> 6)];
            } else {
#pragma unroll
                for (int q = 0; q < EPC / 4; ++q) {
                    const float4 w = *(const float4*)(jb.W + (int64_t)j * jb.K + kp + 4 * q);
                    v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
                }
            }
            *(uint4*)((T*)jb.out + (int64_t)j * jb.K + kp) = D::pack(v);
        }
    } else if (y == b.njobs) {                              // the step's BatchNorm accumulators (forward and backward) start at zero
        for (int i = blockIdx.x * 256 + threadIdx.x; i < b.nzero; i += SM_PREP_GX * 256) b.zero[i] = 0;
    } else if (y <= b.njobs + b.ntrans) {
        transpose_w_job<T>(b.tr[y - b.njobs - 1], blockIdx.x, SM_PREP_GX, tile);
    } else {
        if (blockIdx.x < 48) prep_conv2_body<T>(b.conv2_w, (T*)b.wc2_f, (T*)b.wc2_d, blockIdx.x * 256 + threadIdx.x, 48 * 256);
    }
}
