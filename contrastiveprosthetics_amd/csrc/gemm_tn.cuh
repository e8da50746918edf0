// "TN" GEMM of the weight gradients:  C[p][q] = sum_m X[m][p] * Y[m][q]
//   m = sample rows (huge, the reduction), p = output features of the layer (rows of
//   dW), q = input features.  Both operands are stored sample-major, so the MFMA
//   fragments (8 consecutive m for one column) are read from the row-major LDS tile
//   with the gfx950 transposed read ds_read_b64_tr_b16 (bf16) or strided ds_read_b32
//   (f32).  The sample axis is split over blockIdx.y; every block writes its f32
//   partial tile to its own slab (deterministic, no atomics); reduce_slabs_kernel
//   (elementwise.hip) sums the slabs and applies the BN-fold fix-up.
#pragma once
#include "common.cuh"

enum { YLOAD_PLAIN = 0, YLOAD_CONV = 1,
       YLOAD_BNDROP = 2,     // Y = dropout(BatchNorm(saved activation)) formed while staging (y_scale/y_shift + the dp_* fields; Q = row width)
       YLOAD_F8 = 3,         // Y is stored as e4m3 (ldy in bytes) with the scale 2^*y_exp: converted to T while staging (bf16 kernels only)
       YLOAD_BNDROP_F8 = 4 };// both

struct GemmTNArgs {
    const void* X;          // [M][ldx] T
    const void* Y;          // [M][ldy] T
    float* slabs;           // [S][P][Q]
    const float* y_scale;   // YLOAD_CONV: BN affine of the input channels (64) or nullptr
    const float* y_shift;
    const int* y_exp;       // YLOAD_F8 / YLOAD_BNDROP_F8: device word holding Y's scale exponent
    int64_t M;
    int64_t rows_per_split; // multiple of 32
    int ldx, ldy, P, Q;
    // YLOAD_BNDROP: the dropout of the forward pass (as GemmNTArgs)
    uint32_t dp_thresh, dp_key;
    const uint32_t* dp_salt;
    float dp_inv_keep;
};

template <typename T, int COLS> struct TNPitch {
    static constexpr int value = (sizeof(T) == 2) ? COLS * 2 + 64 : COLS * 4 + 16;
};

// fragment of a [32 rows(m)][COLS] tile: 16-byte chunk holding, for column col0 + (lane&31),
// the rows m_off + KSTEP/2*(lane>>5) + {0..EPC-1}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int PITCH>
__device__ __forceinline__ uint4 tn_frag_bf16(const unsigned char* tile, int m_off, int col0, int lane) {
    // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block, delivered column-major.
    // lane 4q+p of the group addresses row q, columns 4p..4p+3 (verified by tools/mfma_probe.hip).
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int row = m_off + 8 * (g >> 1) + q;
    const int col = col0 + 16 * (g & 1) + 4 * pp;
    const unsigned char* p = tile + row * PITCH + col * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * PITCH));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

template <int PITCH>
__device__ __forceinline__ uint4 tn_frag_f32(const unsigned char* tile, int m_off, int col0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const unsigned char* p = tile + (m_off + 4 * h) * PITCH + (col0 + r) * 4;
    return make_uint4(*(const uint32_t*)p, *(const uint32_t*)(p + PITCH), *(const uint32_t*)(p + 2 * PITCH),
                      *(const uint32_t*)(p + 3 * PITCH));
}

template <typename T, int BP, int BQ, int YLOAD>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs a) {
    using D = DT<T>;
    constexpr int EPC = D::EPC;
    constexpr int KSTEP = D::KSTEP;                 // rows of m per MFMA chunk pair (16 bf16 / 8 f32)
    constexpr int PX = TNPitch<T, BP>::value, PY = TNPitch<T, BQ>::value;
    constexpr int XB = 32 * PX, YB = 32 * PY;
    constexpr int WP_T = BP / 64, WQ_T = BQ / 64;
    constexpr int XCPR = BP / EPC, YCPR = BQ / EPC; // chunks per tile row
    constexpr int X_IT = (32 * XCPR + 255) / 256, Y_IT = (32 * YCPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (XB + YB)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int tiles_q = a.Q / BQ;
    const int tp = blockIdx.x / tiles_q, tq = blockIdx.x % tiles_q;
    const int p0 = tp * BP, q0 = tq * BQ;
    const int64_t mb = (int64_t)blockIdx.y * a.rows_per_split;
    int64_t me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int nsteps = (int)((me - mb + 31) / 32);

    const T* __restrict__ Xg = (const T*)a.X;
    const T* __restrict__ Yg = (const T*)a.Y;
    uint4 xreg[X_IT], yreg[Y_IT];
    float y_deq = 1.f;
    if constexpr (YLOAD == YLOAD_F8 || YLOAD == YLOAD_BNDROP_F8) y_deq = f8_exp2i(-*a.y_exp);

    auto load_tiles = [&](int step) {
        const int64_t ms = mb + (int64_t)step * 32;
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / XCPR, ch = idx % XCPR;
            const int64_t m = ms + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < 32 && m < me) v = *(const uint4*)(Xg + m * a.ldx + p0 + ch * EPC);
            xreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < Y_IT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / YCPR, ch = idx % YCPR;
            const int64_t m = ms + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if constexpr (YLOAD == YLOAD_PLAIN) {
                if (row < 32 && m < me) v = *(const uint4*)(Yg + m * a.ldy + q0 + ch * EPC);
            } else if constexpr (YLOAD == YLOAD_F8 || YLOAD == YLOAD_BNDROP_F8) {
                static_assert(sizeof(T) == 2 || (YLOAD != YLOAD_F8 && YLOAD != YLOAD_BNDROP_F8), "e4m3 operands feed the bf16 kernels");
                if (row < 32 && m < me) {
                    v = f8_chunk_to_bf16(*(const uint2*)((const uint8_t*)a.Y + m * a.ldy + q0 + ch * 8), y_deq);
                    if constexpr (YLOAD == YLOAD_BNDROP_F8)
                        v = bn_drop_chunk<T>(v, a.y_scale, a.y_shift, q0 + ch * EPC, a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key, (uint32_t)m,
                                             (uint32_t)a.Q, a.dp_thresh, a.dp_inv_keep);
                }
            } else if constexpr (YLOAD == YLOAD_BNDROP) {
                if (row < 32 && m < me)
                    v = bn_drop_chunk<T>(*(const uint4*)(Yg + m * a.ldy + q0 + ch * EPC), a.y_scale, a.y_shift, q0 + ch * EPC,
                                         a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key, (uint32_t)m, (uint32_t)a.Q, a.dp_thresh, a.dp_inv_keep);
            } else {
                // q tile index selects the tap; Y row = activation row m + tap - 1 inside the window
                const int tap = tq;
                const int w = (int)(m % 12) + tap - 1;
                if (row < 32 && m < me && w >= 0 && w < 12) {
                    v = *(const uint4*)(Yg + (m + tap - 1) * a.ldy + ch * EPC);
                    if (a.y_scale != nullptr) {
                        float x[EPC];
                        D::unpack(v, x);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) x[e] = fmaf(x[e], a.y_scale[ch * EPC + e], a.y_shift[ch * EPC + e]);
                        v = D::pack(x);
                    }
                }
            }
            yreg[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* Xs = smem + buf * (XB + YB);
        unsigned char* Ys = Xs + XB;
#pragma unroll
        for (int i = 0; i < X_IT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / XCPR, ch = idx % XCPR;
            if (row < 32) *(uint4*)(Xs + row * PX + ch * 16) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < Y_IT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / YCPR, ch = idx % YCPR;
            if (row < 32) *(uint4*)(Ys + row * PY + ch * 16) = yreg[i];
        }
    };

    f32x16 acc[WP_T][WQ_T];
#pragma unroll
    for (int i = 0; i < WP_T; ++i)
#pragma unroll
        for (int j = 0; j < WQ_T; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

    if (nsteps > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) load_tiles(step + 1);
        const unsigned char* Xs = smem + (step & 1) * (XB + YB);
        const unsigned char* Ys = Xs + XB;
#pragma unroll
        for (int ks = 0; ks < 32 / KSTEP; ++ks) {
            uint4 fx[WP_T], fy[WQ_T];
#pragma unroll
            for (int i = 0; i < WP_T; ++i) {
                const int c0 = (wp * WP_T + i) * 32;
                if constexpr (sizeof(T) == 2) fx[i] = tn_frag_bf16<PX>(Xs, ks * KSTEP, c0, lane);
                else fx[i] = tn_frag_f32<PX>(Xs, ks * KSTEP, c0, lane);
            }
#pragma unroll
            for (int j = 0; j < WQ_T; ++j) {
                const int c0 = (wq * WQ_T + j) * 32;
                if constexpr (sizeof(T) == 2) fy[j] = tn_frag_bf16<PY>(Ys, ks * KSTEP, c0, lane);
                else fy[j] = tn_frag_f32<PY>(Ys, ks * KSTEP, c0, lane);
            }
#pragma unroll
            for (int i = 0; i < WP_T; ++i)
#pragma unroll
                for (int j = 0; j < WQ_T; ++j) mma_chunk<T>(fx[i], fy[j], acc[i][j]);
        }
        if (step + 1 < nsteps) store_tiles((step + 1) & 1);
        __syncthreads();
    }

    float* slab = a.slabs + (int64_t)blockIdx.y * a.P * a.Q;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < WP_T; ++i)
#pragma unroll
        for (int j = 0; j < WQ_T; ++j) {
            const int q = q0 + (wq * WQ_T + j) * 32 + r;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int p = p0 + (wp * WP_T + i) * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                slab[(int64_t)p * a.Q + q] = acc[i][j][g];
            }
        }
}

template <typename T, int BP, int BQ, int YLOAD>
static inline hipError_t launch_gemm_tn(const GemmTNArgs& a, int splits, hipStream_t st) {
    dim3 grid((unsigned)((a.P / BP) * (a.Q / BQ)), (unsigned)splits);
    hipLaunchKernelGGL((gemm_tn_kernel<T, BP, BQ, YLOAD>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// The projection's weight gradient behind fc7's dropout, and fc7's BatchNorm-backward sums with it (round 4): ONE pass over the
// saved activation r8 where round 3 made two (gemm_tn_kernel<.., YLOAD_BNDROP> for dW, proj_dgrad_kernel<0> for the sums).
// With keep[m][f] the forward pass's dropout decision, u8 = keep (s r8 + t) / (1 - p) and g = keep (dz Wp) / (1 - p):
//     A[j][f] = sum_m dz[m][j] keep[m][f] r8[m][f]          B[j][f] = sum_m dz[m][j] keep[m][f]
//     dWp[j][f] = (s[f] A + t[f] B) / (1 - p)
//     sum_m g[m][f] = sum_j Wp[j][f] B[j][f] / (1 - p)      sum_m g[m][f] r8[m][f] = sum_j Wp[j][f] A[j][f] / (1 - p)
// so the kernel needs no BatchNorm arithmetic at all: the Y operand of A is r8 with the dropped elements' bits cleared (one AND per
// two elements), the Y operand of B is the constant 1.0 under the same mask, and one hash chain per four elements serves both.
// 32 rows per step, 128 features per workgroup (wave w: features 32w..32w+31 of both products), dz's 16 live columns in a 32-column
// X tile; slabs[split][32][512]: rows 0..15 = A, 16..31 = B (raw: proj_wgrad_finish_kernel applies 1 / (1 - p), s, t and, for an
// e4m3 r8, its scale).  F8: r8 is stored as e4m3 bytes (expanded exactly into the bf16 tile).
// ------------------------------------------------------------------------------------------------------------------------
struct ProjWgradArgs {
    const bf16_t* dz;       // [M][64], columns >= 16 zero (cp_head)
    const void* R;          // [M][512] bf16, or e4m3 bytes
    float* slabs;           // [S][32][512]
    int64_t M, rows_per_split;
    int splits;
    uint32_t dp_thresh, dp_key;
    const uint32_t* dp_salt;
};
// grid: PROJ_WGRAD_GRID(S) blocks.  Block -> (feature block fx of 4, split fy): the four feature blocks of one split are
// dispatched next to each other on ONE XCD (blocks b and b + 8 share an XCD under round-robin dispatch), so the split's dz rows --
// which all four read -- come from HBM once and from that XCD's L2 three times (a (4, S) grid spread them over four XCDs: 86 MB of
// fetches for a 21.5 MB tensor, PMC).
#define PROJ_WGRAD_GRID(S) (32 * (((S) + 7) / 8))

template <bool F8>
__global__ __launch_bounds__(256) void proj_wgrad_sums_kernel(ProjWgradArgs a) {
    const int fx = (blockIdx.x >> 3) & 3, fy = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);
    if (fy >= a.splits) return;
    constexpr int PX = TNPitch<bf16_t, 32>::value, PY = TNPitch<bf16_t, 128>::value;
    constexpr int XB = 32 * PX, YB = 32 * PY, BUF = XB + 2 * YB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = fx * 128;
    const int64_t mb = (int64_t)fy * a.rows_per_split;
    int64_t me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int nsteps = (int)((me - mb + 31) / 32);
    const uint32_t key = a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key;
    uint4 xreg, ya[2], yb[2];
    auto load_tiles = [&](int step) {
        const int64_t ms = mb + (int64_t)step * 32;
        {
            const int row = tid >> 2, ch = tid & 3;
            const int64_t m = ms + row;
            xreg = make_uint4(0, 0, 0, 0);
            if (tid < 128 && m < me) xreg = *(const uint4*)(a.dz + m * 64 + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, ch = idx & 15;
            const int64_t m = ms + row;
            uint4 v = make_uint4(0, 0, 0, 0), o = make_uint4(0, 0, 0, 0);
            if (m < me) {
                if constexpr (F8) v = f8_chunk_to_bf16(*(const uint2*)((const uint8_t*)a.R + m * 512 + q0 + ch * 8), 1.f);
                else v = *(const uint4*)((const bf16_t*)a.R + m * 512 + q0 + ch * 8);
                const uint2 d0 = dropout_quad(key, (uint32_t)m, 512u, (uint32_t)(q0 + ch * 8));
                const uint2 d1 = dropout_quad(key, (uint32_t)m, 512u, (uint32_t)(q0 + ch * 8 + 4));
                auto word_mask = [&](uint32_t dr) {
                    return ((dr & 0xFFFFu) >= a.dp_thresh ? 0xFFFFu : 0u) | ((dr >> 16) >= a.dp_thresh ? 0xFFFF0000u : 0u);
                };
                const uint32_t m0 = word_mask(d0.x), m1 = word_mask(d0.y), m2 = word_mask(d1.x), m3 = word_mask(d1.y);
                v = make_uint4(v.x & m0, v.y & m1, v.z & m2, v.w & m3);
                o = make_uint4(0x3F803F80u & m0, 0x3F803F80u & m1, 0x3F803F80u & m2, 0x3F803F80u & m3);
            }
            ya[i] = v;
            yb[i] = o;
        }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* Xs = smem + buf * BUF;
        unsigned char* As = Xs + XB;
        unsigned char* Bs = As + YB;
        if (tid < 128) *(uint4*)(Xs + (tid >> 2) * PX + (tid & 3) * 16) = xreg;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, ch = idx & 15;
            *(uint4*)(As + row * PY + ch * 16) = ya[i];
            *(uint4*)(Bs + row * PY + ch * 16) = yb[i];
        }
    };
    f32x16 accA, accB;
#pragma unroll
    for (int g = 0; g < 16; ++g) accA[g] = accB[g] = 0.f;
    if (nsteps > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) load_tiles(step + 1);
        const unsigned char* Xs = smem + (step & 1) * BUF;
        const unsigned char* As = Xs + XB;
        const unsigned char* Bs = As + YB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 fx = tn_frag_bf16<PX>(Xs, ks * 16, 0, lane);
            const uint4 fa = tn_frag_bf16<PY>(As, ks * 16, wave * 32, lane);
            const uint4 fb = tn_frag_bf16<PY>(Bs, ks * 16, wave * 32, lane);
            mma_chunk<bf16_t>(fx, fa, accA);
            mma_chunk<bf16_t>(fx, fb, accB);
        }
        if (step + 1 < nsteps) store_tiles((step + 1) & 1);
        __syncthreads();
    }
    // D[p = dz column][q = feature]: accumulator g holds p = (g & 3) + 8 (g >> 2) + 4 h -- the 16 live columns are g < 8
    float* slab = a.slabs + (int64_t)fy * 32 * 512;
    const int r = lane & 31, h = lane >> 5, q = q0 + wave * 32 + r;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int pcol = (g & 3) + 8 * (g >> 2) + 4 * h;
        slab[pcol * 512 + q] = accA[g];
        slab[(16 + pcol) * 512 + q] = accB[g];
    }
}

// slabs -> the projection's weight gradient (reference layout [16][512]) and FOUR partial rows [2][512] of fc7's BatchNorm-backward
// sums in bn_bwd_finalize_kernel's layout (row y sums dz columns 4y..4y+3).  grid (32, 4) blocks x 256 threads = 16 features x 4 dz
// columns x 4 slab quarters: a thread walks every fourth slab, eight loads of both products in flight, 64-byte segments per wave
// row (a fixed order of additions; 32 blocks of whole-S walks took 14 us).
// r_exp (e4m3 r8, else nullptr): its scale exponent.  Wp enters the sums rounded to bf16, as the data-gradient launch reads it.
#define PROJ_FINISH_ROWS 4
__global__ __launch_bounds__(256) void proj_wgrad_finish_kernel(const float* __restrict__ slabs, int S, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ Wp,
                                                                float inv_keep, const int* __restrict__ r_exp, float* __restrict__ dWp,
                                                                float* __restrict__ rows) {
    __shared__ float part[2][4][4][16];
    __shared__ float red[2][4][16];
    const int tid = threadIdx.x, fl = tid & 15, jl = (tid >> 4) & 3, sh = tid >> 6;
    const int f = blockIdx.x * 16 + fl, j = blockIdx.y * 4 + jl;
    const float* pa = slabs + j * 512 + f;
    const float* pb = pa + 16 * 512;
    const int64_t sstride = 32 * 512;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    int k = sh;
    for (; k + 28 < S; k += 32) {
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { va[u] = pa[(k + 4 * u) * sstride]; vb[u] = pb[(k + 4 * u) * sstride]; }
        a0 += (va[0] + va[2]) + (va[4] + va[6]);
        a1 += (va[1] + va[3]) + (va[5] + va[7]);
        b0 += (vb[0] + vb[2]) + (vb[4] + vb[6]);
        b1 += (vb[1] + vb[3]) + (vb[5] + vb[7]);
    }
    for (; k < S; k += 4) { a0 += pa[k * sstride]; b0 += pb[k * sstride]; }
    part[0][sh][jl][fl] = a0 + a1;
    part[1][sh][jl][fl] = b0 + b1;
    __syncthreads();
    if (sh == 0) {
        const float A = ((part[0][0][jl][fl] + part[0][1][jl][fl]) + (part[0][2][jl][fl] + part[0][3][jl][fl])) * inv_keep * (r_exp ? f8_exp2i(-*r_exp) : 1.f);
        const float B = ((part[1][0][jl][fl] + part[1][1][jl][fl]) + (part[1][2][jl][fl] + part[1][3][jl][fl])) * inv_keep;
        dWp[j * 512 + f] = fmaf(scale[f], A, shift[f] * B);
        const float w = bf2f(f2bf(Wp[j * 512 + f]));
        red[0][jl][fl] = w * B;
        red[1][jl][fl] = w * A;
    }
    __syncthreads();
    if (tid < 32) {
        const int which = tid >> 4, c = tid & 15;
        rows[((int64_t)blockIdx.y * 2 + which) * 512 + blockIdx.x * 16 + c] = (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
    }
}
