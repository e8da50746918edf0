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
