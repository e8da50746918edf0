// Shared device helpers for the gfx950 (CDNA4, MI355X) kernels of the contrastive
// sEMG step.  wave = 64 lanes; MFMA tiles are 32x32; LDS tiles are 128-byte rows
// of eight 16-byte chunks.  No portability layer: this is gfx950-only code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

typedef uint16_t bf16_t;                                            // bf16 storage
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CP_WAVE 64

// f(integral_constant<int, 0>{}) .. f(integral_constant<int, N - 1>{}), in that order: a loop whose index is a constant expression
// inside the body (template arguments, register-array indices, the paced epilogues' micro-op tables).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The product library has NO process-wide switches: everything a call depends on travels in its cp_config (include/cpnative.h:
// options, tile_schedule, the synchronised-BatchNorm hook, the gradient tap), so two engines in one process cannot see each other's
// settings.  Only the tools-only build (-DCP_VARIANTS: make -C csrc variants -> build/libcpnative_variants.so), which also carries
// the kernels that were measured and superseded (tools/variants/*.cuh), keeps a global: one switch per superseded kernel, seeded
// ONCE from $CPNATIVE_<NAME> when that library is loaded (tools/ab_env.sh).
#ifdef CP_VARIANTS
struct CpVariantOptions {
    int no_ws = 0, no_wsk = 0, no_wsd = 0, no_wsd_st = 0, staged_r_epilogue = 0, ws32 = 0, wsd32 = 0, tn_w4 = 0, tn16 = 0,
        materialize_u8 = 0, no_proj_fused = 0;
};
static CpVariantOptions g_var;
#endif

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                   // RNE, NaN stays NaN (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
// two f32 -> packed bf16 (one v_cvt_pk_bf16_f32); RELU: max(.,0) on the packed pair as signed 16-bit
// integers (one v_pk_max_i16): a negative bf16, -0 included, is a negative int16, everything else keeps its bits
template <bool RELU>
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    s16x2_t s = __builtin_bit_cast(s16x2_t, __builtin_convertvector(v, bf16x2_t));
    if constexpr (RELU) {
        const s16x2_t z = {0, 0};
        s = __builtin_elementwise_max(s, z);
    }
    return __builtin_bit_cast(uint32_t, s);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return cvt_pk_bf16<false>(lo, hi); }

// ---- per-dtype traits: a "chunk" is 16 bytes -----------------------------------
template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int EPC = 4;                                   // elements per chunk
    static constexpr int BK = 32;                                   // k per 128-byte LDS row
    static constexpr int KSTEP = 8;                                 // k consumed per (h=0,1) chunk pair
    __device__ static __forceinline__ void unpack(const uint4& c, float* o) {
        o[0] = __uint_as_float(c.x); o[1] = __uint_as_float(c.y);
        o[2] = __uint_as_float(c.z); o[3] = __uint_as_float(c.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* v) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
    __device__ static __forceinline__ float round(float v) { return v; }
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int BK = 64;
    static constexpr int KSTEP = 16;
    __device__ static __forceinline__ void unpack(const uint4& c, float* o) {
        o[0] = __uint_as_float(c.x << 16); o[1] = __uint_as_float(c.x & 0xFFFF0000u);
        o[2] = __uint_as_float(c.y << 16); o[3] = __uint_as_float(c.y & 0xFFFF0000u);
        o[4] = __uint_as_float(c.z << 16); o[5] = __uint_as_float(c.z & 0xFFFF0000u);
        o[6] = __uint_as_float(c.w << 16); o[7] = __uint_as_float(c.w & 0xFFFF0000u);
    }
    __device__ static __forceinline__ uint4 pack(const float* v) {
        return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- OCP e4m3 storage (CP_FP8; csrc/fp8.cuh): tensors carry one power-of-two scale, kept as an exponent in device memory ----
__device__ __forceinline__ float f8_exp2i(int e) { return __int_as_float((127 + e) << 23); }      // 2^e, |e| < 127
// 4 e4m3 bytes -> f32 x4 (stored units)
__device__ __forceinline__ void f8_unpack4(uint32_t w, float* o) {
    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
}
// 8 e4m3 bytes times d -> one 16-byte chunk of bf16
__device__ __forceinline__ uint4 f8_chunk_to_bf16(const uint2& raw, float d) {
    float v[8];
    f8_unpack4(raw.x, v);
    f8_unpack4(raw.y, v + 4);
    return make_uint4(pack2bf(v[0] * d, v[1] * d), pack2bf(v[2] * d, v[3] * d), pack2bf(v[4] * d, v[5] * d), pack2bf(v[6] * d, v[7] * d));
}

// 8 e5m2 bytes times d -> one 16-byte chunk of bf16 (exact: two mantissa bits, power-of-two scale)
__device__ __forceinline__ uint4 f8_chunk5_to_bf16(const uint2& raw, float d) {
    const auto a = __builtin_amdgcn_cvt_pk_f32_bf8((int)raw.x, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)raw.x, true);
    const auto c = __builtin_amdgcn_cvt_pk_f32_bf8((int)raw.y, false), e = __builtin_amdgcn_cvt_pk_f32_bf8((int)raw.y, true);
    return make_uint4(pack2bf(a[0] * d, a[1] * d), pack2bf(b[0] * d, b[1] * d), pack2bf(c[0] * d, c[1] * d), pack2bf(e[0] * d, e[1] * d));
}

// ---- MFMA on one 16-byte chunk pair ---------------------------------------------
// a: chunk of the MFMA "A" operand row (lane&31), k-range selected by lane>>5
// b: chunk of the MFMA "B" operand column (lane&31), same k-range.
// bf16: one v_mfma_f32_32x32x16_bf16 (k = 8h..8h+7).  f32: four
// v_mfma_f32_32x32x2_f32, element e pairs k = 4h+e of both operands (exact f32 fma chain).
template <typename T> __device__ __forceinline__ void mma_chunk(const uint4& a, const uint4& b, f32x16& acc);
template <> __device__ __forceinline__ void mma_chunk<bf16_t>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<float>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
// accumulator register g of a 32x32 tile holds D[row = (g&3) + 8*(g>>2) + 4*(lane>>5)][col = lane&31]

// ---- 128-byte-row LDS tile, XOR-swizzled so 16 rows x one chunk column hit 16 slots ----
__device__ __forceinline__ int lds_tile_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- counter-based dropout mask (two 16-bit draws per hash) -----------------------
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// Four 16-bit draws for elements (row, col .. col+3) of an [rows x ld] tensor (ld, col multiples of 4) from ONE hash chain on the
// full-rate 24-bit multiply (v_mul_u32_u24): fold, multiply, key, fold, then two multiplies of the same word by different
// constants give 2 x 32 bits.  Round 3: the first form (two 32-bit multiplies per PAIR -- v_mul_lo_u32 issues at quarter rate)
// made the 8-bit dropout pass and the data-gradient epilogue behind a dropout VALU-bound (SQ_ACTIVE_INST_VALU 2.4x the matrix
// pipe's busy cycles); this one is 18 full-rate instructions per four elements.
// Round 4: ALL 32 bits of the quad index stay live.  A 24-bit multiply reads the low 24 bits of its operand, so round 3's chain kept
// 24 bits of state and tensors of more than 2^24 quads (131,072 rows x 512: the bench's 167,936) re-used whole rows of masks.  Now the
// index's top byte enters through its own multiply, and the second output word is drawn from another 24-bit window of the state.
// tools/dropout_hash_check.py (numpy, bit for bit; a CPU test runs it): no two rows of a 167,936 x 512 or a 335,872 x 512 tensor share a
// mask (round 3's form: 36,864 / 204,800 rows), drop rates 0.0634-0.0636 for p = 0.0635 in all four draws, cross-draw, lag-1, next-row,
// 2^22 / 2^24 / 2^25-quad-lag and key-bit-flip mask correlations all below 0.004.  Limit: row * ld < 2^32 (api.hip checks n_windows).
__device__ __forceinline__ uint2 dropout_quad(uint32_t key, uint32_t row, uint32_t ld, uint32_t col) {
    uint32_t x = ((__umul24(row, ld) + col) >> 2) ^ key;        // (rows and ld below 2^24, rows * ld below 2^32)
    const uint32_t hi = x >> 24;
    x ^= x >> 15;
    x = __umul24(x, 0xB5297Bu) ^ __umul24(hi, 0x9E3779u);
    x ^= (key >> 16) | (key << 16);
    x ^= x >> 13;
    uint32_t a = __umul24(x, 0x8DA6B5u), b = __umul24(x ^ (x >> 11), 0x3C6EF3u);
    a ^= a >> 16;
    b ^= b >> 16;
    return make_uint2(a, b);
}
// draws for col, col+1 (col even): the half of the quad that holds them
__device__ __forceinline__ uint32_t dropout_pair(uint32_t key, uint32_t row, uint32_t ld, uint32_t col) {
    const uint2 q = dropout_quad(key, row, ld, col & ~3u);
    return (col & 2u) ? q.y : q.x;
}
__device__ __forceinline__ float dropout_scale(uint32_t pair, int odd, uint32_t thresh, float inv_keep) {
    uint32_t d = odd ? (pair >> 16) : (pair & 0xFFFFu);
    return d >= thresh ? inv_keep : 0.0f;
}

// one 16-byte chunk of u = dropout(BatchNorm(r)) from the saved activation r, exactly as bn_dropout_apply_kernel writes it
// (column f.. of row `row` of a [.][C] tensor): consumers that can afford the arithmetic read r and never need u in memory
template <typename T>
__device__ __forceinline__ uint4 bn_drop_chunk(const uint4& in, const float* __restrict__ scale, const float* __restrict__ shift, int f,
                                               uint32_t key, uint32_t row, uint32_t C, uint32_t thresh, float inv_keep) {
    using D = DT<T>;
    float v[D::EPC];
    D::unpack(in, v);
#pragma unroll
    for (int e = 0; e < D::EPC; e += 2) {
        const uint32_t pr = dropout_pair(key, row, C, (uint32_t)(f + e));
        const f32x2_t keep = {dropout_scale(pr, 0, thresh, inv_keep), dropout_scale(pr, 1, thresh, inv_keep)};
        const f32x2_t y = __builtin_elementwise_fma((f32x2_t){v[e], v[e + 1]}, (f32x2_t){scale[f + e], scale[f + e + 1]},
                                                    (f32x2_t){shift[f + e], shift[f + e + 1]}) * keep;
        v[e] = y.x;
        v[e + 1] = y.y;
    }
    return D::pack(v);
}

// ---- BatchNorm totals as fixed-point integers (the small-batch path, small.cuh) --------------------------------------------
// fixed-point scales of the accumulators.  Activations: 2^-24 steps, legitimate totals below 2^30 (2,624 windows x 12 positions of
// squares up to 3 x 10^4); gradients (small numbers: the loss carries 1 / (2 N)): 2^-40 steps, totals below 2^14.
// Round 4: a partial sum that is not finite, or too large for that range, cannot be added as an integer (NaN would convert
// to 0 and vanish from the statistics; a large value would wrap).  It adds the sentinel 2^56 instead: with at most 255 adds per total
// (the small-batch launches have <= 164 workgroups) k sentinels stay in [2^56, 2^64) and never wrap to a small number, legitimate
// sums stay below 2^54, so any total at or beyond 2^55 in magnitude reads back as NaN -- the statistics are poisoned, as the f32 sums
// of the large-batch path are.
#define SM_ACT_SHIFT 24
#define SM_GRAD_SHIFT 40
#define SM_ACC_SENTINEL (1ull << 56)
__device__ __forceinline__ void sm_acc_add(long long* acc, float s, int shift) {
    // s * 2^shift is an integer-valued double wherever the f32's last bit is worth >= 2^-shift; smaller bits round away (<= 2^-shift)
    const double d = (double)s * (double)(1ull << shift);
    const unsigned long long q = fabs(d) < 18014398509481984.0 /* 2^54 */ ? (unsigned long long)__double2ll_rn(d) : SM_ACC_SENTINEL;
    atomicAdd((unsigned long long*)acc, q);
}
__device__ __forceinline__ double sm_acc_get(const long long* acc, int shift) {
    const long long v = *acc;
    const unsigned long long mag = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    if (mag >= (1ull << 55)) return (double)__int_as_float(0x7fc00000);
    return (double)v / (double)(1ull << shift);
}

struct SmBN {
    const long long* acc;    // [2][C] totals (sum, sum of squares) as fixed point, or nullptr: `stats` is already final (conv2's BatchNorm)
    int C;
    double count;
    const float* gamma;
    const float* beta;
    float* stats;            // [4][C]
    float* running_mean;     // nullable
    float* running_var;
    int update_running;
    float momentum, eps;
};
// one channel of such a BatchNorm, for a thread that needs a few of them in registers (the conv kernels at small batches): the arithmetic
// of sm_finalize_stats (small.cuh) and bn_finalize_kernel; `writer` (one thread per channel in the whole grid) stores the statistics
// and updates the running ones.
__device__ __forceinline__ void sm_bn_channel(const SmBN& b, int c, bool writer, float& sc, float& sh) {
    const double s1 = sm_acc_get(b.acc + c, SM_ACT_SHIFT), s2 = sm_acc_get(b.acc + b.C + c, SM_ACT_SHIFT);
    const double mu = s1 / b.count;
    double vb = s2 / b.count - mu * mu;
    if (vb < 0) vb = 0;
    const float mean = (float)mu, var = (float)vb;
    const float invstd = 1.0f / sqrtf(var + b.eps);
    sc = b.gamma[c] * invstd;
    sh = b.beta[c] - mean * sc;
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

// ---- reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
