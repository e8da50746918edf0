// 8-bit path of the encoder (BASELINE config 4, CP_FP8): activations stored as OCP e4m3, the fc weights quantised to e4m3 with one
// power-of-two scale per output feature, products on the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4.
//
// Facts this file relies on, measured on an MI355X with tools/fp8_probe.hip (profiles/r03_fp8_probe.txt):
//   * the block-scaled MFMA with e4m3 x e4m3 (and e5m2 x e4m3) operands sustains 3.3-3.7 PFLOP/s on random data, the bf16 16x16x32
//     form 0.8 PFLOP/s in the same loop: under the chip's power cap the 8-bit form is worth 4x, not the 2x of its cycle count;
//   * a lane's E8M0 scale byte (op_sel picks one of the four bytes of the scale register) multiplies that lane's ROW when the four
//     lanes of a row carry the same byte; scales that differ between the k groups of a row do not follow the obvious model, so
//     only per-row (= per output feature) scales are used;
//   * which of its 32 bytes a lane contributes to which k does not matter for a product as long as both operands are loaded with
//     the same (lane >> 4, byte) -> k map, which they are here: both come as 32 consecutive bytes at k = 128*kb + 32*(lane >> 4);
//   * v_cvt_pk_fp8_f32 rounds to nearest even and does NOT saturate (465 -> NaN), v_cvt_pk_bf8_f32 overflows to infinity:
//     every conversion is clamped first (v_med3_f32);
//   * ds_read_b64_tr_b8: per 16 lanes a block of 8 rows x 16 byte-columns, lane 2q + p supplies the address of row q, columns
//     8p .. 8p+7, lane i receives column i of rows 0..7 in its bytes 0..7.
//
// Scaling.  Every stored 8-bit tensor t has ONE power-of-two scale 2^e[t] ("delayed scaling": chosen at the start of a step from
// the largest magnitude the tensor showed in the previous step; defaults on the first step), kept with the running maximum in a
// small device table at the start of the workspace (Fp8State), read by the kernels -- the host never sees it.  Powers of two
// make every scale exact: the input tensor's scale and the BatchNorm fold go into the weights BEFORE they are quantised, the
// weight row's own scale and the output tensor's scale go into the MFMA's scale operand, so an accumulator IS the output in
// stored units and the epilogue is clamp + convert.
//
// Reference anchor: the reference only gestures at reduced precision (code/train.py:6,37,56,97: `amp` imported, `autocast`
// commented out).  Parity of this path is "unpinned by construction": tests report its distance to the f32 CPU restatement of the reference and to the
// bf16 path, and gate on finiteness and on the loss.
#pragma once
#include "gemm_ws.cuh"

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

#define F8_E4M3_MAX 448.f
#define F8_E5M2_MAX 57344.f

// ---- the scale table -------------------------------------------------------------------------------------------------
#define F8_NT 64
enum {
    F8_T_ACT = 0,        // + l, l = 1..8: saved post-ReLU activation of layer l (conv2 output, fc1..fc7 outputs), e4m3
    F8_T_U = 9,          // + i, i = 0..2: dropout(BatchNorm(.)) of fc4..fc6 (the inputs of fc5..fc7), e4m3
    F8_T_GRAD = 16,      // + l: gradient with respect to the pre-activation of layer l, e5m2
};
struct Fp8State {
    int32_t e[F8_NT];        // scale of tensor t = 2^e[t]:  stored = value * 2^e
    uint32_t amax[F8_NT];    // running maximum of |stored-unit value| before clamping, as float bits (atomicMax on non-negative floats)
    uint32_t init;           // 0 in a fresh (zero-filled) workspace
    uint32_t pad[63];
};
static_assert(sizeof(Fp8State) == 768, "Fp8State layout");
#define F8_STATE_BYTES 1024

// (f8_exp2i, f8_unpack4: common.cuh)
// largest e with amax * 2^e <= target (target a power of two times 1.0 or 1.75)
__device__ __forceinline__ int f8_fit_exp(float amax, float target) {
    if (!(amax > 0.f)) return 0;
    int xa, xt;
    const float ma = frexpf(amax, &xa), mt = frexpf(target, &xt);         // m in [0.5, 1)
    int e = xt - xa;
    if (ma > mt) --e;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}

// Start of every forward pass: turn last step's maxima into this step's scales.  Activations aim at half the e4m3 range (the
// maximum is one step old), gradients at 2^10 of e5m2's 2^15.8 (its two mantissa bits lose nothing to headroom).
__global__ void fp8_update_scales_kernel(Fp8State* s, int64_t n_windows) {
    const int t = threadIdx.x;
    if (t >= F8_NT) return;
    const bool grad = t >= F8_T_GRAD;
    const uint32_t was_init = s->init;
    int e = s->e[t];
    if (!was_init) {
        // defaults: post-ReLU activations of BatchNorm-ed inputs are O(1) -> 2^4 (clips at 28); gradients carry the loss's 1 / (2 N)
        int lg = 0;
        while (((int64_t)1 << lg) < 2 * n_windows) ++lg;
        e = grad ? lg + 4 : 4;
    }
    const uint32_t ab = s->amax[t];
    if (ab != 0u) {
        const float true_amax = __uint_as_float(ab) * f8_exp2i(-e);
        e = f8_fit_exp(true_amax, grad ? 1024.f : 224.f);
    }
    s->e[t] = e;
    s->amax[t] = 0u;
    __syncthreads();
    if (t == 0) s->init = 1u;
}

// f32 x4 -> 4 e4m3 bytes (values already clamped to [-448, 448])
__device__ __forceinline__ uint32_t f8_pack4(float a, float b, float c, float d) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}
__device__ __forceinline__ float f8_clamp(float v) { return __builtin_amdgcn_fmed3f(v, -F8_E4M3_MAX, F8_E4M3_MAX); }
// MODE.FP16_OVFL = 1 for the rest of the wave: v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32 then SATURATE finite values beyond the format's range
// (1000 -> 448, 1e6 -> 57344; inf and NaN stay what they are) instead of returning NaN / inf, so the streaming passes and the data-gradient
// epilogues convert without a v_med3_f32 per element (tools/fp8_sat_probe.hip, round 4; a new wave starts from the kernel descriptor's mode).
__device__ __forceinline__ void f8_saturating_conversions() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }
// Running maximum of a tensor: one candidate per wave, and the atomic only where it would raise the word -- thousands of waves end at
// about the same time, and same-address atomics serialise in L2 (the first version, an unconditional atomicMax per wave, made the
// 8-bit dropout pass 3x slower than the bf16 one).  The word is read past L1 (agent scope); a stale read costs one needless atomic.
__device__ __forceinline__ void f8_atomic_amax(uint32_t* dst, float m) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        const uint32_t bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// fold the previous BatchNorm's affine AND the input tensor's scale into a Linear, quantise each output row to e4m3 with its own
// power-of-two scale:  y * 2^eo = sum_k Wq[j][k] * xq[k] * 2^(eo - ej - ei) + b'[j] * 2^eo,   Wq = e4m3(W[j][k] s[k] 2^ej)
//   out_w: [F][K] bytes (fc1: in the internal order k' = w*64 + c);  out_sc: [F] E8M0 bytes 127 + eo - ej - ei for the MFMA's
//   scale operand;  out_b: [F] f32 bias in OUTPUT units (the accumulators start at it).  One block per output row.
//   t_in / t_out: scale-table ids of the layer's input and output tensors (t_out < 0: eo = 0, f32 output).
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fold_linear8_row(const float* __restrict__ W, const float* __restrict__ b,
                                                 const float* __restrict__ s, const float* __restrict__ t,
                                                 uint8_t* __restrict__ out_w, uint8_t* __restrict__ out_sc,
                                                 float* __restrict__ out_b, int K, int mode, const Fp8State* __restrict__ st,
                                                 int t_in, int t_out, int j) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const int ei = st->e[t_in], eo = t_out >= 0 ? st->e[t_out] : 0;
    float wv[3], acc = 0.f, am = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int k = tid + 256 * q;
        wv[q] = 0.f;
        if (k < K) {
            const int ch = mode == 1 ? k / 12 : k;
            float w = W[(int64_t)j * K + k];
            if (s != nullptr) { acc = fmaf(w, t[ch], acc); w *= s[ch]; }
            wv[q] = w;
            am = fmaxf(am, fabsf(w));
        }
    }
    acc = wave_sum(acc);
    am = wave_max(am);
    if ((tid & 63) == 0) { red[0][tid >> 6] = acc; red[1][tid >> 6] = am; }
    __syncthreads();
    acc = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    am = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const int ej = f8_fit_exp(am, F8_E4M3_MAX);
    const float sc = f8_exp2i(ej);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int k = tid + 256 * q;
        if (k < K) {
            const int kp = mode == 1 ? (k % 12) * 64 + k / 12 : k;
            const int p = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(wv[q] * sc), 0.f, 0, false);
            out_w[(int64_t)j * K + kp] = (uint8_t)(p & 255);
        }
    }
    if (tid == 0) {
        int sb = 127 + eo - ej - ei;                 // (the input scale divides out here: x = xq 2^-ei)
        sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
        out_sc[j] = (uint8_t)sb;
        out_b[j] = ((b ? b[j] : 0.f) + acc) * f8_exp2i(eo);
    }
}
__global__ __launch_bounds__(256) void fold_linear8_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                           const float* __restrict__ s, const float* __restrict__ t,
                                                           uint8_t* __restrict__ out_w, uint8_t* __restrict__ out_sc,
                                                           float* __restrict__ out_b, int K, int mode, const Fp8State* __restrict__ st,
                                                           int t_in, int t_out) {
    fold_linear8_row(W, b, s, t, out_w, out_sc, out_b, K, mode, st, t_in, t_out, blockIdx.x);
}
// the seven folds of an evaluation pass with the running statistics in one launch (kernels_misc.cuh, fold_linear_batch_kernel)
struct Fold8Job { const float* W; const float* b; const float* s; const float* t; uint8_t* out_w; uint8_t* out_sc; float* out_b; int K, mode, t_in, t_out; };
struct Fold8Batch { Fold8Job job[7]; };
__global__ __launch_bounds__(256) void fold_linear8_batch_kernel(Fold8Batch fb, const Fp8State* __restrict__ st) {
    const Fold8Job& jb = fb.job[blockIdx.y];
    fold_linear8_row(jb.W, jb.b, jb.s, jb.t, jb.out_w, jb.out_sc, jb.out_b, jb.K, jb.mode, st, jb.t_in, jb.t_out, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight-stationary forward of an fc layer on the block-scaled MFMA:  C = e4m3(clamp(relu(A Wq^T * scales + b))) with the
// BatchNorm sums of the result and its maximum.  Same skeleton as gemm_ws16_kernel (one workgroup per CU, 4 waves = one per SIMD,
// a wave keeps the fragments of 64 features x all K in the accumulator half of its register file for the whole launch) with what
// one-byte operands change:
//   * 64 features x K bytes are 128 (K = 512) or 192 (K = 768) registers: fc1 fits ONE wave, no k split, no partial-sum exchange;
//   * a 64-row tile is 32 / 48 KiB: THREE buffers, the fetches of tile t+2 are requested during tile t (the bf16 kernel's k loop of
//     a tile was shorter than the fetch latency it had to hide with two);
//   * per lane 16 BatchNorm sums + 16 sums of squares run across ALL tiles of the launch and are folded over the 16 sample lanes
//     once at the end (per tile: two VALU instructions per output, no cross-lane traffic);
//   * the outputs of a sample tile leave as one 16-byte store per lane: the four feature tiles' dwords (4 features each) are
//     transposed over the four lanes that hold one sample (v_permlane32_swap, v_permlane16_swap), 16 rows x 64 bytes per store.
//   * the previous tile's epilogue is PACED between the MFMAs of the k loop: micro-operations of 3-5 vector instructions, each a volatile
//     asm statement, behind MFMAs that are volatile asm statements themselves (up to four vector instructions hide
//     behind one MX MFMA; hipcc's own schedule clustered both kinds).  tools/pinned_mfma_audit.py checks the wait states hipcc no
//     longer sees.  (Measured and not kept: the MFMA groups that lead a fetch left empty of micro-operations -- 51.2 / 50.5 us either way.)
// LDS image of a tile: row r at r*K, its 16-byte chunks XOR-swizzled with (r & 15) inside 256-byte groups (applied on the DMA's
// per-lane source), so the 16 rows of a fragment read fall on 16 different chunk positions.
// ------------------------------------------------------------------------------------------------------------------------
struct Ws8Args {
    const uint8_t* A;       // [M][K] e4m3
    const uint8_t* W;       // [F][K] e4m3, row-scaled (fold_linear8_kernel)
    const uint8_t* wsc;     // [F] scale bytes
    const float* bias;      // [F] f32, output units
    uint8_t* C;             // [M][F] e4m3
    float* partials;        // [workers][2][F]: sums of the stored-unit outputs and of their squares
    uint32_t* amax;         // the output tensor's running maximum
    int64_t M;
    int F;
};

template <int FT>
__device__ __forceinline__ f32x4_t mx_mfma(const i32x8_t& w, const i32x8_t& x, const f32x4_t& c, int wscale) {
    // A = weights (e4m3, scale byte FT of wscale), B = sample rows (e4m3, scale 2^0)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, x, c, 0, 0, FT, wscale, 0, 127);
}

// The same instruction as a volatile statement with the accumulator in VGPRs and the weights in AGPRs: it stays where it is written
// (between the micro-operations of a paced epilogue) and the accumulators stay where the epilogue reads them.  The compiler's hazard
// tables do not see inside: callers keep two wait states between a vector write of an operand and the statement, and eleven
// between the statement and a vector read of its result.
#define MX_MFMA_TEXT "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, "
#define MX_SEL0 " op_sel_hi:[0,0,0]"
#define MX_SEL1 " op_sel:[1,0,0] op_sel_hi:[0,0,0]"
#define MX_SEL2 " op_sel_hi:[1,0,0]"
#define MX_SEL3 " op_sel:[1,0,0] op_sel_hi:[1,0,0]"
// B5: the B operand (sample rows) is e5m2 (blgp:1: the data-gradient kernel's gradient rows); otherwise e4m3
template <int FT, bool B5 = false>
__device__ __forceinline__ void mx_mfma_pinned(const i32x8_t& w, const i32x8_t& x, f32x4_t& c, int wscale, int one) {
#define MX_ACC(SEL, FMT) asm volatile(MX_MFMA_TEXT "%0, %3, %4" SEL FMT : "+v"(c) : "a"(w), "v"(x), "v"(wscale), "v"(one))
    if constexpr (!B5) {
        if constexpr (FT == 0) MX_ACC(MX_SEL0, ""); if constexpr (FT == 1) MX_ACC(MX_SEL1, ""); if constexpr (FT == 2) MX_ACC(MX_SEL2, ""); if constexpr (FT == 3) MX_ACC(MX_SEL3, "");
    } else {
        if constexpr (FT == 0) MX_ACC(MX_SEL0, " blgp:1"); if constexpr (FT == 1) MX_ACC(MX_SEL1, " blgp:1"); if constexpr (FT == 2) MX_ACC(MX_SEL2, " blgp:1"); if constexpr (FT == 3) MX_ACC(MX_SEL3, " blgp:1");
    }
#undef MX_ACC
}
// first k block: d = w x + c0 with c0 in registers of its own (the bias, fresh from LDS: no vector write in front of the statement;
// the caller keeps c0 alive -- unwritten -- for seven states behind the statement: LLVM's SMFMA16x16ReadVgprVALUWar rule)
template <int FT>
__device__ __forceinline__ void mx_mfma_pinned0(const i32x8_t& w, const i32x8_t& x, f32x4_t& d, const f32x4_t& c0, int wscale, int one) {
#define MX_C0(SEL) asm volatile(MX_MFMA_TEXT "%5, %3, %4" SEL : "=&v"(d) : "a"(w), "v"(x), "v"(wscale), "v"(one), "v"(c0))
    if constexpr (FT == 0) MX_C0(MX_SEL0); if constexpr (FT == 1) MX_C0(MX_SEL1); if constexpr (FT == 2) MX_C0(MX_SEL2); if constexpr (FT == 3) MX_C0(MX_SEL3);
#undef MX_C0
}
// first k block of a product without bias: d = w x (C = the inline constant 0), e5m2 sample rows
template <int FT>
__device__ __forceinline__ void mx_mfma_pinned_z5(const i32x8_t& w, const i32x8_t& x, f32x4_t& d, int wscale, int one) {
#define MX_Z(SEL) asm volatile(MX_MFMA_TEXT "0, %3, %4" SEL " blgp:1" : "=&v"(d) : "a"(w), "v"(x), "v"(wscale), "v"(one))
    if constexpr (FT == 0) MX_Z(MX_SEL0); if constexpr (FT == 1) MX_Z(MX_SEL1); if constexpr (FT == 2) MX_Z(MX_SEL2); if constexpr (FT == 3) MX_Z(MX_SEL3);
#undef MX_Z
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// K = 512: 64-row tiles, three buffers.  K = 768 (fc1; 192 weight registers): 32-row tiles -- with 64 rows the two accumulator
// sets, the fragments and the running sums no longer fit beside the weights (56 spilled registers) -- and four buffers, so that
// a fetch still has two whole tiles to land.
template <int K> struct Ws8Cfg;
template <> struct Ws8Cfg<512> { static constexpr int RT = 64, NBUF = 3; };
template <> struct Ws8Cfg<768> { static constexpr int RT = 32, NBUF = 4; };

// paced epilogues: micro-operation u of NU = upst * st_count rides behind MFMA g with g NU / NG <= u < (g + 1) NU / NG; the store of sample
// tile st is the last of its upst micro-operations.  How many of a step's stores are issued at or behind MFMA group g_from?
constexpr int paced_stores_from(int ng, int st_count, int upst, int g_from) {
    int n = 0;
    for (int st = 0; st < st_count; ++st) {
        const int u = upst * st + upst - 1, nu = upst * st_count;
        int g = 0;
        while (!(g * nu / ng <= u && u < (g + 1) * nu / ng)) ++g;
        if (g >= g_from) ++n;
    }
    return n;
}
template <int K, bool STATS = true>        // (STATS false: evaluation with the running statistics -- no column sums, gemm_ws.cuh)
__global__ __launch_bounds__(256, 1) void gemm_ws8_kernel(Ws8Args a) {
    constexpr int KB = K / 128, RT = Ws8Cfg<K>::RT, ST = RT / 16, NBUF = Ws8Cfg<K>::NBUF, AHEAD = NBUF - 1;
    constexpr int TILE_BYTES = RT * K, UPW = TILE_BYTES / 1024 / 4, CPR = K / 16;
    // stores of one paced epilogue (see epi_uop below) that are issued AFTER the step's last fetch unit: the store of sample tile st is
    // micro-operation 19 st + 18 of NU = 19 ST, behind MFMA g with g NU / NG <= 19 st + 18 < (g + 1) NU / NG; the last fetch unit leads
    // MFMA group 4 (UPW - 1)
    constexpr int SA = paced_stores_from(KB * ST * 4, ST, 19, 4 * (UPW - 1));
    constexpr bool PF2 = ST < 4;         // few sample tiles: fragments of k block kb + 1 are requested in front of the MFMAs of kb (two sets)
    static_assert(UPW <= KB * ST, "one fetch unit per (k block, sample tile) slot at most");
    static_assert((AHEAD - 1) * UPW + AHEAD * ST <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * TILE_BYTES + 512 * 4];
    float* bias_s = (float*)(smem + NBUF * TILE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + RT - 1) / RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    for (int q = tid; q < a.F; q += 256) bias_s[q] = a.bias[q];
    if (ntile == 0) return;
    const int f0 = fb * 256 + wave * 64;

    // weights: fragment (ft, kb) = row f0 + ft*16 + s16, bytes kb*128 + 32*q4 .. +31; pinned in the accumulator half of the file
    i32x8_t wreg[4][KB];
    int wscale = 0;
    {
        const uint8_t* Wg = a.W + (int64_t)(f0 + s16) * K + 32 * q4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const i32x4_t lo = *(const i32x4_t*)(Wg + (int64_t)ft * 16 * K + kb * 128);
                const i32x4_t hi = *(const i32x4_t*)(Wg + (int64_t)ft * 16 * K + kb * 128 + 16);
                wreg[ft][kb] = (i32x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            wscale |= (int)a.wsc[f0 + ft * 16 + s16] << (8 * ft);
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+a"(wreg[ft][kb]));
    }

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * K), 0x00020000u};
    // fetch unit q of this wave = LDS bytes [(wave*UPW + q) * 1024, +1024) of the tile image: lane l lands on 16-byte slot
    // g = unit*64 + l = (row g / CPR, physical chunk g % CPR) and fetches the logical chunk that belongs there
    uint32_t fsrc[UPW];
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        const int g = (wave * UPW + q) * 64 + lane, row = g / CPR, pc = g % CPR;
        fsrc[q] = (uint32_t)(row * K + (((pc & ~15) | ((pc ^ row) & 15)) << 4));
    }
    auto fetch_unit = [&](uint32_t tile_soff, int buf, int q) {
#ifdef WS8_NO_FETCH
        tile_soff = 0xFFF00000u;                    // ablation build (tools only): every fetch is out of range (zeros, no memory traffic)
#endif
        bufl16_lds(a_rsrc, fsrc[q], tile_soff, lds0 + buf * TILE_BYTES + (wave * UPW + q) * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };
    // (no such tile: an offset past the end of the buffer -- the fetches return zeros into an idle buffer, and the counted waits
    //  below see the same number of operations in every tile)
    auto tile_soff = [&](int ti) -> uint32_t { return ti < ntile ? (uint32_t)(row0(ti) * K) : 0xFFF00000u; };

    float s1[16], s2[16], amax = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) s1[p] = s2[p] = 0.f;
    int scale_one = 127;                                                     // E8M0 2^0 for the sample rows, in a register of its own
    asm volatile("" : "+v"(scale_one));
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.F), 0x00020000);
    const uint32_t c_lane = (uint32_t)(s16 * a.F + f0 + q4 * 16);

    // epilogue of sample tile st of a finished tile: clamp(relu), sums, maximum, convert, transpose over the sample's 4 lanes, store
    auto epi_st = [&](f32x4_t (&old)[4][ST], int st, uint32_t s_old, bool live) {
#ifdef WS8_NO_EPI
        // ablation build (tools only): keep the finished accumulators alive, do nothing with them
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) asm volatile("" :: "v"(old[ft][st]));
        (void)s_old; (void)live;
        return;
#endif
        uint32_t d[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __builtin_amdgcn_fmed3f(old[ft][st][e], 0.f, F8_E4M3_MAX);
                if constexpr (STATS) {
                    const float w = live ? v[e] : 0.f;
                    s1[ft * 4 + e] += w;
                    s2[ft * 4 + e] = fmaf(w, w, s2[ft * 4 + e]);
                }
            }
            amax = fmaxf(fmaxf(amax, old[ft][st][0]), old[ft][st][1]);
            amax = fmaxf(fmaxf(amax, old[ft][st][2]), old[ft][st][3]);
            d[ft] = f8_pack4(v[0], v[1], v[2], v[3]);
        }
        // lane q4 of a sample: dword ft = features ft*16 + 4*q4 .. +3  ->  dword c = features q4*16 + 4*c .. +3
        { const auto x = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false); d[0] = x[0]; d[2] = x[1]; }
        { const auto x = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false); d[1] = x[0]; d[3] = x[1]; }
        { const auto x = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false); d[0] = x[0]; d[1] = x[1]; }
        { const auto x = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false); d[2] = x[0]; d[3] = x[1]; }
        const u32x4_t c = {d[0], d[1], d[2], d[3]};
        store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.F), 0);
    };

    // fragment (kb, st) of a tile = row st*16 + s16, logical chunks kb*8 + q4*2 (+1), physical chunk = logical ^ s16: the XOR reaches
    // bits 0-3 of the chunk index, of which kb owns bit 3 -- so a lane has four byte offsets (half h, parity of kb) and everything else
    // is an immediate of the ds_read (32 v_add_u32 and as many registers of precomputed offsets per tile were spent on it before)
    uint32_t fro[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pk = 0; pk < 2; ++pk) fro[h][pk] = (uint32_t)(s16 * K + ((pk ^ (s16 >> 3)) << 7) + (((q4 * 2 + h) ^ (s16 & 7)) << 4));
    auto load_frag = [&](const unsigned char* const (&Ab)[2][2], int kb, int st) -> i32x8_t {
        const int imm = (kb >> 1) * 256 + st * 16 * K;
        const i32x4_t lo = *(const i32x4_t*)(Ab[0][kb & 1] + imm);
        const i32x4_t hi = *(const i32x4_t*)(Ab[1][kb & 1] + imm);
        return (i32x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // The previous tile's epilogue as NU micro-operations, handed out evenly over the NG = KB * ST * 4 MFMAs of a k loop.  Per sample tile st:
    // 16 x (clamp, sum, sum of squares of ONE output; with the odd ones the maximum and the conversion of the pair), two lane-swap
    // pairs, the store.  tools/coissue8_probe.hip: behind each v_mfma_scale_f32_16x16x128 up to four vector instructions cost 1-4
    // cycles of its 32, and the loop is bound by its fetches (34 us without any epilogue) -- what must not happen is what the compiler's
    // own schedule did with the same instructions: runs of 12-16 bare MFMAs, then runs of 30 vector instructions.
    // Placement is pinned by DATA: an empty `asm volatile` with a "+v" operand redefines a value where it stands, volatile statements
    // keep their order, so whatever consumes the value cannot rise above the statement and whatever produces it cannot sink below.
    // (__builtin_amdgcn_sched_barrier only binds the machine scheduler: with it alone the MFMAs of the last k block sank into the next
    // step and the sums rose to its start.)  The accumulators live in VGPRs that way (the weights hold the AGPR half), so the epilogue
    // reads them in place -- the compiler's own allocation kept them in AGPRs and copied 64 registers per tile.
    constexpr int NG = KB * ST * 4, NU = ST * 19;
    static_assert(NU <= 2 * NG, "at most two micro-operations behind one MFMA");
    float vhold = 0.f, lim = F8_E4M3_MAX;
    uint32_t dq[4] = {0u, 0u, 0u, 0u};
    auto epi_uop = [&](f32x4_t (&old)[4][ST], auto uc, uint32_t s_old) {
        constexpr int u = decltype(uc)::value, st = u / 19, r = u % 19;
#ifdef WS8_NO_EPI
        if constexpr (r == 0) asm volatile("" :: "v"(old[0][st]), "v"(old[1][st]), "v"(old[2][st]), "v"(old[3][st]));   // ablation build (tools only)
        return;
#endif
        if constexpr (r < 16) {
            // one output: clamp (ReLU and the format's range), column sum, column sum of squares; with the odd ones the running maximum of
            // the pair and its conversion into one half of the feature tile's dword.  Whole statements: nothing of them moves, and
            // hipcc pads nothing around them (an empty pin in front of a compiler-placed v_med3 cost one s_nop per output).
            constexpr int ft = r / 4, e = r % 4, q = ft * 4 + e;
            const float o = old[ft][st][e];
            if constexpr ((e & 1) == 0) {
                if constexpr (STATS) asm volatile("v_med3_f32 %0, %3, 0, %4\n\tv_add_f32 %1, %1, %0\n\tv_fmac_f32 %2, %0, %0" : "=&v"(vhold), "+v"(s1[q]), "+v"(s2[q]) : "v"(o), "v"(lim));
                else asm volatile("v_med3_f32 %0, %1, 0, %2" : "=v"(vhold) : "v"(o), "v"(lim));
            } else {
                const float op = old[ft][st][e - 1];
                float v;
                if constexpr (STATS) {
                    if constexpr (e == 1) asm volatile("v_med3_f32 %0, %5, 0, %6\n\tv_add_f32 %1, %1, %0\n\tv_fmac_f32 %2, %0, %0\n\tv_max3_f32 %3, %3, %7, %5\n\tv_cvt_pk_fp8_f32 %4, %8, %0" : "=&v"(v), "+v"(s1[q]), "+v"(s2[q]), "+v"(amax), "+v"(dq[ft]) : "v"(o), "v"(lim), "v"(op), "v"(vhold));
                    else asm volatile("v_med3_f32 %0, %5, 0, %6\n\tv_add_f32 %1, %1, %0\n\tv_fmac_f32 %2, %0, %0\n\tv_max3_f32 %3, %3, %7, %5\n\tv_cvt_pk_fp8_f32 %4, %8, %0 op_sel:[0,0,1]" : "=&v"(v), "+v"(s1[q]), "+v"(s2[q]), "+v"(amax), "+v"(dq[ft]) : "v"(o), "v"(lim), "v"(op), "v"(vhold));
                } else {
                    if constexpr (e == 1) asm volatile("v_med3_f32 %0, %3, 0, %4\n\tv_max3_f32 %1, %1, %5, %3\n\tv_cvt_pk_fp8_f32 %2, %6, %0" : "=&v"(v), "+v"(amax), "+v"(dq[ft]) : "v"(o), "v"(lim), "v"(op), "v"(vhold));
                    else asm volatile("v_med3_f32 %0, %3, 0, %4\n\tv_max3_f32 %1, %1, %5, %3\n\tv_cvt_pk_fp8_f32 %2, %6, %0 op_sel:[0,0,1]" : "=&v"(v), "+v"(amax), "+v"(dq[ft]) : "v"(o), "v"(lim), "v"(op), "v"(vhold));
                }
            }
        } else if constexpr (r == 16) {
            asm volatile("s_nop 1" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));     // (a statement's v_cvt_pk wrote dq[3]: two states in front of the swaps)
            { const auto x = __builtin_amdgcn_permlane32_swap(dq[0], dq[2], false, false); dq[0] = x[0]; dq[2] = x[1]; }
            { const auto x = __builtin_amdgcn_permlane32_swap(dq[1], dq[3], false, false); dq[1] = x[0]; dq[3] = x[1]; }
            asm volatile("" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));
        } else if constexpr (r == 17) {
            { const auto x = __builtin_amdgcn_permlane16_swap(dq[0], dq[1], false, false); dq[0] = x[0]; dq[1] = x[1]; }
            { const auto x = __builtin_amdgcn_permlane16_swap(dq[2], dq[3], false, false); dq[2] = x[0]; dq[3] = x[1]; }
            asm volatile("" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));
        } else {
            const u32x4_t c = {dq[0], dq[1], dq[2], dq[3]};
#ifdef WS8_NO_STORE
            asm volatile("" :: "v"(c));                                      // ablation build (tools only): everything but the store
#else
            store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.F), 0);
#endif
        }
    };

    // K loop of tile ti into acc; the fetches of tile ti + AHEAD go out along the way; WITH_EPI: the previous tile's epilogue rides
    // behind the MFMAs, micro-operations [g NU / NG, (g + 1) NU / NG) behind MFMA g; a group's fetch goes first in its group (the counted
    // waits below need the order of fetches and stores: both are volatile statements)
    auto step = [&](f32x4_t (&acc)[4][ST], f32x4_t (&old)[4][ST], int ti, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const int buf = ti % NBUF;
        const uint32_t next_soff = tile_soff(ti + AHEAD);
        const int nbuf = (ti + AHEAD) % NBUF;
        const uint32_t s_old = (uint32_t)(m_old * a.F);
        f32x4_t b0[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) b0[ft] = *(const f32x4_t*)(bias_s + f0 + ft * 16 + 4 * q4);
        auto keep_b0 = [&]() { asm volatile("" :: "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3])); };
        const unsigned char* const At[2][2] = {{smem + buf * TILE_BYTES + fro[0][0], smem + buf * TILE_BYTES + fro[0][1]},
                                               {smem + buf * TILE_BYTES + fro[1][0], smem + buf * TILE_BYTES + fro[1][1]}};
        i32x8_t fa[PF2 ? 2 : 1][ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[0][st] = load_frag(At, 0, st);
        static_for<KB>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            if constexpr (PF2 && kb + 1 < KB) {
#pragma unroll
                for (int st = 0; st < ST; ++st) fa[(kb + 1) & 1][st] = load_frag(At, kb + 1, st);
            }
            static_for<ST>([&](auto stc) {
                constexpr int st = decltype(stc)::value;
                if constexpr (kb * ST + st < UPW) fetch_unit(next_soff, nbuf, kb * ST + st);
                const i32x8_t& x = fa[PF2 ? (kb & 1) : 0][st];
                static_for<4>([&](auto ftc) {
                    constexpr int ft = decltype(ftc)::value, g = (kb * ST + st) * 4 + ft;
                    if constexpr (kb == 0) mx_mfma_pinned0<ft>(wreg[ft][kb], x, acc[ft][st], b0[ft], wscale, scale_one);
                    else mx_mfma_pinned<ft>(wreg[ft][kb], x, acc[ft][st], wscale, scale_one);
                    if constexpr (WITH_EPI) {
                        constexpr int u0 = g * NU / NG, u1 = (g + 1) * NU / NG;
                        static_for<u1 - u0>([&](auto jc) { epi_uop(old, std::integral_constant<int, u0 + decltype(jc)::value>{}, s_old); });
                    }
                    if constexpr (ft == 3 && !PF2 && kb + 1 < KB) fa[0][st] = load_frag(At, kb + 1, st);
                    // (an MFMA reads its C operand for several cycles: the bias registers of the first k block stay untouched by
                    //  vector writes for at least seven states behind their last MFMA -- here: four more MFMAs)
                    if constexpr (kb == 1 && st == 0 && ft == 3) keep_b0();
                });
            });
        });
        asm volatile("s_nop 11");                                            // (pinned MFMAs in front of whatever reads their results: 12 states)
        // tile ti + 1 must have landed.  vmcnt retires in issue order (tools/vmcnt_order_probe.hip); younger than its LAST fetch are the
        // fetches of tiles ti + 2 .. ti + AHEAD and the stores of the epilogues that ran since: min(AHEAD, ti) epilogues, of the
        // oldest of which only the SA stores behind that step's last fetch count when all AHEAD are there
#if defined(WS8_NO_STORE) || defined(WS8_NO_EPI)
        constexpr int STC = 0, SAC = 0;                                      // ablation builds (tools only): no stores to count
#else
        constexpr int STC = ST, SAC = SA;
#endif
        const int ne = ti < AHEAD ? ti : AHEAD;
        if (ne == 0) wait_vmcnt<(AHEAD - 1) * UPW>();
        else if (ne < AHEAD) { if (ne == 1) wait_vmcnt<(AHEAD - 1) * UPW + STC>(); else wait_vmcnt<(AHEAD - 1) * UPW + 2 * STC>(); }
        else wait_vmcnt<(AHEAD - 1) * UPW + (AHEAD - 1) * STC + SAC>();
        __builtin_amdgcn_s_barrier();
    };
    auto drain = [&](f32x4_t (&old)[4][ST], int64_t m_old) {
        const uint32_t s_old = (uint32_t)(m_old * a.F);
#pragma unroll
        for (int st = 0; st < ST; ++st) epi_st(old, st, s_old, m_old + st * 16 + s16 < a.M);
    };

    f32x4_t accA[4][ST], accB[4][ST];
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) {
        const uint32_t so = tile_soff(t);
#pragma unroll
        for (int q = 0; q < UPW; ++q) fetch_unit(so, t, q);
    }
    wait_vmcnt<(AHEAD - 1) * UPW>();
    __syncthreads();                                                         // bias table + tile 0
    step(accA, accB, 0, std::false_type{}, 0);
    int ti = 1;
    while (ti + 1 < ntile) {
        step(accB, accA, ti, std::true_type{}, row0(ti - 1));
        step(accA, accB, ti + 1, std::true_type{}, row0(ti));
        ti += 2;
    }
    if (ti < ntile) {
        step(accB, accA, ti, std::true_type{}, row0(ti - 1));
        drain(accB, row0(ti));
    } else {
        drain(accA, row0(ntile - 1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // (dummy fetches into LDS must not outlive the workgroup)

    // column sums over the 16 sample lanes: value index ft*4 + e -> two folds of 8; lane s16 < 8 ends with index s16 (and 8 + s16)
    float r1[2], r2[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        float v1[8], v2[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { v1[p] = s1[hh * 8 + p]; v2[p] = s2[hh * 8 + p]; }
        r1[hh] = row16_fold8(v1, lane);
        r2[hh] = row16_fold8(v2, lane);
    }
    if (STATS && s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int idx = hh * 8 + s16, f = f0 + (idx >> 2) * 16 + 4 * q4 + (idx & 3);
            a.partials[(prow * 2 + 0) * a.F + f] = r1[hh];
            a.partials[(prow * 2 + 1) * a.F + f] = r2[hh];
        }
    }
    f8_atomic_amax(a.amax, amax);
}

template <int K>
static inline hipError_t launch_gemm_ws8(const Ws8Args& a, hipStream_t st, int* stat_rows) {
    if ((a.F & 255) || a.F > 512 || a.M <= 0 || (uint64_t)a.M * K >= 0xFFF00000ull) return hipErrorInvalidValue;
    const int nwk = 32 / (a.F >> 8);
    const int64_t tiles = (a.M + Ws8Cfg<K>::RT - 1) / Ws8Cfg<K>::RT, workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
    if (a.partials != nullptr) hipLaunchKernelGGL((gemm_ws8_kernel<K, true>), dim3(256), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_ws8_kernel<K, false>), dim3(256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// u = dropout(BatchNorm(r)) for fc4..fc6 (code/models.py:282,287,292), 8-bit in and out: u8 = e4m3(clamp(u 2^eu)),
// u = mask (s r8 2^-er + t) / (1 - p); the maximum of |u| 2^eu is tracked.  A thread keeps one 16-byte column chunk (16 features).
// Same mask as the bf16 kernels: dropout_pair(key, row, C, col).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_dropout_apply8_kernel(const uint8_t* __restrict__ r, const float* __restrict__ stats,
                                                                uint8_t* __restrict__ u, int64_t rows, int C, uint32_t thresh, uint32_t key,
                                                                float inv_keep, const uint32_t* __restrict__ salt, Fp8State* __restrict__ st,
                                                                int t_in, int t_out) {
    f8_saturating_conversions();
    if (salt) key ^= *salt;
    const float din = f8_exp2i(-st->e[t_in]), dout = f8_exp2i(st->e[t_out]);
    const int cpr = C / 16, rpp = blockDim.x / cpr;
    const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
    const int f = cc * 16;
    // (1 / (1 - p) rides in the scale and the shift: a kept element is one fma, a dropped one a select)
    float sc[16], sh[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { sc[e] = stats[2 * C + f + e] * din * dout * inv_keep; sh[e] = stats[3 * C + f + e] * dout * inv_keep; }
    float amax = 0.f;
    auto apply = [&](const uint4& in, int64_t m) {
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
            f8_unpack4(w[q], v);
            const uint2 qd = dropout_quad(key, (uint32_t)m, (uint32_t)C, (uint32_t)(f + 4 * q));     // four 16-bit draws: columns f + 4q .. + 3
            const uint32_t d[4] = {qd.x & 0xFFFFu, qd.x >> 16, qd.y & 0xFFFFu, qd.y >> 16};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = fmaf(v[e], sc[4 * q + e], sh[4 * q + e]);
                v[e] = d[e] >= thresh ? y : 0.f;
            }
            amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
            amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
            o[q] = f8_pack4(v[0], v[1], v[2], v[3]);
        }
        *(uint4*)(u + m * C + f) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    // four rows in flight per thread (two were: 55-69 us for 172 MB, nowhere near the copy rate; eight: no further gain on 512 workgroups)
    const int64_t step = (int64_t)gridDim.x * rpp;
    int64_t m = (int64_t)blockIdx.x * rpp + rr;
    for (; m + 3 * step < rows; m += 4 * step) {
        const uint4 a0 = *(const uint4*)(r + m * C + f);
        const uint4 a1 = *(const uint4*)(r + (m + step) * C + f);
        const uint4 a2 = *(const uint4*)(r + (m + 2 * step) * C + f);
        const uint4 a3 = *(const uint4*)(r + (m + 3 * step) * C + f);
        apply(a0, m);
        apply(a1, m + step);
        apply(a2, m + 2 * step);
        apply(a3, m + 3 * step);
    }
    for (; m < rows; m += step) apply(*(const uint4*)(r + m * C + f), m);
    f8_atomic_amax(&st->amax[t_out], amax);
}

// e4m3 tensor -> bf16 in true units (exact: three mantissa bits, power-of-two scale).  Bridge for the parts of the backward pass
// that still run on the bf16 kernels, and the debug read-back.
__global__ __launch_bounds__(256) void dequant8_bf16_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ out, int64_t n16,
                                                            const Fp8State* __restrict__ st, int t) {
    const float d = f8_exp2i(-st->e[t]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 c = *(const uint4*)(in + i * 16);
        const uint32_t w[4] = {c.x, c.y, c.z, c.w};
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
            f8_unpack4(w[q], v);
            o[2 * q] = pack2bf(v[0] * d, v[1] * d);
            o[2 * q + 1] = pack2bf(v[2] * d, v[3] * d);
        }
        *(uint4*)(out + i * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        *(uint4*)(out + i * 16 + 8) = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

__global__ void dequant8_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t n, const Fp8State* __restrict__ st, int t) {
    const float d = f8_exp2i(-st->e[t]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = __builtin_amdgcn_cvt_f32_fp8((int)in[i], 0) * d;
}

// ========================================================================================================================
// Backward pass in 8 bits.  Gradients between layers are stored as OCP e5m2 (two mantissa bits, 32 binades: the format made for
// gradients), activations stay e4m3; every product is a block-scaled MFMA with the operand formats mixed per operand
// (cbsz / blgp), f32 accumulation.  Two kinds of gradient tensor travel through HBM:
//   F8_T_GRAD + l : dL/d(pre-activation of layer l) -- the weight gradient's X operand and the data gradient's A operand;
//   F8_T_GB + l   : dL/d(BatchNorm_l output) behind a dropout (masked), rewritten in place into the first kind by bn_relu_bwd8.
// ========================================================================================================================
enum { F8_T_GB = 32 };

__device__ __forceinline__ float f8_clamp5(float v) { return __builtin_amdgcn_fmed3f(v, -F8_E5M2_MAX, F8_E5M2_MAX); }
__device__ __forceinline__ uint32_t f8_pack4_e5m2(float a, float b, float c, float d) {
    int p = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, p, true);
    return (uint32_t)p;
}
__device__ __forceinline__ void f8_unpack4_e5m2(uint32_t w, float* o) {
    const auto lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
    o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
}

// ------------------------------------------------------------------------------------------------------------------------
// W^T for the data gradients, quantised to e4m3 with one power-of-two scale per row of W^T (= input feature k of the layer):
//   out_w[k'][j] = e4m3(W[j][k] 2^ek),  out_sc[k'] = 127 - ek - eg   (eg = scale exponent of the gradient the launch will read)
// so that the accumulators of the data-gradient launch are in TRUE units.  One block per 64 rows of W^T; two sweeps over its
// 512 x 64 slice of W (the second hits L2).  mode 1 (fc1): k' = w*64 + c of k = c*12 + w.
// ------------------------------------------------------------------------------------------------------------------------
struct Transpose8Job { const float* W; uint8_t* out_w; uint8_t* out_sc; int K, mode, t_grad; };
struct Transpose8Batch { Transpose8Job job[CP_N_FC]; };
__global__ __launch_bounds__(256) void transpose_w8_batch_kernel(Transpose8Batch b, const Fp8State* __restrict__ st) {
    // grid (K / 64 column blocks, jobs, 4 quarters of the 512 rows of W): every block finds the column maxima itself (the whole
    // slice, L2-resident after the first reader) and quantises its quarter.  (The first version -- 84 blocks walking 128 + 128
    // dependent-latency loads each -- took 49 us.)
    __shared__ float cmax[4][64];
    __shared__ uint8_t tile[64][64 + 4];
    const Transpose8Job j = b.job[blockIdx.y];
    const int k0 = blockIdx.x * 64;
    if (k0 >= j.K) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int kp = k0 + lane;                                          // row of W^T in the internal order
    const int k = j.mode == 1 ? (kp & 63) * 12 + (kp >> 6) : kp;       // column of W
    float am = 0.f;
    for (int jj = grp; jj < 512; jj += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = j.W[(int64_t)(jj + 4 * u) * j.K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) am = fmaxf(am, fabsf(v[u]));
    }
    cmax[grp][lane] = am;
    __syncthreads();
    am = fmaxf(fmaxf(cmax[0][lane], cmax[1][lane]), fmaxf(cmax[2][lane], cmax[3][lane]));
    const int ek = f8_fit_exp(am, F8_E4M3_MAX);
    const float sc = f8_exp2i(ek);
    if (grp == 0 && blockIdx.z == 0) {
        int sb = 127 - ek - st->e[j.t_grad];
        j.out_sc[kp] = (uint8_t)(sb < 1 ? 1 : (sb > 254 ? 254 : sb));
    }
    for (int j0 = blockIdx.z * 128; j0 < blockIdx.z * 128 + 128; j0 += 64) {
        __syncthreads();
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = j.W[(int64_t)(j0 + grp + 4 * r) * j.K + k];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[r] * sc), 0.f, 0, false);
            tile[lane][grp + 4 * r] = (uint8_t)(p & 255);              // [k row][j]
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = grp + 4 * r;
            j.out_w[(int64_t)(k0 + kk) * 512 + j0 + lane] = tile[kk][lane];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Data gradient of an fc layer:  G_out[m][k] = epilogue( sum_f G[m][f] W[f][k] ),  G e5m2 [M][512], W^T e4m3 row-scaled (true-unit
// accumulators), weight-stationary as gemm_ws8_kernel with 32-row tiles and four A buffers; the wave fetches the 32 x 64 sub-tile
// of the saved e4m3 activation R it needs (2 KiB, two LDS-DMA instructions) during the tile's own k loop and reads it back, in
// the accumulator layout, in that tile's epilogue one k loop later (its own vmcnt is the only ordering).
//   MODE 0: BatchNorm + ReLU backward of the layer below in the epilogue (coef), output = dL/d(its pre-activation), column sums =
//           its bias gradient.
//   MODE 1: behind a dropout: R is the forward pass's dropout OUTPUT u = mask (s r + t) / (1 - p) (e4m3, stored for the weight
//           gradient anyway): u != 0 IS the mask (no hash in the epilogue -- the first build recomputed it and spent 2.4x the matrix
//           pipe's cycles in VALU), output = masked dL/d(BatchNorm output) (F8_T_GB), and the two BatchNorm-backward sums follow
//           from sum g and sum g u per feature:  sum g r = ((1 - p) sum g u - t sum g) / s   (exact where the mask is 1, and g is 0
//           elsewhere).  A kept value that rounded to zero in the forward pass (|u| < 2^-10 / scale) counts as dropped.
// LDS image of R per wave and buffer: row r at r*64, its four 16-byte chunks XORed with (r >> 2) & 3 (rows r, r+4, r+8, r+12 of a
// ds_read_b32 share their bank group otherwise).
// ------------------------------------------------------------------------------------------------------------------------
struct Wsd8Args {
    const uint8_t* A;       // [M][512] e5m2 gradient
    const uint8_t* W;       // [F][512] e4m3 = W^T of the layer, row-scaled (transpose_w8_batch_kernel)
    const uint8_t* wsc;     // [F]
    const uint8_t* R;       // [M][F] e4m3: MODE 0 saved activation of the layer below, MODE 1 its dropout output u
    const float* bn_stats;  // MODE 1: [4][F] mean, invstd, scale s, shift t of the layer below (to turn sum g u into sum g r)
    void* C;                // [M][F] e5m2
    float* partials;        // MODE 0: [workers][F] column sums (true units);  MODE 1: [workers][2][F]
    const float* coef;      // MODE 0: [3][coef_mod]
    int coef_mod;
    Fp8State* st;
    int t_r, t_out;         // scale-table ids of R and of the output
    int64_t M;
    int F;
    uint32_t dp_thresh, dp_key;
    const uint32_t* dp_salt;
    float dp_inv_keep;
};

template <int FT>
__device__ __forceinline__ f32x4_t mx_mfma_g(const i32x8_t& w, const i32x8_t& g, const f32x4_t& c, int wscale) {
    // A = weights (e4m3), B = gradient rows (e5m2)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, g, c, 0, 1, FT, wscale, 0, 127);
}

#define WSD8_RT 32
template <int MODE>
__global__ __launch_bounds__(256, 1) void gemm_wsd8_kernel(Wsd8Args a) {
    f8_saturating_conversions();                       // (the e5m2 stores of the epilogue carry no clamp)
    constexpr bool STATS = MODE == 1;
    constexpr int K = 512, KB = K / 128, RT = WSD8_RT, ST = RT / 16, NBUF = 4, AHEAD = NBUF - 1;
    constexpr int TILE_BYTES = RT * K, UPW = TILE_BYTES / 1024 / 4, CPR = K / 16, R_BYTES = RT * 64, RU = R_BYTES / 1024;
    constexpr int R_OFF = NBUF * TILE_BYTES, RBUF = 3;
    static_assert((AHEAD - 1) * UPW + AHEAD * (ST + RU) <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(16))) unsigned char smem[R_OFF + 4 * RBUF * R_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = a.F >> 8;
    const int nwk = 32 / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + RT - 1) / RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    if (ntile == 0) return;
    const int f0 = fb * 256 + wave * 64;
    const int e_r = a.st->e[a.t_r], e_o = a.st->e[a.t_out];
    const float so = f8_exp2i(e_o);
    const float keep_so = a.dp_inv_keep * so;                                // MODE 1: 1 / (1 - p) and the output scale

    // MODE 0: BatchNorm-backward coefficients of this lane's 16 features, with the scales folded in: the epilogue's result is the
    // output in STORED units:  y 2^eo = ca 2^eo acc + cb 2^(eo - er) r8 + cz 2^eo
    float4 cfa[STATS ? 1 : 4], cfb[STATS ? 1 : 4], cfz[STATS ? 1 : 4];
    if constexpr (!STATS) {
        const float sr = f8_exp2i(e_o - e_r);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float v[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] = a.coef[c * a.coef_mod + (f0 + ft * 16 + 4 * q4 + e) % a.coef_mod];
            cfa[ft] = make_float4(v[0][0] * so, v[0][1] * so, v[0][2] * so, v[0][3] * so);
            cfb[ft] = make_float4(v[1][0] * sr, v[1][1] * sr, v[1][2] * sr, v[1][3] * sr);
            cfz[ft] = make_float4(v[2][0] * so, v[2][1] * so, v[2][2] * so, v[2][3] * so);
        }
    }

    i32x8_t wreg[4][KB];
    int wscale = 0;
    {
        const uint8_t* Wg = a.W + (int64_t)(f0 + s16) * K + 32 * q4;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const i32x4_t lo = *(const i32x4_t*)(Wg + (int64_t)ft * 16 * K + kb * 128);
                const i32x4_t hi = *(const i32x4_t*)(Wg + (int64_t)ft * 16 * K + kb * 128 + 16);
                wreg[ft][kb] = (i32x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            wscale |= (int)a.wsc[f0 + ft * 16 + s16] << (8 * ft);
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) asm volatile("" : "+a"(wreg[ft][kb]));
    }

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint64_t a_base = (uint64_t)(uintptr_t)a.A, r_base = (uint64_t)(uintptr_t)a.R;
    const u32x4_t a_rsrc = {(uint32_t)a_base, (uint32_t)(a_base >> 32) & 0xFFFFu, (uint32_t)(a.M * K), 0x00020000u};
    const u32x4_t r_rsrc = {(uint32_t)r_base, (uint32_t)(r_base >> 32) & 0xFFFFu, (uint32_t)(a.M * a.F), 0x00020000u};
    uint32_t fsrc[UPW];
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        const int g = (wave * UPW + q) * 64 + lane, row = g / CPR, pc = g % CPR;
        fsrc[q] = (uint32_t)(row * K + (((pc & ~15) | ((pc ^ row) & 15)) << 4));
    }
    auto fetch_unit = [&](uint32_t tile_soff, int buf, int q) {
        bufl16_lds(a_rsrc, fsrc[q], tile_soff, lds0 + buf * TILE_BYTES + (wave * UPW + q) * 1024);
    };
    // R: unit k = rows 16k .. 16k+15 of the tile x this wave's 64 bytes; lane l = row 16k + (l >> 2), physical chunk l & 3
    uint32_t rsrc_l[RU];
#pragma unroll
    for (int k = 0; k < RU; ++k) {
        const int row = 16 * k + (lane >> 2);
        rsrc_l[k] = (uint32_t)(row * a.F + f0 + (((lane & 3) ^ ((row >> 2) & 3)) << 4));
    }
    auto fetch_r = [&](uint32_t tile_soff, int buf, int k) {
        bufl16_lds(r_rsrc, rsrc_l[k], tile_soff, lds0 + R_OFF + (wave * RBUF + buf) * R_BYTES + k * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };
    auto tile_soff = [&](int ti) -> uint32_t { return ti < ntile ? (uint32_t)(row0(ti) * K) : 0xFFF00000u; };

    float s1[16], s2[STATS ? 16 : 1], amax = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) s1[p] = 0.f;
#pragma unroll
    for (int p = 0; p < (STATS ? 16 : 1); ++p) s2[p] = 0.f;
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.F), 0x00020000);
    // after the 4-lane transpose a lane owns features q4*16 .. +15 of its sample (16 bytes)
    const uint32_t c_lane = (uint32_t)(s16 * a.F + f0 + q4 * 16);

    auto epi_st = [&](f32x4_t (&old)[4][ST], const unsigned char* Rw, int st, uint32_t s_old, bool live, int64_t /*m_old*/) {
        const int row = st * 16 + s16;
        const int rsw = (row >> 2) & 3;
        uint32_t d[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float rv[4];
            f8_unpack4(*(const uint32_t*)(Rw + row * 64 + ((ft ^ rsw) << 4) + 4 * q4), rv);
            float y[4] = {old[ft][st][0], old[ft][st][1], old[ft][st][2], old[ft][st][3]};
            if constexpr (!STATS) {
                const float4 ca = cfa[ft], cb = cfb[ft], cz = cfz[ft];
                y[0] = rv[0] > 0.f ? fmaf(ca.x, y[0], fmaf(cb.x, rv[0], cz.x)) : 0.f;
                y[1] = rv[1] > 0.f ? fmaf(ca.y, y[1], fmaf(cb.y, rv[1], cz.y)) : 0.f;
                y[2] = rv[2] > 0.f ? fmaf(ca.z, y[2], fmaf(cb.z, rv[2], cz.z)) : 0.f;
                y[3] = rv[3] > 0.f ? fmaf(ca.w, y[3], fmaf(cb.w, rv[3], cz.w)) : 0.f;
            } else {
                y[0] = rv[0] != 0.f ? y[0] * keep_so : 0.f;
                y[1] = rv[1] != 0.f ? y[1] * keep_so : 0.f;
                y[2] = rv[2] != 0.f ? y[2] * keep_so : 0.f;
                y[3] = rv[3] != 0.f ? y[3] * keep_so : 0.f;
            }
            // (the sums are of the values AS STORED: the bias gradient and the BatchNorm-backward sums derived from it must describe
            //  the tensor the next kernels read)
            amax = fmaxf(fmaxf(amax, fabsf(y[0])), fabsf(y[1]));
            amax = fmaxf(fmaxf(amax, fabsf(y[2])), fabsf(y[3]));
            d[ft] = f8_pack4_e5m2(y[0], y[1], y[2], y[3]);
            f8_unpack4_e5m2(d[ft], y);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w = live ? y[e] : 0.f;
                s1[ft * 4 + e] += w;
                if constexpr (STATS) s2[ft * 4 + e] = fmaf(w, rv[e], s2[ft * 4 + e]);
            }
        }
        { const auto x = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false); d[0] = x[0]; d[2] = x[1]; }
        { const auto x = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false); d[1] = x[0]; d[3] = x[1]; }
        { const auto x = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false); d[0] = x[0]; d[1] = x[1]; }
        { const auto x = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false); d[2] = x[0]; d[3] = x[1]; }
        const u32x4_t c = {d[0], d[1], d[2], d[3]};
        store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.F), 0);
    };

    // fragment reads on four lane bases + immediates (gemm_ws8_kernel)
    uint32_t fro[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int pk = 0; pk < 2; ++pk) fro[h][pk] = (uint32_t)(s16 * K + ((pk ^ (s16 >> 3)) << 7) + (((q4 * 2 + h) ^ (s16 & 7)) << 4));
    auto load_frag = [&](const unsigned char* const (&Ab)[2][2], int kb, int st) -> i32x8_t {
        const int imm = (kb >> 1) * 256 + st * 16 * K;
        const i32x4_t lo = *(const i32x4_t*)(Ab[0][kb & 1] + imm);
        const i32x4_t hi = *(const i32x4_t*)(Ab[1][kb & 1] + imm);
        return (i32x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    int scale_one = 127;
    asm volatile("" : "+v"(scale_one));

    // The previous tile's epilogue as micro-operations behind the MFMAs of the k loop (gemm_ws8_kernel has the mechanism: whole asm
    // statements, accumulators in VGPRs).  Here the epilogue is 7 vector instructions per output against ONE MFMA per two outputs -- the
    // launch is bound by its vector work -- so what pacing buys is the k loop's 32 x 32 matrix-pipe cycles underneath it, not the reverse.
    // Per sample tile st: 8 x (pair of outputs: unpack the r / u pair | first output | second output | conversion + maximum | the pair as
    // stored | its sums), two lane-swap pairs, the store: 51 micro-operations; NU = 102 behind NG = 32 MFMAs.
    // The R sub-tile of the tile whose epilogue runs is read at the start of the step, so it is fetched one step EARLIER than round 3's
    // (three buffers per wave): R(t + 1) goes out in step t, is waited for at the end of step t + 1 and read in step t + 2.
    constexpr int NG = KB * ST * 4, UPST = 51, NU = ST * UPST;
    // Two pairs of different feature tiles (A, B) alternate, statement by statement: hipcc pads one state between two asm statements
    // when the second reads what the first wrote (84 s_nop per tile in the first build), and no statement here reads its predecessor's
    // outputs -- two sets of temporaries and two running maxima.
    f32x2_t rvp[2] = {{0.f, 0.f}, {0.f, 0.f}}, ysp[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float yy[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, amax2[2] = {0.f, 0.f};
    uint32_t dq[4] = {0u, 0u, 0u, 0u};
    auto epi_uop = [&](f32x4_t (&old)[4][ST], const uint32_t (&rr)[4][ST], auto uc, uint32_t s_old) {
        constexpr int u = decltype(uc)::value, st = u / UPST, r = u % UPST;
        if constexpr (r < 48) {
            constexpr int blk = r / 12, ab = (r % 12) & 1, part = (r % 12) >> 1, ft = (blk >> 1) * 2 + ab, h = blk & 1, e0 = 2 * h, q = ft * 4 + e0;
            if constexpr (part == 0) {                      // the pair's r / u bytes as f32
                if constexpr (h == 0) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(rvp[ab]) : "v"(rr[ft][st]));
                else asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(rvp[ab]) : "v"(rr[ft][st]));
            } else if constexpr (part < 3) {
                // first / second output of the pair: y = BatchNorm + ReLU backward (MODE 0) or the dropout mask and 1 / (1 - p) (MODE 1)
                constexpr int e = e0 + part - 1;
                const float o = old[ft][st][e], rv = rvp[ab][part - 1];
                if constexpr (!STATS) {
                    const float ca = ((const float*)&cfa[ft])[e], cb = ((const float*)&cfb[ft])[e], cz = ((const float*)&cfz[ft])[e];
                    asm volatile("v_fma_f32 %0, %1, %2, %3\n\tv_fma_f32 %0, %4, %5, %0\n\tv_cmp_lt_f32 vcc, 0, %2\n\tv_cndmask_b32 %0, 0, %0, vcc"
                                 : "=&v"(yy[ab][part - 1]) : "v"(cb), "v"(rv), "v"(cz), "v"(ca), "v"(o) : "vcc");
                } else {
                    asm volatile("v_mul_f32 %0, %1, %2\n\tv_cmp_neq_f32 vcc, 0, %3\n\tv_cndmask_b32 %0, 0, %0, vcc" : "=&v"(yy[ab][part - 1]) : "v"(o), "v"(keep_so), "v"(rv) : "vcc");
                }
            } else if constexpr (part == 3) {               // the pair into its half of the feature tile's dword; the running maximum
                if constexpr (h == 0) asm volatile("v_cvt_pk_bf8_f32 %0, %2, %3\n\tv_max3_f32 %1, %1, |%2|, |%3|" : "+v"(dq[ft]), "+v"(amax2[ab]) : "v"(yy[ab][0]), "v"(yy[ab][1]));
                else asm volatile("v_cvt_pk_bf8_f32 %0, %2, %3 op_sel:[0,0,1]\n\tv_max3_f32 %1, %1, |%2|, |%3|" : "+v"(dq[ft]), "+v"(amax2[ab]) : "v"(yy[ab][0]), "v"(yy[ab][1]));
            } else if constexpr (part == 4) {               // the pair AS STORED, back in f32
                if constexpr (h == 0) asm volatile("v_cvt_pk_f32_bf8 %0, %1" : "=v"(ysp[ab]) : "v"(dq[ft]));
                else asm volatile("v_cvt_pk_f32_bf8_sdwa %0, %1 src0_sel:WORD_1" : "=v"(ysp[ab]) : "v"(dq[ft]));
            } else {                                        // the sums are of the values AS STORED
                if constexpr (STATS) asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_fmac_f32 %2, %4, %6\n\tv_fmac_f32 %3, %5, %7"
                                                  : "+v"(s1[q]), "+v"(s1[q + 1]), "+v"(s2[q]), "+v"(s2[q + 1]) : "v"(ysp[ab][0]), "v"(ysp[ab][1]), "v"(rvp[ab][0]), "v"(rvp[ab][1]));
                else asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3" : "+v"(s1[q]), "+v"(s1[q + 1]) : "v"(ysp[ab][0]), "v"(ysp[ab][1]));
            }
        } else if constexpr (r == 48) {
            asm volatile("s_nop 1" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));     // (a statement's conversion wrote dq[3]: two states in front of the swaps)
            { const auto x = __builtin_amdgcn_permlane32_swap(dq[0], dq[2], false, false); dq[0] = x[0]; dq[2] = x[1]; }
            { const auto x = __builtin_amdgcn_permlane32_swap(dq[1], dq[3], false, false); dq[1] = x[0]; dq[3] = x[1]; }
            asm volatile("" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));
        } else if constexpr (r == 49) {
            { const auto x = __builtin_amdgcn_permlane16_swap(dq[0], dq[1], false, false); dq[0] = x[0]; dq[1] = x[1]; }
            { const auto x = __builtin_amdgcn_permlane16_swap(dq[2], dq[3], false, false); dq[2] = x[0]; dq[3] = x[1]; }
            asm volatile("" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]), "+v"(dq[3]));
        } else {
            const u32x4_t c = {dq[0], dq[1], dq[2], dq[3]};
            store_b128_settled(c, c_rsrc, c_lane, s_old + (uint32_t)(st * 16 * a.F), 0);
        }
    };
    // stores of a step's epilogue at or behind the MFMA group of the step's last R fetch (slot UPW + RU - 1, four MFMAs per slot)
    constexpr int SA = paced_stores_from(NG, ST, UPST, 4 * (UPW + RU - 1));
    static_assert(UPW + RU <= KB * ST, "one fetch per (k block, sample tile) slot");

    auto step = [&](f32x4_t (&acc)[4][ST], f32x4_t (&old)[4][ST], int ti, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const int buf = ti % NBUF;
        const uint32_t next_soff = tile_soff(ti + AHEAD);
        const int nbuf = (ti + AHEAD) % NBUF;
        const uint32_t r_soff = ti + 1 < ntile ? (uint32_t)(row0(ti + 1) * a.F) : 0xFFF00000u;      // R(ti + 1)
        const int rbuf = (ti + 1) % RBUF;
        const uint32_t s_old = (uint32_t)(m_old * a.F);
        const unsigned char* const At[2][2] = {{smem + buf * TILE_BYTES + fro[0][0], smem + buf * TILE_BYTES + fro[0][1]},
                                               {smem + buf * TILE_BYTES + fro[1][0], smem + buf * TILE_BYTES + fro[1][1]}};
        i32x8_t fa[2][ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[0][st] = load_frag(At, 0, st);
        // the finished tile's r / u bytes: dword (ft, st) = features ft*16 + 4*q4 .. +3 of row st*16 + s16 (this wave's own sub-tile)
        uint32_t rr[4][ST];
        if constexpr (WITH_EPI) {
            const unsigned char* Rw = smem + R_OFF + (wave * RBUF + (ti - 1) % RBUF) * R_BYTES;
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                const int row = st * 16 + s16, rsw = (row >> 2) & 3;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) rr[ft][st] = *(const uint32_t*)(Rw + row * 64 + ((ft ^ rsw) << 4) + 4 * q4);
            }
        }
        static_for<KB>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            if constexpr (kb + 1 < KB) {
#pragma unroll
                for (int st = 0; st < ST; ++st) fa[(kb + 1) & 1][st] = load_frag(At, kb + 1, st);
            }
            static_for<ST>([&](auto stc) {
                constexpr int st = decltype(stc)::value, slot = kb * ST + st;
                if constexpr (slot < UPW) fetch_unit(next_soff, nbuf, slot);
                else if constexpr (slot < UPW + RU) fetch_r(r_soff, rbuf, slot - UPW);
                const i32x8_t& x = fa[kb & 1][st];
                static_for<4>([&](auto ftc) {
                    constexpr int ft = decltype(ftc)::value, g = slot * 4 + ft;
                    if constexpr (kb == 0) mx_mfma_pinned_z5<ft>(wreg[ft][kb], x, acc[ft][st], wscale, scale_one);
                    else mx_mfma_pinned<ft, true>(wreg[ft][kb], x, acc[ft][st], wscale, scale_one);
                    if constexpr (WITH_EPI) {
                        constexpr int u0 = g * NU / NG, u1 = (g + 1) * NU / NG;
                        static_for<u1 - u0>([&](auto jc) { epi_uop(old, rr, std::integral_constant<int, u0 + decltype(jc)::value>{}, s_old); });
                    }
                });
            });
        });
        asm volatile("s_nop 11");                                            // (pinned MFMAs in front of whatever reads their results: 12 states)
        // What must have landed: the A fetches of tile ti + 1 (issued in step ti - 2) and R(ti) (issued in step ti - 1, its last piece
        // at slot UPW + RU - 1).  Younger than that piece: the SA stores of step ti - 1 behind it (no epilogue in step 0) and everything of
        // this step -- UPW + RU fetches and ST stores (none in step 0).  Step 0 waits for A(1) behind the prologue's A(2).
        if (ti == 0) wait_vmcnt<2 * UPW + RU>();
        else if (ti == 1) wait_vmcnt<UPW + RU + ST>();
        else wait_vmcnt<UPW + RU + ST + SA>();
        __builtin_amdgcn_s_barrier();
    };
    auto drain = [&](f32x4_t (&old)[4][ST], int ti, int64_t m_old) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned char* Rw = smem + R_OFF + (wave * RBUF + ti % RBUF) * R_BYTES;
        const uint32_t s_old = (uint32_t)(m_old * a.F);
#pragma unroll
        for (int st = 0; st < ST; ++st) epi_st(old, Rw, st, s_old, m_old + st * 16 + s16 < a.M, m_old);
    };

    f32x4_t accA[4][ST], accB[4][ST];
    // prologue: A(0), R(0), A(1), A(2) in this order: the wait below leaves A(1) and A(2) in flight
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) {
        const uint32_t so_ = tile_soff(t);
#pragma unroll
        for (int q = 0; q < UPW; ++q) fetch_unit(so_, t, q);
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < RU; ++k) fetch_r((uint32_t)(row0(0) * a.F), 0, k);
        }
    }
    wait_vmcnt<(AHEAD - 1) * UPW>();
    __syncthreads();
    step(accA, accB, 0, std::false_type{}, 0);
    int ti = 1;
    while (ti + 1 < ntile) {
        step(accB, accA, ti, std::true_type{}, row0(ti - 1));
        step(accA, accB, ti + 1, std::true_type{}, row0(ti));
        ti += 2;
    }
    if (ti < ntile) {
        step(accB, accA, ti, std::true_type{}, row0(ti - 1));
        drain(accB, ti, row0(ti));
    } else {
        drain(accA, ntile - 1, row0(ntile - 1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // column sums: stored units -> true units
    const float u1 = f8_exp2i(-e_o), u2 = f8_exp2i(-e_o - e_r);
    float r1[2], r2[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        float v1[8], v2[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { v1[p] = s1[hh * 8 + p]; v2[p] = STATS ? s2[hh * 8 + p] : 0.f; }
        r1[hh] = row16_fold8(v1, lane) * u1;
        r2[hh] = STATS ? row16_fold8(v2, lane) * u2 : 0.f;
    }
    if (s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int idx = hh * 8 + s16, f = f0 + (idx >> 2) * 16 + 4 * q4 + (idx & 3);
            if constexpr (STATS) {
                // sum g u -> sum g r through the layer's BatchNorm affine: u (1 - p) = s r + t wherever g != 0
                const float sc = a.bn_stats[2 * a.F + f], sh = a.bn_stats[3 * a.F + f], mean = a.bn_stats[f];
                const float gr = fabsf(sc) > 1e-30f ? (r2[hh] / a.dp_inv_keep - sh * r1[hh]) / sc : mean * r1[hh];
                a.partials[(prow * 2 + 0) * a.F + f] = r1[hh];
                a.partials[(prow * 2 + 1) * a.F + f] = gr;
            } else {
                a.partials[prow * a.F + f] = r1[hh];
            }
        }
    }
    f8_atomic_amax(&a.st->amax[a.t_out], fmaxf(amax, fmaxf(amax2[0], amax2[1])));
}

template <int MODE>
static inline hipError_t launch_gemm_wsd8(const Wsd8Args& a, hipStream_t st, int* stat_rows) {
    if ((a.F & 255) || a.F > 768 || a.M <= 0 || (uint64_t)a.M * a.F >= 0xFFF00000ull || !a.R || (MODE == 0 && !a.coef)) return hipErrorInvalidValue;
    const int nwk = 32 / (a.F >> 8);
    const int64_t tiles = (a.M + WSD8_RT - 1) / WSD8_RT, workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
    hipLaunchKernelGGL((gemm_wsd8_kernel<MODE>), dim3(256), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of an fc layer in 8 bits:  P[p][q] = sum_m X[m][p] Y[m][q],  X = dL/d(pre-activation) e5m2 [M][512],
// Y = the layer's input e4m3 [M][Q] (saved activation or dropout output), on v_mfma_scale_f32_32x32x64_f8f6f4 (A = X e5m2,
// B = Y e4m3, scales 2^0: the product comes out in stored units and reduce_slabs_kernel divides the two tensor scales out).
// Same shape as gemm_tn256_kernel: 256 x 256 output tile per block, 8 waves of 128(p) x 64(q), the row axis split over blocks
// (one f32 slab per split), a ring of 4 stages filled by LDS-DMA with counted vmcnt.  What one-byte operands change:
//   * a stage is 64 rows x 256 bytes per operand (16 KiB, as before) but feeds 8 MFMAs of 64 samples per wave instead of 2 x 8 of 16;
//   * fragments (32 samples of one column per lane) are four ds_read_b64_tr_b8 (8 rows x 16 byte-columns per 16 lanes each);
//   * rows past the end of the tensors are ZERO-filled by the buffer bounds check of the LDS-DMA (rows_per_split is a multiple
//     of 64, so only the last stage of the last split is ragged): no clamping, no masking of fragments.
// LDS image of an operand stage: [64 rows][256 B]; the 16-byte chunk index of a row is XORed with 2 * (row & 7): the 8 rows of a
// transposed-read block and the two column blocks of a 32-lane half then fall on 16 different chunk positions.
// ------------------------------------------------------------------------------------------------------------------------
struct GemmTN8Args {
    const uint8_t* X;   // [M][ldx] e5m2
    const uint8_t* Y;   // [M][ldy] e4m3
    float* slabs;       // [splits][P][Q]
    int64_t M;
    int64_t rows_per_split;   // multiple of 64
    int ldx, ldy, P, Q, splits;
    const uint8_t* X2;  // optional second problem of the same shape (gemm_tn256_kernel)
    const uint8_t* Y2;
    float* slabs2;
};
typedef __attribute__((ext_vector_type(2))) int i32x2_t;

// fragment = samples 32*(lane >> 5) .. +31 of column col0 + (lane & 31) of a stage image
__device__ __forceinline__ i32x8_t tn8_frag(const unsigned char* tile, int col0, int lane) {
    const int g16 = lane >> 4, li = lane & 15, q = li >> 1, p = li & 1;
    const int row = 32 * (g16 >> 1) + q;                                // + 8 * read index (row & 7 = q for every read)
    const int lc = (col0 >> 4) + (g16 & 1);
    const unsigned char* ptr = tile + row * 256 + ((lc ^ (2 * q)) << 4) + 8 * p;
    i32x8_t f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const i32x2_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2_t*)(ptr + r * 8 * 256));
        f[2 * r] = v[0];
        f[2 * r + 1] = v[1];
    }
    return f;
}

#define TN8_STAGES 4
__global__ __launch_bounds__(512) void gemm_tn8_kernel(GemmTN8Args a) {
    constexpr int OP_BYTES = 64 * 256, STAGE = 2 * OP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TN8_STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wq = wave & 3;
    const int tiles_q = a.Q / 256, ntiles = (a.P / 256) * tiles_q;
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int per_problem = ntiles * ((a.splits + 7) / 8);
    const bool second = j >= per_problem;
    if (second) j -= per_problem;
    const uint8_t* Xg = second ? a.X2 : a.X;
    const uint8_t* Yg = second ? a.Y2 : a.Y;
    const int tile = j % ntiles;
    const int split = (j / ntiles) * 8 + xcd;
    if (split >= a.splits) return;
    const int p0 = (tile / tiles_q) * 256, q0 = (tile % tiles_q) * 256;
    const int64_t mb = (int64_t)split * a.rows_per_split;
    int64_t me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int nsteps = (int)((me - mb + 63) / 64);

    const uint64_t x_base = (uint64_t)(uintptr_t)Xg, y_base = (uint64_t)(uintptr_t)Yg;
    const u32x4_t x_rsrc = {(uint32_t)x_base, (uint32_t)(x_base >> 32) & 0xFFFFu, (uint32_t)(a.M * a.ldx), 0x00020000u};
    const u32x4_t y_rsrc = {(uint32_t)y_base, (uint32_t)(y_base >> 32) & 0xFFFFu, (uint32_t)(a.M * a.ldy), 0x00020000u};
    // a stage holds 64 rows x 256 B per operand = 16 LDS-DMA instructions of 4 rows; each of the 8 waves issues 2 per operand
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    uint32_t xoff[2], yoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int inst = wave * 2 + i, row = inst * 4 + (lane >> 4), pc = lane & 15;
        const int lc = pc ^ (2 * (row & 7));
        xoff[i] = (uint32_t)(row * a.ldx + p0 + lc * 16);
        yoff[i] = (uint32_t)(row * a.ldy + q0 + lc * 16);
    }
    auto stage = [&](int slot, int step) {
        const int64_t ms = mb + (int64_t)step * 64;
        const uint32_t Xs = lds0 + slot * STAGE, Ys = Xs + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (the whole row offset in the per-lane part: the bounds check that zero-fills rows past the end is on it)
            bufl16_lds(x_rsrc, xoff[i] + (uint32_t)(ms * a.ldx), 0u, Xs + (wave * 2 + i) * 1024);
            bufl16_lds(y_rsrc, yoff[i] + (uint32_t)(ms * a.ldy), 0u, Ys + (wave * 2 + i) * 1024);
        }
    };
    auto wait_and_stage = [&](int step) {
        const int ahead = (nsteps - 1 - step) < (TN8_STAGES - 2) ? (nsteps - 1 - step) : (TN8_STAGES - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (step + TN8_STAGES - 1 < nsteps) stage((step + TN8_STAGES - 1) % TN8_STAGES, step + TN8_STAGES - 1);
    };
#pragma unroll
    for (int s = 0; s < TN8_STAGES - 1; ++s)
        if (s < nsteps) stage(s, s);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][jj][g] = 0.f;
    for (int step = 0; step < nsteps; ++step) {
        wait_and_stage(step);
        const unsigned char* Xs = smem + (step % TN8_STAGES) * STAGE;
        const unsigned char* Ys = Xs + OP_BYTES;
        i32x8_t fx[4], fy[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fx[i] = tn8_frag(Xs, wp * 128 + i * 32, lane);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) fy[jj] = tn8_frag(Ys, wq * 64 + jj * 32, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                acc[i][jj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fx[i], fy[jj], acc[i][jj], 1, 0, 0, 127, 0, 127);
        __builtin_amdgcn_s_setprio(0);
    }
    // The slabs of the 8-bit path are bf16 (round 4, third part): a partial sum over 2,624+ rows of products of e5m2 gradients carries the
    // gradients' own quantisation noise, 2^-3 per element / sqrt(rows) ~ 2^-8.7, so eight mantissa bits round it at its noise level; the
    // 64 (32) slabs are summed in f32 (reduce_slabs_kernel<bf16_t>).  Half the slab bytes written here and read there: this launch
    // moves 316 MB at the copy rate, 66 of them slabs.  (The slab AREA keeps its f32 size and offsets.)
    bf16_t* slab = (bf16_t*)(second ? a.slabs2 : a.slabs) + (int64_t)split * a.P * a.Q;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int q = q0 + wq * 64 + jj * 32 + r;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int p = p0 + wp * 128 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                slab[(int64_t)p * a.Q + q] = f2bf(acc[i][jj][g]);
            }
        }
}

static inline hipError_t launch_gemm_tn8(const GemmTN8Args& a, hipStream_t st) {
    if ((a.P & 255) || (a.Q & 255) || (a.rows_per_split & 63) || (uint64_t)a.M * (a.ldx > a.ldy ? a.ldx : a.ldy) >= 0xFFF00000ull) return hipErrorInvalidValue;
    const int ntiles = (a.P / 256) * (a.Q / 256);
    const int groups = (a.splits + 7) / 8;
    const int problems = a.X2 ? 2 : 1;
    hipLaunchKernelGGL(gemm_tn8_kernel, dim3((unsigned)(problems * groups * 8 * ntiles)), dim3(512), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// BatchNorm + ReLU backward in place behind a dropout:  g (e5m2, F8_T_GB + l, masked dL/d(BN_l output))  ->
// dL/d(pre-activation_l) = [r > 0] (ca g + cb r + cz)  (e5m2, F8_T_GRAD + l), r = the saved e4m3 activation; per-block column
// sums of the result (the bias gradient), its maximum tracked.  A thread keeps one 16-byte column chunk (16 features).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_relu_bwd8_kernel(uint8_t* __restrict__ g, const uint8_t* __restrict__ r, const float* __restrict__ coef,
                                                           float* __restrict__ partials, int64_t rows, int C, Fp8State* __restrict__ st,
                                                           int t_in, int t_r, int t_out) {
    f8_saturating_conversions();
    extern __shared__ float dyn_red[];                  // [rpp][C]
    const int cpr = C / 16, rpp = 256 / cpr;
    const int tid = threadIdx.x, cc = tid % cpr, rr = tid / cpr;
    const int e_in = st->e[t_in], e_r = st->e[t_r], e_o = st->e[t_out];
    const float ka = f8_exp2i(e_o - e_in), kb = f8_exp2i(e_o - e_r), kz = f8_exp2i(e_o);
    float ca[16], cb[16], cz[16], sum[16], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        ca[e] = coef[cc * 16 + e] * ka;
        cb[e] = coef[C + cc * 16 + e] * kb;
        cz[e] = coef[2 * C + cc * 16 + e] * kz;
        sum[e] = 0.f;
    }
    auto apply = [&](const uint4& gq, const uint4& rq, int64_t m) {
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, rw[4] = {rq.x, rq.y, rq.z, rq.w};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gv[4], rv[4], y[4];
            f8_unpack4_e5m2(gw[q], gv);
            f8_unpack4(rw[q], rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = rv[e] > 0.f ? fmaf(ca[4 * q + e], gv[e], fmaf(cb[4 * q + e], rv[e], cz[4 * q + e])) : 0.f;
            amax = fmaxf(fmaxf(amax, fabsf(y[0])), fabsf(y[1]));
            amax = fmaxf(fmaxf(amax, fabsf(y[2])), fabsf(y[3]));
            o[q] = f8_pack4_e5m2(y[0], y[1], y[2], y[3]);
            // column sums of the values AS STORED: the next layer's BatchNorm-backward sums are derived from this bias gradient and
            // from products of the stored tensor (bn_bwd_sums_from_wgrad_kernel), so the two must describe the same numbers
            f8_unpack4_e5m2(o[q], y);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[4 * q + e] += y[e];
        }
        *(uint4*)(g + m * C + cc * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    const int64_t step = (int64_t)gridDim.x * rpp;
    int64_t m = (int64_t)blockIdx.x * rpp + rr;
    // four rows in flight per thread (two were, on four times the workgroups: round 4, third part -- api.hip CAP_BRB8)
    for (; m + 3 * step < rows; m += 4 * step) {
        const uint4 g0 = *(const uint4*)(g + m * C + cc * 16), r0 = *(const uint4*)(r + m * C + cc * 16);
        const uint4 g1 = *(const uint4*)(g + (m + step) * C + cc * 16), r1 = *(const uint4*)(r + (m + step) * C + cc * 16);
        const uint4 g2 = *(const uint4*)(g + (m + 2 * step) * C + cc * 16), r2 = *(const uint4*)(r + (m + 2 * step) * C + cc * 16);
        const uint4 g3 = *(const uint4*)(g + (m + 3 * step) * C + cc * 16), r3 = *(const uint4*)(r + (m + 3 * step) * C + cc * 16);
        apply(g0, r0, m);
        apply(g1, r1, m + step);
        apply(g2, r2, m + 2 * step);
        apply(g3, r3, m + 3 * step);
    }
    for (; m < rows; m += step) apply(*(const uint4*)(g + m * C + cc * 16), *(const uint4*)(r + m * C + cc * 16), m);
    const float un = f8_exp2i(-e_o);
#pragma unroll
    for (int e = 0; e < 16; ++e) dyn_red[rr * C + cc * 16 + e] = sum[e] * un;
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int q = 0; q < rpp; ++q) s += dyn_red[q * C + c];
        partials[(int64_t)blockIdx.x * C + c] = s;
    }
    f8_atomic_amax(&st->amax[t_out], amax);
}

// ------------------------------------------------------------------------------------------------------------------------
// The projection's data gradient behind fc7's dropout, 8-bit form of proj_dgrad_kernel (gemm_ws.cuh): g = mask (dz W) / (1 - p) is a
// rank-16 product.  PASS 1 (coefficients final) is what the step launches: g, r > 0 ? ca g + cb r + cz : 0, stored as e5m2 (F8_T_GRAD + 8),
// column sums = fc7's bias gradient.  PASS 0 -- the same product reduced to the two BatchNorm-backward sums against the saved e4m3
// activation R, nothing stored -- was round 3's way to the coefficients and is no longer instantiated: the sums come with the projection's
// weight gradient now (gemm_tn.cuh, proj_wgrad_sums_kernel<true>).  dz stays bf16 (16 live columns, [M][lda]); W = last_w^T in bf16 [512][K].
// ------------------------------------------------------------------------------------------------------------------------
struct Proj8Args {
    const bf16_t* A;        // dz [M][lda]
    const bf16_t* W;        // [512][K] bf16 (K >= 16: the first 16 columns are read)
    const uint8_t* R;       // [M][512] e4m3
    uint8_t* C;             // [M][512] e5m2 (PASS 1)
    float* partials;
    const float* coef;      // PASS 1: [3][512]
    Fp8State* st;
    int t_r, t_out;
    int64_t M;
    int lda, K;
    uint32_t dp_thresh, dp_key;
    const uint32_t* dp_salt;
    float dp_inv_keep;
};

template <int PASS>
__global__ __launch_bounds__(256, 2) void proj_dgrad8_kernel(Proj8Args a) {
    f8_saturating_conversions();
    constexpr int RT = 32, ST = RT / 16, R_BYTES = RT * 64, F = 512;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 2 * R_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, q4 = lane >> 4;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nfb = F >> 8;
    const int nwk = (gridDim.x >> 3) / nfb;
    const int fb = j % nfb, wkr = j / nfb;
    const int64_t tiles = (a.M + RT - 1) / RT;
    const int first = wkr * 8 + xcd, stride = nwk * 8;
    const int ntile = (wkr < nwk && first < tiles) ? (int)((tiles - first + stride - 1) / stride) : 0;
    if (ntile == 0) return;
    const int f0 = fb * 256 + wave * 64;
    const uint32_t dkey = a.dp_thresh != 0 ? (a.dp_salt ? (a.dp_key ^ *a.dp_salt) : a.dp_key) : 0u;
    const int e_r = a.st->e[a.t_r], e_o = PASS == 1 ? a.st->e[a.t_out] : 0;

    float4 cfa[PASS == 1 ? 4 : 1], cfb[PASS == 1 ? 4 : 1], cfz[PASS == 1 ? 4 : 1];
    if constexpr (PASS == 1) {
        const float so = f8_exp2i(e_o), sr = f8_exp2i(e_o - e_r);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float v[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] = a.coef[c * F + f0 + ft * 16 + 4 * q4 + e];
            cfa[ft] = make_float4(v[0][0] * so, v[0][1] * so, v[0][2] * so, v[0][3] * so);
            cfb[ft] = make_float4(v[1][0] * sr, v[1][1] * sr, v[1][2] * sr, v[1][3] * sr);
            cfz[ft] = make_float4(v[2][0] * so, v[2][1] * so, v[2][2] * so, v[2][3] * so);
        }
    }

    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    s16x4_t wfrag[4];                                                        // W[f0 + ft*16 + s16][4*q4 .. +3]
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) wfrag[ft] = *(const s16x4_t*)(a.W + (int64_t)(f0 + ft * 16 + s16) * a.K + 4 * q4);

    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint64_t r_base = (uint64_t)(uintptr_t)a.R;
    const u32x4_t r_rsrc = {(uint32_t)r_base, (uint32_t)(r_base >> 32) & 0xFFFFu, (uint32_t)(a.M * F), 0x00020000u};
    auto fetch_r = [&](int64_t m0, int buf) {
#pragma unroll
        for (int k = 0; k < R_BYTES / 1024; ++k) {
            const int row = 16 * k + (lane >> 2);
            bufl16_lds(r_rsrc, (uint32_t)((m0 + row) * F + f0 + (((lane & 3) ^ ((row >> 2) & 3)) << 4)), 0u, lds0 + (wave * 2 + buf) * R_BYTES + k * 1024);
        }
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };
    auto load_dz = [&](int64_t m0, s16x4_t (&d)[ST]) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            int64_t m = m0 + st * 16 + s16;
            if (m >= a.M) m = a.M - 1;
            d[st] = *(const s16x4_t*)(a.A + m * a.lda + 4 * q4);
        }
    };

    float s1[16], s2[PASS == 0 ? 16 : 1], amax = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) s1[p] = 0.f;
#pragma unroll
    for (int p = 0; p < (PASS == 0 ? 16 : 1); ++p) s2[p] = 0.f;
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * F), 0x00020000);
    const uint32_t c_lane = (uint32_t)(s16 * F + f0 + q4 * 16);

    s16x4_t dzf[ST], dzn[ST];
    fetch_r(row0(0), 0);
    load_dz(row0(0), dzf);
    for (int ti = 0; ti < ntile; ++ti) {
        const int buf = ti & 1;
        const int64_t m0 = row0(ti);
        // this tile's R sub-tile has landed (own DMA only).  PASS 1: the previous tile's ST stores are younger and stay in flight
        if (PASS == 1 && ti > 0) wait_vmcnt<ST>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ti + 1 < ntile) { fetch_r(row0(ti + 1), buf ^ 1); load_dz(row0(ti + 1), dzn); }
        f32x4_t acc[4][ST];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int st = 0; st < ST; ++st)
                acc[ft][st] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfrag[ft], dzf[st], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const unsigned char* Rw = smem + (wave * 2 + buf) * R_BYTES;
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            const int row = st * 16 + s16;
            const bool live = m0 + row < a.M;
            const int rsw = (row >> 2) & 3;
            uint32_t d[4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                float rv[4];
                f8_unpack4(*(const uint32_t*)(Rw + row * 64 + ((ft ^ rsw) << 4) + 4 * q4), rv);
                float y[4] = {acc[ft][st][0], acc[ft][st][1], acc[ft][st][2], acc[ft][st][3]};
                if (a.dp_thresh != 0) {
                    const uint32_t col = (uint32_t)(f0 + ft * 16 + 4 * q4);
                    const uint32_t m = (uint32_t)(m0 + row);
                    const uint32_t p0 = dropout_pair(dkey, m, (uint32_t)F, col);
                    const uint32_t p1 = dropout_pair(dkey, m, (uint32_t)F, col + 2);
                    y[0] *= dropout_scale(p0, 0, a.dp_thresh, a.dp_inv_keep);
                    y[1] *= dropout_scale(p0, 1, a.dp_thresh, a.dp_inv_keep);
                    y[2] *= dropout_scale(p1, 0, a.dp_thresh, a.dp_inv_keep);
                    y[3] *= dropout_scale(p1, 1, a.dp_thresh, a.dp_inv_keep);
                }
                if constexpr (PASS == 1) {
                    const float4 ca = cfa[ft], cb = cfb[ft], cz = cfz[ft];
                    y[0] = rv[0] > 0.f ? fmaf(ca.x, y[0], fmaf(cb.x, rv[0], cz.x)) : 0.f;
                    y[1] = rv[1] > 0.f ? fmaf(ca.y, y[1], fmaf(cb.y, rv[1], cz.y)) : 0.f;
                    y[2] = rv[2] > 0.f ? fmaf(ca.z, y[2], fmaf(cb.z, rv[2], cz.z)) : 0.f;
                    y[3] = rv[3] > 0.f ? fmaf(ca.w, y[3], fmaf(cb.w, rv[3], cz.w)) : 0.f;
                    amax = fmaxf(fmaxf(amax, fabsf(y[0])), fabsf(y[1]));
                    amax = fmaxf(fmaxf(amax, fabsf(y[2])), fabsf(y[3]));
                    d[ft] = f8_pack4_e5m2(y[0], y[1], y[2], y[3]);
                    f8_unpack4_e5m2(d[ft], y);                       // (column sums of the values as stored)
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float w = live ? y[e] : 0.f;
                    s1[ft * 4 + e] += w;
                    if constexpr (PASS == 0) s2[ft * 4 + e] = fmaf(w, rv[e], s2[ft * 4 + e]);
                }
            }
            if constexpr (PASS == 1) {
                { const auto x = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false); d[0] = x[0]; d[2] = x[1]; }
                { const auto x = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false); d[1] = x[0]; d[3] = x[1]; }
                { const auto x = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false); d[0] = x[0]; d[1] = x[1]; }
                { const auto x = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false); d[2] = x[0]; d[3] = x[1]; }
                const u32x4_t c = {d[0], d[1], d[2], d[3]};
                store_b128_settled(c, c_rsrc, c_lane, (uint32_t)((m0 + st * 16) * F), 0);
            }
        }
#pragma unroll
        for (int st = 0; st < ST; ++st) dzf[st] = dzn[st];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float u1 = f8_exp2i(-e_o), u2 = f8_exp2i(-e_r);
    float r1[2], r2[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        float v1[8], v2[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) { v1[p] = s1[hh * 8 + p]; v2[p] = PASS == 0 ? s2[hh * 8 + p] : 0.f; }
        r1[hh] = row16_fold8(v1, lane) * u1;
        r2[hh] = PASS == 0 ? row16_fold8(v2, lane) * u2 : 0.f;
    }
    if (s16 < 8) {
        const int64_t prow = (int64_t)wkr * 8 + xcd;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int idx = hh * 8 + s16, f = f0 + (idx >> 2) * 16 + 4 * q4 + (idx & 3);
            if constexpr (PASS == 0) {
                a.partials[(prow * 2 + 0) * F + f] = r1[hh];
                a.partials[(prow * 2 + 1) * F + f] = r2[hh];
            } else {
                a.partials[prow * F + f] = r1[hh];
            }
        }
    }
    if constexpr (PASS == 1) f8_atomic_amax(&a.st->amax[a.t_out], amax);
}

template <int PASS>
static inline hipError_t launch_proj_dgrad8(const Proj8Args& a, hipStream_t st, int* stat_rows) {
    if (a.K < 16 || (a.K & 3) || !a.R || (PASS == 1 && !a.coef) || (uint64_t)a.M * 512 >= 0xFFF00000ull) return hipErrorInvalidValue;
    const int blocks = PROJ_DGRAD_BLOCKS, nwk = (blocks >> 3) / 2;
    const int64_t tiles = (a.M + 31) / 32, workers = (int64_t)nwk * 8;
    if (stat_rows) *stat_rows = (int)(tiles < workers ? tiles : workers);
    hipLaunchKernelGGL(proj_dgrad8_kernel<PASS>, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

// test aid (cp_debug_activation): u = dropout(BatchNorm(r)) in f32 from a stored e4m3 activation, the mask of the forward pass
__global__ void debug_bn_dropout8_f32_kernel(const uint8_t* __restrict__ r, const float* __restrict__ stats, float* __restrict__ out,
                                             int64_t rows, int C, uint32_t thresh, uint32_t key, float inv_keep, const uint32_t* __restrict__ salt,
                                             const Fp8State* __restrict__ st, int t_in) {
    if (salt) key ^= *salt;
    const float d = f8_exp2i(-st->e[t_in]);
    const int64_t n = rows * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t m = (uint32_t)(i / C), f = (uint32_t)(i % C);
        const uint32_t pr = dropout_pair(key, m, (uint32_t)C, f & ~1u);
        const float v = __builtin_amdgcn_cvt_f32_fp8((int)r[i], 0) * d;
        out[i] = fmaf(v, stats[2 * C + f], stats[3 * C + f]) * dropout_scale(pr, (int)(f & 1u), thresh, inv_keep);
    }
}

// e5m2 tensor -> bf16 in true units (the gradient tap of the tests)
__global__ __launch_bounds__(256) void dequant5_bf16_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ out, int64_t n4,
                                                            const Fp8State* __restrict__ st, int t) {
    const float d = f8_exp2i(-st->e[t]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float v[4];
        f8_unpack4_e5m2(*(const uint32_t*)(in + i * 4), v);
        *(uint2*)(out + i * 4) = make_uint2(pack2bf(v[0] * d, v[1] * d), pack2bf(v[2] * d, v[3] * d));
    }
}
