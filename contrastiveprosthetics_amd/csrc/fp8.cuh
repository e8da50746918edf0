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
// commented out).  Parity of this path is "unpinned by construction": tests report its distance to the f32 oracle and to the
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
__device__ __forceinline__ void f8_atomic_amax(uint32_t* dst, float m) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(dst, __float_as_uint(m));
}

// ------------------------------------------------------------------------------------------------------------------------
// fold the previous BatchNorm's affine AND the input tensor's scale into a Linear, quantise each output row to e4m3 with its own
// power-of-two scale:  y * 2^eo = sum_k Wq[j][k] * xq[k] * 2^(eo - ej - ei) + b'[j] * 2^eo,   Wq = e4m3(W[j][k] s[k] 2^ej)
//   out_w: [F][K] bytes (fc1: in the internal order k' = w*64 + c);  out_sc: [F] E8M0 bytes 127 + eo - ej - ei for the MFMA's
//   scale operand;  out_b: [F] f32 bias in OUTPUT units (the accumulators start at it).  One block per output row.
//   t_in / t_out: scale-table ids of the layer's input and output tensors (t_out < 0: eo = 0, f32 output).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_linear8_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                           const float* __restrict__ s, const float* __restrict__ t,
                                                           uint8_t* __restrict__ out_w, uint8_t* __restrict__ out_sc,
                                                           float* __restrict__ out_b, int K, int mode, const Fp8State* __restrict__ st,
                                                           int t_in, int t_out) {
    __shared__ float red[2][4];
    const int j = blockIdx.x, tid = threadIdx.x;
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// K = 512: 64-row tiles, three buffers.  K = 768 (fc1; 192 weight registers): 32-row tiles -- with 64 rows the two accumulator
// sets, the fragments and the running sums no longer fit beside the weights (56 spilled registers) -- and four buffers, so that
// a fetch still has two whole tiles to land.
template <int K> struct Ws8Cfg;
template <> struct Ws8Cfg<512> { static constexpr int RT = 64, NBUF = 3; };
template <> struct Ws8Cfg<768> { static constexpr int RT = 32, NBUF = 4; };

template <int K>
__global__ __launch_bounds__(256, 1) void gemm_ws8_kernel(Ws8Args a) {
    constexpr int KB = K / 128, RT = Ws8Cfg<K>::RT, ST = RT / 16, NBUF = Ws8Cfg<K>::NBUF, AHEAD = NBUF - 1;
    constexpr int TILE_BYTES = RT * K, UPW = TILE_BYTES / 1024 / 4, CPR = K / 16;
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
        bufl16_lds(a_rsrc, fsrc[q], tile_soff, lds0 + buf * TILE_BYTES + (wave * UPW + q) * 1024);
    };
    auto row0 = [&](int ti) -> int64_t { return ((int64_t)ti * stride + first) * RT; };
    // (no such tile: an offset past the end of the buffer -- the fetches return zeros into an idle buffer, and the counted waits
    //  below see the same number of operations in every tile)
    auto tile_soff = [&](int ti) -> uint32_t { return ti < ntile ? (uint32_t)(row0(ti) * K) : 0xFFF00000u; };

    float s1[16], s2[16], amax = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) s1[p] = s2[p] = 0.f;
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)((int64_t)a.M * a.F), 0x00020000);
    const uint32_t c_lane = (uint32_t)(s16 * a.F + f0 + q4 * 16);

    // epilogue of sample tile st of a finished tile: clamp(relu), sums, maximum, convert, transpose over the sample's 4 lanes, store
    auto epi_st = [&](f32x4_t (&old)[4][ST], int st, uint32_t s_old, bool live) {
        uint32_t d[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __builtin_amdgcn_fmed3f(old[ft][st][e], 0.f, F8_E4M3_MAX);
                const float w = live ? v[e] : 0.f;
                s1[ft * 4 + e] += w;
                s2[ft * 4 + e] = fmaf(w, w, s2[ft * 4 + e]);
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

    auto load_frag = [&](const unsigned char* At, int kb, int st) -> i32x8_t {
        const int c0 = kb * 8 + q4 * 2;
        const i32x4_t lo = *(const i32x4_t*)(At + st * 16 * K + ((c0 ^ s16) << 4));
        const i32x4_t hi = *(const i32x4_t*)(At + st * 16 * K + (((c0 + 1) ^ s16) << 4));
        return (i32x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // K loop of tile ti into acc; the fetches of tile ti + AHEAD go out along the way; WITH_EPI: the previous tile's epilogue runs
    // behind the loop (gemm_ws16_kernel: vector instructions beside a busy matrix pipe cost more than their own time)
    auto step = [&](f32x4_t (&acc)[4][ST], f32x4_t (&old)[4][ST], int ti, auto with_epi_tag, int64_t m_old) {
        constexpr bool WITH_EPI = decltype(with_epi_tag)::value;
        const int buf = ti % NBUF;
        const uint32_t next_soff = tile_soff(ti + AHEAD);
        const int nbuf = (ti + AHEAD) % NBUF;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const float4 b4 = *(const float4*)(bias_s + f0 + ft * 16 + 4 * q4);
            const f32x4_t b0 = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int st = 0; st < ST; ++st) acc[ft][st] = b0;
        }
        const unsigned char* At = smem + buf * TILE_BYTES + s16 * K;
        i32x8_t fa[PF2 ? 2 : 1][ST];
#pragma unroll
        for (int st = 0; st < ST; ++st) fa[0][st] = load_frag(At, 0, st);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            constexpr int dummy = 0; (void)dummy;
            if (PF2 && kb + 1 < KB) {
#pragma unroll
                for (int st = 0; st < ST; ++st) fa[(kb + 1) & 1][st] = load_frag(At, kb + 1, st);
            }
#pragma unroll
            for (int st = 0; st < ST; ++st) {
                if (kb * ST + st < UPW) fetch_unit(next_soff, nbuf, kb * ST + st);
                const i32x8_t& x = fa[PF2 ? (kb & 1) : 0][st];
                acc[0][st] = mx_mfma<0>(wreg[0][kb], x, acc[0][st], wscale);
                acc[1][st] = mx_mfma<1>(wreg[1][kb], x, acc[1][st], wscale);
                acc[2][st] = mx_mfma<2>(wreg[2][kb], x, acc[2][st], wscale);
                acc[3][st] = mx_mfma<3>(wreg[3][kb], x, acc[3][st], wscale);
                if (!PF2 && kb + 1 < KB) fa[0][st] = load_frag(At, kb + 1, st);
            }
        }
        if constexpr (WITH_EPI) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t s_old = (uint32_t)(m_old * a.F);
#pragma unroll
            for (int st = 0; st < ST; ++st) epi_st(old, st, s_old, true);
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile ti + 1 must have landed.  vmcnt retires in issue order (tools/vmcnt_order_probe.hip); younger than its fetches are the
        // fetches of tiles ti + 2 .. ti + AHEAD and the stores of the epilogues that ran since: min(AHEAD, ti) of them
        const int ne = ti < AHEAD ? ti : AHEAD;
        if (ne == 0) wait_vmcnt<(AHEAD - 1) * UPW>();
        else if (ne == 1) wait_vmcnt<(AHEAD - 1) * UPW + ST>();
        else if (ne == 2 || AHEAD == 2) wait_vmcnt<(AHEAD - 1) * UPW + 2 * ST>();
        else wait_vmcnt<(AHEAD - 1) * UPW + (AHEAD >= 3 ? 3 : 2) * ST>();
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
    if (s16 < 8) {
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
    hipLaunchKernelGGL(gemm_ws8_kernel<K>, dim3(256), dim3(256), 0, st, a);
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
    if (salt) key ^= *salt;
    const float din = f8_exp2i(-st->e[t_in]), dout = f8_exp2i(st->e[t_out]);
    const int cpr = C / 16, rpp = blockDim.x / cpr;
    const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
    const int f = cc * 16;
    float sc[16], sh[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { sc[e] = stats[2 * C + f + e] * din * dout; sh[e] = stats[3 * C + f + e] * dout; }
    float amax = 0.f;
    auto apply = [&](const uint4& in, int64_t m) {
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
            f8_unpack4(w[q], v);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const uint32_t pr = dropout_pair(key, (uint32_t)m, (uint32_t)C, (uint32_t)(f + 4 * q + e));
                v[e] = fmaf(v[e], sc[4 * q + e], sh[4 * q + e]) * dropout_scale(pr, 0, thresh, inv_keep);
                v[e + 1] = fmaf(v[e + 1], sc[4 * q + e + 1], sh[4 * q + e + 1]) * dropout_scale(pr, 1, thresh, inv_keep);
            }
            amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
            amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
            o[q] = f8_pack4(f8_clamp(v[0]), f8_clamp(v[1]), f8_clamp(v[2]), f8_clamp(v[3]));
        }
        *(uint4*)(u + m * C + f) = make_uint4(o[0], o[1], o[2], o[3]);
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
