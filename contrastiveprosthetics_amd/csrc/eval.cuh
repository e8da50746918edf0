// Evaluation post-processing on the device: class-subset prediction,
// prefix majority vote for MANY subsets in one launch, and the confusion matrix.
//   reference: code/models.py:146-163 (argmax, prefix mode, accuracy curve), code/results.py:24-64 (y_pred,
//   voting, confusion matrix), README.md:11-19 (the user picks a subset of the 41 classes at test time).
// Integer outputs only (hit counts, class ids): results do not depend on launch geometry.
#pragma once
#include "common.cuh"

constexpr int SV_T = 41;            // classes
constexpr int SV_MPB = 8;           // subsets handled by one block (round 4: 32 made 2 x B blocks of ~40k serial instructions per thread --
                                    // 257 us for 64 subsets of 160 groups x 25 samples, most of the chip idle)
constexpr int SV_VMAX = 64;         // samples per group supported
constexpr int SV_TILE = SV_T * SV_T;

struct __attribute__((packed, aligned(4))) SvF4 { float x, y, z, w; };

struct SubsetVoteArgs {
    const float* logits;            // [B*V][41][41], group g = b*V + v
    const int64_t* labels;          // [41] = labels[:tasks]
    const uint8_t* masks;           // [n_masks][41]
    unsigned long long* correct;    // [n_masks][V], zeroed by the caller
    int32_t* y_pred;                // [n_masks][B][41] or nullptr
    int64_t B, n_masks;
    int V;
};

// grid (ceil(n_masks / SV_MPB), B), 256 threads, dynamic LDS = sv_lds_bytes(V).
// Phase 1: thread per (sample, row) of the group (V x 41 of them, a few per thread): its 41 logits come straight from global memory into
// registers (41 independent loads; the tiles are L2-resident, every block of the group reads them) and the arg-max over the member
// columns of each of the block's subsets (first maximum wins) goes to preds[m][v][t] (one byte each).  No logits tile in LDS (round 3 staged
// four samples at a time through 27 KB of LDS: two blocks per CU, every LDS latency of phase 2 exposed -- 258 us for 64 subsets of
// 160 x 25 samples; round 4: 19 KB, eight blocks per CU).
// Phase 2: thread per (subset, row): running histogram of the V predictions, mode with ties to the
// smallest class id (torch.mode), one hit per prefix length where the mode equals the row's label.
static inline size_t sv_lds_bytes(int V) {
    return (size_t)SV_MPB * 8 + (size_t)SV_MPB * V * 4 + (size_t)SV_MPB * V * SV_T + 256 * SV_T;
}

__global__ __launch_bounds__(256) void subset_vote_kernel(SubsetVoteArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sv_smem[];
    unsigned long long* bits = (unsigned long long*)sv_smem;                       // [SV_MPB]
    int* hits = (int*)(bits + SV_MPB);                                             // [SV_MPB][V]
    unsigned char* preds = (unsigned char*)(hits + SV_MPB * a.V);                  // [SV_MPB][V][41]
    unsigned char* hist = preds + (size_t)SV_MPB * a.V * SV_T;                     // [256][41]
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * SV_MPB;
    const int mcount = (int)((a.n_masks - m0 < SV_MPB) ? (a.n_masks - m0) : SV_MPB);
    const int V = a.V;

    if (tid < SV_MPB) {
        unsigned long long w = 0;
        if (tid < mcount)
            for (int c = 0; c < SV_T; ++c) w |= (unsigned long long)(a.masks[(m0 + tid) * SV_T + c] != 0) << c;
        bits[tid] = w;
    }
    for (int i = tid; i < SV_MPB * V; i += 256) hits[i] = 0;
    __syncthreads();

    const float* grp = a.logits + b * V * (int64_t)SV_TILE;
    for (int item = tid; item < V * SV_T; item += 256) {
        const int v = item / SV_T, t = item % SV_T;
        // a row is 164 bytes at a 4-byte-aligned address: ten 16-byte loads (legal at 4-byte alignment in the unaligned access mode HSA runs in)
        // and one dword -- 41 dword loads per lane, each touching 64 different cache lines per wave, were most of the kernel's time
        float x[SV_T];
        const SvF4* row4 = (const SvF4*)(grp + (int64_t)item * SV_T);
#pragma unroll
        for (int c4 = 0; c4 < SV_T / 4; ++c4) {
            const SvF4 q = row4[c4];
            x[4 * c4] = q.x; x[4 * c4 + 1] = q.y; x[4 * c4 + 2] = q.z; x[4 * c4 + 3] = q.w;
        }
        x[SV_T - 1] = grp[(int64_t)item * SV_T + SV_T - 1];
        for (int m = 0; m < mcount; ++m) {
            // the subset is the same for every lane: its 41 bits sit in two SGPRs and each column is a scalar branch, so only the
            // MEMBER columns cost vector instructions (a compare and two selects)
            const unsigned long long wv = bits[m];
            const uint32_t wlo = __builtin_amdgcn_readfirstlane((uint32_t)wv), whi = __builtin_amdgcn_readfirstlane((uint32_t)(wv >> 32));
            float best = -__builtin_inff();
            int bi = -1;
#pragma unroll
            for (int c = 0; c < SV_T; ++c) {
                if ((c < 32 ? (wlo >> c) : (whi >> (c - 32))) & 1u) {
                    const bool take = bi < 0 || x[c] > best;            // (first maximum wins; a NaN never replaces a number)
                    best = take ? x[c] : best;
                    bi = take ? c : bi;
                }
            }
            const bool member = (t < 32 ? (wlo >> t) : (whi >> (t - 32))) & 1u;
            preds[((size_t)m * V + v) * SV_T + t] = (unsigned char)(member ? bi : 255);
        }
    }
    __syncthreads();

    unsigned char* h = hist + tid * SV_T;
    for (int p = tid; p < mcount * SV_T; p += 256) {
        const int m = p / SV_T, row = p % SV_T;
        const bool member = (bits[m] >> row) & 1;
        int best = -1;
        if (member) {
            for (int c = 0; c < SV_T; ++c) h[c] = 0;
            const int y = (int)a.labels[row];
            int bestc = 0;
            best = 0;
            for (int w = 0; w < V; ++w) {
                const int q = preds[((size_t)m * V + w) * SV_T + row];
                const int c = ++h[q];
                if (c > bestc || (c == bestc && q < best)) { best = q; bestc = c; }
                if (best == y) atomicAdd(&hits[m * V + w], 1);
            }
        }
        if (a.y_pred) a.y_pred[((m0 + m) * a.B + b) * SV_T + row] = best;
    }
    __syncthreads();
    for (int i = tid; i < mcount * V; i += 256)
        if (hits[i]) atomicAdd(&a.correct[(m0 + i / V) * V + i % V], (unsigned long long)hits[i]);
}

// counts[y_true][y_pred] += 1 over n_groups x 41 predictions, y_true = labels[row]; y_pred < 0 skipped
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ y_pred, const int64_t* __restrict__ labels,
                                                        int64_t n, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int local[SV_T * SV_T];
    for (int i = threadIdx.x; i < SV_T * SV_T; i += 256) local[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int p = y_pred[i];
        const int y = (int)labels[i % SV_T];
        if (p >= 0 && p < SV_T && y >= 0 && y < SV_T) atomicAdd(&local[y * SV_T + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SV_T * SV_T; i += 256)
        if (local[i]) atomicAdd(&counts[i], (unsigned long long)local[i]);
}
