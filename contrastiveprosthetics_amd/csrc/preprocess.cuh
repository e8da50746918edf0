// Raw-sEMG preprocessing on the device: what the reference does on the host, once,
// with SciPy for every (subject, stimulus, repetition) slice of a Ninapro recording --
//   code/load.py:102-109  gain 2^10, 4th-order Butterworth band-pass 20-450 Hz (scipy.signal.lfilter),
//                         moving RMS over 11 samples (scipy.ndimage.uniform_filter1d), keep samples time_mask
//   code/utils.py:79-135, code/load.py:141-148  normalisation statistics over the training slices, normalise
// One thread per (segment, channel) runs the two recurrences along time; 135,792 independent series for the whole
// data set.  The arithmetic follows SciPy's evaluation order and NumPy's dtype rules at every rounding point
// (DESIGN.md 7d lists them), with floating-point contraction switched off, so the kept samples
// are bit-identical to the reference's.
#pragma once
#include "common.cuh"

constexpr int PP_C = 12;            // channels
constexpr int PP_MAXCOEF = 17;      // up to an 8th-order band-pass
constexpr int PP_MAXOUT = 256;      // kept samples per segment
constexpr int PP_MAXWIN = 32;       // RMS window

struct PreprocArgs {
    const float* raw;               // [S][L][12]
    float* out;                     // [S][n_out][12]
    int64_t S;
    int L, n_out, n_coef, win;
    float gain;
    double b[PP_MAXCOEF], a[PP_MAXCOEF];          // normalised (a[0] == 1)
    short t_sorted[PP_MAXOUT], slot_sorted[PP_MAXOUT];   // kept samples sorted by time (a time may repeat)
};

// NB / WIN > 0: coefficient count and RMS window known at compile time (the reference's 9 and 11): the filter state
// and the window of squares stay in registers with static indices.  0 = take them from the arguments (any filter).
template <int NB, int WIN>
__global__ __launch_bounds__(256) void preprocess_kernel(PreprocArgs p) {
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.S * PP_C) return;
    const int64_t s = idx / PP_C;
    const int c = (int)(idx % PP_C);
    const float* x = p.raw + s * (int64_t)p.L * PP_C + c;
    float* o = p.out + s * (int64_t)p.n_out * PP_C + c;
    const int nb = NB > 0 ? NB : p.n_coef, win = WIN > 0 ? WIN : p.win, half = win / 2;
    constexpr int ZN = NB > 0 ? NB - 1 : PP_MAXCOEF - 1;
    constexpr int RN = WIN > 0 ? WIN : PP_MAXWIN;
    double z[ZN];
#pragma unroll
    for (int i = 0; i < ZN; ++i) z[i] = 0.0;
    float ring[RN + 1];             // ring[k] = squared sample of k steps ago
#pragma unroll
    for (int i = 0; i <= RN; ++i) ring[i] = 0.f;
    double tmp = 0.0;
    float sq0 = 0.f;
    int q = 0;                      // next entry of the sorted keep list
    const double dwin = (double)win;
    constexpr int PF = 16;          // samples fetched ahead: the recurrences are serial, the loads need not be
    float xbuf[PF];
    for (int t = 0; t < p.L; ++t) {
        if (t % PF == 0) {
#pragma unroll
            for (int k = 0; k < PF; ++k) xbuf[k] = (t + k < p.L) ? x[(int64_t)(t + k) * PP_C] : 0.f;
        }
        float xsel = xbuf[0];
#pragma unroll
        for (int k = 1; k < PF; ++k) xsel = (t % PF == k) ? xbuf[k] : xsel;
        const float xin = xsel * p.gain;                                 // float32 product (emg_ * 2**10)
        const double xt = (double)xin;
        // scipy.signal.lfilter, direct form II transposed, float64
        const double y = z[0] + p.b[0] * xt;
#pragma unroll
        for (int i = 0; i < ZN - 1; ++i)
            if (i < nb - 2) z[i] = (z[i + 1] + xt * p.b[i + 1]) - y * p.a[i + 1];
        z[nb - 2] = xt * p.b[nb - 1] - y * p.a[nb - 1];
        const float y32 = (float)y;                                      // utils.filter writes back into float32
        const float sq = y32 * y32;                                      // np.square, float32
#pragma unroll
        for (int i = RN; i > 0; --i) ring[i] = ring[i - 1];
        ring[0] = sq;
        if (t == 0) sq0 = sq;
        // scipy.ndimage.uniform_filter1d(mode='nearest'): running float64 sum over the edge-extended line
        const int lead = win - 1 - half;                                 // samples of the first window that lie ahead
        if (t <= lead) {
            if (t == 0) {
                for (int k = 0; k <= half; ++k) tmp += (double)sq;       // `half` edge copies + the sample itself
            } else {
                tmp += (double)sq;
            }
        } else {
            const int l = t - lead;                                      // output position whose window ends at t
            const int back = l - 1 - half;                               // sample that leaves the window (clamped at 0)
            const float leaving = back <= 0 ? sq0 : ring[win];           // t - back == win: the sample of `win` steps ago
            tmp += (double)sq - (double)leaving;
            const int i = l - half;                                      // index after moving_rms's [edge:-edge] slice
            if (i >= 0) {
                const float r = sqrtf((float)(tmp / dwin));              // float32 mean, float32 sqrt
                while (q < p.n_out && p.t_sorted[q] == i) {
                    o[(int64_t)p.slot_sorted[q] * PP_C] = r;
                    ++q;
                }
            }
        }
    }
}

// per-segment channel means (RunningStats.push: X.mean(0)) in float64: seg_mean[S][12]
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ seg, int64_t S, int n_out, double* __restrict__ seg_mean) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * PP_C) return;
    const float* x = seg + (idx / PP_C) * (int64_t)n_out * PP_C + idx % PP_C;
    double s = 0.0;
    for (int i = 0; i < n_out; ++i) s += (double)x[(int64_t)i * PP_C];
    seg_mean[idx] = s / n_out;
}

// mean and sample standard deviation (n-1) of the selected segments' means, per channel; complete: averaged over
// channels as RunningStats(complete=True) does (mean of means, sqrt of the mean variance).  One block of 252
// threads = 21 row lanes x 12 channels (consecutive threads read consecutive doubles); two passes in float64.
__global__ __launch_bounds__(256) void emg_stats_kernel(const double* __restrict__ seg_mean, const uint8_t* __restrict__ use, int64_t S,
                                                        int complete, float* __restrict__ mean_std /* [2][12] */) {
    constexpr int LANES = 21;
    __shared__ double red[LANES][PP_C];
    __shared__ double res[3][PP_C];          // count, mean, variance
    const int tid = threadIdx.x, c = tid % PP_C, lane = tid / PP_C;
    const bool on = tid < LANES * PP_C;
    double cnt = 0, a = 0;
    if (on)
        for (int64_t s = lane; s < S; s += LANES)
            if (!use || use[s]) { cnt += 1.0; a += seg_mean[s * PP_C + c]; }
    if (on) red[lane][c] = a;
    __syncthreads();
    if (tid < PP_C) { double t = 0; for (int l = 0; l < LANES; ++l) t += red[l][tid]; res[1][tid] = t; }
    __syncthreads();
    if (on) red[lane][c] = cnt;
    __syncthreads();
    if (tid < PP_C) { double t = 0; for (int l = 0; l < LANES; ++l) t += red[l][tid]; res[0][tid] = t; res[1][tid] /= t; }
    __syncthreads();
    const double mean = res[1][c];
    double v = 0;
    if (on)
        for (int64_t s = lane; s < S; s += LANES)
            if (!use || use[s]) { const double d = seg_mean[s * PP_C + c] - mean; v += d * d; }
    if (on) red[lane][c] = v;
    __syncthreads();
    if (tid < PP_C) { double t = 0; for (int l = 0; l < LANES; ++l) t += red[l][tid]; res[2][tid] = t / (res[0][tid] - 1.0); }
    __syncthreads();
    if (tid < PP_C) {
        double m = res[1][tid], var = res[2][tid];
        if (complete) {
            m = 0; var = 0;
            for (int k = 0; k < PP_C; ++k) { m += res[1][k]; var += res[2][k]; }
            m /= PP_C; var /= PP_C;
        }
        mean_std[tid] = (float)m;
        mean_std[PP_C + tid] = (float)sqrt(var);
    }
}

// RunningStats.normalize: (X - mean) / std in float32, in place; rows of 12 channels
__global__ __launch_bounds__(256) void emg_normalize_kernel(float* __restrict__ seg, int64_t n, const float* __restrict__ mean_std) {
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % PP_C);
        seg[i] = (seg[i] - mean_std[c]) / mean_std[PP_C + c];
    }
}
