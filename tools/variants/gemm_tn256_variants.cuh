// Tools-only kernels (build/libcpnative_variants.so, make -C contrastiveprosthetics_amd/csrc variants): measured and superseded,
// kept for A/B runs (tools/ab_env.sh, tools/ws_bench.py, tools/gemm_bench.py).  Included by csrc/gemm_tn256.cuh under -DCP_VARIANTS only;
// the product library does not contain them.  weight-gradient variants measured and not faster (16x16x32 form, one wave per SIMD)

// FOUR waves, one per SIMD, each a 128 x 128 piece of the block's 256 x 256 tile (256 accumulator registers): 8 fragments feed 16
// MFMAs per 16-row k step instead of 6 feeding 8 -- a third less LDS read traffic, which in the 8-wave kernel keeps the LDS port as busy
// as the matrix pipe (96 KiB of fragment reads + 32 KiB of DMA writes per 32-row step against 1,024 MFMA cycles per SIMD).  A lone wave
// per SIMD has nothing to cover its LDS latency with, so the fragments of step i+1 are loaded during the MFMAs of step i: the step barrier
// sits in the middle of a step, behind the first k half.  Measured in the step ($CPNATIVE_TN_W4, alternating runs): 148 us per launch
// against 140 for the 8-wave kernel -- parity-green, not faster, not the default.
__global__ __launch_bounds__(256, 1) void gemm_tn256w4_kernel(GemmTN256Args a) {
    constexpr int OP_BYTES = 32 * 512;
    constexpr int STAGE = 2 * OP_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TN256_STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;
    const int tiles_q = a.Q / 256, ntiles = (a.P / 256) * tiles_q;
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    const int per_problem = ntiles * ((a.splits + 7) / 8);
    const bool second = j >= per_problem;
    if (second) j -= per_problem;
    const bf16_t* __restrict__ Xg = second ? a.X2 : a.X;
    const bf16_t* __restrict__ Yg = second ? a.Y2 : a.Y;
    const int tile = j % ntiles;
    const int split = (j / ntiles) * 8 + xcd;
    if (split >= a.splits) return;
    const int p0 = (tile / tiles_q) * 256, q0 = (tile % tiles_q) * 256;
    const int64_t mb = (int64_t)split * a.rows_per_split;
    int64_t me = mb + a.rows_per_split;
    if (me > a.M) me = a.M;
    const int nsteps = (int)((me - mb + 31) / 32);

    // staging as in gemm_tn256_kernel, 4 LDS-DMA instructions per operand, wave and stage
    const int srow_l = lane >> 5;
    const int pb = (lane & 31) >> 2, sub = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    auto stage = [&](int slot, int step) {
        const int64_t ms = mb + (int64_t)step * 32;
        const uint32_t Xs = lds0 + slot * STAGE, Ys = Xs + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int inst = wave * 4 + i;             // 0..15
            const int row = inst * 2 + srow_l;
            int64_t m = ms + row;
            if (m >= me) m = me - 1;
            const int colb = (((pb ^ (row & 3)) << 6) | (sub << 4)) >> 1;
            glds16(Xg + m * a.ldx + p0 + colb, Xs + inst * 1024);
            glds16(Yg + m * a.ldy + q0 + colb, Ys + inst * 1024);
        }
    };
    uint4 fx[2][4], fy[2][4];                            // [k half][tile]
    auto load_frags = [&](int step, int ks, uint4 (&x)[4], uint4 (&y)[4]) {
        const unsigned char* Xs = smem + (step % TN256_STAGES) * STAGE;
        const unsigned char* Ys = Xs + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = tn256_frag(Xs, ks * 16, wp * 128 + i * 32, lane);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) y[jj] = tn256_frag(Ys, ks * 16, wq * 128 + jj * 32, lane);
        const int valid = (int)(me - (mb + (int64_t)step * 32));
        if (valid < 32) {
            const int base = ks * 16 + 8 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t* w = (uint32_t*)&x[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = (base + 2 * e) < valid ? 0xFFFFu : 0u;
                    const uint32_t hi = (base + 2 * e + 1) < valid ? 0xFFFF0000u : 0u;
                    w[e] &= (lo | hi);
                }
            }
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][jj][g] = 0.f;

    // prologue: stages 0..2 requested; stage 0 awaited, its fragments loaded
#pragma unroll
    for (int s = 0; s < TN256_STAGES - 1; ++s)
        if (s < nsteps) stage(s, s);
    {
        const int ahead = nsteps - 1 < 2 ? nsteps - 1 : 2;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    load_frags(0, 0, fx[0], fy[0]);
    load_frags(0, 1, fx[1], fy[1]);
    for (int step = 0; step < nsteps; ++step) {
        // first k half
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mma_chunk<bf16_t>(fx[0][i], fy[0][jj], acc[i][jj]);
        __builtin_amdgcn_s_setprio(0);
        const bool more = step + 1 < nsteps;
        if (more) {
            // stage step+1 landed for every wave (step+2 may stay in flight); every wave is done reading stage step-1
            if (step + 2 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step + TN256_STAGES - 1 < nsteps) stage((step + TN256_STAGES - 1) % TN256_STAGES, step + TN256_STAGES - 1);
            load_frags(step + 1, 0, fx[0], fy[0]);
        }
        // second k half
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) mma_chunk<bf16_t>(fx[1][i], fy[1][jj], acc[i][jj]);
        __builtin_amdgcn_s_setprio(0);
        if (more) load_frags(step + 1, 1, fx[1], fy[1]);
    }

    float* slab = (second ? a.slabs2 : a.slabs) + (int64_t)split * a.P * a.Q;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int q = q0 + wq * 128 + jj * 32 + r;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int p = p0 + wp * 128 + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                slab[(int64_t)p * a.Q + q] = acc[i][jj][g];
            }
        }
}

