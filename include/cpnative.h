/* cpnative -- C ABI of the MI355X-native contrastive sEMG training path.
 *
 * The reference (FibonacciDude/ContrastiveProsthetics) is pure Python/PyTorch and has no
 * FFI layer; its de-facto operator boundary is the Python class surface of `Model`,
 * `EMGNet`, `GLOVENet` (code/models.py), `TaskWrapper` (code/utils.py), `DB23`
 * (code/load.py) and the step in `train_loop` (code/train.py:95-108).  Every entry point
 * below names the reference call site it replaces.  The library is what a maintainer binds
 * with ctypes from those classes; `contrastiveprosthetics_amd/` is
 * exactly such a binding.
 *
 * Conventions: plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 * unless the name ends in `_host`.  `stream` is a hipStream_t passed as void*.  Every call
 * only enqueues work on `stream`: no allocation, no synchronisation, no host read-back.
 * Return value: 0 on success, otherwise a hipError_t (or CP_ERR_* below); the message is
 * available from cp_last_error().  The library is gfx950-only and has no CPU fallback.
 *
 * Activations inside the workspace are stored in `dtype` (f32 = parity path computed with
 * v_mfma_f32_32x32x2_f32, bf16 = throughput path computed with v_mfma_f32_32x32x16_bf16 and
 * f32 accumulation); parameters, gradients, statistics, z, logits and the loss are f32.
 */
#ifndef CPNATIVE_H
#define CPNATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP_VERSION 110            /* 0.1.1: per-call state (cp_config carries options, tile schedule, sync-BN hook, gradient tap) */
#define CP_F32 0
#define CP_BF16 1
#define CP_FP8 2                  /* e4m3 activations and fc weights on the block-scaled MFMA (BASELINE config 4); see cp_config.dtype */
#define CP_TASKS 41               /* code/constants.py:45-48 */
#define CP_EMG_DIM 12             /* code/constants.py:97 */
#define CP_D_E 16                 /* embedding width (code/train.py:183) */
#define CP_N_BN 9                 /* 2 x BatchNorm2d(64) + 7 x BatchNorm1d(512) */
#define CP_N_FC 7
#define CP_ERR_ARG 10001
#define CP_ERR_WORKSPACE 10002

/* Pointers to the trainable tensors of `Model` (values) or to their gradients.
 * Shapes and state_dict keys: SURVEY.md section 8b / code/models.py:248-315, 412-428. */
typedef struct cp_params {
    float* conv1_w;            /* emg_net.conv_emg.0.weight (64,1,3,3) */
    float* conv1_b;            /* emg_net.conv_emg.0.bias   (64)       */
    float* conv2_w;            /* emg_net.conv_emg.3.weight (64,64,3,3)*/
    float* conv2_b;            /* emg_net.conv_emg.3.bias   (64)       */
    float* fc_w[CP_N_FC];      /* emg_net.linear.{0,3,6,9,13,17,21}.weight (512,768 | 512,512) */
    float* fc_b[CP_N_FC];      /* ...bias (512) */
    float* bn_g[CP_N_BN];      /* BN gamma: conv_emg.{2,5}, linear.{2,5,8,11,15,19,23} */
    float* bn_b[CP_N_BN];      /* BN beta */
    float* last_w;             /* emg_net.last.0.weight (16,512), no bias */
    float* easy_w;             /* glove_net.easy.0.weight (16,41) */
    float* easy_b;             /* glove_net.easy.0.bias   (16)    */
} cp_params;

/* running statistics of the stock nn.BatchNorm (code/models.py:238-243); all NULL for AdaBN
 * (code/models.py:17-35: momentum 0, track_running_stats False). */
typedef struct cp_bn_buffers {
    float* running_mean[CP_N_BN];
    float* running_var[CP_N_BN];
} cp_bn_buffers;

/* Synchronised-BatchNorm hook (see cp_config.stats_allreduce below): adds one row of `count` floats over all ranks in place. */
typedef int (*cp_allreduce_fn)(void* user, void* row_dev, int64_t count, void* stream);

/* Test / measurement switches of one call (cp_config.options; all 0 in production).  They select orders or forms of the SAME
 * computation that tests compare.  The library has no process-wide switch and reads no environment variable. */
#define CP_OPT_UNFUSED_BN_BWD 1u  /* BatchNorm + ReLU backward as its own pass behind every data gradient (the f32 path's order, on the bf16 kernels) */
#define CP_OPT_UNPAIRED_WGRAD 2u  /* one weight-gradient launch per layer behind a dropout instead of paired launches */
#define CP_OPT_FP8_BRIDGE 4u      /* CP_FP8: expand the saved 8-bit tensors to bf16 and run the bf16 backward kernels */
#define CP_OPT_NO_SMALL 8u        /* batches of <= 64 groups on the large-batch kernels instead of the small-batch form */
#define CP_OPT_FP8_HEAD_F32 16u   /* CP_FP8: the head's logits from the f32 matrix instruction instead of the block-scaled 8-bit one */

/* How the persistent fc GEMM kernels hand their output tiles to the CUs (cp_config.tile_schedule).  CP_TILES_STATIC: each
 * workgroup owns a fixed list of tiles -- fastest when this process has the GPU to itself (weight-stationary kernels).
 * CP_TILES_DYNAMIC: workgroups draw tiles from per-XCD counters -- 2-4 % slower alone, but a launch that shares CUs with
 * another stream's or process's kernels (a packed sweep, collectives that stay resident for long) no longer waits for its
 * latest-starting workgroup (+35 % with 8-32 CUs held).  BatchNorm partial sums are grouped per sample tile in the dynamic
 * mode and per workgroup in the static one, so the two differ in the last bits; each is run-to-run reproducible.
 * (The dynamic schedule's tile counters live in a module-global device table with one slot per stream: the only device-side
 * state shared between engines, and two streams never share a slot.) */
enum { CP_TILES_STATIC = 0, CP_TILES_DYNAMIC = 1 };

/* Everything a call depends on besides its tensors.  The library keeps NO per-process state for the training path: two
 * engines (two cp_config / workspace pairs) in one process, on one or several streams, do not see each other.  The config of
 * a cp_encoder_backward call must equal that of the cp_encoder_forward call whose workspace it consumes (the library checks
 * the kernel path and the size and returns CP_ERR_ARG otherwise). */
typedef struct cp_config {
    int64_t n_windows;   /* rows through the encoder = groups * 41 (train: B*41, eval: B*41*25) */
    int32_t dtype;       /* CP_F32 | CP_BF16 | CP_FP8 (the first F8_STATE_BYTES = 1024 bytes of a CP_FP8 workspace hold the tensors' scales across
                          * steps: zero them once after allocating it; see cp_fp8_copy_state) */
    int32_t adabn;       /* 1: batch statistics in train AND eval (AdaBN); 0: stock BN */
    int32_t training;    /* 1: model.train()  (batch stats, dropout, running-stat update) */
    uint32_t step_state_lo; /* low / high half of the DEVICE address of a cp_step_state, or 0/0 (see below) */
    float dp_emg;        /* Dropout p after BN of fc4..fc7 (code/models.py:282-297) */
    float bn_momentum;   /* 0.1 */
    float bn_eps;        /* 1e-5 */
    uint32_t step_state_hi;
    uint64_t seed;       /* dropout stream = f(seed, step, layer, element) */
    uint64_t step;
    uint32_t options;    /* CP_OPT_* bits; 0 in production */
    int32_t tile_schedule;   /* CP_TILES_STATIC (0, default) | CP_TILES_DYNAMIC */
    /* ---- synchronised BatchNorm (opt-in: NULL = every rank normalises with its own shard's statistics, which is
     * the reference at B_local, code/models.py:17-35,238-243).  With a hook, every BatchNorm of the sEMG encoder takes its
     * batch statistics -- and, in the backward pass, the two sums of BatchNorm's data gradient -- over ALL ranks: the library
     * folds its partial sums into one row of `count` floats in the workspace and calls stats_allreduce(stats_user, row, count,
     * stream), which must add the rows of all ranks in place, ordered after the work already on `stream` and before what is
     * enqueued on it next (torch.distributed.all_reduce on that memory does exactly this).  stats_world = number of ranks (the
     * element count is scaled by it).  gamma / beta gradients stay this rank's part, as in torch.nn.SyncBatchNorm.  18 calls
     * per training step. */
    cp_allreduce_fn stats_allreduce;
    void* stats_user;
    int32_t stats_world;
    int32_t reserved0;
    /* ---- test aid: while grad_tap is non-NULL, cp_encoder_backward copies every intermediate gradient into it (stream-ordered):
     * 9 slots of n_windows x 768 elements of the compute dtype; slot L = 2..8: dL/d(pre-activation of fc layer L-1), i.e. after
     * BatchNorm + ReLU backward, n_windows x 512; slot 1: dL/d(conv2 pre-activation), slot 0: dL/d(BN1 output), both
     * n_windows x [12 positions][64 channels].  grad_tap_bytes = size of the buffer.  (Slot 0 is no tensor of the step since round 4 --
     * conv2's data gradient is consumed in the accumulators of conv2_dgrad_conv1_kernel -- so with a tap the call runs the stand-alone
     * data-gradient kernel once more to fill it; CP_FP8: slots 0 and 1 hold the bf16 expansion of the e5m2 gradient.) */
    void* grad_tap;
    size_t grad_tap_bytes;
    /* ---- a second stream for the work of cp_encoder_backward that nothing in the step waits for (round 4; all three NULL = one
     * stream, which is what contrastiveprosthetics_amd.engine passes by default: since the projection's and conv2's weight gradients
     * carry BatchNorm-backward sums they are on the critical path, and with only fc5..fc7's left to float one stream measured
     * faster -- DESIGN.md 7j).  The weight gradients of the layers behind a dropout (fc5..fc7) are not on the step's
     * critical path -- their BatchNorm-backward sums come from the data-gradient launches -- while ~40 latency-bound finaliser /
     * fold / reduction launches of the critical path leave most of the chip idle.  With aux_stream (a hipStream_t, ideally of
     * LOWER priority than `stream`) those weight-gradient launches and their slab reductions are enqueued there: aux_fork (a
     * hipEvent_t owned by the caller) is recorded on `stream` and waited for on aux_stream wherever a launch's inputs become
     * final, aux_join is recorded on aux_stream and waited for on `stream` before the call returns (and before fc_grads_ready is
     * recorded), so the caller sees the same stream-ordered semantics as without it.  Same kernels; fc5's and fc4's weight gradients are then summed over
     * 64 row splits each instead of 32 (alone instead of in one paired launch): equal to the rounding of an f32 sum, run-to-run exact.
     * Used by the large-batch 16- and 8-bit paths when dp_emg > 0; ignored elsewhere (f32, small batches, synchronised BatchNorm,
     * a gradient tap). */
    void* aux_stream;
    void* aux_fork;
    void* aux_join;
} cp_config;

/* Per-step values kept in DEVICE memory so that a whole training step can be captured in a HIP graph and replayed
 * (graphs bake kernel arguments; these are the arguments that change from step to step).  The host refreshes the
 * 32 bytes with one asynchronous copy before each replay.  When cp_config.step_state_{lo,hi} hold its address, the
 * dropout stream is f(seed, layer, element) ^ dp_salt (cfg->step is then taken as 0), and cp_l2_adam_step_graph
 * reads the bias corrections and learning rates from it. */
typedef struct cp_step_state {
    uint32_t dp_salt;     /* any function of the step index, e.g. a hash of it */
    float bc1, bc2;       /* 1 - beta1^t, 1 - beta2^t */
    float lr_emg, lr_glove;
    float pad[3];
} cp_step_state;

int cp_version(void);
const char* cp_last_error(void);

/* 1 in the tools-only build (make -C csrc variants), which also carries the superseded kernels of tools/variants/ with one
 * $CPNATIVE_<NAME> switch each; 0 in the product library. */
int cp_has_variants(void);

/* bytes of scratch needed by the calls below for up to `max_windows` encoder rows */
size_t cp_workspace_bytes(int64_t max_windows, int32_t dtype, float dp_emg);

/* TaskWrapper.__getitem__ + DB23.__getitem__/slice_batch + default_collate
 * (code/utils.py:51-64, code/load.py:256-273, code/train.py:86,95) in one launch.
 * table: DB23.EMG_use (table_rows,12) f32 (eval: the same memory viewed as (rows/25,25,12));
 * emg_rand: TaskWrapper.emg_rand (41,D) int64; perm: the B item indices of this batch;
 * x_out: (B,41,V,12) f32 -- the collated EMG tensor in encoder row order. */
int cp_gather_groups(const float* table, int64_t table_rows, const int64_t* emg_rand, int64_t D,
                     const int64_t* perm, int64_t B, int32_t V, float* x_out, void* stream);
/* cp_gather_groups reads a source row outside [0, table_rows) from row 0 (the reference would raise an IndexError at
 * code/load.py:262-266) and counts it; this copies the running count of such rows since the last reset into the DEVICE
 * word count_out (stream-ordered) and, with reset != 0, zeroes it afterwards.  Non-zero = emg_rand, V and the table
 * do not belong together. */
int cp_gather_oob_count(uint32_t* count_out, int32_t reset, void* stream);

/* EMGNet.forward (code/models.py:319-342): conv_emg -> linear -> last.
 * x (n_windows,12) f32, 16-byte aligned (a window's 12 values are read as three 16-byte loads; cp_gather_groups'
 * output and any torch allocation are); z_out (n_windows,16) f32 in the same row order (the regroup of
 * models.py:337-341 is a pure index map applied by cp_head).  Saves what backward needs in ws. */
int cp_encoder_forward(const cp_config* cfg, const cp_params* p, const cp_bn_buffers* bn,
                       const float* x, void* ws, size_t ws_bytes, float* z_out, void* stream);

/* Model.forward's normalise + bmm with GLOVENet.forward's one-hot Linear
 * (code/models.py:121-130, 457-465), Model.loss / contrastive_loopy_loss
 * (code/models.py:132-173, 198-208) and their gradient, fused.
 * labels (B*41) int64; n_groups = B*V; loss_correct[0] = loss, [1] = number of rows whose
 * argmax equals its label; pred (n_groups,41) int32; logits optional (n_groups,41,41) f32.
 * want_grad: also writes dL/dz into ws (consumed by cp_encoder_backward) and the class-encoder
 * gradients grads->easy_w / easy_b. */
int cp_head(const cp_config* cfg, const cp_params* p, const float* z, const int64_t* labels,
            int64_t n_groups, int32_t V, int32_t want_grad, void* ws, size_t ws_bytes,
            float* loss_correct, int32_t* pred, float* logits, cp_params* grads, void* stream);

/* ---- global negatives (an opt-in EXTENSION of the loss, no reference counterpart) ---------------------
 * The reference's class->EMG direction (code/models.py:136-147 on the transposed logits) is a softmax over the 41 windows
 * of ONE group.  With this extension the column of class k of group b ranges over its positive window and every window
 * of another class in the GLOBAL batch (all groups of all ranks):
 *     col[b,k] = -s[b,pos_k,k] + log( exp(s[b,pos_k,k]) + G[k] ),   G[k] = sum_{all windows n, class(n) != k} exp(s[n,k])
 * z_all (n_all_windows,16) f32: the z embeddings of the global batch in window order (the RCCL all-gather of every rank's
 * cp_encoder_forward output; at one rank, that output itself); labels (>= 41) int64: class of position t of a group.
 * gh_out: 128 device floats {G[64], H[64]} (41 used each; H[k] = sum over all groups of 1/(exp(pos) + G[k]) carries the
 * gradient into the negatives).  scratch: cp_global_negatives_scratch_floats(n_all_windows) device floats.
 * cp_head_gneg = cp_head with that table: row direction unchanged, column direction as above; gradients are those of
 * this rank's windows (the data-parallel gradient sum adds the ranks' parts).  One-hot class table, training batches. */
size_t cp_global_negatives_scratch_floats(int64_t n_all_windows);
int cp_global_negatives(const cp_params* p, const float* z_all, int64_t n_all_windows, const int64_t* labels,
                        float* scratch, float* gh_out, void* stream);
int cp_head_gneg(const cp_config* cfg, const cp_params* p, const float* z, const int64_t* labels,
                 int64_t n_groups, int32_t V, int32_t want_grad, void* ws, size_t ws_bytes,
                 float* loss_correct, int32_t* pred, float* logits, cp_params* grads, const float* gh, void* stream);
/* The same table WITHOUT moving z: the class table is replicated, so G and H are sums of per-rank terms and only two 41-float
 * vectors have to cross ranks.  cp_global_negatives_g: this rank's rows' part of G into gh[0..63]; the caller sums gh[0..63] over
 * the ranks (one all-reduce of 64 floats); cp_global_negatives_h: with the summed G in place, this rank's groups' part of H into
 * gh[64..127]; the caller sums that too.  scratch as above for n_local_windows, kept between the two calls (it holds the positives).
 * cp_global_negatives on the gathered rows == _g, sum, _h, sum on each rank's own rows (tests/test_gpu_global_batch.py). */
int cp_global_negatives_g(const cp_params* p, const float* z_local, int64_t n_local_windows, const int64_t* labels,
                          float* scratch, float* gh, void* stream);
int cp_global_negatives_h(int64_t n_local_windows, const int64_t* labels, float* scratch, float* gh, void* stream);


/* autograd of EMGNet (what loss.backward() does at code/train.py:105 for emg_net):
 * consumes dL/dz left in ws by cp_head, writes every emg_net gradient into `grads`. */
int cp_encoder_backward(const cp_config* cfg, const cp_params* p, const float* x, void* ws,
                        size_t ws_bytes, cp_params* grads, void* stream);
/* The same, for data-parallel training: `fc_grads_ready` (a hipEvent_t, or NULL) is recorded on `stream` as soon as every
 * gradient except the conv stack's (conv1, its BatchNorm, conv2, its BatchNorm -- 0.15 of the 8.1 MB) is final, about
 * 0.5 ms before the call's last kernel at 167,936 windows: the caller's all-reduce of that part (torch.distributed /
 * RCCL on another stream, after a wait on the event) runs beside the conv backward.  No reference counterpart: the
 * reference is single-process (code/train.py:105). */
int cp_encoder_backward_ev(const cp_config* cfg, const cp_params* p, const float* x, void* ws,
                           size_t ws_bytes, cp_params* grads, void* stream, void* fc_grads_ready);

/* eval majority vote (code/models.py:151-163): pred (B,V,41) -> curve (B,V) of prefix-mode
 * accuracies, y_pred (B,41) = mode over all V samples. */
int cp_vote(const int32_t* pred, const int64_t* labels, int64_t B, int32_t V, float* curve,
            int32_t* y_pred, void* stream);

/* Class-subset evaluation for MANY subsets in one launch.  The reference's product
 * use-case (README.md:11-19): at test time the user keeps a subset S of the 41 classes; only the EMG rows t in S
 * and the class-encoding columns c in S of every 41 x 41 logits tile take part,
 *     pred[b,v,t] = argmax_{c in S} logits[b*V+v, t, c]            (code/models.py:147, first maximum wins)
 * followed by the prefix majority vote of code/models.py:151-163 (torch.mode: ties -> smallest class id).
 * logits (B*V,41,41) f32 as Model.forward returns them in eval (what results.py:45 saves as logs.npy);
 * labels (41) int64 = labels[:tasks]; masks (n_masks,41) uint8 (non-zero = member);
 * correct (n_masks,V) int64 OVERWRITTEN with the number of (b, t in S) whose mode over the first w+1 samples
 * equals labels[t] (accuracy = correct / (B*|S|); Model.voting_raw lays w = 0..V-1 out over win = 1..249);
 * y_pred optional (n_masks,B,41) int32: mode over all V samples, -1 for rows outside S (results.py:51).
 * V <= 64. */
int cp_subset_vote(const float* logits, const int64_t* labels, int64_t B, int32_t V, const uint8_t* masks,
                   int64_t n_masks, int64_t* correct, int32_t* y_pred, void* stream);

/* sklearn.metrics.confusion_matrix(y_true, y_pred) of code/results.py:58 as counts:
 * counts (41,41) int64 += 1 at [labels[i % 41]][y_pred[i]] for i < n_groups*41; y_pred < 0 is skipped. */
int cp_confusion(const int32_t* y_pred, const int64_t* labels, int64_t n_groups, int64_t* counts, void* stream);

/* Raw-sEMG preprocessing = DB23.get_stim_rep after the slice (code/load.py:102-109) with
 * utils.filter / utils.rms (code/utils.py:137-156), for all segments at once.
 * raw (n_segments, seg_len, 12) f32 on the device: the first seg_len = 2000 + 2*5 samples of each
 * (stimulus, repetition) mask, as scipy.io.loadmat delivers `emg` (float32).  b, a: HOST pointers to the n_coef <= 17
 * IIR coefficients (scipy.signal.butter(4, (20, 450)/1000, "bandpass") in the reference); gain = 2**10;
 * rms_window = 11; time_idx: HOST pointer to the n_out <= 256 kept positions of the RMS series
 * (load.py:115 time_mask, whose uint8 wraps modulo 256 -- pass what the reference computes).
 * out (n_segments, n_out, 12) f32.  Rounding points follow NumPy/SciPy exactly: bit-identical samples. */
int cp_preprocess_emg(const float* raw, int64_t n_segments, int32_t seg_len, const double* b, const double* a,
                      int32_t n_coef, int32_t rms_window, float gain, const int32_t* time_idx, int32_t n_out,
                      float* out, void* stream);

/* utils.RunningStats over preprocessed segments (code/utils.py:79-135; load.py:116,141-144): statistics of the
 * per-segment channel means of the segments with use[s] != 0 (use == NULL: all) -- their mean and sample standard
 * deviation per channel, or averaged over channels when complete != 0.  scratch: n_segments*12 doubles on the
 * device.  mean_std (2,12) f32 on the device. */
int cp_emg_stats(const float* seg, int64_t n_segments, int32_t n_out, const uint8_t* use, int32_t complete,
                 double* scratch, float* mean_std, void* stream);

/* RunningStats.normalize (code/utils.py:134, load.py:148): seg = (seg - mean) / std in place, f32. */
int cp_emg_normalize(float* seg, int64_t n_rows, const float* mean_std, void* stream);

/* ---- glove-angle class encoder (BASELINE config 3) -------------------------------------
 * zg = last(relu(BN(Linear(20->256, no bias)(glove)))), last = Linear(256->16, no bias): the layers the reference
 * keeps as comments in GLOVENet (code/models.py:386-391, 461) plus its built-but-unused `self.last`
 * (code/models.py:425-428).  One row per (group, class).  The contrastive head then takes row (b*41 + j) of zg as
 * the class embedding of position j of group b, instead of the one-hot table's row labels[b*41 + j]. */
typedef struct cp_glove_params {
    float* w1;            /* glove_net.linear.1.weight        (256,20) */
    float* bn_g;          /* glove_net.linear.2[.bn].weight   (256) */
    float* bn_b;          /* glove_net.linear.2[.bn].bias     (256) */
    float* last_w;        /* glove_net.last.0.weight          (16,256) */
    float* running_mean;  /* stock BN buffers (NULL under AdaBN; unused in a gradient struct) */
    float* running_var;
} cp_glove_params;

size_t cp_glove_workspace_bytes(int64_t max_rows, int32_t dtype);

/* GLOVENet.forward, glove branch.  glove (rows,20) f32, rows = B*41; zg (rows,16) f32.  cfg supplies dtype, adabn,
 * training, bn_momentum, bn_eps (n_windows and the dropout fields are not used).  Saves what backward needs in gws. */
int cp_glove_forward(const cp_config* cfg, const cp_glove_params* gp, const float* glove, int64_t rows,
                     void* gws, size_t gws_bytes, float* zg, void* stream);

/* cp_head with per-group class embeddings: same outputs; want_grad (V must be 1) leaves dL/dz in ws for
 * cp_encoder_backward and dL/dzg in gws for cp_glove_backward. */
int cp_head_glove(const cp_config* cfg, const float* z, const float* zg, const int64_t* labels, int64_t n_groups,
                  int32_t V, int32_t want_grad, void* ws, size_t ws_bytes, void* gws, size_t gws_bytes,
                  float* loss_correct, int32_t* pred, float* logits, void* stream);

/* autograd of the glove encoder: consumes dL/dzg left in gws, writes w1, bn_g, bn_b, last_w of `grads`. */
int cp_glove_backward(const cp_config* cfg, const cp_glove_params* gp, int64_t rows, void* gws, size_t gws_bytes,
                      cp_glove_params* grads, void* stream);

/* Model.l2() (code/models.py:225-228, 344-349, 467-472) + optimizer_emg.step() +
 * optimizer_glove.step() (code/train.py:72-73, 101, 107-108) over one flat parameter buffer.
 * Tensor table (host arrays, n <= 64): offset/numel into the flat buffers, group (0 emg_net,
 * 1 glove_net), l2 (1 if the tensor's name contains neither 'bn' nor 'bias').
 * step_index: 1-based Adam step.  grad_scale multiplies the data gradient (1/world_size after an
 * all-reduce sum).  l2_out: device scalar receiving the regulariser value.  scratch: device floats,
 * at least cp_optimizer_scratch_floats(...) long. */
typedef struct cp_adam_hyper {
    float lr_emg, lr_glove, reg_emg, reg_glove;
    float beta1, beta2, eps, grad_scale;
} cp_adam_hyper;
/* cp_l2_adam_step with lr_emg, lr_glove and the bias corrections read from a device cp_step_state (graph replay);
 * the other fields of h are used as given. */
int cp_l2_adam_step_graph(float* params_flat, const float* grads_flat, float* exp_avg, float* exp_avg_sq,
                          const int64_t* offset_host, const int64_t* numel_host, const int32_t* group_host,
                          const int32_t* l2_host, int32_t n, const cp_adam_hyper* h, const cp_step_state* state_dev,
                          float* scratch, float* l2_out, void* stream);
size_t cp_optimizer_scratch_floats(const int64_t* numel_host, int32_t n);
int cp_l2_norms(const float* params_flat, const int64_t* offset_host, const int64_t* numel_host,
                const int32_t* group_host, const int32_t* l2_host, int32_t n, const cp_adam_hyper* h,
                float* scratch, float* l2_out, void* stream);
int cp_l2_adam_step(float* params_flat, const float* grads_flat, float* exp_avg, float* exp_avg_sq,
                    const int64_t* offset_host, const int64_t* numel_host, const int32_t* group_host,
                    const int32_t* l2_host, int32_t n, const cp_adam_hyper* h, int64_t step_index,
                    float* scratch, float* l2_out, void* stream);

/* Optional timing of kernel groups with HIP events recorded on the launch stream (used by
 * bench.py for the live roofline figure).  cp_profile_enable creates the events (never done inside
 * a step); every profiled launch group then records a start/stop pair until max_records are used.
 * cp_profile_summary (after the caller synchronised the stream) sums the elapsed times of one kind. */
enum {
    CP_K_GATHER = 0, CP_K_PREP = 1, CP_K_CONV1_FWD = 2, CP_K_BN_FINALIZE = 3, CP_K_CONV2_FWD = 4,
    CP_K_FOLD = 5, CP_K_FC_FWD = 6, CP_K_DROPOUT = 7, CP_K_PROJ_FWD = 8, CP_K_HEAD = 9,
    CP_K_PROJ_BWD = 10, CP_K_BN_BWD = 11, CP_K_FC_WGRAD = 12, CP_K_REDUCE_SLABS = 13,
    CP_K_FC_DGRAD = 14, CP_K_CONV2_WGRAD = 15, CP_K_CONV2_DGRAD = 16, CP_K_CONV1_BWD = 17,
    CP_K_OPT = 18, CP_K_FC_DGRAD_STATS = 19, CP_K_FC_DGRAD_BN = 20,
    CP_K_FC_FWD_WS = 21,         /* forward fc launches that ran the weight-stationary kernel (K = 512: fc2..fc7) */
    CP_K_FC_DGRAD_CONV = 22,     /* CP_FP8: fc1's data gradient (16-bit output for the conv kernels) -- its own kernel instantiation */
    CP_K_COUNT = 23
};
int cp_profile_enable(uint64_t kind_mask, int32_t max_records);
int cp_profile_disable(void);
/* records again after cp_profile_disable, keeping what was recorded (sampling every n-th step: each recorded launch costs
 * two event records, ~5 us of idle queue apiece) */
int cp_profile_resume(void);
int cp_profile_summary(int32_t kind, double* total_ms, int64_t* count);

/* debug/test access: copy saved activation `layer` (0..8 = post-ReLU pre-BN output of conv1,
 * conv2, fc1..fc7; rows x C in the internal layout, conv layers position-major [w][c]; 9..12 =
 * dropout(BN(.)) of fc4..fc7, present only when dp_emg > 0 and the forward ran in training) to f32.
 * Layer 0 (conv1) is never stored by the forward pass -- its consumers recompute it from x -- so it
 * is recomputed here the same way from `p` and `x` (both may be NULL for the other layers). */
int cp_debug_activation(const cp_config* cfg, const cp_params* p, const float* x, void* ws,
                        size_t ws_bytes, int32_t layer, float* out, void* stream);
/* micro-benchmark access (tools/gemm_bench.py): one fc-layer GEMM launch on caller buffers.
 * kind 0: forward   C[M][F] = relu(A[M][K] W[F][K]^T + bias), column sums -> partials
 * kind 1: data grad C[M][F] = A[M][K] W[F][K]^T, sums against R[M][F] -> partials
 * kind 2: weight grad slabs[S][P][Q] = sum_m X[m][P] Y[m][Q]   (A = X, W = Y, K = P, F = Q)
 * dbg: ablation bits of the bf16 kernels (1 skip MFMA, 2 skip epilogue, 4 skip staging loads). */
/* test aid: `blocks` workgroups of 256 threads that each hold a CU's LDS (so nothing else fits next to them there) and spin
 * for about `microseconds` -- stands in for another stream's kernel (an RCCL collective) competing for CUs. */
int cp_debug_hog(int32_t blocks, int32_t microseconds, void* stream);
int cp_debug_gemm(int32_t dtype, int32_t kind, int64_t M, int32_t K, int32_t F, const void* A,
                  const void* W, void* C, const float* bias, const void* R, float* partials,
                  int32_t dbg, void* stream);


/* BN statistics of `layer` as computed by the last forward: out[4][C] = mean, invstd, scale, shift */
int cp_debug_bn_stats(const cp_config* cfg, void* ws, size_t ws_bytes, int32_t layer, float* out,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
