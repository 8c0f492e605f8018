/*
 * ppo_amd.h — C ABI of libppo_amd.so, the MI355X (gfx950) hot path of the PPO
 * trainer.  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * unless its comment says host.  No entry point allocates, synchronises the
 * device or touches the host copy of any buffer; all work is enqueued on
 * `stream` (a hipStream_t passed as void*, NULL = the default stream) so calls
 * can be captured into a hipGraph.
 *
 * The reference (dremovd/PPO) is 100 % Python and has no FFI; each entry point
 * below replaces the Python function cited beside it.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Return value: 0 on success, <0 on error (PPO_E_*); ppo_last_error() gives a
 * thread-local human-readable message.  Nothing is written on error.
 */
#ifndef PPO_AMD_H
#define PPO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_OK 0
#define PPO_E_INVALID (-1) /* bad shape / null pointer / unsupported kind */
#define PPO_E_HIP (-2)     /* a HIP runtime call failed */
#define PPO_E_ALIGN (-3)   /* pointer or leading dimension not aligned as documented */

/* dtype of the `terminals` / `dones` operand.  The reference's arithmetic
 * depends on it (NumPy promotion, rl/returns.py:24-28): bool terminals make
 * the recurrence run in float64 with only the stored result rounded to f32;
 * float32 terminals (or None) keep everything in float32. */
#define PPO_TERM_NONE 0 /* terminals == NULL: no episode ends inside the rollout */
#define PPO_TERM_U8 1   /* uint8 0/1 (NumPy bool)  -> float64 carry */
#define PPO_TERM_F32 2  /* float32 0.0/1.0         -> float32 carry */

/* which scan kernel to run */
#define PPO_SCAN_AUTO 0
#define PPO_SCAN_COLUMNS 1 /* one thread per 4 env columns, serial over time; bit-exact
                              with the reference's loop order; for wide batches */
#define PPO_SCAN_TILES 2   /* time axis split across the waves of a workgroup and
                              recombined through LDS in float64; for narrow batches */

int ppo_version(void);
const char *ppo_last_error(void);

/*
 * Fused GAE advantages + lambda-returns over a time-major rollout.
 * Replaces rl.returns.gae (rl/returns.py:7-29) and rl.returns.td_lambda
 * (rl/returns.py:58-67) as called by Runner.calculate_returns
 * (rl/rollout.py:1207-1223):
 *
 *   delta_t = r_t + gamma * V_{t+1} * (1 - term_t) - V_t        V_N = final_value
 *   adv_t   = delta_t + gamma*lam_adv * (1 - term_t) * adv_{t+1}         adv_N = 0
 *   g_t     = delta_t + gamma*lam_ret * (1 - term_t) * g_{t+1}           g_N   = 0
 *   ret_t   = f32(g_t) + V_t
 *
 * rewards, values, terminals, adv_out, ret_out: [N, ld] row-major, A <= ld
 * columns used (time-major, env index contiguous).  final_value: [A].
 * adv_out and ret_out may each be NULL (then that output is skipped).
 * gamma / lam_* are the python floats of the reference (double), because the
 * product gamma*lam is formed in double there.
 *
 * Algorithmic HBM traffic: 9 B read + 8 B written per (t, env) element.
 */
int ppo_gae_scan_f32(const float *rewards, const float *values, const float *final_value,
                     const void *terminals, int terminal_kind,
                     float *adv_out, float *ret_out,
                     int N, int A, int64_t ld,
                     double gamma, double lam_adv, double lam_ret,
                     int regime, void *stream);

/*
 * Discounted bootstrapped returns.  Replaces
 * rl.returns.calculate_bootstrapped_returns (rl/returns.py:32-55):
 *   G_t = r_t + G_{t+1} * gamma_t * (1 - done_t),  G_N = final_value.
 * gamma_arr: optional [N, ld] f32 per-element discount (NULL -> scalar gamma,
 * rounded to f32 as the reference does).  done_kind: PPO_TERM_U8 or PPO_TERM_F32.
 */
int ppo_bootstrapped_returns_f32(const float *rewards, const void *dones, int done_kind,
                                 const float *final_value, const float *gamma_arr, double gamma,
                                 float *out, int N, int A, int64_t ld, void *stream);

/* ------------------------------------------------------------------------
 * IMPALA-CNN building blocks (reference: rl/impala.py, rl/models.py:54-99).
 * Activations are NCHW float32, contiguous.  Supported geometries: the layers
 * of the (16, 32, 32)-channel IMPALA stacks on 84x84 (Atari) and 64x64
 * (Procgen) observations; anything else returns PPO_E_INVALID.
 * ---------------------------------------------------------------------- */

/* input transform fused into the convolution's load */
#define PPO_IN_NONE 0 /* float32 input used as is */
#define PPO_IN_RELU 1 /* float32 input, max(x, 0) applied on load (pre-activation blocks, rl/impala.py:73-78) */
#define PPO_IN_U8 2   /* uint8 observation, x / 255 applied on load (rl/models.py:842-848, "scaled") */

/*
 * out[n,o,y,x] = bias[o] + sum_{i,ky,kx} f(in[n,i,y+ky-1,x+kx-1]) * weight[o,i,ky,kx] (+ residual[n,o,y,x])
 * 3x3, stride 1, zero padding 1 (torch.nn.Conv2d as used at rl/impala.py:61-62,96).
 * weight: [cout, cin, 3, 3] (PyTorch layout).  bias, residual: nullable.
 * FLOPs: 2*9*cin*cout*h*w per image, on the f32 MFMA.
 */
int ppo_conv3x3_forward_f32(const void *in, int in_mode, const float *weight, const float *bias,
                            const float *residual, float *out, int n, int cin, int cout, int h, int w,
                            void *stream);

/*
 * Gradient w.r.t. the convolution's (pre-transform) input:
 *   dx[n,i,y,x] = (sum_{o,ky,kx} dy[n,o,y-ky+1,x-kx+1] * weight[o,i,ky,kx]) * [relu_src[n,i,y,x] > 0] + dres[n,i,y,x]
 * relu_src (nullable): the pre-activation tensor the forward pass ReLU-ed on load.
 * dres (nullable): gradient arriving over the residual skip connection.
 * cin/cout are those of the FORWARD convolution.
 */
int ppo_conv3x3_backward_data_f32(const float *dy, const float *weight, const float *relu_src,
                                  const float *dres, float *dx, int n, int cin, int cout, int h, int w,
                                  void *stream);

/*
 * Stack-first convolution fused with the max-pool that follows it (rl/impala.py:104-105):
 *   out = max_pool2d(conv3x3(f(in)) + bias, kernel 3, stride 2, padding 1),   [n,cout,(h+1)/2,(w+1)/2]
 * bit-identical to ppo_conv3x3_forward_f32 followed by ppo_maxpool3x3s2_forward_f32, without the pre-pool
 * map ever reaching HBM.  in_mode: PPO_IN_NONE or PPO_IN_U8.  argmax (nullable) as in the pool entry point.
 */
int ppo_conv3x3_pool_forward_f32(const void *in, int in_mode, const float *weight, const float *bias, float *out,
                                 uint8_t *argmax, int n, int cin, int cout, int h, int w, void *stream);

/*
 * Pre-packed weights.  Each convolution kernel keeps its MFMA A operand in registers; from the raw [cout,cin,3,3]
 * tensor a workgroup has to stage it through LDS first (two barriers and a global-load latency before the first
 * MFMA of every launch).  ppo_conv3x3_pack_weights_f32 writes, for up to 32 layers in one launch, that operand in
 * per-lane order — `transposed` = 0 for the forward kernels, 1 for backward-data (flipped, in/out swapped) —
 * into buffers of ppo_conv3x3_packed_floats(cin, cout, transposed) floats (16-byte aligned); the *_packed_f32
 * entry points are the convolutions above with `packed` in place of `weight` (cin / cout are always those of
 * the FORWARD convolution).  Results are bit-identical; repack after every optimiser step.
 */
typedef struct ppo_pack_job {
    const float *weight; /* raw [cout, cin, 3, 3] */
    float *packed;       /* ppo_conv3x3_packed_floats(cin, cout, transposed) floats */
    int cin, cout, transposed;
} ppo_pack_job;
size_t ppo_conv3x3_packed_floats(int cin, int cout, int transposed);
int ppo_conv3x3_pack_weights_f32(const ppo_pack_job *jobs /* host */, int n_jobs, void *stream);
int ppo_conv3x3_forward_packed_f32(const void *in, int in_mode, const float *packed, const float *bias,
                                   const float *residual, float *out, int n, int cin, int cout, int h, int w,
                                   void *stream);
int ppo_conv3x3_backward_data_packed_f32(const float *dy, const float *packed, const float *relu_src,
                                         const float *dres, float *dx, int n, int cin, int cout, int h, int w,
                                         void *stream);
int ppo_conv3x3_pool_forward_packed_f32(const void *in, int in_mode, const float *packed, const float *bias,
                                        float *out, uint8_t *argmax, int n, int cin, int cout, int h, int w,
                                        void *stream);
/* The same launch reading image i of the batch at in[index[i]] (uint8 observations; index nullable): the minibatch gather of
 * rl/rollout.py:2349-2372 (host fancy-indexing + upload there) happens in the first convolution's own loads instead of in a
 * gather launch and a second copy of the observations.  ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32 is the
 * matching form of the first layer's weight gradient. */
int ppo_conv3x3_pool_forward_packed_indexed_f32(const void *in, const int32_t *index, int in_mode, const float *packed,
                                                const float *bias, float *out, uint8_t *argmax, int n, int cin, int cout,
                                                int h, int w, void *stream);
int ppo_conv3x3_backward_weight_slabs_pooled_indexed_f32(const void *in, const int32_t *index, int in_mode, const float *g,
                                                         const uint8_t *argmax, void *workspace, size_t workspace_bytes,
                                                         int n, int cin, int cout, int h, int w, int *n_slabs, void *stream);
/* Which kernel the uint8 first-layer launches of ppo_conv3x3_pool_forward_*f32 take (rl/impala.py:96,104-105): the LDS
 * form (conv3x3.hip) or the form that pools out of the MFMA accumulators (conv1_pool.hip, 4 x 84 x 84 observations).  Both are
 * bit-identical, so this selects speed, never results: form 0 = from the accumulators wherever that kernel exists, 1 = the LDS
 * form everywhere, -1 = the measured default (today the LDS form: faster inside the pipelined rollout).
 * Returns the previous setting; process-wide, for A/B timing and the tests. */
int ppo_conv1_pool_form(int form);

/*
 * One residual block, q' = q + conv1(relu(conv0(relu(q)))) (rl/impala.py:66-84), as ONE launch for inference on small
 * batches (a rollout group): band by band, the intermediate map stays in LDS (csrc/conv3x3_block.hip).  packed0 / packed1:
 * the two convolutions' pre-packed forward weights (ppo_conv3x3_pack_weights_f32), bias0 / bias1 their biases; in / out
 * [n, channels, h, w].  Bit-identical to ppo_conv3x3_forward_packed_f32(in, RELU, conv0) followed by
 * ppo_conv3x3_forward_packed_f32(., RELU, conv1, residual = in).
 */
int ppo_conv3x3_block_supported(int channels, int h, int w);
int ppo_conv3x3_block_forward_packed_f32(const float *in, const float *packed0, const float *bias0, const float *packed1,
                                         const float *bias1, float *out, int n, int channels, int h, int w, void *stream);

/*
 * Weight and bias gradient of the same convolution (torch autograd of nn.Conv2d):
 *   dweight[o,i,ky,kx] (+)= sum_{n,y,x} dy[n,o,y,x] * f(in[n,i,y+ky-1,x+kx-1])    dbias[o] (+)= sum dy[n,o,y,x]
 * `in`/in_mode: the forward convolution's input and load transform.  workspace: scratch of at
 * least ppo_conv3x3_wgrad_workspace_bytes(cin, cout) bytes (per-workgroup partial slabs, summed in
 * a fixed order: deterministic).  accumulate != 0 adds into dweight/dbias (micro-batches,
 * rl/rollout.py:2331-2374).  dbias nullable.
 */
size_t ppo_conv3x3_wgrad_workspace_bytes(int cin, int cout);
int ppo_conv3x3_backward_weight_f32(const void *in, int in_mode, const float *dy, float *dweight, float *dbias,
                                    void *workspace, size_t workspace_bytes, int n, int cin, int cout, int h,
                                    int w, int accumulate, void *stream);

/*
 * The same weight gradient in two steps, for callers that batch the reductions of many layers into one launch
 * (a backward pass: 15 convolutions): *_slabs_f32 runs the MFMA kernel only and leaves its per-workgroup partial
 * slabs in `workspace` (which the caller keeps until the reduction), writing their number to *n_slabs (host);
 * ppo_conv3x3_wgrad_reduce_f32 then sums the slabs of up to 32 layers (`jobs`: HOST array) in fixed order into
 * dweight / dbias.  Results are bit-identical to ppo_conv3x3_backward_weight_f32.
 */
typedef struct ppo_wgrad_job {
    const float *slabs; /* workspace of the matching *_slabs_f32 call */
    float *dweight;     /* [cout, cin, 3, 3] */
    float *dbias;       /* [cout] or NULL */
    int n_slabs, cin, cout;
    int accumulate;     /* != 0: add into dweight / dbias */
} ppo_wgrad_job;
int ppo_conv3x3_backward_weight_slabs_f32(const void *in, int in_mode, const float *dy, void *workspace,
                                          size_t workspace_bytes, int n, int cin, int cout, int h, int w,
                                          int *n_slabs, void *stream);
/* `count` (1..4) problems of one geometry in one launch: ins / dys / workspaces are HOST arrays of device pointers, each
 * workspace workspace_bytes long; *n_slabs is the slab count of every one of them.  Same slabs as `count` calls. */
int ppo_conv3x3_backward_weight_slabs_batch_f32(const void *const *ins, int in_mode, const float *const *dys,
                                                void *const *workspaces, size_t workspace_bytes, int count, int n, int cin,
                                                int cout, int h, int w, int *n_slabs, void *stream);
/* The same launch with a per-problem load transform: relu[k] != 0 reads problem k's input through the forward's ReLU
 * (block convolutions), 0 reads it raw (a stack-first convolution): the first convolution of the NEXT stack has the
 * geometry of this stack's block convolutions and its gradient is known by the time theirs are, so it rides in their
 * launch as a fifth problem instead of paying a launch of its own (11 - 14 us fixed, profiles/r04_wgrad_fixed_cost.md).
 * Same slabs, same bits as the separate launches of the same grid. */
int ppo_conv3x3_backward_weight_slabs_batch_mixed_f32(const void *const *ins, const int *relu, const float *const *dys,
                                                      void *const *workspaces, size_t workspace_bytes, int count, int n,
                                                      int cin, int cout, int h, int w, int *n_slabs, void *stream);
/* OPT-IN reduced precision (`--precision=medium|low`, /root/reference train.py:166-178): the same launch with every product
 * taken as three bf16 MFMAs on (hi, lo) splits of both operands, float32 accumulation (csrc/wgrad_bf16x3.hip; products carry
 * ~16 bits).  Float32 inputs only (16 or 32 channels: ppo_conv3x3_backward_weight_bf16x3_supported); count 1..5; the slabs
 * have the layout of the float32 kernels', so ppo_conv3x3_wgrad_reduce_f32 folds them.  db is summed in float32. */
int ppo_conv3x3_backward_weight_bf16x3_supported(int cin, int cout, int h, int w);
int ppo_conv3x3_backward_weight_slabs_batch_bf16x3(const float *const *ins, const int *relu, const float *const *dys,
                                                   void *const *workspaces, size_t workspace_bytes, int count, int n, int cin,
                                                   int cout, int h, int w, int *n_slabs, void *stream);
/* A stack's FIRST convolution with its dy taken from the pooled gradient: dy = ppo_maxpool3x3s2_backward_f32(g, argmax)
 * is formed band by band inside the kernel (g [n,cout,h/2,w/2], argmax uint8 likewise), so the pre-pool gradient map
 * never exists in HBM.  Same slabs as the max-pool backward launch followed by ppo_conv3x3_backward_weight_slabs_f32.
 * Only where nothing else reads that map (the first stack: no backward-data into the observations); geometries per
 * ppo_conv3x3_backward_weight_pooled_supported. */
int ppo_conv3x3_backward_weight_pooled_supported(int cin, int cout, int h, int w);
int ppo_conv3x3_backward_weight_slabs_pooled_f32(const void *in, int in_mode, const float *g, const uint8_t *argmax,
                                                 void *workspace, size_t workspace_bytes, int n, int cin, int cout, int h,
                                                 int w, int *n_slabs, void *stream);
int ppo_conv3x3_wgrad_reduce_f32(const ppo_wgrad_job *jobs, int n_jobs, void *stream);

/*
 * 3x3 / stride 2 / pad 1 max pooling (F.max_pool2d, rl/impala.py:105): [n,c,h,w] -> [n,c,(h+1)/2,(w+1)/2].
 * argmax (nullable uint8 [n,c,ho,wo]) records the winning window tap for the backward pass.
 * Backward: din[n,c,h,w] = sum of dout over the windows whose argmax is this element.
 */
int ppo_maxpool3x3s2_forward_f32(const float *in, float *out, uint8_t *argmax, int n, int c, int h, int w,
                                 void *stream);
int ppo_maxpool3x3s2_backward_f32(const float *dout, const uint8_t *argmax, float *din, int n, int c, int h, int w,
                                  void *stream);

/*
 * C[m,n] = epi( sum_k fa(A[m,k]) * fb(B[k,n]) + bias[n] ),  f32 MFMA.
 * A element (m,k) at A[m*a_sm + k*a_sk]; B element (k,n) at B[k*b_sk + n*b_sn]; C row-major with ldc.
 * relu_a / relu_b: apply max(.,0) to that operand on load.  bias [N], mask [M,ldc] nullable;
 * mask gates the result: C = mask > 0 ? C : 0 (ReLU backward).  workspace (nullable): split-K
 * scratch of ppo_gemm_workspace_bytes(M,N,K); without it the kernel runs unsplit.
 * Strides are non-negative; each operand must span less than 2 GiB (the kernels address it with 32-bit byte
 * offsets through a range-checked buffer descriptor) - PPO_E_INVALID otherwise: split the batch.
 * Replaces torch.nn.Linear forward/backward at rl/models.py:84,98,364-366,470-506.
 */
size_t ppo_gemm_workspace_bytes(int M, int N, int K);
int ppo_gemm_f32(const float *A, int64_t a_sm, int64_t a_sk, int relu_a, const float *B, int64_t b_sk, int64_t b_sn,
                 int relu_b, const float *bias, const float *mask, float *C, int64_t ldc, int M, int N, int K,
                 void *workspace, size_t workspace_bytes, void *stream);

/*
 * The tail of every forward pass in one call: h[M,H] = f(x)[M,K] @ W[H,K]^T + b (the encoder's dense layer, f = ReLU when
 * relu_x), heads[M,NH] = g(h) @ Wh[NH,H]^T + bh (the fused policy / value / advantage / TVF heads, g = ReLU when relu_h).
 * Same results, bit for bit, as the two ppo_gemm_f32 calls it stands for; when the dense product runs K-sliced (a
 * workspace of ppo_gemm_workspace_bytes(M,H,K) is given, K >= 512, H <= 256, NH <= 16) the slice reduction and the heads
 * are one launch.  Replaces rl/models.py:84 (dense) + :467-506 (heads) of DualHeadNet.forward.
 */
int ppo_dense_heads_forward_f32(const float *x, int relu_x, const float *W, const float *b, const float *Wh, const float *bh,
                                int relu_h, float *h, float *heads, int M, int K, int H, int NH, void *workspace,
                                size_t workspace_bytes, void *stream);

/*
 * The same, and the rollout's action step on the finished head row in the same launch: what ppo_policy_act_f32 does
 * with greedy = 0 and its own uniforms (log-softmax of the first n_actions head outputs, Gumbel-max sample keyed by
 * (seed, offset + row * n_actions + a), raw logits, value heads; any output pointer may be NULL).  Bit-identical to
 * ppo_dense_heads_forward_f32 followed by ppo_policy_act_f32 - and runs as exactly that when the product is not split
 * or n_actions is not one of 4 / 6 / 15 / 18.  Replaces the tail of Runner.detached_batch_forward + sample_actions
 * (rl/rollout.py:557-598, 636-650) for one env group.
 */
int ppo_dense_heads_act_forward_f32(const float *x, int relu_x, const float *W, const float *b, const float *Wh,
                                    const float *bh, int relu_h, float *h, float *heads, int M, int K, int H, int NH,
                                    void *workspace, size_t workspace_bytes, int n_actions, float temperature,
                                    uint64_t seed, uint64_t offset, float *log_policy, int32_t *actions, float *log_pac,
                                    float *raw_policy, float *values, int n_value_heads, void *stream);
/* The TRAINING counterpart: dense layer + heads + the discrete PPO loss of Runner.train_policy_minibatch
 * (rl/rollout.py:1640-1660, 1682, 1744-1753, 1596-1608; arguments as ppo_ppo_loss_f32) on the finished head row in the
 * finalize launch: dheads and the statistics rows leave from there.  Bit-identical to ppo_dense_heads_forward_f32 followed by
 * ppo_ppo_loss_f32; shapes without a fused form run exactly those launches inside the entry point. */
int ppo_dense_heads_loss_forward_f32(const float *x, int relu_x, const float *W, const float *b, const float *Wh, const float *bh,
                                     int relu_h, float *h, float *heads, int M, int K, int H, int NH, void *workspace,
                                     size_t workspace_bytes, int n_actions, int n_value_heads, const int32_t *actions,
                                     const float *old_log_pac, const float *old_log_policy, const float *advantages,
                                     const float *returns, float eps_clip, float ent_coef, float vf_coef, float grad_scale,
                                     float *dheads, float *stats, const int32_t *index, void *stream);

/*
 * Backward of the fused heads in one launch: dh[B,H] = (dheads[B,NH] @ Wh[NH,H]) * [gate > 0] (gate [B,H] nullable: the
 * ReLU pre-activation), dWh[NH,H] = dheads^T @ f(hin) (f = ReLU when relu_in), dbh[NH] = column sums of dheads
 * (nullable), db_next[H] = column sums of dh (nullable: the bias gradient of the layer below when dh is final).
 * NH <= 16.  The two column sums are bit-identical to ppo_colsum_f32.  Replaces the autograd of rl/models.py:467-506.
 */
int ppo_heads_backward_f32(const float *dheads, const float *hin, int relu_in, const float *gate, const float *Wh, float *dh,
                           float *dWh, float *dbh, float *db_next, int B, int H, int NH, void *stream);
/* out[n] (+)= sum_m X[m*ldx + n]  (bias gradients) */
int ppo_colsum_f32(const float *X, int M, int N, int64_t ldx, float *out, int accumulate, void *stream);

/*
 * Policy post-processing per sample row `heads[b, :]` = [logits(n_actions) | value heads | ...]:
 * log_policy = log_softmax(logits / temperature) (rl/models.py:488) and an action:
 *   greedy == 0: Gumbel-max  argmax_a(log_policy - log(-log u_a))  (rl/utils.py:248-256), with
 *                u from `uniform` [B,n_actions] if given, else from a counter-based generator
 *                keyed by (seed, offset + b*n_actions + a);
 *   greedy != 0: argmax of the logits (rl/models.py:479, run_evaluation.py:621-623).
 * Outputs (each nullable): log_policy [B,n_actions], actions [B] int32, log_pac [B] = log_policy[b, action],
 * raw_policy [B,n_actions] (the logits) and values [B,n_value_heads] (the value-head columns), i.e. the
 * per-step rows the rollout buffer stores (rl/rollout.py:807-815), written straight into it.
 */
int ppo_policy_act_f32(const float *heads, int B, int ldo, int n_actions, float temperature, const float *uniform,
                       uint64_t seed, uint64_t offset, int greedy, float *log_policy, int32_t *actions,
                       float *log_pac, float *raw_policy, float *values, int n_value_heads, void *stream);

/*
 * PPO minibatch loss, forward + gradient w.r.t. the head outputs
 * (Runner.train_policy_minibatch / train_value_heads, rl/rollout.py:1596-1608,1610-1753):
 *   gain_b = min(rho*A, clip(rho,1-eps,1+eps)*A) + ent_coef*H_b - sum_heads vf_coef*(V-R)^2
 *   loss   = mean_b(-gain_b) * loss_scale          =>   grad_scale = loss_scale / B
 * dheads [B,ldo] receives d loss / d heads (zeros in the columns past n_actions+n_value_heads).
 * stats (nullable) [B,8]: loss_clip, entropy, value_loss, clipped(0/1), old_log_pac-log_pac,
 * KL(new||old) term, gain, rho.  old_log_policy [B,n_actions] nullable (only for the KL statistic).
 */
int ppo_ppo_loss_f32(const float *heads, int B, int ldo, int n_actions, int n_value_heads, const int32_t *actions,
                     const float *old_log_pac, const float *old_log_policy, const float *advantages,
                     const float *returns, float eps_clip, float ent_coef, float vf_coef, float grad_scale,
                     float *dheads, float *stats, const int32_t *index, void *stream);
/* index (nullable, [B] int32): sample b reads actions / old_log_pac / old_log_policy / advantages /
 * returns at row index[b] of the whole-batch arrays (the minibatch permutation), so those need no gather. */

/*
 * Elementwise tanh and its backward dx = dy * (1 - y^2), y = tanh(x)  (torch.tanh at rl/models.py:166,
 * 456-457: the MLP encoder and the `tanh` encoder activation of the mujoco configs).
 */
int ppo_tanh_forward_f32(const float *x, float *y, size_t n, void *stream);
int ppo_tanh_backward_f32(const float *dy, const float *y, float *dx, size_t n, void *stream);

/*
 * Value-phase loss of the dual architecture (Runner.train_value_minibatch, rl/rollout.py:1513-1567):
 *   loss_b = sum_{i<n_value_heads} vf_coef (V_i - R_i)^2                         (rl/rollout.py:1596-1608; returns nullable)
 *          + tvf_coef * 0.5 * sqrt(K) * mean_k w_k (T_k - P_k)^2                 (rl/tvf.py:49-73; tvf_returns nullable)
 * V_i = heads[b, value_col + i]; P_k = heads[b, tvf_col + k*tvf_stride] (the `ext` column of TVF head k);
 * returns [*,n_value_heads], tvf_returns [*,n_tvf], tvf_weights [n_tvf] (nullable = 1).  dheads [B,ldo] gets
 * grad_scale * d loss_b / d heads over the WHOLE row (zeros elsewhere).  stats (nullable) [B,4]: value
 * loss, TVF loss, total, 0.  index as in ppo_ppo_loss_f32.  tvf_keep_prob < 1: horizon dropout (rl/tvf.py:64-69) — each
 * (sample, head) TVF term is kept with that probability and weighted 1 / tvf_keep_prob, drawn from the counter-based
 * generator of ppo_policy_act_f32 at (seed, offset + b * n_tvf + k); 1 = off.
 */
int ppo_value_loss_f32(const float *heads, int B, int ldo, int value_col, int n_value_heads, const float *returns,
                       float vf_coef, int tvf_col, int n_tvf, int tvf_stride, const float *tvf_returns,
                       const float *tvf_weights, float tvf_coef, float grad_scale, float *dheads, float *stats,
                       const int32_t *index, float tvf_keep_prob, uint64_t seed, uint64_t offset, void *stream);

/*
 * Distillation-phase loss (Runner.train_distil_minibatch, rl/rollout.py:1331-1449; value_loss "mse"):
 *   loss_b = 0.5 w_k (T_k - P_k)^2  [* sqrt(n_pred) mean_k when vector_targets]  + beta * policy term
 * P_k = heads[b, pred_col + k*pred_stride]; targets [*,n_pred]; weights [n_pred] nullable.
 * Policy term: log_std == NULL -> KL(pi_new || pi_old) with old_policy = old log-probabilities [*,n_actions]
 * (distil loss "kl_policy", :1414-1415); log_std != NULL (gaussian policies, :1401-1409) ->
 * 2 * 0.5 mean_a (mu_old - mu)^2 / (1e-5 + 2 exp(log_std_a)^2) with old_policy = old means (the reference
 * adds this term twice, :1409 and :1419; its scale is kept).
 * stats (nullable) [B,4]: value loss, policy loss, total, mean_k (w_k (P_k - T_k))^2.
 */
int ppo_distil_loss_f32(const float *heads, int B, int ldo, int n_actions, int pred_col, int n_pred, int pred_stride,
                        int vector_targets, const float *targets, const float *weights, const float *old_policy,
                        const float *log_std, float beta, float grad_scale, float *dheads, float *stats,
                        const int32_t *index, void *stream);

/*
 * Gaussian policy (action_dist "gaussian", rl/rollout.py:643-648): mu = heads[b, :n_actions],
 * action = mu + exp(log_std) * n with n ~ N(0,1) from `normal` [B,n_actions] if given, else Box-Muller on the
 * counter-based uniform stream keyed by (seed, offset + b*n_actions + a); deterministic != 0 -> action = mu.
 * Outputs (nullable): actions, log_pac (= Normal(mu, sigma).log_prob(action) per dimension, :1877-1880),
 * raw_policy (mu), all [B,n_actions]; values [B,n_value_heads].
 */
int ppo_gaussian_act_f32(const float *heads, int B, int ldo, int n_actions, const float *log_std, const float *normal,
                         uint64_t seed, uint64_t offset, int deterministic, float *actions, float *log_pac,
                         float *raw_policy, float *values, int n_value_heads, void *stream);

/*
 * PPO minibatch loss for gaussian policies (rl/rollout.py:1693-1704, 1744-1753):
 *   gain_b = mean_a min(rho_a A, clip(rho_a) A) - sum_heads vf_coef (V - R)^2,  rho_a = exp(logN(a_a; mu_a, sigma_a) - old_log_pac_a)
 * actions, old_log_pac [*,n_actions] f32.  dheads as in ppo_ppo_loss_f32; dlog_std_rows (nullable) [B,n_actions]
 * receives each sample's d loss / d log_std (sum the rows with ppo_colsum_f32).  stats (nullable) [B,8] in
 * ppo_ppo_loss_f32's column order (entropy / KL(new||old) columns are 0, as in the reference).
 */
int ppo_gaussian_loss_f32(const float *heads, int B, int ldo, int n_actions, int n_value_heads, const float *actions,
                          const float *old_log_pac, const float *advantages, const float *returns,
                          const float *log_std, float eps_clip, float vf_coef, float grad_scale, float *dheads,
                          float *dlog_std_rows, float *stats, const int32_t *index, void *stream);

/*
 * One optimiser step on a flat parameter buffer: global-norm clip (clip_grad_norm_,
 * rl/rollout.py:1309-1310; max_grad_norm <= 0 disables) then torch.optim.Adam's update
 * (rl/rollout.py:126-141).  grads are divided by grad_div first (world size after an
 * all-reduce SUM).  step >= 1 is the Adam step count AFTER this call.  workspace:
 * ppo_adam_workspace_bytes() bytes.  grad_norm_out (nullable, device): the pre-clip norm.
 */
size_t ppo_adam_workspace_bytes(void);
int ppo_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step,
                      double lr, double beta1, double beta2, double eps, float max_grad_norm, float grad_div,
                      void *workspace, float *grad_norm_out, void *stream);
/* The optimiser step when the per-workgroup partial sums of g^2 already exist (written by the launch that produced the
 * gradients: ppo_mlp_train_f32): only the Adam launch, which re-reduces `partials[0 .. n_partials)` (<= 256) in a fixed
 * order.  Same arithmetic per parameter as ppo_adam_step_f32 (rl/rollout.py:1287-1321). */
int ppo_adam_step_presummed_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                                int64_t step, double lr, double beta1, double beta2, double eps, float max_grad_norm,
                                float grad_div, const float *partials, int n_partials, float *grad_norm_out, void *stream);
/* The same step that also writes every updated parameter j < n_scatter to packed[scatter[2j]] and packed[scatter[2j+1]]
 * (-1 = nowhere): the head of the flat buffer holds the convolution weights, and their pre-packed MFMA operand layouts
 * (ppo_conv3x3_pack_weights_f32: forward, and flipped / transposed for backward-data) are refreshed by the optimiser
 * step itself instead of by a launch of their own before the next forward (rl/rollout.py:1319 optimizer.step()). */
int ppo_adam_step_scatter_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step,
                              double lr, double beta1, double beta2, double eps, float max_grad_norm, float grad_div,
                              void *workspace, float *grad_norm_out, const int32_t *scatter, int64_t n_scatter,
                              float *packed, void *stream);

/*
 * dst[r, :] = src[index[r], :] for r < n_rows; rows are row_bytes bytes (minibatch gather of
 * observations and per-sample data by a permutation; replaces the host fancy-indexing + upload of
 * rl/rollout.py:2349-2372).  Indices outside [0, n_src_rows) read row 0.
 */
int ppo_gather_rows(const void *src, int64_t row_bytes, int64_t n_src_rows, const int32_t *index, int n_rows,
                    void *dst, void *stream);

/*
 * Batch-level advantage normalisation (Runner.train_policy, rl/rollout.py:1887-1900):
 *   ppo_moments_f64   moments[0..2] = { sum x, sum x^2, n } in float64 (device), fixed-order reduction;
 *                     a data-parallel run all-reduces the three doubles before normalising
 *   ppo_normalize_f32 out = (x - mean) / (std + eps), population std; mean_std_out (nullable, [2])
 * workspace: ppo_moments_workspace_bytes() bytes.
 */
size_t ppo_moments_workspace_bytes(void);
int ppo_moments_f64(const float *x, int64_t n, double *moments, void *workspace, void *stream);
int ppo_normalize_f32(const float *x, int64_t n, const double *moments, float eps, float *out, float *mean_std_out,
                      void *stream);
/* dst[i] += src[i], i < n: gradient accumulation over the micro-batches of a minibatch (Runner.train_batch,
 * rl/rollout.py:2331-2374, where autograd accumulates into .grad across `loss_scale = 1 / micro_batches` passes). */
int ppo_accumulate_f32(float *dst, const float *src, int64_t n, void *stream);
/* w[i] *= mask[i], mask in {0, 1}: DualHeadNet.mask_feature_weights (rl/models.py:425-427), the static feature mask of
 * the TVF head (--tvf_feature_sparsity / --tvf_feature_window, rl/models.py:386-421) re-applied after every optimiser
 * step (the reference re-applies it before every forward that evaluates the head, rl/models.py:494-497: same weights
 * at every use). */
int ppo_mask_mul_f32(float *w, const uint8_t *mask, int64_t n, void *stream);

/*
 * Truncated-horizon (TVF) returns: the sampled weighted-n-step estimator of
 * rl/returns_truncated.py:623-693 (_calculate_sampled_return_multi_fast, with _n_step_estimate :558-620 and
 * _interpolate :142-174), as called by rl/tvf.py:250-262.
 *   out[t,a,k] = mean_c( S_n[t,a] + interp(V[t+n,a,:], h_k-n) * D_n[t,a] ),  n = n_eff[k,c]
 *   (rows with t >= N-n bootstrap from V[N,a,:] at horizon h_k-(N-t); horizons with k_zero[k] give 0)
 * rewards [N,A] f32, dones [N,A] u8 (NumPy bool), value_samples [N+1,A,V] f32, out [N,A,K] f32.
 * The caller resolves the horizon logic on the host (ppo_amd/returns_truncated.py) into device tables:
 *   n_eff, nd_index [K,C] i32   n-step per sample (already clipped to h_k) and its slot in the prefix cache
 *   nd_of_n [max_n+1] i32       cache slot of prefix length n (-1: not needed); ND slots
 *   main_plan [K,C,3] i32 + main_w [K,C,2] f64      interpolation at horizon h_k-n: (mode, i0, i1), (w0, w1);
 *   tail_plan [K,N+1,3] i32 + tail_w [K,N+1,2] f64   the same at horizon h_k-j, j = N-t;
 *                               mode 0: zero, 1: V[i0], 2: f32(V[i0]*w0 + V[i1]*w1) evaluated in float64
 *                               a two-column plan names neighbours: i1 == i0 + 1 (the kernel reads them as a pair)
 * workspace: ppo_tvf_returns_workspace_bytes(N, A, ND, K, C) bytes, 32-byte aligned.
 * Algorithmic HBM traffic: 4*(N+1)*A*V read + 4*N*A*K written.
 */
size_t ppo_tvf_returns_workspace_bytes(int N, int A, int ND, int K, int C);
int ppo_tvf_returns_f32(const float *rewards, const uint8_t *dones, const float *value_samples, int N, int A, int V,
                        int K, int C, double gamma, const int32_t *n_eff, const int32_t *nd_index,
                        const int32_t *nd_of_n, int max_n, int ND, const int32_t *main_plan, const double *main_w,
                        const int32_t *tail_plan, const double *tail_w, const uint8_t *k_zero, void *workspace,
                        size_t workspace_bytes, float *out, void *stream);

/* ------------------------------------------------------------------------
 * The two residual blocks of an IMPALA stack as one launch, image resident in LDS
 * (rl/impala.py:66-84 ImpalaResidualBlock.forward x 2, :110-114):
 *   a0 = conv0_0(relu(in)) + b;  q0 = in + conv1_0(relu(a0)) + b;  a1 = conv0_1(relu(q0)) + b;  q1 = q0 + conv1_1(relu(a1)) + b
 * packed_weights / biases: HOST arrays of 4 device pointers (block0.conv0, block0.conv1, block1.conv0,
 * block1.conv1; weights in the forward packed layout of ppo_conv3x3_pack_weights_f32).  a0, q0, a1 are nullable
 * (inference: only q1 is written).  Results are bit-identical to four ppo_conv3x3_forward_packed_f32 calls.
 * ppo_impala_stack_tail_supported tells whether a (channels, h, w) geometry has a kernel (32 @ 11x11).
 * ---------------------------------------------------------------------- */
int ppo_impala_stack_tail_supported(int channels, int h, int w);
int ppo_impala_stack_tail_forward_f32(const float *in, const float *const *packed_weights, const float *const *biases,
                                      float *a0, float *q0, float *a1, float *q1, int n_images, int channels, int h,
                                      int w, void *stream);
/*
 * Backward-data of the same two blocks in one launch (the transposed chain of ppo_conv3x3_backward_data_packed_f32 x 4):
 *   da1 = conv1_1^T(g) * [a1 > 0];  g1 = g + conv0_1^T(da1) * [q0 > 0];
 *   da0 = conv1_0^T(g1) * [a0 > 0]; g0 = g1 + conv0_0^T(da0) * [p > 0]
 * g = d loss / d q1.  packed_weights_t: HOST array of the 4 backward-data packed weights in PROCESSING order
 * (block1.conv1, block1.conv0, block0.conv1, block0.conv0); masks: HOST array of the 4 forward pre-activation
 * maps in the same order (a1, q0, a0, p).  All four outputs are written (the weight gradients read them).
 */
int ppo_impala_stack_tail_backward_f32(const float *g, const float *const *packed_weights_t, const float *const *masks,
                                       float *da1, float *g1, float *da0, float *g0, int n_images, int channels,
                                       int h, int w, void *stream);

/*
 * A whole stack in one launch, for the stack whose input map fits LDS next to its pre-pool map (32 channels,
 * 21x21 -> 11x11): first convolution (no ReLU on read) + bias, 3x3 / stride 2 / pad 1 max-pool with argmax
 * (rl/impala.py:96-99; same tie / padding / NaN rule as ppo_maxpool3x3s2_forward_f32), then the two residual blocks as
 * ppo_impala_stack_tail_forward_f32.  packed_weights / biases: HOST arrays of 5 device pointers (firstconv,
 * block0.conv0, block0.conv1, block1.conv0, block1.conv1).  pooled [n,C,HO,WO], argmax (uint8) and a0, q0, a1 are
 * nullable (inference writes only q1).  Bit-identical to ppo_conv3x3_pool_forward_packed_f32 followed by the tail.
 */
/*
 * ppo_impala_stack_full_backward_f32: backward-data of that stack in one launch.  g = d loss / d q1 [n,C,HO,WO];
 * packed_weights_t: HOST array of 5 backward-data packed weights (block1.conv1, block1.conv0, block0.conv1,
 * block0.conv0, firstconv); masks: HOST array of 4 forward maps (a1, q0, a0, p).  Writes da1, g1, da0, g0 (as
 * ppo_impala_stack_tail_backward_f32), dc = ppo_maxpool3x3s2_backward_f32(g0, argmax) [n,C,h,w] and
 * g_prev = ppo_conv3x3_backward_data_packed_f32(dc, firstconv) [n,C,h,w], all bit-identical to those launches.
 */
int ppo_impala_stack_full_supported(int channels, int h, int w);
/* Chained form: the PREVIOUS stack's two residual blocks (on `in` = its pooled map [n,C,h,w]; pre_* as the tail
 * entry point's arguments, pre_a0 / pre_q0 / pre_a1 / pre_q1 nullable) followed by this whole stack, one launch. */
int ppo_impala_stack_chain_forward_f32(const float *in, const float *const *pre_packed_weights,
                                       const float *const *pre_biases, float *pre_a0, float *pre_q0, float *pre_a1,
                                       float *pre_q1, const float *const *packed_weights, const float *const *biases,
                                       float *pooled, uint8_t *argmax, float *a0, float *q0, float *a1, float *q1,
                                       int n_images, int channels, int h, int w, void *stream);
/* The chained form for SMALL inference batches (a rollout group: at most half as many images as the chip has CUs): every
 * image on TWO workgroups that split the output channels of the five convolutions on the h x w map and exchange their
 * halves after each (csrc/stack_fused.hip stack_chain_split_kernel); only q1 is written.  Bit-identical to
 * ppo_impala_stack_chain_forward_f32.  workspace: ppo_impala_stack_chain_split_workspace_bytes(n, channels, h, w) bytes,
 * 16-byte aligned, ZEROED once by the caller and then left alone (it carries the exchange flags, the launch counter and,
 * in its last 16 bytes, word 2 = an error flag set when a workgroup's partner never arrived within the spin bound). */
size_t ppo_impala_stack_chain_split_workspace_bytes(int n_images, int channels, int h, int w);
int ppo_impala_stack_chain_split_forward_f32(const float *in, const float *const *pre_packed_weights,
                                             const float *const *pre_biases, const float *const *packed_weights,
                                             const float *const *biases, float *q1, void *workspace, size_t workspace_bytes,
                                             int n_images, int channels, int h, int w, void *stream);
int ppo_impala_stack_full_backward_f32(const float *g, const float *const *packed_weights_t, const float *const *masks,
                                       const uint8_t *argmax, float *da1, float *g1, float *da0, float *g0, float *dc,
                                       float *g_prev, int n_images, int channels, int h, int w, void *stream);
int ppo_impala_stack_full_forward_f32(const float *in, const float *const *packed_weights, const float *const *biases,
                                      float *pooled, uint8_t *argmax, float *a0, float *q0, float *a1, float *q1,
                                      int n_images, int channels, int h, int w, void *stream);

/* ------------------------------------------------------------------------
 * Observation normalisation (`--observation_normalization`, rl/models.py:661-694): running per-feature
 * mean / variance (utils.RunningMeanStd, rl/utils.py:379-455) kept on the device in float64, and the
 * transform clamp((x - mu) / (std + eps), -5, 5) applied to the prepared observation (uint8 -> x/255,
 * float32 as is; rl/models.py:824-856) before either network.  x is [B, F] (F = product of input_dims).
 *
 * ppo_obs_moments_f64:     moments[0:F] = sum_b x, moments[F:2F] = sum_b x^2 in float64 (additive across
 *                          data-parallel ranks: all-reduce before the update).
 * ppo_obs_rms_update_f64:  RunningMeanStd.update_from_moments with batch mean / var taken from `moments`
 *                          over batch_count observations; `count` is the running count BEFORE this batch
 *                          (the caller adds batch_count afterwards).  Also refreshes the float32 constants
 *                          mu = mean, std = sqrt(float32(var))  (refresh_normalization_constants, :661-663).
 * ppo_obs_normalize_f32:   out [B, F] float32; bit-identical to the torch expression at :692.
 * ---------------------------------------------------------------------- */
int ppo_obs_moments_f64(const void *x, int is_u8, int B, int F, double *moments, void *stream);
int ppo_obs_rms_update_f64(const double *moments, double batch_count, double count, double *mean, double *var,
                           float *mu, float *std, int F, void *stream);
int ppo_obs_normalize_f32(const void *x, int is_u8, const float *mu, const float *std, float eps, float *out, int B,
                          int F, void *stream);

/* ------------------------------------------------------------------------
 * Synthetic vectorised environment (HOST pointers; runs on host threads).
 * The benchmark workload of SURVEY.md §8(d): obs uint8 i.i.d. uniform, reward ~ N(0,1),
 * done ~ Bernoulli(p_done), auto-reset; stands where the reference has the
 * HybridAsyncVectorEnv worker processes (rl/hybridVecEnv.py:49-203).  obs_out is a
 * caller-owned host buffer [n_envs, obs_bytes] (pinned, so the trainer can H2D it
 * asynchronously); every value depends only on (seed, env_offset + env, env step count).
 * actions[e] < 0 skips env e (rl/wrappers.py:1393-1418).  Per-env outputs other than
 * obs_out are nullable: reward [n] f32, done [n] u8, time [n] i32 (steps since reset, before
 * the auto-reset), ep_score [n] f32, ep_len [n] i32.
 * ---------------------------------------------------------------------- */
void *ppo_synth_env_create(int n_envs, int64_t obs_bytes, uint64_t seed, double p_done, int64_t env_offset,
                           int n_threads);
void ppo_synth_env_destroy(void *env);
int ppo_synth_env_reset(void *env, uint8_t *obs_out);
int ppo_synth_env_step(void *env, const int32_t *actions, uint8_t *obs_out, float *reward_out, uint8_t *done_out,
                       int32_t *time_out, float *ep_score_out, int32_t *ep_len_out);
/* ppo_synth_env_step + the upload of the new observations: the host-to-device copy of each 1 / n_chunks of the envs is
 * queued on `stream` as soon as it has been generated, while the worker threads generate the rest (north_star: "pinned
 * async obs copies into a GPU-resident rollout buffer").  obs_out: pinned host memory; obs_dev: device address of env 0's
 * observation ([n_envs, obs_bytes], e.g. a row slice of Runner.all_obs, rl/rollout.py:189-250). */
int ppo_synth_env_step_upload(void *env, const int32_t *actions, uint8_t *obs_out, float *reward_out, uint8_t *done_out,
                              int32_t *time_out, float *ep_score_out, int32_t *ep_len_out, void *obs_dev, int n_chunks,
                              void *stream);
/* Checkpointing, the counterpart of the worker envs' save_state / restore_state (rl/hybridVecEnv.py:84-105,
 * rl/utils.py:977-1038): per-env generator step count [n] i64, steps since reset [n] i32, running episode score
 * [n] f32.  set_state also rewrites obs_out [n_envs, obs_bytes] with the observation those counters imply. */
int ppo_synth_env_get_state(void *env, int64_t *steps_out, int32_t *time_out, float *score_out);
int ppo_synth_env_set_state(void *env, const int64_t *steps, const int32_t *time, const float *score, uint8_t *obs_out);

/*
 * The MLP networks of the continuous-control / classic configs as fused launches (csrc/mlp_fused.hip):
 *   StandardMLP  x -> fc1 -> tanh -> fc2 (rl/models.py:148-169) -> encoder activation tanh | relu (rl/models.py:456-467)
 *   -> all heads as one [NH, H] product (rl/models.py:364-384, 470-506).
 * ppo_mlp_forward_f32: one launch, heads [B, NH] (h_pre / hact [B, H] nullable: include_features).  Replaces the
 *   reference's `DualHeadNet.forward` for encoder "mlp" (rl/models.py:433-508) in rollouts and evaluations.
 * ppo_mlp_train_f32: forward + one of the four minibatch losses + backward into the gradient tensors, two launches
 *   (rows kernel, weight-gradient kernel); replaces forward / loss / loss.backward() of Runner.train_policy_minibatch,
 *   train_value_minibatch and train_distil_minibatch (rl/rollout.py:1610-1771, 1513-1567, 1331-1449) for these nets.
 *   `index` (nullable, [B] int32) maps minibatch row b to its row in the per-sample loss arrays and, with x_rows > 0, in
 *   x (x is then the whole batch of x_rows rows: the host fancy-indexing of rl/rollout.py:2349-2372 happens in the
 *   kernel's loads; x_rows = 0: x holds the B gathered rows).
 *   `workspace`: ppo_mlp_train_workspace_floats(B, F, H, NH) floats.  `stat_sums` (nullable): the column sums of the
 *   per-sample statistics rows (loss->stats, n_stats columns), added to the row when stat_accumulate.
 *   `partials` / `n_partials`: per-workgroup sums of g^2 over everything written (for ppo_adam_step_presummed_f32).
 *   Gradients of parameters the loss does not reach are exact zeros (the reference leaves them None); dlog_std is
 *   written by every loss kind (zeros unless gaussian).
 */
typedef struct ppo_mlp_net {
    const float *w1, *b1; /* fc1 [H, F], [H] */
    const float *w2, *b2; /* fc2 [H, H], [H] */
    const float *wh, *bh; /* heads [NH, H], [NH] or NULL */
    int F, H, NH;
    int act;              /* encoder activation behind fc2: 1 tanh, 2 relu */
} ppo_mlp_net;
typedef struct ppo_mlp_grads {
    float *dw1, *db1, *dw2, *db2, *dwh, *dbh; /* dbh NULL without head biases */
    float *dlog_std;                          /* [n_log_std] or NULL */
    int n_log_std;
} ppo_mlp_grads;
enum { PPO_MLP_LOSS_VALUE = 1, PPO_MLP_LOSS_DISTIL = 2, PPO_MLP_LOSS_GAUSSIAN = 3, PPO_MLP_LOSS_PPO = 4 };
typedef struct ppo_mlp_loss {
    int kind;         /* PPO_MLP_LOSS_* : the arguments of ppo_value_loss_f32 / ppo_distil_loss_f32 / ppo_gaussian_loss_f32 /
                         ppo_ppo_loss_f32, same meaning */
    float grad_scale;
    float *stats;     /* per-sample statistics rows of that loss kernel, nullable */
    int n_actions, n_value_heads;
    const float *returns;
    float vf_coef;
    /* value */
    int value_col, tvf_col, n_tvf, tvf_stride;
    const float *tvf_returns, *tvf_weights;
    float tvf_coef, tvf_keep_prob;
    uint64_t seed, offset;
    /* distil */
    int pred_col, n_pred, pred_stride, vector_targets;
    const float *targets, *weights, *old_policy, *log_std;
    float beta;
    /* gaussian / discrete policy */
    const float *actions_f;
    const int32_t *actions_i;
    const float *old_log_pac, *old_log_policy, *advantages;
    float eps_clip, ent_coef;
    float *dlog_std_rows; /* gaussian: [B, n_actions] scratch */
} ppo_mlp_loss;
int ppo_mlp_supported(int F, int H, int NH);
int ppo_mlp_forward_f32(const float *x, const ppo_mlp_net *net, const int32_t *index, int B, float *heads, float *h_pre,
                        float *hact, void *stream);
size_t ppo_mlp_train_workspace_floats(int B, int F, int H, int NH);
int ppo_mlp_train_f32(const float *x, const ppo_mlp_net *net, const ppo_mlp_grads *grads, const int32_t *index,
                      int64_t x_rows, int B, const ppo_mlp_loss *loss, float *workspace, float *heads, float *stat_sums,
                      int n_stats, int stat_accumulate, float *partials, int *n_partials, void *stream);

/*
 * OPT-IN reduced precision (csrc/stack_bf16x3.hip): the residual blocks of a 32-channel stack (the launch
 * ppo_impala_stack_tail_forward_f32 makes in exact float32) with every convolution as three bf16 MFMAs on (hi, lo)
 * splits of weights and activations, float32 accumulation - ~16-bit products.  Counterpart in the reference: its
 * `--precision` flag (train.py:166-178), whose default `medium` lets cuDNN run the convolutions in TF32 (10-bit
 * products); the default here, and the benchmark, stay exact float32.
 *   ppo_impala_stack_tail_pack_bf16x3   weights[4] (raw [32, 32, 3, 3] float32) -> packed
 *                                       (ppo_impala_stack_tail_bf16x3_packed_bytes() bytes).  Forward: block0.conv0,
 *                                       block0.conv1, block1.conv0, block1.conv1, transposed = 0.  Backward-data:
 *                                       block1.conv1, block1.conv0, block0.conv1, block0.conv0, transposed = 1.
 *   ppo_impala_stack_tail_forward_bf16x3   as ppo_impala_stack_tail_forward_f32: a0 / q0 / a1 nullable (inference)
 *   ppo_impala_stack_tail_backward_bf16x3  as ppo_impala_stack_tail_backward_f32 (masks: a1, q0, a0, p)
 * 32 channels at 21x21 or 11x11, 16 channels at 42x42 (ppo_impala_stack_tail_bf16x3_supported; the 16-channel form cuts
 * every image into two row windows that recompute a four-row halo - no exchange between workgroups).
 */
typedef struct ppo_split_pack_job {
    const float *weights[4]; /* raw [c, c, 3, 3] each, in the order the kernel walks its layers */
    void *packed;            /* ppo_impala_stack_tail_bf16x3_packed_bytes() bytes, 16-byte aligned */
    int channels;            /* 32 or 16 */
    int transposed;          /* 0 forward, 1 backward-data */
} ppo_split_pack_job;
size_t ppo_impala_stack_tail_bf16x3_packed_bytes(void);
int ppo_impala_stack_tail_bf16x3_supported(int channels, int h, int w);
int ppo_impala_stack_tail_pack_bf16x3(const float *const *weights, void *packed, int channels, int transposed, void *stream);
/* several packings in one launch (host table of at most 8 jobs): what follows every optimiser step in split mode */
int ppo_impala_stack_tail_pack_bf16x3_jobs(const ppo_split_pack_job *jobs /* host */, int n_jobs, void *stream);
int ppo_impala_stack_tail_forward_bf16x3(const float *in, const void *packed, const float *const *biases, float *a0, float *q0,
                                         float *a1, float *q1, int n_images, int channels, int h, int w, void *stream);
int ppo_impala_stack_tail_backward_bf16x3(const float *g, const void *packed_t, const float *const *masks, float *da1, float *g1,
                                          float *da0, float *g0, int n_images, int channels, int h, int w, void *stream);
/* The same two launches with the ReLU gates as SIGN MAPS: signs = host array of 4 device buffers [n, channels / 4, h, w] uint8
 * (ppo_impala_stack_tail_bf16x3_sign_bytes each), bit r of a byte = (channel 4 k + r of that pixel > 0), in the order of `masks`
 * (a1, q0, a0, p).  The forward launch writes them beside its float32 maps (which the weight gradients still read); the
 * backward launch reads them INSTEAD of the four float32 gate maps: 1 byte per 4 elements instead of 16 - its HBM traffic
 * drops from 9 maps to 5.  Same results as the float-gated launch, bit for bit. */
size_t ppo_impala_stack_tail_bf16x3_sign_bytes(int n_images, int channels, int h, int w);
int ppo_impala_stack_tail_forward_signs_bf16x3(const float *in, const void *packed, const float *const *biases, float *a0, float *q0,
                                               float *a1, float *q1, uint8_t *const *signs, int n_images, int channels, int h, int w,
                                               void *stream);
int ppo_impala_stack_tail_backward_signs_bf16x3(const float *g, const void *packed_t, const uint8_t *const *signs, float *da1, float *g1,
                                                float *da0, float *g0, int n_images, int channels, int h, int w, void *stream);

/* One 3x3 convolution (stride 1, zero padding 1) as split-bf16 products, where a convolution is not part of an LDS-resident
 * chain (csrc/conv_bf16x3.hip): the stack-first convolutions of rl/impala.py:96 (forward: bias, raw input) and their
 * backward-data form (the same operator on the transposed packing, bias null) under `--precision=medium|low`.
 *   in [n, cin, h, w] float32 (relu_in != 0: read through max(., 0)), out [n, cout, h, w] float32
 * cin / cout here are the OPERATOR's: for backward-data pass the layer's (cout, cin) and the packing made with
 * transposed = 1.  Geometries: ppo_conv3x3_bf16x3_supported. */
typedef struct ppo_conv_pack_job {
    const float *weight; /* the layer's raw [cout, cin, 3, 3] */
    void *packed;        /* ppo_conv3x3_bf16x3_packed_bytes(cin, cout) bytes, 16-byte aligned */
    int cin, cout;       /* the LAYER's channels (16 or 32 each) */
    int transposed;      /* 0: fragments of the forward operator, 1: of backward-data */
} ppo_conv_pack_job;
int ppo_conv3x3_bf16x3_supported(int cin, int cout, int h, int w);
size_t ppo_conv3x3_bf16x3_packed_bytes(int cin, int cout);
int ppo_conv3x3_pack_bf16x3_jobs(const ppo_conv_pack_job *jobs /* host, at most 8 */, int n_jobs, void *stream);
int ppo_conv3x3_bf16x3(const float *in, int relu_in, const void *packed, const float *bias, float *out, int n, int cin, int cout,
                       int h, int w, void *stream);
/* The same convolution followed by the 3x3 / stride 2 / pad 1 max-pool of rl/impala.py:105 in one launch: out and argmax are
 * [n, cout, (h+1)/2, (w+1)/2]; argmax (nullable) is ppo_maxpool3x3s2_forward_f32's record (winning tap ky * 3 + kx, ties to
 * the first tap), so ppo_maxpool3x3s2_backward_f32 reads it.  The pre-pool map never reaches HBM.  Forward packing only. */
int ppo_conv3x3_pool_bf16x3_supported(int cin, int cout, int h, int w);
int ppo_conv3x3_pool_bf16x3(const float *in, int relu_in, const void *packed, const float *bias, float *out, uint8_t *argmax, int n,
                            int cin, int cout, int h, int w, void *stream);
/* PROTOTYPE (measured for DESIGN.md section 7, not used by the host layer): the same convolution with n_split = 3 parts per
 * operand - (hi, mid, lo), 8 + 8 + 8 mantissa bits = the whole float32 significand - and six of the nine partial products
 * (the dropped ones <= 2^-24 of a product): float32-ACCURATE arithmetic on the bf16 MFMA, not bit-identical to the float32
 * MFMA.  n_split = 2 is ppo_conv3x3_bf16x3.  Three-part geometries: 16->32 and 32->16 at 42x42, 32->32 at 21x21. */
/* ... and the weight-gradient launch with n_split parts per operand (2 = ppo_conv3x3_backward_weight_slabs_batch_bf16x3; 3: the
 * Atari-shaped net's four geometries) */
int ppo_conv3x3_backward_weight_slabs_batch_bf16_split(const float *const *ins, const int *relu, const float *const *dys,
                                                       void *const *workspaces, size_t workspace_bytes, int count, int n, int cin,
                                                       int cout, int h, int w, int n_split, int *n_slabs, void *stream);
size_t ppo_conv3x3_bf16_split_packed_bytes(int cin, int cout, int n_split);
int ppo_conv3x3_pack_bf16_split(const float *weight, void *packed, int cin, int cout, int transposed, int n_split, void *stream);
int ppo_conv3x3_bf16_split(const float *in, int relu_in, const void *packed, const float *bias, float *out, int n, int cin, int cout,
                           int h, int w, int n_split, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPO_AMD_H */
