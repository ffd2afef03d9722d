/* socialways_hip.h - C ABI of the MI355X (gfx950) Social Ways hot path.
 *
 * The reference (crowdbotp/socialways) has no FFI of its own: its hot path is the Python call
 * surface of train.py.  This header is the boundary UNDER that surface: every
 * entry point below replaces the stock-PyTorch ops one reference function dispatches to, and the
 * Python glue in socialways_amd/ (same class / function names as the reference) calls exactly
 * these symbols through ctypes.  INTEGRATION.md shows the binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (int32 for scene offsets), 16-byte
 *     aligned; the caller owns all memory (the library never allocates);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 ok, -1 bad argument, -2 unsupported shape, -3 HIP error (sw_last_error());
 *   - "packed" weight buffers hold the tensors of the reference module's state_dict() in key
 *     order, row-major, each tensor starting on a 4-float boundary (sw_param_offset() is the
 *     authority; only Discriminator.classifier.2.bias (1 float) introduces padding);
 *   - B = agents in the packed batch, H = 64 hidden, Z = 32 noise, To / Tp = obs / pred length,
 *     scene_off = int32[S+1] prefix offsets of the scenes ("sub_batches" of train.py:446-461).
 *   - time-major workspaces ("gsave", "dsave", "*delta*") are opaque to the caller except for
 *     their sizes, which sw_workspace_floats() returns.
 */
#ifndef SOCIALWAYS_HIP_H
#define SOCIALWAYS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SW_OK 0
#define SW_EARG -1
#define SW_ESHAPE -2
#define SW_EHIP -3

#define SW_H 64
#define SW_Z 32

/* parameter groups for sw_param_offset / sw_param_count */
enum { SW_GRP_ENC = 0, SW_GRP_EMB = 1, SW_GRP_ATT = 2, SW_GRP_DEC = 3, SW_GRP_DISC = 4 };

/* workspace ids for sw_workspace_floats */
enum {
  SW_WS_GSAVE = 0,   /* generator forward saves (LSTM acts, x4, decoder acts, attention weights) */
  SW_WS_GDELTA = 1,  /* generator backward deltas                                                */
  SW_WS_DSAVE = 2,   /* discriminator forward saves (per call, nb pred branches)                 */
  SW_WS_DDELTA = 3,  /* discriminator backward deltas                                            */
  SW_WS_WGRAD = 4,   /* split-K partials of the weight-gradient GEMMs                            */
  SW_WS_PAIRS = 5    /* per-pair rows of the social block backward (P pairs)                     */
};

int sw_version(void);
const char* sw_last_error(void);

/* Packed-weight layout: number of floats of group `grp` and the float offset of its `idx`-th
 * state_dict tensor (Tp only matters for SW_GRP_DISC: pred_encoder.0.weight is (32, 4*Tp)). */
int sw_param_count(int grp, int Tp);
int sw_param_offset(int grp, int idx, int Tp);
int sw_param_tensors(int grp);
size_t sw_workspace_floats(int ws_id, int B, int To, int Tp, int nb, long long P);

/* ---- get_traj_4d (train.py:130-138) ------------------------------------------------------- */
int sw_traj_4d(const float* obsv /*[B,To,2]*/, const float* pred /*[B,Tp,2] or NULL*/, int B, int To,
               int Tp, float* obsv4 /*[B,To,4]*/, float* pred4 /*[B,Tp,4] or NULL*/, void* stream);

/* ---- EncoderLstm (train.py:245-269): Linear(4->64) + LSTM(64->64), time-unrolled ----------- */
/* x_mode 0: `x` = positions [B,T,2], the 4-d state is formed on the fly with the OBSERVATION rule
 * of get_traj_4d; x_mode 1: `x` = 4-d states [B,T,4].  h0/c0 NULL = zeros (predict(), :399-401).
 * y/act/x4s may be NULL.  act = [t0+T][B][384] rows (i f g o | c | h), x4s = [..][B][4]; the
 * kernel writes rows t0 .. t0+T-1.                                                             */
int sw_enc_lstm_fwd(const float* x, int x_mode, const float* enc_w, const float* h0, const float* c0,
                    int B, int T, float* hT, float* cT, float* y /*[B,T,64]*/, float* act,
                    float* x4s, int t0, void* stream);
/* Same, plus an auxiliary copy aux_src -> aux_dst (aux_n floats, multiple of 4; aux_src may be host-pinned memory)
 * done by extra workgroups of the same launch: the LSTM kernel is latency-bound and leaves CUs idle, so the
 * training step fetches z from its pinned slot here instead of in a kernel of its own.                        */
int sw_enc_lstm_fwd_aux(const float* x, int x_mode, const float* enc_w, const float* h0, const float* c0, int B, int T,
                        float* hT, float* cT, float* y, float* act, float* x4s, int t0, const float* aux_src,
                        float* aux_dst, long long aux_n, void* stream);
/* BPTT over rows t0+T-1 .. t0 of `act`; dhT/dcT = gradient w.r.t. the final state (dcT may be
 * NULL); dy optional [B,T,64].  Writes dgates rows [t][B][256]; dh0/dc0 optional outputs.       */
int sw_enc_lstm_bwd(const float* enc_w, const float* act, const float* c0, const float* dhT,
                    const float* dcT, const float* dy, int B, int T, int t0, float* dgates,
                    float* dh0, float* dc0, void* stream);

/* ---- SocialFeatures + EmbedSocialFeatures + AttentionPooling (train.py:153-241), fused,
 *      block-diagonal: only in-scene pairs are formed (identical math) -------- */
/* Scenes of up to SW_AMAX = 64 agents run one workgroup per scene (scores in LDS).  Larger scenes are
 * listed by the caller as blocks of 16 query agents, big_blocks = int32 [NB][8] records
 *   { scene, i0, partial row base of the block, partial row base of its scene, blocks in the scene,
 *     block index in the scene, 0, 0 }
 * (partial rows: 132 floats each, sum over big scenes of blocks*agents rows - used by the backward), and
 * run as row-block kernels with an online softmax; wh_ws (132 B floats) receives W h + b [B,64], then
 * v = W3^T (W h + b) [B,64] and c = <b3, W h + b> [B] - what the attention needs of the embedder's last layer fc.4
 * (weight W3, bias b3), which is never run per pair; ml [B,2] the softmax statistics (running max, normaliser)
 * the backward needs.  Amax = largest scene NOT in big_blocks.                                         */
int sw_social_pool_fwd(const float* obsv /*[B,To,2]*/, int To, const float* h /*[B,64]*/,
                       const int* scene_off, int S, int B, int Amax /*<= 64*/,
                       const float* emb_w, const float* att_w, float* S_out /*[B,64]*/,
                       float* attn /*[B,64] softmax weights (row i, column j_local) or NULL*/,
                       const int* big_blocks /*or NULL*/, int NB, float* wh_ws, float* ml /*or NULL*/, void* stream);
/* the same with an auxiliary copy aux_dst[0..aux_n) = aux_src[0..aux_n) (floats, aux_n % 4 == 0) done by extra
 * workgroups of the launch (the captured training step pulls z out of its pinned host slot here)          */
int sw_social_pool_fwd_aux(const float* obsv, int To, const float* h, const int* scene_off, int S, int B, int Amax,
                           const float* emb_w, const float* att_w, float* S_out, float* attn, const int* big_blocks,
                           int NB, float* wh_ws, float* ml, const float* aux_src, float* aux_dst, long long aux_n,
                           void* stream);
/* dense SocialFeatures (train.py:229-241) for the reference's module-level API on small batches */
int sw_social_features(const float* x4_last /*[B,4]*/, int B, float* feat /*[B,B,3]*/, void* stream);
/* EmbedSocialFeatures.forward on R rows of 3 features -> [R,64]; AttentionPooling.forward on a
 * dense (B,B,64) embedding tensor (only the in-scene blocks are read).                           */
int sw_embed_features(const float* feat /*[R,3]*/, long long R, const float* emb_w, float* out /*[R,64]*/,
                      void* stream);
int sw_attention_pool_dense(const float* f /*[B,B,64]*/, const float* h /*[B,64]*/, const int* scene_off,
                            int S, int B, const float* att_w, float* S_out /*[B,64]*/, void* stream);
/* ---- backward passes of the STAND-ALONE sub-modules (the reference's modules are ordinary nn.Modules: a user may
 *      compose AttentionPooling / EmbedSocialFeatures / EncoderLstm / DecoderFC differently from predict() and
 *      back-propagate through them, train.py:153-189, 245-269, 320-335).  Not on the training step's path. ------- */
/* y[r][n] (+)= bias[n] + sum_k x[r][k] w[k*w_rs + n*w_cs]  (w or w^T by the strides; bias may be NULL)              */
int sw_rows_gemm(const float* x, int ldx, const float* w, int w_rs, int w_cs, const float* bias, long long R, int K, int N,
                 float* y, int ldy, int accumulate, void* stream);
/* dW[N][K] (+)= delta^T act, db[N] (+)= column sums of delta (db may be NULL): one problem of the grouped weight-gradient
 * GEMM; wgrad_ws = sw_workspace_floats(SW_WS_WGRAD, ...) floats.  N <= 256; ldd, lda multiples of 4.               */
int sw_linear_wgrad(const float* delta, int ldd, const float* act, int lda, int R, int N, int K, float* dW, int ldw,
                    float* db, float* wgrad_ws, int accumulate, void* stream);
/* EmbedSocialFeatures backward on R rows: recomputes the MLP, leaves rows = h2 [R,64] | dh2 [R,64] | h1 [R,32] | dh1 [R,32] |
 * feat4 [R,4] (196 R floats) for sw_linear_wgrad (dout x h2, dh2 x h1, dh1 x feat4) and dfeat [R,3] (or NULL).        */
int sw_embed_features_bwd(const float* feat /*[R,3]*/, long long R, const float* emb_w, const float* dout /*[R,64]*/,
                          float* rows, float* dfeat, void* stream);
/* AttentionPooling on a dense (B,B,64) embedding tensor, ANY scene size (one workgroup per agent): wh = W h + b [B,64]
 * (sw_rows_gemm), attn [B,B] receives a_ij inside the scene blocks.  Backward: dsig [B,B] scratch, df [B,B,64] (or
 * NULL; zero outside the scene blocks on entry), dwh [B,64] = dL/d(Wh), dh [B,64] = sum_i a_ij dS_i (the caller adds
 * W^T dwh and forms dW, db from dwh and h).                                                                        */
int sw_attention_dense_fwd(const float* f, const float* h, const float* wh, const int* scene_off, int S, int B, float* attn,
                           float* S_out, void* stream);
int sw_attention_dense_bwd(const float* f, const float* h, const float* wh, const float* attn, const float* dS,
                           const int* scene_off, int S, int B, float* dsig, float* df, float* dwh, float* dh, void* stream);
/* Weight gradients of EncoderLstm over T steps from state (h0 or NULL = zeros): act / x4s as sw_enc_lstm_fwd left them,
 * dgates from sw_enc_lstm_bwd; tmp = 2048 floats.  Writes the whole packed gradient buffer d_enc_w.                 */
int sw_enc_lstm_wgrad(const float* enc_w, const float* act, const float* x4s, const float* h0, const float* dgates, int B,
                      int T, float* d_enc_w, float* wgrad_ws, float* tmp, void* stream);
/* Weight gradients of DecoderFC on one batch = a Tp = 1 rollout (gsave / gdelta of sw_dec_rollout_fwd / _bwd called
 * with this To and Tp = 1) with its inputs h, s (NULL = zeros), z; tmp = 2048 floats.  Writes the whole d_dec_w.   */
int sw_dec_fc_wgrad(const float* dec_w, const float* gsave, const float* gdelta, const float* h, const float* s,
                    const float* z, int B, int To, float* d_dec_w, float* wgrad_ws, float* tmp, void* stream);

/* ... and dz [B,32] = dL/dz of that rollout (predict() never needs it: z is data, train.py:473)                    */
int sw_dec_fc_dz(const float* dec_w, const float* gdelta, int B, int To, float* dz, void* stream);

/* pair_off = int64[S+1] prefix sums of n_s^2 over scenes with n_s > 1 (single-agent scenes own no
 * pair rows), P = pair_off[S].  dh is accumulated into.  d_emb_w / d_att_w are overwritten.
 * pair_ws holds sw_workspace_floats(SW_WS_PAIRS, ...) floats.                                    */
/* Host-side handle carrying weight-gradient problems from one call to a later launch (the grouped GEMM launches
 * are occupancy-bound: one launch per backward pass beats one per module).  Not thread-safe per handle.     */
typedef struct sw_wgrad_batch sw_wgrad_batch;
sw_wgrad_batch* sw_wgrad_batch_new(void);
void sw_wgrad_batch_free(sw_wgrad_batch* h);

int sw_social_pool_bwd(const float* obsv, int To, const float* h, const int* scene_off,
                       const long long* pair_off, int S, int B, int Amax, long long P,
                       const float* emb_w, const float* att_w, const float* attn, const float* dS,
                       float* dh, float* d_emb_w, float* d_att_w, float* pair_ws, float* wgrad_ws,
                       /* scenes above 64 agents (see sw_social_pool_fwd): the forward's big_blocks, wh_ws, ml, its
                        * output S_pool, and big_part_ws = 132 floats per partial row (blocks*agents rows per scene) */
                       const int* big_blocks /*or NULL*/, int NB, const float* wh_ws, const float* ml,
                       const float* S_pool, float* big_part_ws,
                       sw_wgrad_batch* defer /*NULL: reduce the weight gradients now; else leave the problems in the
                                               handle for the next sw_gen_wgrad on the same workspace*/,
                       void* stream);

/* ---- predict() decode loop (train.py:415-432): DecoderFC + position integration + the
 *      re-fed EncoderLstm step, Tp times, one persistent kernel (one workgroup per 16-agent tile; with the
 *      step's weight images registered and more than 256 tiles: per 32 agents, two column blocks against
 *      every register-resident weight operand - the same results bit for bit) ------------------------- */
int sw_dec_rollout_fwd(const float* obsv /*[B,To,2]*/, int To, const float* z /*[B,32]*/,
                       const float* S_pool /*[B,64] or NULL = zeros*/, const float* hT, const float* cT,
                       const float* enc_w, const float* dec_w, int B, int Tp,
                       float* pred4 /*[B,Tp,4]*/, float* h_end, float* c_end /*[B,64] or NULL*/,
                       float* gsave /*or NULL*/,
                       /* optional ADE/FDE partial sums of train.py:546-551, one triple per 16-agent tile
                        * (summed by the caller): { sum err / Tp, sum err[:, -1], sum err^2 },
                        * err = |(p_hat - gt) * inv_ss|; NULL to skip */
                       const float* gt /*[B,Tp,2]*/, float inv_ss, float* ade_part /*[ceil(B/16)][3]*/,
                       void* stream);
/* Same, plus (d_w, dsave non-NULL) the observation LSTM of Discriminator.forward (train.py:296-299) on the same
 * positions `obsv` with the packed D weights d_w, run by ceil(B/16) extra workgroups of the launch: the first D
 * pass of a training step (train.py:476-482) does not depend on the generator, and the rollout leaves CUs idle
 * for batches below ~2000 agents.  The LSTM rows land in `dsave` where sw_disc_fwd(save_lstm = 1) would put
 * them; the following sw_disc_fwd(obsv, x_mode = 0, same B / To / d_w, save_lstm = 2) reads them instead of
 * recomputing.                                                                                             */
int sw_dec_rollout_fwd_aux(const float* obsv, int To, const float* z, const float* S_pool, const float* hT,
                           const float* cT, const float* enc_w, const float* dec_w, int B, int Tp, float* pred4,
                           float* h_end, float* c_end, float* gsave, const float* gt, float inv_ss, float* ade_part,
                           const float* d_w /*or NULL*/, float* dsave /*or NULL*/, void* stream);
int sw_dec_rollout_bwd(const float* dpred4 /*[B,Tp,4]*/, const float* enc_w, const float* dec_w,
                       const float* gsave, int B, int To, int Tp, float* gdelta,
                       float* dhT, float* dcT, float* dS_pool /*[B,64]*/, void* stream);
/* Same, plus an auxiliary masked copy aux_dst[i] = aux_mask[i] > 0 ? aux_src[i] : aux_dst[i] (aux_n floats) done by
 * extra workgroups of the launch (idle CUs): the training step's D.load(backup) (train.py:541-542).          */
int sw_dec_rollout_bwd_aux(const float* dpred4, const float* enc_w, const float* dec_w, const float* gsave, int B,
                           int To, int Tp, float* gdelta, float* dhT, float* dcT, float* dS_pool,
                           const float* aux_src, float* aux_dst, const float* aux_mask, long long aux_n, void* stream);
/* weight gradients of the whole generator rollout (encoder + decoder) from gsave/gdelta.
 * part 0 = all; 1 = what dec_rollout_bwd produced (decoder layers + LSTM rows t >= To); 2 = LSTM rows
 * t < To (after enc_lstm_bwd) accumulated onto part 1, then the embed / W_ih split.  Parts 1 and 2
 * may run on different streams with different `wgrad_ws`; `tmp` (2048 floats) links them.        */
int sw_gen_wgrad(const float* enc_w, const float* dec_w, const float* gsave, const float* gdelta, const float* z,
                 const float* S_pool, int B, int To, int Tp, float* d_enc_w, float* d_dec_w, int part,
                 float* wgrad_ws, float* tmp /*[2048]*/,
                 sw_wgrad_batch* pending /*or NULL: problems deferred by sw_social_pool_bwd join this launch*/,
                 void* stream);
/* sw_gen_wgrad(part 0) that ALSO applies the generator's Adam update (train.py:539 `predictor_optimizer.step()`,
 * torch's fused Adam restated like sw_disc_bwd_gan_adam): every generator gradient is finished either by the
 * reduction of the grouped GEMM or by the composition kernel behind it, and the thread that finishes an element
 * updates exp_avg / exp_avg_sq / the weight.  adam_w / adam_m / adam_v / adam_g = the packed weight, moment and
 * gradient buffers of the WHOLE generator (adam_n floats each, one layout; enc_w, dec_w, d_enc_w, d_dec_w and the
 * pending problems' outputs point into them); adam_step = device scalar, 1-based index of the update.  Needs
 * the weight images of this step (sw_gen_images / sw_stage_step_img: the compositions read their step-start
 * snapshot while live weights are overwritten), else SW_EARG.  Single process only: a data-parallel job
 * all-reduces between gradient and update.                                                                */
int sw_gen_wgrad_adam(const float* enc_w, const float* dec_w, const float* gsave, const float* gdelta,
                      const float* z, const float* S_pool, int B, int To, int Tp, float* d_enc_w, float* d_dec_w,
                      float* wgrad_ws, float* tmp /*[2048]*/, sw_wgrad_batch* pending /*or NULL*/, float* adam_w,
                      float* adam_m, float* adam_v, const float* adam_g, long long adam_n, const float* adam_step,
                      double lr, double beta1, double beta2, double eps, void* stream);

/* ---- Discriminator.forward (train.py:294-309) for nb prediction branches sharing one
 *      observation encoding (fake / real of the same batch) ----------------------------------- */
/* x_mode 0: obsv = positions [B,To,2] (4-d state formed on the fly); 1: obsv = obsv_4d [B,To,4].   */
int sw_disc_fwd(const float* obsv, int To, int x_mode, const float* const* pred4 /*nb x [B,Tp,4]*/,
                int nb, const float* d_w, int B, int Tp, float* const* label /*nb x [B,1]*/,
                float* const* code /*nb x [B,2]*/, float* dsave /*or NULL*/,
                int save_lstm /*0: head activations only - enough for a backward that wants d/dpred only; 1: + the
                                LSTM rows; 2: the LSTM rows are ALREADY in dsave (sw_dec_rollout_fwd_aux): the
                                observation LSTM is not run again, h_T is read from them*/,
                float* w_snapshot /*or NULL: receives a copy of d_w (sw_param_count floats) - deepcopy(D), train.py:499*/,
                void* stream);
/* dlabel/dcode: nb x gradients of the loss w.r.t. label / code.  d_d_w NULL = skip weight grads
 * (generator phase), dpred4[k] NULL = skip input grad of branch k.                              */
int sw_disc_bwd(const float* d_w, const float* dsave, const float* const* dlabel,
                const float* const* dcode, int nb, int B, int To, int Tp, float* ddelta,
                float* d_d_w, float* const* dpred4, float* wgrad_ws, void* stream);

/* Same backward with the LSGAN / InfoGAN loss gradients formed inside the kernel from the forward
 * outputs: dlabel_k = 2 (label_k - targets[t_k]) g_label, dcode_0 = 2 (code_0 - z[:, :2]) g_code,
 * dcode_1 = 0 (train.py:484-494, 512-523) - no separate loss kernel on the critical path.          */
int sw_disc_bwd_gan(const float* d_w, const float* dsave, const float* const* label,
                    const float* const* code, const float* targets, int t0, int t1, const float* z /*[B,32]*/,
                    float g_label, float g_code, int nb, int B, int To, int Tp, float* ddelta,
                    float* d_d_w, float* const* dpred4, float* wgrad_ws,
                    float* loss_part /*[ceil(B/16)][3] or NULL: per-tile sums {(label_0-t0)^2, (code_0-z)^2,
                                       (label_1-t1)^2}, the reported MSE terms; column 2 untouched if nb == 1*/,
                    void* stream);
/* The same with the discriminator's Adam update (train.py:384-385: lr, betas, eps; no weight decay) applied by the
 * kernel that finishes the gradients: every discriminator parameter is an output of that reduction, so its thread
 * updates exp_avg, exp_avg_sq and the weight right there - operation for operation torch's fused Adam
 * (fused_adam_utils.cuh) - and the optimizer's own launch disappears.  adam_w must be d_w (the packed weights the pass
 * just read: all of its readers are behind kernel boundaries), adam_m / adam_v the packed moments, adam_step a device
 * scalar holding the 1-based index of this update.  adam_w = NULL: plain sw_disc_bwd_gan.  Single process only: data
 * parallel ranks all-reduce the gradients between this call and their optimizer step.                             */
int sw_disc_bwd_gan_adam(const float* d_w, const float* dsave, const float* const* label, const float* const* code,
                         const float* targets, int t0, int t1, const float* z, float g_label, float g_code, int nb,
                         int B, int To, int Tp, float* ddelta, float* d_d_w, float* const* dpred4, float* wgrad_ws,
                         float* loss_part, float* adam_w, float* adam_m, float* adam_v, const float* adam_step,
                         double lr, double beta1, double beta2, double eps, void* stream);

/* ---- one discriminator UPDATE pass in one launch (train.py:476-495): sw_disc_fwd(nb = 2: fake, real; x_mode 0;
 *      save_lstm = obs_pre ? 2 : 1; w_snapshot) + sw_disc_bwd_gan[_adam](no d/dpred) fused per 16-agent tile - forward,
 *      LSGAN / InfoGAN loss gradients, backward, the two branches' heads side by side on the two wave pairs - followed by
 *      the weight-gradient GEMM (and the Adam update when adam_w = d_w).  Same buffers, same rows, same arguments as the
 *      two calls.  For the shapes that leave CUs idle: sw_disc_update_supported() says whether this (d_w with registered
 *      images - sw_disc_images -, B <= 2048, Tp <= 12) can run; SW_ESHAPE otherwise.                                 */
int sw_disc_update_supported(const float* d_w, int B, int To, int Tp);
int sw_disc_update(const float* obsv /*[B,To,2]*/, int To, const float* const* pred4 /*2 x [B,Tp,4]*/, const float* d_w, int B,
                   int Tp, float* const* label, float* const* code, float* dsave, int obs_pre, float* w_snapshot /*or NULL*/,
                   const float* targets, int t0, int t1, const float* z, float g_label, float g_code, float* ddelta,
                   float* d_d_w, float* wgrad_ws, float* loss_part /*or NULL*/, float* adam_w /*or NULL*/, float* adam_m,
                   float* adam_v, const float* adam_step, double lr, double beta1, double beta2, double eps, void* stream);

/* ---- generator phase in one launch (train.py:510-523, 538): D forward on (obsv, pred_hat) fused with the backward of
 *      its prediction heads: dpred4 = d(g_loss)/d(pred_hat) with g_loss = mse(label, targets[t_idx]) +
 *      w mse(code, z[:, :2]) expressed through g_label = 1/B_global, g_code = w/(2 B_global).  No saves; label / code /
 *      loss_part ([ceil(B/16)][3]: columns 0, 1 = per-tile sums of the squared errors) are optional outputs.   */
int sw_disc_dpred(const float* obsv, int To, int x_mode, const float* pred4 /*[B,Tp,4]*/, const float* d_w, int B, int Tp,
                  const float* targets, int t_idx, const float* z /*[B,32]*/, float g_label, float g_code,
                  float* dpred4 /*[B,Tp,4]*/, float* label /*[B,1] or NULL*/, float* code /*[B,2] or NULL*/,
                  float* loss_part /*or NULL*/, void* stream);

/* ---- LSGAN + InfoGAN losses of train.py:484-494 / 512-523 and their gradients -------------- */
/* t_a = targets[ia], t_b = targets[ib] (read on the device, so a captured hipGraph sees new values).
 * out_sums[3] = { sum (label_a - t_a)^2, sum (code_a - z[:, :2])^2, sum (label_b - t_b)^2 } over the
 * B local rows (label_b may be NULL).  Gradients (any may be NULL):
 *   dlabel_x = 2 (label_x - t_x) g_label,  dcode_a = 2 (code_a - z) g_code,  dcode_b = 0
 * with g_label = 1/B_global and g_code = loss_info_w / (2 B_global) for a mean over the global
 * batch (data-parallel ranks pass the GLOBAL batch size so summed gradients are exact).
 * `scratch` (3*SW_RED_BLOCKS floats, optional): with it, large batches are reduced by up to
 * SW_RED_BLOCKS workgroups and a fixed-order second stage; NULL keeps the one-workgroup path.   */
#define SW_RED_BLOCKS 64
int sw_gan_loss(const float* label_a, const float* targets /*device [>=2]: label-noise scalars*/, int ia,
                const float* code_a, const float* z /*[B,32]*/, const float* label_b, int ib, int B,
                float g_label, float g_code, float* out_sums /*[3]*/, float* dlabel_a, float* dcode_a,
                float* dlabel_b, float* dcode_b, float* scratch /*[3*SW_RED_BLOCKS] or NULL*/, void* stream);

/* ---- optional L2 term of the generator loss (train.py:512,525-526; the "variety" term as written in
 *      train.py:527-536 is the same expression restricted to one agent row):
 *      dpred4[b][t][0:2] += scale * (pred4_hat[b][t][0:2] - gt[b][t][:]) for rows b in [row0,row1).
 *      L2: rows [0,B), scale = loss_l2_w / (B_global * Tp).  Called after sw_disc_bwd_gan wrote dpred4. */
int sw_l2_grad(const float* pred4_hat /*[B,Tp,4]*/, const float* gt /*[B,Tp,2]*/, int B, int Tp, int row0, int row1,
               float scale, float* dpred4 /*[B,Tp,4]*/, void* stream);

/* ---- variety loss with its INTENDED semantics (train.py:527-536 is buggy as written - see SURVEY 0.x / DESIGN):
 *      K rollouts with independent z folded into one batch (copy k = rows [k*B,(k+1)*B)); per agent the copy with
 *      the smallest mean squared error receives dpred4[k*B+b][t][0:2] += scale * (p_hat - p), scale =
 *      loss_l2_w / (B_global * Tp).  kmin [B] (int32) / l2min [B] (the per-agent minimum, for the reported loss)
 *      may be NULL.  K <= 64 (SW_ESHAPE above).                                                              */
int sw_variety_grad(const float* pred4_hat_K /*[K*B,Tp,4]*/, const float* gt /*[B,Tp,2]*/, int K, int B, int Tp,
                    float scale, float* dpred4_K /*[K*B,Tp,4]*/, int* kmin, float* l2min, void* stream);

/* ---- toy statistics (calc_statistics.py:7-66): the O(K^2) distance loops of compute_1nn /
 *      compute_wasserstein.  D[k][i][j] = mean over t in [t0,T) of ||a[i][k][t] - b[j][k][t]||.        */
int sw_traj_dist(const float* a /*[Na,nPed,T,2]*/, const float* b /*[Nb,nPed,T,2]*/, int Na, int Nb, int nPed, int T,
                 int t0, float* D /*[nPed,Na,Nb]*/, void* stream);

/* ---- input staging of a hipGraph-replayed step, one kernel with fixed arguments.  `slot` = host-pinned
 *      (device-mapped) words the host rewrites before each replay: [0,1] device pointer of obsv (B,To,2),
 *      [2,3] device pointer of pred (B,Tp,2), [4] zeros_val, [5] ones_val, [6] / [7] number of D / G Adam
 *      updates applied so far, [8..] z (B*32).
 *      Writes the static buffers of the graph: tracks, real future as (p,v) rows (train.py:135-137),
 *      label-noise scalars, z (z_dst may be NULL: fetched elsewhere) and (steps_dst != NULL) the 1-based Adam step indices of this step's
 *      n_d_updates discriminator updates followed by the generator update. -------------------------- */
#define SW_STAGE_HEADER 8
int sw_stage_step(const float* slot, int B, int To, int Tp, float* obsv_dst /*[B,To,2]*/, float* pred_dst /*[B,Tp,2]*/,
                  float* pred4_dst /*[B,Tp,4]*/, float* targets_dst /*[2]*/, float* z_dst /*[B,32]*/,
                  float* steps_dst /*[n_d_updates+1] or NULL*/, int n_d_updates, void* stream);

/* ---- derived weight images of the generator.  Every workgroup of the encoder / decode launches needs the composed input
 *      matrix W_ih W_embed (train.py:266-268: no non-linearity between embed and the LSTM), fc4 . fc3 and every weight
 *      matrix (transposed in the backward pass) as MFMA A operands.  sw_gen_images derives them ONCE into img
 *      (sw_gen_image_floats() floats) - the compositions, a step-start snapshot of the raw weights behind them, and
 *      OPERAND-LAYOUT images (a wave's operand load = 1 KB of consecutive memory instead of 64 cache-line accesses) -
 *      and REGISTERS them for (enc_w, dec_w) and, when given, for the social block's (emb_w, att_w): until the
 *      registration is dropped, sw_enc_lstm_fwd*, sw_dec_rollout_fwd*, sw_dec_rollout_bwd*, sw_social_pool_fwd/bwd
 *      called with these weight buffers load the images instead of deriving / gathering per workgroup (same values bit
 *      for bit).  The images are valid while the weights are unchanged: the caller drops the registration -
 *      sw_gen_images(NULL, NULL, NULL, NULL, NULL, NULL) - before it updates them.  Registrations (these and
 *      sw_disc_images') are per HOST THREAD: register, launch and drop on the thread that steps the trainer.
 *      sw_stage_step_img = sw_stage_step whose launch derives and registers the images as well (no extra launch).   */
int sw_gen_image_floats(void);
int sw_gen_images(const float* enc_w, const float* dec_w, const float* emb_w /*or NULL*/, const float* att_w /*or NULL*/,
                  float* img, void* stream);
int sw_stage_step_img(const float* slot, int B, int To, int Tp, float* obsv_dst, float* pred_dst, float* pred4_dst,
                      float* targets_dst, float* z_dst, float* steps_dst, int n_d_updates, const float* enc_w,
                      const float* dec_w, const float* emb_w /*or NULL*/, const float* att_w /*or NULL*/, float* img,
                      /* d_img != NULL: the launch also scatters and registers the discriminator's images (sw_disc_images) */
                      const float* d_w /*or NULL*/, float* d_img /*or NULL*/, const int* d_tab /*or NULL*/, void* stream);
/*      sw_stage_step_zdev = sw_stage_step_img for a caller whose z (train.py:473) already lives in device memory
 *      (z_device = 1): slot words [8,9] then hold the device pointer of z (B,32) instead of its values, z_dst is required
 *      and filled from there - no PCIe read of z inside the step (4 MB at 32 768 agents).                               */
int sw_stage_step_zdev(const float* slot, int B, int To, int Tp, float* obsv_dst, float* pred_dst, float* pred4_dst,
                       float* targets_dst, float* z_dst, float* steps_dst, int n_d_updates, const float* enc_w,
                       const float* dec_w, const float* emb_w, const float* att_w, float* img, const float* d_w, float* d_img,
                       const int* d_tab, int z_device, void* stream);

/* ---- derived weight images of the DISCRIMINATOR (Discriminator.forward, train.py:294-309, as the kernels consume it):
 *      MFMA A-operand images of lstm.weight_hh and its transpose, and the eight head matrices transposed and zero-padded
 *      exactly as the backward kernels keep them in LDS (their prologue becomes one contiguous copy).  D's weights change
 *      three times per training step (two Adam updates, D.load(backup)), so the images are maintained element-wise through
 *      a table: sw_disc_image_table(Tp, tab) fills HOST memory with 2 ints per packed float (sw_param_count(SW_GRP_DISC, Tp)
 *      pairs): its image offsets, -1 = none.  sw_disc_images(d_w, img, tab_device, Tp, stream) scatters the whole packed
 *      buffer into img (sw_disc_image_floats(Tp) floats, ZERO-FILLED by the caller once - padding is never written) and
 *      REGISTERS (img, tab) for d_w: until the registration is dropped - sw_disc_images(NULL, NULL, NULL, 0, NULL) -
 *      sw_disc_fwd / sw_disc_dpred / sw_disc_bwd* called with these weights and this Tp read the images, and
 *      sw_disc_bwd_gan_adam / sw_adam_packed keep them current while they update the weights.  Whoever changes the
 *      weights by other means re-scatters or drops the registration.  Same values bit for bit either way.           */
int sw_disc_image_floats(int Tp);
int sw_disc_image_table(int Tp, int* tab_host);
int sw_disc_images(const float* d_w, float* img, const int* tab, int Tp, void* stream);

/* ---- GENERIC-WIDTH path: the per-layer pieces from which socialways_amd/generic.py runs the same model for any
 *      `--hidden-size` (train.py:42-44, 76-81: encoder / social-feature / discriminator widths H, noise H/2) and any
 *      latent-code count (train.py:65).  The fused kernels above hold a 64-unit layer per workgroup in registers and are
 *      the path of every BASELINE config; H > 64 or n_latent_code != 2 run layer by layer through these entry points
 *      plus sw_rows_gemm / sw_linear_wgrad - same mathematics, launch-bound.  All buffers row-major fp32.        */
/* nn.LSTM cell, element-wise part: pre [B][4H] = gate pre-activations (i | f | g | o blocks); gates receives the
 * activated gates, c / h [B][H] the new state; c_prev NULL = zeros.  Backward: dpre [B][4H], dc_prev [B][H] from
 * dh / dc (either may be NULL = zeros).                                                                       */
int sw_lstm_point_fwd(const float* pre, const float* c_prev, int B, int H, float* gates, float* c, float* h, void* stream);
int sw_lstm_point_bwd(const float* gates, const float* c, const float* c_prev, const float* dh, const float* dc, int B,
                      int H, float* dpre, float* dc_prev, void* stream);
/* kind 0: ReLU, 1: LeakyReLU(0.2) (train.py:181-188, 280-292, 324-328); the backward takes the activated values */
int sw_act_fwd(const float* x, long long n, int kind, float* y, void* stream);
int sw_act_bwd(const float* y, const float* dy, long long n, int kind, float* dx, void* stream);
/* sum over an [R][C] block of (a - b)^2 (b NULL: the scalar target[target_idx], read on the device) into out_sum[0]
 * (or NULL), and da = gscale (a - b) (or NULL): the pieces of nn.MSELoss and its gradient (train.py:484-494, 512-523) */
int sw_sqdiff(const float* a, int lda, const float* b, int ldb, const float* target, int target_idx, long long R, int C,
              float gscale, float* out_sum, float* da, int ldda, void* stream);
/* SocialFeatures (train.py:208-241) on the ordered in-scene pairs only: feat [P][4] = (dist, bearing, dca, 0) at row
 * pair_off[s] + i_local n + j_local (pair_off as for sw_social_pool_bwd); x4_last [B][4] = last observed (p, v)   */
int sw_pair_features(const float* x4_last, const int* scene_off, const long long* pair_off, int S, float* feat, void* stream);
/* AttentionPooling (train.py:153-175) on pair rows f [P][F] with wh = W h + b [B][F], h [B][H]: attn [P] receives the
 * softmax weights, S_out [B][H] the pooled states.  Backward: dsig [P] scratch, df [P][F] (or NULL), dwh [B][F] =
 * dL/d(wh), dh [B][H] = sum_i a_ij dS_i (the caller adds the W^T dwh term and forms dW, db).                    */
int sw_attn_pairs_fwd(const float* f, const float* wh, const float* h, const int* scene_off, const long long* pair_off, int S,
                      int B, int F, int H, float* attn, float* S_out, void* stream);
int sw_attn_pairs_bwd(const float* f, const float* wh, const float* h, const float* attn, const float* dS, const int* scene_off,
                      const long long* pair_off, int S, int B, int F, int H, float* dsig, float* df, float* dwh, float* dh,
                      void* stream);

/* ---- Adam on a packed buffer (train.py:379-385: lr, betas, eps; no weight decay), torch's fused Adam restated operation
 *      by operation (the arithmetic of sw_disc_bwd_gan_adam / sw_gen_wgrad_adam as a kernel of its own): the optimizer
 *      step of data-parallel ranks, whose gradients pass through an all-reduce first.  step = device scalar, 1-based
 *      index of the update.  disc_Tp > 0: w is the packed Discriminator and its registered images (sw_disc_images)
 *      follow the update; 0 otherwise.                                                                          */
int sw_adam_packed(float* w, const float* g, float* m, float* v, long long n, const float* step, double lr, double beta1,
                   double beta2, double eps, int disc_Tp, void* stream);

/* sw_enc_lstm_bwd with the masked copy of sw_dec_rollout_bwd_aux (dst[i] = mask[i] > 0 ? src[i] : dst[i]) done by extra
 * workgroups of the launch                                                                                            */
int sw_enc_lstm_bwd_aux(const float* enc_w, const float* act, const float* c0, const float* dhT, const float* dcT,
                        const float* dy, int B, int T, int t0, float* dgates, float* dh0, float* dc0, const float* aux_src,
                        float* aux_dst, const float* aux_mask, long long aux_n, void* stream);
/* sw_disc_dpred (generator phase of train.py:510-523: D forward on obsv [B,To,2] / pred4 + the backward of its prediction
 * heads, loss gradients formed in the kernel, per-tile loss sums to loss_part) and sw_dec_rollout_bwd in ONE launch: the
 * pass is tile-local and the decode BPTT of the same 16 agents is its only consumer.  dpred4 [B,Tp,4] = scratch that
 * receives d(g_loss)/d(pred4).  Same results as the two calls, bit for bit.                                           */
int sw_dec_rollout_bwd_dfuse(const float* obsv, const float* pred4, const float* d_w, const float* targets, int t_idx,
                             const float* z, float g_label, float g_code, float* loss_part, float* dpred4, const float* enc_w,
                             const float* dec_w, const float* gsave, int B, int To, int Tp, float* gdelta, float* dhT, float* dcT,
                             float* dS_pool, void* stream);

/* ---- WIDE path (socialways_amd/wide.py, csrc/sw_wide.hip): the model at hidden sizes H > 64, H % 32 == 0 (WideTrainer.supports; train.py:42-44,
 *      76-81) as one launch per LSTM step and per decoder / head layer over all agents, explicit backward, deferred
 *      weight gradients.  Replaces, per call, the stock-PyTorch ops behind nn.Linear / nn.LSTM / nn.LeakyReLU / nn.ReLU of
 *      train.py:153-335 (forward and autograd backward).  All buffers row-major fp32, row strides in floats.            */
/* y[r][n] = epi(sum_k x[r][k] w[n][k] + bias[n] + cin[r][n]; aux[r][n]); x element (r,k) at x + r*x_rs + k*x_cs, w element
 * (n,k) at w + n*w_rs + k*w_cs; epi 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 / 4: multiply by ReLU' / LeakyReLU'(0.2) taken from
 * the sign of aux (the layer's saved activation).  bias / cin / aux may be NULL.                                         */
int sw_wide_gemm(const float* x, long long x_rs, int x_cs, const float* w, long long w_rs, int w_cs, const float* bias,
                 const float* cin, int cin_ld, const float* aux, int aux_ld, long long R, int K, int N, float* y, int y_ld, int epi,
                 void* stream);
/* one nn.LSTM step on all B agents: pre = Wx x4 + b1 (+ b2) + Whh h_prev (Wx [4H][4], Whh [4H][H], gate blocks i f g o),
 * gates [B][4H] = activated gates, c_out / h_out the new state (h also to h_out2 if given); h_prev / c_prev NULL = zeros */
int sw_wide_lstm_fwd(const float* x4, int x_ld, const float* h_prev, int hp_ld, const float* c_prev, const float* Wx,
                     const float* b1, const float* b2, const float* Whh, int B, int H, float* gates, float* c_out, float* h_out,
                     int h_ld, float* h_out2, int h2_ld, void* stream);
/* its backward: dh = dh_ext + dh_ext2 + dg_next WhhT^T (WhhT [H][4H]; any of the three may be NULL), cell backward with the
 * saved gates / c / c_prev -> dgates [B][4H] (pre-activation gradients) and dc_out [B][H]                               */
int sw_wide_lstm_bwd(const float* dh_ext, int dhe_ld, const float* dh_ext2, int dhe2_ld, const float* dg_next, const float* WhhT,
                     const float* gates, const float* c, const float* c_prev, const float* dc_in, int B, int H, float* dgates,
                     float* dc_out, void* stream);
/* last decoder layer + integration of a decode step (train.py:330, 422-424): v = a3 W4^T + b4 (W4 [2][D3]), p += v,
 * pred4_i[b*pred_ld] = x4_tm[b*4] = (p, v).  Backward of the step: dx4 = dg WxT^T (dg [B][H4] = dgates of the re-fed encoder
 * step or NULL, WxT [4][H4]), dp_run += dpred.p + dx4.p, dv[b] = (dpred.v + dx4.v + dp_run, 0, 0), dz3 [B][D3] = dv W4       */
int sw_wide_out_fwd(const float* a3, int D3, const float* W4, const float* b4, float* p, int B, float* pred4_i, int pred_ld,
                    float* x4_tm, void* stream);
int sw_wide_out_bwd(const float* dpred4_i, int pred_ld, const float* dg, const float* WxT, int H4, float* dp_run, int B, float* dv,
                    const float* W4, int D3, float* dz3, void* stream);
/* out[r][c] = sum_t in[t*t_stride + r*in_ld + c] */
int sw_wide_sum_steps(const float* in, long long t_stride, int in_ld, int T, long long R, int C, float* out, int out_ld,
                      void* stream);
/* transposed copies of ntab matrices of one packed buffer: tab (device, ntab x {src offset, rows, cols, dst offset} int32, in
 * floats), total_tiles = sum over the matrices of ceil(rows/32) ceil(cols/32); dst[dst_off + c*rows + r] = src[src_off + r*cols + c] */
int sw_wide_transpose(const float* src, const int* tab, int ntab, int total_tiles, float* dst, void* stream);
/* MFMA operand images of ntab matrices of one packed buffer (tab: device, ntab x {src offset, R, K, dst offset, transposed,
 * 0} int32; the image of Mx [R][K] - M or, transposed, M^T of M [K][R] - holds the float4 Mx[16t + (l & 15)][16j + 4(l >> 4)..]
 * at dst offset + ((t K/16 + j) 64 + l) 4); total_float4 = sum of R K / 4                                              */
int sw_wide_opimage(const float* src, const int* tab, int ntab, long long total_float4, float* dst, void* stream);
/* LSTM SEQUENCE kernels (hidden sizes 64 and 128: sw_wide_lstm_seq_supported): T steps of sw_wide_lstm_fwd / _bwd in one
 * launch per 16-agent tile, W_hh / W_hh^T register-resident from their operand images.  x4 [T][B][4], gates [T][B][4H],
 * cs [T][B][H], hs [T+1][B][H] (slab 0 = h_0, c_0 = 0), h_last2 = optional second copy of h_T.  Backward: dh_ext (+ dh_ext2)
 * = gradient w.r.t. h_{T-1} from outside, dg_init [B][4H] = dgates of the step behind the sequence or NULL, dc_init [B][H]
 * = gradient w.r.t. c_{T-1} from that step or NULL; writes dgates [T][B][4H].                                          */
int sw_wide_lstm_seq_supported(int H);
int sw_wide_lstm_seq_fwd(const float* x4, const float* Wx, const float* b1, const float* b2, const float* whh_img, int B, int H, int T,
                         float* gates, float* cs, float* hs, float* h_last2, int h2_ld, void* stream);
int sw_wide_lstm_seq_bwd(const float* dh_ext, int dhe_ld, const float* dh_ext2, int dhe2_ld, const float* dg_init, const float* dc_init,
                         const float* whhT_img, const float* gates, const float* cs, int B, int H, int T, float* dgates, void* stream);
/* the decode loop of predict() (train.py:415-432) at 128 hidden units (sw_wide_dec_loop_supported) as ONE launch per 16-agent
 * tile, streaming the operand images of W1[:, :H], W2, W3 and W_hh per step: replaces Tp x {3 sw_wide_gemm, sw_wide_out_fwd,
 * sw_wide_lstm_fwd}.  u [B][2.5H] = W1[:, H:] [S; z] + b1; p0 = last observed positions; leaves a1 / a2 / a3 [Tp][B][.],
 * pred4 [B][Tp][4], x4 rows To.., gates / cs rows To.., hs rows To + 1.., h into cat[i + 1][:, :H] (row stride 2.5H).    */
int sw_wide_dec_loop_supported(int H);
int sw_wide_dec_loop_fwd(const float* w1h_img, const float* w2_img, const float* w3_img, const float* whh_img, const float* u,
                         const float* b2, const float* b3, const float* W4, const float* b4, const float* Wx, const float* bx1,
                         const float* bx2, const float* p0, int p0_ld, float* a1, float* a2, float* a3, float* pred4, float* x4,
                         float* gates, float* cs, float* hs, float* cat, int B, int H, int To, int Tp, void* stream);
/* ... and its backward (data gradients; Tp x {sw_wide_lstm_bwd, sw_wide_out_bwd, 3 sw_wide_gemm} in one launch): images of W_hh^T,
 * W3^T, W2^T, W1[:, :H]^T and of Wx^T zero-padded to 16 rows; leaves dgates rows To.., dv / dz3 / dz2 / dz1 [Tp][B][.], the gradient
 * w.r.t. h_{To-1} from decode step 0 (dhcat_out) and w.r.t. c_{To-1} (dc_out), both [B][H].                              */
int sw_wide_dec_loop_bwd(const float* whhT_img, const float* w3T_img, const float* w2T_img, const float* w1hT_img,
                         const float* wxT_img, const float* W4, const float* dpred4, const float* a1, const float* a2,
                         const float* gates, const float* cs, float* dgates, float* dv, float* dz3, float* dz2, float* dz1,
                         float* dhcat_out, float* dc_out, int B, int H, int To, int Tp, void* stream);
/* the heads of the discriminator (train.py:280-292, 300-309) for all agents and both future branches in ONE launch per
 * direction (instead of nine / seven sw_wide_gemm launches): p = 52 host values in the order {6 operand images (forward: of0,
 * of1, pe0, pe1, cl0, la0; backward: their transposes), 6 biases, cl1 weight, cl1 bias, la1 weight, la1 bias, hT [B][H], px
 * [nb B][4Tp], outputs o1, q1, both, c1, l1, label, code, inputs dlab [nb B][4], dcod [nb B][nlp], deltas dc1, dl1, dboth, dq1,
 * docode, do1, dhT, dpx, then B, H, 4Tp, nb, nl, nlp, need_obs, want_dpred, then the loss block of the forward launch: loss
 * (0 / 1), target index of branch 0 / 1, row stride of z, the two gradient scales (as the bit patterns of doubles), targets,
 * z, part [tiles][3]}: with loss = 1 the forward launch also forms the LSGAN / info-loss gradients dlab / dcod (train.py:484-494,
 * 512-523) and each tile's sums of squares {branch-0 label, branch-0 code_hat, branch-1 label}.                        */
int sw_wide_disc_heads_supported(int H, int K4, int nl);
int sw_wide_disc_heads_fwd(const long long* p, void* stream);
int sw_wide_disc_heads_bwd(const long long* p, void* stream);
/* n weight-gradient problems dW[N][K] = delta^T act, db = column sums (desc: n x {delta, ldd, act, lda, R, N, K, dW, ldw, db}
 * as 64-bit host values) through the grouped split-K GEMM; wgrad_ws = sw_workspace_floats(SW_WS_WGRAD, ...) floats       */
int sw_wide_wgrad(const long long* desc, int n, float* wgrad_ws, void* stream);

/* ---- A two-hop gradient all-reduce over peer-mapped exchange buffers (csrc/sw_comm.hip) - the data-parallel step's
 *      alternative to `torch.distributed.all_reduce` on RCCL (3 flat buckets of 112 / 112 / 344 KB per step; the
 *      reference itself is single-process, train.py has no counterpart).  Every rank allocates ONE exchange buffer
 *      (sw_comm_alloc, sw_comm_bytes(world, max_floats) bytes, zero-filled device memory), exports it (sw_comm_ipc_export:
 *      a 64-byte hipIpc handle the host passes to the peers by any means), imports the peers' (sw_comm_ipc_import) and then
 *      calls sw_allreduce_direct with the `world` buffer addresses as mapped in ITS process (its own at [rank]): grad[0..n)
 *      becomes the element-wise sum over the ranks, every element summed in rank order by one rank (replicas receive
 *      identical bits).  Collective: every rank must issue the same sequence of calls (same n); asynchronous on `stream`,
 *      capturable in a hipGraph.  A peer that never arrives is given up after SW_COMM_TIMEOUT_S seconds (environment,
 *      default 30) instead of hanging the device: the rank whose wait timed out publishes nothing from it (its peers time
 *      out in turn), writes neither the gradient nor - in the _adam form - the weights for those elements, every later call
 *      on its buffer returns at once, and sw_comm_status (which synchronises the device) reports 1.  The caller makes the
 *      status collective before it trusts the results (trainer.train_epoch).  world <= 16.                                 */
long long sw_comm_bytes(int world, long long max_floats);
int sw_comm_alloc(long long bytes, void** ptr);
int sw_comm_free(void* ptr);
int sw_comm_ipc_export(void* ptr, void* handle64 /* 64 bytes out */);
int sw_comm_ipc_import(const void* handle64, void** ptr);
int sw_comm_ipc_close(void* ptr);
int sw_comm_status(const void* own_buf, int* status /* 0 ok, 1 a wait timed out */);
int sw_allreduce_direct(void* const* peer_bufs /* [world] */, int rank, int world, long long max_floats, float* grad,
                        long long n, void* stream);
/* ... and with the optimizer step behind it in the same launch: every rank applies Adam (sw_adam_packed's arithmetic, same
 * arguments; disc_Tp > 0: the packed weights are a Discriminator's and its registered images follow) to its replica from
 * the reduced gradient while it copies it out.  Interoperates with sw_allreduce_direct on other ranks (a rank without rows
 * in a batch exchanges zeros and updates separately). */
int sw_allreduce_direct_adam(void* const* peer_bufs, int rank, int world, long long max_floats, float* grad, long long n,
                             float* w, float* m, float* v, const float* step, double lr, double beta1, double beta2,
                             double eps, int disc_Tp, void* stream);

/* ---- ADE/FDE partial sums of train.py:546-551:
 *      out[3] = { sum_{b,t} err / Tp, sum_b err[:, -1], sum_{b,t} err^2 },  err = |(p_hat - p) / ss|   */
int sw_ade_fde(const float* pred4 /*[B,Tp,4]*/, const float* gt /*[B,Tp,2]*/, int B, int Tp, float inv_ss,
               float* out /*[3]*/, float* scratch /*[3*SW_RED_BLOCKS] or NULL*/, void* stream);

/* ==== MEASUREMENT SECTION - not part of the product surface (the drop-in boundary ends above this line; tests/test_abi.py
 *      checks the two lists separately).  sw_kernel_timing(1): every kernel launch of the library is bracketed by two HIP
 *      events on its own stream (never inside a graph capture) until sw_kernel_timing(0); sw_kernel_timing_read(buf, cap)
 *      waits for the device and writes one line "kernel calls total_us" per kernel, returning the bytes needed.
 *      sw_debug_spin queues a kernel that occupies the stream for ~us microseconds (queued in front of a timed sequence it
 *      lets the host run ahead, so the event intervals hold no launch gaps).  bench.py's roofline pass is their only caller. */
int sw_kernel_timing(int on);
int sw_kernel_timing_read(char* buf, int cap);
int sw_debug_spin(double us, void* stream);

#ifdef __cplusplus
}
#endif
#endif
