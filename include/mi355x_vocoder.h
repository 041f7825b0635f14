/* mi355x_vocoder.h - C ABI of libmi355x_vocoder.so (MI355X / gfx950 native HiFi-GAN vocoder path).
 *
 * Drop-in boundary: the reference has no FFI layer - its boundary is the
 * nn.Module surface of `hifigan_modified` - so each entry point below is what a binding for that
 * surface calls, and cites the reference call site (file:line under the reference root) whose
 * arithmetic it replaces.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *  - All tensor pointers are DEVICE pointers.  `dtype` (mv_dtype) is the storage type of
 *    activations AND weights of that call; accumulation is always fp32.  Small per-sample
 *    vectors that steer a kernel (ODConv attention alpha, GroupNorm mean/rstd, loss scalars) are fp32.
 *  - Activations at this boundary are "NCT" (batch, channel, time) contiguous unless a stride
 *    argument says otherwise, exactly like the reference's tensors.  Entry points whose name ends
 *    in `_cl` take the private channels-last ("NTC") layout used between fused kernels.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised.
 *  - Return value: MV_OK (0); MV_ERR_* (<0) for rejected arguments (nothing was launched);
 *    >0 = the hipError_t of a failed launch.  Nothing throws.
 *  - Re-entrant; no global state; the caller owns every buffer (workspaces are caller-provided).
 */
#ifndef MI355X_VOCODER_H
#define MI355X_VOCODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MV_F32 = 0, MV_BF16 = 1, MV_F16 = 2,
               /* operand mode of the mv_mrf_* entry points only: fp32 storage, every MFMA activation operand a hi + lo f16 pair, the
                * weights a single f16 (pack with this code as well: the image is the MV_F16 one) - two products per MAC instead of
                * MV_F32's three (hi + lo bf16 on both sides).  Not a storage type: every other entry point rejects it. */
               MV_F32_W16 = 3,
               /* mv_mrf_chain_fwd_cl / mv_mrf_chain_out_fwd_cl only: MV_F32_W16 whose INPUT x already is in the chain's internal row
                * format ("pair rows": per time step and 8-channel group 8 f16 hi values followed by 8 f16 lo values, hi + lo = the fp32
                * value to 22 bits; 256 bytes per 64-channel row, the size of the fp32 row).  mv_odconv_cl_fwd_pair writes that
                * format.  Streaming form only (dilations (1, 3, 5)), MV_ERR_UNSUPPORTED otherwise. */
               MV_F32_W16P = 4 } mv_dtype;
typedef enum { MV_ACT_NONE = 0, MV_ACT_LRELU = 1, MV_ACT_TANH = 2, MV_ACT_SILU = 3 } mv_act;

#define MV_OK 0
#define MV_ERR_ARG (-1)      /* shape / size / alignment violates the entry point's contract */
#define MV_ERR_DTYPE (-2)    /* unknown mv_dtype */
#define MV_ERR_UNSUPPORTED (-3)

/* Library identity: ABI version (bumped on any signature change) and the gfx target it was built for. */
int mv_abi_version(void);
const char* mv_build_target(void);

/* ------------------------------------------------------------------------------------------------
 * ODConv kernel attention.   replaces odconv.py:36-40,85 and :136-140,183
 *   alpha[b,:] = softmax_k( w[k,:] . mean_t x[b,:,t] + bias[k] )          x [B,C,T], w [K,C], bias [K]
 *   alpha: fp32 [B,K].  pooled (optional, may be NULL): fp32 [B,C] receives mean_t x (saved for backward). */
int mv_odconv_attn_fwd(const void* x, const void* w, const void* bias, float* alpha, float* pooled,
                       int B, int C, int T, int K, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (Dynamic) 1-D convolution, generic shapes.   replaces odconv.py:89-106 (nbanks=K, alpha!=NULL) and every
 * nn.Conv1d on the path (nbanks=1, alpha=NULL): grc_lora.py:33,57,66,160; discriminators.py:97-107; §A output_proj.
 *   y[b,o,t] = act( sum_k alpha[b,k] ( sum_{c,j} w[k,o,c,j] x[b,c,t*stride-pad+j*dil] + bias[k,o] ) ) + res[b,o,t]
 *   w [nbanks, Cout, Cin/groups, ks]; bias [nbanks, Cout] or NULL; res (optional) has y's layout.
 *   x_bs/y_bs: batch strides in elements, x_cs/y_cs: channel strides (time stride is 1) - lets a call read or
 *   write a channel slice of a wider buffer (the torch.cat of grc_lora.py:159 is written in place). */
int mv_conv1d_fwd(const void* x, const void* w, const void* bias, const float* alpha, const void* res, void* y,
                  int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil, int groups,
                  int nbanks, int act, float slope,
                  long x_bs, long x_cs, long y_bs, long y_cs, int dtype, void* stream);

/* (Dynamic) transposed 1-D convolution.   replaces odconv.py:187-204 (+ the nn.LeakyReLU(0.1) of SURVEY §A item 2 when act=LRELU)
 *   y[b,o,u] = act( sum_k alpha[b,k] ( sum_{c,j: u = t*stride - pad + j*dil} w[k,c,o,j] x[b,c,t] + bias[k,o] ) )
 *   w [nbanks, Cin, Cout, ks]; Tout = (Tin-1)*stride - 2*pad + dil*(ks-1) + out_pad + 1; groups must be 1. */
int mv_conv_transpose1d_fwd(const void* x, const void* w, const void* bias, const float* alpha, void* y,
                            int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil,
                            int nbanks, int act, float slope, int dtype, void* stream);

/* 2-D convolution, stride 1.   replaces discriminators.py:57-65 (Conv2d 3x3 pad 1 + LeakyReLU 0.1)
 *   x [B,Cin,H,W] -> y [B,Cout,H,W'] ; w [Cout,Cin,kh,kw]; H' = H+2ph-kh+1, W' = W+2pw-kw+1. */
int mv_conv2d_fwd(const void* x, const void* w, const void* bias, void* y,
                  int B, int Cin, int H, int W, int Cout, int kh, int kw, int ph, int pw,
                  int act, float slope, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm, two-phase.   replaces grc_lora.py:58,161 and generator.py:170 (nn.GroupNorm, eps 1e-5)
 *   stats: mean/rstd fp32 [B,G] over (C/G channels x T) of x (biased variance).
 *   apply: y = act( (x-mean)*rstd*gw[c] + gb[c] ) * (mask ? mask[b,c,t]*mask_scale : 1) + res
 *   strides as in mv_conv1d_fwd (x and res may be channel slices). mask: uint8 keep-mask of nn.Dropout (grc_lora.py:162). */
int mv_groupnorm_stats(const void* x, float* mean, float* rstd, int B, int C, int T, int G, float eps,
                       long x_bs, long x_cs, int dtype, void* stream);
int mv_groupnorm_apply(const void* x, const float* mean, const float* rstd, const void* gw, const void* gb,
                       const void* res, const uint8_t* mask, float mask_scale, void* y,
                       int B, int C, int T, int G, int act, float slope,
                       long x_bs, long x_cs, long r_bs, long r_cs, long y_bs, long y_cs, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GRC+LoRA weight folding.   replaces the parameter algebra of grc_lora.py:33-57:
 *   conv_g(x) + s*(x^T A B)^T followed by the 1x1 output_projection is ONE dense dilated conv with
 *   w_eff[o,c,j] = sum_o' Wp[o,o'] ( [c in group(o')] Wc[o',c_local,j] + [j==ks/2] s * (A B)[c,o'] ),
 *   b_eff[o] = sum_o' Wp[o,o'] bc[o'] + bp[o].   ks must be odd.  All parameter tensors in `dtype`; outputs in `dtype`. */
int mv_grc_fold_weights(const void* conv_w, const void* conv_b, const void* lora_A, const void* lora_B,
                        const void* lora_scaling, const void* proj_w, const void* proj_b,
                        void* w_eff, void* b_eff, int Cin, int Cout, int ks, int groups, int rank,
                        int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small dense layer  y[m,n] = sum_k x[m,k] w[n,k] + b[n]   (nn.Linear of grc_lora.py:108, generator.py:193-194). */
int mv_linear_fwd(const void* x, const void* w, const void* b, void* y, int M, int N, int Kd, int dtype, void* stream);

/* FiLM modulation.   replaces grc_lora.py:111-129: proj [B,2F] = Linear(cond); gamma = proj[:, :F], beta = proj[:, F:];
 *   channels c >= F pass through (gamma 1, beta 0); if C < F the first C entries are used.  y = x*gamma + beta, x [B,C,T]. */
int mv_film_fwd(const void* x, const void* proj, void* y, int B, int C, int T, int F, int dtype, void* stream);
/* Second-design FiLM (generator.py:193-197): y = scale[b,c]*x + shift[b,c]; scale/shift [B,C] in `dtype`. */
int mv_scale_shift_fwd(const void* x, const void* scale, const void* shift, void* y, int B, int C, int T,
                       int dtype, void* stream);

/* Elementwise y = act(x) (+ res).  n elements. */
int mv_act_fwd(const void* x, const void* res, void* y, long n, int act, float slope, int dtype, void* stream);

/* AvgPool1d(kernel=s, stride=s), floor.   replaces discriminators.py:94,112.  x [B,C,T] -> y [B,C,T/s] */
int mv_avgpool1d_fwd(const void* x, void* y, long rows, int T, int s, int dtype, void* stream);

/* MPD fold.   replaces discriminators.py:72-79 (F.pad zeros right + view(B,C,P,T'/P)).
 *   Writes the padded signal y [rows, Tp] (Tp = ceil(T/P)*P) which, viewed as [rows, P, Tp/P], IS the folded
 *   tensor (row p = contiguous chunk p).  index (optional, int64 [Tp]) receives the source index of every
 *   folded element, -1 for padding - the bit-exact index map the parity tests compare. */
int mv_mpd_fold(const void* x, void* y, int64_t* index, long rows, int T, int P, int dtype, void* stream);

/* Layout transposes between the public NCT layout and the private channels-last NTC layout. */
int mv_nct_to_ntc(const void* x, void* y, int B, int C, int T, int dtype, void* stream);
int mv_ntc_to_nct(const void* x, void* y, int B, int C, int T, int dtype, void* stream);
/* Same transposes with the channel axis zero-padded to / cropped from Cpad on the channels-last side
 * (x [B][C][T] -> y [B][T][Cpad] with zeros in channels C..Cpad-1; x [B][T][Cpad] -> y [B][C][T]): the GRC branch widths
 * (20 and 60 channels, grc_lora.py:86-93) are padded to the 32-channel MFMA granule this way. */
int mv_nct_to_ntc_pad(const void* x, void* y, int B, int C, int T, int Cpad, int dtype, void* stream);
int mv_ntc_to_nct_crop(const void* x, void* y, int B, int C, int T, int Cpad, int dtype, void* stream);
/* x [B][C][T] -> y [B][T][C] where consecutive batches of y are y_batch_stride elements apart: the transposed data lands in a
 * window of a larger pre-zeroed channels-last buffer (time padding for the ODConvTranspose1d adjoint). */
int mv_nct_to_ntc_window(const void* x, void* y, int B, int C, int T, long y_batch_stride, int dtype, void* stream);
/* Data gradient of ODConvTranspose1d (odconv.py:172-205) with kernel_size = 2*stride as a stride-1 two-tap ODConv over rows
 * of `stride` output steps: out [K][Cin][stride*Cout][2], out[k][c][r*Cout+o][q] = kernels[k][c][o][q*stride + r]; feed it
 * to mv_odconv_cl_pack(transposed = 0) and run mv_odconv_cl_fwd on the (time-padded) output gradient. */
int mv_odconvT_adjoint_weights(const void* kernels, void* out, int K, int Cin, int Cout, int ks, int stride, int dtype,
                               void* stream);
/* Bank gradients of ODConvTranspose1d (odconv.py:172-205, kernel_size = 2*stride) on MFMA: x_cl [B][Tin][Cin] (the layer
 * input, channels-last), gp = the time-padded output gradient [B][Tin+1][stride*Cout] (row q, channel r*Cout+o = g[q*stride+r-pad][o]),
 * w = kernels [K][Cin][Cout][ks] in `dtype`, alpha fp32 [B][K].  gw fp32 [K][Cin][Cout][ks] = sum_b alpha[b,k] * (per-sample
 * gradient); galpha fp32 [B][K] += <per-sample gradient, W[k]>.  workspace: mv_odconvT_wgrad_workspace_bytes.
 * mv_odconv_wgrad_reduce is the alpha-chain reduction alone (per-sample tiles and banks in any common layout). */
size_t mv_odconvT_wgrad_workspace_bytes(int B, int Cin, int Cout, int ks, int K, int dtype);
int mv_odconvT_wgrad_mfma(const void* x_cl, const void* gp, const void* w, const float* alpha, float* gw, float* galpha,
                          void* workspace, int B, int Tin, int Cin, int Cout, int ks, int stride, int K, int dtype, void* stream);
int mv_odconv_wgrad_reduce(const float* gws, const void* w, const float* alpha, float* gw, float* galpha, int B, int K,
                           long nelem, int dtype, void* stream);

/* dtype conversion (fp32 <-> bf16/fp16) of n elements. */
int mv_cast(const void* x, int src_dtype, void* y, int dst_dtype, long n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused MultiReceptiveFieldBlock, channels-last.   replaces grc_lora.py:32-68 (x3 branches) + :157-163 for the
 * generator's instance: 64 -> 64 channels, three GRC_LoRA_Block(64, 20, 3, d_i, r) branches, Conv1d(60,64,1),
 * GroupNorm(8,64), Dropout, + x.   x, out: [B][T][64] ("NTC").
 *
 * mv_mrf_params: raw parameter pointers of one block, all in `param_dtype`, index i = branch (conv_layers.i.*):
 *   conv_w [20,16,3] conv_b [20] lora_A [64,r] lora_B [r,20] lora_scaling [1] proj_w [20,20(,1)] proj_b [20]
 *   norm_w/norm_b [20] res_w [20,64(,1)] res_b [20]; fusion_w [64,60(,1)] fusion_b [64] norm2_w/norm2_b [64].
 * mv_mrf_pack folds conv_g + LoRA + output_projection per branch (fp32) and writes MFMA-fragment-ordered weights
 * for storage type `dtype` into `packed` (mv_mrf_packed_bytes(dtype) bytes, 16-byte aligned).  Re-run after every
 * parameter update.  mv_mrf_block_fwd_cl runs the three passes (GN5 statistics, GN8 statistics, output);
 * workspace: mv_mrf_workspace_bytes(B,T,dtype) bytes, fully rewritten by every call (no zero-fill needed).
 * dropout_mask: optional uint8 keep-mask [B][T][64] (training), output scaled by mask_scale = 1/(1-p).
 * dilations: host array of 3 ints (1..16, at most 7 distinct tap offsets).  x and out must not alias. */
typedef struct mv_mrf_params {
  const void* conv_w[3]; const void* conv_b[3]; const void* lora_A[3]; const void* lora_B[3];
  const void* lora_scaling[3]; const void* proj_w[3]; const void* proj_b[3];
  const void* norm_w[3]; const void* norm_b[3]; const void* res_w[3]; const void* res_b[3];
  const void* fusion_w; const void* fusion_b; const void* norm2_w; const void* norm2_b;
} mv_mrf_params;
size_t mv_mrf_packed_bytes(int dtype);
size_t mv_mrf_workspace_bytes(int B, int T, int dtype);
int mv_mrf_pack(const mv_mrf_params* params, int param_dtype, const int* dilations, int lora_rank,
                void* packed, int dtype, void* stream);
int mv_mrf_block_fwd_cl(const void* x, void* out, const void* packed, const int* dilations, void* workspace,
                        const uint8_t* dropout_mask, float mask_scale, int B, int T, float eps, int dtype,
                        void* stream);
/* nblocks consecutive MultiReceptiveFieldBlocks in one call (the generator's `for blk in mrf_blocks: x = blk(x)`, SURVEY.md appendix A
 * / grc_lora.py:157-163, eval mode).  Same arithmetic as nblocks calls of mv_mrf_block_fwd_cl, restructured: block i writes its
 * pre-GroupNorm fusion output once, and GroupNorm(8,64) + the residual add of block i are applied by block i+1's first pass while it
 * loads its input (2 launches and 320 MFMAs per 64-step tile and block instead of 3 and 512).  packed[i] = mv_mrf_pack of block i,
 * dilations = nblocks x 3 ints, workspace >= mv_mrf_chain_workspace_bytes bytes (256-byte aligned, fully rewritten per call).
 * dtype MV_F32_W16 (fp32 x / out, packed[i] = the MV_F16 image): two-product operands; with the reference's dilations (1, 3, 5) the
 * passes run in their streaming form (csrc/mrf_stream.hip).  out may be NULL: the passes run and nothing is materialised (timing). */
size_t mv_mrf_chain_workspace_bytes(int B, int T, int dtype);
int mv_mrf_chain_fwd_cl(const void* x, void* out, const void* const* packed, const int* dilations, int nblocks, void* workspace,
                        int B, int T, float eps, int dtype, void* stream);
/* The chain followed by the generator's output projection + activation (output_proj Conv1d(64,1,ks,pad ks/2),
 * torch.tanh): wave [B][1][T] in `dtype`.  The last block's GroupNorm + residual is applied while the output conv stages its input,
 * so that block's output is never stored.  conv_packed = mv_conv_out_pack_all(output_proj.weight); act as in mv_conv_out_act_cl. */
int mv_mrf_chain_out_fwd_cl(const void* x, void* wave, const void* const* packed, const int* dilations, int nblocks, void* workspace,
                            const void* conv_packed, float conv_bias, int ks, int act, int B, int T, float eps, int dtype,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused ODConv1d / ODConvTranspose1d, channels-last.   replaces odconv.py:73-108 / :172-205 (attention ->
 * sum_k alpha_k W_k -> conv in ONE launch) plus the epilogues that follow it in the generator: FiLM
 * (grc_lora.py:111-129) after input_proj, LeakyReLU(0.1) after every upsampler.
 *   x [B][Tin][Cin], y [B][Tout][Cout] ("NTC").  `packed` = mv_odconv_cl_pack(kernels) (MFMA A-fragment order,
 *   mv_odconv_cl_packed_bytes bytes; kernels are [K,Cout,Cin,ks] (conv) or [K,Cin,Cout,ks] (transposed) in
 *   param_dtype).  bias [K][Cout] in `dtype` or NULL.
 *   alpha: fp32 [B][K] if already known; otherwise pooled_in = the producing launch's pooled_out (fp32 [B][pooled_in_count]:
 *   per sample `slots` partial sums of every GEMM row, pooled_in_count = mv_odconv_cl_pool_floats(producer geometry); a dense
 *   [B][Cin] buffer of channel sums is the special case pooled_in_count = Cin) with att_w [K][Cin], att_b [K] in `dtype`: alpha is
 *   formed in the prologue by summing the partials in a fixed order (no atomics: two runs agree bit for bit).
 *   film_proj (optional): [B][2*film_F] in `dtype`; channels >= film_F pass through.
 *   pooled_out (optional): fp32 [B][mv_odconv_cl_pool_floats(...)], 16-byte aligned, need not be initialised; every element is
 *   written exactly once: SUM over the columns of one workgroup of the stored outputs, per GEMM row (row % Cout = channel).
 *   Supported: Cin % 8 == 0, Cout % 8 == 0, GEMM rows (Cout, or stride*Cout) % 16 == 0; conv: stride 1; transposed:
 *   dilation 1 and ks % stride == 0.  Anything else returns MV_ERR_UNSUPPORTED (use the generic entry points). */
size_t mv_odconv_cl_packed_bytes(int Cin, int Cout, int ks, int stride, int transposed, int K, int dtype);
int mv_odconv_cl_pack(const void* kernels, int param_dtype, void* packed, int Cin, int Cout, int ks, int stride,
                      int transposed, int K, int dtype, void* stream);
int mv_odconv_cl_fwd(const void* x, const void* packed, const void* bias, const float* alpha, const float* pooled_in,
                     int pooled_in_count, const void* att_w, const void* att_b, const void* film_proj, int film_F, void* y,
                     float* pooled_out, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad,
                     int dil, int transposed, int K, int act, float slope, int dtype, void* stream);
/* mv_odconv_cl_fwd in fp32 storage (packed weights, bias, attention head and y are fp32) whose INPUT x is fp16: the first fp32
 * stage of the mixed storage mode reads the fp16 stream directly (widened and split while its tiles are staged) instead of a cast
 * launch.  Only the geometries of the multi-tile kernel (ODConvTranspose1d, ks = 2 * stride, 64 / 128 input channels, small banks);
 * MV_ERR_UNSUPPORTED otherwise - cast and call mv_odconv_cl_fwd. */
int mv_odconv_cl_fwd_in16(const void* x_f16, const void* packed, const void* bias, const float* alpha, const float* pooled_in,
                          int pooled_in_count, const void* att_w, const void* att_b, void* y, float* pooled_out, int B, int Cin,
                          int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil, int transposed, int K, int act,
                          float slope, void* stream);
/* mv_odconv_cl_fwd (dtype MV_F32, no FiLM) whose output rows are written in the MRF chain's pair-row format (see MV_F32_W16P): the
 * last upsampler hands its result to mv_mrf_chain_out_fwd_cl without an fp32 round trip through registers - the chain's first pass
 * reads pair rows by LDS-DMA.  Only the 64 -> 64 channel streaming kernel (ODConvTranspose1d, ks = 2 * stride); MV_ERR_UNSUPPORTED
 * otherwise - call mv_odconv_cl_fwd and pass MV_F32_W16 to the chain.  pooled_out as in mv_odconv_cl_fwd. */
int mv_odconv_cl_fwd_pair(const void* x, const void* packed, const void* bias, const float* alpha, const float* pooled_in,
                          int pooled_in_count, const void* att_w, const void* att_b, void* y_pair, float* pooled_out, int B, int Cin,
                          int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil, int transposed, int K, int act,
                          float slope, void* stream);
/* floats PER SAMPLE that the launch above writes to pooled_out (0 = unsupported geometry): slots x GEMM rows, where the slot
 * count follows the kernel variant the dispatcher picks for this geometry. */
size_t mv_odconv_cl_pool_floats(int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil,
                                int transposed, int K, int act, int has_film, int dtype);
/* the same for a launch through mv_odconv_cl_fwd_in16 (in_f16 != 0, dtype MV_F32): the input-widening variants set their own grids */
size_t mv_odconv_cl_pool_floats_in(int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil,
                                   int transposed, int K, int act, int has_film, int dtype, int in_f16);
/* Generator prologue in one launch: input_proj's attention alpha fp32 [B][K] (odconv.py:36-40) and mel [B][C][T] -> x_cl [B][T][C]
 * (one workgroup per sample), the FiLM projection film_proj [B][F2] = W cond + b with cond = cat(spk [B][ds], emo [B][de])
 * truncated / zero-padded to cond_dim (grc_lora.py:82-105; film_proj NULL = no conditioning; extra workgroups of the same grid,
 * one per (16 rows, 8 samples)), and zeroing of zero_buf[0..zero_n).  MV_ERR_UNSUPPORTED when one sample does not fit LDS. */
int mv_gen_prologue(const void* mel, const void* att_w, const void* att_b, const void* spk, const void* emo, const void* film_w,
                    const void* film_b, float* alpha, void* x_cl, void* film_proj, float* zero_buf, long zero_n, int B, int C,
                    int T, int K, int ds, int de, int cond_dim, int F2, int dtype, void* stream);
/* the same with the INPUTS (mel, spk, emo) in in_dtype = dtype or fp32: a 16-bit-storage front of an fp32 model reads the caller's
 * fp32 tensors directly (the mixed storage mode) instead of three cast launches. */
int mv_gen_prologue_in(const void* mel, const void* att_w, const void* att_b, const void* spk, const void* emo, const void* film_w,
                       const void* film_b, float* alpha, void* x_cl, void* film_proj, float* zero_buf, long zero_n, int B, int C,
                       int T, int K, int ds, int de, int cond_dim, int F2, int in_dtype, int dtype, void* stream);

/* Output projection, channels-last in, waveform out.   replaces SURVEY.md §A item 4: nn.Conv1d(C,1,ks,padding=ks/2) + torch.tanh
 *   x [B][T][C] -> y [B][1][T].  wt = mv_conv_out_pack(weight [1,C,ks]) (fp32 [ks][C]).  C must be 64, ks odd. */
int mv_conv_out_pack(const void* w, int param_dtype, float* wt, int C, int ks, void* stream);
int mv_conv_out_act_cl(const void* x, const float* wt, float bias, void* y, int B, int T, int C, int ks, int pad,
                       int act, int dtype, void* stream);
/* The same output convolution with all three weight images packed once (fp32 | bf16 | f16, mv_conv_out_packed_bytes): 16-bit storage
 * runs on the packed dot-product instructions (v_dot2c_f32_bf16 / _f16: two MACs per lane and instruction, fp32 accumulate, weights
 * in the activation type), fp32 storage on the fp32 image as mv_conv_out_act_cl does. */
size_t mv_conv_out_packed_bytes(int C, int ks);
int mv_conv_out_pack_all(const void* w, int param_dtype, void* packed, int C, int ks, void* stream);
int mv_conv_out_act_packed_cl(const void* x, const void* packed, float bias, void* y, int B, int T, int C, int ks, int pad, int act,
                              int dtype, void* stream);

/* ================================================================================================
 * Backward / training entry points.  Parameter gradients are always fp32.  Data gradients of the convolutions
 * re-use the forward entry points: dgrad(conv1d) = mv_conv_transpose1d_fwd(gy, same weights), dgrad(conv_transpose1d)
 * = mv_conv1d_fwd(gy, same weights), dgrad(conv2d) = mv_conv2d_fwd(gy, mv_conv2d_flip_weights(w)).
 * (autograd of F.conv1d / F.conv_transpose1d / F.conv2d at odconv.py:95-99,192-197, grc_lora.py:33,57,66,160,
 *  discriminators.py:57-65,97-107) */

/* gx = gy * act'(z), expressed through the activation OUTPUT y (LeakyReLU: sign(y); tanh: 1-y^2). */
int mv_act_bwd(const void* gy, const void* y, void* gx, long n, int act, float slope, int dtype, void* stream);

/* conv1d weight gradient (groups = 1, ks <= 16).  y = conv1d(x, w[nbanks][Cout][Cin][ks], alpha).
 *   nbanks == 1: gw [Cout][Cin][ks] = sum_{b,t} gy x ;  nbanks > 1 (ODConv): gw[k] = sum_b alpha[b,k] gW~_b and
 *   galpha[b,k] += <gW~_b, w[k]> (galpha must be zero- or bias-term-initialised by the caller).  ODConv runs in two stages:
 *   per-sample tiles into `workspace` (mv_conv1d_wgrad_workspace_bytes; no cross-sample atomics), then one reduction kernel.
 *   The transposed-conv weight gradient is the same call with the roles swapped (x := gy, gy := x, Cin := Cout, ...):
 *   see functional.py.  Strides as in mv_conv1d_fwd. */
size_t mv_conv1d_wgrad_workspace_bytes(int B, int Cin, int Cout, int ks, int nbanks);
int mv_conv1d_wgrad(const void* x, const void* gy, const void* w, const float* alpha, float* gw, float* galpha,
                    float* workspace, int B, int Cin, int Tin, int Cout, int Tout, int ks, int stride, int pad, int dil, int nbanks,
                    long x_bs, long x_cs, long g_bs, long g_cs, int dtype, void* stream);
/* gbias[k][o] = sum_b alpha[b,k] sum_t gy[b,o,t]; galpha[b,k] += sum_o (sum_t gy[b,o,t]) bias[k][o] (K > 1).
 * rowsum_ws: fp32 [B*C] scratch. */
int mv_bias_grad(const void* gy, const float* alpha, const void* bias, float* rowsum_ws, float* gbias, float* galpha,
                 int B, int C, int T, int K, long g_bs, long g_cs, int dtype, void* stream);
/* ODConv attention backward: from galpha to gWa [K][C], gba [K] and gm [B][C] = (1/T) Wa^T gz, the
 * per-(b,c) constant that mv_add_rowconst adds to every time step of gx. */
int mv_odconv_attn_bwd(const float* alpha, const float* galpha, const float* pooled, const void* wa, float* gwa,
                       float* gba, float* gm, int B, int C, int T, int K, int dtype, void* stream);
int mv_add_rowconst(void* x, const float* v, long rows, int T, int dtype, void* stream);

/* GroupNorm(+activation, +dropout mask) backward.  y = act((x-mean)*rstd*gw+gb) * mask*mask_scale (+ res, whose gradient is gy).
 *   gz_ws: B*C*T elements of `dtype`; ws: fp32 [2*B*G + 2*B*C]; dgw/dgb: fp32 [C] (may be NULL). */
int mv_groupnorm_bwd(const void* x, const void* gy, const float* mean, const float* rstd, const void* gw, const void* gb,
                     const uint8_t* mask, float mask_scale, int act, float slope, void* gz_ws, float* ws, void* gx,
                     float* dgw, float* dgb, int B, int C, int T, int G, long x_bs, long x_cs, long g_bs, long g_cs,
                     int dtype, void* stream);

/* FiLM backward: gx = gy*gamma; gproj [B][2F] fp32 = (sum_t gy*x | sum_t gy).  Linear backward (fp32 gy [M][N]). */
int mv_film_bwd(const void* x, const void* gy, const void* proj, void* gx, float* gproj, int B, int C, int T, int F,
                int dtype, void* stream);
int mv_linear_bwd(const void* x, const void* w, const float* gy, float* gx, float* gw, float* gb, int M, int N, int Kd,
                  int dtype, void* stream);

int mv_avgpool1d_bwd(const void* gy, void* gx, long rows, int T, int s, int dtype, void* stream);
/* strided 2-D copy dst[outer][inner][0..n) = src[outer][inner][0..n) (element strides) - channel concat / slice. */
int mv_copy2d(const void* src, void* dst, int n, int rows_outer, int rows_inner, long s_os, long s_rs, long d_os,
              long d_rs, int dtype, void* stream);
/* wt[c][o][kh-1-i][kw-1-j] = w[o][c][i][j] (conv2d data gradient = conv2d with these weights, same padding for odd kernels) */
int mv_conv2d_flip_weights(const void* w, void* wt, int Cout, int Cin, int kh, int kw, int dtype, void* stream);
int mv_conv2d_wgrad(const void* x, const void* gy, float* gw, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                    int ph, int pw, int dtype, void* stream);

/* Backward of mv_grc_fold_weights: (g_weff, g_beff) fp32 -> fp32 gradients of the seven GRC parameter tensors. */
int mv_grc_fold_bwd(const float* g_weff, const float* g_beff, const void* conv_w, const void* conv_b,
                    const void* lora_A, const void* lora_B, const void* lora_scaling, const void* proj_w,
                    float* g_conv_w, float* g_conv_b, float* g_A, float* g_B, float* g_s, float* g_proj_w,
                    float* g_proj_b, int Cin, int Cout, int ks, int groups, int rank, int param_dtype, void* stream);

/* Losses, value and gradient in one pass.   replaces complete_vocoder.py:103-127,156-176 and conditioned_hifigan.py:234-265
 *   kind 0: mean((x-c)^2)  1: mean|x-y|  2: mean(relu(1-x))  3: mean(relu(1+x))  4: mean((x-y)^2)
 *   loss_acc[0] += weight*value (fp32, atomic); gx (optional) = weight * d value/dx; gy (optional, kinds 1/4) = -gx. */
int mv_loss_fwd_bwd(const void* x, const void* y, float c, float weight, float* loss_acc, void* gx, void* gy, long n,
                    int kind, int dtype, void* stream);
/* x *= factor * (factor_dev ? factor_dev[0] : 1) */
int mv_scale(void* x, const float* factor_dev, float factor, long n, int dtype, void* stream);
/* y = x * factor * (factor_dev ? factor_dev[0] : 1), out of place (loss backward: the saved gradient times the upstream scalar). */
int mv_scale_to(const void* x, void* y, const float* factor_dev, float factor, long n, int dtype, void* stream);

/* Log-mel spectrogram (+ L1 loss against `target`, fp32 [B][n_mels][T/hop]) of wave [B][1][T]; defined by this build (the
 * reference has only placeholders: complete_vocoder.py:210-212, conditioned_hifigan.py:269-274).  Frames: reflect pad
 * (n_fft-hop)/2, periodic Hann, |rFFT| (in-LDS radix-2 FFT when n_fft is a power of two, direct DFT otherwise), mel = fb [n_mels][n_fft/2+1] (fp32) @ mag, log(max(., clamp)).
 *   mel_out (optional fp32 [B][n_mels][T/hop]); loss_acc += weight * mean|logmel - target| (kind 0) or weight * mean (logmel - target)^2 (kind 1);
 *   backward != 0: gwave (fp32 [B][T], zero on entry) += d loss / d wave. */
int mv_mel_loss(const void* wave, const float* fb, const float* target, float* mel_out, float* loss_acc, float* gwave,
                int B, int T, int n_fft, int hop, int n_mels, float clampv, float weight, int kind, int backward,
                int dtype, void* stream);

/* AdamW on a flat fp32 parameter arena (torch.optim.AdamW semantics; conditioned_hifigan.py:219): grads are first
 * gathered into a flat fp32 buffer by mv_multi_gather (descs_dev: device array of {const void* src; long dst_off; long n;
 * int dtype; int pad}; src == NULL -> zeros), which is also the buffer RCCL all-reduces. grad_scale multiplies g (1/world). */
int mv_multi_gather(const void* descs_dev, int n_tensors, long max_len, float* flat, void* stream);
int mv_adamw_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, float grad_scale, void* stream);
/* the same with the step count (>= 1, already incremented) read from device memory: the form a captured HIP graph replays */
int mv_adamw_flat_dev(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, const int* step_dev, float grad_scale, void* stream);
/* uint8 keep-mask of nn.Dropout(p) (grc_lora.py:151,162): mask[i] = 1 with probability 1-p, Philox4x32-10 keyed by `seed`,
 * counter = element index / 8 (16 random bits per element; resolution 2^-16 on p). */
int mv_dropout_mask(uint8_t* mask, long n, float p, long seed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Channels-last MFMA discriminator convolutions.   replaces discriminators.py:57-65 (Conv2d 3x3 pad 1) and :98-106
 * (Conv1d k15 pad 7), with their LeakyReLU(0.1), forward AND backward.   x [B][H][W][Cin], y [B][H][W][Cout], stride 1,
 * odd kernels with "same" padding, Cin % 32 == 0, Cout % 16 == 0 (MSD: H = 1; MPD: H = period).
 *   mv_dconv_pack(w [Cout,Cin,kh,kw], flip): MFMA A-fragment order; flip = 1 packs the data-gradient kernel
 *     (taps flipped, channels transposed) - call mv_dconv_cl_fwd with Cin/Cout swapped on the output gradient.
 *   mv_dconv_cl_fwd: y = act(conv(x) + bias).  act_save (optional, [B][H][W][Cout]): data-gradient mode - the result is
 *     multiplied by LeakyReLU'(.) evaluated on that saved activation, i.e. it is d/d(pre-activation) of the previous layer.
 *   mv_dhead_*: the Cout = 1 layer (weights as fp32 [kh*kw][C] from mv_conv_out_pack-style transposition).
 *   mv_dconv_wgrad_cl: gw fp32 [Cout][Cin][kh][kw] = sum g x (16-bit storage only; 3x3, and 1xk for k in 1,3,5,7,11,15
 *     with W-dilation dil_w, (k-1)*dil_w <= 64 - the GRC/MRF convs of grc_lora.py:36-41 run through the same kernels);
 *     gb (optional, fp32 [Cout]): the bias gradient sum_pos g[pos][o], accumulated from the LDS-resident g tiles;
 *     workspace: mv_dconv_wgrad_workspace_bytes() of device scratch (tap-major partial sums). */
size_t mv_dconv_packed_bytes(int Cout, int Cin, int kh, int kw, int dtype);
int mv_dconv_pack(const void* w, int param_dtype, void* packed, int Cout, int Cin, int kh, int kw, int flip, int dtype,
                  void* stream);
/* n weight tensors packed by one launch.  descs_dev: device array of { const float* src; void* dst; int Cout, Cin, kh, kw, flip, dtype; }
 * (40 bytes each): src = fp32 [Cout][Cin][kh][kw] with Cout % 16 == 0 and Cin % 32 == 0 (no padding), dst = mv_dconv_packed_bytes image,
 * dtype MV_BF16 / MV_F16.  Same result as n mv_dconv_pack calls. */
int mv_dconv_multi_pack(const void* descs_dev, int n, void* stream);
/* mv_dconv_pack with the operator zero-padded to Coutp x Cinp channels (multiples of 32 for the side that is contracted). */
int mv_dconv_pack_pad(const void* w, int param_dtype, void* packed, int Cout, int Cin, int kh, int kw, int Coutp, int Cinp,
                      int flip, int dtype, void* stream);
int mv_dconv_cl_fwd(const void* x, const void* packed, const void* bias, const void* act_save, void* y, int B, int H,
                    int W, int Cin, int Cout, int kh, int kw, int dil_w, int act, float slope, int dtype, void* stream);
/* Head (C = 256 -> 1 channel, discriminators.py:65 / :106) on MFMA with the taps on the matrix rows; 16-bit storage.
 *   mv_dhead_pack: w [1][C][kh][kw] -> forward + data-gradient operators (mv_dhead_packed_bytes).
 *   mv_dhead_fwd:  y [B][H][W] = bias[0] + conv(x [B][H][W][C]); workspace: mv_dhead_workspace_bytes (per-tap partial sums).
 *   mv_dhead_dgrad: gx [B][H][W][C] = LeakyReLU'(xsave) * conv^T(g [B][H][W]). */
/*   mv_dconv_cl_fwd_head: mv_dconv_cl_fwd (dilation 1, no act_save) of the LAST hidden layer (Cout = 256) that also leaves the head's
 *                  per-tap partial sums in head_ws, computed from the output tile while it is still in LDS - the head forward without
 *                  reading the 256-channel activation back from HBM (discriminators.py:65 / :106 applied to :57-64 / :97-105).
 *                  head_packed: mv_dhead_pack's buffer.  MV_ERR_UNSUPPORTED when the geometry has no 256-row workgroup variant.
 *   mv_dhead_sum:  y [B][H][W] = bias[0] + shifted sum of the per-tap partial sums (the second half of mv_dhead_fwd). */
int mv_dconv_cl_fwd_head(const void* x, const void* packed, const void* bias, void* y, const void* head_packed, float* head_ws,
                         int head_kh, int head_kw, int B, int H, int W, int Cin, int Cout, int kh, int kw, int act, float slope,
                         int dtype, void* stream);
int mv_dhead_sum(const float* workspace, const void* bias, void* y, int B, int H, int W, int kh, int kw, int dtype, void* stream);
size_t mv_dhead_packed_bytes(int C, int dtype);
size_t mv_dhead_workspace_bytes(int B, int H, int W);
int mv_dhead_pack(const void* w, int param_dtype, void* packed, int C, int kh, int kw, int dtype, void* stream);
int mv_dhead_fwd(const void* x, const void* packed, const void* bias, float* workspace, void* y, int B, int H, int W, int C,
                 int kh, int kw, int dtype, void* stream);
int mv_dhead_dgrad(const void* g, const void* packed, const void* xsave, void* gx, int B, int H, int W, int C, int kh,
                   int kw, float slope, int dtype, void* stream);
int mv_dhead_wgrad(const void* g, const void* x, float* gw, float* gb, int B, int H, int W, int C, int kh, int kw,
                   int dtype, void* stream);
size_t mv_dconv_wgrad_workspace_bytes(int Cin, int Cout, int kh, int kw);
int mv_dconv_wgrad_cl(const void* x, const void* g, float* gw, float* gb, float* workspace, int B, int H, int W, int Cin,
                      int Cout, int kh, int kw, int dil_w, int dtype, void* stream);
/* The same with a workspace the caller keeps between calls: (Cout*Cin*kh*kw + Cout) floats, ZERO on entry, zero again on exit (the
 * reorder pass clears what it reads), so no fill launch is needed per call.  gb (may be NULL) receives the bias sums. */
int mv_dconv_wgrad_cl_pz(const void* x, const void* g, float* gw, float* gb, float* workspace, int B, int H, int W, int Cin,
                         int Cout, int kh, int kw, int dil_w, int dtype, void* stream);

/* First discriminator layer (1 -> C1 channels, LeakyReLU), channels-last output, and its gradients
 * (discriminators.py:57 / :98 with Cin = 1): x0 [B][H][W], w [C1][kh*kw] (= the [C1,1,kh,kw] parameter), a1/g1 [B][H][W][C1]. */
int mv_dfirst_fwd_cl(const void* x0, const void* w, const void* bias, void* y, int B, int H, int W, int C1, int kh,
                     int kw, float slope, int dtype, void* stream);
size_t mv_dfirst_dgrad_workspace_bytes(int B, int H, int W);
int mv_dfirst_dgrad_cl(const void* g1, const void* w, void* gx0, float* workspace, int B, int H, int W, int C1, int kh, int kw,
                       int dtype, void* stream);
int mv_dfirst_wgrad_cl(const void* g1, const void* x0, float* gw, float* gb, int B, int H, int W, int C1, int kh, int kw,
                       int dtype, void* stream);
/* out [B][H][W][16] = the 16-tap "im2col" of a one-channel map sc [B][H][W]: out[pos][tap] = sc[pos + off(tap)] (flip 0) or
 * sc[pos - off(tap)] (flip 1), zero outside the image and for tap >= kh*kw.  Turns the one-channel weight gradients of the
 * first layer and the head (discriminators.py:57,65 / :98,106) into 1x1 weight-gradient GEMMs for mv_dconv_wgrad_cl. */
int mv_tap_matrix(const void* sc, void* out, int B, int H, int W, int kh, int kw, int flip, int dtype, void* stream);
/* out[c] = sum over rows of x[row][c] (bias gradient of a channels-last tensor); C divides 256. */
int mv_colsum_cl(const void* x, float* out, long rows, int C, int dtype, void* stream);

/* Channel extract / insert for channels-last tensors: out[row] = x[row][c];  y[row][:] = 0, y[row][c] = g[row].
 * Lets the Cout = 1 head use the MFMA convolution with zero-padded channels. */
int mv_take_channel(const void* x, void* out, long rows, int C, int c, int dtype, void* stream);
int mv_put_channel(const void* g, void* y, long rows, int C, int c, int dtype, void* stream);

/* ---- Conditioning producers (reference embedding_extractors.py).  Channels-last [B][T][C] activations;
 * the Conv1d / Linear layers run on mv_dconv_cl_fwd with BatchNorm folded into the packed weights. ---- */
/* y = LayerNorm(x + res) * gamma + beta over the last axis (nn.TransformerEncoderLayer post-norm, embedding_extractors.py:195-203);
 * res may be NULL; C % 8 == 0, C <= 4096. */
int mv_add_layernorm(const void* x, const void* res, const float* gamma, const float* beta, void* y, long rows, int C, float eps,
                     int dtype, void* stream);
/* out fp32 [B][C] = mean over T of x [B][T][C] (SE squeeze :166, utterance mean pooling :241). */
int mv_mean_t_cl(const void* x, float* out, int B, int T, int C, int dtype, void* stream);
/* gate[b][c] = sigmoid(W2 relu(W1 mean_b + b1) + b2): SE_Module.fc (embedding_extractors.py:159-164), fp32 masters W1 [R][C], W2 [C][R]. */
int mv_se_gate(const float* mean, const float* w1, const float* b1, const float* w2, const float* b2, float* gate, int B, int C,
               int R, void* stream);
/* y = x * gate[b][c] + res (SE scale :169 + block residual :149). */
int mv_scale_add_cl(const void* x, const float* gate, const void* res, void* y, int B, int T, int C, int dtype, void* stream);
/* Res2Net chain step (embedding_extractors.py:135-143): cat[row][dst_off..+cs) = src[row][0..cs) (src rows `src_stride` elements apart);
 * when nxt != NULL also nxt[row][0..cs) = u[row][nxt_off..+cs) + src[row][0..cs).  u, cat: [rows][C]; nxt: [rows][cs]. */
int mv_res2_glue(const void* src, int src_stride, const void* u, void* cat, void* nxt, long rows, int C, int cs, int dst_off,
                 int nxt_off, int dtype, void* stream);
/* Attentive statistics pooling (embedding_extractors.py:73-84): w = softmax over the C channels of logits[b][t][:], a = x * w,
 * pooled[b][0..C) = mean_t a, pooled[b][C..2C) = unbiased std_t a.  workspace: mv_asp_workspace_bytes(B, T). */
size_t mv_asp_workspace_bytes(int B, int T);
int mv_asp_pool(const void* x, const void* logits, void* workspace, float* pooled, int B, int T, int C, int dtype, void* stream);
/* y[b][:] = x[b][:] / max(||x[b]||_2, eps) (F.normalize, embedding_extractors.py:90,245); x fp32, y in `dtype`. */
int mv_l2norm_rows(const float* x, void* y, int B, int C, float eps, int dtype, void* stream);
/* Multi-head self-attention of nn.MultiheadAttention (embedding_extractors.py:195-203): qkv [B][T][3*H] = in_proj(x) (q | k | v),
 * out [B][T][H] = concat_h softmax(q_h k_h^T / sqrt(head_dim)) v_h, before out_proj.  16-bit storage with head_dim 16/32/64: MFMA
 * flash kernel; otherwise (fp32 parity grade) a scalar kernel. */
int mv_mha_fwd(const void* qkv, void* out, int B, int T, int nheads, int head_dim, int dtype, void* stream);
/* y [M][N] = act(x [M][K] W^T + bias) for short sequences (M of the order of 1e3 rows): split-K over the 8 waves of a workgroup, operands
 * straight from global memory, `packed` = mv_dconv_pack_pad(w [N][K][1][1], flip 0).  16-bit storage, K % 256 == 0 with K/256 in
 * {1,2,3,4,6,8}, N % 64 == 0; otherwise MV_ERR_UNSUPPORTED (callers then use mv_dconv_cl_fwd). */
int mv_gemm_cl_skinny(const void* x, const void* packed, const void* bias, void* y, int M, int K, int N, int act, float slope,
                      int dtype, void* stream);
/* Fused Res2Net chain of SE_Res2Block (embedding_extractors.py:135-143): u, cat [B][T][8*cs]; cat[:, 0:cs] = u[:, 0:cs],
 * cat[:, i] = conv_i(u[:, i] + cat[:, i-1]) for i = 1..7 (k = 3, dilation dil, 'same' zero padding).  packed = the 7
 * mv_dconv_pack(w_i [cs][cs][1][3]) images back to back, bias [7][cs] in `dtype`.  16-bit storage, cs 32 or 64, dil <= 4;
 * otherwise MV_ERR_UNSUPPORTED (callers then chain mv_dconv_cl_fwd + mv_res2_glue). */
int mv_res2_chain(const void* u, const void* packed, const void* bias, void* cat, int B, int T, int C, int cs, int dil, int dtype,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_VOCODER_H */
