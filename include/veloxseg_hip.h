/* veloxseg_hip.h -- C ABI of libveloxseg_hip.so (gfx950 / MI355X kernels for the VeloxSeg hot path).
 *
 * The reference (JinPLu/VeloxSeg) is pure PyTorch and has no FFI / operator registry: its hot path is
 * the aten calls issued by the model/ and utils/loss.py modules.  Each entry point below names the reference
 * lines whose arithmetic it replaces.  Conventions:
 *   - every tensor is fp32, contiguous, NCDHW (labels: int64 / int32 / uint8), device memory owned by
 *     the caller (PyTorch); the library never allocates;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*), never synchronises, never reads
 *     device memory on the host: all entry points are HIP-graph capturable;
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch/runtime failure, -3 unsupported
 *     shape); vx_last_error() returns a thread-local message.  Nothing aborts.
 *   - dropout sites take (seed_ptr, dstream, p): seed_ptr -> device uint64[2] {seed, step};
 *     p == 0 or seed_ptr == NULL disables dropout.  The same triple regenerates the mask in backward.
 */
#ifndef VELOXSEG_HIP_H
#define VELOXSEG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define VX_ABI_VERSION 1

int vx_abi_version(void);
const char* vx_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Convolutions (cubic kernel K, uniform stride S / padding P, groups G).  Weight layout = PyTorch Conv3d
 * (Cout, Cin/G, K, K, K).  Input = channel concat of x (first C1 channels) and x2 (rest); C1 <= 0 or
 * C1 == Cin means "x only".  ps > 1 stores / reads y through PixelShuffle(ps)
 * (model/components/superpixel.py:16).
 *   replaces: nn.Conv3d in conv_blocks.py:10-17 (DownConv), :51-58 (JLC grouped 1/3/5), :64-68 (JLC 1x1s),
 *             Decoder.py:54-57,73-76,150-158, Encoder.py:334-337 (+ torch.cat :344-347), PWA.py:291-298,
 *             attention_utils.py:56-57,141, MONAI PatchEmbed.proj (Encoder.py:150-156).
 *   ConvTranspose3d k2 s2 (conv_blocks.py:29-35) = the adjoint: forward -> vx_conv3d_bwd_data (bias via
 *   bias_like), input gradient -> vx_conv3d_fwd, weight gradient -> vx_conv3d_bwd_weight with (x, dy) swapped.
 * --------------------------------------------------------------------------------------------- */
int vx_conv3d_fwd(const float* x, const float* x2, int C1, const float* w, const float* bias, float* y,
                  int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream);
/* dx (=|+=) conv^T(dy); dx/dx2 split like the forward input; bias_like (per input channel) is added when not NULL */
int vx_conv3d_bwd_data(const float* dy, const float* w, const float* bias_like, float* dx, float* dx2, int C1,
                       int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps,
                       int accumulate, void* stream);
/* dw += x (*) dy ; db += sum dy   (float atomics into caller-zeroed / running gradient buffers; db may be NULL) */
int vx_conv3d_bwd_weight(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                         int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream);

/* LDS-tiled weight gradient for spatial kernels (same contract as vx_conv3d_bwd_weight): x halo tile in LDS, dy on the scalar path */
int vx_conv3d_bwd_weight_tiled(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                               int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream);
/* the same with a caller-owned workspace: blocks store partial sums (no float atomics on dw from ~1 k blocks), a second kernel folds them.
 * vx_conv3d_bwd_weight_ws_floats returns the workspace size in floats (0 = the plain entry is the better path, < 0 = error). */
/* stem DownConv (Conv3d k7 s4 p3, Encoder.py conv-chain stem): weight + bias gradient as fp32-MFMA tiles over an LDS halo; ws from
 * vx_down_wgrad_ws_floats (0 = shape not covered).  vx_down_wgrad_mfma returns 1 and launches nothing when the shape is not covered. */
int vx_down_wgrad_ws_floats(int B, int Cin, int Di, int Hi, int Wi, int Cout);
int vx_down_wgrad_mfma(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                       int B, int Cin, int Di, int Hi, int Wi, int Cout, void* stream);
/* (round 6) the same with max |x| handed over from the forward (vx_conv_mfma_fwd_mx left its bits at x_absmax; null = find it here): the f16-pipe kernel's scale pass reads dy only */
int vx_down_wgrad_mfma_mx(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats, const unsigned* x_absmax,
                          int B, int Cin, int Di, int Hi, int Wi, int Cout, void* stream);
int vx_down_wgrad_set_f16(int on);   /* A/B (tests): the stem weight gradient on the f16 matrix pipe (default, 128-wide rows, 16 output channels) or the fp32 MFMA kernel */
/* Weight gradient of a dense strided Conv3d (the DownConvs of encoder levels 2 - 4: Conv3d k3 s2 p1, conv_blocks.py:4-21) as a gather-GEMM on v_mfma_f32_16x16x4_f32: exact
 * fp32 products, dw +=, db += (db may be NULL).  _ok: 1 when the shape is covered (Cout 32 / 64 / 128). */
int vx_conv_wgrad_gather_ok(int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P);
int vx_conv_wgrad_gather_mfma(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, void* stream);
int vx_conv_wgrad_gather_set(int on);   /* A/B (tests) */
/* The three weight gradients of a JLC block (grouped convolutions k = 5 / 3 / 1, conv_blocks.py:51-58) at small volumes (<= 512 voxels per sample: the 8^3 / 4^3 levels of
 * the 128^3 configurations, 6^3 / 3^3 of the 96^3 ones) as gather-GEMMs on v_mfma_f32_16x16x4_f32 in ONE launch; dw += ; exact fp32 products. */
int vx_jlc_wgrad_gather_ok(int C, int G, int D, int H, int W);
int vx_jlc_wgrad_gather(const float* x, const float* g1, const float* g3, const float* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W, void* stream);
int vx_jlc_wgrad_gather_set(int on);   /* A/B (tests) */
int vx_conv3d_bwd_weight_ws_floats(int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps);
int vx_conv3d_bwd_weight_tiled_ws(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream);
/* A/B knob of the two entries above: 1 (default) = the JLC grouped convs (k 3 / 5, stride 1, Cin/G == Cout/G in {4, 8}, W % 4 == 0; conv_blocks.py:51-58)
 * take the row-sliding kernel (a thread owns a kw row of taps and slides the input row through registers), 0 = always the (ci, tap)-pair kernel */
int vx_wgrad_set_rows(int on);
/* weight + bias gradient of a 1x1x1 GROUPED conv with Cin == Cout == C (JLC k = 1 branch, conv_blocks.py:51-58): dw (C, C/G) +=, db (C) += (may be NULL).
 * Needs V % 4 == 0 and a group width of 4, 8 or 16; other shapes go through vx_conv3d_bwd_weight_tiled. */
int vx_gconv1_bwd_weight(const float* x, const float* dy, float* dw, float* db, int B, int C, int G, long V, void* stream);
/* stride-1 "same" conv, K in {3,5}, register-blocked + LDS-staged (JLC grouped convs, patch-expand).  w is always the forward
 * weight (Cf_out, Cf_in/G, K,K,K).  wmode 0 = forward (Cin=Cf_in, Cout=Cf_out); wmode 1 = input gradient (x:=dy, Cin=Cf_out, Cout=Cf_in).
 * in_ps / out_ps: PixelShuffle factor of the input / output storage.  accumulate: y += . */
int vx_conv_s1(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W,
               int K, int G, int wmode, int in_ps, int out_ps, int accumulate, void* stream);
/* patch-expand (Conv3d 16 -> 64*Cc, k3 p1, PixelShuffle 4) input gradient as an fp32-MFMA implicit GEMM reading the fine gradient and the
 * tap-major weights in 256-byte runs.  dy_fine: (B, Cc, 4D,4H,4W); w: (64*Cc, 16, 3,3,3); wt_ws: 64*Cc*16*27 floats; dx: (B,16,D,H,W) */
int vx_expand_bwd_data_mfma(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W,
                            int accumulate, void* stream);
/* A/B knob of the entry above: 1 (default) = LDS-tiled kernel (fine-gradient halo AND the group's weights in LDS) when D%4 == H%4 == W%16 == 0,
 * 2 = halo only (weights per tap from global memory), 0 = every operand straight from global */
int vx_expand_set_lds(int on);
int vx_expand_set_fwd_wlds(int on);   /* A/B knob of vx_expand_fwd_mfma: 1 (default) = the weights of a (c, s1) group staged in LDS once per block, 0 = loaded from global memory per tap */
/* forward of the same layer (conv 16 -> 64*Cc, k3 p1, PixelShuffle(4) store) as MFMA tiles over an LDS halo; wt_ws: Cout*16*27 floats.
 * Needs D % 4 == 0, H % 4 == 0, W % 16 == 0: returns 1 (and launches nothing) otherwise -- the caller then uses vx_conv_s1. */
int vx_expand_fwd_mfma(const float* x, const float* w, const float* bias, float* wt_ws, float* y, int B, int Cc, int D, int H, int W, void* stream);
/* patch-expand weight (+bias) gradient on fp32 MFMA: x (B,16,D,H,W) coarse input, xcl_ws = B*D*H*W*16 floats (channels-last copy made here),
 * dy_fine (B,Cc,4D,4H,4W); dw (64*Cc,16,3,3,3) +=, db (64*Cc) += */
int vx_expand_wgrad_mfma(const float* x, float* xcl_ws, const float* dy_fine, float* dw, float* db, int B, int Cc, int D, int H, int W, void* stream);
/* the same weight gradient with every product formed from ns bf16 pieces per fp32 operand (ns = 3: fp32-exact products, fp32 accumulation; ns = 1: plain
   bf16 operands, the bf16 opt-in mode); x is read in
   its own (B, 16, D, H, W) layout.  part_ws: vx_expand_wgrad_split_ws_floats(...) floats of partial sums, added into dw in a fixed order (reproducible).
   Returns 1 when W % 4 != 0 (use vx_expand_wgrad_mfma). */
int vx_expand_wgrad_split_ws_floats(int B, int Cc, int D, int H, int W);
int vx_expand_wgrad_mfma_split(const float* x, const float* dy_fine, float* dw, float* db, float* part_ws, long ws_floats, int B, int Cc, int D, int H, int W,
                               int ns, void* stream);
/* bf16 storage mode of the patch-expand layers (bf16 operands): the pixel-shuffled output y (forward) / the fine gradient dy (input and weight gradients) as bf16 arrays */
int vx_expand_fwd_mfma_bf16_h(const float* x, const float* w, const float* bias, float* wt_ws, void* y, int B, int Cc, int D, int H, int W, int y_h16, void* stream);
int vx_expand_bwd_data_mfma_bf16_h(const void* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, int dy_h16, void* stream);
int vx_expand_wgrad_mfma_split_h(const float* x, const void* dy_fine, float* dw, float* db, float* part_ws, long ws_floats, int B, int Cc, int D, int H, int W,
                                 int ns, int dy_h16, void* stream);
/* bf16 opt-in mode (BASELINE configs[1]: the BraTS bf16 line): the same two layers with bf16 MFMA operands (v_mfma_f32_16x16x32_bf16), fp32
 * accumulation and fp32 tensors in HBM; weights and activations are rounded to bf16 (nearest-even) on their way into the MFMA.  Same arguments and
 * workspaces as the fp32 entries above; return 1 = shape not covered (D, H % 4, W % 16), the caller then uses the fp32 entry. */
int vx_expand_fwd_mfma_bf16(const float* x, const float* w, const float* bias, float* wt_ws, float* y, int B, int Cc, int D, int H, int W, void* stream);
int vx_expand_bwd_data_mfma_bf16(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, void* stream);
/* 1x1x1 convolutions: thread-per-voxel with scalar-path weights (fwd / bwd_data; Cin % 4 == 0), fp32-MFMA GEMM over the voxel
 * axis for the weight gradient.  w: (Cout, Cin).  Same concat / accumulate conventions as vx_conv3d_*. */
int vx_pw_conv_fwd(const float* x, const float* x2, int C1, const float* w, const float* bias, float* y,
                   int B, int Cin, int Cout, long V, void* stream);
int vx_pw_conv_set_v4(int on);   /* A/B knob (default 1): large volumes (V >= 16384, Cout % 16 == 0, Cin <= 64) take the one-wave kernel with 16-byte accesses and the weight tile in LDS */
int vx_pw_conv_bwd_data(const float* dy, const float* w, float* dx, float* dx2, int C1,
                        int B, int Cin, int Cout, long V, int accumulate, void* stream);
int vx_pw_conv_bwd_weight(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                          int B, int Cin, int Cout, long V, void* stream);
int vx_pw_conv_fwd_h(const void* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, long V, int x_h16, void* stream);
int vx_pw_conv_bwd_weight_h(const void* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, long V, int x_h16, void* stream);

/* small-volume variant (fp32 MFMA tiles of 16 channels x 16 voxels): dst[b,m,v] = bias[m] + sum_k Wt(m,k) src[b,k,v].
 * forward: transpose_w=0, Mch=Cout, Kch=Cin; input gradient: transpose_w=1, Mch=Cin, Kch=Cout (src:=dy, dst:=dx).  Cin_of_w = row length of w.
 * src may be a concat (S1 channels from src, rest from src2); dst may be split the same way (D1). */
int vx_pw_conv_mfma(const float* src, const float* src2, int S1, const float* w, int transpose_w, const float* bias,
                    float* dst, float* dst2, int D1, int B, int Mch, int Kch, int Cin_of_w, long V, int accumulate, void* stream);
/* input gradient (=|+= into dx / dx2) and weight / bias gradient (+= into dw, db; db may be NULL) of a 1x1 conv in one launch; same results as
 * vx_pw_conv_mfma(transpose_w = 1) followed by vx_pw_conv_bwd_weight */
int vx_pw_conv_bwd_fused(const float* dy, const float* w, const float* x, const float* x2, int C1, float* dx, float* dx2, float* dw, float* db,
                         int B, int Cin, int Cout, long V, int accumulate, void* stream);
/* the same for large volumes (one voxel per thread for the input gradient: the vx_pw_conv_bwd_data kernel + vx_pw_conv_bwd_weight in one launch); Cin % 4 == 0 */
int vx_pw_conv_bwd_fused_big(const float* dy, const float* w, const float* x, const float* x2, int C1, float* dx, float* dx2, float* dw, float* db,
                             int B, int Cin, int Cout, long V, int accumulate, void* stream);
/* "1x1 conv -> GELU -> dropout -> 1x1 conv" stage of the JLC / FFN blocks (conv_blocks.py:64-68, attention_utils.py:56-66) with the element-wise
 * part in the conv epilogues: fwd writes the pre-activation a and h = drop(gelu(a)); bwd_data writes da = (W2^T dy) * mask * gelu'(a).
 * Same masks as vx_gelu_drop_fwd/_bwd on the same (seed_ptr, dstream, p).  mfma != 0: MFMA tile kernels (small volumes). */
int vx_pw_conv_gelu_fwd(const float* x, const float* w, const float* bias, float* a, float* h, int B, int Cin, int Cout, long V, int mfma,
                        const void* seed_ptr, unsigned long long dstream, float p, void* stream);
int vx_pw_conv_gelu_bwd_data(const float* dy, const float* w, const float* a, float* da, int B, int Cin, int Cout, long V, int mfma,
                             const void* seed_ptr, unsigned long long dstream, float p, void* stream);
/* out = alpha * res + drop(W x + bias): residual + dropout (vx_axpy_drop_fwd's mask) in the epilogue of the 1x1 conv that feeds it; res != out */
int vx_pw_conv_res_fwd(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin, int Cout, long V, int mfma,
                       float alpha, const void* seed_ptr, unsigned long long dstream, float p, void* stream);
/* A/B knob: 1 (default) = 16 x 64 tiles with 16-byte operand loads when V % 4 == 0, 0 = 16 x 16 tiles */
int vx_pw_mfma_set_wide(int on);

/* ---------------------------------------------------------------------------------------------
 * Fused blocks (round 2): the JLC block in 3 + 4 launches (+ 3 weight-gradient launches) and the channel MLP of the JLC / FFN stages in one
 * launch per direction.  Statistics travel as per-block PARTIAL SUMS that the consumer folds (no atomics, no zero-fill launches).
 *
 * vx_mlp_*: out = x + Drop2(W2 . Drop1(GELU(W1 . norm(x) + b1)) + b2), w1 (R, C), w2 (C, R)
 *   norm 0 = InstanceNorm3d(eps): JLC channel stage, conv_blocks.py:64-69 + the residual of :74.  Forward: `part` != NULL = (sum, sumsq)
 *            partials of x, [B*C][nparts][2] doubles (from vx_jlc_mid_fwd), folded here and written to `stats` (B*C, 2) = (mean, rstd);
 *            part == NULL = stats is an input.  Backward: stats is an input; dx receives dn = the gradient at the NORMALISED input and
 *            part_out [B*C][vx_mlp_bwd_nparts][2] floats receives the partials (sum dn, sum dn*nhat) of the InstanceNorm backward.
 *   norm 1 = channels-first LayerNorm(gamma, beta, eps): FFN tail of a PWA block, PWA.py:437 with attention_utils.py:45-71.  Backward: dx is the
 *            complete input gradient INCLUDING the residual branch (dout + ...); dgamma / dbeta are accumulated.
 *   dw1, db1, dw2, db2 are accumulated (float atomics, one per element and block).  Dropout: site1 / p1 on the hidden activation (masks of
 *   vx_gelu_drop_*), site2 / p2 on the output (masks of vx_axpy_drop_*).  vx_mlp_supported: (C, R) in {(16,48), (16,32)}, V % 4 == 0 (C >= 32 runs on the tile-GEMM chains, vx_inmlp_*).
 * --------------------------------------------------------------------------------------------- */
int vx_mlp_supported(int C, int R, long V);
int vx_mlp_bwd_nparts(int B, int C, long V);
int vx_mlp_fwd(const float* x, int norm, const double* part, int nparts, float* stats, const float* gamma, const float* beta,
               const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B, int C, int R, long V, float eps,
               const void* seed_ptr, unsigned long long site1, float p1, unsigned long long site2, float p2, void* stream);
int vx_mlp_bwd(const float* x, int norm, const float* stats, const float* gamma, const float* beta, const float* w1, const float* b1,
               const float* w2, const float* dout, float* dx, float* part_out, float* dgamma, float* dbeta, float* dw1, float* db1,
               float* dw2, float* db2, int B, int C, int R, long V, float eps, const void* seed_ptr, unsigned long long site1, float p1,
               unsigned long long site2, float p2, void* stream);
/* JLC spatial stage o = x + sum_{k=1,3,5} GELU(IN(gconv_k(x))) (conv_blocks.py:51-58,72-73); group width C/G a multiple of 4, V % 4 == 0.
 *   vx_jlc_conv_fwd: y1, y3, y5 = the three grouped convs from one LDS halo tile; part [3][B*C][vx_jlc_ntiles][2] doubles = (sum, sumsq) per tile
 *   vx_jlc_mid_fwd : stats_y [3][B*C][2] = (mean, rstd) of y_k (written); o; part_o [B*C][vx_jlc_nchunks][2] doubles = (sum, sumsq) of o per chunk
 *   vx_jlc_mid_bwd : d_o = dout + IN-backward(dn; part_dn from vx_mlp_bwd); part_t [3][B*C][vx_jlc_nchunks][2] floats = (sum t_k, sum t_k*yhat_k),
 *                    t_k = d_o * GELU'(yhat_k)
 *   vx_jlc_gk      : g_k = rstd_k (t_k - mean t_k - yhat_k mean(t_k yhat_k)), the gradients at the conv outputs
 *   vx_jlc_conv_bwd: dx = d_o + sum_k conv_k^T(g_k) */
int vx_jlc_ntiles(int B, int C, int G, int D, int H, int W);
int vx_jlc_nchunks(long BC, long V);
int vx_jlc_conv_fwd(const float* x, const float* w1, const float* w3, const float* w5, const float* b1, const float* b3, const float* b5,
                    float* y1, float* y3, float* y5, double* part, int B, int C, int G, int D, int H, int W, void* stream);
int vx_jlc_mid_fwd(const float* x, const float* y1, const float* y3, const float* y5, const double* part_y, int nty, float* stats_y, float* o,
                   double* part_o, long BC, long V, float eps, void* stream);
int vx_jlc_mid_bwd(const float* dout, const float* dn, const float* part_dn, int npd, const float* o, const float* stats_o, const float* y1,
                   const float* y3, const float* y5, const float* stats_y, float* d_o, float* part_t, long BC, long V, void* stream);
int vx_jlc_gk(const float* d_o, const float* y1, const float* y3, const float* y5, const float* stats_y, const float* part_t, float* g1, float* g3,
              float* g5, long BC, long V, void* stream);
int vx_jlc_conv_bwd(const float* g1, const float* g3, const float* g5, const float* w1, const float* w3, const float* w5, const float* d_o,
                    float* dx, int B, int C, int G, int D, int H, int W, void* stream);
/* bf16 STORAGE mode (round 6; BASELINE configs[1], reference speed_test.py:122,127 = torch.amp.autocast: 16-bit conv outputs, fp32 statistics and sums).  The *_h
 * entries are the entries above with the BLOCK-INTERNAL tensors of a JLC block -- the conv outputs y_k, o, and in the backward pass dn, d_o and the g_k -- as arrays of
 * 16-bit bf16 elements (`void*`) when h16 != 0 (h16 = 0: exactly the fp32 entries); the block's input x, its output and the incoming gradient dout stay fp32.  A stored
 * value is the round-to-nearest-even bf16 of the fp32 result; InstanceNorm statistics are taken over the ROUNDED values (what every reader loads). */
int vx_jlc_mid_fwd_h(const float* x, const void* y1, const void* y3, const void* y5, const double* part_y, int nty, float* stats_y, void* o,
                     double* part_o, long BC, long V, float eps, int h16, void* stream);
int vx_jlc_mid_bwd_h(const float* dout, const void* dn, const float* part_dn, int npd, const void* o, const float* stats_o, const void* y1,
                     const void* y3, const void* y5, const float* stats_y, void* d_o, float* part_t, long BC, long V, int h16, void* stream);
int vx_jlc_gk_h(const void* d_o, const void* y1, const void* y3, const void* y5, const float* stats_y, const float* part_t, void* g1, void* g3,
                void* g5, long BC, long V, int h16, void* stream);
/* vx_mlp_fwd / vx_mlp_bwd with x (the block's o) and the returned dx (= dn) as bf16 arrays (norm = 0, the JLC form, only) */
int vx_mlp_fwd_h(const void* x, int norm, const double* part, int nparts, float* stats, const float* gamma, const float* beta,
                 const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B, int C, int R, long V, float eps,
                 const void* seed_ptr, unsigned long long site1, float p1, unsigned long long site2, float p2, int h16, void* stream);
int vx_mlp_bwd_h(const void* x, int norm, const float* stats, const float* gamma, const float* beta, const float* w1, const float* b1,
                 const float* w2, const float* dout, void* dx, float* part_out, float* dgamma, float* dbeta, float* dw1, float* db1,
                 float* dw2, float* db2, int B, int C, int R, long V, float eps, const void* seed_ptr, unsigned long long site1, float p1,
                 unsigned long long site2, float p2, int h16, void* stream);

/* The same three grouped convolutions (conv_blocks.py:51-58) and their input gradient as Toeplitz GEMMs on the bf16 matrix pipe with fp32-exact products
 * (csrc/jlc_mfma.hip; group width 4 / 8 / 16, W % 4 == 0):
 *   vx_jlc_tz_ok / _ntiles  : answers (1 = shape supported; tiles per (b, c) = rows of `part`, a function of (C/G, D, H, W) only)
 *   vx_jlc_tz_img_floats    : floats of the operand-image workspace (forward + input-gradient images of the three weight tensors)
 *   vx_jlc_tz_prep          : expands w1 / w3 / w5 into the images (once per step; the backward of the same step reads the second half)
 *   vx_jlc_tz_fwd / _bwd    : drop-in for vx_jlc_conv_fwd / vx_jlc_conv_bwd (same outputs, same `part` layout)
 *   vx_jlc_tz_set_pieces    : pieces per fp32 operand: 3 / 2 / 1 bf16 pieces (six / three / one piece products; 1 = the bf16 opt-in mode) or 22 = two scaled fp16
 *                             pieces (22 significant bits, three piece products; what veloxseg_amd.functional selects for the fp32 mode) */
int vx_jlc_tz_ok(int C, int G, int D, int H, int W);
int vx_jlc_tz_ntiles(int C, int G, int D, int H, int W);
int vx_jlc_tz_img_floats(int C, int G);
int vx_jlc_tz_set_pieces(int ns);
int vx_jlc_tz_set_min_voxels(long v);   /* vx_jlc_tz_ok answers 0 below this many voxels per channel (default 1024: the 12^3 levels and up) */
int vx_jlc_tz_pieces(void);
int vx_jlc_tz_set_debug(int mask);      /* timing experiments only: bit 0 = skip the halo staging, bit 1 = skip the MFMA loops (results are then garbage) */
int vx_jlc_tz_prep(const float* w1, const float* w3, const float* w5, float* img, int C, int G, void* stream);
int vx_jlc_tz_fwd(const float* x, const float* img, const float* b1, const float* b3, const float* b5, float* y1, float* y3, float* y5, double* part,
                  int B, int C, int G, int D, int H, int W, void* stream);
int vx_jlc_tz_bwd(const float* g1, const float* g3, const float* g5, const float* img, const float* w1, const float* d_o, float* dx,
                  int B, int C, int G, int D, int H, int W, void* stream);
/* The coarse levels (<= 8 voxels per axis: 8^3 / 4^3 of the 128^3 configurations, 6^3 / 3^3 of the 96^3 ones; group width 8 / 16) as channels-last implicit GEMMs on
 * the f16 matrix pipe, two scaled fp16 pieces per operand (csrc/jlc_cl.hip; conv_blocks.py:51-58,72-75) -- the shapes the Toeplitz kernels above do not take.  Same
 * contracts: _ok / _ntiles answers, _img_floats + _prep the per-step weight images, _fwd / _bwd drop-ins for vx_jlc_conv_fwd / vx_jlc_conv_bwd (same outputs, same
 * `part` layout with vx_jlc_cl_ntiles rows per (b, c)).  vx_jlc_cl_set_enabled(0): A/B knob (vx_jlc_cl_ok then answers 0). */
int vx_jlc_cl_ok(int C, int G, int D, int H, int W);
int vx_jlc_cl_ntiles(int C, int G, int D, int H, int W);
int vx_jlc_cl_img_floats(int C, int G);
int vx_jlc_cl_set_enabled(int on);
int vx_jlc_cl_prep(const float* w1, const float* w3, const float* w5, float* img, int C, int G, void* stream);
int vx_jlc_cl_fwd(const float* x, const float* img, const float* b1, const float* b3, const float* b5, float* y1, float* y3, float* y5, double* part,
                  int B, int C, int G, int D, int H, int W, void* stream);
int vx_jlc_cl_bwd(const float* g1, const float* g3, const float* g5, const float* img, const float* d_o, float* dx, int B, int C, int G, int D, int H, int W,
                  void* stream);
/* the three weight gradients of the same convolutions in ONE launch on the matrix pipe (accumulated into dw1 / dw3 / dw5 with float atomics, like every weight-gradient
 * entry; a null dw skips that tensor's store); W <= 32, W % 4 == 0, H % 4 == 0, group width 4 / 8 / 16.  Bias gradients are not computed (zero behind an InstanceNorm). */
int vx_jlc_wgrad_tz_ok(int C, int G, int D, int H, int W);
int vx_jlc_wgrad_tz_set_min_voxels(long v);
int vx_jlc_wgrad_tz_set_blocks(int n);     /* blocks per launch; 0 (default): 128 with producer / consumer waves -- half the chip: the step is fastest there -- 256 for the one-role kernel (VELOXSEG_WG_TZ_BLOCKS) */
int vx_jlc_wgrad_tz_set_spec(int on);      /* 1 (default; VELOXSEG_WG_SPEC): 512-thread blocks, waves 4..7 stage step dx + 1 while waves 0..3 run the MFMAs of step dx; 0: the one-role kernel (A/B, tests) */
int vx_jlc_wgrad_tz_set_f16(int on);     /* with vx_jlc_tz_set_pieces(22): 1 = the weight gradients on two scaled fp16 pieces as well (A/B; default 0 = three bf16 pieces) */
int vx_jlc_wgrad_tz(const float* x, const float* g1, const float* g3, const float* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                    void* stream);
/* the same entries with the pieces mode of the operand image given explicitly: the image is laid out for the mode in force when vx_jlc_tz_prep built it, and its
 * consumers (input gradient, deferred weight gradients) must use that mode whatever the process-wide switch says by then (operator code records vx_jlc_tz_pieces()
 * at forward time).  Reference: the three grouped convolutions of conv_blocks.py:51-58 and their autograd. */
int vx_jlc_tz_img_floats_ns(int C, int G, int pieces);
int vx_jlc_tz_prep_ns(const float* w1, const float* w3, const float* w5, float* img, int C, int G, int pieces, void* stream);
int vx_jlc_tz_fwd_ns(const float* x, const float* img, const float* b1, const float* b3, const float* b5, float* y1, float* y3, float* y5, double* part,
                     int B, int C, int G, int D, int H, int W, int pieces, void* stream);
int vx_jlc_tz_bwd_ns(const float* g1, const float* g3, const float* g5, const float* img, const float* w1, const float* d_o, float* dx,
                     int B, int C, int G, int D, int H, int W, int pieces, void* stream);
int vx_jlc_wgrad_tz_ns(const float* x, const float* g1, const float* g3, const float* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                       int pieces, void* stream);
/* bf16 storage mode (h16 != 0 needs pieces = 1, plain bf16 operands): y_k (forward outputs), g_k and d_o (backward inputs) as bf16 arrays; x, dx and the weight
 * gradients stay fp32 */
int vx_jlc_tz_fwd_h(const float* x, const float* img, const float* b1, const float* b3, const float* b5, void* y1, void* y3, void* y5, double* part,
                    int B, int C, int G, int D, int H, int W, int pieces, int h16, void* stream);
int vx_jlc_tz_bwd_h(const void* g1, const void* g3, const void* g5, const float* img, const float* w1, const void* d_o, float* dx,
                    int B, int C, int G, int D, int H, int W, int pieces, int h16, void* stream);
int vx_jlc_wgrad_tz_h(const float* x, const void* g1, const void* g3, const void* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                      int pieces, int h16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * InstanceNorm3d(affine=False, eps) = stats + apply  (common_function.py:63-66; used at conv_blocks.py:18,36,54,65,
 * Encoder.py:334-337, Decoder.py:54-57).  stats[2*bc] = mean, stats[2*bc+1] = rstd.
 *   vx_in_apply_fwd: out = (res?res:0) + sum_{k<nk} act((y_k-mean_k)*rstd_k), act 0=identity 1=exact GELU
 *                    (JLC spatial sum conv_blocks.py:73; DownConv+attn2conv add Encoder.py:351-360; UpConv+skip Decoder.py:85-87)
 *   vx_in_bwd:       dy = rstd*(dz - mean(dz) - z*mean(dz*z)), dz = dout*act'(z); m_ws = 2*BC floats, part_ws = 32*BC doubles of workspace
 * --------------------------------------------------------------------------------------------- */
int vx_in_stats(const float* x, float* stats, double* part_ws, long BC, long V, float eps, void* stream);   /* part_ws: 32*BC doubles */
int vx_in_apply_fwd(const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                    int nk, int act, const float* res, float* out, long BC, long V, void* stream);
int vx_in_bwd(const float* dout, const float* y, const float* stats, int act, float* m_ws, double* part_ws, float* dy, long BC, long V, void* stream);
/* short rows (V <= vx_in_row_max() = 4096): statistics + application of up to 3 inputs in ONE launch (block = one (b,c) row held in registers);
 * s_k (BC,2) receive (mean, rstd); bwd writes dy_k for every non-NULL d_k.  Same arithmetic as the split kernels above. */
int vx_in_row_max(void);
int vx_in_row_fwd(const float* y0, const float* y1, const float* y2, float* s0, float* s1, float* s2, int nk, int act, const float* res,
                  float* out, long BC, long V, float eps, void* stream);
int vx_in_row_bwd(const float* dout, const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                  int nk, int act, float* d0, float* d1, float* d2, long BC, long V, void* stream);
/* the two backward entries with the bias gradient of the producing conv fused in: db_k[c] += sum_{b,v} dy_k (NULL = skip); BC = B*C */
int vx_in_bwd_db(const float* dout, const float* y, const float* stats, int act, float* m_ws, double* part_ws, float* dy, long BC, long V, float* db, int C, void* stream);
/* the same InstanceNorm-sum on long rows in two launches per direction: one partial-sum launch for all nk inputs, and an apply launch whose
 * blocks fold their row's partials themselves (no finalisation launches).  part_ws: nk * BC * 32 doubles.  dy_k may be NULL. */
int vx_in_fwd_split(const float* y0, const float* y1, const float* y2, float* s0, float* s1, float* s2, double* part_ws,
                    int nk, int act, const float* res, float* out, long BC, long V, float eps, void* stream);
int vx_in_bwd_split(const float* dout, const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                    double* part_ws, int nk, int act, float* dy0, float* dy1, float* dy2, const float* add0, long BC, long V, void* stream);   /* add0 (optional): dy0 = add0 + gradient */
int vx_in_row_bwd_add(const float* dout, const float* y0, const float* stats0, int act, const float* add0, float* d0, long BC, long V, void* stream);   /* short rows, single input: d0 = add0 + gradient */
int vx_in_row_bwd_db(const float* dout, const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                     int nk, int act, float* d0, float* d1, float* d2, float* db0, float* db1, float* db2, int C, long BC, long V, void* stream);

/* channels-first LayerNorm over C per voxel, biased variance (attention_utils.py:29-43) */
int vx_ln_cf_fwd(const float* x, const float* gamma, const float* beta, float* out, int B, int C, long V, float eps, void* stream);
int vx_ln_cf_bwd(const float* x, const float* gamma, const float* dout, float* dx, float* dgamma, float* dbeta, float* ws,
                 int B, int C, long V, float eps, void* stream);
int vx_ln_cf_bwd_add(const float* x, const float* gamma, const float* dout, const float* add, float* dx, float* dgamma, float* dbeta, float* ws,
                     int B, int C, long V, float eps, void* stream);   /* dx = add + LayerNorm backward (add != dx) */   /* dgamma/dbeta: += ; ws = 2*B*V floats of workspace */
/* the two halves of vx_ln_cf_bwd(_add) as entries of their own (attention_utils.py:29-43 backward): _data = input gradient (add may be NULL) and the
 * per-voxel statistics in ws (2*B*V floats); _param = dgamma / dbeta (+=) from x, dout and ws.  Only the first is on the backward's dependent chain. */
int vx_ln_cf_bwd_data(const float* x, const float* gamma, const float* dout, const float* add, float* dx, float* ws, int B, int C, long V, float eps, void* stream);
int vx_ln_cf_bwd_param(const float* x, const float* dout, const float* ws, float* dgamma, float* dbeta, int B, int C, long V, void* stream);

/* element-wise pieces: h = drop(gelu(a)) (attention_utils.py:64-66, conv_blocks.py:66); out = alpha*x + drop(z)
 * (PWA.py:377 + :436 double residual, attention_utils.py:68-70, conv_blocks.py:69,74, Encoder.py:196) */
int vx_gelu_drop_fwd(const float* a, float* h, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream);
int vx_gelu_drop_bwd(const float* dh, const float* a, float* da, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream);
int vx_axpy_drop_fwd(const float* x, const float* z, float* out, float alpha, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream);
int vx_axpy_drop_bwd(const float* dout, float* dx, float* dz, float alpha, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream);
int vx_add(const float* a, const float* b, const float* c, float* out, long n, void* stream);          /* out = a + b (+ c) */
/* out[k] = a[k] + b[k] (+ c[k]; c or c[k] may be NULL), k < count <= 16 tensors of n[k] floats, one launch (host arrays of device pointers) */
int vx_add_many(const float* const* a, const float* const* b, const float* const* c, float* const* out, const long* n, int count, void* stream);
int vx_channel_sum(const float* dy, float* db, int B, int C, long V, void* stream);                  /* db[c] += sum_{b,v} dy */
/* PatchMerging.faeture_sample (attention_utils.py:144-159); Dc,Hc,Wc = coarse dims; inverse=1 is the adjoint */
int vx_space_to_depth2(const float* x, float* out, int B, int C, int Dc, int Hc, int Wc, int inverse, void* stream);
/* out[b, c*K^3 + (kd*K+kh)*K + kw, z,y,x] = x[b, c, z*K+kd, y*K+kh, x*K+kw] (K = 2, 3, 4; d,h,w = OUTPUT grid): turns a kernel == stride conv
 * (PatchEmbed, Encoder.py:150-156) into a 1x1 conv over C*K^3 channels with the conv weight viewed as (Cout, C*K^3) */
int vx_patchify(const float* x, float* out, int B, int C, int d, int h, int w, int K, void* stream);
/* the same for a channel slice of a wider tensor: batch_stride = floats between consecutive samples of x (the slice itself contiguous within a sample) */
int vx_patchify_bs(const float* x, long batch_stride, float* out, int B, int C, int d, int h, int w, int K, void* stream);
/* bf16 storage mode: the patchified copy as a bf16 array (out_h16 != 0; patch size 2 / 4), read by vx_pw_conv_fwd_h / vx_pw_conv_bwd_weight_h with x_h16 != 0 */
int vx_patchify_bs_h(const float* x, long batch_stride, void* out, int B, int C, int d, int h, int w, int K, int out_h16, void* stream);
/* (round 6) the patch embedding itself (kernel == stride == 4, no padding; Encoder.py:150-156, MONAI PatchEmbed "conv") WITHOUT a patchified copy: the
 * product and its weight gradient gather the 4 x 4 x 4 patches from x by address.  Do, Ho, Wo: the OUTPUT grid (x is Cin x 4Do x 4Ho x 4Wo per sample,
 * batch_stride floats apart); Cout % 16 == 0, Wo % 4 == 0, Cin <= 8.  The forward sums in the order of vx_patchify + vx_pw_conv_fwd (bit-identical).
 * vx_patch_embed_ok: 1 where the dispatcher takes these kernels (VELOXSEG_PATCH_FUSED=0: never) */
int vx_patch_embed_ok(int Cin, int Cout, int Do, int Ho, int Wo, int K);
int vx_patch_embed_fwd(const float* x, long batch_stride, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Do, int Ho, int Wo, void* stream);
int vx_patch_embed_bwd_weight(const float* x, long batch_stride, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int Do, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Paired-Window Attention (model/components/PWA.py).  Geometry of one layer:
 * --------------------------------------------------------------------------------------------- */
typedef struct VxPwaPlan {
    int grid[3];      /* token grid g */
    int n[3];         /* tokens per window and axis = min_big / min_small (PWA.py:42) */
    int heads, nb;    /* heads, number of window scales (PWA.py:67 `while (bw <= input).any()`) */
    int small[4][3];  /* max-pool size of scale i = min_small * 2^i */
    int nwin[4][3];   /* windows per axis of scale i */
    int woff[4];      /* offset of scale i on the concatenated window axis */
    int Ntot;         /* total windows */
    int l;            /* n[0]*n[1]*n[2] tokens per window per modality */
} VxPwaPlan;

/* window_gathering_3d (PWA.py:106-140) of modality m into tok[B, heads, Ntot, M*l, c]; bwd routes to the first arg-max */
int vx_pwa_gather_fwd(const float* src, float* tok, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream);
int vx_pwa_gather_bwd(const float* src, const float* dtok, float* dsrc, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream);
/* the same gather for ALL 3*M tensors (q0,k0,v0,q1,...) in ONE launch; the arg-max voxel of every pooled cell is saved (int32, token layout) and
 * drives the backward pass (srcs / dsrcs: host arrays of 3*M device pointers) */
int vx_pwa_gather_all_fwd(const float* const* srcs, float* tq, float* tk, float* tv, int* iq, int* ik, int* iv,
                          const VxPwaPlan* plan, int cq, int cv, int M, int B, void* stream);
int vx_pwa_gather_all_bwd(const float* dtq, const float* dtk, const float* dtv, const int* iq, const int* ik, const int* iv, float* const* dsrcs,
                          const VxPwaPlan* plan, int cq, int cv, int M, int B, void* stream);
/* window_scattering_3d (PWA.py:177-200): per-window trilinear, align_corners=True */
int vx_pwa_scatter_fwd(const float* tok, float* out, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream);
int vx_pwa_scatter_bwd(const float* dout, float* dtok, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream);   /* dtok += (caller zeroes it once for all modalities) */
/* all M <= 4 modalities in one launch per kernel kind (host arrays of M device pointers) */
int vx_pwa_scatter_fwd_all(const float* tok, float* const* outs, const VxPwaPlan* plan, int c, int M, int B, void* stream);
int vx_pwa_scatter_bwd_all(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, void* stream);
/* (round 6) the same into a destination the caller did NOT zero: the sole-owner kernels assign and the identity-scale kernel zeroes the window ranges of the
 * scales that add with atomics.  Returns 1 and launches nothing where that does not apply (scale 0 not an identity scale): zero dtok and call vx_pwa_scatter_bwd_all. */
int vx_pwa_scatter_bwd_all_w(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, void* stream);
/* ... and one more buffer zeroed by the same launch (16-byte aligned, extra_floats % 4 == 0; NULL: none): the bias-gradient replicas of the attention backward that follows */
int vx_pwa_scatter_bwd_all_wz(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, float* extra_zero, long extra_floats, void* stream);
int vx_pwa_attn_set_fused_bwd(int on); /* A/B knob: 1 (default) = the dQ and dK/dV passes of vx_pwa_attn_bwd share one launch (interleaved blocks), 0 = two launches */
int vx_pwa_scatter_set_ident(int on);   /* A/B knob: 1 (default) = 1x1x1 small windows take the transpose kernel, 0 = always the general adjoint */
/* MultiModal attention_operation (PWA.py:308-327) + relative bias (attention_utils.py:120-125); table = (Tsz, heads).
 * O: (B,heads,Ntot,M*l,cv); LSE: (B,heads,Ntot,M*l).  bwd: dtable += ; delta_ws = vx_pwa_attn_bwd_ws_floats(plan, B, M) floats
 * (row-sum workspace + replicated bias-gradient tables; contents need not be initialised). */
int vx_pwa_attn_fwd(const float* Q, const float* K, const float* V, const float* table, float* O, float* LSE,
                    const VxPwaPlan* plan, int B, int M, int cq, int cv,
                    const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream);
int vx_pwa_attn_bwd_ws_floats(const VxPwaPlan* plan, int B, int M);   /* answer, not a status; negative = error */
/* (round 6) float offset of the bias-gradient replicas inside delta_ws (they run to its end; answer, not a status), and a one-shot note that the caller has zeroed them:
 * the next attention backward of this host thread on that workspace skips its zeroing launch */
int vx_pwa_attn_bwd_rep_offset(const VxPwaPlan* plan, int B, int M);
int vx_pwa_attn_bwd_mark_rep_zeroed(const float* delta_ws);
int vx_pwa_gather_set_vec(int on);   /* A/B knob for tests: 0 = the one-lane-per-(cell, channel) gather kernels instead of the channel-vectorised ones */
int vx_pwa_attn_bwd(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                    const float* dO, float* dQ, float* dK, float* dV, float* dtable, float* delta_ws,
                    const VxPwaPlan* plan, int B, int M, int cq, int cv,
                    const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream);
/* vx_pwa_attn_bwd without its last step: the bias-table gradient (attention_utils.py:120-125) stays in the replicas inside delta_ws until
 * vx_pwa_attn_bwd_fold adds them to dtable.  _nofold returns 1 (nothing launched) when this geometry / mode folds inside its kernels: use vx_pwa_attn_bwd. */
int vx_pwa_attn_bwd_nofold(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                           const float* dO, float* dQ, float* dK, float* dV, float* delta_ws,
                           const VxPwaPlan* plan, int B, int M, int cq, int cv,
                           const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream);
int vx_pwa_attn_bwd_fold(const float* delta_ws, float* dtable, const VxPwaPlan* plan, int B, int M, void* stream);
/* the folds of up to 8 vx_pwa_attn_bwd_nofold* calls (same B, M) in one launch: host arrays of the calls' delta_ws / dtable / plan pointers */
int vx_pwa_attn_bwd_fold_many(const float* const* delta_ws, float* const* dtables, const VxPwaPlan* const* plans, int count, int B, int M, void* stream);
/* tuning knob of the two entries above: key/query split per 64-row unit (0 = automatic from the unit count, else 1, 2 or 4; clamped to the
 * number of 64-row slabs).  Results are identical up to fp32 summation order; the dropout mask does not depend on it. */
int vx_pwa_attn_set_split(int S);
/* The two entries above run on the matrix cores (csrc/pwa_mfma.hip; head widths (4,4) (8,8) (8,16) (16,32) (16,16) (4,8); same masks, same results up
 * to fp32 summation order).  Forward: QK^T and PV as v_mfma_f32_16x16x4_f32 tiles, K/V of a window staged once per 128-query block, when a window's tokens
 * tile into 16-token blocks (l % 64 == 0).  Backward: ONE pass per (query, key) pair -- S, dP, dS once, then dV, dK, dQ (through a wave-private LDS
 * transpose) and d(bias) -- for any l and M <= 2; selected (vx_pwa_attn_bwd1_ok) where it was measured faster than the VALU kernels: windows whose token count
 * is not a multiple of 16 (the 27- / 216-token windows of the shipped 96^3 configurations).  vx_pwa_attn_set_mfma(mask): bit 0 forward, bit 1 backward (that
 * rule), bit 3 the one-pass backward for every geometry it covers, bit 2 (A/B only) the older two-kernel MFMA backward; default 3.  vx_pwa_attn_mfma_ok answers bit 0 (forward) | bit 1 (two-kernel backward selected)
 * for a geometry those kernels cover and 0 otherwise. */
int vx_pwa_attn_bwd1_ok(const VxPwaPlan* plan, int B, int M, int cq, int cv);
/* Dropout mask words: with p_drop > 0 the forward can keep ONE BIT per (query row, key) -- uint16 W[B*heads*Ntot][ceil(ML/16)][ML], word w of row r = keys
 * 16w .. 16w+15, bit k & 15, 1 = kept; vx_pwa_attn_mbits_words = number of uint16 -- and the one-pass backward reads it instead of drawing the Philox words
 * again (the draw is ~1/3 of the backward's per-pair instruction count).  `mbits` NULL or p_drop == 0: exactly vx_pwa_attn_fwd / _bwd / _bwd_nofold. */
int vx_pwa_attn_mbits_words(const VxPwaPlan* plan, int B, int M);
int vx_pwa_attn_mbits_useful(const VxPwaPlan* plan, int B, int M, int cq, int cv);
int vx_pwa_attn_set_valu_bits(int on);      /* A/B (tests): 1 = the fp32-VALU backward reads the forward's keep bits where windows are aligned (default 0: measured no faster) */   /* 1: the one-pass backward is selected for this geometry and reading the bits beats re-drawing (l % 4 != 0) */
int vx_pwa_attn_fwd_mb(const float* q, const float* k, const float* v, const float* table, float* out, float* lse, const VxPwaPlan* plan, int B, int M, int cq, int cv,
                       const void* seed_ptr, unsigned long long dstream, float p_drop, void* mbits, void* stream);
int vx_pwa_attn_bwd_mb(const float* q, const float* k, const float* v, const float* table, const float* out, const float* lse, const float* dout,
                       float* dq, float* dk, float* dv, float* dtable, float* delta_ws, const VxPwaPlan* plan, int B, int M, int cq, int cv,
                       const void* seed_ptr, unsigned long long dstream, float p_drop, const void* mbits, void* stream);
int vx_pwa_attn_bwd_nofold_mb(const float* q, const float* k, const float* v, const float* table, const float* out, const float* lse, const float* dout,
                              float* dq, float* dk, float* dv, float* delta_ws, const VxPwaPlan* plan, int B, int M, int cq, int cv,
                              const void* seed_ptr, unsigned long long dstream, float p_drop, const void* mbits, void* stream);
int vx_pwa_attn_mfma_ok(const VxPwaPlan* plan, int B, int M, int cq, int cv);
int vx_pwa_attn_set_mfma(int on);
/* One-pass backward on the 16x16x32 f16 matrix pipe (csrc/pwa_mfma.hip vx_pwa_attn_bwd1h_k; PWA.py:308-327): windows of 64 / 512 tokens (l % 64 == 0), M = 2,
 * head widths (4, 4) and (8, 8) -- levels 1 and 2 of the 128^3 configurations.  Every operand enters as TWO fp16 pieces of the value scaled by a power of two
 * taken from the block's own maxima (22 mantissa bits, exact rescale of the fp32 accumulators); the four piece products of S and dP share ONE MFMA (the
 * reduction dimension is only 4 / 8 wide).  With p_drop > 0 it reads the forward's keep bits (vx_pwa_attn_mbits_useful answers 1; without them the fp32
 * kernels run).  vx_pwa_attn_set_f16_bwd(0): A/B knob, those geometries back on the fp32 kernels; default 1 (selected when bit 1 of vx_pwa_attn_set_mfma is set). */
int vx_pwa_attn_bwd1h_ok(const VxPwaPlan* plan, int B, int M, int cq, int cv);
int vx_pwa_attn_set_f16_bwd(int on);
int vx_pwa_attn_set_f16_bwd_ragged(int on);   /* A/B (round 6): window lengths that are not multiples of 64 (27 / 216 / 32 tokens: the shipped 96^3 and Hecktor geometries) on the f16-pipe backward too -- padded 64-token chunks: 0 off, 1 (default, VELOXSEG_F16_BWD_RAGGED) where a window fills >= 3/4 of its chunks (216 tokens), 2 every ragged length (tests) */
int vx_pwa_attn_set_f16_bwd_m1(int on);       /* A/B (round 6): windows of ONE modality (BraTS) on the f16-pipe backward (MF = 1 instance): 0 (default, VELOXSEG_F16_BWD_M1: measured no gain at BraTS' batch of 2), 1 on, 2 on for >= 512 tokens */
int vx_pwa_attn_set_short(int on);       /* 1 (default; VELOXSEG_B1_SHORT): single-modality windows below 512 tokens take the one-pass MFMA backward although l % 16 == 0 (BraTS: 2 x faster than the VALU kernels there); 0: the selection of rounds 3 - 5 (A/B, tests) */

/* ---------------------------------------------------------------------------------------------
 * Loss side (utils/loss.py:30-66, common_function.py:8-14, VeloxSeg.py:177-184)
 *   labels kind: 0 int64, 1 int32, 2 uint8.  seg acc (double): per head [ce_sum, (I,P,T) x (b,c)].
 *   coef (float): per head [w_ce, (alpha,beta) x (b,c)], then [rc_coef, gram_coef].
 * --------------------------------------------------------------------------------------------- */
int vx_seg_loss_fwd(const float* l0, const float* l1, const float* l2, const float* l3, int nh, const void* labels, int lab_kind,
                    double* acc, int B, int C, long V, void* stream);
int vx_sqdiff_sum(const float* a, const float* b, long n, double* acc, void* stream);
/* acc += sum (a - b)^2 with b a channel slice of a wider tensor (b_batch_stride floats between its samples); acc is NOT zeroed: several
 * reconstruction decoders add into one accumulator from their own streams */
int vx_sqdiff_sum_bs(const float* a, const float* b, long n_per_sample, long b_batch_stride, int B, double* acc, void* stream);
int vx_mse_bwd_bs(const float* a, const float* b, long n_per_sample, long b_batch_stride, int B, const float* coef, const float* gout, float* da, void* stream);
/* the sum of squares of vx_sqdiff_sum_bs AND da = scale (a - b) in one pass (staged loss: scale = 2 w_rc / N_rc is known at forward time, utils/loss.py:52-66) */
int vx_sqdiff_sum_grad_bs(const float* a, const float* b, long n_per_sample, long b_batch_stride, int B, double* acc, float scale, float* da, void* stream);
/* bf16 storage mode: a (the reconstruction) and da as bf16 arrays when a_h16 != 0; b (the network input) stays fp32 */
int vx_sqdiff_sum_grad_bs_h(const void* a, const float* b, long n_per_sample, long b_batch_stride, int B, double* acc, float scale, void* da, int a_h16, void* stream);
int vx_loss_finalize(const double* seg_acc, int nh, int B, int C, long V, const float* head_weights,
                     const double* rc_acc, long n_rc, float w_rc,
                     const float* gram_seg, const float* g0, const float* g1, const float* g2, const float* g3, int M, int Cg, float w_f,
                     float* loss_out, float* coef, void* stream);
int vx_seg_loss_bwd(const float* logits, const void* labels, int lab_kind, const float* coef_head, const float* gout,
                    float* dlogits, int B, int C, long V, void* stream);
/* all nh <= 4 heads in one launch (labels read once); head h reads its coefficients at coef + h * coef_stride floats */
int vx_seg_loss_bwd4(const float* lg0, const float* lg1, const float* lg2, const float* lg3, int nh, const void* labels, int lab_kind, const float* coef,
                     int coef_stride, const float* gout, float* dl0, float* dl1, float* dl2, float* dl3, int B, int C, long V, void* stream);
/* Deep-supervision loss with the trilinear up-sampling of VeloxSeg.py:177-184,202 fused in (csrc/loss_ds.hip): head 0 (l0) is full resolution
 * (B, C, D, H, W); heads 1..nh-1 (l1..l3) stay on their own grids, low_dims = host array of 3*(nh-1) ints (d, h, w per head), and are interpolated
 * on the fly (align_corners=True, the arithmetic of vx_upsample_trilinear_fwd).  Same acc / coef layouts as vx_seg_loss_fwd / vx_loss_finalize /
 * vx_seg_loss_bwd4.  Backward: dl0 full resolution, dl1..dl3 on the heads' grids (the adjoint of the interpolation runs inside the kernels);
 * ws = vx_seg_loss_ds_ws_floats(...) floats of workspace.  vx_seg_loss_ds_ok: 1 when the shape is covered (C in 2..4, W % 4 == 0, W/4 divides 64). */
int vx_seg_loss_ds_ok(int C, int D, int H, int W);
int vx_seg_loss_ds_set_columns(int on);      /* 1 (default; VELOXSEG_DS_COLUMNS): the column-owner kernels; 0: the row-sweep kernels (A/B, tests) */
int vx_seg_loss_ds_ws_floats(const int* low_dims, int nh, int B, int C, int D);
int vx_seg_loss_ds_fwd(const float* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                       double* acc, int B, int C, int D, int H, int W, void* stream);
int vx_seg_loss_ds_bwd(const float* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                       const float* coef, int coef_stride, const float* gout, float* dl0, float* dl1, float* dl2, float* dl3, float* ws,
                       int B, int C, int D, int H, int W, void* stream);
/* bf16 storage mode: head 0 (the full-resolution logits) and its gradient dl0 as bf16 arrays when l0_h16 != 0 (column-owner kernels only: vx_seg_loss_ds_h16_ok answers 1
 * when both directions run them at this geometry); the low-resolution heads, accumulators, coefficients and the workspace stay fp32 */
int vx_seg_loss_ds_h16_ok(const int* low_dims, int nh, int B, int C, int D, int H, int W);
int vx_seg_loss_ds_fwd_h(const void* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                         double* acc, int B, int C, int D, int H, int W, int l0_h16, void* stream);
int vx_seg_loss_ds_bwd_h(const void* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                         const float* coef, int coef_stride, const float* gout, void* dl0, float* dl1, float* dl2, float* dl3, float* ws,
                         int B, int C, int D, int H, int W, int l0_h16, void* stream);
int vx_mse_bwd(const float* a, const float* b, const float* coef, const float* gout, float* da, long n, void* stream);
int vx_gram_mse_bwd(const float* gs, const float* g0, const float* g1, const float* g2, const float* g3, int M, const float* coef,
                    const float* gout, float* dgs, float* d0, float* d1, float* d2, float* d3, long n, void* stream);
int vx_gram_fwd(const float* x, float* G, int B, int C, long V, void* stream);
int vx_gram_bwd(const float* x, const float* dG, float* dx, int B, int C, long V, void* stream);
int vx_upsample_trilinear_fwd(const float* x, float* out, long BC, int d, int h, int w, int D, int H, int W, void* stream);
/* adjoint as three separable 1-D passes (D, then H, then W); ws = BC*d*(H*W + h*W) floats */
int vx_upsample_trilinear_bwd(const float* dout, float* dx, float* ws, long BC, int d, int h, int w, int D, int H, int W, void* stream);

/* ConvTranspose3d(k=2, s=2) specialised (conv_blocks.py:29-35): w = (Ci, Co, 2,2,2); x: (B,Ci,d,h,w); y: (B,Co,2d,2h,2w) */
int vx_upconv_set_mfma4(int on);      /* 1 (default; VELOXSEG_UPCONV_MFMA4): 16 x 64 MFMA tiles with 16-byte accesses on the coarse levels when the row length is a multiple of 4; 0: the 16 x 16 tiles (A/B, tests) */
int vx_upconv_k2s2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int d, int h, int wd, void* stream);
int vx_upconv_k2s2_bwd_data(const float* dy, const float* w, float* dx, int B, int Ci, int Co, int d, int h, int wd, void* stream);
/* weight gradient of the same layer as one MFMA GEMM (M = Ci, N = Co * 8 taps, K = B * d * h * wd): dw (Ci, Co, 2, 2, 2) +=.  Ci, Co multiples of 16, Ci <= 128 */
int vx_upconv_k2s2_wgrad_ok(int Ci, int Co);
int vx_upconv_k2s2_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int Co, int d, int h, int wd, void* stream);

/* fused AdamW on flat buffers (torch.optim.AdamW maths; config/train_config_bs4.json:66-72); g is scaled by grad_scale first */
int vx_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, long step, float grad_scale, void* stream);

/* ---- sliding-window inference + label metrics -------------------------------------------------------
 * Replaces monai.inferers.sliding_window_inference as called from utils/inference_runtime.py:4-19 (constant blending) and the
 * argmax + Dice of utils/inference_brats.py:216-217.  Volumes are one batch item, (C, D, H, W) fp32 contiguous. */
int vx_sw_extract(const float* vol, float* win, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0, void* stream);
/* acc[:, window] += weight * win; windows must be accumulated in the reference's window order (one call each) */
int vx_sw_accumulate(const float* win, float* acc, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0, float weight, void* stream);
/* out = acc / (cz[z]*cy[y]*cx[x]) (per-axis window counts; out may alias acc or be NULL), labels = uint8 argmax over C (may be NULL) */
int vx_sw_finalize(const float* acc, float* out, unsigned char* labels, const float* cz, const float* cy, const float* cx,
                   int C, int D, int H, int W, void* stream);
/* labels[b, v] = argmax_c logits[b, c, v] (first maximum, uint8): `outputs[0].argmax(dim=1)` of utils/metric/metrics.py:16 */
int vx_argmax_channels(const float* logits, unsigned char* labels, int B, int C, long V, void* stream);
/* conf[b, g, p] += #voxels with ground truth g and prediction p (uint64, NC*NC per sample; caller zeroes it): the sufficient statistic of
 * metrics_tensor (utils/metric/metrics.py:44-91) and cal_dice (utils/metric/metrics_brats.py:31-35); label widths 1, 4 or 8 bytes */
int vx_confusion(const void* pred, int pred_bytes, const void* gt, int gt_bytes, unsigned long long* conf, int B, long V, int NC, void* stream);

/* ---- training-input stand-in: MONAI CropForegroundd / RandCropByPosNegLabeld / RandRotated of utils/train_autopet.py:132-152 ----
 * x: one sample (C, D, H, W) fp32; labels uint8 / int32 / int64 / fp32 (lab_bytes 1 / 4 / 8 / -4). */
int vx_min_value(const float* x, long n, float* out_init_inf, void* stream);                       /* out[0] = min(out[0], min x); caller presets +inf */
int vx_bbox_gt(const float* x, float thr, int C, int D, int H, int W, int* out6, void* stream);     /* (min d,h,w, max d,h,w) of {x > thr}; caller presets (INT_MAX x3, -1 x3) */
int vx_label_chunk_count(const void* labels, int lab_bytes, long n, int chunk, int fg, int* counts, void* stream);   /* per-chunk counts of label > 0 (fg) or == 0 */
int vx_label_kth_in_chunk(const void* labels, int lab_bytes, long n, long chunk_start, int chunk, int fg, int k, long* out_index, void* stream);
/* rotation about the last spatial axis (the (D,H) plane turns), centre (n-1)/2, border padding; mode 0 bilinear, 1 nearest */
int vx_rotate_z(const float* x, float* out, int C, int D, int H, int W, float cos_a, float sin_a, int mode, void* stream);

/* ---- dense strided convolution on MFMA (groups = 1; k odd <= 7, stride 2..4, pad = k / 2): the DownConv layers between the encoder levels and the
 * stem (Encoder.py:29-58, conv_blocks.py:8-27 of the reference).  Implicit GEMM on v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate.  ws =
 * vx_conv_mfma_ws_floats(Cin, Cout, K, backward) floats for the operand-order weight image (written by every call).  bwd_data: accumulate = 1 adds
 * into dx.  vx_conv_mfma_ok: 1 when the shape is covered (D, H, W multiples of the stride, Cout % 4 == 0). */
int vx_conv_mfma_ok(int Cin, int Cout, int D, int H, int W, int K, int S, int P, int G, int ps);
int vx_conv_mfma_set_stem_f16(int on);   /* opt-in (default 0, or VELOXSEG_STEM_F16): the stem (k = 7, s = 4, p = 3, 16 output channels, W % 64 == 0, H % 16 == 0; 1: <= 2 input channels, 2: also 4) as a Toeplitz GEMM on the f16 pipe with LDS-staged input rows */
int vx_conv_mfma_set_stem_pieces(int np);      /* 2 (default, the fp32 mode): fp32-accurate products from two scaled fp16 pieces; 1 (the bf16 mode): plain fp16 operands, one MFMA per step, half the LDS -- also selects the f16-pipe stem for 4 input channels */
int vx_conv_mfma_stem_pieces(void);
int vx_conv_mfma_ws_floats(int Cin, int Cout, int K, int backward);
int vx_conv_mfma_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int B, int Cin, int D, int H, int W, int Cout, int K, int S, int P,
                     void* stream);
/* (round 6) x_absmax (one unsigned on the device, may be null): where vx_conv_mfma_fwd_writes_absmax(...) == 1 (the f16-pipe stem kernel takes the layer) the bits of max |x|
 * are left there as a by-product of the staging, for vx_down_wgrad_mfma_mx */
int vx_conv_mfma_fwd_writes_absmax(int Cin, int Cout, int D, int H, int W, int K, int S, int P);
int vx_conv_mfma_fwd_mx(const float* x, const float* w, const float* bias, float* y, float* ws, unsigned* x_absmax, int B, int Cin, int D, int H, int W, int Cout,
                        int K, int S, int P, void* stream);
int vx_conv_mfma_bwd_data(const float* dy, const float* w, float* dx, float* ws, int B, int Cin, int D, int H, int W, int Cout, int K, int S, int P, int accumulate,
                          void* stream);

/* ---- patch-expand (Decoder.py:73-76,150-153) with fp32-ACCURATE products on the bf16 matrix pipe: every operand split into ns bf16 pieces (2: 3 products per pair,
 * ~1e-5 relative; 3: 6 products, the fp32 product), fp32 accumulate, fp32 storage.  ns = 22 (forward and input gradient): two fp16 pieces (22 mantissa bits, 3
 * products) of the operand scaled by a power of two chosen per staged tile (weights: per tensor), the fp32 accumulators rescaled exactly -- measured closer to fp64
 * than ns = 3 (4e-7 vs 1.4e-6 of the maximum).  Same contracts as vx_expand_fwd_mfma / vx_expand_bwd_data_mfma; wt_ws holds vx_expand_split_ws_floats(Cc, ns)
 * floats; returns 1 (nothing launched) when the shape is not covered. */
int vx_expand_split_ws_floats(int Cc, int ns);
int vx_expand_fwd_mfma_split(const float* x, const float* w, const float* bias, float* wt_ws, float* y, int B, int Cc, int D, int H, int W, int ns, void* stream);
int vx_expand_bwd_data_mfma_split(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, int ns, void* stream);
/* (round 6) with the weight tensor's scale word handed over from the forward of the same layer (ns = 22: it sits vx_expand_split_ew_offset(Cc) floats into the forward's
 * workspace, 2 floats; NULL = find it here): no memset / max launch in front of the backward's weight image */
int vx_expand_split_ew_offset(int Cc);
/* (round 6) both weight images of a layer built ahead of its forward (fp16-piece mode, ns = 22; workspaces of vx_expand_split_ws_floats(Cc, 22) floats each), and the
 * matrix kernels alone on prepared images (ew_fwd = wt_fwd + vx_expand_split_ew_offset(Cc)) */
int vx_expand_prep_split22(const float* w, float* wt_fwd, float* wt_bwd, int Cc, void* stream);
int vx_expand_fwd_mfma_split_prepared(const float* x, const float* bias, const float* wt_fwd, float* y, int B, int Cc, int D, int H, int W, void* stream);
int vx_expand_bwd_data_mfma_split_prepared(const float* dy_fine, const float* wt_bwd, const float* ew_fwd, float* dx, int B, int Cc, int D, int H, int W, int accumulate, void* stream);
int vx_expand_bwd_data_mfma_split_ew(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, int ns,
                                     const float* ew_fwd, void* stream);

/* ---- fused per-voxel chains of a PWA transformer block, every modality of the block in one launch (csrc/pwa_fused.hip) ------------------------------
 * "pre": xn = LN_channels(x) (attention_utils.py:29-43, eps as given) followed by NS <= 3 1x1 projections out_s = W_s xn + b_s (PWA.py:291-298: q, k, v).
 * With s2d = 1 the input is gathered 8-way strided from a (B, C/8, 2gd, 2gh, 2gw) tensor first, i.e. PatchMerging (attention_utils.py:127-168): LN(8C) and the
 * 8C -> 2C reduction.  Every pointer array is a HOST array (read at the call): M modalities x a fixed number of entries, listed at each entry.  Shapes:
 * x (B, C, V), W_s (J_s, C), out_s (B, J_s, V), xn (B, C, V) or NULL.  C and J_s multiples of 16.  vx_ln_pw_ok: 1 when the shape is covered.
 * Backward: dx = dres + LN'(sum_s W_s^T dout_s); the LayerNorm-parameter sums leave as per-block partial rows part[vx_ln_pw_tiles(B, V)][2C] (dgamma | dbeta)
 * that vx_pw_wgrad_group folds; the weight gradients dW_s = dout_s xn^T are jobs of the same grouped launch. */
int vx_ln_pw_ok(int C, int NS, const int* J, long V, int s2d);
int vx_ln_pw_tiles(int B, long V);
/* per modality 13 entries: x, gamma, beta, w0, b0, w1, b1, w2, b2, xn, out0, out1, out2 */
int vx_ln_pw_fwd(const void* const* ptrs, int M, int NS, const int* J, int B, int C, long V, float eps, int s2d, int gd, int gh, int gw, void* stream);
/* per modality 11 entries: x, gamma, w0, w1, w2, dout0, dout1, dout2, dres (NULL ok; not with s2d), dx, part */
int vx_ln_pw_bwd(const void* const* ptrs, int M, int NS, const int* J, int B, int C, long V, float eps, int s2d, int gd, int gh, int gw, void* stream);
/* "post": y = alpha x + Drop_mix(Wm s + bm) (PWA.py:377,436); out = y + Drop_2(W2 Drop_1(GELU(W1 LN(y) + b1)) + b2) (PWA.py:437, attention_utils.py:45-71).
 * s (B, Cv, V), x / y / out (B, C, V), Wm (C, Cv), W1 (R, C), W2 (C, R).  Dropout masks = those of vx_axpy_drop_* / vx_gelu_drop_* for the same sites.
 * per modality 24 entries: s, x, wm, bm, gamma, beta, w1, b1, w2, b2, y, out, dout, ds, dxres, part, sc_n, sc_h, sc_da, sc_dz, sc_dmix, site_mix, site1, site2
 * (sites: integers cast to pointers).  Backward writes ds = Wm^T dmix, dxres = alpha dy, the partial rows part[vx_pwa_post_tiles(B, V)][2C] and the operands
 * of the weight-gradient jobs: sc_n = LN(y) (C), sc_h (R), sc_da (R), sc_dz (C), sc_dmix (C) -- dW1 = da n^T, dW2 = dz h^T, dWm = dmix s^T. */
int vx_pwa_post_ok(int C, int Cv, int R, long V);
int vx_pwa_post_tiles(int B, long V);
int vx_pwa_post_fwd(const void* const* ptrs, int M, int B, int C, int Cv, int R, long V, float eps, float alpha, const void* seed_ptr, float p_mix, float p_ffn, void* stream);
int vx_pwa_post_bwd(const void* const* ptrs, int M, int B, int C, int Cv, int R, long V, float eps, float alpha, const void* seed_ptr, float p_mix, float p_ffn, void* stream);

/* Channel stage of the JLC block at C = 64 / 128 (reference conv_blocks.py:60-66,74): out = o + Drop(W2 GELU(W1 IN(o) + b1) + b2), IN statistics folded
   from the producer's partial sums `part` ([B*C][nparts][2] doubles; NULL = read `stats` (B*C, 2) mean / rstd).  Same contract as vx_mlp_fwd / vx_mlp_bwd
   with norm = 0; the backward leaves the weight-gradient operands in scratch (dW2 = dz h^T, dW1 = da nhat^T: vx_pw_wgrad_group) and writes
   vx_inmlp_tiles(V) partial-sum rows (sum dn, sum dn nhat) per (b, c). */
int vx_inmlp_ok(int C, int R, long V);
int vx_inmlp_tiles(long V);
int vx_inmlp_fwd(const float* o, const double* part, int nparts, float* stats, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                 int B, int C, int R, long V, float eps, const void* seed_ptr, unsigned long long site, float p_drop, void* stream);
int vx_inmlp_bwd(const float* o, const float* stats, const float* w1, const float* b1, const float* w2, const float* dout, float* dn, float* part_dn,
                 float* sc_n, float* sc_h, float* sc_da, float* sc_dz, int B, int C, int R, long V, const void* seed_ptr, unsigned long long site, float p_drop,
                 void* stream);
/* up to 24 weight gradients of 1x1 convs (dW += dy x^T, db += sum dy) and up to 16 folds of partial rows in one launch.
 * jobs: ptrs 4 per job (x (B,Cin,V), dy (B,Cout,V), dw, db or NULL), dims 4 per job (Cin, Cout, V, B); folds: fptrs 3 per fold (part (rows, 2C), dgamma, dbeta),
 * fdims 2 per fold (C, rows) */
int vx_pw_wgrad_group(const void* const* ptrs, const long* dims, int nj, const void* const* fptrs, const int* fdims, int nf, void* stream);

/* ---- launch tape: a captured training stage replayed as plain launches on several HIP streams -----------------------------------------
 * The reference trains through eager PyTorch (utils/train_autopet.py:233-262: model(), loss, backward(), optimizer.step()); here one captured
 * pass of a stage (hipStreamBeginCapture ... EndCapture -> hipGraph_t, addresses from a private pool) is read back ONCE -- kernel and memset
 * nodes with their launch parameters (memcpy nodes, which ROCm 7.2 cannot read back, are kept as one-node graphs), and the dependency edges -- and replayed every step with hipLaunchKernel on `max_lanes`
 * streams with events only where an edge crosses streams (~3 us of host time per node; hipGraphLaunch on ROCm 7.2 needs ~17 us and
 * serialises the branches).  The caller keeps the hipGraph_t alive as long as the tape: kernel arguments are read from its nodes.
 * vx_tape_replay enqueues behind `stream` and joins every lane back into it before returning (it never blocks the host). */
typedef struct VxTape VxTape;
int vx_tape_build(void* hip_graph, int max_lanes, VxTape** out);
/* the same with a profile-guided layout: dur_us[i] = measured duration of node i of the tape vx_tape_build makes of this graph (vx_tape_profile); nodes are
   list-scheduled (longest remaining path first, earliest-start lane) instead of laid out greedily in capture order */
int vx_tape_build_pgo(void* graph, int max_lanes, const float* dur_us, int n, VxTape** out);
int vx_tape_info(const VxTape* tape, int* n_nodes, int* n_kernels, int* n_lanes, int* n_events);
/* (round 6) lane l of this tape runs on pool stream (l + k) % 4, k = 0 .. 3: a tape replayed beside another one (two window batches of a sliding-window inference in flight) takes
 * the other hardware queues for its main chains */
int vx_tape_set_lane_rotation(VxTape* tape, int k);
int vx_tape_replay(VxTape* tape, void* stream);
int vx_tape_free(VxTape* tape);
/* markers: vx_tape_mark(id, stream) inside the captured code; the tape records an event at that point of its schedule instead of launching anything,
 * vx_tape_wait_marker makes `stream` wait for it (after vx_tape_replay has been called for this step).  vx_tape_has_marker: 1 / 0. */
/* cross-lane dependency without an event (csrc/tape.hip): _set stores `value` to the device word `flag` once the stream reaches it, _wait holds the
 * stream until the word has reached `value` (wrap-around compare).  The tapes use them only when the four lane streams were measured to sit on
 * different hardware queues (vx_tape_lanes_distinct() == 4); otherwise, and with VELOXSEG_TAPE_FLAGS=0, the same dependencies are events. */
int vx_tape_flag_set(void* flag, int value, void* stream);
int vx_tape_flag_wait(const void* flag, int value, void* stream);
int vx_tape_set_flags(int on);   /* A/B: cross-lane dependencies inside a tape through flag kernels (1, default; VELOXSEG_TAPE_FLAGS=0 in the environment turns it off) or events (0) */
/* a poll that waited longer than this gives up WITHOUT trapping: it bumps a host-visible counter and lets its stream continue.  Default 5000 ms
 * (VELOXSEG_TAPE_FLAG_TIMEOUT_MS); 0 = wait for ever.  vx_tape_flag_timeouts(): answer, not a status -- polls that gave up since the last call
 * (clears the count); vx_tape_replay / vx_tape_hop return -3 with a message in vx_last_error() when the count is non-zero at their entry. */
int vx_tape_set_flag_timeout_ms(int ms);
int vx_tape_flag_timeouts(void);
/* `dst` waits for everything enqueued on `src` so far (event record + wait, or -- flags on -- a set kernel on src and a poll kernel on dst);
 * slot 0..255 names the call site, which must always pass the same src stream */
int vx_tape_hop(int slot, void* src, void* dst);
int vx_tape_mark(int id, void* stream);
int vx_tape_wait_marker(VxTape* tape, int id, void* stream);
int vx_tape_has_marker(const VxTape* tape, int id);
/* introspection: stand-alone time of every node (launch order, microseconds, minimum over reps; the nodes run one at a time, so the values
 * computed are those of a replay), and the layout of the tape: lane, workgroups, up to 4 cross-lane waits (-1 padded) and the kernel name per node */
int vx_tape_profile(VxTape* tape, void* stream, int reps, float* us);
int vx_tape_describe(const VxTape* tape, int* lane, int* grid, int* waits4, char* names, int name_stride);
/* schedule audit (veloxseg_amd/tape_audit.py; no counterpart in the reference, whose eager step is ordered by one stream, utils/train_brats2021.py:235-239):
 * vx_tape_waits: every cross-lane wait of every node (-1 padded to `stride`; an error if a node has more);  vx_tape_launch_node: node i alone on `stream` --
 * the audit launches a whole step node by node on ONE stream in random orders that respect lane order + waits, and every such order must give the same result;
 * vx_tape_set_fuzz: multi-lane replays put a 0 .. max_us spin kernel in front of a node with probability `prob` (seeded; max_us = 0 turns it off). */
int vx_tape_waits(const VxTape* tape, int* waits, int stride);
int vx_tape_launch_node(VxTape* tape, int node, void* stream);
int vx_tape_set_fuzz(int seed, float max_us, float prob);
/* raw bytes of launch parameter k of node i (at most `cap`, at most the host allocation that holds it; *got = bytes copied; memset nodes: k = 0 -> {dst, bytes});
 * vx_tape_node_kind: answer -- 0 kernel, 1 memset, 2 copy, 3 marker */
int vx_tape_node_param(const VxTape* tape, int node, int k, int nparams, int size, void* out, int cap, int* got);
int vx_tape_node_kind(const VxTape* tape, int node);
int vx_tape_replay_prefix(VxTape* tape, void* stream, int k);   /* the first k nodes as vx_tape_replay runs them, the rest one after the other on `stream` behind the join */
/* the process-wide stream of lane `lane` (lane % 4).  The four lane streams are chosen at first use so that they sit on different hardware
 * queues (measured with a spinning kernel: streams that share one of ROCm's 4 hardware queues never overlap, and neither does the NULL stream
 * with anything).  A tape with one lane replays on the caller's stream; a tape with more replays on the lane streams, gated by and joined
 * back into the caller's stream. */
int vx_tape_lane_stream(void* any_stream, int lane, void** out);
int vx_spin_us(float us, void* stream);   /* diagnostic: one wave spins for `us` microseconds on `stream` (stand-in for a collective: tools/comm_standin_probe.py) */
int vx_tape_lanes_distinct(void);
int vx_tape_spin_us(void);                        /* answer: spin length (us) of the lane calibration: 60, or longer when one spin alone measured > 80 us (slow launches / event hops: a profiler) */
int vx_tape_lane_on_caller_queue(void);           /* answer: the lane whose stream shares the hardware queue of the stream the lanes were chosen from (placed on lane 2); 4 = none found, 5 = lanes not chosen yet */
int vx_tape_permute_lanes(const int* perm4);      /* lane k is served by the stream that served lane perm4[k] (a permutation of 0..3); diagnostics */      /* answer, not a status: how many of the 4 lane streams were measured to overlap pairwise (-1 before the first use) */

#ifdef __cplusplus
}
#endif
#endif
