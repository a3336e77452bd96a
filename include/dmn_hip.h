/*
 * include/dmn_hip.h -- C ABI of libdmn_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the DMN language-conditioned hot path of BCV-Uniandes/DMS
 * (reference: dmn_pytorch/models/dmn.py).  Every entry point is `extern "C"`, takes
 * plain device pointers + sizes, and replaces the reference interface cited above it.
 *
 * Conventions (all entry points)
 *   - All tensors are fp32, contiguous, row-major, resident in device memory and OWNED
 *     BY THE CALLER; the library never allocates, frees or retains them.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous on it;
 *     nothing inside synchronises the device.  The caller's current device is used.
 *   - Return value: 0 = ok; < 0 = rejected argument (nothing launched);
 *     > 0 = hipError_t from the launch.  dmn_last_error() gives the message of the last
 *     failure on the calling thread.  No C++ exception crosses the boundary.
 *   - Re-entrant, no global mutable state besides the thread-local error string (one exception:
 *     dmn_sru_fwd_scan_noalloc, whose per-GPU scratch array is shared by all its callers).
 */
#ifndef DMN_HIP_H
#define DMN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DMN_ABI_VERSION 3

/* activation ids of the SRU cell (upstream `use_tanh` / `use_relu` switches) */
#define DMN_ACT_IDENTITY 0
#define DMN_ACT_TANH     1
#define DMN_ACT_RELU     2

int dmn_abi_version(void);
const char *dmn_last_error(void);

/*
 * SRU forward scan.  Replaces the third-party `sru` package's `SRU_Compute.forward`
 * -> cupy/pynvrtc kernel `sru_fwd`, reached from dmn.py:104,120 (construction) and
 * dmn.py:156,204 (`h, c = rnn(x)`).
 *   U     (T, B, d, k)  k innermost, k in {3,4}: U = x . W done by the caller (GEMM)
 *   x     (T, B, d)     highway input when k == 3; must be NULL when k == 4
 *   bias  (2*d)         [forget | reset]
 *   c0    (B, d)        or NULL (= zeros, what dmn.py does)
 *   h     (T - h_t0, B, d)  hidden states of steps t >= h_t0 (h_t0 = 0: all steps).
 *                       dmn.py:209 keeps only the last H*W steps of the mSRU, so the
 *                       last layer is called with h_t0 = T - H*W.
 *   c_all (T, B, d)     or NULL (inference: per-step cell states are not stored)
 *   c_last(B, d)        or NULL
 */
int dmn_sru_fwd(const float *U, const float *x, const float *bias, const float *c0,
                float *h, float *c_all, float *c_last,
                int T, int B, int d, int k, int act, float scale_x, int h_t0,
                void *stream);

/*
 * Same operation and reference interface as dmn_sru_fwd (dmn.py:156,204), time-parallel:
 * for FEW columns (B*d small, e.g. batch-1 serving: 1005 columns = 16 wavefronts) the
 * sequence is cut into `nchunks` chunks that run concurrently -- the cell update is the
 * affine map c -> f*c + (1-f)*u0, so chunk aggregates (A, Bv) compose associatively.
 * Inference only (no c_all).  `workspace` holds dmn_sru_scan_workspace_floats(...) floats
 * (2 * nchunks * B * d), caller-owned, 8-byte aligned.
 */
int dmn_sru_scan_workspace_floats(int T, int B, int d, int nchunks);
int dmn_sru_fwd_scan(const float *U, const float *x, const float *bias, const float *c0,
                     float *h, float *c_last, float *workspace,
                     int T, int B, int d, int k, int act, float scale_x, int h_t0,
                     int nchunks, void *stream);

/*
 * The scan of the LAST mSRU layer's tail with its hidden states stored as the image the reference
 * builds from them: dmn.py:209-212 `output[-(H*W):]` -> `permute(1, 2, 0)` -> `view(B, Hm, H, W)`.
 * Same inputs and arithmetic as dmn_sru_fwd over T = H*W steps (the caller passes the tail's U / x and
 * the cell state after the steps before it as c0); instead of h (T,B,d) it writes
 *     out[b*out_bs + j*T + t] = h[t, b, j]
 * so that `out` IS the (B, d, H, W) map -- out_bs (in floats, >= d*T) lets it be the first d channels
 * of the head's cat([x, feat], 1) buffer (dmn.py:337).  The wave transposes 64 steps x 64 columns
 * through LDS and writes runs of 64 consecutive floats.  Bit-identical to dmn_sru_fwd + permute.
 *   nchunks > 1: time-parallel variant for few columns, `workspace` as for dmn_sru_fwd_scan;
 *   nchunks == 1: one pass, workspace may be NULL.
 */
int dmn_sru_fwd_image(const float *U, const float *x, const float *bias, const float *c0,
                      float *out, float *c_last, float *workspace, int T, int B, int d, int k,
                      int act, float scale_x, long long out_bs, int nchunks, void *stream);

/*
 * dmn_sru_fwd_scan WITHOUT a caller workspace, for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  Same arguments, kernels and results (bit for bit) as
 * dmn_sru_fwd_scan; the chunk carries live in a 4 MiB file-scope `static __device__` array that
 * every GPU the library runs on holds until the process ends.  The entry point calls no hipMalloc /
 * hipFree / allocating routine.
 *   Limit: 2 * nchunks * B * d <= DMN_SCAN_SCRATCH_FLOATS, else status -11 and nothing is launched
 *   (e.g. batch-1 serving: 2 * 32 * 1005 = 64 320 floats; 16 samples x 32 chunks still fit).
 *   All calls on one GPU share that GPU's array: calls that OVERLAP on one GPU (two streams, two
 *   threads) race on it -- serialise them, or use dmn_sru_fwd_scan with separate workspaces.
 *   Nothing in the array has to start at zero (every carry is written before it is read).
 *   The FIRST call on a device resolves the array's device address (hipGetSymbolAddress, under a mutex):
 *   call once per device BEFORE a stream capture; every later call only launches kernels.
 */
#define DMN_SCAN_SCRATCH_FLOATS (1 << 20)
int dmn_sru_fwd_scan_noalloc(const float *U, const float *x, const float *bias, const float *c0,
                             float *h, float *c_last,
                             int T, int B, int d, int k, int act, float scale_x, int h_t0,
                             int nchunks, void *stream);

/*
 * State-only scan: the cell state after T steps, nothing else.  Same reference interface as
 * dmn_sru_fwd (the last mSRU layer under dmn.py:204): dmn.py:209 keeps only the last H*W
 * hidden states, and the steps before them influence the output through c alone, which needs
 * neither the reset gate nor the highway term.  The caller therefore projects those steps
 * onto 2 of the k weight columns only (a third less GEMM work) and passes
 *   U2   (T, B, d, 2)  [u0 | forget pre-activation] per column, 8-byte aligned
 *   bias (2*d) as in dmn_sru_fwd (only the forget half is read);  c0 (B, d) or NULL
 *   c_last (B, d): c after step T-1 -- bit-identical to dmn_sru_fwd's c_last on the same u0, u1.
 */
int dmn_sru_state_fwd(const float *U2, const float *bias, const float *c0, float *c_last,
                      int T, int B, int d, void *stream);
/*
 * The same under autograd (the training step: `SRU_Compute.forward/backward` of the last mSRU layer's
 * first T - H*W steps, dmn.py:204,209): the forward also keeps every step's cell state,
 *   c_all (T, B, d);
 * the backward takes dL/dc_last (grad_last (B, d) or NULL = 0) and returns
 *   grad_U2 (T, B, d, 2), grad_bias_f (d) = the forget half of the bias gradient (summed over time and
 *   batch; the reset half is zero for these steps), grad_c0 (B, d);
 *   bias_partials: B*d floats of caller scratch or NULL, as for dmn_sru_bwd.
 */
int dmn_sru_state_fwd_train(const float *U2, const float *bias, const float *c0, float *c_all,
                            float *c_last, int T, int B, int d, void *stream);
int dmn_sru_state_bwd(const float *U2, const float *bias, const float *c0, const float *c_all,
                      const float *grad_last, float *grad_U2, float *grad_bias_f, float *grad_c0,
                      float *bias_partials, int T, int B, int d, void *stream);
/* time-parallel variant for FEW columns (batch-1 serving), as dmn_sru_fwd_scan is to dmn_sru_fwd:
 * chunk aggregates (prod f, c | c_in = 0) composed in time order; `workspace` holds
 * dmn_sru_scan_workspace_floats(T, B, d, nchunks) floats, caller-owned, 8-byte aligned. */
int dmn_sru_state_fwd_scan(const float *U2, const float *bias, const float *c0, float *c_last,
                           float *workspace, int T, int B, int d, int nchunks, void *stream);

/*
 * SRU backward scan (reverse time). Replaces `SRU_Compute.backward` -> `sru_bwd`
 * of the same package (autograd of dmn.py:156,204).
 *   c_all (T,B,d) saved by the forward; grad_h (T,B,d); grad_last (B,d) or NULL.
 *   grad_U (T,B,d,k); grad_x (T,B,d) iff k == 3 else NULL;
 *   grad_bias (2*d) = [forget | reset], summed over time AND batch by the library:
 *     bias_partials != NULL (2*B*d floats of caller scratch): every column's sums over time go there and a
 *       second small launch adds them over the batch in sample order -- what the reference's Python does with
 *       the kernel's (2, B, d) partials -- so two launches on the same inputs give the same bits;
 *     bias_partials == NULL: the entry point zeroes grad_bias on `stream` and every lane adds its column's
 *       sums with one float atomic; the order of the B additions per entry is not fixed, so the last bit may
 *       differ between launches;
 *   grad_c0 (B, d).
 */
int dmn_sru_bwd(const float *U, const float *x, const float *bias, const float *c0,
                const float *c_all, const float *grad_h, const float *grad_last,
                float *grad_U, float *grad_x, float *grad_bias, float *grad_c0, float *bias_partials,
                int T, int B, int d, int k, int act, float scale_x, void *stream);

/*
 * Bias add + sigmoid of the adaptive filter bank, in place: dmn.py:173-174
 * `filters = self.adaptative_filter(linear_in)` + `F.sigmoid(filters)`.  The Linear's GEMM runs
 * bias-free in front of it (library GEMM, beta = 0); this one streaming pass (8 B per element)
 * replaces the bias broadcast, the GEMM's read of it and the separate sigmoid pass.
 *   x (rows, cols) row-major, contiguous: x[r, c] = 1 / (1 + exp(-(x[r, c] + bias[c]))).
 * Inference only (under autograd the torch Linear + sigmoid keep the graph).
 */
int dmn_bias_sigmoid(float *x, const float *bias, long long rows, int cols, void *stream);

/*
 * Coordinate channels + concat.  Replaces BaseDMN.generate_spatial_batch
 * (dmn.py:225-246: host double loop + H2D copy every forward) and
 * torch.cat([vis, spatial], 1) (dmn.py:142).
 *   vis (B, C, H, W) -> V (B, C + 8, H, W), V[:, C:C+8] = [xmin,ymin,xmax,ymax,xctr,yctr,1/W,1/H]
 */
int dmn_cat_coords(const float *vis, float *V, int B, int C, int H, int W, void *stream);

/*
 * Dynamic filters.  Replaces `for t: F.conv2d(vis, filters[t])` + torch.cat
 * (dmn.py:178-184) by one batched thin GEMM:
 *   P[t,b,k,s] = sum_c F[t,b,k,c] * V[b,c,s]
 *   V (B, C, S)   F (L, B, K, C)   P (L, B, K, S)
 */
int dmn_dynfilter_fwd(const float *V, const float *F, float *P,
                      int B, int L, int K, int C, int S, void *stream);

/*
 * The same with a caller-provided workspace of dmn_dynfilter_workspace_floats(B, L, K, C, S) floats
 * (16-byte aligned; NULL = dmn_dynfilter_fwd).  With few samples the contraction over C is spread over
 * one workgroup per (tile, chunk) and a reduction pass; the chunk order of the summation is the same
 * either way, so P is bit-identical to dmn_dynfilter_fwd's (dmn.py:178-184, batch-1 serving).
 */
int dmn_dynfilter_fwd_ws(const float *V, const float *F, float *P, float *workspace,
                         int B, int L, int K, int C, int S, void *stream);
int dmn_dynfilter_workspace_floats(int B, int L, int K, int C, int S);   /* 0 = none / too large */

/* gradients of the above: gF (L,B,K,C) = gP . V^T ; gV (B,C,S) = sum_{t,k} F * gP */
int dmn_dynfilter_bwd(const float *V, const float *F, const float *gP,
                      float *gV, float *gF,
                      int B, int L, int K, int C, int S, void *stream);

/*
 * Concat-free mix.  Replaces expand + torch.cat([vis, lang_mix, p], 2) (dmn.py:152-161,
 * 170,187-189), the 1x1 comb_conv (dmn.py:192) and the view/permute/contiguous into the
 * mSRU sequence order n = s*L + t (dmn.py:196-201).  With comb_conv.weight split by
 * channel block [vis+coords | WE | h | P] the caller supplies two small GEMM results:
 *   A  (B, S, M)  = V[b]^T . Wv^T            (vis + coordinate channels)
 *   Bt (L, B, M)  = [WE_t ; h_t] . Wl^T + bias
 *   P  (L, B, K, S), Wp (M, K) with row stride ldw (a column slice of comb_conv.weight)
 * and this kernel writes
 *   Q[(s*L + t), b, m] = A[b,s,m] + Bt[t,b,m] + sum_k Wp[m,k] * P[t,b,k,s]      (S*L, B, M)
 */
int dmn_mix_bcast_add(const float *A, const float *Bt, const float *P, const float *Wp,
                      int ldw, float *Q, int B, int L, int K, int S, int M, void *stream);

/*
 * dmn_mix_bcast_add fused into the mSRU layer-0 scan (inference).  Replaces the same reference lines
 * as dmn_mix_bcast_add (dmn.py:187-201) plus the first SRUCell's scan under dmn.py:204, for the case
 * where comb_conv has been folded into that cell's projection (M = 4*d, k = 4): the lane that scans
 * column (b, j) builds its four pre-activations
 *   U0[(s*L + t), b, j, 0..3] = A[b,s,4j..] + Bt[t,b,4j..] + sum_k Wp[4j.., k] * P[t,b,k,s]
 * in registers, so U0 (16 B per element) is never written or read.  Operation for operation the
 * arithmetic of dmn_mix_bcast_add followed by dmn_sru_fwd: h is bit-identical to that pair.
 *   A (B, S, 4d), Bt (L, B, 4d) 16-byte aligned; Pt (B, S, L, K) = P transposed by the caller (K <= 16);
 *   Wp (4d, K) row stride ldw; bias (2d); c0 (B, d) or NULL; h (S*L, B, d); c_last (B, d) or NULL.
 * Bound by one wave's instruction issue (~230 cycles per step), not by HBM: worth it from ~200
 * wavefronts of columns (B*d/64) on; below that the two-kernel path with its time-parallel scan wins.
 */
int dmn_mix_sru_fwd(const float *A, const float *Bt, const float *Pt, const float *Wp, int ldw,
                    const float *bias, const float *c0, float *h, float *c_last,
                    int B, int L, int K, int S, int d, int act, void *stream);

/*
 * gradients of dmn_mix_bcast_add w.r.t. A, Bt, P, Wp given gQ (S*L, B, M) -- autograd of dmn.py:187-201
 * in the training step (train.py:275-290):
 *   gA[b,s,m] = sum_t gQ ; gBt[t,b,m] = sum_s gQ ; gP[t,b,k,s] = sum_m Wp[m,k] gQ ;
 *   gWp_part (parts, M, K): partial sums of gWp[m,k] = sum_{t,b,s} gQ * P; the caller adds the `parts`
 *   slabs, parts = dmn_mix_bwd_parts(B, L, K, S, M).
 * Two implementations, chosen by shape (same choice in the two query functions):
 *   - ONE pass over gQ (K <= 10, L <= 20, M % 4 == 0 and B * ceil(M / 512) >= 128 workgroups): a workgroup
 *     owns 512 columns of one sample (256 when 512-column workgroups could not fill the chip) and walks all its
 *     pixels; parts = 4*B; `workspace` holds dmn_mix_bwd_workspace_floats() floats (one share of gP per 256
 *     columns -- enough for either workgroup shape -- added in block order: deterministic).
 *     gQ, gA, gBt, gWp_part 16-byte aligned.
 *   - two passes (pixel pass gA / gP, word pass gBt / gWp) otherwise: parts = L*B, workspace unused (may be
 *     NULL); K <= 16, M % 4 == 0; rows wider than 1024 floats need K <= 10 and L <= 24.
 */
int dmn_mix_bwd_parts(int B, int L, int K, int S, int M);
long long dmn_mix_bwd_workspace_floats(int B, int L, int K, int S, int M);
int dmn_mix_bcast_add_bwd(const float *gQ, const float *P, const float *Wp, int ldw,
                          float *gA, float *gBt, float *gP, float *gWp_part, float *workspace,
                          int B, int L, int K, int S, int M, void *stream);

/*
 * Upsampling glue.  Replaces torch.cat([x, features[i]], 1) followed by
 * nn.Upsample(scale_factor=2, mode='bilinear') (align_corners=False) in
 * UpsamplingModule.forward (dmn.py:324,337-338); the 3x3 convolution that follows
 * stays on MIOpen.
 *   x (B, Cx, H, W), feat (B, Cf, H, W) -> out (B, Cx + Cf, 2H, 2W)
 */
int dmn_cat_upsample2x(const float *x, const float *feat, float *out,
                       int B, int Cx, int Cf, int H, int W, void *stream);

/* gradient of the above: g (B, Cx+Cf, 2H, 2W) -> gx (B,Cx,H,W), gfeat (B,Cf,H,W) (either may be NULL) */
int dmn_cat_upsample2x_bwd(const float *g, float *gx, float *gfeat,
                           int B, int Cx, int Cf, int H, int W, void *stream);

/*
 * High-res head level at inference, Winograd F(m x m, 3x3), m = 2 or 4, split so that the
 * multiply part of the 3x3 convolution is a plain batched GEMM (rocBLAS fp32: ~300 (m=2) /
 * ~530 (m=4) TFLOP/s conv-equivalent vs MIOpen's 111-122).  Replaces, for one level of
 * UpsamplingModule.forward (dmn.py:324,337-338),
 *   conv3x3(Upsample(x2, bilinear)(cat([x, feat], 1)))  ->  transform, (m+2)^2 GEMMs, transform:
 * dmn_wino_cat_upsample2x_transform: cat + upsample + input transform B^T d B in one pass; the
 *   upsampled tensor is never materialised (an up-res tile depends on a 3x3 (m=2) / 4x4 (m=4)
 *   low-res neighbourhood only).  n = m + 2, tiles NT = B * (2H/m) * (2W/m), row-major (b, ty, tx).
 *   x (B, Cx, H, W), feat (B, Cf, H, W) -> V (n*n, Cx+Cf, NT);  m = 4 needs even H and W.
 * caller: M[xi] = U[xi] . V[xi] with U (n*n, Cout, Cx+Cf) = G g G^T of the conv weight
 * dmn_wino_output_transform: out (B, Cout, 2H, 2W) = A^T M A + bias[o] (bias may be NULL)
 *   M (n*n, Cout, NT)
 * fp32 error of the whole level vs direct convolution: ~7e-7 (m=2), ~1e-5 (m=4) relative.
 */
int dmn_wino_cat_upsample2x_transform(const float *x, const float *feat, float *V,
                                      int B, int Cx, int Cf, int H, int W, int m, void *stream);
int dmn_wino_output_transform(const float *M, const float *bias, float *out,
                              int B, int Cout, int H, int W, int m, void *stream);

/*
 * Last head level and out_layer in one pass.  UpsamplingModule's last 3x3 conv (dmn.py:324) is
 * followed directly by the 1x1 out_layer (dmn.py:327,341; no non-linearity between them by
 * default), so their composition is a 3x3 conv to ONE channel; with cat + bilinear x2 fused in:
 *   out (B, 1, 2H, 2W) = conv3x3_pad1( Upsample(x2)(cat([x, feat], 1)), weff ) + bias0[0]
 *   x (B, Cx, H, W), feat (B, Cf, H, W); weff (Cx+Cf, 3, 3) = sum_o w_out[o] * w[o, :, :, :],
 *   bias0 (1) = b_out + sum_o w_out[o] * b[o], both computed by the caller (tiny); bias0 may be NULL.
 */
int dmn_head_final(const float *x, const float *feat, const float *weff, const float *bias0,
                   float *out, int B, int Cx, int Cf, int H, int W, void *stream);

/*
 * High-res head level at inference as "project at low resolution, gather at high resolution".
 * Replaces, for one level of UpsamplingModule.forward (dmn.py:324,337-338),
 *   conv3x3_pad1(Upsample(x2, bilinear)(cat([x, feat], 1))) + bias.
 * Bilinear upsampling (per channel) and a conv tap's channel mixing (per pixel) commute, so the
 * caller applies the nine tap matrices at LOW resolution with a plain library GEMM,
 *   Z[b] (9*Cout, H*W) = Wt (9*Cout, Cx+Cf) . cat(x, feat)[b],  Wt[(ky*3+kx)*Cout + o, c] = w[o, c, ky, kx]
 * (split over K into an x and a feat GEMM, so cat() is never built; 2.25 MACs per output and
 * input channel like Winograd F(4x4), but exact fp32 products), and
 * dmn_tap_upsample_gather does the rest in one HBM-bound pass:
 *   out[b, o, y, x] = bias[o] + sum_{ky,kx} up2(Z[b, ky*3+kx, o])[y + ky - 1, x + kx - 1]
 *   with up2 = bilinear x2 (align_corners=False) and zero outside the (2H, 2W) image.
 *   Z (B, 9, Cout, H, W) -> out (B, Cout, 2H, 2W); bias (Cout) or NULL.
 *   out_bs: floats between consecutive samples of out (Cout*4*H*W when out is compact; larger when
 *   out is the first Cout channels of the NEXT level's [x | feat] buffer, so that level's GEMM reads
 *   one operand with K = Cx + Cf and nobody copies).
 * dmn_tap_project: the projection itself for the 9-row case Cout == 1 (last level composed with
 * out_layer, see dmn_head_final), where a GEMM with 9 rows is a poor fit for a BLAS library:
 *   Z (B, 9, HW) = w (9, Cx+Cf) . cat(x, feat)[b];  x (B, Cx, HW), feat (B, Cf, HW) (either may be empty).
 */
int dmn_tap_upsample_gather(const float *Z, const float *bias, float *out,
                            int B, int Cout, int H, int W, long long out_bs, void *stream);
int dmn_tap_project(const float *x, const float *feat, const float *w, float *Z,
                    int B, int Cx, int Cf, int HW, void *stream);

/*
 * 1x1 convolution to ONE output channel as a channel reduction: replaces
 * `UpsamplingModule.out_layer` (dmn.py:327,341) and `output_collapse` (dmn.py:126-129,213-214):
 *   out[n, p] = sum_c w[c] * x[n, c, p] + bias0[0]      x (N, C, HW), w (C), bias0 (1) or NULL
 */
int dmn_channel_dot(const float *x, const float *w, const float *bias0, float *out,
                    int N, int C, int HW, void *stream);

/*
 * Evaluation threshold sweep.  Replaces, per sample, the loop
 * `for thresh in score_thresh: compute_mask_IU((out > thresh).float(), mask)`
 * (dmn_pytorch/train.py:333-338 and 366-374) by one pass over the pixels.
 *   prob (B, N) probabilities (sigmoid + upsample done by the caller, train.py:361-364)
 *   mask (B, N) target (0/1);  thresholds (NT) ASCENDING, NT <= 64 (device memory)
 *   inter, uni (B, NT): ACCUMULATED INTO (zero them first):
 *     inter[b,j] += sum((prob > th_j) * mask) ; uni[b,j] += sum((prob > th_j) + mask - (prob > th_j) * mask)
 */
int dmn_threshold_iou(const float *prob, const float *mask, const float *thresholds,
                      float *inter, float *uni, int B, int N, int NT, void *stream);

/*
 * Inference-time glue of the DPN backbone (the convolutions stay on MIOpen).
 * dmn_bn_relu: eval-mode BatchNorm2d followed by ReLU, the `act(bn(x))` of CatBnAct /
 * BnActConv2d (dmn_pytorch/models/dpn/dpn_model.py:156-177):
 *   y[n,c,:] = max((x[n,c,:] - mean[c]) * scale[c] + bias[c], 0), scale = weight / sqrt(var + eps)
 * dmn_dpn_merge: DualPathBlock's merge `resid = x_s1 + out1; dense = cat([x_s2, out2])`
 * (dpn_model.py:265-269) together with the next block's `cat(resid, dense)` (dpn_model.py:246):
 *   base (N, Cb, HW) = [x_s1 | x_s2], yv (N, Cy, HW) = [out1 | out2], c1c = width of the residual path
 *   out (N, Cb + Cy - c1c, HW) = [ base[:, :c1c] + yv[:, :c1c] | base[:, c1c:] | yv[:, c1c:] ]
 *   act (same shape) or NULL: relu(bn(out)) with the (mean, scale, bias) of the NEXT consumer's
 *   BatchNorm (the following block's c1x1_a.bn, or conv5_bn_ac), written in the same pass.
 */
/*
 * dmn_gconv3x3_bn_relu: the grouped 3x3 / pad-1 convolution (stride 1 or 2) of a DualPathBlock with
 * the BatchNorm+ReLU before it and the one after it fused in (dpn_model.py:246-263:
 * `c3x3_b` = BnActConv2d, followed by `c1x1_c`'s bn + act), inference:
 *   y = relu(bn2( conv3x3_grouped( relu(bn1(x)) ) ))
 *   x (N, C, H, W), y (N, C, H/stride, W/stride) 16-byte aligned (stride 2: H % 2 == 0, W % 8 == 0); C/groups = cg in {3,4,6,8,12,16,24,32}; no bias
 *   w (groups, cg_in, 9, cg_out): the reference's (C, cg, 3, 3) weight permuted once per weight
 *   version by the caller, so that the output channels of one (ci, tap) are adjacent
 *   (mean1, scale1, bias1) / (mean2, scale2, bias2): (C) each as in dmn_bn_relu, or all NULL
 *   to skip that stage.
 */
int dmn_gconv3x3_bn_relu(const float *x, const float *w,
                         const float *mean1, const float *scale1, const float *bias1,
                         const float *mean2, const float *scale2, const float *bias2,
                         float *y, int N, int C, int H, int W, int groups, int stride, void *stream);
int dmn_bn_relu(const float *x, const float *mean, const float *scale, const float *bias,
                float *y, int N, int C, int HW, void *stream);
/* dmn_bn_relu_s2: y (N, C, H/2, W/2) = relu(bn(x))[:, :, ::2, ::2] -- the pre-activation of the
 * stride-2 1x1 projection `c1x1_w_s2` (dpn_model.py:208-213), which reads a quarter of x only.
 * H % 2 == 0, W % 4 == 0.  x_bs: floats between consecutive samples of x (C*H*W when compact). */
int dmn_bn_relu_s2(const float *x, const float *mean, const float *scale, const float *bias,
                   float *y, int N, int C, int H, int W, long long x_bs, void *stream);
/* dmn_bn_relu_maxpool: y (N, C, (H-1)/2+1, W/2) = MaxPool2d(3, stride 2, padding 1)(relu(bn(x))) -- the
 * stem's bn + act + pool (InputBlock, dpn_model.py:186-197) in one pass over the raw conv output
 * (which stays in memory as pyramid level 0).  W % 4 == 0. */
int dmn_bn_relu_maxpool(const float *x, const float *mean, const float *scale, const float *bias,
                        float *y, int N, int C, int H, int W, void *stream);
int dmn_dpn_merge(const float *base, const float *yv, float *out, float *act,
                  const float *mean, const float *scale, const float *bias,
                  int N, int Cb, int Cy, int c1c, int HW, void *stream);

/*
 * dmn_dpn_stage_pass: dmn_dpn_merge for a stage whose blocks share ONE buffer per sample
 *   buf (N, rows, HW), rows >= Co + inc:  [ dense0 (2*inc) | res (c1c) | slot (inc) | o2_1 | o2_2 | ... ]
 * The caller runs the block's last 1x1 convolution (dpn_model.py:262-264) as a library GEMM with
 * beta = 1 onto rows [res | slot] (slot zeroed), so `resid = x_s1 + out1` (dpn_model.py:265-266) is the
 * GEMM epilogue.  This pass writes act (N, Co, HW) = relu(bn(X)) with X in the reference's channel
 * order [res | dense0 | o2_1 .. | slot] (dpn_model.py:246,267-269) for the (mean, scale, bias) of the
 * next consumer, moves the slot's channels to their final rows [Co .. Co + inc), zeroes
 * the slot, and writes X itself to out (N, Co, HW) when out != NULL (pyramid taps); out_bs = floats
 * between consecutive samples of out (Co*HW when compact; larger when the tap is written straight
 * into the head's [x | feat] buffer).
 */
int dmn_dpn_stage_pass(float *buf, float *act, float *out,
                       const float *mean, const float *scale, const float *bias,
                       int N, int rows, int c1c, int inc, int Co, int HW, long long out_bs, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DMN_HIP_H */
