// kernels.hpp — argument blocks and host launchers shared by the translation units of libfdipt_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short half_t;  // raw bits of the half-precision operand type (common.hpp)

// development switch: product builds compile it to `false` and never read the environment
#ifdef FDIPT_DEV
#include <stdlib.h>
#define FD_DEV_ENV(name) (getenv(name) != nullptr)
#else
#define FD_DEV_ENV(name) false
#endif

struct EdgeTransArgs {
  int B, N;
  const void* z_in;      // [B,N,N,CZ] ZT
  void* z_out;           // [B,N,N,CZ] ZT (may alias z_in)
  const float* e;        // [B,N,CB] f32 initial_embed(node)
  const void *w1, *w2, *wf;  // [H,H],[H,H],[CZ,H] operand precision, row-major (out,in)
  const float *b1, *b2, *bf, *gamma, *beta;
  const float* res_mask;  // [B,N]
  float* trace;           // optional [B,N,N,CZ] f32
  unsigned long long* clock = nullptr;  // optional shader-clock probe (FdiptForwardArgs.clock_out)
};

struct EdgeEmbedArgs {
  int B, N, n_rel, rel_off, num_bins;
  const float* pi;       // [B,N,CZ] f32: W1[:, :d1] pte_i + b1
  const float* pj;       // [B,N,CZ] f32: W1[:, d1:2d1] pte_j
  const float* rtab;     // [B,n_rel,CZ] f32: W1[:, 2d1:2d1+32] index_embedding(rel)
  const float* dtab;     // [num_bins+1,CZ] f32: W1 distogram columns (+ a zero row = "no bin")
  const float* edges;    // [num_bins] f32 lower bin edges
  const int32_t* seq_idx;  // [B,N]
  const float* sc_ca;    // [B,N,3]
  const void *w2, *w3;   // [CZ,CZ] operand precision
  const float *b2, *b3, *gamma, *beta;
  const float* res_mask;
  void* z_out;
  float* trace;
  // optional (bf16 kernel): pair bias of the FIRST block's attention from the LayerNorm epilogue, as ET2Args
  const void* wb_img;
  const float* bb;
  float* bias_out;
  int H;
  int reserve_cus = 0;  // persistent kernel: CUs left to launches of other streams (FdiptForwardArgs.reserve_cus)
  // optional (half-precision kernel with wb_img set; round 6): pair_z of the FIRST block's IPA, as ET2Args
  const void* wdz_img = nullptr;
  const void* wdz_img_lo = nullptr;
  const float* bdz = nullptr;
  half_t* pz_out = nullptr;
};

struct AttnArgs {
  int B, N, H;
  const float *q, *k, *v;  // row r=(b*N+i): q + r*q_ld + h*q_hs etc.
  long q_ld, k_ld, v_ld;
  int q_hs, k_hs, v_hs;
  int C, Dv;        // qk feature dim, value dim
  float scale;      // multiplies QK^T
  const float* bias;      // f32, already scaled: [B,N,N,H] (attention.hip) or [B,H,N,N] (attention2.hip); NULL if none
  const float* res_mask;  // [B,N]
  // IPA only
  const float *qp, *kp, *vp;  // [B,N,H,Pq,3], [B,N,H,Pq,3], [B,N,H,Pv,3] global-frame points (scaled units)
  int Pq, Pv;
  const float* gamma;  // [H] softplus(head_weights) * sqrt(1/(3*(Pq*9/2)))
  const float *rot, *trans;  // [B,N,9], [B,N,3] current frames (scaled units)
  float* probs;  // [B,H,N,N] f32 attention weights (consumed by opair_kernel)
  // output
  float* out;  // row r: out + r*out_ld ; o at h*Dv ; IPA point features at pt_off (see below)
  long out_ld;
  int pt_off;  // = H*C: x plane; y at +H*Pv; z at +2*H*Pv; norm at +3*H*Pv
  int lds_s;   // S row stride in floats
};

struct OPairArgs {
  int B, N, H, CZ, CD;   // CD = CZ/4
  const void* z;         // [B,N,N,CZ] ZT
  const float* probs;    // [B,H,N,N]
  const half_t* probs_h16;  // optional (MFMA kernel): the same weights as bf16 rows [B,N,H,probs_np], zero beyond N; else NULL
  int probs_np;
  const float* wdz;      // [CZ,CD] f32 (down_z weight, transposed)
  const void* wdz_img;   // optional: down_z weight [CD, CZ] as a bf16 fragment image (fd_chain_build_image, natural k) for the MFMA kernel
  // optional (MFMA kernel): image of Wdz - half(Wdz): the down-projection runs on split operands (az = hi + lo, Wdz = hi + lo).
  // It is a per-residue product [8 x 128] x [128 x 32]: its operand rounding is shared by all keys of the row (tests/err_budget.py)
  const void* wdz_img_lo = nullptr;
  const float* bdz;      // [CD]
  float* out;            // row (b*N+i): out + r*out_ld + off + h*CD + d
  half_t* out_h16;      // MFMA kernel: if set, bf16 rows (same out_ld, in elements) INSTEAD of out
  long out_ld;
  int off;
  L2Warm warm = {};  // weights of the kernel launched next (common.hpp: L2 warm-up hand-over)
  const half_t* pz = nullptr;  // round 6: pair_z image of this block (fd_pz_bytes) written by the producer of z -> opair_pz_kernel (z, wdz* unused)
};

struct PointsArgs {
  int B, N, H, Pq, Pv;
  const float* proj;  // [B*N, ld]: q_pts planes at q_off (3 x H*Pq), kv_pts planes at kv_off (3 x H*(Pq+Pv))
  long ld;
  int q_off, kv_off;
  const float* quat;   // [B,N,4] current quaternion
  const float* trans;  // [B,N,3] current translation (scaled)
  float *qp, *kp, *vp, *rot;
  // optional: v_pts as a bf16 hi/lo fragment image for the MFMA o_pt of attention3 (needs Pv == 12): [B*H][3 tiles][Np/16]
  // [64][8]; row 32 dt + (lane & 31) = coordinate (< 36: high part, 36..71: low part = v - bf16(v), else 0), key position
  // 16 s + 8 (lane >> 5) + e with the keys permuted inside every 16-group; pads must be zeroed by the caller once
  unsigned short* vpt;
  int Np;
  // optional (points16_kernel only; merged projection): the node rows as the attention's K / V_hi / V_lo images shared by all heads
  // (layouts: fd_node_images), written by the same launch — a block's 16 keys are one 16-group of those images
  const float* node = nullptr;
  int ld_node = 0;
  unsigned short *nKb = nullptr, *nVt = nullptr, *nVt_lo = nullptr;
  // optional (points16_kernel only; Pq == 8): the key points as attention3's point-logit A fragments (fd_kpf layout below), IEEE fp16
  // hi / lo parts; needs the head weights gamma [H] and res_mask [B,N]
  unsigned short* kpf = nullptr;
  const float* gamma = nullptr;
  const float* res_mask = nullptr;
};
// Key-point fragment image of attention3 (Attn3Args.kpf): [B*H][Np/32][4][64][8] fp16.  With k = the 24 global-frame coordinates of a
// key's 8 points (thirds k0 = [0,8), k1 = [8,16), k2 = [16,24)), h / l = fp16 hi / lo parts (k = h + l to 2^-22), lane = 32 hf + key % 32:
//   fragment 0: hf 0 -> h(k0), hf 1 -> h(k1)        fragment 1: hf 0 -> l(k0), hf 1 -> l(k1)
//   fragment 2: hf 0 -> h(k2), hf 1 -> l(k2)        fragment 3: hf 0 -> h(k2), hf 1 -> X
//   X = [2 m_j, m_j, pad ? -60000 : 0, a, b, c, 0, 0],  a + b + c = -gamma_h |k|^2 / 2 (three fp16 parts), m_j = res_mask, pad = key >= N
// The query side (registers of the attention kernel) pairs them so that six fp16 MFMAs give 1e5 (m_i m_j - 1) + gamma (q.k - |k|^2 / 2):
// the term -gamma |q|^2 / 2 of -gamma |q - k|^2 / 2 is constant along a softmax row and drops out.
#define FD_KPF_FRAGS 4

// Pair bias of the IPA attention, tiled for BOTH sides: [sample*head][query tile][key tile][query in tile][32 keys].
// The producers (pair_bias2_kernel, the EdgeTransition epilogue) hold one query and 32 consecutive keys per wave and write
// whole 128 B rows; attention3's lane (query, key half) reads 16 B pieces of its own row.  (A layout that is linear in
// attention3's lane order made the producer scatter 16 B runs over 8 lines per head: +30 us per EdgeTransition launch.)
// Buffer size: B * H * Np * Np floats, Np = N rounded up to 32.
__host__ __device__ __forceinline__ long fd_bias_frag_off(long bh, int nt, int i, int j) {
  return ((((bh * nt + (i >> 5)) * nt + (j >> 5)) * 32 + (i & 31)) * 32) + (j & 31);
}

struct ET2Args {
  int B, N;
  const half_t* z_in;   // [B,N,N,128] bf16
  half_t* z_out;        // may alias z_in; NULL (edge_transition4 with pz_out and wb_img only): z' is not stored (nothing reads it)
  const float* e;       // [B*N,128] f32 initial_embed(node)
  const half_t* e_h16; // the same rows in bf16 (edge_transition3: fetched by LDS-DMA)
  const float* a1;      // [B*N,384] f32: W1[:, e_i cols] e_i + b1
  const float* af;      // [B*N,128] f32: Wf[:, e_i cols] e_i + bf
  const void* stream;   // pre-swizzled weight stream (fd_et2_build_stream)
  const float *b2, *gamma, *beta, *res_mask;
  float* trace;         // optional [B,N,N,128] f32
  // optional: pair bias of the NEXT block's attention, linear_b(z') / sqrt(3), emitted from the LayerNorm epilogue
  const void* wb_img;   // fd_chain_build_image_scaled(Wb, H, 128, permuted) of the next block (8 KB), or NULL
  const float* bb;      // [H] pre-scaled bias of linear_b
  float* bias_out;      // fragment order (fd_bias_frag_off)
  int H;
  // edge_transition4: per-residue rows as fold fragments (fd_et4_row_images)
  const void* a1_img;   // [ceil(B*N/8)][16][32][8] bf16: A1 | Af rows of 8 consecutive (flattened) residue rows
  const void* b1_img;   // [B][N/4][16][32][8] bf16: B1 | Bf rows (e_j columns) of 4 consecutive j (+ the next sample's)
  int reserve_cus = 0;  // persistent kernels: CUs left to launches of other streams (FdiptForwardArgs.reserve_cus)
  unsigned long long* clock = nullptr;  // optional shader-clock probe (FdiptForwardArgs.clock_out)
  // optional (edge_transition4 with wb_img set; round 6): pair_z = down_z(z') + b of the NEXT block's IPA (ipa_pytorch.py:158,318) from the
  // same epilogue, as the image opair_pz_kernel reads (fd_pz_bytes): 64 B per pair instead of a second pass over the 256 B of z'
  const void* wdz_img = nullptr;     // (edge_embed2 only) down_z [32, 128] as a fragment image, k in hand-off order (fd_chain_build_image_ex(.., permuted = 1, lo = 0));
  const void* wdz_img_lo = nullptr;  // ... of Wdz - half(Wdz) (lo = 1).  edge_transition4 finds both in the last chunk of its weight stream (fd_et4_set_dz)
  const float* bdz = nullptr;        // [32]
  half_t* pz_out = nullptr;          // [B*N][N/4][32][4]
};
// pair_z image: [flattened residue row b N + i][key group j / 4][channel d < 32][key j % 4] half precision; a 32x32x16 MFMA B fragment of
// opair_pz_kernel is two 8 B pieces per lane (groups 4 s + 2 (lane >> 5), + 1 of k-step s), its producers write whole 256 B groups
static inline size_t fd_pz_bytes(int B, int N) { return (size_t)B * N * ((N + 3) / 4) * 256; }
// edge_transition3.hip: 16-pair waves, two waves per SIMD (any N >= 43)
int fd_et3_build_stream(const float* w1, const float* w2, const float* wf, void* stream, hipStream_t st);
size_t fd_et3_stream_bytes();
int fd_edge_transition3(const ET2Args& a, hipStream_t st);
int fd_et3_build_bias_image(const float* wb, int H, float scale, void* img, hipStream_t st);  // 4 KB, for ET2Args.wb_img
int fd_edge_transition3_supported(int N);
// edge_transition4.hip (default, N % 4 == 0): 32-pair waves (8 i x 4 j patches), e_i / e_j parts folded into one k-step
int fd_et4_build_stream(const float* w1, const float* w2, const float* wf, void* stream, hipStream_t st);
int fd_et4_set_dz(void* stream, const void* img_hi, const void* img_lo, hipStream_t st);  // down_z of the next block -> the stream's last chunk
size_t fd_et4_stream_bytes();
int fd_et4_build_bias_image(const float* wb, int H, float scale, void* img, hipStream_t st);  // 8 KB, for ET2Args.wb_img
size_t fd_et4_a_image_bytes(int B, int N);
size_t fd_et4_b_image_bytes(int B, int N);
// rows [B*N][1024] f32 = [A1 | Af | B1 | Bf] -> ET2Args.a1_img / b1_img
int fd_et4_row_images(const float* rows, int B, int N, void* a_img, void* b_img, hipStream_t st);
int fd_edge_transition4(const ET2Args& a, hipStream_t st);
int fd_edge_transition4_supported(int N);
int fd_ee2_build_images(const float* w2, const float* w3, void* img, hipStream_t st);
size_t fd_ee2_image_bytes();
int fd_edge_embed2(const EdgeEmbedArgs& a, const void* img, hipStream_t st);

struct ProjArgs {
  int B, N, H, C, K, PT, Np;  // K = c_s, PT = point columns (3*H*Pq + 3*H*(Pq+Pv)), Np = keys padded to 32
  const float* A;             // [B*N, lda] node representation
  int lda;
  const void* W;              // [3*H*C + PT, K] bf16 fused projection weight
  const void* W_img;          // the same as a fragment image, zero-padded to whole 128-column blocks (ipa_proj2.hip), or NULL
  const void* W_img_lo = nullptr;  // split operands (ipa_proj2.hip): the image of W - half(W), same layout; NULL = plain half operands
  const float* bias;          // [3*H*C + PT]
  float qscale;               // sqrt(1/(3C)) folded into Q
  half_t *Qb, *Kb, *Vt;
  half_t* Vt_lo = nullptr;    // optional (split operands): V - half(V) in the layout of Vt (the attention's P V on split operands)
  float* pts;                 // [B*N, PT]
  int zero_pads;              // also zero the padded keys [N, Np) of Kb / Vt (first use of the buffers in a forward)
  // merged layout (ipa_proj2.hip only): the columns are [q' (H*C) | points] — no k, no v: the logits are q'.s with
  // q' = W_k^T (W_q s + b_q) (a per-(query, head) constant drops out of the softmax), the values are the node rows themselves with
  // W_v folded into the output projection (model.hip: merged weights).  Kb / Vt are then not written (fd_node_images writes them)
  int merged = 0;
};
// the node rows as attention operand images shared by all heads of a sample (merged projection): Kb [B][Np/32][16][64][8] (key rows,
// 256 channels), Vt / Vt_lo [B][8][Np/16][64][8] (channel rows, keys permuted inside every 16-group), padded keys zero
int fd_node_images(int B, int N, int Np, const float* node, int ld, half_t* Kb, half_t* Vt, half_t* Vt_lo, hipStream_t st);
int fd_ipa_proj2_permute_image_q(void* img, int H, int C, int K, hipStream_t st);  // ... of the q' tiles only (merged layout)
int fd_ipa_proj(const ProjArgs& a, hipStream_t st);
int fd_ipa_proj_zero_pads(const ProjArgs& a, void* extra, size_t extra_bytes, hipStream_t st);  // (Kb, Vt and Vt_lo when set)
int fd_ipa_proj2_supported(const ProjArgs& a);
// after the fragment image of the fused projection weight is built: permute the rows of its Q / K tiles (16 B epilogue stores)
int fd_ipa_proj2_permute_image(void* img, int H, int C, int K, hipStream_t st);
int fd_ipa_proj2(const ProjArgs& a, hipStream_t st);  // second generation (ipa_proj2.hip): same outputs

struct Attn3Args {
  int B, N, H, Np;
  const half_t *Qb, *Kb, *Vt;     // operand images written by ipa_proj_kernel
  const half_t* Vt_lo = nullptr;  // optional: V - half(V), same layout: P V (and the value-point sums) on split operands, P = hi + lo too
  int kv_per_sample = 0;          // Kb / Vt / Vt_lo are indexed by sample, not by (sample, head): the node-row images of the merged projection
  const float* bias;              // pre-scaled pair bias in fd_bias_frag_off order (B*H*Np*Np floats)
  const float* res_mask;          // [B,N]
  const float *qp, *kp, *vp;      // [B,N,H,8,3], [B,N,H,8,3], [B,N,H,12,3] global-frame points (scaled units)
  const half_t* vpt;              // v_pts hi/lo fragment image (PointsArgs.vpt)
  const half_t* kpf = nullptr;    // key-point fragment image (PointsArgs.kpf): required
  const float* gamma;             // [H]
  const float *rot, *trans;       // [B,N,9], [B,N,3]
  float* probs;                   // [B,H,N,N]
  half_t* out_h16;               // if set: the output features are written as bf16 rows (same out_ld, in elements) INSTEAD of out
  half_t* probs_h16;             // if set: written INSTEAD, as bf16 rows [B,N,H,Np] (what the MFMA o_pair kernel consumes)
  float* out;                     // feature rows: o at h*256, point features at pt_off
  long out_ld;
  int pt_off;
};
int fd_attention3_supported(const Attn3Args& a);
int fd_attention3(const Attn3Args& a, hipStream_t st);


// sequence-transformer self-attention (attention_seq.hip): bf16, head_dim 80, N <= 512
size_t fd_seq_attention_image_bytes(int B, int N, int H);
int fd_seq_attention_supported(int N, int H, int hd);
int fd_seq_attention(int B, int N, int H, const float* qkv, int ld, float scale, const float* res_mask, void* images,
                     float* out, int out_ld, hipStream_t st);

// fused form: images initialised once per forward, in_proj writes them directly (seq_qkv), then the attention kernel
struct SeqInitExtra {  // once-per-forward fills folded into the sequence-image init launch (all optional)
  void* fill; long fill_n16;       // zero fill, 16 B units
  void* Kb; void* Vt; long BH; int C;  // padded keys of attention3's key / value images (N, Np as the sequence images: Np = ceil32(N))
  void* Vt2 = nullptr;                 // optional second value image (V_lo) with the same pads
};
int fd_seq_images_init(int B, int N, int H, const float* res_mask, void* images, const SeqInitExtra& x, hipStream_t st);
int fd_seq_qkv_supported(int N, int H, int d_model);
// wimg_lo != NULL: split operands (image of W - half(W), fd_chain_build_image_lo)
int fd_seq_qkv(int B, int N, int H, const float* x, int ld_x, const void* wimg, const void* wimg_lo, const float* bias, float scale,
               void* images, hipStream_t st);
int fd_seq_attention_run(int B, int N, int H, const void* images, float* out, int out_ld, const L2Warm* warm, hipStream_t st);
// fp32 mode: the same block structure on fp32 MFMAs, operands straight from the fp32 in_proj rows qkv [B N, ld] = (q | k | v)
int fd_seq_attention_f32_supported(int N, int H, int hd, int ld);
int fd_seq_attention_f32(int B, int N, int H, const float* qkv, int ld, float scale, const float* res_mask, float* out, int out_ld,
                         hipStream_t st);

// row-complete fused per-residue MLPs (rowblock.hip): 32 rows x all output columns per block, up to 3 Linear layers
// (+ReLU) + residual + LayerNorm + row mask; weights as fd_chain_build_image(.., permuted = 0) fragment images
struct RowBlockArgs {
  int M;
  const float* in;
  int ld_in;
  const void *w0, *w1, *w2;
  const void *w0l = nullptr, *w1l = nullptr, *w2l = nullptr;  // *_SPLIT kinds: images of W - half(W) (fd_chain_build_image_lo)
  const float *b0, *b1, *b2;
  const float* residual;        // or NULL
  int ld_res;
  const float *gamma, *beta;    // LayerNorm kinds
  const float* rowmask_post;    // final * mask, or NULL
  float* out;
  int ld_out;
  float* out2;                  // optional: output columns >= split go to out2 (column - split), or NULL
  int ld_out2, split;
  unsigned short* hid_h16;     // optional bf16 copy of the first hidden layer's rows [M, N1], or NULL
  // FD_RB_ET4_IMAGES: the 1024 output columns [A1 | Af | B1 | Bf] leave as edge_transition4's fold-fragment images (bf16)
  void *img_a, *img_b;          // fd_et4_row_images layouts
  int img_B, img_N;
  // FD_RB_TRANSITION_BB: BackboneUpdate (Linear c_s -> 6, fp32) on the output rows + compose_q_update_vec, in place
  const float *bb_w, *bb_b;     // [6, c_s], [6]
  const float* upd_mask;        // [M] or NULL
  float *quat, *trans;          // [M,4], [M,3]
  L2Warm warm = {};             // weights of the kernel launched next (common.hpp: L2 warm-up hand-over)
  // fd_node_embed16 only: one more Linear (256 -> 256, split operands, fd_chain_build_image16 images) on the output rows -> out2 [M, ld_out2]
  // (skip_embed of all trunk blocks stacked)
  const void *w3 = nullptr, *w3l = nullptr;
  const float* b3 = nullptr;
  // fd_transition16 only (round 4): the EdgeTransition row launch folded in — e = initial_embed(out rows) (Linear 256 -> 128) and the
  // 1024 fold columns [A1 | Af | B1 | Bf] of e (Linear 128 -> 1024), split operands, written as edge_transition4's fold-fragment images
  // img_a / img_b (img_B, img_N as for FD_RB_ET4_IMAGES).  we0 / we1 (+ lo): fd_chain_build_image16 images; NULL = not folded in.
  const void *we0 = nullptr, *we0l = nullptr, *we1 = nullptr, *we1l = nullptr;
  const float *be0 = nullptr, *be1 = nullptr;
};
enum { FD_RB_OUTPROJ, FD_RB_FFN, FD_RB_TRANSITION, FD_RB_NODE_EMBED_72, FD_RB_NODE_EMBED_88, FD_RB_TORSION, FD_RB_TRANSITION_BB, FD_RB_ET_ROWS, FD_RB_ET4_ROWS, FD_RB_ET4_IMAGES,
       FD_RB_TRANSITION_BB_SPLIT, FD_RB_NODE_EMBED_72_SPLIT, FD_RB_NODE_EMBED_88_SPLIT, FD_RB_TORSION_SPLIT };
int fd_rowblock(int kind, const RowBlockArgs& a, hipStream_t st);
// FD_RB_TRANSITION_BB_SPLIT on 16-row blocks (rowblock.hip: transition16_kernel); w0 / w1 / w2 and their lo images are fd_chain_build_image16 images
int fd_transition16(const RowBlockArgs& a, hipStream_t st);
int fd_node_embed16(const RowBlockArgs& a, int k0, hipStream_t st);  // FD_RB_NODE_EMBED_*_SPLIT (first image: K padded to 96)
int fd_torsion16(const RowBlockArgs& a, hipStream_t st);             // FD_RB_TORSION_SPLIT

// post-attention half of one encoder layer in one launch (rowblock.hip): x_a = LN1(x + Wo att + bo); out = LN2(x_a + W2 relu(W1 x_a + b1) + b2)
struct TfmrTailArgs {
  int M, ld;                      // rows; common row stride of att / x / out (d_model = 320)
  const float *att, *x;           // attention output rows, layer input rows (residual)
  const void *wo, *w1, *w2;       // fragment images, natural k order (fd_chain_build_image(.., 0))
  const void *wol = nullptr, *w1l = nullptr, *w2l = nullptr, *wpl = nullptr;  // split operands: lo images (fd_chain_build_image_lo); wol selects the split kernel
  const float *bo, *g1, *be1, *b1, *b2, *g2, *be2;
  float* out;                     // must not alias x
  // optional (last layer of the stack): post_tfmr (Linear 320 -> 256, fragment image wp, bias bp) + residual rows pres on the
  // layer's output -> pout; `out` is then not written
  const void* wp = nullptr;
  const float *bp = nullptr, *pres = nullptr;
  float* pout = nullptr;
  int ld_pres = 0, ld_pout = 0;
  L2Warm warm;                    // weights of the kernel launched next (touched once the block's own first loads are out)
  int rows16 = 0;                 // 1: 16-row blocks (tfmr_tail16_kernel, split operands only): wo / w1 / w2 / wp and their lo images are
                                  // fd_chain_build_image16 images
};
int fd_tfmr_tail(const TfmrTailArgs& a, hipStream_t st);
int fd_chain_build_image16(const float* w, int N, int K, int Kpad, int ldw, int lo, void* img, hipStream_t st);

struct ChainArgs {
  int M;
  const float* in;          // [M, ld_in] fp32 input rows
  int ld_in;
  const void* w[3];         // weight images (fd_chain_build_image) of the layers actually present
  const float* b[3];
  const float* residual;    // added to the output layer (fp32) or NULL
  int ld_res;
  const float *gamma, *beta;     // LayerNorm parameters (kinds with LN)
  unsigned short* out_h16;      // optional bf16 copy of the output rows, [M, NOUT] (kinds without LayerNorm), or NULL
  const float* rowmask_pre;      // (W x + b) * mask before the residual, or NULL
  const float* rowmask_post;     // final * mask, or NULL
  float* out;
  int ld_out;
  L2Warm warm = {};  // weights of the kernel launched next (common.hpp: L2 warm-up hand-over)
};
enum { FD_CHAIN_TRANSITION, FD_CHAIN_FFN, FD_CHAIN_OUTPROJ, FD_CHAIN_POST, FD_CHAIN_INPROJ, FD_CHAIN_SKIP, FD_CHAIN_ETINIT,
       FD_CHAIN_A1, FD_CHAIN_AF, FD_CHAIN_NODE_EMBED_72, FD_CHAIN_NODE_EMBED_88, FD_CHAIN_TORSION };
size_t fd_chain_image_bytes(int N, int K);
int fd_chain_build_image(const float* w, int N, int K, int ldw, int permuted, void* img, hipStream_t st);
// the same for W - half(W): the lo part of a weight matrix used as split operands (hi image + lo image = 22 significant bits)
int fd_chain_build_image_lo(const float* w, int N, int K, int ldw, void* img, hipStream_t st);
int fd_chain_build_image_scaled(const float* w, int N, int K, int ldw, int permuted, float scale, void* img, hipStream_t st);
int fd_chain_build_image_ex(const float* w, int N, int K, int ldw, int permuted, int lo, void* img, hipStream_t st);  // any (k order, part)
int fd_chain(int kind, const ChainArgs& a, hipStream_t st);

int fd_linear(int precision, int M, int N, int K, const float* A, int lda, const void* W, int ldw, const float* bias,
              const float* residual, int ldr, const float* rowmask, int relu, float* out, int ldo, hipStream_t st);
int fd_linear_z(int precision, long M, int N, int K, const void* A, const void* W, const float* bias, float* out,
                hipStream_t st);
int fd_layernorm_parts(int M, int D, const float* x, int ldx, const float* parts, int ldr, int nparts, long part_stride,
                       const float* gamma, const float* beta, const float* rowmask, float* out, int ldo, const float* extra,
                       int ld_extra, int n_extra, const L2Warm* warm, hipStream_t st);
// the same with bf16 activation rows (what the bf16 GEMM would round them to anyway)
int fd_linear_splitk_a16(int M, int N, int K, int nsplit, const half_t* A, int lda, const void* W, int ldw, const float* bias,
                         const float* rowmask, float* parts, long part_stride, int ldo, hipStream_t st);
int fd_linear_splitk_split(int M, int N, int K, int nsplit, const float* A, int lda, const float* W, int ldw, const float* bias,
                           const float* rowmask, float* parts, long part_stride, int ldo, hipStream_t st);
// dedicated split-operand kernel for the IPA output projection at the reference widths (gemm.hip: outproj_split_kernel)
int fd_outproj_split_supported(int N, int K);
int fd_outproj_split_slices();
int fd_outproj_split(int M, int N, int K, const float* A, int lda, const void* w_hi, const void* w_lo, const float* bias, const float* rowmask,
                     float* parts, long part_stride, int ldo, hipStream_t st);
int fd_linear_splitk(int M, int N, int K, int nsplit, const float* A, int lda, const void* W, int ldw, const float* bias,
                     const float* rowmask, float* parts, long part_stride, int ldo, hipStream_t st);
int fd_layernorm(int M, int D, const float* x, int ldx, const float* residual, int ldr, const float* gamma,
                 const float* beta, const float* rowmask, float* out, int ldo, hipStream_t st);
int fd_f32_to_half(long n, const float* in, half_t* out, hipStream_t st);
int fd_edge_transition(int precision, int cz, int cb, const EdgeTransArgs& a, hipStream_t st);
int fd_edge_embed(int precision, int cz, const EdgeEmbedArgs& a, hipStream_t st);
int fd_attention(int precision, int ipa, const AttnArgs& a, hipStream_t st);
// fp32 mode, reference widths: the IPA attention with the scores in registers (attention.hip: ipa_attn_f32_kernel)
int fd_ipa_attention_f32_supported(const AttnArgs& a);
int fd_ipa_attention_f32(const AttnArgs& a, hipStream_t st);
int fd_opair(int precision, const OPairArgs& a, hipStream_t st);
int fd_opair_mfma_eligible(int precision, const OPairArgs& a);  // the MFMA kernel will run (it can take probs_h16)
int fd_opair_pz(const OPairArgs& a, hipStream_t st);  // o_pair from the producer-emitted pair_z image (OPairArgs.pz, probs_h16)
int fd_pair_bias2(int B, int N, int H, const void* z, const void* wb, const float* bb, float* out, int frag, hipStream_t st);
// fp32 mode: out[p, h] = z[p, :] . Wb[h, :] + bb[h] over the fp32 pair representation (H = 8, c_z = 128), one streaming pass
int fd_pair_bias_f32(long n_pairs, int H, int CZ, const float* z, const float* wb, const float* bb, float* out, hipStream_t st);
int fd_points(const PointsArgs& a, hipStream_t st);
int fd_compose_q_update(long n, float* quat, float* trans, const float* upd, int ld_upd, const float* mask, hipStream_t st);
int fd_split_rigids(long n, const float* t7, float cs, const float* res_mask, const float* fixed_mask, float* quat,
                    float* trans, float* dmask, const int32_t* cursor, hipStream_t st);
int fd_finish(long n, const float* quat, const float* trans, float cs, const float* psi_un, int ld_psi,
              const float* gt_psi, const float* fixed_mask, float* rigids, float* psi, hipStream_t st);
int fd_build_feats(int B, int N, int use_aatype, int E, const int32_t* aatype, const float* t_emb, const float* t_emb_eps,
                   const float* fixed_mask, const float* idx_emb, float* node_feat, int ld_node, float* pte, int ld_pte,
                   const float* t7, const float* res_mask, float cs, float* quat, float* trans, float* dmask, const float* w1i,
                   const float* w1j, const float* b1, int cz, float* pi, float* pj, const int32_t* cursor, hipStream_t st);
int fd_score_tail(int B, int N, const float* rigids_t, const float* quat, const float* trans, float cs, const float* psi_un,
                  int ld_psi, const float* gt_psi, const float* fixed_mask, const float* res_mask, const double* sigma,
                  const float* t, float min_b, float max_b, float* rigids, float* psi, double* rot_score, float* trans_score,
                  float* ca_out, const float* hid, int ld_hid, int c_hid, const float* torf_w, const float* torf_b,
                  const double* score_table, const double* omega_edges, int n_omega, const int32_t* aatype, const void* bb_tables,
                  float* atom37, float* atom14, const int32_t* cursor, hipStream_t st);
int fd_rot_score(int B, int N, const float* qt, int ld_t, const float* q0, int ld_0, const double* sigma,
                 const float* res_mask, double* score, hipStream_t st);
int fd_trans_score(int B, int N, const float* tt, int ld_t, const float* t0, int ld_0, const float* t, float min_b,
                   float max_b, float cs, const float* res_mask, float* score, hipStream_t st);
int fd_backbone(int n, const float* t7, const float* rot, const float* trans, int ld_trans, const float* psi,
                const int32_t* aatype, const void* tables, float* atom37, float* atom14, hipStream_t st,
                const int32_t* cursor = nullptr);
