/*
 * fdipt.h — C ABI of libfdipt_hip.so: MI355X (gfx950) kernels for the FrameDiPT sampler hot path.
 *
 * The reference (instadeepai/FrameDiPT) is pure Python/PyTorch and has no FFI; this ABI is the
 * boundary a maintainer binds with ctypes.  Each entry point names the
 * reference function(s) it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative FDIPT_E* code otherwise; no exceptions,
 *     no allocation, no global mutable state; one hipStream_t per call (passed as void*).  Calls for different devices
 *     are independent, and so are calls on different streams of one device as long as they share no output / workspace buffer
 *     (forwards of TWO sub-batches may be in flight together: FdiptForwardArgs.reserve_cus; verified bit-identical to the
 *     sequential result in long soaks for two streams, NOT for three or more — DESIGN.md section 5 — so the Python host
 *     refuses more than two).  No device-side state survives a call outside caller-owned buffers; the only atomics
 *     (the optional clock probe, the optional step cursor of the device-resident loop) target a caller-owned buffer.
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller (PyTorch) owns
 *     every buffer, including the workspace (query sizes with the *_bytes functions).
 *   - layouts are row-major contiguous, residue-major.  Quaternions are scalar-first (w,x,y,z);
 *     tensor_7 = quat(4) | translation in Angstrom (3)   (openfold/utils/rigid_utils.py:1200-1230).
 *   - "f32"/"f64" in a parameter comment is the element type of the buffer.
 */
#ifndef FDIPT_H
#define FDIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDIPT_OK 0
#define FDIPT_EINVAL (-1)   /* bad argument (null pointer, size out of range)            */
#define FDIPT_ELAUNCH (-2)  /* HIP launch error (hipGetLastError != hipSuccess)          */
#define FDIPT_ESIZE (-3)    /* workspace too small / N beyond the compiled LDS tiling     */

/* GEMM operand precision of the score network (accumulation is always fp32). */
#define FDIPT_PREC_F32 0  /* v_mfma_f32_32x32x2_f32: exact fp32, parity mode               */
#define FDIPT_PREC_BF16 1 /* v_mfma_f32_32x32x16_bf16: bf16 operands, pair rep kept in bf16.  Only in a library built
                             with -DFDIPT_HALF_BF16 (development comparison); the default build answers FDIPT_EINVAL */
#define FDIPT_PREC_F16 2  /* v_mfma_f32_32x32x16_f16: fp16 operands (11 significant bits, same MFMA rate as bf16), pair
                             rep kept in fp16; frames, points, softmax statistics, LayerNorm and scores stay fp32/fp64.
                             Throughput mode of the default build */

/* FdiptDims.kernel_flags: run a fallback path of the half-precision mode at shapes where the default selection would not
 * (each of them is what some other shape uses anyway; parity tests run them at the golden sizes).  Same flags for
 * fdipt_model_prepare and every forward of a model. */
#define FDIPT_KF_ET3 1           /* EdgeTransition: the 16-pair-wave kernel (default for N % 4 != 0) for every N >= 43   */
#define FDIPT_KF_GENERIC_PAIR 2  /* EdgeTransition / edge embedder: the any-width LDS-chain kernels                      */
#define FDIPT_KF_GENERIC_ATTN 4  /* attention: the LDS-score kernels (the fallback for non-reference widths) in both modes */
#define FDIPT_KF_UNFUSED_NODE 8  /* node path as plain GEMM + LayerNorm launches (default for non-reference widths)     */
#define FDIPT_KF_UNFOLDED 16     /* launch folds off: pair bias / feature split / torsion head / fills as own launches  */
#define FDIPT_KF_NO_SPLIT 32     /* node-path layers on plain half-precision operands instead of split (hi + lo) operands:
                                    ~8 % faster, 5x the error of the predicted frames / psi (precision modes)      */
#define FDIPT_KF_NO_MERGE 64     /* IPA projections in the reference's formulation (k and v explicit) instead of the merged one
                                    (keys = values = the node rows, W_k folded into the query, W_v into the output projection)  */
#define FDIPT_KF_ROWS32 128      /* node path: the 32-row-block kernels (the default for N > 512) instead of the 16-row ones (tails, transition,
                                    node embedder, torsion head) for every N                                                  */
#define FDIPT_KF_PASS_Z 256      /* o_pair as its own pass over the 128 channels of z (round 5's path: sum_j a z, then down_z) instead of the
                                    pair_z image emitted by the producers of z (round 6); the last EdgeTransition then keeps its z' store */
#define FDIPT_KF_ALL 511

typedef void* fdipt_stream_t; /* hipStream_t */

/* ---------------------------------------------------------------- model description -------- */
/* Dimensions of ScoreNetwork (framedipt/model/score_network.py:200-216, config/base.yaml:55-79). */
typedef struct FdiptDims {
  int32_t c_s;         /* node_embed_size = ipa.c_s           (256) */
  int32_t c_z;         /* edge_embed_size = ipa.c_z           (128) */
  int32_t c_hidden;    /* ipa.c_hidden                         (256) */
  int32_t c_skip;      /* ipa.c_skip                           (64)  */
  int32_t no_heads;    /* ipa.no_heads                         (8)   */
  int32_t no_qk_points; /* ipa.no_qk_points                    (8)   */
  int32_t no_v_points; /* ipa.no_v_points                      (12)  */
  int32_t tfmr_heads;  /* ipa.seq_tfmr_num_heads               (4)   */
  int32_t tfmr_layers; /* ipa.seq_tfmr_num_layers              (2)   */
  int32_t num_blocks;  /* ipa.num_blocks                       (4)   */
  int32_t index_embed; /* embed.index_embed_size               (32)  */
  int32_t num_bins;    /* embed.num_bins                       (22)  */
  int32_t use_aatype;  /* 1: node features carry a 21-way aatype one-hot (inpainting / input_aatype) */
  int32_t precision;   /* FDIPT_PREC_*                                                    */
  int32_t kernel_flags; /* FDIPT_KF_* bits; 0 = default kernel selection                  */
  float min_bin;       /* embed.min_bin (1e-5) */
  float max_bin;       /* embed.max_bin (20)   */
  float coordinate_scaling; /* ipa.coordinate_scaling = diffuser.r3.coordinate_scaling (0.1) */
  float r3_min_b;      /* diffuser.r3.min_b (0.1) */
  float r3_max_b;      /* diffuser.r3.max_b (20)  */
} FdiptDims;

/* Number of parameter tensors / total fp32 elements of the reference state_dict for `dims`
 * (same order as ScoreNetwork(...).state_dict(); framedipt_amd/weights.py:param_shapes). */
int fdipt_param_count(const FdiptDims* dims);
int64_t fdipt_param_offset(const FdiptDims* dims, int index); /* element offset of tensor `index` in the flat blob; index==count -> total */

/* Bytes of the derived-weights blob (operand-precision copies, fused / split matrices). */
size_t fdipt_derived_bytes(const FdiptDims* dims);
/* Build the derived blob from the flat fp32 state_dict blob (device, order above).  Once per model. */
int fdipt_model_prepare(const FdiptDims* dims, const float* params_f32, void* derived, fdipt_stream_t stream);

/* ---------------------------------------------------------------- per-batch sample setup ---- */
/* Constant-per-trajectory tables of a batch of B samples with N residues each:
 *   seq_idx [B,N] i32; idx_emb [B,N,index_embed] f32 = get_index_embedding(seq_idx) and
 *   rel_emb [B,n_rel,index_embed] f32 = get_index_embedding(r - rel_off), r in [0,n_rel) — both evaluated
 *   by the host exactly as the reference does in float32 (framedipt/model/score_network.py:17-38).
 * Produces in `setup` the relative-position part of the first edge-embedder layer
 * (score_network.py:98-105,184-187 restructured: concat-free first layer). */
size_t fdipt_setup_bytes(const FdiptDims* dims, int B, int N, int n_rel);
int fdipt_sample_setup(const FdiptDims* dims, const float* params_f32, const void* derived, int B, int N, int n_rel,
                       const float* rel_emb, void* setup, fdipt_stream_t stream);

/* ---------------------------------------------------------------- score network forward ---- */
/* Replaces ScoreNetwork.forward (framedipt/model/score_network.py:218-275) = Embedder.forward (:129-197)
 * + IpaScore.forward (framedipt/model/ipa_pytorch.py:509-572) + compute_backbone
 * (framedipt/protein/all_atom.py:147-176), for a batch of B equally sized samples. */
typedef struct FdiptForwardArgs {
  int32_t B, N, n_rel, rel_off;   /* rel index of pair (i,j) = seq_idx[i]-seq_idx[j]+rel_off          */
  const float* rigids_t;          /* [B,N,7] f32  input frames x_t                                     */
  const float* res_mask;          /* [B,N]   f32                                                      */
  const float* fixed_mask;        /* [B,N]   f32  1 = motif residue (not diffused)                    */
  const float* sc_ca_t;           /* [B,N,3] f32  self-conditioning CA positions (Angstrom)           */
  const int32_t* seq_idx;         /* [B,N]   i32                                                      */
  const float* idx_emb;           /* [B,N,index_embed] f32                                            */
  const int32_t* aatype;          /* [B,N] i32 pre-processed aatype (0..20) or NULL (de novo)         */
  const float* gt_psi;            /* [B,N,2] f32 torsion_angles_sin_cos[...,2,:]                      */
  const float* t;                 /* [B] f32 diffusion time                                           */
  const float* t_emb;             /* [B,index_embed] f32 get_timestep_embedding(t)   (host, float32)  */
  const float* t_emb_eps;         /* [index_embed]   f32 get_timestep_embedding(1e-5) (inpainting)    */
  const double* so3_sigma;        /* [B] f64 discrete_sigma[t_to_idx(t)] (so3_diffuser.py:398)        */
  const void* bb_tables;          /* residue tables for fdipt_backbone_atoms (needed iff atom37/atom14 != NULL) */
  /* outputs */
  float* psi;                     /* [B,N,2]  f32 */
  double* rot_score;              /* [B,N,3]  f64 */
  float* trans_score;             /* [B,N,3]  f32 */
  float* rigids;                  /* [B,N,7]  f32 predicted x_0 frames */
  float* atom37;                  /* [B,N,37,3] f32 or NULL */
  float* atom14;                  /* [B,N,14,3] f32 or NULL */
  /* optional traces for parity tests (NULL to skip): node / pair representation after each block */
  float* trace_node;              /* [num_blocks+1,B,N,c_s] f32: [0]=embedder output               */
  float* trace_edge;              /* [num_blocks,B,N,N,c_z] f32: [0]=embedder output, [b+1]=EdgeTransition b */
  /* optional, parity tests of the per-block sub-modules: [num_blocks,4,B,N,c_s+c_skip] f32 with slot 0 = IPA output
   * (ipa_pytorch.py:531, first c_s columns), 1 = post-IPA LayerNorm (:532, c_s), 2 = sequence-transformer output (:536-538,
   * c_s+c_skip), 3 = BackboneUpdate output (:542-545, 6).  Only the unfused node path keeps these tensors in memory: fp32
   * precision, or FDIPT_KF_UNFUSED_NODE | FDIPT_KF_UNFOLDED; FDIPT_EINVAL otherwise.  NULL to skip. */
  float* trace_inner;
  /* optional profiling: hipEvent_t pairs recorded on `stream` around every EdgeTransition launch
   * (num_blocks-1 pairs, created with fdipt_event_create); NULL to skip */
  void** ev_start;                /* host array of events */
  void** ev_stop;
  /* optional: contiguous copy of the predicted CA positions rigids[...,4:] ([B,N,3] f32, Angstrom), written at the end of
   * the forward.  May alias sc_ca_t (read at the start): the sampler's self-conditioning hand-over
   * (experiments/utils.py:361-366,571-578) then costs no copy.  NULL to skip. */
  float* ca_out;
  /* Concurrent sub-batches (one forward per HIP stream): the persistent pair kernels (edge embedder, EdgeTransition) start
   * on all CUs but `reserve_cus` of them, so that the latency-bound node-path launches of another stream keep finding free
   * CUs while they run.  0 = use every CU (single stream). */
  int32_t reserve_cus;
  /* optional profiling: device buffer of 3 x uint64 (caller-owned, zeroed by the caller).  Thread 0 of every EdgeTransition
   * block adds its shader-clock cycles (s_memtime), its 100 MHz ticks (s_memrealtime) and 1 to [0], [1], [2]: the clock the
   * kernel's blocks actually ran at = [0] / [1] / 10 GHz (bench.py: roofline.clock_ghz).  NULL (the default): the kernels
   * execute no atomics and keep no state outside the caller's buffers. */
  unsigned long long* clock_out;
  /* optional: so3.use_cached_score = True (so3_diffuser.py:389-396) — the rotation-score norm is looked up instead of evaluated:
   * so3_score_table [B, so3_num_omega] f64 = the row of the reference's _score_norms table at each sample's t,
   * so3_omega_edges [so3_num_omega - 1] f64 = discrete_omega[:-1]; index = torch.bucketize(omega, edges).  NULL: the series. */
  const double* so3_score_table;
  const double* so3_omega_edges;
  int32_t so3_num_omega;
  /* optional: step cursor of a device-resident reverse loop (experiments/utils.py:584-602, the `for t in reverse_steps` loop).
   * NULL (the default): every pointer above is what its comment says.  Non-NULL: a device int32[2] = { step index k, ticket }, and
   * the per-step inputs / outputs are BASES of step-major arrays of which the kernels read / write row k = step_cursor[0]:
   *   rigids_t [T+1,B,N,7] (x_t of step k = row k: the rigid trajectory itself), t [T,B], t_emb [T,B,index_embed], so3_sigma [T,B],
   *   so3_score_table [T,B,so3_num_omega], atom37 [T,B,N,37,3] (rigid_0_traj).
   * The launch sequence then does not depend on k: a step (this forward + fdipt_se3_reverse_step_indexed, which advances the
   * cursor) can be captured once as a HIP graph and replayed for every step of every trajectory of the shape. */
  const int32_t* step_cursor;
} FdiptForwardArgs;

size_t fdipt_forward_workspace_bytes(const FdiptDims* dims, int B, int N);
int fdipt_score_forward(const FdiptDims* dims, const float* params_f32, const void* derived, const void* setup,
                        const FdiptForwardArgs* args, void* workspace, size_t workspace_bytes, fdipt_stream_t stream);

/* ---------------------------------------------------------------- sub-modules of the forward ---- */
/* The forward's launch schedule cut at a sub-module boundary, on caller-provided inputs (callers that assemble their own
 * network, per-module parity tests).  Same model handles (dims / params / derived) and workspace as fdipt_score_forward;
 * "pair type" = float in FDIPT_PREC_F32, IEEE half in FDIPT_PREC_F16.  node [B,N,c_s] f32, z [B,N,N,c_z] pair type,
 * rigids [B,N,7] f32 tensor_7 in Angstrom (scaled by coordinate_scaling inside, ipa_pytorch.py:524), res_mask [B,N] f32. */
/* Embedder.forward (framedipt/model/score_network.py:129-197): node and pair embeddings, masked as score_network.py:236-237.
 * Reads B, N, n_rel, rel_off, res_mask, fixed_mask, sc_ca_t, seq_idx, idx_emb, aatype, t_emb, t_emb_eps of `args`. */
int fdipt_edge_embed_fwd(const FdiptDims* dims, const float* params_f32, const void* derived, const void* setup,
                         const FdiptForwardArgs* args, float* node_out, void* z_out, void* workspace, size_t workspace_bytes,
                         fdipt_stream_t stream);
/* Point projections of InvariantPointAttention `block` in the global frame (ipa_pytorch.py:213-239): q_pts, k_pts
 * [B,N,H,no_qk_points,3] and v_pts [B,N,H,no_v_points,3] f32, in scaled units (nm). */
int fdipt_ipa_project_points(const FdiptDims* dims, const float* params_f32, const void* derived, int block, int B, int N,
                             const float* node, const float* rigids, const float* res_mask, float* q_pts, float* k_pts,
                             float* v_pts, void* workspace, size_t workspace_bytes, fdipt_stream_t stream);
/* InvariantPointAttention.forward of `block` (ipa_pytorch.py:170-329: projections, logits with pair bias and point distances,
 * softmax, o / o_pt / o_pair, linear_out) times res_mask (:531): out [B,N,c_s] f32. */
int fdipt_ipa_attention_fwd(const FdiptDims* dims, const float* params_f32, const void* derived, int block, int B, int N,
                            const float* node, const void* z, const float* rigids, const float* res_mask, float* out,
                            void* workspace, size_t workspace_bytes, fdipt_stream_t stream);
/* EdgeTransition.forward of `block` < num_blocks - 1 (ipa_pytorch.py:84-102) times the pair mask (:549): z_out may alias z_in. */
int fdipt_edge_transition_fwd(const FdiptDims* dims, const float* params_f32, const void* derived, int block, int B, int N,
                              const float* node, const float* res_mask, const void* z_in, void* z_out, void* workspace,
                              size_t workspace_bytes, fdipt_stream_t stream);

/* ---------------------------------------------------------------- reverse step ------------- */
/* Replaces SE3Diffuser.reverse (framedipt/diffusion/se3_diffuser.py:346-401) with
 * _extract_trans_rots (:16-23), SO3Diffuser.reverse (so3_diffuser.py:569-602), compose_rotvec
 * (framedipt/data/transforms.py:33-46, SciPy Rotation conventions), R3Diffuser.reverse
 * (r3_diffuser.py:344-385), _assemble_rigid (:26-36) and Rigid.to_tensor_7 (rigid_utils.py:1200-1212),
 * fused, float64 internally.  z_rot/z_trans are N(0,1) draws (host noise tape; scaled by noise_scale here).
 * out_rot (optional) receives the float32 rotation matrices of x_{t-1} as the reference's Rigid holds them. */
int fdipt_se3_reverse_step(int B, int N, const float* rigids_t, const double* rot_score, const float* trans_score,
                           const float* diffuse_mask /* [B,N] f32 or NULL */, const double* z_rot, const double* z_trans,
                           double t, double dt, double noise_scale, int center, int diffuse_rot, int diffuse_trans,
                           double so3_min_sigma, double so3_max_sigma, double r3_min_b, double r3_max_b,
                           double coordinate_scaling, float* rigids_out /* [B,N,7] */, float* out_rot /* [B,N,3,3] or NULL */,
                           fdipt_stream_t stream);
/* The same step followed by all_atom.compute_backbone on x_{t-1} (experiments/utils.py:376-388: the atom37 frame of the
 * trajectory) in the same launch: psi [B,N,2], aatype [B,N] or NULL, tables as for fdipt_backbone_atoms, atom37 [B,N,37,3].
 * rigids_out must not alias rigids_t. */
int fdipt_se3_reverse_step_atoms(int B, int N, const float* rigids_t, const double* rot_score, const float* trans_score,
                                 const float* diffuse_mask, const double* z_rot, const double* z_trans, double t, double dt,
                                 double noise_scale, int center, int diffuse_rot, int diffuse_trans, double so3_min_sigma,
                                 double so3_max_sigma, double r3_min_b, double r3_max_b, double coordinate_scaling,
                                 float* rigids_out, float* out_rot, const float* psi, const int32_t* aatype,
                                 const void* tables, float* atom37, fdipt_stream_t stream);

/* ... and the step's row of trans_traj (experiments/utils.py:390-400) from the same launch:
 * trans_traj[b,i] = diffuse_mask * trans(pred_rigids) + traj_fixed_mask * trans(x_{t-1}); pred_rigids [B,N,7] = the forward's x_0
 * prediction, traj_fixed_mask [B,N] = fixed_mask * res_mask.  trans_traj == NULL: exactly fdipt_se3_reverse_step_atoms. */
int fdipt_se3_reverse_step_traj(int B, int N, const float* rigids_t, const double* rot_score, const float* trans_score,
                                const float* diffuse_mask, const double* z_rot, const double* z_trans, double t, double dt,
                                double noise_scale, int center, int diffuse_rot, int diffuse_trans, double so3_min_sigma,
                                double so3_max_sigma, double r3_min_b, double r3_max_b, double coordinate_scaling,
                                float* rigids_out, float* out_rot, const float* psi, const int32_t* aatype,
                                const void* tables, float* atom37, const float* pred_rigids, const float* traj_fixed_mask,
                                float* trans_traj, fdipt_stream_t stream);

/* The same fused step (reverse + backbone frame + trans_traj row) of a device-resident loop, addressed through a step cursor:
 * one step k = step_cursor[0] of the `for t in reverse_steps` loop (experiments/utils.py:584-602 -> one_step_inference :292-412).
 * Reads x_t = rigid_traj[k], writes x_{t-1} = rigid_traj[k+1], prot_traj[k], trans_traj[k]; uses z_rot[k], z_trans[k],
 * t = t_table[k]; the last block to finish sets step_cursor[0] = k + 1 (step_cursor[1] is its ticket, 0 between launches).  The
 * launch arguments do not depend on k, so a captured HIP graph of a step replays for every k. */
typedef struct FdiptReverseIndexed {
  int32_t B, N;
  float* rigid_traj;             /* [T+1,B,N,7] f32: row 0 = x_T */
  const double* rot_score;       /* [B,N,3] f64 (the forward's output buffer) */
  const float* trans_score;      /* [B,N,3] f32 */
  const float* diffuse_mask;     /* [B,N] f32 or NULL */
  const double* z_rot;           /* [T-1,B,N,3] f64 N(0,1) */
  const double* z_trans;         /* [T-1,B,N,3] f64 N(0,1) */
  const double* t_table;         /* [T] f64: reverse_steps (device) */
  double dt, noise_scale;
  int32_t center, diffuse_rot, diffuse_trans;
  double so3_min_sigma, so3_max_sigma, r3_min_b, r3_max_b, coordinate_scaling;
  const float* psi;              /* [B,N,2] f32 (forward output) */
  const int32_t* aatype;         /* [B,N] i32 or NULL */
  const void* bb_tables;
  float* prot_traj;              /* [T,B,N,37,3] f32 or NULL */
  const float* pred_rigids;      /* [B,N,7] f32 (forward output) */
  const float* traj_fixed_mask;  /* [B,N] f32 */
  float* trans_traj;             /* [T,B,N,3] f32 or NULL */
  int32_t* step_cursor;          /* device int32[2] */
} FdiptReverseIndexed;
int fdipt_se3_reverse_step_indexed(const FdiptReverseIndexed* args, fdipt_stream_t stream);
/* compute_backbone of n = B*N frames (tensor_7) into row step_cursor[0] of atom37_rows [T,n,37,3] (rigid_0_traj rows built with the
 * caller's aatype view, experiments/utils.py:397-402). */
int fdipt_backbone_atoms_indexed(int n, const float* t7, const float* psi, const int32_t* aatype, const void* tables,
                                 float* atom37_rows, const int32_t* step_cursor, fdipt_stream_t s);

/* ---------------------------------------------------------------- EigenFold confidence score (f4) */
/* SE3Diffuser.forward: one-step forward noising q(x_t | x_{t-1}) (framedipt/diffusion/se3_diffuser.py:50-95; r3_diffuser.py:122-161
 * with center=False; so3_diffuser.py:408-443).  State = what the reference carries between steps: float32 rotation matrices
 * rot [B,N,3,3] and translations trans [B,N,3] in Angstrom; z_* are N(0,1) draws (float64); diffuse_mask [B,N] or NULL.
 * rigids_t (optional, [B,N,7]) receives Rigid.to_tensor_7 of the result (the next forward's input). */
int fdipt_se3_forward_step(int B, int N, const float* rot_t_1, const float* trans_t_1, const float* diffuse_mask, const double* z_rot,
                           const double* z_trans, double t_1, double dt, double noise_scale, double so3_min_sigma, double so3_max_sigma,
                           double r3_min_b, double r3_max_b, double coordinate_scaling, float* rot_t, float* trans_t, float* rigids_t,
                           fdipt_stream_t stream);
/* SE3Diffuser.log_prob_backward / log_prob_forward (se3_diffuser.py:97-196; r3_diffuser.py:163-260; so3_diffuser.py:99-119,466-567;
 * r3_utils.py:10-42) of one step, summed over the diffused residues of each sample in float64:
 * out[b] = { log p_trans(x_{t-1}|x_t), log p_rot(x_{t-1}|x_t), log q_trans(x_t|x_{t-1}), log q_rot(x_t|x_{t-1}) }.
 * rot_score [B,N,3] float64 / trans_score [B,N,3] float32 are the model's scores at time t; both NULL: forward terms only. */
int fdipt_se3_step_log_prob(int B, int N, const float* rot_t, const float* trans_t, const float* rot_t_1, const float* trans_t_1,
                            const double* rot_score, const float* trans_score, const float* diffuse_mask, double t, double t_1, double dt,
                            double so3_min_sigma, double so3_max_sigma, double r3_min_b, double r3_max_b, double coordinate_scaling,
                            double* out, fdipt_stream_t stream);
/* Terminal term of logp_confidence_score (experiments/utils.py:846-866): out[b] = { sum log N(0,1)(scaled trans_T), log(1/pi^2) * n_diffused } */
int fdipt_se3_prior_log_prob(int B, int N, const float* trans_T, const float* diffuse_mask, double coordinate_scaling, double* out,
                             fdipt_stream_t stream);

/* ---------------------------------------------------------------- frame algebra (a8) ------- */
/* openfold/utils/rigid_utils.py free functions and Rigid/Rotation methods, n independent items, f32. */
int fdipt_quat_to_rot(int n, const float* quat, float* rot, fdipt_stream_t s);           /* :185 */
int fdipt_rot_to_quat(int n, const float* rot, float* quat, fdipt_stream_t s);           /* :208 (sign may differ) */
int fdipt_quat_multiply(int n, const float* q1, const float* q2, float* out, fdipt_stream_t s); /* :254 */
int fdipt_quat_multiply_by_vec(int n, const float* q, const float* v, float* out, fdipt_stream_t s); /* :266 */
int fdipt_invert_quat(int n, const float* q, float* out, fdipt_stream_t s);              /* :282 */
int fdipt_rigid_apply(int n, const float* t7, const float* pts, float* out, fdipt_stream_t s);        /* :1104 */
int fdipt_rigid_invert_apply(int n, const float* t7, const float* pts, float* out, fdipt_stream_t s); /* :1118 */
int fdipt_rigid_compose(int n, const float* a_t7, const float* b_t7, float* out_rot, float* out_trans, fdipt_stream_t s); /* :1065 */
int fdipt_rigid_invert(int n, const float* t7, float* out_rot, float* out_trans, fdipt_stream_t s);   /* :1132 */
int fdipt_rigid_compose_q_update(int n, const float* t7, const float* upd6, const float* mask, float* out_t7,
                                 fdipt_stream_t s);                                      /* :587,:1039 (fork: update_mask) */
int fdipt_quat_to_rotvec(int n, const float* q, float* rotvec, fdipt_stream_t s);        /* framedipt/data/transforms.py:53-69 */
int fdipt_rigid_from_3_points(int n, const float* p_neg_x_axis, const float* origin, const float* p_xy_plane, float eps,
                              float* rot /* [n,3,3]; the translation is `origin` */, fdipt_stream_t s);  /* :1233-1275 */
/* SO(3) exp / log of the geomstats fork (framedipt/diffusion/so3_utils.py): rot_mat_from_axis_angle_by_exp_map (:90-100),
 * rotation_vector_from_matrix (:120-190, with regularize :193-231), omega (:103-117); float64 buffers */
int fdipt_so3_exp_geomstats(int n, const double* rotvec, double* rot, fdipt_stream_t s);
int fdipt_so3_log_geomstats(int n, const double* rot, double* rotvec, fdipt_stream_t s);
int fdipt_so3_omega(int n, const double* rot, double eps, double* angle, fdipt_stream_t s);
/* SciPy Rotation conventions (float64): exp = from_rotvec().as_matrix(), log = from_matrix().as_rotvec() */
int fdipt_so3_exp(int n, const double* rotvec, double* rot, fdipt_stream_t s);
int fdipt_so3_log(int n, const double* rot, double* rotvec, fdipt_stream_t s);

/* ---------------------------------------------------------------- scores / backbone -------- */
/* SE3Diffuser.calc_rot_score (se3_diffuser.py:281-292 -> so3_diffuser.py:373-402,18-77,122-191): 1000-term IGSO(3)
 * series; quats_t = noisy x_t, quats_0 = prediction, sigma[B] as in FdiptForwardArgs. */
int fdipt_igso3_rot_score(int B, int N, const float* quats_t, const float* quats_0, const double* sigma,
                          const float* res_mask, double* score, fdipt_stream_t s);
/* ... with so3.use_cached_score = True: score_table [B, n_omega], omega_edges [n_omega - 1] as in FdiptForwardArgs. */
int fdipt_igso3_rot_score_cached(int B, int N, const float* quats_t, const float* quats_0, const double* score_table,
                                 const double* omega_edges, int n_omega, const float* res_mask, double* score, fdipt_stream_t s);
/* SE3Diffuser.calc_trans_score(use_torch=True, scale=True) (se3_diffuser.py:269-279 -> r3_diffuser.py:410-440). */
int fdipt_r3_trans_score(int B, int N, const float* trans_t, const float* trans_0, const float* t, float min_b,
                         float max_b, float coordinate_scaling, const float* res_mask, float* score, fdipt_stream_t s);
/* all_atom.compute_backbone (framedipt/protein/all_atom.py:147-176): frames given as tensor_7 (rot==NULL) or as
 * rot [n,3,3] + trans [n,3]; tables = default_frames[21,8,4,4] | ideal_pos[21,14,3] | atom_mask[21,14] (f32) then
 * group_idx[21,14] (i32) packed as in framedipt_amd/residue_tables.py. */
int fdipt_backbone_atoms(int n, const float* t7, const float* rot, const float* trans, const float* psi,
                         const int32_t* aatype, const void* tables, float* atom37, float* atom14, fdipt_stream_t s);

/* ---------------------------------------------------------------- building blocks ---------- */
/* Exposed for parity tests and for callers that assemble their own network. */
/* out[M,N] = epilogue(A[M,K] W[N,K]^T + bias): relu, +residual, *rowmask.  K % 8 == 0, lda/ldw in elements.
 * W is fp32 (precision F32) or the half-precision operand type (F16 / BF16, as produced by fdipt_model_prepare). */
int fdipt_linear(int precision, int M, int N, int K, const float* A, int lda, const void* W, int ldw, const float* bias,
                 const float* residual, int ldr, const float* rowmask, int relu, float* out, int ldo, fdipt_stream_t s);
int fdipt_layernorm(int M, int D, const float* x, const float* residual, const float* gamma, const float* beta,
                    const float* rowmask, float* out, fdipt_stream_t s);
/* self-test of the MFMA fragment maps used by every kernel: returns max |err| of a 64x64x64 product vs fp64 host math
 * through *max_err_host (host pointer). */
int fdipt_selftest_mfma(int precision, double* max_err_host);

/* HIP event helpers for bench.py (kernel timing on the stream the kernels are launched on). */
int fdipt_event_create(void** ev_host);
int fdipt_event_destroy(void* ev);
int fdipt_event_record(void* ev, fdipt_stream_t s);
int fdipt_event_elapsed_ms(void* start, void* stop, float* ms_host); /* synchronises on `stop` */

const char* fdipt_version(void);
/* Lengths N at which the half-precision forward switches kernel variants, ascending, into bounds_host[capacity] (host); returns their
 * number.  Two samples whose lengths (rounded up to 4) fall between the same bounds run the same kernels: padding inside a class does
 * not change a sample's bits (framedipt_amd/sharding.py: kernel_class). */
int fdipt_kernel_class_bounds(int32_t* bounds_host, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* FDIPT_H */
