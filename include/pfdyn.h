/* pfdyn.h -- C ABI of libpfdyn.so: the MI355X (gfx950) implementation of PharmacoForge's
 * per-timestep denoising network and reverse-diffusion loop.
 *
 * Boundary replaced (reference = eflynn8/pharmacophore-diffusion, paths relative to its root):
 *   pharmacoforge/models/dynamics_gvp.py:131-185   PharmRecDynamicsGVP.forward   -> pf_dynamics_forward
 *   pharmacoforge/models/dynamics_gvp.py:187-246   add/remove_pharm_edges        -> inside pf_dynamics_forward
 *                                                                                  (pf_debug_get_edges exposes them)
 *   pharmacoforge/models/pharmacodiff.py:380-431   sample_p_zs_given_zt          -> pf_denoise_step
 *   pharmacoforge/models/pharmacodiff.py:433-488   sample_given_receptor (loop)  -> pf_sample
 *   pharmacoforge/models/pharmacodiff.py:88-108    com_removal                   -> inside pf_denoise_step / pf_sample
 *   pharmacoforge/dataset/protein_pharm_dataset.py:234-236  static pp radius graph -> pf_build_pp_edges
 *   checkpoint key layout                                   -> pf_set_weight (names are the
 *                                                                                  reference state_dict keys)
 * The reference has no FFI of its own: the seam is a Python method call, so these entry points are
 * what a ctypes binding added to the reference would bind.
 *
 * Conventions
 *   - every function returns 0 on success or a negative pf_status; pf_last_error(h) gives text;
 *     no exceptions cross the ABI;
 *   - the caller owns every tensor it passes (PyTorch allocations as raw device pointers + sizes);
 *     the library owns its packed weights and a workspace sized by pf_set_pocket_batch;
 *   - "dev" pointers are device memory of the current HIP device, "host" pointers host memory;
 *     all tensors are dense row-major fp32 (indices int32);
 *   - all work is ordered on the caller's stream: when the stream reaches the end of a call's work,
 *     every output of the call is complete.  Two calls also use streams of the handle, joined to the
 *     caller's stream by events -- pf_set_pocket_batch uploads its tables on a copy stream (the upload
 *     of the next batch runs under the kernels of the current one), pf_train_backward builds its work
 *     lists and sums finished gradient copies on a side stream.  No call synchronises the device
 *     except pf_create / pf_commit_weights / pf_set_pocket_batch when an allocation has to grow, and
 *     the pf_debug_* readers;
 *   - a handle is bound to one device and is not thread-safe (one handle per GPU / process).
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     PF_ERR_HIP.
 */
#ifndef PFDYN_H
#define PFDYN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 1

typedef struct pf_handle pf_handle;
typedef void* pf_stream;           /* hipStream_t */

typedef enum pf_status {
    PF_OK = 0,
    PF_ERR_ARG = -1,           /* bad argument / unsupported configuration */
    PF_ERR_HIP = -2,           /* HIP runtime error (no device, OOM, launch failure) */
    PF_ERR_STATE = -3,         /* call order violated (weights not committed, no batch set ...) */
    PF_ERR_WEIGHT = -4,        /* unknown / missing / mis-shaped weight tensor */
    PF_ERR_EXCHANGE = -5       /* pf_sample_status: a workgroup hand-over inside a launch timed out; the run is invalid, repeat it */
} pf_status;

/* message_norm of GVPMultiEdgeConv (gvp.py:351,373-389) */
#define PF_NORM_MEAN 0         /* 'mean': mean reducer, divide by 1 */
#define PF_NORM_VALUE 1        /* number > 0: sum reducer, divide by message_norm_value */
#define PF_NORM_GRAPH 2        /* number == 0: sum reducer, divide by (edges into ntype)/(nodes of ntype)+1 per graph */

/* Mirrors PharmRecDynamicsGVP.__init__ (dynamics_gvp.py:96-97) + graph_cutoffs (configs/dev.yml:67-68). */
typedef struct pf_config {
    int32_t abi_version;        /* = PF_ABI_VERSION */
    int32_t pharm_nf;           /* n_pharm_scalars (6) */
    int32_t rec_nf;             /* n_prot_scalars  (11) */
    int32_t vector_size;        /* must be 16 */
    int32_t n_hidden_scalars;   /* must be 128 */
    int32_t n_convs;
    int32_t n_message_gvps;
    int32_t n_update_gvps;
    int32_t n_noise_gvps;
    int32_t message_norm_mode;  /* PF_NORM_* */
    float   message_norm_value;
    int32_t ff_k;               /* 0 = radius graph with cutoff_ff (max 200 neighbours) */
    int32_t pf_k;               /* 0 = radius with cutoff_pf (max 100 neighbours) */
    float   cutoff_pp, cutoff_pf, cutoff_fp, cutoff_ff;
    float   rbf_dmax;           /* 15 (gvp.py:350) */
    int32_t rbf_dim;            /* must be 16 */
} pf_config;

/* per-step scalars of sample_p_zs_given_zt (pharmacodiff.py:387-420), fp32, computed by the host */
typedef struct pf_step_coef {
    float t;                    /* (s+1)/T : the timestep fed to the dynamics            (:470) */
    float alpha_t_given_s;      /*                                                        (:156) */
    float var_terms;            /* sigma2_t_given_s / alpha_t_given_s / sigma_t           (:397) */
    float sigma;                /* sigma_t_given_s * sigma_s / sigma_t                    (:400) */
    float ep_zt;                /* endpoint param: alpha_t_given_s*sigma_s^2/sigma_t^2    (:414) */
    float ep_pred;              /* endpoint param: alpha_s*sigma2_t_given_s/sigma_t^2     (:414) */
} pf_step_coef;

const char* pf_version(void);
const char* pf_last_error(const pf_handle* h);     /* h may be NULL: last error of pf_create */

/* -- lifetime ------------------------------------------------------------------------------ */
int  pf_create(const pf_config* cfg, pf_handle** out);
void pf_destroy(pf_handle* h);

/* -- weights: one call per tensor of the reference state_dict under "dynamics." -------------
 * name: the reference key, e.g. "dynamics.noise_predictor.conv_layers.0.edge_message_fns.prot_pp_prot.0.Wh"
 * data: host fp32, row-major, shape[ndim].  Zero-size tensors (dropout dummy_param) and
 * "gamma.gamma" are accepted and ignored.  pf_commit_weights fails if any tensor is missing. */
int pf_set_weight(pf_handle* h, const char* name, const float* host_data, int32_t ndim, const int64_t* shape);
int pf_commit_weights(pf_handle* h);

/* -- static per-batch data (the DGL-free batch container) -----------------------------------
 * B graphs; graph g owns prot atoms [prot_ptr[g], prot_ptr[g+1]) and pharmacophore centers
 * [pharm_ptr[g], pharm_ptr[g+1]).  pp edges (static prot->prot, batch-local prot ids) may be in
 * any order (destination-sorted lists inside their graphs take a vectorised pass; anything else a scalar one that also
 * reports what is wrong).  prot_x / prot_h are copied into the handle's memory.  Asynchronous: the host tables are built in
 * pinned memory owned by the handle and uploaded with one copy (on the handle's copy stream, ordered before everything the
 * caller enqueues on `stream` afterwards); the host arrays may be freed on return.  The call synchronises the device only
 * when the workspace or a table buffer has to grow. */
int pf_set_pocket_batch(pf_handle* h, int32_t B, const int32_t* host_prot_ptr, const int32_t* host_pharm_ptr,
                        const float* dev_prot_x, const float* dev_prot_h,
                        int64_t n_pp, const int32_t* host_pp_src, const int32_t* host_pp_dst, pf_stream stream);

/* The same bind for callers whose pocket data lives on the HOST (the sampling drivers batch pockets on the host):
 * prot_x [Np,3] / prot_h [Np,rec_nf] are host fp32 arrays; they travel in the same single staged upload as the index
 * tables and the one-hot check runs on the host copy, so the call never waits for the device. */
int pf_set_pocket_batch_host(pf_handle* h, int32_t B, const int32_t* host_prot_ptr, const int32_t* host_pharm_ptr,
                             const float* host_prot_x, const float* host_prot_h,
                             int64_t n_pp, const int32_t* host_pp_src, const int32_t* host_pp_dst, pf_stream stream);

/* Optional, BEFORE the next pf_set_pocket_batch / _host call: host_rep[g] = index of the graph that represents graph g's
 * pocket (host_rep[r] == r for a representative).  The caller claims that graph g is a COPY of graph host_rep[g] -- same
 * atoms, features, coordinates and pp edges -- which is how every sampling batch of the reference is built
 * (copy_graph, utils/unorganized_utils.py:28-81; pharmacodiff.py:541-545; generate_pharmacophores.py:329-333).  The bind
 * verifies the claim (everything it can see on the host) and fails with PF_ERR_ARG if it does not hold.  Effect: during
 * sampling, copies at the same timestep share conv layer 0's protein->protein messages (computed once per pocket
 * instead of once per copy, gvp.py:545-549 being a pure function of pocket geometry, element types and t there);
 * outputs are the same up to fp32 summation order.  The groups apply to one bind only -- the next bind call consumes the
 * claim whether it succeeds or is rejected (argument checks included).  What "verifies" covers: ptr arrays and pp edges
 * always (they are host arrays), coordinates and features on the host for pf_set_pocket_batch_host.  With pf_set_pocket_batch
 * (DEVICE coordinates / features) the rows are compared on the device by a small launch of the bind (bit patterns, every atom
 * against the same atom of its representative); the verdict comes back through pinned memory and the first call that would
 * share messages (pf_dynamics_forward / pf_denoise_step / pf_sample of that batch) reads it and fails with PF_ERR_ARG if the
 * claim is false -- nothing is ever computed on a false claim. */
int pf_set_pocket_groups(pf_handle* h, int32_t B, const int32_t* host_rep);

/* Optional, right after pf_set_pocket_batch: the caller states whether every row of prot_h is an element one-hot
 * (what the reference's dataset / CLI always produce: protein_pharm_dataset.py:129, generate_pharmacophores.py:105-118).
 * pf_set_pocket_batch checks this on the device and the first inference call that wants the answer waits for it; a caller
 * that already knows (it built the rows, or checked them on the host) says so here and no call ever waits: the whole
 * bind is then asynchronous on `stream`.  Declaring 1 for rows that are not one-hots gives wrong results. */
int pf_declare_onehot_features(pf_handle* h, int32_t is_onehot);

/* radius_graph(prot, r=cutoff_pp, max_num_neighbors) per graph on the device; results on the host.
 * Call with host_src == NULL to get the edge count (return value >= 0), then again with buffers. */
int64_t pf_build_pp_edges(pf_handle* h, int32_t B, const int32_t* host_prot_ptr, const float* dev_prot_x,
                          int32_t max_num_neighbors, int32_t* host_src, int32_t* host_dst, int64_t capacity,
                          pf_stream stream);

/* -- the boundary function: (eps_h, eps_x) = dynamics(g, t) ---------------------------------
 * dev_prot_x may be NULL (use the coordinates currently held by the handle).
 * Pure with respect to the handle's sampling state. */
int pf_dynamics_forward(pf_handle* h, const float* dev_prot_x, const float* dev_pharm_x, const float* dev_pharm_h,
                        const float* dev_t /*[B]*/, float* dev_eps_h /*[Nf,pharm_nf]*/, float* dev_eps_x /*[Nf,3]*/,
                        pf_stream stream);

/* -- sampling state machine (sample_given_receptor, pharmacodiff.py:433-488) ----------------
 * begin : prot -= init_pharm_com[graph] (NULL: protein COM); x_t,h_t := init noise
 * step  : one sample_p_zs_given_zt with injected noise (x columns before h columns)
 * end   : remove protein COM, add initial protein COM, multiply h by feat_norm_constant */
int pf_sample_begin(pf_handle* h, const float* dev_init_pharm_com /*[B,3] or NULL*/,
                    const float* dev_noise0 /*[Nf,3+pharm_nf]*/, pf_stream stream);
int pf_denoise_step(pf_handle* h, const pf_step_coef* coef, const float* dev_noise /*[Nf,3+pharm_nf]*/,
                    int32_t endpoint_param_coord, int32_t endpoint_param_feat, pf_stream stream);
int pf_sample_end(pf_handle* h, float feat_norm_constant, float* dev_x0 /*[Nf,3]*/, float* dev_h0 /*[Nf,pharm_nf]*/,
                  pf_stream stream);
/* Validity of the sampling runs that ended (pf_sample_end / pf_sample) since the last call.  For small batches a step's last
 * launch hands the noise prediction from its node + head workgroups to its update + build workgroups through polled exchange
 * words (k_rg_node_hs_build); the poll is bounded, so a hand-over that never arrives (never observed: producers do not wait and
 * the whole grid is resident) ends the launch on zeros instead of hanging the device -- and the run is INVALID.  The count of
 * such time-outs is copied to pinned host memory behind the results of pf_sample_end, on the caller's stream: call this AFTER
 * waiting for x_0 / h_0 (stream or event synchronisation -- the call itself touches no device and never blocks) and BEFORE
 * using them.  Returns PF_OK, or PF_ERR_EXCHANGE with *n_timeouts (may be NULL) = the new time-outs; the handle then runs the
 * separate launches (as with PFDYN_HS_BUILD=0), so repeating the run on it is safe.  A caller that never asks is told by the
 * next pf_sample_begin on the handle, which fails with the same code.  The reference has nothing to correspond: its
 * sample_given_receptor (pharmacodiff.py:433-514) is host-sequenced. */
int pf_sample_status(pf_handle* h, int32_t* n_timeouts);
/* current frame in the caller's frame of reference (get_pos_feat_for_visual, pharmacodiff.py:360-378) */
int pf_sample_frame(pf_handle* h, float feat_norm_constant, float* dev_x /*[Nf,3]*/, float* dev_h /*[Nf,pharm_nf]*/,
                    pf_stream stream);
/* Optional, after pf_set_pocket_batch and before a loop of pf_denoise_step calls: the timesteps (pf_step_coef::t) the loop will visit.  The first
 * conv layer's protein-side messages depend on t only through one encoder output per element type
 * (dynamics_gvp.py:107-117 feeding gvp.py:545-549); their tables are computed here in one launch per 64 timesteps
 * instead of one small launch in front of every step.  pf_sample does this itself; results never depend on it. */
int pf_prepare_timesteps(pf_handle* h, const float* host_t, int32_t n, pf_stream stream);
/* whole loop: begin + n_steps x step + end, all enqueued on `stream` without host synchronisation.
 * host_coef[i] is the i-th iteration's coefficients (s = T-1-i); dev_noise is [n_steps+1, Nf, 3+pharm_nf].
 * dev_traj_x / dev_traj_h (optional) receive n_steps+1 frames. */
int pf_sample(pf_handle* h, int32_t n_steps, const pf_step_coef* host_coef, const float* dev_noise,
              const float* dev_init_pharm_com, int32_t endpoint_param_coord, int32_t endpoint_param_feat,
              float feat_norm_constant, float* dev_x0, float* dev_h0, float* dev_traj_x, float* dev_traj_h,
              pf_stream stream);

/* -- training step: gradients of the dynamics (autograd through PharmRecDynamicsGVP.forward, ------
 *    dynamics_gvp.py:131-185, as used by PharmacophoreDiff.forward / training_step, pharmacodiff.py:162-276)
 * Parameters and gradients are exchanged as ONE flat fp32 vector in the reference's state-dict order
 * (pf_param_layout enumerates name / offset / numel; a tensor's gradient sits at the offset of the tensor).
 * pf_train_forward  = the boundary function in train() mode: same outputs as pf_dynamics_forward plus GVPDropout
 *                     (gvp.py:118-149, rate dropout_p, counter-based masks from `seed`) at the two call sites of every
 *                     conv layer; keeps each layer's input state for the backward pass.
 * pf_train_backward = d(loss)/d(parameters) given d(loss)/d(eps_h), d(loss)/d(eps_x) of that forward.  The loss itself
 *                     (pharmacodiff.py:208-232) is elementwise on eps and is the caller's.
 * The optimiser is the caller's too (optim.Adam, pharmacodiff.py:253): after a step, push the new values with
 * pf_set_weight + pf_commit_weights. */
int pf_param_count(pf_handle* h, int64_t* n_params, int32_t* n_tensors);
int pf_param_layout(pf_handle* h, int32_t index, const char** name, int64_t* offset, int64_t* numel);
int pf_train_forward(pf_handle* h, const float* dev_prot_x /*[Np,3] or NULL*/, const float* dev_pharm_x /*[Nf,3]*/,
                     const float* dev_pharm_h /*[Nf,pharm_nf]*/, const float* dev_t /*[B]*/, float dropout_p, uint32_t seed,
                     float* dev_eps_h, float* dev_eps_x, pf_stream stream);
int pf_train_backward(pf_handle* h, const float* dev_g_eps_h /*[Nf,pharm_nf]*/, const float* dev_g_eps_x /*[Nf,3]*/,
                      float* dev_grad /*[n_params]*/, pf_stream stream);
/* The loss around the dynamics, fused (PharmacophoreDiff.forward, pharmacodiff.py:162-243, with the noise parameterisation
 * of both outputs -- endpoint_param_feat / endpoint_param_coord false): per graph the COM of the clean centers is taken
 * off the centers and the bound pocket (:176-183), z_t = alpha_t x0 + sigma_t eps with alpha / sigma read from the caller's
 * tables at t_int (:186-197: dev_alpha[k] = alpha(gamma(k / T)), likewise sigma), the COM of the noised centers is removed
 * when remove_com (:199-205), the dynamics run in train() mode (pf_train_forward), and dev_out receives
 *   [0] pos loss  [1] feat loss  (:208-232, weighted by 1 - t when weighted_loss)
 *   [2] position error  [3] weighted position error  [4] accuracy  [5] weighted accuracy  (:234-241)
 *   [6] total loss = [0] + [1]  [7] total error = [2] + 1 - [4]  [8] weighted total error = [3] + 1 - [5]  (what
 *       training_step / validation_step derive, :274-277, :303-306).
 * dev_pharm_h0 are the raw feature one-hots (divided by feat_norm inside).  The protein coordinates are the bound ones.
 * pf_train_loss_backward(g_pos, g_feat) = d(g_pos * pos loss + g_feat * feat loss)/d(parameters), the two upstream
 * scalars read from device memory; it consumes the state of the forward (one backward per forward).
 * pf_train_loss_backward_out takes the upstream gradient of all nine outputs instead (what autograd hands the node that
 * produced dev_out): g_pos = g_out[0] + g_out[6], g_feat = g_out[1] + g_out[6]; the metrics' entries are ignored. */
int pf_train_loss_forward(pf_handle* h, const float* dev_pharm_x0 /*[Nf,3]*/, const float* dev_pharm_h0 /*[Nf,pharm_nf]*/,
                          const int32_t* dev_t_int /*[B]*/, const float* dev_eps_x /*[Nf,3]*/, const float* dev_eps_h /*[Nf,pharm_nf]*/,
                          const float* dev_alpha /*[>= max t_int + 1]*/, const float* dev_sigma, int32_t n_timesteps, float feat_norm,
                          int32_t remove_com, int32_t weighted_loss, float dropout_p, uint32_t seed, float* dev_out /*[9]*/,
                          pf_stream stream);
int pf_train_loss_backward(pf_handle* h, const float* dev_g_pos /*[1]*/, const float* dev_g_feat /*[1]*/,
                           float* dev_grad /*[n_params]*/, pf_stream stream);
int pf_train_loss_backward_out(pf_handle* h, const float* dev_g_out /*[9]*/, float* dev_grad /*[n_params]*/, pf_stream stream);
/* the flat parameter vector on the device: set = copy in + refresh the packed MFMA-fragment weights by a device gather
 * (what an optimiser step calls instead of 245 x pf_set_weight + pf_commit_weights); get = copy out */
int pf_set_flat_params(pf_handle* h, const float* dev_flat /*[n_params]*/, pf_stream stream);
int pf_get_flat_params(pf_handle* h, float* dev_flat /*[n_params]*/, pf_stream stream);
/* one fused Adam step (torch.optim.Adam semantics: L2 weight decay into the gradient, bias correction, no amsgrad;
 * pharmacodiff.py:253) on the caller's flat vectors [n_params], followed by pf_set_flat_params(dev_params).
 * step counts from 1. */
int pf_adam_step(pf_handle* h, float* dev_params, const float* dev_grad, float* dev_exp_avg, float* dev_exp_avg_sq,
                 int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, pf_stream stream);
/* Arithmetic of the training step's dense Linears.  The reference trains in fp32 (pharmacodiff.py:162-263; gvp.py:96,101 force
 * .float()) and PF_TRAIN_F32 -- the default, and the only mode the parity tests against the reference's gradients use -- does
 * the same.  PF_TRAIN_BF16 (BASELINE config 5's "bf16" leg; no reference counterpart) runs to_feats_out and
 * scalar_to_vector_gates of the message chains' forward and the corresponding products of every gradient kernel (input and
 * weight gradients of to_feats_out; in the message chains also of the gates) on bf16 matrix instructions: operands rounded to
 * nearest even, fp32 accumulation.  Master weights, saved activations, LayerNorm, the vector channel, scatter sums, the loss
 * and the optimiser stay fp32.  Contract (tests/test_gpu_train.py): per-tensor gradient cosine >= 0.999 against the fp32
 * path.  Applies to the following pf_train_* calls; a forward of one precision cannot be followed by a backward of the other. */
#define PF_TRAIN_F32 0
#define PF_TRAIN_BF16 1
int pf_train_set_precision(pf_handle* h, int32_t precision);
int pf_train_get_precision(pf_handle* h, int32_t* precision);
/* tests: make the following pf_train_forward / pf_train_backward calls on this batch use the given multipliers
 * [n_convs][2][N][144] (layout of pf_debug_dropout_mask) instead of the built-in generator; NULL restores it.  The
 * buffer must stay alive until the backward call has finished. */
int pf_debug_set_dropout_masks(pf_handle* h, const float* dev_masks);
/* the {0, 1/(1-p)} multipliers pf_train_forward applies in conv layer `layer` (which: 0 message dropout, gvp.py:518;
 * 1 residual dropout, gvp.py:529): dev_out[node][144] = 128 scalar features then 16 vector channels, global node ids
 * (protein atoms first). */
int pf_debug_dropout_mask(pf_handle* h, int32_t layer, int32_t which, float dropout_p, uint32_t seed,
                          float* dev_out /*[N,144]*/, pf_stream stream);

/* -- introspection for tests / profiling ----------------------------------------------------- */
/* edges of the last dynamics call; etype: 0 ff, 1 pf, 2 fp, 3 pp; ids are ntype-local (reference
 * convention).  host_src == NULL returns the count.  Synchronises `stream`. */
int64_t pf_debug_get_edges(pf_handle* h, int32_t etype, int32_t* host_src, int32_t* host_dst, int64_t capacity,
                           pf_stream stream);
/* run one GVPMultiEdgeConv layer (gvp.py:459-538) on caller-provided node features, on the edges
 * built from the given coordinates: tests the conv kernels with non-zero vector inputs. */
int pf_debug_conv_layer(pf_handle* h, int32_t layer, const float* dev_prot_x, const float* dev_pharm_x,
                        const float* dev_h_prot /*[Np,128]*/, const float* dev_v_prot /*[Np,16,3]*/,
                        const float* dev_h_pharm, const float* dev_v_pharm,
                        float* dev_out_h_prot, float* dev_out_v_prot, float* dev_out_h_pharm, float* dev_out_v_pharm,
                        pf_stream stream);
/* per-kernel timing with HIP events recorded on the caller's stream around every launch of the
 * selected kernel classes (bit k of kernel_mask): 0 encode, 1 build_edges (0 and 1 share one launch unless
 * either is being timed), 2 edge_msg (one wave per tile), 3 node_update (one wave per tile), 4 noise_head,
 * 5 step_update, 6 edge_msg_coop, 7 node_update_coop (four waves per tile: launches with few tiles),
 * 8 edge_msg_coop of the last conv layer (when n_convs > 1).  pf_profile_read synchronises `stream`, returns the summed device
 * time [ms] and launch count per class since the last read, and resets the counters; pf_profile_enable only changes the
 * mask, so a caller can bracket a subset of its calls (a pair of event records costs ~10 us of stream time). */
#define PF_NUM_KERNEL_CLASSES 9
int pf_profile_enable(pf_handle* h, uint32_t kernel_mask);
int pf_profile_read(pf_handle* h, double* total_ms /*[9]*/, int64_t* launches /*[9]*/, pf_stream stream);
/* The same for the gradient kernels of pf_train_backward (mask bits 9..12 of pf_profile_enable): 0 noise-head backward,
 * 1 node-update backward, 2 edge-message backward (one launch per GVP level), 3 reserved. */
int pf_profile_read_train(pf_handle* h, double* total_ms /*[4]*/, int64_t* launches /*[4]*/, pf_stream stream);
/* work of the last dynamics call: `flops` / `bytes` = reference-equivalent (SURVEY.md 8(d) formulas on the actual
 * edge counts n_edges[4] = ff, pf, fp, pp, every layer dense); `executed_flops` / `executed_edges[n_convs]` = what the
 * kernels compute after dead-work elimination (last layer: pharm side only; layer before it: active atoms only). */
int pf_debug_work(pf_handle* h, double* flops, double* bytes, int64_t* n_edges /*[4]*/, double* executed_flops,
                  int64_t* executed_edges /*[n_convs]*/, pf_stream stream);
/* row / edge counts of the last dynamics call (or of the edges the last pf_denoise_step built): out[8] = ff, pf, fp, static pp edges,
 * "pa" edges (pp edges into active atoms: what the pruned conv layer computes of the pp etype), active atoms, centers, atoms --
 * what a per-launch FLOP count needs (bench.py: roofline.launches) */
int pf_debug_counts(pf_handle* h, int64_t* out /*[8]*/, pf_stream stream);
/* kernel family of the edge-message launch of conv layer `layer` in the last dynamics call (the launch policy depends
 * on the batch: pf_host.cpp LaunchPolicy): *rows_per_wave = 4 or 8 (row-group kernels, pf_rg.hip: k_rg_edge), 16 (16-row
 * items on the four waves of a workgroup, pf_n16.hip: k_n16_edge; 17 = the fused launch k_n16_fused, whose items also
 * compute conv layer 0's node update of their source rows), 32 (one wave per 32-row tile: k_edge_msg) or 128 (four waves per
 * 32-row tile: k_edge_msg_coop / coop2).  layer == n_convs asks about the launch behind the last conv layer's edge messages:
 * when the last call was the dynamics call of a pf_denoise_step whose node update + noise head + sampler update + edge build
 * ran as ONE tail launch (one workgroup per graph): 4 (pf_rg.hip: k_rg_tail, two two-wave items of four centers) or 16
 * (pf_n16.hip: k_n16_tail, PFDYN_TAIL_FORM=n16); 2 when the node + head items and the per-graph sampler update + edge build ran as
 * different workgroups of one launch (pf_rg.hip: k_rg_node_hs_build, the default for small batches; PFDYN_HS_BUILD=0 switches it
 * off); else 0 */
int pf_debug_kernel_family(pf_handle* h, int32_t layer, int32_t* rows_per_wave);
/* Work a denoising step's merged last launch did AHEAD for the next dynamics call (after a pf_denoise_step whose timestep plan was
 * announced; zeros otherwise).  out[0] = "pa" edges (pp edges into the active atoms) of conv layer 0 whose messages the next call will
 * NOT compute -- they were computed ahead and the update + build found their graphs' regions unchanged; out[1] = "pa" edges computed
 * ahead; out[2] = 1 if the centers' encoder outputs and h_src products were left in tables (center hoist), out[3] = centers covered.
 * Also: pf_debug_kernel_family(layer = n_convs + 1) = 1 when the LAST call's ff / fp items started from those tables, (n_convs + 2) = 1
 * when it skipped regions computed ahead.  Synchronises the stream. */
int pf_debug_ahead(pf_handle* h, int64_t* out /*[4]*/, pf_stream stream);
/* the exchange time-outs of k_rg_node_hs_build counted on this handle since it was created (cumulative; pf_sample_status is the
 * product-path check and reports new ones per run).  Synchronises the device. */
int pf_debug_xchg_timeouts(pf_handle* h, int32_t* n);
/* diagnostic: drop_word != 0 makes the producers of the merged launch skip one exchange word (word 0 of center 0), so that its
 * consumer times out; poll_max > 0 shortens the poll bound (default 65,536 rounds) so that a forced time-out costs microseconds.
 * (0, 0) restores the product behaviour.  tests/test_gpu_n16.py: the time-out must surface on the same run. */
int pf_debug_xchg_fault(pf_handle* h, int32_t drop_word, int32_t poll_max);
/* the noise prediction of the last dynamics call of a pf_denoise_step (what its sampler update consumed) */
int pf_debug_last_eps(pf_handle* h, float* dev_eps_h /*[Nf,pharm_nf] or NULL*/, float* dev_eps_x /*[Nf,3] or NULL*/, pf_stream stream);
/* static hoist of conv layer 0's protein-protein messages in the last dynamics call: *rows_per_wave = 0 (not used: training,
 * tile kernels, protein features that are not element one-hots, PFDYN_NO_L0_HOIST=1), 4 / 8 rows per hoisted wave (row-group
 * kernels: pp edges start at their second message GVP) or 16 (n16 kernels: pp AND pf edges start from a type-table row) */
int pf_debug_l0_hoist(pf_handle* h, int32_t* rows_per_wave);
/* One chain of the row-group kernels (pf_rg.hip: rg_gvp / rg_flush / rg_layernorm, 4 rows per wave) on caller-supplied rows,
 * with the committed weights -- the unit-level checker against the reference's own module outputs (gvp.py:89-116, 152-166;
 * dynamics_gvp.py:10-42).  All pointers are device memory; no batch needs to be bound.
 *   kind 0  message chain of conv `layer`, edge type `sub` (0 ff, 1 pf, 2 fp, 3 pp; gvp.py:545-549):
 *           s_in [n][144] = [h_src, rbf], v_in [n][17][3] = [x_hat, v_src]  ->  s_out [n][128], v_out [n][16][3]
 *   kind 1  update chain of conv `layer`, node type `sub` (0 prot, 1 pharm): [n][128], [n][16][3] -> same shapes
 *   kind 2  GVPLayerNorm of conv `layer`: sub = 2 * node type + (0 message_layer_norms, 1 update_layer_norms)
 *   kind 3  NoisePredictionBlock: s_in [n][128], v_in [n][16][3]  ->  s_out [n][pharm_nf] (eps_h), v_out [n][3] (eps_x)
 *   kind 16 / 17  the message / update chain (kinds 0 / 1, same rows) on the n16 kernels' chain code (pf_n16.hip: n16_block,
 *           16 rows per workgroup of four waves); needs n_message_gvps >= 2 */
int pf_debug_chain(pf_handle* h, int32_t kind, int32_t layer, int32_t sub, int32_t n_rows, const float* dev_s_in,
                   const float* dev_v_in, float* dev_s_out, float* dev_v_out, pf_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PFDYN_H */
