/*
 * cgsvmc.h -- C ABI of libcgsvmc_hip.so: the MI355X (gfx950) implementation of the
 * batched VMC inner loop of ClarkResearchGroup/cgs-vmc.
 *
 * The reference has no FFI: its boundary is the Python object API plus
 * tf.Session.run(op) on op handles.  Each entry point below is what a
 * ctypes binding for one of those op handles / object methods calls; the reference
 * interface it replaces is cited as file:line under /root/reference/cgs_vmc.
 *
 * Conventions: every function returns 0 on success or a negative vmc_status; the
 * message is available from vmc_last_error().  The caller owns all host buffers; the
 * library owns all device memory.  Calls are synchronous with respect to the host on
 * return unless stated otherwise (they are stream-ordered internally).  One vmc_ctx per
 * GPU; a ctx is not thread-safe; distinct ctxs are independent.
 *
 * Parameter vector order (wavefunctions.py:167-175, creation order of the snt.Linear
 * variables of FullyConnectedNetwork, wavefunctions.py:345-349):
 *   w_1[N,H] b_1[H] w_2[H,H] b_2[H] ... w_L[H,H] b_L[H] w_out[H,1] b_out[1],
 * every matrix row-major w[in][out]; P = N*H + H + (L-1)*(H*H + H) + H + 1 floats.
 * RestrictedBoltzmannNetwork (wavefunctions.py:391-452; Sonnet creates the variables when the
 * module is first connected, so the onsite layer of _build line 436 comes first):
 *   w_on[N,1] b_on[1] w_1[N,H] b_1[H] w_2[H,H] b_2[H] ... w_{L+1}[H,H] b_{L+1}[H],
 * P = N + 1 + N*H + H + L*(H*H + H) floats (L = num_fc_layers >= 0 relu layers).
 * Conv2DNetwork (wavefunctions.py:531-615) and ResNet2D (wavefunctions.py:710-809): one snt.Conv2D
 * per layers.Conv2dPeriodic in connection order -- conv_2d: num_conv_layers of them; res_net_2d:
 * the initial convolution, then first_conv, second_conv of every ResBlock2d (layers.py:200-201) --
 * each w[k][k][in_channels][F] (row-major) followed by b[F]; in_channels = 1 for the first, F after;
 * P = k*k*F + F + (n_conv - 1)*(k*k*F*F + F).
 * Conv1DNetwork (wavefunctions.py:455-527) and ResNet1D (wavefunctions.py:618-707): the same with
 * snt.Conv1D variables w[k][in_channels][F], b[F] per layers.Conv1dPeriodic;
 * P = k*F + F + (n_conv - 1)*(k*F*F + F).
 */
#ifndef CGSVMC_H_
#define CGSVMC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vmc_ctx vmc_ctx;

typedef enum {
  VMC_OK = 0,
  VMC_ERR_INVALID = -1,     /* bad argument (ValueError in the reference API)      */
  VMC_ERR_UNSUPPORTED = -2, /* ansatz shape / activation without a HIP kernel       */
  VMC_ERR_HIP = -3,         /* HIP runtime failure                                  */
  VMC_ERR_STATE = -4        /* call order (e.g. accumulate before set_bonds)        */
} vmc_status;

/* which parameter set: psi is the trained wavefunction, omega its frozen supervisor
 * copy (copy.deepcopy(wavefunction), training.py:660). */
enum { VMC_PSI = 0, VMC_OMEGA = 1 };

/* accumulate / apply modes (training.GROUND_STATE_OPTIMIZERS, training.py:913-917) */
enum { VMC_MODE_ENERGY_GRADIENT = 0, VMC_MODE_LOG_OVERLAP_ITSWO = 1 };

/* wavefunctions.WAVEFUNCTION_TYPES with kernels (wavefunctions.py:1157-1170) */
enum { VMC_ANSATZ_FULLY_CONNECTED = 0, VMC_ANSATZ_RBM = 1, VMC_ANSATZ_CONV_2D = 2,
       VMC_ANSATZ_RES_NET_2D = 3, VMC_ANSATZ_CONV_1D = 4, VMC_ANSATZ_RES_NET_1D = 5 };

/* layers.NONLINEARITIES ids (layers.py:13-21).  Every id is accepted as hidden and as output
 * activation of every ansatz type with kernels. */
enum { VMC_ACT_RELU = 0, VMC_ACT_EXP = 1, VMC_ACT_COS = 2, VMC_ACT_TAN = 3, VMC_ACT_TANH = 4,
       VMC_ACT_SIGMOID = 5, VMC_ACT_IDENTITY = 6 };

typedef struct {
  int32_t n_sites;           /* hparams.num_sites                (utils.py:98)       */
  int32_t batch_size;        /* chains owned by THIS ctx         (utils.py:135)      */
  int32_t num_layers;        /* hparams.num_fc_layers (utils.py:104); conv_2d: num_conv_layers
                                (108); res_net_2d: num_resnet_blocks (114)                     */
  int32_t layer_size;        /* hparams.fc_layer_size (utils.py:105): fully_connected and rbm take up
                                to 4096 units -- the fused kernels up to 512 (any nonlinearity),
                                beyond that the general multi-launch path
                                (materialised rows + GEMMs; two [131072][units] float
                                buffers per ctx: 4 GiB at 4096 units).  Convolutional ansatz types:
                                num_conv_filters (111): the fused kernels up to 64 (four 16-channel MFMA
                                blocks), the general path (feature maps in HBM, a GEMM per convolution) to 1024 */
  int32_t nonlinearity;      /* VMC_ACT_*: hparams.nonlinearity  (utils.py:128)      */
  int32_t output_activation; /* VMC_ACT_*: hparams.output_activation (utils.py:129)  */
  int32_t device;            /* HIP device ordinal                                   */
  int32_t chain_offset;      /* global id of local chain 0 (multi-GPU sharding)      */
  int32_t ansatz;            /* VMC_ANSATZ_*: hparams.wavefunction_type              */
  int32_t reserved;          /* 0                                                    */
  uint64_t seed;             /* Philox key                                           */
  void* stream;              /* hipStream_t to launch on, or NULL for the null stream*/
  /* convolutional ansatz types only (ignored otherwise) */
  int32_t kernel_size;       /* hparams.kernel_size (utils.py:110): fused kernels 1..9, general path to 31 */
  int32_t size_x, size_y;    /* hparams.size_x, size_y (utils.py:99-100); n_sites = size_x*size_y
                                (ignored by the 1-D types: the chain has n_sites sites)          */
  int32_t reserved2;         /* 0                                                    */
} vmc_desc;

/* Number of parameters P for a given shape (no ctx needed). */
int64_t vmc_num_params(int32_t n_sites, int32_t layer_size, int32_t num_layers);
int64_t vmc_num_params_ansatz(int32_t ansatz, int32_t n_sites, int32_t layer_size,
                              int32_t num_layers);
/* convolutional ansatz types: layer_size = num_conv_filters, num_layers as in vmc_desc */
int64_t vmc_num_params_conv(int32_t ansatz, int32_t num_layers, int32_t num_filters,
                            int32_t kernel_size);

/* FullyConnectedNetwork.__init__ + graph_builders.get_configs: allocates everything.
 * wavefunctions.py:331-353, graph_builders.py:92-125. */
int vmc_create(const vmc_desc* desc, vmc_ctx** out);
void vmc_destroy(vmc_ctx* ctx);
const char* vmc_last_error(const vmc_ctx* ctx); /* ctx may be NULL: last create error */

/* HeisenbergHamiltonian(bonds, j_x, j_z), operators.py:215-225.  j_x / j_z are
 * per-bond arrays of length n_bonds (constant arrays reproduce the reference). */
int vmc_set_bonds(vmc_ctx* ctx, int32_t n_bonds, const int32_t* ij /*[n_bonds][2]*/,
                  const float* j_x, const float* j_z);

/* Variable assignment / read-back (tf.train.Saver restore/save, run_training.py:134-146;
 * module_transfer_ops, wavefunctions.py:300-325). theta has P floats. */
int vmc_set_params(vmc_ctx* ctx, int which, const float* theta);
int vmc_get_params(vmc_ctx* ctx, int which, float* theta);
int vmc_transfer_params(vmc_ctx* ctx); /* psi -> omega: TrainOpsSWO.update_supervisor */

/* The CONFIGS variable, graph_builders.py:92-125: [batch_size][n_sites] float32 +-1. */
int vmc_set_configs(vmc_ctx* ctx, const float* configs);
int vmc_get_configs(vmc_ctx* ctx, float* configs);

/* exp_norm_shift (wavefunctions.py:206-232), per parameter set. */
int vmc_set_shift(vmc_ctx* ctx, int which, float shift);
int vmc_get_shift(vmc_ctx* ctx, int which, float* shift);

/* Wavefunction.__call__ (wavefunctions.py:355-371) on `configs` ([n_rows][n_sites],
 * host) or, when configs == NULL, on the ctx's chains (n_rows must be batch_size).
 * Writes the pre-exp logit and/or psi = exp(logit - shift); either may be NULL. */
int vmc_amplitude(vmc_ctx* ctx, int which, const float* configs, int64_t n_rows,
                  float* logit, float* psi);

/* n_steps x session.run(mc_step): graph_builders.py:38-89, driven by training.py:608-609
 * and evaluation.py:138-145.  One launch; chain state stays on the GPU.  *accepted (may be
 * NULL) receives the total acceptance_count (graph_builders.py:86). */
int vmc_mc_steps(vmc_ctx* ctx, int64_t n_steps, int64_t* accepted);

/* Test hook: one mc_step with externally supplied proposals: site to lower i_up, site to
 * raise i_dn, acceptance uniform u (all [batch_size]); accept_mask [batch_size] out. */
int vmc_mc_step_injected(vmc_ctx* ctx, const int32_t* i_up, const int32_t* i_dn,
                         const float* u, uint8_t* accept_mask);

/* Test hook: the proposals the sampler would draw at absolute step `step` for the
 * current chains (graph_builders.py:59-65), without moving. */
int vmc_debug_proposals(vmc_ctx* ctx, uint64_t step, int32_t* i_up, int32_t* i_dn, float* u);
/* Diagnostic: runs n_steps mc_steps through the s_memtime-stamped instantiation of the sweep
 * kernel and returns the mean shader cycles per step of up to 16 phases (see
 * engine.debug_sweep_profile for their names).  Never quote its run time. */
int vmc_debug_sweep_profile(vmc_ctx* ctx, int64_t n_steps, double* phase_cycles /*[16]*/);
int vmc_get_step_counter(vmc_ctx* ctx, uint64_t* step);
int vmc_set_step_counter(vmc_ctx* ctx, uint64_t step);

/* HeisenbergHamiltonian.local_value (operators.py:249-259) of parameter set `which` on
 * the ctx's chains: eloc [batch_size] (may be NULL), *mean = mean over the batch
 * (evaluation.py:102).  diag/offdiag_over_psi (may be NULL) are the two terms of
 * HeisenbergHamiltonian.build (operators.py:227-247) with the off-diagonal one already
 * divided by psi. */
int vmc_local_energy(vmc_ctx* ctx, int which, float* eloc, double* mean);
int vmc_local_energy_terms(vmc_ctx* ctx, int which, float* diag, float* offdiag_over_psi);

/* TrainOps.accumulate_gradients: training.py:539-558 (mode ENERGY_GRADIENT) or
 * training.py:661-695 (mode LOG_OVERLAP_ITSWO, beta = hparams.time_evolution_beta). */
int vmc_accumulate(vmc_ctx* ctx, int mode, float beta);
/* TrainOps.reset_gradients: tf.variables_initializer(tf.local_variables()). */
int vmc_reset_accumulators(vmc_ctx* ctx);
/* Accumulator buffer: [g1 (P) | g2 (P) | e_total e_count r_total r_count g_count 0 0 0]
 * = 2P+8 floats.  The device pointer is exposed so the host can all-reduce it in place
 * (RCCL via torch.distributed); *n_floats = 2P+8. */
int vmc_accumulators_devptr(vmc_ctx* ctx, void** dev_ptr, int64_t* n_floats);
/* The same all-reduce for hosts without torch.distributed: nccl_comm is an existing ncclComm_t of
 * RCCL (NULL or world_size <= 1: no-op); stream-ordered on the ctx's stream, g_count corrected.
 * librccl is resolved with dlopen at the first call. */
int vmc_allreduce_accumulators(vmc_ctx* ctx, void* nccl_comm, int32_t world_size);
int vmc_get_accumulators(vmc_ctx* ctx, float* host /*[2P+8]*/);
int vmc_set_accumulators(vmc_ctx* ctx, const float* host /*[2P+8]*/);

/* TrainOps.apply_gradients: gradient formula (training.py:560-564 or 697-699) + TF1
 * Adam (training.py:84-91).  *energy (may be NULL) = TrainOps.metrics / .energy. */
int vmc_apply_adam(vmc_ctx* ctx, int mode, float lr, float beta1, float beta2, float eps,
                   double* energy);
int vmc_get_gradient(vmc_ctx* ctx, int mode, float* grad /*[P]*/);
int vmc_mean_energy(vmc_ctx* ctx, double* energy);
int vmc_get_adam_state(vmc_ctx* ctx, float* m, float* v, int64_t* t);
int vmc_set_adam_state(vmc_ctx* ctx, const float* m, const float* v, int64_t t);

/* Whole-epoch entry points: the op sequences of run_optimization_epoch in
 * ONE host call, no Python round trip per op.
 * EnergyGradient (training.py:608-617): n_eq_steps mc_steps, update_norm (skipped when
 * max_value <= 0), reset, n_batches x [accumulate, n_mc_steps mc_steps].  apply_gradients is
 * left to the caller (the multi-GPU all-reduce sits in between).
 * LogOverlapITSWO (training.py:750-763): n_eq_steps mc_steps, update_norm, update_supervisor,
 * n_batches x [n_mc_steps mc_steps, reset, accumulate, Adam]; *energy = mean of the last batch.
 * Single-GPU only (Adam needs the reduced accumulators every batch). */
int vmc_epoch_energy_gradient(vmc_ctx* ctx, int64_t n_eq_steps, int32_t n_batches,
                              int64_t n_mc_steps, float max_value);
int vmc_epoch_log_overlap(vmc_ctx* ctx, float beta, int64_t n_eq_steps, int32_t n_batches,
                          int64_t n_mc_steps, float max_value, float lr, float beta1, float beta2,
                          float eps, double* energy);

/* ---- chains sharded over ranks (the reference is single-process) ----------------
 * Rank r owns chains [r B/G, (r+1) B/G) (vmc_desc.chain_offset); the only exchanges are a SUM
 * all-reduce of the accumulator buffer per optimizer step, a MAX all-reduce of one float for
 * update_norm and a SUM all-reduce of P+1 floats per CG iteration of the SR extension.  The
 * entries below issue them IN STREAM, between the kernels of the epoch, so that a multi-rank
 * epoch is one host call per rank exactly like the single-rank one.
 *
 * Transport: an RCCL communicator (`nccl_comm` = ncclComm_t; librccl is resolved with dlopen at
 * first use, the copy torch has already loaded if there is one) or, when nccl_comm == NULL and
 * world_size > 1, the device hook of vmc_set_device_allreduce (below) or the host hook registered
 * with vmc_set_host_allreduce: the library copies the buffer to pinned host memory, calls the hook
 * (which must all-reduce host_buf in place over all ranks: gloo, MPI, ...), and copies it back.  nccl_comm == NULL with world_size <= 1 is the
 * single-rank no-op; a 1-rank communicator still goes through ncclAllReduce. */
enum { VMC_REDUCE_SUM = 0, VMC_REDUCE_MAX = 1,
       VMC_REDUCE_SUM_F64 = 2 /* the buffer holds n DOUBLES (vmc_evaluate's batch means) */ };
typedef int (*vmc_host_allreduce_fn)(void* user, float* host_buf, int64_t n_elements, int32_t op);
int vmc_set_host_allreduce(vmc_ctx* ctx, vmc_host_allreduce_fn hook, void* user);
/* CONTRACT CHANGE (round 4 -> 5): op VMC_REDUCE_SUM_F64 hands the hook n_elements DOUBLES in host_buf
 * (cast the pointer).  A hook written for ops 0 / 1 would silently reduce half of them as floats, so the
 * library only passes op 2 to a host hook that has declared it: vmc_set_host_allreduce_caps(ctx,
 * VMC_HOST_REDUCE_CAP_F64) after registering it (registering a hook clears the caps).  Without the
 * declaration the entries that need it (vmc_evaluate with world_size > 1 on the host hook) return
 * VMC_ERR_UNSUPPORTED.  The device hook always receives op 2 (its contract has had it from the start). */
enum { VMC_HOST_REDUCE_CAP_F64 = 1 };
int vmc_set_host_allreduce_caps(vmc_ctx* ctx, int32_t caps);
/* Third transport, for hosts whose collective library keeps its communicator to itself but reduces
 * device memory in stream order (torch.distributed's ProcessGroupNCCL = RCCL on ROCm): with
 * nccl_comm == NULL and world_size > 1 the library calls this hook -- when one is registered it wins
 * over the host hook -- at the point of the epoch where the all-reduce belongs.  The hook must ENQUEUE
 * an in-place all-reduce of n_elements (float32; float64 for VMC_REDUCE_SUM_F64) at dev_buf that is
 * ordered after the work already on `stream` (the ctx's hipStream_t, NULL = the null stream) and
 * before whatever is enqueued on it afterwards; it need not wait for completion.  No staging copy,
 * no host synchronisation. */
typedef int (*vmc_device_allreduce_fn)(void* user, void* dev_buf, int64_t n_elements, int32_t op,
                                       void* stream);
int vmc_set_device_allreduce(vmc_ctx* ctx, vmc_device_allreduce_fn hook, void* user);
/* Communicator life cycle for hosts whose collective library does not expose its ncclComm_t
 * (torch.distributed): rank 0 draws the 128-byte ncclUniqueId and shares it by any channel, every
 * rank then calls vmc_rccl_comm_create (ncclCommInitRank on `device`). */
int vmc_rccl_unique_id(uint8_t id[128]);
int vmc_rccl_comm_create(const uint8_t id[128], int32_t world_size, int32_t rank, int32_t device,
                         void** nccl_comm);
int vmc_rccl_comm_destroy(void* nccl_comm);
const char* vmc_rccl_last_error(void);
/* File the library's nccl* entry points were resolved from: the librccl next to the libamdhip64 this
 * library is bound to (a process can hold two ROCm stacks -- torch bundles its own -- and streams,
 * buffers and communicators of one must not be handed to the other), else the first librccl.so.1 on
 * the loader path; "" when none was found.  A caller's ncclComm_t must come from THIS instance: a
 * communicator of another librccl in the process is rejected by ncclAllReduce as corrupted. */
const char* vmc_rccl_library_path(void);
/* PCI bus id ("0000:75:00.0") of HIP device `device`, and the file of the libamdhip64 behind this
 * library's hip* symbols -- both answered by THE runtime the library is bound to, so a caller that
 * wants to tell two ranks' devices apart never has to open a HIP runtime of its own by soname (a
 * second runtime in the process: see vmc_rccl_library_path).  Needs no ctx; `len` >= 16. */
int vmc_device_pci_bus_id(int32_t device, char* buf, int32_t len);
const char* vmc_hip_runtime_path(void);
/* Test / diagnostic hook: in-place all-reduce of n_floats of host data through the same transport
 * (staged through the ctx's device scratch for RCCL); op = VMC_REDUCE_*. */
int vmc_debug_allreduce(vmc_ctx* ctx, void* nccl_comm, int32_t world_size, float* host,
                        int64_t n_floats, int32_t op);
/* Wavefunction.update_norm with max_b psi taken over the chains of ALL ranks (logit domain). */
int vmc_update_norm_dist(vmc_ctx* ctx, void* nccl_comm, int32_t world_size, float max_value);
/* vmc_epoch_energy_gradient over sharded chains: update_norm sees the global maximum and the
 * accumulators are all-reduced (g_count corrected) before the call returns, ready for
 * vmc_apply_adam / vmc_sr_solve_dist on every rank. */
int vmc_epoch_energy_gradient_dist(vmc_ctx* ctx, void* nccl_comm, int32_t world_size,
                                   int64_t n_eq_steps, int32_t n_batches, int64_t n_mc_steps,
                                   float max_value);
/* vmc_epoch_log_overlap over sharded chains (training.py:750-763): per batch
 * [n_mc_steps mc_steps, reset, accumulate, all-reduce, Adam] with no host synchronisation;
 * every rank applies the identical Adam step.  With a 1-rank communicator the result is
 * bit-identical to vmc_epoch_log_overlap. */
int vmc_epoch_log_overlap_dist(vmc_ctx* ctx, void* nccl_comm, int32_t world_size, float beta,
                               int64_t n_eq_steps, int32_t n_batches, int64_t n_mc_steps,
                               float max_value, float lr, float beta1, float beta2, float eps,
                               double* energy);

/* MonteCarloOperatorEvaluator.run_evaluation (evaluation.py:113-152) in ONE host call: n_eq_steps
 * mc_steps, then n_samples x [batch mean of the Hamiltonian's local value (evaluation.py:102),
 * n_mc_steps mc_steps].  The batch sums stay on the device; with sharded chains (transport as
 * above) the n_samples per-rank means are SUM-all-reduced in float64 in one collective at the end
 * and divided by world_size, exactly what the op-by-op loop does per sample.  means[n_samples]
 * receives the list run_evaluation returns (as doubles; the Python side rounds to float32 like
 * tf.reduce_mean); *accepted (may be NULL) the acceptance count of THIS rank's chains over the
 * n_samples x n_mc_steps measurement steps (the count the reference computes and drops). */
int vmc_evaluate(vmc_ctx* ctx, void* nccl_comm, int32_t world_size, int64_t n_eq_steps,
                 int32_t n_samples, int64_t n_mc_steps, double* means, int64_t* accepted);

/* Stochastic reconfiguration -- EXTENSION: named by the north star, absent from the reference
 * (training.py has only the plain energy gradient + Adam), so these entries replace no reference
 * interface; they sit where TrainOpsTraditional.apply_gradients (training.py:560-567) sits.
 *   S = <O O^T> - <O><O>^T, f = <E O> - <E><O>, (S + diag_shift I) x = f, theta -= lr x
 * over every sample of the vmc_accumulate(ENERGY_GRADIENT) calls since the last reset.
 * vmc_sr_reserve(n) allocates the sample store for n accumulate calls (chains, activations,
 * back-propagated deltas: 2 L B Hp + B N floats each; convolutional types: the taped inputs and
 * deltas of every convolution, (2 n_conv - 1) B CS floats) and switches recording on; 0 frees it.
 * Covered: fully_connected and rbm (every width the library takes), the convolutional types; exp output.
 * Matrix-free conjugate gradients: vmc_sr_begin (x = 0, r = p = f from the accumulators, which
 * must already be all-reduced), then per iteration vmc_sr_matvec_partial (this rank's
 * sum_b (O_b . p) O_b into the P+1-float buffer of vmc_sr_buffer_devptr, last float =
 * sum_b O_b . p; SUM all-reduce it across ranks) and vmc_sr_cg_update (*rr = |r|^2 after the
 * step).  vmc_sr_solve runs the loop on one GPU until |r| <= tol |f| or max_iter.
 * vmc_sr_apply does theta -= lr x; *energy = TrainOps.metrics as in vmc_apply_adam. */
int vmc_sr_reserve(vmc_ctx* ctx, int32_t n_batches);
int vmc_sr_num_stored(vmc_ctx* ctx, int32_t* n);
int vmc_sr_begin(vmc_ctx* ctx, double* rr0);
int vmc_sr_matvec_partial(vmc_ctx* ctx);
/* The same matvec in two phases, for every path (required on the general convolution path, kernel path 6, whose
 * per-sample weights O_b . p are centred on their mean over ALL ranks; elsewhere phase 1 does nothing and phase 2 is
 * vmc_sr_matvec_partial): phase 1 leaves sum_b O_b . p of this rank's samples in the buffer's last float (the rest
 * zero); SUM all-reduce that float across ranks; phase 2 fills the buffer as vmc_sr_matvec_partial does.  Then the
 * all-reduce of the whole buffer and vmc_sr_cg_update as above.  (An extension like all of SR: no reference line.) */
int vmc_sr_matvec_phase1(vmc_ctx* ctx);
int vmc_sr_matvec_phase2(vmc_ctx* ctx);
int vmc_sr_buffer_devptr(vmc_ctx* ctx, void** dev_ptr, int64_t* n_floats);
int vmc_sr_get_buffer(vmc_ctx* ctx, float* host /*[P+1]*/);   /* host staging (non-RCCL backends) */
int vmc_sr_set_buffer(vmc_ctx* ctx, const float* host /*[P+1]*/);
int vmc_sr_cg_update(vmc_ctx* ctx, float diag_shift, double* rr);
int vmc_sr_solve(vmc_ctx* ctx, float diag_shift, float tol, int32_t max_iter, int32_t* iters,
                 double* rel_residual);
/* vmc_sr_solve with the stored samples sharded over ranks: the P+1-float all-reduce of every CG
 * iteration is issued in stream (transport as above); every rank runs the identical recurrence.
 * The accumulators must already be all-reduced. */
int vmc_sr_solve_dist(vmc_ctx* ctx, void* nccl_comm, int32_t world_size, float diag_shift, float tol,
                      int32_t max_iter, int32_t* iters, double* rel_residual);
int vmc_sr_get_solution(vmc_ctx* ctx, float* x /*[P]*/);
int vmc_sr_apply(vmc_ctx* ctx, float lr, double* energy);
/* test hook: out = (S + diag_shift I) v over the stored samples (single rank) */
int vmc_sr_debug_matvec(vmc_ctx* ctx, const float* v /*[P]*/, float diag_shift, float* out);

/* Wavefunction.update_norm (wavefunctions.py:261-288) on psi(chains). */
int vmc_update_norm(vmc_ctx* ctx, float max_value);

/* Per-kernel HIP-event timing on the ctx's stream (bench.py's roofline leg).  on = 1 times
 * every region, on = 2 only "sweep" and "tail_eloc" (each recorded event drains the pipeline
 * between two kernels, ~3.5 us: ten of them per step cost 1.8 % of the step they measure).
 * names: "sweep", "tail_eloc", "tail_amp", "z1", "bond_list", "eloc_reduce", "grad",
 * "adam", "sr_matvec".  ms = summed elapsed, launches = number of timed launches. */
int vmc_timing_enable(vmc_ctx* ctx, int on);
int vmc_timing_reset(vmc_ctx* ctx);
int vmc_timing_get(vmc_ctx* ctx, const char* name, double* ms, int64_t* launches);
/* rows the last local-energy call actually evaluated (antiparallel bonds) */
int vmc_last_connected_rows(vmc_ctx* ctx, int64_t* rows);
/* Diagnostic: which kernel family serves this ctx.  0: fused, register-resident rows (<= 256 hidden
 * units); 1: fused, LDS-operand rows (257 .. 512 units); 2: general multi-launch path (> 512 units,
 * or CGS_VMC_WIDE_FAST=0); 3: convolutional kernels; 4: the 3 x bf16 split experiment of the row kernel
 * (CGS_VMC_SPLIT_BF16=1: fully_connected, relu, 193 .. 256 units; fp32 results from the bf16 matrix cores,
 * cgs_vmc_amd/csrc/tail_split.hip -- never the headline configuration). */
int vmc_debug_kernel_path(vmc_ctx* ctx, int32_t* path);
/* Chains per workgroup of the fused dense sampler: 16 (k_sweep16, sweep16.hpp) or 8 (k_sweep8, sweep8.hip: chosen by
 * vmc_create when sixteen-chain tiles would occupy at most half of the CUs -- BASELINE configs 2 and 5 -- or by
 * CGS_VMC_SWEEP_TILE=8|16).  Both kernels implement graph_builders.py:38-89 and produce the same chains bit for bit.
 * set = 0 queries, 8 / 16 switches (test hook; 8 is refused where no k_sweep8 exists for the shape). */
int vmc_debug_sweep_tile(vmc_ctx* ctx, int32_t set, int32_t* chains);
/* Would vmc_mc_steps(n_steps) run the patch sampler of the general convolution path (k_cgen_patch_sweep,
 * cgs_vmc_amd/csrc/conv_patch.hip)?  It recomputes, per step, only the two boxes of every convolution that the exchanged
 * pair of graph_builders.py:67-71 reaches through layers.py:118-160's taps -- the four convolutional ansatz types at <= 16
 * filters on a lattice wider than the last box, launches of >= 8 steps; CGS_VMC_CONV_PATCH=0 never, =2 wherever the shape
 * allows -- and gives the chains of the full-forward sampler bit for bit.  *patch: 1 / 0.
 * The local energies' connected configurations (operators.py:162-169) run through the same kernel's second form.
 * vmc_create sends a convolutional shape that the fused kernels would take to the general path (vmc_debug_kernel_path 6)
 * when these kernels beat them: the boxes of all convolutions at most a fifth of the lattice x convolutions (e.g. 20 x 20
 * sites, 3 x 16 filters 3 x 3).  CGS_VMC_CONV_GENERAL=0 keeps the fused kernels there (their gradient and stochastic-
 * reconfiguration kernels are the faster ones: a solve over many stored samples may prefer them). */
int vmc_debug_conv_patch(vmc_ctx* ctx, int64_t n_steps, int32_t* patch);
int vmc_synchronize(vmc_ctx* ctx);

/* Test hook: C[M,N] = op(A) op(B) through the library's fp32 MFMA GEMM
 * (A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; host buffers). */
int vmc_debug_gemm(vmc_ctx* ctx, int32_t M, int32_t N, int32_t K, const float* A, int64_t sam,
                   int64_t sak, int64_t a_len, const float* B, int64_t sbk, int64_t sbn,
                   int64_t b_len, float* C);

#ifdef __cplusplus
}
#endif
#endif /* CGSVMC_H_ */
