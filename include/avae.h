/*
 * avae.h -- C ABI of libavae: the MI355X (gfx950) associative-VAE training path.
 *
 * This is the drop-in boundary of the repository.  The reference (navigator8972/vae_assoc) has
 * no FFI of its own: its boundary is the Python class AssocVariationalAutoEncoder whose methods
 * each end in one TensorFlow `sess.run` (reference vae_assoc.py:383,389,399-402,417-418,423-424).
 * Every entry point below replaces one of those `sess.run` calls (cited per function); the
 * Python class in vae_assoc_amd/vae_assoc.py keeps the reference's method surface and calls
 * these through ctypes.
 *
 * Conventions
 *  - plain C types only; device pointers are raw `float*` into HBM owned by the caller
 *    (PyTorch-ROCm tensors are used purely as containers on the Python side);
 *  - every function returns 0 on success, nonzero on failure; the message is available from
 *    avae_last_error(); no C++ exception crosses the ABI;
 *  - one handle = one model replica on one GPU; calls on one handle serialise on an internal
 *    mutex (the reference's callers use the session from a worker thread and the GUI thread,
 *    baxter_vae_assoc_writer.py:651-673); different handles are independent;
 *  - `stream` is a hipStream_t passed as void* (NULL = the HIP null stream).  Work is
 *    enqueued asynchronously; a call only synchronises when it has to hand a host value back
 *    (a non-NULL `cost_host`, get/set of parameters, save/load);
 *  - matrices are row-major; an input batch of modality m is [rows, n_input_m] float32 with a
 *    row stride (leading dimension, in floats) given by `x_ld[m]` (NULL = dense, ld = n_input).
 *
 * Flat parameter order (avae_get_params / avae_set_params / avae_get_grads / Adam state) is the
 * reference's variable-creation order, per modality (vae_assoc.py:185-215,257-300):
 *   enc W1[n_in,H1] b1 W2[H1,H2] b2 ... Wmu[HL,n_z] bmu Wsig[HL,n_z] bsig
 *   dec V1[n_z,H1] c1 V2[H1,H2] c2 ... Vout[HL,n_in] cout         each W row-major [in,out].
 */
#ifndef AVAE_H_
#define AVAE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVAE_ABI_VERSION 4
#define AVAE_MAX_MODALITIES 4
#define AVAE_MAX_HIDDEN 8

/* hidden-layer transfer function: reference `transfer_fct` (vae_assoc.py:26,48,502) */
enum { AVAE_ACT_IDENTITY = 0, AVAE_ACT_RELU = 1, AVAE_ACT_SOFTPLUS = 2, AVAE_ACT_SIGMOID = 3, AVAE_ACT_TANH = 4 };
/* arithmetic type of the GEMM operands (accumulation, losses, latent maths and Adam are fp32) */
enum { AVAE_F32 = 0, AVAE_BF16 = 1 };
/* who runs the gradient all-reduce (avae_config.use_comm) */
enum { AVAE_COMM_NONE = 0, AVAE_COMM_RCCL = 1, AVAE_COMM_IPC = 2 };
#define AVAE_IPC_HANDLE_BYTES 128
#define AVAE_MAX_WORLD 8

typedef struct avae_modality {
    int32_t n_input;                    /* reference network_architecture["n_input"] */
    int32_t n_hidden_layers;            /* 2 in the reference (n_hidden_recog_1/2)   */
    int32_t n_hidden[AVAE_MAX_HIDDEN];  /* encoder widths; the MLP decoder reuses them (vae_assoc.py:257,280,293) */
    int32_t binary;                     /* 1: Bernoulli recon + sigmoid output (:321-324,:293-297); 0: Gaussian (:327-328,:299-303) */
    float weight;                       /* reference `weights[m]` (:319,:340) */
    int32_t hidden_conv;                /* 1: conv encoder / deconv decoder branch (vae_assoc.py:169-210,249-291; deconv.py).
                                           Needs binary=1 and n_input=784 (the branch is hard-wired to 28x28 images);
                                           n_hidden[0..1] = n_hidden_recog_1/2 (conv depths), conv_gener = n_hidden_gener_1/2 */
    int32_t conv_gener[2];
    int32_t reserved;
} avae_modality;

typedef struct avae_config {
    int32_t abi_version;                /* AVAE_ABI_VERSION */
    int32_t n_modalities;
    avae_modality mod[AVAE_MAX_MODALITIES];
    int32_t n_z;                        /* taken from modality 0 in the reference (:89) */
    int32_t batch_size;                 /* rows per train/eval step on THIS replica (reference batch_size, :27,:90) */
    int32_t batch_global;               /* divisor of the mean terms; 0 -> batch_size.  world_size*batch_size under data parallelism */
    int32_t row_offset;                 /* global row index of local row 0 (rank*batch_size): keys the internal eps generator */
    int32_t activation;                 /* AVAE_ACT_* */
    int32_t compute_dtype;              /* AVAE_F32 | AVAE_BF16 */
    int32_t device;                     /* HIP device ordinal */
    int32_t use_graph;                  /* 1: replay the step as a captured hipGraph */
    float assoc_lambda;                 /* reference assoc_lambda (:29,:369) */
    float learning_rate;                /* reference learning_rate (:50,:374) */
    float beta1, beta2, adam_eps;       /* TF-1 AdamOptimizer defaults 0.9 / 0.999 / 1e-8 when all three are 0 */
    uint64_t seed;                      /* Philox key of the internal eps generator */
    void* workspace;                    /* optional caller-owned device memory (>= avae_workspace_bytes); NULL -> hipMalloc */
    size_t workspace_bytes;
    /* Library-owned gradient collective (the reference is single-process, vae_assoc.py:66).
     * use_comm = AVAE_COMM_RCCL: avae_create builds an RCCL communicator of world_size ranks from nccl_id (the 128 bytes of an
     *   ncclUniqueId made by avae_comm_unique_id on rank 0 and handed to every rank by whatever bootstrap the host has --
     *   torch.distributed here); the collective is ncclAllReduce on the library's comm stream.
     * use_comm = AVAE_COMM_IPC: the library's own one-shot all-reduce over hipIpc peers (push reduce-scatter + push all-gather,
     *   every xGMI link at once).  avae_create allocates the replica's exchange block; the host hands every
     *   rank's avae_comm_ipc_handle bytes round and calls avae_comm_ipc_attach before the first step.
     * Either way avae_train_step(s) run backward -> all-reduce -> Adam per bucket on the library's own streams, the decoder-side
     * bucket's all-reduce overlapping the encoder's backward pass, sixteen steps per captured hipGraph.
     * use_comm = AVAE_COMM_NONE: a host that owns the collective drives the same buckets through avae_stage_batches /
     * avae_dp_backward / avae_dp_apply.  batch_global / row_offset above stay the caller's to set (world_size*batch_size,
     * rank*batch_size). */
    int32_t use_comm;                   /* AVAE_COMM_* */
    int32_t world_size;
    int32_t rank;
    int32_t comm_buckets;               /* 0 or 2: two buckets (decoder side first); 1: ONE all-reduce of the whole gradient buffer */
    uint8_t nccl_id[128];
    int32_t wire_dtype;                 /* AVAE_F32 (default) | AVAE_BF16: gradient element type on the wire; the cost slot is always fp32 */
    int32_t reserved2[3];
} avae_config;

typedef struct avae_handle avae_handle;

/* Size of the single device allocation a replica needs (parameters, Adam state, compute-dtype
 * shadows, activations, gradients).  Lets the caller allocate it as a torch tensor. */
int avae_workspace_bytes(const avae_config* cfg, size_t* bytes);

/* Replaces AssocVariationalAutoEncoder.__init__ graph/session construction (vae_assoc.py:26-71).
 * Parameters start at zero: call avae_set_params (weights are injected explicitly because the
 * TF RNG stream of xavier_init, :11-18, cannot be reproduced). */
int avae_create(const avae_config* cfg, avae_handle** out);
void avae_destroy(avae_handle* h);
/* h may be NULL: returns the message of the last failed avae_create on this thread. */
const char* avae_last_error(const avae_handle* h);

int avae_param_count(const avae_handle* h, size_t* n);
int avae_get_params(avae_handle* h, float* host_dst);         /* flat order above, host memory */
int avae_set_params(avae_handle* h, const float* host_src);
int avae_get_grads(avae_handle* h, float* host_dst);          /* gradient of the last step, flat order (parity tests) */
/* Adam slots + step counter: what tf.train.Saver checkpoints besides the weights (vae_assoc.py:70,427-463). */
int avae_get_opt_state(avae_handle* h, float* host_m, float* host_v, int64_t* step);
int avae_set_opt_state(avae_handle* h, const float* host_m, const float* host_v, int64_t step);

/* partial_fit (vae_assoc.py:378-386): one sess.run((optimizer, cost)).
 *   x_dev[m]  device [batch_size, n_input_m] float32, row stride x_ld[m]
 *   eps_dev   device [batch_size, n_z] float32 shared by all modalities (:90), or NULL ->
 *             internal Philox4x32-10 normals keyed by (seed, step, global row)
 *   cost_host NULL -> fully asynchronous; else receives the cost of this step's forward pass
 *             (pre-update weights, as in the reference) after a stream synchronise. */
int avae_train_step(avae_handle* h, const float* const* x_dev, const int32_t* x_ld,
                    const float* eps_dev, float* cost_host, void* stream);
/* The inner batch loop of train() (vae_assoc.py:541-550 over DataSet.next_batch's consecutive slices,
 * dataset.py:22-43) as ONE submission: exactly n_steps successive avae_train_step calls, step i on rows
 * [i*batch_size, (i+1)*batch_size) of every x_dev[m] (row stride x_ld[m]) and of eps_dev (dense [.., n_z];
 * NULL -> internal generator).  Steps are replayed sixteen (then four) to a hipGraph whose first kernel stages
 * all of the replay's batches at once, so the host is out of the loop and the replay boundary and the staging
 * launch are paid once per sixteen steps.  cost_host (optional) receives the LAST step's cost; every step's
 * cost is in avae_cost_history. */
int avae_train_steps(avae_handle* h, int32_t n_steps, const float* const* x_dev, const int32_t* x_ld,
                     const float* eps_dev, float* cost_host, void* stream);
/* ---- the data-parallel seam (the reference has none: one tf.InteractiveSession, vae_assoc.py:66).  The gradient buffer is ONE
 * device array of avae_grad_buffer's n_floats: the local gradient in the master layout (encoder sides of every modality, then
 * decoder sides, pads zero) with the local cost in its last float.  It is cut into buckets of ONE contiguous range each:
 * bucket 0 = decoder sides + the cost slot, bucket 1 = encoder sides; comm_buckets = 1 and models with a conv modality have the
 * single bucket 0 = the whole buffer.  avae_dp_plan is host-only (no GPU needed): offs/counts hold n_buckets entries.
 * Host-owned collective:  avae_stage_batches(n) ; per step j:  avae_dp_backward(j, 0) -> SUM-all-reduce bucket 0's range ->
 * avae_dp_backward(j, 1) -> all-reduce bucket 1's range -> avae_dp_apply(0) -> avae_dp_apply(1). */
int avae_stage_batches(avae_handle* h, int32_t n_steps, const float* const* x_dev, const int32_t* x_ld,
                       const float* eps_dev, void* stream);
int avae_grad_buffer(avae_handle* h, float** dev_ptr, size_t* n_floats);
int avae_dp_plan(const avae_config* cfg, int32_t* n_buckets, int32_t* n_ranges, int64_t* offs, int64_t* counts);
int avae_dp_backward(avae_handle* h, int32_t j, int32_t bucket, void* stream);
int avae_dp_apply(avae_handle* h, int32_t bucket, float* cost_host, void* stream);
/* 128 bytes of a fresh ncclUniqueId (rank 0 calls this; every rank passes the same bytes in avae_config.nccl_id). */
int avae_comm_unique_id(void* id128);
/* AVAE_COMM_IPC bring-up: AVAE_IPC_HANDLE_BYTES describing this replica's exchange block (a hipIpcMemHandle_t + its size and
 * rank); the host gathers every rank's bytes IN RANK ORDER and hands all world_size * AVAE_IPC_HANDLE_BYTES to attach, which maps
 * the peers' blocks.  Both are collective in the sense that every rank must call them before any rank trains. */
int avae_comm_ipc_handle(avae_handle* h, void* handle_out);
int avae_comm_ipc_attach(avae_handle* h, const void* handles_by_rank);
/* Costs of the most recent `n` applied steps (oldest first), without having synchronised per step. */
int avae_cost_history(avae_handle* h, int32_t n, float* host_dst, int64_t* last_step);

/* evaluate_cost (vae_assoc.py:388-391): forward + loss, no update. */
int avae_eval_cost(avae_handle* h, const float* const* x_dev, const int32_t* x_ld,
                   const float* eps_dev, float* cost_host, void* stream);
/* transform (vae_assoc.py:393-403): posterior mean (and optionally log-variance) of modality m;
 * any row count. */
int avae_encode(avae_handle* h, int32_t m, const float* x_dev, int32_t x_ld, int32_t rows,
                float* mu_dev, float* logvar_dev, void* stream);
/* generate (vae_assoc.py:405-419): decoder of modality m on fed z [rows, n_z]; any row count. */
int avae_decode(avae_handle* h, int32_t m, const float* z_dev, int32_t rows, float* xhat_dev, void* stream);
/* generate() as the reference's callers use it -- every modality's decoder on the same z (vae_assoc.py:405-419 loops over the
 * modalities; baxter_vae_assoc_writer.py:141-147,259-304 and vae_assoc_model_viewer.py:107-113 call it 10-50 times per search
 * iteration with one live row).  xhat_dev[m] = device [rows, n_input_m] float32, dense.  For 1..64 rows (and for chunks of
 * batch_size rows) the call is one launch that carries the call's pointers by value (it stages z, runs the decoder's first layer of
 * every modality and publishes the pointers to the device) + one replay of a captured graph [remaining decoder launches of all
 * modalities, output launch storing straight into xhat_dev]; conv decoders go through avae_decode's path.  Calls on one handle
 * must be issued on one stream at a time (the published pointers belong to the latest call). */
int avae_generate(avae_handle* h, const float* z_dev, int32_t rows, float* const* xhat_dev, void* stream);
/* reconstruct (vae_assoc.py:421-425): encode -> z = mu + exp(lv/2)*eps -> decode for modality m.
 * eps_dev [rows, n_z] or NULL (internal generator, a fresh draw per call as in the reference). */
int avae_reconstruct(avae_handle* h, int32_t m, const float* x_dev, int32_t x_ld, const float* eps_dev,
                     int32_t rows, float* xhat_dev, void* stream);

/* save_model / restore_model (vae_assoc.py:427-463): own flat file (config echo + params + Adam
 * slots + step); TF .ckpt files cannot be read offline. */
int avae_save(avae_handle* h, const char* path);
int avae_load(avae_handle* h, const char* path);

/* ---- AVAE_INTROSPECTION: bench.py / tests only; no caller of the reference's surface needs anything below ---- */
int avae_synchronize(avae_handle* h);
/* Average device time (ms) of every launch of the step over the calls since the last reset,
 * measured with hipEvents on the stream the kernels were launched on (hipExtLaunchKernel start/stop
 * events: the dispatch's own begin and end, what rocprofv3 --kernel-trace reports).  Only recorded
 * while timing is enabled (it forces eager launches instead of graph replay); "_null_kernel" is a
 * one-store kernel timed the same way.  Report: one line "<name> <calls> <avg_ms> <min_ms>" per launch. */
int avae_timing_enable(avae_handle* h, int32_t on);
int avae_timing_report(avae_handle* h, char* buf, size_t buf_bytes);
/* The gradient collective of bucket `bucket` alone, on `stream` (micro-benchmarks and tests of the exchange; the train calls run it
 * on the library's comm stream inside the step).  Every rank must make the same sequence of calls. */
int avae_comm_allreduce(avae_handle* h, int32_t bucket, void* stream);
/* Copies a named internal tensor to the host as fp32 (tests): "mulv<m>" [batch,2*n_z], "eps" [batch,n_z], "E<m>_<k>" / "D<m>_<k>" the
 * stored output of encoder / decoder hidden layer k of modality m [batch, width] (the last forward pass's relu decisions);
 * "shadow_err" -> {max |W - theta|, max |W^T - theta|, layers checked, worst layer}: the compute-dtype weight shadows against the
 * parameters rounded once (must be 0, 0 after any call). */
int avae_debug_fetch(avae_handle* h, const char* name, float* host_dst, size_t max_floats, size_t* n_floats);

#ifdef __cplusplus
}
#endif
#endif /* AVAE_H_ */
