// Host side of libavae: memory plan, work-item tables, hipGraph capture and the C ABI
// (include/avae.h).  No per-step allocation: everything is carved once from one workspace.
#include "avae_device.h"
#include "../../include/avae.h"

#include <dlfcn.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

using namespace avae;

namespace {

thread_local std::string g_create_error;

struct Err : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIP_OK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw Err(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +    \
                      std::to_string(__LINE__) + ")");                                           \
    } while (0)

// The calling thread's current HIP device is put back on exit: a process may hold a model on device k while its own (torch)
// current device is j, and every ABI call switching the thread to k would silently move the caller's later allocations.
struct DeviceGuard {
    int prev = -1, want = -1;
    explicit DeviceGuard(int dev) : want(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) HIP_OK(hipSetDevice(want));
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// A launch rejected by the runtime (grid / LDS size, bad kernel arguments) must surface as this call's error, not later as a
// wrong number.  Also legal while a stream is being captured.
#define LAUNCH_OK(what)                                                                          \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) throw Err(std::string("launch of ") + (what) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int kServeRing = 16;         // call records in the serving path's pinned-host ring (= graphs per serving plan, one per slot)
constexpr int kMultiSteps = 16;         // most whole steps per replay of a multi-step graph (avae_train_steps) = staging sets
constexpr int kMultiSizes[2] = {16, 4};  // captured replay lengths: a run of n batches goes 16,16,...,4,4,...,1,1

inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ----------------------------------------------------------------------------- memory plan
struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = rup(off + bytes, 256); return o; }
};

struct Dense {             // one dense layer, weights + bias as W_aug [(in+1)][ld]
    int in = 0, out = 0;
    int ld = 0;            // master (theta, m, v, g) leading dim = rup(out, KU)
    int ldw = 0;           // W-shadow leading dim: ld, or ld + KU where ld would be a channel-aliasing stride (spread_ld)
    int ldt = 0;           // W^T-shadow leading dim = rup(in+1, KU)
    size_t master = 0;     // offset in floats inside the theta/m/v/g regions
    size_t W = 0, Wt = 0;  // byte offsets in the workspace
    bool head = false;     // [mu|sigma] fused head: flat layout is Wmu,bmu,Wsig,bsig
};

constexpr int kRowsumBlocks = 256;   // workgroups of the bias-gradient row sums (k_rowsum)

struct Act {               // activation or gradient: row-major [rows p][ld] (every consumer reads it as stored)
    int width = 0;
    int rows = 0;          // logical rows (batch, or batch * output pixels for a conv stage)
    int ld = 0;
    size_t rm = 0;
    bool ones = false;
};

// One conv-like stage of the conv/deconv branch: gather (im2col) -> GEMM on the patch matrix.
struct ConvStage {
    ConvGeom g{};          // for the training batch
    Dense d;               // in = k*k*Cin, out = Cout (W_aug: bias row only fed when `bias`)
    bool bias = false;
    int act = AVAE_ACT_IDENTITY;   // transfer function applied to this stage's output
    int flat = 0;          // flat-parameter form: 0 conv filter [k,k,ci,co] (no bias); 1 transposed-conv filter
                           // [k,k,co,ci] + bias[co]; 2 dense W[in,out] + b; 3 fused head Wmu,bmu,Wsig,bsig
    Act P;                 // patch matrix, rows = B*OH*OW, width = k*k*Cin
    Act Y;                 // stage output (row-major only), NHWC rows
    Act dY;                // gradient w.r.t. the stage's pre-activation output (row-major + transposed)
    size_t dP = 0;         // fp32 patch gradients [rows][lddp]
    int lddp = 0;
    int ksplit = 1, kchunk = 0;    // weight gradient: K = rows is cut into ksplit chunks of kchunk K tiles (see WorkItem)
    size_t part = 0;       // fp32 slices [ksplit][K+1][d.ld] of the split weight gradient
    bool adj = false;      // transposed conv whose input gradient is a GEMM on the patch matrix of its OUTPUT gradient
    Act Padj;              // that patch matrix, rows = B*IH*IW, width = k*k*Cout
    size_t Wadj = 0;       // adjoint filter shadow [Cin (padded)][ldadj]
    int ldadj = 0;
    size_t Wf = 0;         // its transpose [k*k*Cout (padded)][ldf]: B operand of the forward scatter product
    int ldf = 0;
    size_t T = 0;          // fp32 scatter product [B*IH*IW][ldT] = X . Wadj, overlap-added into Y by k_col2im (forward mode)
    int ldT = 0;
    size_t Gadj = 0;       // fp32 filter gradient in the adjoint frame [Cin][ldga] (+ aksplit slices at apart)
    int ldga = 0, aksplit = 1, akchunk = 0;
    size_t apart = 0;
    size_t rs_part = 0;    // bias gradient: partial column sums of dY [kRowsumBlocks][rs_cols4]
    int rs_cols4 = 0;
    // Implicit-GEMM routes of the training plan (ConvA, avae_device.h): the stage's patch matrix is never stored.
    bool impl = false;     // forward + filter gradient: A = implicit P of the stage's input (needs Cin * elem_size % 16 == 0)
    bool impl_w = false;   // filter gradient: dW = P^T . dY with P implicit (gather form on the stage's input); impl implies it (no stored P then)
    bool impl_bwd = false; // input gradient: A = implicit patch matrix of the stage's OUTPUT gradient with the adjoint geometry, B = Wadj
                           // (needs Cout * elem_size % 16 == 0 and an input that has a gradient)
    int cin_real = 0;      // channels of the stage's input in the flat parameter layout (the first decoder stage pads n_z up to a whole chunk)
    bool dense_map = false;// the direct one-channel 28x28 stage: its output / output gradient are stored as dense rows [B][784 (+1)],
                           // which makes the dense output layer behind it an ordinary dense layer (no flatten gather, no col2im)
    bool thin = false;     // one output channel: direct kernels (k_thin) instead of im2col -> GEMM -> col2im in the training plan
    size_t thin_part = 0;  // its filter-gradient partial sums [thin_blocks][Kp]
    int thin_blocks = 0, thin_kp = 0;
};

struct Mod {
    int n_in = 0, L = 0;
    std::vector<int> hs;
    std::vector<Dense> enc, dec;   // hidden layers
    Dense head, outl;
    Act X0, Z, dH, dO;
    std::vector<Act> E, D, dE, dD;
    size_t X32 = 0, mulv = 0, g0 = 0, out32 = 0;
    int ld32 = 0;
    bool conv = false;             // hidden_conv branch: cenc = E1,E2,E3,HEAD  cdec = T1,T2,T3,T4,OUT
    std::vector<ConvStage> cenc, cdec;
};

struct Launch {
    std::string name;
    int type = 0;          // 0: grouped GEMM kernel, 1: k_gather (im2col), 2: k_col2im, 3: k_reduce (split-K slices -> gradient)
    int cfg = 0, first = 0, count = 0, blocks = 0, lds = 0;
    int lean_act = 0;      // tile configurations 10 / 11 (lean head launches): the tail's transfer function
    int grid_x = 1, grid_y = 1;   // grouped kernel: tile slot x item
    bool tn = false;              // K-major operands (the weight-gradient launches)
    LaunchArgs args{};
    TnLaunchArgs targs{};         // tn launches carry compact items instead
    GatherArgs ga{};
    Col2imArgs ca{};
    ReduceArgs ra{};
    ThinArgs ta{};
    WadjArgs wa{};
    GpermArgs gp{};
    RowsumArgs rs{};
};

struct TimingRec { hipEvent_t a, b; int launch_name; };

}  // namespace

struct avae_handle {
    avae_config cfg{};
    std::mutex mu;
    std::string err;
    int es = 2, KU = 64;
    int B = 0, Bp = 0, ldB = 0;   // batch, batch padded to 128 rows, rup(batch, KU)
    int nz = 0, M = 0, ld_eps = 0;
    std::vector<Mod> mods;
    size_t P_flat = 0, P_int = 0, P_enc = 0; // flat API count, internal padded count (floats), floats of the encoder sides (they come first)
    size_t off_theta = 0, off_m = 0, off_v = 0, off_g = 0;   // byte offsets; g has P_int + 64 floats
    size_t stage_lo = 0, stage_bytes = 0;   // input-staging set 0 (X0 / X32 of every modality + eps); sets 1..kMultiSteps-1 follow it
    size_t off_eps = 0, off_partial = 0, off_state = 0, off_items = 0, off_adam = 0, off_inf = 0, off_stamps = 0, off_latent = 0;
    int n_partial = 0;
    size_t ws_bytes = 0;
    unsigned char* ws = nullptr;
    bool own_ws = false;
    hipStream_t cap_stream = nullptr;

    std::vector<WorkItem> items;            // training + eval tables (host mirror)
    std::vector<Launch> fwd, bwd;           // training launches: forward, dgrad chain
    std::vector<Launch> wgrad;              // all weight gradients (-> [all-reduce ->] k_adam)
    std::vector<Launch> wgrad_adam;         // small nets, one replica: the same launch with the Adam step in its epilogue (k_small_tn): no k_adam launch
    // Data-parallel buckets: bucket 0 = the decoder side of every modality (output layer, decoder layers: their weight gradients
    // can be taken as soon as bwd_dec1_latent has run), bucket 1 = the encoder side (heads, encoder layers: after the last dgrad).
    // The all-reduce of bucket 0 runs beside the encoder's dgrad chain and bucket 1's weight gradients; Adam is applied per bucket.
    int n_buckets = 1;                      // 1: models with a conv modality (their helper launches are not bucketed)
    int bwd_split = 0;                      // h->bwd[0 .. bwd_split) belongs to bucket 0's part of the step, the rest to bucket 1's
    std::vector<Launch> wgrad_b[2];
    struct Range { size_t off, count; };    // floats inside the gradient buffer (the cost slot rides at the end of the last range of bucket 0)
    std::vector<Range> ranges_b[2];
    int adam_first_b[2] = {0, 0}, adam_count_b[2] = {0, 0}, adam_blocks_b[2] = {0, 0};   // items of the bucket-ordered copy of the Adam table
    std::vector<AdamItem> adam_items_b;     // [bucket 0 items ..., bucket 1 items ...], tile bases per bucket
    size_t off_adam_b = 0;
    // library-owned collective (RCCL): one communicator per replica, its own stream, events between the two streams
    void* comm = nullptr;                   // AVAE_COMM_RCCL: the ncclComm_t
    int comm_world = 1, comm_rank = 0;
    bool comm_on = false;                   // a library-owned collective (either backend) is up
    size_t off_wire = 0;                    // RCCL with bf16 on the wire: the packed gradient [P_int] bf16
    // AVAE_COMM_IPC: this replica's exchange block (uncached device memory, exported to the peers), the peers' blocks as mapped here
    unsigned char* ipc_block = nullptr;
    size_t ipc_bytes = 0;
    unsigned char* ipc_peer[kMaxWorld] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ipc_opened[kMaxWorld] = {false, false, false, false, false, false, false, false};
    bool ipc_attached = false;
    IpcArgs ipc_args{};                     // layout + peers; off / granules / cost_idx are filled per call
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_grad[2] = {nullptr, nullptr}, ev_red[2] = {nullptr, nullptr};
    std::vector<hipGraphExec_t> g_dp[2];    // per staging set: captured segment of bucket b (forward + backward part + its weight gradients)
    // whole runs of data-parallel steps as ONE graph (staging + per step: segments, all-reduces on the comm stream, Adam per bucket):
    hipGraphExec_t g_dpm[2] = {nullptr, nullptr};
    hipGraph_t g_dpm_graph[2] = {nullptr, nullptr};
    hipGraphNode_t g_dpm_prep[2] = {nullptr, nullptr};
    bool dp_graph_failed = false;           // RCCL refused stream capture on this stack: host-stepped pipeline instead
    // Single-replica overlap inside the captured graphs (the same buckets): the decoder side's weight gradients + Adam run on a side
    // stream from the moment bwd_dec1_latent is done, beside the encoder's backward pass, its weight gradients / Adam and -- in the
    // multi-step graphs -- the NEXT step's encoder forward; the next step's first decoder launch waits for them.
    bool overlap = false, ov_split_wgrad = false;
    int fwd_dec_first = 0;                  // index in fwd of the first launch that reads decoder weights
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side = nullptr;
    Launch cost_only;                       // eval: K_COST alone, no step bump
    std::vector<AdamItem> adam_items;
    int adam_blocks = 0;
    // inference tables: per modality, device slots at off_inf
    struct Inf { std::vector<WorkItem> items; std::vector<Launch> launches; int rows = -1; size_t dev_off = 0; };
    std::vector<Inf> inf_enc, inf_dec;
    // serving (avae_generate): per row bucket one staging launch per call + one captured graph of [grouped decoder launches of all
    // modalities, slot-indirect output move]
    struct Serve { int bucket = 0; std::vector<WorkItem> items; std::vector<Launch> launches; hipGraphExec_t graph = nullptr; ServeArgs in;
                   Launch in_launch; bool fused_in = false;
                   ServeInArgs in_lean; int in_lean_grid = 0; bool lean_in = false;
                   hipGraphExec_t ring_graph[kServeRing] = {}; };       // [staging launch reading ring record i, the plan's launches]
    std::vector<Serve> serve;
    size_t off_slot = 0, off_serve_count = 0;
    // Serving without a per-call eager launch: the call's record {z, rows, outputs} goes into a slot of a pinned-host ring and the
    // graph of that slot is replayed -- its first kernel reads the record over PCIe.  `consumed` (pinned too) is the device's count
    // of records read: the host never runs more than the ring ahead of it.
    ServeSlot* serve_ring = nullptr;
    unsigned long long* serve_consumed = nullptr;
    unsigned long long serve_calls = 0;
    size_t off_chain = 0;
    size_t off_consts = 0, off_conv_tab = 0;   // 32 B {zeros | one, 0...}; device copy of conv_tab
    std::vector<ConvA> conv_tab;             // implicit patch matrices of the training plan

    hipGraphExec_t g_full = nullptr, g_eval = nullptr;
    hipGraphExec_t g_multi[2] = {nullptr, nullptr};   // kMultiSizes[i] whole steps per replay (avae_train_steps)
    hipGraph_t g_full_graph = nullptr, g_multi_graph[2] = {nullptr, nullptr};   // templates, kept: their staging-kernel nodes are re-parameterised per replay
    hipGraphNode_t g_full_prep = nullptr, g_multi_prep[2] = {nullptr, nullptr};

    // Calls on one handle share its activation buffers and serving slot: a call on a different stream than the previous one is
    // ordered behind that one's work (an event recorded on the old stream when the switch is seen: nothing per call otherwise).
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    hipEvent_t ev_switch = nullptr;
    unsigned draw_id = 0;                   // eval / reconstruct calls that drew their own eps (keys the generator: a fresh draw per call)
    bool timing = false;
    bool debug_sync = false;
    std::vector<std::string> tnames;
    std::vector<TimingRec> trecs;

    template <typename T> T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
    DevState* state() const { return at<DevState>(off_state); }
    float* grad() const { return at<float>(off_g); }
};

namespace {

// Row strides that are a multiple of 512 bytes put the rows of an operand tile (one 128-byte chunk per row and K step) on a
// handful of the L2's address-interleaved channels (a 2048-byte stride: C4's W shadows, 1024 x bf16 -- its dgrad K loop ran
// 1.03 us per tile against 0.90 for the forward's 2176-byte rows).  One more K unit (128 bytes) of zero padding per row avoids it.
inline int spread_ld(int ld, int KU, int es) {
    static const bool off = std::getenv("AVAE_NO_LD_SPREAD") != nullptr;
    return (!off && ((size_t)ld * es) % 512 == 0) ? ld + KU : ld;
}

Act make_act(Bump& b, int width, bool ones, int rows, int KU, int es, int ld = 0) {
    Act a;
    a.width = width;
    a.rows = rows;
    a.ones = ones;
    a.ld = ld > 0 ? ld : spread_ld((int)rup(width + 1, KU), KU, es);       // ld given: a buffer that is never a GEMM operand
    a.rm = b.take(rup(rows, kRowAlign) * (size_t)a.ld * es);
    return a;
}

Dense make_dense(Bump& b, size_t& pint, int in, int out, int KU, int es, bool head) {
    Dense d;
    d.in = in; d.out = out; d.head = head;
    d.ld = (int)rup(out, 32);                      // theta / m / v / g rows: whole 128-byte lines (k_adam streams them: rows cut at 16 bytes
                                                   // cost it 9.0 -> 11.75 us on C2), not padded to the K unit.  This layout is what the gradient
                                                   // all-reduce puts on the wire: 6.13 MB for C2's 5.87 MB of parameters (6.6 MB padded to K)
    d.ldw = spread_ld((int)rup(out, KU), KU, es);
    d.ldt = spread_ld((int)rup(in + 1, KU), KU, es);
    d.master = pint;
    pint += (size_t)(in + 1) * d.ld;
    d.W = b.take(rup(in + 1, kRowAlign) * (size_t)d.ldw * es);
    d.Wt = b.take(rup(out, kRowAlign) * (size_t)d.ldt * es);
    return d;
}

void check_config(const avae_config& c) {
    if (c.abi_version != AVAE_ABI_VERSION) throw Err("avae_config.abi_version mismatch");
    if (c.n_modalities < 1 || c.n_modalities > AVAE_MAX_MODALITIES) throw Err("n_modalities out of range");
    if (c.n_z < 1 || c.n_z > 64) throw Err("n_z must be in [1,64] (the fused head/latent epilogue holds 2*n_z columns in one tile)");
    if (c.batch_size < 1) throw Err("batch_size must be positive");
    if (c.compute_dtype != AVAE_F32 && c.compute_dtype != AVAE_BF16) throw Err("compute_dtype must be AVAE_F32 or AVAE_BF16");
    if (c.activation < AVAE_ACT_IDENTITY || c.activation > AVAE_ACT_TANH) throw Err("unknown activation");
    if (c.comm_buckets != 0 && c.comm_buckets != 1 && c.comm_buckets != 2) throw Err("comm_buckets must be 0, 1 or 2");
    if (c.wire_dtype != AVAE_F32 && c.wire_dtype != AVAE_BF16) throw Err("wire_dtype must be AVAE_F32 or AVAE_BF16");
    if (c.use_comm < AVAE_COMM_NONE || c.use_comm > AVAE_COMM_IPC) throw Err("use_comm must be AVAE_COMM_NONE, AVAE_COMM_RCCL or AVAE_COMM_IPC");
    for (int m = 0; m < c.n_modalities; ++m) {
        const avae_modality& mo = c.mod[m];
        if (mo.n_input < 1) throw Err("n_input must be positive");
        if (mo.hidden_conv) {
            // the reference's branch is hard-wired to 28x28: 28 -> 14 -> 7 -> 3 in the encoder (vae_assoc.py:171-199),
            // 1 -> 3 -> 7 -> 14 -> 28 in the decoder (:250-278) and a dense n_input x n_input on top (:287)
            if (mo.n_input != 784) throw Err("hidden_conv needs n_input = 784 (28x28 images)");
            if (!mo.binary) throw Err("hidden_conv needs a binary modality (the reference's non-binary conv decoder is shape-broken, vae_assoc.py:299)");
            if (mo.n_hidden_layers != 2 || mo.conv_gener[0] < 2 || mo.conv_gener[1] < 1) throw Err("hidden_conv needs n_hidden_recog_1/2 and n_hidden_gener_1 (>=2) / n_hidden_gener_2");
        }
        if (mo.n_hidden_layers < 1 || mo.n_hidden_layers > AVAE_MAX_HIDDEN) throw Err("n_hidden_layers out of range");
        for (int k = 0; k < mo.n_hidden_layers; ++k)
            if (mo.n_hidden[k] < 1) throw Err("hidden width must be positive");
    }
}

// Lays out the whole workspace; with h->ws == nullptr it only computes sizes.
// workgroups per image of the direct one-channel stage: 1 on the block-structured path (the reference's geometry), else kThinSplit
inline bool thin_fast(const ConvGeom& g) { return thin_fast_geometry(g) && !std::getenv("AVAE_NO_THIN_FAST"); }
// workgroups per image of the direct one-channel stage in mode 0 / 1 / 2 (AVAE_THIN_FAST_SPLIT="a,b,c": A/B runs)
inline int thin_split(const ConvGeom& g, int mode) {
    if (!thin_fast(g)) return kThinSplit;
    int sp[3] = {1, 4, 1};       // measured on c2conv (tools/thin_split_ab.sh): forward 7.1 us at 1 (12.3 at 2, 21.6 at 4), dX 10.4 / 7.6 / 5.8, dF 8.3 / 8.1 / 10.5
    if (const char* e = std::getenv("AVAE_THIN_FAST_SPLIT")) std::sscanf(e, "%d,%d,%d", &sp[0], &sp[1], &sp[2]);
    const int v = sp[mode];
    return v == 1 || v == 2 || v == 4 ? v : 1;
}

void plan_memory(avae_handle* h) {
    const avae_config& c = h->cfg;
    h->es = c.compute_dtype == AVAE_BF16 ? 2 : 4;
    h->KU = kTileBytesK / h->es;
    h->B = c.batch_size;
    h->Bp = (int)rup(h->B, kRowAlign);
    h->ldB = (int)rup(h->B, h->KU);
    h->nz = c.n_z;
    h->ld_eps = (int)rup(c.n_z, 4);
    h->M = c.n_modalities;
    const int KU = h->KU, es = h->es, Bp = h->Bp, ldB = h->ldB, nz = h->nz, B = h->B;
    Bump b;
    size_t pint = 0, pflat = 0;
    h->mods.clear();
    // Input staging first, as one contiguous block, followed by kMultiSteps-1 more copies of it: the multi-step graph
    // stages all its batches with one launch and step j of a replay reads set j (build_training_plan relocates the
    // pointers); every other path uses set 0.
    std::vector<Act> stage_x0(h->M);
    std::vector<size_t> stage_x32(h->M);
    h->stage_lo = b.off;
    for (int m = 0; m < h->M; ++m) {
        const avae_modality& mo = c.mod[m];
        stage_x0[m] = make_act(b, mo.n_input, mo.hidden_conv == 0, B, KU, es);
        stage_x32[m] = b.take((size_t)B * rup(mo.n_input, 8) * 4);
    }
    h->off_eps = b.take((size_t)B * h->ld_eps * 4);
    h->stage_bytes = b.off - h->stage_lo;
    b.take(h->stage_bytes * (kMultiSteps - 1));
    for (int m = 0; m < h->M; ++m) {
        const avae_modality& mo = c.mod[m];
        Mod md;
        md.n_in = mo.n_input;
        md.L = mo.n_hidden_layers;
        md.hs.assign(mo.n_hidden, mo.n_hidden + md.L);
        md.conv = mo.hidden_conv != 0;
        auto act_B = [&](int width, bool ones) { return make_act(b, width, ones, B, KU, es); };
        if (!md.conv) {
            int prev = md.n_in;
            for (int k = 0; k < md.L; ++k) { md.enc.push_back(make_dense(b, pint, prev, md.hs[k], KU, es, false)); pflat += (size_t)(prev + 1) * md.hs[k]; prev = md.hs[k]; }
            md.head = make_dense(b, pint, prev, 2 * nz, KU, es, true);
            pflat += (size_t)(prev + 1) * 2 * nz;
            prev = nz;
            for (int k = 0; k < md.L; ++k) { md.dec.push_back(make_dense(b, pint, prev, md.hs[k], KU, es, false)); pflat += (size_t)(prev + 1) * md.hs[k]; prev = md.hs[k]; }
            md.outl = make_dense(b, pint, prev, md.n_in, KU, es, false);
            pflat += (size_t)(prev + 1) * md.n_in;
            md.X0 = stage_x0[m];
            for (int k = 0; k < md.L; ++k) md.E.push_back(act_B(md.hs[k], true));
            md.Z = act_B(nz, true);
            for (int k = 0; k < md.L; ++k) md.D.push_back(act_B(md.hs[k], true));
            for (int k = 0; k < md.L; ++k) md.dE.push_back(act_B(md.hs[k], false));
            md.dH = act_B(2 * nz, false);
            for (int k = 0; k < md.L; ++k) md.dD.push_back(act_B(md.hs[k], false));
            md.dO = act_B(md.n_in, false);
        } else {
            const int R1 = md.hs[0], R2 = md.hs[1], G1 = mo.conv_gener[0], G2 = mo.conv_gener[1];
            md.L = 0;
            md.X0 = stage_x0[m];
            md.Z = act_B(nz, false);
            md.dH = act_B(2 * nz, false);
            md.dO = act_B(md.n_in, false);
            const bool no_impl = std::getenv("AVAE_NO_IMPLICIT") != nullptr;
            auto stage = [&](int IH, int Cin_in, int OH, int Cout, int k, int so, int d, int pad, bool bias, int act, int flat, bool plain_out, bool has_dgrad) {
                ConvStage st;
                // whole 16-byte chunks of channels per tap where that takes a few zero channels (the first decoder stage reads z: n_z
                // channels, padded here; the pad channels' filter rows stay zero and are not part of the flat parameters)
                const int cpc = 16 / es;
                const int Cin = (IH == 1 && flat == 1 && !no_impl) ? (int)rup(Cin_in, cpc) : Cin_in;
                st.cin_real = Cin_in;
                st.g = ConvGeom{B, IH, IH, Cin, OH, OH, k, so, d, pad, 0, 0, bias ? 1 : 0};
                st.bias = bias; st.act = act; st.flat = flat;
                const int K = k * k * Cin, rows = B * OH * OH;
                st.d = make_dense(b, pint, K, Cout, KU, es, flat == 3);
                st.P = make_act(b, K, bias, rows, KU, es);
                const bool thin = Cout == 1 && plain_out && flat == 1 && IH > 1 && IH * IH * Cin <= kThinIn && OH * OH <= kThinOut && K + 1 <= kThinF &&
                                  !std::getenv("AVAE_NO_THIN");
                // (the dense layer behind the flattened 28x28 map, flat == 2, is never a conv: it reads the map as stored -- dense_map -- or its flatten gather)
                // (row -> (image, pixel) by magic-number division: exact while rows x pixels-per-image < 2^32)
                const bool fits = (uint64_t)rows * (uint64_t)(OH * OH) < (1ull << 32) && (uint64_t)B * IH * IH * (uint64_t)(IH * IH) < (1ull << 32);
                {   // Which products of which stage run as implicit GEMMs.  The gather is dense in the conv direction (big map -> small
                    // map: every tap of every row is a real pixel) and mostly zeros in the transposed direction (zero insertion / full
                    // padding: 3 of 4 taps of a stride-2 transposed conv read the zero block), where the scatter product on the small
                    // side (round 2's adjoint-frame route) or the explicit patch gradients do a quarter of the work.  Measured on c2conv
                    // (tools/conv_policy_ab.sh, one box, ms per step): everything explicit 0.2715; everything implicit 0.2696 (22
                    // launches: conv forward 7.3 us vs 8.2 + 4.4, heads 7.5 vs 4.2 + 5.3, first decoder stage 4.2 vs 4.3 + 4.4, its
                    // latent dgrad 6.8 vs 4.5 + 6.0 -- but transposed-conv forward 24.3 vs 11.7 + 7.2, conv dgrad 23.6 vs 10.4 + 6.6,
                    // gather-form filter gradients 43 vs 17); implicit where the gather is dense 0.2376 (29 launches) = the default:
                    //   E  conv stages        forward + filter gradient implicit, input gradient by patch gradients + col2im
                    //   H  heads, D1 first decoder stage (1x1 input)   everything implicit
                    //   DT transposed convs   round 2's adjoint-frame route (scatter product + overlap-add, Padj for both gradients);
                    //                         their implicit input gradient alone ("DT:b": 8.3 us vs 5.9 with Padj at hand) 0.2436
                    // Letters f / w / b = forward / filter gradient / input gradient; AVAE_IMPL_POLICY overrides (A/B runs, tests).
                    const char* pol = std::getenv("AVAE_IMPL_POLICY");
                    const std::string policy = pol ? pol : "E:fw,H:fwb,D1:fwb,DT:";
                    const std::string cls = flat == 0 ? "E" : flat == 3 ? "H" : (flat == 1 && IH == 1) ? "D1" : flat == 1 ? "DT" : "X";
                    std::string letters;
                    for (size_t pos = 0; pos < policy.size();) {
                        const size_t e = policy.find(',', pos), c = policy.find(':', pos);
                        const size_t end = e == std::string::npos ? policy.size() : e;
                        if (c != std::string::npos && c < end && policy.substr(pos, c - pos) == cls) letters = policy.substr(c + 1, end - c - 1);
                        pos = end + 1;
                    }
                    const bool geo = !no_impl && fits && !thin && flat != 2 && (so == 1 || so == 2) && (d == 1 || d == 2);
                    st.impl = geo && letters.find('f') != std::string::npos && (Cin * es) % 16 == 0;
                    st.impl_w = geo && (st.impl || letters.find('w') != std::string::npos) && (Cin * es) % 16 == 0;
                    st.impl_bwd = geo && letters.find('b') != std::string::npos && has_dgrad && (Cout * es) % 16 == 0;
                    // Wide latents (2 n_z > 64): the head and latent-dgrad launches run on the 128-column tiles, which carry no
                    // gather -- the heads' forward and the first decoder stage's latent gradient stay on their explicit routes there
                    // (tools/fuzz_parity.py found the planner throwing for n_z = 33 with a conv modality).
                    if (2 * h->nz > 64) {
                        if (cls == "H") st.impl = false;
                        if (cls == "D1") st.impl_bwd = false;
                    }
                }
                if (plain_out) {     // hidden conv stage: own output / gradient buffers
                    // (the direct stage's one-channel maps are read and written pixel by pixel: the LAST map (28x28), whose consumer
                    // is the dense output layer, is stored as dense rows [B][784 | 1]: that layer then reads it as any dense layer
                    // reads its input.  AVAE_NO_IMPLICIT keeps round 2's compact 8-elements-per-pixel layout + flatten gather.)
                    st.dense_map = thin && OH == 28 && !no_impl;
                    if (st.dense_map) {
                        st.Y = make_act(b, OH * OH, true, B, KU, es);
                        st.dY = make_act(b, OH * OH, false, B, KU, es);
                    } else {
                        const int compact = thin && OH == 28 ? 8 : 0;
                        st.Y = make_act(b, Cout, false, rows, KU, es, compact);
                        st.dY = make_act(b, Cout, false, rows, KU, es, compact);
                    }
                }
                st.lddp = (int)rup(K, 8);
                st.dP = b.take((size_t)rows * st.lddp * 4);
                {   // a conv stage's weight gradient sums over batch x output pixels (up to 200 704 rows) into a few tiles:
                    // cut that K range so that the launch has a few hundred workgroups of at least 8 K tiles each
                    const int steps = (int)(rup(rows, KU) / KU), tiles = ((K + 1 + 63) / 64) * ((Cout + 63) / 64);
                    int split = std::min(steps / 8, (512 + tiles - 1) / tiles);
                    if (split > 1) {
                        st.kchunk = (steps + split - 1) / split;
                        st.ksplit = (steps + st.kchunk - 1) / st.kchunk;
                        st.part = b.take((size_t)st.ksplit * (K + 1) * st.d.ld * 4);
                    }
                }
                // transposed convs that upsample (d > 1) or pad heavily have an OUTPUT far larger than their input: the fp32 patch
                // gradients [B*OH*OW][k*k*Cin] of the plain path (160 MB for 7x7x32 -> 14x14x16) are replaced by the compute-type
                // patch matrix of the output gradient, [B*IH*IW][k*k*Cout] (11 MB), times the adjoint filter
                if (flat == 1 && plain_out && !thin && !st.impl && IH > 1 && Cout <= 64 && (long)IH * IH * Cout < (long)OH * OH * Cin * 2 && !std::getenv("AVAE_NO_ADJ")) {
                    st.adj = true;
                    st.Padj = make_act(b, k * k * Cout, false, B * IH * IH, KU, es);
                    st.ldadj = (int)rup(k * k * Cout, KU);
                    st.Wadj = b.take(rup(Cin, kRowAlign) * (size_t)st.ldadj * es);
                    const int KA = k * k * Cout, rin = B * IH * IH;
                    st.ldf = (int)rup(Cin, KU);
                    st.Wf = b.take(rup(KA, kRowAlign) * (size_t)st.ldf * es);
                    st.ldT = (int)rup(KA, 8);
                    st.T = b.take((size_t)rin * st.ldT * 4);
                    st.ldga = (int)rup(KA, 4);
                    st.Gadj = b.take((size_t)Cin * st.ldga * 4);
                    {   // split of the adjoint filter gradient's K range (= input pixels), as for the plain stages above
                        const int steps = (int)(rup(rin, KU) / KU), tiles = ((Cin + 63) / 64) * ((KA + 63) / 64);
                        const int split = std::min(steps / 8, (512 + tiles - 1) / tiles);
                        if (split > 1) {
                            st.akchunk = (steps + split - 1) / split;
                            st.aksplit = (steps + st.akchunk - 1) / st.akchunk;
                            st.apart = b.take((size_t)st.aksplit * Cin * st.ldga * 4);
                        }
                    }
                    st.rs_cols4 = (int)rup(Cout, 4);
                    st.rs_part = b.take((size_t)kRowsumBlocks * st.rs_cols4 * 4);
                }
                if (st.impl_bwd && !st.Wadj) {      // adjoint filter shadow [Cin][(kh', kw', co)]: the B operand of the implicit input gradient
                    st.ldadj = (int)rup(k * k * Cout, KU);
                    st.Wadj = b.take(rup(Cin, kRowAlign) * (size_t)st.ldadj * es);
                }
                if (thin) {
                    st.thin = true;
                    st.thin_kp = (int)rup(K + 1, 4);
                    st.thin_blocks = B * thin_split(st.g, 2);      // filter-gradient partial sums: one slice per workgroup
                    st.thin_part = b.take((size_t)st.thin_blocks * st.thin_kp * 4);
                }
                pflat += (size_t)k * k * Cin_in * Cout + (flat == 0 ? 0 : Cout);
                return st;
            };
            // encoder: conv k5 s2 SAME (pad-before 1), conv k5 s2 SAME, conv k5 s1 VALID, flatten(3x3) + dense heads
            md.cenc.push_back(stage(28, 1, 14, R1, 5, 2, 1, 1, false, AVAE_ACT_IDENTITY, 0, true, false));
            md.cenc.push_back(stage(14, R1, 7, 2 * R1, 5, 2, 1, 1, false, AVAE_ACT_IDENTITY, 0, true, true));
            md.cenc.push_back(stage(7, 2 * R1, 3, R2, 5, 1, 1, 0, false, AVAE_ACT_IDENTITY, 0, true, true));
            md.cenc.push_back(stage(3, R2, 1, 2 * nz, 3, 1, 1, 0, true, AVAE_ACT_IDENTITY, 3, false, true));
            // decoder: transposed convs as convs of the dilated input with the flipped filter (pad = k-1-pad_before)
            md.cdec.push_back(stage(1, nz, 3, G1, 3, 1, 1, 2, true, AVAE_ACT_SIGMOID, 1, true, true));
            md.cdec.push_back(stage(3, G1, 7, G1 / 2, 5, 1, 1, 4, true, AVAE_ACT_SIGMOID, 1, true, true));
            md.cdec.push_back(stage(7, G1 / 2, 14, G2, 5, 1, 2, 3, true, AVAE_ACT_SIGMOID, 1, true, true));
            md.cdec.push_back(stage(14, G2, 28, 1, 5, 1, 2, 3, true, AVAE_ACT_SIGMOID, 1, true, true));
            md.cdec.push_back(stage(28, 1, 1, md.n_in, 28, 1, 1, 0, true, AVAE_ACT_IDENTITY, 2, false, true));
            md.head = md.cenc[3].d;
            md.outl = md.cdec[4].d;
            auto src = [](ConvStage& st, int sb, int sp) { st.g.src_sb = sb; st.g.src_sp = sp; };
            src(md.cenc[0], md.X0.ld, 1);
            for (int i = 1; i < 4; ++i) { const ConvStage& pv = md.cenc[i - 1]; src(md.cenc[i], pv.g.OH * pv.g.OW * pv.Y.ld, pv.Y.ld); }
            src(md.cdec[0], md.Z.ld, 0);
            for (int i = 1; i < 5; ++i) {
                const ConvStage& pv = md.cdec[i - 1];
                if (pv.dense_map) src(md.cdec[i], pv.Y.ld, 1);        // one channel, dense rows: image stride = row stride, pixel stride 1
                else src(md.cdec[i], pv.g.OH * pv.g.OW * pv.Y.ld, pv.Y.ld);
            }
        }
        md.ld32 = (int)rup(md.n_in, 8);      // multiple of the widest epilogue vector (8 elements)
        md.X32 = stage_x32[m];
        md.out32 = b.take((size_t)B * md.ld32 * 4);
        md.mulv = b.take((size_t)B * 2 * nz * 4);
        md.g0 = b.take((size_t)B * 3 * h->ld_eps * 4);  // [g0mu | g0lv | dz/dlv factor], each roundup(n_z, 4) wide
        h->mods.push_back(std::move(md));
    }
    // Master layout (theta, m, v, g share it): the ENCODER side of every modality first, then every DECODER side, the step's cost in
    // the float right behind -- so each data-parallel bucket is ONE contiguous float range (bucket 1 = [0, P_enc), bucket 0 =
    // [P_enc, P_int + 1)) and the whole buffer one range.  The flat API order (reference creation order) is untouched: it is
    // mapped through Dense::master.
    {
        size_t pm = 0;
        auto place = [&](Dense& d) { d.master = pm; pm += (size_t)(d.in + 1) * d.ld; };
        for (Mod& md : h->mods) {
            if (md.conv) { for (ConvStage& st : md.cenc) place(st.d); md.head = md.cenc[3].d; }
            else { for (Dense& d : md.enc) place(d); place(md.head); }
        }
        h->P_enc = pm;
        for (Mod& md : h->mods) {
            if (md.conv) { for (ConvStage& st : md.cdec) place(st.d); md.outl = md.cdec[4].d; }
            else { for (Dense& d : md.dec) place(d); place(md.outl); }
        }
        if (pm != pint) throw Err("internal error: master layout size");
    }
    h->P_int = pint;
    h->P_flat = pflat;
    h->off_theta = b.take(pint * 4);
    h->off_m = b.take(pint * 4);
    h->off_v = b.take(pint * 4);
    h->off_g = b.take((pint + 64) * 4);      // + cost slot (element P_int), padded
    if (c.use_comm == AVAE_COMM_RCCL && c.wire_dtype == AVAE_BF16) h->off_wire = b.take((pint + 64) * 2);     // the gradient as it travels
    // cost partial slots: one per output-loss tile (the smallest tile, 32x32, bounds the count) + latent tiles
    int slots = (B + kLatentRows - 1) / kLatentRows;
    for (int m = 0; m < h->M; ++m) slots += ((B + 31) / 32) * ((h->mods[m].n_in + 31) / 32);
    h->n_partial = slots;
    h->off_partial = b.take((size_t)slots * 4);
    h->off_state = b.take(sizeof(DevState));
    // work-item tables: training items + eval cost item; inference tables per modality
    size_t n_train_items = 0;
    for (int m = 0; m < h->M; ++m) n_train_items += 6 * (size_t)h->mods[m].L + 8;
    n_train_items += 8;
    h->off_items = b.take(n_train_items * sizeof(WorkItem));
    size_t n_adam = 0;
    for (int m = 0; m < h->M; ++m) n_adam += h->mods[m].conv ? 9 : 2 * (size_t)h->mods[m].L + 2;
    h->off_adam = b.take(n_adam * sizeof(AdamItem));
    h->off_adam_b = b.take(n_adam * sizeof(AdamItem));
    h->off_slot = b.take(sizeof(ServeSlot));
    h->off_serve_count = b.take(8);
    h->off_consts = b.take(32);
    h->off_chain = b.take((kMaxMod * 64 + 1) * 4);      // k_chain2: a ticket counter per (modality, row block), + its error word
    h->off_conv_tab = b.take(4 * kMaxConvA * kMaxMod * sizeof(ConvA));
    h->off_inf = b.off;
    for (int m = 0; m < h->M; ++m) b.take(2 * ((size_t)h->mods[m].L + 1) * sizeof(WorkItem));
#ifdef AVAE_STAMPS
    h->off_stamps = b.take((size_t)kStampLaunches * kStampBlocks * kStampWords * 8);
#endif
    h->ws_bytes = b.off;
}

// ----------------------------------------------------------------------------- flat <-> internal
template <bool ToInternal>
void convert_params(const avae_handle* h, float* flat, float* internal) {
    size_t f = 0;
    auto dense = [&](const Dense& d) {
        float* I = internal + d.master;
        if (!d.head) {
            for (int r = 0; r < d.in; ++r)
                for (int c = 0; c < d.out; ++c, ++f) { if (ToInternal) I[(size_t)r * d.ld + c] = flat[f]; else flat[f] = I[(size_t)r * d.ld + c]; }
            for (int c = 0; c < d.out; ++c, ++f) { if (ToInternal) I[(size_t)d.in * d.ld + c] = flat[f]; else flat[f] = I[(size_t)d.in * d.ld + c]; }
        } else {
            const int nz = d.out / 2;
            for (int half = 0; half < 2; ++half) {
                for (int r = 0; r < d.in; ++r)
                    for (int c = 0; c < nz; ++c, ++f) { float& x = I[(size_t)r * d.ld + half * nz + c]; if (ToInternal) x = flat[f]; else flat[f] = x; }
                for (int c = 0; c < nz; ++c, ++f) { float& x = I[(size_t)d.in * d.ld + half * nz + c]; if (ToInternal) x = flat[f]; else flat[f] = x; }
            }
        }
    };
    auto xfer = [&](float& x) { if (ToInternal) x = flat[f]; else flat[f] = x; ++f; };
    auto conv_stage = [&](const ConvStage& st) {
        const Dense& d = st.d;
        float* I = internal + d.master;
        const int k = st.g.k, Ci = st.g.Cin, Co = d.out, K = k * k * Ci;
        if (st.flat == 0) {            // tf.nn.conv2d filter [k,k,ci,co] == matrix rows (kh,kw,ci) x cols co; no bias
            for (int r = 0; r < K; ++r) for (int c = 0; c < Co; ++c) xfer(I[(size_t)r * d.ld + c]);
        } else if (st.flat == 1) {     // conv2d_transpose filter [k,k,co,ci] (deconv.py:78), flipped into the matrix; bias[co]
            // (ci runs over the reference's channels; a stage whose input is padded to whole chunks keeps zero rows for the pads)
            for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) for (int co = 0; co < Co; ++co) for (int ci = 0; ci < st.cin_real; ++ci)
                xfer(I[(size_t)(((k - 1 - kh) * k + (k - 1 - kw)) * Ci + ci) * d.ld + co]);
            for (int co = 0; co < Co; ++co) xfer(I[(size_t)K * d.ld + co]);
        } else {
            dense(d);                  // dense on the flattened map (2) or the fused heads (3: d.head)
        }
    };
    for (const Mod& md : h->mods) {
        if (md.conv) {
            for (const ConvStage& st : md.cenc) conv_stage(st);
            for (const ConvStage& st : md.cdec) conv_stage(st);
            continue;
        }
        for (const Dense& d : md.enc) dense(d);
        dense(md.head);
        for (const Dense& d : md.dec) dense(d);
        dense(md.outl);
    }
    if (f != h->P_flat) throw Err("internal error: flat parameter count mismatch");
}

// ----------------------------------------------------------------------------- work items
WorkItem gemm_item(int kind, int M, int N, int K, const void* A, int lda, const void* B, int ldb) {
    WorkItem w;
    std::memset(&w, 0, sizeof(w));
    w.kind = kind; w.M = M; w.N = N; w.K = K; w.A = A; w.lda = lda; w.B = B; w.ldb = ldb;
    return w;
}

struct Builder {
    avae_handle* h;
    std::vector<WorkItem>& items;
    int B;      // rows of this plan (batch or inference rows)
    bool train; // training plan (losses, gradients) or inference plan
    int next_slot = 0;
    Builder(avae_handle* h_, std::vector<WorkItem>& it, int rows, bool tr) : h(h_), items(it), B(rows), train(tr) {}
    template <typename T> T* p(size_t off) const { return h->at<T>(off); }
    int K_of(int n) const { return (int)rup(n, h->KU); }

    WorkItem fwd_hidden(const Act& in, const Dense& d, const Act& out) {
        WorkItem w = gemm_item(K_FWD_HIDDEN, B, d.out, K_of(d.in + 1), p<void>(in.rm), in.ld, p<void>(d.Wt), d.ldt);
        w.act = h->cfg.activation;
        w.out0 = p<void>(out.rm); w.ld0 = out.ld;
        w.kin = d.in; w.aux1 = p<unsigned char>(d.W) + (size_t)d.in * d.ldw * h->es; w.ld1 = d.ldw;   // bias row (see WorkItem::bias_ep)
        return w;
    }
    WorkItem fwd_head(const Mod& md, bool with_z, bool implicit = false) {
        const Act& in = md.conv ? md.cenc[3].P : md.E.back();
        WorkItem w = gemm_item(K_FWD_HEAD, B, 2 * h->nz, K_of(md.head.in + 1), p<void>(in.rm), in.ld, p<void>(md.head.Wt), md.head.ldt);
        if (implicit) {          // conv encoder: the flattened 3x3 map, gathered by the GEMM itself
            ConvGeom g = md.cenc[3].g; g.B = B;
            w.A = p<void>(h->off_consts); w.lda = 0;
            w.conv = add_conv(g, p<void>(md.cenc[2].Y.rm), true);
        }
        w.nz = h->nz;
        w.out0 = p<void>(md.mulv); w.ld0 = 2 * h->nz;
        w.out1 = with_z ? p<void>(md.Z.rm) : nullptr; w.ld1 = md.Z.ld;
        w.aux0 = p<void>(h->off_eps); w.ldx = h->ld_eps;
        return w;
    }
    WorkItem fwd_out(const Mod& md, int m, bool loss) {
        const Act& in = md.conv ? (md.cdec[3].dense_map ? md.cdec[3].Y : md.cdec[4].P) : md.D.back();
        WorkItem w = gemm_item(loss ? K_FWD_OUT_LOSS : K_FWD_OUT_STORE, B, md.n_in, K_of(md.outl.in + 1), p<void>(in.rm), in.ld,
                               p<void>(md.outl.Wt), md.outl.ldt);
        w.binary = h->cfg.mod[m].binary ? 1 : 0;
        if (loss && !md.conv) { w.kin = md.outl.in; w.aux1 = p<unsigned char>(md.outl.W) + (size_t)md.outl.in * md.outl.ldw * h->es; w.ld1 = md.outl.ldw; }
        const float bg = (float)(h->cfg.batch_global > 0 ? h->cfg.batch_global : h->cfg.batch_size);
        if (loss) {
            w.scale = w.binary ? h->cfg.mod[m].weight / bg : h->cfg.mod[m].weight;
            w.aux0 = p<void>(md.X32); w.ldx = md.ld32;
            w.out0 = p<void>(md.dO.rm); w.ld0 = md.dO.ld;
            w.partial = p<float>(h->off_partial);
        } else {
            w.out0 = p<void>(md.out32); w.ld0 = md.ld32;
        }
        return w;
    }
    WorkItem dgrad_hidden(const Act& dA, const Dense& d, const Act& yprev, const Act& dprev) {
        WorkItem w = gemm_item(K_DGRAD_HIDDEN, B, d.in, K_of(d.out), p<void>(dA.rm), dA.ld, p<void>(d.W), d.ldw);
        w.act = h->cfg.activation;
        w.aux0 = p<void>(yprev.rm); w.ldx = yprev.ld;
        w.out0 = p<void>(dprev.rm); w.ld0 = dprev.ld;
        return w;
    }
    WorkItem dgrad_latent(const Mod& md) {
        const Dense& d = md.dec[0];
        const Act& dA = md.dD[0];
        WorkItem w = gemm_item(K_DGRAD_LATENT, B, h->nz, K_of(d.out), p<void>(dA.rm), dA.ld, p<void>(d.W), d.ldw);
        w.nz = h->nz;
        w.aux2 = p<void>(md.g0);
        w.out0 = p<void>(md.dH.rm); w.ld0 = md.dH.ld;
        return w;
    }
    // ---- conv branch: every stage is a GEMM on its patch matrix
    int conv_rows(const ConvStage& st) const { return B * st.g.OH * st.g.OW; }
    WorkItem conv_fwd(const ConvStage& st) {
        WorkItem w = gemm_item(K_FWD_HIDDEN, conv_rows(st), st.d.out, K_of(st.d.in + 1), p<void>(st.P.rm), st.P.ld, p<void>(st.d.Wt), st.d.ldt);
        w.act = st.act;
        w.out0 = p<void>(st.Y.rm); w.ld0 = st.Y.ld;
        return w;
    }
    WorkItem conv_dgrad(const ConvStage& st, const Act& dA) {          // fp32 patch gradients dP = dA . W^T
        WorkItem w = gemm_item(K_DGRAD_F32, conv_rows(st), st.d.in, K_of(st.d.out), p<void>(dA.rm), dA.ld, p<void>(st.d.W), st.d.ldw);
        w.out0 = p<void>(st.dP); w.ld0 = st.lddp;
        return w;
    }
    // input gradient of a transposed-conv stage through the adjoint patch matrix (ConvStage::adj)
    GatherSeg adj_gather_seg(const ConvStage& st) {
        GatherSeg g;
        std::memset(&g, 0, sizeof(g));
        const ConvGeom& f = st.g;
        // rows = the stage's input pixels, columns = (kh', kw', co) with kh' = k-1-kh: source row oh = (ih*d + kh' - (k-1-pad))/so
        g.g = ConvGeom{B, f.OH, f.OW, st.d.out, f.IH, f.IW, f.k, f.d, f.so, f.k - 1 - f.pad, f.OH * f.OW * st.dY.ld, st.dY.ld, 0};
        g.src = p<void>(st.dY.rm);
        g.P = p<void>(st.Padj.rm); g.ldp = st.Padj.ld;
        tile_shape(B * f.IH * f.IW, f.k * f.k * st.d.out, &g.cl, &g.rpt, &g.tiles_r, &g.tiles_c);
        return g;
    }
    // forward of such a stage: T = X . Wadj (rows = its input pixels), then overlap-add + bias + transfer function
    WorkItem adj_fwd(const ConvStage& st, const ConvStage& prev) {
        const int rows = B * st.g.IH * st.g.IW, KA = st.g.k * st.g.k * st.d.out;
        WorkItem w = gemm_item(K_DGRAD_F32, rows, KA, K_of(st.g.Cin), p<void>(prev.Y.rm), prev.Y.ld, p<void>(st.Wf), st.ldf);
        w.out0 = p<void>(st.T); w.ld0 = st.ldT;
        return w;
    }
    Col2imSeg adj_overlap_seg(const ConvStage& st) {
        Col2imSeg c;
        std::memset(&c, 0, sizeof(c));
        const ConvGeom& f = st.g;
        c.g = ConvGeom{B, f.OH, f.OW, st.d.out, f.IH, f.IW, f.k, f.d, f.so, f.k - 1 - f.pad, 0, 0, 0};
        c.dP = p<float>(st.T); c.lddp = st.ldT;
        c.bias = p<unsigned char>(st.d.Wt) + (size_t)st.d.in * h->es; c.bias_ld = st.d.ldt;
        c.act = st.act;
        c.dA = p<void>(st.Y.rm); c.lda = st.Y.ld;
        tile_shape(B * f.OH * f.OW, st.d.out, &c.cl, &c.rpt, &c.tiles_r, &c.tiles_c);
        return c;
    }
    // filter gradient in the adjoint frame: Gadj[ci][(kh', kw', co)] = sum over input pixels of X[p][ci] * Padj[p][...]
    WorkItem adj_wgrad(const ConvStage& st, const ConvStage& prev) {
        const int rows = B * st.g.IH * st.g.IW, KA = st.g.k * st.g.k * st.d.out;
        WorkItem w = gemm_item(K_WGRAD, st.g.Cin, KA, K_of(rows), p<void>(prev.Y.rm), prev.Y.ld, p<void>(st.Padj.rm), st.Padj.ld);
        w.out0 = p<void>(st.Gadj); w.ld0 = st.ldga;
        if (st.aksplit > 1) { w.ksplit = st.aksplit; w.kchunk = st.akchunk; w.out1 = p<void>(st.apart); }
        w.bias_row = -1;
        return w;
    }
    WorkItem adj_dgrad(const ConvStage& st, const ConvStage& prev) {
        const int rows = B * st.g.IH * st.g.IW, KA = st.g.k * st.g.k * st.d.out;
        WorkItem w = gemm_item(K_DGRAD_HIDDEN, rows, st.g.Cin, K_of(KA), p<void>(st.Padj.rm), st.Padj.ld, p<void>(st.Wadj), st.ldadj);
        w.act = prev.act;
        w.aux0 = p<void>(prev.Y.rm); w.ldx = prev.Y.ld;
        w.out0 = p<void>(prev.dY.rm); w.ld0 = prev.dY.ld;
        return w;
    }
    // the one-output-channel stage as direct kernels: `prev` produced its input
    ThinSeg thin_seg(const ConvStage& st, const ConvStage& prev) {
        ThinSeg t;
        std::memset(&t, 0, sizeof(t));
        t.g = st.g; t.g.B = B;
        t.X = p<void>(prev.Y.rm);
        t.Wt = p<void>(st.d.Wt);
        t.Y = p<void>(st.Y.rm); t.ldy = st.Y.ld;
        t.act = st.act; t.act_in = prev.act;
        t.dY = p<void>(st.dY.rm); t.lddy = st.dY.ld;
        if (st.dense_map) { t.img_y = st.Y.ld; t.ldy = 1; t.img_dy = st.dY.ld; t.lddy = 1; }      // dense rows: pixel stride 1, image stride = row stride
        t.split = thin_split(st.g, 0); t.fast = thin_fast(st.g) ? 1 : 0;      // (the training launches set the split of their mode: thin_launch)
        t.dX = p<void>(prev.dY.rm); t.lddx = prev.dY.ld;
        t.part = p<float>(st.thin_part); t.Kp = st.thin_kp;
        return t;
    }
    // tile shape of the element-wise conv helpers: 4*cl columns (cl lanes of a quad each, a power of two <= 16) x (256/cl)*rpt
    // rows; outputs with few columns get tall tiles instead of idle lanes, outputs with few rows one row per thread so
    // that there are enough workgroups
    static void tile_shape(int rows, int cols, int* cl, int* rpt, int* tiles_r, int* tiles_c) {
        int c = 1;
        while (c < 16 && 4 * c < cols) c *= 2;
        *cl = c;
        *rpt = (long)rows * ((cols + 4 * c - 1) / (4 * c)) >= 4L * (kThreads / c) * 512 ? 4 : 1;      // >= 512 workgroups at 4 rows/thread
        const int tr = (kThreads / c) * *rpt;
        *tiles_r = (rows + tr - 1) / tr; *tiles_c = (cols + 4 * c - 1) / (4 * c);
    }
    GatherSeg gather_seg(const ConvStage& st, const void* src) {
        GatherSeg g;
        std::memset(&g, 0, sizeof(g));
        g.g = st.g; g.g.B = B;
        g.src = src;
        g.P = p<void>(st.P.rm); g.ldp = st.P.ld;
        const int M = conv_rows(st), KC = st.d.in + (st.bias ? 1 : 0);
        tile_shape(M, KC, &g.cl, &g.rpt, &g.tiles_r, &g.tiles_c);
        return g;
    }
    // gradient of stage `st`'s input: into the producing stage `prev` (nullptr: latent mode -> dH of `md`)
    Col2imSeg col2im_seg(const ConvStage& st, const ConvStage* prev, const Mod& md) {
        Col2imSeg c;
        std::memset(&c, 0, sizeof(c));
        c.g = st.g; c.g.B = B;
        c.dP = p<float>(st.dP); c.lddp = st.lddp;
        int C;
        if (prev) {
            c.yprev = prev->act == AVAE_ACT_IDENTITY ? nullptr : p<void>(prev->Y.rm); c.ldy = prev->Y.ld; c.act = prev->act;
            c.dA = p<void>(prev->dY.rm); c.lda = prev->dY.ld;
            C = st.g.Cin;
        } else {
            c.g0 = p<float>(md.g0); c.nz = h->nz;
            c.dA = p<void>(md.dH.rm); c.lda = md.dH.ld;
            C = 2 * h->nz;
        }
        const int R = B * st.g.IH * st.g.IW;
        tile_shape(R, C, &c.cl, &c.rpt, &c.tiles_r, &c.tiles_c);
        return c;
    }
    // ---- implicit-GEMM routes (ConvA): descriptors are appended to the handle's table, items carry the 1-based index
    static unsigned magic(int x) { return x <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)x) + 1u; }
    int add_conv(const ConvGeom& g, const void* src, bool ones) {
        ConvA c;
        std::memset(&c, 0, sizeof(c));
        c.src = src; c.consts = p<void>(h->off_consts);
        c.M = g.B * g.OH * g.OW; c.K = g.k * g.k * g.Cin;
        c.OHW = g.OH * g.OW; c.OW = g.OW; c.IH = g.IH; c.IW = g.IW; c.Cin = g.Cin;
        c.k = g.k; c.so = g.so; c.d = g.d; c.pad = g.pad; c.src_sb = g.src_sb; c.src_sp = g.src_sp; c.ones = ones ? 1 : 0;
        c.mg_ohw = magic(c.OHW); c.mg_ow = magic(c.OW); c.mg_k = magic(c.k); c.mg_cin = magic(c.Cin);
        if ((c.d != 1 && c.d != 2) || (c.so != 1 && c.so != 2)) throw Err("internal error: implicit conv geometry (strides 1 and 2 only)");
        if ((size_t)h->conv_tab.size() >= 4 * (size_t)kMaxConvA * kMaxMod) throw Err("internal error: too many implicit patch matrices");
        h->conv_tab.push_back(c);
        return (int)h->conv_tab.size();
    }
    // forward of stage `st` on the map `src` (strides in st.g): Y = act(P . W_aug), P implicit
    WorkItem conv_fwd_impl(const ConvStage& st, const void* src) {
        ConvGeom g = st.g; g.B = B;
        WorkItem w = gemm_item(K_FWD_HIDDEN, conv_rows(st), st.d.out, K_of(st.d.in + 1), p<void>(h->off_consts), 0, p<void>(st.d.Wt), st.d.ldt);
        w.act = st.act;
        w.out0 = p<void>(st.Y.rm); w.ld0 = st.Y.ld;
        w.conv = add_conv(g, src, st.bias);
        return w;
    }
    // adjoint geometry of stage `st`: rows = its input pixels, source = its output gradient `dY` (element strides sb / sp)
    ConvGeom adj_geom(const ConvStage& st, int sb, int sp) const {
        const ConvGeom& f = st.g;
        return ConvGeom{B, f.OH, f.OW, st.d.out, f.IH, f.IW, f.k, f.d, f.so, f.k - 1 - f.pad, sb, sp, 0};
    }
    // input gradient of stage `st` as an implicit GEMM on its output gradient: dX = (Padj . Wadj^T) * act'(stored input)
    WorkItem adj_dgrad_impl(const ConvStage& st, const void* dY, int sb, int sp, int prev_act, const Act& prevY, const Act& prevdY) {
        const int rows = B * st.g.IH * st.g.IW, KA = st.g.k * st.g.k * st.d.out;
        WorkItem w = gemm_item(K_DGRAD_HIDDEN, rows, st.g.Cin, K_of(KA), p<void>(h->off_consts), 0, p<void>(st.Wadj), st.ldadj);
        w.act = prev_act;
        w.aux0 = p<void>(prevY.rm); w.ldx = prevY.ld;
        w.out0 = p<void>(prevdY.rm); w.ld0 = prevdY.ld;
        w.conv = add_conv(adj_geom(st, sb, sp), dY, false);
        return w;
    }
    // the first decoder stage's input is z: its input gradient takes the latent epilogue (dz -> [dmu | dlv], reparameterisation)
    WorkItem adj_dgrad_latent_impl(const ConvStage& st, const Mod& md) {
        const int KA = st.g.k * st.g.k * st.d.out;
        WorkItem w = gemm_item(K_DGRAD_LATENT, B, h->nz, K_of(KA), p<void>(h->off_consts), 0, p<void>(st.Wadj), st.ldadj);
        w.nz = h->nz;
        w.aux2 = p<void>(md.g0);
        w.out0 = p<void>(md.dH.rm); w.ld0 = md.dH.ld;
        w.conv = add_conv(adj_geom(st, st.g.OH * st.g.OW * st.dY.ld, st.dY.ld), p<void>(st.dY.rm), false);
        return w;
    }
    // filter gradient of stage `st`: dW_aug = P^T . dA with P implicit (K-major operand of the weight-gradient kernel)
    WorkItem wgrad_impl(const ConvStage& st, const void* src, const Act& dA) {
        ConvGeom g = st.g; g.B = B;
        const int rows = conv_rows(st);
        WorkItem w = gemm_item(K_WGRAD, st.d.in + 1, st.d.out, K_of(rows), p<void>(h->off_consts), 0, p<void>(dA.rm), dA.ld);
        w.out0 = h->grad() + st.d.master; w.ld0 = st.d.ld;
        if (st.ksplit > 1) { w.ksplit = st.ksplit; w.kchunk = st.kchunk; w.out1 = p<void>(st.part); }
        w.conv = add_conv(g, src, st.bias);
        return w;
    }
    // dW_aug[m][n] = sum_k X_aug[k][m] * dA[k][n]: both operands are read as stored (row = sample k), see the K-major
    // ("TN") images in avae_kernels.hip; rows >= batch of either buffer are zero padding
    WorkItem wgrad(const Act& x, const Dense& d, const Act& dA) {
        WorkItem w = gemm_item(K_WGRAD, d.in + 1, d.out, K_of(x.rows), p<void>(x.rm), x.ld, p<void>(dA.rm), dA.ld);
        w.out0 = h->grad() + d.master; w.ld0 = d.ld;
        return w;
    }
    WorkItem wgrad_stage(const ConvStage& st, const Act& dA) {
        WorkItem w = wgrad(st.P, st.d, dA);
        if (st.ksplit > 1) { w.ksplit = st.ksplit; w.kchunk = st.kchunk; w.out1 = p<void>(st.part); }
        return w;
    }
    WorkItem latent() {
        WorkItem w;
        std::memset(&w, 0, sizeof(w));
        w.kind = K_LATENT; w.M = B; w.nz = h->nz; w.n_mod = h->M;
        const float bg = (float)(h->cfg.batch_global > 0 ? h->cfg.batch_global : h->cfg.batch_size);
        w.inv_bg = 1.0f / bg; w.lambda = h->cfg.assoc_lambda;
        const void** in_slots[kMaxMod] = {&w.A, &w.B, &w.aux0, &w.aux1};
        void** out_slots[kMaxMod - 1] = {&w.out0, &w.out1, &w.out2};
        for (int m = 0; m < h->M; ++m) {
            *in_slots[m] = p<void>(h->mods[m].mulv);
            if (m < kMaxMod - 1) *out_slots[m] = p<void>(h->mods[m].g0); else w.aux2 = p<void>(h->mods[m].g0);
            w.wts[m] = h->cfg.mod[m].weight;
        }
        w.eps = p<float>(h->off_eps); w.ldx = h->ld_eps;
        w.partial = p<float>(h->off_partial);
        return w;
    }
    WorkItem cost(bool bump) {
        WorkItem w;
        std::memset(&w, 0, sizeof(w));
        w.kind = K_COST; w.partial = p<float>(h->off_partial); w.bump_step = bump ? 1 : 0;
        w.scale = h->cfg.learning_rate; w.lambda = h->cfg.beta1; w.inv_bg = h->cfg.beta2;
        w.out0 = h->grad() + h->P_int;
        return w;
    }
};

inline bool is_gemm(int kind) { return kind <= K_WGRAD || kind == K_DGRAD_F32 || kind == K_SERVE_Z; }

// Fixes the tile configuration of one launch and lays its items' tiles out back to back.
Launch finish_launch(avae_handle* h, std::vector<WorkItem>& items, int first, int count, const std::string& name, int* next_slot) {
    Launch L;
    L.name = name; L.first = first; L.count = count;
    bool need128 = false, narrow = false;
    long tiles128 = 0;
    for (int i = first; i < first + count; ++i) {
        const WorkItem& w = items[i];
        if ((w.kind == K_FWD_HEAD || w.kind == K_DGRAD_LATENT) && 2 * w.nz > 64) need128 = true;
        if (is_gemm(w.kind)) tiles128 += (long)((w.M + 127) / 128) * ((w.N + 127) / 128);
        if (is_gemm(w.kind) && (w.N <= 64 || w.M <= 64 || w.conv > 0)) narrow = true;      // a 128-wide tile would be mostly padding (implicit patch matrices: small tiles only)
    }
    L.cfg = (need128 || (tiles128 >= 192 && !narrow)) ? 1 : 0;
    {   // 256x128 tiles, 8 waves, one workgroup per CU: a third more flops per operand byte pulled from L2 into LDS,
        // and that fill rate (~60-70 GB/s per CU) is what bounds the 128x128 loop.  Plain single-C-tile kinds on big
        // problems only.  (AVAE_NO_256=1 keeps the 128x128 kernel: A/B measurements.)
        bool plain = std::getenv("AVAE_NO_256") == nullptr;
        long tiles256 = 0;
        for (int i = first; i < first + count; ++i) {
            const WorkItem& w = items[i];
            plain = plain && (w.kind == K_FWD_HIDDEN || w.kind == K_DGRAD_HIDDEN || w.kind == K_WGRAD);
            tiles256 += (long)((w.M + 255) / 256) * ((w.N + 127) / 128);
        }
        if (L.cfg == 1 && plain && tiles256 >= 192) L.cfg = 2;
    }
    if (L.cfg == 0 && !std::getenv("AVAE_NO_32")) {      // few 64x64 tiles: half-height tiles put twice the CUs on the launch
        long t64 = 0;
        bool nt = true;
        for (int i = first; i < first + count; ++i) {
            const WorkItem& w = items[i];
            if (is_gemm(w.kind)) t64 += (long)((w.M + 63) / 64) * ((w.N + 63) / 64);
            nt = nt && w.kind != K_WGRAD && w.kind != K_DGRAD_F32;
        }
        if (nt && t64 > 0 && t64 <= 128) L.cfg = 3;
        bool head = false;
        for (int i = first; i < first + count; ++i) head = head || items[i].kind == K_FWD_HEAD || items[i].kind == K_DGRAD_LATENT || items[i].kind == K_SERVE_Z;
        if (L.cfg == 3 && !head && !std::getenv("AVAE_NO_32x32")) L.cfg = 5;      // no kind that needs 2*n_z columns in one tile
        if (L.cfg == 5 && !std::getenv("AVAE_NO_LEAN")) {       // the chain's two kinds, one of them (and one transfer function) per launch: the lean kernel (k_small)
            bool lean = true;
            const int k0 = items[first].kind, a0 = items[first].act;
            for (int i = first; i < first + count; ++i)
                lean = lean && items[i].kind == k0 && items[i].act == a0 && !items[i].bias_ep && items[i].K > 0 && items[i].conv == 0;
            if (lean && (k0 == K_FWD_HIDDEN || k0 == K_DGRAD_HIDDEN)) L.cfg = 7;
            bool store = true;                                    // the inference / serving output launch: k_small's third kind
            for (int i = first; i < first + count; ++i) store = store && items[i].kind == K_FWD_OUT_STORE && !items[i].bias_ep && items[i].K > 0 && items[i].conv == 0;
            if (store) L.cfg = 7;
            bool loss = false, only_loss = true;                  // the output + loss launch (with the latent item riding in it): k_small_loss
            for (int i = first; i < first + count; ++i) {
                loss = loss || items[i].kind == K_FWD_OUT_LOSS;
                only_loss = only_loss && ((items[i].kind == K_FWD_OUT_LOSS && !items[i].bias_ep && items[i].K > 0 && items[i].conv == 0) || items[i].kind == K_LATENT);
            }
            if (L.cfg == 5 && loss && only_loss && !std::getenv("AVAE_NO_LEAN_LOSS")) L.cfg = 9;
        }
    }
    if (L.cfg == 1 && need128 && !std::getenv("AVAE_NO_64x128")) {      // wide-latent head launches: few tiles, K loop = load latency
        long t = 0;
        bool nt = true;
        for (int i = first; i < first + count; ++i) {
            const WorkItem& w = items[i];
            if (is_gemm(w.kind)) t += (long)((w.M + 63) / 64) * ((w.N + 127) / 128);
            nt = nt && w.kind != K_WGRAD && w.kind != K_DGRAD_F32;
        }
        if (nt && t <= 256) L.cfg = 4;
    }
    if (L.cfg == 1 && !std::getenv("AVAE_NO_LOSS8")) {      // the output + loss launch of the big nets: 256x64 tiles, 8 waves, register epilogue
        long t = 0;
        bool loss_only = true;
        for (int i = first; i < first + count; ++i) {
            const WorkItem& w = items[i];
            if (is_gemm(w.kind)) { loss_only = loss_only && w.kind == K_FWD_OUT_LOSS; t += (long)((w.M + 255) / 256) * ((w.N + 63) / 64); }
        }
        if (loss_only && t >= 192) L.cfg = 6;
    }
    for (int i = first; i < first + count; ++i)
        if (items[i].conv > 0 && !(L.cfg == 0 || L.cfg == 3 || L.cfg == 5)) throw Err("internal error: an implicit patch matrix on a tile configuration without the gather");
    const int T = (L.cfg == 1 || L.cfg == 2 || L.cfg == 4) ? 128 : (L.cfg == 5 || L.cfg == 7 || L.cfg == 9) ? 32 : 64;
    const int TM = (L.cfg == 2 || L.cfg == 6) ? 256 : (L.cfg == 3 || L.cfg == 5 || L.cfg == 7 || L.cfg == 9) ? 32 : L.cfg == 4 ? 64 : T;
    if ((L.cfg == 2 || L.cfg == 6) && !std::getenv("AVAE_NO_BIAS_EP"))      // 8-wave NT tiles: bias in the epilogue where that saves a K tile
        for (int i = first; i < first + count; ++i) {
            WorkItem& w = items[i];
            if ((w.kind == K_FWD_HIDDEN || w.kind == K_FWD_OUT_LOSS) && w.kin > 0 && w.kin % h->KU == 0 && w.K == w.kin + h->KU) { w.K = w.kin; w.bias_ep = 1; }
        }
    int max_tiles = 1;
    for (int i = first; i < first + count; ++i) {
        WorkItem& w = items[i];
        if (is_gemm(w.kind)) {
            if (w.bias_row >= 0) {       // (< 0: a product without a bias row, e.g. the adjoint-frame filter gradients)
                w.bias_row = 0;
                if (w.kind == K_WGRAD && w.M > 1 && (w.M - 1) % TM == 0 && !std::getenv("AVAE_NO_BIAS_MFMA")) w.bias_row = w.M - 1;
            }
            w.tiles_m = (w.M - (w.bias_row > 0 ? 1 : 0) + TM - 1) / TM;
            w.tiles_n = (w.N + T - 1) / T;
            if ((w.kind == K_FWD_HEAD || w.kind == K_DGRAD_LATENT) && w.tiles_n != 1) throw Err("internal error: head tile");
            if (w.kind == K_FWD_OUT_LOSS) { w.slot_base = *next_slot; *next_slot += w.tiles_m * w.tiles_n; }
        } else if (w.kind == K_LATENT) {
            w.tiles_m = (w.M + kLatentRows - 1) / kLatentRows; w.tiles_n = 1;
            w.slot_base = *next_slot; *next_slot += w.tiles_m;
        } else {   // K_COST
            w.tiles_m = w.tiles_n = 1;
            w.n_slots = *next_slot;
        }
        max_tiles = std::max(max_tiles, w.tiles_m * w.tiles_n * (w.kind == K_WGRAD && w.ksplit > 1 ? w.ksplit : 1));
    }
    // grid: x = tile slot (multiple of 8: the kernel's XCD-aware tile order needs it), y = item
    L.grid_x = (max_tiles + 7) & ~7; L.grid_y = count;
    L.blocks = L.grid_x * L.grid_y;
    {
        bool two = false;
        for (int i = first; i < first + count; ++i) two = two || items[i].kind == K_FWD_HEAD || items[i].kind == K_DGRAD_LATENT || items[i].kind == K_SERVE_Z;
        L.lds = L.cfg == 7 ? 4 * 64 * kTileBytesK : L.cfg == 9 ? 4 * 64 * kTileBytesK + 64 : tile_lds_bytes(L.cfg, two);
    }
    L.tn = true;
    for (int i = first; i < first + count; ++i) L.tn = L.tn && items[i].kind == K_WGRAD;
    if (count > (L.tn ? kMaxTnItems : kMaxItemsPerLaunch)) throw Err("internal error: too many items in one launch");
    std::memset(&L.args, 0, sizeof(L.args));
    std::memset(&L.targs, 0, sizeof(L.targs));
    if (L.tn) {
        // big weight gradients: concentrate each item on 8/G XCDs (the launch is bound by L2 misses); G must leave no XCD
        // without an item, so it divides the item count
        int G = 1;
        if (L.cfg == 2) for (int g : {4, 2}) if (count % g == 0) { G = g; break; }
        if (const char* e = std::getenv("AVAE_TN_G")) G = std::atoi(e);
        if (G != 1 && G != 2 && G != 4 && G != 8) throw Err("AVAE_TN_G must be 1, 2, 4 or 8");
        auto tiles_of = [&](const WorkItem& w) { return w.tiles_m * w.tiles_n * (w.ksplit > 1 ? w.ksplit : 1); };
        // The grid is (longest entry) x (entries), so layers of very different size would make it mostly padding
        // workgroups (C2: 1248 workgroups for 416 tiles).  Items are therefore cut into entries of at most SX tiles, SX
        // chosen to launch the fewest workgroups within the entry table's size.
        int SX = L.grid_x;
        if (G == 1 && !std::getenv("AVAE_NO_TN_SLICES")) {
            long best = -1;
            for (int sx = 8; sx <= L.grid_x; sx += 8) {
                long entries = 0;
                for (int i = first; i < first + count; ++i) entries += (tiles_of(items[i]) + sx - 1) / sx;
                if (entries > kMaxTnItems) continue;
                if (best < 0 || entries * sx <= best) { best = entries * sx; SX = sx; }
            }
        }
        int n = 0;
        for (int i = 0; i < count; ++i) {
            const WorkItem& w = items[first + i];
            const int nt = tiles_of(w);
            for (int off = 0; off < nt; off += SX) {
                if (n >= kMaxTnItems) throw Err("internal error: too many weight-gradient entries");
                TnItem& t = L.targs.items[n++];
                t.A = w.A; t.B = w.B; t.out = reinterpret_cast<float*>(w.ksplit > 1 ? w.out1 : w.out0);
                t.bias_row = std::max(w.bias_row, 0);
                t.M = w.bias_row > 0 ? w.M - 1 : w.M; t.N = w.N; t.K = w.K; t.lda = w.lda; t.ldb = w.ldb; t.ld0 = w.ld0;
                t.tiles_m = w.tiles_m; t.tiles_n = w.tiles_n; t.ksplit = w.ksplit; t.kchunk = w.kchunk;
                t.tile_off = off; t.tile_cnt = std::min(SX, nt - off); t.conv = w.conv;
            }
        }
        if (G > 1 && !std::getenv("AVAE_NO_TN_BALANCE")) {
            // Entry y of a group of G runs on XCDs [ (y % G) * 8/G, ... ): its position decides WHERE its tiles run.  Layers
            // differ in size (C4: 32, 28, 8 and 4 tiles), and in plan order two of the four XCD pairs end up with 66 tiles per
            // XCD -- a third round of 62 us for four tiles on a 32-CU XCD (195 us; the other XCDs are done after 138).  Longest
            // first onto the lightest position balances the positions; G = 4 or 2 by the fewest rounds that leaves.
            std::vector<TnItem> ent(L.targs.items, L.targs.items + n);
            std::stable_sort(ent.begin(), ent.end(), [](const TnItem& a, const TnItem& b) { return a.tile_cnt > b.tile_cnt; });
            auto pack = [&](int g, std::vector<std::vector<TnItem>>& bins) {
                bins.assign(g, {});
                std::vector<long> load(g, 0);
                const int cap = kMaxTnItems / g;
                for (const TnItem& e : ent) {
                    int best = -1;
                    for (int p = 0; p < g; ++p) if ((int)bins[p].size() < cap && (best < 0 || load[p] < load[best])) best = p;
                    bins[best].push_back(e); load[best] += e.tile_cnt;
                }
                const long per_xcd = (*std::max_element(load.begin(), load.end()) * g + 7) / 8;
                return (per_xcd + 31) / 32;                       // rounds on a 32-CU XCD at one workgroup per CU
            };
            std::vector<std::vector<TnItem>> bins, bins2;
            if (!std::getenv("AVAE_TN_G") && G == 4 && n * 2 <= kMaxTnItems * 2) {
                const long r4 = pack(4, bins), r2 = pack(2, bins2);
                if (r2 < r4) { G = 2; bins.swap(bins2); }
            } else {
                pack(G, bins);
            }
            size_t groups = 0;
            for (const auto& bn : bins) groups = std::max(groups, bn.size());
            std::memset(L.targs.items, 0, sizeof(L.targs.items));
            for (size_t g = 0; g < groups; ++g)
                for (int p = 0; p < G; ++p) if (g < bins[p].size()) L.targs.items[g * G + p] = bins[p][g];      // holes: tile_cnt 0
            n = (int)groups * G;
        }
        L.grid_x = SX;
        L.targs.n_items = n;
        L.targs.grid_x = L.grid_x;
        L.targs.xcd_group = G;
        L.grid_y = (n + G - 1) / G * G;
        L.blocks = L.grid_x * L.grid_y;
    } else {
        L.args.n_items = count;
        L.args.grid_x = L.grid_x;
        for (int i = 0; i < count; ++i) L.args.items[i] = items[first + i];
    }
    L.args.conv_tab = L.targs.conv_tab = h->at<ConvA>(h->off_conv_tab);
    if (const char* e = std::getenv("AVAE_SCHED")) L.args.sched = L.targs.sched = std::atoi(e);
    if (*next_slot > h->n_partial) throw Err("internal error: cost partial slots overflow");
    return L;
}

void dp_ranges(const avae_handle* h, int* n_buckets, std::vector<avae_handle::Range> (&out)[2]);

// k_small_tn: hand every XCD its own list of (layer, tile range) -- TnLaunchArgs::pieces.  The launch's entries (layers cut into
// slices for the grid's sake) are merged back into whole layers; the layers, longest first, go whole to the fullest XCD that still
// has room under the quota of ceil(tiles / 8) (best fit), and a layer that fits nowhere is cut: the emptiest XCD is filled up and
// the rest tried again.  Leaves the launch as it was when an XCD would need more than kTnPieces pieces.
void xcd_pieces(Launch& L) {
    TnLaunchArgs& a = L.targs;
    std::vector<TnItem> layers;
    for (int i = 0; i < a.n_items; ++i) {
        const TnItem& e = a.items[i];
        if (e.tile_cnt <= 0) continue;
        bool seen = false;
        for (const TnItem& l : layers) seen = seen || l.out == e.out;
        if (seen) continue;
        TnItem l = e;
        l.tile_off = 0; l.tile_cnt = e.tiles_m * e.tiles_n;
        if (l.tile_cnt > 65535) return;
        layers.push_back(l);
    }
    long total = 0;
    for (const TnItem& l : layers) total += l.tile_cnt;
    if (layers.empty() || total > 8L * 65535) return;
    const int quota = (int)((total + 7) / 8);
    std::vector<int> order(layers.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return layers[x].tile_cnt > layers[y].tile_cnt; });
    struct Bin { int load = 0; std::vector<TnPiece> pc; };
    Bin bins[8];
    for (int li : order) {
        int off = 0, left = layers[li].tile_cnt;
        while (left > 0) {
            int best = -1;
            for (int b = 0; b < 8; ++b)          // whole (what is left of it): the fullest XCD it fits
                if ((int)bins[b].pc.size() < kTnPieces && bins[b].load + left <= quota && (best < 0 || bins[b].load > bins[best].load)) best = b;
            int take = left;
            if (best < 0) {                      // cut: fill the emptiest XCD
                for (int b = 0; b < 8; ++b)
                    if ((int)bins[b].pc.size() < kTnPieces && bins[b].load < quota && (best < 0 || bins[b].load < bins[best].load)) best = b;
                if (best < 0) return;
                take = std::min(left, quota - bins[best].load);
            }
            bins[best].load += take;
            bins[best].pc.push_back(TnPiece{(unsigned short)li, (unsigned short)off, (unsigned short)bins[best].load, 0});
            off += take; left -= take;
        }
    }
    int longest = 0;
    for (int b = 0; b < 8; ++b) {
        longest = std::max(longest, bins[b].load);
        for (int i = 0; i < kTnPieces; ++i)
            a.pieces[b][i] = i < (int)bins[b].pc.size() ? bins[b].pc[i] : TnPiece{0, 0, (unsigned short)bins[b].load, 0};
    }
    std::memset(a.items, 0, sizeof(a.items));
    for (size_t i = 0; i < layers.size(); ++i) a.items[i] = layers[i];
    a.n_items = (int)layers.size();
    a.xcd_pieces = 1;
    L.grid_x = 8 * longest; L.grid_y = 1; L.blocks = L.grid_x;
    a.grid_x = L.grid_x;
}

void build_training_plan(avae_handle* h) {
    h->items.clear(); h->fwd.clear(); h->bwd.clear(); h->wgrad.clear(); h->conv_tab.clear();
    Builder bd(h, h->items, h->B, true);
    int slot = 0;
    int Lmax = 0;
    for (const Mod& md : h->mods) Lmax = std::max(Lmax, md.L);
    auto group = [&](const std::string& name, std::vector<Launch>& dst, auto&& fill) {
        const int first = (int)h->items.size();
        fill();
        const int count = (int)h->items.size() - first;
        if (count <= 0) return;
        dst.push_back(finish_launch(h, h->items, first, count, name, &slot));
    };
    // Small nets: the launch that follows the heads (the decoder's first layer, K = n_z + 1) and the one that follows bwd_dec1_latent
    // (the heads' input gradient, K = 2 n_z) multiply over one or two K tiles and are pure launch overhead (4.3 us each on C2).  When the
    // producing launch runs on 32x64 tiles they become the tail product of its items (WorkItem::tail_*) and disappear.  The weights
    // they read are final since the previous step's Adam, their inputs exist only in the producing workgroup -- nothing else changes.
    auto fuse_tail = [&](std::vector<Launch>& ls, int mode) {
        if (ls.size() < 2 || std::getenv("AVAE_NO_TAIL")) return false;
        Launch& Lp = ls[ls.size() - 2];          // producer: fwd_head / bwd_dec1_latent
        Launch& Lc = ls[ls.size() - 1];          // consumer: one K tile
        if (Lp.type != 0 || Lc.type != 0 || Lp.tn || Lc.tn || Lp.cfg != 3) return false;
        const int want_p = mode == 1 ? K_FWD_HEAD : K_DGRAD_LATENT, want_c = mode == 1 ? K_FWD_HIDDEN : K_DGRAD_HIDDEN;
        std::vector<int> prod(Lc.count, -1);
        for (int i = 0; i < Lc.count; ++i) {
            const WorkItem& c = h->items[Lc.first + i];
            if (c.kind != want_c || c.bias_ep || c.K > 2 * h->KU || c.N > 1024) return false;      // (the narrow result is one 64-column tile)
            for (int k = 0; k < Lp.count; ++k) {
                const WorkItem& n = h->items[Lp.first + k];
                if (n.kind != want_p || n.M != c.M) continue;
                const void* res = mode == 1 ? n.out1 : n.out0;
                const int ldres = mode == 1 ? n.ld1 : n.ld0;
                if (res && res == c.A && ldres == c.lda) prod[i] = k;
            }
            if (prod[i] < 0) return false;
        }
        for (int i = 0; i < Lc.count; ++i) {
            const WorkItem c = h->items[Lc.first + i];
            for (WorkItem* n : {&h->items[Lp.first + prod[i]], &Lp.args.items[prod[i]]}) {
                n->tail_mode = mode; n->tail_w = c.B; n->tail_ldw = c.ldb; n->tail_n = c.N; n->tail_out = c.out0; n->tail_ldo = c.ld0;
                n->tail_aux = c.aux0; n->tail_ldx = c.ldx; n->tail_act = c.act; n->tail_kt = c.K / h->KU;
                n->tiles_n = (c.N + 63) / 64;                    // tail slices of 64 columns, a workgroup each
            }
        }
        int max_tiles = 1;
        for (int k = 0; k < Lp.count; ++k) max_tiles = std::max(max_tiles, h->items[Lp.first + k].tiles_m * h->items[Lp.first + k].tiles_n);
        Lp.grid_x = (max_tiles + 7) & ~7; Lp.blocks = Lp.grid_x * Lp.grid_y; Lp.args.grid_x = Lp.grid_x;
        Lp.name += "+" + Lc.name;
        ls.pop_back();
        // the lean frame for the fused head launches (k_small_latb): every GEMM item of the kind, one transfer function, its narrow result and
        // the consumer's K within the 8-KiB image
        if (!std::getenv("AVAE_NO_LEAN") && !std::getenv("AVAE_NO_LEAN_HEAD")) {
            bool lean = true;
            int act = -1;
            for (int k = 0; k < Lp.count; ++k) {
                const WorkItem& n = h->items[Lp.first + k];
                if (n.kind == K_COST) continue;
                lean = lean && n.kind == want_p && n.tail_mode == mode
                            && (act < 0 || act == n.tail_act) && n.K > 0 && 2 * n.nz <= 64 && n.tail_kt >= 1 && n.tail_kt <= 2;
                act = n.tail_act;
            }
            if (lean && mode == 1)
                for (int k = 0; k < Lp.count; ++k) lean = lean && h->items[Lp.first + k].kind == K_FWD_HEAD && h->items[Lp.first + k].out1 != nullptr;
            if (lean) { Lp.cfg = mode == 2 ? 10 : 11; Lp.lds = small_head_lds_bytes(); Lp.lean_act = act; }
        }
        return true;
    };
    // conv-branch helper launches (one segment per conv modality)
    // (`want`: which conv modalities take part -- stages route per modality: patch-matrix, adjoint-frame or direct)
    auto every_conv = [](const Mod&) { return true; };
    auto gather_launch = [&](const std::string& name, std::vector<Launch>& dst, auto&& stage_of, auto&& src_of, auto&& want) {
        Launch L;
        L.name = name; L.type = 1;
        int base = 0;
        for (Mod& md : h->mods) if (md.conv && want(md)) {
            GatherSeg g = bd.gather_seg(stage_of(md), src_of(md));
            g.tile_base = base; base += g.tiles_r * g.tiles_c;
            L.ga.seg[L.ga.n_seg++] = g;
        }
        L.blocks = base;
        if (base > 0) dst.push_back(L);
    };
    auto col2im_launch = [&](const std::string& name, std::vector<Launch>& dst, auto&& stage_of, auto&& prev_of, auto&& want) {
        Launch L;
        L.name = name; L.type = 2;
        int base = 0;
        for (Mod& md : h->mods) if (md.conv && want(md)) {
            Col2imSeg c = bd.col2im_seg(stage_of(md), prev_of(md), md);
            c.tile_base = base; base += c.tiles_r * c.tiles_c;
            L.ca.seg[L.ca.n_seg++] = c;
        }
        L.blocks = base;
        if (base > 0) dst.push_back(L);
    };
    // the decoder's last transposed conv (one output channel) runs as direct kernels: mode 0 forward, 1 input gradient,
    // 2 filter-gradient partial sums
    auto thin_launch = [&](const std::string& name, std::vector<Launch>& dst, int mode, int i) {
        Launch L;
        L.name = name; L.type = 4;
        L.ta.mode = mode;
        int base = 0;
        for (Mod& md : h->mods) if (md.conv && md.cdec[i].thin) {
            ThinSeg t = bd.thin_seg(md.cdec[i], md.cdec[i - 1]);
            t.split = thin_split(md.cdec[i].g, mode);
            t.block_base = base;
            base += h->B * t.split;                        // workgroups per image
            L.ta.seg[L.ta.n_seg++] = t;
        }
        L.blocks = base;
        if (base > 0) dst.push_back(L);
    };
    auto is_adj = [&](int i) { bool t = false; for (const Mod& md : h->mods) t = t || (md.conv && i >= 0 && i <= 3 && md.cdec[i].adj); return t; };
    auto is_thin = [&](int i) { bool t = false; for (const Mod& md : h->mods) t = t || (md.conv && md.cdec[i].thin); return t; };
    bool any_conv = false;
    for (const Mod& md : h->mods) any_conv = any_conv || md.conv;
    // ---- forward.  Modalities with fewer hidden layers simply sit out a launch; data dependencies
    // are per modality and launches are stream-ordered, so this is always safe.
    for (int k = 0; k < Lmax; ++k)
        group("fwd_enc" + std::to_string(k + 1), h->fwd, [&] {
            for (Mod& md : h->mods) if (k < md.L) h->items.push_back(bd.fwd_hidden(k == 0 ? md.X0 : md.E[k - 1], md.enc[k], md.E[k]));
        });
    // Experiment (AVAE_CHAIN2=1): fwd_enc1 and fwd_enc2 of the small nets as ONE launch behind a row-block-local hand-off
    // (k_chain2) instead of a kernel boundary.  Needs both on the lean 32x32 kernel, equal widths (the same tile grid) and whole quads.
    if (std::getenv("AVAE_CHAIN2") && !any_conv && h->fwd.size() == 2 && h->fwd[0].cfg == 7 && h->fwd[1].cfg == 7 && h->fwd[0].count == h->fwd[1].count
        && 2 * h->fwd[0].count <= kMaxItemsPerLaunch) {
        Launch& A = h->fwd[0];
        const Launch& Bn = h->fwd[1];
        bool ok = true;
        for (int i = 0; i < A.count; ++i) {
            const WorkItem& a = A.args.items[i];
            const WorkItem& c = Bn.args.items[i];
            ok = ok && a.kind == K_FWD_HIDDEN && c.kind == K_FWD_HIDDEN && a.act == c.act && a.M == c.M && a.N == c.N && a.tiles_m == c.tiles_m && a.tiles_n == c.tiles_n
                    && c.A == a.out0 && c.lda == a.ld0 && a.N % 4 == 0 && a.tiles_m <= 64;
        }
        if (ok) {
            for (int i = 0; i < A.count; ++i) A.args.items[A.count + i] = Bn.args.items[i];
            A.args.n_items = 2 * A.count;
            A.cfg = 13; A.name += "+" + Bn.name;
            h->fwd.pop_back();
        }
    }
    if (any_conv) {   // conv encoder: three convs, then the flatten + dense heads.  Per modality and stage: implicit GEMM (the patch
                      // matrix is gathered by the GEMM's own operand loads), or im2col launch + GEMM where the channels are no whole chunks
        for (int i = 0; i < 3; ++i) {
            auto src_of = [&, i](Mod& md) { return i == 0 ? h->at<void>(md.X0.rm) : h->at<void>(md.cenc[i - 1].Y.rm); };
            gather_launch("conv_enc" + std::to_string(i + 1) + "_im2col", h->fwd, [&](Mod& md) -> const ConvStage& { return md.cenc[i]; },
                          src_of, [i](const Mod& md) { return !md.cenc[i].impl; });
            group("conv_enc" + std::to_string(i + 1), h->fwd, [&] {
                for (Mod& md : h->mods) if (md.conv) h->items.push_back(md.cenc[i].impl ? bd.conv_fwd_impl(md.cenc[i], src_of(md)) : bd.conv_fwd(md.cenc[i])); });
        }
        gather_launch("conv_head_im2col", h->fwd, [&](Mod& md) -> const ConvStage& { return md.cenc[3]; },
                      [&](Mod& md) { return h->at<void>(md.cenc[2].Y.rm); }, [](const Mod& md) { return !md.cenc[3].impl; });
    }
    group("fwd_head", h->fwd, [&] { for (Mod& md : h->mods) h->items.push_back(bd.fwd_head(md, true, md.conv && md.cenc[3].impl)); });
    // KL + association terms and their (mu, lv) gradients need every modality's (mu, lv) (ready after fwd_head) and are needed by
    // bwd_dec1_latent.  Small nets: they ride in the fwd_out_loss launch (the decoder's hidden launches stay plain GEMM launches).
    // Big nets, whose loss launch runs one 144-KB workgroup per CU on the 8-wave tile: there the item's workgroups would each take a
    // CU's only slot and send the launch into a second round (C4: 512 workgroups, 33 us), so it gets a launch of its own here.
    bool latent_alone = false;
    {
        long t64 = 0, t128 = 0;
        bool narrow = false;
        for (const Mod& md : h->mods) {
            t64 += (long)((h->B + 255) / 256) * ((md.n_in + 63) / 64);
            t128 += (long)((h->B + 127) / 128) * ((md.n_in + 127) / 128);
            narrow = narrow || md.n_in <= 64 || h->B <= 64;
        }
        latent_alone = t64 >= 192 && t128 >= 192 && !narrow && !std::getenv("AVAE_NO_LOSS8");      // = finish_launch's choice of cfg 6
    }
    if (latent_alone) group("latent", h->fwd, [&] { h->items.push_back(bd.latent()); });
    h->fwd_dec_first = (int)h->fwd.size();
    for (int k = 0; k < std::max(Lmax, 1); ++k) {
        group("fwd_dec" + std::to_string(k + 1), h->fwd, [&] {
            for (Mod& md : h->mods) if (k < md.L) h->items.push_back(bd.fwd_hidden(k == 0 ? md.Z : md.D[k - 1], md.dec[k], md.D[k]));
        });
        if (k == 0 && !any_conv && !latent_alone && fuse_tail(h->fwd, 1)) h->fwd_dec_first = (int)h->fwd.size() - 1;      // fwd_dec1 rides in fwd_head
    }
    if (any_conv) {   // adjoint filter shadows of the transposed-conv stages that run through Padj / the scatter product (refreshed once per step)
        Launch L;
        L.name = "conv_wadj"; L.type = 6;
        int base = 0;
        for (Mod& md : h->mods) if (md.conv) for (const ConvStage& st : md.cdec) if (st.adj) {
            WadjSeg& g = L.wa.seg[L.wa.n_seg++];
            g.Wt = h->at<void>(st.d.Wt); g.Wadj = h->at<void>(st.Wadj); g.Wf = h->at<void>(st.Wf); g.ldt = st.d.ldt; g.ldadj = st.ldadj; g.ldf = st.ldf;
            g.k = st.g.k; g.Cin = st.g.Cin; g.Cout = st.d.out;
            g.block_base = base; base += (st.g.Cin * st.g.k * st.g.k * st.d.out + kThreads - 1) / kThreads;
        }
        L.blocks = base;
        if (base > 0 && std::getenv("AVAE_NO_WADJ_FOLD")) h->fwd.push_back(L);       // default: k_adam writes these shadows (AdamItem::Wadj)
    }
    if (any_conv) {   // deconv decoder: four transposed convs (+ bias + sigmoid each), then the dense output layer
        for (int i = 0; i < 4; ++i) {
            // every modality routes its stage i on its own (depths differ between modalities): direct (one output channel), implicit
            // GEMM in gather form, round 2's adjoint-frame scatter product, or im2col + GEMM
            auto plain = [&, i](const Mod& md) { return !md.cdec[i].thin && !md.cdec[i].adj && !md.cdec[i].impl; };
            auto src_of = [&, i](Mod& md) { return i == 0 ? h->at<void>(md.Z.rm) : h->at<void>(md.cdec[i - 1].Y.rm); };
            if (is_thin(i)) thin_launch("conv_dec" + std::to_string(i + 1) + "_direct", h->fwd, 0, i);
            if (is_adj(i)) {       // scatter product on the stage's (small) input, then overlap-add + bias + transfer function
                group("conv_dec" + std::to_string(i + 1) + "_scatter", h->fwd, [&] {
                    for (Mod& md : h->mods) if (md.conv && md.cdec[i].adj) h->items.push_back(bd.adj_fwd(md.cdec[i], md.cdec[i - 1])); });
                Launch L;
                L.name = "conv_dec" + std::to_string(i + 1) + "_overlap"; L.type = 2;
                int base = 0;
                for (Mod& md : h->mods) if (md.conv && md.cdec[i].adj) {
                    Col2imSeg c = bd.adj_overlap_seg(md.cdec[i]);
                    c.tile_base = base; base += c.tiles_r * c.tiles_c;
                    L.ca.seg[L.ca.n_seg++] = c;
                }
                L.blocks = base;
                h->fwd.push_back(L);
            }
            gather_launch("conv_dec" + std::to_string(i + 1) + "_im2col", h->fwd, [&](Mod& md) -> const ConvStage& { return md.cdec[i]; }, src_of, plain);
            group("conv_dec" + std::to_string(i + 1), h->fwd, [&] {
                for (Mod& md : h->mods) if (md.conv) {
                    if (md.cdec[i].impl) h->items.push_back(bd.conv_fwd_impl(md.cdec[i], src_of(md)));
                    else if (plain(md)) h->items.push_back(bd.conv_fwd(md.cdec[i]));
                } });
        }
        gather_launch("conv_out_im2col", h->fwd, [&](Mod& md) -> const ConvStage& { return md.cdec[4]; },
                      [&](Mod& md) { return h->at<void>(md.cdec[3].Y.rm); }, [](const Mod& md) { return !md.cdec[3].dense_map; });
    }
    group("fwd_out_loss", h->fwd, [&] {
        for (int m = 0; m < h->M; ++m) h->items.push_back(bd.fwd_out(h->mods[m], m, true));
        if (!latent_alone) h->items.push_back(bd.latent());
    });
    // ---- backward: the dgrad chain (one launch per layer, all modalities), then EVERY weight gradient in the last
    // launch(es), then k_adam.  Conv stages go GEMM (fp32 patch gradients) -> k_col2im (sum + act' -> gradient of the
    // producing stage).
    group("bwd_out", h->bwd, [&] {
        for (Mod& md : h->mods) {
            if (md.conv && md.cdec[3].dense_map) {      // the 28x28 map is a dense row: an ordinary dense dgrad, sigmoid' of the stored map
                const ConvStage& s3 = md.cdec[3];
                WorkItem w = bd.dgrad_hidden(md.dO, md.outl, s3.Y, s3.dY);
                w.act = s3.act;
                h->items.push_back(w);
            }
            else if (md.conv) h->items.push_back(bd.conv_dgrad(md.cdec[4], md.dO));
            else h->items.push_back(bd.dgrad_hidden(md.dO, md.outl, md.D.back(), md.dD.back()));
        }
    });
    bool latent_conv_impl = false;          // some conv modality's first decoder stage takes the implicit latent dgrad (rides in bwd_dec1_latent)
    if (any_conv) {
        for (int i = 4; i >= 1; --i) {       // stage i's patch gradients -> dY of stage i-1; then stage i-1's input gradient
            // (per modality; a direct, adjoint-frame or implicit stage has no patch gradients: its own launch wrote the producing stage's gradient)
            auto explicit_i = [&, i](const Mod& md) { return !md.cdec[i].thin && !md.cdec[i].adj && !md.cdec[i].impl_bwd && !(i == 4 && md.cdec[3].dense_map); };
            auto plain_j = [&, i](const Mod& md) { return !md.cdec[i - 1].thin && !md.cdec[i - 1].adj && !md.cdec[i - 1].impl_bwd; };
            col2im_launch("conv_dec" + std::to_string(i) + "_col2im", h->bwd, [&](Mod& md) -> const ConvStage& { return md.cdec[i]; },
                          [&](Mod& md) -> const ConvStage* { return &md.cdec[i - 1]; }, explicit_i);
            if (is_thin(i - 1)) thin_launch("conv_bwd_dec" + std::to_string(i) + "_direct", h->bwd, 1, i - 1);
            if (is_adj(i - 1)) {     // patch matrix of stage i-1's output gradient -> GEMM with the adjoint filter -> dY of stage i-2
                Launch L;
                L.name = "conv_bwd_dec" + std::to_string(i) + "_adj_im2col"; L.type = 1;
                int base = 0;
                for (Mod& md : h->mods) if (md.conv && md.cdec[i - 1].adj && (!md.cdec[i - 1].impl_bwd || !md.cdec[i - 1].impl_w)) {      // (Padj feeds the adjoint dgrad and the adjoint-frame filter gradient)
                    GatherSeg g = bd.adj_gather_seg(md.cdec[i - 1]);
                    g.tile_base = base; base += g.tiles_r * g.tiles_c;
                    L.ga.seg[L.ga.n_seg++] = g;
                }
                L.blocks = base;
                if (base > 0) h->bwd.push_back(L);
                group("conv_bwd_dec" + std::to_string(i) + "_adj", h->bwd, [&] {
                    for (Mod& md : h->mods) if (md.conv && md.cdec[i - 1].adj && !md.cdec[i - 1].impl_bwd) h->items.push_back(bd.adj_dgrad(md.cdec[i - 1], md.cdec[i - 2])); });
            }
            group("conv_bwd_dec" + std::to_string(i), h->bwd, [&] {
                for (Mod& md : h->mods) if (md.conv) {
                    const ConvStage& st = md.cdec[i - 1];
                    if (st.impl_bwd && i - 1 >= 1)          // implicit GEMM on stage i-1's own output gradient: writes dY of stage i-2
                        h->items.push_back(bd.adj_dgrad_impl(st, h->at<void>(st.dY.rm), st.g.OH * st.g.OW * st.dY.ld, st.dY.ld, md.cdec[i - 2].act, md.cdec[i - 2].Y, md.cdec[i - 2].dY));
                    else if (st.impl_bwd) latent_conv_impl = true;
                    else if (plain_j(md)) h->items.push_back(bd.conv_dgrad(st, st.dY));
                } });
        }
        // first decoder stage: its input is z -> latent mode turns dz into [dmu | dlv]
        col2im_launch("conv_dec1_latent", h->bwd, [&](Mod& md) -> const ConvStage& { return md.cdec[0]; },
                      [&](Mod&) -> const ConvStage* { return nullptr; }, [](const Mod& md) { return !md.cdec[0].impl_bwd; });
    }
    for (int k = Lmax - 1; k >= 1; --k)
        group("bwd_dec" + std::to_string(k + 1), h->bwd, [&] {
            for (Mod& md : h->mods) if (k < md.L) h->items.push_back(bd.dgrad_hidden(md.dD[k], md.dec[k], md.D[k - 1], md.dD[k - 1]));
        });
    group("bwd_dec1_latent", h->bwd, [&] {
        for (Mod& md : h->mods) {
            if (!md.conv) h->items.push_back(bd.dgrad_latent(md));
            else if (md.cdec[0].impl_bwd) h->items.push_back(bd.adj_dgrad_latent_impl(md.cdec[0], md));
        }
        h->items.push_back(bd.cost(true));                       // every cost partial is final since fwd_out_loss: cost, step counter, lr_t
    });
    (void)latent_conv_impl;
    h->bwd_split = (int)h->bwd.size();      // everything from here on belongs to the encoder side
    group("bwd_head", h->bwd, [&] {
        for (Mod& md : h->mods) {
            if (md.conv && md.cenc[3].impl_bwd)     // implicit: the heads' input gradient straight into the 3x3 map's gradient rows
                h->items.push_back(bd.adj_dgrad_impl(md.cenc[3], h->at<void>(md.dH.rm), md.dH.ld, md.dH.ld, md.cenc[2].act, md.cenc[2].Y, md.cenc[2].dY));
            else if (md.conv) h->items.push_back(bd.conv_dgrad(md.cenc[3], md.dH));
            else h->items.push_back(bd.dgrad_hidden(md.dH, md.head, md.E.back(), md.dE.back()));
        }
    });
    if (!any_conv) fuse_tail(h->bwd, 2);                                      // bwd_head rides in bwd_dec1_latent
    if (any_conv) {
        for (int i = 3; i >= 1; --i) {
            col2im_launch("conv_enc" + std::to_string(i) + "_col2im", h->bwd, [&](Mod& md) -> const ConvStage& { return md.cenc[i]; },
                          [&](Mod& md) -> const ConvStage* { return &md.cenc[i - 1]; }, [i](const Mod& md) { return !md.cenc[i].impl_bwd; });
            if (i >= 2) group("conv_bwd_enc" + std::to_string(i), h->bwd, [&] {
                for (Mod& md : h->mods) if (md.conv) {
                    const ConvStage& st = md.cenc[i - 1];
                    if (st.impl_bwd) h->items.push_back(bd.adj_dgrad_impl(st, h->at<void>(st.dY.rm), st.g.OH * st.g.OW * st.dY.ld, st.dY.ld, md.cenc[i - 2].act, md.cenc[i - 2].Y, md.cenc[i - 2].dY));
                    else h->items.push_back(bd.conv_dgrad(st, st.dY));
                } });
        }
    }
    for (int k = Lmax - 1; k >= 1; --k) {
        group("bwd_enc" + std::to_string(k + 1), h->bwd, [&] {
            for (Mod& md : h->mods) if (k < md.L) h->items.push_back(bd.dgrad_hidden(md.dE[k], md.enc[k], md.E[k - 1], md.dE[k - 1]));
        });
    }
    // ---- mixed models: the MLP modalities' hidden-layer launches (a few tiles, 4-5 us each as launches of their own) ride in plain
    // GEMM launches of the conv modality.  Taken from the end of each chain backwards, a launch sinks into the LATEST such launch
    // that still runs before the first reader of its outputs -- so a chain enc1 -> enc2 ends up in two consecutive conv launches.
    // Safe: every buffer is written once per step, so a product may run any time between its inputs' writers and its readers.
    if (any_conv && !std::getenv("AVAE_NO_SINK")) {
        auto plain_kind = [](int k) { return k == K_FWD_HIDDEN || k == K_DGRAD_HIDDEN || k == K_DGRAD_F32; };
        // only launches of the MLP modalities move: an item that writes a conv modality's buffer stays where the plan put it (its readers
        // include helper launches -- the direct stage, im2col / col2im -- that the dependency scan below does not look into)
        std::vector<const void*> conv_bufs;
        for (const Mod& md : h->mods) if (md.conv) {
            for (const Act* a : {&md.Z, &md.dH, &md.dO}) conv_bufs.push_back(h->at<void>(a->rm));
            for (const std::vector<ConvStage>* side : {&md.cenc, &md.cdec})
                for (const ConvStage& st : *side) { conv_bufs.push_back(h->at<void>(st.Y.rm)); conv_bufs.push_back(h->at<void>(st.dY.rm)); }
        }
        auto mlp_item = [&](const WorkItem& w) {
            if (w.conv > 0) return false;
            for (const void* q : conv_bufs) if (q == w.out0 || q == w.A) return false;
            return true;
        };
        auto reads = [&](const Launch& L, const std::vector<const void*>& outs) {
            if (L.type != 0) return false;                  // the conv helper launches touch the conv modality's buffers only
            for (int i = 0; i < L.count; ++i) {
                const WorkItem& w = h->items[L.first + i];
                const void* csrc = w.conv > 0 ? h->conv_tab[w.conv - 1].src : nullptr;      // (an implicit patch matrix reads its source map)
                for (const void* q : {w.A, w.B, w.aux0, w.aux1, w.aux2, (const void*)w.tail_w, (const void*)w.tail_aux, csrc})
                    for (const void* o : outs) if (q && q == o) return true;
            }
            return false;
        };
        auto sink = [&](std::vector<Launch>& ls, int* index_a, int* index_b) {
            for (int a = (int)ls.size() - 1; a >= 0; --a) {
                Launch& P = ls[a];
                if (P.type != 0 || P.tn || P.name.compare(0, 5, "conv_") == 0 || P.name.find('+') != std::string::npos) continue;
                bool movable = true;
                std::vector<const void*> outs;
                for (int i = 0; i < P.count; ++i) {
                    const WorkItem& w = h->items[P.first + i];
                    movable = movable && (w.kind == K_FWD_HIDDEN || w.kind == K_DGRAD_HIDDEN) && !w.bias_ep && mlp_item(w);
                    outs.push_back(w.out0);
                }
                if (!movable) continue;
                int target = -1;
                for (int q = a + 1; q < (int)ls.size(); ++q) {
                    const Launch& Q = ls[q];
                    if (reads(Q, outs)) break;
                    bool ok = Q.type == 0 && !Q.tn && Q.name.compare(0, 5, "conv_") == 0 && Q.count + P.count <= kMaxItemsPerLaunch && Q.cfg != 2 && Q.cfg != 6;
                    for (int i = 0; ok && i < Q.count; ++i) ok = plain_kind(h->items[Q.first + i].kind);
                    if (ok) target = q;
                }
                if (target < 0) continue;
                Launch& Q = ls[target];
                const int first = (int)h->items.size();
                for (int i = 0; i < Q.count; ++i) { const WorkItem w = h->items[Q.first + i]; h->items.push_back(w); }
                for (int i = 0; i < P.count; ++i) { const WorkItem w = h->items[P.first + i]; h->items.push_back(w); }
                int dummy_slot = 0;                          // (no kind in here owns cost-partial slots)
                Launch merged = finish_launch(h, h->items, first, Q.count + P.count, Q.name + "+" + P.name, &dummy_slot);
                ls[target] = merged;
                ls.erase(ls.begin() + a);
                for (int* ix : {index_a, index_b}) if (ix && a < *ix) --*ix;
            }
        };
        // ... and where no such launch lies between a launch and its reader (the backward pass: the MLP modality's dgrad layers come
        // after the conv decoder's), it is HOISTED into the earliest one behind the writer of its input gradient.
        auto writes = [&](const Launch& L, const std::vector<const void*>& ins) {
            if (L.type != 0) return false;
            for (int i = 0; i < L.count; ++i) {
                const WorkItem& w = h->items[L.first + i];
                for (const void* q : {(const void*)w.out0, (const void*)w.out1, (const void*)w.tail_out})
                    for (const void* o : ins) if (q && q == o) return true;
            }
            return false;
        };
        auto hoist = [&](std::vector<Launch>& ls, int* index_a) {
            for (int a = 0; a < (int)ls.size(); ++a) {
                Launch& P = ls[a];
                if (P.type != 0 || P.tn || P.name.compare(0, 5, "conv_") == 0 || P.name.find('+') != std::string::npos) continue;
                bool movable = true;
                std::vector<const void*> ins;
                for (int i = 0; i < P.count; ++i) {
                    const WorkItem& w = h->items[P.first + i];
                    movable = movable && (w.kind == K_FWD_HIDDEN || w.kind == K_DGRAD_HIDDEN) && !w.bias_ep && mlp_item(w);
                    ins.push_back(w.A);
                }
                if (!movable) continue;
                int writer = -1;
                for (int q = 0; q < a; ++q) if (writes(ls[q], ins)) writer = q;
                if (writer < 0) continue;                    // (its input comes from the other list: leave it where the plan put it)
                int target = -1;
                for (int q = writer + 1; q < a && target < 0; ++q) {
                    const Launch& Q = ls[q];
                    bool ok = Q.type == 0 && !Q.tn && Q.name.compare(0, 5, "conv_") == 0 && Q.count + P.count <= kMaxItemsPerLaunch && Q.cfg != 2 && Q.cfg != 6;
                    for (int i = 0; ok && i < Q.count; ++i) ok = plain_kind(h->items[Q.first + i].kind);
                    if (ok) target = q;
                }
                if (target < 0) continue;
                Launch& Q = ls[target];
                const int first = (int)h->items.size();
                for (int i = 0; i < Q.count; ++i) { const WorkItem w = h->items[Q.first + i]; h->items.push_back(w); }
                for (int i = 0; i < P.count; ++i) { const WorkItem w = h->items[P.first + i]; h->items.push_back(w); }
                int dummy_slot = 0;
                Launch merged = finish_launch(h, h->items, first, Q.count + P.count, Q.name + "+" + P.name, &dummy_slot);
                ls[target] = merged;
                ls.erase(ls.begin() + a);
                if (index_a && a < *index_a) --*index_a;
                --a;
            }
        };
        sink(h->fwd, &h->fwd_dec_first, nullptr);
        sink(h->bwd, &h->bwd_split, nullptr);
        hoist(h->bwd, &h->bwd_split);
    }
    // ---- every weight gradient in the last launch(es) of the step: they depend only on stored activations /
    // activation gradients, and nothing reads the weights after them, so k_adam follows directly (after the all-reduce
    // under data parallelism).  Fusing Adam into these epilogues was measured and dropped: equal on the small nets
    // (C2 91.5 vs 91.2 us/step), slower on the big ones (C4 958 vs 922: a fused tile streams 448 KB and stalls the
    // MFMA loops it alternates with).
    {
        std::vector<WorkItem> wg;
        for (Mod& md : h->mods) {
            if (md.conv) {
                if (md.cdec[3].dense_map) wg.push_back(bd.wgrad(md.cdec[3].Y, md.cdec[4].d, md.dO));       // the dense output layer on the dense map
                else wg.push_back(bd.wgrad_stage(md.cdec[4], md.dO));
                for (int i = 3; i >= 0; --i) {
                    const ConvStage& st = md.cdec[i];
                    if (st.thin) continue;
                    if (st.impl_w) wg.push_back(bd.wgrad_impl(st, i == 0 ? h->at<void>(md.Z.rm) : h->at<void>(md.cdec[i - 1].Y.rm), st.dY));
                    else if (st.adj) wg.push_back(bd.adj_wgrad(st, md.cdec[i - 1]));
                    else wg.push_back(bd.wgrad_stage(st, st.dY));
                }
                if (md.cenc[3].impl_w) wg.push_back(bd.wgrad_impl(md.cenc[3], h->at<void>(md.cenc[2].Y.rm), md.dH));
                else wg.push_back(bd.wgrad_stage(md.cenc[3], md.dH));
                for (int i = 2; i >= 0; --i) {
                    const ConvStage& st = md.cenc[i];
                    if (st.impl_w) wg.push_back(bd.wgrad_impl(st, i == 0 ? h->at<void>(md.X0.rm) : h->at<void>(md.cenc[i - 1].Y.rm), st.dY));
                    else wg.push_back(bd.wgrad_stage(st, st.dY));
                }
                continue;
            }
            wg.push_back(bd.wgrad(md.D.back(), md.outl, md.dO));
            for (int k = md.L - 1; k >= 1; --k) wg.push_back(bd.wgrad(md.D[k - 1], md.dec[k], md.dD[k]));
            wg.push_back(bd.wgrad(md.Z, md.dec[0], md.dD[0]));
            wg.push_back(bd.wgrad(md.E.back(), md.head, md.dH));
            for (int k = md.L - 1; k >= 1; --k) wg.push_back(bd.wgrad(md.E[k - 1], md.enc[k], md.dE[k]));
            wg.push_back(bd.wgrad(md.X0, md.enc[0], md.dE[0]));
        }
        // big problems: the narrow products (heads, first decoder layer) get launches of their own, or their presence
        // would hold the wide ones on 64x64 tiles (finish_launch picks one tile shape per launch)
        auto is_narrow = [](const WorkItem& w) { return w.N <= 64 || w.M <= 64 || w.conv > 0; };    // (implicit patch matrices: small tiles only)
        auto wgrad_launches = [&](const std::vector<WorkItem>& set, std::vector<Launch>& dst, const std::string& prefix) {
            long wide128 = 0;
            for (const WorkItem& w : set) if (!is_narrow(w)) wide128 += (long)((w.M + 127) / 128) * ((w.N + 127) / 128);
            std::vector<std::vector<WorkItem>> chunks;
            auto chunk_up = [&](const std::vector<WorkItem>& v) {
                for (size_t i0 = 0; i0 < v.size(); i0 += kMaxTnItems)
                    chunks.emplace_back(v.begin() + i0, v.begin() + std::min(v.size(), i0 + (size_t)kMaxTnItems));
            };
            if (wide128 >= 192) {
                std::vector<WorkItem> wide, narrow;
                for (const WorkItem& w : set) (is_narrow(w) ? narrow : wide).push_back(w);
                chunk_up(wide); chunk_up(narrow);
            } else {
                chunk_up(set);
            }
            for (size_t c = 0; c < chunks.size(); ++c) {
                group(prefix + (chunks.size() > 1 ? std::to_string(c + 1) : std::string()), dst,
                      [&] { for (const WorkItem& w : chunks[c]) h->items.push_back(w); });
            }
        };
        wgrad_launches(wg, h->wgrad, "wgrad");
        // the same weight gradients cut into the two data-parallel buckets (MLP-only models)
        h->wgrad_b[0].clear(); h->wgrad_b[1].clear();
        dp_ranges(h, &h->n_buckets, h->ranges_b);
        if (h->n_buckets == 2) {
            std::vector<WorkItem> set[2];
            for (Mod& md : h->mods) {
                set[0].push_back(bd.wgrad(md.D.back(), md.outl, md.dO));
                for (int k = md.L - 1; k >= 1; --k) set[0].push_back(bd.wgrad(md.D[k - 1], md.dec[k], md.dD[k]));
                set[0].push_back(bd.wgrad(md.Z, md.dec[0], md.dD[0]));
                set[1].push_back(bd.wgrad(md.E.back(), md.head, md.dH));
                for (int k = md.L - 1; k >= 1; --k) set[1].push_back(bd.wgrad(md.E[k - 1], md.enc[k], md.dE[k]));
                set[1].push_back(bd.wgrad(md.X0, md.enc[0], md.dE[0]));
            }
            wgrad_launches(set[0], h->wgrad_b[0], "wgrad_dec");
            wgrad_launches(set[1], h->wgrad_b[1], "wgrad_enc");
            long wide = 0;
            for (const WorkItem& w : wg) if (!is_narrow(w)) wide += (long)((w.M + 127) / 128) * ((w.N + 127) / 128);
            // Measured and switched off (AVAE_OVERLAP=1 turns it on, results are bitwise the same): C2 95.8 vs 64.1 us/step, C4 0.616
            // vs 0.587 ms -- in the single-replica graphs the fork / join edges cost far more than the ~8 us of Adam they hide (the
            // data-parallel graph, where the side stream carries the RCCL kernels, does gain: 71 us with the extra collective work).
            h->overlap = std::getenv("AVAE_OVERLAP") != nullptr;
            // small nets: the weight-gradient launch is a fraction of a round of the chip and splits for free; big nets keep it whole
            // (C4: 134 us as one launch, 111 + 119 as two) and only Adam is cut
            h->ov_split_wgrad = wide < 192;
        }
        for (int i = 1; i <= 3; ++i) if (is_thin(i)) thin_launch("conv_dec" + std::to_string(i + 1) + "_wgrad_direct", h->wgrad, 2, i);
        {   // bias gradients of the adjoint-frame stages = column sums of their output gradient, first level
            Launch L;
            L.name = "conv_bias_rowsum"; L.type = 8;
            int base = 0;
            for (Mod& md : h->mods) if (md.conv) for (const ConvStage& st : md.cdec) if (st.adj && !st.impl_w) {
                RowsumSeg& g = L.rs.seg[L.rs.n_seg++];
                g.src = h->at<void>(st.dY.rm); g.part = h->at<float>(st.rs_part); g.ld = st.dY.ld;
                g.rows = h->B * st.g.OH * st.g.OW; g.cols = st.d.out; g.cols4 = st.rs_cols4; g.n_blocks = kRowsumBlocks;
                g.cpow = 4; while (g.cpow < g.cols4) g.cpow *= 2;
                if (g.cpow > 64) throw Err("internal error: bias row sums hold at most 64 columns");
                g.block_base = base; base += kRowsumBlocks;
            }
            L.blocks = base;
            if (base > 0) h->wgrad.push_back(L);
        }
        {   // ... and its per-image partial sums -> the gradient buffer (one column of the padded matrix)
            Launch R;
            R.name = "conv_wgrad_sums"; R.type = 5;
            int base = 0;
            for (Mod& md : h->mods) if (md.conv) for (const ConvStage& st : md.cdec) if (st.adj && !st.impl_w) {      // second level of the row sums
                ReduceSeg& g = R.ra.seg[R.ra.n_seg++];
                g.dst = h->grad() + st.d.master + (size_t)st.d.in * st.d.ld; g.src = h->at<float>(st.rs_part);
                g.n = st.d.out; g.parts = kRowsumBlocks; g.stride = st.rs_cols4; g.dst_ld = 1;
                g.block_base = base; base += (g.n + 3) / 4;
            }
            for (Mod& md : h->mods) if (md.conv) for (const ConvStage& st : md.cdec) if (st.thin) {
                ReduceSeg& g = R.ra.seg[R.ra.n_seg++];
                g.dst = h->grad() + st.d.master; g.src = h->at<float>(st.thin_part);
                g.n = st.d.in + 1; g.parts = st.thin_blocks; g.stride = st.thin_kp; g.dst_ld = st.d.ld;
                g.block_base = base; base += (g.n + 3) / 4;
            }
            R.blocks = base;
            for (int i = 0; i < R.ra.n_seg; ++i) { R.ra.seg[i].colsum = 1; R.ra.seg[i].perm_k = 0; }
            if (base > 0) h->wgrad.push_back(R);
        }
        {   // the split-K slices of the conv stages' weight gradients -> the gradient buffer, in fixed order
            Launch R;
            R.name = "wgrad_reduce"; R.type = 3;
            int base = 0;
            // (the column sums above write other destinations: when both exist they share one launch, k_sums)
            Launch* prev = (!h->wgrad.empty() && h->wgrad.back().type == 5 && !std::getenv("AVAE_NO_SUMS_MERGE")) ? &h->wgrad.back() : nullptr;
            if (prev) { R = *prev; R.name = prev->name + "+wgrad_reduce"; R.type = 9; base = prev->blocks; }
            for (const WorkItem& w : wg) if (w.ksplit > 1) {
                if (R.ra.n_seg >= kMaxReduceSegs) throw Err("internal error: too many split weight gradients");
                ReduceSeg& g = R.ra.seg[R.ra.n_seg++];
                g.dst = reinterpret_cast<float*>(w.out0); g.src = reinterpret_cast<const float*>(w.out1);
                g.n = w.M * w.ld0; g.parts = w.ksplit; g.stride = (long long)w.M * w.ld0;
                g.colsum = 0; g.perm_k = 0;
                // an adjoint-frame filter gradient: its slice sums go straight into the gradient buffer's layout (no k_gperm pass)
                if (!std::getenv("AVAE_NO_SUMS_MERGE"))
                    for (Mod& md : h->mods) if (md.conv) for (const ConvStage& st : md.cdec) if (st.adj && st.aksplit > 1 && g.dst == h->at<float>(st.Gadj)) {
                        g.dst = h->grad() + st.d.master;
                        g.perm_k = st.g.k; g.perm_cin = st.g.Cin; g.perm_cout = st.d.out; g.perm_ldga = st.ldga; g.perm_ld = st.d.ld;
                    }
                g.block_base = base; base += (g.n / 4 + kThreads - 1) / kThreads;
            }
            R.blocks = base;
            if (prev && base > prev->blocks) *prev = R;              // merged: replaces the column-sum launch
            else if (!prev && base > 0) h->wgrad.push_back(R);
        }
        {   // adjoint-frame filter gradients -> the gradient buffer's layout
            Launch L;
            L.name = "conv_wgrad_perm"; L.type = 7;
            int base = 0;
            for (Mod& md : h->mods) if (md.conv) for (const ConvStage& st : md.cdec) if (st.adj && !st.impl_w) {
                if (st.aksplit > 1 && !std::getenv("AVAE_NO_SUMS_MERGE")) continue;      // permuted by its slice sum (ReduceSeg::perm_*)
                GpermSeg& g = L.gp.seg[L.gp.n_seg++];
                g.Gadj = h->at<float>(st.Gadj); g.G = h->grad() + st.d.master; g.ldga = st.ldga; g.ld = st.d.ld;
                g.k = st.g.k; g.Cin = st.g.Cin; g.Cout = st.d.out;
                g.block_base = base; base += (st.g.Cin * st.g.k * st.g.k * st.d.out + kThreads - 1) / kThreads;
            }
            L.blocks = base;
            if (base > 0) h->wgrad.push_back(L);
        }
    }
    // ---- small nets: the optimiser in the epilogue of the (single) weight-gradient launch.  Only for the plain single-replica step
    // (under data parallelism the all-reduce sits between the two; avae_dp_backward / avae_dp_apply keep them apart as well).
    h->wgrad_adam.clear();
    if (!any_conv && h->wgrad.size() == 1 && h->wgrad[0].tn && h->wgrad[0].cfg == 0 && h->wgrad[0].targs.xcd_group == 1
        && !std::getenv("AVAE_NO_LEAN") && !std::getenv("AVAE_NO_ADAM_FUSE")) {
        Launch L = h->wgrad[0];
        bool ok = true;
        for (int i = 0; i < L.targs.n_items && ok; ++i) {
            TnItem& t = L.targs.items[i];
            ok = t.ksplit <= 1 && t.bias_row == 0 && t.K > 0;
            const Dense* dm = nullptr;
            for (const Mod& md : h->mods) {
                auto look = [&](const Dense& d) { if (h->grad() + d.master == t.out) dm = &d; };
                for (const Dense& d : md.enc) look(d);
                look(md.head);
                for (const Dense& d : md.dec) look(d);
                look(md.outl);
            }
            ok = ok && dm && t.M == dm->in + 1 && t.N == dm->out && t.ld0 == dm->ld;
            if (ok) { t.W = h->at<void>(dm->W); t.Wt = h->at<void>(dm->Wt); t.ldw = dm->ldw; t.ldt = dm->ldt; }
        }
        if (ok) {
            L.cfg = 12; L.name = "wgrad+adam";
            L.targs.adam = 1;
            L.targs.d_theta = h->at<float>(h->off_theta) - h->grad();
            L.targs.d_m = h->at<float>(h->off_m) - h->grad();
            L.targs.d_v = h->at<float>(h->off_v) - h->grad();
            L.targs.beta1 = h->cfg.beta1; L.targs.beta2 = h->cfg.beta2; L.targs.eps = h->cfg.adam_eps;
            L.targs.st = h->state();
            if (!std::getenv("AVAE_NO_XCD_PIECES")) xcd_pieces(L);
            h->wgrad_adam.push_back(L);
        }
    }
    // ---- weight warm-up: a launch on the 8-wave NT tiles pulls the weight panels of the next grouped NT launch into the Infinity
    // Cache (see k_grouped); forward and backward form one sequence (the step graph runs them back to back)
    if (!std::getenv("AVAE_NO_WARM")) {
        std::vector<Launch*> seq;
        for (Launch& L : h->fwd) seq.push_back(&L);
        for (Launch& L : h->bwd) seq.push_back(&L);
        for (size_t i = 0; i < seq.size(); ++i) {
            Launch& L = *seq[i];
            if (L.type != 0 || L.tn || (L.cfg != 2 && L.cfg != 6)) continue;
            for (size_t j = i + 1; j < seq.size(); ++j) {
                const Launch& Nx = *seq[j];
                if (Nx.type != 0) continue;                 // helper launches of the conv branch in between
                if (Nx.tn) break;
                for (int k = 0; k < Nx.args.n_items && L.args.n_pf < kMaxPf; ++k) {
                    const WorkItem& w = Nx.args.items[k];
                    if (!is_gemm(w.kind) || w.kind == K_WGRAD) continue;
                    const long bytes = (long)w.N * w.ldb * h->es;          // B = weight shadow: N rows of ldb elements
                    if (bytes < 128 * 1024) continue;                     // small panels come with the first tile anyway
                    L.args.pf_ptr[L.args.n_pf] = w.B; L.args.pf_lines[L.args.n_pf] = (int)(bytes / 128); ++L.args.n_pf;
                }
                break;
            }
        }
    }
    // ---- eval: forward launches + a lone cost reduction that does not bump the step
    {
        const int first = (int)h->items.size();
        h->items.push_back(bd.cost(false));
        h->cost_only = finish_launch(h, h->items, first, 1, "cost_reduce", &slot);
    }

    // ---- Adam tiles
    h->adam_items.clear();
    int base = 0;
    auto add = [&](const Dense& d) {
        AdamItem a;
        std::memset(&a, 0, sizeof(a));
        a.theta = h->at<float>(h->off_theta) + d.master;
        a.m = h->at<float>(h->off_m) + d.master;
        a.v = h->at<float>(h->off_v) + d.master;
        a.g = h->grad() + d.master;
        a.W = h->at<void>(d.W); a.Wt = h->at<void>(d.Wt);
        a.rows = d.in + 1; a.cols = d.out; a.ld = d.ld; a.ldw = d.ldw; a.ldt = d.ldt;
        a.tiles_r = (a.rows + kAdamRows - 1) / kAdamRows; a.tiles_c = (a.cols + 63) / 64; a.tile_base = base;
        base += a.tiles_r * a.tiles_c;
        h->adam_items.push_back(a);
    };
    for (const Mod& md : h->mods) {
        if (md.conv) {
            for (const std::vector<ConvStage>* side : {&md.cenc, &md.cdec})
                for (const ConvStage& st : *side) {
                    add(st.d);
                    // adjoint filter shadows by the same pass (no k_wadj launch): the implicit input gradients' B operand, and -- round 2's
                    // adjoint-frame route -- the forward scatter product's
                    if ((st.adj && !std::getenv("AVAE_NO_WADJ_FOLD")) || st.impl_bwd) {
                        AdamItem& a = h->adam_items.back();
                        a.Wadj = h->at<void>(st.Wadj); a.ldadj = st.ldadj; a.adj_k = st.g.k; a.adj_cin = st.g.Cin;
                        if (st.adj) { a.Wf = h->at<void>(st.Wf); a.ldf = st.ldf; }
                    }
                }
            continue;
        }
        for (const Dense& d : md.enc) add(d);
        add(md.head);
        for (const Dense& d : md.dec) add(d);
        add(md.outl);
    }
    h->adam_blocks = base;
    // bucket-ordered copy: per modality the items are [enc x L, head, dec x L, out]
    h->adam_items_b.clear();
    for (int bk = 0; bk < 2; ++bk) {
        h->adam_first_b[bk] = (int)h->adam_items_b.size();
        int tb = 0;
        size_t i = 0;
        for (const Mod& md : h->mods) {
            const size_t n_enc = md.conv ? md.cenc.size() : (size_t)md.L + 1, n_all = md.conv ? md.cenc.size() + md.cdec.size() : 2 * (size_t)md.L + 2;
            for (size_t k = 0; k < n_all; ++k) {
                const int which = h->n_buckets == 1 ? 0 : (k < n_enc ? 1 : 0);
                if (which == bk) {
                    AdamItem a = h->adam_items[i + k];
                    a.tile_base = tb; tb += a.tiles_r * a.tiles_c;
                    h->adam_items_b.push_back(a);
                }
            }
            i += n_all;
        }
        h->adam_count_b[bk] = (int)h->adam_items_b.size() - h->adam_first_b[bk];
        h->adam_blocks_b[bk] = tb;
    }
}

void build_inference(avae_handle* h, int m, bool enc, int rows) {
    avae_handle::Inf& inf = enc ? h->inf_enc[m] : h->inf_dec[m];
    if (inf.rows == rows) return;
    inf.items.clear(); inf.launches.clear();
    Builder bd(h, inf.items, rows, false);
    Mod& md = h->mods[m];
    int slot = 0;
    auto one = [&](const WorkItem& w, const std::string& name) {
        inf.items.push_back(w);
        inf.launches.push_back(finish_launch(h, inf.items, (int)inf.items.size() - 1, 1, name, &slot));
    };
    auto gather1 = [&](const ConvStage& st, const void* src, const std::string& name) {
        Launch L;
        L.name = name; L.type = 1;
        GatherSeg g = bd.gather_seg(st, src);
        g.tile_base = 0;
        L.ga.seg[0] = g; L.ga.n_seg = 1;
        L.blocks = g.tiles_r * g.tiles_c;
        inf.launches.push_back(L);
    };
    if (enc && md.conv) {
        for (int i = 0; i < 3; ++i) {
            gather1(md.cenc[i], i == 0 ? h->at<void>(md.X0.rm) : h->at<void>(md.cenc[i - 1].Y.rm), "inf_conv_im2col");
            one(bd.conv_fwd(md.cenc[i]), "inf_conv_enc");
        }
        gather1(md.cenc[3], h->at<void>(md.cenc[2].Y.rm), "inf_conv_im2col");
        one(bd.fwd_head(md, true), "inf_head");
    } else if (enc) {
        for (int k = 0; k < md.L; ++k) one(bd.fwd_hidden(k == 0 ? md.X0 : md.E[k - 1], md.enc[k], md.E[k]), "inf_enc");
        one(bd.fwd_head(md, true), "inf_head");
    } else if (md.conv) {
        for (int i = 0; i < 4; ++i) {
            if (md.cdec[i].dense_map) {      // the one-channel 28x28 stage writes dense rows: the direct kernel, as in training
                Launch L;
                L.name = "inf_conv_dec_direct"; L.type = 4;
                L.ta.mode = 0;
                ThinSeg t = bd.thin_seg(md.cdec[i], md.cdec[i - 1]);
                t.block_base = 0;
                L.ta.seg[0] = t; L.ta.n_seg = 1;
                L.blocks = rows * t.split;
                inf.launches.push_back(L);
                continue;
            }
            gather1(md.cdec[i], i == 0 ? h->at<void>(md.Z.rm) : h->at<void>(md.cdec[i - 1].Y.rm), "inf_conv_im2col");
            one(bd.conv_fwd(md.cdec[i]), "inf_conv_dec");
        }
        if (!md.cdec[3].dense_map) gather1(md.cdec[4], h->at<void>(md.cdec[3].Y.rm), "inf_conv_im2col");
        one(bd.fwd_out(md, m, false), "inf_out");
    } else {
        for (int k = 0; k < md.L; ++k) one(bd.fwd_hidden(k == 0 ? md.Z : md.D[k - 1], md.dec[k], md.D[k]), "inf_dec");
        one(bd.fwd_out(md, m, false), "inf_out");
    }
    inf.rows = rows;
}

// ----------------------------------------------------------------------------- execution
int tname_id(avae_handle* h, const std::string& n) {
    for (size_t i = 0; i < h->tnames.size(); ++i) if (h->tnames[i] == n) return (int)i;
    h->tnames.push_back(n);
    return (int)h->tnames.size() - 1;
}

struct Timed {
    avae_handle* h; hipStream_t s; bool on; hipEvent_t a = nullptr, b = nullptr; int id = 0;
    Timed(avae_handle* h_, hipStream_t s_, const std::string& name) : h(h_), s(s_), on(h_->timing && h_->trecs.size() < 200000) {
        if (!on) return;
        id = tname_id(h, name);
        HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
        t_launch_events = LaunchEvents{a, b};         // the launch inside this scope stamps them with its own begin/end
    }
    ~Timed() {
        if (!on) return;
        if (t_launch_events.start) {                  // nothing was launched (empty segment): record an empty bracket
            t_launch_events = LaunchEvents{nullptr, nullptr};
            (void)hipEventRecord(a, s); (void)hipEventRecord(b, s);
        }
        h->trecs.push_back(TimingRec{a, b, id});
    }
};

void run_launches(avae_handle* h, const std::vector<Launch>& ls, hipStream_t s, int stamp_base = -1) {
    int k = 0;
    static const char* dup = std::getenv("AVAE_DUP");      // diagnostics: run the named launch twice, the second one timed as "<name>_again"
    for (const Launch& L0 : ls) {
      for (int rep = 0; rep < ((dup && h->timing && L0.name == dup) ? 2 : 1); ++rep) {
        Launch Lr;
        if (rep) { Lr = L0; Lr.name += "_again"; }
        const Launch& L = rep ? Lr : L0;
        Timed t(h, s, L.name);
        unsigned long long* stamps = nullptr;
#ifdef AVAE_STAMPS
        if (stamp_base >= 0) stamps = h->at<unsigned long long>(h->off_stamps);
#endif
        if (L.type == 1) launch_gather(h->cfg.compute_dtype, L.ga, L.blocks, s);
        else if (L.type == 2) launch_col2im(h->cfg.compute_dtype, L.ca, L.blocks, s);
        else if (L.type == 3) launch_reduce(L.ra, L.blocks, s);
        else if (L.type == 4) launch_thin(h->cfg.compute_dtype, L.ta, L.blocks, s);
        else if (L.type == 5) launch_colsum(L.ra, L.blocks, s);
        else if (L.type == 6) launch_wadj(h->cfg.compute_dtype, L.wa, L.blocks, s);
        else if (L.type == 7) launch_gperm(L.gp, L.blocks, s);
        else if (L.type == 8) launch_rowsum(h->cfg.compute_dtype, L.rs, L.blocks, s);
        else if (L.type == 9) launch_sums(L.ra, L.blocks, s);
        else if (L.cfg == 13) launch_chain2(h->cfg.compute_dtype, L.args, L.grid_x, L.grid_y, L.lds, h->at<unsigned>(h->off_chain), h->at<unsigned>(h->off_chain) + kMaxMod * 64, s);
        else if (L.cfg == 11) launch_small_head(h->cfg.compute_dtype, L.lean_act, L.args, L.grid_x, L.grid_y, L.lds, s, stamps, stamp_base + k);
        else if (L.cfg == 10) launch_small_latb(h->cfg.compute_dtype, L.lean_act, L.args, L.grid_x, L.grid_y, L.lds, h->state(), s, stamps, stamp_base + k);
        else if (L.cfg == 9) launch_small_loss(h->cfg.compute_dtype, L.args, L.grid_x, L.grid_y, L.lds, s, stamps, stamp_base + k);
        else if (L.cfg == 7) launch_small(h->cfg.compute_dtype, L.args, L.grid_x, L.grid_y, L.lds, s, stamps, stamp_base + k);
        else if (L.tn && L.cfg == 12) launch_small_tn(h->cfg.compute_dtype, L.targs, L.grid_x, L.grid_y, s, stamps, stamp_base + k);
        else if (L.tn) launch_grouped_tn(h->cfg.compute_dtype, L.cfg, L.targs, L.grid_x, L.grid_y, L.lds, h->state(), s, stamps, stamp_base + k);
        else launch_grouped(h->cfg.compute_dtype, L.cfg, L.args, L.grid_x, L.grid_y, L.lds, h->state(), s, stamps, stamp_base + k);
        LAUNCH_OK(L.name);
        if (h->debug_sync) {      // AVAE_DEBUG_SYNC=1: name the launch a fault belongs to
            std::fprintf(stderr, "[avae] launch %s type=%d cfg=%d items=%d blocks=%d\n", L.name.c_str(), L.type, L.cfg, L.count, L.blocks);
            for (int i = 0; i < L.args.n_items && L.type == 0; ++i)
                std::fprintf(stderr, "        item kind=%d M=%d N=%d K=%d lda=%d ldb=%d tiles=%dx%d\n", L.args.items[i].kind, L.args.items[i].M,
                             L.args.items[i].N, L.args.items[i].K, L.args.items[i].lda, L.args.items[i].ldb, L.args.items[i].tiles_m, L.args.items[i].tiles_n);
            for (int i = 0; i < L.ga.n_seg && L.type == 1; ++i)
                std::fprintf(stderr, "        gather B=%d IH=%d Cin=%d OH=%d k=%d so=%d d=%d pad=%d sb=%d sp=%d ones=%d ldp=%d tiles=%dx%d\n", L.ga.seg[i].g.B, L.ga.seg[i].g.IH,
                             L.ga.seg[i].g.Cin, L.ga.seg[i].g.OH, L.ga.seg[i].g.k, L.ga.seg[i].g.so, L.ga.seg[i].g.d, L.ga.seg[i].g.pad, L.ga.seg[i].g.src_sb, L.ga.seg[i].g.src_sp,
                             L.ga.seg[i].g.ones, L.ga.seg[i].ldp, L.ga.seg[i].tiles_r, L.ga.seg[i].tiles_c);
            for (int i = 0; i < L.ca.n_seg && L.type == 2; ++i)
                std::fprintf(stderr, "        col2im B=%d IH=%d Cin=%d OH=%d k=%d lddp=%d lda=%d g0=%p tiles=%dx%d\n", L.ca.seg[i].g.B, L.ca.seg[i].g.IH, L.ca.seg[i].g.Cin,
                             L.ca.seg[i].g.OH, L.ca.seg[i].g.k, L.ca.seg[i].lddp, L.ca.seg[i].lda, (const void*)L.ca.seg[i].g0, L.ca.seg[i].tiles_r, L.ca.seg[i].tiles_c);
            std::fflush(stderr);
            HIP_OK(hipStreamSynchronize(s));
        }
      }
        ++k;
    }
}

void run_adam(avae_handle* h, int mode, hipStream_t s, int bucket = -1) {
    AdamArgs a;
    int blocks = h->adam_blocks;
    if (bucket < 0) {
        a.items = h->at<AdamItem>(h->off_adam);
        a.n_items = (int)h->adam_items.size();
        if (a.n_items > kMaxAdamItems) throw Err("internal error: too many Adam items");
        for (int i = 0; i < a.n_items; ++i) a.base[i] = h->adam_items[i].tile_base;
    } else {
        a.items = h->at<AdamItem>(h->off_adam_b) + h->adam_first_b[bucket];
        a.n_items = h->adam_count_b[bucket];
        for (int i = 0; i < a.n_items; ++i) a.base[i] = h->adam_items_b[h->adam_first_b[bucket] + i].tile_base;
        blocks = h->adam_blocks_b[bucket];
        if (blocks == 0) return;
    }
    a.book = (bucket <= 0) ? 1 : 0;            // the bucket that carries the cost slot books the step's cost
    a.mode = mode;
    a.lr = h->cfg.learning_rate; a.beta1 = h->cfg.beta1; a.beta2 = h->cfg.beta2; a.eps = h->cfg.adam_eps;
    a.st = h->state();
    a.cost_src = h->grad() + h->P_int;
    Timed t(h, s, mode == 0 ? (bucket < 0 ? "adam" : bucket == 0 ? "adam_dec" : "adam_enc") : "shadow_refresh");
    launch_adam(h->cfg.compute_dtype, a, blocks, s);
    LAUNCH_OK(mode == 0 ? "adam" : "shadow_refresh");
}

// stages the caller's batch (and eps) into the internal compute-dtype buffers
PrepArgs make_prep_batch(avae_handle* h, const float* const* x, const int32_t* x_ld, const float* eps, int rows,
                         unsigned long long salt, int n_steps = 1) {
    PrepArgs a;
    std::memset(&a, 0, sizeof(a));
    int base = 0;
    for (int m = 0; m < h->M; ++m) {
        const Mod& md = h->mods[m];
        PrepSeg& g = a.seg[m];
        g.src = x[m]; g.src_ld = (x_ld && x_ld[m] > 0) ? x_ld[m] : md.n_in;
        g.rows = rows; g.cols = md.n_in;
        g.dst32 = h->at<float>(md.X32); g.ld32 = md.ld32;
        g.dstc = h->at<void>(md.X0.rm); g.ldc = md.X0.ld;
        g.tiles_r = (rows + 63) / 64; g.tiles_c = (md.n_in + 63) / 64; g.tile_base = base;
        base += g.tiles_r * g.tiles_c;
    }
    a.n_seg = h->M; a.total_tiles = base;
    a.eps_src = eps; a.eps_dst = h->at<float>(h->off_eps); a.eps_rows = rows; a.nz = h->nz; a.eps_ld = h->ld_eps;
    a.eps_blocks = (rows * ((h->nz + 3) / 4) + kThreads - 1) / kThreads;
    a.row_offset = h->cfg.row_offset; a.seed = h->cfg.seed; a.st = h->state(); a.stream_salt = salt;
    a.n_steps = n_steps; a.blocks_per_step = a.total_tiles + a.eps_blocks; a.set_stride = (long long)h->stage_bytes;
    return a;
}

void run_prep_batch(avae_handle* h, const float* const* x, const int32_t* x_ld, const float* eps, int rows,
                    unsigned long long salt, hipStream_t s) {
    const PrepArgs a = make_prep_batch(h, x, x_ld, eps, rows, salt);
    Timed t(h, s, "prep");
    launch_prep(h->cfg.compute_dtype, a, s);
    LAUNCH_OK("prep");
}

void run_prep_single(avae_handle* h, const float* src, int src_ld, int rows, int cols, const Act& dst, float* dst32, int ld32,
                     bool do_eps, const float* eps, unsigned long long salt, hipStream_t s, int row0 = 0) {
    PrepArgs a;
    std::memset(&a, 0, sizeof(a));
    if (src) {
        PrepSeg& g = a.seg[0];
        g.src = src; g.src_ld = src_ld; g.rows = rows; g.cols = cols;
        g.dst32 = dst32; g.ld32 = ld32;
        g.dstc = h->at<void>(dst.rm); g.ldc = dst.ld;
        g.tiles_r = (rows + 63) / 64; g.tiles_c = (cols + 63) / 64; g.tile_base = 0;
        a.n_seg = 1; a.total_tiles = g.tiles_r * g.tiles_c;
    }
    if (do_eps) {
        a.eps_src = eps; a.eps_dst = h->at<float>(h->off_eps); a.eps_rows = rows; a.nz = h->nz; a.eps_ld = h->ld_eps;
        a.eps_blocks = (rows * ((h->nz + 3) / 4) + kThreads - 1) / kThreads;
    }
    a.row_offset = h->cfg.row_offset + row0; a.seed = h->cfg.seed; a.st = h->state(); a.stream_salt = salt;
    launch_prep(h->cfg.compute_dtype, a, s);
    LAUNCH_OK("prep");
}

// The same launch reading staging set j instead of set 0: every pointer into set 0 moves by j * stage_bytes.
Launch relocated(const avae_handle* h, const Launch& L0, int j) {
    if (j == 0) return L0;
    Launch L = L0;
    const unsigned char* lo = h->ws + h->stage_lo;
    const unsigned char* hi = lo + h->stage_bytes;
    const size_t delta = (size_t)j * h->stage_bytes;
    auto fix = [&](auto& ptr) {
        const unsigned char* q = reinterpret_cast<const unsigned char*>(ptr);
        if (q >= lo && q < hi) ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(const_cast<unsigned char*>(q) + delta);
    };
    for (int i = 0; i < L.args.n_items && L.type == 0; ++i) {
        WorkItem& w = L.args.items[i];
        fix(w.A); fix(w.B); fix(w.out0); fix(w.out1); fix(w.out2); fix(w.aux0); fix(w.aux1); fix(w.aux2); fix(w.eps);
        fix(w.tail_w); fix(w.tail_out); fix(w.tail_aux);
    }
    for (int i = 0; i < L.targs.n_items && L.type == 0; ++i) { TnItem& t = L.targs.items[i]; fix(t.A); fix(t.B); fix(t.out); }
    for (int i = 0; i < L.ga.n_seg && L.type == 1; ++i) fix(L.ga.seg[i].src);
    return L;
}

hipGraphExec_t capture(avae_handle* h, const std::function<void(hipStream_t)>& body) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIP_OK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    try { body(h->cap_stream); } catch (...) { (void)hipStreamEndCapture(h->cap_stream, &g); if (g) (void)hipGraphDestroy(g); throw; }
    HIP_OK(hipStreamEndCapture(h->cap_stream, &g));
    HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIP_OK(hipGraphDestroy(g));
    return ge;
}

// Same, for a graph whose first node is the input-staging kernel: keeps the template graph and that node so the
// caller's pointers can be patched in before a replay (hipGraphExecKernelNodeSetParams).
hipGraphExec_t capture_with_prep(avae_handle* h, const std::function<void(hipStream_t)>& body, hipGraph_t* graph_out,
                                 hipGraphNode_t* prep_node) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIP_OK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    try { body(h->cap_stream); }
    catch (...) { (void)hipStreamEndCapture(h->cap_stream, &g); if (g) (void)hipGraphDestroy(g); throw; }
    HIP_OK(hipStreamEndCapture(h->cap_stream, &g));
    try {
        size_t n_root = 0;
        HIP_OK(hipGraphGetRootNodes(g, nullptr, &n_root));
        if (n_root != 1) throw Err("internal error: the captured step is not a chain");
        HIP_OK(hipGraphGetRootNodes(g, prep_node, &n_root));
        hipGraphNodeType ty;
        HIP_OK(hipGraphNodeGetType(*prep_node, &ty));
        if (ty != hipGraphNodeTypeKernel) throw Err("internal error: the captured step does not start with the staging kernel");
        HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    } catch (...) { (void)hipGraphDestroy(g); throw; }
    *graph_out = g;
    return ge;
}

// Points one captured staging node at the caller's batch (or run of n_steps consecutive batches).
void patch_prep(avae_handle* h, hipGraphExec_t ge, hipGraphNode_t node, const float* const* x, const int32_t* x_ld, const float* eps,
                int n_steps = 1) {
    PrepArgs a = make_prep_batch(h, x, x_ld, eps, h->B, 0x7261696eull, n_steps);
    void* kp[1] = {&a};
    hipKernelNodeParams np;
    std::memset(&np, 0, sizeof(np));
    np.func = const_cast<void*>(prep_kernel(h->cfg.compute_dtype));
    np.gridDim = dim3((a.total_tiles + a.eps_blocks) * n_steps); np.blockDim = dim3(kThreads);
    np.sharedMemBytes = 0; np.kernelParams = kp; np.extra = nullptr;
    HIP_OK(hipGraphExecKernelNodeSetParams(ge, node, &np));
}

void fill_ones(avae_handle* h, const Act& a, hipStream_t s) {
    if (!a.ones) return;
    const unsigned bits = h->es == 2 ? 0x3F80u : 0x3F800000u;
    launch_fill(h->at<void>(a.rm), h->es, bits, a.width, a.ld, h->B, s);             // column `width`, rows < B
    LAUNCH_OK("fill");
}

void init_device(avae_handle* h) {
    hipStream_t s = h->cap_stream;
    HIP_OK(hipMemsetAsync(h->ws, 0, h->ws_bytes, s));
    for (const Mod& md : h->mods) {
        for (int j = 0; j < kMultiSteps; ++j) {       // every staging set carries its own constant-1 column
            Act x = md.X0;
            x.rm += (size_t)j * h->stage_bytes;
            fill_ones(h, x, s);
        }
        for (const Act& a : md.E) fill_ones(h, a, s);
        fill_ones(h, md.Z, s);
        for (const Act& a : md.D) fill_ones(h, a, s);
        if (md.conv && md.cdec[3].dense_map) fill_ones(h, md.cdec[3].Y, s);      // the dense 28x28 map feeds the output layer: its bias column
    }
    build_training_plan(h);
    HIP_OK(hipMemcpyAsync(h->at<void>(h->off_adam), h->adam_items.data(), h->adam_items.size() * sizeof(AdamItem), hipMemcpyHostToDevice, s));
    HIP_OK(hipMemcpyAsync(h->at<void>(h->off_adam_b), h->adam_items_b.data(), h->adam_items_b.size() * sizeof(AdamItem), hipMemcpyHostToDevice, s));
    {   // the two constant chunks an implicit patch matrix reads instead of an invalid tap / as its bias column, and the descriptors
        static const unsigned char c32_bf16[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x80, 0x3F};         // bf16 1.0 = 0x3F80, little endian
        static const unsigned char c32_f32[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x80, 0x3F};      // fp32 1.0 = 0x3F800000
        HIP_OK(hipMemcpyAsync(h->at<void>(h->off_consts), h->es == 2 ? c32_bf16 : c32_f32, 32, hipMemcpyHostToDevice, s));
        if (!h->conv_tab.empty()) {
            const unsigned char* lo = h->ws + h->stage_lo;
            for (const ConvA& c : h->conv_tab)
                if (reinterpret_cast<const unsigned char*>(c.src) >= lo && reinterpret_cast<const unsigned char*>(c.src) < lo + h->stage_bytes * kMultiSteps)
                    throw Err("internal error: an implicit patch matrix on a staged input (not relocatable per step)");
            HIP_OK(hipMemcpyAsync(h->at<void>(h->off_conv_tab), h->conv_tab.data(), h->conv_tab.size() * sizeof(ConvA), hipMemcpyHostToDevice, s));
        }
    }
    size_t off = h->off_inf;
    h->inf_enc.assign(h->M, avae_handle::Inf());
    h->inf_dec.assign(h->M, avae_handle::Inf());
    for (int m = 0; m < h->M; ++m) {
        h->inf_enc[m].dev_off = off; off += ((size_t)h->mods[m].L + 1) * sizeof(WorkItem);
        h->inf_dec[m].dev_off = off; off += ((size_t)h->mods[m].L + 1) * sizeof(WorkItem);
    }
    HIP_OK(hipStreamSynchronize(s));
    if (h->cfg.use_graph) {
        const bool tsave = h->timing;
        h->timing = false;
        std::vector<const float*> x0(h->M, h->at<float>(h->mods[0].X32));    // placeholders, patched per step
        if (h->overlap && h->n_buckets == 2) {
            HIP_OK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
        } else {
            h->overlap = false;
        }
        // one step on staging set j inside a capture; `pending` = the side stream still carries the previous step's decoder bucket
        auto step_body = [&](hipStream_t cs, int j, bool& pending, int stamp_base) {
            auto run = [&](const std::vector<Launch>& ls, size_t lo, size_t hi, hipStream_t st, int sb) {
                std::vector<Launch> moved;
                for (size_t i = lo; i < hi && i < ls.size(); ++i) moved.push_back(relocated(h, ls[i], j));
                run_launches(h, moved, st, sb);
            };
            if (!h->overlap) {
                run(h->fwd, 0, h->fwd.size(), cs, stamp_base); run(h->bwd, 0, h->bwd.size(), cs, stamp_base < 0 ? -1 : stamp_base + (int)h->fwd.size());
                if (!h->wgrad_adam.empty()) {       // the optimiser rides in the weight-gradient launch
                    run(h->wgrad_adam, 0, 1, cs, stamp_base < 0 ? -1 : stamp_base + (int)(h->fwd.size() + h->bwd.size()));
                    return;
                }
                run(h->wgrad, 0, h->wgrad.size(), cs, stamp_base < 0 ? -1 : stamp_base + (int)(h->fwd.size() + h->bwd.size()));
                run_adam(h, 0, cs);
                return;
            }
            run(h->fwd, 0, (size_t)h->fwd_dec_first, cs, -1);
            if (pending) { HIP_OK(hipStreamWaitEvent(cs, h->ev_side, 0)); pending = false; }      // decoder weights of the previous step are final
            run(h->fwd, (size_t)h->fwd_dec_first, h->fwd.size(), cs, -1);
            run(h->bwd, 0, (size_t)h->bwd_split, cs, -1);
            if (h->ov_split_wgrad) {
                HIP_OK(hipEventRecord(h->ev_fork, cs));
                HIP_OK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
                run(h->wgrad_b[0], 0, h->wgrad_b[0].size(), h->side_stream, -1);
                run_adam(h, 0, h->side_stream, 0);
                HIP_OK(hipEventRecord(h->ev_side, h->side_stream));
                run(h->bwd, (size_t)h->bwd_split, h->bwd.size(), cs, -1);
                run(h->wgrad_b[1], 0, h->wgrad_b[1].size(), cs, -1);
                run_adam(h, 0, cs, 1);
            } else {        // Adam of the encoder side first and alone (it is HBM-bound: two at once gain nothing), then the decoder side's beside
                            // the next step's encoder forward (MFMA-bound launches with register-file room for Adam's small waves)
                run(h->bwd, (size_t)h->bwd_split, h->bwd.size(), cs, -1);
                run(h->wgrad, 0, h->wgrad.size(), cs, -1);
                run_adam(h, 0, cs, 1);
                HIP_OK(hipEventRecord(h->ev_fork, cs));
                HIP_OK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
                run_adam(h, 0, h->side_stream, 0);
                HIP_OK(hipEventRecord(h->ev_side, h->side_stream));
            }
            pending = true;
        };
        auto one_step = [&](hipStream_t cs) {
            run_prep_batch(h, x0.data(), nullptr, nullptr, h->B, 0x7261696eull, cs);
            bool pending = false;
            step_body(cs, 0, pending, 0);
            if (pending) HIP_OK(hipStreamWaitEvent(cs, h->ev_side, 0));
        };
        h->g_full = capture_with_prep(h, one_step, &h->g_full_graph, &h->g_full_prep);
        // avae_train_steps: kMultiSteps whole steps per replay (a replay boundary costs ~5 us of idle GPU on this stack),
        // their batches staged by ONE launch into the kMultiSteps staging sets; step j's launches read set j
        for (int gi = 0; gi < 2; ++gi) h->g_multi[gi] = capture_with_prep(h, [&](hipStream_t cs) {
            const PrepArgs a = make_prep_batch(h, x0.data(), nullptr, nullptr, h->B, 0x7261696eull, kMultiSizes[gi]);
            launch_prep(h->cfg.compute_dtype, a, cs);
            LAUNCH_OK("prep");
            bool pending = false;
            for (int j = 0; j < kMultiSizes[gi]; ++j) step_body(cs, j, pending, -1);
            if (pending) HIP_OK(hipStreamWaitEvent(cs, h->ev_side, 0));        // join before the graph ends
        }, &h->g_multi_graph[gi], &h->g_multi_prep[gi]);
        h->g_eval = capture(h, [&](hipStream_t cs) { run_launches(h, h->fwd, cs); run_launches(h, std::vector<Launch>{h->cost_only}, cs); });
        h->timing = tsave;
    }
}

void comm_check_error(avae_handle* h);

void fetch_cost(avae_handle* h, float* cost_host, bool from_state, hipStream_t s) {
    if (!cost_host) return;
    const void* src = from_state ? (const void*)&h->state()->last_cost : (const void*)(h->grad() + h->P_int);
    HIP_OK(hipMemcpyAsync(cost_host, src, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    comm_check_error(h);
}

void copy_rows(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, hipStream_t s) {
    HIP_OK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToDevice, s));
}

// start of every stream-taking call (under the handle's mutex)
hipStream_t on_stream(avae_handle* h, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (h->has_last_stream && s != h->last_stream) {
        if (!h->ev_switch) HIP_OK(hipEventCreateWithFlags(&h->ev_switch, hipEventDisableTiming));
        if (hipEventRecord(h->ev_switch, h->last_stream) == hipSuccess) HIP_OK(hipStreamWaitEvent(s, h->ev_switch, 0));
        else (void)hipGetLastError();          // the previous stream no longer exists: its work is done
    }
    h->last_stream = s; h->has_last_stream = true;
    return s;
}

template <typename F> int guarded(avae_handle* h, F&& f) {
    if (!h) return 1;
    std::lock_guard<std::mutex> lk(h->mu);
    try {
        DeviceGuard dg(h->cfg.device);
        f();
        return 0;
    } catch (const std::exception& e) {
        h->err = e.what();
        return 2;
    }
}

void master_to_host(avae_handle* h, size_t off, std::vector<float>& host) {
    host.resize(h->P_int);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(host.data(), h->at<void>(off), h->P_int * 4, hipMemcpyDeviceToHost));
}
void host_to_master(avae_handle* h, size_t off, const std::vector<float>& host) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(h->at<void>(off), host.data(), h->P_int * 4, hipMemcpyHostToDevice));
}

// ----------------------------------------------------------------------------- RCCL (loaded at run time)
// The process usually has an RCCL already (PyTorch-ROCm bundles one): that copy is used when present, the system's otherwise.
// Only the five entry points the gradient all-reduce needs are bound; no RCCL header is needed at build time.
struct Rccl {
    struct Id { char b[128]; };                 // ncclUniqueId (passed by value)
    typedef int (*get_id_t)(void*);
    typedef int (*init_rank_t)(void**, int, Id, int);
    typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef int (*destroy_t)(void*);
    typedef const char* (*err_t)(int);
    typedef int (*group_t)();
    void* lib = nullptr;
    get_id_t get_id = nullptr; init_rank_t init_rank = nullptr; all_reduce_t all_reduce = nullptr; destroy_t destroy = nullptr;
    err_t err_str = nullptr; group_t group_start = nullptr, group_end = nullptr;
    static Rccl& get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] {
            const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
            for (const char* n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;       // the copy already in the process
            for (const char* n : names) { if (r.lib) break; r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
            if (!r.lib) return;
            r.get_id = (get_id_t)dlsym(r.lib, "ncclGetUniqueId"); r.init_rank = (init_rank_t)dlsym(r.lib, "ncclCommInitRank");
            r.all_reduce = (all_reduce_t)dlsym(r.lib, "ncclAllReduce"); r.destroy = (destroy_t)dlsym(r.lib, "ncclCommDestroy");
            r.err_str = (err_t)dlsym(r.lib, "ncclGetErrorString");
            r.group_start = (group_t)dlsym(r.lib, "ncclGroupStart"); r.group_end = (group_t)dlsym(r.lib, "ncclGroupEnd");
        });
        if (!r.lib || !r.get_id || !r.init_rank || !r.all_reduce || !r.destroy || !r.group_start || !r.group_end)
            throw Err("RCCL is not available (librccl.so could not be loaded): the library-owned gradient all-reduce needs it");
        return r;
    }
};
#define NCCL_OK(expr)                                                                            \
    do {                                                                                         \
        int r_ = (expr);                                                                         \
        if (r_ != 0) throw Err(std::string(#expr) + ": " + (Rccl::get().err_str ? Rccl::get().err_str(r_) : "RCCL error") + " (" + std::to_string(r_) + ")"); \
    } while (0)
constexpr int kNcclFloat32 = 7, kNcclSum = 0;        // rccl.h: ncclFloat32, ncclSum (ncclBfloat16 = 9)

constexpr int kNcclBfloat16 = 9;
constexpr uint32_t kIpcMagic = 0x41564950u;      // "AVIP"
struct IpcHandleBytes {                          // what avae_comm_ipc_handle hands out (AVAE_IPC_HANDLE_BYTES)
    hipIpcMemHandle_t mem;                       // 64 bytes
    uint64_t bytes, raw_ptr, layout;             // block size; the exporter's own pointer (same-process attach); layout check word
    int32_t rank, world, pid;
    uint32_t magic;
};
static_assert(sizeof(IpcHandleBytes) <= AVAE_IPC_HANDLE_BYTES, "IPC handle bytes");
static_assert(kMaxWorld == AVAE_MAX_WORLD, "world size limit");

void comm_streams(avae_handle* h) {
    int lo = 0, hi = 0;                          // the collective's kernels should start the moment their gradients exist
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (std::getenv("AVAE_COMM_NO_PRIO")) hi = 0;     // (A/B)
    HIP_OK(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi));
    for (int b = 0; b < 2; ++b) {
        HIP_OK(hipEventCreateWithFlags(&h->ev_grad[b], hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&h->ev_red[b], hipEventDisableTiming));
    }
    h->comm_world = h->cfg.world_size; h->comm_rank = h->cfg.rank;
    h->comm_on = true;
}

void comm_init_rccl(avae_handle* h) {
    Rccl& R = Rccl::get();
    Rccl::Id id;
    std::memcpy(id.b, h->cfg.nccl_id, 128);
    {   // RCCL prints a version banner on stdout when its first communicator comes up; the host's stdout may be a protocol
        // (bench.py: one JSON line): send it to stderr
        std::fflush(stdout);
        const int saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
        const int rc = R.init_rank(&h->comm, h->cfg.world_size, id, h->cfg.rank);
        std::fflush(stdout);
        if (saved >= 0) { (void)dup2(saved, 1); (void)close(saved); }
        NCCL_OK(rc);
    }
    comm_streams(h);
}

// Exchange block of the one-shot all-reduce: [flag1 | flag2 | costs | seq | err | slots (one per source rank) | result area]
void comm_init_ipc(avae_handle* h) {
    const int N = h->cfg.world_size;
    IpcArgs& a = h->ipc_args;
    std::memset(&a, 0, sizeof(a));
    int blocks = 256;                             // one workgroup per CU: the exchange is three memory passes over the range, and 64
                                                  // workgroups moved them at < 1 TB/s (14 us for C2's 6 MB with one rank, measured)
    if (const char* e = std::getenv("AVAE_IPC_BLOCKS")) blocks = std::atoi(e);
    if (blocks < 1 || blocks > 1024) throw Err("AVAE_IPC_BLOCKS out of range");
    const bool bf = h->cfg.wire_dtype == AVAE_BF16;
    const size_t gb = bf ? 16 : 32, G = h->P_int / 8, S = (G + N - 1) / N;        // (P_int is a multiple of 32 floats)
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = rup(off + bytes, 256); return o; };
    a.off_flag1 = (long long)take((size_t)blocks * kMaxWorld * 4);
    a.off_flag2 = (long long)take((size_t)blocks * kMaxWorld * 4);
    a.off_cost = (long long)take(kMaxWorld * 4);
    a.off_seq = (long long)take((size_t)blocks * 4);
    a.off_err = (long long)take(4);
    a.slot_stride = (long long)rup(S * gb, 256);
    a.off_slots = (long long)take((size_t)N * a.slot_stride);
    a.off_res = (long long)take(G * gb);
    h->ipc_bytes = off;
    a.g = h->grad(); a.world = N; a.rank = h->cfg.rank; a.wire_bf16 = bf ? 1 : 0; a.blocks = blocks;
    double ms = 20000.0;                          // a peer that is this late has failed: raise the error word instead of hanging
    if (const char* e = std::getenv("AVAE_IPC_TIMEOUT_MS")) ms = std::atof(e);
    a.timeout_ticks = (unsigned long long)(ms * 1e5);                             // 100 MHz wall clock
    // uncached: neither this device's L2s nor a peer's keep lines of it (RCCL allocates its own buffers the same way)
    HIP_OK(hipExtMallocWithFlags(reinterpret_cast<void**>(&h->ipc_block), h->ipc_bytes, hipDeviceMallocUncached));
    HIP_OK(hipMemset(h->ipc_block, 0, h->ipc_bytes));
    HIP_OK(hipDeviceSynchronize());
    a.peer[h->cfg.rank] = h->ipc_block; h->ipc_peer[h->cfg.rank] = h->ipc_block;
    comm_streams(h);
    h->ipc_attached = N == 1;
}

void comm_init(avae_handle* h) {
    if (h->cfg.world_size < 1 || h->cfg.world_size > kMaxWorld || h->cfg.rank < 0 || h->cfg.rank >= h->cfg.world_size)
        throw Err("use_comm: world_size must be in [1," + std::to_string(kMaxWorld) + "] and rank in [0, world_size)");
    if (h->cfg.use_comm == AVAE_COMM_RCCL) comm_init_rccl(h);
    else if (h->cfg.use_comm == AVAE_COMM_IPC) comm_init_ipc(h);
    else throw Err("use_comm must be AVAE_COMM_NONE, AVAE_COMM_RCCL or AVAE_COMM_IPC");
}

uint64_t ipc_layout_word(const avae_handle* h) {
    return (uint64_t)h->P_int * 1315423911ull ^ ((uint64_t)h->ipc_args.blocks << 48) ^ ((uint64_t)h->ipc_args.wire_bf16 << 40) ^ (uint64_t)h->ipc_bytes;
}

void comm_check_error(avae_handle* h) {       // after a synchronise: did a bounded wait of the IPC all-reduce give up?
    if (!h->ipc_block) return;
    unsigned e = 0;
    HIP_OK(hipMemcpy(&e, h->ipc_block + h->ipc_args.off_err, 4, hipMemcpyDeviceToHost));
    if (e) {
        std::string who;
        for (int r = 0; r < kMaxWorld; ++r) if (e & (1u << (8 + r))) who += " " + std::to_string(r);
        throw Err("IPC all-reduce: timed out waiting for rank(s)" + who + " (phase mask " + std::to_string(e & 0xFFu) +
                  "): a peer stopped, or the ranks did not make the same sequence of training calls");
    }
}

// ranges of bucket b, from the memory plan alone (so that avae_dp_plan needs no device): ONE contiguous range per bucket
void dp_ranges(const avae_handle* h, int* n_buckets, std::vector<avae_handle::Range> (&out)[2]) {
    bool any_conv = false;
    for (const Mod& md : h->mods) any_conv = any_conv || md.conv;
    out[0].clear(); out[1].clear();
    // comm_buckets = 1: north_star's literal design, a single all-reduce of the whole buffer (gradient + cost)
    if (any_conv || h->cfg.comm_buckets == 1) { *n_buckets = 1; out[0].push_back({0, h->P_int + 1}); return; }
    *n_buckets = 2;
    out[1].push_back({0, h->P_enc});                               // encoder sides of every modality
    out[0].push_back({h->P_enc, h->P_int + 1 - h->P_enc});         // decoder sides + the cost slot (element P_int)
}

// forward + the backward part of bucket b + its weight gradients, on staging set j
void dp_segment(avae_handle* h, int j, int bucket, hipStream_t s, bool direct = false) {
    if (bucket < 0 || bucket >= h->n_buckets) throw Err("data-parallel bucket out of range");
    if (j < 0 || j >= kMultiSteps) throw Err("staging set out of range");
    auto body = [&](hipStream_t cs) {
        auto run = [&](const std::vector<Launch>& ls, size_t lo, size_t hi) {
            std::vector<Launch> moved;
            for (size_t i = lo; i < hi && i < ls.size(); ++i) moved.push_back(relocated(h, ls[i], j));
            run_launches(h, moved, cs);
        };
        if (h->n_buckets == 1) { run(h->fwd, 0, h->fwd.size()); run(h->bwd, 0, h->bwd.size()); run(h->wgrad, 0, h->wgrad.size()); return; }
        if (bucket == 0) { run(h->fwd, 0, h->fwd.size()); run(h->bwd, 0, (size_t)h->bwd_split); run(h->wgrad_b[0], 0, h->wgrad_b[0].size()); }
        else { run(h->bwd, (size_t)h->bwd_split, h->bwd.size()); run(h->wgrad_b[1], 0, h->wgrad_b[1].size()); }
    };
    if (h->cfg.use_graph && !h->timing && !direct) {
        if (h->g_dp[bucket].empty()) h->g_dp[bucket].assign(kMultiSteps, nullptr);
        if (!h->g_dp[bucket][j]) h->g_dp[bucket][j] = capture(h, body);
        HIP_OK(hipGraphLaunch(h->g_dp[bucket][j], s));
    } else {
        body(s);
    }
}

// SUM-all-reduce of bucket b's range of the gradient buffer on the comm stream (either backend)
void dp_allreduce(avae_handle* h, int b, hipStream_t cs) {
    const avae_handle::Range& r = h->ranges_b[b][0];
    float* g = h->grad();
    const bool has_cost = r.off + r.count == h->P_int + 1;
    const size_t n = has_cost ? r.count - 1 : r.count;               // the gradient part: a multiple of 32 floats
    Timed t(h, cs, h->n_buckets == 1 ? "allreduce" : b == 0 ? "allreduce_dec" : "allreduce_enc");
    t_launch_events = LaunchEvents{nullptr, nullptr};               // (several enqueues: bracket them with plain records instead)
    if (t.on) (void)hipEventRecord(t.a, cs);
    if (h->cfg.use_comm == AVAE_COMM_IPC) {
        if (!h->ipc_attached) throw Err("AVAE_COMM_IPC: avae_comm_ipc_attach has not been called on this replica");
        IpcArgs a = h->ipc_args;
        a.off = (long long)r.off; a.granules = (long long)(n / 8); a.cost_idx = has_cost ? (long long)h->P_int : -1;
        launch_ipc_allreduce(a, cs);
        LAUNCH_OK("ipc_allreduce");
    } else {
        Rccl& R = Rccl::get();
        if (h->cfg.wire_dtype == AVAE_BF16) {                      // bf16 on the wire, the cost beside it in fp32
            unsigned short* wire = h->at<unsigned short>(h->off_wire) + r.off;
            launch_wire_pack(g + r.off, wire, (long long)n, cs); LAUNCH_OK("wire_pack");
            NCCL_OK(R.group_start());
            NCCL_OK(R.all_reduce(wire, wire, n, kNcclBfloat16, kNcclSum, h->comm, cs));
            if (has_cost) NCCL_OK(R.all_reduce(g + h->P_int, g + h->P_int, 1, kNcclFloat32, kNcclSum, h->comm, cs));
            NCCL_OK(R.group_end());
            launch_wire_unpack(g + r.off, wire, (long long)n, cs); LAUNCH_OK("wire_unpack");
        } else {
            NCCL_OK(R.all_reduce(g + r.off, g + r.off, r.count, kNcclFloat32, kNcclSum, h->comm, cs));
        }
    }
    if (t.on) (void)hipEventRecord(t.b, cs);
}

// one step of the library-owned data-parallel pipeline on staging set j (main stream s, collective on the comm stream)
void dp_step(avae_handle* h, int j, hipStream_t s, bool direct = false) {
    if (h->n_buckets == 1) {
        // one bucket: nothing to overlap, so everything stays on ONE stream -- a kernel on a second stream inside a captured graph
        // costs fork / join edges (measured, one rank: 0.0583 -> see DESIGN section 6; the two-bucket pipeline pays 35 us for them)
        dp_segment(h, j, 0, s, direct);
        dp_allreduce(h, 0, s);
        run_adam(h, 0, s, -1);
        return;
    }
    for (int b = 0; b < h->n_buckets; ++b) {
        dp_segment(h, j, b, s, direct);
        HIP_OK(hipEventRecord(h->ev_grad[b], s));
        HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_grad[b], 0));
        dp_allreduce(h, b, h->comm_stream);
        HIP_OK(hipEventRecord(h->ev_red[b], h->comm_stream));
    }
    for (int b = 0; b < h->n_buckets; ++b) {
        HIP_OK(hipStreamWaitEvent(s, h->ev_red[b], 0));
        run_adam(h, 0, s, h->n_buckets == 1 ? -1 : b);
    }
}

// ----------------------------------------------------------------------------- serving
// generate() of every (MLP) modality as ONE graph replay per call (the reference's CEM / GUI callers decode
// 10-50 times per iteration, baxter_vae_assoc_writer.py:141-147,259-304, vae_assoc_model_viewer.py:107-113).
avae_handle::Serve& serve_plan(avae_handle* h, int bucket) {
    for (avae_handle::Serve& sv : h->serve) if (sv.bucket == bucket) return sv;
    h->serve.emplace_back();
    avae_handle::Serve& sv = h->serve.back();
    sv.bucket = bucket;
    Builder bd(h, sv.items, bucket, false);
    int slot = 0, Lmax = 0;
    for (const Mod& md : h->mods) Lmax = std::max(Lmax, md.L);
    auto group = [&](const std::string& name, auto&& fill) {
        const int first = (int)sv.items.size();
        fill();
        const int count = (int)sv.items.size() - first;
        if (count > 0) sv.launches.push_back(finish_launch(h, sv.items, first, count, name, &slot));
    };
    // The decoder's first layer (K = n_z + 1) rides in the per-call staging launch when that fits its tail product (K_SERVE_Z: one K
    // tile of bf16, two of fp32, at most 1024 units, 32-row tiles); otherwise it is the graph's first launch behind k_serve.
    {
        const int first = (int)sv.items.size();
        bool fits = !std::getenv("AVAE_NO_TAIL") && bucket <= 4096;
        for (Mod& md : h->mods) {
            const WorkItem c = bd.fwd_hidden(md.Z, md.dec[0], md.D[0]);
            fits = fits && c.K <= (h->es == 2 ? 1 : 2) * h->KU && c.N <= 1024 && h->nz + 1 <= 64;
            WorkItem w = gemm_item(K_SERVE_Z, bucket, c.N, 0, nullptr, 0, nullptr, 0);
            w.nz = h->nz; w.partial = reinterpret_cast<float*>(h->at<ServeSlot>(h->off_slot));
            w.tail_mode = 1; w.tail_w = c.B; w.tail_ldw = c.ldb; w.tail_n = c.N; w.tail_out = c.out0; w.tail_ldo = c.ld0; w.tail_act = c.act;
            w.tail_kt = c.K / h->KU;
            sv.items.push_back(w);
        }
        if (fits) {
            sv.in_launch = finish_launch(h, sv.items, first, (int)sv.items.size() - first, "serve_in+serve_dec1", &slot);
            if (sv.in_launch.cfg != 3) fits = false;
        }
        if (fits && !std::getenv("AVAE_NO_LEAN")) {            // the same launch as a kernel of its own with a small argument block (k_serve_in)
            ServeInArgs& ia = sv.in_lean;
            std::memset(&ia, 0, sizeof(ia));
            ia.slot = h->at<ServeSlot>(h->off_slot); ia.n_mod = h->M; ia.nz = h->nz; ia.bucket = bucket; ia.tiles_m = (bucket + 31) / 32;
            int max_slices = 1;
            for (int m = 0; m < h->M; ++m) {
                const WorkItem& w = sv.items[first + m];
                ServeInMod& md = ia.mod[m];
                md.w = w.tail_w; md.out = w.tail_out; md.n = w.tail_n; md.ldw = w.tail_ldw; md.ldo = w.tail_ldo; md.kt = w.tail_kt; md.act = w.tail_act;
                md.slices = (w.tail_n + 63) / 64;
                max_slices = std::max(max_slices, md.slices);
            }
            sv.in_lean_grid = ia.tiles_m * max_slices;
            sv.lean_in = true;
        }
        if (!fits) sv.items.resize(first);
        sv.fused_in = fits;
    }
    for (int k = sv.fused_in ? 1 : 0; k < Lmax; ++k)
        group("serve_dec" + std::to_string(k + 1), [&] { for (Mod& md : h->mods) if (k < md.L) sv.items.push_back(bd.fwd_hidden(k == 0 ? md.Z : md.D[k - 1], md.dec[k], md.D[k])); });
    group("serve_out", [&] {
        for (int m = 0; m < h->M; ++m) {
            WorkItem w = bd.fwd_out(h->mods[m], m, false);
            w.aux2 = h->at<void>(h->off_slot); w.n_mod = m;       // K_FWD_OUT_STORE: rows and destination come from the slot
            sv.items.push_back(w);
        }
    });
    ServeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.slot = h->at<ServeSlot>(h->off_slot); a.n_mod = h->M; a.nz = h->nz; a.bucket = bucket;
    for (int m = 0; m < h->M; ++m) {
        const Mod& md = h->mods[m];
        a.Z[m] = h->at<void>(md.Z.rm); a.ldz[m] = md.Z.ld;
    }
    a.blocks_per_mod = std::max(1, std::min(8, (bucket * h->nz + kThreads - 1) / kThreads));
    sv.in = a;                                                   // the per-call staging launch (ahead of the graph)
    auto body = [&](hipStream_t cs) { run_launches(h, sv.launches, cs); };
    const bool tsave = h->timing;
    h->timing = false;
    sv.graph = capture(h, body);
    h->timing = tsave;
    return sv;
}

}  // namespace

// ============================================================================= C ABI
extern "C" {

int avae_workspace_bytes(const avae_config* cfg, size_t* bytes) {
    try {
        if (!cfg || !bytes) throw Err("null argument");
        check_config(*cfg);
        avae_handle tmp;
        tmp.cfg = *cfg;
        plan_memory(&tmp);
        *bytes = tmp.ws_bytes;
        return 0;
    } catch (const std::exception& e) { g_create_error = e.what(); return 2; }
}

int avae_create(const avae_config* cfg, avae_handle** out) {
    avae_handle* h = nullptr;
    try {
        if (!cfg || !out) throw Err("null argument");
        check_config(*cfg);
        h = new avae_handle();
        h->cfg = *cfg;
        if (const char* e = std::getenv("AVAE_DEBUG_SYNC")) h->debug_sync = e[0] == '1';
        if (h->debug_sync) h->cfg.use_graph = 0;
        if (h->cfg.beta1 == 0.f && h->cfg.beta2 == 0.f && h->cfg.adam_eps == 0.f) { h->cfg.beta1 = 0.9f; h->cfg.beta2 = 0.999f; h->cfg.adam_eps = 1e-8f; }
        int ndev = 0;
        HIP_OK(hipGetDeviceCount(&ndev));
        if (ndev < 1) throw Err("no HIP device: libavae has no CPU fallback");
        if (h->cfg.device < 0 || h->cfg.device >= ndev) throw Err("device ordinal out of range");
        DeviceGuard dg(h->cfg.device);
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, h->cfg.device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw Err(std::string("libavae is built for gfx950 only; device is ") + prop.gcnArchName);
        plan_memory(h);
        if (cfg->workspace) {
            if (cfg->workspace_bytes < h->ws_bytes) throw Err("workspace too small");
            if (reinterpret_cast<uintptr_t>(cfg->workspace) % 256) throw Err("workspace must be 256-byte aligned");
            h->ws = reinterpret_cast<unsigned char*>(cfg->workspace);
        } else {
            HIP_OK(hipMalloc(reinterpret_cast<void**>(&h->ws), h->ws_bytes));
            h->own_ws = true;
        }
        HIP_OK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
        init_device(h);
        if (h->cfg.use_comm) comm_init(h);
        *out = h;
        return 0;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        if (h) {
            if (h->comm) { try { (void)Rccl::get().destroy(h->comm); } catch (...) {} }
            if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
            if (h->ipc_block) (void)hipFree(h->ipc_block);
            if (h->own_ws && h->ws) (void)hipFree(h->ws);
            if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
            delete h;
        }
        return 2;
    }
}

void avae_destroy(avae_handle* h) {
    if (!h) return;
    int prev_dev = -1;
    if (hipGetDevice(&prev_dev) != hipSuccess) prev_dev = -1;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (hipGraphExec_t g : {h->g_full, h->g_multi[0], h->g_multi[1], h->g_eval}) if (g) (void)hipGraphExecDestroy(g);
    for (int b = 0; b < 2; ++b) for (hipGraphExec_t g : h->g_dp[b]) if (g) (void)hipGraphExecDestroy(g);
    for (avae_handle::Serve& sv : h->serve) { if (sv.graph) (void)hipGraphExecDestroy(sv.graph); for (hipGraphExec_t g : sv.ring_graph) if (g) (void)hipGraphExecDestroy(g); }
    if (h->serve_ring) (void)hipHostFree(h->serve_ring);
    if (h->ev_switch) (void)hipEventDestroy(h->ev_switch);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    for (int gi = 0; gi < 2; ++gi) { if (h->g_dpm[gi]) (void)hipGraphExecDestroy(h->g_dpm[gi]); if (h->g_dpm_graph[gi]) (void)hipGraphDestroy(h->g_dpm_graph[gi]); }
    if (h->comm) { try { (void)Rccl::get().destroy(h->comm); } catch (...) {} }
    for (int r = 0; r < kMaxWorld; ++r) if (h->ipc_opened[r] && h->ipc_peer[r]) (void)hipIpcCloseMemHandle(h->ipc_peer[r]);
    if (h->ipc_block) (void)hipFree(h->ipc_block);
    for (int b = 0; b < 2; ++b) { if (h->ev_grad[b]) (void)hipEventDestroy(h->ev_grad[b]); if (h->ev_red[b]) (void)hipEventDestroy(h->ev_red[b]); }
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    for (hipGraph_t g : {h->g_full_graph, h->g_multi_graph[0], h->g_multi_graph[1]}) if (g) (void)hipGraphDestroy(g);
    for (TimingRec& r : h->trecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->own_ws && h->ws) (void)hipFree(h->ws);
    if (prev_dev >= 0 && prev_dev != h->cfg.device) (void)hipSetDevice(prev_dev);
    delete h;
}

const char* avae_last_error(const avae_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int avae_param_count(const avae_handle* h, size_t* n) {
    if (!h || !n) return 1;
    *n = h->P_flat;
    return 0;
}

int avae_get_params(avae_handle* h, float* host_dst) {
    return guarded(h, [&] { std::vector<float> I; master_to_host(h, h->off_theta, I); convert_params<false>(h, host_dst, I.data()); });
}

int avae_set_params(avae_handle* h, const float* host_src) {
    return guarded(h, [&] {
        std::vector<float> I(h->P_int, 0.0f);
        convert_params<true>(h, const_cast<float*>(host_src), I.data());
        host_to_master(h, h->off_theta, I);
        run_adam(h, 1, nullptr);          // rebuild the compute-dtype shadows W / W^T
        HIP_OK(hipDeviceSynchronize());
    });
}

int avae_get_grads(avae_handle* h, float* host_dst) {
    return guarded(h, [&] { std::vector<float> I; master_to_host(h, h->off_g, I); convert_params<false>(h, host_dst, I.data()); });
}

int avae_get_opt_state(avae_handle* h, float* host_m, float* host_v, int64_t* step) {
    return guarded(h, [&] {
        std::vector<float> I;
        if (host_m) { master_to_host(h, h->off_m, I); convert_params<false>(h, host_m, I.data()); }
        if (host_v) { master_to_host(h, h->off_v, I); convert_params<false>(h, host_v, I.data()); }
        if (step) { long long st = 0; HIP_OK(hipDeviceSynchronize()); HIP_OK(hipMemcpy(&st, &h->state()->step, sizeof(st), hipMemcpyDeviceToHost)); *step = st; }
    });
}

int avae_set_opt_state(avae_handle* h, const float* host_m, const float* host_v, int64_t step) {
    return guarded(h, [&] {
        std::vector<float> I(h->P_int, 0.0f);
        if (host_m) { convert_params<true>(h, const_cast<float*>(host_m), I.data()); host_to_master(h, h->off_m, I); }
        if (host_v) { std::fill(I.begin(), I.end(), 0.0f); convert_params<true>(h, const_cast<float*>(host_v), I.data()); host_to_master(h, h->off_v, I); }
        long long st = step;
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(&h->state()->step, &st, sizeof(st), hipMemcpyHostToDevice));
    });
}

// Data-parallel runs of consecutive batches: one staging launch for up to kMultiSteps batches, then per step
// backward on its staging set -> (caller's all-reduce) -> apply.
int avae_stage_batches(avae_handle* h, int32_t n_steps, const float* const* x_dev, const int32_t* x_ld, const float* eps_dev, void* stream) {
    return guarded(h, [&] {
        if (n_steps < 1 || n_steps > kMultiSteps) throw Err("avae_stage_batches: n_steps must be in [1," + std::to_string(kMultiSteps) + "]");
        hipStream_t s = on_stream(h, stream);
        const PrepArgs a = make_prep_batch(h, x_dev, x_ld, eps_dev, h->B, 0x7261696eull, n_steps);
        Timed t(h, s, "prep");
        launch_prep(h->cfg.compute_dtype, a, s);
        LAUNCH_OK("prep");
    });
}

// one single-replica step
void train_one(avae_handle* h, const float* const* x_dev, const int32_t* x_ld, const float* eps_dev, hipStream_t s) {
    if (h->comm_on) {                   // library-owned collective: stage, then the bucketed pipeline
        const PrepArgs a = make_prep_batch(h, x_dev, x_ld, eps_dev, h->B, 0x7261696eull, 1);
        { Timed t(h, s, "prep"); launch_prep(h->cfg.compute_dtype, a, s); LAUNCH_OK("prep"); }
        dp_step(h, 0, s);
        return;
    }
    if (h->g_full && !h->timing) {      // the whole step, staging kernel included, is one graph replay
        patch_prep(h, h->g_full, h->g_full_prep, x_dev, x_ld, eps_dev);
        HIP_OK(hipGraphLaunch(h->g_full, s));
        return;
    }
    run_prep_batch(h, x_dev, x_ld, eps_dev, h->B, 0x7261696eull, s);
    run_launches(h, h->fwd, s); run_launches(h, h->bwd, s);
    if (!h->wgrad_adam.empty()) run_launches(h, h->wgrad_adam, s);
    else { run_launches(h, h->wgrad, s); run_adam(h, 0, s); }
    if (h->timing) {      // floor of the measurement: a one-store kernel (partial slot 0 is rewritten every step anyway)
        Timed t(h, s, "_null_kernel");
        launch_fill(h->at<void>(h->off_partial), 4, 0u, 0, 1, 1, s);
        LAUNCH_OK("_null_kernel");
    }
}

int avae_train_step(avae_handle* h, const float* const* x_dev, const int32_t* x_ld, const float* eps_dev, float* cost_host, void* stream) {
    return guarded(h, [&] {
        hipStream_t s = on_stream(h, stream);
        train_one(h, x_dev, x_ld, eps_dev, s);
        fetch_cost(h, cost_host, true, s);
    });
}

int avae_train_steps(avae_handle* h, int32_t n_steps, const float* const* x_dev, const int32_t* x_ld, const float* eps_dev,
                     float* cost_host, void* stream) {
    return guarded(h, [&] {
        if (n_steps < 1) throw Err("avae_train_steps: n_steps must be >= 1");
        hipStream_t s = on_stream(h, stream);
        std::vector<const float*> x(h->M);
        auto batch = [&](int i) {       // rows [i*B, (i+1)*B) of every modality and of eps
            for (int m = 0; m < h->M; ++m) {
                const size_t ld = (x_ld && x_ld[m] > 0) ? (size_t)x_ld[m] : (size_t)h->mods[m].n_in;
                x[m] = x_dev[m] + (size_t)i * h->B * ld;
            }
            return eps_dev ? eps_dev + (size_t)i * h->B * h->nz : nullptr;
        };
        int i = 0;
        if (h->comm_on) {               // data parallel: batches staged kMultiSteps at a time, then backward -> all-reduce -> Adam per bucket and step
            // Runs of 16 (then 4) steps as ONE captured graph -- staging kernel, segments, ncclAllReduce on the comm stream (it
            // joins the capture through the events), Adam per bucket -- so that the host is out of the loop exactly as in the
            // single-replica path.  If RCCL refuses capture on this stack the host steps the same pipeline.
            for (int gi = 0; gi < 2 && h->cfg.use_graph && !h->timing && !h->dp_graph_failed; ++gi)
                for (; i + kMultiSizes[gi] <= n_steps; i += kMultiSizes[gi]) {
                    if (!h->g_dpm[gi]) {
                        std::vector<const float*> x0(h->M, h->at<float>(h->mods[0].X32));
                        try {
                            h->g_dpm[gi] = capture_with_prep(h, [&](hipStream_t cs) {
                                const PrepArgs a = make_prep_batch(h, x0.data(), nullptr, nullptr, h->B, 0x7261696eull, kMultiSizes[gi]);
                                launch_prep(h->cfg.compute_dtype, a, cs);
                                LAUNCH_OK("prep");
                                for (int j = 0; j < kMultiSizes[gi]; ++j) dp_step(h, j, cs, true);
                            }, &h->g_dpm_graph[gi], &h->g_dpm_prep[gi]);
                        } catch (const std::exception& e) {
                            h->dp_graph_failed = true;
                            h->g_dpm[gi] = nullptr;
                            (void)hipGetLastError();
                            if (std::getenv("AVAE_DEBUG_DP")) std::fprintf(stderr, "[avae] data-parallel graph capture failed (%s): host-stepped pipeline\n", e.what());
                            break;
                        }
                    }
                    const float* e = batch(i);
                    patch_prep(h, h->g_dpm[gi], h->g_dpm_prep[gi], x.data(), x_ld, e, kMultiSizes[gi]);
                    HIP_OK(hipGraphLaunch(h->g_dpm[gi], s));
                }
            for (; i < n_steps; i += kMultiSteps) {
                const int n = std::min(kMultiSteps, n_steps - i);
                const float* e = batch(i);
                const PrepArgs a = make_prep_batch(h, x.data(), x_ld, e, h->B, 0x7261696eull, n);
                { Timed t(h, s, "prep"); launch_prep(h->cfg.compute_dtype, a, s); LAUNCH_OK("prep"); }
                for (int j = 0; j < n; ++j) dp_step(h, j, s);
            }
            fetch_cost(h, cost_host, true, s);
            return;
        }
        for (int gi = 0; gi < 2 && !h->timing; ++gi)
            for (; h->g_multi[gi] && i + kMultiSizes[gi] <= n_steps; i += kMultiSizes[gi]) {
                const float* e = batch(i);
                patch_prep(h, h->g_multi[gi], h->g_multi_prep[gi], x.data(), x_ld, e, kMultiSizes[gi]);
                HIP_OK(hipGraphLaunch(h->g_multi[gi], s));
            }
        for (; i < n_steps; ++i) {
            const float* e = batch(i);
            train_one(h, x.data(), x_ld, e, s);
        }
        fetch_cost(h, cost_host, true, s);
    });
}

int avae_grad_buffer(avae_handle* h, float** dev_ptr, size_t* n_floats) {
    if (!h || !dev_ptr || !n_floats) return 1;
    *dev_ptr = h->grad();
    *n_floats = h->P_int + 1;     // gradient (internal padded layout) + the cost slot
    return 0;
}

int avae_dp_plan(const avae_config* cfg, int32_t* n_buckets, int32_t* n_ranges, int64_t* offs, int64_t* counts) {
    try {
        if (!cfg || !n_buckets || !n_ranges || !offs || !counts) throw Err("null argument");
        check_config(*cfg);
        avae_handle tmp;
        tmp.cfg = *cfg;
        plan_memory(&tmp);
        std::vector<avae_handle::Range> r[2];
        int nb = 0;
        dp_ranges(&tmp, &nb, r);
        *n_buckets = nb;
        int k = 0;
        for (int b = 0; b < 2; ++b) {
            n_ranges[b] = (int32_t)r[b].size();
            for (const avae_handle::Range& x : r[b]) { offs[k] = (int64_t)x.off; counts[k] = (int64_t)x.count; ++k; }
        }
        return 0;
    } catch (const std::exception& e) { g_create_error = e.what(); return 2; }
}

int avae_dp_backward(avae_handle* h, int32_t j, int32_t bucket, void* stream) {
    return guarded(h, [&] { dp_segment(h, j, bucket, on_stream(h, stream)); });
}

int avae_dp_apply(avae_handle* h, int32_t bucket, float* cost_host, void* stream) {
    return guarded(h, [&] {
        if (bucket < 0 || bucket >= h->n_buckets) throw Err("data-parallel bucket out of range");
        hipStream_t s = on_stream(h, stream);
        run_adam(h, 0, s, h->n_buckets == 1 ? -1 : bucket);
        fetch_cost(h, cost_host, true, s);
    });
}

int avae_comm_unique_id(void* id128) {
    try {
        if (!id128) throw Err("null argument");
        NCCL_OK(Rccl::get().get_id(id128));
        return 0;
    } catch (const std::exception& e) { g_create_error = e.what(); return 2; }
}

int avae_comm_ipc_handle(avae_handle* h, void* handle_out) {
    return guarded(h, [&] {
        if (!handle_out) throw Err("null argument");
        if (!h->ipc_block) throw Err("avae_comm_ipc_handle: the replica was not created with use_comm = AVAE_COMM_IPC");
        IpcHandleBytes hb;
        std::memset(&hb, 0, sizeof(hb));
        HIP_OK(hipIpcGetMemHandle(&hb.mem, h->ipc_block));
        hb.bytes = h->ipc_bytes; hb.raw_ptr = reinterpret_cast<uint64_t>(h->ipc_block); hb.layout = ipc_layout_word(h);
        hb.rank = h->cfg.rank; hb.world = h->cfg.world_size; hb.pid = (int32_t)getpid(); hb.magic = kIpcMagic;
        std::memset(handle_out, 0, AVAE_IPC_HANDLE_BYTES);
        std::memcpy(handle_out, &hb, sizeof(hb));
    });
}

int avae_comm_ipc_attach(avae_handle* h, const void* handles_by_rank) {
    return guarded(h, [&] {
        if (!handles_by_rank) throw Err("null argument");
        if (!h->ipc_block) throw Err("avae_comm_ipc_attach: the replica was not created with use_comm = AVAE_COMM_IPC");
        if (h->ipc_attached && h->cfg.world_size > 1) throw Err("avae_comm_ipc_attach: already attached");
        const unsigned char* raw = reinterpret_cast<const unsigned char*>(handles_by_rank);
        for (int r = 0; r < h->cfg.world_size; ++r) {
            IpcHandleBytes hb;
            std::memcpy(&hb, raw + (size_t)r * AVAE_IPC_HANDLE_BYTES, sizeof(hb));
            if (hb.magic != kIpcMagic || hb.rank != r || hb.world != h->cfg.world_size)
                throw Err("avae_comm_ipc_attach: entry " + std::to_string(r) + " is not rank " + std::to_string(r) + "'s handle of this job");
            if (hb.bytes != h->ipc_bytes || hb.layout != ipc_layout_word(h))
                throw Err("avae_comm_ipc_attach: rank " + std::to_string(r) + " was built with a different model / wire type / AVAE_IPC_BLOCKS");
            if (r == h->cfg.rank) continue;
            void* ptr = nullptr;
            if (hb.pid == (int32_t)getpid()) ptr = reinterpret_cast<void*>(hb.raw_ptr);          // a replica of this very process: its pointer is ours
            else { HIP_OK(hipIpcOpenMemHandle(&ptr, hb.mem, hipIpcMemLazyEnablePeerAccess)); h->ipc_opened[r] = true; }
            h->ipc_peer[r] = reinterpret_cast<unsigned char*>(ptr);
            h->ipc_args.peer[r] = h->ipc_peer[r];
        }
        h->ipc_attached = true;
    });
}

int avae_cost_history(avae_handle* h, int32_t n, float* host_dst, int64_t* last_step) {
    return guarded(h, [&] {
        if (n < 0 || n > kCostHist) throw Err("cost history request out of range");
        HIP_OK(hipDeviceSynchronize());
        comm_check_error(h);
        std::vector<unsigned char> buf(sizeof(DevState));
        HIP_OK(hipMemcpy(buf.data(), h->state(), sizeof(DevState), hipMemcpyDeviceToHost));
        const DevState* st = reinterpret_cast<const DevState*>(buf.data());
        if (st->step < n) throw Err("fewer steps applied than requested");
        for (int i = 0; i < n; ++i) host_dst[i] = st->cost_hist[(st->step - n + i) % kCostHist];
        if (last_step) *last_step = st->step;
    });
}

int avae_eval_cost(avae_handle* h, const float* const* x_dev, const int32_t* x_ld, const float* eps_dev, float* cost_host, void* stream) {
    return guarded(h, [&] {
        hipStream_t s = on_stream(h, stream);
        // a fresh eps per call, as each sess.run of the reference draws one (vae_assoc.py:90,388-391): the draw counter keys it
        const unsigned long long draw = eps_dev ? 0ull : (unsigned long long)((++h->draw_id) & 0x3FFFFFu) << 34;
        run_prep_batch(h, x_dev, x_ld, eps_dev, h->B, 0x6576616cull /*eval*/ | draw, s);
        if (h->g_eval && !h->timing) HIP_OK(hipGraphLaunch(h->g_eval, s));
        else { run_launches(h, h->fwd, s); run_launches(h, std::vector<Launch>{h->cost_only}, s); }
        fetch_cost(h, cost_host, false, s);
    });
}

static void run_inference(avae_handle* h, int m, bool enc, int rows, hipStream_t s) {
    build_inference(h, m, enc, rows);
    avae_handle::Inf& inf = enc ? h->inf_enc[m] : h->inf_dec[m];
    run_launches(h, inf.launches, s);          // items travel by value in the kernel arguments
}

int avae_encode(avae_handle* h, int32_t m, const float* x_dev, int32_t x_ld, int32_t rows, float* mu_dev, float* logvar_dev, void* stream) {
    return guarded(h, [&] {
        if (m < 0 || m >= h->M) throw Err("modality index out of range");
        if (rows < 0) throw Err("rows must be >= 0");
        hipStream_t s = on_stream(h, stream);
        const Mod& md = h->mods[m];
        const int ld = x_ld > 0 ? x_ld : md.n_in;
        const size_t nzb = (size_t)h->nz * 4;
        for (int r0 = 0; r0 < rows; r0 += h->B) {
            const int n = std::min(h->B, rows - r0);
            run_prep_single(h, x_dev + (size_t)r0 * ld, ld, n, md.n_in, md.X0, nullptr, 0, true, nullptr, 0x656e63ull, s);
            run_inference(h, m, true, n, s);
            if (mu_dev) copy_rows(mu_dev + (size_t)r0 * h->nz, nzb, h->at<float>(md.mulv), 2 * nzb, nzb, n, s);
            if (logvar_dev) copy_rows(logvar_dev + (size_t)r0 * h->nz, nzb, h->at<float>(md.mulv) + h->nz, 2 * nzb, nzb, n, s);
        }
    });
}

int avae_decode(avae_handle* h, int32_t m, const float* z_dev, int32_t rows, float* xhat_dev, void* stream) {
    return guarded(h, [&] {
        if (m < 0 || m >= h->M) throw Err("modality index out of range");
        if (rows < 0) throw Err("rows must be >= 0");
        hipStream_t s = on_stream(h, stream);
        const Mod& md = h->mods[m];
        for (int r0 = 0; r0 < rows; r0 += h->B) {
            const int n = std::min(h->B, rows - r0);
            run_prep_single(h, z_dev + (size_t)r0 * h->nz, h->nz, n, h->nz, md.Z, nullptr, 0, false, nullptr, 0, s);
            run_inference(h, m, false, n, s);
            copy_rows(xhat_dev + (size_t)r0 * md.n_in, (size_t)md.n_in * 4, h->at<float>(md.out32), (size_t)md.ld32 * 4, (size_t)md.n_in * 4, n, s);
        }
    });
}

int avae_generate(avae_handle* h, const float* z_dev, int32_t rows, float* const* xhat_dev, void* stream) {
    return guarded(h, [&] {
        if (rows < 0) throw Err("rows must be >= 0");
        if (!z_dev || !xhat_dev) throw Err("null argument");
        hipStream_t s = on_stream(h, stream);
        bool any_conv = false;
        for (const Mod& md : h->mods) any_conv = any_conv || md.conv;
        if (any_conv || !h->cfg.use_graph || h->timing) {     // conv decoders (and the diagnostic modes) go modality by modality
            for (int m = 0; m < h->M; ++m) {
                const Mod& md = h->mods[m];
                for (int r0 = 0; r0 < rows; r0 += h->B) {
                    const int n = std::min(h->B, rows - r0);
                    run_prep_single(h, z_dev + (size_t)r0 * h->nz, h->nz, n, h->nz, md.Z, nullptr, 0, false, nullptr, 0, s);
                    build_inference(h, m, false, n);
                    run_launches(h, h->inf_dec[m].launches, s);
                    copy_rows(xhat_dev[m] + (size_t)r0 * md.n_in, (size_t)md.n_in * 4, h->at<float>(md.out32), (size_t)md.ld32 * 4, (size_t)md.n_in * 4, n, s);
                }
            }
            return;
        }
        for (int r0 = 0; r0 < rows; r0 += h->B) {
            const int n = std::min(h->B, rows - r0);
            const int bucket = (n <= 64 && h->B > 64) ? 64 : h->B;       // the CEM / GUI callers' 1-64 rows, or a batch-sized chunk
            avae_handle::Serve& sv = serve_plan(h, bucket);
            ServeSlot sl;
            std::memset(&sl, 0, sizeof(sl));
            sl.z = z_dev + (size_t)r0 * h->nz; sl.rows = n;
            for (int m = 0; m < h->M; ++m) sl.out[m] = xhat_dev[m] + (size_t)r0 * h->mods[m].n_in;
            // Measured (tools/serve_latency.py, one box, 1 / 64 rows, C ABI back to back, us per call): staging launch + graph of the
            // remaining launches 19.1 / 19.9 (host side 10.2: GPU-bound -- three dependent kernels and a replay boundary of ~5 us);
            // every launch eager 14.6 / 15.6 (host-bound at 13.7 / 14.6: three launches) = the default since round 3; the pinned-host
            // ring with the staging launch as the graph's first node (no eager launch: VERDICT r2 #9) 20.6 / 21.4 -- every workgroup
            // of the first kernel reads the call record over PCIe, and the call was never host-bound.  AVAE_SERVE_GRAPH=1 /
            // AVAE_SERVE_RING=1 select the other two for A/B.
            static const bool use_ring = std::getenv("AVAE_SERVE_RING") != nullptr, eager = !use_ring && std::getenv("AVAE_SERVE_GRAPH") == nullptr;
            if (sv.fused_in && sv.lean_in && eager) {          // no graph at all: the staging launch and the plan's launches, eagerly
                sv.in_lean.call = sl;
                launch_serve_in(h->cfg.compute_dtype, sv.in_lean, sv.in_lean_grid, s); LAUNCH_OK("serve_in+serve_dec1");
                run_launches(h, sv.launches, s);
                continue;
            }
            if (sv.fused_in && sv.lean_in && use_ring) {
                // ONE graph replay per call: the record goes into the pinned ring, the slot's graph starts with the staging launch
                if (!h->serve_ring) {
                    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&h->serve_ring), kServeRing * sizeof(ServeSlot) + 64, hipHostMallocMapped | hipHostMallocPortable));
                    std::memset(h->serve_ring, 0, kServeRing * sizeof(ServeSlot) + 64);
                    h->serve_consumed = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(h->serve_ring) + kServeRing * sizeof(ServeSlot));
                    HIP_OK(hipMemsetAsync(h->at<void>(h->off_serve_count), 0, 8, s));
                }
                const unsigned long long id = h->serve_calls;
                const int slot = (int)(id % kServeRing);
                // flow control: slot id % R was last used by call id - R; it is free once the device has STARTED call id - R + 1
                // (calls run in order, so call id - R is over then)
                while (id + 2 > __atomic_load_n(h->serve_consumed, __ATOMIC_ACQUIRE) + kServeRing) sched_yield();
                h->serve_ring[slot] = sl;
                __atomic_thread_fence(__ATOMIC_RELEASE);
                if (!sv.ring_graph[slot]) {
                    void* rec_dev = nullptr;
                    void* con_dev = nullptr;
                    HIP_OK(hipHostGetDevicePointer(&rec_dev, h->serve_ring + slot, 0));
                    HIP_OK(hipHostGetDevicePointer(&con_dev, h->serve_consumed, 0));
                    ServeInArgs ia = sv.in_lean;
                    ia.rec = reinterpret_cast<const ServeSlot*>(rec_dev);
                    ia.consumed = reinterpret_cast<unsigned long long*>(con_dev);
                    ia.count = h->at<unsigned long long>(h->off_serve_count);
                    const bool tsave = h->timing;
                    h->timing = false;
                    sv.ring_graph[slot] = capture(h, [&](hipStream_t cs) {
                        launch_serve_in(h->cfg.compute_dtype, ia, sv.in_lean_grid, cs); LAUNCH_OK("serve_in+serve_dec1");
                        run_launches(h, sv.launches, cs);
                    });
                    h->timing = tsave;
                }
                HIP_OK(hipGraphLaunch(sv.ring_graph[slot], s));
                ++h->serve_calls;
                continue;
            }
            if (sv.fused_in && sv.lean_in) {
                sv.in_lean.call = sl;
                launch_serve_in(h->cfg.compute_dtype, sv.in_lean, sv.in_lean_grid, s); LAUNCH_OK("serve_in+serve_dec1");
            } else if (sv.fused_in) {    // the staging launch also runs the decoder's first layer: its items carry the call by value
                Launch& L = sv.in_launch;
                for (int m = 0; m < h->M; ++m) {
                    WorkItem& w = L.args.items[m];
                    w.aux0 = sl.z; w.n_slots = sl.rows;
                    w.out0 = sl.out[0]; w.out1 = sl.out[1]; w.out2 = sl.out[2]; w.aux1 = sl.out[3];
                }
                launch_grouped(h->cfg.compute_dtype, L.cfg, L.args, L.grid_x, L.grid_y, L.lds, h->state(), s, nullptr, 0); LAUNCH_OK("serve_in+serve_dec1");
            } else {
                launch_serve(h->cfg.compute_dtype, sv.in, sl, sv.in.blocks_per_mod * h->M, s); LAUNCH_OK("serve_in");
            }
            HIP_OK(hipGraphLaunch(sv.graph, s));
        }
    });
}

int avae_reconstruct(avae_handle* h, int32_t m, const float* x_dev, int32_t x_ld, const float* eps_dev, int32_t rows, float* xhat_dev, void* stream) {
    return guarded(h, [&] {
        if (m < 0 || m >= h->M) throw Err("modality index out of range");
        if (rows < 0) throw Err("rows must be >= 0");
        hipStream_t s = on_stream(h, stream);
        const Mod& md = h->mods[m];
        const int ld = x_ld > 0 ? x_ld : md.n_in;
        // a fresh eps per call and per modality, as each sess.run of the reference draws one (vae_assoc.py:423-424): the salt's
        // high word carries (draw counter, modality), the generator's row index is the row of the whole input (r0 + row)
        const unsigned long long draw = eps_dev ? 0ull : ((unsigned long long)((++h->draw_id) & 0x3FFFFFu) << 34) | ((unsigned long long)m << 32);
        for (int r0 = 0; r0 < rows; r0 += h->B) {
            const int n = std::min(h->B, rows - r0);
            run_prep_single(h, x_dev + (size_t)r0 * ld, ld, n, md.n_in, md.X0, nullptr, 0, true,
                            eps_dev ? eps_dev + (size_t)r0 * h->nz : nullptr, 0x7265636full | draw, s, r0);
            run_inference(h, m, true, n, s);
            run_inference(h, m, false, n, s);
            copy_rows(xhat_dev + (size_t)r0 * md.n_in, (size_t)md.n_in * 4, h->at<float>(md.out32), (size_t)md.ld32 * 4, (size_t)md.n_in * 4, n, s);
        }
    });
}

// ---- checkpoint: "AVAECKPT" | u32 version | u32 n_mod | u32 n_z | per modality {n_input, L, hs[L], conv, gener1, gener2} | u64 P | i64 step | theta | m | v
int avae_save(avae_handle* h, const char* path) {
    return guarded(h, [&] {
        std::vector<float> I, th(h->P_flat), mm(h->P_flat), vv(h->P_flat);
        master_to_host(h, h->off_theta, I); convert_params<false>(h, th.data(), I.data());
        master_to_host(h, h->off_m, I); convert_params<false>(h, mm.data(), I.data());
        master_to_host(h, h->off_v, I); convert_params<false>(h, vv.data(), I.data());
        long long step = 0;
        HIP_OK(hipMemcpy(&step, &h->state()->step, sizeof(step), hipMemcpyDeviceToHost));
        FILE* f = std::fopen(path, "wb");
        if (!f) throw Err(std::string("cannot open for writing: ") + path);
        auto w32 = [&](uint32_t v) { std::fwrite(&v, 4, 1, f); };
        std::fwrite("AVAECKPT", 1, 8, f);
        w32(2); w32((uint32_t)h->M); w32((uint32_t)h->nz);
        for (int m = 0; m < h->M; ++m) {
            const Mod& md = h->mods[m];
            w32((uint32_t)md.n_in); w32((uint32_t)md.hs.size()); for (int x : md.hs) w32((uint32_t)x);
            w32(md.conv ? 1u : 0u); w32((uint32_t)h->cfg.mod[m].conv_gener[0]); w32((uint32_t)h->cfg.mod[m].conv_gener[1]);
        }
        uint64_t P = h->P_flat; std::fwrite(&P, 8, 1, f);
        int64_t st = step; std::fwrite(&st, 8, 1, f);
        bool ok = std::fwrite(th.data(), 4, P, f) == P && std::fwrite(mm.data(), 4, P, f) == P && std::fwrite(vv.data(), 4, P, f) == P;
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) throw Err(std::string("short write: ") + path);
    });
}

int avae_load(avae_handle* h, const char* path) {
    return guarded(h, [&] {
        FILE* f = std::fopen(path, "rb");
        if (!f) throw Err(std::string("cannot open for reading: ") + path);
        std::vector<float> th, mm, vv;
        int64_t st = 0;
        try {
            char magic[8];
            auto r32 = [&]() { uint32_t v = 0; if (std::fread(&v, 4, 1, f) != 1) throw Err("truncated checkpoint"); return v; };
            if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "AVAECKPT", 8) != 0) throw Err("not an AVAE checkpoint");
            if (r32() != 2) throw Err("unsupported checkpoint version");
            if ((int)r32() != h->M || (int)r32() != h->nz) throw Err("checkpoint architecture mismatch (modalities / n_z)");
            for (int m = 0; m < h->M; ++m) {
                const Mod& md = h->mods[m];
                if ((int)r32() != md.n_in || r32() != (uint32_t)md.hs.size()) throw Err("checkpoint architecture mismatch (n_input / depth)");
                for (int x : md.hs) if ((int)r32() != x) throw Err("checkpoint architecture mismatch (hidden width)");
                const bool conv = r32() != 0;
                const int g1 = (int)r32(), g2 = (int)r32();
                if (conv != md.conv || (conv && (g1 != h->cfg.mod[m].conv_gener[0] || g2 != h->cfg.mod[m].conv_gener[1])))
                    throw Err("checkpoint architecture mismatch (conv branch)");
            }
            uint64_t P = 0;
            if (std::fread(&P, 8, 1, f) != 1 || P != h->P_flat) throw Err("checkpoint parameter count mismatch");
            if (std::fread(&st, 8, 1, f) != 1) throw Err("truncated checkpoint");
            th.resize(P); mm.resize(P); vv.resize(P);
            if (std::fread(th.data(), 4, P, f) != P || std::fread(mm.data(), 4, P, f) != P || std::fread(vv.data(), 4, P, f) != P)
                throw Err("truncated checkpoint");
        } catch (...) { std::fclose(f); throw; }
        std::fclose(f);
        std::vector<float> I(h->P_int, 0.0f);
        convert_params<true>(h, th.data(), I.data()); host_to_master(h, h->off_theta, I);
        std::fill(I.begin(), I.end(), 0.0f); convert_params<true>(h, mm.data(), I.data()); host_to_master(h, h->off_m, I);
        std::fill(I.begin(), I.end(), 0.0f); convert_params<true>(h, vv.data(), I.data()); host_to_master(h, h->off_v, I);
        long long step = st;
        HIP_OK(hipMemcpy(&h->state()->step, &step, sizeof(step), hipMemcpyHostToDevice));
        run_adam(h, 1, nullptr);
        HIP_OK(hipDeviceSynchronize());
    });
}

int avae_synchronize(avae_handle* h) {
    return guarded(h, [&] { HIP_OK(hipDeviceSynchronize()); comm_check_error(h); });
}

int avae_timing_enable(avae_handle* h, int32_t on) {
    return guarded(h, [&] {
        HIP_OK(hipDeviceSynchronize());
        for (TimingRec& r : h->trecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        h->trecs.clear();
        h->timing = on != 0;
    });
}

// one line per launch name: "<name> <calls> <avg_ms> <min_ms>\n"
int avae_timing_report(avae_handle* h, char* buf, size_t buf_bytes) {
    return guarded(h, [&] {
        HIP_OK(hipDeviceSynchronize());
        // Each interval is the dispatch's own begin -> end (hipExtLaunchKernel events), the times rocprofv3
        // --kernel-trace reports.  While timing is on every step also times a null kernel (one 4-byte store),
        // reported as "_null_kernel": the floor of what one launch can measure.  Nothing is subtracted.
        const double cal_ms = 0.0;
        std::vector<double> sum(h->tnames.size(), 0.0), mn(h->tnames.size(), 1e30);
        std::vector<long> cnt(h->tnames.size(), 0);
        for (TimingRec& r : h->trecs) {
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, r.a, r.b));
            sum[r.launch_name] += ms; cnt[r.launch_name] += 1; mn[r.launch_name] = std::min(mn[r.launch_name], (double)ms);
        }
        std::string out;
        for (size_t i = 0; i < h->tnames.size(); ++i) {
            if (!cnt[i]) continue;
            char line[256];
            std::snprintf(line, sizeof(line), "%s %ld %.6f %.6f\n", h->tnames[i].c_str(), cnt[i], std::max(0.0, sum[i] / cnt[i] - cal_ms), std::max(0.0, mn[i] - cal_ms));
            out += line;
        }
        if (!buf || buf_bytes == 0) throw Err("null buffer");
        std::snprintf(buf, buf_bytes, "%s", out.c_str());
    });
}

int avae_comm_allreduce(avae_handle* h, int32_t bucket, void* stream) {
    return guarded(h, [&] {
        if (!h->comm_on) throw Err("avae_comm_allreduce: the replica has no library-owned collective (use_comm)");
        if (bucket < 0 || bucket >= h->n_buckets) throw Err("data-parallel bucket out of range");
        dp_allreduce(h, bucket, on_stream(h, stream));
    });
}

int avae_debug_fetch(avae_handle* h, const char* name, float* host_dst, size_t max_floats, size_t* n_floats) {
    return guarded(h, [&] {
        std::string n(name ? name : "");
        const void* src = nullptr;
        size_t cnt = 0;
        if (n == "eps") {     // internal rows are padded to a multiple of 4 floats: hand back dense [B][n_z]
            cnt = (size_t)h->B * h->nz;
            if (cnt > max_floats) throw Err("debug_fetch: destination too small");
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemcpy2D(host_dst, (size_t)h->nz * 4, h->at<void>(h->off_eps), (size_t)h->ld_eps * 4, (size_t)h->nz * 4, h->B, hipMemcpyDeviceToHost));
            if (n_floats) *n_floats = cnt;
            return;
        }
#ifdef AVAE_STAMPS
        else if (n == "stamps") { src = h->at<void>(h->off_stamps); cnt = (size_t)kStampLaunches * kStampBlocks * kStampWords * 2; }
#endif
        else if ((n[0] == 'E' || n[0] == 'D') && n.size() >= 4 && n.find('_') != std::string::npos) {
            // "E<m>_<k>" / "D<m>_<k>": the stored output of encoder / decoder hidden layer k of modality m, dense [B][width] as fp32
            // (tests: the relu decisions of the last forward pass)
            const int m = std::atoi(n.c_str() + 1), k = std::atoi(n.c_str() + n.find('_') + 1);
            if (m < 0 || m >= h->M || h->mods[m].conv || k < 0 || k >= h->mods[m].L) throw Err("debug_fetch: no such hidden layer");
            const Act& a = n[0] == 'E' ? h->mods[m].E[k] : h->mods[m].D[k];
            cnt = (size_t)h->B * a.width;
            if (cnt > max_floats) throw Err("debug_fetch: destination too small");
            HIP_OK(hipDeviceSynchronize());
            std::vector<unsigned char> raw((size_t)h->B * a.ld * h->es);
            HIP_OK(hipMemcpy(raw.data(), h->at<void>(a.rm), raw.size(), hipMemcpyDeviceToHost));
            for (int r = 0; r < h->B; ++r)
                for (int c = 0; c < a.width; ++c) {
                    float v;
                    if (h->es == 2) { const uint32_t u = (uint32_t)reinterpret_cast<const uint16_t*>(raw.data())[(size_t)r * a.ld + c] << 16; std::memcpy(&v, &u, 4); }
                    else v = reinterpret_cast<const float*>(raw.data())[(size_t)r * a.ld + c];
                    host_dst[(size_t)r * a.width + c] = v;
                }
            if (n_floats) *n_floats = cnt;
            return;
        }
        else if (n == "shadow_err") {
            // tests: the compute-dtype shadows the optimiser rewrites must BE the parameters (rounded once): W[r][c] and W^T[c][r]
            // against theta[r][c] for every dense layer and conv stage -> {max |W - theta|, max |W^T - theta|, layers checked, worst layer}.
            // A stale or misplaced shadow entry only shows in the NEXT step's arithmetic; this sees it at once, on any configuration.
            if (max_floats < 4) throw Err("debug_fetch: destination too small");
            HIP_OK(hipDeviceSynchronize());
            auto round_ct = [&](float v) {
                if (h->es == 4) return v;
                uint32_t u; std::memcpy(&u, &v, 4);
                if ((u & 0x7fffffffu) > 0x7f800000u) return v;                       // NaN: compared as is
                u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u;                   // round to nearest even, as the device conversion
                float r; std::memcpy(&r, &u, 4); return r;
            };
            auto elem = [&](const std::vector<unsigned char>& raw, size_t i) {
                float v;
                if (h->es == 2) { const uint32_t u = (uint32_t)reinterpret_cast<const uint16_t*>(raw.data())[i] << 16; std::memcpy(&v, &u, 4); }
                else v = reinterpret_cast<const float*>(raw.data())[i];
                return v;
            };
            float ew = 0.f, et = 0.f;
            int cnt = 0, worst = -1;
            auto check = [&](const Dense& d) {
                if (d.in <= 0 || d.out <= 0) return;
                std::vector<float> th((size_t)(d.in + 1) * d.ld);
                std::vector<unsigned char> w((size_t)(d.in + 1) * d.ldw * h->es), wt((size_t)d.out * d.ldt * h->es);
                HIP_OK(hipMemcpy(th.data(), h->at<float>(h->off_theta) + d.master, th.size() * 4, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(w.data(), h->at<void>(d.W), w.size(), hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(wt.data(), h->at<void>(d.Wt), wt.size(), hipMemcpyDeviceToHost));
                float a = 0.f, b2 = 0.f;
                for (int r = 0; r <= d.in; ++r)
                    for (int c = 0; c < d.out; ++c) {
                        const float want = round_ct(th[(size_t)r * d.ld + c]);
                        a = std::max(a, std::fabs(elem(w, (size_t)r * d.ldw + c) - want));
                        b2 = std::max(b2, std::fabs(elem(wt, (size_t)c * d.ldt + r) - want));
                    }
                if (a > ew || b2 > et) worst = cnt;
                ew = std::max(ew, a); et = std::max(et, b2);
                ++cnt;
            };
            for (const Mod& md : h->mods) {
                for (const Dense& d : md.enc) check(d);
                check(md.head);
                for (const Dense& d : md.dec) check(d);
                check(md.outl);
                if (md.conv) { for (const ConvStage& st : md.cenc) check(st.d); for (const ConvStage& st : md.cdec) check(st.d); }
            }
            host_dst[0] = ew; host_dst[1] = et; host_dst[2] = (float)cnt; host_dst[3] = (float)worst;
            if (n_floats) *n_floats = 4;
            return;
        }
        else if (n.rfind("mulv", 0) == 0 || n.rfind("g0_", 0) == 0) {
            const bool g0 = n[0] == 'g';
            const int m = std::atoi(n.c_str() + (g0 ? 3 : 4));
            if (m < 0 || m >= h->M) throw Err("debug_fetch: modality out of range");
            src = h->at<void>(g0 ? h->mods[m].g0 : h->mods[m].mulv); cnt = g0 ? (size_t)h->B * 3 * h->ld_eps : (size_t)h->B * 2 * h->nz;
        } else throw Err("debug_fetch: unknown tensor " + n);
        if (cnt > max_floats) throw Err("debug_fetch: destination too small");
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(host_dst, src, cnt * 4, hipMemcpyDeviceToHost));
        if (n_floats) *n_floats = cnt;
    });
}

}  // extern "C"
