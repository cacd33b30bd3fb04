// Device-visible descriptors shared by the host planner (avae_host.hip) and the gfx950 kernels
// (avae_kernels.hip).  A training step is a fixed list of launches; each launch runs ONE
// grouped kernel over a table of work items passed by value in its kernel arguments; nothing in it
// changes after avae_create, which is what lets the whole step be captured once as a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avae {

constexpr int kMaxMod = 4;
constexpr int kThreads = 256;          // 4 wavefronts of 64
constexpr int kTileBytesK = 128;       // bytes of K per staged tile row: 64 bf16 or 32 f32
constexpr int kRowAlign = 256;         // row counts of GEMM operands are padded to this (largest tile)
constexpr int kLatentRows = 16;        // rows per latent work-item tile (many small blocks: the item is latency-bound)
constexpr int kAdamRows = 16;          // k_adam tiles are kAdamRows x 64: measured 64 -> 32 -> 16 rows = 10.3 -> 9.8 -> 9.2 us on C2 (more,
                                       // lighter workgroups balance better over 256 CUs and overlap each other's load and store
                                       // phases); 8 rows (128 threads) 9.3 us and 16-byte segments in the transposed shadow
constexpr int kCostHist = 4096;        // ring of per-step costs kept on the device

// Work-item kinds.  Forward / dgrad kinds compute C[M,N] = sum_k A[m][k] * B[n][k] ("NT": both operands K-contiguous),
// K_WGRAD computes C[M,N] = sum_k A[k][m] * B[k][n] ("TN": both operands as stored, K = batch); they differ otherwise
// only in the LDS-staged epilogue.  Every result is stored once, row-major.
enum Kind : int {
    K_FWD_HIDDEN = 0,    // Y = act(X_aug . W_aug)                      -> Y               (vae_assoc.py:187,203,259,282)
    K_FWD_HEAD = 1,      // [mu|lv] = H_aug . Whead_aug; z = mu+exp(lv/2)*eps -> mulv, Z   (:217-221,:102-103)
    K_FWD_OUT_LOSS = 2,  // logits; recon loss + dLoss/dlogits         -> dA, cost partial (:293-303,:321-328)
    K_FWD_OUT_STORE = 3, // x_hat = sigmoid(logits) | logits           -> fp32 (inference)
    K_DGRAD_HIDDEN = 4,  // dA_prev = (dA . W^T) * act'(Y_prev)        -> dA_prev
    K_DGRAD_LATENT = 5,  // dz = dA . V1^T; dmu, dlv via reparam + static latent grads -> dH
    K_WGRAD = 6,         // dW_aug = X_aug^T . dA (bias grad = last row) -> fp32 gradient
    K_DGRAD_F32 = 10,    // dP = dA . W^T stored as fp32 (conv branch: patch gradients, summed by k_col2im)
    K_LATENT = 7,        // KL(q||N(0,I)) + association penalty: cost partials + static (mu,lv) grads (:335-366)
    K_COST = 8,          // fixed-order sum of the cost partials -> grad[cost slot]; bumps the step counter
    K_SERVE_Z = 11,      // serving (avae_generate): no product of its own (K = 0) -- the call's fp32 z rows (aux0, n_slots rows, dense
                         // [rows][nz]) go into the 32x64 tile's LDS image and the tail product (tail_*) is the decoder's first
                         // layer; workgroup 0 publishes the call's slot (partial = ServeSlot*, out0/out1/out2/aux1 = the call's
                         // output pointers of modality 0..3).  Launched per call with these fields patched by value.
};

struct WorkItem {
    int kind;
    int M, N, K;             // K is the padded extent (multiple of the K unit)
    int lda, ldb;            // in elements of the compute type
    int tiles_m, tiles_n;
    int act;                 // AVAE_ACT_*
    int ld0, ld1, ldx;
    int nz;
    int binary;
    int ksplit, kchunk;      // K_WGRAD with a long K (conv stages: K = batch x output pixels): the K range is cut into ksplit
                             // chunks of kchunk K tiles, chunk s writing its own fp32 slice out1 + s*M*ld0 (summed by k_reduce)
    int bias_row;            // K_WGRAD, > 0: the tiles cover rows [0, M) = the weight rows only and the bias gradient (row
                             // bias_row = M, the column sums of dA) comes from one extra MFMA per fragment with an all-ones A
                             // operand in the first tile row -- set by the host when M+1 rows would cost a whole tile row more
    int bias_ep;             // forward kinds on the 8-wave tiles, set by the host when fan-in % K-unit == 0: K stops at the fan-in
                             // (the bias row would cost a whole K tile more) and the epilogue adds bias[n] = aux1[n] instead
    int kin;                 // forward kinds: fan-in of the layer (the bias is row kin of W_aug)
    int tile_off, tile_cnt;  // K_WGRAD launches only (TnItem)
    int slot_base;           // first cost-partial slot written by this item
    int n_slots;             // K_COST: number of partial slots to sum
    int n_mod;               // K_LATENT
    int bump_step;           // K_COST: 1 in training graphs
    float scale;             // w/B_global (mean terms) or w (sum terms)
    float lambda;            // K_LATENT
    float inv_bg;            // 1/B_global
    const void* A;
    const void* B;
    void* out0;
    void* out1;
    void* out2;
    const void* aux0;
    const void* aux1;
    const void* aux2;
    float* partial;          // cost partial slots
    const float* eps;        // K_LATENT: the step's eps [B][ldx]
    float wts[kMaxMod];      // K_LATENT: per-modality cost weights
    // Tail product (small nets, 32x64 head tiles): K_FWD_HEAD / K_DGRAD_LATENT items leave a narrow result (z, or [dmu | dlv]) whose
    // only consumer is a layer that multiplies over one or two K tiles (the decoder's first layer, K = n_z + 1; the heads' input
    // gradient, K = 2 n_z) and used to be a launch of its own.  With tail_mode set the same workgroup multiplies its 32 result rows, still in
    // LDS, with that layer's weight shadow (B fragments straight from global memory into registers) and stores the layer's
    // output: forward  out = act(z_aug . W^T),  dgrad  out = ([dmu|dlv] . W^T) * act'(aux).  The item's tiles_n then counts tail
    // SLICES of 64 output columns: slice s of a row block is a workgroup of its own that recomputes the 32x64 head tile (cheap: the
    // slices of a row block share an XCD and its L2) and multiplies columns [64 s, 64 s + 64); slice 0 also stores the head's results.
    const void* tail_w;      // [tail_n][tail_ldw]: K-contiguous weight shadow of the consuming layer
    void* tail_out;          // [rows][tail_ldo]
    const void* tail_aux;    // mode 2: stored output of the layer the gradient belongs to [rows][tail_ldx]
    int tail_mode;           // 0: none; 1: forward hidden layer; 2: dgrad of a hidden layer
    int tail_n;              // units of the consuming layer (columns of its output)
    int tail_ldw, tail_ldo, tail_ldx;
    int tail_act;
    int tail_kt;             // K tiles of the consuming layer's product: 1, or 2 (fp32 [dmu | dlv] with n_z > 16)
    int conv;                // > 0: the A operand is an implicit patch matrix, descriptor conv - 1 of the launch's ConvA table (below)
    // K_LATENT reuses the pointer fields: [mu|lv] inputs of modality 0..3 = A, B, aux0, aux1;
    // static-gradient outputs g0 of modality 0..3 = out0, out1, out2, aux2.
    // K_COST: scale = lr, lambda = beta1, inv_bg = beta2 (it also publishes this step's Adam lr_t).
};

// One launch = up to kMaxItemsPerLaunch work items, passed BY VALUE in the kernel-argument
// segment together with the first block index of every item: a workgroup finds and reads its item
// without a chain of dependent loads from HBM, and no device-side table has to be kept in sync.
// Implicit-GEMM conv / transposed conv (reference vae_assoc.py:169-199,249-278, deconv.py:107): the patch matrix
//   P[(b,oh,ow)][(kh,kw,ci)] = X[b, (oh*so + kh - pad)/d, (ow*so + kw - pad)/d, ci]   (0 where not divisible / out of range)
// is never stored: a GEMM whose A operand is P computes the address of every 16-byte chunk it stages -- Cin*elem_size is a multiple
// of 16, so a chunk is 8 (bf16) / 4 (fp32) channels of ONE tap of one source pixel; an invalid tap reads a 16-byte block of
// zeros; the chunk at column K = k*k*Cin reads {1, 0, ...} when `ones` (the bias column of a stored patch matrix), zeros beyond.
// The LDS-DMA takes a per-lane source address, so the staged image is byte for byte the one a stored P would give.
// One geometry covers conv (so = stride, d = 1), transposed conv in gather form (so = 1, d = stride, pad = k-1-pad_before, the
// filter flipped in the matrix), flatten + dense (k = map size, VALID) and, with so <-> d swapped on the stage's OUTPUT gradient,
// every input gradient (adjoint form).  Divisions are multiplications by magic numbers (mg_x = floor(2^32 / x) + 1).
struct ConvA {
    const void* src;         // source map, compute dtype: element (b, pixel, c) at b*src_sb + pixel*src_sp + c
    const void* consts;      // [16 B of zeros][16 B {1, 0, ...} in the compute dtype]
    int M, K;                // rows B*OH*OW (rows beyond read zeros), k*k*Cin
    int OHW, OW, IH, IW, Cin;
    int k, so, d, pad;
    int src_sb, src_sp;
    int ones;
    unsigned mg_ohw, mg_ow, mg_k, mg_cin;
};
constexpr int kMaxConvA = 12;
constexpr int kMaxItemsPerLaunch = 12;
constexpr int kMaxPf = 4;
// Grid: x = tile slot inside an item (a multiple of 8, at least the largest item's tile count), y = item.
// A workgroup knows its item from blockIdx.y alone, so its WorkItem is ONE scalar-load round trip into the
// kernel-argument segment away; workgroups beyond their item's last tile exit at once.
struct LaunchArgs {
    int n_items;
    int grid_x;
    int sched;               // diagnostics (AVAE_SCHED): 0 = the shipped in-loop order of the 8-wave tiles, 1 = round 1's order
    int n_pf;                // 8-wave NT tiles: weight panels of the NEXT launch to pull into the Infinity Cache (0..kMaxPf)
    const void* pf_ptr[4];   // ... their first bytes
    int pf_lines[4];         // ... and sizes in 128-byte lines
    const ConvA* conv_tab;   // device table of the launch's implicit patch matrices (WorkItem::conv indexes it; null: none)
    WorkItem items[kMaxItemsPerLaunch];
};

// The weight-gradient launches (K-major operands, K_WGRAD only) carry a compact item: 80 bytes instead of 192, so every
// weight gradient of a step fits ONE launch's kernel-argument segment (4 KiB) and the tiles of all layers share the CUs'
// rounds (C4: 800 tiles in 4 rounds of one launch instead of 2 + 2 rounds of two).
struct TnItem {
    const void* A;           // X_aug   [K = batch rows][lda]
    const void* B;           // dA      [K][ldb]
    float* out;              // fp32 gradient [M][ld0]; split-K: slice s at out + s*M*ld0
    int M, N, K;
    int lda, ldb, ld0;
    int tiles_m, tiles_n;
    int ksplit, kchunk;
    int bias_row;            // see WorkItem::bias_row
    int tile_off, tile_cnt;  // this entry covers tiles [tile_off, tile_off + tile_cnt) of the item's tile list: the host cuts
    int conv;                // items with many tiles into several entries so that the grid (x = longest entry) is not mostly padding.
                             // conv > 0: A (the K-major operand X) is the implicit patch matrix conv - 1 of TnLaunchArgs::conv_tab
    // k_small_tn with the optimiser in its epilogue (TnLaunchArgs::adam): the layer's two compute-dtype shadows
    void* W; void* Wt;       // [M][ldw] (dgrad operand), [N][ldt] (forward operand)
    int ldw, ldt;
};
constexpr int kMaxTnItems = 32;
constexpr int kTnPieces = 8;
struct TnPiece { unsigned short item, tile_off, cum_end, pad; };      // tiles [tile_off, ...) of items[item]; cum_end = positions of the XCD up to and including this piece
struct DevState;
struct TnLaunchArgs {
    int n_items;
    int grid_x;
    int xcd_group;           // G = 1, 2, 4 or 8 items share the 8 XCDs: each item's tiles run on 8/G of them (see k_grouped)
    int sched;               // as LaunchArgs::sched
    // Small nets, one replica: the Adam step rides in this launch (k_small_tn<CT, true>) -- a tile's gradient block is applied to the
    // parameters as it leaves the accumulators, and k_adam's launch disappears.  theta / m / v of an element sit at its gradient's
    // address + a constant number of floats (the four fp32 arrays share the master layout).
    int adam;
    long long d_theta, d_m, d_v;
    float beta1, beta2, eps;
    const DevState* st;      // lr_t of this step (published by the K_COST item earlier in the step)
    const ConvA* conv_tab;
    // k_small_tn, XCD-owned pieces (xcd_pieces = 1): `items` holds WHOLE layers and XCD c runs the tiles of pieces[c][0..] in that
    // order -- workgroup x (XCD x % 8, position x / 8 on it) serves the piece whose [previous cum_end, cum_end) holds its position.
    // A layer's tiles then meet in ONE L2 (or two, where a layer is cut to balance the XCDs): its X / dA column panels come from
    // HBM once instead of once per XCD (C2: 11.8 of the launch's 33.3 MB of reads were such re-reads).
    int xcd_pieces;
    TnPiece pieces[8][kTnPieces];
    TnItem items[kMaxTnItems];
};
static_assert(sizeof(TnLaunchArgs) <= 4096, "kernel arguments are passed by value: 4 KiB");

struct DevState {
    long long step;          // number of applied Adam steps
    float last_cost;
    float lr_t;              // lr*sqrt(1-b2^t)/(1-b1^t) of the step being applied (written by K_COST)
    float cost_hist[kCostHist];
};

// One dense layer's optimiser tile table entry (Adam + compute-dtype shadow refresh).
struct AdamItem {
    float* theta;
    float* m;
    float* v;
    const float* g;
    void* W;                 // shadow [in+1 (padded)][ld]   : B operand of dgrad
    void* Wt;                // shadow [out (padded)][ldt]   : B operand of forward
    int rows, cols;          // in+1, out
    int ld, ldt;
    int tiles_r, tiles_c, tile_base;
    int ldw;                 // leading dim of the W shadow (>= ld: see spread_ld in avae_host.hip)
    // transposed-conv stages on the adjoint-frame route (adj_k > 0): the same pass also writes the adjoint filter shadows k_wadj
    // used to build in a launch of its own -- element (row (kh_m, kw_m, ci), col co) of the matrix goes, with kh = k-1-kh_m,
    // kw = k-1-kw_m and r = (kh*k + kw)*cols + co, to Wadj[ci][r] and Wf[r][ci]
    void* Wadj; void* Wf;
    int ldadj, ldf, adj_k, adj_cin;
};

constexpr int kMaxAdamItems = 80;      // kMaxMod * (2*AVAE_MAX_HIDDEN + 2)
struct AdamArgs {
    const AdamItem* items;
    int n_items;
    int base[kMaxAdamItems];           // first tile of every item (by value: no dependent scan through HBM)
    int mode;                // 0: Adam update; 1: refresh shadows from theta only
    int book;                // 1: this launch books the step's cost (last_cost, cost history)
    float lr, beta1, beta2, eps;
    DevState* st;
    const float* cost_src;   // grad[cost slot]
};

// Input staging ("prep"): fp32 rows -> compute-dtype copy (+ exact fp32 copy for the losses),
// and the eps tensor (copied from the caller or generated with Philox4x32-10).
struct PrepSeg {
    const float* src; int src_ld;
    int rows, cols;
    float* dst32; int ld32;        // nullable
    void* dstc; int ldc;           // row-major compute dtype
    int tiles_r, tiles_c, tile_base;
};
struct PrepArgs {
    PrepSeg seg[kMaxMod];
    int n_seg;
    int total_tiles;               // tiles of all segments; eps blocks follow
    // eps
    const float* eps_src;          // nullable -> Philox
    float* eps_dst;                // nullable -> no eps work
    int eps_rows, nz, eps_blocks, eps_ld;   // eps_ld = roundup(nz, 4): row stride of the internal eps buffer
    int row_offset;
    unsigned long long seed;
    const DevState* st;
    unsigned long long stream_salt;   // distinguishes train / eval / reconstruct draws
    // batched staging (avae_train_steps): the launch stages n_steps consecutive batches at once.  Batch j reads rows
    // [j*rows, (j+1)*rows) of every source (and of eps_src), writes staging set j (every destination moved by
    // j*set_stride bytes) and draws eps for step st->step + j.  blocks_per_step = total_tiles + eps_blocks.
    int n_steps, blocks_per_step;
    long long set_stride;
};

// Fixed-order sum of the split-K slices of a weight gradient: dst[i] = sum_s src[s*stride + i]  (no atomics: reproducible).
struct ReduceSeg { float* dst; const float* src; int n, parts; long long stride; int block_base;
                   int dst_ld;      // dst_ld > 0: element i goes to dst[i * dst_ld] (a one-column gradient in a padded matrix)
                   int colsum;      // 1: k_colsum's tree (many short partial vectors), 0: k_reduce's slice sum -- one launch may carry both (k_sums)
                   // slice sums of an adjoint-frame filter gradient [Cin][(kh', kw', co)] (leading dim perm_ldga) go straight to the
                   // gradient buffer's layout [(kh, kw, ci)][co] (leading dim perm_ld, kh = k-1-kh'): perm_k > 0, dst = that matrix
                   int perm_k, perm_cin, perm_cout, perm_ldga, perm_ld; };
constexpr int kMaxReduceSegs = 40;          // up to 9 split conv stages per modality x 4 modalities (+ bias sums)
struct ReduceArgs { ReduceSeg seg[kMaxReduceSegs]; int n_seg; };

// Conv / transposed-conv layers (reference vae_assoc.py:169-199,249-278, deconv.py:107) run on the
// same GEMM kernel through explicit patch matrices.  One geometry covers both directions:
//   P[(b,oh,ow)][(kh,kw,ci)] = X[b, (oh*so + kh - pad)/d, (ow*so + kw - pad)/d, ci]   (0 if not divisible / out of range)
// conv: so = stride, d = 1, pad = TF pad-before;  transposed conv: so = 1, d = stride, pad = k-1-pad_before with
// the filter flipped when it is laid out as a matrix;  flatten+dense: k = map size, VALID.
struct ConvGeom {
    int B, IH, IW, Cin, OH, OW, k, so, d, pad;
    int src_sb, src_sp;        // element strides of the source: batch, pixel (channel stride 1)
    int ones;                  // append the constant-1 bias column at index k*k*Cin
};
// k_gather (im2col): source activations -> patch matrix P (row-major + transposed), compute dtype
struct GatherSeg {
    ConvGeom g;
    const void* src;
    void* P; int ldp;
    int tiles_r, tiles_c, tile_base;
    int cl, rpt;                      // tile shape, see Col2imSeg
};
struct GatherArgs { GatherSeg seg[kMaxMod]; int n_seg; };
// k_col2im: fp32 patch gradients dP -> gradient of the layer input, times act'(stored input), written
// row-major + transposed in the compute dtype; latent mode turns dz into [dmu | dlv] (reparameterisation).
struct Col2imSeg {
    ConvGeom g;
    const float* dP; int lddp;
    const void* yprev; int ldy;      // stored output of the producing layer (nullable -> identity)
    int act;                          // AVAE_ACT_* of the producing layer
    void* dA; int lda;                // [B*IH*IW][lda]
    const float* g0; int nz;          // latent mode when g0 != nullptr
    const void* bias; int bias_ld;    // forward mode when bias != nullptr (a transposed conv as scatter product + overlap-add):
                                      // dA = act(sum + bias[c * bias_ld]) instead of sum * act'(yprev)
    int tiles_r, tiles_c, tile_base;
    int cl, rpt;                      // tile shape: cl (1..16, power of two) lanes of 4 columns, rpt (1 or 4) rows per thread:
                                      // (256/cl)*rpt rows x 4*cl columns -- narrow outputs get tall tiles, short ones many tiles
};
struct Col2imArgs { Col2imSeg seg[kMaxMod]; int n_seg; };

// The last transposed-conv stage (G2 channels -> ONE 28x28 map, vae_assoc.py:274-278) as direct kernels.  On the GEMM path
// its single output channel is one useful column of a 32-wide tile, its patch matrix is 180 MB and its fp32 patch gradients
// 321 MB per step at batch 256; computed directly the stage reads its 1.6 MB input and the 400 filter taps.
// One workgroup per image, the image staged in LDS:
//   mode 0  forward    : Y[p] = act(bias + sum_taps X[b, ih, iw, :] . F[kh, kw, :])
//   mode 1  input grad : dX[b, ih, iw, c] = act_in'(X) * sum_taps dY[b, oh, ow] * F[kh, kw, c]
//   mode 2  filter grad: part[image][Kp]   (summed over images by k_colsum)
constexpr int kThinIn = 4096, kThinOut = 1024, kThinF = 1024;   // k_thin: one image staged in LDS (floats)
constexpr int kThinSplit = 4;          // workgroups per image (each stages the image and takes a quarter of the outputs): one
                                       // workgroup per CU leaves the LDS latency of these loops exposed
struct ThinSeg {
    ConvGeom g;
    const void* X;           // stage input = stored output of the producing stage, [b*src_sb + pixel*src_sp + c]
    const void* Wt;          // filter shadow [1][ldt]: taps (kh, kw, c), bias at index k*k*Cin
    void* Y; int ldy;        // stage output, column 0 of [rows][ldy]
    int act, act_in;         // transfer function of this stage / of the producing stage
    const void* dY; int lddy;   // gradient w.r.t. this stage's pre-activation output, column 0
    void* dX; int lddx;      // gradient w.r.t. the producing stage's pre-activation output [pixels][lddx]
    float* part;             // mode 2: [blocks][Kp]
    int Kp;                  // roundup(k*k*Cin + 1, 4)
    int block_base;
    int img_y, img_dy;       // > 0: elements from one image's map to the next in Y / dY (dense rows); 0: OH*OW*ldy / OH*OW*lddy
    int split;               // workgroups per image: kThinSplit, or 1 on the block-structured path below
    int fast;                // the reference's geometry (k 5, stride-2 transposed conv in gather form: so 1, d 2, pad 3, OH = 2 IH, 8 or 16
                             // input channels): one workgroup per image; forward = a thread per 2x2 output block over its 3x3 input
                             // neighbourhood (every tap test folds at compile time), input gradient = a thread per (pixel, channel quad)
                             // over the 5x5 window, filter gradient = a thread per (tap, tenth of the pixels) for all channels
};
inline bool thin_fast_geometry(const ConvGeom& g) {
    return g.k == 5 && g.so == 1 && g.d == 2 && g.pad == 3 && g.OH == 2 * g.IH && g.OW == 2 * g.IW && (g.Cin == 8 || g.Cin == 16) && g.ones;
}
struct ThinArgs { ThinSeg seg[kMaxMod]; int n_seg; int mode; };
void launch_thin(int compute_dtype, const ThinArgs& a, int n_blocks, hipStream_t s);
void launch_colsum(const ReduceArgs& a, int n_blocks, hipStream_t s);

// Adjoint filter shadow of a transposed-conv stage: Wadj[ci][(kh', kw', co)] = Wt[co][((k-1-kh')*k + (k-1-kw'))*Cin + ci], the B
// operand of the stage's input gradient computed as a GEMM on the patch matrix of its OUTPUT gradient.
// ... and its transpose Wf[(kh', kw', co)][ci], the B operand of the stage's FORWARD pass as a scatter product (below).
struct WadjSeg { const void* Wt; void* Wadj; void* Wf; int ldt, ldadj, ldf, k, Cin, Cout, block_base; };
struct WadjArgs { WadjSeg seg[2 * kMaxMod]; int n_seg; };
void launch_wadj(int compute_dtype, const WadjArgs& a, int n_blocks, hipStream_t s);
// Filter gradient of such a stage, computed in the adjoint frame Gadj[ci][(kh', kw', co)] = sum_pixels X[p][ci] * Padj[p][...],
// moved into the gradient buffer's layout G[(kh, kw, ci)][co] (kh = k-1-kh').
struct GpermSeg { const float* Gadj; float* G; int ldga, ld, k, Cin, Cout, block_base; };
struct GpermArgs { GpermSeg seg[2 * kMaxMod]; int n_seg; };
void launch_gperm(const GpermArgs& a, int n_blocks, hipStream_t s);
// Column sums of a compute-type matrix, first level: workgroup j adds rows j, j + blocks, ... -> part[j][cols4] (the bias
// gradient of a conv stage = column sums of its output gradient; k_colsum adds the partial sums in order).
struct RowsumSeg { const void* src; float* part; int ld, rows, cols, cols4, cpow, n_blocks, block_base; };
struct RowsumArgs { RowsumSeg seg[2 * kMaxMod]; int n_seg; };
void launch_rowsum(int compute_dtype, const RowsumArgs& a, int n_blocks, hipStream_t s);

// Serving (avae_generate): the captured decode graph's output launch reads the caller's pointers and row count from this
// device-side slot (written by the per-call staging launch), so that one graph serves every call without re-parameterising its nodes.
struct ServeSlot {
    const float* z;                // [rows][n_z] fp32, dense
    float* out[kMaxMod];           // per modality [rows][n_input] fp32, dense
    int rows;
    int pad;
};
struct ServeArgs {               // the per-call staging launch: z -> Z of every modality (rows beyond the call's rows zero)
    ServeSlot* slot;
    int n_mod, nz, bucket;         // rows of the captured plan (>= the call's rows)
    void* Z[kMaxMod]; int ldz[kMaxMod];            // decoder inputs, compute dtype, [bucket][ldz]
    int blocks_per_mod;
};
void launch_serve(int compute_dtype, const ServeArgs& a, const ServeSlot& call, int n_blocks, hipStream_t s);
// k_serve_in: the staging launch that also runs the decoder's first layer (small nets)
struct ServeInMod { const void* w; void* out; int n, ldw, ldo, kt, act, slices; };   // first decoder layer of one modality: weight shadow [n][ldw], output [bucket][ldo]
struct ServeInArgs {
    ServeSlot call;                // this call's pointers and row count (by value: the per-call eager launch)
    const ServeSlot* rec;          // != null: read the call from this record instead -- a slot of the handle's pinned-host ring, written by
                                   // the host just before it replays the graph this launch is the first node of (no eager launch per call)
    unsigned long long* consumed;  // pinned-host word: number of ring records the device has read (the host's flow control)
    unsigned long long* count;     // device-side counter behind it
    ServeSlot* slot;               // where the graph's output launch reads them
    int n_mod, nz, bucket, tiles_m;
    ServeInMod mod[kMaxMod];
};
void launch_serve_in(int compute_dtype, const ServeInArgs& a, int grid_x, hipStream_t s);

void launch_gather(int compute_dtype, const GatherArgs& a, int n_blocks, hipStream_t s);
void launch_col2im(int compute_dtype, const Col2imArgs& a, int n_blocks, hipStream_t s);

// ---- gradient exchange (avae_comm.hip)
// One-shot all-reduce over hipIpc peers ("a hand-rolled P2P reduce-scatter/all-gather over hipIpc peers"):
// every rank owns an exchange block (uncached device memory, mapped by every peer); a range of the gradient buffer is cut into
// `world` shards of 8-float granules, shard j is reduced by rank j:
//   phase 1  push     rank r stores its values of shard j into slot r of rank j's block           (all 7 links at once)
//   phase 2  reduce   rank j adds slots 0..world-1 in RANK ORDER (its own from g), stores the sum into every rank's result area
//   phase 3  collect  every rank copies the other shards' sums from its result area into g
// Workgroup w of every rank works on chunk w of every shard and signals / waits on workgroup w of its peers only (a flag word per
// (workgroup, source rank), value = the call's sequence number), so there is no grid barrier and no reset; every wait is bounded by
// a wall-clock timeout that raises the block's error word instead of hanging.  Cross-GPU traffic is stores only (posted writes).
constexpr int kMaxWorld = 8;
constexpr int kIpcThreads = 256;
struct IpcArgs {
    float* g;                       // local gradient buffer (master layout + cost slot)
    long long off, granules;        // range = floats [off, off + 8*granules)
    long long cost_idx;             // >= 0: g[cost_idx] is summed over the ranks in fp32 beside the range
    int world, rank, wire_bf16, blocks;
    unsigned char* peer[kMaxWorld]; // every rank's exchange block as mapped in this process (peer[rank] = the own block)
    long long off_flag1, off_flag2; // unsigned [blocks][kMaxWorld]: flag of (workgroup, source rank)
    long long off_cost;             // float [kMaxWorld]: the ranks' local costs
    long long off_seq;              // unsigned [blocks]: calls completed by workgroup w (local use only)
    long long off_err;              // unsigned: nonzero after a timed-out wait (local use only)
    long long off_slots, slot_stride, off_res;   // bytes
    unsigned long long timeout_ticks;            // of the 100 MHz wall clock
};
void launch_ipc_allreduce(const IpcArgs& a, hipStream_t s);
// bf16 wire format for the RCCL path: gradient range -> bf16 wire buffer, and back after the all-reduce
void launch_wire_pack(const float* g, void* wire, long long n, hipStream_t s);
void launch_wire_unpack(float* g, const void* wire, long long n, hipStream_t s);

// Diagnostic build only (-DAVAE_STAMPS): thread 0 of every block records s_memrealtime (100 MHz)
// at kernel entry / after the item lookup / after the first staged tile / after the K loop / at
// the end, plus s_memtime (shader clock) at entry and end.  No stamp executes in the product build.
struct LaunchEvents { hipEvent_t start, stop; };
extern thread_local LaunchEvents t_launch_events;   // armed by the host in timing mode, consumed by the next launch

constexpr int kStampLaunches = 32, kStampBlocks = 1024, kStampWords = 8;

// launchers implemented in avae_kernels.hip
int tile_lds_bytes(int tile_cfg, bool two_c_tiles);
void launch_grouped(int compute_dtype, int tile_cfg, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes,
                    DevState* st, hipStream_t s, unsigned long long* stamps = nullptr, int launch_id = 0);
void launch_grouped_tn(int compute_dtype, int tile_cfg, const TnLaunchArgs& args, int grid_x, int grid_y, int lds_bytes,
                       DevState* st, hipStream_t s, unsigned long long* stamps = nullptr, int launch_id = 0);
void launch_adam(int compute_dtype, const AdamArgs& a, int n_blocks, hipStream_t s);
void launch_prep(int compute_dtype, const PrepArgs& a, hipStream_t s);
const void* prep_kernel(int compute_dtype);          // for hipGraphExecKernelNodeSetParams on the captured staging node
void launch_fill(void* base, int elem_bytes, unsigned bits, long long start, long long stride, int count, hipStream_t s);
void launch_sums(const ReduceArgs& a, int n_blocks, hipStream_t s);
int small_head_lds_bytes();
void launch_small_latb(int compute_dtype, int act, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, DevState* st, hipStream_t s,
                       unsigned long long* stamps, int launch_id);
void launch_small_head(int compute_dtype, int act, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, hipStream_t s,
                       unsigned long long* stamps, int launch_id);
void launch_small_tn(int compute_dtype, const TnLaunchArgs& args, int grid_x, int grid_y, hipStream_t s, unsigned long long* stamps, int launch_id);
void launch_small_loss(int compute_dtype, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, hipStream_t s, unsigned long long* stamps, int launch_id);
void launch_small(int compute_dtype, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, hipStream_t s, unsigned long long* stamps, int launch_id);
void launch_reduce(const ReduceArgs& a, int n_blocks, hipStream_t s);
void launch_chain2(int compute_dtype, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, unsigned* counters, unsigned* err, hipStream_t s);

}  // namespace avae
