// gfx950 (MI355X, CDNA4) kernels of the associative-VAE training path.
//
// One grouped kernel does every matrix product of the step, on the operands AS THEY ARE STORED (row-major, K padded with
// zeros to 128-byte units); no transposed copy of any activation or gradient exists:
//     forward   Y  = X_aug . W_aug        "NT"  A = X_aug [B][in+1]     B = W_aug^T shadow [out][in+1]   (both K-contiguous)
//     dgrad     dX = dA . W^T             "NT"  A = dA    [B][out]      B = W_aug shadow   [in+1][out]
//     wgrad     dW_aug = X_aug^T . dA     "TN"  A = X_aug [B][in+1]     B = dA [B][out]    (K = batch: K-major LDS images,
//                                               fragments by ds_read_b64_tr_b16)
// The bias is the last row of W_aug and every activation carries a constant-1 column, so the bias add and the bias gradient
// fall out of the same MFMA products.
//
// Per workgroup: a BM x BN output tile -- 32x32 ... 128x128 with 4 wave64s (2x2), or 256x128 with 8 wave64s (4x2) --
// v_mfma_f32_16x16x32_bf16 (bf16 operands) or v_mfma_f32_16x16x4_f32 (exact fp32), fp32 accumulation in registers, a 2- to
// 4-stage LDS-DMA ring (global_load_lds_dwordx4, XOR-swizzled 128-byte rows: conflict-free ds_read_b128, up to RING-1 K tiles
// in flight), and an epilogue that fuses the activation / reparameterisation / loss / gradient maths and writes every result
// ONCE, row-major, with 16-byte stores.  What bounds the loop and why the tiles are what they are: DESIGN.md, section 4.
//
// Reference maths: /root/reference/vae_assoc.py:163-222 (encoder), :243-304 (decoder),
// :306-371 (losses), :373-374 (Adam); restated for CPU in oracle/vae_assoc_oracle.py.
#include "avae_device.h"
#include <hip/hip_ext.h>
#include "../../include/avae.h"

namespace avae {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // native vector: stays in registers (HIP's uint4 struct does not)


// ------------------------------------------------------------------ element helpers
template <typename CT> __device__ __forceinline__ CT to_ct(float v);
template <> __device__ __forceinline__ float to_ct<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 to_ct<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float bf16_bits_to_float(unsigned b) { return __uint_as_float(b << 16); }

template <typename CT> __device__ __forceinline__ void load4(const CT* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<__bf16>(const __bf16* p, float v[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = bf16_bits_to_float(t.x & 0xffffu); v[1] = bf16_bits_to_float(t.x >> 16);
    v[2] = bf16_bits_to_float(t.y & 0xffffu); v[3] = bf16_bits_to_float(t.y >> 16);
}

template <typename OT> __device__ __forceinline__ void store4(OT* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const float v[4]) {
    bf16x4 t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = t;
}
// Store up to 4 consecutive elements; a partial quad is written element-wise so that the
// padding (and the constant-1 column that lives in it) is never touched.
template <typename OT> __device__ __forceinline__ void store_row(OT* p, const float v[4], int nvalid) {
    if (nvalid >= 4) { store4<OT>(p, v); return; }
#pragma unroll
    for (int e = 0; e < 3; ++e) if (e < nvalid) p[e] = to_ct<OT>(v[e]);
}

// 16-byte vector access of VW consecutive elements (VW = 8 bf16 or 4 fp32 per lane): a 16-B store
// costs the same issue slot as an 8-B one and the epilogues are store-issue bound.
template <typename T> struct Vec16 { static constexpr int VW = 16 / (int)sizeof(T); };
template <typename AT, int VW> __device__ __forceinline__ void load_vec(const AT* p, float* v);
template <> __device__ __forceinline__ void load_vec<float, 4>(const float* p, float* v) { load4<float>(p, v); }
template <> __device__ __forceinline__ void load_vec<float, 8>(const float* p, float* v) { load4<float>(p, v); load4<float>(p + 4, v + 4); }
template <> __device__ __forceinline__ void load_vec<__bf16, 8>(const __bf16* p, float* v) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bf16_bits_to_float(t[i] & 0xffffu); v[2 * i + 1] = bf16_bits_to_float(t[i] >> 16); }
}
template <> __device__ __forceinline__ void load_vec<__bf16, 4>(const __bf16* p, float* v) { load4<__bf16>(p, v); }
template <typename OT, int VW> __device__ __forceinline__ void store_vec(OT* p, const float* v, int nvalid);
template <> __device__ __forceinline__ void store_vec<float, 4>(float* p, const float* v, int nvalid) { store_row<float>(p, v, nvalid); }
template <> __device__ __forceinline__ void store_vec<__bf16, 8>(__bf16* p, const float* v, int nvalid) {
    if (nvalid >= 8) {
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
        const bf16x8_t t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
        // Written through to memory (sc0 sc1): the big launches leave 17-34 MB of results dirty in L2, which the
        // end-of-kernel release then has to write back before the next launch may start (3.4 us gap); stores that go
        // through as they are issued take 2 us off every big forward launch and change nothing for the small ones.
        typedef __attribute__((ext_vector_type(4))) unsigned u4_;
        const u4_ raw = __builtin_bit_cast(u4_, t);
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(raw) : "memory");   // (s_nop 1: the store must have read its data registers before hipcc's next instruction may write them)
        return;
    }
#pragma unroll
    for (int e = 0; e < 7; ++e) if (e < nvalid) p[e] = (__bf16)v[e];
}

template <typename CT> __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c);
template <> __device__ __forceinline__ void mma<__bf16>(const u32x4& a, const u32x4& b, f32x4& c) {
#ifdef AVAE_ABL_NO_MFMA     /* diagnostic: keep the operands live, skip the matrix pipe (results are garbage) */
    asm volatile("" ::"v"(a), "v"(b));
    return;
#endif
    // lane l: A[row l&15][k 8*(l>>4)..+7], B[k 8*(l>>4)..+7][col l&15]; C col l&15, row 4*(l>>4)+reg
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(const u32x4& a, const u32x4& b, f32x4& c) {
    // 16x16x4 f32: lane l holds A[l&15][k = l>>4].  The lane's 4 consecutive floats are fed to 4
    // MFMAs; MFMA e therefore sums k in {4q+e}: a permutation of the 16-float slab, identical
    // for A and B, so the product is the exact fp32 fma chain over all 16 k.
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
}

// SWAP: the operands trade places, so the accumulator holds the transposed block (row on the lane, 4 consecutive columns in the
// registers): see the register epilogue below.  The fragment layouts of A and B are mirror images, so the same LDS reads serve.
template <typename CT, bool SWAP> __device__ __forceinline__ void mma_sel(const u32x4& a, const u32x4& b, f32x4& c) {
    if constexpr (SWAP) mma<CT>(b, a, c); else mma<CT>(a, b, c);
}

// ---- K-major ("TN") operand images, used by the weight gradients: dW[m][n] = sum_k X[k][m] * dA[k][n] reads X and dA
// as they are stored (row = batch sample k), so no transposed copy of any activation or gradient has to exist.
// Image of one operand part: sub-images of EPR k-rows x 128 bytes (EPR = 64 bf16 / 32 fp32 columns), sub-image s
// holding columns [s*EPR, (s+1)*EPR).  16-byte chunk c of row k lives at chunk c ^ tn_swz(k):
//   bf16: fragments come from ds_read_b64_tr_b16 (per 16 lanes a 4-row x 16-column block, delivered column-major:
//         lane 4q+p supplies the address of row q, columns 4p..4p+3 and receives column (lane & 15) of the 4 rows);
//         a 32-lane half reads rows {r, r+2 | r+1, r+3} of two blocks 8 rows apart in the same columns, which the
//         swizzle bits (row bit 1 -> chunk bit 1, row bit 3 -> chunk bit 2) spread over all 64 banks;
//   fp32: plain ds_read_b32, lane (m, g) takes rows 4g..4g+3 of a 16-row slab; odd rows are shifted by 4 chunks.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
template <typename CT> __device__ __forceinline__ int tn_swz(int k);
template <> __device__ __forceinline__ int tn_swz<__bf16>(int k) { return (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2); }
template <> __device__ __forceinline__ int tn_swz<float>(int k) { return (k & 1) << 2; }
// byte offset, inside the operand part, of the lane's first read for the fragment whose 16 columns start at col0
template <typename CT> __device__ __forceinline__ int tn_frag_off(int col0, int lane);
template <> __device__ __forceinline__ int tn_frag_off<__bf16>(int col0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int sub = col0 >> 6, mc = col0 & 63;
    const int s = (((q >> 1) & 1) << 1) | ((g & 1) << 2);                 // tn_swz of every row this lane addresses
    return (sub * 64 + 8 * g + q) * kTileBytesK + ((((mc >> 3) + (p >> 1)) ^ s) << 4) + 8 * (p & 1);
}
template <> __device__ __forceinline__ int tn_frag_off<float>(int col0, int lane) {
    const int g = lane >> 4, mc = (col0 & 31) + (lane & 15);
    return ((col0 >> 5) * 32 + 4 * g) * kTileBytesK + ((mc >> 2) << 4) + ((mc & 3) << 2);   // row 4g of slab 0, chunk not yet swizzled
}
template <typename CT> __device__ __forceinline__ u32x4 tn_frag(const unsigned char* part, int off, int slab);
template <> __device__ __forceinline__ u32x4 tn_frag<__bf16>(const unsigned char* part, int off, int slab) {
    // Inline asm on purpose: through the builtin the compiler guards every transposed read with s_waitcnt vmcnt(0)
    // (it treats it as possibly aliasing the LDS-DMA refills in flight), which serialises the ring.  The price is that
    // the results are only valid after tn_frags_ready() below.
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(part + off + slab * (32 * kTileBytesK));   // K slab = 32 rows
    unsigned long long lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));                 // k = 8g .. 8g+3
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(hi) : "v"(a));      // k = 8g+4 .. 8g+7 (4 rows of 128 B further)
    return u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
}
// all transposed reads issued so far have landed; `f` are the fragments whose consumers must come after the wait
template <int N> __device__ __forceinline__ void tn_frags_ready(u32x4 (&f)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}
__device__ __forceinline__ void tn_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ u32x4 tn_frag<float>(const unsigned char* part, int off, int slab) {
    const int o = off + slab * (16 * kTileBytesK);                           // K slab = 16 rows
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)                                              // row 4g+e: odd rows sit 4 chunks (64 B) away
        v[e] = *reinterpret_cast<const unsigned*>(part + ((o + e * kTileBytesK) ^ ((e & 1) << 6)));
    return v;
}

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp): the epilogues are
// latency-critical and libm's expf/logf/division cost tens of instructions each.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ float flog(float x) { return __logf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float a) { return frcp(1.0f + fexp(-a)); }
// 2*sinh(h) without cancellation: series below |h| = 0.5, exponentials above
__device__ __forceinline__ float two_sinh(float h) {
    const float h2 = h * h;
    const float ser = 2.0f * h * (1.0f + h2 * (1.0f / 6.0f + h2 * (1.0f / 120.0f + h2 * (1.0f / 5040.0f + h2 * (1.0f / 362880.0f)))));
    const float e = fexp(h);
    return fabsf(h) < 0.5f ? ser : e - frcp(e);
}

__device__ __forceinline__ float act_fwd(int act, float a) {
    switch (act) {
        case AVAE_ACT_RELU: return fmaxf(a, 0.0f);
        case AVAE_ACT_SOFTPLUS: return fmaxf(a, 0.0f) + log1pf(fexp(-fabsf(a)));
        case AVAE_ACT_SIGMOID: return sigmoidf_(a);
        case AVAE_ACT_TANH: { const float e = fexp(-2.0f * fabsf(a)); const float t = (1.0f - e) * frcp(1.0f + e); return a < 0.0f ? -t : t; }
        default: return a;
    }
}
// Compile-time variants: the epilogue loops are instantiated per transfer function and selected by
// ONE wave-uniform switch outside the loop (a per-element switch kept every branch's libm code
// in the unrolled loop body and cost ~1.3 us per tile).
template <int ACT> __device__ __forceinline__ float act_fwd_t(float a) { return act_fwd(ACT, a); }
__device__ __forceinline__ float act_bwd(int act, float y);
template <int ACT> __device__ __forceinline__ float act_bwd_t(float y) { return act_bwd(ACT, y); }
#define AVAE_ACT_DISPATCH(act, CALL)                                   \
    switch (act) {                                                     \
        case AVAE_ACT_RELU: { constexpr int ACT = AVAE_ACT_RELU; CALL; } break;          \
        case AVAE_ACT_SOFTPLUS: { constexpr int ACT = AVAE_ACT_SOFTPLUS; CALL; } break;  \
        case AVAE_ACT_SIGMOID: { constexpr int ACT = AVAE_ACT_SIGMOID; CALL; } break;    \
        case AVAE_ACT_TANH: { constexpr int ACT = AVAE_ACT_TANH; CALL; } break;          \
        default: { constexpr int ACT = AVAE_ACT_IDENTITY; CALL; } break;                 \
    }

// derivative of the transfer function expressed through its OUTPUT y
__device__ __forceinline__ float act_bwd(int act, float y) {
    switch (act) {
        case AVAE_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case AVAE_ACT_SOFTPLUS: return 1.0f - fexp(-y);
        case AVAE_ACT_SIGMOID: return y * (1.0f - y);
        case AVAE_ACT_TANH: return 1.0f - y * y;
        default: return 1.0f;
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// global store of the wave (its release fence drains vmcnt), which in the epilogues puts a full
// store round trip (~1 us) in front of each barrier; the data exchanged here lives in LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sum over the NW*64 threads of the block, same value returned to every thread, fixed order.
template <int NW = 4> __device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    float t = (red[0] + red[1]) + (red[2] + red[3]);
    if constexpr (NW == 8) t += (red[4] + red[5]) + (red[6] + red[7]);
    return t;
}

// One parameter's TF-1 Adam step -- ONE definition (explicit fused multiply-adds, nothing left to context-dependent contraction) for
// k_adam and for the weight-gradient launch that carries the optimiser in its epilogue, so that both round alike.
__device__ __forceinline__ void adam_update(float g, float& m, float& v, float& th, float omb1, float omb2, float lr_t, float eps) {
#pragma clang fp contract(off)
    m = __builtin_fmaf(g - m, omb1, m);
    v = __builtin_fmaf(__builtin_fmaf(g, g, -v), omb2, v);
    th -= (m * lr_t) / (sqrtf(v) + eps);
}

// Reconstruction loss of one element and its gradient w.r.t. the logit -- ONE definition for the three kernels that carry the loss
// epilogue, with floating-point contraction off inside it, so that every route rounds alike (the routes are checked against each other
// bitwise).  sl = scale * loss, da = scale * dloss/da.
// Bernoulli: -(x log(1e-3+p) + (1-x) log(1e-3+1-p)), p = sigmoid(a)  (vae_assoc.py:321-324);  Gaussian: (x-a)^2 / 2  (:327-328)
__device__ __forceinline__ void loss_bernoulli(float a, float x, float sc, float& sl, float& da) {
#pragma clang fp contract(off)
    const float en = fexp(-a), p = frcp(1.0f + en);
    const float lp = 1e-3f + p, lq = 1e-3f + 1.0f - p;
    const float loss = -(x * flog(lp) + (1.0f - x) * flog(lq));
    sl = sc * loss;
    da = sc * p * (1.0f - p) * ((1.0f - x) * lp - x * lq) * frcp(lp * lq);
}
__device__ __forceinline__ void loss_gauss(float a, float x, float sc, float& sl, float& da) {
#pragma clang fp contract(off)
    const float df = a - x;
    sl = sc * 0.5f * df * df;
    da = sc * df;
}

// Index of element (r, c) of an fp32 accumulator tile in LDS (row stride LDC = BN + 4).  Columns are
// XOR-swizzled by the row's 8-row group in units of 4 floats: a row access (16-B groups) stays
// aligned and contiguous per group, while the column walk of the transposed store -- lanes 8 rows
// apart, 8*LDC = 0 mod 32 banks, i.e. all on ONE bank unswizzled -- spreads over 8 banks.
template <int LDC> __device__ __forceinline__ int cs_idx(int r, int c) { return r * LDC + (c ^ (((r >> 3) & 7) << 2)); }

// ------------------------------------------------------------------ epilogue passes
// The accumulator tile sits in LDS as fp32 [BM][BN+4].  A pass is written in three phases so that
// all of a thread's LDS reads and auxiliary global loads are in flight together (no per-quad
// load->wait->store chains): A) unconditional loads (addresses clamped into the allocation),
// B) the fused element maths, C) predicated stores.  Partial quads are stored element-wise so the
// zero padding and the constant-1 column are never touched.
template <typename OT, typename AT, bool HAS_AUX, bool WRITEBACK, int BM, int BN, int NT = kThreads, typename Op>
__device__ __forceinline__ void tile_pass(float* Cs, OT* out, int ld0, const AT* aux, int ldx,
                                          int M, int N, int m0, int n0, Op op) {
    constexpr int VW = Vec16<OT>::VW;                     // elements per 16-byte store
    constexpr int LDC = BN + 4, QC = BN / VW, NG = BM * QC;   // NG groups of VW elements in the tile
    constexpr int NQT = (NG + NT - 1) / NT;               // groups per thread (the smallest tile has fewer groups than threads)
    constexpr bool PARTIAL = NG % NT != 0;
    // groups handled together: <= 32 elements per array in flight -- 64 in the 8-wave kernel's passes with an auxiliary
    // operand (hidden-layer dgrad: act'(Y_prev)), whose load latency is then exposed once per tile instead of twice
    constexpr int MAXE = (HAS_AUX && NT >= 512) ? 64 : 32;
    constexpr int NQ = NQT < MAXE / VW ? NQT : MAXE / VW;
    static_assert(NQT % NQ == 0, "tile / thread mapping");
    const int tid = threadIdx.x;
    for (int q0 = 0; q0 < NQT; q0 += NQ) {
        float c[NQ][VW], a[NQ][VW];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = PARTIAL ? min(tid + (q0 + q) * NT, NG - 1) : tid + (q0 + q) * NT, row = idx / QC, c0 = (idx - row * QC) * VW;
#pragma unroll
            for (int h = 0; h < VW / 4; ++h) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(Cs + cs_idx<LDC>(row, c0 + 4 * h));
                c[q][4 * h] = t[0]; c[q][4 * h + 1] = t[1]; c[q][4 * h + 2] = t[2]; c[q][4 * h + 3] = t[3];
            }
            if (HAS_AUX) {
                const int gr = min(m0 + row, M - 1), gc = min(n0 + c0, ldx - VW);   // clamped => that group is never stored
                load_vec<AT, VW>(aux + (size_t)gr * ldx + gc, a[q]);
            } else {
#pragma unroll
                for (int e = 0; e < VW; ++e) a[q][e] = 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int raw = tid + (q0 + q) * NT;
            const bool mine = !PARTIAL || raw < NG;       // threads beyond the tile's last group compute on a clamped copy, store nothing
            const int idx = PARTIAL ? min(raw, NG - 1) : raw, row = idx / QC, c0 = (idx - row * QC) * VW;
            const bool rok = mine && m0 + row < M;
#pragma unroll
            for (int e = 0; e < VW; ++e) c[q][e] = op(c[q][e], a[q][e], rok && (n0 + c0 + e < N));
            if (WRITEBACK && mine) {
#pragma unroll
                for (int h = 0; h < VW / 4; ++h)
                    *reinterpret_cast<f32x4*>(Cs + cs_idx<LDC>(row, c0 + 4 * h)) = f32x4{c[q][4 * h], c[q][4 * h + 1], c[q][4 * h + 2], c[q][4 * h + 3]};
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int raw = tid + (q0 + q) * NT;
            const int idx = PARTIAL ? min(raw, NG - 1) : raw, row = idx / QC, c0 = (idx - row * QC) * VW;
            const int grow = m0 + row, gcol = n0 + c0;
            if ((!PARTIAL || raw < NG) && grow < M && gcol < N) store_vec<OT, VW>(out + (size_t)grow * ld0 + gcol, c[q], N - gcol);
        }
    }
}

// ------------------------------------------------------------------ non-GEMM work items
// KL(q||N(0,I)) (vae_assoc.py:335-337) and the symmetric-KL association penalty (:346-366) with
// their gradients w.r.t. (mu, lv).  The log-determinant terms of the two directed KLs cancel, so
// per sample and dimension  S = 1/2 [ e^a + e^-a - 2 + D^2 (e^-lvi + e^-lvj) ],  a = lvi-lvj,
// D = mui-muj;  e^a + e^-a - 2 is evaluated as (2 sinh(a/2))^2 to avoid cancellation.
template <int NW = 4> __device__ __forceinline__ void latent_item(const WorkItem& w, int t, float* red) {
    constexpr int kThreads = NW * 64;        // (shadows avae::kThreads: the 8-wave kernels run this item with 512 threads)
    const float* mulv[kMaxMod] = {reinterpret_cast<const float*>(w.A), reinterpret_cast<const float*>(w.B),
                                  reinterpret_cast<const float*>(w.aux0), reinterpret_cast<const float*>(w.aux1)};
    float* g0[kMaxMod] = {reinterpret_cast<float*>(w.out0), reinterpret_cast<float*>(w.out1),
                          reinterpret_cast<float*>(w.out2), reinterpret_cast<float*>(const_cast<void*>(w.aux2))};
    const int nz = w.nz, nz2 = 2 * w.nz, M = w.M;
    float csum = 0.0f;
    for (int idx = threadIdx.x; idx < kLatentRows * nz; idx += kThreads) {
        const int row = idx / nz, d = idx - row * nz;
        const int grow = t * kLatentRows + row;
        if (grow >= M) continue;
        float mu[kMaxMod], lv[kMaxMod], gmu[kMaxMod], glv[kMaxMod], en[kMaxMod];
        const float eps_v = w.eps[(size_t)grow * w.ldx + d];
#pragma unroll
        for (int m = 0; m < kMaxMod; ++m) {
            mu[m] = lv[m] = gmu[m] = glv[m] = 0.0f; en[m] = 1.0f;
            if (m < w.n_mod) {
                mu[m] = mulv[m][(size_t)grow * nz2 + d];
                lv[m] = mulv[m][(size_t)grow * nz2 + nz + d];
            }
        }
#pragma unroll
        for (int m = 0; m < kMaxMod; ++m) {
            if (m < w.n_mod) {
                const float el = fexp(lv[m]);
                en[m] = frcp(el);                                   // e^-lv
                const float s = w.wts[m] * w.inv_bg;
                csum += s * (-0.5f * (1.0f + lv[m] - mu[m] * mu[m] - el));
                gmu[m] = s * mu[m];
                glv[m] = 0.5f * s * (el - 1.0f);
            }
        }
#pragma unroll
        for (int i = 0; i < kMaxMod; ++i) {
#pragma unroll
            for (int j = i + 1; j < kMaxMod; ++j) {
                if (j < w.n_mod) {
                    const float a = lv[i] - lv[j], dl = mu[i] - mu[j];
                    const float sh = two_sinh(0.5f * a);             // e^(a/2) - e^(-a/2)
                    const float dsh = two_sinh(a);                   // e^a - e^-a
                    csum += w.lambda * 0.5f * (sh * sh + dl * dl * (en[i] + en[j]));
                    const float gm = w.lambda * dl * (en[i] + en[j]);
                    gmu[i] += gm; gmu[j] -= gm;
                    glv[i] += 0.5f * w.lambda * (dsh - dl * dl * en[i]);
                    glv[j] += 0.5f * w.lambda * (-dsh - dl * dl * en[j]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < kMaxMod; ++m) {
            if (m < w.n_mod) {
                g0[m][(size_t)grow * 3 * w.ldx + d] = gmu[m];                 // row = [g0mu | g0lv | F], each ldx = roundup(n_z, 4) wide
                g0[m][(size_t)grow * 3 * w.ldx + w.ldx + d] = glv[m];
                // reparameterisation factor for the backward pass: d z / d lv = 1/2 exp(lv/2) eps
                g0[m][(size_t)grow * 3 * w.ldx + 2 * w.ldx + d] = 0.5f * eps_v * fexp(0.5f * lv[m]);
            }
        }
    }
    const float total = block_sum<NW>(csum, red);
    if (threadIdx.x == 0) w.partial[w.slot_base + t] = total;
}

template <int NW = 4> __device__ __forceinline__ void cost_item(const WorkItem& w, DevState* st, float* red) {
    constexpr int kThreads = NW * 64;
    float s = 0.0f;
    for (int i = threadIdx.x; i < w.n_slots; i += kThreads) s += w.partial[i];
    const float total = block_sum<NW>(s, red);
    if (threadIdx.x == 0) {
        reinterpret_cast<float*>(w.out0)[0] = total;
        if (w.bump_step) {
            const long long tnew = st->step + 1;
            st->step = tnew;
            // TF-1 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), published for the Adam kernel of this step
            const double b1t = pow((double)w.lambda, (double)tnew), b2t = pow((double)w.inv_bg, (double)tnew);
            st->lr_t = (float)((double)w.scale * sqrt(1.0 - b2t) / (1.0 - b1t));
            st->last_cost = total;                                   // local cost; the multi-replica path
            st->cost_hist[(tnew - 1) % kCostHist] = total;           // overwrites it with the all-reduced one
        }
    }
}

// ------------------------------------------------------------------ register epilogue of the 8-wave NT tiles
// The 8-wave kernels multiply with the MFMA operands SWAPPED (A := the weight fragment, B := the activation fragment), so an
// accumulator holds the transposed 16x16 block: lane (q = lane >> 4, c = lane & 15) has output ROW c and the four consecutive
// COLUMNS 4q .. 4q+3 of the block in its four registers -- row-contiguous data per lane, no trip through LDS.  fp32 results are
// stored as they stand (16 bytes per lane and block).  bf16 results are packed to 8 bytes and one v_permlane16_swap per dword
// (odd 16-lane rows of the first operand <-> even rows of the second) turns two neighbouring blocks (j, j+1) into 8 consecutive
// columns per lane:  q=0: block j cols 0-7 | q=1: block j+1 cols 0-7 | q=2: block j cols 8-15 | q=3: block j+1 cols 8-15,
// i.e. ONE 16-byte store per lane and block pair, 64 contiguous bytes per output row and instruction.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void store16_wt(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
    const u32x4 raw = {a, b, c, d};      // written through (sc0 sc1), as store_vec<__bf16, 8>: see there
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(raw) : "memory");   // (s_nop 1: the store must have read its data registers before hipcc's next instruction may write them)
}
// Applies f(i, j, r, acc) to every accumulator element and stores the MI x NI blocks of this wave.  rbase = first row of the
// wave's sub-tile + (lane & 15); cwave = first column of the wave's sub-tile.  EXEC must be full (the swaps cross lanes).
template <typename CT, int MI, int NI, typename F>
__device__ __forceinline__ void regep_store(const f32x4 (&acc)[MI][NI], CT* out, int ld0, int M, int N, int rbase, int cwave, int lane, F f) {
    const int q = lane >> 4;
    if constexpr (sizeof(CT) == 2) {
        static_assert(NI % 2 == 0, "bf16 register epilogue pairs neighbouring column blocks");
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = rbase + 16 * i;
#pragma unroll
            for (int jp = 0; jp < NI / 2; ++jp) {
                float va[4], vb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { va[r] = f(i, 2 * jp, r, acc[i][2 * jp][r]); vb[r] = f(i, 2 * jp + 1, r, acc[i][2 * jp + 1][r]); }
                unsigned a0 = pack_bf16(va[0], va[1]), a1 = pack_bf16(va[2], va[3]);
                unsigned b0 = pack_bf16(vb[0], vb[1]), b1 = pack_bf16(vb[2], vb[3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                a0 = s0[0]; b0 = s0[1]; a1 = s1[0]; b1 = s1[1];
                const int col = cwave + 16 * (2 * jp + (q & 1)) + 8 * (q >> 1);
                if (row < M && col < N) {
                    CT* p = out + (size_t)row * ld0 + col;
                    if (col + 8 <= N) store16_wt(p, a0, a1, b0, b1);
                    else {               // the row's last, partial group: element-wise, the padding (and the constant-1 column) stays untouched
                        const unsigned wv[4] = {a0, a1, b0, b1};
#pragma unroll
                        for (int e = 0; e < 7; ++e)
                            if (col + e < N) reinterpret_cast<unsigned short*>(p)[e] = (unsigned short)(wv[e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = rbase + 16 * i;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = f(i, j, r, acc[i][j][r]);
                const int col = cwave + 16 * j + 4 * q;
                if (row < M && col < N) store_row<float>(reinterpret_cast<float*>(out) + (size_t)row * ld0 + col, v, N - col);
            }
        }
    }
}
// 4 consecutive elements of the compute type at p -> floats (8-byte load for bf16, 16-byte for fp32)
template <typename CT> struct Quad;
template <> struct Quad<__bf16> { typedef uint2 raw; };
template <> struct Quad<float> { typedef f32x4 raw; };
template <typename CT> __device__ __forceinline__ float quad_elem(const typename Quad<CT>::raw& t, int r);
template <> __device__ __forceinline__ float quad_elem<__bf16>(const uint2& t, int r) {
    const unsigned w = r < 2 ? t.x : t.y;
    return bf16_bits_to_float((r & 1) ? (w >> 16) : (w & 0xffffu));
}
template <> __device__ __forceinline__ float quad_elem<float>(const f32x4& t, int r) { return t[r]; }

// ------------------------------------------------------------------ the grouped kernel
// LDS budget of one workgroup: a ring of RING linear K-tile stages, re-used by the epilogue as one
// (or, for the head / latent kinds, two) fp32 accumulator tiles.  Passed as dynamic LDS so that a
// launch only pays for what its kinds need: 64x64 tiles keep 4 stages (up to 3 K tiles in flight,
// the small-batch case is latency-bound); 128x128 tiles keep 2 stages = 66 KiB so that TWO
// workgroups share a CU and one's MFMA phase covers the other's barrier / DMA wait.
template <int BM, int BN, int RING> struct TileSmem {
    static constexpr int kStage = (BM + BN) * kTileBytesK;     // linear 128-B rows (LDS-DMA image), XOR-swizzled chunks
    static constexpr int kStages = RING * kStage;
    static constexpr int kC = BM * (BN + 4) * 4;               // one fp32 accumulator tile
};
// tile_cfg: 0 = 64x64 tile, 4-stage ring; 1 = 128x128 tile, 2-stage ring, two workgroups per CU;
//           2 = 256x128 tile, 3-stage ring, one workgroup per CU (big single-C-tile kinds only)
//           3 = 32x64 tile, 4-stage ring (NT launches with too few 64x64 tiles to occupy the chip; a 6-stage ring is slower:
//               issuing five tiles up front costs more than the latency it hides, 69.1 vs 67.1 us/step on C2)
//           4 = 64x128 tile, 4-stage ring (wide-latent head / latent-dgrad launches with few tiles: three K tiles in
//               flight instead of one, twice the workgroups of cfg 1)
//           5 = 32x32 tile, 4-stage ring (as 3, for launches without head kinds: a wave issues 2 refill pieces per K tile
//               instead of 3, and the issue cost of those pieces is what paces the K loop of a lone workgroup)
//           6 = 256x64 tile, 8 waves, 3-stage ring: the output + reconstruction-loss launch of the big nets (its epilogue is
//               bound by ~5 transcendentals per element: 256x64 tiles of a 784 + 147-column output give one tile per CU)
int tile_lds_bytes(int tile_cfg, bool two_c_tiles) {
    if (tile_cfg == 6) return TileSmem<256, 64, 3>::kStages + 64;
    if (tile_cfg == 5) return TileSmem<32, 32, 4>::kStages + 64;
    if (tile_cfg == 4) {
        const int st = TileSmem<64, 128, 4>::kStages, c4 = TileSmem<64, 128, 4>::kC * (two_c_tiles ? 2 : 1);
        return (st > c4 ? st : c4) + 64;
    }
    const int stages = tile_cfg == 3 ? TileSmem<32, 64, 4>::kStages : tile_cfg == 2 ? TileSmem<256, 128, 3>::kStages : tile_cfg == 1 ? TileSmem<128, 128, 2>::kStages : TileSmem<64, 64, 4>::kStages;
    const int c = (tile_cfg == 3 ? TileSmem<32, 64, 4>::kC : tile_cfg == 2 ? TileSmem<256, 128, 3>::kC : tile_cfg == 1 ? TileSmem<128, 128, 2>::kC : TileSmem<64, 64, 4>::kC) * (two_c_tiles ? 2 : 1);
    return (stages > c ? stages : c) + 64;                     // + block-reduction scratch
}

extern __shared__ __attribute__((aligned(16))) unsigned char avae_dyn_smem[];

template <bool TN> struct ArgsOf { typedef LaunchArgs type; };
template <> struct ArgsOf<true> { typedef TnLaunchArgs type; };

// ---- implicit patch matrix (ConvA, avae_device.h): address of the 16-byte chunk P[m][kk ..] a lane stages
__device__ __forceinline__ unsigned mg_div(unsigned n, unsigned mg) { return mg ? __umulhi(n, mg) : n; }     // mg = 0: divisor 1
struct ConvRow { const unsigned char* base; int hs, ws; bool ok; };     // row m = (b, oh, ow): image base, oh*so - pad, ow*so - pad
struct ConvTap { int kh, kw, ci, kind; };                                // column kk = (kh, kw, ci); kind 1: the ones chunk, 2: zeros
template <int ES> __device__ __forceinline__ ConvRow conv_row(const ConvA& c, int m) {
    ConvRow r;
    r.ok = m < c.M;
    const unsigned mm = r.ok ? (unsigned)m : 0u;
    const unsigned b = mg_div(mm, c.mg_ohw), rem = mm - b * (unsigned)c.OHW, oh = mg_div(rem, c.mg_ow), ow = rem - oh * (unsigned)c.OW;
    r.base = reinterpret_cast<const unsigned char*>(c.src) + (size_t)b * c.src_sb * ES;
    r.hs = (int)oh * c.so - c.pad; r.ws = (int)ow * c.so - c.pad;
    return r;
}
__device__ __forceinline__ ConvTap conv_tap(const ConvA& c, int kk) {
    ConvTap t;
    t.kind = kk < c.K ? 0 : (kk == c.K && c.ones ? 1 : 2);
    const unsigned k2 = t.kind ? 0u : (unsigned)kk;
    const unsigned kp = mg_div(k2, c.mg_cin), kh = mg_div(kp, c.mg_k);
    t.ci = (int)(k2 - kp * (unsigned)c.Cin); t.kh = (int)kh; t.kw = (int)(kp - kh * (unsigned)c.k);
    return t;
}
template <int ES> __device__ __forceinline__ const unsigned char* conv_addr(const ConvA& c, const ConvRow& r, const ConvTap& t) {
    const unsigned char* z = reinterpret_cast<const unsigned char*>(c.consts);
    if (t.kind) return z + (t.kind == 1 && r.ok ? 16 : 0);
    const int nh = r.hs + t.kh, nw = r.ws + t.kw;
    bool ok = r.ok && nh >= 0 && nw >= 0;
    int ih = nh, iw = nw;
    if (c.d == 2) { ok = ok && !((nh | nw) & 1); ih = nh >> 1; iw = nw >> 1; }     // (the branch uses strides 1 and 2 only; the host checks)
    ok = ok && ih < c.IH && iw < c.IW;
    return ok ? r.base + ((size_t)(ih * c.IW + iw) * c.src_sp + t.ci) * ES : z;
}

__device__ __forceinline__ WorkItem item_of(const LaunchArgs& args, int y) { return args.items[y]; }
__device__ __forceinline__ WorkItem item_of(const TnLaunchArgs& args, int y) {
    const TnItem ti = args.items[y];
    WorkItem w = {};
    w.kind = K_WGRAD;
    w.M = ti.M; w.N = ti.N; w.K = ti.K; w.lda = ti.lda; w.ldb = ti.ldb; w.ld0 = ti.ld0;
    w.tiles_m = ti.tiles_m; w.tiles_n = ti.tiles_n; w.ksplit = ti.ksplit; w.kchunk = ti.kchunk; w.bias_row = ti.bias_row;
    w.tile_off = ti.tile_off; w.tile_cnt = ti.tile_cnt; w.conv = ti.conv;
    w.A = ti.A; w.B = ti.B; w.out0 = ti.out; w.out1 = ti.out;
    return w;
}

// NP > 0: NP extra "producer" waves (wave index >= NW) issue every operand refill (LDS-DMA) of the workgroup, the NW "consumer"
// waves only read fragments and multiply.  Why (tools/loopstamps.py, tools/fill_probe2.hip): a 48-KB tile is 48 wave-instructions
// of 1 KiB and the CU's texture-address path takes them at 64 B/clk -- ~16 clocks each, 770 clocks per tile -- with the issuing
// wave stalled meanwhile; issued by the multiplying waves those stalls (420 clocks per wave and tile) sat on the critical path
// of every tile whichever way the waves were ordered.  Producers stall alone.
template <typename CT, int BM, int BN, int RING, int NW = 4, bool TN = false, int NP = 0>
__global__ void __launch_bounds__((NW + NP) * 64, (RING == 2 ? 2 : 1)) k_grouped(const typename ArgsOf<TN>::type args, DevState* st, int lds_bytes,
                                                      unsigned long long* stamps, int launch_id) {
    unsigned char* smem = avae_dyn_smem;
    float* red = reinterpret_cast<float*>(smem + lds_bytes - 64);
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#define AVAE_STAMP(i) { __builtin_amdgcn_sched_barrier(0); sv[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
#define AVAE_STAMP_FLUSH() { const unsigned lin_ = blockIdx.y * gridDim.x + blockIdx.x;                                        \
    if (stamps && threadIdx.x == 0 && launch_id < kStampLaunches && lin_ < (unsigned)kStampBlocks) {                            \
        for (int i_ = 0; i_ < kStampWords; ++i_) stamps[((size_t)launch_id * kStampBlocks + lin_) * kStampWords + i_] = sv[i_]; } }
    AVAE_STAMP(0)
#else
#define AVAE_STAMP(i)
#define AVAE_STAMP_FLUSH()
#endif
#ifdef AVAE_STAMPS_PRO       /* words 6 and 7 belong to the prologue stamps in that build */
#define AVAE_STAMP_EP(i)
#else
#define AVAE_STAMP_EP(i) AVAE_STAMP(i)
#endif
#if defined(AVAE_STAMPS) && defined(AVAE_LOOPSTAMPS)
    // diagnostic: shader-clock cycles per section of the K loop, summed over the K tiles of this workgroup (wave 0)
    unsigned long long la[6] = {0, 0, 0, 0, 0, 0}, lt = 0;
#define AVAE_LT0() { __builtin_amdgcn_sched_barrier(0); lt = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#define AVAE_LT(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); la[i] += n_ - lt; lt = n_; __builtin_amdgcn_sched_barrier(0); }
#define AVAE_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define AVAE_LT0()
#define AVAE_LT(i)
#define AVAE_LGKM0()
#endif

    // Which item, and which of its S contiguous tile-list chunks, this workgroup serves.  Hardware deals workgroups
    // round-robin over the 8 XCDs (private 4 MiB L2 each) in linear order y*grid_x + x, and grid_x is a multiple of 8, so
    // x%8 names the XCD.  Default (G = 1, S = 8): item y is spread over all 8 XCDs, XCD c taking chunk c of its tile list
    // (tiles that share A row panels / B column panels meet in one L2; heavy and light items stay balanced).
    // Big weight gradients (G > 1): the G items y0..y0+G-1 share the 8 XCDs instead, item y0+i owning XCDs [i*S, (i+1)*S)
    // with S = 8/G -- the 32 tiles resident on an XCD then come from one or two items rather than eight, and stream
    // a third to a half of the operand panels through its L2 (C4: the launch is bound by L2 misses, not by tile count).
    int item_y = blockIdx.y, part = blockIdx.x & 7, idx = blockIdx.x >> 3, S = 8;
    if constexpr (TN) {
        const int G = args.xcd_group;
        if (G > 1) {
            S = 8 / G;
            item_y = (blockIdx.y / G) * G + part / S;
            idx += (blockIdx.y % G) * (gridDim.x >> 3);
            part = part % S;
            if (item_y >= args.n_items) return;                       // last group of a launch whose item count G does not divide
        }
    }
    const WorkItem w = item_of(args, item_y);                     // one burst of scalar loads from the kernarg segment
    // ... and it has to BE one burst: left alone, the compiler fetches the few fields the tile-index branch below needs,
    // waits, and only then fetches the operand pointers -- two kernel-argument round trips (~0.25 us each) before the
    // first load of every workgroup.  Naming the fields here puts all their s_loads ahead of the first wait.
    if constexpr (TN) asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out0), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.kchunk), "s"(lds_bytes));
    else asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out0), "s"(w.aux0), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.ldx), "s"(lds_bytes));
#ifdef AVAE_STAMPS_PRO       /* diagnostic: where the time before the first load goes (6: item descriptor here, 7: addresses set up) */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    AVAE_STAMP(6)
#endif
    int t;                       // bijective for any tile count; only speed depends on the order
    {
        const int nt = TN ? w.tile_cnt : w.tiles_m * w.tiles_n;
        const int q = nt / S, r = nt - q * S;
        if (idx >= q + (part < r ? 1 : 0)) return;                // padding block behind the chunk's last tile
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx + (TN ? w.tile_off : 0);
    }
    int ks = 0;                  // split-K chunk of this workgroup (weight gradients with a long K only)
    if constexpr (TN) {
        if (w.ksplit > 1) { const int per = w.tiles_m * w.tiles_n; ks = t / per; t -= ks * per; }
    }
    constexpr int NT = NW * 64;
    // The two non-GEMM kinds are handled at the END of the kernel: their bodies are large (cost_item drags in a
    // double-precision pow) and, sitting here, they would separate the prologue from the code every GEMM workgroup
    // runs before it can issue its first load -- an instruction-cache miss on the critical path of every launch.
    // 8-wave NT tiles keep their results in registers through the epilogue (REGEP, below).  Of them the 256x128 tile carries the
    // plain kinds (hidden-layer forward / dgrad), the 256x64 tile the output + loss kind and the latent / cost items beside it.
    constexpr bool REGEP = NW == 8 && !TN;
    constexpr bool LOSS_TILE = REGEP && BN == 64;
    bool non_gemm = false;
    if constexpr (!TN && (NW == 4 || LOSS_TILE)) non_gemm = w.kind == K_LATENT || w.kind == K_COST;
    if constexpr (NP > 0) { if (non_gemm && threadIdx.x >= NW * 64) return; }      // the non-GEMM items are run by the consumer waves
    if (!non_gemm) {

    constexpr int WM = BM / (NW / 2), WN = BN / 2;   // per-wave sub-tile: waves are arranged (NW/2) x 2
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int LDC = BN + 4;
    constexpr int ES = (int)sizeof(CT);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // Tail product (WorkItem::tail_*): the item's tiles_n counts tail SLICES, not column tiles -- every slice's workgroup computes
    // the (single) 32x64 head tile of its row block and then its own 64 columns of the consuming layer.
    constexpr bool TAIL = !TN && NW == 4 && BM == 32 && BN == 64;
    bool tail_on = false;
    if constexpr (TAIL) tail_on = w.tail_mode != 0;
    const int tm = t / w.tiles_n, tn_raw = t - tm * w.tiles_n;
    const int tn = tail_on ? 0 : tn_raw, ts = tail_on ? tn_raw : 0;
    const int m0 = tm * BM, n0 = tn * BN;

    // NT: the tile's rows are rows of A / B (K contiguous).  TN: its rows are K (batch samples), its columns m0.. / n0..
    const unsigned char* Ag = reinterpret_cast<const unsigned char*>(w.A) + (TN ? (size_t)m0 * ES : (size_t)m0 * w.lda * ES);
    const unsigned char* Bg = reinterpret_cast<const unsigned char*>(w.B) + (TN ? (size_t)n0 * ES : (size_t)n0 * w.ldb * ES);
    const size_t lda_b = (size_t)w.lda * ES, ldb_b = (size_t)w.ldb * ES;
    constexpr int EPR = kTileBytesK / ES;          // elements per 128-byte image row = K extent of one stage
    int nk = (w.K * ES) / kTileBytesK;
    int k_first = 0;             // first K tile of this workgroup
    if constexpr (TN) {
        if (w.ksplit > 1) { k_first = ks * w.kchunk; nk = min(w.kchunk, nk - k_first); }
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K loop: LDS-DMA ring.  Each wave-instruction global_load_lds_dwordx4 moves 8 rows x 128 B
    // (1 KiB) straight from HBM/L2 into LDS (no staging registers); kRing stages, up to kRing-1 K
    // tiles in flight, one barrier per tile.  The LDS image is linear (a DMA cannot scatter), so the
    // bank-conflict swizzle is applied on the per-lane SOURCE address and again on the ds_read:
    // 16-B chunk c of tile row r lives at chunk c ^ ((r >> 1) & 7)  (conflict-free ds_read_b128).
    // Ordering: a tile may be read after (this wave's counted vmcnt) + (a barrier every wave passed);
    // a stage is re-filled only after the barrier that follows its last read.
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    constexpr int R8 = (BM + BN) / 8;          // wave-instructions per tile
    constexpr int NDW = NP > 0 ? NP : NW;      // waves that issue them
    constexpr int NCH = R8 / NDW;              // per issuing wave
    static_assert(R8 % NDW == 0, "refill pieces divide evenly over the issuing waves");
    const int fr = lane & 15, fq = lane >> 4;
    const int dwave = NP > 0 ? (wave >= NW ? wave - NW : 0) : wave;      // index among the issuing waves
    const unsigned char* src[NCH];
    size_t kadv[NCH];                                                    // bytes from one K tile to the next (TN: EPR rows)
    // Implicit patch matrix as the A operand (ConvA): on the small tiles only (conv stages have <= 64 output channels or are
    // forced onto these tiles by the host).  The A pieces of an issuing wave are its first NA pieces.
    constexpr bool IMPL = BM <= 64 && NP == 0 && (BM / 8) % NDW == 0;
    constexpr int NA = IMPL ? (BM / 8) / NDW : 1;
    bool impl = false;
    ConvA cv = {};
    ConvRow crow[NA];                                                    // NT: the piece's row (fixed over K)
    ConvTap ctap[NA];                                                    // TN: the piece's column chunk (fixed over K)
    int cpos[NA];                                                        // NT: element offset of the chunk inside a K tile; TN: row inside a K tile
    if constexpr (IMPL) {
        if (w.conv > 0) { impl = true; cv = args.conv_tab[w.conv - 1]; }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int r = (c * NDW + dwave) * 8 + (lane >> 3);              // row of the (A part, then B part) tile image
        if constexpr (TN) {
            static_assert(BM % EPR == 0 && BN % EPR == 0, "K-major images are made of EPR x EPR sub-images");
            const bool is_a = r < BM;
            const int rr = is_a ? r : r - BM, sub = rr / EPR, k = rr - sub * EPR;
            const int lc = (lane & 7) ^ tn_swz<CT>(k);                  // logical chunk this lane fetches
            const size_t ld_b = is_a ? lda_b : ldb_b;
            kadv[c] = (size_t)EPR * ld_b;
            src[c] = (is_a ? Ag : Bg) + (size_t)k * ld_b + (size_t)sub * kTileBytesK + lc * 16 + (size_t)k_first * kadv[c];
            if constexpr (IMPL) { if (impl && c < NA) { ctap[c] = conv_tap(cv, m0 + sub * EPR + lc * (16 / ES)); cpos[c] = k_first * EPR + k; } }
        } else {
            const int lc = (lane & 7) ^ ((r >> 1) & 7);                 // logical chunk this lane fetches
            src[c] = (r < BM ? Ag + (size_t)r * lda_b : Bg + (size_t)(r - BM) * ldb_b) + lc * 16;
            kadv[c] = kTileBytesK;
            if constexpr (IMPL) { if (impl && c < NA) { crow[c] = conv_row<ES>(cv, m0 + r); cpos[c] = lc * (16 / ES); } }
        }
    }
#ifdef AVAE_STAMPS_PRO
    asm volatile("" :: "v"(src[0]));
    AVAE_STAMP(7)
#endif
    const int sw0 = (fq ^ (fr >> 1)) * 16;                              // swizzled chunk of K-slab 0; slab 1 = ^64
    const int aoff = (wr * WM + fr) * kTileBytesK;
    const int boff = (BM + wc * WN + fr) * kTileBytesK;
    int offA[MI], offB[NI];                                              // TN: first-read offsets of the wave's fragments
#pragma unroll
    for (int i = 0; i < MI; ++i) offA[i] = TN ? tn_frag_off<CT>(wr * WM + i * 16, lane) : 0;
#pragma unroll
    for (int j = 0; j < NI; ++j) offB[j] = TN ? BM * kTileBytesK + tn_frag_off<CT>(wc * WN + j * 16, lane) : 0;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int dwave_u = __builtin_amdgcn_readfirstlane(dwave);
    const bool producer = NP > 0 && wave_u >= NW;
    // weight gradients whose bias row would need a tile row of its own (in = k * BM): the first tile row's top waves add
    // it from the B fragments they hold anyway -- one MFMA per fragment against an all-ones A operand (every row of the
    // 16x16 result is the column sum) instead of a fifth row of tiles that multiplies 255 rows of padding.
    f32x4 accb[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool do_bias = false;
    if constexpr (TN) do_bias = w.bias_row > 0 && tm == 0 && (wave_u >> 1) == 0;
    const unsigned one_bits = sizeof(CT) == 2 ? 0x3F803F80u : 0x3F800000u;
    const u32x4 ones = {one_bits, one_bits, one_bits, one_bits};

#define AVAE_DMA(kt, buf)                                                                              \
    {                                                                                                  \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                              \
            const unsigned char* sp_ = src[c] + (size_t)(kt) * (TN ? kadv[c] : (size_t)kTileBytesK);   \
            if constexpr (IMPL) { if (impl && c < NA) {     /* implicit patch matrix: this lane's chunk of K tile kt */ \
                if constexpr (TN) sp_ = conv_addr<ES>(cv, conv_row<ES>(cv, cpos[c] + (kt) * EPR), ctap[c]); \
                else sp_ = conv_addr<ES>(cv, crow[c], conv_tap(cv, (kt) * EPR + cpos[c]));              \
            } }                                                                                        \
            __builtin_amdgcn_global_load_lds((gp_t)sp_,                                                \
                (lp_t)(smem + (buf) * TileSmem<BM, BN, RING>::kStage + (c * NDW + dwave_u) * 1024), 16, 0, 0); \
        }                                                                                              \
    }
#define AVAE_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#ifdef AVAE_ABL_NO_DMA      /* diagnostic: loop without the operand refills (results are garbage) */
#define AVAE_ABL_DMA(kt, buf)
#else
#define AVAE_ABL_DMA(kt, buf) AVAE_DMA(kt, buf)
#endif
    // The refill DMA of the stage freed by the previous tile is issued between the first slab's
    // fragment reads and its MFMAs, so the issue cost (~60-100 cycles per 1-KiB piece) overlaps
    // matrix-pipe work instead of preceding it.  With 8 waves (two per SIMD: waves i and i+4) one barrier
    // per tile keeps all waves in step, so both waves of a SIMD would hit their DMA burst -- where the wave
    // stalls on the CU's vector-memory path and the matrix pipe idles -- at the same time; waves 4-7 therefore
    // issue their refill AFTER their MFMAs: one wave of each SIMD multiplies while the other issues.
    // Revised (in-loop cycle stamps, tools/loopstamps.py): with waves 4-7 late and waves 0-3 issuing between their reads, a
    // tile took ~2100 cycles: reads 260 -> [wave 0: DMA issue 420] -> both waves' MFMAs 1024 -> [wave 4: DMA issue 420] -- the
    // late half's issue ran with the matrix pipe idle.  Now the two halves are complementary: waves 0-3 read, multiply, then
    // issue (late); waves 4-7 issue FIRST (the stage freed by the previous tile is free from the barrier on), then read and
    // multiply -- each half's DMA issue runs under the other half's MFMAs.  args.sched (AVAE_SCHED) selects the old order for A/B.
    const int sched = args.sched;
    const bool dma_late = NW == 8 && (sched == 1 ? wave_u >= 4 : wave_u < 4);
    const bool dma_first = NW == 8 && sched != 1 && wave_u >= 4;
    /* fragment f of K-slab `slab`: NT = one ds_read_b128 of the row image, TN = transposed reads of the K-major image */
#define AVAE_FRAG(nt_off, tn_offs, f, slab)                                                            \
    (TN ? tn_frag<CT>(Sb, (tn_offs)[f], slab)                                                          \
        : *reinterpret_cast<const u32x4*>(Sb + (nt_off) + (f) * 16 * kTileBytesK + ((slab) ? (sw0 ^ 64) : sw0)))
#define AVAE_COMPUTE(buf, do_dma, dma_kt, dma_buf)                                                     \
    {                                                                                                  \
        const unsigned char* Sb = smem + (buf) * TileSmem<BM, BN, RING>::kStage;                       \
        if ((do_dma) && dma_first) AVAE_ABL_DMA(dma_kt, dma_buf)                                       \
        u32x4 a0[MI], b0[NI], a1[MI], b1[NI];      /* both K slabs' fragments: the second slab's LDS  */ \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)   /* latency hides behind the first slab's MFMAs */ \
            a0[i] = AVAE_FRAG(aoff, offA, i, 0);                                                       \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                 \
            b0[j] = AVAE_FRAG(boff, offB, j, 0);                                                       \
        AVAE_LGKM0(); AVAE_LT(2)                                                                       \
        if ((do_dma) && !dma_late && !dma_first) AVAE_ABL_DMA(dma_kt, dma_buf)                         \
        AVAE_LT(3)                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
            a1[i] = AVAE_FRAG(aoff, offA, i, 1);                                                       \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                 \
            b1[j] = AVAE_FRAG(boff, offB, j, 1);                                                       \
        if constexpr (TN && sizeof(CT) == 2) { tn_wait_lds(); tn_frags_ready(a0); tn_frags_ready(b0); tn_frags_ready(a1); tn_frags_ready(b1); } \
        AVAE_LGKM0(); AVAE_LT(4)                                                                       \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
            _Pragma("unroll") for (int j = 0; j < NI; ++j) mma_sel<CT, REGEP>(a0[i], b0[j], acc[i][j]); \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                 \
            _Pragma("unroll") for (int j = 0; j < NI; ++j) mma_sel<CT, REGEP>(a1[i], b1[j], acc[i][j]); \
        if constexpr (TN) if (do_bias) {                                                               \
            _Pragma("unroll") for (int j = 0; j < NI; ++j) { mma<CT>(ones, b0[j], accb[j]); mma<CT>(ones, b1[j], accb[j]); } \
        }                                                                                              \
        if ((do_dma) && dma_late) AVAE_ABL_DMA(dma_kt, dma_buf)                                        \
        AVAE_MFMA_DRAIN()                                                                              \
        AVAE_LT(5)                                                                                     \
    }

#if defined(AVAE_STAMPS) && defined(AVAE_LOOPSTAMPS)
#define AVAE_MFMA_DRAIN() { float d_; asm volatile("v_mov_b32 %0, %1" : "=v"(d_) : "v"(acc[MI - 1][NI - 1][3])); asm volatile("" :: "v"(d_)); }
#else
#define AVAE_MFMA_DRAIN()
#endif
    AVAE_STAMP(1)
    // Register epilogue: the operand the epilogue multiplies / compares with (dgrad: the stored output of the producing layer,
    // for act'; loss: the exact fp32 inputs) is fetched into registers while the last two K tiles are multiplied -- MI*NI loads
    // per lane, issued after the last refill DMA (so the counted waits above them stay exact) and allowed to stay in flight
    // by the last tile's wait (vmcnt counts them: they are the NAUX youngest operations).
    constexpr bool PREF = REGEP && sizeof(CT) == 2;        // (fp32 operands: the fp32 kernel has no registers to spare)
    constexpr int NAUX = MI * NI;
    typename Quad<CT>::raw yv[MI][NI];
    f32x4 xv[LOSS_TILE ? MI : 1][LOSS_TILE ? NI : 1];
    const int ep_row = m0 + wr * WM + (lane & 15), ep_col = n0 + wc * WN + 4 * (lane >> 4);   // + 16 i, + 16 j
    bool aux_inflight = false;
    auto load_aux = [&]() {
        if constexpr (REGEP) {
            if (!LOSS_TILE && w.kind == K_DGRAD_HIDDEN) {
                const CT* Yp = reinterpret_cast<const CT*>(w.aux0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        yv[i][j] = *reinterpret_cast<const typename Quad<CT>::raw*>(
                            Yp + (size_t)min(ep_row + 16 * i, w.M - 1) * w.ldx + min(ep_col + 16 * j, w.ldx - 4));
            }
            if constexpr (LOSS_TILE) {
                if (w.kind == K_FWD_OUT_LOSS) {
                    const float* X = reinterpret_cast<const float*>(w.aux0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            xv[i][j] = *reinterpret_cast<const f32x4*>(X + (size_t)min(ep_row + 16 * i, w.M - 1) * w.ldx + min(ep_col + 16 * j, w.ldx - 4));
                }
            }
        }
    };
    const bool has_aux = REGEP && (w.kind == K_DGRAD_HIDDEN || w.kind == K_FWD_OUT_LOSS);
    /* tile kt has landed once at most min(rem, RING-2) younger tiles are still in flight (vmcnt needs an immediate) */
#define AVAE_WAIT_TILE(rem, last_wait)                                                                 \
    {                                                                                                  \
        static_assert(RING >= 2 && RING <= 6, "wait ladder written for rings of 2 to 6 stages");       \
        const int nb = (rem) < RING - 2 ? (rem) : RING - 2;                                            \
        if (RING >= 6 && nb == 4) AVAE_WAIT(4 * NCH);                                                  \
        else if (RING >= 5 && nb == 3) AVAE_WAIT(3 * NCH);                                             \
        else if (RING >= 4 && nb == 2) AVAE_WAIT(2 * NCH);                                             \
        else if (RING >= 3 && nb == 1) AVAE_WAIT(NCH);                                                 \
        else { last_wait; }                                                                            \
    }
    if constexpr (NP > 0) {
        if (producer) {
            // ---- producer waves: prologue, then per tile: own pieces of tile kt landed -> barrier (every consumer has read
            // tile kt-1) -> refill the stage tile kt-1 occupied with tile kt+RING-1.  Same barrier count as the consumers.
            const int npro = nk < RING - 1 ? nk : RING - 1;
            for (int p = 0; p < npro; ++p) AVAE_DMA(p, p)
            int buf = 0;
            for (int kt = 0; kt < nk; ++kt) {
                AVAE_WAIT_TILE(nk - 1 - kt, AVAE_WAIT(0))
                asm volatile("s_barrier" ::: "memory");
                const int fill = buf == 0 ? RING - 1 : buf - 1;
                if (kt + RING - 1 < nk) AVAE_ABL_DMA(kt + RING - 1, fill)
                buf = buf + 1 == RING ? 0 : buf + 1;
            }
            // the barriers of the consumers' epilogue (this wave has nothing else to do there)
            if constexpr (TN) { lds_barrier(); lds_barrier(); }
            else if constexpr (LOSS_TILE) { if (w.kind == K_FWD_OUT_LOSS) { lds_barrier(); lds_barrier(); } }
            return;
        }
        // ---- consumer waves: no vector-memory traffic in the loop except the epilogue operand's prefetch, which is theirs alone
        if (PREF && has_aux && nk < 2) { load_aux(); aux_inflight = true; }
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            AVAE_LT0()
            AVAE_LT(0)
            asm volatile("s_barrier" ::: "memory");
            AVAE_LT(1)
            if (kt == 0) { AVAE_STAMP(2) }
            if constexpr (PREF) { if (has_aux && nk >= 2 && kt == nk - 2) { load_aux(); aux_inflight = true; } }
            AVAE_COMPUTE(buf, false, 0, 0)
            buf = buf + 1 == RING ? 0 : buf + 1;
        }
    } else {
    if (PREF && has_aux && nk < 2) load_aux();     // a single K tile: ahead of its DMA (vmcnt retires in order)
    const int npro = nk < RING - 1 ? nk : RING - 1;
    // Weight warm-up for the NEXT launch (8-wave NT tiles): its weight panels were last written by k_adam half a step ago and have
    // left the Infinity Cache; fetched cold they cost that launch 2.5-3 us (a launch repeated right away, its operands warm, ran
    // 23.0 -> 20.1 us).  Every wave touches one dword of n_pf x 64 of their 128-byte lines -- after tile 0's wait, so the first
    // tile does not queue behind these loads; they are older than the refill of tile 2, which therefore retires behind them two
    // iterations later, and they are counted in tile 1's wait (vmcnt counts in issue order).
    int n_pf = 0;
    unsigned pf_v[kMaxPf] = {0u, 0u, 0u, 0u};     // destinations of the warm-up loads: read only after the loop's last wait, so the
                                                  // registers stay reserved while the loads are in flight
    if constexpr (REGEP && RING == 3) n_pf = args.n_pf;
    const bool pf_early = nk < 3;                 // short K loops: ahead of everything (the first wait then covers them)
    auto warm_next = [&]() {
      if constexpr (REGEP && RING == 3) {
        const unsigned nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned idx = (unsigned)tid * nwg + wg;          // lines dealt round-robin over the workgroups
#pragma unroll
        for (int r = 0; r < kMaxPf; ++r)
            if (r < n_pf) {          // every lane loads (out-of-range lanes re-read the last line): exactly one instruction per wave and range
                const unsigned char* p = reinterpret_cast<const unsigned char*>(args.pf_ptr[r]) + (size_t)min(idx, (unsigned)args.pf_lines[r] - 1u) * 128u;
                asm volatile("global_load_dword %0, %1, off" : "=v"(pf_v[r]) : "v"(p));
            }
      }
    };
    {
        if constexpr (REGEP && RING == 3) { if (n_pf > 0 && pf_early) warm_next(); }
        for (int p = 0; p < npro; ++p) AVAE_DMA(p, p)
    }
    int buf = 0;                               // stage of tile kt; the refill goes to the stage freed by tile kt-1
    for (int kt = 0; kt < nk; ++kt) {
        AVAE_LT0()
        // last tile: the epilogue's operand (issued behind the last refill) may still be on its way
        if (REGEP && RING == 3 && kt == 1 && n_pf > 0 && !pf_early) {
            if (n_pf == 1) AVAE_WAIT(NCH + 1); else if (n_pf == 2) AVAE_WAIT(NCH + 2); else if (n_pf == 3) AVAE_WAIT(NCH + 3); else AVAE_WAIT(NCH + 4);
        } else
        AVAE_WAIT_TILE(nk - 1 - kt, if (PREF && aux_inflight) AVAE_WAIT(NAUX); else AVAE_WAIT(0))
        AVAE_LT(0)
        asm volatile("s_barrier" ::: "memory");
        AVAE_LT(1)
        if (kt == 0) { AVAE_STAMP(2) }
        if constexpr (REGEP && RING == 3) { if (kt == 0 && n_pf > 0 && !pf_early) warm_next(); }
        if constexpr (PREF) {
            static_assert(!PREF || RING == 3, "the aux prefetch is placed for a 3-stage ring: no refill is issued in the last two iterations");
            if (has_aux && nk >= 2 && kt == nk - 2) { load_aux(); aux_inflight = true; }
        }
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        AVAE_COMPUTE(buf, kt + RING - 1 < nk, kt + RING - 1, fill)
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
    asm volatile("" :: "v"(pf_v[0]), "v"(pf_v[1]), "v"(pf_v[2]), "v"(pf_v[3]));
    }
#undef AVAE_WAIT_TILE
    lds_barrier();
    AVAE_STAMP(3)
    // ---- tail product: wave v of slice ts owns row block (v & 1) and the column-block pair 2 ts + (v >> 1) of the consuming layer,
    // i.e. 16 rows x 32 columns.  Its B fragments -- weight-shadow rows c0 + 16 j + (lane & 15), 16-byte chunks (lane >> 4) and
    // (lane >> 4) + 4 -- come straight from global memory into registers, issued here so that they travel while the head's own
    // epilogue runs; a block past the layer's last unit re-reads its last row (never stored).
    // column blocks per wave; 64-byte K slabs at most: the narrow result has <= 64 columns = one K tile of bf16, up to two of fp32
    // (bf16: compile-time 2 slabs, no guards -- the guarded 4-slab code cost the bf16 launches 0.7 us each)
    constexpr int TJ = 2, TSL = sizeof(CT) == 2 ? 2 : 4;
    u32x4 tb[TSL][TJ];
    int t_sl = 0;
    typename Quad<CT>::raw ty[TJ];
    const int t_i = wave_u & 1, t_c0 = (2 * ts + (wave_u >> 1)) * 32;
    if constexpr (TAIL) {
        if (tail_on) {
            const unsigned char* wb = reinterpret_cast<const unsigned char*>(w.tail_w);
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const unsigned char* pw = wb + (size_t)min(t_c0 + 16 * j + fr, w.tail_n - 1) * w.tail_ldw * ES + fq * 16;
#pragma unroll
                for (int sl = 0; sl < TSL; ++sl)
                    if (TSL == 2 || sl < 2 * w.tail_kt) tb[sl][j] = *reinterpret_cast<const u32x4*>(pw + 64 * sl);
            }
            t_sl = 2 * w.tail_kt;
            if (w.tail_mode == 2) {
                const CT* Yp = reinterpret_cast<const CT*>(w.tail_aux);
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    ty[j] = *reinterpret_cast<const typename Quad<CT>::raw*>(
                        Yp + (size_t)min(m0 + 16 * t_i + fr, w.M - 1) * w.tail_ldx + min(t_c0 + 16 * j + 4 * fq, w.tail_ldx - 4));
            }
        }
    }
    // Zt: the narrow result as the kind's epilogue left it in LDS (fp32, cs_idx layout), kcols valid columns; one_col: index of the
    // constant-1 (bias) column the stored copy carries, -1 for none.  Same rounding points as the stored copy, same K order.
    const float* t_src = nullptr;            // set by the kinds that leave a narrow result for the tail: the LDS image, its valid columns, its constant-1 column
    int t_cols = 0, t_one = -1;
    auto tail_product = [&](const float* Zt, int kcols, int one_col) __attribute__((always_inline)) {     // (must not become a function: its captures would live in scratch)
        if constexpr (TAIL) {
            AVAE_STAMP_EP(6)
            constexpr int EPC = 16 / ES;
            u32x4 ta[TSL];
#pragma unroll
            for (int sl = 0; sl < TSL; ++sl) {
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int d = (fq + 4 * sl) * EPC + e;
                    v[e] = d < kcols ? Zt[cs_idx<LDC>(16 * t_i + fr, d)] : (d == one_col ? 1.0f : 0.0f);
                }
                if constexpr (sizeof(CT) == 2) ta[sl] = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                else ta[sl] = __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]});
            }
            f32x4 tacc[1][TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                tacc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sl = 0; sl < TSL; ++sl)
                    if (TSL == 2 || sl < t_sl) mma<CT>(tb[sl][j], ta[sl], tacc[0][j]);       // swapped operands: row on the lane, 4 consecutive units in the registers
            }
            AVAE_STAMP_EP(7)
            CT* To = reinterpret_cast<CT*>(w.tail_out);
            if (w.tail_mode == 1) {
                AVAE_ACT_DISPATCH(w.tail_act, (regep_store<CT, 1, TJ>(tacc, To, w.tail_ldo, w.M, w.tail_n, m0 + 16 * t_i + fr, t_c0, lane,
                    [&](int, int, int, float v) { return act_fwd_t<ACT>(v); })))
            } else {
                AVAE_ACT_DISPATCH(w.tail_act, (regep_store<CT, 1, TJ>(tacc, To, w.tail_ldo, w.M, w.tail_n, m0 + 16 * t_i + fr, t_c0, lane,
                    [&](int, int j, int r, float v) { return v * act_bwd_t<ACT>(quad_elem<CT>(ty[j], r)); })))
            }
        }
    };
#if defined(AVAE_STAMPS) && defined(AVAE_LOOPSTAMPS)
    for (int i_ = 0; i_ < 6; ++i_) sv[1 + i_] = la[i_];
    sv[7] = (unsigned long long)nk;
    AVAE_STAMP_FLUSH()
    return;                                    // diagnostic build: no epilogue, results are garbage
#endif
#undef AVAE_DMA
#undef AVAE_WAIT
#undef AVAE_COMPUTE
#undef AVAE_FRAG

    if constexpr (REGEP) {
        // ---- register epilogue (see regep_store): fused maths on the accumulators as they stand, 16-byte stores, no LDS
        const int M = w.M, N = w.N, cwave = n0 + wc * WN;
        float bias[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[j][r] = 0.0f;
        if (w.bias_ep) {             // bias[n] = row `fan-in` of the W_aug shadow (aux1), added here instead of riding in a K tile of its own
            const CT* bp = reinterpret_cast<const CT*>(w.aux1);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const typename Quad<CT>::raw t = *reinterpret_cast<const typename Quad<CT>::raw*>(bp + min(ep_col + 16 * j, w.ld1 - 4));
#pragma unroll
                for (int r = 0; r < 4; ++r) bias[j][r] = quad_elem<CT>(t, r);
            }
        }
        if constexpr (!LOSS_TILE) {
            if (w.kind == K_FWD_HIDDEN) {
                CT* Y = reinterpret_cast<CT*>(w.out0);
                AVAE_ACT_DISPATCH(w.act, (regep_store<CT, MI, NI>(acc, Y, w.ld0, M, N, ep_row, cwave, lane,
                    [&](int, int j, int r, float v) { return act_fwd_t<ACT>(v + bias[j][r]); })))
            } else if (w.kind == K_DGRAD_HIDDEN) {
                if constexpr (!PREF) load_aux();
                CT* dX = reinterpret_cast<CT*>(w.out0);
                AVAE_ACT_DISPATCH(w.act, (regep_store<CT, MI, NI>(acc, dX, w.ld0, M, N, ep_row, cwave, lane,
                    [&](int i, int j, int r, float v) { return v * act_bwd_t<ACT>(quad_elem<CT>(yv[i][j], r)); })))
            }
        } else {
            if (w.kind == K_FWD_OUT_LOSS) {
                // Bernoulli: -sum x log(1e-3+p) + (1-x) log(1e-3+1-p), p = sigmoid(a)  (:321-324), mean over batch (:340)
                // Gaussian : sum (x-a)^2 / 2 over the WHOLE batch, not averaged          (:327-328,:340)
                if constexpr (!PREF) load_aux();
                CT* dA = reinterpret_cast<CT*>(w.out0);
                const float sc = w.scale;
                float csum = 0.0f;
                if (w.binary) {
                    regep_store<CT, MI, NI>(acc, dA, w.ld0, M, N, ep_row, cwave, lane, [&](int i, int j, int r, float a0) {
                        const bool ok = ep_row + 16 * i < M && ep_col + 16 * j + r < N;
                        float sl, da;
                        loss_bernoulli(a0 + bias[j][r], xv[i][j][r], sc, sl, da);
                        csum += ok ? sl : 0.0f;
                        return ok ? da : 0.0f;
                    });
                } else {
                    regep_store<CT, MI, NI>(acc, dA, w.ld0, M, N, ep_row, cwave, lane, [&](int i, int j, int r, float a0) {
                        const bool ok = ep_row + 16 * i < M && ep_col + 16 * j + r < N;
                        float sl, da;
                        loss_gauss(a0 + bias[j][r], xv[i][j][r], sc, sl, da);
                        csum += ok ? sl : 0.0f;
                        return ok ? da : 0.0f;
                    });
                }
                const float total = block_sum<NW>(csum, red);
                if (tid == 0) w.partial[w.slot_base + t] = total;
            }
        }
        AVAE_STAMP(4)
        AVAE_STAMP_FLUSH()
        return;
    }
    // ---- epilogue: accumulators -> LDS tile (fp32), then kind-specific fused passes
    float* Cs = reinterpret_cast<float*>(smem);
    {
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[cs_idx<LDC>(wr * WM + i * 16 + fq * 4 + r, wc * WN + j * 16 + fr)] = acc[i][j][r];
    }
    lds_barrier();
    AVAE_STAMP(5)

    const int M = w.M, N = w.N;
    constexpr int QC = BN / 4;
    if constexpr (TN) {      // the weight gradients are the only K-major products
        const int rows = M + (w.bias_row > 0 ? 1 : 0);
        float* G = w.ksplit > 1 ? reinterpret_cast<float*>(w.out1) + (size_t)ks * rows * w.ld0 : reinterpret_cast<float*>(w.out0);
        tile_pass<float, float, false, false, BM, BN, NT>(Cs, G, w.ld0, (const float*)nullptr, 4, M, N, m0, n0,
            [](float c, float, bool) { return c; });
        if (do_bias && lane < 16) {               // C row 0 of the ones product: lanes 0-15, register 0
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = n0 + wc * WN + j * 16 + lane;
                if (col < N) G[(size_t)w.bias_row * w.ld0 + col] = accb[j][0];
            }
        }
    } else
    switch (w.kind) {
    case K_FWD_HIDDEN: {
        CT* Y = reinterpret_cast<CT*>(w.out0);
        AVAE_ACT_DISPATCH(w.act, (tile_pass<CT, CT, false, false, BM, BN, NT>(Cs, Y, w.ld0, (const CT*)nullptr, 4, M, N, m0, n0,
            [](float c, float, bool) { return act_fwd_t<ACT>(c); })))
        AVAE_STAMP_EP(6)
    } break;
    case K_FWD_HEAD: {
      if constexpr (NW == 4) {     // 8-wave tiles carry the plain kinds only (host: finish_launch)
        // columns [0,nz) = mu, [nz,2nz) = log sigma^2 (vae_assoc.py:217-221); z = mu + sqrt(exp(lv))*eps (:102-103)
        const int nz = w.nz;
        float* Zs = Cs + BM * LDC;                         // second fp32 tile: z
        const float* eps = reinterpret_cast<const float*>(w.aux0);
        if (w.out1) {
            // z = mu + exp(lv/2) * eps: four latent dims per thread and chunk, every eps read in flight before the first
            // use, no division inside the loops (chunk c = tid + k*256 -> (row, quad) by stepping)
            const int QE = w.ldx >> 2;                      // eps row = QE quads (ldx = roundup(n_z, 4))
            constexpr int NCK = BM * 16 / kThreads;          // n_z <= 64 -> QE <= 16 -> at most BM*16 chunks
            const int drow = kThreads / QE, dq = kThreads - drow * QE;
            int row = tid / QE, q = tid - row * QE;
            f32x4 ev[NCK];
            int rws[NCK], qs[NCK];
#pragma unroll
            for (int k = 0; k < NCK; ++k) {
                rws[k] = row; qs[k] = q;
                ev[k] = *reinterpret_cast<const f32x4*>(eps + (size_t)min(m0 + min(row, BM - 1), M - 1) * w.ldx + 4 * q);
                row += drow; q += dq;
                if (q >= QE) { q -= QE; ++row; }
            }
#pragma unroll
            for (int k = 0; k < NCK; ++k) {
                if (rws[k] < BM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int d = 4 * qs[k] + e;
                        if (d < nz)
                            Zs[cs_idx<LDC>(rws[k], d)] = __builtin_fmaf(fexp(0.5f * Cs[cs_idx<LDC>(rws[k], nz + d)]), ev[k][e], Cs[cs_idx<LDC>(rws[k], d)]);
                    }
                }
            }
        }
        if (ts == 0)             // (tail slices > 0 recompute the head tile for its z only)
            tile_pass<float, float, false, false, BM, BN>(Cs, reinterpret_cast<float*>(w.out0), w.ld0, (const float*)nullptr, 4,
                                                          M, 2 * nz, m0, 0, [](float c, float, bool) { return c; });
        if (w.out1) {
            lds_barrier();
            if (ts == 0)
                tile_pass<CT, CT, false, false, BM, BN>(Zs, reinterpret_cast<CT*>(w.out1), w.ld1, (const CT*)nullptr, 8,
                                                        M, nz, m0, 0, [](float c, float, bool) { return c; });
            if (tail_on) { t_src = Zs; t_cols = nz; t_one = nz; }     // the decoder's first layer on [z | 1]
        }
      }
    } break;
    case K_FWD_OUT_LOSS: {
      if constexpr (NW == 4) {     // 8-wave tiles carry the plain kinds only (host: finish_launch)
        // Bernoulli: -sum x log(1e-3+p) + (1-x) log(1e-3+1-p), p = sigmoid(a)  (:321-324), mean over batch (:340)
        // Gaussian : sum (x-a)^2 / 2 over the WHOLE batch, not averaged          (:327-328,:340)
        CT* dA = reinterpret_cast<CT*>(w.out0);
        const float* X = reinterpret_cast<const float*>(w.aux0);
        const float sc = w.scale;
        float csum = 0.0f;
        if (w.binary) {
            tile_pass<CT, float, true, false, BM, BN>(Cs, dA, w.ld0, X, w.ldx, M, N, m0, n0,
                [&csum, sc](float a, float x, bool ok) {
                    float sl, da;
                    loss_bernoulli(a, x, sc, sl, da);
                    csum += ok ? sl : 0.0f;
                    return ok ? da : 0.0f;
                });
        } else {
            tile_pass<CT, float, true, false, BM, BN>(Cs, dA, w.ld0, X, w.ldx, M, N, m0, n0,
                [&csum, sc](float a, float x, bool ok) {
                    float sl, da;
                    loss_gauss(a, x, sc, sl, da);
                    csum += ok ? sl : 0.0f;
                    return ok ? da : 0.0f;
                });
        }
        const float total = block_sum(csum, red);
        if (tid == 0) w.partial[w.slot_base + t] = total;
      }
    } break;
    case K_FWD_OUT_STORE: {
      if constexpr (NW == 4) {     // 8-wave tiles carry the plain kinds only (host: finish_launch)
        if (w.aux2) {            // serving (avae_generate): straight into the caller's dense [rows][N] of modality n_mod, through the slot
            const ServeSlot* sp = reinterpret_cast<const ServeSlot*>(w.aux2);
            float* dst = sp->out[w.n_mod];
            const int rows = sp->rows;
            for (int i = tid; i < BM * BN; i += NT) {
                const int r = i / BN, c = i - r * BN;
                if (m0 + r < rows && n0 + c < N) {
                    const float a = Cs[cs_idx<LDC>(r, c)];
                    dst[(size_t)(m0 + r) * N + n0 + c] = w.binary ? sigmoidf_(a) : a;
                }
            }
            break;
        }
        float* O = reinterpret_cast<float*>(w.out0);
        if (w.binary)
            tile_pass<float, float, false, false, BM, BN>(Cs, O, w.ld0, (const float*)nullptr, 4, M, N, m0, n0,
                [](float a, float, bool) { return sigmoidf_(a); });
        else
            tile_pass<float, float, false, false, BM, BN>(Cs, O, w.ld0, (const float*)nullptr, 4, M, N, m0, n0,
                [](float a, float, bool) { return a; });
      }
    } break;
    case K_DGRAD_HIDDEN: {
        CT* dX = reinterpret_cast<CT*>(w.out0);
        const CT* Yp = reinterpret_cast<const CT*>(w.aux0);
        AVAE_ACT_DISPATCH(w.act, (tile_pass<CT, CT, true, false, BM, BN, NT>(Cs, dX, w.ld0, Yp, w.ldx, M, N, m0, n0,
            [](float c, float y, bool) { return c * act_bwd_t<ACT>(y); })))
    } break;
    case K_DGRAD_LATENT: {
      if constexpr (NW == 4) {     // 8-wave tiles carry the plain kinds only (host: finish_launch)
        // dz -> (dmu, dlv): dmu = dz + g0mu; dlv = dz * F + g0lv with F = 1/2 exp(lv/2) eps   (reparam :102-103);
        // g0 = [g0mu | g0lv | F] per row comes from the K_LATENT item of the forward pass
        const int nz = w.nz;
        float* Zs = Cs + BM * LDC;
        const float* g0 = reinterpret_cast<const float*>(w.aux2);
        {
            // four latent dims per thread and chunk, all g0 reads in flight before the first use, no division in the loops
            const int lde = (nz + 3) & ~3, QE = lde >> 2;   // g0 row = [g0mu | g0lv | F], each lde = roundup(n_z, 4) wide
            constexpr int NCK = BM * 16 / kThreads;          // n_z <= 64 -> QE <= 16 -> at most BM*16 chunks
            const int drow = kThreads / QE, dq = kThreads - drow * QE;
            int row = tid / QE, q = tid - row * QE;
            constexpr int HALF = NCK > 4 ? NCK / 2 : NCK;    // two batches on the 128-row tile: 3 x 4 quads in flight each
#pragma unroll
            for (int k0 = 0; k0 < NCK; k0 += HALF) {
                f32x4 gm[HALF], gl[HALF], gf[HALF];
                int rws[HALF], qs[HALF];
#pragma unroll
                for (int k = 0; k < HALF; ++k) {
                    rws[k] = row; qs[k] = q;
                    const float* gr = g0 + (size_t)min(m0 + min(row, BM - 1), M - 1) * 3 * lde + 4 * q;
                    gm[k] = *reinterpret_cast<const f32x4*>(gr);
                    gl[k] = *reinterpret_cast<const f32x4*>(gr + lde);
                    gf[k] = *reinterpret_cast<const f32x4*>(gr + 2 * lde);
                    row += drow; q += dq;
                    if (q >= QE) { q -= QE; ++row; }
                }
#pragma unroll
                for (int k = 0; k < HALF; ++k) {
                    if (rws[k] < BM) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int d = 4 * qs[k] + e;
                            if (d < nz) {
                                const float dz = Cs[cs_idx<LDC>(rws[k], d)];
                                Zs[cs_idx<LDC>(rws[k], d)] = dz + gm[k][e];
                                Zs[cs_idx<LDC>(rws[k], nz + d)] = __builtin_fmaf(dz, gf[k][e], gl[k][e]);
                            }
                        }
                    }
                }
            }
        }
        lds_barrier();
        if (ts == 0)
            tile_pass<CT, CT, false, false, BM, BN>(Zs, reinterpret_cast<CT*>(w.out0), w.ld0, (const CT*)nullptr, 8,
                                                    M, 2 * nz, m0, 0, [](float c, float, bool) { return c; });
        if (tail_on) { t_src = Zs; t_cols = 2 * nz; t_one = -1; }    // the heads' input gradient on [dmu | dlv]
      }
    } break;
    case K_SERVE_Z: {
      if constexpr (TAIL) {
        const int nz = w.nz, rows = w.n_slots;
        float* Zs = Cs + BM * LDC;
        const float* zsrc = reinterpret_cast<const float*>(w.aux0);
        for (int i = tid; i < BM * nz; i += NT) {
            const int r = i / nz, d = i - r * nz;
            Zs[cs_idx<LDC>(r, d)] = m0 + r < rows ? zsrc[(size_t)(m0 + r) * nz + d] : 0.0f;
        }
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
            ServeSlot* sp = reinterpret_cast<ServeSlot*>(w.partial);
            sp->z = zsrc; sp->rows = rows;
            sp->out[0] = reinterpret_cast<float*>(w.out0); sp->out[1] = reinterpret_cast<float*>(w.out1);
            sp->out[2] = reinterpret_cast<float*>(w.out2); sp->out[3] = reinterpret_cast<float*>(const_cast<void*>(w.aux1));
        }
        lds_barrier();
        if (tail_on) { t_src = Zs; t_cols = nz; t_one = nz; }        // the decoder's first layer on [z | 1]
      }
    } break;
    case K_DGRAD_F32:
    case K_WGRAD: {
        float* G = reinterpret_cast<float*>(w.out0);
        tile_pass<float, float, false, false, BM, BN, NT>(Cs, G, w.ld0, (const float*)nullptr, 4, M, N, m0, n0,
            [](float c, float, bool) { return c; });
    } break;
    default: break;
    }
    if (t_src) tail_product(t_src, t_cols, t_one);                   // (the one call site: see tail_product)
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
    return;
    }   // GEMM kinds
    if constexpr (!TN && (NW == 4 || (NW == 8 && BN == 64))) {
        if (w.kind == K_LATENT) latent_item<NW>(w, t, red);
        else cost_item<NW>(w, st, red);
        AVAE_STAMP(4)
        AVAE_STAMP_FLUSH()
    }
}

// ------------------------------------------------------------------ the lean 32x32 kernel (small nets' hidden layers)
// C1 / C2 / C5 run chains of launches whose K loops are a third of their time; the rest is once-through code, and on a lone wave per
// SIMD every instruction of it is ~5 cycles and every cold 64-byte line of code a miss (a handful of never-taken
// branches cost 0.7 us per launch).  k_grouped's 32x32 instance carries every kind's epilogue behind a switch and stages its
// result through LDS; this kernel is the same product for the two kinds that make up the chain -- K_FWD_HIDDEN, K_DGRAD_HIDDEN --
// and nothing else: kind and transfer function are template parameters, the MFMA operands are swapped (row on the lane, four
// consecutive columns in the registers) and the 16x16 block of a wave goes from the accumulator straight to memory.  Same tiles,
// same K order, same rounding points as k_grouped<CT, 32, 32, 4>: bitwise the same results (AVAE_NO_LEAN=1 for A/B).
template <typename CT, int KIND, int ACT>
__global__ void __launch_bounds__(kThreads) k_small(const LaunchArgs args, unsigned long long* stamps, int launch_id) {
    constexpr int BM = 32, RING = 4, ES = (int)sizeof(CT), NCH = 2;
    constexpr int kStage = 64 * kTileBytesK;                   // A rows 0..31, B rows 32..63 of one K tile
    unsigned char* smem = avae_dyn_smem;
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    AVAE_STAMP(0)
    const int part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const WorkItem w = args.items[blockIdx.y];
    asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out0), "s"(w.aux0), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.ldx));
    int t;
    {
        const int nt = w.tiles_m * w.tiles_n;
        const int q = nt >> 3, r = nt & 7;
        if (idx >= q + (part < r ? 1 : 0)) return;
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tm = t / w.tiles_n, tn = t - tm * w.tiles_n;
    const int m0 = tm * BM, n0 = tn * 32;
    const int nk = (w.K * ES) / kTileBytesK;
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    // this wave's two 1-KiB pieces of a stage: rows 8 wave .. of the A part and of the B part (chunk c of row r at c ^ ((r >> 1) & 7))
    const int prow = wave * 8 + (lane >> 3), lc = ((lane & 7) ^ ((prow >> 1) & 7)) * 16;
    const unsigned char* srcA = reinterpret_cast<const unsigned char*>(w.A) + (size_t)(m0 + prow) * w.lda * ES + lc;
    const unsigned char* srcB = reinterpret_cast<const unsigned char*>(w.B) + (size_t)(n0 + prow) * w.ldb * ES + lc;
    // dgrad: act'(stored output of the layer below), 4 consecutive columns of this lane's row -- older than every tile, so the counted
    // waits below stay exact and its round trip hides behind the first tile's
    const int orow = m0 + wr * 16 + fr, ocol = n0 + wc * 16 + 4 * fq;
    typename Quad<CT>::raw yq;
    if constexpr (KIND == 1)
        yq = *reinterpret_cast<const typename Quad<CT>::raw*>(reinterpret_cast<const CT*>(w.aux0) + (size_t)min(orow, w.M - 1) * w.ldx + min(ocol, w.ldx - 4));
#define AVAE_S_DMA(kt, buf)                                                                                             \
    { __builtin_amdgcn_global_load_lds((gp_t)(srcA + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kStage + wave * 1024), 16, 0, 0);          \
      __builtin_amdgcn_global_load_lds((gp_t)(srcB + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kStage + (4 + wave) * 1024), 16, 0, 0); }
    AVAE_STAMP(1)
    const int npro = nk < RING - 1 ? nk : RING - 1;
    for (int p = 0; p < npro; ++p) AVAE_S_DMA(p, p)
    const int sw0 = (fq ^ (fr >> 1)) * 16;
    const int aoff = (wr * 16 + fr) * kTileBytesK, boff = (32 + wc * 16 + fr) * kTileBytesK;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // two younger tiles (2 pieces each) may be on their way
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt == 0) { AVAE_STAMP(2) }
        const unsigned char* Sb = smem + buf * kStage;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(Sb + aoff + sw0), b0 = *reinterpret_cast<const u32x4*>(Sb + boff + sw0);
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + RING - 1 < nk) AVAE_S_DMA(kt + RING - 1, fill)
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(Sb + aoff + (sw0 ^ 64)), b1 = *reinterpret_cast<const u32x4*>(Sb + boff + (sw0 ^ 64));
        mma<CT>(b0, a0, acc);                                  // swapped: acc[e] = C[row fr][col 4 fq + e]
        mma<CT>(b1, a1, acc);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
#undef AVAE_S_DMA
    AVAE_STAMP(3)
    if constexpr (KIND == 2) {       // K_FWD_OUT_STORE (inference): x_hat = sigmoid(logits) | logits as fp32, to the plan's buffer or, when serving,
        float v[4];                  // straight into the caller's dense [rows][N] of modality n_mod through the slot (avae_generate)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = w.binary ? sigmoidf_(acc[e]) : acc[e];
        if (w.aux2) {
            const ServeSlot* sp = reinterpret_cast<const ServeSlot*>(w.aux2);
            float* dst = sp->out[w.n_mod] + (size_t)orow * w.N + ocol;
            if (orow < sp->rows) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (ocol + e < w.N) dst[e] = v[e];
            }
        } else if (orow < w.M && ocol < w.N) {
            store_row<float>(reinterpret_cast<float*>(w.out0) + (size_t)orow * w.ld0 + ocol, v, w.N - ocol);
        }
    } else if (orow < w.M && ocol < w.N) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (KIND == 0) v[e] = act_fwd_t<ACT>(acc[e]);
            else v[e] = acc[e] * act_bwd_t<ACT>(quad_elem<CT>(yq, e));
        }
        store_row<CT>(reinterpret_cast<CT*>(w.out0) + (size_t)orow * w.ld0 + ocol, v, w.N - ocol);
    }
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
}

// Timing mode (avae_timing_enable): the host arms one (start, stop) event pair per launch; the launch then
// goes through hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin/end timestamps --
// the same signal times rocprofv3 --kernel-trace reports -- instead of bracketing it with marker packets.
thread_local LaunchEvents t_launch_events = {nullptr, nullptr};

#define AVAE_LAUNCH(kernel, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                     \
        if (t_launch_events.start) {                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, t_launch_events.start, t_launch_events.stop, 0, __VA_ARGS__); \
            t_launch_events = LaunchEvents{nullptr, nullptr};                                                \
        } else {                                                                                             \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                               \
        }                                                                                                    \
    } while (0)

template <typename K> static void set_max_lds(K kernel) {
    // > 64 KiB of dynamic LDS has to be opted into once per kernel
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// The small nets' output + loss launch on the same lean frame (tile configuration 9): K_FWD_OUT_LOSS items (Bernoulli or Gaussian per
// item) and the K_LATENT item that rides with them.  The exact fp32 inputs the loss compares with are fetched ahead of the first tile;
// loss, gradient and the tile's cost partial come from the accumulators (one LDS round for the fixed-order block sum).
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_small_loss(const LaunchArgs args, unsigned long long* stamps, int launch_id) {
    constexpr int BM = 32, RING = 4, ES = (int)sizeof(CT);
    constexpr int kStage = 64 * kTileBytesK;
    unsigned char* smem = avae_dyn_smem;
    float* red = reinterpret_cast<float*>(smem + RING * kStage);
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    AVAE_STAMP(0)
    const int part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const WorkItem w = args.items[blockIdx.y];
    asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out0), "s"(w.aux0), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.ldx));
    int t;
    {
        const int nt = w.tiles_m * w.tiles_n;
        const int q = nt >> 3, r = nt & 7;
        if (idx >= q + (part < r ? 1 : 0)) return;
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx;
    }
    if (w.kind == K_LATENT) {
        latent_item<4>(w, t, red);
        AVAE_STAMP(4)
        AVAE_STAMP_FLUSH()
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tm = t / w.tiles_n, tn = t - tm * w.tiles_n;
    const int m0 = tm * BM, n0 = tn * 32;
    const int nk = (w.K * ES) / kTileBytesK;
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    const int prow = wave * 8 + (lane >> 3), lc = ((lane & 7) ^ ((prow >> 1) & 7)) * 16;
    const unsigned char* srcA = reinterpret_cast<const unsigned char*>(w.A) + (size_t)(m0 + prow) * w.lda * ES + lc;
    const unsigned char* srcB = reinterpret_cast<const unsigned char*>(w.B) + (size_t)(n0 + prow) * w.ldb * ES + lc;
    const int orow = m0 + wr * 16 + fr, ocol = n0 + wc * 16 + 4 * fq;
    const f32x4 xq = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(w.aux0) + (size_t)min(orow, w.M - 1) * w.ldx + min(ocol, w.ldx - 4));
#define AVAE_S_DMA(kt, buf)                                                                                             \
    { __builtin_amdgcn_global_load_lds((gp_t)(srcA + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kStage + wave * 1024), 16, 0, 0);          \
      __builtin_amdgcn_global_load_lds((gp_t)(srcB + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kStage + (4 + wave) * 1024), 16, 0, 0); }
    AVAE_STAMP(1)
    const int npro = nk < RING - 1 ? nk : RING - 1;
    for (int p = 0; p < npro; ++p) AVAE_S_DMA(p, p)
    const int sw0 = (fq ^ (fr >> 1)) * 16;
    const int aoff = (wr * 16 + fr) * kTileBytesK, boff = (32 + wc * 16 + fr) * kTileBytesK;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt == 0) { AVAE_STAMP(2) }
        const unsigned char* Sb = smem + buf * kStage;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(Sb + aoff + sw0), b0 = *reinterpret_cast<const u32x4*>(Sb + boff + sw0);
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + RING - 1 < nk) AVAE_S_DMA(kt + RING - 1, fill)
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(Sb + aoff + (sw0 ^ 64)), b1 = *reinterpret_cast<const u32x4*>(Sb + boff + (sw0 ^ 64));
        mma<CT>(b0, a0, acc);
        mma<CT>(b1, a1, acc);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
#undef AVAE_S_DMA
    AVAE_STAMP(3)
    // (loss_bernoulli / loss_gauss; the Bernoulli term is a mean over the batch, the Gaussian a sum over the WHOLE batch: w.scale, :340)
    const float sc = w.scale;
    float csum = 0.0f, v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = orow < w.M && ocol + e < w.N;
        float sl, da;
        if (w.binary) loss_bernoulli(acc[e], xq[e], sc, sl, da);
        else loss_gauss(acc[e], xq[e], sc, sl, da);
        csum += ok ? sl : 0.0f;
        v[e] = ok ? da : 0.0f;
    }
    if (orow < w.M && ocol < w.N) store_row<CT>(reinterpret_cast<CT*>(w.out0) + (size_t)orow * w.ld0 + ocol, v, w.N - ocol);
    const float total = block_sum<4>(csum, red);
    if (tid == 0) w.partial[w.slot_base + t] = total;
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
}
void launch_small_loss(int compute_dtype, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, hipStream_t s, unsigned long long* stamps, int launch_id) {
    const dim3 grid(grid_x, grid_y), block(kThreads);
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_small_loss<__bf16>), grid, block, lds_bytes, s, args, stamps, launch_id);
    else AVAE_LAUNCH((k_small_loss<float>), grid, block, lds_bytes, s, args, stamps, launch_id);
}

// ---- the two head launches of the small nets on the lean frame (tile configurations 10 and 11): 32x64 tile, narrow result into an
// LDS image in the K-tile layout the tail product reads its A fragments from (128-byte rows, chunk c of row r at c ^ ((r >> 1) & 7),
// K tile kt at kt * 32 rows), tail product (WorkItem::tail_*) with kind and transfer function at compile time.
constexpr int kHeadStage = (32 + 64) * kTileBytesK, kHeadImg = 2 * 32 * kTileBytesK;
int small_head_lds_bytes() { return 4 * kHeadStage + kHeadImg + 64; }

template <typename CT> __device__ __forceinline__ void img_put(unsigned char* img, int row, int d, float v) {
    const int b = d * (int)sizeof(CT), bo = b & 127;
    *reinterpret_cast<CT*>(img + (b >> 7) * (32 * kTileBytesK) + row * kTileBytesK + ((((bo >> 4) ^ ((row >> 1) & 7)) << 4) | (bo & 15))) = to_ct<CT>(v);
}

// bwd_dec1_latent + bwd_head: dz = dA . V1^T from the accumulators; dmu = dz + g0mu, dlv = dz * F + g0lv per lane (the latent item's
// static gradients fetched ahead of the first tile); [dmu | dlv] -> dH and -> the image; tail = the heads' input gradient.  The step's
// K_COST item rides in the launch.
template <typename CT, int ACT>
__global__ void __launch_bounds__(kThreads) k_small_latb(const LaunchArgs args, DevState* st, unsigned long long* stamps, int launch_id) {
    constexpr int RING = 4, ES = (int)sizeof(CT), TSL = ES == 2 ? 2 : 4;
    unsigned char* smem = avae_dyn_smem;
    unsigned char* img = smem + RING * kHeadStage;
    float* red = reinterpret_cast<float*>(img + kHeadImg);
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    AVAE_STAMP(0)
    const int part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const WorkItem w = args.items[blockIdx.y];
    asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out0), "s"(w.aux2), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.tail_w), "s"(w.tail_out), "s"(w.tail_aux));
    int t;
    {
        const int nt = w.tiles_m * w.tiles_n;
        const int q = nt >> 3, r = nt & 7;
        if (idx >= q + (part < r ? 1 : 0)) return;
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx;
    }
    if (w.kind == K_COST) {
        cost_item<4>(w, st, red);
        AVAE_STAMP(4)
        AVAE_STAMP_FLUSH()
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tm = t / w.tiles_n, ts = t - tm * w.tiles_n;     // row block, tail slice
    const int m0 = tm * 32;
    const int nk = (w.K * ES) / kTileBytesK, nz = w.nz;
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    reinterpret_cast<f32x4*>(img)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};                 // the image's padding columns: zero (8 KiB = 2 x 256 x 16 B)
    reinterpret_cast<f32x4*>(img)[tid + kThreads] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int orow = m0 + wr * 16 + fr;
    const int lde = (nz + 3) & ~3;
    const int t_i = wave & 1, t_c0 = (2 * ts + (wave >> 1)) * 32, t_sl = 2 * w.tail_kt;
    // ---- K loop: three 1-KiB pieces per wave and stage (A rows 8 wave .., B rows 8 wave .. and 32 + 8 wave ..)
    const int prow = wave * 8 + (lane >> 3), lc = ((lane & 7) ^ ((prow >> 1) & 7)) * 16;
    const unsigned char* srcA = reinterpret_cast<const unsigned char*>(w.A) + (size_t)(m0 + prow) * w.lda * ES + lc;
    const unsigned char* srcB = reinterpret_cast<const unsigned char*>(w.B) + (size_t)prow * w.ldb * ES + lc;
    const size_t b32 = (size_t)32 * w.ldb * ES;
#define AVAE_H_DMA(kt, buf)                                                                                             \
    { __builtin_amdgcn_global_load_lds((gp_t)(srcA + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kHeadStage + wave * 1024), 16, 0, 0);       \
      __builtin_amdgcn_global_load_lds((gp_t)(srcB + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kHeadStage + (4 + wave) * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gp_t)(srcB + b32 + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kHeadStage + (8 + wave) * 1024), 16, 0, 0); }
    AVAE_STAMP(1)
    const int npro = nk < RING - 1 ? nk : RING - 1;
    for (int p = 0; p < npro; ++p) AVAE_H_DMA(p, p)
    // ---- operands of the epilogue and of the tail: EXACTLY kExtra loads per wave, issued BEHIND the prologue's tiles so that the first
    // tile does not queue behind them; the tiles of the prologue are older, so while one of them is awaited these loads may stay in
    // flight and are counted in its wait (vmcnt retires in issue order).  Always TSL slabs per column block: the count is static.
    constexpr int kExtra = 6 + 2 * TSL + 2;
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);       // (neither the compiler nor the scheduler may move them ahead of the tiles)
    f32x4 gm[2], gl[2], gf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* gr = reinterpret_cast<const float*>(w.aux2) + (size_t)min(orow, w.M - 1) * 3 * lde + min(wc * 32 + j * 16 + 4 * fq, lde - 4);
        gm[j] = *reinterpret_cast<const f32x4*>(gr); gl[j] = *reinterpret_cast<const f32x4*>(gr + lde); gf[j] = *reinterpret_cast<const f32x4*>(gr + 2 * lde);
    }
    u32x4 tb[TSL][2];
    typename Quad<CT>::raw ty[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned char* pw = reinterpret_cast<const unsigned char*>(w.tail_w) + (size_t)min(t_c0 + 16 * j + fr, w.tail_n - 1) * w.tail_ldw * ES + fq * 16;
#pragma unroll
        for (int sl = 0; sl < TSL; ++sl) tb[sl][j] = *reinterpret_cast<const u32x4*>(pw + (sl < t_sl ? 64 * sl : 0));
        ty[j] = *reinterpret_cast<const typename Quad<CT>::raw*>(reinterpret_cast<const CT*>(w.tail_aux) + (size_t)min(m0 + 16 * t_i + fr, w.M - 1) * w.tail_ldx
                                                                   + min(t_c0 + 16 * j + 4 * fq, w.tail_ldx - 4));
    }
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    const int sw0 = (fq ^ (fr >> 1)) * 16;
    const int aoff = (wr * 16 + fr) * kTileBytesK, boff = (32 + wc * 32 + fr) * kTileBytesK;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        if (kt < npro) {             // a tile of the prologue: the extra loads are younger than it
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 + kExtra) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + kExtra) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kExtra) : "memory");
        } else {
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        if (kt == 0) { AVAE_STAMP(2) }
        const unsigned char* Sb = smem + buf * kHeadStage;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(Sb + aoff + sw0);
        const u32x4 b00 = *reinterpret_cast<const u32x4*>(Sb + boff + sw0), b01 = *reinterpret_cast<const u32x4*>(Sb + boff + 16 * kTileBytesK + sw0);
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + RING - 1 < nk) AVAE_H_DMA(kt + RING - 1, fill)
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(Sb + aoff + (sw0 ^ 64));
        const u32x4 b10 = *reinterpret_cast<const u32x4*>(Sb + boff + (sw0 ^ 64)), b11 = *reinterpret_cast<const u32x4*>(Sb + boff + 16 * kTileBytesK + (sw0 ^ 64));
        mma<CT>(b00, a0, acc[0]); mma<CT>(b01, a0, acc[1]);
        mma<CT>(b10, a1, acc[0]); mma<CT>(b11, a1, acc[1]);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
#undef AVAE_H_DMA
    // The count of the extra loads has to hold for EVERY transfer function: with the identity, act'(y) = 1 needs no y, the compiler
    // dropped the two `ty` loads, and the waits above -- written for kExtra younger loads -- let two pieces of the awaited tile stay in
    // flight: a K tile was read before it had landed about once in 60 steps (tools/fuzz_parity.py api found runs that differed from
    // themselves).  Naming the registers as inputs of an empty statement HERE (behind the loop: no wait is forced early) keeps the loads.
    asm volatile("" :: "v"(ty[0]), "v"(ty[1]));
    AVAE_STAMP(3)
    // ---- dz -> (dmu, dlv): into the image (every slice) and into dH (slice 0)
    {
        CT* dH = reinterpret_cast<CT*>(w.out0) + (size_t)orow * w.ld0;
        const bool store = ts == 0 && orow < w.M;
        const int lrow = wr * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = wc * 32 + j * 16 + 4 * fq + e;
                if (d < nz) {
                    const float dz = acc[j][e];
                    const float dmu = dz + gm[j][e], dlv = __builtin_fmaf(dz, gf[j][e], gl[j][e]);
                    img_put<CT>(img, lrow, d, dmu);
                    img_put<CT>(img, lrow, nz + d, dlv);
                    if (store) { dH[d] = to_ct<CT>(dmu); dH[nz + d] = to_ct<CT>(dlv); }
                }
            }
    }
    lds_barrier();
    AVAE_STAMP(5)
    // ---- tail product: the heads' input gradient on [dmu | dlv]
    {
        const int arow = 16 * t_i + fr;
        f32x4 tacc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
        for (int sl = 0; sl < TSL; ++sl)
            if (TSL == 2 || sl < t_sl) {
                const u32x4 ta = *reinterpret_cast<const u32x4*>(img + (sl >> 1) * (32 * kTileBytesK) + arow * kTileBytesK + (((fq + 4 * (sl & 1)) ^ ((arow >> 1) & 7)) << 4));
                mma<CT>(tb[sl][0], ta, tacc[0][0]);
                mma<CT>(tb[sl][1], ta, tacc[0][1]);
            }
        regep_store<CT, 1, 2>(tacc, reinterpret_cast<CT*>(w.tail_out), w.tail_ldo, w.M, w.tail_n, m0 + arow, t_c0, lane,
            [&](int, int j, int r, float v) { return v * act_bwd_t<ACT>(quad_elem<CT>(ty[j], r)); });
    }
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
}
// fwd_head + fwd_dec1: [mu | lv] from the accumulators -> mulv (fp32, slice 0) and -> an fp32 tile in LDS (the ring is free by then);
// z = mu + exp(lv / 2) * eps by one thread per (row, four latent dims) with eps fetched ahead of the first tile -> Z (slice 0) and
// -> the image, with its constant-1 column; tail = the decoder's first layer.
template <typename CT, int ACT>
__global__ void __launch_bounds__(kThreads) k_small_head(const LaunchArgs args, unsigned long long* stamps, int launch_id) {
    constexpr int RING = 4, ES = (int)sizeof(CT), TSL = ES == 2 ? 2 : 4, LDM = 68;
    unsigned char* smem = avae_dyn_smem;
    unsigned char* img = smem + RING * kHeadStage;
    float* MV = reinterpret_cast<float*>(smem);                // [32][LDM] fp32, over the ring (after the loop's last barrier)
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    AVAE_STAMP(0)
    const int part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const WorkItem w = args.items[blockIdx.y];
    asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out0), "s"(w.out1), "s"(w.aux0), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.tail_w), "s"(w.tail_out));
    int t;
    {
        const int nt = w.tiles_m * w.tiles_n;
        const int q = nt >> 3, r = nt & 7;
        if (idx >= q + (part < r ? 1 : 0)) return;
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tm = t / w.tiles_n, ts = t - tm * w.tiles_n;
    const int m0 = tm * 32;
    const int nk = (w.K * ES) / kTileBytesK, nz = w.nz;
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    reinterpret_cast<f32x4*>(img)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4*>(img)[tid + kThreads] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- fetched ahead of the first tile: eps of this thread's (row, four dims), the tail's weight fragments
    const int zrow = tid >> 3, zg = tid & 7;
    const f32x4 ev = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(w.aux0) + (size_t)min(m0 + zrow, w.M - 1) * w.ldx + min(4 * zg, w.ldx - 4));
    const int t_i = wave & 1, t_c0 = (2 * ts + (wave >> 1)) * 32, t_sl = 2 * w.tail_kt;
    u32x4 tb[TSL][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned char* pw = reinterpret_cast<const unsigned char*>(w.tail_w) + (size_t)min(t_c0 + 16 * j + fr, w.tail_n - 1) * w.tail_ldw * ES + fq * 16;
#pragma unroll
        for (int sl = 0; sl < TSL; ++sl) if (TSL == 2 || sl < t_sl) tb[sl][j] = *reinterpret_cast<const u32x4*>(pw + 64 * sl);
    }
    const int prow = wave * 8 + (lane >> 3), lc = ((lane & 7) ^ ((prow >> 1) & 7)) * 16;
    const unsigned char* srcA = reinterpret_cast<const unsigned char*>(w.A) + (size_t)(m0 + prow) * w.lda * ES + lc;
    const unsigned char* srcB = reinterpret_cast<const unsigned char*>(w.B) + (size_t)prow * w.ldb * ES + lc;
    const size_t b32 = (size_t)32 * w.ldb * ES;
#define AVAE_H_DMA(kt, buf)                                                                                             \
    { __builtin_amdgcn_global_load_lds((gp_t)(srcA + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kHeadStage + wave * 1024), 16, 0, 0);       \
      __builtin_amdgcn_global_load_lds((gp_t)(srcB + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kHeadStage + (4 + wave) * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gp_t)(srcB + b32 + (size_t)(kt) * kTileBytesK), (lp_t)(smem + (buf) * kHeadStage + (8 + wave) * 1024), 16, 0, 0); }
    AVAE_STAMP(1)
    const int npro = nk < RING - 1 ? nk : RING - 1;
    for (int p = 0; p < npro; ++p) AVAE_H_DMA(p, p)
    const int sw0 = (fq ^ (fr >> 1)) * 16;
    const int aoff = (wr * 16 + fr) * kTileBytesK, boff = (32 + wc * 32 + fr) * kTileBytesK;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt == 0) { AVAE_STAMP(2) }
        const unsigned char* Sb = smem + buf * kHeadStage;
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(Sb + aoff + sw0);
        const u32x4 b00 = *reinterpret_cast<const u32x4*>(Sb + boff + sw0), b01 = *reinterpret_cast<const u32x4*>(Sb + boff + 16 * kTileBytesK + sw0);
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + RING - 1 < nk) AVAE_H_DMA(kt + RING - 1, fill)
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(Sb + aoff + (sw0 ^ 64));
        const u32x4 b10 = *reinterpret_cast<const u32x4*>(Sb + boff + (sw0 ^ 64)), b11 = *reinterpret_cast<const u32x4*>(Sb + boff + 16 * kTileBytesK + (sw0 ^ 64));
        mma<CT>(b00, a0, acc[0]); mma<CT>(b01, a0, acc[1]);
        mma<CT>(b10, a1, acc[0]); mma<CT>(b11, a1, acc[1]);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
#undef AVAE_H_DMA
    lds_barrier();                                              // every wave is done with the ring: MV may overwrite it
    AVAE_STAMP(3)
    // ---- [mu | lv]: into MV (every slice) and into mulv (slice 0), columns [0, nz) = mu, [nz, 2 nz) = log sigma^2 (vae_assoc.py:217-221)
    {
        const int orow = m0 + wr * 16 + fr, lrow = wr * 16 + fr;
        float* mulv = reinterpret_cast<float*>(w.out0) + (size_t)orow * w.ld0;
        const bool store = ts == 0 && orow < w.M;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c0 = wc * 32 + j * 16 + 4 * fq;
            *reinterpret_cast<f32x4*>(MV + lrow * LDM + c0) = acc[j];
            if (store) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c0 + e < 2 * nz) mulv[c0 + e] = acc[j][e];
            }
        }
    }
    lds_barrier();
    AVAE_STAMP(5)
    // ---- z = mu + sqrt(exp(lv)) * eps (:102-103): thread (row, four dims) -> Z (slice 0) and the image; [z | 1] is the tail's operand
    {
        CT* Z = reinterpret_cast<CT*>(w.out1) + (size_t)(m0 + zrow) * w.ld1;
        const bool store = ts == 0 && m0 + zrow < w.M;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = 4 * zg + e;
            if (d < nz) {
                const float z = __builtin_fmaf(fexp(0.5f * MV[zrow * LDM + nz + d]), ev[e], MV[zrow * LDM + d]);
                img_put<CT>(img, zrow, d, z);
                if (store) Z[d] = to_ct<CT>(z);
            }
        }
        if (tid < 32) img_put<CT>(img, tid, nz, 1.0f);         // the constant-1 (bias) column of [z | 1]
    }
    lds_barrier();
    AVAE_STAMP_EP(6)
    // ---- tail product: the decoder's first layer on [z | 1]
    {
        const int arow = 16 * t_i + fr;
        f32x4 tacc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
        for (int sl = 0; sl < TSL; ++sl)
            if (TSL == 2 || sl < t_sl) {
                const u32x4 ta = *reinterpret_cast<const u32x4*>(img + (sl >> 1) * (32 * kTileBytesK) + arow * kTileBytesK + (((fq + 4 * (sl & 1)) ^ ((arow >> 1) & 7)) << 4));
                mma<CT>(tb[sl][0], ta, tacc[0][0]);
                mma<CT>(tb[sl][1], ta, tacc[0][1]);
            }
        regep_store<CT, 1, 2>(tacc, reinterpret_cast<CT*>(w.tail_out), w.tail_ldo, w.M, w.tail_n, m0 + arow, t_c0, lane,
            [&](int, int, int, float v) { return act_fwd_t<ACT>(v); });
    }
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
}
// every transfer function of the ABI has its instance of the lean kernels (the choice is made per launch on the host)
void launch_small_head(int compute_dtype, int act, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, hipStream_t s,
                       unsigned long long* stamps, int launch_id) {
    const dim3 grid(grid_x, grid_y), block(kThreads);
#define K_HEAD_B(A) k_small_head<__bf16, A>
#define K_HEAD_F(A) k_small_head<float, A>
    if (compute_dtype == AVAE_BF16) {
        switch (act) {
            case AVAE_ACT_RELU: AVAE_LAUNCH((K_HEAD_B(AVAE_ACT_RELU)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            case AVAE_ACT_SOFTPLUS: AVAE_LAUNCH((K_HEAD_B(AVAE_ACT_SOFTPLUS)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            case AVAE_ACT_SIGMOID: AVAE_LAUNCH((K_HEAD_B(AVAE_ACT_SIGMOID)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            case AVAE_ACT_TANH: AVAE_LAUNCH((K_HEAD_B(AVAE_ACT_TANH)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            default: AVAE_LAUNCH((K_HEAD_B(AVAE_ACT_IDENTITY)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
        }
    } else {
        switch (act) {
            case AVAE_ACT_RELU: AVAE_LAUNCH((K_HEAD_F(AVAE_ACT_RELU)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            case AVAE_ACT_SOFTPLUS: AVAE_LAUNCH((K_HEAD_F(AVAE_ACT_SOFTPLUS)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            case AVAE_ACT_SIGMOID: AVAE_LAUNCH((K_HEAD_F(AVAE_ACT_SIGMOID)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            case AVAE_ACT_TANH: AVAE_LAUNCH((K_HEAD_F(AVAE_ACT_TANH)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
            default: AVAE_LAUNCH((K_HEAD_F(AVAE_ACT_IDENTITY)), grid, block, lds_bytes, s, args, stamps, launch_id); break;
        }
    }
#undef K_HEAD_B
#undef K_HEAD_F
}
void launch_small_latb(int compute_dtype, int act, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, DevState* st, hipStream_t s,
                       unsigned long long* stamps, int launch_id) {
    const dim3 grid(grid_x, grid_y), block(kThreads);
#define K_LATB_B(A) k_small_latb<__bf16, A>
#define K_LATB_F(A) k_small_latb<float, A>
    if (compute_dtype == AVAE_BF16) {
        switch (act) {
            case AVAE_ACT_RELU: AVAE_LAUNCH((K_LATB_B(AVAE_ACT_RELU)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            case AVAE_ACT_SOFTPLUS: AVAE_LAUNCH((K_LATB_B(AVAE_ACT_SOFTPLUS)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            case AVAE_ACT_SIGMOID: AVAE_LAUNCH((K_LATB_B(AVAE_ACT_SIGMOID)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            case AVAE_ACT_TANH: AVAE_LAUNCH((K_LATB_B(AVAE_ACT_TANH)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            default: AVAE_LAUNCH((K_LATB_B(AVAE_ACT_IDENTITY)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
        }
    } else {
        switch (act) {
            case AVAE_ACT_RELU: AVAE_LAUNCH((K_LATB_F(AVAE_ACT_RELU)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            case AVAE_ACT_SOFTPLUS: AVAE_LAUNCH((K_LATB_F(AVAE_ACT_SOFTPLUS)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            case AVAE_ACT_SIGMOID: AVAE_LAUNCH((K_LATB_F(AVAE_ACT_SIGMOID)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            case AVAE_ACT_TANH: AVAE_LAUNCH((K_LATB_F(AVAE_ACT_TANH)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
            default: AVAE_LAUNCH((K_LATB_F(AVAE_ACT_IDENTITY)), grid, block, lds_bytes, s, args, st, stamps, launch_id); break;
        }
    }
#undef K_LATB_B
#undef K_LATB_F
}

// ---- the small nets' weight-gradient launch on the lean frame, with the optimiser in its epilogue (tile configuration 12):
// dW[m][n] = sum_k X[k][m] * dA[k][n], 64x64 tiles, K-major operand images and fragment reads exactly as k_grouped's TN instance
// (tn_frag_off / tn_frag), MFMA operands swapped so that a lane holds one gradient row's four consecutive columns.  ADAM: the block
// is applied to theta / m / v as it leaves the accumulators (adam_update, the arithmetic of k_adam), the gradient itself is still
// stored (avae_get_grads, the cost slot's neighbours), and both compute-dtype shadows of the layer are refreshed -- k_adam's launch,
// its boundary and the gradient's read-back disappear.  Launches with split-K or all-ones bias rows stay on the general kernel.
template <typename CT, bool ADAM>
__global__ void __launch_bounds__(kThreads) k_small_tn(const TnLaunchArgs args, unsigned long long* stamps, int launch_id) {
    constexpr int BM = 64, RING = 4, ES = (int)sizeof(CT), EPR = kTileBytesK / ES, NCH = 4;
    constexpr int kStage = 2 * BM * kTileBytesK;               // A part (BM columns) then B part: 16 KiB, EPR k-rows
    unsigned char* smem = avae_dyn_smem;
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    AVAE_STAMP(0)
    const int part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int entry = blockIdx.y, t_piece = -1;
    if (args.xcd_pieces) {       // XCD-owned pieces: this XCD's list of (layer, tile range); one more kernel-argument round trip than below
        const TnPiece* row = args.pieces[part];
        int e = 0, lo = 0;
#pragma unroll
        for (int i = 0; i < kTnPieces; ++i) { const int ce = row[i].cum_end; if (idx >= ce) { e = i + 1; lo = ce; } }
        if (e >= kTnPieces) return;                                   // behind this XCD's last tile
        entry = row[e].item; t_piece = row[e].tile_off + idx - lo;
    }
    const TnItem w = args.items[entry];
    asm volatile("" :: "s"(w.A), "s"(w.B), "s"(w.out), "s"(w.lda), "s"(w.ldb), "s"(w.K), "s"(w.ld0), "s"(w.tile_off));
    float lr_t = 0.f;
    if constexpr (ADAM) lr_t = args.st->lr_t;        // older than every tile: whatever load it becomes, the counted waits hold
    int t;
    if (t_piece >= 0) {
        t = t_piece;
    } else {
        const int nt = w.tile_cnt;
        const int q = nt >> 3, r = nt & 7;
        if (idx >= q + (part < r ? 1 : 0)) return;
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx + w.tile_off;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tm = t / w.tiles_n, tn = t - tm * w.tiles_n;
    const int m0 = tm * BM, n0 = tn * BM;
    const int nk = (w.K * ES) / kTileBytesK;
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    const unsigned char* src[NCH];
    size_t kadv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int r = (c * 4 + wave) * 8 + (lane >> 3);
        const bool is_a = c < 2;
        const int rr = is_a ? r : r - BM, sub = rr / EPR, k = rr - sub * EPR;
        const int lc = (lane & 7) ^ tn_swz<CT>(k);
        const size_t ld_b = (size_t)(is_a ? w.lda : w.ldb) * ES;
        kadv[c] = (size_t)EPR * ld_b;
        src[c] = reinterpret_cast<const unsigned char*>(is_a ? w.A : w.B) + (size_t)(is_a ? m0 : n0) * ES + (size_t)k * ld_b + (size_t)sub * kTileBytesK + lc * 16;
    }
#define AVAE_T_DMA(kt, buf)                                                                                             \
    { _Pragma("unroll") for (int c = 0; c < NCH; ++c)                                                                   \
        __builtin_amdgcn_global_load_lds((gp_t)(src[c] + (size_t)(kt) * kadv[c]), (lp_t)(smem + (buf) * kStage + (c * 4 + wave) * 1024), 16, 0, 0); }
    int offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { offA[i] = tn_frag_off<CT>(wr * 32 + i * 16, lane); offB[i] = BM * kTileBytesK + tn_frag_off<CT>(wc * 32 + i * 16, lane); }
    AVAE_STAMP(1)
    const int npro = nk < RING - 1 ? nk : RING - 1;
    for (int p = 0; p < npro; ++p) AVAE_T_DMA(p, p)
    // ADAM: this thread's theta / m / v of the block (4 rows x 16 bytes x 3 arrays), fetched in ONE burst at the top of the epilogue
    // (clamped addresses: every lane loads something valid).  Fetching them beside the K loop instead (behind the prologue's tiles,
    // counted in the loop's waits) was built and measured: no faster (C2: 12.4 us either way) -- the launch is paced by the bytes its
    // epilogue moves, not by this round trip.
    const int c4 = (tid & 15) * 4;
    f32x4 pth[4], pmm[4], pvv[4];
    auto fetch_state = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int grow = m0 + (tid >> 4) + 16 * k, gcol = n0 + c4;
            const bool ok = grow < w.M && gcol < w.N;
            const float* gp = w.out + (size_t)(ok ? grow : m0) * w.ld0 + (ok ? gcol : n0);
            pth[k] = *reinterpret_cast<const f32x4*>(gp + args.d_theta);
            pmm[k] = *reinterpret_cast<const f32x4*>(gp + args.d_m);
            pvv[k] = *reinterpret_cast<const f32x4*>(gp + args.d_v);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (kt == 0) { AVAE_STAMP(2) }
        const unsigned char* Sb = smem + buf * kStage;
        u32x4 a0[2], b0[2], a1[2], b1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a0[i] = tn_frag<CT>(Sb, offA[i], 0); b0[i] = tn_frag<CT>(Sb, offB[i], 0); }
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + RING - 1 < nk) AVAE_T_DMA(kt + RING - 1, fill)
#pragma unroll
        for (int i = 0; i < 2; ++i) { a1[i] = tn_frag<CT>(Sb, offA[i], 1); b1[i] = tn_frag<CT>(Sb, offB[i], 1); }
        if constexpr (sizeof(CT) == 2) { tn_wait_lds(); tn_frags_ready(a0); tn_frags_ready(b0); tn_frags_ready(a1); tn_frags_ready(b1); }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma<CT>(b0[j], a0[i], acc[i][j]);       // swapped: acc[i][j][e] = dW[row fr of block i][col 4 fq + e of block j]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma<CT>(b1[j], a1[i], acc[i][j]);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
#undef AVAE_T_DMA
    AVAE_STAMP(3)
    if constexpr (!ADAM) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + wr * 32 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wc * 32 + j * 16 + 4 * fq;
                if (row < w.M && col < w.N) {
                    const float g[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    store_row<float>(w.out + (size_t)row * w.ld0 + col, g, w.N - col);
                }
            }
        }
    } else {
        // The optimiser wants the block ROW-contiguous (16 lanes x 16 B = a row's 256 bytes per access, as k_adam reads theta / m / v; from
        // the accumulators a lane owns one row's 16 bytes and a wave touches 16 rows at 64 bytes: 15.3 us for the launch that way):
        // accumulators -> an fp32 tile in LDS (the ring is free), then k_adam's tile body on it, W^T through the same tile.
        constexpr int LDT = 65;
        float* T = reinterpret_cast<float*>(smem);
        fetch_state();                         // all twelve loads of the thread at once: ONE round trip (they used to be fetched per row group, behind the previous group's stores)
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(wr * 32 + i * 16 + fr) * LDT + wc * 32 + j * 16 + 4 * fq + e] = acc[i][j][e];
        lds_barrier();
        const float omb1 = 1.0f - args.beta1, omb2 = 1.0f - args.beta2;
        const int r4 = (tid & 15) * 4;                          // W^T: a lane takes four consecutive rows (m) of one column (n)
        // The update itself runs for all four row groups of the thread BEFORE any store and outside any branch (clamped addresses
        // made every fetched value a number): the wait for the twelve loads then sits in straight-line code.  Placed behind a
        // row-validity branch it was repeated before every group -- as vmcnt(0), which on this ISA also waits for the previous
        // group's stores to reach L2: four store round trips in a row.
        float g[4][4], th[4][4], m[4][4], v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (tid >> 4) + 16 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[k][e] = T[r * LDT + c4 + e]; th[k][e] = pth[k][e]; m[k][e] = pmm[k][e]; v[k][e] = pvv[k][e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) adam_update(g[k][e], m[k][e], v[k][e], th[k][e], omb1, omb2, lr_t, args.eps);
#pragma unroll
            for (int e = 0; e < 4; ++e) T[r * LDT + c4 + e] = th[k][e];
        }
        const bool full = m0 + BM <= w.M && n0 + BM <= w.N;     // a tile inside the layer (most of them): whole 16-byte stores, no bounds
        auto store_group = [&](int k, float* gp, CT* wp) {
            store4<float>(gp, g[k]);
            store4<float>(gp + args.d_theta, th[k]); store4<float>(gp + args.d_m, m[k]); store4<float>(gp + args.d_v, v[k]);
            store4<CT>(wp, th[k]);
        };
        if (full) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int grow = m0 + (tid >> 4) + 16 * k, gcol = n0 + c4;
                store_group(k, w.out + (size_t)grow * w.ld0 + gcol, reinterpret_cast<CT*>(w.W) + (size_t)grow * w.ldw + gcol);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int grow = m0 + (tid >> 4) + 16 * k, gcol = n0 + c4;
                float* gp = w.out + (size_t)grow * w.ld0 + gcol;
                CT* wp = reinterpret_cast<CT*>(w.W) + (size_t)grow * w.ldw + gcol;
                const int nv = w.N - gcol;
                if (grow < w.M && nv >= 4) {
                    store_group(k, gp, wp);
                } else if (grow < w.M && nv > 0) {
                    // a partial quad never touches the padding behind column N
#pragma unroll
                    for (int e = 0; e < 3; ++e) if (e < nv) {
                        gp[e] = g[k][e]; gp[args.d_theta + e] = th[k][e]; gp[args.d_m + e] = m[k][e]; gp[args.d_v + e] = v[k][e]; wp[e] = to_ct<CT>(th[k][e]);
                    }
                    // Keeps this arm's last store from being merged with the 16-byte stores of the other arm: hipcc (ROCm 7.2) sank
                    // the two into one shared "fourth dword" store and left its value unset on the whole-quad lanes -- fp32
                    // shadows came out with every fourth column stale (tools/adam_fuse_diag.py found it).
                    asm volatile("; partial quad" ::: "memory");
                }
            }
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (tid >> 4) + 16 * k;
            const int gcol = n0 + c, grow = m0 + r4;
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = T[(r4 + e) * LDT + c];
            CT* tp = reinterpret_cast<CT*>(w.Wt) + (size_t)gcol * w.ldt + grow;
            if (full) store4<CT>(tp, t);
            else if (gcol < w.N && grow < w.M) store_row<CT>(tp, t, w.M - grow);
        }
    }
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
}
void launch_small_tn(int compute_dtype, const TnLaunchArgs& args, int grid_x, int grid_y, hipStream_t s, unsigned long long* stamps, int launch_id) {
    const dim3 grid(grid_x, grid_y), block(kThreads);
    const int lds_bytes = 4 * 2 * 64 * kTileBytesK;
    if (compute_dtype == AVAE_BF16) {
        if (args.adam) AVAE_LAUNCH((k_small_tn<__bf16, true>), grid, block, lds_bytes, s, args, stamps, launch_id);
        else AVAE_LAUNCH((k_small_tn<__bf16, false>), grid, block, lds_bytes, s, args, stamps, launch_id);
    } else {
        if (args.adam) AVAE_LAUNCH((k_small_tn<float, true>), grid, block, lds_bytes, s, args, stamps, launch_id);
        else AVAE_LAUNCH((k_small_tn<float, false>), grid, block, lds_bytes, s, args, stamps, launch_id);
    }
}

template <typename CT, int KIND>
static void launch_small_act(int act, const LaunchArgs& args, dim3 grid, int lds_bytes, hipStream_t s, unsigned long long* stamps, int launch_id) {
    const dim3 block(kThreads);
    switch (act) {
        case AVAE_ACT_RELU: AVAE_LAUNCH((k_small<CT, KIND, AVAE_ACT_RELU>), grid, block, lds_bytes, s, args, stamps, launch_id); break;
        case AVAE_ACT_SOFTPLUS: AVAE_LAUNCH((k_small<CT, KIND, AVAE_ACT_SOFTPLUS>), grid, block, lds_bytes, s, args, stamps, launch_id); break;
        case AVAE_ACT_SIGMOID: AVAE_LAUNCH((k_small<CT, KIND, AVAE_ACT_SIGMOID>), grid, block, lds_bytes, s, args, stamps, launch_id); break;
        case AVAE_ACT_TANH: AVAE_LAUNCH((k_small<CT, KIND, AVAE_ACT_TANH>), grid, block, lds_bytes, s, args, stamps, launch_id); break;
        default: AVAE_LAUNCH((k_small<CT, KIND, AVAE_ACT_IDENTITY>), grid, block, lds_bytes, s, args, stamps, launch_id); break;
    }
}
// tile_cfg 7 (host: finish_launch): every item of the launch is of ONE kind (hidden forward, hidden dgrad or output store) and one transfer function
void launch_small(int compute_dtype, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, hipStream_t s, unsigned long long* stamps, int launch_id) {
    const dim3 grid(grid_x, grid_y);
    const int kind = args.items[0].kind == K_DGRAD_HIDDEN ? 1 : 0, act = args.items[0].act;
    if (args.items[0].kind == K_FWD_OUT_STORE) {
        const dim3 block(kThreads);
        if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_small<__bf16, 2, AVAE_ACT_IDENTITY>), grid, block, lds_bytes, s, args, stamps, launch_id);
        else AVAE_LAUNCH((k_small<float, 2, AVAE_ACT_IDENTITY>), grid, block, lds_bytes, s, args, stamps, launch_id);
        return;
    }
    if (compute_dtype == AVAE_BF16) { if (kind) launch_small_act<__bf16, 1>(act, args, grid, lds_bytes, s, stamps, launch_id); else launch_small_act<__bf16, 0>(act, args, grid, lds_bytes, s, stamps, launch_id); }
    else { if (kind) launch_small_act<float, 1>(act, args, grid, lds_bytes, s, stamps, launch_id); else launch_small_act<float, 0>(act, args, grid, lds_bytes, s, stamps, launch_id); }
}

// ------------------------------------------------------------------ experiment: two chain links in ONE launch
// fwd_enc1 -> fwd_enc2 of the small nets behind a row-block-local hand-off instead of a kernel boundary.  A tile (tm, tn) of layer 2
// needs the 32 rows tm of layer 1, i.e. the tiles_n tiles of its own row block -- with batch 256 that is 8 row blocks, and the lean
// kernels' tile order already gives an XCD the 16 (7) column tiles of ONE row block.  Workgroup (tm, tn) computes its layer-1 tile,
// stores it write-through (sc1: nothing of it is left dirty in this XCD's L2), drains its stores, takes a ticket on the row block's
// counter (agent-scope atomic; monotonic over launches: no reset), polls until the block's tiles_n tickets of this launch are in
// (bounded spin), then computes its layer-2 tile from the handed-off rows (first touch of those lines in this kernel: the CU's L1
// cannot hold them; the LDS-DMA loads carry sc1 as well).  Placement decides speed only: tickets and sc1 accesses are agent-scope.
// Items: args.items[y] = layer 1 of modality y, args.items[n_items/2 + y] = layer 2.  AVAE_CHAIN2=1 turns it on (A/B).
template <typename CT, int ACT>
__global__ void __launch_bounds__(kThreads) k_chain2(const LaunchArgs args, unsigned* counters, unsigned* err) {
    constexpr int BM = 32, RING = 4, ES = (int)sizeof(CT);
    constexpr int kStage = 64 * kTileBytesK;
    unsigned char* smem = avae_dyn_smem;
    const int part = blockIdx.x & 7, idx = blockIdx.x >> 3, half = args.n_items >> 1;
    const WorkItem w1 = args.items[blockIdx.y], w2 = args.items[half + blockIdx.y];
    asm volatile("" :: "s"(w1.A), "s"(w1.B), "s"(w1.out0), "s"(w1.lda), "s"(w1.ldb), "s"(w1.K), "s"(w1.ld0), "s"(w2.B), "s"(w2.out0), "s"(w2.ldb), "s"(w2.K), "s"(w2.ld0));
    int t;
    {
        const int nt = w1.tiles_m * w1.tiles_n;
        const int q = nt >> 3, r = nt & 7;
        if (idx >= q + (part < r ? 1 : 0)) return;
        t = (part < r ? part * (q + 1) : r * (q + 1) + (part - r) * q) + idx;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int tm = t / w1.tiles_n, tn = t - tm * w1.tiles_n;
    const int m0 = tm * BM, n0 = tn * 32;
    typedef const __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    const int prow = wave * 8 + (lane >> 3), lc = ((lane & 7) ^ ((prow >> 1) & 7)) * 16;
    const int sw0 = (fq ^ (fr >> 1)) * 16;
    const int aoff = (wr * 16 + fr) * kTileBytesK, boff = (32 + wc * 16 + fr) * kTileBytesK;
    const int orow = m0 + wr * 16 + fr, ocol = n0 + wc * 16 + 4 * fq;
    f32x4 acc;
#define AVAE_C_LOOP(W, AUX)                                                                                               \
    {                                                                                                                     \
        const unsigned char* srcA = reinterpret_cast<const unsigned char*>((W).A) + (size_t)(m0 + prow) * (W).lda * ES + lc;   \
        const unsigned char* srcB = reinterpret_cast<const unsigned char*>((W).B) + (size_t)(n0 + prow) * (W).ldb * ES + lc;   \
        const int nk = ((W).K * ES) / kTileBytesK;                                                                        \
        auto dma = [&](int kt, int buf) __attribute__((always_inline)) {                                                  \
            __builtin_amdgcn_global_load_lds((gp_t)(srcA + (size_t)kt * kTileBytesK), (lp_t)(smem + buf * kStage + wave * 1024), 16, 0, AUX);       \
            __builtin_amdgcn_global_load_lds((gp_t)(srcB + (size_t)kt * kTileBytesK), (lp_t)(smem + buf * kStage + (4 + wave) * 1024), 16, 0, 0); };  \
        const int npro = nk < RING - 1 ? nk : RING - 1;                                                                   \
        for (int p = 0; p < npro; ++p) dma(p, p);                                                                         \
        acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                                                  \
        int buf = 0;                                                                                                      \
        for (int kt = 0; kt < nk; ++kt) {                                                                                 \
            const int rem = nk - 1 - kt;                                                                                  \
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                \
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                           \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
            asm volatile("s_barrier" ::: "memory");                                                                       \
            const unsigned char* Sb = smem + buf * kStage;                                                                 \
            const u32x4 a0 = *reinterpret_cast<const u32x4*>(Sb + aoff + sw0), b0 = *reinterpret_cast<const u32x4*>(Sb + boff + sw0);   \
            const int fill = buf == 0 ? RING - 1 : buf - 1;                                                               \
            if (kt + RING - 1 < nk) dma(kt + RING - 1, fill);                                                             \
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(Sb + aoff + (sw0 ^ 64)), b1 = *reinterpret_cast<const u32x4*>(Sb + boff + (sw0 ^ 64)); \
            mma<CT>(b0, a0, acc);                                                                                         \
            mma<CT>(b1, a1, acc);                                                                                         \
            buf = buf + 1 == RING ? 0 : buf + 1;                                                                          \
        }                                                                                                                 \
    }
    // ---- link 1
    AVAE_C_LOOP(w1, 0)
    if (orow < w1.M && ocol < w1.N) {
        CT* dst = reinterpret_cast<CT*>(w1.out0) + (size_t)orow * w1.ld0 + ocol;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fwd_t<ACT>(acc[e]);
        if (w1.N - ocol >= 4) {          // the lane's 4 results as one write-through store (8 bytes of bf16, 16 of fp32)
            if constexpr (ES == 2) {
                const unsigned long long pk = (unsigned long long)pack_bf16(v[0], v[1]) | ((unsigned long long)pack_bf16(v[2], v[3]) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), __builtin_bit_cast(unsigned long long, float2{v[0], v[1]}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, __builtin_bit_cast(unsigned long long, float2{v[2], v[3]}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            for (int e = 0; e < w1.N - ocol; ++e) {
                if constexpr (ES == 2) __hip_atomic_store(reinterpret_cast<unsigned short*>(dst) + e, __builtin_bit_cast(unsigned short, to_ct<CT>(v[e])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(reinterpret_cast<float*>(dst) + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- hand-off: every storing wave drains, the workgroup meets, one lane takes the ticket and polls for the row block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = counters + blockIdx.y * 64 + tm;           // (<= 64 row blocks per modality: the host checks)
        const unsigned n = (unsigned)w1.tiles_n;
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / n + 1u) * n;
        int spins = 0;
        while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // never hang: flag it and go on
        }
    }
    __syncthreads();
    // ---- link 2 on the handed-off rows (A = layer 1's output: sc1 on its LDS-DMA loads)
    AVAE_C_LOOP(w2, 16)
#undef AVAE_C_LOOP
    if (orow < w2.M && ocol < w2.N) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fwd_t<ACT>(acc[e]);
        store_row<CT>(reinterpret_cast<CT*>(w2.out0) + (size_t)orow * w2.ld0 + ocol, v, w2.N - ocol);
    }
}
template <typename CT> static void launch_chain2_act(int act, const LaunchArgs& args, dim3 grid, int lds, unsigned* counters, unsigned* err, hipStream_t s) {
    const dim3 block(kThreads);
    switch (act) {
    case AVAE_ACT_RELU: AVAE_LAUNCH((k_chain2<CT, AVAE_ACT_RELU>), grid, block, lds, s, args, counters, err); break;
    case AVAE_ACT_SOFTPLUS: AVAE_LAUNCH((k_chain2<CT, AVAE_ACT_SOFTPLUS>), grid, block, lds, s, args, counters, err); break;
    case AVAE_ACT_SIGMOID: AVAE_LAUNCH((k_chain2<CT, AVAE_ACT_SIGMOID>), grid, block, lds, s, args, counters, err); break;
    case AVAE_ACT_TANH: AVAE_LAUNCH((k_chain2<CT, AVAE_ACT_TANH>), grid, block, lds, s, args, counters, err); break;
    default: AVAE_LAUNCH((k_chain2<CT, AVAE_ACT_IDENTITY>), grid, block, lds, s, args, counters, err); break;
    }
}
void launch_chain2(int compute_dtype, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes, unsigned* counters, unsigned* err, hipStream_t s) {
    const dim3 grid(grid_x, grid_y);
    if (compute_dtype == AVAE_BF16) launch_chain2_act<__bf16>(args.items[0].act, args, grid, lds_bytes, counters, err, s);
    else launch_chain2_act<float>(args.items[0].act, args, grid, lds_bytes, counters, err, s);
}

// Refill-issuing waves of the 8-wave K-major (weight-gradient) tiles (see k_grouped).  Measured on C4: wgrad 143 -> 135 us with 4
// producers; the NT tiles (hidden forward / dgrad) lost 1.5-3 us per launch with them (three waves per SIMD cap the kernel at
// 168 VGPRs -- the register epilogue spills -- and their loop is paced by the arrival of the tiles either way), so they keep
// issuing their own refills.
constexpr int kProducers = 4;

static void grouped_attrs_once() {
    static const bool once = [] {
        set_max_lds(k_grouped<__bf16, 64, 64, 4>); set_max_lds(k_grouped<__bf16, 128, 128, 2>); set_max_lds(k_grouped<__bf16, 256, 128, 3, 8, false, 0>);
        set_max_lds(k_grouped<float, 64, 64, 4>); set_max_lds(k_grouped<float, 128, 128, 2>); set_max_lds(k_grouped<float, 256, 128, 3, 8, false, 0>);
        set_max_lds(k_grouped<__bf16, 32, 64, 4>); set_max_lds(k_grouped<float, 32, 64, 4>);
        set_max_lds(k_grouped<__bf16, 64, 128, 4>); set_max_lds(k_grouped<float, 64, 128, 4>);
        set_max_lds(k_grouped<__bf16, 32, 32, 4>); set_max_lds(k_grouped<float, 32, 32, 4>);
        set_max_lds(k_grouped<__bf16, 256, 64, 3, 8, false, 0>); set_max_lds(k_grouped<float, 256, 64, 3, 8, false, 0>);
        set_max_lds(k_grouped<__bf16, 64, 64, 4, 4, true>); set_max_lds(k_grouped<__bf16, 128, 128, 2, 4, true>); set_max_lds(k_grouped<__bf16, 256, 128, 3, 8, true, kProducers>);
        set_max_lds(k_grouped<float, 64, 64, 4, 4, true>); set_max_lds(k_grouped<float, 128, 128, 2, 4, true>); set_max_lds(k_grouped<float, 256, 128, 3, 8, true, kProducers>);
        return true;
    }();
    (void)once;
}

#define AVAE_GO(CT, TNF)                                                                                                     \
    do {                                                                                                                     \
        if (tile_cfg == 0) AVAE_LAUNCH((k_grouped<CT, 64, 64, 4, 4, TNF>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);        \
        else if (tile_cfg == 2) AVAE_LAUNCH((k_grouped<CT, 256, 128, 3, 8, TNF, (TNF ? kProducers : 0)>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id); \
        else AVAE_LAUNCH((k_grouped<CT, 128, 128, 2, 4, TNF>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);                    \
    } while (0)

void launch_grouped(int compute_dtype, int tile_cfg, const LaunchArgs& args, int grid_x, int grid_y, int lds_bytes,
                    DevState* st, hipStream_t s, unsigned long long* stamps, int launch_id) {
    grouped_attrs_once();
    dim3 grid(grid_x, grid_y), block(tile_cfg == 2 || tile_cfg == 6 ? 512 : kThreads);
    if (tile_cfg == 6) {
        if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_grouped<__bf16, 256, 64, 3, 8, false, 0>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
        else AVAE_LAUNCH((k_grouped<float, 256, 64, 3, 8, false, 0>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
    } else if (tile_cfg == 3) {
        if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_grouped<__bf16, 32, 64, 4>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
        else AVAE_LAUNCH((k_grouped<float, 32, 64, 4>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
    } else if (tile_cfg == 5) {
        if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_grouped<__bf16, 32, 32, 4>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
        else AVAE_LAUNCH((k_grouped<float, 32, 32, 4>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
    } else if (tile_cfg == 4) {
        if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_grouped<__bf16, 64, 128, 4>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
        else AVAE_LAUNCH((k_grouped<float, 64, 128, 4>), grid, block, lds_bytes, s, args, st, lds_bytes, stamps, launch_id);
    } else if (compute_dtype == AVAE_BF16) AVAE_GO(__bf16, false);
    else AVAE_GO(float, false);
}

// The weight-gradient launches: K-major operands, compact items (tile shapes 0, 1, 2 only).
void launch_grouped_tn(int compute_dtype, int tile_cfg, const TnLaunchArgs& args, int grid_x, int grid_y, int lds_bytes,
                       DevState* st, hipStream_t s, unsigned long long* stamps, int launch_id) {
    grouped_attrs_once();
    dim3 grid(grid_x, grid_y), block(tile_cfg == 2 ? (8 + kProducers) * 64 : kThreads);
    if (compute_dtype == AVAE_BF16) AVAE_GO(__bf16, true);
    else AVAE_GO(float, true);
}
#undef AVAE_GO

// ------------------------------------------------------------------ Adam + shadow refresh
// TF-1 AdamOptimizer dense update (reference vae_assoc.py:373-374; TF training_ops ApplyAdam):
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m += (g-m)(1-b1);  v += (g^2-v)(1-b2);
//   theta -= lr_t*m/(sqrt(v)+eps)                       (epsilon outside the bias correction)
// Tile-wise over each layer's [in+1][out] matrix so the same pass emits the compute-dtype shadow
// W (row-major, dgrad operand) and, through an LDS transpose, W^T (forward operand).
template <typename CT, int TR>
__global__ void __launch_bounds__(kThreads) k_adam(AdamArgs a) {
    static_assert(TR % 16 == 0 && TR <= 64, "a pass covers 16 rows x 16 quads");
    constexpr int NTH = kThreads;
    __shared__ float T[TR][65];
    // Tiles are taken LAST FIRST: the launch streams 32 bytes per parameter through the Infinity Cache (256 MiB), so what it
    // touches last is what the next step's first launches find there -- the encoders' weight shadows, not the decoders'.
    const int bid = (int)gridDim.x - 1 - (int)blockIdx.x, tid = threadIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_items; ++i)
        if (bid >= a.base[i]) it = i;                       // kernel-argument table, ascending
    const AdamItem w = a.items[it];
    const int t = bid - w.tile_base;
    const int tr = t / w.tiles_c, tc = t - tr * w.tiles_c;
    const int r0 = tr * TR, c0 = tc * 64;

    if (a.mode == 0 && a.book && tid == 0 && bid == 0) {    // step already counts this update (bumped by K_COST)
        const float c = *a.cost_src;
        a.st->last_cost = c;
        a.st->cost_hist[(a.st->step - 1) % kCostHist] = c;
    }
    const float lr_t = a.mode == 0 ? a.st->lr_t : 0.0f;
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;

    const int c4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < TR / (NTH / 16); ++i) {
        const int r = (tid >> 4) + (NTH / 16) * i;
        const int grow = r0 + r, gcol = c0 + c4;
        float th[4] = {0.f, 0.f, 0.f, 0.f};
        if (grow < w.rows && gcol < w.cols) {
            const size_t off = (size_t)grow * w.ld + gcol;   // ld % 4 == 0: the quad is in-row and 16-B aligned
            load4<float>(w.theta + off, th);
            if (a.mode == 0) {
                float g[4], m[4], v[4];
                load4<float>(w.g + off, g);
                load4<float>(w.m + off, m);
                load4<float>(w.v + off, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) adam_update(g[e], m[e], v[e], th[e], omb1, omb2, lr_t, a.eps);
                const int nv = w.cols - gcol;
                store_row<float>(w.theta + off, th, nv);
                store_row<float>(w.m + off, m, nv);
                store_row<float>(w.v + off, v, nv);
            }
            store_row<CT>(reinterpret_cast<CT*>(w.W) + (size_t)grow * w.ldw + gcol, th, w.cols - gcol);
            if (w.adj_k > 0 && grow < w.adj_k * w.adj_k * w.adj_cin) {       // (not the bias row)
                const int kp = grow / w.adj_cin, ci = grow - kp * w.adj_cin, kh = w.adj_k - 1 - kp / w.adj_k, kw = w.adj_k - 1 - kp % w.adj_k;
                const int r = (kh * w.adj_k + kw) * w.cols + gcol;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gcol + e < w.cols) {
                        const CT v = to_ct<CT>(th[e]);
                        reinterpret_cast<CT*>(w.Wadj)[(size_t)ci * w.ldadj + r + e] = v;
                        if (w.Wf) reinterpret_cast<CT*>(w.Wf)[(size_t)(r + e) * w.ldf + ci] = v;
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r][c4 + e] = th[e];
    }
    lds_barrier();
    constexpr int RL = TR / 4;                              // lanes along the rows of the transposed store
    const int r4 = (tid % RL) * 4;
#pragma unroll
    for (int i = 0; i < 64 / (NTH / RL); ++i) {
        const int c = tid / RL + (NTH / RL) * i;
        const int gcol = c0 + c, grow = r0 + r4;
        if (gcol < w.cols && grow < w.rows) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = T[r4 + e][c];
            store_row<CT>(reinterpret_cast<CT*>(w.Wt) + (size_t)gcol * w.ldt + grow, v, w.rows - grow);
        }
    }
}

void launch_adam(int compute_dtype, const AdamArgs& a, int n_blocks, hipStream_t s) {
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_adam<__bf16, kAdamRows>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_adam<float, kAdamRows>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}

// ------------------------------------------------------------------ input staging + eps
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename CT>
__global__ void __launch_bounds__(kThreads) k_prep(PrepArgs a) {
    const int tid = threadIdx.x;
    const int bstep = a.n_steps > 1 ? (int)blockIdx.x / a.blocks_per_step : 0;     // which of the batched steps
    const int bid = (int)blockIdx.x - bstep * a.blocks_per_step;
    const size_t set_off = (size_t)bstep * (size_t)a.set_stride;                      // bytes to that step's staging set
    if (bid >= a.total_tiles) {
        // eps: one quad of dims per thread
        if (!a.eps_dst) return;
        const int nq = (a.nz + 3) / 4;
        const int q = (bid - a.total_tiles) * kThreads + tid;
        if (q >= a.eps_rows * nq) return;
        const int row = q / nq, d4 = q - row * nq;
        float n[4];
        if (a.eps_src) {
#pragma unroll
            for (int e = 0; e < 4; ++e)      // caller's eps is dense [rows][n_z], batches back to back
                n[e] = (4 * d4 + e < a.nz) ? a.eps_src[((size_t)bstep * a.eps_rows + row) * a.nz + 4 * d4 + e] : 0.0f;
        } else {
            const unsigned long long step = (unsigned long long)a.st->step + (unsigned long long)bstep;
            unsigned r[4];
            // counter = (global row, dim quad ^ salt.hi << 8, step lo, step hi ^ salt.lo): the salt's low word names the stream
            // (train / eval / reconstruct), its high word the draw (per-handle draw counter, modality) -- 0 for training
            philox4x32_10((unsigned)(a.row_offset + row), (unsigned)d4 ^ ((unsigned)(a.stream_salt >> 32) << 8), (unsigned)step,
                          (unsigned)(step >> 32) ^ (unsigned)a.stream_salt,
                          (unsigned)a.seed, (unsigned)(a.seed >> 32), r);
            // Box-Muller on (0,1) uniforms built from the top 24 bits
            const float u0 = ((r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float ra = sqrtf(-2.0f * flog(u0)), rb = sqrtf(-2.0f * flog(u2));
            // v_sin_f32 / v_cos_f32 take their argument in revolutions: sin(2*pi*u) directly
            n[0] = ra * __builtin_amdgcn_cosf(u1); n[1] = ra * __builtin_amdgcn_sinf(u1);
            n[2] = rb * __builtin_amdgcn_cosf(u3); n[3] = rb * __builtin_amdgcn_sinf(u3);
        }
        float* eps_dst = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(a.eps_dst) + set_off);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * d4 + e < a.nz) eps_dst[(size_t)row * a.eps_ld + 4 * d4 + e] = n[e];
        return;
    }
    int it = 0;
    while (it + 1 < a.n_seg && bid >= a.seg[it + 1].tile_base) ++it;
    const PrepSeg& w = a.seg[it];
    const int t = bid - w.tile_base;
    const int tr = t / w.tiles_c, tc = t - tr * w.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;
    const int c4 = (tid & 15) * 4;
    const float* src = w.src + (size_t)bstep * w.rows * w.src_ld;
    float* dst32 = w.dst32 ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(w.dst32) + set_off) : nullptr;
    CT* dstc = reinterpret_cast<CT*>(reinterpret_cast<unsigned char*>(w.dstc) + set_off);
    // the caller's rows as one 16-byte load per quad whatever their alignment (column blocks of a [rows][931] matrix start anywhere)
    typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // dword-aligned quad: one global_load_dwordx4 (the
                                                                                  // hardware takes any dword alignment in global memory)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i;
        const int grow = r0 + r, gcol = c0 + c4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (grow < w.rows && gcol < w.cols) {
            if (gcol + 3 < w.cols) {
                const f32x4_a4 q = *reinterpret_cast<const f32x4_a4*>(src + (size_t)grow * w.src_ld + gcol);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gcol + e < w.cols) v[e] = src[(size_t)grow * w.src_ld + gcol + e];
            }
            const int nv = w.cols - gcol;
            if (dst32) store_row<float>(dst32 + (size_t)grow * w.ld32 + gcol, v, nv);
            store_row<CT>(dstc + (size_t)grow * w.ldc + gcol, v, nv);
        }
    }
}

const void* prep_kernel(int compute_dtype) {
    return compute_dtype == AVAE_BF16 ? reinterpret_cast<const void*>(k_prep<__bf16>) : reinterpret_cast<const void*>(k_prep<float>);
}

void launch_prep(int compute_dtype, const PrepArgs& a, hipStream_t s) {
    const int n_blocks = (a.total_tiles + a.eps_blocks) * (a.n_steps > 1 ? a.n_steps : 1);
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_prep<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_prep<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}

// ------------------------------------------------------------------ conv branch: im2col / col2im
template <typename CT> __device__ __forceinline__ float ct_load(const CT* p);
template <> __device__ __forceinline__ float ct_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ct_load<__bf16>(const __bf16* p) { return (float)*p; }

// Patch matrix of one conv-like layer, 64 x 64 tiles (rows = output pixels, cols = (kh,kw,ci) [+ ones]).
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_gather(GatherArgs a) {
    const int bid = blockIdx.x, tid = threadIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].tile_base) it = i;
    const GatherSeg& w = a.seg[it];
    const ConvGeom g = w.g;
    const int t = bid - w.tile_base;
    const int tr = t / w.tiles_c, tc = t - tr * w.tiles_c;
    const int rstep = kThreads / w.cl;                      // rows per pass
    const int r0 = tr * rstep * w.rpt, c0 = tc * 4 * w.cl;
    const int M = g.B * g.OH * g.OW, K = g.k * g.k * g.Cin, KC = K + (g.ones ? 1 : 0);
    const CT* src = reinterpret_cast<const CT*>(w.src);
    const int c4 = (tid & (w.cl - 1)) * 4;
    const int kc = c0 + c4;
    // The thread's 4 consecutive patch columns: with Cin % 4 == 0 they are 4 channels of ONE filter position (kh, kw), so the
    // position is decoded once per thread (not per element and row) and the 4 channels are one vector load; the dilation d is 1 or 2
    // (mask / shift instead of % and /).  The scalar path below keeps single-channel inputs (first conv stage) and odd shapes.
    const bool quad = (g.Cin & 3) == 0 && (g.d == 1 || g.d == 2) && kc < K;
    const int kpos = quad ? kc / g.Cin : 0, ci0 = quad ? kc - kpos * g.Cin : 0, kh0 = kpos / g.k, kw0 = kpos - kh0 * g.k;
    const int dmask = g.d - 1, dshift = g.d >> 1;
    // scalar path (single-channel first conv stage, odd channel counts): the four columns' (kh, kw, ci) are decoded ONCE per thread
    // as well -- they used to be decoded per element AND row (conv_enc1_im2col: 9.4 us for 3.2 MB)
    const bool pow2d = g.d == 1 || g.d == 2;
    int ekh[4], ekw[4], eci[4];
    if (!quad) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kidx = kc + e, kp = kidx / g.Cin;
            eci[e] = kidx - kp * g.Cin; ekh[e] = kp / g.k; ekw[e] = kp - ekh[e] * g.k;
        }
    }
    auto row_pass = [&](int i) {
        const int r = tid / w.cl + rstep * i;
        const int m = r0 + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < M && kc < KC) {
            const int b = m / (g.OH * g.OW), rem = m - b * g.OH * g.OW, oh = rem / g.OW, ow = rem - oh * g.OW;
            if (quad) {
                const int nh = oh * g.so + kh0 - g.pad, nw = ow * g.so + kw0 - g.pad;
                if (nh >= 0 && nw >= 0 && !(nh & dmask) && !(nw & dmask)) {
                    const int ih = nh >> dshift, iw = nw >> dshift;
                    if (ih < g.IH && iw < g.IW)
                        load4<CT>(src + (size_t)b * g.src_sb + (size_t)(ih * g.IW + iw) * g.src_sp + ci0, v);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kidx = kc + e;
                    if (kidx < K) {
                        const int ci = eci[e], kh = ekh[e], kw = ekw[e];
                        const int nh = oh * g.so + kh - g.pad, nw = ow * g.so + kw - g.pad;
                        if (nh >= 0 && nw >= 0 && (pow2d ? !((nh | nw) & dmask) : (nh % g.d == 0 && nw % g.d == 0))) {
                            const int ih = pow2d ? nh >> dshift : nh / g.d, iw = pow2d ? nw >> dshift : nw / g.d;
                            if (ih < g.IH && iw < g.IW)
                                v[e] = ct_load<CT>(src + (size_t)b * g.src_sb + (size_t)(ih * g.IW + iw) * g.src_sp + ci);
                        }
                    } else if (kidx == K && g.ones) {
                        v[e] = 1.0f;
                    }
                }
            }
            store_row<CT>(reinterpret_cast<CT*>(w.P) + (size_t)m * w.ldp + kc, v, KC - kc);
        }
    };
    if (w.rpt == 4) {        // unrolled: the four rows' loads are in flight together
#pragma unroll
        for (int i = 0; i < 4; ++i) row_pass(i);
    } else {
        row_pass(0);
    }
}

template <typename CT>
__global__ void __launch_bounds__(kThreads) k_col2im(Col2imArgs a) {
    const int bid = blockIdx.x, tid = threadIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].tile_base) it = i;
    const Col2imSeg& w = a.seg[it];
    const ConvGeom g = w.g;
    const int t = bid - w.tile_base;
    const int tr = t / w.tiles_c, tc = t - tr * w.tiles_c;
    const int rstep = kThreads / w.cl;                      // rows per pass
    const int r0 = tr * rstep * w.rpt, c0 = tc * 4 * w.cl;
    const int R = g.B * g.IH * g.IW;                      // input pixels
    const bool latent = w.g0 != nullptr;
    const int C = latent ? 2 * w.nz : g.Cin;              // output columns
    const int c4 = (tid & (w.cl - 1)) * 4;
    auto row_pass = [&](int i) {
        const int r = tid / w.cl + rstep * i;
        const int pix = r0 + r, cc = c0 + c4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (pix < R && cc < C) {
            const int b = pix / (g.IH * g.IW), rem = pix - b * g.IH * g.IW, ih = rem / g.IW, iw = rem - ih * g.IW;
            // Fast path: the thread's 4 columns are 4 channels of one pixel (not straddling the mu | lv halves of the latent
            // mode), so the k x k walk over the patch gradients is done ONCE for all four with one 16-byte load per position; the
            // output stride so is 1 or 2 (mask / shift instead of % and /).  Each dP element is read exactly once overall.
            // taps that reach this pixel: output position o = (i*d + pad - k)/so must be an integer in [0, O), i.e.
            // k in [i*d + pad - (O-1)*so, i*d + pad] within [0, k), in steps of so from the right residue -- the walk visits only
            // those (the flatten stage, k = 28, has exactly one)
            const int th = ih * g.d + g.pad, tw = iw * g.d + g.pad;
            int kh_lo = max(0, th - (g.OH - 1) * g.so), kw_lo = max(0, tw - (g.OW - 1) * g.so);
            kh_lo += (th - kh_lo) % g.so;                 // th >= kh_lo >= 0
            kw_lo += (tw - kw_lo) % g.so;
            const int kh_hi = min(g.k - 1, th), kw_hi = min(g.k - 1, tw);
            const bool quad = (g.Cin & 3) == 0 && (g.so == 1 || g.so == 2) && cc + 3 < C && (!latent || (w.nz & 3) == 0);
            if (quad) {
                const int ci0 = latent ? (cc < w.nz ? cc : cc - w.nz) : cc;
                const int smask = g.so - 1, sshift = g.so >> 1;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int kh = kh_lo; kh <= kh_hi; kh += g.so) {
                    const int oh = (th - kh) >> sshift;
                    for (int kw = kw_lo; kw <= kw_hi; kw += g.so) {
                        const int ow = (tw - kw) >> sshift;
                        float t[4];
                        load4<float>(w.dP + (size_t)((b * g.OH + oh) * g.OW + ow) * w.lddp + (kh * g.k + kw) * g.Cin + ci0, t);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] += t[e];
                    }
                }
                if (w.bias) {        // forward mode: overlap-add of a scatter product, + bias, transfer function
                    const CT* bs = reinterpret_cast<const CT*>(w.bias);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = act_fwd(w.act, acc[e] + ct_load<CT>(bs + (size_t)(ci0 + e) * w.bias_ld));
                } else if (latent) { // dmu = dz + g0mu ; dlv = dz * F + g0lv   (g0 = [g0mu | g0lv | F], reparameterisation)
                    const int lde = (w.nz + 3) & ~3;
                    const float* gr = w.g0 + (size_t)pix * 3 * lde;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = cc < w.nz ? acc[e] + gr[ci0 + e] : acc[e] * gr[2 * lde + ci0 + e] + gr[lde + ci0 + e];
                } else if (w.yprev) {
                    float y[4];
                    load4<CT>(reinterpret_cast<const CT*>(w.yprev) + (size_t)pix * w.ldy + ci0, y);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] *= act_bwd(w.act, y[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[e];
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = cc + e;
                if (col >= C) continue;
                const int ci = latent ? (col < w.nz ? col : col - w.nz) : col;
                float acc = 0.0f;
                for (int kh = kh_lo; kh <= kh_hi; kh += g.so) {
                    const int oh = (th - kh) / g.so;
                    for (int kw = kw_lo; kw <= kw_hi; kw += g.so) {
                        const int ow = (tw - kw) / g.so;
                        acc += w.dP[(size_t)((b * g.OH + oh) * g.OW + ow) * w.lddp + (kh * g.k + kw) * g.Cin + ci];
                    }
                }
                if (w.bias) {
                    acc = act_fwd(w.act, acc + ct_load<CT>(reinterpret_cast<const CT*>(w.bias) + (size_t)ci * w.bias_ld));
                } else if (latent) { // dmu = dz + g0mu ; dlv = dz * F + g0lv   (g0 = [g0mu | g0lv | F], reparameterisation)
                    const int lde = (w.nz + 3) & ~3;                      // g0 row = [g0mu | g0lv | F], each roundup(n_z, 4) wide
                    const float* gr = w.g0 + (size_t)pix * 3 * lde;
                    acc = col < w.nz ? acc + gr[ci] : acc * gr[2 * lde + ci] + gr[lde + ci];
                } else if (w.yprev) {
                    acc *= act_bwd(w.act, ct_load<CT>(reinterpret_cast<const CT*>(w.yprev) + (size_t)pix * w.ldy + ci));
                }
                v[e] = acc;
            }
            }
            store_row<CT>(reinterpret_cast<CT*>(w.dA) + (size_t)pix * w.lda + cc, v, C - cc);
        }
    };
    if (w.rpt == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) row_pass(i);
    } else {
        row_pass(0);
    }
}

void launch_gather(int compute_dtype, const GatherArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_gather<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_gather<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}
void launch_col2im(int compute_dtype, const Col2imArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_col2im<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_col2im<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}

// ------------------------------------------------------------------ adjoint filter shadows (transposed-conv input gradients)
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_wadj(WadjArgs a) {
    const int bid = blockIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    const WadjSeg& w = a.seg[it];
    const int KA = w.k * w.k * w.Cout;
    const int i = (bid - w.block_base) * kThreads + (int)threadIdx.x;
    if (i >= w.Cin * KA) return;
    const int ci = i / KA, r = i - ci * KA, kp = r / w.Cout, co = r - kp * w.Cout, kh = kp / w.k, kw = kp - kh * w.k;
    const CT* Wt = reinterpret_cast<const CT*>(w.Wt);
    const CT v = Wt[(size_t)co * w.ldt + ((w.k - 1 - kh) * w.k + (w.k - 1 - kw)) * w.Cin + ci];
    reinterpret_cast<CT*>(w.Wadj)[(size_t)ci * w.ldadj + r] = v;
    reinterpret_cast<CT*>(w.Wf)[(size_t)r * w.ldf + ci] = v;
}

__global__ void __launch_bounds__(kThreads) k_gperm(GpermArgs a) {
    const int bid = blockIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    const GpermSeg& w = a.seg[it];
    const int KA = w.k * w.k * w.Cout;
    const int i = (bid - w.block_base) * kThreads + (int)threadIdx.x;
    if (i >= w.Cin * KA) return;
    const int ci = i / KA, r = i - ci * KA, kp = r / w.Cout, co = r - kp * w.Cout, kh = kp / w.k, kw = kp - kh * w.k;
    w.G[(size_t)(((w.k - 1 - kh) * w.k + (w.k - 1 - kw)) * w.Cin + ci) * w.ld + co] = w.Gadj[(size_t)ci * w.ldga + r];
}

void launch_gperm(const GpermArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    AVAE_LAUNCH(k_gperm, dim3(n_blocks), dim3(kThreads), 0, s, a);
}

template <typename CT>
__global__ void __launch_bounds__(kThreads) k_rowsum(RowsumArgs a) {
    __shared__ float red[kThreads];
    const int bid = blockIdx.x, tid = threadIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    const RowsumSeg& w = a.seg[it];
    const int j = bid - w.block_base;
    // lanes: column c = tid % cp (cp = the power of two >= cols, <= 64), row group q = tid / cp of nq = 256 / cp;
    // rows j + n_blocks * (q + nq * m), four loads in flight
    const int cp = w.cpow, nq = kThreads / cp;
    const int c = tid & (cp - 1), q = tid / cp;
    const CT* src = reinterpret_cast<const CT*>(w.src);
    float acc = 0.0f;
    if (c < w.cols) {
        const int step = w.n_blocks * nq;
        for (int r = j + w.n_blocks * q; r < w.rows; r += 4 * step) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = r + u * step < w.rows ? ct_load<CT>(src + (size_t)(r + u * step) * w.ld + c) : 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    }
    red[tid] = acc;
    lds_barrier();
    if (q == 0 && c < w.cols4) {
        float t = 0.0f;
        for (int k = 0; k < nq; ++k) t += red[k * cp + c];
        w.part[(size_t)j * w.cols4 + c] = c < w.cols ? t : 0.0f;
    }
}

void launch_rowsum(int compute_dtype, const RowsumArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_rowsum<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_rowsum<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}

void launch_wadj(int compute_dtype, const WadjArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_wadj<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_wadj<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}

// ------------------------------------------------------------------ single-output-channel transposed-conv stage, direct
// n / q for the two strides the branch uses (1, 2) without an integer division; `ok` = divisible
__device__ __forceinline__ int div_small(int n, int q, bool& ok) {
    if (q == 1) { ok = true; return n; }
    if (q == 2) { ok = !(n & 1); return n >> 1; }
    ok = n % q == 0;
    return n / q;
}

// non-negative remainder
__device__ __forceinline__ int mod_small(int n, int q) {
    if (q == 1) return 0;
    if (q == 2) return n & 1;
    const int r = n % q;
    return r < 0 ? r + q : r;
}

// The reference's last decoder stage (5x5 transposed conv, stride 2, TF-SAME crop 1 / 2, 14x14xCin -> 28x28x1; gather form: ih =
// (oh + kh - 3) / 2), block-structured, ONE workgroup per image (ThinSeg::fast): the generic loops below spend most of their
// instructions on tap arithmetic and, split four ways, stage every image four times.
//   forward   output block (2i + dh, 2j + dw), dh, dw in {0, 1}, reads input pixels (i - 1 + a, j - 1 + b), a, b in {0, 1, 2}, through
//             tap kh = 2a + 1 - dh (kw alike) where that is < 5: 25 (pixel, output) pairs per block, known at compile time
//   dX        input pixel (ih, iw) collects dY[2 ih + 3 - kh][2 iw + 3 - kw] F[kh][kw][c] over the 5x5 taps (same order as the generic loop)
//   dF        thread (tap, s) adds X[p][:] dY[...] over pixels p = s, s + 10, ...; the ten partial sums of a tap are added in order
template <typename CT, int CQ, int MODE>
__device__ __forceinline__ void thin_fast_body(const ThinSeg& w, const ConvGeom& g, int b, int sub, int bid, const CT* X,
                                               float* sx, const float* sf, const float* sdy, float* sred) {
    constexpr int mode = MODE;
    constexpr int Cin = 4 * CQ, K = 25 * Cin, S = 10;
    const int tid = threadIdx.x, IH = g.IH, IW = g.IW, OH = g.OH, OW = g.OW, npi = IH * IW, npo = OH * OW;
    if constexpr (mode == 0) {
        CT* Y = reinterpret_cast<CT*>(w.Y) + (size_t)b * (w.img_y > 0 ? w.img_y : npo * w.ldy);
        const float bias = sf[K];
        const int per = (npi + w.split - 1) / w.split, blk_end = min(npi, (sub + 1) * per);
        for (int blk = sub * per + tid; blk < blk_end; blk += kThreads) {
            const int i0 = blk / IW, j0 = blk - i0 * IW;
            float acc[4] = {bias, bias, bias, bias};
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) {
                const int ih = i0 - 1 + aa;
                if (ih < 0 || ih >= IH) continue;
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    const int iw = j0 - 1 + bb;
                    if (iw < 0 || iw >= IW) continue;
                    const float* xr = sx + (ih * IW + iw) * Cin;
#pragma unroll
                    for (int cq = 0; cq < CQ; ++cq) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(xr + 4 * cq);
#pragma unroll
                        for (int dh = 0; dh < 2; ++dh) {
                            const int kh = 2 * aa + 1 - dh;
                            if (kh >= 5) continue;
#pragma unroll
                            for (int dw = 0; dw < 2; ++dw) {
                                const int kw = 2 * bb + 1 - dw;
                                if (kw >= 5) continue;
                                const f32x4 f = *reinterpret_cast<const f32x4*>(sf + (kh * 5 + kw) * Cin + 4 * cq);
                                float& o = acc[dh * 2 + dw];
                                o += x[0] * f[0]; o += x[1] * f[1]; o += x[2] * f[2]; o += x[3] * f[3];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw)
                    Y[(size_t)((2 * i0 + dh) * OW + 2 * j0 + dw) * w.ldy] = (CT)act_fwd(w.act, acc[dh * 2 + dw]);
        }
    } else if constexpr (mode == 1) {
        CT* dX = reinterpret_cast<CT*>(w.dX) + (size_t)b * npi * w.lddx;
        const int per = (npi * CQ + w.split - 1) / w.split, idx_end = min(npi * CQ, (sub + 1) * per);
        for (int idx = sub * per + tid; idx < idx_end; idx += kThreads) {
            const int pix = idx / CQ, c0 = (idx - pix * CQ) * 4;
            const int ih = pix / IW, iw = pix - ih * IW;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 5; ++kh) {
                const int oh = 2 * ih + 3 - kh;
                if ((unsigned)oh >= (unsigned)OH) continue;
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) {
                    const int ow = 2 * iw + 3 - kw;
                    if ((unsigned)ow >= (unsigned)OW) continue;
                    const float dy = sdy[oh * OW + ow];
                    const f32x4 f = *reinterpret_cast<const f32x4*>(sf + (kh * 5 + kw) * Cin + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += dy * f[e];
                }
            }
            float xv[4];
            load4<CT>(X + (size_t)pix * g.src_sp + c0, xv);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= act_bwd(w.act_in, xv[e]);
            store4<CT>(dX + (size_t)pix * w.lddx + c0, acc);
        }
    } else {
        if (tid < 64) {                                     // bias gradient: this workgroup's share of the map's sum, 64 strided partial sums added in order below
            const int per = (npo + w.split - 1) / w.split, e = min(npo, (sub + 1) * per);
            float sacc = 0.0f;
            for (int i = sub * per + tid; i < e; i += 64) sacc += sdy[i];
            sred[tid] = sacc;
        }
        const int pper = (npi + w.split - 1) / w.split, p_end = min(npi, (sub + 1) * pper);      // this workgroup's share of the input pixels
        const int tap = tid / S, s = tid - tap * S;
        f32x4 acc[CQ];
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) acc[cq] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tap < 25) {
            const int kh = tap / 5, kw = tap - kh * 5;
            for (int p = sub * pper + s; p < p_end; p += S) {
                const int ih = p / IW, iw = p - ih * IW;
                const int oh = 2 * ih + 3 - kh, ow = 2 * iw + 3 - kw;
                if ((unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW) {
                    const float dy = sdy[oh * OW + ow];
#pragma unroll
                    for (int cq = 0; cq < CQ; ++cq) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(sx + p * Cin + 4 * cq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[cq][e] += dy * x[e];
                    }
                }
            }
        }
        lds_barrier();                                       // every read of the staged image is done: its space takes the partial sums
        if (tap < 25) {
#pragma unroll
            for (int cq = 0; cq < CQ; ++cq) *reinterpret_cast<f32x4*>(sx + (tap * S + s) * Cin + 4 * cq) = acc[cq];
        }
        lds_barrier();
        for (int t = tid; t < w.Kp; t += kThreads) {
            float v = 0.0f;
            if (t < K) {
                const int tp = t / Cin, c = t - tp * Cin;
#pragma unroll
                for (int q = 0; q < S; ++q) v += sx[(tp * S + q) * Cin + c];
            } else if (t == K) {
                for (int i = 0; i < 64; ++i) v += sred[i];
            }
            w.part[(size_t)(bid - w.block_base) * w.Kp + t] = v;
        }
    }
}

// ... as kernels of their own (one per mode and channel count: each gets the registers its loops need, and the generic kernel below
// keeps its occupancy).  Every segment of the launch is on this path (host: launch_thin).
template <typename CT, int CQ, int MODE>
__global__ void __launch_bounds__(kThreads) k_thin_fast(ThinArgs a) {
    __shared__ __attribute__((aligned(16))) float sx[kThinIn];
    __shared__ __attribute__((aligned(16))) float sf[kThinF + 4];
    __shared__ float sdy[kThinOut];
    __shared__ float sred[64];
    constexpr int Cin = 4 * CQ;
    const int bid = blockIdx.x, tid = threadIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    const ThinSeg& w = a.seg[it];
    const ConvGeom g = w.g;
    const int b = (bid - w.block_base) / w.split, sub = (bid - w.block_base) % w.split;      // image, share of its work
    const CT* X = reinterpret_cast<const CT*>(w.X) + (size_t)b * g.src_sb;
    const int npi = g.IH * g.IW, npo = g.OH * g.OW;
    if constexpr (MODE != 1) {                            // input planes
        for (int i = tid; i < npi * CQ; i += kThreads) {
            const int px = i / CQ, c = (i - px * CQ) * 4;
            float v[4];
            load4<CT>(X + (size_t)px * g.src_sp + c, v);
            *reinterpret_cast<f32x4*>(sx + px * Cin + c) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
    if constexpr (MODE != 2) {                            // filter taps + bias
        const CT* F = reinterpret_cast<const CT*>(w.Wt);
        for (int i = tid; i < 25 * Cin + 1; i += kThreads) sf[i] = ct_load<CT>(F + i);
    }
    if constexpr (MODE != 0) {                            // output-gradient map
        const CT* dY = reinterpret_cast<const CT*>(w.dY) + (size_t)b * (w.img_dy > 0 ? w.img_dy : npo * w.lddy);
        for (int i = tid; i < npo; i += kThreads) sdy[i] = ct_load<CT>(dY + (size_t)i * w.lddy);
    }
    lds_barrier();
    thin_fast_body<CT, CQ, MODE>(w, g, b, sub, bid, X, sx, sf, sdy, sred);
}

// One workgroup per image in every mode: the image's input planes (IH*IW*Cin <= kThinIn), its output(-gradient) map
// (OH*OW <= kThinOut) and the filter (Kp <= kThinF) sit in LDS as fp32; products and sums are fp32 (operands were rounded to
// the compute type when they were stored, as on the GEMM path).
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_thin(ThinArgs a) {
    __shared__ __attribute__((aligned(16))) float sx[kThinIn];
    __shared__ __attribute__((aligned(16))) float sf[kThinF + 4];
    __shared__ float sdy[kThinOut];
    __shared__ float sred[64];
    const int bid = blockIdx.x, tid = threadIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    const ThinSeg& w = a.seg[it];
    const ConvGeom g = w.g;
    const int b = (bid - w.block_base) / w.split, sub = (bid - w.block_base) % w.split;     // image, share of its outputs (fast path: one workgroup per image)
    const CT* X = reinterpret_cast<const CT*>(w.X) + (size_t)b * g.src_sb;
    const CT* F = reinterpret_cast<const CT*>(w.Wt);
    const int K = g.k * g.k * g.Cin, Cin = g.Cin;
    const int npi = g.IH * g.IW, npo = g.OH * g.OW;
    if (a.mode != 1) {                                    // input planes
        if ((Cin & 3) == 0) {
            const int Q = Cin >> 2;
            for (int i = tid; i < npi * Q; i += kThreads) {
                const int px = i / Q, c = (i - px * Q) * 4;
                float v[4];
                load4<CT>(X + (size_t)px * g.src_sp + c, v);
                *reinterpret_cast<f32x4*>(sx + px * Cin + c) = f32x4{v[0], v[1], v[2], v[3]};
            }
        } else {
            for (int i = tid; i < npi * Cin; i += kThreads) {
                const int px = i / Cin, c = i - px * Cin;
                sx[i] = ct_load<CT>(X + (size_t)px * g.src_sp + c);
            }
        }
    }
    if (a.mode != 2)                                      // filter taps (+ bias at K)
        for (int i = tid; i < K + 1; i += kThreads) sf[i] = ct_load<CT>(F + i);
    if (a.mode != 0) {                                    // output-gradient map
        const CT* dY = reinterpret_cast<const CT*>(w.dY) + (size_t)b * (w.img_dy > 0 ? w.img_dy : npo * w.lddy);
        for (int i = tid; i < npo; i += kThreads) sdy[i] = ct_load<CT>(dY + (size_t)i * w.lddy);
    }
    lds_barrier();

    if (a.mode == 0) {
        // Y[oh, ow] = act(bias + sum over taps (kh, kw) with ih = (oh*so + kh - pad)/d an integer in [0, IH) (same for w))
        CT* Y = reinterpret_cast<CT*>(w.Y) + (size_t)b * (w.img_y > 0 ? w.img_y : npo * w.ldy);
        const int p_end = min(npo, (sub + 1) * ((npo + w.split - 1) / w.split));
        for (int p = sub * ((npo + w.split - 1) / w.split) + tid; p < p_end; p += kThreads) {
            const int oh = p / g.OW, ow = p - oh * g.OW;
            const int bh = oh * g.so - g.pad, bw = ow * g.so - g.pad;
            int kh0 = max(0, -bh), kw0 = max(0, -bw);
            kh0 += mod_small(-(bh + kh0), g.d);           // first tap whose source row exists: (bh + kh) divisible by d
            kw0 += mod_small(-(bw + kw0), g.d);
            bool ok_;
            float acc = g.ones ? sf[K] : 0.0f;
            for (int kh = kh0, ih = div_small(bh + kh0, g.d, ok_); kh < g.k && ih < g.IH; kh += g.d, ++ih)
                for (int kw = kw0, iw = div_small(bw + kw0, g.d, ok_); kw < g.k && iw < g.IW; kw += g.d, ++iw) {
                    const float* xr = sx + (ih * g.IW + iw) * Cin;
                    const float* fr = sf + (kh * g.k + kw) * Cin;
                    if ((Cin & 3) == 0) {
                        for (int c = 0; c < Cin; c += 4) {
                            const f32x4 x = *reinterpret_cast<const f32x4*>(xr + c), f = *reinterpret_cast<const f32x4*>(fr + c);
                            acc += x[0] * f[0]; acc += x[1] * f[1]; acc += x[2] * f[2]; acc += x[3] * f[3];
                        }
                    } else {
                        for (int c = 0; c < Cin; ++c) acc += xr[c] * fr[c];
                    }
                }
            Y[(size_t)p * w.ldy] = (CT)act_fwd(w.act, acc);
        }
    } else if (a.mode == 1) {
        // dX[ih, iw, c] = act_in'(X) * sum over taps with oh = (ih*d + pad - kh)/so an integer in [0, OH) of dY[oh, ow] * F[kh, kw, c]
        const int Q = (Cin + 3) >> 2;
        CT* dX = reinterpret_cast<CT*>(w.dX) + (size_t)b * npi * w.lddx;
        const int per = (npi * Q + w.split - 1) / w.split, i_end = min(npi * Q, (sub + 1) * per);
        for (int idx = sub * per + tid; idx < i_end; idx += kThreads) {
            const int pix = idx / Q, c0 = (idx - pix * Q) * 4;
            const int ih = pix / g.IW, iw = pix - ih * g.IW;
            const int th = ih * g.d + g.pad, tw = iw * g.d + g.pad;
            int kh_lo = max(0, th - (g.OH - 1) * g.so), kw_lo = max(0, tw - (g.OW - 1) * g.so);
            kh_lo += mod_small(th - kh_lo, g.so);
            kw_lo += mod_small(tw - kw_lo, g.so);
            const int kh_hi = min(g.k - 1, th), kw_hi = min(g.k - 1, tw);
            const int nc = min(4, Cin - c0);
            bool ok_;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int kh = kh_lo; kh <= kh_hi; kh += g.so) {
                const float* yr = sdy + div_small(th - kh, g.so, ok_) * g.OW;
                for (int kw = kw_lo; kw <= kw_hi; kw += g.so) {
                    const float dy = yr[div_small(tw - kw, g.so, ok_)];
                    const float* fr = sf + (kh * g.k + kw) * Cin + c0;     // (sf is padded: reading past nc is harmless)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += dy * fr[e];
                }
            }
            const CT* xp = X + (size_t)pix * g.src_sp + c0;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nc) acc[e] *= act_bwd(w.act_in, ct_load<CT>(xp + e));
            store_row<CT>(dX + (size_t)pix * w.lddx + c0, acc, nc);
        }
    } else {
        // dF[kh, kw, c] = sum_{ih, iw} X[ih, iw, c] * dY[(ih*d + pad - kh)/so, (iw*d + pad - kw)/so]; thread t owns tap t (and
        // t + 256, ...); part[image][K] = bias gradient = sum of the map
        {   // bias: 64 strided partial sums, then thread 0 adds them in order
            if (tid < 64) {                                     // this workgroup's share of the map
                const int per = (npo + w.split - 1) / w.split, e = min(npo, (sub + 1) * per);
                float sacc = 0.0f;
                for (int i = sub * per + tid; i < e; i += 64) sacc += sdy[i];
                sred[tid] = sacc;
            }
            lds_barrier();
        }
        for (int t = tid; t < w.Kp; t += kThreads) {
            float acc = 0.0f;
            if (t < K) {
                const int kp = t / Cin, c = t - kp * Cin, kh = kp / g.k, kw = kp - kh * g.k;
                const int hper = (g.IH + w.split - 1) / w.split;      // this workgroup's share of the input rows
                for (int ih = sub * hper; ih < min(g.IH, (sub + 1) * hper); ++ih) {
                    bool okh;
                    const int oh = div_small(ih * g.d + g.pad - kh, g.so, okh);
                    if (!okh || oh < 0 || oh >= g.OH) continue;
                    const float* xr = sx + ih * g.IW * Cin + c;
                    const float* yr = sdy + oh * g.OW;
                    if (g.so == 1) {                    // ow = iw*d + pad - kw: a contiguous run of iw, no test inside the loop
                        const int off = g.pad - kw;
                        const int lo = off >= 0 ? 0 : (-off + g.d - 1) / g.d;
                        const int hi = min(g.IW - 1, off <= g.OW - 1 ? (g.OW - 1 - off) / g.d : -1);
#pragma unroll 7
                        for (int iw = lo; iw <= hi; ++iw) acc += xr[iw * Cin] * yr[iw * g.d + off];
                    } else {
                        for (int iw = 0; iw < g.IW; ++iw) {
                            bool okw;
                            const int ow = div_small(iw * g.d + g.pad - kw, g.so, okw);
                            if (okw && ow >= 0 && ow < g.OW) acc += xr[iw * Cin] * yr[ow];
                        }
                    }
                }
            } else if (t == K && g.ones) {
                for (int i = 0; i < 64; ++i) acc += sred[i];
            }
            w.part[(size_t)(bid - w.block_base) * w.Kp + t] = acc;
        }
    }
}

// Column sums of per-workgroup partial sums (the direct stage's filter gradient, the bias row sums):
// dst[i*dst_ld] = sum_s src[s*stride + i].  A workgroup takes 4 elements; lane group q of 64 adds slices q, q+64, ... in order,
// then the 64 group sums are added in order: a fixed tree, reproducible.
__device__ __forceinline__ void colsum_body(const ReduceSeg& g, int bid, float (*red)[5]) {
    const int tid = threadIdx.x;
    const int e = tid & 3, q = tid >> 2;
    const int i = (bid - g.block_base) * 4 + e;
    float acc = 0.0f;
    if (i < g.n) {
        for (int s0 = q; s0 < g.parts; s0 += 64 * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = g.src[(size_t)min(s0 + 64 * u, g.parts - 1) * g.stride + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (s0 + 64 * u < g.parts) acc += v[u];
        }
    }
    red[q][e] = acc;
    lds_barrier();
    if (q == 0 && i < g.n) {
        float t = red[0][e];
        for (int k = 1; k < 64; ++k) t += red[k][e];
        g.dst[(size_t)i * (g.dst_ld > 0 ? g.dst_ld : 1)] = t;
    }
}
__global__ void __launch_bounds__(kThreads) k_colsum(ReduceArgs a) {
    __shared__ float red[64][5];
    const int bid = blockIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    colsum_body(a.seg[it], bid, red);
}

void launch_colsum(const ReduceArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    AVAE_LAUNCH(k_colsum, dim3(n_blocks), dim3(kThreads), 0, s, a);
}

template <typename CT, int CQ> static void launch_thin_fast(const ThinArgs& a, int n_blocks, hipStream_t s) {
    if (a.mode == 0) AVAE_LAUNCH((k_thin_fast<CT, CQ, 0>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else if (a.mode == 1) AVAE_LAUNCH((k_thin_fast<CT, CQ, 1>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_thin_fast<CT, CQ, 2>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}
void launch_thin(int compute_dtype, const ThinArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    bool fast = a.n_seg > 0;
    for (int i = 0; i < a.n_seg; ++i) fast = fast && a.seg[i].fast && a.seg[i].g.Cin == a.seg[0].g.Cin;
    if (fast) {
        const bool c16 = a.seg[0].g.Cin == 16;
        if (compute_dtype == AVAE_BF16) { if (c16) launch_thin_fast<__bf16, 4>(a, n_blocks, s); else launch_thin_fast<__bf16, 2>(a, n_blocks, s); }
        else { if (c16) launch_thin_fast<float, 4>(a, n_blocks, s); else launch_thin_fast<float, 2>(a, n_blocks, s); }
        return;
    }
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_thin<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else AVAE_LAUNCH((k_thin<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
}

// ------------------------------------------------------------------ serving: slot-indirect input / output moves
// Launched per call, ahead of the captured graph, with the call's pointers and row count BY VALUE: it stages z and publishes the
// slot the graph's output-store launch reads (one launch instead of a slot write + an in-graph staging kernel; the decoder's
// results go from the output launch's tile straight to the caller's buffers).
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_serve(ServeArgs a, ServeSlot call) {
    const int m = blockIdx.x / a.blocks_per_mod, b = blockIdx.x - m * a.blocks_per_mod;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.slot = call;
    // z rows -> Z[m] (compute dtype); padded rows of the bucket are written as zeros
    const int total = a.bucket * a.nz;
    for (int i = b * kThreads + threadIdx.x; i < total; i += a.blocks_per_mod * kThreads) {
        const int r = i / a.nz, c = i - r * a.nz;
        const float v = r < call.rows ? call.z[(size_t)r * a.nz + c] : 0.0f;
        reinterpret_cast<CT*>(a.Z[m])[(size_t)r * a.ldz[m] + c] = to_ct<CT>(v);
    }
}
// The per-call launch of avae_generate for the small nets, lean: no K loop at all -- the call's fp32 z rows go into the tail's LDS
// image (rows beyond the call's zero, constant-1 column included), the decoder's first layer of modality blockIdx.y is the tail
// product (slice = 64 output columns per workgroup), workgroup (0, 0) publishes the call's slot.  A 300-byte argument block instead of
// the grouped kernel's 3.4 KB: the host side of a call is the launch of this kernel + one graph replay.
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_serve_in(ServeInArgs a) {
    constexpr int ES = (int)sizeof(CT), TSL = ES == 2 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) unsigned char img[kHeadImg];
    const ServeInMod md = a.mod[blockIdx.y];
    const int tm = blockIdx.x / md.slices, ts = blockIdx.x - tm * md.slices;
    if (tm >= a.tiles_m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int m0 = tm * 32, nz = a.nz;
    // the call: by value in the kernel arguments (eager launch per call), or -- as the first node of a captured graph -- from the
    // record the host wrote into its pinned ring before the replay (one uniform read over PCIe per workgroup)
    ServeSlot call_v = a.call;
    if (a.rec) call_v = *a.rec;
    const ServeSlot& call = call_v;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        *a.slot = call;
        if (a.rec) { const unsigned long long n = *a.count + 1; *a.count = n; __hip_atomic_store(a.consumed, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    const int t_i = wave & 1, t_c0 = (2 * ts + (wave >> 1)) * 32, t_sl = 2 * md.kt;
    u32x4 tb[TSL][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned char* pw = reinterpret_cast<const unsigned char*>(md.w) + (size_t)min(t_c0 + 16 * j + fr, md.n - 1) * md.ldw * ES + fq * 16;
#pragma unroll
        for (int sl = 0; sl < TSL; ++sl) if (TSL == 2 || sl < t_sl) tb[sl][j] = *reinterpret_cast<const u32x4*>(pw + 64 * sl);
    }
    reinterpret_cast<f32x4*>(img)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4*>(img)[tid + kThreads] = f32x4{0.f, 0.f, 0.f, 0.f};
    lds_barrier();
    {
        const int zrow = tid >> 3, zg = tid & 7;
        if (m0 + zrow < call.rows) {
            // every latent column: 8 threads per row take 4 columns each per trip (n_z up to 63 needs two trips)
            for (int d0 = 4 * zg; d0 < nz; d0 += 32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = d0 + e;
                    if (d < nz) img_put<CT>(img, zrow, d, call.z[(size_t)(m0 + zrow) * nz + d]);
                }
            }
        }
        if (tid < 32) img_put<CT>(img, tid, nz, 1.0f);
    }
    lds_barrier();
    const int arow = 16 * t_i + fr;
    f32x4 tacc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int sl = 0; sl < TSL; ++sl)
        if (TSL == 2 || sl < t_sl) {
            const u32x4 ta = *reinterpret_cast<const u32x4*>(img + (sl >> 1) * (32 * kTileBytesK) + arow * kTileBytesK + (((fq + 4 * (sl & 1)) ^ ((arow >> 1) & 7)) << 4));
            mma<CT>(tb[sl][0], ta, tacc[0][0]);
            mma<CT>(tb[sl][1], ta, tacc[0][1]);
        }
    AVAE_ACT_DISPATCH(md.act, (regep_store<CT, 1, 2>(tacc, reinterpret_cast<CT*>(md.out), md.ldo, a.bucket, md.n, m0 + arow, t_c0, lane,
        [&](int, int, int, float v) { return act_fwd_t<ACT>(v); })))
}
void launch_serve_in(int compute_dtype, const ServeInArgs& a, int grid_x, hipStream_t s) {
    const dim3 grid(grid_x, a.n_mod), block(kThreads);
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_serve_in<__bf16>), grid, block, 0, s, a);
    else AVAE_LAUNCH((k_serve_in<float>), grid, block, 0, s, a);
}

void launch_serve(int compute_dtype, const ServeArgs& a, const ServeSlot& call, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) AVAE_LAUNCH((k_serve<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a, call);
    else AVAE_LAUNCH((k_serve<float>), dim3(n_blocks), dim3(kThreads), 0, s, a, call);
}

// ------------------------------------------------------------------ split-K reduction
__device__ __forceinline__ void reduce_body(const ReduceSeg& g, int bid) {
    const int i = ((bid - g.block_base) * kThreads + (int)threadIdx.x) * 4;       // n and stride are multiples of 4
    if (i >= g.n) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // slices are added in index order (reproducible); eight loads are in flight at a time -- one load per loop trip would
    // serialise `parts` (up to ~100) memory round trips
    for (int s0 = 0; s0 < g.parts; s0 += 8) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = min(s0 + u, g.parts - 1);        // clamped: the surplus loads are discarded below
            load4<float>(g.src + (size_t)sl * g.stride + i, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < g.parts) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
            }
    }
    if (g.perm_k > 0) {          // (k_gperm's index map, applied as the sums are stored)
        const int KA = g.perm_k * g.perm_k * g.perm_cout;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = i + e, ci = idx / g.perm_ldga, r = idx - ci * g.perm_ldga;
            if (r < KA) {
                const int kp = r / g.perm_cout, co = r - kp * g.perm_cout, kh = kp / g.perm_k, kw = kp - kh * g.perm_k;
                g.dst[(size_t)(((g.perm_k - 1 - kh) * g.perm_k + (g.perm_k - 1 - kw)) * g.perm_cin + ci) * g.perm_ld + co] = acc[e];
            }
        }
        return;
    }
    store4<float>(g.dst + i, acc);
}
__global__ void __launch_bounds__(kThreads) k_reduce(ReduceArgs a) {
    const int bid = blockIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    reduce_body(a.seg[it], bid);
}
// both kinds of sums behind the weight-gradient launches in ONE launch (they are independent: different destinations)
__global__ void __launch_bounds__(kThreads) k_sums(ReduceArgs a) {
    __shared__ float red[64][5];
    const int bid = blockIdx.x;
    int it = 0;
    for (int i = 1; i < a.n_seg; ++i) if (bid >= a.seg[i].block_base) it = i;
    if (a.seg[it].colsum) colsum_body(a.seg[it], bid, red);         // (block-uniform branch: the barrier inside is safe)
    else reduce_body(a.seg[it], bid);
}
void launch_sums(const ReduceArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    AVAE_LAUNCH(k_sums, dim3(n_blocks), dim3(kThreads), 0, s, a);
}

void launch_reduce(const ReduceArgs& a, int n_blocks, hipStream_t s) {
    if (n_blocks <= 0) return;
    AVAE_LAUNCH(k_reduce, dim3(n_blocks), dim3(kThreads), 0, s, a);
}

// ------------------------------------------------------------------ strided fill (constant-1 columns)
__global__ void k_fill(unsigned char* base, int elem_bytes, unsigned bits, long long start, long long stride, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned char* p = base + (size_t)(start + (long long)i * stride) * elem_bytes;
    if (elem_bytes == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)bits;
    else *reinterpret_cast<unsigned*>(p) = bits;
}

void launch_fill(void* base, int elem_bytes, unsigned bits, long long start, long long stride, int count, hipStream_t s) {
    if (count <= 0) return;
    AVAE_LAUNCH(k_fill, dim3((count + 255) / 256), dim3(256), 0, s, reinterpret_cast<unsigned char*>(base),
                       elem_bytes, bits, start, stride, count);
}

}  // namespace avae
