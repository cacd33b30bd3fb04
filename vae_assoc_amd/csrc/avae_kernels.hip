// gfx950 (MI355X, CDNA4) kernels of the associative-VAE training path.
//
// One grouped kernel does every matrix product of the step.  All products are brought to the
// same "NT" form  C[M,N] = sum_k A[m][k] * B[n][k]  (both operands K-contiguous in HBM, K padded
// with zeros to 128-byte units) by keeping, next to every activation / gradient / weight, its
// transposed copy, written by the producing epilogue:
//     forward   Y  = X_aug . W_aug        A = X_aug  [B][in+1]      B = W_aug^T [out][in+1]
//     dgrad     dX = dA . W^T             A = dA     [B][out]       B = W_aug   [in+1][out]
//     wgrad     dW_aug = X_aug^T . dA     A = X_aug^T [in+1][B]     B = dA^T    [out][B]
// The bias is the last row of W_aug and every activation carries a constant-1 column, so the
// bias add and the bias gradient fall out of the same MFMA products.
//
// Per 256-thread workgroup: a BM x BN output tile (64x64 or 128x128), 4 wave64s in a 2x2
// arrangement, v_mfma_f32_16x16x32_bf16 (bf16 operands) or v_mfma_f32_16x16x4_f32 (exact fp32),
// fp32 accumulation in registers, register-staged global->LDS double buffering with 144-byte
// LDS rows (128 B of K + 16 B pad: conflict-free ds_read_b128 over 16 rows), and an LDS-staged
// epilogue that fuses the activation / reparameterisation / loss / gradient maths and writes both
// the row-major and the transposed result with coalesced vector stores.
//
// Reference maths: /root/reference/vae_assoc.py:163-222 (encoder), :243-304 (decoder),
// :306-371 (losses), :373-374 (Adam); restated for CPU in oracle/vae_assoc_oracle.py.
#include "avae_device.h"
#include "../../include/avae.h"

namespace avae {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // native vector: stays in registers (HIP's uint4 struct does not)

constexpr int kLdsRow = kTileBytesK + 16;   // 144 B

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

template <typename CT> __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c);
template <> __device__ __forceinline__ void mma<__bf16>(const u32x4& a, const u32x4& b, f32x4& c) {
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

__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + expf(-a)); }

__device__ __forceinline__ float act_fwd(int act, float a) {
    switch (act) {
        case AVAE_ACT_RELU: return fmaxf(a, 0.0f);
        case AVAE_ACT_SOFTPLUS: return fmaxf(a, 0.0f) + log1pf(expf(-fabsf(a)));
        case AVAE_ACT_SIGMOID: return sigmoidf_(a);
        case AVAE_ACT_TANH: return tanhf(a);
        default: return a;
    }
}
// derivative of the transfer function expressed through its OUTPUT y
__device__ __forceinline__ float act_bwd(int act, float y) {
    switch (act) {
        case AVAE_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case AVAE_ACT_SOFTPLUS: return 1.0f - expf(-y);
        case AVAE_ACT_SIGMOID: return y * (1.0f - y);
        case AVAE_ACT_TANH: return 1.0f - y * y;
        default: return 1.0f;
    }
}

// Sum over the 256 threads of the block, same value returned to every thread, fixed order.
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------ epilogue passes
template <typename OT, int BM, int BN>
__device__ __forceinline__ void transposed_store(const float* Cs, OT* out, int ld, int M, int N, int m0, int n0) {
    constexpr int LDC = BN + 4;
    constexpr int QR = BM / 4;
    for (int idx = threadIdx.x; idx < BN * QR; idx += kThreads) {
        const int col = idx / QR, r4 = (idx - col * QR) * 4;
        const int gcol = n0 + col, grow = m0 + r4;
        if (gcol < N && grow < M) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = Cs[(r4 + e) * LDC + col];
            store_row<OT>(out + (size_t)gcol * ld + grow, v, M - grow);
        }
    }
}

// ------------------------------------------------------------------ non-GEMM work items
// KL(q||N(0,I)) (vae_assoc.py:335-337) and the symmetric-KL association penalty (:346-366) with
// their gradients w.r.t. (mu, lv).  The log-determinant terms of the two directed KLs cancel, so
// per sample and dimension  S = 1/2 [ e^a + e^-a - 2 + D^2 (e^-lvi + e^-lvj) ],  a = lvi-lvj,
// D = mui-muj;  e^a + e^-a - 2 is evaluated as (2 sinh(a/2))^2 to avoid cancellation.
__device__ void latent_item(const WorkItem& w, int t, float* red) {
    const int nz = w.nz, nz2 = 2 * w.nz, M = w.M;
    float csum = 0.0f;
    for (int idx = threadIdx.x; idx < kLatentRows * nz; idx += kThreads) {
        const int row = idx / nz, d = idx - row * nz;
        const int grow = t * kLatentRows + row;
        if (grow >= M) continue;
        float mu[kMaxMod], lv[kMaxMod], gmu[kMaxMod], glv[kMaxMod];
#pragma unroll
        for (int m = 0; m < kMaxMod; ++m) {
            mu[m] = lv[m] = gmu[m] = glv[m] = 0.0f;
            if (m < w.n_mod) {
                mu[m] = w.mulv[m][(size_t)grow * nz2 + d];
                lv[m] = w.mulv[m][(size_t)grow * nz2 + nz + d];
                const float el = expf(lv[m]);
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
                    const float eni = expf(-lv[i]), enj = expf(-lv[j]);
                    const float sh = 2.0f * sinhf(0.5f * a);
                    const float dsh = 2.0f * sinhf(a);          // e^a - e^-a
                    csum += w.lambda * 0.5f * (sh * sh + dl * dl * (eni + enj));
                    const float gm = w.lambda * dl * (eni + enj);
                    gmu[i] += gm; gmu[j] -= gm;
                    glv[i] += 0.5f * w.lambda * (dsh - dl * dl * eni);
                    glv[j] += 0.5f * w.lambda * (-dsh - dl * dl * enj);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < kMaxMod; ++m) {
            if (m < w.n_mod) {
                w.g0[m][(size_t)grow * nz2 + d] = gmu[m];
                w.g0[m][(size_t)grow * nz2 + nz + d] = glv[m];
            }
        }
    }
    const float total = block_sum(csum, red);
    if (threadIdx.x == 0) w.partial[w.slot_base + t] = total;
}

__device__ void cost_item(const WorkItem& w, DevState* st, float* red) {
    float s = 0.0f;
    for (int i = threadIdx.x; i < w.n_slots; i += kThreads) s += w.partial[i];
    const float total = block_sum(s, red);
    if (threadIdx.x == 0) {
        reinterpret_cast<float*>(w.out0)[0] = total;
        if (w.bump_step) st->step += 1;
    }
}

// ------------------------------------------------------------------ the grouped kernel
template <int BM, int BN> struct TileSmem {
    static constexpr int kStage = (BM + BN) * kLdsRow;
    static constexpr int kStages = 2 * kStage;
    static constexpr int kC = BM * (BN + 4) * 4;
    static constexpr int kMain = kStages > kC ? kStages : kC;
    static constexpr int kTotal = kMain + 64;
};

template <typename CT, int BM, int BN>
__global__ void __launch_bounds__(kThreads) k_grouped(const WorkItem* __restrict__ items, int n_items, DevState* st,
                                                      unsigned long long* stamps, int launch_id) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[TileSmem<BM, BN>::kTotal];
    float* red = reinterpret_cast<float*>(smem + TileSmem<BM, BN>::kMain);
#ifdef AVAE_STAMPS
    unsigned long long sv[kStampWords] = {0, 0, 0, 0, 0, 0, 0, 0};
#define AVAE_STAMP(i) { __builtin_amdgcn_sched_barrier(0); sv[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
#define AVAE_STAMP_FLUSH() { if (stamps && threadIdx.x == 0 && launch_id < kStampLaunches && blockIdx.x < kStampBlocks) { \
        sv[6] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
        for (int i_ = 0; i_ < kStampWords; ++i_) stamps[((size_t)launch_id * kStampBlocks + blockIdx.x) * kStampWords + i_] = sv[i_]; } }
    AVAE_STAMP(0)
    sv[5] = __builtin_amdgcn_s_memtime();
#else
#define AVAE_STAMP(i)
#define AVAE_STAMP_FLUSH()
#endif

    const int bid = blockIdx.x;
    int it = 0;
    while (it + 1 < n_items && bid >= items[it + 1].tile_base) ++it;
    const WorkItem& w = items[it];
    const int t = bid - w.tile_base;
    if (w.kind == K_LATENT) { latent_item(w, t, red); AVAE_STAMP(4) AVAE_STAMP_FLUSH() return; }
    if (w.kind == K_COST) { cost_item(w, st, red); AVAE_STAMP(4) AVAE_STAMP_FLUSH() return; }

    constexpr int WM = BM / 2, WN = BN / 2;       // per-wave sub-tile (2x2 waves)
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int NCA = BM * 8 / kThreads, NCB = BN * 8 / kThreads;   // 16-B chunks per thread per tile
    constexpr int LDC = BN + 4;
    constexpr int ES = (int)sizeof(CT);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tm = t / w.tiles_n, tn = t - tm * w.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const unsigned char* Ag = reinterpret_cast<const unsigned char*>(w.A) + (size_t)m0 * w.lda * ES;
    const unsigned char* Bg = reinterpret_cast<const unsigned char*>(w.B) + (size_t)n0 * w.ldb * ES;
    const size_t lda_b = (size_t)w.lda * ES, ldb_b = (size_t)w.ldb * ES;
    const int nk = (w.K * ES) / kTileBytesK;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Register-staged double buffering: tile kt+1 is fetched from HBM/L2 into registers while the
    // MFMAs of tile kt run out of LDS; it is written to the other LDS buffer afterwards.  One
    // barrier per K tile.  (Plain arrays + fully unrolled loops only: anything fancier lands the
    // staging registers in scratch.)
    typedef const __attribute__((address_space(1))) u32x4* gp_t;
    u32x4 ra[NCA], rb[NCB];
    const int srow = tid >> 3, sch = tid & 7;       // this thread stages rows srow + 32*c, 16-B chunk sch
    const unsigned char* Ath = Ag + srow * lda_b + sch * 16;
    const unsigned char* Bth = Bg + srow * ldb_b + sch * 16;
    const int soff = srow * kLdsRow + sch * 16;
    const int fr = lane & 15, fq = lane >> 4;
    const int aoff = (wr * WM + fr) * kLdsRow + fq * 16;
    const int boff = BM * kLdsRow + (wc * WN + fr) * kLdsRow + fq * 16;

#define AVAE_GLOAD(kt)                                                                                 \
    {                                                                                                  \
        _Pragma("unroll") for (int c = 0; c < NCA; ++c)                                                \
            ra[c] = *(gp_t)(Ath + (size_t)(32 * c) * lda_b + (size_t)(kt) * kTileBytesK);              \
        _Pragma("unroll") for (int c = 0; c < NCB; ++c)                                                \
            rb[c] = *(gp_t)(Bth + (size_t)(32 * c) * ldb_b + (size_t)(kt) * kTileBytesK);              \
    }
#define AVAE_SWRITE(buf)                                                                               \
    {                                                                                                  \
        unsigned char* Sb = smem + (buf) * TileSmem<BM, BN>::kStage + soff;                            \
        _Pragma("unroll") for (int c = 0; c < NCA; ++c)                                                \
            *reinterpret_cast<u32x4*>(Sb + 32 * c * kLdsRow) = ra[c];                                  \
        _Pragma("unroll") for (int c = 0; c < NCB; ++c)                                                \
            *reinterpret_cast<u32x4*>(Sb + (BM + 32 * c) * kLdsRow) = rb[c];                           \
    }
#define AVAE_COMPUTE(buf)                                                                              \
    {                                                                                                  \
        const unsigned char* Sb = smem + (buf) * TileSmem<BM, BN>::kStage;                             \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                \
            u32x4 a[MI], b[NI];                                                                        \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                             \
                a[i] = *reinterpret_cast<const u32x4*>(Sb + aoff + i * 16 * kLdsRow + s * 64);         \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                             \
                b[j] = *reinterpret_cast<const u32x4*>(Sb + boff + j * 16 * kLdsRow + s * 64);         \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                             \
                _Pragma("unroll") for (int j = 0; j < NI; ++j) mma<CT>(a[i], b[j], acc[i][j]);         \
        }                                                                                              \
    }

    AVAE_STAMP(1)
    AVAE_GLOAD(0)
    AVAE_SWRITE(0)
    __syncthreads();
    AVAE_STAMP(2)
    for (int kt = 0; kt < nk - 1; ++kt) {
        AVAE_GLOAD(kt + 1)
        AVAE_COMPUTE(kt & 1)
        AVAE_SWRITE((kt + 1) & 1)
        __syncthreads();
    }
    AVAE_COMPUTE((nk - 1) & 1)
    __syncthreads();
    AVAE_STAMP(3)
#undef AVAE_GLOAD
#undef AVAE_SWRITE
#undef AVAE_COMPUTE

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
                    Cs[(wr * WM + i * 16 + fq * 4 + r) * LDC + wc * WN + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();

    const int M = w.M, N = w.N;
    constexpr int QC = BN / 4;
    switch (w.kind) {
    case K_FWD_HIDDEN: {
        CT* Y = reinterpret_cast<CT*>(w.out0);
        for (int idx = tid; idx < BM * QC; idx += kThreads) {
            const int row = idx / QC, c4 = (idx - row * QC) * 4;
            const int grow = m0 + row, gcol = n0 + c4;
            if (grow < M && gcol < N) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = act_fwd(w.act, Cs[row * LDC + c4 + e]); Cs[row * LDC + c4 + e] = v[e]; }
                store_row<CT>(Y + (size_t)grow * w.ld0 + gcol, v, N - gcol);
            }
        }
        if (w.out1) {
            __syncthreads();
            transposed_store<CT, BM, BN>(Cs, reinterpret_cast<CT*>(w.out1), w.ld1, M, N, m0, n0);
        }
    } break;
    case K_FWD_HEAD: {
        // columns [0,nz) = mu, [nz,2nz) = log sigma^2 (vae_assoc.py:217-221); z = mu + sqrt(exp(lv))*eps (:102-103)
        const int nz = w.nz;
        float* mulv = reinterpret_cast<float*>(w.out0);
        CT* Z = reinterpret_cast<CT*>(w.out1);
        const float* eps = reinterpret_cast<const float*>(w.aux0);
        for (int idx = tid; idx < BM * nz; idx += kThreads) {
            const int row = idx / nz, d = idx - row * nz;
            const int grow = m0 + row;
            if (grow < M) {
                const float mu = Cs[row * LDC + d], lv = Cs[row * LDC + nz + d];
                mulv[(size_t)grow * w.ld0 + d] = mu;
                mulv[(size_t)grow * w.ld0 + nz + d] = lv;
                if (Z) {
                    const float z = mu + expf(0.5f * lv) * eps[(size_t)grow * nz + d];
                    Z[(size_t)grow * w.ld1 + d] = to_ct<CT>(z);
                    Cs[row * LDC + d] = z;
                }
            }
        }
        if (Z && w.out2) {
            __syncthreads();
            transposed_store<CT, BM, BN>(Cs, reinterpret_cast<CT*>(w.out2), w.ld2, M, nz, m0, 0);
        }
    } break;
    case K_FWD_OUT_LOSS: {
        // Bernoulli: -sum x log(1e-3+p) + (1-x) log(1e-3+1-p), p = sigmoid(a)  (:321-324), mean over batch (:340)
        // Gaussian : sum (x-a)^2 / 2 over the WHOLE batch, not averaged          (:327-328,:340)
        CT* dA = reinterpret_cast<CT*>(w.out0);
        const float* X = reinterpret_cast<const float*>(w.aux0);
        float csum = 0.0f;
        for (int idx = tid; idx < BM * QC; idx += kThreads) {
            const int row = idx / QC, c4 = (idx - row * QC) * 4;
            const int grow = m0 + row, gcol = n0 + c4;
            if (grow < M && gcol < N) {
                float v[4], x[4];
                load4<float>(X + (size_t)grow * w.ldx + gcol, x);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = Cs[row * LDC + c4 + e];
                    float da = 0.0f;
                    if (gcol + e < N) {
                        if (w.binary) {
                            const float p = sigmoidf_(a);
                            const float lp = 1e-3f + p, lq = 1e-3f + 1.0f - p;
                            csum += -w.scale * (x[e] * logf(lp) + (1.0f - x[e]) * logf(lq));
                            da = w.scale * p * (1.0f - p) * (-x[e] / lp + (1.0f - x[e]) / lq);
                        } else {
                            const float df = a - x[e];
                            csum += w.scale * 0.5f * df * df;
                            da = w.scale * df;
                        }
                    }
                    v[e] = da;
                    Cs[row * LDC + c4 + e] = da;
                }
                store_row<CT>(dA + (size_t)grow * w.ld0 + gcol, v, N - gcol);
            }
        }
        const float total = block_sum(csum, red);
        if (tid == 0) w.partial[w.slot_base + t] = total;
        if (w.out1) {
            __syncthreads();
            transposed_store<CT, BM, BN>(Cs, reinterpret_cast<CT*>(w.out1), w.ld1, M, N, m0, n0);
        }
    } break;
    case K_FWD_OUT_STORE: {
        float* O = reinterpret_cast<float*>(w.out0);
        for (int idx = tid; idx < BM * QC; idx += kThreads) {
            const int row = idx / QC, c4 = (idx - row * QC) * 4;
            const int grow = m0 + row, gcol = n0 + c4;
            if (grow < M && gcol < N) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = Cs[row * LDC + c4 + e]; v[e] = w.binary ? sigmoidf_(a) : a; }
                store_row<float>(O + (size_t)grow * w.ld0 + gcol, v, N - gcol);
            }
        }
    } break;
    case K_DGRAD_HIDDEN: {
        CT* dX = reinterpret_cast<CT*>(w.out0);
        const CT* Yp = reinterpret_cast<const CT*>(w.aux0);
        for (int idx = tid; idx < BM * QC; idx += kThreads) {
            const int row = idx / QC, c4 = (idx - row * QC) * 4;
            const int grow = m0 + row, gcol = n0 + c4;
            if (grow < M && gcol < N) {
                float v[4], y[4];
                load4<CT>(Yp + (size_t)grow * w.ldx + gcol, y);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = Cs[row * LDC + c4 + e] * act_bwd(w.act, y[e]); Cs[row * LDC + c4 + e] = v[e]; }
                store_row<CT>(dX + (size_t)grow * w.ld0 + gcol, v, N - gcol);
            }
        }
        if (w.out1) {
            __syncthreads();
            transposed_store<CT, BM, BN>(Cs, reinterpret_cast<CT*>(w.out1), w.ld1, M, N, m0, n0);
        }
    } break;
    case K_DGRAD_LATENT: {
        // dz -> (dmu, dlv): dmu = dz + g0mu; dlv = dz * 1/2 exp(lv/2) eps + g0lv   (reparam :102-103)
        const int nz = w.nz;
        CT* dH = reinterpret_cast<CT*>(w.out0);
        const float* mulv = reinterpret_cast<const float*>(w.aux0);
        const float* eps = reinterpret_cast<const float*>(w.aux1);
        const float* g0 = reinterpret_cast<const float*>(w.aux2);
        for (int idx = tid; idx < BM * nz; idx += kThreads) {
            const int row = idx / nz, d = idx - row * nz;
            const int grow = m0 + row;
            if (grow < M) {
                const float dz = Cs[row * LDC + d];
                const float lv = mulv[(size_t)grow * 2 * nz + nz + d];
                const float dmu = dz + g0[(size_t)grow * 2 * nz + d];
                const float dlv = dz * 0.5f * expf(0.5f * lv) * eps[(size_t)grow * nz + d] + g0[(size_t)grow * 2 * nz + nz + d];
                dH[(size_t)grow * w.ld0 + d] = to_ct<CT>(dmu);
                dH[(size_t)grow * w.ld0 + nz + d] = to_ct<CT>(dlv);
                Cs[row * LDC + d] = dmu;
                Cs[row * LDC + nz + d] = dlv;
            }
        }
        if (w.out1) {
            __syncthreads();
            transposed_store<CT, BM, BN>(Cs, reinterpret_cast<CT*>(w.out1), w.ld1, M, 2 * nz, m0, 0);
        }
    } break;
    case K_WGRAD: {
        float* G = reinterpret_cast<float*>(w.out0);
        for (int idx = tid; idx < BM * QC; idx += kThreads) {
            const int row = idx / QC, c4 = (idx - row * QC) * 4;
            const int grow = m0 + row, gcol = n0 + c4;
            if (grow < M && gcol < N) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Cs[row * LDC + c4 + e];
                store_row<float>(G + (size_t)grow * w.ld0 + gcol, v, N - gcol);
            }
        }
    } break;
    default: break;
    }
    AVAE_STAMP(4)
    AVAE_STAMP_FLUSH()
}

void launch_grouped(int compute_dtype, int tile_cfg, const WorkItem* items, int n_items, int n_blocks,
                    DevState* st, hipStream_t s, unsigned long long* stamps, int launch_id) {
    dim3 grid(n_blocks), block(kThreads);
    if (compute_dtype == AVAE_BF16) {
        if (tile_cfg == 0) hipLaunchKernelGGL((k_grouped<__bf16, 64, 64>), grid, block, 0, s, items, n_items, st, stamps, launch_id);
        else hipLaunchKernelGGL((k_grouped<__bf16, 128, 128>), grid, block, 0, s, items, n_items, st, stamps, launch_id);
    } else {
        if (tile_cfg == 0) hipLaunchKernelGGL((k_grouped<float, 64, 64>), grid, block, 0, s, items, n_items, st, stamps, launch_id);
        else hipLaunchKernelGGL((k_grouped<float, 128, 128>), grid, block, 0, s, items, n_items, st, stamps, launch_id);
    }
}

// ------------------------------------------------------------------ Adam + shadow refresh
// TF-1 AdamOptimizer dense update (reference vae_assoc.py:373-374; TF training_ops ApplyAdam):
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m += (g-m)(1-b1);  v += (g^2-v)(1-b2);
//   theta -= lr_t*m/(sqrt(v)+eps)                       (epsilon outside the bias correction)
// Tile-wise over each layer's [in+1][out] matrix so the same pass emits the compute-dtype shadow
// W (row-major, dgrad operand) and, through an LDS transpose, W^T (forward operand).
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_adam(AdamArgs a) {
    __shared__ float T[64][65];
    __shared__ float s_lr;
    const int bid = blockIdx.x, tid = threadIdx.x;
    int it = 0;
    while (it + 1 < a.n_items && bid >= a.items[it + 1].tile_base) ++it;
    const AdamItem& w = a.items[it];
    const int t = bid - w.tile_base;
    const int tr = t / w.tiles_c, tc = t - tr * w.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;

    if (a.mode == 0 && tid == 0) {
        const double ts = (double)a.st->step;     // already counts this step (bumped by K_COST)
        const double b1t = pow((double)a.beta1, ts), b2t = pow((double)a.beta2, ts);
        s_lr = (float)((double)a.lr * sqrt(1.0 - b2t) / (1.0 - b1t));
        if (bid == 0) {
            const float c = *a.cost_src;
            a.st->last_cost = c;
            a.st->cost_hist[(a.st->step - 1) % kCostHist] = c;
        }
    }
    __syncthreads();
    const float lr_t = a.mode == 0 ? s_lr : 0.0f;
    const float omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;

    const int c4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i;
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
                for (int e = 0; e < 4; ++e) {
                    m[e] += (g[e] - m[e]) * omb1;
                    v[e] += (g[e] * g[e] - v[e]) * omb2;
                    th[e] -= (m[e] * lr_t) / (sqrtf(v[e]) + a.eps);
                }
                const int nv = w.cols - gcol;
                store_row<float>(w.theta + off, th, nv);
                store_row<float>(w.m + off, m, nv);
                store_row<float>(w.v + off, v, nv);
            }
            store_row<CT>(reinterpret_cast<CT*>(w.W) + off, th, w.cols - gcol);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r][c4 + e] = th[e];
    }
    __syncthreads();
    const int r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (tid >> 4) + 16 * i;
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
    if (compute_dtype == AVAE_BF16) hipLaunchKernelGGL((k_adam<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((k_adam<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
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
    __shared__ float T[64][65];
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (bid >= a.total_tiles) {
        // eps: one quad of dims per thread; counter = (global row, quad, step lo, step hi ^ salt)
        if (!a.eps_dst) return;
        const int nq = (a.nz + 3) / 4;
        const int q = (bid - a.total_tiles) * kThreads + tid;
        if (q >= a.eps_rows * nq) return;
        const int row = q / nq, d4 = q - row * nq;
        float n[4];
        if (a.eps_src) {
#pragma unroll
            for (int e = 0; e < 4; ++e) n[e] = (4 * d4 + e < a.nz) ? a.eps_src[(size_t)row * a.nz + 4 * d4 + e] : 0.0f;
        } else {
            const unsigned long long step = (unsigned long long)a.st->step;
            unsigned r[4];
            philox4x32_10((unsigned)(a.row_offset + row), (unsigned)d4, (unsigned)step,
                          (unsigned)(step >> 32) ^ (unsigned)a.stream_salt,
                          (unsigned)a.seed, (unsigned)(a.seed >> 32), r);
            // Box-Muller on (0,1) uniforms built from the top 24 bits
            const float u0 = ((r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
            const float tw = 6.28318530717958647692f;
            n[0] = ra * cosf(tw * u1); n[1] = ra * sinf(tw * u1);
            n[2] = rb * cosf(tw * u3); n[3] = rb * sinf(tw * u3);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * d4 + e < a.nz) a.eps_dst[(size_t)row * a.nz + 4 * d4 + e] = n[e];
        return;
    }
    int it = 0;
    while (it + 1 < a.n_seg && bid >= a.seg[it + 1].tile_base) ++it;
    const PrepSeg& w = a.seg[it];
    const int t = bid - w.tile_base;
    const int tr = t / w.tiles_c, tc = t - tr * w.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;
    const int c4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i;
        const int grow = r0 + r, gcol = c0 + c4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (grow < w.rows && gcol < w.cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gcol + e < w.cols) v[e] = w.src[(size_t)grow * w.src_ld + gcol + e];
            const int nv = w.cols - gcol;
            if (w.dst32) store_row<float>(w.dst32 + (size_t)grow * w.ld32 + gcol, v, nv);
            store_row<CT>(reinterpret_cast<CT*>(w.dstc) + (size_t)grow * w.ldc + gcol, v, nv);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r][c4 + e] = v[e];
    }
    if (!w.dstct) return;
    __syncthreads();
    const int r4 = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (tid >> 4) + 16 * i;
        const int gcol = c0 + c, grow = r0 + r4;
        if (gcol < w.cols && grow < w.rows) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = T[r4 + e][c];
            store_row<CT>(reinterpret_cast<CT*>(w.dstct) + (size_t)gcol * w.ldct + grow, v, w.rows - grow);
        }
    }
}

void launch_prep(int compute_dtype, const PrepArgs& a, hipStream_t s) {
    const int n_blocks = a.total_tiles + a.eps_blocks;
    if (n_blocks <= 0) return;
    if (compute_dtype == AVAE_BF16) hipLaunchKernelGGL((k_prep<__bf16>), dim3(n_blocks), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((k_prep<float>), dim3(n_blocks), dim3(kThreads), 0, s, a);
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
    hipLaunchKernelGGL(k_fill, dim3((count + 255) / 256), dim3(256), 0, s, reinterpret_cast<unsigned char*>(base),
                       elem_bytes, bits, start, stride, count);
}

}  // namespace avae
