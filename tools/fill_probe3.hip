// Diagnostic micro-benchmark (not part of the product): the K loop of C4's hidden launches (256x128 tile, 8 waves of 64x64, LDS-DMA
// ring, ds_read_b128 fragments, v_mfma_f32_16x16x32_bf16) with the SAME LDS budget cut two ways --
//   A  3 stages of 64 K elements (128-byte rows, 48 KB per stage: two of three stages in flight = 96 KB)   = k_grouped
//   B  6 stages of 32 K elements ( 64-byte rows, 24 KB per stage: five of six in flight = 120 KB), a barrier per 32 K elements
// to answer one question: does the loop speed up when more bytes are in flight per CU (HISTORY.md: "the stage time is a latency")?
// Geometry of C4: two items of [4096 x 1024] x [1024 x 1024]^T, bf16 random data, k_grouped's XCD map.  No epilogue (the sums are
// folded into one word per lane so that nothing is optimised away).
// build: hipcc --offload-arch=gfx950 -O3 tools/fill_probe3.hip -o /tmp/fp3 ; run: /tmp/fp3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef const __attribute__((address_space(1))) void* gp_t;
typedef __attribute__((address_space(3))) void* lp_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int ROWB, int RING, int MF>       // MF: 1 = MFMAs + fragment reads, 0 = fill only
__global__ void __launch_bounds__(512) probe(const unsigned char* A, const unsigned char* B, int lda_b, int kbytes, float* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.y, part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = part * 16 + idx;
    const int tm = t >> 3, tn = t & 7;
    const unsigned char* Ag = A + (size_t)item * 4096 * lda_b + (size_t)tm * 256 * lda_b;
    const unsigned char* Bg = B + (size_t)item * 1024 * lda_b + (size_t)tn * 128 * lda_b;
    constexpr int STAGE = 384 * ROWB, CPR = ROWB / 16, RPP = 1024 / ROWB, NCH = STAGE / 1024 / 8;
    constexpr int SLABS = ROWB / 64;
    auto swz = [](int r) { return ROWB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    const unsigned char* src[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int r = (c * 8 + wave) * RPP + lane / CPR, ch = (lane % CPR) ^ swz(r);
        src[c] = (r < 256 ? Ag + (size_t)r * lda_b : Bg + (size_t)(r - 256) * lda_b) + ch * 16;
    }
    auto dma = [&](int kt, int buf) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            __builtin_amdgcn_global_load_lds((gp_t)(src[c] + (size_t)kt * ROWB), (lp_t)(smem + buf * STAGE + (c * 8 + wave) * 1024), 16, 0, 0);
    };
    const int steps = kbytes / ROWB;
    const int fr = lane & 15, fq = lane >> 4, wr = wave >> 1, wc = wave & 1;
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + fr, rb = 256 + wc * 64 + i * 16 + fr;
        aoff[i] = ra * ROWB + ((fq ^ swz(ra)) * 16);
        boff[i] = rb * ROWB + ((fq ^ swz(rb)) * 16);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < RING - 1 && p < steps; ++p) dma(p, p);
    int buf = 0;
    for (int kt = 0; kt < steps; ++kt) {
        const int rem = steps - 1 - kt;
        // tiles younger than kt that may stay in flight: min(rem, RING-2)
        if (rem >= RING - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * NCH) : "memory");
        else if (RING > 3 && rem == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NCH) : "memory");
        else if (RING > 3 && rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        const unsigned char* Sb = smem + buf * STAGE;
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (MF) {
#pragma unroll
            for (int s = 0; s < SLABS; ++s) {
                bf16x8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = *reinterpret_cast<const bf16x8*>(Sb + (aoff[i] ^ (s * 64)));
                    b[i] = *reinterpret_cast<const bf16x8*>(Sb + (boff[i] ^ (s * 64)));
                }
                if (s == 0 && kt + RING - 1 < steps) dma(kt + RING - 1, fill);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            acc[0][0][0] += (float)Sb[tid * 16];
            if (kt + RING - 1 < steps) dma(kt + RING - 1, fill);
        }
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123456.789f) sink[0] = s;
}

template <int ROWB, int RING, int MF> static float run_warm(const unsigned char* A, const unsigned char* B, int lda_b, int kbytes, float* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<ROWB, RING, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int lds = RING * 384 * ROWB;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<ROWB, RING, MF>), dim3(128, 2), dim3(512), lds, 0, A, B, lda_b, kbytes, sink);
    (void)hipEventRecord(a);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((probe<ROWB, RING, MF>), dim3(128, 2), dim3(512), lds, 0, A, B, lda_b, kbytes, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 200 * 1000.f;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int lda_b = 1088 * 2;
    unsigned char *A, *B; float* sink;
    const size_t na = (size_t)2 * 4096 * lda_b, nb = (size_t)2 * 1024 * lda_b;
    (void)hipMalloc(&A, na); (void)hipMalloc(&B, nb); (void)hipMalloc(&sink, 64);
    {   // random bf16 in (-1, 1): the clock the chip holds under MFMA load depends on the data
        std::vector<unsigned short> h(na / 2);
        srand(1);
        auto fill = [&](size_t n) { for (size_t i = 0; i < n; ++i) { float v = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &v, 4); h[i] = (unsigned short)(u >> 16); } };
        fill(na / 2); (void)hipMemcpy(A, h.data(), na, hipMemcpyHostToDevice);
        fill(nb / 2); (void)hipMemcpy(B, h.data(), nb, hipMemcpyHostToDevice);
    }
    printf("per launch, back to back (operands warm in L2 / Infinity Cache); K = 512 and K = 1024 elements; us per 64 K elements from the difference\n");
#define ROW(name, ROWB, RING, MF) { const float t8 = run_warm<ROWB, RING, MF>(A, B, lda_b, 1024, sink), t16 = run_warm<ROWB, RING, MF>(A, B, lda_b, 2048, sink);    \
        printf("%-58s K=512 %6.2f us  K=1024 %6.2f us  => %.3f us per 64-K step\n", name, t8, t16, (t16 - t8) / 8); }
    for (int rep = 0; rep < 3; ++rep) {
        ROW("A  3 x 48 KB stages (64 K), MFMA + fragment reads", 128, 3, 1)
        ROW("B  6 x 24 KB stages (32 K), MFMA + fragment reads", 64, 6, 1)
        ROW("B' 5 x 24 KB stages (32 K), MFMA + fragment reads", 64, 5, 1)
        ROW("B\" 4 x 24 KB stages (32 K), MFMA + fragment reads", 64, 4, 1)
        ROW("A  fill only", 128, 3, 0)
        ROW("B  fill only", 64, 6, 0)
    }
    return 0;
}
