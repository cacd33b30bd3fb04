// Diagnostic micro-benchmark (not part of the product): is the operand fill of the 256x128 8-wave tile paced by the latency of the
// lines that miss the XCD's L2 (first touch of every line comes from the Infinity Cache / HBM; the sharers of a line run in
// lockstep and all wait for that one miss), and does an L2 prefetch D tiles ahead remove it?
// Geometry of C4's hidden launches: two items of [4096 x K] x [1024 x K]^T, K = 1024 bf16, 256x128 tiles, k_grouped's XCD map
// (XCD c owns 16 consecutive tiles of each item = 2 row panels x 8 column panels), 3-stage ring, no MFMA.
//   V0  streaming, as k_grouped                        V1  every step re-reads K tile 0 (all L2 hits after the first step)
//   V2  V0 + a 9th wave that touches one dword of every 128-B line of tile kt+D (fire and forget)
//   V3  V0 + every wave touches the lines of ITS pieces of tile kt+2+D right after its refill DMA (in-order vmcnt: counted)
// build: hipcc --offload-arch=gfx950 -O3 tools/fill_probe2.hip -o /tmp/fp2 ; run: /tmp/fp2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef const __attribute__((address_space(1))) void* gp_t;
typedef __attribute__((address_space(3))) void* lp_t;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int V, int D>
__global__ void __launch_bounds__(V == 2 ? 576 : 512) probe(const unsigned char* A, const unsigned char* B, int lda_b, int steps, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // k_grouped's tile order: x % 8 = XCD, contiguous chunk of the item's tile list per XCD
    const int item = blockIdx.y, part = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int t = part * 16 + idx;                     // 128 tiles per item, 16 per XCD
    const int tm = t >> 3, tn = t & 7;
    const unsigned char* Ag = A + (size_t)item * 4096 * lda_b + (size_t)tm * 256 * lda_b;
    const unsigned char* Bg = B + (size_t)item * 1024 * lda_b + (size_t)tn * 128 * lda_b;
    constexpr int STAGE = (256 + 128) * 128, RING = 3, NCH = 6;
    unsigned acc = 0;
    if (V == 2 && wave == 8) {                         // prefetcher: 384 lines per tile = 6 loads of 64 lines
        for (int kt = 0; kt < steps; ++kt) {
            const int pk = kt + D;
            if (pk < steps) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int r = c * 64 + lane;
                    const unsigned char* p = (r < 256 ? Ag + (size_t)r * lda_b : Bg + (size_t)(r - 256) * lda_b) + (size_t)pk * 128;
                    unsigned v;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p));
                }
            }
            asm volatile("s_barrier" ::: "memory");
        }
        return;
    }
    const unsigned char* src[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int r = (c * 8 + wave) * 8 + (lane >> 3);
        src[c] = (r < 256 ? Ag + (size_t)r * lda_b : Bg + (size_t)(r - 256) * lda_b) + (lane & 7) * 16;
    }
    // V3: lanes 0..47 each own one line of this wave's 6 pieces (8 rows each)
    const int pr = (((lane >> 3) * 8 + wave) * 8 + (lane & 7));
    const unsigned char* psrc = (pr < 256 ? Ag + (size_t)pr * lda_b : Bg + (size_t)(pr - 256) * lda_b);
    auto dma = [&](int kt, int buf) {
        const int k = V == 1 ? 0 : kt;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            __builtin_amdgcn_global_load_lds((gp_t)(src[c] + (size_t)k * 128), (lp_t)(smem + buf * STAGE + (c * 8 + wave) * 1024), 16, 0, 0);
    };
    auto pf = [&](int kt) {
        if (V == 3 && kt < steps && lane < 48) { unsigned v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(psrc + (size_t)kt * 128)); }
    };
    constexpr int XPF = V == 3 ? 1 : 0;                 // extra vmcnt entries per iteration
    if (V == 3) for (int k = 2; k < 2 + D && k < steps; ++k) pf(k);     // warm the first D tiles (not counted precisely: waited by the first waits)
    dma(0, 0); dma(1, 1);
    int buf = 0;
    for (int kt = 0; kt < steps; ++kt) {
        if (kt + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH + 2 * XPF) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        acc += smem[buf * STAGE + tid * 16];             // touch the stage
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + 2 < steps) dma(kt + 2, fill);
        pf(kt + 2 + D);
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int V, int D> static float run(const unsigned char* A, const unsigned char* B, int lda_b, int steps, unsigned* sink, unsigned char* trash, size_t trash_bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float tot = 0.f;
    for (int i = 0; i < 25; ++i) {
        (void)hipMemsetAsync(trash, i, trash_bytes, 0);       // 512 MiB written: the operands leave the Infinity Cache (cold, as after a long step)
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((probe<V, D>), dim3(128, 2), dim3(V == 2 ? 576 : 512), 3 * 384 * 128, 0, A, B, lda_b, steps, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (i >= 5) tot += ms;
    }
    return tot / 20 * 1000.f;
}
template <int V, int D> static float run_warm(const unsigned char* A, const unsigned char* B, int lda_b, int steps, unsigned* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<V, D>), dim3(128, 2), dim3(V == 2 ? 576 : 512), 3 * 384 * 128, 0, A, B, lda_b, steps, sink);
    (void)hipEventRecord(a);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((probe<V, D>), dim3(128, 2), dim3(V == 2 ? 576 : 512), 3 * 384 * 128, 0, A, B, lda_b, steps, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 50 * 1000.f;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    fprintf(stderr, "start\n");
    const int lda_b = 1088 * 2;
    unsigned char *A, *B, *trash; unsigned* sink;
    const size_t tb = (size_t)512 << 20;
    (void)hipMalloc(&A, (size_t)2 * 4096 * lda_b); (void)hipMalloc(&B, (size_t)2 * 1024 * lda_b); (void)hipMalloc(&sink, 64); (void)hipMalloc(&trash, tb);
    (void)hipMemset(A, 1, (size_t)2 * 4096 * lda_b); (void)hipMemset(B, 1, (size_t)2 * 1024 * lda_b);
    fprintf(stderr, "allocated: %s\n", hipGetErrorString(hipGetLastError()));
#define ROW(name, V, D) { const float w8 = run_warm<V, D>(A, B, lda_b, 8, sink), w16 = run_warm<V, D>(A, B, lda_b, 16, sink);            \
        const float c16 = run<V, D>(A, B, lda_b, 16, sink, trash, tb);                                                                     \
        printf("%-44s warm: 8 steps %6.2f us, 16 steps %6.2f us => %.3f us/step | cold 16 steps %6.2f us\n", name, w8, w16, (w16 - w8) / 8, c16); }
    for (int rep = 0; rep < 2; ++rep) {
        ROW("V0 streaming (k_grouped)", 0, 0)
        ROW("V1 same K tile every step (L2 hits)", 1, 0)
        ROW("V2 + prefetch wave, D=2", 2, 2)
        ROW("V2 + prefetch wave, D=4", 2, 4)
        ROW("V2 + prefetch wave, D=6", 2, 6)
        ROW("V3 + in-wave prefetch after the DMA, D=2", 3, 2)
        ROW("V3 + in-wave prefetch after the DMA, D=4", 3, 4)
    }
    return 0;
}
