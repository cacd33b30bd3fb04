// Diagnostic micro-benchmark (not part of the product): what bounds the operand fill of the 256x128 8-wave tile?
// One 512-thread workgroup per CU, 64 K-steps of a [256 + 128 rows] x 128 B tile per workgroup, no MFMA:
//   V0  both operands by LDS-DMA (48 KB per step and CU)            -- what k_grouped does
//   V1  A by LDS-DMA (32 KB), B straight into registers, every wave its 64 x 64 B-fragment set (8 x 16 B per lane:
//       64 KB per step and CU through the vector L1, 4x redundant across the row waves)
//   V2  A by LDS-DMA only (32 KB)                                     -- floor of V1
// build: hipcc --offload-arch=gfx950 -O3 tools/fill_probe.hip -o /tmp/fill_probe ; run: /tmp/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const __attribute__((address_space(1))) void* gp_t;
typedef __attribute__((address_space(3))) void* lp_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

template <int V>
__global__ void __launch_bounds__(512) probe(const unsigned char* A, const unsigned char* B, int lda_b, int steps, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;                       // 16 x 8 tiles of a [4096 x 1024] problem, twice (two items)
    const int tm = (tile & 127) >> 3, tn = tile & 7;
    const unsigned char* Ag = A + (size_t)(tile >> 7) * 4096 * lda_b + (size_t)tm * 256 * lda_b;
    const unsigned char* Bg = B + (size_t)(tile >> 7) * 1024 * lda_b + (size_t)tn * 128 * lda_b;
    constexpr int STAGE = (256 + 128) * 128, RING = 3;
    constexpr int NCH = V == 0 ? 6 : 4;                 // 1-KiB pieces per wave and step
    const unsigned char* src[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int r = (c * 8 + wave) * 8 + (lane >> 3);
        src[c] = (r < 256 ? Ag + (size_t)r * lda_b : Bg + (size_t)(r - 256) * lda_b) + (lane & 7) * 16;
    }
    // B fragments: lane (l & 15) -> row, (l >> 4) -> 16-byte k chunk; wave column wc = wave & 1 owns 64 rows
    const unsigned char* bsrc = Bg + (size_t)((wave & 1) * 64 + (lane & 15)) * lda_b + (lane >> 4) * 16;
    unsigned acc = 0;
    auto dma = [&](int kt, int buf) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            __builtin_amdgcn_global_load_lds((gp_t)(src[c] + (size_t)kt * 128), (lp_t)(smem + buf * STAGE + (c * 8 + wave) * 1024), 16, 0, 0);
    };
    // B loads are inline asm (the compiler would guard an ordinary load used beside LDS-DMA with vmcnt(0), draining the ring) and
    // are issued one step ahead, BEFORE that iteration's refill DMA, so that the counted wait retires them with the tile.
    u32x4 fc[8], fn[8];
    auto bload = [&](int kt, u32x4 (&f)[8]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* p0 = bsrc + (size_t)j * 16 * lda_b + (size_t)kt * 128;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f[2 * j]) : "v"(p0));
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(f[2 * j + 1]) : "v"(p0));
        }
    };
    dma(0, 0);
    if (V == 1) bload(0, fc);
    dma(1, 1);
    int buf = 0;
    for (int kt = 0; kt < steps; ++kt) {
        if (kt + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        acc += smem[buf * STAGE + tid * 16];             // touch the stage
        if (V == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(fc[j])); acc += fc[j][0] ^ fc[j][3]; }
            if (kt + 1 < steps) bload(kt + 1, fn);
        }
        const int fill = buf == 0 ? RING - 1 : buf - 1;
        if (kt + 2 < steps) dma(kt + 2, fill);
        if (V == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) fc[j] = fn[j];
        }
        buf = buf + 1 == RING ? 0 : buf + 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int V> static float run(const unsigned char* A, const unsigned char* B, int lda_b, int steps, unsigned* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(512), 3 * 384 * 128, 0, A, B, lda_b, steps, sink);
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(512), 3 * 384 * 128, 0, A, B, lda_b, steps, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 50 * 1000.f;
}

int main() {
    const int lda_b = 1088 * 2, steps = 17;               // K = 1088 bf16 as in C4's hidden layers
    unsigned char *A, *B; unsigned* sink;
    hipMalloc(&A, (size_t)2 * 4096 * lda_b); hipMalloc(&B, (size_t)2 * 1024 * lda_b); hipMalloc(&sink, 64);
    hipMemset(A, 1, (size_t)2 * 4096 * lda_b); hipMemset(B, 1, (size_t)2 * 1024 * lda_b);
    for (int rep = 0; rep < 2; ++rep) {
        const float t0 = run<0>(A, B, lda_b, steps, sink), t1 = run<1>(A, B, lda_b, steps, sink), t2 = run<2>(A, B, lda_b, steps, sink);
        printf("17 steps: V0 both by LDS-DMA %.2f us | V1 A by DMA + B to registers %.2f us | V2 A by DMA only %.2f us\n", t0, t1, t2);
    }
    const float l0 = run<0>(A, B, lda_b, 8, sink), l1 = run<1>(A, B, lda_b, 8, sink), l2 = run<2>(A, B, lda_b, 8, sink);
    printf(" 8 steps: V0 %.2f us | V1 %.2f us | V2 %.2f us  => per step: V0 %.3f V1 %.3f V2 %.3f us\n", l0, l1, l2,
           (run<0>(A, B, lda_b, 17, sink) - l0) / 9, (run<1>(A, B, lda_b, 17, sink) - l1) / 9, (run<2>(A, B, lda_b, 17, sink) - l2) / 9);
    return 0;
}
