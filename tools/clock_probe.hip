// Diagnostic (not part of the product): the shader clock under load.  s_memtime counts shader-clock cycles, s_memrealtime a constant
// 100 MHz: their ratio over a long loop is the clock the wave actually ran at.  Variants: a wave that only sleeps; every CU busy with
// dependent bf16 MFMAs (8 waves per CU, four independent accumulators per wave); MFMAs + LDS reads; MFMAs while other workgroups
// stream 1 GiB through HBM.  C4's GEMM fractions are quoted against the nominal 2.5 PFLOP/s = 2.4 GHz x 256 CUs x 4096 flop/clk.
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void __launch_bounds__(512) burn(long long* out, int iters, int mode, const float4* stream, size_t n_stream) {
    __shared__ float lds[8192];
    long long a0, b0, a1, b1;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = 1.0f;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a0), "=s"(b0) :: "memory");
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.001f * (threadIdx.x & 7)); y[e] = (__bf16)1.0f; }
    float acc = 0.0f;
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) asm volatile("s_sleep 8" ::: "memory");
    } else if (mode == 3 && blockIdx.x >= gridDim.x / 2) {          // half of the workgroups stream from HBM
        const size_t st = (size_t)(gridDim.x / 2) * blockDim.x;
        for (size_t k = (size_t)(blockIdx.x - gridDim.x / 2) * blockDim.x + threadIdx.x; k < n_stream; k += st) { const float4 v = stream[k]; acc += v.x; }
    } else {
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c3, 0, 0, 0);
            if (mode == 2) acc += lds[(threadIdx.x * 4 + i) & 8191] + lds[(threadIdx.x * 4 + 2048 + i) & 8191];
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a1), "=s"(b1) :: "memory");
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = a1 - a0; out[2 * blockIdx.x + 1] = b1 - b0; }
    if (c0[0] + c1[1] + c2[2] + c3[3] + acc == 12345.678f) out[0] = 0;
}

int main() {
    const int nwg = 256;
    long long* out; float4* big;
    const size_t nbig = ((size_t)1 << 30) / 16;
    hipMalloc(&out, nwg * 16); hipMalloc(&big, nbig * 16); hipMemset(big, 0, nbig * 16);
    long long h[2 * nwg];
    const char* names[4] = {"sleeping waves", "bf16 MFMAs on every CU (8 waves/CU)", "MFMAs + LDS reads", "MFMAs on half the CUs, HBM stream on the other half"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(burn, dim3(nwg), dim3(512), 0, 0, out, mode == 0 ? 20000 : 60000, mode, big, nbig);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double cyc = 0, rt = 0;
            for (int i = 0; i < nwg / 2; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }       // the MFMA half in mode 3
            std::printf("%-52s %7.0f MHz   (%.1f us per workgroup)\n", names[mode], cyc / rt * 100.0, rt / (nwg / 2) / 100.0);
        }
    return 0;
}
