// Diagnostic: prints what v_permlane16_swap does to two registers (the lane map the bf16 register epilogue relies on).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    const auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = s[0]; out[64 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) printf("row %d: first %u..%u  second %u..%u\n", r, h[16 * r], h[16 * r + 15], h[64 + 16 * r], h[64 + 16 * r + 15]);
    // expected by avae_kernels.hip::regep_store: first = {a.row0, b.row0, a.row2, b.row2}, second = {a.row1, b.row1, a.row3, b.row3}
    const bool ok = h[0] == 1000 && h[16] == 2000 && h[32] == 1032 && h[48] == 2032 && h[64] == 1016 && h[80] == 2016 && h[96] == 1048 && h[112] == 2048;
    printf("permlane16_swap map %s\n", ok ? "OK" : "UNEXPECTED");
    return ok ? 0 : 1;
}
