// Diagnostic (not part of the product): are two hipGraphLaunch calls on ONE stream ordered as stream semantics promise -- can the ROOT
// kernel of the second launch start before the last kernel of the first has finished?  And does hipGraphExecKernelNodeSetParams on an
// exec whose previous launch is still in flight leak into that launch?
//   test 1  graph A = {slow_fill(buf, v)} (every workgroup spins ~30 us, then writes v), graph B = {check(buf, v)}; A, B, A, B ... with v
//           patched into both before every pair: check must always see v everywhere.
//   test 2  ONE graph {stamp(out[i], v)} launched 2000 times back to back, v and the output slot patched before every launch: slot i
//           must hold i.
// build: hipcc --offload-arch=gfx950 -O3 tools/graph_order_probe.hip -o /tmp/gop ; run: /tmp/gop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void slow_fill(unsigned* buf, int n, unsigned v, int spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = v;
}
__global__ void check(const unsigned* buf, int n, unsigned v, unsigned* bad) {
    unsigned b = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b += buf[i] != v;
    if (b) atomicAdd(bad, b);
}
__global__ void stamp(unsigned* out, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = v; }

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipStream_t s; OK(hipStreamCreate(&s));
    const int n = 1 << 20;
    unsigned *buf, *bad, *out;
    OK(hipMalloc(&buf, n * 4)); OK(hipMalloc(&bad, 4)); OK(hipMalloc(&out, 4096 * 4));
    OK(hipMemset(buf, 0, n * 4)); OK(hipMemset(bad, 0, 4)); OK(hipMemset(out, 0xff, 4096 * 4));
    // ---- test 1
    {
        hipGraph_t ga, gb; hipGraphExec_t ea, eb; hipGraphNode_t na, nb; size_t one = 1;
        unsigned v = 0; int nn = n, spin = 3000;     // 100 MHz wall clock: 30 us
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(slow_fill, dim3(256), dim3(256), 0, s, buf, nn, v, spin);
        OK(hipStreamEndCapture(s, &ga)); OK(hipGraphInstantiate(&ea, ga, nullptr, nullptr, 0)); OK(hipGraphGetRootNodes(ga, &na, &one));
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, s, (const unsigned*)buf, nn, v, bad);
        OK(hipStreamEndCapture(s, &gb)); OK(hipGraphInstantiate(&eb, gb, nullptr, nullptr, 0)); one = 1; OK(hipGraphGetRootNodes(gb, &nb, &one));
        for (int it = 1; it <= 400; ++it) {
            v = (unsigned)it;
            void* pa[4] = {&buf, &nn, &v, &spin};
            hipKernelNodeParams kp; memset(&kp, 0, sizeof(kp));
            kp.func = (void*)slow_fill; kp.gridDim = dim3(256); kp.blockDim = dim3(256); kp.kernelParams = pa;
            OK(hipGraphExecKernelNodeSetParams(ea, na, &kp));
            const unsigned* cb = buf;
            void* pb[4] = {&cb, &nn, &v, &bad};
            memset(&kp, 0, sizeof(kp));
            kp.func = (void*)check; kp.gridDim = dim3(256); kp.blockDim = dim3(256); kp.kernelParams = pb;
            OK(hipGraphExecKernelNodeSetParams(eb, nb, &kp));
            OK(hipGraphLaunch(ea, s));
            OK(hipGraphLaunch(eb, s));
        }
        OK(hipStreamSynchronize(s));
        unsigned hb = 0; OK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("test 1 (root kernel of the next graph launch vs the previous launch's kernel, args patched per pair): %u stale elements seen in 400 pairs\n", hb);
    }
    // ---- test 2
    {
        hipGraph_t g; hipGraphExec_t e; hipGraphNode_t nd; size_t one = 1;
        unsigned v = 0; unsigned* o = out;
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, o, v);
        OK(hipStreamEndCapture(s, &g)); OK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0)); OK(hipGraphGetRootNodes(g, &nd, &one));
        for (int rep = 0; rep < 5; ++rep) {
            for (int i = 0; i < 2000; ++i) {
                v = (unsigned)(rep * 2000 + i); o = out + i;
                void* p[2] = {&o, &v};
                hipKernelNodeParams kp; memset(&kp, 0, sizeof(kp));
                kp.func = (void*)stamp; kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.kernelParams = p;
                OK(hipGraphExecKernelNodeSetParams(e, nd, &kp));
                OK(hipGraphLaunch(e, s));
            }
            OK(hipStreamSynchronize(s));
            std::vector<unsigned> h(2000);
            OK(hipMemcpy(h.data(), out, 2000 * 4, hipMemcpyDeviceToHost));
            int wrong = 0, first = -1;
            for (int i = 0; i < 2000; ++i) if (h[i] != (unsigned)(rep * 2000 + i)) { if (first < 0) first = i; ++wrong; }
            printf("test 2 rep %d (exec patched while earlier launches of it are in flight): %d of 2000 slots wrong%s\n", rep, wrong, wrong ? " (a launch ran with a LATER call's arguments)" : "");
            if (wrong) printf("        first wrong slot %d holds %u\n", first, h[first]);
        }
    }
    return 0;
}
