// Diagnostic (not part of the product): what a launch's FIRST loads cost after a kernel boundary, by where the line is.
// C2's launches spend 0.5 us between kernel entry and the arrival of their work item (a scalar load from the kernel-argument
// segment) and another 0.5 us on the first operand tile: two cold round trips in series, 1 us of a ~3.3-us launch.  This probe
// asks whether the PREVIOUS launch could warm those lines: does a line touched by kernel A survive the boundary in the XCD's L2
// (scalar and vector path), and what do the Infinity Cache and HBM cost?
//   reader: every workgroup times ONE dependent load (scalar s_load_dword or vector global_load_dword) of its own 128-byte line
//   cases : cold (after a 1-GiB sweep), mall (after a 64-MiB sweep: out of L2, in the Infinity Cache), touched by the previous
//           kernel through the vector path / the scalar path on the SAME workgroup slot (same XCD), touched by another XCD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/boundary_probe.hip -o gpurun_out/boundary_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kWG = 256;              // one workgroup per CU, dealt round-robin over the 8 XCDs
constexpr int kLineInts = 32;         // 128 bytes

__global__ void sweep(float4* p, size_t n) {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < n; i += st) { float4 v = p[i]; v.x += 1.0f; p[i] = v; }
}

// shift: which workgroup's line this workgroup touches (0: its own -> same XCD as the reader; 1: the next one -> another XCD)
__global__ void touch_vec(const int* buf, int shift, int* sink) {
    if (threadIdx.x == 0) {
        const int v = buf[((blockIdx.x + shift) % kWG) * kLineInts];
        if (v == 0x7fffffff) sink[0] = v;
    }
}
__global__ void touch_scalar(const int* buf, int shift, int* sink) {
    const int v = __builtin_nontemporal_load(buf + ((blockIdx.x + shift) % kWG) * kLineInts + 0 * (int)threadIdx.x);   // uniform address
    if (v == 0x7fffffff && threadIdx.x == 0) sink[0] = v;
}

// the producer side of an activation hand-over: the previous kernel WRITES the line (32 lanes x 4 bytes), plain or written through
__global__ void write_plain(int* buf, int shift) {
    if (threadIdx.x < 32) buf[((blockIdx.x + shift) % kWG) * kLineInts + threadIdx.x] = (int)threadIdx.x;
}
__global__ void write_through(int* buf, int shift) {
    if (threadIdx.x < 32) {
        int* p = buf + ((blockIdx.x + shift) % kWG) * kLineInts + threadIdx.x;
        const int v = (int)threadIdx.x;
        asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    }
}

__global__ void read_scalar(const int* buf, long long* out, int* sink) {
    const int* p = buf + blockIdx.x * kLineInts;
    long long t0, t1;
    int v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; if (v == 0x7fffffff) sink[0] = v; }
}
__global__ void read_vec(const int* buf, long long* out, int* sink) {
    const int* p = buf + blockIdx.x * kLineInts + (threadIdx.x & 31);
    long long t0, t1;
    int v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; if (v == 0x7fffffff) sink[0] = v; }
}

// n64: how many consecutive 64-byte lines one burst of s_load_dwordx16 fetches (a 256-byte work item = 4)
template <int N64>
__global__ void read_scalar_burst(const int* buf, long long* out, int* sink) {
    typedef int v16 __attribute__((ext_vector_type(16)));
    const int* p = buf + blockIdx.x * 4 * kLineInts;           // 512 bytes apart
    long long t0, t1;
    v16 a, b, c, d;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (N64 == 1) asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(p) : "memory");
    else asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                      : "=s"(a), "=s"(b), "=s"(c), "=s"(d) : "s"(p) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    int v = a[0];
    if (N64 == 4) v += b[1] + c[2] + d[3];
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; if (v == 0x7fffffff) sink[0] = v; }
}
// the launch's own kernel-argument segment: a 320-byte by-value argument, timed like k_grouped's work-item fetch (4 lines from
// offset 64 on), first thing in the kernel
struct Blob { int w[80]; };
__global__ void read_kernarg(Blob blob, long long* out, int* sink) {
    typedef int v16 __attribute__((ext_vector_type(16)));
    const void* p = (const void*)(unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    long long t0, t1;
    v16 a, b, c, d;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_load_dwordx16 %0, %4, 0x40\n\ts_load_dwordx16 %1, %4, 0x80\n\ts_load_dwordx16 %2, %4, 0xc0\n\ts_load_dwordx16 %3, %4, 0x100\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(a), "=s"(b), "=s"(c), "=s"(d) : "s"(p) : "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    const int v = a[0] + b[1] + c[2] + d[3];
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; if (v == 0x7fffffff) sink[0] = v; }
}

__global__ void calibrate(long long* out) {      // s_memtime ticks per s_memrealtime tick (10 ns)
    long long a0, a1, b0, b1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a0), "=s"(b0) :: "memory");
    for (int i = 0; i < 2000; ++i) asm volatile("s_sleep 10" ::: "memory");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a1), "=s"(b1) :: "memory");
    if (threadIdx.x == 0) { out[0] = a1 - a0; out[1] = b1 - b0; }
}

int main() {
    int* buf; long long* out; int* sink; float4* big;
    const size_t big_bytes = (size_t)1 << 30;
    OK(hipMalloc(&buf, kWG * kLineInts * 4 * 64));       // 64 sets of lines, one per trial: a trial never re-reads a line it warmed itself
    OK(hipMalloc(&out, kWG * 8)); OK(hipMalloc(&sink, 4)); OK(hipMalloc(&big, big_bytes));
    OK(hipMemset(buf, 0, kWG * kLineInts * 4 * 64)); OK(hipMemset(big, 0, big_bytes));
    hipStream_t s; OK(hipStreamCreate(&s));
    std::vector<long long> h(kWG);
    int trial = 0;
    auto report = [&](const char* name) {
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), out, kWG * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // s_memtime counts at 100 MHz on this part?  print raw ticks and let the reader scale: measured against a known delay below
        std::printf("%-58s median %6lld  p10 %6lld  p90 %6lld ticks\n", name, h[kWG / 2], h[kWG / 10], h[kWG * 9 / 10]);
    };
    auto lines = [&]() { return buf + (size_t)(trial++ % 64) * kWG * kLineInts; };
    for (int rep = 0; rep < 2; ++rep) {
        std::printf("---- pass %d\n", rep);
        for (int vec = 0; vec < 2; ++vec) {
            auto reader = [&](const int* b) { if (vec) hipLaunchKernelGGL(read_vec, dim3(kWG), dim3(64), 0, s, b, out, sink); else hipLaunchKernelGGL(read_scalar, dim3(kWG), dim3(64), 0, s, b, out, sink); };
            const char* path = vec ? "vector load" : "scalar load";
            char name[128];
            const int* b = lines();
            hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, big_bytes / 16);
            reader(b); std::snprintf(name, sizeof name, "%s, cold (behind a 1-GiB sweep)", path); report(name);
            b = lines();
            hipLaunchKernelGGL(touch_vec, dim3(kWG), dim3(64), 0, s, b, 0, sink);
            hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, ((size_t)64 << 20) / 16);
            reader(b); std::snprintf(name, sizeof name, "%s, touched, then a 64-MiB sweep (Infinity Cache)", path); report(name);
            b = lines();
            hipLaunchKernelGGL(touch_vec, dim3(kWG), dim3(64), 0, s, b, 0, sink);
            reader(b); std::snprintf(name, sizeof name, "%s, previous kernel touched it (vector, same XCD)", path); report(name);
            b = lines();
            hipLaunchKernelGGL(touch_scalar, dim3(kWG), dim3(64), 0, s, b, 0, sink);
            reader(b); std::snprintf(name, sizeof name, "%s, previous kernel touched it (scalar, same XCD)", path); report(name);
            b = lines();
            hipLaunchKernelGGL(touch_vec, dim3(kWG), dim3(64), 0, s, b, 1, sink);
            reader(b); std::snprintf(name, sizeof name, "%s, previous kernel touched it (vector, OTHER XCD)", path); report(name);
            b = lines();
            reader(b); reader(b); std::snprintf(name, sizeof name, "%s, the same reader twice (second run)", path); report(name);
        }
    }
    std::printf("---- a line WRITTEN by the previous kernel, read by the vector path\n");
    for (int rep = 0; rep < 2; ++rep) {
        int* b = const_cast<int*>(lines());
        hipLaunchKernelGGL(write_plain, dim3(kWG), dim3(64), 0, s, b, 0);
        hipLaunchKernelGGL(read_vec, dim3(kWG), dim3(64), 0, s, b, out, sink); report("plain stores, same XCD");
        b = const_cast<int*>(lines());
        hipLaunchKernelGGL(write_through, dim3(kWG), dim3(64), 0, s, b, 0);
        hipLaunchKernelGGL(read_vec, dim3(kWG), dim3(64), 0, s, b, out, sink); report("write-through (sc0 sc1) stores, same XCD");
        b = const_cast<int*>(lines());
        hipLaunchKernelGGL(write_plain, dim3(kWG), dim3(64), 0, s, b, 1);
        hipLaunchKernelGGL(read_vec, dim3(kWG), dim3(64), 0, s, b, out, sink); report("plain stores, OTHER XCD");
        b = const_cast<int*>(lines());
        hipLaunchKernelGGL(write_through, dim3(kWG), dim3(64), 0, s, b, 1);
        hipLaunchKernelGGL(read_vec, dim3(kWG), dim3(64), 0, s, b, out, sink); report("write-through (sc0 sc1) stores, OTHER XCD");
    }
    std::printf("---- bursts of s_load_dwordx16 (what a 256-byte work item costs)\n");
    int* buf2; OK(hipMalloc(&buf2, kWG * 512 * 8)); OK(hipMemset(buf2, 0, kWG * 512 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        const int* b1 = buf2 + (size_t)(2 * rep) * kWG * 128;
        const int* b4 = buf2 + (size_t)(2 * rep + 1) * kWG * 128;
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, big_bytes / 16);
        hipLaunchKernelGGL(read_scalar_burst<1>, dim3(kWG), dim3(64), 0, s, b1, out, sink); report("1 x 64 B, cold");
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, big_bytes / 16);
        hipLaunchKernelGGL(read_scalar_burst<4>, dim3(kWG), dim3(64), 0, s, b4, out, sink); report("4 x 64 B, cold");
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, ((size_t)64 << 20) / 16);
        hipLaunchKernelGGL(read_scalar_burst<1>, dim3(kWG), dim3(64), 0, s, b1, out, sink); report("1 x 64 B, Infinity Cache");
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, ((size_t)64 << 20) / 16);
        hipLaunchKernelGGL(read_scalar_burst<4>, dim3(kWG), dim3(64), 0, s, b4, out, sink); report("4 x 64 B, Infinity Cache");
        hipLaunchKernelGGL(read_scalar_burst<4>, dim3(kWG), dim3(64), 0, s, b4, out, sink); report("4 x 64 B, L2 (previous kernel read them)");
    }
    std::printf("---- the launch's own kernel-argument segment (4 x 64 B of a by-value argument), eager and graph replay\n");
    Blob blob; for (int i = 0; i < 80; ++i) blob.w[i] = i;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, ((size_t)64 << 20) / 16);
        hipLaunchKernelGGL(read_kernarg, dim3(kWG), dim3(64), 0, s, blob, out, sink); report("eager launch, behind a 64-MiB sweep");
        hipLaunchKernelGGL(touch_vec, dim3(kWG), dim3(64), 0, s, buf, 0, sink);
        hipLaunchKernelGGL(read_kernarg, dim3(kWG), dim3(64), 0, s, blob, out, sink); report("eager launch, behind a tiny kernel");
    }
    {
        hipGraph_t g; hipGraphExec_t ge;
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, ((size_t)64 << 20) / 16);
        hipLaunchKernelGGL(read_kernarg, dim3(kWG), dim3(64), 0, s, blob, out, sink);
        OK(hipStreamEndCapture(s, &g));
        OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) { OK(hipGraphLaunch(ge, s)); report("graph replay, behind a 64-MiB sweep"); }
        hipGraph_t g2; hipGraphExec_t ge2;
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(touch_vec, dim3(kWG), dim3(64), 0, s, buf, 0, sink);
        hipLaunchKernelGGL(read_kernarg, dim3(kWG), dim3(64), 0, s, blob, out, sink);
        OK(hipStreamEndCapture(s, &g2));
        OK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) { OK(hipGraphLaunch(ge2, s)); report("graph replay, behind a tiny kernel"); }
    }
    hipLaunchKernelGGL(calibrate, dim3(1), dim3(64), 0, s, out);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), out, 16, hipMemcpyDeviceToHost);
    std::printf("(s_memtime: %lld ticks in %lld x 10 ns -> %.2f ns per tick)\n", h[0], h[1], 10.0 * (double)h[1] / (double)h[0]);
    return 0;
}
