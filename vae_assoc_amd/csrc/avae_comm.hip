// Gradient exchange kernels of libavae (gfx950): the one-shot all-reduce over hipIpc peers and the bf16 wire format.
// The reference has nothing here (one tf.InteractiveSession, vae_assoc.py:66); SURVEY.md section 5 / 8(e) is the specification:
// one flat gradient buffer, SUM over the ranks, direct reduce-scatter + all-gather over the fully connected xGMI mesh instead
// of a ring that one link bounds, optionally bf16 on the wire.  Protocol and memory layout: avae_device.h (IpcArgs).
#include "avae_device.h"

namespace avae {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// System-scope (sc0 sc1) accesses: stores are written through to the memory they address -- a peer's HBM over xGMI, or the own
// exchange block -- and loads are served from memory, never from this XCD's L2 or the CU's L1.  The exchange blocks are uncached
// allocations as well; the scope bits make the protocol independent of how an imported mapping is cached.
__device__ __forceinline__ void store16_sys(void* p, u32x4 v) {
    // (s_nop 1 inside the string: hipcc does not pad an asm statement's hazards, and its next instruction may overwrite the data
    // registers before a 16-byte store has read them -- the first 8 bytes of 2 % of the granules carried an address once)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
// Loads are compiler-visible (two 8-byte relaxed system-scope atomic loads = global_load_dwordx2 sc0 sc1): an inline-asm load whose
// wait is a separate statement is NOT safe -- the compiler believes the value is there when the asm statement ends and may copy the
// destination registers before the data has landed (seen here: 2 % of the granules came back with a stale first half once the
// surrounding loops were unrolled further; the parity tests on real gradients had passed).
__device__ __forceinline__ u32x4 load16_sys(const void* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned bf16_rne(float f) {           // round to nearest even; NaN stays NaN (cast form, see the guide)
    const __bf16 b = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_up(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ u32x4 pack8(const f32x4 a, const f32x4 b) {
    u32x4 r;
    r.x = bf16_rne(a.x) | (bf16_rne(a.y) << 16); r.y = bf16_rne(a.z) | (bf16_rne(a.w) << 16);
    r.z = bf16_rne(b.x) | (bf16_rne(b.y) << 16); r.w = bf16_rne(b.z) | (bf16_rne(b.w) << 16);
    return r;
}
__device__ __forceinline__ void unpack8(const u32x4 r, f32x4& a, f32x4& b) {
    a = f32x4{bf16_up(r.x & 0xFFFFu), bf16_up(r.x >> 16), bf16_up(r.y & 0xFFFFu), bf16_up(r.y >> 16)};
    b = f32x4{bf16_up(r.z & 0xFFFFu), bf16_up(r.z >> 16), bf16_up(r.w & 0xFFFFu), bf16_up(r.w >> 16)};
}

// All stores of this workgroup have left: every storing wave drains its own (write-through) stores -- vmcnt counts a system-scope
// store until the memory it addresses has acknowledged it --, then the workgroup meets; the flag stores follow (guide: R1 form).
__device__ __forceinline__ void drain_block() {
    wait_vm0();
    __syncthreads();
}
// Wave 0: lane r waits for flag[r] == seq (r != me), bounded.  Returns false after a timeout (error word raised).
__device__ __forceinline__ bool wait_flags(const unsigned* flags, unsigned seq, int world, int me, unsigned long long timeout,
                                           unsigned* err, unsigned code) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) ok_s = 1;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int r = (int)threadIdx.x;
        if (r < world && r != me) {
            const unsigned long long t0 = wall_clock64();
            // (a flag only ever grows: >= seq also accepts a peer that is already one call ahead on this word -- it cannot be,
            // the protocol keeps peers within one call of each other per workgroup, but the comparison costs nothing)
            while ((int)(__hip_atomic_load(flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > timeout) {
                    __hip_atomic_fetch_or(err, code | (1u << (8 + r)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok_s = 0;
                    break;
                }
            }
        }
    }
    __syncthreads();
    return ok_s != 0;
}

template <bool BF16>
__global__ void __launch_bounds__(kIpcThreads) k_ipc_allreduce(IpcArgs a) {
    constexpr int GB = BF16 ? 16 : 32;                       // wire bytes per granule of 8 floats
    const int w = blockIdx.x, tid = threadIdx.x, me = a.rank, N = a.world;
    unsigned char* mine = a.peer[me];
    unsigned* seqp = reinterpret_cast<unsigned*>(mine + a.off_seq) + w;
    unsigned* err = reinterpret_cast<unsigned*>(mine + a.off_err);
    const unsigned seq = *seqp + 1u;                         // this call's number (the same on every rank: they make the same calls)
    if (N == 1) { if (tid == 0) *seqp = seq; return; }       // the sum over one rank is the gradient as it stands (cost included)
    const long long S = (a.granules + N - 1) / N;            // granules per shard
    const long long C = (S + a.blocks - 1) / a.blocks;       // granules per (shard, workgroup) chunk
    auto shard_len = [&](int j) { const long long lo = (long long)j * S; return lo >= a.granules ? 0ll : (a.granules - lo < S ? a.granules - lo : S); };
    auto chunk_len = [&](int j) { const long long sl = shard_len(j), lo = (long long)w * C; return lo >= sl ? 0ll : (sl - lo < C ? sl - lo : C); };
    float* const g = a.g + a.off;
    const long long c0 = (long long)w * C;                   // first granule of this workgroup's chunk inside a shard

    // ---- phase 1: push my values of shard j to rank j's slot `me` (peers in ring order from me + 1: the links fill evenly)
    for (int d = 1; d < N; ++d) {
        const int j = (me + d) % N;
        const long long n = chunk_len(j);
        unsigned char* dst = a.peer[j] + a.off_slots + (long long)me * a.slot_stride + c0 * GB;
        const float* src = g + ((long long)j * S + c0) * 8;
        constexpr int U1 = 4;                                   // granules per thread and trip: their loads are in flight together
        for (long long i0 = tid; i0 < n; i0 += (long long)U1 * kIpcThreads) {
            f32x4 lo[U1], hi[U1];
#pragma unroll
            for (int u = 0; u < U1; ++u) {
                const long long i = i0 + (long long)u * kIpcThreads;
                if (i < n) { lo[u] = *reinterpret_cast<const f32x4*>(src + i * 8); hi[u] = *reinterpret_cast<const f32x4*>(src + i * 8 + 4); }
            }
#pragma unroll
            for (int u = 0; u < U1; ++u) {
                const long long i = i0 + (long long)u * kIpcThreads;
                if (i >= n) continue;
                if (BF16) store16_sys(dst + i * 16, pack8(lo[u], hi[u]));
                else { store16_sys(dst + i * 32, __builtin_bit_cast(u32x4, lo[u])); store16_sys(dst + i * 32 + 16, __builtin_bit_cast(u32x4, hi[u])); }
            }
        }
        if (w == 0 && tid == 0 && a.cost_idx >= 0)            // the local cost, fp32, to every peer (workgroup 0's flags cover it)
            __hip_atomic_store(reinterpret_cast<float*>(a.peer[j] + a.off_cost) + me, a.g[a.cost_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    drain_block();
    if (tid < N && tid != me)
        __hip_atomic_store(reinterpret_cast<unsigned*>(a.peer[tid] + a.off_flag1) + (long long)w * kMaxWorld + me, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    // ---- phase 2: my shard = the sum over the ranks in rank order; the sum goes to every rank's result area and to my own g
    bool ok = wait_flags(reinterpret_cast<const unsigned*>(mine + a.off_flag1) + (long long)w * kMaxWorld, seq, N, me, a.timeout_ticks, err, 1u);
    {
        const long long n = ok ? chunk_len(me) : 0;
        float* own = g + ((long long)me * S + c0) * 8;
        const unsigned char* slots = mine + a.off_slots + c0 * GB;
        const long long res_off = a.off_res + ((long long)me * S + c0) * GB;
        constexpr int U2 = BF16 ? 4 : 2;                      // granules per thread and trip: all their slot loads are in flight together
        for (long long i0 = tid; i0 < n; i0 += (long long)U2 * kIpcThreads) {
            u32x4 raw[U2][kMaxWorld][BF16 ? 1 : 2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const long long i = i0 + (long long)u * kIpcThreads;
                if (i < n) {
#pragma unroll
                    for (int r = 0; r < kMaxWorld; ++r)
                        if (r < N && r != me) {
                            raw[u][r][0] = load16_sys(slots + (long long)r * a.slot_stride + i * GB);
                            if (!BF16) raw[u][r][BF16 ? 0 : 1] = load16_sys(slots + (long long)r * a.slot_stride + i * GB + 16);
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const long long i = i0 + (long long)u * kIpcThreads;
                if (i >= n) continue;
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                const f32x4 mlo = *reinterpret_cast<const f32x4*>(own + i * 8), mhi = *reinterpret_cast<const f32x4*>(own + i * 8 + 4);
#pragma unroll
                for (int r = 0; r < kMaxWorld; ++r)
                    if (r < N) {
                        f32x4 xl, xh;
                        if (r == me) { xl = mlo; xh = mhi; }
                        else {
                            if (BF16) unpack8(raw[u][r][0], xl, xh);
                            else { xl = __builtin_bit_cast(f32x4, raw[u][r][0]); xh = __builtin_bit_cast(f32x4, raw[u][r][BF16 ? 0 : 1]); }
                        }
                        lo += xl; hi += xh;
                    }
                if (BF16) {                                   // every rank keeps the SAME (wire-rounded) sum: replicas stay bit-identical
                    const u32x4 p = pack8(lo, hi);
                    unpack8(p, lo, hi);
                    for (int d = 1; d < N; ++d) store16_sys(a.peer[(me + d) % N] + res_off + i * 16, p);
                } else {
                    for (int d = 1; d < N; ++d) {
                        unsigned char* q = a.peer[(me + d) % N] + res_off + i * 32;
                        store16_sys(q, __builtin_bit_cast(u32x4, lo)); store16_sys(q + 16, __builtin_bit_cast(u32x4, hi));
                    }
                }
                *reinterpret_cast<f32x4*>(own + i * 8) = lo; *reinterpret_cast<f32x4*>(own + i * 8 + 4) = hi;
            }
        }
        if (w == 0 && tid == 0 && a.cost_idx >= 0 && ok) {    // costs: every rank adds all of them in rank order itself
            float c = 0.f;
            const float* cs = reinterpret_cast<const float*>(mine + a.off_cost);
            for (int r = 0; r < N; ++r) c += r == me ? a.g[a.cost_idx] : __hip_atomic_load(cs + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a.g[a.cost_idx] = c;
        }
    }
    drain_block();
    if (tid < N && tid != me)
        __hip_atomic_store(reinterpret_cast<unsigned*>(a.peer[tid] + a.off_flag2) + (long long)w * kMaxWorld + me, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    // ---- phase 3: the other shards' sums, from my result area into g
    ok = wait_flags(reinterpret_cast<const unsigned*>(mine + a.off_flag2) + (long long)w * kMaxWorld, seq, N, me, a.timeout_ticks, err, 2u) && ok;
    if (ok)
        for (int d = 1; d < N; ++d) {
            const int j = (me + d) % N;
            const long long n = chunk_len(j);
            const unsigned char* src = mine + a.off_res + ((long long)j * S + c0) * GB;
            float* dst = g + ((long long)j * S + c0) * 8;
            constexpr int U3 = 4, Q = BF16 ? 1 : 2;
            for (long long i0 = tid; i0 < n; i0 += (long long)U3 * kIpcThreads) {
                u32x4 p[U3][Q];
#pragma unroll
                for (int u = 0; u < U3; ++u) {
                    const long long i = i0 + (long long)u * kIpcThreads;
                    if (i < n) {
#pragma unroll
                        for (int q = 0; q < Q; ++q) p[u][q] = load16_sys(src + i * GB + 16 * q);
                    }
                }
#pragma unroll
                for (int u = 0; u < U3; ++u) {
                    const long long i = i0 + (long long)u * kIpcThreads;
                    if (i >= n) continue;
                    f32x4 lo, hi;
                    if (BF16) unpack8(p[u][0], lo, hi);
                    else { lo = __builtin_bit_cast(f32x4, p[u][0]); hi = __builtin_bit_cast(f32x4, p[u][Q - 1]); }
                    *reinterpret_cast<f32x4*>(dst + i * 8) = lo; *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = hi;
                }
            }
        }
    if (tid == 0) *seqp = seq;
}

__global__ void __launch_bounds__(256) k_wire_pack(const float* g, u32x4* wire, long long granules) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < granules; i += (long long)gridDim.x * 256)
        wire[i] = pack8(*reinterpret_cast<const f32x4*>(g + i * 8), *reinterpret_cast<const f32x4*>(g + i * 8 + 4));
}
__global__ void __launch_bounds__(256) k_wire_unpack(float* g, const u32x4* wire, long long granules) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < granules; i += (long long)gridDim.x * 256) {
        f32x4 lo, hi;
        unpack8(wire[i], lo, hi);
        *reinterpret_cast<f32x4*>(g + i * 8) = lo; *reinterpret_cast<f32x4*>(g + i * 8 + 4) = hi;
    }
}

}  // namespace

void launch_ipc_allreduce(const IpcArgs& a, hipStream_t s) {
    if (a.wire_bf16) hipLaunchKernelGGL(k_ipc_allreduce<true>, dim3(a.blocks), dim3(kIpcThreads), 0, s, a);
    else hipLaunchKernelGGL(k_ipc_allreduce<false>, dim3(a.blocks), dim3(kIpcThreads), 0, s, a);
}

void launch_wire_pack(const float* g, void* wire, long long n, hipStream_t s) {
    const long long gr = n / 8;
    if (gr <= 0) return;
    hipLaunchKernelGGL(k_wire_pack, dim3((unsigned)((gr + 255) / 256 < 1024 ? (gr + 255) / 256 : 1024)), dim3(256), 0, s, g, reinterpret_cast<u32x4*>(wire), gr);
}
void launch_wire_unpack(float* g, const void* wire, long long n, hipStream_t s) {
    const long long gr = n / 8;
    if (gr <= 0) return;
    hipLaunchKernelGGL(k_wire_unpack, dim3((unsigned)((gr + 255) / 256 < 1024 ? (gr + 255) / 256 : 1024)), dim3(256), 0, s, g, reinterpret_cast<const u32x4*>(wire), gr);
}

}  // namespace avae
