// The data-parallel gradient exchange WITHOUT a collective library (SURVEY 8b's second cut; replaces dist.all_reduce(SUM) of
// cleanrl/ppo_atari_multigpu.py:360-367 on the persistent flat gradient buffer): every rank of the node owns ONE device segment that its
// peers map through HIP IPC (over xGMI between GPUs), and the all-reduce is five small launches on the caller's stream -- no host
// round trip, no second stream, legal inside a hipGraph capture, so an update slot of world > 1 is ONE graph like world = 1's.
//
//   segment of rank r  = | ready[q], q < world | got[q], q < world | data (cap floats) | out (cap floats) |      (flags 128 bytes apart)
//
//   A  publish  (rank r)   data_r := g (a local copy at HBM rate); the launch's last workgroup then PUSHES ready_q[r] := k to every rank q
//                          (k = the round's number, counted on the device: nothing of a call depends on host state, a graph replays it).
//   w  wait     (rank r)   ONE wave: waits for ready_r[q] >= k, q < world (a spin on LOCAL memory).
//   B  reduce   (rank r)   adds slice r of data_0 .. data_{w-1} IN RANK ORDER (w - 1 remote reads of n / w floats) and PUSHES the sums into
//                          slice r of out_q for every q (posted writes); the last workgroup pushes got_q[r] := k.
//   w  wait     (rank q)   ONE wave: waits for got_q[s] >= k, s < world (local).
//   C  collect  (rank q)   g := out_q.
// The waits are launches of their own so that exactly one wave per rank ever spins: a wide kernel whose every workgroup waits holds a wave
// (registers; with a barrier variable, LDS) on every CU it lands on, and kernels that need a CU's whole register file or LDS -- kernels R / U / G
// / H of a peer PROCESS ON THE SAME DEVICE, the one-GPU test runs -- then cannot be placed anywhere while the peer they belong to is exactly who
// the waiters wait for (seen: 3 x 128 waiting workgroups against a fourth rank's backward pass, a deadlock until the timeout).
//
// Per rank 2 (w - 1) / w x n floats cross the links, spread over all w - 1 of them at once (xGMI is point to point: a ring would use one
// link per direction); every element is summed by exactly one rank in one fixed order, so all ranks hold the same bits (what
// tests/test_gpu_multirank.py asserts of the replicas) and the result does not depend on timing.  One buffer of each kind suffices: rank r
// overwrites data_r in round k + 1 only after its own C of round k, which waited for got_r[s] = k of every s, i.e. every reader of data_r
// in round k is done; rank s overwrites out_q in round k + 1 only after ready_s[q] = k + 1, which q pushed after its C of round k.
// Memory model: the segments are fine-grained device memory, flags are read and written with system-scope atomics, payload stores are
// released by a system-scope fence before the flag that announces them and acquired by one behind the flag's observation (the AMDGPU
// memory model's recipe for memory shared between agents).  A wait gives up after `timeout_ms` (a peer died, a rank skipped a call): it
// records the round and the peer in a host-visible word and every later wait of the communicator returns at once -- a broken exchange
// costs one timeout and surfaces in mi355ppo_dp_comm_status, it never hangs the device.
// NOT run across GPUs by any build round (one GPU per box): tests/test_gpu_multirank.py runs 2 and 4 PROCESSES on one device, where the
// peers' segments are IPC mappings of the same HBM -- the protocol, the IPC plumbing and the arithmetic, not the fabric.
#include "common.h"
#include <string.h>

namespace mi355ppo {

constexpr int kDpMaxWorld = MI355PPO_DP_MAX_WORLD;
constexpr int kDpFlagWords = 32;                                   // one flag per 128-byte line
constexpr size_t kDpHeader = 2 * kDpMaxWorld * kDpFlagWords * 4;    // ready[8] | got[8]
constexpr int kDpThreads = 256;

typedef float dp_f32x4 __attribute__((ext_vector_type(4)));

struct DpView {                     // what a launch needs, by value
    unsigned char* seg[kDpMaxWorld];      // every rank's segment as mapped into THIS process (seg[rank]: the local allocation)
    unsigned* local;                // private words: [0] round, [16] ticket of A, [32] ticket of B
    unsigned* err;                  // host-visible: [0] 0 = fine, else the round a wait gave up in; [1] the peer waited for; [2] 1 = in B, 2 = in C
    unsigned long long ticks;       // timeout in wall-clock ticks
    long long cap;                  // floats per buffer
    int world, rank;
};

__device__ __forceinline__ unsigned* dp_ready(unsigned char* seg, int q) { return reinterpret_cast<unsigned*>(seg) + q * kDpFlagWords; }
__device__ __forceinline__ unsigned* dp_got(unsigned char* seg, int q) { return reinterpret_cast<unsigned*>(seg) + (kDpMaxWorld + q) * kDpFlagWords; }
__device__ __forceinline__ float* dp_data(unsigned char* seg) { return reinterpret_cast<float*>(seg + kDpHeader); }
__device__ __forceinline__ float* dp_out(unsigned char* seg, long long cap) { return reinterpret_cast<float*>(seg + kDpHeader) + cap; }

// flag >= k (rounds wrap after 2^32 calls: compared as a signed difference), or give up
__device__ bool dp_wait(const unsigned* flag, unsigned k, const DpView& v, int peer, unsigned where) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spin = 1;; ++spin) {
        const unsigned f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(f - k) >= 0) return true;
        __builtin_amdgcn_s_sleep(8);
        if ((spin & 63u) != 0u) continue;                          // the clock and the error word (host memory) every 64th look
        if (__hip_atomic_load(v.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;      // the communicator is already broken
        if (wall_clock64() - t0 > v.ticks) {
            unsigned expect = 0u;
            if (__hip_atomic_compare_exchange_strong(v.err, &expect, k ? k : 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                __hip_atomic_store(v.err + 1, (unsigned)peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(v.err + 2, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return false;
        }
    }
}

// The last workgroup of a launch to arrive (every other one has fenced its stores before taking its ticket); the answer is wave 0's (uniform in
// it, false in the other waves; no LDS).
__device__ __forceinline__ bool dp_last_block(unsigned* ticket) {
    __threadfence_system();
    __syncthreads();
    unsigned last = 0u;
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1 ? 1u : 0u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return threadIdx.x < 64 && __builtin_amdgcn_readfirstlane((int)last) != 0;
}

// ---- A: data := g, then ready_q[rank] := k on every rank q
__global__ __launch_bounds__(kDpThreads) void dp_publish_kernel(DpView v, const float* __restrict__ g, long long n) {
    const unsigned k = __hip_atomic_load(v.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    float* const data = dp_data(v.seg[v.rank]);
    const long long nv = n >> 2, stride = (long long)gridDim.x * kDpThreads;
    for (long long i = (long long)blockIdx.x * kDpThreads + threadIdx.x; i < nv; i += stride)
        reinterpret_cast<dp_f32x4*>(data)[i] = reinterpret_cast<const dp_f32x4*>(g)[i];
    if (blockIdx.x == 0 && threadIdx.x < 4) {                       // the last, partial vector: zero-filled (B adds whole vectors)
        const long long i = 4 * nv + threadIdx.x;
        if ((n & 3) != 0) data[i] = i < n ? g[i] : 0.0f;
    }
    if (!dp_last_block(v.local + 16)) return;
    if (threadIdx.x == 0) __hip_atomic_store(v.local, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // B and C of this round read k here
    if ((int)threadIdx.x < v.world)
        __hip_atomic_store(dp_ready(v.seg[threadIdx.x], v.rank), k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- w: one wave; lane q waits for flag q of this rank's segment (GOT = 0: ready, 1: got) to reach the round
template <int GOT>
__global__ __launch_bounds__(64) void dp_wait_kernel(DpView v) {
    const unsigned k = __hip_atomic_load(v.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < v.world)
        dp_wait(GOT ? dp_got(v.seg[v.rank], threadIdx.x) : dp_ready(v.seg[v.rank], threadIdx.x), k, v, threadIdx.x, GOT ? 2u : 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// ---- B: slice `rank` of every rank's data, summed in rank order, into slice `rank` of every rank's out; then got_q[rank] := k
__global__ __launch_bounds__(kDpThreads) void dp_reduce_kernel(DpView v, long long n) {
    const unsigned k = __hip_atomic_load(v.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                   // (behind the wait launch's observation of the flags)
    const long long nv = (n + 3) >> 2, per = (nv + v.world - 1) / v.world;
    const long long lo = per * v.rank, hi = lo + per < nv ? lo + per : nv;
    const long long stride = (long long)gridDim.x * kDpThreads;
    for (long long i = lo + (long long)blockIdx.x * kDpThreads + threadIdx.x; i < hi; i += stride) {
        dp_f32x4 x[kDpMaxWorld];
#pragma unroll
        for (int q = 0; q < kDpMaxWorld; ++q)
            if (q < v.world) x[q] = reinterpret_cast<const dp_f32x4*>(dp_data(v.seg[q]))[i];
        dp_f32x4 s = x[0];
#pragma unroll
        for (int q = 1; q < kDpMaxWorld; ++q)
            if (q < v.world) s += x[q];                             // ((x0 + x1) + x2) + ...: one order, on one rank
#pragma unroll
        for (int q = 0; q < kDpMaxWorld; ++q)
            if (q < v.world) reinterpret_cast<dp_f32x4*>(dp_out(v.seg[q], v.cap))[i] = s;
    }
    if (!dp_last_block(v.local + 32)) return;
    if ((int)threadIdx.x < v.world)
        __hip_atomic_store(dp_got(v.seg[threadIdx.x], v.rank), k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- C: g := out
__global__ __launch_bounds__(kDpThreads) void dp_collect_kernel(DpView v, float* __restrict__ g, long long n) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const float* const out = dp_out(v.seg[v.rank], v.cap);
    const long long nv = n >> 2, stride = (long long)gridDim.x * kDpThreads;
    for (long long i = (long long)blockIdx.x * kDpThreads + threadIdx.x; i < nv; i += stride)
        reinterpret_cast<dp_f32x4*>(g)[i] = reinterpret_cast<const dp_f32x4*>(out)[i];
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const long long i = 4 * nv + threadIdx.x;
        if (i < n) g[i] = out[i];
    }
}

}  // namespace mi355ppo

using namespace mi355ppo;

struct mi355ppo_dp_comm {
    DpView v;
    int device;
    bool opened[kDpMaxWorld];       // seg[q] is an IPC mapping this communicator opened
    bool connected;
    unsigned* err_host;             // the host address of v.err
    size_t seg_bytes;
};

namespace {
int dp_fail(const char* what, hipError_t e) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MI355PPO_EHIP;
}
void dp_release(mi355ppo_dp_comm* c) {
    for (int q = 0; q < kDpMaxWorld; ++q)
        if (c->opened[q] && c->v.seg[q]) (void)hipIpcCloseMemHandle(c->v.seg[q]);
    if (c->v.seg[c->v.rank]) (void)hipFree(c->v.seg[c->v.rank]);
    if (c->v.local) (void)hipFree(c->v.local);
    if (c->err_host) (void)hipHostFree(c->err_host);
    delete c;
}
}  // namespace

extern "C" {

int mi355ppo_dp_comm_create(int world, int rank, int64_t max_floats, double timeout_ms, mi355ppo_dp_comm** out) {
    if (!out || world < 1 || world > kDpMaxWorld || rank < 0 || rank >= world || max_floats < 1 || !(timeout_ms > 0.0)) {
        set_error("mi355ppo_dp_comm_create: world in 1..%d, rank in 0..world-1, max_floats >= 1, timeout_ms > 0", kDpMaxWorld);
        return MI355PPO_EINVAL;
    }
    *out = nullptr;
    mi355ppo_dp_comm* c = new mi355ppo_dp_comm();
    c->v.world = world;
    c->v.rank = rank;
    c->v.cap = (max_floats + 1023) / 1024 * 1024;
    c->seg_bytes = kDpHeader + 2 * (size_t)c->v.cap * sizeof(float);
    hipError_t e = hipGetDevice(&c->device);
    int khz = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device);
    if (e != hipSuccess || khz <= 0) {
        dp_release(c);
        return e != hipSuccess ? dp_fail("mi355ppo_dp_comm_create: device query", e) : (set_error("mi355ppo_dp_comm_create: no wall-clock rate"), MI355PPO_EHIP);
    }
    c->v.ticks = (unsigned long long)(timeout_ms * (double)khz);
    void* seg = nullptr;
    // fine-grained: coherent between agents while kernels run (flags AND payload); a coarse-grained segment is only defined at kernel boundaries
    e = hipExtMallocWithFlags(&seg, c->seg_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { dp_release(c); return dp_fail("mi355ppo_dp_comm_create: hipExtMallocWithFlags(fine-grained)", e); }
    c->v.seg[rank] = static_cast<unsigned char*>(seg);
    if ((e = hipMemset(seg, 0, kDpHeader)) != hipSuccess) { dp_release(c); return dp_fail("mi355ppo_dp_comm_create: hipMemset", e); }
    void* local = nullptr;
    if ((e = hipMalloc(&local, 256)) != hipSuccess || (e = hipMemset(local, 0, 256)) != hipSuccess) {
        c->v.local = static_cast<unsigned*>(local);
        dp_release(c);
        return dp_fail("mi355ppo_dp_comm_create: hipMalloc", e);
    }
    c->v.local = static_cast<unsigned*>(local);
    void* eh = nullptr;
    if ((e = hipHostMalloc(&eh, 64, hipHostMallocMapped)) != hipSuccess) { dp_release(c); return dp_fail("mi355ppo_dp_comm_create: hipHostMalloc", e); }
    c->err_host = static_cast<unsigned*>(eh);
    c->err_host[0] = c->err_host[1] = c->err_host[2] = 0u;
    void* ed = nullptr;
    if ((e = hipHostGetDevicePointer(&ed, eh, 0)) != hipSuccess) { dp_release(c); return dp_fail("mi355ppo_dp_comm_create: hipHostGetDevicePointer", e); }
    c->v.err = static_cast<unsigned*>(ed);
    if ((e = hipDeviceSynchronize()) != hipSuccess) { dp_release(c); return dp_fail("mi355ppo_dp_comm_create: hipDeviceSynchronize", e); }
    c->connected = world == 1;
    *out = c;
    return MI355PPO_OK;
}

int mi355ppo_dp_comm_handle(mi355ppo_dp_comm* c, unsigned char* handle) {
    static_assert(sizeof(hipIpcMemHandle_t) == MI355PPO_DP_HANDLE_BYTES, "the header's handle size");
    if (!c || !handle) { set_error("mi355ppo_dp_comm_handle: null argument"); return MI355PPO_EINVAL; }
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, c->v.seg[c->v.rank]);
    if (e != hipSuccess) return dp_fail("mi355ppo_dp_comm_handle: hipIpcGetMemHandle", e);
    memcpy(handle, &h, sizeof(h));
    return MI355PPO_OK;
}

int mi355ppo_dp_comm_connect(mi355ppo_dp_comm* c, const unsigned char* handles) {
    if (!c || !handles) { set_error("mi355ppo_dp_comm_connect: null argument"); return MI355PPO_EINVAL; }
    if (c->connected) { set_error("mi355ppo_dp_comm_connect: already connected"); return MI355PPO_EINVAL; }
    for (int q = 0; q < c->v.world; ++q) {
        if (q == c->v.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)q * MI355PPO_DP_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("mi355ppo_dp_comm_connect: hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e));
            return MI355PPO_EHIP;
        }
        c->v.seg[q] = static_cast<unsigned char*>(p);
        c->opened[q] = true;
    }
    c->connected = true;
    return MI355PPO_OK;
}

int mi355ppo_dp_allreduce_sum_f32(mi355ppo_dp_comm* c, float* grads, int64_t n, void* stream) {
    if (!c || !grads || n < 0) { set_error("mi355ppo_dp_allreduce_sum_f32: null argument or n < 0"); return MI355PPO_EINVAL; }
    if (!c->connected) { set_error("mi355ppo_dp_allreduce_sum_f32: mi355ppo_dp_comm_connect has not run"); return MI355PPO_EINVAL; }
    if (n > c->v.cap) { set_error("mi355ppo_dp_allreduce_sum_f32: n = %lld above the communicator's capacity %lld", (long long)n, c->v.cap); return MI355PPO_EINVAL; }
    if (!aligned(grads, 16)) { set_error("mi355ppo_dp_allreduce_sum_f32: grads must be 16-byte aligned"); return MI355PPO_EALIGN; }
    if (n == 0 || c->v.world == 1) return MI355PPO_OK;
    hipStream_t s = as_stream(stream);
    const long long nv = (n + 3) / 4, per = (nv + c->v.world - 1) / c->v.world;
    static_assert(kDpMaxWorld <= 64, "the flags of a launch are pushed by one wave");
    auto blocks = [](long long vecs, int cap) { const long long b = (vecs + kDpThreads - 1) / kDpThreads; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); };
    hipLaunchKernelGGL(dp_publish_kernel, dim3(blocks(nv, 128)), dim3(kDpThreads), 0, s, c->v, grads, (long long)n);
    hipLaunchKernelGGL(dp_wait_kernel<0>, dim3(1), dim3(64), 0, s, c->v);
    hipLaunchKernelGGL(dp_reduce_kernel, dim3(blocks(per, 128)), dim3(kDpThreads), 0, s, c->v, (long long)n);
    hipLaunchKernelGGL(dp_wait_kernel<1>, dim3(1), dim3(64), 0, s, c->v);
    hipLaunchKernelGGL(dp_collect_kernel, dim3(blocks(nv, 128)), dim3(kDpThreads), 0, s, c->v, grads, (long long)n);
    return check_launch("mi355ppo_dp_allreduce_sum_f32");
}

int mi355ppo_dp_comm_status(mi355ppo_dp_comm* c, int* round, int* peer, int* phase) {
    if (!c) { set_error("mi355ppo_dp_comm_status: null communicator"); return MI355PPO_EINVAL; }
    const volatile unsigned* e = c->err_host;
    if (round) *round = (int)e[0];
    if (peer) *peer = (int)e[1];
    if (phase) *phase = (int)e[2];
    return e[0] == 0u ? MI355PPO_OK : MI355PPO_ETIMEOUT;
}

int mi355ppo_dp_comm_destroy(mi355ppo_dp_comm* c) {
    if (!c) return MI355PPO_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    dp_release(c);
    return MI355PPO_OK;
}

}  // extern "C"
