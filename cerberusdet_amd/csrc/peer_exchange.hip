// SyncBatchNorm statistics exchange by peer writes (gfx950, one node): the small all-reduce between a BatchNorm layer's statistics kernel and its
// normalisation kernel (reference train.py:140-143 turns every BatchNorm into SyncBatchNorm: 2 x 97 tiny collectives per task pass).
//
// As RCCL calls these are ~350 host-enqueued collectives per iteration, each a kernel launch + a ring/tree protocol for 0.3 - 15 KB, on the dependent
// chain of every layer. Here every rank owns ONE exchange buffer that all its peers map through HIP IPC (hipIpcGetMemHandle / OpenMemHandle, exchanged
// once at start-up). One single-workgroup kernel per exchange:
//   1. writes the rank's vector into its row of the slot in EVERY rank's buffer (its own included) -- peer stores over xGMI --, fences at system
//      scope and publishes a per-(slot, rank) flag = the exchange's epoch in every buffer;
//   2. spins (bounded) until all `world` flags of the slot in its OWN buffer carry the epoch, then sums the `world` rows in rank order -- the same
//      order on every rank, so all ranks hold bit-identical sums (RCCL's ring order is not specified; for two ranks a + b is the same either way).
// Slots are double-buffered by epoch parity: a rank can only reach epoch e + 2 of a slot after every peer has finished reading epoch e (it needs their
// e + 1 flags first). A wait that exceeds its (wall-clock) budget poisons the vector with NaN and raises a sticky error word instead of hanging the GPU
// or handing on stale rows.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace cdet {

// phase: 0 = the whole exchange; 1 = write + publish only, 2 = wait + sum only (the two halves as separate launches with a host-side barrier
// between them: for ranks that SHARE one GPU, whose kernels the driver time-slices instead of running them side by side -- a spinning kernel
// would wait for a peer that cannot run; tests/test_gpu_distributed.py). spin: clock budget of the wait (a missing peer is an error, not a hang).
__global__ __launch_bounds__(256) void peer_allreduce_kernel(float* __restrict__ vec, int n, const uint64_t* __restrict__ peers, int world, int rank,
                                                             long long data_off, long long flag_off, unsigned epoch, unsigned* __restrict__ err,
                                                             int phase, unsigned long long spin) {
    const int t = threadIdx.x;
    const int par = (int)(epoch & 1u);
    const long long row = ((long long)par * world + rank) * n;
    // 1. my vector into my row of the slot on every rank
    if (phase != 2) {
    for (int p = 0; p < world; ++p) {
        float* dst = reinterpret_cast<float*>(peers[p]) + data_off + row;
        for (int i = t; i < n; i += 256) __hip_atomic_store(dst + i, vec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (t < world) {
        unsigned* f = reinterpret_cast<unsigned*>(peers[t]) + flag_off + par * world + rank;
        __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    }
    if (phase == 1) return;
    // 2. all rows of the slot in MY buffer
    __shared__ int timed_out;
    if (t == 0) timed_out = 0;
    __syncthreads();
    if (t < world) {
        const unsigned* f = reinterpret_cast<const unsigned*>(peers[rank]) + flag_off + par * world + t;
        const unsigned long long t0 = wall_clock64();  // constant-rate counter (s_memrealtime), not the shader clock: the budget is wall time
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
            if (wall_clock64() - t0 > spin) {
                // sticky: the first time-out of the run stays in the error word (the host raises wherever it looks next, peer_exchange.py::check)
                unsigned expect = 0u;
                __hip_atomic_compare_exchange_strong(err, &expect, 1u + (unsigned)t, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
    __threadfence_system();
    if (timed_out) {
        // never hand stale rows (epoch e - 2) on as statistics: poison the vector so that the loss of this very iteration shows it
        for (int i = t; i < n; i += 256) vec[i] = __builtin_nanf("");
        return;
    }
    const float* src = reinterpret_cast<const float*>(peers[rank]) + data_off + (long long)par * world * n;
    for (int i = t; i < n; i += 256) {
        float s = 0.f;
        for (int r = 0; r < world; ++r) s += __hip_atomic_load(src + (long long)r * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        vec[i] = s;
    }
}

}  // namespace cdet

using namespace cdet;

extern "C" int cdet_peer_alloc(int64_t bytes, void** out) {
    CDET_CHECK_ARG(out && bytes > 0, "cdet_peer_alloc: bad arguments");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, (size_t)bytes);
    }
    if (e != hipSuccess) {
        set_error("cdet_peer_alloc: %s", hipGetErrorString(e));
        return -(int)e;
    }
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("cdet_peer_alloc: memset: %s", hipGetErrorString(e));
        return -(int)e;
    }
    *out = p;
    return 0;
}

extern "C" int cdet_peer_free(void* p) {
    if (p) (void)hipFree(p);
    return 0;
}

extern "C" int cdet_peer_export(void* p, void* handle64) {
    CDET_CHECK_ARG(p && handle64, "cdet_peer_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handles are 64 bytes");
    hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), p);
    if (e != hipSuccess) {
        set_error("cdet_peer_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

extern "C" int cdet_peer_import(const void* handle64, void** out) {
    CDET_CHECK_ARG(handle64 && out, "cdet_peer_import: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        set_error("cdet_peer_import: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
        return -(int)e;
    }
    *out = p;
    return 0;
}

extern "C" int cdet_peer_close(void* p) {
    if (p) (void)hipIpcCloseMemHandle(p);
    return 0;
}

extern "C" int cdet_peer_allreduce(float* vec, int32_t n, const void* peer_table, int32_t world, int32_t rank, int64_t data_off, int64_t flag_off,
                                   uint32_t epoch, void* err, int32_t phase, void* stream) {
    CDET_CHECK_ARG(vec && peer_table && err && n > 0, "cdet_peer_allreduce: bad arguments");
    CDET_CHECK_ARG(world >= 1 && world <= 64 && rank >= 0 && rank < world && epoch != 0, "cdet_peer_allreduce: bad world / rank / epoch");
    CDET_CHECK_ARG(phase >= 0 && phase <= 2, "cdet_peer_allreduce: phase must be 0 (whole exchange), 1 (publish) or 2 (collect)");
    // wait budget per exchange in wall time: the `peer_spin_ms` switch (default 10 min, the order of a collective library's watchdog: ordinary rank
    // skew -- a host-bound loader, a plan compile on one rank, checkpoint IO -- must never trip it; peer_exchange.py sets a few seconds around its
    // start-up self-test, which exists to catch a mapping that opens but does not carry peer stores). The wall-clock rate is the CURRENT device's.
    static int khz_of[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int khz = (dev >= 0 && dev < 64) ? khz_of[dev] : 0;
    if (khz <= 0) {
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;  // gfx9: 100 MHz
        (void)hipGetLastError();
        if (dev >= 0 && dev < 64) khz_of[dev] = khz;
    }
    const int ms = sw(SW_PEER_SPIN_MS) > 0 ? sw(SW_PEER_SPIN_MS) : 600000;
    const unsigned long long spin = (unsigned long long)ms * (unsigned long long)khz;
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, vec, n, (const uint64_t*)peer_table, world, rank,
                       (long long)data_off, (long long)flag_off, epoch, (unsigned*)err, phase, spin);
    CDET_LAUNCH_CHECK();
    return 0;
}
