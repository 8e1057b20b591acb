// ks_xchg.hip -- LDS-free gradient all-reduce over peer-mapped memory (C ABI: include/kinova_rollout.h, kr_xchg_*).
//
// The one exchange step of the path is the learner's gradient mean over the ranks (DDPGfD on env shards): two
// flat fp32 buffers of ~0.35 MB per update.  RCCL's all-reduce kernels need LDS, and the stepping kernel (k_env_step) holds
// all of every CU's LDS for the whole env-step, so an RCCL collective issued beside it only starts once stepping workgroups
// retire: the learner's chain (critic backward -> all-reduce -> critic step + actor backward -> all-reduce) then runs
// BEHIND the simulator instead of beside it.  At these sizes a ring is latency, not bandwidth: every rank can simply read
// its peers' buffers.  One process per GPU; each rank owns an exchange block [flags | data] in uncached device memory,
// exported with hipIpcGetMemHandle and mapped by every peer (xGMI peer access on a node); one kernel per all-reduce:
//
//   wait   until every peer has finished READING my data block of the previous call (flag set B, deferred from last call)
//   copy   my gradient slice -> my data block
//   signal every peer's flag set A[block][me] = epoch; wait for A[block][r] == epoch from every peer r
//   reduce out[i] = (sum over ranks r = 0 .. world-1, in that order, of data_r[i]) / world   -> bitwise identical on all ranks
//   signal every peer's flag set B[block][me] = epoch
//
// Workgroups are independent (a flag row per workgroup), 256 threads, no LDS, ~32 registers: they run on the registers and
// issue slots the stepping kernel leaves free, like the learner's other launches.  Spins are bounded (wall clock, 60 s unless
// KS_XCHG_TIMEOUT_S says otherwise; <= 0 waits for ever like a library collective): a peer that never arrives makes the call set a
// sticky error word instead of hanging the GPU - from then on this rank's gradients are NOT reduced, so the host must look at
// kr_xchg_status: PeerExchange.check() does, and the trainer calls it every few hundred updates and at every flush.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/kinova_rollout.h"
#include "../../include/kinova_sim.h"

namespace {

constexpr int XB = 32;            // workgroups per all-reduce (each owns 1/32 of the buffer and one row of every flag set)
constexpr int XT = 256;           // threads per workgroup
constexpr int XR = KR_XCHG_MAX_RANKS;
constexpr long long TICKS_PER_S = 100000000ll;   // wall_clock64: 100 MHz
constexpr double DEFAULT_TIMEOUT_S = 60.0;       // KS_XCHG_TIMEOUT_S overrides; <= 0: wait for ever (what a library collective does)

struct Flags {
    uint32_t a[XB][XR];           // arrival of epoch e: peer r's data block holds its gradient slice
    uint32_t b[XB][XR];           // departure of epoch e: peer r has finished reading my data block
    uint32_t error;               // set (and never cleared) by my own kernel when a spin ran out: the epoch of the failure
    uint32_t pad[63];
};

struct Peers {
    Flags* flags[XR];
    const float* data[XR];
};

__device__ __forceinline__ bool spin_until(const uint32_t* p, uint32_t want, long long spin_ticks) {
    const long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        if (spin_ticks > 0 && wall_clock64() - t0 > spin_ticks) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

__global__ __launch_bounds__(XT) void k_xchg_allreduce(Peers peers, int world, int me, uint32_t epoch, float* __restrict__ grad, long count,
                                                       float* __restrict__ mine, long per, long long spin_ticks) {
    Flags* my = peers.flags[me];
    const int blk = blockIdx.x, tid = threadIdx.x;
    // Slice of this workgroup, in float4 units.  The partition is FIXED at create time (per = ceil(capacity / 4 / XB)), not derived
    // from this call's count: a workgroup's departure flags only say that the peers' workgroup `blk` has finished reading region
    // `blk` of my block, so region `blk` must be the same words in every call - with a count-dependent split (the learner
    // alternates 88321 / 88068 floats) workgroup blk of call e+1 overwrote up to 31 float4 that a peer's workgroup blk-1 of call e
    // could still be reading.  The last count % 4 elements belong to the workgroup whose region holds float4 index n4.
    const long n4 = count >> 2, lo = blk * per < n4 ? blk * per : n4, hi = lo + per < n4 ? lo + per : n4;
    const long tail = (blk == (int)(n4 / per) && tid < (count & 3)) ? 4 * n4 + tid : -1;
    // (no __syncthreads_and / shared flags: they would cost LDS.  The waiting lanes sit in wave 0; a spin that runs out sets the
    // block's sticky error word, which everybody reads after the barrier)
    // peers have read what the previous call left in my data block
    if (epoch > 1 && tid < world && tid != me && !spin_until(&my->b[blk][tid], epoch - 1, spin_ticks))
        __hip_atomic_store(&my->error, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    float4* dst = (float4*)mine;
    const float4* src = (const float4*)grad;
    for (long i = lo + tid; i < hi; i += XT) dst[i] = src[i];
    if (tail >= 0) mine[tail] = grad[tail];
    __threadfence_system();
    __syncthreads();
    if (tid < world && tid != me) {
        __hip_atomic_store(&peers.flags[tid]->a[blk][me], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!spin_until(&my->a[blk][tid], epoch, spin_ticks)) __hip_atomic_store(&my->error, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (__hip_atomic_load(&my->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        const float inv = 1.0f / (float)world;
        float4* out = (float4*)grad;
        for (long i = lo + tid; i < hi; i += XT) {
            float4 s = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < world; r++) {
                const float4 v = ((const float4*)peers.data[r])[i];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
            out[i] = s;
        }
        if (tail >= 0) {
            float t = 0.f;
            for (int r = 0; r < world; r++) t += peers.data[r][tail];
            grad[tail] = t * inv;
        }
    }
    __syncthreads();
    if (tid < world && tid != me) __hip_atomic_store(&peers.flags[tid]->b[blk][me], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct kr_xchg {
    int world = 0, rank = 0;
    long capacity = 0;                 // floats
    long per = 0;                      // float4 per workgroup region: ceil(capacity / 4 / XB), fixed for the life of the block
    long long spin_ticks = 0;          // bound of every wait in wall-clock ticks; 0 = unbounded
    void* block = nullptr;             // [Flags | data]
    void* mapped[XR] = {};
    Peers peers{};
    uint32_t epoch = 0;
    bool connected = false;
};

extern "C" {

int kr_xchg_create(kr_xchg** out, int32_t world, int32_t rank, int64_t max_count, uint8_t* handle_out) {
    if (!out || !handle_out || world < 2 || world > XR || rank < 0 || rank >= world || max_count <= 0) return KS_ERR_INVALID;
    kr_xchg* x = new kr_xchg;
    x->world = world; x->rank = rank;
    x->capacity = (max_count + 3) / 4 * 4;
    x->per = (x->capacity / 4 + XB - 1) / XB;
    double timeout_s = DEFAULT_TIMEOUT_S;
    if (const char* e = getenv("KS_XCHG_TIMEOUT_S")) timeout_s = atof(e);
    x->spin_ticks = timeout_s > 0 ? (long long)(timeout_s * (double)TICKS_PER_S) : 0;
    const size_t bytes = sizeof(Flags) + (size_t)x->capacity * sizeof(float);
    // uncached (fine-grained) device memory: peers' stores to the flags and loads of the data must not meet a stale L2 line
    if (hipExtMallocWithFlags(&x->block, bytes, hipDeviceMallocUncached) != hipSuccess) { delete x; return KS_ERR_HIP; }
    if (hipMemset(x->block, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(x->block); delete x; return KS_ERR_HIP; }
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == KR_XCHG_HANDLE_BYTES, "IPC handle size");
    if (hipIpcGetMemHandle(&h, x->block) != hipSuccess) { (void)hipFree(x->block); delete x; return KS_ERR_HIP; }
    std::memcpy(handle_out, &h, sizeof h);
    *out = x;
    return KS_OK;
}

int kr_xchg_connect(kr_xchg* x, const uint8_t* handles) {
    if (!x || !handles || x->connected) return KS_ERR_INVALID;
    for (int r = 0; r < x->world; r++) {
        void* p = x->block;
        if (r != x->rank) {
            hipIpcMemHandle_t h;
            std::memcpy(&h, handles + (size_t)r * KR_XCHG_HANDLE_BYTES, sizeof h);
            if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return KS_ERR_HIP;
            x->mapped[r] = p;
        }
        x->peers.flags[r] = (Flags*)p;
        x->peers.data[r] = (const float*)((const char*)p + sizeof(Flags));
    }
    x->connected = true;
    return KS_OK;
}

int kr_xchg_allreduce_mean(kr_xchg* x, float* grad, int64_t count, void* stream) {
    if (!x || !x->connected || !grad || count <= 0 || count > x->capacity || ((uintptr_t)grad & 15) != 0) return KS_ERR_INVALID;
    x->epoch++;
    hipLaunchKernelGGL(k_xchg_allreduce, dim3(XB), dim3(XT), 0, (hipStream_t)stream, x->peers, x->world, x->rank, x->epoch, grad, (long)count,
                       (float*)((char*)x->block + sizeof(Flags)), x->per, x->spin_ticks);
    return hipGetLastError() == hipSuccess ? KS_OK : KS_ERR_HIP;
}

int kr_xchg_status(kr_xchg* x, uint32_t* failed_epoch) {
    if (!x || !failed_epoch) return KS_ERR_INVALID;
    uint32_t e = 0;
    if (hipMemcpy(&e, &((Flags*)x->block)->error, sizeof e, hipMemcpyDeviceToHost) != hipSuccess) return KS_ERR_HIP;
    *failed_epoch = e;
    return KS_OK;
}

void kr_xchg_destroy(kr_xchg* x) {
    if (!x) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < x->world; r++)
        if (x->mapped[r]) (void)hipIpcCloseMemHandle(x->mapped[r]);
    if (x->block) (void)hipFree(x->block);
    delete x;
}

}  // extern "C"
