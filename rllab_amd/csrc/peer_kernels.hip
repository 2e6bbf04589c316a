// peer_kernels.hip -- one-shot sum all-reduce of a small float64 vector across the GPUs of one node, inside the
// launch train of the update ("sum all-reduce (RCCL over xGMI, or one-shot P2P write +
// fixed-order local sum for determinism)").
//
// What it replaces: the sharded TRPO update needs the sum over ranks of the flat gradient and of each of the
// cg_iters Fisher-vector products (rllab/optimizers/conjugate_gradient_optimizer.py:229-262 evaluates f_grad once
// and f_Hx_plain cg_iters + 1 times per update; the reference itself is single-process).  Through torch.distributed
// each of those is a host-issued RCCL call on CG's critical path (~28 us apiece even with one rank); all messages are
// <= 8 P bytes (12.6 KB for the headline net), i.e. pure latency.  Here every rank WRITES its row straight into every
// peer's mailbox (peer-mapped hipIpc memory over xGMI), raises a per-(slot, source) flag, waits for the world's
// flags in its own mailbox and sums the world's rows IN RANK ORDER: bit-identical results on every rank, no host
// call between the product and the CG algebra, one small launch.
//
// Mailbox of one rank (device memory of that rank, exported with hipIpcGetMemHandle, opened by every peer):
//   flags  uint64 [2][MAX_WORLD]          flags[slot][src] = sequence number of the row src last wrote into slot
//   rows   double [2][world][max_n]       rows[slot][src][i]
// Reduction number seq (1, 2, 3, ...; identical on all ranks, they issue the same launch train) uses slot seq & 1.
// Two slots suffice: a rank can start writing reduction seq + 2 into a peer only after that peer raised its flag for
// seq + 1, which it does after it finished reading seq.
// Every spin is bounded BY WALL CLOCK (s_memrealtime, 100 MHz: a straggler's GC pause or first-launch module load is
// milliseconds, the limit is seconds); a missing peer sets *err, the launch completes and its output is POISONED with
// NaN -- the consumer (CG / the line search) then rejects the step instead of applying an update computed from stale
// rows; the host raises at its next check of the error word.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/rllab_amd.h"
#include "capi_util.h"

namespace rl {

constexpr int PEER_MAX_WORLD = 8;
constexpr int PEER_THREADS = 1024;
constexpr size_t PEER_HEADER = 2 * PEER_MAX_WORLD * sizeof(unsigned long long);    // the flag words

struct PeerArgs {
    int n, rank, world, max_n;
    unsigned long long seq;
    double* data;
    char* box[PEER_MAX_WORLD];     // box[rank] = own mailbox, box[p] = peer p's (IPC mapping)
    int* err;
    long long spin_limit;          // iterations (tests of the give-up path)
    long long tick_limit;          // 100 MHz wall-clock ticks
};

__device__ __forceinline__ unsigned long long* flag_of(char* box, int slot, int src) {
    return reinterpret_cast<unsigned long long*>(box) + slot * PEER_MAX_WORLD + src;
}
__device__ __forceinline__ double* row_of(char* box, int slot, int src, int world, int max_n) {
    return reinterpret_cast<double*>(box + PEER_HEADER) + ((size_t)slot * world + src) * max_n;
}

__global__ void __launch_bounds__(PEER_THREADS) peer_allreduce_kernel(PeerArgs a) {
    const int slot = (int)(a.seq & 1ull);
    // 1. my row into every mailbox of the world (mine included), write-through to the owner's memory
    for (int p = 0; p < a.world; ++p) {
        double* dst = row_of(a.box[p], slot, a.rank, a.world, a.max_n);
        for (int i = threadIdx.x; i < a.n; i += PEER_THREADS)
            __hip_atomic_store(dst + i, a.data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();                       // every lane's stores are out before the flags go up
    __syncthreads();
    if (threadIdx.x < a.world)
        __hip_atomic_store(flag_of(a.box[threadIdx.x], slot, a.rank), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait for the world's rows in MY mailbox (one lane per source rank, bounded)
    __shared__ int gave_up;
    if (threadIdx.x == 0) gave_up = 0;
    __syncthreads();
    if (threadIdx.x < a.world) {
        unsigned long long* f = flag_of(a.box[a.rank], slot, threadIdx.x);
        long long spins = 0;
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > a.spin_limit || (long long)wall_clock64() - t0 > a.tick_limit) {
                atomicExch(a.err, 1 + (int)threadIdx.x);
                atomicExch(&gave_up, 1);
                break;
            }
        }
    }
    __syncthreads();
    if (gave_up) {
        // a row of the world never arrived: whatever sits in the mailbox is stale.  NaN out the result so that nothing
        // downstream (CG's dot products, the line search's acceptance test) can mistake it for a sum.
        for (int i = threadIdx.x; i < a.n; i += PEER_THREADS) a.data[i] = __builtin_nan("");
        return;
    }
    __threadfence_system();                       // acquire: the rows behind the flags
    // 3. fixed-order sum: the same bits on every rank
    for (int i = threadIdx.x; i < a.n; i += PEER_THREADS) {
        double s = 0.0;
        for (int p = 0; p < a.world; ++p)
            s += __hip_atomic_load(row_of(a.box[a.rank], slot, p, a.world, a.max_n) + i, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        a.data[i] = s;
    }
}

}  // namespace rl

using namespace rl;

extern "C" size_t rl_peer_mailbox_bytes(int world, int max_n) {
    if (world < 1 || world > PEER_MAX_WORLD || max_n < 1) return 0;
    return PEER_HEADER + (size_t)2 * world * max_n * sizeof(double);
}

extern "C" int rl_peer_alloc(size_t bytes, void** dev_ptr_out) {
    if (!dev_ptr_out || bytes == 0) return set_error(RL_ERR_ARG, "rl_peer_alloc: bad argument");
    void* p = nullptr;
    // fine-grained device memory: visible to peers inside a running kernel (what a flag protocol needs); plain
    // hipMalloc is coarse-grained, coherent across devices only at kernel boundaries
    // (no fallback to hipMalloc: a coarse-grained mailbox would pass every single-GPU test and hand peers stale rows)
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return set_error(RL_ERR_HIP, "rl_peer_alloc: fine-grained device memory unavailable (%s)", hipGetErrorString(e));
    }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return set_error(RL_ERR_HIP, "rl_peer_alloc: %s", hipGetErrorString(e)); }
    *dev_ptr_out = p;
    return 0;
}

extern "C" int rl_peer_free(void* dev_ptr) {
    if (!dev_ptr) return 0;
    hipError_t e = hipFree(dev_ptr);
    return e == hipSuccess ? 0 : set_error(RL_ERR_HIP, "rl_peer_free: %s", hipGetErrorString(e));
}

extern "C" int rl_peer_export(void* dev_ptr, void* handle_out64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands IPC handles around as 64 opaque bytes");
    if (!dev_ptr || !handle_out64) return set_error(RL_ERR_ARG, "rl_peer_export: bad argument");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, dev_ptr);
    if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle_out64, &h, sizeof(h));
    return 0;
}

extern "C" int rl_peer_open(const void* handle64, void** dev_ptr_out) {
    if (!handle64 || !dev_ptr_out) return set_error(RL_ERR_ARG, "rl_peer_open: bad argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return set_error(RL_ERR_HIP, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *dev_ptr_out = p;
    return 0;
}

extern "C" int rl_peer_close(void* dev_ptr) {
    if (!dev_ptr) return 0;
    hipError_t e = hipIpcCloseMemHandle(dev_ptr);
    return e == hipSuccess ? 0 : set_error(RL_ERR_HIP, "hipIpcCloseMemHandle: %s", hipGetErrorString(e));
}

extern "C" int rl_peer_allreduce_sum(int n, double* data, int rank, int world, void* const* mailboxes, int max_n,
                                     uint64_t seq, int* err_dev, int64_t spin_limit, void* stream) {
    if (n <= 0 || !data || !mailboxes || !err_dev || world < 1 || world > PEER_MAX_WORLD || rank < 0 || rank >= world ||
        n > max_n || seq == 0)
        return set_error(RL_ERR_ARG, "rl_peer_allreduce_sum: bad argument (n = %d, max_n = %d, rank %d of %d)", n, max_n,
                         rank, world);
    PeerArgs a;
    a.n = n; a.rank = rank; a.world = world; a.max_n = max_n; a.seq = seq; a.data = data; a.err = err_dev;
    for (int p = 0; p < PEER_MAX_WORLD; ++p) a.box[p] = p < world ? (char*)mailboxes[p] : nullptr;
    for (int p = 0; p < world; ++p)
        if (!a.box[p]) return set_error(RL_ERR_ARG, "rl_peer_allreduce_sum: mailbox %d is null", p);
    // a peer that has not delivered after 10 s of wall clock is an error.  spin_limit > 0 (iterations of s_sleep 8 + one
    // load; the binding passes RLLAB_PEER_SPIN_LIMIT): tests of the give-up path make it milliseconds.
    a.tick_limit = 10ll * 100000000ll;
    a.spin_limit = spin_limit > 0 ? (long long)spin_limit : (1ll << 62);
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(PEER_THREADS), 0, (hipStream_t)stream, a);
    return check_launch("peer_allreduce_kernel");
}
