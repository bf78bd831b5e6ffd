// Data-parallel gradient exchange: RCCL all-reduce over xGMI on the ctx stream.
// librccl is loaded lazily (dlopen) so that single-GPU runs never touch it.  The reference has no
// collective code at all; one process per GPU, gradients summed per net bucket.
#include <dlfcn.h>
#include <string.h>
#include <unistd.h>

#include "common.h"

namespace {

// minimal RCCL surface (nccl.h ABI): ncclUniqueId is 128 opaque bytes
typedef struct { char internal[128]; } rccl_uid;
typedef void* rccl_comm;
enum { RCCL_FLOAT32 = 7, RCCL_BFLOAT16 = 9 };
enum { RCCL_SUM = 0, RCCL_MAX = 2 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(rccl_uid*) = nullptr;
    int (*CommInitRank)(rccl_comm*, int, rccl_uid, int) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*CommCount)(rccl_comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

RcclApi g_api;

// RCCL prints a start-up banner (ROCm version / hostname / library path) on STDOUT while it initialises.  The
// launcher contract reserves rank 0's stdout for one JSON line, so fd 1 points at stderr for the duration of the
// calls that can trigger it.
struct StdoutToStderr {
    int saved;
    StdoutToStderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) dup2(2, 1);
    }
    ~StdoutToStderr() {
        fflush(stdout);
        if (saved >= 0) {
            dup2(saved, 1);
            close(saved);
        }
    }
};

int load_rccl() {
    if (g_api.handle) return 0;
    // the ROCm install this library was linked against first: a bare soname would resolve to whatever copy a host
    // process has already mapped (e.g. the librccl bundled with a torch wheel, built against another HIP runtime)
    const char* names[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_api.handle) break;
    }
    GHM_CHECK(g_api.handle != nullptr, "cannot dlopen librccl: %s", dlerror());
#define RCCL_SYM(field, name)                                          \
    *(void**)(&g_api.field) = dlsym(g_api.handle, name);               \
    GHM_CHECK(g_api.field != nullptr, "librccl lacks symbol %s", name)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    RCCL_SYM(CommInitRank, "ncclCommInitRank");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(CommCount, "ncclCommCount");
    RCCL_SYM(AllReduce, "ncclAllReduce");
    RCCL_SYM(ReduceScatter, "ncclReduceScatter");
    RCCL_SYM(AllGather, "ncclAllGather");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    return 0;
}

#define GHM_RCCL(expr)                                                                        \
    do {                                                                                      \
        int _r = (expr);                                                                      \
        if (_r != 0) {                                                                        \
            ghm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_api.GetErrorString(_r)); \
            return -4;                                                                        \
        }                                                                                     \
    } while (0)

// fp32 <-> bf16 (round to nearest even) over a contiguous range: the reduced-precision exchange buffer
__global__ __launch_bounds__(256) void comm_f32_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned u = __float_as_uint(x[i]);
    unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    if ((u & 0x7f800000u) == 0x7f800000u) r = u;          // inf / nan keep their class (a nan stays a nan: the quiet bit is in the top half)
    y[i] = (unsigned short)(r >> 16);
}

__global__ __launch_bounds__(256) void comm_bf16_to_f32_kernel(const unsigned short* __restrict__ y, float* __restrict__ x, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = __uint_as_float((unsigned)y[i] << 16);
}

}  // namespace

extern "C" {

int ghm_comm_unique_id(uint8_t id_out[128]) {
    StdoutToStderr quiet;
    if (int e = load_rccl()) return e;
    rccl_uid id;
    GHM_RCCL(g_api.GetUniqueId(&id));
    memcpy(id_out, id.internal, 128);
    return 0;
}

int ghm_comm_init(ghm_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[128]) {
    StdoutToStderr quiet;
    if (int e = load_rccl()) return e;
    GHM_CHECK(ctx->comm == nullptr, "communicator already initialised");
    GHM_HIP(hipSetDevice(ctx->device));
    rccl_uid uid;
    memcpy(uid.internal, id, 128);
    rccl_comm comm = nullptr;
    GHM_RCCL(g_api.CommInitRank(&comm, world, uid, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    // RCCL sets its channels up (and prints) on the first collective: do that one here, inside the collective
    // initialisation call and with stdout parked, so that no later all-reduce blocks the host or carries state
    float* warm = nullptr;
    GHM_HIP(hipMalloc((void**)&warm, 256));
    GHM_HIP(hipMemsetAsync(warm, 0, 256, ctx->stream));
    GHM_RCCL(g_api.AllReduce(warm, warm, 1, RCCL_FLOAT32, RCCL_SUM, comm, ctx->stream));
    GHM_HIP(hipStreamSynchronize(ctx->stream));
    GHM_HIP(hipFree(warm));
    return 0;
}

int ghm_comm_count(ghm_ctx* ctx, int32_t* nranks) {
    GHM_CHECK(ctx->comm != nullptr, "ghm_comm_count on a context without a communicator");
    int n = 0;
    GHM_RCCL(g_api.CommCount((rccl_comm)ctx->comm, &n));      // what RCCL itself says, not what the caller passed in
    *nranks = n;
    return 0;
}

int ghm_comm_destroy(ghm_ctx* ctx) {
    if (ctx->comm && g_api.handle) {
        GHM_RCCL(g_api.CommDestroy((rccl_comm)ctx->comm));
    }
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
    return 0;
}

int ghm_allreduce_sum(ghm_ctx* ctx, float* buf, int64_t n) {
    // no identity shortcut: a caller that reduces on a context without a communicator is reducing on the wrong
    // context (its gradients would silently stay local while being scaled by 1/world)
    GHM_CHECK(ctx->comm != nullptr, "ghm_allreduce_sum on a context without a communicator (ghm_comm_init)");
    if (ctx->rec) {         // part of a recorded step: the collective is issued at every replay
        ghm_step* st = ctx->rec;
        rccl_comm comm = (rccl_comm)ctx->comm;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            if (g_api.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_SUM, comm, s) != 0 && st->err == hipSuccess)
                st->err = hipErrorUnknown;
        });
        return 0;
    }
    GHM_RCCL(g_api.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_SUM, (rccl_comm)ctx->comm, ctx->stream));
    return 0;
}

// The same sum with HALF the bytes on the links (opt-in, GanStep(exchange_mode='allreduce_bf16')): the fp32 range is rounded to
// bf16 into ``scratch`` (n halfwords), summed by RCCL in bf16 and widened back into ``buf`` -- a REDUCED-PRECISION exchange (8
// significant bits per contribution, rounded again per reduction hop), for configurations where the 226 MB of fp32 gradients
// per step are exposed on the xGMI links (BASELINE config 4: bf16 compute at 6 ms per step).  Recordable like the fp32 form.
int ghm_allreduce_sum_bf16(ghm_ctx* ctx, float* buf, int64_t n, void* scratch) {
    GHM_CHECK(ctx->comm != nullptr, "ghm_allreduce_sum_bf16 on a context without a communicator (ghm_comm_init)");
    GHM_CHECK(buf && scratch && n > 0, "ghm_allreduce_sum_bf16: null argument");
    unsigned short* const h = (unsigned short*)scratch;
    hipLaunchKernelGGL(comm_f32_to_bf16_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, (const float*)buf, h, (long)n);
    GHM_LAUNCH_CHECK();
    rccl_comm comm = (rccl_comm)ctx->comm;
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            if (g_api.AllReduce(h, h, (size_t)n, RCCL_BFLOAT16, RCCL_SUM, comm, s) != 0 && st->err == hipSuccess) st->err = hipErrorUnknown;
        });
    } else {
        GHM_RCCL(g_api.AllReduce(h, h, (size_t)n, RCCL_BFLOAT16, RCCL_SUM, comm, ctx->stream));
    }
    hipLaunchKernelGGL(comm_bf16_to_f32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, (const unsigned short*)h, buf, (long)n);
    GHM_LAUNCH_CHECK();
    return 0;
}

// Sharded update (SURVEY 8e gives the all-reduce form; this is the reduce-scatter / all-gather form of the same exchange):
// buf holds world x shard elements; after ghm_reduce_scatter_sum rank r's shard [r * shard, (r + 1) * shard) holds the sum
// over the ranks (the other shards are unspecified), after ghm_all_gather every rank holds every rank's shard.  Both in
// place (RCCL's in-place forms: recv = send + rank * shard / send = recv + rank * shard), both recordable in a step.
int ghm_reduce_scatter_sum(ghm_ctx* ctx, float* buf, int64_t shard) {
    GHM_CHECK(ctx->comm != nullptr, "ghm_reduce_scatter_sum on a context without a communicator (ghm_comm_init)");
    rccl_comm comm = (rccl_comm)ctx->comm;
    float* mine = buf + (size_t)ctx->rank * shard;
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            if (g_api.ReduceScatter(buf, mine, (size_t)shard, RCCL_FLOAT32, RCCL_SUM, comm, s) != 0 && st->err == hipSuccess)
                st->err = hipErrorUnknown;
        });
        return 0;
    }
    GHM_RCCL(g_api.ReduceScatter(buf, mine, (size_t)shard, RCCL_FLOAT32, RCCL_SUM, comm, ctx->stream));
    return 0;
}

int ghm_all_gather(ghm_ctx* ctx, float* buf, int64_t shard) {
    GHM_CHECK(ctx->comm != nullptr, "ghm_all_gather on a context without a communicator (ghm_comm_init)");
    rccl_comm comm = (rccl_comm)ctx->comm;
    const float* mine = buf + (size_t)ctx->rank * shard;
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            if (g_api.AllGather(mine, buf, (size_t)shard, RCCL_FLOAT32, comm, s) != 0 && st->err == hipSuccess)
                st->err = hipErrorUnknown;
        });
        return 0;
    }
    GHM_RCCL(g_api.AllGather(mine, buf, (size_t)shard, RCCL_FLOAT32, comm, ctx->stream));
    return 0;
}

int ghm_allreduce_max(ghm_ctx* ctx, float* buf, int64_t n) {
    GHM_CHECK(ctx->comm != nullptr, "ghm_allreduce_max on a context without a communicator (ghm_comm_init)");
    if (ctx->rec) {         // recordable like the sum (a fp16 overflow flag agreed on by every rank of the sharded update)
        ghm_step* st = ctx->rec;
        rccl_comm comm = (rccl_comm)ctx->comm;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            if (g_api.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_MAX, comm, s) != 0 && st->err == hipSuccess)
                st->err = hipErrorUnknown;
        });
        return 0;
    }
    GHM_RCCL(g_api.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, RCCL_MAX, (rccl_comm)ctx->comm, ctx->stream));
    return 0;
}

}  // extern "C"
