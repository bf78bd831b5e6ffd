// Internal helpers shared by the libghm.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/ghm.h"

#define GHM_MAX_TIMERS 4096

struct ghm_ctx;

struct ghm_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ghm_ctx* owner = nullptr;       // the context whose workspace pointers the captured launches carry (pinned while the graph lives)
};

struct ghm_ctx;

// A train step as ONE call (ghm_step_run).  Two forms:
//   graphs:   the captured HIP graph of each stage stream (ghm_step_build);
//   recorded: the host's own launch sequence -- every kernel launch, stream wait, memset, timer and collective issued
//             on the attached contexts between ghm_step_record_begin / _end is appended to ``cmds`` instead of being
//             executed, and replayed in that order on the same streams.  This is the eager multi-stream schedule
//             without the per-launch host work of the caller (HIP graphs replay forked streams slowly on this stack).
struct ghm_step {
    int n = 0;
    ghm_ctx* ctx[8] = {};
    ghm_graph* graph[8] = {};
    bool recorded = false, recording = false;
    std::vector<std::function<void()>> cmds;
    std::vector<hipEvent_t> events;   // the cross-stream ordering events of the recorded waits: created once, re-recorded per replay
    long runs = 0;              // completed replays
    int timer_stride = 0;       // recorded timer slots advance by this much per replay (0: reuse the slots)
    hipError_t err = hipSuccess;
};

struct ghm_ctx {
    ghm_step* rec = nullptr;       // non-null while a recorded step is being built on this context
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start[GHM_MAX_TIMERS] = {};
    hipEvent_t ev_stop[GHM_MAX_TIMERS] = {};
    void* comm = nullptr;          // ncclComm_t (RCCL), owned by comm.hip
    int rank = 0, world = 1;
    int num_cu = 256;
    bool capturing = false;
    void* scratch = nullptr;       // library-owned workspace (split-K partials, reduction partials)
    size_t scratch_bytes = 0;
    // recorded steps / captured graphs hold the workspace pointer BY VALUE: while any exists on this context
    // (``pinned`` > 0) an outgrown workspace block is retired (kept allocated until the context dies), never freed
    int pinned = 0;
    std::vector<void*> retired;
    size_t retired_bytes = 0;
    std::vector<ghm_graph*> graphs;   // live captured graphs of this context (their ``owner`` is cleared if the context dies first)
    float* ls_state = nullptr;     // dynamic loss scale {scale, 1/scale, clean steps, overflow flag, skipped steps, ...} or null
    float* zeros = nullptr;        // 256 B of zeros: the source of padding elements for LDS-DMA row staging
    int* tickets = nullptr;        // zeroed arrival counters of the folded split-K reductions (self-resetting)
};

// Tuning / ablation switches (GHM_* environment variables) are read ONCE per call site and cached: nothing on the launch
// path calls getenv() per launch.  ghm_options_reload() (include/ghm.h) makes every site re-read its variable (tests and
// sweep tools that flip a switch inside one process).
extern int g_ghm_opt_epoch;
#define GHM_OPT(name)                                            \
    ([]() -> const char* {                                       \
        static int ep_ = -1;                                     \
        static const char* v_ = nullptr;                         \
        if (ep_ != g_ghm_opt_epoch) {                            \
            v_ = getenv(name);                                   \
            ep_ = g_ghm_opt_epoch;                               \
        }                                                        \
        return v_;                                               \
    }())

// CU count the context-free predicates / workspace queries plan with (ghm_lp_supported, ghm_conv2d_pool_supported,
// ghm_dgrad_dact_supported, ghm_conv2d_wgrad_workspace, ghm_conv2d_variant): the CU count of the device the process's
// contexts live on (256 before the first ghm_ctx_create), so that a predicate and the launch it vouches for -- which
// plans with ctx->num_cu -- always agree
int ghm_plan_cus();

// grow-only workspace owned by the ctx; growing is illegal while a graph is being captured
int ghm_scratch(ghm_ctx* ctx, size_t bytes, void** out);
void ghm_unpin(ghm_ctx* ctx);     // a graph / recorded step that pinned the context's workspace is gone

void ghm_set_error(const char* fmt, ...);

// Split-K reductions folded into their producer kernel: every block writes its partial slice, then takes a ticket at
// its output tile's counter; the block that draws the last ticket sums all slices in FIXED order (bit-repeatable whoever
// arrives last) and finishes the tile.  ghm_tickets: the context's counter array (zero between launches: the last block
// resets its counter) or null when the launch has more tiles than counters or the fold is not enabled (GHM_SPLITK_FOLD:
// see ghm_tickets in ctx.hip for why it is opt-in).
#define GHM_MAX_TICKETS (1 << 18)
int* ghm_tickets(ghm_ctx* ctx, long tiles);
#if defined(__HIPCC__)
__device__ __forceinline__ bool ghm_last_arrival(int* ticket, int nsplit) {
    __shared__ int s_last_;
    __threadfence();                    // release: this block's slice is visible device-wide before its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = atomicAdd(ticket, 1);
        s_last_ = (t == nsplit - 1);
        if (s_last_) *ticket = 0;       // nobody else touches this counter in this launch any more
    }
    __syncthreads();
    const bool last = s_last_ != 0;
    if (last) __threadfence();          // acquire: the other blocks' slices
    return last;
}
#endif

#define GHM_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            ghm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

#define GHM_CHECK(cond, ...)                 \
    do {                                     \
        if (!(cond)) {                       \
            ghm_set_error(__VA_ARGS__);      \
            return -2;                       \
        }                                    \
    } while (0)

#define GHM_LAUNCH_CHECK() GHM_HIP(hipGetLastError())

// Every kernel launch of the library goes through this: executed now, or appended to the step being recorded on the
// context (the arguments are captured by value).  ``ctx`` is the ghm_ctx* in scope at every launch site.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, ...) GHM_LAUNCH_IMPL((kernel), __VA_ARGS__)
// TUNING ONLY: GHM_SKIP_KERNELS=<substring>[,<substring>...] drops every launch whose kernel expression contains one of
// the substrings (results are wrong; tells what a kernel family costs inside the overlapped schedule)
bool ghm_skip_kernel(const char* name);
#define GHM_LAUNCH_IMPL(kernel, grid, block, lds, stream, ...)                                       \
    do {                                                                                               \
        if (ghm_skip_kernel(#kernel)) break;                                                           \
        if (ctx->rec) {                                                                                \
            const dim3 g_ = (grid), b_ = (block);                                                      \
            const size_t l_ = (size_t)(lds);                                                           \
            hipStream_t s_ = (stream);                                                                 \
            ctx->rec->cmds.emplace_back([=]() { kernel<<<g_, b_, l_, s_>>>(__VA_ARGS__); });          \
        } else {                                                                                       \
            kernel<<<(grid), (block), (lds), (stream)>>>(__VA_ARGS__);                                 \
        }                                                                                              \
    } while (0)

// the same for the other stream operations (copies, memsets, waits, timers, collectives)
template <typename F>
static inline void ghm_submit(ghm_ctx* ctx, F f) {
    if (ctx->rec)
        ctx->rec->cmds.emplace_back(f);
    else
        f();
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// activation and its derivative expressed through the OUTPUT (valid for all five kinds)
// XCD-aware block remap: consecutive logical tiles (which share halo pixels / operand tiles) land on the same XCD's
// L2 -- workgroups are dispatched round-robin over the 8 XCDs.  Bijective for any grid size.
__device__ __forceinline__ int ghm_xcd_remap(int bid, int nb) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float ghm_act(float v, int act, float alpha) {
    switch (act) {
        case GHM_ACT_RELU: return v > 0.f ? v : 0.f;
        case GHM_ACT_LRELU: return v > 0.f ? v : alpha * v;
        case GHM_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case GHM_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// bit 4 of a pooling mask byte: the pooled activation is > 0 (all the backward pass needs of it for relu / leaky relu,
// so the pooled fp32 tensor itself need not be read -- or written, when every consumer reads its q copy)
#define GHM_POOL_SIGN 16u
__device__ __forceinline__ float ghm_dact_from_sign(unsigned mask_byte, int act, float alpha) {
    return (act == GHM_ACT_LINEAR || (mask_byte & GHM_POOL_SIGN)) ? 1.f : (act == GHM_ACT_RELU ? 0.f : alpha);
}

__device__ __forceinline__ float ghm_dact_from_out(float y, int act, float alpha) {
    switch (act) {
        case GHM_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case GHM_ACT_LRELU: return y > 0.f ? 1.f : alpha;
        case GHM_ACT_SIGMOID: return y * (1.f - y);
        case GHM_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

// elementwise.hip: the reduction passes of ghm_bn_backward (dgamma / dbeta + the sums in the workspace tail)
extern "C" int ghm_bn_backward_sums(ghm_ctx* ctx, const float* dout, int64_t ds, const float* y, int64_t ys, const float* x,
                                    int64_t xs, int32_t N, int32_t C, int32_t HW, const float* mean, const float* inv,
                                    float* dgamma, float* dbeta, int32_t act, float alpha, int32_t accumulate, void* ws,
                                    const float* gamma, const float* beta);

extern "C" int ghm_bn_backward_finish(ghm_ctx* ctx, const double* wsd, int32_t C, int32_t S, float* sums, float* dgamma,
                                      float* dbeta, int32_t accumulate);

// ---- conv_thin.hip: layers with <= 4 channels on one side and large maps (HBM-bound) ----
bool thin_fanout_fwd_ok(const ghm_conv_desc* d, int act);   // the kernel's epilogue does linear / relu / lrelu
int thin_fanout_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                    float* y, int act, float alpha, int accumulate, void* yq = nullptr, long yq_nstride = 0, int q_dt = 0);
bool thin_fanout_fwd_pool_ok(const ghm_conv_desc* d, int act);    // + activation + 2x2 max-pool in the epilogue
int thin_fanout_fwd_pool(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                         float* pooled, unsigned char* mask, int act, float alpha, void* yq = nullptr, long yq_nstride = 0,
                         int q_dt = 0);
bool thin_fanout_pool_q_ok(const ghm_conv_desc* d);      // may the pooled forward also write its q copy?
bool thin_fanout_dgrad_ok(const ghm_conv_desc* d, int act);
int thin_fanout_dgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                      float* dx, int act, float alpha, int accumulate);
bool thin_fanin_s2_ok(const ghm_conv_desc* d, const float* dx);
int thin_fanin_s2(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                  float* dx, int act, float alpha, int accumulate);
bool thin_wgrad_ok(const ghm_conv_desc* d, const float* x, const float* dy);
int thin_wgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* dy, float* dwp, int accumulate);
// conv_lp.hip: conv + activation + 2x2 max-pool on the low-precision matrix cores
bool lp_conv_pool_supported(const ghm_conv_desc* d, int act, int dtype);
int lp_conv_fwd_pool(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const void* wq, const float* bias, float* pooled,
                     unsigned char* mask, int act, float alpha, int dtype);
bool lp_dgrad_s2_single_pass(const ghm_conv_desc* d, int dtype);
int lp_dgrad_s2_dact(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const void* wqT, float* dx, const float* dact_y,
                     long dact_nstride, int dact, float dact_alpha, int dtype);
// conv_small.hip: layers whose maps are at most 16 x 16 pixels (the U-Net's inner half, the first DCGAN generator stages,
// the last discriminator stages) on the bf16 / fp16 matrix cores as TWO launches per layer: a gather GEMM over (row tile,
// pixel group, split) and a finishing kernel per 8 channels that also runs the BatchNorm behind the convolution.
// conv_lp.hip routes the forward products (kind 0) and data gradients (kind 1) served there.
struct SmBn {           // BatchNorm behind the convolution, folded into the finishing kernel (training statistics)
    const float* gamma;
    const float* beta;
    float* mean;
    float* inv;
    float* run_mean;    // or null
    float* run_inv;
    float eps, run_alpha;
    float* y;           // fp32 act(bn(conv)) or null (the q copy goes to sm_conv's outq)
    long y_nstride;
};
bool sm_use(const ghm_conv_desc* d, int kind, int dtype);
int sm_finish_launch(ghm_ctx* ctx, const float* partial, int splits, int R, int N, int HW, const float* bias, float* out32,
                     long out_nstride, int accumulate, int act, float alpha, void* outq, long outq_ns, int dtype, const SmBn* bn);
// conv_lp.hip: a forward convolution that lp_conv_kernel runs in split-K form (16 x 16 maps) can end in the same finishing
// kernel, BatchNorm included
bool lp_fwd_splitk_bn_ok(const ghm_conv_desc* d, int dtype);
int lp_fwd_splitk_bn(ghm_ctx* ctx, const ghm_conv_desc* d, const void* xq, long xq_ns, const void* wq, const float* bias,
                     float* conv_out, void* yq, long yq_ns, int act, float alpha, int dtype, const SmBn* bn);
int sm_conv(ghm_ctx* ctx, const ghm_conv_desc* d, int kind, const void* inq, long inq_ns, const float* in32, const void* wq,
            const float* bias, float* out32, long out_nstride, void* outq, long outq_ns, int act, float alpha, int accumulate,
            int dtype, const SmBn* bn);
// conv_thin_lp.hip: the fused first-layer forward (<= 4 channels -> 64 filters, activation, 2x2 max-pool) on the bf16 / fp16
// matrix cores for the reduced-precision modes
bool thin_pool_lp_ok(const ghm_conv_desc* d, int act, float alpha, int dtype);
int thin_pool_lp(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias, float* pooled,
                 unsigned char* mask, int act, float alpha, void* yq, long yq_nstride, int dtype);
// split-K epilogue of a forward-form convolution: out = act(sum of S partial slices [S][R][N*H*W] + bias (+ out))
int ghm_splitk_finish(ghm_ctx* ctx, const float* partial, int S, float* out, const float* bias, int N, int R, int H,
                      int W, long out_nstride, int act, float alpha, int accumulate);
// out[i] (+)= sum over S slices of part[s*split_stride + i], fixed order (conv_igemm.hip)
int ghm_reduce_splits(ghm_ctx* ctx, const float* part, int S, long n, long split_stride, float* out, int accumulate);
bool thin_fanin_s1_fwd_ok(const ghm_conv_desc* d);
int thin_fanin_s1_fwd(ghm_ctx* ctx, const ghm_conv_desc* d, const float* x, const float* wp, const float* bias,
                      float* y, int act, float alpha, int accumulate);
bool thin_fanin_s1_dgrad_ok(const ghm_conv_desc* d);
int thin_fanin_s1_dgrad(ghm_ctx* ctx, const ghm_conv_desc* d, const float* dy, const float* wp, const float* bias,
                        float* dx, int act, float alpha, int accumulate);

