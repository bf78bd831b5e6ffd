// Context, device memory, stream, HIP-graph capture and event timers of libghm.so.
#include <stdarg.h>
#include <string.h>

#include <chrono>

#include "common.h"

static thread_local char g_err[1024] = "";

void ghm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_ghm_opt_epoch = 0;
bool ghm_skip_kernel(const char* name) {
    const char* pats = GHM_OPT("GHM_SKIP_KERNELS");
    if (!pats) return false;
    char buf[512];
    strncpy(buf, pats, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = 0;
    for (char* tok = strtok(buf, ","); tok; tok = strtok(nullptr, ","))
        if (*tok && strstr(name, tok)) return true;
    return false;
}

int* ghm_tickets(ghm_ctx* ctx, long tiles) {
    // OFF unless GHM_SPLITK_FOLD is set.  Measured (round 3, joint fp32 step): 152.9 img/s folded against 169.0 with the
    // separate reduction launches -- the release / acquire pair around the ticket is an L2 write-back + invalidate per
    // block on this multi-XCD part (the eight L2s are not coherent with each other), and with three other streams resident
    // it evicts THEIR working sets too.  The ~75 small reduction launches per step cost 0.6 ms (GHM_SKIP_KERNELS ablation:
    // 23.61 -> 23.01 ms); the fold costs 2.5 ms.  Kept for single-stream use and as the record of why it is not the default.
    if (tiles > GHM_MAX_TICKETS || !ctx->tickets || !GHM_OPT("GHM_SPLITK_FOLD")) return nullptr;
    return ctx->tickets;
}

// flags of the events that order one COMPUTE stream of the GPU behind another: every consumer is a kernel on the SAME device,
// so a device-scope release is enough (hipEventReleaseToDevice); HIP's default is a system-scope release (visible to the host
// and to peers), i.e. a heavier cache write-back at every one of the ~100 cross-stream dependencies of a step.  ``system``:
// edges whose other side is NOT a kernel of this device keep the default -- the communication stream (RCCL reads and writes
// these buffers from peer GPUs over xGMI) and the persistent events the host synchronises on / that order host-to-device
// copies.  GHM_EVENT_SYSTEM_SCOPE=1 restores the default everywhere for A/B.
static unsigned ghm_event_flags(bool system) {
    static int sys_ = -1;
    if (sys_ < 0) sys_ = getenv("GHM_EVENT_SYSTEM_SCOPE") ? 1 : 0;
    return hipEventDisableTiming | ((sys_ || system) ? 0u : (unsigned)hipEventReleaseToDevice);
}

static int g_plan_cus = 256;
int ghm_plan_cus() { return g_plan_cus; }

int ghm_scratch(ghm_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->scratch_bytes) {
        GHM_CHECK(!ctx->capturing, "workspace would grow (%zu -> %zu bytes) inside graph capture: run the program "
                  "eagerly once before capturing", ctx->scratch_bytes, bytes);
        GHM_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) {
            if (ctx->pinned > 0) {          // a recorded step or graph replays launches that carry this pointer
                ctx->retired.push_back(ctx->scratch);
                ctx->retired_bytes += ctx->scratch_bytes;
            } else {
                GHM_HIP(hipFree(ctx->scratch));
            }
        }
        const size_t want = bytes + bytes / 4;
        GHM_HIP(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return 0;
}

// the last graph / recorded step that replays this context's workspace pointers is gone: the outgrown blocks that were
// kept alive for it can be freed
void ghm_unpin(ghm_ctx* ctx) {
    if (ctx->pinned > 0) --ctx->pinned;
    if (ctx->pinned == 0 && !ctx->retired.empty()) {
        (void)hipStreamSynchronize(ctx->stream);
        for (void* p : ctx->retired) (void)hipFree(p);
        ctx->retired.clear();
        ctx->retired_bytes = 0;
    }
}

extern "C" {

const char* ghm_last_error(void) { return g_err; }

int ghm_options_reload(void) {
    ++g_ghm_opt_epoch;
    return 0;
}

int ghm_device_count(int32_t* n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        c = 0;
        (void)hipGetLastError();
    }
    *n = c;
    return 0;
}

static int ctx_create_impl(int32_t device, int prio_arg, bool prio_given, ghm_ctx** out);

int ghm_ctx_create(int32_t device, ghm_ctx** out) { return ctx_create_impl(device, 0, false, out); }

// a context whose stream has the given HIP priority (negative = more urgent, clamped to the device's range)
int ghm_ctx_create_prio(int32_t device, int32_t priority, ghm_ctx** out) { return ctx_create_impl(device, priority, true, out); }

}  // extern "C"

static int ctx_create_impl(int32_t device, int prio_arg, bool prio_given, ghm_ctx** out) {
    GHM_HIP(hipSetDevice(device));
    ghm_ctx* c = new ghm_ctx();
    c->device = device;
    // tuning: GHM_STREAM_PRIO = comma-separated HIP stream priorities by order of context creation (lower = more urgent)
    int prio = prio_arg;
    static int created = 0;
    if (const char* f = getenv("GHM_STREAM_PRIO")) {
        const char* p = f;
        for (int i = 0; i < created && p; ++i) { p = strchr(p, ','); if (p) ++p; }
        if (p && !prio_given) prio = atoi(p);
    }
    ++created;
    if (prio != 0) {
        int lo = 0, hi = 0;
        GHM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (prio < hi) prio = hi;
        if (prio > lo) prio = lo;
        GHM_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio));
    } else {
        GHM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    hipDeviceProp_t prop;
    GHM_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    g_plan_cus = c->num_cu;
    GHM_HIP(hipMalloc((void**)&c->zeros, 256));
    GHM_HIP(hipMemset(c->zeros, 0, 256));
    GHM_HIP(hipMalloc((void**)&c->tickets, GHM_MAX_TICKETS * sizeof(int)));
    GHM_HIP(hipMemset(c->tickets, 0, GHM_MAX_TICKETS * sizeof(int)));
    *out = c;
    return 0;
}

extern "C" {

int ghm_ctx_destroy(ghm_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm) ghm_comm_destroy(ctx);
    for (ghm_graph* g : ctx->graphs) g->owner = nullptr;       // graphs may outlive their context: they just stop pinning it
    for (int i = 0; i < GHM_MAX_TIMERS; ++i) {
        if (ctx->ev_start[i]) (void)hipEventDestroy(ctx->ev_start[i]);
        if (ctx->ev_stop[i]) (void)hipEventDestroy(ctx->ev_stop[i]);
    }
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (void* p : ctx->retired) (void)hipFree(p);
    if (ctx->zeros) (void)hipFree(ctx->zeros);
    if (ctx->tickets) (void)hipFree(ctx->tickets);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int ghm_device_info(ghm_ctx* ctx, char* name, int32_t name_len, int32_t* num_cu, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    GHM_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}

int ghm_scratch_info(ghm_ctx* ctx, int64_t* bytes, int64_t* retired_bytes, int32_t* pinned) {
    if (bytes) *bytes = (int64_t)ctx->scratch_bytes;
    if (retired_bytes) *retired_bytes = (int64_t)ctx->retired_bytes;
    if (pinned) *pinned = ctx->pinned;
    return 0;
}

int ghm_alloc(ghm_ctx* ctx, size_t bytes, void** out) {
    GHM_CHECK(!ctx->capturing, "ghm_alloc inside graph capture");
    GHM_HIP(hipSetDevice(ctx->device));
    if (bytes == 0) bytes = 16;
    GHM_HIP(hipMalloc(out, bytes));
    return 0;
}

int ghm_free(ghm_ctx* ctx, void* ptr) {
    GHM_CHECK(!ctx->capturing, "ghm_free inside graph capture");
    GHM_HIP(hipSetDevice(ctx->device));
    GHM_HIP(hipFree(ptr));
    return 0;
}

int ghm_h2d(ghm_ctx* ctx, void* dst, const void* src_host, size_t bytes) {
    GHM_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the runtime stages it, but keep the host buffer's lifetime simple
    if (!ctx->capturing) GHM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- asynchronous input path (no reference counterpart: pix2pix.py:201-212 hands train_fn pageable numpy arrays and
// waits): page-locked host staging + a copy that does NOT synchronise, so the batch of step i+1 travels while step i runs ----
int ghm_host_alloc(size_t bytes, void** out) {
    GHM_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return 0;
}

int ghm_host_free(void* ptr) {
    if (ptr) GHM_HIP(hipHostFree(ptr));
    return 0;
}

int ghm_h2d_async(ghm_ctx* ctx, void* dst, const void* src_pinned, size_t bytes) {
    // src_pinned must come from ghm_host_alloc and stay untouched until the context's stream has passed this copy
    GHM_CHECK(!ctx->capturing && !ctx->rec, "ghm_h2d_async inside a capture / recording");
    GHM_HIP(hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

// persistent events: a point of one context's stream that other contexts wait for LATER (ghm_stream_wait can only name
// "everything enqueued so far").  The input pipeline needs exactly that: the stage streams of step i wait for the upload of
// ITS batch, not for whatever the copy stream was told to do after it.
int ghm_event_create(ghm_ctx* ctx, void** out) {
    GHM_HIP(hipSetDevice(ctx->device));
    hipEvent_t ev;
    GHM_HIP(hipEventCreateWithFlags(&ev, ghm_event_flags(true)));      // the host waits on these (ghm_event_sync), copies sit behind them
    *out = (void*)ev;
    return 0;
}

int ghm_event_destroy(void* ev) {
    if (ev) GHM_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

// (inside a RECORDED step both are entries of the step: a replay records / waits again -- the per-net "parameters gathered"
// events of the sharded update, which a step records at its end and the NEXT replay's forward waits for; not inside a HIP graph
// capture)
int ghm_event_record(ghm_ctx* ctx, void* ev) {
    GHM_CHECK(!ctx->capturing || ctx->rec, "ghm_event_record inside a graph capture");
    hipStream_t s = ctx->stream;
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        st->cmds.emplace_back([=]() {
            hipError_t e = hipEventRecord((hipEvent_t)ev, s);
            if (e != hipSuccess && st->err == hipSuccess) st->err = e;
        });
        return 0;
    }
    GHM_HIP(hipEventRecord((hipEvent_t)ev, s));
    return 0;
}

int ghm_event_wait(ghm_ctx* ctx, void* ev) {
    GHM_CHECK(!ctx->capturing || ctx->rec, "ghm_event_wait inside a graph capture");
    hipStream_t s = ctx->stream;
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        st->cmds.emplace_back([=]() {
            hipError_t e = hipStreamWaitEvent(s, (hipEvent_t)ev, 0);
            if (e != hipSuccess && st->err == hipSuccess) st->err = e;
        });
        return 0;
    }
    GHM_HIP(hipStreamWaitEvent(s, (hipEvent_t)ev, 0));
    return 0;
}

int ghm_event_sync(void* ev) {
    GHM_HIP(hipEventSynchronize((hipEvent_t)ev));
    return 0;
}

// ---- hardware-queue probe ----
// ROCm multiplexes HIP streams onto a few hardware queues, and a stream that is blocked (or busy copying) stalls every
// stream that shares its queue.  ghm_queue_interference(a, b, probe) measures it: a long spin kernel runs on ``probe``,
// ``a`` is made to wait for it (hipStreamWaitEvent: a is now blocked for the spin's length), and a trivial kernel is timed
// on ``b``: *delay_us is how long b's kernel waited -- a few microseconds on its own queue, the spin's length on a's.
namespace {
__global__ void spin_kernel(long ticks) {           // wall_clock64(): the constant 100 MHz counter (10 ns per tick)
    const long t0 = (long)wall_clock64();
    long t = t0;
    for (int i = 0; i < (1 << 20) && t - t0 < ticks; ++i) {      // (bounded: can never hang the GPU)
        __builtin_amdgcn_s_sleep(64);
        t = (long)wall_clock64();
    }
}
__global__ void nop_kernel() {}
}  // namespace

int ghm_queue_interference(ghm_ctx* a, ghm_ctx* b, ghm_ctx* probe, int32_t spin_us, float* delay_us) {
    GHM_CHECK(a->device == b->device && a->device == probe->device, "ghm_queue_interference: one device");
    GHM_CHECK(!a->rec && !b->rec && !probe->rec && !a->capturing && !b->capturing, "ghm_queue_interference inside a recording");
    GHM_HIP(hipSetDevice(a->device));
    GHM_HIP(hipStreamSynchronize(a->stream));
    GHM_HIP(hipStreamSynchronize(b->stream));
    GHM_HIP(hipStreamSynchronize(probe->stream));
    hipEvent_t gate;
    GHM_HIP(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    // (the shader clock paces the spin: ~2.4 GHz when busy, lower from idle -- the spin is then LONGER than asked for,
    // which only sharpens the measurement)
    spin_kernel<<<1, 64, 0, probe->stream>>>((long)spin_us * 100L);
    GHM_HIP(hipEventRecord(gate, probe->stream));
    GHM_HIP(hipStreamWaitEvent(a->stream, gate, 0));         // a sits in this wait until the spin ends
    const auto h0 = std::chrono::steady_clock::now();
    nop_kernel<<<1, 64, 0, b->stream>>>();
    GHM_HIP(hipStreamSynchronize(b->stream));                // returns at once on b's own queue, after the spin on a's
    const auto h1 = std::chrono::steady_clock::now();
    GHM_HIP(hipStreamSynchronize(a->stream));
    GHM_HIP(hipStreamSynchronize(probe->stream));
    *delay_us = std::chrono::duration<float, std::micro>(h1 - h0).count();
    (void)hipEventDestroy(gate);
    return 0;
}

int ghm_d2h(ghm_ctx* ctx, void* dst_host, const void* src, size_t bytes) {
    GHM_CHECK(!ctx->capturing, "ghm_d2h inside graph capture");
    GHM_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GHM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ghm_d2d(ghm_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess && st->err == hipSuccess) st->err = e;
        });
        return 0;
    }
    GHM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int ghm_memset_zero(ghm_ctx* ctx, void* dst, size_t bytes) {
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        hipStream_t s = ctx->stream;
        st->cmds.emplace_back([=]() {
            hipError_t e = hipMemsetAsync(dst, 0, bytes, s);
            if (e != hipSuccess && st->err == hipSuccess) st->err = e;
        });
        return 0;
    }
    GHM_HIP(hipMemsetAsync(dst, 0, bytes, ctx->stream));
    return 0;
}

int ghm_sync(ghm_ctx* ctx) {
    GHM_CHECK(!ctx->capturing, "ghm_sync inside graph capture");
    GHM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ghm_stream_wait(ghm_ctx* ctx, ghm_ctx* other) {
    // everything enqueued on ``ctx`` after this call runs after everything enqueued on ``other`` so far
    GHM_CHECK(ctx->device == other->device, "ghm_stream_wait across devices");
    // TUNING ONLY (results are wrong: the streams race): no cross-stream ordering at all -- what the event machinery itself
    // costs in the overlapped schedule
    if (GHM_OPT("GHM_SKIP_STREAM_WAITS")) return 0;
    // an edge into or out of a communication stream orders memory that peer GPUs touch: system-scope release
    const bool sys_edge = ctx->comm != nullptr || other->comm != nullptr;
    if (ctx->rec) {
        ghm_step* st = ctx->rec;
        hipStream_t mine = ctx->stream, theirs = other->stream;
        if (GHM_OPT("GHM_REC_EVENT_PER_REPLAY")) {          // the round 1-4 form (A/B): an event created and destroyed per replay
            st->cmds.emplace_back([=]() {
                hipEvent_t ev;
                hipError_t e = hipEventCreateWithFlags(&ev, ghm_event_flags(sys_edge));
                if (e == hipSuccess) e = hipEventRecord(ev, theirs);
                if (e == hipSuccess) e = hipStreamWaitEvent(mine, ev, 0);
                if (e == hipSuccess) e = hipEventDestroy(ev);
                if (e != hipSuccess && st->err == hipSuccess) st->err = e;
            });
            return 0;
        }
        // one event per recorded wait, owned by the step: a replay re-records it (a wait holds the record it was given, so the
        // next replay's record does not disturb a wait still pending) -- two runtime calls per edge instead of four
        hipEvent_t ev;
        GHM_HIP(hipEventCreateWithFlags(&ev, ghm_event_flags(sys_edge)));
        st->events.push_back(ev);
        st->cmds.emplace_back([=]() {
            hipError_t e = hipEventRecord(ev, theirs);
            if (e == hipSuccess) e = hipStreamWaitEvent(mine, ev, 0);
            if (e != hipSuccess && st->err == hipSuccess) st->err = e;
        });
        return 0;
    }
    hipEvent_t ev;
    GHM_HIP(hipEventCreateWithFlags(&ev, ghm_event_flags(sys_edge)));
    GHM_HIP(hipEventRecord(ev, other->stream));
    GHM_HIP(hipStreamWaitEvent(ctx->stream, ev, 0));
    GHM_HIP(hipEventDestroy(ev));       // destruction is deferred by the runtime until the event has fired
    return 0;
}

int ghm_capture_begin(ghm_ctx* ctx) {
    GHM_CHECK(!ctx->capturing && !ctx->rec, "nested capture");
    GHM_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return 0;
}

int ghm_capture_end(ghm_ctx* ctx, ghm_graph** out) {
    GHM_CHECK(ctx->capturing, "capture_end without capture_begin");
    ctx->capturing = false;
    ghm_graph* g = new ghm_graph();
    hipError_t e = hipStreamEndCapture(ctx->stream, &g->graph);
    if (e != hipSuccess) {
        delete g;
        ghm_set_error("hipStreamEndCapture -> %s", hipGetErrorString(e));
        return -1;
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        delete g;
        ghm_set_error("hipGraphInstantiate -> %s", hipGetErrorString(e));
        return -1;
    }
    ++ctx->pinned;
    g->owner = ctx;
    ctx->graphs.push_back(g);
    *out = g;
    return 0;
}

int ghm_graph_launch(ghm_ctx* ctx, ghm_graph* g) {
    GHM_HIP(hipGraphLaunch(g->exec, ctx->stream));
    return 0;
}

int ghm_graph_destroy(ghm_graph* g) {
    if (!g) return 0;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->owner) {
        auto& v = g->owner->graphs;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == g) { v.erase(v.begin() + i); break; }
        ghm_unpin(g->owner);
    }
    delete g;
    return 0;
}

// ---- a whole train step as one call (struct ghm_step: common.h) ----
int ghm_step_build(int32_t n, ghm_ctx* const* ctxs, ghm_graph* const* graphs, ghm_step** out) {
    GHM_CHECK(n >= 1 && n <= 8, "ghm_step_build: 1..8 stages");
    ghm_step* s = new ghm_step();
    s->n = n;
    for (int i = 0; i < n; ++i) {
        GHM_CHECK(ctxs[i] && graphs[i], "ghm_step_build: null stage %d", i);
        s->ctx[i] = ctxs[i];
        s->graph[i] = graphs[i];
    }
    *out = s;
    return 0;
}

int ghm_step_record_begin(int32_t n, ghm_ctx* const* ctxs, ghm_step** out) {
    GHM_CHECK(n >= 1 && n <= 8, "ghm_step_record_begin: 1..8 contexts");
    for (int i = 0; i < n; ++i)
        GHM_CHECK(ctxs[i] && !ctxs[i]->rec && !ctxs[i]->capturing, "ghm_step_record_begin: context %d is busy", i);
    ghm_step* s = new ghm_step();
    s->n = n;
    s->recording = true;
    for (int i = 0; i < n; ++i) {
        s->ctx[i] = ctxs[i];
        ctxs[i]->rec = s;
        ctxs[i]->capturing = true;      // the same restrictions as graph capture: no allocation, no synchronisation
    }
    *out = s;
    return 0;
}

int ghm_step_record_end(ghm_step* s) {
    GHM_CHECK(s && s->recording, "ghm_step_record_end without ghm_step_record_begin");
    for (int i = 0; i < s->n; ++i) {
        s->ctx[i]->rec = nullptr;
        s->ctx[i]->capturing = false;
        ++s->ctx[i]->pinned;
    }
    s->recording = false;
    s->recorded = true;
    return 0;
}

int ghm_step_timer_stride(ghm_step* s, int32_t stride) {
    s->timer_stride = stride;
    return 0;
}

int ghm_step_run(ghm_step* s) {
    if (s->recorded) {
        GHM_CHECK(!s->recording, "ghm_step_run while the step is still being recorded");
        GHM_HIP(hipSetDevice(s->ctx[0]->device));
        for (auto& c : s->cmds) c();
        ++s->runs;
        if (s->err != hipSuccess) {
            ghm_set_error("recorded step: %s", hipGetErrorString(s->err));
            s->err = hipSuccess;
            return -1;
        }
        GHM_HIP(hipGetLastError());
        return 0;
    }
    for (int i = 0; i < s->n; ++i) GHM_HIP(hipGraphLaunch(s->graph[i]->exec, s->ctx[i]->stream));
    return 0;
}

int ghm_step_destroy(ghm_step* s) {
    if (s && s->recording) ghm_step_record_end(s);
    if (s && s->recorded)
        for (int i = 0; i < s->n; ++i) ghm_unpin(s->ctx[i]);
    if (s)
        for (hipEvent_t ev : s->events) (void)hipEventDestroy(ev);
    delete s;          // graphs stay owned by their creator (ghm_graph_destroy)
    return 0;
}

static void timer_record(ghm_ctx* ctx, int slot, bool start) {
    if (!ctx->ev_start[slot]) {
        (void)hipEventCreate(&ctx->ev_start[slot]);
        (void)hipEventCreate(&ctx->ev_stop[slot]);
    }
    (void)hipEventRecord(start ? ctx->ev_start[slot] : ctx->ev_stop[slot], ctx->stream);
}

int ghm_timer_start(ghm_ctx* ctx, int32_t slot) {
    GHM_CHECK(slot >= 0 && slot < GHM_MAX_TIMERS, "timer slot out of range");
    if (ctx->rec) {         // recorded: the slot advances by the step's timer stride on every replay
        ghm_step* st = ctx->rec;
        st->cmds.emplace_back([=]() { timer_record(ctx, (int)((slot + st->runs * st->timer_stride) % GHM_MAX_TIMERS), true); });
        return 0;
    }
    if (!ctx->ev_start[slot]) {
        GHM_HIP(hipEventCreate(&ctx->ev_start[slot]));
        GHM_HIP(hipEventCreate(&ctx->ev_stop[slot]));
    }
    GHM_HIP(hipEventRecord(ctx->ev_start[slot], ctx->stream));
    return 0;
}

int ghm_timer_stop(ghm_ctx* ctx, int32_t slot) {
    if (ctx->rec) {
        GHM_CHECK(slot >= 0 && slot < GHM_MAX_TIMERS, "timer slot out of range");
        ghm_step* st = ctx->rec;
        st->cmds.emplace_back([=]() { timer_record(ctx, (int)((slot + st->runs * st->timer_stride) % GHM_MAX_TIMERS), false); });
        return 0;
    }
    GHM_CHECK(slot >= 0 && slot < GHM_MAX_TIMERS && ctx->ev_stop[slot], "timer slot not started");
    GHM_HIP(hipEventRecord(ctx->ev_stop[slot], ctx->stream));
    return 0;
}

int ghm_timer_elapsed_ms(ghm_ctx* ctx, int32_t slot, float* ms) {
    GHM_CHECK(slot >= 0 && slot < GHM_MAX_TIMERS && ctx->ev_stop[slot], "timer slot not started");
    GHM_HIP(hipEventSynchronize(ctx->ev_stop[slot]));
    GHM_HIP(hipEventElapsedTime(ms, ctx->ev_start[slot], ctx->ev_stop[slot]));
    return 0;
}

}  // extern "C"
