// engine.hip -- the hand-scheduled per-layer pipeline that replaces the reference's ggml graphs
// (rwkv_build_serial_graph / rwkv_build_sequential_graph, rwkv_graph.inc:611-866). One code path serves T = 1 (decode)
// and T > 1 (sequence mode); state and weights never leave HBM between calls.
#include "model.h"

#include <condition_variable>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <thread>

namespace rwkvmi {

#define HIP_CTX_OK(CTX, CALL) \
    do { hipError_t e_ = (CALL); RW_CTX_CHECK((CTX), RWKV_ERROR_GRAPH, false, e_ == hipSuccess, "HIP error: %s", hipGetErrorString(e_)); } while (0)

void ctx_fail(struct ::rwkv_context * ctx, int flags, const char * file, int line, const char * expr, const char * fmt, ...) {
    ctx->last_error |= flags;
    if (!ctx->print_errors) return;
    if (fmt && fmt[0]) {
        va_list ap;
        va_start(ap, fmt);
        vfprintf(stderr, fmt, ap);
        va_end(ap);
    }
    fprintf(stderr, "\n%s:%d: %s\n", file, line, expr);
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------------------------
// The persistent decode kernel needs every CU of its device at once. Two contexts of one process (rwkv_clone_context: one
// clone per thread, rwkv.h:64-68) launching it concurrently on their own streams could each become half resident and spin
// until both time out. Launches are therefore chained per device: a context's launch waits (on the device, not the host) for
// the completion event of the previous persistent launch of the process. Other processes cannot be seen from here: a poll
// time-out is recovered from by dropping to the per-layer launches (recover_from_abort).
// ---------------------------------------------------------------------------------------------------------------
static constexpr int k_max_devices = 64;
static std::mutex g_mega_mu[k_max_devices + 1];   // per device: a streamed rwkv_eval holds it across its host-blocking slice uploads, other devices go on
static hipEvent_t g_mega_last[k_max_devices] = {};
static rwkv_context * g_mega_last_owner[k_max_devices] = {};
static int g_mega_contexts[k_max_devices] = {};   // contexts of this process holding a persistent kernel, per device

static std::mutex & mega_mu(int dev) { return g_mega_mu[(dev >= 0 && dev < k_max_devices) ? dev : k_max_devices]; }
static int mega_chain_count(rwkv_context * ctx, int delta) {
    const int dev = ctx->model->device;
    std::lock_guard<std::mutex> lk(mega_mu(dev));
    if (dev < 0 || dev >= k_max_devices) return 0;
    return g_mega_contexts[dev] += delta;
}

static void mega_chain_begin(rwkv_context * ctx) {
    const int dev = ctx->model->device;
    mega_mu(dev).lock();
    // (chain_covered: a pipeline hop has already made this stream wait on exactly that launch -- one barrier packet in front of the launch, not two)
    if (dev >= 0 && dev < k_max_devices && g_mega_last[dev] && g_mega_last_owner[dev] != ctx && g_mega_last[dev] != ctx->chain_covered) (void) hipStreamWaitEvent(ctx->stream, g_mega_last[dev], 0);
}
static void mega_chain_end(rwkv_context * ctx) {
    const int dev = ctx->model->device;
    // (a single persistent context per device -- the usual case -- records nothing: no marker between graph replays)
    if (dev >= 0 && dev < k_max_devices && g_mega_contexts[dev] > 1 && ctx->mega_done && hipEventRecord(ctx->mega_done, ctx->stream) == hipSuccess) {
        g_mega_last[dev] = ctx->mega_done; g_mega_last_owner[dev] = ctx;
    }
    mega_mu(dev).unlock();
}
hipEvent_t mega_chain_marker(rwkv_context * ctx) {
    const int dev = ctx->model->device;
    if (dev < 0 || dev >= k_max_devices) return nullptr;
    std::lock_guard<std::mutex> lk(mega_mu(dev));
    return (g_mega_last_owner[dev] == ctx && g_mega_last[dev] == ctx->mega_done) ? ctx->mega_done : nullptr;
}
static void mega_chain_forget(rwkv_context * ctx) {
    const int dev = ctx->model->device;
    std::lock_guard<std::mutex> lk(mega_mu(dev));
    if (dev >= 0 && dev < k_max_devices && g_mega_last_owner[dev] == ctx) {
        if (g_mega_last[dev]) (void) hipEventSynchronize(g_mega_last[dev]);
        g_mega_last[dev] = nullptr; g_mega_last_owner[dev] = nullptr;
    }
}

// what decided a context's single-token path, in words (rwkv_mi_persist_info)
static const char * kind_name(int k) { return k == 2 ? "ring" : (k == 1 ? "regs" : (k == 3 ? "k47" : "none")); }
static void note(rwkv_context * ctx, const char * fmt, ...) {
    char buf[320];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (!ctx->persist_note.empty()) ctx->persist_note += "; ";
    ctx->persist_note += buf;
}

// Which single-token path is fastest depends on the device: the persistent kernels are bound by cross-XCD hand-over latency, the
// seven-launch path by launch boundaries (measured: 1.5-1.6 ms vs 2.6 ms per token on most MI355X boxes, but on some boxes four of
// the eight XCDs lag and a chain of all-to-all hand-overs runs at their pace: 3.4 ms vs 2.9 ms). A few eager tokens on zeroed state
// settle it per context at creation: the persistent kernel on the LDS-DMA weight ring (ring_v6.hip), the one on register prefetch
// (mega_v6.hip; only built for the comparison when RWKV_MI_PERSIST does not name one) and the seven launches are timed, the fastest
// stays, the others are freed. The state is (re)initialised by every caller afterwards.
static void calibrate_decode_path(rwkv_context * ctx) {
    if (!ctx->mega || !ctx->fused_v6) return;
    const char * e = getenv("RWKV_MI_NO_AUTOTUNE");
    if (e && e[0] == '1') return;
    Model & m = *ctx->model;
    uint32_t * tok = nullptr;
    if (hipMalloc((void **) &tok, 256) != hipSuccess) return;
    const size_t sbytes = (size_t) m.state_len() * sizeof(float);
    bool ok = hipMemsetAsync(tok, 0, 256, ctx->stream) == hipSuccess;
    for (int i = 0; i < 2; i++) ok = ok && hipMemsetAsync(ctx->state[i], 0, sbytes, ctx->stream) == hipSuccess;
    uint32_t * saved_tokens = ctx->d_tokens;
    ctx->d_tokens = tok;
    // candidates: [0] the handle create_context made, [1] the other persistent kernel (when the environment names none), fused = nullptr
    void * cand[2] = {ctx->mega, nullptr};
    const char * pk = getenv("RWKV_MI_PERSIST");
    // (the register-prefetch kernel only where it has been seen within 2 % of the ring: D = 2048 -- profiles/r04y_prefill_1b6_q4_0_kernel_stats.csv
    //  calibration rows, 710 vs 726 us; at D = 4096 / 2560 the ring wins by 10 % and more and timing a third path cost every context
    //  creation ~15 ms. RWKV_MI_PERSIST=regs still names it.)
    if (!(pk && pk[0]) && mega_v6_kind(cand[0]) == 2 && m.n_embed() == 2048) cand[1] = mega_v6_create_kind(m, 1);
    // (with logits: the ring kernel runs ln_out + head inside its launch, the other paths as launches of their own -- part of what is compared)
    auto run = [&](void * h, int n) { ctx->mega = h; for (int i = 0; i < n && ok; i++) ok = forward(ctx, 1, m.has_head); };
    auto timed = [&](void * h) -> float {
        run(h, 2);
        ok = ok && hipEventRecord(ctx->ev0, ctx->stream) == hipSuccess;
        run(h, 6);
        ok = ok && hipEventRecord(ctx->ev1, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        float ms = 0.0f;
        ok = ok && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess;
        return ms;
    };
    float t[2] = {1e30f, 1e30f};
    bool bad[2] = {false, true};
    for (int i = 0; i < 2; i++) {
        if (!cand[i]) continue;
        t[i] = timed(cand[i]);
        bad[i] = !ok || mega_v6_aborted(cand[i], ctx->stream);
        if (bad[i]) { if (mega_v6_aborted_cached(cand[i])) (void) mega_v6_clear_abort(cand[i], ctx->stream); ok = true; t[i] = 1e30f; }
    }
    const float t_fused = timed(nullptr);
    (void) hipStreamSynchronize(ctx->stream);
    ctx->d_tokens = saved_tokens;
    ctx->cur = 0;
    ctx->last_error = 0;
    (void) hipFree(tok);
    const int best = t[1] < t[0] ? 1 : 0;
    const bool keep = ok && !bad[best] && !(t_fused < 0.97f * t[best]);
    for (int i = 0; i < 2; i++) if (cand[i] && bad[i]) note(ctx, "calibration: the %s kernel timed out (not every workgroup resident)", kind_name(mega_v6_kind(cand[i])));
    if (cand[best] && !bad[best]) note(ctx, "calibration: %s %.3f ms / token against %.3f ms for the per-layer launches: %s", kind_name(mega_v6_kind(cand[best])), t[best] / 6.0f, t_fused / 6.0f, keep ? "kept" : "dropped");
    for (int i = 0; i < 2; i++) if (cand[i] && !(keep && i == best)) mega_v6_destroy(cand[i]);
    ctx->mega = keep ? cand[best] : nullptr;
    if (ok) m.decode_choice.store(keep ? mega_v6_kind(cand[best]) : 3);
    if (!keep) { mega_chain_forget(ctx); mega_chain_count(ctx, -1); }
}

// The same question for RWKV-4 / RWKV-7: the persistent launch of persist_v47.hip against the fused per-layer launches (fused_v7.hip).
static void calibrate_decode_path_v47(rwkv_context * ctx) {
    if (!ctx->mega) return;
    const char * e = getenv("RWKV_MI_NO_AUTOTUNE");
    if (e && e[0] == '1') return;
    Model & m = *ctx->model;
    uint32_t * tok = nullptr;
    if (hipMalloc((void **) &tok, 256) != hipSuccess) return;
    bool ok = hipMemsetAsync(tok, 0, 256, ctx->stream) == hipSuccess;
    uint32_t * saved_tokens = ctx->d_tokens;
    ctx->d_tokens = tok;
    void * const h = ctx->mega;
    auto timed = [&](void * hh) -> float {
        ctx->mega = hh;
        ctx->cur = 0;
        ok = ok && state_from_host(ctx, nullptr);
        for (int i = 0; i < 2 && ok; i++) ok = forward(ctx, 1, m.has_head);
        ok = ok && hipEventRecord(ctx->ev0, ctx->stream) == hipSuccess;
        for (int i = 0; i < 6 && ok; i++) ok = forward(ctx, 1, m.has_head);
        ok = ok && hipEventRecord(ctx->ev1, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        float ms = 0.0f;
        ok = ok && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess;
        return ms;
    };
    float t_p = timed(h);
    bool bad = !ok || mega_v6_aborted(h, ctx->stream);
    if (bad) { if (mega_v6_aborted_cached(h)) (void) mega_v6_clear_abort(h, ctx->stream); ok = true; t_p = 1e30f; }
    float t_fused = timed(nullptr);
    // six tokens each on a device that may be busy: a margin under 10 % is timed once more and the smaller figures decide (the choice is kept for
    // every later context of the model)
    if (!bad && ok && fabsf(t_fused - t_p) < 0.10f * t_p) {
        const float p2 = timed(h);
        const bool bad2 = !ok || mega_v6_aborted(h, ctx->stream);
        if (bad2) { if (mega_v6_aborted_cached(h)) (void) mega_v6_clear_abort(h, ctx->stream); ok = true; bad = true; t_p = 1e30f; }
        else { t_p = p2 < t_p ? p2 : t_p; const float f2 = timed(nullptr); t_fused = f2 < t_fused ? f2 : t_fused; }
    }
    (void) hipStreamSynchronize(ctx->stream);
    ctx->d_tokens = saved_tokens;
    ctx->cur = 0;
    ctx->last_error = 0;
    (void) hipFree(tok);
    const bool keep = ok && !bad && !(t_fused < 0.97f * t_p);
    if (bad) note(ctx, "calibration: the k47 kernel timed out (not every workgroup resident)");
    else note(ctx, "calibration: k47 %.3f ms / token against %.3f ms for the per-layer launches: %s", t_p / 6.0f, t_fused / 6.0f, keep ? "kept" : "dropped");
    if (!keep) mega_v6_destroy(h);
    ctx->mega = keep ? h : nullptr;
    if (ok) m.decode_choice.store(keep ? 4 : 3);
    if (!keep) { mega_chain_forget(ctx); mega_chain_count(ctx, -1); }
}

// After a poll time-out of the persistent kernel (the device was shared): the stream is drained, the abort word cleared, the
// persistent path and the captured graphs dropped; the context continues on the per-layer launches. The state buffer the failed
// step READ is intact (the kernel only writes the other one): the caller may flip `cur` back and repeat the step.
void recover_from_abort(rwkv_context * ctx) {
    if (!ctx->mega) return;
    note(ctx, "a poll of the %s kernel timed out at run time (not every workgroup resident: is the GPU shared?): fell back to the per-layer launches", kind_name(mega_v6_kind(ctx->mega)));
    (void) hipStreamSynchronize(ctx->stream);
    (void) mega_v6_clear_abort(ctx->mega, ctx->stream);
    mega_chain_forget(ctx);
    mega_chain_count(ctx, -1);
    mega_v6_destroy(ctx->mega);
    ctx->mega = nullptr;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (ctx->graph_exec[a][b]) { (void) hipGraphExecDestroy(ctx->graph_exec[a][b]); ctx->graph_exec[a][b] = nullptr; }
}

rwkv_context * create_context(Model * m, uint32_t n_threads) {
    std::unique_ptr<rwkv_context> ctx(new (std::nothrow) rwkv_context());
    if (!ctx) {
        // a model that no context references yet (rwkv_init_from_file / rwkv_mi_init_stage) would be orphaned
        if (m->refcount.load() == 0) { m->refcount++; release_model(m); }
        global_fail(RWKV_ERROR_CTX | RWKV_ERROR_ALLOC, __FILE__, __LINE__, "ctx != nullptr", "Failed to allocate rwkv_context");
        return nullptr;
    }
    ctx->model = m;
    ctx->n_threads = n_threads;
    m->refcount++;
    auto fail = [&](hipError_t e) {
        global_fail(RWKV_ERROR_CTX | RWKV_ERROR_ALLOC, __FILE__, __LINE__, "hip allocation", "HIP error: %s", hipGetErrorString(e));
        destroy_context(ctx.release());
        return (rwkv_context *) nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(m->device)) != hipSuccess) return fail(e);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e);
    const size_t sbytes = (size_t) m->state_len() * sizeof(float);
    for (int i = 0; i < 2; i++) if ((e = hipMalloc((void **) &ctx->state[i], sbytes)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **) &ctx->d_logits, (size_t) m->n_vocab() * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void **) &ctx->d_next_token, 64)) != hipSuccess) return fail(e);
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return fail(e);
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return fail(e);
    if ((e = hipEventCreateWithFlags(&ctx->mega_done, hipEventDisableTiming)) != hipSuccess) return fail(e);
    prefill_prepare_current_device();   // dynamic-LDS limits of the sequence-mode kernels: per device
    const char * g = getenv("RWKV_MI_NO_GRAPH");
    ctx->use_graph = !(g && g[0] == '1');
    const char * nf = getenv("RWKV_MI_NO_FUSED");
    if (!(nf && nf[0] == '1') && fused_v6_supported(*m)) {
        if ((e = hipMalloc(&ctx->fused_scratch, fused_v6_scratch_bytes(*m))) != hipSuccess) return fail(e);
        ctx->fused_v6 = true;
        const char * nm = getenv("RWKV_MI_NO_MEGA");
        const int known = m->decode_choice.load();      // (what an earlier context of this model measured)
        if (!(nm && nm[0] == '1') && known != 3) ctx->mega = known ? mega_v6_create_kind(*m, known) : mega_v6_create(*m);
        if (nm && nm[0] == '1') note(ctx.get(), "RWKV_MI_NO_MEGA=1");
        else if (known == 3) note(ctx.get(), "an earlier context of this model measured the per-layer launches faster");
        else if (!ctx->mega) { const char * why = persist_unavailable_reason(*m); note(ctx.get(), "%s", why ? why : "the persistent kernel could not be built (device memory?)"); }
        // a second persistent context on this device: launches the first one made while it was alone carry no completion event
        if (ctx->mega && mega_chain_count(ctx.get(), +1) > 1) (void) hipDeviceSynchronize();
        if (!known) calibrate_decode_path(ctx.get());
    }
    if (!(nf && nf[0] == '1') && fused_v7_supported(*m)) {
        if ((e = hipMalloc(&ctx->fused_scratch, fused_v7_scratch_bytes(*m))) != hipSuccess) return fail(e);
        ctx->fused_v7 = true;
    }
    if (!(nf && nf[0] == '1') && fused_v4_supported(*m)) {
        if ((e = hipMalloc(&ctx->fused_scratch, fused_v4_scratch_bytes(*m))) != hipSuccess) return fail(e);
        ctx->fused_v4 = true;
    }
    if (ctx->fused_v7 || ctx->fused_v4) {
        // one persistent launch per token (persist_v47.hip) where the geometry has a variant; the fused launches stay as the fall-back
        const char * nm = getenv("RWKV_MI_NO_MEGA");
        const int known = m->decode_choice.load();
        if (!(nm && nm[0] == '1') && known != 3) ctx->mega = p47_create(*m);
        if (nm && nm[0] == '1') note(ctx.get(), "RWKV_MI_NO_MEGA=1");
        else if (known == 3) note(ctx.get(), "an earlier context of this model measured the per-layer launches faster");
        else if (!ctx->mega) { const char * why = persist_unavailable_reason(*m); note(ctx.get(), "%s", why ? why : "the persistent kernel could not be built (device memory?)"); }
        if (ctx->mega && mega_chain_count(ctx.get(), +1) > 1) (void) hipDeviceSynchronize();
        if (!known) calibrate_decode_path_v47(ctx.get());
    }
    if (!ctx->fused_v6 && !ctx->fused_v7 && !ctx->fused_v4) { const char * why = persist_unavailable_reason(*m); note(ctx.get(), "%s", (nf && nf[0] == '1') ? "RWKV_MI_NO_FUSED=1" : (why ? why : "no fused layer for this model")); }
    // A new context starts from the reference's fresh state (rwkv_eval.inc:224-241), whatever the calibration left behind:
    // rwkv_mi_eval_resident / rwkv_mi_decode_greedy / rwkv_mi_stage_step continue from the resident state without a load.
    ctx->cur = 0;
    if (!state_from_host(ctx.get(), nullptr) || (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e);
    ctx->last_error = 0;
    return ctx.release();
}

static void drop_graphs(rwkv_context * ctx) {
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (ctx->graph_exec[a][b]) { (void) hipGraphExecDestroy(ctx->graph_exec[a][b]); ctx->graph_exec[a][b] = nullptr; }
}

void destroy_context(rwkv_context * ctx) {
    if (!ctx) return;
    if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
    if (ctx->abi_streamer) { abi_streamer_free(ctx->abi_streamer); ctx->abi_streamer = nullptr; }
    drop_graphs(ctx);
    for (int i = 0; i < 2; i++) if (ctx->state[i]) (void) hipFree(ctx->state[i]);
    if (ctx->scratch) (void) hipFree(ctx->scratch);
    if (ctx->fused_scratch) (void) hipFree(ctx->fused_scratch);
    if (ctx->mega) { mega_chain_forget(ctx); mega_chain_count(ctx, -1); mega_v6_destroy(ctx->mega); }
    if (ctx->mega_done) (void) hipEventDestroy(ctx->mega_done);
    if (ctx->d_tokens) (void) hipFree(ctx->d_tokens);
    if (ctx->d_logits) (void) hipFree(ctx->d_logits);
    if (ctx->d_next_token) (void) hipFree(ctx->d_next_token);
    if (ctx->d_probs) (void) hipFree(ctx->d_probs);
    if (ctx->d_rng_counter) (void) hipFree(ctx->d_rng_counter);
    if (ctx->h_tokens) (void) hipHostFree(ctx->h_tokens);
    if (ctx->ev0) (void) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void) hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->prof.events) (void) hipEventDestroy(e);
    if (ctx->stream && ctx->owns_stream) { matvec_f_release_stream(ctx->stream); (void) hipStreamDestroy(ctx->stream); }
    release_model(ctx->model);
    delete ctx;
}

// Scratch for T tokens: one allocation carved into named activations.
bool ensure_scratch(rwkv_context * ctx, int64_t T) {
    if (T <= ctx->scratch_T) return true;
    Model & m = *ctx->model;
    const size_t D = (size_t) m.n_embed(), F = (size_t) m.ffn_size;
    const size_t LR = (size_t)(m.max_lowrank > 0 ? m.max_lowrank : 32);
    const size_t KQ = D > F ? D : F;  // widest quantised activation row
    auto fsz = [&](size_t n) { return align_up((size_t) T * n * sizeof(float), 256); };
    const size_t n_D = 3 + 6 + 6 + 3 + 1 + 1;  // x xn sx | m[6] | r k v g w a | t0 t1 t2 | out | v_first
    size_t total = n_D * fsz(D) + fsz(F) + 2 * fsz(LR) + align_up(D * sizeof(float), 256);
    total += align_up((size_t) T * KQ, 256) + 3 * align_up((size_t) T * (KQ / 32) * 4, 256);
    constexpr size_t k_ws_part = (size_t) 16 << 20;   // split-walk partial sums: up to 64 tiles x 8 parts x 32 KiB
    constexpr int    k_ws_counters = 1024;
    if (T >= k_mfma_min_tokens) total += align_up(tile_act_bytes(T, (int64_t) KQ), 256) + 5 * align_up(tile_act_bytes(T, (int64_t) D), 256) + k_ws_part + k_ws_counters * sizeof(int);
    if (ctx->scratch) { HIP_CTX_OK(ctx, hipStreamSynchronize(ctx->stream)); (void) hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_T = 0; }
    drop_graphs(ctx);
    HIP_CTX_OK(ctx, hipMalloc(&ctx->scratch, total));
    ctx->scratch_bytes = total;
    uint8_t * p = (uint8_t *) ctx->scratch;
    auto takef = [&](size_t n) { float * r = (float *) p; p += fsz(n); return r; };
    auto & b = ctx->b;
    b.x = takef(D); b.xn = takef(D); b.sx = takef(D);
    for (int i = 0; i < 6; i++) b.m[i] = takef(D);
    b.r = takef(D); b.k = takef(D); b.v = takef(D); b.g = takef(D); b.w = takef(D); b.a = takef(D);
    b.t0 = takef(D); b.t1 = takef(D); b.t2 = takef(D); b.out = takef(D); b.v_first = takef(D);
    b.ffk = takef(F); b.lr1 = takef(LR); b.lr2 = takef(LR);
    b.xlast = (float *) p; p += align_up(D * sizeof(float), 256);
    b.qa.q = (int8_t *) p; p += align_up((size_t) T * KQ, 256);
    b.qa.d = (float *) p; p += align_up((size_t) T * (KQ / 32) * 4, 256);
    b.qa.s = (float *) p; p += align_up((size_t) T * (KQ / 32) * 4, 256);
    b.qa.isum = (int *) p; p += align_up((size_t) T * (KQ / 32) * 4, 256);
    b.tile = nullptr;
    b.ws = MmqWs();
    for (int i = 0; i < 5; i++) b.tiles[i] = nullptr;
    if (T >= k_mfma_min_tokens) {
        b.tile = p; p += align_up(tile_act_bytes(T, (int64_t) KQ), 256);
        for (int i = 0; i < 5; i++) { b.tiles[i] = p; p += align_up(tile_act_bytes(T, (int64_t) D), 256); }
        b.ws.part = (float *) p; b.ws.part_bytes = k_ws_part; p += k_ws_part;
        b.ws.counters = (int *) p; b.ws.n_counters = k_ws_counters; p += k_ws_counters * sizeof(int);
        HIP_CTX_OK(ctx, hipMemsetAsync(b.ws.counters, 0, k_ws_counters * sizeof(int), ctx->stream));
    }
    ctx->scratch_T = T;
    return true;
}

std::atomic<int> g_test_fail_state_init{0};   // (tests: librwkv_testhooks.so arms it; the next state initialisation fails once)

bool state_from_host(rwkv_context * ctx, const float * state_in) {
    Model & m = *ctx->model;
    if (g_test_fail_state_init.load(std::memory_order_relaxed) > 0 && g_test_fail_state_init.fetch_sub(1) > 0)
        RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH, false, false, "state initialisation failed (injected by the test hook)");
    float * dst = ctx->state[ctx->cur];
    if (state_in) {
        HIP_CTX_OK(ctx, hipMemcpyAsync(dst, state_in, (size_t) m.state_len() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    } else if (m.arch_major >= 5) {
        HIP_CTX_OK(ctx, hipMemsetAsync(dst, 0, (size_t) m.state_len() * sizeof(float), ctx->stream));
    } else {
        launch_fill_state_v4(dst, m.n_layer(), m.n_embed(), ctx->stream);
    }
    return true;
}

bool state_to_host(rwkv_context * ctx, float * state_out) {
    Model & m = *ctx->model;
    HIP_CTX_OK(ctx, hipMemcpyAsync(state_out, ctx->state[ctx->cur], (size_t) m.state_len() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// layer pipeline
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct Runner {
    rwkv_context * ctx;
    Model & m;
    hipStream_t st;
    int64_t T, D, H, S;
    rwkv_context::Buf & b;

    bool failed = false;   // a launch of the pass could not be made (allocation failure): its outputs are not valid

    // inputs quantised ahead of their products, several per launch (sequence mode): source pointer -> tile image
    struct Pre { const float * x = nullptr; int wtype = -1; int64_t K = 0; TileAct ta; } pre[5];
    void prequant(int n, const float * const * xs, int64_t K, int wtype) {
        for (auto & e : pre) e = Pre();
        if (!(dtype_quantized(wtype) && T >= k_mfma_min_tokens && b.tiles[0] && K == D && n <= 5)) return;
        TileAct tas[5];
        for (int i = 0; i < n; i++) { tas[i] = tile_act_at(b.tiles[i], T, K); pre[i].x = xs[i]; pre[i].wtype = wtype; pre[i].K = K; pre[i].ta = tas[i]; }
        launch_quantize_act_tiles_batched(n, xs, T, K, wtype, tas, st);
    }
    const TileAct * find_pre(const float * x, int64_t K, int wtype) const {
        for (auto & e : pre) if (e.x == x && e.K == K && e.wtype == wtype) return &e.ta;
        return nullptr;
    }
    void drop_pre() { for (auto & e : pre) e = Pre(); }
    // tile images for n outputs of a fused mix (sequence mode, all consumers quantised with `wtype`): registered like prequant's
    bool fused_outs(int n, float * const * keys, int wtype, TileAct * tas) {
        if (!(dtype_quantized(wtype) && T >= k_mfma_min_tokens && b.tiles[0] && D % 256 == 0 && n <= 5)) return false;
        for (auto & e : pre) e = Pre();
        for (int i = 0; i < n; i++) { tas[i] = tile_act_at(b.tiles[i], T, D); pre[i].x = keys[i]; pre[i].wtype = wtype; pre[i].K = D; pre[i].ta = tas[i]; }
        return true;
    }

    // y_i[T][N] = epi_i(W_i . x_i[T][K]) for up to 4 matrices of one shape: one launch in sequence mode when every input was quantised ahead
    void mm_batch(int n, const DevTensor * const * Ws, const float * const * xs, float * const * ys, const Epi * epis) {
        bool batched = T >= k_mfma_min_tokens && b.tile && !ctx->prof.on;
        TileAct tas[4];
        for (int i = 0; i < n && batched; i++) {
            const TileAct * ta = find_pre(xs[i], Ws[i]->cols(), Ws[i]->type);
            if (!ta || Ws[i]->type != Ws[0]->type || Ws[i]->rows() != Ws[0]->rows() || Ws[i]->cols() != Ws[0]->cols()) batched = false; else tas[i] = *ta;
        }
        if (batched && launch_mmq_mfma_batched(n, Ws, tas, ys, epis, T, Ws[0]->rows(), &b.ws, st)) return;
        for (int i = 0; i < n; i++) mm(Ws[i], xs[i], ys[i], epis[i]);
    }

    // y[T][N] = epi(W . x[T][K])    (ggml_mul_mat)
    void mm(const DevTensor * W, const float * x, float * y, const Epi & epi = Epi()) {
        const int64_t N = W->rows(), K = W->cols();
        // (also for the narrow / short low-rank matrices of RWKV-6: routing those to the token-tiled kernel was measured -- 33.6 instead of
        //  23.6 ms per 1024-token pass)
        if (dtype_quantized(W->type) && T >= k_mfma_min_tokens && b.tile) {
            // sequence mode: int8 GEMM on the matrix cores (prefill.hip), bit-identical to the single-token kernel per (row, token)
            TileAct ta;
            if (const TileAct * p = find_pre(x, K, W->type)) ta = *p;
            else { ta = tile_act_at(b.tile, T, K); launch_quantize_act_tiles(x, T, K, W->type, ta, st); }
            auto & pf = ctx->prof;
            if (pf.on) {   // live per-launch timing of the GEMM (rwkv_mi_profile_prefill): `bytes` carries the launch's integer operations
                if (pf.used * 2 + 2 > pf.events.size()) {
                    hipEvent_t a = nullptr, c = nullptr;
                    (void) hipEventCreate(&a); (void) hipEventCreate(&c);
                    pf.events.push_back(a); pf.events.push_back(c); pf.bytes.push_back(0);
                }
                pf.bytes[pf.used] = 2ull * (uint64_t) T * (uint64_t) N * (uint64_t) K;
                (void) hipEventRecord(pf.events[pf.used * 2], st);
            }
            // (the tile-major weight image or the walk table could not be allocated: the product was NOT computed -- the pass fails,
            //  forward() reports it; the f32 activations may not even exist here when the mix wrote tile images only)
            if (!launch_mmq_mfma(*W, ta, T, y, N, epi, st, &b.ws)) { ctx->last_error |= RWKV_ERROR_GRAPH | RWKV_ERROR_ALLOC; failed = true; }
            if (pf.on) { (void) hipEventRecord(pf.events[pf.used * 2 + 1], st); pf.used++; }
        } else if (dtype_quantized(W->type)) {
            launch_quantize_act(x, T, K, b.qa, st);
            auto & pf = ctx->prof;
            const bool timed = pf.on && T == 1 && W->type == (int) m.header.data_type;
            if (timed) {
                if (pf.used * 2 + 2 > pf.events.size()) {
                    hipEvent_t a = nullptr, c = nullptr;
                    (void) hipEventCreate(&a); (void) hipEventCreate(&c);
                    pf.events.push_back(a); pf.events.push_back(c); pf.bytes.push_back(0);
                }
                // algorithmic bytes of this launch: the weight rows once + the quantised activation + the outputs
                pf.bytes[pf.used] = W->nbytes + (uint64_t) K + (uint64_t)(K / 32) * 12 + (uint64_t) N * 4;
                (void) hipEventRecord(pf.events[pf.used * 2], st);
            }
            launch_matvec_q(*W, b.qa, T, y, N, epi, st);
            if (timed) { (void) hipEventRecord(pf.events[pf.used * 2 + 1], st); pf.used++; }
        } else {
            launch_matvec_f(*W, x, K, T, y, N, epi, st);
        }
    }
    // group norm (+ gate) of the WKV output and the output projection behind it; sequence mode: the norm writes the projection's quantised
    // input image itself (k_groupnorm_seq_q), the f32 values never leave the chip
    void gn_out(const LayerW & L, float eps, const float * gate) {
        TileAct ta[1]; float * keys[1] = {b.out};
        if (fused_outs(1, keys, L.att_output->type, ta) && !ctx->prof.on &&
            launch_groupnorm_seq_q(b.out, f(L.att_ln_x_w), f(L.att_ln_x_b), eps, gate, T, H, S, ta[0], L.att_output->type, st)) {}
        else { drop_pre(); launch_groupnorm(b.out, f(L.att_ln_x_w), f(L.att_ln_x_b), eps, gate, nullptr, nullptr, nullptr, nullptr, T, H, S, st); }
        mm(L.att_output, b.out, b.x, epi(EPI_ADD_RES, nullptr, b.x));
        drop_pre();
    }
    // WKV-5/6: long sequences on the lane-pipelined kernel (one wave per value column), otherwise one wave per head
    void wkv6(const float * r, const float * k, const float * v, const float * u, int u_per_chan, const float * w, int w_mode,
              const float * state_in, float * state_out, float * out) {
        if (S == 64 && T >= k_mfma_min_tokens) launch_wkv6_seq(r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H, st);
        else launch_wkv6(r, k, v, u, u_per_chan, w, w_mode, state_in, state_out, out, T, H, S, st);
    }
    static const float * f(const DevTensor * t) { return (const float *) t->data; }
    static Epi epi(int op, const float * bias = nullptr, const float * res = nullptr, const float * aux = nullptr) {
        Epi e; e.op = op; e.bias = bias; e.res = res; e.aux = aux; return e;
    }

    // channel mixing (rwkv_ffn_v4_v5 :484-511, rwkv_ffn_v6 :513-531, rwkv_ffn_v7 :533-543)
    void ffn(const LayerW & L, const float * sin, float * sout) {
        launch_layernorm(b.x, T, D, f(L.ln2_w), f(L.ln2_b), b.xn, st);
        MixArgs a; a.xn = b.xn; a.carry_in = sin; a.carry_out = sout;
        if (m.arch_major <= 5) { a.mode = 0; a.n_out = 2; a.coef[0] = f(L.ffn_time_mix_k); a.coef[1] = f(L.ffn_time_mix_r); }
        else if (m.arch_major == 6) { a.mode = 1; a.n_out = 2; a.coef[0] = f(L.ffn_time_maa_k); a.coef[1] = f(L.ffn_time_maa_r); }
        else { a.mode = 1; a.n_out = 1; a.coef[0] = f(L.ffn_x_k); }
        a.out[0] = b.m[0]; a.out[1] = b.m[1];
        TileAct tas[5];
        if ((m.arch_major == 7 || L.ffn_receptance->type == L.ffn_key->type) && L.ffn_key->cols() == D && fused_outs(a.n_out, a.out, L.ffn_key->type, tas)) {
            // sequence mode: the mix writes its outputs (RWKV-7: its one output) as quantised tile images (their only consumers are the products below)
            a.out[0] = nullptr; a.out[1] = nullptr;
            launch_mix_seq_q(a, T, D, st, tas, L.ffn_key->type);
        } else {
            launch_mix(a, T, D, st);
            if (m.arch_major != 7) { const float * xs[2] = {b.m[0], b.m[1]}; prequant(2, xs, D, L.ffn_key->type); }
        }
        // Sequence mode, exact arm: the key product's only consumer is the value product, so its epilogue writes relu(k)^2 straight as that
        // product's quantised input image (prefill.hip MmqQOut) -- b.ffk is never written, its quantiser launch is gone. Needs the key
        // product's own input quantised ahead (its image then is not b.tile, which receives the output).
        TileAct fk; bool fkq = false;
        if (T >= k_mfma_min_tokens && b.tile && !ctx->prof.on && dtype_quantized(L.ffn_key->type) && dtype_quantized(L.ffn_value->type)) {
            // (the receptance product runs between the two: its input must be quantised ahead as well, or it would be quantised into b.tile)
            const bool rec_ok = m.arch_major == 7 || find_pre(b.m[1], L.ffn_receptance->cols(), L.ffn_receptance->type) != nullptr;
            if (const TileAct * in = rec_ok ? find_pre(b.m[0], L.ffn_key->cols(), L.ffn_key->type) : nullptr) {
                fk = tile_act_at(b.tile, T, L.ffn_key->rows());
                fkq = launch_mmq_mfma_q(*L.ffn_key, *in, T, epi(EPI_RELU_SQ), fk, L.ffn_value->type, st);
            }
        }
        if (!fkq) mm(L.ffn_key, b.m[0], b.ffk, epi(EPI_RELU_SQ));
        if (m.arch_major != 7) mm(L.ffn_receptance, b.m[1], b.r);
        drop_pre();
        if (fkq) { pre[0].x = b.ffk; pre[0].wtype = L.ffn_value->type; pre[0].K = L.ffn_value->cols(); pre[0].ta = fk; }
        if (m.arch_major == 7) mm(L.ffn_value, b.ffk, b.x, epi(EPI_ADD_RES, nullptr, b.x));
        else mm(L.ffn_value, b.ffk, b.x, epi(EPI_SIGMUL_ADD_RES, nullptr, b.x, b.r));
        drop_pre();
    }

    // rwkv_att_v4 (:163-197)
    void att_v4(const LayerW & L, const float * sin, float * sout) {
        launch_layernorm(b.x, T, D, f(L.ln1_w), f(L.ln1_b), b.xn, st);
        MixArgs a; a.xn = b.xn; a.carry_in = sin + D; a.carry_out = sout + D; a.mode = 0; a.n_out = 3;
        a.coef[0] = f(L.att_time_mix_k); a.coef[1] = f(L.att_time_mix_v); a.coef[2] = f(L.att_time_mix_r);
        a.out[0] = b.m[0]; a.out[1] = b.m[1]; a.out[2] = b.m[2];
        launch_mix(a, T, D, st);
        mm(L.att_receptance, b.m[2], b.r, epi(EPI_SIGMOID));
        mm(L.att_key, b.m[0], b.k);
        mm(L.att_value, b.m[1], b.v);
        launch_wkv4(b.k, b.v, b.r, f(L.att_time_first), f(L.att_time_decay), sin + 2 * D, sin + 3 * D, sin + 4 * D,
                    sout + 2 * D, sout + 3 * D, sout + 4 * D, b.out, T, D, st);
        mm(L.att_output, b.out, b.x, epi(EPI_ADD_RES, nullptr, b.x));
    }

    // rwkv_att_v5 (:199-292)
    void att_v5(const LayerW & L, const float * sin, float * sout) {
        const bool v52 = m.arch_minor >= 2;
        launch_layernorm(b.x, T, D, f(L.ln1_w), f(L.ln1_b), b.xn, st);
        MixArgs a; a.xn = b.xn; a.carry_in = sin + D; a.carry_out = sout + D; a.mode = 0; a.n_out = v52 ? 4 : 3;
        a.coef[0] = f(L.att_time_mix_k); a.coef[1] = f(L.att_time_mix_v); a.coef[2] = f(L.att_time_mix_r);
        if (v52) a.coef[3] = f(L.att_time_mix_g);
        for (int i = 0; i < 4; i++) a.out[i] = b.m[i];
        launch_mix(a, T, D, st);
        mm(L.att_receptance, b.m[2], b.r);
        mm(L.att_key, b.m[0], b.k);
        mm(L.att_value, b.m[1], b.v);
        if (v52) mm(L.att_gate, b.m[3], b.g, epi(EPI_SILU));
        wkv6(b.r, b.k, b.v, v52 ? f(L.att_time_faaaa) : f(L.att_time_first), v52 ? 1 : 0, f(L.att_time_decay), v52 ? 1 : 0,
             sin + 2 * D, sout + 2 * D, b.out);
        gn_out(L, 1e-5f, v52 ? b.g : nullptr);
    }

    // rwkv_att_v6 (:294-385)
    void att_v6(const LayerW & L, const float * sin, float * sout) {
        launch_layernorm(b.x, T, D, f(L.ln1_w), f(L.ln1_b), b.xn, st);
        MixArgs a; a.xn = b.xn; a.carry_in = sin + D; a.carry_out = sout + D; a.mode = 1; a.n_out = 1;
        a.coef[0] = f(L.att_time_maa_x); a.out[0] = b.m[5]; a.sx = b.sx;
        {
            // sequence mode: the mix's only consumer is the W1 product -- it writes that product's quantised input image (and sx, which the
            // five mixes read) instead of f32 values for a quantiser launch
            TileAct ta[1];
            if (fused_outs(1, a.out, L.att_time_maa_w1->type, ta) && getenv("RWKV_MI_NO_MIX_QUANT") == nullptr) {
                MixArgs q = a; q.out[0] = nullptr;
                if (!launch_mix_seq_q(q, T, D, st, ta, L.att_time_maa_w1->type)) { drop_pre(); launch_mix(a, T, D, st); }
            } else { drop_pre(); launch_mix(a, T, D, st); }
        }
        const int64_t R5 = L.att_time_maa_w1->ne[1], R = R5 / 5;
        mm(L.att_time_maa_w1, b.m[5], b.lr1, epi(EPI_TANH));
        drop_pre();
        V6Mix2Args v; v.w2 = f(L.att_time_maa_w2); v.tl = b.lr1; v.sx = b.sx; v.xn = b.xn;
        v.maa[0] = f(L.att_time_maa_w); v.maa[1] = f(L.att_time_maa_k); v.maa[2] = f(L.att_time_maa_v);
        v.maa[3] = f(L.att_time_maa_r); v.maa[4] = f(L.att_time_maa_g);
        for (int i = 0; i < 5; i++) v.out[i] = b.m[i];  // xw, xk, xv, xr, xg
        bool fused_q = false;
        {
            const int wt = L.att_receptance->type;
            TileAct tas[5];
            if (L.att_key->type == wt && L.att_value->type == wt && L.att_gate->type == wt && L.att_time_decay_w1->type == wt &&
                (R == 32 || R == 64) && fused_outs(5, v.out, wt, tas)) {
                // sequence mode: the five mixed inputs leave the kernel as quantised tile images (their only consumers are products)
                for (int i = 0; i < 5; i++) v.out[i] = nullptr;
                fused_q = launch_v6_mix2_seq(v, T, D, R, st, tas, wt);
                if (!fused_q) { drop_pre(); for (int i = 0; i < 5; i++) v.out[i] = b.m[i]; }
            }
        }
        if (!fused_q && !(T >= k_mfma_min_tokens && launch_v6_mix2_seq(v, T, D, R, st))) launch_v6_mix2(v, T, D, R, st);
        {
            // the five mixed inputs are quantised by one launch, the four D x D projections run as one launch (sequence mode)
            const float * xs[5] = {b.m[3], b.m[1], b.m[2], b.m[4], b.m[0]};
            if (!fused_q) prequant(5, xs, D, L.att_receptance->type);
            const DevTensor * Ws[4] = {L.att_receptance, L.att_key, L.att_value, L.att_gate};
            float * ys[4] = {b.r, b.k, b.v, b.g};
            const Epi es[4] = {Epi(), Epi(), Epi(), epi(EPI_SILU)};
            mm_batch(4, Ws, xs, ys, es);
        }
        mm(L.att_time_decay_w1, b.m[0], b.lr2, epi(EPI_TANH));
        drop_pre();
        // decay_w2 consumes [T][DR] rows of lr2
        mm(L.att_time_decay_w2, b.lr2, b.w, epi(EPI_V6_DECAY, f(L.att_time_decay)));
        wkv6(b.r, b.k, b.v, f(L.att_time_faaaa), 1, b.w, 2, sin + 2 * D, sout + 2 * D, b.out);
        gn_out(L, 64e-5f, b.g);
    }

    // rwkv_att_v7 (:387-482)
    void att_v7(const LayerW & L, int layer, const float * sin, float * sout) {
        launch_layernorm(b.x, T, D, f(L.ln1_w), f(L.ln1_b), b.xn, st);
        MixArgs a; a.xn = b.xn; a.carry_in = sin + D; a.carry_out = sout + D; a.mode = 1; a.n_out = 6;
        for (int i = 0; i < 6; i++) { a.coef[i] = f(L.att_x_rwkvag) + (int64_t) i * D; a.out[i] = b.m[i]; }  // r, w, k, v, a, g
        launch_mix(a, T, D, st);
        mm(L.att_receptance, b.m[0], b.r);
        mm(L.att_g1, b.m[5], b.lr1, epi(EPI_SIGMOID));
        mm(L.att_g2, b.lr1, b.g);
        mm(L.att_a1, b.m[4], b.lr1);
        mm(L.att_a2, b.lr1, b.a, epi(EPI_BIAS_SIGMOID, f(L.att_a0)));
        mm(L.att_w1, b.m[1], b.lr1, epi(EPI_TANH));
        mm(L.att_w2, b.lr1, b.w, epi(EPI_V7_DECAY, f(L.att_w0)));
        mm(L.att_key, b.m[2], b.k);
        launch_v7_kprep(b.k, b.a, f(L.att_k_k), f(L.att_k_a), b.t0 /* k' */, b.t1 /* -kk */, b.t2 /* kk*a */, T, H, S, st);
        mm(L.att_value, b.m[3], b.v);
        if (layer == 0) {
            launch_copy_f32(b.v_first, b.v, T * D, st);
        } else {
            mm(L.att_v1, b.m[3], b.lr1);
            mm(L.att_v2, b.lr1, b.sx, epi(EPI_BIAS_SIGMOID, f(L.att_v0)));
            launch_v7_vmix(b.v, b.v_first, b.sx, T * D, st);
        }
        static const bool no_seq7 = getenv("RWKV_MI_NO_WKV7_SEQ") != nullptr;   // (measurement aid: the single-token form over the whole sequence)
        if (S == 64 && T >= k_mfma_min_tokens && !no_seq7) launch_wkv7_seq(b.r, b.w, b.t0, b.v, b.t1, b.t2, sin + 2 * D, sout + 2 * D, b.out, T, H, st);
        else launch_wkv7(b.r, b.w, b.t0, b.v, b.t1, b.t2, sin + 2 * D, sout + 2 * D, b.out, T, H, S, st);
        launch_groupnorm(b.out, f(L.att_ln_x_w), f(L.att_ln_x_b), 64e-5f, b.g, b.t0, b.r, b.v, f(L.att_r_k), T, H, S, st);
        mm(L.att_output, b.out, b.x, epi(EPI_ADD_RES, nullptr, b.x));
    }

    void run_embed() {
        if (T == 1 && ctx->mega && m.has_embed && mega_v6_folds_embed(ctx->mega)) return;   // inside the persistent launch
        if (m.has_embed) launch_embed_ln0(*m.emb, ctx->d_tokens, T, D, f(m.ln0_w), f(m.ln0_b), b.x, st);
    }
    // layers [lb, le) of the stage (absolute layer ids); returns true when the launch also produced the logits (ring kernel, last layers)
    bool run_layers(uint32_t lb, uint32_t le, bool want_logits) {
        const float * sin = ctx->state[ctx->cur];
        float * sout = ctx->state[ctx->cur ^ 1];
        const int64_t per_layer = m.state_per_layer();
        if (T == 1 && ctx->mega) {
            const bool whole = lb == m.layer_begin && le == m.layer_end;
            const bool head_done = want_logits && m.has_head && le == m.layer_end && mega_v6_folds_head(ctx->mega);
            const float * s0 = sin + (int64_t) m.layer_begin * per_layer;
            float * o0 = sout + (int64_t) m.layer_begin * per_layer;
            // (persist_v47.hip: the launch starts from the token id and ends with the argmax where the stage has embedding / head)
            const uint32_t * tok = (lb == m.layer_begin && m.has_embed && mega_v6_folds_embed(ctx->mega)) ? ctx->d_tokens : nullptr;
            uint32_t * ntok = ctx->ntok_out ? ctx->ntok_out : (tok ? ctx->d_tokens : ctx->d_next_token);    // the greedy loops read the next token where the embedding reads it
            if (whole) mega_v6_forward(ctx->mega, b.x, s0, o0, st, &ctx->prof, head_done ? ctx->d_logits : nullptr, b.v_first, tok, ntok);
            else {
                // (a range's state pointers are those of ITS first layer for persist_v47.hip, of the stage's first layer for the ring kernel)
                const bool own_base = mega_v6_kind(ctx->mega) == 3;
                const int64_t so = own_base ? (int64_t) (lb - m.layer_begin) * per_layer : 0;
                mega_v6_forward_range(ctx->mega, b.x, s0 + so, o0 + so, st, nullptr, head_done ? ctx->d_logits : nullptr, (int) (lb - m.layer_begin), (int) (le - m.layer_begin), b.v_first, tok, ntok);
            }
            return head_done;
        }
        for (uint32_t i = lb; i < le; i++) {
            const LayerW & L = m.layers[i];
            const float * li = sin + (int64_t) i * per_layer;
            float * lo = sout + (int64_t) i * per_layer;
            if (T == 1 && ctx->fused_v6) { fused_v6_layer(m, L, b.x, li, lo, ctx->fused_scratch, st, &ctx->prof); continue; }
            if (T == 1 && ctx->fused_v4) { fused_v4_layer(m, L, b.x, li, lo, ctx->fused_scratch, st, &ctx->prof); continue; }
            if (T == 1 && ctx->fused_v7) { fused_v7_layer(m, L, (int) i, b.x, b.v_first, li, lo, ctx->fused_scratch, st, &ctx->prof); continue; }
            switch (m.arch_major) {
                case 4: att_v4(L, li, lo); break;
                case 5: att_v5(L, li, lo); break;
                case 6: att_v6(L, li, lo); break;
                case 7: att_v7(L, (int) i, li, lo); break;
                default: break;
            }
            ffn(L, li, lo);
        }
        return false;
    }
    void run_head() {
        // ln_out on the last token only, then the head projection (rwkv_graph.inc:704-708, 851-854)
        launch_layernorm(b.x + (T - 1) * D, 1, D, f(m.ln_out_w), f(m.ln_out_b), b.xlast, st);
        const int64_t Tsave = T; T = 1;
        mm(m.head, b.xlast, ctx->d_logits);
        T = Tsave;
    }
    void run(bool want_logits) {
        run_embed();
        const bool head_done = run_layers(m.layer_begin, m.layer_end, want_logits);
        if (want_logits && m.has_head && !head_done) run_head();
    }
};

}  // namespace

// Where a single-token step that produces logits has ALSO left their argmax (persist_v47.hip folds it into the launch), or nullptr: the
// greedy loops then need no argmax launch -- and no copy when that is where their next step reads the token.
uint32_t * folded_argmax_target(const rwkv_context * ctx) {
    if (!ctx->mega || !ctx->model->has_head || !mega_v6_folds_argmax(ctx->mega)) return nullptr;
    return (ctx->model->has_embed && mega_v6_folds_embed(ctx->mega)) ? ctx->d_tokens : ctx->d_next_token;
}

int64_t handoff_len(const Model & m) { return m.arch_major == 7 ? 2 * m.n_embed() : m.n_embed(); }

bool forward(rwkv_context * ctx, int64_t T, bool want_logits) {
    if (!ensure_scratch(ctx, T)) return false;
    Model & m = *ctx->model;
    Runner r{ctx, m, ctx->stream, T, m.n_embed(), m.head_count, m.head_size, ctx->b};
    const bool chained = T == 1 && ctx->mega;
    if (chained) mega_chain_begin(ctx);
    r.run(want_logits);
    if (chained) mega_chain_end(ctx);
    ctx->cur ^= 1;
    HIP_CTX_OK(ctx, hipGetLastError());
    RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH | RWKV_ERROR_ALLOC, false, !r.failed, "a sequence-mode product could not be launched (out of device memory for the tile-major weight image?)");
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// rwkv_eval with the caller's state STREAMED (rwkv_eval.inc:2-22,38-76; rwkv.h:106-108: state_in / state_out are host buffers on
// every call). The plain form is upload (34.6 MB for the 7B) -> token -> download, serial: 2.97 ms per token against 1.54 ms with the
// state resident (round 3: 337 vs 649 tokens/s). Here the token is cut into up to eight layer groups:
//     this thread      for each group: upload its state slice (copy stream), record; the compute stream waits for THAT slice only and
//                      runs the group's layers (the ring kernel on a layer range: one launch per group; the per-layer paths: their
//                      launches), records "group done"
//     download thread  for each group: a second copy stream waits for "group done" and brings the group's new state slice back
// so the PCIe traffic of both directions runs under the layers of other groups; exposed are the first slice's upload, the last
// slice's download and the logits. Nothing is assumed about the caller's memory (pageable copies through the runtime's staging path,
// both directions at once from two threads); state_in == state_out is fine (a slice is read before its group runs, written after).
// The state on the device is complete afterwards (state[cur]): a poll time-out of the persistent kernel falls back as before.
// ---------------------------------------------------------------------------------------------------------------
constexpr int k_abi_max_groups = 32;
struct AbiStreamer {
    hipStream_t up = nullptr, down = nullptr;
    std::vector<hipEvent_t> ev_up, ev_done;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    // job of the download thread
    int n_groups = 0, published = 0, finished = 0;     // groups of this call / "group done" recorded so far / downloaded so far
    bool quit = false, failed = false;
    int device = 0;
    float * h_out = nullptr; const float * d_out = nullptr;
    std::vector<std::pair<int64_t, int64_t>> slices;    // (offset, count) in floats per group
    uint64_t call = 0, seen = 0;

    void run() {
        (void) hipSetDevice(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || call != seen; });
            if (quit) return;
            seen = call;
            int g = 0;
            while (g < n_groups) {
                cv.wait(lk, [&] { return quit || published > g || g >= n_groups; });
                if (quit) return;
                if (g >= n_groups) break;       // (the call failed before this group ran)
                const auto sl = slices[(size_t) g];
                hipEvent_t ev = ev_done[(size_t) g];
                lk.unlock();
                bool ok = hipStreamWaitEvent(down, ev, 0) == hipSuccess &&
                          hipMemcpyAsync(h_out + sl.first, d_out + sl.first, (size_t) sl.second * sizeof(float), hipMemcpyDeviceToHost, down) == hipSuccess &&
                          hipStreamSynchronize(down) == hipSuccess;
                lk.lock();
                if (!ok) failed = true;
                g++;
                finished = g;
                cv.notify_all();
            }
        }
    }
};

void abi_streamer_free(void * p) {
    AbiStreamer * a = (AbiStreamer *) p;
    if (!a) return;
    { std::lock_guard<std::mutex> lk(a->mu); a->quit = true; }
    a->cv.notify_all();
    if (a->worker.joinable()) a->worker.join();
    for (hipEvent_t e : a->ev_up) (void) hipEventDestroy(e);
    for (hipEvent_t e : a->ev_done) (void) hipEventDestroy(e);
    if (a->up) (void) hipStreamDestroy(a->up);
    if (a->down) (void) hipStreamDestroy(a->down);
    delete a;
}

bool forward_streamed_eligible(const rwkv_context * ctx) {
    const char * e = getenv("RWKV_MI_ABI_STREAM");       // 0: off, 1: whatever the size of the state (tests), unset: from 4 MB of state
    if ((e && e[0] == '0') || !ctx->stages.empty() || !ctx->owns_stream) return false;
    const Model & m = *ctx->model;
    if (m.layer_end - m.layer_begin < 2 || !m.has_embed || !m.has_head) return false;
    if (ctx->mega && !mega_v6_has_range(ctx->mega)) return false;     // (the register-prefetch kernel has no layer-range launch)
    if (e && e[0] == '1') return true;
    return (size_t) m.state_len() * sizeof(float) >= ((size_t) 4 << 20);   // small states: the serial copies are already cheap
}

// One token (already in ctx->d_tokens) from the caller's state_in (nullptr: the fresh state) into the caller's state_out (nullptr: none).
// On return the streams are drained; *aborted reports a poll time-out of the persistent kernel (the caller repeats the step).
bool forward_streamed(rwkv_context * ctx, bool want_logits, const float * h_in, float * h_out, float * h_logits, bool * aborted) {
    *aborted = false;
    Model & m = *ctx->model;
    if (!ensure_scratch(ctx, 1)) return false;
    AbiStreamer * a = (AbiStreamer *) ctx->abi_streamer;
    if (!a) {
        a = new (std::nothrow) AbiStreamer();
        RW_CTX_CHECK(ctx, RWKV_ERROR_ALLOC, false, a != nullptr, "out of memory");
        a->device = m.device;
        bool ok = hipStreamCreateWithFlags(&a->up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&a->down, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < k_abi_max_groups && ok; i++) {
            hipEvent_t e1 = nullptr, e2 = nullptr;
            ok = hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess;
            if (e1) a->ev_up.push_back(e1);
            if (e2) a->ev_done.push_back(e2);
        }
        if (!ok) { abi_streamer_free(a); RW_CTX_CHECK(ctx, RWKV_ERROR_ALLOC, false, false, "cannot create the copy streams of the streamed rwkv_eval"); }
        a->worker = std::thread([a] { a->run(); });
        ctx->abi_streamer = a;
    }
    const uint32_t L = m.layer_end - m.layer_begin;
    // Layer groups. What a call cannot hide is the upload of its FIRST slice and the download of its LAST one, so the groups are small at both
    // ends and grow towards the middle: 1 1 2 4 8 | 8 4 2 1 1 layers for 32 (round 4 cut eight even groups: 4.3 MB exposed at each end of a 7B
    // token, now 1.1 MB; two more launches). RWKV_MI_ABI_GROUPS=even restores the even cut (A/B).
    std::vector<uint32_t> gsz;
    {
        static const bool even = [] { const char * e = getenv("RWKV_MI_ABI_GROUPS"); return e && e[0] == 'e'; }();
        const char * lst = getenv("RWKV_MI_ABI_GROUPS");     // (measurement aid: an explicit list of group sizes, e.g. 1,3,12,12,3,1)
        if (lst && lst[0] >= '0' && lst[0] <= '9') {
            uint32_t sum = 0;
            for (const char * q = lst; *q && (int) gsz.size() < k_abi_max_groups;) { const uint32_t v = (uint32_t) strtoul(q, (char **) &q, 10); if (v == 0) break; gsz.push_back(v); sum += v; if (*q == ',') q++; }
            if (sum != L) gsz.clear();
        }
        if (!gsz.empty()) {}
        else if (even || L < 8) { const uint32_t G0 = L < 8 ? L : 8; for (uint32_t g = 0; g < G0; g++) gsz.push_back((uint32_t) ((uint64_t) L * (g + 1) / G0 - (uint64_t) L * g / G0)); }
        else {
            std::vector<uint32_t> left, right;
            uint32_t rest = L;
            for (uint32_t i = 0, sz = 1; rest > 0; i++) {
                const uint32_t a = sz < rest ? sz : rest; left.push_back(a); rest -= a;
                if (rest > 0) { const uint32_t b = sz < rest ? sz : rest; right.push_back(b); rest -= b; }
                if (i >= 1 && sz < 8) sz *= 2;      // 1 1 2 4 8 8 ...
                if ((int) (left.size() + right.size()) >= k_abi_max_groups - 2 && rest > 0) { left.back() += rest; rest = 0; }
            }
            gsz = left;
            for (size_t i = right.size(); i-- > 0;) gsz.push_back(right[i]);
        }
    }
    const int G = (int) gsz.size();
    const int64_t per = m.state_per_layer();
    float * sin = ctx->state[ctx->cur];
    float * sout = ctx->state[ctx->cur ^ 1];
    // every fallible step of the set-up comes BEFORE the download job is published: a return between arming the worker and the final wait
    // would leave it inside a stale job holding the caller's pointer (it then finished the NEXT call's groups twice, the second time after
    // that call had returned)
    if (!h_in) { if (!state_from_host(ctx, nullptr)) return false; }
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    {
        std::lock_guard<std::mutex> lk(a->mu);
        a->slices.clear();
        uint32_t lb = m.layer_begin;
        for (int g = 0; g < G; g++) {
            const uint32_t le = lb + gsz[(size_t) g];
            ranges.push_back({lb, le});
            a->slices.push_back({(int64_t) lb * per, (int64_t) (le - lb) * per});
            lb = le;
        }
        // (the LAST group's slice is brought back by this thread on the compute stream, right behind its layers and in front of the logits: handing
        //  it to the download thread cost a wake-up of that thread and one of this one on the only part of the call nothing overlaps)
        a->n_groups = h_out ? G - 1 : 0; a->published = 0; a->finished = 0; a->failed = false;
        a->h_out = h_out; a->d_out = sout;
        a->call++;
    }
    a->cv.notify_all();
    Runner r{ctx, m, ctx->stream, 1, m.n_embed(), m.head_count, m.head_size, ctx->b};
    r.run_embed();
    const bool chained = ctx->mega != nullptr;
    if (chained) mega_chain_begin(ctx);
    bool ok = true, head_done = false;
    for (int g = 0; g < G && ok; g++) {
        const auto sl = a->slices[(size_t) g];
        if (h_in) {
            ok = hipMemcpyAsync(sin + sl.first, h_in + sl.first, (size_t) sl.second * sizeof(float), hipMemcpyHostToDevice, a->up) == hipSuccess &&
                 hipEventRecord(a->ev_up[(size_t) g], a->up) == hipSuccess && hipStreamWaitEvent(ctx->stream, a->ev_up[(size_t) g], 0) == hipSuccess;
            if (!ok) break;
        }
        head_done = r.run_layers(ranges[(size_t) g].first, ranges[(size_t) g].second, want_logits) || head_done;
        ok = hipEventRecord(a->ev_done[(size_t) g], ctx->stream) == hipSuccess;
        if (ok && h_out && g + 1 < G) { { std::lock_guard<std::mutex> lk(a->mu); a->published = g + 1; } a->cv.notify_all(); }
    }
    if (ok && h_out) {
        const auto sl = a->slices[(size_t) (G - 1)];
        ok = hipMemcpyAsync(h_out + sl.first, sout + sl.first, (size_t) sl.second * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    }
    if (chained) mega_chain_end(ctx);
    if (ok && want_logits && !head_done) r.run_head();
    if (ok && h_logits) ok = hipMemcpyAsync(h_logits, ctx->d_logits, (size_t) m.n_vocab() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    const bool ctl_ok = !ctx->mega || mega_v6_ctl_fetch(ctx->mega, ctx->stream);
    const bool sync_ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
    {   // the download thread must be done with this call whatever happened above (it holds the caller's pointer)
        std::unique_lock<std::mutex> lk(a->mu);
        if (!ok) { a->n_groups = a->published; }     // (groups that never ran are not downloaded)
        a->cv.notify_all();
        a->cv.wait(lk, [&] { return a->finished >= a->n_groups; });
        ok = ok && !a->failed;
    }
    ctx->cur ^= 1;
    RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH, false, ok && sync_ok, "HIP error in the streamed rwkv_eval: %s", hipGetErrorString(hipGetLastError()));
    RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH | RWKV_ERROR_ALLOC, false, !r.failed, "a product could not be launched");
    if (ctx->mega && (!ctl_ok || mega_v6_aborted_cached(ctx->mega))) { recover_from_abort(ctx); *aborted = true; }
    return true;
}

// A single-token step that is ONE persistent launch issued directly (no graph): no embedding launch in front of it (none to do, or folded)
// and no head launches behind it (none wanted, or folded). A whole model additionally folds the argmax (the greedy loops' condition since
// round 5); a pipeline stage qualifies since round 6 -- a chain of N stages paid N replays of a one-node graph per token
// (RWKV_MI_STAGE_GRAPH=1 restores that for A/B). runner.cpp relies on this: only a direct launch sees a changed mega_v6_set_x_out.
bool single_launch_step(const rwkv_context * ctx, bool want_logits) {
    if (!ctx->mega) return false;
    const Model & mm = *ctx->model;
    static const bool replay = getenv("RWKV_MI_STAGE_GRAPH") != nullptr;
    const bool one_launch = (!mm.has_embed || mega_v6_folds_embed(ctx->mega)) && (!(want_logits && mm.has_head) || mega_v6_folds_head(ctx->mega));
    if (!one_launch) return false;
    return (mm.has_embed && mm.has_head) ? mega_v6_folds_argmax(ctx->mega) : !replay;
}

// Single-token step through a captured hipGraph: one graph per (state parity, logits on/off), replayed per token so
// that the ~100 short launches of a decode step cost one graph launch on the host.
bool forward_decode(rwkv_context * ctx, bool want_logits) {
    if (!ctx->use_graph) return forward(ctx, 1, want_logits);
    // one launch per token (persist_v47.hip with embedding, head and argmax inside it): a direct launch from a busy stream costs the host
    // 3 - 5 us, the replay of a one-node graph 10 - 16 (guide row graph-replay-floor) -- at 250 us per token of the 169M that is the gap
    // Round 6: the same holds for every stage of a layer pipeline whose step is ONE persistent launch (single_launch_step below).
    if (single_launch_step(ctx, want_logits)) return forward(ctx, 1, want_logits);
    if (!ensure_scratch(ctx, 1)) return false;
    hipGraphExec_t & ge = ctx->graph_exec[ctx->cur][want_logits ? 1 : 0];
    if (!ge) {
        Model & m = *ctx->model;
        hipGraph_t graph = nullptr;
        HIP_CTX_OK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        Runner r{ctx, m, ctx->stream, 1, m.n_embed(), m.head_count, m.head_size, ctx->b};
        r.run(want_logits);
        HIP_CTX_OK(ctx, hipStreamEndCapture(ctx->stream, &graph));
        hipError_t e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
        (void) hipGraphDestroy(graph);
        RW_CTX_CHECK(ctx, RWKV_ERROR_GRAPH, false, e == hipSuccess, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    const bool chained = ctx->mega != nullptr;
    if (chained) mega_chain_begin(ctx);
    const hipError_t le = hipGraphLaunch(ge, ctx->stream);
    if (chained) mega_chain_end(ctx);
    HIP_CTX_OK(ctx, le);
    ctx->cur ^= 1;
    return true;
}

}  // namespace rwkvmi
