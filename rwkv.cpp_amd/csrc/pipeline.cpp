// pipeline.cpp -- the layer pipeline behind the UNMODIFIED rwkv.h ABI, inside one process:
//
//     RWKV_MI_DEVICES=0,1,2,3  (or 0-3)      rwkv_init_from_file() builds one stage per listed device
//
// The reference exposes device placement through rwkv_init_from_file(..., n_gpu_layers) (a CPU / one-GPU layer split,
// rwkv_model_loading.inc:129-142); here the layers are cut into contiguous ranges balanced by streamed bytes (the head counts on the
// last stage), every stage owns its weights and its slice of the recurrent state in its device's HBM, and a token walks the chain:
// stage s runs its layers on its own stream, its residual stream (plus v_first for RWKV-7) is copied device-to-device
// (hipMemcpyPeerAsync: xGMI between GPUs) and stage s + 1's stream waits on an event -- no host round trip inside a token, the host
// only enqueues. This is the same partitioning as the one-process-per-GPU pipeline over RCCL send/recv (rwkv.cpp_amd/pipeline.py,
// the north-star's multi-node-capable form); this file is the form a plain C caller of rwkv.h can reach.
// A device may be listed more than once (stages sharing a GPU: how the tests run the chain on a one-GPU box).
#include "model.h"

#include <cstdlib>
#include <cstring>
#include <string>

namespace rwkvmi {

#define HIP_FRONT_OK(CTX, CALL) \
    do { hipError_t e_ = (CALL); RW_CTX_CHECK((CTX), RWKV_ERROR_GRAPH, false, e_ == hipSuccess, "HIP error: %s", hipGetErrorString(e_)); } while (0)

static bool parse_devices(const char * spec, std::vector<int> & out) {
    out.clear();
    const char * p = spec;
    while (*p) {
        char * end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') { p++; b = strtol(p, &end, 10); if (end == p || b < a) return false; p = end; }
        for (long d = a; d <= b; d++) out.push_back((int) d);
        if (*p == ',') p++;
        else if (*p) return false;
    }
    return !out.empty();
}

// contiguous ranges with (roughly) equal streamed bytes; the head's bytes count on the last stage
static void partition(const std::vector<uint64_t> & per_layer, uint64_t head, size_t n_stages, std::vector<std::pair<uint32_t, uint32_t>> & ranges) {
    const size_t L = per_layer.size();
    if (n_stages > L) n_stages = L;
    uint64_t total = head;
    for (uint64_t v : per_layer) total += v;
    ranges.clear();
    uint32_t begin = 0;
    uint64_t acc = 0;
    for (size_t s = 0; s < n_stages; s++) {
        const uint64_t target = total * (s + 1) / n_stages;
        uint32_t end = begin;
        const uint32_t must_leave = (uint32_t) (n_stages - 1 - s);   // one layer at least for every later stage
        while (end < L - must_leave && (end == begin || acc + per_layer[end] / 2 <= target)) { acc += per_layer[end]; end++; }
        if (s + 1 == n_stages) { while (end < L) { acc += per_layer[end]; end++; } }
        ranges.emplace_back(begin, end);
        begin = end;
    }
}

void pipeline_destroy(rwkv_context * front) {
    if (!front) return;
    for (rwkv_context * s : front->stages) {
        if (!s) continue;
        (void) hipSetDevice(s->model->device);
        if (s->handoff_ev) { (void) hipEventDestroy(s->handoff_ev); s->handoff_ev = nullptr; }
        if (s->consumed_ev) { (void) hipEventDestroy(s->consumed_ev); s->consumed_ev = nullptr; }
        destroy_context(s);
    }
    front->stages.clear();
    delete front;
}

static rwkv_context * make_front() {
    rwkv_context * f = new (std::nothrow) rwkv_context();
    return f;
}

static bool finish_stage(rwkv_context * c) {
    return hipEventCreateWithFlags(&c->handoff_ev, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&c->consumed_ev, hipEventDisableTiming) == hipSuccess;
}

rwkv_context * pipeline_create(const char * path, uint32_t n_threads, const char * devices) {
    std::vector<int> devs;
    RW_CHECK(RWKV_ERROR_ARGS, nullptr, parse_devices(devices, devs), "RWKV_MI_DEVICES=\"%s\" is not a device list (e.g. 0,1,2,3 or 0-3)", devices);
    int n_dev = 0;
    RW_CHECK(RWKV_ERROR_CTX | RWKV_ERROR_UNSUPPORTED, nullptr, hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0, "No HIP device is visible");
    for (int d : devs) RW_CHECK(RWKV_ERROR_ARGS, nullptr, d < n_dev, "RWKV_MI_DEVICES names device %d, but only %d are visible", d, n_dev);
    std::vector<uint64_t> per_layer;
    uint64_t head = 0;
    if (!scan_stage_costs(path, per_layer, head)) return nullptr;
    std::vector<std::pair<uint32_t, uint32_t>> ranges;
    partition(per_layer, head, devs.size(), ranges);
    rwkv_context * front = make_front();
    RW_CHECK(RWKV_ERROR_CTX | RWKV_ERROR_ALLOC, nullptr, front != nullptr, "Failed to allocate rwkv_context");
    int prev_dev = 0;
    (void) hipGetDevice(&prev_dev);
    bool multi_device = false;
    for (size_t s = 1; s < devs.size(); s++) multi_device = multi_device || devs[s] != devs[0];
    for (size_t s = 0; s < ranges.size(); s++) {
        if (hipSetDevice(devs[s]) != hipSuccess) { pipeline_destroy(front); global_fail(RWKV_ERROR_CTX, __FILE__, __LINE__, "hipSetDevice", "cannot select device %d", devs[s]); return nullptr; }
        Model * m = load_model(path, ranges[s].first, ranges[s].second);
        rwkv_context * c = m ? create_context(m, n_threads) : nullptr;
        if (!c || !finish_stage(c)) { if (c) front->stages.push_back(c); pipeline_destroy(front); (void) hipSetDevice(prev_dev); return nullptr; }
        front->stages.push_back(c);
        if (multi_device) prefill_prepare_current_device();   // (kernel attributes are per device; the launchers set them once per process)
        // direct peer copies where the topology allows them (otherwise the runtime stages the copy)
        if (s > 0 && devs[s] != devs[s - 1]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[s], devs[s - 1]) == hipSuccess && can) (void) hipDeviceEnablePeerAccess(devs[s - 1], 0);
            (void) hipSetDevice(devs[s - 1]);
            if (hipDeviceCanAccessPeer(&can, devs[s - 1], devs[s]) == hipSuccess && can) (void) hipDeviceEnablePeerAccess(devs[s], 0);
            (void) hipGetLastError();   // "already enabled" is not an error
        }
    }
    (void) hipSetDevice(prev_dev);
    front->model = front->stages[0]->model;   // header / architecture queries (n_vocab, n_embed, n_layer, state length) are global
    front->n_threads = n_threads;
    return front;
}

rwkv_context * pipeline_clone(rwkv_context * front, uint32_t n_threads) {
    rwkv_context * c = make_front();
    RW_CHECK(RWKV_ERROR_CTX | RWKV_ERROR_ALLOC, nullptr, c != nullptr, "Failed to allocate rwkv_context");
    for (rwkv_context * s : front->stages) {
        (void) hipSetDevice(s->model->device);
        rwkv_context * n = create_context(s->model, n_threads);
        if (!n || !finish_stage(n)) { if (n) c->stages.push_back(n); pipeline_destroy(c); return nullptr; }
        c->stages.push_back(n);
    }
    c->model = c->stages[0]->model;
    c->n_threads = n_threads;
    c->print_errors = front->print_errors;
    return c;
}

// state slices: a stage owns layers [layer_begin, layer_end) of the caller's state vector
static bool stage_state_in(rwkv_context * c, const float * state_in) {
    Model & m = *c->model;
    const int64_t per = m.state_per_layer();
    const int64_t off = (int64_t) m.layer_begin * per, cnt = (int64_t) (m.layer_end - m.layer_begin) * per;
    float * dst = c->state[c->cur] + off;
    if (state_in) {
        HIP_FRONT_OK(c, hipMemcpyAsync(dst, state_in + off, (size_t) cnt * sizeof(float), hipMemcpyHostToDevice, c->stream));
        return true;
    }
    return state_from_host(c, nullptr);   // fresh state (whole buffer: cheap, and the v4 pattern needs the layer layout)
}
static bool stage_state_out(rwkv_context * c, float * state_out) {
    Model & m = *c->model;
    const int64_t per = m.state_per_layer();
    const int64_t off = (int64_t) m.layer_begin * per, cnt = (int64_t) (m.layer_end - m.layer_begin) * per;
    HIP_FRONT_OK(c, hipMemcpyAsync(state_out + off, c->state[c->cur] + off, (size_t) cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return true;
}

// tokens[0 .. n) in passes of at most `chunk` tokens; logits of the last token; everything enqueued, one synchronisation at the end
bool pipeline_eval(rwkv_context * front, const uint32_t * tokens, size_t n, size_t chunk, const float * state_in, float * state_out, float * logits_out) {
    auto & st = front->stages;
    const size_t S = st.size();
    auto fail = [&](rwkv_context * c) { front->last_error |= c->last_error ? c->last_error : (int) RWKV_ERROR_GRAPH; return false; };
    for (rwkv_context * c : st) {
        c->last_error = 0; c->print_errors = front->print_errors;
        if (hipSetDevice(c->model->device) != hipSuccess || !stage_state_in(c, state_in)) return fail(c);
    }
    if (chunk == 0 || chunk > 1024) chunk = 1024;
    size_t done = 0;
    while (done < n) {
        const size_t T = (n - done) < chunk ? (n - done) : chunk;
        const bool last = done + T == n;
        for (size_t s = 0; s < S; s++) {
            rwkv_context * c = st[s];
            Model & m = *c->model;
            if (hipSetDevice(m.device) != hipSuccess || !ensure_scratch(c, (int64_t) T)) return fail(c);
        }
        for (size_t s = 0; s < S; s++) {
            rwkv_context * c = st[s];
            Model & m = *c->model;
            const size_t D = (size_t) m.n_embed();
            if (hipSetDevice(m.device) != hipSuccess) return fail(c);
            if (s == 0) {
                if (!upload_tokens_for(c, tokens + done, T)) return fail(c);
            } else {
                rwkv_context * p = st[s - 1];
                // the previous stage's residual stream (its scratch x, [T][D]) -> this stage's, device to device, on the PRODUCER's stream
                // (behind its layers); this stage's stream then waits on the event
                if (hipSetDevice(p->model->device) != hipSuccess) return fail(p);
                // (not before this stage has finished the previous pass: its x is both the input and the running residual stream)
                if (done > 0 && hipStreamWaitEvent(p->stream, c->consumed_ev, 0) != hipSuccess) return fail(p);
                if (hipMemcpyPeerAsync(c->b.x, m.device, p->b.x, p->model->device, T * D * sizeof(float), p->stream) != hipSuccess) return fail(c);
                if (m.arch_major == 7 && hipMemcpyPeerAsync(c->b.v_first, m.device, p->b.v_first, p->model->device, T * D * sizeof(float), p->stream) != hipSuccess) return fail(c);
                if (hipEventRecord(p->handoff_ev, p->stream) != hipSuccess) return fail(p);
                // A MIDDLE stage is done with its x (input, running residual stream AND source of the copy above) only now: behind its
                // outgoing copy, not behind its layers. (Recorded after forward(), the stage before it could overwrite p's x in the next
                // pass while this copy -- itself held back by the wait on c->consumed_ev -- was still pending: the next stage then
                // received rows of the next chunk. Seen as a rare mismatch of eval_sequence_in_chunks on a three-stage chain.)
                if (s - 1 > 0 && !last && hipEventRecord(p->consumed_ev, p->stream) != hipSuccess) return fail(p);
                if (hipSetDevice(m.device) != hipSuccess || hipStreamWaitEvent(c->stream, p->handoff_ev, 0) != hipSuccess) return fail(c);
            }
            const bool want = last && m.has_head && logits_out != nullptr;
            const bool ok = T == 1 ? forward_decode(c, want) : forward(c, (int64_t) T, want);
            if (!ok) return fail(c);
            // the LAST stage has no outgoing copy: done with its x behind its layers
            if (s > 0 && s + 1 == S && !last && hipEventRecord(c->consumed_ev, c->stream) != hipSuccess) return fail(c);
        }
        done += T;
    }
    for (rwkv_context * c : st) {
        if (hipSetDevice(c->model->device) != hipSuccess) return fail(c);
        if (state_out && !stage_state_out(c, state_out)) return fail(c);
    }
    rwkv_context * tail = st[S - 1];
    if (logits_out && hipMemcpyAsync(logits_out, tail->d_logits, (size_t) tail->model->n_vocab() * sizeof(float), hipMemcpyDeviceToHost, tail->stream) != hipSuccess) return fail(tail);
    for (rwkv_context * c : st) {
        if (hipSetDevice(c->model->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return fail(c);
        if (c->mega && mega_v6_aborted(c->mega, c->stream)) { recover_from_abort(c); front->last_error |= RWKV_ERROR_GRAPH; return false; }
    }
    return true;
}

}  // namespace rwkvmi
