// runner.cpp -- the greedy decode loop of a layer pipeline, enqueued from C++ (no interpreter between tokens):
//
//   * inside one process over the stage chain of RWKV_MI_DEVICES (pipeline.cpp): rwkv_mi_decode_greedy_streams()
//   * one process per GPU over RCCL send / recv on the stage's stream:              rwkv_mi_stage_run()
//
// Both run the SAME per-(token, decode stream, stage) iteration below and differ only in the Hop that carries a message from one
// stage to the next (the residual stream forward, the chosen token back from the last stage to the first):
//
//   LocalHop   a mailbox in the receiver's HBM, hipMemcpyPeerAsync on the sender's stream (xGMI between GPUs), two events
//   RcclHop    ncclSend / ncclRecv of librccl.so, which is dlopen'ed on first use -- librwkv.so keeps its one-GPU dependency list
//
// S decode streams (clones: shared weights, own recurrent state) are interleaved, so that with S >= number of stages every stage
// has work while the others run theirs (rwkv.cpp_amd/pipeline.py states the protocol; this is its loop without Python per token).
// The reference has no counterpart: its only device split is the CPU / one-GPU layer split of rwkv_model_loading.inc:129-142.
#include "model.h"
#include "rwkv_mi355x.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

using namespace rwkvmi;

namespace {

#define RUN_OK(CTX, CALL) \
    do { hipError_t e_ = (CALL); RW_CTX_CHECK((CTX), RWKV_ERROR_GRAPH, false, e_ == hipSuccess, "HIP error: %s", hipGetErrorString(e_)); } while (0)

// ---------------------------------------------------------------------------------------------------------------
// hops
// ---------------------------------------------------------------------------------------------------------------
struct Hop {
    virtual ~Hop() {}
    // message `msg` (0: the residual stream / the token, 1: RWKV-7's v_first) of decode stream j; both calls only enqueue. The sender's
    // call for a message is made before the receiver's.
    virtual bool send(int j, int msg, const void * src, size_t bytes, hipStream_t st) = 0;   // current device = the sender's
    virtual bool recv(int j, int msg, void * dst, size_t bytes, hipStream_t st) = 0;         // current device = the receiver's
    // A hop that can write straight into the receiver's buffer is told where message (j, msg) is consumed; recv() into that address is then
    // a wait only, and the receiver calls release() once everything that reads (or overwrites and forwards) the buffer is enqueued.
    virtual bool bind(int, int, void *) { return false; }
    virtual bool release(int, int, hipStream_t) { return true; }
    // A bound message whose buffer the SENDER's kernel can write itself (same device, or the peer mapping is enabled): the address, else
    // nullptr. The sender then brackets the launch that writes it with produce_begin() (the receiver is done with the previous content)
    // and produce_end() (the message is there) instead of calling send().
    virtual void * direct_target(int, int) { return nullptr; }
    virtual bool produce_begin(int, int, hipStream_t) { return false; }
    // done: an event already recorded behind the producing launch on the sender's stream (the per-device chain's marker), or nullptr -- the
    // hop then records its own
    virtual bool produce_end(int, int, hipStream_t, hipEvent_t /* done */) { return false; }
    // the event the last recv() of (j, msg) made the stream wait on (bound messages; else nullptr)
    virtual hipEvent_t waited_on(int, int) { return nullptr; }
    // The hop is an edge of a CLOSED loop (greedy decode: the token comes back from the last stage): every buffer's reuse is then ordered by
    // the loop itself -- stage s can only start token t + 1 behind the last stage's token t, which is behind stage s + 1's token t -- and the
    // `taken` events of bound messages are neither recorded nor waited for.
    virtual void set_closed_loop(bool) {}
};
constexpr int k_hop_msgs = 2;

struct LocalHop : Hop {
    int src_dev, dst_dev;
    struct Slot { void * box = nullptr; void * direct = nullptr; hipEvent_t ready = nullptr, taken = nullptr, ready_now = nullptr; bool used = false, released = false; };
    // One mailbox and one event pair PER MESSAGE of a stream's iteration: a stage of an RWKV-7 chain sends x and then v_first. Through
    // one box the second send only waited for the previous ITERATION's `taken` and overwrote x before the receiver had run -- both
    // receives then read v_first (round-3 review; tests/test_gpu_pipeline_cpp.py::test_rwkv7_greedy_loop_through_a_chain).
    //
    // Round 6: a BOUND slot has no mailbox in the path. The peer copy lands where the receiving stage reads the message (its residual
    // stream, its token word) -- one copy per hop instead of two -- and `taken` is recorded by release() at the end of the receiver's
    // iteration: that buffer is input, running residual stream and source of the receiver's own outgoing copy, so it is free only behind
    // all three (pipeline.cpp's consumed_ev, same reason). A send waits for the latest release, also the first send of the token
    // feedback: the first stage's token word holds the seed token until its first iteration has read it.
    // Round 6, second step: where the sender's kernel can reach the receiver's buffer (one device, or peer access from the sender's
    // device), the stage's last layer stores the residual stream THERE (mega_v6_set_x_out) and the hop is two event operations: no copy
    // kernel between the stages (11 us from the end of the launch to the start of the copy, 4.7 us of copy, 17 us to the start of the next
    // stage's launch in profiles/r06_hop_trace.txt).
    std::vector<Slot> slots;   // [stream][message]
    size_t cap;
    bool peer_ok = false;      // kernels on src_dev may store to memory of dst_dev
    bool closed = false;       // set_closed_loop
    void set_closed_loop(bool on) override { closed = on && getenv("RWKV_MI_HOP_TAKEN") == nullptr; }   // (RWKV_MI_HOP_TAKEN=1: A/B, tests; read per call)
    LocalHop(int sdev, int ddev, int n_streams, size_t bytes, bool & ok) : src_dev(sdev), dst_dev(ddev), slots((size_t) n_streams * k_hop_msgs), cap(bytes) {
        peer_ok = sdev == ddev;
        if (!peer_ok && hipSetDevice(sdev) == hipSuccess) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, sdev, ddev) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(ddev, 0);
                peer_ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void) hipGetLastError();
        }
        ok = hipSetDevice(ddev) == hipSuccess;
        for (Slot & s : slots) {
            ok = ok && hipMalloc(&s.box, bytes) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.taken, hipEventDisableTiming) == hipSuccess;
        }
    }
    ~LocalHop() override {
        (void) hipSetDevice(dst_dev);
        for (Slot & s : slots) {
            if (s.box) (void) hipFree(s.box);
            if (s.ready) (void) hipEventDestroy(s.ready);
            if (s.taken) (void) hipEventDestroy(s.taken);
        }
    }
    Slot * slot(int j, int msg) { return (msg < 0 || msg >= k_hop_msgs || j < 0 || (size_t) j * k_hop_msgs >= slots.size()) ? nullptr : &slots[(size_t) j * k_hop_msgs + (size_t) msg]; }
    bool bind(int j, int msg, void * dst) override {
        const char * e = getenv("RWKV_MI_HOP");      // RWKV_MI_HOP=mailbox: round 5's two-copy hop (A/B, tests; read per call)
        const bool mailbox = e && e[0] == 'm';
        Slot * s = slot(j, msg);
        if (!s || mailbox || !dst) return false;
        s->direct = dst;
        return true;
    }
    bool send(int j, int msg, const void * src, size_t bytes, hipStream_t st) override {
        Slot * s = slot(j, msg);
        if (!s || bytes > cap) return false;
        if (s->direct) {
            if (!closed && s->released && hipStreamWaitEvent(st, s->taken, 0) != hipSuccess) return false;   // the receiver is done with what the buffer held
            if (hipMemcpyPeerAsync(s->direct, dst_dev, src, src_dev, bytes, st) != hipSuccess) return false;
            s->used = true; s->ready_now = s->ready;
            return hipEventRecord(s->ready, st) == hipSuccess;
        }
        if (s->used && hipStreamWaitEvent(st, s->taken, 0) != hipSuccess) return false;   // the previous message has left the mailbox
        if (hipMemcpyPeerAsync(s->box, dst_dev, src, src_dev, bytes, st) != hipSuccess) return false;
        s->used = true;
        return hipEventRecord(s->ready, st) == hipSuccess;
    }
    bool recv(int j, int msg, void * dst, size_t bytes, hipStream_t st) override {
        Slot * s = slot(j, msg);
        if (!s || bytes > cap || !s->used) return false;
        if (s->direct) {
            if (!s->ready_now || hipStreamWaitEvent(st, s->ready_now, 0) != hipSuccess) return false;
            return dst == s->direct;                                                         // (already there)
        }
        if (hipStreamWaitEvent(st, s->ready, 0) != hipSuccess) return false;
        if (hipMemcpyAsync(dst, s->box, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
        return hipEventRecord(s->taken, st) == hipSuccess;
    }
    bool release(int j, int msg, hipStream_t st) override {
        Slot * s = slot(j, msg);
        if (!s) return false;
        if (!s->direct || closed) return true;
        s->released = true;
        return hipEventRecord(s->taken, st) == hipSuccess;
    }
    hipEvent_t waited_on(int j, int msg) override { Slot * s = slot(j, msg); return (s && s->direct) ? s->ready_now : nullptr; }
    void * direct_target(int j, int msg) override {
        const char * e = getenv("RWKV_MI_HOP");      // RWKV_MI_HOP=copy: one peer copy per hop (A/B, tests; read per call)
        const bool copy_only = e && e[0] == 'c';
        Slot * s = slot(j, msg);
        return (s && s->direct && peer_ok && !copy_only) ? s->direct : nullptr;
    }
    bool produce_begin(int j, int msg, hipStream_t st) override {
        Slot * s = slot(j, msg);
        if (!s || !s->direct) return false;
        return closed || !s->released || hipStreamWaitEvent(st, s->taken, 0) == hipSuccess;
    }
    bool produce_end(int j, int msg, hipStream_t st, hipEvent_t done) override {
        Slot * s = slot(j, msg);
        if (!s || !s->direct) return false;
        s->used = true;
        const bool own = getenv("RWKV_MI_HOP_OWN_EVENT") != nullptr;       // (A/B, tests: always the hop's own event; read per call)
        if (done && src_dev == dst_dev && !own) { s->ready_now = done; return true; }   // the chain's marker behind the launch IS "the message is there"
        s->ready_now = s->ready;
        return hipEventRecord(s->ready, st) == hipSuccess;
    }
};

// ---- librccl.so, bound at run time (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclSend, ncclRecv, ncclCommDestroy) ----
struct UniqueId { char internal[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct Rccl {
    void * lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
    std::string why;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && Send && Recv; }
};
Rccl & rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // an already loaded copy first (torch.distributed brings its own): two RCCL instances in one process would not share topology state
        const char * names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char * n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.lib) break; }
        for (const char * n : names) { if (r.lib) break; r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
        if (!r.lib) { const char * e = dlerror(); r.why = e ? e : "librccl.so not found"; return; }
        r.GetUniqueId = (int (*)(UniqueId *)) dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (int (*)(void **, int, UniqueId, int)) dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (int (*)(void *)) dlsym(r.lib, "ncclCommDestroy");
        r.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t)) dlsym(r.lib, "ncclSend");
        r.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t)) dlsym(r.lib, "ncclRecv");
        r.GetErrorString = (const char * (*)(int)) dlsym(r.lib, "ncclGetErrorString");
        if (!r.ok()) r.why = "librccl.so lacks a point-to-point entry point";
    });
    return r;
}
constexpr int k_nccl_uint8 = 1;   // ncclUint8 (rccl.h)

// A point-to-point edge of an RCCL communicator. `side` = true puts the transfer on an own stream, tied to the stage's stream with
// events: the token feedback (last -> first) must not queue behind the forward hops of its rank. Point-to-point calls of one stream
// run in order and a send completes only against its receive, so with everything on one stream a two-rank pipeline with two decode
// streams dead-locks: rank 0 is in send x(t, 1) while rank 1 is in send token(t, 0), each waiting for the other's next receive.
// Forward hops form a chain without cycles and stay on the stage's stream; the feedback gets its own communicator and stream.
struct RcclHop : Hop {
    void * comm; int peer; bool side;
    hipStream_t own = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    RcclHop(void * c, int p, bool s, bool & ok) : comm(c), peer(p), side(s) {
        ok = true;
        if (side) ok = hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ev_a, hipEventDisableTiming) == hipSuccess &&
                       hipEventCreateWithFlags(&ev_b, hipEventDisableTiming) == hipSuccess;
    }
    ~RcclHop() override {
        if (own) { (void) hipStreamSynchronize(own); (void) hipStreamDestroy(own); }
        if (ev_a) (void) hipEventDestroy(ev_a);
        if (ev_b) (void) hipEventDestroy(ev_b);
    }
    bool send(int, int, const void * src, size_t bytes, hipStream_t st) override {   // (point-to-point calls of one communicator and stream are ordered: no mailbox)
        Rccl & r = rccl();
        if (!side) return r.Send(src, bytes, k_nccl_uint8, peer, comm, st) == 0;
        // behind the producer of src; the stage's stream does not wait for the transfer (src is next written a whole round trip later)
        if (hipEventRecord(ev_a, st) != hipSuccess || hipStreamWaitEvent(own, ev_a, 0) != hipSuccess) return false;
        return r.Send(src, bytes, k_nccl_uint8, peer, comm, own) == 0;
    }
    bool recv(int, int, void * dst, size_t bytes, hipStream_t st) override {
        Rccl & r = rccl();
        if (!side) return r.Recv(dst, bytes, k_nccl_uint8, peer, comm, st) == 0;
        // not before the stage's stream has reached this point (dst is still read by what is queued ahead; and a receive that sits on
        // a compute unit long before its sender has anything keeps that unit from the persistent decode kernel)
        if (hipEventRecord(ev_a, st) != hipSuccess || hipStreamWaitEvent(own, ev_a, 0) != hipSuccess) return false;
        if (r.Recv(dst, bytes, k_nccl_uint8, peer, comm, own) != 0) return false;
        return hipEventRecord(ev_b, own) == hipSuccess && hipStreamWaitEvent(st, ev_b, 0) == hipSuccess;
    }
};

// ---- a communicator WITHOUT RCCL: mailboxes in device memory shared through HIP IPC, hand-shakes through POSIX shared memory ----
// For tests of the multi-process loop on ONE GPU (RCCL refuses two ranks on one device): the protocol of rwkv_mi_stage_run -- forward
// hops, the token feedback on its own communicator, several decode streams in flight -- is the same, only the transport differs. Every
// rank owns one mailbox per communicator (in a communicator a rank receives from exactly one peer: rank - 1 on the forward one, the last
// rank on the feedback one): k_ipc_slots slots of k_ipc_slot_bytes. A message is complete on the host before it is announced (the sender
// synchronises its stream, then bumps `sent`; the receiver waits for `sent`, copies out, synchronises, bumps `taken`): no inter-process
// events, which makes the hop slow and the ordering plain. Not a production transport.
constexpr int k_ipc_slots = 32;
constexpr size_t k_ipc_slot_bytes = 64 * 1024;
struct IpcBox {
    std::atomic<int> state;                  // 0: nothing, 1: handle published
    hipIpcMemHandle_t mem;
    std::atomic<uint64_t> sent[k_ipc_slots], taken[k_ipc_slots];
};
struct IpcShm { std::atomic<int> ready; int world; IpcBox box[16]; };
struct IpcWorld {
    std::string name; int rank = 0, world = 0, device = 0;
    IpcShm * shm = nullptr;
    void * mine = nullptr;                   // this rank's mailbox (device memory)
    void * peer[16] = {};                    // opened mailboxes of the ranks this one sends to
    uint64_t got[k_ipc_slots] = {};          // messages received per slot of this rank's mailbox
    uint64_t put[16][k_ipc_slots] = {};      // messages sent per (destination, slot)
    ~IpcWorld() {
        for (void * p : peer) if (p) (void) hipIpcCloseMemHandle(p);
        if (mine) (void) hipFree(mine);
        if (shm) (void) munmap(shm, sizeof(IpcShm));
        if (rank == 0 && !name.empty()) (void) shm_unlink(name.c_str());
    }
};
static bool spin_until(const std::function<bool()> & pred, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 0;; i++) {
        if (pred()) return true;
        if ((i & 255u) == 255u) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
            std::this_thread::yield();
        }
    }
}
struct IpcHop : Hop {
    IpcWorld * w; int peer;                  // peer: the rank this hop sends to / receives from
    IpcHop(IpcWorld * world, int p) : w(world), peer(p) {}
    bool send(int j, int msg, const void * src, size_t bytes, hipStream_t st) override {
        const int s = j * k_hop_msgs + msg;
        if (s >= k_ipc_slots || bytes > k_ipc_slot_bytes || peer < 0 || peer >= w->world) return false;
        IpcBox & b = w->shm->box[peer];
        if (!w->peer[peer]) {
            if (!spin_until([&] { return b.state.load(std::memory_order_acquire) == 1; }, 60.0)) return false;
            if (hipIpcOpenMemHandle(&w->peer[peer], b.mem, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return false;
        }
        const uint64_t n = w->put[peer][s];
        if (!spin_until([&] { return b.taken[s].load(std::memory_order_acquire) >= n; }, 60.0)) return false;   // the previous message has left the slot
        if (hipMemcpyAsync((char *) w->peer[peer] + (size_t) s * k_ipc_slot_bytes, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        w->put[peer][s] = n + 1;
        b.sent[s].store(n + 1, std::memory_order_release);
        return true;
    }
    bool recv(int j, int msg, void * dst, size_t bytes, hipStream_t st) override {
        const int s = j * k_hop_msgs + msg;
        if (s >= k_ipc_slots || bytes > k_ipc_slot_bytes) return false;
        IpcBox & b = w->shm->box[w->rank];
        const uint64_t n = w->got[s];
        if (!spin_until([&] { return b.sent[s].load(std::memory_order_acquire) > n; }, 60.0)) return false;
        if (hipMemcpyAsync(dst, (const char *) w->mine + (size_t) s * k_ipc_slot_bytes, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        w->got[s] = n + 1;
        b.taken[s].store(n + 1, std::memory_order_release);
        return true;
    }
};

// what rwkv_mi_comm_init / rwkv_mi_comm_init_ipc hand out
struct MiComm { void * nccl = nullptr; IpcWorld * ipc = nullptr; };

// ---------------------------------------------------------------------------------------------------------------
// one stage's share of the loop
// ---------------------------------------------------------------------------------------------------------------
struct StagePart {
    std::vector<rwkv_context *> h;     // one context per decode stream (same stage, same device)
    Hop * in = nullptr;                // residual stream from the previous stage (nullptr on the first)
    Hop * out = nullptr;               // ... to the next stage (nullptr on the last)
    Hop * tok_in = nullptr;            // chosen token from the last stage (first stage of a chain of more than one stage)
    Hop * tok_out = nullptr;           // ... to the first stage (last stage of such a chain)
    uint32_t * d_hist = nullptr;       // (last stage) [n_streams][n_tokens] chosen tokens, device
    size_t n_tokens = 0;
    std::vector<char> hist_in_launch;  // (last stage) per decode stream: the persistent launch appends the token it picks to d_hist itself
    std::vector<char> x_direct;        // per decode stream: the stage's launch stores the residual stream in the next stage's buffer itself
    std::vector<char> tok_direct;      // (last stage) per decode stream: the launch stores the chosen token in the first stage's token word itself
};

// stage `p`, decode stream j, token index t: receive, run the layers, send. Everything is enqueued on the context's stream.
bool stage_iteration(StagePart & p, rwkv_context * err, size_t t, int j) {
    rwkv_context * c = p.h[(size_t) j];
    Model & m = *c->model;
    const size_t D = (size_t) m.n_embed();
    auto fail = [&]() { err->last_error |= c->last_error ? c->last_error : (int) RWKV_ERROR_GRAPH; return false; };
    if (hipSetDevice(m.device) != hipSuccess || !ensure_scratch(c, 1)) return fail();
    if (m.has_embed) {
        if (t > 0 && p.tok_in && !p.tok_in->recv(j, 0, c->d_tokens, sizeof(uint32_t), c->stream)) return fail();
    } else {
        if (!p.in || !p.in->recv(j, 0, c->b.x, D * sizeof(float), c->stream)) return fail();
        if (m.arch_major == 7 && !p.in->recv(j, 1, c->b.v_first, D * sizeof(float), c->stream)) return fail();
    }
    const bool xd = !m.has_head && (size_t) j < p.x_direct.size() && p.x_direct[(size_t) j] && c->mega != nullptr;
    if (xd && !p.out->produce_begin(j, 0, c->stream)) return fail();
    // (last stage of a chain) the launch leaves the chosen token in the first stage's token word itself
    const bool td = m.has_head && !m.has_embed && p.tok_out && (size_t) j < p.tok_direct.size() && p.tok_direct[(size_t) j] && c->mega != nullptr &&
                    (size_t) j < p.hist_in_launch.size() && p.hist_in_launch[(size_t) j];
    if (td) { if (!p.tok_out->produce_begin(j, 0, c->stream)) return fail(); c->ntok_out = (uint32_t *) p.tok_out->direct_target(j, 0); }
    // (what the stream was just made to wait on -- the previous stage's launch, when the hop used the device chain's marker for it)
    c->chain_covered = m.has_embed ? (p.tok_in && t > 0 ? p.tok_in->waited_on(j, 0) : nullptr) : p.in->waited_on(j, 0);
    const bool fwd = forward_decode(c, m.has_head);
    c->chain_covered = nullptr; c->ntok_out = nullptr;
    if (!fwd) return fail();
    if (td) {
        if (!p.tok_out->produce_end(j, 0, c->stream, mega_chain_marker(c))) return fail();
    } else if (m.has_head) {
        // the chosen token: where this context's embedding reads it (a one-stage "chain"), else in the slot the feedback hop sends from
        uint32_t * dst = m.has_embed ? c->d_tokens : c->d_next_token;
        if (folded_argmax_target(c) != dst) launch_argmax(c->d_logits, m.n_vocab(), dst, c->stream);   // (else the persistent launch left it there)
        const bool in_launch = (size_t) j < p.hist_in_launch.size() && p.hist_in_launch[(size_t) j] && folded_argmax_target(c) == dst;
        if (!in_launch && hipMemcpyAsync(p.d_hist + (size_t) j * p.n_tokens + t, dst, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return fail();
        if (p.tok_out && t + 1 < p.n_tokens && !p.tok_out->send(j, 0, dst, sizeof(uint32_t), c->stream)) return fail();
    } else {
        if (xd) { if (!p.out->produce_end(j, 0, c->stream, mega_chain_marker(c))) return fail(); }
        else if (!p.out || !p.out->send(j, 0, c->b.x, D * sizeof(float), c->stream)) return fail();
        if (m.arch_major == 7 && !p.out->send(j, 1, c->b.v_first, D * sizeof(float), c->stream)) return fail();
    }
    // what this iteration received (or, for the token word of the first stage, what it was seeded with) may be overwritten from here on
    if (m.has_embed) {
        if (p.tok_in && !p.tok_in->release(j, 0, c->stream)) return fail();
    } else {
        if (!p.in->release(j, 0, c->stream)) return fail();
        if (m.arch_major == 7 && !p.in->release(j, 1, c->stream)) return fail();
    }
    return true;
}

// (last stage) the persistent launches of every decode stream append the tokens they pick to the history themselves: no copy per token.
// Only when the launch really folds the argmax into the slot the loop reads; undone by hist_off() on every exit (the history is freed).
void hist_on(StagePart & p) {
    p.hist_in_launch.assign(p.h.size(), 0);
    if (!p.d_hist) return;
    for (size_t j = 0; j < p.h.size(); j++) {
        rwkv_context * c = p.h[j];
        if (!c->mega || !c->model->has_head || folded_argmax_target(c) == nullptr) continue;
        if (hipSetDevice(c->model->device) != hipSuccess) continue;
        if (mega_v6_set_history(c->mega, p.d_hist + j * p.n_tokens, p.n_tokens, c->stream)) p.hist_in_launch[j] = 1;
    }
}
void hist_off(StagePart & p) {
    for (size_t j = 0; j < p.hist_in_launch.size(); j++) {
        rwkv_context * c = p.h[j];
        if (!p.hist_in_launch[j] || !c->mega) continue;
        if (hipSetDevice(c->model->device) == hipSuccess) (void) mega_v6_set_history(c->mega, nullptr, 0, c->stream);
        p.hist_in_launch[j] = 0;
    }
}
struct HistScope { StagePart & p; explicit HistScope(StagePart & q) : p(q) { hist_on(p); } ~HistScope() { hist_off(p); } };

// The stages whose single-token step is one directly issued persistent launch store their output where the next stage reads it (a graph
// replay would keep the pointer it was captured with: those stages keep the copy). Undone on every exit: the target belongs to this call.
void x_direct_on(StagePart & p) {
    p.x_direct.assign(p.h.size(), 0);
    if (!p.out) return;
    for (size_t j = 0; j < p.h.size(); j++) {
        rwkv_context * c = p.h[j];
        void * tgt = p.out->direct_target((int) j, 0);
        if (!tgt || !c->mega || c->model->has_head || !(!c->use_graph || single_launch_step(c, false))) continue;
        if (mega_v6_set_x_out(c->mega, (float *) tgt)) p.x_direct[j] = 1;
    }
}
void x_direct_off(StagePart & p) {
    for (size_t j = 0; j < p.x_direct.size(); j++) if (p.x_direct[j] && p.h[j]->mega) (void) mega_v6_set_x_out(p.h[j]->mega, nullptr);
    p.x_direct.clear(); p.tok_direct.clear();
}
// (last stage) the token feedback the same way: a launch that folds the argmax AND appends to the history itself (hist_on) can leave the token
// in the first stage's token word -- nothing on this side reads it then
void tok_direct_on(StagePart & p) {
    p.tok_direct.assign(p.h.size(), 0);
    if (!p.tok_out) return;
    for (size_t j = 0; j < p.h.size(); j++) {
        rwkv_context * c = p.h[j];
        if (!p.tok_out->direct_target((int) j, 0) || !c->mega || !c->model->has_head || c->model->has_embed || folded_argmax_target(c) == nullptr) continue;
        if (!(j < p.hist_in_launch.size() && p.hist_in_launch[j]) || !(!c->use_graph || single_launch_step(c, true))) continue;
        p.tok_direct[j] = 1;
    }
}
struct XDirectScope {
    std::vector<StagePart> & ps;
    explicit XDirectScope(std::vector<StagePart> & q) : ps(q) { for (StagePart & p : ps) { x_direct_on(p); tok_direct_on(p); } }
    ~XDirectScope() { for (StagePart & p : ps) x_direct_off(p); }
};

bool seed_tokens(StagePart & p, rwkv_context * err, const uint32_t * first_tokens) {
    for (size_t j = 0; j < p.h.size(); j++) {
        rwkv_context * c = p.h[j];
        if (hipSetDevice(c->model->device) != hipSuccess || !upload_tokens_for(c, first_tokens + j, 1)) { err->last_error |= c->last_error ? c->last_error : (int) RWKV_ERROR_GRAPH; return false; }
    }
    return true;
}

// drains every context of a part; a poll time-out of the persistent kernel invalidates the run
bool drain(StagePart & p, rwkv_context * err) {
    bool ok = true;
    for (rwkv_context * c : p.h) {
        if (hipSetDevice(c->model->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { ok = false; continue; }
        if (c->mega && mega_v6_aborted(c->mega, c->stream)) { recover_from_abort(c); ok = false; }
    }
    if (!ok) err->last_error |= RWKV_ERROR_GRAPH;
    return ok;
}

struct DevMem {
    void * p = nullptr; int dev = 0;
    ~DevMem() { if (p) { (void) hipSetDevice(dev); (void) hipFree(p); } }
};

}  // namespace

namespace rwkvmi {

// ---- the stage chain of one process (fronts[j]->stages, or plain one-device contexts = chains of one stage) ----
bool pipeline_decode_greedy(rwkv_context * const * fronts, size_t n_streams, const uint32_t * first_tokens, size_t n_tokens, uint32_t * tokens_out, float * elapsed_ms) {
    rwkv_context * f0 = fronts[0];
    auto stage_of = [](rwkv_context * f, size_t s) { return f->stages.empty() ? f : f->stages[s]; };
    const size_t S = f0->stages.empty() ? 1 : f0->stages.size();
    for (size_t j = 0; j < n_streams; j++) {
        rwkv_context * f = fronts[j];
        RW_CTX_CHECK(f0, RWKV_ERROR_ARGS, false, f != nullptr && (f->stages.empty() ? 1 : f->stages.size()) == S, "decode stream %zu is not a context of the same chain", j);
        for (size_t s = 0; s < S; s++) {
            const Model & a = *stage_of(f, s)->model, & b = *stage_of(f0, s)->model;
            RW_CTX_CHECK(f0, RWKV_ERROR_ARGS, false, a.device == b.device && a.layer_begin == b.layer_begin && a.layer_end == b.layer_end, "decode stream %zu is not a clone of stream 0", j);
        }
        RW_CTX_CHECK(f0, RWKV_ERROR_ARGS, false, first_tokens[j] < (uint32_t) f0->model->n_vocab(), "Token of stream %zu is out of range", j);
        // one recurrent state and one token buffer per context: the same context twice would interleave two streams on them
        for (size_t i = 0; i < j; i++) RW_CTX_CHECK(f0, RWKV_ERROR_ARGS, false, fronts[i] != f, "decode streams %zu and %zu are the same context", i, j);
    }
    int prev_dev = 0;
    (void) hipGetDevice(&prev_dev);
    struct Restore { int d; ~Restore() { (void) hipSetDevice(d); } } restore{prev_dev};
    std::vector<StagePart> parts(S);
    std::vector<std::unique_ptr<Hop>> hops;
    const size_t hb = (size_t) f0->model->n_embed() * sizeof(float);
    for (size_t s = 0; s < S; s++) {
        for (size_t j = 0; j < n_streams; j++) { rwkv_context * c = stage_of(fronts[j], s); c->last_error = 0; c->print_errors = f0->print_errors; parts[s].h.push_back(c); }
        parts[s].n_tokens = n_tokens;
    }
    auto add_hop = [&](size_t from, size_t to, size_t bytes) -> Hop * {
        bool ok = false;
        hops.emplace_back(new LocalHop(parts[from].h[0]->model->device, parts[to].h[0]->model->device, (int) n_streams, bytes, ok));
        return ok ? hops.back().get() : nullptr;
    };
    for (size_t s = 0; s + 1 < S; s++) {
        Hop * h = add_hop(s, s + 1, hb);
        RW_CTX_CHECK(f0, RWKV_ERROR_ALLOC, false, h != nullptr, "cannot allocate the hand-over buffers of stage %zu", s);
        parts[s].out = h; parts[s + 1].in = h;
    }
    if (S > 1) {
        Hop * h = add_hop(S - 1, 0, sizeof(uint32_t));
        RW_CTX_CHECK(f0, RWKV_ERROR_ALLOC, false, h != nullptr, "cannot allocate the token feedback buffers");
        parts[S - 1].tok_out = h; parts[0].tok_in = h;
    }
    DevMem hist; hist.dev = parts[S - 1].h[0]->model->device;
    RW_CTX_CHECK(f0, RWKV_ERROR_ALLOC, false, hipSetDevice(hist.dev) == hipSuccess && hipMalloc(&hist.p, n_streams * n_tokens * sizeof(uint32_t)) == hipSuccess, "cannot allocate the token history");
    parts[S - 1].d_hist = (uint32_t *) hist.p;
    if (!seed_tokens(parts[0], f0, first_tokens)) return false;
    // the hops deliver straight into the buffers the receiving stages read (fixed for the length of this call: T stays 1)
    for (size_t s = 0; s < S; s++)
        for (size_t j = 0; j < n_streams; j++) {
            rwkv_context * c = parts[s].h[j];
            if (hipSetDevice(c->model->device) != hipSuccess || !ensure_scratch(c, 1)) { f0->last_error |= c->last_error ? c->last_error : (int) RWKV_ERROR_ALLOC; return false; }
            if (parts[s].in) { (void) parts[s].in->bind((int) j, 0, c->b.x); if (c->model->arch_major == 7) (void) parts[s].in->bind((int) j, 1, c->b.v_first); }
            if (parts[s].tok_in) (void) parts[s].tok_in->bind((int) j, 0, c->d_tokens);
        }
    for (auto & h : hops) h->set_closed_loop(S > 1);       // (S > 1: the token feedback closes the loop for every decode stream)
    HistScope hist_scope(parts[S - 1]);
    XDirectScope x_scope(parts);
    for (size_t s = 0; s < S; s++) for (rwkv_context * c : parts[s].h) { if (hipSetDevice(c->model->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { f0->last_error |= RWKV_ERROR_GRAPH; return false; } }
    const auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    for (size_t t = 0; t < n_tokens && ok; t++)
        for (size_t j = 0; j < n_streams && ok; j++)
            for (size_t s = 0; s < S && ok; s++) ok = stage_iteration(parts[s], f0, t, (int) j);
    bool clean = true;
    for (size_t s = 0; s < S; s++) clean = drain(parts[s], f0) && clean;   // (always: nothing may be in flight when the hops are freed)
    const auto t1 = std::chrono::steady_clock::now();
    RW_CTX_CHECK(f0, RWKV_ERROR_GRAPH, false, ok, "greedy decode through the stage chain failed: %s", hipGetErrorString(hipGetLastError()));
    RW_CTX_CHECK(f0, RWKV_ERROR_GRAPH, false, clean, "the persistent decode kernel of a stage timed out; the stage continues on the per-layer launches");
    if (elapsed_ms) *elapsed_ms = (float) std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (tokens_out) RUN_OK(f0, hipMemcpy(tokens_out, hist.p, n_streams * n_tokens * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return true;
}

// resident state of a chain: every stage owns the slice of its layers
bool pipeline_state_load(rwkv_context * front, const float * state_in) {
    for (rwkv_context * c : front->stages) {
        Model & m = *c->model;
        c->last_error = 0; c->print_errors = front->print_errors;
        bool ok = hipSetDevice(m.device) == hipSuccess;
        if (ok && state_in) {
            const int64_t per = m.state_per_layer();
            const int64_t off = (int64_t) m.layer_begin * per, cnt = (int64_t) (m.layer_end - m.layer_begin) * per;
            ok = hipMemcpyAsync(c->state[c->cur] + off, state_in + off, (size_t) cnt * sizeof(float), hipMemcpyHostToDevice, c->stream) == hipSuccess;
        } else if (ok) ok = state_from_host(c, nullptr);
        ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
        if (!ok) { front->last_error |= c->last_error ? c->last_error : (int) RWKV_ERROR_GRAPH; return false; }
    }
    return true;
}
bool pipeline_state_store(rwkv_context * front, float * state_out) {
    for (rwkv_context * c : front->stages) {
        Model & m = *c->model;
        const int64_t per = m.state_per_layer();
        const int64_t off = (int64_t) m.layer_begin * per, cnt = (int64_t) (m.layer_end - m.layer_begin) * per;
        const bool ok = hipSetDevice(m.device) == hipSuccess &&
                        hipMemcpyAsync(state_out + off, c->state[c->cur] + off, (size_t) cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                        hipStreamSynchronize(c->stream) == hipSuccess;
        if (!ok) { front->last_error |= RWKV_ERROR_GRAPH; return false; }
    }
    return true;
}

}  // namespace rwkvmi

extern "C" {

RWKV_API bool rwkv_mi_decode_greedy_streams(struct rwkv_context * const * ctxs, size_t n_streams, const uint32_t * first_tokens, size_t n_tokens,
                                            uint32_t * tokens_out, float * elapsed_ms) {
    if (!ctxs || n_streams == 0 || !ctxs[0]) return false;
    ctxs[0]->last_error = RWKV_ERROR_NONE;
    RW_CTX_CHECK(ctxs[0], RWKV_ERROR_ARGS, false, first_tokens != nullptr && n_tokens > 0, "first_tokens is NULL or n_tokens is 0");
    return pipeline_decode_greedy(ctxs, n_streams, first_tokens, n_tokens, tokens_out, elapsed_ms);
}

// ---- one process per GPU: communicators of librccl.so ----
RWKV_API bool rwkv_mi_comm_unique_id(void * id_out, size_t capacity) {
    Rccl & r = rccl();
    if (!r.ok() || !id_out || capacity < sizeof(UniqueId)) { if (!r.ok()) fprintf(stderr, "rwkv_mi_comm_unique_id: %s\n", r.why.c_str()); return false; }
    UniqueId id;
    if (r.GetUniqueId(&id) != 0) return false;
    memcpy(id_out, &id, sizeof(id));
    return true;
}
RWKV_API void * rwkv_mi_comm_init(const void * id128, int rank, int world) {
    Rccl & r = rccl();
    if (!r.ok() || !id128) { if (!r.ok()) fprintf(stderr, "rwkv_mi_comm_init: %s\n", r.why.c_str()); return nullptr; }
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    void * comm = nullptr;
    const int e = r.CommInitRank(&comm, world, id, rank);
    if (e != 0) { fprintf(stderr, "rwkv_mi_comm_init: ncclCommInitRank failed: %s\n", r.GetErrorString ? r.GetErrorString(e) : "?"); return nullptr; }
    MiComm * c = new MiComm();
    c->nccl = comm;
    return c;
}
// The same kind of handle over HIP-IPC mailboxes (see IpcWorld): `name` = a POSIX shared-memory name ("/something") unique to this
// communicator of this run, the same on every rank; rank 0 creates the segment. Current device = the rank's device.
RWKV_API void * rwkv_mi_comm_init_ipc(const char * name, int rank, int world) {
    if (!name || name[0] != '/' || rank < 0 || world < 1 || world > 16 || rank >= world) return nullptr;
    std::unique_ptr<IpcWorld> w(new IpcWorld());
    w->rank = rank; w->world = world;
    (void) hipGetDevice(&w->device);
    int fd = -1;
    if (rank == 0) {
        (void) shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(IpcShm)) != 0) { if (fd >= 0) close(fd); fprintf(stderr, "rwkv_mi_comm_init_ipc: cannot create %s\n", name); return nullptr; }
    } else {
        if (!spin_until([&] { fd = shm_open(name, O_RDWR, 0600); if (fd < 0) return false; struct stat sb; if (fstat(fd, &sb) != 0 || (size_t) sb.st_size < sizeof(IpcShm)) { close(fd); fd = -1; return false; } return true; }, 60.0)) {
            fprintf(stderr, "rwkv_mi_comm_init_ipc: %s never appeared\n", name); return nullptr;
        }
    }
    void * m = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return nullptr;
    w->shm = (IpcShm *) m;
    w->name = name;
    if (rank == 0) { w->shm->world = world; w->shm->ready.store(1, std::memory_order_release); }      // (a fresh segment is zero-filled)
    else if (!spin_until([&] { return w->shm->ready.load(std::memory_order_acquire) == 1; }, 60.0)) return nullptr;
    if (hipMalloc(&w->mine, (size_t) k_ipc_slots * k_ipc_slot_bytes) != hipSuccess) return nullptr;
    IpcBox & b = w->shm->box[rank];
    if (hipIpcGetMemHandle(&b.mem, w->mine) != hipSuccess) { fprintf(stderr, "rwkv_mi_comm_init_ipc: hipIpcGetMemHandle failed: %s\n", hipGetErrorString(hipGetLastError())); return nullptr; }
    b.state.store(1, std::memory_order_release);
    MiComm * c = new MiComm();
    c->ipc = w.release();
    return c;
}
RWKV_API void rwkv_mi_comm_free(void * comm) {
    MiComm * c = (MiComm *) comm;
    if (!c) return;
    Rccl & r = rccl();
    if (c->nccl && r.ok()) (void) r.CommDestroy(c->nccl);
    delete c->ipc;
    delete c;
}
RWKV_API bool rwkv_mi_comm_available(void) { return rccl().ok(); }

// This rank's stage of the pipeline for n_tokens greedy tokens on n_streams decode streams (handles: contexts of rwkv_mi_init_stage and
// its clones, bound to ONE stream with rwkv_mi_set_stream or on their own). comm_fwd carries the residual stream rank -> rank + 1,
// comm_fb the chosen token from the last rank to rank 0 (two communicators: see RcclHop). tokens_out ([n_streams][n_tokens]) is
// filled on the last rank. elapsed_ms: this rank's wall time of the loop including the final drain.
RWKV_API bool rwkv_mi_stage_run(struct rwkv_context * const * handles, size_t n_streams, const uint32_t * first_tokens, size_t n_tokens,
                                int rank, int world, void * comm_fwd, void * comm_fb, uint32_t * tokens_out, float * elapsed_ms) {
    if (!handles || n_streams == 0 || !handles[0]) return false;
    rwkv_context * c0 = handles[0];
    c0->last_error = RWKV_ERROR_NONE;
    RW_CTX_CHECK(c0, RWKV_ERROR_ARGS, false, c0->stages.empty(), "rwkv_mi_stage_run takes stage contexts (rwkv_mi_init_stage), not a RWKV_MI_DEVICES chain");
    RW_CTX_CHECK(c0, RWKV_ERROR_ARGS, false, n_tokens > 0 && rank >= 0 && rank < world, "bad n_tokens / rank / world");
    const Model & m = *c0->model;
    RW_CTX_CHECK(c0, RWKV_ERROR_ARGS, false, m.has_embed == (rank == 0) && m.has_head == (rank == world - 1), "the stage's layer range does not fit rank %d of %d", rank, world);
    const MiComm * cf = (const MiComm *) comm_fwd, * cb = (const MiComm *) comm_fb;
    RW_CTX_CHECK(c0, RWKV_ERROR_ARGS, false, world == 1 || (cf && cb && ((cf->nccl && cb->nccl && rccl().ok()) || (cf->ipc && cb->ipc))),
                 "a pipeline of %d ranks needs two communicators of one kind (rwkv_mi_comm_init / rwkv_mi_comm_init_ipc)", world);
    RW_CTX_CHECK(c0, RWKV_ERROR_ARGS, false, !m.has_embed || first_tokens != nullptr, "the first stage needs the seed tokens");
    StagePart part;
    for (size_t j = 0; j < n_streams; j++) {
        RW_CTX_CHECK(c0, RWKV_ERROR_ARGS, false, handles[j] && handles[j]->model->device == m.device, "decode stream %zu is not on this stage's device", j);
        handles[j]->last_error = 0;
        part.h.push_back(handles[j]);
    }
    part.n_tokens = n_tokens;
    RUN_OK(c0, hipSetDevice(m.device));
    std::unique_ptr<Hop> in, out, tin, tout;
    bool ok = true, o = true;
    auto hop = [&](const MiComm * c, int peer, bool side) -> Hop * {
        if (c->ipc) return new IpcHop(c->ipc, peer);
        Hop * h = new RcclHop(c->nccl, peer, side, o);
        ok = ok && o;
        return h;
    };
    if (rank > 0) { in.reset(hop(cf, rank - 1, false)); part.in = in.get(); }
    if (rank + 1 < world) { out.reset(hop(cf, rank + 1, false)); part.out = out.get(); }
    if (world > 1 && rank == 0) { tin.reset(hop(cb, world - 1, true)); part.tok_in = tin.get(); }
    if (world > 1 && rank == world - 1) { tout.reset(hop(cb, 0, true)); part.tok_out = tout.get(); }
    RW_CTX_CHECK(c0, RWKV_ERROR_ALLOC, false, ok, "cannot create the streams of the token feedback");
    DevMem hist; hist.dev = m.device;
    if (m.has_head) {
        RUN_OK(c0, hipMalloc(&hist.p, n_streams * n_tokens * sizeof(uint32_t)));
        part.d_hist = (uint32_t *) hist.p;
    }
    if (m.has_embed && !seed_tokens(part, c0, first_tokens)) return false;
    HistScope hist_scope(part);
    for (rwkv_context * c : part.h) RUN_OK(c0, hipStreamSynchronize(c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t t = 0; t < n_tokens && ok; t++)
        for (size_t j = 0; j < n_streams && ok; j++) ok = stage_iteration(part, c0, t, (int) j);
    const bool clean = drain(part, c0);
    in.reset(); out.reset(); tin.reset(); tout.reset();   // (drains the feedback streams)
    const auto t1 = std::chrono::steady_clock::now();
    RW_CTX_CHECK(c0, RWKV_ERROR_GRAPH, false, ok, "the stage's decode loop failed: %s", hipGetErrorString(hipGetLastError()));
    RW_CTX_CHECK(c0, RWKV_ERROR_GRAPH, false, clean, "the persistent decode kernel timed out; the stage continues on the per-layer launches");
    if (elapsed_ms) *elapsed_ms = (float) std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (tokens_out && m.has_head) RUN_OK(c0, hipMemcpy(tokens_out, hist.p, n_streams * n_tokens * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return true;
}

}  // extern "C"
