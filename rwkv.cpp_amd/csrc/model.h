// model.h -- weights resident in HBM + the per-context execution state of librwkv.so.
#pragma once
#include <string>

#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>

#include "common.h"
#include "kernels.h"

namespace rwkvmi {

// Parameter slots of one layer; names follow the reference's key table (rwkv_model_loading.inc:132-282).
struct LayerW {
    const DevTensor *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
    // v4 / v5
    const DevTensor *att_time_mix_k = nullptr, *att_time_mix_v = nullptr, *att_time_mix_r = nullptr, *att_time_mix_g = nullptr;
    const DevTensor *att_time_first = nullptr, *att_time_decay = nullptr, *att_time_faaaa = nullptr;
    const DevTensor *att_key = nullptr, *att_value = nullptr, *att_receptance = nullptr, *att_output = nullptr, *att_gate = nullptr;
    const DevTensor *att_ln_x_w = nullptr, *att_ln_x_b = nullptr;
    // v6
    const DevTensor *att_time_maa_x = nullptr, *att_time_maa_w = nullptr, *att_time_maa_k = nullptr, *att_time_maa_v = nullptr,
                    *att_time_maa_r = nullptr, *att_time_maa_g = nullptr, *att_time_maa_w1 = nullptr, *att_time_maa_w2 = nullptr,
                    *att_time_decay_w1 = nullptr, *att_time_decay_w2 = nullptr;
    // v7
    const DevTensor *att_x_rwkvag = nullptr, *att_w0 = nullptr, *att_w1 = nullptr, *att_w2 = nullptr, *att_a0 = nullptr, *att_a1 = nullptr,
                    *att_a2 = nullptr, *att_g1 = nullptr, *att_g2 = nullptr, *att_v0 = nullptr, *att_v1 = nullptr, *att_v2 = nullptr,
                    *att_r_k = nullptr, *att_k_k = nullptr, *att_k_a = nullptr;
    // channel mixing
    const DevTensor *ffn_time_mix_k = nullptr, *ffn_time_mix_r = nullptr, *ffn_time_maa_k = nullptr, *ffn_time_maa_r = nullptr, *ffn_x_k = nullptr;
    const DevTensor *ffn_key = nullptr, *ffn_value = nullptr, *ffn_receptance = nullptr;
};

struct Model {
    FileHeader header{};
    int arch_major = 4, arch_minor = 0;
    int64_t head_count = 0, head_size = 0, ffn_size = 0;
    int64_t max_lowrank = 0;  // widest intermediate of a low-rank pair (v6 5*r, decay rank; v7 ranks)
    // Single-token path the first context of this model measured as fastest on its device (engine.hip, calibrate_decode_path):
    // 0 not measured, 1 persistent kernel on register prefetch, 2 persistent kernel on the LDS-DMA ring, 3 seven launches per layer
    // (RWKV-6; RWKV-4 / RWKV-7: 4 persistent kernel of persist_v47.hip, 3 the fused per-layer launches).
    // Clones reuse it instead of timing every path again.
    mutable std::atomic<int> decode_choice{0};

    // Device images a decode path derives from the weights alone (ring_v6.hip: the per-workgroup weight streams, the blocked W2, the
    // layer table): built by the first context that needs them, shared by every context of the model, freed with the last holder.
    mutable std::mutex derived_mu;
    mutable void * ring_shared = nullptr;

    std::vector<std::unique_ptr<DevTensor>> tensors;
    std::unordered_map<std::string, DevTensor *> by_name;

    const DevTensor *emb = nullptr, *ln0_w = nullptr, *ln0_b = nullptr, *ln_out_w = nullptr, *ln_out_b = nullptr, *head = nullptr;
    std::vector<LayerW> layers;  // indexed by absolute layer id; only [layer_begin, layer_end) are populated

    // pipeline stage owned by this process (whole model by default)
    uint32_t layer_begin = 0, layer_end = 0;
    bool has_embed = true, has_head = true;

    void * arena = nullptr;       // one HBM allocation holding every parameter plane
    size_t arena_bytes = 0;
    int device = 0;
    uint64_t bytes_per_token = 0; // algorithmic bytes of one decoded token on this stage
    uint64_t weight_bytes = 0;
    double load_seconds = 0.0;    // file payload -> HBM (pass 2 of load_model: reads, copies, re-pack kernels)

    std::atomic<int> refcount{0};

    int64_t n_embed() const { return header.n_embed; }
    int64_t n_vocab() const { return header.n_vocab; }
    int64_t n_layer() const { return header.n_layer; }
    int64_t state_per_layer() const { return arch_major >= 5 ? n_embed() * (2 + head_size) : n_embed() * 5; }
    int64_t state_len() const { return state_per_layer() * n_layer(); }
};

// ---- sequence mode on the matrix cores (prefill.hip) ----
struct TileAct {   // quantised activations of T tokens in the tile-major image (see prefill.hip)
    int8_t * q = nullptr; float * d = nullptr; float * s = nullptr; float * o = nullptr;
    int64_t T_pad = 0;
};
size_t  tile_act_bytes(int64_t T, int64_t K);
TileAct tile_act_at(void * base, int64_t T, int64_t K);
void launch_quantize_act_tiles(const float * x, int64_t T, int64_t K, int wtype, const TileAct & out, hipStream_t st);
// up to 5 inputs of the same shape in one launch
void launch_quantize_act_tiles_batched(int n, const float * const * xs, int64_t T, int64_t K, int wtype, const TileAct * outs, hipStream_t st);
// sequence-mode mixes writing their outputs as tile images (prefill.hip); `outs` = one image per output
bool launch_v6_mix2_seq(const V6Mix2Args & a, int64_t T, int64_t D, int64_t R, hipStream_t st, const TileAct * outs = nullptr, int wtype = 0);
bool launch_mix_seq_q(const MixArgs & a, int64_t T, int64_t D, hipStream_t st, const TileAct * outs, int wtype);
bool launch_groupnorm_seq_q(const float * x, const float * lw, const float * lb, float eps, const float * gate, int64_t T, int64_t H, int64_t S,
                            const TileAct & out, int wtype, hipStream_t st);
// workspace of the split walk (GEMMs with too few output tiles for the chip): partial sums + one zeroed counter per tile
struct MmqWs { float * part = nullptr; size_t part_bytes = 0; int * counters = nullptr; int n_counters = 0; };
bool launch_mmq_mfma(const DevTensor & W, const TileAct & x, int64_t T, float * y, int64_t ldy, const Epi & epi, hipStream_t st, const MmqWs * ws = nullptr);
bool launch_mmq_mfma_q(const DevTensor & W, const TileAct & x, int64_t T, const Epi & epi, const TileAct & out, int out_wtype, hipStream_t st);   // output only as the next product's quantised image
// up to 4 products of the same shape and type in one launch (y_i = epi_i(W_i . x_i))
bool launch_mmq_mfma_batched(int n, const DevTensor * const * Ws, const TileAct * xs, float * const * ys, const Epi * epis, int64_t T, int64_t ldy,
                             const MmqWs * ws, hipStream_t st);
bool ensure_pf(const DevTensor & W, hipStream_t st);
void prefill_prepare_current_device();   // per-device kernel attributes of the sequence-mode kernels (multi-device processes)
void free_pf(const DevTensor & W);
bool launch_wkv6_seq(const float * r, const float * k, const float * v, const float * u, int u_per_chan, const float * w, int w_mode,
                     const float * state_in, float * state_out, float * out, int64_t T, int64_t H, hipStream_t st);
constexpr int64_t k_mfma_min_tokens = 32;   // sequence calls of at least this many tokens per pass take the GEMM path

// temperature / top-p sampling on the logits in HBM (sampling.hip). u < 0: draw from the counter-based generator (seed, *counter; the
// counter is advanced on the device). The token is written to out_token (and to hist[hist_pos] when hist is given).
void launch_sample(const float * logits, int n, float temperature, float top_p, float u, unsigned long long seed, unsigned long long * counter,
                   float * probs, uint32_t * out_token, uint32_t * hist, int hist_pos, hipStream_t st);

// Loads [layer_begin, layer_end) of the file (layer_end == UINT32_MAX: all layers) onto the current HIP device.
// Returns nullptr with the thread-local error set, like the reference loader (rwkv_model_loading.inc:288-419).
Model * load_model(const char * path, uint32_t layer_begin, uint32_t layer_end);
void    release_model(Model * m);
bool    scan_stage_costs(const char * path, std::vector<uint64_t> & per_layer, uint64_t & head_bytes);

}  // namespace rwkvmi

// ---------------------------------------------------------------------------------------------------------------
// The context (opaque to callers).
// ---------------------------------------------------------------------------------------------------------------

struct rwkv_context {
    rwkvmi::Model * model = nullptr;
    uint32_t n_threads = 0;
    int  last_error = 0;
    bool print_errors = false;  // a fresh reference context starts silent (rwkv.cpp:74 value-initialises it)

    hipStream_t stream = nullptr;
    bool owns_stream = true;   // false after rwkv_mi_set_stream (the caller's stream, e.g. torch's current stream)

    // Device-resident recurrent state, ping-pong (kernels read [cur], write [cur ^ 1]).
    float * state[2] = {nullptr, nullptr};
    int cur = 0;

    // Scratch for T tokens (grown on demand).
    int64_t scratch_T = 0;
    void *  scratch = nullptr;
    size_t  scratch_bytes = 0;
    struct Buf {
        float *x, *xn, *sx, *m[6], *r, *k, *v, *g, *w, *a, *t0, *t1, *t2, *out, *ffk, *lr1, *lr2, *v_first, *xlast;
        rwkvmi::QAct qa;
        void * tile = nullptr;   // tile-major quantised activations (T >= k_mfma_min_tokens)
        void * tiles[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // the same for inputs quantised ahead, several per launch (row length D)
        rwkvmi::MmqWs ws;         // workspace of the split walk (prefill.hip)
    } b{};

    uint32_t * d_tokens = nullptr;
    int64_t    d_tokens_cap = 0;
    float *    d_logits = nullptr;
    uint32_t * d_next_token = nullptr;
    float *    d_probs = nullptr;                 // sampler scratch (n_vocab floats), allocated on first use
    unsigned long long * d_rng_counter = nullptr;

    // pinned host staging for tokens / logits
    uint32_t * h_tokens = nullptr;
    int64_t    h_tokens_cap = 0;

    // captured single-token graphs: [cur][with_logits]
    hipGraphExec_t graph_exec[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool use_graph = true;

    // Layer pipeline inside ONE process (RWKV_MI_DEVICES, pipeline.cpp): this context is then only the front of a chain of stage
    // contexts, one per listed device; every rwkv.h entry point walks the chain.
    std::vector<rwkv_context *> stages;
    hipEvent_t handoff_ev = nullptr;   // (stage contexts) residual stream handed to the next stage
    hipEvent_t consumed_ev = nullptr;  // (stage contexts) this stage is done with the pass whose input it was handed

    void * abi_streamer = nullptr;   // copy streams + download thread of the streamed rwkv_eval (engine.hip), created on first use

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t mega_done = nullptr;   // completion of this context's latest persistent-kernel launch (engine.hip: launches are chained per device)
    uint32_t * ntok_out = nullptr;        // (runner.cpp) where a launch that folds the argmax leaves the chosen token instead of d_tokens / d_next_token:
                                          // the first stage's token word, through the peer mapping (the token feedback without a copy)
    hipEvent_t chain_covered = nullptr;   // (runner.cpp) an event this context's stream already waits on for the coming step: when it is the device's latest
                                          // persistent launch, the per-device chain does not wait on it a second time
    std::string persist_note;         // why this context runs the single-token path it runs (rwkv_mi_persist_info: chosen kernel, calibration, fall-backs)

    // fused single-token path (fused_v6.hip) when the model qualifies
    bool  fused_v6 = false;
    bool  fused_v7 = false;   // fused RWKV-7 layer (fused_v7.hip)
    bool  fused_v4 = false;   // fused RWKV-4 / RWKV-5 layer (fused_v4.hip)
    void * fused_scratch = nullptr;
    // persistent whole-stage decode kernel (mega_v6.hip) when the model and the device qualify; takes precedence
    void * mega = nullptr;

    // Live per-launch timing of the dominant kernel (the quantised single-token projection) with HIP events on this
    // context's stream; filled by rwkv_mi_profile_decode, used by bench.py's roofline figure.
    struct Prof {
        bool on = false;
        std::vector<hipEvent_t> events;  // pairs
        std::vector<uint64_t> bytes;     // per pair
        size_t used = 0;
        double total_ms = 0.0;
        uint64_t launches = 0, total_bytes = 0;
    } prof;
};

namespace rwkvmi {

rwkv_context * create_context(Model * m, uint32_t n_threads);
void destroy_context(rwkv_context * ctx);

extern std::atomic<int> g_test_fail_state_init;   // test hook (rwkv_mi_test_fail_state_init): the next n state initialisations fail
// state upload / download / init on the device-resident state
bool state_from_host(rwkv_context * ctx, const float * state_in /* NULL = fresh */);
bool state_to_host(rwkv_context * ctx, float * state_out);

// Runs T tokens (already in ctx->d_tokens) through the layers of this stage. Reads state[cur], writes state[cur^1], flips cur.
// x_in / x_out: residual stream hand-off for pipeline stages (nullptr on a full model). Logits land in ctx->d_logits.
bool forward(rwkv_context * ctx, int64_t T, bool want_logits);

// fused RWKV-6 decode layer (fused_v6.hip)
bool   fused_v6_supported(const Model & m);
size_t fused_v6_scratch_bytes(const Model & m);
void   fused_v6_layer(const Model & m, const LayerW & L, float * x, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf);
// fused RWKV-7 decode layer (fused_v7.hip): five launches per layer
bool   fused_v7_supported(const Model & m);
size_t fused_v7_scratch_bytes(const Model & m);
void   fused_v7_layer(const Model & m, const LayerW & L, int layer, float * x, float * v_first, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf);
// fused RWKV-4 decode layer (fused_v7.hip): four launches per layer
bool   fused_v4_supported(const Model & m);
size_t fused_v4_scratch_bytes(const Model & m);
void   fused_v4_layer(const Model & m, const LayerW & L, float * x, const float * sin, float * sout, void * scratch, hipStream_t st, rwkv_context::Prof * pf);
// persistent single-launch RWKV-6 decode over all layers of the stage (mega_v6.hip)
void *   mega_v6_create(const Model & m);   // nullptr: not applicable. RWKV_MI_PERSIST = ring | regs names the kernel (default: ring, else regs)
void *   mega_v6_create_kind(const Model & m, int kind);   // 1: register prefetch, 2: LDS-DMA weight ring
void     mega_v6_destroy(void * h);
// logits != nullptr and mega_v6_folds_head(h): ln_out + the head projection run inside the launch (the caller skips its own)
void     mega_v6_forward(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits = nullptr, float * v_first = nullptr,
                         const uint32_t * tok = nullptr, uint32_t * next_tok = nullptr);
// a layer range [l0, l1) of the stage (ring_v6.hip and persist_v47.hip have one; indices into the stage's own layers)
void     mega_v6_forward_range(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, int l0, int l1, float * v_first = nullptr,
                               const uint32_t * tok = nullptr, uint32_t * next_tok = nullptr);
bool     mega_v6_has_range(void * h);
bool     mega_v6_folds_embed(void * h);      // the launch starts from the token id: the caller skips its embedding + ln0 launch
bool     mega_v6_folds_argmax(void * h);     // a launch that produces logits also writes their argmax to next_tok
bool     mega_v6_set_history(void * h, uint32_t * hist, size_t n, hipStream_t st);   // (folds_argmax) tokens appended on the device (at most n), no copy per token
bool     mega_v6_folds_head(void * h);
bool     mega_v6_set_x_out(void * h, float * x_out);        // the stage's last layer leaves x there (pipeline hops without a copy); nullptr: in place
bool     mega_v6_ctl_fetch(void * h, hipStream_t st);       // async copy of the control words into the pinned mirror
bool     mega_v6_aborted_cached(void * h);                  // the mirror's abort word (valid after the stream was synchronised)
bool     mega_v6_aborted(void * h, hipStream_t st);         // fetch + synchronise + check
bool     mega_v6_clear_abort(void * h, hipStream_t st);
bool     mega_v6_force_abort(void * h, hipStream_t st);   // test hook: sets the abort word as a timed-out poll would
unsigned * mega_v6_ctl(void * h);
unsigned * p47_ctl(void * h);
unsigned * ring_v6_ctl(void * h);
const char * persist_unavailable_reason(const Model & m);   // nullptr: a persistent kernel exists for this model on this device
bool     mega_v6_set_tag(void * h, unsigned base, hipStream_t st);
unsigned mega_v6_generation(void * h, hipStream_t st);   // the hand-over generation the next launch starts from
uint64_t mega_v6_bytes(void * h);
bool     mega_v6_trace(void * h, int layer, long long * out, bool fetch);
int      mega_v6_kind(void * h);            // 1: register prefetch (mega_v6.hip), 2: LDS-DMA weight ring (ring_v6.hip), 3: RWKV-4 / RWKV-7 (persist_v47.hip)
// persistent single-launch RWKV-4 / RWKV-7 decode (persist_v47.hip); reached through the mega_v6_* entry points
void *   p47_create(const Model & m);
void     p47_destroy(void * h);
void     p47_forward_range(void * h, float * x, float * v_first, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, int l0, int l1,
                           float * logits = nullptr, const uint32_t * tok = nullptr, uint32_t * next_tok = nullptr);
bool     p47_folds_embed(void * h);
bool     p47_folds_head(void * h);
bool     p47_set_history(void * h, uint32_t * hist, size_t n, hipStream_t st);
void     p47_set_x_out(void * h, float * x_out);
int      p47_layers(void * h);
bool     p47_ctl_fetch(void * h, hipStream_t st);
bool     p47_aborted_cached(void * h);
unsigned p47_generation_cached(void * h);
bool     p47_clear_abort(void * h, hipStream_t st);
bool     p47_set_tag(void * h, unsigned base, hipStream_t st);
uint64_t p47_bytes(void * h);
bool     p47_trace(void * h, int layer, long long * out, bool fetch);
// the same persistent launch on the LDS-DMA weight ring (ring_v6.hip); reached through the mega_v6_* entry points
void *   ring_v6_create(const Model & m);
void     ring_v6_destroy(void * h);
void     ring_v6_forward(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, const uint32_t * tok = nullptr, uint32_t * next_tok = nullptr);
void     ring_v6_forward_range(void * h, float * x, const float * sin, float * sout, hipStream_t st, rwkv_context::Prof * pf, float * logits, int l0, int l1,
                               const uint32_t * tok = nullptr, uint32_t * next_tok = nullptr);
bool     ring_v6_folds_head(void * h);
bool     ring_v6_folds_embed(void * h);
bool     ring_v6_folds_argmax(void * h);
bool     ring_v6_set_history(void * h, uint32_t * hist, size_t n, hipStream_t st);
void     ring_v6_set_x_out(void * h, float * x_out);
bool     ring_v6_ctl_fetch(void * h, hipStream_t st);
bool     ring_v6_aborted_cached(void * h);
unsigned ring_v6_generation_cached(void * h);
bool     ring_v6_clear_abort(void * h, hipStream_t st);
bool     ring_v6_set_tag(void * h, unsigned base, hipStream_t st);
uint64_t ring_v6_bytes(void * h);
bool     ring_v6_trace(void * h, int layer, long long * out, bool fetch);
// ---- layer pipeline in one process (pipeline.cpp) ----
bool upload_tokens_for(rwkv_context * ctx, const uint32_t * tokens, size_t n);   // api.cpp: pinned staging + async copy into ctx->d_tokens
rwkv_context * pipeline_create(const char * path, uint32_t n_threads, const char * devices);
void pipeline_destroy(rwkv_context * front);
bool pipeline_eval(rwkv_context * front, const uint32_t * tokens, size_t n, size_t chunk, const float * state_in, float * state_out, float * logits_out);
rwkv_context * pipeline_clone(rwkv_context * front, uint32_t n_threads);
// runner.cpp: resident state of a chain; greedy decode of n_streams contexts (chains of the same stages, or one-device contexts) interleaved
bool pipeline_state_load(rwkv_context * front, const float * state_in);
bool pipeline_state_store(rwkv_context * front, float * state_out);
bool pipeline_decode_greedy(rwkv_context * const * fronts, size_t n_streams, const uint32_t * first_tokens, size_t n_tokens, uint32_t * tokens_out, float * elapsed_ms);
// after a poll time-out of the persistent kernel: drain, clear, drop the persistent path (see engine.hip)
void recover_from_abort(rwkv_context * ctx);
// rwkv_eval with the caller's state streamed group by group under the layers (engine.hip)
bool forward_streamed_eligible(const rwkv_context * ctx);
bool forward_streamed(rwkv_context * ctx, bool want_logits, const float * h_in, float * h_out, float * h_logits, bool * aborted);
void abi_streamer_free(void * p);
// single-token forward through the captured hipGraph (falls back to forward() when capture is disabled)
bool forward_decode(rwkv_context * ctx, bool want_logits);
bool single_launch_step(const rwkv_context * ctx, bool want_logits);
hipEvent_t mega_chain_marker(rwkv_context * ctx);   // the event recorded behind ctx's persistent launch if that launch is the latest of its device (else nullptr)   // the step is one directly issued persistent launch (no graph replay)
// grows the per-context activation scratch to hold T tokens
bool ensure_scratch(rwkv_context * ctx, int64_t T);
uint32_t * folded_argmax_target(const rwkv_context * ctx);
// hand-off buffer size in floats: D, or 2 D for RWKV-7 (x and v_first travel together)
int64_t handoff_len(const Model & m);

}  // namespace rwkvmi
