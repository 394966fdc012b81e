// model.hip -- model file -> HBM. Restates what the reference loader decides (rwkv_model_loading.inc:288-419):
// two passes over the file, architecture detection by tensor names (:319-340), the per-architecture parameter table
// (:132-282), head_count/head_size derivation (:403-409), the embedding shape check (:411-416).
// What is new: every parameter goes to one HBM arena; quantised tensors are re-packed into aligned planes on the GPU.
#include "model.h"

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cinttypes>
#include <condition_variable>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace rwkvmi {

#define HIP_OK_OR(RET, FLAGS, CALL) \
    do { hipError_t e_ = (CALL); RW_CHECK((FLAGS), RET, e_ == hipSuccess, "HIP error: %s", hipGetErrorString(e_)); } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

namespace {

struct PlaneSizes { size_t data = 0, qs = 0, qh = 0, sc = 0; };

PlaneSizes plane_sizes(const TensorInfo & t) {
    PlaneSizes p;
    if (!dtype_quantized(t.type)) { p.data = align_up(t.nbytes, 256); return p; }
    const size_t nblk = (size_t) t.nelements() / 32;
    p.qs = align_up(nblk * (t.type == T_Q8_0 ? 32 : 16), 256);
    if (t.type == T_Q5_0 || t.type == T_Q5_1) p.qh = align_up(nblk * 4, 256);
    p.sc = align_up(nblk * ((t.type == T_Q4_1 || t.type == T_Q5_1) ? 4 : 2), 256);
    return p;
}

bool layer_of(const std::string & name, uint32_t & layer) {
    if (name.compare(0, 7, "blocks.") != 0) return false;
    layer = (uint32_t) strtoul(name.c_str() + 7, nullptr, 10);
    return true;
}

}  // namespace

static bool bind_params(Model & m) {
    bool ok = true;
    auto get = [&](const std::string & key) -> const DevTensor * {
        auto it = m.by_name.find(key);
        if (it == m.by_name.end()) {
            global_fail(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_PARAM_MISSING, __FILE__, __LINE__, "parameter present", "Model parameter %s not found", key.c_str());
            ok = false;
            return nullptr;
        }
        return it->second;
    };
    if (m.has_embed) {
        m.emb = get("emb.weight");
        m.ln0_w = get("blocks.0.ln0.weight");
        m.ln0_b = get("blocks.0.ln0.bias");
    }
    if (m.has_head) {
        m.ln_out_w = get("ln_out.weight");
        m.ln_out_b = get("ln_out.bias");
        m.head = get("head.weight");
    }
    m.layers.assign(m.header.n_layer, LayerW{});
    for (uint32_t i = m.layer_begin; i < m.layer_end && ok; i++) {
        LayerW & L = m.layers[i];
        const std::string p = "blocks." + std::to_string(i) + ".";
        L.ln1_w = get(p + "ln1.weight"); L.ln1_b = get(p + "ln1.bias");
        L.ln2_w = get(p + "ln2.weight"); L.ln2_b = get(p + "ln2.bias");
        L.att_key = get(p + "att.key.weight"); L.att_value = get(p + "att.value.weight");
        L.att_receptance = get(p + "att.receptance.weight"); L.att_output = get(p + "att.output.weight");
        L.ffn_key = get(p + "ffn.key.weight"); L.ffn_value = get(p + "ffn.value.weight");
        if (m.arch_major != 7) L.ffn_receptance = get(p + "ffn.receptance.weight");
        switch (m.arch_major) {
            case 4:
                L.att_time_mix_k = get(p + "att.time_mix_k"); L.att_time_mix_v = get(p + "att.time_mix_v"); L.att_time_mix_r = get(p + "att.time_mix_r");
                L.att_time_first = get(p + "att.time_first"); L.att_time_decay = get(p + "att.time_decay");
                L.ffn_time_mix_k = get(p + "ffn.time_mix_k"); L.ffn_time_mix_r = get(p + "ffn.time_mix_r");
                break;
            case 5:
                L.att_time_mix_k = get(p + "att.time_mix_k"); L.att_time_mix_v = get(p + "att.time_mix_v"); L.att_time_mix_r = get(p + "att.time_mix_r");
                if (m.arch_minor >= 2) {
                    L.att_time_faaaa = get(p + "att.time_faaaa"); L.att_time_mix_g = get(p + "att.time_mix_g"); L.att_gate = get(p + "att.gate.weight");
                } else {
                    L.att_time_first = get(p + "att.time_first");
                }
                L.att_time_decay = get(p + "att.time_decay");
                L.att_ln_x_w = get(p + "att.ln_x.weight"); L.att_ln_x_b = get(p + "att.ln_x.bias");
                L.ffn_time_mix_k = get(p + "ffn.time_mix_k"); L.ffn_time_mix_r = get(p + "ffn.time_mix_r");
                break;
            case 6:
                L.att_time_maa_x = get(p + "att.time_maa_x"); L.att_time_maa_w = get(p + "att.time_maa_w"); L.att_time_maa_k = get(p + "att.time_maa_k");
                L.att_time_maa_v = get(p + "att.time_maa_v"); L.att_time_maa_r = get(p + "att.time_maa_r"); L.att_time_maa_g = get(p + "att.time_maa_g");
                L.att_time_maa_w1 = get(p + "att.time_maa_w1"); L.att_time_maa_w2 = get(p + "att.time_maa_w2");
                L.att_time_faaaa = get(p + "att.time_faaaa"); L.att_time_decay = get(p + "att.time_decay");
                L.att_time_decay_w1 = get(p + "att.time_decay_w1"); L.att_time_decay_w2 = get(p + "att.time_decay_w2");
                L.att_gate = get(p + "att.gate.weight");
                L.att_ln_x_w = get(p + "att.ln_x.weight"); L.att_ln_x_b = get(p + "att.ln_x.bias");
                L.ffn_time_maa_k = get(p + "ffn.time_maa_k"); L.ffn_time_maa_r = get(p + "ffn.time_maa_r");
                break;
            case 7:
                L.att_x_rwkvag = get(p + "att.x_rwkvag");
                L.att_w0 = get(p + "att.w0"); L.att_w1 = get(p + "att.w1"); L.att_w2 = get(p + "att.w2");
                L.att_a0 = get(p + "att.a0"); L.att_a1 = get(p + "att.a1"); L.att_a2 = get(p + "att.a2");
                L.att_g1 = get(p + "att.g1"); L.att_g2 = get(p + "att.g2");
                if (i != 0) { L.att_v0 = get(p + "att.v0"); L.att_v1 = get(p + "att.v1"); L.att_v2 = get(p + "att.v2"); }
                L.att_r_k = get(p + "att.r_k"); L.att_k_k = get(p + "att.k_k"); L.att_k_a = get(p + "att.k_a");
                L.att_ln_x_w = get(p + "att.ln_x.weight"); L.att_ln_x_b = get(p + "att.ln_x.bias");
                L.ffn_x_k = get(p + "ffn.x_k");
                break;
            default: break;
        }
    }
    return ok;
}

// Which tensors must be f32 vectors (consumed by elementwise kernels), which may be matrices of any dtype.
static bool validate_shapes(Model & m) {
    const int64_t D = m.n_embed();
    auto is_matrix = [](const DevTensor * t) { return t != nullptr; };
    (void) is_matrix;
    bool ok = true;
    auto vec_f32 = [&](const DevTensor * t, int64_t n, const char * what) {
        if (!t) return;
        if (t->type != T_F32 || t->ne[0] * t->ne[1] * t->ne[2] != n) {
            global_fail(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_SHAPE, __FILE__, __LINE__, "vector parameter is FP32 with the expected length",
                        "Parameter %s (%s) must be FP32 with %" PRId64 " elements", t->name.c_str(), what, n);
            ok = false;
        }
    };
    auto mat = [&](const DevTensor * t, int64_t K, int64_t N, const char * what) {
        if (!t) return;
        // Every projection kernel of this library walks a row in 32-element steps (quantised blocks; ggml's 32-partial dot order
        // for F32/F16). The reference accepts other lengths for F32/F16 matrices; this build does not and says so.
        if (t->ne[0] % 32 != 0) {
            global_fail(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_UNSUPPORTED, __FILE__, __LINE__, "matrix rows are a multiple of 32 elements",
                        "Parameter %s (%s) has rows of %" PRId64 " elements; librwkv.so for MI355X supports row lengths (n_embed, ffn size, low-rank sizes) "
                        "that are multiples of 32 only", t->name.c_str(), what, t->ne[0]);
            ok = false;
            return;
        }
        if ((K > 0 && t->ne[0] != K) || (N > 0 && t->ne[1] * t->ne[2] != N)) {
            global_fail(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_DIMENSION, __FILE__, __LINE__, "matrix parameter has the expected shape",
                        "Parameter %s (%s) has unexpected shape [%" PRId64 ", %" PRId64 ", %" PRId64 "]", t->name.c_str(), what, t->ne[0], t->ne[1], t->ne[2]);
            ok = false;
        }
    };
    if (m.has_embed) { vec_f32(m.ln0_w, D, "ln0"); vec_f32(m.ln0_b, D, "ln0"); }
    if (m.has_head) { vec_f32(m.ln_out_w, D, "ln_out"); vec_f32(m.ln_out_b, D, "ln_out"); mat(m.head, D, m.n_vocab(), "head"); }
    const int64_t H = m.head_count, S = m.head_size;
    for (uint32_t i = m.layer_begin; i < m.layer_end; i++) {
        const LayerW & L = m.layers[i];
        const int64_t F = L.ffn_key ? L.ffn_key->ne[1] : 0;
        vec_f32(L.ln1_w, D, "ln1"); vec_f32(L.ln1_b, D, "ln1"); vec_f32(L.ln2_w, D, "ln2"); vec_f32(L.ln2_b, D, "ln2");
        mat(L.att_key, D, D, "att.key"); mat(L.att_value, D, D, "att.value"); mat(L.att_receptance, D, D, "att.receptance");
        mat(L.att_output, D, D, "att.output"); mat(L.att_gate, D, D, "att.gate");
        mat(L.ffn_key, D, F, "ffn.key"); mat(L.ffn_value, F, D, "ffn.value"); mat(L.ffn_receptance, D, D, "ffn.receptance");
        vec_f32(L.att_time_mix_k, D, "mix"); vec_f32(L.att_time_mix_v, D, "mix"); vec_f32(L.att_time_mix_r, D, "mix"); vec_f32(L.att_time_mix_g, D, "mix");
        vec_f32(L.ffn_time_mix_k, D, "mix"); vec_f32(L.ffn_time_mix_r, D, "mix");
        vec_f32(L.att_ln_x_w, D, "ln_x"); vec_f32(L.att_ln_x_b, D, "ln_x");
        if (m.arch_major == 4) { vec_f32(L.att_time_first, D, "time_first"); vec_f32(L.att_time_decay, D, "time_decay"); }
        if (m.arch_major == 5) {
            if (m.arch_minor >= 2) { vec_f32(L.att_time_faaaa, D, "time_faaaa"); vec_f32(L.att_time_decay, D, "time_decay"); }
            else { vec_f32(L.att_time_first, H, "time_first"); vec_f32(L.att_time_decay, H, "time_decay"); }
        }
        if (m.arch_major == 6) {
            vec_f32(L.att_time_maa_x, D, "maa"); vec_f32(L.att_time_maa_w, D, "maa"); vec_f32(L.att_time_maa_k, D, "maa"); vec_f32(L.att_time_maa_v, D, "maa");
            vec_f32(L.att_time_maa_r, D, "maa"); vec_f32(L.att_time_maa_g, D, "maa"); vec_f32(L.ffn_time_maa_k, D, "maa"); vec_f32(L.ffn_time_maa_r, D, "maa");
            vec_f32(L.att_time_faaaa, D, "time_faaaa"); vec_f32(L.att_time_decay, D, "time_decay");
            mat(L.att_time_maa_w1, D, 0, "time_maa_w1"); mat(L.att_time_decay_w1, D, 0, "time_decay_w1");
            if (L.att_time_maa_w1 && L.att_time_maa_w2) {
                const int64_t R5 = L.att_time_maa_w1->ne[1];
                const DevTensor * w2 = L.att_time_maa_w2;
                if (R5 % 5 != 0 || w2->type != T_F32 || w2->ne[0] != R5 / 5 || w2->ne[1] != D || w2->ne[2] != 5 || (R5 / 5) % 4 != 0) {
                    global_fail(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_SHAPE, __FILE__, __LINE__, "time_maa_w2 is FP32 [r, D, 5]", "Parameter %s has an unexpected shape or type", w2->name.c_str());
                    ok = false;
                }
            }
            if (L.att_time_decay_w1 && L.att_time_decay_w2) mat(L.att_time_decay_w2, L.att_time_decay_w1->ne[1], D, "time_decay_w2");
        }
        if (m.arch_major == 7) {
            vec_f32(L.att_x_rwkvag, 6 * D, "x_rwkvag"); vec_f32(L.att_w0, D, "w0"); vec_f32(L.att_a0, D, "a0"); vec_f32(L.att_v0, D, "v0");
            vec_f32(L.att_r_k, D, "r_k"); vec_f32(L.att_k_k, D, "k_k"); vec_f32(L.att_k_a, D, "k_a"); vec_f32(L.ffn_x_k, D, "x_k");
            mat(L.att_w1, D, 0, "w1"); mat(L.att_a1, D, 0, "a1"); mat(L.att_g1, D, 0, "g1"); mat(L.att_v1, D, 0, "v1");
            if (L.att_w1) mat(L.att_w2, L.att_w1->ne[1], D, "w2");
            if (L.att_a1) mat(L.att_a2, L.att_a1->ne[1], D, "a2");
            if (L.att_g1) mat(L.att_g2, L.att_g1->ne[1], D, "g2");
            if (L.att_v1) mat(L.att_v2, L.att_v1->ne[1], D, "v2");
        }
        (void) S;
    }
    return ok;
}

// Bytes a decoded token streams per layer / for the head, from the file's tensor directory alone (used to balance pipeline stages).
bool scan_stage_costs(const char * path, std::vector<uint64_t> & per_layer, uint64_t & head_bytes) {
    FILE * f = fopen(path, "rb");
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_OPEN, false, f != nullptr, "Failed to open file %s", path);
    std::unique_ptr<FILE, int (*)(FILE *)> fguard(f, fclose);
    struct stat st;
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_STAT, false, fstat(fileno(f), &st) == 0, "Failed to stat file %s", path);
    FileHeader h;
    if (!read_file_header(f, h)) { global_fail(RWKV_ERROR_FILE, __FILE__, __LINE__, "read_file_header", "Invalid file header"); return false; }
    per_layer.assign(h.n_layer, 0);
    head_bytes = 0;
    while ((uint64_t) ftello(f) < (uint64_t) st.st_size) {
        TensorInfo ti;
        if (!read_tensor_info(f, ti)) { global_fail(RWKV_ERROR_MODEL_PARAMS, __FILE__, __LINE__, "read_tensor_info", "Failed to read a model parameter"); return false; }
        uint32_t li;
        if (layer_of(ti.name, li)) { if (li < h.n_layer) per_layer[li] += ti.nbytes; }
        else if (ti.name != "emb.weight") head_bytes += ti.nbytes;
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_FILE_READ, false, fseeko(f, (off_t) ti.nbytes, SEEK_CUR) == 0, "Failed to seek past parameter %s", ti.name.c_str());
    }
    return true;
}

Model * load_model(const char * path, uint32_t layer_begin, uint32_t layer_end) {
    FILE * f = fopen(path, "rb");
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_OPEN, nullptr, f != nullptr, "Failed to open file %s", path);
    std::unique_ptr<FILE, int (*)(FILE *)> fguard(f, fclose);
    struct stat st;
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_STAT, nullptr, fstat(fileno(f), &st) == 0, "Failed to stat file %s", path);
    const uint64_t file_size = (uint64_t) st.st_size;

    std::unique_ptr<Model> m(new Model());
    if (!read_file_header(f, m->header)) {
        global_fail(RWKV_ERROR_FILE, __FILE__, __LINE__, "read_file_header", "Invalid file header");
        return nullptr;
    }

    // No CPU path exists in this library: without a gfx950 device the load fails loudly.
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    RW_CHECK(RWKV_ERROR_CTX | RWKV_ERROR_UNSUPPORTED, nullptr, e == hipSuccess && n_dev > 0,
             "No HIP device is visible: this library has no CPU path (%s)", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    int device = 0;
    HIP_OK_OR(nullptr, RWKV_ERROR_CTX, hipGetDevice(&device));
    hipDeviceProp_t prop;
    HIP_OK_OR(nullptr, RWKV_ERROR_CTX, hipGetDeviceProperties(&prop, device));
    RW_CHECK(RWKV_ERROR_CTX | RWKV_ERROR_UNSUPPORTED, nullptr, strncmp(prop.gcnArchName, "gfx950", 6) == 0,
             "Device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);

    m->device = device;

    // pass 1: tensor directory
    std::vector<TensorInfo> infos;
    while ((uint64_t) ftello(f) < file_size) {
        TensorInfo ti;
        if (!read_tensor_info(f, ti)) {
            global_fail(RWKV_ERROR_MODEL_PARAMS, __FILE__, __LINE__, "read_tensor_info", "Failed to read a model parameter");
            return nullptr;
        }
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_FILE_READ, nullptr, ti.file_offset + ti.nbytes <= file_size,
                 "Parameter %s is truncated", ti.name.c_str());
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_DIMENSION, nullptr, !dtype_quantized(ti.type) || ti.ne[0] % 32 == 0,
                 "Quantized parameter %s has a row length that is not a multiple of 32", ti.name.c_str());
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_FILE_READ, nullptr, fseeko(f, (off_t) ti.nbytes, SEEK_CUR) == 0,
                 "Failed to seek to next tensor after parameter %s", ti.name.c_str());
        infos.push_back(std::move(ti));
    }

    auto has = [&](const char * key) { for (const TensorInfo & t : infos) if (t.name == key) return true; return false; };
    m->arch_major = 4; m->arch_minor = 0;
    if (has("blocks.0.att.ln_x.weight")) { m->arch_major = 5; m->arch_minor = has("blocks.0.att.gate.weight") ? 2 : 1; }
    if (has("blocks.0.att.time_maa_x")) { m->arch_major = 6; m->arch_minor = 0; }
    if (has("blocks.0.att.r_k")) { m->arch_major = 7; m->arch_minor = 0; }

    const uint32_t L = m->header.n_layer;
    m->layer_begin = layer_begin > L ? L : layer_begin;
    m->layer_end = layer_end > L ? L : layer_end;
    RW_CHECK(RWKV_ERROR_ARGS, nullptr, m->layer_begin < m->layer_end, "Empty layer range [%u, %u)", layer_begin, layer_end);
    m->has_embed = m->layer_begin == 0;
    m->has_head = m->layer_end == L;

    // which tensors this stage owns
    auto wanted = [&](const TensorInfo & t) {
        uint32_t li;
        if (layer_of(t.name, li)) {
            if (t.name.find(".ln0.") != std::string::npos) return m->has_embed;
            return li >= m->layer_begin && li < m->layer_end;
        }
        if (t.name == "emb.weight") return m->has_embed;
        return m->has_head;  // ln_out.*, head.weight
    };

    size_t total = 0, max_raw = 0;
    for (const TensorInfo & t : infos) {
        if (!wanted(t)) continue;
        const PlaneSizes p = plane_sizes(t);
        total += p.data + p.qs + p.qh + p.sc;
        const bool staged = dtype_quantized(t.type) || (t.type == T_F32 && t.ndim == 3 && t.name.find("att.time_maa_w2") != std::string::npos);
        if (staged && t.nbytes > max_raw) max_raw = (size_t) t.nbytes;
    }
    HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_ALLOC, hipMalloc(&m->arena, total > 0 ? total : 256));
    m->arena_bytes = total;
    struct ArenaGuard { Model * m; bool armed = true; ~ArenaGuard() { if (armed && m->arena) { (void) hipFree(m->arena); m->arena = nullptr; } } } aguard{m.get()};

    // pass 2: payloads (the reference does one malloc / fread / ggml_backend_tensor_set per tensor,
    // rwkv_file_format.inc:302-313). The wanted tensors are cut into pieces of at most k_piece bytes; a small pool of reader threads
    // pread()s the pieces, in file order, into a ring of PINNED host buffers; this thread enqueues, per piece and in order, the
    // host-to-device copy (a real DMA from pinned memory, asynchronous -- a copy out of a pageable mapping is staged synchronously inside
    // the runtime) and, for quantised payloads and the RWKV-6 mix matrix, the re-pack / transpose kernel out of a device staging
    // buffer. Reads, copies and kernels of different pieces overlap; a slot is refilled when the event behind its last use has completed.
    const auto t_load0 = std::chrono::steady_clock::now();
    constexpr size_t k_piece = (size_t) 32 << 20;
    constexpr int k_slots = 4, k_readers = 4;
    struct Piece { const TensorInfo * t; DevTensor * dt; uint64_t off, bytes; int64_t blk0, nblk; int kind; };   // kind 0 plain, 1 quantised, 2 w2
    std::vector<Piece> pieces;
    uint8_t * cursor = (uint8_t *) m->arena;
    for (const TensorInfo & t : infos) {
        if (!wanted(t)) continue;
        std::unique_ptr<DevTensor> dt(new DevTensor());
        dt->name = t.name; dt->type = t.type; dt->ndim = t.ndim; dt->nbytes = t.nbytes;
        for (int i = 0; i < 3; i++) dt->ne[i] = t.ne[i];
        const PlaneSizes p = plane_sizes(t);
        const bool w2 = t.type == T_F32 && t.ndim == 3 && t.name.find("att.time_maa_w2") != std::string::npos && t.ne[2] == 5;
        if (!dtype_quantized(t.type)) {
            dt->data = cursor; cursor += p.data;
            if (w2) {
                // v6 mix matrix: stored transposed ([5][R][D]) so that the mix kernels read it coalesced -- one piece
                RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_UNSUPPORTED, nullptr, t.nbytes <= k_piece, "Parameter %s is too large for the load pipeline", t.name.c_str());
                pieces.push_back(Piece{&t, dt.get(), 0, t.nbytes, 0, 0, 2});
            } else {
                for (uint64_t o = 0; o < t.nbytes; o += k_piece) pieces.push_back(Piece{&t, dt.get(), o, t.nbytes - o < k_piece ? t.nbytes - o : k_piece, 0, 0, 0});
            }
        } else {
            dt->qs = cursor; cursor += p.qs;
            if (p.qh) { dt->qh = (uint32_t *) cursor; cursor += p.qh; }
            dt->sc = cursor; cursor += p.sc;
            const uint64_t bsz = dtype_block_bytes(t.type);
            const int64_t nblk = t.nelements() / 32, per = (int64_t) (k_piece / bsz);
            for (int64_t b0 = 0; b0 < nblk; b0 += per) {
                const int64_t nb = nblk - b0 < per ? nblk - b0 : per;
                pieces.push_back(Piece{&t, dt.get(), (uint64_t) b0 * bsz, (uint64_t) nb * bsz, b0, nb, 1});
            }
        }
        m->weight_bytes += t.nbytes;
        m->by_name[dt->name] = dt.get();
        m->tensors.push_back(std::move(dt));
    }

    struct Pipe {
        int fd = -1;
        void * h[k_slots] = {}; void * d[k_slots] = {}; hipEvent_t ev[k_slots] = {};
        hipStream_t st = nullptr;
        std::vector<std::thread> th;
        std::mutex mu; std::condition_variable cv;
        std::vector<int> state;          // per piece: 0 not read, 1 read (slot filled), 2 enqueued (event recorded), -1 read error
        std::atomic<size_t> next{0};
        bool stop = false;
        ~Pipe() {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            for (std::thread & t : th) if (t.joinable()) t.join();
            if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
            for (int i = 0; i < k_slots; i++) { if (h[i]) (void) hipHostFree(h[i]); if (d[i]) (void) hipFree(d[i]); if (ev[i]) (void) hipEventDestroy(ev[i]); }
            if (fd >= 0) close(fd);
        }
    } pp;
    pp.fd = open(path, O_RDONLY);
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_OPEN, nullptr, pp.fd >= 0, "Failed to open file %s", path);
    (void) posix_fadvise(pp.fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    HIP_OK_OR(nullptr, RWKV_ERROR_CTX, hipStreamCreateWithFlags(&pp.st, hipStreamNonBlocking));
    bool any_staged = false;
    for (const Piece & pc : pieces) any_staged = any_staged || pc.kind != 0;
    for (int i = 0; i < k_slots && !pieces.empty(); i++) {
        HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_ALLOC, hipHostMalloc(&pp.h[i], k_piece, hipHostMallocDefault));
        if (any_staged) HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_ALLOC, hipMalloc(&pp.d[i], k_piece));
        HIP_OK_OR(nullptr, RWKV_ERROR_CTX, hipEventCreateWithFlags(&pp.ev[i], hipEventDisableTiming));
    }
    pp.state.assign(pieces.size(), 0);
    auto reader = [&pp, &pieces, device]() {
        (void) hipSetDevice(device);
        for (;;) {
            const size_t i = pp.next.fetch_add(1);
            if (i >= pieces.size()) return;
            const int slot = (int) (i % k_slots);
            if (i >= (size_t) k_slots) {
                // the slot's previous piece must have been enqueued (event recorded) and its copy / kernel finished
                std::unique_lock<std::mutex> lk(pp.mu);
                pp.cv.wait(lk, [&] { return pp.stop || pp.state[i - k_slots] == 2; });
                if (pp.stop) return;
                lk.unlock();
                if (hipEventSynchronize(pp.ev[slot]) != hipSuccess) { std::lock_guard<std::mutex> g(pp.mu); pp.state[i] = -1; pp.cv.notify_all(); return; }
            }
            const Piece & pc = pieces[i];
            uint64_t done = 0;
            bool ok = true;
            while (done < pc.bytes) {
                const ssize_t r = pread(pp.fd, (uint8_t *) pp.h[slot] + done, pc.bytes - done, (off_t) (pc.t->file_offset + pc.off + done));
                if (r <= 0) { if (r < 0 && errno == EINTR) continue; ok = false; break; }
                done += (uint64_t) r;
            }
            { std::lock_guard<std::mutex> g(pp.mu); pp.state[i] = ok ? 1 : -1; }
            pp.cv.notify_all();
            if (!ok) return;
        }
    };
    for (int i = 0; i < k_readers && (size_t) i < pieces.size(); i++) pp.th.emplace_back(reader);
    for (size_t i = 0; i < pieces.size(); i++) {
        const Piece & pc = pieces[i];
        const int slot = (int) (i % k_slots);
        {
            std::unique_lock<std::mutex> lk(pp.mu);
            pp.cv.wait(lk, [&] { return pp.state[i] != 0; });
            RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_READ, nullptr, pp.state[i] == 1, "Failed to read parameter %s from %s", pc.t->name.c_str(), path);
        }
        DevTensor * dt = pc.dt;
        if (pc.kind == 0) {
            HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_DATA, hipMemcpyAsync((uint8_t *) dt->data + pc.off, pp.h[slot], pc.bytes, hipMemcpyHostToDevice, pp.st));
        } else {
            HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_DATA, hipMemcpyAsync(pp.d[slot], pp.h[slot], pc.bytes, hipMemcpyHostToDevice, pp.st));
            if (pc.kind == 2) {
                launch_transpose_w2((const float *) pp.d[slot], (float *) dt->data, pc.t->ne[1], pc.t->ne[0], pp.st);
            } else {
                const int64_t qsb = pc.t->type == T_Q8_0 ? 32 : 16, scb = (pc.t->type == T_Q4_1 || pc.t->type == T_Q5_1) ? 4 : 2;
                launch_repack(pc.t->type, (const uint8_t *) pp.d[slot], pc.nblk, dt->qs + pc.blk0 * qsb, dt->qh ? dt->qh + pc.blk0 : nullptr,
                              (uint8_t *) dt->sc + pc.blk0 * scb, pp.st);
            }
        }
        HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_DATA, hipEventRecord(pp.ev[slot], pp.st));
        { std::lock_guard<std::mutex> g(pp.mu); pp.state[i] = 2; }
        pp.cv.notify_all();
    }
    HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_DATA, hipStreamSynchronize(pp.st));
    HIP_OK_OR(nullptr, RWKV_ERROR_MODEL | RWKV_ERROR_DATA, hipGetLastError());
    m->load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load0).count();

    if (!bind_params(*m)) return nullptr;

    // head geometry (rwkv_model_loading.inc:403-409)
    {
        const TensorInfo * ref = nullptr;
        const char * key = m->arch_major == 7 ? "blocks.0.att.r_k" : "blocks.0.att.time_decay";
        for (const TensorInfo & t : infos) if (t.name == key) ref = &t;
        if (m->arch_major == 7) { RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_PARAM_MISSING, nullptr, ref, "%s", key); m->head_count = ref->ne[1]; }
        else if (m->arch_major >= 5) { RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_PARAM_MISSING, nullptr, ref, "%s", key); m->head_count = ref->ne[2]; }
        if (m->arch_major >= 5) {
            RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_SHAPE, nullptr, m->head_count > 0 && m->n_embed() % m->head_count == 0, "Bad head count %" PRId64, m->head_count);
            m->head_size = m->n_embed() / m->head_count;
        }
    }
    // embedding shape (rwkv_model_loading.inc:411-416) -- checked from the directory so that every stage verifies it
    for (const TensorInfo & t : infos) {
        if (t.name != "emb.weight") continue;
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_SHAPE, nullptr, t.ndim == 2, "Unexpected dimension count of embedding matrix %d", t.ndim);
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_DIMENSION, nullptr, t.ne[0] == m->header.n_embed, "Unexpected dimension of embedding matrix %" PRId64, t.ne[0]);
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_DIMENSION, nullptr, t.ne[1] == m->header.n_vocab, "Unexpected dimension of embedding matrix %" PRId64, t.ne[1]);
    }
    if (!validate_shapes(*m)) return nullptr;

    const LayerW & L0 = m->layers[m->layer_begin];
    m->ffn_size = L0.ffn_key->ne[1];
    for (uint32_t i = m->layer_begin; i < m->layer_end; i++) {
        const LayerW & Lw = m->layers[i];
        for (const DevTensor * t : {Lw.att_time_maa_w1, Lw.att_time_decay_w1, Lw.att_w1, Lw.att_a1, Lw.att_g1, Lw.att_v1})
            if (t && t->ne[1] > m->max_lowrank) m->max_lowrank = t->ne[1];
        if (Lw.ffn_key->ne[1] > m->ffn_size) m->ffn_size = Lw.ffn_key->ne[1];
    }

    // algorithmic bytes of one decoded token: every tensor of the stage once, except the embedding of which
    // one row is read; + state read and write; + logits written.
    {
        uint64_t b = m->weight_bytes;
        if (m->emb) b = b - m->emb->nbytes + (uint64_t) m->n_embed() * dtype_block_bytes(m->emb->type) / dtype_block_elems(m->emb->type);
        b += 2ull * (uint64_t) m->state_per_layer() * (m->layer_end - m->layer_begin) * 4ull;
        if (m->has_head) b += (uint64_t) m->n_vocab() * 4ull;
        m->bytes_per_token = b;
    }
    aguard.armed = false;
    return m.release();
}

void release_model(Model * m) {
    if (!m) return;
    if (--m->refcount > 0) return;
    for (const auto & t : m->tensors) free_pf(*t);
    if (m->arena) (void) hipFree(m->arena);
    delete m;
}

}  // namespace rwkvmi
