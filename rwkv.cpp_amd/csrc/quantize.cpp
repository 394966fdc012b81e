// quantize.cpp -- rwkv_quantize_model_file: FP32/FP16 model file -> Q4_0 | Q4_1 | Q5_0 | Q5_1 | Q8_0 file.
// File-to-file, host only. Follows the reference's selection rule (rwkv_quantize.inc:1-13,137-140: only 2-D FP32/FP16
// tensors, never emb/head/v7 low-rank/r_k) and ggml's reference block quantisers, so the output is
// byte-identical to the fixtures the reference ships (tests/tiny-rwkv-*-Q5_0.bin / -Q5_1.bin).
#include "common.h"
#include <cinttypes>

#include <cmath>
#include <cstring>
#include <sys/stat.h>

namespace rwkvmi {

static inline uint16_t f32_to_f16(float f) {
    // round-to-nearest-even conversion via the compiler's native half type
    const _Float16 h = (_Float16) f;
    uint16_t u; memcpy(&u, &h, 2); return u;
}
static inline float f16_to_f32(uint16_t u) {
    _Float16 h; memcpy(&h, &u, 2); return (float) h;
}

static bool tensor_needs_quant(const std::string & name) {
    static const char * const never[] = {"att.v1", "att.v2", "att.g1", "att.g2", "att.a1", "att.a2", "att.w1", "att.w2", "att.r_k"};
    if (name == "emb.weight" || name == "head.weight") return false;
    for (const char * n : never) if (name.find(n) != std::string::npos) return false;
    return true;
}

template <typename T> static inline T min_(T a, T b) { return a < b ? a : b; }

// One row of n (multiple of 32) floats -> blocks.
static void quantize_row(int type, const float * x, uint8_t * y, int64_t n) {
    const int64_t nb = n / 32;
    for (int64_t b = 0; b < nb; b++, x += 32) {
        switch (type) {
            case T_Q4_0: {
                float amax = 0.0f, mx = 0.0f;
                for (int j = 0; j < 32; j++) if (amax < fabsf(x[j])) { amax = fabsf(x[j]); mx = x[j]; }
                const float d = mx / -8, id = d ? 1.0f / d : 0.0f;
                const uint16_t dh = f32_to_f16(d); memcpy(y, &dh, 2);
                for (int j = 0; j < 16; j++) {
                    const uint8_t q0 = min_<int>(15, (int8_t)(x[j] * id + 8.5f)), q1 = min_<int>(15, (int8_t)(x[16 + j] * id + 8.5f));
                    y[2 + j] = (uint8_t)(q0 | (q1 << 4));
                }
                y += 18; break; }
            case T_Q4_1: {
                float mn = INFINITY, mx = -INFINITY;
                for (int j = 0; j < 32; j++) { if (x[j] < mn) mn = x[j]; if (x[j] > mx) mx = x[j]; }
                const float d = (mx - mn) / 15, id = d ? 1.0f / d : 0.0f;
                const uint16_t dh = f32_to_f16(d), mh = f32_to_f16(mn); memcpy(y, &dh, 2); memcpy(y + 2, &mh, 2);
                for (int j = 0; j < 16; j++) {
                    const uint8_t q0 = min_<int>(15, (int8_t)((x[j] - mn) * id + 0.5f)), q1 = min_<int>(15, (int8_t)((x[16 + j] - mn) * id + 0.5f));
                    y[4 + j] = (uint8_t)(q0 | (q1 << 4));
                }
                y += 20; break; }
            case T_Q5_0: {
                float amax = 0.0f, mx = 0.0f;
                for (int j = 0; j < 32; j++) if (amax < fabsf(x[j])) { amax = fabsf(x[j]); mx = x[j]; }
                const float d = mx / -16, id = d ? 1.0f / d : 0.0f;
                const uint16_t dh = f32_to_f16(d); memcpy(y, &dh, 2);
                uint32_t qh = 0;
                for (int j = 0; j < 16; j++) {
                    const uint8_t q0 = min_<int>(31, (int8_t)(x[j] * id + 16.5f)), q1 = min_<int>(31, (int8_t)(x[16 + j] * id + 16.5f));
                    y[6 + j] = (uint8_t)((q0 & 0x0F) | ((q1 & 0x0F) << 4));
                    qh |= (uint32_t)((q0 & 0x10) >> 4) << j;
                    qh |= (uint32_t)((q1 & 0x10) >> 4) << (j + 16);
                }
                memcpy(y + 2, &qh, 4);
                y += 22; break; }
            case T_Q5_1: {
                float mn = INFINITY, mx = -INFINITY;
                for (int j = 0; j < 32; j++) { if (x[j] < mn) mn = x[j]; if (x[j] > mx) mx = x[j]; }
                const float d = (mx - mn) / 31, id = d ? 1.0f / d : 0.0f;
                const uint16_t dh = f32_to_f16(d), mh = f32_to_f16(mn); memcpy(y, &dh, 2); memcpy(y + 2, &mh, 2);
                uint32_t qh = 0;
                for (int j = 0; j < 16; j++) {
                    const uint8_t q0 = (uint8_t)((x[j] - mn) * id + 0.5f), q1 = (uint8_t)((x[16 + j] - mn) * id + 0.5f);
                    y[8 + j] = (uint8_t)((q0 & 0x0F) | ((q1 & 0x0F) << 4));
                    qh |= (uint32_t)((q0 & 0x10) >> 4) << j;
                    qh |= (uint32_t)((q1 & 0x10) >> 4) << (j + 16);
                }
                memcpy(y + 4, &qh, 4);
                y += 24; break; }
            default: {  // Q8_0
                float amax = 0.0f;
                for (int j = 0; j < 32; j++) if (fabsf(x[j]) > amax) amax = fabsf(x[j]);
                const float d = amax / 127, id = d ? 1.0f / d : 0.0f;
                const uint16_t dh = f32_to_f16(d); memcpy(y, &dh, 2);
                for (int j = 0; j < 32; j++) ((int8_t *) y)[2 + j] = (int8_t) roundf(x[j] * id);
                y += 34; break; }
        }
    }
}

struct FileCloser { void operator()(FILE * f) const { if (f) fclose(f); } };

}  // namespace rwkvmi

using namespace rwkvmi;

extern "C" RWKV_API bool rwkv_quantize_model_file(const char * in_path, const char * out_path, const char * type_name) {
    g_last_error = RWKV_ERROR_NONE;
    RW_CHECK(RWKV_ERROR_ARGS, false, in_path && out_path && type_name, "NULL argument");
    const int out_type = dtype_from_name(type_name);
    RW_CHECK(RWKV_ERROR_ARGS | RWKV_ERROR_DATA_TYPE, false, out_type >= 0 && dtype_quantized(out_type), "Unsupported output data type (%s)", type_name);

    if (g_print_errors) fprintf(stderr, "Loading model from '%s'\n", in_path);
    std::unique_ptr<FILE, FileCloser> in(fopen(in_path, "rb"));
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_OPEN, false, in != nullptr, "Failed to open %s for reading", in_path);
    struct stat st;
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_STAT, false, fstat(fileno(in.get()), &st) == 0, "failed to stat file %s", in_path);
    std::unique_ptr<FILE, FileCloser> out(fopen(out_path, "wb"));
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_OPEN, false, out != nullptr, "Failed to open %s for writing", out_path);

    FileHeader hdr;
    if (!read_file_header(in.get(), hdr)) { global_fail(RWKV_ERROR_FILE, __FILE__, __LINE__, "read_file_header", "Invalid file header"); return false; }
    RW_CHECK(RWKV_ERROR_FILE, false, hdr.data_type == T_F32 || hdr.data_type == T_F16,
             "Unsupported input data type (%s); needs to be FP32 or FP16", dtype_name((int) hdr.data_type));
    FileHeader oh = hdr;
    oh.version = RWKV_FILE_VERSION;
    oh.data_type = (uint32_t) out_type;
    RW_CHECK(RWKV_ERROR_FILE | RWKV_ERROR_FILE_WRITE, false, fwrite(&oh, sizeof oh, 1, out.get()) == 1, "Failed to write file header");

    std::vector<uint8_t> raw, packed;
    std::vector<float> f32;
    uint64_t orig_total = 0, new_total = 0;
    while ((uint64_t) ftello(in.get()) < (uint64_t) st.st_size) {
        TensorInfo t;
        if (!read_tensor_info(in.get(), t)) { global_fail(RWKV_ERROR_MODEL_PARAMS, __FILE__, __LINE__, "read_tensor_info", "Failed to read tensor header"); return false; }
        raw.resize(t.nbytes);
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_FILE_READ, false, t.nbytes == 0 || fread(raw.data(), 1, t.nbytes, in.get()) == t.nbytes,
                 "Failed to read tensor data of %s", t.name.c_str());
        const uint8_t * payload = raw.data();
        uint64_t out_bytes = t.nbytes;
        int write_type = t.type;
        const bool quantize = (t.type == T_F32 || t.type == T_F16) && t.ndim == 2 && tensor_needs_quant(t.name);
        // the reference hands every such tensor to ggml_quantize_chunk, which aborts on a row length that is not a whole number of blocks
        RW_CHECK(RWKV_ERROR_MODEL_PARAMS | RWKV_ERROR_SHAPE, false, !quantize || t.ne[0] % 32 == 0,
                 "Tensor %s has rows of %" PRId64 " elements: quantised formats need a multiple of 32", t.name.c_str(), t.ne[0]);
        if (g_print_errors) fprintf(stderr, "%48s - [%5u, %5u, %5u], type = %6s ", t.name.c_str(), (unsigned) t.ne[0], (unsigned) t.ne[1], (unsigned) t.ne[2], dtype_name(t.type));
        if (quantize) {
            const int64_t n = t.nelements();
            f32.resize((size_t) n);
            if (t.type == T_F16) { const uint16_t * h = (const uint16_t *) raw.data(); for (int64_t i = 0; i < n; i++) f32[i] = f16_to_f32(h[i]); }
            else memcpy(f32.data(), raw.data(), (size_t) n * 4);
            out_bytes = tensor_nbytes(out_type, t.ne[0], t.ne[1], t.ne[2]);
            packed.resize(out_bytes);
            const size_t row_bytes = (size_t)(t.ne[0] / 32) * dtype_block_bytes(out_type);
            for (int64_t r = 0; r < t.ne[1]; r++) quantize_row(out_type, f32.data() + r * t.ne[0], packed.data() + (size_t) r * row_bytes, t.ne[0]);
            payload = packed.data();
            write_type = out_type;
            if (g_print_errors) fprintf(stderr, "-> %6s size = %8.2f MB -> %8.2f MB\n", dtype_name(out_type), t.nbytes / 1048576.0, out_bytes / 1048576.0);
        } else if (g_print_errors) {
            fprintf(stderr, "size = %8.3f MB\n", t.nbytes / 1048576.0);
        }
        uint32_t th[6] = {(uint32_t) t.ndim, (uint32_t) t.name.size(), (uint32_t) write_type, (uint32_t) t.ne[0], (uint32_t) t.ne[1], (uint32_t) t.ne[2]};
        bool ok = fwrite(th, sizeof(uint32_t), 3 + (size_t) t.ndim, out.get()) == 3 + (size_t) t.ndim;
        ok = ok && fwrite(t.name.data(), 1, t.name.size(), out.get()) == t.name.size();
        ok = ok && (out_bytes == 0 || fwrite(payload, 1, out_bytes, out.get()) == out_bytes);
        RW_CHECK(RWKV_ERROR_FILE_WRITE, false, ok, "Failed to write tensor %s", t.name.c_str());
        orig_total += t.nbytes;
        new_total += out_bytes;
    }
    if (g_print_errors) {
        fprintf(stderr, "original size     = %8.2f MB\n", orig_total / 1048576.0);
        fprintf(stderr, "quantized size    = %8.2f MB\n", new_total / 1048576.0);
        fprintf(stderr, "compression ratio = %8.2f\n", orig_total / (double) (new_total ? new_total : 1));
    }
    return true;
}
