// format.cpp -- rwkv.cpp model-file container: header, tensor headers, type table, error plumbing.
// Restates the on-disk contract of the reference (docs/FILE_FORMAT.md:10-68, rwkv_file_format.inc:5-24,102-197,
// rwkv_utilities.inc:1-3) -- the file format is the one thing this library must read bit-for-bit.
#include "common.h"

#include <cstdarg>
#include <cstring>

namespace rwkvmi {

thread_local int  g_last_error = RWKV_ERROR_NONE;
thread_local bool g_print_errors = true;

void global_fail(int flags, const char * file, int line, const char * expr, const char * fmt, ...) {
    g_last_error |= flags;
    if (!g_print_errors) return;
    if (fmt && fmt[0]) {
        va_list ap;
        va_start(ap, fmt);
        vfprintf(stderr, fmt, ap);
        va_end(ap);
    }
    fprintf(stderr, "\n%s:%d: %s\n", file, line, expr);
}

struct TypeRow { int id; const char * name; size_t bytes; int elems; };
// ids 4..6 were removed upstream; 10..16 (Q8_1, K-quants) are named by the reference but never produced by its tools.
static const TypeRow k_types[] = {
    {T_F32, "FP32", 4, 1},   {T_F16, "FP16", 2, 1},   {T_Q4_0, "Q4_0", 18, 32}, {T_Q4_1, "Q4_1", 20, 32},
    {T_Q5_0, "Q5_0", 22, 32}, {T_Q5_1, "Q5_1", 24, 32}, {T_Q8_0, "Q8_0", 34, 32},
};
static const char * k_all_names[T_COUNT + 1] = {
    "FP32", "FP16", "Q4_0", "Q4_1", "Q4_1_O", "Q4_2", "Q4_3", "Q5_0", "Q5_1", "Q8_0", "Q8_1",
    "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "Q8_K", "unknown"
};

static const TypeRow * find_type(int t) {
    for (const TypeRow & r : k_types) if (r.id == t) return &r;
    return nullptr;
}

bool dtype_supported(int t) { return find_type(t) != nullptr; }
bool dtype_quantized(int t) { const TypeRow * r = find_type(t); return r && r->elems == 32; }
size_t dtype_block_bytes(int t) { const TypeRow * r = find_type(t); return r ? r->bytes : 0; }
int dtype_block_elems(int t) { const TypeRow * r = find_type(t); return r ? r->elems : 0; }
const char * dtype_name(int t) { return (t >= 0 && t < T_COUNT) ? k_all_names[t] : k_all_names[T_COUNT]; }
int dtype_from_name(const char * s) {
    for (int i = 0; i < T_COUNT; i++) if (strcmp(s, k_all_names[i]) == 0) return i;
    return -1;
}

uint64_t tensor_nbytes(int type, int64_t n0, int64_t n1, int64_t n2) {
    return (uint64_t) dtype_block_bytes(type) * (uint64_t)(n0 * n1 * n2) / (uint64_t) dtype_block_elems(type);
}

bool read_file_header(FILE * f, FileHeader & h) {
    RW_CHECK(RWKV_ERROR_FILE_READ, false, fread(&h, sizeof(FileHeader), 1, f) == 1, "%s", "");
    RW_CHECK(RWKV_ERROR_FILE_MAGIC, false, h.magic == RWKV_FILE_MAGIC, "%s", "");
    RW_CHECK(RWKV_ERROR_FILE_VERSION, false, h.version >= RWKV_FILE_VERSION_MIN && h.version <= RWKV_FILE_VERSION_MAX,
             "Unsupported file version %u", h.version);
    RW_CHECK(RWKV_ERROR_DATA_TYPE, false, h.data_type < (uint32_t) T_COUNT,
             "Model data type out of range (%u > %d)", h.data_type, T_COUNT - 1);
    RW_CHECK(RWKV_ERROR_DATA_TYPE, false, dtype_supported((int) h.data_type),
             "Models in %s format cannot be loaded anymore because the format was removed.\n"
             "You need to quantize the model into another format", dtype_name((int) h.data_type));
    // Quantised files written before the 2023-05 ggml block-format change carry version 100 and are refused.
    RW_CHECK(RWKV_ERROR_DATA_TYPE, false, !dtype_quantized((int) h.data_type) || h.version == RWKV_FILE_VERSION_1,
             "The quantized model file in %s format was created with an old version of rwkv.cpp and can not be loaded anymore.\n"
             "You need to requantize the model", dtype_name((int) h.data_type));
    return true;
}

bool read_tensor_info(FILE * f, TensorInfo & t) {
    uint32_t fixed[3];
    RW_CHECK(RWKV_ERROR_FILE_READ, false, fread(fixed, sizeof(uint32_t), 3, f) == 3, "%s", "");
    const uint32_t dim_count = fixed[0], key_length = fixed[1], data_type = fixed[2];
    RW_CHECK(RWKV_ERROR_SHAPE, false, dim_count >= 1 && dim_count <= 3, "Tensor has an invalid shape (%u dimensions)", dim_count);
    RW_CHECK(RWKV_ERROR_DATA_TYPE, false, data_type < (uint32_t) T_COUNT, "Tensor data type out of range (%u > %d)", data_type, T_COUNT - 1);
    RW_CHECK(RWKV_ERROR_DATA_TYPE, false, dtype_supported((int) data_type), "Tensor data type (%s) is no longer supported", dtype_name((int) data_type));
    uint32_t dims[3] = {1, 1, 1};
    RW_CHECK(RWKV_ERROR_FILE_READ, false, fread(dims, sizeof(uint32_t), dim_count, f) == dim_count, "%s", "");
    RW_CHECK(RWKV_ERROR_KEY, false, key_length > 0 && key_length < 4096, "Implausible tensor key length %u", key_length);
    std::string key(key_length, '\0');
    RW_CHECK(RWKV_ERROR_FILE_READ, false, fread(&key[0], 1, key_length, f) == key_length, "Failed to read tensor name");
    t.name = std::move(key);
    t.type = (int) data_type;
    t.ndim = (int) dim_count;
    for (int i = 0; i < 3; i++) t.ne[i] = dims[i];
    t.nbytes = tensor_nbytes(t.type, t.ne[0], t.ne[1], t.ne[2]);
    const off_t pos = ftello(f);
    RW_CHECK(RWKV_ERROR_FILE_READ, false, pos >= 0, "%s", "");
    t.file_offset = (uint64_t) pos;
    return true;
}

}  // namespace rwkvmi
