// common.h -- shared declarations of librwkv.so (MI355X-native rwkv.cpp drop-in).
#pragma once

#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <string>
#include <vector>
#include <unordered_map>
#include <memory>

#include "rwkv.h"

// ---------------------------------------------------------------------------------------------------------------
// Error state. Conventions follow the reference (rwkv_error_handling.inc:1-95): a thread-local global for load /
// quantise errors, a per-context word for eval errors, messages "\n<file>:<line>: <expr>\n" on stderr when enabled.
// ---------------------------------------------------------------------------------------------------------------

namespace rwkvmi {

extern thread_local int  g_last_error;    // rwkv_error_flags bits
extern thread_local bool g_print_errors;  // default true

void global_fail(int flags, const char * file, int line, const char * expr, const char * fmt, ...) __attribute__((format(printf, 5, 6)));
void ctx_fail(struct ::rwkv_context * ctx, int flags, const char * file, int line, const char * expr, const char * fmt, ...) __attribute__((format(printf, 6, 7)));

}  // namespace rwkvmi

// Global (load/quantise) check: record flags, print, return RET.
#define RW_CHECK(FLAGS, RET, COND, ...) \
    do { if (!(COND)) { ::rwkvmi::global_fail((FLAGS), __FILE__, __LINE__, #COND, __VA_ARGS__); return RET; } } while (0)
// Per-context (eval) check.
#define RW_CTX_CHECK(CTX, FLAGS, RET, COND, ...) \
    do { if (!(COND)) { ::rwkvmi::ctx_fail((CTX), (FLAGS), __FILE__, __LINE__, #COND, __VA_ARGS__); return RET; } } while (0)

namespace rwkvmi {

// ---------------------------------------------------------------------------------------------------------------
// On-disk format (docs/FILE_FORMAT.md; rwkv_file_format.inc:5-24, 102-197)
// ---------------------------------------------------------------------------------------------------------------

enum DType : int {
    T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q4_1_O = 4, T_Q4_2 = 5, T_Q4_3 = 6, T_Q5_0 = 7, T_Q5_1 = 8, T_Q8_0 = 9,
    T_COUNT = 17
};

bool        dtype_supported(int t);     // one of the seven loadable types
bool        dtype_quantized(int t);
size_t      dtype_block_bytes(int t);   // bytes per block (or per element for F32/F16)
int         dtype_block_elems(int t);   // 32 for quantised types, 1 otherwise
const char* dtype_name(int t);
int         dtype_from_name(const char * s);  // -1 when unknown

struct FileHeader {
    uint32_t magic, version, n_vocab, n_embed, n_layer, data_type;
};

struct TensorInfo {
    std::string name;
    int      type = 0;
    int      ndim = 0;
    int64_t  ne[3] = {1, 1, 1};   // ggml order: ne[0] is the contiguous (row) dimension
    uint64_t file_offset = 0;     // offset of the data bytes in the file
    uint64_t nbytes = 0;          // rwkv_utilities.inc:1-3: type_size * n / block
    int64_t  nelements() const { return ne[0] * ne[1] * ne[2]; }
};

// Parses the 24-byte header with the reference's checks and error flags (rwkv_file_format.inc:115-142).
bool read_file_header(FILE * f, FileHeader & h);
// Parses one tensor header + key at the current position (rwkv_file_format.inc:152-197); leaves the file at the data.
bool read_tensor_info(FILE * f, TensorInfo & t);
uint64_t tensor_nbytes(int type, int64_t n0, int64_t n1, int64_t n2);

// ---------------------------------------------------------------------------------------------------------------
// Device-side tensors
// ---------------------------------------------------------------------------------------------------------------

// A parameter resident in HBM. F32/F16 tensors keep the file layout. Quantised tensors are re-packed at load into
// planes so that every plane is naturally aligned (only the FILE format is fixed; see DESIGN.md "HBM layout"):
//   qs : 16 B (Q4_x, Q5_x) or 32 B (Q8_0) of codes per block, [rows][blocks]
//   qh : u32 fifth bits per block (Q5_x only)
//   sc : fp16 d per block (Q4_0, Q5_0, Q8_0) or {fp16 d, fp16 m} per block (Q4_1, Q5_1)
struct DevTensor {
    std::string name;
    int      type = 0;
    int      ndim = 0;
    int64_t  ne[3] = {1, 1, 1};
    uint64_t nbytes = 0;       // algorithmic bytes (file dtype)
    void *   data = nullptr;   // F32 / F16 payload
    uint8_t *  qs = nullptr;
    uint32_t * qh = nullptr;
    void *     sc = nullptr;
    // second, tile-major image of a quantised matrix for the sequence-mode GEMM (prefill.hip), built on first use with T >= 32
    mutable uint8_t *  pf_qs = nullptr;
    mutable uint32_t * pf_sc = nullptr;
    mutable uint32_t * pf_qh = nullptr;
    mutable int64_t    pf_rows = 0;
    int64_t rows() const { return ne[1] * ne[2]; }
    int64_t cols() const { return ne[0]; }
};

}  // namespace rwkvmi
