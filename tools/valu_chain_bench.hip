// Dependent-issue latency of VALU chains on one wave (gfx950): how long a wave takes for 64 dependent f32 adds, with and without DPP
// sources, with independent work between the adds, and with a second wave on the same SIMD. Behind DESIGN.md's WKV-7 sequence kernel:
// its per-token floor is one 64-add ordered chain per row.
// build: make -C tools valu_chain_bench; run on an idle MI355X: ./valu_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

template <int MODE> __global__ __launch_bounds__(64) void k(float * out, const float * in, int iters) {
    float x = in[threadIdx.x], p = in[64 + threadIdx.x], q = in[128 + threadIdx.x];
    float a0 = q, a1 = q + 1.0f, a2 = q + 2.0f, a3 = q + 3.0f;
    for (int i = 0; i < iters; i++) {
        if constexpr (MODE == 0) {           // plain dependent adds
            asm volatile(REP64("v_add_f32 %0, %1, %0\n\t") : "+v"(x) : "v"(p));
        } else if constexpr (MODE == 1) {    // DPP source, dependent through src1
            asm volatile(REP64("v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(x) : "v"(p));
        } else if constexpr (MODE == 2) {    // DPP adds with one independent VALU instruction after each
            asm volatile(REP64("v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mul_f32 %2, %2, %1\n\t")
                         : "+v"(x), "+v"(a0) : "v"(p));
        } else if constexpr (MODE == 3) {    // ... two independent instructions after each
            asm volatile(REP64("v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mul_f32 %2, %2, %1\n\tv_mul_f32 %3, %3, %1\n\t")
                         : "+v"(x), "+v"(a0), "+v"(a1) : "v"(p));
        } else if constexpr (MODE == 4) {    // plain adds with one independent instruction after each
            asm volatile(REP64("v_add_f32 %0, %1, %0\n\tv_mul_f32 %2, %2, %1\n\t") : "+v"(x), "+v"(a0) : "v"(p));
        } else if constexpr (MODE == 5) {    // DPP adds with s_nop 1 in front (what the compiler emits for a DPP instruction reading a fresh register)
            asm volatile(REP64("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(x) : "v"(p));
        } else if constexpr (MODE == 6) {    // 64 independent adds (issue rate)
            asm volatile(REP16("v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %4, %1\n\tv_add_f32 %2, %4, %2\n\tv_add_f32 %3, %4, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p));
        } else if constexpr (MODE == 7) {    // packed dependent chain
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f y = {x, a0}, pp = {p, p};
            asm volatile(REP64("v_pk_add_f32 %0, %1, %0\n\t") : "+v"(y) : "v"(pp));
            x = y.x; a0 = y.y;
        } else if constexpr (MODE == 9) {    // the same add in the 8-byte VOP3 encoding
            asm volatile(REP64("v_add_f32_e64 %0, %1, %0\n\t") : "+v"(x) : "v"(p));
        } else if constexpr (MODE == 10) {   // 64 independent adds, 8-byte encoding
            asm volatile(REP16("v_add_f32_e64 %0, %4, %0\n\tv_add_f32_e64 %1, %4, %1\n\tv_add_f32_e64 %2, %4, %2\n\tv_add_f32_e64 %3, %4, %3\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(p));
        } else if constexpr (MODE == 8) {    // DPP adds with three independent instructions after each
            asm volatile(REP64("v_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mul_f32 %2, %2, %1\n\tv_mul_f32 %3, %3, %1\n\tv_mul_f32 %4, %4, %1\n\t")
                         : "+v"(x), "+v"(a0), "+v"(a1), "+v"(a2) : "v"(p));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + a0 + a1 + a2 + a3;
}

template <int MODE> static void run(const char * what, int wgs, float * out, const float * in) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(64), 0, 0, out, in, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(64), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-70s wgs %5d: %7.1f ns per 64-add block\n", what, wgs, ms * 1e6 / iters); fflush(stdout);
}

int main() {
    float * in, * out;
    hipMalloc(&in, 192 * 4); hipMalloc(&out, 8192 * 64 * 4);
    std::vector<float> h(192, 0.0f); hipMemcpy(in, h.data(), 192 * 4, hipMemcpyHostToDevice);
    for (int wgs : {256, 2048, 4096}) {      // 1 wave per CU, 2 per SIMD, 4 per SIMD
        run<0>("64 dependent v_add_f32", wgs, out, in);
        run<1>("64 dependent v_add_f32_dpp (row_newbcast source 0)", wgs, out, in);
        run<5>("same, s_nop 1 in front of each", wgs, out, in);
        run<4>("64 dependent v_add_f32 + 1 independent v_mul each", wgs, out, in);
        run<2>("64 dependent v_add_f32_dpp + 1 independent v_mul each", wgs, out, in);
        run<3>("64 dependent v_add_f32_dpp + 2 independent v_mul each", wgs, out, in);
        run<8>("64 dependent v_add_f32_dpp + 3 independent v_mul each", wgs, out, in);
        run<6>("64 independent v_add_f32", wgs, out, in);
        run<7>("64 dependent v_pk_add_f32", wgs, out, in);
        run<9>("64 dependent v_add_f32_e64 (8-byte encoding of the same add)", wgs, out, in);
        run<10>("64 independent v_add_f32_e64", wgs, out, in);
    }
    return 0;
}
