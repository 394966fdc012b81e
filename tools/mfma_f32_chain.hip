// Is v_mfma_f32_16x16x4_f32 a chain of single fused multiply-adds in k order, bit for bit? (the exact sequence-mode arm for F16 / F32 matrices
// rests on it: ggml's partial sum p is the chain fma(w[p + 32 j], x[p + 32 j], .) over j, and one instruction adds four links of it)
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_f32_chain.hip -o tools/mfma_f32_chain ; run: tools/mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float * A, const float * B, const float * C, float * D, int n) {
    const int lane = threadIdx.x;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        const float * a = A + (size_t) it * 64, * b = B + (size_t) it * 64; const float * c = C + (size_t) it * 256;
        f4 acc;
        for (int r = 0; r < 4; r++) acc[r] = c[(4 * (lane / 16) + r) * 16 + lane % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[lane], b[lane], acc, 0, 0, 0);    // a[16 k + i], b[16 k + j]
        for (int r = 0; r < 4; r++) D[(size_t) it * 256 + (4 * (lane / 16) + r) * 16 + lane % 16] = acc[r];
    }
}
int main() {
    const int n = 4096;
    std::vector<float> A(n * 64), B(n * 64), C(n * 256), D(n * 256);
    srand(12345);
    auto rnd = [](int mode) {
        float v = (float) rand() / RAND_MAX * 2.0f - 1.0f;
        if (mode == 1) v *= ldexpf(1.0f, rand() % 40 - 20);
        if (mode == 2) { unsigned u; memcpy(&u, &v, 4); u &= 0xFFFFE000u; memcpy(&v, &u, 4); }   // fp16-like mantissa
        return v;
    };
    for (int it = 0; it < n; it++) { const int mode = it % 3; for (int i = 0; i < 64; i++) { A[it * 64 + i] = rnd(mode); B[it * 64 + i] = rnd(mode); } for (int i = 0; i < 256; i++) C[it * 256 + i] = rnd(mode) * 4.0f; }
    float * dA, * dB, * dC, * dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, n);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long bad_fwd = 0, bad_rev = 0, bad_sum = 0, tot = 0;
    for (int it = 0; it < n; it++) for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        const float * a = &A[it * 64], * b = &B[it * 64];
        float f = C[it * 256 + i * 16 + j], r = f;
        for (int kk = 0; kk < 4; kk++) f = fmaf(a[16 * kk + i], b[16 * kk + j], f);
        for (int kk = 3; kk >= 0; kk--) r = fmaf(a[16 * kk + i], b[16 * kk + j], r);
        double s = C[it * 256 + i * 16 + j]; for (int kk = 0; kk < 4; kk++) s += (double) a[16 * kk + i] * b[16 * kk + j];
        const float got = D[it * 256 + i * 16 + j];
        tot++; bad_fwd += memcmp(&got, &f, 4) != 0; bad_rev += memcmp(&got, &r, 4) != 0; bad_sum += got != (float) s;
    }
    printf("v_mfma_f32_16x16x4_f32 against: fmaf chain k = 0..3: %ld of %ld differ; chain k = 3..0: %ld differ; one rounding of the exact sum: %ld differ\n", bad_fwd, tot, bad_rev, bad_sum);
    return bad_fwd != 0;
}
