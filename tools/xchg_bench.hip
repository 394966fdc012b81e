// Models one phase hand-over of mega_v6.hip: 256 workgroups x (1 comm wave + 7 workers); every worker publishes one 16-byte
// tagged unit per round, every comm wave polls all 1792 units, then the workgroup barrier. Optional background streaming
// by the workers (LOADS 1 KB loads per worker per round) to see what bulk traffic does to the hand-over latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t xrsrc;

template <int LOADS, int POLLU, int NCOMM, int SLEEP>
__global__ void __launch_bounds__(512) k(void * xch, unsigned xbytes, const int4 * bulk, size_t bulk_n16, int rounds, int * sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x;
    const xrsrc xr = __builtin_amdgcn_make_buffer_rsrc(xch, 0, (int) xbytes, 0x00020000);
    int acc = 0;
    for (int r = 1; r <= rounds; r++) {
        const int buf = (r & 1) * 2048;
        if (wave < NCOMM) {
            v4u v[POLLU];
            for (long spin = 0; spin < 4000000; spin++) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < POLLU; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, (buf + lane + (u * NCOMM + wave) * 64) * 16, 0, 16);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < POLLU; u++) ok = ok && (lane + (u * NCOMM + wave) * 64 >= 1792 || v[u].w == (unsigned) r);
                if (__all(ok)) break;
                if (SLEEP) __builtin_amdgcn_s_sleep(1);
            }
            acc += v[0].x;
        } else {
            int4 w[LOADS > 0 ? LOADS : 1];
            if (LOADS > 0) {
                const size_t base = ((size_t) r * 7919u * 1792u + (size_t) (blk * 7 + (wave + 6) % 7) * LOADS) * 64 % (bulk_n16 - 64 * LOADS);
#pragma unroll
                for (int i = 0; i < LOADS; i++) w[i] = bulk[base + i * 64 + lane];
            }
            if (lane == 0) for (int j = wave - NCOMM; j < 7; j += 8 - NCOMM) { const v4u v = {(unsigned) r, 1u, 2u, (unsigned) r}; __builtin_amdgcn_raw_buffer_store_b128(v, xr, (buf + blk * 7 + j) * 16, 0, 16); }
            if (LOADS > 0) {
#pragma unroll
                for (int i = 0; i < LOADS; i++) acc += w[i].x;
            }
        }
        __syncthreads();
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

template <int LOADS, int POLLU, int NCOMM, int SLEEP>
static void run(const char * name, void * xch, const int4 * bulk, size_t n16, int * sink) {
    const int rounds = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(xch, 0, 4096 * 16));
        CK(hipEventRecord(a));
        k<LOADS, POLLU, NCOMM, SLEEP><<<256, 512>>>(xch, 4096 * 16, bulk, n16, rounds, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("%-44s %.2f us/round  (bulk %.2f TB/s)\n", name, ms * 1000.0 / rounds, (double) LOADS * 1024 * 1792 * rounds / (ms * 1e-3) / 1e12);
    }
}

int main() {
    void * xch; int4 * bulk; int * sink;
    const size_t n16 = (size_t) 1 << 28;   // 4 GiB of bulk data
    CK(hipMalloc(&xch, 4096 * 16)); CK(hipMalloc(&bulk, n16 * 16)); CK(hipMalloc(&sink, 4)); CK(hipMemset(bulk, 1, n16 * 16));
    run<0, 28, 1, 1>("1 comm wave, 28 polls/lane, sleep", xch, bulk, n16, sink);
    run<0, 28, 1, 0>("1 comm wave, 28 polls/lane, no sleep", xch, bulk, n16, sink);
    run<0, 14, 2, 0>("2 comm waves, 14 polls/lane, no sleep", xch, bulk, n16, sink);
    run<0, 7, 4, 0>("4 comm waves, 7 polls/lane, no sleep", xch, bulk, n16, sink);
    run<0, 4, 7, 0>("7 comm waves, 4 polls/lane, no sleep", xch, bulk, n16, sink);
    run<0, 1, 1, 0>("1 comm wave, 64 units only", xch, bulk, n16, sink);
    return 0;
}
