// Premise check for a loader-wave design: can dedicated waves stream weights at full rate with a SHALLOW per-wave queue
// while the hand-over of xchg_bench.hip keeps its idle latency? 256 workgroups x 8 waves: wave 0 polls, waves 1..NPROD
// publish one unit per round, the last NLOAD waves stream a big buffer continuously, keeping at most Q loads in flight,
// until the rounds are over. Prints us/round and the bulk rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define VMCNT_IMM(Q) (((Q) & 0xF) | (((Q) >> 4) << 14) | (0x7 << 4) | (0xF << 8))

template <int NLOAD, int Q>
__global__ void __launch_bounds__(512) k(void * xch, unsigned xbytes, const int4 * bulk, size_t bulk_n16, int rounds, unsigned * stop, unsigned long long * moved, int * sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xch, 0, (int) xbytes, 0x00020000);
    int acc = 0;
    __shared__ unsigned cnt;
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (wave >= 8 - NLOAD) {          // loader: stream until told to stop
        const size_t stride = (size_t) 256 * NLOAD * 64;
        size_t pos = ((size_t) blk * NLOAD + (wave - (8 - NLOAD))) * 64 + lane;
        unsigned long long n = 0;
        for (long it = 0; it < 4000000 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; it++) {
            int4 w[Q];
#pragma unroll
            for (int i = 0; i < Q; i++) { w[i] = bulk[pos % bulk_n16]; pos += stride; }
#pragma unroll
            for (int i = 0; i < Q; i++) asm volatile("" :: "v"(w[i].x), "v"(w[i].y), "v"(w[i].z), "v"(w[i].w));
            n += Q;
        }
        if (lane == 0) atomicAdd(moved, n * 1024ull);
        return;
    }
    for (int r = 1; r <= rounds; r++) {
        const int buf = (r & 1) * 2048;
        if (wave == 0) {
            v4u v[28];
            for (long spin = 0; spin < 4000000; spin++) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < 28; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, (buf + lane + u * 64) * 16, 0, 16);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 28; u++) ok = ok && (lane + u * 64 >= 1792 || v[u].w == (unsigned) r);
                if (__all(ok)) break;
            }
            acc += v[0].x;
        } else if (lane == 0) {
            for (int j = wave - 1; j < 7; j += 7 - NLOAD) { const v4u v = {(unsigned) r, 1u, 2u, (unsigned) r}; __builtin_amdgcn_raw_buffer_store_b128(v, xr, (buf + blk * 7 + j) * 16, 0, 16); }
        }
        // workgroup barrier among the non-loader waves only: named barriers are not available, so spin on an LDS counter
        if (lane == 0) __hip_atomic_fetch_add(&cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (long z = 0; z < 50000000 && __hip_atomic_load(&cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned) r * (8 - NLOAD); z++) __builtin_amdgcn_s_sleep(1);
    }
    if (blk == 0 && tid == 0) __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 0x7fffffff) sink[0] = acc;
}

template <int NLOAD, int Q>
static void run(void * xch, const int4 * bulk, size_t n16, unsigned * stop, unsigned long long * moved, int * sink) {
    const int rounds = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(xch, 0, 4096 * 16)); CK(hipMemset(stop, 0, 4)); CK(hipMemset(moved, 0, 8));
        CK(hipEventRecord(a));
        k<NLOAD, Q><<<256, 512>>>(xch, 4096 * 16, bulk, n16, rounds, stop, moved, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long mv; CK(hipMemcpy(&mv, moved, 8, hipMemcpyDeviceToHost));
        if (rep) printf("%d loader waves, <= %2d loads in flight each: %.2f us/round, bulk %.2f TB/s\n", NLOAD, Q, ms * 1000.0 / rounds, (double) mv / (ms * 1e-3) / 1e12);
    }
}

int main() {
    void * xch; int4 * bulk; int * sink; unsigned * stop; unsigned long long * moved;
    const size_t n16 = (size_t) 1 << 28;
    CK(hipMalloc(&xch, 4096 * 16)); CK(hipMalloc(&bulk, n16 * 16)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&stop, 4)); CK(hipMalloc(&moved, 8)); CK(hipMemset(bulk, 1, n16 * 16));
    run<1, 2>(xch, bulk, n16, stop, moved, sink);
    run<1, 8>(xch, bulk, n16, stop, moved, sink);
    run<2, 2>(xch, bulk, n16, stop, moved, sink);
    run<2, 4>(xch, bulk, n16, stop, moved, sink);
    run<2, 8>(xch, bulk, n16, stop, moved, sink);
    run<2, 16>(xch, bulk, n16, stop, moved, sink);
    run<3, 4>(xch, bulk, n16, stop, moved, sink);
    run<3, 8>(xch, bulk, n16, stop, moved, sink);
    return 0;
}
