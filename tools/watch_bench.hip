// watch_bench.hip -- what the WATCH in front of a hand-over's sweep costs by the path it reads through (round 6).
//
// The ring kernel (ring_v6.hip) gathers a hand-over in two stages: every gathering wave first watches ONE tagged unit until its tag turns
// (gather_hint: buffer_load_dwordx4 sc1, one request per attempt), then sweeps its share of the vector. DESIGN.md 7.2c blames the watch's
// latency on the queue it shares with the CU's own LDS-DMA fills (guide row handoff-1to1: "the price sits in the consumer CU's memory
// queue"). A scalar load does not travel through the vector memory pipe of the CU: this bench measures a chain of dependent all-to-all
// hand-overs (256 workgroups, one unit each, every workgroup gathers all 256) with the watch done
//     V   buffer_load_dword sc1              (what the kernel does)
//     S   s_load_dword glc                   (scalar data cache bypassed, L2-served)
//     SI  s_dcache_inv + s_load_dword        (scalar cache invalidated per attempt)
//     N   no watch: the sweep polls from the start
// idle and beside a loader wave per CU that streams through an LDS ring (global_load_lds_dwordx4 nt, DEPTH instructions in flight).
// A watch that never sees the store (a stale line served by the reader's L2) ends in the round's time-out and is reported as STALE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t xrsrc;

enum { W_VEC = 0, W_SCALAR_GLC = 1, W_SCALAR_INV = 2, W_NONE = 3 };

template <int MODE, int DEPTH, int NPOLL>
__global__ void __launch_bounds__(512) k(unsigned * xch, const unsigned char * bulk, size_t bulk_bytes, int rounds, unsigned long long * out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned stop, done_round;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = blockIdx.x;
    if (tid == 0) { stop = 0u; done_round = 0u; }
    __syncthreads();
    if (wave == 0) {
        if (DEPTH == 0) return;
        // loader: 1-KiB LDS-DMA fills round and round a 64 KiB ring, DEPTH in flight
        const unsigned ring = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem;
        const size_t span = bulk_bytes / 256;   // this workgroup's slice of the bulk buffer
        unsigned long long src = (unsigned long long) bulk + (size_t) blk * span;
        const unsigned long long end = src + span - 65536;
        unsigned long long n = 0;
        unsigned roff = 0;
        while (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(ring + roff);
            const unsigned voff = (unsigned) lane * 16u;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff), "s"(src), "s"(m0v) : "memory");
            if (DEPTH <= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH <= 16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (DEPTH <= 32) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
            src += 4096; if (src >= end) src = (unsigned long long) bulk + (size_t) blk * span;
            roff = (roff + 4096u) & 65535u;
            n += 4096;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) out[2 + blk] = n;
        return;
    }
    if (wave < 1 || wave > NPOLL) return;
    const xrsrc xr = __builtin_amdgcn_make_buffer_rsrc(xch, 0, 2 * 512 * 16, 0x00020000);
    unsigned stale = 0;
    for (int r = 1; r <= rounds; r++) {
        const int buf = (r & 1) * 512;
        if (lane == 0 && wave == 1) { const v4u v = {(unsigned) r, 1u, 2u, (unsigned) r}; __builtin_amdgcn_raw_buffer_store_b128(v, xr, (buf + blk) * 16, 0, 16); }
        if (NPOLL > 1) for (int i = 0; i < (wave - 1) * 2; i++) __builtin_amdgcn_s_sleep(2);   // stagger the pollers (~110 ns apart at 2.4 GHz)
        const int wu = buf + ((blk * 37 + 11) & 255);
        if (MODE == W_VEC) {
            for (int spin = 0; spin < 2000000; spin++) {
                asm volatile("" ::: "memory");
                const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(xr, wu * 16 + 12, 0, 16);
                if ((int) __builtin_amdgcn_readfirstlane(t) >= r) break;
                __builtin_amdgcn_s_sleep(1);
            }
        } else if ((MODE == W_SCALAR_GLC || MODE == W_SCALAR_INV) && stale < 3u) {   // (three time-outs: the path does not see the stores; the rest of the run sweeps only)
            const unsigned * wp = xch + (size_t) wu * 4 + 3;
            int spin = 0;
            for (; spin < 20000; spin++) {
                unsigned t;
                if (MODE == W_SCALAR_GLC) asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(wp) : "memory");
                else asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(wp) : "memory");
                if ((int) t >= r) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (spin >= 20000) stale++;
        }
        // the sweep: 256 units, four per lane
        for (int spin = 0; spin < 2000000; spin++) {
            if (NPOLL > 1 && (int) __hip_atomic_load(&done_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= r) break;
            asm volatile("" ::: "memory");
            v4u v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(xr, (buf + lane + 64 * u) * 16, 0, 16);
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; u++) ok = ok && (int) v[u].w >= r;
            if (__all(ok)) { if (NPOLL > 1 && lane == 0) __hip_atomic_store(&done_round, (unsigned) r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (wave != 1) return;
    if (lane == 0) { __hip_atomic_store(&stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (blk == 0) out[0] = stale; }
}

template <int MODE, int DEPTH, int NPOLL = 1>
static void run(const char * name, unsigned * xch, const unsigned char * bulk, size_t bulk_bytes, unsigned long long * out) {
    const int rounds = MODE == W_VEC || MODE == W_NONE ? 4000 : 1000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *) k<MODE, DEPTH, NPOLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(xch, 0, 2 * 512 * 16)); CK(hipMemset(out, 0, 8 * 300));
        CK(hipEventRecord(a));
        k<MODE, DEPTH, NPOLL><<<256, 512, 100 * 1024>>>(xch, bulk, bulk_bytes, rounds, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[300]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long bytes = 0; for (int i = 0; i < 256; i++) bytes += h[2 + i];
        if (rep) printf("%-34s %6.2f us / hand-over   stream %.2f TB/s%s\n", name, ms * 1000.0 / rounds, (double) bytes / (ms * 1e-3) / 1e12, h[0] ? "   STALE (watch timed out)" : "");
    }
}

int main() {
    unsigned * xch; unsigned char * bulk; unsigned long long * out;
    const size_t bulk_bytes = (size_t) 8 << 30;
    CK(hipMalloc(&xch, 2 * 512 * 16)); CK(hipMalloc(&bulk, bulk_bytes)); CK(hipMalloc(&out, 8 * 300)); CK(hipMemset(bulk, 1, bulk_bytes));
#define ROW(D) \
    run<W_VEC, D>("vector watch sc1, depth " #D, xch, bulk, bulk_bytes, out); \
    run<W_NONE, D>("no watch (sweep polls), depth " #D, xch, bulk, bulk_bytes, out); \
    run<W_SCALAR_GLC, D>("scalar watch glc, depth " #D, xch, bulk, bulk_bytes, out); \
    run<W_SCALAR_INV, D>("scalar watch dcache_inv, depth " #D, xch, bulk, bulk_bytes, out);
    ROW(0) ROW(16) ROW(48)
    // the same chain with several waves of the workgroup sweeping the same units a fraction of a round trip apart (the first to see the
    // round complete ends it): what a finer sampling of the hand-over's completion is worth
    run<W_NONE, 0, 1>("sweep polls, 1 wave, depth 0", xch, bulk, bulk_bytes, out);
    run<W_NONE, 0, 2>("sweep polls, 2 waves, depth 0", xch, bulk, bulk_bytes, out);
    run<W_NONE, 0, 4>("sweep polls, 4 waves, depth 0", xch, bulk, bulk_bytes, out);
    run<W_NONE, 0, 7>("sweep polls, 7 waves, depth 0", xch, bulk, bulk_bytes, out);
    run<W_VEC, 0, 4>("vector watch, 4 waves, depth 0", xch, bulk, bulk_bytes, out);
    run<W_VEC, 0, 7>("vector watch, 7 waves, depth 0", xch, bulk, bulk_bytes, out);
    run<W_NONE, 16, 1>("sweep polls, 1 wave, depth 16", xch, bulk, bulk_bytes, out);
    run<W_NONE, 16, 4>("sweep polls, 4 waves, depth 16", xch, bulk, bulk_bytes, out);
    run<W_NONE, 16, 7>("sweep polls, 7 waves, depth 16", xch, bulk, bulk_bytes, out);
    return 0;
}
