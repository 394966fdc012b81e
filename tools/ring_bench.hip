// Step (1) of DESIGN.md section 9.1: the LDS-DMA weight ring alone. 256 workgroups x 8 waves (one per CU). Wave 0 is the loader: it
// streams this CU's private region of a big buffer through a ring of NSLOT x 16 KiB LDS slots with global_load_lds_dwordx4 (nt or
// default policy), at most DEPTH fills in flight; NCONS consumer waves read every slot (ds_read_b128, each wave its share of the slot's 16 one-KiB rows),
// fold it into an int8 dot product and release the slot. Prints the time per 16-KiB fill, GB/s per CU and for the chip, and checks
// the consumers' sum against the host's (the ring protocol must neither read a slot early nor overwrite it early).
//
//   ring_bench [MB per CU, default 16] [repetitions, default 5]
//
// Protocol (all words in LDS, monotonic counters, no resets):
//   landed    number of fills whose data is in LDS (written by the loader after s_waitcnt vmcnt(16 * (DEPTH - 1)): fills land in order)
//   (relaxed LDS accesses everywhere: LDS operations of a wave execute in order, the DMA's completion is what vmcnt counts, and a
//    release / acquire pair would make the compiler wait for EVERY outstanding fill)
//   released  [NSLOT] number of consumer waves that are done with the slot, summed over generations; the loader may refill slot s
//             for fill f (generation g = f / NSLOT) once released[s] == NCONS * g
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int SLOT = 16384, NSLOT = 8, NT = 512;
typedef int v4i __attribute__((ext_vector_type(4)));

template <int NCONS, int DEPTH, bool NTP, bool SADDR = false>
__global__ void __launch_bounds__(NT) k_ring(const unsigned char * __restrict__ buf, size_t bytes_per_cu, long long * __restrict__ sums) {
    static_assert(DEPTH >= 1 && DEPTH <= 4, "vmcnt has 6 bits: 16 DMA instructions per fill, at most 3 fills behind the one waited for");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ unsigned landed;
    __shared__ unsigned released[NSLOT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) landed = 0;
    if (tid < NSLOT) released[tid] = 0;
    __syncthreads();
    const unsigned char * mine = buf + (size_t) blockIdx.x * bytes_per_cu;
    const int n_fills = (int) (bytes_per_cu / SLOT);
    const unsigned lds_base = (unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) lds;
    if (wave == 0) {
        // ---- loader ----
        for (int f = 0; f < n_fills + DEPTH - 1; f++) {
            if (f < n_fills) {
                const int s = f % NSLOT;
                const unsigned need = (unsigned) NCONS * (unsigned) (f / NSLOT);
                for (long spin = 0; spin < 100000000 && __hip_atomic_load(&released[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need; spin++) __builtin_amdgcn_s_sleep(1);
                const unsigned char * src = mine + (size_t) f * SLOT + lane * 16;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned) (s * SLOT + i * 1024));
                    const unsigned char * p = src + i * 1024;
                    unsigned keep;
                    if (SADDR) {
                        // scalar base + per-lane 32-bit offset (the form ring_v6.hip's loader uses)
                        const unsigned long long sb = (unsigned long long) (mine + (size_t) f * SLOT + i * 1024);
                        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) sb), hi = __builtin_amdgcn_readfirstlane((unsigned) (sb >> 32));
                        const unsigned long long sbu = ((unsigned long long) hi << 32) | lo;
                        const unsigned voff = (unsigned) lane * 16u;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(sbu), "s"(dst) : "memory");
                    } else if (NTP) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
                    else     asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
                }
            }
            // fill f - (DEPTH - 1) has landed once at most 16 * (DEPTH - 1) DMA instructions are outstanding (the tail issues nothing
            // new, so wait for everything there)
            const int done = f - (DEPTH - 1);
            if (done >= 0) {
                if (f < n_fills) {
                    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (lane == 0) __hip_atomic_store(&landed, (unsigned) (f < n_fills ? done + 1 : n_fills), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (f >= n_fills) break;          // (everything has landed)
            }
        }
        return;
    }
    if (wave > NCONS) return;
    // ---- consumers: a slot is 16 rows of 1 KiB (one wave read each); wave c = 1 .. NCONS takes the rows c - 1, c - 1 + NCONS, ... ----
    int acc = 0;
    const int ones = 0x01010101;
    for (int f = 0; f < n_fills; f++) {
        const int s = f % NSLOT;
        for (long spin = 0; spin < 100000000 && __hip_atomic_load(&landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= (unsigned) f; spin++) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");                          // (compiler: the slot is read after the flag)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r % NCONS != wave - 1) continue;                  // (wave-uniform)
            const v4i w = *reinterpret_cast<const v4i *>(lds + s * SLOT + r * 1024 + lane * 16);
            acc = __builtin_amdgcn_sdot4(w[0], ones, acc, false);
            acc = __builtin_amdgcn_sdot4(w[1], ones, acc, false);
            acc = __builtin_amdgcn_sdot4(w[2], ones, acc, false);
            acc = __builtin_amdgcn_sdot4(w[3], ones, acc, false);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot's reads are done before it is handed back
        if (lane == 0) __hip_atomic_fetch_add(&released[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    long long tot = acc;
    for (int sh = 32; sh; sh >>= 1) tot += __shfl_xor(tot, sh);
    if (lane == 0) atomicAdd((unsigned long long *) &sums[blockIdx.x], (unsigned long long) tot);
}

template <int NCONS, int DEPTH, bool NTP, bool SADDR = false>
static void run(const unsigned char * d_buf, size_t bytes_per_cu, long long * d_sums, const std::vector<long long> & want, int reps) {
    const size_t lds = (size_t) NSLOT * SLOT;
    CK(hipFuncSetAttribute((const void *) k_ring<NCONS, DEPTH, NTP, SADDR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    bool ok = true;
    for (int r = 0; r < reps; r++) {
        CK(hipMemset(d_sums, 0, 256 * sizeof(long long)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_ring<NCONS, DEPTH, NTP, SADDR>), dim3(256), dim3(NT), lds, 0, d_buf, bytes_per_cu, d_sums);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        std::vector<long long> got(256);
        CK(hipMemcpy(got.data(), d_sums, 256 * sizeof(long long), hipMemcpyDeviceToHost));
        for (int b = 0; b < 256; b++) ok = ok && got[b] == want[b];
    }
    const double fills = (double) (bytes_per_cu / SLOT);
    printf("cons %d depth %d %s: %8.3f ms  %6.3f us per 16-KiB fill  %6.1f GB/s per CU  %6.2f TB/s chip  sums %s\n", NCONS, DEPTH, SADDR ? "nt saddr" : (NTP ? "nt     " : "default"),
           best, best * 1e3 / fills, bytes_per_cu / (best * 1e-3) / 1e9, 256.0 * bytes_per_cu / (best * 1e-3) / 1e12, ok ? "OK" : "WRONG");
}

int main(int argc, char ** argv) {
    const size_t mb = argc > 1 ? (size_t) atol(argv[1]) : 16;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t bytes_per_cu = mb << 20, total = bytes_per_cu * 256;
    std::vector<unsigned char> h(total);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < total; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; for (int j = 0; j < 8; j++) h[i + j] = (unsigned char) (x >> (8 * j)); }
    std::vector<long long> want(256, 0);
    for (int b = 0; b < 256; b++) { long long s = 0; const unsigned char * p = h.data() + (size_t) b * bytes_per_cu; for (size_t i = 0; i < bytes_per_cu; i++) s += (signed char) p[i]; want[b] = s; }
    unsigned char * d_buf = nullptr;
    long long * d_sums = nullptr;
    CK(hipMalloc(&d_buf, total)); CK(hipMalloc(&d_sums, 256 * sizeof(long long)));
    CK(hipMemcpy(d_buf, h.data(), total, hipMemcpyHostToDevice));
    printf("%zu MiB per CU, %zu GiB in all, ring %d x %d KiB\n", mb, total >> 30, NSLOT, SLOT >> 10);
    run<1, 2, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<2, 2, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<4, 2, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<4, 4, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<4, 4, false>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<7, 4, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<7, 1, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<4, 2, true, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<4, 4, true, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    run<0, 2, true>(d_buf, bytes_per_cu, d_sums, want, reps);
    return 0;
}
