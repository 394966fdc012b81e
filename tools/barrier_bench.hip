// Microbenchmark: grid-wide barrier cost + cross-XCD data visibility on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned * ctr, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > 20000000) { ok = false; break; } }
        } else {
            __builtin_amdgcn_s_waitcnt(0);  // all stores issued & acked (vmcnt/lgkmcnt 0)
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > 20000000) { ok = false; break; } }
        }
    }
    __syncthreads();
    return ok;
}

// Each round: every block writes NV floats (its slice of a vector), barrier, then reads the WHOLE vector (NB*NV floats)
// and checks it; measures rounds/s.
template <int MODE>
__global__ void __launch_bounds__(512) k_bar(unsigned * ctr, float * vec, int rounds, int nv, int * bad, long long * cyc) {
    const int nb = gridDim.x;
    long long t0 = __builtin_readcyclecounter();
    int errs = 0;
    for (int r = 0; r < rounds; r++) {
        float * v = vec + (size_t) (r & 1) * nb * nv;
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {
            const float val = (float) (r * 7 + blockIdx.x * nv + i);
            if (MODE == 0) v[blockIdx.x * nv + i] = val;
            else __hip_atomic_store(&v[blockIdx.x * nv + i], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_barrier<MODE>(ctr, (unsigned) (r + 1) * nb)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
        for (int i = threadIdx.x; i < nb * nv; i += blockDim.x) {
            float got;
            if (MODE == 0) got = v[i];
            else got = __hip_atomic_load(&v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != (float) (r * 7 + i)) errs++;
        }
    }
    if (errs) atomicAdd(bad, errs);
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = __builtin_readcyclecounter() - t0;
}

template <int MODE>
static void run(const char * name, bool uncached, int nb, int nv, int rounds) {
    unsigned * ctr; float * vec; int * bad; long long * cyc;
    CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&cyc, 8));
    if (uncached) CK(hipExtMallocWithFlags((void **) &vec, (size_t) 2 * nb * nv * 4, hipDeviceMallocUncached));
    else CK(hipMalloc(&vec, (size_t) 2 * nb * nv * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(ctr, 0, 256)); CK(hipMemset(bad, 0, 4));
        CK(hipEventRecord(a));
        k_bar<MODE><<<nb, 512>>>(ctr, vec, rounds, nv, bad, cyc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (rep) printf("%-34s nb=%d nv=%d: %.2f us/round, errors=%d\n", name, nb, nv, ms * 1000.0 / rounds, hb);
    }
    CK(hipFree(ctr)); CK(hipFree(vec)); CK(hipFree(bad)); CK(hipFree(cyc));
}

// MODE 2: per-block arrival flags (no RMW contention), every block polls all flags; data via plain wide accesses.
__device__ __forceinline__ bool flag_barrier(unsigned * flags, unsigned epoch, int nb) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
        long spins = 0;
        for (;;) {
            bool all = true;
            for (int i = threadIdx.x; i < nb; i += 64) all &= __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
            if (__all(all)) break;
            if (++spins > 2000000) { ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

template <int WIDE>
__global__ void __launch_bounds__(512) k_flag(unsigned * flags, float * vec, int rounds, int nv, int * bad) {
    const int nb = gridDim.x;
    int errs = 0;
    for (int r = 0; r < rounds; r++) {
        float * v = vec + (size_t) (r & 1) * nb * nv;
        for (int i = threadIdx.x; i < nv; i += blockDim.x) v[blockIdx.x * nv + i] = (float) (r * 7 + blockIdx.x * nv + i);
        if (!flag_barrier(flags, (unsigned) r + 1, nb)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
        if (WIDE) {
            const float4 * v4 = (const float4 *) v;
            for (int i = threadIdx.x; i < nb * nv / 4; i += blockDim.x) {
                const float4 g = v4[i];
                const float e = (float) (r * 7 + i * 4);
                if (g.x != e || g.y != e + 1 || g.z != e + 2 || g.w != e + 3) errs++;
            }
        } else {
            for (int i = threadIdx.x; i < nb * nv; i += blockDim.x) if (v[i] != (float) (r * 7 + i)) errs++;
        }
    }
    if (errs) atomicAdd(bad, errs);
}

template <int WIDE>
static void runf(const char * name, bool uncached, int nb, int nv, int rounds) {
    unsigned * flags; float * vec; int * bad;
    CK(hipMalloc(&bad, 4));
    if (uncached) { CK(hipExtMallocWithFlags((void **) &vec, (size_t) 2 * nb * nv * 4 + 16, hipDeviceMallocUncached)); CK(hipExtMallocWithFlags((void **) &flags, 4096, hipDeviceMallocUncached)); }
    else { CK(hipMalloc(&vec, (size_t) 2 * nb * nv * 4 + 16)); CK(hipMalloc(&flags, 4096)); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(flags, 0, 4096)); CK(hipMemset(bad, 0, 4));
        CK(hipEventRecord(a));
        k_flag<WIDE><<<nb, 512>>>(flags, vec, rounds, nv, bad);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (rep) printf("%-34s nb=%d nv=%d: %.2f us/round, errors=%d\n", name, nb, nv, ms * 1000.0 / rounds, hb);
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
    const int nb = p.multiProcessorCount;
    run<0>("fence+normal mem", false, nb, 16, 2000);
    run<0>("fence+normal mem", false, nb, 64, 2000);
    run<1>("relaxed atomics+normal mem", false, nb, 16, 2000);
    run<1>("relaxed atomics+normal mem", false, nb, 64, 2000);
    run<1>("relaxed atomics+uncached mem", true, nb, 16, 2000);
    run<0>("fence+uncached mem", true, nb, 16, 2000);
    run<0>("fence+normal mem, 0 data", false, nb, 0, 2000);
    run<1>("relaxed, 0 data", false, nb, 0, 2000);
    runf<0>("flags, normal mem, 0 data", false, nb, 0, 2000);
    runf<0>("flags, uncached, 0 data", true, nb, 0, 2000);
    runf<1>("flags, uncached, wide plain", true, nb, 16, 2000);
    runf<1>("flags, uncached, wide plain", true, nb, 64, 2000);
    runf<1>("flags, normal, wide plain", false, nb, 16, 2000);
    return 0;
}
