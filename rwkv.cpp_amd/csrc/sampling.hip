// sampling.hip -- temperature / top-p sampling on the device (the reference samples on the host from downloaded logits,
// python/sampling.py:10-52): softmax, nucleus cut-off, p^(1/temperature), renormalise, draw -- one launch of one workgroup on the
// logits that are already in HBM; the chosen token lands where the next embedding lookup reads it, so a sampling decode loop
// never leaves the device (rwkv_mi_decode_sample). Semantics follow sample_probs() statement by statement:
//   probs = softmax(logits);  top_p == 0 -> 1;  temperature == 0 -> argmax;
//   top_p < 1: cutoff = the probability at which the descending cumulative sum first exceeds top_p; probs < cutoff -> 0;
//   temperature != 1: probs = probs^(1/temperature);  probs /= sum;  token = first index whose cumulative probability exceeds u.
// The cut-off is found without sorting: the largest threshold t (bisection over the float bit pattern, 31 reductions) with
// sum{p >= t} > top_p is exactly that probability. Sums are f32 in a fixed order (per-thread contiguous chunks, then a tree): runs
// are reproducible; against numpy's sequential cumsum the result can differ only when u or top_p falls within rounding of a boundary.
#include "kdev.h"
#include "model.h"

namespace rwkvmi {

__device__ __forceinline__ float block_sum_f(float v, float * red /* [32] */) {
    v = wave_sum_f(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) t += red[w];
    return t;
}

// splitmix64 -> uniform in [0, 1) with 24 bits
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long counter) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (counter + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float) (z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(1024) void k_sample(const float * __restrict__ logits, int n, float temperature, float top_p, float u_in,
                                                 unsigned long long seed, unsigned long long * counter /* read and advanced by thread 0; may be NULL */,
                                                 float * __restrict__ probs, uint32_t * __restrict__ out_token, uint32_t * __restrict__ hist, int hist_pos) {
    __shared__ float red[32];
    __shared__ float l_scan[1024];
    __shared__ int l_pick;
    __shared__ unsigned long long l_ctr;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int C = (n + NT - 1) / NT;                 // contiguous chunk per thread
    const int i0 = tid * C, i1 = i0 + C < n ? i0 + C : n;
    // softmax
    float m = -INFINITY;
    for (int i = i0; i < i1; i++) m = fmaxf(m, logits[i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    for (int w = 0; w < (NT >> 6); w++) m = fmaxf(m, red[w]);
    float part = 0.0f;
    for (int i = i0; i < i1; i++) { const float e = det_expf(logits[i] - m); probs[i] = e; part += e; }
    const float total = block_sum_f(part, red);
    const float inv = 1.0f / total;
    for (int i = i0; i < i1; i++) probs[i] *= inv;
    if (top_p == 0.0f) top_p = 1.0f;
    int pick = -1;
    if (temperature == 0.0f) {
        // argmax, first index of the maximum
        float best = -1.0f; int bi = 0x7fffffff;
        for (int i = i0; i < i1; i++) if (probs[i] > best) { best = probs[i]; bi = i; }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, WAVE); const int oi = __shfl_xor(bi, o, WAVE);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        __shared__ float l_bv[16]; __shared__ int l_bi[16];
        if ((tid & 63) == 0) { l_bv[tid >> 6] = best; l_bi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (NT >> 6); w++) if (l_bv[w] > best || (l_bv[w] == best && l_bi[w] < bi)) { best = l_bv[w]; bi = l_bi[w]; }
            l_pick = bi;
        }
        __syncthreads();
        pick = l_pick;
    } else {
        unsigned cutoff_bits = 0u;
        if (top_p < 1.0f) {
            // largest bit pattern T with sum{p : bits(p) >= T} > top_p  (probabilities are non-negative floats: bit order = value order)
            unsigned T = 0u;
            // (when not even the sum over ALL probabilities exceeds top_p -- float rounding of the softmax -- the reference's
            //  argmax over an all-false mask is 0: the cut-off is the LARGEST probability, i.e. 1 * inv)
            float g0 = 0.0f;
            for (int i = i0; i < i1; i++) g0 += probs[i];
            g0 = block_sum_f(g0, red);
            const bool none = !(g0 > top_p);
            for (int bit = 30; bit >= 0; bit--) {
                const unsigned cand = T | (1u << bit);
                float g = 0.0f;
                for (int i = i0; i < i1; i++) { const float p = probs[i]; g += __float_as_uint(p) >= cand ? p : 0.0f; }
                g = block_sum_f(g, red);
                if (g > top_p) T = cand;
            }
            cutoff_bits = none ? __float_as_uint(1.0f * inv) : T;
        }
        const float it = 1.0f / temperature;
        part = 0.0f;
        for (int i = i0; i < i1; i++) {
            float p = probs[i];
            if (__float_as_uint(p) < cutoff_bits) p = 0.0f;
            else if (temperature != 1.0f) p = p > 0.0f ? powf(p, it) : 0.0f;
            probs[i] = p;
            part += p;
        }
        // inclusive scan of the per-thread sums (Hillis-Steele), then the thread whose range holds u * total walks its chunk
        l_scan[tid] = part;
        __syncthreads();
        for (int o = 1; o < NT; o <<= 1) {
            const float add = tid >= o ? l_scan[tid - o] : 0.0f;
            __syncthreads();
            l_scan[tid] += add;
            __syncthreads();
        }
        const float all = l_scan[NT - 1];
        if (tid == 0) { l_ctr = counter ? *counter : 0ull; l_pick = 0x7fffffff; }
        __syncthreads();
        const unsigned long long ctr = l_ctr;
        const float u = (u_in >= 0.0f ? u_in : uniform01(seed, ctr)) * all;
        const float before = tid ? l_scan[tid - 1] : 0.0f;
        if (i0 < i1 && before <= u && u < l_scan[tid]) {
            // (float prefix sums of a Hillis-Steele scan need not be monotone: more than one thread may see u in its interval -- the
            //  lowest candidate index wins)
            float acc = before;
            int found = -1, last_pos = -1;
            for (int i = i0; i < i1; i++) { acc += probs[i]; if (probs[i] > 0.0f) last_pos = i; if (found < 0 && acc > u) found = i; }
            const int cand = found >= 0 ? found : last_pos;   // (acc can fall short of l_scan[tid] by rounding: the chunk's last candidate)
            if (cand >= 0) atomicMin(&l_pick, cand);
        }
        __syncthreads();
        if (l_pick == 0x7fffffff && tid == 0) {
            // u landed on / beyond the total through rounding: the last token with non-zero probability
            int lp = 0;
            for (int i = n - 1; i >= 0; i--) if (probs[i] > 0.0f) { lp = i; break; }
            l_pick = lp;
        }
        __syncthreads();
        pick = l_pick;
        if (tid == 0 && counter) *counter = ctr + 1;
    }
    if (pick < 0 || pick >= n) pick = 0;   // (all-NaN probabilities: the token must stay a row of the embedding table)
    if (tid == 0) { *out_token = (uint32_t) pick; if (hist) hist[hist_pos] = (uint32_t) pick; }
}

void launch_sample(const float * logits, int n, float temperature, float top_p, float u, unsigned long long seed, unsigned long long * counter,
                   float * probs, uint32_t * out_token, uint32_t * hist, int hist_pos, hipStream_t st) {
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(1024), 0, st, logits, n, temperature, top_p, u, seed, counter, probs, out_token, hist, hist_pos);
}

}  // namespace rwkvmi
