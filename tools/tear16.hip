// Are 16-byte aligned 16-byte vector stores / loads single-copy atomic on this device (agent scope, sc1)?
// Writers rewrite units {t, ~t, t*3, t} with increasing t; readers look for torn units.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

// the product's access path (mega_v6.hip): raw buffer descriptor, sc1 (agent scope) only
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, int u, uint4 v) {
    v4u t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, u * 16, 0, 16);
}
__device__ __forceinline__ uint4 ld16(__amdgpu_buffer_rsrc_t r, int u) {
    asm volatile("" ::: "memory");
    v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, u * 16, 0, 16);
    return make_uint4(t.x, t.y, t.z, t.w);
}

__global__ void k(uint4 * buf, int n_units, int iters, unsigned long long * torn, unsigned long long * seen) {
    const int writer = blockIdx.x & 1;
    const int u = (blockIdx.x >> 1) * blockDim.x + threadIdx.x;
    if (u >= n_units) return;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *) buf, 0, n_units * 16, 0x00020000);
    unsigned long long bad = 0, changes = 0;
    unsigned last = 0;
    for (int i = 1; i <= iters; i++) {
        if (writer) {
            const unsigned t = (unsigned) i;
            st16(r, u, make_uint4(t, ~t, t * 3u, t));
        } else {
            const uint4 v = ld16(r, u);
            if (!(v.y == ~v.x && v.z == v.x * 3u && v.w == v.x)) bad++;
            if (v.x != last) { changes++; last = v.x; }
        }
    }
    if (!writer) { atomicAdd(torn, bad); atomicAdd(seen, changes); }
}

int main() {
    const int n_units = 64 * 256, iters = 20000;
    uint4 * buf; unsigned long long * cnt;
    CK(hipMalloc(&buf, n_units * 16)); CK(hipMalloc(&cnt, 16));
    uint4 init = make_uint4(0, ~0u, 0, 0);
    uint4 * h = new uint4[n_units]; for (int i = 0; i < n_units; i++) h[i] = init;
    CK(hipMemcpy(buf, h, n_units * 16, hipMemcpyHostToDevice)); CK(hipMemset(cnt, 0, 16));
    k<<<2 * (n_units / 256), 256>>>(buf, n_units, iters, cnt, cnt + 1);
    CK(hipDeviceSynchronize());
    unsigned long long r[2]; CK(hipMemcpy(r, cnt, 16, hipMemcpyDeviceToHost));
    printf("reads %llu, value changes observed %llu, TORN %llu\n", (unsigned long long) n_units * iters, r[1], r[0]);
    return 0;
}
