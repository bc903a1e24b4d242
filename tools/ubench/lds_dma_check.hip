#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// each lane loads 4 x 16 B pieces of "its" 64-byte entry into a wave-private staging region, reads them back
__global__ void k(const uint4* __restrict__ src, uint4* dst) {
    __shared__ uint4 buf[8 * 256];                       // 8 waves x 4 KiB
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint4* mine = src + (size_t)(blockIdx.x * 512 + (tid * 7 % 512)) * 4;     // scattered 64-byte entries
    uint4* region = buf + w * 256;
#pragma unroll
    for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((gptr)(mine + p), (lptr)(region + p * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint4 r[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = region[p * 64 + lane];
#pragma unroll
    for (int p = 0; p < 4; ++p) dst[(size_t)(blockIdx.x * 512 + tid) * 4 + p] = r[p];
}
int main() {
    const int blocks = 4, n = blocks * 512 * 4;
    std::vector<uint4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i * 3 + 1, i ^ 0x5555, ~i);
    uint4 *d, *e; hipMalloc(&d, n * 16); hipMalloc(&e, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    k<<<blocks, 512>>>(d, e);
    hipMemcpy(o.data(), e, n * 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < blocks; ++b) for (int t = 0; t < 512; ++t) for (int p = 0; p < 4; ++p) {
        uint4 w = h[(size_t)(b * 512 + (t * 7 % 512)) * 4 + p], g = o[(size_t)(b * 512 + t) * 4 + p];
        if (w.x != g.x || w.y != g.y || w.z != g.z || w.w != g.w) ++bad;
    }
    printf("lds-dma layout check: %s (%d bad)\n", bad ? "MISMATCH" : "ok", bad);
    return bad != 0;
}
