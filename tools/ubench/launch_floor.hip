// What does ONE dependent launch cost on this stack?  The latency regime is ~90 launches of a few sweeps each: the fixed
// cost per launch — dispatch, the first HBM / L2 round trip, the store tail, the end-of-kernel release — bounds it from below.
// A chain of N dependent launches on one stream: (a) an empty kernel, (b) a kernel that loads 32 bytes per thread and stores them
// (65536 threads = a 2^16-element secp256k1 pass), (c) the same through LDS with one barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 a = in[2 * i], b = in[2 * i + 1];
    out[2 * i] = a; out[2 * i + 1] = b;
}
__global__ void k_copy_lds(const uint4* __restrict__ in, uint4* __restrict__ out) {
    __shared__ uint4 t[512];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    t[2 * threadIdx.x] = in[2 * i]; t[2 * threadIdx.x + 1] = in[2 * i + 1];
    __syncthreads();
    const unsigned j = threadIdx.x ^ 64u;
    out[2 * i] = t[2 * j]; out[2 * i + 1] = t[2 * j + 1];
}
int main() {
    const int N = 2000, n = 1 << 16;
    uint4 *a, *b; (void)hipMalloc(&a, n * 32); (void)hipMalloc(&b, n * 32); (void)hipMemset(a, 1, n * 32);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* what, auto launch) {
        for (int i = 0; i < 50; ++i) launch(i);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < N; ++i) launch(i);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %.2f us per launch\n", what, ms * 1e3 / N);
    };
    run("empty kernel, 1 workgroup", [&](int) { k_empty<<<1, 64>>>(); });
    run("empty kernel, 256 workgroups x 256 threads", [&](int) { k_empty<<<256, 256>>>(); });
    run("copy 2 MiB (256 x 256 threads, 32 B each), dependent chain", [&](int i) { if (i & 1) k_copy<<<256, 256>>>(b, a); else k_copy<<<256, 256>>>(a, b); });
    run("same through LDS with one barrier", [&](int i) { if (i & 1) k_copy_lds<<<256, 256>>>(b, a); else k_copy_lds<<<256, 256>>>(a, b); });
    run("copy 1 MiB (128 x 256 threads)", [&](int i) { if (i & 1) k_copy<<<128, 256>>>(b, a); else k_copy<<<128, 256>>>(a, b); });
    return 0;
}
