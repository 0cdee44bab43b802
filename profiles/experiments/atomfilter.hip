// scratch: throughput of returning 64-bit atomicExch at random slots of an HBM table (a "recently seen key" filter)
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
__global__ void k(unsigned long long* t, unsigned mask_bits, size_t n, unsigned long long* sink, int dup_gap) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // every key occurs twice, dup_gap apart in the stream
        size_t id = (i / (2 * (size_t)dup_gap)) * dup_gap + (i % dup_gap);
        unsigned long long x = id * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        unsigned long long old = atomicExch(&t[x >> (64 - mask_bits)], x);
        acc += (old == x);
    }
    if (acc) atomicAdd(sink, acc);
}
int main(int argc, char** argv) {
    size_t n = 380000000ull;
    unsigned long long *t, *sink; hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bits : {18, 20, 22, 23, 24, 26}) {
        hipMalloc(&t, 8ull << bits);
        for (int gap : {100000, 1000000}) {
            float best = 1e9f; unsigned long long hs = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipMemset(t, 0xff, 8ull << bits); hipMemset(sink, 0, 8);
                hipEventRecord(e0, 0);
                k<<<256 * 8, 256>>>(t, bits, n, sink, gap);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                hipMemcpy(&hs, sink, 8, hipMemcpyDeviceToHost);
            }
            printf("table 2^%d (%5.1f MB) dup gap %7d: %.3f ms for %zu M atomics (%.1f G/s), duplicates caught %.1f%%\n", bits, (8ull << bits) / 1e6, gap, best, n / 1000000, n / best / 1e6, 100.0 * hs / (n / 2));
            fflush(stdout);
        }
        hipFree(t);
    }
    return 0;
}
