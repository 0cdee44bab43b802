// scratch: rocPRIM onesweep configurations for the basefc partial sort (64-bit keys, ~30 sorted bits, 200 M keys).
// usage: sortcfg [n_keys]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

__global__ void fill(uint64_t* k, size_t n, int begin) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    // rows roughly follow the stream position (coordinate-sorted reads), cells and UMI bits random
    uint64_t row = (uint64_t)((double)i / n * 33000.0) + (x & 63);
    uint64_t cell = (x >> 8) % 10000;
    k[i] = ((row * 16384 + cell) << begin) | ((x >> 24) & ((1ull << begin) - 1));
}
__global__ void check(const uint64_t* k, size_t n, int begin, unsigned* bad) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    if ((k[i] >> begin) > (k[i + 1] >> begin)) atomicAdd(bad, 1u);
}

template <class Cfg> static void run(const char* name, uint64_t* src, uint64_t* a, uint64_t* b, size_t n, int begin, int top) {
    size_t tb = 0;
    rocprim::radix_sort_keys<Cfg>(nullptr, tb, a, b, n, (unsigned)begin, (unsigned)top, (hipStream_t)0);
    void* tmp; hipMalloc(&tmp, tb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipMemcpy(a, src, n * 8, hipMemcpyDeviceToDevice);
        hipEventRecord(e0, 0);
        hipError_t er = rocprim::radix_sort_keys<Cfg>(tmp, tb, a, b, n, (unsigned)begin, (unsigned)top, (hipStream_t)0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (er != hipSuccess) { printf("%s: error %d\n", name, (int)er); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    check<<<(unsigned)((n + 255) / 256), 256>>>(b, n, begin, bad);
    unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("%-28s bits [%d,%d): %.3f ms  (%.1f Gkeys/s)  out-of-order=%u  tmp=%.1f MB\n", name, begin, top, best, n / best / 1e6, hb, tb / 1e6);
    fflush(stdout);
    hipFree(tmp); hipFree(bad);
}

template <class Cfg, class V> static void runp(const char* name, uint64_t* src, uint64_t* a, uint64_t* b, V* va, V* vb, size_t n, int top) {
    size_t tb = 0;
    rocprim::radix_sort_pairs<Cfg>(nullptr, tb, a, b, va, vb, n, 0u, (unsigned)top, (hipStream_t)0);
    void* tmp; hipMalloc(&tmp, tb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipMemcpy(a, src, n * 8, hipMemcpyDeviceToDevice);
        hipEventRecord(e0, 0);
        hipError_t er = rocprim::radix_sort_pairs<Cfg>(tmp, tb, a, b, va, vb, n, 0u, (unsigned)top, (hipStream_t)0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        if (er != hipSuccess) { printf("%s: error %d\n", name, (int)er); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s pairs(%zu B values) n=%zu bits [0,%d): %.3f ms\n", name, sizeof(V), n, top, best); fflush(stdout);
    hipFree(tmp);
}
using namespace rocprim;
template <unsigned HB, unsigned HI, unsigned SB, unsigned SI>
using pcfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<SB, SI>, 8, block_radix_rank_algorithm::match>>;
template <unsigned HB, unsigned HI>
using hcfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<1024, 8>, 8, block_radix_rank_algorithm::match>>;
template <unsigned BS, unsigned IPT, unsigned BITS, block_radix_rank_algorithm A = block_radix_rank_algorithm::match>
using cfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, BITS, A>>;

int main(int argc, char** argv) {
    size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 200000000ull;
    int begin = 24, top = 24 + 30;
    uint64_t *src, *a, *b; hipMalloc(&src, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    fill<<<(unsigned)((n + 255) / 256), 256>>>(src, n, begin);
    hipDeviceSynchronize();
    {
        size_t m = 72000000; uint64_t* v; hipMalloc(&v, 2 * m * 8);
        runp<default_config, uint64_t>("default", src, a, b, v, v + m, m, 56);
        runp<pcfg<512, 32, 1024, 8>, uint64_t>("hist 512x32 sort 1024x8", src, a, b, v, v + m, m, 56);
        runp<pcfg<512, 32, 512, 12>, uint64_t>("hist 512x32 sort 512x12", src, a, b, v, v + m, m, 56);
        runp<pcfg<512, 32, 512, 16>, uint64_t>("hist 512x32 sort 512x16", src, a, b, v, v + m, m, 56);
        runp<pcfg<512, 32, 1024, 6>, uint64_t>("hist 512x32 sort 1024x6", src, a, b, v, v + m, m, 56);
        size_t m2 = 25000000; uint8_t* w = (uint8_t*)v;
        runp<default_config, uint8_t>("default", src, a, b, w, w + m2, m2, 56);
        runp<pcfg<512, 32, 1024, 8>, uint8_t>("hist 512x32 sort 1024x8", src, a, b, w, w + m2, m2, 56);
        runp<pcfg<512, 32, 512, 16>, uint8_t>("hist 512x32 sort 512x16", src, a, b, w, w + m2, m2, 56);
        runp<pcfg<512, 32, 512, 12>, uint8_t>("hist 512x32 sort 512x12", src, a, b, w, w + m2, m2, 56);
    }
    return 0;
}
