// micro-benchmark: rocPRIM onesweep radix sort of 64-bit keys with different digit widths / tile shapes
#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
template <class Cfg> static void run(const char* name, size_t n, uint64_t* src, uint64_t* a, uint64_t* b, int top, const std::vector<uint64_t>& ref) {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    size_t tb = 0; void* tmp = nullptr;
    rocprim::radix_sort_keys<Cfg>(tmp, tb, a, b, n, 0u, (unsigned)top, s);
    hipMalloc(&tmp, tb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 4; it++) {
        hipMemcpyAsync(a, src, n * 8, hipMemcpyDeviceToDevice, s);
        hipEventRecord(e0, s);
        hipError_t e = rocprim::radix_sort_keys<Cfg>(tmp, tb, a, b, n, 0u, (unsigned)top, s);
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        if (e != hipSuccess) { printf("%s err %d\n", name, (int)e); return; }
    }
    std::vector<uint64_t> out(n); hipMemcpy(out.data(), b, n * 8, hipMemcpyDeviceToHost);
    printf("%-28s top=%d n=%zu  %.3f ms  ok=%d tmp=%zu MB\n", name, top, n, best, (int)(out == ref), tb >> 20);
    hipFree(tmp);
}
int main() {
    const size_t n = 29500000;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> h(n);
    for (auto& x : h) x = ((rng() % 33472) << 48) | ((rng() % 5000) << 35) | ((1ull << 24) | (rng() & 0xFFFFFF));
    uint64_t *src, *a, *b; hipMalloc(&src, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemcpy(src, h.data(), n * 8, hipMemcpyHostToDevice);
    std::sort(h.begin(), h.end());
    using namespace rocprim;
    run<default_config>("default", n, src, a, b, 64, h);
#define CFG(T, I, B) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<T, I>, kernel_config<T, I>, B, block_radix_rank_algorithm::match>>
    run<CFG(512, 12, 8)>("512x12 b8", n, src, a, b, 64, h);
    run<CFG(512, 12, 7)>("512x12 b7", n, src, a, b, 64, h);
    run<CFG(256, 16, 8)>("256x16 b8", n, src, a, b, 64, h);
    run<CFG(1024, 8, 8)>("1024x8 b8", n, src, a, b, 64, h);
    run<CFG(512, 16, 8)>("512x16 b8", n, src, a, b, 64, h);
    run<CFG(512, 12, 9)>("512x12 b9", n, src, a, b, 64, h);
    run<CFG(1024, 12, 9)>("1024x12 b9", n, src, a, b, 64, h);
    return 0;
}
