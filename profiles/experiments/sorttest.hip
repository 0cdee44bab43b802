#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
int main() {
    for (size_t n : {1000ul, 55522ul, 3000000ul}) {
        std::mt19937_64 rng(1);
        std::vector<uint64_t> h(n);
        for (auto& x : h) x = ((rng() % 200) << 56) | ((rng() % 100) << 49) | ((1ull << 24) | (rng() & 0xFFFFFF));
        uint64_t *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
        hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice);
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        int ranges[2][2] = {{0, 25}, {49, 64}};
        for (auto& r : ranges) {
            size_t tb = 0; void* tmp = nullptr;
            rocprim::radix_sort_keys(tmp, tb, a, b, n, (unsigned)r[0], (unsigned)r[1], s);
            hipMalloc(&tmp, tb);
            hipError_t e = rocprim::radix_sort_keys(tmp, tb, a, b, n, (unsigned)r[0], (unsigned)r[1], s);
            hipStreamSynchronize(s);
            printf("n=%zu range [%d,%d) tmp=%zu err=%d\n", n, r[0], r[1], tb, (int)e);
            hipFree(tmp); std::swap(a, b);
        }
        std::vector<uint64_t> out(n);
        hipMemcpy(out.data(), a, n * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("n=%zu sorted_ok=%d\n", n, (int)(out == h));
    }
    return 0;
}
