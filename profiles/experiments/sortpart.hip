// does rocprim::radix_sort_keys honour begin_bit > 0 at every size?  (sort by bits [25, 55) only)
#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
static int run(unsigned B, unsigned E) {
    for (size_t n : {6160ul, 1000000ul}) {
        std::mt19937_64 rng(1);
        std::vector<uint64_t> h(n);
        for (auto& x : h) x = ((rng() % 200) << 56) | ((rng() % 1000) << 46) | ((rng() & 1 ? (1ull << 45) : (1ull << 24)) | (rng() & 0xFFFFFF));
        uint64_t *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
        hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice);
        hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        size_t tb = 0; void* tmp = nullptr;
        rocprim::radix_sort_keys(tmp, tb, a, b, n, B, E, s);
        hipMalloc(&tmp, tb);
        hipError_t e = rocprim::radix_sort_keys(tmp, tb, a, b, n, B, E, s);
        hipStreamSynchronize(s);
        std::vector<uint64_t> out(n);
        hipMemcpy(out.data(), b, n * 8, hipMemcpyDeviceToHost);
        bool ok = true; const uint64_t m = (E >= 64 ? ~0ull : ((1ull << E) - 1)) & ~((1ull << B) - 1);
        for (size_t i = 1; i < n; i++) if ((out[i - 1] & m) > (out[i] & m)) { ok = false; break; }
        std::sort(out.begin(), out.end()); std::sort(h.begin(), h.end());
        printf("[%u,%u) n=%zu err=%d sorted_by_masked_bits=%d same_multiset=%d\n", B, E, n, (int)e, (int)ok, (int)(out == h));
        hipFree(tmp); hipFree(a); hipFree(b);
    }
    return 0;
}
int main() { run(46, 64); run(40, 64); run(44, 64); run(46, 62); run(30, 48); run(25, 55); run(33, 64); run(32, 64); return 0; }
