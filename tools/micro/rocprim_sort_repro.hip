// Stand-alone check of the suspicion behind fmsi_sorted.hip's SortConfig: rocprim::radix_sort_pairs over double buffers with
// (begin_bit, end_bit) = (32, 64) -- keys compared by their upper halves only -- for sizes on both sides of the merge-sort window (1 K .. 1 M
// items), with the default configuration and with the merge-sort path switched off.  For each: are the pairs sorted by the compared bits,
// is the order of equal keys the input's (stability -- the sorted walk relies on it), do two runs agree?
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/micro/rocprim_sort_repro.hip -o tools/micro/rocprim_sort_repro && tools/micro/rocprim_sort_repro
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

using NoMerge = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

template <class Config>
int run(size_t n, unsigned begin_bit, unsigned end_bit, unsigned distinct_bits, const char* name, int* bad_total) {
    std::mt19937_64 rng(12345 + n);
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> vals(n);
    // few distinct compared values (many ties), random lower halves
    for (size_t i = 0; i < n; ++i) keys[i] = ((rng() & ((1ull << distinct_bits) - 1)) << begin_bit) | (rng() & ((1ull << begin_bit) - 1)) | (end_bit < 64 ? (rng() << end_bit) : 0ull);
    std::iota(vals.begin(), vals.end(), 0u);
    uint64_t *k0, *k1;
    uint32_t *v0, *v1;
    CHECK(hipMalloc(&k0, n * 8)); CHECK(hipMalloc(&k1, n * 8)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4));
    std::vector<uint32_t> first;
    int bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemcpy(k0, keys.data(), n * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(v0, vals.data(), n * 4, hipMemcpyHostToDevice));
        rocprim::double_buffer<uint64_t> kb(k0, k1);
        rocprim::double_buffer<uint32_t> vb(v0, v1);
        size_t temp_bytes = 0;
        CHECK((rocprim::radix_sort_pairs<Config>(nullptr, temp_bytes, kb, vb, n, begin_bit, end_bit, 0)));
        void* temp;
        CHECK(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
        CHECK((rocprim::radix_sort_pairs<Config>(temp, temp_bytes, kb, vb, n, begin_bit, end_bit, 0)));
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> ko(n);
        std::vector<uint32_t> vo(n);
        CHECK(hipMemcpy(ko.data(), kb.current(), n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(vo.data(), vb.current(), n * 4, hipMemcpyDeviceToHost));
        CHECK(hipFree(temp));
        const uint64_t mask = (end_bit == 64 ? ~0ull : ((1ull << end_bit) - 1)) & ~((1ull << begin_bit) - 1);
        size_t unsorted = 0, unstable = 0, mismatched = 0;
        for (size_t i = 0; i < n; ++i) {
            if (ko[i] != keys[vo[i]]) ++mismatched;
            if (i && (ko[i - 1] & mask) > (ko[i] & mask)) ++unsorted;
            if (i && (ko[i - 1] & mask) == (ko[i] & mask) && vo[i - 1] > vo[i]) ++unstable;
        }
        size_t differs = 0;
        if (rep == 0) first = vo;
        else for (size_t i = 0; i < n; ++i) differs += first[i] != vo[i];
        if (unsorted || unstable || mismatched || differs) {
            ++bad;
            printf("  %-9s n=%-9zu bits [%u,%u) rep %d: key/value pairs torn %zu, out of order %zu, ties out of input order %zu, differs from rep 0 at %zu\n", name, n, begin_bit, end_bit, rep,
                   mismatched, unsorted, unstable, differs);
        }
    }
    if (!bad) printf("  %-9s n=%-9zu bits [%u,%u) ok (sorted, stable, three runs alike)\n", name, n, begin_bit, end_bit);
    *bad_total += bad;
    CHECK(hipFree(k0)); CHECK(hipFree(k1)); CHECK(hipFree(v0)); CHECK(hipFree(v1));
    return 0;
}

int main() {
    int bad = 0;
    for (size_t n : {600ul, 2000ul, 5000ul, 40000ul, 300000ul, 1000000ul, 3000000ul}) {
        for (unsigned bits : {6u, 20u}) {
            if (run<rocprim::default_config>(n, 32, 64, bits, "default", &bad)) return 2;
            if (run<NoMerge>(n, 32, 64, bits, "no-merge", &bad)) return 2;
        }
        if (run<rocprim::default_config>(n, 40, 64, 12, "default", &bad)) return 2;
        if (run<rocprim::default_config>(n, 0, 64, 30, "default", &bad)) return 2;
        // (what the index builder and the dictionaries ask for: the LOW bits, a partial range from bit 0)
        if (run<rocprim::default_config>(n, 0, 60, 30, "default", &bad)) return 2;
        if (run<rocprim::default_config>(n, 0, 44, 30, "default", &bad)) return 2;
    }
    printf("%s\n", bad ? "FAULTS SEEN" : "no fault seen");
    return bad ? 1 : 0;
}
