// Reproducer for DESIGN.md 7 "Finding": the kLCP compare fused into emit_rows_kernel gave run-to-run differing bits on gfx950
// for rows >= 2^16 (round 1).  The kernel's kLCP part needs no real suffix array -- it compares the k - 1 bases at two positions
// taken from ANY array -- so this tool feeds the round-1 kernel text of random words and a random permutation, and holds its bits
// against (a) the stand-alone klcp_kernel of the product and (b) the host.  Variants switch parts of the fused kernel off to find
// what it takes to lose bits:  V0 the fused kernel as it was; V1 without the LDS counters / __syncthreads tail; V2 the kLCP compare
// alone with the same ballots; V3 as V0 but `klcp` written by every wave's lane 0 AFTER the barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/klcp_fused_repro.hip -o tools/klcp_fused_repro && ./tools/klcp_fused_repro [rows] [k] [runs]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

struct U4 {
    uint32_t a, c, g, t;
};
constexpr uint32_t kBlockDwords = 16;

__host__ __device__ inline uint64_t bases_at(const uint64_t* text2, uint64_t p) {
    const uint64_t w = p >> 5;
    const uint32_t sh = static_cast<uint32_t>(p & 31u) * 2u;
    uint64_t v = text2[w] >> sh;
    if (sh) v |= text2[w + 1] << (64u - sh);
    return v;
}

template <int V>
__global__ __launch_bounds__(256) void emit_rows_fused(const uint64_t* __restrict__ text2, const uint64_t* __restrict__ mask_text,
                                                       const uint32_t* __restrict__ sa_sorted, uint64_t len, int k, int all_ones_mask,
                                                       uint32_t* __restrict__ blocks, U4* __restrict__ block_counts, uint32_t* __restrict__ mask_counts,
                                                       uint64_t* __restrict__ klcp, uint32_t* __restrict__ dollar_row) {
    __shared__ uint32_t s_cnt[4][5];
    const uint64_t n = len + 1;
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t wave_in_block = threadIdx.x >> 6;
    uint32_t sym = 0, mbit = 0, kbit = 0;
    const bool valid = row < n;
    if (valid) {
        const uint64_t s = row == 0 ? len : sa_sorted[row - 1];
        if (V != 2) {
            if (s == 0) *dollar_row = static_cast<uint32_t>(row);
            else sym = static_cast<uint32_t>(text2[(s - 1) >> 5] >> (2 * ((s - 1) & 31))) & 3u;
            if (s != len) {
                if (all_ones_mask) mbit = s + static_cast<uint64_t>(k) <= len;
                else mbit = static_cast<uint32_t>(mask_text[s >> 6] >> (s & 63)) & 1u;
            }
        }
        if (klcp && row < len) {
            const uint64_t s2 = sa_sorted[row];
            const uint64_t km1 = static_cast<uint64_t>(k - 1);
            if (km1 <= len && s <= len - km1 && s2 <= len - km1) {
                const uint64_t m = km1 >= 32 ? ~0ull : ((1ull << (2 * km1)) - 1ull);
                kbit = km1 == 0 || ((bases_at(text2, s) ^ bases_at(text2, s2)) & m) == 0;
            }
        }
    }
    const uint64_t hi = __ballot(valid && (sym & 2u));
    const uint64_t lo = __ballot(valid && (sym & 1u));
    const uint64_t mm = __ballot(valid && mbit);
    const uint64_t kk = __ballot(valid && kbit);
    const uint64_t vv = __ballot(valid);
    if ((threadIdx.x & 63) == 0) {
        const uint64_t w = row >> 6;
        if (vv) {
            if (V != 2) {
                uint32_t* blk = blocks + (w >> 1) * kBlockDwords;
                const uint32_t half = static_cast<uint32_t>(w & 1) * 2;
                blk[4 + half] = static_cast<uint32_t>(hi);
                blk[5 + half] = static_cast<uint32_t>(hi >> 32);
                blk[8 + half] = static_cast<uint32_t>(lo);
                blk[9 + half] = static_cast<uint32_t>(lo >> 32);
                blk[12 + half] = static_cast<uint32_t>(mm);
                blk[13 + half] = static_cast<uint32_t>(mm >> 32);
            }
            if (klcp && V != 3) klcp[w] = kk;
        }
        if (V == 0 || V == 3) {
            s_cnt[wave_in_block][0] = __popcll(vv & ~hi & ~lo);
            s_cnt[wave_in_block][1] = __popcll(vv & ~hi & lo);
            s_cnt[wave_in_block][2] = __popcll(hi & ~lo);
            s_cnt[wave_in_block][3] = __popcll(hi & lo);
            s_cnt[wave_in_block][4] = __popcll(mm);
        }
    }
    if (V == 0 || V == 3) {
        __syncthreads();
        if (threadIdx.x < 2) {
            const uint64_t b = static_cast<uint64_t>(blockIdx.x) * 2 + threadIdx.x;
            const uint32_t w0 = threadIdx.x * 2, w1 = w0 + 1;
            block_counts[b] = U4{s_cnt[w0][0] + s_cnt[w1][0], s_cnt[w0][1] + s_cnt[w1][1], s_cnt[w0][2] + s_cnt[w1][2], s_cnt[w0][3] + s_cnt[w1][3]};
            mask_counts[b] = s_cnt[w0][4] + s_cnt[w1][4];
        }
        if (V == 3 && (threadIdx.x & 63) == 0 && vv && klcp) klcp[row >> 6] = kk;
    }
}

// the product's stand-alone kernel (build_index.hip), k <= 33
__global__ __launch_bounds__(256) void klcp_alone(const uint64_t* __restrict__ text2, const uint32_t* __restrict__ sa_sorted, uint64_t len, int k,
                                                  uint64_t* __restrict__ klcp) {
    const uint64_t row = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    bool kbit = false;
    if (row < len) {
        const uint64_t s = row == 0 ? len : sa_sorted[row - 1];
        const uint64_t s2 = sa_sorted[row];
        const uint64_t km1 = static_cast<uint64_t>(k - 1);
        if (km1 <= len && s <= len - km1 && s2 <= len - km1) {
            const uint64_t m = km1 >= 32 ? ~0ull : ((1ull << (2 * km1)) - 1ull);
            kbit = km1 == 0 || ((bases_at(text2, s) ^ bases_at(text2, s2)) & m) == 0;
        }
    }
    const uint64_t kk = __ballot(kbit);
    if ((threadIdx.x & 63) == 0 && row <= len) klcp[row >> 6] = kk;
}

static uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int V>
static int run_variant(const char* name, const uint64_t* d_text, const uint64_t* d_mask, const uint32_t* d_sa, uint64_t len, int k, uint32_t* d_blocks, U4* d_counts,
                       uint32_t* d_mcounts, uint64_t* d_klcp, uint32_t* d_dollar, const std::vector<uint64_t>& want, int runs) {
    const uint64_t n = len + 1, words = (n + 63) / 64;
    const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
    std::vector<uint64_t> got(words);
    int bad_runs = 0;
    for (int r = 0; r < runs; ++r) {
        CHECK(hipMemset(d_klcp, 0, (words + 1) * 8));
        hipLaunchKernelGGL((emit_rows_fused<V>), dim3(grid), dim3(256), 0, 0, d_text, d_mask, d_sa, len, k, 0, d_blocks, d_counts, d_mcounts, d_klcp, d_dollar);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), d_klcp, words * 8, hipMemcpyDeviceToHost));
        uint64_t diff_words = 0, lost = 0, extra = 0, first = ~0ull;
        for (uint64_t w = 0; w < words; ++w) {
            uint64_t wantw = want[w];
            if (w == words - 1 && (len & 63)) wantw &= (1ull << (len & 63)) - 1;   // (rows >= len carry no bit)
            const uint64_t g = (w == words - 1 && (len & 63)) ? (got[w] & ((1ull << (len & 63)) - 1)) : got[w];
            if (g != wantw) {
                ++diff_words;
                lost += __builtin_popcountll(wantw & ~g);
                extra += __builtin_popcountll(g & ~wantw);
                if (first == ~0ull) first = w * 64;
            }
        }
        if (diff_words) {
            ++bad_runs;
            printf("  %s run %d: %llu words differ (first row %llu), %llu bits lost, %llu extra\n", name, r, (unsigned long long)diff_words, (unsigned long long)first,
                   (unsigned long long)lost, (unsigned long long)extra);
        }
    }
    printf("%s: %d of %d runs differ from the host\n", name, bad_runs, runs);
    return bad_runs;
}

int main(int argc, char** argv) {
    const uint64_t len = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3000000ull;
    const int k = argc > 2 ? atoi(argv[2]) : 9;   // small k: many set bits even for a random "suffix array"
    const int runs = argc > 3 ? atoi(argv[3]) : 6;
    const uint64_t n = len + 1, t_words = (len >> 5) + 2, m_words = (len >> 6) + 2, words = (n + 63) / 64, n_blocks = ((n + 127) >> 7) + 1;
    uint64_t seed = 12345;
    std::vector<uint64_t> text(t_words, 0), mask(m_words, 0);
    for (uint64_t w = 0; w < t_words; ++w) text[w] = splitmix(seed) & splitmix(seed) & 0x5555555555555555ull;   // bases A/C only: k - 1 equal bases are common
    for (uint64_t p = len; p < t_words * 32; ++p) text[p >> 5] &= ~(3ull << (2 * (p & 31)));
    for (uint64_t w = 0; w < m_words; ++w) mask[w] = splitmix(seed);
    std::vector<uint32_t> sa(len);
    for (uint64_t i = 0; i < len; ++i) sa[i] = static_cast<uint32_t>(i);
    for (uint64_t i = len - 1; i > 0; --i) std::swap(sa[i], sa[splitmix(seed) % (i + 1)]);
    // host recomputation
    std::vector<uint64_t> want(words + 1, 0);
    const uint64_t km1 = static_cast<uint64_t>(k - 1), m = km1 >= 32 ? ~0ull : ((1ull << (2 * km1)) - 1ull);
    uint64_t ones = 0;
    for (uint64_t row = 0; row < len; ++row) {
        const uint64_t s = row == 0 ? len : sa[row - 1], s2 = sa[row];
        if (km1 <= len && s <= len - km1 && s2 <= len - km1 && (km1 == 0 || ((bases_at(text.data(), s) ^ bases_at(text.data(), s2)) & m) == 0)) {
            want[row >> 6] |= 1ull << (row & 63);
            ++ones;
        }
    }
    printf("rows %llu, k %d: %llu kLCP bits set on the host\n", (unsigned long long)n, k, (unsigned long long)ones);
    uint64_t *d_text, *d_mask, *d_klcp;
    uint32_t *d_sa, *d_blocks, *d_mcounts, *d_dollar;
    U4* d_counts;
    CHECK(hipMalloc(&d_text, t_words * 8));
    CHECK(hipMalloc(&d_mask, m_words * 8));
    CHECK(hipMalloc(&d_sa, len * 4));
    CHECK(hipMalloc(&d_klcp, (words + 1) * 8));
    CHECK(hipMalloc(&d_blocks, n_blocks * 64));
    CHECK(hipMalloc(&d_counts, (n_blocks + 1) * sizeof(U4)));
    CHECK(hipMalloc(&d_mcounts, (n_blocks + 1) * 4));
    CHECK(hipMalloc(&d_dollar, 4));
    CHECK(hipMemcpy(d_text, text.data(), t_words * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_mask, mask.data(), m_words * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_sa, sa.data(), len * 4, hipMemcpyHostToDevice));
    // the stand-alone kernel first
    {
        std::vector<uint64_t> got(words);
        CHECK(hipMemset(d_klcp, 0, (words + 1) * 8));
        hipLaunchKernelGGL(klcp_alone, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, 0, d_text, d_sa, len, k, d_klcp);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), d_klcp, words * 8, hipMemcpyDeviceToHost));
        uint64_t diff = 0;
        for (uint64_t w = 0; w < words; ++w) diff += got[w] != want[w];
        printf("stand-alone klcp_kernel: %llu words differ from the host\n", (unsigned long long)diff);
    }
    // Phase B -- the round-1 FLOW around the kernel: the suffix array sat on the device with its tied rows still unordered, the host
    // ordered every group of tied rows and patched it back with one small hipMemcpy per group from a std::vector (pageable memory),
    // and the fused kernel was launched right behind the last patch.  Here: groups of 2..9 consecutive rows are shuffled on the
    // device copy, the host patches the final order back group by group, the kernel runs at once.
    {
        std::vector<uint32_t> shuffled = sa;
        std::vector<std::pair<uint64_t, uint32_t>> groups;
        for (uint64_t r = 0; r + 10 < len;) {
            const uint32_t cnt = 2 + static_cast<uint32_t>(splitmix(seed) % 8);
            for (uint32_t i = cnt - 1; i > 0; --i) std::swap(shuffled[r + i], shuffled[r + splitmix(seed) % (i + 1)]);
            groups.emplace_back(r, cnt);
            r += cnt + splitmix(seed) % 40;
        }
        const uint32_t grid = static_cast<uint32_t>((n + 255) / 256);
        std::vector<uint64_t> got(words);
        std::vector<uint32_t> grp;
        int bad_b = 0;
        for (int r = 0; r < runs; ++r) {
            CHECK(hipMemcpy(d_sa, shuffled.data(), len * 4, hipMemcpyHostToDevice));
            CHECK(hipMemset(d_klcp, 0, (words + 1) * 8));
            CHECK(hipDeviceSynchronize());
            for (const auto& g : groups) {
                grp.assign(sa.begin() + g.first, sa.begin() + g.first + g.second);
                CHECK(hipMemcpy(d_sa + g.first, grp.data(), g.second * 4, hipMemcpyHostToDevice));
            }
            hipLaunchKernelGGL((emit_rows_fused<0>), dim3(grid), dim3(256), 0, 0, d_text, d_mask, d_sa, len, k, 0, d_blocks, d_counts, d_mcounts, d_klcp, d_dollar);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(got.data(), d_klcp, words * 8, hipMemcpyDeviceToHost));
            uint64_t diff = 0, first = ~0ull;
            for (uint64_t w = 0; w < words; ++w)
                if (got[w] != want[w]) {
                    ++diff;
                    if (first == ~0ull) first = w * 64;
                }
            if (diff) {
                ++bad_b;
                printf("  phase B run %d: %llu words differ (first row %llu)\n", r, (unsigned long long)diff, (unsigned long long)first);
            }
        }
        printf("phase B (%zu small pageable H2D patches, then the fused kernel): %d of %d runs differ from the host\n", groups.size(), bad_b, runs);
        CHECK(hipMemcpy(d_sa, sa.data(), len * 4, hipMemcpyHostToDevice));
    }
    int bad = 0;
    bad += run_variant<0>("V0 fused (round 1)", d_text, d_mask, d_sa, len, k, d_blocks, d_counts, d_mcounts, d_klcp, d_dollar, want, runs);
    bad += run_variant<1>("V1 no LDS tail", d_text, d_mask, d_sa, len, k, d_blocks, d_counts, d_mcounts, d_klcp, d_dollar, want, runs);
    bad += run_variant<2>("V2 compare + ballots only", d_text, d_mask, d_sa, len, k, d_blocks, d_counts, d_mcounts, d_klcp, d_dollar, want, runs);
    bad += run_variant<3>("V3 klcp stored after barrier", d_text, d_mask, d_sa, len, k, d_blocks, d_counts, d_mcounts, d_klcp, d_dollar, want, runs);
    return bad ? 1 : 0;
}
