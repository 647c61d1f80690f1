// C++ caller of include/rdst.hpp, written like the reference's own unit tests
// (src/radix_sort.rs:146-340: random input, compare with the standard library's sort).
// Built and run by tests/test_cpp_mirror.py.  Exit code 0 = all checks passed.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "rdst.hpp"

template <typename T, typename U>
static bool same_bits(const std::vector<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}

// one line per check on stderr, flushed, BEFORE the call: an uncaught rdst::Error aborts the process at the first
// throw, and the log then names the call that threw (type, length) and everything that ran before it
#define PROGRESS(what, n) do { std::fprintf(stderr, "[check] %s n=%zu\n", what, (std::size_t)(n)); std::fflush(stderr); } while (0)

template <typename T>
static int check_int(std::size_t n, unsigned seed) {
    PROGRESS(__PRETTY_FUNCTION__, n);
    std::mt19937_64 rng(seed);
    std::vector<T> v(n);
    for (auto& x : v) x = static_cast<T>(rng());
    std::vector<T> expect = v;
    std::sort(expect.begin(), expect.end());  // reference oracle: std sort (src/test_utils.rs:119)
    rdst::radix_sort_unstable(v);
    return same_bits<T, T>(v, expect) ? 0 : 1;
}

template <typename T, typename U>
static int check_float(std::size_t n, unsigned seed) {
    PROGRESS(__PRETTY_FUNCTION__, n);
    std::mt19937_64 rng(seed);
    std::vector<T> v(n);
    for (auto& x : v) { U b = static_cast<U>(rng()); std::memcpy(&x, &b, sizeof b); }  // any bit pattern
    std::vector<T> expect = v;
    auto key = [](T f) { U u; std::memcpy(&u, &f, sizeof u); const U msb = U(1) << (sizeof(U) * 8 - 1); return (u & msb) ? U(~u) : U(u ^ msb); };
    std::sort(expect.begin(), expect.end(), [&](T a, T b) { return key(a) < key(b); });  // total_cmp order (src/radix_sort.rs:97-144)
    rdst::radix_sort_builder(v).with_parallel(false).sort();
    return same_bits<T, U>(v, expect) ? 0 : 1;
}

// benches/struct_sort.rs:11-27: a large struct sorted by one f32 field
struct LargeStruct {
    std::uint64_t a;
    float sort_key;
    std::uint32_t id;
    double payload[5];
};

static int check_struct(std::size_t n, unsigned seed) {
    PROGRESS("check_struct", n);
    std::mt19937_64 rng(seed);
    std::vector<LargeStruct> v(n);
    for (std::size_t i = 0; i < n; ++i) {
        const std::uint32_t bits = static_cast<std::uint32_t>(rng()) & 0xC07FFFFFu;  // few exponents: many equal keys
        std::memcpy(&v[i].sort_key, &bits, 4);
        v[i].a = rng();
        v[i].id = static_cast<std::uint32_t>(i);
        for (double& p : v[i].payload) p = static_cast<double>(rng() % 1000);
    }
    std::vector<LargeStruct> expect = v;
    auto key = [](float f) { std::uint32_t u; std::memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u ^ 0x80000000u); };
    std::stable_sort(expect.begin(), expect.end(), [&](const LargeStruct& x, const LargeStruct& y) { return key(x.sort_key) < key(y.sort_key); });
    rdst::radix_sort_unstable_by_field(v, &LargeStruct::sort_key);
    return std::memcmp(v.data(), expect.data(), n * sizeof(LargeStruct)) == 0 ? 0 : 1;
}

template <std::size_t N>
static int check_bytes(std::size_t n, unsigned seed) {  // [u8; N] sorts lexicographically (src/radix_sort.rs:221-229)
    PROGRESS(__PRETTY_FUNCTION__, n);
    std::mt19937_64 rng(seed);
    std::vector<std::array<std::uint8_t, N>> v(n);
    for (auto& a : v) for (auto& b : a) b = static_cast<std::uint8_t>(rng() % 5 ? rng() : 0);
    auto expect = v;
    std::sort(expect.begin(), expect.end());
    rdst::radix_sort_unstable(v);
    return v == expect ? 0 : 1;
}

int main() {
    int bad = 0;
    std::vector<std::array<std::uint8_t, 3>> arr = {{1, 2, 3}, {3, 2, 1}, {3, 3, 3}, {0, 255, 7}};  // src/radix_sort.rs:221-229
    rdst::radix_sort_unstable(arr);
    bad += !(arr == std::vector<std::array<std::uint8_t, 3>>{{0, 255, 7}, {1, 2, 3}, {3, 2, 1}, {3, 3, 3}});
    for (std::size_t n : {0ul, 1ul, 2ul, 1000ul, 300007ul})
        bad += check_bytes<1>(n, 21) + check_bytes<3>(n, 22) + check_bytes<4>(n, 23) + check_bytes<7>(n, 24) + check_bytes<8>(n, 25) + check_bytes<11>(n, 26) + check_bytes<16>(n, 27);
    for (std::size_t n : {0ul, 1ul, 2ul, 1000ul, 250001ul}) bad += check_struct(n, 11);
    std::vector<std::uint32_t> doc = {3, 1, 2};  // src/radix_sort.rs:11-14
    rdst::radix_sort_unstable(doc);
    bad += !(doc == std::vector<std::uint32_t>{1, 2, 3});
    std::vector<std::uint32_t> simple = {55, 22, 73, 4, 89, 0, 100, 3};  // examples/simple_usage.rs:4-7
    rdst::radix_sort_unstable(simple);
    bad += !(simple == std::vector<std::uint32_t>{0, 3, 4, 22, 55, 73, 89, 100});
    for (std::size_t n : {0ul, 1ul, 2ul, 129ul, 100000ul, 3000001ul}) {
        bad += check_int<std::uint32_t>(n, 1) + check_int<std::uint64_t>(n, 2) + check_int<std::int32_t>(n, 3) + check_int<std::int64_t>(n, 4);
        bad += check_float<float, std::uint32_t>(n, 5) + check_float<double, std::uint64_t>(n, 6);
        bad += check_int<std::uint8_t>(n, 7) + check_int<std::int16_t>(n, 8);
    }
    // with_low_mem_tuner(): the device's low-memory route (src/radix_sort_builder.rs:74-77), same answer
    {
        PROGRESS("low-memory route", 3000001);
        std::mt19937_64 rng(99);
        std::vector<std::uint64_t> v(3000001);
        for (auto& x : v) x = rng();
        auto expect = v;
        std::sort(expect.begin(), expect.end());
        rdst::radix_sort_builder(v).with_low_mem_tuner().sort();
        bad += !(v == expect);
    }
    // a CPU tuner cannot be honoured here: throws, data untouched
    std::vector<std::uint32_t> keep = {9, 8, 7, 6};
    try { rdst::radix_sort_builder(keep).with_single_threaded_tuner().sort(); ++bad; } catch (const rdst::Error&) {}
    bad += !(keep == std::vector<std::uint32_t>{9, 8, 7, 6});
    rdst::tuner::GpuTuner gpu(2);
    rdst::radix_sort_builder(keep).with_tuner(&gpu).sort();
    bad += !(keep == std::vector<std::uint32_t>{6, 7, 8, 9});
    std::printf("%s (%d failed checks)\n", bad ? "FAIL" : "ok", bad);
    return bad ? 1 : 0;
}
