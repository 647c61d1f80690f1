// rdst.hpp — header-only C++ mirror of rdst's public surface for the device route.
//
// The reference is a Rust crate; where its toolchain is absent the host side above the C ABI
// (include/rdst_hip.h) is C++ with the reference's names, argument meaning and error behaviour:
//
//   rdst::radix_sort_unstable(v)                        RadixSort::radix_sort_unstable     src/radix_sort.rs:21-45
//   rdst::radix_sort_unstable_by_field(v, &T::key)      the same on structs keyed by a field benches/struct_sort.rs:11-27
//   rdst::radix_sort_builder(v).with_*().sort()         RadixSortBuilder                   src/radix_sort_builder.rs:8-158
//   rdst::RadixKey<T>::LEVELS / kind                    RadixKey for the built-in types    src/radix_key_impl.rs:1-185
//   rdst::tuner::{Tuner, TuningParams, Algorithm, ...}  pub mod tuner                      src/tuner.rs:1-40, src/tuners/*.rs
//
// `v` is a std::vector<T> or a (T*, len) slice of a built-in key type.  Sorting is in place and
// blocking, like the reference.  The reference's `sort()` is infallible; this route can fail
// (no device, out of device memory, ...): on failure the slice is untouched and rdst::Error is
// thrown, so a caller can still run the reference's CPU path on the same data.  There is no CPU
// fallback inside this header — the CPU algorithms stay in the reference crate.
#ifndef RDST_HPP
#define RDST_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <array>
#include <vector>

#include "rdst_hip.h"

namespace rdst {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

// RadixKey (src/radix_key.rs:1-5) for the built-in types whose mapping the device knows.
template <typename T, typename = void> struct RadixKey;  // not defined: no device mapping for T
template <typename T>
struct RadixKey<T, std::enable_if_t<std::is_integral_v<T> && !std::is_same_v<T, bool> && sizeof(T) <= 8>> {
    static constexpr std::size_t LEVELS = sizeof(T);
    static constexpr rdst_key_kind kind = std::is_signed_v<T> ? RDST_KEY_SIGNED : RDST_KEY_UNSIGNED;
};
template <> struct RadixKey<float> { static constexpr std::size_t LEVELS = 4; static constexpr rdst_key_kind kind = RDST_KEY_FLOAT; };
template <> struct RadixKey<double> { static constexpr std::size_t LEVELS = 8; static constexpr rdst_key_kind kind = RDST_KEY_FLOAT; };
template <std::size_t N>
struct RadixKey<std::array<std::uint8_t, N>, std::enable_if_t<(N >= 1 && N <= 16)>> {  // [u8; N], src/radix_key_impl.rs:78-85: lexicographic
    static constexpr std::size_t LEVELS = N;
    static constexpr rdst_key_kind kind = RDST_KEY_BYTES_BE;
};

namespace tuner {

enum class Algorithm : int {  // src/tuner.rs:12-22 + the device routes
    MtOop = RDST_ALGO_MT_OOP, MtLsb = RDST_ALGO_MT_LSB, Scanning = RDST_ALGO_SCANNING,
    Recombinating = RDST_ALGO_RECOMBINATING, Comparative = RDST_ALGO_COMPARATIVE, LrLsb = RDST_ALGO_LR_LSB,
    Lsb = RDST_ALGO_LSB, Regions = RDST_ALGO_REGIONS, Ska = RDST_ALGO_SKA,
    GpuLsd = RDST_ALGO_GPU_LSD, GpuSharded = RDST_ALGO_GPU_SHARDED
};

struct TuningParams {  // src/tuner.rs:2-8; parent_len < 0 encodes None
    std::size_t threads, level, total_levels, input_len;
    std::int64_t parent_len;
};

struct Tuner {  // src/tuner.rs:33-35
    virtual ~Tuner() = default;
    virtual Algorithm pick_algorithm(const TuningParams& p, const std::uint64_t (&counts)[256]) const = 0;
};

namespace detail {
inline Algorithm table(int id, const TuningParams& p, const std::uint64_t (&counts)[256], std::uint64_t gpu_min_len) {
    rdst_tuning_params c{p.threads, p.level, p.total_levels, p.input_len, p.parent_len};
    const int r = rdst_pick_algorithm(id, &c, counts, gpu_min_len);
    if (r < 0) throw Error(r, "rdst_pick_algorithm rejected its arguments");
    return static_cast<Algorithm>(r);
}
}  // namespace detail

struct StandardTuner : Tuner {  // src/tuners/standard_tuner.rs:10-64
    Algorithm pick_algorithm(const TuningParams& p, const std::uint64_t (&c)[256]) const override { return detail::table(RDST_TUNER_STANDARD, p, c, 0); }
};
struct LowMemoryTuner : Tuner {  // src/tuners/low_memory_tuner.rs:13-43
    Algorithm pick_algorithm(const TuningParams& p, const std::uint64_t (&c)[256]) const override { return detail::table(RDST_TUNER_LOW_MEMORY, p, c, 0); }
};
struct SingleThreadedTuner : Tuner {  // src/tuners/single_threaded_tuner.rs:13-43
    Algorithm pick_algorithm(const TuningParams& p, const std::uint64_t (&c)[256]) const override { return detail::table(RDST_TUNER_SINGLE_THREADED, p, c, 0); }
};
constexpr std::uint64_t kGpuMinLenHostSlice = RDST_GPU_MIN_LEN_HOST_SLICE;  // where a host slice is better off on the device
struct GpuTuner : Tuner {  // StandardTuner, except whole top-level slices of >= gpu_min_len elements go to the device
    std::uint64_t gpu_min_len = 0;
    explicit GpuTuner(std::uint64_t min_len = 0) : gpu_min_len(min_len) {}
    Algorithm pick_algorithm(const TuningParams& p, const std::uint64_t (&c)[256]) const override { return detail::table(RDST_TUNER_GPU, p, c, gpu_min_len); }
};

}  // namespace tuner

template <typename T>
class RadixSortBuilder {  // src/radix_sort_builder.rs:8-158
    T* data_;
    std::size_t len_;
    bool multi_threaded_ = true;          // accepted for source compatibility; the device route has no use for it
    bool device_default_ = true;          // no tuner chosen: the slice goes to the device
    bool low_memory_ = false;             // with_low_mem_tuner(): the device's low-memory route
    const tuner::Tuner* tuner_ = nullptr;

   public:
    RadixSortBuilder(T* data, std::size_t len) : data_(data), len_(len) {
        static_assert(RadixKey<T>::LEVELS != 0, "RadixKey must have at least 1 level");  // radix_sort_builder.rs:22
    }
    RadixSortBuilder& with_parallel(bool parallel) { multi_threaded_ = parallel; return *this; }
    RadixSortBuilder& with_tuner(const tuner::Tuner* t) { tuner_ = t; device_default_ = false; low_memory_ = false; return *this; }  // the last with_*tuner call wins (radix_sort_builder.rs:53-147)
    // with_low_mem_tuner (src/radix_sort_builder.rs:74-77) trades speed for memory in the reference (Ska / Regions instead of
    // the out-of-place sorts): so does the device route — the keys and a scratch of len / 64 elements instead of two arrays
    // (rdst_hip_sort_device_lowmem behind rdst_hip_opts::low_memory).  with_single_threaded_tuner selects among the
    // reference's CPU algorithms, which are not shipped here: like a user tuner that answers with a CPU algorithm, sort() throws.
    RadixSortBuilder& with_low_mem_tuner() { low_memory_ = true; tuner_ = nullptr; device_default_ = true; return *this; }
    RadixSortBuilder& with_single_threaded_tuner() { static const tuner::SingleThreadedTuner t; return with_tuner(&t); }

    void sort() {
        if (len_ <= 1) return;  // radix_sort_builder.rs:151
        if (!device_default_) {
            // Sorter::handle_chunk hands the tuner the top-level histogram (src/sorter.rs:67-76);
            // a host-side count of one level is cheap next to the transfer that follows.
            std::uint64_t counts[256] = {};
            const int top = static_cast<int>(RadixKey<T>::LEVELS) - 1;
            for (std::size_t i = 0; i < len_; ++i) ++counts[top_digit(data_[i], top)];
            const tuner::TuningParams p{1, static_cast<std::size_t>(top), RadixKey<T>::LEVELS, len_, -1};
            const auto a = tuner_->pick_algorithm(p, counts);
            if (a != tuner::Algorithm::GpuLsd && a != tuner::Algorithm::GpuSharded)
                throw Error(RDST_ERR_UNSUPPORTED, "tuner picked a CPU algorithm: those stay in the reference crate; this header ships the device route only");
        }
        rdst_hip_opts opts{-1, low_memory_ ? 1 : 0, 0};
        const int rc = rdst_hip_sort(data_, len_, sizeof(T), RadixKey<T>::kind, RadixKey<T>::LEVELS, &opts);
        if (rc != RDST_OK) throw Error(rc, rdst_hip_last_error());
    }

   private:
    static std::uint8_t top_digit(const T& v, int level) {  // RadixKey::get_level (src/radix_key_impl.rs)
        if constexpr (RadixKey<T>::kind == RDST_KEY_BYTES_BE) {
            return reinterpret_cast<const std::uint8_t*>(&v)[RadixKey<T>::LEVELS - 1 - static_cast<std::size_t>(level)];
        } else {
        using U = std::conditional_t<sizeof(T) == 1, std::uint8_t, std::conditional_t<sizeof(T) == 2, std::uint16_t,
                  std::conditional_t<sizeof(T) == 4, std::uint32_t, std::uint64_t>>>;
        U u;
        __builtin_memcpy(&u, &v, sizeof u);
        constexpr U msb = U(U(1) << (sizeof(U) * 8 - 1));
        if (RadixKey<T>::kind == RDST_KEY_SIGNED) u = U(u ^ msb);
        else if (RadixKey<T>::kind == RDST_KEY_FLOAT) u = U(u ^ ((u & msb) ? U(~U(0)) : msb));
        return static_cast<std::uint8_t>(u >> (level * 8));
        }
    }
};

template <typename T> RadixSortBuilder<T> radix_sort_builder(std::vector<T>& v) { return RadixSortBuilder<T>(v.data(), v.size()); }
template <typename T> RadixSortBuilder<T> radix_sort_builder(T* data, std::size_t len) { return RadixSortBuilder<T>(data, len); }
template <typename T> void radix_sort_unstable(std::vector<T>& v) { radix_sort_builder(v).sort(); }
template <typename T> void radix_sort_unstable(T* data, std::size_t len) { radix_sort_builder(data, len).sort(); }

// A slice of structs whose key is one built-in field — what `impl RadixKey for LargeStruct` with
// `get_level` = the field's expresses in the reference (benches/struct_sort.rs:11-27,
// examples/impl_radix_key.rs:32-56).  Rows with equal keys keep their order.
template <typename T, typename KeyT>
void radix_sort_unstable_by_field(T* data, std::size_t len, KeyT T::*field) {
    static_assert(std::is_trivially_copyable<T>::value, "rows are moved as bytes");
    static_assert(sizeof(KeyT) == 4 || sizeof(KeyT) == 8, "the device route takes 4- or 8-byte key fields");
    if (len <= 1) return;
    const std::size_t offset = static_cast<std::size_t>(reinterpret_cast<const char*>(&(data[0].*field)) - reinterpret_cast<const char*>(&data[0]));
    const int rc = rdst_hip_sort_records(data, len, sizeof(T), static_cast<std::uint32_t>(offset), sizeof(KeyT), RadixKey<KeyT>::kind, nullptr);
    if (rc != RDST_OK) throw Error(rc, rdst_hip_last_error());
}
template <typename T, typename KeyT>
void radix_sort_unstable_by_field(std::vector<T>& v, KeyT T::*field) { radix_sort_unstable_by_field(v.data(), v.size(), field); }

}  // namespace rdst
#endif  // RDST_HPP
