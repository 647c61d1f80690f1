// Minimal reproducer for the round-1 abort ("0x2"): does a kernel ever see OLD contents of a
// device buffer after a stream-ordered H2D copy from pageable host memory?  Two allocation schemes, as in
// the two versions of rdst_hip_sort's host path:
//   pool   hipMallocAsync / hipFreeAsync per call on one kept non-blocking stream (commit 74cafa2)
//   kept   one hipMalloc buffer reused by every call on that stream                (commit bf16c3a, HEAD)
// Per call: fill a pageable host vector with a pattern of the call, H2D on the stream, then TWO passes over the
// buffer by kernels with different block-to-address maps (like K1 and K3: different CUs / XCDs read each
// line), each summing what it sees; the sums go back and are compared with the host's.  Between calls the
// sizes change, and (like the C++ test's other entry points) a blocking-stream hipMalloc/hipMemcpy/hipFree
// round runs.  Prints one line per mismatch and a summary.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/pool_h2d_probe.hip -o tools/probe/pool_h2d_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void sum_forward(const uint32_t* p, uint64_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += p[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ void sum_pieces(const uint32_t* p, uint64_t n, unsigned long long* out) {  // contiguous piece per block, like K1
    const uint64_t piece = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * piece, hi = lo + piece < n ? lo + piece : n;
    unsigned long long acc = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += p[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// kernel-written data read back by the NEXT launch with another block-to-address map (what a scatter pass
// and its successor do with the tmp half): fill = f(index, salt), then sum it
__global__ void fill_pieces(uint32_t* p, uint64_t n, uint32_t salt) {
    const uint64_t piece = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * piece, hi = lo + piece < n ? lo + piece : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) p[i] = (uint32_t)(i * 2246822519u) ^ salt;
}

int run(bool pool, int calls, int* mismatches) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, 16));
    unsigned long long* h_out;
    CK(hipHostMalloc(&h_out, 16));
    void* kept = nullptr;
    size_t kept_bytes = 0;
    // byte sizes of the keys of the C++ test's calls, in its order (tests/cpp/test_rdst_hpp.cpp: eight types at
    // 100 000 elements, then u32 / u64 / i32 / i64 at 3 000 001 — the call that threw was the last), as dwords
    const size_t sizes[] = {100000, 200000, 100000, 200000, 100000, 200000, 25000, 50000, 3000001, 6000002, 3000001, 6000002};
    for (int c = 0; c < calls; ++c) {
        const size_t n = sizes[c % 12];
        std::vector<uint32_t> host(n);  // pageable
        unsigned long long expect = 0;
        for (size_t i = 0; i < n; ++i) { host[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(c * 0x9E3779B9u); expect += host[i]; }
        const size_t bytes = n * 4, half = (bytes + 255) / 256 * 256;
        void* d = nullptr;
        if (pool) {
            CK(hipMallocAsync(&d, 2 * half, s));
        } else {
            if (kept_bytes < 2 * half) {
                if (kept) { CK(hipStreamSynchronize(s)); CK(hipFree(kept)); }
                CK(hipMalloc(&kept, 2 * half + half));
                kept_bytes = 2 * half + half;
            }
            d = kept;
        }
        CK(hipMemsetAsync(d_out, 0, 16, s));
        CK(hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(sum_pieces, dim3(256), dim3(1024), 0, s, (const uint32_t*)d, (uint64_t)n, d_out);
        hipLaunchKernelGGL(sum_forward, dim3(1024), dim3(256), 0, s, (const uint32_t*)d, (uint64_t)n, d_out + 1);
        // like the sort: passes write the other half, the next launch reads it back — eight rounds, like eight levels
        unsigned long long expect2 = 0;
        bool bad2 = false;
        CK(hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        const bool bad1 = h_out[0] != expect || h_out[1] != expect;
        for (int round = 0; round < 8; ++round) {
            uint32_t* half2 = (uint32_t*)((char*)d + ((round & 1) ? 0 : half));  // tmp half, keys half, tmp half ...
            const uint32_t salt = (uint32_t)(c * 131 + round) * 0x9E3779B9u;
            expect2 = 0;
            for (size_t i = 0; i < n; ++i) expect2 += (uint32_t)((uint32_t)(i * 2246822519u) ^ salt);
            CK(hipMemsetAsync(d_out, 0, 16, s));
            hipLaunchKernelGGL(fill_pieces, dim3(997), dim3(768), 0, s, half2, (uint64_t)n, salt);
            hipLaunchKernelGGL(sum_forward, dim3(1024), dim3(256), 0, s, (const uint32_t*)half2, (uint64_t)n, d_out);
            hipLaunchKernelGGL(sum_pieces, dim3(256), dim3(1024), 0, s, (const uint32_t*)half2, (uint64_t)n, d_out + 1);
            CK(hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (h_out[0] != expect2 || h_out[1] != expect2) { bad2 = true; printf("%s call %d n=%zu round %d: kernel-written data read back WRONG (forward %s, pieces %s)\n", pool ? "pool" : "kept", c, n, round, h_out[0] == expect2 ? "ok" : "bad", h_out[1] == expect2 ? "ok" : "bad"); }
        }
        if (bad1 || bad2) {
            ++*mismatches;
            if (bad1) printf("%s call %d n=%zu: H2D data read back wrong\n", pool ? "pool" : "kept", c, n);
        }
        std::vector<uint32_t> back(n);
        CK(hipMemcpyAsync(back.data(), d, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (pool) CK(hipFreeAsync(d, s));
        if (c % 12 == 11) {  // another entry point's habits in between: blocking stream, plain allocations
            hipStream_t s2;
            void* q;
            CK(hipStreamCreate(&s2));
            CK(hipMalloc(&q, 300007 * 16));
            CK(hipMemcpyAsync(q, host.data(), bytes < 300007 * 16 ? bytes : 300007 * 16, hipMemcpyHostToDevice, s2));
            CK(hipStreamSynchronize(s2));
            CK(hipFree(q));
            CK(hipStreamDestroy(s2));
        }
    }
    if (kept) CK(hipFree(kept));
    return 0;
}

int main(int argc, char** argv) {
    const int calls = argc > 1 ? atoi(argv[1]) : 200;
    int bad_pool = 0, bad_kept = 0;
    if (int rc = run(true, calls, &bad_pool)) return rc;
    if (int rc = run(false, calls, &bad_kept)) return rc;
    printf("pool: %d of %d calls saw stale data; kept: %d of %d\n", bad_pool, calls, bad_kept, calls);
    return 0;
}
