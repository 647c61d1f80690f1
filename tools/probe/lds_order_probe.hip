// Development aid: in which order does the LDS service the lanes of ONE wave instruction that hit the
// same address with a returning atomic add?  (Not a documented property.)  Every wave keeps a private
// 256-bin table like the sort's; lanes add 1 to the bin of a pseudo-random 8-bit digit and compare the
// returned value with the stable rank (number of lower lanes with the same digit) from ballots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t peers_below(uint32_t d) {
    uint64_t same = ~0ull;
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        same &= bit ? bal : ~bal;
    }
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
}

__global__ __launch_bounds__(768) void probe(uint32_t rounds, uint32_t bins_mask, unsigned long long* mismatches, unsigned long long* groups) {
    __shared__ uint32_t tab[12 * 256];
    __shared__ uint32_t noise[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* wh = tab + wave * 256;
    for (int j = 0; j < 4; ++j) wh[lane + 64 * j] = 0;
    for (int j = tid; j < 4096; j += 768) noise[j] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * 768u + tid) * 2654435761u + 12345u;
    unsigned long long bad = 0, grp = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        x = x * 1664525u + 1013904223u;
        const uint32_t d = (x >> 13) & bins_mask;
        const uint32_t b0 = wh[d];
        const uint32_t below = peers_below(d);
        __builtin_amdgcn_wave_barrier();
        const uint32_t got = atomicAdd(&wh[d], 1u);  // ds_add_rtn_u32
        if (got != b0 + below) ++bad;
        if (below) ++grp;
        atomicAdd(&noise[(x >> 5) & 4095], 1u);      // other traffic on the same LDS
    }
    if (bad) atomicAdd(mismatches, bad);
    if (grp) atomicAdd(groups, grp);
}

int main() {
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    for (uint32_t mask : {255u, 63u, 7u, 1u, 0u}) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(2048), dim3(768), 0, 0, 2000u, mask, d, d + 1);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("bins %3u: %llu lane-rounds with a lower peer, %llu returned values differ from the stable rank\n", mask + 1, h[1], h[0]);
    }
    return 0;
}
