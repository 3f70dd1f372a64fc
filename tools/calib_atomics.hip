// calib_atomics.hip — what a returning atomic on a table of counters costs on MI355X, by memory scope.
// k_guide_locate hands every record its rank inside its spatial leaf with one returning atomicAdd on the leaf's counter
// (one counter per leaf, ~3 x 10^4 of them, the leaves of a wave's 64 records all different beyond the first bounce):
// 60 % of that kernel's time. HIP's atomicAdd is agent scope: on a chip of eight XCDs with an L2 each, that is an atomic
// executed at the memory side (TCC_EA0_ATOMIC). An atomic of WORKGROUP scope is executed in the L2 of the XCD the
// workgroup runs on — coherent among the workgroups of that XCD only, so it needs one copy of the counters per XCD
// (XCC_ID register) and a sum over the copies afterwards.
// Usage: calib_atomics <counters> <atomics, in 2^20> [repeats]  -> JSON lines: scope, rate, and a check of the sums.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t v) {
    v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull;
    v ^= v >> 27; v *= 0x94d049bb133111ebull;
    return v ^ (v >> 31);
}
// HW_REG_XCC_ID (id 20), bits 3:0: s_getreg_b32 simm16 = (size - 1) << 11 | offset << 6 | id
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u; }

template <int SCOPE>  // 0: agent scope on one table; 1: workgroup scope on the XCD's own copy
__global__ __launch_bounds__(256) void k_atomics(uint32_t* counters, uint32_t n_counters, uint64_t n_atomics, uint32_t* rank_sum, uint32_t* xcd_of_block) {
    const uint32_t xcd = xcc_id();
    if (threadIdx.x == 0 && xcd_of_block) xcd_of_block[blockIdx.x] = xcd;
    uint32_t* mine = SCOPE == 1 ? counters + (size_t)xcd * n_counters : counters;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_atomics; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(mix(i) % n_counters);
        uint32_t old;
        if (SCOPE == 1) old = __hip_atomic_fetch_add(mine + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else old = atomicAdd(mine + k, 1u);
        acc += old;
    }
    if (acc == 0x12345678u) *rank_sum = acc;  // (keeps the returned value alive)
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: calib_atomics <counters> <atomics in 2^20> [repeats]\n");
        return 2;
    }
    const uint32_t n_counters = (uint32_t)atoll(argv[1]);
    const uint64_t n_atomics = (uint64_t)atoll(argv[2]) << 20;
    const int repeats = argc > 3 ? atoi(argv[3]) : 3;
    uint32_t *d_counters = nullptr, *d_sink = nullptr, *d_xcd = nullptr;
    const int grid = 256 * 8;
    CHECK(hipMalloc((void**)&d_counters, (size_t)8 * n_counters * sizeof(uint32_t)));
    CHECK(hipMalloc((void**)&d_sink, sizeof(uint32_t)));
    CHECK(hipMalloc((void**)&d_xcd, grid * sizeof(uint32_t)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int scope = 0; scope < 2; scope++) {
        float best = 1e30f;
        for (int r = 0; r < repeats; r++) {
            CHECK(hipMemset(d_counters, 0, (size_t)8 * n_counters * sizeof(uint32_t)));
            CHECK(hipEventRecord(e0));
            if (scope == 0) hipLaunchKernelGGL(k_atomics<0>, dim3(grid), dim3(256), 0, 0, d_counters, n_counters, n_atomics, d_sink, d_xcd);
            else hipLaunchKernelGGL(k_atomics<1>, dim3(grid), dim3(256), 0, 0, d_counters, n_counters, n_atomics, d_sink, d_xcd);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // every atomic must be counted exactly once: the sum over the copies of every counter
        std::vector<uint32_t> h((size_t)8 * n_counters);
        CHECK(hipMemcpy(h.data(), d_counters, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t total = 0;
        for (uint32_t v : h) total += v;
        std::vector<uint32_t> hx(grid);
        CHECK(hipMemcpy(hx.data(), d_xcd, grid * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, round_robin = 0;
        for (int b = 0; b < grid; b++) {
            per_xcd[hx[b] & 7u]++;
            round_robin += (hx[b] == (uint32_t)(b % 8)) ? 1u : 0u;
        }
        printf("{\"scope\": \"%s\", \"counters\": %u, \"atomics\": %llu, \"ms\": %.3f, \"G_atomics_per_s\": %.2f, \"sum_of_counters\": %llu, \"exact\": %s, "
               "\"blocks_per_xcd\": [%u, %u, %u, %u, %u, %u, %u, %u], \"blocks_on_xcd_b_mod_8\": %u}\n",
               scope == 0 ? "agent (one table)" : "workgroup (one copy per XCD)", n_counters, (unsigned long long)n_atomics, best, n_atomics / best * 1e-6,
               (unsigned long long)total, total == n_atomics ? "true" : "false", per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7],
               round_robin);
    }
    return 0;
}
