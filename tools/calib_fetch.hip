// calib_fetch.hip — calibrates rocprofv3's FETCH_SIZE on the access patterns of k_shade_guided (MI355X_MICROARCH.md,
// "HBM": the counter reads 1/2 of the bytes of a wide coalesced stream; "other access widths are uncalibrated").
// Every kernel reads a KNOWN number of bytes: mode 0 streams the table once, 16 bytes per lane, coalesced; modes 32 /
// 64 / 128 make `n_reads` scattered reads of that many contiguous bytes per lane at a hashed, width-aligned offset
// (the quadtree-node and jump-table pattern: one lane, one 32-byte record somewhere in a table no L2 holds).
// Usage: calib_fetch <mode> <table MiB> <reads, in 2^20> [repeats]   -> one JSON line with the kernel's own time.
// Run it under `rocprofv3 --pmc FETCH_SIZE` (and, separately, `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`) and
// divide by the byte count it prints: tools/gpu_calib_fetch.sh does both and writes profiles/r03_fetch_calibration.json.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t v) {  // splitmix64 finaliser
    v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull;
    v ^= v >> 27; v *= 0x94d049bb133111ebull;
    return v ^ (v >> 31);
}

__global__ __launch_bounds__(256) void k_stream(const float4* table, uint64_t n_vec, float* sink) {
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) {
        const float4 v = table[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) *sink = acc;
}

template <int VECS>  // VECS float4 = 16 * VECS contiguous bytes per read
__global__ __launch_bounds__(256) void k_scatter(const float4* table, uint64_t n_slots, uint64_t n_reads, float* sink) {
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t slot = mix(i) % n_slots;
        const float4* p = table + slot * VECS;
#pragma unroll
        for (int k = 0; k < VECS; k++) {
            const float4 v = p[k];
            acc += v.x + v.w;
        }
    }
    if (acc == 1.2345f) *sink = acc;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: calib_fetch <mode 0|32|64|128> <table MiB> <reads in 2^20> [repeats]\n");
        return 2;
    }
    const int mode = atoi(argv[1]);
    const uint64_t table_bytes = (uint64_t)atoll(argv[2]) << 20;
    const uint64_t n_reads = (uint64_t)atoll(argv[3]) << 20;
    const int repeats = argc > 4 ? atoi(argv[4]) : 3;
    if (table_bytes < (1u << 20) || table_bytes > (64ull << 30) || (mode != 0 && mode != 32 && mode != 64 && mode != 128)) return 2;
    float4* table = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(table, 0, table_bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < repeats; r++) {
        CHECK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, table, table_bytes / 16, sink);
        else if (mode == 32) hipLaunchKernelGGL(k_scatter<2>, dim3(grid), dim3(256), 0, 0, table, table_bytes / 32, n_reads, sink);
        else if (mode == 64) hipLaunchKernelGGL(k_scatter<4>, dim3(grid), dim3(256), 0, 0, table, table_bytes / 64, n_reads, sink);
        else hipLaunchKernelGGL(k_scatter<8>, dim3(grid), dim3(256), 0, 0, table, table_bytes / 128, n_reads, sink);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double bytes = mode == 0 ? (double)table_bytes : (double)n_reads * mode;
    printf("{\"mode\": %d, \"table_mib\": %llu, \"reads\": %llu, \"requested_bytes_per_launch\": %.0f, \"launches\": %d, \"best_ms\": %.4f, \"requested_GBps\": %.1f, \"Greads_per_s\": %.2f}\n",
           mode, (unsigned long long)(table_bytes >> 20), (unsigned long long)(mode == 0 ? table_bytes / 16 : n_reads), bytes, repeats, best, bytes / best * 1e-6,
           (mode == 0 ? (double)(table_bytes / 16) : (double)n_reads) / best * 1e-6);
    (void)hipFree(table);
    (void)hipFree(sink);
    return 0;
}
