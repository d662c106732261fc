// tools/ubench/gbench.hip -- cost of gather loads (timing study): cycles per wave-instruction per CU for 64-lane gathers
// from an L2-resident table, by width, by number of active lanes and by locality.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE: 0 dword, 1 dwordx2, 2 dwordx4 ; ACTIVE lanes (others masked off); LOCAL: 0 random lines, 1 consecutive 16 B pieces, 2 random within 4 KiB (unit-local)
template <int W, int ACTIVE, int LOCAL>
__global__ void __launch_bounds__(1024) k_gather(const uint8_t* tab, uint32_t span_mask, uint32_t* out, uint64_t* cyc, int iters) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t ad[4], st[4];
    for (int i = 0; i < 4; i++) {
        x = x * 1664525u + 1013904223u; ad[i] = (x >> 4) & span_mask & ~15u;
        x = x * 1664525u + 1013904223u; st[i] = ((x >> 4) & span_mask & ~15u) | 64;
        if (LOCAL == 1) { ad[i] = (blockIdx.x * 65536 + (threadIdx.x >> 6) * 4096 + lane * 16 + i * 1024) & span_mask; st[i] = 16384 * 16; }
        if (LOCAL == 2) { ad[i] = ((blockIdx.x * 65536 + (threadIdx.x >> 6) * 4096) & span_mask) + ((x >> 8) & 4080); st[i] = 4096 * 16 * 64; }
    }
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) ad[i] = (ad[i] + st[i]) & span_mask;
        if (lane < ACTIVE) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (W == 0) v[i] = *reinterpret_cast<const uint32_t*>(tab + ad[i]);
                if (W == 1) { uint2 t = *reinterpret_cast<const uint2*>(tab + ad[i]); v[i] = t.x ^ t.y; }
                if (W == 2) { uint4 t = *reinterpret_cast<const uint4*>(tab + ad[i]); v[i] = t.x ^ t.y ^ t.z ^ t.w; }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) acc += v[i];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    const size_t span = 16u << 20;     // 16 MiB table: 2 MiB per XCD share... L2 resident mostly (4 MiB per XCD)
    uint8_t* tab; CK(hipMalloc(&tab, span + 4096)); CK(hipMemset(tab, 1, span + 4096));
    uint32_t* out; uint64_t* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 16 * 8));
    const int iters = 512;
    auto run = [&](const char* name, auto kern, uint32_t mask) {
        kern<<<256, 1024>>>(tab, mask, out, cyc, iters); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); kern<<<256, 1024>>>(tab, mask, out, cyc, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> hc(256 * 16); CK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto v : hc) s += v; s /= hc.size();
        const double n = 16.0 * iters * 4;     // wave-instructions per CU
        printf("%-52s wall %.3f ms  %.1f ticks per wave-instr per CU  (wall %.1f ns)\n", name, ms, s / n, ms * 1e6 / n);
    };
    const uint32_t m1 = (1u << 20) - 1, m16 = (16u << 20) - 1;
#define R(W, A, L, NAME, M) run(NAME, k_gather<W, A, L>, M);
    R(0, 64, 0, "dword   64 lanes random, 1 MiB table", m1)
    R(1, 64, 0, "dwordx2 64 lanes random, 1 MiB table", m1)
    R(2, 64, 0, "dwordx4 64 lanes random, 1 MiB table", m1)
    R(2, 64, 0, "dwordx4 64 lanes random, 16 MiB table", m16)
    R(1, 32, 0, "dwordx2 32 lanes random, 1 MiB table", m1)
    R(1, 16, 0, "dwordx2 16 lanes random, 1 MiB table", m1)
    R(1, 8, 0, "dwordx2  8 lanes random, 1 MiB table", m1)
    R(2, 16, 0, "dwordx4 16 lanes random, 1 MiB table", m1)
    R(2, 64, 1, "dwordx4 64 lanes coalesced 1 KiB", m16)
    R(1, 64, 2, "dwordx2 64 lanes random within the wave's 4 KiB", m16)
    R(2, 64, 2, "dwordx4 64 lanes random within the wave's 4 KiB", m16)
    return 0;
}
