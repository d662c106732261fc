// tools/ubench/ubench.hip -- instruction-rate microbenchmarks behind the scan kernel's cost model (timing study, not product).
//   hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
// Reports cycles per wave-instruction (per SIMD for VALU, per CU for LDS) at 16 waves per CU, from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIter = 512;

// ---- VALU: 8 independent chains, 16 instructions per iteration ---------------------------------------------------
template <int OP>
__global__ void __launch_bounds__(1024) k_valu(uint32_t* out, uint64_t* cyc, uint32_t seed, uint32_t sk) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * seed;
    uint32_t k = sk;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; it++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(k));
                if (OP == 1) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));
                if (OP == 2) asm volatile("v_alignbit_b32 %0, %0, %0, 1" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(k));
                if (OP == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(k));
                if (OP == 6) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(k));
                if (OP == 7) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(k));
                if (OP == 8) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "s"(k));
                if (OP == 9) asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(a[i]) : "s"(k));
                if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(a[i]));
                if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (OP == 12) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (OP == 13) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
            }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- LDS: 8 reads in flight per iteration, addresses from a per-lane pseudo-random or structured stream -------------
// MODE 0: ds_read_b32 random over `span` bytes; 1: ds_read_u8 random; 2: ds_read_u8 over 27 letters (class-table like);
// 3: ds_bpermute_b32 random lane; 4: ds_read_b32 linear (conflict free); 5: ds_read_u16 random; 6: ds_read_b64 random
template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(uint32_t* out, uint64_t* cyc, uint32_t span, uint32_t seed) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < span / 4; i += blockDim.x) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + seed + blockIdx.x;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; it++) {
        uint32_t ad[8], v[8];
        uint32_t v2[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            x = x * 1664525u + 1013904223u;     // (the address stream costs VALU too: subtract the VALU-only run)
            uint32_t r = x >> 8;
            if (MODE == 0) ad[i] = (r % span) & ~3u;
            if (MODE == 1) ad[i] = r % span;
            if (MODE == 2) ad[i] = 97 + (r % 26);
            if (MODE == 3) ad[i] = (r & 63) << 2;
            if (MODE == 4) ad[i] = ((threadIdx.x & 63) * 4 + (r & 0xFF00)) % span;
            if (MODE == 5) ad[i] = (r % span) & ~1u;
            if (MODE == 6) ad[i] = (r % span) & ~7u;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0 || MODE == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(ad[i]));
            if (MODE == 1 || MODE == 2) asm volatile("ds_read_u8 %0, %1" : "=v"(v[i]) : "v"(ad[i]));
            if (MODE == 3) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[i]) : "v"(ad[i]), "v"(x));
            if (MODE == 5) asm volatile("ds_read_u16 %0, %1" : "=v"(v[i]) : "v"(ad[i]));
            if (MODE == 6) { uint64_t t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(ad[i])); v[i] = (uint32_t)t; v2[i] = (uint32_t)(t >> 32); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) { acc += v[i]; if (MODE == 6) acc += v2[i]; }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// the address stream alone (same VALU, no LDS)
__global__ void __launch_bounds__(1024) k_lds_base(uint32_t* out, uint64_t* cyc, uint32_t span, uint32_t seed) {
    uint32_t x = threadIdx.x * 2654435761u + seed + blockIdx.x;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            x = x * 1664525u + 1013904223u;
            uint32_t r = x >> 8;
            uint32_t a = (r % span) & ~3u;
            asm volatile("" : "+v"(a));
            acc += a;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <class F>
static void run(const char* name, F launch, int threads, double n_inst_per_wave, bool per_cu) {
    const int blocks = 256;
    uint32_t* out; uint64_t* cyc;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CK(hipMalloc(&cyc, (size_t)blocks * 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(blocks, threads, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(blocks, threads, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = threads / 64;
    std::vector<uint64_t> h((size_t)blocks * waves);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (auto v : h) sum += (double)v;
    const double wave_cyc = sum / h.size();          // s_memtime ticks: 100 MHz?  report both
    // instructions per SIMD = waves/4 * n_inst_per_wave ; per CU = waves * n_inst_per_wave
    const double per = per_cu ? waves * n_inst_per_wave : waves / 4.0 * n_inst_per_wave;
    printf("%-34s waves/CU %2d  wall %.3f ms  memtime ticks/wave %.0f  => %.2f ticks per wave-instr per %s  (wall: %.2f ns)\n", name, waves, ms,
           wave_cyc, wave_cyc / per, per_cu ? "CU" : "SIMD", ms * 1e6 / per);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const char* vn[] = {"v_mad_u32_u24", "v_bfe_u32", "v_alignbit_b32", "v_lshrrev_b32", "v_mul_lo_u32", "v_and_b32", "v_add3_u32",
                        "v_mul_u32_u24", "v_dot4_u32_u8", "v_perm_b32", "v_lshl_add_u32", "v_cndmask_b32", "v_mov_dpp row_shr", "v_cmp_eq_u32"};
    const double nv = (double)kIter * 16;
    for (int threads : {256, 512, 1024}) {
#define V(OP) run(vn[OP], [&](int b, int t, uint32_t* o, uint64_t* c) { k_valu<OP><<<b, t>>>(o, c, 12345u, 77u); }, threads, nv, false);
        V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7) V(8) V(9) V(10) V(11) V(12) V(13)
    }
    const double nl = (double)kIter * 8;
    const uint32_t span = 65536;
    for (int threads : {256, 1024}) {
        run("lds address stream only (VALU)", [&](int b, int t, uint32_t* o, uint64_t* c) { k_lds_base<<<b, t>>>(o, c, span, 1u); }, threads, nl, true);
#define L(M, NAME) run(NAME, [&](int b, int t, uint32_t* o, uint64_t* c) { k_lds<M><<<b, t, span>>>(o, c, span, 1u); }, threads, nl, true);
        L(0, "ds_read_b32 random 64K") L(1, "ds_read_u8 random 64K") L(2, "ds_read_u8 26 letters") L(3, "ds_bpermute_b32 random")
        L(4, "ds_read_b32 lane-linear") L(5, "ds_read_u16 random 64K") L(6, "ds_read_b64 random 64K")
    }
    return 0;
}
