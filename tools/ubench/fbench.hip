// tools/ubench/fbench.hip -- filter-loop prototypes for the scan kernel (timing study, not product).
// One wave per 4 KiB unit, lane k owns 64 consecutive bytes, 16 waves per workgroup share the LDS tables.
//   V0: stride 1, class lookup per byte, direct K^4 bitmap (the round-1 loop)
//   V1: stride 2, K^4 bitmap               V2: stride 2, [3-class word][front-class bit] direct
//   V3: stride 2, hashed word index        V4: stride 1, [x3 word][front bit] direct
//   V5: stride 2, K^4 bitmap, coalesced 1 KiB rounds (lane k owns bytes [1024 r + 16 k, +16)), history via DPP wave_shr
// plus LDS instruction-rate probes with cheap address streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct __attribute__((packed, aligned(1))) U128u { uint32_t x, y, z, w; };
typedef __attribute__((address_space(3))) const uint8_t lds_u8;
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
__device__ __forceinline__ uint32_t mad24s(uint32_t a, uint32_t sb, uint32_t c) {
    uint32_t d; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sb), "v"(c)); return d;
}
__device__ __forceinline__ uint32_t mul24s(uint32_t a, uint32_t sb) {
    uint32_t d; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sb)); return d;
}

struct FP { const uint8_t* text; uint64_t n_units; const uint8_t* cls; const uint32_t* filt; uint32_t filt_words; uint32_t kp; uint32_t* out; uint32_t hshift; };

template <int V>
__global__ void __launch_bounds__(1024) k_filter(const FP P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* cls = smem;
    uint32_t* filt = reinterpret_cast<uint32_t*>(smem + 256);
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = P.cls[i];
    for (uint32_t i = threadIdx.x; i < P.filt_words; i += blockDim.x) filt[i] = P.filt[i];
    __syncthreads();
    lds_u8* lcls = (lds_u8*)0;
    lds_u32* lfilt = (lds_u32*)256;
    lds_u8* lfilt8 = (lds_u8*)256;
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t kp = __builtin_amdgcn_readfirstlane(P.kp), kp2 = kp * kp, kp4 = kp * 4, kp24 = kp2 * 4;
    const uint64_t stride = (uint64_t)gridDim.x * 16;
    for (uint64_t u = (uint64_t)blockIdx.x * 16 + wave; u < P.n_units; u += stride) {
        const uint8_t* src = P.text + u * 4096 + lane * 64;
        uint32_t m0 = 0, m1 = 0;
        uint32_t acc = 0;
        U128u nxt = *reinterpret_cast<const U128u*>(src);
        uint32_t cp = 0, pm1 = 0, pm2 = 0, c3 = 0;     // history (zero at the lane start: prototype)
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
            if (q + 1 < 4) nxt = *reinterpret_cast<const U128u*>(src + (q + 1) * 16);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (V == 0) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t cl = lcls[(w[d] >> (8 * b)) & 0xFF];
                        const uint32_t pair = mad24s(cp, kp, cl);
                        const uint32_t x = mad24s(pm2, kp2, pair);
                        pm2 = pm1; pm1 = pair; cp = cl;
                        const uint32_t fw = lfilt[x >> 5];
                        acc = __builtin_amdgcn_alignbit(fw >> (x & 31), acc, 1);
                    }
                } else if (V == 4) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const uint32_t cl = lcls[(w[d] >> (8 * b)) & 0xFF];
                        const uint32_t pair = mad24s(cp, kp, cl);            // (c[i-1], c[i])
                        const uint32_t x3 = mad24s(pm2, kp2, pair);          // pm2 = c[i-2] here
                        const uint32_t fw = lfilt[x3];
                        acc = __builtin_amdgcn_alignbit(fw >> (c3 & 31), acc, 1);
                        c3 = pm2; pm2 = cp; cp = cl;
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t co = lcls[(w[d] >> (16 * h)) & 0xFF];        // odd-position byte (first of the pair)
                        const uint32_t ce = lcls[(w[d] >> (16 * h + 8)) & 0xFF];    // even-position byte
                        if (V == 1) {
                            const uint32_t pair = mad24s(co, kp, ce);
                            const uint32_t x = mad24s(pm1, kp2, pair);
                            pm1 = pair;
                            const uint32_t fw = lfilt[x >> 5];
                            acc = __builtin_amdgcn_alignbit(fw >> (x & 31), acc, 1);
                        } else if (V == 2) {
                            // window (c3, cp, co, ce): word = (cp, co, ce), bit = c3
                            const uint32_t pair = mad24s(co, kp, ce);
                            const uint32_t x3 = mad24s(cp, kp2, pair);
                            const uint32_t fw = lfilt[x3];
                            acc = __builtin_amdgcn_alignbit(fw >> (c3 & 31), acc, 1);
                            c3 = co; cp = ce;
                        } else if (V == 3) {
                            const uint32_t pair = mad24s(co, kp, ce);
                            const uint32_t x3 = mad24s(cp, kp2, pair);
                            const uint32_t hsh = mul24s(x3, 0x9E3779u);
                            const uint32_t fw = *(lds_u32*)(256 + ((hsh >> P.hshift) & ~3u));
                            acc = __builtin_amdgcn_alignbit(fw >> (c3 & 31), acc, 1);
                            c3 = co; cp = ce;
                        }
                    }
                }
            }
            if (V == 0 || V == 4) { if (q & 1) { if (q == 1) m0 = acc; else m1 = acc; } }
            else if (q == 3) m0 = acc;
        }
        uint32_t f = __popc(m0) + __popc(m1);
        for (int s = 32; s; s >>= 1) f += __shfl_xor(f, s, 64);
        if (lane == 0) P.out[u] = f;
    }
}

__global__ void __launch_bounds__(1024) k_filter5(const FP P) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint8_t* cls = smem;
    uint32_t* filt = reinterpret_cast<uint32_t*>(smem + 256);
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) cls[i] = P.cls[i];
    for (uint32_t i = threadIdx.x; i < P.filt_words; i += blockDim.x) filt[i] = P.filt[i];
    __syncthreads();
    lds_u8* lcls = (lds_u8*)0;
    lds_u32* lfilt = (lds_u32*)256;
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t kp = __builtin_amdgcn_readfirstlane(P.kp), kp2 = kp * kp;
    const uint64_t stride = (uint64_t)gridDim.x * 16;
    for (uint64_t u = (uint64_t)blockIdx.x * 16 + wave; u < P.n_units; u += stride) {
        const uint8_t* src = P.text + u * 4096 + lane * 16;
        uint32_t acc = 0;
        uint32_t carry = 0;
        U128u nxt = *reinterpret_cast<const U128u*>(src);
        for (uint32_t r = 0; r < 4; r++) {
            const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
            if (r + 1 < 4) nxt = *reinterpret_cast<const U128u*>(src + (r + 1) * 1024);
            uint32_t q[8];
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t co = lcls[(w[d] >> (16 * h)) & 0xFF];
                    const uint32_t ce = lcls[(w[d] >> (16 * h + 8)) & 0xFF];
                    q[2 * d + h] = mad24s(co, kp, ce);
                }
            // pair in front of the lane's piece: the previous lane's last pair (lane 0: the previous round's lane 63)
            uint32_t qm = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)q[7], 0x138, 0xF, 0xF, false);   // wave_shr:1
            carry = __builtin_amdgcn_readlane(q[7], 63);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t x = mad24s(qm, kp2, q[k]);
                qm = q[k];
                const uint32_t fw = lfilt[x >> 5];
                acc = __builtin_amdgcn_alignbit(fw >> (x & 31), acc, 1);
            }
        }
        uint32_t f = __popc(acc);
        for (int s = 32; s; s >>= 1) f += __shfl_xor(f, s, 64);
        if (lane == 0) P.out[u] = f;
    }
}

// ---- LDS rates with cheap address streams: ad = (ad + step) & mask (2 VALU per LDS instruction) ----------------------
template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(uint32_t* out, uint64_t* cyc, uint32_t span, int iters, int with_lds) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < span / 4; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t ad[8], st[8];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; i++) {
        x = x * 1664525u + 1013904223u; ad[i] = (x >> 8) % span;
        x = x * 1664525u + 1013904223u; st[i] = ((x >> 8) % span) | 4;
        if (MODE == 2) { ad[i] = 97 + (x >> 9) % 26; st[i] = 0; }            // letters: class-table shaped
        if (MODE == 3) { ad[i] = (threadIdx.x & 63) * 4; st[i] = 256; }       // conflict free
    }
    const uint32_t mask = MODE == 1 || MODE == 2 ? span - 1 : (span - 1) & ~3u;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) ad[i] = (ad[i] + st[i]) & mask;
        if (with_lds) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0 || MODE == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(ad[i]));
                if (MODE == 1 || MODE == 2) asm volatile("ds_read_u8 %0, %1" : "=v"(v[i]) : "v"(ad[i]));
                if (MODE == 4) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[i]) : "v"(ad[i] & 0xFC), "v"(x));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) acc += v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) acc += ad[i];
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static uint64_t rng_s = 88172645463325252ull;
static uint32_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (uint32_t)(rng_s >> 16); }

int main(int argc, char** argv) {
    const uint64_t n_units = argc > 1 ? atoll(argv[1]) : 1000000;
    const uint32_t kp = 27;
    // text: letters with an English-like skew + spaces
    static const double freq[27] = {18, 8.2, 1.5, 2.8, 4.3, 12.7, 2.2, 2.0, 6.1, 7.0, 0.15, 0.77, 4.0, 2.4, 6.7, 7.5, 1.9, 0.095, 6.0, 6.3, 9.1, 2.8, 0.98, 2.4, 0.15, 2.0, 0.074};
    std::vector<uint8_t> cum;
    for (int c = 0; c < 27; c++) for (int k = 0; k < (int)(freq[c] * 20 + 1); k++) cum.push_back(c == 0 ? ' ' : 'a' + c - 1);
    const size_t chunk = 64 << 20;
    std::vector<uint8_t> h(chunk);
    for (auto& b : h) b = cum[rnd() % cum.size()];
    uint8_t* d_text;
    const size_t nbytes = n_units * 4096;
    CK(hipMalloc(&d_text, nbytes + 64));
    for (size_t o = 0; o < nbytes; o += chunk) CK(hipMemcpy(d_text + o, h.data(), std::min(chunk, nbytes - o), hipMemcpyHostToDevice));
    std::vector<uint8_t> cls(256, 0);
    for (int c = 1; c < 27; c++) cls['a' + c - 1] = c;
    uint8_t* d_cls; CK(hipMalloc(&d_cls, 256)); CK(hipMemcpy(d_cls, cls.data(), 256, hipMemcpyHostToDevice));
    uint32_t* d_out; CK(hipMalloc(&d_out, n_units * 4));
    auto bench = [&](const char* name, int v, uint32_t words, double density, uint32_t hshift) {
        std::vector<uint32_t> f(words, 0);
        for (uint64_t k = 0; k < (uint64_t)(density * words * 32); k++) { uint32_t r = rnd(); f[(r >> 5) % words] |= 1u << (r & 31); }
        uint32_t* d_f; CK(hipMalloc(&d_f, words * 4)); CK(hipMemcpy(d_f, f.data(), words * 4, hipMemcpyHostToDevice));
        FP P{d_text, n_units, d_cls, d_f, words, kp, d_out, hshift};
        const size_t lds = 256 + (size_t)words * 4;
        auto launch = [&]() {
            switch (v) {
#define CASE(V) case V: CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); k_filter<V><<<256, 1024, lds>>>(P); break;
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4)
                case 5: CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); k_filter5<<<256, 1024, lds>>>(P); break;
            }
        };
        launch(); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        std::vector<uint32_t> o(1000); CK(hipMemcpy(o.data(), d_out, 4000, hipMemcpyDeviceToHost));
        double fl = 0; for (auto x : o) fl += x;
        printf("%-44s %.3f ms  %.0f GB/s  flagged/unit %.1f\n", name, ms, nbytes / ms / 1e6, fl / 1000);
        CK(hipFree(d_f));
    };
    bench("V0 stride1 K^4 bitmap (round 1)", 0, (27 * 27 * 27 * 27 + 31) / 32, 0.08, 0);
    bench("V4 stride1 [x3][front bit]", 4, 27 * 27 * 27, 0.08, 0);
    bench("V1 stride2 K^4 bitmap", 1, (27 * 27 * 27 * 27 + 31) / 32, 0.16, 0);
    bench("V5 stride2 K^4 bitmap coalesced rounds", 5, (27 * 27 * 27 * 27 + 31) / 32, 0.16, 0);
    bench("V2 stride2 [x3][front bit] direct", 2, 27 * 27 * 27, 0.16, 0);
    bench("V3 stride2 hashed 16K words", 3, 16384, 0.16, 16);
    bench("V3 stride2 hashed 8K words", 3, 8192, 0.16, 17);

    // LDS rates
    uint32_t* out; uint64_t* cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 16 * 8));
    const int iters = 4096;
    auto lrun = [&](const char* name, int mode, int with) {
        const uint32_t span = 65536;
        auto launch = [&]() {
            switch (mode) {
#define LC(M) case M: k_lds<M><<<256, 1024, span>>>(out, cyc, span, iters, with); break;
                LC(0) LC(1) LC(2) LC(3) LC(4)
            }
        };
        launch(); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> hc(256 * 16); CK(hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto v : hc) s += v; s /= hc.size();
        const double n = 16.0 * iters * 8;     // wave-instructions per CU
        printf("%-36s lds=%d  wall %.3f ms  %.2f cycles (memtime) per wave-instr per CU, wall %.2f ns\n", name, with, ms, s / n, ms * 1e6 / n);
    };
    for (int with : {0, 1}) {
        lrun("ds_read_b32 random 64K", 0, with);
        lrun("ds_read_u8 random 64K", 1, with);
        lrun("ds_read_u8 letters (class table)", 2, with);
        lrun("ds_read_b32 conflict-free", 3, with);
        lrun("ds_bpermute_b32", 4, with);
    }
    return 0;
}
