// Shared device/host helpers for libttmi (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>

#define TTMI_OK 0
#define TTMI_EINVAL (-1)

extern "C" const char* ttmi_last_error(void);
void ttmi_set_error(const char* fmt, ...);

#define TTMI_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ttmi_set_error(__VA_ARGS__);        \
            return TTMI_EINVAL;                 \
        }                                       \
    } while (0)

// launch check: returns hipError_t (>0) through the C ABI, never throws
#define TTMI_LAUNCH_CHECK(name)                                              \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            ttmi_set_error("%s: %s", name, hipGetErrorString(e__));          \
            return (int)e__;                                                 \
        }                                                                    \
    } while (0)

typedef uint16_t bf16_t;   // raw bf16 bits in memory

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; plain cast lets hipcc emit v_cvt_pk_bf16_f32 (NaN-safe)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);
    return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout: keep element `idx` of dropout site `seed` iff hash(seed, idx) >= p * 2^32.
// Stateless, so the backward pass regenerates the identical mask from (seed, idx).  Returns 0 or 1/(1-p).
struct DropSpec {
    float p = 0.f;            // drop probability (0 = disabled)
    unsigned seed = 0;
    // optional device word mixed into the seed when the kernel starts (ttmi_set_dropout_salt): a training step captured in a HIP graph has
    // its host-drawn seeds baked into the kernel arguments; the caller bumps this word on the device before every replay, so every step
    // still draws new masks (forward and backward of one step read the same value)
    const unsigned* salt = nullptr;
};
// call ONCE at kernel entry (one scalar load), then use the result with drop_mult
__device__ __forceinline__ DropSpec drop_live(DropSpec d) {
    if (d.p > 0.f && d.salt) d.seed ^= *d.salt * 0x9E3779B1u;
    d.salt = nullptr;
    return d;
}
__host__ __device__ __forceinline__ unsigned ttmi_hash32(unsigned seed, unsigned long long idx) {
    unsigned x = (unsigned)idx * 0x9E3779B1u ^ seed ^ ((unsigned)(idx >> 32) * 0x7F4A7C15u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ float drop_mult(const DropSpec& d, unsigned long long idx) {
    if (d.p <= 0.f) return 1.f;
    const unsigned thresh = (unsigned)(d.p * 4294967296.0);
    return ttmi_hash32(d.seed, idx) >= thresh ? 1.f / (1.f - d.p) : 0.f;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
