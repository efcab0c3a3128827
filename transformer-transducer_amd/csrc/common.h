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
// ONE v_cvt_pk_bf16_f32 (round 5; converting each half and OR-ing them cost a second instruction per pair - same rounding, same bits)
typedef float ttmi_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 ttmi_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(ttmi_f32x2_t{lo, hi}, ttmi_bf16x2_t));
}

// sum over the 64 lanes, returned in every lane: five DPP adds inside the rows of 16, two row broadcasts, one readlane (the __shfl_xor
// butterfly is six ds_bpermute round trips through the LDS crossbar; the LayerNorm kernels chain two to four of these per row)
__device__ __forceinline__ float wave_sum(float v) {
#define TTMI_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false))
    TTMI_DPP_ADD(0xB1, 0xF);      // quad_perm [1,0,3,2]
    TTMI_DPP_ADD(0x4E, 0xF);      // quad_perm [2,3,0,1]
    TTMI_DPP_ADD(0x141, 0xF);     // row_half_mirror
    TTMI_DPP_ADD(0x140, 0xF);     // row_mirror: every lane holds its row's sum
    TTMI_DPP_ADD(0x142, 0xA);     // row_bcast:15 into rows 1 and 3
    TTMI_DPP_ADD(0x143, 0xC);     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
#undef TTMI_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout.  ONE 32-bit hash word serves an aligned group of four consecutive elements: element idx is kept iff
//     rotl(hash(seed, idx >> 2), 8 (idx & 3)) ^ K[idx & 3]  >=  p * 2^32
// (the finalizer's bytes are mixed, so the four rotations decide near-independently; v_mul_lo_u32 is quarter rate and the hash has four of
// them, which made the mask the bound of the LayerNorm kernels while every element had its own word).  Stateless: the backward pass
// regenerates the identical mask from (seed, idx).  Multipliers are 0 or 1/(1-p).
struct DropSpec {
    float p = 0.f;            // drop probability (0 = disabled)
    unsigned seed = 0;
    // optional device word mixed into the seed when the kernel starts (ttmi_set_dropout_salt): a training step captured in a HIP graph has
    // its host-drawn seeds baked into the kernel arguments; the caller bumps this word on the device before every replay, so every step
    // still draws new masks (forward and backward of one step read the same value)
    const unsigned* salt = nullptr;
};
// call ONCE at kernel entry (one scalar load), then use the result with drop_mult / drop_mult4
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
__host__ __device__ __forceinline__ unsigned drop_lane_word(unsigned h, unsigned k) {      // k = idx & 3
    const unsigned r = (h << (8u * k)) | (h >> ((32u - 8u * k) & 31u));
    return r ^ (k * 0x9E3779B9u);
}
__host__ __device__ __forceinline__ float drop_mult(const DropSpec& d, unsigned long long idx) {
    if (d.p <= 0.f) return 1.f;
    const unsigned thresh = (unsigned)(d.p * 4294967296.0);
    return drop_lane_word(ttmi_hash32(d.seed, idx >> 2), (unsigned)idx & 3u) >= thresh ? 1.f / (1.f - d.p) : 0.f;
}
// the multipliers of elements idx .. idx + 3: one hash word when idx is a multiple of four (the vector paths), four otherwise
__host__ __device__ __forceinline__ void drop_mult4(const DropSpec& d, unsigned long long idx, float (&m)[4]) {
    if (d.p <= 0.f) { m[0] = m[1] = m[2] = m[3] = 1.f; return; }
    if ((idx & 3ull) == 0) {
        const unsigned thresh = (unsigned)(d.p * 4294967296.0);
        const float keep = 1.f / (1.f - d.p);
        const unsigned h = ttmi_hash32(d.seed, idx >> 2);
        m[0] = h >= thresh ? keep : 0.f;
        m[1] = (((h << 8) | (h >> 24)) ^ 0x9E3779B9u) >= thresh ? keep : 0.f;
        m[2] = (((h << 16) | (h >> 16)) ^ (2u * 0x9E3779B9u)) >= thresh ? keep : 0.f;
        m[3] = (((h << 24) | (h >> 8)) ^ (3u * 0x9E3779B9u)) >= thresh ? keep : 0.f;
    } else {
        for (int j = 0; j < 4; ++j) m[j] = drop_mult(d, idx + j);
    }
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- counted waits on the vector-memory counter, checked against the compiled ISA (tests/test_isa_invariants.py).  vmcnt counts loads, stores,
// atomics and LDS-DMA together, in issue order; `s_waitcnt vmcnt(n)` returns when at most the n YOUNGEST are outstanding.  A counted wait is
// therefore right iff at least n vector-memory operations are issued between the last operation that must have completed and the wait, on
// EVERY path.  TTMI_VM_GUARD(id) marks the point before which everything must be complete when the matching TTMI_VM_WAIT(id, n) returns; the
// test builds the kernel's control-flow graph from the ISA and proves min over paths (operations between a GUARD and its WAIT) >= n.  (Round 3
// shipped a vmcnt(63) that guarded nothing - a step issued 32 operations, not 64 - and showed as a rare illegal access only in the full C5
// step.)  Both are asm comments / one instruction: no code beyond the wait itself.
#define TTMI_VM_GUARD(id) asm volatile("; TTMI_GUARD " id ::: "memory")
#define TTMI_VM_WAIT(id, n) asm volatile("s_waitcnt vmcnt(%0) ; TTMI_WAIT " id ::"n"(n) : "memory")
