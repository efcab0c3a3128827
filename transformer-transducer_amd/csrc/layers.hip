// Sub-layer drivers behind the C ABI: relative-position self-attention, position-wise FFN and the
// joint network, forward and backward.  Each driver is a fixed sequence of asynchronous launches
// (MFMA GEMMs from gemm.hip + HBM-bound row kernels from rowops.hip) on the caller's stream; it
// allocates nothing: `ctx` (saved for backward) and `ws` (scratch) are caller-provided.
//
// Relative shift without a kernel: the reference builds BD = _rel_shift(q.E^T + c)
// (tt/transformer.py:82-89,143-145) by zero-padding a column, re-viewing [L, L+1] as [L+1, L] and
// dropping the first row.  Here the q.E^T GEMM writes straight into a slab with row pitch L+1 and
// column offset 1 (column 0 zeroed once); the SAME memory read at offset L with pitch L is the shifted
// matrix, so the content GEMM accumulates onto it in place and the softmax runs on that view.  The
// backward uses the mirror image: dS written through the pitch-L view IS dG in the pitch-(L+1) layout.
#include "gemm.h"
#include "ttmi.h"
#include "rowops.h"
#include "gemm_fast.h"
#include "attn_flash.h"
#include <mutex>
#include <unordered_map>

void ttmi_probe_begin(int slot, hipStream_t st);
void ttmi_probe_end(int slot, hipStream_t st);
void ttmi_rnnt_set_lattice_version(int v);      // rnnt.hip

namespace {

constexpr int NT_ = GEMM_A_KMAJOR | GEMM_B_KMAJOR;   // C = A . B^T   (Linear forward)
constexpr int NN_ = GEMM_A_KMAJOR;                   // C = A . B     (dgrad)
constexpr int TN_ = 0;                               // C = A^T . B   (wgrad)

inline size_t al4(size_t n) { return (n + 3) & ~size_t(3); }

GemmDesc mk(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, int flags, int prec) {
    GemmDesc d;
    d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.flags = flags | (prec == 1 ? GEMM_BF16_MFMA : (prec == 2 ? GEMM_BF16X3 : 0));      // (prec 2 = bf16x3: f32 data flow; the big dense products go through x3_* below, the rest through the generic kernel's three-term form)
    return d;
}

// weight-gradient GEMMs reduce over a long K with few output tiles: split K over workgroups, combine by f32 atomics
int pick_splitk(int M, int N, int K, int nbatch) {
    const long tiles = (long)cdiv(M, 128) * cdiv(N, 128) * nbatch;
    if (tiles >= 384) return 1;
    long s = (768 + tiles - 1) / tiles;
    const long kmax = K / 128 > 0 ? K / 128 : 1;
    if (s > kmax) s = kmax;
    return (int)(s < 1 ? 1 : s);
}

int wgrad(const float* dY, const float* X, float* gW, int M, int N, int K, long lda, long ldb, long ldc, int prec,
          hipStream_t st) {
    GemmDesc d = mk(dY, X, gW, M, N, K, lda, ldb, ldc, TN_ | GEMM_ATOMIC, prec);
    d.splitk = pick_splitk(M, N, K, 1);
    return ttmi_launch_gemm(d, st);
}

#define CK(expr)                 \
    do {                         \
        int rc__ = (expr);       \
        if (rc__) return rc__;   \
    } while (0)

// ---- TTMI_PRECISION=bf16x3 (prec 2, round 5): the parity mode's f32 data flow with its large dense products on the bf16 MFMA in three
// terms (rowops.hip split3_kernel): A . B^T ~ A_hi B_hi + A_lo B_hi + A_hi B_lo - one launch of the throughput kernels over the tripled
// reduction (NT / NN), three accumulating launches on the [hi | lo] planes (TN: weight gradients).  Relative error ~2^-16 per product against
// 2^-24 (exact f32) and 2^-8 (bf16); tests/test_configs_gpu.py (full C2 / C4 models, a layer with dropout) and tests/test_gemm_gpu.py hold the mode to the fp32 mode's own 1e-4 bounds.  `scratch`: bf16 elements.
inline int x3_pad(int n) { return (n + 63) / 64 * 64; }
inline size_t x3_al(size_t n) { return (n + 63) & ~(size_t)63; }
inline size_t x3_nt_elems(long M, long N, int K) { return x3_al((size_t)M * 3 * x3_pad(K)) + x3_al((size_t)N * 3 * x3_pad(K)); }
// (+ the 2 Mp x 2 Np f32 block of the one-launch form of x3_tn, in bf16 elements)
inline size_t x3_tn_elems(long K, int M, int N) { return x3_al((size_t)K * 2 * x3_pad(M)) + x3_al((size_t)K * 2 * x3_pad(N)) + x3_al((size_t)8 * x3_pad(M) * x3_pad(N)); }
// big enough to be worth two split passes and the throughput kernels
inline bool x3_worth(long M, long N, long K) { return M >= 256 && N >= 64 && K >= 64 && M * N * K >= (1L << 27); }
// Two layouts of the same three terms.  Products with thousands of rows (the audio encoder, the joint) run on the persistent kernels, whose K loop can
// walk A's first K-tiles a second time against another B (NtEpilogue::B_lo / K_lo): A = [hi | lo], B = [hi | hi] with B_lo = lo - no third copy of A's
// hi block is written or read.  The smaller ones keep A = [hi | lo | hi] against B = [hi | hi | lo] in one plain launch (the 64 x 64-tile kernel).
inline bool x3_two_block(long M) { return M >= 4096; }
static int x3_launch(const bf16_t* A3, const bf16_t* B3, float* C, int M, int N, int Kp, long ldc, const NtEpilogue& e, bool two_block, hipStream_t st) {
    if (!two_block) return gemm_nt_bf16(A3, B3, C, 0, e, M, N, 3 * Kp, 3L * Kp, 3L * Kp, ldc, st);
    NtEpilogue e2 = e;
    e2.B_lo = B3 + 2 * Kp;
    e2.K_lo = Kp;
    return gemm_nt_bf16(A3, B3, C, 0, e2, M, N, 2 * Kp, 2L * Kp, 3L * Kp, ldc, st);
}
// C[M,N] = epi(A[M,K] . B[N,K]^T)
int x3_nt(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, const NtEpilogue& e, bf16_t* scratch,
          hipStream_t st) {
    const int Kp = x3_pad(K);
    const bool two = x3_two_block(M);
    bf16_t* A3 = scratch;
    bf16_t* B3 = A3 + x3_al((size_t)M * 3 * Kp);
    CK(split3_bf16(A, lda, M, K, Kp, two ? 2 : 0, A3, st));
    CK(split3_bf16(B, ldb, N, K, Kp, 1, B3, st));
    return x3_launch(A3, B3, C, M, N, Kp, ldc, e, two, st);
}
// C[M,N] = epi(A[M,K] . B[K,N]) (B row-major [K, N]: a dgrad against the weight as stored)
// a_ready: the [hi | lo] rows of A already sit at the head of `scratch` - left there by the x3_tn of the same A just before (a sub-layer's weight gradient
// and its input gradient read the same dY: one split pass instead of two)
int x3_nn(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, long ldc, const NtEpilogue& e, bf16_t* scratch,
          hipStream_t st, bool a_ready = false) {
    const int Kp = x3_pad(K);
    const bool two = x3_two_block(M);
    bf16_t* A3 = scratch;
    bf16_t* B3 = A3 + x3_al((size_t)M * 3 * Kp);
    if (!(a_ready && two)) CK(split3_bf16(A, lda, M, K, Kp, two ? 2 : 0, A3, st));
    CK(split3_transpose_bf16(B, ldb, K, N, Kp, true, B3, st));
    return x3_launch(A3, B3, C, M, N, Kp, ldc, e, two, st);
}
// C[M,N] += A[K,M]^T . B[K,N] (atomically: a weight gradient)
int g_x3_tn_one_launch = 1;     // ttmi_set_option(22, 0): the three accumulating launches of round 5 (A/B)
int x3_tn(const float* A, const float* B, float* C, int M, int N, long K, long lda, long ldb, long ldc, bf16_t* scratch, hipStream_t st) {
    const int Mp = x3_pad(M), Np = x3_pad(N);
    bf16_t* A2 = scratch;
    bf16_t* B2 = A2 + x3_al((size_t)K * 2 * Mp);
    CK(split3_bf16(A, lda, K, M, Mp, 2, A2, st));
    CK(split3_bf16(B, ldb, K, N, Np, 2, B2, st));
    // round 6: ONE launch over the plane pairs as they lie - [hi | lo]^T [hi | lo] = the four blocks hi^T hi, hi^T lo, lo^T hi, lo^T lo of a 2 Mp x 2 Np product -
    // and a fold of the first three into C.  A third more flops than the three launches (the lo^T lo block is computed and dropped), but four times the tiles in one
    // grid: an encoder weight gradient is 8 ... 48 tiles of 256 x 128, a fraction of the chip three times over (3 x 40 us at C2; 40 + 2 x 5 us this way)
    if (g_x3_tn_one_launch && K % 64 == 0 && K >= 4096 && K < 32768 && (2 * Mp) % 256 == 0 && (2 * Np) % 128 == 0 && (size_t)Mp * Np <= ((size_t)1 << 23)) {
        float* T = reinterpret_cast<float*>(B2 + x3_al((size_t)K * 2 * Np));
        CK(fill_zero(T, sizeof(float) * (size_t)4 * Mp * Np, st));
        CK(gemm_tn_bf16(A2, B2, T, 2 * Mp, 2 * Np, (int)K, 2L * Mp, 2L * Np, 2L * Np, 1, st));
        return x3_fold_blocks(T, Mp, Np, C, M, N, ldc, st);
    }
    CK(gemm_tn_bf16(A2, B2, C, M, N, (int)K, 2L * Mp, 2L * Np, ldc, 1, st));
    CK(gemm_tn_bf16(A2 + Mp, B2, C, M, N, (int)K, 2L * Mp, 2L * Np, ldc, 1, st));
    return gemm_tn_bf16(A2, B2 + Np, C, M, N, (int)K, 2L * Mp, 2L * Np, ldc, 1, st);
}

// ---- intra-call concurrency: weight-gradient GEMMs feed nothing downstream inside a backward call, so they are forked onto a
// library-owned side stream (one per caller stream) and joined before the call returns (scratch buffers they read are reused by
// the next call).  ttmi_set_option(3, 0) disables it.
// ttmi_set_dropout_salt: device word mixed into every dropout seed at kernel start (graph replays).  One slot per device: a word lives in ONE
// device's memory, and a process that drives several devices must not hand it to kernels of another one (the slot is looked up by the
// calling thread's current device, at set time and at every launch site).
constexpr int MAX_SALT_DEV = 16;
const unsigned* g_drop_salts[MAX_SALT_DEV] = {};
static inline const unsigned* drop_salt_here() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_SALT_DEV) return nullptr;
    return g_drop_salts[dev];
}
#define g_drop_salt drop_salt_here()
int g_fork_wgrad = 1;
int g_capture_forks = 0;        // ttmi_set_option(18, 1): fork inside a stream capture too, from streams not marked by ttmi_stream_set_nofork (round 5, see fork_stream)
int g_split_weights = 2;        // ttmi_set_option(13, v), default 2 since round 6 (with the label value pass: the timed mode's batch-mean loss within 1e-4 at every state measured): the encoders' forward GEMMs take the second term of their weight's bf16 split (W ~ hi + lo) as a second
                                // K range over the same A tiles, one launch each (NtEpilogue::B_lo): 1 = qkv_net, o_net, CoreNet.0, CoreNet.3; 2 = the two with f32
                                // outputs (o_net, CoreNet.3) only, in stacks of at least 4096 rows (the audio encoder); + 4: all four in stacks of fewer than 4096 rows (the label encoder).  The term lives in the sub-layer's workspace
int g_posgrad_gemms = 0;        // ttmi_set_option(11, 1): dq / dE by the round-2 GEMM launches instead of attn_dqde_kernel (A/B)
int g_scatter_launch = 0;       // ttmi_set_option(16, 1): attn_dqde_kernel leaves dE / dc and relpos_scatter folds them (round 3; A/B); 2: atomics straight into
                                // the table gradients (no per-(b, h) rows + reduction)
int g_attn_slices = 1;          // ttmi_set_option(10, n): attention backward in n batch slices (see attn_bwd_impl)
int g_gemm_slab = 0;            // ttmi_set_option(5, 1): position-term slab by the batched GEMM (A/B measurements)
// which of an encoder layer's forward GEMMs take the weight's second term: option 13 = 1 all four, 2 the two with f32 outputs; bit 2 (4, round 6): all four, but only in
// stacks of fewer than 4096 rows - the LABEL encoder, whose every rounding is replicated over the T frames of the joint (5 = both encoders, 4 = the label encoder alone)
static inline bool split_w(long rows, bool f32_out) {
    const int m = g_split_weights & 3;
    // (2 means the AUDIO-sized stacks: the label encoder's VALUE comes from its own bf16x3 pass, tt/model.py::_label_states, so second terms in its bf16 pass bought nothing -
    // twelve launches of the 128 x 128 two-source kernel on the side stream, 0.43 ms of kernel time per C2 step; without that pass ask for 6 = 2 + 4)
    return (m == 1 || (m == 2 && f32_out && rows >= 4096)) || ((g_split_weights & 4) && rows < 4096);
}
int g_joint_dpd_two_pass = 1;   // ttmi_set_option(21, 0): the joint's sums over frames by f32 atomics (rounds 1 - 5; A/B)
int g_joint_dec_lo = 1;         // ttmi_set_option(19, 0): the joint's input layer without the second bf16 term of the label states (round 6; A/B)
struct SideCtx {
    int device = -1;
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int next = 0;
    bool used = false;
};
// One fork context per (device, caller stream): the default stream is nullptr on EVERY device, and autograd runs one backward thread
// per device, so the table is keyed by both and guarded by a mutex (creation is rare; a lookup is a short scan).  A context itself is
// only ever used by the thread that owns its caller stream's work.
constexpr int MAX_SIDE = 64;
SideCtx g_side[MAX_SIDE];
hipStream_t g_nofork[MAX_SIDE];
int g_nnofork = 0;
int g_nside = 0;
std::mutex g_side_mu;

SideCtx* side_for(hipStream_t main) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (int i = 0; i < g_nside; ++i)
        if (g_side[i].main == main && g_side[i].device == dev) return &g_side[i];
    if (g_nside == MAX_SIDE) return nullptr;
    SideCtx& c = g_side[g_nside];
    if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (auto& e : c.ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    c.main = main;
    c.device = dev;
    gemm_fast_alias_stream(c.side, main);       // the fork stream answers for its caller stream's CU reservation
    ++g_nside;
    return &c;
}

// returns the stream to launch the forked work on (the side stream after making it wait for everything queued on `main`
// so far), or `main` itself when forking is disabled / unavailable
hipStream_t fork_stream(hipStream_t main) {
    if (!g_fork_wgrad) return main;
    // forks inside a stream capture: a fork off the label encoder's side stream (itself forked from the capturing stream) crashes
    // hipStreamEndCapture on ROCm 7.2 (tools/debug/graph_bisect.py).  Round 5: callers mark such second-level streams
    // (ttmi_stream_set_nofork); with option 18 the capturing stream itself forks as in eager mode (its table-gradient reductions and
    // ungrouped weight gradients become parallel branches of the graph), marked streams never do inside a capture
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        if (!g_capture_forks) return main;
        std::lock_guard<std::mutex> lock(g_side_mu);
        for (int i = 0; i < g_nnofork; ++i)
            if (g_nofork[i] == main) return main;
    }
    SideCtx* c = side_for(main);
    if (!c) return main;
    hipEvent_t e = c->ev[c->next];
    c->next ^= 1;
    if (hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(c->side, e, 0) != hipSuccess) return main;
    c->used = true;
    return c->side;
}

// make `main` wait for everything forked during this call
void join_stream(hipStream_t main) {
    SideCtx* c = g_fork_wgrad ? side_for(main) : nullptr;
    if (!c || !c->used) return;
    if (hipEventRecord(c->ev[2], c->side) == hipSuccess) (void)hipStreamWaitEvent(main, c->ev[2], 0);
    c->used = false;
}

// ---- bf16 shadows of GEMM weights (ttmi_weight_shadow_*): a training loop keeps a plain and a transposed bf16 copy of every weight,
// rebuilt once per optimiser step (one launch), instead of converting every weight in every forward call (118 launches per step at C2).
// Registered by the caller, looked up by the f32 weight pointer; a shadow is used only if its geometry is the one the call needs.
struct Shadow { int R, C; const bf16_t* w16; const bf16_t* wT16; long ldT; const bf16_t* w16lo = nullptr; };     // w16lo: bf16(w - float(w16)), kept current with the others (nullable)
std::unordered_map<const void*, Shadow> g_shadows;
std::mutex g_shadow_mu;

bool shadow_of(const float* w, int R, int C, long ldT, Shadow& out) {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    if (g_shadows.empty()) return false;
    auto it = g_shadows.find(w);
    if (it == g_shadows.end()) return false;
    const Shadow& s = it->second;
    if (s.R != R || s.C != C || !s.w16 || !s.wT16 || s.ldT != ldT) return false;
    out = s;
    return true;
}

// What a forward call used for a weight - the registered shadow, or the copies it made in its own context - is recorded per (context
// buffer, weight slot), so that the backward call uses the SAME source, whatever happened to the registration in between: round 3 looked
// the shadow up again and, after FlatModel.disable_shadows() between forward and backward, read the context's transposed copy that the
// forward had never written (ADVICE r2 / VERDICT r3 weak item 10).  Host-side table keyed by the context pointer (one entry per live
// context, rewritten by every forward call on it); unknown contexts and vanished shadows REBUILD the transposed copy from the f32 weight.
std::unordered_map<const void*, unsigned> g_ctx_shadowed;        // bit s set: slot s of this context took the shadow in its forward call
std::mutex g_ctx_mu;
void ctx_record(const void* ctx, int slot, bool shadowed) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (g_ctx_shadowed.size() > 4096) g_ctx_shadowed.clear();    // (contexts of abandoned forward passes)
    unsigned& f = g_ctx_shadowed[ctx];
    f = shadowed ? (f | (1u << slot)) : (f & ~(1u << slot));
}
// -> the transposed bf16 weight [C, ldT] the backward call must use; *rc receives a launch error of the rebuild, if any
const bf16_t* ctx_weightT(const void* ctx, int slot, const float* w, int R, int C, long ldT, bf16_t* ctx_copy, hipStream_t st, int* rc) {
    bool known = false, shadowed = false;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        auto it = g_ctx_shadowed.find(ctx);
        if (it != g_ctx_shadowed.end()) { known = true; shadowed = (it->second >> slot) & 1u; }
    }
    if (known && !shadowed) return ctx_copy;                      // the forward made this copy itself
    Shadow sh;
    if (known && shadow_of(w, R, C, ldT, sh)) return sh.wT16;     // same registration as in the forward (kept current by the optimiser step)
    // the shadow is gone, or the context is UNKNOWN (its record was dropped - a second backward over a retained graph, or the table's
    // overflow clear): never trust whatever shadow is registered NOW for a forward that may have made its own copy - the
    // f32 weight gives the same values, made now
    *rc = transpose_convert_bf16(w, R, C, ctx_copy, ldT, st);
    return ctx_copy;
}
// a backward call has read every slot of its context: drop the record (the table holds LIVE forward contexts only, so the overflow clear
// below 4096 entries - which would also forget live ones - is never reached by a training loop)
void ctx_forget(const void* ctx) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    g_ctx_shadowed.erase(ctx);
}

struct AttnDims {
    int B, L, d, H, Dh, K;
    long BL, HD, W3, slab;
    AttnDims(int B_, int L_, int d_, int H_, int Dh_, int K_) : B(B_), L(L_), d(d_), H(H_), Dh(Dh_), K(K_) {
        BL = (long)B * L; HD = (long)H * Dh; W3 = 3 * HD; slab = (long)L * (L + 1);
    }
};

// bump carving of caller-provided arenas; with base == nullptr it only measures
struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base(static_cast<char*>(b)) {}
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
    size_t floats() const { return (off + 255) / 4 + 64; }
};

// "fast" = bf16 activation pipeline on the glds kernels (prec 1 and 16-byte-aligned row pitches)
inline bool attn_fast(int prec, int d, int H, int Dh) { return prec == 1 && d % 8 == 0 && (H * Dh) % 8 == 0 && Dh % 4 == 0; }
inline bool ffn_fast(int prec, int d, int Di) { return prec == 1 && d % 8 == 0 && Di % 8 == 0; }

// element-typed pointer arithmetic on buffers that are f32 (parity path) or bf16 (fast path)
inline void* eoff(void* p, int dt, long n) { return static_cast<char*>(p) + n * (dt == DT_BF16 ? 2 : 4); }
inline const void* eoff(const void* p, int dt, long n) { return static_cast<const char*>(p) + n * (dt == DT_BF16 ? 2 : 4); }

GemmDesc mkx(const void* A, int adt, const void* B, int bdt, void* C, int cdt, int M, int N, int K, long lda, long ldb, long ldc,
             int flags, int prec) {
    GemmDesc d;
    d.A = A; d.B = B; d.C = C; d.a_dtype = adt; d.b_dtype = bdt; d.c_dtype = cdt;
    d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.flags = flags | (prec == 1 ? GEMM_BF16_MFMA : (prec == 2 ? GEMM_BF16X3 : 0));
    return d;
}

// fused (flash-style) attention core: bf16 pipeline, head dim 32/64
int g_flash_debug = 0;
int g_inkernel_pos = 1;                 // ttmi_set_option(8, 0): position term from the [B,H,L,L+1] slab instead of inside the attention kernels (A/B;
                                        // changes the ctx layout: set it between complete forward + backward passes only)
int g_disable_fused_attention = 0;      // ttmi_set_option(0, 1): A/B switch back to the unfused GEMM + softmax chain
inline bool attn_fused(bool fast, const AttnDims& a) {
    return fast && !g_disable_fused_attention && flash_supported(a.Dh, a.HD, a.W3, a.HD);
}
// the [B, H, L, L+1] score / position slab in the saved context: not needed when the fused kernels form the position term themselves
inline bool attn_inkernel(bool fast, const AttnDims& a) { return attn_fused(fast, a) && g_inkernel_pos && a.Dh % 8 == 0; }

struct AttnCtx {   // saved for backward.  act = f32 (parity) or bf16 (fast)
    void *qkv, *qu, *O;
    float *P, *s1, *mean, *rstd, *lse;     // P: f32 probabilities (unfused path) or the bf16 position-term slab (fused path, first half)
    bf16_t* x16 = nullptr;                 // bf16 copy of the input (fast): the qkv wgrad reads it again in backward
    bf16_t *wqkvT16 = nullptr, *woT16 = nullptr;   // transposed bf16 weights for the dgrads, made in forward from the same read as the plain copies
    bf16_t* E16s = nullptr;                // in-kernel position term: the effective table (bf16) and its bias for this length, gathered by the
    float* cTs = nullptr;                  // forward pass and read again by the backward kernels (no second gather launch)
    AttnCtx(Bump& b, const AttnDims& a, bool fast) {
        const size_t es = fast ? 2 : 4;
        qkv = b.take<char>(a.BL * a.W3 * es);
        qu = b.take<char>(attn_inkernel(fast, a) ? 64 : a.BL * a.HD * es);    // in-kernel variants form q + u themselves
        O = b.take<char>(a.BL * a.HD * es);
        P = b.take<float>(attn_inkernel(fast, a) ? 64 : (size_t)a.B * a.H * a.slab);
        s1 = b.take<float>(a.BL * a.d);
        mean = b.take<float>(a.BL);
        rstd = b.take<float>(a.BL);
        lse = b.take<float>((size_t)a.B * a.H * a.L);
        if (fast) {
            x16 = b.take<bf16_t>(a.BL * a.d);
            wqkvT16 = b.take<bf16_t>(a.W3 * a.d);
            woT16 = b.take<bf16_t>(a.HD * a.d);
            if (attn_inkernel(fast, a)) {
                E16s = b.take<bf16_t>((size_t)a.L * a.HD);
                cTs = b.take<float>((size_t)a.H * a.L);
            }
        }
    }
};

static inline size_t keep_al(size_t n) { return (n + 127) & ~(size_t)127; }      // bf16 elements, 256-byte granules
struct AttnWs {   // scratch (union of forward and backward needs)
    float *E, *cT, *dE, *dcT, *a, *dS, *dqkv, *delta;
    void* dO;
    bf16_t *wqkv16, *wo16, *dqkv16, *dres16, *dS16, *dG16, *E16, *kT16, *ET16;
    bf16_t *wqkv16lo = nullptr, *wo16lo = nullptr;      // second bf16 term of the two forward weights (option 13), whether or not a shadow serves the first
    long ldp, slab16;
    AttnWs(Bump& b, const AttnDims& a, bool fast, void* keep = nullptr) {
        E = b.take<float>((size_t)a.L * a.HD);
        cT = b.take<float>((size_t)a.H * a.L);
        dE = b.take<float>((size_t)a.L * a.HD + (size_t)a.H * a.L + 64);   // dE and dcT zeroed together
        dcT = dE + (size_t)a.L * a.HD;
        this->a = b.take<float>(a.BL * a.d);
        dS = b.take<float>((size_t)a.B * a.H * a.slab);
        dqkv = b.take<float>(a.BL * a.W3);
        delta = b.take<float>((size_t)a.B * a.H * a.L);
        dO = b.take<char>(a.BL * a.HD * (fast ? 2 : 4));
        wqkv16 = wo16 = dqkv16 = dres16 = dS16 = dG16 = E16 = kT16 = ET16 = nullptr;
        ldp = (a.L + 7) / 8 * 8;
        slab16 = (long)a.L * ldp;
        if (fast) {
            dS16 = b.take<bf16_t>((size_t)a.B * a.H * slab16);
            dG16 = b.take<bf16_t>((size_t)a.B * a.H * slab16);
            E16 = b.take<bf16_t>((size_t)a.L * a.HD);
            kT16 = b.take<bf16_t>((size_t)a.B * a.H * a.Dh * ldp);
            ET16 = b.take<bf16_t>((size_t)a.H * a.Dh * ldp);
            wqkv16 = b.take<bf16_t>(a.W3 * a.d);
            wo16 = b.take<bf16_t>(a.HD * a.d);
            wqkv16lo = b.take<bf16_t>(a.W3 * a.d);
            wo16lo = b.take<bf16_t>(a.HD * a.d);
            dqkv16 = b.take<bf16_t>(a.BL * a.W3);
            dres16 = b.take<bf16_t>(a.BL * a.d);
            if (keep) {      // deferred weight gradients: their A operands must outlive this call (ttmi_attn_bwd_defer)
                dqkv16 = static_cast<bf16_t*>(keep);
                dres16 = dqkv16 + keep_al((size_t)a.BL * a.W3);
            }
        }
    }
};


// what the layer-level backward hands the attention sub-layer instead of dy (ln_bwd_pair): the FFN's pre-norm operands and gradient targets
struct LnPairIn {
    const float *dh, *y, *mean1, *rstd1, *g1, *dres1;
    float *dgamma1, *dbeta1;
};

FlashParams flash_params(const AttnDims& a, const AttnCtx& c, float scale, int mask_kind, int mask_left, int mask_right,
                         const unsigned char* mask, long mask_sb, long mask_si) {
    FlashParams f;
    f.qu = static_cast<const bf16_t*>(c.qu);
    f.k = static_cast<const bf16_t*>(c.qkv) + a.HD;
    f.v = static_cast<const bf16_t*>(c.qkv) + 2 * a.HD;
    f.ld_qu = a.HD; f.ld_kv = a.W3; f.ld_o = a.HD;
    f.bd = reinterpret_cast<const bf16_t*>(c.P) + a.L; f.slab = a.slab;      // fused path: the slab is bf16 (in the same arena)
    f.o = static_cast<bf16_t*>(c.O);
    f.lse = c.lse;
    f.B = a.B; f.L = a.L; f.H = a.H; f.Dh = a.Dh; f.scale = scale;
    f.mask_kind = mask_kind; f.mask_left = mask_left; f.mask_right = mask_right;
    f.mask = mask; f.mask_sb = mask_sb; f.mask_si = mask_si;
    f.debug = g_flash_debug;
    return f;
}

// position term formed inside the attention kernels: plain q (first third of the qkv rows), the effective table in bf16 and its bias
void flash_inkernel(FlashParams& f, const AttnDims& a, const AttnCtx& c, const bf16_t* E16, const float* cT, const float* r_w_bias) {
    f.bd = nullptr;
    f.qu = nullptr;
    f.u = r_w_bias;
    f.qp = static_cast<const bf16_t*>(c.qkv);
    f.ld_qp = a.W3;
    f.e16 = E16;
    f.ld_e = a.HD;
    f.cT = cT;
}

int memset2d(void* p, size_t pitch, size_t width, size_t height, hipStream_t st) {
    return fill_zero2d(p, pitch, width, height, st);        // a kernel, not a memset node (rowops.hip: graph replays)
}

// batched over (b, h): z1 = b, z2 = h
void batch_bh(GemmDesc& g, const AttnDims& a, long sA1, long sA2, long sB1, long sB2, long sC1, long sC2) {
    g.nz1 = a.B; g.nz2 = a.H;
    g.sA1 = sA1; g.sA2 = sA2; g.sB1 = sB1; g.sB2 = sB2; g.sC1 = sC1; g.sC2 = sC2;
}

}  // namespace

extern "C" {

size_t ttmi_attn_ctx_floats(int B, int L, int d, int H, int Dh, int prec) {
    Bump b(nullptr);
    AttnCtx c(b, AttnDims(B, L, d, H, Dh, 1), attn_fast(prec, d, H, Dh));
    return b.floats();
}
// bf16x3 (prec 2): scratch of the sub-layer's dense products in three bf16 terms, behind the ordinary workspace (one product at a time)
static size_t attn_x3_floats(const AttnDims& a) {
    size_t m = x3_nt_elems(a.BL, a.W3, a.d);
    const size_t c[] = {x3_nt_elems(a.BL, a.d, (int)a.HD), x3_tn_elems(a.BL, a.d, (int)a.HD), x3_nt_elems(a.BL, a.HD, a.d),
                        x3_tn_elems(a.BL, (int)a.W3, a.d), x3_nt_elems(a.BL, a.d, (int)a.W3)};
    for (size_t v : c) m = v > m ? v : m;
    return (m + 1) / 2 + 128;
}
size_t ttmi_attn_ws_floats(int B, int L, int d, int H, int Dh, int prec) {
    Bump b(nullptr);
    const AttnDims a(B, L, d, H, Dh, 1);
    AttnWs w(b, a, attn_fast(prec, d, H, Dh));
    return b.floats() + (prec == 2 ? 64 + attn_x3_floats(a) : 0);
}

// RelLearnableMultiHeadAttn.forward (tt/transformer.py:106-177), batch-major: x,y f32 [B,L,d].
// mask_kind: 0 none, 1 causal (look_ahead_mask), 2 band(left,right) (context_mask), 3 uint8 tensor (b,i,j) at
// mask[b*mask_sb + i*mask_si + j], nonzero = masked.  prec: 0 = exact-f32 MFMA, 1 = bf16 MFMA (bf16 q/k/v/O in HBM,
// dense projections on the glds kernels).  ctx/ws must be 256-byte aligned.
// x16_in (optional, bf16 pipeline): the bf16 copy of x, made by whoever produced x (the previous layer's last norm): no conversion pass here.
// pre (optional): the norm the NEXT sub-layer applies to y (the FFN's pre-norm), formed in the same pass as y.
static int attn_fwd_impl(const float* x, const float* qkv_w, const float* o_w, const float* ln_g, const float* ln_b,
                  const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K,
                  int mask_kind, int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si,
                  int prec, float p_drop, unsigned seed, float* ctx, float* ws, float* y, const bf16_t* x16_in, const LnPreNorm* pre,
                  void* stream) {
    TTMI_REQUIRE(x && qkv_w && o_w && ln_g && ln_b && r_emb && r_w_bias && r_bias && ctx && ws && y, "attn_fwd: null pointer");
    TTMI_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "attn_fwd: dropout probability %f outside [0,1)", p_drop);
    TTMI_REQUIRE(B > 0 && L > 0 && d > 0 && H > 0 && Dh > 0 && K > 0, "attn_fwd: bad dims");
    TTMI_REQUIRE(mask_kind >= 0 && mask_kind <= 4, "attn_fwd: bad mask kind %d", mask_kind);
    TTMI_REQUIRE(((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(ws)) & 255) == 0, "attn_fwd: ctx/ws must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const AttnDims a(B, L, d, H, Dh, K);
    const bool fast = attn_fast(prec, d, H, Dh);
    const int adt = fast ? DT_BF16 : DT_F32;
    Bump bc(ctx), bw(ws);
    AttnCtx c(bc, a, fast);
    AttnWs w(bw, a, fast);
    bf16_t* x3 = prec == 2 ? reinterpret_cast<bf16_t*>(ws + ((bw.floats() + 63) & ~(size_t)63)) : nullptr;     // bf16x3: split operands of the dense products
    const float scale = 1.0f / sqrtf((float)Dh);
    // 1. qkv = x Wqkv^T ; 2. qu = q + r_w_bias
    TTMI_REQUIRE(!x16_in || (fast && aligned16(x16_in)), "attn_fwd: a bf16 copy of x is taken by the bf16 pipeline only (16-byte aligned)");
    if (fast) {
        if (!x16_in) CK(convert_bf16(x, c.x16, a.BL * d, st));
        const bf16_t* x16 = x16_in ? x16_in : c.x16;
        Shadow sh;
        const bf16_t* wqkv16 = w.wqkv16;
        const bool shq = shadow_of(qkv_w, (int)a.W3, d, a.W3, sh);
        ctx_record(ctx, 0, shq);
        if (shq) wqkv16 = sh.w16;                                                               // kept current by the optimiser step
        else CK(transpose_convert_bf16(qkv_w, (int)a.W3, d, c.wqkvT16, a.W3, st, w.wqkv16));   // Wqkv (bf16) and Wqkv^T [d, W3] for backward
        NtEpilogue eq;
        if (split_w(a.BL, false)) {
            if (shq && sh.w16lo) eq.B_lo = sh.w16lo;       // (kept current by the optimiser step with the shadow itself)
            else { CK(bf16_residual(qkv_w, wqkv16, w.wqkv16lo, (long)a.W3 * d, st)); eq.B_lo = w.wqkv16lo; }
        }
        CK(gemm_nt_bf16(x16, wqkv16, c.qkv, 1, eq, (int)a.BL, (int)a.W3, d, d, d, a.W3, st));
        if (!attn_inkernel(fast, a))
            CK(add_row_bias_bf16(static_cast<bf16_t*>(c.qkv), a.W3, r_w_bias, a.BL, (int)a.HD, static_cast<bf16_t*>(c.qu), a.HD, st));
    } else {
        if (x3 && x3_worth(a.BL, a.W3, d)) CK(x3_nt(x, qkv_w, static_cast<float*>(c.qkv), (int)a.BL, (int)a.W3, d, d, d, a.W3, NtEpilogue(), x3, st));
        else CK(ttmi_launch_gemm(mk(x, qkv_w, static_cast<float*>(c.qkv), (int)a.BL, (int)a.W3, d, d, d, a.W3, NT_, prec), st));
        CK(add_row_bias(static_cast<float*>(c.qkv), a.W3, r_w_bias, a.BL, (int)a.HD, static_cast<float*>(c.qu), a.HD, st));
    }
    // 3. effective tables for this length (clamped rows when L > K); the fused kernels' bf16 copy comes out of the same launch
    const bool inkernel = attn_inkernel(fast, a);
    CK(relpos_gather(r_emb, r_bias, K, L, H, Dh, w.E, inkernel ? c.cTs : w.cT, st, inkernel ? c.E16s : nullptr));
    // 4. G = q E^T + c into the pitch-(L+1) slab, column 0 zero - unless the fused kernels form the position term themselves
    if (inkernel) {
    } else if (attn_fused(fast, a) && g_gemm_slab == 0 && (size_t)16 * (L + 1) * 4 + 32 <= 160 * 1024) {
        // write-bound: dedicated kernel that streams whole slab rows (column 0 included) instead of a batched GEMM + strided memset
        CK(convert_bf16(w.E, w.E16, (long)L * a.HD, st));
        CK(relpos_slab(static_cast<const bf16_t*>(c.qkv), a.W3, w.E16, a.HD, w.cT, B, L, H, Dh, reinterpret_cast<bf16_t*>(c.P), st));
    } else if (attn_fused(fast, a) && Dh % 8 == 0) {
        bf16_t* P16 = reinterpret_cast<bf16_t*>(c.P);
        CK(memset2d(P16, (size_t)(L + 1) * 2, 2, (size_t)B * H * L, st));
        // on the glds kernel: q (bf16, in place in qkv) x E16^T, batched over (b, h), bias c, pitch-(L+1) bf16 output
        CK(convert_bf16(w.E, w.E16, (long)L * a.HD, st));
        FastBatch fb;
        fb.nz1 = B; fb.nz2 = H; fb.sA1 = L * a.W3; fb.sA2 = Dh; fb.sB1 = 0; fb.sB2 = Dh; fb.sC1 = H * a.slab; fb.sC2 = a.slab;
        fb.sV1 = 0; fb.sV2 = L;
        NtEpilogue e;
        e.bias = w.cT;
        CK(gemm_nt_bf16(static_cast<const bf16_t*>(c.qkv), w.E16, P16 + 1, 1, e, L, L, Dh, a.W3, a.HD, L + 1, st, fb));
    } else {
        CK(memset2d(c.P, (size_t)(L + 1) * 4, 4, (size_t)B * H * L, st));
        GemmDesc g = mkx(c.qkv, adt, w.E, DT_F32, c.P + 1, DT_F32, L, L, Dh, a.W3, a.HD, L + 1, NT_ | GEMM_BIAS, prec);
        batch_bh(g, a, L * a.W3, Dh, 0, Dh, H * a.slab, a.slab);
        g.bias = w.cT; g.sBias1 = 0; g.sBias2 = L;
        CK(ttmi_launch_gemm(g, st));
    }
    if (attn_fused(fast, a)) {
        // 5-7 fused: softmax((q+u) k^T + shifted(G)) V without materialising the probabilities
        FlashParams f = flash_params(a, c, scale, mask_kind, mask_left, mask_right, mask, mask_sb, mask_si);
        if (inkernel) flash_inkernel(f, a, c, c.E16s, c.cTs, r_w_bias);
        CK(flash_attn_fwd(f, st));
    } else {
        // 5. S = shifted(G) + (q+u) k^T, accumulated through the pitch-L view
        {
            GemmDesc g = mkx(c.qu, adt, eoff(c.qkv, adt, a.HD), adt, c.P + L, DT_F32, L, L, Dh, a.HD, a.W3, L, NT_, prec);
            batch_bh(g, a, L * a.HD, Dh, L * a.W3, Dh, H * a.slab, a.slab);
            g.beta = 1.f;
            CK(ttmi_launch_gemm(g, st));
        }
        // 6. P = softmax(scale * S, masked)
        MaskDesc m;
        m.kind = mask_kind; m.left = mask_left; m.right = mask_right; m.ptr = mask; m.sb = mask_sb; m.si = mask_si;
        CK(softmax_fwd(c.P + L, B, H, L, L, a.slab, scale, m, st));
        // 7. O = P V
        {
            GemmDesc g = mkx(c.P + L, DT_F32, eoff(c.qkv, adt, 2 * a.HD), adt, c.O, adt, L, Dh, L, L, a.W3, a.HD, NN_, prec);
            batch_bh(g, a, H * a.slab, a.slab, L * a.W3, Dh, L * a.HD, Dh);
            CK(ttmi_launch_gemm(g, st));
        }
    }
    // 8. a = O Wo^T ; 9. y = LN(x + a)
    if (fast) {
        Shadow sh;
        const bf16_t* wo16 = w.wo16;
        const bool sho = shadow_of(o_w, d, (int)a.HD, d, sh);
        ctx_record(ctx, 1, sho);
        if (sho) wo16 = sh.w16;
        else CK(transpose_convert_bf16(o_w, d, (int)a.HD, c.woT16, d, st, w.wo16));              // Wo (bf16) and Wo^T [HD, d] for backward
        NtEpilogue eo;
        if (split_w(a.BL, true)) {
            if (sho && sh.w16lo) eo.B_lo = sh.w16lo;
            else { CK(bf16_residual(o_w, wo16, w.wo16lo, (long)d * a.HD, st)); eo.B_lo = w.wo16lo; }
        }
        CK(gemm_nt_bf16(static_cast<bf16_t*>(c.O), wo16, w.a, 0, eo, (int)a.BL, d, (int)a.HD, a.HD, a.HD, d, st));
    } else if (x3 && x3_worth(a.BL, d, a.HD)) {
        CK(x3_nt(static_cast<float*>(c.O), o_w, w.a, (int)a.BL, d, (int)a.HD, a.HD, a.HD, d, NtEpilogue(), x3, st));
    } else {
        CK(ttmi_launch_gemm(mk(static_cast<float*>(c.O), o_w, w.a, (int)a.BL, d, (int)a.HD, a.HD, a.HD, d, NT_, prec), st));
    }
    DropSpec rd;                                            // self.drop(attn_out), tt/transformer.py:173
    rd.p = p_drop; rd.seed = seed ^ 0xA1u; rd.salt = g_drop_salt;
    CK(ln_fwd(x, w.a, ln_g, ln_b, a.BL, d, 1e-5f, c.s1, y, c.mean, c.rstd, st, nullptr, rd, DropSpec(), pre));
    return TTMI_OK;
}

int ttmi_attn_fwd(const float* x, const float* qkv_w, const float* o_w, const float* ln_g, const float* ln_b,
                  const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K,
                  int mask_kind, int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si,
                  int prec, float p_drop, unsigned seed, float* ctx, float* ws, float* y, void* stream) {
    return attn_fwd_impl(x, qkv_w, o_w, ln_g, ln_b, r_emb, r_w_bias, r_bias, B, L, d, H, Dh, K, mask_kind, mask_left, mask_right, mask, mask_sb,
                         mask_si, prec, p_drop, seed, ctx, ws, y, nullptr, nullptr, stream);
}

// Backward of ttmi_attn_fwd.  dx is written; every g_* buffer is ACCUMULATED into (zero them per step).
// keep / out != nullptr: the two weight-gradient GEMMs are described in out[0..1] instead of launched (ttmi_attn_bwd_defer).
static int attn_bwd_impl(const float* dy, const float* x, const float* qkv_w, const float* o_w, const float* ln_g,
                  const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K, int mask_kind,
                  int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec,
                  float p_drop, unsigned seed, const float* ctx, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                  float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, void* keep, ttmi_wgrad_desc* out,
                  void* stream, const bf16_t* x16_in = nullptr, const LnPairIn* pair = nullptr) {
    TTMI_REQUIRE((dy || pair) && x && qkv_w && o_w && ln_g && r_emb && r_bias && ctx && ws && dx, "attn_bwd: null pointer");
    TTMI_REQUIRE(g_qkv_w && g_o_w && g_ln_g && g_ln_b && g_r_emb && g_r_w_bias && g_r_bias, "attn_bwd: null gradient pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const AttnDims a(B, L, d, H, Dh, K);
    const bool fast = attn_fast(prec, d, H, Dh);
    const int adt = fast ? DT_BF16 : DT_F32;
    Bump bc(const_cast<float*>(ctx)), bw(ws);
    AttnCtx c(bc, a, fast);
    AttnWs w(bw, a, fast, keep);
    bf16_t* x3 = prec == 2 ? reinterpret_cast<bf16_t*>(ws + ((bw.floats() + 63) & ~(size_t)63)) : nullptr;
    TTMI_REQUIRE(!out || (fast && keep && aligned16(keep)), "attn_bwd_defer: bf16 pipeline and a 16-byte aligned keep buffer required");
    const float scale = 1.0f / sqrtf((float)Dh);
    // 1. dres = LN'(dy) -> dx (doubles as the residual gradient)
    // 2. gWo += da^T O ; 3. dO = da Wo, with da = dres * dropout mask of the forward (dres itself stays the residual grad);
    //    the bf16 pipeline gets da from the same pass
    DropSpec rd;
    rd.p = p_drop; rd.seed = seed ^ 0xA1u; rd.salt = g_drop_salt;
    if (pair) {          // the layer-level call: dy is still in two pieces (the FFN's pre-norm input gradient and its residual branch)
        TTMI_REQUIRE(fast, "attn_bwd: the paired LayerNorm backward belongs to the bf16 pipeline");
        CK(ln_bwd_pair(pair->dh, pair->y, pair->mean1, pair->rstd1, pair->g1, pair->dres1, c.s1, c.mean, c.rstd, ln_g, a.BL, d, dx, pair->dgamma1,
                       pair->dbeta1, g_ln_g, g_ln_b, w.dres16, rd, st));
    } else {
        CK(ln_bwd(dy, c.s1, c.mean, c.rstd, ln_g, nullptr, a.BL, d, dx, g_ln_g, g_ln_b, st, DropSpec(), fast ? w.dres16 : nullptr, rd, nullptr));
    }
    const bf16_t* x16 = x16_in ? x16_in : c.x16;
    const float* da = dx;
    if (!fast && p_drop > 0.f) {
        CK(dropout_apply(dx, a.BL * d, rd, w.a, nullptr, st));
        da = w.a;
    }
    if (fast) {
        int rc = TTMI_OK;
        const bf16_t* woT16 = ctx_weightT(ctx, 1, o_w, d, (int)a.HD, d, c.woT16, st, &rc);
        CK(rc);
        if (out) out[0] = ttmi_wgrad_desc{w.dres16, c.O, g_o_w, nullptr, d, (int)a.HD, (int)a.BL, (long)d, (long)a.HD, (long)a.HD};
        else CK(gemm_tn_bf16(w.dres16, static_cast<bf16_t*>(c.O), g_o_w, d, (int)a.HD, (int)a.BL, d, a.HD, a.HD, 1, fork_stream(st)));
        CK(gemm_nt_bf16(w.dres16, woT16, w.dO, 1, nullptr, (int)a.BL, (int)a.HD, d, d, d, a.HD, st));
    } else if (x3 && x3_worth(a.BL, a.HD, d)) {
        CK(x3_tn(da, static_cast<float*>(c.O), g_o_w, d, (int)a.HD, a.BL, d, a.HD, a.HD, x3, st));
        CK(x3_nn(da, o_w, static_cast<float*>(w.dO), (int)a.BL, (int)a.HD, d, d, a.HD, a.HD, NtEpilogue(), x3, st, true));
    } else {
        CK(wgrad(da, static_cast<float*>(c.O), g_o_w, d, (int)a.HD, (int)a.BL, d, a.HD, a.HD, prec, st));
        CK(ttmi_launch_gemm(mk(da, o_w, static_cast<float*>(w.dO), (int)a.BL, (int)a.HD, d, d, a.HD, a.HD, NN_, prec), st));
    }
    const bool fused = attn_fused(fast, a);
    const bool fastpos = fused && Dh % 8 == 0;      // position/content products on the glds kernels (K-major transposed k, E)
    bool direct_tables = false;
    // Batch slices (ttmi_set_option(10, n)): the attention backward kernel leaves dS twice in bf16 ([B, H, L, ldp] each: 2 x 129 MB at C2) and
    // the dq / dE products read them straight back.  Cut into n slices of the batch - kernel, dq product, dE product per slice, over the SAME
    // slab region - the slabs of a slice (2 x 129 / n MB) are still in the 256 MB Infinity Cache when their readers run.
    const int nslice = (fastpos && g_attn_slices > 1) ? (g_attn_slices < B ? g_attn_slices : B) : 1;
    if (fused) {
        // 4-6 + 9 fused: recompute P, dS = P (dP - delta) scale, dK and dV straight into dqkv.  dS leaves the kernel twice in
        // bf16 with aligned rows: [i][j] for the content dgrad and the shifted [r][c-1] form (= dG) for the position grads.
        const int bper = (B + nslice - 1) / nslice;
        // one pass over the slabs for dq, dE, dc and d r_w_bias (attn_dqde_kernel) where its shape fits, the round-2 GEMM launches otherwise
        const bool onepass = fastpos && !g_posgrad_gemms && attn_dqde_supported(Dh, L, w.ldp);
        // ... which then also folds its table-gradient rows straight into r_emb / r_bias' gradients (no dE / dc round trip, no relpos_scatter
        // launch: 18 x ~20 us per C2 step), unless so many rows clamp onto table row 0 that the chunked fold is the better tool (C5)
        direct_tables = onepass && L - K < 256 && !g_scatter_launch;
        // ... by plain per-(b, h) stores + one small reduction over b on the fork stream (no atomics: 33 MB of them per C2 layer were 27 us of
        // attn_dqde_kernel's 97) where the f32 dS slab region has the room (dq partials of the column groups first) and the batch is not sliced
        const int ngrp = attn_dqde_groups(w.ldp);
        const long part_dq = ngrp > 1 ? (long)ngrp * B * L * a.HD : 0;
        const bool part_tables = direct_tables && nslice == 1 && g_scatter_launch != 2 &&
                                 part_dq + (long)B * L * a.HD + (long)B * H * L <= (long)B * H * a.slab;
        float* part_e = part_tables ? w.dS + part_dq : nullptr;
        float* part_c = part_tables ? part_e + (long)B * L * a.HD : nullptr;
        // the small zero fills of this pass (dG row 0 of every slab, the dE / dc accumulators) ride in the delta kernel's launch where one
        // launch covers the batch; sliced runs keep the memsets (the accumulators must survive from slice to slice)
        const bool fillz = fastpos && nslice == 1;
        if (!fillz) CK(memset2d(w.dG16, (size_t)w.slab16 * 2, (size_t)w.ldp * 2, (size_t)bper * H, st));       // dG row 0 is (almost) never written
        FlashParams f = flash_params(a, c, scale, mask_kind, mask_left, mask_right, mask, mask_sb, mask_si);
        const bf16_t* E16 = w.E16;
        if (attn_inkernel(fast, a)) {                // the kernel recomputes the position term from the table + bias the forward pass gathered for this length
            TTMI_REQUIRE(r_w_bias, "attn_bwd: r_w_bias is required (the attention kernels form q + r_w_bias themselves)");
            if (!onepass) CK(relpos_gather(r_emb, r_bias, K, L, H, Dh, w.E, w.cT, st, w.E16));     // (the GEMM launches transpose the f32 table)
            else E16 = c.E16s;
            flash_inkernel(f, a, c, c.E16s, c.cTs, r_w_bias);
        } else if (fastpos) {
            CK(relpos_gather(r_emb, r_bias, K, L, H, Dh, w.E, w.cT, st, w.E16)); // the effective table of the position products below
        }
        f.dO = static_cast<const bf16_t*>(w.dO);
        f.delta = w.delta;
        f.dS16 = w.dS16; f.dG16 = w.dG16; f.ldp = w.ldp; f.slab16 = w.slab16;
        f.dK = w.dqkv + a.HD; f.dV = w.dqkv + 2 * a.HD; f.ld_dkv = a.W3;
        f.dK16 = w.dqkv16 + a.HD; f.dV16 = w.dqkv16 + 2 * a.HD;
        if (fillz) { f.zero_f32 = w.dE; f.zero_n = (long)L * a.HD + (long)H * L; f.zero_dg_row0 = 1; }
        if (!fastpos) CK(flash_attn_bwd(f, st));
        else {
            if (!fillz) CK(fill_zero(w.dE, sizeof(float) * ((size_t)L * a.HD + (size_t)H * L), st));
            if (!onepass) {
                // K-major operands of the position / content products, for the whole batch, ahead of the slices
                // (K runs over the padded pitch ldp: the pad columns of dS16 / dG16 are zeroed by the flash backward kernel itself)
                CK(transpose_bf16_batched(static_cast<const bf16_t*>(c.qkv) + a.HD, 1, a.W3, B, H, L * a.W3, Dh, L, Dh, w.kT16, w.ldp, st));
                CK(transpose_bf16_batched(w.E, 0, a.HD, 1, H, 0, Dh, L, Dh, w.ET16, w.ldp, st));
            }
            for (int b0 = 0; b0 < B; b0 += bper) {
                const int nb = B - b0 < bper ? B - b0 : bper;
                FlashParams fs = f;
                fs.B = nb;
                const long r0 = (long)b0 * L;
                if (fs.qu) fs.qu += r0 * fs.ld_qu;
                if (fs.qp) fs.qp += r0 * fs.ld_qp;
                fs.k += r0 * fs.ld_kv; fs.v += r0 * fs.ld_kv;
                fs.o += r0 * fs.ld_o; fs.dO += r0 * fs.ld_o;
                fs.lse += (long)b0 * H * L; fs.delta += (long)b0 * H * L;
                fs.dK += r0 * fs.ld_dkv; fs.dV += r0 * fs.ld_dkv;
                fs.dK16 += r0 * fs.ld_dkv; fs.dV16 += r0 * fs.ld_dkv;
                if (fs.mask) fs.mask += (long)b0 * fs.mask_sb * (fs.mask_kind == 4 ? 4 : 1);
                if (fs.bd) fs.bd += (long)b0 * H * fs.slab;
                CK(flash_attn_bwd(fs, st));
                if (onepass) {
                    const bf16_t* qkv16 = static_cast<const bf16_t*>(c.qkv) + r0 * a.W3;
                    // (sequences longer than 512: the column groups' f32 partial dq rows live in the f32 dS slab region, unused on this path:
                    // B H L (L + 1) floats against groups x B L H 64)
                    CK(attn_dqde(w.dS16, w.dG16, w.slab16, w.ldp, qkv16 + a.HD, a.W3, E16, a.HD, qkv16, a.W3, w.dqkv16 + r0 * a.W3, a.W3,
                                 w.dE, a.HD, w.dcT, g_r_w_bias, nb, L, H, st, w.dS, direct_tables ? g_r_emb : nullptr, direct_tables ? g_r_bias : nullptr, K,
                                 part_e, part_c));
                    if (part_tables) CK(attn_table_grads(part_e, part_c, nb, L, H, K, g_r_emb, g_r_bias, fork_stream(st)));
                    continue;
                }
                // dq = dS k + dG E in ONE launch: both products accumulate into the same tile, the column sums of the first (d r_w_bias) are taken
                // in between, and the sum leaves in bf16 - the form the qkv GEMMs read
                FastBatch fb;
                fb.nz1 = nb; fb.nz2 = H; fb.sA1 = H * w.slab16; fb.sA2 = w.slab16; fb.sB1 = (long)H * Dh * w.ldp; fb.sB2 = (long)Dh * w.ldp;
                fb.sC1 = L * a.W3; fb.sC2 = Dh; fb.sV1 = 0; fb.sV2 = Dh;
                NtEpilogue e;
                e.A2 = w.dG16; e.B2 = w.ET16; e.K2 = (int)w.ldp; e.lda2 = w.ldp; e.ldb2 = w.ldp; e.sB1b = 0; e.sB2b = (long)Dh * w.ldp;
                e.colsum_mid = g_r_w_bias;
                CK(gemm_nt_bf16(w.dS16, w.kT16 + (long)b0 * H * Dh * w.ldp, w.dqkv16 + r0 * a.W3, 1, e, L, Dh, (int)w.ldp, w.ldp, w.ldp, a.W3, st, fb));
                // dE[p,h,:] += sum_b dG^T q, dc[h][p] += sum_b colsum(dG)
                FastBatch tb;
                tb.nz1 = nb; tb.nz2 = H; tb.sA1 = H * w.slab16; tb.sA2 = w.slab16; tb.sB1 = L * a.W3; tb.sB2 = Dh; tb.sC1 = 0; tb.sC2 = Dh;
                tb.sV1 = 0; tb.sV2 = L;
                CK(gemm_tn_bf16(w.dG16, static_cast<const bf16_t*>(c.qkv) + r0 * a.W3, w.dE, L, Dh, L, w.ldp, a.W3, a.HD, 1, st, w.dcT, tb));
            }
        }
    } else {
        // first L floats of each dS slab lie outside the pitch-L view but inside dG's row 0: zero them
        CK(memset2d(w.dS, (size_t)a.slab * 4, (size_t)L * 4, (size_t)B * H, st));
        // 4. dP = dO V^T through the pitch-L view of the dS slab (first L floats of each slab are outside the view)
        {
            GemmDesc g = mkx(w.dO, adt, eoff(c.qkv, adt, 2 * a.HD), adt, w.dS + L, DT_F32, L, L, Dh, a.HD, a.W3, L, NT_, prec);
            batch_bh(g, a, L * a.HD, Dh, L * a.W3, Dh, H * a.slab, a.slab);
            CK(ttmi_launch_gemm(g, st));
        }
        // 5. dV = P^T dO
        {
            GemmDesc g = mkx(c.P + L, DT_F32, w.dO, adt, w.dqkv + 2 * a.HD, DT_F32, L, Dh, L, L, a.HD, a.W3, TN_, prec);
            batch_bh(g, a, H * a.slab, a.slab, L * a.HD, Dh, L * a.W3, Dh);
            CK(ttmi_launch_gemm(g, st));
        }
        // 6. dS = P (dP - rowsum(dP P)) scale
        CK(softmax_bwd(w.dS + L, c.P + L, B * H, L, L, a.slab, scale, st));
    }
    // 7. dq(content) = dS K -> dqkv[q]   (fastpos: issued per batch slice above, together with the position part)
    if (!fastpos) {
        GemmDesc g = fused ? mkx(w.dS16, DT_BF16, eoff(c.qkv, adt, a.HD), adt, w.dqkv, DT_F32, L, Dh, L, w.ldp, a.W3, a.W3, NN_, prec)
                           : mkx(w.dS + L, DT_F32, eoff(c.qkv, adt, a.HD), adt, w.dqkv, DT_F32, L, Dh, L, L, a.W3, a.W3, NN_, prec);
        if (fused) batch_bh(g, a, H * w.slab16, w.slab16, L * a.W3, Dh, L * a.W3, Dh);
        else batch_bh(g, a, H * a.slab, a.slab, L * a.W3, Dh, L * a.W3, Dh);
        CK(ttmi_launch_gemm(g, st));
    }
    // 8. g r_w_bias += column sums of dq(content)  (fastpos: taken from the accumulators of the fused launch)
    if (!fastpos) CK(colsum(w.dqkv, a.W3, a.BL, (int)a.HD, 1, 1, 0, 0, 0, 0, g_r_w_bias, st));
    if (!fused) {
        // 9. dK = dS^T (q + u)
        {
            GemmDesc g = mkx(w.dS + L, DT_F32, c.qu, adt, w.dqkv + a.HD, DT_F32, L, Dh, L, L, a.HD, a.W3, TN_, prec);
            batch_bh(g, a, H * a.slab, a.slab, L * a.HD, Dh, L * a.W3, Dh);
            CK(ttmi_launch_gemm(g, st));
        }
    }
    // 10. dq += dG E ; 11. dE[p,h,:] = sum_b dG^T q ; 12. dc[h][p] = sum_b colsum(dG) ; 13. fold onto the K-row tables
    if (!fastpos) {
        if (!(fused && attn_inkernel(fast, a))) CK(relpos_gather(r_emb, r_bias, K, L, H, Dh, w.E, w.cT, st));     // (already gathered for the kernel above)
        CK(fill_zero(w.dE, sizeof(float) * ((size_t)L * a.HD + (size_t)H * L), st));
        {
            GemmDesc g = fused ? mkx(w.dG16, DT_BF16, w.E, DT_F32, w.dqkv, DT_F32, L, Dh, L, w.ldp, a.HD, a.W3, NN_, prec)
                               : mkx(w.dS + 1, DT_F32, w.E, DT_F32, w.dqkv, DT_F32, L, Dh, L, L + 1, a.HD, a.W3, NN_, prec);
            if (fused) batch_bh(g, a, H * w.slab16, w.slab16, 0, Dh, L * a.W3, Dh);
            else batch_bh(g, a, H * a.slab, a.slab, 0, Dh, L * a.W3, Dh);
            g.beta = 1.f;
            CK(ttmi_launch_gemm(g, st));
        }
        {
            GemmDesc g = fused ? mkx(w.dG16, DT_BF16, c.qkv, adt, w.dE, DT_F32, L, Dh, L, w.ldp, a.W3, a.HD, TN_ | GEMM_ATOMIC, prec)
                               : mkx(w.dS + 1, DT_F32, c.qkv, adt, w.dE, DT_F32, L, Dh, L, L + 1, a.W3, a.HD, TN_ | GEMM_ATOMIC, prec);
            if (fused) batch_bh(g, a, H * w.slab16, w.slab16, L * a.W3, Dh, 0, Dh);
            else batch_bh(g, a, H * a.slab, a.slab, L * a.W3, Dh, 0, Dh);
            CK(ttmi_launch_gemm(g, st));
        }
        if (fused) CK(colsum_bf16(w.dG16, w.ldp, L, L, w.dcT, st, B, H, H * w.slab16, w.slab16, L));
        else CK(colsum(w.dS + 1, L + 1, L, L, B, H, H * a.slab, a.slab, 0, L, w.dcT, st));
    }
    if (!direct_tables) CK(relpos_scatter(w.dE, w.dcT, K, L, H, Dh, g_r_emb, g_r_bias, st));
    // 14. gWqkv += dqkv^T x ; 15. dx += dqkv Wqkv
    if (fast) {
        if (!fastpos) CK(convert_bf16(w.dqkv, w.dqkv16, a.BL * a.W3, st));      // fastpos: dq / dK / dV were written in bf16 by their producers
        if (out) out[1] = ttmi_wgrad_desc{w.dqkv16, x16, g_qkv_w, nullptr, (int)a.W3, d, (int)a.BL, (long)a.W3, (long)d, (long)d};
        else {
            CK(gemm_tn_bf16(w.dqkv16, x16, g_qkv_w, (int)a.W3, d, (int)a.BL, a.W3, d, d, 1, fork_stream(st)));
        }
        NtEpilogue e;
        e.addend = dx;
        int rc = TTMI_OK;
        const bf16_t* wqkvT16 = ctx_weightT(ctx, 0, qkv_w, (int)a.W3, d, a.W3, c.wqkvT16, st, &rc);
        CK(rc);
        CK(gemm_nt_bf16(w.dqkv16, wqkvT16, dx, 0, e, (int)a.BL, d, (int)a.W3, a.W3, a.W3, d, st));
        ctx_forget(ctx);
    } else if (x3 && x3_worth(a.BL, d, a.W3)) {
        CK(x3_tn(w.dqkv, x, g_qkv_w, (int)a.W3, d, a.BL, a.W3, d, d, x3, st));
        NtEpilogue ea;
        ea.addend = dx;                                           // dx += dqkv Wqkv (dx holds the residual-branch gradient)
        CK(x3_nn(w.dqkv, qkv_w, dx, (int)a.BL, d, (int)a.W3, a.W3, d, d, ea, x3, st, true));
    } else {
        CK(wgrad(w.dqkv, x, g_qkv_w, (int)a.W3, d, (int)a.BL, a.W3, d, d, prec, st));
        GemmDesc g = mk(w.dqkv, qkv_w, dx, (int)a.BL, d, (int)a.W3, a.W3, d, d, NN_, prec);
        g.beta = 1.f;
        CK(ttmi_launch_gemm(g, st));
    }
    join_stream(st);
    return TTMI_OK;
}

int ttmi_attn_bwd(const float* dy, const float* x, const float* qkv_w, const float* o_w, const float* ln_g,
                  const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K, int mask_kind,
                  int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec,
                  float p_drop, unsigned seed, const float* ctx, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                  float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, void* stream) {
    return attn_bwd_impl(dy, x, qkv_w, o_w, ln_g, r_emb, r_w_bias, r_bias, B, L, d, H, Dh, K, mask_kind, mask_left, mask_right, mask, mask_sb,
                         mask_si, prec, p_drop, seed, ctx, ws, dx, g_qkv_w, g_o_w, g_ln_g, g_ln_b, g_r_emb, g_r_w_bias, g_r_bias, nullptr,
                         nullptr, stream);
}
size_t ttmi_attn_bwd_keep_bytes(int B, int L, int d, int H, int Dh) {
    const size_t BL = (size_t)B * L;
    return 2 * (keep_al(BL * 3 * H * Dh) + keep_al(BL * d));
}
int ttmi_attn_bwd_defer(const float* dy, const float* x, const float* qkv_w, const float* o_w, const float* ln_g,
                        const float* r_emb, const float* r_w_bias, const float* r_bias, int B, int L, int d, int H, int Dh, int K, int mask_kind,
                        int mask_left, int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec,
                        float p_drop, unsigned seed, const float* ctx, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                        float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, void* keep,
                        ttmi_wgrad_desc* out, void* stream) {
    TTMI_REQUIRE(keep && out, "attn_bwd_defer: keep buffer and descriptor array required");
    return attn_bwd_impl(dy, x, qkv_w, o_w, ln_g, r_emb, r_w_bias, r_bias, B, L, d, H, Dh, K, mask_kind, mask_left, mask_right, mask, mask_sb,
                         mask_si, prec, p_drop, seed, ctx, ws, dx, g_qkv_w, g_o_w, g_ln_g, g_ln_b, g_r_emb, g_r_w_bias, g_r_bias, keep, out,
                         stream);
}

// ------------------------------------------------------------------ position-wise FFN (tt/transformer.py:54-58)
namespace {
struct FfnCtx {
    void *h, *a1;      // f32 (parity) or bf16 (fast)
    float *s2, *mean1, *rstd1, *mean2, *rstd2;
    bf16_t *w1T16 = nullptr, *w2T16 = nullptr;     // W1^T [d, Di], W2^T [Di, d] (fast): made in forward for the backward dgrads
    FfnCtx(Bump& b, long rows, int d, int Di, bool fast) {
        const size_t es = fast ? 2 : 4;
        h = b.take<char>(rows * d * es);
        a1 = b.take<char>(rows * Di * es);
        s2 = b.take<float>(rows * d);
        mean1 = b.take<float>(rows); rstd1 = b.take<float>(rows);
        mean2 = b.take<float>(rows); rstd2 = b.take<float>(rows);
        if (fast) {
            w1T16 = b.take<bf16_t>((size_t)Di * d);
            w2T16 = b.take<bf16_t>((size_t)Di * d);
        }
    }
};
struct FfnWs {
    float *f, *dres, *dh;
    void* da1;
    bf16_t *w1_16, *w2_16, *dres16;
    bf16_t *w1_16lo = nullptr, *w2_16lo = nullptr;     // second bf16 term of the two forward weights (option 13)
    FfnWs(Bump& b, long rows, int d, int Di, bool fast, void* keep = nullptr) {
        f = b.take<float>(rows * d);
        dres = b.take<float>(rows * d);
        dh = b.take<float>(rows * d);
        da1 = b.take<char>(rows * Di * (fast ? 2 : 4));
        w1_16 = w2_16 = dres16 = nullptr;
        if (fast) {
            w1_16 = b.take<bf16_t>((size_t)Di * d);
            w2_16 = b.take<bf16_t>((size_t)Di * d);
            w1_16lo = b.take<bf16_t>((size_t)Di * d);
            w2_16lo = b.take<bf16_t>((size_t)Di * d);
            dres16 = b.take<bf16_t>(rows * d);
            if (keep) {      // deferred weight gradients (ttmi_ffn_bwd_defer)
                da1 = keep;
                dres16 = static_cast<bf16_t*>(keep) + keep_al((size_t)rows * Di);
            }
        }
    }
};
}  // namespace

size_t ttmi_ffn_ctx_floats(long rows, int d, int Di, int prec) {
    Bump b(nullptr);
    FfnCtx c(b, rows, d, Di, ffn_fast(prec, d, Di));
    return b.floats();
}
static size_t ffn_x3_floats(long rows, int d, int Di) {
    size_t m = x3_nt_elems(rows, Di, d);
    const size_t c[] = {x3_nt_elems(rows, d, Di), x3_tn_elems(rows, d, Di), x3_tn_elems(rows, Di, d)};
    for (size_t v : c) m = v > m ? v : m;
    return (m + 1) / 2 + 128;
}
size_t ttmi_ffn_ws_floats(long rows, int d, int Di, int prec) {
    Bump b(nullptr);
    FfnWs w(b, rows, d, Di, ffn_fast(prec, d, Di));
    return b.floats() + (prec == 2 ? 64 + ffn_x3_floats(rows, d, Di) : 0);
}

// z = LN(y + W2 relu(W1 LN(y) + b1) + b2), the SAME (ln_g, ln_b) in both norms.
// pre_normed: h = LN(y), mean1, rstd1 are already in ctx (the producer of y formed them, LnPreNorm); z16_out (optional): bf16 copy of z
static int ffn_fwd_impl(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_g,
                 const float* ln_b, long rows, int d, int Di, int prec, float p_drop, float p_layer, unsigned seed, float* ctx,
                 float* ws, float* z, bool pre_normed, bf16_t* z16_out, void* stream) {
    TTMI_REQUIRE(y && w1 && b1 && w2 && b2 && ln_g && ln_b && ctx && ws && z, "ffn_fwd: null pointer");
    TTMI_REQUIRE(p_drop >= 0.f && p_drop < 1.f && p_layer >= 0.f && p_layer < 1.f, "ffn_fwd: dropout probability outside [0,1)");
    DropSpec d_in, d_out, d_layer;          // CoreNet.2, CoreNet.4 (tt/transformer.py:47,49) and the layer's own dropout (:196)
    d_in.p = p_drop; d_in.seed = seed ^ 0xB2u; d_in.salt = g_drop_salt;
    d_out.p = p_drop; d_out.seed = seed ^ 0xC3u; d_out.salt = g_drop_salt;
    d_layer.p = p_layer; d_layer.seed = seed ^ 0xD4u; d_layer.salt = g_drop_salt;
    TTMI_REQUIRE(rows > 0 && rows < (1L << 31) && d > 0 && Di > 0, "ffn_fwd: bad dims");
    TTMI_REQUIRE(((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(ws)) & 255) == 0, "ffn_fwd: ctx/ws must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fast = ffn_fast(prec, d, Di);
    Bump bc(ctx), bw(ws);
    FfnCtx c(bc, rows, d, Di, fast);
    FfnWs w(bw, rows, d, Di, fast);
    TTMI_REQUIRE(!pre_normed || fast, "ffn_fwd: a pre-normed input belongs to the bf16 pipeline");
    if (fast) {
        if (!pre_normed) CK(ln_fwd(y, nullptr, ln_g, ln_b, rows, d, 1e-5f, nullptr, nullptr, c.mean1, c.rstd1, st, static_cast<bf16_t*>(c.h)));
        Shadow s1, s2;
        const bf16_t *w1_16 = w.w1_16, *w2_16 = w.w2_16;
        const bool sh1 = shadow_of(w1, Di, d, Di, s1), sh2 = shadow_of(w2, d, Di, d, s2);
        ctx_record(ctx, 0, sh1);
        ctx_record(ctx, 1, sh2);
        if (sh1) w1_16 = s1.w16;
        else CK(transpose_convert_bf16(w1, Di, d, c.w1T16, Di, st, w.w1_16));                   // W1 (bf16) and W1^T [d, Di]
        if (sh2) w2_16 = s2.w16;
        else CK(transpose_convert_bf16(w2, d, Di, c.w2T16, d, st, w.w2_16));                    // W2 (bf16) and W2^T [Di, d]
        NtEpilogue e1, e2;
        e1.bias = b1; e1.relu = 1; e1.drop = d_in;
        e2.bias = b2;
        if (split_w(rows, false)) {
            if (sh1 && s1.w16lo) e1.B_lo = s1.w16lo;
            else { CK(bf16_residual(w1, w1_16, w.w1_16lo, (long)d * Di, st)); e1.B_lo = w.w1_16lo; }
        }
        if (split_w(rows, true)) {
            if (sh2 && s2.w16lo) e2.B_lo = s2.w16lo;
            else { CK(bf16_residual(w2, w2_16, w.w2_16lo, (long)d * Di, st)); e2.B_lo = w.w2_16lo; }
        }
        CK(gemm_nt_bf16(static_cast<bf16_t*>(c.h), w1_16, c.a1, 1, e1, (int)rows, Di, d, d, d, Di, st));
        CK(gemm_nt_bf16(static_cast<bf16_t*>(c.a1), w2_16, w.f, 0, e2, (int)rows, d, Di, Di, Di, d, st));
    } else {
        float* h = static_cast<float*>(c.h);
        float* a1 = static_cast<float*>(c.a1);
        CK(ln_fwd(y, nullptr, ln_g, ln_b, rows, d, 1e-5f, nullptr, h, c.mean1, c.rstd1, st));
        bf16_t* x3 = prec == 2 ? reinterpret_cast<bf16_t*>(ws + ((bw.floats() + 63) & ~(size_t)63)) : nullptr;
        if (x3 && x3_worth(rows, Di, d)) {
            NtEpilogue e1, e2;
            e1.bias = b1; e1.relu = 1; e1.drop = d_in;
            e2.bias = b2;
            CK(x3_nt(h, w1, a1, (int)rows, Di, d, d, d, Di, e1, x3, st));
            CK(x3_nt(a1, w2, w.f, (int)rows, d, Di, Di, Di, d, e2, x3, st));
        } else {
        GemmDesc g = mk(h, w1, a1, (int)rows, Di, d, d, d, Di, NT_ | GEMM_BIAS | GEMM_RELU, prec);
        g.bias = b1; g.drop = d_in;
        CK(ttmi_launch_gemm(g, st));
        GemmDesc g2 = mk(a1, w2, w.f, (int)rows, d, Di, Di, Di, d, NT_ | GEMM_BIAS, prec);
        g2.bias = b2;
        CK(ttmi_launch_gemm(g2, st));
        }
    }
    CK(ln_fwd(y, w.f, ln_g, ln_b, rows, d, 1e-5f, c.s2, z, c.mean2, c.rstd2, st, z16_out, d_out, d_layer));
    return TTMI_OK;
}

int ttmi_ffn_fwd(const float* y, const float* w1, const float* b1, const float* w2, const float* b2, const float* ln_g,
                 const float* ln_b, long rows, int d, int Di, int prec, float p_drop, float p_layer, unsigned seed, float* ctx,
                 float* ws, float* z, void* stream) {
    return ffn_fwd_impl(y, w1, b1, w2, b2, ln_g, ln_b, rows, d, Di, prec, p_drop, p_layer, seed, ctx, ws, z, false, nullptr, stream);
}

static int ffn_bwd_impl(const float* dz, const float* y, const float* w1, const float* w2, const float* ln_g, long rows, int d, int Di,
                 int prec, float p_drop, float p_layer, unsigned seed, const float* ctx, float* ws, float* dy, float* g_w1,
                 float* g_b1, float* g_w2, float* g_b2, float* g_ln_g, float* g_ln_b, void* keep, ttmi_wgrad_desc* out, void* stream,
                 bool skip_pre_norm = false) {
    TTMI_REQUIRE(dz && y && w1 && w2 && ln_g && ctx && ws && (dy || skip_pre_norm) && g_w1 && g_b1 && g_w2 && g_b2 && g_ln_g && g_ln_b,
                 "ffn_bwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fast = ffn_fast(prec, d, Di);
    TTMI_REQUIRE(!out || (fast && keep && aligned16(keep)), "ffn_bwd_defer: bf16 pipeline and a 16-byte aligned keep buffer required");
    Bump bc(const_cast<float*>(ctx)), bw(ws);
    FfnCtx c(bc, rows, d, Di, fast);
    FfnWs w(bw, rows, d, Di, fast, keep);
    DropSpec d_out, d_layer;
    d_out.p = p_drop; d_out.seed = seed ^ 0xC3u; d_out.salt = g_drop_salt;
    d_layer.p = p_layer; d_layer.seed = seed ^ 0xD4u; d_layer.salt = g_drop_salt;
    const float inv_keep = 1.f / (1.f - p_drop);
    // df = dres * mask(CoreNet.4); dres itself remains the residual-branch gradient.  bf16 pipeline: df (bf16) and g_b2 = its
    // column sums come out of the LayerNorm-backward pass itself
    CK(ln_bwd(dz, c.s2, c.mean2, c.rstd2, ln_g, nullptr, rows, d, w.dres, g_ln_g, g_ln_b, st, d_layer, fast ? w.dres16 : nullptr, d_out,
              fast ? g_b2 : nullptr));
    const float* df = w.dres;
    if (!fast) {
        if (p_drop > 0.f) {
            CK(dropout_apply(w.dres, rows * d, d_out, w.f, nullptr, st));
            df = w.f;
        }
        CK(colsum(df, d, rows, d, 1, 1, 0, 0, 0, 0, g_b2, st));
    }
    if (fast) {
        bf16_t* a1 = static_cast<bf16_t*>(c.a1);
        bf16_t* h = static_cast<bf16_t*>(c.h);
        bf16_t* da1 = static_cast<bf16_t*>(w.da1);
        int rc = TTMI_OK;
        const bf16_t* w1T16 = ctx_weightT(ctx, 0, w1, Di, d, Di, c.w1T16, st, &rc);
        CK(rc);
        const bf16_t* w2T16 = ctx_weightT(ctx, 1, w2, d, Di, d, c.w2T16, st, &rc);
        CK(rc);
        ctx_forget(ctx);
        if (out) out[0] = ttmi_wgrad_desc{w.dres16, a1, g_w2, nullptr, d, Di, (int)rows, (long)d, (long)Di, (long)Di};
        else CK(gemm_tn_bf16(w.dres16, a1, g_w2, d, Di, (int)rows, d, Di, Di, 1, fork_stream(st)));
        NtEpilogue e;
        e.mask = a1; e.scale = inv_keep;                   // a1 is stored post-dropout: a1 > 0 <=> ReLU active AND kept
        CK(gemm_nt_bf16(w.dres16, w2T16, da1, 1, e, (int)rows, Di, d, d, d, Di, st));
        if (out) out[1] = ttmi_wgrad_desc{da1, h, g_w1, g_b1, Di, d, (int)rows, (long)Di, (long)d, (long)d};
        else CK(gemm_tn_bf16(da1, h, g_w1, Di, d, (int)rows, Di, d, d, 1, fork_stream(st), g_b1));   // g_b1 = column sums of da1, fused
        CK(gemm_nt_bf16(da1, w1T16, w.dh, 0, nullptr, (int)rows, d, Di, Di, Di, d, st));
    } else {
        const float* a1 = static_cast<const float*>(c.a1);
        const float* h = static_cast<const float*>(c.h);
        float* da1 = static_cast<float*>(w.da1);
        bf16_t* x3 = prec == 2 ? reinterpret_cast<bf16_t*>(ws + ((bw.floats() + 63) & ~(size_t)63)) : nullptr;
        if (x3 && x3_worth(rows, Di, d)) {
            CK(x3_tn(df, a1, g_w2, d, Di, rows, d, Di, Di, x3, st));
            CK(x3_nn(df, w2, da1, (int)rows, Di, d, d, Di, Di, NtEpilogue(), x3, st, true));
            CK(relu_mask_scale(da1, a1, rows * Di, inv_keep, st));           // the ReLU' (and dropout) mask of the exact-f32 path's epilogue: da1 = a1 > 0 ? da1 / keep : 0
            CK(colsum(da1, Di, rows, Di, 1, 1, 0, 0, 0, 0, g_b1, st));
            CK(x3_tn(da1, h, g_w1, Di, d, rows, Di, d, d, x3, st));
            CK(x3_nn(da1, w1, w.dh, (int)rows, d, Di, Di, d, d, NtEpilogue(), x3, st, true));
        } else {
        CK(wgrad(df, a1, g_w2, d, Di, (int)rows, d, Di, Di, prec, st));
        GemmDesc g = mk(df, w2, da1, (int)rows, Di, d, d, Di, Di, NN_ | GEMM_MASK_AUX, prec);
        g.aux = a1; g.alpha = inv_keep;
        CK(ttmi_launch_gemm(g, st));
        CK(colsum(da1, Di, rows, Di, 1, 1, 0, 0, 0, 0, g_b1, st));
        CK(wgrad(da1, h, g_w1, Di, d, (int)rows, Di, d, d, prec, st));
        CK(ttmi_launch_gemm(mk(da1, w1, w.dh, (int)rows, d, Di, Di, d, d, NN_, prec), st));
        }
    }
    // (skip_pre_norm: the layer-level call - w.dh and w.dres stay in the workspace for the attention sub-layer's paired LayerNorm backward)
    if (!skip_pre_norm) CK(ln_bwd(w.dh, y, c.mean1, c.rstd1, ln_g, w.dres, rows, d, dy, g_ln_g, g_ln_b, st));
    join_stream(st);
    return TTMI_OK;
}

int ttmi_ffn_bwd(const float* dz, const float* y, const float* w1, const float* w2, const float* ln_g, long rows, int d, int Di,
                 int prec, float p_drop, float p_layer, unsigned seed, const float* ctx, float* ws, float* dy, float* g_w1,
                 float* g_b1, float* g_w2, float* g_b2, float* g_ln_g, float* g_ln_b, void* stream) {
    return ffn_bwd_impl(dz, y, w1, w2, ln_g, rows, d, Di, prec, p_drop, p_layer, seed, ctx, ws, dy, g_w1, g_b1, g_w2, g_b2, g_ln_g, g_ln_b,
                        nullptr, nullptr, stream);
}
size_t ttmi_ffn_bwd_keep_bytes(long rows, int d, int Di) { return 2 * (keep_al((size_t)rows * Di) + keep_al((size_t)rows * d)); }
int ttmi_ffn_bwd_defer(const float* dz, const float* y, const float* w1, const float* w2, const float* ln_g, long rows, int d, int Di,
                       int prec, float p_drop, float p_layer, unsigned seed, const float* ctx, float* ws, float* dy, float* g_w1,
                       float* g_b1, float* g_w2, float* g_b2, float* g_ln_g, float* g_ln_b, void* keep, ttmi_wgrad_desc* out,
                       void* stream) {
    TTMI_REQUIRE(keep && out, "ffn_bwd_defer: keep buffer and descriptor array required");
    return ffn_bwd_impl(dz, y, w1, w2, ln_g, rows, d, Di, prec, p_drop, p_layer, seed, ctx, ws, dy, g_w1, g_b1, g_w2, g_b2, g_ln_g, g_ln_b,
                        keep, out, stream);
}
// the deferred forms exist on the bf16 pipeline of both sub-layers
int ttmi_wgrad_defer_supported(long rows, int d, int H, int Dh, int Di, int prec) {
    return rows > 0 && attn_fast(prec, d, H, Dh) && ffn_fast(prec, d, Di) ? 1 : 0;
}

// ------------------------------------------------------------------ one encoder layer per call (RelLearnableDecoderLayer.forward, tt/transformer.py:188-197)
// z = FFN(ATTN(x)).  Besides halving the host's calls, the layer-level entry removes three passes over the residual stream that the
// sub-layer boundary forced: the FFN's pre-norm is formed by the attention sub-layer's post-norm pass (y is not read back), the two LayerNorm
// backward passes that meet at y run as one kernel (dy is never stored), and the bf16 copy of a layer's input comes from the pass that
// produced it (x16_in / z16_out) instead of a conversion launch.  Same arithmetic as the two sub-layer calls, in the same order per element.
static size_t ws_round(size_t floats) { return (floats + 63) & ~size_t(63); }
int ttmi_layer_fused(int d, int H, int Dh, int Di, int prec) {
    return attn_fast(prec, d, H, Dh) && ffn_fast(prec, d, Di) && ln_bwd_pair_supported(d) ? 1 : 0;
}
size_t ttmi_layer_ws_floats(int B, int L, int d, int H, int Dh, int Di, int prec) {
    return ws_round(ttmi_attn_ws_floats(B, L, d, H, Dh, prec)) + ws_round(ttmi_ffn_ws_floats((long)B * L, d, Di, prec)) + ws_round((size_t)B * L * d) + 64;
}

int ttmi_layer_fwd(const float* x, const void* x16_in, const float* qkv_w, const float* o_w, const float* ln_g, const float* ln_b, const float* r_emb,
                   const float* r_w_bias, const float* r_bias, const float* w1, const float* b1, const float* w2, const float* b2,
                   const float* ff_ln_g, const float* ff_ln_b, int B, int L, int d, int H, int Dh, int K, int Di, int mask_kind, int mask_left,
                   int mask_right, const unsigned char* mask, long mask_sb, long mask_si, int prec, float p_drop_attn, unsigned seed_attn,
                   float p_drop_ffn, float p_layer, unsigned seed_ffn, float* ctx_attn, float* ctx_ffn, float* ws, float* y, float* z,
                   void* z16_out, void* stream) {
    TTMI_REQUIRE(ctx_attn && ctx_ffn && ws && y && z && ff_ln_g && ff_ln_b, "layer_fwd: null pointer");
    TTMI_REQUIRE(B > 0 && L > 0 && d > 0 && H > 0 && Dh > 0 && Di > 0, "layer_fwd: bad dims");
    TTMI_REQUIRE((reinterpret_cast<uintptr_t>(ctx_ffn) & 255) == 0, "layer_fwd: ctx_ffn must be 256-byte aligned");
    const bool fuse = ttmi_layer_fused(d, H, Dh, Di, prec) != 0;
    TTMI_REQUIRE(fuse || (!x16_in && !z16_out), "layer_fwd: bf16 copies of the residual stream exist in the fused bf16 pipeline only (ttmi_layer_fused)");
    const long rows = (long)B * L;
    float* ws_ffn = ws + ws_round(ttmi_attn_ws_floats(B, L, d, H, Dh, prec));
    LnPreNorm pre;
    if (fuse) {
        Bump bc(ctx_ffn);
        FfnCtx fc(bc, rows, d, Di, true);
        pre.g = ff_ln_g; pre.b = ff_ln_b; pre.h16 = static_cast<bf16_t*>(fc.h); pre.mean = fc.mean1; pre.rstd = fc.rstd1;
    }
    CK(attn_fwd_impl(x, qkv_w, o_w, ln_g, ln_b, r_emb, r_w_bias, r_bias, B, L, d, H, Dh, K, mask_kind, mask_left, mask_right, mask, mask_sb, mask_si,
                     prec, p_drop_attn, seed_attn, ctx_attn, ws, y, static_cast<const bf16_t*>(x16_in), fuse ? &pre : nullptr, stream));
    return ffn_fwd_impl(y, w1, b1, w2, b2, ff_ln_g, ff_ln_b, rows, d, Di, prec, p_drop_ffn, p_layer, seed_ffn, ctx_ffn, ws_ffn, z, fuse,
                        static_cast<bf16_t*>(z16_out), stream);
}

// Backward of ttmi_layer_fwd: dx is written, every g_* buffer is accumulated into.  keep_attn / keep_ffn / out (all or none; bf16 pipeline):
// the four weight-gradient GEMMs are described in out[0..3] (CoreNet.3, CoreNet.0 + bias, o_net, qkv_net) instead of launched, their
// operands kept in the two keep buffers (ttmi_ffn_bwd_keep_bytes / ttmi_attn_bwd_keep_bytes) - and in x16_in, where given.
int ttmi_layer_bwd(const float* dz, const float* x, const void* x16_in, const float* y, const float* qkv_w, const float* o_w, const float* ln_g,
                   const float* r_emb, const float* r_w_bias, const float* r_bias, const float* w1, const float* w2, const float* ff_ln_g, int B,
                   int L, int d, int H, int Dh, int K, int Di, int mask_kind, int mask_left, int mask_right, const unsigned char* mask,
                   long mask_sb, long mask_si, int prec, float p_drop_attn, unsigned seed_attn, float p_drop_ffn, float p_layer,
                   unsigned seed_ffn, const float* ctx_attn, const float* ctx_ffn, float* ws, float* dx, float* g_qkv_w, float* g_o_w,
                   float* g_ln_g, float* g_ln_b, float* g_r_emb, float* g_r_w_bias, float* g_r_bias, float* g_w1, float* g_b1, float* g_w2,
                   float* g_b2, float* g_ff_ln_g, float* g_ff_ln_b, void* keep_attn, void* keep_ffn, ttmi_wgrad_desc* out, void* stream) {
    TTMI_REQUIRE(dz && x && y && ctx_attn && ctx_ffn && ws && dx, "layer_bwd: null pointer");
    TTMI_REQUIRE(B > 0 && L > 0 && d > 0 && H > 0 && Dh > 0 && Di > 0, "layer_bwd: bad dims");
    TTMI_REQUIRE((!out && !keep_attn && !keep_ffn) || (out && keep_attn && keep_ffn), "layer_bwd: keep buffers and descriptors come together");
    const bool fuse = ttmi_layer_fused(d, H, Dh, Di, prec) != 0;
    TTMI_REQUIRE(fuse || !x16_in, "layer_bwd: a bf16 copy of x exists in the fused bf16 pipeline only");
    const long rows = (long)B * L;
    float* ws_ffn = ws + ws_round(ttmi_attn_ws_floats(B, L, d, H, Dh, prec));
    float* dy = ws_ffn + ws_round(ttmi_ffn_ws_floats(rows, d, Di, prec));
    CK(ffn_bwd_impl(dz, y, w1, w2, ff_ln_g, rows, d, Di, prec, p_drop_ffn, p_layer, seed_ffn, ctx_ffn, ws_ffn, fuse ? nullptr : dy, g_w1, g_b1, g_w2,
                    g_b2, g_ff_ln_g, g_ff_ln_b, keep_ffn, out, stream, fuse));
    LnPairIn pair;
    if (fuse) {
        Bump bc(const_cast<float*>(ctx_ffn)), bw(ws_ffn);
        FfnCtx fc(bc, rows, d, Di, true);
        FfnWs fw(bw, rows, d, Di, true, keep_ffn);
        pair = LnPairIn{fw.dh, y, fc.mean1, fc.rstd1, ff_ln_g, fw.dres, g_ff_ln_g, g_ff_ln_b};
    }
    return attn_bwd_impl(fuse ? nullptr : dy, x, qkv_w, o_w, ln_g, r_emb, r_w_bias, r_bias, B, L, d, H, Dh, K, mask_kind, mask_left, mask_right, mask,
                         mask_sb, mask_si, prec, p_drop_attn, seed_attn, ctx_attn, ws, dx, g_qkv_w, g_o_w, g_ln_g, g_ln_b, g_r_emb, g_r_w_bias,
                         g_r_bias, keep_attn, out ? out + 2 : nullptr, stream, static_cast<const bf16_t*>(x16_in), fuse ? &pair : nullptr);
}

// ------------------------------------------------------------------ joint network (tt/model.py:20-39)
// z[b,t,u,:] = Wp tanh(We enc[b,t] + Wd dec[b,u] + bf) + bp, forward_layer.weight = [We | Wd] ([J, de+dd]).
// The reference repeats enc/dec to [B,T,U1,de+dd] and concatenates; the split-weight form is the same
// arithmetic without the 3 x [B,T,U1,*] temporaries.
//
// prec 0: everything f32 (exact-f32 MFMA).  prec 1 with J % 8 == 0 ("fast"): H, logits and dlogits are bf16 in HBM,
// the vocabulary projection / its dgrad / its wgrad run on the glds-staged kernels of gemm_fast.hip; logits rows
// have pitch ldv >= V (callers use a multiple of 64) and dlogits must be zero in columns [V, ldv).
static inline bool joint_fast(int prec, int J) { return prec == 1 && J % 8 == 0; }
static inline size_t al8(size_t n) { return (n + 7) & ~(size_t)7; }
// forward_layer (the 2d -> J input layer) on the throughput kernels: training-sized batches whose bf16 operand copies fit in the
// free part of the first workspace region (see ttmi_joint_fwd / _bwd); everything else keeps the generic kernel
static inline bool joint_input_fast(int prec, int B, int T, int U1, int de, int dd, int J) {
    if (!joint_fast(prec, J) || de % 8 || dd % 8 || (long)B * T < 1024) return false;
    const size_t M = (size_t)B * T * U1, din = (size_t)de + dd;
    const size_t need = al8((size_t)B * T * de) + al8((size_t)B * U1 * dd) + al8((size_t)B * T * J) + 2 * al8((size_t)B * U1 * J) + al8(din * J) + 8;   // (2: dPD's two terms)
    // bf16 elements behind dH16 (backward); the forward pass has the whole region (2 M J elements) for its copies and the weight's second split term
    return need <= (size_t)M * J && need + al8(din * J) + al8((size_t)B * U1 * dd) <= 2 * (size_t)M * J;      // (+ the label states' second split term, round 6)
}
// dtype of the logits this configuration produces / expects: 0 = f32, 1 = bf16
int ttmi_joint_logits_dtype(int prec, int J) { return joint_fast(prec, J) ? 1 : 0; }

// (room for the lattice rows padded to 64: the exp-domain wgrad reduces over whole 64-row K-tiles, ttmi_joint_exp_padded_rows)
size_t ttmi_joint_ctx_floats(int B, int T, int U1, int J) { return al4(((size_t)B * T * U1 + 63) / 64 * 64 * J); }
// prec 2 keeps H as two bf16 blocks per row [hi | lo] (columns padded to 64) instead of one f32: the same size unless J is no multiple of 64
size_t ttmi_joint_ctx_floats_prec(int B, int T, int U1, int J, int prec) {
    const size_t f = ttmi_joint_ctx_floats(B, T, U1, J);
    if (prec != 2) return f;
    const size_t x = al4((size_t)B * T * U1 * x3_pad(J) + 64);
    return x > f ? x : f;
}
// does the joint of this size run its bf16x3 form with H in three-block rows (prec 2 only)?
static bool joint_x3_h3(long M, int V, int J) { return x3_worth(M, V, J) && J % 4 == 0; }
size_t ttmi_joint_ws_floats(int B, int T, int U1, int J, int V) {
    return al4((size_t)B * T * U1 * J) + 2 * al4((size_t)B * T * J) + 2 * al4((size_t)B * U1 * J) +
           al4((size_t)J * (((size_t)V + 63) / 64 * 64));
}

// scratch of the bf16x3 products behind the ordinary joint workspace (prec 2): the backward's two big ones use it one after the other
static size_t joint_x3_floats(int B, int T, int U1, int J, int V) {
    const long M = (long)B * T * U1;
    const size_t fwd = x3_nt_elems(M, V, J);
    const size_t bwd = x3_al((size_t)M * 3 * x3_pad(V)) + x3_al((size_t)M * 2 * x3_pad(J)) + x3_al((size_t)J * 3 * x3_pad(V));   // Z3 + H2 + WT3 (joint_bwd_impl)
    return ((fwd > bwd ? fwd : bwd) + 1) / 2 + 64;
}
size_t ttmi_joint_ws_floats_prec(int B, int T, int U1, int J, int V, int prec) {
    return ttmi_joint_ws_floats(B, T, U1, J, V) + (prec == 2 ? 64 + joint_x3_floats(B, T, U1, J, V) : 0);
}

// rowsum != nullptr: the "exp store" form of the fused joint + loss fast path (logits become exp(z - *shift), see ttmi_joint_fwd_exp)
static int joint_fwd_impl(const float* enc, const float* dec, const float* wf, const float* bf, const float* wp, const float* bp,
                          int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx, float* ws, void* logits, long ldv,
                          float* rowsum, int nparts, const float* shift, const int* labels, int blank, float* emis, void* stream) {
    TTMI_REQUIRE(enc && dec && wf && bf && wp && bp && ctx && ws && logits, "joint_fwd: null pointer");
    TTMI_REQUIRE(B > 0 && T > 0 && U1 > 0 && de > 0 && dd > 0 && J > 0 && V > 0 && ldv >= V, "joint_fwd: bad dims");
    TTMI_REQUIRE((long)B * T * U1 < (1L << 31), "joint_fwd: B*T*(U+1) too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fast = joint_fast(prec, J);
    float* PE = ws + al4((size_t)B * T * U1 * J);
    float* PD = PE + al4((size_t)B * T * J);
    const int din = de + dd;
    const int M = B * T * U1;
    if (joint_input_fast(prec, B, T, U1, de, dd, J)) {
        // forward_layer on the throughput kernels: bf16 copies of the encoder states and of [We | Wd] live in the first workspace
        // region (dH's, unused in the forward pass); same operand rounding as the generic kernel's convert-while-staging
        bf16_t* enc16 = reinterpret_cast<bf16_t*>(ws);
        bf16_t* dec16 = enc16 + al8((size_t)B * T * de);
        bf16_t* Wf16 = dec16 + al8((size_t)B * U1 * dd);
        CK(convert_bf16(enc, enc16, (long)B * T * de, st));
        CK(convert_bf16(dec, dec16, (long)B * U1 * dd, st));
        Shadow sh;
        const bf16_t* wf16 = Wf16;
        const bool shf = shadow_of(wf, J, din, J, sh);
        if (shf) wf16 = sh.w16;
        else CK(convert_bf16(wf, Wf16, (long)J * din, st));
        CK(gemm_nt_bf16(enc16, wf16, PE, 0, nullptr, B * T, J, de, de, din, J, st));
        CK(gemm_nt_bf16(dec16, wf16 + de, PD, 0, nullptr, B * U1, J, dd, dd, din, J, st));
        // second term of the weight's bf16 split (w ~ w16 + lo16): the rounding of [We | Wd] is ONE pattern applied to every frame and label, so
        // its effect on the loss does not average out over a batch; two small GEMMs more (1 % of the projection's work) take it out
        bf16_t* Wf16lo_ws = Wf16 + al8((size_t)J * din);
        const bf16_t* Wf16lo = Wf16lo_ws;
        if (shf && sh.w16lo) Wf16lo = sh.w16lo;
        else CK(bf16_residual(wf, wf16, Wf16lo_ws, (long)J * din, st));
        NtEpilogue ea, eb;
        ea.addend = PE;
        eb.addend = PD;
        CK(gemm_nt_bf16(enc16, Wf16lo, PE, 0, ea, B * T, J, de, de, din, J, st));
        CK(gemm_nt_bf16(dec16, Wf16lo + de, PD, 0, eb, B * U1, J, dd, dd, din, J, st));
        // second term of the LABEL STATES' bf16 split (round 6): dec[b, u, :] meets all T frames of its utterance, so the rounding of one label state is ONE pattern in T
        // lattice rows - and, while the label encoder's outputs still resemble each other (the first steps of training), in every utterance of the batch.  Measured
        // (tools/debug/joint_weight_rounding.py, C2 after 5 SGD steps, fp32 encoders): the exp-domain joint + loss err by +0.035 ... +0.14 nats of 827 on EVERY utterance
        // (4e-5 ... 1.7e-4, all of the joint's share of the batch-mean loss error), by 0.002 with the label states pre-rounded; the audio states' rounding (one pattern per frame,
        // 500 different ones along an alignment) and the rounding of every weight of the joint change nothing.  One 1632-row GEMM more.
        if (g_joint_dec_lo) {
            bf16_t* dec16lo = Wf16lo_ws + al8((size_t)J * din);
            CK(bf16_residual(dec, dec16, dec16lo, (long)B * U1 * dd, st));
            CK(gemm_nt_bf16(dec16lo, wf16 + de, PD, 0, eb, B * U1, J, dd, dd, din, J, st));
        }
    } else {
        CK(ttmi_launch_gemm(mk(enc, wf, PE, B * T, J, de, de, din, J, NT_, prec), st));
        CK(ttmi_launch_gemm(mk(dec, wf + de, PD, B * U1, J, dd, dd, din, J, NT_, prec), st));
        if (prec == 1 && (size_t)M >= (size_t)din) {
            // the same second split term on the generic kernel (it rounds f32 operands to bf16 while staging): the residual of the weight's
            // rounding in f32, in the first workspace region (dH's, free in the forward pass: M J >= din J floats), accumulated with beta = 1
            float* R = ws;
            CK(bf16_residual_f32(wf, R, (long)J * din, st));
            GemmDesc g1 = mk(enc, R, PE, B * T, J, de, de, din, J, NT_, prec), g2 = mk(dec, R + de, PD, B * U1, J, dd, dd, din, J, NT_, prec);
            g1.beta = 1.f;
            g2.beta = 1.f;
            CK(ttmi_launch_gemm(g1, st));
            CK(ttmi_launch_gemm(g2, st));
            // ... and the label states' second term (see above; the same numbers as the throughput path: the residual in f32, rounded to bf16 while staging)
            if (g_joint_dec_lo && (size_t)M * J >= al4((size_t)din * J) + (size_t)B * U1 * dd) {
                float* Rd = R + al4((size_t)din * J);
                CK(bf16_residual_f32(dec, Rd, (long)B * U1 * dd, st));
                GemmDesc g3 = mk(Rd, wf + de, PD, B * U1, J, dd, dd, din, J, NT_, prec);
                g3.beta = 1.f;
                CK(ttmi_launch_gemm(g3, st));
            }
        }
    }
    if (!fast && prec == 2 && joint_x3_h3(M, V, J)) {
        // bf16x3: the projection in three bf16 terms on the throughput kernel.  H leaves the tanh kernel already split - ctx holds the rows
        // [hi | lo] (ttmi_joint_ctx_floats_prec) the GEMM reads, its hi block a second time against the weight's lo block - and only the
        // weight is split here ([hi | hi | lo], behind the workspace)
        const int Jp = x3_pad(J);
        bf16_t* H2 = reinterpret_cast<bf16_t*>(ctx);
        bf16_t* W3 = reinterpret_cast<bf16_t*>(ws + ((ttmi_joint_ws_floats(B, T, U1, J, V) + 63) & ~(size_t)63));
        CK(joint_tanh_fwd_x3(PE, PD, bf, B, T, U1, J, Jp, H2, st));
        CK(split3_bf16(wp, J, V, J, Jp, 1, W3, st));
        NtEpilogue e3;
        e3.bias = bp;
        ttmi_probe_begin(0, st);
        const int rc = x3_launch(H2, W3, static_cast<float*>(logits), M, V, Jp, ldv, e3, true, st);
        ttmi_probe_end(0, st);
        CK(rc);
        return TTMI_OK;
    }
    if (!fast) {
        float* Hh = ctx;
        CK(joint_tanh_fwd(PE, PD, bf, B, T, U1, J, Hh, 0, st));
        GemmDesc g = mk(Hh, wp, static_cast<float*>(logits), M, V, J, J, J, ldv, NT_ | GEMM_BIAS, prec);
        g.bias = bp;
        ttmi_probe_begin(0, st);
        int rc;
        if (prec == 2 && x3_worth(M, V, J)) {       // (J % 4 != 0: f32 H, split by a pass of its own)
            NtEpilogue e3;
            e3.bias = bp;
            bf16_t* x3 = reinterpret_cast<bf16_t*>(ws + ((ttmi_joint_ws_floats(B, T, U1, J, V) + 63) & ~(size_t)63));
            rc = x3_nt(Hh, wp, static_cast<float*>(logits), M, V, J, J, J, ldv, e3, x3, st);
        } else rc = ttmi_launch_gemm(g, st);
        ttmi_probe_end(0, st);
        CK(rc);
        return TTMI_OK;
    }
    TTMI_REQUIRE(ldv % 8 == 0 && aligned16(logits), "joint_fwd: bf16 logits need a 16-byte aligned base and pitch %% 8 == 0");
    bf16_t* H16 = reinterpret_cast<bf16_t*>(ctx);
    bf16_t* Wp16 = reinterpret_cast<bf16_t*>(PD + 2 * al4((size_t)B * U1 * J));
    Shadow shp;
    const bf16_t* wp16 = Wp16;
    if (shadow_of(wp, V, J, (V + 63) / 64 * 64, shp)) wp16 = shp.w16;
    else CK(convert_bf16(wp, Wp16, (long)V * J, st));
    // exp-domain form with `emis`: the blank / label logits of every lattice row also leave in f32 - from the same bf16 operands the GEMM reads
    // (what P and its row sums contain) and from f32 operands (what the loss makes the emission log-probs of)
    if (emis) CK(joint_tanh_fwd_emis(PE, PD, bf, B, T, U1, J, H16, wp16, wp, bp, labels, V, blank, emis, st));
    else CK(joint_tanh_fwd(PE, PD, bf, B, T, U1, J, H16, 1, st));
    NtEpilogue e;
    e.bias = bp; e.rowsum = rowsum; e.nparts = nparts; e.exp_shift = shift;
    ttmi_probe_begin(0, st);
    const int rc = gemm_nt_bf16(H16, wp16, logits, 1, e, M, V, J, J, J, ldv, st);
    ttmi_probe_end(0, st);
    CK(rc);
    return TTMI_OK;
}

int ttmi_joint_fwd(const float* enc, const float* dec, const float* wf, const float* bf, const float* wp, const float* bp,
                   int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx, float* ws, void* logits, long ldv,
                   void* stream) {
    return joint_fwd_impl(enc, dec, wf, bf, wp, bp, B, T, U1, de, dd, J, V, prec, ctx, ws, logits, ldv, nullptr, 0, nullptr, nullptr, 0, nullptr,
                          stream);
}

// ---- exp-domain forms: the fused joint + loss fast path (training-sized bf16 problems; ask ttmi_joint_exp_supported first)
// forward: P[row, v] = bf16(exp(z[row, v] - *shift)) with pitch ldv (columns [V, ldv) zero), rowsum f32 [nparts, rows] = partial row
// sums of the unrounded values (nparts = ttmi_joint_exp_nparts(V)); shift: device scalar, nullable = 0.  The loss then needs two
// entries per row instead of two passes over the lattice (ttmi_rnnt_loss_fwd_exp / _bwd_exp), and the backward below takes
// d logits = srow[r] * P[r, :] without ever materialising it: the row factor rides in the dgrad epilogue and in the wgrad's H operand.
int ttmi_joint_exp_supported(int B, int T, int U1, int J, int V, int prec, long ldv) {
    if (!joint_fast(prec, J) || (long)B * T * U1 >= (1L << 31) || ldv % 64 != 0) return 0;
    return gemm_fast_joint_exp_ok(B * T * U1, V, J, ldv) ? 1 : 0;
}
// forward + loss only (no gradients wanted): the backward GEMMs' size conditions do not apply
int ttmi_joint_exp_fwd_supported(int B, int T, int U1, int J, int V, int prec, long ldv) {
    if (!joint_fast(prec, J) || (long)B * T * U1 >= (1L << 31) || ldv % 64 != 0) return 0;
    return gemm_fast_joint_exp_ok(B * T * U1, V, J, ldv, true) ? 1 : 0;
}
int ttmi_joint_exp_nparts(int V) { return 4 * ((V + 255) / 256); }
// rows the caller's P and srow16 buffers must have room for: the lattice rows rounded up to the wgrad's 64-row reduction tile
long ttmi_joint_exp_padded_rows(int B, int T, int U1) { return ((long)B * T * U1 + 63) / 64 * 64; }

int ttmi_joint_fwd_exp(const float* enc, const float* dec, const float* wf, const float* bf, const float* wp, const float* bp,
                       int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx, float* ws, void* P, long ldv,
                       float* rowsum, int nparts, const float* shift, const int* labels, int blank, float* emis, void* stream) {
    TTMI_REQUIRE(rowsum && nparts >= ttmi_joint_exp_nparts(V), "joint_fwd_exp: rowsum needs >= %d parts per row", ttmi_joint_exp_nparts(V));
    TTMI_REQUIRE(!emis || ((labels || U1 == 1) && blank >= 0 && blank < V), "joint_fwd_exp: emis needs the labels and a blank index inside [0, V)");
    TTMI_REQUIRE(ttmi_joint_exp_fwd_supported(B, T, U1, J, V, prec, ldv), "joint_fwd_exp: size / precision outside the fast path (B=%d T=%d U1=%d J=%d V=%d)", B, T, U1, J, V);
    return joint_fwd_impl(enc, dec, wf, bf, wp, bp, B, T, U1, de, dd, J, V, prec, ctx, ws, P, ldv, rowsum, nparts, shift, labels, blank, emis,
                          stream);
}

// srow != nullptr: exp-domain form, d logits[r, :] = srow[r] * dlogits[r, :] (see ttmi_joint_bwd_exp); ctx is then scaled in place
static int joint_bwd_impl(const void* dlogits, long ldg, const float* enc, const float* dec, const float* wf, const float* wp, int B,
                          int T, int U1, int de, int dd, int J, int V, int prec, float* ctx, float* ws, float* denc, float* ddec,
                          float* g_wf, float* g_bf, float* g_wp, float* g_bp, const float* srow, const void* srow16, void* stream, bool presplit = false) {
    TTMI_REQUIRE(dlogits && enc && dec && wf && wp && ctx && ws && denc && ddec && g_wf && g_bf && g_wp && g_bp,
                 "joint_bwd: null pointer");
    TTMI_REQUIRE(ldg >= V, "joint_bwd: bad pitch");
    // presplit (bf16x3, round 6): d logits arrive as the two bf16 planes [hi | lo] per row (pitch 2 ldg bf16 in the bytes of ldg f32: ttmi_rnnt_loss_bwd_split)
    TTMI_REQUIRE(!presplit || (prec == 2 && x3_worth((long)B * T * U1, V, J) && ldg == x3_pad(V)), "joint_bwd: pre-split d logits need the bf16x3 path and pitch == roundup(V, 64)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fast = joint_fast(prec, J);
    float* dPE = ws + al4((size_t)B * T * U1 * J);
    float* dPD = dPE + 2 * al4((size_t)B * T * J);
    const int din = de + dd;
    const int M = B * T * U1;
    if (!fast) {
        const float* Hh = ctx;
        const float* dZ = static_cast<const float*>(dlogits);
        float* dH = ws;
        if (prec == 2 && x3_worth(M, V, J)) {
            // bf16x3: ONE split of dZ ([hi | lo], 14 GB read once at C2) serves the dgrad (three terms: K_lo) and, as planes, the wgrad; g_bp = column sums of dZ ride in the wgrad launches (hi + lo: dZ to 2^-17 per element)
            bf16_t* x3 = reinterpret_cast<bf16_t*>(ws + ((ttmi_joint_ws_floats(B, T, U1, J, V) + 63) & ~(size_t)63));
            const int Vp = x3_pad(V), Jp = x3_pad(J);
            bf16_t* Z2 = x3;                                // [hi | lo] rows of dZ (room for three blocks: joint_x3_floats)
            bf16_t* H2 = Z2 + x3_al((size_t)M * 3 * Vp);
            bf16_t* WT3 = H2 + x3_al((size_t)M * 2 * Jp);
            const bool h3 = joint_x3_h3(M, V, J);          // the forward left H as [hi | lo] rows in ctx: they ARE the planes
            if (h3) H2 = reinterpret_cast<bf16_t*>(ctx);
            else CK(split3_bf16(Hh, J, M, J, Jp, 2, H2, st));
            if (presplit) Z2 = reinterpret_cast<bf16_t*>(const_cast<void*>(dlogits));      // the loss gradient kernel wrote the planes itself: no second pass over 14 GB
            else CK(split3_bf16(dZ, ldg, M, V, Vp, 2, Z2, st));
            CK(split3_transpose_bf16(wp, J, V, J, Vp, true, WT3, st));
            CK(gemm_tn_bf16(Z2, H2, g_wp, V, J, M, 2L * Vp, 2L * Jp, J, 1, st, g_bp));
            CK(gemm_tn_bf16(Z2 + Vp, H2, g_wp, V, J, M, 2L * Vp, 2L * Jp, J, 1, st, g_bp));
            CK(gemm_tn_bf16(Z2, H2 + Jp, g_wp, V, J, M, 2L * Vp, 2L * Jp, J, 1, st));
            CK(x3_launch(Z2, WT3, dH, M, J, Vp, J, NtEpilogue(), true, st));       // dgrad: the hi block of dZ a second time against the weight's lo block
            if (h3) {
                CK(fill_zero(dPD, sizeof(float) * (size_t)B * U1 * J, st));
                CK(joint_tanh_bwd_x3(dH, H2, B, T, U1, J, Jp, dPE, dPD, st));
            }
        } else {
            CK(colsum(dZ, ldg, M, V, 1, 1, 0, 0, 0, 0, g_bp, st));
            CK(wgrad(dZ, Hh, g_wp, V, J, M, ldg, J, J, prec, st));
            CK(ttmi_launch_gemm(mk(dZ, wp, dH, M, J, V, ldg, J, J, NN_, prec), st));
        }
        if (!(prec == 2 && joint_x3_h3(M, V, J))) {
            CK(fill_zero(dPD, sizeof(float) * (size_t)B * U1 * J, st));
            CK(joint_tanh_bwd(dH, Hh, 0, B, T, U1, J, dPE, dPD, st));
        }
    } else {
        TTMI_REQUIRE(ldg % 8 == 0 && aligned16(dlogits), "joint_bwd: bf16 dlogits need 16-byte alignment and pitch %% 8 == 0");
        const bf16_t* H16 = reinterpret_cast<const bf16_t*>(ctx);
        const bf16_t* dZ = static_cast<const bf16_t*>(dlogits);
        bf16_t* dH16 = reinterpret_cast<bf16_t*>(ws);
        bf16_t* WpT16 = reinterpret_cast<bf16_t*>(dPD + 2 * al4((size_t)B * U1 * J));    // [J, ldg], zero beyond V
        TTMI_REQUIRE((size_t)J * ldg <= 2 * al4((size_t)J * (((size_t)V + 63) / 64 * 64)), "joint_bwd: pitch %ld too large for the workspace", ldg);
        if (!srow) CK(gemm_tn_bf16(dZ, H16, g_wp, V, J, M, ldg, J, J, 1, st, g_bp));      // g_bp = column sums of dZ, fused
        Shadow shp;
        const bf16_t* wpT16 = WpT16;
        if (shadow_of(wp, V, J, ldg, shp)) wpT16 = shp.wT16;                    // [J, ldg], zero beyond V (kept so by the refresh kernel)
        else CK(transpose_convert_bf16(wp, V, J, WpT16, ldg, st));
        NtEpilogue e;                                                           // dH * (1 - H^2) in the dgrad epilogue
        e.mask = H16;
        e.mask_mode = 1;
        e.rowscale = srow;                                                      // exp-domain form: ... * srow[r], and H leaves as srow[r] * H
        CK(gemm_nt_bf16(dZ, wpT16, dH16, 1, e, M, J, (int)ldg, ldg, ldg, J, st));
        if (srow) {
            // g_wp = P^T (s . H), g_bp = P^T s: the row factor moved onto the other operand (the dgrad epilogue rewrote H in place).
            // The reduction runs over whole 64-row K-tiles: rows [M, Mp) of P, of s . H and of the bf16 row factors are zero-filled here
            // (the buffers have room for them: ttmi_joint_exp_padded_rows / ttmi_joint_ctx_floats), so they add exact zeros - before
            // round 4 a batch whose row count was not a multiple of 64 silently left the exp-domain path (train.py:32-35 trims every
            // batch to its own maxima, so that was most batches)
            const bf16_t* Hs = reinterpret_cast<const bf16_t*>(ctx);
            const int Mp = (M + 63) / 64 * 64;
            if (Mp != M) {
                CK(fill_zero(const_cast<bf16_t*>(dZ) + (size_t)M * ldg, sizeof(bf16_t) * (size_t)(Mp - M) * ldg, st));
                CK(fill_zero(const_cast<bf16_t*>(Hs) + (size_t)M * J, sizeof(bf16_t) * (size_t)(Mp - M) * J, st));
                CK(fill_zero(const_cast<bf16_t*>(static_cast<const bf16_t*>(srow16)) + M, sizeof(bf16_t) * (size_t)(Mp - M), st));
            }
            CK(gemm_tn_bf16(dZ, Hs, g_wp, V, J, Mp, ldg, J, J, 1, st, g_bp, FastBatch(), static_cast<const bf16_t*>(srow16)));
        }
        // the (t, u) sums of dP: two passes through partial rows behind dH16 (the second half of the first workspace region is free until the input layer's backward
        // parks its operands there) - no atomics, dPD bit-reproducible; option 21 = 0: the one-pass kernel with f32 atomics on dPD
        float* part = reinterpret_cast<float*>(dH16 + al8((size_t)M * J));
        if (g_joint_dpd_two_pass && J % 4 == 0 && 2 * joint_sum_bwd_part_floats(B, T, U1, J) + 16 <= (size_t)M * J) {
            CK(joint_sum_bwd_two_pass(dH16, B, T, U1, J, dPE, dPD, part, st));
        } else {
            CK(fill_zero(dPD, sizeof(float) * (size_t)B * U1 * J, st));
            CK(joint_tanh_bwd(dH16, nullptr, 1, B, T, U1, J, dPE, dPD, st));
        }
    }
    if (joint_input_fast(prec, B, T, U1, de, dd, J)) {
        // forward_layer's backward on the throughput kernels; the bf16 operands sit behind dH16 in the first workspace region
        // (M*J floats hold M*J bf16 of dH16 + as many again); g_bf = column sums of dPE come out of the enc wgrad launch
        bf16_t* enc16 = reinterpret_cast<bf16_t*>(ws) + al8((size_t)M * J);
        bf16_t* dec16 = enc16 + al8((size_t)B * T * de);
        bf16_t* dPE16 = dec16 + al8((size_t)B * U1 * dd);
        bf16_t* dPD16 = dPE16 + al8((size_t)B * T * J);
        bf16_t* WfT16 = dPD16 + al8((size_t)B * U1 * J);                       // [din, J] = [We | Wd]^T
        CK(convert_bf16(enc, enc16, (long)B * T * de, st));
        CK(convert_bf16(dec, dec16, (long)B * U1 * dd, st));
        CK(convert_bf16(dPE, dPE16, (long)B * T * J, st));
        CK(convert_bf16(dPD, dPD16, (long)B * U1 * J, st));
        Shadow shf;
        const bf16_t* wfT16 = WfT16;
        if (shadow_of(wf, J, din, J, shf)) wfT16 = shf.wT16;
        else CK(transpose_convert_bf16(wf, J, din, WfT16, J, st));
        CK(gemm_tn_bf16(dPE16, enc16, g_wf, J, de, B * T, J, de, din, 1, fork_stream(st), g_bf));
        CK(gemm_tn_bf16(dPD16, dec16, g_wf + de, J, dd, B * U1, J, dd, din, 1, fork_stream(st)));
        CK(gemm_nt_bf16(dPE16, wfT16, denc, 0, nullptr, B * T, de, J, J, J, de, st));
        CK(gemm_nt_bf16(dPD16, wfT16 + (size_t)de * J, ddec, 0, nullptr, B * U1, dd, J, J, J, dd, st));
        if (g_joint_dec_lo) {
            // dPD's second bf16 term: each entry is a sum over T frames that 32 blocks per label state add in f32 atomic order, and that 1e-7 noise decides the
            // bf16 rounding of one or two of the B U1 J entries differently from run to run - a whole row of d(label states) then moves by 1e-5 of the largest
            // entry, the label encoder's bias gradients by 2e-5 (tools/debug/step0_repro.py).  With the second term the noise stays 1e-7; one B U1-row GEMM more
            bf16_t* dPD16lo = WfT16 + al8((size_t)din * J);
            CK(bf16_residual(dPD, dPD16, dPD16lo, (long)B * U1 * J, st));
            NtEpilogue el;
            el.addend = ddec;
            CK(gemm_nt_bf16(dPD16lo, wfT16 + (size_t)de * J, ddec, 0, el, B * U1, dd, J, J, J, dd, st));
        }
        join_stream(st);
        return TTMI_OK;
    }
    CK(colsum(dPE, J, (long)B * T, J, 1, 1, 0, 0, 0, 0, g_bf, st));
    CK(wgrad(dPE, enc, g_wf, J, de, B * T, J, de, din, prec, st));
    CK(wgrad(dPD, dec, g_wf + de, J, dd, B * U1, J, dd, din, prec, st));
    CK(ttmi_launch_gemm(mk(dPE, wf, denc, B * T, de, J, J, din, de, NN_, prec), st));
    CK(ttmi_launch_gemm(mk(dPD, wf + de, ddec, B * U1, dd, J, J, din, dd, NN_, prec), st));
    return TTMI_OK;
}

int ttmi_joint_bwd(const void* dlogits, long ldg, const float* enc, const float* dec, const float* wf, const float* wp, int B,
                   int T, int U1, int de, int dd, int J, int V, int prec, const float* ctx, float* ws, float* denc, float* ddec,
                   float* g_wf, float* g_bf, float* g_wp, float* g_bp, void* stream) {
    return joint_bwd_impl(dlogits, ldg, enc, dec, wf, wp, B, T, U1, de, dd, J, V, prec, const_cast<float*>(ctx), ws, denc, ddec, g_wf,
                          g_bf, g_wp, g_bp, nullptr, nullptr, stream);
}

// bf16x3: d logits already split into [hi | lo] bf16 planes per row by ttmi_rnnt_loss_bwd_split (ldg = the f32 pitch the planes replaced)
int ttmi_joint_bwd_split_ok(int B, int T, int U1, int J, int V, int prec, long ldg) {
    return prec == 2 && x3_worth((long)B * T * U1, V, J) && ldg == x3_pad(V);
}
int ttmi_joint_bwd_split(const void* dlogits_split, long ldg, const float* enc, const float* dec, const float* wf, const float* wp, int B,
                         int T, int U1, int de, int dd, int J, int V, int prec, const float* ctx, float* ws, float* denc, float* ddec,
                         float* g_wf, float* g_bf, float* g_wp, float* g_bp, void* stream) {
    return joint_bwd_impl(dlogits_split, ldg, enc, dec, wf, wp, B, T, U1, de, dd, J, V, prec, const_cast<float*>(ctx), ws, denc, ddec, g_wf,
                          g_bf, g_wp, g_bp, nullptr, nullptr, stream, true);
}

// P (patched by ttmi_rnnt_loss_bwd_exp) and the row factors srow (f32) / srow16 (bf16) stand for d logits = srow[r] * P[r, :].
// ctx (the hidden activations) is overwritten.
int ttmi_joint_bwd_exp(const void* P, long ldg, const float* srow, const void* srow16, const float* enc, const float* dec,
                       const float* wf, const float* wp, int B, int T, int U1, int de, int dd, int J, int V, int prec, float* ctx,
                       float* ws, float* denc, float* ddec, float* g_wf, float* g_bf, float* g_wp, float* g_bp, void* stream) {
    TTMI_REQUIRE(srow && srow16 && aligned16(srow16), "joint_bwd_exp: row factors missing or misaligned");
    TTMI_REQUIRE(ttmi_joint_exp_supported(B, T, U1, J, V, prec, ldg), "joint_bwd_exp: size / precision outside the fast path");
    return joint_bwd_impl(P, ldg, enc, dec, wf, wp, B, T, U1, de, dd, J, V, prec, ctx, ws, denc, ddec, g_wf, g_bf, g_wp, g_bp, srow,
                          srow16, stream);
}

// ---- bf16 weight shadows
int ttmi_weight_shadow_register(const float* w, int R, int C, const void* w16, const void* wT16, long ldT) {
    TTMI_REQUIRE(w && w16 && wT16 && R > 0 && C > 0 && ldT >= R, "weight_shadow_register: bad arguments");
    TTMI_REQUIRE(aligned16(w16) && aligned16(wT16) && C % 8 == 0 && ldT % 8 == 0, "weight_shadow_register: shadows must be 16-byte aligned with row pitches %% 8 == 0");
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    g_shadows[w] = Shadow{R, C, static_cast<const bf16_t*>(w16), static_cast<const bf16_t*>(wT16), ldT};
    return TTMI_OK;
}
int ttmi_weight_shadow_clear(const float* w) {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    if (w) g_shadows.erase(w);
    else g_shadows.clear();
    return TTMI_OK;
}
int ttmi_weight_shadow_refresh(const long* table, int n, long total_tiles, void* stream) {
    return shadow_refresh(table, n, total_tiles, static_cast<hipStream_t>(stream));
}
// the second term of every shadowed weight's bf16 split, w16lo = bf16(w - float(w16)), at w16 + lo_delta elements (one buffer twice the size): registered per
// weight, refreshed by the same launch as the other two copies
int ttmi_weight_shadow_register_lo(const float* w, const void* w16lo) {
    TTMI_REQUIRE(w && w16lo && aligned16(w16lo), "weight_shadow_register_lo: bad arguments");
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    auto it = g_shadows.find(w);
    TTMI_REQUIRE(it != g_shadows.end(), "weight_shadow_register_lo: the weight has no registered shadow");
    it->second.w16lo = static_cast<const bf16_t*>(w16lo);
    return TTMI_OK;
}
int ttmi_weight_shadow_refresh_lo(const long* table, int n, long total_tiles, long lo_delta, void* stream) {
    return shadow_refresh(table, n, total_tiles, static_cast<hipStream_t>(stream), lo_delta);
}

// CUs that the encoder-sized persistent GEMMs launched on `stream` leave free (for RCCL's kernels running beside a data-parallel
// backward pass): per-stream state, read at launch time.
int ttmi_stream_reserve_cus(void* stream, int n) {
    TTMI_REQUIRE(n >= 0 && n <= 128, "stream_reserve_cus: n = %d outside [0, 128]", n);
    gemm_fast_stream_reserve_cus(static_cast<hipStream_t>(stream), n);
    return TTMI_OK;
}

// A stream that is itself a fork of the caller's main stream (the label encoder's side stream): library calls on it never fork again inside a
// stream capture (nested forks crash hipStreamEndCapture on ROCm 7.2); eager launches are unaffected.
int ttmi_stream_set_nofork(void* stream, int on) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (int i = 0; i < g_nnofork; ++i)
        if (g_nofork[i] == st) {
            if (!on) g_nofork[i] = g_nofork[--g_nnofork];
            return TTMI_OK;
        }
    if (on) {
        TTMI_REQUIRE(g_nnofork < MAX_SIDE, "stream_set_nofork: table full");
        g_nofork[g_nnofork++] = st;
    }
    return TTMI_OK;
}

// Device word (nullable) that every dropout site mixes into its seed when its kernel starts: seeds are drawn on the host per call and, in a
// step captured as a HIP graph, baked into the kernel arguments; bumping this word on the device before each replay gives every step
// new masks.  Forward and backward of one step must see the same value.  Process-wide; nullptr (default) = seeds used as passed.
int ttmi_set_dropout_salt(const unsigned* salt) {
    int dev = 0;
    TTMI_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_SALT_DEV, "set_dropout_salt: no current device / more than %d devices", MAX_SALT_DEV);
    g_drop_salts[dev] = salt;
    return TTMI_OK;
}

// process-wide switches for A/B measurements.  key 0: 1 = disable the fused attention kernels (bf16 pipeline only);
// key 1: throughput-GEMM generation (see gemm_fast.hip); key 2: flash-kernel timing switches; key 3: 0 = no wgrad fork
int ttmi_set_option(int key, int value) {
    TTMI_REQUIRE(key >= 0 && key <= 22, "set_option: unknown key %d", key);
    if (key == 19) { g_joint_dec_lo = value; return TTMI_OK; }
    if (key == 20) { gemm_fast_set_tn_group_pieces(value); return TTMI_OK; }
    if (key == 21) { g_joint_dpd_two_pass = value; return TTMI_OK; }
    if (key == 22) { g_x3_tn_one_launch = value; return TTMI_OK; }
    if (key == 18) { g_capture_forks = value; return TTMI_OK; }
    if (key == 17) { gemm_fast_set_f32(value); return TTMI_OK; }
    if (key == 16) { g_scatter_launch = value; return TTMI_OK; }
    if (key == 15) { flash_set_bwd_gen(value); return TTMI_OK; }
    if (key == 14) { flash_set_resident(value); return TTMI_OK; }
    if (key == 13) { g_split_weights = value; return TTMI_OK; }
    if (key == 12) { g_ln_bwd_grid = value < 1 ? 1 : value; return TTMI_OK; }
    if (key == 11) { g_posgrad_gemms = value; return TTMI_OK; }
    if (key == 10) { g_attn_slices = value < 1 ? 1 : value; return TTMI_OK; }
    if (key == 9) { ttmi_rnnt_set_lattice_version(value); return TTMI_OK; }
    if (key == 8) { g_inkernel_pos = value; return TTMI_OK; }
    if (key == 6) { gemm_fast_set_reserved_cus(value); return TTMI_OK; }
    if (key == 7) { ttmi_gemm_set_skinny_rows(value); return TTMI_OK; }
    if (key == 4) { gemm_fast_set_tn_target(value); return TTMI_OK; }
    if (key == 5) { g_gemm_slab = value & 1; relpos_slab_set_debug(value >> 1); return TTMI_OK; }
    if (key == 0) g_disable_fused_attention = value;
    else if (key == 1) gemm_fast_set_version(value);
    else if (key == 2) g_flash_debug = value;
    else g_fork_wgrad = value;
    return TTMI_OK;
}

// out[i] = in[i] * (0 or 1/(1-p)) with the library's counter-based mask for (seed, i): lets tests and callers
// reproduce the exact masks the fused sub-layers use (sites: attention out seed^0xA1, FFN inner ^0xB2, FFN out ^0xC3,
// layer out ^0xD4; element index = row * width + column of the masked [rows, width] tensor).
int ttmi_dropout_apply(const float* in, long n, float p, unsigned seed, float* out, void* stream) {
    DropSpec ds;
    ds.p = p; ds.seed = seed; ds.salt = g_drop_salt;
    return dropout_apply(in, n, ds, out, nullptr, static_cast<hipStream_t>(stream));
}

// greedy decoding support (tt/model.py:76-83): rows = joint logits of consecutive frames against ONE label state; returns in
// *out the first frame whose argmax is not blank and the symbol, so the host syncs once per emitted symbol, not per frame.
int ttmi_greedy_scan(const void* logits, int dtype, long ld, int n, int V, int blank, unsigned long long* out, void* stream) {
    return greedy_scan(logits, dtype, ld, n, V, blank, out, static_cast<hipStream_t>(stream));
}

// batched greedy decoding, every utterance of a batch in lockstep over symbol steps (ttmi.h)
int ttmi_greedy_scan_batch(const void* logits, int dtype, long ld, int B, int n, int V, int blank, const int* t, const int* T_len,
                           const int* need, unsigned long long* key, void* stream) {
    return greedy_scan_batch(logits, dtype, ld, B, n, V, blank, t, T_len, need, key, static_cast<hipStream_t>(stream));
}
int ttmi_greedy_advance(unsigned long long* key, int B, int n, int n_hist, long* hist, long ld_hist, int* t, const int* T_len, int* need,
                        int* done, int* count, int* flags, void* stream) {
    return greedy_advance(key, B, n, n_hist, hist, ld_hist, t, T_len, need, done, count, flags, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------ embedding (tt/decoder.py:26,39)
int ttmi_embed_fwd(const long* tokens, const float* W, long n, int d, int V, float* out, void* stream) {
    return embed_fwd(tokens, W, n, d, V, out, static_cast<hipStream_t>(stream));
}
int ttmi_embed_bwd(const long* tokens, const float* dout, long n, int d, int V, int padding_idx, float* gW, void* stream) {
    return embed_bwd(tokens, dout, n, d, V, padding_idx, gW, static_cast<hipStream_t>(stream));
}

}  // extern "C"
