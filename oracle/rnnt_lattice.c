/* CPU restatement of the RNN-T loss (forward-backward lattice + gradient w.r.t.
 * logits).  TEST INFRASTRUCTURE ONLY - the checker for the HIP kernels, never
 * linked into the product library.
 *
 * The reference obtains this op from the un-vendored third-party package
 * `warprnnt_pytorch` (requirements.txt:7, no version pin; call sites
 * train.py:13,53,231).  This file restates the published algorithm (Graves
 * 2012, "Sequence Transduction with RNNs", eqs. 16-20) under that call
 * contract: logits in (softmax taken internally), blank id, per-utterance
 * lengths, gradient returned w.r.t. the logits.  Parity is pinned by
 * tests/test_oracle_rnnt.py (public warp-transducer known-answer vector,
 * brute-force enumeration, float64 autograd fixtures); the reference holds no
 * vector of its own for this op ("parity unpinned" at the reference boundary).
 *
 * Layout: logits [B, T, U1, V] row-major float; labels [B, U] int32.
 * Build: make -C oracle   ->  oracle/librnnt_oracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double lae(double a, double b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    double m = a > b ? a : b;
    return m + log1p(exp(-fabs(a - b)));
}

/* returns 0 on success.  costs[B]; grad same shape as logits (already scaled by
 * grad_scale, e.g. 1/B for reduction='mean'); grad may be NULL. */
int rnnt_oracle_f32(const float* logits, const int* labels, const int* act_lens, const int* label_lens,
                    int B, int T, int U1, int V, int blank, float grad_scale, float* costs, float* grad) {
    const int U = U1 - 1;
    double* lpb = (double*)malloc(sizeof(double) * T * U1);
    double* lpl = (double*)malloc(sizeof(double) * T * U1);
    double* lse = (double*)malloc(sizeof(double) * T * U1);
    double* al = (double*)malloc(sizeof(double) * T * U1);
    double* be = (double*)malloc(sizeof(double) * T * U1);
    if (!lpb || !lpl || !lse || !al || !be) return 1;
    if (grad) memset(grad, 0, sizeof(float) * (size_t)B * T * U1 * V);
    for (int b = 0; b < B; ++b) {
        const int Tb = act_lens[b], Ub = label_lens[b];
        if (Tb < 1 || Tb > T || Ub < 0 || Ub > U) return 2;
        const float* x = logits + (size_t)b * T * U1 * V;
        const int* y = labels + (size_t)b * U;
        /* rows are independent: threads over t (the GPU box's host cores; results do not depend on the thread count) */
        #pragma omp parallel for schedule(static)
        for (int t = 0; t < Tb; ++t)
            for (int u = 0; u <= Ub; ++u) {
                const float* r = x + ((size_t)t * U1 + u) * V;
                float mx = r[0];
                for (int v = 1; v < V; ++v) mx = r[v] > mx ? r[v] : mx;
                double s = 0;
                for (int v = 0; v < V; ++v) s += exp((double)r[v] - mx);
                double l = mx + log(s);
                lse[t * U1 + u] = l;
                lpb[t * U1 + u] = r[blank] - l;
                lpl[t * U1 + u] = u < Ub ? r[y[u]] - l : -INFINITY;
            }
        for (int t = 0; t < Tb; ++t)
            for (int u = 0; u <= Ub; ++u) {
                if (t == 0 && u == 0) { al[0] = 0; continue; }
                double a = t > 0 ? al[(t - 1) * U1 + u] + lpb[(t - 1) * U1 + u] : -INFINITY;
                double c = u > 0 ? al[t * U1 + u - 1] + lpl[t * U1 + u - 1] : -INFINITY;
                al[t * U1 + u] = lae(a, c);
            }
        for (int t = Tb - 1; t >= 0; --t)
            for (int u = Ub; u >= 0; --u) {
                if (t == Tb - 1 && u == Ub) { be[t * U1 + u] = lpb[t * U1 + u]; continue; }
                double a = t < Tb - 1 ? be[(t + 1) * U1 + u] + lpb[t * U1 + u] : -INFINITY;
                double c = u < Ub ? be[t * U1 + u + 1] + lpl[t * U1 + u] : -INFINITY;
                be[t * U1 + u] = lae(a, c);
            }
        const double ll = be[0];
        costs[b] = (float)(-ll);
        if (!grad) continue;
        float* g = grad + (size_t)b * T * U1 * V;
        #pragma omp parallel for schedule(static)
        for (int t = 0; t < Tb; ++t)
            for (int u = 0; u <= Ub; ++u) {
                const float* r = x + ((size_t)t * U1 + u) * V;
                float* gr = g + ((size_t)t * U1 + u) * V;
                const double a = al[t * U1 + u], l = lse[t * U1 + u];
                const double c = a + be[t * U1 + u] - ll - l;
                for (int v = 0; v < V; ++v) gr[v] = (float)(grad_scale * exp(c + r[v]));
                double eb;
                if (t == Tb - 1 && u == Ub) eb = exp(a + lpb[t * U1 + u] - ll);
                else if (t < Tb - 1) eb = exp(a + lpb[t * U1 + u] + be[(t + 1) * U1 + u] - ll);
                else eb = 0;
                gr[blank] -= (float)(grad_scale * eb);
                if (u < Ub) gr[y[u]] -= (float)(grad_scale * exp(a + lpl[t * U1 + u] + be[t * U1 + u + 1] - ll));
            }
    }
    free(lpb); free(lpl); free(lse); free(al); free(be);
    return 0;
}
