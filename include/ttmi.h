/* libttmi - MI355X-native Transformer-Transducer hot path, C ABI.
 *
 * Every entry point: plain pointers and sizes, no torch types; all pointers are
 * DEVICE pointers unless noted; `stream` is a hipStream_t (0 = default stream);
 * nothing is allocated or freed inside (workspaces are caller-provided, sizes via
 * *_workspace_bytes); launches are asynchronous on `stream`, no device-wide sync.
 * Return value: 0 ok, <0 invalid argument (message in ttmi_last_error()), >0 a
 * hipError_t.  No C++ exception crosses the boundary.
 *
 * The reference (zzpDapeng/Transformer-Transducer) is pure Python and has no FFI
 * layer; each function cites the reference lines whose arithmetic it replaces
 * (paths relative to the reference root).  The Python classes that bind these are
 * in transformer-transducer_amd/{tt,warprnnt_pytorch}; see INTEGRATION.md.
 */
#ifndef TTMI_H
#define TTMI_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

int ttmi_version(void);
const char* ttmi_last_error(void);   /* thread-local, valid until the next failing call */

/* ---- RNN-T loss: replaces warprnnt_pytorch.RNNTLoss (train.py:13,53,231) ------------- */
size_t ttmi_rnnt_workspace_bytes(int B, int T, int U1);
/* logits f32 [B,T,U1,V]; labels i32 [B,U1-1]; act_lens,label_lens i32 [B]; costs f32 [B] */
int ttmi_rnnt_loss_fwd(const float* logits, const int* labels, const int* act_lens, const int* label_lens, int B, int T,
                       int U1, int V, int blank, void* workspace, float* costs, void* stream);
/* grad[b] = scale * grad_out[b*grad_out_stride] * d costs[b] / d logits; grad may alias logits */
int ttmi_rnnt_loss_bwd(const float* logits, const int* labels, const int* act_lens, const int* label_lens, int B, int T,
                       int U1, int V, int blank, const void* workspace, const float* grad_out, int grad_out_stride,
                       float scale, float* grad, void* stream);

#ifdef __cplusplus
}
#endif
#endif
