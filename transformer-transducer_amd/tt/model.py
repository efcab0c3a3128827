"""Transducer / JointNet with the reference's API surface (tt/model.py): .encoder/.decoder/.joint,
forward(inputs[B,T,d], targets[B,U]) -> logits[B,T,U+1,V], decode/recognize (greedy), beam search."""
import copy
import heapq

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ttmi import ops
from ttmi.ops import MaskSpec
from tt.decoder import BuildDecoder
from tt.encoder import BuildEncoder
from tt.transformer import default_precision, grad_targets
from tt.utils import context_mask, look_ahead_mask  # noqa: F401  (re-exported like the reference module)


class _JointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, dec, wf, bf, wp, bp, prec):
        enc, dec = enc.contiguous(), dec.contiguous()
        params = (wf, bf, wp, bp)
        wf, bf, wp, bp = (t.detach() for t in (wf, bf, wp, bp))
        logits, saved = ops.joint_fwd(enc, dec, wf, bf, wp, bp, prec)
        ctx.save_for_backward(enc, dec, wf, wp, saved)
        ctx.prec = prec
        ctx.params = params
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        enc, dec, wf, wp, saved = ctx.saved_tensors
        g, rets, after = grad_targets(ctx.params, ("wf", "bf", "wp", "bp"))
        denc, ddec = ops.joint_bwd(dlogits, enc, dec, wf, wp, saved, ctx.prec, g)
        for cb in after:
            cb()
        return (denc, ddec, *rets, None)


class JointNet(nn.Module):
    """logits = project_layer(tanh(forward_layer(cat(enc, dec)))) evaluated in split-weight form
    (forward_layer.weight = [W_enc | W_dec]); accepts [B,T,de]/[B,U,dd] (lattice) or two 1-D vectors (decode)."""

    def __init__(self, input_size, inner_dim, vocab_size):
        super().__init__()
        self.forward_layer = nn.Linear(input_size, inner_dim, bias=True)
        self.tanh = nn.Tanh()
        self.project_layer = nn.Linear(inner_dim, vocab_size, bias=True)

    def forward(self, enc_state, dec_state):
        if enc_state.dim() == 3 and dec_state.dim() == 3:
            squeeze = None
        else:
            assert enc_state.dim() == dec_state.dim()
            squeeze = enc_state.shape[:-1]
            enc_state = enc_state.reshape(-1, 1, enc_state.shape[-1])     # N vectors -> N lattices of 1 x 1
            dec_state = dec_state.reshape(-1, 1, dec_state.shape[-1])
        out = _JointFn.apply(enc_state, dec_state, self.forward_layer.weight, self.forward_layer.bias,
                             self.project_layer.weight, self.project_layer.bias, default_precision())
        return out if squeeze is None else out.reshape(*squeeze, out.shape[-1])


class _LabelStateGraphs:
    """Greedy decoding re-runs the label encoder on the whole token history after every emitted symbol (tt/model.py:75,88 of the
    reference): ~100 tiny launches whose cost is the host issuing them.  One captured graph per history length L replays them with a
    single launch; the token history lives on the device (`master`), each graph reads its own static copy and writes a static output.
    Graphs are captured lazily, share one memory pool and are dropped when the label encoder's parameter storage or the precision
    mode changes."""
    MAX_L = 128

    def __init__(self, model, device):
        self.decoder, self.device = model.decoder, device
        self.master = torch.zeros(1, self.MAX_L, dtype=torch.long, device=device)
        self.stream = torch.cuda.Stream(device)
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs = {}
        self.key = self.weights_key(model)
        # every captured graph bakes in the address of this stream's scratch arena (ttmi.ops.scratch), which is re-allocated when a
        # longer history needs more: size it for MAX_L before the first capture, and drop the graphs should it ever move all the same
        cur = torch.cuda.current_stream(device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.decoder(torch.zeros(1, self.MAX_L, dtype=torch.long, device=device))
        cur.wait_stream(self.stream)
        self.arena = ops.scratch_generation(device, self.stream)

    @staticmethod
    def weights_key(model):
        return tuple(p.data_ptr() for p in model.decoder.parameters()) + (default_precision(),)

    def set_token(self, pos, tok):
        self.master[0, pos] = tok                                   # a device fill, no synchronisation

    def state(self, L):
        """label-encoder output at the last position of master[:, :L] -> [1, 1, d] (static buffer of graph L)"""
        if ops.scratch_generation(self.device, self.stream) != self.arena:      # the arena moved: every graph points at freed memory
            self.graphs.clear()
            self.arena = ops.scratch_generation(self.device, self.stream)
        entry = self.graphs.get(L)
        if entry is None:
            tok = self.master[:, :L].clone()
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                for _ in range(2):                                  # warm-up outside the capture: scratch arenas, kernel attributes
                    self.decoder(tok)
            cur.wait_stream(self.stream)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self.pool, stream=self.stream):
                out = self.decoder(tok)[:, -1:, :]
            assert ops.scratch_generation(self.device, self.stream) == self.arena, "scratch arena grew during a capture"
            entry = self.graphs[L] = (graph, tok, out)
        graph, tok, out = entry
        tok.copy_(self.master[:, :L])
        graph.replay()
        return out


class Transducer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.encoder = BuildEncoder(config)
        self.decoder = BuildDecoder(config)
        self.joint = JointNet(input_size=config.joint.input_size, inner_dim=config.joint.inner_size,
                              vocab_size=config.vocab_size)
        if config.share_embedding:
            # the reference's branch dereferences self.decoder.embedding, which does not exist (tt/model.py:53-56)
            raise AttributeError("'BuildDecoder' object has no attribute 'embedding' (share_embedding is broken upstream)")

    def forward(self, inputs, targets):
        targets = F.pad(targets, pad=[1, 0, 0, 0], value=0)                 # leading blank / SOS
        audio_mask = self._audio_mask(inputs)
        if self.config.overlap_label_encoder and inputs.is_cuda:
            # the label encoder (tiny, launch-latency-bound kernels) is independent of the audio encoder until the joint:
            # run it on a side stream so both fill the chip together; autograd replays the same streams in backward
            main, side = torch.cuda.current_stream(inputs.device), ops.side_stream(inputs.device)
            side.wait_stream(main)                                          # the side stream needs `targets`, nothing else
            # host order: the audio encoder's long kernels are queued FIRST, the label encoder's ~400 tiny launches are issued while the
            # GPU is busy with them (issued first, they left the chip nearly idle for the ~1 ms the host needs to enqueue them)
            enc_state = self.encoder(inputs, audio_mask)
            with torch.cuda.stream(side):
                dec_state = self.decoder(targets, MaskSpec(1))
            main.wait_stream(side)
            dec_state.record_stream(main)
        else:
            enc_state = self.encoder(inputs, audio_mask)
            dec_state = self.decoder(targets, MaskSpec(1))                  # == look_ahead_mask(targets)[:, :, None]
        return self.joint(enc_state, dec_state)

    def _audio_mask(self, inputs):
        """Reference behaviour is audio_mask=None (tt/model.py:60-61).  Opt-in `config.streaming` (absent in the
        reference YAMLs => None => off): {'left': l, 'right': r} band mask or {'chunk': c, 'left': l} block mask."""
        s = self.config.streaming
        if not s:
            return None
        if "chunk" in s:
            from tt.utils import chunk_mask
            return chunk_mask(inputs, s["chunk"], s.get("left", 0))[:, :, None]
        return MaskSpec(2, left=int(s.get("left", 0)), right=int(s.get("right", 0)))

    @torch.no_grad()
    def decode(self, enc_state, lengths, block=64):
        """Greedy: <= 1 symbol per frame, label encoder re-run on the whole history WITHOUT look-ahead mask (tt/model.py:70-90).
        Same token sequence as the reference's per-frame loop, but frames are scored `block` at a time against the current
        label state and the first non-blank frame is found on the device (ttmi_greedy_scan): one host sync per emitted
        symbol instead of one per frame.  The label-encoder re-runs replay captured graphs (one per history length, see
        _LabelStateGraphs)."""
        token_list = [0]
        dev = enc_state.device
        T = int(lengths)
        graphs = self._label_state_graphs(dev)

        def label_state():
            """label-encoder output for the current history -> [1, 1, d]"""
            L = len(token_list)
            if graphs is not None and L <= graphs.MAX_L:
                graphs.set_token(L - 1, token_list[-1])
                return graphs.state(L)
            return self.decoder(torch.tensor([token_list], dtype=torch.long, device=dev))[:, -1:, :]

        dec_state = label_state()
        t = 0
        while t < T:
            n = min(block, T - t)
            logits = self.joint(enc_state[t:t + n].unsqueeze(0), dec_state)                                  # [1, n, 1, V]
            row, tok = ops.greedy_scan(logits[0, :, 0, :])
            if tok is None:
                t += n
                continue
            if tok >= self.config.vocab_size:   # NaN logits (bad weights / inputs) come back as an impossible symbol: fail here, not later
                raise RuntimeError("greedy decode: the joint produced no finite maximum at frame %d (NaN logits?)" % (t + row))
            token_list.append(tok)
            dec_state = label_state()
            t += row + 1                                                    # the emitting frame is consumed
        return token_list[1:]

    def _label_state_graphs(self, dev):
        """the per-model graph cache of `decode` (None: CPU tensors, or switched off with config.decode_graphs = False)"""
        if dev.type != "cuda" or self.config.decode_graphs is False:
            return None
        g = self.__dict__.get("_decode_graphs")
        if g is None or g.device != dev or g.key != _LabelStateGraphs.weights_key(self):
            g = self.__dict__["_decode_graphs"] = _LabelStateGraphs(self, dev)
        return g

    @torch.no_grad()
    def recognize(self, inputs, inputs_length=None, audio_mask=None):
        enc_states = self.encoder(inputs, audio_mask)
        return [self.decode(enc_states[b], inputs_length[b]) for b in range(inputs.size(0))]

    @torch.no_grad()
    def beam_search(self, enc_state, lengths, beam_width=5):
        """Restatement of the reference's beam search (tt/model.py:110-179) including its quirks: frames advance on the
        currently most probable hypothesis; expansion happens only when that hypothesis predicts a non-blank; child
        token lists persist across expansions (they are appended to, never re-seeded from their parents)."""
        dev = enc_state.device

        def posterior(tokens, t):
            d = self.decoder(torch.tensor([tokens], dtype=torch.long, device=dev))[:, -1, :]
            return F.softmax(self.joint(enc_state[t].view(-1), d.view(-1)), dim=0)

        hyps = [[0] for _ in range(beam_width)]
        score = np.zeros((beam_width,), dtype=float)
        child = [[[0] for _ in range(beam_width)] for _ in range(beam_width)]
        child_score = np.zeros((beam_width, beam_width), dtype=float)
        first = True
        for t in range(int(lengths)):
            lead = int(score.argmax())
            if int(torch.argmax(posterior(hyps[lead], t)).item()) == 0:
                continue
            for k in range(beam_width):
                values, indices = torch.topk(posterior(hyps[k], t), k=beam_width + 1, dim=0)
                values, indices = values.tolist(), indices.tolist()
                drop = indices.index(0) if 0 in indices else len(indices) - 1
                indices.pop(drop)
                values.pop(drop)
                for i, tok in enumerate(indices):
                    if first:
                        child[i][k].append(tok)
                    else:
                        child[k][i].append(tok)
                if first:
                    child_score[:, k] = np.log(values)
                else:
                    child_score[k] = score[k] + np.log(values)
            if first:
                first = False
                for i in range(beam_width):
                    hyps[i] = copy.deepcopy(child[i][0])
                    score[i] = child_score[i, 0]
            else:
                best = heapq.nlargest(beam_width, range(beam_width ** 2), child_score.take)
                for i, idx in enumerate(best):
                    score[i] = child_score[idx // beam_width, idx % beam_width]
                    hyps[i] = copy.deepcopy(child[idx // beam_width][idx % beam_width])
        return hyps[int(score.argmax())][1:]

    @torch.no_grad()
    def recognize_beam_search(self, inputs, inputs_length, audio_mask=None):
        enc_states = self.encoder(inputs, audio_mask)
        return [self.beam_search(enc_states[b], inputs_length[b], beam_width=5) for b in range(inputs.size(0))]
