"""Sliding-window streaming recogniser on the MI355X path: the window logic of the reference's
audio/streamRec_unlimit_dynamic_window.py:111-216 without its microphone and GUI.

Per audio window (15 999 samples every 15 519, :53-54): log-mel, drop the 3 incomplete frames at the end (:128), stack each frame with
3 frames of history carried over from the previous window (:134-137), subsample by 3 continuing the global phase (:142-148); once more
than n_layer * right_context subsampled frames lie beyond the decoding position (or on the last window) run the audio encoder on
[position - n_layer * left_context, end) under context_mask(left, right) and keep the centre (:153-179); greedy search over the kept
frames, at most one symbol per frame, label encoder re-run on the last 40 symbols (:181-204).  Quirks that change tokens are kept and
tested against a reference run: the first window loses its first 3 frames; the label history carries no leading blank after the first
symbol; with right_frame == 0 the kept slice `[left:-0]` is empty, so the last window is encoded but never decoded.

Everything stays on the device: features are assembled by the front-end kernels, the encoder takes the band mask parametrically, frames
are scored in blocks and the first emitting frame is found on the device (one host sync per emitted symbol, as in Transducer.decode).
"""
import torch

from . import frontend, ops
from .ops import MaskSpec


class StreamingRecognizer:
    WIN_AUDIO, AUDIO_STEP = 15999, 15519             # samples (:53-54)

    def __init__(self, model, left_context=None, right_context=None, n_layer=None, n_mels=128, max_history=40, block=64,
                 sample_rate=16000):
        enc = model.config.enc
        self.model = model
        self.left = enc.left_context if left_context is None else left_context
        self.right = enc.right_context if right_context is None else right_context
        n_layer = enc.n_layer if n_layer is None else n_layer
        self.left_len, self.right_len = n_layer * self.left, n_layer * self.right
        self.n_mels, self.max_history, self.block, self.sr = n_mels, max_history, block, sample_rate
        self.device = next(model.parameters()).device
        self.reset()

    def reset(self):
        d = self.device
        self.tail = torch.empty(0, self.n_mels, device=d)          # the last 3 log-mel rows seen (history for the stacking)
        self.n_concat = 0                                          # stacked rows so far (the subsampling phase)
        self.sub = torch.empty(0, 4 * self.n_mels, device=d)       # subsampled rows from global index `base` on
        self.base = 0
        self.total = 0                                             # subsampled rows so far
        self.pos = 0                                               # next frame to decode
        self.result, self.breaks, self.windows = [], [], []
        self.blank_frame = 0
        with torch.no_grad():
            self.dec_state = self.model.decoder(torch.zeros(1, 1, dtype=torch.long, device=d))[:, -1:, :]      # :113-115

    @torch.no_grad()
    def feed(self, win_log_mel, last=False):
        """win_log_mel [frames, n_mels] (device tensor or numpy) = get_feature(audio window); returns the symbols emitted by this window"""
        feat = torch.as_tensor(win_log_mel, dtype=torch.float32).to(self.device)[:-3]
        n = feat.shape[0]
        rows = torch.cat([self.tail, feat], 0)
        self.tail = rows[-3:].clone()
        if rows.shape[0] > 3:
            stacked = frontend.stack_subsample(rows[None], None, 3, 0, 1)[0][0][3:]      # rows 0..2 only serve as history (dropped, :137)
            skip = (0, 2, 1)[self.n_concat % 3]
            self.n_concat += stacked.shape[0]
            fresh = stacked[skip::3]
            if fresh.shape[0]:
                self.sub = torch.cat([self.sub, fresh], 0)
                self.total += fresh.shape[0]
        emitted = []
        if self.total - self.pos > self.right_len or last:
            left_frame, right_frame = self.left_len, self.right_len
            start = self.pos - left_frame
            if start < 0:
                left_frame, start = self.pos, 0
            if last:
                right_frame = 0
            win = self.sub[start - self.base:][None]
            enc = self.model.encoder(win, MaskSpec(2, left=self.left, right=self.right))
            self.windows.append((start, self.total, left_frame, right_frame, enc))
            eff = enc[0, left_frame:-right_frame]                  # :177-179 verbatim: empty when right_frame == 0
            self._greedy(eff, emitted)
            self.pos += eff.shape[0]
            keep = max(0, self.pos - self.left_len)                # rows in front of the next window's history are never read again
            self.sub, self.base = self.sub[keep - self.base:], keep
        return emitted

    def _greedy(self, eff, emitted):
        t, T = 0, eff.shape[0]
        while t < T:
            n = min(self.block, T - t)
            logits = self.model.joint(eff[t:t + n].unsqueeze(0), self.dec_state)              # [1, n, 1, V]
            row, tok = ops.greedy_scan(logits[0, :, 0, :])
            if self.result:
                self.blank_frame += n if tok is None else row
            if tok is None:
                t += n
                continue
            if tok >= self.model.config.vocab_size:
                raise RuntimeError("streaming decode: the joint produced no finite maximum (NaN logits?)")
            if self.blank_frame >= 15:                             # the GUI starts a new line here (:188-191)
                self.breaks.append(len(self.result))
            self.result.append(tok)
            emitted.append(tok)
            hist = torch.tensor([self.result[-self.max_history:]], dtype=torch.long, device=self.device)
            self.dec_state = self.model.decoder(hist)[:, -1:, :]   # :197-203: history of at most 40 symbols, no leading blank
            self.blank_frame = 0
            t += row + 1

    @torch.no_grad()
    def feed_audio(self, samples, last=False):
        """samples: one audio window (int16, device tensor or numpy); log-mel by the GPU front-end (get_feature, tt/utils.py:182-193)"""
        w = torch.as_tensor(samples).to(device=self.device, dtype=torch.int16).reshape(1, -1)
        n = torch.tensor([w.shape[1]], dtype=torch.int32, device=self.device)
        return self.feed(frontend.log_mel(w, n, self.sr, self.n_mels, "ln")[0], last)

    def recognize(self, audio):
        """a complete recording through the window loop of :118-127,206-212: windows of WIN_AUDIO samples every AUDIO_STEP; the window that
        reaches the end of the recording is the (shorter) last one"""
        self.reset()
        n, p = len(audio), 0
        while True:
            last = p + self.WIN_AUDIO >= n
            self.feed_audio(audio[p:n] if last else audio[p:p + self.WIN_AUDIO], last)
            if last:
                return list(self.result)
            p += self.AUDIO_STEP
