"""Long-form streaming: overlap the Firefly vocoder with LM generation (BASELINE configs[4]).

The reference vocodes a whole utterance after generation finishes (server/lib/handlers/speech.rs:98-129; SURVEY.md §3c:
"no LM/vocoder overlap and no frame-streaming in the reference").  Every convolution of the 1.4+/1.5 codec is causal
(codec/utils/mod.rs:53-62,110-122), so the PCM of frames [a, b) depends only on codes [a - R, b) with a finite receptive
field R: chunks can be vocoded while the LM is still generating, bit-identically to one-shot decoding.

R in frames (2048 samples): conv_post 12/2048 + per HiFi-GAN stage 180 samples (k=11 branch: 2 convs x 10 taps x (1+3+5))
at 2048, 1024, 512, 256 and 32 samples/frame + conv_pre 12/4 + ConvNeXt depthwise 6/4 + 6/2  ~= 14.5 frames; HALO = 24.
"""
import queue
import threading
import time

import numpy as np

HALO = 24


def decode_chunk(codec, codes, a, b, halo=HALO):
    """PCM of frames [a, b) of `codes` (8, T): decode [a - halo, b) and keep the tail."""
    lo = max(0, a - halo)
    pcm = codec.decode(np.ascontiguousarray(codes[None, :, lo:b]))[0, 0]
    return pcm[2048 * (a - lo):]


class StreamingSynth:
    """generate_blocking + FireflyCodec.decode with the vocoder running in a worker thread on its own HIP stream."""

    def __init__(self, lm, codec, chunk=64, halo=HALO):
        self.lm, self.codec, self.chunk, self.halo = lm, codec, chunk, halo

    def __call__(self, prompt, max_new_tokens, **gen_kw):
        Cb = self.lm.cfg["num_codebooks"]
        frames, q, pcm_parts = [], queue.Queue(), []
        t_busy = [0.0]

        def worker():
            done_upto = 0
            while True:
                item = q.get()
                if item is None:
                    break
                n = item
                # vocode every complete chunk available so far
                while done_upto + self.chunk <= n:
                    codes = np.array(frames[: done_upto + self.chunk], np.uint32).T
                    t0 = time.perf_counter()
                    pcm_parts.append(decode_chunk(self.codec, codes, done_upto, done_upto + self.chunk, self.halo))
                    t_busy[0] += time.perf_counter() - t0
                    done_upto += self.chunk
            n = len(frames)
            if n > done_upto:  # tail
                codes = np.array(frames, np.uint32).T
                t0 = time.perf_counter()
                pcm_parts.append(decode_chunk(self.codec, codes, done_upto, n, self.halo))
                t_busy[0] += time.perf_counter() - t0

        th = threading.Thread(target=worker)
        th.start()

        def on_frame(idx, codes):
            frames.append(list(codes))
            if len(frames) % self.chunk == 0:
                q.put(len(frames))
            return False

        t0 = time.perf_counter()
        out = self.lm.generate_blocking(prompt, max_new_tokens, on_frame=on_frame, **gen_kw)
        t_lm = time.perf_counter() - t0
        q.put(None)
        th.join()
        t_all = time.perf_counter() - t0
        assert out.shape[1] == len(frames) and Cb == out.shape[0]
        pcm = np.concatenate(pcm_parts) if pcm_parts else np.zeros(0, np.float32)
        self.stats = dict(frames=len(frames), lm_s=t_lm, vocoder_busy_s=t_busy[0], total_s=t_all,
                          overlap_efficiency=(t_lm + t_busy[0] - t_all) / max(t_busy[0], 1e-9))
        return out, pcm
