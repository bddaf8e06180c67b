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
    """generate_blocking + FireflyCodec.decode with the vocoder running in a worker thread on its own HIP stream.

    chunk: frames per vocoder call once the stream is under way; first_chunk: size of the first call (time to first audio).  Every call
    re-decodes `halo` frames of left context (all convolutions are causal), so larger steady-state chunks cost less: 24 / 64 = 37 % extra
    vocoder work at chunk 64, 9 % at 256.  The codes live in one preallocated (C, cap) array that the frame callback fills column by
    column (no per-chunk rebuild).  stats: frames, lm_s, vocoder_busy_s, total_s, overlap_efficiency, first_audio_s (time from the call
    to the first PCM chunk)."""

    def __init__(self, lm, codec, chunk=256, halo=HALO, first_chunk=32, inline=False, stateful=None):
        """inline: vocode a chunk inside the frame callback (the LM pauses for the 3-4 ms of a 256-frame chunk) instead of in the worker
        thread.  On ONE device the persistent decode kernels hold every CU, so a concurrent vocoder only advances one kernel per LM kernel
        boundary and both sides pay for the switching; time-slicing at chunk granularity costs the vocoder's own time and no more."""
        self.lm, self.codec, self.chunk, self.halo, self.first_chunk = lm, codec, chunk, halo, min(first_chunk, chunk)
        self.inline = inline
        # stateful: the codec carries the convolutions' left context between chunks on the device (fs_codec_stream_*): no frame is decoded
        # twice; chunks shorter than 16 frames (a stream's tail) still take the halo path.  Default: whenever the codec offers it.
        # (stateful=True: required; None: used when the codec can open a stream -- the reduced test topology and the f32 mode cannot)
        self.stateful = stateful

    def __call__(self, prompt, max_new_tokens, **gen_kw):
        Cb = self.lm.cfg["num_codebooks"]
        cap = max(1, max_new_tokens - np.asarray(prompt).shape[1] + 2) + 1
        codes = np.zeros((Cb, cap), np.uint32)
        n_frames = [0]
        q, pcm_parts, errors = queue.Queue(), [], []
        t_busy, t_first = [0.0], [None]
        t0 = time.perf_counter()

        min_frames = getattr(self.codec, "STREAM_MIN_FRAMES", 16)
        stateful = bool(self.stateful) or (self.stateful is None and hasattr(self.codec, "stream_decode"))
        if stateful:
            try:
                self.codec.stream_begin()
            except Exception:
                if self.stateful:
                    raise
                stateful = False
        # A stateful stream advances the device-side left context only through stream_decode, which needs >= min_frames frames: a NON-final
        # chunk below that would have to take the stateless halo path and the next stateful chunk would start from a stale context (wrong,
        # discontinuous PCM).  So the chunk sizes of a stateful stream are raised to min_frames; only the final tail may be shorter (it is
        # decoded with a halo from the codes, and nothing follows it).
        first_chunk = max(self.first_chunk, min_frames) if stateful else self.first_chunk
        chunk = max(self.chunk, min_frames) if stateful else self.chunk

        def vocode(a, b, final=False):
            t1 = time.perf_counter()
            if stateful and b - a >= min_frames:
                pcm_parts.append(self.codec.stream_decode(np.ascontiguousarray(codes[:, a:b])))
            else:
                assert final or not stateful, "a non-final chunk of a stateful stream below the codec's minimum chunk"
                pcm_parts.append(decode_chunk(self.codec, codes, a, b, self.halo))
            t_busy[0] += time.perf_counter() - t1
            if t_first[0] is None:
                t_first[0] = time.perf_counter() - t0

        inline_upto = [0]

        def worker():
            try:
                done_upto = 0
                while True:
                    n = q.get()
                    final = n is None
                    if final:
                        if errors:
                            return
                        n = n_frames[0]
                        done_upto = max(done_upto, inline_upto[0])
                    while True:  # vocode every complete chunk available so far (the first one is shorter)
                        step = first_chunk if done_upto == 0 else chunk
                        if done_upto + step > n:
                            break
                        vocode(done_upto, done_upto + step)
                        done_upto += step
                    if final:
                        if n > done_upto:
                            vocode(done_upto, n, final=True)  # tail
                        return
            except BaseException as e:  # surfaced by the caller after join
                errors.append(e)

        th = threading.Thread(target=worker, daemon=True)
        th.start()

        def on_frame(idx, fr):
            codes[:, idx] = fr
            n_frames[0] = idx + 1
            done = idx + 1
            if done == first_chunk or (done > first_chunk and (done - first_chunk) % chunk == 0):
                if self.inline:
                    step = first_chunk if done == first_chunk else chunk
                    try:
                        vocode(done - step, done)
                        inline_upto[0] = done
                    except BaseException as e:  # (an exception must not escape a ctypes callback: record it, stop generating)
                        errors.append(e)
                else:
                    q.put(done)
            return bool(errors)  # stop generating if the vocoder thread died

        try:
            out = self.lm.generate_blocking(prompt, max_new_tokens, on_frame=on_frame, **gen_kw)
            t_lm = time.perf_counter() - t0
        finally:
            q.put(None)  # the worker always gets its sentinel, also when generate_blocking raises
            th.join()
            if stateful:
                try:
                    self.codec.stream_end()
                except Exception:
                    pass
        if errors:
            raise errors[0]
        t_all = time.perf_counter() - t0
        assert out.shape[1] == n_frames[0] and Cb == out.shape[0] and np.array_equal(out, codes[:, : n_frames[0]])
        pcm = np.concatenate(pcm_parts) if pcm_parts else np.zeros(0, np.float32)
        self.stats = dict(stateful=stateful, frames=n_frames[0], lm_s=t_lm, vocoder_busy_s=t_busy[0], total_s=t_all, first_audio_s=t_first[0],
                          overlap_efficiency=(t_lm + t_busy[0] - t_all) / max(t_busy[0], 1e-9))
        return out, pcm
