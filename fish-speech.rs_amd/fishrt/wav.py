"""16-bit mono PCM WAV writer with the reference's sample conversion (fish_speech_core/lib/audio/wav.rs:27-58):
i16 = trunc(clamp(x, -1, 1) * 32767), 44-byte canonical header."""
import struct

import numpy as np


def pcm_to_i16(samples):
    x = np.clip(np.asarray(samples, np.float32), np.float32(-1.0), np.float32(1.0)) * np.float32(32767.0)
    return np.trunc(x).astype("<i2")  # Rust `as i16` truncates toward zero


def write_pcm_as_wav(fileobj_or_path, samples, sample_rate):
    s = np.asarray(samples)
    data = (s.astype("<i2") if s.dtype == np.int16 else pcm_to_i16(s.reshape(-1))).tobytes()
    n_channels, total = 1, 12 + 24 + len(data) + 8
    hdr = b"RIFF" + struct.pack("<I", total - 8) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, n_channels, int(sample_rate),
                                                                                 int(sample_rate) * 2 * n_channels, 2, 16)
    blob = hdr + b"data" + struct.pack("<I", len(data)) + data
    if hasattr(fileobj_or_path, "write"):
        fileobj_or_path.write(blob)
    else:
        with open(fileobj_or_path, "wb") as f:
            f.write(blob)
    return len(blob)
