#!/usr/bin/env python3
"""Second half of tools/make_reference_dumps.sh: turn what the reference's `llama_generate` printed into the files the pin test reads.
  reference_dumps_meta.py OUT_DIR CHECKPOINT TEXT FISH_VERSION MAX_NEW_TOKENS
Reads OUT_DIR/llama_generate.log (the binary prints "Input tokens:\\n[..]" = row 0 of its prompt, fish_speech_core/src/bin/llama_generate.rs:82-88) and
OUT_DIR/ref_codes.npy; writes OUT_DIR/prompt.npy ((9, L) u32: a text-only prompt has zeros in the codebook rows, text/prompt.rs:53-104) and OUT_DIR/meta.json.
(tests/test_prompt_formats.py runs this on a fabricated log so that the one-command pin cannot rot.)"""
import json
import re
import sys

import numpy as np


def parse_input_tokens(log_text):
    m = re.search(r"Input tokens:\s*\[([0-9,\s]+)\]", log_text)
    if not m:
        raise ValueError("llama_generate did not print its input tokens")
    return np.array([int(t) for t in m.group(1).replace("\n", " ").split(",") if t.strip()], np.uint32)


def write_meta(out, ckpt, text, ver, maxnew, num_codebooks=8):
    row0 = parse_input_tokens(open(f"{out}/llama_generate.log").read())
    prompt = np.zeros((num_codebooks + 1, row0.size), np.uint32)
    prompt[0] = row0
    np.save(f"{out}/prompt.npy", prompt)
    codes = np.load(f"{out}/ref_codes.npy")
    assert codes.ndim == 2 and codes.shape[0] == num_codebooks, f"expected ({num_codebooks}, n) codes from llama_generate, got {codes.shape}"
    meta = {"checkpoint": ckpt, "text": text, "fish_version": ver, "max_new_tokens": int(maxnew), "repetition_penalty": 1.2,
            "frames": int(codes.shape[-1]), "prompt_positions": int(row0.size),
            "made_by": "tools/make_reference_dumps.sh (reference binaries llama_generate --temp 0, vocoder; CPU f32)"}
    with open(f"{out}/meta.json", "w") as f:
        json.dump(meta, f, indent=1)
    return meta, prompt


if __name__ == "__main__":
    out, ckpt, text, ver, maxnew = sys.argv[1:6]
    meta, prompt = write_meta(out, ckpt, text, ver, maxnew)
    print(f"wrote {out}: codes (8, {meta['frames']}), prompt {prompt.shape}")
