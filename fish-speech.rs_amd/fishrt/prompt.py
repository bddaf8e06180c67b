"""Prompt / voice formats on either side of the LM hot path: the (num_codebooks + 1, L) u32 prompt layout
of `PromptEncoder` (fish_speech_core/lib/text/prompt.rs:28-156), `.npy` voice prompts (prompt.rs:159-198) and the voice
directory with its `index.json` (server/lib/utils/mod.rs:17-55).  Host-side only; the tokenizer stays with the caller
(any object with `encode(text) -> ids` and `token_to_id(token) -> id | None`, e.g. `tokenizers.Tokenizer` wrapped by
`HFTokenizer` below)."""
import json
import os

import numpy as np

FISH_1_5 = "fish1.5"   # WhichLM::Fish(Fish1_5) and WhichLM::DualAR take the semantic-range branch (prompt.rs:69-73,84-87)
FISH_1_4 = "fish1.4"   # Fish <= 1.4: single <|semantic|> id, codes shifted by +1 (prompt.rs:75-78,88-91)


class HFTokenizer:
    """Adapter for `tokenizers.Tokenizer`: encode without added special tokens (prompt.rs:31-34)."""

    def __init__(self, tok):
        self.tok = tok

    def encode(self, text):
        return list(self.tok.encode(text, add_special_tokens=False).ids)

    def token_to_id(self, token):
        return self.tok.token_to_id(token)


class PromptEncoder:
    def __init__(self, tokenizer, num_codebooks=8, model_type=FISH_1_5):
        self.tokenizer, self.num_codebooks, self.model_type = tokenizer, int(num_codebooks), model_type

    # prompt.rs:28-42: row 0 = token ids, rows 1..num_codebooks = 0
    def tokenize_text(self, text):
        try:
            ids = self.tokenizer.encode(text)
        except Exception as e:  # candle: Error::Msg("Could not tokenize: ...")
            raise RuntimeError(f"Could not tokenize: {e!r}")
        out = np.zeros((self.num_codebooks + 1, len(ids)), np.uint32)
        out[0] = np.asarray(ids, np.uint32)
        return out

    # prompt.rs:44-51
    def encode_text(self, role, content=None):
        s = f"<|im_start|>{role}\n{content}<|im_end|>" if content is not None else f"<|im_start|>{role}\n"
        return self.tokenize_text(s)

    # prompt.rs:53-95
    def encode_vq(self, prompt_tokens=None):
        prefix = self.tokenize_text("<|im_start|>assistant\n" + ("<|voice|>" if self.model_type == FISH_1_5 else ""))
        if prompt_tokens is None:
            return prefix
        suffix = self.tokenize_text("<|im_end|>")
        pt = np.asarray(prompt_tokens)
        if pt.ndim != 2:
            raise RuntimeError(f"unexpected rank, expected: 2, got: {pt.ndim}")  # dims2()
        pt = pt.astype(np.uint32)
        seqlen = pt.shape[1]
        if self.model_type == FISH_1_5:
            start = self.tokenizer.token_to_id("<|semantic:0|>")
            if start is None:
                raise RuntimeError("tokenizer has no <|semantic:0|> token")  # the reference unwrap()s
            sem = (np.float64(start) + pt[0].astype(np.float64)).astype(np.uint32)  # `semantic_start as f64 + codes[0]`
            span = np.concatenate([sem[None], pt], 0)
        else:
            sid = self.tokenizer.token_to_id("<|semantic|>")
            sem = np.full((1, seqlen), 5 if sid is None else sid, np.uint32)   # unwrap_or(5)
            span = np.concatenate([sem, pt + np.uint32(1)], 0)                  # codes + 1 for Fish <= 1.4
        if span.shape[0] != self.num_codebooks + 1:
            raise RuntimeError(f"shape mismatch in cat for dim 0: expected {self.num_codebooks + 1} rows, got {span.shape[0]}")
        return np.concatenate([prefix, span, suffix], 1)

    # prompt.rs:97-106
    def encode_conditioning_prompt(self, prompt_text, prompt_tensor):
        return np.concatenate([self.encode_text("user", prompt_text), self.encode_vq(prompt_tensor)], 1)

    # prompt.rs:108-156 -> (num_conditioning_tokens, [prompt per chunk])
    def encode_sequence(self, chunks, sysprompt_text=None, cached_speaker=None, assume_kv_cache=False):
        if len(chunks) == 0:
            raise RuntimeError("Input text cannot be empty")
        sysprompt = self.encode_text("system", sysprompt_text) if sysprompt_text is not None else None
        n_cond = (sysprompt.shape[1] if sysprompt is not None else 0) + (cached_speaker.shape[1] if cached_speaker is not None else 0)
        parts = [p for p in (sysprompt, cached_speaker) if p is not None]
        cond = np.concatenate(parts, 1) if parts else None
        assistant_start = self.encode_vq(None)
        out = []
        for i, chunk in enumerate(chunks):
            prompt = []
            if cond is not None and (i == 0 or not assume_kv_cache):
                prompt.append(cond)
            prompt.append(self.encode_text("user", chunk))
            prompt.append(assistant_start)
            out.append(np.concatenate(prompt, 1))
        return n_cond, out


def load_prompt_text(path, num_codebooks=8):
    """prompt.rs:159-198: (num_codebooks, T) or (1, num_codebooks, T) .npy of any integer dtype -> u32 (num_codebooks, T)."""
    a = np.load(path)
    a = a.astype(np.uint32)
    if a.ndim == 2:
        if a.shape[0] == num_codebooks:
            return a
        raise RuntimeError(f"Expected {num_codebooks} codebooks but got {a.shape[0]}")
    if a.ndim == 3 and a.shape[0] == 1:
        if a.shape[1] == num_codebooks:
            return a[0]
        raise RuntimeError(f"Expected {num_codebooks} codebooks but got {a.shape[1]}")
    raise RuntimeError(f"Incorrect prompt token dimensions for {path!r}: {a.ndim}")


def load_speaker_prompts(voice_dir, tokenizer, num_codebooks=8, model_type=FISH_1_5):
    """server/lib/utils/mod.rs:17-55: {name: conditioning prompt}, default prompt.  index.json: {"speakers": {name: text}}."""
    try:
        with open(os.path.join(voice_dir, "index.json")) as f:
            index = json.load(f)
    except OSError:
        raise RuntimeError("Failed to open speaker index.json")
    enc = PromptEncoder(tokenizer, num_codebooks, model_type)
    speakers, default = {}, None
    for name, text in index["speakers"].items():
        codes = load_prompt_text(os.path.join(voice_dir, f"{name}.npy"), num_codebooks)
        prompt = enc.encode_conditioning_prompt(text, codes)
        if name == "default":
            default = prompt
        speakers[name] = prompt
    if default is None:
        raise RuntimeError("No default speaker found in index.json and voices directory")
    return speakers, default
