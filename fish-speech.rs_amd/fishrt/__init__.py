"""fishrt: MI355X-native Fish-Speech hot path (dual-AR decode + Firefly vocoder) behind the C ABI of include/fishrt.h.
Python here is only the host-side mirror of the reference's PyO3 surface; all compute is hand-written HIP in libfishrt.so."""
from . import config
from ._ffi import LIB_PATH, SYMBOLS, lib
from .codec import FireflyCodec
from .lm import LM, DualARTransformer
from .stream import StreamingSynth, decode_chunk

__all__ = ["config", "lib", "LIB_PATH", "SYMBOLS", "DualARTransformer", "LM", "FireflyCodec", "StreamingSynth", "decode_chunk"]
