// Firefly vocoder engine -- placeholder until the HIP kernels land (next milestone); fails loudly, never falls back.
#include "codec_engine.h"

#include "fs_common.h"

namespace fs {
CodecBase* make_codec(int, int) { throw Error("fs_codec: HIP vocoder kernels not built into this libfishrt yet"); }
}  // namespace fs
