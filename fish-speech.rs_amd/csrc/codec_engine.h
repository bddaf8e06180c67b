// Host-side engine of the Firefly-GAN-VQ vocoder (FireflyCodec::decode, codec/firefly.rs:42-48).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace fs {

class CodecBase {
  public:
    virtual ~CodecBase() {}
    virtual void load_synthetic(uint64_t seed) = 0;
    virtual void load_safetensors(const std::string& path) = 0;
    virtual void decode(const uint32_t* codes, int b, int T, float* pcm_out) = 0;
    // stateful streaming: consecutive chunks of ONE code sequence; concatenated PCM == decode of the whole sequence, bit for bit
    virtual void stream_begin() = 0;
    virtual void stream_decode(const uint32_t* codes, int T, float* pcm_out) = 0;
    virtual void stream_end() = 0;
    virtual void encode(const float* pcm, int n, uint32_t* codes_out, size_t cap, size_t* L_out) = 0;
    virtual int sample_rate() = 0;
    virtual void set_precision(int mode) = 0;  // 0 = f32 (exact f32 products), 1 = bf16x3, 2 = f16 (default); decode only
    virtual int precision() = 0;
    virtual void set_range_check(bool on) = 0;      // fs_codec_set_range_check
    virtual void range_stats(uint64_t* out5, double* last_rms) = 0;   // fs_codec_range_stats
};

CodecBase* make_codec(int device, int channel_div);

}  // namespace fs
