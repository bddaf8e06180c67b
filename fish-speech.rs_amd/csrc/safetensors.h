// Minimal safetensors reader (header JSON + mmap), enough for the reference's checkpoints
// (`model.safetensors`, fish_speech_python/src/lm.rs:41-56; safetensors 0.4.5 format: u64 LE header length,
// JSON object name -> {dtype, shape, data_offsets}, raw little-endian data).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "fs_common.h"

namespace fs {

struct StTensor {
    std::string dtype;
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

class SafeTensors {
  public:
    explicit SafeTensors(const std::string& path) {
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw Error("cannot open " + path);
        struct stat sb;
        if (fstat(fd_, &sb) != 0) throw Error("cannot stat " + path);
        size_ = (size_t)sb.st_size;
        if (size_ < 8) throw Error("safetensors file too small: " + path);
        base_ = (const uint8_t*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (base_ == MAP_FAILED) throw Error("mmap failed: " + path);
        uint64_t hlen;
        std::memcpy(&hlen, base_, 8);
        if (8 + hlen > size_) throw Error("corrupt safetensors header: " + path);
        parse(std::string((const char*)base_ + 8, (size_t)hlen), base_ + 8 + hlen, size_ - 8 - hlen);
    }
    ~SafeTensors() {
        if (base_ && base_ != MAP_FAILED) munmap((void*)base_, size_);
        if (fd_ >= 0) close(fd_);
    }
    const StTensor* find(const std::string& name) const {
        auto it = t_.find(name);
        return it == t_.end() ? nullptr : &it->second;
    }
    // element `i` as f32 (F32 / BF16 / F16 supported)
    static void to_f32(const StTensor& t, float* dst) {
        const int64_t n = t.numel();
        if (t.dtype == "F32") {
            std::memcpy(dst, t.data, (size_t)n * 4);
        } else if (t.dtype == "BF16") {
            const uint16_t* s = (const uint16_t*)t.data;
            for (int64_t i = 0; i < n; ++i) dst[i] = bf16_to_f32_host(s[i]);
        } else if (t.dtype == "F16") {
            const uint16_t* s = (const uint16_t*)t.data;
            for (int64_t i = 0; i < n; ++i) {
                const uint32_t h = s[i], sign = (h & 0x8000u) << 16;
                uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FF, u;
                if (e == 0) {
                    if (m == 0) u = sign;
                    else { e = 127 - 15 + 1; while (!(m & 0x400)) { m <<= 1; --e; } u = sign | (e << 23) | ((m & 0x3FF) << 13); }
                } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
                else u = sign | ((e + 127 - 15) << 23) | (m << 13);
                std::memcpy(&dst[i], &u, 4);
            }
        } else {
            throw Error("unsupported safetensors dtype " + t.dtype);
        }
    }

  private:
    // tiny JSON scanner for the flat header layout
    void parse(const std::string& h, const uint8_t* data, size_t data_len) {
        size_t i = 0;
        auto ws = [&]() { while (i < h.size() && (h[i] == ' ' || h[i] == '\n' || h[i] == '\t' || h[i] == '\r')) ++i; };
        auto expect = [&](char c) { ws(); if (i >= h.size() || h[i] != c) throw Error(std::string("safetensors header: expected '") + c + "'"); ++i; };
        auto str = [&]() {
            ws();
            if (h[i] != '"') throw Error("safetensors header: expected string");
            std::string s;
            for (++i; i < h.size() && h[i] != '"'; ++i) { if (h[i] == '\\' && i + 1 < h.size()) ++i; s.push_back(h[i]); }
            ++i;
            return s;
        };
        auto num = [&]() { ws(); int64_t v = 0; while (i < h.size() && h[i] >= '0' && h[i] <= '9') v = v * 10 + (h[i++] - '0'); return v; };
        std::function<void()> skip = [&]() {  // skip any JSON value
            ws();
            if (h[i] == '"') { str(); return; }
            if (h[i] == '{' || h[i] == '[') {
                const char open = h[i], close = open == '{' ? '}' : ']';
                ++i;
                for (ws(); h[i] != close; ws()) { if (open == '{') { str(); expect(':'); } skip(); ws(); if (h[i] == ',') ++i; }
                ++i;
                return;
            }
            while (i < h.size() && h[i] != ',' && h[i] != '}' && h[i] != ']') ++i;
        };
        expect('{');
        for (ws(); i < h.size() && h[i] != '}'; ws()) {
            const std::string name = str();
            expect(':');
            if (name == "__metadata__") { skip(); ws(); if (h[i] == ',') ++i; continue; }
            StTensor t;
            int64_t b = 0, e = 0;
            expect('{');
            for (ws(); h[i] != '}'; ws()) {
                const std::string key = str();
                expect(':');
                if (key == "dtype") t.dtype = str();
                else if (key == "shape") { expect('['); for (ws(); h[i] != ']'; ws()) { t.shape.push_back(num()); ws(); if (h[i] == ',') ++i; } ++i; }
                else if (key == "data_offsets") { expect('['); b = num(); expect(','); e = num(); expect(']'); }
                else skip();
                ws();
                if (h[i] == ',') ++i;
            }
            ++i;
            if (e < b || (size_t)e > data_len) throw Error("safetensors: bad offsets for " + name);
            t.data = data + b;
            t.nbytes = (size_t)(e - b);
            t_[name] = t;
            ws();
            if (i < h.size() && h[i] == ',') ++i;
        }
    }
    int fd_ = -1;
    const uint8_t* base_ = nullptr;
    size_t size_ = 0;
    std::map<std::string, StTensor> t_;
};

}  // namespace fs
