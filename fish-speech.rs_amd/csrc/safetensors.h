// Minimal safetensors reader (header JSON + mmap), enough for the reference's checkpoints
// (`model.safetensors`, fish_speech_python/src/lm.rs:41-56; safetensors 0.4.5 format: u64 LE header length,
// JSON object name -> {dtype, shape, data_offsets}, raw little-endian data).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <climits>
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

// read-only mapping of a file; unmaps / closes on destruction, so a throwing SafeTensors constructor leaks nothing
struct MappedFile {
    int fd = -1;
    const uint8_t* base = nullptr;
    size_t size = 0;
    explicit MappedFile(const std::string& path) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error("cannot open " + path);
        struct stat sb;
        if (fstat(fd, &sb) != 0) { close(fd); fd = -1; throw Error("cannot stat " + path); }
        size = (size_t)sb.st_size;
        if (size < 8) { close(fd); fd = -1; throw Error("safetensors file too small: " + path); }
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); fd = -1; throw Error("mmap failed: " + path); }
        base = (const uint8_t*)m;
    }
    ~MappedFile() {
        if (base) munmap((void*)base, size);
        if (fd >= 0) close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
};

inline size_t st_elem_size(const std::string& dtype) {
    if (dtype == "F32") return 4;
    if (dtype == "BF16" || dtype == "F16") return 2;
    throw Error("unsupported safetensors dtype " + dtype);
}

class SafeTensors {
  public:
    explicit SafeTensors(const std::string& path) : f_(path) {
        uint64_t hlen;
        std::memcpy(&hlen, f_.base, 8);
        if (hlen > f_.size - 8) throw Error("corrupt safetensors header: " + path);  // (no 8 + hlen: it wraps for a huge hlen)
        parse(std::string((const char*)f_.base + 8, (size_t)hlen), f_.base + 8 + hlen, f_.size - 8 - (size_t)hlen);
    }
    const StTensor* find(const std::string& name) const {
        auto it = t_.find(name);
        return it == t_.end() ? nullptr : &it->second;
    }
    // what candle's VarBuilder::get checks: the tensor exists with exactly this shape
    const StTensor& get(const std::string& name, const std::vector<int64_t>& shape) const {
        const StTensor* t = find(name);
        if (!t) throw Error("cannot find tensor " + name);
        if (t->shape != shape) {
            auto fmt = [](const std::vector<int64_t>& v) { std::string r = "["; for (size_t i = 0; i < v.size(); ++i) r += (i ? ", " : "") + std::to_string(v[i]); return r + "]"; };
            throw Error("shape mismatch for " + name + ": expected " + fmt(shape) + ", got " + fmt(t->shape));
        }
        return *t;
    }
    // all elements as f32 (F32 / BF16 / F16); the byte count was checked against shape x dtype when the header was parsed
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
    // tiny JSON scanner for the flat header layout; every character access is bounds-checked (a truncated header is an error, not a read
    // past the string)
    void parse(const std::string& h, const uint8_t* data, size_t data_len) {
        size_t i = 0;
        auto at = [&](size_t k) -> char { if (k >= h.size()) throw Error("safetensors header: truncated"); return h[k]; };
        auto ws = [&]() { while (i < h.size() && (h[i] == ' ' || h[i] == '\n' || h[i] == '\t' || h[i] == '\r')) ++i; };
        auto expect = [&](char c) { ws(); if (at(i) != c) throw Error(std::string("safetensors header: expected '") + c + "'"); ++i; };
        auto str = [&]() {
            ws();
            if (at(i) != '"') throw Error("safetensors header: expected string");
            std::string s;
            for (++i; at(i) != '"'; ++i) { if (h[i] == '\\') { ++i; (void)at(i); } s.push_back(h[i]); }
            ++i;
            return s;
        };
        auto num = [&]() {
            ws();
            if (at(i) < '0' || h[i] > '9') throw Error("safetensors header: expected a non-negative integer");
            int64_t v = 0;
            while (i < h.size() && h[i] >= '0' && h[i] <= '9') {
                if (v > (INT64_MAX - 9) / 10) throw Error("safetensors header: integer overflow");
                v = v * 10 + (h[i++] - '0');
            }
            return v;
        };
        int depth = 0;
        std::function<void()> skip = [&]() {  // skip any JSON value
            ws();
            if (at(i) == '"') { str(); return; }
            if (h[i] == '{' || h[i] == '[') {
                if (++depth > 64) throw Error("safetensors header: nesting too deep");
                const char open = h[i], close = open == '{' ? '}' : ']';
                ++i;
                for (ws(); at(i) != close; ws()) { if (open == '{') { str(); expect(':'); } skip(); ws(); if (at(i) == ',') ++i; }
                ++i;
                --depth;
                return;
            }
            while (i < h.size() && h[i] != ',' && h[i] != '}' && h[i] != ']') ++i;
        };
        expect('{');
        for (ws(); at(i) != '}'; ws()) {
            const std::string name = str();
            expect(':');
            if (name == "__metadata__") { skip(); ws(); if (at(i) == ',') ++i; continue; }
            StTensor t;
            int64_t b = 0, e = 0;
            bool have_off = false;
            expect('{');
            for (ws(); at(i) != '}'; ws()) {
                const std::string key = str();
                expect(':');
                if (key == "dtype") t.dtype = str();
                else if (key == "shape") { expect('['); for (ws(); at(i) != ']'; ws()) { t.shape.push_back(num()); ws(); if (at(i) == ',') ++i; } ++i; }
                else if (key == "data_offsets") { expect('['); b = num(); expect(','); e = num(); expect(']'); have_off = true; }
                else skip();
                ws();
                if (at(i) == ',') ++i;
            }
            ++i;
            if (!have_off || e < b || (uint64_t)e > (uint64_t)data_len) throw Error("safetensors: bad offsets for " + name);
            // the byte range must be exactly shape x dtype (a short range would be read past; safetensors itself rejects such files)
            uint64_t numel = 1;
            for (int64_t d : t.shape) {
                if (d < 0 || (d > 0 && numel > UINT64_MAX / (uint64_t)d)) throw Error("safetensors: bad shape for " + name);
                numel *= (uint64_t)d;
            }
            const uint64_t es = st_elem_size(t.dtype);
            if (numel > UINT64_MAX / es || numel * es != (uint64_t)(e - b))
                throw Error("safetensors: " + name + " has " + std::to_string(e - b) + " bytes, its shape and dtype need " + std::to_string(numel * es));
            t.data = data + b;
            t.nbytes = (size_t)(e - b);
            t_[name] = t;
            ws();
            if (i < h.size() && h[i] == ',') ++i;
        }
    }
    MappedFile f_;
    std::map<std::string, StTensor> t_;
};

}  // namespace fs
