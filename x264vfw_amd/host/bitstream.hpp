// bitstream.hpp — RBSP bit writer, Exp-Golomb codes and NAL encapsulation (ITU-T H.264 7.2, 7.4.1, 9.1).
// Host side of the drop-in: plays the role of [x264-upstream] common/bitstream.h + encoder/set.c helpers.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace x264host {

class BitWriter {
public:
    void reset() { buf_.clear(); acc_ = 0; nbits_ = 0; }
    void put(uint32_t value, int n)          // n <= 32, MSB first; a 64-bit accumulator, whole bytes leave it at once
    {
        if (n <= 0) return;
        acc_ = (acc_ << n) | (uint64_t)(n == 32 ? value : value & ((1u << n) - 1u));
        nbits_ += n;
        while (nbits_ >= 8) { nbits_ -= 8; buf_.push_back((uint8_t)(acc_ >> nbits_)); }
    }
    void put1(int b) { put((uint32_t)b & 1, 1); }
    void ue(uint32_t v)                      // 9.1: codeNum v
    {
        const uint64_t x = (uint64_t)v + 1;
        const int len = 63 - __builtin_clzll(x);
        if (len) put(0, len);
        if (len + 1 > 32) { put((uint32_t)(x >> 32), len + 1 - 32); put((uint32_t)x, 32); }
        else put((uint32_t)x, len + 1);
    }
    void se(int v) { ue(v <= 0 ? (uint32_t)(-2 * (int64_t)v) : (uint32_t)(2 * (int64_t)v - 1)); }
    void te(int range, int v) { if (range == 1) put1(!v); else ue((uint32_t)v); }   // te(v) with cMax = range
    void trailing() { put1(1); align_zero(); }                                       // rbsp_trailing_bits
    void align_zero() { if (nbits_) put(0, 8 - nbits_); }
    size_t bits() const { return buf_.size() * 8 + nbits_; }
    const std::vector<uint8_t> &bytes() const { return buf_; }
    void reserve(size_t n) { buf_.reserve(n); }
    void append(const BitWriter &o)          // every bit of `o`, in order, behind the bits written so far
    {
        const uint8_t *d = o.buf_.data();
        const size_t nb = o.buf_.size();
        if (nbits_ == 0) buf_.insert(buf_.end(), d, d + nb);
        else {
            size_t i = 0;
            for (; i + 4 <= nb; i += 4) put((uint32_t)d[i] << 24 | (uint32_t)d[i + 1] << 16 | (uint32_t)d[i + 2] << 8 | d[i + 3], 32);
            for (; i < nb; i++) put(d[i], 8);
        }
        if (o.nbits_) put((uint32_t)(o.acc_ & ((1u << o.nbits_) - 1u)), o.nbits_);
    }
private:
    std::vector<uint8_t> buf_;
    uint64_t acc_ = 0;                       // only the low nbits_ (< 8 between calls) bits are pending
    int nbits_ = 0;
};

// Append one NAL unit (header + RBSP with emulation prevention, 7.4.1) to `out`.
// annexb: 4-byte start code for the first NAL of an access unit / parameter sets, else 3-byte;
// !annexb: 4-byte big-endian length prefix (mp4/mkv/flv muxers, codec.c:1121-1143).
inline void append_nal(std::vector<uint8_t> &out, int nal_ref_idc, int nal_unit_type, const std::vector<uint8_t> &rbsp,
                       bool annexb, bool long_startcode)
{
    size_t start = out.size();
    if (annexb) {
        if (long_startcode) out.push_back(0);
        out.push_back(0); out.push_back(0); out.push_back(1);
    } else
        out.insert(out.end(), 4, 0);
    size_t payload0 = out.size();
    out.push_back((uint8_t)((nal_ref_idc << 5) | nal_unit_type));
    int zeros = 0;
    for (uint8_t b : rbsp) {
        if (zeros >= 2 && b <= 3) { out.push_back(3); zeros = 0; }
        out.push_back(b);
        zeros = b == 0 ? zeros + 1 : 0;
    }
    if (!annexb) {
        uint32_t n = (uint32_t)(out.size() - payload0);
        out[start] = (uint8_t)(n >> 24); out[start + 1] = (uint8_t)(n >> 16); out[start + 2] = (uint8_t)(n >> 8); out[start + 3] = (uint8_t)n;
    }
}

}  // namespace x264host
