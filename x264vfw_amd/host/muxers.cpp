// muxers.cpp — file output behind `--output`: what the reference's cli_output_t implementations write
// (output/raw.c, output/matroska.c + matroska_ebml.c, output/flv.c + flv_bytestream.c; selected by select_output, codec.c:1111-1164),
// re-designed as three small classes over one interface.  Container semantics follow the reference file by file:
//   raw : parameter sets, then every frame's NAL bytes as they come (Annex-B)                       (output/raw.c:41-55)
//   mkv : EBML header, Segment of unknown size, SegmentInfo (TimecodeScale 50000 ns, Duration patched at close), one
//         V_MPEG4/ISO/AVC track whose CodecPrivate is the avcC record built from SPS/PPS, frames as SimpleBlocks with 16-bit
//         cluster-relative timecodes, a new Cluster on timecode overflow or past 1 MiB; the SEI of the headers is prepended to
//         the first frame                                                        (matroska.c:117-218, matroska_ebml.c:317-506)
//   flv : 13-byte file header, onMetaData script tag (7 entries; duration / filesize / videodatarate patched at close), AVC
//         sequence-header tag (avcC), one video tag per frame with millisecond DTS and CTS offset    (flv.c:63-356)
//   mp4 : what the reference gets from L-SMASH for a regular file (mp4_lsmash.c:193-446), written directly: ftyp (mp42 / mp41 / isom),
//         one progressively written mdat, moov at close — mvhd (timescale 600), one video trak with an explicit edit (elst:
//         presentation duration, media_time = first CTS), mdhd timescale = timebase_den, avc1 sample entry with avcC / colr (nclx) /
//         pasp / btrt, stts / ctts / stss / stsc (chunks of about half a second) / stsz / stco|co64; DTS and CTS are
//         (dts|pts + start_offset) * timebase_num, the last sample's duration is largest_pts - second_largest_pts, and the SEI
//         of the headers goes in front of the first sample.  Checked by the reference tree's own L-SMASH as demuxer (tests).
// avi (libavformat) stays "not compiled in", as in a reference build without that library.
#include "host.hpp"
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace x264host {

namespace {

struct Bytes {
    std::vector<uint8_t> d;
    void u8(unsigned v) { d.push_back((uint8_t)v); }
    void be(uint64_t v, int n) { for (int i = n - 1; i >= 0; i--) d.push_back((uint8_t)(v >> (8 * i))); }
    void raw(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; d.insert(d.end(), b, b + n); }
    void str(const char *s) { raw(s, strlen(s)); }
    size_t size() const { return d.size(); }
};

// avcC (ISO/IEC 14496-15 AVCDecoderConfigurationRecord) from 4-byte-prefixed SPS / PPS NALs, 4-byte NAL lengths
void put_avcc(Bytes &b, const uint8_t *sps, int sps_size, const uint8_t *pps, int pps_size)
{
    b.u8(1); b.u8(sps[1]); b.u8(sps[2]); b.u8(sps[3]);
    b.u8(0xff);                       // reserved 111111 + lengthSizeMinusOne 3
    b.u8(0xe1);                       // reserved 111 + one SPS
    b.be((uint64_t)sps_size, 2); b.raw(sps, (size_t)sps_size);
    b.u8(1);                          // one PPS
    b.be((uint64_t)pps_size, 2); b.raw(pps, (size_t)pps_size);
}

// ------------------------------------------------------------------ raw ------------------------------------------------------------------
class RawMuxer : public Muxer {
public:
    explicit RawMuxer(FILE *f) : fp_(f) {}
    int set_param(const x264_param_t *) override { return 0; }
    int write_headers(const x264_nal_t *nal) override
    {
        const int n = nal[0].i_payload + nal[1].i_payload + nal[2].i_payload;          // SPS, PPS, SEI are contiguous (codec.c:1650)
        return fwrite(nal[0].p_payload, (size_t)n, 1, fp_) == 1 ? n : -1;
    }
    int write_frame(const uint8_t *p, int size, const x264_picture_t *) override { return fwrite(p, (size_t)size, 1, fp_) == 1 ? size : -1; }
    int close(int64_t, int64_t) override { const int rc = fclose(fp_); fp_ = nullptr; return rc; }
    ~RawMuxer() override { if (fp_) fclose(fp_); }
private:
    FILE *fp_;
};

// ------------------------------------------------------------------ mkv ------------------------------------------------------------------
struct Ebml : Bytes {
    void id(uint32_t v) { int n = v >= 0x1000000 ? 4 : v >= 0x10000 ? 3 : v >= 0x100 ? 2 : 1; be(v, n); }
    void size_of(uint64_t v)                                  // shortest EBML size that is not the reserved all-ones pattern
    {
        int n = 1;
        while (n < 8 && v >= ((uint64_t)1 << (7 * n)) - 1) n++;
        be(v | ((uint64_t)1 << (7 * n)), n);
    }
    void uint_el(uint32_t i, uint64_t v) { int n = 1; while (n < 8 && (v >> (8 * n))) n++; id(i); size_of((uint64_t)n); be(v, n); }
    void str_el(uint32_t i, const char *s) { id(i); size_of(strlen(s)); str(s); }
    void bin_el(uint32_t i, const void *p, size_t n) { id(i); size_of(n); raw(p, n); }
    void float_el(uint32_t i, float f) { uint32_t u; memcpy(&u, &f, 4); id(i); size_of(4); be(u, 4); }
    void master(uint32_t i, const Ebml &c) { id(i); size_of(c.size()); raw(c.d.data(), c.size()); }
};

class MkvMuxer : public Muxer {
public:
    explicit MkvMuxer(FILE *f) : fp_(f) {}
    int set_param(const x264_param_t *p) override
    {
        frame_duration_ = p->i_fps_num > 0 && !p->b_vfr_input ? (int64_t)p->i_fps_den * 1000000000ll / p->i_fps_num : 0;
        int64_t dw = width_ = p->i_width, dh = height_ = p->i_height;
        if (p->vui.i_sar_width && p->vui.i_sar_height && p->vui.i_sar_width != p->vui.i_sar_height) {
            if (p->vui.i_sar_width > p->vui.i_sar_height) dw = dw * p->vui.i_sar_width / p->vui.i_sar_height;
            else dh = dh * p->vui.i_sar_height / p->vui.i_sar_width;
        }
        d_width_ = (int)dw; d_height_ = (int)dh;
        tb_num_ = p->i_timebase_num; tb_den_ = p->i_timebase_den;
        return 0;
    }
    int write_headers(const x264_nal_t *nal) override
    {
        if (!width_ || !height_ || !d_width_ || !d_height_ || wrote_header_) return -1;
        const int sps_size = nal[0].i_payload - 4, pps_size = nal[1].i_payload - 4, sei_size = nal[2].i_payload;
        Bytes avcc;
        put_avcc(avcc, nal[0].p_payload + 4, sps_size, nal[1].p_payload + 4, pps_size);
        Ebml out, c;
        c.uint_el(0x4286, 1); c.uint_el(0x42f7, 1); c.uint_el(0x42f2, 4); c.uint_el(0x42f3, 8);       // EBMLVersion, ReadVersion, MaxIDLength, MaxSizeLength
        c.str_el(0x4282, "matroska"); c.uint_el(0x4287, 2); c.uint_el(0x4285, 2);                     // DocType, DocTypeVersion, DocTypeReadVersion
        out.master(0x1a45dfa3, c);
        out.id(0x18538067); out.be(0x01ffffffffffffffull, 8);                                         // Segment, size unknown (streamed)
        Ebml info;
        info.str_el(0x4d80, "x264vfw-mi355x matroska writer"); info.str_el(0x5741, "x264vfw-mi355x r1");  // MuxingApp, WritingApp
        info.uint_el(0x2ad7b1, kTimescale);                                                           // TimecodeScale
        info.float_el(0x4489, 0.f);                                                                   // Duration: patched at close
        const size_t dur_in_info = info.size() - 4;
        out.id(0x1549a966); out.size_of(info.size());
        duration_pos_ = (long)(out.size() + dur_in_info);
        out.raw(info.d.data(), info.size());
        Ebml video, track, tracks;
        video.uint_el(0xb0, (uint64_t)width_); video.uint_el(0xba, (uint64_t)height_);                // PixelWidth, PixelHeight
        video.uint_el(0x54b2, 0); video.uint_el(0x54b0, (uint64_t)d_width_); video.uint_el(0x54ba, (uint64_t)d_height_);   // DisplayUnit pixels, DisplayWidth/Height
        track.uint_el(0xd7, 1); track.uint_el(0x73c5, 1); track.uint_el(0x83, 1); track.uint_el(0x9c, 0);   // TrackNumber, TrackUID, TrackType video, FlagLacing
        track.str_el(0x86, "V_MPEG4/ISO/AVC");                                                        // CodecID
        track.bin_el(0x63a2, avcc.d.data(), avcc.size());                                             // CodecPrivate
        if (frame_duration_) track.uint_el(0x23e383, (uint64_t)frame_duration_);                      // DefaultDuration
        track.master(0xe0, video);
        tracks.master(0xae, track);
        out.master(0x1654ae6b, tracks);
        if (fwrite(out.d.data(), out.size(), 1, fp_) != 1) return -1;
        wrote_header_ = true;
        frame_.raw(nal[2].p_payload, (size_t)sei_size);                                               // the SEI travels with the first frame
        return sei_size + sps_size + pps_size;
    }
    int write_frame(const uint8_t *p, int size, const x264_picture_t *pic) override
    {
        frame_.raw(p, (size_t)size);
        const int64_t stamp = (int64_t)((double)pic->i_pts * 1e9 * tb_num_ / tb_den_ + 0.5);          // ns
        if (stamp > max_tc_) max_tc_ = stamp;
        int64_t delta = stamp / kTimescale - cluster_tc_;
        if (have_cluster_ && (delta > 32767 || delta < -32768)) { if (flush_cluster() < 0) return -1; }
        if (!have_cluster_) {
            cluster_tc_ = stamp / kTimescale;
            cluster_.d.clear();
            cluster_.uint_el(0xe7, (uint64_t)cluster_tc_);                                            // Timecode
            have_cluster_ = true;
            delta = 0;
        }
        cluster_.id(0xa3); cluster_.size_of(frame_.size() + 4);                                      // SimpleBlock
        cluster_.size_of(1);                                                                          // track number
        cluster_.be((uint64_t)(delta & 0xffff), 2);
        cluster_.u8((pic->b_keyframe ? 0x80u : 0u) | (pic->i_type == X264_TYPE_B ? 1u : 0u));         // keyframe / discardable
        cluster_.raw(frame_.d.data(), frame_.size());
        frame_.d.clear();
        if (cluster_.size() > (1u << 20) && flush_cluster() < 0) return -1;
        return size;
    }
    int close(int64_t largest_pts, int64_t second_largest_pts) override
    {
        int ret = flush_cluster();
        const int64_t last_delta = tb_den_ ? (int64_t)((double)(largest_pts - second_largest_pts) * 1e9 * tb_num_ / tb_den_ + 0.5) : 0;
        if (wrote_header_) {
            const int64_t total = max_tc_ + (frame_duration_ ? frame_duration_ : last_delta);
            Ebml f;
            float v = (float)((double)total / kTimescale);
            uint32_t u; memcpy(&u, &v, 4); f.be(u, 4);
            if (fseek(fp_, duration_pos_, SEEK_SET) || fwrite(f.d.data(), 4, 1, fp_) != 1) ret = -1;
        }
        if (fclose(fp_)) ret = -1;
        fp_ = nullptr;
        return ret;
    }
    ~MkvMuxer() override { if (fp_) fclose(fp_); }
private:
    int flush_cluster()
    {
        if (!have_cluster_) return 0;
        Ebml out;
        out.master(0x1f43b675, cluster_);
        have_cluster_ = false;
        return fwrite(out.d.data(), out.size(), 1, fp_) == 1 ? 0 : -1;
    }
    static constexpr int64_t kTimescale = 50000;
    FILE *fp_;
    int width_ = 0, height_ = 0, d_width_ = 0, d_height_ = 0;
    int64_t frame_duration_ = 0, max_tc_ = 0, cluster_tc_ = 0;
    uint32_t tb_num_ = 1, tb_den_ = 25;
    long duration_pos_ = 0;
    bool wrote_header_ = false, have_cluster_ = false;
    Ebml cluster_;
    Bytes frame_;
};

// ------------------------------------------------------------------ flv ------------------------------------------------------------------
class FlvMuxer : public Muxer {
public:
    explicit FlvMuxer(FILE *f) : fp_(f) {}
    int set_param(const x264_param_t *p) override
    {
        Bytes b;
        b.str("FLV"); b.u8(1); b.u8(1); b.be(9, 4); b.be(0, 4);                  // signature, version, video only, header size, PreviousTagSize0
        Bytes m;                                                                 // script-data tag body
        m.u8(2); amf_string(m, "onMetaData");
        m.u8(8); m.be(7, 4);                                                     // ECMA array of 7
        amf_string(m, "width"); amf_double(m, p->i_width);
        amf_string(m, "height"); amf_double(m, p->i_height);
        amf_string(m, "framerate");
        if (!p->b_vfr_input) amf_double(m, (double)p->i_fps_num / p->i_fps_den);
        else { framerate_pos_ = (long)(b.size() + 11 + m.size() + 1); amf_double(m, 0); }
        amf_string(m, "videocodecid"); amf_double(m, 7);
        amf_string(m, "duration"); duration_pos_ = (long)(b.size() + 11 + m.size() + 1); amf_double(m, 0);
        amf_string(m, "filesize"); filesize_pos_ = (long)(b.size() + 11 + m.size() + 1); amf_double(m, 0);
        amf_string(m, "videodatarate"); bitrate_pos_ = (long)(b.size() + 11 + m.size() + 1); amf_double(m, 0);
        amf_string(m, ""); m.u8(9);                                              // end of object
        tag(b, 18, 0, m);
        fps_num_ = p->i_fps_num; fps_den_ = p->i_fps_den;
        timebase_ = (double)p->i_timebase_num / p->i_timebase_den;
        return fwrite(b.d.data(), b.size(), 1, fp_) == 1 ? 0 : -1;
    }
    int write_headers(const x264_nal_t *nal) override
    {
        const int sps_size = nal[0].i_payload - 4, pps_size = nal[1].i_payload - 4, sei_size = nal[2].i_payload;
        sei_.assign(nal[2].p_payload, nal[2].p_payload + sei_size);              // deferred until the first frame (players expect that)
        Bytes v;
        v.u8(0x17); v.u8(0); v.be(0, 3);                                         // key frame + AVC, sequence header, composition time 0
        put_avcc(v, nal[0].p_payload + 4, sps_size, nal[1].p_payload + 4, pps_size);
        Bytes b;
        tag(b, 9, 0, v);
        return fwrite(b.d.data(), b.size(), 1, fp_) == 1 ? sei_size + sps_size + 4 + pps_size + 4 : -1;
    }
    int write_frame(const uint8_t *p, int size, const x264_picture_t *pic) override
    {
        if (!nframes_) delay_ = -pic->i_dts;
        const int64_t dts = ms(pic->i_dts + delay_), cts = ms(pic->i_pts + delay_);
        Bytes v;
        v.u8((pic->b_keyframe ? 0x10u : 0x20u) | 7u); v.u8(1); v.be((uint64_t)((cts - dts) & 0xffffff), 3);      // frame type + AVC, NALU, CTS offset
        v.raw(sei_.data(), sei_.size());
        sei_.clear();
        v.raw(p, (size_t)size);
        Bytes b;
        tag(b, 9, dts, v);
        nframes_++;
        return fwrite(b.d.data(), b.size(), 1, fp_) == 1 ? size : -1;
    }
    int close(int64_t largest_pts, int64_t second_largest_pts) override
    {
        int ret = 0;
        const double total = nframes_ == 1 ? (fps_num_ ? (double)fps_den_ / fps_num_ : 0) : (double)(2 * largest_pts - second_largest_pts) * timebase_;
        if (total > 0) {
            fseek(fp_, 0, SEEK_END);
            const double filesize = (double)ftell(fp_);
            if (framerate_pos_) ret |= patch(framerate_pos_, (double)nframes_ / total);
            ret |= patch(duration_pos_, total);
            ret |= patch(filesize_pos_, filesize);
            ret |= patch(bitrate_pos_, filesize * 8 / (total * 1000));
        }
        if (fclose(fp_)) ret = -1;
        fp_ = nullptr;
        return ret;
    }
    ~FlvMuxer() override { if (fp_) fclose(fp_); }
private:
    static void amf_string(Bytes &b, const char *s) { b.be(strlen(s), 2); b.str(s); }
    static void amf_double(Bytes &b, double v) { uint64_t u; memcpy(&u, &v, 8); b.u8(0); b.be(u, 8); }
    static void tag(Bytes &b, int type, int64_t ts, const Bytes &body)          // tag header + body + PreviousTagSize
    {
        b.u8((unsigned)type); b.be(body.size(), 3); b.be((uint64_t)(ts & 0xffffff), 3); b.u8((unsigned)((ts >> 24) & 0xff)); b.be(0, 3);
        b.raw(body.d.data(), body.size());
        b.be(body.size() + 11, 4);
    }
    int64_t ms(int64_t t) const { return (int64_t)((double)t * timebase_ * 1000 + 0.5); }
    int patch(long pos, double v)
    {
        uint64_t u; memcpy(&u, &v, 8);
        Bytes b; b.be(u, 8);
        return !fseek(fp_, pos, SEEK_SET) && fwrite(b.d.data(), 8, 1, fp_) == 1 ? 0 : -1;
    }
    FILE *fp_;
    std::vector<uint8_t> sei_;
    long framerate_pos_ = 0, duration_pos_ = 0, filesize_pos_ = 0, bitrate_pos_ = 0;
    uint32_t fps_num_ = 25, fps_den_ = 1;
    double timebase_ = 0.04;
    int64_t delay_ = 0;
    int nframes_ = 0;
};


// ------------------------------------------------------------------ mp4 ------------------------------------------------------------------
class Mp4Muxer : public Muxer {
public:
    explicit Mp4Muxer(FILE *f) : fp_(f) {}
    int set_param(const x264_param_t *p) override
    {
        const int delay_frames = p->i_bframe ? (p->i_bframe_pyramid ? 2 : 1) : 0;    // mp4_lsmash.c:198-204 (no --dts-compress in the driver: x1)
        (void)delay_frames;
        media_ts_ = p->i_timebase_den; time_inc_ = p->i_timebase_num;
        if (!media_ts_ || !time_inc_) return -1;
        width_ = p->i_width; height_ = p->i_height;
        disp_w_ = (uint32_t)p->i_width << 16; disp_h_ = (uint32_t)p->i_height << 16;
        if (p->vui.i_sar_width && p->vui.i_sar_height) {                             // mp4_lsmash.c:245-256
            const double sar = (double)p->vui.i_sar_width / p->vui.i_sar_height;
            if (sar > 1.0) disp_w_ = (uint32_t)(disp_w_ * sar); else disp_h_ = (uint32_t)(disp_h_ / sar);
            par_h_ = (uint32_t)p->vui.i_sar_width; par_v_ = (uint32_t)p->vui.i_sar_height;
        }
        prim_ = p->vui.i_colorprim; trc_ = p->vui.i_transfer;
        matrix_ = p->vui.i_colmatrix >= 0 ? p->vui.i_colmatrix : 2; full_ = p->vui.b_fullrange > 0;
        Bytes b;
        b.be(28, 4); b.str("ftyp"); b.str("mp42"); b.be(0, 4); b.str("mp42"); b.str("mp41"); b.str("isom");
        b.be(1, 4); b.str("mdat"); b.be(0, 8);                                       // largesize form: patched at close, no 4 GiB limit
        mdat_pos_ = 28;
        pos_ = (int64_t)b.size();
        return fwrite(b.d.data(), b.size(), 1, fp_) == 1 ? 0 : -1;
    }
    int write_headers(const x264_nal_t *nal) override
    {
        const int sps_size = nal[0].i_payload - 4, pps_size = nal[1].i_payload - 4, sei_size = nal[2].i_payload;
        sps_.assign(nal[0].p_payload + 4, nal[0].p_payload + 4 + sps_size);
        pps_.assign(nal[1].p_payload + 4, nal[1].p_payload + 4 + pps_size);
        sei_.assign(nal[2].p_payload, nal[2].p_payload + sei_size);                  // goes in front of the first sample (mp4_lsmash.c:377-383)
        return sei_size + sps_size + pps_size;
    }
    int write_frame(const uint8_t *p, int size, const x264_picture_t *pic) override
    {
        if (samples_.empty()) start_offset_ = -pic->i_dts;
        Sample sm;
        sm.dts = (uint64_t)(pic->i_dts + start_offset_) * time_inc_;
        sm.cts = (uint64_t)(pic->i_pts + start_offset_) * time_inc_;
        sm.size = (uint32_t)(sei_.size() + (size_t)size); sm.sync = pic->b_keyframe != 0; sm.pos = pos_;
        if (!sei_.empty() && fwrite(sei_.data(), sei_.size(), 1, fp_) != 1) return -1;
        sei_.clear();
        if (fwrite(p, (size_t)size, 1, fp_) != 1) return -1;
        pos_ += sm.size;
        samples_.push_back(sm);
        return size;
    }
    int close(int64_t largest_pts, int64_t second_largest_pts) override
    {
        int ret = 0;
        const uint32_t movie_ts = 600;                                               // L-SMASH's default movie timescale
        const size_t n = samples_.size();
        const uint64_t last_delta = (uint64_t)((largest_pts - second_largest_pts) ? (largest_pts - second_largest_pts) : 1) * time_inc_;
        uint64_t media_dur = 0;
        std::vector<uint32_t> delta(n);
        for (size_t i = 0; i < n; i++) { delta[i] = (uint32_t)(i + 1 < n ? samples_[i + 1].dts - samples_[i].dts : last_delta); media_dur += delta[i]; }
        const uint64_t first_cts = (uint64_t)start_offset_ * time_inc_;
        const uint64_t pres_dur = n ? (uint64_t)(((double)((uint64_t)(largest_pts + (int64_t)(last_delta / time_inc_)) * time_inc_) / media_ts_) * movie_ts) : 0;   // mp4_lsmash.c:146-150
        // ---- sample tables ----
        Bytes stts, ctts, stss, stsc, stsz, stco;
        {   // stts: runs of equal deltas
            std::vector<std::pair<uint32_t, uint32_t>> runs;
            for (size_t i = 0; i < n; i++) { if (!runs.empty() && runs.back().second == delta[i]) runs.back().first++; else runs.push_back({1u, delta[i]}); }
            full(stts, "stts", 0, 0); stts.be(runs.size(), 4);
            for (auto &r : runs) { stts.be(r.first, 4); stts.be(r.second, 4); }
            close_box(stts);
        }
        bool need_ctts = false, all_sync = true;
        for (const Sample &sm : samples_) { need_ctts |= sm.cts != sm.dts; all_sync &= sm.sync; }
        if (need_ctts) {
            std::vector<std::pair<uint32_t, uint32_t>> runs;
            for (const Sample &sm : samples_) { const uint32_t o = (uint32_t)(sm.cts - sm.dts); if (!runs.empty() && runs.back().second == o) runs.back().first++; else runs.push_back({1u, o}); }
            full(ctts, "ctts", 0, 0); ctts.be(runs.size(), 4);
            for (auto &r : runs) { ctts.be(r.first, 4); ctts.be(r.second, 4); }
            close_box(ctts);
        }
        if (!all_sync) {
            uint32_t cnt = 0;
            for (const Sample &sm : samples_) cnt += sm.sync;
            full(stss, "stss", 0, 0); stss.be(cnt, 4);
            for (size_t i = 0; i < n; i++) if (samples_[i].sync) stss.be(i + 1, 4);
            close_box(stss);
        }
        // chunks: consecutive samples of about half a second (one track: every chunk is contiguous in the mdat)
        std::vector<std::pair<size_t, uint32_t>> chunks;                             // first sample, sample count
        {
            uint64_t acc = 0;
            for (size_t i = 0; i < n; i++) {
                if (chunks.empty() || acc * 2 >= media_ts_) { chunks.push_back({i, 0u}); acc = 0; }
                chunks.back().second++; acc += delta[i];
            }
        }
        full(stsc, "stsc", 0, 0);
        {
            std::vector<std::pair<uint32_t, uint32_t>> runs;                         // first_chunk, samples_per_chunk
            for (size_t c = 0; c < chunks.size(); c++) if (runs.empty() || runs.back().second != chunks[c].second) runs.push_back({(uint32_t)c + 1, chunks[c].second});
            stsc.be(runs.size(), 4);
            for (auto &r : runs) { stsc.be(r.first, 4); stsc.be(r.second, 4); stsc.be(1, 4); }
        }
        close_box(stsc);
        full(stsz, "stsz", 0, 0); stsz.be(0, 4); stsz.be(n, 4);
        uint32_t max_size = 0; uint64_t total = 0;
        for (const Sample &sm : samples_) { stsz.be(sm.size, 4); max_size = sm.size > max_size ? sm.size : max_size; total += sm.size; }
        close_box(stsz);
        const bool co64 = pos_ > 0xffffffffll;
        full(stco, co64 ? "co64" : "stco", 0, 0); stco.be(chunks.size(), 4);
        for (auto &c : chunks) stco.be((uint64_t)samples_[c.first].pos, co64 ? 8 : 4);
        close_box(stco);
        // btrt: decoding buffer = largest sample, peak rate over any one-second span of samples, average rate
        uint32_t avg_rate = 0, max_rate = 0;
        if (media_dur) avg_rate = (uint32_t)((double)total * 8 * media_ts_ / (double)media_dur);
        for (size_t i = 0, j = 0; i < n; i++) {
            uint64_t bytes = 0, dur = 0;
            for (j = i; j < n && dur < media_ts_; j++) { bytes += samples_[j].size; dur += delta[j]; }
            const uint32_t r = (uint32_t)(dur >= media_ts_ ? bytes * 8 : (dur ? (double)bytes * 8 * media_ts_ / (double)dur : 0));
            max_rate = r > max_rate ? r : max_rate;
            if (j == n) break;
        }
        // ---- stsd: avc1 + avcC + colr + pasp + btrt ----
        Bytes avc1; box(avc1, "avc1");
        avc1.be(0, 6); avc1.be(1, 2);                                                // reserved, data_reference_index
        avc1.be(0, 16);                                                              // pre_defined / reserved
        avc1.be((uint64_t)width_, 2); avc1.be((uint64_t)height_, 2);
        avc1.be(0x00480000, 4); avc1.be(0x00480000, 4); avc1.be(0, 4); avc1.be(1, 2);       // 72 dpi, reserved, frame_count
        avc1.be(0, 32);                                                              // compressorname
        avc1.be(0x0018, 2); avc1.be(0xffff, 2);                                      // depth, pre_defined -1
        { Bytes c; box(c, "avcC"); put_avcc(c, sps_.data(), (int)sps_.size(), pps_.data(), (int)pps_.size());
          if (sps_.size() > 1 && (sps_[1] == 100 || sps_[1] == 110 || sps_[1] == 122 || sps_[1] == 144)) { c.u8(0xfc | 1); c.u8(0xf8); c.u8(0xf8); c.u8(0); }   // High profiles: chroma_format 4:2:0, 8-bit luma / chroma, no SPS extensions
          close_box(c); avc1.raw(c.d.data(), c.size()); }
        { Bytes c; box(c, "colr"); c.str("nclx"); c.be((uint64_t)prim_, 2); c.be((uint64_t)trc_, 2); c.be((uint64_t)matrix_, 2); c.u8(full_ ? 0x80 : 0); close_box(c); avc1.raw(c.d.data(), c.size()); }
        if (par_h_ && par_v_) { Bytes c; box(c, "pasp"); c.be(par_h_, 4); c.be(par_v_, 4); close_box(c); avc1.raw(c.d.data(), c.size()); }
        { Bytes c; box(c, "btrt"); c.be(max_size, 4); c.be(max_rate, 4); c.be(avg_rate, 4); close_box(c); avc1.raw(c.d.data(), c.size()); }
        close_box(avc1);
        Bytes stsd; full(stsd, "stsd", 0, 0); stsd.be(1, 4); stsd.raw(avc1.d.data(), avc1.size()); close_box(stsd);
        Bytes stbl; box(stbl, "stbl");
        for (const Bytes *t : {&stsd, &stts, &ctts, &stss, &stsc, &stsz, &stco}) stbl.raw(t->d.data(), t->size());
        close_box(stbl);
        Bytes vmhd; full(vmhd, "vmhd", 0, 1); vmhd.be(0, 8); close_box(vmhd);
        Bytes dinf; box(dinf, "dinf"); { Bytes d; full(d, "dref", 0, 0); d.be(1, 4); { Bytes u; full(u, "url ", 0, 1); close_box(u); d.raw(u.d.data(), u.size()); } close_box(d); dinf.raw(d.d.data(), d.size()); } close_box(dinf);
        Bytes minf; box(minf, "minf"); for (const Bytes *t : {&vmhd, &dinf, &stbl}) minf.raw(t->d.data(), t->size()); close_box(minf);
        Bytes mdhd; full(mdhd, "mdhd", 1, 0); mdhd.be(0, 8); mdhd.be(0, 8); mdhd.be(media_ts_, 4); mdhd.be(media_dur, 8); mdhd.be(0x55c4, 2); mdhd.be(0, 2); close_box(mdhd);   // language "und"
        Bytes hdlr; full(hdlr, "hdlr", 0, 0); hdlr.be(0, 4); hdlr.str("vide"); hdlr.be(0, 12); hdlr.str("x264gpu Video Media Handler"); hdlr.u8(0); close_box(hdlr);
        Bytes mdia; box(mdia, "mdia"); for (const Bytes *t : {&mdhd, &hdlr, &minf}) mdia.raw(t->d.data(), t->size()); close_box(mdia);
        Bytes tkhd; full(tkhd, "tkhd", 0, 7); tkhd.be(0, 4); tkhd.be(0, 4); tkhd.be(1, 4); tkhd.be(0, 4); tkhd.be(pres_dur, 4);      // enabled | in movie | in preview
        tkhd.be(0, 8); tkhd.be(0, 2); tkhd.be(0, 2); tkhd.be(0, 2); tkhd.be(0, 2); matrix(tkhd); tkhd.be(disp_w_, 4); tkhd.be(disp_h_, 4); close_box(tkhd);
        Bytes edts; box(edts, "edts"); { Bytes e; full(e, "elst", 1, 0); e.be(1, 4); e.be(pres_dur, 8); e.be(first_cts, 8); e.be(0x00010000, 4); close_box(e); edts.raw(e.d.data(), e.size()); } close_box(edts);
        Bytes trak; box(trak, "trak"); for (const Bytes *t : {&tkhd, &edts, &mdia}) trak.raw(t->d.data(), t->size()); close_box(trak);
        Bytes mvhd; full(mvhd, "mvhd", 0, 0); mvhd.be(0, 4); mvhd.be(0, 4); mvhd.be(movie_ts, 4); mvhd.be(pres_dur, 4); mvhd.be(0x00010000, 4); mvhd.be(0x0100, 2);
        mvhd.be(0, 10); matrix(mvhd); mvhd.be(0, 24); mvhd.be(2, 4); close_box(mvhd);
        Bytes moov; box(moov, "moov"); moov.raw(mvhd.d.data(), mvhd.size()); moov.raw(trak.d.data(), trak.size()); close_box(moov);
        if (fwrite(moov.d.data(), moov.size(), 1, fp_) != 1) ret = -1;
        Bytes sz; sz.be((uint64_t)(pos_ - mdat_pos_), 8);                            // mdat largesize
        if (fseek(fp_, (long)mdat_pos_ + 8, SEEK_SET) || fwrite(sz.d.data(), 8, 1, fp_) != 1) ret = -1;
        if (fclose(fp_)) ret = -1;
        fp_ = nullptr;
        return ret;
    }
    ~Mp4Muxer() override { if (fp_) fclose(fp_); }
private:
    struct Sample { uint64_t dts, cts; int64_t pos; uint32_t size; bool sync; };
    // a box is opened with a zero size and closed by patching the size of the whole buffer (one buffer per box)
    static void box(Bytes &b, const char *type) { b.be(0, 4); b.str(type); }
    static void full(Bytes &b, const char *type, unsigned version, unsigned flags) { box(b, type); b.u8(version); b.be(flags, 3); }
    static void close_box(Bytes &b) { const uint32_t n = (uint32_t)b.size(); for (int i = 0; i < 4; i++) b.d[(size_t)i] = (uint8_t)(n >> (8 * (3 - i))); }
    static void matrix(Bytes &b) { const uint32_t m[9] = {0x00010000, 0, 0, 0, 0x00010000, 0, 0, 0, 0x40000000}; for (uint32_t v : m) b.be(v, 4); }
    FILE *fp_;
    std::vector<uint8_t> sps_, pps_, sei_;
    std::vector<Sample> samples_;
    uint64_t media_ts_ = 25, time_inc_ = 1;
    int64_t start_offset_ = 0, pos_ = 0, mdat_pos_ = 0;
    int width_ = 0, height_ = 0, prim_ = 2, trc_ = 2, matrix_ = 2;
    bool full_ = false;
    uint32_t disp_w_ = 0, disp_h_ = 0, par_h_ = 0, par_v_ = 0;
};

}  // namespace

// select_output (codec.c:1111-1164): by --muxer or, for "auto", by the file name's extension.  annexb_out tells the caller how
// the encoder must frame its NAL units (raw: start codes + repeated headers; containers: 4-byte lengths, headers once).
Muxer *open_muxer(const char *filename, const char *muxer, int *annexb_out, const char **error)
{
    std::string ext = muxer ? muxer : "auto";
    if (ext == "auto") { const char *dot = strrchr(filename, '.'); ext = dot ? dot + 1 : ""; }
    for (char &ch : ext) ch = (char)tolower((unsigned char)ch);
    *error = nullptr;
    if (ext == "avi") { *error = "not compiled with this output support"; return nullptr; }
    FILE *f = fopen(filename, "w+b");
    if (!f) { *error = "could not open output file"; return nullptr; }
    if (ext == "mkv") { *annexb_out = 0; return new MkvMuxer(f); }
    if (ext == "flv") { *annexb_out = 0; return new FlvMuxer(f); }
    if (ext == "mp4") { *annexb_out = 0; return new Mp4Muxer(f); }
    *annexb_out = 1;
    return new RawMuxer(f);
}

}  // namespace x264host

// ---- diagnostics (tests only; include/x264gpu_host.h): the muxers without an encoder, fed with caller-supplied NAL bytes ----
extern "C" {

void *x264host_mux_open(const char *filename, const char *muxer, int *annexb)
{
    const char *err = nullptr;
    int a = 1;
    x264host::Muxer *m = x264host::open_muxer(filename, muxer, &a, &err);
    if (annexb) *annexb = a;
    return m;
}

int x264host_mux_set_param(void *h, int width, int height, uint32_t fps_num, uint32_t fps_den, uint32_t timebase_num, uint32_t timebase_den,
                           int sar_width, int sar_height, int vfr)
{
    x264_param_t p;
    memset(&p, 0, sizeof(p));
    p.i_width = width; p.i_height = height; p.i_fps_num = fps_num; p.i_fps_den = fps_den; p.i_timebase_num = timebase_num; p.i_timebase_den = timebase_den;
    p.vui.i_sar_width = sar_width; p.vui.i_sar_height = sar_height; p.b_vfr_input = vfr; p.i_frame_packing = -1;
    p.vui.i_colorprim = 2; p.vui.i_transfer = 2; p.vui.i_colmatrix = -1; p.vui.b_fullrange = -1;      /* x264_param_default */
    return ((x264host::Muxer *)h)->set_param(&p);
}

/* sps / pps / sei: NAL units with their 4-byte prefix, as x264_encoder_headers returns them */
int x264host_mux_write_headers(void *h, const uint8_t *sps, int sps_size, const uint8_t *pps, int pps_size, const uint8_t *sei, int sei_size)
{
    std::vector<uint8_t> all;
    all.insert(all.end(), sps, sps + sps_size); all.insert(all.end(), pps, pps + pps_size); all.insert(all.end(), sei, sei + sei_size);
    x264_nal_t nal[3];
    memset(nal, 0, sizeof(nal));
    nal[0].p_payload = all.data(); nal[0].i_payload = sps_size;
    nal[1].p_payload = all.data() + sps_size; nal[1].i_payload = pps_size;
    nal[2].p_payload = all.data() + sps_size + pps_size; nal[2].i_payload = sei_size;
    return ((x264host::Muxer *)h)->write_headers(nal);
}

int x264host_mux_write_frame(void *h, const uint8_t *payload, int size, int64_t pts, int64_t dts, int keyframe, int type)
{
    x264_picture_t pic;
    memset(&pic, 0, sizeof(pic));
    pic.i_pts = pts; pic.i_dts = dts; pic.b_keyframe = keyframe; pic.i_type = type;
    return ((x264host::Muxer *)h)->write_frame(payload, size, &pic);
}

int x264host_mux_close(void *h, int64_t largest_pts, int64_t second_largest_pts)
{
    x264host::Muxer *m = (x264host::Muxer *)h;
    const int rc = m->close(largest_pts, second_largest_pts);
    delete m;
    return rc;
}

}  // extern "C"
