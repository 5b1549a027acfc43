// vfw.cpp — boundary B2: Linux re-host of the x264vfw driver shell for the compress path.
//   DriverProc                 driverproc.c:89-301     message dispatch, DRV_OPEN returns the CODEC* as id
//   compress_get_format/_get_size/_query   codec.c:581-652
//   compress_begin             codec.c:1381-1684       CONFIG + extra cmdline -> x264_param_t -> x264_encoder_open
//   compress / encode_frame    codec.c:1686-1835       ICCOMPRESS in, contiguous NALs out, AVIIF_KEYFRAME
//   compress_end / frames_info codec.c:1838-1894
// Same message set, return conventions and sticky-error behaviour; the Windows-only parts (registry, dialogs,
// log window, VirtualDub hack, file muxers, decoder) are out of scope.
#include "host.hpp"
#include "../../include/vfw_shim.h"
#include "../../include/x264gpu_host.h"
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

const DWORD kFourccOut[] = { mmioFOURCC('H', '2', '6', '4'), mmioFOURCC('h', '2', '6', '4'), mmioFOURCC('X', '2', '6', '4'),
                             mmioFOURCC('x', '2', '6', '4'), mmioFOURCC('A', 'V', 'C', '1'), mmioFOURCC('a', 'v', 'c', '1'),
                             mmioFOURCC('V', 'S', 'S', 'H') };                                       /* codec.c:112-121 */
const char *const kPresets[] = { "ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo" };
const char *const kTunes[] = { "", "film", "animation", "grain", "stillimage", "psnr", "ssim" };
const char *const kProfiles[] = { "", "baseline", "main", "high" };
const int kLevels[] = { -1, 10, 9, 11, 12, 13, 20, 21, 22, 30, 31, 32, 40, 41, 42, 50, 51, 52, 60, 61, 62 };

enum { CSP_NONE = 0 };                                         /* the rest are the driver's ids: X264GPU_CSP_* (csp.h:30-44) */

struct CODEC {                          /* x264vfw.h:187-252, compress-side members */
    x264_t *h;
    X264VFW_CONFIG config;
    int b_encoder_error;
    BITMAPINFOHEADER *prev_lpbiOutput;
    DWORD prev_output_biSizeImage;
    int b_check_size;
    int b_use_vd_hack; DWORD save_fourcc;      /* X264VFW_USE_VIRTUALDUB_HACK (x264vfw.h:199-202): delayed pictures answered with a one-byte XVID-style drop frame */
    int b_warn_frame_loss;
    int i_frame_remain, i_frame_total;
    uint32_t i_fps_num, i_fps_den;
    x264_picture_t conv_pic;
    int b_user_ref;
    uint8_t *d_raw; size_t raw_cap;     /* device copy of the caller's frame in its native colourspace */
    int colmatrix709, fullrange;        /* x264vfw_csp_init arguments (codec.c:1570-1577,1672) */
    x264host::Muxer *cli_hout; int b_cli_output, b_no_output;   /* file output instead of the VfW buffer (codec.c:1111-1164,1609-1663; output/raw.c, matroska.c, flv.c) */
    int64_t largest_pts, second_largest_pts;      /* for close_file (codec.c:1858-1866) */
    std::string log;
    std::string stats_path;             /* the 2-pass statistics file (codec.c:1386,1447,1537-1541) */
};

void config_defaults(X264VFW_CONFIG *c)
{
    memset(c, 0, sizeof(*c));
    c->i_format_version = X264VFW_FORMAT_VERSION;
    c->i_preset = 5;                    /* medium (config.c:96) */
    c->i_level = 0;                     /* auto */
    c->i_encoding_type = 2;             /* single pass CRF after the GordianKnot remap (config.c:205-228,256) */
    c->i_qp = 23; c->i_rf_constant = 230; c->i_passbitrate = 800; c->i_pass = 1;
    c->b_fast1pass = 0; c->b_createstats = 0; c->b_updatestats = 1;      /* config.c:114-116 */
    snprintf(c->stats, sizeof(c->stats), "./x264.stats");               /* config.c:140 (".\\x264.stats") */
    c->i_output_mode = 0; c->b_vd_hack = 0; c->output_file[0] = 0;      /* config.c:118,121,142 */
    c->i_log_level = 2;                 /* warning */
    c->i_sar_width = c->i_sar_height = 1;
}

void log_cb(void *priv, int level, const char *fmt, va_list ap)
{
    CODEC *codec = (CODEC *)priv;
    char buf[1024];
    vsnprintf(buf, sizeof(buf), fmt, ap);
    if (codec) codec->log += buf;
    if (level <= X264_LOG_ERROR) fputs(buf, stderr);
}
void vlog(CODEC *codec, int level, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    log_cb(codec, level, fmt, ap);
    va_end(ap);
}

int get_csp(const BITMAPINFOHEADER *hdr)          /* codec.c:187-231 */
{
    /* For YUV the bitmap is always top-down regardless of the biHeight sign */
    switch (hdr->biCompression) {
    case mmioFOURCC('I', '4', '2', '0'): case mmioFOURCC('I', 'Y', 'U', 'V'): return X264GPU_CSP_I420;
    case mmioFOURCC('Y', 'V', '1', '2'): return X264GPU_CSP_YV12;
    case mmioFOURCC('Y', 'V', '1', '6'): return X264GPU_CSP_YV16;
    case mmioFOURCC('Y', 'V', '2', '4'): return X264GPU_CSP_YV24;
    case mmioFOURCC('Y', 'U', 'Y', 'V'): case mmioFOURCC('Y', 'U', 'Y', '2'): return X264GPU_CSP_YUYV;
    case mmioFOURCC('U', 'Y', 'V', 'Y'): case mmioFOURCC('H', 'D', 'Y', 'C'): return X264GPU_CSP_UYVY;
    case BI_RGB: {
        const int vflip = hdr->biHeight < 0 ? 0 : X264GPU_CSP_VFLIP;      /* bottom-up DIB */
        if (hdr->biBitCount == 24) return X264GPU_CSP_BGR | vflip;
        if (hdr->biBitCount == 32) return X264GPU_CSP_BGRA | vflip;
        return CSP_NONE;
    }
    default: return CSP_NONE;        /* NV12 input belongs to an NV12 encoder (csp.c:489-491), not built */
    }
}
bool supported_fourcc(DWORD f) { for (DWORD k : kFourccOut) if (k == f) return true; return false; }

LRESULT compress_get_size(BITMAPINFO *out)        /* codec.c:618-621 */
{
    return ((out->bmiHeader.biWidth + 15) & ~15) * ((out->bmiHeader.biHeight + 31) & ~31) * 3 + 4096;
}

LRESULT compress_get_format(CODEC *codec, BITMAPINFO *in, BITMAPINFO *out)
{
    if (!out) return sizeof(BITMAPINFOHEADER);
    BITMAPINFOHEADER *ih = &in->bmiHeader, *oh = &out->bmiHeader;
    if (get_csp(ih) == CSP_NONE) return ICERR_BADFORMAT;
    int w = ih->biWidth, h = abs(ih->biHeight);
    if (w <= 0 || h <= 0 || (w % 2) || (h % 2)) return ICERR_BADFORMAT;
    memset(oh, 0, sizeof(*oh));
    oh->biSize = sizeof(*oh); oh->biWidth = w; oh->biHeight = h; oh->biPlanes = 1; oh->biBitCount = 24;
    int fi = codec->config.i_fourcc;
    oh->biCompression = kFourccOut[fi >= 0 && fi < 7 ? fi : 0];
    oh->biSizeImage = (DWORD)compress_get_size(out);
    return ICERR_OK;
}

LRESULT compress_query(CODEC *codec, BITMAPINFO *in, BITMAPINFO *out)
{
    BITMAPINFOHEADER *ih = &in->bmiHeader;
    if (get_csp(ih) == CSP_NONE) return ICERR_BADFORMAT;
    int w = ih->biWidth, h = abs(ih->biHeight);
    if (w <= 0 || h <= 0 || (w % 2) || (h % 2)) return ICERR_BADFORMAT;
    if (!out) return ICERR_OK;
    if (w != out->bmiHeader.biWidth || h != out->bmiHeader.biHeight) return ICERR_BADFORMAT;
    if (!supported_fourcc(out->bmiHeader.biCompression)) return ICERR_BADFORMAT;
    return ICERR_OK;
}

int encode_frame(CODEC *codec, x264_picture_t *pic, x264_picture_t *pic_out, uint8_t *buf, DWORD buf_size, int *got_picture);

LRESULT compress_end(CODEC *codec)
{
    if (codec->h) {
        if (!codec->b_encoder_error && codec->b_cli_output && x264_encoder_delayed_frames(codec->h)) {
            /* flush delayed frames into the output file (codec.c:1842-1856; b_flush_delayed == file output here):
             * GOP-parallel mode (--threads G) holds (G-1)*keyint frames back */
            x264_picture_t pic_out;
            int got_picture;
            vlog(codec, X264_LOG_DEBUG, "flush delayed frames\n");
            do {
                if (encode_frame(codec, nullptr, &pic_out, nullptr, 0, &got_picture) < 0) break;
            } while (x264_encoder_delayed_frames(codec->h));
        }
        x264_encoder_close(codec->h);
        codec->h = nullptr;
    }
    x264_picture_clean(&codec->conv_pic);
    memset(&codec->conv_pic, 0, sizeof(codec->conv_pic));
    if (codec->d_raw) { x264gpu_free(codec->d_raw); codec->d_raw = nullptr; codec->raw_cap = 0; }
    if (codec->cli_hout) { codec->cli_hout->close(codec->largest_pts, codec->second_largest_pts); delete codec->cli_hout; codec->cli_hout = nullptr; }   /* cli_output.close_file */
    codec->b_cli_output = 0;
    codec->b_encoder_error = 0;
    return ICERR_OK;
}

// split the extra command line like codec.c:1169-1223 (whitespace separated, double quotes group)
std::vector<std::string> split_cmdline(const char *s)
{
    std::vector<std::string> out;
    std::string cur;
    bool in_q = false, have = false;
    for (; *s; s++) {
        if (*s == '"') { in_q = !in_q; have = true; }
        else if (!in_q && (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n')) { if (have) { out.push_back(cur); cur.clear(); have = false; } }
        else { cur += *s; have = true; }
    }
    if (have) out.push_back(cur);
    return out;
}

bool option_takes_value(const std::string &name)
{
    static const char *const flags[] = { "fast-firstpass", "slow-firstpass", "no-b-adapt", "open-gop", "bluray-compat", "intra-refresh",
        "no-scenecut", "nf", "no-deblock", "interlaced", "no-interlaced", "tff", "bff", "constrained-intra", "cabac", "no-cabac", "asm", "no-asm",
        "weightb", "no-weightb", "psy", "no-psy", "mixed-refs", "no-mixed-refs", "chroma-me", "no-chroma-me", "8x8dct", "no-8x8dct",
        "fast-pskip", "no-fast-pskip", "dct-decimate", "no-dct-decimate", "mbtree", "no-mbtree", "sliced-threads", "no-sliced-threads",
        "thread-input", "deterministic", "non-deterministic", "cpu-independent", "psnr", "no-psnr", "ssim", "no-ssim", "quiet", "verbose",
        "progress", "no-progress", "aud", "no-aud", "force-cfr", "pic-struct", "fake-interlaced", "stitchable", "filler", "vd-hack", "no-output",
        "stdout", "stdin", "dts-compress", 0 };
    for (int i = 0; flags[i]; i++) if (name == flags[i]) return false;
    return true;
}

LRESULT compress_begin(CODEC *codec, BITMAPINFO *in, BITMAPINFO *out)
{
    X264VFW_CONFIG *cfg = &codec->config;
    compress_end(codec);                                                    /* destroy previous handle (codec.c:1394) */
    codec->log.clear();
    if (compress_query(codec, in, out) != ICERR_OK) {
        vlog(codec, X264_LOG_ERROR, "incompatible input/output frame format (encode)\n");
        codec->b_encoder_error = 1;
        return ICERR_BADFORMAT;
    }
    codec->b_check_size = out->bmiHeader.biSizeImage != 0;
    codec->b_use_vd_hack = cfg->b_vd_hack != 0;                             /* codec.c:1410; --vd-hack sets it too (codec.c:1313) */
    codec->save_fourcc = out->bmiHeader.biCompression;                      /* codec.c:1411 */
    codec->b_user_ref = 0;
    codec->i_frame_remain = codec->i_frame_total ? codec->i_frame_total : -1;

    const char *preset = cfg->i_preset >= 0 && cfg->i_preset < 10 ? kPresets[cfg->i_preset] : nullptr;
    const char *profile = cfg->i_profile > 0 && cfg->i_profile < 4 ? kProfiles[cfg->i_profile] : nullptr;
    std::string tune = cfg->i_tuning > 0 && cfg->i_tuning < 7 ? kTunes[cfg->i_tuning] : "";
    if (cfg->b_fastdecode) tune += tune.empty() ? "fastdecode" : ",fastdecode";
    if (cfg->b_zerolatency) tune += tune.empty() ? "zerolatency" : ",zerolatency";
    std::vector<std::string> argv = split_cmdline(cfg->extra_cmdline);
    std::string preset_s = preset ? preset : "", profile_s = profile ? profile : "";
    cfg->stats[sizeof(cfg->stats) - 1] = 0; cfg->output_file[sizeof(cfg->output_file) - 1] = 0;
    std::string out_file = cfg->i_output_mode == 1 ? cfg->output_file : "-", muxer = "auto";      /* codec.c:1545-1546 */
    std::string stats_path = cfg->stats;                                   /* codec.c:1447,1537-1541 */
    int fast1pass = 0;
    codec->b_no_output = 0; codec->b_cli_output = 0;
    for (size_t i = 0; i + 1 < argv.size(); i++) {                          /* presets first (parse_preset_tune, codec.c:1198-1223) */
        if (argv[i] == "--preset") preset_s = argv[i + 1];
        if (argv[i] == "--tune") tune = argv[i + 1];
    }
    x264_param_t param;
    if (x264_param_default_preset(&param, preset_s.empty() ? nullptr : preset_s.c_str(), tune.empty() ? nullptr : tune.c_str()) < 0) {
        vlog(codec, X264_LOG_ERROR, "x264_param_default_preset failed\n");
        goto fail;
    }
    param.i_width = in->bmiHeader.biWidth;
    param.i_height = abs(in->bmiHeader.biHeight);
    param.i_csp = X264_CSP_I420;                                           /* choose_output_csp (codec.c:269-300,1472) with b_keep_input_csp = 0 */
    if (cfg->i_colorspace != 0) vlog(codec, X264_LOG_INFO, "colorspace %d (keep the input's colourspace): this build codes YUV 4:2:0 only — the input is converted\n", cfg->i_colorspace);
    param.i_frame_total = codec->i_frame_total;
    if (codec->i_fps_num > 0 && codec->i_fps_den > 0) { param.i_fps_num = codec->i_fps_num; param.i_fps_den = codec->i_fps_den; }
    param.i_level_idc = cfg->i_level >= 0 && cfg->i_level < (int)(sizeof(kLevels) / sizeof(kLevels[0])) ? kLevels[cfg->i_level] : -1;
    param.rc.b_stat_write = param.rc.b_stat_read = 0;
    switch (cfg->i_encoding_type) {                                         /* codec.c:1490-1533 */
    /* (single-pass modes write the statistics file when the dialog's "create stats" box is ticked: codec.c:1495,1501,1507,1513) */
    case 0: param.rc.i_rc_method = X264_RC_CQP; param.rc.i_qp_constant = 0; param.rc.b_stat_write = cfg->b_createstats != 0; break;
    case 1: param.rc.i_rc_method = X264_RC_CQP; param.rc.i_qp_constant = cfg->i_qp; param.rc.b_stat_write = cfg->b_createstats != 0; break;
    case 2: param.rc.i_rc_method = X264_RC_CRF; param.rc.f_rf_constant = (float)cfg->i_rf_constant * 0.1f; param.rc.b_stat_write = cfg->b_createstats != 0; break;
    case 3: param.rc.i_rc_method = X264_RC_ABR; param.rc.i_bitrate = cfg->i_passbitrate; param.rc.b_stat_write = cfg->b_createstats != 0; break;
    case 4:     /* multipass (codec.c:1516-1533): pass 1 writes the statistics and returns no stream, pass N reads them; the file is --stats' (config.c:140:
                 * ".\\x264.stats"); fast first pass only with --fast-firstpass (config.c:114 default 0) */
        param.rc.i_rc_method = X264_RC_ABR; param.rc.i_bitrate = cfg->i_passbitrate;
        if (cfg->i_pass <= 1) { codec->b_no_output = 1; fast1pass = cfg->b_fast1pass != 0; param.rc.b_stat_write = 1; }      /* codec.c:1519-1524 */
        else { param.rc.b_stat_write = cfg->b_updatestats != 0; param.rc.b_stat_read = 1; }                                  /* codec.c:1525-1529 */
        break;
    default: goto fail;
    }
    param.vui.i_sar_width = cfg->i_sar_width; param.vui.i_sar_height = cfg->i_sar_height;
    param.pf_log = log_cb; param.p_log_private = codec; param.i_log_level = cfg->i_log_level - 1;
    param.analyse.b_psnr = cfg->i_encoding_type > 0 && cfg->i_log_level >= 3 && cfg->b_psnr;
    param.analyse.b_ssim = cfg->i_encoding_type > 0 && cfg->i_log_level >= 3 && cfg->b_ssim;
    param.cpu = cfg->b_no_asm ? 0 : param.cpu;
    for (size_t i = 0; i < argv.size(); i++) {                              /* parse_cmdline (codec.c:1225-1378) */
        const std::string &a = argv[i];
        if (a.size() < 3 || a[0] != '-' || a[1] != '-') { vlog(codec, X264_LOG_ERROR, "unknown option or absent argument: '%s'\n", a.c_str()); goto fail; }
        std::string name = a.substr(2), value;
        bool has_value = false;
        size_t eq = name.find('=');
        if (eq != std::string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); has_value = true; }
        else if (option_takes_value(name) && i + 1 < argv.size()) { value = argv[++i]; has_value = true; }
        if (name == "preset" || name == "tune") continue;
        if (name == "profile") { profile_s = value; continue; }
        if (name == "ref") codec->b_user_ref = 1;
        if (name == "quiet") { param.i_log_level = X264_LOG_NONE; continue; }
        if (name == "range") {                                              /* OPT_RANGE (codec.c:1322-1329): auto / tv / pc */
            int r = value == "auto" ? -1 : value == "tv" ? 0 : value == "pc" ? 1 : -2;
            if (r == -2) { vlog(codec, X264_LOG_ERROR, "unknown range '%s'\n", value.c_str()); goto fail; }
            param.vui.b_fullrange = r;
            continue;
        }
        if (name == "output") { out_file = value; continue; }                /* OPT_OUTPUT (codec.c:1261-1263) */
        if (name == "muxer") { muxer = value; continue; }
        if (name == "no-output") { codec->b_no_output = 1; continue; }
        if (name == "fast-firstpass") { fast1pass = 1; continue; }               /* OPT_FASTFIRSTPASS / OPT_SLOWFIRSTPASS (codec.c:1296-1302) */
        if (name == "slow-firstpass") { fast1pass = 0; continue; }
        if (name == "stats") { stats_path = value; continue; }
        if (name == "vd-hack") { codec->b_use_vd_hack = 1; continue; }       /* OPT_VD_HACK (codec.c:1313-1315) */
        if (name == "dts-compress") {
            vlog(codec, X264_LOG_WARNING, "not supported option: '%s'\n", a.c_str());
            continue;
        }
        int rc = x264_param_parse(&param, name.c_str(), has_value ? value.c_str() : nullptr);
        if (rc == X264_PARAM_BAD_NAME) { vlog(codec, X264_LOG_ERROR, "unknown option: '%s'\n", a.c_str()); goto fail; }
        if (rc == X264_PARAM_BAD_VALUE) { vlog(codec, X264_LOG_ERROR, "invalid argument: '%s' = '%s'\n", a.c_str(), value.c_str()); goto fail; }
    }
    if (param.rc.b_stat_write || param.rc.b_stat_read) {                    /* codec.c:1537-1541 */
        codec->stats_path = stats_path.empty() ? "./x264.stats" : stats_path;
        param.rc.psz_stat_out = param.rc.psz_stat_in = &codec->stats_path[0];
    }
    if (fast1pass && param.rc.b_stat_write && !param.rc.b_stat_read) x264_param_apply_fastfirstpass(&param);      /* codec.c:1580-1581 */
    param.b_vfr_input = 0;                                                  /* VFW supports only CFR (codec.c:1567-1569) */
    param.i_timebase_num = param.i_fps_den; param.i_timebase_den = param.i_fps_num;
    param.vui.b_fullrange = param.vui.b_fullrange == 1;
    if (param.vui.i_colmatrix < 0) param.vui.i_colmatrix = 2;
    codec->colmatrix709 = param.vui.i_colmatrix == 1; codec->fullrange = param.vui.b_fullrange;      /* x264vfw_csp_init (csp.c:452-485) */
    if (x264_param_apply_profile(&param, profile_s.empty() ? nullptr : profile_s.c_str()) < 0) {
        vlog(codec, X264_LOG_ERROR, "x264_param_apply_profile failed\n");
        goto fail;
    }
    if (!codec->b_user_ref) {                                               /* level-based ref clamp (codec.c:1592-1606) */
        int mbs = ((param.i_width + 15) >> 4) * ((param.i_height + 15) >> 4);
        for (int i = 0; x264_levels[i].level_idc != 0; i++)
            if (param.i_level_idc == x264_levels[i].level_idc) {
                while (mbs * param.i_frame_reference > x264_levels[i].dpb && param.i_frame_reference > 1) param.i_frame_reference--;
                break;
            }
    }
    param.b_annexb = 1; param.b_repeat_headers = 1;                         /* VFW needs SPS/PPS before each keyframe */
    if (out_file == "-" && param.i_threads > 1) {
        /* the VfW buffer cannot take late frames (the reference warns "few frames probably would be lost", codec.c:1800-1810);
         * GOP-parallel coding is for file output, where compress_end flushes */
        vlog(codec, X264_LOG_WARNING, "--threads %d needs --output <file> (frames would arrive late): threads 1\n", param.i_threads);
        param.i_threads = 1;
    }
    if (out_file != "-") {                                                  /* select_output (codec.c:1111-1164) */
        codec->b_cli_output = 1;
        codec->largest_pts = codec->second_largest_pts = -1;
        if (!codec->b_no_output) {
            int annexb = 1;
            const char *err = nullptr;
            codec->cli_hout = x264host::open_muxer(out_file.c_str(), muxer.c_str(), &annexb, &err);
            if (!codec->cli_hout) {
                if (err && !strncmp(err, "not compiled", 12)) vlog(codec, X264_LOG_ERROR, "not compiled with this output support (raw, mkv, flv and mp4 are built in)\n");
                else vlog(codec, X264_LOG_ERROR, "could not open output file: '%s'\n", out_file.c_str());
                goto fail;
            }
            param.b_annexb = annexb; param.b_repeat_headers = annexb;       /* containers: length-prefixed NALs, headers once (codec.c:1121-1143) */
        }
    }
    codec->h = x264_encoder_open(&param);
    if (!codec->h) { vlog(codec, X264_LOG_ERROR, "x264_encoder_open failed\n"); goto fail; }
    x264_encoder_parameters(codec->h, &param);
    if (codec->b_cli_output) codec->b_use_vd_hack = 0;                        /* codec.c:1635 */
    codec->b_warn_frame_loss = !(codec->b_use_vd_hack || codec->b_cli_output);  /* codec.c:1665 */
    if (codec->cli_hout) {                                                  /* set_param + write_headers (codec.c:1632-1663) */
        x264_nal_t *hn; int nh;
        if (codec->cli_hout->set_param(&param) < 0 || x264_encoder_headers(codec->h, &hn, &nh) < 0 || (!param.b_repeat_headers && codec->cli_hout->write_headers(hn) < 0)) {
            vlog(codec, X264_LOG_ERROR, "can't write headers to outfile\n");
            goto fail;
        }
    }
    if (x264_picture_alloc(&codec->conv_pic, param.i_csp, param.i_width, param.i_height) < 0) {
        vlog(codec, X264_LOG_ERROR, "x264_picture_alloc failed\n");
        goto fail;
    }
    return ICERR_OK;
fail:
    codec->b_encoder_error = 1;
    compress_end(codec);
    codec->b_encoder_error = 1;
    return ICERR_ERROR;
}

int encode_frame(CODEC *codec, x264_picture_t *pic, x264_picture_t *pic_out, uint8_t *buf, DWORD buf_size, int *got_picture)
{
    x264_nal_t *nal;
    int i_nal;
    *got_picture = 0;
    int size = x264_encoder_encode(codec->h, &nal, &i_nal, pic, pic_out);
    if (size < 0) { vlog(codec, X264_LOG_ERROR, "x264_encoder_encode failed\n"); return -1; }
    if (size) {
        *got_picture = 1;
        if (codec->b_cli_output && pic_out) {
            if (pic_out->i_pts > codec->largest_pts) { codec->second_largest_pts = codec->largest_pts; codec->largest_pts = pic_out->i_pts; }
            else if (pic_out->i_pts > codec->second_largest_pts) codec->second_largest_pts = pic_out->i_pts;
        }
        if (!codec->b_no_output && codec->b_cli_output && codec->cli_hout->write_frame(nal[0].p_payload, size, pic_out) < 0) {   /* cli_output.write_frame */
            vlog(codec, X264_LOG_ERROR, "can't write frame to outfile\n");
            return -1;
        }
        if (!(codec->b_no_output || codec->b_cli_output) && buf) {
            if ((DWORD)size > buf_size && codec->b_check_size) {
                vlog(codec, X264_LOG_ERROR, "output frame buffer too small (size %d / needed %d)\n", (int)buf_size, size);
                return -1;
            }
            memcpy(buf, nal[0].p_payload, size);                            /* NALs are contiguous from nal[0] */
        } else
            size = 0;
    }
    return size;
}

LRESULT compress(CODEC *codec, ICCOMPRESS *icc)
{
    if (!codec->h || codec->b_encoder_error) return ICERR_ERROR;
    BITMAPINFOHEADER *inhdr = icc->lpbiInput, *outhdr = icc->lpbiOutput;
    x264_picture_t pic_out;
    int got_picture, i_out;
    /* "buggy apps" workaround (codec.c:1743-1753) */
    if (codec->prev_lpbiOutput == outhdr && outhdr->biSizeImage < codec->prev_output_biSizeImage) outhdr->biSizeImage = codec->prev_output_biSizeImage;
    codec->prev_lpbiOutput = outhdr;
    codec->prev_output_biSizeImage = outhdr->biSizeImage;
    if (codec->i_frame_remain) {
        if (codec->i_frame_remain != -1) codec->i_frame_remain--;
        int csp = get_csp(inhdr), w = inhdr->biWidth, h = abs(inhdr->biHeight);
        if (csp == CSP_NONE) { vlog(codec, X264_LOG_ERROR, "unknown input frame colorspace\n"); codec->b_encoder_error = 1; return ICERR_BADFORMAT; }
        /* x264vfw_img_fill (codec.c:304-379) over the caller's buffer, then csp.convert[] (codec.c:1774) — on the device:
         * the native frame is uploaded once and converted straight into the encoder's I420 staging buffer */
        long off[3]; int st[3];
        const long n = x264gpu_csp_img_fill(csp, w, h, off, st);
        if (n < 0) { vlog(codec, X264_LOG_ERROR, "unknown input frame colorspace\n"); codec->b_encoder_error = 1; return ICERR_BADFORMAT; }
        if ((size_t)n > codec->raw_cap) {
            if (codec->d_raw) x264gpu_free(codec->d_raw);
            codec->d_raw = nullptr; codec->raw_cap = 0;
            if (x264gpu_malloc((void **)&codec->d_raw, (size_t)n) != X264GPU_OK) { vlog(codec, X264_LOG_ERROR, "device allocation failed\n"); codec->b_encoder_error = 1; return ICERR_MEMORY; }
            codec->raw_cap = (size_t)n;
        }
        uint8_t *d_i420 = x264gpu_host_input_i420(codec->h);
        const uint8_t *src[3] = { codec->d_raw + off[0], codec->d_raw + off[1], codec->d_raw + off[2] };
        uint8_t *dst[3] = { d_i420, d_i420 + (size_t)w * h, d_i420 + (size_t)w * h + (size_t)(w / 2) * (h / 2) };
        const int dst_stride[3] = { w, w / 2, w / 2 };
        if (x264gpu_memcpy_h2d(codec->d_raw, icc->lpInput, (size_t)n, nullptr) != X264GPU_OK ||
            x264gpu_csp_to_i420(src, st, csp, w, h, codec->colmatrix709, codec->fullrange, dst, dst_stride, nullptr) != X264GPU_OK) {
            vlog(codec, X264_LOG_ERROR, "colorspace conversion failed\n");
            codec->b_encoder_error = 1;
            return ICERR_ERROR;
        }
        x264_picture_t pic = codec->conv_pic;                 /* header fields (pts, type) of the running picture */
        pic.img.i_csp = X264_CSP_I420; pic.img.i_plane = 3;
        for (int k = 0; k < 3; k++) { pic.img.plane[k] = dst[k]; pic.img.i_stride[k] = dst_stride[k]; }
        i_out = encode_frame(codec, &pic, &pic_out, (uint8_t *)icc->lpOutput, outhdr->biSizeImage, &got_picture);
        codec->conv_pic.i_pts++;
    } else
        i_out = encode_frame(codec, nullptr, &pic_out, (uint8_t *)icc->lpOutput, outhdr->biSizeImage, &got_picture);
    if (i_out < 0) { codec->b_encoder_error = 1; return ICERR_ERROR; }
    if (!got_picture && codec->b_warn_frame_loss) {                          /* codec.c:1798-1807 */
        codec->b_warn_frame_loss = 0;
        vlog(codec, X264_LOG_WARNING, "Few frames probably would be lost. Ways to fix this:\n");
        vlog(codec, X264_LOG_WARNING, " - if you use VirtualDub or its fork than you can enable 'VirtualDub Hack' option\n");
        vlog(codec, X264_LOG_WARNING, " - you can enable 'File' output mode\n");
        vlog(codec, X264_LOG_WARNING, " - you can enable 'Zero Latency' option\n");
    }
    if (codec->b_use_vd_hack && !got_picture && (outhdr->biSizeImage > 0 || !codec->b_check_size)) {      /* codec.c:1809-1820 */
        /* no picture came back (B pictures / the lookahead hold it): a one-byte drop frame under the XVID fourcc, which VirtualDub and its forks
         * take as "delayed" and answer with as many extra calls at the end of the stream */
        *icc->lpdwFlags = 0;
        ((uint8_t *)icc->lpOutput)[0] = 0x7f;
        outhdr->biSizeImage = 1;
        outhdr->biCompression = mmioFOURCC('X', 'V', 'I', 'D');
    } else {
        *icc->lpdwFlags = got_picture && pic_out.b_keyframe ? AVIIF_KEYFRAME : 0;
        outhdr->biSizeImage = i_out;
        outhdr->biCompression = codec->save_fourcc;                          /* codec.c:1830 */
    }
    return ICERR_OK;
}

}  // namespace

extern "C" LRESULT DriverProc(DWORD_PTR dwDriverId, HDRVR hDriver, UINT uMsg, LPARAM lParam1, LPARAM lParam2)
{
    CODEC *codec = (CODEC *)dwDriverId;
    switch (uMsg) {
    case DRV_LOAD: case DRV_FREE: return DRV_OK;
    case DRV_OPEN: {
        ICOPEN *icopen = (ICOPEN *)lParam2;
        if (icopen && icopen->fccType != ICTYPE_VIDEO) return 0;
        codec = new (std::nothrow) CODEC();
        if (!codec) { if (icopen) icopen->dwError = ICERR_MEMORY; return 0; }
        codec->h = nullptr; codec->b_encoder_error = 0; codec->prev_lpbiOutput = nullptr; codec->prev_output_biSizeImage = 0;
        codec->d_raw = nullptr; codec->raw_cap = 0; codec->colmatrix709 = 0; codec->fullrange = 0;
        codec->cli_hout = nullptr; codec->b_cli_output = 0; codec->b_no_output = 0;
        memset(&codec->conv_pic, 0, sizeof(codec->conv_pic));
        config_defaults(&codec->config);
        codec->i_frame_total = 0; codec->i_fps_num = codec->i_fps_den = 0;
        if (icopen) icopen->dwError = ICERR_OK;
        return (LRESULT)codec;
    }
    case DRV_CLOSE:
        compress_end(codec);                     /* compress_end doesn't always get called by hosts (driverproc.c:131-139) */
        delete codec;
        return DRV_OK;
    case DRV_QUERYCONFIGURE: return 0;
    case DRV_CONFIGURE: return DRV_CANCEL;
    case ICM_GETSTATE:
        if (!lParam1) return sizeof(X264VFW_CONFIG);
        if ((size_t)lParam2 != sizeof(X264VFW_CONFIG)) return ICERR_BADSIZE;
        memcpy((void *)lParam1, &codec->config, sizeof(X264VFW_CONFIG));
        ((X264VFW_CONFIG *)lParam1)->i_format_version = X264VFW_FORMAT_VERSION;
        return ICERR_OK;
    case ICM_SETSTATE:
        if (!lParam1) { config_defaults(&codec->config); return 0; }
        if ((size_t)lParam2 != sizeof(X264VFW_CONFIG) || ((X264VFW_CONFIG *)lParam1)->i_format_version != X264VFW_FORMAT_VERSION) return 0;
        memcpy(&codec->config, (void *)lParam1, sizeof(X264VFW_CONFIG));
        return sizeof(X264VFW_CONFIG);
    case ICM_GETINFO: {
        ICINFO *ii = (ICINFO *)lParam1;
        if ((size_t)lParam2 < sizeof(ICINFO)) return 0;
        memset(ii, 0, sizeof(*ii));
        ii->dwSize = sizeof(ICINFO); ii->fccType = ICTYPE_VIDEO; ii->fccHandler = mmioFOURCC('X', '2', '6', '4');
        ii->dwFlags = VIDCF_COMPRESSFRAMES | VIDCF_FASTTEMPORALC; ii->dwVersion = 0; ii->dwVersionICM = ICVERSION;
        const char *nm = "x264vfw", *ds = "x264vfw - H.264/MPEG-4 AVC codec";
        for (int i = 0; nm[i] && i < 15; i++) ii->szName[i] = (uint16_t)nm[i];
        for (int i = 0; ds[i] && i < 127; i++) ii->szDescription[i] = (uint16_t)ds[i];
        return sizeof(ICINFO);
    }
    case ICM_CONFIGURE: case ICM_ABOUT: return ICERR_OK;       /* dialogs are out of scope; lParam1 == -1 is the capability query */
    case ICM_GET: return lParam1 ? ICERR_OK : 0;
    case ICM_SET: return 0;
    case ICM_COMPRESS_GET_FORMAT: return compress_get_format(codec, (BITMAPINFO *)lParam1, (BITMAPINFO *)lParam2);
    case ICM_COMPRESS_GET_SIZE: return compress_get_size((BITMAPINFO *)lParam2);
    case ICM_COMPRESS_QUERY: return compress_query(codec, (BITMAPINFO *)lParam1, (BITMAPINFO *)lParam2);
    case ICM_COMPRESS_BEGIN: return compress_begin(codec, (BITMAPINFO *)lParam1, (BITMAPINFO *)lParam2);
    case ICM_COMPRESS: return compress(codec, (ICCOMPRESS *)lParam1);
    case ICM_COMPRESS_END:
        codec->i_frame_total = 0; codec->i_fps_num = codec->i_fps_den = 0;
        return compress_end(codec);
    case ICM_COMPRESS_FRAMES_INFO: {
        ICCOMPRESSFRAMES *icf = (ICCOMPRESSFRAMES *)lParam1;
        codec->i_frame_total = icf->lFrameCount; codec->i_fps_num = icf->dwRate; codec->i_fps_den = icf->dwScale;
        return ICERR_OK;
    }
    default:
        return uMsg < DRV_USER ? 0 /* DefDriverProc */ : ICERR_UNSUPPORTED;
    }
}

/* test hook: the driver's log text of the last session (the Windows build shows it in a list box) */
extern "C" const char *x264vfw_shim_log(DWORD_PTR dwDriverId) { return dwDriverId ? ((CODEC *)dwDriverId)->log.c_str() : ""; }
