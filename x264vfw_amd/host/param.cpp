// param.cpp — x264_param_* API as consumed by the x264vfw driver (boundary B1).
//   x264_param_default            config.c:1410         defaults printed by the help text config.c:1544-1775
//   x264_param_default_preset     codec.c:1463          preset/tune deltas exactly as listed at config.c:1460-1527
//   x264_param_parse              codec.c:1349          option names = long_options[] of codec.c:831-999
//   x264_param_apply_fastfirstpass codec.c:1581         config.c:1535-1538
//   x264_param_apply_profile      codec.c:1584          profiles listed at config.c:1440-1456
//   x264_levels                   codec.c:1596-1599     ITU-T H.264 Table A-1
// Every option the driver can forward is accepted; options that configure tools the MI355X pipeline
// does not implement yet are stored (x264_encoder_parameters returns the *effective* values).
#include "host.hpp"
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

extern "C" {

const x264_level_t x264_levels[] = {
    { 10, 1485, 99, 396, 64, 175, 64 },       { 9, 1485, 99, 396, 128, 350, 64 },        /* "1b" */
    { 11, 3000, 396, 900, 192, 500, 128 },    { 12, 6000, 396, 2376, 384, 1000, 128 },
    { 13, 11880, 396, 2376, 768, 2000, 128 }, { 20, 11880, 396, 2376, 2000, 2000, 128 },
    { 21, 19800, 792, 4752, 4000, 4000, 256 }, { 22, 20250, 1620, 8100, 4000, 4000, 256 },
    { 30, 40500, 1620, 8100, 10000, 10000, 256 }, { 31, 108000, 3600, 18000, 14000, 14000, 512 },
    { 32, 216000, 5120, 20480, 20000, 20000, 512 }, { 40, 245760, 8192, 32768, 20000, 25000, 512 },
    { 41, 245760, 8192, 32768, 50000, 62500, 512 }, { 42, 522240, 8704, 34816, 50000, 62500, 512 },
    { 50, 589824, 22080, 110400, 135000, 135000, 512 }, { 51, 983040, 36864, 184320, 240000, 240000, 512 },
    { 52, 2073600, 36864, 184320, 240000, 240000, 512 }, { 60, 4177920, 139264, 696320, 240000, 240000, 8192 },
    { 61, 8355840, 139264, 696320, 480000, 480000, 8192 }, { 62, 16711680, 139264, 696320, 800000, 800000, 8192 },
    { 0, 0, 0, 0, 0, 0, 0 } };

void x264_param_default(x264_param_t *p)
{
    memset(p, 0, sizeof(*p));
    p->cpu = 1;
    p->i_threads = 0;
    p->b_deterministic = 1;
    p->i_csp = X264_CSP_I420;
    p->i_level_idc = -1;
    p->vui.i_overscan = 0; p->vui.i_vidformat = 5; p->vui.b_fullrange = -1;
    p->vui.i_colorprim = 2; p->vui.i_transfer = 2; p->vui.i_colmatrix = -1;
    p->i_fps_num = 25; p->i_fps_den = 1; p->i_timebase_num = 0; p->i_timebase_den = 0;
    p->i_frame_reference = 3;
    p->i_keyint_max = 250; p->i_keyint_min = 0;                   /* 0 = auto */
    p->i_scenecut_threshold = 40;
    p->i_bframe = 3; p->i_bframe_adaptive = X264_B_ADAPT_FAST; p->i_bframe_bias = 0; p->i_bframe_pyramid = X264_B_PYRAMID_NORMAL;
    p->b_deblocking_filter = 1;
    p->b_cabac = 1; p->i_cabac_init_idc = 0;
    p->i_log_level = X264_LOG_INFO;
    p->analyse.intra = X264_ANALYSE_I4x4 | X264_ANALYSE_I8x8;
    p->analyse.inter = X264_ANALYSE_I4x4 | X264_ANALYSE_I8x8 | X264_ANALYSE_PSUB16x16 | X264_ANALYSE_BSUB16x16;
    p->analyse.b_transform_8x8 = 1;
    p->analyse.i_weighted_pred = X264_WEIGHTP_SMART; p->analyse.b_weighted_bipred = 1;
    p->analyse.i_direct_mv_pred = X264_DIRECT_PRED_SPATIAL;
    p->analyse.i_me_method = X264_ME_HEX; p->analyse.i_me_range = 16; p->analyse.i_mv_range = -1; p->analyse.i_mv_range_thread = -1;
    p->analyse.i_subpel_refine = 7;
    p->analyse.b_chroma_me = 1; p->analyse.b_mixed_references = 1; p->analyse.i_trellis = 1; p->analyse.b_fast_pskip = 1;
    p->analyse.b_dct_decimate = 1;
    p->analyse.f_psy_rd = 1.0f; p->analyse.f_psy_trellis = 0.0f; p->analyse.b_psy = 1;
    p->analyse.i_luma_deadzone[0] = 21; p->analyse.i_luma_deadzone[1] = 11;
    p->rc.i_rc_method = X264_RC_CRF;
    p->rc.i_qp_constant = 23; p->rc.i_qp_min = 0; p->rc.i_qp_max = 69; p->rc.i_qp_step = 4;
    p->rc.f_rf_constant = 23.0f; p->rc.f_rate_tolerance = 1.0f;
    p->rc.f_vbv_buffer_init = 0.9f; p->rc.f_ip_factor = 1.4f; p->rc.f_pb_factor = 1.3f;
    p->rc.i_aq_mode = X264_AQ_VARIANCE; p->rc.f_aq_strength = 1.0f;
    p->rc.b_mb_tree = 1; p->rc.i_lookahead = 40;
    p->rc.f_qcompress = 0.6f; p->rc.f_qblur = 0.5f; p->rc.f_complexity_blur = 20.0f;
    p->b_repeat_headers = 1; p->b_annexb = 1;
    p->b_vfr_input = 1;
    p->i_frame_packing = -1;
    p->i_slice_count = 0;
}

static int apply_preset(x264_param_t *p, const char *preset)
{
    static const char *const names[] = { "ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo", 0 };
    char *end;
    long idx = strtol(preset, &end, 10);
    if (*end == 0 && idx >= 0 && idx < 10) preset = names[idx];
    if (!strcasecmp(preset, "ultrafast")) {
        p->analyse.b_transform_8x8 = 0; p->rc.i_aq_mode = 0; p->i_bframe_adaptive = X264_B_ADAPT_NONE; p->i_bframe = 0;
        p->b_cabac = 0; p->b_deblocking_filter = 0; p->rc.b_mb_tree = 0; p->analyse.i_me_method = X264_ME_DIA;
        p->analyse.b_mixed_references = 0; p->analyse.inter = 0; p->analyse.intra = 0; p->rc.i_lookahead = 0; p->i_frame_reference = 1;
        p->i_scenecut_threshold = 0; p->analyse.i_subpel_refine = 0; p->analyse.i_trellis = 0; p->analyse.b_weighted_bipred = 0;
        p->analyse.i_weighted_pred = X264_WEIGHTP_NONE;
    } else if (!strcasecmp(preset, "superfast")) {
        p->rc.b_mb_tree = 0; p->analyse.i_me_method = X264_ME_DIA; p->analyse.b_mixed_references = 0;
        p->analyse.inter = X264_ANALYSE_I8x8 | X264_ANALYSE_I4x4; p->rc.i_lookahead = 0; p->i_frame_reference = 1;
        p->analyse.i_subpel_refine = 1; p->analyse.i_trellis = 0; p->analyse.i_weighted_pred = X264_WEIGHTP_SIMPLE;
    } else if (!strcasecmp(preset, "veryfast")) {
        p->analyse.b_mixed_references = 0; p->rc.i_lookahead = 10; p->i_frame_reference = 1; p->analyse.i_subpel_refine = 2;
        p->analyse.i_trellis = 0; p->analyse.i_weighted_pred = X264_WEIGHTP_SIMPLE;
    } else if (!strcasecmp(preset, "faster")) {
        p->analyse.b_mixed_references = 0; p->rc.i_lookahead = 20; p->i_frame_reference = 2; p->analyse.i_subpel_refine = 4;
        p->analyse.i_weighted_pred = X264_WEIGHTP_SIMPLE;
    } else if (!strcasecmp(preset, "fast")) {
        p->rc.i_lookahead = 30; p->i_frame_reference = 2; p->analyse.i_subpel_refine = 6; p->analyse.i_weighted_pred = X264_WEIGHTP_SIMPLE;
    } else if (!strcasecmp(preset, "medium")) {
    } else if (!strcasecmp(preset, "slow")) {
        p->analyse.i_direct_mv_pred = X264_DIRECT_PRED_AUTO; p->rc.i_lookahead = 50; p->i_frame_reference = 5;
        p->analyse.i_subpel_refine = 8; p->analyse.i_trellis = 2;
    } else if (!strcasecmp(preset, "slower")) {
        p->i_bframe_adaptive = X264_B_ADAPT_TRELLIS; p->analyse.i_direct_mv_pred = X264_DIRECT_PRED_AUTO; p->analyse.i_me_method = X264_ME_UMH;
        p->analyse.inter |= X264_ANALYSE_PSUB8x8; p->rc.i_lookahead = 60; p->i_frame_reference = 8; p->analyse.i_subpel_refine = 9;
        p->analyse.i_trellis = 2;
    } else if (!strcasecmp(preset, "veryslow")) {
        p->i_bframe_adaptive = X264_B_ADAPT_TRELLIS; p->i_bframe = 8; p->analyse.i_direct_mv_pred = X264_DIRECT_PRED_AUTO;
        p->analyse.i_me_method = X264_ME_UMH; p->analyse.i_me_range = 24; p->analyse.inter |= X264_ANALYSE_PSUB8x8;
        p->i_frame_reference = 16; p->analyse.i_subpel_refine = 10; p->analyse.i_trellis = 2; p->rc.i_lookahead = 60;
    } else if (!strcasecmp(preset, "placebo")) {
        p->i_bframe = 16; p->i_bframe_adaptive = X264_B_ADAPT_TRELLIS; p->analyse.i_direct_mv_pred = X264_DIRECT_PRED_AUTO;
        p->analyse.b_fast_pskip = 0; p->analyse.i_me_method = X264_ME_TESA; p->analyse.i_me_range = 24;
        p->analyse.inter |= X264_ANALYSE_PSUB8x8; p->rc.i_lookahead = 60; p->i_frame_reference = 16;
        p->analyse.i_subpel_refine = 11; p->analyse.i_trellis = 2;
    } else
        return -1;
    return 0;
}

static int apply_tune(x264_param_t *p, const char *tune)
{
    char *tmp = strdup(tune), *save = 0;
    int psy_set = 0, rc = 0;
    for (char *s = strtok_r(tmp, ",./-+", &save); s && !rc; s = strtok_r(0, ",./-+", &save)) {
        int psy = 1;
        if (!strncasecmp(s, "film", 4)) { p->i_deblocking_filter_alphac0 = -1; p->i_deblocking_filter_beta = -1; p->analyse.f_psy_trellis = 0.15f; }
        else if (!strncasecmp(s, "animation", 9)) {
            p->i_frame_reference = p->i_frame_reference > 1 ? p->i_frame_reference * 2 : 1;
            p->i_deblocking_filter_alphac0 = 1; p->i_deblocking_filter_beta = 1; p->analyse.f_psy_rd = 0.4f;
            p->rc.f_aq_strength = 0.6f; p->i_bframe += 2;
        } else if (!strncasecmp(s, "grain", 5)) {
            p->i_deblocking_filter_alphac0 = -2; p->i_deblocking_filter_beta = -2; p->analyse.f_psy_trellis = 0.25f;
            p->analyse.b_dct_decimate = 0; p->rc.f_pb_factor = 1.1f; p->rc.f_ip_factor = 1.1f; p->rc.f_aq_strength = 0.5f;
            p->analyse.i_luma_deadzone[0] = 6; p->analyse.i_luma_deadzone[1] = 6; p->rc.f_qcompress = 0.8f;
        } else if (!strncasecmp(s, "stillimage", 10)) {
            p->i_deblocking_filter_alphac0 = -3; p->i_deblocking_filter_beta = -3; p->analyse.f_psy_rd = 2.0f;
            p->analyse.f_psy_trellis = 0.7f; p->rc.f_aq_strength = 1.2f;
        } else if (!strncasecmp(s, "psnr", 4)) { p->rc.i_aq_mode = X264_AQ_NONE; p->analyse.b_psy = 0; }
        else if (!strncasecmp(s, "ssim", 4)) { p->rc.i_aq_mode = X264_AQ_AUTOVARIANCE; p->analyse.b_psy = 0; }
        else if (!strncasecmp(s, "fastdecode", 10)) {
            psy = 0; p->b_deblocking_filter = 0; p->b_cabac = 0; p->analyse.b_weighted_bipred = 0; p->analyse.i_weighted_pred = X264_WEIGHTP_NONE;
        } else if (!strncasecmp(s, "zerolatency", 11)) {
            psy = 0; p->rc.i_lookahead = 0; p->i_bframe = 0; p->b_vfr_input = 0; p->rc.b_mb_tree = 0; p->b_sliced_threads = 1;
        } else { rc = -1; break; }
        if (psy && psy_set++) rc = -1;          /* only one psy tuning at a time (config.c:1509) */
    }
    free(tmp);
    return rc;
}

int x264_param_default_preset(x264_param_t *p, const char *preset, const char *tune)
{
    x264_param_default(p);
    if (preset && apply_preset(p, preset) < 0) return -1;
    if (tune && apply_tune(p, tune) < 0) return -1;
    return 0;
}

void x264_param_apply_fastfirstpass(x264_param_t *p)
{
    if (!p->rc.b_stat_write || p->rc.b_stat_read) return;     /* first pass only */
    p->i_frame_reference = 1;
    p->analyse.b_transform_8x8 = 0;
    p->analyse.inter = 0;
    p->analyse.i_me_method = X264_ME_DIA;
    if (p->analyse.i_subpel_refine > 2) p->analyse.i_subpel_refine = 2;
    p->analyse.i_trellis = 0;
    p->analyse.b_fast_pskip = 1;
}

int x264_param_apply_profile(x264_param_t *p, const char *profile)
{
    if (!profile) return 0;
    enum { BASELINE, MAIN, HIGH, HIGH10, HIGH422, HIGH444 } prof;
    if (!strcasecmp(profile, "baseline")) prof = BASELINE;
    else if (!strcasecmp(profile, "main")) prof = MAIN;
    else if (!strcasecmp(profile, "high")) prof = HIGH;
    else if (!strcasecmp(profile, "high10")) prof = HIGH10;
    else if (!strcasecmp(profile, "high422")) prof = HIGH422;
    else if (!strcasecmp(profile, "high444")) prof = HIGH444;
    else return -1;
    if (prof < HIGH444 && p->rc.i_rc_method == X264_RC_CQP && p->rc.i_qp_constant <= 0) return -1;   /* lossless needs high444 */
    if (prof <= MAIN) p->analyse.b_transform_8x8 = 0;
    if (prof == BASELINE) {
        p->analyse.b_transform_8x8 = 0; p->b_cabac = 0; p->i_bframe = 0; p->analyse.i_weighted_pred = X264_WEIGHTP_NONE;
        if (p->b_interlaced) return -1;
    }
    return 0;
}

/* ---- option parsing ---- */
static int parse_bool(const char *v, int *err)
{
    if (!v || !strcasecmp(v, "1") || !strcasecmp(v, "true") || !strcasecmp(v, "yes")) return 1;
    if (!strcasecmp(v, "0") || !strcasecmp(v, "false") || !strcasecmp(v, "no")) return 0;
    *err = 1;
    return 0;
}
static int parse_int(const char *v, int *err)
{
    if (!v) { *err = 1; return 0; }
    char *end;
    long r = strtol(v, &end, 0);
    if (end == v || *end) *err = 1;
    return (int)r;
}
static double parse_float(const char *v, int *err)
{
    if (!v) { *err = 1; return 0; }
    char *end;
    double r = strtod(v, &end);
    if (end == v || *end) *err = 1;
    return r;
}
static int parse_enum(const char *v, const char *const *names, int *dst)
{
    if (!v) return -1;
    for (int i = 0; names[i]; i++)
        if (names[i][0] && !strcasecmp(v, names[i])) { *dst = i; return 0; }
    return -1;
}

int x264_param_parse(x264_param_t *p, const char *name, const char *value)
{
    if (!name) return X264_PARAM_BAD_NAME;
    char nm[64];
    size_t n = strlen(name);
    if (n >= sizeof(nm)) return X264_PARAM_BAD_NAME;
    for (size_t i = 0; i <= n; i++) nm[i] = name[i] == '_' ? '-' : name[i];
    const char *o = nm;
    int neg = 0, err = 0;
    if (!strncmp(o, "no-", 3)) { neg = 1; o += 3; }
    else if (!strncmp(o, "no", 2) && strcmp(o, "nr") && strcmp(o, "non-deterministic") && strcmp(o, "nal-hrd")) { neg = 1; o += 2; }
    if (neg) value = (value && (!strcasecmp(value, "0") || !strcasecmp(value, "false") || !strcasecmp(value, "no"))) ? "1" : "0";
#define OPT(s) else if (!strcmp(o, s))
#define B(field) do { (field) = parse_bool(value, &err); } while (0)
#define I(field) do { (field) = parse_int(value, &err); } while (0)
#define F(field) do { (field) = (float)parse_float(value, &err); } while (0)
    if (0) ;
    OPT("asm") { int v = parse_bool(value, &err); p->cpu = v ? 1 : 0; }
    OPT("threads") { if (value && !strcasecmp(value, "auto")) p->i_threads = 0; else I(p->i_threads); }
    OPT("lookahead-threads") { if (!(value && !strcasecmp(value, "auto"))) (void)parse_int(value, &err); }
    OPT("sliced-threads") B(p->b_sliced_threads);
    OPT("sync-lookahead") { if (!(value && !strcasecmp(value, "auto"))) (void)parse_int(value, &err); }
    OPT("deterministic") B(p->b_deterministic);
    OPT("non-deterministic") { p->b_deterministic = !parse_bool(value, &err); }
    OPT("cpu-independent") (void)parse_bool(value, &err);
    OPT("level") {
        if (value && !strcmp(value, "1b")) p->i_level_idc = 9;
        else { double v = parse_float(value, &err); p->i_level_idc = v < 7 ? (int)(10 * v + .5) : (int)v; }
    }
    OPT("bluray-compat") B(p->b_bluray_compat);
    OPT("avcintra-class") (void)parse_int(value, &err);
    OPT("sar") { if (!value || (sscanf(value, "%d:%d", &p->vui.i_sar_width, &p->vui.i_sar_height) != 2 && sscanf(value, "%d/%d", &p->vui.i_sar_width, &p->vui.i_sar_height) != 2)) err = 1; }
    OPT("overscan") err |= parse_enum(value, x264_overscan_names, &p->vui.i_overscan) < 0;
    OPT("videoformat") err |= parse_enum(value, x264_vidformat_names, &p->vui.i_vidformat) < 0;
    OPT("fullrange") B(p->vui.b_fullrange);
    OPT("colorprim") err |= parse_enum(value, x264_colorprim_names, &p->vui.i_colorprim) < 0;
    OPT("transfer") err |= parse_enum(value, x264_transfer_names, &p->vui.i_transfer) < 0;
    OPT("colormatrix") err |= parse_enum(value, x264_colmatrix_names, &p->vui.i_colmatrix) < 0;
    OPT("chromaloc") { I(p->vui.i_chroma_loc); err |= p->vui.i_chroma_loc < 0 || p->vui.i_chroma_loc > 5; }
    OPT("fps") {
        if (!value) err = 1;
        else if (sscanf(value, "%u/%u", &p->i_fps_num, &p->i_fps_den) != 2) {
            double f = parse_float(value, &err);
            if (f > 0 && f <= 4294967.0) { p->i_fps_num = (uint32_t)(f * 1000 + .5); p->i_fps_den = 1000; } else err = 1;
        }
    }
    OPT("ref") I(p->i_frame_reference);
    OPT("keyint") { if (value && strstr(value, "infinite")) p->i_keyint_max = X264_KEYINT_MAX_INFINITE; else I(p->i_keyint_max); }
    OPT("min-keyint") I(p->i_keyint_min);
    OPT("scenecut") { if (!value || !strcmp(value, "1") || neg) { p->i_scenecut_threshold = parse_bool(value, &err) ? 40 : 0; } else I(p->i_scenecut_threshold); }
    OPT("intra-refresh") B(p->b_intra_refresh);
    OPT("bframes") I(p->i_bframe);
    OPT("b-adapt") { if (neg || !value) p->i_bframe_adaptive = parse_bool(value, &err); else I(p->i_bframe_adaptive); }
    OPT("b-bias") I(p->i_bframe_bias);
    OPT("b-pyramid") { if (parse_enum(value, x264_b_pyramid_names, &p->i_bframe_pyramid) < 0) I(p->i_bframe_pyramid); }
    OPT("open-gop") B(p->b_open_gop);
    OPT("nf") { p->b_deblocking_filter = !parse_bool(value, &err); }
    OPT("deblock") { p->b_deblocking_filter = parse_bool(value, &err); err = 0; if (value && strcmp(value, "0") && strcmp(value, "1")) { /* a:b form */
            int a = 0, b = 0, k = sscanf(value, "%d:%d", &a, &b); if (k < 1) k = sscanf(value, "%d,%d", &a, &b);
            if (k >= 1) { p->b_deblocking_filter = 1; p->i_deblocking_filter_alphac0 = a; p->i_deblocking_filter_beta = k == 2 ? b : a; } else err = 1; } }
    OPT("filter") { int a = 0, b = 0, k = value ? sscanf(value, "%d:%d", &a, &b) : 0; if (k >= 1) { p->b_deblocking_filter = 1; p->i_deblocking_filter_alphac0 = a; p->i_deblocking_filter_beta = k == 2 ? b : a; } else err = 1; }
    OPT("slice-max-size") I(p->i_slice_max_size);
    OPT("slice-max-mbs") I(p->i_slice_max_mbs);
    OPT("slice-min-mbs") (void)parse_int(value, &err);
    OPT("slices") I(p->i_slice_count);
    OPT("slices-max") (void)parse_int(value, &err);
    OPT("cabac") B(p->b_cabac);
    OPT("cabac-idc") I(p->i_cabac_init_idc);
    OPT("interlaced") B(p->b_interlaced);
    OPT("tff") B(p->b_interlaced);
    OPT("bff") B(p->b_interlaced);
    OPT("fake-interlaced") B(p->b_fake_interlaced);
    OPT("constrained-intra") B(p->b_constrained_intra);
    OPT("cqm") { if (!value || (strcmp(value, "flat") && strcmp(value, "jvt"))) err = 1; else if (!strcmp(value, "jvt")) err = 1; /* flat only */ }
    OPT("cqmfile") err = 1;
    OPT("cqm4") err = 1; OPT("cqm4i") err = 1; OPT("cqm4iy") err = 1; OPT("cqm4ic") err = 1; OPT("cqm4p") err = 1;
    OPT("cqm4py") err = 1; OPT("cqm4pc") err = 1; OPT("cqm8") err = 1; OPT("cqm8i") err = 1; OPT("cqm8p") err = 1;
    OPT("log") I(p->i_log_level);
    OPT("dump-yuv") {}
    OPT("analyse") { goto partitions; }
    OPT("partitions") {
    partitions:
        if (!value) err = 1;
        else {
            p->analyse.inter = 0;
            if (strstr(value, "none")) p->analyse.inter = 0;
            if (strstr(value, "all")) p->analyse.inter = ~0u;
            if (strstr(value, "i4x4")) p->analyse.inter |= X264_ANALYSE_I4x4;
            if (strstr(value, "i8x8")) p->analyse.inter |= X264_ANALYSE_I8x8;
            if (strstr(value, "p8x8")) p->analyse.inter |= X264_ANALYSE_PSUB16x16;
            if (strstr(value, "p4x4")) p->analyse.inter |= X264_ANALYSE_PSUB8x8;
            if (strstr(value, "b8x8")) p->analyse.inter |= X264_ANALYSE_BSUB16x16;
        }
    }
    OPT("8x8dct") B(p->analyse.b_transform_8x8);
    OPT("weightb") B(p->analyse.b_weighted_bipred);
    OPT("weight-b") B(p->analyse.b_weighted_bipred);
    OPT("weightp") I(p->analyse.i_weighted_pred);
    OPT("direct") err |= parse_enum(value, x264_direct_pred_names, &p->analyse.i_direct_mv_pred) < 0;
    OPT("direct-pred") err |= parse_enum(value, x264_direct_pred_names, &p->analyse.i_direct_mv_pred) < 0;
    OPT("chroma-qp-offset") I(p->analyse.i_chroma_qp_offset);
    OPT("me") err |= parse_enum(value, x264_motion_est_names, &p->analyse.i_me_method) < 0;
    OPT("merange") I(p->analyse.i_me_range);
    OPT("me-range") I(p->analyse.i_me_range);
    OPT("mvrange") I(p->analyse.i_mv_range);
    OPT("mv-range") I(p->analyse.i_mv_range);
    OPT("mvrange-thread") I(p->analyse.i_mv_range_thread);
    OPT("subme") I(p->analyse.i_subpel_refine);
    OPT("subq") I(p->analyse.i_subpel_refine);
    OPT("psy-rd") { if (!value || (sscanf(value, "%f:%f", &p->analyse.f_psy_rd, &p->analyse.f_psy_trellis) < 1 && sscanf(value, "%f,%f", &p->analyse.f_psy_rd, &p->analyse.f_psy_trellis) < 1)) err = 1; }
    OPT("psy") B(p->analyse.b_psy);
    OPT("chroma-me") B(p->analyse.b_chroma_me);
    OPT("mixed-refs") B(p->analyse.b_mixed_references);
    OPT("trellis") I(p->analyse.i_trellis);
    OPT("fast-pskip") B(p->analyse.b_fast_pskip);
    OPT("dct-decimate") B(p->analyse.b_dct_decimate);
    OPT("deadzone-inter") I(p->analyse.i_luma_deadzone[0]);
    OPT("deadzone-intra") I(p->analyse.i_luma_deadzone[1]);
    OPT("nr") I(p->analyse.i_noise_reduction);
    OPT("bitrate") { I(p->rc.i_bitrate); p->rc.i_rc_method = X264_RC_ABR; }
    OPT("qp") { I(p->rc.i_qp_constant); p->rc.i_rc_method = X264_RC_CQP; }
    OPT("qp-constant") { I(p->rc.i_qp_constant); p->rc.i_rc_method = X264_RC_CQP; }
    OPT("crf") { F(p->rc.f_rf_constant); p->rc.i_rc_method = X264_RC_CRF; }
    OPT("crf-max") F(p->rc.f_rf_constant_max);
    OPT("rc-lookahead") I(p->rc.i_lookahead);
    OPT("qpmin") I(p->rc.i_qp_min);
    OPT("qpmax") I(p->rc.i_qp_max);
    OPT("qpstep") I(p->rc.i_qp_step);
    OPT("ratetol") { if (value && !strncmp(value, "inf", 3)) p->rc.f_rate_tolerance = 1e9f; else F(p->rc.f_rate_tolerance); }
    OPT("vbv-maxrate") I(p->rc.i_vbv_max_bitrate);
    OPT("vbv-bufsize") I(p->rc.i_vbv_buffer_size);
    OPT("vbv-init") F(p->rc.f_vbv_buffer_init);
    OPT("ipratio") F(p->rc.f_ip_factor);
    OPT("pbratio") F(p->rc.f_pb_factor);
    OPT("aq-mode") I(p->rc.i_aq_mode);
    OPT("aq-strength") F(p->rc.f_aq_strength);
    OPT("pass") { int v = parse_int(value, &err); if (v < 0 || v > 3) err = 1; else { p->rc.b_stat_write = v & 1; p->rc.b_stat_read = v & 2; } }
    OPT("stats") { p->rc.psz_stat_in = strdup(value); p->rc.psz_stat_out = strdup(value); }      /* (x264_param_parse; the driver sets both itself, codec.c:1537-1541) */
    OPT("qcomp") F(p->rc.f_qcompress);
    OPT("mbtree") B(p->rc.b_mb_tree);
    OPT("qblur") F(p->rc.f_qblur);
    OPT("cplxblur") F(p->rc.f_complexity_blur);
    OPT("zones") { p->rc.psz_zones = value ? strdup(value) : nullptr; }
    OPT("psnr") B(p->analyse.b_psnr);
    OPT("ssim") B(p->analyse.b_ssim);
    OPT("aud") B(p->b_aud);
    OPT("sps-id") I(p->i_sps_id);
    OPT("global-header") { p->b_repeat_headers = !parse_bool(value, &err); }
    OPT("repeat-headers") B(p->b_repeat_headers);
    OPT("annexb") B(p->b_annexb);
    OPT("force-cfr") { p->b_vfr_input = !parse_bool(value, &err); }
    OPT("nal-hrd") { static const char *const names[] = { "none", "vbr", "cbr", 0 }; err |= parse_enum(value, names, &p->i_nal_hrd) < 0; }
    OPT("filler") (void)parse_bool(value, &err);
    OPT("pic-struct") B(p->b_pic_struct);
    OPT("crop-rect") {}
    OPT("frame-packing") I(p->i_frame_packing);
    OPT("stitchable") B(p->b_stitchable);
    OPT("verbose") { p->i_log_level = X264_LOG_DEBUG; }
    OPT("progress") {}
    OPT("stdout") {} OPT("stdin") {}
    else return X264_PARAM_BAD_NAME;
#undef OPT
#undef B
#undef I
#undef F
    return err ? X264_PARAM_BAD_VALUE : 0;
}

/* ---- pictures ---- */
void x264_picture_init(x264_picture_t *pic)
{
    memset(pic, 0, sizeof(*pic));
    pic->i_type = X264_TYPE_AUTO;
}

int x264_picture_alloc(x264_picture_t *pic, int i_csp, int w, int h)
{
    x264_picture_init(pic);
    int csp = i_csp & X264_CSP_MASK;
    if (csp != X264_CSP_I420 && csp != X264_CSP_YV12 && csp != X264_CSP_NV12) return -1;
    if (w <= 0 || h <= 0) return -1;
    pic->img.i_csp = i_csp;
    pic->img.i_plane = csp == X264_CSP_NV12 ? 2 : 3;
    size_t ysz = (size_t)w * h, csz = (size_t)((w + 1) / 2) * ((h + 1) / 2);
    uint8_t *buf = (uint8_t *)malloc(ysz + 2 * csz + 64);
    if (!buf) return -1;
    pic->img.plane[0] = buf; pic->img.i_stride[0] = w;
    if (csp == X264_CSP_NV12) { pic->img.plane[1] = buf + ysz; pic->img.i_stride[1] = 2 * ((w + 1) / 2); }
    else {
        pic->img.plane[1] = buf + ysz; pic->img.plane[2] = buf + ysz + csz;
        pic->img.i_stride[1] = pic->img.i_stride[2] = (w + 1) / 2;
    }
    return 0;
}

void x264_picture_clean(x264_picture_t *pic)
{
    /* safe on a zeroed struct: codec.c:1872-1873 calls it on every end, even if never allocated */
    free(pic->img.plane[0]);
    memset(pic, 0, sizeof(*pic));
}

}  /* extern "C" */
