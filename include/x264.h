/* include/x264.h — boundary B1: the subset of the libx264 C API that the x264vfw driver consumes
 * (SURVEY.md §8b), re-declared for the MI355X encoder.  This is NOT x264's header: libx264 is absent
 * from the reference tree (/root/reference/Makefile:21-23), so every type, field and function below is
 * derived from the reference's own call sites, cited per item.  Source compatibility with the driver
 * is the goal (same names, argument meaning and error behaviour), not binary compatibility.
 */
#ifndef X264GPU_X264_H
#define X264GPU_X264_H
#include <stdarg.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X264_BUILD 157                 /* era inferred in SURVEY.md §0; printed by config.c:521,1172 */
#define X264_VERSION " mi355x-r1"      /* output/matroska.c:160 appends it to the writing-app string */
#define X264_BIT_DEPTH 8               /* config.c:1393-1399 */
#define X264_CHROMA_FORMAT 1           /* 4:2:0 only */

/* colourspaces: ordering matters — ">= X264_CSP_BGR" means the RGB family (codec.c:1571,1574) */
#define X264_CSP_MASK 0x00ff
#define X264_CSP_NONE 0x0000
#define X264_CSP_I420 0x0001           /* codec.c:276 */
#define X264_CSP_YV12 0x0002
#define X264_CSP_NV12 0x0003
#define X264_CSP_I422 0x0005
#define X264_CSP_I444 0x0008
#define X264_CSP_BGR  0x000a
#define X264_CSP_BGRA 0x000b
#define X264_CSP_VFLIP 0x1000

#define X264_RC_CQP 0                  /* codec.c:1493-1517 */
#define X264_RC_CRF 1
#define X264_RC_ABR 2

#define X264_LOG_NONE (-1)             /* codec.c:680-692,1274-1283 */
#define X264_LOG_ERROR 0
#define X264_LOG_WARNING 1
#define X264_LOG_INFO 2
#define X264_LOG_DEBUG 3

#define X264_TYPE_AUTO 0x0000          /* codec.c:1783; output/matroska.c:201 */
#define X264_TYPE_IDR 0x0001
#define X264_TYPE_I 0x0002
#define X264_TYPE_P 0x0003
#define X264_TYPE_BREF 0x0004
#define X264_TYPE_B 0x0005
#define X264_TYPE_KEYFRAME 0x0006
#define IS_X264_TYPE_I(x) ((x) == X264_TYPE_I || (x) == X264_TYPE_IDR || (x) == X264_TYPE_KEYFRAME)

#define X264_NAL_HRD_NONE 0            /* codec.c:1126-1129 */
#define X264_NAL_HRD_VBR 1
#define X264_NAL_HRD_CBR 2

#define X264_PARAM_BAD_NAME (-1)       /* codec.c:1357-1361 */
#define X264_PARAM_BAD_VALUE (-2)

#define X264_ME_DIA 0
#define X264_ME_HEX 1
#define X264_ME_UMH 2
#define X264_ME_ESA 3
#define X264_ME_TESA 4
#define X264_DIRECT_PRED_NONE 0
#define X264_DIRECT_PRED_SPATIAL 1
#define X264_DIRECT_PRED_TEMPORAL 2
#define X264_DIRECT_PRED_AUTO 3
#define X264_B_ADAPT_NONE 0
#define X264_B_ADAPT_FAST 1
#define X264_B_ADAPT_TRELLIS 2
#define X264_B_PYRAMID_NONE 0
#define X264_B_PYRAMID_STRICT 1
#define X264_B_PYRAMID_NORMAL 2
#define X264_WEIGHTP_NONE 0
#define X264_WEIGHTP_SIMPLE 1
#define X264_WEIGHTP_SMART 2
#define X264_AQ_NONE 0
#define X264_AQ_VARIANCE 1
#define X264_AQ_AUTOVARIANCE 2
#define X264_AQ_AUTOVARIANCE_BIASED 3
#define X264_ANALYSE_I4x4 0x0001u
#define X264_ANALYSE_I8x8 0x0002u
#define X264_ANALYSE_PSUB16x16 0x0010u
#define X264_ANALYSE_PSUB8x8 0x0020u
#define X264_ANALYSE_BSUB16x16 0x0100u
#define X264_KEYINT_MAX_INFINITE (1 << 30)

/* name tables the GUI/help text prints (config.c:1560,1642,1649,1710-1731) */
static const char *const x264_direct_pred_names[] = { "none", "spatial", "temporal", "auto", 0 };
static const char *const x264_motion_est_names[] = { "dia", "hex", "umh", "esa", "tesa", 0 };
static const char *const x264_b_pyramid_names[] = { "none", "strict", "normal", 0 };
static const char *const x264_overscan_names[] = { "undef", "show", "crop", 0 };
static const char *const x264_vidformat_names[] = { "component", "pal", "ntsc", "secam", "mac", "undef", 0 };
static const char *const x264_colorprim_names[] = { "", "bt709", "undef", "", "bt470m", "bt470bg", "smpte170m", "smpte240m", "film", "bt2020", "smpte428", "smpte431", "smpte432", 0 };
static const char *const x264_transfer_names[] = { "", "bt709", "undef", "", "bt470m", "bt470bg", "smpte170m", "smpte240m", "linear", "log100", "log316", "iec61966-2-4", "bt1361e", "iec61966-2-1", "bt2020-10", "bt2020-12", "smpte2084", "smpte428", "arib-std-b67", 0 };
static const char *const x264_colmatrix_names[] = { "GBR", "bt709", "undef", "", "fcc", "bt470bg", "smpte170m", "smpte240m", "YCgCo", "bt2020nc", "bt2020c", "smpte2085", 0 };

typedef struct x264_t x264_t;          /* opaque encoder handle (x264vfw.h:187) */

typedef struct x264_level_t {          /* codec.c:1596-1599: level_idc + dpb (in macroblocks), 0-terminated */
    int level_idc;
    int mbps;
    int frame_size;
    int dpb;
    int bitrate;
    int cpb;
    int mv_range;
} x264_level_t;
extern const x264_level_t x264_levels[];

typedef struct x264_param_t {
    unsigned int cpu;                  /* codec.c:1560 ("no asm" -> 0); ignored by the GPU path */
    int i_threads;                     /* G > 1: G closed GOPs of the stream in lock-step (dealt to the visible devices); frames come back (G-1) x keyint calls late, byte-identical */
    int b_sliced_threads;              /* x264 slice threads (--sliced-threads, --tune zerolatency): i_threads slices per picture, each analysed on its own */
    int b_deterministic;
    int i_width, i_height;             /* codec.c:1470-1471 */
    int i_csp;                         /* codec.c:1472 */
    int i_level_idc;                   /* codec.c:1483 (-1 = auto) */
    int i_frame_total;                 /* codec.c:1475 */
    int i_nal_hrd;                     /* codec.c:1126 */
    struct {
        int i_sar_height, i_sar_width; /* codec.c:1551-1552 */
        int i_overscan, i_vidformat, b_fullrange, i_colorprim, i_transfer, i_colmatrix, i_chroma_loc; /* codec.c:1571-1577 */
    } vui;
    int i_frame_reference;             /* codec.c:1602 */
    int i_keyint_max, i_keyint_min;
    int i_scenecut_threshold;
    int b_intra_refresh;
    int i_bframe, i_bframe_adaptive, i_bframe_bias, i_bframe_pyramid;
    int b_open_gop;
    int b_bluray_compat;
    int b_deblocking_filter;
    int i_deblocking_filter_alphac0, i_deblocking_filter_beta;
    int b_cabac;
    int i_cabac_init_idc;
    int b_interlaced;
    int b_constrained_intra;
    void (*pf_log)(void *, int i_level, const char *psz, va_list);   /* codec.c:1555 */
    void *p_log_private;
    int i_log_level;
    struct {
        unsigned int intra, inter;     /* partitions (X264_ANALYSE_*) */
        int b_transform_8x8;
        int i_weighted_pred, b_weighted_bipred;
        int i_direct_mv_pred;
        int i_chroma_qp_offset;
        int i_me_method, i_me_range, i_mv_range, i_mv_range_thread, i_subpel_refine;
        int b_chroma_me, b_mixed_references, i_trellis, b_fast_pskip, b_dct_decimate, i_noise_reduction;
        float f_psy_rd, f_psy_trellis;
        int b_psy;
        int i_luma_deadzone[2];        /* {inter, intra} */
        int b_psnr, b_ssim;            /* codec.c:1558-1559 */
    } analyse;
    struct {
        int i_rc_method;               /* codec.c:1493-1517 */
        int i_qp_constant, i_qp_min, i_qp_max, i_qp_step;
        int i_bitrate;
        float f_rf_constant, f_rf_constant_max, f_rate_tolerance;
        int i_vbv_max_bitrate, i_vbv_buffer_size;
        float f_vbv_buffer_init, f_ip_factor, f_pb_factor;
        int i_aq_mode;
        float f_aq_strength;
        int b_mb_tree, i_lookahead;
        int b_stat_write;              /* codec.c:1488-1541 */
        char *psz_stat_out;
        int b_stat_read;
        char *psz_stat_in;
        float f_qcompress, f_qblur, f_complexity_blur;
        char *psz_zones;               /* --zones <start>,<end>,q=<qp> | b=<bitrate factor> [ / ... ] (x264_param_parse; reaches the driver through its extra command line, codec.c:831-999) */
    } rc;
    int b_aud, b_repeat_headers, b_annexb;      /* codec.c:1611-1615 */
    int i_sps_id;
    int b_vfr_input;                            /* codec.c:1567 */
    uint32_t i_fps_num, i_fps_den, i_timebase_num, i_timebase_den;   /* codec.c:1476-1480,1568-1569 */
    int i_frame_packing;
    int b_stitchable;
    int i_slice_max_size, i_slice_max_mbs, b_fake_interlaced, b_pic_struct;      /* accepted so that the session can say they are not implemented (no effect) */
    int i_slice_count;                          /* --slices N: N slices per picture (at most one per macroblock row), each its own wavefront; filtered across (idc 0) */
} x264_param_t;

typedef struct x264_image_t {          /* codec.c:304-379 fills it over the caller's buffer */
    int i_csp;
    int i_plane;
    int i_stride[4];
    uint8_t *plane[4];
} x264_image_t;

typedef struct x264_picture_t {
    int i_type;                        /* in: X264_TYPE_AUTO; out: actual type (codec.c:1783,1824) */
    int i_qpplus1;
    int b_keyframe;                    /* out (codec.c:1824) */
    int64_t i_pts, i_dts;              /* codec.c:1787; output/ *.c */
    x264_image_t img;
    void *opaque;
} x264_picture_t;

typedef struct x264_nal_t {            /* codec.c:1703,1719: all NALs of a call are contiguous from nal[0].p_payload */
    int i_ref_idc;
    int i_type;
    int b_long_startcode;
    int i_first_mb, i_last_mb;
    int i_payload;
    uint8_t *p_payload;
    int i_padding;
} x264_nal_t;

void x264_param_default(x264_param_t *);                                              /* config.c:1410 */
int  x264_param_default_preset(x264_param_t *, const char *preset, const char *tune); /* codec.c:1463 */
int  x264_param_parse(x264_param_t *, const char *name, const char *value);           /* codec.c:1349 */
void x264_param_apply_fastfirstpass(x264_param_t *);                                  /* codec.c:1581 */
int  x264_param_apply_profile(x264_param_t *, const char *profile);                   /* codec.c:1584 */

int  x264_picture_alloc(x264_picture_t *pic, int i_csp, int i_width, int i_height);   /* codec.c:1673 */
void x264_picture_clean(x264_picture_t *pic);                                         /* codec.c:1872 */
void x264_picture_init(x264_picture_t *pic);

#define x264_encoder_open x264_encoder_open_157                                       /* codec.c:1623 (macro upstream too) */
x264_t *x264_encoder_open(x264_param_t *);
void x264_encoder_parameters(x264_t *, x264_param_t *);                               /* codec.c:1630 */
int  x264_encoder_headers(x264_t *, x264_nal_t **pp_nal, int *pi_nal);                /* codec.c:1650 */
int  x264_encoder_encode(x264_t *, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_in, x264_picture_t *pic_out); /* codec.c:1693 */
int  x264_encoder_delayed_frames(x264_t *);                                           /* codec.c:1848,1854 */
void x264_encoder_close(x264_t *);                                                    /* codec.c:1857 */

#ifdef __cplusplus
}
#endif
#endif
