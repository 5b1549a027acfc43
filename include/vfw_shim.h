/* include/vfw_shim.h — boundary B2: the slice of the Video-for-Windows driver ABI that x264vfw implements
 * (driverproc.c:89-301, driverproc.def:5-7), re-declared for Linux so the ICM message protocol can be
 * replayed against the MI355X encoder.  Numeric values follow Microsoft's vfw.h / mmsystem.h conventions
 * ([WinSDK], absent from the reference tree and from this image): DRV_USER = ICM_USER = 0x4000,
 * ICM_COMPRESS = ICM_USER+8, ICERR_BADFORMAT = -2, ...  Only what the compress path touches is declared.
 */
#ifndef X264GPU_VFW_SHIM_H
#define X264GPU_VFW_SHIM_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef intptr_t LRESULT;
typedef intptr_t LPARAM;
typedef uintptr_t DWORD_PTR;
typedef void *HDRVR;
typedef uint32_t DWORD;
typedef int32_t LONG;
typedef uint16_t WORD;
typedef uint32_t UINT;

#define mmioFOURCC(a, b, c, d) ((DWORD)(uint8_t)(a) | ((DWORD)(uint8_t)(b) << 8) | ((DWORD)(uint8_t)(c) << 16) | ((DWORD)(uint8_t)(d) << 24))
#define BI_RGB 0

/* driver messages (mmsystem.h) */
#define DRV_LOAD 0x0001
#define DRV_ENABLE 0x0002
#define DRV_OPEN 0x0003
#define DRV_CLOSE 0x0004
#define DRV_DISABLE 0x0005
#define DRV_FREE 0x0006
#define DRV_CONFIGURE 0x0007
#define DRV_QUERYCONFIGURE 0x0008
#define DRV_USER 0x4000
#define DRV_OK 1
#define DRV_CANCEL 0

/* ICM messages (vfw.h) */
#define ICM_USER (DRV_USER + 0x0000)
#define ICM_RESERVED_LOW (DRV_USER + 0x1000)
#define ICM_GETSTATE (ICM_RESERVED_LOW + 0)
#define ICM_SETSTATE (ICM_RESERVED_LOW + 1)
#define ICM_GETINFO (ICM_RESERVED_LOW + 2)
#define ICM_CONFIGURE (ICM_RESERVED_LOW + 10)
#define ICM_ABOUT (ICM_RESERVED_LOW + 11)
#define ICM_GET (ICM_RESERVED_LOW + 17)
#define ICM_SET (ICM_RESERVED_LOW + 18)
#define ICM_COMPRESS_GET_FORMAT (ICM_USER + 4)
#define ICM_COMPRESS_GET_SIZE (ICM_USER + 5)
#define ICM_COMPRESS_QUERY (ICM_USER + 6)
#define ICM_COMPRESS_BEGIN (ICM_USER + 7)
#define ICM_COMPRESS (ICM_USER + 8)
#define ICM_COMPRESS_END (ICM_USER + 9)
#define ICM_DECOMPRESS_GET_FORMAT (ICM_USER + 10)
#define ICM_DECOMPRESS_QUERY (ICM_USER + 11)
#define ICM_DECOMPRESS_BEGIN (ICM_USER + 12)
#define ICM_DECOMPRESS (ICM_USER + 13)
#define ICM_DECOMPRESS_END (ICM_USER + 14)
#define ICM_COMPRESS_FRAMES_INFO (ICM_USER + 70)

#define ICERR_OK 0
#define ICERR_UNSUPPORTED (-1)
#define ICERR_BADFORMAT (-2)
#define ICERR_MEMORY (-3)
#define ICERR_INTERNAL (-4)
#define ICERR_BADFLAGS (-5)
#define ICERR_BADPARAM (-6)
#define ICERR_BADSIZE (-7)
#define ICERR_ERROR (-100)

#define ICTYPE_VIDEO mmioFOURCC('v', 'i', 'd', 'c')
#define VIDCF_COMPRESSFRAMES 0x0008
#define VIDCF_FASTTEMPORALC 0x0020
#define AVIIF_KEYFRAME 0x00000010
#define ICCOMPRESS_KEYFRAME 0x00000001
#define ICVERSION 0x0104

typedef struct BITMAPINFOHEADER {
    DWORD biSize;
    LONG biWidth, biHeight;
    WORD biPlanes, biBitCount;
    DWORD biCompression, biSizeImage;
    LONG biXPelsPerMeter, biYPelsPerMeter;
    DWORD biClrUsed, biClrImportant;
} BITMAPINFOHEADER;
typedef struct BITMAPINFO { BITMAPINFOHEADER bmiHeader; DWORD bmiColors[1]; } BITMAPINFO;

typedef struct ICOPEN {
    DWORD dwSize, fccType, fccHandler, dwVersion, dwFlags;
    LRESULT dwError;
    void *pV1Reserved, *pV2Reserved;
    DWORD dnDevNode;
} ICOPEN;

typedef struct ICINFO {
    DWORD dwSize, fccType, fccHandler, dwFlags, dwVersion, dwVersionICM;
    uint16_t szName[16], szDescription[128], szDriver[128];
} ICINFO;

typedef struct ICCOMPRESS {             /* members used: codec.c:1730-1731,1767,1786,1816-1830 */
    DWORD dwFlags;
    BITMAPINFOHEADER *lpbiOutput;
    void *lpOutput;
    BITMAPINFOHEADER *lpbiInput;
    void *lpInput;
    DWORD *lpckid;
    DWORD *lpdwFlags;
    LONG lFrameNum;
    DWORD dwFrameSize, dwQuality;
    BITMAPINFOHEADER *lpbiPrev;
    void *lpPrev;
} ICCOMPRESS;

typedef struct ICCOMPRESSFRAMES {       /* members used: codec.c:1881-1883 */
    DWORD dwFlags;
    BITMAPINFOHEADER *lpbiOutput;
    LPARAM lOutput;
    BITMAPINFOHEADER *lpbiInput;
    LPARAM lInput;
    LONG lStartFrame, lFrameCount, lQuality, lDataRate, lKeyRate;
    DWORD dwRate, dwScale, dwOverheadPerFrame, dwReserved2;
    void *GetData, *PutData;
} ICCOMPRESSFRAMES;

/* Driver configuration blob exchanged by ICM_GETSTATE / ICM_SETSTATE: the reference's CONFIG (x264vfw.h:121-167, format version 4), field for field and in its
 * order; strings are UTF-8 here (UTF-16 in the Windows build).  Defaults: config.c:96-143. */
#define X264VFW_FORMAT_VERSION 4
#define X264VFW_MAX_PATH 260
typedef struct X264VFW_CONFIG {
    int i_format_version;
    /* Basic */
    int i_preset, i_tuning, i_profile, i_level;        /* indices into the tables of codec.c / config.c:96-104 */
    int i_colorspace;                                  /* 0 = convert to YUV 4:2:0 (CSP_CONVERT_TO_I420), else keep the input's colourspace (codec.c:1472): this build codes
                                                        * 4:2:0 only and says so in the log when asked to keep another one */
    int b_fastdecode, b_zerolatency;
    /* Rate control */
    int i_encoding_type;                               /* 0 lossless, 1 CQP, 2 CRF, 3 ABR, 4 2-pass (codec.c:1490-1533) */
    int i_qp, i_rf_constant, i_passbitrate, i_pass;    /* rf constant x10 (config.c:111) */
    int b_fast1pass, b_createstats, b_updatestats;     /* x264vfw.h:138-140: fast first pass; statistics file in single-pass modes; pass N rewrites it (config.c:114-116: 0, 0, 1) */
    char stats[X264VFW_MAX_PATH];                      /* the statistics file (config.c:140: ".\\x264.stats"); --stats on the extra command line overrides it */
    /* Output */
    int i_output_mode;                                 /* 0 = VFW (the caller's AVI), 1 = file: the stream also goes to output_file through the muxers (codec.c:1545) */
    int i_fourcc;
    int b_vd_hack;                                     /* X264VFW_USE_VIRTUALDUB_HACK (codec.c:1410) */
    char output_file[X264VFW_MAX_PATH];
    /* Sample aspect ratio */
    int i_sar_width, i_sar_height;
    /* Debug */
    int i_log_level;
    int b_psnr, b_ssim, b_no_asm;
    /* Decoder / AVI muxer (x264vfw.h:158-163; out of scope here: kept so that the blob has the reference's fields) */
    int b_disable_decoder;
    /* Extra command line */
    char extra_cmdline[4096];
} X264VFW_CONFIG;

LRESULT DriverProc(DWORD_PTR dwDriverId, HDRVR hDriver, UINT uMsg, LPARAM lParam1, LPARAM lParam2);   /* driverproc.c:89 */

#ifdef __cplusplus
}
#endif
#endif
