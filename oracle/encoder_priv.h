/* oracle/encoder_priv.h — shared state of the CPU checker's frame encoder (TEST INFRASTRUCTURE ONLY; see x264o.h).
 * encoder.c owns the frame-level stages (ingest, quantiser maps, deblocking, half-pel planes, DPB rotation); analyse.c owns the
 * per-macroblock work of [x264-upstream] encoder/analyse.c + me.c + macroblock.c behind x264_encoder_encode (codec.c:1693). */
#ifndef X264O_ENCODER_PRIV_H
#define X264O_ENCODER_PRIV_H
#include "x264o.h"
#include "x264gpu.h"

#define PAD 32          /* luma padding of reference planes */
#define CPAD 16         /* chroma padding (samples) */
#define MVCOST_HALF 32768
#define X264O_MAX_REFS 8                         /* list 0 of a P slice: up to 5 pictures + the duplicates of --weightp 2 */
#define X264O_MAX_SLOTS 128                      /* reference pictures of the DPB (cfg.dpb or cfg.refs) + the picture being built */
#define COST_MAX (1 << 28)

typedef struct x264o_encoder x264o_encoder;

struct x264o_encoder {
    x264gpu_config cfg;
    int mbw, mbh, cw, ch;
    int fs;                      /* fenc stride (luma and NV12 chroma) */
    pixel *fenc_y, *fenc_uv;
    int rs;                      /* reference plane stride */
    size_t plane_bytes, cplane_bytes;
    pixel *luma[X264O_MAX_SLOTS];   /* DPB slots: 4 padded planes each (refs + the picture being built) */
    pixel *chroma[X264O_MAX_SLOTS]; /* padded NV12 */
    int slots;                   /* refs + 1 */
    int nref;                    /* reference pictures usable by the current slice in list 0 (= nref_l[0]) */
    /* the slice's reference lists as DPB slots (x264_reference_build_list; x264gpu_pic): list 0 = earlier pictures, nearest first; list 1 (B) = later
     * ones; implicit bi-prediction weights of every (list-0, list-1) pair (h->mb.bipred_weight: the weight of the list-0 sample, of 64) */
    int nref_l[2], lslot[2][X264GPU_MAX_LIST];
    int bipred_weight[X264GPU_MAX_LIST][X264GPU_MAX_LIST];
    int keep;                    /* the picture being coded will be a reference */
    /* --weightp: explicit luma weights of list 0's indices and x264's blind duplicate of reference 0 (h->mb.ref_blind_dupe; -1 = none) */
    struct { int on, denom, scale, offset; } wl0[X264GPU_MAX_LIST];
    struct { int on[2], denom, scale[2], offset[2]; } wc0[X264GPU_MAX_LIST];       /* ... and the chroma weights (Cb, Cr; one denominator) */
    int blind_dupe;
    int row0, row1;              /* macroblock rows [row0, row1) of the slice being coded (x264 slice threads: cfg.slices per picture) */
    int cur;                     /* DPB slot being reconstructed */
    /* per-picture motion side data living with the DPB slot (x264_frame_t): mv16x16 (= h->mb.mvr[0][0], the 16x16 search result
     * in reference 0 of every macroblock), mb_type, the number of references the picture was coded with, its POC */
    int16_t (*mv16[X264O_MAX_SLOTS])[2];
    uint8_t *mbtype[X264O_MAX_SLOTS];
    int slot_nref[X264O_MAX_SLOTS], slot_poc[X264O_MAX_SLOTS], slot_ref0poc[X264O_MAX_SLOTS];   /* ..., POC of the picture's own reference 0 */
    int16_t (*mvr[X264O_MAX_REFS])[2];   /* h->mb.mvr[0][r], r >= 1: 16x16 search results per reference index of the picture being coded */
    int16_t (*mvr1[X264O_MAX_REFS])[2];  /* h->mb.mvr[1][r] (B slices) */
    /* what spatial direct prediction reads from the first picture of list 1 (x264_frame_t ref[] / mv[] of the co-located macroblock): per 8x8
     * block the reference index the block used (list 0's, else list 1's; -1 intra) and that vector */
    int8_t (*colref[X264O_MAX_SLOTS])[4];
    /* ... and what temporal direct prediction reads: the block's own LIST-0 index (-1: none / intra), the POCs behind the picture's list 0 when it was
     * coded (x264_frame_t ref_poc[0]); of the B picture being coded: its mode, whether both modes are probed (--direct auto), the probe counts
     * (h->stat.frame.i_direct_score), map_col_to_list0 and dist_scale_factor[r][0] */
    int8_t (*colref0[X264O_MAX_SLOTS])[4];
    int slot_l0poc[X264O_MAX_SLOTS][X264GPU_MAX_LIST];
    int direct_temporal, direct_auto, direct_score[2], map_col_to_list0[X264GPU_MAX_LIST], dist_scale[X264GPU_MAX_LIST];
    int16_t (*colmv[X264O_MAX_SLOTS])[4][2];
    const int16_t *lowres_mv1;   /* B: lookahead vectors towards the first picture of list 1 (lowres_mvs[1][d]) */
    int poc;                     /* POC of the picture being coded (2 x pictures since the IDR) */
    uint16_t *cost_mv[52];       /* lambda-scaled mv bit costs per qp, centred at MVCOST_HALF */
    x264o_quant_tables qt;
    int have_ref, ring_cur, ring_poc, last_slot;      /* x264o_encoder_encode's sliding window: pictures coded since the IDR, next slot, next POC */
    int slice_type;              /* slice being encoded */
    uint8_t *mbqp;               /* quantiser of every macroblock of the picture being coded (slice quantiser, + AQ offset) */
    float qpm_next;              /* x264o_encoder_set_qpm */
    const float *ext_off;        /* quantiser offsets handed in for the next picture (lookahead: AQ - macroblock-tree), or NULL */
    const int16_t *lowres_mv;    /* optional lookahead vectors of the next picture against its predecessor (x264 fenc->lowres_mvs[0][0]),
                                  * [nmb][2] in lowres quarter-pels, first entry 0x7fff = absent */
    /* state of the macroblock loop (x264: h->stat.frame, h->mb) */
    x264gpu_mb *mbs;
    int16_t *levels;
    int *mb_bits;                /* optional (tests): per macroblock, the CAVLC bit count the RD code predicts for the final macroblock */
    int last_qp;                 /* QP_Y of the previous macroblock in coding order as the entropy coder sees it (h->mb.i_last_qp): mb_qp_delta bits of the RD costs */
    int intra_count;             /* intra macroblocks coded so far in this slice (slice threads) / picture (--slices N): h->stat.frame.i_mb_count[I_*] */
    /* CABAC sessions with RD: the slice's context states as the entropy coding of the finished macroblocks leaves them (h->cabac), the
     * previous macroblock's mb_qp_delta and every 8x8 block's |mvd| (cabac_rd.cpp) */
    uint8_t cabac_state[460];
    int last_dqp;
    int b_trellis;               /* the macroblock's FINAL encode of a trellis session is running (h->mb.b_trellis under --trellis 1) */
    uint8_t *amvd, *amvd1;
};

/* cabac_rd.cpp */
typedef struct x264o_cabac_ctx {
    const x264gpu_mb *mbs;
    const int16_t *levels;
    int mbw, mbh, first_row;
    int pslice, num_ref, t8mode;
    uint8_t *amvd;
    uint8_t *state;
    int last_dqp, last_qp;
    int bslice, num_ref1;        /* B slice: mb_skip / mb_type / sub_mb_type of B, both lists' ref_idx and mvd */
    uint8_t *amvd1;              /* list 1's |mvd| */
} x264o_cabac_ctx;
void x264o_cabac_init_states(uint8_t *state, int pslice, int qp);
long x264o_cabac_mb(x264o_cabac_ctx *c, int mbx, int mby, int size_mode);
const uint16_t *x264o_cabac_entropy(void);

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int median3(int a, int b, int c) { int mn = a < b ? a : b, mx = a < b ? b : a; return c < mn ? mn : c > mx ? mx : c; }
static inline int bs_size_ue(int v) { int n = 0; v++; while (v >> (n + 1)) n++; return 2 * n + 1; }

/* DPB slot of reference index r of list l of the current slice (list 0 of a P slice: r = 0 is the most recent picture) */
static inline int ref_slot_l(const x264o_encoder *e, int l, int r) { return e->lslot[l][r]; }
static inline int ref_slot(const x264o_encoder *e, int r) { return e->lslot[0][r]; }
static inline pixel *luma_plane(const x264o_encoder *e, int slot, int k) { return e->luma[slot] + k * e->plane_bytes + (size_t)PAD * e->rs + PAD; }
static inline pixel *chroma_plane(const x264o_encoder *e, int slot) { return e->chroma[slot] + (size_t)CPAD * e->rs + 2 * CPAD; }

int x264o_lambda(int qp);
int x264o_lambda2(int qp);
const uint16_t *x264o_cost_mv_for(x264o_encoder *e, int qp);
/* analyse.c: analysis + encode of macroblock (mbx,mby) of the slice being coded; fills e->mbs[], e->levels and the reconstruction */
void x264o_macroblock(x264o_encoder *e, int mbx, int mby);

#endif
