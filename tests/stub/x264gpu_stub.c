/* tests/stub/x264gpu_stub.c — a CPU stand-in for the B3 device library (include/x264gpu.h), TEST INFRASTRUCTURE ONLY.
 *
 * Lets the CPU test suite drive the host shell (x264_* API, GOP-parallel scheduling, the dealing of GOP slots to several devices)
 * without a GPU: "device memory" is malloc'ed, the frame pipeline is the CPU checker (oracle/, liboracle.so), the device count comes from
 * X264GPU_STUB_DEVICES.  It is built into tests/stub/_build/ under the device library's soname together with a copy of the host
 * library linked against it (tests/stub/Makefile); nothing under x264vfw_amd/ links or loads it, and bench.py never does. */
#include "x264gpu.h"
#include "x264o.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
/* The checker behind this stand-in (oracle/) keeps scratch in file-scope variables: it is test infrastructure, written for one caller.  The host library calls the
 * device ABI from several threads (one per device under --threads G, one per session in a batch): every entry point that computes takes this lock, so the threads'
 * ORDER is still the host's business and the checker sees one call at a time. */
static pthread_mutex_t g_oracle_mu = PTHREAD_MUTEX_INITIALIZER;
static void oracle_unlock(int *held) { if (*held) pthread_mutex_unlock(&g_oracle_mu); }
#define ORACLE_LOCK() int oracle_held_ __attribute__((cleanup(oracle_unlock))) = (pthread_mutex_lock(&g_oracle_mu), 1)

typedef struct x264o_encoder x264o_encoder;
x264o_encoder *x264o_encoder_create(const x264gpu_config *cfg);
void x264o_encoder_destroy(x264o_encoder *e);
int x264o_encoder_mb_count(const x264o_encoder *e);
void x264o_encoder_set_qp(x264o_encoder *e, int qp_i, int qp_p);
void x264o_encoder_set_qpm(x264o_encoder *e, float qpm);
void x264o_encoder_set_mb_qp_offsets(x264o_encoder *e, const float *off);
int x264o_encoder_encode(x264o_encoder *e, const uint8_t *i420, int slice_type, x264gpu_mb *mbs, int16_t *levels);
void x264o_encoder_get_recon(x264o_encoder *e, uint8_t *out);

static __thread int t_dev = 0;
static __thread char t_err[256] = "";
static long g_calls[16];                       /* x264gpu_encode_frames calls per device */

static int ndev(void) { const char *e = getenv("X264GPU_STUB_DEVICES"); int n = e ? atoi(e) : 2; return n < 1 ? 1 : n > 16 ? 16 : n; }
static int fail(const char *what) { snprintf(t_err, sizeof(t_err), "stub: %s", what); return X264GPU_EINVAL; }

int x264gpu_abi_version(void) { return X264GPU_ABI_VERSION; }
int x264gpu_device_count(void) { return ndev(); }
int x264gpu_set_device(int dev) { if (dev < 0 || dev >= ndev()) return fail("no such device"); t_dev = dev; return X264GPU_OK; }
int x264gpu_get_device(int *dev) { *dev = t_dev; return X264GPU_OK; }
const char *x264gpu_last_error(void) { return t_err; }
int x264gpu_malloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? X264GPU_OK : X264GPU_ENOMEM; }
int x264gpu_free(void *p) { free(p); return X264GPU_OK; }
int x264gpu_memcpy_h2d(void *d, const void *s, size_t n, void *st) { memcpy(d, s, n); return X264GPU_OK; }
int x264gpu_memcpy_d2h(void *d, const void *s, size_t n, void *st) { memcpy(d, s, n); return X264GPU_OK; }
int x264gpu_memcpy_d2d(void *d, const void *s, size_t n, void *st) { memmove(d, s, n); return X264GPU_OK; }
int x264gpu_memset(void *d, int v, size_t n, void *st) { memset(d, v, n); return X264GPU_OK; }
int x264gpu_stream_sync(void *st) { return X264GPU_OK; }
int x264gpu_stream_create(void **st) { static int dummy; *st = &dummy; return X264GPU_OK; }
int x264gpu_stream_destroy(void *st) { return st ? X264GPU_OK : fail("stream_destroy: null"); }
/* (the stand-in runs everything at call time: an event is always reached) */
int x264gpu_event_create(void **ev) { static int dummy; *ev = &dummy; return X264GPU_OK; }
int x264gpu_event_destroy(void *ev) { return ev ? X264GPU_OK : fail("event_destroy: null"); }
int x264gpu_event_record(void *ev, void *st) { return ev ? X264GPU_OK : fail("event_record: null"); }
int x264gpu_event_sync(void *ev) { return ev ? X264GPU_OK : fail("event_sync: null"); }
static long g_views;
long x264gpu_stub_views(void) { return g_views; }          /* launch contexts created over a parent's DPB */
long x264gpu_stub_encode_calls(int dev) { return dev >= 0 && dev < 16 ? g_calls[dev] : -1; }

struct x264gpu_encoder { x264gpu_config cfg; x264o_encoder **e; int dev, nmb; const float *off; int8_t *sqp; float *sqpm; float qpm; struct x264gpu_encoder *view_of; };

int x264gpu_encoder_create(x264gpu_encoder **out, const x264gpu_config *cfg)
{ ORACLE_LOCK();
    if (!out || !cfg || cfg->streams < 1) return fail("encoder_create arguments");
    /* the device's structural checks (csrc/encoder.hip): RD at subme >= 8, deblock-aware RD and B pictures at subme >= 9 run in the refinement instantiations only */
    if (cfg->rd && cfg->subme >= 8 && !(cfg->cabac && (cfg->me_method == 1 || cfg->me_method == 2))) return fail("RD at subme >= 8 needs CABAC and me hex / umh");
    if ((cfg->rd & 64) && !(cfg->cabac && (cfg->me_method == 1 || cfg->me_method == 2))) return fail("deblock-aware RD needs CABAC and me hex / umh");
    if (cfg->dpb > 0 && cfg->subme >= 9 && !(cfg->rd && cfg->cabac && (cfg->me_method == 1 || cfg->me_method == 2))) return fail("B pictures at subme 9 need RD, CABAC and me hex / umh");
    x264gpu_encoder *g = calloc(1, sizeof(*g));
    g->cfg = *cfg; g->dev = t_dev;
    g->e = calloc((size_t)cfg->streams, sizeof(*g->e));
    x264gpu_config one = *cfg; one.streams = 1;
    for (int s = 0; s < cfg->streams; s++) g->e[s] = x264o_encoder_create(&one);
    g->nmb = x264o_encoder_mb_count(g->e[0]);
    *out = g;
    return X264GPU_OK;
}
/* a view (a second launch context over the parent's DPB): the stand-in runs every call at call time, so the parent's checker encoders serve it — only the per-launch
 * settings (quantiser offsets, per-stream quantisers) are its own */
int x264gpu_encoder_create_view(x264gpu_encoder **out, x264gpu_encoder *parent)
{
    if (!out || !parent || parent->view_of || parent->cfg.dpb <= 0) return fail("encoder_create_view arguments");
    x264gpu_encoder *g = calloc(1, sizeof(*g));
    g->cfg = parent->cfg; g->dev = parent->dev; g->e = parent->e; g->nmb = parent->nmb; g->view_of = parent;
    g_views++;
    *out = g;
    return X264GPU_OK;
}
/* the levels packed in place (include/x264gpu.h): plain C, macroblock by macroblock */
int x264gpu_pack_levels(int16_t *lv, int streams, int nmb, x264gpu_level_index *index, uint32_t *kept, void *st)
{
    if (!lv || !index || streams < 1 || nmb < 1) return fail("pack_levels arguments");
    for (int s = 0; s < streams; s++) {
        int16_t *base = lv + (size_t)s * nmb * X264GPU_MB_LEVELS;
        uint32_t at = 0;
        for (int i = 0; i < nmb; i++) {
            int16_t mb[X264GPU_MB_LEVELS];
            memcpy(mb, base + (size_t)i * X264GPU_MB_LEVELS, sizeof(mb));
            uint32_t groups = 0;
            for (int g = 0; g < X264GPU_MB_LEVELS / 16; g++) for (int k = 0; k < 16; k++) if (mb[g * 16 + k]) { groups |= 1u << g; break; }
            index[(size_t)s * nmb + i].at = at; index[(size_t)s * nmb + i].groups = groups;
            for (int g = 0; g < X264GPU_MB_LEVELS / 16; g++) if (groups >> g & 1) { memcpy(base + (size_t)at * 16, mb + g * 16, 32); at++; }
        }
        if (kept) kept[s] = at;
    }
    return X264GPU_OK;
}
int x264gpu_stream_wait_event(void *st, void *ev) { return ev ? X264GPU_OK : fail("stream_wait_event: null"); }
void x264gpu_encoder_destroy(x264gpu_encoder *g)
{
    if (!g) return;
    if (!g->view_of) { for (int s = 0; s < g->cfg.streams; s++) x264o_encoder_destroy(g->e[s]); free(g->e); }
    free(g->sqp); free(g->sqpm); free(g);
}
int x264gpu_encoder_mb_count(const x264gpu_encoder *g) { return g ? g->nmb : 0; }
int x264gpu_encoder_set_qp(x264gpu_encoder *g, int qp_i, int qp_p) { g->cfg.qp_i = qp_i; g->cfg.qp_p = qp_p; return X264GPU_OK; }
int x264gpu_encoder_set_mb_qp_offsets(x264gpu_encoder *g, const float *off) { g->off = off; return X264GPU_OK; }
int x264gpu_encoder_set_stream_qpms(x264gpu_encoder *g, const int8_t *qps, const float *qpms)
{
    free(g->sqp); g->sqp = NULL; free(g->sqpm); g->sqpm = NULL;
    if (qps) { g->sqp = malloc((size_t)g->cfg.streams); memcpy(g->sqp, qps, (size_t)g->cfg.streams); }
    if (qps && qpms) {
        g->sqpm = malloc((size_t)g->cfg.streams * sizeof(float)); memcpy(g->sqpm, qpms, (size_t)g->cfg.streams * sizeof(float));
        for (int s = 0; s < g->cfg.streams; s++) if (qpms[s] != 0.f && !(qpms[s] > (float)qps[s] - 1.f && qpms[s] < (float)qps[s] + 1.f)) return fail("qpm is not near qp");
    }
    return X264GPU_OK;
}
int x264gpu_encoder_set_stream_qps(x264gpu_encoder *g, const int8_t *qps) { return x264gpu_encoder_set_stream_qpms(g, qps, NULL); }
int x264gpu_encoder_set_qpm(x264gpu_encoder *g, float qpm) { if (!(qpm >= 0.f && qpm < 52.f)) return fail("qpm out of range"); g->qpm = qpm; return X264GPU_OK; }
int x264gpu_trellis_blocks(const int16_t *c, int n, int cat, int qp, int intra, const uint8_t *s, int16_t *l, uint8_t *z, void *st) { return fail("trellis primitive: not in the stub"); }
int x264gpu_encoder_cabac_states(x264gpu_encoder *g, int stream, int slice, uint8_t *out) { return fail("context states: not in the stub"); }
void x264o_encoder_set_lowres_mvs(x264o_encoder *e, const int16_t *mv);
void x264o_encoder_set_lowres_mvs1(x264o_encoder *e, const int16_t *mv);
int x264gpu_encoder_set_lowres_mvs(x264gpu_encoder *g, const int16_t *mv) { if (g->cfg.streams != 1 && mv) return fail("lowres vectors: one stream in the stub"); x264o_encoder_set_lowres_mvs(g->e[0], mv); return X264GPU_OK; }
int x264gpu_encoder_set_lowres_mvs1(x264gpu_encoder *g, const int16_t *mv) { if (g->cfg.streams != 1 && mv) return fail("lowres vectors: one stream in the stub"); x264o_encoder_set_lowres_mvs1(g->e[0], mv); return X264GPU_OK; }

/* ---- the lookahead's frame costs in x264's structure: oracle/slicetype.c behind the x264gpu_slicetype_* ABI ("device" pointers are host pointers) ---- */
typedef struct x264o_slicetype x264o_slicetype;
x264o_slicetype *x264o_slicetype_create(int width, int height, int slots, int bframes, int me_method, int subme, int me_range, int weightb, int mv_range, int do_edges);
void x264o_slicetype_destroy(x264o_slicetype *st);
int x264o_slicetype_put_frame(x264o_slicetype *st, int slot, const uint8_t *i420);
int x264o_slicetype_frame_cost(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1);
int x264o_slicetype_intra_mbs(const x264o_slicetype *st, int slot, int d0);
int x264o_slicetype_cost_est(const x264o_slicetype *st, int slot, int d0, int d1);
const int16_t *x264o_slicetype_mvs(const x264o_slicetype *st, int slot, int list, int dist);
const int *x264o_slicetype_mv_costs(const x264o_slicetype *st, int slot, int list, int dist);
const int *x264o_slicetype_intra_costs(const x264o_slicetype *st, int slot);
const uint16_t *x264o_slicetype_lowres_costs(const x264o_slicetype *st, int slot, int d0, int d1);
struct x264gpu_slicetype { x264o_slicetype *st; };
int x264gpu_slicetype_create(x264gpu_slicetype **out, int w, int h, int streams, int slots, int bframes, int me_method, int subme, int me_range, int weightb, int mv_range, int do_edges)
{ ORACLE_LOCK();
    if (streams != 1) return fail("stub slicetype: one stream");
    if (slots > 128) return fail("stub slicetype: at most 128 slots");
    x264gpu_slicetype *s = calloc(1, sizeof(*s));
    s->st = x264o_slicetype_create(w, h, slots, bframes, me_method, subme, me_range, weightb, mv_range, do_edges);
    *out = s;
    return X264GPU_OK;
}
void x264gpu_slicetype_destroy(x264gpu_slicetype *s) { if (s) { x264o_slicetype_destroy(s->st); free(s); } }
int x264gpu_slicetype_put_frame(x264gpu_slicetype *s, int slot, const uint8_t *i420, void *stream) { ORACLE_LOCK(); return x264o_slicetype_put_frame(s->st, slot, i420) ? fail("slicetype slot") : X264GPU_OK; }
int x264gpu_slicetype_frame_cost(x264gpu_slicetype *s, int s0, int s1, int sb, int d0, int d1, int32_t *h_score, void *stream)
{ ORACLE_LOCK();
    const int c = x264o_slicetype_frame_cost(s->st, s0, s1, sb, d0, d1);
    if (c < 0) return fail("slicetype triple");
    h_score[0] = c;
    return X264GPU_OK;
}
int x264gpu_slicetype_intra_mbs(x264gpu_slicetype *s, int slot, int d0, int idx) { return x264o_slicetype_intra_mbs(s->st, slot, d0); }
int x264gpu_slicetype_cost_est(x264gpu_slicetype *s, int slot, int d0, int d1, int idx) { return x264o_slicetype_cost_est(s->st, slot, d0, d1); }
const int16_t *x264gpu_slicetype_lowres_mvs(x264gpu_slicetype *s, int slot, int list, int dist)
{
    const int16_t *m = x264o_slicetype_mvs(s->st, slot, list, dist);
    return m[0] == 0x7fff ? NULL : m;
}
const int *x264gpu_slicetype_lowres_mv_costs(x264gpu_slicetype *s, int slot, int list, int dist) { return x264o_slicetype_mv_costs(s->st, slot, list, dist); }
const int *x264gpu_slicetype_intra_costs(x264gpu_slicetype *s, int slot) { return x264o_slicetype_intra_costs(s->st, slot); }
const uint16_t *x264gpu_slicetype_lowres_costs(x264gpu_slicetype *s, int slot, int d0, int d1) { return x264o_slicetype_lowres_costs(s->st, slot, d0, d1); }
int x264o_slicetype_frame_cost_w(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1, int on, int scale, int denom, int offset);
void x264o_slicetype_pixel_stats(x264o_slicetype *st, int slot, const uint8_t *i420, uint64_t out[2]);
long x264o_slicetype_weight_cost(x264o_slicetype *st, int sf, int sr, int dist, int on, int scale, int denom, int offset);
void x264o_slicetype_chroma_stats(x264o_slicetype *st, int slot, const uint8_t *i420, uint64_t out[4]);
long x264o_slicetype_weight_cost_chroma(x264o_slicetype *st, int sf, const uint8_t *i420_fenc, const uint8_t *i420_ref, int dist, int plane, int on, int scale, int denom, int offset);
int x264gpu_slicetype_frame_cost_w(x264gpu_slicetype *s, int s0, int s1, int sb, int d0, int d1, int on, int scale, int denom, int offset, int32_t *h_score, void *stream)
{ ORACLE_LOCK();
    const int c = x264o_slicetype_frame_cost_w(s->st, s0, s1, sb, d0, d1, on, scale, denom, offset);
    if (c < 0) return fail("slicetype triple");
    h_score[0] = c;
    return X264GPU_OK;
}
int x264gpu_slicetype_pixel_stats(x264gpu_slicetype *s, int slot, const uint8_t *i420, uint64_t *out, void *stream) { ORACLE_LOCK(); x264o_slicetype_pixel_stats(s->st, slot, i420, out); return X264GPU_OK; }
int x264gpu_slicetype_chroma_stats(x264gpu_slicetype *s, int slot, const uint8_t *i420, uint64_t *out, void *stream) { ORACLE_LOCK(); (void)stream; x264o_slicetype_chroma_stats(s->st, slot, i420, out); return X264GPU_OK; }
int x264gpu_slicetype_weight_cost_chroma(x264gpu_slicetype *s, int sf, const uint8_t *i420_fenc, const uint8_t *i420_ref, int dist, int plane, int on, int scale, int denom, int offset, int64_t *h_cost, void *stream)
{ ORACLE_LOCK();
    (void)stream;
    if (plane < 1 || plane > 2) return fail("weight_cost_chroma: plane 1 or 2");
    const long c = x264o_slicetype_weight_cost_chroma(s->st, sf, i420_fenc, i420_ref, dist, plane, on, scale, denom, offset);
    if (c < 0) return fail("weight_cost_chroma: the picture has no costs yet");
    *h_cost = c;
    return X264GPU_OK;
}
int x264gpu_slicetype_weight_cost(x264gpu_slicetype *s, int sf, int sr, int dist, int on, int scale, int denom, int offset, int64_t *h_cost, void *stream)
{ ORACLE_LOCK();
    const long c = x264o_slicetype_weight_cost(s->st, sf, sr, dist, on, scale, denom, offset);
    if (c < 0) return fail("weight cost: no intra costs");
    h_cost[0] = c;
    return X264GPU_OK;
}
void x264o_slicetype_set_aq(x264o_slicetype *st, int slot, const float *aq);
void x264o_slicetype_clear_propagate(x264o_slicetype *st, int slot);
int x264o_slicetype_propagate(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1, int referenced);
int x264o_slicetype_finish(x264o_slicetype *st, int slot, float strength, float weightdelta, float *out);
const int32_t *x264o_slicetype_propagate_cost(x264o_slicetype *st, int slot);
void x264o_slicetype_set_bframe_bias(x264o_slicetype *st, int bias);
int x264gpu_slicetype_set_bframe_bias(x264gpu_slicetype *s, int bias) { x264o_slicetype_set_bframe_bias(s->st, bias); return X264GPU_OK; }
int x264gpu_slicetype_set_row_mode(x264gpu_slicetype *s, int serial) { (void)s; (void)serial; return X264GPU_OK; }          /* (a launch geometry: nothing to model) */
int x264o_slicetype_cost_aq(x264o_slicetype *st, int slot, int d0, int d1);
int x264gpu_slicetype_cost_aq(x264gpu_slicetype *s, int slot, int d0, int d1, int32_t *h_score, void *stream) { ORACLE_LOCK(); const int c = x264o_slicetype_cost_aq(s->st, slot, d0, d1); if (c < 0) return fail("slicetype cost_aq"); h_score[0] = c; return X264GPU_OK; }
int x264gpu_slicetype_set_aq(x264gpu_slicetype *s, int slot, const float *aq, void *stream) { x264o_slicetype_set_aq(s->st, slot, aq); return X264GPU_OK; }
int x264gpu_slicetype_clear_propagate(x264gpu_slicetype *s, int slot, void *stream) { x264o_slicetype_clear_propagate(s->st, slot); return X264GPU_OK; }
int x264gpu_slicetype_propagate(x264gpu_slicetype *s, int s0, int s1, int sb, int d0, int d1, int referenced, void *stream) { ORACLE_LOCK(); return x264o_slicetype_propagate(s->st, s0, s1, sb, d0, d1, referenced) ? fail("macroblock-tree: costs of the triple missing") : X264GPU_OK; }
int x264gpu_slicetype_finish(x264gpu_slicetype *s, int slot, float strength, float weightdelta, float *out, void *stream) { ORACLE_LOCK(); return x264o_slicetype_finish(s->st, slot, strength, weightdelta, out) ? fail("macroblock-tree: no intra costs") : X264GPU_OK; }
const int32_t *x264gpu_slicetype_propagate_cost(x264gpu_slicetype *s, int slot) { return x264o_slicetype_propagate_cost(s->st, slot); }
int x264gpu_encode_frames(x264gpu_encoder *g, const uint8_t *i420, int slice_type, x264gpu_mb *mb, int16_t *lv, void *st)
{ ORACLE_LOCK();
    if (g->dev != t_dev) return fail("encoder used from a thread bound to another device");
    const size_t fsz = (size_t)g->cfg.width * g->cfg.height * 3 / 2;
    g_calls[g->dev]++;
    for (int s = 0; s < g->cfg.streams; s++) {
        const int qi = g->sqp ? g->sqp[s] : g->cfg.qp_i, qp = g->sqp ? g->sqp[s] : g->cfg.qp_p;
        x264o_encoder_set_qp(g->e[s], qi, qp);
        x264o_encoder_set_qpm(g->e[s], g->sqp ? (g->sqpm ? g->sqpm[s] : 0.f) : g->qpm);
        x264o_encoder_set_mb_qp_offsets(g->e[s], g->off ? g->off + (size_t)s * g->nmb : NULL);
        if (x264o_encoder_encode(g->e[s], i420 + s * fsz, slice_type, mb + (size_t)s * g->nmb, lv + (size_t)s * g->nmb * X264GPU_MB_LEVELS)) return fail("P picture without a reference");
    }
    return X264GPU_OK;
}
int x264o_encoder_encode_pic(x264o_encoder *e, const uint8_t *i420, const x264gpu_pic *pic, x264gpu_mb *mbs, int16_t *levels);
int x264gpu_encode_pictures(x264gpu_encoder *g, const uint8_t *i420, const x264gpu_pic *pics, x264gpu_mb *mb, int16_t *lv, void *st)
{ ORACLE_LOCK();
    if (g->dev != t_dev) return fail("encoder used from a thread bound to another device");
    const size_t fsz = (size_t)g->cfg.width * g->cfg.height * 3 / 2;
    g_calls[g->dev]++;
    /* the device's structural checks (csrc/encoder.hip x264gpu_encode_pictures): lock-step streams share everything but the quantiser and its fraction */
    for (int s = 0; s < g->cfg.streams; s++) {
        if (pics[s].qpm != 0.f && !(pics[s].qpm > (float)pics[s].qp - 1.f && pics[s].qpm < (float)pics[s].qp + 1.f)) return fail("qpm is not near qp");
        if (s && !(pics[s].slice_type == pics[0].slice_type && pics[s].poc == pics[0].poc && pics[s].dst == pics[0].dst && pics[s].keep == pics[0].keep &&
                   pics[s].nref[0] == pics[0].nref[0] && pics[s].nref[1] == pics[0].nref[1] && !memcmp(pics[s].slot, pics[0].slot, sizeof(pics[0].slot)) &&
                   pics[s].blind_dupe == pics[0].blind_dupe && !memcmp(pics[s].wl0, pics[0].wl0, sizeof(pics[0].wl0)) && !memcmp(pics[s].wc0, pics[0].wc0, sizeof(pics[0].wc0)) &&
                   (pics[s].direct_auto != 0) == (pics[0].direct_auto != 0))) return fail("lock-step streams must share the picture structure");
    }
    for (int s = 0; s < g->cfg.streams; s++) {
        x264o_encoder_set_mb_qp_offsets(g->e[s], g->off ? g->off + (size_t)s * g->nmb : NULL);
        if (x264o_encoder_encode_pic(g->e[s], i420 + s * fsz, &pics[s], mb + (size_t)s * g->nmb, lv + (size_t)s * g->nmb * X264GPU_MB_LEVELS)) return fail("picture control rejected");
    }
    return X264GPU_OK;
}
void x264o_encoder_direct_scores(const x264o_encoder *e, int out[2]);
int x264gpu_encoder_direct_scores(x264gpu_encoder *g, int *h_scores) { for (int s = 0; s < g->cfg.streams; s++) x264o_encoder_direct_scores(g->e[s], h_scores + 2 * s); return X264GPU_OK; }
int x264gpu_encoder_get_recon(x264gpu_encoder *g, int s, uint8_t *out, void *st) { x264o_encoder_get_recon(g->e[s], out); return X264GPU_OK; }
int x264gpu_encoder_get_recon_slot(x264gpu_encoder *g, int s, int slot, uint8_t *out, void *st) { return fail("recon by slot: not in the stub"); }

struct x264gpu_lookahead { x264o_lookahead *la; int w, h, nb; };
int x264gpu_lookahead_create(x264gpu_lookahead **out, int w, int h, int streams, int me_range, int subme)
{ ORACLE_LOCK();
    if (streams != 1) return fail("stub lookahead: one stream");
    x264gpu_lookahead *l = calloc(1, sizeof(*l));
    l->la = x264o_lookahead_create(w, h, me_range, subme); l->w = w; l->h = h; l->nb = ((w + 15) / 16) * ((h + 15) / 16);
    *out = l;
    return X264GPU_OK;
}
void x264gpu_lookahead_destroy(x264gpu_lookahead *l) { if (l) { x264o_lookahead_destroy(l->la); free(l); } }
int x264gpu_lookahead_frame_cost(x264gpu_lookahead *l, const uint8_t *i420, int reset, int32_t *out, int32_t *blocks, void *st)
{ ORACLE_LOCK();
    int32_t *tmp = blocks ? blocks : malloc((size_t)l->nb * 4 * sizeof(int32_t));
    const int rc = x264o_lookahead_frame_cost(l->la, i420, reset, out, tmp);
    if (!blocks) free(tmp);
    return rc ? fail("lookahead") : X264GPU_OK;
}
int x264gpu_lookahead_aq_offsets(x264gpu_lookahead *l, const uint8_t *i420, float strength, float *out, void *st)
{ ORACLE_LOCK();
    x264o_aq_offsets(i420, l->w, l->h, strength, out);
    return X264GPU_OK;
}

int x264gpu_lookahead_aq_offsets_mode(x264gpu_lookahead *l, const uint8_t *i420, int mode, float strength, float *out, void *st)
{ ORACLE_LOCK();
    (void)st;
    if (mode < 1 || mode > 3) return fail("aq mode 1..3");
    x264o_aq_offsets_mode(i420, l->w, l->h, mode, strength, out);
    return X264GPU_OK;
}
int x264gpu_lookahead_mbtree(x264gpu_lookahead *l, const int32_t *const *info, const float *const *aq, int n, float strength, float *out, void *st)
{ ORACLE_LOCK();
    x264o_mbtree((l->w + 15) / 16, (l->h + 15) / 16, info, aq, n, strength, out);
    return X264GPU_OK;
}
long x264gpu_csp_img_fill(int csp, int width, int height, long off[3], int stride[3]) { return x264o_csp_img_fill(csp, width, height, off, stride); }
int x264gpu_csp_to_i420(const uint8_t *const src[3], const int ss[3], int csp, int w, int h, int m709, int full, uint8_t *const dst[3], const int ds[3], void *st)
{ ORACLE_LOCK();
    return x264o_csp_to_i420(dst, ds, src, ss, csp, w, h, m709, full) ? fail("csp") : X264GPU_OK;
}
