// encoder.hip — Tier-2 frame pipeline of include/x264gpu.h: host-side orchestration of the HIP
// kernels that replace the per-frame work of x264_encoder_encode() (reference call site
// codec.c:1693) for a lock-step batch of independent closed-GOP streams.
//
// Stage order per call (all on the caller's HIP stream, no host synchronisation, graph-capturable):
//   ingest -> per-macroblock quantisers -> macroblock loop (k_mb.hip.h: analysis + encode of every macroblock in raster order,
//   one wavefront per stream) -> QP_Y inheritance -> deblock wavefront -> half-pel planes + border expansion.
//   Device-resident DPB: refs + 1 slots of {4 padded luma planes, padded NV12 chroma, 16x16 vectors, macroblock types} per stream.
#include "enc_common.hip.h"
#include "k_encode.hip.h"
#include <stdlib.h>
#include "k_deblock.hip.h"
#include "cabac_layout.hip.h"
#include <math.h>
#include <vector>
#include <string.h>
#include <new>

namespace x264gpu {
void launch_mb_slice_dia(const EncK &k, int streams, bool big_margin, hipStream_t st);
void launch_mb_slice_hex(const EncK &k, int streams, bool big_margin, hipStream_t st);
void launch_mb_slice_umh(const EncK &k, int streams, bool big_margin, hipStream_t st);
void launch_mb_slice_esa(const EncK &k, int streams, bool big_margin, hipStream_t st);
void launch_mb_slice_intra(const EncK &k, int streams, hipStream_t st);
void launch_mb_slice_b_hex(const EncK &k, int streams, hipStream_t st);       // B slices (mb_slice_b*.hip): RD sessions with CABAC
void launch_mb_slice_b_dia(const EncK &k, int streams, hipStream_t st);
void launch_mb_slice_b_umh(const EncK &k, int streams, hipStream_t st);
void launch_mb_slice_b_esa(const EncK &k, int streams, hipStream_t st);
int trellis_table_ptrs(const uint16_t **su, const uint8_t **tu, const int **l2);        // prim_kernels.hip
int cabac_chain_table(const uint32_t **out);                                              // prim_kernels.hip
int launch_hpel_filter(uint8_t *planes, size_t plane_bytes, int stride, int w, int h, int pad, int batch,
                       size_t batch_bytes, hipStream_t st);
}
using namespace x264gpu;

struct x264gpu_encoder {
    x264gpu_config cfg;
    EncK k;                       // template of the kernel argument block (pointers refreshed per call)
    uint8_t *fenc_y = nullptr, *fenc_uv = nullptr;
    uint8_t *luma[8] = {}, *chroma[8] = {};      // picture slots: the DPB (cfg.dpb or cfg.refs pictures) + the picture being reconstructed
    int slots = 2, have = 0;                     // have = pictures in the DPB since the last IDR (x264gpu_encode_frames' sliding window)
    int16_t *mv16[8] = {};                       // per slot: 16x16 search results in reference 0 of list 0 (x264 frame->mv16x16 = h->mb.mvr[0][0])
    uint8_t *mbtype[8] = {};                     // per slot: macroblock types (x264 frame->mb_type)
    int8_t *colref[8] = {}; int16_t *colmv[8] = {};      // per slot (sessions with B pictures): what spatial direct prediction reads of a co-located picture
    int8_t *colref0[8] = {};                             // ... and temporal direct prediction: the blocks' own list-0 indices; the POCs behind each slot's list 0
    // what the host knows about the pictures in the slots (set when a picture is issued): one block, so that the launch contexts of a session (views, below) share it
    struct DpbMeta { int slot_l0poc[8][8] = {}; int slot_nref[8] = {}, slot_poc[8] = {}, slot_ref0poc[8] = {}; } meta_own, *meta = &meta_own;
    // a VIEW (x264gpu_encoder_create_view): a second launch context over the parent's DPB — picture slots and their side data are the parent's (not freed here), the
    // per-launch scratch (source planes, per-reference vector arrays, |mvd| / total_coeff, quantiser tables) its own: pictures of one session that share only FINISHED
    // references can then be in flight together on different streams
    x264gpu_encoder *view_of = nullptr;
    hipStream_t last_stream = nullptr;               // the stream of the last encode_core call (what a read-back of its results waits for)
    uint8_t *direct_flags = nullptr, *direct_flags_base = nullptr; int direct_flags_sel = 0; int *dscore = nullptr; bool use_direct_flags = false;      // --direct temporal / auto: per-stream modes of the B picture, the probe counts
    int16_t *mvr[8] = {};                        // per combined reference index >= 1: 16x16 search results of the picture being coded
    int poc = 0, ring = 2;                       // the sliding window: next POC, slots in rotation (refs + 1)
    const int16_t *lowres_mv = nullptr, *lowres_mv1 = nullptr;
    int cur = 0, last = 0;                       // next slot of the sliding window; slot of the picture coded last
    unsigned long long *dbg = nullptr;   // diagnostics buffer set by x264gpu_encoder_set_debug
    uint8_t *tc = nullptr;               // RD: total_coeff of every block of the picture being coded
    uint8_t *amvd = nullptr;             // CABAC RD: |mvd| of every 8x8 block of the picture being coded
    uint32_t *cab_out = nullptr;         // ... and the context variables every slice's wavefront ended with: [stream][slice][3][64] (tests; layout: cabac_rd.hip.h cab_locate)
    int *perm = nullptr; unsigned *wtime = nullptr; bool wt_valid[4] = { false, false, false, false };      // load balance: EncK.perm / wtime, one history per picture kind (I, P, B reference, B)
    int *sl_stat = nullptr, *sl_rerun = nullptr;     // --slices N: per (stream, slice) intra statistics of the speculative slice passes (EncK.sl_stat)
    unsigned long long *prof = nullptr;  // MB_PROF builds: phase counters of the last macroblock-loop launch
    int *wf_progress = nullptr;          // [streams][2][WFG_ROWS] row counters of the multi-workgroup wavefront kernels
    // adaptive quantisation: per-macroblock quantisers and the per-quantiser tables (built when aq_mode != 0)
    uint8_t *mbqp = nullptr;
    const float *ext_off = nullptr;      // quantiser offsets handed in by the caller (lookahead), [streams][nmb] single floats
    int8_t *stream_qp = nullptr; float *stream_qpm = nullptr;         // device copies of the per-stream slice quantisers (x264gpu_encoder_set_stream_qps): two, used in turn,
    int stream_qp_sel = 0;               // so that an encode still in flight on the caller's stream keeps reading the set it was issued with
    bool use_stream_qp = false;
    float qpm_next = 0.f;                // x264gpu_encoder_set_qpm: the float quantiser of x264gpu_encode_frames' pictures (0 = the integer one)
    Q4 *q4tab = nullptr; Q8 *q8tab = nullptr; int *lambda_tab = nullptr; uint16_t *cost_all = nullptr;
    // optional per-stage profiling: (NSTAGE+1) events per armed call
    hipEvent_t *ev = nullptr;
    int *ev_mask = nullptr;       // per call: bit i = stage i ran
    int prof_calls = 0, prof_cap = 0;
};
enum { NSTAGE = 6 };

static const char *const kStageNames[] = { "ingest", "macroblocks", "unused", "settle_qp", "deblock", "hpel_filter" };

// every quantiser-dependent value for all 52 quantisers (adaptive quantisation reads them per macroblock)
static int build_aq_tables(x264gpu_encoder *e)
{
    QuantCfg qc; qc.deadzone_inter = e->cfg.deadzone_inter; qc.deadzone_intra = e->cfg.deadzone_intra;
    std::vector<Q4> q4(52 * 4); std::vector<Q8> q8(52 * 2); std::vector<int> lam(52);
    std::vector<uint16_t> cost((size_t)52 * 2 * MVCOST_HALF);
    for (int qp = 0; qp < 52; qp++) {
        for (int l = 0; l < 4; l++) q4[qp * 4 + l] = make_q4(qp, l, qc);
        for (int l = 0; l < 2; l++) q8[qp * 2 + l] = make_q8(qp, l, qc);
        lam[qp] = lambda_of(qp);
        uint16_t *h = cost.data() + (size_t)qp * 2 * MVCOST_HALF;
        for (int i = 0; i < MVCOST_HALF; i++) {
            const float logs = i ? log2f((float)(i + 1)) * 2.0f + 1.718f : 0.718f;      // x264_analyse_init_costs
            int c = (int)((float)lam[qp] * logs + 0.5f);
            if (c > 65535) c = 65535;
            h[MVCOST_HALF + i] = (uint16_t)c; h[MVCOST_HALF - i] = (uint16_t)c;
        }
        h[0] = h[1];
    }
    hipError_t er = hipMalloc((void **)&e->q4tab, q4.size() * sizeof(Q4));
    if (er == hipSuccess) er = hipMalloc((void **)&e->q8tab, q8.size() * sizeof(Q8));
    if (er == hipSuccess) er = hipMalloc((void **)&e->lambda_tab, lam.size() * sizeof(int));
    if (er == hipSuccess) er = hipMalloc((void **)&e->cost_all, cost.size() * sizeof(uint16_t));
    if (er == hipSuccess) er = hipMalloc((void **)&e->mbqp, (size_t)e->cfg.streams * e->k.nmb);
    if (er == hipSuccess) er = hipMemcpy(e->q4tab, q4.data(), q4.size() * sizeof(Q4), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMemcpy(e->q8tab, q8.data(), q8.size() * sizeof(Q8), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMemcpy(e->lambda_tab, lam.data(), lam.size() * sizeof(int), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMemcpy(e->cost_all, cost.data(), cost.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (er != hipSuccess) return set_err(er == hipErrorOutOfMemory ? X264GPU_ENOMEM : X264GPU_EHIP, "aq tables", er);
    return X264GPU_OK;
}

extern "C" {

static int encoder_create_impl(x264gpu_encoder **out, const x264gpu_config *cfg, x264gpu_encoder *parent)
{
    ARG_TRY(out && cfg);
    ARG_TRY(cfg->width >= 16 && cfg->height >= 16 && !(cfg->width & 1) && !(cfg->height & 1) && cfg->streams >= 1);
    ARG_TRY(cfg->qp_i >= 0 && cfg->qp_i <= 51 && cfg->qp_p >= 0 && cfg->qp_p <= 51 && cfg->refs >= 1 && cfg->refs <= 5);
    ARG_TRY(cfg->trellis == 0 || (cfg->trellis > 0 && cfg->trellis < 128 && (cfg->trellis & 63) && cfg->rd && cfg->cabac));      // trellis sites (mask; x264 --trellis 1 = 63, --trellis 2 = 63 + 64): RD sessions with CABAC
    ARG_TRY(!cfg->rd || (cfg->subme >= 6 && cfg->subme <= 9 && cfg->psy_rd_q8 >= 0 && cfg->psy_rd_q8 <= 2560));      // RD: x264's i_mbrd 1 (bit counts of the session's entropy coder)
    // rd > 1: RD refinement of the chosen type (x264 subme 8, i_mbrd 2): bit 0 on + a mask of refinement sites in bits 1..5 (x264 = all five: 63); CABAC, hex / umh
    ARG_TRY(cfg->rd >= 0 && cfg->rd < 128 && (cfg->rd < 2 || (cfg->rd & 1)) && (!(cfg->rd & 62) || (cfg->cabac && cfg->subme >= 8 && (cfg->me_method == 1 || cfg->me_method == 2))));      // bit 6: deblock-aware RD (x264 subme 9)      // (rd 0 with subme >= 8: the sub-pel iteration table of those levels without RD, as before)
    // --subme 8 and up with RD: 4 half-pel + 10 quarter-pel iterations reach 4.5 samples from the full-pel vector — only the refinement instantiations stage
    // that neighbourhood (CABAC, hex / umh); B slices search that far from subme 9 on (one level down)
    ARG_TRY(!cfg->rd || cfg->subme < 8 || (cfg->cabac && (cfg->me_method == 1 || cfg->me_method == 2)));
    ARG_TRY(!(cfg->rd & 64) || (cfg->cabac && (cfg->me_method == 1 || cfg->me_method == 2)));          // deblock-aware RD: in the refinement instantiations
    ARG_TRY(cfg->dpb == 0 || cfg->subme < 9 || (cfg->rd && cfg->cabac && (cfg->me_method == 1 || cfg->me_method == 2)));
    ARG_TRY(cfg->slices >= 0 && (cfg->slices <= 1 || cfg->slices <= (cfg->height + 15) / 16 / (cfg->slices_plain ? 1 : 4)));      // x264 slice threads: at least four macroblock rows each; --slices N: one
    ARG_TRY(cfg->slices_plain == 0 || cfg->slices_plain == 1);
    ARG_TRY(cfg->width <= 4096 && cfg->height <= 2304 && cfg->me_range >= 4 && cfg->me_range <= (cfg->me_method == 2 ? 64 : 16));
    ARG_TRY(cfg->me_method >= 0 && cfg->me_method <= 3);
    ARG_TRY(cfg->dpb == 0 || (cfg->dpb >= cfg->refs && cfg->dpb <= 7));
    x264gpu_encoder *e = new (std::nothrow) x264gpu_encoder();
    if (!e) return set_err(X264GPU_ENOMEM, "encoder", hipSuccess);
    e->cfg = *cfg;
    EncK &k = e->k;
    memset(&k, 0, sizeof(k));
    k.w = cfg->width; k.h = cfg->height;
    k.mbw = (k.w + 15) / 16; k.mbh = (k.h + 15) / 16; k.nmb = k.mbw * k.mbh;
    k.cw = k.mbw * 16; k.ch = k.mbh * 16;
    k.fs = (k.cw + 63) / 64 * 64;
    k.rs = (k.cw + 2 * PAD + 63) / 64 * 64;
    k.fency_bytes = (size_t)k.fs * k.ch;
    k.fencuv_bytes = (size_t)k.fs * k.ch / 2;
    k.plane_bytes = (size_t)k.rs * (k.ch + 2 * PAD);
    k.luma_bytes = 4 * k.plane_bytes;
    k.cplane_bytes = (size_t)k.rs * (k.ch / 2 + 2 * CPAD);
    k.me_range = cfg->me_range; k.subme = cfg->subme; k.dct_decimate = cfg->dct_decimate;
    k.slices = cfg->slices > 1 ? cfg->slices : 1; k.slices_plain = cfg->slices_plain != 0; k.cabac = cfg->cabac != 0;
    k.rd = cfg->rd; k.psy = cfg->psy != 0; k.psy_rd_q8 = cfg->psy_rd_q8;
    k.partitions = cfg->partitions; k.chroma_qp_offset = cfg->chroma_qp_offset; k.dct8x8 = cfg->dct8x8; k.me_method = cfg->me_method; k.chroma_me = cfg->chroma_me != 0; k.mixed_refs = cfg->mixed_refs != 0;
    k.alpha_off = cfg->deblock_alpha * 2; k.beta_off = cfg->deblock_beta * 2;
    k.deblock_rdo = cfg->deblock && (cfg->rd & 64) ? 1 : 0;
    const size_t S = (size_t)cfg->streams;
    hipError_t er = hipSuccess;
    auto alloc = [&](void **p, size_t n, int fill) {
        if (er != hipSuccess) return;
        er = hipMalloc(p, n);
        if (er == hipSuccess) er = hipMemset(*p, fill, n);
    };
    alloc((void **)&e->fenc_y, S * k.fency_bytes, 0);
    alloc((void **)&e->fenc_uv, S * k.fencuv_bytes, 0);
    e->slots = (cfg->dpb > 0 ? cfg->dpb : cfg->refs) + 1; e->ring = cfg->refs + 1;
    if (parent) {
        e->view_of = parent; e->meta = &parent->meta_own;
        for (int i = 0; i < 8; i++) { e->luma[i] = parent->luma[i]; e->chroma[i] = parent->chroma[i]; e->mv16[i] = parent->mv16[i]; e->mbtype[i] = parent->mbtype[i];
                                      e->colref[i] = parent->colref[i]; e->colmv[i] = parent->colmv[i]; e->colref0[i] = parent->colref0[i]; }
    } else {
    for (int i = 0; i < e->slots; i++) {
        alloc((void **)&e->luma[i], S * k.luma_bytes, 0);
        alloc((void **)&e->chroma[i], S * k.cplane_bytes, 0);
    }
    for (int i = 0; i < e->slots; i++) {
        alloc((void **)&e->mv16[i], S * k.nmb * 2 * sizeof(int16_t), 0);
        alloc((void **)&e->mbtype[i], S * k.nmb, 0);
        if (cfg->dpb > 0) { alloc((void **)&e->colref[i], S * k.nmb * 4, 0); alloc((void **)&e->colmv[i], S * k.nmb * 8 * sizeof(int16_t), 0); alloc((void **)&e->colref0[i], S * k.nmb * 4, 0xff); }
    }
    }
    for (int r = 1; r < (cfg->dpb > 0 ? 8 : cfg->refs); r++) alloc((void **)&e->mvr[r], S * k.nmb * 2 * sizeof(int16_t), 0);
    alloc((void **)&e->wf_progress, S * 2 * WFG_ROWS * sizeof(int), 0);
    if (cfg->rd && !cfg->cabac) alloc((void **)&e->tc, S * k.nmb * 24, 0);
    if (cfg->rd && cfg->cabac) { alloc((void **)&e->amvd, S * k.nmb * (cfg->dpb > 0 ? 16 : 8), 0); alloc((void **)&e->cab_out, S * (cfg->slices > 1 ? cfg->slices : 1) * 192 * sizeof(uint32_t), 0); }
    if (S >= 512 && !getenv("X264GPU_NO_BALANCE")) { alloc((void **)&e->perm, S * sizeof(int), 0); alloc((void **)&e->wtime, 4 * S * sizeof(unsigned), 0); }
    if (cfg->slices_plain && cfg->slices > 1) { alloc((void **)&e->sl_stat, 3 * S * cfg->slices * 4 * sizeof(int), 0);      /* one history per picture kind: P, B reference, B */ alloc((void **)&e->sl_rerun, S * cfg->slices * sizeof(int), 0); }
#ifdef MB_PROF
    alloc((void **)&e->prof, S * 32 * sizeof(unsigned long long), 0);
#endif
    if (er != hipSuccess) { x264gpu_encoder_destroy(e); return set_err(er == hipErrorOutOfMemory ? X264GPU_ENOMEM : X264GPU_EHIP, "encoder buffers", er); }
    const int rc = build_aq_tables(e);       // the macroblock loop reads every quantiser-dependent value per macroblock
    if (rc) { x264gpu_encoder_destroy(e); return rc; }
    // the device tables the macroblock loop reads (the chain table of the CABAC size pricing, the trellis quantiser's): built here, once per device, so that the
    // encode path allocates and uploads nothing (no host synchronisation in it; they live as long as the library is loaded: one set per device, shared by every encoder)
    if (cfg->rd && cfg->cabac) { const uint32_t *ct; const int r2 = cabac_chain_table(&ct); if (r2 != X264GPU_OK) { x264gpu_encoder_destroy(e); return r2; } }
    if (cfg->trellis) { const uint16_t *a; const uint8_t *b; const int *c; const int r2 = trellis_table_ptrs(&a, &b, &c); if (r2 != X264GPU_OK) { x264gpu_encoder_destroy(e); return r2; } }
    *out = e;
    return X264GPU_OK;
}
int x264gpu_encoder_create(x264gpu_encoder **out, const x264gpu_config *cfg) { return encoder_create_impl(out, cfg, nullptr); }
int x264gpu_encoder_create_view(x264gpu_encoder **out, x264gpu_encoder *parent)
{
    ARG_TRY(out && parent && !parent->view_of && parent->cfg.dpb > 0);          // views of x264gpu_encode_pictures sessions (the DPB model names every slot)
    return encoder_create_impl(out, &parent->cfg, parent);
}

int x264gpu_encoder_set_qp(x264gpu_encoder *e, int qp_i, int qp_p)
{
    ARG_TRY(e && qp_i >= 0 && qp_i <= 51 && qp_p >= 0 && qp_p <= 51);
    e->cfg.qp_i = qp_i; e->cfg.qp_p = qp_p;
    return X264GPU_OK;
}

int x264gpu_encoder_set_mb_qp_offsets(x264gpu_encoder *e, const float *d_offsets)
{
    ARG_TRY(e);
    e->ext_off = d_offsets;
    return X264GPU_OK;
}

// every stream its own slice quantiser and its float quantiser (qpms == NULL: the integer one as a float): two sets, used in turn
static int set_stream_qps_f(x264gpu_encoder *e, const int8_t *qps, const float *qpms)
{
    const size_t S = (size_t)e->cfg.streams;
    for (size_t s = 0; s < S; s++) ARG_TRY(qps[s] >= 0 && qps[s] <= 51);
    if (!e->stream_qp) HIP_TRY(hipMalloc((void **)&e->stream_qp, 2 * S));
    if (!e->stream_qpm) HIP_TRY(hipMalloc((void **)&e->stream_qpm, 2 * S * sizeof(float)));
    e->stream_qp_sel ^= 1;
    std::vector<float> f(S);
    for (size_t s = 0; s < S; s++) f[s] = qpms && qpms[s] != 0.f ? qpms[s] : (float)qps[s];
    HIP_TRY(hipMemcpy(e->stream_qp + (size_t)e->stream_qp_sel * S, qps, S, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->stream_qpm + (size_t)e->stream_qp_sel * S, f.data(), S * sizeof(float), hipMemcpyHostToDevice));
    e->use_stream_qp = true;
    return X264GPU_OK;
}

int x264gpu_encoder_set_stream_qps(x264gpu_encoder *e, const int8_t *qps)
{
    ARG_TRY(e);
    if (!qps) { e->use_stream_qp = false; return X264GPU_OK; }
    return set_stream_qps_f(e, qps, nullptr);
}
int x264gpu_encoder_set_stream_qpms(x264gpu_encoder *e, const int8_t *qps, const float *qpms)
{
    ARG_TRY(e);
    if (!qps) { e->use_stream_qp = false; return X264GPU_OK; }
    if (qpms) for (int s = 0; s < e->cfg.streams; s++) ARG_TRY(qpms[s] == 0.f || (qpms[s] > (float)qps[s] - 1.f && qpms[s] < (float)qps[s] + 1.f));
    return set_stream_qps_f(e, qps, qpms);
}
int x264gpu_encoder_set_qpm(x264gpu_encoder *e, float qpm)
{
    ARG_TRY(e && qpm >= 0.f && qpm < 52.f);
    e->qpm_next = qpm;
    return X264GPU_OK;
}

static void profile_free(x264gpu_encoder *e)
{
    if (e->ev) {
        for (int i = 0; i < e->prof_cap * (NSTAGE + 1); i++) (void)hipEventDestroy(e->ev[i]);
        delete[] e->ev; delete[] e->ev_mask;
    }
    e->ev = nullptr; e->ev_mask = nullptr; e->prof_cap = e->prof_calls = 0;
}

int x264gpu_encoder_profile_begin(x264gpu_encoder *e, int max_calls)
{
    ARG_TRY(e && max_calls > 0 && max_calls <= 100000);
    profile_free(e);
    e->ev = new (std::nothrow) hipEvent_t[(size_t)max_calls * (NSTAGE + 1)];
    e->ev_mask = new (std::nothrow) int[max_calls];
    if (!e->ev || !e->ev_mask) return set_err(X264GPU_ENOMEM, "profile events", hipSuccess);
    for (int i = 0; i < max_calls * (NSTAGE + 1); i++) HIP_TRY(hipEventCreate(&e->ev[i]));
    e->prof_cap = max_calls;
    return X264GPU_OK;
}

int x264gpu_encoder_profile_end(x264gpu_encoder *e, void *stream, double *ms_sum, int *launches)
{
    ARG_TRY(e && ms_sum && launches);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < NSTAGE; i++) { ms_sum[i] = 0; launches[i] = 0; }
    for (int c = 0; c < e->prof_calls; c++) {
        hipEvent_t *ev = e->ev + (size_t)c * (NSTAGE + 1);
        for (int i = 0; i < NSTAGE; i++) {
            if (!(e->ev_mask[c] >> i & 1)) continue;
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            ms_sum[i] += ms; launches[i]++;
        }
    }
    profile_free(e);
    return X264GPU_OK;
}

void x264gpu_encoder_destroy(x264gpu_encoder *e)
{
    if (!e) return;
    profile_free(e);
    (void)hipFree(e->fenc_y); (void)hipFree(e->fenc_uv);
    if (!e->view_of)
    for (int i = 0; i < 8; i++) { (void)hipFree(e->luma[i]); (void)hipFree(e->chroma[i]); (void)hipFree(e->mv16[i]); (void)hipFree(e->mbtype[i]); (void)hipFree(e->colref[i]); (void)hipFree(e->colmv[i]); (void)hipFree(e->colref0[i]); }
    (void)hipFree(e->direct_flags_base); (void)hipFree(e->dscore);
    for (int i = 0; i < 8; i++) (void)hipFree(e->mvr[i]);
    (void)hipFree(e->wf_progress);
    (void)hipFree(e->tc);
    (void)hipFree(e->amvd);
    (void)hipFree(e->cab_out);
    (void)hipFree(e->sl_stat); (void)hipFree(e->sl_rerun); (void)hipFree(e->perm); (void)hipFree(e->wtime);
    (void)hipFree(e->prof);
    (void)hipFree(e->stream_qp); (void)hipFree(e->stream_qpm); (void)hipFree(e->mbqp); (void)hipFree(e->q4tab); (void)hipFree(e->q8tab); (void)hipFree(e->lambda_tab); (void)hipFree(e->cost_all);
    delete e;
}

int x264gpu_encoder_mb_count(const x264gpu_encoder *e) { return e ? e->k.nmb : 0; }
int x264gpu_encoder_set_debug(x264gpu_encoder *e, void *d_counters) { ARG_TRY(e); e->dbg = (unsigned long long *)d_counters; return X264GPU_OK; }
int x264gpu_encoder_stage_count(void) { return (int)(sizeof(kStageNames) / sizeof(kStageNames[0])); }
const char *x264gpu_encoder_stage_name(int i) { return i >= 0 && i < x264gpu_encoder_stage_count() ? kStageNames[i] : ""; }

// MB_PROF builds (tools/mb_prof.py): the phase counters of the last macroblock-loop launch, [streams][32] cycle counts; not part of the ABI
int x264gpu_encoder_mb_prof(x264gpu_encoder *e, unsigned long long *out)
{
    ARG_TRY(e && out);
    if (!e->prof) return X264GPU_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, e->prof, (size_t)e->cfg.streams * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return X264GPU_OK;
}

// tests: the CABAC context variables ((pStateIdx << 1) | valMPS, 460 of them) the wavefront of (stream, slice) ended the last picture with —
// RD sessions with cabac only.  They must equal what the arithmetic coder of the host (and so a decoder) holds at the end of that slice.
int x264gpu_encoder_cabac_states(x264gpu_encoder *e, int stream, int slice, uint8_t *out460)
{
    ARG_TRY(e && out460 && stream >= 0 && stream < e->cfg.streams && slice >= 0 && slice < (e->cfg.slices > 1 ? e->cfg.slices : 1));
    if (!e->cab_out) return X264GPU_EINVAL;
    uint32_t w[192];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(w, e->cab_out + ((size_t)stream * (e->cfg.slices > 1 ? e->cfg.slices : 1) + slice) * 192, sizeof(w), hipMemcpyDeviceToHost));
    for (int c = 0; c < 460; c++) {
        int reg, lane, sh;
        out460[c] = cab_locate(c, reg, lane, sh) ? (uint8_t)(w[reg * 64 + lane] >> sh) : 0;
    }
    return X264GPU_OK;
}

int x264gpu_encoder_set_lowres_mvs(x264gpu_encoder *e, const int16_t *d_mvs) { ARG_TRY(e); e->lowres_mv = d_mvs; return X264GPU_OK; }
int x264gpu_encoder_set_lowres_mvs1(x264gpu_encoder *e, const int16_t *d_mvs) { ARG_TRY(e); e->lowres_mv1 = d_mvs; return X264GPU_OK; }

// Load balance of the lock-step batch.  The macroblock loop is one wavefront a stream and the chip holds two of them per SIMD: a launch lasts as long
// as its slowest SIMD, i.e. as the largest SUM of two streams' work — and streams differ (content): measured slowest / mean stream 1.08 - 1.25.
// Workgroups are dispatched in index order and the second half of them lands on SIMDs that already hold one of the first half, so the streams are
// ordered by the time they took in the last picture of the same kind: the slowest first (longest job first), and the second half in REVERSE order
// (the fastest stream joins the slowest on its SIMD).  No macroblock's arithmetic changes — only which workgroup codes which stream.
__global__ void __launch_bounds__(1024) k_balance(const unsigned *wtime, int *perm, int streams)
{
    // rank of every stream by (time descending, index ascending): streams <= a few thousand, an O(n^2 / threads) count per thread
    for (int s = threadIdx.x; s < streams; s += blockDim.x) {
        const unsigned t = wtime[s];
        int rank = 0;
        for (int o = 0; o < streams; o++) { const unsigned u = wtime[o]; rank += (u > t || (u == t && o < s)) ? 1 : 0; }
        const int half = (streams + 1) / 2;
        // rank r (0 = slowest): the slower half keeps its order, the faster half is dealt backwards behind it
        perm[rank < half ? rank : half + (streams - 1 - rank)] = s;
    }
}

// --slices N in P pictures (EncK.sl_stat): with the intra counts the slices reported, does every slice's window of harmless prior counts hold the
// sum of the counts before it?  A slice whose window misses runs again on the sum as it stands now.  One thread per stream.
// guess: before the first pass — the counts of the P picture before this one are the assumption (any assumption gives the same result;
// a good one saves passes).
__global__ void k_slice_priors(EncK k, int streams, int guess)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= streams) return;
    int sum = 0;
    for (int sl = 0; sl < k.slices; sl++) {
        int *st = k.sl_stat + ((size_t)s * k.slices + sl) * 4;
        const bool again = guess || !(st[1] <= sum && sum < st[2]);
        k.sl_rerun[(size_t)s * k.slices + sl] = again;
        if (again) st[3] = sum;
        sum += st[0];
    }
}

}  // extern "C"

// x264gpu_pack_levels: one wavefront per stream walks its macroblocks in order; lane l < LEVELS / 8 holds 8 levels (16 bytes) of the macroblock, two lanes a group.
// In place: the packed position of a group is never beyond its own, and a macroblock is read before anything is written at or beyond its start.
__global__ __launch_bounds__(64) void k_pack_levels(int16_t *lv, x264gpu_level_index *index, uint32_t *kept, int nmb)
{
    constexpr int Q = X264GPU_MB_LEVELS / 8;          // 16-byte quads a macroblock (52)
    static_assert(X264GPU_MB_LEVELS % 16 == 0 && Q <= 64, "two lanes a group of 16 levels");
    const int s = blockIdx.x, lane = threadIdx.x;
    uint4 *base = reinterpret_cast<uint4 *>(lv + (size_t)s * nmb * X264GPU_MB_LEVELS);
    x264gpu_level_index *ix = index + (size_t)s * nmb;
    unsigned at = 0;          // groups kept so far
    uint4 nxt = lane < Q && nmb > 0 ? base[lane] : make_uint4(0, 0, 0, 0);
    for (int i = 0; i < nmb; i++) {
        const uint4 v = nxt;
        if (i + 1 < nmb && lane < Q) nxt = base[(size_t)(i + 1) * Q + lane];          // (the next macroblock's quads are read before this one's are written: they may land there)
        const unsigned long long nzq = __ballot((v.x | v.y | v.z | v.w) != 0);
        const unsigned groups = (unsigned)__ballot(lane < Q / 2 && ((nzq >> (2 * lane)) & 3));
        if (lane < Q && (groups >> (lane >> 1) & 1)) base[(size_t)(at + __popc(groups & ((1u << (lane >> 1)) - 1u))) * 2 + (lane & 1)] = v;
        if (lane == 0) { ix[i].at = at; ix[i].groups = groups; }
        at += __popc(groups);
    }
    if (kept && lane == 0) kept[s] = at;
}

extern "C" int x264gpu_pack_levels(int16_t *d_levels, int streams, int mb_count, x264gpu_level_index *d_index, uint32_t *d_kept, void *stream)
{
    if (!d_levels || !d_index || streams < 1 || mb_count < 1) return set_err(X264GPU_EINVAL, "x264gpu_pack_levels", hipSuccess);
    hipLaunchKernelGGL(k_pack_levels, dim3(streams), dim3(64), 0, (hipStream_t)stream, d_levels, d_index, d_kept, mb_count);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

// what a later B picture's spatial direct prediction reads of this picture when it heads that picture's list 1 (x264 frame->ref[] / mv[] of the
// co-located macroblock): per 8x8 block the reference index it used — list 0's, else list 1's, -1 for intra — and that vector
__global__ void k_col_from_records(EncK k)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i >= k.nmb * 4) return;
    const int mbi = i >> 2, b8 = i & 3;
    const x264gpu_mb *m = k.mb + (size_t)s * k.nmb + mbi;
    const int t = m->type;
    int r = -1, vx = 0, vy = 0;
    if (t > X264GPU_MB_I16x16) {
        const bool use1 = t >= X264GPU_MB_B_DIRECT && m->ref[b8] < 0;
        r = use1 ? m->ref1[b8] : m->ref[b8]; vx = use1 ? m->mv1[b8][0] : m->mv[b8][0]; vy = use1 ? m->mv1[b8][1] : m->mv[b8][1];
    }
    k.colref_cur[((size_t)s * k.nmb + mbi) * 4 + b8] = (int8_t)r;
    k.colref0_cur[((size_t)s * k.nmb + mbi) * 4 + b8] = (int8_t)(t > X264GPU_MB_I16x16 ? m->ref[b8] : -1);
    int16_t *o = k.colmv_cur + (((size_t)s * k.nmb + mbi) * 4 + b8) * 2;
    o[0] = (int16_t)vx; o[1] = (int16_t)vy;
}

// One picture per stream, every stream with the SAME structure (slice type, destination slot, reference lists, POC); quantisers may differ
// per stream (x264gpu_encoder_set_stream_qps) or be pic.qp for all
static int encode_core(x264gpu_encoder *e, const uint8_t *d_i420, const x264gpu_pic &pic, x264gpu_mb *d_mb, int16_t *d_levels, hipStream_t st)
{
    int slice_type = pic.slice_type;
    e->last_stream = st;
    if (slice_type == X264GPU_SLICE_I_NONIDR) slice_type = X264GPU_SLICE_I;            // same kernels; only the DPB handling differs
    const bool bslice = slice_type == X264GPU_SLICE_B;
    ARG_TRY(pic.dst >= 0 && pic.dst < e->slots && pic.qp >= 0 && pic.qp <= 51);
    ARG_TRY(!bslice || e->cfg.dpb > 0);      // B pictures: with RD (subme >= 7) CABAC sizes only; below, x264 analyses B slices without RD
    const int n0 = slice_type == X264GPU_SLICE_I ? 0 : pic.nref[0], n1 = bslice ? pic.nref[1] : 0;
    ARG_TRY(pic.qpm == 0.f || (pic.qpm > (float)pic.qp - 1.f && pic.qpm < (float)pic.qp + 1.f));          // qp is the rounding of qpm
    ARG_TRY(n0 >= 0 && n0 <= 7 && n1 >= 0 && n1 <= 3 && n0 + n1 <= 8 && (slice_type == X264GPU_SLICE_I || n0 > 0) && (!bslice || n1 > 0));      // list 0: up to 5 pictures + --weightp duplicates
    for (int l = 0; l < 2; l++) for (int r = 0; r < (l ? n1 : n0); r++) ARG_TRY(pic.slot[l][r] >= 0 && pic.slot[l][r] < e->slots && pic.slot[l][r] != pic.dst);
    const int S = e->cfg.streams;
    EncK k = e->k;
    const int qp = pic.qp;
    e->poc = pic.poc;
    k.i420 = d_i420; k.fenc_y = e->fenc_y; k.fenc_uv = e->fenc_uv;
    const int cur = pic.dst;
    k.rec_luma = e->luma[cur]; k.rec_chroma = e->chroma[cur];
    k.nref = n0; k.nref1 = n1;
    // --weightp: explicit luma weights of a P picture's list 0 and x264's blind duplicate of reference 0
    k.blind_dupe = 0; k.wp_any = 0;
    for (int r = 0; r < 8; r++) { k.wl0[r] = 0; k.refpic[r] = r; k.wc0[2 * r] = k.wc0[2 * r + 1] = 0; }
    k.wc_any = 0;
    if (slice_type == X264GPU_SLICE_P) {
        for (int r = 0; r < n0; r++)
            if (pic.wl0[r].on) {
                ARG_TRY(pic.wl0[r].denom >= 0 && pic.wl0[r].denom <= 7 && pic.wl0[r].scale >= -128 && pic.wl0[r].scale <= 127 && pic.wl0[r].offset >= -128 && pic.wl0[r].offset <= 127);
                k.wl0[r] = (int)(uint8_t)(int8_t)pic.wl0[r].offset | (int)(uint8_t)(int8_t)pic.wl0[r].scale << 8 | pic.wl0[r].denom << 16 | 1 << 24;
                k.wp_any = 1;
            }
        for (int r = 0; r < n0; r++)
            for (int c = 0; c < 2; c++)
                if (pic.wc0[r].on[c]) {
                    ARG_TRY(pic.wc0[r].denom >= 0 && pic.wc0[r].denom <= 7 && pic.wc0[r].scale[c] >= -128 && pic.wc0[r].scale[c] <= 127 && pic.wc0[r].offset[c] >= -128 && pic.wc0[r].offset[c] <= 127);
                    k.wc0[2 * r + c] = (int)(uint8_t)(int8_t)pic.wc0[r].offset[c] | (int)(uint8_t)(int8_t)pic.wc0[r].scale[c] << 8 | pic.wc0[r].denom << 16 | 1 << 24;
                    k.wp_any = 1; k.wc_any = 1;
                }
        if (pic.blind_dupe > 0) {
            ARG_TRY(pic.blind_dupe == 1 && n0 >= 2 && pic.slot[0][1] == pic.slot[0][0]);      // x264 places it right behind reference 0
            k.blind_dupe = 1; k.wp_any = 1;
        }
        // duplicates (x264 places them right behind reference 0: the offset - 1 copy, and under a weighted reference 0 its unweighted copy)
        int npics = 0;
        for (int r = 0; r < n0; r++) {
            int first = r;
            for (int q = 0; q < r; q++) if (pic.slot[0][q] == pic.slot[0][r]) { first = q; break; }
            k.refpic[r] = first == r ? npics++ : k.refpic[first];
            if (first != r) k.wp_any = 1;
        }
    }
    auto slot_of = [&](int ri) { return (int)(ri < n0 ? pic.slot[0][ri] : pic.slot[1][ri - n0]); };      // combined index: list 0, then list 1
    for (int r = 0; r < 8; r++) {
        const int slot = n0 + n1 > 0 ? slot_of(r < n0 + n1 ? r : 0) : cur;
        k.ref_luma[r] = e->luma[slot]; k.ref_chroma[r] = e->chroma[slot];
    }
    // motion side data (x264: h->mb.mvr, fref[0][0]->mv16x16 / mb_type / i_ref, POC distances)
    const int s0 = n0 ? slot_of(0) : cur;
    k.mv16_cur = e->mv16[cur]; k.mv16_ref0 = e->mv16[s0]; k.mbtype_cur = e->mbtype[cur]; k.mbtype_ref0 = e->mbtype[s0];
    for (int r = 0; r < 8; r++) k.mvr[r] = e->mvr[r];
    k.temporal = k.nref > 0 && e->meta->slot_nref[s0] > 0;
    for (int r = 0; r < 8; r++) k.tscale[r] = 0;
    if (k.temporal) {
        const int delta = e->meta->slot_poc[s0] - e->meta->slot_ref0poc[s0], inv = (256 + delta / 2) / delta;
        for (int r = 0; r < n0 + n1; r++) k.tscale[r] = (e->poc - e->meta->slot_poc[slot_of(r)]) * inv;
    }
    memset(k.biw, 32, sizeof(k.biw));
    if (bslice) {
        // x264_macroblock_bipred_init: implicit weights from the POC distances (8.4.2.3.1)
        for (int r0 = 0; r0 < n0; r0++)
            for (int r1 = 0; r1 < n1; r1++) {
                const int poc0 = e->meta->slot_poc[pic.slot[0][r0]], poc1 = e->meta->slot_poc[pic.slot[1][r1]];
                const int td = min(max(poc1 - poc0, -128), 127);
                int dsf = 256;
                if (td) { const int tb = min(max(e->poc - poc0, -128), 127), tx = (16384 + (abs(td) >> 1)) / td; dsf = min(max((tb * tx + 32) >> 6, -1024), 1023); }
                dsf >>= 2;
                const int w = (e->cfg.weightb && dsf >= -64 && dsf <= 128) ? 64 - dsf : 32;
                ARG_TRY(w > 0 && w < 64);      // list 0 before, list 1 after the picture: always inside (the device averages in 16-bit lanes)
                k.biw[r0][r1] = (uint8_t)w;
            }
        k.colref = e->colref[pic.slot[1][0]]; k.colmv = e->colmv[pic.slot[1][0]];
        // --direct temporal / auto (x264_macroblock_slice_init: map_col_to_list0, dist_scale_factor[r][0])
        const int cs = pic.slot[1][0];
        k.colref0 = e->colref0[cs]; k.coltype = e->mbtype[cs];
        k.direct_auto = pic.direct_auto != 0;
        k.direct_flags = e->use_direct_flags ? e->direct_flags : nullptr;
        for (int i = 0; i < 8; i++) {
            k.map_col[i] = -2; k.dist_scale[i] = 256;
            if (i < e->meta->slot_nref[cs]) for (int j = 0; j < n0; j++) if (e->meta->slot_poc[pic.slot[0][j]] == e->meta->slot_l0poc[cs][i]) { k.map_col[i] = j; break; }
            if (i < n0) {
                const int poc0 = e->meta->slot_poc[pic.slot[0][i]], poc1 = e->meta->slot_poc[pic.slot[1][0]];
                const int td = min(max(poc1 - poc0, -128), 127);
                if (td) { const int tb = min(max(e->poc - poc0, -128), 127), tx = (16384 + (abs(td) >> 1)) / td; k.dist_scale[i] = min(max((tb * tx + 32) >> 6, -1024), 1023); }
            }
        }
        if (k.direct_auto) {
            const size_t nds = (size_t)S * (e->cfg.slices > 1 ? e->cfg.slices : 1) * 2;          // [stream][slice][temporal, spatial]: every slice's wavefront stores its own
            if (!e->dscore) HIP_TRY(hipMalloc((void **)&e->dscore, nds * sizeof(int)));
            HIP_TRY(hipMemsetAsync(e->dscore, 0, nds * sizeof(int), st));
        }
        k.dscore = e->dscore;
    }
    k.colref_cur = e->colref[cur]; k.colmv_cur = e->colmv[cur]; k.colref0_cur = e->colref0[cur];
    e->meta->slot_nref[cur] = k.nref; e->meta->slot_poc[cur] = e->poc; e->meta->slot_ref0poc[cur] = k.nref ? e->meta->slot_poc[s0] : 0;
    for (int r = 0; r < 8; r++) e->meta->slot_l0poc[cur][r] = r < k.nref && slice_type != X264GPU_SLICE_I ? e->meta->slot_poc[pic.slot[0][r]] : 0;
    k.prof = e->prof; k.tc = e->tc; k.amvd = e->amvd; k.cab_out = e->cab_out;
    {
        // load balance: this picture's workgroup -> stream order from the times of the last picture of its kind; this picture's times replace them
        const int kind = slice_type == X264GPU_SLICE_I ? 0 : !bslice ? 1 : pic.keep ? 2 : 3;
        k.perm = nullptr; k.wtime = e->wtime ? e->wtime + (size_t)kind * S : nullptr;
        if (e->perm && e->wt_valid[kind] && k.slices <= 1 && S <= 8192) { hipLaunchKernelGGL(k_balance, dim3(1), dim3(1024), 0, st, k.wtime, e->perm, S); k.perm = e->perm; }
        if (k.wtime) e->wt_valid[kind] = true;
    }
    k.sl_stat = slice_type != X264GPU_SLICE_I && e->sl_stat ? e->sl_stat + (size_t)(!bslice ? 0 : pic.keep ? 1 : 2) * S * (size_t)e->cfg.slices * 4 : nullptr;      // the first guess: the last picture of the same kind
    k.sl_rerun = e->sl_rerun; k.sl_pass = 0;
    k.trellis = e->cfg.trellis; k.tr_su = nullptr; k.tr_tu = nullptr; k.tr_l2 = nullptr;
    if (k.trellis) { const int rc = trellis_table_ptrs(&k.tr_su, &k.tr_tu, &k.tr_l2); if (rc != X264GPU_OK) return rc; }
    k.ctab = nullptr;
    if (k.rd && k.cabac) { const int rc = cabac_chain_table(&k.ctab); if (rc != X264GPU_OK) return rc; }
    k.lowres_mv = e->lowres_mv; k.lowres_mv1 = e->lowres_mv1; k.fast_pskip = e->cfg.fast_pskip; k.mv_range = e->cfg.mv_range;
    k.mb = d_mb; k.levels = d_levels;
    k.qp = qp; k.lambda = lambda_of(qp); k.qpc = chroma_qp_of(qp, e->cfg.chroma_qp_offset);
    k.slice_type = slice_type;
    k.dct_decimate = e->cfg.dct_decimate || bslice;      // x264: B slices decimate whatever --no-dct-decimate says (h->mb.b_dct_decimate)
    k.partitions = (slice_type == X264GPU_SLICE_I && (e->cfg.partitions & 0x100)) ? (e->cfg.partitions >> 8) & 6 : e->cfg.partitions & 7;
    if (bslice && (e->cfg.partitions & 0x100)) k.partitions = (k.partitions & 6) | ((e->cfg.partitions >> 11) & 1);      // --partitions b8x8 (bit 11) instead of p8x8
    k.dbg = e->dbg;

    hipEvent_t *ev = nullptr;
    int mask = 0;
    if (e->ev && e->prof_calls < e->prof_cap) ev = e->ev + (size_t)e->prof_calls * (NSTAGE + 1);
#define STAGE_MARK(i) do { if (ev) { HIP_TRY(hipEventRecord(ev[i], st)); } } while (0)
    STAGE_MARK(0);
    hipLaunchKernelGGL(k_ingest, dim3((k.cw / 4 + 255) / 256, k.ch, S), dim3(256), 0, st, k);
    // per-macroblock quantisers: always materialised (the macroblock loop reads every quantiser-dependent value per macroblock)
    const bool aq = e->cfg.aq_mode != 0 || e->ext_off != nullptr || e->use_stream_qp;
    k.stream_qp = e->use_stream_qp ? e->stream_qp + (size_t)e->stream_qp_sel * e->cfg.streams : nullptr; k.stream_qpm = e->use_stream_qp ? e->stream_qpm + (size_t)e->stream_qp_sel * e->cfg.streams : nullptr;
    k.mbqp = e->mbqp; k.q4tab = e->q4tab; k.q8tab = e->q8tab; k.lambda_tab = e->lambda_tab; k.cost_all = e->cost_all; k.aq_strength = e->cfg.aq_strength; k.qp_snap = e->cfg.aq_mode != 0; k.qpm = pic.qpm != 0.f ? pic.qpm : (float)pic.qp;
    if (e->ext_off || !e->cfg.aq_mode) hipLaunchKernelGGL(k_apply_qp_offsets, dim3((k.nmb + 255) / 256, S), dim3(256), 0, st, k, e->ext_off);
    else hipLaunchKernelGGL(k_aq, dim3((k.nmb + 15) / 16, S), dim3(256), 0, st, k);
    mask |= 1;
    STAGE_MARK(1);
    // the macroblock loop: one wavefront per stream, raster order (sub-pel neighbourhood margin 2 px up to subme 7, 5 px above)
    if (slice_type == X264GPU_SLICE_I) launch_mb_slice_intra(k, S, st);        // (RD instantiations inside, chosen by k.rd)
    else {
        const auto launch_b = k.me_method == 0 ? launch_mb_slice_b_dia : k.me_method == 2 ? launch_mb_slice_b_umh : k.me_method == 3 ? launch_mb_slice_b_esa : launch_mb_slice_b_hex;
        const auto launch_p = k.me_method == 0 ? launch_mb_slice_dia : k.me_method == 2 ? launch_mb_slice_umh : k.me_method == 3 ? launch_mb_slice_esa : launch_mb_slice_hex;
        const auto launch = [&](const EncK &kk, int streams, bool big_margin, hipStream_t s_) { if (bslice) launch_b(kk, streams, s_); else launch_p(kk, streams, big_margin, s_); };
        if (k.sl_stat) hipLaunchKernelGGL(k_slice_priors, dim3((S + 63) / 64), dim3(64), 0, st, k, S, 1);
        launch(k, S, k.subme >= 8, st);
        // --slices N: slice i is final once slices 0..i-1 are, so slices - 1 rounds of "check the assumed counts, run the slices again whose
        // decisions hang on a wrong one" always end at x264's serial result; a round in which nothing is flagged costs two empty launches
        if (k.sl_stat)
            for (int pass = 1; pass < k.slices; pass++) {
                hipLaunchKernelGGL(k_slice_priors, dim3((S + 63) / 64), dim3(64), 0, st, k, S, 0);
                k.sl_pass = pass;
                launch(k, S, k.subme >= 8, st);
            }
        k.sl_pass = 0;
    }
    if (e->cfg.dpb > 0 && pic.keep) hipLaunchKernelGGL(k_col_from_records, dim3((k.nmb * 4 + 255) / 256, S), dim3(256), 0, st, k);
    mask |= 2;
    STAGE_MARK(2);
    STAGE_MARK(3);
    if (aq) { hipLaunchKernelGGL(k_settle_qp, dim3(S, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k); mask |= 8; }      // QP_Y inheritance before the deblocking filter reads the records
    STAGE_MARK(4);
    // Few streams in flight (single-stream latency): the rows of ONE picture are dealt to several workgroups with row counters in global
    // memory and agent-scope hand-offs.  All workgroups of a launch must be resident at once: streams x workgroups <= 128.
    static const bool mwg_off = getenv("X264GPU_WAVEFRONT_1WG") != nullptr;
    const int npairs = (k.mbh + 1) / 2, dwg = (npairs + 3) / 4;
    const bool mwg = !mwg_off && S * dwg <= 128 && k.mbh > 4;
    k.wf_progress = e->wf_progress;
    if (mwg) HIP_TRY(hipMemsetAsync(e->wf_progress, 0, (size_t)S * 2 * WFG_ROWS * sizeof(int), st));
    if (e->cfg.deblock) {
        if (mwg) hipLaunchKernelGGL(k_deblock2<true>, dim3(S, dwg), dim3(256), 0, st, k);
        else {
            static const int db_waves = getenv("X264GPU_DEBLOCK_WAVES") ? atoi(getenv("X264GPU_DEBLOCK_WAVES")) : 16;     // A/B knob: 4, 8 or 16
            hipLaunchKernelGGL(k_deblock2<false>, dim3(S), dim3((db_waves == 4 || db_waves == 8 ? db_waves : 16) * 64), 0, st, k);
        }
        mask |= 16;
    }
    STAGE_MARK(5);
    if (pic.keep) {       // only pictures that will be referenced need their half-pel planes and borders (x264: fdec->b_kept_as_ref)
        launch_hpel_filter(e->luma[cur], k.plane_bytes, k.rs, k.cw, k.ch, PAD, S, k.luma_bytes, st);
        hipLaunchKernelGGL(k_chroma_border, dim3((k.cw / 2 + 2 * CPAD + 255) / 256, k.ch / 2 + 2 * CPAD, S), dim3(256), 0, st, k);
        mask |= 32;
    }
    STAGE_MARK(6);
#undef STAGE_MARK
    if (ev) e->ev_mask[e->prof_calls++] = mask;
    HIP_TRY(hipGetLastError());
    e->last = cur;
    return X264GPU_OK;
}

extern "C" {

int x264gpu_encode_frames(x264gpu_encoder *e, const uint8_t *d_i420, int slice_type, x264gpu_mb *d_mb,
                          int16_t *d_levels, void *stream)
{
    ARG_TRY(e && d_i420 && d_mb && d_levels);
    ARG_TRY(slice_type == X264GPU_SLICE_I || slice_type == X264GPU_SLICE_P || slice_type == X264GPU_SLICE_I_NONIDR);
    ARG_TRY(slice_type != X264GPU_SLICE_P || e->have > 0);
    if (slice_type == X264GPU_SLICE_I) { e->have = 0; e->poc = 0; }                    // IDR empties the DPB
    x264gpu_pic pic;
    memset(&pic, 0, sizeof(pic));
    pic.slice_type = slice_type; pic.qp = slice_type == X264GPU_SLICE_P ? e->cfg.qp_p : e->cfg.qp_i; pic.poc = e->poc; pic.dst = e->cur; pic.keep = 1; pic.qpm = e->qpm_next;
    pic.nref[0] = slice_type == X264GPU_SLICE_P ? (e->have < e->cfg.refs ? e->have : e->cfg.refs) : 0;
    for (int r = 0; r < pic.nref[0]; r++) pic.slot[0][r] = (int8_t)((e->cur - 1 - r + 2 * e->ring) % e->ring);
    const int rc = encode_core(e, d_i420, pic, d_mb, d_levels, (hipStream_t)stream);
    if (rc != X264GPU_OK) return rc;
    e->cur = (e->cur + 1) % e->ring;
    e->have++;
    e->poc += 2;
    return X264GPU_OK;
}

int x264gpu_encode_pictures(x264gpu_encoder *e, const uint8_t *d_i420, const x264gpu_pic *pics, x264gpu_mb *d_mb, int16_t *d_levels, void *stream)
{
    ARG_TRY(e && d_i420 && pics && d_mb && d_levels);
    // the streams of a call share the picture structure (lock-step GOPs: slice type, POC, DPB slots, lists, explicit weights); their quantisers —
    // integer part and the fraction of a rate-controlled session's float quantiser — may differ
    const int S = e->cfg.streams;
    bool same_qp = true;
    for (int s = 0; s < S; s++) ARG_TRY(pics[s].qpm == 0.f || (pics[s].qpm > (float)pics[s].qp - 1.f && pics[s].qpm < (float)pics[s].qp + 1.f));
    for (int s = 1; s < S; s++) {
        ARG_TRY(pics[s].slice_type == pics[0].slice_type && pics[s].poc == pics[0].poc && pics[s].dst == pics[0].dst && pics[s].keep == pics[0].keep &&
                pics[s].nref[0] == pics[0].nref[0] && pics[s].nref[1] == pics[0].nref[1] && !memcmp(pics[s].slot, pics[0].slot, sizeof(pics[0].slot)) &&
                pics[s].blind_dupe == pics[0].blind_dupe && !memcmp(pics[s].wl0, pics[0].wl0, sizeof(pics[0].wl0)) && !memcmp(pics[s].wc0, pics[0].wc0, sizeof(pics[0].wc0)));
        same_qp = same_qp && pics[s].qp == pics[0].qp && pics[s].qpm == pics[0].qpm;
    }
    if (!same_qp) {
        std::vector<int8_t> q((size_t)S); std::vector<float> f((size_t)S);
        for (int s = 0; s < S; s++) { q[(size_t)s] = (int8_t)pics[s].qp; f[(size_t)s] = pics[s].qpm; }
        const int rc = set_stream_qps_f(e, q.data(), f.data());
        if (rc != X264GPU_OK) return rc;
    }
    // --direct temporal / auto: each stream's own mode (the one field of the structure that may differ)
    e->use_direct_flags = false;
    if (pics[0].slice_type == X264GPU_SLICE_B) {
        bool any = false;
        for (int s = 0; s < S; s++) { any = any || pics[s].direct_temporal; ARG_TRY((pics[s].direct_auto != 0) == (pics[0].direct_auto != 0)); }
        if (any) {
            std::vector<uint8_t> f((size_t)S);
            for (int s = 0; s < S; s++) f[(size_t)s] = pics[s].direct_temporal != 0;
            // two halves used in turn (as stream_qp is): the B picture of the call before may still be reading the other one on the caller's stream
            if (!e->direct_flags_base) HIP_TRY(hipMalloc((void **)&e->direct_flags_base, 2 * (size_t)S));
            e->direct_flags_sel ^= 1;
            e->direct_flags = e->direct_flags_base + (size_t)e->direct_flags_sel * (size_t)S;
            HIP_TRY(hipMemcpy(e->direct_flags, f.data(), (size_t)S, hipMemcpyHostToDevice));
            e->use_direct_flags = true;
        }
    }
    const int rc = encode_core(e, d_i420, pics[0], d_mb, d_levels, (hipStream_t)stream);
    e->use_direct_flags = false;
    if (!same_qp) (void)x264gpu_encoder_set_stream_qps(e, nullptr);
    if (rc == X264GPU_OK) e->have++;
    return rc;
}

// --direct auto: the probe counts of the last B picture coded with direct_auto (synchronises with the device)
int x264gpu_encoder_direct_scores(x264gpu_encoder *e, int *h_scores)
{
    ARG_TRY(e && h_scores && e->dscore);
    HIP_TRY(hipStreamSynchronize(e->last_stream));          // (the stream of the last picture: other sessions on the device are not stalled)
    const int S = e->cfg.streams, nsl = e->cfg.slices > 1 ? e->cfg.slices : 1;
    std::vector<int> tmp((size_t)S * nsl * 2);
    HIP_TRY(hipMemcpy(tmp.data(), e->dscore, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int s = 0; s < S; s++) {
        h_scores[2 * s] = h_scores[2 * s + 1] = 0;
        for (int sl = 0; sl < nsl; sl++) { h_scores[2 * s] += tmp[((size_t)s * nsl + sl) * 2]; h_scores[2 * s + 1] += tmp[((size_t)s * nsl + sl) * 2 + 1]; }
    }
    return X264GPU_OK;
}

}  // extern "C"

// crop + de-interleave the newest reference into I420 (parity tests, PSNR)
__global__ __launch_bounds__(256) void k_get_recon(const uint8_t *__restrict__ luma00, const uint8_t *__restrict__ chroma00,
                                                   int rs, int w, int h, uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    out[(size_t)y * w + x] = luma00[(size_t)y * rs + x];
    if (y < h / 2 && x < w / 2) {
        uint8_t *u = out + (size_t)w * h, *v = u + (size_t)(w / 2) * (h / 2);
        u[(size_t)y * (w / 2) + x] = chroma00[(size_t)y * rs + 2 * x];
        v[(size_t)y * (w / 2) + x] = chroma00[(size_t)y * rs + 2 * x + 1];
    }
}

// I420 pictures -> the reconstruction planes of the slot being built (luma + NV12), one thread per luma sample pair row-wise
__global__ void k_put_recon(uint8_t *luma00, uint8_t *chroma00, int rs, int w, int h, const uint8_t *__restrict__ in, size_t luma_bytes, size_t cplane_bytes)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, s = blockIdx.z;
    const uint8_t *src = in + (size_t)s * (w * h * 3 / 2);
    uint8_t *l = luma00 + (size_t)s * luma_bytes, *c = chroma00 + (size_t)s * cplane_bytes;
    if (x < w) l[(size_t)y * rs + x] = src[(size_t)y * w + x];
    if (y < h / 2 && x < w / 2) {
        const uint8_t *u = src + (size_t)w * h, *v = u + (size_t)(w / 2) * (h / 2);
        c[(size_t)y * rs + 2 * x] = u[(size_t)y * (w / 2) + x];
        c[(size_t)y * rs + 2 * x + 1] = v[(size_t)y * (w / 2) + x];
    }
}

// A9 as a primitive: the in-loop filter alone, on `streams` given pictures (I420, width and height multiples of 16) with given macroblock
// records (types, quantisers, nnz / cbp / transform size, references, vectors), through the same kernel the frame pipeline launches.
extern "C" int x264gpu_encoder_deblock_pictures(x264gpu_encoder *e, const uint8_t *d_i420, const x264gpu_mb *d_mb, uint8_t *d_out, void *stream)
{
    ARG_TRY(e && d_i420 && d_mb && d_out && !(e->cfg.width & 15) && !(e->cfg.height & 15));
    hipStream_t st = (hipStream_t)stream;
    const int S = e->cfg.streams;
    EncK k = e->k;
    k.rec_luma = e->luma[e->cur]; k.rec_chroma = e->chroma[e->cur];
    k.mb = const_cast<x264gpu_mb *>(d_mb);
    uint8_t *l00 = e->luma[e->cur] + (size_t)PAD * k.rs + PAD, *c00 = e->chroma[e->cur] + (size_t)CPAD * k.rs + 2 * CPAD;
    hipLaunchKernelGGL(k_put_recon, dim3((k.w + 255) / 256, k.h, S), dim3(256), 0, st, l00, c00, k.rs, k.w, k.h, d_i420, k.luma_bytes, k.cplane_bytes);
    k.wf_progress = e->wf_progress;
    hipLaunchKernelGGL(k_deblock2<false>, dim3(S), dim3(16 * 64), 0, st, k);
    for (int s = 0; s < S; s++)
        hipLaunchKernelGGL(k_get_recon, dim3((k.w + 255) / 256, k.h), dim3(256), 0, st, l00 + (size_t)s * k.luma_bytes, c00 + (size_t)s * k.cplane_bytes, k.rs, k.w, k.h,
                           d_out + (size_t)s * (k.w * k.h * 3 / 2));
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

extern "C" int x264gpu_encoder_get_recon(x264gpu_encoder *e, int stream_idx, uint8_t *d_out, void *stream)
{
    ARG_TRY(e && d_out && stream_idx >= 0 && stream_idx < e->cfg.streams && e->have > 0);
    const EncK &k = e->k;
    const int slot = e->last;
    const uint8_t *l = e->luma[slot] + (size_t)stream_idx * k.luma_bytes + (size_t)PAD * k.rs + PAD;
    const uint8_t *c = e->chroma[slot] + (size_t)stream_idx * k.cplane_bytes + (size_t)CPAD * k.rs + 2 * CPAD;
    hipLaunchKernelGGL(k_get_recon, dim3((k.w + 255) / 256, k.h), dim3(256), 0, (hipStream_t)stream, l, c, k.rs, k.w, k.h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

extern "C" int x264gpu_encoder_get_recon_slot(x264gpu_encoder *e, int stream_idx, int slot, uint8_t *d_out, void *stream)
{
    ARG_TRY(e && d_out && stream_idx >= 0 && stream_idx < e->cfg.streams && slot >= 0 && slot < e->slots);
    const EncK &k = e->k;
    const uint8_t *l = e->luma[slot] + (size_t)stream_idx * k.luma_bytes + (size_t)PAD * k.rs + PAD;
    const uint8_t *c = e->chroma[slot] + (size_t)stream_idx * k.cplane_bytes + (size_t)CPAD * k.rs + 2 * CPAD;
    hipLaunchKernelGGL(k_get_recon, dim3((k.w + 255) / 256, k.h), dim3(256), 0, (hipStream_t)stream, l, c, k.rs, k.w, k.h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}
