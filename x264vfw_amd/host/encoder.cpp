// encoder.cpp — x264_encoder_* API (boundary B1) over the MI355X hot path (boundary B3, libx264gpu.so).
//   x264_encoder_open        codec.c:1623    x264_encoder_parameters  codec.c:1630
//   x264_encoder_headers     codec.c:1650    x264_encoder_encode      codec.c:1693
//   x264_encoder_delayed_frames codec.c:1848 x264_encoder_close       codec.c:1857
// Contracts kept: NULL / negative on failure, diagnostics only through pf_log, all NALs of a
// call contiguous from nal[0].p_payload, buffers valid until the next call, param strings copied at open,
// pic_out->{i_type,b_keyframe,i_pts,i_dts} filled.  No CPU fallback: open fails without a GPU.
#include "host.hpp"
#include "dpb.hpp"
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <deque>
#include <memory>
#include <string>
#include <atomic>
#include <thread>
#include <chrono>
#include <mutex>
#include <condition_variable>

using namespace x264host;

struct x264_t {
    x264_param_t param;
    std::string stat_in, stat_out;
    x264gpu_encoder *gpu = nullptr;
    int mbw = 0, mbh = 0, nmb = 0;
    int qp_i = 23, qp_p = 23, pic_init_qp = 26;
    int keyint = 250;
    int log2_max_frame_num = 8;
    int level_idc = 40, profile_idc = 66;
    int frame_no = 0;            // frames submitted
    int frames_since_idr = 0;
    int frame_num = 0;
    int idr_pic_id = 0;
    int sei_sent = 0;
    uint8_t *d_in = nullptr;
    x264gpu_mb *d_mb = nullptr;
    int16_t *d_lv = nullptr;
    std::vector<uint8_t> h_in;
    std::vector<x264gpu_mb> h_mb;
    std::vector<int16_t> h_lv;
    std::vector<uint8_t> out;            // bitstream of the current call
    std::vector<x264_nal_t> nals;
    std::vector<size_t> nal_off;
    SliceStats last_stats = { 0 };
    // ---- lookahead-driven decisions (threads 1 only): scenecut and CRF, both fed by x264gpu_lookahead_frame_cost ----
    x264gpu_lookahead *la = nullptr;
    int32_t *d_la = nullptr;             // device: the four sums of the last picture
    int keyint_min = 25;
    bool crf = false;
    struct {                             // [x264-upstream] encoder/ratecontrol.c, the CRF branch of rate_estimate_qscale (restated from memory)
        double rate_factor_constant = 1, qcompress = 0.6, ip_factor = 1.4, ip_offset = 0, dur_ratio = 1;
        double cplxsum = 0, cplxcount = 0, accum_p_qp = 0, accum_p_norm = 0, lmin = 0, lmax = 0;
        double last_qscale_for[2] = { 0, 0 };       // [0] I, [1] P
        int last_non_b_is_i = 1;
        // single-pass ABR (--bitrate): the 1-pass branch of rate_estimate_qscale + the bookkeeping of x264_ratecontrol_end
        double bitrate = 0, fps = 25, cplxr_sum = 0, wanted_bits_window = 0, abr_buffer = 0, total_bits = 0, last_rceq = 1, lstep = 1.3195;
        double qpa_last = 0;                        // quantiser of the picture whose size arrives next (ratecontrol_end)
    } rc;
    bool abr = false;
    double t_b[5] = { 0, 0, 0, 0, 0 };          // ... sessions on the DPB model: slice-type analysis, GPU hot path, download, entropy coding, pictures
    double t_phase[6] = { 0, 0, 0, 0, 0, 0 };   // X264GPU_HOST_TIMING=1: seconds in copy-in, upload + lookahead, GPU, download, entropy coding, calls
    // ---- lookahead queue (threads 1): pictures wait here rc-lookahead deep when the macroblock-tree needs to see what follows them ----
    struct QEntry { int64_t pts; int slot; int type; int scenecut; int32_t costs[4]; x264_image_t img; int qp; int buf; bool launched; float qpm = 0.f; };      // type: 0 P, 1 I, 2 IDR; qpm: the float quantiser handed to the device (0: none); qp / buf / launched: set by gpu_stage
    std::deque<QEntry> queue;
    int L = 0, Q = 1;                    // pictures held back; ring slots (L + 1)
    std::vector<uint8_t *> q_raw;        // device: source pictures (slot 0 is d_in when nothing is held back: zero-copy input)
    std::vector<int32_t *> q_info;       // device: lookahead block records per slot
    void *q_block[4] = { nullptr, nullptr, nullptr, nullptr };      // device: the blocks the queue's per-slot arrays are cut from (raw pictures, block records, AQ offsets, tree offsets)
    std::vector<float *> q_aq;           // device: AQ offsets per slot (x264 f_qp_offset_aq, single floats)
    float *d_tree = nullptr;             // device: macroblock-tree quantiser offsets of the picture being coded
    long la_count = 0; int la_gop = 0;   // pictures seen by the lookahead; distance from the last IDR at lookahead time
    bool mbtree = false; float aq_strength = 0.f, tree_strength = 0.f;      // x264_adaptive_quant_frame's strength (mode 1: aq-strength * 1.0397f; 2 / 3: aq-strength), macroblock_tree_finish's 5.0f * (1.0f - qcomp)
    int aq_mode = 0;                     // --aq-mode (1 variance, 2 auto-variance, 3 auto-variance biased); modes 2 / 3 always arrive as offsets computed when the picture comes in
    int cavlc_threads = 1;               // row bands of a slice coded in parallel (threads 1 sessions; GOP-parallel ones use a thread per GOP)
    // ---- pipelined threads-1 sessions (CRF with pictures held back anyway): the GPU stage of picture n+1 runs in a helper thread while
    //      the calling thread entropy-codes picture n; every picture is handed back one call later than without it ----
    bool pipeline = false;
    std::thread gpu_thread;
    int gpu_rc = 0;                      // result of the GPU stage in flight
    int rc_frames = 0;                   // pictures that went through rate control (frames_done of rc_pick_qp)
    int device = 0;
    std::vector<float> gop_qpm;          // ... and its float quantiser (x264 rc->qpm)
    std::vector<int8_t> gop_qp;          // GOP-parallel CRF: the quantiser of every ring picture (slot * keyint + position), decided on arrival
    struct Zone { int start, end; bool force_qp; int qp; float bitrate_factor; };      // x264_zone_t: pictures start..end (display order) at quantiser qp, or at bitrate_factor times their bits
    std::vector<Zone> zones;
    float last_qpm = 0.f;                // ... and its float quantiser as the device got it (x264 rc->qpm; 0 = the integer one)
    int last_qp = 0, last_scenecut = 0;  // diagnostics: quantiser and scenecut flag of the last coded picture
    int32_t last_costs[4] = { 0, 0, 0, 0 };
    // ---- GOP-parallel mode (--threads G > 1): G closed GOPs of the one stream are coded in lock-step on G stream slots of the
    //      GPU encoder; frames come out in order, (G-1)*keyint calls late.  Fixed keyint + CQP make the GOPs independent, so
    //      the bytes equal the serial encode's (tests/test_gpu_host.py::test_gop_parallel_equals_serial).
    int G = 1;
    int slices = 1;               // x264 slice threads: slices per picture (own wavefront + own NAL each)
    int slices_plain = 0;         // ... or x264 --slices N: the same split, filtered across the boundaries, up to one slice per macroblock row
    // The G slots are dealt to the visible devices (slot s -> device s % D, its local slot s / D): every device runs its slots in lock-step
    // with its own encoder, ring and download buffers, issued by one host thread per device; closed GOPs are independent, so there is no
    // exchange between devices and the frames still leave in stream order (north star: "frames of one stream shard one-per-GPU").
    struct DevCtx {
        int dev = 0, nsl = 0, base = 0;  // device ordinal; slots it owns; its first row in the host download buffers
        x264gpu_encoder *gpu = nullptr;
        uint8_t *d_ring = nullptr;       // [keyint positions][nsl slots] tight I420 pictures of the batch being gathered
        x264gpu_mb *d_mb = nullptr; int16_t *d_lv = nullptr;
    };
    std::vector<DevCtx> devs;            // GOP-parallel mode only (threads 1 sessions use gpu / d_mb / d_lv below on the caller's device)
    long submitted = 0, emitted = 0;     // frames in / out
    int next_pos = 0;                    // first position of the current batch not yet coded
    bool flushed = false;                // the partly gathered batch has been coded (flush calls only drain after that)
    bool failed = false;                 // GOP-parallel mode: a GPU call failed; the session only returns errors from now on
    std::string gpu_err;                 // threads 1, pipelined: the helper thread's x264gpu_last_error() text (that buffer is thread-local)
    struct Coded { std::vector<uint8_t> bytes; std::vector<size_t> off; std::vector<int> types; int idr; int ref_idc = -1, i_type = 0, disp = -1; };      // (ref_idc / i_type / disp: GOP slots with B pictures — a slot index is a CODING position there)
    std::deque<Coded> ready;             // coded frames [emitted, emitted + ready.size())
    std::deque<int64_t> pts;             // pts of frames not yet emitted
    std::vector<Coded> slotbuf;          // G x keyint frames of the batch being coded (index slot * keyint + pos)
    std::vector<uint8_t> slot_have;      // which of them are coded AND joined (written by the calling thread only)
    int pool_t = -1, pool_nslots = 0, pool_slot0 = 0;    // position / slot count / first slot the running CAVLC threads are coding
    // ---- GOP slots with B pictures (--threads G --bframes N under a constant quantiser): every slot is a closed GOP on the DPB model; the slots run the
    //      same plan in lock-step (--b-adapt 0, no scenecut: picture c of the coding order has the same type, lists and marking in every GOP), a
    //      mini-GOP is coded once the batch's last GOP has delivered its closing picture; the stream's last, shorter GOP is coded alone at the flush.
    //      Frames leave in CODING order with x264's pts / dts, byte-identical to the threads-1 session (tests/test_shard_cpu.py).
    bool gopb = false;
    std::vector<std::pair<int, int>> gorder;      // coding order of a full GOP: (display index in the GOP, PIC_*)
    int gb_next = 0;                              // next coding position of the batch being gathered
    struct GopDpb { Dpb dpb; int l0ref0poc[8] = { 0 }; } gdpb;
    std::vector<std::thread> pool;       // CAVLC threads of the position coded last: they overlap the GPU work of the next one
    std::vector<x264gpu_mb> h_mb2;       // second download buffers (the pool reads one pair while the next position lands in the other)
    std::vector<int16_t> h_lv2;
    int dl = 0;                          // download buffer in use for the NEXT position
    // ---- sessions with B pictures (threads 1; x264 --bframes N --b-pyramid): pictures wait in display order until the mini-GOP they belong to
    //      is closed by a P / I picture (x264_slicetype_decide with --b-adapt 0: N B pictures between non-B pictures, fewer in front of a keyframe
    //      or at the end), then leave in coding order: the closing picture, the B-reference of the run, the other B pictures.  The DPB, the
    //      reference lists and the slice header's share of them come from host/dpb.hpp ----
    int bframes = 0, bpyramid = 0, log2_max_poc_lsb = 0;
    // dpbmode: the session runs on the DPB model (host/dpb.hpp) and x264gpu_encode_pictures — every session with B pictures, and sessions
    // without them that use --weightp 2 (whose duplicate references need explicit lists); weightp: the effective --weightp (0 or 2)
    bool dpbmode = false; int weightp = 0;
    bool weightp_fake = false;           // x264 X264_WEIGHTP_FAKE: --weightp 0 with macroblock-tree and psy: the lookahead still looks for fades, for the tree's sake alone
    Dpb dpb;
    struct BEntry { int64_t pts; int frame; int slot; int forced; int scenecut; int32_t costs[4]; x264_image_t img;      // forced: 0 auto, 1 I, 2 IDR
                    int type = 0; int b_scenecut = 1;         // slicetype analysis: the type decided so far (ST_*), "may still be a real scene cut"
                    Dpb::LumaWeight w;                        // x264_weights_analyse's luma weight of reference 0 when the picture is coded as P (--weightp)
                    float weighted_cost_delta[18] = { 0 }; }; // f_weighted_cost_delta[distance - 1]: weighted / unweighted cost where the fake analysis found a luma weight
    // x264's lookahead in its own structure (x264_slicetype_analyse: scenecut against the last non-B picture with flash detection, --b-adapt 1)
    // on the device's frame costs of arbitrary (p0, p1, b) triples; the half-resolution planes of a queued picture live in the slicetype object's
    // slot of the same number as its raw picture
    x264gpu_slicetype *st = nullptr;
    bool have_last_nonb = false; BEntry last_nonb;
    int last_keyframe = 0;                // display index of the last IDR picture decided (x264 h->lookahead->i_last_keyframe)
    int badapt = 0;
    std::vector<float *> q_tree;         // device, per queue slot: the quantiser offsets the macroblock-tree left with the picture (AQ offsets until it ran)
    bool st_aq_costs = false;            // AQ session without macroblock-tree on the DPB model: the rate control reads the AQ-weighted frame costs (i_cost_est_aq)
    int st_wait = 0;                     // pictures the lookahead holds before a decision (x264 i_slicetype_length: max(bframes, rc-lookahead under mbtree))
    // cross-session batcher (X264GPU_BATCH=N): N sessions of equal geometry and toolset share ONE device encoder with N streams; the pictures
    // they submit are coded in one lock-step launch, every session entropy-codes its own stream on its caller's thread
    struct BatchGroup *batch = nullptr; int batch_idx = -1, batch_n = 0;
    void *up_stream = nullptr;           // a batch session's upload stream (one of the group's, async groups): its pictures go up while the group's round runs on the compute stream
    // batch sessions with overlap (BatchGroup::overlap): the picture just submitted is downloaded and entropy-coded by a helper thread while the group's next round runs;
    // its NAL units leave with the NEXT call (one picture of delay).  Two slots used in turn: the one being filled, the one waiting to be handed out
    struct Deferred { std::thread th; bool valid = false; std::atomic<bool> hurry{ false }; std::string err; std::vector<uint8_t> out; std::vector<size_t> off; std::vector<int> types; int nal_ref_idc = 0;
                      std::vector<x264gpu_mb> mb; std::vector<x264gpu_level_index> ix; std::unique_ptr<int16_t[]> lv; SliceStats stats = { 0 };      // (lv: never cleared — what is downloaded is what is read)
                      int i_type = 0, b_keyframe = 0; int64_t pts = 0, dts = 0; x264_image_t img;
                      int qp = 0, scenecut = 0; float qpm = 0.f; int32_t costs[4] = { 0, 0, 0, 0 }; };      // the decision hooks' values of THIS picture (x264host_last_decision / _last_qpm)
    Deferred defer[2]; int defer_cur = 0;
    struct BPlanned { BEntry e; int type; };                                                                            // type: PIC_*
    // ---- several pictures of ONE session in flight (threads-1 sessions on the DPB model; x264's frame threads overlap pictures too).  The b pictures of a mini-GOP, the
    //      B reference between two finished P pictures and the next P picture share only FINISHED references: each is issued through a launch context of its own (the
    //      encoder or a view of it: own scratch, the shared DPB) on a stream of its own, behind the events of the pictures it references, into a slot no picture in
    //      flight reads or writes; pictures are planned, issued and handed back in coding order, so the stream is the serial session's byte for byte ----
    struct LaunchCtx { x264gpu_encoder *gpu = nullptr; void *stream = nullptr, *ev = nullptr; x264gpu_mb *d_mb = nullptr; int16_t *d_lv = nullptr; bool busy = false; };
    struct Inflight { BPlanned pl; x264gpu_pic pic; SliceParams sp; int ctx = 0, nal_ref_idc = 0; unsigned slots_used = 0; char direct_char = '-'; };
    void *ev_la = nullptr;               // the default stream's position when a picture is issued: its upload, offsets and lowres vectors are complete behind it
    std::vector<LaunchCtx> lctx; std::deque<Inflight> fl; int inflight = 1; int slot_writer[8] = { -1, -1, -1, -1, -1, -1, -1, -1 }; int last_retired_slot = -1;
    std::deque<BEntry> bq;
    std::deque<BPlanned> bcoding;
    std::vector<int64_t> all_pts;        // every pts seen, in display order (the dts delay line)
    long coded_count = 0;
    double slot_qp_rc[8] = { 0 };        // CRF: the quantiser (float) every kept picture was given, by DPB slot (x264 f_qp_avg_rc)
    int slot_ptype[8] = { 0 };           // ... and its picture type
    // --direct temporal / auto (x264 h->stat.i_direct_score, frame->i_poc_l0ref0): 1 spatial, 2 temporal, 3 auto; the running skip-probe counts of
    // temporal [0] / spatial [1] prediction; the POC behind reference 0 of list 0 of every kept picture (INT_MIN: it had none)
    int direct_mode = 1, direct_score[2] = { 0, 0 }, slot_l0ref0poc[8] = { 0 };
    // 2-pass (x264 ratecontrol.c; the driver's encoding type 4, codec.c:1516-1541): pass 1 appends one line per coded picture to the statistics file;
    // pass 2 reads them — picture types, bits split into texture / vectors / the rest, the quantiser they were coded at — and spreads the requested
    // size over the pictures (init_pass2), then follows the plan with feedback (rate_estimate_qscale's 2-pass branch)
    struct Pass2Entry { char type = 'P'; int in = 0, out = 0, icount = 0, kept_as_ref = 1; double qp = 0, qscale = 0, new_qscale = 0, blurred = 0, expected_bits = 0, dur = 1;
                        long tex = 0, mv = 0, misc = 0; };
    bool pass1 = false, pass2 = false;
    FILE *stat_file = nullptr;
    std::vector<Pass2Entry> p2;                       // by display index ("in:")
    std::vector<int> p2_out;                          // coding order -> display index
    double p2_expected_sum = 0, p2_total_bits = 0, p2_final_bits = 0, p2_abr_buffer = 0;
    char last_direct_char = '-';
};

// ---- cross-session batcher ------------------------------------------------------------------------------------------------------
// The reference opens one CODEC / x264_t per stream (driverproc.c:110-128); the device is fast only when many streams are coded in lock-step
// (x264gpu_config.streams).  With X264GPU_BATCH=N in the environment, the first N sessions opened with the same geometry and toolset (and
// a fixed picture structure: no scenecut / b-adapt / mbtree, so that picture k has the same type in all of them) form a group around one
// device encoder with N streams.  A session's x264_encoder_encode hands its picture to the group and waits; the call that completes the
// round launches the hot path for all streams; every caller then downloads its own records and entropy-codes its own stream on its own
// thread.  Each stream is coded exactly as a session of its own would code it (streams never interact): the bytes are the same.
struct BatchGroup {
    std::mutex m; std::condition_variable cv;
    x264gpu_config cfg; int N = 0, device = 0;
    x264gpu_encoder *gpu = nullptr; uint8_t *d_in = nullptr; x264gpu_mb *d_mb = nullptr; int16_t *d_lv = nullptr;
    // overlap: the records / levels of round k are downloaded and entropy-coded (by a helper thread of every session) WHILE round k + 1 runs: a second pair of
    // output buffers used in turn, a stream of the group's own for the downloads; every session hands its pictures back one call later
    bool overlap = false; x264gpu_mb *d_mb2 = nullptr; int16_t *d_lv2 = nullptr; void *dl_stream = nullptr;
    // ... and (async) the callers do not wait for the round either: it is QUEUED on the group's own compute stream behind the round before, an event behind it tells the
    // download of its results when it is done; the callers go on to copy in and upload their next pictures (on upload streams of their own) while the device works
    bool async = false; void *cs = nullptr, *ev[2] = { nullptr, nullptr };
    // the levels leave the device packed (x264gpu_pack_levels behind every round, in place): a member downloads its records, its index and the part of its levels that is kept
    // (~10 % at medium; dense, 2048 members x 7 MB a round were what the first rounds waited for: fresh pages of the download buffers, 15 GB a round over the link)
    bool pack = false; x264gpu_level_index *d_ix = nullptr, *d_ix2 = nullptr;
    std::vector<void *> up_streams;          // the members' uploads: a handful of streams dealt round-robin (a stream per member was 0.7 ms to create and 0.6 ms to destroy, x 2048, serialised in the runtime)
    long ev_round[2] = { 0, 0 }, ev_done[2] = { 0, 0 }; bool ev_waiting[2] = { false, false }; std::string ev_err;      // per buffer pair: the round recorded behind it (1-based), the last one known complete, a member is waiting for the event
    long launched = 0;            // rounds whose kernels have been issued: the helper threads start entropy coding round k once round k + 1 is on the device (or when asked to hurry),
                                  // so that the host cores are the callers' while the next pictures are uploaded and submitted
    bool running = false;         // a round is being waited for with the group's lock released (overlap): nobody starts another
    int leaving = 0; bool orphaned = false;      // batch_leave: leavers between "decided to run the round" and "ran it"; the last member left meanwhile (the leaver destroys the group)
    size_t insz = 0, nmb = 0;
    std::vector<char> member, arrived; int joined = 0, active = 0, n_arrived = 0;
    std::vector<x264gpu_pic> pics; long round = 0; int round_rc = 0; std::string err;
    bool closed = false;          // a member left: no more joiners (batch_leave)
    // X264GPU_BATCH_TIMING=1: where the members' threads spent their time, summed over members (printed when the group goes): waiting for a round's event,
    // downloading records / levels, waiting for the next round's launch before the slices are written, writing them, waiting in batch_submit for the round to fill
    std::atomic<long> t_us[6] = {};
    std::atomic<long> r_dl[64] = {}, r_sl[64] = {}, r_dl_first[64] = {}, r_dl_last[64] = {};          // per round: download / slice seconds summed over the members; when the first / last download ended
    std::vector<long> tl_launch, tl_done, tl_host;          // ... and per round: issued, its event seen, its results on the host (microseconds; the first launch = 0)
    bool timing = getenv("X264GPU_BATCH_TIMING") != nullptr;
};
static inline long us_now() { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::mutex g_batch_mu;
static std::vector<BatchGroup *> g_batch_groups;

static void batch_destroy(BatchGroup *g)
{
    const long t_destroy0 = g->timing ? us_now() : 0;
    struct Tm { BatchGroup *g; long t0; bool on; ~Tm() { if (on) fprintf(stderr, "x264gpu batch: the group's device memory and streams took %.2f s to release\n", (us_now() - t0) / 1e6); } } tm{ g, t_destroy0, g->timing };
    if (g->timing)
        fprintf(stderr, "x264gpu batch timing, seconds summed over %d members: event wait %.1f, download %.1f, wait for the next launch %.1f, slices %.1f, submit wait %.1f, join of the helper %.1f\n", g->N,
                g->t_us[0] / 1e6, g->t_us[1] / 1e6, g->t_us[2] / 1e6, g->t_us[3] / 1e6, g->t_us[4] / 1e6, g->t_us[5] / 1e6);
    if (g->timing && !g->tl_launch.empty()) {
        fprintf(stderr, "x264gpu batch rounds (s from the first launch): issued / event seen / results on the host:");
        for (size_t i = 0; i < g->tl_launch.size(); i++)
            fprintf(stderr, "  %zu: %.2f / %.2f / %.2f", i, (g->tl_launch[i] - g->tl_launch[0]) / 1e6, i < g->tl_done.size() ? (g->tl_done[i] - g->tl_launch[0]) / 1e6 : -1., i < g->tl_host.size() ? (g->tl_host[i] - g->tl_launch[0]) / 1e6 : -1.);
        fprintf(stderr, "\n");
        fprintf(stderr, "x264gpu batch rounds: download s per member / slices s per member / first .. last download done (s from the first launch):");
        for (size_t i = 0; i < g->tl_launch.size() && i < 64; i++)
            fprintf(stderr, "  %zu: %.2f / %.2f / %.2f .. %.2f", i, g->r_dl[i] / 1e6 / g->N, g->r_sl[i] / 1e6 / g->N, (g->r_dl_first[i] - g->tl_launch[0]) / 1e6, (g->r_dl_last[i] - g->tl_launch[0]) / 1e6);
        fprintf(stderr, "\n");
    }
    if (g->gpu) x264gpu_encoder_destroy(g->gpu);
    if (g->d_in) x264gpu_free(g->d_in);
    if (g->d_mb) x264gpu_free(g->d_mb);
    if (g->d_lv) x264gpu_free(g->d_lv);
    if (g->d_mb2) x264gpu_free(g->d_mb2);
    if (g->d_lv2) x264gpu_free(g->d_lv2);
    if (g->d_ix) x264gpu_free(g->d_ix);
    if (g->d_ix2) x264gpu_free(g->d_ix2);
    if (g->cs) x264gpu_stream_sync(g->cs);
    if (g->dl_stream) x264gpu_stream_destroy(g->dl_stream);
    for (void *st : g->up_streams) { x264gpu_stream_sync(st); x264gpu_stream_destroy(st); }
    for (int i = 0; i < 2; i++) if (g->ev[i]) x264gpu_event_destroy(g->ev[i]);
    if (g->cs) x264gpu_stream_destroy(g->cs);
    delete g;
}
// -> the group and the stream index of the caller, or nullptr (setup failed: last error set)
static BatchGroup *batch_join(const x264gpu_config &cfg1, int N, size_t insz, size_t nmb, int *idx)
{
    std::lock_guard<std::mutex> lk(g_batch_mu);
    int dev = 0;
    (void)x264gpu_get_device(&dev);
    for (BatchGroup *g : g_batch_groups) {
        x264gpu_config a = g->cfg, b = cfg1;
        a.streams = b.streams = 0;
        if (g->N == N && g->device == dev && g->joined < N && !g->closed && !memcmp(&a, &b, sizeof(a))) {
            std::lock_guard<std::mutex> lg(g->m);
            *idx = g->joined++; g->active++; g->member[(size_t)*idx] = 1;
            g->cv.notify_all();
            return g;
        }
    }
    BatchGroup *g = new BatchGroup();
    g->cfg = cfg1; g->cfg.streams = N; g->N = N; g->device = dev; g->insz = insz; g->nmb = nmb;
    g->member.assign((size_t)N, 0); g->arrived.assign((size_t)N, 0); g->pics.resize((size_t)N);
    if (x264gpu_encoder_create(&g->gpu, &g->cfg) != X264GPU_OK ||
        x264gpu_malloc((void **)&g->d_in, (size_t)N * insz) != X264GPU_OK ||
        x264gpu_malloc((void **)&g->d_mb, (size_t)N * nmb * sizeof(x264gpu_mb)) != X264GPU_OK ||
        x264gpu_malloc((void **)&g->d_lv, (size_t)N * nmb * X264GPU_MB_LEVELS * sizeof(int16_t)) != X264GPU_OK) { batch_destroy(g); return nullptr; }
    {
        const char *oe = getenv("X264GPU_BATCH_OVERLAP");
        if (!(oe && oe[0] == '0') && !getenv("X264GPU_DUMP_RECORDS") &&
            x264gpu_malloc((void **)&g->d_mb2, (size_t)N * nmb * sizeof(x264gpu_mb)) == X264GPU_OK &&
            x264gpu_malloc((void **)&g->d_lv2, (size_t)N * nmb * X264GPU_MB_LEVELS * sizeof(int16_t)) == X264GPU_OK && x264gpu_stream_create(&g->dl_stream) == X264GPU_OK) g->overlap = true;
        const char *ae = getenv("X264GPU_BATCH_ASYNC");
        if (g->overlap && !(ae && ae[0] == '0') && x264gpu_stream_create(&g->cs) == X264GPU_OK && x264gpu_event_create(&g->ev[0]) == X264GPU_OK && x264gpu_event_create(&g->ev[1]) == X264GPU_OK) g->async = true;
        if (g->async && !getenv("X264GPU_BATCH_DENSE") && x264gpu_malloc((void **)&g->d_ix, (size_t)N * nmb * sizeof(x264gpu_level_index)) == X264GPU_OK &&
            x264gpu_malloc((void **)&g->d_ix2, (size_t)N * nmb * sizeof(x264gpu_level_index)) == X264GPU_OK) g->pack = true;
        if (g->async) for (int i = 0; i < 16 && i < N; i++) { void *st = nullptr; if (x264gpu_stream_create(&st) == X264GPU_OK) g->up_streams.push_back(st); else break; }
    }
    g->joined = 1; g->active = 1; g->member[0] = 1; *idx = 0;
    g_batch_groups.push_back(g);
    return g;
}
// the launch of a complete round; g->m is held through lk (released while an overlapping group waits for its kernels)
static void batch_run_round(BatchGroup *g, std::unique_lock<std::mutex> &lk)
{
    int first = -1;
    for (int s = 0; s < g->N; s++) if (g->arrived[(size_t)s]) { first = s; break; }
    g->round_rc = 0; g->err.clear();
    if (first >= 0) {
        for (int s = 0; s < g->N; s++) {
            if (!g->arrived[(size_t)s]) { g->pics[(size_t)s] = g->pics[(size_t)first]; continue; }      // a stream whose session has gone: coded along, thrown away
            const x264gpu_pic &a = g->pics[(size_t)s], &b = g->pics[(size_t)first];
            if (a.slice_type != b.slice_type || a.poc != b.poc || a.dst != b.dst || a.keep != b.keep || a.nref[0] != b.nref[0] || a.nref[1] != b.nref[1] ||
                memcmp(a.slot, b.slot, sizeof(a.slot)) || a.blind_dupe != b.blind_dupe) { g->round_rc = -1; g->err = "the sessions of a batch must submit pictures of the same structure (same picture count, keyint, bframes, forced types)"; }
        }
        const bool second = g->overlap && (g->round & 1);          // the output buffers of this round (the other pair may still be downloading)
        if (!g->round_rc && x264gpu_encode_pictures(g->gpu, g->d_in, g->pics.data(), second ? g->d_mb2 : g->d_mb, second ? g->d_lv2 : g->d_lv, g->async ? g->cs : nullptr) != X264GPU_OK) { g->round_rc = -1; g->err = x264gpu_last_error(); }
        if (!g->round_rc && g->pack && x264gpu_pack_levels(second ? g->d_lv2 : g->d_lv, g->N, (int)g->nmb, second ? g->d_ix2 : g->d_ix, nullptr, g->cs) != X264GPU_OK) { g->round_rc = -1; g->err = x264gpu_last_error(); }
        if (!g->round_rc && g->async) {
            // queued, not awaited: the event behind the round is what its downloads wait for (batch_download)
            if (x264gpu_event_record(g->ev[second ? 1 : 0], g->cs) != X264GPU_OK) { g->round_rc = -1; g->err = x264gpu_last_error(); }
            g->launched++;
            if (g->timing) g->tl_launch.push_back(us_now());
            g->ev_round[second ? 1 : 0] = g->round + 1;
        } else
        // overlap: the downloads run on the group's own stream, which does not wait for the default one: the round must be complete before anyone is told
        if (!g->round_rc && g->overlap) {
            g->launched++; g->running = true;
            g->cv.notify_all();          // the helper threads of the round before: the device is busy again, the host cores are theirs
            lk.unlock();
            const bool ok = x264gpu_stream_sync(nullptr) == X264GPU_OK;
            std::string e = ok ? std::string() : std::string(x264gpu_last_error());
            lk.lock();
            g->running = false;
            if (!ok) { g->round_rc = -1; g->err = e; }
        }
    }
    g->round++; g->n_arrived = 0;
    std::fill(g->arrived.begin(), g->arrived.end(), 0);
    g->cv.notify_all();
}
// hands picture `pic` of stream s to the group and waits for the round that codes it; *buf = which pair of output buffers holds the round's results
static int batch_submit(BatchGroup *g, int s, const uint8_t *d_src, const x264gpu_pic &pic, int *buf, std::string &err)
{
    // (async: on the group's compute stream, i.e. behind the round before — which may still be reading d_in — and in front of this round's launch)
    if (x264gpu_memcpy_d2d(g->d_in + (size_t)s * g->insz, d_src, g->insz, g->async ? g->cs : nullptr) != X264GPU_OK) { err = x264gpu_last_error(); return -1; }
    std::unique_lock<std::mutex> lk(g->m);
    // every member must have been opened before the first picture is coded: a late joiner would be a picture behind for good
    if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->joined == g->N; })) { err = "X264GPU_BATCH: fewer sessions were opened than the batch size"; return -1; }
    g->pics[(size_t)s] = pic; g->arrived[(size_t)s] = 1; g->n_arrived++;
    const long my_round = g->round;
    *buf = g->overlap ? (int)(my_round & 1) : 0;
    const long t0 = g->timing ? us_now() : 0;
    struct Acc { BatchGroup *g; long t0; ~Acc() { if (g->timing) g->t_us[4] += us_now() - t0; } } acc{ g, t0 };
    if (g->n_arrived >= g->active && !g->running) batch_run_round(g, lk);
    else if (!g->cv.wait_for(lk, std::chrono::seconds(600), [&] { return g->round != my_round; })) {
        // a member neither submitted its picture nor closed: give up on this session (the others keep waiting for it, or for its close)
        g->arrived[(size_t)s] = 0; g->n_arrived--;
        err = "X264GPU_BATCH: another session of the batch stopped submitting pictures";
        return -1;
    }
    if (g->round_rc) { err = g->err; return -1; }
    return 0;
}
// stream s' records and levels of the round whose results lie in buffer pair `buf`
// (h_ix: the group packs its levels — the member's index; h_lv then receives only the kept groups)
static int batch_download(BatchGroup *g, int s, int buf, x264gpu_mb *h_mb, int16_t *h_lv, std::string &err, x264gpu_level_index *h_ix = nullptr)
{
    const x264gpu_mb *dm = buf ? g->d_mb2 : g->d_mb; const int16_t *dl = buf ? g->d_lv2 : g->d_lv;
    // (async groups: the members' downloads are dealt to the group's sixteen upload / download streams)
    void *st = g->overlap ? (g->up_streams.empty() ? g->dl_stream : g->up_streams[(size_t)s % g->up_streams.size()]) : nullptr;
    const long t0 = g->timing ? us_now() : 0;
    if (g->async) {
        // ONE thread waits for the round's event, the other members sleep on the group's condition variable (2048 helper threads in hipEventSynchronize would
        // take the host's cores from the callers that are copying in and uploading the next pictures)
        std::unique_lock<std::mutex> lk(g->m);
        const long want = g->ev_round[buf ? 1 : 0];          // the round recorded behind this buffer pair (it cannot be re-recorded before every member has this round's results)
        while (g->ev_done[buf ? 1 : 0] < want) {
            if (!g->ev_waiting[buf ? 1 : 0]) {
                g->ev_waiting[buf ? 1 : 0] = true;
                lk.unlock();
                const bool ok = x264gpu_event_sync(g->ev[buf ? 1 : 0]) == X264GPU_OK;
                const long t_ev = g->timing ? us_now() : 0;
                const std::string e = ok ? std::string() : std::string(x264gpu_last_error());
                lk.lock();
                if (g->timing) { g->tl_done.push_back(t_ev); g->tl_host.push_back(us_now()); }
                g->ev_waiting[buf ? 1 : 0] = false;
                if (!ok) { g->ev_err = e; g->ev_done[buf ? 1 : 0] = want; g->cv.notify_all(); err = e; return -1; }
                g->ev_done[buf ? 1 : 0] = want;
                g->cv.notify_all();
            } else g->cv.wait(lk);
        }
        if (!g->ev_err.empty()) { err = g->ev_err; return -1; }
    }
    const long t1 = g->timing ? us_now() : 0;
    const long rnd = g->async ? g->ev_round[buf ? 1 : 0] - 1 : 0;
    struct Acc { BatchGroup *g; long t0, t1, rnd; ~Acc() { if (g->timing) { const long t2 = us_now(); g->t_us[0] += t1 - t0; g->t_us[1] += t2 - t1;
        if (rnd >= 0 && rnd < 64) { g->r_dl[rnd] += t2 - t1; long f = g->r_dl_first[rnd].load(); while ((f == 0 || t2 < f) && !g->r_dl_first[rnd].compare_exchange_weak(f, t2)) {} long l = g->r_dl_last[rnd].load(); while (t2 > l && !g->r_dl_last[rnd].compare_exchange_weak(l, t2)) {} } } } } acc{ g, t0, t1, rnd };
    if (g->pack) {
        if (!h_ix) { err = "X264GPU_BATCH: the group's levels are packed"; return -1; }
        const x264gpu_level_index *di = (buf ? g->d_ix2 : g->d_ix) + (size_t)s * g->nmb;
        if (x264gpu_memcpy_d2h(h_mb, dm + (size_t)s * g->nmb, g->nmb * sizeof(x264gpu_mb), st) != X264GPU_OK ||
            x264gpu_memcpy_d2h(h_ix, di, g->nmb * sizeof(x264gpu_level_index), st) != X264GPU_OK) { err = x264gpu_last_error(); return -1; }
        const size_t kept = ((size_t)h_ix[g->nmb - 1].at + (size_t)__builtin_popcount(h_ix[g->nmb - 1].groups)) * 16;          // levels
        if (kept > g->nmb * (size_t)X264GPU_MB_LEVELS) { err = "X264GPU_BATCH: level index out of range"; return -1; }
        if (kept && x264gpu_memcpy_d2h(h_lv, dl + (size_t)s * g->nmb * X264GPU_MB_LEVELS, kept * sizeof(int16_t), st) != X264GPU_OK) { err = x264gpu_last_error(); return -1; }
        return 0;
    }
    if (x264gpu_memcpy_d2h(h_mb, dm + (size_t)s * g->nmb, g->nmb * sizeof(x264gpu_mb), st) != X264GPU_OK ||
        x264gpu_memcpy_d2h(h_lv, dl + (size_t)s * g->nmb * X264GPU_MB_LEVELS, g->nmb * X264GPU_MB_LEVELS * sizeof(int16_t), st) != X264GPU_OK) { err = x264gpu_last_error(); return -1; }
    return 0;
}
static void batch_leave(BatchGroup *g, int s)
{
    // the registry lock first (the order batch_join takes them in): "last" is decided and the group unlisted under it, so that a joiner can
    // never be handed a group that is about to be destroyed; a group that lost a member takes no more joiners (closed), its seats stay empty
    // ... and ONLY that under it: the round the others were waiting for (a whole device launch plus downloads) runs after the registry lock is released,
    // under the group's own lock — every x264_encoder_open / close that touches the batcher would otherwise wait for a GPU round
    bool last, run = false;
    {
        std::lock_guard<std::mutex> reg(g_batch_mu);
        std::unique_lock<std::mutex> lk(g->m);
        g->member[(size_t)s] = 0; g->active--; g->closed = true;
        last = g->active == 0;
        run = !last && g->n_arrived >= g->active && g->n_arrived > 0;      // the others were only waiting for this session
        if (run) g->leaving++;          // keeps the group alive across the gap below: a member that times out and closes meanwhile must not destroy it under this thread
        if (last)
            for (size_t i = 0; i < g_batch_groups.size(); i++) if (g_batch_groups[i] == g) { g_batch_groups.erase(g_batch_groups.begin() + (long)i); break; }
        if (last && g->leaving > 0) { g->orphaned = true; last = false; }          // the leaver still inside destroys it when it is done
    }
    if (run) {
        bool destroy = false;
        {
            std::unique_lock<std::mutex> lk(g->m);
            if (g->active > 0 && g->n_arrived >= g->active && g->n_arrived > 0 && !g->running) batch_run_round(g, lk);
            g->leaving--;
            destroy = g->orphaned && g->leaving == 0;
        }
        if (destroy) batch_destroy(g);
    }
    if (last) batch_destroy(g);
}

// [x264-upstream] encoder/ratecontrol.c qp2qscale / qscale2qp: single floats (powf / log2f), as x264 has them
static inline double rc_qp2qscale(double qp) { return (double)(0.85f * powf(2.0f, ((float)qp - 12.0f) / 6.0f)); }
static inline double rc_qscale2qp(double qscale) { return (double)(12.0f + 6.0f * log2f((float)qscale / 0.85f)); }
static inline double FL(double v) { return (double)(float)v; }          // an assignment to one of x264's float variables
// fdec->f_qp_avg_rc as x264 arrives at it: rc->qpa_rc (a float) gathers qpm * mb_width row by row (x264_ratecontrol_mb), x264_ratecontrol_end divides by the macroblock count
static inline double rc_qp_avg_rc(float qpm, int mbw, int mbh) { float a = 0.f; for (int y = 0; y < mbh; y++) a += qpm * mbw; return (double)(a / (float)(mbw * mbh)); }

static void xlog(const x264_param_t *p, int level, const char *fmt, ...)
{
    if (!p->pf_log || level > p->i_log_level) return;
    va_list ap;
    va_start(ap, fmt);
    p->pf_log(p->p_log_private, level, fmt, ap);
    va_end(ap);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
// get_zone ([x264-upstream] encoder/ratecontrol.c): the LAST zone that holds the picture (display index) wins
static const x264_t::Zone *get_zone(const x264_t *h, int frame)
{
    for (size_t i = h->zones.size(); i-- > 0;) if (frame >= h->zones[i].start && frame <= h->zones[i].end) return &h->zones[i];
    return nullptr;
}
// parse_zones / parse_zone: "<start>,<end>,q=<int>" or "<start>,<end>,b=<float>", zones separated by '/'.  (x264 also lets a zone carry other options after
// the first; those reconfigure the encoder for the zone's pictures and are not implemented: said in the log, the zone keeps its quantiser part)
static void parse_zones(x264_t *h, const char *str)
{
    if (!str || !*str) return;
    std::string all(str);
    size_t pos = 0;
    while (pos <= all.size()) {
        const size_t e = all.find('/', pos);
        const std::string z = all.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        pos = e == std::string::npos ? all.size() + 1 : e + 1;
        if (z.empty()) continue;
        x264_t::Zone zn = { 0, 0, false, 0, 1.f };
        int len = 0;
        if (sscanf(z.c_str(), "%d,%d,q=%d%n", &zn.start, &zn.end, &zn.qp, &len) >= 3) zn.force_qp = true;
        else if (sscanf(z.c_str(), "%d,%d,b=%f%n", &zn.start, &zn.end, &zn.bitrate_factor, &len) >= 3) zn.force_qp = false;
        else if (sscanf(z.c_str(), "%d,%d%n", &zn.start, &zn.end, &len) >= 2) zn.bitrate_factor = 1.f;
        else { xlog(&h->param, X264_LOG_ERROR, "invalid zone: \"%s\"\n", z.c_str()); continue; }
        if (zn.start > zn.end) { xlog(&h->param, X264_LOG_ERROR, "invalid zone: start=%d end=%d\n", zn.start, zn.end); continue; }
        if (!zn.force_qp && zn.bitrate_factor <= 0) { xlog(&h->param, X264_LOG_ERROR, "invalid zone: bitrate_factor=%f\n", zn.bitrate_factor); continue; }
        if ((size_t)len < z.size()) xlog(&h->param, X264_LOG_WARNING, "zone %d,%d: per-zone encoder options (\"%s\") are not implemented in the MI355X path: the zone keeps its quantiser / bitrate part only\n", zn.start, zn.end, z.c_str() + len);
        h->zones.push_back(zn);
    }
}
// x264_ratecontrol_start, constant quantiser: a zone shifts the picture's quantiser by (its qp - the P quantiser), or by -6 log2f(bitrate factor)
static int cqp_zone(const x264_t *h, const x264_t::Zone &z, int q)
{
    float qf = (float)q;
    if (z.force_qp) qf += (float)(z.qp - h->qp_p); else qf -= 6.f * log2f(z.bitrate_factor);
    const float lo = (float)h->param.rc.i_qp_min, hi = (float)(h->param.rc.i_qp_max < 51 ? h->param.rc.i_qp_max : 51);
    qf = qf < lo ? lo : qf > hi ? hi : qf;
    return clampi((int)(qf + 0.5f), 0, 51);
}

// threads that entropy-code row bands of ONE slice (write_slice): X264GPU_CAVLC_THREADS overrides `dflt`
static int cavlc_threads_default(int dflt)
{
    const char *e = getenv("X264GPU_CAVLC_THREADS");
    return clampi(e ? atoi(e) : dflt, 1, 64);
}

// Coding order of a closed GOP of n pictures under --b-adapt 0 (x264_slicetype_decide with a fixed pattern, as bmode_decide walks it for a session
// without lookahead): an IDR picture, then runs of `bframes` B pictures closed by a P picture, the last picture always P; each closing picture first,
// then under --b-pyramid the middle B of a run of two or more as a reference, then the other B pictures in display order.
static std::vector<std::pair<int, int>> gop_coding_order(int n, int bframes, int bpyramid)
{
    std::vector<std::pair<int, int>> out;
    std::vector<int> run;
    for (int i = 0; i < n; i++) {
        const bool closes = i == 0 || (int)run.size() == bframes || i == n - 1;
        if (!closes) { run.push_back(i); continue; }
        out.push_back({ i, i == 0 ? PIC_IDR : PIC_P });
        const int j = (int)run.size(), bref = bpyramid && j > 1 ? (j - 1) / 2 : -1;
        if (bref >= 0) out.push_back({ run[(size_t)bref], PIC_BREF });
        for (int q = 0; q < j; q++) if (q != bref) out.push_back({ run[(size_t)q], PIC_B });
        run.clear();
    }
    return out;
}

static int pick_level(const x264_param_t *p, int mbs, int refs)
{
    double fps = p->i_fps_den ? (double)p->i_fps_num / p->i_fps_den : 25.0;
    for (int i = 0; x264_levels[i].level_idc; i++) {
        const x264_level_t &l = x264_levels[i];
        if (l.level_idc == 9) continue;
        if (l.frame_size >= mbs && l.mbps >= (int)(mbs * fps) && l.dpb >= mbs * refs) return l.level_idc;
    }
    return 62;
}

static const char kSeiText[] = "x264vfw-mi355x hot path r1 - H.264/MPEG-4 AVC codec - options: cavlc ip p8x8 i4x4 i8x8 8x8dct hex ref<=4 cqp";

static SpsParams make_sps(const x264_t *h)
{
    const x264_param_t &p = h->param;
    SpsParams s = {};
    s.profile_idc = h->profile_idc; s.level_idc = h->level_idc; s.sps_id = p.i_sps_id;
    s.mbw = h->mbw; s.mbh = h->mbh; s.crop_right = h->mbw * 16 - p.i_width; s.crop_bottom = h->mbh * 16 - p.i_height;
    s.num_ref_frames = h->param.i_frame_reference; s.log2_max_frame_num = h->log2_max_frame_num;
    s.sar_w = p.vui.i_sar_width; s.sar_h = p.vui.i_sar_height; s.fullrange = p.vui.b_fullrange;
    s.colorprim = p.vui.i_colorprim; s.transfer = p.vui.i_transfer; s.colmatrix = p.vui.i_colmatrix;
    s.overscan = p.vui.i_overscan; s.vidformat = p.vui.i_vidformat;
    s.num_units_in_tick = p.i_timebase_num; s.time_scale = p.i_timebase_den * 2;
    s.constraint_set0 = h->profile_idc == 66; s.constraint_set1 = h->profile_idc <= 77;
    s.mv_range = h->param.analyse.i_mv_range;
    if (h->dpbmode) { s.num_ref_frames = h->dpb.max_dpb; s.log2_max_poc_lsb = h->log2_max_poc_lsb; s.num_reorder_frames = h->dpb.num_reorder; }
    return s;
}
static PpsParams make_pps(const x264_t *h)
{
    PpsParams pp = { h->param.i_sps_id, h->param.i_sps_id, h->param.b_cabac, h->param.i_frame_reference, h->pic_init_qp, h->param.analyse.i_chroma_qp_offset,
                     h->param.analyse.b_transform_8x8 };
    if (h->bframes && h->param.analyse.b_weighted_bipred) pp.weighted_bipred_idc = 2;
    pp.weighted_pred = h->weightp > 0;
    return pp;
}
// appends SPS, PPS (and optionally the version SEI) to h->out, recording NAL offsets and types
// access unit delimiter (--aud, 7.3.2.4, Table 7-5): primary_pic_type 0 = I slices only, 1 = I and P, 2 = I, P and B (what x264 writes for its three
// slice types); first NAL of the access unit (long start code)
static void write_aud(std::vector<uint8_t> &out, int pic_type /* 0 I, 1 P, 2 B */, bool annexb)
{
    BitWriter bw;
    bw.put((unsigned)pic_type, 3);
    bw.trailing();
    append_nal(out, 0, 9, bw.bytes(), annexb, true);
}

static void emit_sets(x264_t *h, std::vector<int> &types, bool sei)
{
    const bool annexb = h->param.b_annexb != 0;
    h->nal_off.push_back(h->out.size()); types.push_back(7);
    write_sps(h->out, make_sps(h), annexb);
    h->nal_off.push_back(h->out.size()); types.push_back(8);
    write_pps(h->out, make_pps(h), annexb);
    if (sei) {
        h->nal_off.push_back(h->out.size()); types.push_back(6);
        write_sei_version(h->out, kSeiText, annexb);
    }
}

extern "C" {
static bool p2_load(x264_t *h, const char *path);
static bool p2_init(x264_t *h);

x264_t *x264_encoder_open(x264_param_t *param)
{
    if (!param) return nullptr;
    if (param->i_width < 16 || param->i_height < 16 || (param->i_width & 1) || (param->i_height & 1)) {
        xlog(param, X264_LOG_ERROR, "invalid width x height (%dx%d)\n", param->i_width, param->i_height);
        return nullptr;
    }
    if ((param->i_csp & X264_CSP_MASK) != X264_CSP_I420) { xlog(param, X264_LOG_ERROR, "only i420 input is supported\n"); return nullptr; }
    if (x264gpu_device_count() < 1) { xlog(param, X264_LOG_ERROR, "no MI355X device visible (there is no CPU fallback)\n"); return nullptr; }
    x264_t *h = new x264_t();
    h->param = *param;
    x264_param_t &p = h->param;
    if (p.rc.psz_stat_in) { h->stat_in = p.rc.psz_stat_in; p.rc.psz_stat_in = &h->stat_in[0]; }     // codec.c:1386 stack buffers
    if (p.rc.psz_stat_out) { h->stat_out = p.rc.psz_stat_out; p.rc.psz_stat_out = &h->stat_out[0]; }
    h->mbw = (p.i_width + 15) / 16; h->mbh = (p.i_height + 15) / 16; h->nmb = h->mbw * h->mbh;
    if (!p.i_timebase_num || !p.i_timebase_den) { p.i_timebase_num = p.i_fps_den; p.i_timebase_den = p.i_fps_num; }

    // ---- effective parameters: what this round's pipeline implements (reported back via encoder_parameters) ----
    // B pictures: on the device in CABAC sessions, one GOP in flight (settled below, once those are known)
    p.i_bframe = clampi(p.i_bframe, 0, 16);
    if (p.i_frame_reference > 5) { xlog(&p, X264_LOG_INFO, "ref %d -> 5 (DPB of the MI355X path holds up to 5 references)\n", p.i_frame_reference); p.i_frame_reference = 5; }
    if (p.i_frame_reference < 1) p.i_frame_reference = 1;
    p.analyse.b_mixed_references = p.analyse.b_mixed_references && p.i_frame_reference > 1;      // x264 validate_parameters
    p.b_cabac = p.b_cabac != 0;
    if (p.b_cabac && p.i_cabac_init_idc != 0) { xlog(&p, X264_LOG_WARNING, "cabac-idc %d: only the context tables of cabac_init_idc 0 (x264's default) are in the MI355X path: cabac-idc 0\n", p.i_cabac_init_idc); p.i_cabac_init_idc = 0; }
    // --weightp: settled below with the B-picture settings (2 = x264's blind duplicate of reference 0 in sessions that run on the DPB model;
    // the fade analysis that produces other weights — x264_weights_analyse, all that --weightp 1 does — is not implemented)
    p.analyse.i_weighted_pred = clampi(p.analyse.i_weighted_pred, X264_WEIGHTP_NONE, X264_WEIGHTP_SMART); p.analyse.b_weighted_bipred = p.analyse.b_weighted_bipred != 0;
    p.analyse.b_transform_8x8 = p.analyse.b_transform_8x8 != 0;
    if (p.analyse.inter & X264_ANALYSE_PSUB8x8) xlog(&p, X264_LOG_WARNING, "partitions p4x4 (8x4 / 4x8 / 4x4 searches inside P_8x8) are not implemented in the MI355X path: p8x8 stays, p4x4 off\n");
    p.analyse.inter &= X264_ANALYSE_I4x4 | X264_ANALYSE_I8x8 | X264_ANALYSE_PSUB16x16 | X264_ANALYSE_BSUB16x16; p.analyse.intra &= X264_ANALYSE_I4x4 | X264_ANALYSE_I8x8;
    if (!p.analyse.b_transform_8x8) { p.analyse.inter &= ~X264_ANALYSE_I8x8; p.analyse.intra &= ~X264_ANALYSE_I8x8; }   // as x264 validate_parameters
    p.analyse.i_trellis = clampi(p.analyse.i_trellis, 0, 2);          // settled below, once the sub-pel level is known
    p.i_scenecut_threshold = clampi(p.i_scenecut_threshold, 0, 100);
    if (p.analyse.i_me_method > X264_ME_ESA) { xlog(&p, X264_LOG_WARNING, "me tesa is not implemented in the MI355X path yet: me esa\n"); p.analyse.i_me_method = X264_ME_ESA; }
    p.analyse.i_me_range = clampi(p.analyse.i_me_range, 4, p.analyse.i_me_method == X264_ME_UMH ? 64 : 16);     // x264 caps dia/hex at 16; esa: the LDS search window
    // subme 6 / 7 = RD mode decision in P and I slices (x264's i_mbrd 1): on the device, with the bit counts of the session's entropy coder
    // (CAVLC: exact; CABAC: x264's size-only coder on the slice's context states, which the device carries through the macroblock loop);
    // 8 adds RD refinement of the chosen type's vectors and intra modes in I and P slices (i_mbrd 2: x264_me_refine_qpel_rd, intra_rd_refine; B slices
    // analyse one level down, i.e. as at subme 7): on the device in CABAC sessions under --me hex / umh.  9 adds the refinement in B slices (per-list
    // x264_me_refine_qpel_rd, x264_me_refine_bidir_rd, intra_rd_refine), chroma in their sub-pel costs and the deblock-aware RD costs.  10 and 11 (QP-RD,
    // full-RD trellis) are not implemented.  The highest level whose behaviour IS implemented is reported back
    if (p.analyse.i_subpel_refine > 9) { xlog(&p, X264_LOG_WARNING, "subme %d: QP-RD / full-RD trellis of the levels above 9 are not implemented yet: subme 9\n", p.analyse.i_subpel_refine); p.analyse.i_subpel_refine = 9; }
    if (p.analyse.i_subpel_refine >= 8 && (!p.b_cabac || (p.analyse.i_me_method != X264_ME_HEX && p.analyse.i_me_method != X264_ME_UMH))) {
        xlog(&p, X264_LOG_WARNING, "subme %d (RD refinement) needs CABAC and me hex / umh in the MI355X path: subme 7\n", p.analyse.i_subpel_refine); p.analyse.i_subpel_refine = 7;
    }
    p.analyse.i_subpel_refine = clampi(p.analyse.i_subpel_refine, 0, 9);
    // trellis 1 = the final encode of every macroblock quantised by x264's trellis search on the slice's CABAC state: on the device where that state
    // lives, i.e. in CABAC sessions with RD (subme >= 6); trellis 2 = also the block encodes of the intra analysis and every RD candidate
    if (p.analyse.i_trellis && (!p.b_cabac || p.analyse.i_subpel_refine < 6)) {
        xlog(&p, X264_LOG_WARNING, "trellis %d needs CABAC and subme >= 6 in the MI355X path (the search reads the CABAC state the device carries for RD): trellis 0\n", p.analyse.i_trellis);
        p.analyse.i_trellis = 0;
    }
    if (p.analyse.f_psy_trellis > 0) { xlog(&p, X264_LOG_WARNING, "psy-trellis is not implemented in the MI355X path: psy-trellis 0\n"); p.analyse.f_psy_trellis = 0; }
    p.analyse.b_psy = p.analyse.b_psy != 0;
    if (!p.analyse.b_psy) { p.analyse.f_psy_rd = 0; p.analyse.f_psy_trellis = 0; }       // x264 validate_parameters
    p.analyse.f_psy_rd = p.analyse.f_psy_rd < 0 ? 0 : p.analyse.f_psy_rd > 10 ? 10 : p.analyse.f_psy_rd;
    const int psy_rd_q8 = p.analyse.i_subpel_refine >= 6 ? (int)(p.analyse.f_psy_rd * 256.0f + 0.5f) : 0;      // h->mb.i_psy_rd
    // psy RD raises luma quality at chroma's cost, so x264 lowers the chroma quantiser offset to compensate (encoder.c, validate / mb init)
    int eff_chroma_qp_offset = p.analyse.i_chroma_qp_offset;
    if (psy_rd_q8) eff_chroma_qp_offset -= p.analyse.f_psy_rd < 0.25f ? 1 : 2;
    eff_chroma_qp_offset = clampi(eff_chroma_qp_offset, -12, 12);
    p.analyse.i_chroma_qp_offset = eff_chroma_qp_offset;          // x264 changes the parameter in place too: the PPS carries it
    p.analyse.b_fast_pskip = p.analyse.b_fast_pskip != 0;
    p.analyse.b_chroma_me = p.analyse.b_chroma_me != 0;
    if (p.b_interlaced) { xlog(&p, X264_LOG_WARNING, "interlaced coding is not implemented in the MI355X path: progressive\n"); p.b_interlaced = 0; }
    if (p.b_constrained_intra) { xlog(&p, X264_LOG_WARNING, "constrained-intra is not implemented in the MI355X path: off\n"); p.b_constrained_intra = 0; }
    if (p.b_intra_refresh) { xlog(&p, X264_LOG_WARNING, "intra-refresh is not implemented in the MI355X path: off\n"); p.b_intra_refresh = 0; }
    if (p.i_nal_hrd) { xlog(&p, X264_LOG_WARNING, "nal-hrd needs VBV, which is not implemented in the MI355X path: none\n"); p.i_nal_hrd = 0; }
    // --sliced-threads / --tune zerolatency: i_threads is the number of slices per picture (x264 validate_parameters: at most one per four
    // macroblock rows; auto = as many as that allows, where x264 would count host cores) and there is one GOP in flight
    h->slices = 1;
    if (p.b_sliced_threads) {
        const int max_slices = (p.i_height + 15) / 16 / 4 > 1 ? (p.i_height + 15) / 16 / 4 : 1;
        h->slices = p.i_threads <= 0 ? max_slices : p.i_threads < max_slices ? p.i_threads : max_slices;
        if (p.i_threads > 0 && h->slices != p.i_threads) xlog(&p, X264_LOG_INFO, "sliced threads %d -> %d (four macroblock rows per slice)\n", p.i_threads, h->slices);
        p.i_threads = 1;
        if (h->slices < 2) p.b_sliced_threads = 0;
        // x264 has no word for "slice threads AND several GOPs in flight"; this library can do both (slices x GOP slots wavefronts per stream).
        // X264GPU_GOP_SLOTS=G asks for it: the --threads G mode below (fixed keyint, delay (G-1) x keyint) with every picture in slices
        if (const char *gs = getenv("X264GPU_GOP_SLOTS")) { const int g = atoi(gs); if (g > 1) p.i_threads = g; }
        if (p.i_slice_count > 1) xlog(&p, X264_LOG_INFO, "slices %d ignored under slice threads (%d slices)\n", p.i_slice_count, h->slices);
        if (h->slices > 1) p.i_slice_count = 0;
    }
    // --slices N (x264 slices_write: slice i ends at macroblock row (mbh * (i + 1) + N / 2) / N, the split slice threads use too; validate_parameters
    // clips N to the macroblock rows): every slice a wavefront of its own here, which is what makes one stream faster without any delay.
    // Unlike slice threads the loop filter crosses the boundaries (disable_deblocking_filter_idc 0).
    if (!p.b_sliced_threads && p.i_slice_count > 1) {
        const int mbh = (p.i_height + 15) / 16;
        p.i_slice_count = p.i_slice_count < mbh ? p.i_slice_count : mbh;
        h->slices = p.i_slice_count; h->slices_plain = h->slices > 1;
    } else if (!p.b_sliced_threads) p.i_slice_count = 0;
    p.i_threads = clampi(p.i_threads, 1, 256);                 // --threads G: GOPs coded in lock-step (1 = no delay)
    if (p.i_bframe) {
        // (below --subme 7 x264 analyses B slices without RD: k_mb_b.inc's NORD flow; from 7 up their RD decisions count CABAC sizes or CAVLC bits)
        // (--threads G: closed GOPs in lock-step carry B pictures when every GOP has the same picture structure and quantisers: constant quantiser)
        const char *why = (p.i_threads > 1 && p.rc.i_rc_method != X264_RC_CQP) ? "threads 1, or a constant quantiser with --threads G" : p.i_keyint_max < 2 ? "keyint > 1" :
                          (p.rc.i_rc_method == X264_RC_ABR && p.rc.i_bitrate <= 0) ? "constant-quantiser, CRF or ABR rate control with a bitrate" : nullptr;
        if (why) { xlog(&p, X264_LOG_WARNING, "B-frames need %s in the MI355X path: bframes 0\n", why); p.i_bframe = 0; }
    }
    if (p.i_bframe && p.i_threads > 1) {
        if (p.i_bframe_adaptive) { xlog(&p, X264_LOG_INFO, "b-adapt needs threads 1 (GOPs in lock-step have a fixed structure): b-adapt 0\n"); p.i_bframe_adaptive = 0; }
        if (p.analyse.i_direct_mv_pred == 3) { xlog(&p, X264_LOG_INFO, "direct auto needs threads 1 (its choice follows the pictures coded before, across GOPs): spatial\n"); p.analyse.i_direct_mv_pred = 1; }
    }
    if (p.i_bframe) {
        p.i_bframe_adaptive = clampi(p.i_bframe_adaptive, 0, 2);
        if (p.i_bframe_pyramid == 1) { xlog(&p, X264_LOG_INFO, "b-pyramid strict -> normal\n"); p.i_bframe_pyramid = 2; }
        if (p.i_bframe < 2) p.i_bframe_pyramid = 0;
        if (p.b_open_gop) { xlog(&p, X264_LOG_WARNING, "open-gop is not implemented in the MI355X path: closed GOPs\n"); p.b_open_gop = 0; }
        if (p.analyse.i_direct_mv_pred < 1 || p.analyse.i_direct_mv_pred > 3) { xlog(&p, X264_LOG_INFO, "direct %d -> spatial (B macroblocks without direct prediction are not in the MI355X path)\n", p.analyse.i_direct_mv_pred); p.analyse.i_direct_mv_pred = 1; }
        if (p.analyse.i_direct_mv_pred != 1 && getenv("X264GPU_BATCH")) { xlog(&p, X264_LOG_INFO, "direct temporal / auto: not in cross-session batches: spatial\n"); p.analyse.i_direct_mv_pred = 1; }
    } else { p.i_bframe_pyramid = 0; p.analyse.b_weighted_bipred = 0; }
    h->bframes = p.i_bframe; h->bpyramid = p.i_bframe_pyramid ? 1 : 0;
    h->direct_mode = p.i_bframe ? p.analyse.i_direct_mv_pred : 1;
    if (p.i_keyint_max <= 0) p.i_keyint_max = 1;
    if (p.analyse.i_weighted_pred == X264_WEIGHTP_SIMPLE && !h->bframes) { xlog(&p, X264_LOG_INFO, "weightp 1 (weights for fades, no duplicate references) runs in sessions with B pictures only: weightp 0\n"); p.analyse.i_weighted_pred = X264_WEIGHTP_NONE; }
    if (p.analyse.i_weighted_pred == X264_WEIGHTP_SMART && !h->bframes) {
        // without B pictures the session can still run on the DPB model, if nothing of the other path is asked for
        const bool tree = p.rc.b_mb_tree && p.rc.i_rc_method != X264_RC_CQP && p.rc.i_lookahead > 0;
        const char *why = p.i_threads > 1 ? "threads 1" : tree ? "no mbtree" :
                          (p.rc.i_rc_method == X264_RC_ABR && p.rc.i_bitrate <= 0) ? "constant-quantiser, CRF or ABR rate control with a bitrate" : p.i_keyint_max < 2 ? "keyint > 1" : nullptr;
        if (why) { xlog(&p, X264_LOG_WARNING, "weightp 2 without B-frames needs %s in the MI355X path: weightp 0\n", why); p.analyse.i_weighted_pred = X264_WEIGHTP_NONE; }
    }
    if (p.analyse.i_weighted_pred == X264_WEIGHTP_SMART && p.i_frame_reference < 2) p.analyse.i_weighted_pred = X264_WEIGHTP_NONE;      // a duplicate needs two references (x264: never placed)
    h->weightp = p.analyse.i_weighted_pred;
    h->dpbmode = h->bframes > 0 || h->weightp == X264_WEIGHTP_SMART;
    if (const char *be = getenv("X264GPU_BATCH")) {
        // cross-session batcher: this session joins (or starts) a group of N sessions coded in lock-step, if its picture structure is fixed
        const int bn = atoi(be);
        if (bn >= 2) {
            const bool tree = p.rc.b_mb_tree && p.rc.i_rc_method != X264_RC_CQP && p.rc.i_lookahead > 0;
            const char *why = p.i_threads > 1 ? "threads 1" : h->slices > 1 || p.b_sliced_threads ? "one slice per picture" : p.i_scenecut_threshold > 0 ? "scenecut 0" :
                              (h->bframes && p.i_bframe_adaptive) ? "b-adapt 0" : tree ? "no-mbtree" : p.rc.i_rc_method == X264_RC_ABR ? "constant-quantiser or CRF rate control" : nullptr;
            if (why) xlog(&p, X264_LOG_WARNING, "X264GPU_BATCH needs %s (picture k must have the same type in every session of a batch): this session runs on its own\n", why);
            else { h->batch_n = bn; h->dpbmode = true; }
        }
    }
    if (h->dpbmode && !h->bframes) { p.rc.b_mb_tree = 0; }
    if (h->weightp) xlog(&p, X264_LOG_INFO, "weightp %d: weights for fades from the lookahead (luma, and the chroma planes beside it)%s\n", h->weightp, h->weightp == X264_WEIGHTP_SMART ? ", duplicates of reference 0 on every P picture" : "");
    h->keyint = p.i_keyint_max;
    // rate control: constant QP (X264_RC_CQP, codec.c:1498-1502) and single-pass CRF without AQ / mbtree (codec.c:1504-1507, the
    // driver's default session) when one GOP is in flight; ABR and CRF under --threads > 1 map to their nominal quantiser
    int qp = p.rc.i_rc_method == X264_RC_CQP ? p.rc.i_qp_constant : p.rc.i_rc_method == X264_RC_CRF ? (int)(p.rc.f_rf_constant + 0.5f) : 26;
    h->crf = p.rc.i_rc_method == X264_RC_CRF && p.rc.f_rf_constant >= 1.0f;       // also under --threads G: its quantisers follow from the lookahead costs alone
    // 2-pass: the second pass plans every picture's quantiser from the first pass' statistics; sessions on the DPB model (B pictures or --weightp 2)
    h->pass2 = p.rc.b_stat_read && p.rc.i_rc_method == X264_RC_ABR && p.rc.i_bitrate > 0 && p.i_threads <= 1 && h->dpbmode && p.rc.psz_stat_in && !getenv("X264GPU_BATCH");
    h->pass1 = p.rc.b_stat_write && !p.rc.b_stat_read && p.i_threads <= 1 && h->dpbmode && p.rc.psz_stat_out && !getenv("X264GPU_BATCH");
    if ((p.rc.b_stat_read && !h->pass2) || (p.rc.b_stat_write && !p.rc.b_stat_read && !h->pass1)) {
        // a pass whose statistics cannot be honoured keeps its rate control: it runs as the single-pass session of the same method (ABR at i_bitrate
        // for the driver's multipass encodes, codec.c:1509-1527), as every pass did before 2-pass existed here — not at a constant quantiser
        xlog(&p, X264_LOG_WARNING, "2-pass statistics need threads 1 and B-frames or weightp 2 (the DPB-model path) in the MI355X path: this pass runs as a single pass without them\n");
        if (!h->pass2) p.rc.b_stat_read = 0;
        if (!h->pass1) p.rc.b_stat_write = 0;
    }
    h->abr = p.rc.i_rc_method == X264_RC_ABR && p.i_threads <= 1 && p.rc.i_bitrate > 0 && !p.rc.b_stat_read;      // single pass, no VBV
    if (p.rc.b_stat_write && p.rc.b_stat_read && !h->pass2) xlog(&p, X264_LOG_INFO, "this pass runs without the statistics: they stay as the first pass wrote them\n");
    if (p.rc.i_rc_method != X264_RC_CQP && !h->crf && !h->abr && !h->pass2) xlog(&p, X264_LOG_WARNING, "this rate control mode is not implemented yet (ABR with --threads > 1): constant qp %d\n", qp);
    if (qp < 1) { xlog(&p, X264_LOG_WARNING, "lossless is not supported: qp 1\n"); qp = 1; }
    if (!h->crf && !h->abr && !h->pass2) p.rc.i_rc_method = X264_RC_CQP;
    if ((h->pass1 || h->pass2) && p.rc.b_mb_tree) { xlog(&p, X264_LOG_INFO, "2-pass: the macroblock-tree statistics file is not implemented in the MI355X path: mbtree 0 in both passes\n"); p.rc.b_mb_tree = 0; }
    parse_zones(h, p.rc.psz_zones);
    if (!h->zones.empty()) xlog(&p, X264_LOG_INFO, "%d zone%s (quantiser / bitrate factor per range of pictures)\n", (int)h->zones.size(), h->zones.size() > 1 ? "s" : "");
    if (!h->zones.empty() && h->pass2) xlog(&p, X264_LOG_WARNING, "zones are not applied to the second pass' plan in the MI355X path (the first pass and single-pass sessions honour them)\n");
    if (p.rc.i_vbv_max_bitrate > 0 || p.rc.i_vbv_buffer_size > 0) xlog(&p, X264_LOG_WARNING, "VBV (vbv-maxrate / vbv-bufsize) is not implemented in the MI355X path: unconstrained\n");
    p.rc.i_vbv_max_bitrate = 0; p.rc.i_vbv_buffer_size = 0;
    if (p.analyse.i_noise_reduction) { xlog(&p, X264_LOG_WARNING, "nr (noise reduction) is not implemented in the MI355X path: nr 0\n"); p.analyse.i_noise_reduction = 0; }
    if (p.i_slice_max_size > 0 || p.i_slice_max_mbs > 0) { xlog(&p, X264_LOG_WARNING, "slice-max-size / slice-max-mbs are not implemented in the MI355X path (slices are cut by --slices N or slice threads only)\n"); p.i_slice_max_size = p.i_slice_max_mbs = 0; }
    if (p.b_fake_interlaced || p.b_pic_struct) { xlog(&p, X264_LOG_WARNING, "fake-interlaced / pic-struct are not implemented in the MI355X path: off\n"); p.b_fake_interlaced = p.b_pic_struct = 0; }
    if (p.b_bluray_compat) { xlog(&p, X264_LOG_WARNING, "bluray-compat is not implemented in the MI355X path: off\n"); p.b_bluray_compat = 0; }
    // adaptive quantisation: variance AQ (mode 1) under CRF / ABR; x264 itself switches AQ off under constant QP and at strength 0
    if (p.rc.i_rc_method == X264_RC_CQP || p.rc.f_aq_strength <= 0) p.rc.i_aq_mode = X264_AQ_NONE;
    // macroblock-tree: needs a rate-controlled session and pictures held back (rc-lookahead); x264 switches it off under constant QP
    if (p.rc.i_rc_method == X264_RC_CQP || p.rc.i_lookahead <= 0) p.rc.b_mb_tree = 0;
    if (p.rc.b_mb_tree && p.i_threads > 1) { xlog(&p, X264_LOG_INFO, "mbtree needs threads 1 (GOPs in lock-step are coded before what follows them is seen): mbtree 0\n"); p.rc.b_mb_tree = 0; }
    p.rc.b_mb_tree = p.rc.b_mb_tree != 0;
    p.rc.i_lookahead = p.rc.b_mb_tree ? clampi(p.rc.i_lookahead, 1, p.i_keyint_max < 250 ? (p.i_keyint_max > 1 ? p.i_keyint_max : 1) : 250) : 0;
    p.rc.i_aq_mode = clampi(p.rc.i_aq_mode, 0, 3);
    if (p.rc.i_aq_mode > X264_AQ_VARIANCE && (!h->dpbmode || getenv("X264GPU_BATCH"))) {
        xlog(&p, X264_LOG_WARNING, "aq-mode %d needs B-frames or weightp 2 (the DPB-model path, one session a device encoder) in the MI355X path: aq-mode 1\n", p.rc.i_aq_mode); p.rc.i_aq_mode = X264_AQ_VARIANCE;
    }
    p.rc.i_qp_constant = clampi(qp, 1, 51);
    p.rc.i_qp_min = clampi(p.rc.i_qp_min, 1, 51); p.rc.i_qp_max = clampi(p.rc.i_qp_max, p.rc.i_qp_min, 51);
    if (p.i_threads > 1 && p.i_scenecut_threshold) { xlog(&p, X264_LOG_INFO, "scenecut needs threads 1 (GOPs in lock-step have a fixed structure): scenecut 0\n"); p.i_scenecut_threshold = 0; }
    // x264 validate_parameters: min-keyint auto = min(keyint / 10, fps), then [1, keyint / 2 + 1]
    if (p.i_keyint_min <= 0) { const int fps = (int)(p.i_fps_num / (p.i_fps_den ? p.i_fps_den : 1)); p.i_keyint_min = p.i_keyint_max / 10 < fps ? p.i_keyint_max / 10 : fps; }
    p.i_keyint_min = clampi(p.i_keyint_min, 1, p.i_keyint_max / 2 + 1);
    h->keyint_min = p.i_keyint_min;
    h->qp_p = p.rc.i_qp_constant;
    h->qp_i = clampi((int)(h->qp_p - 6.0 * log2(p.rc.f_ip_factor > 0 ? p.rc.f_ip_factor : 1.0) + 0.5), 1, 51);
    h->pic_init_qp = h->crf || h->abr ? 26 : clampi(h->qp_p, 0, 51);          // CRF moves the slice quantiser both ways: centre the +-26 range of slice_qp_delta
    h->profile_idc = p.analyse.b_transform_8x8 ? 100 : p.b_cabac ? 77 : 66;        // High for the 8x8 transform, Main for CABAC alone, else Baseline-compatible
    h->level_idc = p.i_level_idc > 0 ? p.i_level_idc : pick_level(&p, h->nmb, p.i_frame_reference);
    p.i_level_idc = h->level_idc;
    h->log2_max_frame_num = 4;
    {   // x264 sps init: max_frame_num = keyint * (1 + b-pyramid) + 1
        const long max_frame_num = (long)(h->keyint < 65536 ? h->keyint : 65535) * (h->bpyramid + 1) + 1;
        while ((1L << h->log2_max_frame_num) <= max_frame_num && h->log2_max_frame_num < 16) h->log2_max_frame_num++;
    }
    if (h->weightp && h->profile_idc == 66) h->profile_idc = 77;                  // explicit weighted prediction is a Main profile tool
    if (h->dpbmode) {
        // x264 sps init: pic_order_cnt_type 0 with room for the largest POC distance of a mini-GOP (type 2 without B pictures); the DPB model owns
        // frame_num / lists / marking
        h->dpb.configure(p.i_frame_reference, h->bframes, h->bpyramid, h->log2_max_frame_num, h->weightp);
        const int max_delta_poc = (h->bframes + 2) * (h->bpyramid + 1) * 2;
        h->log2_max_poc_lsb = h->bframes ? 4 : 0;
        while (h->bframes && (1 << h->log2_max_poc_lsb) <= max_delta_poc * 2) h->log2_max_poc_lsb++;
        h->level_idc = p.i_level_idc > 0 ? p.i_level_idc : pick_level(&p, h->nmb, h->dpb.max_dpb);
        p.i_level_idc = h->level_idc;
    }

    x264gpu_config cfg = {};
    // GOP-parallel factor: bounded by the device ring (keyint x G pictures) staying under 24 GB
    h->G = p.i_threads;
    {
        const double pic = (double)p.i_width * p.i_height * 1.5;
        while (h->G > 1 && pic * h->keyint * h->G > 24e9) h->G--;
        if (h->keyint >= (1 << 20)) h->G = 1;                  // "infinite" keyint: nothing to run in parallel
        if (h->G != p.i_threads) xlog(&p, X264_LOG_INFO, "threads %d -> %d (GOP ring of keyint %d pictures)\n", p.i_threads, h->G, h->keyint);
        p.i_threads = h->G;
    }
    cfg.width = p.i_width; cfg.height = p.i_height; cfg.streams = h->G; cfg.refs = p.i_frame_reference; cfg.slices = h->slices; cfg.slices_plain = h->slices_plain; cfg.cabac = p.b_cabac;
    cfg.qp_i = h->qp_i; cfg.qp_p = h->qp_p; cfg.me_range = p.analyse.i_me_range; cfg.subme = p.analyse.i_subpel_refine;
    cfg.deblock = p.b_deblocking_filter; cfg.deblock_alpha = p.i_deblocking_filter_alphac0; cfg.deblock_beta = p.i_deblocking_filter_beta;
    cfg.chroma_qp_offset = eff_chroma_qp_offset;
    cfg.rd = p.analyse.i_subpel_refine >= 8 ? 63 : p.analyse.i_subpel_refine >= 6; cfg.psy_rd_q8 = psy_rd_q8;      // 63: RD + every refinement site (x264's i_mbrd 2)
    if (p.analyse.i_subpel_refine >= 9 && p.b_deblocking_filter) cfg.rd |= 64;          // h->mb.b_deblock_rdo: whole-macroblock RD costs measured after the loop filter
    cfg.trellis = p.analyse.i_trellis == 2 ? 63 + 64 : p.analyse.i_trellis ? 63 : 0;         // every quantiser call of the final encode; + 64: of the analysis too
    cfg.psy = cfg.rd && p.analyse.b_psy;           // x264: the chroma lambda offset table follows b_psy, whatever the psy-rd strength
    cfg.deadzone_inter = p.analyse.i_luma_deadzone[0]; cfg.deadzone_intra = p.analyse.i_luma_deadzone[1];
    cfg.dct_decimate = p.analyse.b_dct_decimate;
    // P slices follow analyse.inter, I slices analyse.intra (bit8 marks the separate I-slice set)
    cfg.partitions = ((p.analyse.inter & X264_ANALYSE_PSUB16x16) ? 1 : 0) | ((p.analyse.inter & X264_ANALYSE_I4x4) ? 2 : 0) | ((p.analyse.inter & X264_ANALYSE_I8x8) ? 4 : 0) |
                     0x100 | ((p.analyse.intra & X264_ANALYSE_I4x4) ? 0x200 : 0) | ((p.analyse.intra & X264_ANALYSE_I8x8) ? 0x400 : 0) |
                     ((p.analyse.inter & X264_ANALYSE_BSUB16x16) ? 0x800 : 0);      // B slices: b8x8
    cfg.dct8x8 = p.analyse.b_transform_8x8;
    cfg.me_method = p.analyse.i_me_method == X264_ME_DIA ? 0 : p.analyse.i_me_method == X264_ME_HEX ? 1 : p.analyse.i_me_method == X264_ME_UMH ? 2 : 3;
    cfg.aq_mode = p.rc.i_aq_mode == X264_AQ_VARIANCE; cfg.aq_strength = p.rc.f_aq_strength * 1.0397f;
    h->aq_mode = p.rc.i_aq_mode;
    cfg.mixed_refs = p.analyse.b_mixed_references && (p.analyse.inter & X264_ANALYSE_PSUB16x16) != 0;
    cfg.chroma_me = p.analyse.b_chroma_me && p.analyse.i_subpel_refine >= 5;     // x264: h->mb.b_chroma_me in P slices
    cfg.fast_pskip = p.analyse.b_fast_pskip;
    // x264 validate_parameters: --mvrange defaults to the level's limit (x264_levels[].mv_range), never above 512 here
    if (p.analyse.i_mv_range <= 0) {
        p.analyse.i_mv_range = 512;
        for (int i = 0; x264_levels[i].level_idc; i++) if (x264_levels[i].level_idc == h->level_idc) p.analyse.i_mv_range = x264_levels[i].mv_range;
    }
    p.analyse.i_mv_range = clampi(p.analyse.i_mv_range, 32, 512);
    cfg.mv_range = p.analyse.i_mv_range;
    h->inflight = 1;
    if (h->dpbmode && p.i_bframe > 0 && h->G == 1 && !h->batch_n && !h->pass1 && !h->pass2 && !h->abr && h->direct_mode != 3) {
        // (--direct auto chooses a B picture's mode from the skip counts of the one before, ABR and 2-pass a picture's quantiser from the sizes of the ones before:
        //  those sessions code one picture at a time)
        const char *ie = getenv("X264GPU_INFLIGHT");
        const int want = getenv("X264GPU_DUMP_RECORDS") ? 1 : ie ? atoi(ie) : 4,          // (the debugging dump follows the serial path)
                  extra = want - 1 < 7 - h->dpb.max_dpb ? want - 1 : 7 - h->dpb.max_dpb;
        if (extra >= 1) { h->inflight = extra + 1; h->dpb.extra_slots = extra; }
    }
    if (h->dpbmode) { cfg.dpb = h->dpb.max_dpb + h->dpb.extra_slots; cfg.weightb = p.analyse.b_weighted_bipred; }
    size_t insz = (size_t)p.i_width * p.i_height * 3 / 2;
    (void)x264gpu_get_device(&h->device);
    bool ok_setup = x264gpu_malloc((void **)&h->d_in, insz) == X264GPU_OK;
    if (ok_setup && h->G > 1) {
        // GOP-parallel: the slots are dealt to the devices (X264GPU_DEVICES caps how many are used)
        int D = x264gpu_device_count();
        if (const char *de = getenv("X264GPU_DEVICES")) { const int v = atoi(de); if (v >= 1 && v < D) D = v; }
        D = clampi(D, 1, h->G);
        h->devs.resize((size_t)D);
        int base = 0;
        for (int d = 0; d < D && ok_setup; d++) {
            x264_t::DevCtx &dc = h->devs[(size_t)d];
            dc.dev = D == 1 ? h->device : d; dc.nsl = (h->G - d + D - 1) / D; dc.base = base; base += dc.nsl;
            cfg.streams = dc.nsl;
            ok_setup = x264gpu_set_device(dc.dev) == X264GPU_OK && x264gpu_encoder_create(&dc.gpu, &cfg) == X264GPU_OK &&
                       x264gpu_malloc((void **)&dc.d_mb, (size_t)dc.nsl * h->nmb * sizeof(x264gpu_mb)) == X264GPU_OK &&
                       x264gpu_malloc((void **)&dc.d_lv, (size_t)dc.nsl * h->nmb * X264GPU_MB_LEVELS * sizeof(int16_t)) == X264GPU_OK &&
                       x264gpu_malloc((void **)&dc.d_ring, (size_t)dc.nsl * h->keyint * insz) == X264GPU_OK;
        }
        const std::string err = ok_setup ? "" : x264gpu_last_error();
        (void)x264gpu_set_device(h->device);
        if (!ok_setup) { xlog(&p, X264_LOG_ERROR, "GPU encoder setup failed: %s\n", err.c_str()); x264_encoder_close(h); return nullptr; }
        if (D > 1) xlog(&p, X264_LOG_INFO, "GOP slots on %d devices (%d + ... per device)\n", D, h->devs[0].nsl);
    } else if (ok_setup && h->batch_n) {
        h->batch = batch_join(cfg, h->batch_n, insz, (size_t)h->nmb, &h->batch_idx);
        ok_setup = h->batch != nullptr;
        if (ok_setup && h->batch->async && !h->batch->up_streams.empty()) h->up_stream = h->batch->up_streams[(size_t)h->batch_idx % h->batch->up_streams.size()];      // (the group's; without one the uploads wait on the default stream: slower, not wrong)
        if (ok_setup) xlog(&p, X264_LOG_INFO, "X264GPU_BATCH: stream %d of a batch of %d sessions%s\n", h->batch_idx, h->batch_n, h->batch->async ? " (rounds queued: uploads overlap the device)" : "");
    } else if (ok_setup) {
        ok_setup = x264gpu_encoder_create(&h->gpu, &cfg) == X264GPU_OK &&
                   x264gpu_malloc((void **)&h->d_mb, (size_t)h->nmb * sizeof(x264gpu_mb)) == X264GPU_OK &&
                   x264gpu_malloc((void **)&h->d_lv, (size_t)h->nmb * X264GPU_MB_LEVELS * sizeof(int16_t)) == X264GPU_OK;
        if (ok_setup && h->inflight > 1) {
            h->lctx.resize((size_t)h->inflight);
            ok_setup = x264gpu_event_create(&h->ev_la) == X264GPU_OK;
            for (int i = 0; i < h->inflight && ok_setup; i++) {
                x264_t::LaunchCtx &c = h->lctx[(size_t)i];
                if (i == 0) { c.gpu = h->gpu; c.d_mb = h->d_mb; c.d_lv = h->d_lv; }
                else ok_setup = x264gpu_encoder_create_view(&c.gpu, h->gpu) == X264GPU_OK &&
                                x264gpu_malloc((void **)&c.d_mb, (size_t)h->nmb * sizeof(x264gpu_mb)) == X264GPU_OK &&
                                x264gpu_malloc((void **)&c.d_lv, (size_t)h->nmb * X264GPU_MB_LEVELS * sizeof(int16_t)) == X264GPU_OK;
                ok_setup = ok_setup && x264gpu_stream_create(&c.stream) == X264GPU_OK && x264gpu_event_create(&c.ev) == X264GPU_OK;
            }
            if (ok_setup) xlog(&p, X264_LOG_INFO, "up to %d pictures of the session in flight (pictures that share only finished references; the stream is the serial one)\n", h->inflight);
        }
    }
    if (!ok_setup) {
        xlog(&p, X264_LOG_ERROR, "GPU encoder setup failed: %s\n", x264gpu_last_error());
        x264_encoder_close(h);
        return nullptr;
    }
    if ((p.i_scenecut_threshold > 0 && !h->dpbmode) || h->crf || h->abr || h->aq_mode >= 2) {       // (sessions on the DPB model take scene cuts from x264's own analysis below)
        if (x264gpu_lookahead_create(&h->la, p.i_width, p.i_height, 1, p.analyse.i_me_range, p.analyse.i_subpel_refine) != X264GPU_OK ||
            x264gpu_malloc((void **)&h->d_la, 4 * sizeof(int32_t)) != X264GPU_OK) {
            xlog(&p, X264_LOG_ERROR, "GPU lookahead setup failed: %s\n", x264gpu_last_error());
            x264_encoder_close(h);
            return nullptr;
        }
    }
    // lookahead queue: rc-lookahead pictures are held back when the macroblock-tree is on (x264's sync lookahead), none otherwise
    h->mbtree = p.rc.b_mb_tree && h->la != nullptr;
    h->weightp_fake = !h->weightp && h->mbtree && p.analyse.b_psy && h->dpbmode;          // x264 validate_parameters: X264_WEIGHTP_FAKE (sessions on the DPB model: the others' tree has no weight analysis)
    h->L = h->mbtree ? p.rc.i_lookahead : 0;
    // pictures are held back anyway and the quantisers do not depend on coded sizes: overlap the GPU stage of the next picture with
    // the entropy coding of this one (one more picture of delay); X264GPU_HOST_PIPELINE=0 keeps the two stages in one call
    { const char *pe = getenv("X264GPU_HOST_PIPELINE"); h->pipeline = h->G == 1 && h->L > 0 && h->crf && !(pe && pe[0] == '0'); }
    h->Q = h->L + 1 + (h->pipeline ? 1 : 0);
    if (h->dpbmode) {
        if (h->L > 60) { xlog(&p, X264_LOG_INFO, "rc-lookahead %d -> 60 in sessions with B pictures (the lookahead keeps every queued picture's half-resolution planes and searches on the device)\n", h->L); h->L = 60; p.rc.i_lookahead = 60; }
        h->pipeline = false; h->Q = h->L + 2 * (h->bframes + 1) + 2;      // the lookahead window + display-order queue + the mini-GOP being coded
    }
    h->st_wait = h->bframes > h->L ? h->bframes : h->L;
    // x264 h->frames.i_delay: the trellis over picture types looks max(bframes, 3) * 4 pictures ahead
    if (h->dpbmode && h->bframes && p.i_bframe_adaptive == 2) { const int d = (h->bframes > 3 ? h->bframes : 3) * 4; if (d > h->st_wait) h->st_wait = d; if (h->Q < h->st_wait + 2 * (h->bframes + 1) + 2) h->Q = h->st_wait + 2 * (h->bframes + 1) + 2; }
    if (h->dpbmode && h->inflight > 1) h->Q += h->inflight + h->bframes + 1;           // ... + the pictures in flight (their source pictures and offsets are read when their kernels run)
    if (h->dpbmode && h->Q > 128) {          // the lookahead object holds 128 pictures
        const int over = h->Q - 128;
        xlog(&p, X264_LOG_INFO, "lookahead window shortened by %d pictures (128 pictures are held at most)\n", over);
        h->L = h->L > over ? h->L - over : 0; p.rc.i_lookahead = h->L;
        if (h->st_wait > 128 - 2 * (h->bframes + 1) - 2) h->st_wait = 128 - 2 * (h->bframes + 1) - 2;
        h->Q = 128;
    }
    h->last_keyframe = -p.i_keyint_max;
    h->badapt = h->bframes ? p.i_bframe_adaptive : 0;
    // (sessions with a fixed picture structure — no scenecut, b-adapt 0, no mbtree — run without the lookahead object: no fade weights and no
    // lookahead vectors as search candidates there; that is also what makes the sessions of a batch equal to the same sessions run alone)
    if (h->dpbmode && (h->badapt || p.i_scenecut_threshold > 0 || h->mbtree)) {
        // x264's own lookahead structure: frame costs of (p0, p1, b) triples on the half-resolution planes (x264_slicetype_analyse)
        if (x264gpu_slicetype_create(&h->st, p.i_width, p.i_height, 1, h->Q, h->bframes, p.analyse.i_me_method, p.analyse.i_subpel_refine, p.analyse.i_me_range,
                                     p.analyse.b_weighted_bipred, p.analyse.i_mv_range, h->mbtree ? 1 : 0) != X264GPU_OK) {
            xlog(&p, X264_LOG_ERROR, "GPU lookahead setup failed: %s\n", x264gpu_last_error());
            x264_encoder_close(h);
            return nullptr;
        }
        p.i_bframe_bias = clampi(p.i_bframe_bias, -90, 100);
        (void)x264gpu_slicetype_set_bframe_bias(h->st, p.i_bframe_bias);          // --b-bias also scales the B costs of slicetype_frame_cost
    }
    h->aq_strength = cfg.aq_mode ? cfg.aq_strength : h->aq_mode >= 2 ? p.rc.f_aq_strength : 0.f;      // (modes 2 / 3: the plain strength, x264_adaptive_quant_frame scales it by the picture's mean itself)
    h->st_aq_costs = h->st && h->la && !h->mbtree && h->aq_strength != 0.f && (h->crf || h->abr);
    h->tree_strength = 5.0f * (1.0f - p.rc.f_qcompress);
    h->q_raw.assign((size_t)h->Q, nullptr); h->q_info.assign((size_t)h->Q, nullptr); h->q_aq.assign((size_t)h->Q, nullptr); h->q_tree.assign((size_t)h->Q, nullptr);
    if (h->Q == 1) h->q_raw[0] = h->d_in;            // no delay: the staging buffer is the one slot; with a delay the ring is separate,
                                                     // because a zero-copy caller rewrites the staging buffer every call
    {
        // one device block per array, cut into the queue's slots (an allocation and a release each cost a fraction of a millisecond: thousands of sessions open and close)
        bool ok = true;
        const size_t Q = (size_t)h->Q, insz_al = (insz + 255) & ~(size_t)255, nmbf = ((size_t)h->nmb * sizeof(float) + 255) & ~(size_t)255, ninfo = ((size_t)h->nmb * 4 * sizeof(int32_t) + 255) & ~(size_t)255;
        const bool want_aq = h->mbtree || h->st_aq_costs || h->aq_mode >= 2, want_tree = h->mbtree && h->dpbmode;
        if (!h->q_raw[0]) ok = x264gpu_malloc((void **)&h->q_block[0], Q * insz_al) == X264GPU_OK;
        if (ok && h->mbtree) ok = x264gpu_malloc((void **)&h->q_block[1], Q * ninfo) == X264GPU_OK;
        if (ok && want_aq) ok = x264gpu_malloc((void **)&h->q_block[2], Q * nmbf) == X264GPU_OK;
        if (ok && want_tree) ok = x264gpu_malloc((void **)&h->q_block[3], Q * nmbf) == X264GPU_OK;
        for (size_t i = 0; i < Q && ok; i++) {
            if (!h->q_raw[i]) h->q_raw[i] = (uint8_t *)h->q_block[0] + i * insz_al;
            if (h->mbtree) h->q_info[i] = (int32_t *)((uint8_t *)h->q_block[1] + i * ninfo);
            if (want_aq) h->q_aq[i] = (float *)((uint8_t *)h->q_block[2] + i * nmbf);
            if (want_tree) h->q_tree[i] = (float *)((uint8_t *)h->q_block[3] + i * nmbf);
        }
        if (ok && h->mbtree) ok = x264gpu_malloc((void **)&h->d_tree, (size_t)h->nmb * sizeof(float)) == X264GPU_OK;
        if (!ok) {
            xlog(&p, X264_LOG_ERROR, "GPU lookahead queue setup failed: %s\n", x264gpu_last_error());
            x264_encoder_close(h);
            return nullptr;
        }
    }
    if (h->crf || h->abr) {
        // x264_ratecontrol_new: rate_factor_constant = base_cplx^(1 - qcomp) / qp2qscale(crf), base_cplx = mbs * (bframes ? 120 : 80)
        auto qp2qscale = [](double q) { return rc_qp2qscale(q); };
        h->rc.qcompress = p.rc.f_qcompress; h->rc.ip_factor = fabs(p.rc.f_ip_factor) > 0 ? fabs(p.rc.f_ip_factor) : 1.0;
        h->rc.ip_offset = 6.0 * log2f((float)h->rc.ip_factor);          // x264_ratecontrol_init_reconfigurable: 6.0 * log2f( f_ip_factor )
        if (p.rc.b_mb_tree) h->rc.qcompress = 1.0;                     // x264_ratecontrol_new: the tree does the complexity weighting, CRF shifts by 13.5 (1 - qcomp)
        h->rc.rate_factor_constant = pow((double)h->nmb * (h->bframes ? 120.0 : 80.0), 1.0 - h->rc.qcompress) / qp2qscale(p.rc.f_rf_constant + (p.rc.b_mb_tree ? (1.0 - p.rc.f_qcompress) * 13.5 : 0.0));
        h->rc.last_qscale_for[0] = h->rc.last_qscale_for[1] = qp2qscale(p.rc.f_rf_constant);
        // x264_ratecontrol_new (b_abr = CRF and ABR alike): the running P quantiser starts with a hundredth of a picture at ABR_INIT_QP (CRF: the rate factor; ABR: 24)
        h->rc.accum_p_norm = .01; h->rc.accum_p_qp = (h->crf ? (double)p.rc.f_rf_constant : 24.0) * h->rc.accum_p_norm;
        h->rc.lmin = qp2qscale(p.rc.i_qp_min); h->rc.lmax = qp2qscale(p.rc.i_qp_max);
        double dur = p.i_fps_num ? (double)p.i_fps_den / p.i_fps_num : 0.04;
        dur = dur < 0.01 ? 0.01 : dur > 1.0 ? 1.0 : dur;              // CLIP_DURATION
        h->rc.dur_ratio = dur / 0.04;                                  // BASE_FRAME_DURATION
        if (h->abr) {
            // x264_ratecontrol_new / x264_ratecontrol_init_reconfigurable, ABR without VBV
            const double abr_init_qp = 24.0;
            h->rc.bitrate = p.rc.i_bitrate * 1000.0; h->rc.fps = p.i_fps_num ? (double)p.i_fps_num / p.i_fps_den : 25.0;
            h->rc.cplxr_sum = 0.01 * pow(7.0e5, h->rc.qcompress) * pow((double)h->nmb, 0.5);
            h->rc.wanted_bits_window = h->rc.bitrate / h->rc.fps;
            h->rc.abr_buffer = 2.0 * (p.rc.f_rate_tolerance > 0.01f ? p.rc.f_rate_tolerance : 0.01f) * h->rc.bitrate;
            h->rc.lstep = pow(2.0, (p.rc.i_qp_step > 0 ? p.rc.i_qp_step : 4) / 6.0);
            h->rc.last_qscale_for[0] = h->rc.last_qscale_for[1] = qp2qscale(abr_init_qp);
        }
    }
    h->rc.fps = p.i_fps_num ? (double)p.i_fps_num / p.i_fps_den : 25.0;
    if (h->pass2) {
        if (!p2_load(h, p.rc.psz_stat_in) || !p2_init(h)) { x264_encoder_close(h); return nullptr; }
        xlog(&p, X264_LOG_INFO, "2-pass: %d pictures planned from the first pass' statistics, %.1f kbit expected before the last one\n", (int)h->p2.size(), h->p2_final_bits / 1000.0);
    }
    // (the driver's N-th pass asks for both: statistics read AND written again — codec.c:1519-1541 with its fixed updatestats — so that a further pass plans from this one's pictures)
    const bool stat_update = h->pass2 && p.rc.b_stat_write && p.rc.psz_stat_out;
    if (h->pass1 || stat_update) {
        // x264 writes <stats>.temp and renames it when the encoder closes; the first line names the options the second pass must agree with
        h->stat_file = fopen((std::string(p.rc.psz_stat_out) + ".temp").c_str(), "wb");
        if (!h->stat_file) { xlog(&p, X264_LOG_ERROR, "ratecontrol_init: can't open stats file\n"); x264_encoder_close(h); return nullptr; }
        fprintf(h->stat_file, "#options: %dx%d fps=%u/%u timebase=%u/%u bitdepth=8 cabac=%d ref=%d bframes=%d b_pyramid=%d b_adapt=%d weightp=%d keyint=%d rc=%s\n", p.i_width, p.i_height,
                p.i_fps_num, p.i_fps_den, p.i_fps_den, p.i_fps_num, p.b_cabac, p.i_frame_reference, p.i_bframe, p.i_bframe_pyramid, p.i_bframe_adaptive, p.analyse.i_weighted_pred, p.i_keyint_max,
                p.rc.i_rc_method == X264_RC_ABR ? "abr" : p.rc.i_rc_method == X264_RC_CRF ? "crf" : "cqp");
    }
    { const unsigned hw = std::thread::hardware_concurrency(); h->cavlc_threads = h->G > 1 ? 1 : cavlc_threads_default(hw >= 32 ? 16 : hw >= 16 ? 8 : hw >= 4 ? (int)hw / 2 : 1); }
    h->h_in.resize(insz);
    // (a batch session whose pictures are downloaded and coded by its helper threads keeps the records in the two deferred slots instead: 7 MB less to clear per open)
    if (!(h->batch && h->batch->overlap)) { h->h_mb.resize((size_t)h->G * h->nmb); h->h_lv.resize((size_t)h->G * h->nmb * X264GPU_MB_LEVELS); }
    if (h->pipeline) { h->h_mb2.resize(h->h_mb.size()); h->h_lv2.resize(h->h_lv.size()); }
    if (h->G > 1) {
        const size_t n = (size_t)h->G * h->keyint;
        h->gop_qp.assign(n, (int8_t)h->qp_p); h->gop_qpm.assign(n, 0.f);
        h->slotbuf.resize(n);
        h->slot_have.assign(n, 0);
        h->h_mb2.resize(h->h_mb.size()); h->h_lv2.resize(h->h_lv.size());
        if (h->dpbmode) {
            h->gopb = true;
            h->gorder = gop_coding_order(h->keyint, h->bframes, h->bpyramid);
            h->gdpb.dpb.configure(p.i_frame_reference, h->bframes, h->bpyramid, h->log2_max_frame_num, h->weightp);
            xlog(&p, X264_LOG_INFO, "%d GOP slots in lock-step with B pictures (bframes %d, b-pyramid %d, constant quantiser): pictures leave in coding order, %d calls late\n", h->G, h->bframes, h->bpyramid, (h->G - 1) * h->keyint + h->bframes + 1);
        }
    }
    xlog(&p, X264_LOG_INFO, "MI355X hot path: %dx%d, %d MBs, CQP I:%d P:%d, keyint %d, level %d\n", p.i_width, p.i_height, h->nmb,
         h->qp_i, h->qp_p, h->keyint, h->level_idc);
    return h;
}

void x264_encoder_parameters(x264_t *h, x264_param_t *param)
{
    if (!h || !param) return;
    *param = h->param;
    if (h->slices > 1 && !h->slices_plain && h->G <= 1) param->i_threads = h->slices;      // slice threads: i_threads is the slice count, as in x264
}

static void publish_nals(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, const std::vector<int> &types)
{
    h->nals.resize(h->nal_off.size());
    for (size_t i = 0; i < h->nal_off.size(); i++) {
        x264_nal_t &n = h->nals[i];
        memset(&n, 0, sizeof(n));
        size_t end = i + 1 < h->nal_off.size() ? h->nal_off[i + 1] : h->out.size();
        n.p_payload = h->out.data() + h->nal_off[i];
        n.i_payload = (int)(end - h->nal_off[i]);
        n.i_type = types[i];
        n.i_ref_idc = types[i] == 6 ? 0 : types[i] == 1 ? 2 : 3;
        n.b_long_startcode = 1;
    }
    *pp_nal = h->nals.data();
    *pi_nal = (int)h->nals.size();
}

int x264_encoder_headers(x264_t *h, x264_nal_t **pp_nal, int *pi_nal)
{
    if (!h || !pp_nal || !pi_nal) return -1;
    h->out.clear(); h->nal_off.clear();
    std::vector<int> types;
    emit_sets(h, types, true);               // nal[0]=SPS nal[1]=PPS nal[2]=SEI (output/raw.c:41-47)
    publish_nals(h, pp_nal, pi_nal, types);
    return (int)h->out.size();
}

// the float quantiser as the device takes it: beside its rounding (a quantiser clipped into 1..51 from outside has no fraction to carry: 0 = the integer one)
static float near_qpm(double qpf, int qp) { const float f = (float)qpf; return f > (float)qp - 1.f && f < (float)qp + 1.f ? f : 0.f; }
// ---- GOP-parallel mode ------------------------------------------------------------------------------------------------
// Frame i of the stream belongs to GOP g = i / keyint at position t = i % keyint; GOP g runs on slot g % G of batch g / G.
// Pictures are uploaded into a position-major device ring, so the G pictures of one position are contiguous = one
// x264gpu_encode_frames call.  A position is coded when the batch's last GOP delivers it (or at flush with the slots that
// exist); the G slices are entropy-coded by a thread each.  Frames leave in stream order, one per call.
static void join_pool(x264_t *h)
{
    for (auto &th : h->pool) th.join();
    h->pool.clear();
    // only now do the frames of that position count as coded (the workers never touch the bookkeeping)
    for (int s = h->pool_slot0; s < h->pool_slot0 + h->pool_nslots; s++) h->slot_have[(size_t)s * h->keyint + h->pool_t] = 1;
    h->pool_nslots = 0; h->pool_slot0 = 0;
}

static int rc_pick_qp(x264_t *h, bool is_i, const int32_t costs[4], int frames_done, int frame);

// GOP slots with B pictures: coding position c of `order` for the slots [slot0, slot0 + nslots) of batch `batch` — the plan from the DPB model (the
// same for every slot), one x264gpu_encode_pictures per device that owns one of the slots, the slices written by the pool while the next position runs.
// 0, or -1 after a GPU failure (the session is then dead, as in code_position).
static int code_position_b(x264_t *h, int batch, int c, int slot0, int nslots, const std::vector<std::pair<int, int>> &order, x264_t::GopDpb &gd)
{
    const x264_param_t &p = h->param;
    const size_t insz = (size_t)p.i_width * p.i_height * 3 / 2;
    const int G = h->G, K = h->keyint, disp = order[(size_t)c].first, type = order[(size_t)c].second;
    x264gpu_mb *hmb = h->dl ? h->h_mb2.data() : h->h_mb.data();
    int16_t *hlv = h->dl ? h->h_lv2.data() : h->h_lv.data();
    h->dl ^= 1;
    // the disposable pictures coded right behind this one (x264_reference_hierarchy_reset looks at them)
    int fc[16], ff[16], nf = 0;
    for (size_t i = (size_t)c + 1; i < order.size() && nf < 16 && order[i].second == PIC_B; i++) { fc[nf] = (int)i; ff[nf] = order[i].first; nf++; }
    const DpbPlan plan = gd.dpb.plan(type, disp, nf, fc, ff, nullptr);
    x264gpu_pic pic = plan.pic;
    if ((type == PIC_B || type == PIC_BREF) && h->direct_mode == 2) {
        // x264 slice_header_init: temporal direct prediction only when the co-located picture's reference 0 is this picture's reference 0
        const bool temporal = pic.nref[0] && pic.nref[1] && gd.l0ref0poc[pic.slot[1][0]] == plan.list_poc[0][0];
        pic.direct_temporal = temporal; pic.direct_auto = 0;
        gd.dpb.set_direct(pic.direct_temporal, 0);
    }
    if (plan.nal_ref_idc) gd.l0ref0poc[pic.dst] = pic.nref[0] ? plan.list_poc[0][0] : INT_MIN;
    // constant quantiser by picture type (x264_ratecontrol_start), a zone shifts it by the picture's display index in the stream
    const double pb_offset = 6.0 * log2f(fabs(p.rc.f_pb_factor) > 0 ? fabsf(p.rc.f_pb_factor) : 1.0f);
    const int qb = clampi((int)(h->qp_p + pb_offset + 0.5), 0, 51);
    const int q_type = type <= PIC_I ? h->qp_i : type == PIC_P ? h->qp_p : type == PIC_BREF ? (qb + h->qp_p) / 2 : qb;
    std::vector<int> qps((size_t)G, q_type);
    if (!h->zones.empty())
        for (int s = slot0; s < slot0 + nslots; s++)
            if (const x264_t::Zone *z = get_zone(h, (int)(((long)batch * G + s) * K + disp))) qps[(size_t)s] = cqp_zone(h, *z, q_type);
    SliceParams sp = {};
    sp.mbw = h->mbw; sp.mbh = h->mbh; sp.pic_init_qp = h->pic_init_qp; sp.log2_max_frame_num = h->log2_max_frame_num; sp.log2_max_poc_lsb = h->log2_max_poc_lsb;
    sp.pps_id = p.i_sps_id; sp.num_ref_default = p.i_frame_reference; sp.num_ref1_default = 1;
    sp.disable_deblock_idc = p.b_deblocking_filter ? 0 : 1; sp.alpha_off_div2 = p.i_deblocking_filter_alphac0; sp.beta_off_div2 = p.i_deblocking_filter_beta;
    sp.transform8x8_mode = p.analyse.b_transform_8x8; sp.cabac = p.b_cabac; sp.slices_plain = h->slices_plain;
    gd.dpb.fill(sp);
    gd.dpb.commit();
    const int D = (int)h->devs.size();
    std::vector<std::string> errs((size_t)D);
    auto run_dev = [&](int d) {
        x264_t::DevCtx &dc = h->devs[(size_t)d];
        bool mine = false;
        for (int s = slot0; s < slot0 + nslots; s++) mine |= s % D == d;
        if (!mine) return;                                     // (the stream's last, shorter GOP is coded alone: only its device runs)
        bool ok = D == 1 || x264gpu_set_device(dc.dev) == X264GPU_OK;
        std::vector<x264gpu_pic> pics((size_t)dc.nsl, pic);
        for (int l = 0; l < dc.nsl; l++) pics[(size_t)l].qp = qps[(size_t)(l * D + d)];
        ok = ok && x264gpu_encode_pictures(dc.gpu, dc.d_ring + (size_t)disp * dc.nsl * insz, pics.data(), dc.d_mb, dc.d_lv, nullptr) == X264GPU_OK &&
             x264gpu_memcpy_d2h(hmb + (size_t)dc.base * h->nmb, dc.d_mb, (size_t)dc.nsl * h->nmb * sizeof(x264gpu_mb), nullptr) == X264GPU_OK &&
             x264gpu_memcpy_d2h(hlv + (size_t)dc.base * h->nmb * X264GPU_MB_LEVELS, dc.d_lv, (size_t)dc.nsl * h->nmb * X264GPU_MB_LEVELS * sizeof(int16_t), nullptr) == X264GPU_OK;
        if (!ok) errs[(size_t)d] = std::string("device ") + std::to_string(dc.dev) + ": " + x264gpu_last_error();
    };
    if (D == 1) run_dev(0);
    else {
        std::vector<std::thread> ths;
        for (int d = 0; d < D; d++) ths.emplace_back(run_dev, d);
        for (auto &th : ths) th.join();
    }
    for (int d = 0; d < D; d++)
        if (!errs[(size_t)d].empty()) {
            xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", errs[(size_t)d].c_str());
            join_pool(h); h->failed = true;
            return -1;
        }
    join_pool(h);
    const int ref_idc = plan.nal_ref_idc;
    auto work = [h, batch, c, disp, type, hmb, hlv, qps, D, sp, ref_idc](int s) {
        const x264_param_t &p = h->param;
        const int G = h->G;
        const size_t row = (size_t)h->devs[(size_t)(s % D)].base + (size_t)(s / D);
        x264_t::Coded &cd = h->slotbuf[(size_t)s * h->keyint + c];
        cd.bytes.clear(); cd.off.clear(); cd.types.clear();
        cd.idr = type == PIC_IDR; cd.ref_idc = ref_idc; cd.disp = disp;
        cd.i_type = type == PIC_IDR ? X264_TYPE_IDR : type == PIC_I ? X264_TYPE_I : type == PIC_P ? X264_TYPE_P : type == PIC_BREF ? X264_TYPE_BREF : X264_TYPE_B;
        const long gop = (long)batch * G + s;
        const bool annexb = p.b_annexb != 0;
        if (p.b_aud) { cd.off.push_back(cd.bytes.size()); cd.types.push_back(9); write_aud(cd.bytes, type <= PIC_I ? 0 : type == PIC_P ? 1 : 2, annexb); }
        if (cd.idr && p.b_repeat_headers) {
            cd.off.push_back(cd.bytes.size()); cd.types.push_back(7); write_sps(cd.bytes, make_sps(h), annexb);
            cd.off.push_back(cd.bytes.size()); cd.types.push_back(8); write_pps(cd.bytes, make_pps(h), annexb);
            if (gop == 0) { cd.off.push_back(cd.bytes.size()); cd.types.push_back(6); write_sei_version(cd.bytes, kSeiText, annexb); }
        }
        SliceParams sps = sp;
        sps.qp = qps[(size_t)s]; sps.idr_pic_id = (int)(gop & 0xffff);
        const size_t before = cd.off.size();
        write_picture(cd.bytes, &cd.off, sps, h->slices, hmb + row * h->nmb, hlv + row * h->nmb * X264GPU_MB_LEVELS, annexb, before == 0, nullptr);
        for (size_t i = before; i < cd.off.size(); i++) cd.types.push_back(cd.idr ? 5 : 1);
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthr = (int)(hw ? (hw < (unsigned)nslots ? hw : (unsigned)nslots) : 1);
    h->pool_t = c; h->pool_nslots = nslots; h->pool_slot0 = slot0;
    for (int th = 0; th < nthr; th++)
        h->pool.emplace_back([work, th, nthr, slot0, nslots]() { for (int s = slot0 + th; s < slot0 + nslots; s += nthr) work(s); });
    return 0;
}

// 0, or -1 after a GPU failure: the session is then dead (h->failed: every later call returns < 0 and nothing counts as delayed,
// so the caller's flush loop — codec.c:1842-1856 — ends instead of spinning on frames that will never be coded)
static int code_position(x264_t *h, int batch, int t, int nslots_with_t)
{
    const x264_param_t &p = h->param;
    const size_t insz = (size_t)p.i_width * p.i_height * 3 / 2;
    const int st = t == 0 ? X264GPU_SLICE_I : X264GPU_SLICE_P, G = h->G;
    // GPU + download of this position run while the CAVLC threads of the previous position are still coding from the
    // other buffer pair
    x264gpu_mb *hmb = h->dl ? h->h_mb2.data() : h->h_mb.data();
    int16_t *hlv = h->dl ? h->h_lv2.data() : h->h_lv.data();
    h->dl ^= 1;
    std::vector<int8_t> qps; std::vector<float> qpms;
    if (h->crf) {
        qps.assign((size_t)G, (int8_t)(t == 0 ? h->qp_i : h->qp_p)); qpms.assign((size_t)G, 0.f);
        for (int s = 0; s < nslots_with_t; s++) { qps[(size_t)s] = h->gop_qp[(size_t)s * h->keyint + t]; qpms[(size_t)s] = h->gop_qpm[(size_t)s * h->keyint + t]; }
    } else if (!h->abr && !h->zones.empty()) {
        // constant quantiser with zones: slot s of this batch holds picture (batch * G + s) * keyint + t of the stream
        qps.assign((size_t)G, (int8_t)(t == 0 ? h->qp_i : h->qp_p)); qpms.assign((size_t)G, 0.f);
        for (int s = 0; s < nslots_with_t; s++)
            if (const x264_t::Zone *z = get_zone(h, (int)(((long)batch * G + s) * h->keyint + t))) qps[(size_t)s] = (int8_t)cqp_zone(h, *z, qps[(size_t)s]);
    }
    // every device codes position t of its slots; one host thread per device issues the work and collects the results
    const int D = (int)h->devs.size();
    std::vector<std::string> errs((size_t)D);
    auto run_dev = [&](int d) {
        x264_t::DevCtx &dc = h->devs[(size_t)d];
        bool ok = D == 1 || x264gpu_set_device(dc.dev) == X264GPU_OK;
        if (ok && !qps.empty()) {
            std::vector<int8_t> q((size_t)dc.nsl); std::vector<float> qm((size_t)dc.nsl);
            for (int l = 0; l < dc.nsl; l++) { q[(size_t)l] = qps[(size_t)(l * D + d)]; qm[(size_t)l] = qpms[(size_t)(l * D + d)]; }
            ok = x264gpu_encoder_set_stream_qpms(dc.gpu, q.data(), qm.data()) == X264GPU_OK;
        }
        ok = ok && x264gpu_encode_frames(dc.gpu, dc.d_ring + (size_t)t * dc.nsl * insz, st, dc.d_mb, dc.d_lv, nullptr) == X264GPU_OK &&
             x264gpu_memcpy_d2h(hmb + (size_t)dc.base * h->nmb, dc.d_mb, (size_t)dc.nsl * h->nmb * sizeof(x264gpu_mb), nullptr) == X264GPU_OK &&
             x264gpu_memcpy_d2h(hlv + (size_t)dc.base * h->nmb * X264GPU_MB_LEVELS, dc.d_lv, (size_t)dc.nsl * h->nmb * X264GPU_MB_LEVELS * sizeof(int16_t), nullptr) == X264GPU_OK;
        if (!ok) errs[(size_t)d] = std::string("device ") + std::to_string(dc.dev) + ": " + x264gpu_last_error();      // the error text is per thread
    };
    if (D == 1) run_dev(0);
    else {
        std::vector<std::thread> ths;
        for (int d = 0; d < D; d++) ths.emplace_back(run_dev, d);
        for (auto &th : ths) th.join();
    }
    for (int d = 0; d < D; d++)
        if (!errs[(size_t)d].empty()) {
            xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", errs[(size_t)d].c_str());
            join_pool(h); h->failed = true;
            return -1;
        }
    join_pool(h);
    auto work = [h, batch, t, st, hmb, hlv, qps, D](int s) {
        const x264_param_t &p = h->param;
        const int G = h->G;
        const size_t row = (size_t)h->devs[(size_t)(s % D)].base + (size_t)(s / D);       // where slot s landed in the download buffers
        x264_t::Coded &c = h->slotbuf[(size_t)s * h->keyint + t];
        c.bytes.clear(); c.off.clear(); c.types.clear();
        c.idr = t == 0;
        const long gop = (long)batch * G + s;
        if (p.b_aud) { c.off.push_back(c.bytes.size()); c.types.push_back(9); write_aud(c.bytes, t == 0 ? 0 : 1, p.b_annexb != 0); }
        if (c.idr && p.b_repeat_headers) {
            const bool annexb = p.b_annexb != 0;
            c.off.push_back(c.bytes.size()); c.types.push_back(7); write_sps(c.bytes, make_sps(h), annexb);
            c.off.push_back(c.bytes.size()); c.types.push_back(8); write_pps(c.bytes, make_pps(h), annexb);
            if (gop == 0) { c.off.push_back(c.bytes.size()); c.types.push_back(6); write_sei_version(c.bytes, kSeiText, annexb); }
        }
        SliceParams sp = {};
        sp.mbw = h->mbw; sp.mbh = h->mbh; sp.slice_type = st; sp.qp = !qps.empty() ? (int)qps[(size_t)s] : c.idr ? h->qp_i : h->qp_p; sp.pic_init_qp = h->pic_init_qp;
        sp.frame_num = t & ((1 << h->log2_max_frame_num) - 1); sp.log2_max_frame_num = h->log2_max_frame_num;
        sp.idr = c.idr; sp.idr_pic_id = (int)(gop & 0xffff); sp.nal_ref_idc = c.idr ? 3 : 2; sp.pps_id = p.i_sps_id;
        sp.num_ref_default = p.i_frame_reference;
        sp.num_ref = t < p.i_frame_reference ? (t > 0 ? t : 1) : p.i_frame_reference;
        sp.disable_deblock_idc = p.b_deblocking_filter ? 0 : 1; sp.slices_plain = h->slices_plain;
        sp.alpha_off_div2 = p.i_deblocking_filter_alphac0; sp.beta_off_div2 = p.i_deblocking_filter_beta;
        sp.transform8x8_mode = p.analyse.b_transform_8x8; sp.cabac = p.b_cabac;
        {
            const size_t before = c.off.size();
            write_picture(c.bytes, &c.off, sp, h->slices, hmb + row * h->nmb, hlv + row * h->nmb * X264GPU_MB_LEVELS, p.b_annexb != 0, before == 0, nullptr);
            for (size_t i = before; i < c.off.size(); i++) c.types.push_back(c.idr ? 5 : 1);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthr = (int)(hw ? (hw < (unsigned)nslots_with_t ? hw : (unsigned)nslots_with_t) : 1);
    h->pool_t = t; h->pool_nslots = nslots_with_t;
    for (int th = 0; th < nthr; th++)
        h->pool.emplace_back([work, th, nthr, nslots_with_t]() { for (int s = th; s < nslots_with_t; s += nthr) work(s); });
    return 0;
}

// coded frames move to the ordered output queue once every earlier frame is there.  slotbuf is indexed inside a batch
// (slot * keyint + position); an entry is reused by the next batch only long after it was drained.
static void drain_coded(x264_t *h)
{
    const long K = h->keyint, per_batch = (long)h->G * K;
    for (long g = h->emitted + (long)h->ready.size(); g < h->submitted; g++) {
        const long j = g % per_batch;
        const size_t idx = (size_t)(j / K) * K + (size_t)(j % K);
        if (!h->slot_have[idx]) break;                       // not coded yet (or its CAVLC threads have not been joined)
        h->ready.push_back(std::move(h->slotbuf[idx]));
        h->slot_have[idx] = 0;
    }
}

static int encode_gop_parallel(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_in, x264_picture_t *pic_out, bool resident)
{
    const x264_param_t &p = h->param;
    const int G = h->G, K = h->keyint;
    const size_t insz = (size_t)p.i_width * p.i_height * 3 / 2;
    const long per_batch = (long)G * K;
    if (h->failed) return -1;
    if (pic_in) {
        if (h->flushed) { xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: pictures after a flush are not supported in GOP-parallel mode\n"); return -1; }
        const long i = h->submitted, b = i / per_batch, r = i % per_batch;
        const int s = (int)(r / K), t = (int)(r % K);
        // a new batch may only start gathering once the previous one is fully coded and drained into the output queue
        const int D = (int)h->devs.size();
        x264_t::DevCtx &dc = h->devs[(size_t)(s % D)];
        uint8_t *dst = dc.d_ring + ((size_t)t * dc.nsl + (size_t)(s / D)) * insz;
        const void *src = resident ? (const void *)h->d_in : (const void *)h->h_in.data();
        if ((resident ? x264gpu_memcpy_d2d(dst, src, insz, nullptr) : x264gpu_memcpy_h2d(dst, src, insz, nullptr)) != X264GPU_OK) {
            xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: upload failed: %s\n", x264gpu_last_error());
            return -1;
        }
        // the lookahead lives on the caller's device: with several devices the picture also goes to the staging buffer there
        const uint8_t *la_src = dst;
        if (h->crf && D > 1 && dc.dev != h->device) {
            if (!resident && x264gpu_memcpy_h2d(h->d_in, src, insz, nullptr) != X264GPU_OK) { xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: upload failed: %s\n", x264gpu_last_error()); return -1; }
            la_src = h->d_in;
        }
        if (h->crf) {
            // CRF: the picture's quantiser follows from the lookahead costs and the pictures before it, all known now (rc_pick_qp)
            int32_t costs[4];
            if (x264gpu_lookahead_frame_cost(h->la, la_src, i == 0, h->d_la, nullptr, nullptr) != X264GPU_OK ||
                x264gpu_memcpy_d2h(costs, h->d_la, sizeof(costs), nullptr) != X264GPU_OK) {
                xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: lookahead failed: %s\n", x264gpu_last_error());
                return -1;
            }
            h->gop_qp[(size_t)s * K + t] = (int8_t)rc_pick_qp(h, t == 0, costs, (int)i, (int)i);
            h->gop_qpm[(size_t)s * K + t] = near_qpm(h->rc.qpa_last, h->gop_qp[(size_t)s * K + t]);
        }
        h->pts.push_back(pic_in->i_pts);
        if (h->gopb) h->all_pts.push_back(pic_in->i_pts);
        h->submitted++;
        if (s == G - 1 && h->gopb) {
            // every slot holds display picture t now: the coding positions whose picture (and, being in coding order behind their closing picture,
            // whose references) have arrived
            if (t == 0) { h->gb_next = 0; }
            while (h->gb_next < K && h->gorder[(size_t)h->gb_next].first <= t) {
                if (code_position_b(h, (int)b, h->gb_next, 0, G, h->gorder, h->gdpb) < 0) return -1;
                h->gb_next++;
            }
            h->next_pos = t + 1 == K ? 0 : t + 1;
        } else
        if (s == G - 1) {                                      // the batch's last GOP delivers position t: every slot has it
            if (code_position(h, (int)b, t, G) < 0) return -1;
            h->next_pos = t + 1 == K ? 0 : t + 1;
        } else if (!h->pool.empty()) join_pool(h);             // gathering phase: the last position's CAVLC had a whole call to finish
        drain_coded(h);
    } else {
        // flush: code what the partly gathered batch holds, position by position, with the slots that have that position
        const long i = h->submitted, b = i == 0 ? 0 : (i - 1) / per_batch, r = i - b * per_batch;    // r frames in the last batch
        if (h->gopb && !h->flushed && r > 0 && !(r == per_batch && h->next_pos == 0)) {
            // the complete GOPs finish their plan in lock-step; then the stream's last, shorter GOP alone: its own coding order on a DPB model of its own,
            // from its IDR picture on (what the lock-step rounds coded of it while it was the batch's last slot is coded again: its tail differs)
            const int full = (int)(r / K), part = (int)(r % K);
            const bool last_ran = full == G - 1 && part > 0;          // (the partial GOP sat in the last slot: the rounds so far included it)
            if (!last_ran) h->gb_next = 0;                            // the last slot never delivered: nothing of this batch has been coded yet
            if (full > 0)
                for (int c = h->gb_next; c < K; c++)
                    if (code_position_b(h, (int)b, c, 0, full, h->gorder, h->gdpb) < 0) return -1;
            if (part > 0) {
                join_pool(h);
                x264_t::GopDpb gp;
                gp.dpb.configure(p.i_frame_reference, h->bframes, h->bpyramid, h->log2_max_frame_num, h->weightp);
                const std::vector<std::pair<int, int>> po = gop_coding_order(part, h->bframes, h->bpyramid);
                for (int c = 0; c < part; c++) { h->slot_have[(size_t)full * K + c] = 0; }
                for (int c = 0; c < part; c++)
                    if (code_position_b(h, (int)b, c, full, 1, po, gp) < 0) return -1;
            }
            h->gb_next = 0; h->next_pos = 0;
        } else
        if (!h->flushed && r > 0 && !(r == per_batch && h->next_pos == 0)) {
            const int full = (int)(r / K), part = (int)(r % K);            // `full` complete GOPs, then `part` frames
            for (int t = h->next_pos; t < K; t++) {
                const int nslots = full + (t < part ? 1 : 0);
                if (nslots <= 0) break;
                if (code_position(h, (int)b, t, nslots) < 0) return -1;
            }
            h->next_pos = 0;
        }
        h->flushed = true;
        join_pool(h);
        drain_coded(h);
    }
    if (h->ready.empty()) return 0;
    // ---- emit frame h->emitted ----
    x264_t::Coded c = std::move(h->ready.front());
    h->ready.pop_front();
    h->out = std::move(c.bytes);
    h->nal_off = c.off;
    publish_nals(h, pp_nal, pi_nal, c.types);
    if (h->gopb) for (size_t i = 0; i < h->nals.size(); i++) if (c.types[i] == 1 || c.types[i] == 5) h->nals[i].i_ref_idc = c.ref_idc;
    if (pic_out && h->gopb) {
        // coding order: the picture's own pts; x264's dts: the k-th coded picture takes the pts of display picture k - delay (the first ones shifted back)
        x264_picture_init(pic_out);
        pic_out->i_type = c.i_type; pic_out->b_keyframe = c.idr;
        const long k = h->emitted, gop_base = k - k % K, delay = !h->bframes ? 0 : h->bpyramid ? 2 : 1;
        const size_t np = h->all_pts.size();
        const size_t di = (size_t)(gop_base + c.disp);
        pic_out->i_pts = h->all_pts[di < np ? di : np - 1];
        if (k >= delay) pic_out->i_dts = h->all_pts[(size_t)(k - delay) < np ? (size_t)(k - delay) : np - 1];
        else pic_out->i_dts = h->all_pts[(size_t)k < np ? (size_t)k : np - 1] - (h->all_pts[(size_t)delay < np ? (size_t)delay : np - 1] - h->all_pts[0]);
    } else
    if (pic_out) {
        x264_picture_init(pic_out);
        pic_out->i_type = c.idr ? X264_TYPE_IDR : X264_TYPE_P;
        pic_out->b_keyframe = c.idr;
        pic_out->i_pts = h->pts.front(); pic_out->i_dts = h->pts.front();
    }
    h->pts.pop_front();
    h->emitted++;
    h->frame_no++;
    return (int)h->out.size();
}

// rate_estimate_qscale for the picture about to be coded (single-pass CRF / ABR): a function of the lookahead costs, the picture type and
// the running rate-control state only (ABR adds the coded sizes through x264_ratecontrol_end), so under CRF it can run when a picture
// ARRIVES — which is what lets GOP-parallel sessions keep CRF's quantisers.  frames_done = pictures decided before this one.
static int rc_pick_qp(x264_t *h, bool is_i, const int32_t costs[4], int frames_done, int frame)
{
    const x264_param_t &p = h->param;
    // rate_estimate_qscale: q = rceq / rate_factor; rceq = blurred_complexity^(1 - qcomp), or under macroblock-tree (which does the
    // complexity weighting itself) the frame-duration term alone; an I picture after P pictures takes the running P quantiser /
    // ipratio; the quantiser is qscale2qp(q) rounded, within [qpmin, qpmax]
    auto qp2qscale = [](double q) { return rc_qp2qscale(q); };
    auto qscale2qp = [](double qs) { return rc_qscale2qp(qs); };
    const double satd = is_i ? costs[0] : costs[1];
    h->rc.cplxsum = h->rc.cplxsum * 0.5 + satd / h->rc.dur_ratio;
    h->rc.cplxcount = h->rc.cplxcount * 0.5 + 1.0;
    double q, overflow = 1.0;
    const double rate_factor = h->crf ? h->rc.rate_factor_constant : h->rc.wanted_bits_window / h->rc.cplxr_sum;
    if (satd > 0) {
        h->rc.last_rceq = h->mbtree ? pow(1.0 / h->rc.dur_ratio, 1.0 - p.rc.f_qcompress) : pow(h->rc.cplxsum / h->rc.cplxcount, 1.0 - h->rc.qcompress);
        q = FL(h->rc.last_rceq / rate_factor);          // (rate_estimate_qscale's q is a float: every assignment rounds)
    } else q = FL(h->rc.last_qscale_for[is_i ? 0 : 1]);
    // get_qscale: a zone forces its quantiser or scales the picture's bits (an I picture after P pictures still takes the running P quantiser below, as in x264)
    if (const x264_t::Zone *z = get_zone(h, frame)) q = FL(z->force_qp ? qp2qscale(z->qp) : q / z->bitrate_factor);
    if (h->abr && satd > 0) {
        // pull towards the target: bits so far against time so far, within an abr_buffer that grows with sqrt(time)
        const double time_done = frames_done / h->rc.fps, wanted_bits = time_done * h->rc.bitrate;
        if (wanted_bits > 0) {
            const double buf = h->rc.abr_buffer * (time_done > 1.0 ? sqrt(time_done) : 1.0);
            overflow = 1.0 + (h->rc.total_bits - wanted_bits) / buf;
            overflow = overflow < 0.5 ? 0.5 : overflow > 2.0 ? 2.0 : overflow;
            q = FL(q * overflow);
        }
    }
    if (is_i && h->keyint > 1 && !h->rc.last_non_b_is_i) q = FL(qp2qscale(h->rc.accum_p_qp / h->rc.accum_p_norm) / h->rc.ip_factor);
    else if (frames_done > 0) {
        if (h->abr) {       // asymmetric clipping against the last quantiser of the same picture type (qpstep)
            double lmin = h->rc.last_qscale_for[is_i ? 0 : 1] / h->rc.lstep, lmax = h->rc.last_qscale_for[is_i ? 0 : 1] * h->rc.lstep;
            if (overflow > 1.1 && frames_done > 3) lmax *= h->rc.lstep;
            else if (overflow < 0.9) lmin /= h->rc.lstep;
            q = FL(q < lmin ? lmin : q > lmax ? lmax : q);
        }
    } else if (h->crf && h->rc.qcompress != 1.0) q = FL(qp2qscale(p.rc.f_rf_constant) / h->rc.ip_factor);       // very first picture: ABR_INIT_QP / ipratio
    q = FL(q < h->rc.lmin ? h->rc.lmin : q > h->rc.lmax ? h->rc.lmax : q);
    h->rc.last_qscale_for[is_i ? 0 : 1] = q;
    if (frames_done == 0) h->rc.last_qscale_for[1] = q * h->rc.ip_factor;
    double qpf = qscale2qp(q);
    qpf = qpf < p.rc.i_qp_min ? p.rc.i_qp_min : qpf > p.rc.i_qp_max ? p.rc.i_qp_max : qpf;
    const int qp_now = clampi((int)(qpf + 0.5), 1, 51);
    h->rc.accum_p_qp = h->rc.accum_p_qp * 0.95 + (is_i ? qpf + h->rc.ip_offset : qpf);      // accum_p_qp_update
    h->rc.accum_p_norm = h->rc.accum_p_norm * 0.95 + 1.0;
    h->rc.last_non_b_is_i = is_i;
    h->rc.qpa_last = qpf;
    return qp_now;
}

// GPU stage of queue[idx]: macroblock-tree over the pictures queued behind it (up to the next intra picture), rate control, the hot
// path and the download of its records / levels into host buffer pair `buf` — in a helper thread when `async` (join_gpu waits).
static void join_gpu(x264_t *h) { if (h->gpu_thread.joinable()) h->gpu_thread.join(); }

static int gpu_stage(x264_t *h, size_t idx, int buf, bool async)
{
    const x264_param_t &p = h->param;
    x264_t::QEntry &e = h->queue[idx];
    const bool idr = e.type == 2, intra_pic = e.type == 1, is_i = idr || intra_pic;
    int qp_now = is_i ? h->qp_i : h->qp_p;
    if (!h->crf && !h->abr && !h->zones.empty()) {          // constant quantiser with zones: every picture names its quantiser
        if (const x264_t::Zone *z = get_zone(h, h->rc_frames)) qp_now = cqp_zone(h, *z, qp_now);
        if (x264gpu_encoder_set_qp(h->gpu, qp_now, qp_now) != X264GPU_OK) return -1;
    }
    if (h->mbtree) {
        // macroblock_tree: this picture and the P pictures behind it that (transitively) reference it; an intra picture ends the chain
        const int32_t *info[256]; const float *aq[256];
        int n = 0;
        for (size_t j = idx; j < h->queue.size(); j++) {
            const x264_t::QEntry &q = h->queue[j];
            if (n > 0 && q.type != 0) break;
            info[n] = h->q_info[(size_t)q.slot]; aq[n] = h->q_aq[(size_t)q.slot];
            if (++n == 256) break;
        }
        if (x264gpu_lookahead_mbtree(h->la, info, h->aq_strength != 0.f ? aq : nullptr, n, h->tree_strength, h->d_tree, nullptr) != X264GPU_OK ||
            x264gpu_encoder_set_mb_qp_offsets(h->gpu, h->d_tree) != X264GPU_OK) {
            xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: macroblock-tree failed: %s\n", x264gpu_last_error());
            return -1;
        }
    }
    if (h->crf || h->abr) {
        qp_now = rc_pick_qp(h, is_i, e.costs, h->rc_frames, h->rc_frames);
        // x264_ratecontrol_mb_qp adds the AQ / macroblock-tree offsets to the FLOAT quantiser (rc->qpm) before the one rounding
        e.qpm = near_qpm(h->rc.qpa_last, qp_now);
        if (x264gpu_encoder_set_qp(h->gpu, qp_now, qp_now) != X264GPU_OK || x264gpu_encoder_set_qpm(h->gpu, e.qpm) != X264GPU_OK) return -1;
    }
    h->rc_frames++;
    e.qp = qp_now; e.buf = buf; e.launched = true;
    const int st = idr ? X264GPU_SLICE_I : intra_pic ? X264GPU_SLICE_I_NONIDR : X264GPU_SLICE_P;
    const uint8_t *src = h->q_raw[(size_t)e.slot];
    x264gpu_mb *hmb = buf ? h->h_mb2.data() : h->h_mb.data();
    int16_t *hlv = buf ? h->h_lv2.data() : h->h_lv.data();
    auto run = [h, src, st, hmb, hlv]() {
        h->gpu_rc = x264gpu_encode_frames(h->gpu, src, st, h->d_mb, h->d_lv, nullptr) != X264GPU_OK ||
                    x264gpu_memcpy_d2h(hmb, h->d_mb, h->h_mb.size() * sizeof(x264gpu_mb), nullptr) != X264GPU_OK ||
                    x264gpu_memcpy_d2h(hlv, h->d_lv, h->h_lv.size() * sizeof(int16_t), nullptr) != X264GPU_OK ? -1 : 0;
        if (h->gpu_rc) h->gpu_err = x264gpu_last_error();          // the error text lives in this thread's buffer: keep it for the caller's log
    };
    if (async) h->gpu_thread = std::thread([h, run]() { x264gpu_set_device(h->device); run(); });
    else run();
    return 0;
}

// Hands back the oldest picture of the lookahead queue: its GPU stage (unless a helper thread already ran it), then headers + entropy
// coding.  Returns the bytes of its NAL units, 0 when the pipelined session only started a picture.
static int encode_queued(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_out, bool flushing)
{
    const x264_param_t &p = h->param;
    if (!h->queue.front().launched) {
        if (gpu_stage(h, 0, 0, h->pipeline) < 0) return -1;
        if (h->pipeline && !flushing) return 0;              // its results are collected by the next call
    }
    join_gpu(h);
    if (h->gpu_rc) { xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", h->gpu_err.c_str()); return -1; }
    // pipelined: the next picture has its whole lookahead window (or the input has ended): start its GPU stage behind this one's coding
    if (h->pipeline && h->queue.size() >= 2 && (flushing || (int)h->queue.size() >= h->L + 2) &&
        gpu_stage(h, 1, h->queue.front().buf ^ 1, true) < 0) return -1;
    const x264_t::QEntry e = h->queue.front();
    const bool idr = e.type == 2, intra_pic = e.type == 1;
    const int qp_now = e.qp;
    const int st = idr ? X264GPU_SLICE_I : intra_pic ? X264GPU_SLICE_I_NONIDR : X264GPU_SLICE_P;
    const x264gpu_mb *hmb = e.buf ? h->h_mb2.data() : h->h_mb.data();
    const int16_t *hlv = e.buf ? h->h_lv2.data() : h->h_lv.data();
    h->last_scenecut = e.scenecut; h->last_qp = qp_now; h->last_qpm = e.qpm;
    memcpy(h->last_costs, e.costs, sizeof(e.costs));
    if (idr) { h->frames_since_idr = 0; h->frame_num = 0; }
    // ---- host: headers + entropy coding ----
    h->out.clear(); h->nal_off.clear();
    std::vector<int> types;
    if (p.b_aud) { h->nal_off.push_back(h->out.size()); types.push_back(9); write_aud(h->out, st != X264GPU_SLICE_P ? 0 : 1, p.b_annexb != 0); }
    if (idr && p.b_repeat_headers) {
        emit_sets(h, types, !h->sei_sent);
        h->sei_sent = 1;
    }
    SliceParams sp = {};
    sp.mbw = h->mbw; sp.mbh = h->mbh; sp.slice_type = st == X264GPU_SLICE_P ? X264GPU_SLICE_P : X264GPU_SLICE_I; sp.qp = qp_now; sp.pic_init_qp = h->pic_init_qp;
    sp.frame_num = h->frame_num; sp.log2_max_frame_num = h->log2_max_frame_num;
    sp.idr = idr; sp.idr_pic_id = h->idr_pic_id; sp.nal_ref_idc = idr ? 3 : 2; sp.pps_id = p.i_sps_id;
    sp.num_ref_default = p.i_frame_reference;
    sp.num_ref = h->frames_since_idr < p.i_frame_reference ? (h->frames_since_idr > 0 ? h->frames_since_idr : 1) : p.i_frame_reference;
    sp.disable_deblock_idc = p.b_deblocking_filter ? 0 : 1; sp.slices_plain = h->slices_plain;
    sp.alpha_off_div2 = p.i_deblocking_filter_alphac0; sp.beta_off_div2 = p.i_deblocking_filter_beta;
    sp.transform8x8_mode = p.analyse.b_transform_8x8; sp.cabac = p.b_cabac;
    h->last_stats.skip = 0;
    {
        const size_t before = h->nal_off.size();
        write_picture(h->out, &h->nal_off, sp, h->slices, hmb, hlv, p.b_annexb != 0, before == 0, &h->last_stats, h->cavlc_threads);
        for (size_t i = before; i < h->nal_off.size(); i++) types.push_back(idr ? 5 : 1);
    }
    publish_nals(h, pp_nal, pi_nal, types);
    if (pic_out) {
        x264_picture_init(pic_out);
        pic_out->i_type = idr ? X264_TYPE_IDR : intra_pic ? X264_TYPE_I : X264_TYPE_P;
        pic_out->b_keyframe = idr;
        pic_out->i_pts = e.pts; pic_out->i_dts = e.pts;
        pic_out->img = e.img;
    }
    if (idr) h->idr_pic_id = (h->idr_pic_id + 1) & 0xffff;
    h->frame_num = (h->frame_num + 1) & ((1 << h->log2_max_frame_num) - 1);
    if (h->abr) {
        // x264_ratecontrol_end: what this picture's bits say about the rate factor, and the bits the window now expects
        const double bits = 8.0 * (double)h->out.size();
        h->rc.total_bits += bits;
        h->rc.cplxr_sum += bits * rc_qp2qscale(rc_qp_avg_rc((float)h->rc.qpa_last, h->mbw, h->mbh)) / h->rc.last_rceq;          // (rc->qpa_rc: the float gathered row by row)
        h->rc.wanted_bits_window += h->rc.bitrate / h->rc.fps;
    }
    h->frames_since_idr++;
    h->frame_no++;
    h->queue.pop_front();
    return (int)h->out.size();
}


// ---- sessions with B pictures ---------------------------------------------------------------------------------------------------
// x264_slicetype_decide with --b-adapt 0 over the display-order queue: the pictures up to the next non-B picture become one mini-GOP.
// A forced I / IDR picture closes the run in front of it (the picture before an IDR becomes P: closed GOPs); otherwise the run is
// `bframes` long, or what is left when the input ends (the last picture is never B).  Returns false while more input is needed.
// ---- x264_slicetype_analyse / x264_slicetype_decide on the device's frame costs (sessions whose slicetype object exists: scenecut or
//      --b-adapt 1).  frames[0] = the last non-B picture, frames[1..] = the pictures waiting in display order ----
enum { ST_AUTO = 0, ST_IDR, ST_I, ST_P, ST_BREF, ST_B };
struct StFrames { x264_t *h; std::vector<x264_t::BEntry *> f; };
// x264_weights_analyse: guess scale and offset of each plane from the two pictures' statistics, cost the candidates around the guess — luma on the
// half-resolution planes (per 8x8 block min(mbcmp, intra cost)), the chroma planes at full resolution on the blocks' DC differences — keep a weight
// if it saves more than 0.2 %.  b_lookahead: luma alone, the guess alone, reference in place (called before a P cost is searched); else, for the P
// picture about to be coded: +- the distances of the sub-pel level around the guess, the reference motion-compensated by the lookahead's vectors,
// and the chroma planes once luma has a weight.
static Dpb::LumaWeight st_weights_analyse(x264_t *h, x264_t::BEntry &fenc, const x264_t::BEntry &ref, int dist, bool b_lookahead)
{
    Dpb::LumaWeight none, w;
    if (dist >= 1 && dist <= 18) fenc.weighted_cost_delta[dist - 1] = 0;
    uint64_t sf[6], sr[6];
    if (x264gpu_slicetype_pixel_stats(h->st, fenc.slot, h->q_raw[(size_t)fenc.slot], sf, nullptr) != X264GPU_OK ||
        x264gpu_slicetype_pixel_stats(h->st, ref.slot, h->q_raw[(size_t)ref.slot], sr, nullptr) != X264GPU_OK) { h->failed = true; return none; }
    const int nplanes = b_lookahead ? 1 : 3;
    if (!b_lookahead && (x264gpu_slicetype_chroma_stats(h->st, fenc.slot, h->q_raw[(size_t)fenc.slot], sf + 2, nullptr) != X264GPU_OK ||
                         x264gpu_slicetype_chroma_stats(h->st, ref.slot, h->q_raw[(size_t)ref.slot], sr + 2, nullptr) != X264GPU_OK)) { h->failed = true; return none; }
    const float epsilon = 1.f / 128.f;
    float guess_scale[3] = { 1, 1, 1 }, fenc_mean[3] = { 0, 0, 0 }, ref_mean[3] = { 0, 0, 0 };
    for (int plane = 0; plane < nplanes; plane++) {
        const int zero_bias = !sr[2 * plane + 1];
        const float fenc_var = (float)(sf[2 * plane + 1] + (uint64_t)zero_bias), ref_var = (float)(sr[2 * plane + 1] + (uint64_t)zero_bias);
        guess_scale[plane] = sqrtf(fenc_var / ref_var);
        const float npix = plane ? (float)(h->mbw * 8) * (float)(h->mbh * 8) : (float)(h->mbw * 16) * (float)(h->mbh * 16);
        fenc_mean[plane] = (float)(sf[2 * plane] + (uint64_t)zero_bias) / npix; ref_mean[plane] = (float)(sr[2 * plane] + (uint64_t)zero_bias) / npix;
    }
    int chroma_denom = 7;
    if (!b_lookahead)          // make sure both chroma scale factors fit
        while (chroma_denom > 0) {
            const float thresh = 127.f / (1 << chroma_denom);
            if (guess_scale[1] < thresh && guess_scale[2] < thresh) break;
            chroma_denom--;
        }
    static const uint8_t check_distance[12][2] = { { 0, 0 }, { 0, 0 }, { 0, 1 }, { 0, 1 }, { 0, 1 }, { 0, 1 }, { 0, 1 }, { 1, 1 }, { 1, 1 }, { 2, 1 }, { 2, 1 }, { 4, 2 } };
    const int sub = clampi(h->param.analyse.i_subpel_refine, 0, 11);
    const int scale_dist = b_lookahead ? 0 : check_distance[sub][0], offset_dist = b_lookahead ? 0 : check_distance[sub][1];
    bool planes_on[3] = { false, false, false };
    int p_scale[3] = { 1, 1, 1 }, p_denom[3] = { 0, 0, 0 }, p_off[3] = { 0, 0, 0 };
    // (the chroma planes are not checked in the lookahead, or if there was no luma weight)
    for (int plane = 0; plane < nplanes && !(plane && !planes_on[0]); plane++) {
        if (fabsf(ref_mean[plane] - fenc_mean[plane]) < 0.5f && fabsf(1.f - guess_scale[plane]) < epsilon) continue;      // early termination
        int mindenom, minscale, minoff = 0;
        if (plane) {
            mindenom = chroma_denom;
            minscale = clampi((int)roundf(guess_scale[plane] * (1 << chroma_denom)), 0, 255);
            if (minscale > 127) { planes_on[1] = planes_on[2] = false; break; }
        } else {
            // weight_get_h264( round( guess_scale * 128 ), 0 )
            mindenom = 7; minscale = (int)roundf(guess_scale[0] * 128);
            while (mindenom > 0 && minscale > 127) { mindenom--; minscale >>= 1; }
            if (minscale > 127) minscale = 127;
        }
        auto cost_of = [&](int on, int scale, int denom, int offset, int64_t &score) {
            if (!plane) return x264gpu_slicetype_weight_cost(h->st, fenc.slot, ref.slot, dist, on, scale, denom, offset, &score, nullptr) == X264GPU_OK;
            return x264gpu_slicetype_weight_cost_chroma(h->st, fenc.slot, h->q_raw[(size_t)fenc.slot], h->q_raw[(size_t)ref.slot], dist, plane, on, scale, denom, offset, &score, nullptr) == X264GPU_OK;
        };
        int32_t dummy = 0;
        int64_t score = 0;
        if ((!plane && x264gpu_slicetype_frame_cost(h->st, fenc.slot, fenc.slot, fenc.slot, 0, 0, &dummy, nullptr) != X264GPU_OK) ||       // the picture's intra costs
            !cost_of(0, 1, 0, 0, score)) { h->failed = true; return none; }
        const unsigned origscore = (unsigned)score;
        unsigned minscore = origscore;
        if (!minscore) continue;
        const int start_scale = clampi(minscale - scale_dist, 0, 127), end_scale = clampi(minscale + scale_dist, 0, 127);
        bool found = false;
        for (int i_scale = start_scale; i_scale <= end_scale; i_scale++) {
            int cur_scale = i_scale;
            int cur_offset = (int)(fenc_mean[plane] - ref_mean[plane] * cur_scale / (1 << mindenom) + 0.5f * b_lookahead);
            if (cur_offset < -128 || cur_offset > 127) {
                cur_offset = clampi(cur_offset, -128, 127);
                float cs = (1 << mindenom) * (fenc_mean[plane] - cur_offset) / ref_mean[plane] + 0.5f;
                cur_scale = (int)(cs < 0 ? 0 : cs > 127 ? 127 : cs);
            }
            const int start_offset = clampi(cur_offset - offset_dist, -128, 127), end_offset = clampi(cur_offset + offset_dist, -128, 127);
            for (int i_off = start_offset; i_off <= end_offset; i_off++) {
                if (!cost_of(1, cur_scale, mindenom, i_off, score)) { h->failed = true; return none; }
                if ((unsigned)score < minscore) { minscore = (unsigned)score; minscale = cur_scale; minoff = i_off; found = true; }
                if (minoff == start_offset && i_off != start_offset) break;          // the previous offset was better: no more
            }
        }
        if (!plane) while (mindenom > 0 && !(minscale & 1)) { mindenom--; minscale >>= 1; }      // a smaller denominator if possible
        if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f) continue;
        planes_on[plane] = true; p_scale[plane] = minscale; p_denom[plane] = mindenom; p_off[plane] = minoff;
        if (h->weightp_fake && !plane && dist >= 1 && dist <= 18) fenc.weighted_cost_delta[dist - 1] = (float)minscore / origscore;
    }
    if (!planes_on[0]) return none;           // (x264 keeps chroma weights only beside a luma weight: they are not even analysed without one)
    w.on = 1; w.scale = p_scale[0]; w.denom = p_denom[0]; w.offset = p_off[0];
    if (planes_on[1] || planes_on[2]) {
        // optimise and unify the chroma denominator: a plane weighted alone leaves the other with the implicit scale 1 << denom, which 7 cannot carry
        int denom = planes_on[1] ? p_denom[1] : p_denom[2];
        const bool both = planes_on[1] && planes_on[2];
        while ((!both && denom == 7) || (denom > 0 && !(planes_on[1] && (p_scale[1] & 1)) && !(planes_on[2] && (p_scale[2] & 1)))) {
            denom--;
            for (int i = 1; i <= 2; i++) if (planes_on[i]) { p_scale[i] >>= 1; p_denom[i] = denom; }
        }
        w.cdenom = denom;
        for (int c = 0; c < 2; c++) if (planes_on[c + 1]) { w.con[c] = 1; w.cscale[c] = p_scale[c + 1]; w.coffset[c] = p_off[c + 1]; }
    }
    return w;
}

static int st_cost(StFrames &F, int p0, int p1, int b)
{
    int32_t sc = 0;
    Dpb::LumaWeight w;
    // slicetype_frame_cost: a P cost that is searched for the first time runs on the reference weighted by the lookahead's analysis
    if ((F.h->weightp || F.h->weightp_fake) && p1 == b && b != p0 && !x264gpu_slicetype_lowres_mvs(F.h->st, F.f[(size_t)b]->slot, 0, b - p0) &&
        x264gpu_slicetype_cost_est(F.h->st, F.f[(size_t)b]->slot, b - p0, 0, 0) < 0)
        w = st_weights_analyse(F.h, *F.f[(size_t)b], *F.f[(size_t)p0], b - p0, true);
    if (x264gpu_slicetype_frame_cost_w(F.h->st, F.f[(size_t)p0]->slot, F.f[(size_t)p1]->slot, F.f[(size_t)b]->slot, b - p0, p1 - b, w.on, w.scale, w.denom, w.offset, &sc, nullptr) != X264GPU_OK) {
        xlog(&F.h->param, X264_LOG_ERROR, "lookahead frame cost failed: %s\n", x264gpu_last_error());
        F.h->failed = true;
    }
    return sc;
}
// scenecut_internal: P cost against I cost of frames[p1], the bias growing with the distance from the last keyframe
static bool st_scenecut_internal(StFrames &F, int p0, int p1)
{
    x264_t *h = F.h;
    const x264_param_t &p = h->param;
    st_cost(F, p0, p1, p1);
    const x264_t::BEntry *fr = F.f[(size_t)p1];
    const int icost = x264gpu_slicetype_cost_est(h->st, fr->slot, 0, 0, 0), pcost = x264gpu_slicetype_cost_est(h->st, fr->slot, p1 - p0, 0, 0);
    const int gop = fr->frame - h->last_keyframe;
    const float tmax = (float)(p.i_scenecut_threshold / 100.0);
    float tmin = (float)(tmax * 0.25), bias;
    if (p.i_keyint_min == p.i_keyint_max) tmin = tmax;
    if (gop <= p.i_keyint_min / 4) bias = tmin / 4;
    else if (gop <= p.i_keyint_min) bias = tmin * gop / p.i_keyint_min;
    else bias = tmin + (tmax - tmin) * (gop - p.i_keyint_min) / (p.i_keyint_max - p.i_keyint_min);
    return pcost >= (1.0 - bias) * icost;
}
// scenecut: with B pictures a short flash between two scenes must not become a keyframe (x264 looks one picture past p1 under --b-adapt 1)
static bool st_scenecut(StFrames &F, int p0, int p1, bool real, int num_frames, int i_max_search)
{
    x264_t *h = F.h;
    if (real && h->bframes) {
        const int origmaxp1 = p0 + 1 + (h->badapt == 2 ? h->bframes : 1), maxp1 = origmaxp1 < num_frames ? origmaxp1 : num_frames;      // the trellis may put bframes pictures between p0 and p1
        for (int curp1 = p1; curp1 <= maxp1; curp1++)
            if (!st_scenecut_internal(F, p0, curp1))
                for (int i = curp1; i > p0; i--) F.f[(size_t)i]->b_scenecut = 0;          // nothing between p0 and curp1 can be a real scene cut
        for (int curp0 = p0; curp0 <= maxp1; curp0++)
            if (origmaxp1 > i_max_search || (curp0 < maxp1 && st_scenecut_internal(F, curp0, maxp1)))
                F.f[(size_t)curp0]->b_scenecut = 0;                                     // the p0 of a scene cut cannot be the p1 of one
    }
    if (!F.f[(size_t)p1]->b_scenecut) return false;
    return st_scenecut_internal(F, p0, p1);
}
// x264's macroblock_tree over frames[0 .. num_frames] with the types decided so far (tests/mbtree_walk.py is the same walk): every picture hands
// the cost its references explain back to them, last picture first; the next picture to be coded (and the B-reference of its run) get their
// quantiser offsets.  b_intra: the pass x264 runs for a keyframe after it was decided (frames[0] = that keyframe).
static void st_macroblock_tree(x264_t *h, StFrames &F, int num_frames, bool b_intra)
{
    const int idx = b_intra ? 0 : 1;
    auto isb = [&](int i) { return F.f[(size_t)i]->type == ST_B || F.f[(size_t)i]->type == ST_BREF; };
    auto slot = [&](int i) { return F.f[(size_t)i]->slot; };
    auto prop = [&](int p0, int p1, int b, int referenced) {
        if (x264gpu_slicetype_propagate(h->st, slot(p0), slot(p1), slot(b), b - p0, p1 - b, referenced, nullptr) != X264GPU_OK) {
            xlog(&h->param, X264_LOG_ERROR, "macroblock-tree failed: %s\n", x264gpu_last_error());
            h->failed = true;
        }
    };
    auto clear = [&](int i) { if (x264gpu_slicetype_clear_propagate(h->st, slot(i), nullptr) != X264GPU_OK) h->failed = true; };
    auto finish = [&](int i, int ref0_distance) {
        st_cost(F, i, i, i);          // (the intra costs the analysis left with the picture; a no-op when they exist)
        // macroblock_tree_finish: a fade the (fake) weight analysis explained is not held against the picture
        float weightdelta = 0.0;
        if (ref0_distance >= 1 && ref0_distance <= 18 && F.f[(size_t)i]->weighted_cost_delta[ref0_distance - 1] > 0) weightdelta = (float)(1.0 - F.f[(size_t)i]->weighted_cost_delta[ref0_distance - 1]);
        if (x264gpu_slicetype_finish(h->st, slot(i), h->tree_strength, weightdelta, h->q_tree[(size_t)slot(i)], nullptr) != X264GPU_OK) h->failed = true;
    };
    if (b_intra) st_cost(F, 0, 0, 0);
    int i = num_frames;
    while (i > 0 && isb(i)) i--;
    int last_nonb = i, bframes = 0;
    if (last_nonb < idx) return;
    clear(last_nonb);
    while (i-- > idx) {
        int cur_nonb = i;
        while (isb(cur_nonb) && cur_nonb > 0) cur_nonb--;
        if (cur_nonb < idx) break;
        // (distances beyond bframes + 1 cannot occur: the analysis never leaves longer runs)
        st_cost(F, cur_nonb, last_nonb, last_nonb);
        clear(cur_nonb);
        bframes = last_nonb - cur_nonb - 1;
        if (h->bpyramid && bframes > 1) {
            const int middle = (bframes + 1) / 2 + cur_nonb;
            st_cost(F, cur_nonb, last_nonb, middle);
            clear(middle);
            while (i > cur_nonb) {
                const int p0 = i > middle ? middle : cur_nonb, p1 = i < middle ? middle : last_nonb;
                if (i != middle) { st_cost(F, p0, p1, i); prop(p0, p1, i, 0); }
                i--;
            }
            prop(cur_nonb, last_nonb, middle, 1);
        } else
            while (i > cur_nonb) { st_cost(F, cur_nonb, last_nonb, i); prop(cur_nonb, last_nonb, i, 0); i--; }
        prop(cur_nonb, last_nonb, last_nonb, 1);
        last_nonb = cur_nonb;
        if (h->failed) return;
    }
    finish(last_nonb, last_nonb);
    if (h->bpyramid && bframes > 1) finish(last_nonb + (bframes + 1) / 2, 0);
}

// x264 slicetype_path_cost: the cost of coding frames[1 ..] with the types in `path` ('P' / 'B' / 'I' per picture) — each non-B picture against the one
// before it, the B pictures between them against both (through the middle one under b-pyramid); stops early beyond `threshold`
static uint64_t st_path_cost(StFrames &F, const char *path0, uint64_t threshold)
{
    x264_t *h = F.h;
    uint64_t cost = 0;
    int loc = 1, cur_nonb = 0;
    const char *path = path0 - 1;          // the first path element is the second frame
    while (path[loc]) {
        int next_nonb = loc;
        while (path[next_nonb] == 'B') next_nonb++;
        cost += path[next_nonb] == 'P' ? st_cost(F, cur_nonb, next_nonb, next_nonb) : st_cost(F, next_nonb, next_nonb, next_nonb);
        if (cost > threshold || h->failed) break;
        if (h->bpyramid && next_nonb - cur_nonb > 2) {
            const int middle = cur_nonb + (next_nonb - cur_nonb) / 2;
            cost += st_cost(F, cur_nonb, next_nonb, middle);
            for (int next_b = loc; next_b < middle && cost < threshold; next_b++) cost += st_cost(F, cur_nonb, middle, next_b);
            for (int next_b = middle + 1; next_b < next_nonb && cost < threshold; next_b++) cost += st_cost(F, middle, next_nonb, next_b);
        } else
            for (int next_b = loc; next_b < next_nonb && cost < threshold; next_b++) cost += st_cost(F, cur_nonb, next_nonb, next_b);
        loc = next_nonb + 1;
        cur_nonb = next_nonb;
    }
    return cost;
}

// x264 slicetype_path (--b-adapt 2): the best way to code the first `length` pictures ends in 0 .. bframes B pictures and a P picture behind the
// best way to code the pictures in front of them (Viterbi over the lengths; best_paths is indexed by length modulo 17)
constexpr int ST_PATH_MAX = 96;
static void st_path(StFrames &F, int length, char (*best_paths)[ST_PATH_MAX + 1])
{
    x264_t *h = F.h;
    char paths[2][ST_PATH_MAX + 1];
    const int num_paths = h->bframes + 1 < length ? h->bframes + 1 : length;
    uint64_t best_cost = ~0ull >> 1;
    int best_possible = 0, idx = 0;
    memset(paths, 0, sizeof(paths));
    for (int path = 0; path < num_paths; path++) {
        const int len = length - (path + 1);
        memcpy(paths[idx], best_paths[len % 17], (size_t)len);
        memset(paths[idx] + len, 'B', (size_t)path);
        paths[idx][len + path] = 'P'; paths[idx][len + path + 1] = 0;
        int possible = 1;
        for (int i = 1; i <= length; i++) {
            const int t = F.f[(size_t)i]->type;
            if (t == ST_AUTO) continue;
            if (t == ST_B || t == ST_BREF) possible = possible && (i < len || i == length || paths[idx][i - 1] == 'B');
            else {
                possible = possible && (i < len || paths[idx][i - 1] != 'B');
                paths[idx][i - 1] = t == ST_I || t == ST_IDR ? 'I' : 'P';
            }
        }
        if (possible || !best_possible) {
            if (possible && !best_possible) best_cost = ~0ull >> 1;
            const uint64_t cost = st_path_cost(F, paths[idx], best_cost);
            if (cost < best_cost) { best_cost = cost; best_possible = possible; idx ^= 1; }
        }
    }
    memcpy(best_paths[length % 17], paths[idx ^ 1], (size_t)length);
    best_paths[length % 17][length] = 0;
}

static void st_analyse(x264_t *h, StFrames &F, int framecnt, bool keyframe = false)
{
    const x264_param_t &p = h->param;
    auto type = [&](int i) -> int & { return F.f[(size_t)i]->type; };
    auto forced = [&](int i) { const int f = F.f[(size_t)i]->forced; return f == 2 ? ST_IDR : f == 1 ? ST_I : ST_AUTO; };
    auto auto_or_i = [](int t) { return t == ST_AUTO || t == ST_I || t == ST_IDR; };
    const int i_max_search = framecnt;
    if (!framecnt) return;
    const int keyint_limit = p.i_keyint_max - F.f[0]->frame + h->last_keyframe - 1;
    int num_frames = framecnt < keyint_limit ? framecnt : keyint_limit;
    const int orig_num_frames = num_frames;
    if (p.analyse.b_psy && h->mbtree) num_frames = framecnt;           // psy-wise the pictures before a keyframe must not lose their share of the tree
    else if (num_frames <= 0) { type(1) = ST_I; return; }
    // a picture whose type the caller forced ends the window in front of it (x264 warns and overrides; here the analysis stops short)
    for (int j = 2; j <= num_frames; j++) if (forced(j) != ST_AUTO) { num_frames = j - 1; break; }
    if (!keyframe && auto_or_i(type(1)) && p.i_scenecut_threshold && st_scenecut(F, 0, 1, true, orig_num_frames, i_max_search)) {
        if (type(1) == ST_AUTO) type(1) = ST_I;
        return;
    }
    int num_bframes = 0, reset_start, num_analysed = num_frames;
    if (h->bframes) {
        if (h->badapt == 2) {
            if (num_frames > ST_PATH_MAX) num_frames = ST_PATH_MAX;
            if (num_frames > 1) {
                static thread_local char best_paths[17][ST_PATH_MAX + 1];
                memset(best_paths, 0, sizeof(best_paths));
                best_paths[1][0] = 'P';
                const int best_path_index = num_frames % 17;
                for (int j = 2; j <= num_frames && !h->failed; j++) st_path(F, j, best_paths);
                if (h->failed) return;
                for (int j = 1; j < num_frames; j++) {
                    if (best_paths[best_path_index][j - 1] != 'B') { if (type(j) == ST_AUTO || type(j) == ST_B || type(j) == ST_BREF) type(j) = ST_P; }
                    else if (type(j) == ST_AUTO) type(j) = ST_B;
                }
            }
            if (type(num_frames) == ST_AUTO || type(num_frames) == ST_B || type(num_frames) == ST_BREF) type(num_frames) = ST_P;
            while (num_bframes < num_frames && type(num_bframes + 1) == ST_B) num_bframes++;
        } else if (h->badapt == 1) {
            // X264_B_ADAPT_FAST as the x264 generation this host restates has it (the one whose trellis loader and scene-cut loop know forced types): picture j becomes
            // a B picture when the path "..BP" from the last non-B picture costs less than "..PP" (slicetype_path_cost on both), runs no longer than --bframes.
            // (Older x264 compared pairwise frame costs against thresholds — INTER_THRESH / P_SENS_BIAS; which of the two the driver's core 157 carries cannot be
            // checked here: DESIGN.md §0.)
            auto isb = [&](int i) { return type(i) == ST_B || type(i) == ST_BREF; };
            int last_nonb = 0, num_bf = h->bframes;
            char path[ST_PATH_MAX + 4];
            for (int j = 1; j < num_frames && !h->failed; j++) {
                if (j - 1 > 0 && isb(j - 1)) num_bf--;
                else { last_nonb = j - 1; num_bf = h->bframes; }
                if (!num_bf) { if (type(j) == ST_AUTO || isb(j)) type(j) = ST_P; continue; }
                if (type(j) != ST_AUTO) continue;
                if (isb(j + 1)) { type(j) = ST_P; continue; }
                const int bfr = j - last_nonb - 1;
                StFrames sub;
                sub.h = h;
                sub.f.assign(F.f.begin() + last_nonb, F.f.end());
                memset(path, 'B', (size_t)bfr);
                strcpy(path + bfr, "PP");
                const uint64_t cost_p = st_path_cost(sub, path, ~0ull >> 1);
                strcpy(path + bfr, "BP");
                const uint64_t cost_b = st_path_cost(sub, path, cost_p);
                type(j) = cost_b < cost_p ? ST_B : ST_P;
            }
            if (h->failed) return;
            if (type(num_frames) == ST_AUTO || type(num_frames) == ST_B || type(num_frames) == ST_BREF) type(num_frames) = ST_P;
            while (num_bframes < num_frames && type(num_bframes + 1) == ST_B) num_bframes++;
        } else {
            num_bframes = num_frames - 1 < h->bframes ? num_frames - 1 : h->bframes;
            for (int j = 1; j < num_frames; j++) type(j) = (j % (num_bframes + 1)) ? ST_B : ST_P;
            type(num_frames) = ST_P;
        }
        // scene cut inside the first mini-GOP: the picture in front of it closes the run
        for (int j = 1; j < num_bframes + 1; j++)
            if (forced(j) == ST_AUTO && auto_or_i(forced(j + 1)) && p.i_scenecut_threshold && st_scenecut(F, j, j + 1, false, orig_num_frames, i_max_search)) {
                type(j) = ST_P;
                num_analysed = j;
                break;
            }
        reset_start = keyframe ? 1 : num_bframes + 2 < num_analysed + 1 ? num_bframes + 2 : num_analysed + 1;
    } else {
        for (int j = 1; j <= num_frames; j++) if (auto_or_i(forced(j))) type(j) = ST_P;
        reset_start = keyframe ? 1 : 2;
    }
    // the macroblock-tree over the window, no farther than a keyframe interval
    if (h->mbtree) st_macroblock_tree(h, F, num_frames < p.i_keyint_max ? num_frames : p.i_keyint_max, keyframe);
    if (h->failed) return;
    // enforce the keyframe limit
    {
        int last_keyframe = h->last_keyframe, last_possible = 0;
        for (int j = 1; j <= num_frames; j++) {
            int kd = F.f[(size_t)j]->frame - last_keyframe;
            if (auto_or_i(forced(j))) last_possible = j;
            if (kd >= p.i_keyint_max) {
                if (last_possible != 0 && last_possible != j) { j = last_possible; kd = F.f[(size_t)j]->frame - last_keyframe; }
                last_possible = 0;
                if (type(j) != ST_IDR) type(j) = ST_IDR;
            }
            if (type(j) == ST_I && kd >= p.i_keyint_min) type(j) = ST_IDR;
            if (type(j) == ST_IDR) { last_keyframe = F.f[(size_t)j]->frame; if (j > 1 && (type(j - 1) == ST_B || type(j - 1) == ST_BREF)) type(j - 1) = ST_P; }
        }
    }
    // the pictures behind the first mini-GOP are decided again when their turn comes
    for (int j = reset_start; j <= framecnt; j++) type(j) = forced(j);
}

// x264_slicetype_decide: -> index of the picture that closes the first mini-GOP of the queue and its type (PIC_*)
static bool st_decide(x264_t *h, bool flushing, int &j_out, int &closing_out)
{
    const x264_param_t &p = h->param;
    const int n = (int)h->bq.size();
    if (!flushing && n <= h->st_wait) return false;
    for (auto &e : h->bq) e.type = e.forced == 2 ? ST_IDR : e.forced == 1 ? ST_I : ST_AUTO;
    if (h->pass2) {
        // x264_ratecontrol_slice_type: the second pass codes every picture as the type the first pass gave it (the B-reference of a run is placed by
        // the same rule in both passes)
        for (auto &e : h->bq) {
            if (e.frame >= (int)h->p2.size()) continue;
            const char t = h->p2[(size_t)e.frame].type;
            e.type = t == 'I' ? ST_IDR : t == 'i' ? ST_I : t == 'P' ? ST_P : ST_B;
        }
    } else
    if (h->have_last_nonb && ((h->bframes && h->badapt) || p.i_scenecut_threshold || h->mbtree)) {
        StFrames F;
        F.h = h;
        F.f.push_back(&h->last_nonb);
        const int framecnt = n < h->st_wait + 1 ? n : h->st_wait + 1;          // what the lookahead holds for sure (deterministic mode), except at the end
        for (int i = 0; i < framecnt; i++) F.f.push_back(&h->bq[(size_t)i]);
        st_analyse(h, F, framecnt);
        if (h->failed) return false;
    }
    int bfr;
    for (bfr = 0;; bfr++) {
        x264_t::BEntry &frm = h->bq[(size_t)bfr];
        if (frm.frame - h->last_keyframe >= p.i_keyint_max) frm.type = ST_IDR;              // limit the GOP size
        if (frm.type == ST_I && frm.frame - h->last_keyframe >= p.i_keyint_min) frm.type = ST_IDR;
        if (frm.type == ST_IDR) {                                                          // close the GOP
            h->last_keyframe = frm.frame;
            // x264 keeps i_type on the frame; here the queue's types are re-derived from `forced` on every call, so the decision is pinned
            // there: the IDR stays an IDR when it is reached after the run in front of it (which closes as P) has been coded
            if (bfr > 0) { frm.forced = 2; bfr--; h->bq[(size_t)bfr].type = ST_P; }
        }
        if (bfr == h->bframes || bfr + 1 >= n) { if (frm.type == ST_AUTO || frm.type == ST_B || frm.type == ST_BREF) frm.type = ST_P; }
        if (frm.type == ST_AUTO) frm.type = ST_B;
        else if (frm.type != ST_B && frm.type != ST_BREF) break;
    }
    const int t = h->bq[(size_t)bfr].type;
    j_out = bfr; closing_out = t == ST_IDR ? PIC_IDR : t == ST_I ? PIC_I : PIC_P;
    return true;
}

static bool bmode_decide(x264_t *h, bool flushing)
{
    if (!h->bcoding.empty() || h->bq.empty()) return !h->bcoding.empty();
    const int n = (int)h->bq.size();
    if (!flushing && n <= (h->st ? h->st_wait : h->bframes)) return false;          // the lookahead x264 keeps in front of the slice-type decision
    int j = -1;                                       // index of the closing picture
    if (h->st) {
        int closing = PIC_P;
        if (!st_decide(h, flushing, j, closing)) return false;
        if (h->weightp && closing == PIC_P && h->have_last_nonb) {
            // x264_slicetype_decide: "analyse for weighted P frames" — the picture about to be coded against the last non-B picture
            h->bq[(size_t)j].w = st_weights_analyse(h, h->bq[(size_t)j], h->last_nonb, j + 1, false);
            if (h->failed) return false;
        }
        if (h->crf || h->abr) {
            // x264_rc_analyse_slice: the closing picture's complexity is its frame cost as the type it was given — the I cost, or the P cost
            // against the last non-B picture (distance = run length + 1), from the lookahead that decided the types
            x264_t::BEntry &c = h->bq[(size_t)j];
            int32_t ic = 0, pc = 0;
            bool ok = x264gpu_slicetype_frame_cost(h->st, c.slot, c.slot, c.slot, 0, 0, &ic, nullptr) == X264GPU_OK;
            if (ok && closing == PIC_P && h->have_last_nonb) ok = x264gpu_slicetype_frame_cost(h->st, h->last_nonb.slot, c.slot, c.slot, j + 1, 0, &pc, nullptr) == X264GPU_OK;
            else pc = ic;
            if (ok && h->st_aq_costs) {
                // x264_rc_analyse_slice: "in AQ, use the weighted score instead" (without macroblock-tree; with it the rate factor does not read the cost)
                const bool isp = closing == PIC_P && h->have_last_nonb;
                ok = x264gpu_slicetype_cost_aq(h->st, c.slot, 0, 0, &ic, nullptr) == X264GPU_OK && (!isp || x264gpu_slicetype_cost_aq(h->st, c.slot, j + 1, 0, &pc, nullptr) == X264GPU_OK);
                if (!isp) pc = ic;
            }
            if (!ok) { xlog(&h->param, X264_LOG_ERROR, "lookahead frame cost failed: %s\n", x264gpu_last_error()); h->failed = true; return false; }
            c.costs[0] = ic; c.costs[1] = pc;
        }
        h->bcoding.push_back({ h->bq[(size_t)j], closing });
        h->last_nonb = h->bq[(size_t)j]; h->have_last_nonb = true;
        const int bref = h->bpyramid && j > 1 ? (j - 1) / 2 : -1;
        if (bref >= 0) h->bcoding.push_back({ h->bq[(size_t)bref], PIC_BREF });
        for (int i = 0; i < j; i++) if (i != bref) h->bcoding.push_back({ h->bq[(size_t)i], PIC_B });
        h->bq.erase(h->bq.begin(), h->bq.begin() + j + 1);
        if (h->mbtree && (closing == PIC_IDR || closing == PIC_I)) {
            // x264 lookahead_slicetype_decide: "for MB-tree, we have to perform propagation analysis on I-frames too" — the analysis again
            // with the keyframe as frames[0]; it decides nothing, its tree reaches the keyframe itself
            StFrames F;
            F.h = h;
            F.f.push_back(&h->last_nonb);
            const int n2 = (int)h->bq.size(), framecnt = n2 < h->st_wait + 1 - (j + 1) ? n2 : (h->st_wait + 1 - (j + 1) > 0 ? h->st_wait + 1 - (j + 1) : 0);
            for (auto &e : h->bq) e.type = e.forced == 2 ? ST_IDR : e.forced == 1 ? ST_I : ST_AUTO;
            for (int i = 0; i < framecnt; i++) F.f.push_back(&h->bq[(size_t)i]);
            h->last_nonb.type = closing == PIC_IDR ? ST_IDR : ST_I;
            if (framecnt > 0) st_analyse(h, F, framecnt, true);
            else st_macroblock_tree(h, F, 0, true);
            if (h->failed) return false;
        }
        return true;
    }
    if (h->bq[0].forced) j = 0;
    else {
        for (int i = 0; i < n && i <= h->bframes; i++) {
            if (h->bq[(size_t)i].forced == 2) { j = i > 0 ? i - 1 : 0; break; }        // IDR next: the picture before it closes the run as P
            if (h->bq[(size_t)i].forced == 1) { j = i; break; }                        // I picture: B pictures in front of it may predict from it
            if (i == h->bframes) { j = i; break; }
        }
        if (j < 0) { if (!flushing) return false; j = n - 1; }                      // end of input: the last picture closes the run
    }
    const int closing = h->bq[(size_t)j].forced == 2 ? PIC_IDR : h->bq[(size_t)j].forced == 1 ? PIC_I : PIC_P;
    h->bcoding.push_back({ h->bq[(size_t)j], closing });
    const int bref = h->bpyramid && j > 1 ? (j - 1) / 2 : -1;
    if (bref >= 0) h->bcoding.push_back({ h->bq[(size_t)bref], PIC_BREF });
    for (int i = 0; i < j; i++) if (i != bref) h->bcoding.push_back({ h->bq[(size_t)i], PIC_B });
    h->bq.erase(h->bq.begin(), h->bq.begin() + j + 1);
    return true;
}

// ---- 2-pass rate control (x264 ratecontrol.c: x264_ratecontrol_new's statistics parser, init_pass2, get_qscale / get_diff_limited_q,
//      qscale2bits; no VBV, no zones, no macroblock-tree file: the tree is off in these sessions) ----
static double p2_qp2qscale(double q) { return rc_qp2qscale(q); }
static double p2_qscale2qp(double qs) { return rc_qscale2qp(qs); }
static double p2_qscale2bits(const x264_t::Pass2Entry &e, double qscale)
{
    if (qscale < 0.1) qscale = 0.1;
    return (e.tex + .1) * pow(e.qscale / qscale, 1.1) + e.mv * pow((e.qscale > 1 ? e.qscale : 1) / (qscale > 1 ? qscale : 1), 0.5) + e.misc;
}
static bool p2_load(x264_t *h, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) { xlog(&h->param, X264_LOG_ERROR, "ratecontrol_init: can't open stats file\n"); return false; }
    char line[2048];
    std::vector<x264_t::Pass2Entry> raw;
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#') continue;
        x264_t::Pass2Entry e;
        long long dur = 0, cpbdur = 0;
        float q = 0, aq = 0;
        int tex = 0, mv = 0, misc = 0, imb = 0, pmb = 0, smb = 0;
        char d = '-';
        if (sscanf(line, " in:%d out:%d type:%c dur:%lld cpbdur:%lld q:%f aq:%f tex:%d mv:%d misc:%d imb:%d pmb:%d smb:%d d:%c", &e.in, &e.out, &e.type, &dur, &cpbdur, &q, &aq,
                   &tex, &mv, &misc, &imb, &pmb, &smb, &d) < 13) { fclose(f); xlog(&h->param, X264_LOG_ERROR, "statistics are damaged at line %d, parser out\n", (int)raw.size() + 1); return false; }
        e.qp = q; e.qscale = p2_qp2qscale(q); e.tex = tex; e.mv = mv; e.misc = misc; e.icount = imb; e.dur = dur > 0 ? (double)dur : 1.0; e.kept_as_ref = e.type != 'b';
        raw.push_back(e);
    }
    fclose(f);
    if (raw.empty()) { xlog(&h->param, X264_LOG_ERROR, "empty stats file\n"); return false; }
    h->p2.assign(raw.size(), x264_t::Pass2Entry()); h->p2_out.assign(raw.size(), 0);
    for (const auto &e : raw) {
        if (e.in < 0 || e.in >= (int)raw.size() || e.out < 0 || e.out >= (int)raw.size()) { xlog(&h->param, X264_LOG_ERROR, "bad frame number (%d) at stats line\n", e.in); return false; }
        h->p2[(size_t)e.in] = e; h->p2_out[(size_t)e.out] = e.in;
    }
    return true;
}
// x264's slice type of a statistics character as the rate control groups them: 0 I, 1 P, 2 B
static int p2_kind(char t) { return t == 'I' || t == 'i' ? 0 : t == 'P' ? 1 : 2; }
static bool p2_init(x264_t *h)
{
    const x264_param_t &p = h->param;
    std::vector<x264_t::Pass2Entry> &E = h->p2;
    const int n = (int)E.size();
    const double fps = h->rc.fps > 0 ? h->rc.fps : 25.0, nmb = h->nmb;
    double duration = 0;
    for (auto &e : E) duration += e.dur;
    duration /= fps * E[0].dur;                               // (durations are in ticks of one picture here: constant frame rate, codec.c:1476-1480)
    const double all_available_bits = p.rc.i_bitrate * 1000.0 * duration;
    const double qblur = p.rc.f_qblur, cplxblur = p.rc.f_complexity_blur, qcompress = p.rc.f_qcompress;
    const int filter_size = (int)(qblur * 4) | 1;
    const double base_cplx = nmb * (p.i_bframe ? 120 : 80);
    const double lstep = pow(2.0, p.rc.i_qp_step / 6.0), lmin = p2_qp2qscale(p.rc.i_qp_min), lmax = p2_qp2qscale(p.rc.i_qp_max);
    const double ipf = fabs(p.rc.f_ip_factor) > 0 ? fabs(p.rc.f_ip_factor) : 1.0, pbf = fabs(p.rc.f_pb_factor) > 0 ? fabs(p.rc.f_pb_factor) : 1.0;
    double all_const_bits = 0;
    for (auto &e : E) all_const_bits += e.misc;
    if (all_available_bits < all_const_bits) {
        xlog(&p, X264_LOG_ERROR, "requested bitrate is too low. estimated minimum is %d kbps\n", (int)(all_const_bits * fps / (n * 1000.)));
        return false;
    }
    // blur the complexities (not the quantisers: one very simple picture must not drag its neighbours down); per unit of BASE_FRAME_DURATION as x264 has it
    const double frame_duration = (1.0 / fps < 0.01 ? 0.01 : 1.0 / fps > 1.0 ? 1.0 : 1.0 / fps) / 0.04;
    for (int i = 0; i < n; i++) {
        double weight_sum = 0, cplx_sum = 0, weight = 1.0;
        for (int j = 1; j < cplxblur * 2 && j < n - i; j++) {
            const auto &r = E[(size_t)(i + j)];
            weight *= 1 - pow((float)r.icount / (float)h->nmb, 2);          // (x264: a float division — i_count and nmb are integers there)
            if (weight < .0001) break;
            const double g = weight * exp(-j * j / 200.0);
            weight_sum += g; cplx_sum += g * (p2_qscale2bits(r, 1) - r.misc) / frame_duration;
        }
        weight = 1.0;
        for (int j = 0; j <= cplxblur * 2 && j <= i; j++) {
            const auto &r = E[(size_t)(i - j)];
            const double g = weight * exp(-j * j / 200.0);
            weight_sum += g; cplx_sum += g * (p2_qscale2bits(r, 1) - r.misc) / frame_duration;
            weight *= 1 - pow((float)r.icount / (float)h->nmb, 2);          // (x264: a float division — i_count and nmb are integers there)
            if (weight < .0001) break;
        }
        E[(size_t)i].blurred = (double)(float)(cplx_sum / weight_sum);          // (ratecontrol_entry_t keeps blurred_complexity as a float)
    }
    // the rate factor: multiplied into every picture's RCEQ value it makes the sizes add up to the request (no closed form: qscale2bits does not invert)
    std::vector<double> qscale((size_t)n), blurred((size_t)n);
    double last_q[3], accum_p_qp = 0, accum_p_norm = 0, last_accum_p_norm = 1;
    int last_non_b = -1;
    auto get_qscale = [&](const x264_t::Pass2Entry &e, double rate_factor) {
        double q = pow(e.blurred, 1 - qcompress);
        if (!std::isfinite(q) || e.tex + e.mv == 0) q = last_q[p2_kind(e.type)];
        else q /= rate_factor;
        return q;
    };
    auto diff_limited = [&](const x264_t::Pass2Entry &e, double q) {
        const int kind = p2_kind(e.type);
        const double last_p_q = last_q[1], last_non_b_q = last_non_b >= 0 ? last_q[last_non_b] : q;
        if (kind == 0) {
            const double iq = q, pq = accum_p_norm > 0 ? p2_qp2qscale(accum_p_qp / accum_p_norm) : q;
            if (accum_p_norm <= 0) q = iq;
            else if (p.rc.f_ip_factor < 0) q = iq / ipf;
            else if (accum_p_norm >= 1) q = pq / ipf;
            else q = accum_p_norm * pq / ipf + (1 - accum_p_norm) * iq;
        } else if (kind == 2) {
            if (p.rc.f_pb_factor > 0) q = last_non_b_q;
            if (!e.kept_as_ref) q *= pbf;
        } else if (last_non_b == 1 && e.tex == 0) q = last_p_q;
        if (last_non_b == kind && (kind != 0 || last_accum_p_norm < 1)) {
            const double lq = last_q[kind];
            q = q > lq * lstep ? lq * lstep : q < lq / lstep ? lq / lstep : q;
        }
        last_q[kind] = q;
        if (kind != 2) last_non_b = kind;
        if (kind == 0) { last_accum_p_norm = accum_p_norm; accum_p_norm = 0; accum_p_qp = 0; }
        if (kind == 1) { const float mask = (float)(1 - pow((float)e.icount / (float)h->nmb, 2)); accum_p_qp          /* (a float in x264) */ = mask * (p2_qscale2qp(q) + accum_p_qp); accum_p_norm = mask * (1 + accum_p_norm); }
        return q;
    };
    double expected_bits = 1;
    last_q[0] = last_q[1] = last_q[2] = pow(base_cplx, 1 - qcompress);
    for (int i = 0; i < n; i++) { const double q = get_qscale(E[(size_t)i], 1.0); expected_bits += p2_qscale2bits(E[(size_t)i], q); last_q[p2_kind(E[(size_t)i].type)] = q; }
    const double step_mult = all_available_bits / expected_bits;
    double rate_factor = 0;
    for (double step = 1E4 * step_mult; step > 1E-7 * step_mult; step *= 0.5) {
        expected_bits = 0;
        rate_factor += step;
        last_non_b = -1; last_accum_p_norm = 1; accum_p_norm = 0; accum_p_qp = 0;
        last_q[0] = last_q[1] = last_q[2] = pow(base_cplx, 1 - qcompress) / rate_factor;
        for (int i = 0; i < n; i++) { qscale[(size_t)i] = get_qscale(E[(size_t)i], rate_factor); last_q[p2_kind(E[(size_t)i].type)] = qscale[(size_t)i]; }
        for (int i = n - 1; i >= 0; i--) qscale[(size_t)i] = diff_limited(E[(size_t)i], qscale[(size_t)i]);       // fixed I / B quantisers relative to P
        if (filter_size > 1) {                                  // smooth the curve over pictures of the same kind
            for (int i = 0; i < n; i++) {
                double q = 0.0, sum = 0.0;
                for (int j = 0; j < filter_size; j++) {
                    const int idx = i + j - filter_size / 2;
                    const double d = idx - i, coeff = qblur == 0 ? 1.0 : exp(-d * d / (qblur * qblur));
                    if (idx < 0 || idx >= n) continue;
                    if (p2_kind(E[(size_t)i].type) != p2_kind(E[(size_t)idx].type)) continue;
                    q += qscale[(size_t)idx] * coeff; sum += coeff;
                }
                blurred[(size_t)i] = q / sum;
            }
        } else blurred = qscale;
        for (int i = 0; i < n; i++) {
            double q = blurred[(size_t)i];
            q = q < lmin ? lmin : q > lmax ? lmax : q;          // clip_qscale without VBV
            E[(size_t)i].new_qscale = q;
            expected_bits += p2_qscale2bits(E[(size_t)i], q);
        }
        if (expected_bits > all_available_bits) rate_factor -= step;
    }
    // the plan in coding order: what should have been spent when each picture starts
    expected_bits = 0;
    for (int k = 0; k < n; k++) { auto &e = E[(size_t)h->p2_out[(size_t)k]]; e.expected_bits = expected_bits; expected_bits += p2_qscale2bits(e, e.new_qscale); }
    h->p2_final_bits = n > 0 ? E[(size_t)h->p2_out[(size_t)(n - 1)]].expected_bits : 0;          // x264: entry_out[num_entries - 1]->expected_bits — what should have been spent BEFORE the last picture
    if (fabs(expected_bits / all_available_bits - 1.0) > 0.01) {
        double avgq = 0;
        for (auto &e : E) avgq += e.new_qscale;
        avgq = p2_qscale2qp(avgq / n);
        xlog(&p, X264_LOG_WARNING, "Error: 2pass curve failed to converge\n");
        xlog(&p, X264_LOG_WARNING, "target: %.2f kbit/s, expected: %.2f kbit/s, avg QP: %.4f\n", (double)p.rc.i_bitrate, expected_bits / duration / 1000., avgq);
    }
    h->p2_abr_buffer = 2 * p.rc.f_rate_tolerance * p.rc.i_bitrate * 1000.0;
    return true;
}
// rate_estimate_qscale, 2-pass branch: the planned quantiser of display picture `frame`, pulled by how far the coded size is from the plan
static double p2_pick_qscale(x264_t *h, int frame, long coded_so_far)
{
    const x264_param_t &p = h->param;
    const int n = (int)h->p2.size();
    if (frame >= n) return h->p2[(size_t)(n - 1)].new_qscale;           // (x264: "2nd pass has more frames than 1st pass", then constant quantiser)
    const x264_t::Pass2Entry &e = h->p2[(size_t)frame];
    double abr_buffer = h->p2_abr_buffer;
    if (n > coded_so_far) {           // adjust the buffer by the distance to the end of the video
        const double video_pos = h->p2_final_bits > 0 ? e.expected_bits / h->p2_final_bits : 1.0, scale_factor = sqrt((1 - video_pos) * n);
        abr_buffer *= 0.5 * (scale_factor > 0.5 ? scale_factor : 0.5);
    }
    const double diff = (double)((long long)h->p2_total_bits - (long long)e.expected_bits);          // (x264: int64_t diff = predicted_bits - (int64_t)rce.expected_bits)
    double q = e.new_qscale, c = (abr_buffer - diff) / abr_buffer;
    q /= c < .5 ? .5 : c > 2 ? 2 : c;
    if (coded_so_far >= h->rc.fps && h->p2_expected_sum >= 1) {          // x264: h->i_frame >= rcc->fps && rcc->expected_bits_sum >= 1
        const double cur_time = (double)coded_so_far / n, w = cur_time * 100 < 0 ? 0 : cur_time * 100 > 1 ? 1 : cur_time * 100;
        q *= pow(h->p2_total_bits / h->p2_expected_sum, w);
    }
    const double lmin = p2_qp2qscale(p.rc.i_qp_min), lmax = p2_qp2qscale(p.rc.i_qp_max);
    return q < lmin ? lmin : q > lmax ? lmax : q;
}

// rate control of one picture of a B session: constant quantisers (x264 rc->qp_constant[] with --ipratio / --pbratio) or CRF (rate_estimate_qscale:
// I / P as without B pictures; B pictures take the distance-weighted average of their nearest references' quantisers plus the pb offset)
static int bmode_qp(x264_t *h, const x264_t::BPlanned &pl, const DpbPlan &plan, double *qp_float)
{
    const x264_param_t &p = h->param;
    const double pb_offset = 6.0 * log2f(fabs(p.rc.f_pb_factor) > 0 ? fabsf(p.rc.f_pb_factor) : 1.0f);          // rc->pb_offset = 6.0 * log2f( f_pb_factor )
    const bool is_i = pl.type == PIC_IDR || pl.type == PIC_I, is_b = pl.type == PIC_B || pl.type == PIC_BREF;
    if (h->pass2) {
        double q = p2_qscale2qp(p2_pick_qscale(h, pl.e.frame, h->coded_count));
        q = q < p.rc.i_qp_min ? p.rc.i_qp_min : q > p.rc.i_qp_max ? p.rc.i_qp_max : q;
        *qp_float = q;
        return clampi((int)(q + 0.5), 1, 51);
    }
    if (!h->crf && !h->abr) {
        const int qb = clampi((int)(h->qp_p + pb_offset + 0.5), 0, 51);
        int q = is_i ? h->qp_i : !is_b ? h->qp_p : pl.type == PIC_BREF ? (qb + h->qp_p) / 2 : qb;
        if (const x264_t::Zone *z = get_zone(h, pl.e.frame)) q = cqp_zone(h, *z, q);
        *qp_float = q;
        return q;
    }
    if (!is_b) {
        const int q = rc_pick_qp(h, is_i, pl.e.costs, h->rc_frames, pl.e.frame);
        *qp_float = h->rc.qpa_last;
        return q;
    }
    const int s0 = plan.pic.slot[0][0], s1 = plan.pic.slot[1][0];
    const bool i0 = h->slot_ptype[s0] == PIC_IDR || h->slot_ptype[s0] == PIC_I, i1 = h->slot_ptype[s1] == PIC_IDR || h->slot_ptype[s1] == PIC_I;
    const int dt0 = abs(plan.pic.poc - plan.list_poc[0][0]), dt1 = abs(plan.pic.poc - plan.list_poc[1][0]);
    // rate_estimate_qscale's B branch in x264's own types: float q0, q1, q (f_qp_avg_rc of the nearest references), double offsets; the result goes
    // through qp2qscale and x264_ratecontrol_start's qscale2qp like every quantiser
    float q0 = (float)h->slot_qp_rc[s0], q1 = (float)h->slot_qp_rc[s1], qf;
    if (h->slot_ptype[s0] == PIC_BREF) q0 = (float)(q0 - pb_offset / 2);
    if (h->slot_ptype[s1] == PIC_BREF) q1 = (float)(q1 - pb_offset / 2);
    if (i0 && i1) qf = (float)((q0 + q1) / 2 + h->rc.ip_offset);
    else if (i0) qf = q1;
    else if (i1) qf = q0;
    else qf = (q0 * dt1 + q1 * dt0) / (dt0 + dt1);
    qf = (float)(qf + (pl.type == PIC_BREF ? pb_offset / 2 : pb_offset));
    double q = rc_qscale2qp(rc_qp2qscale(qf));
    q = q < p.rc.i_qp_min ? p.rc.i_qp_min : q > p.rc.i_qp_max ? p.rc.i_qp_max : q;
    // x264_ratecontrol_start: accum_p_qp_update runs for every picture type — a B picture's quantiser enters the running average an I picture
    // after P pictures takes its quantiser from
    h->rc.accum_p_qp = h->rc.accum_p_qp * 0.95 + q;
    h->rc.accum_p_norm = h->rc.accum_p_norm * 0.95 + 1.0;
    *qp_float = q;
    return clampi((int)(q + 0.5), p.rc.i_qp_min, p.rc.i_qp_max);
}

// hands out a picture a helper thread finished (batch sessions with overlap): waits for the thread, publishes its NAL units; 0 when the slot is empty
static int publish_deferred(x264_t *h, x264_t::Deferred &d, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_out)
{
    if (!d.valid) return 0;
    if (d.th.joinable()) {
        d.hurry = true;
        if (h->batch) { std::lock_guard<std::mutex> lg(h->batch->m); h->batch->cv.notify_all(); }
        const long t0 = h->batch && h->batch->timing ? us_now() : 0;
        d.th.join();
        if (h->batch && h->batch->timing) h->batch->t_us[5] += us_now() - t0;
    }
    d.valid = false;
    if (!d.err.empty()) { xlog(&h->param, X264_LOG_ERROR, "x264_encoder_encode: download of a batched picture failed: %s\n", d.err.c_str()); h->failed = true; return -1; }
    h->out.swap(d.out); h->nal_off = d.off; h->last_stats = d.stats;
    // the diagnostics hooks describe the picture whose NAL units this call returns, not the one submitted meanwhile
    h->last_qp = d.qp; h->last_qpm = d.qpm; h->last_scenecut = d.scenecut; memcpy(h->last_costs, d.costs, sizeof(d.costs));
    publish_nals(h, pp_nal, pi_nal, d.types);
    for (size_t i = 0; i < h->nals.size(); i++) if (d.types[i] == 1 || d.types[i] == 5) h->nals[i].i_ref_idc = d.nal_ref_idc;
    if (pic_out) {
        x264_picture_init(pic_out);
        pic_out->i_type = d.i_type; pic_out->b_keyframe = d.b_keyframe; pic_out->i_pts = d.pts; pic_out->i_dts = d.dts; pic_out->img = d.img;
    }
    return (int)h->out.size();
}

// codes the next picture in coding order; returns the bytes of its NAL units, 0 when the queue still waits for input
static int encode_bmode(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_out, bool flushing)
{
    const x264_param_t &p = h->param;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tb0 = now(), tb1 = 0;
    auto BPHASE = [&](int i) { tb1 = now(); h->t_b[i] += tb1 - tb0; tb0 = tb1; };
    const bool decided = bmode_decide(h, flushing);
    BPHASE(0);
    if (!decided) return flushing ? publish_deferred(h, h->defer[h->defer_cur ^ 1].valid ? h->defer[h->defer_cur ^ 1] : h->defer[h->defer_cur], pp_nal, pi_nal, pic_out) : 0;
    const x264_t::BPlanned pl = h->bcoding.front();
    h->bcoding.pop_front();
    // the disposable pictures coded right behind this one (x264_reference_hierarchy_reset looks at them)
    int fc[16], ff[16], nf = 0;
    for (size_t i = 0; i < h->bcoding.size() && nf < 16 && h->bcoding[i].type == PIC_B; i++) { fc[nf] = (int)(h->coded_count + 1 + (long)i); ff[nf] = h->bcoding[i].e.frame; nf++; }
    const DpbPlan plan = h->dpb.plan(pl.type, pl.e.frame, nf, fc, ff, pl.type == PIC_P && pl.e.w.on ? &pl.e.w : nullptr);
    x264gpu_pic pic = plan.pic;
    double qpf = 0;
    pic.qp = bmode_qp(h, pl, plan, &qpf);
    // x264_ratecontrol_mb_qp: a macroblock's quantiser is round(rc->qpm + its AQ / macroblock-tree offset) with qpm the picture's FLOAT quantiser
    pic.qpm = near_qpm(qpf, pic.qp);
    h->rc_frames++;
    bool direct_auto_write = false;
    if ((pl.type == PIC_B || pl.type == PIC_BREF) && h->direct_mode != 1) {
        // x264 slice_header_init: temporal direct prediction is considered only when the co-located picture's reference 0 is this picture's reference 0;
        // --direct auto: the mode whose skip probe passed more often so far (h->stat.i_direct_score), every macroblock probing both again
        bool spatial = true;
        if (pic.nref[0] && pic.nref[1] && h->slot_l0ref0poc[pic.slot[1][0]] == plan.list_poc[0][0]) {
            direct_auto_write = h->direct_mode == 3;
            spatial = direct_auto_write ? h->direct_score[1] > h->direct_score[0] : false;
        }
        pic.direct_temporal = !spatial; pic.direct_auto = direct_auto_write;
        h->dpb.set_direct(pic.direct_temporal, pic.direct_auto);
    }
    if (h->st) {
        // x264_mb_predict_mv_ref16x16: the lookahead's vectors towards reference 0 of each list as search candidates, when that search ran
        // (fenc->lowres_mvs[list][distance - 1], distances up to bframes + 1)
        const int dist0 = pic.nref[0] ? (pic.poc - plan.list_poc[0][0]) / 2 : 0, dist1 = pic.nref[1] ? (plan.list_poc[1][0] - pic.poc) / 2 : 0;
        const int16_t *m0 = dist0 >= 1 && dist0 <= h->bframes + 1 ? x264gpu_slicetype_lowres_mvs(h->st, pl.e.slot, 0, dist0) : nullptr;
        const int16_t *m1 = dist1 >= 1 && dist1 <= h->bframes + 1 ? x264gpu_slicetype_lowres_mvs(h->st, pl.e.slot, 1, dist1) : nullptr;
        x264gpu_encoder_set_lowres_mvs(h->gpu, m0);
        x264gpu_encoder_set_lowres_mvs1(h->gpu, m1);
    }
    const float *d_offsets = nullptr;
    if (h->st && h->mbtree)          // P / I / B-reference pictures: what the tree left (AQ - tree); other B pictures: the AQ offsets alone (x264 f_qp_offset_aq)
        d_offsets = pl.type == PIC_B ? h->q_aq[(size_t)pl.e.slot] : h->q_tree[(size_t)pl.e.slot];
    else if (h->aq_mode >= 2 && h->aq_strength != 0.f) d_offsets = h->q_aq[(size_t)pl.e.slot];      // --aq-mode 2 / 3: the offsets computed when the picture arrived
    if (d_offsets) x264gpu_encoder_set_mb_qp_offsets(h->gpu, d_offsets);
    int bbuf = 0;
    bool deferred = false;
    if (h->batch) {
        std::string berr;
        deferred = h->batch->overlap;          // the download and the entropy coding of this picture run beside the group's next round (below)
        if (batch_submit(h->batch, h->batch_idx, h->q_raw[(size_t)pl.e.slot], pic, &bbuf, berr) ||
            (!deferred && batch_download(h->batch, h->batch_idx, bbuf, h->h_mb.data(), h->h_lv.data(), berr))) {
            xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", berr.c_str());
            h->failed = true;
            return -1;
        }
    } else
    if (x264gpu_encode_pictures(h->gpu, h->q_raw[(size_t)pl.e.slot], &pic, h->d_mb, h->d_lv, nullptr) != X264GPU_OK || (getenv("X264GPU_HOST_TIMING") && (x264gpu_stream_sync(nullptr), BPHASE(1), false)) ||
        x264gpu_memcpy_d2h(h->h_mb.data(), h->d_mb, h->h_mb.size() * sizeof(x264gpu_mb), nullptr) != X264GPU_OK ||
        x264gpu_memcpy_d2h(h->h_lv.data(), h->d_lv, h->h_lv.size() * sizeof(int16_t), nullptr) != X264GPU_OK) {
        xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", x264gpu_last_error());
        h->failed = true;
        return -1;
    }
    BPHASE(2);
    if (const char *dd = getenv("X264GPU_DUMP_RECORDS")) {          // debugging aid: the records and levels of every coded picture, and what was asked of the device
        char fn[512];
        snprintf(fn, sizeof(fn), "%s/pic%04ld.bin", dd, h->coded_count);
        if (FILE *f = fopen(fn, "wb")) {
            fwrite(&pic, sizeof(pic), 1, f);
            fwrite(h->h_mb.data(), sizeof(x264gpu_mb), h->h_mb.size(), f);
            fwrite(h->h_lv.data(), sizeof(int16_t), h->h_lv.size(), f);
            std::vector<float> off((size_t)h->nmb, 0.f);          // ... and the per-macroblock quantiser offsets it was coded with (zeros: none handed in)
            if (d_offsets) x264gpu_memcpy_d2h(off.data(), d_offsets, off.size() * sizeof(float), nullptr);
            fwrite(off.data(), sizeof(float), off.size(), f);
            fclose(f);
        }
    }
    if (plan.nal_ref_idc) { h->slot_qp_rc[pic.dst] = rc_qp_avg_rc((float)qpf, h->mbw, h->mbh); h->slot_ptype[pic.dst] = pl.type; }
    if (plan.nal_ref_idc) h->slot_l0ref0poc[pic.dst] = pic.nref[0] ? plan.list_poc[0][0] : INT_MIN;
    h->last_direct_char = (pl.type == PIC_B || pl.type == PIC_BREF) ? (pic.direct_temporal ? 't' : 's') : '-';
    if (direct_auto_write) {
        // x264_encoder_frame_end ("somewhat arbitrary time constants"): the running counts decay once they exceed a picture's worth, then take this picture's
        int sc[2] = { 0, 0 };
        if (x264gpu_encoder_direct_scores(h->gpu, sc) != X264GPU_OK) { xlog(&p, X264_LOG_ERROR, "direct auto: %s\n", x264gpu_last_error()); h->failed = true; return -1; }
        if (h->direct_score[0] + h->direct_score[1] > h->mbw * h->mbh) for (int i = 0; i < 2; i++) h->direct_score[i] = h->direct_score[i] * 9 / 10;
        for (int i = 0; i < 2; i++) h->direct_score[i] += sc[i];
    }
    h->last_scenecut = pl.e.scenecut; h->last_qp = pic.qp; h->last_qpm = pic.qpm;
    memcpy(h->last_costs, pl.e.costs, sizeof(pl.e.costs));
    const bool idr = pl.type == PIC_IDR;
    h->out.clear(); h->nal_off.clear();
    std::vector<int> types;
    if (p.b_aud) { h->nal_off.push_back(h->out.size()); types.push_back(9); write_aud(h->out, pl.type <= PIC_I ? 0 : pl.type == PIC_P ? 1 : 2, p.b_annexb != 0); }
    if (idr && p.b_repeat_headers) { emit_sets(h, types, !h->sei_sent); h->sei_sent = 1; }
    SliceParams sp = {};
    sp.mbw = h->mbw; sp.mbh = h->mbh; sp.qp = pic.qp; sp.pic_init_qp = h->pic_init_qp; sp.log2_max_frame_num = h->log2_max_frame_num; sp.log2_max_poc_lsb = h->log2_max_poc_lsb;
    sp.idr_pic_id = h->idr_pic_id; sp.pps_id = p.i_sps_id; sp.num_ref_default = p.i_frame_reference; sp.num_ref1_default = 1;
    sp.disable_deblock_idc = p.b_deblocking_filter ? 0 : 1; sp.alpha_off_div2 = p.i_deblocking_filter_alphac0; sp.beta_off_div2 = p.i_deblocking_filter_beta;
    sp.transform8x8_mode = p.analyse.b_transform_8x8; sp.cabac = p.b_cabac; sp.slices_plain = h->slices_plain;
    h->dpb.fill(sp);
    if (deferred) {
        // everything the slice writer needs is fixed now: the helper thread downloads this stream's records and levels (on the group's download stream) and writes the
        // slices behind the header NAL units; the DPB moves on at once (the next picture's plan needs it), the NAL units leave with the next call
        x264_t::Deferred &d = h->defer[h->defer_cur];
        if (d.th.joinable()) d.th.join();          // (handed out two calls ago: long finished)
        d.out = h->out; d.off = h->nal_off; d.types = types; d.err.clear(); d.nal_ref_idc = plan.nal_ref_idc; d.stats = SliceStats{ 0 };
        d.mb.resize((size_t)h->nmb);
        if (h->batch->pack) d.ix.resize((size_t)h->nmb);
        if (!d.lv) d.lv.reset(new (std::nothrow) int16_t[(size_t)h->nmb * X264GPU_MB_LEVELS]);
        if (!d.lv) { xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: out of memory (download buffers)\n"); h->failed = true; return -1; }
        d.i_type = idr ? X264_TYPE_IDR : pl.type == PIC_I ? X264_TYPE_I : pl.type == PIC_P ? X264_TYPE_P : pl.type == PIC_BREF ? X264_TYPE_BREF : X264_TYPE_B;
        d.b_keyframe = idr; d.pts = pl.e.pts; d.img = pl.e.img;
        d.qp = pic.qp; d.qpm = pic.qpm; d.scenecut = pl.e.scenecut; memcpy(d.costs, pl.e.costs, sizeof(d.costs));
        // (the deferred return skips x264_ratecontrol_end and the statistics line below: batched sessions are never ABR / 2-pass — x264_encoder_open admits
        //  constant-quantiser and CRF sessions into a batch only)
        if (h->abr || h->pass1 || h->pass2) { xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: a batched session cannot run rate control that reads the coded sizes\n"); h->failed = true; return -1; }
        {
            const long k = h->coded_count, delay = !h->bframes ? 0 : h->bpyramid ? 2 : 1;
            const size_t np = h->all_pts.size();
            if (k >= delay) d.dts = h->all_pts[(size_t)(k - delay) < np ? (size_t)(k - delay) : np - 1];
            else d.dts = h->all_pts[(size_t)k < np ? (size_t)k : np - 1] - (h->all_pts[(size_t)delay < np ? (size_t)delay : np - 1] - h->all_pts[0]);
        }
        BatchGroup *g = h->batch;
        const int bidx = h->batch_idx, slices = h->slices, threads = h->cavlc_threads, dev = h->device;
        const bool annexb = p.b_annexb != 0, first = d.off.empty();
        long my_launched;
        { std::lock_guard<std::mutex> lg(g->m); my_launched = g->launched; }
        d.hurry = false;
        d.th = std::thread([&d, g, bidx, bbuf, sp, slices, threads, annexb, first, idr, dev, my_launched]() {
            x264gpu_set_device(dev);
            x264gpu_level_index *ix = d.ix.empty() ? nullptr : d.ix.data();
            if (batch_download(g, bidx, bbuf, d.mb.data(), d.lv.get(), d.err, ix)) return;
            const long t0 = g->timing ? us_now() : 0;
            {
                // the slices are written once the group's next round is on the device (the callers need the cores to get it there), or when the picture is asked for
                std::unique_lock<std::mutex> lk(g->m);
                g->cv.wait_for(lk, std::chrono::seconds(30), [&] { return g->launched > my_launched || d.hurry.load() || g->closed; });
            }
            const long t1 = g->timing ? us_now() : 0;
            const size_t before = d.off.size();
            write_picture(d.out, &d.off, sp, slices, d.mb.data(), d.lv.get(), annexb, first, &d.stats, threads, ix);
            if (g->timing) { const long t2 = us_now(); g->t_us[2] += t1 - t0; g->t_us[3] += t2 - t1; if (my_launched >= 1 && my_launched <= 64) g->r_sl[my_launched - 1] += t2 - t1; }
            for (size_t i = before; i < d.off.size(); i++) d.types.push_back(idr ? 5 : 1);
        });
        d.valid = true;
        h->dpb.commit();
        BPHASE(3);
        h->t_b[4] += 1;
        if (idr) h->idr_pic_id = (h->idr_pic_id + 1) & 0xffff;
        h->coded_count++;
        h->frame_no++;
        h->defer_cur ^= 1;
        return publish_deferred(h, h->defer[h->defer_cur], pp_nal, pi_nal, pic_out);          // the picture of the call before (0: none yet)
    }
    h->last_stats.skip = 0;
    {
        const size_t before = h->nal_off.size();
        write_picture(h->out, &h->nal_off, sp, h->slices, h->h_mb.data(), h->h_lv.data(), p.b_annexb != 0, before == 0, &h->last_stats, h->cavlc_threads);
        for (size_t i = before; i < h->nal_off.size(); i++) types.push_back(idr ? 5 : 1);
    }
    h->dpb.commit();
    publish_nals(h, pp_nal, pi_nal, types);
    for (size_t i = 0; i < h->nals.size(); i++) if (types[i] == 1 || types[i] == 5) h->nals[i].i_ref_idc = plan.nal_ref_idc;
    if (pic_out) {
        x264_picture_init(pic_out);
        pic_out->i_type = idr ? X264_TYPE_IDR : pl.type == PIC_I ? X264_TYPE_I : pl.type == PIC_P ? X264_TYPE_P : pl.type == PIC_BREF ? X264_TYPE_BREF : X264_TYPE_B;
        pic_out->b_keyframe = idr;
        pic_out->i_pts = pl.e.pts;
        // x264: the k-th coded picture's dts is the pts of display picture k - delay; the first `delay` ones are shifted back by the delay's duration
        const long k = h->coded_count, delay = !h->bframes ? 0 : h->bpyramid ? 2 : 1;
        const size_t np = h->all_pts.size();
        if (k >= delay) pic_out->i_dts = h->all_pts[(size_t)(k - delay) < np ? (size_t)(k - delay) : np - 1];
        else pic_out->i_dts = h->all_pts[(size_t)k < np ? (size_t)k : np - 1] - (h->all_pts[(size_t)delay < np ? (size_t)delay : np - 1] - h->all_pts[0]);
        pic_out->img = pl.e.img;
    }
    if (h->pass1 || h->pass2) {
        // x264_ratecontrol_end: the picture's line of the statistics file / the second pass' account of what was spent against the plan
        const long total = (long)h->out.size() * 8;
        long imb = 0, pmb = 0, smb = 0;
        double aqsum = 0;
        for (size_t i = 0; i < (size_t)h->nmb; i++) {
            const x264gpu_mb &m = h->h_mb[i];
            if (m.type <= X264GPU_MB_I16x16) imb++; else if (m.type == X264GPU_MB_P_SKIP || m.type == X264GPU_MB_B_SKIP) smb++; else pmb++;
            aqsum += m.qp;
        }
        if (h->stat_file) {
            const char t = idr ? 'I' : pl.type == PIC_I ? 'i' : pl.type == PIC_P ? 'P' : pl.type == PIC_BREF ? 'B' : 'b';
            const long mv = h->last_stats.mv_bits, tex = h->last_stats.tex_bits, misc = total - mv - tex;
            fprintf(h->stat_file, "in:%d out:%ld type:%c dur:%d cpbdur:%d q:%.2f aq:%.2f tex:%ld mv:%ld misc:%ld imb:%ld pmb:%ld smb:%ld d:%c ref:;\n", pl.e.frame, h->coded_count, t, 1, 1, qpf,
                    aqsum / h->nmb, tex, mv, misc, imb, pmb, smb, h->last_direct_char);
        }
        if (h->pass2) {
            h->p2_total_bits += (double)total;
            if (pl.e.frame < (int)h->p2.size()) h->p2_expected_sum += p2_qscale2bits(h->p2[(size_t)pl.e.frame], p2_qp2qscale(qpf));
        }
    }
    if (h->abr) {
        // x264_ratecontrol_end: what the picture took moves the rate factor of the pictures to come (a B picture's quantiser is an offset of
        // its neighbours': its bits count divided by pbratio)
        const double bits = (double)h->out.size() * 8.0, pb = fabs(p.rc.f_pb_factor) > 0 ? fabs(p.rc.f_pb_factor) : 1.0;
        const bool is_b = pl.type == PIC_B || pl.type == PIC_BREF;
        h->rc.total_bits += bits;
        h->rc.cplxr_sum += bits * rc_qp2qscale(rc_qp_avg_rc((float)qpf, h->mbw, h->mbh)) / (h->rc.last_rceq * (is_b ? pb : 1.0));
        h->rc.wanted_bits_window += h->rc.bitrate / h->rc.fps;
    }
    BPHASE(3);
    h->t_b[4] += 1;
    if (idr) h->idr_pic_id = (h->idr_pic_id + 1) & 0xffff;
    h->coded_count++;
    h->frame_no++;
    return (int)h->out.size();
}

// ---- several pictures of the session in flight (x264_t::Inflight) ----
// issues the next picture of the coding order through a free launch context; false: nothing could be issued (no picture decided, no free context, no free slot)
static bool inflight_issue(x264_t *h, bool flushing)
{
    const x264_param_t &p = h->param;
    int ci = -1;
    for (int i = 0; i < (int)h->lctx.size(); i++) if (!h->lctx[(size_t)i].busy) { ci = i; break; }
    if (ci < 0) return false;
    if (h->bcoding.empty() && !bmode_decide(h, flushing)) return false;
    if (h->failed) return false;
    // the slots the pictures in flight write or read stay out of the choice of a destination
    unsigned avoid = 0;
    for (const x264_t::Inflight &f : h->fl) avoid |= f.slots_used;
    h->dpb.avoid = avoid;
    if (!h->dpb.has_free_slot()) return false;
    const x264_t::BPlanned pl = h->bcoding.front();
    h->bcoding.pop_front();
    int fc[16], ff[16], nf = 0;
    const long coded_index = h->coded_count + (long)h->fl.size();          // this picture's place in the coding order
    for (size_t i = 0; i < h->bcoding.size() && nf < 16 && h->bcoding[i].type == PIC_B; i++) { fc[nf] = (int)(coded_index + 1 + (long)i); ff[nf] = h->bcoding[i].e.frame; nf++; }
    const DpbPlan plan = h->dpb.plan(pl.type, pl.e.frame, nf, fc, ff, pl.type == PIC_P && pl.e.w.on ? &pl.e.w : nullptr);
    x264_t::Inflight f;
    f.pl = pl; f.ctx = ci; f.nal_ref_idc = plan.nal_ref_idc;
    x264gpu_pic pic = plan.pic;
    double qpf = 0;
    pic.qp = bmode_qp(h, pl, plan, &qpf);
    pic.qpm = near_qpm(qpf, pic.qp);
    h->rc_frames++;
    if ((pl.type == PIC_B || pl.type == PIC_BREF) && h->direct_mode != 1) {
        // x264 slice_header_init: temporal direct prediction only when the co-located picture's reference 0 is this picture's reference 0 (--direct auto runs serially)
        const bool temporal = pic.nref[0] && pic.nref[1] && h->slot_l0ref0poc[pic.slot[1][0]] == plan.list_poc[0][0];
        pic.direct_temporal = temporal; pic.direct_auto = 0;
        h->dpb.set_direct(pic.direct_temporal, 0);
    }
    x264_t::LaunchCtx &c = h->lctx[(size_t)ci];
    if (h->st) {
        const int dist0 = pic.nref[0] ? (pic.poc - plan.list_poc[0][0]) / 2 : 0, dist1 = pic.nref[1] ? (plan.list_poc[1][0] - pic.poc) / 2 : 0;
        const int16_t *m0 = dist0 >= 1 && dist0 <= h->bframes + 1 ? x264gpu_slicetype_lowres_mvs(h->st, pl.e.slot, 0, dist0) : nullptr;
        const int16_t *m1 = dist1 >= 1 && dist1 <= h->bframes + 1 ? x264gpu_slicetype_lowres_mvs(h->st, pl.e.slot, 1, dist1) : nullptr;
        x264gpu_encoder_set_lowres_mvs(c.gpu, m0);
        x264gpu_encoder_set_lowres_mvs1(c.gpu, m1);
    }
    const float *d_offsets = nullptr;
    if (h->st && h->mbtree) d_offsets = pl.type == PIC_B ? h->q_aq[(size_t)pl.e.slot] : h->q_tree[(size_t)pl.e.slot];
    else if (h->aq_mode >= 2 && h->aq_strength != 0.f) d_offsets = h->q_aq[(size_t)pl.e.slot];
    x264gpu_encoder_set_mb_qp_offsets(c.gpu, d_offsets);
    // behind the pictures in flight that write a slot this picture reads (its references; their side data lives with the slot)
    f.slots_used = 1u << pic.dst;
    bool ok = true;
    for (int l = 0; l < 2 && ok; l++)
        for (int r = 0; r < pic.nref[l] && ok; r++) {
            const int sl = pic.slot[l][r];
            f.slots_used |= 1u << sl;
            const int w = h->slot_writer[sl];
            if (w >= 0 && w != ci) ok = x264gpu_stream_wait_event(c.stream, h->lctx[(size_t)w].ev) == X264GPU_OK;
        }
    // ... and behind the default stream: this picture's upload, its quantiser offsets and the lookahead's vectors were produced there
    ok = ok && x264gpu_event_record(h->ev_la, h->up_stream) == X264GPU_OK && x264gpu_stream_wait_event(c.stream, h->ev_la) == X264GPU_OK;
    if (ok && h->up_stream) ok = x264gpu_event_record(h->ev_la, nullptr) == X264GPU_OK && x264gpu_stream_wait_event(c.stream, h->ev_la) == X264GPU_OK;
    ok = ok && x264gpu_encode_pictures(c.gpu, h->q_raw[(size_t)pl.e.slot], &pic, c.d_mb, c.d_lv, c.stream) == X264GPU_OK && x264gpu_event_record(c.ev, c.stream) == X264GPU_OK;
    if (!ok) { xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", x264gpu_last_error()); h->failed = true; return false; }
    c.busy = true;
    h->slot_writer[pic.dst] = ci;
    if (plan.nal_ref_idc) { h->slot_qp_rc[pic.dst] = rc_qp_avg_rc((float)qpf, h->mbw, h->mbh); h->slot_ptype[pic.dst] = pl.type; }
    if (plan.nal_ref_idc) h->slot_l0ref0poc[pic.dst] = pic.nref[0] ? plan.list_poc[0][0] : INT_MIN;
    f.direct_char = (pl.type == PIC_B || pl.type == PIC_BREF) ? (pic.direct_temporal ? 't' : 's') : '-';
    f.pic = pic;
    SliceParams sp = {};
    sp.mbw = h->mbw; sp.mbh = h->mbh; sp.qp = pic.qp; sp.pic_init_qp = h->pic_init_qp; sp.log2_max_frame_num = h->log2_max_frame_num; sp.log2_max_poc_lsb = h->log2_max_poc_lsb;
    sp.pps_id = p.i_sps_id; sp.num_ref_default = p.i_frame_reference; sp.num_ref1_default = 1;
    sp.disable_deblock_idc = p.b_deblocking_filter ? 0 : 1; sp.alpha_off_div2 = p.i_deblocking_filter_alphac0; sp.beta_off_div2 = p.i_deblocking_filter_beta;
    sp.transform8x8_mode = p.analyse.b_transform_8x8; sp.cabac = p.b_cabac; sp.slices_plain = h->slices_plain;
    h->dpb.fill(sp);
    h->dpb.commit();
    f.sp = sp;
    h->fl.push_back(f);
    return true;
}

// hands back the oldest picture in flight: waits for its launch context, downloads its records, writes its NAL units
static int inflight_retire(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_out)
{
    const x264_param_t &p = h->param;
    x264_t::Inflight f = h->fl.front();
    h->fl.pop_front();
    x264_t::LaunchCtx &c = h->lctx[(size_t)f.ctx];
    if (x264gpu_event_sync(c.ev) != X264GPU_OK ||
        x264gpu_memcpy_d2h(h->h_mb.data(), c.d_mb, h->h_mb.size() * sizeof(x264gpu_mb), c.stream) != X264GPU_OK ||
        x264gpu_memcpy_d2h(h->h_lv.data(), c.d_lv, h->h_lv.size() * sizeof(int16_t), c.stream) != X264GPU_OK) {
        xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: GPU hot path failed: %s\n", x264gpu_last_error());
        h->failed = true;
        return -1;
    }
    c.busy = false;
    if (h->slot_writer[f.pic.dst] == f.ctx) h->slot_writer[f.pic.dst] = -1;
    h->last_retired_slot = f.pic.dst;
    const x264_t::BPlanned &pl = f.pl;
    h->last_direct_char = f.direct_char;
    h->last_scenecut = pl.e.scenecut; h->last_qp = f.pic.qp; h->last_qpm = f.pic.qpm;
    memcpy(h->last_costs, pl.e.costs, sizeof(pl.e.costs));
    const bool idr = pl.type == PIC_IDR;
    h->out.clear(); h->nal_off.clear();
    std::vector<int> types;
    if (p.b_aud) { h->nal_off.push_back(h->out.size()); types.push_back(9); write_aud(h->out, pl.type <= PIC_I ? 0 : pl.type == PIC_P ? 1 : 2, p.b_annexb != 0); }
    if (idr && p.b_repeat_headers) { emit_sets(h, types, !h->sei_sent); h->sei_sent = 1; }
    SliceParams sp = f.sp;
    sp.idr_pic_id = h->idr_pic_id;
    h->last_stats.skip = 0;
    {
        const size_t before = h->nal_off.size();
        write_picture(h->out, &h->nal_off, sp, h->slices, h->h_mb.data(), h->h_lv.data(), p.b_annexb != 0, before == 0, &h->last_stats, h->cavlc_threads);
        for (size_t i = before; i < h->nal_off.size(); i++) types.push_back(idr ? 5 : 1);
    }
    publish_nals(h, pp_nal, pi_nal, types);
    for (size_t i = 0; i < h->nals.size(); i++) if (types[i] == 1 || types[i] == 5) h->nals[i].i_ref_idc = f.nal_ref_idc;
    if (pic_out) {
        x264_picture_init(pic_out);
        pic_out->i_type = idr ? X264_TYPE_IDR : pl.type == PIC_I ? X264_TYPE_I : pl.type == PIC_P ? X264_TYPE_P : pl.type == PIC_BREF ? X264_TYPE_BREF : X264_TYPE_B;
        pic_out->b_keyframe = idr;
        pic_out->i_pts = pl.e.pts;
        const long k = h->coded_count, delay = !h->bframes ? 0 : h->bpyramid ? 2 : 1;
        const size_t np = h->all_pts.size();
        if (k >= delay) pic_out->i_dts = h->all_pts[(size_t)(k - delay) < np ? (size_t)(k - delay) : np - 1];
        else pic_out->i_dts = h->all_pts[(size_t)k < np ? (size_t)k : np - 1] - (h->all_pts[(size_t)delay < np ? (size_t)delay : np - 1] - h->all_pts[0]);
        pic_out->img = pl.e.img;
    }
    h->t_b[4] += 1;
    if (idr) h->idr_pic_id = (h->idr_pic_id + 1) & 0xffff;
    h->coded_count++;
    h->frame_no++;
    return (int)h->out.size();
}

static int encode_bmode_inflight(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_out, bool flushing)
{
    // issue what can be issued (a new picture's arrival may decide a mini-GOP: its closing picture, its B reference, its b pictures, as far as contexts and slots go) ...
    while (inflight_issue(h, flushing)) ;
    if (h->failed) return -1;
    if (h->fl.empty()) return 0;
    // ... and hand back the oldest picture once `inflight` pictures are out (x264's frame threads: i_thread_frames - 1 more calls of delay), when a decided picture
    // waits for a context or a slot, or when the input has ended; else this call returns nothing (a delayed frame).  The pictures behind the oldest keep running:
    // in the steady state a call issues one picture and waits for one that was issued inflight - 1 calls ago.
    if (!(flushing || (int)h->fl.size() >= h->inflight || !h->bcoding.empty())) return 0;
    return inflight_retire(h, pp_nal, pi_nal, pic_out);
}

int x264_encoder_encode(x264_t *h, x264_nal_t **pp_nal, int *pi_nal, x264_picture_t *pic_in, x264_picture_t *pic_out)
{
    if (!h || !pp_nal || !pi_nal) return -1;
    *pi_nal = 0; *pp_nal = nullptr;
    if (h->failed) return -1;
    if (!pic_in) {      // flush: GOP-parallel batches, or the pictures still waiting in the lookahead queue, one per call
        if (h->G > 1) return encode_gop_parallel(h, pp_nal, pi_nal, nullptr, pic_out, false);
        if (h->dpbmode) return h->inflight > 1 ? encode_bmode_inflight(h, pp_nal, pi_nal, pic_out, true) : encode_bmode(h, pp_nal, pi_nal, pic_out, true);
        return h->queue.empty() ? 0 : encode_queued(h, pp_nal, pi_nal, pic_out, true);
    }
    const x264_param_t &p = h->param;
    const int w = p.i_width, ht = p.i_height;
    if ((pic_in->img.i_csp & X264_CSP_MASK) != X264_CSP_I420 || pic_in->img.i_plane < 3) {
        xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: input picture must be I420\n");
        return -1;
    }
    // Zero-copy input (x264gpu_host_input_i420): the caller (the VfW shell after the device-side colourspace conversion)
    // already placed a tight I420 picture in the encoder's device staging buffer.
    const bool resident = pic_in->img.plane[0] == h->d_in;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t1;
#define PHASE(i) do { t1 = now(); h->t_phase[i] += t1 - t0; t0 = t1; } while (0)
    // ---- frame copy-in: three strided planes -> one tightly packed I420 buffer -> HBM (replaces x264_frame_copy_picture) ----
    uint8_t *dst = h->h_in.data();
    for (int pl = 0; pl < 3 && !resident; pl++) {
        int pw = pl ? w / 2 : w, ph = pl ? ht / 2 : ht;
        const uint8_t *src = pic_in->img.plane[pl];
        for (int y = 0; y < ph; y++, dst += pw, src += pic_in->img.i_stride[pl]) memcpy(dst, src, pw);
    }
    if (h->G > 1) return encode_gop_parallel(h, pp_nal, pi_nal, pic_in, pic_out, resident);
    PHASE(0);
    // ---- the picture enters the lookahead queue (x264_lookahead_put_frame): upload, frame cost against the previous source picture,
    //      AQ offsets, slice-type decision (keyint / forced type / scenecut) — all causal, so they are taken on arrival ----
    const int slot = (int)(h->la_count % h->Q);
    uint8_t *d_raw = h->q_raw[(size_t)slot];
    if ((!resident && x264gpu_memcpy_h2d(d_raw, h->h_in.data(), h->h_in.size(), h->up_stream) != X264GPU_OK) ||
        (resident && d_raw != h->d_in && (x264gpu_memcpy_d2d(d_raw, h->d_in, h->h_in.size(), h->up_stream) != X264GPU_OK || (h->up_stream && x264gpu_stream_sync(h->up_stream) != X264GPU_OK)))) {
        xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: upload failed: %s\n", x264gpu_last_error());
        return -1;
    }
    x264_t::QEntry e = {};
    e.pts = pic_in->i_pts; e.slot = slot; e.img = pic_in->img;
    bool idr = h->la_count == 0 || h->la_gop >= h->keyint || pic_in->i_type == X264_TYPE_IDR || pic_in->i_type == X264_TYPE_KEYFRAME, intra_pic = false;
    if (h->la) {
        if (x264gpu_lookahead_frame_cost(h->la, d_raw, h->la_count == 0, h->d_la, h->mbtree ? h->q_info[(size_t)slot] : nullptr, nullptr) != X264GPU_OK ||
            ((h->mbtree || h->st_aq_costs || h->aq_mode >= 2) && h->aq_strength != 0.f && x264gpu_lookahead_aq_offsets_mode(h->la, d_raw, h->aq_mode >= 2 ? h->aq_mode : 1, h->aq_strength, h->q_aq[(size_t)slot], nullptr) != X264GPU_OK) ||
            x264gpu_memcpy_d2h(e.costs, h->d_la, sizeof(e.costs), nullptr) != X264GPU_OK) {
            xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: lookahead failed: %s\n", x264gpu_last_error());
            return -1;
        }
        if (!idr && p.i_scenecut_threshold > 0 && h->la_count > 0) {
            // scenecut_internal: the bias grows with the distance from the last keyframe.  A cut at or beyond min-keyint becomes
            // an IDR picture, one inside min-keyint an I picture that keeps the references (x264_slicetype_decide).
            const int gop = h->la_gop, kmin = h->keyint_min, kmax = h->keyint;
            const double tmax = p.i_scenecut_threshold / 100.0, tmin = kmin == kmax ? tmax : tmax * 0.25;
            double bias;
            if (gop <= kmin / 4) bias = tmin / 4;
            else if (gop <= kmin) bias = tmin * gop / kmin;
            else bias = tmin + (tmax - tmin) * (gop - kmin) / (kmax - kmin);
            e.scenecut = (double)e.costs[1] >= (1.0 - bias) * (double)e.costs[0];
            if (e.scenecut) { if (gop >= kmin) idr = true; else intra_pic = true; }
        }
    }
    e.type = idr ? 2 : intra_pic ? 1 : 0;
    h->la_gop = idr ? 1 : h->la_gop + 1;
    h->la_count++;
    if (h->dpbmode) {
        x264_t::BEntry be = {};
        be.pts = e.pts; be.frame = (int)(h->la_count - 1); be.slot = slot; be.forced = e.type; be.scenecut = e.scenecut; be.img = e.img;
        memcpy(be.costs, e.costs, sizeof(e.costs));
        if (pic_in->i_type == X264_TYPE_I) be.forced = 1;
        if (h->st) {
            // the slice-type analysis decides keyframes and scene cuts itself: only what the caller forced stays forced
            be.forced = pic_in->i_type == X264_TYPE_IDR || pic_in->i_type == X264_TYPE_KEYFRAME ? 2 : pic_in->i_type == X264_TYPE_I ? 1 : 0;
            be.scenecut = 0;
            bool ok = x264gpu_slicetype_put_frame(h->st, slot, d_raw, nullptr) == X264GPU_OK;
            if (ok && h->weightp) { uint64_t stats[2]; ok = x264gpu_slicetype_pixel_stats(h->st, slot, d_raw, stats, nullptr) == X264GPU_OK; }      // x264_adaptive_quant_frame: i_pixel_sum / i_pixel_ssd
            if (ok && h->mbtree) {
                // x264_adaptive_quant_frame: the AQ offsets weight the lookahead's costs and are what the tree starts from (f_qp_offset = f_qp_offset_aq)
                if (h->aq_strength == 0.f) ok = x264gpu_memset(h->q_aq[(size_t)slot], 0, (size_t)h->nmb * sizeof(float), nullptr) == X264GPU_OK;
                ok = ok && x264gpu_slicetype_set_aq(h->st, slot, h->aq_strength != 0.f ? h->q_aq[(size_t)slot] : nullptr, nullptr) == X264GPU_OK &&
                     x264gpu_memcpy_d2d(h->q_tree[(size_t)slot], h->q_aq[(size_t)slot], (size_t)h->nmb * sizeof(float), nullptr) == X264GPU_OK;
            }
            if (ok && h->st_aq_costs) ok = x264gpu_slicetype_set_aq(h->st, slot, h->q_aq[(size_t)slot], nullptr) == X264GPU_OK;      // i_inv_qscale_factor for i_cost_est_aq
            if (!ok) {
                xlog(&p, X264_LOG_ERROR, "x264_encoder_encode: lookahead failed: %s\n", x264gpu_last_error());
                return -1;
            }
        }
        h->bq.push_back(be);
        h->all_pts.push_back(e.pts);
        PHASE(1);
        const int size = h->inflight > 1 ? encode_bmode_inflight(h, pp_nal, pi_nal, pic_out, false) : encode_bmode(h, pp_nal, pi_nal, pic_out, false);
        PHASE(4);
        h->t_phase[5] += 1;
        return size;
    }
    h->queue.push_back(e);
    PHASE(1);
    if ((int)h->queue.size() <= h->L) return 0;                        // still filling the lookahead: no picture yet (codec.c:1693, size 0)
    const int size = encode_queued(h, pp_nal, pi_nal, pic_out, false);
    PHASE(4);
    h->t_phase[5] += 1;
#undef PHASE
    return size;
}

int x264_encoder_delayed_frames(x264_t *h) { return !h || h->failed ? 0 : h->G > 1 ? (int)(h->submitted - h->emitted) : h->dpbmode ? (int)(h->bq.size() + h->bcoding.size() + h->fl.size()) + (h->defer[0].valid ? 1 : 0) + (h->defer[1].valid ? 1 : 0) : (int)h->queue.size(); }

void x264_encoder_close(x264_t *h)
{
    if (!h) return;
    join_pool(h);
    join_gpu(h);
    for (auto &d : h->defer) if (d.th.joinable()) { d.hurry = true; if (h->batch) { std::lock_guard<std::mutex> lg(h->batch->m); h->batch->cv.notify_all(); } d.th.join(); }
    if (getenv("X264GPU_HOST_TIMING") && h->t_phase[5] > 0)
        fprintf(stderr, "x264gpu host timing, ms per call over %.0f calls: copy-in %.2f, upload+lookahead %.2f, GPU %.2f, download %.2f, entropy %.2f\n", h->t_phase[5],
                1e3 * h->t_phase[0] / h->t_phase[5], 1e3 * h->t_phase[1] / h->t_phase[5], 1e3 * h->t_phase[2] / h->t_phase[5], 1e3 * h->t_phase[3] / h->t_phase[5], 1e3 * h->t_phase[4] / h->t_phase[5]);
    if (getenv("X264GPU_HOST_TIMING") && h->t_b[4] > 0)
        fprintf(stderr, "x264gpu host timing (DPB model), ms per picture over %.0f pictures: slice-type analysis %.2f, GPU hot path %.2f, download %.2f, entropy coding %.2f\n", h->t_b[4],
                1e3 * h->t_b[0] / h->t_b[4], 1e3 * h->t_b[1] / h->t_b[4], 1e3 * h->t_b[2] / h->t_b[4], 1e3 * h->t_b[3] / h->t_b[4]);
    if (h->stat_file) {
        fclose(h->stat_file); h->stat_file = nullptr;
        const std::string out = h->param.rc.psz_stat_out ? h->param.rc.psz_stat_out : "";
        // x264_ratecontrol_delete: a second pass that stopped short of the first one's pictures keeps the complete statistics it read
        if (h->pass2 && h->coded_count < (long)h->p2.size()) { remove((out + ".temp").c_str()); xlog(&h->param, X264_LOG_INFO, "2-pass: %ld of %d pictures coded: the statistics file keeps the first pass' lines\n", h->coded_count, (int)h->p2.size()); }
        else if (!out.empty() && rename((out + ".temp").c_str(), out.c_str())) xlog(&h->param, X264_LOG_ERROR, "failed to rename \"%s.temp\" to \"%s\"\n", out.c_str(), out.c_str());
    }
    for (size_t i = 0; i < h->lctx.size(); i++) {
        x264_t::LaunchCtx &c = h->lctx[i];
        if (c.stream) { x264gpu_stream_sync(c.stream); x264gpu_stream_destroy(c.stream); }
        if (c.ev) x264gpu_event_destroy(c.ev);
        if (i > 0) { if (c.gpu) x264gpu_encoder_destroy(c.gpu); if (c.d_mb) x264gpu_free(c.d_mb); if (c.d_lv) x264gpu_free(c.d_lv); }
    }
    h->lctx.clear();
    if (h->ev_la) x264gpu_event_destroy(h->ev_la);
    if (h->batch) { batch_leave(h->batch, h->batch_idx); h->batch = nullptr; }
    h->up_stream = nullptr;          // (the batch group's)
    if (h->gpu) x264gpu_encoder_destroy(h->gpu);
    if (h->d_in) x264gpu_free(h->d_in);
    if (h->d_mb) x264gpu_free(h->d_mb);
    if (h->d_lv) x264gpu_free(h->d_lv);
    for (auto &dc : h->devs) {
        if (h->devs.size() > 1) (void)x264gpu_set_device(dc.dev);
        if (dc.gpu) x264gpu_encoder_destroy(dc.gpu);
        if (dc.d_mb) x264gpu_free(dc.d_mb);
        if (dc.d_lv) x264gpu_free(dc.d_lv);
        if (dc.d_ring) x264gpu_free(dc.d_ring);
    }
    if (h->devs.size() > 1) (void)x264gpu_set_device(h->device);
    for (void *blk : h->q_block) if (blk) x264gpu_free(blk);          // the queue's slots (source pictures, lookahead records, AQ and tree offsets) are cuts of these
    if (h->d_tree) x264gpu_free(h->d_tree);
    if (h->la) x264gpu_lookahead_destroy(h->la);
    if (h->st) x264gpu_slicetype_destroy(h->st);
    if (h->d_la) x264gpu_free(h->d_la);
    delete h;
}

/* test/diagnostic hooks (not part of the x264 API): entropy-code caller-supplied records, fetch the GPU recon */
int x264host_write_slice(int mbw, int mbh, int slice_type, int qp, int pic_init_qp, int frame_num, int log2_max_frame_num,
                         int idr, int idr_pic_id, int disable_deblock_idc, int num_ref, int num_ref_default, int transform8x8_mode,
                         const x264gpu_mb *mbs, const int16_t *levels, uint8_t *out, int cap, int *skipped)
{
    SliceParams sp = {};
    sp.mbw = mbw; sp.mbh = mbh; sp.slice_type = slice_type; sp.qp = qp; sp.pic_init_qp = pic_init_qp; sp.frame_num = frame_num;
    sp.log2_max_frame_num = log2_max_frame_num; sp.idr = idr; sp.idr_pic_id = idr_pic_id; sp.nal_ref_idc = idr ? 3 : 2;
    sp.num_ref = num_ref; sp.num_ref_default = num_ref_default;
    sp.disable_deblock_idc = disable_deblock_idc; sp.transform8x8_mode = transform8x8_mode;
    std::vector<uint8_t> v;
    SliceStats stt = { 0 };
    write_slice(v, sp, mbs, levels, true, true, &stt, cavlc_threads_default(1));
    if (skipped) *skipped = stt.skip;
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}

// a whole picture of `slices` slices through either writer (tests: multi-slice pictures)
int x264host_write_picture(int mbw, int mbh, int slice_type, int qp, int pic_init_qp, int frame_num, int log2_max_frame_num,
                           int idr, int idr_pic_id, int disable_deblock_idc, int num_ref, int num_ref_default, int transform8x8_mode, int cabac, int slices,
                           const x264gpu_mb *mbs, const int16_t *levels, uint8_t *out, int cap, int *skipped)
{
    SliceParams sp = {};
    sp.mbw = mbw; sp.mbh = mbh; sp.slice_type = slice_type; sp.qp = qp; sp.pic_init_qp = pic_init_qp; sp.frame_num = frame_num;
    sp.log2_max_frame_num = log2_max_frame_num; sp.idr = idr; sp.idr_pic_id = idr_pic_id; sp.nal_ref_idc = idr ? 3 : 2;
    sp.num_ref = num_ref; sp.num_ref_default = num_ref_default;
    sp.disable_deblock_idc = disable_deblock_idc; sp.transform8x8_mode = transform8x8_mode; sp.cabac = cabac;
    sp.slices_plain = slices < 0;                 // -N: N slices as x264's --slices N codes them (filtered across), N: as its slice threads do
    if (slices < 0) slices = -slices;
    std::vector<uint8_t> v;
    SliceStats stt = { 0 };
    write_picture(v, nullptr, sp, slices, mbs, levels, true, true, &stt, 4);
    if (skipped) *skipped = stt.skip;
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}

// the same through the CABAC writer (tests: records from the CPU checker -> bytes -> checker decoder)
int x264host_write_slice_cabac(int mbw, int mbh, int slice_type, int qp, int pic_init_qp, int frame_num, int log2_max_frame_num,
                               int idr, int idr_pic_id, int disable_deblock_idc, int num_ref, int num_ref_default, int transform8x8_mode,
                               const x264gpu_mb *mbs, const int16_t *levels, uint8_t *out, int cap, int *skipped)
{
    SliceParams sp = {};
    sp.mbw = mbw; sp.mbh = mbh; sp.slice_type = slice_type; sp.qp = qp; sp.pic_init_qp = pic_init_qp; sp.frame_num = frame_num;
    sp.log2_max_frame_num = log2_max_frame_num; sp.idr = idr; sp.idr_pic_id = idr_pic_id; sp.nal_ref_idc = idr ? 3 : 2;
    sp.num_ref = num_ref; sp.num_ref_default = num_ref_default;
    sp.disable_deblock_idc = disable_deblock_idc; sp.transform8x8_mode = transform8x8_mode; sp.cabac = 1;
    std::vector<uint8_t> v;
    SliceStats stt = { 0 };
    write_slice(v, sp, mbs, levels, true, true, &stt, 1);
    if (skipped) *skipped = stt.skip;
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}

int x264host_write_headers_cabac(int width, int height, int level_idc, int log2_max_frame_num, int pic_init_qp, int chroma_qp_offset,
                                 uint32_t num_units_in_tick, uint32_t time_scale, int num_ref, int transform8x8_mode, int cabac, uint8_t *out, int cap);
int x264host_write_headers(int width, int height, int level_idc, int log2_max_frame_num, int pic_init_qp, int chroma_qp_offset,
                           uint32_t num_units_in_tick, uint32_t time_scale, int num_ref, int transform8x8_mode, uint8_t *out, int cap)
{
    return x264host_write_headers_cabac(width, height, level_idc, log2_max_frame_num, pic_init_qp, chroma_qp_offset, num_units_in_tick, time_scale, num_ref, transform8x8_mode, 0, out, cap);
}
int x264host_write_headers_cabac(int width, int height, int level_idc, int log2_max_frame_num, int pic_init_qp, int chroma_qp_offset,
                                 uint32_t num_units_in_tick, uint32_t time_scale, int num_ref, int transform8x8_mode, int cabac, uint8_t *out, int cap)
{
    SpsParams s = {};
    s.profile_idc = transform8x8_mode ? 100 : cabac ? 77 : 66; s.level_idc = level_idc; s.mbw = (width + 15) / 16; s.mbh = (height + 15) / 16;
    s.crop_right = s.mbw * 16 - width; s.crop_bottom = s.mbh * 16 - height; s.num_ref_frames = num_ref; s.log2_max_frame_num = log2_max_frame_num;
    s.fullrange = 0; s.colorprim = 2; s.transfer = 2; s.colmatrix = 2; s.vidformat = 5;
    s.num_units_in_tick = num_units_in_tick; s.time_scale = time_scale; s.constraint_set0 = !transform8x8_mode && !cabac; s.constraint_set1 = !transform8x8_mode;
    std::vector<uint8_t> v;
    write_sps(v, s, true);
    PpsParams pp = { 0, 0, cabac, num_ref, pic_init_qp, chroma_qp_offset, transform8x8_mode };
    write_pps(v, pp, true);
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}

/* ---- tests: the DPB model (dpb.hpp) and a slice writer driven by it, for B-picture streams built from the CPU checker's records ---- */
void *x264host_dpb_new(int frame_reference, int bframes, int b_pyramid, int log2_max_frame_num, int weightp)
{
    Dpb *d = new Dpb();
    d->configure(frame_reference, bframes, b_pyramid, log2_max_frame_num, weightp);
    return d;
}
void x264host_dpb_free(void *h) { delete (Dpb *)h; }
int x264host_dpb_info(void *h, int *max_dpb, int *num_reorder) { Dpb *d = (Dpb *)h; *max_dpb = d->max_dpb; *num_reorder = d->num_reorder; return d->slots(); }
/* info[0..7] = frame_num, nal_ref_idc, n_mmco, reorder commands of list 0, of list 1, ... */
int x264host_dpb_plan(void *h, int type, int frame, int n_follow, const int *follow_coded, const int *follow_frame, x264gpu_pic *pic_out, int *info)
{
    const DpbPlan &p = ((Dpb *)h)->plan(type, frame, n_follow, follow_coded, follow_frame);
    *pic_out = p.pic;
    if (info) { info[0] = p.frame_num; info[1] = p.nal_ref_idc; info[2] = p.n_mmco; info[3] = p.reorder[0].n; info[4] = p.reorder[1].n; }
    return 0;
}
/* ... with a luma weight for reference 0 (x264_weights_analyse's result): w = { scale, denom, offset } */
int x264host_dpb_plan_w(void *h, int type, int frame, int n_follow, const int *follow_coded, const int *follow_frame, const int *w, x264gpu_pic *pic_out, int *info)
{
    Dpb::LumaWeight lw;
    if (w) { lw.on = 1; lw.scale = w[0]; lw.denom = w[1]; lw.offset = w[2]; }
    const DpbPlan &p = ((Dpb *)h)->plan(type, frame, n_follow, follow_coded, follow_frame, w ? &lw : nullptr);
    *pic_out = p.pic;
    if (info) { info[0] = p.frame_num; info[1] = p.nal_ref_idc; info[2] = p.n_mmco; info[3] = p.reorder[0].n; info[4] = p.reorder[1].n; }
    return 0;
}
/* ... and chroma weights beside it: w = { scale, denom, offset, chroma denom, Cb on, Cb scale, Cb offset, Cr on, Cr scale, Cr offset } */
int x264host_dpb_plan_wc(void *h, int type, int frame, int n_follow, const int *follow_coded, const int *follow_frame, const int *w, x264gpu_pic *pic_out, int *info)
{
    Dpb::LumaWeight lw;
    lw.on = 1; lw.scale = w[0]; lw.denom = w[1]; lw.offset = w[2]; lw.cdenom = w[3];
    for (int c = 0; c < 2; c++) { lw.con[c] = w[4 + 3 * c]; lw.cscale[c] = w[5 + 3 * c]; lw.coffset[c] = w[6 + 3 * c]; }
    const DpbPlan &p = ((Dpb *)h)->plan(type, frame, n_follow, follow_coded, follow_frame, &lw);
    *pic_out = p.pic;
    if (info) { info[0] = p.frame_num; info[1] = p.nal_ref_idc; info[2] = p.n_mmco; info[3] = p.reorder[0].n; info[4] = p.reorder[1].n; }
    return 0;
}
void x264host_dpb_commit(void *h) { ((Dpb *)h)->commit(); }
void x264host_dpb_set_direct(void *h, int temporal, int auto_write) { ((Dpb *)h)->set_direct(temporal, auto_write); }
int x264host_write_slice_dpb(void *h, int mbw, int mbh, int qp, int pic_init_qp, int log2_max_frame_num, int log2_max_poc_lsb, int idr_pic_id,
                             int disable_deblock_idc, int num_ref_default, int transform8x8_mode, const x264gpu_mb *mbs, const int16_t *levels,
                             uint8_t *out, int cap, int *skipped)
{
    SliceParams sp = {};
    sp.mbw = mbw; sp.mbh = mbh; sp.qp = qp; sp.pic_init_qp = pic_init_qp; sp.log2_max_frame_num = log2_max_frame_num; sp.log2_max_poc_lsb = log2_max_poc_lsb;
    sp.idr_pic_id = idr_pic_id; sp.num_ref_default = num_ref_default; sp.num_ref1_default = 1;
    sp.disable_deblock_idc = disable_deblock_idc; sp.transform8x8_mode = transform8x8_mode; sp.cabac = 1;
    ((Dpb *)h)->fill(sp);
    std::vector<uint8_t> v;
    SliceStats stt = { 0 };
    write_slice(v, sp, mbs, levels, true, true, &stt, 1);
    if (skipped) *skipped = stt.skip;
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}
// the same picture in `slices` slices (N: as x264's slice threads cut it, -N: as --slices N does)
int x264host_write_picture_dpb(void *h, int mbw, int mbh, int qp, int pic_init_qp, int log2_max_frame_num, int log2_max_poc_lsb, int idr_pic_id,
                               int disable_deblock_idc, int num_ref_default, int transform8x8_mode, int slices, const x264gpu_mb *mbs, const int16_t *levels,
                               uint8_t *out, int cap, int *skipped)
{
    SliceParams sp = {};
    sp.mbw = mbw; sp.mbh = mbh; sp.qp = qp; sp.pic_init_qp = pic_init_qp; sp.log2_max_frame_num = log2_max_frame_num; sp.log2_max_poc_lsb = log2_max_poc_lsb;
    sp.idr_pic_id = idr_pic_id; sp.num_ref_default = num_ref_default; sp.num_ref1_default = 1;
    sp.disable_deblock_idc = disable_deblock_idc; sp.transform8x8_mode = transform8x8_mode; sp.cabac = 1;
    sp.slices_plain = slices < 0;
    if (slices < 0) slices = -slices;
    ((Dpb *)h)->fill(sp);
    std::vector<uint8_t> v;
    SliceStats stt = { 0 };
    write_picture(v, nullptr, sp, slices > 1 ? slices : 1, mbs, levels, true, true, &stt, 1);
    if (skipped) *skipped = stt.skip;
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}
int x264host_write_headers_b(int width, int height, int level_idc, int log2_max_frame_num, int pic_init_qp, int chroma_qp_offset, uint32_t num_units_in_tick,
                             uint32_t time_scale, int num_ref_default, int transform8x8_mode, int cabac, int num_ref_frames, int log2_max_poc_lsb, int num_reorder,
                             int weighted_bipred_idc, int weighted_pred, uint8_t *out, int cap)
{
    SpsParams s = {};
    s.profile_idc = transform8x8_mode ? 100 : 77; s.level_idc = level_idc; s.mbw = (width + 15) / 16; s.mbh = (height + 15) / 16;
    s.crop_right = s.mbw * 16 - width; s.crop_bottom = s.mbh * 16 - height; s.num_ref_frames = num_ref_frames; s.log2_max_frame_num = log2_max_frame_num;
    s.fullrange = 0; s.colorprim = 2; s.transfer = 2; s.colmatrix = 2; s.vidformat = 5;
    s.num_units_in_tick = num_units_in_tick; s.time_scale = time_scale; s.constraint_set0 = 0; s.constraint_set1 = !transform8x8_mode;
    s.log2_max_poc_lsb = log2_max_poc_lsb; s.num_reorder_frames = num_reorder;
    std::vector<uint8_t> v;
    write_sps(v, s, true);
    PpsParams pp = { 0, 0, cabac, num_ref_default, pic_init_qp, chroma_qp_offset, transform8x8_mode };
    pp.weighted_bipred_idc = weighted_bipred_idc; pp.weighted_pred = weighted_pred;
    write_pps(v, pp, true);
    if ((int)v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}

/* device pointer of the encoder's input staging buffer: a tight I420 picture (Y w*h, U, V).  A picture whose plane[0]
 * equals this pointer is encoded in place, without the host copy-in and upload (used by the VfW shell, vfw.cpp). */
uint8_t *x264gpu_host_input_i420(x264_t *h) { return h ? h->d_in : nullptr; }

int x264host_last_decision(x264_t *h, int *qp, int *scenecut, int32_t costs[4])
{
    if (!h) return -1;
    if (qp) *qp = h->last_qp;
    if (scenecut) *scenecut = h->last_scenecut;
    if (costs) memcpy(costs, h->last_costs, sizeof(h->last_costs));
    return 0;
}

/* tests: the second pass' plan — the quantiser scale init_pass2 gave every picture of the statistics file (display order), and what should have been spent before each;
 * returns the number of pictures planned (0: not a second pass) */
int x264host_pass2_plan(x264_t *h, double *new_qscale, double *expected_bits, int n)
{
    if (!h || !h->pass2) return 0;
    for (int i = 0; i < n && i < (int)h->p2.size(); i++) { if (new_qscale) new_qscale[i] = h->p2[(size_t)i].new_qscale; if (expected_bits) expected_bits[i] = h->p2[(size_t)i].expected_bits; }
    return (int)h->p2.size();
}
/* tests: the float quantiser (x264 rc->qpm) the last coded picture's macroblock quantisers were rounded from; 0 = its integer quantiser */
float x264host_last_qpm(x264_t *h) { return h ? h->last_qpm : 0.f; }

int x264host_pictures_in_flight(x264_t *h) { return h ? h->inflight : 0; }

int x264host_get_recon(x264_t *h, uint8_t *i420_out)
{
    if (!h || !h->gpu) return -1;
    join_gpu(h);                             // a pipelined session: this is the picture whose GPU stage ran last, not the one handed back last
    size_t n = (size_t)h->param.i_width * h->param.i_height * 3 / 2;
    uint8_t *d = nullptr;
    if (x264gpu_malloc((void **)&d, n) != X264GPU_OK) return -1;
    // (several pictures in flight: the picture handed back last — its slot is not reused before the next call issues a picture)
    int rc = h->inflight > 1 && h->last_retired_slot >= 0 ? x264gpu_encoder_get_recon_slot(h->gpu, 0, h->last_retired_slot, d, nullptr) : x264gpu_encoder_get_recon(h->gpu, 0, d, nullptr);
    if (rc == X264GPU_OK) rc = x264gpu_memcpy_d2h(i420_out, d, n, nullptr);
    x264gpu_free(d);
    return rc;
}

}  /* extern "C" */
