// tools/multi_session.cpp — N x264_encoder_open sessions on N host threads through the cross-session batcher (X264GPU_BATCH), the way N instances of the
// reference's driver would call the library (driverproc.c:110-128 one CODEC per stream; codec.c:1463,1623,1693,1848-1857 the call sequence): host pictures
// in, Annex-B out, every thread feeds and drains its own session.  bench.py's `e2e.multi_session_*` leg runs this instead of Python threads: 2048 Python
// threads hand the interpreter lock to each other between the calls, which is the harness's time, not the library's.
//
//   multi_session <frames.yuv> <width> <height> <source frames in the file> <sessions> <frames each> <qp>      -> one JSON line on stdout
//
// Measures what bench.py's Python leg measures: all sessions' frames / (first open .. last close), and the spans setup (until every session is open: no
// picture can be coded before), coding, teardown.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "x264.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: multi_session frames.yuv W H nsrc sessions frames_each qp\n"); return 2; }
    const char *path = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]), nsrc = atoi(argv[4]), ns = atoi(argv[5]), n = atoi(argv[6]);
    const std::string qp = argv[7];
    const size_t fsz = (size_t)w * h * 3 / 2;
    std::vector<uint8_t> src(fsz * (size_t)nsrc);
    {
        FILE *f = fopen(path, "rb");
        if (!f || fread(src.data(), 1, src.size(), f) != src.size()) { fprintf(stderr, "multi_session: cannot read %d frames from %s\n", nsrc, path); return 2; }
        fclose(f);
    }
    setenv("X264GPU_BATCH", std::to_string(ns).c_str(), 1);
    std::vector<double> opened((size_t)ns, 0.0), coded((size_t)ns, 0.0);
    std::vector<long> bytes((size_t)ns, 0);
    std::atomic<int> errors{ 0 };
    // MULTI_SESSION_TRACE=1: when the LAST session came back from call k (seconds from the start), and the longest any session spent inside call k
    const bool trace = getenv("MULTI_SESSION_TRACE") != nullptr;
    std::vector<double> ret_at((size_t)ns * (size_t)(n + 8), 0.0), spent((size_t)ns * (size_t)(n + 8), 0.0);
    const double t0 = now();
    auto one = [&](int idx) {
        x264_param_t p;
        if (x264_param_default_preset(&p, "medium", nullptr) != 0) { errors++; return; }
        p.i_width = w; p.i_height = h; p.i_csp = X264_CSP_I420; p.i_fps_num = 25; p.i_fps_den = 1; p.i_log_level = X264_LOG_NONE;
        const char *opts[][2] = { { "qp", qp.c_str() }, { "keyint", "250" }, { "scenecut", "0" }, { "b-adapt", "0" }, { "threads", "1" } };
        for (auto &o : opts) if (x264_param_parse(&p, o[0], o[1]) != 0) { errors++; return; }
        p.b_annexb = 1; p.b_repeat_headers = 1;
        x264_t *e = x264_encoder_open(&p);
        if (!e) { errors++; return; }
        opened[(size_t)idx] = now();
        x264_picture_t pic, out;
        if (x264_picture_alloc(&pic, X264_CSP_I420, w, h) != 0) { errors++; x264_encoder_close(e); return; }
        x264_nal_t *nal = nullptr; int nn = 0, got = 0;
        long total = 0;
        for (int i = 0; i < n; i++) {
            const uint8_t *f = src.data() + fsz * (size_t)(i % nsrc);
            const size_t off[3] = { 0, (size_t)w * h, (size_t)w * h * 5 / 4 }, sz[3] = { (size_t)w * h, (size_t)w * h / 4, (size_t)w * h / 4 };
            for (int pl = 0; pl < 3; pl++) {
                const int pw = pl ? w / 2 : w, ph = pl ? h / 2 : h;
                if (pic.img.i_stride[pl] == pw) memcpy(pic.img.plane[pl], f + off[pl], sz[pl]);
                else for (int y = 0; y < ph; y++) memcpy(pic.img.plane[pl] + (size_t)y * pic.img.i_stride[pl], f + off[pl] + (size_t)y * pw, (size_t)pw);
            }
            pic.i_pts = i;
            const double tc = trace ? now() : 0;
            const int size = x264_encoder_encode(e, &nal, &nn, &pic, &out);
            if (trace) { ret_at[(size_t)idx * (size_t)(n + 8) + (size_t)i] = now(); spent[(size_t)idx * (size_t)(n + 8) + (size_t)i] = now() - tc; }
            if (size < 0) { errors++; break; }
            if (size > 0) { got++; total += size; }
        }
        while (!errors && x264_encoder_delayed_frames(e)) {
            const int size = x264_encoder_encode(e, &nal, &nn, nullptr, &out);
            if (size <= 0) { errors++; break; }
            got++; total += size;
        }
        coded[(size_t)idx] = now();
        if (got != n) errors++;
        bytes[(size_t)idx] = total;
        x264_picture_clean(&pic);
        x264_encoder_close(e);
    };
    std::vector<std::thread> th;
    th.reserve((size_t)ns);
    for (int i = 0; i < ns; i++) th.emplace_back(one, i);
    for (auto &t : th) t.join();
    const double t1 = now();
    if (trace)
        for (int i = 0; i < n; i++) {
            double last = 0, longest = 0, first = 1e30;
            for (int k = 0; k < ns; k++) { const double r = ret_at[(size_t)k * (size_t)(n + 8) + (size_t)i], d = spent[(size_t)k * (size_t)(n + 8) + (size_t)i]; if (r > last) last = r; if (r < first) first = r; if (d > longest) longest = d; }
            fprintf(stderr, "call %d: first session back at %.2f s, last at %.2f s, longest call %.2f s\n", i, first - t0, last - t0, longest);
        }
    if (errors) { printf("{\"error\": \"%d sessions failed\"}\n", errors.load()); return 1; }
    double last_open = 0, last_coded = 0; long tot = 0;
    for (int i = 0; i < ns; i++) { if (opened[(size_t)i] > last_open) last_open = opened[(size_t)i]; if (coded[(size_t)i] > last_coded) last_coded = coded[(size_t)i]; tot += bytes[(size_t)i]; }
    printf("{\"fps\": %.2f, \"kB_per_frame\": %.1f, \"setup_s\": %.2f, \"coding_s\": %.2f, \"teardown_s\": %.2f, \"fps_coding_span\": %.2f, \"sessions\": %d, \"frames_each\": %d, \"driver\": \"c++ (tools/multi_session.cpp)\"}\n",
           (double)ns * n / (t1 - t0), (double)tot / ((double)ns * n) / 1e3, last_open - t0, last_coded - last_open, t1 - last_coded, (double)ns * n / (last_coded - last_open), ns, n);
    return 0;
}
