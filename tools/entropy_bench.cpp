// tools/entropy_bench.cpp — the host's slice writers (host/cabac.cpp, host/slice.cpp) timed on records the device produced: the files X264GPU_DUMP_RECORDS writes
// (x264gpu_pic, records, levels, quantiser offsets of one coded picture).  Prints, per file, the picture type, the bytes of the slice NAL unit, a checksum of them
// and the milliseconds of one write_picture call (best of `reps`) — the regression check and the stopwatch for work on the entropy coder.
//   entropy_bench <width> <height> <reps> pic0000.bin [pic0001.bin ...]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "../x264vfw_amd/host/host.hpp"

using namespace x264host;

static uint64_t fnv(const std::vector<uint8_t> &v) { uint64_t h = 1469598103934665603ull; for (uint8_t b : v) { h ^= b; h *= 1099511628211ull; } return h; }

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: entropy_bench W H reps pic.bin ...\n"); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), reps = atoi(argv[3]);
    const int mbw = (w + 15) / 16, mbh = (h + 15) / 16; const size_t nmb = (size_t)mbw * mbh;
    double total = 0;
    for (int a = 4; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        x264gpu_pic pic;
        std::vector<x264gpu_mb> mb(nmb);
        std::vector<int16_t> lv(nmb * X264GPU_MB_LEVELS);
        if (fread(&pic, sizeof(pic), 1, f) != 1 || fread(mb.data(), sizeof(x264gpu_mb), nmb, f) != nmb || fread(lv.data(), sizeof(int16_t), lv.size(), f) != lv.size()) { fprintf(stderr, "%s: short file\n", argv[a]); return 2; }
        fclose(f);
        SliceParams sp = {};
        sp.mbw = mbw; sp.mbh = mbh; sp.qp = pic.qp; sp.pic_init_qp = 26; sp.log2_max_frame_num = 4; sp.log2_max_poc_lsb = 6;
        sp.slice_type = pic.slice_type == X264GPU_SLICE_I_NONIDR ? X264GPU_SLICE_I : pic.slice_type;
        sp.idr = pic.slice_type == X264GPU_SLICE_I; sp.nal_ref_idc = pic.keep ? 2 : 0; sp.poc = pic.poc;
        sp.num_ref = pic.nref[0]; sp.num_ref1 = pic.nref[1]; sp.num_ref_default = 3; sp.num_ref1_default = 1;
        sp.direct_spatial = !pic.direct_temporal; sp.transform8x8_mode = 1; sp.cabac = 1;
        double best = 1e9; std::vector<uint8_t> out; SliceStats st = { 0 };
        for (int r = 0; r < reps; r++) {
            out.clear();
            const auto t0 = std::chrono::steady_clock::now();
            write_picture(out, nullptr, sp, 1, mb.data(), lv.data(), true, true, &st, 1);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms < best) best = ms;
        }
        printf("%s: slice type %d, %zu bytes, fnv %016llx, %.2f ms\n", argv[a], pic.slice_type, out.size(), (unsigned long long)fnv(out), best);
        total += best;
    }
    printf("total %.2f ms\n", total);
    return 0;
}
