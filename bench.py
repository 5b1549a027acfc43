#!/usr/bin/env python3
"""bench.py — 1080p yuv420p frames/sec of the MI355X encode hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
      N > 1 without a launcher: this script starts `python -m torch.distributed.run --nproc-per-node N` on itself (before any
      GPU call) and exits with its code; under a launcher (WORLD_SIZE set) it is one rank of N.

A *step* is one lock-step pass of the hot path over one batch of synthetic input: every one of the `--streams` independent
closed-GOP streams on this GPU advances by one CODED picture through the C ABI x264gpu_encode_pictures() of libx264gpu.so:
ingest -> per-macroblock quantisers -> the macroblock loop in x264's own raster order (one wavefront per stream: motion search
with neighbour predictors in one or both reference lists, spatial direct / bi-prediction in B pictures, intra analysis on
reconstructed neighbours, RD mode decision priced on the live CABAC state, trellis) -> deblock wavefront -> half-pel planes of
the pictures that are kept as references.  Inputs are resident in HBM before the timed region: W + K DISTINCT frames per
sequence (no clip replay), scene cut every 97 frames.  The streams run preset medium's picture structure (bframes 3, b-pyramid
normal, closed GOPs) in coding order: a first short GOP of W pictures is the warmup, the timed K steps start on the IDR picture
of the next GOP — I, P, B-reference and b pictures all lie inside the timed window (`config.workload` counts them).  Streams are
independent (BASELINE.json config 5), so N GPUs shard streams one set per GPU with no collective in the data path ("scaling": "weak").

WHAT IS MEASURED IS x264's preset=medium TOOLSET AS THE DEVICE RUNS IT: CABAC, ref 3 + mixed refs, hex, subme 7 (RD mode decision with
CABAC sizes, psy-rd 1.0), trellis 1, 8x8dct, all partitions, B pictures (spatial direct, weightb, b-pyramid) — `config.toolset_gaps`
lists what still differs (weightp's fade analysis, b-adapt 1, the lookahead-driven tools).  `--bframes 0` measures the I / P stream of the earlier
rounds, `--rd cavlc` medium --no-cabac, `--rd off` the subme-5 toolset, for comparison.

Rank 0 prints ONE JSON line with `roofline` (the macroblock kernel, HIP-event timed inside the timed region, HBM fraction + VALU
issue utilisation from the committed PMC profile), `cpu_baseline` (the oracle restatement on 1 and on all host cores, plus a
run-time probe for a real libx264 / x264 / ffmpeg), and `e2e` (ONE stream through x264_encoder_encode: PCIe + host entropy coding
included).
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TOOLSET_GAPS = "x264 medium with a fixed picture structure and constant quantisers: the lock-step batch codes the same picture type in every stream, so the lookahead's DECISIONS (b-adapt 1, scenecut, fade weights, AQ / mbtree / rate control) are not taken — its device work for the same streams is timed beside `value` (`lookahead`: lowres, AQ offsets, frame costs, macroblock-tree; `value_with_lookahead`); --weightp 2's duplicate reference with offset -1 on every P picture IS in; entropy coding (CABAC bitstream writing) runs on host threads and is outside `value` (inside `e2e`)"
TOOLSET_GAPS_NOB = "x264 medium minus: B-frames (bframes 3 -> 0), the fade analysis of weightp 2; entropy coding (CABAC/CAVLC bitstream writing) runs on host threads and is outside `value` (inside `e2e`)"
TOOLSET_GAPS_NORD = "x264 medium minus: RD mode decision + psy-rd (subme 7 -> 5), trellis 1, the lookahead's decisions (b-adapt 1, fade weights, rate control); entropy coding runs on host threads and is outside `value`"



def cpu_quota():
    """CPUs of time the control group grants this process (cpu.max "quota period"), None when uncapped: the pool's GPU boxes show 256 hardware threads and grant 16 —
    what `cpu_baseline.cores` processes and the host side of the e2e legs really share"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bframes", type=int, default=3, help="B pictures between references (medium: 3, with b-pyramid normal and weightb); 0 = the I / P stream of the earlier rounds")
    ap.add_argument("--weightp", type=int, default=2, choices=[0, 2], help="2 (medium): x264's blind duplicate of reference 0 with luma offset -1 on every P picture with two or more references")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--streams", type=int, default=2048, help="independent closed-GOP streams per GPU (lock-step batch; one wavefront each)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic sequences generated per GPU (replicated up to --streams)")
    ap.add_argument("--clip", type=int, default=0, help="(unused: every stream holds warmup + steps distinct frames)")
    ap.add_argument("--keyint", type=int, default=60)
    ap.add_argument("--qp", type=int, default=23)
    ap.add_argument("--refs", type=int, default=3, help="reference frames (medium: 3)")
    ap.add_argument("--preset", default="medium", choices=["medium", "ultrafast", "slow"])
    ap.add_argument("--subme", type=int, default=0, choices=[0, 8, 9], help="with --preset slow: 9 = BASELINE.json configs[3]'s level (RD refinement in B slices too, deblock-aware RD); default: the preset's own (slow: 8)")
    ap.add_argument("--rd", default="cabac", choices=["cabac", "cavlc", "off"], help="RD mode decision (subme 7, psy-rd 1.0) with the sizes of medium's CABAC (default, the headline), of CAVLC (medium --no-cabac), or off (subme 5: SATD decisions)")
    ap.add_argument("--no-trellis", action="store_true", help="headline toolset without trellis 1 (for comparison)")
    ap.add_argument("--aq", action="store_true", help="variance AQ on (per-macroblock quantisers, the CRF / ABR path); the headline metric is CQP and leaves it off, as x264 does")
    ap.add_argument("--cpu-frames", type=int, default=10, help="frames per core of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--e2e-legs", default="all", choices=["all", "sessions"], help="'sessions': only the threads-1 and multi-session legs of the end-to-end sample")
    ap.add_argument("--e2e-sessions", type=int, default=2048, help="sessions of the multi-session end-to-end sample (cross-session batcher; 0 = skip)")
    ap.add_argument("--e2e-frames", type=int, default=10, help="frames of the single-stream threads-1 end-to-end sample (0 = skip e2e)")
    ap.add_argument("--cpu-procs", type=int, default=64, help="processes of the many-core leg of the CPU baseline (cpu_baseline.cores reports what was used)")
    ap.add_argument("--cpu-frames-all", type=int, default=3, help="frames per core of the every-core leg of the CPU baseline (shorter: it runs one process per core)")
    ap.add_argument("--lookahead", type=int, default=1, help="1: also time the lookahead's device work (lowres, AQ, frame costs, macroblock-tree) for the same streams and report value_with_lookahead")
    ap.add_argument("--content", default="noise", choices=["noise", "smooth", "survey"], help="synthetic content: 'noise' (default, the headline) = moving rectangles of per-pixel "
                    "white noise + strong sensor noise, harder than camera material; 'smooth' = the same scene with band-limited textures and light noise; "
                    "'survey' = SURVEY.md §8(d) read with natural textures: gradient + 3 moving band-limited rectangles + noise of +-4 on luma, +-2 on chroma")
    ap.add_argument("--survey-leg", type=int, default=1, help="1 (default): time the same job on --content survey too and report it as value_survey_content in the same line; 0 = skip")
    ap.add_argument("--cpu-worker", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args()


def toolset(args):
    """config.c:1460-1498 preset deltas restricted to what the pipeline implements"""
    t = {"medium": dict(refs=args.refs, subme=5, deblock=1, partitions=7, dct8x8=1, me_method=1, chroma_me=1, mixed_refs=1),
         "ultrafast": dict(refs=1, subme=0, deblock=0, partitions=0x100, dct8x8=0, me_method=0, chroma_me=0, mixed_refs=0),
         "slow": dict(refs=4, subme=5, deblock=1, partitions=7, dct8x8=1, me_method=2, chroma_me=1, mixed_refs=1)}[args.preset]
    t = dict(t, fast_pskip=1, mv_range=512, cabac=0 if args.preset == "ultrafast" else 1)          # medium's entropy coder (it runs on the host; the analysis costs know it)
    if args.rd != "off" and args.preset != "ultrafast":
        t = dict(t, cabac=int(args.rd == "cabac"), rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)            # x264 lowers the chroma offset by 2 under psy-rd >= 0.25
        if args.rd == "cabac" and not args.no_trellis:
            t = dict(t, trellis=63)                               # medium's --trellis 1: every quantiser call of the final encode
        if args.preset == "slow" and args.rd == "cabac":
            # config.c:1482-1484: slow = --direct auto --rc-lookahead 50 --ref 5 --subme 8 --trellis 2 (+ --me umh above); subme 8 = RD refinement of the
            # P partitions' vectors and of the intra modes (rd 63: every site); --direct auto stays spatial in the lock-step batch (sessions run it)
            t = dict(t, refs=5, subme=8, rd=63)
            if args.subme == 9:          # configs[3]: slow + --me umh + --subme 9 — i_mbrd 2 in B slices as well, h->mb.b_deblock_rdo, chroma in the B slices' sub-pel costs
                t = dict(t, subme=9, rd=63 | 64)
            if not args.no_trellis:
                t = dict(t, trellis=127)
    if args.aq:
        t = dict(t, aq_mode=1, aq_strength=1.0397)
    if args.bframes and args.preset != "ultrafast":
        # (B slices' RD decisions count CABAC sizes or CAVLC bits; with --rd off they are analysed without RD, as x264 does below subme 7)
        t = dict(t, dpb=max(t["refs"], 4 if args.bframes > 1 else 2), weightb=1)        # x264: sps num_ref_frames = max(ref, 4 under b-pyramid, 1 + reorder depth)
    else:
        args.bframes = 0
    return t


# ---------------------------------------------------------------------------------------------------------------------------
# CPU side: runs BEFORE this process touches the GPU (child processes are started from a GPU-free parent)
def cpu_worker(args):
    """one core: the oracle restatement on `cpu_frames` frames of the bench workload; prints seconds"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from x264vfw_amd.synth import synth_frames
    from x264vfw_amd import gop
    from x264vfw_amd import host_api as HL           # the product's DPB model (host/dpb.hpp) plans the reference lists; no GPU call behind it
    frames = synth_frames(args.width, args.height, args.cpu_frames, seed=0x264 + args.cpu_worker, scene_len=97)
    tools = toolset(args)
    enc = O.OracleEncoder(O.default_config(args.width, args.height, qp_i=max(0, args.qp - 3), qp_p=args.qp, **tools))
    order = gop.schedule(display_types(args.cpu_frames, args.bframes, args.cpu_frames), 1)
    dpb = gop.HostDpb(HL, tools["refs"], args.bframes, 1, weightp=args.weightp)
    t0 = time.perf_counter()
    for k, (disp, pt) in enumerate(order):
        pic, _ = dpb.plan(pt, disp, gop.follow_of(order, k))
        pic.qp = qp_of(args, pt)
        enc.encode_pic(np.ascontiguousarray(frames[disp]), pic)
        dpb.commit()
    print(json.dumps({"seconds": time.perf_counter() - t0}), flush=True)


def display_types(n, bframes, gop_len):
    """picture types of n display frames: closed GOPs of gop_len frames, runs of `bframes` B pictures between references, never a B picture last"""
    t = []
    for i in range(n):
        g = i % gop_len
        last = g == gop_len - 1 or i == n - 1
        t.append("I" if g == 0 else "P" if (g % (bframes + 1) == 0 or last) else "B")
    return "".join(t)


def qp_of(args, pt):
    """constant quantisers as x264 derives them (--ipratio 1.4 -> -3, --pbratio 1.3 -> +2; a B reference sits between P and B)"""
    return max(0, args.qp - 3) if pt <= 1 else args.qp if pt == 2 else args.qp + 2 if pt == 4 else args.qp + 1


def x264_probe(args):
    """Is a real x264 reachable on this box (libx264.so*, the x264 CLI, ffmpeg with libx264)?  If the CLI is, time preset=medium
    on a bounded raw sample with 1 and with all threads.  Nothing of the kind ships in the image: normally {"found": false}."""
    import ctypes.util
    libs = sorted(set(glob.glob("/usr/lib*/**/libx264.so*", recursive=True) + glob.glob("/usr/local/lib*/libx264.so*") + glob.glob("/opt/**/libx264.so*", recursive=True)))
    fl = ctypes.util.find_library("x264")
    cli, ff = shutil.which("x264"), shutil.which("ffmpeg")
    ff_has = False
    if ff:
        try:
            ff_has = b"libx264" in subprocess.run([ff, "-hide_banner", "-encoders"], capture_output=True, timeout=20).stdout
        except Exception:
            ff_has = False
    out = {"found": bool(libs or fl or cli or ff_has), "libx264": libs[:3] or fl, "x264_cli": cli, "ffmpeg_libx264": bool(ff_has)}
    if cli:
        try:
            import numpy as np
            from x264vfw_amd.synth import synth_frames
            n = 24
            raw = "/tmp/bench_x264_probe.yuv"
            with open(raw, "wb") as f:
                for fr in synth_frames(args.width, args.height, n, seed=0x264, scene_len=97):
                    f.write(np.ascontiguousarray(fr).tobytes())
            for thr in (1, os.cpu_count() or 1):
                t0 = time.perf_counter()
                subprocess.run([cli, "--preset", "medium", "--qp", str(args.qp), "--threads", str(thr), "--input-res", f"{args.width}x{args.height}",
                                "--fps", "25", "-o", "/tmp/bench_x264_probe.264", raw], capture_output=True, timeout=300, check=True)
                out[f"x264_medium_fps_threads{thr}"] = round(n / (time.perf_counter() - t0), 2)
            os.remove(raw)
        except Exception as e:      # the probe never fails the bench
            out["error"] = repr(e)[:200]
    return out


def cpu_baseline(args):
    """the oracle (kind "port") on 1 core and on every host core (one stream per core), bounded sample"""
    ncpu = min(os.cpu_count() or 1, max(1, args.cpu_procs))          # a bounded sample: the all-cores leg uses at most --cpu-procs processes
    base = [sys.executable, os.path.abspath(__file__), "--width", str(args.width), "--height", str(args.height), "--qp", str(args.qp),
            "--keyint", str(args.keyint), "--refs", str(args.refs), "--preset", args.preset, "--bframes", str(args.bframes)] + (["--aq"] if args.aq else []) + ["--rd", args.rd] + (["--no-trellis"] if args.no_trellis else [])
    nall = max(2, min(args.cpu_frames, args.cpu_frames_all))

    def run(n, frames):
        t0 = time.perf_counter()
        ps = [subprocess.Popen(base + ["--cpu-frames", str(frames), "--cpu-worker", str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for i in range(n)]
        secs = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1])["seconds"] for p in ps]
        return n * frames / max(secs), time.perf_counter() - t0
    f1, w1 = run(1, args.cpu_frames)
    fn, wn = run(ncpu, nall) if ncpu > 1 else (f1, w1)
    return {"value": round(fn, 3), "unit": "frames/s", "cores": ncpu, "cpu_quota": cpu_quota(), "kind": "port",
            "value_1core": round(f1, 3),
            "sample": f"{args.width}x{args.height}, one GOP in coding order ({display_types(args.cpu_frames, args.bframes, args.cpu_frames)}), oracle/analyse.c + encoder.c: one process alone on {args.cpu_frames} frames ({w1:.1f} s wall), then {ncpu} processes side by side (of {os.cpu_count()} hardware threads), "
                      f"{ncpu} streams at once, {nall} frames each ({wn:.1f} s wall) — the builder's own CPU restatement, NOT x264",
            "x264_probe": x264_probe(args)}


# ---------------------------------------------------------------------------------------------------------------------------
def synth_batch(torch, streams, frames, w, h, seed, device, smooth=False, scene_len=0, luma_noise=None):
    """[frames, streams, w*h*3/2] uint8 I420 on the device: gradient + 3 moving textured rectangles +
    per-pixel noise, generated with torch ops (plumbing only)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy = torch.arange(h, device=device).view(h, 1)
    xx = torch.arange(w, device=device).view(1, w)
    out = torch.empty((frames, streams, w * h * 3 // 2), dtype=torch.uint8, device=device)
    vel = ((3, 1), (-2, 2), (5, -3))
    for s in range(streams):
        def new_scene(idx):
            base = 90 + 7 * ((s + 3 * idx) % 8)
            grad = (base + xx * 60 // w + yy * 50 // h).to(torch.int16)
            tex = [torch.randint(0, 256, (h // 3, w // 3), generator=g, device=device, dtype=torch.int16) for _ in vel]
            if smooth:       # band-limited textures: two 7x7 box blurs of the noise, contrast restored
                F = torch.nn.functional
                for i, t in enumerate(tex):
                    f = t.float()[None, None]
                    for _ in range(2):
                        f = F.avg_pool2d(F.pad(f, (3, 3, 3, 3), mode="reflect"), 7, stride=1)
                    f = (f - f.mean()) * 6 + 128
                    tex[i] = f[0, 0].clamp(0, 255).to(torch.int16)
            pos = [(int(torch.randint(0, w, (1,), generator=g, device=device)), int(torch.randint(0, h, (1,), generator=g, device=device))) for _ in vel]
            return grad, tex, pos
        grad, tex, pos = new_scene(0)
        for n in range(frames):
            if scene_len and n and n % scene_len == 0:
                grad, tex, pos = new_scene(n // scene_len)          # hard scene change
            y = grad.clone()
            u = torch.full((h // 2, w // 2), 118, dtype=torch.int16, device=device)
            v = torch.full((h // 2, w // 2), 134, dtype=torch.int16, device=device)
            for (vx, vy), t, (px, py) in zip(vel, tex, pos):
                th, tw = t.shape
                ys = (torch.arange(th, device=device) + py + vy * n) % h
                xs = (torch.arange(tw, device=device) + px + vx * n) % w
                y[ys.view(-1, 1), xs.view(1, -1)] = 40 + t * 150 // 255
                u[(ys[::2] // 2).view(-1, 1), (xs[::2] // 2).view(1, -1)] = 100 + t[::2, ::2] * 40 // 255
                v[(ys[::2] // 2).view(-1, 1), (xs[::2] // 2).view(1, -1)] = 150 - t[::2, ::2] * 40 // 255
            na = luma_noise if luma_noise is not None else 1 if smooth else 4
            y = (y + torch.randint(-na, na + 1, y.shape, generator=g, device=device, dtype=torch.int16)).clamp_(16, 235)
            u = (u + torch.randint(-2, 3, u.shape, generator=g, device=device, dtype=torch.int16)).clamp_(16, 240)
            v = (v + torch.randint(-2, 3, v.shape, generator=g, device=device, dtype=torch.int16)).clamp_(16, 240)
            out[n, s] = torch.cat([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).to(torch.uint8)
    return out


def pmc_evidence(kernel_substr, avg_launch_ms, streams_per_launch, content):
    """HBM bytes per launch of the dominant kernel and its VALU instruction count from the committed rocprofv3 --pmc passes of this
    command (tools/profile_round.sh -> profiles/r05_pmc_per_launch.json; FETCH_SIZE and WRITE_SIZE in separate passes, KB -> bytes,
    read side doubled as MI355X_MICROARCH.md prescribes for gfx950).  The macroblock loop runs as three instantiations (I, P, B slices): the
    profile's entry `_mb_loop_timed_window` is their average over the launches of the timed window, the same launches bench.py's own event
    timing averages.  PMC counters cannot be read from inside an un-profiled run, so these are the profile's figures scaled per stream; null
    without a profile of this workload."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_per_launch.json") for r in (6, 5, 4)) if os.path.exists(q)), None)      # the newest round's profile
    if not path:
        return None, None, None
    tab = json.load(open(path))
    per = tab.get("_workload", {}).get("streams_per_launch")
    if tab.get("_workload", {}).get("content", "noise") != content:
        return None, None, None                               # counters of another workload say nothing about this one
    t = tab.get("_mb_loop_timed_window") if "k_mb_slice" in kernel_substr else next((tab[k] for k in tab if kernel_substr in k), None)
    if not per or not t:
        return None, None, None
    scale = streams_per_launch / per
    traffic = int(t["hbm_bytes_per_launch_corrected"] * scale) if "hbm_bytes_per_launch_corrected" in t else None
    valu = None
    if "SQ_INSTS_VALU" in t and avg_launch_ms > 0:
        simds, clk = 256 * 4, 2.4e9                      # a wave64 VALU op occupies its SIMD16 for 4 cycles
        insts = t["SQ_INSTS_VALU"] * scale
        valu = {"insts_per_launch": int(insts), "insts_per_macroblock": round(t["SQ_INSTS_VALU"] / t.get("macroblocks_per_launch", 1), 1) if t.get("macroblocks_per_launch") else None,
                "issue_util": round(insts * 4 / (simds * clk * avg_launch_ms * 1e-3), 4), "source": "profiles/" + os.path.basename(path)}
    return traffic, valu, "profiles/" + os.path.basename(path)


def loaded_product_libraries():
    """Real paths of the product libraries this process has mapped (/proc/self/maps), for the bench line: what was measured."""
    found = {}
    try:
        for line in open("/proc/self/maps"):
            path = line.split(None, 5)[5].strip() if len(line.split(None, 5)) == 6 else ""
            base = os.path.basename(path)
            if base.startswith("libx264gpu") and base.endswith(".so"):
                found[base] = os.path.realpath(path)
    except OSError:
        pass
    return found


def csp_probe(torch, lib, dev, W, H, frames=64, iters=10):
    """next-row f1 evidence (not part of `value`): BGRA -> I420 ingest of `frames` 1080p pictures per launch, timed with
    events on the current stream; a pure streaming kernel, so its HBM fraction is the meaningful one."""
    csp = 9                                              # X264GPU_CSP_BGRA
    off, st = (C.c_long * 3)(), (C.c_int * 3)()
    n = lib.x264gpu_csp_img_fill(csp, W, H, off, st)
    d_src = torch.randint(0, 256, (frames, n), dtype=torch.uint8, device=dev)
    osz = W * H * 3 // 2
    d_dst = torch.empty((frames, osz), dtype=torch.uint8, device=dev)
    src = (C.c_void_p * 3)(d_src.data_ptr(), d_src.data_ptr(), d_src.data_ptr())
    dst = (C.c_void_p * 3)(d_dst.data_ptr(), d_dst.data_ptr() + W * H, d_dst.data_ptr() + W * H + (W // 2) * (H // 2))
    dstr = (C.c_int * 3)(W, W // 2, W // 2)
    cur = torch.cuda.current_stream(dev).cuda_stream

    def run():
        lib.check(lib.x264gpu_csp_to_i420_batch(src, st, n, csp, W, H, 0, 0, dst, dstr, osz, frames, cur), "csp batch")
    run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    gbs = (n + osz) * frames / (ms * 1e-3) / 1e9
    return {"kernel": "k_csp_bgr<4> (BGRA -> I420)", "frames_per_launch": frames, "avg_launch_ms": round(ms, 4), "alg_bytes_per_frame": n + osz,
            "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "frames_per_s": round(frames / (ms * 1e-3), 1)}


def lookahead_probe(torch, lib, dev, args, data, first, K, S, W, H):
    """The DEVICE work of x264's lookahead for the same streams and pictures as the timed window, run beside it (the lock-step batch codes a fixed
    picture structure, so the decisions themselves are discarded): per display picture x264_frame_init_lowres + x264_adaptive_quant_frame, the two
    frame costs --b-adapt 1 spends on a new picture (x264's fast mode: the paths "..BP" and "..PP" from the last non-B picture through slicetype_path_cost),
    and when a mini-GOP closes macroblock_tree over it (clear / propagate / finish) — [x264-upstream] encoder/slicetype.c
    x264_slicetype_analyse, macroblock_tree; csrc/slicetype.hip, csrc/lookahead.hip.  Returns the time per picture and the launches made."""
    import numpy as np
    nslots = args.bframes + 5
    st, la = C.c_void_p(), C.c_void_p()
    lib.check(lib.x264gpu_slicetype_create(C.byref(st), W, H, S, nslots, args.bframes, 1, 7, 16, 1, 512, 1), "slicetype_create")
    lib.check(lib.x264gpu_lookahead_create(C.byref(la), W, H, S, 16, 7), "lookahead_create")
    nb = ((W + 15) // 16) * ((H + 15) // 16)
    aq = torch.empty((S, nb), dtype=torch.float32, device=dev)          # x264's f_qp_offset_aq / f_qp_offset: single floats
    offs = torch.empty((S, nb), dtype=torch.float32, device=dev)
    score = np.zeros(S, np.int32)
    calls = {"lowres": 0, "aq": 0, "frame_cost": 0, "propagate": 0, "finish": 0}
    ms = {k_: 0.0 for k_ in calls}
    sl = lambda i: i % nslots

    class timed:          # wall time of one group of launches (synchronised: the calls are tens of milliseconds each at this batch size)
        def __init__(self, key): self.key = key
        def __enter__(self): self.t = time.perf_counter()
        def __exit__(self, *a):
            torch.cuda.synchronize(dev)
            ms[self.key] += (time.perf_counter() - self.t) * 1e3
            calls[self.key] += 1

    def cost(p0, p1, b):
        with timed("frame_cost"):
            lib.check(lib.x264gpu_slicetype_frame_cost(st, sl(p0), sl(p1), sl(b), b - p0, p1 - b, score.ctypes.data, None), "slicetype_frame_cost")

    def prop(p0, p1, b, ref):
        with timed("propagate"):
            lib.check(lib.x264gpu_slicetype_propagate(st, sl(p0), sl(p1), sl(b), b - p0, p1 - b, ref, None), "slicetype_propagate")

    def path_cost(base, path):          # x264 slicetype_path_cost over pictures base + 1 .. base + len(path) (b-pyramid: through the middle B picture of a run)
        loc, cur, n = 1, 0, len(path)
        while loc <= n:
            nxt = loc
            while nxt <= n and path[nxt - 1] == "B":
                nxt += 1
            if nxt > n:
                break
            cost(base + cur, base + nxt, base + nxt)
            if nxt - cur > 2:
                mid = cur + (nxt - cur) // 2
                cost(base + cur, base + nxt, base + mid)
                for b_ in range(loc, mid):
                    cost(base + cur, base + mid, base + b_)
                for b_ in range(mid + 1, nxt):
                    cost(base + mid, base + nxt, base + b_)
            else:
                for b_ in range(loc, nxt):
                    cost(base + cur, base + nxt, base + b_)
            loc, cur = nxt + 1, nxt

    types = display_types(K, args.bframes, max(K, 1))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    p0 = 0
    for i in range(K):
        with timed("lowres"):
            lib.check(lib.x264gpu_slicetype_put_frame(st, sl(i), data[first + i].data_ptr(), None), "slicetype_put_frame")
        with timed("aq"):
            lib.check(lib.x264gpu_lookahead_aq_offsets(la, data[first + i].data_ptr(), 1.0397, aq.data_ptr(), None), "aq_offsets")
            lib.check(lib.x264gpu_slicetype_set_aq(st, sl(i), aq.data_ptr(), None), "set_aq")
        if i == 0:
            cost(0, 0, 0)
            continue
        # --b-adapt 1 (x264's fast mode: the path "..BP" against "..PP" from the last non-B picture, slicetype_path_cost on both) for picture j = i - 1, which needs
        # picture i: the frame costs those two paths ask for (a cost computed once stays with its picture: repeated requests return at once)
        j = i - 1
        if j >= 1:
            ln = max(q for q in range(j) if types[q] != "B")
            run = j - ln - 1
            if run < args.bframes:
                path_cost(ln, "B" * run + "PP")
                path_cost(ln, "B" * run + "BP")
        if types[i] != "B":
            cost(p0, i, i)
            n = i - p0 - 1
            lib.check(lib.x264gpu_slicetype_clear_propagate(st, sl(i), None), "clear_propagate")
            lib.check(lib.x264gpu_slicetype_clear_propagate(st, sl(p0), None), "clear_propagate")
            if n > 1:
                mid = p0 + (n + 1) // 2
                cost(p0, i, mid)
                lib.check(lib.x264gpu_slicetype_clear_propagate(st, sl(mid), None), "clear_propagate")
                for j in range(i - 1, p0, -1):
                    if j != mid:
                        a, b = (mid if j > mid else p0), (mid if j < mid else i)
                        cost(a, b, j); prop(a, b, j, 0)
                prop(p0, i, mid, 1)
            else:
                for j in range(i - 1, p0, -1):
                    cost(p0, i, j); prop(p0, i, j, 0)
            prop(p0, i, i, 1)
            cost(p0, p0, p0)
            with timed("finish"):
                lib.check(lib.x264gpu_slicetype_finish(st, sl(p0), 2.0, 0.0, offs.data_ptr(), None), "slicetype_finish")
            p0 = i
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    lib.x264gpu_slicetype_destroy(st)
    lib.x264gpu_lookahead_destroy(la)
    return dt, {k_: {"calls": calls[k_], "ms": round(ms[k_], 1)} for k_ in calls}


def e2e_probe(args):
    """ONE 1080p stream through the B1 boundary x264_encoder_encode() with host pictures, the way the driver calls it (codec.c:1693):
    copy-in, upload over PCIe, GPU hot path, record / level download, host CAVLC.  Constant quantiser (the parity configuration of
    SURVEY.md §8d), scene cut every 97 source frames.  threads 1: every call returns its picture; threads G: G closed GOPs in lock-step
    (delay (G - 1) x keyint + 1 pictures, byte-identical stream under CQP + fixed keyint)."""
    import numpy as np
    from x264vfw_amd import host_api as HL
    from x264vfw_amd.synth import synth_frames
    H = HL.H
    w, h = args.width, args.height
    planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]

    def run(n, threads, keyint, src, sliced=False, gop_slots=0, slices=0, inflight=None):
        os.environ.pop("X264GPU_INFLIGHT", None)
        if inflight is not None:
            os.environ["X264GPU_INFLIGHT"] = str(inflight)          # pictures of the session in flight (read at x264_encoder_open; 0: one picture a call)
        p = HL.Param()
        assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
        p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
        p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
        for k, v in (("qp", str(args.qp)), ("keyint", str(keyint)), ("threads", str(threads))):
            assert H.x264_param_parse(C.byref(p), k.encode(), v.encode()) == 0
        if sliced:
            assert H.x264_param_parse(C.byref(p), b"sliced-threads", None) == 0
        if slices:
            assert H.x264_param_parse(C.byref(p), b"slices", str(slices).encode()) == 0
        os.environ.pop("X264GPU_GOP_SLOTS", None)
        if gop_slots:
            os.environ["X264GPU_GOP_SLOTS"] = str(gop_slots)
        if gop_slots or (threads > 1 and not sliced):
            assert H.x264_param_parse(C.byref(p), b"min-keyint", str(keyint).encode()) == 0
            assert H.x264_param_parse(C.byref(p), b"scenecut", b"0") == 0
        p.b_annexb, p.b_repeat_headers = 1, 1
        h_ = H.x264_encoder_open_157(C.byref(p))
        assert h_
        pic, out = HL.Picture(), HL.Picture()
        assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
        nal, nn = C.POINTER(HL.Nal)(), C.c_int()
        got = total = 0
        first = None
        t0 = time.perf_counter()
        for i in range(n):
            f = src[i % len(src)]
            for pl, (sz, off) in enumerate(planes):
                C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
            pic.i_pts = i
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
            assert size >= 0
            got += size > 0
            total += size
            if size > 0 and first is None:
                first = i
        while H.x264_encoder_delayed_frames(h_):
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(out))
            assert size > 0
            got += 1
            total += size
        dt = time.perf_counter() - t0
        H.x264_encoder_close(h_)
        assert got == n
        delays.append(n if first is None else first)          # pictures handed in before the first one came back
        return round(n / dt, 2), round(total / n / 1e3, 1)

    def run_sessions(ns, n, src):
        """ns sessions on ns host threads through the cross-session batcher (X264GPU_BATCH): medium with a fixed picture structure (no scenecut, b-adapt 0,
        constant quantisers), every thread feeds host pictures and entropy-codes its own stream; -> frames/s over all sessions, kB per frame"""
        import threading
        os.environ["X264GPU_BATCH"] = str(ns)
        res, errs = [0] * ns, []
        marks = {"opened": [0.0] * ns, "coded": [0.0] * ns}          # per session: every session open (the batch's first launch can start); its last picture out

        def one(idx):
            try:
                p = HL.Param()
                assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
                p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
                p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
                for k, v in (("qp", str(args.qp)), ("keyint", "250"), ("scenecut", "0"), ("b-adapt", "0"), ("threads", "1")):
                    assert H.x264_param_parse(C.byref(p), k.encode(), v.encode()) == 0
                p.b_annexb, p.b_repeat_headers = 1, 1
                h_ = H.x264_encoder_open_157(C.byref(p))
                assert h_
                pic, out = HL.Picture(), HL.Picture()
                assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
                marks["opened"][idx] = time.perf_counter()
                nal, nn = C.POINTER(HL.Nal)(), C.c_int()
                got = total = 0
                for i in range(n):
                    f = src[(i + idx) % len(src)] if i else src[idx % len(src)]
                    for pl, (sz, off) in enumerate(planes):
                        C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
                    pic.i_pts = i
                    size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
                    assert size >= 0
                    got += size > 0
                    total += size
                while H.x264_encoder_delayed_frames(h_):
                    size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(out))
                    assert size > 0
                    got += 1
                    total += size
                marks["coded"][idx] = time.perf_counter()
                H.x264_encoder_close(h_)
                assert got == n
                res[idx] = total
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))
        ths = [threading.Thread(target=one, args=(i,)) for i in range(ns)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        os.environ.pop("X264GPU_BATCH", None)
        assert not errs, errs[:3]
        # the same frames over the span from the last session's open (no picture can be coded before: the batch waits for its members) to the last picture out
        steady = max(marks["coded"]) - max(marks["opened"])
        run_sessions.detail = {"setup_s": round(max(marks["opened"]) - t0, 2), "coding_s": round(steady, 2), "teardown_s": round(t0 + dt - max(marks["coded"]), 2),
                               "fps_coding_span": round(ns * n / steady, 2)}
        return round(ns * n / dt, 2), round(sum(res) / (ns * n) / 1e3, 1)

    def run_sessions_native(ns, n, src):
        """the same leg with a C++ caller (tools/multi_session.cpp, built on demand with g++): 2048 Python threads spend seconds a round handing the interpreter lock to each
        other between the calls — the harness's time, not the library's.  None when the driver cannot be built (the Python leg then stands)."""
        exe = os.path.join(ROOT, "tools", "_build", "multi_session")
        try:
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            srcf = os.path.join(ROOT, "tools", "multi_session.cpp")
            if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(srcf):
                subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", exe, srcf, "-L" + os.path.join(ROOT, "x264vfw_amd"), "-lx264gpu_host",
                                "-Wl,-rpath,$ORIGIN/../../x264vfw_amd"], check=True, capture_output=True, timeout=120)
            raw = "/dev/shm/bench_multi_session_%d.yuv" % os.getpid()
            with open(raw, "wb") as f:
                for fr in src:
                    f.write(np.ascontiguousarray(fr).tobytes())
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("X264GPU_LIB", "X264GPU_HOST_LIB")}
                out = subprocess.run([exe, raw, str(w), str(h), str(len(src)), str(ns), str(n), str(args.qp)], capture_output=True, timeout=1800, env=env)
            finally:
                os.remove(raw)
            if out.returncode != 0:
                return None
            return json.loads(out.stdout.decode().strip().splitlines()[-1])
        except Exception:
            return None

    n1 = args.e2e_frames
    delays = []
    src = synth_frames(w, h, max(n1, 16), seed=0x264, scene_len=97)
    # one session, threads 1: the library's default keeps up to four pictures of the session in flight (launch contexts over the shared DPB; the stream is the serial
    # one byte for byte: tests/test_gpu_host.py::test_pictures_in_flight_equal_serial); the same session one picture a call beside it
    nf1 = max(n1, 24)
    f1, kb1 = run(nf1, 1, 250, src)
    d_f1 = delays[-1]
    f1s, kb1s = run(n1, 1, 250, src, inflight=0)
    d_f1s = delays[-1]
    os.environ.pop("X264GPU_INFLIGHT", None)
    one = {"threads1_fps": f1, "threads1_frames": nf1, "threads1_kB_per_frame": kb1, "threads1_pictures_in_flight": 4, "threads1_delay_frames": d_f1,
           "threads1_serial_fps": f1s, "threads1_serial_frames": n1, "threads1_serial_delay_frames": d_f1s}
    fm, kbm = (None, None)
    if args.e2e_sessions > 1:
        nat = run_sessions_native(args.e2e_sessions, n1, src)
        if nat and "fps" in nat:
            fm, kbm = nat["fps"], nat["kB_per_frame"]
            run_sessions.detail = {"setup_s": nat["setup_s"], "coding_s": nat["coding_s"], "teardown_s": nat["teardown_s"], "fps_coding_span": nat["fps_coding_span"], "driver": nat["driver"]}
        else:
            fm, kbm = run_sessions(args.e2e_sessions, n1, src)
            run_sessions.detail["driver"] = "python threads (the C++ driver did not build or failed)"
    # ... and the same leg on lighter content (the rectangles' per-pixel noise smoothed, a little sensor noise on top: ~58 kB a picture instead of ~171): how far the API path
    # follows the device when the host — 16 CPUs on this pool's boxes, `host_cpu_quota` — has less to entropy-code
    light = None
    if args.e2e_sessions > 1 and fm is not None:
        try:
            from scipy.ndimage import uniform_filter
            rng = np.random.default_rng(1)
            lsrc = []
            for fr in src[:16]:
                fr = np.ascontiguousarray(fr).copy()
                y = uniform_filter(fr[:w * h].reshape(h, w).astype(np.float32), 7) + rng.normal(0, 1.5, (h, w))
                fr[:w * h] = np.clip(y + 0.5, 0, 255).astype(np.uint8).reshape(-1)
                lsrc.append(fr)
            nat = run_sessions_native(args.e2e_sessions, n1, lsrc)
            if nat and "fps" in nat:
                light = {"fps": nat["fps"], "kB_per_frame": nat["kB_per_frame"], "fps_coding_span": nat["fps_coding_span"], "setup_s": nat["setup_s"], "coding_s": nat["coding_s"], "teardown_s": nat["teardown_s"]}
        except Exception:
            light = None
    one["multi_session_light_content"] = light
    if args.e2e_legs != "all":
        return {"what": "x264_encoder_encode end to end, 1920x1080 (only the threads-1 and multi-session legs were asked for)", **one,
                "multi_session_fps": fm, "multi_session_sessions": args.e2e_sessions, "multi_session_frames_each": n1, "multi_session_kB_per_frame": kbm, "multi_session_spans": getattr(run_sessions, "detail", None), "host_cores": os.cpu_count(), "host_cpu_quota": cpu_quota()}
    # --threads G: 32 closed GOPs of the one stream in lock-step, medium's B pictures in every GOP (bframes 3, b-pyramid, weightb; b-adapt 0 and no scene cuts:
    # a fixed structure).  keyint 12 keeps the leg within the bench's minutes (a slot is one wavefront: ~0.5 pictures/s; keyint 250 x 32 slots would be 8000 pictures)
    G, K = 32, 12
    fg, kbg = run(G * K, G, K, src)
    d_thr = delays[-1]
    ns = (h + 15) // 16 // 4                        # x264 slice threads: at most one slice per four macroblock rows
    fs, kbs = run(max(n1, 24), ns, 250, src, sliced=True)
    d_sliced = delays[-1]
    fsg, kbsg = run(G * K * 2, ns, K, src, sliced=True, gop_slots=G)
    nr = (h + 15) // 16                             # x264 --slices N: down to one macroblock row per slice (filtered across the boundaries)
    fr, kbr = run(max(n1, 48), 1, 250, src, slices=nr)
    d_rows = delays[-1]
    frg, _ = run(G * K * 2, G, K, src, slices=nr)
    os.environ.pop("X264GPU_GOP_SLOTS", None)
    return {"what": "ONE 1920x1080 stream through x264_encoder_encode (host pictures in, Annex-B out: PCIe + host entropy coding included), CQP, preset medium as implemented (the threads-1 and slice legs with B pictures, b-adapt 1 and scene cuts: delays as measured; the --threads G legs with B pictures too: closed GOPs in lock-step on the DPB model, b-adapt 0, no scene cuts)",
            **one,
            "multi_session_fps": fm, "multi_session_sessions": args.e2e_sessions, "multi_session_frames_each": n1, "multi_session_kB_per_frame": kbm, "multi_session_spans": getattr(run_sessions, "detail", None),
            "multi_session_what": "that many x264_encoder_open sessions on as many host threads through the cross-session batcher (X264GPU_BATCH): one lock-step device launch per picture, host pictures in, every thread entropy-codes its own stream; session setup and teardown inside the timed span",
            "sliced_threads_fps": fs, "sliced_threads_slices": ns, "sliced_threads_delay_frames": d_sliced, "sliced_threads_kB_per_frame": kbs,
            "sliced_threads_gop_slots32_fps": fsg, "sliced_threads_gop_slots32_delay_frames": (G - 1) * K + 1,
            "slices_per_row_fps": fr, "slices_per_row_slices": nr, "slices_per_row_delay_frames": d_rows, "slices_per_row_kB_per_frame": kbr,
            "slices_per_row_threads32_fps": frg, "slices_per_row_threads32_delay_frames": (G - 1) * K + 1,
            "threads32_fps": fg, "threads32_frames": G * K, "threads32_keyint": K, "threads32_bframes": 3, "threads32_delay_frames": d_thr, "threads32_kB_per_frame": kbg,
            "host_cores": os.cpu_count(), "host_cpu_quota": cpu_quota()}


def main():
    args = parse_args()
    if args.cpu_worker >= 0:
        return cpu_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU over RCCL, before anything in this process touches the GPU
        port = str(29500 + os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # the measured libraries are the product's own: an environment override (the hook tools/ and the stub tests use) would let any other
    # library — the oracle behind the B3 ABI included — stand in for them, so the benchmark refuses to run under one
    for var in ("X264GPU_LIB", "X264GPU_HOST_LIB"):
        if os.environ.get(var):
            raise SystemExit(f"bench.py: {var} is set ({os.environ[var]}): the benchmark measures x264vfw_amd/libx264gpu.so and libx264gpu_host.so only; unset it")
    cpu = cpu_baseline(args) if (rank == 0 and world == 1 and args.cpu_frames > 0) else None      # before the GPU is touched (child processes)

    import torch
    from x264vfw_amd import shard
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from x264vfw_amd import lib
    from x264vfw_amd.lib import Config, MB_LEVELS

    W, H, S = args.width, args.height, args.streams
    K, Wu = args.steps, args.warmup
    gids = shard.stream_ids(rank, world, S)          # global stream ids of this rank (seeds only)
    tools = toolset(args)
    from x264vfw_amd import gop
    from x264vfw_amd import host_api as HL
    from x264vfw_amd.lib import Pic

    # ---- the pictures of every stream: a short first GOP (the warmup) and the GOP the timed region codes from its IDR picture on ----
    Wu = max(Wu, 1)
    types = display_types(Wu, args.bframes, Wu) + display_types(K, args.bframes, max(K, 1))
    order = gop.schedule(types, 1)                    # coding order: (display index, PIC_IDR / _I / _P / _BREF / _B)
    assert len(order) == Wu + K and order[Wu][1] == 0, "the timed region starts on an IDR picture"
    D = max(1, min(S, args.distinct))
    L = Wu + K

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    def timed_pass(content, keep_last=False):
        """Wu untimed + K timed pictures of S lock-step streams on synthetic `content`, inputs resident in HBM before the clock starts."""
        base = synth_batch(torch, D, L, W, H, shard.stream_seed(0x264, gids[0]), dev, smooth=content != "noise", scene_len=97, luma_noise=4 if content == "survey" else None)
        data = base if D == S else base.repeat(1, (S + D - 1) // D, 1)[:, :S].contiguous()
        del base
        cfg = Config(**dict(dict(width=W, height=H, streams=S, qp_i=max(0, args.qp - 3), qp_p=args.qp, me_range=16, deblock_alpha=0, deblock_beta=0,
                                 chroma_qp_offset=0, deadzone_inter=21, deadzone_intra=11, dct_decimate=1), **tools))
        h = C.c_void_p()
        lib.check(lib.x264gpu_encoder_create(C.byref(h), C.byref(cfg)), "encoder_create")
        n = lib.x264gpu_encoder_mb_count(h)
        mbs = torch.empty((S, n, 64), dtype=torch.uint8, device=dev)
        lvs = torch.empty((S, n, MB_LEVELS), dtype=torch.int16, device=dev)
        stream = torch.cuda.Stream(device=dev)
        # every stream has the same picture structure: the plans are made once, before anything is timed
        dpb = gop.HostDpb(HL, tools["refs"], args.bframes, 1, weightp=args.weightp)
        plans = []
        for k_, (disp, pt) in enumerate(order):
            pic, _ = dpb.plan(pt, disp, gop.follow_of(order, k_))
            pic.qp = qp_of(args, pt)
            plans.append((disp, (Pic * S)(*([pic] * S))))
            dpb.commit()

        def step(c):
            disp, arr = plans[c]
            lib.check(lib.x264gpu_encode_pictures(h, data[disp].data_ptr(), arr, mbs.data_ptr(), lvs.data_ptr(), stream.cuda_stream), "encode_pictures")

        for c in range(Wu):
            step(c)
        sync()
        lib.check(lib.x264gpu_encoder_profile_begin(h, K), "profile_begin")
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        sync()
        t0 = time.perf_counter()
        evs[0].record(stream)
        for c in range(K):
            step(Wu + c)
            evs[c + 1].record(stream)
        sync()
        dt = time.perf_counter() - t0
        step_ms = [evs[c].elapsed_time(evs[c + 1]) for c in range(K)]
        nst = lib.x264gpu_encoder_stage_count()
        names = [lib.x264gpu_encoder_stage_name(i).decode() for i in range(nst)]
        a, b = (C.c_double * nst)(), (C.c_int * nst)()
        lib.check(lib.x264gpu_encoder_profile_end(h, stream.cuda_stream, a, b), "profile_end")
        dt = shard.max_over_ranks(dt, dist, dev)
        res = {"dt": dt, "fps": shard.aggregate_fps(S, K, world, dt), "step_ms": step_ms, "names": names, "ms": list(a), "cnt": list(b), "nst": nst, "mbt": None}
        if rank == 0 and keep_last:
            import numpy as np
            res["mbt"] = np.bincount(mbs[:min(S, 64)].cpu().numpy()[:, :, 0].reshape(-1), minlength=11)
        lib.x264gpu_encoder_destroy(h)
        del mbs, lvs
        torch.cuda.empty_cache()
        return res, data

    # the same job on SURVEY.md 8(d)'s generator first (reported as value_survey_content beside the headline), then the headline's own content
    survey = None
    if args.survey_leg and args.content != "survey":
        survey, sdata = timed_pass("survey")
        del sdata
        torch.cuda.empty_cache()
    main_res, data = timed_pass(args.content, keep_last=True)
    dt, fps, step_ms, names, ms, cnt, nst = (main_res[k_] for k_ in ("dt", "fps", "step_ms", "names", "ms", "cnt", "nst"))
    # ---- roofline of the dominant kernel: the macroblock loop (k_mb_slice) ----
    Sb = 1.5 * W * H
    # algorithmic HBM bytes per frame and stage (DESIGN.md "kernels"): planes each stage must read / write once.  The macroblock loop reads the
    # source and one reference per list it predicts from and writes the reconstruction it keeps (I 2 S, P 3 S, B reference 4 S, b 3 S)
    tname = {0: "I", 1: "I", 2: "P", 3: "Bref", 4: "b"}
    kind = {0: "I", 1: "I", 2: "P", 3: "B", 4: "B"}
    tcount = {"I": 0, "P": 0, "Bref": 0, "b": 0}
    tms = {"I": 0.0, "P": 0.0, "Bref": 0.0, "b": 0.0}
    for c in range(K):
        tcount[tname[order[Wu + c][1]]] += 1
        tms[tname[order[Wu + c][1]]] += step_ms[c]
    alg_mb = (2 * tcount["I"] + 3 * tcount["P"] + 4 * tcount["Bref"] + 3 * tcount["b"]) * Sb / K
    alg = {"ingest": 2 * Sb, "macroblocks": alg_mb, "unused": 0, "settle_qp": 0, "deblock": 2 * Sb, "hpel_filter": 4.5 * W * H}
    dom = max(range(nst), key=lambda i: ms[i])
    avg_ms = ms[dom] / max(cnt[dom], 1)
    achieved = alg[names[dom]] * S / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, valu, pmc_src = pmc_evidence("k_mb_slice" if names[dom] == "macroblocks" else names[dom], avg_ms, S, args.content)
    if args.rd != "cabac" or args.no_trellis:
        traffic, valu, pmc_src = None, None, None                            # the committed counters are the headline kernel's (RD with CABAC sizes)
    per_type = {t: round(tms[t] / tcount[t], 2) for t in tms if tcount[t]}
    # the same per-type step times in the proportions of a --keyint GOP (1 I, then runs of bframes B pictures closed by P pictures)
    blended = None
    if all(t in per_type for t in (("I", "P", "Bref", "b") if args.bframes > 1 else ("I", "P", "b") if args.bframes else ("I", "P"))):
        gt = display_types(args.keyint, args.bframes, args.keyint)
        go = gop.schedule(gt, 1)
        tot = sum(per_type[tname[pt]] for _, pt in go)
        blended = round(S * world * len(go) / (tot * 1e-3), 2)
    roof = {"bound": "hbm", "binding_resource": "instruction issue of a raster-serial macroblock loop (neither HBM nor MFMA binds this integer path; the HBM fraction is reported because the contract asks for one of the two)", "kernel": "k_mb_slice (macroblock loop)" if names[dom] == "macroblocks" else names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "valu": valu,
            # traffic and valu.insts_* are the committed rocprofv3 --pmc passes' figures (counters cannot be read inside an un-profiled run), scaled to this
            # run's streams; valu_issue_frac = those wave-level VALU instructions / (1024 SIMDs x 2.4 GHz / 4 cycles an instruction) / THIS run's kernel time
            "traffic_from": pmc_src if traffic is not None else None, "valu_from": pmc_src if valu else None,
            "valu_issue_frac": valu["issue_util"] if valu else None,
            "note": "two wavefronts per SIMD (256 VGPRs, 20 KB of LDS each); each waits 0.42 - 0.46 of its cycles (LDS / L2 round trips of a raster-serial loop), VALU issue 0.42 - 0.48 of the SIMDs' peak: profiles/r06_pmc_per_launch.json, DESIGN.md 5",
            "avg_launch_ms": round(avg_ms, 4), "step_ms_by_picture_type": per_type, "frames_per_s_in_keyint_proportions": blended,
            "stage_ms_per_step": {names[i]: round(ms[i] / K, 4) for i in range(nst) if names[i] != "unused"}}
    mix = " + ".join(f"{tcount[t]} {t}" for t in ("I", "P", "Bref", "b") if tcount[t])
    out = {"metric": "1080p yuv420p frames/sec at preset=medium as the device runs it (see config.toolset_gaps), hot path on MI355X",
           "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wu,
           "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"{W}x{H} yuv420p, {S} closed-GOP streams/GPU x {K} coded pictures from an IDR on ({mix}; coding order of {types[Wu:]}), distinct frames, CQP {max(0, args.qp - 3)}/{args.qp}/{args.qp + 2}, "
                                  f"preset {args.preset}, content '{args.content}'",
                      "toolset": tools, "weightp": args.weightp, "toolset_gaps": TOOLSET_GAPS_NORD if args.rd == "off" or args.preset == "ultrafast" else (TOOLSET_GAPS if args.bframes else TOOLSET_GAPS_NOB) + (" [--rd cavlc: medium --no-cabac]" if args.rd == "cavlc" else ""),
                      "streams_per_gpu": S, "distinct_sequences": D, "frames_per_sequence": L, "content": args.content, "frames_per_step": S * world, "keyint_for_proportions": args.keyint,
                      "mb_per_frame": ((W + 15) // 16) * ((H + 15) // 16),
                      # launches of the three macroblock-loop instantiations before / inside the timed window (tools/profile_summarise.py folds the counters of the timed ones)
                      "launch_plan": {t: {"warmup": sum(1 for _, pt in order[:Wu] if kind[pt] == t), "timed": sum(1 for _, pt in order[Wu:] if kind[pt] == t)} for t in ("I", "P", "B")}},
           "roofline": roof}
    if survey is not None:
        out["value_survey_content"] = round(survey["fps"], 2)
        out["ms_per_step_survey_content"] = round(survey["dt"] / K * 1e3, 4)
    if rank == 0:
        mbt = main_res["mbt"]
        tot = float(mbt.sum())
        out["config"]["mb_type_share_last_step"] = {k: round(int(v) / tot, 4) for k, v in (("I4x4", mbt[0]), ("I8x8", mbt[1]), ("I16x16", mbt[2]), ("P16x16/16x8/8x16", mbt[4]),
                                                                                          ("P8x8", mbt[5]), ("P_Skip", mbt[6]), ("B_Direct", mbt[7]), ("B_Skip", mbt[8]), ("B L0/L1/Bi", mbt[9]), ("B_8x8", mbt[10]))}
        out["csp_ingest"] = csp_probe(torch, lib, dev, W, H)
        out["config"]["product_libraries"] = loaded_product_libraries()
    if args.lookahead and args.bframes:
        # ---- the lookahead's device work for the same streams and pictures, beside the headline ----
        la_dt, la_calls = lookahead_probe(torch, lib, dev, args, data, Wu, K, S, W, H)
        la_dt = shard.max_over_ranks(la_dt, dist, dev)
        out["lookahead"] = {"what": "the device work of x264's lookahead for the same streams and pictures (lowres planes, AQ mode 1 offsets, --b-adapt 1's frame costs, path costs + "
                                    "macroblock_tree per mini-GOP), run after the timed window; its decisions are discarded (the lock-step batch codes a fixed picture structure)",
                            "ms_per_picture": round(la_dt / K * 1e3, 3), "launch_groups": la_calls,
                            "value_with_lookahead": round(S * world * K / (dt + la_dt), 2), "share_of_step": round(la_dt / (dt + la_dt), 4)}
    del data
    torch.cuda.empty_cache()
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and args.e2e_frames > 0:
            out["e2e"] = e2e_probe(args)
        if world == 1:
            # a decoder nobody here wrote, if the box has one (tools/decoder_probe.py): the only independent check the entropy tables and the
            # normative reconstruction can get while no libx264 exists to pin the oracle
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import decoder_probe
            out["decoder_probe"] = decoder_probe.probe()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
