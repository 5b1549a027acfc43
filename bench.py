#!/usr/bin/env python3
"""bench.py — 1080p yuv420p frames/sec of the MI355X encode hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N>1 is launched by torch.distributed.run)

A *step* is one lock-step pass of the hot path over one batch of synthetic input: every one of the
`--streams` independent closed-GOP streams on this GPU advances by one frame (ingest -> ME/analysis ->
DCT/quant/recon -> intra wavefront -> deblock wavefront -> half-pel planes), through the C ABI
x264gpu_encode_frames() of libx264gpu.so.  Inputs are resident in HBM before the timed region.  The
timed K steps start on an IDR boundary and contain the I/P mix of closed GOPs with --keyint (60).
Frames of different streams/GOPs are independent (config 5 of BASELINE.json), so N GPUs shard streams
one set per GPU with no collective in the data path ("scaling": "weak").

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed inside the timed region)
and `cpu_baseline` (the oracle restatement on one host core, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_batch(torch, streams, frames, w, h, seed, device):
    """[frames, streams, w*h*3/2] uint8 I420 on the device: gradient + 3 moving textured rectangles +
    per-pixel noise, generated with torch ops (plumbing only)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy = torch.arange(h, device=device).view(h, 1)
    xx = torch.arange(w, device=device).view(1, w)
    out = torch.empty((frames, streams, w * h * 3 // 2), dtype=torch.uint8, device=device)
    vel = ((3, 1), (-2, 2), (5, -3))
    for s in range(streams):
        base = 90 + 7 * (s % 8)
        grad = (base + xx * 60 // w + yy * 50 // h).to(torch.int16)
        tex = [torch.randint(0, 256, (h // 3, w // 3), generator=g, device=device, dtype=torch.int16) for _ in vel]
        pos = [(int(torch.randint(0, w, (1,), generator=g, device=device)), int(torch.randint(0, h, (1,), generator=g, device=device))) for _ in vel]
        for n in range(frames):
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
            y = (y + torch.randint(-4, 5, y.shape, generator=g, device=device, dtype=torch.int16)).clamp_(16, 235)
            u = (u + torch.randint(-2, 3, u.shape, generator=g, device=device, dtype=torch.int16)).clamp_(16, 240)
            v = (v + torch.randint(-2, 3, v.shape, generator=g, device=device, dtype=torch.int16)).clamp_(16, 240)
            out[n, s] = torch.cat([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).to(torch.uint8)
    return out


def pmc_evidence(stage, avg_launch_ms, frames_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
    command (tools/profile_round.sh -> profiles/r01_pmc_per_launch.json; FETCH_SIZE and WRITE_SIZE collected in
    separate passes, KB -> bytes, read side doubled as MI355X_MICROARCH.md prescribes for gfx950), plus the
    VALU issue utilisation the same passes imply (the path is integer-VALU bound, not HBM bound).  PMC counters
    cannot be read from inside an un-profiled run, so this is the profile's figure, valid only for the default
    workload (the profile records its frames per launch); anything else reports null."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_per_launch.json")
    if not os.path.exists(path):
        return None, None
    tab = json.load(open(path))
    if tab.get("_workload", {}).get("frames_per_launch") != frames_per_launch:
        return None, None
    key = [k for k in tab if ("k_" + stage) in k]
    if not key or "hbm_bytes_per_launch_corrected" not in tab[key[0]]:
        return None, None
    t = tab[key[0]]
    valu = None
    if "SQ_INSTS_VALU" in t and avg_launch_ms > 0:
        simds, clk = 256 * 4, 2.4e9                      # wave64 VALU op = 4 issue cycles on a SIMD16
        valu = {"insts_per_launch": int(t["SQ_INSTS_VALU"]), "issue_util": round(t["SQ_INSTS_VALU"] * 4 / (simds * clk * avg_launch_ms * 1e-3), 3),
                "source": "profiles/r01_pmc_per_launch.json"}
    return int(t["hbm_bytes_per_launch_corrected"]), valu


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


def lookahead_probe(torch, lib, dev, W, H, frames, iters=6):
    """next-row f2 evidence (not part of `value`): lookahead frame cost (half-resolution planes + 8x8 search + intra SATD,
    x264_slicetype_frame_cost) of `frames` 1080p pictures per launch pair, on a moving synthetic sequence."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import synth_frames
    seq = synth_frames(W, H, 3, seed=0x264, scene_len=10 ** 9)
    pics = [torch.from_numpy(np.stack([f] * frames)).to(dev) for f in seq]
    la = C.c_void_p()
    lib.check(lib.x264gpu_lookahead_create(C.byref(la), W, H, frames, 16, 7), "lookahead_create")
    d_out = torch.zeros((frames, 4), dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream(dev).cuda_stream
    for i in range(2):
        lib.check(lib.x264gpu_lookahead_frame_cost(la, pics[i].data_ptr(), int(i == 0), d_out.data_ptr(), None, cur), "lookahead")
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        lib.check(lib.x264gpu_lookahead_frame_cost(la, pics[(i + 2) % 3].data_ptr(), 0, d_out.data_ptr(), None, cur), "lookahead")
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    o = d_out[0].cpu().numpy()
    lib.x264gpu_lookahead_destroy(la)
    return {"kernels": "k_la_lowres + k_la_cost", "frames_per_launch": frames, "avg_launch_ms": round(ms, 4),
            "frames_per_s": round(frames / (ms * 1e-3), 1), "intra_cost": int(o[0]), "p_cost": int(o[1])}


def cpu_baseline(w, h, nframes, keyint, tools):
    """oracle/ (CPU restatement, one core) on a bounded sample of the same workload — the checker timed as
    a baseline, never the product."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from synth import synth_frames
    frames = synth_frames(w, h, nframes, seed=0x264, scene_len=10 ** 9)
    enc = O.OracleEncoder(O.default_config(w, h, **tools))
    t0 = time.perf_counter()
    for i, f in enumerate(frames):
        enc.encode(np.ascontiguousarray(f), 2 if i % keyint == 0 else 0)
    dt = time.perf_counter() - t0
    enc.close()
    return nframes / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--streams", type=int, default=512, help="independent closed-GOP streams per GPU (lock-step batch)")
    ap.add_argument("--groups", type=int, default=2, help="stream groups on separate HIP streams (stage overlap)")
    ap.add_argument("--keyint", type=int, default=60)
    ap.add_argument("--qp", type=int, default=23)
    ap.add_argument("--refs", type=int, default=3, help="reference frames (medium: 3)")
    ap.add_argument("--preset", default="medium", choices=["medium", "ultrafast", "slow"],
                    help="toolset of the other BASELINE.json configs (default medium = the headline); slow runs hex instead of umh")
    ap.add_argument("--aq", action="store_true", help="variance AQ on (per-macroblock quantisers, the CRF / ABR path); the headline metric is CQP and leaves it off, as x264 does")
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    from x264vfw_amd import shard
    rank, local_rank, world = shard.env_rank_world()
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
    per = shard.split_groups(S, args.groups)
    G = len(per)
    gids = shard.stream_ids(rank, world, S)          # global stream ids of this rank (seeds only)
    qp_i, qp_p = max(0, args.qp - 3), args.qp      # CQP ladder: ipratio 1.4 ~ -3 (x264 CQP convention)

    # toolsets (config.c:1460-1498 preset deltas restricted to what the pipeline implements)
    tools = {"medium": dict(refs=args.refs, subme=7, deblock=1, partitions=7, dct8x8=1, me_method=1, chroma_me=1, mixed_refs=1),
             "ultrafast": dict(refs=1, subme=0, deblock=0, partitions=0x100, dct8x8=0, me_method=0, chroma_me=0, mixed_refs=0),
             "slow": dict(refs=4, subme=9, deblock=1, partitions=7, dct8x8=1, me_method=2, chroma_me=1, mixed_refs=1)}[args.preset]
    if args.aq:
        tools = dict(tools, aq_mode=1, aq_strength_q8=266)
    # ---- inputs resident in HBM: warmup frames + K timed frames per stream ----
    nfr = Wu + K
    data = [synth_batch(torch, per[g], nfr, W, H, shard.stream_seed(0x264, gids[sum(per[:g])]), dev) for g in range(G)]
    encs, hs, mbs, lvs, streams = [], [], [], [], []
    for g in range(G):
        cfg = Config(width=W, height=H, streams=per[g], qp_i=qp_i, qp_p=qp_p, me_range=16, deblock_alpha=0, deblock_beta=0,
                     chroma_qp_offset=0, deadzone_inter=21, deadzone_intra=11, dct_decimate=1, **tools)
        h = C.c_void_p()
        lib.check(lib.x264gpu_encoder_create(C.byref(h), C.byref(cfg)), "encoder_create")
        n = lib.x264gpu_encoder_mb_count(h)
        hs.append(h)
        mbs.append(torch.empty((per[g], n, 64), dtype=torch.uint8, device=dev))
        lvs.append(torch.empty((per[g], n, MB_LEVELS), dtype=torch.int16, device=dev))
        streams.append(torch.cuda.Stream(device=dev))

    def step(i, first_of_gop):
        st = 2 if first_of_gop else 0
        for g in range(G):
            lib.check(lib.x264gpu_encode_frames(hs[g], data[g][i].data_ptr(), st, mbs[g].data_ptr(), lvs[g].data_ptr(),
                                                streams[g].cuda_stream), "encode_frames")

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()

    for i in range(Wu):
        step(i, i == 0)
    sync()
    for g in range(G):
        lib.check(lib.x264gpu_encoder_profile_begin(hs[g], K), "profile_begin")
    sync()
    t0 = time.perf_counter()
    for i in range(K):
        step(Wu + i, i % args.keyint == 0)
    sync()
    dt = time.perf_counter() - t0

    nst = lib.x264gpu_encoder_stage_count()
    names = [lib.x264gpu_encoder_stage_name(i).decode() for i in range(nst)]
    ms = [0.0] * nst
    cnt = [0] * nst
    for g in range(G):
        a, b = (C.c_double * nst)(), (C.c_int * nst)()
        lib.check(lib.x264gpu_encoder_profile_end(hs[g], streams[g].cuda_stream, a, b), "profile_end")
        for i in range(nst):
            ms[i] += a[i]
            cnt[i] += b[i]
    dt = shard.max_over_ranks(dt, dist, dev)
    fps = shard.aggregate_fps(S, K, world, dt)
    # ---- roofline of the dominant kernel (largest summed device time on this rank) ----
    Sb = 1.5 * W * H
    # algorithmic HBM bytes per frame and stage (DESIGN.md "kernels"): planes each stage must read/write once
    alg = {"ingest": 2 * Sb, "analyse_p": 2 * W * H, "encode_inter": 3 * Sb, "intra": 2 * Sb, "deblock": 2 * Sb,
           "hpel_filter": 4 * W * H + 0.5 * W * H}
    dom = max(range(nst), key=lambda i: ms[i])
    frames_per_launch = S / G            # every launch of a group covers its streams (one frame each)
    avg_ms = ms[dom] / max(cnt[dom], 1)
    achieved = alg[names[dom]] * frames_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, valu = pmc_evidence(names[dom], avg_ms, S // G)
    roof = {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "valu": valu,
            "avg_launch_ms": round(avg_ms, 4),
            "stage_ms_per_step": {names[i]: round(ms[i] / K / G, 4) for i in range(nst)}}
    out = {"metric": "1080p yuv420p frames/sec at preset=medium (I/P subset, CQP), hot path on MI355X",
           "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wu,
           "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"{W}x{H} yuv420p, {S} closed-GOP streams/GPU x {K} frames, keyint {args.keyint}, "
                                  f"CQP {qp_i}/{qp_p}, preset {args.preset} toolset: {tools}",
                      "streams_per_gpu": S, "stream_groups": G, "frames_per_step": S * world},
           "roofline": roof}
    if rank == 0:
        import numpy as np
        types = np.bincount(mbs[0].cpu().numpy()[:, :, 0].reshape(-1), minlength=7)
        out["config"]["mb_types_last_step"] = {"I4x4": int(types[0]), "I8x8": int(types[1]), "I16x16": int(types[2]), "P16x16/16x8/8x16": int(types[4]), "P8x8": int(types[5])}
        out["csp_ingest"] = csp_probe(torch, lib, dev, W, H)
        out["lookahead"] = lookahead_probe(torch, lib, dev, W, H, min(S // G, 256))
        if args.cpu_frames > 0:
            cfps, cdt = cpu_baseline(W, H, args.cpu_frames, args.keyint, tools)
            out["cpu_baseline"] = {"value": round(cfps, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{args.cpu_frames} frames {W}x{H} (1 I + {args.cpu_frames - 1} P), oracle/encoder.c single thread, {cdt:.1f} s"}
        print(json.dumps(out), flush=True)
    for h in hs:
        lib.x264gpu_encoder_destroy(h)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
