"""CPU tests of the multi-device paths of the host shell, driven through x264_encoder_encode() with the stand-in device
library of tests/stub/ (the oracle behind the B3 ABI, several "devices"):
  * ONE stream, --threads G: the G closed-GOP slots are dealt to the visible devices, one host thread per device — the stream must be
    byte-identical to the serial (threads 1) encode, and every device must have done work;
  * world-size-2 gloo: one stream per rank (BASELINE.json config 5), each rank encodes through its own device; rank 0 gathers the
    results and checks them against a single-process encode of the same streams + the MAX-over-ranks bookkeeping of bench.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
RUN = os.path.join(HERE, "stub", "run_host.py")
subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "stub")])    # once, here: the child processes only load what it built


def run_host(w, h, n, seed, opts, devices, inflight=None):
    env = dict(os.environ, X264GPU_STUB_DEVICES=str(devices))
    env.pop("X264GPU_DEVICES", None)
    env.pop("X264GPU_INFLIGHT", None)
    if inflight is not None: env["X264GPU_INFLIGHT"] = str(inflight)
    args = [sys.executable, RUN, str(w), str(h), str(n), str(seed)] + [k if v is None else f"{k}={v}" for k, v in opts.items()]
    out = subprocess.run(args, env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


@pytest.mark.parametrize("devices,threads", [(2, 4), (3, 5), (2, 2), (4, 3)])
def test_gop_slots_dealt_to_devices_equal_the_serial_stream(devices, threads):
    w, h, n = 96, 80, 23
    opts = {"qp": 27, "keyint": 4, "min-keyint": 4, "scenecut": 0, "ref": 2, "bframes": 0, "weightp": 0}      # (I / P pictures; B pictures: the test below)
    serial = run_host(w, h, n, 7, dict(opts, threads=1), 1)
    par = run_host(w, h, n, 7, dict(opts, threads=threads), devices)
    assert par["frames"] == serial["frames"] == n
    assert par["sha"] == serial["sha"], "GOP-parallel stream over several devices differs from the serial one"
    used = [c for c in par["calls"] if c > 0]
    assert len(used) == min(devices, threads), par["calls"]            # every device that owns a slot coded its positions
    assert serial["calls"][0] == n and sum(par["calls"]) < n * min(devices, threads)


@pytest.mark.parametrize("devices,threads,n,keyint,extra", [
    (2, 3, 23, 8, {}),                                            # two full GOPs + a last, shorter one (coded alone at the flush)
    (3, 4, 27, 9, {"weightp": 2, "ref": 3}),                      # --weightp 2's duplicates on the P pictures; a last GOP of 0 pictures left over: 27 = 3 x 9
    (2, 2, 21, 8, {"b-pyramid": "none", "bframes": 2}),           # the shorter GOP sits in the batch's LAST slot: the lock-step rounds had begun to code it
    (4, 4, 14, 6, {"direct": "temporal", "bframes": 3}),          # two batches: 4 slots x 6 pictures would be 24 — one partly gathered batch only
    (1, 2, 5, 12, {"bframes": 3}),                                # fewer pictures than one GOP
    (2, 3, 41, 7, {"bframes": 1, "slices": 2}),                   # several batches (7 x 3 = 21 pictures each), slices
])
def test_gop_slots_with_b_pictures_equal_the_serial_stream(devices, threads, n, keyint, extra):
    """north_star's own split at the real preset: closed GOPs of ONE stream, one per slot / device, WITH medium's B pictures (bframes 3, b-pyramid, weightb):
    every slot runs the DPB model's plan in lock-step; bytes, picture types, pts / dts and nal_ref_idc equal the threads-1 session's
    (codec.c:933 forwards --threads, :848 closed GOPs)."""
    w, h = 96, 80
    opts = dict({"qp": 27, "keyint": keyint, "min-keyint": keyint, "scenecut": 0, "ref": 2, "bframes": 3, "b-adapt": 0, "weightp": 0}, **extra)
    serial = run_host(w, h, n, 11, dict(opts, threads=1), 1)
    par = run_host(w, h, n, 11, dict(opts, threads=threads), devices)
    assert par["frames"] == serial["frames"] == n
    assert sorted(serial["pts"]) == list(range(n)) and serial["pts"] != sorted(serial["pts"]), "the serial session codes B pictures (output order differs from display order)"
    assert par["sha"] == serial["sha"], "GOP slots with B pictures: the stream differs from the serial one"
    assert par["meta"] == serial["meta"], (par["pts"], serial["pts"], par["dts"], serial["dts"])
    assert sum(1 for c in par["calls"] if c > 0) == min(devices, threads, -(-n // keyint))


@pytest.mark.parametrize("n,opts", [
    (23, {"qp": 27, "keyint": 12, "scenecut": 0, "ref": 2, "bframes": 3, "b-adapt": 0, "weightp": 2}),                 # medium's mini-GOP: P, Bref, b, b in flight together
    (23, {"crf": 23, "rc-lookahead": 10, "bframes": 3, "b-adapt": 1}),                                                  # the tree's offsets and CRF quantisers, decided ahead of the coding
    (19, {"qp": 25, "bframes": 2, "b-pyramid": "none", "ref": 3}),                                                     # no B reference: both b pictures behind the P picture only
    (23, {"crf": 22, "b-adapt": 2, "weightp": 2, "rc-lookahead": 6, "bframes": 3, "direct": "temporal"}),               # temporal direct reads the co-located picture's side data: an event away
    (3, {"qp": 27, "bframes": 3}),                                                                                     # fewer pictures than contexts: everything at the flush
    (17, {"qp": 26, "bframes": 1, "ref": 1, "keyint": 5, "min-keyint": 5, "scenecut": 0}),                              # IDR pictures empty the DPB while pictures are in flight
    (21, {"crf": 24, "bframes": 3, "ref": 4, "aud": None, "slices": 2, "aq-mode": 2, "no-mbtree": None}),
])
@pytest.mark.parametrize("inflight", [2, 4])
def test_pictures_in_flight_equal_the_serial_stream(n, opts, inflight):
    """Several pictures of ONE session in flight (x264_t::Inflight: launch contexts over the shared DPB, a picture behind the events of its references): bytes, picture types,
    pts / dts and nal_ref_idc are those of the session that codes one picture a call (X264GPU_INFLIGHT=0) — what the reference's frame threads promise
    ([x264-upstream] encoder/encoder.c x264_encoder_encode: i_thread_frames pictures in flight, the stream does not depend on the thread count beyond the lookahead's)."""
    w, h = 96, 80
    serial = run_host(w, h, n, 5, dict(opts, threads=1), 1, inflight=0)
    fl = run_host(w, h, n, 5, dict(opts, threads=1), 1, inflight=inflight)
    assert serial["views"] == 0 and fl["views"] == inflight - 1, (serial["views"], fl["views"])      # the launch contexts exist: the path under test ran
    assert fl["frames"] == serial["frames"] == n
    assert fl["sha"] == serial["sha"], "pictures in flight: the stream differs from the serial one"
    assert fl["meta"] == serial["meta"], (fl["pts"], serial["pts"], fl["dts"], serial["dts"])


def test_device_cap_env_and_crf_across_devices():
    """X264GPU_DEVICES caps the devices used; CRF quantisers (decided on arrival, per GOP slot) survive the dealing"""
    w, h, n = 96, 80, 16
    opts = {"crf": 26, "keyint": 4, "min-keyint": 4, "scenecut": 0, "no-mbtree": None, "threads": 4}
    one = run_host(w, h, n, 3, opts, 1)
    two = run_host(w, h, n, 3, opts, 2)
    assert one["sha"] == two["sha"] and sum(1 for c in two["calls"] if c) == 2
    env = dict(os.environ, X264GPU_STUB_DEVICES="3", X264GPU_DEVICES="1")
    out = subprocess.run([sys.executable, RUN, str(w), str(h), str(n), "3"] + [k if v is None else f"{k}={v}" for k, v in opts.items()],
                         env=env, capture_output=True, timeout=600)
    capped = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert capped["sha"] == one["sha"] and sum(1 for c in capped["calls"] if c) == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      X264GPU_STUB_DEVICES=str(world), X264_HOST_STUB="1")
    import time
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(HERE, "stub"))
    sys.path.insert(0, os.path.dirname(HERE))
    import run_host                                             # loads the stub-backed host library in this fresh process
    from x264vfw_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard.stream_ids(rank, world, 1)
    dist.barrier()
    t0 = time.perf_counter()
    res, _ = run_host.encode(96, 80, 9, shard.stream_seed(0x264, ids[0]), {"qp": 28, "keyint": 3, "min-keyint": 3, "scenecut": 0, "threads": 1})
    dt = time.perf_counter() - t0
    mx = shard.max_over_ranks(dt, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, ids, res["sha"], res["frames"], dt))
    if rank == 0:
        q.put((gathered, mx, shard.aggregate_fps(1, 9, world, mx)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_stream_each_with_real_encodes():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    gathered, mx, fps = q.get(timeout=300)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    gathered = sorted(gathered)
    assert [g[1] for g in gathered] == [[0], [1]]
    # each rank's stream equals the same stream encoded alone in a single process
    for rank, ids, sha, frames, dt in gathered:
        ref = run_host(96, 80, 9, 0x264 + ids[0], {"qp": 28, "keyint": 3, "min-keyint": 3, "scenecut": 0, "threads": 1}, 1)
        assert (sha, frames) == (ref["sha"], 9)
    assert gathered[0][2] != gathered[1][2]                                     # different seeds, different streams
    assert abs(mx - max(g[4] for g in gathered)) < 1e-9 and abs(fps - 1 * 9 * 2 / mx) < 1e-6
