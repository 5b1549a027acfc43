"""File muxers of the host shell (`--output x.mkv | x.flv | x.h264`; host/muxers.cpp) without a GPU: a slice
stream made by the host entropy coder from oracle records goes through each muxer and is read back by independent container
readers (tests/container_parse.py); the elementary stream recovered from the container decodes to the oracle's reconstruction."""
import ctypes as C

import numpy as np
import pytest

import container_parse as CP
import host_lib as HL
import oracle_lib as O
from synth import synth_frames

H = HL.H
H.x264host_mux_open.restype = C.c_void_p
H.x264host_mux_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
H.x264host_mux_set_param.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int]
H.x264host_mux_write_headers.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
H.x264host_mux_write_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int]
H.x264host_mux_close.argtypes = [C.c_void_p, C.c_int64, C.c_int64]


def annexb_nals(stream):
    parts = stream.split(b"\0\0\1")
    return [p[:-1] if p.endswith(b"\0") and i + 1 < len(parts) else p for i, p in enumerate(parts)][1:]


def make_stream(w, h, nfr, keyint):
    """-> (sps, pps, [(nal bytes, is_idr)], [reconstructions]) from the oracle pipeline + host slice writer"""
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    enc = O.OracleEncoder(O.default_config(w, h, refs=2, partitions=3))
    hdr = annexb_nals(HL.write_headers(w, h, 30, 8, 23, 0, 1, 50, 2, 0))
    frames, recs = [], []
    for i, f in enumerate(synth_frames(w, h, nfr, seed=6)):
        idr = i % keyint == 0
        mbs, lv = enc.encode(f, 2 if idr else 0)
        nal, _ = HL.write_slice(mbw, mbh, 2 if idr else 0, 20 if idr else 23, 23, i % keyint, 8, int(idr), i // keyint, 0, mbs, lv,
                                num_ref=max(1, min(i % keyint, 2)), num_ref_default=2)
        frames.append((annexb_nals(nal)[0], idr)); recs.append(enc.recon())
    return hdr[0], hdr[1], frames, recs


def pref(n):
    return len(n).to_bytes(4, "big") + n


def mux(path, kind, w, h, sps, pps, frames, fps=(25, 1), sar=(1, 1)):
    annexb = C.c_int()
    m = H.x264host_mux_open(str(path).encode(), kind, C.byref(annexb))
    assert m
    assert H.x264host_mux_set_param(m, w, h, fps[0], fps[1], fps[1], fps[0], sar[0], sar[1], 0) == 0
    sei = b"\x06\x05\x04test\x80"
    if annexb.value:
        hs, hp, he = b"\0\0\0\1" + sps, b"\0\0\0\1" + pps, b"\0\0\1" + sei
    else:
        hs, hp, he = pref(sps), pref(pps), pref(sei)
    assert H.x264host_mux_write_headers(m, hs, len(hs), hp, len(hp), he, len(he)) > 0
    for i, (nal, idr) in enumerate(frames):
        payload = (b"\0\0\0\1" + nal) if annexb.value else pref(nal)
        assert H.x264host_mux_write_frame(m, payload, len(payload), i, i, int(idr), 1 if idr else 3) == len(payload)
    assert H.x264host_mux_close(m, len(frames) - 1, len(frames) - 2) == 0
    return annexb.value, sei


@pytest.fixture(scope="module")
def stream():
    return (208, 120) + make_stream(208, 120, 7, 4)


def decode_check(w, h, sps, pps, nals, recs):
    es = b"\0\0\0\1" + sps + b"\0\0\0\1" + pps + b"".join(b"\0\0\0\1" + n for n in nals)
    dec = O.h264_decode(es, len(recs), w, h)
    for i, r in enumerate(recs):
        np.testing.assert_array_equal(dec[i], r, err_msg=f"picture {i} recovered from the container")


def test_mkv(stream, tmp_path):
    w, h, sps, pps, frames, recs = stream
    annexb, sei = mux(tmp_path / "a.mkv", b"auto", w, h, sps, pps, frames, fps=(30000, 1001), sar=(4, 3))
    assert annexb == 0
    m = CP.mkv_read((tmp_path / "a.mkv").read_bytes())
    assert (m["doctype"], m["codec"], m["timescale"]) == ("matroska", "V_MPEG4/ISO/AVC", 50000)
    assert (m["width"], m["height"], m["d_width"], m["d_height"]) == (w, h, w * 4 // 3, h)          # SAR widens the display size
    assert m["default_duration"] == 1001 * 1000000000 // 30000
    a = CP.avcc_read(m["avcc"])
    assert (a["sps"], a["pps"], a["profile"], a["level"]) == (sps, pps, sps[1], sps[3])
    assert len(m["frames"]) == len(frames) and [f["key"] for f in m["frames"]] == [idr for _, idr in frames]
    # timecodes in units of 50 us: pts * 1001/30000 s
    assert [f["timecode"] for f in m["frames"]] == [int((i * 1e9 * 1001 / 30000 + 0.5)) // 50000 for i in range(len(frames))]
    assert abs(m["duration"] - (m["frames"][-1]["timecode"] + m["default_duration"] / 50000)) < 1
    nals = [CP.length_prefixed_nals(f["data"]) for f in m["frames"]]
    assert nals[0][0] == sei and all(len(n) == 1 for n in nals[1:])                                   # the SEI rides with the first frame
    decode_check(w, h, sps, pps, [n[-1] for n in nals], recs)


def test_mkv_starts_a_new_cluster_when_the_16_bit_timecode_overflows(stream, tmp_path):
    w, h, sps, pps, frames, recs = stream
    mux(tmp_path / "slow.mkv", b"mkv", w, h, sps, pps, frames, fps=(1, 1))                         # 1 s per picture = 20000 ticks of 50 us
    m = CP.mkv_read((tmp_path / "slow.mkv").read_bytes())
    assert m["clusters"] >= 4 and [f["timecode"] for f in m["frames"]] == [20000 * i for i in range(len(frames))]


def test_flv(stream, tmp_path):
    w, h, sps, pps, frames, recs = stream
    annexb, sei = mux(tmp_path / "a.flv", b"auto", w, h, sps, pps, frames, fps=(25, 1))
    assert annexb == 0
    raw = (tmp_path / "a.flv").read_bytes()
    f = CP.flv_read(raw)
    assert (f["meta"]["width"], f["meta"]["height"], f["meta"]["framerate"], f["meta"]["videocodecid"]) == (w, h, 25.0, 7.0)
    n = len(frames)
    assert f["meta"]["filesize"] == len(raw) and abs(f["meta"]["duration"] - n / 25) < 1e-9          # (2 * largest - second largest) * timebase
    assert abs(f["meta"]["videodatarate"] - len(raw) * 8 / (n / 25 * 1000)) < 1e-6
    a = CP.avcc_read(f["avcc"])
    assert (a["sps"], a["pps"]) == (sps, pps)
    assert [x["dts"] for x in f["frames"]] == [40 * i for i in range(n)] and all(x["cts_offset"] == 0 for x in f["frames"])
    assert [x["key"] for x in f["frames"]] == [idr for _, idr in frames]
    nals = [CP.length_prefixed_nals(x["data"]) for x in f["frames"]]
    assert nals[0][0] == sei
    decode_check(w, h, sps, pps, [x[-1] for x in nals], recs)


def test_raw_and_unsupported(stream, tmp_path):
    w, h, sps, pps, frames, recs = stream
    annexb, _ = mux(tmp_path / "a.h264", b"auto", w, h, sps, pps, frames)
    assert annexb == 1
    es = (tmp_path / "a.h264").read_bytes()
    dec = O.h264_decode(es, len(recs), w, h)
    assert all(np.array_equal(d, r) for d, r in zip(dec, recs))
    assert not H.x264host_mux_open(str(tmp_path / "x.avi").encode(), b"auto", None)                   # needs libavformat: not built in


def check_mp4(path, w, h, sps, pps, frames, recs, sei, fps, sar):
    """own box reader: structure, tables, timing (mp4_lsmash.c semantics); -> the parsed file"""
    raw = path.read_bytes()
    m = CP.mp4_read(raw)
    n = len(frames)
    assert m["order"] == ["ftyp", "mdat", "moov"] and (m["major"], m["minor"], m["brands"]) == (b"mp42", 0, [b"mp42", b"mp41", b"isom"])
    assert (m["movie_timescale"], m["media_timescale"], m["handler"], m["tkhd_flags"], m["track_id"], m["next_track"]) == (600, fps[0], b"vide", 7, 1, 2)
    assert (m["width"], m["height"], m["depth"]) == (w, h, 0x18)
    dw, dh = w << 16, h << 16
    if sar != (1, 1):
        r = sar[0] / sar[1]
        dw, dh = (int(dw * r), dh) if r > 1 else (dw, int(dh / r))
    assert m["display"] == (dw, dh) and m.get("pasp", (1, 1)) == sar
    assert m["ext"] == ["avcC", "colr", "pasp", "btrt"] and m["colr"] == (b"nclx", 2, 2, 2, 0)           # a 1:1 SAR is still a stated SAR (mp4_lsmash.c:245)
    a = CP.avcc_read(m["avcc"])
    assert (a["sps"], a["pps"], a["profile"], a["level"]) == (sps, pps, sps[1], sps[3])
    # timing: dts = cts = i * timebase_num in a timescale of timebase_den; the last delta repeats the previous one
    assert m["deltas"] == [fps[1]] * n and m["media_duration"] == n * fps[1]
    assert [(x["dts"], x["cts"]) for x in m["samples"]] == [(i * fps[1], i * fps[1]) for i in range(n)]
    pres = int(n * fps[1] / fps[0] * 600)
    assert m["movie_duration"] == pres and m["track_duration"] == pres and m["edit"] == (pres, 0, 0x10000)
    assert [x["key"] for x in m["samples"]] == [idr for _, idr in frames]
    assert sum(m["chunks"]) == n and max(m["chunks"]) * fps[1] * 2 <= fps[0] + 2 * fps[1]                    # about half a second per chunk
    nals = [CP.length_prefixed_nals(x["data"]) for x in m["samples"]]
    assert nals[0][0] == sei and all(len(x) == 1 for x in nals[1:])                                         # the SEI rides with the first sample
    total = sum(len(x["data"]) for x in m["samples"])
    assert m["btrt"][0] == max(len(x["data"]) for x in m["samples"]) and abs(m["btrt"][2] - total * 8 * fps[0] / (n * fps[1])) < 1 and m["btrt"][1] >= m["btrt"][2] - 1
    decode_check(w, h, sps, pps, [x[-1] for x in nals], recs)
    return m


@pytest.mark.parametrize("fps,sar", [((25, 1), (1, 1)), ((30000, 1001), (4, 3)), ((24, 1), (8, 9))])
def test_mp4(stream, tmp_path, fps, sar):
    w, h, sps, pps, frames, recs = stream
    annexb, sei = mux(tmp_path / "a.mp4", b"auto", w, h, sps, pps, frames, fps=fps, sar=sar)
    assert annexb == 0
    check_mp4(tmp_path / "a.mp4", w, h, sps, pps, frames, recs, sei, fps, sar)


@pytest.mark.skipif(not __import__("os").path.exists(O.LSMASH_REF), reason="oracle/_ref/liblsmash_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("fps,sar", [((25, 1), (1, 1)), ((30000, 1001), (4, 3))])
def test_mp4_reads_back_through_the_reference_lsmash(stream, tmp_path, fps, sar):
    """the file demuxed by the L-SMASH of the reference tree (the library its mp4 output is built on, mp4_lsmash.c): every sample, its
    timestamps and sync flag, the timescales / durations / edit, the sample entry and the avcC parameter sets are what went in"""
    w, h, sps, pps, frames, recs = stream
    path = tmp_path / "b.mp4"
    annexb, sei = mux(path, b"mp4", w, h, sps, pps, frames, fps=fps, sar=sar)
    info, samples, data = O.lsmash_read_mp4(path)
    n = len(frames)
    assert (info.movie_timescale, info.media_timescale, info.n_samples, info.width, info.height) == (600, fps[0], n, w, h)
    assert (info.par_h, info.par_v) == sar
    pres = int(n * fps[1] / fps[0] * 600)
    assert (info.media_duration, info.movie_duration, info.track_duration) == (n * fps[1], pres, pres)
    assert (info.n_edits, info.edit_duration, info.edit_start_time, info.edit_rate) == (1, pres, 0, 0x10000)
    assert (info.primaries, info.transfer, info.matrix, info.full_range) == (2, 2, 2, 0)
    avcc = bytes(info.avcc[:info.avcc_size])
    assert sps in avcc and pps in avcc and avcc[4:8] == b"avcC"
    assert [(s.dts, s.cts, s.sync) for s in samples] == [(i * fps[1], i * fps[1], int(idr)) for i, (_, idr) in enumerate(frames)]
    expect = [(pref(sei) if i == 0 else b"") + pref(nal) for i, (nal, _) in enumerate(frames)]
    assert data == expect
    m = CP.mp4_read(path.read_bytes())
    assert [s.pos for s in samples] == [x["pos"] for x in m["samples"]]


def test_mp4_composition_offsets_and_chunks(stream, tmp_path):
    """timestamps that differ between decode and presentation order (what a stream with reordered pictures hands the muxer) go into a
    ctts box, long sequences into several chunks; the start offset (first dts < 0, x264's B-frame convention) is folded into the edit
    list's media_time — the container reader and the reference tree's L-SMASH agree on every sample"""
    w, h, sps, pps, frames, recs = stream
    path = tmp_path / "c.mp4"
    annexb = C.c_int()
    m = H.x264host_mux_open(str(path).encode(), b"mp4", C.byref(annexb))
    assert m and H.x264host_mux_set_param(m, w, h, 25, 1, 1, 25, 1, 1, 0) == 0
    sei = b"\x06\x05\x04test\x80"
    hs, hp, he = pref(sps), pref(pps), pref(sei)
    assert H.x264host_mux_write_headers(m, hs, len(hs), hp, len(hp), he, len(he)) > 0
    n = 40
    # decode order with dts starting one tick early, as x264 does for one B-frame of delay: pts pattern 0, 2, 1, 4, 3, ...
    pts = [0] + [i + 1 if i % 2 else i - 1 for i in range(1, n)]
    dts = [i - 1 for i in range(n)]
    for i in range(n):
        nal, idr = frames[i % len(frames)]
        payload = pref(nal)
        assert H.x264host_mux_write_frame(m, payload, len(payload), pts[i], dts[i], int(i % 8 == 0), 1) == len(payload)
    assert H.x264host_mux_close(m, max(pts), sorted(pts)[-2]) == 0
    r = CP.mp4_read(path.read_bytes())
    assert [(x["dts"], x["cts"]) for x in r["samples"]] == [(dts[i] + 1, pts[i] + 1) for i in range(n)]     # + start offset 1
    assert r["edit"][1] == 1 and [x["key"] for x in r["samples"]] == [i % 8 == 0 for i in range(n)]
    assert len(r["chunks"]) >= 3 and sum(r["chunks"]) == n                                                  # 40 samples at 25 fps: half-second chunks
    if __import__("os").path.exists(O.LSMASH_REF):
        info, samples, data = O.lsmash_read_mp4(path)
        assert info.n_samples == n and info.edit_start_time == 1
        assert [(s.dts, s.cts, s.sync) for s in samples] == [(dts[i] + 1, pts[i] + 1, int(i % 8 == 0)) for i in range(n)]
        assert data == [x["data"] for x in r["samples"]]
