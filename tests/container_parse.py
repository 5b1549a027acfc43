"""Minimal readers of the two container formats the file muxers write (tests only): Matroska / EBML and FLV.  Written against the
format definitions (matroska.org element table; Adobe FLV v10.1), not against the muxer code."""
import struct

MASTER_IDS = {0x1a45dfa3, 0x18538067, 0x1549a966, 0x1654ae6b, 0xae, 0xe0, 0x1f43b675}


def _vint(b, o, keep_marker):
    first = b[o]
    n = 1
    while n <= 8 and not (first & (0x80 >> (n - 1))):
        n += 1
    v = int.from_bytes(b[o:o + n], "big")
    if not keep_marker:
        v &= (1 << (7 * n)) - 1
    return v, n


def ebml_elements(b, o, end):
    """-> list of (id, payload bytes or [children], absolute payload offset)"""
    out = []
    while o < end:
        eid, n = _vint(b, o, True); o += n
        size, n = _vint(b, o, False); o += n
        unknown = size == (1 << (7 * n)) - 1
        stop = end if unknown else o + size
        if eid in MASTER_IDS:
            out.append((eid, ebml_elements(b, o, stop), o))
        else:
            out.append((eid, b[o:stop], o))
        o = stop
    return out


def find(elems, eid):
    return [e for e in elems if e[0] == eid]


def mkv_read(b):
    top = ebml_elements(b, 0, len(b))
    header = find(top, 0x1a45dfa3)[0][1]
    seg = find(top, 0x18538067)[0][1]
    info = find(seg, 0x1549a966)[0][1]
    track = find(find(seg, 0x1654ae6b)[0][1], 0xae)[0][1]
    video = find(track, 0xe0)[0][1]
    u = lambda e: int.from_bytes(e[1], "big")
    frames = []
    for cl in find(seg, 0x1f43b675):
        tc = u(find(cl[1], 0xe7)[0])
        for blk in find(cl[1], 0xa3):
            d = blk[1]
            tn, n = _vint(d, 0, False)
            delta = struct.unpack(">h", d[n:n + 2])[0]
            flags = d[n + 2]
            frames.append({"track": tn, "timecode": tc + delta, "key": bool(flags & 0x80), "discardable": bool(flags & 1), "data": d[n + 3:]})
    return {"doctype": find(header, 0x4282)[0][1].decode(), "timescale": u(find(info, 0x2ad7b1)[0]),
            "duration": struct.unpack(">f", find(info, 0x4489)[0][1])[0], "codec": find(track, 0x86)[0][1].decode(),
            "avcc": find(track, 0x63a2)[0][1], "default_duration": u(find(track, 0x23e383)[0]) if find(track, 0x23e383) else 0,
            "width": u(find(video, 0xb0)[0]), "height": u(find(video, 0xba)[0]), "d_width": u(find(video, 0x54b0)[0]),
            "d_height": u(find(video, 0x54ba)[0]), "frames": frames, "clusters": len(find(seg, 0x1f43b675))}


def avcc_read(a):
    assert a[0] == 1 and a[4] == 0xff and a[5] == 0xe1
    n = int.from_bytes(a[6:8], "big")
    sps = a[8:8 + n]
    o = 8 + n
    assert a[o] == 1
    m = int.from_bytes(a[o + 1:o + 3], "big")
    return {"profile": a[1], "compat": a[2], "level": a[3], "sps": sps, "pps": a[o + 3:o + 3 + m]}


def length_prefixed_nals(d):
    out, o = [], 0
    while o < len(d):
        n = int.from_bytes(d[o:o + 4], "big")
        out.append(d[o + 4:o + 4 + n]); o += 4 + n
    assert o == len(d)
    return out


def flv_read(b):
    assert b[:3] == b"FLV" and b[3] == 1 and b[4] == 1 and int.from_bytes(b[5:9], "big") == 9 and b[9:13] == b"\0\0\0\0"
    o, tags = 13, []
    while o < len(b):
        typ, size = b[o], int.from_bytes(b[o + 1:o + 4], "big")
        ts = int.from_bytes(b[o + 4:o + 7], "big") | (b[o + 7] << 24)
        assert b[o + 8:o + 11] == b"\0\0\0"
        body = b[o + 11:o + 11 + size]
        assert int.from_bytes(b[o + 11 + size:o + 15 + size], "big") == size + 11, "PreviousTagSize"
        tags.append((typ, ts, body)); o += 15 + size
    meta = {}
    typ, ts, m = tags[0]
    assert typ == 18 and m[0] == 2
    n = int.from_bytes(m[1:3], "big"); assert m[3:3 + n] == b"onMetaData"
    p = 3 + n
    assert m[p] == 8; cnt = int.from_bytes(m[p + 1:p + 5], "big"); p += 5
    for _ in range(cnt):
        k = int.from_bytes(m[p:p + 2], "big"); key = m[p + 2:p + 2 + k].decode(); p += 2 + k
        assert m[p] == 0; meta[key] = struct.unpack(">d", m[p + 1:p + 9])[0]; p += 9
    assert m[p:p + 3] == b"\0\0\x09"
    typ, ts, s = tags[1]
    assert typ == 9 and s[0] == 0x17 and s[1] == 0 and s[2:5] == b"\0\0\0"
    frames = []
    for typ, ts, v in tags[2:]:
        assert typ == 9 and v[1] == 1 and (v[0] & 15) == 7
        frames.append({"dts": ts, "cts_offset": int.from_bytes(v[2:5], "big"), "key": (v[0] >> 4) == 1, "data": v[5:]})
    return {"meta": meta, "avcc": s[5:], "frames": frames}


# ---- ISO base media (mp4) ----
def mp4_boxes(b, o, end):
    """-> [(type, payload start, box end)] of the boxes laid out in b[o:end]"""
    out = []
    while o + 8 <= end:
        size, typ, hdr = int.from_bytes(b[o:o + 4], "big"), b[o + 4:o + 8].decode("latin1"), 8
        if size == 1:
            size, hdr = int.from_bytes(b[o + 8:o + 16], "big"), 16
        elif size == 0:
            size = end - o
        assert size >= hdr and o + size <= end, f"box {typ} at {o} overruns its parent"
        out.append((typ, o + hdr, o + size))
        o += size
    assert o == end
    return out


def _child(b, boxes, typ):
    m = [x for x in boxes if x[0] == typ]
    assert len(m) == 1, f"expected one {typ} box, found {len(m)}"
    return m[0]


def mp4_read(b):
    """One-video-track mp4 as written by host/muxers.cpp -> dict with the tables resolved into per-sample records"""
    u32 = lambda o: int.from_bytes(b[o:o + 4], "big")
    u64 = lambda o: int.from_bytes(b[o:o + 8], "big")
    top = mp4_boxes(b, 0, len(b))
    r = {"order": [t for t, _, _ in top]}
    _, o, e = _child(b, top, "ftyp")
    r["major"], r["minor"], r["brands"] = b[o:o + 4], u32(o + 4), [b[i:i + 4] for i in range(o + 8, e, 4)]
    mdat = _child(b, top, "mdat")
    moov = mp4_boxes(b, *_child(b, top, "moov")[1:])
    _, o, e = _child(b, moov, "mvhd")
    assert b[o] == 0
    r["movie_timescale"], r["movie_duration"], r["next_track"] = u32(o + 12), u32(o + 16), u32(e - 4)
    trak = mp4_boxes(b, *_child(b, moov, "trak")[1:])
    _, o, e = _child(b, trak, "tkhd")
    r["tkhd_flags"], r["track_id"], r["track_duration"] = u32(o) & 0xffffff, u32(o + 12), u32(o + 20)
    r["display"] = (u32(e - 8), u32(e - 4))
    elst = _child(b, mp4_boxes(b, *_child(b, trak, "edts")[1:]), "elst")
    o = elst[1]
    assert b[o] == 1 and u32(o + 4) == 1
    r["edit"] = (u64(o + 8), u64(o + 16), u32(o + 24))
    mdia = mp4_boxes(b, *_child(b, trak, "mdia")[1:])
    _, o, e = _child(b, mdia, "mdhd")
    assert b[o] == 1
    r["media_timescale"], r["media_duration"] = u32(o + 20), u64(o + 24)
    _, o, e = _child(b, mdia, "hdlr")
    r["handler"] = b[o + 8:o + 12]
    minf = mp4_boxes(b, *_child(b, mdia, "minf")[1:])
    _child(b, minf, "vmhd"); _child(b, minf, "dinf")
    stbl = mp4_boxes(b, *_child(b, minf, "stbl")[1:])
    _, o, e = _child(b, stbl, "stsd")
    assert u32(o + 4) == 1
    entry = mp4_boxes(b, o + 8, e)
    assert len(entry) == 1 and entry[0][0] == "avc1"
    eo = entry[0][1]
    r["width"], r["height"], r["depth"] = int.from_bytes(b[eo + 24:eo + 26], "big"), int.from_bytes(b[eo + 26:eo + 28], "big"), int.from_bytes(b[eo + 74:eo + 76], "big")
    ext = mp4_boxes(b, eo + 78, entry[0][2])
    r["ext"] = [t for t, _, _ in ext]
    _, o, e = _child(b, ext, "avcC")
    r["avcc"] = b[o:e]
    _, o, e = _child(b, ext, "colr")
    r["colr"] = (b[o:o + 4], int.from_bytes(b[o + 4:o + 6], "big"), int.from_bytes(b[o + 6:o + 8], "big"), int.from_bytes(b[o + 8:o + 10], "big"), b[o + 10] >> 7)
    if "pasp" in r["ext"]:
        _, o, e = _child(b, ext, "pasp")
        r["pasp"] = (u32(o), u32(o + 4))
    _, o, e = _child(b, ext, "btrt")
    r["btrt"] = (u32(o), u32(o + 4), u32(o + 8))
    _, o, e = _child(b, stbl, "stts")
    deltas = []
    for i in range(u32(o + 4)):
        deltas += [u32(o + 12 + 8 * i)] * u32(o + 8 + 8 * i)
    _, o, e = _child(b, stbl, "stsz")
    assert u32(o + 4) == 0
    sizes = [u32(o + 12 + 4 * i) for i in range(u32(o + 8))]
    n = len(sizes)
    assert len(deltas) == n
    sync = [True] * n
    if any(t == "stss" for t, _, _ in stbl):
        _, o, e = _child(b, stbl, "stss")
        ss = {u32(o + 8 + 4 * i) for i in range(u32(o + 4))}
        sync = [i + 1 in ss for i in range(n)]
    offs = [0] * n
    if any(t == "ctts" for t, _, _ in stbl):
        _, o, e = _child(b, stbl, "ctts")
        offs = []
        for i in range(u32(o + 4)):
            offs += [u32(o + 12 + 8 * i)] * u32(o + 8 + 8 * i)
    _, o, e = _child(b, stbl, "stsc")
    runs = [(u32(o + 8 + 12 * i), u32(o + 12 + 12 * i), u32(o + 16 + 12 * i)) for i in range(u32(o + 4))]
    co = [x for x in stbl if x[0] in ("stco", "co64")]
    assert len(co) == 1
    w8 = 8 if co[0][0] == "co64" else 4
    o = co[0][1]
    chunk_off = [int.from_bytes(b[o + 8 + w8 * i:o + 8 + w8 * (i + 1)], "big") for i in range(u32(o + 4))]
    per_chunk = []
    for ci in range(len(chunk_off)):
        cur = [x for x in runs if x[0] <= ci + 1][-1]
        assert cur[2] == 1
        per_chunk.append(cur[1])
    assert sum(per_chunk) == n
    r["chunks"] = per_chunk
    samples, k, dts = [], 0, 0
    for ci, cnt in enumerate(per_chunk):
        pos = chunk_off[ci]
        for _ in range(cnt):
            assert mdat[1] <= pos and pos + sizes[k] <= mdat[2], "sample outside the mdat"
            samples.append({"dts": dts, "cts": dts + offs[k], "key": sync[k], "data": b[pos:pos + sizes[k]], "pos": pos})
            pos += sizes[k]; dts += deltas[k]; k += 1
    r["samples"], r["deltas"] = samples, deltas
    return r
