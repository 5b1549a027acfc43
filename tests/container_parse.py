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
