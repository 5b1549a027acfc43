"""Independent numpy restatement of the NORMATIVE parts of ITU-T H.264 used to pin the oracle
(tests/test_oracle_spec.py).  Written straight from the clause text in matrix / per-sample form, on
purpose structured differently from oracle/*.c (which uses butterflies and shared helpers):
clause numbers are given per function.  Nothing here is imported by the product."""
import numpy as np

CF4 = np.array([[1, 1, 1, 1], [2, 1, -1, -2], [1, -1, -1, 1], [1, -2, 2, -1]], np.int64)
H4 = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]], np.int64)
H2 = np.array([[1, 1], [1, -1]], np.int64)
V4 = np.array([[10, 16, 13], [11, 18, 14], [13, 20, 16], [14, 23, 18], [16, 25, 20], [18, 29, 23]], np.int64)  # Table 8-? v(m, 0..2)
V8 = np.array([[20, 18, 32, 19, 25, 24], [22, 19, 35, 21, 28, 26], [26, 23, 42, 24, 33, 31],
               [28, 25, 45, 26, 35, 33], [32, 28, 51, 30, 40, 38], [36, 32, 58, 34, 46, 43]], np.int64)


def fwd4(x):
    """core forward transform W = Cf X Cf^T (the JM/x264 forward; integer exact)"""
    return CF4 @ x.astype(np.int64) @ CF4.T


def inv4(d):
    """8.5.12.2 (8-338 .. 8-354): rows, then columns, then (x+32)>>6"""
    d = d.astype(np.int64)
    f = np.empty_like(d)
    for i in range(4):
        e0, e1 = d[i, 0] + d[i, 2], d[i, 0] - d[i, 2]
        e2, e3 = (d[i, 1] >> 1) - d[i, 3], d[i, 1] + (d[i, 3] >> 1)
        f[i] = [e0 + e3, e1 + e2, e1 - e2, e0 - e3]
    r = np.empty_like(d)
    for j in range(4):
        g0, g1 = f[0, j] + f[2, j], f[0, j] - f[2, j]
        g2, g3 = (f[1, j] >> 1) - f[3, j], f[1, j] + (f[3, j] >> 1)
        r[:, j] = [g0 + g3, g1 + g2, g1 - g2, g0 - g3]
    return (r + 32) >> 6


def _inv8_1d(a):
    """8.5.13 one-dimensional transform on a length-8 vector (8-355 ..)"""
    e = [0] * 8
    e[0] = a[0] + a[4]
    e[1] = -a[3] + a[5] - a[7] - (a[7] >> 1)
    e[2] = a[0] - a[4]
    e[3] = a[1] + a[7] - a[3] - (a[3] >> 1)
    e[4] = (a[2] >> 1) - a[6]
    e[5] = -a[1] + a[7] + a[5] + (a[5] >> 1)
    e[6] = a[2] + (a[6] >> 1)
    e[7] = a[3] + a[5] + a[1] + (a[1] >> 1)
    f = [0] * 8
    f[0] = e[0] + e[6]
    f[1] = e[1] + (e[7] >> 2)
    f[2] = e[2] + e[4]
    f[3] = e[3] + (e[5] >> 2)
    f[4] = e[2] - e[4]
    f[5] = (e[3] >> 2) - e[5]
    f[6] = e[0] - e[6]
    f[7] = e[7] - (e[1] >> 2)
    return [f[0] + f[7], f[2] + f[5], f[4] + f[3], f[6] + f[1], f[6] - f[1], f[4] - f[3], f[2] - f[5], f[0] - f[7]]


def inv8(d):
    d = d.astype(np.int64)
    g = np.array([_inv8_1d([int(v) for v in d[i]]) for i in range(8)], np.int64)
    m = np.array([_inv8_1d([int(v) for v in g[:, j]]) for j in range(8)], np.int64).T
    return (m + 32) >> 6


def level_scale4(qp_rem, i, j):
    """8.5.9: normAdjust4x4 * flat weight 16"""
    if i % 2 == 0 and j % 2 == 0:
        return 16 * V4[qp_rem, 0]
    if i % 2 == 1 and j % 2 == 1:
        return 16 * V4[qp_rem, 1]
    return 16 * V4[qp_rem, 2]


def level_scale8(qp_rem, i, j):
    if i % 4 == 0 and j % 4 == 0:
        k = 0
    elif i % 2 == 1 and j % 2 == 1:
        k = 1
    elif i % 4 == 2 and j % 4 == 2:
        k = 2
    elif (i % 4 == 0 and j % 2 == 1) or (i % 2 == 1 and j % 4 == 0):
        k = 3
    elif (i % 4 == 0 and j % 4 == 2) or (i % 4 == 2 and j % 4 == 0):
        k = 4
    else:
        k = 5
    return 16 * V8[qp_rem, k]


def dequant4(c, qp):
    """8.5.12.1"""
    out = np.empty((4, 4), np.int64)
    for i in range(4):
        for j in range(4):
            ls = level_scale4(qp % 6, i, j)
            out[i, j] = (int(c[i, j]) * ls) << (qp // 6 - 4) if qp >= 24 else (int(c[i, j]) * ls + (1 << (3 - qp // 6))) >> (4 - qp // 6)
    return out


def dequant8(c, qp):
    """8.5.13 scaling"""
    out = np.empty((8, 8), np.int64)
    for i in range(8):
        for j in range(8):
            ls = level_scale8(qp % 6, i, j)
            out[i, j] = (int(c[i, j]) * ls) << (qp // 6 - 6) if qp >= 36 else (int(c[i, j]) * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)
    return out


def luma_dc_dequant(c, qp):
    """8.5.10: f = H c H, then scaling with LevelScale(qp%6,0,0)"""
    f = H4 @ c.astype(np.int64) @ H4
    ls = level_scale4(qp % 6, 0, 0)
    if qp >= 36:
        return (f * ls) << (qp // 6 - 6)
    return (f * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)


def chroma_dc_dequant(c, qp):
    """8.5.11.1/2 for 4:2:0"""
    f = H2 @ c.astype(np.int64) @ H2
    return ((f * level_scale4(qp % 6, 0, 0)) << (qp // 6)) >> 5


# ---- intra prediction, per-sample from the clause text; p(x,y) accessor with x,y >= -1 ----
def _p(img, x0, y0):
    return lambda x, y: int(img[y0 + y, x0 + x])


def pred4x4(img, x0, y0, mode, tr_avail):
    """8.3.1.2.1 - 8.3.1.2.9 (modes 0..8); DC variants via mode 9 (left), 10 (top), 11 (128)"""
    p0 = _p(img, x0, y0)

    def p(x, y):
        if y == -1 and x > 3 and not tr_avail:
            return p0(3, -1)
        return p0(x, y)
    o = np.zeros((4, 4), np.int64)
    for y in range(4):
        for x in range(4):
            if mode == 0:
                v = p(x, -1)
            elif mode == 1:
                v = p(-1, y)
            elif mode == 2:
                v = (sum(p(i, -1) for i in range(4)) + sum(p(-1, i) for i in range(4)) + 4) >> 3
            elif mode == 9:
                v = (sum(p(-1, i) for i in range(4)) + 2) >> 2
            elif mode == 10:
                v = (sum(p(i, -1) for i in range(4)) + 2) >> 2
            elif mode == 11:
                v = 128
            elif mode == 3:
                v = (p(6, -1) + 3 * p(7, -1) + 2) >> 2 if x == 3 and y == 3 else (p(x + y, -1) + 2 * p(x + y + 1, -1) + p(x + y + 2, -1) + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (p(x - y - 2, -1) + 2 * p(x - y - 1, -1) + p(x - y, -1) + 2) >> 2
                elif x < y:
                    v = (p(-1, y - x - 2) + 2 * p(-1, y - x - 1) + p(-1, y - x) + 2) >> 2
                else:
                    v = (p(0, -1) + 2 * p(-1, -1) + p(-1, 0) + 2) >> 2
            elif mode == 5:
                z = 2 * x - y
                if z in (0, 2, 4, 6):
                    v = (p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 1) >> 1
                elif z in (1, 3, 5):
                    v = (p(x - (y >> 1) - 2, -1) + 2 * p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(-1, y - 1) + 2 * p(-1, y - 2) + p(-1, y - 3) + 2) >> 2
            elif mode == 6:
                z = 2 * y - x
                if z in (0, 2, 4, 6):
                    v = (p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 1) >> 1
                elif z in (1, 3, 5):
                    v = (p(-1, y - (x >> 1) - 2) + 2 * p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(x - 1, -1) + 2 * p(x - 2, -1) + p(x - 3, -1) + 2) >> 2
            elif mode == 7:
                if y in (0, 2):
                    v = (p(x + (y >> 1), -1) + p(x + (y >> 1) + 1, -1) + 1) >> 1
                else:
                    v = (p(x + (y >> 1), -1) + 2 * p(x + (y >> 1) + 1, -1) + p(x + (y >> 1) + 2, -1) + 2) >> 2
            else:
                z = x + 2 * y
                if z in (0, 2, 4):
                    v = (p(-1, y + (x >> 1)) + p(-1, y + (x >> 1) + 1) + 1) >> 1
                elif z in (1, 3):
                    v = (p(-1, y + (x >> 1)) + 2 * p(-1, y + (x >> 1) + 1) + p(-1, y + (x >> 1) + 2) + 2) >> 2
                elif z == 5:
                    v = (p(-1, 2) + 3 * p(-1, 3) + 2) >> 2
                else:
                    v = p(-1, 3)
            o[y, x] = v
    return o


def pred16x16(img, x0, y0, mode):
    """8.3.3.1-8.3.3.4; modes 0 V,1 H,2 DC,3 plane,4 DC-left,5 DC-top,6 128"""
    p = _p(img, x0, y0)
    o = np.zeros((16, 16), np.int64)
    if mode == 3:
        H = sum((i + 1) * (p(8 + i, -1) - p(6 - i, -1)) for i in range(8))
        V = sum((i + 1) * (p(-1, 8 + i) - p(-1, 6 - i)) for i in range(8))
        a, b, c = 16 * (p(-1, 15) + p(15, -1)), (5 * H + 32) >> 6, (5 * V + 32) >> 6
    for y in range(16):
        for x in range(16):
            if mode == 0:
                v = p(x, -1)
            elif mode == 1:
                v = p(-1, y)
            elif mode == 2:
                v = (sum(p(i, -1) for i in range(16)) + sum(p(-1, i) for i in range(16)) + 16) >> 5
            elif mode == 4:
                v = (sum(p(-1, i) for i in range(16)) + 8) >> 4
            elif mode == 5:
                v = (sum(p(i, -1) for i in range(16)) + 8) >> 4
            elif mode == 6:
                v = 128
            else:
                v = min(255, max(0, (a + b * (x - 7) + c * (y - 7) + 16) >> 5))
            o[y, x] = v
    return o


def pred_chroma8x8(img, x0, y0, mode):
    """8.3.4.1-8.3.4.4 (4:2:0); modes 0 DC,1 H,2 V,3 plane,4 DC-left-only,5 DC-top-only,6 128"""
    p = _p(img, x0, y0)
    o = np.zeros((8, 8), np.int64)
    if mode == 3:
        H = sum((i + 1) * (p(4 + i, -1) - p(2 - i, -1)) for i in range(4))
        V = sum((i + 1) * (p(-1, 4 + i) - p(-1, 2 - i)) for i in range(4))
        a, b, c = 16 * (p(-1, 7) + p(7, -1)), (34 * H + 32) >> 6, (34 * V + 32) >> 6
    for y in range(8):
        for x in range(8):
            xo, yo = x & 4, y & 4
            top = sum(p(xo + i, -1) for i in range(4)) if mode in (0, 5) else 0
            left = sum(p(-1, yo + i) for i in range(4)) if mode in (0, 4) else 0
            if mode == 1:
                v = p(-1, y)
            elif mode == 2:
                v = p(x, -1)
            elif mode == 6:
                v = 128
            elif mode == 3:
                v = min(255, max(0, (a + b * (x - 3) + c * (y - 3) + 16) >> 5))
            elif mode == 4:
                v = (left + 2) >> 2
            elif mode == 5:
                v = (top + 2) >> 2
            else:  # 8.3.4.1-3 with both neighbours available
                if (xo, yo) in ((0, 0), (4, 4)):
                    v = (top + left + 4) >> 3
                elif xo == 4:
                    v = (top + 2) >> 2
                else:
                    v = (left + 2) >> 2
            o[y, x] = v
    return o


def pred8x8(img, x0, y0, mode, avail_tl, avail_tr):
    """8.3.2.2 with reference filtering 8.3.2.2.1; all of left/top available; modes 0..8"""
    p0 = _p(img, x0, y0)

    def praw(x, y):
        if y == -1 and x > 7 and not avail_tr:
            return p0(7, -1)
        return p0(x, y)
    pf = {}
    pf[(0, -1)] = (praw(-1, -1) + 2 * praw(0, -1) + praw(1, -1) + 2) >> 2 if avail_tl else (3 * praw(0, -1) + praw(1, -1) + 2) >> 2
    for x in range(1, 15):
        pf[(x, -1)] = (praw(x - 1, -1) + 2 * praw(x, -1) + praw(x + 1, -1) + 2) >> 2
    pf[(15, -1)] = (praw(14, -1) + 3 * praw(15, -1) + 2) >> 2
    if avail_tl:
        pf[(-1, -1)] = (praw(0, -1) + 2 * praw(-1, -1) + praw(-1, 0) + 2) >> 2
    pf[(-1, 0)] = (praw(-1, -1) + 2 * praw(-1, 0) + praw(-1, 1) + 2) >> 2 if avail_tl else (3 * praw(-1, 0) + praw(-1, 1) + 2) >> 2
    for y in range(1, 7):
        pf[(-1, y)] = (praw(-1, y - 1) + 2 * praw(-1, y) + praw(-1, y + 1) + 2) >> 2
    pf[(-1, 7)] = (praw(-1, 6) + 3 * praw(-1, 7) + 2) >> 2
    p = lambda x, y: pf[(x, y)]
    o = np.zeros((8, 8), np.int64)
    for y in range(8):
        for x in range(8):
            if mode == 0:
                v = p(x, -1)
            elif mode == 1:
                v = p(-1, y)
            elif mode == 2:
                v = (sum(p(i, -1) for i in range(8)) + sum(p(-1, i) for i in range(8)) + 8) >> 4
            elif mode == 3:
                v = (p(14, -1) + 3 * p(15, -1) + 2) >> 2 if x == 7 and y == 7 else (p(x + y, -1) + 2 * p(x + y + 1, -1) + p(x + y + 2, -1) + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (p(x - y - 2, -1) + 2 * p(x - y - 1, -1) + p(x - y, -1) + 2) >> 2
                elif x < y:
                    v = (p(-1, y - x - 2) + 2 * p(-1, y - x - 1) + p(-1, y - x) + 2) >> 2
                else:
                    v = (p(0, -1) + 2 * p(-1, -1) + p(-1, 0) + 2) >> 2
            elif mode == 5:
                z = 2 * x - y
                if z >= 0 and z % 2 == 0:
                    v = (p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 1) >> 1
                elif z > 0:
                    v = (p(x - (y >> 1) - 2, -1) + 2 * p(x - (y >> 1) - 1, -1) + p(x - (y >> 1), -1) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(-1, y - 2 * x - 1) + 2 * p(-1, y - 2 * x - 2) + p(-1, y - 2 * x - 3) + 2) >> 2
            elif mode == 6:
                z = 2 * y - x
                if z >= 0 and z % 2 == 0:
                    v = (p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 1) >> 1
                elif z > 0:
                    v = (p(-1, y - (x >> 1) - 2) + 2 * p(-1, y - (x >> 1) - 1) + p(-1, y - (x >> 1)) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(x - 2 * y - 1, -1) + 2 * p(x - 2 * y - 2, -1) + p(x - 2 * y - 3, -1) + 2) >> 2
            elif mode == 7:
                if y % 2 == 0:
                    v = (p(x + (y >> 1), -1) + p(x + (y >> 1) + 1, -1) + 1) >> 1
                else:
                    v = (p(x + (y >> 1), -1) + 2 * p(x + (y >> 1) + 1, -1) + p(x + (y >> 1) + 2, -1) + 2) >> 2
            else:
                z = x + 2 * y
                if z > 13:
                    v = p(-1, 7)
                elif z == 13:
                    v = (p(-1, 6) + 3 * p(-1, 7) + 2) >> 2
                elif z % 2 == 0:
                    v = (p(-1, y + (x >> 1)) + p(-1, y + (x >> 1) + 1) + 1) >> 1
                else:
                    v = (p(-1, y + (x >> 1)) + 2 * p(-1, y + (x >> 1) + 1) + p(-1, y + (x >> 1) + 2) + 2) >> 2
            o[y, x] = v
    return o


# ---- inter prediction samples ----
def luma_sample(img, xi, yi, xf, yf):
    """8.4.2.2.1: predicted luma sample at integer (xi,yi) + fraction (xf,yf) quarter units, reference
    picture `img` with coordinate clamping (8-241/8-242)."""
    h, w = img.shape

    def G(x, y):
        return int(img[min(max(y, 0), h - 1), min(max(x, 0), w - 1)])

    def tap(v):
        return v[0] - 5 * v[1] + 20 * v[2] + 20 * v[3] - 5 * v[4] + v[5]

    def clip(v):
        return min(255, max(0, v))
    b1 = lambda x, y: tap([G(x + k, y) for k in range(-2, 4)])
    h1 = lambda x, y: tap([G(x, y + k) for k in range(-2, 4)])
    bb = lambda x, y: clip((b1(x, y) + 16) >> 5)
    hh = lambda x, y: clip((h1(x, y) + 16) >> 5)
    jj = lambda x, y: clip((tap([b1(x, y + k) for k in range(-2, 4)]) + 512) >> 10)
    Gv, b, hv, j = G(xi, yi), bb(xi, yi), hh(xi, yi), jj(xi, yi)
    m, s = hh(xi + 1, yi), bb(xi, yi + 1)
    Hh, M = G(xi + 1, yi), G(xi, yi + 1)
    avg = lambda a, c: (a + c + 1) >> 1
    table = {(0, 0): Gv, (1, 0): avg(Gv, b), (2, 0): b, (3, 0): avg(b, Hh),
             (0, 1): avg(Gv, hv), (1, 1): avg(b, hv), (2, 1): avg(b, j), (3, 1): avg(b, m),
             (0, 2): hv, (1, 2): avg(hv, j), (2, 2): j, (3, 2): avg(j, m),
             (0, 3): avg(hv, M), (1, 3): avg(hv, s), (2, 3): avg(j, s), (3, 3): avg(s, m)}
    return table[(xf, yf)]


def chroma_sample(plane, xi, yi, xf, yf):
    """8.4.2.2.2 (no clamping needed by the caller's choice of coordinates)"""
    A, B, Cc, D = int(plane[yi, xi]), int(plane[yi, xi + 1]), int(plane[yi + 1, xi]), int(plane[yi + 1, xi + 1])
    return ((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * Cc + xf * yf * D + 32) >> 6


# ---- deblocking, one line of samples (8.7.2.3 / 8.7.2.4) ----
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101,
                    113, 127, 144, 162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16,
                   16, 17, 17, 18, 18]
TC0 = {1: [0] * 23 + [1] * 10 + [2] * 4 + [3] * 3 + [4] * 3 + [5, 6, 6, 7, 8, 9, 10, 11, 13],
       2: [0] * 21 + [1] * 10 + [2] * 4 + [3] * 3 + [4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17],
       }
TC0[3] = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5,
          6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25]


def deblock_line(p, q, bs, index_a, index_b, chroma):
    """p = [p0,p1,p2,p3], q = [q0,..]; returns filtered copies"""
    p, q = list(p), list(q)
    alpha, beta = ALPHA[index_a], BETA[index_b]
    if bs == 0 or not (abs(p[0] - q[0]) < alpha and abs(p[1] - p[0]) < beta and abs(q[1] - q[0]) < beta):
        return p, q
    clip3 = lambda lo, hi, v: min(hi, max(lo, v))
    clip1 = lambda v: min(255, max(0, v))
    ap, aq = abs(p[2] - p[0]), abs(q[2] - q[0])
    np_, nq = list(p), list(q)
    if bs < 4:
        tc0 = TC0[bs][index_a]
        tc = tc0 + 1 if chroma else tc0 + (1 if ap < beta else 0) + (1 if aq < beta else 0)
        delta = clip3(-tc, tc, (((q[0] - p[0]) << 2) + (p[1] - q[1]) + 4) >> 3)
        np_[0], nq[0] = clip1(p[0] + delta), clip1(q[0] - delta)
        if not chroma and ap < beta:
            np_[1] = p[1] + clip3(-tc0, tc0, (p[2] + ((p[0] + q[0] + 1) >> 1) - (p[1] << 1)) >> 1)
        if not chroma and aq < beta:
            nq[1] = q[1] + clip3(-tc0, tc0, (q[2] + ((p[0] + q[0] + 1) >> 1) - (q[1] << 1)) >> 1)
    else:
        small = abs(p[0] - q[0]) < ((alpha >> 2) + 2)
        if not chroma and ap < beta and small:
            np_[0] = (p[2] + 2 * p[1] + 2 * p[0] + 2 * q[0] + q[1] + 4) >> 3
            np_[1] = (p[2] + p[1] + p[0] + q[0] + 2) >> 2
            np_[2] = (2 * p[3] + 3 * p[2] + p[1] + p[0] + q[0] + 4) >> 3
        else:
            np_[0] = (2 * p[1] + p[0] + q[1] + 2) >> 2
        if not chroma and aq < beta and small:
            nq[0] = (p[1] + 2 * p[0] + 2 * q[0] + 2 * q[1] + q[2] + 4) >> 3
            nq[1] = (p[0] + q[0] + q[1] + q[2] + 2) >> 2
            nq[2] = (2 * q[3] + 3 * q[2] + q[1] + q[0] + p[0] + 4) >> 3
        else:
            nq[0] = (2 * q[1] + q[0] + p[1] + 2) >> 2
    return np_, nq


# ---- deblocking of a whole picture (8.7, frame macroblocks, P / I slices, 4:2:0) ----
QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]     # Table 8-15
_BLK = [[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]]                                                # 4x4 block index of (by, bx)


def deblock_picture(Y, U, V, mbs, mbw, mbh, alpha_off_div2, beta_off_div2, chroma_qp_index_offset):
    """Y [16 mbh, 16 mbw], U / V half size (int arrays, modified in place); mbs: records with fields type (0..2 intra, 4 / 5 inter, 6 skip),
    qp, nnz (bit per 4x4 luma block), cbp_luma, transform8x8, ref[4], mv[4][2] per 8x8.  Macroblocks in raster order; in each, the
    vertical edges left to right, then the horizontal edges top to bottom (8.7), luma and both chroma planes."""
    intra = lambda m: m["type"] <= 2

    def coded(m, bx, by):
        if m["type"] == 6:
            return False
        if m["transform8x8"]:
            return bool((int(m["cbp_luma"]) >> ((by >> 1) * 2 + (bx >> 1))) & 1)
        return bool((int(m["nnz"]) >> _BLK[by][bx]) & 1)

    def bs_of(mp, pbx, pby, mq, qbx, qby, mb_edge):
        if intra(mp) or intra(mq):
            return 4 if mb_edge else 3
        if coded(mp, pbx, pby) or coded(mq, qbx, qby):
            return 2
        pi, qi = (pby >> 1) * 2 + (pbx >> 1), (qby >> 1) * 2 + (qbx >> 1)
        if int(mp["ref"][pi]) != int(mq["ref"][qi]):
            return 1
        if abs(int(mp["mv"][pi][0]) - int(mq["mv"][qi][0])) >= 4 or abs(int(mp["mv"][pi][1]) - int(mq["mv"][qi][1])) >= 4:
            return 1
        return 0

    def filt(plane, x, y, dx, dy, bs, qp_p, qp_q, chroma):
        """one line of samples across the edge at (x, y): p_i = plane[y - (i+1) dy, x - (i+1) dx], q_i = plane[y + i dy, x + i dx]"""
        if chroma:
            qp_p = QPC[min(51, max(0, qp_p + chroma_qp_index_offset))]
            qp_q = QPC[min(51, max(0, qp_q + chroma_qp_index_offset))]
        av = (qp_p + qp_q + 1) >> 1
        ia, ib = min(51, max(0, av + 2 * alpha_off_div2)), min(51, max(0, av + 2 * beta_off_div2))
        n = 2 if chroma else 4
        p = [int(plane[y - (i + 1) * dy, x - (i + 1) * dx]) for i in range(n)]
        q = [int(plane[y + i * dy, x + i * dx]) for i in range(n)]
        if chroma:
            p += [0, 0]
            q += [0, 0]
        p2, q2 = deblock_line(p, q, bs, ia, ib, chroma)
        for i in range(n):
            plane[y - (i + 1) * dy, x - (i + 1) * dx] = p2[i]
            plane[y + i * dy, x + i * dx] = q2[i]

    for mby in range(mbh):
        for mbx in range(mbw):
            m = mbs[mby * mbw + mbx]
            for vertical in (True, False):
                for e in range(4):
                    if e == 0 and (mbx == 0 if vertical else mby == 0):
                        continue                                    # picture edge
                    if (e & 1) and m["transform8x8"]:
                        continue                                    # no 4-sample transform edge inside an 8x8 transform block
                    for k in range(16):                             # position along the edge
                        if vertical:
                            qbx, qby = e, k >> 2
                            mp = mbs[mby * mbw + mbx - 1] if e == 0 else m
                            pbx, pby = (3 if e == 0 else e - 1), qby
                        else:
                            qbx, qby = k >> 2, e
                            mp = mbs[(mby - 1) * mbw + mbx] if e == 0 else m
                            pbx, pby = qbx, (3 if e == 0 else e - 1)
                        bs = bs_of(mp, pbx, pby, m, qbx, qby, e == 0)
                        if not bs:
                            continue
                        x, y = (16 * mbx + 4 * e, 16 * mby + k) if vertical else (16 * mbx + k, 16 * mby + 4 * e)
                        filt(Y, x, y, 1 if vertical else 0, 0 if vertical else 1, bs, int(mp["qp"]), int(m["qp"]), False)
                        if not (e & 1) and not (k & 1):             # chroma: edges 0 and 2 of the luma grid, every second luma position
                            cx, cy = (8 * mbx + 2 * e, 8 * mby + (k >> 1)) if vertical else (8 * mbx + (k >> 1), 8 * mby + 2 * e)
                            for pl in (U, V):
                                filt(pl, cx, cy, 1 if vertical else 0, 0 if vertical else 1, bs, int(mp["qp"]), int(m["qp"]), True)
