"""oracle/decide.py — CHECKER twin of the lookahead's decisions and the single-pass rate control (test infrastructure: only tests/ import it).

The product decides picture types and quantisers in host/encoder.cpp on the device's frame costs.  This file restates the same parts of libx264
a second time, independently of that code, in plain Python over the CPU checker's frame costs (oracle/slicetype.c through
tests/oracle_lib.OracleSlicetype), so that a session's decisions can be compared with something other than themselves:

  [x264-upstream] encoder/slicetype.c  x264_slicetype_decide (keyint / min-keyint, closed GOPs, the run of B pictures), x264_slicetype_analyse with
                                       --b-adapt 0 / 1 (the cost comparisons and thresholds of the "fast" B decision) / 2 (slicetype_path, slicetype_path_cost:
                                       the Viterbi search over the lengths of the window, i_delay = max(bframes, 3) * 4), scenecut / scenecut_internal
                                       (the bias growing with the distance from the last keyframe, the flash test under B pictures)
  [x264-upstream] encoder/ratecontrol.c  rate_estimate_qscale for CRF (short-term complexity blur, get_qscale, the I picture after P pictures taking the
                                       running P quantiser / ipratio, the very first picture), the B quantiser from its nearest references (+ pbratio
                                       offsets, half for a B-reference), accum_p_qp_update, x264_ratecontrol_start's rounding

Restated from memory of upstream like the rest of oracle/ (libx264 is not in the reference tree): parity unpinned.  Out of this twin's reach (not
restated here): macroblock-tree, weight analysis, AQ-weighted costs, ABR feedback, 2-pass.
"""
import math

AUTO, IDR, I, P, BREF, B = 0, 1, 2, 3, 4, 5


# [x264-upstream] encoder/ratecontrol.c qp2qscale / qscale2qp are single-float functions (powf / log2f): the C library's own, through ctypes
import ctypes as _C
_libm = _C.CDLL("libm.so.6")
_libm.powf.restype = _C.c_float; _libm.powf.argtypes = [_C.c_float, _C.c_float]
_libm.log2f.restype = _C.c_float; _libm.log2f.argtypes = [_C.c_float]
_f = lambda x: _C.c_float(x).value


def qp2qscale(qp):
    return _f(_f(0.85) * _libm.powf(2.0, _f(_f(_f(qp) - _f(12.0)) / _f(6.0))))


def qscale2qp(qscale):
    return _f(_f(12.0) + _f(_f(6.0) * _libm.log2f(_f(_f(qscale) / _f(0.85)))))


class Params:
    def __init__(self, mbw, mbh, keyint=250, min_keyint=0, scenecut=40, bframes=3, b_adapt=1, b_pyramid=1, b_bias=0, crf=23.0, qcomp=0.6, ip_factor=1.4,
                 pb_factor=1.3, qpmin=0, qpmax=51, fps=25.0):
        self.mbw, self.mbh = mbw, mbh
        self.keyint, self.scenecut, self.bframes, self.b_adapt, self.b_pyramid, self.b_bias = keyint, scenecut, bframes, b_adapt, b_pyramid, b_bias
        if min_keyint <= 0:          # validate_parameters: auto = min(keyint / 10, fps), then [1, keyint / 2 + 1]
            min_keyint = min(keyint // 10, int(fps))
        self.min_keyint = max(1, min(min_keyint, keyint // 2 + 1))
        # (x264_param_t carries these as single floats: the doubles of the rate control start from the float's value)
        self.crf, self.qcomp, self.ip_factor, self.pb_factor, self.qpmin, self.qpmax, self.fps = _f(crf), _f(qcomp), _f(ip_factor), _f(pb_factor), qpmin, qpmax, fps


class Frame:
    def __init__(self, index, slot):
        self.frame, self.slot, self.type, self.b_scenecut = index, slot, AUTO, 1


class Lookahead:
    """the pictures waiting for a type + the frame costs behind the decisions.  costs: an object with cost(s0, s1, sb, d0, d1) -> int,
    cost_est(slot, d0, d1) and intra_mbs(slot, d0) over numbered slots (tests/oracle_lib.OracleSlicetype)"""

    def __init__(self, params, costs):
        self.p, self.c = params, costs
        self.next = []                 # display order
        self.last_nonb = None
        self.last_keyframe = -params.keyint

    # slicetype_frame_cost of frames[b] predicted from frames[p0] (and frames[p1])
    def cost(self, fr, p0, p1, b):
        return self.c.cost(fr[p0].slot, fr[p1].slot, fr[b].slot, b - p0, p1 - b)

    def scenecut_internal(self, fr, p0, p1):
        p = self.p
        self.cost(fr, p0, p1, p1)
        f = fr[p1]
        icost, pcost = self.c.cost_est(f.slot, 0, 0), self.c.cost_est(f.slot, p1 - p0, 0)
        gop = f.frame - self.last_keyframe
        tmax = p.scenecut / 100.0
        tmin = tmax * 0.25
        if p.min_keyint == p.keyint:
            tmin = tmax
        if gop <= p.min_keyint // 4:
            bias = tmin / 4
        elif gop <= p.min_keyint:
            bias = tmin * gop / p.min_keyint
        else:
            bias = tmin + (tmax - tmin) * (gop - p.min_keyint) / (p.keyint - p.min_keyint)
        return pcost >= (1.0 - bias) * icost

    def scenecut(self, fr, p0, p1, real, num_frames, i_max_search):
        p = self.p
        if real and p.bframes:
            # a flash between two pictures of one scene is not a scene cut: look one picture past p1 (the whole run under the trellis)
            origmaxp1 = p0 + 1 + (p.bframes if p.b_adapt == 2 else 1)
            maxp1 = min(origmaxp1, num_frames)
            for curp1 in range(p1, maxp1 + 1):
                if not self.scenecut_internal(fr, p0, curp1):
                    for i in range(curp1, p0, -1):
                        fr[i].b_scenecut = 0
            for curp0 in range(p0, maxp1 + 1):
                if origmaxp1 > i_max_search or (curp0 < maxp1 and self.scenecut_internal(fr, curp0, maxp1)):
                    fr[curp0].b_scenecut = 0
        if not fr[p1].b_scenecut:
            return False
        return self.scenecut_internal(fr, p0, p1)

    def analyse(self, framecnt):
        """x264_slicetype_analyse over frames[0] = the last non-B picture and frames[1 ..] = the queue"""
        p = self.p
        fr = [self.last_nonb] + self.next[:framecnt]
        i_max_search = framecnt
        if not framecnt:
            return
        keyint_limit = p.keyint - fr[0].frame + self.last_keyframe - 1
        num_frames = min(framecnt, keyint_limit)
        orig_num_frames = num_frames
        if num_frames <= 0:
            fr[1].type = I
            return
        if fr[1].type in (AUTO, I, IDR) and p.scenecut and self.scenecut(fr, 0, 1, True, orig_num_frames, i_max_search):
            if fr[1].type == AUTO:
                fr[1].type = I
            return
        num_bframes, num_analysed = 0, num_frames
        if p.bframes:
            if p.b_adapt == 1:
                i_mb_count = (p.mbw - 2) * (p.mbh - 2) if p.mbw > 2 and p.mbh > 2 else p.mbw * p.mbh
                i = 0
                while i <= num_frames - 2:
                    cost2p1 = self.cost(fr, i, i + 2, i + 2)
                    if self.c.intra_mbs(fr[i + 2].slot, 2) > i_mb_count // 2:
                        fr[i + 1].type = P
                        fr[i + 2].type = P
                        i += 2
                        continue
                    cost1b1 = self.cost(fr, i, i + 2, i + 1)
                    cost1p0 = self.cost(fr, i, i + 1, i + 1)
                    cost2p0 = self.cost(fr, i + 1, i + 2, i + 2)
                    if cost1p0 + cost2p0 < cost1b1 + cost2p1:
                        fr[i + 1].type = P
                        i += 1
                        continue
                    fr[i + 1].type = B
                    j = i + 2
                    while j <= min(i + p.bframes, num_frames - 1):
                        pthresh = max(300 - (50 - p.b_bias) * (j - i - 1), 30)          # P_SENS_BIAS
                        pcost = self.cost(fr, i, j + 1, j + 1)
                        if pcost > pthresh * i_mb_count or self.c.intra_mbs(fr[j + 1].slot, j - i + 1) > i_mb_count // 3:
                            break
                        fr[j].type = B
                        j += 1
                    fr[j].type = P
                    i = j
                fr[num_frames].type = P
                while num_bframes < num_frames and fr[num_bframes + 1].type == B:
                    num_bframes += 1
            elif p.b_adapt == 2:
                # B_ADAPT_TRELLIS ([x264-upstream] encoder/slicetype.c slicetype_path / slicetype_path_cost): a Viterbi search over the lengths of the window —
                # the best path of every length built from the best paths of the shorter ones + a run of 0 .. bframes B pictures closed by a P
                if num_frames > 1:
                    best_paths = {0: "", 1: "P"}
                    for length in range(2, num_frames + 1):
                        self.path(fr, length, best_paths)
                    best = best_paths[num_frames % 17]
                    num_bframes = len(best) - len(best.lstrip("B"))
                    for j in range(1, num_frames):
                        fr[j].type = B if best[j - 1] == "B" else P
                fr[num_frames].type = P
            else:
                num_bframes = min(num_frames - 1, p.bframes)
                for j in range(1, num_frames):
                    fr[j].type = B if j % (num_bframes + 1) else P
                fr[num_frames].type = P
            # a scene cut inside the first run: the picture in front of it closes the run
            for j in range(1, num_bframes + 1):
                if p.scenecut and self.scenecut(fr, j, j + 1, False, orig_num_frames, i_max_search):
                    fr[j].type = P
                    num_analysed = j
                    break
            reset_start = min(num_bframes + 2, num_analysed + 1)
        else:
            for j in range(1, num_frames + 1):
                fr[j].type = P
            reset_start = 2
        # enforce the keyframe limit
        last_keyframe, last_possible = self.last_keyframe, 0
        j = 1
        while j <= num_frames:
            kd = fr[j].frame - last_keyframe
            last_possible = j                      # (no type is forced in this twin's sessions: every picture may become the keyframe)
            if kd >= p.keyint:
                last_possible = 0
                if fr[j].type != IDR:
                    fr[j].type = IDR
            if fr[j].type == I and kd >= p.min_keyint:
                fr[j].type = IDR
            if fr[j].type == IDR:
                last_keyframe = fr[j].frame
                if j > 1 and fr[j - 1].type in (B, BREF):
                    fr[j - 1].type = P
            j += 1
        # the types beyond the first run are decided again when their turn comes
        for j in range(reset_start, framecnt + 1):
            fr[j].type = AUTO

    def path_cost(self, fr, path, threshold):
        """slicetype_path_cost: the frame costs of the pictures along a path ('P' / 'B' / 'I' for frames 1 ..), given up once above the best so far"""
        p = self.p
        cost, loc, cur_nonb, n = 0, 1, 0, len(path)
        while loc <= n:
            next_nonb = loc
            while next_nonb <= n and path[next_nonb - 1] == "B":
                next_nonb += 1
            if next_nonb > n:
                break
            cost += self.cost(fr, cur_nonb, next_nonb, next_nonb) if path[next_nonb - 1] == "P" else self.cost(fr, next_nonb, next_nonb, next_nonb)
            if cost > threshold:
                break
            if p.b_pyramid and next_nonb - cur_nonb > 2:
                middle = cur_nonb + (next_nonb - cur_nonb) // 2
                cost += self.cost(fr, cur_nonb, next_nonb, middle)
                for nb in range(loc, middle):
                    if cost >= threshold:
                        break
                    cost += self.cost(fr, cur_nonb, middle, nb)
                for nb in range(middle + 1, next_nonb):
                    if cost >= threshold:
                        break
                    cost += self.cost(fr, middle, next_nonb, nb)
            else:
                for nb in range(loc, next_nonb):
                    if cost >= threshold:
                        break
                    cost += self.cost(fr, cur_nonb, next_nonb, nb)
            loc = next_nonb + 1
            cur_nonb = next_nonb
        return cost

    def path(self, fr, length, best_paths):
        """slicetype_path: the cheapest way to code frames 1 .. length, ending in a run of 0 .. bframes B pictures and a P"""
        p = self.p
        best_cost, best_possible, best = 1 << 62, 0, None
        for run in range(min(p.bframes + 1, length)):
            ln = length - (run + 1)
            cand = list(best_paths[ln % 17][:ln] + "B" * run + "P")
            possible = 1
            for i in range(1, length + 1):
                t = fr[i].type
                if t == AUTO:
                    continue
                if t in (B, BREF):
                    possible = possible and (i < ln or i == length or cand[i - 1] == "B")
                else:
                    possible = possible and (i < ln or cand[i - 1] != "B")
                    cand[i - 1] = "I" if t in (I, IDR) else "P"
            cand = "".join(cand)
            if possible or not best_possible:
                if possible and not best_possible:
                    best_cost = 1 << 62
                c = self.path_cost(fr, cand, best_cost)
                if c < best_cost:
                    best_cost, best_possible, best = c, possible, cand
        best_paths[length % 17] = best

    def decide(self, flushing, wait):
        """x264_slicetype_decide: -> (index of the picture closing the first run, its type) or None while more input is needed"""
        p = self.p
        n = len(self.next)
        if not n or (not flushing and n <= wait):
            return None
        # (a keyframe found behind a run of B pictures closed that run as P; it keeps its type until its own turn — x264 keeps i_type on the frame)
        for f in self.next:
            f.type = IDR if getattr(f, "pinned_idr", False) else AUTO
        if self.next[0].type == IDR:
            return 0, IDR
        if self.last_nonb is not None and ((p.bframes and p.b_adapt) or p.scenecut):
            self.analyse(min(n, wait + 1))
        bfr = 0
        while True:
            frm = self.next[bfr]
            if frm.frame - self.last_keyframe >= p.keyint:          # limit the GOP size
                frm.type = IDR
            if frm.type == I and frm.frame - self.last_keyframe >= p.min_keyint:
                frm.type = IDR
            if frm.type == IDR:                                       # closed GOPs: the picture in front of the keyframe ends the run as P
                self.last_keyframe = frm.frame
                if bfr > 0:
                    frm.pinned_idr = True
                    bfr -= 1
                    self.next[bfr].type = P
                    frm = self.next[bfr]
            if bfr == p.bframes or bfr + 1 >= n:
                if frm.type in (AUTO, B, BREF):
                    frm.type = P
            if frm.type == AUTO:
                frm.type = B
            elif frm.type not in (B, BREF):
                break
            bfr += 1
        return bfr, self.next[bfr].type


class RateControl:
    """single-pass CRF (x264 ratecontrol.c), constant frame rate, no macroblock-tree, no VBV"""

    def __init__(self, p):
        self.p = p
        self.cplxsum = self.cplxcount = 0.0
        dur = min(max(1.0 / p.fps, 0.01), 1.0)                      # CLIP_DURATION
        self.dur_ratio = dur / 0.04                                   # BASE_FRAME_DURATION
        self.rate_factor_constant = (p.mbw * p.mbh * (120.0 if p.bframes else 80.0)) ** (1.0 - p.qcomp) / qp2qscale(p.crf)
        self.ip_offset = 6.0 * _libm.log2f(p.ip_factor)             # x264_ratecontrol_init_reconfigurable: 6.0 * log2f( f_ip_factor )
        self.pb_offset = 6.0 * _libm.log2f(p.pb_factor)
        self.accum_p_qp = self.accum_p_norm = 0.0
        self.last_non_b_is_i = True                                   # x264_ratecontrol_new: last_non_b_pict_type = SLICE_TYPE_I
        self.last_qscale_for = [qp2qscale(p.crf)] * 2
        self.frames_done = 0

    def _accum(self, qp, is_i):
        self.accum_p_qp = self.accum_p_qp * 0.95 + (qp + self.ip_offset if is_i else qp)
        self.accum_p_norm = self.accum_p_norm * 0.95 + 1.0

    def nonb(self, is_i, satd):
        """-> (integer quantiser, float quantiser) of an I or P picture whose frame cost is satd"""
        p = self.p
        self.cplxsum = self.cplxsum * 0.5 + satd / self.dur_ratio
        self.cplxcount = self.cplxcount * 0.5 + 1.0
        if satd > 0:
            q = _f((self.cplxsum / self.cplxcount) ** (1.0 - p.qcomp) / self.rate_factor_constant)          # (rate_estimate_qscale's q is a float: every assignment rounds)
        else:
            q = _f(self.last_qscale_for[0 if is_i else 1])
        if is_i and p.keyint > 1 and not self.last_non_b_is_i:
            q = _f(qp2qscale(self.accum_p_qp / self.accum_p_norm) / p.ip_factor)
        elif self.frames_done == 0 and p.qcomp != 1.0:
            q = _f(qp2qscale(p.crf) / p.ip_factor)
        q = _f(min(max(q, qp2qscale(p.qpmin)), qp2qscale(p.qpmax)))
        self.last_qscale_for[0 if is_i else 1] = q
        if self.frames_done == 0:
            self.last_qscale_for[1] = q * p.ip_factor
        qpf = min(max(qscale2qp(q), p.qpmin), p.qpmax)
        self._accum(qpf, is_i)
        self.last_non_b_is_i = is_i
        self.frames_done += 1
        return min(max(int(qpf + 0.5), 1), 51), qpf

    def qp_avg_rc(self, qpm):
        """fdec->f_qp_avg_rc: rc->qpa_rc (a float) gathers qpm * mb_width row by row, x264_ratecontrol_end divides by the macroblock count"""
        a = 0.0
        for _ in range(self.p.mbh):
            a = _f(a + _f(_f(qpm) * self.p.mbw))
        return _f(a / (self.p.mbw * self.p.mbh))

    def b(self, poc, ref0, ref1, kept_as_ref):
        """a B picture's quantiser from its nearest references: ref = (poc, type, float quantiser)"""
        p = self.p
        i0, i1 = ref0[1] in (I, IDR), ref1[1] in (I, IDR)
        dt0, dt1 = abs(poc - ref0[0]), abs(poc - ref1[0])
        # x264's own types: float q0, q1, q (the references' f_qp_avg_rc), double offsets; then qp2qscale and x264_ratecontrol_start's qscale2qp
        q0, q1 = self.qp_avg_rc(ref0[2]), self.qp_avg_rc(ref1[2])
        if ref0[1] == BREF:
            q0 = _f(q0 - self.pb_offset / 2)
        if ref1[1] == BREF:
            q1 = _f(q1 - self.pb_offset / 2)
        if i0 and i1:
            q = _f(_f(_f(q0 + q1) / 2) + self.ip_offset)
        elif i0:
            q = q1
        elif i1:
            q = q0
        else:
            q = _f(_f(_f(q0 * dt1) + _f(q1 * dt0)) / (dt0 + dt1))
        q = _f(q + (self.pb_offset / 2 if kept_as_ref else self.pb_offset))
        q = qscale2qp(qp2qscale(q))
        q = min(max(q, p.qpmin), p.qpmax)
        self._accum(q, False)
        self.frames_done += 1
        return min(max(int(q + 0.5), p.qpmin), p.qpmax), q


def run_session(frames, params, costs, slots):
    """the pictures of `frames` (display order, I420 arrays) through the decisions: -> [(display index, type, qp, float qp)] in coding order.
    costs: tests/oracle_lib.OracleSlicetype created with `slots` slots; the twin puts every picture into slot (index mod slots)"""
    p = params
    la, rc = Lookahead(p, costs), RateControl(p)
    wait = p.bframes                                    # h->frames.i_delay without macroblock-tree: the run length
    if p.b_adapt == 2 and p.bframes:
        wait = max(p.bframes, 3) * 4                    # ... B_ADAPT_TRELLIS: the window of its path search
    out = []
    kept = {}                                           # display index -> (type, float qp) of the pictures kept as references

    def code_run(flushing):
        r = la.decide(flushing, wait)
        if r is None:
            return False
        j, closing = r
        run = la.next[:j + 1]
        closer = run[j]
        # x264_rc_analyse_slice: the closing picture's complexity is its frame cost as the type it got
        icost = costs.cost(closer.slot, closer.slot, closer.slot, 0, 0)
        pcost = icost
        if closing == P and la.last_nonb is not None:
            pcost = costs.cost(la.last_nonb.slot, closer.slot, closer.slot, closer.frame - la.last_nonb.frame, 0)
        is_i = closing in (I, IDR)
        qp, qpf = rc.nonb(is_i, icost if is_i else pcost)
        out.append((closer.frame, closing, qp, qpf))
        kept[closer.frame] = (closing, qpf)
        la.last_nonb = closer
        bref = (j - 1) // 2 if p.b_pyramid and j > 1 else -1
        for i in ([bref] if bref >= 0 else []) + [i for i in range(j) if i != bref]:
            f = run[i]
            t = BREF if i == bref else B
            before, after = max(k for k in kept if k < f.frame), min(k for k in kept if k > f.frame)        # nearest references in display order
            q, qf = rc.b(2 * f.frame, (2 * before,) + kept[before], (2 * after,) + kept[after], t == BREF)
            out.append((f.frame, t, q, qf))
            if t == BREF:
                kept[f.frame] = (BREF, qf)
        del la.next[:j + 1]
        return True

    for i, _ in enumerate(frames):
        costs.put(i % slots, frames[i])
        la.next.append(Frame(i, i % slots))
        while code_run(False):
            pass
    while la.next and code_run(True):
        pass
    return out
