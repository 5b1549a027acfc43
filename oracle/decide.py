"""oracle/decide.py — CHECKER twin of the lookahead's decisions and the single-pass rate control (test infrastructure: only tests/ import it).

The product decides picture types and quantisers in host/encoder.cpp on the device's frame costs.  This file restates the same parts of libx264
a second time, independently of that code, in plain Python over the CPU checker's frame costs (oracle/slicetype.c through
tests/oracle_lib.OracleSlicetype), so that a session's decisions can be compared with something other than themselves:

  [x264-upstream] encoder/slicetype.c  x264_slicetype_decide (keyint / min-keyint, closed GOPs, the run of B pictures), x264_slicetype_analyse with
                                       --b-adapt 0 / 1 (the "fast" B decision: path cost of ..BP against ..PP) / 2 (slicetype_path, slicetype_path_cost:
                                       the Viterbi search over the lengths of the window, i_delay = max(bframes, 3) * 4), scenecut / scenecut_internal
                                       (the bias growing with the distance from the last keyframe, the flash test under B pictures)
  [x264-upstream] encoder/ratecontrol.c  rate_estimate_qscale for CRF (short-term complexity blur, get_qscale, the I picture after P pictures taking the
                                       running P quantiser / ipratio, the very first picture), the B quantiser from its nearest references (+ pbratio
                                       offsets, half for a B-reference), accum_p_qp_update, x264_ratecontrol_start's rounding

  [x264-upstream] encoder/slicetype.c  the DEFAULT session's additions: macroblock_tree (the walk over the window with the types decided so far, the keyframe's own
                                       pass: lookahead_slicetype_decide's second analysis), i_delay = max(run length, rc-lookahead), the whole window analysed
                                       when psy and the tree are on, weights_analyse( b_lookahead = 1 ) in front of every P cost searched for the first time
                                       (--weightp, or X264_WEIGHTP_FAKE for the tree's weightdelta) and weights_analyse( b_lookahead = 0 ) of the P picture about to be coded
                                       (scales / offsets around the guess, chroma planes, the unified chroma denominator), the AQ offsets every picture arrives with
                                       (x264_adaptive_quant_frame, mode 1) as the tree's inverse quantiser scales and base
  [x264-upstream] encoder/ratecontrol.c  single-pass ABR: the rate factor from the bits window, the overflow pull, the qpstep clip, x264_ratecontrol_end's feedback (fed the session's real sizes)
  [x264-upstream] encoder/ratecontrol.c  get_qscale under macroblock-tree (the duration term alone, CRF shifted by 13.5 (1 - qcomp), qcompress 1)

Restated from memory of upstream like the rest of oracle/ (libx264 is not in the reference tree): parity unpinned.  Out of this twin's reach (not
restated here): VBV.  (The second pass has its twins at the end of this file: init_pass2 for the plan,
pass2_quantisers for the feedback on top of it.)
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
                 pb_factor=1.3, qpmin=0, qpmax=51, fps=25.0, mbtree=False, aq_strength=0.0, weightp=0, rc_lookahead=0, psy=True, zones=(), bitrate=0, rate_tolerance=1.0, qpstep=4, subme=7):
        self.mbw, self.mbh = mbw, mbh
        self.keyint, self.scenecut, self.bframes, self.b_adapt, self.b_pyramid, self.b_bias = keyint, scenecut, bframes, b_adapt, b_pyramid, b_bias
        if min_keyint <= 0:          # validate_parameters: auto = min(keyint / 10, fps), then [1, keyint / 2 + 1]
            min_keyint = min(keyint // 10, int(fps))
        self.min_keyint = max(1, min(min_keyint, keyint // 2 + 1))
        # the default session: macroblock-tree over rc_lookahead pictures, AQ mode 1 (aq_strength = --aq-strength x 1.0397f; 0 = off), --weightp
        self.mbtree, self.aq_strength, self.weightp, self.rc_lookahead, self.psy = mbtree, _f(aq_strength), weightp, rc_lookahead, psy
        self.weightp_fake = not weightp and mbtree and psy          # validate_parameters: X264_WEIGHTP_FAKE
        # single-pass ABR (--bitrate, kbit/s; 0 = CRF): rate_estimate_qscale's 1-pass branch with the feedback of x264_ratecontrol_end (the coded sizes come from the session)
        self.bitrate, self.rate_tolerance, self.qpstep = bitrate * 1000.0, max(_f(rate_tolerance), 0.01), qpstep
        self.subme = subme                 # the final weight analysis searches around its guess at the distances of the sub-pel level
        self.zones = list(zones)           # --zones: (start, end, 'q', qp) or (start, end, 'b', bitrate factor), display indices; the last one that holds a picture wins
        self.tree_strength = _f(_f(5.0) * _f(_f(1.0) - _f(qcomp)))   # macroblock_tree_finish: 5.0f * (1.0f - f_qcompress)
        # (x264_param_t carries these as single floats: the doubles of the rate control start from the float's value)
        self.crf, self.qcomp, self.ip_factor, self.pb_factor, self.qpmin, self.qpmax, self.fps = _f(crf), _f(qcomp), _f(ip_factor), _f(pb_factor), qpmin, qpmax, fps


class Frame:
    def __init__(self, index, slot):
        self.frame, self.slot, self.type, self.b_scenecut = index, slot, AUTO, 1
        self.raw = None                    # the source picture (weights_analyse reads its statistics)
        self.aq = None                     # f_qp_offset_aq: the AQ offsets it arrived with
        self.tree = None                   # f_qp_offset: what the last macroblock_tree pass over it left (starts as the AQ offsets)
        self.weighted_cost_delta = {}      # f_weighted_cost_delta[distance]


class Lookahead:
    """the pictures waiting for a type + the frame costs behind the decisions.  costs: an object with cost(s0, s1, sb, d0, d1) -> int,
    cost_est(slot, d0, d1) and intra_mbs(slot, d0) over numbered slots (tests/oracle_lib.OracleSlicetype)"""

    def __init__(self, params, costs):
        self.p, self.c = params, costs
        self.next = []                 # display order
        self.last_nonb = None
        self.last_keyframe = -params.keyint
        self.stats = {"lookahead_weights": 0, "final_weights": 0, "weightdelta": 0}          # how often the analysis found a weight / a finished picture carried a weightdelta

    # slicetype_frame_cost of frames[b] predicted from frames[p0] (and frames[p1])
    def cost(self, fr, p0, p1, b):
        w = None
        # slicetype_frame_cost: a P cost searched for the first time runs on the reference weighted by the lookahead's analysis
        if (self.p.weightp or self.p.weightp_fake) and p1 == b and b != p0 and self.c.cost_est(fr[b].slot, b - p0, 0) < 0 and self.c.mvs(fr[b].slot, 0, b - p0)[0][0] == 0x7fff:
            w = self.weights_analyse(fr[b], fr[p0], b - p0)
        return self.c.cost(fr[p0].slot, fr[p1].slot, fr[b].slot, b - p0, p1 - b, weight=w)

    def weights_analyse(self, fenc, ref, dist, b_lookahead=True):
        """x264's weights_analyse.  b_lookahead: luma alone, the guess alone, the reference in place (in front of a P cost searched for the first time) -> (scale, denom,
        offset) or None.  Else (the P picture about to be coded against the last non-B picture): scales / offsets around the guess at the distances of the sub-pel level, the
        reference motion-compensated by the lookahead's vectors, the chroma planes once luma has a weight, the chroma denominator unified
        -> None or {"luma": (scale, denom, offset), "cdenom": d, "chroma": [None | (scale, offset), None | (scale, offset)]}"""
        c, p = self.c, self.p
        fenc.weighted_cost_delta[dist] = 0.0
        sf, sr = [int(v) for v in c.pixel_stats(fenc.slot, fenc.raw)], [int(v) for v in c.pixel_stats(ref.slot, ref.raw)]
        nplanes = 1 if b_lookahead else 3
        if not b_lookahead:
            sf += [int(v) for v in c.chroma_stats(fenc.slot, fenc.raw)]
            sr += [int(v) for v in c.chroma_stats(ref.slot, ref.raw)]
        guess_scale, fenc_mean, ref_mean = [1.0] * 3, [0.0] * 3, [0.0] * 3
        for pl in range(nplanes):
            zero_bias = 0 if sr[2 * pl + 1] else 1
            fenc_var, ref_var = _f(float(sf[2 * pl + 1] + zero_bias)), _f(float(sr[2 * pl + 1] + zero_bias))
            guess_scale[pl] = _f(math.sqrt(_f(fenc_var / ref_var)))
            npix = _f(_f(p.mbw * 8) * _f(p.mbh * 8)) if pl else _f(_f(p.mbw * 16) * _f(p.mbh * 16))
            fenc_mean[pl], ref_mean[pl] = _f(_f(float(sf[2 * pl] + zero_bias)) / npix), _f(_f(float(sr[2 * pl] + zero_bias)) / npix)
        chroma_denom = 7
        if not b_lookahead:          # make sure both chroma scale factors fit
            while chroma_denom > 0:
                thresh = _f(127.0 / (1 << chroma_denom))
                if guess_scale[1] < thresh and guess_scale[2] < thresh:
                    break
                chroma_denom -= 1
        check_distance = [(0, 0), (0, 0), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (1, 1), (1, 1), (2, 1), (2, 1), (4, 2)]
        scale_dist, offset_dist = (0, 0) if b_lookahead else check_distance[min(max(p.subme, 0), 11)]
        roundf = lambda v: int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)
        on, w_scale, w_denom, w_off = [False] * 3, [1] * 3, [0] * 3, [0] * 3
        for pl in range(nplanes):
            if pl and not on[0]:
                break          # (the chroma planes are not checked if there was no luma weight)
            if abs(_f(ref_mean[pl] - fenc_mean[pl])) < 0.5 and abs(_f(1.0 - guess_scale[pl])) < 1.0 / 128.0:
                continue
            if pl:
                mindenom = chroma_denom
                minscale = min(max(roundf(_f(guess_scale[pl] * (1 << chroma_denom))), 0), 255)
                if minscale > 127:
                    on[1] = on[2] = False
                    break
            else:
                mindenom, minscale = 7, roundf(_f(guess_scale[0] * 128))
                while mindenom > 0 and minscale > 127:
                    mindenom -= 1
                    minscale >>= 1
                minscale = min(minscale, 127)
            minoff = 0
            if pl:
                cost_of = lambda w: c.weight_cost_chroma(fenc.slot, fenc.raw, ref.raw, dist, pl, w) & 0xffffffff
            else:
                c.cost(fenc.slot, fenc.slot, fenc.slot, 0, 0)                       # the picture's intra costs
                cost_of = lambda w: c.weight_cost(fenc.slot, ref.slot, dist, w) & 0xffffffff
            origscore = cost_of(None)
            minscore = origscore
            if not minscore:
                continue
            found = False
            for i_scale in range(min(max(minscale - scale_dist, 0), 127), min(max(minscale + scale_dist, 0), 127) + 1):
                cur_scale = i_scale
                cur_offset = int(_f(_f(fenc_mean[pl] - _f(_f(ref_mean[pl] * cur_scale) / (1 << mindenom))) + (0.5 if b_lookahead else 0.0)))
                if cur_offset < -128 or cur_offset > 127:
                    cur_offset = min(max(cur_offset, -128), 127)
                    cs = _f(_f(_f((1 << mindenom) * _f(fenc_mean[pl] - cur_offset)) / ref_mean[pl]) + 0.5)
                    cur_scale = int(min(max(cs, 0), 127))
                start_offset, end_offset = min(max(cur_offset - offset_dist, -128), 127), min(max(cur_offset + offset_dist, -128), 127)
                for i_off in range(start_offset, end_offset + 1):
                    score = cost_of((cur_scale, mindenom, i_off))
                    if score < minscore:
                        minscore, minscale, minoff, found = score, cur_scale, i_off, True
                    if minoff == start_offset and i_off != start_offset:          # the previous offset was better: no more
                        break
            if not pl:
                while mindenom > 0 and not (minscale & 1):                     # a smaller denominator if possible
                    mindenom -= 1
                    minscale >>= 1
            if not found or (minscale == 1 << mindenom and minoff == 0) or _f(_f(float(minscore)) / _f(float(origscore))) > _f(0.998):
                continue
            on[pl], w_scale[pl], w_denom[pl], w_off[pl] = True, minscale, mindenom, minoff
            if not pl:
                self.stats["lookahead_weights" if b_lookahead else "final_weights"] = self.stats.get("lookahead_weights" if b_lookahead else "final_weights", 0) + 1
                if p.weightp_fake:
                    fenc.weighted_cost_delta[dist] = _f(_f(float(minscore)) / _f(float(origscore)))
        if not on[0]:
            return None
        if b_lookahead:
            return (w_scale[0], w_denom[0], w_off[0])
        cdenom = 0
        if on[1] or on[2]:
            # unify the chroma denominator: a plane weighted alone leaves the other with the implicit scale 1 << denom, which 7 cannot carry
            cdenom = w_denom[1] if on[1] else w_denom[2]
            both = on[1] and on[2]
            while (not both and cdenom == 7) or (cdenom > 0 and not (on[1] and (w_scale[1] & 1)) and not (on[2] and (w_scale[2] & 1))):
                cdenom -= 1
                for i in (1, 2):
                    if on[i]:
                        w_scale[i] >>= 1
        return {"luma": (w_scale[0], w_denom[0], w_off[0]), "cdenom": cdenom, "chroma": [(w_scale[i], w_off[i]) if on[i] else None for i in (1, 2)]}

    def macroblock_tree(self, fr, num_frames, b_intra):
        """x264's macroblock_tree over frames[0 .. num_frames] with the types decided so far; the next picture to be coded (and the B-reference of its run) get
        their quantiser offsets (Frame.tree)"""
        p, c = self.p, self.c
        idx = 0 if b_intra else 1
        isb = lambda i: fr[i].type in (B, BREF)
        prop = lambda p0, p1, b, ref: c.propagate(fr[p0].slot, fr[p1].slot, fr[b].slot, b - p0, p1 - b, ref)

        def finish(i, ref0_distance):
            self.cost(fr, i, i, i)
            d = fr[i].weighted_cost_delta.get(ref0_distance, 0.0) if ref0_distance else 0.0
            weightdelta = _f(1.0 - d) if d > 0 else 0.0
            self.stats["weightdelta"] += int(weightdelta != 0.0)
            fr[i].tree = c.finish(fr[i].slot, p.tree_strength, weightdelta)
        if b_intra:
            self.cost(fr, 0, 0, 0)
        i = num_frames
        while i > 0 and isb(i):
            i -= 1
        last_nonb, bframes = i, 0
        if last_nonb < idx:
            return
        c.clear_propagate(fr[last_nonb].slot)
        while i > idx:
            i -= 1
            cur_nonb = i
            while isb(cur_nonb) and cur_nonb > 0:
                cur_nonb -= 1
            if cur_nonb < idx:
                break
            self.cost(fr, cur_nonb, last_nonb, last_nonb)
            c.clear_propagate(fr[cur_nonb].slot)
            bframes = last_nonb - cur_nonb - 1
            if p.b_pyramid and bframes > 1:
                middle = (bframes + 1) // 2 + cur_nonb
                self.cost(fr, cur_nonb, last_nonb, middle)
                c.clear_propagate(fr[middle].slot)
                while i > cur_nonb:
                    p0, p1 = (middle if i > middle else cur_nonb), (middle if i < middle else last_nonb)
                    if i != middle:
                        self.cost(fr, p0, p1, i)
                        prop(p0, p1, i, 0)
                    i -= 1
                prop(cur_nonb, last_nonb, middle, 1)
            else:
                while i > cur_nonb:
                    self.cost(fr, cur_nonb, last_nonb, i)
                    prop(cur_nonb, last_nonb, i, 0)
                    i -= 1
            prop(cur_nonb, last_nonb, last_nonb, 1)
            last_nonb = cur_nonb
        finish(last_nonb, last_nonb)
        if p.b_pyramid and bframes > 1:
            finish(last_nonb + (bframes + 1) // 2, 0)

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

    def analyse(self, framecnt, keyframe=False):
        """x264_slicetype_analyse over frames[0] = the last non-B picture and frames[1 ..] = the queue; keyframe: the pass over a keyframe just decided
        (frames[0]): nothing is decided, its macroblock-tree reaches frames[0]"""
        p = self.p
        fr = [self.last_nonb] + self.next[:framecnt]
        i_max_search = framecnt
        if not framecnt:
            return
        keyint_limit = p.keyint - fr[0].frame + self.last_keyframe - 1
        num_frames = min(framecnt, keyint_limit)
        orig_num_frames = num_frames
        if p.psy and p.mbtree:
            num_frames = framecnt              # psy-wise the pictures in front of a keyframe must not lose their share of the tree
        elif num_frames <= 0:
            fr[1].type = I
            return
        if not keyframe and fr[1].type in (AUTO, I, IDR) and p.scenecut and self.scenecut(fr, 0, 1, True, orig_num_frames, i_max_search):
            if fr[1].type == AUTO:
                fr[1].type = I
            return
        num_bframes, num_analysed = 0, num_frames
        if p.bframes:
            if p.b_adapt == 1:
                # X264_B_ADAPT_FAST (the x264 generation whose lookahead knows forced types): picture j is a B picture when the path "..BP" from the last non-B
                # picture costs less than "..PP" (slicetype_path_cost on both), runs no longer than --bframes
                isb = lambda i: fr[i].type in (B, BREF)
                last_nonb, num_bf = 0, p.bframes
                for j in range(1, num_frames):
                    if j - 1 > 0 and isb(j - 1):
                        num_bf -= 1
                    else:
                        last_nonb, num_bf = j - 1, p.bframes
                    if not num_bf:
                        if fr[j].type == AUTO or isb(j):
                            fr[j].type = P
                        continue
                    if fr[j].type != AUTO:
                        continue
                    if isb(j + 1):
                        fr[j].type = P
                        continue
                    run = j - last_nonb - 1
                    cost_p = self.path_cost(fr[last_nonb:], "B" * run + "PP", 1 << 62)
                    cost_b = self.path_cost(fr[last_nonb:], "B" * run + "BP", cost_p)
                    fr[j].type = B if cost_b < cost_p else P
                if fr[num_frames].type in (AUTO, B, BREF):
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
            reset_start = 1 if keyframe else min(num_bframes + 2, num_analysed + 1)
        else:
            for j in range(1, num_frames + 1):
                fr[j].type = P
            reset_start = 1 if keyframe else 2
        if p.mbtree:
            self.macroblock_tree(fr, min(num_frames, p.keyint), keyframe)
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
        if self.last_nonb is not None and ((p.bframes and p.b_adapt) or p.scenecut or p.mbtree):
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
        # x264_ratecontrol_new: under macroblock-tree qcompress = 1 (the tree does the complexity weighting) and CRF shifts by 13.5 (1 - qcomp)
        self.qcompress = 1.0 if p.mbtree else p.qcomp
        self.rate_factor_constant = (p.mbw * p.mbh * (120.0 if p.bframes else 80.0)) ** (1.0 - self.qcompress) / qp2qscale(p.crf + ((1.0 - p.qcomp) * 13.5 if p.mbtree else 0.0))
        self.ip_offset = 6.0 * _libm.log2f(p.ip_factor)             # x264_ratecontrol_init_reconfigurable: 6.0 * log2f( f_ip_factor )
        self.pb_offset = 6.0 * _libm.log2f(p.pb_factor)
        # x264_ratecontrol_new: rc->accum_p_norm = .01; rc->accum_p_qp = ABR_INIT_QP * rc->accum_p_norm (ABR_INIT_QP = the rate factor under CRF)
        self.accum_p_norm = .01
        self.accum_p_qp = p.crf * self.accum_p_norm
        self.last_non_b_is_i = True                                   # x264_ratecontrol_new: last_non_b_pict_type = SLICE_TYPE_I
        self.last_qscale_for = [qp2qscale(p.crf)] * 2
        self.frames_done = 0
        self.abr = p.bitrate > 0
        if self.abr:          # x264_ratecontrol_new / x264_ratecontrol_init_reconfigurable without VBV (ABR_INIT_QP = 24 outside CRF)
            self.accum_p_qp = 24.0 * self.accum_p_norm
            self.cplxr_sum = .01 * 7.0e5 ** self.qcompress * (p.mbw * p.mbh) ** 0.5
            self.wanted_bits_window = p.bitrate / p.fps
            self.last_qscale_for = [qp2qscale(24.0)] * 2
            self.lstep = 2.0 ** (p.qpstep / 6.0)
            self.total_bits, self.last_rceq = 0.0, 1.0

    def _accum(self, qp, is_i):
        self.accum_p_qp = self.accum_p_qp * 0.95 + (qp + self.ip_offset if is_i else qp)
        self.accum_p_norm = self.accum_p_norm * 0.95 + 1.0

    def nonb(self, is_i, satd, frame=-1):
        """-> (integer quantiser, float quantiser) of an I or P picture (display index `frame`) whose frame cost is satd"""
        p = self.p
        self.cplxsum = self.cplxsum * 0.5 + satd / self.dur_ratio
        self.cplxcount = self.cplxcount * 0.5 + 1.0
        if satd > 0:
            # get_qscale: under macroblock-tree the frame-duration term alone
            rceq = (1.0 / self.dur_ratio) ** (1.0 - p.qcomp) if p.mbtree else (self.cplxsum / self.cplxcount) ** (1.0 - self.qcompress)
            self.last_rceq = rceq
            q = _f(rceq / (self.wanted_bits_window / self.cplxr_sum if self.abr else self.rate_factor_constant))          # (rate_estimate_qscale's q is a float: every assignment rounds)
        else:
            q = _f(self.last_qscale_for[0 if is_i else 1])
        # get_qscale: a zone forces its quantiser or scales the picture's bits
        for z in reversed(p.zones):
            if z[0] <= frame <= z[1]:
                q = _f(qp2qscale(z[3]) if z[2] == 'q' else q / _f(z[3]))
                break
        overflow = 1.0
        if self.abr and satd > 0:          # pull towards the target: the bits so far against the time so far, within a buffer that grows with sqrt(time)
            time_done = self.frames_done / p.fps
            wanted_bits = time_done * p.bitrate
            if wanted_bits > 0:
                abr_buffer = 2 * p.rate_tolerance * p.bitrate * max(1.0, math.sqrt(time_done))
                overflow = min(max(1.0 + (self.total_bits - wanted_bits) / abr_buffer, .5), 2.0)
                q = _f(q * overflow)
        if is_i and p.keyint > 1 and not self.last_non_b_is_i:
            q = _f(qp2qscale(self.accum_p_qp / self.accum_p_norm) / p.ip_factor)
        elif self.frames_done > 0 and self.abr:          # asymmetric clipping against the last quantiser of the picture type
            lmin, lmax = self.last_qscale_for[0 if is_i else 1] / self.lstep, self.last_qscale_for[0 if is_i else 1] * self.lstep
            if overflow > 1.1 and self.frames_done > 3:
                lmax *= self.lstep
            elif overflow < 0.9:
                lmin /= self.lstep
            q = _f(min(max(q, lmin), lmax))
        elif self.frames_done == 0 and not self.abr and self.qcompress != 1.0:
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

    def end(self, bits, is_b, qpf):
        """x264_ratecontrol_end of single-pass ABR: what the picture took moves the rate factor of the pictures to come (a B picture's quantiser is an offset of its
        neighbours': its bits count divided by pbratio)"""
        if not self.abr:
            return
        self.total_bits += bits
        self.cplxr_sum += bits * qp2qscale(self.qp_avg_rc(qpf)) / (self.last_rceq * (abs(self.p.pb_factor) if is_b else 1.0))          # rc->qpa_rc: the float gathered row by row
        self.wanted_bits_window += self.p.bitrate / self.p.fps

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


def run_session(frames, params, costs, slots, aq_of=None, sizes=None):
    """the pictures of `frames` (display order, I420 arrays) through the decisions: -> [(display index, type, qp, float qp, offsets)] in coding order;
    offsets: the per-macroblock quantiser offsets the picture is coded with (float32 array; None when the session has neither tree nor AQ).
    costs: tests/oracle_lib.OracleSlicetype created with `slots` slots (do_edges = 1 for a macroblock-tree session); the twin puts every picture into
    slot (index mod slots).  aq_of(i420) -> the picture's AQ offsets (oracle_lib.aq_offsets with params.aq_strength) for sessions with AQ"""
    p = params
    la, rc = Lookahead(p, costs), RateControl(p)
    wait = p.bframes                                    # h->frames.i_delay without macroblock-tree: the run length
    if p.b_adapt == 2 and p.bframes:
        wait = max(p.bframes, 3) * 4                    # ... B_ADAPT_TRELLIS: the window of its path search
    if p.mbtree:
        wait = max(wait, p.rc_lookahead)                # ... and the tree's window
    out = []
    kept = {}                                           # display index -> (type, float qp) of the pictures kept as references
    weights = {}                                        # display index of a P picture -> its explicit weights (weights_analyse, b_lookahead = 0) or None
    nsize = [0]                                         # ABR: sizes (bytes) of the session's coded pictures, coding order, fed back as the session saw them

    def spent(is_b, qf):
        if sizes is not None:
            rc.end(sizes[nsize[0]] * 8.0, is_b, qf)
            nsize[0] += 1

    def code_run(flushing):
        r = la.decide(flushing, wait)
        if r is None:
            return False
        j, closing = r
        run = la.next[:j + 1]
        closer = run[j]
        if p.weightp and closing == P and la.last_nonb is not None:
            # x264_slicetype_decide: "analyse for weighted P frames" — the picture about to be coded against the last non-B picture
            weights[closer.frame] = la.weights_analyse(closer, la.last_nonb, closer.frame - la.last_nonb.frame, False)
        # x264_rc_analyse_slice: the closing picture's complexity is its frame cost as the type it got
        icost = costs.cost(closer.slot, closer.slot, closer.slot, 0, 0)
        pcost = icost
        if closing == P and la.last_nonb is not None:
            pcost = costs.cost(la.last_nonb.slot, closer.slot, closer.slot, closer.frame - la.last_nonb.frame, 0)
        is_i = closing in (I, IDR)
        qp, qpf = rc.nonb(is_i, icost if is_i else pcost, closer.frame)
        coded = [(closer, closing, qp, qpf)]
        spent(False, qpf)
        kept[closer.frame] = (closing, qpf)
        la.last_nonb = closer
        bref = (j - 1) // 2 if p.b_pyramid and j > 1 else -1
        for i in ([bref] if bref >= 0 else []) + [i for i in range(j) if i != bref]:
            f = run[i]
            t = BREF if i == bref else B
            before, after = max(k for k in kept if k < f.frame), min(k for k in kept if k > f.frame)        # nearest references in display order
            q, qf = rc.b(2 * f.frame, (2 * before,) + kept[before], (2 * after,) + kept[after], t == BREF)
            coded.append((f, t, q, qf))
            spent(True, qf)
            if t == BREF:
                kept[f.frame] = (BREF, qf)
        del la.next[:j + 1]
        if p.mbtree and is_i:
            # lookahead_slicetype_decide: "for MB-tree, we have to perform propagation analysis on I-frames too" — the analysis again with the keyframe as frames[0]
            for f in la.next:
                f.type = IDR if getattr(f, "pinned_idr", False) else AUTO
            closer.type = closing
            framecnt = max(min(len(la.next), wait + 1 - (j + 1)), 0)
            if framecnt > 0:
                la.analyse(framecnt, True)
            else:
                la.macroblock_tree([closer], 0, True)
        # x264_ratecontrol_mb_qp: kept pictures read f_qp_offset (what the tree left), the other B pictures f_qp_offset_aq
        for f, t, q, qf in coded:
            off = (f.aq if t == B else f.tree) if (p.mbtree or p.aq_strength) else None
            out.append((f.frame, t, q, qf, off))
        return True

    for i, _ in enumerate(frames):
        costs.put(i % slots, frames[i])
        fr = Frame(i, i % slots)
        fr.raw = frames[i]
        if p.mbtree or p.aq_strength:
            fr.aq = aq_of(frames[i]) if p.aq_strength else None
            if p.mbtree:
                costs.set_aq(fr.slot, fr.aq)
            import numpy as _np
            fr.aq = fr.aq if fr.aq is not None else _np.zeros(p.mbw * p.mbh, _np.float32)
            fr.tree = fr.aq
        la.next.append(fr)
        while code_run(False):
            pass
    while la.next and code_run(True):
        pass
    run_session.last_stats = la.stats
    run_session.last_weights = weights
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------------------------------------
# 2-pass: x264's init_pass2 ([x264-upstream] encoder/ratecontrol.c) over a statistics file — the complexity blur, get_qscale, get_diff_limited_q, the smoothing of the curve,
# the search for the rate factor that makes qscale2bits add up to the requested size.  Doubles as in x264; `mask` of the P-quantiser accumulator is a float there.
# No VBV, no zones, no macroblock-tree file (none of them is in the product's second pass).
class StatEntry:
    def __init__(self, line):
        f = dict(kv.split(":", 1) for kv in line.split() if ":" in kv)
        self.frame, self.out, self.type = int(f["in"]), int(f["out"]), f["type"]
        self.qscale = qp2qscale(_f(float(f["q"])))
        self.tex, self.mv, self.misc, self.icount = int(f["tex"]), int(f["mv"]), int(f["misc"]), int(f["imb"])
        self.kind = 0 if self.type in "Ii" else 1 if self.type == "P" else 2          # SLICE_TYPE_I / _P / _B as the rate control groups them
        self.kept_as_ref = self.type != "b"
        self.blurred = self.new_qscale = self.expected_bits = 0.0


def qscale2bits(e, qscale):
    qscale = max(qscale, 0.1)
    return (e.tex + .1) * (e.qscale / qscale) ** 1.1 + e.mv * (max(e.qscale, 1) / max(qscale, 1)) ** 0.5 + e.misc


def init_pass2(stat_lines, nmb, bitrate_kbps, fps=25.0, bframes=3, qcomp=0.6, qblur=0.5, cplxblur=20.0, ip_factor=1.4, pb_factor=1.3, qpmin=0, qpmax=51, qpstep=4):
    """-> the entries in display order with new_qscale (the plan) and expected_bits (what should have been spent when the picture starts, coding order)"""
    ent = [StatEntry(ln) for ln in stat_lines if not ln.startswith("#")]
    E = [None] * len(ent)
    for e in ent:
        E[e.frame] = e
    n = len(E)
    qcomp, qblur, cplxblur, ipf, pbf = _f(qcomp), _f(qblur), _f(cplxblur), abs(_f(ip_factor)), abs(_f(pb_factor))
    frame_duration = min(max(1.0 / fps, 0.01), 1.0) / 0.04          # CLIP_DURATION / BASE_FRAME_DURATION, constant frame rate
    all_available_bits = bitrate_kbps * 1000.0 * (n / fps)
    filter_size = int(qblur * 4) | 1
    base_cplx = nmb * (120 if bframes else 80)
    lstep, lmin, lmax = 2.0 ** (qpstep / 6.0), qp2qscale(qpmin), qp2qscale(qpmax)
    assert all_available_bits >= sum(e.misc for e in E), "requested bitrate is too low"
    for i, e in enumerate(E):          # blur the complexities
        weight_sum = cplx_sum = 0.0
        weight, j = 1.0, 1
        while j < cplxblur * 2 and j < n - i:
            r = E[i + j]
            weight *= 1 - _f(_f(r.icount) / _f(nmb)) ** 2
            if weight < .0001:
                break
            g = weight * math.exp(-j * j / 200.0)
            weight_sum += g
            cplx_sum += g * (qscale2bits(r, 1) - r.misc) / frame_duration
            j += 1
        weight, j = 1.0, 0
        while j <= cplxblur * 2 and j <= i:
            r = E[i - j]
            g = weight * math.exp(-j * j / 200.0)
            weight_sum += g
            cplx_sum += g * (qscale2bits(r, 1) - r.misc) / frame_duration
            weight *= 1 - _f(_f(r.icount) / _f(nmb)) ** 2
            if weight < .0001:
                break
            j += 1
        e.blurred = _f(cplx_sum / weight_sum)          # (ratecontrol_entry_t keeps blurred_complexity as a float)
    st = {"last_q": [0.0] * 3, "last_non_b": -1, "last_accum_p_norm": 1.0, "accum_p_norm": 0.0, "accum_p_qp": 0.0}

    def get_qscale(e, rate_factor):
        q = e.blurred ** (1 - qcomp)
        return st["last_q"][e.kind] if not math.isfinite(q) or e.tex + e.mv == 0 else q / rate_factor

    def get_diff_limited_q(e, q):
        k = e.kind
        last_p_q = st["last_q"][1]
        last_non_b_q = st["last_q"][st["last_non_b"]] if st["last_non_b"] >= 0 else q          # (last_non_b_pict_type = -1 at the start: never read before it is set)
        if k == 0:
            iq = q
            if st["accum_p_norm"] <= 0:
                q = iq
            elif ip_factor < 0:
                q = iq / ipf
            else:
                pq = qp2qscale(st["accum_p_qp"] / st["accum_p_norm"])
                q = pq / ipf if st["accum_p_norm"] >= 1 else st["accum_p_norm"] * pq / ipf + (1 - st["accum_p_norm"]) * iq
        elif k == 2:
            if pb_factor > 0:
                q = last_non_b_q
            if not e.kept_as_ref:
                q *= pbf
        elif st["last_non_b"] == 1 and e.tex == 0:
            q = last_p_q
        if st["last_non_b"] == k and (k != 0 or st["last_accum_p_norm"] < 1):
            last_q = st["last_q"][k]
            q = min(max(q, last_q / lstep), last_q * lstep)
        st["last_q"][k] = q
        if k != 2:
            st["last_non_b"] = k
        if k == 0:
            st["last_accum_p_norm"], st["accum_p_norm"], st["accum_p_qp"] = st["accum_p_norm"], 0.0, 0.0
        if k == 1:
            mask = _f(1 - _f(_f(e.icount) / _f(nmb)) ** 2)          # (a float in x264)
            st["accum_p_qp"] = mask * (qscale2qp(q) + st["accum_p_qp"])
            st["accum_p_norm"] = mask * (1 + st["accum_p_norm"])
        return q

    expected_bits = 1.0
    st["last_q"] = [base_cplx ** (1 - qcomp)] * 3
    for e in E:
        q = get_qscale(e, 1.0)
        expected_bits += qscale2bits(e, q)
        st["last_q"][e.kind] = q
    step_mult = all_available_bits / expected_bits
    rate_factor, step = 0.0, 1E4 * step_mult
    while step > 1E-7 * step_mult:
        expected_bits = 0.0
        rate_factor += step
        st.update(last_non_b=-1, last_accum_p_norm=1.0, accum_p_norm=0.0, accum_p_qp=0.0)
        st["last_q"] = [base_cplx ** (1 - qcomp) / rate_factor] * 3
        qs = []
        for e in E:
            qs.append(get_qscale(e, rate_factor))
            st["last_q"][e.kind] = qs[-1]
        for i in range(n - 1, -1, -1):          # fixed I / B quantisers relative to P
            qs[i] = get_diff_limited_q(E[i], qs[i])
        if filter_size > 1:                     # smooth the curve over pictures of one kind
            bl = []
            for i in range(n):
                q = ssum = 0.0
                for j in range(filter_size):
                    idx = i + j - filter_size // 2
                    d = idx - i
                    coeff = 1.0 if qblur == 0 else math.exp(-d * d / (qblur * qblur))
                    if idx < 0 or idx >= n or E[i].kind != E[idx].kind:
                        continue
                    q += qs[idx] * coeff
                    ssum += coeff
                bl.append(q / ssum)
        else:
            bl = qs
        for e, q in zip(E, bl):
            e.new_qscale = min(max(q, lmin), lmax)
            expected_bits += qscale2bits(e, e.new_qscale)
        if expected_bits > all_available_bits:
            rate_factor -= step
        step *= 0.5
    acc = 0.0
    for e in sorted(E, key=lambda x: x.out):
        e.expected_bits = acc
        acc += qscale2bits(e, e.new_qscale)
    return E


def pass2_quantisers(E, sizes, bitrate_kbps, fps=25.0, rate_tolerance=1.0, qpmin=0, qpmax=51):
    """rate_estimate_qscale's 2-pass branch over a plan of init_pass2, fed the sizes (bytes, coding order) the second pass' pictures really had:
    -> [(display index, integer quantiser, float quantiser)] in coding order"""
    n = len(E)
    order = sorted(E, key=lambda e: e.out)
    final_bits = order[-1].expected_bits          # entry_out[num_entries - 1]->expected_bits: what should have been spent BEFORE the last picture
    lmin, lmax = qp2qscale(qpmin), qp2qscale(qpmax)
    total_bits = expected_sum = 0.0
    out = []
    for k, e in enumerate(order):          # k = h->i_frame: pictures coded so far
        abr_buffer = 2 * max(_f(rate_tolerance), 0.01) * bitrate_kbps * 1000.0
        if n > k:          # the buffer shrinks towards the end of the video
            video_pos = e.expected_bits / final_bits if final_bits > 0 else 1.0
            abr_buffer *= 0.5 * max(math.sqrt((1 - video_pos) * n), 0.5)
        diff = int(total_bits) - int(e.expected_bits)
        q = e.new_qscale / min(max((abr_buffer - diff) / abr_buffer, .5), 2.0)
        if k >= fps and expected_sum >= 1:          # the achieved against the expected bitrate so far
            w = min(max(k / n * 100, 0.0), 1.0)
            q *= (total_bits / expected_sum) ** w
        q = min(max(q, lmin), lmax)
        qpf = min(max(qscale2qp(q), qpmin), qpmax)
        out.append((e.frame, min(max(int(qpf + 0.5), 1), 51), qpf))
        total_bits += sizes[k] * 8.0
        expected_sum += qscale2bits(e, qp2qscale(qpf))
    return out
