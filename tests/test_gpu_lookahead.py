"""-m gpu: lookahead frame cost (x264gpu_lookahead_frame_cost; [x264-upstream] slicetype.c x264_slicetype_frame_cost) vs
oracle/lookahead.c, bit-exact: per-block intra / best costs and the frame sums, over consecutive pictures (the motion field of
one picture predicts the next), scene changes and resets."""
import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frames

pytestmark = pytest.mark.gpu


def run_case(w, h, frames, resets=(), me_range=16, subme=7):
    from gpu_enc import GpuLookahead
    ol, gl = O.OracleLookahead(w, h, me_range, subme), GpuLookahead(w, h, 1, me_range, subme)
    mbw = (w + 15) // 16
    for i, f in enumerate(frames):
        rs = i == 0 or i in resets
        o_out, o_blk = ol.frame_cost(f, rs)
        g_out, g_blk = gl.frame_cost([f], rs)
        if not np.array_equal(g_blk[0], o_blk):
            bad = np.nonzero((g_blk[0] != o_blk).any(1))[0]
            j = int(bad[0])
            pytest.fail(f"{w}x{h} frame {i}: {len(bad)} blocks differ; first {j} (x={j % mbw}, y={j // mbw}) gpu={g_blk[0][j]} oracle={o_blk[j]}")
        assert np.array_equal(g_out[0], o_out), f"{w}x{h} frame {i}: sums gpu={g_out[0]} oracle={o_out}"
    ol.close(); gl.close()


@pytest.mark.parametrize("w,h,n,kw", [
    (176, 144, 5, {}),
    (352, 288, 4, {}),
    (208, 120, 4, {}),                       # odd number of block rows (8 macroblock rows -> 4 groups, 120 -> 8 rows) and ragged height
    (80, 48, 4, {}),                         # 5 x 3 blocks: odd in both directions
    (32, 32, 3, {}),                         # 2 x 2 blocks: every block counts for the frame score
    (16, 16, 3, {}),
    (352, 288, 3, dict(me_range=8, subme=1)),       # subme <= 1: DC/H/V only
    (720, 304, 3, {}),
])
def test_lookahead_cost_bitexact(gpu, w, h, n, kw):
    run_case(w, h, synth_frames(w, h, n, seed=w + h), **kw)


def test_lookahead_scene_change_and_reset(gpu):
    w, h = 352, 288
    a, b = synth_frames(w, h, 4, seed=5), synth_frames(w, h, 3, seed=77)
    run_case(w, h, a + b + a[:2], resets=(9,))


def test_lookahead_streams_are_independent(gpu):
    from gpu_enc import GpuLookahead
    w, h, S = 176, 144, 3
    seqs = [synth_frames(w, h, 4, seed=10 + s) for s in range(S)]
    gl = GpuLookahead(w, h, S)
    ols = [O.OracleLookahead(w, h) for _ in range(S)]
    for i in range(4):
        g_out, g_blk = gl.frame_cost([seqs[s][i] for s in range(S)], i == 0)
        for s in range(S):
            o_out, o_blk = ols[s].frame_cost(seqs[s][i], i == 0)
            assert np.array_equal(g_out[s], o_out) and np.array_equal(g_blk[s], o_blk), f"stream {s} frame {i}"


def test_scene_change_is_visible_in_the_costs(gpu):
    """continuous content: P cost well below the intra cost; a hard cut: every block intra (what scenecut keys on)"""
    from gpu_enc import GpuLookahead
    w, h = 352, 288
    a, b = synth_frames(w, h, 3, seed=5), synth_frames(w, h, 2, seed=77)
    gl = GpuLookahead(w, h)
    outs = [gl.frame_cost([f], i == 0)[0][0] for i, f in enumerate(a + b)]
    assert outs[2][1] < 0.8 * outs[2][0]
    assert outs[3][1] >= 0.95 * outs[3][0] and outs[3][2] > 0.8 * outs[3][3]


@pytest.mark.parametrize("w,h,n,strength,with_aq", [(352, 288, 7, 2.0, True), (208, 120, 5, 2.0, False), (80, 48, 4, 3.0, True), (720, 304, 4, 1.0, True), (16, 16, 3, 2.0, True), (1920, 1080, 3, 2.0, True)])
def test_aq_offsets_and_mbtree_bitexact(gpu, w, h, n, strength, with_aq):
    """x264gpu_lookahead_aq_offsets and x264gpu_lookahead_mbtree (macroblock_tree over n consecutive pictures) vs oracle/lookahead.c: x264's single-float
    expressions, the same bits on both sides (strength = 5.0f * (1.0f - qcomp): 2.0 at the default 0.6)"""
    import torch
    from gpu_enc import GpuLookahead
    frames = synth_frames(w, h, n, seed=3 * w + h)
    ol, gl = O.OracleLookahead(w, h), GpuLookahead(w, h)
    bw, bh = (w + 15) // 16, (h + 15) // 16
    o_infos, g_infos, o_aqs, g_aqs = [], [], [], []
    for i, f in enumerate(frames):
        o_infos.append(ol.frame_cost(f, i == 0)[1])
        g_infos.append(torch.from_numpy(gl.frame_cost([f], i == 0)[1].copy()).cuda())
        o_aqs.append(O.aq_offsets(f, w, h, O.AQ1))
        g_aqs.append(gl.aq_offsets([f], O.AQ1))
        assert g_aqs[-1].cpu().numpy()[0].tobytes() == o_aqs[-1].tobytes(), f"aq offsets of picture {i}"
        assert np.array_equal(g_infos[-1].cpu().numpy()[0], o_infos[-1]), f"block records of picture {i}"
    for first in range(n):                                  # every suffix, as the queue drains at the end of a stream
        want = O.mbtree(bw, bh, o_infos[first:], o_aqs[first:] if with_aq else None, strength)
        got = gl.mbtree(g_infos[first:], g_aqs[first:] if with_aq else None, strength)[0]
        assert got.dtype == want.dtype == np.float32 and got.tobytes() == want.tobytes(), f"suffix {first}: {np.nonzero(got != want)[0][:5]}"
    # the tree only ever lowers quantisers (more bits where later pictures keep looking), and a lone picture gets none of it
    alone = O.mbtree(bw, bh, o_infos[-1:], o_aqs[-1:], strength)
    assert np.array_equal(alone, o_aqs[-1])
    assert (O.mbtree(bw, bh, o_infos[1:], o_aqs[1:], strength) <= o_aqs[1]).all()


# ---- the lookahead in x264's structure: slicetype_frame_cost(p0, p1, b) for any triple (x264gpu_slicetype_* vs oracle/slicetype.c) ----
def run_slicetype(w, h, n, seed, triples, streams=1, row_mode=-1, **kw):
    from gpu_enc import GpuSlicetype
    frames = synth_frames(w, h, n, seed=seed)
    og, gg = O.OracleSlicetype(w, h, **kw), GpuSlicetype(w, h, streams=streams, **kw)
    if row_mode >= 0:
        from x264vfw_amd import lib as gpu_lib
        gpu_lib.check(gpu_lib.x264gpu_slicetype_set_row_mode(gg.h, row_mode), "set_row_mode")
    for i, f in enumerate(frames):
        og.put(i, f)
        gg.put(i, [f] * streams)
    bw = (w + 15) // 16
    for (p0, p1, b) in triples:
        d0, d1 = b - p0, p1 - b
        oc, gc = og.cost(p0, p1, b, d0, d1), gg.cost(p0, p1, b, d0, d1)
        what = f"{w}x{h} cost(p0={p0}, p1={p1}, b={b})"
        for l, d in ((0, d0), (1, d1)):
            if d > 0:
                om, gm = og.mvs(b, l, d), gg.mvs(b, l, d)
                for s in range(streams):
                    bad = np.nonzero((gm[s] != om).any(1))[0]
                    assert bad.size == 0, f"{what}: list {l} vectors differ in {bad.size} blocks; first {bad[0]} (x={bad[0] % bw}, y={bad[0] // bw}) gpu={gm[s][bad[0]]} cpu={om[bad[0]]}"
                    assert np.array_equal(gg.mv_costs(b, l, d)[s], og.mv_costs(b, l, d)), f"{what}: list {l} costs differ"
        for s in range(streams):
            assert np.array_equal(gg.intra_costs(b)[s], og.intra_costs(b)), f"{what}: intra costs differ"
            ol, gl = og.lowres_costs(b, d0, d1), gg.lowres_costs(b, d0, d1)[s]
            bad = np.nonzero(ol != gl)[0]
            assert bad.size == 0, f"{what}: lowres_costs differ in {bad.size} blocks; first {bad[0]} gpu={gl[bad[0]]:#x} cpu={ol[bad[0]]:#x}"
            assert gc[s] == oc, f"{what}: score gpu={gc[s]} cpu={oc}"
        if d1 == 0 and d0 > 0:
            assert gg.intra_mbs(b, d0) == og.intra_mbs(b, d0), f"{what}: intra block counts differ"
    og.close(); gg.close()


# what x264_slicetype_analyse asks for around one mini-GOP: scenecut (P against the previous picture), b-adapt 1's four costs, longer runs,
# and the costs x264_rc_analyse_slice reads once the types are known; repeated triples come from the memo
ADAPT = [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 2, 2), (0, 3, 3), (0, 3, 1), (0, 3, 2), (0, 4, 4), (0, 4, 2), (0, 4, 1), (0, 4, 3), (2, 4, 3), (0, 2, 1), (4, 4, 4), (4, 5, 5)]


@pytest.mark.parametrize("w,h,n,seed,triples,kw", [
    (176, 144, 6, 3, ADAPT, {}),                                    # no mbtree: the picture's edge blocks are not costed
    (176, 144, 6, 3, ADAPT, dict(do_edges=1)),                      # mbtree / VBV sessions: every block
    (352, 288, 6, 5, ADAPT, dict(do_edges=1, weightb=0)),
    (208, 120, 5, 7, ADAPT[:12], dict(me_range=8)),
    (80, 48, 5, 9, ADAPT[:12], dict(do_edges=1)),
    (32, 32, 4, 1, [(0, 1, 1), (0, 2, 2), (0, 2, 1), (0, 3, 3), (0, 3, 2)], {}),       # 2 x 2 blocks: edges always
    (352, 288, 5, 2, ADAPT[:12], dict(subme=1, me_method=0)),        # subme <= 1: dia, SAD, level 2, half-pel bidirectional vectors
    (720, 304, 4, 4, ADAPT[:7], dict(do_edges=1)),
])
def test_slicetype_costs_bitexact(gpu, w, h, n, seed, triples, kw):
    run_slicetype(w, h, n, seed, triples, **kw)


def test_slicetype_costs_headline_size(gpu):
    """1920x1080: 120 x 68 blocks, the row wavefront at its full depth"""
    run_slicetype(1920, 1080, 4, 8, [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 2, 2), (0, 3, 3), (0, 3, 1), (3, 3, 3)], do_edges=1)


def test_slicetype_costs_multistream(gpu):
    run_slicetype(176, 144, 5, 6, ADAPT[:10], streams=3, do_edges=1)


@pytest.mark.parametrize("w,h,kw", [(176, 144, dict(do_edges=1)), (352, 288, {}), (1920, 1080, dict(do_edges=1))])
def test_slicetype_costs_one_wavefront_per_stream(gpu, w, h, kw):
    """the other geometry of x264gpu_slicetype_frame_cost (one wavefront walks every block row of its stream: x264gpu_slicetype_set_row_mode 1) gives the
    same vectors, costs and scores as the row pipeline and the CPU checker"""
    run_slicetype(w, h, 5, 11, ADAPT[:12] if w < 1000 else ADAPT[:6], streams=2, row_mode=1, **kw)


# ---- macroblock-tree through B pictures: x264's macroblock_tree walked over both implementations (tests/mbtree_walk.py) ----
@pytest.mark.parametrize("w,h,types,pyramid,b_intra,aq,seed", [
    (176, 144, "PBBBPBBPBP", True, False, True, 3),             # b-pyramid, runs of 3 / 2 / 1
    (176, 144, "PBBBPBBPBP", False, False, True, 3),
    (208, 120, "IPPBPBBBP", True, True, True, 5),               # the keyframe's own pass (b_intra): the tree reaches frames[0]
    (352, 288, "PBBBPBBBPP", True, False, False, 7),            # without AQ
    (80, 48, "PBPBBP", True, False, True, 9),
])
def test_mbtree_through_b_pictures_bitexact(gpu, w, h, types, pyramid, b_intra, aq, seed):
    from gpu_enc import GpuSlicetype
    from mbtree_walk import macroblock_tree
    n = len(types)
    frames = synth_frames(w, h, n, seed=seed)
    og, gg = O.OracleSlicetype(w, h, slots=n + 1, do_edges=1), GpuSlicetype(w, h, slots=n + 1, do_edges=1)
    for i, f in enumerate(frames):
        og.put(i, f); gg.put(i, [f])
        if aq:
            a = O.aq_offsets(f, w, h)
            og.set_aq(i, a); gg.set_aq(i, a)
    slots = list(range(n))
    oo = macroblock_tree(og, slots, types, n - 1, b_intra, pyramid, O.TREE)
    go = macroblock_tree(gg, slots, types, n - 1, b_intra, pyramid, O.TREE)
    assert sorted(oo) == sorted(go) and len(oo) >= 1
    for i in range(n):
        assert np.array_equal(gg.propagate_cost(i)[0], og.propagate_cost(i)), f"propagate cost of picture {i} differs"
    for k in oo:
        assert go[k][0].tobytes() == oo[k].tobytes(), f"offsets of picture {k} differ"
        assert (oo[k] != (O.aq_offsets(frames[k], w, h) if aq else 0)).any(), "the tree moved nothing"
    # macroblock_tree_finish's weightdelta (a fade the lookahead's weight analysis explained: 1 - weighted / unweighted cost) enters the log2 ratio
    k = sorted(oo)[0]
    wd_o, wd_g = og.finish(slots[k], O.TREE, 0.125), gg.finish(slots[k], O.TREE, 0.125)[0]
    assert wd_g.tobytes() == wd_o.tobytes() and (wd_o < oo[k]).any() and not (wd_o > oo[k]).any()
    og.close(); gg.close()


@pytest.mark.parametrize("w,h,seed,edges", [(176, 144, 3, 0), (352, 288, 5, 1), (48, 32, 7, 0)])
def test_aq_weighted_frame_costs_bitexact(gpu, w, h, seed, edges):
    """fenc->i_cost_est_aq: the I / P / B costs of a triple weighted block by block with the inverse quantiser scale of the picture's AQ offsets
    (x264 slicetype_mb_cost) — the complexity x264_rc_analyse_slice uses in AQ sessions without macroblock-tree; with no offsets set it is the plain
    sum of the (capped) block costs"""
    from gpu_enc import GpuSlicetype
    n = 4
    frames = synth_frames(w, h, n, seed=seed)
    og, gg = O.OracleSlicetype(w, h, slots=n + 1, do_edges=edges), GpuSlicetype(w, h, slots=n + 1, do_edges=edges)
    for i, f in enumerate(frames):
        og.put(i, f); gg.put(i, [f])
    for (p0, p1, b) in [(0, 0, 0), (0, 1, 1), (0, 3, 3), (0, 3, 1), (1, 3, 2)]:
        assert gg.cost(p0, p1, b, b - p0, p1 - b)[0] == og.cost(p0, p1, b, b - p0, p1 - b)
        plain = og.cost_aq(b, b - p0, p1 - b)
        assert gg.cost_aq(b, b - p0, p1 - b) == [plain]
    for i, f in enumerate(frames):
        a = O.aq_offsets(f, w, h)
        og.set_aq(i, a); gg.set_aq(i, a)
    moved = 0
    for (p0, p1, b) in [(0, 0, 0), (0, 1, 1), (0, 3, 3), (0, 3, 1), (1, 3, 2)]:
        c = og.cost_aq(b, b - p0, p1 - b)
        assert gg.cost_aq(b, b - p0, p1 - b) == [c] and c > 0
        moved += c != og.cost_est(b, b - p0, p1 - b)
    assert moved >= 3
    og.close(); gg.close()


def fade_frames(w, h, n, seed, step=6):
    """a clip fading to black: picture i = picture of a moving scene scaled by (1 - i * step / 100), the chroma pulled towards 128"""
    fr = synth_frames(w, h, n, seed=seed)
    out = []
    for i, f in enumerate(fr):
        a = max(0.0, 1.0 - i * step / 100.0)
        g = f.astype(np.float32)
        g[:w * h] *= a
        g[w * h:] = 128 + (g[w * h:] - 128) * a
        out.append(np.clip(np.rint(g), 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("w,h,seed,kw", [(176, 144, 3, {}), (352, 288, 5, dict(do_edges=1)), (208, 120, 7, dict(subme=1, me_method=0))])
def test_weight_analysis_primitives_bitexact(gpu, w, h, seed, kw):
    """x264_weights_analyse's building blocks on a fade: luma statistics, the cost of predicting a picture from its predecessor under several explicit
    weights (reference in place, and motion-compensated by the lookahead's vectors once that search has run), and the P cost searched on a weighted reference"""
    from gpu_enc import GpuSlicetype
    n = 4
    frames = fade_frames(w, h, n, seed)
    og, gg = O.OracleSlicetype(w, h, **kw), GpuSlicetype(w, h, **kw)
    for i, f in enumerate(frames):
        og.put(i, f); gg.put(i, [f])
        assert np.array_equal(gg.pixel_stats(i)[0], og.pixel_stats(i, f)), f"statistics of picture {i}"
        assert np.array_equal(gg.chroma_stats(i)[0], og.chroma_stats(i, f)), f"chroma statistics of picture {i}"
    for i in range(1, n):
        assert gg.cost(i, i, i, 0, 0)[0] == og.cost(i, i, i, 0, 0)
    weights = [None, (60, 6, 0), (59, 6, 1), (117, 7, -2), (1, 0, -3), (127, 7, 0)]
    for wgt in weights:
        assert gg.weight_cost(2, 1, 1, wgt)[0] == og.weight_cost(2, 1, 1, wgt), f"weight {wgt}, reference in place"
    # the P cost of picture 2 from picture 1, searched on the weighted reference (lookahead mode), then the costs on the compensated reference
    assert gg.cost(1, 2, 2, 1, 0, weight=(60, 6, 0))[0] == og.cost(1, 2, 2, 1, 0, weight=(60, 6, 0))
    assert np.array_equal(gg.mvs(2, 0, 1)[0], og.mvs(2, 0, 1))
    for wgt in weights:
        assert gg.weight_cost(2, 1, 1, wgt)[0] == og.weight_cost(2, 1, 1, wgt), f"weight {wgt}, compensated reference"
    for wgt in weights:                      # (after the search: the chroma planes compensated by the half-resolution vectors)
        for plane in (1, 2):
            assert gg.weight_cost_chroma(2, 1, 1, plane, wgt)[0] == og.weight_cost_chroma(2, frames[2], frames[1], 1, plane, wgt), f"chroma plane {plane}, weight {wgt}"
            assert gg.weight_cost_chroma(3, 1, 2, plane, wgt)[0] == og.weight_cost_chroma(3, frames[3], frames[1], 2, plane, wgt), f"chroma plane {plane}, weight {wgt}, reference in place"
    assert og.weight_cost(2, 1, 1, (60, 6, 0)) < og.weight_cost(2, 1, 1, None), "on a fade the right weight must pay"
    og.close(); gg.close()


@pytest.mark.parametrize("w,h,mode,strength", [(176, 144, 2, 1.0), (176, 144, 3, 1.0), (352, 288, 2, 0.8), (208, 120, 3, 1.3), (1920, 1080, 3, 1.0)])
def test_aq_mode_2_and_3_bitexact(gpu, w, h, mode, strength):
    """--aq-mode 2 (auto-variance) / 3 (auto-variance with a bias to dark scenes): x264_adaptive_quant_frame's float path — (energy + 1)^(1/8) per
    macroblock, the picture's mean and mean square summed in raster order, the offsets — on the device equal to the CPU checker's, two streams"""
    from gpu_enc import GpuLookahead
    frames = synth_frames(w, h, 2, seed=w + mode)
    gl = GpuLookahead(w, h, streams=2)
    g = gl.aq_offsets_mode(frames, mode, strength).cpu().numpy()
    for s in range(2):
        o = O.aq_offsets_mode(frames[s], w, h, mode, strength)
        assert g[s].tobytes() == o.tobytes(), f"stream {s}: {np.nonzero(g[s] != o)[0][:5]}"
    assert np.abs(g).max() > 0.25                       # (offsets of a quarter of a quantiser step and more: the mode does something)
    gl.close()
