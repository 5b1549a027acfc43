"""CPU: the serial restatement of the residual coding (tests/cabac_levels_ref.py) that the device's one-walk-per-macroblock pricing is checked against through the
primitive x264gpu_cabac_level_walk: its state machine is 9.3.4.2's, its node walk x264's coeff_abs_level1_ctx / coeff_abs_levelgt1_ctx / coeff_abs_level_transition,
its significance map 7.3.5.3.3's — known-answer cases typed by hand here.  (It is a second restatement beside oracle/cabac_rd.cpp, with its own control flow; the
pipeline tests compare the device with that one, the primitive test with this one.)"""
import random

import cabac_levels_ref as R


def test_step_is_the_h264_state_machine():
    # 9.3.4.2: an MPS moves pStateIdx up (62 at most), an LPS follows transIdxLps and flips valMPS at state 0; costs come from the generated table
    assert R.step((0 << 1) | 0, 1) == ((0 << 1) | 1, R._ENT[1])
    assert R.step((62 << 1) | 1, 1)[0] == (62 << 1) | 1
    assert R.step((10 << 1) | 1, 0)[0] == (R.TRANS_LPS[10] << 1) | 1
    assert R.step((10 << 1) | 0, 0) == ((11 << 1) | 0, R._ENT[20])


def test_node_walk_of_a_block():
    # one coefficient of 1: ctx 1 gets a 0; then a 3: ctx 2 gets a 1 and ctx 5 gets 1, 0; then a 1: ctx 0 (node 4) gets a 0
    ctx = [0] * 10
    bits = R.block_levels([1, 3, 1], ctx, 0)
    want = [0] * 10
    want[1], c1 = R.step(0, 0)
    want[2], c2 = R.step(0, 1)
    s5, c3 = R.step(0, 1); want[5], c4 = R.step(s5, 0)
    want[0], c5 = R.step(0, 0)
    assert ctx == want and bits == c1 + c2 + c3 + c4 + c5 + 3 * 256


def test_random_cases_are_reproducible_and_cover_every_category():
    cases = R.random_cases(200, 7)
    again = R.random_cases(200, 7)
    assert [c[6] for c in cases] == [c[6] for c in again]
    kinds = {c[1]["cat0"] for c in cases}
    assert kinds == {2, 5, 1, -1}
    assert any(c[1]["ldc"] for c in cases) and any(c[1]["nzdc"] == 3 for c in cases) and any(c[1]["nzac"] for c in cases)
    assert any(max(abs(x) for x in c[0]) >= 15 for c in cases)          # escapes


def test_significance_map_of_a_block():
    # coefficients at positions 1 and 3 of a 16-coefficient block: sig 0 = 0, sig 1 = 1 / last 1 = 0, sig 2 = 0, sig 3 = 1 / last 3 = 1; nothing after
    sig, last = [0] * 15, [0] * 15
    bits = R.block_sigmap([0, 5, 0, -1] + [0] * 12, sig, last, 0)
    e = lambda b: R.step(0, b)
    assert sig[:4] == [e(0)[0], e(1)[0], e(0)[0], e(1)[0]] and sig[4:] == [0] * 11
    assert last[1] == e(0)[0] and last[3] == e(1)[0] and last[0] == last[2] == 0
    assert bits == 2 * e(0)[1] + 2 * e(1)[1] + e(0)[1] + e(1)[1]
    # a coefficient in the block's final position has neither flag
    sig, last = [0] * 3, [0] * 3
    assert R.block_sigmap([0, 0, 0, 7], sig, last, 0) == 3 * e(0)[1] and last == [0, 0, 0]
