// cabac_layout.hip.h — where the device keeps the CABAC context variables of a slice (cabac_rd.hip.h codes with them, encoder.hip unpacks them
// for the tests).
#pragma once

namespace x264gpu {

// Where context variable ctx lives: register (0 = a: header syntax, coded_block_flags and the transform-size flag, four to a dword;
// 1 = r: the residual contexts of block categories 0..4, one BYTE PER CATEGORY in every lane; 2 = r8: those of 8x8 blocks), lane, bit shift.
// In r and r8 the lane is the context's ROLE: lane p = significant_coeff_flag of scan position p, lane 16 + p = last_significant_coeff_flag
// of position p, lane 32 + q = coeff_abs_level_minus1 context q (chroma DC, which shares its byte with luma DC, sits in lanes 48..63
// instead; 8x8 blocks: lane = ctxIdxInc).  All significance / last flags of a 4x4 block then code in ONE step, every lane its own context,
// and the level bins run as a loop in which only the lanes of the two contexts concerned act.
__host__ __device__ inline bool cab_locate(int ctx, int &reg, int &lane, int &sh)
{
    const int sig_off[6] = { 105, 120, 134, 149, 152, 402 }, last_off[6] = { 166, 181, 195, 210, 213, 417 }, abs_off[6] = { 227, 237, 247, 257, 266, 426 };
    const int n_sig[6] = { 15, 14, 15, 3, 14, 15 }, n_last[6] = { 15, 14, 15, 3, 14, 9 }, n_abs[6] = { 10, 10, 10, 9, 10, 10 }, byte_of[6] = { 3, 1, 0, 3, 2, 0 };
    if (ctx < 105) { reg = 0; lane = ctx >> 2; sh = (ctx & 3) * 8; return true; }
    if (ctx >= 399 && ctx <= 401) { const int c = ctx - 399 + 105; reg = 0; lane = c >> 2; sh = (c & 3) * 8; return true; }
    for (int cat = 0; cat < 6; cat++) {
        int role = -1, i = 0;
        if (ctx >= sig_off[cat] && ctx < sig_off[cat] + n_sig[cat]) { role = 0; i = ctx - sig_off[cat]; }
        else if (ctx >= last_off[cat] && ctx < last_off[cat] + n_last[cat]) { role = 1; i = ctx - last_off[cat]; }
        else if (ctx >= abs_off[cat] && ctx < abs_off[cat] + n_abs[cat]) { role = 2; i = ctx - abs_off[cat]; }
        if (role < 0) continue;
        reg = cat == 5 ? 2 : 1; sh = 8 * byte_of[cat];
        lane = cat == 3 ? (role == 0 ? 48 + i : role == 1 ? 52 + i : 55 + i) : role * 16 + i;
        return true;
    }
    return false;
}

}  // namespace x264gpu
