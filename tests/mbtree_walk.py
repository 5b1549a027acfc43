"""x264's macroblock_tree ([x264-upstream] encoder/slicetype.c) as a walk over a slicetype object (oracle_lib.OracleSlicetype or
gpu_enc.GpuSlicetype: cost / clear_propagate / propagate / finish): frames[0] = the last non-B picture, frames[1..num_frames] the
queued ones with their decided types ('I', 'P', 'B').  The product's host code (host/encoder.cpp st_macroblock_tree) is the same walk."""


def macroblock_tree(st, slots, types, num_frames, b_intra, pyramid, strength):
    """-> {frame index: offsets} of the pictures x264 finishes: the next non-B picture (index 1.. or 0 for b_intra) and its B-reference"""
    idx = 0 if b_intra else 1
    isb = lambda i: types[i] == 'B'
    cost = lambda p0, p1, b: st.cost(slots[p0], slots[p1], slots[b], b - p0, p1 - b)
    prop = lambda p0, p1, b, ref: st.propagate(slots[p0], slots[p1], slots[b], b - p0, p1 - b, ref)
    if b_intra:
        cost(0, 0, 0)
    i = num_frames
    while i > 0 and isb(i):
        i -= 1
    last_nonb = i
    if last_nonb < idx:
        return {}
    st.clear_propagate(slots[last_nonb])
    bframes = 0
    while i > idx:
        i -= 1
        cur_nonb = i
        while isb(cur_nonb) and cur_nonb > 0:
            cur_nonb -= 1
        if cur_nonb < idx:
            break
        cost(cur_nonb, last_nonb, last_nonb)
        st.clear_propagate(slots[cur_nonb])
        bframes = last_nonb - cur_nonb - 1
        if pyramid and bframes > 1:
            middle = (bframes + 1) // 2 + cur_nonb
            cost(cur_nonb, last_nonb, middle)
            st.clear_propagate(slots[middle])
            while i > cur_nonb:
                p0 = middle if i > middle else cur_nonb
                p1 = middle if i < middle else last_nonb
                if i != middle:
                    cost(p0, p1, i)
                    prop(p0, p1, i, 0)
                i -= 1
            prop(cur_nonb, last_nonb, middle, 1)
        else:
            while i > cur_nonb:
                cost(cur_nonb, last_nonb, i)
                prop(cur_nonb, last_nonb, i, 0)
                i -= 1
        prop(cur_nonb, last_nonb, last_nonb, 1)
        last_nonb = cur_nonb
    # (x264 reads the intra costs the slice-type analysis left with these pictures; asking for the I cost is a no-op then)
    cost(last_nonb, last_nonb, last_nonb)
    out = {last_nonb: st.finish(slots[last_nonb], strength)}
    if pyramid and bframes > 1:
        m = last_nonb + (bframes + 1) // 2
        cost(m, m, m)
        out[m] = st.finish(slots[m], strength)
    return out
