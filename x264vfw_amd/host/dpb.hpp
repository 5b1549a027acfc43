// dpb.hpp — the decoded picture buffer as x264 runs it ([x264-upstream] encoder/encoder.c x264_reference_build_list, x264_reference_check_reorder,
// x264_reference_hierarchy_reset, x264_reference_update, behind x264_encoder_encode, reference call site codec.c:1693): which pictures a slice may
// reference, in which order, and what the slice header must say so that a decoder builds the same lists (frame_num, POC, ref_pic_list_modification,
// memory_management_control_operation).  Pictures live in the device encoder's DPB slots (x264gpu_pic); this class hands out the slots.
//   * list 0 = kept pictures before the picture in display order, list 1 (B) = after it, each NEAREST IN DISPLAY ORDER FIRST, cut to --ref / to
//     min(--ref, num_reorder_frames);
//   * the default initial order of 8.2.4.2 is by picture number (P) / POC (B): where x264's order differs, the whole list is sent as modifications;
//   * sliding window of sps num_ref_frames pictures; under --b-pyramid a B-reference whose mini-GOP holds a b displayed later than the reorder depth
//     allows removes pictures from the END of list 0 by MMCO (x264: i_mmco_remove_from_end) so the delayed b's find room;
//   * --weightp 2: every P picture with at least two references gets a DUPLICATE of reference 0 at index 1 carrying the luma weight
//     {scale 1, denom 0, offset -1} (x264 weighted_reference_duplicate, h->mb.ref_blind_dupe): the list grows by one, the whole list is sent as
//     modifications (the duplicate is a picture-number difference of 0, coded as abs_diff_pic_num_minus1 = MaxFrameNum - 1);
//   * --weightp 1 / 2 with a luma weight from x264_weights_analyse (fades): reference 0 carries it; under --weightp 2 an UNWEIGHTED duplicate
//     and (offset > -128) a duplicate with the same scale and offset - 1 follow it: [w ref0, w(offset - 1) dupe, unweighted dupe, ref1, ..].
#pragma once
#include "host.hpp"
#include <stdlib.h>
#include <algorithm>
#include <vector>

namespace x264host {

enum { PIC_IDR = 0, PIC_I = 1, PIC_P = 2, PIC_BREF = 3, PIC_B = 4 };

struct DpbPlan {
    x264gpu_pic pic;             // slice type, POC, destination slot, the reference lists as slots (qp left to the caller)
    int type;                    // PIC_*
    int frame;                   // display index
    int frame_num;               // as sent (not yet reduced modulo MaxFrameNum)
    int nal_ref_idc, idr;
    int num_ref[2];
    SliceParams::Reorder reorder[2];
    int n_mmco, mmco_diff[16], mmco_poc[16];
    int list_poc[2][X264GPU_MAX_LIST];       // POC of every list entry (diagnostics / tests)
};

class Dpb {
public:
    struct Ref { int slot, frame, frame_num, poc, type; };
    int max_dpb = 1, max_ref0 = 1, max_ref1 = 0, pyramid = 0, num_reorder = 0, log2_max_frame_num = 4, weightp = 0;
    std::vector<Ref> refs;       // h->frames.reference: kept pictures in coding order
    int frame_num = 0, last_idr = 0;
    DpbPlan last;

    // sps->i_num_ref_frames, vui.i_num_reorder_frames and the list limits of x264 for --ref / --bframes / --b-pyramid
    void configure(int frame_reference, int bframes, int b_pyramid, int log2_max_fn, int weighted_pred = 0)
    {
        weightp = weighted_pred;
        pyramid = bframes > 1 ? b_pyramid : 0;
        num_reorder = pyramid ? 2 : bframes ? 1 : 0;
        max_dpb = std::max(std::max(frame_reference, 1 + num_reorder), pyramid ? 4 : 1);
        if (max_dpb > 7) max_dpb = 7;
        max_ref0 = frame_reference;
        max_ref1 = std::min(num_reorder, frame_reference);
        log2_max_frame_num = log2_max_fn;
        refs.clear(); frame_num = 0; last_idr = 0;
    }
    // extra_slots / avoid: a session with several pictures in flight (host/encoder.cpp: Inflight) owns more slots than the DPB holds pictures and keeps the
    // slots its pictures in flight write or read out of the choice of a destination
    int extra_slots = 0; unsigned avoid = 0;
    int slots() const { return max_dpb + 1 + extra_slots; }
    bool has_free_slot() const { for (int s = 0; s < slots(); s++) { bool used = (avoid >> s) & 1; for (const Ref &r : refs) used |= r.slot == s; if (!used) return true; } return false; }

    // Plans the picture `frame` (display index) of type `type`.  follow_coded / follow_frame: coding index and display index of the disposable
    // pictures that follow it immediately in coding order (x264 reads them from h->frames.current); coded = this picture's coding index.
    // x264_weights_analyse's result for reference 0: the luma weight and, once luma has one, the chroma planes' (Cb, Cr; one denominator)
    struct LumaWeight { int on = 0, scale = 1, denom = 0, offset = 0; int con[2] = { 0, 0 }, cscale[2] = { 1, 1 }, cdenom = 0, coffset[2] = { 0, 0 }; };
    const DpbPlan &plan(int type, int frame, int n_follow = 0, const int *follow_coded = nullptr, const int *follow_frame = nullptr, const LumaWeight *w0 = nullptr)
    {
        DpbPlan &p = last;
        p = DpbPlan();
        p.type = type; p.frame = frame;
        if (type == PIC_IDR) { frame_num = 0; last_idr = frame; refs.clear(); }
        p.frame_num = frame_num;
        p.idr = type == PIC_IDR;
        p.nal_ref_idc = type == PIC_IDR ? 3 : type == PIC_B ? 0 : 2;
        p.pic.slice_type = type == PIC_IDR ? X264GPU_SLICE_I : type == PIC_I ? X264GPU_SLICE_I_NONIDR : type == PIC_P ? X264GPU_SLICE_P : X264GPU_SLICE_B;
        p.pic.poc = 2 * (frame - last_idr);
        p.pic.keep = p.nal_ref_idc != 0;
        // a free slot: any not held by a kept picture
        for (int s = 0; s < slots(); s++) { bool used = (avoid >> s) & 1; for (const Ref &r : refs) used |= r.slot == s; if (!used) { p.pic.dst = s; break; } }
        // x264_reference_hierarchy_reset (I, P, BREF): room for the delayed b's of the mini-GOP
        int remove_from_end = 0;
        if (type == PIC_I || type == PIC_P || type == PIC_BREF) {
            bool has_delay = false;
            for (int i = 0; i < n_follow; i++) has_delay |= follow_coded[i] != follow_frame[i] + num_reorder;
            if (has_delay && pyramid) remove_from_end = std::max((int)refs.size() + 2 - max_dpb, 0);
        }
        if (type == PIC_IDR || type == PIC_I) return p;
        // ---- x264_reference_build_list ----
        std::vector<Ref> l[2];
        for (const Ref &r : refs) { if (r.poc < p.pic.poc) l[0].push_back(r); else if (r.poc > p.pic.poc) l[1].push_back(r); }
        if (remove_from_end) {
            std::stable_sort(l[0].begin(), l[0].end(), [](const Ref &a, const Ref &b) { return a.frame > b.frame; });
            for (int i = (int)l[0].size() - 1; i >= (int)l[0].size() - remove_from_end && i >= 0; i--) {
                p.mmco_poc[p.n_mmco] = l[0][(size_t)i].poc; p.mmco_diff[p.n_mmco++] = frame_num - l[0][(size_t)i].frame_num;
            }
        }
        for (int k = 0; k < 2; k++)      // nearest in display order first (x264 bubble-sorts by |frame distance|; the sort is stable there too)
            std::stable_sort(l[k].begin(), l[k].end(), [&](const Ref &a, const Ref &b) { return abs(frame - a.frame) < abs(frame - b.frame); });
        // x264_reference_check_reorder: is this the order a decoder would initialise?
        bool reorder[2] = { false, false };
        for (int k = 0; k <= (type == PIC_P ? 0 : 1); k++)
            for (size_t i = 0; i + 1 < l[k].size(); i++) {
                const int fdiff = l[k][i + 1].frame_num - l[k][i].frame_num, pdiff = l[k][i + 1].poc - l[k][i].poc;
                if (type == PIC_P ? fdiff > 0 : k == 1 ? pdiff < 0 : pdiff > 0) { reorder[k] = true; break; }
            }
        if ((int)l[1].size() > max_ref1) l[1].resize((size_t)max_ref1);
        if ((int)l[0].size() > max_ref0) l[0].resize((size_t)max_ref0);
        if (type == PIC_P) l[1].clear();
        p.pic.blind_dupe = -1;
        if (type == PIC_P && weightp && w0 && w0->on && !l[0].empty()) {
            // x264_reference_build_list with fenc->weight[0][0] set: a pure offset is sent with denominator 0
            LumaWeight w = *w0;
            if (w.scale == 1 << w.denom) { w.scale = 1; w.denom = 0; }
            p.pic.wl0[0].on = 1; p.pic.wl0[0].denom = (int8_t)w.denom; p.pic.wl0[0].scale = (int16_t)w.scale; p.pic.wl0[0].offset = (int16_t)w.offset;
            // the chroma planes' weights ride on index 0 alone (x264's duplicates carry w[1].weightfn = w[2].weightfn = NULL); weighted_pred_init drops a
            // weight that changes nothing
            for (int c = 0; c < 2; c++)
                if (w.con[c] && !(w.cscale[c] == 1 << w.cdenom && w.coffset[c] == 0)) {
                    p.pic.wc0[0].on[c] = 1; p.pic.wc0[0].denom = (int8_t)w.cdenom; p.pic.wc0[0].scale[c] = (int16_t)w.cscale[c]; p.pic.wc0[0].offset[c] = (int16_t)w.coffset[c];
                }
            if (weightp == 2 && l[0].size() > 1) {
                l[0].insert(l[0].begin() + 1, l[0][0]);             // the unweighted duplicate
                reorder[0] = true;
                if (w.offset > -128) {                              // ... and in front of it the duplicate one offset step down
                    l[0].insert(l[0].begin() + 1, l[0][0]);
                    p.pic.blind_dupe = 1;
                    p.pic.wl0[1] = p.pic.wl0[0]; p.pic.wl0[1].offset = (int16_t)(w.offset - 1);
                }
            }
        } else if (type == PIC_P && weightp == 2 && l[0].size() > 1) {       // the blind duplicate of reference 0
            l[0].insert(l[0].begin() + 1, l[0][0]);
            reorder[0] = true;
            p.pic.blind_dupe = 1;
            p.pic.wl0[1].on = 1; p.pic.wl0[1].denom = 0; p.pic.wl0[1].scale = 1; p.pic.wl0[1].offset = -1;
        }
        for (int k = 0; k < 2; k++) {
            p.num_ref[k] = p.pic.nref[k] = (int)l[k].size();
            for (size_t i = 0; i < l[k].size(); i++) { p.pic.slot[k][i] = (int8_t)l[k][i].slot; p.list_poc[k][i] = l[k][i].poc; }
            if (!reorder[k]) continue;
            int pred = frame_num;
            p.reorder[k].n = (int)l[k].size();
            for (size_t i = 0; i < l[k].size(); i++) {
                const int diff = l[k][i].frame_num - pred;
                p.reorder[k].cmd[i].idc = diff > 0;
                p.reorder[k].cmd[i].arg = (abs(diff) - 1) & ((1 << log2_max_frame_num) - 1);
                pred = l[k][i].frame_num;
            }
        }
        return p;
    }

    // x264_reference_update after the planned picture has been coded
    void commit()
    {
        const DpbPlan &p = last;
        if (!p.nal_ref_idc) return;
        for (int i = 0; i < p.n_mmco; i++)
            for (size_t j = 0; j < refs.size(); j++) if (refs[j].poc == p.mmco_poc[i]) { refs.erase(refs.begin() + (long)j); break; }
        refs.push_back(Ref{ p.pic.dst, p.frame, p.frame_num, p.pic.poc, p.type });
        if ((int)refs.size() > max_dpb) refs.erase(refs.begin());
        frame_num++;
    }

    // --direct temporal / auto: the mode of the planned B picture (the caller's decision; plan() leaves spatial)
    void set_direct(int temporal, int auto_write) { last.pic.direct_temporal = temporal; last.pic.direct_auto = auto_write; }
    // the slice header's share of a plan
    void fill(SliceParams &sp) const
    {
        const DpbPlan &p = last;
        sp.slice_type = p.pic.slice_type == X264GPU_SLICE_I_NONIDR ? X264GPU_SLICE_I : p.pic.slice_type;
        sp.direct_spatial = !p.pic.direct_temporal;
        sp.frame_num = p.frame_num; sp.idr = p.idr; sp.nal_ref_idc = p.nal_ref_idc; sp.poc = p.pic.poc;
        sp.num_ref = p.num_ref[0]; sp.num_ref1 = p.num_ref[1];
        sp.reorder[0] = p.reorder[0]; sp.reorder[1] = p.reorder[1];
        sp.n_mmco = p.n_mmco;
        for (int i = 0; i < p.n_mmco; i++) sp.mmco_diff[i] = p.mmco_diff[i];
        sp.weighted_pred = weightp > 0;
        for (int i = 0; i < X264GPU_MAX_LIST; i++) { sp.wl0[i].on = p.pic.wl0[i].on; sp.wl0[i].denom = p.pic.wl0[i].denom; sp.wl0[i].scale = p.pic.wl0[i].scale; sp.wl0[i].offset = p.pic.wl0[i].offset; }
        for (int i = 0; i < X264GPU_MAX_LIST; i++)
            for (int c = 0; c < 2; c++) { sp.wc0[i].on[c] = p.pic.wc0[i].on[c]; sp.wc0[i].denom = p.pic.wc0[i].denom; sp.wc0[i].scale[c] = p.pic.wc0[i].scale[c]; sp.wc0[i].offset[c] = p.pic.wc0[i].offset[c]; }
    }
};

}  // namespace x264host
