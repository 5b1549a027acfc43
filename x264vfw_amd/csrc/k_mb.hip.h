// k_mb.hip.h — the macroblock loop of one slice in x264's own structure, raster-serial: x264_macroblock_analyse + x264_macroblock_encode
// per macroblock ([x264-upstream] encoder/analyse.c, me.c, macroblock.c behind x264_encoder_encode, reference call site codec.c:1693).
//
// Why raster-serial: every decision of a macroblock reads macroblocks BEFORE it in raster order of the same slice — motion vector
// predictors / search candidates / P_Skip vector from the coded neighbours, intra prediction from their reconstruction, the
// fast-intra heuristic from the running count of intra macroblocks (and, once RD is in, the live CABAC state) — so the only exact
// decomposition is: ONE wavefront per picture walks the macroblocks in order and spends its 64 lanes inside the macroblock
// (candidates x rows in the searches, modes x rows in the intra analysis, 4x4 blocks x rows in the transforms); the chip is filled by
// independent pictures (streams / closed GOPs) in flight.  Deblocking and the half-pel planes stay frame-level kernels behind it.
//
// Restates oracle/analyse.c bit-exactly (records, levels, reconstruction).
#pragma once
#include "deblock_line.hip.h"
#include "enc_common.hip.h"
#include "k_analyse.hip.h"
#include "intra8.hip.h"
#include "rd.hip.h"

namespace x264gpu {

constexpr int IT_STRIDE = 32;                 // luma tile stride (bytes)
constexpr int IT_ORG = IT_STRIDE + 4;         // offset of sample (0,0): row -1 and columns -4..-1 precede it
constexpr int IT_SIZE = 17 * IT_STRIDE + 8;
constexpr int MB_COST_MAX = 1 << 28;
enum { D_16x16 = 0, D_16x8 = 1, D_8x16 = 2, D_8x8 = 3 };
enum { ME_16 = 0, ME_8 = 1, ME_16x8 = 5, ME_8x16 = 7, ME_COUNT = 9 };     // slots of the per-macroblock search results

// Reference cache: one LDS slot per reference (slot = ref % 3) holding RC_ROWS x RC_COLS samples of ALL FOUR half-pel planes around the
// macroblock displaced by the vector it was last centred on (the predictor of the first search that needed the reference).  Start
// candidates, the full-pel search and the sub-pel refinement of every search of the macroblock on that reference read it when the samples
// they touch lie inside, and re-centre it (one trip to memory for all four planes) when they do not: most searches of a macroblock then
// cost no trip at all.  esa uses the same bytes as ONE +-17 full-pel window (WIN_ROWS x WIN_STRIDE); umh roams global memory for its
// full-pel steps and uses the slot for the sub-pel ones.
constexpr int MVC_N = 512;
constexpr int RC_ROWS = 30, RC_PD = 10, RC_COLS = 4 * RC_PD, RC_PLANE_DW = RC_ROWS * RC_PD, RC_SLOT_DW = 4 * RC_PLANE_DW, RC_MX = 12, RC_MY = 7;
// a slot's geometry as a type (the LDS layout names one: MbLds the macroblock loop's 30 x 40, the lookahead's StLds a wider, flatter one): PD dwords a row
template <int PD_, int ROWS_> struct RcGeo {
    static constexpr int PD = PD_, ROWS = ROWS_, COLS = 4 * PD_, PLANE_DW = ROWS_ * PD_, SLOT_DW = 4 * PLANE_DW, NT = (PLANE_DW + 63) / 64;      // NT: loads a lane issues per plane to fill a slot
    static_assert(PD_ == 10 || PD_ == 16, "row length: 10 dwords (division by multiplication below) or 16");
    static __device__ __forceinline__ int row_of(int i) { return PD_ == 16 ? i >> 4 : (i * 205) >> 11; }      // i / PD for i < 320 (PD 10)
};
// tags of the three reference-cache slots, as a lane-indexed register table like MeState (lane = slot): which picture the slot holds and where
// its sample (0, 0) lies.  (As nine named fields selected by `slot` the structure stayed in scratch memory: a conditional over lvalues is a
// select of addresses.)
struct WinTags { int tref, tx, ty; };

template <int M> struct MbLds {
    // where a (re-)centred reference-cache slot lies around the block: rc_mx columns to its left, rc_my rows above it (a 16x16 macroblock sits in the middle)
    static constexpr int rc_mx = RC_MX, rc_my = RC_MY;
    using rcg = RcGeo<RC_PD, RC_ROWS>;
    __attribute__((aligned(16))) uint32_t rc[3 * RC_SLOT_DW];        // >= WIN_ROWS * WIN_STRIDE bytes (esa)
    uint32_t csub[CSubGeo<M>::DWORDS];
    __attribute__((aligned(16))) uint8_t src[16 * 16];    // the source macroblock, row-major (the searches read rows of it)
    __attribute__((aligned(16))) uint8_t csrc[8 * 16];    // its NV12 chroma rows
    uint16_t mvcost[MVC_N];                    // mv-cost table of this macroblock's quantiser for |mv - mvp| < MVC_N quarter-samples (it is symmetric)
    __attribute__((aligned(8))) uint8_t tile[IT_SIZE];
    __attribute__((aligned(8))) uint8_t tile8[IT_SIZE];
    __attribute__((aligned(8))) int16_t lv8[256];
    __attribute__((aligned(8))) int16_t lv4[256];
    __attribute__((aligned(8))) uint8_t pred8tab[9 * 64];
    uint8_t nb[NB_SIZE];
    uint8_t cnb[2][CNB_SIZE];
    uint8_t U[U_SIZE];
    uint8_t U8[U8_SIZE];
    uint8_t modes4[16], modes8[16], nmodes[8];
    __attribute__((aligned(16))) x264gpu_mb rec;   // the record being built (lane 0 fills it, 16 lanes store it)
    // search state
    int16_t cand[16][2];                       // filtered start candidates of the search in progress
    int slw[3];                                // --slices N in P pictures (EncK.sl_stat): window [0], [1]) of harmless prior intra counts, [2] the assumed one
};

struct MbCtx {
    int s, lane, mbx, mby, mbi, px, py;
    int sy;                       // macroblock row within the slice: nothing above row 0 of the slice is available
    const uint8_t *fenc, *fuv;
    int qp, qpc, lambda, subme;
    bool satd, chroma_me;
    const uint16_t *cost_base;
    int mvmin0, mvmax0, mvmin1, mvmax1, smin0, smax0, smin1, smax1, fmin0, fmax0, fmin1, fmax1;
    int nref;
};

// Phase timers of the macroblock loop (builds with -DMB_PROF only: tools/mb_prof.py); otherwise empty
enum { PH_SETUP, PH_ME_PRED, PH_ME_WIN, PH_ME_FPEL, PH_ME_SUBSTAGE, PH_ME_SUBPEL, PH_ME_GLUE, PH_PSKIP, PH_INTRA_CHROMA, PH_INTRA, PH_ENC_INTER, PH_ENC_INTRA, PH_STORE, PH_COUNT };
#ifdef MB_PROF
struct Prof {
    unsigned long long t, acc[32], t2;          // 0..15: the phases / counts of the macroblock loop; 16..31: inside the CABAC pricing (cabac_rd.hip.h, -DMB_PROF_RD)
    __device__ __forceinline__ void start() { for (int i = 0; i < 32; i++) acc[i] = 0; t = __builtin_readcyclecounter(); t2 = t; }
    __device__ __forceinline__ void mark(int i) { const unsigned long long n = __builtin_readcyclecounter(); acc[i] += n - t; t = n; }
    __device__ __forceinline__ void count(int i) { acc[i]++; }
    __device__ __forceinline__ void count(int i, int n) { acc[i] += (unsigned long long)n; }
    // a second, nested clock (the phases' clock t keeps running): begin2 .. mark2
    __device__ __forceinline__ void begin2() { t2 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark2(int i) { const unsigned long long n = __builtin_readcyclecounter(); acc[i] += n - t2; t2 = n; }
};
#else
struct Prof {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void count(int) {}
    __device__ __forceinline__ void count(int, int) {}
    __device__ __forceinline__ void begin2() {}
    __device__ __forceinline__ void mark2(int) {}
};
#endif

// a wave-uniform value the compiler cannot prove uniform (it came through LDS or a vector load): say so, it then lives in an SGPR
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void uni_set(int &v) { v = __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void uni_set(unsigned &v) { v = (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void uni_set(bool &v) { v = __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ void uni_set(unsigned long long &v) { v = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32); }
template <class... T> __device__ __forceinline__ void uni_all(T &...v) { (uni_set(v), ...); }

// Per-macroblock search state held in registers, one table entry per lane (every value is wave-uniform): reading entry i is a
// v_readlane, writing it a select -- no LDS round trip.
struct MeState {
    int mvx, mvy, mvpx, mvpy, cost, costmv, ref, refcost;   // lane = slot of the search result (ME_COUNT)
    int cref, cmvx, cmvy;                                   // lane = motion cache grid index: (y + 1) * 4 + x + 1, x = -1..2, y = -1..1
    int mvcx, mvcy;                                         // lane = ref * 5 + q: a->l0.mvc[ref][0] = 16x16 vector, [1..4] = 8x8 vectors
    int inx, iny;                                           // lane = raw candidate i of the search in progress
    int cdir;                                               // B slices: lane = motion cache grid index: the block is predicted by direct inference
    // B slices keep list 1 sixteen lanes up in every table: result slots 16 + slot (the bi-predictive 16x16 pair lives in slots 32 / 48), cache grid
    // 16 + index, mvc 32 + ref * 5 + q
};
__device__ __forceinline__ int rl(int v, int i) { return __builtin_amdgcn_readlane(v, i); }
__device__ __forceinline__ void wl(int &v, int lane, int i, int x) { v = lane == i ? x : v; }
__device__ __forceinline__ void me_store(MeState &S, int lane, int slot, int mvx, int mvy, int cost, int cost_mv, int ref, int refcost, int mvpx, int mvpy)
{
    const bool m = lane == slot;
    S.mvx = m ? mvx : S.mvx; S.mvy = m ? mvy : S.mvy; S.cost = m ? cost : S.cost; S.costmv = m ? cost_mv : S.costmv;
    S.ref = m ? ref : S.ref; S.refcost = m ? refcost : S.refcost; S.mvpx = m ? mvpx : S.mvpx; S.mvpy = m ? mvpy : S.mvpy;
}

__device__ __forceinline__ void lds_sync() { lds_order(); }

// x264_macroblock_deblock ([x264-upstream] encoder/macroblock.c; h->mb.b_deblock_rdo, --subme 9 and up; oracle macroblock_deblock): the INTERNAL luma edges
// of a whole-macroblock RD candidate's reconstruction, held as a 16x16 byte tile in LDS (row stride 16), are loop-filtered before its distortion is
// measured.  Lane r < 16 owns row r for the vertical edges, then column r for the horizontal ones (a row / column is touched by its owner alone, so only
// the two directions need a barrier between them).  nz16: bit y * 4 + x = the 4x4 block at (x, y) is coded; mvm: bit dir * 2 + half = the two 8x8 blocks
// either side of edge 2 in that half differ in a reference index or by a vector component of 4 quarter samples or more.
__device__ __forceinline__ void mb_deblock_rdo(uint8_t *tile, int lane, bool intra, bool t8, unsigned nz16, unsigned mvm, int qp, int aoff, int boff)
{
    const int ia = min(max(qp + aoff, 0), 51), ib = min(max(qp + boff, 0), 51);
    const int alpha = d_alpha_table[ia], beta = d_beta_table[ib];
    if (!alpha || !beta) return;
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
        if (lane < 16) {
            const int i = lane >> 2;
            for (int e = 1; e < 4; e++) {
                if (t8 && (e & 1)) continue;
                const int x = dir ? i : e, y = dir ? e : i, xn = dir ? x : x - 1, yn = dir ? y - 1 : y;
                const int bs = intra ? 3 : (((nz16 >> (y * 4 + x)) | (nz16 >> (yn * 4 + xn))) & 1) ? 2 : (e == 2 && ((mvm >> (dir * 2 + (i >> 1))) & 1)) ? 1 : 0;
                if (bs) filter_luma_line(tile + (dir ? 4 * e * 16 + lane : lane * 16 + 4 * e), dir ? 16 : 1, alpha, beta, d_tc0_table[ia][bs - 1], bs);
            }
        }
        lds_sync();
    }
}


// ------------------------------------------------------------------------------------------------
// motion vector prediction on the cache grid (oracle predict_mv / predict_mv_pskip)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mb_predict_mv(const MeState &S, int partition, int bx8, int by8, int w8, int ref, int &mvpx, int &mvpy, int go = 0)
{
    const int ia = (by8 + 1) * 4 + bx8 + go, ib = by8 * 4 + bx8 + 1 + go;      // go: 0 = list 0's cache, 16 = list 1's
    int ic = by8 * 4 + bx8 + w8 + 1 + go;
    if (rl(S.cref, ic) == -2) ic = by8 * 4 + bx8 + go;
    const int ra = rl(S.cref, ia), rb = rl(S.cref, ib), rc = rl(S.cref, ic);
    const int ax = rl(S.cmvx, ia), ay = rl(S.cmvy, ia), bx = rl(S.cmvx, ib), by = rl(S.cmvy, ib), cx = rl(S.cmvx, ic), cy = rl(S.cmvy, ic);
    if (partition == D_16x8) {
        if (by8 == 0) { if (rb == ref) { mvpx = bx; mvpy = by; return; } }
        else if (ra == ref) { mvpx = ax; mvpy = ay; return; }
    } else if (partition == D_8x16) {
        if (bx8 == 0) { if (ra == ref) { mvpx = ax; mvpy = ay; return; } }
        else if (rc == ref) { mvpx = cx; mvpy = cy; return; }
    }
    const int cnt = (ra == ref) + (rb == ref) + (rc == ref);
    if (cnt == 1) {
        if (ra == ref) { mvpx = ax; mvpy = ay; } else if (rb == ref) { mvpx = bx; mvpy = by; } else { mvpx = cx; mvpy = cy; }
    } else if (cnt == 0 && rb == -2 && rc == -2 && ra != -2) { mvpx = ax; mvpy = ay; }
    else { mvpx = median3(ax, bx, cx); mvpy = median3(ay, by, cy); }
    mvpx = uni(mvpx); mvpy = uni(mvpy);
}

// ------------------------------------------------------------------------------------------------
// One block search = x264_me_search_ref (oracle me_search_ref), or only the sub-pel refinement of an earlier result
// (x264_me_refine_qpel).  Lane = (candidate = lane >> 4, row = lane & 15): four candidates are costed side by side, a lane holds
// one row of the block (rows >= H idle).  Every update replays x264's in-order "first strictly better wins" as the minimum of
// (cost << t | order).
// ------------------------------------------------------------------------------------------------
struct MeJob {
    int W, H, ox, oy, ref;
    int mvpx, mvpy;
    int n_mvc;                 // raw candidates in L.mvc_in
    bool search;               // false: refinement only (b_refine_qpel)
    bool qonly = false;        // x264_me_refine_qpel_refdupe: a search that starts at the given vector and skips the full-pel stage
    int hp_it, qp_it;
    bool use_thresh;
};

// one row of W pixels at picture position (x, y) displaced by the quarter-pel vector, from global memory (mc.get_ref)
__device__ __forceinline__ void mc_row_global(const uint8_t *__restrict__ p00, size_t pb, int rs, int x, int y, int mvx, int mvy, bool w16, uint32_t out[4])
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const long base = (long)(y + (mvy >> 2)) * rs + x + (mvx >> 2);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3, pl1 = (kQpelPlane1Packed >> (2 * idx)) & 3;
    const uint8_t *a = p00 + pl0 * pb + base + ((mvy & 3) == 3 ? rs : 0), *b = p00 + pl1 * pb + base + ((mvx & 3) == 3 ? 1 : 0);
    const bool avg = (idx & 5) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[i] = 0;
        if (i < 2 || w16) {
            const uint32_t va = load_u32_unaligned(a + 4 * i);
            out[i] = avg ? avg4_u8(va, load_u32_unaligned(b + 4 * i)) : va;
        }
    }
}

// this lane's row of W pixels at picture position (x, y) displaced by the quarter-pel vector, from a reference-cache slot whose sample
// (0, 0) is picture position (X0, Y0) (mc.get_ref on the cached planes)
template <class G = RcGeo<RC_PD, RC_ROWS>>
__device__ __forceinline__ void rc_row(const uint32_t *slot, int X0, int Y0, int x, int y, int mvx, int mvy, bool w16, uint32_t out[4])
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3, pl1 = (kQpelPlane1Packed >> (2 * idx)) & 3;
    const int bx = x + (mvx >> 2) - X0, by = y + (mvy >> 2) - Y0;
    const int o0 = (by + ((mvy & 3) == 3 ? 1 : 0)) * G::COLS + bx, o1 = by * G::COLS + bx + ((mvx & 3) == 3 ? 1 : 0);
    const uint32_t *wa = slot + pl0 * G::PLANE_DW + (o0 >> 2), *wb = slot + pl1 * G::PLANE_DW + (o1 >> 2);
    const bool avg = (idx & 5) != 0;
    uint32_t a[5], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { a[i] = (i < 3 || w16) ? wa[i] : 0u; b[i] = (i < 3 || w16) ? wb[i] : 0u; }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t pa = __builtin_amdgcn_alignbyte(a[i + 1], a[i], o0 & 3), pb_ = __builtin_amdgcn_alignbyte(b[i + 1], b[i], o1 & 3);
        out[i] = (i < 2 || w16) ? (avg ? avg4_u8(pa, pb_) : pa) : 0u;
    }
}

// WP: explicit weights can occur (P slices under --weightp); B slices instantiate without them
// (LDS: the macroblock loop's MbLds<M>, or the lookahead's own smaller layout with the members this function touches — csrc/slicetype.hip)
template <int M, int ME, bool WP = true, typename LDS = MbLds<M>>
__device__ __forceinline__ void me_search(const EncK &k, LDS &L, const MbCtx &c, const MeJob &j, int &mvx, int &mvy, int &cost, int &cost_mv, int &halfpel_thresh, const MeState &S, WinTags &wtg, Prof &pf)
{
    pf.mark(PH_ME_GLUE);
    const int lane = relane(c.lane), r = lane & 15, cnd = lane >> 4;
    const bool w16 = j.W == 16, rowok = r < j.H;
    const uint8_t *p00 = ref_plane00(k, c.s, j.ref);
    const size_t pb = k.plane_bytes;
    const int bx = c.px + j.ox, by = c.py + j.oy;
    // explicit luma weight of this reference (P slices, --weightp): every sample fetched below goes through it after interpolation; the
    // reference cache holds the unweighted planes under the PICTURE's tag, so an index and its duplicate share a slot
    const int wpk = WP && k.wp_any ? uni(k.wl0[j.ref]) : 0;
    const int wcu = WP && k.wc_any ? uni(k.wc0[2 * j.ref]) : 0, wcv = WP && k.wc_any ? uni(k.wc0[2 * j.ref + 1]) : 0;
    const bool wt = WP && (wpk >> 24) != 0;
    const int cref = WP && k.wp_any ? ref_picture(k, j.ref) : j.ref;
    // (the upper half of an 8-pixel row's registers stays zero: weighting it would turn it into the offset and into cost)
#define WP4X4(p) do { if (wt) { p[0] = wp4(p[0], wpk); p[1] = wp4(p[1], wpk); if (w16) { p[2] = wp4(p[2], wpk); p[3] = wp4(p[3], wpk); } } } while (0)
    uint32_t e[4] = { 0, 0, 0, 0 };
    if (rowok) {
        const uint8_t *f = L.src + (j.oy + r) * 16 + j.ox;
        const uint2 a = *(const uint2 *)f; e[0] = a.x; e[1] = a.y;
        if (w16) { const uint2 b = *(const uint2 *)(f + 8); e[2] = b.x; e[3] = b.y; }
    }
    const uint16_t *cmx = c.cost_base + MVCOST_HALF - j.mvpx, *cmy = c.cost_base + MVCOST_HALF - j.mvpy;
    // bits of a vector: both components from the LDS copy of the table; a component 128 samples or more from the predictor reads the table itself
    auto mvc = [&](int qx, int qy) {
        const int dx = abs(qx - j.mvpx), dy = abs(qy - j.mvpy);
        if (__builtin_expect(__any(dx >= MVC_N || dy >= MVC_N), 0)) return (int)cmx[qx] + (int)cmy[qy];
        return (int)L.mvcost[dx] + (int)L.mvcost[dy];
    };
    // SAD of this lane's row at a quarter-pel vector, from global memory; summed over the candidate's 16 lanes
    auto sad_global = [&](int qx, int qy) {
        uint32_t p[4];
        mc_row_global(p00, pb, k.rs, bx, by + r, qx, qy, w16, p);
        WP4X4(p);
        unsigned sd = __builtin_amdgcn_sad_u8(p[0], e[0], 0u);
        sd = __builtin_amdgcn_sad_u8(p[1], e[1], sd); sd = __builtin_amdgcn_sad_u8(p[2], e[2], sd); sd = __builtin_amdgcn_sad_u8(p[3], e[3], sd);
        return row16_sum(rowok ? (int)sd : 0);
    };
    const int fmin0 = c.fmin0, fmax0 = c.fmax0, fmin1 = c.fmin1, fmax1 = c.fmax1;
    constexpr bool umh = ME == 2, esa = ME == 3, cached = ME == 0 || ME == 1;      // cached: the full-pel steps read the reference-cache slot
    // ---- this reference's slot of the reference cache ----
    const int slot = cref >= 3 ? cref - 3 : cref;
    using G = typename LDS::rcg;
    uint32_t *rslot = L.rc + slot * G::SLOT_DW;
    int X0 = __builtin_amdgcn_readlane(wtg.tx, slot), Y0 = __builtin_amdgcn_readlane(wtg.ty, slot);
    bool rhave = __builtin_amdgcn_readlane(wtg.tref, slot) == cref;
    auto rc_inside = [&](int x0, int y0, int x1, int y1) { return rhave && x0 >= X0 && x1 <= X0 + G::COLS && y0 >= Y0 && y1 <= Y0 + G::ROWS; };
    // the block displaced by full-pel (mx +- rad, my +- rad) / by the quarter-pel vector (qx, qy) / anywhere within M samples of full-pel (cx, cy)
    auto in_fpel = [&](int mx, int my, int rad) { return rc_inside(bx + mx - rad, by + my - rad, bx + mx + rad + j.W, by + my + rad + j.H); };
    auto in_qpel = [&](int qx, int qy) { return rc_inside(bx + (qx >> 2), by + (qy >> 2), bx + (qx >> 2) + j.W + 1, by + (qy >> 2) + j.H + 1); };
    auto in_sub = [&](int cx, int cy) { return rc_inside(bx + cx - M, by + cy - M, bx + cx + j.W + M + 1, by + cy + j.H + M + 1); };
    // (re-)centre the slot on the macroblock displaced by (cx, cy): twenty requests per lane, then (after other work) the copy into LDS
    constexpr int NV = 4 * G::NT;
    auto rc_issue = [&](int cx, int cy, uint32_t v[NV]) {
        X0 = clampi((c.px + cx - LDS::rc_mx) & ~3, -PAD, k.cw + PAD - G::COLS); Y0 = clampi(c.py + cy - LDS::rc_my, -PAD, k.ch + PAD - G::ROWS);
#pragma unroll
        for (int t = 0; t < G::NT; t++) {
            const int i = lane + 64 * t, row = G::row_of(i), col = i - row * G::PD;
            const long o = (long)(Y0 + row) * k.rs + X0 + 4 * col;
#pragma unroll
            for (int pl = 0; pl < 4; pl++) v[pl * G::NT + t] = i < G::PLANE_DW ? *(const uint32_t *)(p00 + pl * pb + o) : 0u;
        }
    };
    auto rc_commit = [&](const uint32_t v[NV]) {
        lds_sync();
#pragma unroll
        for (int t = 0; t < G::NT; t++) {
            const int i = lane + 64 * t;
#pragma unroll
            for (int pl = 0; pl < 4; pl++) if (i < G::PLANE_DW) rslot[pl * G::PLANE_DW + i] = v[pl * G::NT + t];
        }
        rhave = true;
#ifndef MB_PROF_RD
        pf.count(15);
#endif
        { const bool m = lane == slot; wtg.tref = m ? cref : wtg.tref; wtg.tx = m ? X0 : wtg.tx; wtg.ty = m ? Y0 : wtg.ty; }
        lds_sync();
        X0 = uni(X0); Y0 = uni(Y0);          // (wave-uniform; behind the lane-dependent stores above the compiler takes the window's origin for divergent otherwise)
    };
    auto rc_stage = [&](int cx, int cy) { uint32_t v[NV]; rc_issue(cx, cy, v); rc_commit(v); };
    // the chroma taps of the search START are requested before the full-pel search runs and used if the search ends there
    constexpr bool spec = cached && M == 2;
    uint32_t spc[2];
    bool spec_on = false;
    int spx = 0, spy = 0;

    if (j.search && !j.qonly) {
        int bmx, bmy, bcost, bpred_cost = MB_COST_MAX, bpred_mx = 0, bpred_my = 0, pmx, pmy;
        const bool sub3 = c.subme >= 3;
        // ---- predictor + candidates (L.cand[0] = the predictor, then the surviving candidates in order) ----
        int pmvx, pmvy;
        if (sub3) { pmvx = clampi(j.mvpx, fmin0 * 4, fmax0 * 4); pmvy = clampi(j.mvpy, fmin1 * 4, fmax1 * 4); pmx = (pmvx + 2) >> 2; pmy = (pmvy + 2) >> 2; }
        else { pmx = clampi((j.mvpx + 2) >> 2, fmin0, fmax0); pmy = clampi((j.mvpy + 2) >> 2, fmin1, fmax1); pmvx = pmx * 4; pmvy = pmy * 4; }
        // lane i filters raw candidate i as x264 does (zero and the predictor itself are not tried again); a candidate equal to an earlier one
        // (or, after clipping, to the predictor) cannot beat it under "first strictly better wins" and is not costed twice
        int n;
        {
            int mx = 0, my = 0;
            bool keep = false;
            if (lane < j.n_mvc) {
                mx = S.inx; my = S.iny;
                if (sub3) { keep = (mx | my) && !(mx == pmvx && my == pmvy); mx = clampi(mx, fmin0 * 4, fmax0 * 4); my = clampi(my, fmin1 * 4, fmax1 * 4); }
                else { mx = clampi((mx + 2) >> 2, fmin0, fmax0); my = clampi((my + 2) >> 2, fmin1, fmax1); keep = (mx | my) && !(mx == pmx && my == pmy); mx *= 4; my *= 4; }
            }
            const int pk = (mx << 16) | (my & 0xffff);
            if (pk == ((pmvx << 16) | (pmvy & 0xffff))) keep = false;
            const unsigned long long km0 = __ballot(keep);
            for (int t = 0; t + 1 < j.n_mvc; t++)
                if ((km0 >> t) & 1) { const int o = __builtin_amdgcn_readlane(pk, t); if (lane > t && pk == o) keep = false; }
            const unsigned long long km = __ballot(keep);
            const int rank = __builtin_popcountll(km & ((1ull << lane) - 1));
            if (keep) { L.cand[1 + rank][0] = (int16_t)mx; L.cand[1 + rank][1] = (int16_t)my; }
            if (lane == 0) { L.cand[0][0] = (int16_t)pmvx; L.cand[0][1] = (int16_t)pmvy; }
            n = 1 + __builtin_popcountll(km);
        }
        lds_sync();
        // ---- the start candidates.  dia / hex: every full-pel step reads the reference-cache slot, with LDS slices of the mv-cost table; esa:
        //      one +-17 full-pel window around the start; umh roams up to ~1.5 x merange from the start on global memory.
        //      One trip to memory at most: the slot around the predictor (if it does not hold that area yet), the cost slices, and the rows of
        //      those candidates the slot does not hold are all requested before the first result is used ----
        int i_me_range = k.me_range;
        unsigned key = 0xffffffffu;
        int pmv_cost = 0;
        {
            const bool pre = cached && !in_fpel(pmx, pmy, 2);
            uint32_t rv[NV];
            if (pre) { rhave = false; rc_issue(pmx, pmy, rv); rhave = true; }          // geometry known now, samples after rc_commit
            uint32_t pp[3][4];
            int cm[3], cqx[3], cqy[3];
            bool inl[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                pp[t][0] = pp[t][1] = pp[t][2] = pp[t][3] = 0; cm[t] = 0; cqx[t] = cqy[t] = 0; inl[t] = false;
                if (t * 4 < n) {
                    const int i = t * 4 + cnd, ii = i < n ? i : 0;
                    const int qx = L.cand[ii][0], qy = L.cand[ii][1];
                    cqx[t] = qx; cqy[t] = qy;
                    inl[t] = !esa && __all(in_qpel(qx, qy));                           // the whole group of four from the slot, or from memory
                    if (!inl[t]) { mc_row_global(p00, pb, k.rs, bx, by + r, qx, qy, w16, pp[t]); WP4X4(pp[t]); }
                    cm[t] = (sub3 || ii > 0) ? mvc(qx, qy) : 0;       // below subme 3 the rounded predictor is costed without its vector bits
                }
            }
            if (pre) { rhave = false; rc_commit(rv); }
            else lds_sync();
            uni_all(X0, Y0, rhave);
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (t * 4 < n) {
                    const int i = t * 4 + cnd;
                    if (inl[t]) { rc_row<G>(rslot, X0, Y0, bx, by + r, cqx[t], cqy[t], w16, pp[t]); WP4X4(pp[t]); }
                    unsigned sd = __builtin_amdgcn_sad_u8(pp[t][0], e[0], 0u);
                    sd = __builtin_amdgcn_sad_u8(pp[t][1], e[1], sd); sd = __builtin_amdgcn_sad_u8(pp[t][2], e[2], sd); sd = __builtin_amdgcn_sad_u8(pp[t][3], e[3], sd);
                    const int cst = row16_sum(rowok ? (int)sd : 0) + cm[t];
                    const unsigned kk = i < n ? ((unsigned)cst << 4) | (unsigned)i : 0xffffffffu;
                    if (t == 0) pmv_cost = __builtin_amdgcn_readlane(cst, 0);
                    key = min(key, rows_min_u32(kk));
                }
        }
        {
            const int bi = key & 15;
            bpred_cost = (int)(key >> 4); bpred_mx = L.cand[bi][0]; bpred_my = L.cand[bi][1];
        }
        const bool pmv_nonzero = (pmvx | pmvy) != 0;
        bmx = (bpred_mx + 2) >> 2; bmy = (bpred_my + 2) >> 2;
        bmx = __builtin_amdgcn_readfirstlane(bmx); bmy = __builtin_amdgcn_readfirstlane(bmy);
        lds_sync();
        pf.mark(PH_ME_PRED);
        if (cached) {
            if (!in_fpel(bmx, bmy, 2)) rc_stage(bmx, bmy);                                     // a far candidate won
            uni_all(X0, Y0, rhave);
        }
        pf.mark(PH_ME_WIN);
        // esa: its own window (aliases the slots: their tags are dropped when it is staged)
        int wx0 = 0, wy0 = 0;
#define MVC(qx, qy) mvc(qx, qy)
        auto fpel = [&](int mx, int my) {       // full-pel candidate cost (valid after the row sum)
            if (umh) return sad_global(mx * 4, my * 4) + mvc(mx * 4, my * 4);
            const uint8_t *wrow = esa ? (const uint8_t *)L.rc + (by + my + r - wy0) * WIN_STRIDE : (const uint8_t *)rslot + (by + my + r - Y0) * G::COLS;
            const int xo = bx + mx - (esa ? wx0 : X0);
            int sd = 0;
            if (wt) {       // x264 searches the weighted copy of the plane (p_fref_w): pointwise, so weight the row read from the unweighted one
                uint32_t p[4];
                row16_lds(wrow, xo, p);
                WP4X4(p);
                unsigned s2 = __builtin_amdgcn_sad_u8(p[0], e[0], 0u);
                s2 = __builtin_amdgcn_sad_u8(p[1], e[1], s2);
                if (w16) { s2 = __builtin_amdgcn_sad_u8(p[2], e[2], s2); s2 = __builtin_amdgcn_sad_u8(p[3], e[3], s2); }
                sd = rowok ? (int)s2 : 0;
            } else
            if (rowok) sd = w16 ? sad_row16_lds(wrow, xo, e) : sad8_lds(wrow, xo, e[0], e[1]);
            return row16_sum(sd) + MVC(mx * 4, my * 4);
        };
        // blocks of eight rows (8x8, 16x8): eight candidates a batch — lane = (candidate = lane >> 3, row = lane & 7); the source row comes from LDS
        // again (it is not kept in registers beside e[])
        const bool h8 = cached && j.H == 8;
        auto fpel8 = [&](int mx, int my) {
            const int r8 = lane & 7;
            const uint8_t *f = L.src + (j.oy + r8) * 16 + j.ox;
            const uint8_t *wrow = (const uint8_t *)rslot + (by + my + r8 - Y0) * G::COLS;
            const int xo = bx + mx - X0;
            int sd;
            if (wt) {
                uint32_t p[4];
                row16_lds(wrow, xo, p);
                WP4X4(p);
                const uint2 sa = *(const uint2 *)f;                 // (an 8-pixel row at column 8 is only 8-byte aligned)
                unsigned s2 = __builtin_amdgcn_sad_u8(p[0], sa.x, 0u);
                s2 = __builtin_amdgcn_sad_u8(p[1], sa.y, s2);
                if (w16) { const uint2 sb = *(const uint2 *)(f + 8); s2 = __builtin_amdgcn_sad_u8(p[2], sb.x, s2); s2 = __builtin_amdgcn_sad_u8(p[3], sb.y, s2); }
                sd = (int)s2;
            } else
            if (w16) { const uint4 sv = *(const uint4 *)f; const uint32_t e8[4] = { sv.x, sv.y, sv.z, sv.w }; sd = sad_row16_lds(wrow, xo, e8); }
            else { const uint2 sv = *(const uint2 *)f; sd = sad8_lds(wrow, xo, sv.x, sv.y); }
            sd = quad_sum(sd); sd += dpp<DPP_ROW_HALF_MIRROR>(sd);
            return sd + MVC(mx * 4, my * 4);
        };
        // the slot follows a search that walks out of it (dia / hex; the cost slices reach +-24 around their centre: merange <= 16)
        auto ensure = [&](int mx, int my, int rad) { if (cached && !in_fpel(mx, my, rad)) rc_stage(mx, my); uni_all(X0, Y0, rhave); };
        {   // the rounded best predictor and the zero vector: groups 0 and 1
            int c_round, c_zero;
            if (!cached) {
                const int qx = cnd == 0 ? bmx * 4 : 0, qy = cnd == 0 ? bmy * 4 : 0;
                const int cst = sad_global(qx, qy) + mvc(qx, qy);
                c_round = __builtin_amdgcn_readlane(cst, 0); c_zero = __builtin_amdgcn_readlane(cst, 16);
            } else {
                const bool zlds = in_fpel(0, 0, 0);
                int cst;
                if (zlds) cst = fpel(cnd == 0 ? bmx : 0, cnd == 0 ? bmy : 0);
                else {
                    cst = fpel(bmx, bmy);
                    const int z = sad_global(0, 0) + mvc(0, 0);
                    if (cnd == 1) cst = z;
                }
                c_round = __builtin_amdgcn_readlane(cst, 0); c_zero = __builtin_amdgcn_readlane(cst, 16);
            }
            if (sub3) {
                bcost = ((bpred_mx | bpred_my) & 3) ? c_round : bpred_cost;
                if (pmv_nonzero) { if ((bmx | bmy) && c_zero < bcost) { bcost = c_zero; bmx = 0; bmy = 0; } }
                else if (pmv_cost < bcost) { bcost = pmv_cost; bmx = 0; bmy = 0; }
            } else {
                bcost = bpred_cost;
                if (pmv_nonzero && c_zero < bcost) { bcost = c_zero; bmx = 0; bmy = 0; }
            }
        }
        bmx = __builtin_amdgcn_readfirstlane(bmx); bmy = __builtin_amdgcn_readfirstlane(bmy);
        if (spec && c.subme >= 2 && c.chroma_me) {
            spec_on = true; spx = bmx; spy = bmy;
            chroma_issue2(spc, ref_chroma00(k, c.s, j.ref), k.rs, ((bx >> 1) + (spx >> 1) - CSubGeo<M>::MG) & ~1, (by >> 1) + (spy >> 1) - CSubGeo<M>::MG,
                          CSubGeo<M>::ndw(j.W >> 1), CSubGeo<M>::rows(j.H >> 1), lane);
        }
        if (esa) {      // its window around the start
            lds_sync();
            wtg.tref = -1; rhave = false;
            wx0 = clampi((bx + bmx - WIN_R) & ~7, -PAD, k.cw + PAD - WIN_COLS); wy0 = clampi(by + bmy - WIN_R, -PAD, k.ch + PAD - WIN_ROWS);
            for (int i = lane; i < WIN_ROWS * 8; i += 64) {
                const int row = i >> 3, col = (i & 7) * 8;
                const uint2 v = *(const uint2 *)(p00 + (long)(wy0 + row) * k.rs + wx0 + col);
                uint32_t *d = (uint32_t *)((uint8_t *)L.rc + row * WIN_STRIDE + col);
                d[0] = v.x; d[1] = v.y;
            }
        }
        if (esa) lds_sync();
        bool hexrefine = true;
        if (umh) {
            // X264_ME_UMH (oracle me_search_ref case 2): four candidates per step, in-order "strictly better wins" = min of (cost << 2 | order)
            auto inrange = [&](int mx, int my) { return mx >= fmin0 && mx <= fmax0 && my >= fmin1 && my <= fmax1; };
            auto step = [&](int mx, int my, bool valid) {
                const int sx = valid ? mx : bmx, sy = valid ? my : bmy;            // masked candidates stay inside the padded plane
                const int cst = fpel(sx, sy);
                unsigned kk = valid ? ((unsigned)cst << 2) | (unsigned)cnd : 0xffffffffu;
                kk = rows_min_u32(kk);
                if (kk != 0xffffffffu && (int)(kk >> 2) < bcost) {
                    const int wl = (int)(kk & 3) * 16;
                    bcost = (int)(kk >> 2); bmx = __shfl(sx, wl); bmy = __shfl(sy, wl);
                }
            };
            const int d1x = cnd == 2 ? -1 : cnd == 3 ? 1 : 0, d1y = cnd == 0 ? -1 : cnd == 1 ? 1 : 0;       // DIA1: (0,-1) (0,1) (-1,0) (1,0)
            const int shift = (w16 ? 0 : 1) + (j.H == 16 ? 0 : 1);
#define UMH_TH(v) (bcost < ((v) >> shift))
            const int ucost1 = bcost;
            step(pmx + d1x, pmy + d1y, true);
            if (pmx | pmy) step(d1x, d1y, true);
            const int ucost2 = bcost;
            if ((bmx | bmy) && ((bmx - pmx) | (bmy - pmy))) step(bmx + d1x, bmy + d1y, true);
            int cross_start = bcost == ucost2 ? 3 : 1;
            const int omx = bmx, omy = bmy;
            auto cross = [&](int start, int xmax, int ymax) {
                for (int i0 = start; i0 < xmax; i0 += 4) {
                    const int i = i0 + 2 * (cnd >> 1), mx = omx + ((cnd & 1) ? -i : i);
                    step(mx, omy, i < xmax && ((cnd & 1) ? mx >= fmin0 : mx <= fmax0));
                }
                for (int i0 = start; i0 < ymax; i0 += 4) {
                    const int i = i0 + 2 * (cnd >> 1), my = omy + ((cnd & 1) ? -i : i);
                    step(omx, my, i < ymax && ((cnd & 1) ? my >= fmin1 : my <= fmax1));
                }
            };
            bool done = false;
            if (bcost == ucost2 && UMH_TH(2000)) {
                step(omx + (cnd == 0 ? 0 : cnd == 1 ? -1 : cnd == 2 ? 1 : -2), omy + (cnd == 0 ? -2 : cnd == 3 ? 0 : -1), true);     // (0,-2) (-1,-1) (1,-1) (-2,0)
                step(omx + (cnd == 0 ? 2 : cnd == 1 ? -1 : cnd == 2 ? 1 : 0), omy + (cnd == 0 ? 0 : cnd == 3 ? 2 : 1), true);       // (2,0) (-1,1) (1,1) (0,2)
                if (bcost == ucost1 && UMH_TH(500)) done = true;
                else if (bcost == ucost2) {
                    const int r1 = (i_me_range >> 1) | 1;
                    cross(3, r1, r1);
                    step(omx + (cnd == 0 ? -1 : cnd == 1 ? 1 : cnd == 2 ? -2 : 2), omy + (cnd < 2 ? -2 : -1), true);                // (-1,-2) (1,-2) (-2,-1) (2,-1)
                    step(omx + (cnd == 0 ? -2 : cnd == 1 ? 2 : cnd == 2 ? -1 : 1), omy + (cnd < 2 ? 1 : 2), true);                  // (-2,1) (2,1) (-1,2) (1,2)
                    if (bcost == ucost2) done = true;
                    cross_start = r1 + 2;
                }
            }
            if (!done) {
                if (j.n_mvc) {      // adaptive search range: agreement of the predictors x SAD level
                    int mvd, denom = 1;
                    const bool is16 = w16 && j.H == 16;
                    if (j.n_mvc == 1) mvd = is16 ? 25 : abs(j.mvpx - rl(S.inx, 0)) + abs(j.mvpy - rl(S.iny, 0));
                    else {
                        denom = j.n_mvc - 1; mvd = 0;
                        if (!is16) { mvd = abs(j.mvpx - rl(S.inx, 0)) + abs(j.mvpy - rl(S.iny, 0)); denom++; }
                        for (int i = 0; i < j.n_mvc - 1; i++) mvd += abs(rl(S.inx, i) - rl(S.inx, i + 1)) + abs(rl(S.iny, i) - rl(S.iny, i + 1));
                    }
                    const int sad_ctx = UMH_TH(1000) ? 0 : UMH_TH(2000) ? 1 : UMH_TH(4000) ? 2 : 3;
                    const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                    i_me_range = (i_me_range * (int)((0x6544544444434433ull >> (4 * (mvd_ctx * 4 + sad_ctx))) & 15)) >> 2;
                }
                cross(cross_start, i_me_range, i_me_range >> 1);
                step(omx + ((cnd & 2) ? 2 : -2), omy + ((cnd & 1) ? 2 : -2), true);                                                 // (-2,-2) (-2,2) (2,-2) (2,2)
                const int hx = bmx, hy = bmy;
                int i = 1;
                do {
                    for (int ps = 0; ps < 4; ps++) {
                        const int jj = 4 * ps + cnd, mx = hx + umh_hex4_dx(jj) * i, my = hy + umh_hex4_dy(jj) * i;
                        step(mx, my, inrange(mx, my));
                    }
                } while (++i <= i_me_range >> 2);
                if (!inrange(bmx, bmy)) done = true;
            }
            hexrefine = !done;
#undef UMH_TH
        }
        if (ME == 3) {
            // X264_ME_ESA: every position of the clipped +-merange rectangle (width rounded up to 4, positions right of the full-pel limit
            // skipped), raster order, strictly better wins = minimum of (cost << 11 | raster index)
            const int rr = k.me_range;
            const int min_x = max(bmx - rr, fmin0), min_y = max(bmy - rr, fmin1), max_x = min(bmx + rr, fmax0), max_y = min(bmy + rr, fmax1);
            const int width = (max_x - min_x + 3) & ~3;
            unsigned kmin = 0xffffffffu;
            for (int my = min_y; my <= max_y; my++)
                for (int x4 = 0; x4 < width; x4 += 4) {
                    const int mx = min_x + x4 + cnd;
                    const bool ok = mx <= fmax0;
                    const unsigned cst = (unsigned)fpel(ok ? mx : min_x, my);
                    if (ok) kmin = min(kmin, (cst << 11) | (unsigned)((my - min_y) * width + x4 + cnd));
                }
            kmin = wave_min_u32(kmin);
            if (kmin != 0xffffffffu && (int)(kmin >> 11) < bcost) {
                const int idx = (int)(kmin & 2047);
                bcost = (int)(kmin >> 11); bmx = min_x + idx % width; bmy = min_y + idx / width;
            }
        } else if (ME == 0) {
            // X264_ME_DIA: the four neighbours are the four lane groups; the centre wins ties
            int it = k.me_range;
            do {
                ensure(bmx, bmy, 1);
                const unsigned kk = rows_min_u32(((unsigned)fpel(bmx + (cnd == 2 ? -1 : cnd == 3 ? 1 : 0), bmy + (cnd == 0 ? -1 : cnd == 1 ? 1 : 0)) << 2) | (unsigned)cnd);
                if ((int)(kk >> 2) >= bcost) break;
                const int q = kk & 3;
                bcost = (int)(kk >> 2);
                bmx += q == 2 ? -1 : q == 3 ? 1 : 0; bmy += q == 0 ? -1 : q == 1 ? 1 : 0;
            } while (--it && bmx >= fmin0 && bmx <= fmax0 && bmy >= fmin1 && bmy <= fmax1);
        } else if (hexrefine) {
            // X264_ME_HEX (and the tail of umh): hexagon, then square refine
            unsigned hk = (unsigned)bcost << 3;
            ensure(bmx, bmy, 2);
            if (h8) {       // the six points in one batch
                const int c8 = lane >> 3, i = 1 + (c8 < 6 ? c8 : 0);
                unsigned kk = ((unsigned)fpel8(bmx + hex_dx(i), bmy + hex_dy(i)) << 3) | (unsigned)(i + 1);
                if (c8 >= 6) kk = 0xffffffffu;
                hk = min(hk, halfrows_min_u32(kk));
            } else {
                int i = 1 + cnd;
                unsigned kk = ((unsigned)fpel(bmx + hex_dx(i), bmy + hex_dy(i)) << 3) | (unsigned)(i + 1);
                hk = min(hk, rows_min_u32(kk));
                i = 5 + (cnd & 1);
                kk = ((unsigned)fpel(bmx + hex_dx(i), bmy + hex_dy(i)) << 3) | (unsigned)(i + 1);
                if (cnd >= 2) kk = 0xffffffffu;
                hk = min(hk, rows_min_u32(kk));
            }
            if (hk & 7) {
                int dir = (int)(hk & 7) - 2;
                bmx += hex_dx(dir + 1); bmy += hex_dy(dir + 1);
                for (int it = (i_me_range >> 1) - 1; it > 0 && bmx >= fmin0 && bmx <= fmax0 && bmy >= fmin1 && bmy <= fmax1; it--) {
                    hk &= ~7u;
                    ensure(bmx, bmy, 2);
                    const int cc = cnd < 3 ? cnd : 0;
                    unsigned kk = ((unsigned)fpel(bmx + hex_dx(dir + cc), bmy + hex_dy(dir + cc)) << 3) | (unsigned)(cc + 1);
                    if (cnd >= 3) kk = 0xffffffffu;
                    hk = min(hk, rows_min_u32(kk));
                    if (!(hk & 7)) break;
                    dir += (int)(hk & 7) - 2;
                    dir = dir < 0 ? 5 : dir > 5 ? 0 : dir;
                    bmx += hex_dx(dir + 1); bmy += hex_dy(dir + 1);
                }
            }
            bcost = (int)(hk >> 3);
            unsigned sk = (unsigned)bcost << 4;
            ensure(bmx, bmy, 1);
            if (h8) {       // the eight points in one batch
                const int q = 1 + (lane >> 3);
                sk = min(sk, halfrows_min_u32(((unsigned)fpel8(bmx + sq_dx(q), bmy + sq_dy(q)) << 4) | (unsigned)q));
            } else {
                int q = 1 + cnd;
                sk = min(sk, rows_min_u32(((unsigned)fpel(bmx + sq_dx(q), bmy + sq_dy(q)) << 4) | (unsigned)q));
                q = 5 + cnd;
                sk = min(sk, rows_min_u32(((unsigned)fpel(bmx + sq_dx(q), bmy + sq_dy(q)) << 4) | (unsigned)q));
            }
            const int bd = sk & 15;
            bcost = (int)(sk >> 4);
            if (bd) { bmx += sq_dx(bd); bmy += sq_dy(bd); }
        }
        // ---- -> quarter-pel vector ----
        if (!sub3) {
            cost = bcost;
            if (bmx == pmx && bmy == pmy) cost += MVC(bmx * 4, bmy * 4);        // the real cost of the predictor
            mvx = bmx * 4; mvy = bmy * 4;
        } else if (bpred_cost < bcost) {
            mvx = bpred_mx; mvy = bpred_my; cost = bpred_cost;
        } else { mvx = bmx * 4; mvy = bmy * 4; cost = bcost; }
        mvx = __builtin_amdgcn_readfirstlane(mvx); mvy = __builtin_amdgcn_readfirstlane(mvy);
        cost_mv = MVC(mvx, mvy);
#undef MVC
        pf.mark(PH_ME_FPEL);
        if (c.subme < 2) return;
    }

    // ---- refine_subpel ----
    const bool refq = !j.search;
    int bmx = mvx, bmy = mvy, bcost = cost;
    const bool chroma_me = c.chroma_me;
    if (j.hp_it && c.subme < 3) {     // the sub-pel component of the predicted vector (anywhere: global memory)
        const int mx = clampi(j.mvpx, c.smin0 + 2, c.smax0 - 2), my = clampi(j.mvpy, c.smin1 + 2, c.smax1 - 2);
        if ((mx - bmx) | (my - bmy)) {
            const int cst = __builtin_amdgcn_readlane(sad_global(mx, my), 0) + mvc(mx, my);
            if (cst < bcost) { bcost = cst; bmx = mx; bmy = my; }
        }
    }
    // every sample the diamonds below can touch lies within M px of the rounded start: the reference-cache slot must hold that area
    const int ctrx = (bmx + 2) >> 2, ctry = (bmy + 2) >> 2;
    lds_sync();
    if (!in_sub(ctrx, ctry)) rc_stage(ctrx, ctry);
    uni_all(X0, Y0, rhave);
    const bool spec_hit = spec && spec_on && ctrx == spx && ctry == spy;
    // chroma: lane r owns row (r & 3) of 4x4 chroma block r >> 2 of the partition, both planes
    const int cbw = j.W >> 3, ncb = cbw * (j.H >> 3), cblk = r >> 2;
    const bool cact = cblk < ncb;
    const int ccx = (j.ox >> 1) + (cblk % cbw) * 4, ccy = (j.oy >> 1) + (cblk / cbw) * 4 + (r & 3);
    uint32_t ce0 = 0, ce1 = 0;
    uint32_t *cb = L.csub;
    const int cndw = CSubGeo<M>::ndw(j.W >> 1), cnr = CSubGeo<M>::rows(j.H >> 1);
    const int cx0c = ((bx >> 1) + (ctrx >> 1) - CSubGeo<M>::MG) & ~1, cy0c = (by >> 1) + (ctry >> 1) - CSubGeo<M>::MG;
    if (chroma_me) {
        if (cact) { const uint2 v = *(const uint2 *)(L.csrc + ccy * 16 + 2 * ccx); ce0 = v.x; ce1 = v.y; }
        if (spec_hit) chroma_commit2(cb, spc, cndw, cnr, lane);
        else chroma_stage(cb, ref_chroma00(k, c.s, j.ref), k.rs, cx0c, cy0c, cndw, cnr, lane, 64);
    }
    lds_sync();
    pf.mark(PH_ME_SUBSTAGE);
    const s16x2 sg1 = pk_sign(lane & 1), sg2 = pk_sign(lane & 2);
    auto mvc2 = [&](int qx, int qy) { return mvc(qx, qy); };
    auto fetch2 = [&](int qx, int qy, uint32_t p[4]) {
        p[0] = p[1] = p[2] = p[3] = 0;
        if (rowok) { rc_row<G>(rslot, X0, Y0, bx, by + r, qx, qy, w16, p); WP4X4(p); }
    };
    auto sad2 = [&](int qx, int qy) {
        uint32_t p[4];
        fetch2(qx, qy, p);
        unsigned sd = __builtin_amdgcn_sad_u8(p[0], e[0], 0u);
        sd = __builtin_amdgcn_sad_u8(p[1], e[1], sd); sd = __builtin_amdgcn_sad_u8(p[2], e[2], sd); sd = __builtin_amdgcn_sad_u8(p[3], e[3], sd);
        return row16_sum(rowok ? (int)sd : 0) + mvc2(qx, qy);
    };
    auto cmp2 = [&](int qx, int qy) {           // mbcmp (SATD above subme 1) + vector bits, without chroma
        if (!c.satd) return sad2(qx, qy);
        uint32_t p[4];
        fetch2(qx, qy, p);
        return row16_sum(w16 ? satd16x4_half_pk<4>(e, p, sg1, sg2) : satd16x4_half_pk<2>(e, p, sg1, sg2)) + mvc2(qx, qy);       // 8-pixel rows: half the work
    };
    auto chroma2 = [&](int qx, int qy) {
        // (explicit chroma weights are rare — a fade: their arithmetic lives in a second instantiation, the common one is the code without them)
        int h = 0;
        if (wcu | wcv) { if (cact) h = chroma_me_lds<true>(cb, cndw, cx0c, cy0c, (c.px >> 1) + ccx, (c.py >> 1) + ccy, qx, qy, ce0, ce1, sg1, sg2, wcu, wcv); }
        else if (cact) h = chroma_me_lds<false>(cb, cndw, cx0c, cy0c, (c.px >> 1) + ccx, (c.py >> 1) + ccy, qx, qy, ce0, ce1, sg1, sg2);
        return row16_sum(h);
    };
    // half-pel diamond on SAD: (0,-2) (0,2) (-2,0) (2,0)
    for (int it = j.hp_it; it > 0; it--) {
        const int cx = bmx + (cnd == 2 ? -2 : cnd == 3 ? 2 : 0), cy = bmy + (cnd == 0 ? -2 : cnd == 1 ? 2 : 0);
        const unsigned kk = rows_min_u32(((unsigned)sad2(cx, cy) << 2) | (unsigned)cnd);
        if ((int)(kk >> 2) >= bcost) break;
        const int b = kk & 3;
        bcost = (int)(kk >> 2);
        bmx += b == 2 ? -2 : b == 3 ? 2 : 0; bmy += b == 0 ? -2 : b == 1 ? 2 : 0;
    }
    if (!refq && (c.satd || chroma_me)) {
        bcost = __builtin_amdgcn_readlane(cmp2(bmx, bmy), 0);
        if (chroma_me) bcost += __builtin_amdgcn_readlane(chroma2(bmx, bmy), 0);
    }
    if (j.use_thresh) {
        if (((bcost * 7) >> 3) > halfpel_thresh) { mvx = bmx; mvy = bmy; cost = bcost; pf.mark(PH_ME_SUBPEL); return; }
        else if (bcost < halfpel_thresh) halfpel_thresh = bcost;
    }
    if (c.subme != 1) {
        int bdir = -1;
        for (int it = j.qp_it; it > 0; it--) {
            if (bmy <= c.smin1 || bmy >= c.smax1 || bmx <= c.smin0 || bmx >= c.smax0) break;
            const int cx = bmx + (cnd == 2 ? -1 : cnd == 3 ? 1 : 0), cy = bmy + (cnd == 0 ? -1 : cnd == 1 ? 1 : 0);
            const bool skip = !refq && (cnd ^ 1) == bdir;             // do not step straight back
            int cst = cmp2(cx, cy);
            if (chroma_me && __any(cst < bcost && !skip)) cst += chroma2(cx, cy);
            unsigned kk = skip ? 0xffffffffu : ((unsigned)cst << 2) | (unsigned)cnd;
            kk = rows_min_u32(kk);
            if ((int)(kk >> 2) >= bcost) break;
            bcost = (int)(kk >> 2);
            bdir = kk & 3;
            bmx += bdir == 2 ? -1 : bdir == 3 ? 1 : 0; bmy += bdir == 0 ? -1 : bdir == 1 ? 1 : 0;
        }
    } else if (bmy > c.smin1 && bmy < c.smax1 && bmx > c.smin0 && bmx < c.smax0) {
        const int cx = bmx + (cnd == 2 ? -1 : cnd == 3 ? 1 : 0), cy = bmy + (cnd == 0 ? -1 : cnd == 1 ? 1 : 0);
        const unsigned kk = rows_min_u32(((unsigned)sad2(cx, cy) << 2) | (unsigned)cnd);
        if ((int)(kk >> 2) < bcost) {
            const int b = kk & 3;
            bcost = (int)(kk >> 2);
            bmx += b == 2 ? -1 : b == 3 ? 1 : 0; bmy += b == 0 ? -1 : b == 1 ? 1 : 0;
        }
    }
    mvx = __builtin_amdgcn_readfirstlane(bmx); mvy = __builtin_amdgcn_readfirstlane(bmy); cost = bcost;
    cost_mv = mvc2(mvx, mvy);
    pf.mark(PH_ME_SUBPEL);
#undef WP4X4
}


// ------------------------------------------------------------------------------------------------
// intra analysis (oracle analyse_intra / analyse_intra_chroma; x264_mb_analyse_intra)
// ------------------------------------------------------------------------------------------------
// x264_chroma_lambda2_offset_tab: 256 * 2^((i - 12) / 3)
static __constant__ uint16_t c_chroma_lambda2_offset[37] = { 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290, 1625, 2048, 2580, 3250, 4096,
                                                             5160, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768, 41285, 52015, 65535 };
static __constant__ int c_lambda2_tab[52] = { 14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1160, 1462, 1843, 2322, 2926,
                                              3686, 4644, 5852, 7373, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628, 117964,
                                              148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436 };

__device__ __forceinline__ int blkidx_of(int bx, int by) { return ((by >> 1) * 2 + (bx >> 1)) * 4 + (by & 1) * 2 + (bx & 1); }

// neighbour availability of 4x4 block b / 8x8 block i8 (oracle i4_avail / i8_avail)
__device__ __forceinline__ int i4_avail(int mbx, int mby, int mbw, int b)
{
    const int bx = z_bx(b), by = z_by(b);
    int a = 0;
    if (bx > 0 || mbx > 0) a |= AVAIL_LEFT;
    if (by > 0 || mby > 0) a |= AVAIL_TOP;
    if ((bx > 0 || mbx > 0) && (by > 0 || mby > 0)) a |= AVAIL_TOPLEFT;
    if (by == 0) { if (mby > 0 && (bx < 3 || mbx + 1 < mbw)) a |= AVAIL_TOPRIGHT; }
    else if (bx < 3 && blkidx_of(bx + 1, by - 1) < b) a |= AVAIL_TOPRIGHT;
    return a;
}
__device__ __forceinline__ int i8_avail(bool left, bool top, bool topright, int i8)
{
    const int x8 = i8 & 1, y8 = i8 >> 1;
    int a = 0;
    if (x8 || left) a |= AVAIL_LEFT;
    if (y8 || top) a |= AVAIL_TOP;
    if ((x8 || left) && (y8 || top)) a |= AVAIL_TOPLEFT;
    if (i8 == 0 ? top : i8 == 1 ? topright : i8 == 2) a |= AVAIL_TOPRIGHT;
    return a;
}
// predicted intra 4x4 mode (8.3.1.1; oracle i4_pred_mode): nm = edge modes of the left / top macroblocks
__device__ __forceinline__ int i4_pred_mode(const uint8_t *nm, int mbx, int mby, int b, const uint8_t *cur)
{
    const int bx = z_bx(b), by = z_by(b);
    int ma, mb_;
    if (bx > 0) ma = cur[blkidx_of(bx - 1, by)];
    else if (mbx > 0) ma = nm[by];
    else return 2;
    if (by > 0) mb_ = cur[blkidx_of(bx, by - 1)];
    else if (mby > 0) mb_ = nm[4 + bx];
    else return 2;
    return min(ma, mb_);
}

// x264's mode choice of one 4x4 / 8x8 block from the raw costs of the nine modes (oracle analyse_intra: V / H / DC first, then only the
// directional modes near the favoured direction; list order and "first strictly better" decide ties).  raw(m) = cost of mode m (the
// caller fetches it from the lanes that computed it).  Returns the cost incl. the predicted-mode bonus; lists are nibble strings, 15 ends one.
template <class F>
__device__ __forceinline__ int pick_intra_mode(F raw, int avail, int pm, int lambda, bool is4, bool every_mode, int &bestm, bool no_stop = false)
{
    const int all3 = AVAIL_LEFT | AVAIL_TOP | AVAIL_TOPLEFT;
    const int id = (avail & all3) == all3 ? 4 : avail & (AVAIL_LEFT | AVAIL_TOP);
    unsigned rest;
    int best = MB_COST_MAX;
    bestm = 2;
    if (id >= 3) {
        int sv = raw(0), sh = raw(1), sdc = raw(2);
        const bool fv = sh > sv;
        if (pm == 0) sv -= 3 * lambda; else if (pm == 1) sh -= 3 * lambda; else if (pm == 2) sdc -= 3 * lambda;
        best = sdc; bestm = 2;
        if (sh < best) { best = sh; bestm = 1; }
        if (sv < best) { best = sv; bestm = 0; }
        // x264's analysis shortcut (modes near the favoured direction only) — or, when RD decides and fast-intra is off, every remaining mode
        rest = every_mode ? (id == 4 ? 0xF876543u : 0xF873u) : id == 4 ? (fv ? 0xF7543u : 0xF864u) : (fv ? 0xF73u : 0xF8u);
    } else rest = id == 0 ? 0xF2u : id == 1 ? 0xF812u : 0xF7302u;
    if (is4) {
        if (best > 0)
            for (; (rest & 15) != 15; rest >>= 4) {
                const int m = (int)(rest & 15);
                int cst = raw(m);
                if (pm == m) { cst -= 3 * lambda; if (cst <= 0) { best = cst; bestm = m; break; } }
                if (cst < best) { best = cst; bestm = m; }
            }
    } else {
        for (; (rest & 15) != 15 && (best >= 0 || no_stop); rest >>= 4) {          // (no_stop: x264's i_mbrd >= 2 wants every 8x8 mode's cost for the refinement)
            const int m = (int)(rest & 15);
            const int cst = raw(m) - (pm == m ? 3 * lambda : 0);
            if (cst < best) { best = cst; bestm = m; }
        }
    }
    return best;
}

}  // namespace x264gpu
#include "cabac_rd.hip.h"          // needs the motion cache and the intra-mode helpers above
#include "trellis.hip.h"
namespace x264gpu {

// The trellis sites of a macroblock's FINAL encode (x264 --trellis 1; k.trellis = the mask of sites, 63 = all): what they need to run the
// search of trellis.hip.h on the slice's live context variables (read only)
enum { TR_P4 = 1, TR_P8 = 2, TR_C = 4, TR_I16 = 8, TR_I4 = 16, TR_I8 = 32 };
struct TrCtx { int on; uint32_t r, r8, model; TrellisTab tt; };
// nblk blocks of category CAT at coefs (scan order, `stride` entries apart), eight per pass; returns the mask of blocks left non-zero
template <int CAT>
__device__ __forceinline__ unsigned trellis_run(const TrCtx &tr, int16_t *coefs, int stride, int nblk, int qp, bool intra, int lane)
{
    unsigned out = 0;
    for (int b0 = 0; b0 < nblk; b0 += 8)
        out |= trellis_blocks<CAT>((lds_i16 *)(coefs + b0 * stride), stride, min(8, nblk - b0), qp, intra, tr.model, tr.tt, CAT == 5 ? tr.r8 : tr.r) << b0;
    return out;
}
__device__ __forceinline__ void load_levels_scan(const int16_t *src, int v[4], int j)
{
    const unsigned z = scan_nibbles(j);
    v[0] = src[z & 15]; v[1] = src[(z >> 4) & 15]; v[2] = src[(z >> 8) & 15]; v[3] = src[z >> 12];
}


// Final encode of an Intra_4x4 / Intra_8x8 macroblock of a trellis session (x264: i_skip_intra is 0 under --trellis 1, so the blocks are coded
// again): the modes of the analysis (L.modes4 / L.modes8), every block predicted from its re-coded neighbours in the tile, transformed,
// quantised by the trellis search, reconstructed.  Levels go to lvw in their final layout; returns the non-zero flags (R.nnz4 / R.nnz8 form).
template <int M>
__device__ __forceinline__ unsigned mb_encode_i4x4_trellis(const EncK &k, MbLds<M> &L, const MbCtx &c, uint32_t cz, uint32_t t4, const Q4 &q4, const TrCtx &trc, int16_t *lvw)
{
    // The blocks in ten rounds along the anti-diagonals x + 2 y instead of sixteen steps in coding order: a block's neighbours (left, top, top left, and the
    // top right one where the coding order makes it available) lie on earlier diagonals, the search of a block reads only the slice's context variables,
    // so the two blocks of a round are independent and share ONE search call (it takes up to eight blocks at the latency of one) — same levels, same samples.
    const int lane = relane(c.lane), j = lane & 3, slot = (lane >> 2) & 1;          // lanes 0..3: the round's first block, 4..7: its second
    uint8_t *tile = L.tile + IT_ORG;
    unsigned nnz4 = 0;
    for (int t = 0; t < 10; t++) {
        // the blocks (bx, by) with bx + 2 by == t: by from max(0, ceil((t - 3) / 2)) to min(3, t / 2); at most two
        const int by0 = t <= 3 ? 0 : (t - 2) >> 1, nb = min(3, t >> 1) - by0 + 1;
        int e[4] = { 0, 0, 0, 0 }, p[4] = { 0, 0, 0, 0 }, v[4];
        uint32_t bp = 0;
        uint8_t *bt = tile;
        int my_idx = 0;
        int16_t *const pair = &L.cand[0][0];          // two blocks' coefficients side by side (the search's candidate list is idle during the encode passes)
        for (int q = 0; q < nb; q++) {
            const int by = by0 + q, bx = t - 2 * by, idx = blkidx_of(bx, by);
            const int avail = i4_avail(c.mbx, c.sy, k.mbw, idx);
            lds_sync();
            uint8_t *bq = tile + by * 4 * IT_STRIDE + bx * 4;
            pred4_build_u(L.U, bq, IT_STRIDE, avail, lane);
            const uint32_t pr = pred4_row4(L.U, t4);
            const uint32_t en = (uint32_t)__shfl((int)cz, idx * 4 + j);
            const int bm = L.modes4[idx];
            const uint32_t bpq = (uint32_t)__shfl((int)pr, bm * 4 + j);
            if (slot == q) { unpack4(en, e); unpack4(bpq, p); bp = bpq; bt = bq; my_idx = idx; }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = e[i] - p[i];
        dct4_quad(v, lane);
        lds_sync();
        if (lane < 4 * nb) store_levels_scan(pair + slot * 16, v, j);
        lds_sync();
        trellis_run<2>(trc, pair, 16, nb, c.qp, true, lane);
        lds_sync();
        load_levels_scan(pair + slot * 16, v, j);
        if (lane < 4 * nb) store_levels_scan(lvw + my_idx * 16, v, j);
        const bool nz = quad_or((v[0] | v[1] | v[2] | v[3]) != 0 ? 1 : 0) != 0;
        dequant4_row(v, q4, j);
        idct4_quad(v, lane);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] += p[i];
        if (lane < 4 * nb) *(uint32_t *)(bt + j * IT_STRIDE) = nz ? pack4_clip(v) : bp;
        const unsigned long long bal = __ballot(nz && j == 0 && lane < 4 * nb);
        for (int q = 0; q < nb; q++) if ((bal >> (4 * q)) & 1) { const int by = by0 + q; nnz4 |= 1u << blkidx_of(t - 2 * by, by); }
    }
    lds_sync();
    return nnz4;
}
template <int M>
__device__ __forceinline__ unsigned mb_encode_i8x8_trellis(const EncK &k, MbLds<M> &L, const MbCtx &c, uint32_t cz, const Q8 &q8, const TrCtx &trc, int16_t *lvw, int &cbp8)
{
    const int lane = relane(c.lane), g = lane >> 3, r8 = lane & 7;
    const bool left = c.mbx > 0, top = c.sy > 0, topright = top && c.mbx + 1 < k.mbw;
    uint8_t *tile8 = L.tile8 + IT_ORG;
    unsigned nnz8 = 0;
    cbp8 = 0;
    for (int idx = 0; idx < 4; idx++) {
        const int x8 = idx & 1, y8 = idx >> 1, avail = i8_avail(left, top, topright, idx);
        lds_sync();
        uint8_t *bt = tile8 + y8 * 8 * IT_STRIDE + x8 * 8;
        pred8_build_u(L.U8, bt, IT_STRIDE, avail, lane);
        const int src = idx * 16 + (r8 >> 2) * 8 + (r8 & 3);
        const uint32_t elo = (uint32_t)__shfl((int)cz, src), ehi = (uint32_t)__shfl((int)cz, src + 4);
        uint32_t p1lo, p1hi, p2lo, p2hi;
        pred8_row8(L.U8, L.pred8tab, g, r8, p1lo, p1hi);
        pred8_row8(L.U8, L.pred8tab, 8, r8, p2lo, p2hi);
        const int bm = L.modes8[idx * 4];
        const uint32_t plo = bm == 8 ? (uint32_t)__shfl((int)p2lo, r8) : (uint32_t)__shfl((int)p1lo, bm * 8 + r8);
        const uint32_t phi = bm == 8 ? (uint32_t)__shfl((int)p2hi, r8) : (uint32_t)__shfl((int)p1hi, bm * 8 + r8);
        int e[8], p[8], v[8];
        unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
        fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
        int mf[4], bs[4], dq[4];
        q8_row(q8, r8, mf, bs, dq);
        if (g == 0)
#pragma unroll
            for (int i = 0; i < 8; i++) lvw[idx * 64 + c_zigzag8_inv[r8 * 8 + i]] = (int16_t)v[i];
        lds_sync();
        trellis_run<5>(trc, lvw + idx * 64, 64, 1, c.qp, true, lane);
        lds_sync();
        unsigned mlo = 0, mhi = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int z = c_zigzag8_inv[r8 * 8 + i];
            v[i] = lvw[idx * 64 + z];
            if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
        }
        lds_sync();
        if (g == 0)          // the levels leave in the interleaved form (level z of the zigzag at block 4 idx + (z & 3), entry z >> 2)
#pragma unroll
            for (int i = 0; i < 8; i++) { const int z = c_zigzag8_inv[r8 * 8 + i]; lvw[(idx * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)v[i]; }
        mlo = group8_or(mlo); mhi = group8_or(mhi);
        const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
#pragma unroll
        for (int q = 0; q < 4; q++) nnz8 |= (mask & (0x1111111111111111ull << q)) ? 1u << (idx * 4 + q) : 0u;
        if (mask) cbp8 |= 1 << idx;
        const int qb = q8.qp / 6 - 6;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = dequant_one(v[i], dq[i & 3], qb);
        inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
        if (g == 0) {
            *(uint32_t *)(bt + r8 * IT_STRIDE) = pack4_clip8lo(v);
            *(uint32_t *)(bt + r8 * IT_STRIDE + 4) = pack4_clip8hi(v);
        }
    }
    lds_sync();
    nnz8 = (unsigned)__builtin_amdgcn_readfirstlane((int)nnz8);
    cbp8 = __builtin_amdgcn_readfirstlane(cbp8);
    return nnz8;
}

struct IntraRes { int satd_i16, satd_i8, satd_i4, pred16; unsigned nnz4, nnz8; int cbp8;
                  // by-products intra_rd_refine reads (x264 i_satd_i16x16_dir / i_satd_i8x8_dir): the Intra_16x16 costs in the order of the mode list,
                  // the Intra_8x8 costs (+ 4 lambda) as a lane-indexed table, lane = block * 9 + mode
                  int dir16[4], dir8v; };

// Needs the neighbour samples in L.tile / L.tile8 / L.nb and the neighbour macroblocks' edge modes in L.nmodes.
template <int M>
__device__ __forceinline__ void mb_analyse_intra(const EncK &k, MbLds<M> &L, const MbCtx &c, uint32_t cz, uint32_t t4, int parts, int i_satd_inter,
                                                 bool fast_intra, bool early_term, bool mbrd, const Q4 &q4, const Q8 &q8, IntraRes &R, const TrCtx *tra = nullptr)
{
    const bool RF2 = mbrd && ((k.rd >> 1) & 31) && (k.slice_type != X264GPU_SLICE_B || k.subme >= 9);      // x264's i_mbrd >= 2 (RD refinement, subme >= 8; a B slice analyses one level down)
    const bool every_mode = RF2 || (mbrd && !fast_intra);          // x264: i_mbrd >= 1 + b_fast_intra
    const int lane = relane(c.lane), j = lane & 3, zx = z_x0(lane), zy = z_y(lane), lambda = c.lambda;
    const bool left = c.mbx > 0, top = c.sy > 0, topright = top && c.mbx + 1 < k.mbw;
    const int sm = min(c.subme, 10);
    const int b_type_cost = k.slice_type == X264GPU_SLICE_B ? 9 * lambda : 0;      // B slices: the macroblock type prefix of an intra type (x264 i_mb_b_cost_table[I_*] = 9)
    uint8_t *tile = L.tile + IT_ORG, *tile8 = L.tile8 + IT_ORG;
    R.satd_i16 = R.satd_i8 = R.satd_i4 = MB_COST_MAX; R.pred16 = 0; R.nnz4 = R.nnz8 = 0; R.cbp8 = 0;
    R.dir16[0] = R.dir16[1] = R.dir16[2] = R.dir16[3] = MB_COST_MAX; R.dir8v = MB_COST_MAX;
    // ---- 16x16 ----
    {
        const Pred16 pp = pred16_setup(L.nb, lane);
        const int lut = sm < 3 ? 2 : sm < 5 ? 3 : 4;
        const int thresh16 = fast_intra ? (lut * i_satd_inter) >> 1 : MB_COST_MAX;
        auto cost16 = [&](int m) {
            const uint32_t pr = pred16_row4(L.nb, pp, m, zx, zy);
            const int sig = m > PRED16_P ? PRED16_DC : m;
            return wave_sum(c.satd ? satd4_half(cz, pr, lane) : sad4(cz, pr)) + lambda * bs_size_ue(sig);
        };
        if (left && top) {
            for (int m = 0; m < 3; m++) { const int cst = cost16(m); R.dir16[m] = cst; if (cst < R.satd_i16) { R.satd_i16 = cst; R.pred16 = m; } }
            if (R.satd_i16 <= thresh16) { const int cst = cost16(PRED16_P); R.dir16[3] = cst; if (cst < R.satd_i16) { R.satd_i16 = cst; R.pred16 = PRED16_P; } }
        } else {
            const int m0 = left ? PRED16_DC_LEFT : top ? PRED16_DC_TOP : PRED16_DC_128, m1 = left ? PRED16_H : PRED16_V;
            { const int cst = cost16(m0); R.dir16[0] = cst; if (cst < R.satd_i16) { R.satd_i16 = cst; R.pred16 = m0; } }
            if (left || top) { const int cst = cost16(m1); R.dir16[1] = cst; if (cst < R.satd_i16) { R.satd_i16 = cst; R.pred16 = m1; } }
        }
        R.satd_i16 += b_type_cost;
        if (R.satd_i16 > thresh16) return;
    }
    // ---- 8x8: R8 layout, lane = (mode group, row); eight modes in one pass, the ninth in a second ----
    if ((parts & 4) && k.dct8x8) {
        const int thresh = mbrd ? MB_COST_MAX : min(i_satd_inter, R.satd_i16);       // RD: every block is analysed
        const int g = lane >> 3, r8 = lane & 7;
        if (lane < 16) L.modes8[lane] = 2;
        int i_cost = lambda * 4 + b_type_cost, idx;
        for (idx = 0;; idx++) {
            const int x8 = idx & 1, y8 = idx >> 1, avail = i8_avail(left, top, topright, idx);
            lds_sync();
            const int pm = i4_pred_mode(L.nmodes, c.mbx, c.sy, idx * 4, L.modes8);
            uint8_t *bt = tile8 + y8 * 8 * IT_STRIDE + x8 * 8;
            pred8_build_u(L.U8, bt, IT_STRIDE, avail, lane);
            const int src = idx * 16 + (r8 >> 2) * 8 + (r8 & 3);
            const uint32_t elo = (uint32_t)__shfl((int)cz, src), ehi = (uint32_t)__shfl((int)cz, src + 4);
            uint32_t p1lo, p1hi, p2lo, p2hi;
            auto cost8 = [&](uint32_t plo, uint32_t phi) {
                int h;
                if (c.satd) { h = sa8d_r8_half(elo, ehi, plo, phi, lane); h += dpp<DPP_XOR1>(h); h += dpp<DPP_XOR2>(h); h += xor4(h); return (2 * h + 2) >> 2; }
                h = (int)__builtin_amdgcn_sad_u8(plo, elo, __builtin_amdgcn_sad_u8(phi, ehi, 0u));
                h += dpp<DPP_XOR1>(h); h += dpp<DPP_XOR2>(h); h += xor4(h);
                return h;
            };
            pred8_row8(L.U8, L.pred8tab, g, r8, p1lo, p1hi);
            const int c1 = cost8(p1lo, p1hi);
            pred8_row8(L.U8, L.pred8tab, 8, r8, p2lo, p2hi);
            const int c2 = cost8(p2lo, p2hi);
            int bm;
            const int best = pick_intra_mode([&](int m) { return m < 8 ? rl(c1, m * 8) : rl(c2, 0); }, avail, pm, lambda, false, every_mode, bm, RF2);
            i_cost += best + 3 * lambda;
            if (lane < 4) L.modes8[idx * 4 + lane] = (uint8_t)bm;
            if (RF2) {          // i_satd_i8x8_dir[idx][mode]: the mode's cost with the predicted-mode bonus, + 4 lambda
                const int m = lane - idx * 9;
                int raw = rl(c2, 0);
#pragma unroll
                for (int mm = 0; mm < 8; mm++) { const int v = rl(c1, mm * 8); raw = m == mm ? v : raw; }
                if (m >= 0 && m < 9) R.dir8v = raw - (pm == m ? 3 * lambda : 0) + 4 * lambda;
            }
            if (idx < 3 && i_cost > thresh) break;
            // code the block (the next ones predict from it; the last one so that the result is complete if Intra8x8 wins)
            const uint32_t plo = bm == 8 ? (uint32_t)__shfl((int)p2lo, r8) : (uint32_t)__shfl((int)p1lo, bm * 8 + r8);
            const uint32_t phi = bm == 8 ? (uint32_t)__shfl((int)p2hi, r8) : (uint32_t)__shfl((int)p1hi, bm * 8 + r8);
            int e[8], p[8], v[8];
            unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
            fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
            int mf[4], bs[4], dq[4];
            q8_row(q8, r8, mf, bs, dq);
            unsigned mlo = 0, mhi = 0;
            const bool tr8 = tra && (tra->on & TR_I8);         // --trellis 2: the block the following ones predict from is the searched one
            if (tr8) {
                if (g == 0)
#pragma unroll
                    for (int i = 0; i < 8; i++) L.lv8[idx * 64 + c_zigzag8_inv[r8 * 8 + i]] = (int16_t)v[i];
                lds_sync();
                trellis_run<5>(*tra, L.lv8 + idx * 64, 64, 1, c.qp, true, lane);
                lds_sync();
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = L.lv8[idx * 64 + c_zigzag8_inv[r8 * 8 + i]];
                lds_sync();
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (!tr8) v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
                const int z = c_zigzag8_inv[r8 * 8 + i];
                if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
                if (g == 0) L.lv8[(idx * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)v[i];
            }
            mlo = group8_or(mlo); mhi = group8_or(mhi);
            const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
#pragma unroll
            for (int q = 0; q < 4; q++) R.nnz8 |= (mask & (0x1111111111111111ull << q)) ? 1u << (idx * 4 + q) : 0u;
            if (mask) R.cbp8 |= 1 << idx;
            const int qb = q8.qp / 6 - 6;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = dequant_one(v[i], dq[i & 3], qb);
            inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
            if (g == 0) {
                *(uint32_t *)(bt + r8 * IT_STRIDE) = pack4_clip8lo(v);
                *(uint32_t *)(bt + r8 * IT_STRIDE + 4) = pack4_clip8hi(v);
            }
            if (idx == 3) break;
        }
        lds_sync();
        R.nnz8 = (unsigned)__builtin_amdgcn_readfirstlane((int)R.nnz8);
        R.cbp8 = __builtin_amdgcn_readfirstlane(R.cbp8);
        if (idx == 3) R.satd_i8 = i_cost;
        else i_cost = (i_cost * (idx == 0 ? 1024 : idx == 1 ? 512 : 341)) >> 8;
        const int thr8 = sm < 3 ? 4 : sm < 6 ? 5 : 6;
        if (early_term && min(i_cost, R.satd_i16) > (int)(((long long)i_satd_inter * thr8) >> 2)) return;
    }
    // ---- 4x4: nine modes per block in parallel (one quad of lanes per mode), blocks in coding order ----
    if (parts & 2) {
        int thresh = early_term ? min(min(i_satd_inter, R.satd_i16), R.satd_i8) : MB_COST_MAX;
        if (early_term && mbrd) thresh = (int)((long long)thresh * (fast_intra ? 9 : 10) / 8);       // RD: a little slack, the SATD order is not final
        if (lane < 16) L.modes4[lane] = 2;
        int i_cost = lambda * (24 + 16) + b_type_cost, idx;
        for (idx = 0;; idx++) {
            const int bx = z_bx(idx), by = z_by(idx);
            const int avail = i4_avail(c.mbx, c.sy, k.mbw, idx);
            lds_sync();
            const int pm = i4_pred_mode(L.nmodes, c.mbx, c.sy, idx, L.modes4);
            uint8_t *bt = tile + by * 4 * IT_STRIDE + bx * 4;
            pred4_build_u(L.U, bt, IT_STRIDE, avail, lane);
            const uint32_t pr = pred4_row4(L.U, t4);
            const uint32_t en = (uint32_t)__shfl((int)cz, idx * 4 + j);
            const int sat = quad_sum(c.satd ? satd4_half(en, pr, lane) : sad4(en, pr));
            int bm;
            const int best = pick_intra_mode([&](int m) { return rl(sat, m * 4); }, avail, pm, lambda, true, every_mode, bm);
            i_cost += best + 3 * lambda;
            if (lane == 0) L.modes4[idx] = (uint8_t)bm;
            if (idx < 15 && i_cost > thresh) break;
            const uint32_t bp = (uint32_t)__shfl((int)pr, bm * 4 + j);
            int e[4], p[4], v[4];
            unpack4(en, e); unpack4(bp, p);
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = e[t] - p[t];
            dct4_quad(v, lane);
            if (tra && (tra->on & TR_I4)) {
                if (lane < 4) store_levels_scan(L.lv4 + idx * 16, v, j);
                lds_sync();
                trellis_run<2>(*tra, L.lv4 + idx * 16, 16, 1, c.qp, true, lane);
                lds_sync();
                load_levels_scan(L.lv4 + idx * 16, v, j);
                lds_sync();
            } else quant4_row(v, q4, j);
            const bool nz = quad_or((v[0] | v[1] | v[2] | v[3]) != 0 ? 1 : 0) != 0;
            if (lane < 4) store_levels_scan(L.lv4 + idx * 16, v, j);
            dequant4_row(v, q4, j);
            idct4_quad(v, lane);
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] += p[t];
            if (lane < 4) *(uint32_t *)(bt + j * IT_STRIDE) = nz ? pack4_clip(v) : bp;
            if (__builtin_amdgcn_readfirstlane((int)nz)) R.nnz4 |= 1u << idx;
            if (idx == 15) break;
        }
        lds_sync();
        if (idx == 15) R.satd_i4 = i_cost;
    }
}

// chroma intra: mode decision (oracle analyse_intra_chroma).  Lanes 0..31 (plane = lane >> 4); needs L.cnb.
template <int M>
__device__ __forceinline__ int mb_intra_chroma_cost(const EncK &k, MbLds<M> &L, const MbCtx &c, int &predc, int *dirc = nullptr /* [4]: the candidates' costs in list order */)
{
    const int lane = relane(c.lane), pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, j = lane & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j;
    const bool left = c.mbx > 0, top = c.sy > 0;
    const uint8_t *cnb = L.cnb[pl];
    const PredC pc = predc_setup(cnb);
    const uint2 fe = *(const uint2 *)(L.csrc + cyy * 16 + 2 * cx0);
    const uint32_t cenc = nv12_pick(fe.x, fe.y, pl);
    // candidate list by availability, a nibble string: DC H V P | DC_LEFT H | DC_TOP V | DC_128
    const unsigned lst = left && top ? 0x3210u : left ? 0x14u : top ? 0x25u : 0x6u;
    const int n = left && top ? 4 : (left || top) ? 2 : 1;
    int bestc = MB_COST_MAX;
    predc = (int)(lst & 15);
    for (int i = 0; i < n; i++) {
        const int m = (int)((lst >> (4 * i)) & 15), sig = m > PREDC_P ? PREDC_DC : m;
        const uint32_t pr = predc_row4(cnb, pc, m, ci, j);
        const int hs = c.satd ? satd4_half(cenc, pr, lane) : sad4(cenc, pr);
        const int cst = wave_sum(lane < 32 ? hs : 0) + c.lambda * bs_size_ue(sig);
        if (dirc) dirc[i] = cst;
        if (cst < bestc) { bestc = cst; predc = m; }
    }
    return bestc;
}

// ------------------------------------------------------------------------------------------------
// chroma residual with x264's variance early termination (oracle encode_chroma).  Lanes 0..31: plane = lane >> 4,
// 4x4 block = (lane >> 2) & 3, row = lane & 3.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mb_chroma_residual(uint32_t enc, uint32_t pred, const Q4 &q, bool inter, bool decimate, int lane, int16_t *lv,
                                                       unsigned &nnz_bits, int &cbp_chroma, const TrCtx *tr = nullptr)
{
    const bool trellis = tr && (tr->on & TR_C);          // final encode of a trellis session: both quantisers below are the search (lv = LDS)
    lane = relane(lane);
    const int c = (lane >> 4) & 1, i = (lane >> 2) & 3, j = lane & 3;
    const bool act = lane < 32;
    int e[4], p[4], v[4];
    unpack4(enc, e); unpack4(pred, p);
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = act ? e[t] - p[t] : 0;
    // early termination (inter, dct-decimate, chroma quantiser >= 18): small residual variance -> DC only, or nothing
    bool et = false, et_drop_dc = false;
    if (inter && decimate && q.qp >= 18) {
        const int thresh = (c_lambda2_tab[q.qp] + 32) >> 6;
        const int sm = row16_sum(v[0] + v[1] + v[2] + v[3]), sq = row16_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        const int var = sq - (int)(((long long)sm * sm) >> 6);
        const int score = __builtin_amdgcn_readlane(var, 0) + __builtin_amdgcn_readlane(var, 16);
        et = score < thresh * 4;
        et_drop_dc = sq <= thresh;
    }
    dct4_quad(v, lane);
    int dcs[4];
#pragma unroll
    for (int b = 0; b < 4; b++) dcs[b] = __shfl(v[0], (lane & 48) + 4 * b);
    if (j == 0) v[0] = 0;
    if (trellis) {
        if (act) store_levels_scan(lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16, v, j);
        lds_sync();
        trellis_run<4>(*tr, lv + X264GPU_LV_CHROMA_AC, 16, 8, q.qp, !inter, lane);
        lds_sync();
        if (act) load_levels_scan(lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16, v, j); else v[0] = v[1] = v[2] = v[3] = 0;
    } else quant4_row(v, q, j);
    unsigned mask = quad_or((int)scan_mask(v, j));
    int big = quad_or(any_big(v) ? 1 : 0);
    bool nz = mask != 0;
    int score = 0;
    if (inter && decimate) {
        int s = nz ? (big ? 9 : decimate_from_mask(mask, 1)) : 0;
        score = row16_sum(j == 0 ? s : 0);
    }
    bool plane_ac = row16_or(nz ? 1 : 0) != 0;
    if (plane_ac && inter && decimate && score < 7) plane_ac = false;
    if (et) plane_ac = false;
    int f[4];
    { int a = dcs[0] + dcs[1], b = dcs[0] - dcs[1], cc = dcs[2] + dcs[3], d = dcs[2] - dcs[3];
      f[0] = a + cc; f[1] = b + d; f[2] = a - cc; f[3] = b - d; }
    int ldc[4], nzdc = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) { ldc[b] = quant_one(f[b], q.mf[0] >> 1, q.bias[0] << 1); nzdc |= ldc[b]; }
    if (trellis) {
        lds_sync();
        if (act && i == 0 && j == 0)
#pragma unroll
            for (int b = 0; b < 4; b++) lv[X264GPU_LV_CHROMA_DC + c * 4 + b] = (int16_t)f[b];
        lds_sync();
        trellis_run<3>(*tr, lv + X264GPU_LV_CHROMA_DC, 4, 2, q.qp, !inter, lane);
        lds_sync();
        nzdc = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { ldc[b] = lv[X264GPU_LV_CHROMA_DC + c * 4 + b]; nzdc |= ldc[b]; }
        lds_sync();
    }
    if (et && et_drop_dc) { ldc[0] = ldc[1] = ldc[2] = ldc[3] = 0; nzdc = 0; }
    if (nzdc && !plane_ac) {
        const int dmf = q.dq[0] << (q.qp / 6);
        if (dmf <= 32 * 64) {
            auto rnd = [&](int o[4]) {
                const int d0 = ldc[0] + ldc[1], d1 = ldc[2] + ldc[3], d2 = ldc[0] - ldc[1], d3 = ldc[2] - ldc[3];
                o[0] = ((d0 + d1) * dmf >> 5) + 32; o[1] = ((d0 - d1) * dmf >> 5) + 32;
                o[2] = ((d2 + d3) * dmf >> 5) + 32; o[3] = ((d2 - d3) * dmf >> 5) + 32;
            };
            int ref[4], out[4];
            rnd(ref);
            if (!((ref[0] | ref[1] | ref[2] | ref[3]) >> 6)) { ldc[0] = ldc[1] = ldc[2] = ldc[3] = 0; nzdc = 0; }
            else {
                int left = 0;
#define X264GPU_OPT_DC(C) { int level = ldc[C]; const int sign = level >> 31 | 1; \
                    while (level) { ldc[C] = level - sign; rnd(out); \
                        if (((ref[0] ^ out[0]) | (ref[1] ^ out[1]) | (ref[2] ^ out[2]) | (ref[3] ^ out[3])) >> 6) { left = 1; ldc[C] = level; break; } \
                        level -= sign; } }
                X264GPU_OPT_DC(3) X264GPU_OPT_DC(1) X264GPU_OPT_DC(2) X264GPU_OPT_DC(0)
#undef X264GPU_OPT_DC
                if (!left) { ldc[0] = ldc[1] = ldc[2] = ldc[3] = 0; nzdc = 0; }
                else nzdc = 1;
            }
        }
    }
    int dq[4] = { 0, 0, 0, 0 };
    if (nzdc) {
        int a = ldc[0] + ldc[1], b = ldc[0] - ldc[1], cc = ldc[2] + ldc[3], d = ldc[2] - ldc[3];
        int g[4] = { a + cc, b + d, a - cc, b - d };
        int ls = q.dq[0] << (q.qp / 6);
#pragma unroll
        for (int b2 = 0; b2 < 4; b2++) dq[b2] = (g[b2] * ls) >> 5;
    }
    const bool keep = plane_ac && nz;
    if (act) {
        int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16;
        int z[4] = { 0, 0, 0, 0 };
        store_levels_scan(l, keep ? v : z, j);
        if (i == 0 && j == 0)
#pragma unroll
            for (int b = 0; b < 4; b++) lv[X264GPU_LV_CHROMA_DC + c * 4 + b] = (int16_t)ldc[b];
    }
    if (keep) dequant4_row(v, q, j);
    else { v[0] = v[1] = v[2] = v[3] = 0; }
    if (j == 0) v[0] = dq[i];
    idct4_quad(v, lane);
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] += p[t];
    unsigned long long bal = __ballot(act && keep && j == 0);
    unsigned long long bdc = __ballot(act && nzdc != 0 && i == 0 && j == 0);
    unsigned bits = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) bits |= (unsigned)((bal >> (4 * b)) & 1) << (16 + b);
    bits |= (unsigned)(bdc & 1) << 25;
    bits |= (unsigned)((bdc >> 16) & 1) << 26;
    nnz_bits |= bits;
    cbp_chroma = (bits & 0x00ff0000u) ? 2 : (bits & 0x06000000u) ? 1 : 0;
    return pack4_clip(v);
}

// store the reconstructed chroma rows of lanes 0..31 (U lanes 0..15, V lanes 16..31) as NV12
__device__ __forceinline__ void mb_store_chroma(uint8_t *ruv, int rs, int lane, uint32_t crec)
{
    const int ci = (lane >> 2) & 3, j = lane & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j;
    const uint32_t other = (uint32_t)__shfl_xor((int)crec, 16);
    if (lane < 16) {
        const uint32_t u = crec, w = other;
        uint2 o;
        o.x = (u & 0xff) | ((w & 0xff) << 8) | ((u & 0xff00) << 8) | ((w & 0xff00) << 16);
        o.y = ((u >> 16) & 0xff) | (((w >> 16) & 0xff) << 8) | ((u >> 24) << 16) | ((w >> 24) << 24);
        *(uint2 *)(ruv + (size_t)cyy * rs + 2 * cx0) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// x264_macroblock_probe_skip_internal (oracle probe_skip_pred): would the residual of this prediction code to nothing?  pred: this lane's luma row
// (Z layout); cenc / cpred: lanes 0..31 = chroma plane (lane >> 4) & 1, 4x4 block (lane >> 2) & 3, row lane & 3
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool mb_probe_skip_pred(const MbCtx &c, uint32_t cz, uint32_t pred, uint32_t cenc, uint32_t cpred, const Q4 &ql, const Q4 &qc)
{
    const int lane = relane(c.lane), j = lane & 3;
    bool ok;
    {
        int e[4], p[4], v[4];
        unpack4(cz, e); unpack4(pred, p);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = e[i] - p[i];
        dct4_quad(v, lane);
        quant4_row(v, ql, j);
        const unsigned mask = (unsigned)quad_or((int)scan_mask(v, j));
        const int big = quad_or(any_big(v) ? 1 : 0);
        const int sc = mask ? (big ? 9 : decimate_from_mask(mask, 0)) : 0;
        ok = wave_sum(j == 0 ? sc : 0) < 6;
    }
    {
        const bool act = lane < 32;
        int e[4], p[4], v[4];
        unpack4(cenc, e); unpack4(cpred, p);
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = act ? e[t] - p[t] : 0;
        const int thresh = (c_lambda2_tab[qc.qp] + 32) >> 6;
        const int ssd = row16_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        dct4_quad(v, lane);
        int dcs[4];
#pragma unroll
        for (int b = 0; b < 4; b++) dcs[b] = __shfl(v[0], (lane & 48) + 4 * b);
        int f[4];
        { int a = dcs[0] + dcs[1], b = dcs[0] - dcs[1], cc = dcs[2] + dcs[3], d = dcs[2] - dcs[3];
          f[0] = a + cc; f[1] = b + d; f[2] = a - cc; f[3] = b - d; }
        int nzdc = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) nzdc |= quant_one(f[b], qc.mf[0] >> 1, qc.bias[0] << 1);
        if (j == 0) v[0] = 0;
        quant4_row(v, qc, j);
        const unsigned mask = (unsigned)quad_or((int)scan_mask(v, j));
        const int big = quad_or(any_big(v) ? 1 : 0);
        const int sc = mask ? (big ? 9 : decimate_from_mask(mask, 1)) : 0;
        const int score = row16_sum(j == 0 ? sc : 0);
        const bool fail = ssd >= thresh && (nzdc != 0 || (ssd >= thresh * 4 && score >= 7));
        ok = ok && !__builtin_amdgcn_readlane((int)fail, 0) && !__builtin_amdgcn_readlane((int)fail, 16);
    }
    return ok;
}

// x264_macroblock_probe_pskip (oracle probe_pskip): would the macroblock code to nothing at the skip vector?
__device__ __forceinline__ bool mb_probe_pskip(const EncK &k, const MbCtx &c, uint32_t cz, int pmx, int pmy, const Q4 &ql, const Q4 &qc)
{
    const int lane = relane(c.lane), j = lane & 3, zx = z_x0(lane), zy = z_y(lane);
    const int mvx = clampi(pmx, c.mvmin0, c.mvmax0), mvy = clampi(pmy, c.mvmin1, c.mvmax1);
    uint32_t pred = mc_luma_row4(ref_plane00(k, c.s, 0), k.plane_bytes, k.rs, c.px + zx, c.py + zy, mvx, mvy);
    if (k.wp_any && (k.wl0[0] >> 24)) pred = wp4(pred, k.wl0[0]);
    const int pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j;
    uint32_t pu, pv;
    mc_chroma_row4(ref_chroma00(k, c.s, 0), k.rs, c.mbx * 8 + cx0, c.mby * 8 + cyy, mvx, mvy, pu, pv);
    if (k.wc_any) { if (k.wc0[0] >> 24) pu = wp4(pu, k.wc0[0]); if (k.wc0[1] >> 24) pv = wp4(pv, k.wc0[1]); }
    const uint2 fe = *(const uint2 *)(c.fuv + (size_t)cyy * k.fs + 2 * cx0);
    return mb_probe_skip_pred(c, cz, pred, nv12_pick(fe.x, fe.y, pl), pl ? pv : pu, ql, qc);
}

// ------------------------------------------------------------------------------------------------
// B slices: how a macroblock is predicted, as per-lane values keyed by the lane's 8x8 block (lane >> 4 in the Z layout): reference index
// and vector in list 0 and in list 1, reference -1 = the list is not used.  Block b's values are read from lane 16 b.
// ------------------------------------------------------------------------------------------------
struct BCfg { int r0, x0, y0, r1, x1, y1; };
// x264_me_refine_bidir's dia4d: every offset in up to two of the four components (list-0 x, y, list-1 x, y), one byte each
#define D4(a, b, c, d) ((uint32_t)(uint8_t)(int8_t)(a) | ((uint32_t)(uint8_t)(int8_t)(b) << 8) | ((uint32_t)(uint8_t)(int8_t)(c) << 16) | ((uint32_t)(uint8_t)(int8_t)(d) << 24))
static __constant__ const uint32_t c_dia4d[33] = {
    D4(0, 0, 0, 0),
    D4(0, 0, 0, 1), D4(0, 0, 0, -1), D4(0, 0, 1, 0), D4(0, 0, -1, 0), D4(0, 1, 0, 0), D4(0, -1, 0, 0), D4(1, 0, 0, 0), D4(-1, 0, 0, 0),
    D4(0, 0, 1, 1), D4(0, 0, -1, -1), D4(0, 1, 1, 0), D4(0, -1, -1, 0), D4(1, 1, 0, 0), D4(-1, -1, 0, 0), D4(1, 0, 0, 1), D4(-1, 0, 0, -1),
    D4(0, 1, 0, 1), D4(0, -1, 0, -1), D4(1, 0, 1, 0), D4(-1, 0, -1, 0), D4(0, 0, -1, 1), D4(0, 0, 1, -1), D4(0, -1, 1, 0), D4(0, 1, -1, 0),
    D4(-1, 1, 0, 0), D4(1, -1, 0, 0), D4(1, 0, 0, -1), D4(-1, 0, 0, 1), D4(0, -1, 0, 1), D4(0, 1, 0, -1), D4(-1, 0, 1, 0), D4(1, 0, -1, 0) };
#undef D4

// x264_mb_mc of a B macroblock for this lane's row of four luma samples (Z layout) and, in lanes 0..31, its row of the chroma 4x4 block
// (lane >> 2) & 3 of plane (lane >> 4) & 1: from list 0, list 1, or both averaged with the pair's implicit weight (biwv: lane r0 * 4 + r1)
__device__ __forceinline__ void b_predict(const EncK &k, const MbCtx &c, const BCfg &g0, int biwv, uint32_t &pred, uint32_t &cpred)
{
    const int lane = relane(c.lane), zx = z_x0(lane), zy = z_y(lane), j4 = lane & 3;
    // x264 mb_mc_*xywh: vectors are clipped to the macroblock's mv_min / mv_max before the fetch — a spatial-direct vector is a neighbour's, taken as it
    // is, and can point farther than the padding reaches (same samples inside the replicated border)
    BCfg g = g0;
    g.x0 = clampi(g.x0, c.mvmin0, c.mvmax0); g.y0 = clampi(g.y0, c.mvmin1, c.mvmax1); g.x1 = clampi(g.x1, c.mvmin0, c.mvmax0); g.y1 = clampi(g.y1, c.mvmin1, c.mvmax1);
    {
        // (both lists are fetched whether or not the block uses them — an unused list reads reference 0 at its clipped vector and is dropped: four
        //  loads in flight together instead of two latencies one after the other behind per-lane branches)
        const uint32_t p0 = mc_luma_row4(ref_plane00(k, c.s, max(g.r0, 0)), k.plane_bytes, k.rs, c.px + zx, c.py + zy, g.x0, g.y0);
        const uint32_t p1 = mc_luma_row4(ref_plane00(k, c.s, k.nref + max(g.r1, 0)), k.plane_bytes, k.rs, c.px + zx, c.py + zy, g.x1, g.y1);
        const int w = __shfl(biwv, max(g.r0, 0) * 4 + max(g.r1, 0));
        pred = g.r0 >= 0 ? (g.r1 >= 0 ? avg_weight4_u8(p0, p1, w) : p0) : p1;
    }
    {
        const int pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j4;
        const int r0 = __shfl(g.r0, ci * 16), x0 = __shfl(g.x0, ci * 16), y0 = __shfl(g.y0, ci * 16);
        const int r1 = __shfl(g.r1, ci * 16), x1 = __shfl(g.x1, ci * 16), y1 = __shfl(g.y1, ci * 16);
        uint32_t u0 = 0, v0 = 0, u1 = 0, v1 = 0;
        mc_chroma_row4(ref_chroma00(k, c.s, max(r0, 0)), k.rs, c.mbx * 8 + cx0, c.mby * 8 + cyy, x0, y0, u0, v0);
        mc_chroma_row4(ref_chroma00(k, c.s, k.nref + max(r1, 0)), k.rs, c.mbx * 8 + cx0, c.mby * 8 + cyy, x1, y1, u1, v1);
        const int w = __shfl(biwv, max(r0, 0) * 4 + max(r1, 0));
        const uint32_t a = pl ? v0 : u0, b = pl ? v1 : u1;
        cpred = r0 >= 0 ? (r1 >= 0 ? avg_weight4_u8(a, b, w) : a) : b;
    }
}

// ------------------------------------------------------------------------------------------------
// The slice kernel: one wavefront per stream walks the macroblocks in raster order.
// ------------------------------------------------------------------------------------------------
// M: sub-pel neighbourhood margin (2 px up to subme 7, 5 above); ME: --me method (its own instantiation each: the roaming umh / esa code
// costs the hexagon kernel registers otherwise); PS: P slice (I slices: k_mb_slice<2, 1, false>, mb_slice_intra.hip); RD: x264's RD mode
// decision of subme 6 / 7 with CAVLC bit counts (own instantiations: the candidate passes cost registers and code)
#ifndef MB_WAVES_PER_EU
#define MB_WAVES_PER_EU 2
#endif
// RD: 0 = SATD decisions (subme <= 5); 1 = RD mode decision with CAVLC bit counts (rd.hip.h); 2 = with CABAC context states and sizes (cabac_rd.hip.h);
// 3 = 2 + the trellis quantiser in the final encode (trellis.hip.h) — an instantiation of its own: the search's registers would cost the others spills;
// 4 = 3 + the search in the intra analysis' block encodes and in every RD candidate (x264 --trellis 2);
// 5 / 6 = 3 / 4 + RD refinement of the chosen type's vectors and intra modes (x264 subme >= 8: k_mb_refine.inc)
// BS: B slice (PS is set as well: an inter slice) — RD instantiations with CABAC only; its own analysis and candidate order (k_mb_b.inc)
template <int M, int ME, bool PS, int RD = 0, bool BS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MB_WAVES_PER_EU, 4))) void k_mb_slice(EncK k)
{
    static_assert(!BS || PS, "B slices are inter slices");
    static_assert(RD != 7 || BS, "RD 7: B slices only");
    __shared__ __attribute__((aligned(16))) MbLds<M> L;
    // x264_me_refine_bidir: bit set of the vector quadruples already costed (4096 bits).  It lives in the chroma sub-pel staging area, which only
    // holds data DURING a search: a buffer of its own made the B instantiations' LDS 20 992 B a wavefront — seven instead of eight wavefronts a CU,
    // so the last 256 of 2048 streams ran as a second round
    static_assert(!BS || CSubGeo<M>::DWORDS >= 128, "b_visited aliases L.csub");
#define b_visited L.csub      /* (not a pointer variable: a generic pointer into LDS trips an instruction-selection bug of this compiler in the umh instantiation) */
    const int lane0 = threadIdx.x, s = k.perm ? uni(k.perm[blockIdx.x]) : (int)blockIdx.x;
    int lane = lane0;
    const unsigned long long wt0 = k.wtime ? __builtin_readcyclecounter() : 0ull;
#ifdef X264GPU_POISON
    // poison builds (tools/poison_check.sh): whatever the last wavefront left in LDS must not matter — start from a pattern that no test leaves there
    for (int i = lane; i < (int)(sizeof(L) / 4); i += 64) ((uint32_t *)&L)[i] = 0xCDCDCDCDu;
    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
#endif
    // x264 slice threads: blockIdx.y = slice of the picture, macroblock rows [row0, row1) (k.slices == 1: the whole picture)
    const int nsl = k.slices > 1 ? k.slices : 1, row0 = (k.mbh * (int)blockIdx.y + nsl / 2) / nsl, row1 = (k.mbh * ((int)blockIdx.y + 1) + nsl / 2) / nsl;
    const int mb_first = row0 * k.mbw, mb_end = row1 * k.mbw;
    // --slices N in P pictures: the intra count of the slices before this one as assumed for this pass, and the window of counts that leave
    // every fast-intra decision of the slice as taken (EncK.sl_stat)
    int intra_prior = 0;
    if (PS && k.sl_stat) {
        const size_t si = (size_t)s * nsl + blockIdx.y;
        if (k.sl_pass && !k.sl_rerun[si]) return;
        intra_prior = uni(k.sl_stat[si * 4 + 3]);
        if (lane == 0) { L.slw[0] = 0; L.slw[1] = 0x7fffffff; L.slw[2] = intra_prior; }
    }
    const uint32_t t4 = (lane >> 2) < 9 ? ((const uint32_t *)c_pred4_table.t)[lane] : 0x01010101u * U_DC;
    for (int i = lane; i < 144; i += 64) ((uint32_t *)L.pred8tab)[i] = ((const uint32_t *)c_pred8_table)[i];
    lds_sync();
    x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    constexpr bool pslice = PS;
    constexpr bool TRL = RD >= 3;               // trellis sites compiled in
    constexpr bool TRL2 = RD == 4 || RD == 6;   // --trellis 2: the search also inside the intra analysis and in every RD candidate                 // I slices run their own instantiation (no search code, a fraction of the registers)
    constexpr bool NORD = RD == 0 || RD == 7;   // B slices analysed without RD (x264 below --subme 7); RD 7 = that + the slice's CABAC state and the trellis quantiser in the final encode (--subme 6 --trellis 1 / 2)
    constexpr bool REF = RD == 5 || RD == 6;               // RD refinement of the chosen type (x264 subme >= 8, i_mbrd 2; k_mb_refine.inc): RD 5 = 3 + refinement, 6 = 4 + refinement
    int j4 = lane & 3, zx = z_x0(lane), zy = z_y(lane);
#define RELANE() do { lane = relane(lane0); c.lane = lane; j4 = lane & 3; zx = z_x0(lane); zy = z_y(lane); } while (0)
    int intra_count = intra_prior, cost_qp = -1;          // intra macroblocks so far: of the slice (slice threads), of the picture (--slices N)
    // RD instantiation: levels of the candidate being costed (and of the final macroblock, before they go out), total_coeff of the left / top
    // macroblocks' blocks for the nC of the bit counts; the quantiser the previous coded macroblock left (mb_qp_delta bits)
    __shared__ __attribute__((aligned(16))) int16_t rd_lvs[RD ? X264GPU_MB_LEVELS : 1];
    __shared__ uint8_t rd_ntc[2][RD == 1 ? 24 : 1];
#ifdef X264GPU_POISON
    for (int i = lane; i < (RD ? X264GPU_MB_LEVELS : 1); i += 64) rd_lvs[i] = (int16_t)0xCDCD;
    __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier();
#endif
    // x264_slice_write: the slice's quantiser (header, context initialisation, start of the mb_qp_delta chain) is its FIRST macroblock's
    int last_qp = uni((int)k.mbqp[(size_t)s * k.nmb + mb_first]);
    // CABAC RD: the slice's context variables (two registers, see cabac_rd.hip.h), the probability model, the previous macroblock's mb_qp_delta
    Cab cab = { 0, 0, 0, 0, 0 };
    uint32_t cab_modelv = 0;
    int last_dqp = 0;
    if constexpr (RD >= 2) { cab_init(cab, lane, pslice, last_qp); cab_modelv = cab_model(lane); }
    Prof pf;
    pf.start();
    int biwv = 32;                  // B: lane r0 * 4 + r1 = implicit weight of the list-0 sample of that reference pair
    if constexpr (BS) {
#pragma unroll
        for (int r0 = 0; r0 < 5; r0++)
#pragma unroll
            for (int r1 = 0; r1 < 4; r1++) biwv = lane == r0 * 4 + r1 ? (int)k.biw[r0][r1] : biwv;
    }
    int16_t *mv16 = k.mv16_cur + (size_t)s * k.nmb * 2;
    uint8_t *mbtype_cur = k.mbtype_cur + (size_t)s * k.nmb;

    int ds_t = 0, ds_s = 0;          // --direct auto: this slice's skip-probe counts of the temporal / spatial mode (stored, not added, at the slice's end: a repeated slice pass of --slices N replaces its own)
    for (int mbi = mb_first; mbi < mb_end; mbi++) {
        MbCtx c;
        RELANE();
        c.s = s; c.mbi = mbi; c.mbx = mbi % k.mbw; c.mby = mbi / k.mbw; c.sy = c.mby - row0; c.px = c.mbx * 16; c.py = c.mby * 16;
        c.fenc = k.fenc_y + (size_t)s * k.fency_bytes + (size_t)c.py * k.fs + c.px;
        c.fuv = k.fenc_uv + (size_t)s * k.fencuv_bytes + (size_t)c.mby * 8 * k.fs + c.px;
        c.qp = __builtin_amdgcn_readfirstlane((int)k.mbqp[(size_t)s * k.nmb + mbi]);
        // (the state carried from macroblock to macroblock is wave-uniform; said explicitly, because one value of it the compiler takes for divergent makes the
        //  quantiser, lambda and with them every decision of the next macroblock a vector value and the state machine below divergent control flow)
        last_qp = uni(last_qp); last_dqp = uni(last_dqp); intra_count = uni(intra_count); cost_qp = uni(cost_qp); ds_t = uni(ds_t); ds_s = uni(ds_s);
        if (k.qp_snap && abs(c.qp - last_qp) == 1) c.qp = last_qp;      // x264_macroblock_analyse under AQ: within 1 of the previous macroblock's quantiser = that quantiser
        c.qpc = (int)d_chroma_qp_table[min(max(c.qp + k.chroma_qp_offset, 0), 51)];
        c.lambda = k.lambda_tab[c.qp];
        c.subme = min(max(k.subme, 0), 11);
        if (BS && (c.subme == 6 || c.subme == 8)) c.subme--;          // x264_macroblock_thread_init: B slices analyse one sub-pel level down (8 -> 7: no RD refinement there below subme 9)
        c.satd = c.subme > 1; c.chroma_me = pslice && k.chroma_me && (BS ? REF && c.subme >= 9 : c.subme >= 5);      // (x264_macroblock_thread_init: B slices carry chroma in the sub-pel costs from subme 9 up)
        c.cost_base = k.cost_all + (size_t)c.qp * 2 * MVCOST_HALF;
        if (pslice && c.qp != cost_qp) {          // the mv-cost table of this quantiser (symmetric: non-negative differences only) into LDS
            cost_qp = c.qp;
            lds_sync();
            const uint32_t *src = (const uint32_t *)(c.cost_base + MVCOST_HALF);
            for (int i = lane; i < MVC_N / 2; i += 64) ((uint32_t *)L.mvcost)[i] = src[i];
            lds_sync();
        }
        c.nref = k.nref;
        const Q4 &q_li = k.q4tab[c.qp * 4 + 0], &q_lp = k.q4tab[c.qp * 4 + 1], &q_ci = k.q4tab[c.qpc * 4 + 2], &q_cp = k.q4tab[c.qpc * 4 + 3];
        const Q8 &q8i = k.q8tab[c.qp * 2 + 0], &q8p = k.q8tab[c.qp * 2 + 1];
        const bool left = c.mbx > 0, top = c.sy > 0, topright = top && c.mbx + 1 < k.mbw, topleft = top && left;
        const int mbx = c.mbx, mby = c.mby;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        auto intra_t = [](int t) { return t >= 0 && t <= 3; };
        uint8_t *rec = rec_plane00(k, s) + (size_t)c.py * k.rs + c.px;
        uint8_t *ruv = rec_chroma00(k, s) + (size_t)(mby * 8) * k.rs + c.px;
        int16_t *lv = k.levels + ((size_t)s * k.nmb + mbi) * X264GPU_MB_LEVELS;
        uint8_t *tile = L.tile + IT_ORG, *tile8 = L.tile8 + IT_ORG;

        // ---- everything the macroblock reads of its neighbours, in ONE trip to memory: all requests first (no load depends on another),
        //      the LDS stores after.  Reconstruction row -1 / column -1, chroma ring, neighbour types, edge modes, motion of the 8x8 blocks
        //      around the macroblock, the source samples ----
        uint8_t v_y = 128, v_c = 128;
        int n_type = -1, n_mode = 2, nb_type = -1, mc_type = -1, mc_ref = 0, mc_vx = 0, mc_vy = 0;
        if (lane < 25) {
            const int x = lane - 1;
            if (top && (x >= 0 || left) && (x < 16 || topright)) v_y = rec[-(long)k.rs + x];
        } else if (lane >= 32 && lane < 48) {
            if (left) v_y = rec[(long)(lane - 32) * k.rs - 1];
        } else if (lane >= 48 && lane < 56) {          // edge modes of the left / top macroblocks
            const int i = lane - 48;
            const x264gpu_mb *n = i < 4 ? (left ? mbs + mbi - 1 : nullptr) : (top ? mbs + mbi - k.mbw : nullptr);
            if (n) { n_type = n->type; n_mode = n->i4_mode[i < 4 ? blkidx_of(3, i) : blkidx_of(i - 4, 3)]; }
        } else if (lane >= 56 && lane < 60) {          // types of the left / top / top-left / top-right macroblocks
            const int q = lane - 56;
            const int nbi = q == 0 ? (left ? mbi - 1 : -1) : q == 1 ? (top ? mbi - k.mbw : -1) : q == 2 ? (topleft ? mbi - k.mbw - 1 : -1) : (topright ? mbi - k.mbw + 1 : -1);
            if (nbi >= 0) nb_type = mbs[nbi].type;
        }
        {
            const int t = lane & 15, pl = (lane >> 4) & 1;
            if (lane < 32) { if (t < 9) { const int x = t - 1; if (top && (x >= 0 || left)) v_c = ruv[-(long)k.rs + 2 * x + pl]; } }
            else if (lane < 48) { if (left) v_c = ruv[(long)(t & 7) * k.rs - 2 + ((t >> 3) & 1)]; }
        }
        int mc_dir = 0;
        if (BS ? (lane < 32 && (lane & 15) < 12) : (lane < 12 && pslice)) {                     // motion cache grid: x = -1..2, y = -1..1 (B: list 1 sixteen lanes up)
            const int gl = lane & 15, gx = (gl & 3) - 1, gy = (gl >> 2) - 1;
            int nbi = -1, blk = 0;
            if (gy < 0) {
                if (gx < 0) { if (topleft) { nbi = mbi - k.mbw - 1; blk = 3; } }
                else if (gx < 2) { if (top) { nbi = mbi - k.mbw; blk = 2 + gx; } }
                else if (topright) { nbi = mbi - k.mbw + 1; blk = 2; }
            } else if (gx < 0 && left) { nbi = mbi - 1; blk = 1 + 2 * gy; }
            if (nbi >= 0) {
                const x264gpu_mb *n = mbs + nbi; mc_type = n->type;
                if (BS && lane >= 16) { mc_ref = n->ref1[blk]; mc_vx = n->mv1[blk][0]; mc_vy = n->mv1[blk][1]; }
                else { mc_ref = n->ref[blk]; mc_vx = n->mv[blk][0]; mc_vy = n->mv[blk][1]; }
                if constexpr (BS) mc_dir = mc_type == X264GPU_MB_B_DIRECT || mc_type == X264GPU_MB_B_SKIP || (mc_type == X264GPU_MB_B_8x8 && ((n->direct8 >> blk) & 1));
            }
        }
        const uint32_t cz = *(const uint32_t *)(c.fenc + (size_t)zy * k.fs + zx);
        uint32_t csv = 0;
        if (lane < 32) csv = *(const uint32_t *)(c.fuv + (size_t)(lane >> 2) * k.fs + (lane & 3) * 4);
        uint8_t ntcv = 0;            // RD: total_coeff of the left (lanes 0..23) / top (lanes 24..47) macroblock's blocks
        uint32_t cnbv = 0;           // CABAC RD: lanes 0..3 / 4..7 = dwords 0, 1, 6, 11 of the left / top record, lanes 8, 9 / 10, 11 = their |mvd| bytes
        if constexpr (RD >= 2) {
            if (lane < 8) { const bool av = lane < 4 ? left : top; static_assert(sizeof(x264gpu_mb) == 64, "record"); const int dw = (lane & 3) == 0 ? 0 : (lane & 3) == 1 ? 1 : (lane & 3) == 2 ? 6 : 11;
                            if (av) cnbv = ((const uint32_t *)(mbs + (lane < 4 ? mbi - 1 : mbi - k.mbw)))[dw]; }
            else if (lane < (BS ? 16 : 12)) {           // (B: 16 bytes a macroblock, list 1's |mvd| behind list 0's — lanes 12..15)
                const bool lf = (lane & 3) < 2, av = lf ? left : top;
                if (av) cnbv = ((const uint32_t *)(k.amvd + ((size_t)s * k.nmb + (lf ? mbi - 1 : mbi - k.mbw)) * (BS ? 16 : 8)))[(lane & 1) + (lane >= 12 ? 2 : 0)];
            }
        }
        if constexpr (RD == 1) {
            if (k.rd) {
                if (lane < 24) { if (left) ntcv = k.tc[((size_t)s * k.nmb + mbi - 1) * 24 + lane]; }
                else if (lane < 48) { if (top) ntcv = k.tc[((size_t)s * k.nmb + mbi - k.mbw) * 24 + lane - 24]; }
            }
        }
        // ---- ... into LDS / registers ----
        lds_sync();
        if (lane < 25) {
            const int x = lane - 1;
            tile[-IT_STRIDE + x] = v_y; tile8[-IT_STRIDE + x] = v_y;
            if (lane < 21) L.nb[NB_TOP + x] = v_y;
        } else if (lane >= 32 && lane < 48) {
            const int y = lane - 32;
            tile[y * IT_STRIDE - 1] = v_y; tile8[y * IT_STRIDE - 1] = v_y; L.nb[NB_LEFT + y] = v_y;
        } else if (lane >= 48 && lane < 56) {
            L.nmodes[lane - 48] = (uint8_t)((n_type == X264GPU_MB_I4x4 || n_type == X264GPU_MB_I8x8) ? n_mode : 2);
        }
        {
            const int t = lane & 15, pl = (lane >> 4) & 1;
            if (lane < 32) { if (t < 9) L.cnb[pl][CNB_TOP + t - 1] = v_c; }
            else if (lane < 48) L.cnb[(t >> 3) & 1][CNB_LEFT + (t & 7)] = v_c;
        }
        const int type_left = rl(nb_type, 56), type_top = rl(nb_type, 57), type_tl = rl(nb_type, 58), type_tr = rl(nb_type, 59);
        // motion cache: neighbours' references / vectors at 8x8 granularity; this macroblock's blocks start unavailable
        MeState S = {};
        if (BS ? (lane < 32 && (lane & 15) < 12) : lane < 12) {
            int ref = -2, vx = 0, vy = 0;
            if (mc_type >= 0) { if (mc_type <= 3) ref = -1; else { ref = mc_ref; vx = mc_vx; vy = mc_vy; } }
            if (BS && ref < 0) { vx = 0; vy = 0; if (ref != -2) ref = -1; }         // a list the block does not use: reference -1, zero vector
            S.cref = ref; S.cmvx = vx; S.cmvy = vy; S.cdir = mc_dir;
        }
        *(uint32_t *)(L.src + zy * 16 + zx) = cz;
        if constexpr (RD == 1) { if (lane < 48) rd_ntc[lane >= 24][lane >= 24 ? lane - 24 : lane] = ntcv; }
        if (lane < 32) *(uint32_t *)(L.csrc + (lane >> 2) * 16 + (lane & 3) * 4) = csv;
        lds_sync();

        x264gpu_mb &recd = L.rec;
        if (lane < 16) ((uint32_t *)&L.rec)[lane] = 0;
        lds_sync();
        if (lane == 0) recd.qp = (uint8_t)c.qp;
        const int parts = k.partitions;
        const bool early_term = c.subme < 11;
        int mb_type = X264GPU_MB_I16x16, i_cost = 0, predc = 0, satd_chroma = MB_COST_MAX;
        IntraRes IR;
        int rf_dirc[4] = { MB_COST_MAX, MB_COST_MAX, MB_COST_MAX, MB_COST_MAX };          // i_satd_chroma_dir in list order (RD refinement)
        bool pskip = false;
        int cost8x8 = MB_COST_MAX, satd16x8 = MB_COST_MAX, satd8x16 = MB_COST_MAX;      // SATD costs of the shapes (MB_COST_MAX: not analysed / terminated early)
        const bool rdon = RD && k.rd != 0;
        int i_inter_satd = MB_COST_MAX;            // best SATD cost of the inter analysis (the RD thresholds hang on it)
        int pskx = 0, psky = 0, best_part = D_16x16;
        bool fast_intra = false;
        pf.mark(PH_SETUP);

        if (pslice) {
            // ---- limits ----
            const int fr = 4 * (k.mv_range > 0 ? k.mv_range : 512);
            c.mvmin0 = 4 * (-16 * mbx - 24); c.mvmax0 = 4 * (16 * (k.mbw - mbx - 1) + 24);
            c.mvmin1 = 4 * (-16 * mby - 24); c.mvmax1 = 4 * (16 * (k.mbh - mby - 1) + 24);
            c.smin0 = clampi(c.mvmin0, -fr, fr - 1); c.smax0 = clampi(c.mvmax0, -fr, fr - 1);
            c.smin1 = clampi(c.mvmin1, -fr, fr - 1); c.smax1 = clampi(c.mvmax1, -fr, fr - 1);
            c.fmin0 = (c.smin0 >> 2) + 6; c.fmax0 = (c.smax0 >> 2) - 6; c.fmin1 = (c.smin1 >> 2) + 6; c.fmax1 = (c.smax1 >> 2) - 6;
            // ---- fast intra decision, skip vector, fast skip ----
            if (early_term && mbi - mb_first > 4) {
                const int colo = BS ? -1 : uni((int)k.mbtype_ref0[(size_t)s * k.nmb + mbi]);      // the co-located type counts in P slices only
                const bool near_intra = intra_t(type_left) || intra_t(type_top) || intra_t(type_tl) || intra_t(type_tr) || intra_t(colo);
                fast_intra = !(c.subme > 2 && (near_intra || mbi - mb_first < 3 * intra_count));      // (x264: always fast-intra below subme 3)
                if (k.sl_stat && c.subme > 2 && !near_intra && lane == 0) {
                    const int prior = L.slw[2], need = (mbi - mb_first) / 3 + 1 - (intra_count - prior);      // the smallest prior count that makes "a third so far are intra" true
                    if (prior >= need) L.slw[0] = max(L.slw[0], need); else L.slw[1] = min(L.slw[1], need);
                }
            }
            if constexpr (!BS) {
            {
                const int ra = rl(S.cref, 4), rb = rl(S.cref, 1);
                if (ra == -2 || rb == -2 || (ra == 0 && !(rl(S.cmvx, 4) | rl(S.cmvy, 4))) || (rb == 0 && !(rl(S.cmvx, 1) | rl(S.cmvy, 1)))) pskx = psky = 0;
                else mb_predict_mv(S, D_16x16, 0, 0, 2, 0, pskx, psky);
            }
            bool try_skip = false;
            if (k.fast_pskip) {
                if (c.subme >= 3) try_skip = true;
                else if (type_left == X264GPU_MB_P_SKIP || type_top == X264GPU_MB_P_SKIP || type_tl == X264GPU_MB_P_SKIP || type_tr == X264GPU_MB_P_SKIP)
                    pskip = mb_probe_pskip(k, c, cz, pskx, psky, q_lp, q_cp);
            }
            pf.mark(PH_PSKIP);
            if (pskip) {
                if (lane == 0) { mv16[2 * mbi] = 0; mv16[2 * mbi + 1] = 0; }
                if (lane >= 1 && lane < c.nref) { int16_t *m = k.mvr[lane] + ((size_t)s * k.nmb + mbi) * 2; m[0] = 0; m[1] = 0; }
            }
            // ---- 16x16 search candidates of every reference (oracle predict_mv_ref16x16): lookahead vector, 16x16 results of the left / top /
            //      top-left / top-right macroblocks in that reference, co-located vectors of reference 0 scaled by the POC distances.
            //      Slot order: [lookahead] left top top-left top-right [co-located, its right neighbour, its lower neighbour]; lane i of a
            //      16-lane group fetches raw candidate i, groups 0..3 serve references 0..3 in ONE trip to memory ----
            const bool has_lr = k.lowres_mv && (k.lowres_mv + (size_t)s * k.nmb * 2)[0] != 0x7fff;
            const bool t1 = k.temporal && mbx < k.mbw - 1, t2 = k.temporal && mby < k.mbh - 1;
            auto gather16 = [&](int r, int ln, int &vx, int &vy) {
                const int i = ln & 15;
                const int16_t *mvr = (r == 0 ? k.mv16_cur : r == 1 ? k.mvr[1] : r == 2 ? k.mvr[2] : r == 3 ? k.mvr[3] : r == 4 ? k.mvr[4] : r == 5 ? k.mvr[5] : k.mvr[6]) + (size_t)s * k.nmb * 2;
                const int base = has_lr && r == 0 ? 1 : 0, q = i - base;
                vx = 0; vy = 0;
                if (base && i == 0) { const int16_t *lm = k.lowres_mv + (size_t)s * k.nmb * 2; vx = lm[2 * mbi] * 2; vy = lm[2 * mbi + 1] * 2; }
                else if (q >= 0 && q < 4) {
                    const int nb = q == 0 ? (left ? mbi - 1 : -1) : q == 1 ? (top ? mbi - k.mbw : -1) : q == 2 ? (topleft ? mbi - k.mbw - 1 : -1) : (topright ? mbi - k.mbw + 1 : -1);
                    if (nb >= 0) { vx = mvr[2 * nb]; vy = mvr[2 * nb + 1]; }
                } else if (q >= 4 && q < 7 && k.temporal) {
                    // co-located; then its right neighbour (if there is one), then its lower neighbour (if there is one)
                    const int16_t *l0 = k.mv16_ref0 + (size_t)s * k.nmb * 2;
                    const int scale = r == 0 ? k.tscale[0] : r == 1 ? k.tscale[1] : r == 2 ? k.tscale[2] : r == 3 ? k.tscale[3] : r == 4 ? k.tscale[4] : r == 5 ? k.tscale[5] : k.tscale[6], jx = q - 4;
                    const bool ok = jx == 0 || (jx == 1 && (t1 || t2)) || (jx == 2 && t1 && t2);
                    const int at = jx == 0 ? mbi : (jx == 1 && t1) ? mbi + 1 : mbi + k.mbw;
                    if (ok) { vx = (l0[2 * at] * scale + 128) >> 8; vy = (l0[2 * at + 1] * scale + 128) >> 8; }
                }
            };
            int gvx = 0, gvy = 0;
            if (!pskip && (lane >> 4) < c.nref) gather16(lane >> 4, lane, gvx, gvy);
            // ---- inter analysis: one search call site, walked by a small state machine (oracle analyse_inter_*) ----
            const bool psub16 = (parts & 1) != 0, mixed = k.mixed_refs != 0;
            int stage = 0, part = 0, kk = 0, nk = c.nref;
            int halfpel_thresh = 0x7fffffff;
            int i_maxref = c.nref - 1, ref_a = 0, ref_b = 0;      // 8x8: highest reference tried; 16x8 / 8x16: the two candidate references
            int est1 = 0;
            int sat8[4] = { 0, 0, 0, 0 };
            // --weightp 2: the blind duplicate of reference 0 (always index 1) is never searched — it always follows a search of reference 0 on the
            // same block and starts from that result (prev_*; x264_me_refine_qpel_refdupe).  0 = the slice has none
            const int dupe = k.blind_dupe;
            int prev_mvx = 0, prev_mvy = 0, prev_cost = 0;
            bool done = pskip;
            WinTags wt;
            wt.tref = -1; wt.tx = wt.ty = 0;
            wl(S.cost, lane, ME_16, 0x7fffffff);
            i_cost = 0x7fffffff;
            // a 16x8 / 8x16 half: candidate references = those of its two 8x8 blocks (lower index first); before the first half also the
            // cost estimate of the second one and an empty cache for this macroblock's blocks
            auto setup_half = [&](int st, int pt) {
                lds_sync();
                const int b0 = st == 2 ? 2 * pt : pt, b1 = st == 2 ? 2 * pt + 1 : pt + 2;
                const int r0 = rl(S.ref, ME_8 + b0), r1 = rl(S.ref, ME_8 + b1);
                ref_a = min(r0, r1); ref_b = max(r0, r1); nk = ref_a == ref_b ? 1 : 2;
                if (pt == 0) {
                    const int o0 = st == 2 ? 2 : 1;
                    const int avg = (rl(S.costmv, ME_8 + o0) + rl(S.refcost, ME_8 + o0) + rl(S.costmv, ME_8 + 3) + rl(S.refcost, ME_8 + 3) + 1) >> 1;
                    est1 = (st == 2 ? sat8[2] : sat8[1]) + sat8[3] + avg;
                    if (lane == 5 || lane == 6 || lane == 9 || lane == 10) S.cref = -2;
                }
            };
            while (!done) {
                lds_sync();
                MeJob jb;
                int slot, r;
                jb.search = stage < 4; jb.use_thresh = false; jb.n_mvc = 0; jb.qonly = false;
                if (stage == 0) {
                    r = kk; slot = ME_16;
                    jb.W = 16; jb.H = 16; jb.ox = 0; jb.oy = 0;
                    mb_predict_mv(S, D_16x16, 0, 0, 2, r, jb.mvpx, jb.mvpy);
                    // candidates (oracle predict_mv_ref16x16): lookahead vector, 16x16 results of the left / top / top-left / top-right macroblocks
                    // in this reference, co-located vectors of reference 0 scaled by the POC distances
                    {
                        // raw candidate i of reference r: gathered for references 0..3 in one trip before the loop (gvx / gvy, lane = r * 16 + i)
                        const int base = has_lr ? (r == 0 ? 1 : 0) : 0;
                        if (r < 4) { S.inx = __shfl(gvx, r * 16 + (lane & 15)); S.iny = __shfl(gvy, r * 16 + (lane & 15)); }
                        else { int vx, vy; gather16(r, lane, vx, vy); S.inx = vx; S.iny = vy; }
                        jb.n_mvc = base + 4 + (k.temporal ? 1 + (t1 ? 1 : 0) + (t2 ? 1 : 0) : 0);
                    }
                    lds_sync();
                    jb.use_thresh = early_term && c.nref > 1;
                    jb.hp_it = c.subme >= 2 ? (c.subme < 6 ? 1 : c.subme < 8 ? 2 : 4) : 0;
                    jb.qp_it = c.subme < 4 ? 0 : c.subme == 4 ? 1 : c.subme < 8 ? 2 : 10;
                } else if (stage == 1) {
                    // mixed references: 0 .. i_maxref, then the duplicate if the early termination cut the loop short of it; else the 16x16 choice
                    // (a duplicate stands for reference 0 there: its reference bits rarely pay in P_8x8)
                    r = mixed ? (kk > i_maxref ? dupe : kk) : (dupe && rl(S.ref, ME_16) == dupe ? 0 : rl(S.ref, ME_16)); slot = ME_8 + part;
                    jb.W = 8; jb.H = 8; jb.ox = 8 * (part & 1); jb.oy = 8 * (part >> 1);
                    mb_predict_mv(S, D_8x8, part & 1, part >> 1, 1, r, jb.mvpx, jb.mvpy);
                    { const int q = r * 5 + min(lane, 4); S.inx = __shfl(S.mvcx, q); S.iny = __shfl(S.mvcy, q); }        // lane i <- mvc[r][i], i <= part
                    jb.n_mvc = part + 1;
                } else if (stage == 2) {
                    r = kk == 0 ? ref_a : ref_b; slot = ME_16x8 + part;
                    jb.W = 16; jb.H = 8; jb.ox = 0; jb.oy = 8 * part;
                    mb_predict_mv(S, D_16x8, 0, part, 2, r, jb.mvpx, jb.mvpy);
                    { const int q = r * 5 + (lane == 0 ? 0 : lane < 3 ? 2 * part + lane : 0); S.inx = __shfl(S.mvcx, q); S.iny = __shfl(S.mvcy, q); }
                    jb.n_mvc = 3;
                } else if (stage == 3) {
                    r = kk == 0 ? ref_a : ref_b; slot = ME_8x16 + part;
                    jb.W = 8; jb.H = 16; jb.ox = 8 * part; jb.oy = 0;
                    mb_predict_mv(S, D_8x16, part, 0, 1, r, jb.mvpx, jb.mvpy);
                    { const int q = r * 5 + (lane == 0 ? 0 : lane == 1 ? part + 1 : lane == 2 ? part + 3 : 0); S.inx = __shfl(S.mvcx, q); S.iny = __shfl(S.mvcy, q); }
                    jb.n_mvc = 3;
                } else {
                    // x264_me_refine_qpel of the winner's blocks
                    slot = best_part == D_16x16 ? ME_16 : best_part == D_16x8 ? ME_16x8 + part : best_part == D_8x16 ? ME_8x16 + part : ME_8 + part;
                    r = rl(S.ref, slot);
                    jb.W = best_part == D_16x16 || best_part == D_16x8 ? 16 : 8; jb.H = best_part == D_16x16 || best_part == D_8x16 ? 16 : 8;
                    jb.ox = best_part == D_8x16 ? 8 * part : best_part == D_8x8 ? 8 * (part & 1) : 0;
                    jb.oy = best_part == D_16x8 ? 8 * part : best_part == D_8x8 ? 8 * (part >> 1) : 0;
                    jb.mvpx = rl(S.mvpx, slot); jb.mvpy = rl(S.mvpy, slot);
                    jb.hp_it = c.subme == 1 ? 1 : 0;
                    jb.qp_it = c.subme == 1 ? 1 : c.subme >= 2 && c.subme <= 5 ? (c.subme == 2 ? 1 : 2) : 0;
                }
                if (stage >= 1 && stage <= 3) {
                    jb.hp_it = c.subme >= 2 ? (c.subme < 6 ? 1 : c.subme < 8 ? 2 : 4) : 0;
                    jb.qp_it = c.subme < 4 ? 0 : c.subme == 4 ? 1 : c.subme < 8 ? 2 : 10;
                }
                jb.ref = r;
                if (dupe && r == dupe && (stage < 2 || (stage < 4 && kk == 1 && ref_a == 0))) {
                    jb.qonly = true; jb.hp_it = 0; jb.qp_it = c.subme < 4 ? 0 : c.subme == 4 ? 1 : 2;      // min(2, subpel_iterations[subme][3])
                }
                lds_sync();
                const int rc = ref_bits(c.nref, r) * c.lambda;
                int mvx = 0, mvy = 0, cost = 0, cost_mv = 0;
                if (jb.qonly) { mvx = prev_mvx; mvy = prev_mvy; cost = prev_cost; }
                if (stage == 0) halfpel_thresh -= rc;
                if (stage == 4) { mvx = rl(S.mvx, slot); mvy = rl(S.mvy, slot); cost = rl(S.cost, slot) - rl(S.refcost, slot); cost_mv = rl(S.costmv, slot); }
                me_search<M, ME>(k, L, c, jb, mvx, mvy, cost, cost_mv, halfpel_thresh, S, wt, pf);
                pf.count(13 + (stage == 4));
                prev_mvx = mvx; prev_mvy = mvy; prev_cost = cost + rc;
                lds_sync();
                // ---- merge / advance ----
                if (stage == 0) {
                    if (lane == 0) {
                        int16_t *mo = (r == 0 ? k.mv16_cur : k.mvr[r]) + ((size_t)s * k.nmb + mbi) * 2;
                        mo[0] = (int16_t)mvx; mo[1] = (int16_t)mvy;
                    }
                    wl(S.mvcx, lane, r * 5, mvx); wl(S.mvcy, lane, r * 5, mvy);
                    if (r == 0 && try_skip && cost - cost_mv < 300 * c.lambda && abs(mvx - pskx) + abs(mvy - psky) <= 1 && mb_probe_pskip(k, c, cz, pskx, psky, q_lp, q_cp)) {
                        pskip = true; done = true;
                        if (lane >= 1 && lane < c.nref) { int16_t *m = k.mvr[lane] + ((size_t)s * k.nmb + mbi) * 2; m[0] = 0; m[1] = 0; }
                        continue;
                    }
                    cost += rc; halfpel_thresh += rc;
                    if (cost < rl(S.cost, ME_16)) me_store(S, lane, ME_16, mvx, mvy, cost, cost_mv, r, rc, jb.mvpx, jb.mvpy);
                    if (++kk < nk) continue;
                    // 16x16 done
                    i_cost = rl(S.cost, ME_16); best_part = D_16x16;
                    if (!psub16) { stage = 4; part = 0; if (!c.subme || rdon) done = true; continue; }
                    stage = 1; part = 0; kk = 0;
                    i_maxref = c.nref - 1;
                    if (mixed) {
                        if (early_term && i_maxref > 0 && (rl(S.ref, ME_16) == 0 || (dupe && rl(S.ref, ME_16) == dupe)) && type_top > 0 && type_left > 0) {
                            auto nbr = [&](int g) { const int v = rl(S.cref, g); return dupe && v == dupe ? 0 : v; };      // a neighbour's duplicate does not count
                            i_maxref = max(max(max(nbr(0), nbr(1)), max(nbr(2), nbr(3))), max(max(nbr(4), nbr(8)), 0));
                        }
                        nk = i_maxref + 1 + (dupe > i_maxref ? 1 : 0);
                    } else {
                        nk = 1;
                        if (dupe && rl(S.ref, ME_16) == dupe) { wl(S.mvcx, lane, 0, rl(S.mvx, ME_16)); wl(S.mvcy, lane, 0, rl(S.mvy, ME_16)); }      // mvc[0] = the 16x16 winner's vector (x264 analyse_inter_p8x8)
                    }
                    if (lane == 5 || lane == 6 || lane == 9 || lane == 10) S.cref = -2;      // this macroblock's blocks: not decided yet
                    wl(S.cost, lane, ME_8, 0x7fffffff);
                    continue;
                }
                if (stage == 1) {
                    const int rcost = mixed ? rc : ((r || k.cabac) ? rc : 0);        // CAVLC: reference 0 of P_8x8 costs nothing (P_8x8ref0)
                    cost += rcost;
                    wl(S.mvcx, lane, r * 5 + part + 1, mvx); wl(S.mvcy, lane, r * 5 + part + 1, mvy);
                    if (kk == 0 || cost < rl(S.cost, slot)) me_store(S, lane, slot, mvx, mvy, cost, cost_mv, r, rcost, jb.mvpx, jb.mvpy);
                    if (++kk < nk) continue;
                    // block done: into the cache, cost bookkeeping
                    {
                        const int g = ((part >> 1) + 1) * 4 + (part & 1) + 1;
                        wl(S.cref, lane, g, rl(S.ref, slot)); wl(S.cmvx, lane, g, rl(S.mvx, slot)); wl(S.cmvy, lane, g, rl(S.mvy, slot));
                        const int sv = rl(S.cost, slot) - (rl(S.costmv, slot) + rl(S.refcost, slot));
                        if (part == 0) sat8[0] = sv; else if (part == 1) sat8[1] = sv; else if (part == 2) sat8[2] = sv; else sat8[3] = sv;
                        S.cost += (lane == slot && !k.cabac) ? c.lambda : 0;        // sub-macroblock type: free under CABAC without sub-8x8 analysis
                    }
                    kk = 0;
                    if (++part < 4) continue;
                    cost8x8 = rl(S.cost, ME_8) + rl(S.cost, ME_8 + 1) + rl(S.cost, ME_8 + 2) + rl(S.cost, ME_8 + 3);
                    if (mixed && !k.cabac && !(rl(S.ref, ME_8) | rl(S.ref, ME_8 + 1) | rl(S.ref, ME_8 + 2) | rl(S.ref, ME_8 + 3))) cost8x8 -= ref_bits(c.nref, 0) * c.lambda * 4;
                    const int c16 = rl(S.cost, ME_16);
                    if (!early_term || cost8x8 < c16) { best_part = D_8x8; i_cost = cost8x8; }
                    const int th = rl(S.costmv, ME_8 + 1) + rl(S.costmv, ME_8 + 2);
                    if (!early_term || cost8x8 < c16 + th) { stage = 2; part = 0; kk = 0; setup_half(2, 0); }
                    else { stage = 4; part = 0; if (!c.subme || rdon) done = true; }
                    continue;
                }
                if (stage == 2 || stage == 3) {
                    cost += rc;
                    if (kk == 0 || cost < rl(S.cost, slot)) me_store(S, lane, slot, mvx, mvy, cost, cost_mv, r, rc, jb.mvpx, jb.mvpy);
                    if (++kk < nk) continue;
                    // this half is done; early termination on the first half plus the estimate of the second
                    bool shape_done = part == 0 && early_term && rl(S.cost, slot) + est1 > i_cost * (4 + (rdon ? 1 : 0)) / 4;      // (RD sessions keep shapes within a quarter above the best)
                    if (!shape_done) {
                        {
                            const int g0 = stage == 2 ? (part + 1) * 4 + 1 : 5 + part, g1 = stage == 2 ? g0 + 1 : g0 + 4;
                            const bool m = lane == g0 || lane == g1;
                            const int fr = rl(S.ref, slot), fx = rl(S.mvx, slot), fy = rl(S.mvy, slot);
                            S.cref = m ? fr : S.cref; S.cmvx = m ? fx : S.cmvx; S.cmvy = m ? fy : S.cmvy;
                        }
                        if (part == 0) { part = 1; kk = 0; setup_half(stage, 1); continue; }
                        const int total = rl(S.cost, slot - 1) + rl(S.cost, slot);
                        if (stage == 2) satd16x8 = total; else satd8x16 = total;
                        if (total < i_cost) { i_cost = total; best_part = stage == 2 ? D_16x8 : D_8x16; }
                    }
                    if (stage == 2) { stage = 3; part = 0; kk = 0; setup_half(3, 0); }
                    else { stage = 4; part = 0; if (!c.subme || rdon) done = true; }
                    continue;
                }
                // stage 4: refinement results
                wl(S.mvx, lane, slot, mvx); wl(S.mvy, lane, slot, mvy); wl(S.cost, lane, slot, cost); wl(S.costmv, lane, slot, cost_mv);
                const int np = best_part == D_16x16 ? 1 : best_part == D_8x8 ? 4 : 2;
                if (++part < np) continue;
                i_cost = 0;
                for (int i = 0; i < np; i++) i_cost += rl(S.cost, (best_part == D_16x16 ? ME_16 : best_part == D_16x8 ? ME_16x8 : best_part == D_8x16 ? ME_8x16 : ME_8) + i);
                done = true;
            }
            lds_sync();
            pf.mark(PH_ME_GLUE);
            }      // !BS
        }
        // ---- intra analysis (P slices: against the inter cost; chroma-ME decides the chroma mode first and carries its cost) ----
        if constexpr (BS) { /* B slices: after the inter RD decision (k_mb_b.inc) */ }
        else if (!pskip) {
            const int i_satd_inter = pslice ? i_cost : MB_COST_MAX;
            i_inter_satd = i_satd_inter;
            // the chroma mode depends on the neighbours only: decided here for every macroblock that is analysed (x264 does it here under chroma-ME,
            // otherwise only for macroblocks that end up intra — same mode either way); its cost enters the comparison under chroma-ME only
            TrCtx tra_ctx;                   // --trellis 2: the analysis' block encodes run the search too
            tra_ctx.on = 0; tra_ctx.r = 0; tra_ctx.r8 = 0; tra_ctx.model = 0; tra_ctx.tt.size_unary = nullptr; tra_ctx.tt.trans_unary = nullptr; tra_ctx.tt.lambda2 = nullptr;
            if constexpr (TRL2) {
                if ((k.trellis & 64) && RD && k.rd) { tra_ctx.on = k.trellis & 63; tra_ctx.r = cab.r; tra_ctx.r8 = cab.r8; tra_ctx.model = cab_modelv; tra_ctx.tt.size_unary = k.tr_su; tra_ctx.tt.trans_unary = k.tr_tu; tra_ctx.tt.lambda2 = k.tr_l2; }
            }
            satd_chroma = mb_intra_chroma_cost(k, L, c, predc, rf_dirc);
            pf.mark(PH_INTRA_CHROMA);
            mb_analyse_intra(k, L, c, cz, t4, parts, c.chroma_me ? i_satd_inter - satd_chroma : i_satd_inter, fast_intra, early_term, rdon, q_li, q8i, IR, TRL2 && tra_ctx.on ? &tra_ctx : nullptr);
            if (c.chroma_me) { IR.satd_i16 += satd_chroma; IR.satd_i8 += satd_chroma; IR.satd_i4 += satd_chroma; }
            if (pslice) {
                if (lane == 0) { recd.aux[0] = i_satd_inter; recd.aux[1] = min(min(IR.satd_i16, IR.satd_i8), IR.satd_i4); recd.aux[2] = rl(S.cost, ME_16); }
                mb_type = best_part == D_8x8 ? X264GPU_MB_P_8x8 : X264GPU_MB_P_L0;
                if (IR.satd_i16 < i_cost) { i_cost = IR.satd_i16; mb_type = X264GPU_MB_I16x16; }
                if (IR.satd_i8 < i_cost) { i_cost = IR.satd_i8; mb_type = X264GPU_MB_I8x8; }
                if (IR.satd_i4 < i_cost) { i_cost = IR.satd_i4; mb_type = X264GPU_MB_I4x4; }
            } else {
                i_cost = IR.satd_i16; mb_type = X264GPU_MB_I16x16;
                if (IR.satd_i4 < i_cost) { i_cost = IR.satd_i4; mb_type = X264GPU_MB_I4x4; }
                if (IR.satd_i8 < i_cost) { i_cost = IR.satd_i8; mb_type = X264GPU_MB_I8x8; }
            }
        } else { mb_type = X264GPU_MB_P_SKIP; i_cost = 0; }
        if (lane == 0) recd.cost = i_cost;
        pf.mark(PH_INTRA);

        // ---- x264_macroblock_encode.  One pass of the code below codes the macroblock.  The RD instantiation first runs it for candidates
        //      (e_type / e_part / e_t8, results kept on chip, nothing stored) and costs each with x264_rd_cost_mb: SSD + psy + lambda2 x bits ----
        unsigned nnz = 0;
        int cbp_luma = 0, cbp_chroma = 0;
        int e_type = mb_type, e_part = best_part, e_t8 = -1;          // what this pass codes; e_t8: -1 = SA8D vs SATD decides, else the transform size
        bool commit = true;
        // RD bookkeeping (x264_mb_analysis_t l0.i_rd16x16 / i_cost16x8 / i_cost8x16 / i_cost8x8 turned into RD costs, the intra ones likewise)
        int rd_ph = 0, rd16 = MB_COST_MAX, rd16x8 = MB_COST_MAX, rd8x16 = MB_COST_MAX, rd8x8 = MB_COST_MAX, rd_best = MB_COST_MAX, rd_part = D_16x16, rd_t8 = 0;
        int rd_satd_inter = 0, rd_isatd = 0, rd_thresh = 0, rd_ithresh = 0, rd_i16 = MB_COST_MAX, rd_i4 = MB_COST_MAX, rd_i8 = MB_COST_MAX, fenc_e4 = 0, fenc_e8 = 0;
        const int lambda2 = c_lambda2_tab[c.qp];
        bool rd_run = false, rd_skip16 = false;
        if constexpr (RD) {
            rd_run = rdon && !pskip;
            if (rd_run) {
                commit = false;
                const int i_satd_intra = min(min(IR.satd_i16, IR.satd_i8), IR.satd_i4);
                rd_satd_inter = i_inter_satd;
                rd_isatd = min(i_inter_satd, i_satd_intra);
                rd_thresh = early_term ? rd_isatd * 5 / 4 + 1 : MB_COST_MAX;
                if (k.psy_rd_q8) { int e4, e8; psy_energy_z(cz, lane, e4, e8); fenc_e4 = wave_sum(e4) >> 1; fenc_e8 = wave_sum(e8) >> 2; }      // fenc_hadamard_cache
            }
        }
        int rec_type = mb_type, chroma_l2off = 256;
        if constexpr (RD) { if (k.psy) chroma_l2off = c_chroma_lambda2_offset[min(max(c.qp - c.qpc + 12, 0), 36)]; }
        // B slices: the direct prediction, the candidate being coded (ecfg; euse: two bits per 8x8 block, 0 list 0 / 1 list 1 / 2 both / 3 direct), the
        // SATD and RD costs of x264_mb_analysis_t's B fields, the per-block / per-half list decisions
        BCfg dcfg = { -1, 0, 0, -1, 0, 0 }, ecfg = { -1, 0, 0, -1, 0, 0 };
        unsigned euse = 0;
        bool d_avail = true;             // the direct prediction exists (temporal direct: not when a co-located block's reference is out of reach)
        int bskip_cost = MB_COST_MAX, cost16direct = MB_COST_MAX, cost8d_0 = MB_COST_MAX, cost8d_1 = MB_COST_MAX, cost8d_2 = MB_COST_MAX, cost8d_3 = MB_COST_MAX, cost16bi = MB_COST_MAX;
        int cost8x8bi = MB_COST_MAX, cost16x8bi = MB_COST_MAX, cost8x16bi = MB_COST_MAX;
        int rd_dir = MB_COST_MAX, rd_l0 = MB_COST_MAX, rd_l1 = MB_COST_MAX, rd_bi = MB_COST_MAX, rd_8x8 = MB_COST_MAX, rd_16x8 = MB_COST_MAX, rd_8x16 = MB_COST_MAX;
        int sub8_0 = 0, sub8_1 = 0, sub8_2 = 0, sub8_3 = 0, p16x8_0 = 0, p16x8_1 = 0, p8x16_0 = 0, p8x16_1 = 0;
        int b_type = X264GPU_MB_B_SKIP, b_part = D_16x16, b_use16 = 0;
        uint32_t b_cenc = 0;             // this lane's chroma source row (lanes 0..31: plane (lane >> 4) & 1, 4x4 block (lane >> 2) & 3, row lane & 3)
        WinTags wt;
        wt.tref = -1; wt.tx = wt.ty = 0;
        // RD refinement (REF instantiations): the coroutine's state (k_mb_refine.inc) — what is being refined, where its walk stands, the candidate
        // out for costing, x264's left-over caches (non_zero_count of this macroblock as the last encode left it, |mvd| of the parts done)
        bool rf_on = false;              // the refinement coroutine owns the passes (P / I slices: rd_ph 9; B slices: after k_mb_b.inc's decision)
        int rf_kind = 0, rf_pk = 0, rf_st = 0, rf_part = 0, rf_j = 0, rf_i = 0, rf_wait = 0;
        int rf_bmx = 0, rf_bmy = 0, rf_omx = 0, rf_omy = 0, rf_pmx = 0, rf_pmy = 0, rf_dir = -2, rf_odir = 0, rf_bsatd = 0, rf_pmvchk = 0;
        int rf_cx = 0, rf_cy = 0, rf_cdir = -99, rf_mvpx = 0, rf_mvpy = 0, rf_mv0x = 0, rf_mv0y = 0, rf_f4 = 0, rf_f8 = 0;
        int rf_lmx = 0, rf_lmy = 0, rf_priced = 0, rf_done = 0, rf_ref = 0, rf_slot = 0, rf_pm8 = 0, rf_b0 = 0, rf_b1 = -1;
        unsigned long long rf_bcost = 0, rf_cost = 0, rf_amvd = 0;
        // ... of a B inter type (k_mb_b_rdrefine.inc, rf_kind 2): the part's list use (0 / 1 one list, 2 both), both lists' predicted vectors, list 1's
        // candidate / best vector of a bi-predicted part, its |mvd| cache, the candidate's chroma prediction, the best pair of a round
        int rf_l = 0, rf_mvp0x = 0, rf_mvp0y = 0, rf_mvp1x = 0, rf_mvp1y = 0, rf_bm1x = 0, rf_bm1y = 0, rf_c1x = 0, rf_c1y = 0, rf_bestj = 0;
        unsigned long long rf_amvd1 = 0;
        uint32_t rf_cpred = 0;
        unsigned rf_nnzc = 0;
        uint32_t rf_pred = 0;
        // ... of the intra refinement: the mode out for costing (rf_cm), the best so far, the chroma pass' transform switch and last coded block pattern,
        // i_cbp_i8x8_luma; an Intra_4x4 block's nine candidate encodes side by side (lane = mode * 4 + row: levels, reconstruction, distortion, non-zero)
        int rf_cm = 0, rf_bm = 0, rf_bdct = 0, rf_cbpc = 0, rf_cbp_i8 = 0, rf_old = 0;
        unsigned long long rf_list = 0;
        int rf4_v0 = 0, rf4_v1 = 0, rf4_v2 = 0, rf4_v3 = 0;
        uint32_t rf4_rz = 0, rf8_blo = 0, rf8_bhi = 0;
        int rf8_v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, rf8c_v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, rf8_cbp = 0;
        uint32_t rf8c_lo = 0, rf8c_hi = 0;
        unsigned rf8_n4 = 0;
        if constexpr (BS) {
            rd_run = true; commit = false;
            const int ci_ = (lane >> 2) & 3;
            const uint2 fe = *(const uint2 *)(L.csrc + ((ci_ >> 1) * 4 + j4) * 16 + 2 * (ci_ & 1) * 4);
            b_cenc = nv12_pick(fe.x, fe.y, (lane >> 4) & 1);
        }
        // The state of the pass loop is wave-uniform, but the compiler cannot see it: wherever a lane-dependent `if` sits directly in front of a jump, its
        // join block is the jump's target and every state variable merged there counts as divergent — from then on the state machine is vector code
        // under exec masks (1395 s_and_saveexec in the B instantiation) with its state in vector registers.  STATE_UNIFORM() re-reads the state into
        // scalar registers at the loop heads and stage boundaries.
#define STATE_UNIFORM() do { \
        uni_all(rd_ph, commit, rd_run, e_type, e_part, e_t8, mb_type, rec_type, best_part, i_cost, predc, satd_chroma, pskip, fast_intra, rd_skip16, nnz, cbp_luma, cbp_chroma); \
        uni_all(rd16, rd16x8, rd8x16, rd8x8, rd_best, rd_part, rd_t8, rd_satd_inter, rd_isatd, rd_thresh, rd_ithresh, rd_i16, rd_i4, rd_i8, fenc_e4, fenc_e8, chroma_l2off); \
        uni_all(IR.satd_i16, IR.satd_i8, IR.satd_i4, IR.pred16, IR.nnz4, IR.nnz8, IR.cbp8, cost8x8, satd16x8, satd8x16, i_inter_satd, pskx, psky); \
        if constexpr (BS) { \
            uni_all(euse, d_avail, bskip_cost, cost16direct, cost8d_0, cost8d_1, cost8d_2, cost8d_3, cost16bi, cost8x8bi, cost16x8bi, cost8x16bi); \
            uni_all(rd_dir, rd_l0, rd_l1, rd_bi, rd_8x8, rd_16x8, rd_8x16, sub8_0, sub8_1, sub8_2, sub8_3, p16x8_0, p16x8_1, p8x16_0, p8x16_1, b_type, b_part, b_use16, ds_t, ds_s); \
        } \
        if constexpr (REF) { \
            uni_all(rf_on, rf_kind, rf_pk, rf_st, rf_part, rf_j, rf_i, rf_wait, rf_bmx, rf_bmy, rf_omx, rf_omy, rf_pmx, rf_pmy, rf_dir, rf_odir, rf_bsatd, rf_pmvchk); \
            uni_all(rf_cx, rf_cy, rf_cdir, rf_mvpx, rf_mvpy, rf_mv0x, rf_mv0y, rf_f4, rf_f8, rf_lmx, rf_lmy, rf_priced, rf_done, rf_ref, rf_slot, rf_pm8, rf_b0, rf_b1); \
            uni_all(rf_bcost, rf_cost, rf_amvd, rf_l, rf_mvp0x, rf_mvp0y, rf_mvp1x, rf_mvp1y, rf_bm1x, rf_bm1y, rf_c1x, rf_c1y, rf_bestj, rf_amvd1, rf_nnzc); \
            uni_all(rf_cm, rf_bm, rf_bdct, rf_cbpc, rf_cbp_i8, rf_old, rf_list, rf8_cbp, rf8_n4); \
        } } while (0)
        for (;;) {
        RELANE();
        STATE_UNIFORM();
        if constexpr (BS) {
            if (rd_run && !commit) {
                if (!rf_on || rf_kind == 2) {          // (the refinement of an inter type runs inside k_mb_b.inc's last phase: k_mb_b_rdrefine.inc)
#include "k_mb_b.inc"
                }
                if constexpr (REF) {
                    // --subme 9: the decision's intra winner once more on RD cost (k_mb_b.inc switched rf_on on at its last phase)
                    if (rf_on && rf_kind != 2 && !commit) {
                        bool go = false;
#include "k_mb_refine.inc"
                        (void)go;
                    }
                }
            }
        } else if constexpr (RD) {
            if (rd_run && !commit) {
                // next candidate in x264's order: P16x16, 16x8, 8x16, 8x8 (x264_mb_analyse_p_rd), the other transform size of the winner
                // (x264_mb_analyse_transform_rd), I16x16, I4x4, I8x8 (x264_intra_rd); then the final pass
                const int c16 = pslice ? rl(S.cost, ME_16) : MB_COST_MAX;
                for (;; rd_ph++) {
                    bool go = false;
                    if (rd_ph == 0) { go = pslice && ((rl(S.ref, ME_16) == 0 && rl(S.mvx, ME_16) == pskx && rl(S.mvy, ME_16) == psky) || !early_term || c16 <= rd_isatd * 3 / 2); e_type = X264GPU_MB_P_L0; e_part = D_16x16; e_t8 = 0; }
                    else if (rd_ph == 1) { go = pslice && satd16x8 < rd_thresh; e_type = X264GPU_MB_P_L0; e_part = D_16x8; e_t8 = 0; }
                    else if (rd_ph == 2) { go = pslice && satd8x16 < rd_thresh; e_type = X264GPU_MB_P_L0; e_part = D_8x16; e_t8 = 0; }
                    else if (rd_ph == 3) { go = pslice && cost8x8 < rd_thresh; e_type = X264GPU_MB_P_8x8; e_part = D_8x8; e_t8 = 0; }
                    else if (rd_ph == 4) {
                        rd_best = rd16; rd_part = D_16x16;
                        if (rd16x8 < rd_best) { rd_best = rd16x8; rd_part = D_16x8; }
                        if (rd8x16 < rd_best) { rd_best = rd8x16; rd_part = D_8x16; }
                        if (rd8x8 < rd_best) { rd_best = rd8x8; rd_part = D_8x8; }
                        go = pslice && rd_best < MB_COST_MAX && k.dct8x8; e_type = rd_part == D_8x8 ? X264GPU_MB_P_8x8 : X264GPU_MB_P_L0; e_part = rd_part; e_t8 = 1;
                    } else if (rd_ph == 5) {
                        rd_ithresh = !early_term || !pslice ? MB_COST_MAX : rd_satd_inter * 5 / 4 + 1;
                        go = IR.satd_i16 < rd_ithresh; e_type = X264GPU_MB_I16x16;
                    } else if (rd_ph == 6) { go = IR.satd_i4 < rd_ithresh; e_type = X264GPU_MB_I4x4; }
                    else if (rd_ph == 7) { go = IR.satd_i8 < rd_ithresh; e_type = X264GPU_MB_I8x8; }
                    else if (rd_ph == 8) {
                        // the decision on RD costs, then the real pass
                        int best = pslice ? rd_best : MB_COST_MAX;
                        e_type = rd_part == D_8x8 ? X264GPU_MB_P_8x8 : X264GPU_MB_P_L0; e_part = rd_part; e_t8 = rd_t8;
                        if (pslice) {
                            if (rd_i16 < best) { best = rd_i16; e_type = X264GPU_MB_I16x16; }
                            if (rd_i8 < best) { best = rd_i8; e_type = X264GPU_MB_I8x8; }
                            if (rd_i4 < best) { best = rd_i4; e_type = X264GPU_MB_I4x4; }
                        } else {
                            best = rd_i16; e_type = X264GPU_MB_I16x16;
                            if (rd_i4 < best) { best = rd_i4; e_type = X264GPU_MB_I4x4; }
                            if (rd_i8 < best) { best = rd_i8; e_type = X264GPU_MB_I8x8; }
                        }
                        i_cost = best; mb_type = e_type; best_part = e_part;
                        if (rd_skip16) {           // as x264_macroblock_analyse leaves a P_SKIP found here: no cost, the other references' predictors zero
                            i_cost = 0; e_type = X264GPU_MB_P_SKIP; mb_type = e_type;          // and the final encode is x264_macroblock_encode_skip
                            if (lane == 0) { recd.aux[0] = 0; recd.aux[1] = 0; recd.aux[2] = 0; }
                            if (lane >= 1 && lane < c.nref) { int16_t *m = k.mvr[lane] + ((size_t)s * k.nmb + mbi) * 2; m[0] = 0; m[1] = 0; }
                        }
                        if (lane == 0) recd.cost = i_cost;
                        commit = true; go = true;
                        if constexpr (REF) {
                            // x264_macroblock_analyse: "if( analysis.i_mbrd >= 2 )" — the chosen type's vectors / intra modes once more on RD cost
                            const int sites = (k.rd >> 1) & 31;          // 1 inter vectors, 2 Intra_16x16 mode, 4 chroma mode, 8 Intra_4x4 modes, 16 Intra_8x8 modes
                            const bool inter_w = e_type >= X264GPU_MB_P_L0;
                            if (sites && !rd_skip16 && (inter_w ? (sites & 1) != 0 : (sites & 30) != 0)) {
                                commit = false; go = false; rd_ph = 9; rf_on = true;
                                rf_kind = inter_w ? 0 : 1; rf_st = 0; rf_part = 0; rf_wait = 0; rf_done = 0; rf_amvd = 0; rf_pk = 0;
                                if (lane == 5 || lane == 6 || lane == 9 || lane == 10) S.cref = -2;      // the motion cache of this macroblock's blocks starts empty
                            }
                        }
                    }
                    if constexpr (REF) {
                        if (rd_ph == 9 && !commit) {
#include "k_mb_refine.inc"
                        }
                    }
                    if (go) break;
                }
            }
        }
        STATE_UNIFORM();
        nnz = 0; cbp_luma = 0; cbp_chroma = 0;
        TrCtx trc;                                                // trellis: the final pass of RD sessions with CABAC (x264 --trellis 1)
        trc.on = 0; trc.r = 0; trc.r8 = 0; trc.model = 0; trc.tt.size_unary = nullptr; trc.tt.trans_unary = nullptr; trc.tt.lambda2 = nullptr;
        if constexpr (RD >= 3) {
            if ((commit || TRL2) && (k.trellis & 63) && rdon) { trc.on = k.trellis & 63; trc.r = cab.r; trc.r8 = cab.r8; trc.model = cab_modelv; trc.tt.size_unary = k.tr_su; trc.tt.trans_unary = k.tr_tu; trc.tt.lambda2 = k.tr_l2; }
        }
        int rd_t8cur = 0;                                         // transform_size_8x8_flag of what this pass codes
        rec_type = e_type;
        int16_t *lvw = RD ? (int16_t *)rd_lvs : lv;               // RD: levels stay on chip until the final pass has its bit counts' totals
        int ssd_y = 0, ssd_c = 0, en4 = 0, en8 = 0;          // per-lane shares of the distortion terms of the candidate
        // deblock-aware RD (x264 b_deblock_rdo): a whole-macroblock candidate's luma goes through a 16x16 tile in LDS (the chroma sub-pel staging area,
        // idle during the encode passes), is loop-filtered along its internal edges there, and the distortion terms are taken from the result
        uint8_t *const dtile = (uint8_t *)L.csub;
        bool dbr = false;
        unsigned db_mvm = 0;           // bit dir * 2 + half: the 8x8 blocks either side of edge 2 differ in motion (boundary strength 1)
        if constexpr (REF) dbr = k.deblock_rdo && rd_run && !commit;          // (the refinement instantiations only: launches with cfg.rd bit 6 go to them)
        bool part_pass = false;
        if constexpr (REF) part_pass = rf_pk != 0 && !commit;
        dbr = dbr && !part_pass;          // (x264_rd_cost_part does not filter: whole-macroblock costs only)
        if (part_pass) {
            if constexpr (REF) {
            if (rf_pk == 1) {
                // ---- x264_rd_cost_part of an inter part: x264_macroblock_encode_p8x8 of its 8x8 blocks with the prediction as COST_MV_SATD left it
                //      (rf_pred) — luma with the macroblock's transform size, per-8x8 decimation only, the chroma 4x4 block under each 8x8 without its DC ----
                const unsigned pm8 = (unsigned)rf_pm8;
                const bool t8 = e_t8 != 0;
                const uint32_t pred = rf_pred;
                rd_t8cur = t8;
                if (t8) {
                    uint32_t elo, ehi, plo, phi;
                    z_to_r8(cz, lane, elo, ehi); z_to_r8(pred, lane, plo, phi);
                    const int row = lane & 7, i8 = (lane >> 3) & 3;
                    int e[8], p[8], v[8];
                    unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
                    fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
                    int mf[4], bs[4], dq[4];
                    q8_row(q8p, row, mf, bs, dq);
                    unsigned mlo = 0, mhi = 0, big = 0;
                    const bool tr8 = TRL2 && (trc.on & TR_P8) != 0;
                    if (tr8) {
                        if (lane < 32)
#pragma unroll
                            for (int i = 0; i < 8; i++) lvw[i8 * 64 + c_zigzag8_inv[row * 8 + i]] = (int16_t)v[i];
                        lds_sync();
                        trellis_run<5>(trc, lvw, 64, 4, c.qp, false, lane);
                        lds_sync();
#pragma unroll
                        for (int i = 0; i < 8; i++) v[i] = lvw[i8 * 64 + c_zigzag8_inv[row * 8 + i]];
                        lds_sync();
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (!tr8) v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
                        const int z = c_zigzag8_inv[row * 8 + i];
                        if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
                        big |= abs(v[i]) > 1 ? 1u : 0u;
                    }
                    mlo = group8_or(mlo); mhi = group8_or(mhi); big = group8_or(big);
                    const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
                    bool keep = mask != 0;
                    if (k.dct_decimate && !tr8) keep = keep && (big || decimate64_from_mask(mask) >= 4);
                    const bool inp = lane < 32 && ((pm8 >> i8) & 1);
                    if (inp) {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int z = c_zigzag8_inv[row * 8 + i];
                            lvw[(i8 * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)(keep ? v[i] : 0);
                        }
                    }
                    unsigned n4 = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) n4 |= (mask & (0x1111111111111111ull << q)) ? 1u << q : 0u;
                    if (!keep) n4 = 0;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const unsigned ng = (pm8 >> g) & 1 ? (unsigned)__builtin_amdgcn_readlane((int)n4, g * 8) : 0u;
                        nnz |= ng << (4 * g);
                        cbp_luma |= ng ? 1 << g : 0;
                    }
                    const int qb = q8p.qp / 6 - 6;
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = keep ? dequant_one(v[i], dq[i & 3], qb) : 0;
                    inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
                    uint2 o;
                    o.x = pack4_clip8lo(v); o.y = pack4_clip8hi(v);
                    int e4 = 0, e8 = 0;
                    psy_energy_r8(o.x, o.y, lane, e4, e8);
                    if (inp) { ssd_y = ssd4_u8(elo, o.x) + ssd4_u8(ehi, o.y); en4 = e4; en8 = e8; }
                } else {
                    int e[4], p[4], v[4];
                    unpack4(cz, e); unpack4(pred, p);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = e[i] - p[i];
                    dct4_quad(v, lane);
                    if (TRL2 && (trc.on & TR_P4)) {
                        store_levels_scan(lvw + (lane >> 2) * 16, v, j4);
                        lds_sync();
                        trellis_run<2>(trc, lvw, 16, 16, c.qp, false, lane);
                        lds_sync();
                        load_levels_scan(lvw + (lane >> 2) * 16, v, j4);
                        lds_sync();
                    } else quant4_row(v, q_lp, j4);
                    const unsigned mask = (unsigned)quad_or((int)scan_mask(v, j4));
                    const bool nz = mask != 0;
                    bool keep = nz;
                    if (k.dct_decimate) {
                        const int big = quad_or(any_big(v) ? 1 : 0);
                        const int sc = nz ? (big ? 9 : decimate_from_mask(mask, 0)) : 0;
                        const int score8 = row16_sum(j4 == 0 ? sc : 0);
                        keep = nz && score8 >= 4;
                    }
                    const bool inp = ((pm8 >> (lane >> 4)) & 1) != 0;
                    if (inp) { int z[4] = { 0, 0, 0, 0 }; store_levels_scan(lvw + (lane >> 2) * 16, keep ? v : z, j4); }
                    if (!keep) v[0] = v[1] = v[2] = v[3] = 0;
                    dequant4_row(v, q_lp, j4);
                    idct4_quad(v, lane);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += p[i];
                    const uint32_t rz = pack4_clip(v);
                    int e4 = 0, e8 = 0;
                    psy_energy_z(rz, lane, e4, e8);
                    if (inp) { ssd_y = ssd4_u8(cz, rz); en4 = e4; }
                    if (lane < 32 && ((pm8 >> (lane >> 3)) & 1)) en8 = e8;
                    const unsigned long long bal = __ballot(keep && inp && j4 == 0);
#pragma unroll
                    for (int b = 0; b < 16; b++) nnz |= (unsigned)((bal >> (4 * b)) & 1) << b;
#pragma unroll
                    for (int i8 = 0; i8 < 4; i8++) cbp_luma |= ((nnz >> (4 * i8)) & 15) ? 1 << i8 : 0;
                }
                {
                    // chroma: prediction at the candidate vector, the 4x4 block under each of the part's 8x8 blocks, AC only
                    const int pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j4;
                    const bool act = lane < 32 && ((pm8 >> ci) & 1);
                    uint32_t cpred;
                    if constexpr (BS) cpred = rf_cpred;          // (B: the coroutine's b_predict of the candidate — one list or both averaged)
                    else {
                        uint32_t pu, pv;
                        mc_chroma_row4(ref_chroma00(k, s, rf_ref), k.rs, mbx * 8 + cx0, mby * 8 + cyy, rf_cx, rf_cy, pu, pv);
                        if (k.wc_any) { const int wu = uni(k.wc0[2 * rf_ref]), wv = uni(k.wc0[2 * rf_ref + 1]); if (wu >> 24) pu = wp4(pu, wu); if (wv >> 24) pv = wp4(pv, wv); }
                        cpred = pl ? pv : pu;
                    }
                    const uint2 fe = *(const uint2 *)(L.csrc + cyy * 16 + 2 * cx0);
                    const uint32_t cenc = nv12_pick(fe.x, fe.y, pl);
                    int e[4], p[4], v[4];
                    unpack4(cenc, e); unpack4(cpred, p);
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = act ? e[t] - p[t] : 0;
                    dct4_quad(v, lane);
                    if (j4 == 0) v[0] = 0;
                    int16_t *l = lvw + X264GPU_LV_CHROMA_AC + (pl * 4 + ci) * 16;
                    if (TRL2 && (trc.on & TR_C)) {
                        if (lane < 32) store_levels_scan(l, v, j4);
                        lds_sync();
                        trellis_run<4>(trc, lvw + X264GPU_LV_CHROMA_AC, 16, 8, q_cp.qp, false, lane);
                        lds_sync();
                        if (lane < 32) load_levels_scan(l, v, j4); else v[0] = v[1] = v[2] = v[3] = 0;
                        lds_sync();
                    } else quant4_row(v, q_cp, j4);
                    const bool nzc = quad_or((int)scan_mask(v, j4)) != 0 && act;
                    if (act) { int z[4] = { 0, 0, 0, 0 }; store_levels_scan(l, nzc ? v : z, j4); }
                    if (!nzc) v[0] = v[1] = v[2] = v[3] = 0;
                    dequant4_row(v, q_cp, j4);
                    idct4_quad(v, lane);
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] += p[t];
                    const uint32_t crec = pack4_clip(v);
                    if (act) ssd_c = ssd4_u8(cenc, crec);
                    const unsigned long long bal = __ballot(nzc && j4 == 0);
#pragma unroll
                    for (int b = 0; b < 8; b++) nnz |= (unsigned)((bal >> (4 * b)) & 1) << (16 + b);
                    cbp_chroma = 2;
                }
                // x264's non_zero_count cache after this encode: the part's entries are fresh, the rest stays what it was
                {
                    const unsigned lm = ((pm8 & 1) ? 0x000fu : 0) | ((pm8 & 2) ? 0x00f0u : 0) | ((pm8 & 4) ? 0x0f00u : 0) | ((pm8 & 8) ? 0xf000u : 0);
                    const unsigned cm = (pm8 | (pm8 << 4)) << 16;
                    const unsigned lfl = t8 ? ((cbp_luma & 1) ? 0x000fu : 0) | ((cbp_luma & 2) ? 0x00f0u : 0) | ((cbp_luma & 4) ? 0x0f00u : 0) | ((cbp_luma & 8) ? 0xf000u : 0) : nnz & 0xffffu;
                    rf_nnzc = (rf_nnzc & ~(lm | cm)) | (lfl & lm) | (nnz & cm);
                }
            } else if (rf_pk == 2) {
                // rd_cost_i4x4: the block's candidate encodes were done side by side by the coroutine; this mode's levels sit in the level buffer
                const int idx = rf_i, dn = ((const int *)L.cand)[rf_cm], bnz = dn & 1;
                nnz = (unsigned)bnz << idx;
                rf_nnzc = (rf_nnzc & ~(1u << idx)) | ((unsigned)bnz << idx);
                ssd_y = lane == 0 ? dn >> 1 : 0;
            } else if (rf_pk == 3) {
                // rd_cost_i8x8: predict block rf_i with mode rf_cm from the refined neighbours in the tile, transform, quantise, reconstruct
                const int idx = rf_i, x8 = idx & 1, y8 = idx >> 1, g = lane >> 3, r8 = lane & 7;
                const int avail = i8_avail(left, top, topright, idx);
                uint8_t *bt = tile8 + y8 * 8 * IT_STRIDE + x8 * 8;
                lds_sync();
                pred8_build_u(L.U8, bt, IT_STRIDE, avail, lane);
                const int src = idx * 16 + (r8 >> 2) * 8 + (r8 & 3);
                const uint32_t elo = (uint32_t)__shfl((int)cz, src), ehi = (uint32_t)__shfl((int)cz, src + 4);
                uint32_t plo, phi;
                pred8_row8(L.U8, L.pred8tab, rf_cm, r8, plo, phi);
                int e[8], p[8], v[8];
                unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
                fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
                int mf[4], bs[4], dq[4];
                q8_row(q8i, r8, mf, bs, dq);
                unsigned mlo = 0, mhi = 0;
                const bool tr8 = TRL2 && (trc.on & TR_I8) != 0;
                if (tr8) {
                    if (g == 0)
#pragma unroll
                        for (int i = 0; i < 8; i++) lvw[idx * 64 + c_zigzag8_inv[r8 * 8 + i]] = (int16_t)v[i];
                    lds_sync();
                    trellis_run<5>(trc, lvw + idx * 64, 64, 1, c.qp, true, lane);
                    lds_sync();
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = lvw[idx * 64 + c_zigzag8_inv[r8 * 8 + i]];
                    lds_sync();
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (!tr8) v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
                    const int z = c_zigzag8_inv[r8 * 8 + i];
                    if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
                    if (g == 0) lvw[(idx * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)v[i];
                    rf8c_v[i] = v[i];
                }
                mlo = group8_or(mlo); mhi = group8_or(mhi);
                const unsigned long long mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)mlo) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)mhi) << 32);
#pragma unroll
                for (int q = 0; q < 4; q++) nnz |= (mask & (0x1111111111111111ull << q)) ? 1u << (idx * 4 + q) : 0u;
                cbp_luma = (rf_cbp_i8 & ~(1 << idx)) | (mask ? 1 << idx : 0);
                const int qb = q8i.qp / 6 - 6;
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = dequant_one(v[i], dq[i & 3], qb);
                inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
                const uint32_t rlo = pack4_clip8lo(v), rhi = pack4_clip8hi(v);
                rf8c_lo = rlo; rf8c_hi = rhi;
                // ssd_plane( PIXEL_8x8 ): SSD + |hadamard_ac of the reconstruction - of the source| (lanes 0..7 = the block's rows; the others carry copies)
                int e4, e8, f4, f8;
                psy_energy_r8(rlo, rhi, lane, e4, e8);
                psy_energy_r8(elo, ehi, lane, f4, f8);
                if (g == 0) { ssd_y = ssd4_u8(elo, rlo) + ssd4_u8(ehi, rhi); en4 = e4; en8 = e8; }
                rf_f4 = wave_sum(g == 0 ? f4 : 0) >> 1; rf_f8 = wave_sum(g == 0 ? f8 : 0) >> 2;
                rd_t8cur = 1;
            } else if (rf_pk == 4) {
                // rd_cost_chroma: the chroma planes predicted with mode rf_cm, coded (unless a mode without residual was already seen), no luma
                const int pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j4;
                const PredC pc = predc_setup(L.cnb[pl]);
                const uint2 fe = *(const uint2 *)(L.csrc + cyy * 16 + 2 * cx0);
                const uint32_t cenc = nv12_pick(fe.x, fe.y, pl), cpred = predc_row4(L.cnb[pl], pc, rf_cm, ci, j4);
                uint32_t crec = cpred;
                if (rf_bdct) crec = mb_chroma_residual(cenc, cpred, q_ci, false, false, lane, lvw, nnz, cbp_chroma, TRL ? &trc : nullptr);
                if (lane < 32) ssd_c = ssd4_u8(cenc, crec);
                if (rf_bdct) rf_nnzc = (rf_nnzc & 0xffffu) | (cbp_chroma == 2 ? nnz & 0x00ff0000u : 0u);      // (what x264_mb_encode_chroma leaves in the non_zero_count cache)
            }
            }
        } else
        if (e_type >= X264GPU_MB_P_L0) {
            // this lane's 8x8 block's motion (Z layout: lane >> 4)
            const int b8 = lane >> 4;
            int lmx = 0, lmy = 0, lref = 0;
            const bool eskip = BS ? e_type == X264GPU_MB_B_SKIP : (pskip || e_type == X264GPU_MB_P_SKIP);      // probed, or found by the RD test of the 16x16 result at the skip vector: no residual
            const int pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j4;
            uint32_t pred, cpred;
            int mv0x = 0, mv0y = 0, ref0 = 0;
            if constexpr (BS) {
                pf.begin2();
                b_predict(k, c, ecfg, biwv, pred, cpred);
                { const int tch = wave_sum((int)(pred & 1) + (int)(cpred & 1)); if (tch == 12345678) pf.count(30); }      // (MB_PROF: the prediction has arrived)
                pf.mark2(24);
                if (commit && (lane & 15) == 0) {      // one lane per 8x8 block writes that block's motion in both lists
                    recd.ref[b8] = (int8_t)ecfg.r0; recd.mv[b8][0] = (int16_t)ecfg.x0; recd.mv[b8][1] = (int16_t)ecfg.y0;
                    recd.ref1[b8] = (int8_t)ecfg.r1; recd.mv1[b8][0] = (int16_t)ecfg.x1; recd.mv1[b8][1] = (int16_t)ecfg.y1;
                }
                if (commit && lane == 0) {
                    recd.partition = (uint8_t)e_part;
                    recd.direct8 = (uint8_t)(((euse & 3) == 3 ? 1 : 0) | (((euse >> 2) & 3) == 3 ? 2 : 0) | (((euse >> 4) & 3) == 3 ? 4 : 0) | (((euse >> 6) & 3) == 3 ? 8 : 0));
                }
            } else {
            if (eskip) { lmx = clampi(pskx, c.mvmin0, c.mvmax0); lmy = clampi(psky, c.mvmin1, c.mvmax1); lref = 0; }
            else {
                const int slot = e_part == D_16x16 ? ME_16 : e_part == D_16x8 ? ME_16x8 + (b8 >> 1) : e_part == D_8x16 ? ME_8x16 + (b8 & 1) : ME_8 + b8;
                lmx = __shfl(S.mvx, slot); lmy = __shfl(S.mvy, slot); lref = __shfl(S.ref, slot);      // slot varies with the lane (its 8x8 block)
            }
            pred = mc_luma_row4(ref_plane00(k, s, lref), k.plane_bytes, k.rs, c.px + zx, c.py + zy, lmx, lmy);
            if (k.wp_any) { const int wpk = k.wl0[lref]; if (wpk >> 24) pred = wp4(pred, wpk); }      // lref varies with the lane's 8x8 block
            // chroma prediction: chroma 4x4 block ci <-> luma 8x8 ci
            const int cmvx = __shfl(lmx, ci * 16), cmvy = __shfl(lmy, ci * 16), cref = __shfl(lref, ci * 16);
            uint32_t pu, pv;
            mc_chroma_row4(ref_chroma00(k, s, cref), k.rs, mbx * 8 + cx0, mby * 8 + cyy, cmvx, cmvy, pu, pv);
            cpred = pl ? pv : pu;
            if (k.wc_any) { const int wck = k.wc0[2 * cref + pl]; if (wck >> 24) cpred = wp4(cpred, wck); }      // (cref varies with the lane's chroma block)
            mv0x = eskip ? pskx : __builtin_amdgcn_readlane(lmx, 0); mv0y = eskip ? psky : __builtin_amdgcn_readlane(lmy, 0); ref0 = __builtin_amdgcn_readlane(lref, 0);
            if (commit && (lane & 15) == 0) {      // one lane per 8x8 block writes that block's motion
                recd.mv[lane >> 4][0] = (int16_t)(eskip ? pskx : lmx); recd.mv[lane >> 4][1] = (int16_t)(eskip ? psky : lmy); recd.ref[lane >> 4] = (int8_t)lref;
            }
            if (commit && lane == 0) recd.partition = (uint8_t)(eskip ? 0 : e_part);
            }
            if (dbr) {
                // boundary strength 1 across edge 2: the 8x8 blocks on either side differ in a reference index or by four quarter samples in a vector
                // component (deblock_strength on the motion cache: raw indices, both lists in B slices)
                int r0v, x0v, y0v, r1v = 0, x1v = 0, y1v = 0;
                if constexpr (BS) { r0v = ecfg.r0; x0v = ecfg.x0; y0v = ecfg.y0; r1v = ecfg.r1; x1v = ecfg.x1; y1v = ecfg.y1; }
                else { r0v = lref; x0v = eskip ? pskx : lmx; y0v = eskip ? psky : lmy; }
                auto differ = [&](int ka, int kb) __attribute__((always_inline)) {
                    const int la = ka * 16, lb = kb * 16;
                    bool d = rl(r0v, la) != rl(r0v, lb) || abs(rl(x0v, la) - rl(x0v, lb)) >= 4 || abs(rl(y0v, la) - rl(y0v, lb)) >= 4;
                    if (BS) d = d || rl(r1v, la) != rl(r1v, lb) || abs(rl(x1v, la) - rl(x1v, lb)) >= 4 || abs(rl(y1v, la) - rl(y1v, lb)) >= 4;
                    return d;
                };
                db_mvm = (differ(1, 0) ? 1u : 0u) | (differ(3, 2) ? 2u : 0u) | (differ(2, 0) ? 4u : 0u) | (differ(3, 1) ? 8u : 0u);
            }
            if (eskip) {
                *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = pred;
                mb_store_chroma(ruv, k.rs, lane, cpred);
                if (lane < 52) { uint4 z; z.x = z.y = z.z = z.w = 0; *(uint4 *)(lvw + lane * 8) = z; }
            } else {
                const uint32_t enc = cz;
                bool t8 = false;
                uint32_t elo = 0, ehi = 0, plo = 0, phi = 0;
                if (k.dct8x8 && e_t8 != 0) {                        // (e_t8 == 0: a candidate with the 4x4 transform — no row layout needed)
                    z_to_r8(enc, lane, elo, ehi); z_to_r8(pred, lane, plo, phi);
                    if (e_t8 >= 0) t8 = e_t8 != 0;                  // RD: the transform size belongs to the candidate
                    else {
                        const int h8 = sa8d_r8_half(elo, ehi, plo, phi, lane);
                        const int cost8 = (2 * wave_sum(lane < 32 ? h8 : 0) + 2) >> 2, cost4 = wave_sum(satd4_half(enc, pred, lane));
                        t8 = cost8 < cost4;
                    }
                }
                if (t8) {
                    const int row = lane & 7, i8 = (lane >> 3) & 3;
                    int e[8], p[8], v[8];
                    unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
                    fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
                    int mf[4], bs[4], dq[4];
                    q8_row(q8p, row, mf, bs, dq);
                    unsigned mlo = 0, mhi = 0, big = 0;
                    const bool tr8 = TRL && (trc.on & TR_P8) != 0;
                    if (tr8) {
                        if (lane < 32)
#pragma unroll
                            for (int i = 0; i < 8; i++) lvw[i8 * 64 + c_zigzag8_inv[row * 8 + i]] = (int16_t)v[i];
                        lds_sync();
                        trellis_run<5>(trc, lvw, 64, 4, c.qp, false, lane);
                        lds_sync();
#pragma unroll
                        for (int i = 0; i < 8; i++) v[i] = lvw[i8 * 64 + c_zigzag8_inv[row * 8 + i]];
                        lds_sync();
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (!tr8) v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
                        const int z = c_zigzag8_inv[row * 8 + i];
                        if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
                        big |= abs(v[i]) > 1 ? 1u : 0u;
                    }
                    mlo = group8_or(mlo); mhi = group8_or(mhi); big = group8_or(big);
                    const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
                    bool keep = mask != 0;
                    if (k.dct_decimate && !tr8) {          // (x264_macroblock_encode: the 8x8 trellis is its own decimation under CABAC)
                        const int sc = keep ? (big ? 9 : decimate64_from_mask(mask)) : 0;
                        const int mbscore = __builtin_amdgcn_readlane(sc, 0) + __builtin_amdgcn_readlane(sc, 8) + __builtin_amdgcn_readlane(sc, 16) + __builtin_amdgcn_readlane(sc, 24);
                        keep = keep && sc >= 4 && mbscore >= 6;
                    }
                    if (lane < 32) {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int z = c_zigzag8_inv[row * 8 + i];
                            lvw[(i8 * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)(keep ? v[i] : 0);
                        }
                    }
                    unsigned n4 = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) n4 |= (mask & (0x1111111111111111ull << q)) ? 1u << q : 0u;
                    if (!keep) n4 = 0;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const unsigned ng = (unsigned)__builtin_amdgcn_readlane((int)n4, g * 8);
                        nnz |= ng << (4 * g);
                        cbp_luma |= ng ? 1 << g : 0;
                    }
                    const int qb = q8p.qp / 6 - 6;
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = keep ? dequant_one(v[i], dq[i & 3], qb) : 0;
                    inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
                    if (lane < 32) {
                        uint2 o;
                        o.x = pack4_clip8lo(v); o.y = pack4_clip8hi(v);
                        if (commit) *(uint2 *)(rec + (size_t)((i8 >> 1) * 8 + row) * k.rs + (i8 & 1) * 8) = o;
                        if constexpr (RD) {      // R8 layout: a quad of lanes is four rows of one 8x8 block, so both halves are 4x4 blocks for the SATD
                            ssd_y = ssd4_u8(elo, o.x) + ssd4_u8(ehi, o.y);
                            psy_energy_r8(o.x, o.y, lane, en4, en8);
                            if (dbr) *(uint2 *)(dtile + ((i8 >> 1) * 8 + row) * 16 + (i8 & 1) * 8) = o;
                        }
                    }
                } else {
                    int e[4], p[4], v[4];
                    unpack4(enc, e); unpack4(pred, p);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = e[i] - p[i];
                    dct4_quad(v, lane);
                    if (TRL && (trc.on & TR_P4)) {
                        store_levels_scan(lvw + (lane >> 2) * 16, v, j4);
                        lds_sync();
                        trellis_run<2>(trc, lvw, 16, 16, c.qp, false, lane);
                        lds_sync();
                        load_levels_scan(lvw + (lane >> 2) * 16, v, j4);
                        lds_sync();
                    } else quant4_row(v, q_lp, j4);
                    const unsigned mask = (unsigned)quad_or((int)scan_mask(v, j4));
                    const bool nz = mask != 0;
                    bool keep = nz;
                    if (k.dct_decimate) {
                        const int big = quad_or(any_big(v) ? 1 : 0);
                        const int sc = nz ? (big ? 9 : decimate_from_mask(mask, 0)) : 0;
                        const int score8 = row16_sum(j4 == 0 ? sc : 0);
                        bool any8 = row16_or(nz ? 1 : 0) != 0;
                        const int mbscore = wave_sum(((lane & 15) == 0 && any8) ? score8 : 0);
                        if (any8 && score8 < 4) any8 = false;
                        keep = nz && any8 && mbscore >= 6;
                    }
                    { int z[4] = { 0, 0, 0, 0 }; store_levels_scan(lvw + (lane >> 2) * 16, keep ? v : z, j4); }
                    if (!keep) v[0] = v[1] = v[2] = v[3] = 0;
                    dequant4_row(v, q_lp, j4);
                    idct4_quad(v, lane);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += p[i];
                    {
                        const uint32_t rz = pack4_clip(v);
                        if (commit) *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = rz;
                        if constexpr (RD) { ssd_y = ssd4_u8(cz, rz); psy_energy_z(rz, lane, en4, en8); if (dbr) *(uint32_t *)(dtile + zy * 16 + zx) = rz; }
                    }
                    const unsigned long long bal = __ballot(keep && j4 == 0);
#pragma unroll
                    for (int b = 0; b < 16; b++) nnz |= (unsigned)((bal >> (4 * b)) & 1) << b;
#pragma unroll
                    for (int i8 = 0; i8 < 4; i8++) cbp_luma |= ((nnz >> (4 * i8)) & 15) ? 1 << i8 : 0;
                }
                pf.mark2(25);
                const uint2 fe = *(const uint2 *)(L.csrc + cyy * 16 + 2 * cx0);
                const uint32_t cenc = nv12_pick(fe.x, fe.y, pl);
                const uint32_t crec = mb_chroma_residual(cenc, cpred, q_cp, true, k.dct_decimate != 0, lane, lvw, nnz, cbp_chroma, TRL ? &trc : nullptr);
                pf.mark2(26);
                if (commit) mb_store_chroma(ruv, k.rs, lane, crec);
                if constexpr (RD) { if (lane < 32) ssd_c = ssd4_u8(cenc, crec); }
                if (lane >= 32 && lane < 40) lvw[X264GPU_LV_LUMA_DC + (lane - 32) * 2] = 0, lvw[X264GPU_LV_LUMA_DC + (lane - 32) * 2 + 1] = 0;
                if (lane >= 40 && lane < 44) lvw[408 + (lane - 40) * 2] = 0, lvw[408 + (lane - 40) * 2 + 1] = 0;
                rd_t8cur = t8 && cbp_luma;
                if (commit && lane == 0) recd.transform8x8 = (uint8_t)(t8 && cbp_luma);
                // P_L0 16x16, reference 0, the skip vector, nothing coded: P_SKIP
                if (!BS && e_type == X264GPU_MB_P_L0 && e_part == D_16x16 && !(cbp_luma | cbp_chroma) && ref0 == 0 && mv0x == pskx && mv0y == psky)
                    rec_type = X264GPU_MB_P_SKIP;
                // B_DIRECT with nothing coded: B_SKIP
                if (BS && e_type == X264GPU_MB_B_DIRECT && !(cbp_luma | cbp_chroma)) rec_type = X264GPU_MB_B_SKIP;
            }
            pf.mark(PH_ENC_INTER);
        } else {
            // ---- intra macroblock ----
            if (commit && lane < 4) recd.ref[lane] = -1;
            if (e_type == X264GPU_MB_I8x8) {
                rd_t8cur = 1;
                if (commit && lane == 0) recd.transform8x8 = 1;
                if (commit && lane < 16) recd.i4_mode[lane] = L.modes8[lane];
                nnz = IR.nnz8; cbp_luma = IR.cbp8;
                const bool tri8 = TRL && !TRL2 && (trc.on & TR_I8) != 0;         // (--trellis 2: the analysis' blocks were searched already, they are final)
                if (tri8) nnz = mb_encode_i8x8_trellis(k, L, c, cz, q8i, trc, lvw, cbp_luma);
                {
                    const uint32_t rz = *(const uint32_t *)(tile8 + zy * IT_STRIDE + zx);
                    if (commit) *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = rz;
                    if constexpr (RD) { ssd_y = ssd4_u8(cz, rz); psy_energy_z(rz, lane, en4, en8); if (dbr) *(uint32_t *)(dtile + zy * 16 + zx) = rz; }
                }
                if (!tri8) *(uint2 *)(lvw + lane * 4) = *(const uint2 *)(L.lv8 + lane * 4);
                if (lane < 16) lvw[X264GPU_LV_LUMA_DC + lane] = 0;
            } else if (e_type == X264GPU_MB_I4x4) {
                if (commit && lane < 16) recd.i4_mode[lane] = L.modes4[lane];
                nnz = IR.nnz4;
                const bool tri4 = TRL && !TRL2 && (trc.on & TR_I4) != 0;
                if (tri4) nnz = mb_encode_i4x4_trellis(k, L, c, cz, t4, q_li, trc, lvw);
                for (int i8 = 0; i8 < 4; i8++) if ((nnz >> (4 * i8)) & 15) cbp_luma |= 1 << i8;
                {
                    const uint32_t rz = *(const uint32_t *)(tile + zy * IT_STRIDE + zx);
                    if (commit) *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = rz;
                    if constexpr (RD) { ssd_y = ssd4_u8(cz, rz); psy_energy_z(rz, lane, en4, en8); if (dbr) *(uint32_t *)(dtile + zy * 16 + zx) = rz; }
                }
                if (!tri4) *(uint2 *)(lvw + lane * 4) = *(const uint2 *)(L.lv4 + lane * 4);
                if (lane < 16) lvw[X264GPU_LV_LUMA_DC + lane] = 0;
            } else {
                // x264_mb_encode_i16x16 (AC decimated as a whole in P slices)
                const int mode16 = IR.pred16;
                if (commit && lane == 0) recd.i16_mode = (uint8_t)(mode16 > PRED16_P ? PRED16_DC : mode16);
                const Pred16 pp = pred16_setup(L.nb, lane);
                int e[4], p[4], v[4];
                unpack4(cz, e); unpack4(pred16_row4(L.nb, pp, mode16, zx, zy), p);
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = e[t] - p[t];
                dct4_quad(v, lane);
                const int dcv = v[0];
                if (j4 == 0) v[0] = 0;
                if (TRL && (trc.on & TR_I16)) {
                    store_levels_scan(lvw + (lane >> 2) * 16, v, j4);
                    lds_sync();
                    trellis_run<1>(trc, lvw, 16, 16, c.qp, true, lane);
                    lds_sync();
                    load_levels_scan(lvw + (lane >> 2) * 16, v, j4);
                    lds_sync();
                } else quant4_row(v, q_li, j4);
                const unsigned mask = (unsigned)quad_or((int)scan_mask(v, j4));
                bool nz = mask != 0;
                if (pslice && k.dct_decimate) {
                    const int big = quad_or(any_big(v) ? 1 : 0);
                    const int sc = nz ? (big ? 9 : decimate_from_mask(mask, 1)) : 0;
                    if (wave_sum(j4 == 0 ? sc : 0) < 6) nz = false;
                }
                { int z[4] = { 0, 0, 0, 0 }; store_levels_scan(lvw + (lane >> 2) * 16, nz ? v : z, j4); }
                if (!nz) v[0] = v[1] = v[2] = v[3] = 0;
                dequant4_row(v, q_li, j4);
                const unsigned long long bal = __ballot(nz && j4 == 0);
                unsigned acn = 0;
#pragma unroll
                for (int b = 0; b < 16; b++) acn |= (unsigned)((bal >> (4 * b)) & 1) << b;
                int dc[4];
#pragma unroll
                for (int cc = 0; cc < 4; cc++) dc[cc] = __shfl(dcv, 4 * blkidx_of(cc, j4));
                had4x4_quad(dc, lane);
#pragma unroll
                for (int cc = 0; cc < 4; cc++) dc[cc] = (dc[cc] + 1) >> 1;
                if (TRL && (trc.on & TR_I16)) {
                    lds_sync();
                    if (lane < 4) store_levels_scan(lvw + X264GPU_LV_LUMA_DC, dc, j4);
                    lds_sync();
                    trellis_run<0>(trc, lvw + X264GPU_LV_LUMA_DC, 16, 1, c.qp, true, lane);
                    lds_sync();
                    load_levels_scan(lvw + X264GPU_LV_LUMA_DC, dc, j4);
                    lds_sync();
                } else
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) dc[cc] = quant_one(dc[cc], q_li.mf[0] >> 1, q_li.bias[0] << 1);
                const bool nzdc = quad_or((dc[0] | dc[1] | dc[2] | dc[3]) != 0 ? 1 : 0) != 0;
                if (lane < 4) store_levels_scan(lvw + X264GPU_LV_LUMA_DC, dc, j4);
                had4x4_quad(dc, lane);
                {
                    const int ls = q_li.dq[0], qb = c.qp / 6 - 6;
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) dc[cc] = dequant_one(dc[cc], ls, qb);
                }
                {
                    const int b = lane >> 2, bx = z_bx(b), by = z_by(b);
                    int t0 = __shfl(dc[0], by), t1 = __shfl(dc[1], by), t2 = __shfl(dc[2], by), t3 = __shfl(dc[3], by);
                    const int mine = bx == 0 ? t0 : bx == 1 ? t1 : bx == 2 ? t2 : t3;
                    if (j4 == 0) v[0] = nzdc ? mine : 0;
                }
                idct4_quad(v, lane);
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] += p[t];
                {
                    const uint32_t rz = pack4_clip(v);
                    if (commit) *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = rz;
                    if constexpr (RD) { ssd_y = ssd4_u8(cz, rz); psy_energy_z(rz, lane, en4, en8); if (dbr) *(uint32_t *)(dtile + zy * 16 + zx) = rz; }
                }
                nnz = acn | (nzdc ? 1u << 24 : 0);
                cbp_luma = acn ? 15 : 0;
            }
            if (lane >= 16 && lane < 24) lvw[408 + lane - 16] = 0;
            // chroma: prediction of the mode chosen above, residual
            {
                const int pl = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j4;
                const PredC pc = predc_setup(L.cnb[pl]);
                const uint2 fe = *(const uint2 *)(L.csrc + cyy * 16 + 2 * cx0);
                const uint32_t cenc = nv12_pick(fe.x, fe.y, pl), cpred = predc_row4(L.cnb[pl], pc, predc, ci, j4);
                if (commit && lane == 0) recd.chroma_mode = (uint8_t)(predc > PREDC_P ? PREDC_DC : predc);
                const uint32_t crec = mb_chroma_residual(cenc, cpred, q_ci, false, false, lane, lvw, nnz, cbp_chroma, TRL ? &trc : nullptr);
                if (commit) mb_store_chroma(ruv, k.rs, lane, crec);
                if constexpr (RD) { if (lane < 32) ssd_c = ssd4_u8(cenc, crec); }
            }
            if (commit) intra_count++;
            pf.mark(PH_ENC_INTRA);
        }
        if constexpr (RD) {
            if (dbr) {
                // x264_rd_cost_mb with b_deblock_rdo: x264_macroblock_deblock between the encode and the distortion
                const bool intra_c = e_type < X264GPU_MB_P_L0;
                const bool skipc = rec_type == X264GPU_MB_P_SKIP || rec_type == X264GPU_MB_B_SKIP || e_type == X264GPU_MB_P_SKIP || e_type == X264GPU_MB_B_SKIP;
                if (!skipc && !((e_part == D_16x16 && !cbp_luma && !intra_c) || c.qp <= 15 - min(k.alpha_off, k.beta_off) - max(k.chroma_qp_offset, 0))) {
                    unsigned nz16 = 0;          // bit y * 4 + x: the 4x4 block at (x, y) is coded (an 8x8-transform block flags all four)
#pragma unroll
                    for (int b = 0; b < 16; b++) {
                        const int x = (b & 1) + ((b >> 2) & 1) * 2, y = ((b >> 1) & 1) + (b >> 3) * 2;
                        const unsigned f = rd_t8cur ? (unsigned)(cbp_luma >> (b >> 2)) & 1u : (nnz >> b) & 1u;
                        nz16 |= f << (y * 4 + x);
                    }
                    lds_sync();
                    mb_deblock_rdo(dtile, lane, intra_c, rd_t8cur != 0, nz16, db_mvm, c.qp, k.alpha_off, k.beta_off);
                    const uint32_t rzf = *(const uint32_t *)(dtile + zy * 16 + zx);
                    ssd_y = ssd4_u8(cz, rzf); psy_energy_z(rzf, lane, en4, en8);
                }
            }
        }
        if constexpr (RD >= 2) {
            // ---- CABAC: the candidate priced on a copy of the slice's context variables; the finished macroblock moves them on ----
            STATE_UNIFORM();
            uni_all(rd_t8cur, db_mvm);
            lds_sync();
            struct RdMark { Prof &p; __device__ ~RdMark() {
#ifdef MB_PROF_RD
                p.mark(15);           // MB_PROF_RD builds: slot 15 = cycles of the CABAC pricing / evolution instead of the staging count
#endif
            } } rd_mark{ pf };
            pf.begin2();
            CabIn ci;
            ci.pslice = pslice; ci.left = left; ci.top = top; ci.nref = c.nref; ci.t8mode = k.dct8x8;
            ci.bslice = BS; ci.nref1 = k.nref1; ci.buse = euse; ci.b_r0 = ecfg.r0; ci.b_x0 = ecfg.x0; ci.b_y0 = ecfg.y0; ci.b_r1 = ecfg.r1; ci.b_x1 = ecfg.x1; ci.b_y1 = ecfg.y1;
            ci.lamvd1 = 0; ci.tamvd1 = 0;
            if constexpr (BS) {
                ci.lamvd1 = (unsigned long long)rl(cnbv, 12) | ((unsigned long long)rl(cnbv, 13) << 32);
                ci.tamvd1 = (unsigned long long)rl(cnbv, 14) | ((unsigned long long)rl(cnbv, 15) << 32);
            }
            ci.type = rec_type; ci.part = e_part; ci.cbp_luma = cbp_luma; ci.cbp_chroma = cbp_chroma; ci.nnz = nnz;
            ci.i16mode = IR.pred16 > PRED16_P ? PRED16_DC : IR.pred16; ci.cmode = predc > PREDC_P ? PREDC_DC : predc;
            ci.qp = c.qp; ci.last_qp = last_qp; ci.last_dqp = last_dqp;
            {
                const uint32_t l0 = rl(cnbv, 0), l1 = rl(cnbv, 1), l6 = rl(cnbv, 2), l11 = rl(cnbv, 3), t0 = rl(cnbv, 4), t1 = rl(cnbv, 5), t6 = rl(cnbv, 6), t11 = rl(cnbv, 7);
                ci.ltype = l0 & 255; ci.lcmode = (l0 >> 16) & 255; ci.lcbp_luma = l1 & 255; ci.lcbp_chroma = (l1 >> 8) & 255; ci.lt8 = l6 >> 24; ci.lnnz = l11;
                ci.ttype = t0 & 255; ci.tcmode = (t0 >> 16) & 255; ci.tcbp_luma = t1 & 255; ci.tcbp_chroma = (t1 >> 8) & 255; ci.tt8 = t6 >> 24; ci.tnnz = t11;
                ci.lamvd = (unsigned long long)rl(cnbv, 8) | ((unsigned long long)rl(cnbv, 9) << 32);
                ci.tamvd = (unsigned long long)rl(cnbv, 10) | ((unsigned long long)rl(cnbv, 11) << 32);
            }
            ci.t8 = rd_t8cur;
            ci.pm = 0; ci.pm_b0 = 0; ci.pm_b1 = -1; ci.pm_dx = ci.pm_dy = ci.pm_sx = ci.pm_sy = 0; ci.pm_nnzc = 0;
            ci.pm_lists = 1; ci.pm_dx1 = ci.pm_dy1 = ci.pm_sx1 = ci.pm_sy1 = 0;
            if (rd_run && !commit) {
                int cost = 0;
                unsigned long long cost64 = 0;          // part costs (RD refinement): 8 more bits than x264_rd_cost_mb's
                int dist = wave_sum(ssd_y);
                int src_e4 = fenc_e4, src_e8 = fenc_e8;
                if constexpr (REF) { if (part_pass) { src_e4 = rf_pk == 2 ? 0 : rf_f4; src_e8 = rf_pk == 2 ? 0 : rf_f8; } }      // (an Intra_4x4 block's psy term is in its distortion already)
                if (k.psy_rd_q8) {
                    const int e4 = wave_sum(en4) >> 1, e8 = wave_sum(en8) >> 2;      // pixel_hadamard_ac of the reconstruction (16x16, or the part)
                    dist += (((abs(e4 - src_e4) + abs(e8 - src_e8)) >> 1) * k.psy_rd_q8 * c.lambda + 128) >> 8;
                }
                dist += (int)(((long long)wave_sum(ssd_c) * chroma_l2off + 128) >> 8);
                if (!part_pass && (rec_type == X264GPU_MB_P_SKIP || rec_type == X264GPU_MB_B_SKIP)) cost = dist + ((lambda2 + 128) >> 8);
                else {
                    if constexpr (REF) {
                        if (part_pass && rf_pk == 1) {
                            // partition_size_cabac: the part's vector difference against the |mvd| of the parts refined before / the neighbours
                            const int x8 = rf_b0 & 1, y8 = rf_b0 >> 1;
                            ci.pm = 1; ci.pm_b0 = rf_b0; ci.pm_b1 = rf_b1; ci.pm_nnzc = rf_nnzc; ci.t8 = e_t8 != 0;
                            ci.pm_lists = 1;
                            if constexpr (BS) {
                                // a B part: the vector difference(s) of the list(s) it uses (rf_l 0 / 1 / 2 both); no sub_mb_type bins
                                ci.pm_lists = rf_l == 2 ? 3 : 1 << rf_l;
                                ci.pm_dx = rf_cx - rf_mvp0x; ci.pm_dy = rf_cy - rf_mvp0y; ci.pm_dx1 = rf_c1x - rf_mvp1x; ci.pm_dy1 = rf_c1y - rf_mvp1y;
                            } else { ci.pm_dx = rf_cx - rf_mvpx; ci.pm_dy = rf_cy - rf_mvpy; }
                            for (int comp = 0; comp < 2; comp++) {
                                const int la = x8 > 0 ? (int)((rf_amvd >> (8 * ((y8 * 2 + x8 - 1) * 2 + comp))) & 255) : left ? (int)((ci.lamvd >> (8 * ((y8 * 2 + 1) * 2 + comp))) & 255) : 0;
                                const int ta = y8 > 0 ? (int)((rf_amvd >> (8 * (((y8 - 1) * 2 + x8) * 2 + comp))) & 255) : top ? (int)((ci.tamvd >> (8 * ((2 + x8) * 2 + comp))) & 255) : 0;
                                if (comp) ci.pm_sy = la + ta; else ci.pm_sx = la + ta;
                                if constexpr (BS) {
                                    const int la1 = x8 > 0 ? (int)((rf_amvd1 >> (8 * ((y8 * 2 + x8 - 1) * 2 + comp))) & 255) : left ? (int)((ci.lamvd1 >> (8 * ((y8 * 2 + 1) * 2 + comp))) & 255) : 0;
                                    const int ta1 = y8 > 0 ? (int)((rf_amvd1 >> (8 * (((y8 - 1) * 2 + x8) * 2 + comp))) & 255) : top ? (int)((ci.tamvd1 >> (8 * ((2 + x8) * 2 + comp))) & 255) : 0;
                                    if (comp) ci.pm_sy1 = la1 + ta1; else ci.pm_sx1 = la1 + ta1;
                                }
                            }
                        }
                        if (part_pass && rf_pk >= 2) {
                            // partition_i4x4_size_cabac / partition_i8x8_size_cabac / chroma_size_cabac
                            ci.pm = rf_pk; ci.pm_b0 = rf_i; ci.pm_b1 = -1; ci.pm_nnzc = rf_nnzc;
                            ci.type = rf_pk == 2 ? X264GPU_MB_I4x4 : rf_pk == 3 ? X264GPU_MB_I8x8 : e_type;
                            if (rf_pk == 3) ci.t8 = 1;
                            if (rf_pk == 4) ci.cmode = rf_cm > PREDC_P ? PREDC_DC : rf_cm;
                        }
                    }
                    Cab tmp = cab;
                    tmp.f8 = 0; tmp.f8v = 0;
                    int dq;
                    unsigned long long av1;
                    ci.size = true;
                    pf.mark2(27);
                    cab_mb(tmp, cab_modelv, lane, ci, S, rd_lvs, L.modes4, L.modes8, L.nmodes, c.mbx, c.sy, dq, av1, k.ctab, pf);
                    pf.begin2();
                    if constexpr (REF) { if (part_pass && rf_pk == 4) dist = wave_sum(ssd_c); }          // rd_cost_chroma: the two planes' SSD as it is
                    const int l2p = REF && part_pass && rf_pk == 4 ? c_lambda2_tab[c.qpc] : lambda2;
                    if (part_pass) cost64 = ((unsigned long long)(unsigned)dist << 8) + (((unsigned long long)cab_total(tmp) * (unsigned long long)l2p + 128) >> 8);
                    else cost = dist + (int)(((unsigned long long)cab_total(tmp) * (unsigned long long)lambda2 + 32768) >> 16);
                }
                cost = uni(cost);          // (wave-uniform, but the bit count came through vector code: say so, or every RD cost of the macroblock lives in a vector register)
                if constexpr (REF) {
                    if (!part_pass) {
                        // what this whole-macroblock encode leaves in x264's non_zero_count cache (flags; an 8x8 transform block sets its four entries alike)
                        const unsigned ex = ((cbp_luma & 1) ? 0x000fu : 0) | ((cbp_luma & 2) ? 0x00f0u : 0) | ((cbp_luma & 4) ? 0x0f00u : 0) | ((cbp_luma & 8) ? 0xf000u : 0);
                        const bool skc = rec_type == X264GPU_MB_P_SKIP || rec_type == X264GPU_MB_B_SKIP;
                        rf_nnzc = skc ? 0u : (rd_t8cur ? ex : nnz & 0xffffu & ex) | (cbp_chroma == 2 ? nnz & 0x00ff0000u : 0u);
                    }
                    if (rf_on) { rf_cost = part_pass ? cost64 : (unsigned long long)(unsigned)cost; continue; }
                }
                if constexpr (BS) {
                    // x264_mb_analyse_b_rd / _transform_rd / x264_intra_rd: where the candidate's cost goes
                    if (rd_ph == 1 || rd_ph == 7) rd_dir = cost;
                    else if (rd_ph == 2 || rd_ph == 8) rd_l0 = cost;
                    else if (rd_ph == 3 || rd_ph == 9) rd_l1 = cost;
                    else if (rd_ph == 4 || rd_ph == 10) rd_bi = cost;
                    else if (rd_ph == 11) rd_8x8 = cost;
                    else if (rd_ph == 12) rd_16x8 = cost;
                    else if (rd_ph == 13) rd_8x16 = cost;
                    else if (rd_ph == 14) { if (rd_best >= cost) { if (rd_best > 0) rd_satd_inter = (int)((long long)rd_satd_inter * cost / rd_best); rd_best = cost; rd_t8 = 1; } }
                    else if (rd_ph == 15) rd_i16 = cost;
                    else if (rd_ph == 16) rd_i4 = cost;
                    else if (rd_ph == 17) { rd_i8 = cost; rf_cbp_i8 = cbp_luma; }
                } else
                if (rd_ph == 0) {
                    rd16 = cost;
                    if (rec_type == X264GPU_MB_P_SKIP) {          // the 16x16 result is the skip vector and nothing would be coded: P_SKIP, analysis over
                        rd_best = cost; rd_part = D_16x16; rd_t8 = 0; rd_i16 = rd_i4 = rd_i8 = MB_COST_MAX; rd_ph = 8 - 1; rd_skip16 = true;
                    }
                } else if (rd_ph == 1) rd16x8 = cost;
                else if (rd_ph == 2) rd8x16 = cost;
                else if (rd_ph == 3) rd8x8 = cost;
                else if (rd_ph == 4) { if (rd_best >= cost) { if (rd_best > 0) rd_satd_inter = (int)((long long)rd_satd_inter * cost / rd_best); rd_best = cost; rd_t8 = 1; } }
                else if (rd_ph == 5) rd_i16 = cost;
                else if (rd_ph == 6) rd_i4 = cost;
                else if (rd_ph == 7) { rd_i8 = cost; rf_cbp_i8 = cbp_luma; }
                rd_ph++;
                pf.mark2(28);
                continue;
            }
            // the final macroblock: its levels go out, its bins move the slice's context variables on, its |mvd| stay for the neighbours
            for (int i = lane; i < X264GPU_MB_LEVELS / 2; i += 64) ((uint32_t *)lv)[i] = ((const uint32_t *)rd_lvs)[i];
            if (rdon) {
                int dq;
                unsigned long long av1;
                ci.size = false;
                const unsigned long long av = cab_mb(cab, cab_modelv, lane, ci, S, rd_lvs, L.modes4, L.modes8, L.nmodes, c.mbx, c.sy, dq, av1, k.ctab, pf);
                last_dqp = dq;
                if (lane < 2) ((uint32_t *)(k.amvd + ((size_t)s * k.nmb + mbi) * (BS ? 16 : 8)))[lane] = (uint32_t)(av >> (32 * lane));
                if (BS && lane >= 2 && lane < 4) ((uint32_t *)(k.amvd + ((size_t)s * k.nmb + mbi) * 16))[lane] = (uint32_t)(av1 >> (32 * (lane - 2)));
            }
        }
        if constexpr (RD == 1) {
            // ---- bits of the macroblock layer as CAVLC would write them (x264_macroblock_size_cavlc), and the blocks' total_coeff ----
            int mb_bits = 0, my_tc = 0;
            lds_sync();
            if (rd_run || (rdon && commit)) {
                const bool i16 = e_type == X264GPU_MB_I16x16, intra = e_type < X264GPU_MB_P_L0;
                // residual: one lane per block — 0..15 luma (block order), 16..23 chroma AC (plane * 4 + block), 24 luma DC, 25 / 26 chroma DC
                const int16_t *myl = rd_lvs;
                int myn = 0;
                bool coded = false;
                if (lane < 16) { coded = (cbp_luma >> (lane >> 2)) & 1; myl = rd_lvs + lane * 16 + (i16 ? 1 : 0); myn = i16 ? 15 : 16; }
                else if (lane < 24) { coded = cbp_chroma == 2; myl = rd_lvs + X264GPU_LV_CHROMA_AC + (lane - 16) * 16 + 1; myn = 15; }
                else if (lane == 24) { coded = i16; myl = rd_lvs + X264GPU_LV_LUMA_DC; myn = 16; }
                else if (lane < 27) { coded = cbp_chroma != 0; myl = rd_lvs + X264GPU_LV_CHROMA_DC + (lane - 25) * 4; myn = 4; }
                const bool skp = rec_type == X264GPU_MB_P_SKIP || rec_type == X264GPU_MB_B_SKIP;
                if (skp) coded = false;
                if (coded && lane < 24) for (int i = 0; i < myn; i++) my_tc += myl[i] != 0;
                // nC: average of the left and upper blocks' totals where they exist (inside the macroblock: other lanes; outside: rd_ntc)
                int na = -1, nb = -1;
                {
                    const bool chroma = lane >= 16 && lane < 24;
                    const int b = lane == 24 ? 0 : lane & 15, ci = (lane - 16) & 3, cb = 16 + ((lane - 16) & 4);
                    const int bx = chroma ? ci & 1 : z_bx(b), by = chroma ? ci >> 1 : z_by(b);
                    const int l_in = chroma ? cb + by * 2 : blkidx_of(bx - 1, by), t_in = chroma ? cb + bx : blkidx_of(bx, by - 1);
                    const int l_out = chroma ? cb + by * 2 + 1 : blkidx_of(3, by), t_out = chroma ? cb + 2 + bx : blkidx_of(bx, 3);
                    const int tl = __shfl(my_tc, bx > 0 ? l_in : 0), tt = __shfl(my_tc, by > 0 ? t_in : 0);
                    if (lane < 25) {
                        if (bx > 0 && lane != 24) na = tl; else if (left) na = rd_ntc[0][l_out];
                        if (by > 0 && lane != 24) nb = tt; else if (top) nb = rd_ntc[1][t_out];
                    }
                }
                const int nC = lane >= 25 ? -1 : (na >= 0 && nb >= 0 ? (na + nb + 1) >> 1 : na >= 0 ? na : nb >= 0 ? nb : 0);
                if (rd_run && !commit && !skp) {
                    mb_bits = wave_sum(coded ? cavlc_block_bits(myl, myn, nC) : 0);
                    // header
                    if (BS && !intra) {
                        // cavlc_mb_header_b as a count: mb_type of Table 7-14, the sub_mb_types of B_8x8 (direct 0, L0 1, L1 2, Bi 3), te(v) reference indices
                        // of list 0 then list 1, vector differences of list 0 then list 1, each over the partitions that use the list
                        if (e_type == X264GPU_MB_B_DIRECT) mb_bits += 1;
                        else {
                            const int np = e_part == D_16x16 ? 1 : e_part == D_8x8 ? 4 : 2;
                            auto b8_of = [&](int kp) { return e_part == D_8x8 ? kp : e_part == D_16x8 ? 2 * kp : kp; };
                            if (e_part == D_8x8) { mb_bits += bs_size_ue_d(22); for (int kp = 0; kp < 4; kp++) { const int u = (int)((euse >> (2 * kp)) & 3); mb_bits += bs_size_ue_d(u == 3 ? 0 : 1 + u); } }
                            else if (e_part == D_16x16) mb_bits += bs_size_ue_d(1 + (int)(euse & 3));
                            else {
                                const int u0 = (int)(euse & 3), u1 = (int)((e_part == D_16x8 ? euse >> 4 : euse >> 2) & 3);
                                const int pair = u0 == 0 ? (u1 == 0 ? 0 : u1 == 1 ? 2 : 4) : u0 == 1 ? (u1 == 0 ? 3 : u1 == 1 ? 1 : 5) : 6 + u1;
                                mb_bits += bs_size_ue_d(4 + 2 * pair + (e_part == D_8x16 ? 1 : 0));
                            }
                            const int sc0 = S.cref, sc1 = S.cmvx, sc2 = S.cmvy;
                            for (int l = 0; l < 2; l++) {
                                const int nr = l ? k.nref1 : k.nref, go = 16 * l;
                                for (int kp = 0; kp < np; kp++) {
                                    const int b8 = b8_of(kp), u = (int)((euse >> (2 * b8)) & 3);
                                    if (nr > 1 && !(u == 3 || u == 1 - l)) mb_bits += nr == 2 ? 1 : bs_size_ue_d(rl(l ? ecfg.r1 : ecfg.r0, b8 * 16));
                                }
                                if ((lane == 5 + go || lane == 6 + go || lane == 9 + go || lane == 10 + go)) S.cref = -2;
                                for (int kp = 0; kp < np; kp++) {
                                    const int b8 = b8_of(kp), u = (int)((euse >> (2 * b8)) & 3), x8 = b8 & 1, y8 = b8 >> 1;
                                    const int w8 = e_part == D_16x16 || e_part == D_16x8 ? 2 : 1, h8 = e_part == D_16x16 || e_part == D_8x16 ? 2 : 1;
                                    const int r = rl(l ? ecfg.r1 : ecfg.r0, b8 * 16), vx = rl(l ? ecfg.x1 : ecfg.x0, b8 * 16), vy = rl(l ? ecfg.y1 : ecfg.y0, b8 * 16);
                                    const int g0 = (y8 + 1) * 4 + x8 + 1 + go;
                                    const bool mine = lane == g0 || (w8 == 2 && lane == g0 + 1) || (h8 == 2 && lane == g0 + 4) || (w8 == 2 && h8 == 2 && lane == g0 + 5);
                                    if (!(u == 3 || u == 1 - l)) {
                                        S.cref = mine ? r : S.cref;
                                        int px, py;
                                        mb_predict_mv(S, e_part, x8, y8, w8, r, px, py, go);
                                        mb_bits += bs_size_se_d(vx - px) + bs_size_se_d(vy - py);
                                    }
                                    S.cref = mine ? (r >= 0 ? r : -1) : S.cref; S.cmvx = mine ? (r >= 0 ? vx : 0) : S.cmvx; S.cmvy = mine ? (r >= 0 ? vy : 0) : S.cmvy;
                                }
                            }
                            S.cref = sc0; S.cmvx = sc1; S.cmvy = sc2;
                        }
                        mb_bits += cavlc_cbp_bits(cbp_luma | (cbp_chroma << 4), true);
                        if (k.dct8x8 && cbp_luma) mb_bits += 1;
                    } else
                    if (!intra) {
                        const int np = e_part == D_16x16 ? 1 : e_part == D_8x8 ? 4 : 2;
                        mb_bits += bs_size_ue_d(e_part) + (e_part == D_8x8 ? 4 : 0);
                        const int sc0 = S.cref, sc1 = S.cmvx, sc2 = S.cmvy;
                        if (lane == 5 || lane == 6 || lane == 9 || lane == 10) S.cref = -2;
                        for (int kp = 0; kp < np; kp++) {
                            const int x8 = e_part == D_8x16 ? kp : e_part == D_8x8 ? kp & 1 : 0, y8 = e_part == D_16x8 ? kp : e_part == D_8x8 ? kp >> 1 : 0;
                            const int w8 = e_part == D_16x16 || e_part == D_16x8 ? 2 : 1, h8 = e_part == D_16x16 || e_part == D_8x16 ? 2 : 1;
                            const int slot = e_part == D_16x16 ? ME_16 : e_part == D_16x8 ? ME_16x8 + kp : e_part == D_8x16 ? ME_8x16 + kp : ME_8 + kp;
                            const int r = rl(S.ref, slot), vx = rl(S.mvx, slot), vy = rl(S.mvy, slot);
                            if (c.nref > 1) mb_bits += c.nref == 2 ? 1 : bs_size_ue_d(r);
                            const int g0 = (y8 + 1) * 4 + x8 + 1;
                            const bool mine = lane == g0 || (w8 == 2 && lane == g0 + 1) || (h8 == 2 && lane == g0 + 4) || (w8 == 2 && h8 == 2 && lane == g0 + 5);
                            S.cref = mine ? r : S.cref;                       // the partition's reference is cached before its vector is predicted
                            int px, py;
                            mb_predict_mv(S, e_part, x8, y8, w8, r, px, py);
                            mb_bits += bs_size_se_d(vx - px) + bs_size_se_d(vy - py);
                            S.cmvx = mine ? vx : S.cmvx; S.cmvy = mine ? vy : S.cmvy;
                        }
                        S.cref = sc0; S.cmvx = sc1; S.cmvy = sc2;
                        mb_bits += cavlc_cbp_bits(cbp_luma | (cbp_chroma << 4), true);
                        if (k.dct8x8 && cbp_luma) mb_bits += 1;
                    } else {
                        const int off = BS ? 23 : pslice ? 5 : 0;
                        if (i16) { const int m16 = IR.pred16 > PRED16_P ? PRED16_DC : IR.pred16; mb_bits += bs_size_ue_d(off + 1 + m16 + 4 * cbp_chroma + (cbp_luma ? 12 : 0)); }
                        else {
                            mb_bits += bs_size_ue_d(off) + (k.dct8x8 ? 1 : 0);
                            int fb = 0;
                            if (e_type == X264GPU_MB_I4x4) { if (lane < 16) fb = i4_pred_mode(L.nmodes, c.mbx, c.sy, lane, L.modes4) == L.modes4[lane] ? 1 : 4; }
                            else if (lane < 4) fb = i4_pred_mode(L.nmodes, c.mbx, c.sy, lane * 4, L.modes8) == L.modes8[lane * 4] ? 1 : 4;
                            mb_bits += wave_sum(fb);
                            mb_bits += cavlc_cbp_bits(cbp_luma | (cbp_chroma << 4), false);
                        }
                        mb_bits += bs_size_ue_d(predc > PREDC_P ? PREDC_DC : predc);
                    }
                    if (cbp_luma || cbp_chroma || i16) mb_bits += bs_size_se_d(c.qp - last_qp);
                }
            }
            if (rd_run && !commit) {
                // ---- x264_rd_cost_mb: SSD of luma (+ the psy-rd energy term) and chroma (scaled by the chroma lambda offset) + lambda2 x bits ----
                int dist = wave_sum(ssd_y);
                if (k.psy_rd_q8) {
                    const int e4 = wave_sum(en4) >> 1, e8 = wave_sum(en8) >> 2;      // pixel_hadamard_ac_16x16 of the reconstruction
                    dist += (((abs(e4 - fenc_e4) + abs(e8 - fenc_e8)) >> 1) * k.psy_rd_q8 * c.lambda + 128) >> 8;
                }
                dist += (int)(((long long)wave_sum(ssd_c) * chroma_l2off + 128) >> 8);
                const int cost = rec_type == X264GPU_MB_P_SKIP || rec_type == X264GPU_MB_B_SKIP ? dist + ((lambda2 + 128) >> 8) : dist + (int)(((long long)mb_bits * lambda2 + 128) >> 8);
                if constexpr (BS) {
                    // x264_mb_analyse_b_rd / _transform_rd / x264_intra_rd: where the candidate's cost goes (as in the CABAC instantiations)
                    if (rd_ph == 1 || rd_ph == 7) rd_dir = cost;
                    else if (rd_ph == 2 || rd_ph == 8) rd_l0 = cost;
                    else if (rd_ph == 3 || rd_ph == 9) rd_l1 = cost;
                    else if (rd_ph == 4 || rd_ph == 10) rd_bi = cost;
                    else if (rd_ph == 11) rd_8x8 = cost;
                    else if (rd_ph == 12) rd_16x8 = cost;
                    else if (rd_ph == 13) rd_8x16 = cost;
                    else if (rd_ph == 14) { if (rd_best >= cost) { if (rd_best > 0) rd_satd_inter = (int)((long long)rd_satd_inter * cost / rd_best); rd_best = cost; rd_t8 = 1; } }
                    else if (rd_ph == 15) rd_i16 = cost;
                    else if (rd_ph == 16) rd_i4 = cost;
                    else if (rd_ph == 17) rd_i8 = cost;
                } else
                if (rd_ph == 0) {
                    rd16 = cost;
                    if (rec_type == X264GPU_MB_P_SKIP) {          // the 16x16 result is the skip vector and nothing would be coded: P_SKIP, analysis over
                        rd_best = cost; rd_part = D_16x16; rd_t8 = 0; rd_i16 = rd_i4 = rd_i8 = MB_COST_MAX; rd_ph = 8 - 1; rd_skip16 = true;
                    }
                } else if (rd_ph == 1) rd16x8 = cost;
                else if (rd_ph == 2) rd8x16 = cost;
                else if (rd_ph == 3) rd8x8 = cost;
                else if (rd_ph == 4) { if (rd_best >= cost) { if (rd_best > 0) rd_satd_inter = (int)((long long)rd_satd_inter * cost / rd_best); rd_best = cost; rd_t8 = 1; } }
                else if (rd_ph == 5) rd_i16 = cost;
                else if (rd_ph == 6) rd_i4 = cost;
                else if (rd_ph == 7) rd_i8 = cost;
                rd_ph++;
                continue;
            }
            // the final macroblock: its levels go out, its blocks' totals stay for the neighbours' nC
            for (int i = lane; i < X264GPU_MB_LEVELS / 2; i += 64) ((uint32_t *)lv)[i] = ((const uint32_t *)rd_lvs)[i];
            if (rdon && lane < 24) k.tc[((size_t)s * k.nmb + mbi) * 24 + lane] = (uint8_t)my_tc;
        }
        {   // h->mb.i_last_qp: a macroblock that sends mb_qp_delta sets it (an I16x16 with nothing coded never raises it)
            const bool i16e = rec_type == X264GPU_MB_I16x16;
            if (rec_type != X264GPU_MB_P_SKIP && (cbp_luma || cbp_chroma || i16e) && !(i16e && !cbp_luma && !cbp_chroma && !((nnz >> 24) & 1) && c.qp > last_qp)) last_qp = c.qp;
        }
        break;
        }
        // (rec_type / nnz / cbp of the final pass are needed below: they live outside the loop)
        if (lane == 0) { recd.nnz = nnz; recd.cbp_luma = (uint8_t)cbp_luma; recd.cbp_chroma = (uint8_t)cbp_chroma; recd.type = (uint8_t)rec_type; mbtype_cur[mbi] = (uint8_t)rec_type; }
        lds_sync();
        if (lane < 16) ((uint32_t *)(mbs + mbi))[lane] = ((const uint32_t *)&L.rec)[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        pf.mark(PH_STORE);
    }
    if (BS && k.dscore && k.direct_auto && lane == 0) { int *d = k.dscore + ((size_t)s * nsl + blockIdx.y) * 2; d[0] = ds_t; d[1] = ds_s; }
    if (PS && k.sl_stat && lane == 0) { int *st = k.sl_stat + ((size_t)s * nsl + blockIdx.y) * 4; st[0] = intra_count - L.slw[2]; st[1] = L.slw[0]; st[2] = L.slw[1]; }
    if (k.wtime && lane == 0 && blockIdx.y == 0) k.wtime[s] = (unsigned)min((unsigned long long)(__builtin_readcyclecounter() - wt0) >> 6, 0xffffffffull);
    if constexpr (RD >= 2) {
        if (k.cab_out) { uint32_t *o = k.cab_out + ((size_t)s * (k.slices > 1 ? k.slices : 1) + blockIdx.y) * 192; o[lane] = cab.a; o[64 + lane] = cab.r; o[128 + lane] = cab.r8; }
    }
#ifdef MB_PROF
    if (lane == 0 && k.prof) for (int i = 0; i < 32; i++) k.prof[(size_t)s * 32 + i] = pf.acc[i];
#endif
}

// The launch of the macroblock loop: one workgroup of one wavefront per stream and slice.  (The single workgroups of several streams' launches — a session's pictures
// in flight — run side by side at a lone picture's rate, 1.0 - 1.1 s against 0.97 s: the dispatcher does not stack them on one SIMD; profiles/r06_inflight_kernel_timeline.txt)
static inline void mb_launch(void (*kern)(EncK), const EncK &k, int streams, hipStream_t st)
{
    hipLaunchKernelGGL(kern, dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
}

}  // namespace x264gpu
