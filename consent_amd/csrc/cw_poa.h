/*
 * cw_poa.h -- partial-order alignment of one segment pile per wavefront (A4d).
 *
 * One 64-lane wave owns one task from the list the index kernel emitted.  Same code, three memory tiers:
 *   S        (cw_poa_kernel)        graph + DP matrix in LDS                              small segments (most tasks)
 *   M1/M2/L  (cw_poa_slab_kernel)   graph in LDS, matrix in an L2-resident per-wave slab   long segments, 3 size classes
 *   G        (cw_poa_big_kernel)    everything in a per-wave global slab                   the rare huge graph
 * A task that outgrows its tier is handed to the next one and redone there from scratch.
 *
 * Per member of the pile (policies: include/cw_policy.h):
 *   row metadata  one lane per rank: base, predecessor rows (first one inline, the rest in a CSR list)
 *   DP fill       lanes = sequence positions; rows = nodes in rank order; the horizontal gap recurrence
 *                 H[i][j] = max(H[i][j], H[i][j-1]+g) is a DPP inclusive prefix-max of H[i][j]-j*g (linear gaps);
 *                 rows of <= 128 columns keep the previous row in registers so a chain of nodes never waits on
 *                 memory; the next row's metadata is fetched while the current row computes
 *   traceback     wave-uniform walk (diagonal, then vertical, then horizontal); all three candidate cells and
 *                 the predecessor's metadata are requested together, one memory round trip per step
 *   merge         one lane per sequence position: resolve node / sibling / fresh node, assign fresh ids and
 *                 edge ids by ballot prefix, place fresh nodes in the rank order with one prefix sum
 *                 (equivalent to the sequential insertions of cw_policy.h "rank order")
 *   consensus     column-majority vote, one lane per rank, ordered compaction by ballot
 * All arithmetic is integer; int16 cells in tiers S and M (|score| <= 8*(nodes+len) < 2^15 under their caps),
 * int32 cells in tier G.
 */
#ifndef CW_POA_H
#define CW_POA_H

#include "cw_device.h"

/* tier S: graph and DP matrix in LDS (per wave) */
#define CW_POA_NC 128   /* nodes            */
#define CW_POA_EC 384   /* edges            */
#define CW_POA_LC 127   /* member length    */
#define CW_POA_HC 2048  /* DP cells (int16) */
#define CW_POA_DC 0     /* traceback direction words: (rows x 64-column chunks) pairs of u64 */
#define CW_POA_WAVES 4
#define CW_POA_CHUNK_S 8   /* tasks a yielding wave of tier S runs before it ends (~0.3 ms each) */
#define CW_POA_CHUNK_M1 2  /* ... of tier M1 (~1 ms each); tiers M2 and L: one */
/* tiers M1 / M2 / L: graph in LDS, DP matrix (int16) in a per-wave global slab that stays L2 / Infinity-Cache
   resident.  Direction words are off there: they cost more occupancy than they save (measured). */
#define CW_POAM1_NC 256
#define CW_POAM1_EC 640
#define CW_POAM1_LC 255
#ifndef CW_POAM1_WAVES
#define CW_POAM1_WAVES 4
#endif
#define CW_POAM1_ROUTE 208 /* tasks expected to grow beyond this many nodes go to M2 at once (fewer late hand-overs) */
#define CW_POAM2_NC 512
#define CW_POAM2_EC 1280
#define CW_POAM2_LC 511
#ifndef CW_POAM2_WAVES
#define CW_POAM2_WAVES 3
#endif
#define CW_POAM2_ROUTE CW_POAM2_NC
#define CW_POAL_NC 1536
#define CW_POAL_EC 4096
#define CW_POAL_LC 1023
#define CW_POAL_WAVES 1
/* tier G: everything in a per-wave global slab (int32 cells) */
#define CW_POAB_NC 4096 /* (2048 until round 5: the graphs that stopped a window in the fuzzers were all this tier's) */
#define CW_POAB_EC 8192
#define CW_POAB_LC 2047 /* round 6 (1023 through round 5): a window of 1500 bases whose chain is sparse has pieces of more than 1023 bases; the cell budget is what it was */
#define CW_POAB_HC ((CW_POAB_NC + 1) * 1024) /* cells of the int32 matrix: rows x (bases + 1) of an alignment must fit (4096 nodes against 1023 bases, 2047 against 2047) */

/* bytes of the graph part of a slab when every array lives in it (tiers S and G) */
#define CW_POA_EW_BYTES(EC) (CW_CONS_HEAVIEST_BUNDLE ? 2 * (EC) : 0) /* edge weights, kept only under the heaviest-bundle policy (cw_policy.h) */
#define CW_POA_SW (CW_POA_MODE == CW_POA_MODE_SW) /* local mode (cw_policy.h, round 5): rows 0 and column 0 are 0, no cell below 0, the end cell is the best cell anywhere, the walk
                                                    stops at a cell of value 0.  Runs on the matrix paths only: no recorded decisions (CW_POA_CODES 0, CW_Q_CODES 0), no direction words */
#define CW_POA_OV (CW_POA_MODE == CW_POA_MODE_OV || CW_POA_SW) /* overlap mode (cw_policy.h): column 0 is free, the end cell is the best cell of a sink row, the walk stops in column 0
                                                                  (the local mode shares all three, with "any row" for "a sink's row") */
#if CW_POA_SW || CW_POA_AFFINE /* (the affine gap model: cw_poa_a.h, the global-memory tier only) */
#define CW_POA_CODES 0
#define CW_Q_CODES 0
#endif
#ifndef CW_POA_VPROBE
#define CW_POA_VPROBE 1 /* the tile traceback looks down the column when it is inside a long vertical run (poa_run, poa_trace_c) */
#endif
#define CW_POA_GRAPH_BYTES(NC, EC, LC) (((NC) * 29 + (EC) * 6 + CW_POA_EW_BYTES(EC) + 7 * ((LC) + 1) + 64 + 15) / 16 * 16)

/* Slab tiers (M1 / M2 / L): LDS holds only what the fill and the traceback read ("hot": rank metadata, predecessor lists, first
   predecessors, rank <-> node maps, in-edge heads and degrees, bases, sequence ranks); what only the rank bookkeeping before a fill
   and the merge touch -- aligned-node lists, in-edge lists (source, next) and tails, coverage counts, the scratch of the rank placement,
   the per-position results of the merge passes -- lives in the wave's global slab ("cold"), a few dependent reads per node there.
   That is what lets more waves share a CU's LDS where the time goes: M1 6.4 KB per wave (five work-groups per CU), M2 12.9 KB (three
   waves in 38 KB), L 37 KB (it fits the holes the other tiers leave). */
#define CW_POA_HOT2_BYTES(NC, EC, LC) (((NC) * 17 + (EC) * 2 + 3 * ((LC) + 1) + 64 + 15) / 16 * 16)
#ifndef CW_RING
#define CW_RING 16                               /* cw_poa_c.h: rows of the LDS ring of the recorded-decision fill (a power of two).  8 rows measured: tiers S / M1 +4 % (more rows
                                                    come back from the slab), tier M2 -10 % (the LDS they give up), step unchanged */
#endif
#define CW_POA_RING_BYTES (CW_RING * 64 * 2)
#define CW_POA_HOT2C_BYTES(NC, EC, LC) (CW_POA_HOT2_BYTES(NC, EC, LC) + (4 * (NC) + 15) / 16 * 16 + CW_POA_RING_BYTES + (((NC) / 32 + 2) * 4 + 15) / 16 * 16) /* + the chain tables p2/p4, the row ring and the slab-row flags of cw_poa_c.h (tiers M1 / M2) */
#ifndef CW_M2_CODES
#define CW_M2_CODES 0 /* tier M2 on the recorded-decision path (measured slower: 85 % of its rows are linear) */
#endif
/* LDS per wave of slab tier T (1 = M1, 2 = M2): the ring and the flags only where the recorded-decision fill runs (tier M2 carried their 2.1 KB
   per wave unused until round 4: LDS is what the tiers compete for) */
#define CW_POA_HOTC_OF_TIER(T) ((T) == 1 || CW_M2_CODES)
#ifndef CW_M2_DIRS
#define CW_M2_DIRS 0 /* 1: tier M2 writes tier L's direction words (in its slab, where tier M1 keeps code words) and walks them run by run instead of reading 8x8 tiles of
                        the matrix back: its traceback costs as many wave-cycles as its fill (15.0 against 16.5 G per depth-150 batch), tier L's a quarter */
#endif
#ifndef CW_M2_CHAIN_TABS
#define CW_M2_CHAIN_TABS 0 /* 1: tier M2 keeps the traceback's chain tables p2 / p4 in LDS (2 KB per wave: a tile's eight rows in three dependent reads instead of
                              seven).  Without them its work-group is 38.6 KB -- the size of M1's, L's and Q's: any four fit a CU -- and the two-engine step 1.8 ms
                              shorter (61.6 -> 59.8 ms, four alternating runs each on one box) */
#endif
#define CW_POA_HOT2T_BYTES(T, NC, EC, LC) (CW_POA_HOTC_OF_TIER(T) ? CW_POA_HOT2C_BYTES(NC, EC, LC) : CW_POA_HOT2_BYTES(NC, EC, LC) + (CW_M2_CHAIN_TABS ? (4 * (NC) + 15) / 16 * 16 : 0))
#ifndef CW_L_COLD_NODES
#define CW_L_COLD_NODES 0 /* 1: tier L keeps in_head / indeg / nbase / nalc / has_out in its slab, not in LDS (37 -> 26 KB per wave).  Measured: depth 150 unchanged within
                             noise, depth 30 -- where tier L is the long pole -- 1 ms slower with one engine (tier L 16.0 -> 17.1 ms).  Off. */
#endif
#define CW_POA_HOT2L_BYTES(NC, EC, LC) (CW_L_COLD_NODES ? ((NC) * 10 + (EC) * 2 + 3 * ((LC) + 1) + 64 + 15) / 16 * 16 : CW_POA_HOT2_BYTES(NC, EC, LC))
#define CW_POA_COLD2_BYTES(NC, EC, LC) (((NC) * 19 + (EC) * 4 + CW_POA_EW_BYTES(EC) + 4 * ((LC) + 1) + 255) / 256 * 256) /* (NC * 7 of it: tier L's per-node arrays, CW_L_COLD_NODES) */
#ifndef CW_S_EDGES_LDS
#define CW_S_EDGES_LDS 1 /* tier S keeps its in-edge lists and coverage counts in LDS (1.8 KB): the metadata pass walks them for every member */
#endif
#ifndef CW_S_LCODES_WORDS
#define CW_S_LCODES_WORDS 0 /* tier S: code words of an alignment in LDS when groups of eight rows x columns + 64 fit this many words (cw_poa_c.h), so that the
                               traceback's tile trips wait for LDS, not for the L2.  Measured with 384 words (93 % of tier S's members fit): traceback -11 %
                               wave-cycles, tier S's kernel 34.5 -> 33.2 ms -- and the step +1.4 ms, because the 1.5 KB per wave are LDS the other tiers'
                               work-groups no longer get.  Off. */
#endif
#define CW_POA_SLAB_BYTES (CW_POA_HOT2C_BYTES(CW_POA_NC, CW_POA_EC, CW_POA_LC) + (CW_S_EDGES_LDS ? 4 * CW_POA_EC + CW_POA_EW_BYTES(CW_POA_EC) + 2 * CW_POA_NC : 0) + 4 * CW_S_LCODES_WORDS) /* tier S, LDS per wave (round 4: laid out like M1 / M2) */
#define CW_POA_HSLAB_BYTES(NC, LC) ((((NC) + 1) * ((LC) + 1) * 2 + 255) / 256 * 256)
#define CW_POA_DSLAB_PAIRS(NC, LC) ((NC) * (((LC) + 64) / 64))
#define CW_POA_DSLAB_BYTES(NC, LC) ((CW_POA_DSLAB_PAIRS(NC, LC) * 16 + 255) / 256 * 256)
#define CW_POA_SLAB2_TOTAL(NC, EC, LC) (CW_POA_HSLAB_BYTES(NC, LC) + CW_POA_DSLAB_BYTES(NC, LC) + CW_POA_COLD2_BYTES(NC, EC, LC))

/* the row word (PoaMem::rmeta): base | linear << 2 | sink << 3 | slab << 4 | kind << 5 | n_pred << 8 (11 bits) | x << 19 (13 bits) */
#define CW_RM_NP(m) (int)(((m) >> 8) & 0x7FFu)
#define CW_RM_LIN(m) (((m) & 4u) != 0u)
#define CW_RM_X(m) (int)((m) >> 19)
#define CW_RM_MAX_PRED 2047u
static_assert(CW_POAB_EC <= 8192 && CW_POAL_EC <= 8192 && CW_POAM2_EC <= 8192 && CW_POAM1_EC <= 8192 && CW_POA_EC <= 8192, "the row word's x field (13 bits) holds a list offset < EC or a DP row <= NC");
static_assert(CW_POAB_NC <= 8191 && CW_POAL_NC <= 8191, "the row word's x field (13 bits) holds a DP row <= NC");
#define CW_RM_WORD(base, np, lin, sink, kind, x) ((uint32_t)(base) | ((lin) ? 4u : 0u) | ((sink) ? 8u : 0u) | ((uint32_t)(kind) << 5) | ((uint32_t)(np) << 8) | ((uint32_t)(x) << 19))
struct PoaComm; /* cw_poa_w.h: mailbox and boundary words of a multi-wave tier-L work-group */
template <typename HT>
struct PoaMem {
    HT* H;
    unsigned long long* dirs; /* traceback codes, 2 bits per cell as two ballots per (row, chunk): 0 diagonal and
                                 1 vertical through the first predecessor, 2 horizontal, 3 = compare cell values */
    uint32_t* rmeta;    /* rank -> the row word (CW_RM_*): base, n_pred (the virtual start counts as 1), sink = no out-edge, linear = the only predecessor is
                           the rank before, x = that predecessor's DP row when n_pred is 1, else the offset of the node's list in plist; kind and slab
                           bit: what cw_poa_c.h's fill does with the row.  The fills read this ONE word per row */
    uint16_t* rpred0;   /* rank -> DP row of its first predecessor (0 = virtual start)                           */
    uint16_t* p2;       /* rank -> DP row two / four steps up the first-predecessor chain, CW_NONE16 beyond the start */
    uint16_t* p4;       /* (only where the traceback walks matrix tiles: tier M1; else NULL)                      */
    uint16_t* plist;    /* predecessor DP rows in in-edge order (CSR); doubles as a u32 histogram during merges  */
    uint16_t* ncov;     /* node -> sequences through it                                                          */
    uint16_t* nal;      /* node -> 3 aligned node ids                                                            */
    uint16_t* in_head;  /* node -> first in-edge, CW_NONE16 if none                                              */
    uint16_t* in_tail;
    uint16_t* indeg;
    uint16_t* efrom;    /* edge -> source node                                                                   */
    uint16_t* enext;    /* edge -> next in-edge of the same target                                               */
    uint16_t* ew;       /* edge -> sequences whose path uses it (heaviest-bundle policy only, else NULL)         */
    uint16_t* r2n;      /* rank -> node                                                                          */
    uint16_t* n2r;
    uint16_t* rtmp;     /* new rank order while fresh nodes are being placed                                     */
    uint16_t* seqrank;  /* sequence position -> rank of the node it is aligned to, CW_NONE16 for an insertion    */
    uint16_t* pcur;     /* sequence position -> node it resolves to                                              */
    uint16_t* pat;      /* sequence position -> rank slot (old order) a fresh node goes in front of              */
    uint8_t* nbase;     /* node -> base code                                                                     */
    uint8_t* nalc;      /* node -> number of aligned nodes (0..3)                                                */
    uint8_t* has_out;
    uint8_t* sq;        /* current member, base codes                                                            */
    uint32_t n_cap, e_cap, l_cap, h_cap, d_cap;
    bool runs;          /* consume runs of equal moves per round trip (pays when paths have long straight stretches) */
    uint32_t* codes;    /* cw_poa_c.h: the traceback's decisions, four bits per cell (LDS in tier S, the wave's slab in tiers M1 / M2), c_cap words */
    int16_t* ring;      /* cw_poa_c.h: the last CW_RING rows of the fill (LDS; tiers M1 / M2) */
    uint32_t* gflag;    /* cw_poa_c.h: one bit per rank: the row is also written to the slab (a later row needs it from more than CW_RING ranks back,
                           or it belongs to a node with more than three in-edges, or it is the fourth or a later predecessor of such a node); NULL where every row is kept anyway */
    uint32_t c_cap;
    uint32_t* lcodes;   /* cw_poa_c.h, tier S: room in LDS for the code words of a small alignment (lc_cap words), or NULL */
    uint32_t lc_cap;
#ifdef CW_DIAG
    unsigned long long* diag; /* diagnostic build: ten counters of this tier in BatchCounters::prof (rows / linear rows / far loads / predecessor trips of the
                                 unpacked fill, rows / linear / far loads of the packed fill, traceback trips, slow steps, members) */
#endif
    PoaComm* comm;      /* cw_poa_w.h: non-NULL in a tier-L work-group of several waves (this is wave 0, the others serve its FILL commands) */
    bool pad64;         /* the matrix is in a slab with 64 cells of slack behind it: rows of <= 64 columns are stored and loaded by all 64 lanes
                           (no execution mask round the store of a row, no mask round a far row's load; see poa_fill) */
};

template <typename HT>
__device__ __forceinline__ PoaMem<HT> poa_carve(uint8_t* base, uint32_t nc, uint32_t ec, uint32_t lc, uint32_t hc, uint32_t dc,
                                                HT* h_ext = nullptr, unsigned long long* d_ext = nullptr, uint8_t* cold = nullptr,
                                                bool cold_edges = false, bool chain_tabs = false, bool cold_nodes = false) {
    PoaMem<HT> M;
    uint8_t* p = base;
    uint8_t* pc = cold; /* merge-only arrays: in the slab when given, else with the rest */
    if (h_ext) M.H = h_ext;
    else { M.H = (HT*)p; p += (size_t)hc * sizeof(HT); }
    if (d_ext) M.dirs = d_ext;
    else { M.dirs = (unsigned long long*)p; p += (size_t)dc * 16; }
    M.rmeta = (uint32_t*)p; p += 4 * nc;
    M.plist = (uint16_t*)p; p += 2 * ec;   /* 4-byte aligned: follows rmeta */
    M.ew = nullptr;
    if (!(pc && cold_edges)) { M.efrom = (uint16_t*)p; p += 2 * ec; M.enext = (uint16_t*)p; p += 2 * ec; if (CW_CONS_HEAVIEST_BUNDLE) { M.ew = (uint16_t*)p; p += 2 * ec; } }
    M.rpred0 = (uint16_t*)p; p += 2 * nc;
    if (chain_tabs) { M.p2 = (uint16_t*)p; p += 2 * nc; M.p4 = (uint16_t*)p; p += 2 * nc; } else { M.p2 = nullptr; M.p4 = nullptr; }
    if (!(pc && cold_edges)) { M.ncov = (uint16_t*)p; p += 2 * nc; }
    if (pc) { M.nal = (uint16_t*)pc; pc += 6 * nc; } else { M.nal = (uint16_t*)p; p += 6 * nc; }
    if (pc && cold_nodes) { M.in_head = (uint16_t*)pc; pc += 2 * nc; } else { M.in_head = (uint16_t*)p; p += 2 * nc; }
    if (pc) { M.in_tail = (uint16_t*)pc; pc += 2 * nc; } else { M.in_tail = (uint16_t*)p; p += 2 * nc; }
    if (pc && cold_nodes) { M.indeg = (uint16_t*)pc; pc += 2 * nc; } else { M.indeg = (uint16_t*)p; p += 2 * nc; }
    M.r2n = (uint16_t*)p; p += 2 * nc;
    M.n2r = (uint16_t*)p; p += 2 * nc;
    if (pc) { M.rtmp = (uint16_t*)pc; pc += 2 * nc; } else { M.rtmp = (uint16_t*)p; p += 2 * nc; }
    M.seqrank = (uint16_t*)p; p += 2 * (lc + 1);
    if (pc) { M.pcur = (uint16_t*)pc; pc += 2 * (lc + 1); M.pat = (uint16_t*)pc; pc += 2 * (lc + 1); }
    else { M.pcur = (uint16_t*)p; p += 2 * (lc + 1); M.pat = (uint16_t*)p; p += 2 * (lc + 1); }
    if (pc && cold_edges) { M.efrom = (uint16_t*)pc; pc += 2 * ec; M.enext = (uint16_t*)pc; pc += 2 * ec; if (CW_CONS_HEAVIEST_BUNDLE) { M.ew = (uint16_t*)pc; pc += 2 * ec; } M.ncov = (uint16_t*)pc; pc += 2 * nc; }
    if (pc && cold_nodes) { M.nbase = pc; pc += nc; M.nalc = pc; pc += nc; M.has_out = pc; pc += nc; } /* tier L: per-node arrays the metadata pass, the end cell and the merge read once per
                                                                                                           member -- next to a millisecond of fill -- leave LDS (10.5 of its 37 KB per wave) */
    else { M.nbase = p; p += nc; M.nalc = p; p += nc; M.has_out = p; p += nc; }
    M.sq = p; p += lc + 1;
    M.n_cap = nc; M.e_cap = ec; M.l_cap = lc; M.h_cap = hc; M.d_cap = dc; M.runs = false; M.pad64 = false;
    M.codes = nullptr; M.ring = nullptr; M.gflag = nullptr; M.c_cap = 0; M.lcodes = nullptr; M.lc_cap = 0; M.comm = nullptr;
#ifdef CW_DIAG
    M.diag = nullptr;
#endif
    return M;
}

/*
 * DP fill with the previous row resident in registers: NCH chunks of 64 columns per row (cols <= 64*NCH).
 * A row whose predecessor is the row just computed needs no memory read at all; other predecessor rows are
 * fetched for all chunks at once (one round trip per row).  The next row's metadata is requested while the
 * current row computes.
 */
template <typename HT, int NCH, bool DIRS, bool PAD = false>
__device__ __forceinline__ void poa_fill(const PoaMem<HT>& M, const int n, const int cols, const int lane, const bool use_dirs_) {
    const int hs = cols; /* row stride of the matrix.  PAD (slab tiers, NCH == 1): every lane stores and loads without an execution mask -- a lane beyond
                            the member's columns then writes into the first cells of the NEXT row(s), which are computed and stored later (rows are
                            written in rank order, read only when complete), and reads cells nobody uses; the slab has 64 cells of slack */
    const bool use_dirs = DIRS && use_dirs_; /* direction words exist in tier L only: everywhere else the branch is compiled out of the row loop */
    const int nch = (cols + 63) >> 6;
    const int G = CW_POA_GAP, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH;
    /* the last RC rows stay in registers (rc_[0] = the previous row): a predecessor a few ranks back -- the other arm of a bubble --
       costs no memory round trip */
    constexpr int RC = NCH <= 4 ? 2 : 1;
    int sq_[NCH], rc_[RC][NCH];
    bool act[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int j = c * 64 + lane;
        act[c] = j < cols;
        sq_[c] = (j > 0 && act[c]) ? (int)M.sq[j - 1] : -1;
#pragma unroll
        for (int k = 0; k < RC; ++k) rc_[k][c] = CW_POA_SW ? 0 : j * G; /* row 0 */
    }
    uint32_t meta_n = M.rmeta[0];
#ifdef CW_DIAG
    uint32_t dg_lin = 0, dg_far = 0, dg_pred = 0, dg_far16 = 0;
#endif
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta_n);
        if (r + 1 < n) meta_n = M.rmeta[r + 1];
        const int base = (int)(meta & 3u), np = CW_RM_NP(meta), off = CW_RM_X(meta), pr0 = off; /* pr0 is meaningful when np == 1 */
#ifdef CW_DIAG
        if (CW_RM_LIN(meta)) dg_lin++; else { dg_pred += (uint32_t)np; for (int q = 0; q < np; ++q) { const int prow = (np == 1) ? pr0 : (int)M.plist[off + q]; if (i - prow > RC) dg_far++; if (i - prow > 16) dg_far16++; } }
#endif
        int v[NCH], dgv[NCH], upv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { v[c] = CW_NEG; dgv[c] = CW_NEG; upv[c] = CW_NEG; }
        if (CW_RM_LIN(meta)) { /* see poa_fill_pk */
            int carry_in = CW_NEG;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int upc = rc_[0][c];
                const int dg = cw_wave_shr1(upc, carry_in);
                if (c + 1 < NCH) carry_in = cw_lane_value(upc, 63);
                const int s_ = (sq_[c] == base) ? MS : XS;
                dgv[c] = dg + s_; upv[c] = upc + G;
                v[c] = max(dgv[c], upv[c]);
            }
        } else
        for (int q = 0; q < np; ++q) {
            const int prow = (np == 1) ? pr0 : __builtin_amdgcn_readfirstlane((int)M.plist[off + q]);
            const int dist = i - prow;
            int up[NCH];
            if (dist <= RC) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    up[c] = rc_[0][c];
#pragma unroll
                    for (int k = 1; k < RC; ++k) up[c] = (dist == k + 1) ? rc_[k][c] : up[c];
                }
            } else {
                /* no fence: a lane reads back only the columns it wrote itself (program order of one work-item), and a fence here
                   would wait for the store of the row just finished before the load could even be issued */
                const int pr = prow * hs;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int j = c * 64 + lane;
                    up[c] = (PAD || act[c]) ? (int)M.H[pr + j] : CW_NEG;
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(up[c])); /* see poa_fill_pk: keeps the wait for this load out of the common path */
            }
            int carry_in = CW_NEG;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int dg = cw_wave_shr1(up[c], carry_in); /* the cell up-left: lane l-1 of the same row, column 0 has none */
                carry_in = cw_lane_value(up[c], 63);
                const int s = (sq_[c] == base) ? MS : XS;
                dgv[c] = dg + s; upv[c] = up[c] + G;
                v[c] = max(v[c], max(dgv[c], upv[c]));
            }
        }
        if (CW_POA_OV) v[0] = lane == 0 ? 0 : v[0]; /* overlap mode: the graph's prefix is free */
        if (CW_POA_SW) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) v[c] = max(v[c], 0); /* local mode: no cell below 0 (clamping the candidates is clamping the cells: a gap from a clamped cell is negative) */
        }
        int carry = CW_NEG;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int j = c * 64 + lane;
            /* (no mask for the lanes beyond the member's columns: a prefix max runs left to right, and everything to the right of the last
               column -- in this chunk and in the chunks after it -- is beyond the columns too: what they hold reaches no cell that is read) */
            int w = v[c] - j * G;
            w = cw_wave_scan_max(w);
            w = max(w, carry);
            carry = cw_lane_value(w, 63);
            const int nv = w + j * G;
#pragma unroll
            for (int k = RC - 1; k > 0; --k) rc_[k][c] = rc_[k - 1][c];
            rc_[0][c] = nv;
            if (PAD || act[c]) M.H[i * hs + j] = (HT)nv;
            if (use_dirs && c < nch) {
                /* single-predecessor row: the code the traceback would derive (diagonal, then vertical, then horizontal) */
                unsigned long long b0, b1;
                if (np == 1) {
                    const unsigned long long bd = __ballot(j > 0 && nv == dgv[c]), bu = __ballot(nv == upv[c]);
                    b0 = ~bd & bu; b1 = ~bd & ~bu;
                } else {
                    b0 = b1 = ~0ull;
                }
                if (lane == 0) { M.dirs[(r * nch + c) * 2] = b0; M.dirs[(r * nch + c) * 2 + 1] = b1; }
            }
        }
    }
#ifdef CW_DIAG
    if (lane == 0 && M.diag) { atomicAdd(&M.diag[0], (unsigned long long)n); atomicAdd(&M.diag[1], (unsigned long long)dg_lin); atomicAdd(&M.diag[2], (unsigned long long)dg_far); atomicAdd(&M.diag[3], (unsigned long long)dg_pred); atomicAdd(&M.diag[10], (unsigned long long)dg_far16); }
#endif
    cw_wave_sync();
}

/* ---- packed int16 arithmetic: two DP columns per lane (VOP3P v_pk_add/sub/max_i16) ------------------------------ */
typedef short cw_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_add(int a, int b) { return __builtin_bit_cast(int, (cw_s2)(__builtin_bit_cast(cw_s2, a) + __builtin_bit_cast(cw_s2, b))); }
__device__ __forceinline__ int pk_sub(int a, int b) { return __builtin_bit_cast(int, (cw_s2)(__builtin_bit_cast(cw_s2, a) - __builtin_bit_cast(cw_s2, b))); }
__device__ __forceinline__ int pk_max(int a, int b) { return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(cw_s2, a), __builtin_bit_cast(cw_s2, b))); }
__device__ __forceinline__ int pk_make(int lo, int hi) { return (lo & 0xFFFF) | (hi << 16); }
__device__ __forceinline__ int pk_splat_lo(int a) { return __builtin_amdgcn_perm(a, a, 0x05040504); } /* (lo, lo) */
/* emask holds, per half, one bit per base the column's sequence position could match (bit b of the half: its base is b); the row's
   substitution scores are then a shift by the node's base, a mask and one packed multiply-add: MATCH where the bit is set, MISMATCH elsewhere */
__device__ __forceinline__ int pk_score(int emask, int base) {
    const int t = (int)(((unsigned)emask >> base) & 0x00010001u);
    const cw_s2 r = __builtin_bit_cast(cw_s2, t) * (cw_s2)(CW_POA_MATCH - CW_POA_MISMATCH) + (cw_s2)(CW_POA_MISMATCH);
    return __builtin_bit_cast(int, r);
}
#define CW_NEG16 (-30000)
#define CW_NEGPK ((int)0x8AD08AD0) /* (CW_NEG16, CW_NEG16) */

/* direction-word addressing: unpacked rows hold 2 words per 64 columns; packed rows 4 words per 128 columns
   (even columns, odd columns) x (low bit, high bit), bit = lane that owns the column pair */
__device__ __forceinline__ int poa_dir_code(const unsigned long long* dirs, int r, int c, int nch, bool packed) {
    if (!packed) {
        const int w = (r * nch + (c >> 6)) * 2;
        return (int)((dirs[w] >> (c & 63)) & 1ull) | ((int)((dirs[w + 1] >> (c & 63)) & 1ull) << 1);
    }
    const int w = (r * nch + (c >> 7)) * 4 + (c & 1) * 2, bit = (c & 127) >> 1;
    return (int)((dirs[w] >> bit) & 1ull) | ((int)((dirs[w + 1] >> bit) & 1ull) << 1);
}

/*
 * Packed DP fill for rows of more than 64 columns in the int16 tiers: NCH2 chunks of 128 columns, lane l of chunk c owns
 * columns 128c+2l (low half) and 128c+2l+1 (high half).  The diagonal operand is one v_alignbit of the row above with
 * its wave_shr:1 copy; the horizontal recurrence is an in-lane step plus a packed DPP prefix max.  Row stride hs is even
 * so that a lane's pair is one aligned 32-bit access.  |values| stay far from the int16 range (CW_NEG16 headroom).
 */
template <int NCH2, bool DIRS>
__device__ __forceinline__ void poa_fill_pk(const PoaMem<int16_t>& M, const int n, const int cols, const int hs, const int lane,
                                            const bool use_dirs_) {
    const bool use_dirs = DIRS && use_dirs_;
    const int G = CW_POA_GAP;
    const int GPK = pk_make(G, G);
    const int nch = (cols + 127) >> 7;
    constexpr int RC = NCH2 <= 2 ? 3 : NCH2 <= 4 ? 2 : 1; /* rows kept in registers, rc_[0] = the previous one (see poa_fill) */
    int rc_[RC][NCH2], jg[NCH2], amask[NCH2], qpk[NCH2];
    int* Hw = (int*)M.H;
#pragma unroll
    for (int c = 0; c < NCH2; ++c) {
        const int j0 = c * 128 + 2 * lane, j1 = j0 + 1;
        jg[c] = pk_make(j0 * G, j1 * G);
#pragma unroll
        for (int k = 0; k < RC; ++k) rc_[k][c] = CW_POA_SW ? 0 : jg[c]; /* row 0 */
        amask[c] = (j0 < cols ? 0xFFFF : 0) | (j1 < cols ? (int)0xFFFF0000 : 0);
        const int q0 = (j0 >= 1 && j0 < cols) ? (int)M.sq[j0 - 1] : -1, q1 = (j1 < cols) ? (int)M.sq[j1 - 1] : -1;
        qpk[c] = (q0 >= 0 ? 1 << q0 : 0) | (q1 >= 0 ? 1 << (16 + q1) : 0); /* see pk_score */
    }
    uint32_t meta_n = M.rmeta[0];
#ifdef CW_DIAG
    uint32_t dg_lin = 0, dg_far = 0, dg_far16 = 0;
#endif
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta_n);
        if (r + 1 < n) meta_n = M.rmeta[r + 1];
        const int base = (int)(meta & 3u), np = CW_RM_NP(meta), off = CW_RM_X(meta), pr0 = off; /* pr0 is meaningful when np == 1 */
#ifdef CW_DIAG
        if (CW_RM_LIN(meta)) dg_lin++; else for (int q = 0; q < np; ++q) { const int prow = (np == 1) ? pr0 : (int)M.plist[off + q]; if (i - prow > RC) dg_far++; if (i - prow > 16) dg_far16++; }
#endif
        int v[NCH2], dgv[NCH2], upv[NCH2], srow[NCH2];
#pragma unroll
        for (int c = 0; c < NCH2; ++c) {
            v[c] = CW_NEGPK; dgv[c] = CW_NEGPK; upv[c] = CW_NEGPK;
            srow[c] = pk_score(qpk[c], base); /* the row's substitution scores, branch-free */
        }
        /* the usual row -- one predecessor, the row just computed (a linear stretch of the graph): straight-line code, no predecessor
           loop, no choice of the source row.  Read in the ISA: the general loop below costs ~12 taken branches and ~130 instructions per
           row (814 cycles per row measured in tier L); this path is the ~45 vector instructions the recurrence needs */
        if (CW_RM_LIN(meta)) {
            int carry_in = CW_NEGPK;
#pragma unroll
            for (int c = 0; c < NCH2; ++c) {
                const int upc = rc_[0][c];
                const int sh = CW_DPP(carry_in, upc, 0x138, 0xF);
                if (c + 1 < NCH2) carry_in = cw_lane_value(upc, 63);
                const int dg = __builtin_amdgcn_alignbit(upc, sh, 16);
                dgv[c] = pk_add(dg, srow[c]); upv[c] = pk_add(upc, GPK);
                v[c] = pk_max(dgv[c], upv[c]);
            }
        } else
        for (int q = 0; q < np; ++q) {
            const int prow = (np == 1) ? pr0 : __builtin_amdgcn_readfirstlane((int)M.plist[off + q]);
            int up[NCH2];
            const int dist = i - prow;
            if (dist <= RC) {
#pragma unroll
                for (int c = 0; c < NCH2; ++c) {
                    up[c] = rc_[0][c];
#pragma unroll
                    for (int k = 1; k < RC; ++k) up[c] = (dist == k + 1) ? rc_[k][c] : up[c];
                }
            } else {
                /* no fence: a lane reads back only the columns it wrote itself (program order of one work-item), and a fence here
                   would wait for the store of the row just finished before the load could even be issued */
#pragma unroll
                for (int c = 0; c < NCH2; ++c) {
                    const int j0 = c * 128 + 2 * lane;
                    up[c] = (j0 < cols) ? Hw[(prow * hs + j0) >> 1] : CW_NEGPK;
                }
                /* the loaded row is "used" here, inside the branch: otherwise the compiler waits for it (s_waitcnt vmcnt(0)) where the two
                   branches meet, i.e. in EVERY row -- and vmcnt also counts the stores of the row before, so every row waited for the
                   previous row to reach the L2 (read in the ISA: 834 cycles per row in tier L where the arithmetic needs ~200) */
#pragma unroll
                for (int c = 0; c < NCH2; ++c) asm volatile("" : "+v"(up[c]));
            }
            int carry_in = CW_NEGPK;
#pragma unroll
            for (int c = 0; c < NCH2; ++c) {
                const int sh = CW_DPP(carry_in, up[c], 0x138, 0xF);       /* lane l-1's pair; lane 0: last pair of the chunk before */
                carry_in = cw_lane_value(up[c], 63);
                const int dg = __builtin_amdgcn_alignbit(up[c], sh, 16);   /* (col 2l-1, col 2l) of the row above */
                dgv[c] = pk_add(dg, srow[c]); upv[c] = pk_add(up[c], GPK);
                v[c] = pk_max(v[c], pk_max(dgv[c], upv[c]));
            }
        }
        if (CW_POA_OV) v[0] = lane == 0 ? (int)((unsigned)v[0] & 0xFFFF0000u) : v[0]; /* overlap mode: column 0 (lane 0's even half) is free */
        if (CW_POA_SW) {
#pragma unroll
            for (int c = 0; c < NCH2; ++c) v[c] = pk_max(v[c], 0); /* local mode: no cell below 0 */
        }
        unsigned carry = 0u; /* biased (value + 32768): 0 is "nothing yet" */
#pragma unroll
        for (int c = 0; c < NCH2; ++c) {
            int w = pk_sub(v[c], jg[c]);
            /* (no mask for halves beyond the member's columns, see poa_fill: they only ever reach cells further right) */
            w = pk_max(w, (w << 16) | 0x8AD0);                                   /* odd column sees the even one of its lane */
            /* the lane's running max (its high half) as a biased unsigned number: the prefix max over the lanes is then six fused
               v_max_u32_dpp (VOP3P has no DPP form, and 0 is both the fill value of a shift and the identity of the max) */
            const unsigned inc = cw_wave_scan_max_u32(((unsigned)w >> 16) ^ 0x8000u);
            unsigned ex = (unsigned)CW_DPP(0, (int)inc, 0x138, 0xF);              /* exclusive: lanes before this one ... */
            if (c > 0) ex = max(ex, carry);                                        /* ... and the chunks before */
            w = pk_max(w, pk_splat_lo((int)(ex ^ 0x8000u)));
            if (c + 1 < NCH2) carry = max(carry, (unsigned)cw_lane_value((int)inc, 63));
            const int nv = pk_add(w, jg[c]);
#pragma unroll
            for (int k = RC - 1; k > 0; --k) rc_[k][c] = rc_[k - 1][c];
            rc_[0][c] = nv;
            const int j0 = c * 128 + 2 * lane;
            if (j0 < cols) Hw[(i * hs + j0) >> 1] = nv;
            if (use_dirs && c < nch) {
                /* code 0 diagonal, 1 vertical, 2 horizontal (single predecessor), 3 = compare cell values: straight from four
                   half-word compares to the four ballots (bit 0 = vertical, bit 1 = horizontal; both = 3) */
                unsigned long long e0, e1, o0, o1;
                if (np == 1) {
                    const bool de = j0 > 0 && (short)nv == (short)dgv[c], ue = (short)nv == (short)upv[c];
                    const bool dd = (short)((unsigned)nv >> 16) == (short)((unsigned)dgv[c] >> 16), uo = (short)((unsigned)nv >> 16) == (short)((unsigned)upv[c] >> 16);
                    const unsigned long long bde = __ballot(de), bue = __ballot(ue), bdo = __ballot(dd), buo = __ballot(uo);
                    e0 = ~bde & bue; e1 = ~bde & ~bue; o0 = ~bdo & buo; o1 = ~bdo & ~buo;
                } else {
                    e0 = e1 = o0 = o1 = ~0ull;
                }
                if (lane == 0) {
                    unsigned long long* d = M.dirs + (size_t)(r * nch + c) * 4;
                    d[0] = e0; d[1] = e1; d[2] = o0; d[3] = o1;
                }
            }
        }
    }
#ifdef CW_DIAG
    if (lane == 0 && M.diag) { atomicAdd(&M.diag[4], (unsigned long long)n); atomicAdd(&M.diag[5], (unsigned long long)dg_lin); atomicAdd(&M.diag[6], (unsigned long long)dg_far); atomicAdd(&M.diag[11], (unsigned long long)dg_far16); }
#endif
    cw_wave_sync();
}

/* ---- round 5: FOUR MEMBERS PER FILL in the tall tiers ---------------------------------------------------------------------------------------
 * The tasks of tiers M2 and L are the ragged first and last segments of a window: one long member makes a graph of hundreds of nodes, then dozens
 * of short pieces are aligned against all of it (85-94 % of these tiers' members have at most 31 bases; 79-85 % of those change nothing in the
 * graph but coverage counts).  A fill of such a member is tall and a quarter of a wave wide, and its rows depend on one another -- but the NEXT
 * members' fills are the same rows of the same graph as long as the merge in between adds no node and no edge.  So up to four consecutive members
 * of at most 31 bases are filled TOGETHER, one per 16-lane DPP row (two packed columns per lane), under the one scalar row loop of the graph; each
 * is then traced back and merged in turn, and the first merge that changes the graph's structure voids the fills behind it (those members are
 * filled again, with their successors).  Same cells, same direction words (layout of a packed 128-column row, member g in columns 32g .. 32g + 31),
 * same results; half the row passes of these tiers at depth 150. */
#ifndef CW_POA_GROUP_FILL
#define CW_POA_GROUP_FILL 1
#endif
#define CW_GF_LC 31  /* longest member of a group fill */
#define CW_GF_HS 128 /* row stride of its matrix: four members x 32 columns */
template <bool DIRS>
__device__ __forceinline__ void poa_fill_pk4(const PoaMem<int16_t>& M, const int n, const int lane, const bool use_dirs_, const uint32_t lens4) {
    const bool use_dirs = DIRS && use_dirs_;
    const int G = CW_POA_GAP;
    const int GPK = pk_make(G, G);
    const int gl = lane & 15, mg = lane >> 4;
    const int j0 = 2 * gl, j1 = j0 + 1;
    const int jg = pk_make(j0 * G, j1 * G);
    int rc0 = CW_POA_SW ? 0 : jg, rc1 = rc0, rc2 = rc0; /* rows i-1, i-2, i-3 (row 0) */
    int* Hw = (int*)M.H;
    /* the members' bases: member g at M.sq[32 g ..]; a position beyond its member matches nothing (0xFF) */
    const int q0 = j0 >= 1 ? (int)M.sq[32 * mg + j0 - 1] : 255, q1 = (int)M.sq[32 * mg + j1 - 1];
    const int qpk = (q0 < 4 ? 1 << q0 : 0) | (q1 < 4 ? 1 << (16 + q1) : 0);
    /* only the columns a member has are stored (lens4: the members' lengths, a byte each, 0 = no member in that row of lanes): what lies to the right
       of them is computed, read by nobody, and would triple the bytes these tiers write */
    const bool st = j0 <= (int)((lens4 >> (8 * mg)) & 255u) && ((lens4 >> (8 * mg)) & 255u) != 0u;
    if (st) Hw[lane] = rc0; /* row 0 */
    uint32_t meta_n = M.rmeta[0];
    for (int r = 0; r < n; ++r) {
        const int i = r + 1;
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta_n);
        if (r + 1 < n) meta_n = M.rmeta[r + 1];
        const int base = (int)(meta & 3u), np = CW_RM_NP(meta), off = CW_RM_X(meta), pr0 = off;
        const int srow = pk_score(qpk, base);
        int v, dgv = CW_NEGPK, upv = CW_NEGPK;
        if (CW_RM_LIN(meta)) {
            const int sh = CW_DPP(CW_NEGPK, rc0, 0x111, 0xF); /* row_shr:1: the neighbour inside the member's 16 lanes, none for its column 0 */
            dgv = pk_add(__builtin_amdgcn_alignbit(rc0, sh, 16), srow); upv = pk_add(rc0, GPK);
            v = pk_max(dgv, upv);
        } else {
            v = CW_NEGPK;
            for (int q = 0; q < np; ++q) {
                const int prow = (np == 1) ? pr0 : __builtin_amdgcn_readfirstlane((int)M.plist[off + q]);
                const int dist = i - prow;
                int up;
                if (dist <= 3) up = dist == 1 ? rc0 : dist == 2 ? rc1 : rc2;
                else { up = st ? Hw[prow * (CW_GF_HS / 2) + lane] : CW_NEGPK; asm volatile("" : "+v"(up)); } /* (see poa_fill_pk: the wait stays inside the branch) */
                const int sh = CW_DPP(CW_NEGPK, up, 0x111, 0xF);
                dgv = pk_add(__builtin_amdgcn_alignbit(up, sh, 16), srow); upv = pk_add(up, GPK);
                v = pk_max(v, pk_max(dgv, upv));
            }
        }
        if (CW_POA_OV) v = gl == 0 ? (int)((unsigned)v & 0xFFFF0000u) : v; /* overlap mode: every member's column 0 is free */
        if (CW_POA_SW) v = pk_max(v, 0);
        int w = pk_sub(v, jg);
        w = pk_max(w, (w << 16) | 0x8AD0); /* the odd column sees the even one of its lane */
        unsigned inc = ((unsigned)w >> 16) ^ 0x8000u; /* the prefix max stays inside the member's 16 lanes: four row shifts */
        inc = max(inc, (unsigned)CW_DPP(0, (int)inc, 0x111, 0xF)); inc = max(inc, (unsigned)CW_DPP(0, (int)inc, 0x112, 0xF));
        inc = max(inc, (unsigned)CW_DPP(0, (int)inc, 0x114, 0xF)); inc = max(inc, (unsigned)CW_DPP(0, (int)inc, 0x118, 0xF));
        const unsigned ex = (unsigned)CW_DPP(0, (int)inc, 0x111, 0xF);
        w = pk_max(w, pk_splat_lo((int)(ex ^ 0x8000u)));
        const int nv = pk_add(w, jg);
        rc2 = rc1; rc1 = rc0; rc0 = nv;
        if (st) Hw[i * (CW_GF_HS / 2) + lane] = nv;
        if (use_dirs) { /* as poa_fill_pk, one 128-column chunk per row */
            unsigned long long e0, e1, o0, o1;
            if (np == 1) {
                const bool de = j0 > 0 && (short)nv == (short)dgv, ue = (short)nv == (short)upv;
                const bool dd = (short)((unsigned)nv >> 16) == (short)((unsigned)dgv >> 16), uo = (short)((unsigned)nv >> 16) == (short)((unsigned)upv >> 16);
                const unsigned long long bde = __ballot(de), bue = __ballot(ue), bdo = __ballot(dd), buo = __ballot(uo);
                e0 = ~bde & bue; e1 = ~bde & ~bue; o0 = ~bdo & buo; o1 = ~bdo & ~buo;
            } else {
                e0 = e1 = o0 = o1 = ~0ull;
            }
            if (lane == 0) { unsigned long long* d = M.dirs + (size_t)r * 4; d[0] = e0; d[1] = e1; d[2] = o0; d[3] = o1; }
        }
    }
    cw_wave_sync();
}

#include "cw_poa_c.h"
#ifndef CW_POA_REPLAY
#define CW_POA_REPLAY 1 /* 0: every member is aligned, as through round 5 (tests/test_gpu_variants.py builds both replay switches off) */
#endif
#ifndef CW_POA_LW
#define CW_POA_LW 0 /* round 6, tier "LW" (-DCW_POA_LW=1; not with the local alignment mode): tier-L tasks whose members are wide on average run on the four waves
                       of a work-group (cw_poa_w.h; a second instance of the tier-L kernel on its own list and stream).  Bit-identical (tests/test_gpu_variants.py)
                       and OFF: measured on one box at depth 150 (three runs each, alternating), a batch alone on the GPU takes 76.3 -> 70.6 ms with it (the
                       batch's eight such tasks: the longest 26 -> 19 ms; tier L's kernel 60 -> 40 ms), and the step with three batches in flight 52.5 -> 53.1 ms
                       (four engines: 50.8 -> 52.1): the fifth stream and the four-wave work-groups cost the other tiers more than the stragglers' tail,
                       which other batches' work fills anyway, was costing.  The native driver's runs did not move (0.92-1.24 s either way). */
#endif
#if CW_POA_LW && CW_POA_SW
#error "cw_poa.h: tier LW (cw_poa_w.h) has no local alignment mode"
#endif
#define CW_POAL_MW 4 /* waves of such a work-group */
#if CW_POA_LW
#include "cw_poa_w.h"
#else
#define CW_POA_COMM_BYTES 0
#endif

/* One traceback step at a node with several predecessors (or whose step the direction words left open), decided from the cell
 * values in the order of preference of cw_policy.h: diagonal through the in-edges in order, then vertical through them, then
 * horizontal.  Lanes = in-edges: every candidate cell is requested at once, one memory round trip for the whole step.
 * Returns false when no move explains the cell (capacity/overflow paths report it). */
template <typename HT>
__device__ __forceinline__ bool poa_slow_step(const PoaMem<HT>& M, const int i, const int j, const int hs, const int pr0, const int lane,
                                              int* pi_out, int* pj_out) {
    const int G = CW_POA_GAP, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH;
    const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)M.rmeta[i - 1]);
    const int base = (int)(meta & 3u), np = CW_RM_NP(meta), off = CW_RM_X(meta); /* off: only read when np > 1 */
    int pi = i, pj = j;
    bool found = false;
    if (np <= 64) {
        const bool mine = lane < np;
        const int pr = mine ? ((np == 1) ? pr0 : (int)M.plist[off + lane]) : 0;
        int hd = CW_NEG * 2, hv = CW_NEG * 2;
        if (mine) { hv = (int)M.H[pr * hs + j]; if (j != 0) hd = (int)M.H[pr * hs + j - 1]; }
        const int h = (int)M.H[i * hs + j];
        const int hh = j != 0 ? (int)M.H[i * hs + j - 1] : CW_NEG * 2;
        const int sx = (j != 0 && (int)M.sq[j - 1] == base) ? MS : XS;
        const unsigned long long bd = __ballot(mine && j != 0 && h == hd + sx);
        const unsigned long long bv = __ballot(mine && h == hv + G);
        if (bd) { pi = __builtin_amdgcn_readlane(pr, __builtin_amdgcn_readfirstlane(__ffsll((long long)bd) - 1)); pj = j - 1; found = true; }
        else if (bv) { pi = __builtin_amdgcn_readlane(pr, __builtin_amdgcn_readfirstlane(__ffsll((long long)bv) - 1)); pj = j; found = true; }
        else if (__builtin_amdgcn_readfirstlane((j != 0 && h == hh + G) ? 1 : 0)) { pi = i; pj = j - 1; found = true; }
    } else { /* more in-edges than lanes: one at a time */
        const int h = M.H[i * hs + j];
        if (j != 0) {
            const int sx = ((int)M.sq[j - 1] == base) ? MS : XS;
            for (int q = 0; q < np && !found; ++q) {
                const int pr = (int)M.plist[off + q];
                if (h == (int)M.H[pr * hs + j - 1] + sx) { pi = pr; pj = j - 1; found = true; }
            }
        }
        for (int q = 0; q < np && !found; ++q) {
            const int pr = (int)M.plist[off + q];
            if (h == (int)M.H[pr * hs + j] + G) { pi = pr; pj = j; found = true; }
        }
        if (!found && j != 0 && h == (int)M.H[i * hs + j - 1] + G) { pi = i; pj = j - 1; found = true; }
    }
    *pi_out = pi; *pj_out = pj;
    return found;
}


#if CW_CONS_HEAVIEST_BUNDLE
/* cw_policy.h CW_POA_CONSENSUS_HEAVIEST_BUNDLE, on ONE lane: a recurrence over the nodes in rank order, once per task (scores by node in
   rmeta, the chosen source in rpred0 -- both free after the last member).  Returns the consensus length; writes it if it fits. */
template <typename HT>
__device__ __forceinline__ uint32_t poa_consensus_hb(const PoaMem<HT>& M, const int n, const PoaTask& t, const DevScratch& sc) {
    uint32_t* score = M.rmeta;
    uint16_t* pred = M.rpred0;
    int end = -1;
    uint32_t end_score = 0;
    for (int r = 0; r < n; ++r) {
        const int v = M.r2n[r];
        int p = -1;
        uint32_t wb = 0, ps = 0;
        for (uint32_t e = M.in_head[v]; e != CW_NONE16; e = M.enext[e]) {
            const int u = M.efrom[e];
            const uint32_t w = M.ew[e], su = score[u];
            if (p < 0 || wb < w || (wb == w && ps <= su)) { wb = w; p = u; ps = su; }
        }
        const uint32_t s = p < 0 ? 0u : wb + ps;
        score[v] = s;
        pred[v] = p < 0 ? (uint16_t)CW_NONE16 : (uint16_t)p;
        if (!M.has_out[v] && (end < 0 || end_score < s)) { end = v; end_score = s; }
    }
    uint32_t len = 0;
    for (int v = end; v >= 0; v = pred[v] == CW_NONE16 ? -1 : (int)pred[v]) ++len;
    if (len <= t.out_cap) {
        uint32_t k = len;
        for (int v = end; v >= 0; v = pred[v] == CW_NONE16 ? -1 : (int)pred[v]) sc.arena[t.out_off + --k] = CW_ACGT(M.nbase[v]);
    }
    return len;
}
#endif

/* Returns 1 = done, 2 = a capacity of this memory class was exceeded, 3 = output capacity exceeded / internal. */
/* PK: 0 = one column per lane (int32 tier G); 1 = two columns per lane in packed int16 for rows wider than 64 columns (tiers S, M1,
   M2, whose capacities keep every score far inside int16); 2 = the same for tier L while nodes + columns <= CW_POA_PK_SPAN, i.e.
   while |score| <= 8 * span stays clear of the packed "minus infinity" even after the per-column gap offsets are taken out. */
#define CW_POA_PK_SPAN 2600
/* CM: 0 = the matrix path only; 2 = members of <= 63 bases take the recorded-decision path of cw_poa_c.h (tiers S / M1 / M2) */
#ifndef CW_POA_CODES
#define CW_POA_CODES 1
#endif
#include "cw_poa_a.h"

template <typename HT, int PK, int CM = 0, int LCAP = 1023> /* LCAP: the tier's longest member -- fills for wider rows are not compiled into its kernel */
__device__ __forceinline__ int poa_run(const PoaMem<HT>& M, const PoaTask& t, const DevBatch& b, const DevScratch& sc, const int lane,
                       unsigned long long (&acc)[6]) {
    if constexpr (CW_POA_AFFINE && sizeof(HT) != 4) return 2; /* the affine gap model keeps three int32 layers: every task is handed on to the global-memory tier */
    unsigned long long _pt = __builtin_readcyclecounter();
#define POA_PROF(slot) do { const unsigned long long _n = __builtin_readcyclecounter(); acc[slot] += _n - _pt; _pt = _n; } while (0)
    const int G = CW_POA_GAP, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH;
    int n = 0, ne = 0, nseq = 0, tpl_nodes = 0;
    bool meta_ok = false;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int gf_left = 0, gf_idx = 0; /* group fill (poa_fill_pk4): members behind the current one whose rows are already in the matrix; the current one's place in the group */
    /* the replay of a repeated member (round 6; cw_poa_q.h has the same in its per-group form): a member that is, base for base, the member aligned just before
       it -- which changed nothing in the graph but coverage counts -- would make the same fill, walk and path: its merge is that path's coverage counts
       once more (M.pcur, M.sq still hold path and bases).  Measured on the checker: half of the members of the large tiers' tasks (the short ragged pieces of a
       window's first and last segment), 2 % of tier S's.  Not under the heaviest-bundle policy (the path's edge weights would have to go up too). */
    bool prev_clean = false; /* the member before this one was aligned and added neither a node nor an edge */
    int prev_L = -1;

    for (uint32_t mi = 0; mi < t.n_members; ++mi) {
        const PoaMember pm = sc.members[t.member_off + mi];
        const int L = __builtin_amdgcn_readfirstlane((int)pm.len); /* sizes steer every loop below: keep them in scalar registers */
        n = __builtin_amdgcn_readfirstlane(n); ne = __builtin_amdgcn_readfirstlane(ne);
        if ((uint32_t)L > M.l_cap) return 2;
#if CW_POA_REPLAY
        if (!CW_CONS_HEAVIEST_BUNDLE && prev_clean && L == prev_L) {
            const uint32_t* words_ = b.bases + b.seq_word_off[pm.seq];
            bool same = true;
            for (int j = lane; j < L; j += 64) same = same && M.sq[j] == (uint8_t)cw_base_at(words_, pm.start + j);
            if (__ballot(!same) == 0ull) {
                for (int j = lane; j < L; j += 64) { const int cur = M.pcur[j]; M.ncov[cur] = (uint16_t)(M.ncov[cur] + 1); } /* (a path visits a node once; lane j owns position j, as in the merge) */
                nseq++;
                if constexpr (CW_POA_GROUP_FILL != 0 && PK != 0 && CM == 0 && sizeof(HT) == 2 && LCAP >= 511) {
                    if (gf_left > 0 && meta_ok) { ++gf_idx; --gf_left; } /* its rows of a group fill stay unread */
                }
                cw_wave_sync();
                continue;
            }
        }
#endif
        {
            const uint32_t* words = b.bases + b.seq_word_off[pm.seq];
            for (int j = lane; j < L; j += 64) M.sq[j] = (uint8_t)cw_base_at(words, pm.start + j);
        }
        cw_wave_sync();
        nseq++;
        prev_clean = false; prev_L = L;
        if (n == 0) { /* first member: a chain */
            if ((uint32_t)L > M.n_cap || (uint32_t)L > M.e_cap) return 2;
            for (int j = lane; j < L; j += 64) {
                M.nbase[j] = M.sq[j]; M.ncov[j] = 1; M.nalc[j] = 0;
                M.in_head[j] = j ? (uint16_t)(j - 1) : CW_NONE16; M.in_tail[j] = M.in_head[j];
                M.indeg[j] = j ? 1 : 0; M.has_out[j] = (j < L - 1) ? 1 : 0;
                M.r2n[j] = (uint16_t)j; M.n2r[j] = (uint16_t)j;
                if (j) { M.efrom[j - 1] = (uint16_t)(j - 1); M.enext[j - 1] = CW_NONE16; if (CW_CONS_HEAVIEST_BUNDLE) M.ew[j - 1] = 1; }
            }
            n = L; ne = L - 1; tpl_nodes = L; meta_ok = false;
            cw_wave_sync();
            continue;
        }
        const int cols = L + 1;
        /* group fill (tiers M2 / L, matrix path): is this member's fill already in the matrix -- filled together with its predecessor, and has no merge
           since changed the graph's structure (meta_ok) -- or does it open a group with the members behind it? */
        bool grp = false, grp_fill = false;
        int jo = 0, gk = 1;
        if constexpr (CW_POA_GROUP_FILL != 0 && PK != 0 && CM == 0 && sizeof(HT) == 2 && LCAP >= 511) {
            if (gf_left > 0 && meta_ok) { grp = true; jo = 32 * (++gf_idx); --gf_left; }
            else {
                gf_left = 0; gf_idx = 0;
                if (L <= CW_GF_LC && (uint32_t)((n + 1) * CW_GF_HS) <= M.h_cap && n + 32 <= CW_POA_PK_SPAN) {
                    while (gk < 4 && mi + (uint32_t)gk < t.n_members && __builtin_amdgcn_readfirstlane((int)sc.members[t.member_off + mi + gk].len) <= CW_GF_LC) ++gk;
                    if (gk >= 2) { grp = true; grp_fill = true; gf_left = gk - 1; }
                }
            }
        }
        const bool packed = grp || (PK && cols > 64 && (PK == 1 || n + cols <= CW_POA_PK_SPAN)); /* two columns per lane (int16 tiers, wide rows) */
        /* pad64 rests on an ordering the HSA memory model does not spell out: a lane beyond the member's columns stores into the first
           cells of LATER rows, which another lane of the same wave overwrites when that row is finished -- correct iff two stores of one wave to
           one address commit in issue order (they do on gfx9: one wave's vector memory instructions reach the L2 in order).  -DCW_NO_PAD64 builds
           the masked variant; tests/test_gpu_variants.py runs both against the oracle. */
#ifdef CW_NO_PAD64
        const bool pad = false;
#else
        const bool pad = PK != 0 && !packed && M.pad64 && cols <= 64;
#endif
        const int hs = grp ? CW_GF_HS : packed ? ((cols + 1) & ~1) : cols;    /* row stride of the DP matrix */
        if ((uint32_t)((n + 1) * hs + (pad ? 64 : 0)) > M.h_cap) return 2;

        /* ---- per-rank metadata (parallel over ranks) ---- */
        if (!meta_ok) {
            int run = 0;
            if (CM == 2 && M.gflag) { /* cw_poa_c.h: which rows the fill also writes to the slab */
                for (int w = lane; w < (int)(M.n_cap / 32 + 2); w += 64) M.gflag[w] = 0u;
                cw_wave_sync();
            }
            for (int r0 = 0; r0 < n; r0 += 64) {
                const int r = r0 + lane;
                const int node = r < n ? M.r2n[r] : 0;
                const int d = r < n ? M.indeg[node] : 0;
                const int inc = cw_wave_scan_add(d);
                const int off = run + inc - d;
                if (r < n) {
                    int q = off, first = 0;
                    bool far_ = false; /* a predecessor more than CW_RING ranks back */
                    for (uint32_t e = M.in_head[node]; e != CW_NONE16; e = M.enext[e]) {
                        const int pr = M.n2r[M.efrom[e]] + 1;
                        if (q == off) first = pr;
                        M.plist[q++] = (uint16_t)pr;
                        far_ = far_ || r + 1 - pr > CW_RING;
                        if (CM == 2 && M.gflag && (r + 1 - pr > CW_RING || q - off > 3)) /* (the fourth in-edge and later: the traceback compares their cells, cw_poa_c.h) */
                            __hip_atomic_fetch_or((cwc_l32)M.gflag + ((pr - 1) >> 5), 1u << ((pr - 1) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    if (CM == 2 && M.gflag && d > 3) __hip_atomic_fetch_or((cwc_l32)M.gflag + (r >> 5), 1u << (r & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    M.rpred0[r] = (uint16_t)first;
                    const uint32_t np_ = (uint32_t)(d ? d : 1);
                    const bool lin_ = np_ == 1u && first == r;
                    /* cw_poa_c.h row kinds: 0 the rank before (registers); 1 / 2 / 3 one, two, three predecessors that are all in the row
                       store (any earlier row in tier S; at most CW_RING ranks back in tiers M1 / M2); 5 everything else */
                    uint32_t kind_ = lin_ ? 0u : 5u;
                    if (CM != 0 && !lin_ && np_ <= 3u && first != 0 && !far_) kind_ = np_;
                    M.rmeta[r] = CW_RM_WORD(M.nbase[node], np_, lin_, !M.has_out[node], kind_, np_ == 1u ? first : off);
                }
                if (__ballot(r < n && (uint32_t)d > CW_RM_MAX_PRED) != 0ull) return 2; /* more in-edges than the row word's 11 bits count: a node needs more than 2047 distinct sources for that
                                                                                      (members > 2047; tier G then reports rc 2 and the window stops on CW_WHY_POA) */
                run += cw_lane_value(inc, 63);
            }
            meta_ok = true;
            cw_wave_sync();
            if (CM == 2 && M.gflag) { /* the slab bit of the row word: set by the rows that will read the row back from the slab */
                for (int r = lane; r < n; r += 64) if ((M.gflag[r >> 5] >> (r & 31)) & 1u) M.rmeta[r] |= 16u;
                cw_wave_sync();
            }
            if (M.p2) { /* two and four steps up the first-predecessor chain: a traceback tile gets its eight rows in three reads, not seven */
                for (int r = lane; r < n; r += 64) {
                    const int a = M.rpred0[r];
                    M.p2[r] = a > 0 ? M.rpred0[a - 1] : CW_NONE16;
                }
                cw_wave_sync();
                for (int r = lane; r < n; r += 64) {
                    const uint32_t b2 = M.p2[r];
                    M.p4[r] = (b2 != CW_NONE16 && b2 > 0) ? M.p2[b2 - 1] : CW_NONE16;
                }
                cw_wave_sync();
            }
        }
        POA_PROF(0);

        /* ---- DP fill, end cell, traceback ---- */
        for (int j = lane; j < L; j += 64) M.seqrank[j] = CW_NONE16;
        bool coded = false; /* members of <= 63 bases in tiers S / M1 / M2: the fill records the traceback's decisions (cw_poa_c.h) */
        if constexpr (CM != 0 && CW_POA_CODES != 0) {
            if (cols <= 64 && M.codes != nullptr) {
                if ((uint32_t)(((n + 7) >> 3) * 64) > M.c_cap || (uint32_t)((n + 1) * 64) > M.h_cap) return 2;
                coded = true;
            }
        }
#ifdef CW_POA_VERIFY
        const bool run_matrix_path = true;
#else
        const bool run_matrix_path = !coded && !CW_POA_AFFINE;
#endif
        if constexpr (CW_POA_AFFINE && sizeof(HT) == 4) { /* cw_poa_a.h: fill, end cell and walk back under the affine gap model */
            int end_row = 0;
            const int arc = poa_affine_align(M, n, L, lane, &end_row);
            if (arc) return arc;
            POA_PROF(1);
        }
        if (run_matrix_path) {
        PoaMem<HT> V = M; /* the member's view of the matrix: in a group fill its columns begin at jo */
        V.H = M.H + jo;
        if (!grp) { for (int j = lane; j < cols; j += 64) M.H[j] = (HT)(CW_POA_SW ? 0 : j * G); }
        cw_wave_sync();
        const int nch = grp ? 1 : packed ? (cols + 127) >> 7 : (cols + 63) >> 6; /* direction-word chunks per row */
        const bool use_dirs = !CW_POA_OV && (uint32_t)(n * nch * (packed ? 2 : 1)) <= M.d_cap; /* (overlap mode: the tile walk below only) */
        if constexpr (PK != 0) {
            if (grp) {
                if constexpr (CW_POA_GROUP_FILL != 0 && CM == 0 && sizeof(HT) == 2 && LCAP >= 511) {
                    if (grp_fill) { /* this member and the gk - 1 behind it, one per 16-lane row: their bases side by side in M.sq */
                        for (int x = lane; x < 128; x += 64) { const int g_ = x >> 5, j_ = x & 31; if (g_ >= 1) M.sq[x] = 255; (void)j_; }
                        if (lane >= L && lane < 32) M.sq[lane] = 255;
                        uint32_t lens4 = (uint32_t)L;
                        for (int g_ = 1; g_ < gk; ++g_) {
                            const PoaMember pg = sc.members[t.member_off + mi + g_];
                            const uint32_t* wg = b.bases + b.seq_word_off[pg.seq];
                            if (lane < (int)pg.len) M.sq[32 * g_ + lane] = (uint8_t)cw_base_at(wg, pg.start + lane);
                            lens4 |= (uint32_t)__builtin_amdgcn_readfirstlane((int)pg.len) << (8 * g_);
                        }
                        cw_wave_sync();
                        poa_fill_pk4<PK == 2>(M, n, lane, use_dirs, lens4);
                    }
                }
            }
            else if (!packed) {
                if (cols <= 64) { if (pad) poa_fill<HT, 1, PK == 2, true>(M, n, cols, lane, use_dirs); else poa_fill<HT, 1, PK == 2>(M, n, cols, lane, use_dirs); }
                else if constexpr (PK == 2 && LCAP > 511) { /* a very large graph in tier L: one column per lane */
                    if (cols <= 128) poa_fill<HT, 2, PK == 2>(M, n, cols, lane, use_dirs);
                    else if (cols <= 256) poa_fill<HT, 4, PK == 2>(M, n, cols, lane, use_dirs);
                    else if (cols <= 512) poa_fill<HT, 8, PK == 2>(M, n, cols, lane, use_dirs);
                    else poa_fill<HT, 16, PK == 2>(M, n, cols, lane, use_dirs);
                }
            }
            else if (cols <= 128) poa_fill_pk<1, PK == 2>(M, n, cols, hs, lane, use_dirs);
#if CW_POA_LW
            else if (PK == 2 && M.comm != nullptr) { if constexpr (PK == 2 && sizeof(HT) == 2) poa_fill_mw<true>(M, n, cols, hs, lane, use_dirs); } /* tier L: the chunks of the row on the waves of the work-group (cw_poa_w.h) */
#endif
            else if constexpr (LCAP > 127) {
                if (cols <= 256) poa_fill_pk<2, PK == 2>(M, n, cols, hs, lane, use_dirs);
                else if constexpr (LCAP > 255) {
                    if (cols <= 512) poa_fill_pk<4, PK == 2>(M, n, cols, hs, lane, use_dirs);
                    else if constexpr (PK == 2 && LCAP > 511) poa_fill_pk<8, PK == 2>(M, n, cols, hs, lane, use_dirs);
                }
            }
        } else {
            if (cols <= 64) poa_fill<HT, 1, PK == 2>(M, n, cols, lane, use_dirs);
            else if (cols <= 128) poa_fill<HT, 2, PK == 2>(M, n, cols, lane, use_dirs);
            else if (cols <= 256) poa_fill<HT, 4, PK == 2>(M, n, cols, lane, use_dirs);
            else if (cols <= 512) poa_fill<HT, 8, PK == 2>(M, n, cols, lane, use_dirs);
            else if (cols <= 1024) poa_fill<HT, 16, PK == 2>(M, n, cols, lane, use_dirs);
            else poa_fill<HT, 32, PK == 2>(M, n, cols, lane, use_dirs); /* tier G only: members of up to 2047 bases (round 6) */
        }
        POA_PROF(1);
        if (PK == 2 && LCAP > 511 && lane == 0) { atomicAdd(&sc.ctr->prof[46], (unsigned long long)n * (unsigned long long)nch); atomicAdd(&sc.ctr->prof[47], (unsigned long long)n); }

        /* ---- end cell: best sink in the last column, lowest rank on ties ---- */
        int bi, bj = L;
        if (CW_POA_OV) { /* overlap mode: the best cell of a sink's row, columns 1..L; lowest rank, then lowest column on ties */
            int bs = CW_NEG * 2, br = 0x7FFFFFFF, bc = L;
            for (int r = lane; r < n; r += 64) {
                if (!CW_POA_SW && M.has_out[M.r2n[r]]) continue; /* (local mode: any row) */
                for (int j = 1; j <= L; ++j) {
                    const int h = V.H[(r + 1) * hs + j];
                    if (h > bs) { bs = h; br = r; bc = j; } /* ranks ascend within a lane, columns within a row */
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o), oc = __shfl_xor(bc, o);
                if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; bc = oc; }
            }
            bi = __builtin_amdgcn_readfirstlane(br) + 1;
            bj = __builtin_amdgcn_readfirstlane(bc);
        } else {
            int bs = CW_NEG * 2, br = 0x7FFFFFFF;
            for (int r = lane; r < n; r += 64) {
                if (M.has_out[M.r2n[r]]) continue;
                const int h = V.H[(r + 1) * hs + L];
                if (h > bs) { bs = h; br = r; } /* ranks ascend within a lane */
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o);
                if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; }
            }
            bi = __builtin_amdgcn_readfirstlane(br) + 1; /* wave-uniform from here on */
        }

        /* ---- traceback (wave-uniform); records seqrank[j] = rank aligned to sequence position j ---- */
        {
            int i = bi, j = bj;
            if (use_dirs && M.runs) {
                /* Direction words: every lane looks at the cell it would reach if the path kept going the way it starts
                   (diagonal: (i-t, j-t); vertical: (i-t, j); horizontal: (i, j-t)) along a linear stretch of the graph
                   (first predecessor == previous rank), so one round trip consumes a whole run of equal moves. */
                while (i > 0) {
                    const int t = lane;
                    const int row = i - t;
                    const bool rvalid = row >= 1;
                    const int prow = rvalid ? (int)M.rpred0[row - 1] : -1;
                    const bool linear = rvalid && prow == row - 1;
                    int cD = 3, cV = 3, cH = 3;
                    if (rvalid) {
                        cV = poa_dir_code(M.dirs, row - 1, j + jo, nch, packed);
                        if (j - t >= 1) cD = poa_dir_code(M.dirs, row - 1, j - t + jo, nch, packed);
                    }
                    if (j - t >= 1) cH = poa_dir_code(M.dirs, i - 1, j - t + jo, nch, packed);
                    const int code0 = __builtin_amdgcn_readlane(cV, 0);
                    const int pr0 = __builtin_amdgcn_readlane(prow, 0);
                    if (code0 == 2) {
                        const unsigned long long bad = ~__ballot(cH == 2);
                        const int run = bad ? (__ffsll((long long)bad) - 1) : 64;
                        j -= run;
                    } else if (code0 == 3) {
                        /* several predecessors: decide from the cell values (same order of preference) */
                        int pi, pj;
                        if (!poa_slow_step(V, i, j, hs, pr0, lane, &pi, &pj)) return 3;
                        if (pj != j && pi != i && lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                        i = pi; j = pj;
                    } else if (pr0 != i - 1) {
                        /* the predecessor is not the previous rank: a single step */
                        if (code0 == 0) { if (lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1); j--; }
                        i = pr0;
                    } else if (code0 == 0) {
                        const unsigned long long bad = ~__ballot(cD == 0 && linear);
                        const int run = bad ? (__ffsll((long long)bad) - 1) : 64;
                        if (t < run) M.seqrank[j - 1 - t] = (uint16_t)(i - 1 - t);
                        i -= run; j -= run;
                    } else {
                        const unsigned long long bad = ~__ballot(cV == 1 && linear);
                        const int run = bad ? (__ffsll((long long)bad) - 1) : 64;
                        i -= run;
                    }
                }
            } else if (use_dirs) {
                /* direction words, one step per LDS round trip: the row's words and its first predecessor together */
                while (i > 0) {
                    const int code = poa_dir_code(M.dirs, i - 1, j + jo, nch, packed);
                    const int pr0 = M.rpred0[i - 1];
                    if (code == 0) {
                        if (lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                        i = pr0; j--;
                    } else if (code == 1) {
                        i = pr0;
                    } else if (code == 2) {
                        j--;
                    } else {
                        int pi, pj;
                        if (!poa_slow_step(V, i, j, hs, pr0, lane, &pi, &pj)) return 3;
                        if (pj != j && pi != i && lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                        i = pi; j = pj;
                    }
                }
            } else {
                /* no direction words (matrix lives in the slab): walk 8x8 tiles of the matrix held one cell per lane, so
                   that up to 14 steps cost one memory round trip.  The tile's rows are the first-predecessor chain
                   i, p(i), p(p(i)), ...; lane (tr,tc) = (lane>>3, lane&7) holds H[chain[tr]][j-tc].  Every lane decides the move
                   out of its own cell from its three neighbours in the tile (same order of preference as everywhere: diagonal
                   through the first predecessor, other predecessors, vertical, horizontal); the path is then three ballots
                   followed on the scalar unit. */
                const int tr = lane >> 3, tc = lane & 7;
#ifdef CW_DIAG
                if (lane == 0 && M.diag) atomicAdd(&M.diag[9], 1ull);
#endif
                bool vprobe = false; /* the last trip was seven vertical moves in one column: look down the column */
                while (i > 0) {
#ifdef CW_DIAG
                    if (lane == 0 && M.diag) atomicAdd(&M.diag[7], 1ull);
#endif
                    i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j); /* wave-uniform: keep the walk on the scalar unit */
#if CW_POA_VPROBE
                    /* Round 5: the tall tiers align short members against graphs of hundreds of nodes -- most of such a path is ONE vertical run of
                       hundreds of moves, seven per tile trip.  Inside a run the sixty-four lanes look down the column instead: lane t holds cell
                       (i - t, j); along a stretch of the graph where every node's only predecessor is the rank before, the cell moves vertically iff
                       the diagonal candidate does not explain it (the order of preference) and the cell above + gap does.  The leading lanes that
                       do are consumed at once: up to 63 moves per round trip. */
                    if (vprobe) {
                        const int row_t = i - lane;
                        const uint32_t m_t = row_t >= 1 ? M.rmeta[row_t - 1] : 0u;
                        int c_t = 0, d_t = 0;
                        if (row_t >= 0) c_t = (int)V.H[row_t * hs + j];
                        if (row_t >= 1 && j > 0) d_t = (int)V.H[(row_t - 1) * hs + j - 1];
                        const int u_t = __shfl_down(c_t, 1); /* the cell above: lane t + 1's */
                        const int s_t = (j > 0 && (int)M.sq[j - 1] == (int)(m_t & 3u)) ? MS : XS;
                        bool vert_t = row_t >= 1 && lane < 63 && CW_RM_LIN(m_t) && !(j > 0 && c_t == d_t + s_t) && c_t == u_t + G;
                        if (CW_POA_SW) vert_t = vert_t && c_t != 0; /* (local mode: the walk stops at a cell of value 0) */
                        if (CW_POA_OV && j <= 0) vert_t = false;    /* (overlap mode: it stops in column 0: the tile's business) */
                        const unsigned long long vb = __ballot(vert_t);
                        const int run = __ffsll((long long)~vb) - 1;
                        i -= run;
                        vprobe = run >= 32;
                        if (run > 0) continue;
                    }
#endif
                    int row = i;
                    if (M.p2) { /* tr steps up the chain: 4 + 2 + 1 (unconditional reads and selects, see cw_poa_c.h) */
                        { uint32_t v = M.p4[row - 1]; asm volatile("" : "+v"(v)); row = (tr & 4) ? (v == CW_NONE16 ? -1 : (int)v) : row; }
                        { uint32_t v = M.p2[max(row, 1) - 1]; asm volatile("" : "+v"(v)); row = (tr & 2) ? ((row > 0 && v != CW_NONE16) ? (int)v : -1) : row; }
                        { uint32_t v = M.rpred0[max(row, 1) - 1]; asm volatile("" : "+v"(v)); row = (tr & 1) ? (row > 0 ? (int)v : -1) : row; }
                    } else {
                        int ch[8];
                        ch[0] = i;
#pragma unroll
                        for (int q = 1; q < 8; ++q) ch[q] = ch[q - 1] > 0 ? (int)M.rpred0[ch[q - 1] - 1] : -1;
#pragma unroll
                        for (int q = 1; q < 8; ++q) row = (tr == q) ? ch[q] : row;
                    }
                    const int col = j - tc;
                    const bool valid = row >= 0 && col >= 0;
                    const int hv = valid ? (int)V.H[row * hs + col] : 0;
                    const int meta_r = row >= 1 ? (int)M.rmeta[row - 1] : 0; /* the 8 lanes of a tile row read one word */
                    const int sq_c = col >= 1 ? (int)M.sq[col - 1] : 255;
                    const int av = __shfl_down(hv, 9), bv = __shfl_down(hv, 8), lv = __shfl_down(hv, 1); /* (tr+1,tc+1), (tr+1,tc), (tr,tc+1) */
                    /* 0 diagonal, 1 vertical, 2 horizontal; 3 several predecessors, 4 no move explains the cell, 5 neighbours outside
                       the tile, 6 the virtual start row */
                    /* (selects, not branches: as an if-chain this was six nested execution-mask regions per tile) */
                    const bool c_start = row <= 0 || (CW_POA_OV && col <= 0) || (CW_POA_SW && valid && hv == 0), c_edge = tr == 7 || (tc == 7 && col > 0); /* (overlap mode: the walk stops in column 0; local mode: also at a cell of value 0) */
                    const bool c_d = col > 0 && hv == av + (sq_c == (meta_r & 3) ? MS : XS);
                    const bool c_m = CW_RM_NP((uint32_t)meta_r) != 1, c_v = hv == bv + G, c_h = col > 0 && hv == lv + G;
                    const int code = c_start ? 6 : c_edge ? 5 : c_d ? 0 : c_m ? 3 : c_v ? 1 : c_h ? 2 : 4;
                    const unsigned long long m_d = __ballot(code == 0), m_v = __ballot(code == 1), m_h = __ballot(code == 2);
                    /* the walk, a whole diagonal run per trip: the cells pos, pos + 9, ... are one 64-bit mask; the first of them that does not
                       move diagonally ends the run (every diagonal of the tile ends in an edge cell, which never does), then one vertical or
                       horizontal step.  A step at a time on the scalar unit was ~20 instructions for each of up to 14 steps. */
                    int pos = 0;
                    unsigned long long on_diag = 0ull;
                    for (;;) {
                        const unsigned long long dm = 0x8040201008040201ull << pos;
                        const unsigned long long stop = ~m_d & dm;
                        const int first = stop ? __ffsll((long long)stop) - 1 : 63;
                        on_diag |= dm & ((1ull << first) - 1ull);
                        pos = first;
                        if ((m_v >> pos) & 1ull) pos += 8;
                        else if ((m_h >> pos) & 1ull) pos += 1;
                        else break;
                    }
                    if ((on_diag >> lane) & 1ull) M.seqrank[col - 1] = (uint16_t)(row - 1);
                    const int end_code = __builtin_amdgcn_readlane(code, pos);
                    i = __builtin_amdgcn_readlane(row, pos);
                    j -= pos & 7;
                    vprobe = CW_POA_VPROBE && pos == 56 && on_diag == 0ull; /* seven vertical moves, no other: probably inside a long run */
                    if (end_code == 4) return 3;
                    if (end_code == 6) i = 0;
                    if (end_code == 3) {
                        /* a node with several predecessors: one step decided from direct reads (same order of preference) */
                        const int pr0 = __builtin_amdgcn_readfirstlane((int)M.rpred0[i - 1]);
                        int pi, pj;
#ifdef CW_DIAG
                        if (lane == 0 && M.diag) atomicAdd(&M.diag[8], 1ull);
#endif
                        if (!poa_slow_step(V, i, j, hs, pr0, lane, &pi, &pj)) return 3;
                        if (pj != j && pi != i && lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                        i = pi; j = pj;
                    }
                }
            }
            /* i == 0: the remaining sequence positions are insertions, already CW_NONE16 */
        }
#ifdef CW_POA_VERIFY
        if (coded) { /* keep the matrix path's answer, then let the recorded-decision path give its own */
            cw_wave_sync();
            for (int j = lane; j < L; j += 64) { M.pat[j] = M.seqrank[j]; M.seqrank[j] = CW_NONE16; }
            if (lane == 0) M.pat[L] = (uint16_t)bi;
            cw_wave_sync();
        }
#endif
        }
        if constexpr (CM != 0 && CW_POA_CODES != 0) {
            if (coded) {
                cw_wave_sync();
                const bool lc = M.lcodes != nullptr && (uint32_t)(((n + 7) >> 3) * cols + 64) <= M.lc_cap; /* the code words fit the LDS area (stride = columns) */
#ifdef CW_DIAG
                if (lane == 0 && M.diag) { atomicAdd(&M.diag[9], 1ull); if (lc) atomicAdd(&M.diag[8], 1ull); } /* members on this path / with their code words in LDS */
#endif
                const int be_c = poa_fill_c<CM>(M, n, cols, lane, lc);
                const int bi_c = be_c & 0xFFFF, bj_c = be_c >> 16; /* end row, end column (the last one unless cw_policy.h's overlap mode) */
                POA_PROF(1);
                if (!poa_trace_c<CM>(M, n, bi_c, bj_c, cols, lane, lc)) return 3;
#ifdef CW_POA_VERIFY
                cw_wave_sync();
                {
                    bool bad = false;
                    for (int j = lane; j < L; j += 64) bad = bad || M.pat[j] != M.seqrank[j];
                    const unsigned long long bb = __ballot(bad);
                    const bool bad_bi = (int)M.pat[L] != bi_c;
                    if (lane == 0) {
                        atomicAdd(&sc.ctr->prof[120], 1ull);
                        if (bb != 0ull || bad_bi) {
                            if (atomicAdd(&sc.ctr->prof[121], 1ull) == 0ull) { /* the first difference: task window, member, columns, rows, end rows */
                                sc.ctr->prof[122] = ((unsigned long long)t.window << 32) | mi;
                                sc.ctr->prof[123] = ((unsigned long long)cols << 32) | (unsigned)n;
                                sc.ctr->prof[124] = ((unsigned long long)M.pat[L] << 32) | (unsigned)bi_c;
                                sc.ctr->prof[125] = bb;
                            }
                        }
                    }
                }
#endif
            }
        }
        cw_wave_sync();
        POA_PROF(2);

        /* ---- merge the path into the graph: one lane per sequence position ---- */
        {
            const int n_old = n;
            bool edges_added = false;
            const int chunks = (L + 63) >> 6;
            /* pass A (last chunk first): resolve existing nodes, find the rank slot of fresh ones */
            int next_rank = -1; /* rank aligned to the nearest later position that has one */
            for (int c = chunks - 1; c >= 0; --c) {
                const int j = c * 64 + lane;
                const bool act = j < L;
                const uint32_t rk = act ? M.seqrank[j] : CW_NONE16;
                const unsigned long long has = __ballot(act && rk != CW_NONE16);
                /* nearest later aligned position: inside this chunk if any, else carried from later chunks */
                const unsigned long long later = has & ~(lt_mask | (1ull << lane));
                const int later_rank = (int)(uint32_t)__shfl((int)rk, later ? (__ffsll((long long)later) - 1) : 0);
                const int qr = later ? later_rank : next_rank;
                const int first_rank = (int)(uint32_t)__shfl((int)rk, has ? (__ffsll((long long)has) - 1) : 0);
                uint32_t cur = CW_NONE16, at = CW_NONE16;
                if (act) {
                    const int bcode = M.sq[j];
                    if (rk != CW_NONE16) {
                        const int pn = M.r2n[rk];
                        if (M.nbase[pn] == bcode) cur = (uint32_t)pn;
                        else {
                            const int ac = M.nalc[pn];
                            int last = (int)rk;
                            for (int a = 0; a < ac; ++a) {
                                const int v = M.nal[pn * 3 + a];
                                if (M.nbase[v] == bcode) cur = (uint32_t)v;
                                last = max(last, (int)M.n2r[v]);
                            }
                            if (cur == CW_NONE16) at = (uint32_t)(last + 1); /* new member of pn's column */
                        }
                    } else if (qr < 0) {
                        at = (uint32_t)n_old; /* insertion with no aligned position after it: goes last */
                    } else {
                        /* insertion: in front of the column of the next position that is aligned to a node */
                        const int q = M.r2n[qr];
                        int first = qr;
                        for (int a = 0; a < M.nalc[q]; ++a) first = min(first, (int)M.n2r[M.nal[q * 3 + a]]);
                        at = (uint32_t)first;
                    }
                    M.pcur[j] = (uint16_t)cur;
                    M.pat[j] = (uint16_t)at;
                }
                if (has) next_rank = first_rank;
            }
            cw_wave_sync();
            /* pass B (first chunk first): ids for fresh nodes, node records, coverage */
            int fresh_total = 0;
            for (int c = 0; c < chunks; ++c) {
                const int j = c * 64 + lane;
                const bool act = j < L;
                const bool fresh = act && M.pcur[j] == CW_NONE16;
                const unsigned long long fb = __ballot(fresh);
                if (fresh) {
                    const int cur = n_old + fresh_total + __popcll(fb & lt_mask);
                    if ((uint32_t)cur < M.n_cap) {
                        M.pcur[j] = (uint16_t)cur;
                        M.nbase[cur] = M.sq[j]; M.ncov[cur] = 1; M.nalc[cur] = 0;
                        M.in_head[cur] = CW_NONE16; M.in_tail[cur] = CW_NONE16; M.indeg[cur] = 0; M.has_out[cur] = 0;
                        const uint32_t rk = M.seqrank[j];
                        if (rk != CW_NONE16) { /* joins the column of the node it was aligned to */
                            const int pn = M.r2n[rk];
                            const int ac = M.nalc[pn];
                            for (int a = 0; a < ac; ++a) {
                                const int v = M.nal[pn * 3 + a];
                                M.nal[cur * 3 + a] = (uint16_t)v;
                                M.nal[v * 3 + M.nalc[v]] = (uint16_t)cur; M.nalc[v] = (uint8_t)(M.nalc[v] + 1);
                            }
                            M.nal[cur * 3 + ac] = (uint16_t)pn; M.nalc[cur] = (uint8_t)(ac + 1);
                            M.nal[pn * 3 + ac] = (uint16_t)cur; M.nalc[pn] = (uint8_t)(ac + 1);
                        }
                    }
                } else if (act) {
                    const int cur = M.pcur[j];
                    M.ncov[cur] = (uint16_t)(M.ncov[cur] + 1);
                }
                fresh_total += __popcll(fb);
            }
            if ((uint32_t)(n_old + fresh_total) > M.n_cap) return 2;
            cw_wave_sync();
            /* pass C: place the fresh nodes in the rank order (one histogram + one prefix sum) */
            if (fresh_total > 0) {
                uint32_t* hist = (uint32_t*)M.plist; /* n_old + 1 counters */
                for (int r = lane; r <= n_old; r += 64) hist[r] = 0;
                cw_wave_sync();
                for (int c = 0; c < chunks; ++c) {
                    const int j = c * 64 + lane;
                    if (j < L && M.pat[j] != CW_NONE16) atomicAdd(&hist[M.pat[j]], 1u);
                }
                cw_wave_sync();
                int run = 0;
                for (int r0 = 0; r0 < n_old; r0 += 64) {
                    const int r = r0 + lane;
                    const int hcount = r < n_old ? (int)hist[r] : 0;
                    const int inc = cw_wave_scan_add(hcount);
                    if (r < n_old) {
                        const int nr = r + run + inc; /* shifted by every fresh node placed at a slot <= r */
                        const int v = M.r2n[r];
                        M.rtmp[nr] = (uint16_t)v;
                        M.n2r[v] = (uint16_t)nr;
                    }
                    run += cw_lane_value(inc, 63);
                }
                for (int c = 0; c < chunks; ++c) {
                    const int j = c * 64 + lane;
                    if (j < L && M.pat[j] != CW_NONE16) {
                        const int cur = M.pcur[j];
                        const int nr = (int)M.pat[j] + (cur - n_old); /* slot + fresh nodes before it on the path */
                        M.rtmp[nr] = (uint16_t)cur;
                        M.n2r[cur] = (uint16_t)nr;
                    }
                }
                cw_wave_sync();
                n = n_old + fresh_total;
                for (int r = lane; r < n; r += 64) M.r2n[r] = M.rtmp[r];
                meta_ok = false;
                cw_wave_sync();
            }
            /* pass D: edges between consecutive sequence positions */
            for (int c = 0; c < chunks; ++c) {
                const int j = c * 64 + lane;
                const bool act = j < L && j > 0;
                int head = 0, cur = 0;
                bool add = false;
                if (act) {
                    head = M.pcur[j - 1]; cur = M.pcur[j];
                    add = true;
                    for (uint32_t e = M.in_head[cur]; e != CW_NONE16; e = M.enext[e])
                        if (M.efrom[e] == (uint16_t)head) { add = false; if (CW_CONS_HEAVIEST_BUNDLE) M.ew[e] = (uint16_t)(M.ew[e] + 1); break; } /* (a path uses an edge once: no two lanes meet here) */
                }
                const unsigned long long ab = __ballot(add);
                const int total = __popcll(ab);
                if ((uint32_t)(ne + total) > M.e_cap) return 2;
                if (add) {
                    const int e = ne + __popcll(ab & lt_mask);
                    M.efrom[e] = (uint16_t)head; M.enext[e] = CW_NONE16;
                    if (CW_CONS_HEAVIEST_BUNDLE) M.ew[e] = 1;
                    const uint32_t tl = M.in_tail[cur];
                    if (tl == CW_NONE16) M.in_head[cur] = (uint16_t)e; else M.enext[tl] = (uint16_t)e;
                    M.in_tail[cur] = (uint16_t)e;
                    M.indeg[cur] = (uint16_t)(M.indeg[cur] + 1);
                    M.has_out[head] = 1;
                }
                if (total) { ne += total; meta_ok = false; edges_added = true; }
            }
            cw_wave_sync();
            prev_clean = n == n_old && !edges_added; /* nothing but coverage counts changed: the next member may be a replay of this one */
        }
        POA_PROF(3);
    }

    /* ---- consensus: column-majority vote, or the heaviest bundle (cw_policy.h CW_POA_CONSENSUS) ---- */
    uint32_t out_len = 0;
#if CW_CONS_HEAVIEST_BUNDLE
    cw_wave_sync();
    if (lane == 0) out_len = poa_consensus_hb(M, n, t, sc);
    out_len = (uint32_t)__shfl((int)out_len, 0);
#else
    for (int r0 = 0; r0 < n; r0 += 64) {
        const int r = r0 + lane;
        int emit = -1;
        if (r < n) {
            const int v = M.r2n[r];
            const int ac = M.nalc[v];
            bool first = true;
            for (int a = 0; a < ac; ++a) if (M.n2r[M.nal[v * 3 + a]] < r) first = false;
            if (first) {
                int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                int tpl_code = -1;
                for (int c = 0; c <= ac; ++c) {
                    const int u = M.r2n[r + c];
                    const int code = M.nbase[u];
                    const int cv = M.ncov[u];
                    c0 += code == 0 ? cv : 0; c1 += code == 1 ? cv : 0; c2 += code == 2 ? cv : 0; c3 += code == 3 ? cv : 0;
                    if (u < tpl_nodes) tpl_code = code;
                }
                const int gaps = nseq - (c0 + c1 + c2 + c3);
                int top = 0, tc = c0;
                if (c1 > tc) { top = 1; tc = c1; }
                if (c2 > tc) { top = 2; tc = c2; }
                if (c3 > tc) { top = 3; tc = c3; }
                if (!CW_CONS_DROPS(gaps, tc)) { /* cw_policy.h "switches" */
                    const int tplc = tpl_code == 0 ? c0 : tpl_code == 1 ? c1 : tpl_code == 2 ? c2 : tpl_code == 3 ? c3 : -1;
                    if (CW_CONS_TEMPLATE_WINS_TIES && tplc == tc) top = tpl_code;
                    emit = top;
                }
            }
        }
        const unsigned long long bal = __ballot(emit >= 0);
        const uint32_t idx = out_len + (uint32_t)__popcll(bal & lt_mask);
        if (emit >= 0 && idx < t.out_cap) sc.arena[t.out_off + idx] = CW_ACGT(emit);
        out_len += (uint32_t)__popcll(bal);
    }
#endif
    if (out_len > t.out_cap) return 3;
    if (lane == 0) sc.seg_len[t.seg_slot] = out_len;
    POA_PROF(4);
#undef POA_PROF
    return 1;
}

__device__ __forceinline__ void poa_flush_prof(const DevScratch& sc, int base, const unsigned long long (&acc)[6], int lane) {
    if (lane == 0) {
        for (int q = 0; q < 5; ++q) atomicAdd(&sc.ctr->prof[base + q], acc[q]);
        atomicMax(&sc.ctr->prof[36 + (base - 8) / 5], acc[5]); /* the longest single task of this tier */
    }
}

__device__ __forceinline__ void poa_hand_over(const DevScratch& sc, const PoaTask& t, uint32_t ti, int rc, int next_tier) {
    /* lane 0 only: rc 2 = this tier's capacity was exceeded -> next tier; rc 3 = output capacity / internal -> window overflow */
    if (rc == 2 && next_tier < CW_TIERS) {
        const uint32_t bi = atomicAdd(&sc.ctr->n_over[next_tier], 1u);
        /* the entry itself is the flag (pre-set to 0xFFFFFFFF): tier L may be consuming this list while we produce */
        if (bi < sc.list_cap) __hip_atomic_store(&sc.over_list[next_tier][bi], ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else { sc.win[t.window].status = CW_WIN_OVERFLOW; sc.win[t.window].pad_ = CW_WHY_POA; sc.ctr->any_overflow = 1; }
    } else if (rc != 1) {
        sc.win[t.window].status = CW_WIN_OVERFLOW; sc.win[t.window].pad_ = CW_WHY_POA; sc.ctr->any_overflow = 1;
    }
    sc.tasks[ti].state = (uint32_t)rc;
}

/* a producing tier's work-group is done: publish (release) and count it */
__device__ __forceinline__ void poa_producer_done(const DevScratch& sc) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&sc.ctr->done_wgs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* ---- tier S: one task per wave, work-stealing over the task list ---------------------------------------------------
 * Round 4: laid out like tiers M1 / M2 -- what the fill and the traceback read in LDS (5.9 KB per wave with the row ring of cw_poa_c.h,
 * where graph, merge arrays, a 2048-cell matrix and the code words took 12.9), everything only the merge and the rank bookkeeping touch,
 * the matrix of the rare member of more than 63 bases, flagged rows and the code words in a slab the wave claims (sc.slab[0]).
 * Capacities: CW_POA_NC nodes, CW_POA_LC bases; the 2048-cell limit is gone. */
#ifndef CW_S_EU
#define CW_S_EU 4
#endif
#ifndef CW_M1_EU
#define CW_M1_EU 4
#endif
__global__ void __launch_bounds__(64 * CW_POA_WAVES, CW_S_EU) cw_poa_kernel(DevBatch b, DevScratch sc) { /* round 6: four waves per SIMD (128 VGPRs, no scratch): at five (96 VGPRs) the round-5 kernel spilled 28 VGPRs into its row loops, and its measured residency is 3.6 waves per SIMD (LDS-bound) anyway */
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t gw = 0;
    if (lane == 0) { /* claim a free slab, see cw_poa_slab_kernel */
        const uint32_t n_slots = sc.slots[0];
        uint32_t s = (uint32_t)(((unsigned long long)(blockIdx.x * CW_POA_WAVES + wave) * 2654435761ull) % n_slots);
        for (;;) {
            if (__hip_atomic_load(&sc.slot_busy[0][s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
                atomicCAS(&sc.slot_busy[0][s], 0u, 1u) == 0u) break;
            s = s + 1u == n_slots ? 0u : s + 1u;
        }
        gw = s;
    }
    gw = (uint32_t)__builtin_amdgcn_readfirstlane((int)gw);
    typedef __attribute__((address_space(1))) uint8_t* cw_gptr;
    uint8_t* my_slab = (uint8_t*)(cw_gptr)(sc.slab[0] + (size_t)gw * sc.slab_bytes[0]);
    int16_t* hslab = (int16_t*)my_slab;
    unsigned long long* dslab = (unsigned long long*)(my_slab + CW_POA_HSLAB_BYTES(CW_POA_NC, CW_POA_LC));
    uint8_t* cold = my_slab + CW_POA_HSLAB_BYTES(CW_POA_NC, CW_POA_LC) + CW_POA_DSLAB_BYTES(CW_POA_NC, CW_POA_LC);
    PoaMem<int16_t> M = poa_carve<int16_t>(lds + (size_t)wave * CW_POA_SLAB_BYTES, CW_POA_NC, CW_POA_EC, CW_POA_LC, (CW_POA_NC + 1) * (CW_POA_LC + 1), 0, hslab, dslab, cold, !CW_S_EDGES_LDS, true);
    static_assert(CW_POA_SMAX * (CW_POA_NC + CW_POA_LC) <= CW_POA_I16_BOUND && 4 * CW_POA_SMAX * (CW_POA_NC + 64) <= CW_POA_I16_BOUND, "include/cw_policy.h \"Bounds\": tier S");
    M.H = hslab; M.dirs = dslab;
    {
        uint8_t* extra = lds + (size_t)wave * CW_POA_SLAB_BYTES + (CW_POA_SLAB_BYTES - 4 * CW_S_LCODES_WORDS - CW_POA_RING_BYTES - CW_POA_GFLAG_BYTES(CW_POA_NC));
        M.ring = (int16_t*)extra; M.gflag = (uint32_t*)(extra + CW_POA_RING_BYTES);
        M.codes = (uint32_t*)dslab; M.c_cap = (uint32_t)(CW_POA_DSLAB_BYTES(CW_POA_NC, CW_POA_LC) / 4);
        if (CW_S_LCODES_WORDS) { M.lcodes = (uint32_t*)(lds + (size_t)wave * CW_POA_SLAB_BYTES + (CW_POA_SLAB_BYTES - 4 * CW_S_LCODES_WORDS)); M.lc_cap = CW_S_LCODES_WORDS; }
    }
    M.pad64 = true;
    const uint32_t n_tasks = min(sc.ctr->n_tasks, sc.task_cap);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
#ifdef CW_DIAG
    M.diag = &sc.ctr->prof[72];
#endif
    /* see cw_poa_slab_kernel: all but the last persist_wgs work-groups take a chunk of tasks and end */
    const bool yields = blockIdx.x + sc.persist_wgs[0] < gridDim.x;
    uint32_t ran = 0;
    {   /* what outgrew tier Q (its kernel ran before this one on the same stream: the list is complete) */
        const uint32_t n_q = min(sc.ctr->n_over[0], sc.list_cap);
        for (;;) {
            uint32_t qi = 0;
            if (lane == 0) qi = atomicAdd(&sc.ctr->next_over[0], 1u);
            qi = (uint32_t)__shfl((int)qi, 0);
            if (qi >= n_q) break;
            const uint32_t ti = sc.over_list[0][qi];
            const PoaTask t = sc.tasks[ti];
            const int rc = poa_run<int16_t, 1, 2, CW_POA_LC>(M, t, b, sc, lane, acc);
            if (lane == 0) poa_hand_over(sc, t, ti, rc, 3);
            cw_wave_sync();
        }
    }
    for (;;) {
        uint32_t ti = 0;
        if (lane == 0) ti = atomicAdd(&sc.ctr->next_task, 1u);
        ti = (uint32_t)__shfl((int)ti, 0);
        if (ti >= n_tasks) break;
        const PoaTask t = sc.tasks[ti];
        if (t.state != 0) continue; /* routed to a larger tier by the index kernel */
        const unsigned long long _t0 = __builtin_readcyclecounter();
        const int rc = poa_run<int16_t, 1, 2, CW_POA_LC>(M, t, b, sc, lane, acc);
        { const unsigned long long d = __builtin_readcyclecounter() - _t0; acc[5] = d > acc[5] ? d : acc[5]; }
        if (lane == 0) poa_hand_over(sc, t, ti, rc, 3);
        cw_wave_sync();
        if (yields && ++ran >= CW_POA_CHUNK_S) break;
    }
    poa_flush_prof(sc, 8, acc, lane);
    if (lane == 0) __hip_atomic_store(&sc.slot_busy[0][gw], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); /* the slab goes back */
    poa_producer_done(sc);
}

/* ---- tiers M1 / M2 / L: graph in LDS, DP matrix in this wave's global slab ------------------------ */
/* PASS 0 works through the tasks the index kernel routed to this tier (all tiers run concurrently on their own
   streams); tier L additionally drains the live overflow queue.  PASS 1 (tier L only, after the join) takes what is left. */
#define CW_POAL_LDS_BYTES (CW_POA_HOT2L_BYTES(CW_POAL_NC, CW_POAL_EC, CW_POAL_LC) * CW_POAL_WAVES) /* tier L's work-group */
#define CW_POALW_LDS_BYTES (CW_POA_HOT2L_BYTES(CW_POAL_NC, CW_POAL_EC, CW_POAL_LC) + CW_POA_COMM_BYTES) /* ... and the four-wave one's (tier LW: one task) */
/* MW > 1 (tier LW, TIER = 3): ONE task per work-group of MW waves -- wave 0 is "the" wave of the code below, the others serve its FILL commands
   (cw_poa_w.h) and share its LDS arrays; LIST: the routed list the kernel works through (tier LW: list 5, which the product build has free) */
template <int NC, int EC, int LC, int WAVES, int TIER, int PASS, int MW = 1, int LIST = TIER>
__global__ void __launch_bounds__(64 * WAVES * MW, TIER == 1 ? CW_M1_EU : TIER == 2 ? 4 : 1) /* M1 and M2: four waves per SIMD (128 VGPRs; M1 at five spilled 17 VGPRs, round 6; M2 spills 3 at four and none at three -- 148 VGPRs -- where a batch alone on the GPU measured 60.2-60.4 ms against 59.6-59.7: left at four) */
cw_poa_slab_kernel(DevBatch b, DevScratch sc) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    static_assert(MW == 1 || (TIER == 3 && WAVES == 1 && MW == CW_POAL_MW && CW_POA_LW), "several waves per task: tier L's wide tasks only");
    const int mw_wave = MW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int wave = MW > 1 ? 0 : (int)(threadIdx.x >> 6);
    PoaComm* const comm = MW > 1 ? (PoaComm*)(lds + CW_POA_HOT2L_BYTES(NC, EC, LC) * WAVES) : nullptr;
    /* Yielding persistence.  The four tier kernels run side by side and share each CU's LDS; a work-group that loops until its tier's
       list is empty keeps its LDS for the whole stage, so whichever kernel reaches a CU first owns it (measured: tier S held every CU
       for 27 ms of a depth-150 batch while the long tasks of tier L had not started).  Here only the LAST persist_wgs work-groups of
       a grid are persistent; every earlier one takes a chunk of tasks and ends, and the dispatcher hands its LDS to the next pending
       work-group of ANY tier: the mix on a CU follows the remaining work instead of the launch order.  A wave's slab is therefore
       not tied to its block index: it claims a free one (there are more slabs than waves the hardware can hold at once). */
    uint32_t gw = 0;
    if (lane == 0 && mw_wave == 0) {
        const uint32_t n_slots = sc.slots[TIER];
        uint32_t s = (uint32_t)(((unsigned long long)(blockIdx.x * WAVES + wave) * 2654435761ull) % n_slots);
        for (;;) {
            if (__hip_atomic_load(&sc.slot_busy[TIER][s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
                atomicCAS(&sc.slot_busy[TIER][s], 0u, 1u) == 0u) break;
            s = s + 1u == n_slots ? 0u : s + 1u;
        }
        gw = s;
#if CW_POA_LW
        if constexpr (MW > 1) { comm->slab = s; comm->seq = 0u; comm->cmd = 0u; for (int x = 0; x < CW_POAL_MW; ++x) { comm->done[x] = 0u; comm->ready[x] = 0u; } }
#endif
    }
#if CW_POA_LW
    if constexpr (MW > 1) { __syncthreads(); if (lane == 0) gw = comm->slab; } /* (the one barrier of the kernel: the helpers learn the slab) */
#endif
    gw = (uint32_t)__builtin_amdgcn_readfirstlane((int)gw);
    const bool yields = PASS == 0 && MW == 1 && blockIdx.x + sc.persist_wgs[TIER] < gridDim.x; /* (the few work-groups of tier LW stay until their list is empty) */
    /* the slab is global memory: say so, or every access to the DP matrix is a flat_* instruction (both wait counters, aperture check) */
    typedef __attribute__((address_space(1))) uint8_t* cw_gptr;
    uint8_t* my_slab = (uint8_t*)(cw_gptr)(sc.slab[TIER] + (size_t)gw * sc.slab_bytes[TIER]);
    int16_t* hslab = (int16_t*)my_slab;
    unsigned long long* dslab = (unsigned long long*)(my_slab + CW_POA_HSLAB_BYTES(NC, LC));
    uint8_t* cold = my_slab + CW_POA_HSLAB_BYTES(NC, LC) + CW_POA_DSLAB_BYTES(NC, LC);
    constexpr uint32_t slab = TIER <= 2 ? CW_POA_HOT2T_BYTES(TIER, NC, EC, LC) : CW_POA_HOT2L_BYTES(NC, EC, LC);
    PoaMem<int16_t> M = poa_carve<int16_t>(lds + (size_t)wave * slab, NC, EC, LC, (NC + 1) * (LC + 1), (TIER >= 3 || (TIER == 2 && CW_M2_DIRS)) ? CW_POA_DSLAB_PAIRS(NC, LC) : 0, hslab, dslab, cold,
                                           true, TIER == 1 || (TIER == 2 && (CW_M2_CHAIN_TABS || CW_M2_CODES)), TIER >= 3 && CW_L_COLD_NODES);
    M.H = hslab; M.dirs = dslab; /* again, without poa_carve's either-or: these two are now provably global pointers */
    /* include/cw_policy.h "Bounds": the int16 values of this tier's fills stay inside +-CW_POA_I16_BOUND -- by its capacities, or (tier L under scores
       beyond 11) by handing a graph of more nodes than that allows on to tier G */
    static_assert(TIER == 3 || CW_POA_SMAX * (NC + LC) <= CW_POA_I16_BOUND, "matrix / packed fills of this tier");
    static_assert(4 * CW_POA_SMAX * (NC + 64) <= CW_POA_I16_BOUND || !(TIER == 1 || (TIER == 2 && CW_M2_CODES)), "recorded decisions of this tier");
    if constexpr (TIER == 3 && CW_POA_SMAX * (NC + LC) > CW_POA_I16_BOUND) M.n_cap = (uint32_t)(CW_POA_I16_BOUND / CW_POA_SMAX - LC);
    static_assert(TIER != 3 || CW_POA_I16_BOUND / CW_POA_SMAX > LC + 64, "tier L keeps a useful node capacity under these scores");
    if (TIER <= 2 && CW_POA_HOTC_OF_TIER(TIER)) { /* cw_poa_c.h: ring and flags behind the hot arrays in LDS, code words where tier L keeps its direction words */
        uint8_t* extra = lds + (size_t)wave * slab + (slab - CW_POA_RING_BYTES - CW_POA_GFLAG_BYTES(NC));
        M.ring = (int16_t*)extra; M.gflag = (uint32_t*)(extra + CW_POA_RING_BYTES);
        M.codes = (uint32_t*)dslab; M.c_cap = (uint32_t)(CW_POA_DSLAB_BYTES(NC, LC) / 4);
    }
    M.runs = TIER >= 3 || (TIER == 2 && CW_M2_DIRS); /* tier L: long graphs against short members, long vertical runs (direction words, whole runs per round trip) */
    M.comm = comm;
#ifdef CW_DIAG
    M.diag = &sc.ctr->prof[72 + 12 * TIER];
#endif
    M.pad64 = true; /* the slab's matrix area is (NC + 1) x (LC + 1) cells: a matrix of <= 64 columns leaves more than 64 cells free */
    /* the large tiers have few, long tasks and share their SIMDs with up to three waves of the small tiers: let them issue first,
       or tier L is still running long after the others have finished (depth 150) */
    if (TIER == 3) __builtin_amdgcn_s_setprio(3);
#if CW_POA_LW
    if constexpr (MW > 1) {
        if (mw_wave > 0) { poa_mw_serve<true>(M, lane, mw_wave); return; } /* until wave 0 posts EXIT */
    }
#endif
#ifndef CW_M2_PRIO
#define CW_M2_PRIO 1
#endif
    if (TIER == 2) __builtin_amdgcn_s_setprio(CW_M2_PRIO);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    auto run_task = [&](uint32_t ti) {
        const PoaTask t = sc.tasks[ti];
        if (t.n_members == 0) return; /* a neutral entry: the chain kernel ran out of task or list slots (cw_chain.h "cap_ok") */
        const unsigned long long _t0 = __builtin_readcyclecounter(), _w0 = wall_clock64();
        /* tier M2's rows are 85 % linear (long graphs, short members): there the matrix fill's 37-instruction row beats the recorded
           decisions' 52, and its slower traceback does not make up for it (measured: 35.8 against 38.5 G wave-cycles per batch) */
        const int rc = poa_run<int16_t, ((TIER < 3 && !(TIER == 2 && CW_M2_DIRS)) ? 1 : 2), (TIER == 1 || (TIER == 2 && CW_M2_CODES) ? 2 : 0), LC>(M, t, b, sc, lane, acc);
        const unsigned long long _t1 = __builtin_readcyclecounter();
        acc[5] = _t1 - _t0 > acc[5] ? _t1 - _t0 : acc[5];
        if (lane == 0 && sc.task_dbg) { /* inspection aid (CW_TASK_TRACE): when each task of the slab tiers ran (10 ns units since the tier sort), where, and how it ended */
            uint4 d;
            d.x = (uint32_t)(_w0 - sc.ctr->prof[41]); d.y = (uint32_t)(wall_clock64() - _w0); d.z = (uint32_t)TIER | ((uint32_t)rc << 8) | ((uint32_t)PASS << 16) | (MW > 1 ? 1u << 24 : 0u); d.w = gw;
            sc.task_dbg[ti] = d;
        }
        if (lane == 0) poa_hand_over(sc, t, ti, rc, TIER < 3 ? 3 : 4);
        cw_wave_sync();
    };
    if (PASS == 0) {
        const uint32_t* list = sc.tier_list[LIST];
        const uint32_t n_work = min(sc.ctr->n_tier[LIST], sc.list_cap);
        uint32_t ran = 0;
        for (;;) {
            uint32_t mi = 0;
            if (lane == 0) mi = atomicAdd(&sc.ctr->next_tier[LIST], 1u);
            mi = (uint32_t)__shfl((int)mi, 0);
            if (mi >= n_work) break;
            run_task(list[mi]);
            if (yields && ++ran >= (TIER == 1 ? CW_POA_CHUNK_M1 : 1u)) break;
        }
    }
    if (TIER == 3 && MW == 1 && !yields && (PASS == 1 || gridDim.x - 1u - blockIdx.x < sc.linger_wgs)) {
        /* Live queue (only the last few work-groups of the grid stay for it: a lingering tier-L work-group holds 37 KB of LDS that
           the other tiers could use): tasks that outgrow tiers S/M1/M2 while those kernels are still running on their own streams are
           picked up here at once instead of waiting for a later pass.  An entry is its own flag (0xFFFFFFFF = not yet
           written); we stop when every producing work-group has signed off and the queue is drained.  Every wait is
           bounded: whatever is left (e.g. if this kernel ran before its producers) is handled by the PASS 1 launch. */
        uint32_t idle = 0;
        for (;;) {
            uint32_t claimed = 0xFFFFFFFFu, stop = 0;
            if (lane == 0) {
                const uint32_t done = __hip_atomic_load(&sc.ctr->done_wgs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t n = min(__hip_atomic_load(&sc.ctr->n_over[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), sc.list_cap);
                uint32_t c = __hip_atomic_load(&sc.ctr->next_over[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c < n) {
                    if (__hip_atomic_compare_exchange_strong(&sc.ctr->next_over[3], &c, c + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) claimed = c;
                } else if (PASS == 1 || done >= sc.producer_wgs) {
                    stop = 1;
                }
            }
            claimed = (uint32_t)__shfl((int)claimed, 0);
            stop = (uint32_t)__shfl((int)stop, 0);
            if (stop) break;
            if (claimed == 0xFFFFFFFFu) {
                if (++idle > (1u << 13)) break; /* ~15 ms of polling: give up, PASS 1 takes the rest */
                __builtin_amdgcn_s_sleep(64);
                continue;
            }
            idle = 0;
            uint32_t ti = 0xFFFFFFFFu;
            for (uint32_t spin = 0; spin < (1u << 20) && ti == 0xFFFFFFFFu; ++spin)
                ti = __hip_atomic_load(&sc.over_list[3][claimed], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ti == 0xFFFFFFFFu) { /* cannot happen: a claimed slot is written right after its index was taken */
                if (lane == 0) sc.ctr->any_overflow = 2;
                break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            run_task(ti);
        }
    } else if (PASS == 1) {
        /* not used for tiers below L */
    }
#if CW_POA_LW
    if constexpr (MW > 1) { /* the helpers leave */
        if (lane == 0) comm->cmd = CW_MW_CMD_EXIT;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) cww_store(&comm->seq, cww_load(&comm->seq) + 1u);
    }
#endif
    poa_flush_prof(sc, 8 + 5 * TIER, acc, lane);
    if (lane == 0) __hip_atomic_store(&sc.slot_busy[TIER][gw], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); /* the slab goes back */
    if (PASS == 0 && TIER < 3) poa_producer_done(sc);
}

/* ---- largest tasks first ---------------------------------------------------------------------------
 * A task that outgrows a tier is redone in tier L, and tiers M2 and L hold few, long tasks: whatever is going to take long or to outgrow
 * its tier has to start early, or tier L finishes long after everything else (depth 150: the last hand-overs used to land when M1/M2
 * were nearly done).  One work-group per tier orders that tier's routed list by estimated cost, largest first: a counting sort into 64
 * classes (quarter octaves of members x longest member squared for M2 and L, steps of the depth-aware graph-size estimate for M1;
 * the order inside a class does not matter), with per-wave counters so that the LDS atomics of one wave do not queue behind the
 * other fifteen.  The second buffer is the tier's hand-over list, which nothing uses before the tier kernels run (tier L's is
 * re-initialised afterwards by the kernel itself). */
#define CW_SORT_CLASSES 128
#define CW_SORT_LDS_CLS 131072 /* classes of the first so many list entries are kept in LDS between the two passes */
__device__ __forceinline__ uint32_t cw_sort_class(uint32_t n_members, uint32_t len_pair, int tier) {
    /* PoaTask::max_len = longest member | mean member length << 16.  A task's time is its members' fills: rows (the graph: about the
       longest member, growing with every member) x columns (that member's length), so members x mean length x longest member; the
       piles of the first and last segment of a window are ragged (one long member among short ones), which the longest member alone
       overstates by far (measured: ordering by members x longest^2 left 27-ms tasks of tier L to start after 40 ms) */
    const uint32_t max_len = len_pair & 0xFFFFu, mean_len = len_pair >> 16;
    uint32_t c;
    if (tier == 0) { /* tier Q: the four tasks of a wave advance in lock step, so neighbours in the list should be alike: longest members first,
                        then by how many there are */
#ifdef CW_Q_SORT_R4 /* rounds 3-4: longest member first, three steps of eight members */
        const uint32_t nm = n_members >> 3;
        return (CW_SORT_CLASSES - 1) - ((max_len < 32u ? max_len : 31u) * 4u + (nm < 3u ? nm : 3u));
#else
        /* round 5: a wave's four tasks end when its longest one does, and a task's time is members x rows x ...: since tier Q also takes the deep piles'
           tasks (up to maxMSA members), the member count comes first -- eight classes, roughly geometric -- then the longest member in steps of two */
        /* (round 6, after the replay of repeated members: sixteen classes of member count and eight of length instead -- tier Q's kernel 8.86 -> 8.70 ms, the step
           unchanged within its noise over four alternating runs: not kept.  Nor is the order by the members that are NOT repeats, counted by the chain kernel where
           it writes the member lists: tier Q's kernel 8.85 -> 8.45 ms, the chain kernel 2.61 -> 3.05 ms for reading every member's bases: even) */
        const uint32_t mc = n_members < 4u ? 0u : n_members < 8u ? 1u : n_members < 12u ? 2u : n_members < 16u ? 3u : n_members < 24u ? 4u : n_members < 32u ? 5u : n_members < 64u ? 6u : 7u;
        return (CW_SORT_CLASSES - 1) - (mc * 16u + ((max_len < 32u ? max_len : 31u) >> 1));
#endif
    }
    if (tier == 5) { /* tier H (cw_poa_q.h): as tier Q, the longest member in steps of four */
        const uint32_t mc = n_members < 4u ? 0u : n_members < 8u ? 1u : n_members < 12u ? 2u : n_members < 16u ? 3u : n_members < 24u ? 4u : n_members < 32u ? 5u : n_members < 64u ? 6u : 7u;
        return (CW_SORT_CLASSES - 1) - (mc * 16u + ((max_len < 64u ? max_len : 63u) >> 2));
    }
    if (tier == 1) {
        c = ((max_len * (15u + n_members / 5u) + 9u) / 10u) >> 2;
    } else {
        /* fitted on the task timeline of a depth-150 batch (tools/task_trace.py, CW_FIT): tier L time ~ members^1.76 x mean length^0.84 x
           longest^0.21, tier M2 members^1.54 x mean^0.49 x longest^0.48 -- the graph grows with every member, so members count twice */
        const unsigned long long cost = (unsigned long long)n_members * n_members * (mean_len ? mean_len : max_len) + 1ull;
        const int lg = 63 - __clzll((long long)cost);                                   /* floor(log2) */
        const uint32_t frac = lg >= 3 ? (uint32_t)((cost >> (lg - 3)) & 7ull) : 0u;     /* next three bits: eighths of an octave */
        const int q = lg * 8 + (int)frac - 8 * 13;                                      /* costs below 2^13 share the last class */
        c = q < 0 ? 0u : (uint32_t)q;
    }
    return (CW_SORT_CLASSES - 1) - (c < CW_SORT_CLASSES ? c : CW_SORT_CLASSES - 1); /* class 0 = the largest */
}
__global__ void __launch_bounds__(1024) cw_sort_tier_kernel(DevScratch sc, uint32_t lds_cls) {
    __shared__ uint32_t cnt[16][CW_SORT_CLASSES];
    __shared__ uint32_t tot_c[CW_SORT_CLASSES];
    extern __shared__ __attribute__((aligned(16))) uint8_t cls_lds[]; /* lds_cls bytes (<= CW_SORT_LDS_CLS) */
    const int tier = blockIdx.x == 3 ? 0 : blockIdx.x == 4 ? 5 : 1 + (int)blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63; /* 0 = tier Q's list, 5 = tier H's */
    if (blockIdx.x == 0 && threadIdx.x == 0) sc.ctr->prof[41] = wall_clock64(); /* time base of the task trace */
    const int cls_tier = tier == 5 && sc.use_lw ? 3 : tier; /* list 5 holds tier L's wide tasks (tier LW): ordered like tier L's */
    const uint32_t n = min(sc.ctr->n_tier[tier], sc.list_cap);
    const uint32_t nthr = blockDim.x, nwv = nthr >> 6; /* 4 .. 16 waves: a small work-group finds room beside another batch's persistent kernels */
    uint32_t* list = sc.tier_list[tier];
    uint32_t* tmp = sc.over_list[tier];
    if (n < 2) return;
    for (uint32_t x = threadIdx.x; x < 16 * CW_SORT_CLASSES; x += nthr) (&cnt[0][0])[x] = 0;
    __syncthreads();
    /* every wave owns a contiguous slice of the list in both passes; eight entries per lane in flight (two dependent global reads each) */
    const uint32_t per = (n + nwv - 1u) / nwv, lo = min(n, (uint32_t)wave * per), hi = min(n, lo + per);
    for (uint32_t x0 = lo; x0 < hi; x0 += 512) {
        uint32_t ti[8];
        uint2 nm[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t x = x0 + (uint32_t)j * 64u + (uint32_t)lane; ti[j] = x < hi ? list[x] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < 8; ++j) nm[j] = ti[j] != 0xFFFFFFFFu ? *(const uint2*)&sc.tasks[ti[j]].n_members : make_uint2(0, 0); /* n_members, max_len */
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = x0 + (uint32_t)j * 64u + (uint32_t)lane;
            if (ti[j] != 0xFFFFFFFFu) {
                const uint32_t c = cw_sort_class(nm[j].x, nm[j].y, cls_tier);
                atomicAdd(&cnt[wave][c], 1u);
                if (x < lds_cls) cls_lds[x] = (uint8_t)c;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < CW_SORT_CLASSES) { uint32_t t = 0; for (uint32_t w = 0; w < nwv; ++w) t += cnt[w][threadIdx.x]; tot_c[threadIdx.x] = t; }
    __syncthreads();
    if (threadIdx.x < CW_SORT_CLASSES) { /* class-major, then wave: exclusive offsets */
        uint32_t before = 0;
        for (uint32_t c = 0; c < threadIdx.x; ++c) before += tot_c[c];
        for (uint32_t w = 0; w < nwv; ++w) { const uint32_t k = cnt[w][threadIdx.x]; cnt[w][threadIdx.x] = before; before += k; }
    }
    __syncthreads();
    for (uint32_t x0 = lo; x0 < hi; x0 += 512) {
        uint32_t ti[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t x = x0 + (uint32_t)j * 64u + (uint32_t)lane; ti[j] = x < hi ? list[x] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = x0 + (uint32_t)j * 64u + (uint32_t)lane;
            if (ti[j] != 0xFFFFFFFFu) {
                uint32_t c;
                if (x < lds_cls) c = cls_lds[x];
                else { const uint2 v = *(const uint2*)&sc.tasks[ti[j]].n_members; c = cw_sort_class(v.x, v.y, cls_tier); }
                tmp[atomicAdd(&cnt[wave][c], 1u)] = ti[j];
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < n; x += nthr) list[x] = tmp[x];
    if (tier == 3) { /* the live queue of tier L: an entry is its own flag */
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < n; x += nthr) tmp[x] = 0xFFFFFFFFu;
    }
}

/* ---- tier G: everything in this wave's global slab (int32 cells) -------------------------------- */
__global__ void __launch_bounds__(64 * CW_POA_WAVES) cw_poa_big_kernel(DevBatch b, DevScratch sc) {
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * CW_POA_WAVES + (threadIdx.x >> 6);
    if (gw >= sc.slots[4]) return;
    const PoaMem<int32_t> M = poa_carve<int32_t>(sc.slab[4] + (size_t)gw * sc.slab_bytes[4], CW_POAB_NC, CW_POAB_EC, CW_POAB_LC, CW_POAB_HC, 0);
    const uint32_t n_big = min(sc.ctr->n_over[4], sc.list_cap);
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (;;) {
        uint32_t bi = 0;
        if (lane == 0) bi = atomicAdd(&sc.ctr->next_over[4], 1u);
        bi = (uint32_t)__shfl((int)bi, 0);
        if (bi >= n_big) break;
        const uint32_t ti = sc.over_list[4][bi];
        const PoaTask t = sc.tasks[ti];
        const int rc = poa_run<int32_t, 0, 0>(M, t, b, sc, lane, acc);
        if (lane == 0) poa_hand_over(sc, t, ti, rc, CW_TIERS);
        cw_wave_sync();
    }
    poa_flush_prof(sc, 8 + 5 * 4, acc, lane);
}

#endif
