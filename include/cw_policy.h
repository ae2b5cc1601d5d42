/*
 * cw_policy.h -- every free policy of the segmented-POA restatement, named in one place.
 *
 * SHARED CONSTANTS (not oracle code).  This header carries no algorithm, only constants; it is the
 * single file both the CPU oracle (oracle/) and the HIP engine (consent_amd/csrc/) include so that
 * the two sides cannot drift apart on a tie-break.
 *
 * Why these are "policies": the reference calls `MSABMAAC(piles, merSize, bmeanSup, solidThresh,
 * minAnchors, maxMSA, path)` (/root/reference/src/correctionMSA.cpp:32,54) which lives in the
 * un-vendored submodule Malfoy/BMEAN (+ the spoa library bundled inside it, reference Makefile:3).
 * Neither is present under /root/reference and the commit pins are unknown (.gitmodules has URLs
 * only), so the arithmetic below restates the *published* algorithm (CONSENT README.md:93-100,
 * Sci Rep 11:761; spoa = Lee 2002 POA + heaviest bundle, Vaser 2017) and every decision the
 * publication leaves open is a constant here.  PARITY FOR THESE ROWS IS UNPINNED.
 */
#ifndef CW_POLICY_H
#define CW_POLICY_H

/* --- POA alignment engine (A4d) ------------------------------------------------------------- */
/* Global (Needleman-Wunsch) alignment of each segment string against the partial-order graph:   */
/* segments are cut at shared anchors, so both ends are pinned and a global mode is the natural  */
/* choice.  Linear gap model.  Scores are spoa's command-line defaults (m=5, n=-4, g=-8).        */
/* Overridable at build time on BOTH sides (-DCW_POA_MATCH=2 -DCW_POA_MISMATCH=-4 -DCW_POA_GAP=-4 ...): spoa's library defaults differ   */
/* between its versions, and the version BMEAN bundles is unknown.  Everything in the engine is derived from these three (the recorded-   */
/* decision fill's scaled constants 4 * (MATCH - GAP), 4 * (MISMATCH - GAP), 4 * GAP included); tests/test_gpu_policy.py builds both      */
/* sides with another triple.  Bounds: below.                                                                                             */
#ifndef CW_POA_MATCH
#define CW_POA_MATCH      5
#endif
#ifndef CW_POA_MISMATCH
#define CW_POA_MISMATCH (-4)
#endif
#ifndef CW_POA_GAP
#define CW_POA_GAP      (-8)
#endif
#if CW_POA_GAP >= 0 || CW_POA_GAP < -16 || CW_POA_MATCH <= 0 || CW_POA_MATCH > 16 || CW_POA_MISMATCH > 0 || CW_POA_MISMATCH < -16 || CW_POA_MATCH <= CW_POA_MISMATCH
#error "cw_policy.h: CW_POA_MATCH in 1..16, CW_POA_MISMATCH in -16..0, CW_POA_GAP in -16..-1 (linear gap, int16 DP tiers)"
#endif
/* Bounds (round 6: |score| <= 16; <= 8 through round 5).  With SMAX the largest absolute score, a DP value of a graph of n nodes against L bases lies
   inside +-SMAX * (n + L); the engine's int16 tiers keep every value inside +-29000 (their "no value" is -30000):
     matrix and packed fills (members of more than 63 bases, tiers M2 / L):  SMAX * (nodes + bases) <= 29000;
     recorded decisions (values x 4, members of at most 63 bases):            4 * SMAX * (nodes + 64) <= 29000.
   Every tier's capacities satisfy both up to SMAX = 16 except tier L's 1536 nodes + 1023 bases, which do up to SMAX = 11: beyond that tier L hands a
   graph of more than 29000 / SMAX - 1023 nodes on to the int32 tier G (cw_poa.h CW_POA_NCAP_I16).  The oracle computes in int32. */
#define CW_POA_ABS_(x) ((x) < 0 ? -(x) : (x))
#define CW_POA_MAX2_(a, b) ((a) > (b) ? (a) : (b))
#define CW_POA_SMAX CW_POA_MAX2_(CW_POA_MATCH, CW_POA_MAX2_(CW_POA_ABS_(CW_POA_MISMATCH), CW_POA_ABS_(CW_POA_GAP)))
#define CW_POA_I16_BOUND 29000

/* Traceback preference at a cell (spoa sisd engine order): diagonal through the in-edges in     */
/* insertion order, then vertical (graph node against a gap) through the in-edges in insertion   */
/* order, then horizontal (sequence base against a gap).                                         */
/* End cell for the global mode: the sink node (no out-edge) of LOWEST topological rank among    */
/* those with the best last-column score (strict '<' while scanning ranks upward).               */

/* Rank order of the graph (rows of the DP, columns of the MSA): maintained incrementally, never     */
/* re-sorted.  The first sequence's nodes take ranks 0..len-1.  A fresh node created for a sequence   */
/* base is ranked (a) right after the last member of the column it joins, when it is a mismatch       */
/* against an existing node, or (b) right before the first member of the column of the next path      */
/* node that already has a rank (at the very end if there is none), when it is an insertion.          */
/* The members of one column therefore always hold consecutive ranks.  Only two things depend on the  */
/* order beyond being topological: the tie-break between equally good end nodes, and the order in    */
/* which independent insertion columns appear in the consensus.                                       */

/* Consensus of a segment = column-majority vote over the MSA encoded by the graph (BMEAN's        */
/* "easy consensus"), NOT the heaviest bundle: with anchors that occasionally hit a spurious copy of  */
/* a k-mer, one member of a segment can be hundreds of bases longer than the others, and a           */
/* sum-of-weights path would follow that singleton detour.  Columns = aligned groups in topological  */
/* order.  gaps = sequences - sum(base counts).  Column dropped iff gaps > every base count.         */
/* Otherwise the most frequent base; ties -> the template's base if tied for the top, else the       */
/* smallest code (A<C<G<T).                                                                          */

/* --- switches ------------------------------------------------------------------------------- */
/* Policies with more than one defensible reading are build-time switches that BOTH sides honour     */
/* (-DNAME=value for oracle/ and for consent_amd/csrc/; tests/test_gpu_policy.py builds both with a   */
/* non-default consensus and checks that they still agree with each other and differ from the        */
/* default).  A value that is named here but not implemented stops the build on both sides.          */
#define CW_POA_MODE_NW 0 /* global alignment of the segment against the graph (implemented)          */
#define CW_POA_MODE_SW 1 /* local (Smith-Waterman, spoa's kSW as published): first row and first column 0, no cell below 0, the alignment ends in the best
                            cell anywhere (columns 1..L, lowest rank then lowest column on ties) and stops at the first cell of value 0 (or in the first row /
                            column); the bases outside it are insertions.  Implemented on both sides since round 5; the engine runs it on its matrix paths
                            only (no recorded decisions, tier Q with its DP matrix: cw_poa.h) */
#define CW_POA_MODE_OV 2 /* semi-global / overlap, as spoa's kOV is published: first row gap-penalised, first column free, the alignment
                            ends in the best cell (columns 1..L, lowest rank then lowest column on ties) of a node without out-edges and stops where
                            it reaches the first row or column; the bases beyond the end cell and before the stop are insertions.  Implemented on both
                            sides since round 4, in every tier and on both POA paths of the engine (matrix and recorded decisions) */
#ifndef CW_POA_MODE
#define CW_POA_MODE CW_POA_MODE_NW
#endif
#define CW_POA_CONSENSUS_MAJORITY 0        /* column-majority vote over the MSA (implemented)        */
#define CW_POA_CONSENSUS_HEAVIEST_BUNDLE 1 /* heaviest bundle (Lee 2002, as in spoa's generate_consensus): implemented on both sides since round 4:
                                              edge weight = sequences whose path uses the edge; nodes in rank order: the in-edge of largest weight
                                              is the node's choice, on equal weights the one whose source has the larger score, on equal scores
                                              the LATER in-edge (spoa's `<=`); score = that weight + the source's score (0 without in-edges); the
                                              consensus ends in the best-scoring SINK (lowest rank on ties: where spoa completes a branch that
                                              stops inside the graph, this rule never starts one) and is read back along the choices */
#ifndef CW_POA_CONSENSUS
#define CW_POA_CONSENSUS CW_POA_CONSENSUS_MAJORITY
#endif
#define CW_CONS_TIE_TEMPLATE 0 /* equally frequent bases in a column -> the template's if among them, else the smallest code */
#define CW_CONS_TIE_SMALLEST 1 /* -> the smallest code (A<C<G<T)                                      */
#ifndef CW_POA_CONS_TIE
#define CW_POA_CONS_TIE CW_CONS_TIE_TEMPLATE
#endif
#define CW_CONS_GAP_STRICT 0   /* a column is dropped iff gaps > every base count                     */
#define CW_CONS_GAP_TIE_DROPS 1 /* ... iff gaps >= the largest base count                             */
#ifndef CW_POA_CONS_GAP
#define CW_POA_CONS_GAP CW_CONS_GAP_STRICT
#endif
#define CW_CHAIN_TIE_SMALLEST_SUCCESSOR 0 /* equal length and score -> the smallest successor index (strict '>' while scanning the successors upward) */
#define CW_CHAIN_TIE_LARGEST_SUCCESSOR 1  /* -> the largest ('>=' on the score: a memoised recursion that keeps the LAST best successor); both sides since round 6 */
#ifndef CW_CHAIN_TIE
#define CW_CHAIN_TIE CW_CHAIN_TIE_SMALLEST_SUCCESSOR
#endif
#define CW_SEG_MISSING_ANCHOR_DROP 0 /* a sequence lacking an anchor of a segment is left out of that segment */
#define CW_SEG_MISSING_ANCHOR_EXTRAPOLATE 1 /* both sides since round 6: an anchor of the chain that a sequence lacks is placed where the TEMPLATE's spacing puts it,
                                               counted from the nearest chain anchor the sequence does hold -- the one before it, else the one after: q_i = p_j + (t_i - t_j),
                                               t = the anchors' template positions -- inside [0, min(length, 65534)]; then the segments are cut as under DROP with q for
                                               p (a piece whose ends come out reversed or equal is left out; a sequence holding no chain anchor stays out of everything) */
#ifndef CW_SEG_MISSING_ANCHOR
#define CW_SEG_MISSING_ANCHOR CW_SEG_MISSING_ANCHOR_DROP
#endif
#if (CW_POA_MODE != CW_POA_MODE_NW && CW_POA_MODE != CW_POA_MODE_OV && CW_POA_MODE != CW_POA_MODE_SW) || (CW_POA_CONSENSUS != CW_POA_CONSENSUS_MAJORITY && CW_POA_CONSENSUS != CW_POA_CONSENSUS_HEAVIEST_BUNDLE) || \
    (CW_CHAIN_TIE != CW_CHAIN_TIE_SMALLEST_SUCCESSOR && CW_CHAIN_TIE != CW_CHAIN_TIE_LARGEST_SUCCESSOR) || (CW_SEG_MISSING_ANCHOR != CW_SEG_MISSING_ANCHOR_DROP && CW_SEG_MISSING_ANCHOR != CW_SEG_MISSING_ANCHOR_EXTRAPOLATE)
#error "cw_policy.h: this value of CW_POA_MODE / CW_POA_CONSENSUS / CW_CHAIN_TIE / CW_SEG_MISSING_ANCHOR is named but not implemented (oracle/cw_oracle.cpp and consent_amd/csrc/ would both have to change)"
#endif
/* the column vote, one place for both sides: drop the column? / take the template's base on a tie? */
#define CW_CONS_DROPS(gaps, top_count) (CW_POA_CONS_GAP == CW_CONS_GAP_STRICT ? (gaps) > (top_count) : (gaps) >= (top_count))
#define CW_CONS_TEMPLATE_WINS_TIES (CW_POA_CONS_TIE == CW_CONS_TIE_TEMPLATE)
#define CW_CONS_HEAVIEST_BUNDLE (CW_POA_CONSENSUS == CW_POA_CONSENSUS_HEAVIEST_BUNDLE)

/* --- anchor index (A4a) --------------------------------------------------------------------- */
/* A k-mer occurring twice inside ANY single sequence of the pile is never an anchor.            */
/* A k-mer must occur in at least `anchor_support` distinct sequences (compared as                */
/* occurrences < support  =>  dropped; support 0 keeps everything).                              */

/* --- chaining (A4b) ------------------------------------------------------------------------- */
/* score(a,b) = number of sequences that contain both anchors with pos(a) < pos(b).              */
/* Sequences holding them in the opposite order are simply not counted (no veto).                */
/* Edge a->b (a before b on the template) is usable iff score(a,b) >= anchor_support.            */
/* best(a) = longest chain starting at a; ties on length -> larger summed score; remaining ties  */
/* -> the SMALLEST successor index (strict '>' while scanning successors upward).                */
/* Chain start: scan template anchors from last to first, strict '>' on length, then strict '>'  */
/* on score => on full ties the LARGEST start index wins.                                        */

/* --- segmentation (A4c) --------------------------------------------------------------------- */
/* Segment 0      : seq[0, pos(c0))               for sequences holding c0.                      */
/* Segment i      : seq[pos(c_{i-1}), pos(c_i))   for sequences holding both, in that order;     */
/*                  a sequence lacking either anchor (or holding them reversed) is left out of   */
/*                  that segment only.                                                           */
/* Segment m      : seq[pos(c_{m-1}), end)        for sequences holding c_{m-1}.                 */
/* Empty strings are skipped.  Members keep pile order (template first); the first `max_msa`     */
/* non-empty members are aligned, the rest ignored.                                              */

/* --- DBG polish constants that ARE pinned by in-tree reference code -------------------------- */
#define CW_DBG_ZONE          3   /* correctionDBG.cpp:102 */
#define CW_DBG_MAX_BRANCHES 50   /* correctionDBG.cpp:100 */
#define CW_DBG_MAX_ANCHORS   5   /* correctionDBG.cpp:144 */

/* --- read re-assembly: alignConsensus uses StripedSmithWaterman::Aligner (Complete-Striped-Smith-Waterman-
 * Library, bundled inside the absent BMEAN submodule; call sites correctionAlignment.cpp:48-52,90,110).  Restated from the
 * library's published algorithm: default Aligner() scores, exact local alignment with affine gaps (a gap of length L
 * costs open + (L-1)*extend), and its position rules:
 *   end   = first reference column whose column maximum exceeds every earlier one; inside it the smallest query index
 *           holding that maximum;
 *   begin = the same search on the reversed prefixes, scanning the reference backwards from the end column and stopping
 *           at the first column whose running maximum reaches the forward score;
 *   cigar = banded traceback between begin and end (band |ref_len - query_len| + 1, doubled until the score is reached),
 *           only its inserted/deleted base totals are used (correctionAlignment.cpp:28-45,111).
 * str2num on the mixed-case overlap strings (correctionAlignment.cpp:9): every character other than 'A','C','G' counts as T.
 * PARITY UNPINNED (library absent).                                                                                     */
/* The gap model of the POA.  LINEAR: a gap of length L costs L * CW_POA_GAP (the default: what the three-score engine call of the spoa that BMEAN
 * bundles takes).  AFFINE (spoa >= 3 offers one; round 5, policy insurance): a gap of length L costs CW_POA_GAP_OPEN + (L - 1) * CW_POA_GAP_EXT, with
 * CW_POA_GAP_OPEN <= CW_POA_GAP_EXT < 0 -- Gotoh's three layers on the graph, global mode only:
 *   F[i][j] = max over the in-edges p of i, in order: max(H[p][j] + open, F[p][j] + ext)           (a gap in the sequence: the node is skipped)
 *   E[i][j] = max(H[i][j-1] + open, E[i][j-1] + ext)                                               (a gap in the graph: the base is an insertion)
 *   H[i][j] = max(max over p (H[p][j-1] + s), F[i][j], E[i][j]);  column 0: H = F, no E;  the start row: H[0][0] = 0, H[0][j] = E[0][j] = open + (j-1) ext, no F
 *   end cell as in the linear model (best node without out-edges in the last column, lowest rank on ties); the walk back keeps a layer:
 *     in H: diagonal through the in-edges in order, else F if H == F, else E;
 *     in F: the first in-edge with F == H[p][j] + open (-> p, layer H), else the first with F == F[p][j] + ext (-> p, layer F);
 *     in E: H[i][j-1] + open (-> layer H) before E[i][j-1] + ext (-> layer E).
 * Implemented on both sides: oracle/cw_oracle.cpp PoaGraph::align, consent_amd/csrc/cw_poa_a.h (every task runs in the global-memory tier:
 * three int32 layers per cell; a build for checking a policy, not for speed).  PARITY UNPINNED like the rest of A4 (spoa absent). */
#define CW_POA_GAP_MODEL_LINEAR 0
#define CW_POA_GAP_MODEL_AFFINE 1
#ifndef CW_POA_GAP_MODEL
#define CW_POA_GAP_MODEL CW_POA_GAP_MODEL_LINEAR
#endif
#ifndef CW_POA_GAP_OPEN
#define CW_POA_GAP_OPEN CW_POA_GAP
#endif
#ifndef CW_POA_GAP_EXT
#define CW_POA_GAP_EXT (-6)
#endif
#define CW_POA_AFFINE (CW_POA_GAP_MODEL == CW_POA_GAP_MODEL_AFFINE)
#if CW_POA_GAP_MODEL != CW_POA_GAP_MODEL_LINEAR && CW_POA_GAP_MODEL != CW_POA_GAP_MODEL_AFFINE
#error "cw_policy.h: CW_POA_GAP_MODEL is LINEAR (0) or AFFINE (1)"
#endif
#if CW_POA_AFFINE && (CW_POA_MODE != CW_POA_MODE_NW || CW_POA_GAP_OPEN > CW_POA_GAP_EXT || CW_POA_GAP_EXT >= 0 || CW_POA_GAP_OPEN < -64)
#error "cw_policy.h: the affine gap model is implemented for the global mode, with -64 <= CW_POA_GAP_OPEN <= CW_POA_GAP_EXT < 0"
#endif

#define CW_SSW_MATCH     2
#define CW_SSW_MISMATCH  2
#define CW_SSW_GAP_OPEN  3
#define CW_SSW_GAP_EXT   1

#endif /* CW_POLICY_H */
