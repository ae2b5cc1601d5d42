/*
 * cw_poa_a.h -- the POA alignment under cw_policy.h's AFFINE gap model (CW_POA_GAP_MODEL_AFFINE; round 5, policy insurance).
 *
 * Gotoh's three layers on the graph, global mode, exactly as the policy block states them and oracle/cw_oracle.cpp restates them:
 * H, F (the node is skipped), E (the base is an insertion), three int32 planes of (n + 1) x (L + 1) cells in the wave's slab -- which is why
 * a build with this policy runs every task in the global-memory tier (poa_run hands a task on from any other tier at once; the chain kernel
 * routes nothing to tiers Q / H).  A build for checking a policy, not for speed: one wave per task, a row at a time, the walk back one step
 * per round trip.
 *
 * fill   lanes over the columns of a row, 64 at a time; F and the diagonal are plain reads of the in-edges' rows (plist order);
 *        E[j] = max over t < j of (H'[t] + open + (j - 1 - t) ext), H' = the cell without its E, is an exclusive prefix max of
 *        H'[t] - t ext over the lanes with a carry between chunks -- the recurrence E[j] = max(H[j-1] + open, E[j-1] + ext) without its
 *        dominated terms (open <= ext: a cell raised by E never opens a better gap than the one that raised it), so the VALUES of all
 *        three planes are the oracle's and the walk back can compare them the way the oracle does.
 * walk   wave-uniform, the layer kept in a register; the rules of the policy block word for word.  Records seqrank[j] = rank aligned
 *        to sequence position j like every other path of poa_run; the merge that follows is shared.
 */
#ifndef CW_POA_A_H
#define CW_POA_A_H

#define CW_AFF_NEG (-(1 << 28))

/* returns 0, or poa_run's codes: 2 = the three planes do not fit this slab, 3 = no move explains a cell (cannot happen) */
template <typename HT>
__device__ __forceinline__ int poa_affine_align(const PoaMem<HT>& M, const int n, const int L, const int lane, int* end_row) {
    static_assert(sizeof(HT) == 4, "the affine model runs in the int32 tier");
    const int GO = CW_POA_GAP_OPEN, GE = CW_POA_GAP_EXT, MS = CW_POA_MATCH, XS = CW_POA_MISMATCH;
    const int cols = L + 1;
    const size_t plane = (size_t)(n + 1) * (size_t)cols;
    if (3 * plane > (size_t)M.h_cap) return 2;
    int* const Hp = (int*)M.H;
    int* const Fp = Hp + plane;
    int* const Ep = Fp + plane;
    /* the start row */
    for (int j = lane; j < cols; j += 64) {
        const int v = j == 0 ? 0 : GO + (j - 1) * GE;
        Hp[j] = v; Fp[j] = CW_AFF_NEG; Ep[j] = j == 0 ? CW_AFF_NEG : v;
    }
    cw_wave_sync();
    for (int r = 0; r < n; ++r) {
        const uint32_t meta = M.rmeta[r];
        const int np = CW_RM_NP(meta), x = CW_RM_X(meta), base = (int)(meta & 3u);
        int* const hrow = Hp + (size_t)(r + 1) * cols;
        int* const frow = Fp + (size_t)(r + 1) * cols;
        int* const erow = Ep + (size_t)(r + 1) * cols;
        int carry = CW_AFF_NEG; /* max of H'[t] - t ext over the chunks before this one */
        for (int c0 = 0; c0 < cols; c0 += 64) {
            const int j = c0 + lane;
            const bool act = j < cols;
            int fv = CW_AFF_NEG, dv = CW_AFF_NEG;
            if (act) {
                const int s = j >= 1 ? ((int)M.sq[j - 1] == base ? MS : XS) : 0;
                for (int p = 0; p < np; ++p) {
                    const int pr = np == 1 ? x : (int)M.plist[x + p];
                    const int* ph = Hp + (size_t)pr * cols;
                    const int* pf = Fp + (size_t)pr * cols;
                    fv = max(fv, max(ph[j] + GO, pf[j] + GE));
                    if (j >= 1) dv = max(dv, ph[j - 1] + s);
                }
            }
            const int hq = j == 0 ? fv : max(dv, fv);
            const unsigned key = act ? (unsigned)(hq - j * GE + 0x40000000) : 0u;
            const unsigned inc = cw_wave_scan_max_u32(key);
            const int before = (int)(unsigned)CW_DPP(0, (int)inc, 0x138, 0xF) - 0x40000000; /* lanes before this one in the chunk (lane 0: far below anything) */
            const int ex = max(before, carry);
            const int ev = j >= 1 ? ex + GO + (j - 1) * GE : CW_AFF_NEG;
            const int hv = max(hq, ev);
            if (act) { hrow[j] = hv; frow[j] = fv; erow[j] = j >= 1 ? max(ev, CW_AFF_NEG) : CW_AFF_NEG; }
            carry = max(carry, (int)(unsigned)cw_lane_value((int)inc, 63) - 0x40000000);
        }
        cw_wave_sync();
    }
    /* end cell: the best node without out-edges in the last column, lowest rank on ties */
    int bs = -2147483647 - 1, br = 0x7FFFFFFF;
    for (int r = lane; r < n; r += 64) {
        if (M.has_out[M.r2n[r]]) continue;
        const int h = Hp[(size_t)(r + 1) * cols + L];
        if (h > bs) { bs = h; br = r; } /* ranks ascend within a lane */
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int os = __shfl_xor(bs, o), orr = __shfl_xor(br, o);
        if (os > bs || (os == bs && orr < br)) { bs = os; br = orr; }
    }
    int i = __builtin_amdgcn_readfirstlane(br) + 1, j = L, layer = 0; /* 0 H, 1 F, 2 E */
    *end_row = i;
    /* the walk back (wave-uniform: every lane reads the same cells) */
    for (int guard = 0; !(i == 0 && j == 0 && layer == 0); ++guard) {
        if (guard > 2 * (n + L) + 8) return 3;
        i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j); layer = __builtin_amdgcn_readfirstlane(layer);
        const size_t at = (size_t)i * cols + j;
        if (layer == 0) {
            const int h = Hp[at];
            bool found = false;
            if (i != 0 && j != 0) {
                const uint32_t meta = M.rmeta[i - 1];
                const int np = CW_RM_NP(meta), x = CW_RM_X(meta);
                const int s = (int)M.sq[j - 1] == (int)(meta & 3u) ? MS : XS;
                for (int p = 0; p < np && !found; ++p) {
                    const int pr = np == 1 ? x : (int)M.plist[x + p];
                    if (h == Hp[(size_t)pr * cols + j - 1] + s) {
                        if (lane == 0) M.seqrank[j - 1] = (uint16_t)(i - 1);
                        i = pr; j = j - 1; found = true;
                    }
                }
            }
            if (!found) {
                if (i != 0 && h == Fp[at]) layer = 1;
                else if (j != 0 && h == Ep[at]) layer = 2;
                else return 3;
            }
        } else if (layer == 1) {
            const int f = Fp[at];
            const uint32_t meta = M.rmeta[i - 1];
            const int np = CW_RM_NP(meta), x = CW_RM_X(meta);
            bool found = false;
            for (int p = 0; p < np && !found; ++p) {
                const int pr = np == 1 ? x : (int)M.plist[x + p];
                if (f == Hp[(size_t)pr * cols + j] + GO) { i = pr; layer = 0; found = true; }
            }
            for (int p = 0; p < np && !found; ++p) {
                const int pr = np == 1 ? x : (int)M.plist[x + p];
                if (f == Fp[(size_t)pr * cols + j] + GE) { i = pr; found = true; }
            }
            if (!found) return 3;
        } else {
            const int e = Ep[at];
            if (e == Hp[at - 1] + GO) layer = 0;
            else if (e != Ep[at - 1] + GE) return 3;
            j = j - 1; /* (an insertion: seqrank[j] stays CW_NONE16) */
        }
    }
    return 0;
}

#endif
