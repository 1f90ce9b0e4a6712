// sto_lean.h — the active-set storage solve of the copper plate, written for instruction count (round 4).
//
// Included by kernels_agents.hip after the helpers it shares with sto_warm_body (box2, scan_sum, the segmented max/min
// scan, the clamp-map scans of the left-to-right release pass, acc_add). Replaces optimize_subproblem(::Storage)
// (reference src/optimization/subproblems.jl:107-207) exactly as sto_warm_body does — same contact-set guess, same
// bracketed Newton, same KKT certificate and repair rule, same hand-over to the scan body, same LDS protocol. First written
// for the case the headline configurations run (no lines, horizon T == LPS * NCH, LPS <= 32), then widened: LPS = 64, horizons
// that end inside the lane group (FULLT = false), and network items whose price tables are empty (LINES). What changed is HOW each step is
// written (profiles/r03_valu.json: 3 177 vector instructions per wave on config4, every one 4 cycles; the static mix
// was 30 % moves and 17 % selects around ~180 exec-mask regions):
//   * segment sums without selects: a step that starts a segment carries keep = 0.0, every other step keep = 1.0, and the
//     running sums are fma(run, keep, x) inside the lane and across lanes; the level a segment starts from is ADDED to the
//     first step's net charge, so the prefix is the level itself (no base array, no second LDS hand-off);
//   * cross-lane scans without lane-validity tests: the first lane of a group always starts a segment, so by the time a
//     lane could read past its group's first lane its keep is 0 — what it reads there (another group's finite sums, or
//     the zeros DPP's bound_ctrl supplies) is multiplied away; DPP reads are `update_dpp(0, x, .., bound_ctrl)`: one move
//     per dword, no copy of the old value in front;
//   * the certificate's two right-to-left chains as GATED max / min chains — flo_i = max(mlo_i, min(g_i, flo_{i+1})) with
//     g_i = +inf behind an empty contact and -inf otherwise (mirrored for fhi) — one double and one gate per chain instead
//     of a (lo, hi) clamp pair each; the horizon is a blocked element, so a group's last lane needs no test either;
//   * the Newton brackets live in registers (sto_warm_body: four LDS round trips per segment end and iteration);
//   * convergence is a ballot, not a max reduction of doubles;
//   * a Newton step that leaves every step of the wave on its piece is exact (the sums are piecewise linear): the second
//     evaluation then only produces (D, C) at the new price, the sums follow by one fma per step, no scan.
// The arithmetic that decides anything (tolerances, bracket rules, the flat-segment jump, the repair rule, both release
// passes) is sto_warm_body's; results agree to rounding (sums are associated differently).
#pragma once
#include <type_traits>

namespace dopf {

// element at a 32-bit BYTE offset from a uniform base: scalar base + vector offset addressing (the same offset serves the
// arrays that share a layout: D, C and the stored prices) instead of a 64-bit address per access
template <class Tp>
__device__ __forceinline__ Tp *lz_at(Tp *base, unsigned byte_off)
{
    return reinterpret_cast<Tp *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<Tp>::type *>(base)) + byte_off);
}

template <int CTRL>
__device__ __forceinline__ double lz_dpp(double x)          // DPP read, lanes without a source get 0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int lz_dppi(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
// DPP read, lanes without a source get `idv` (one more move per dword). ZERO_OK: the caller's lanes without a source
// ignore what they read (groups of at most one row: the header), so the cheaper zero-filling read will do.
template <int CTRL, bool ZERO_OK>
__device__ __forceinline__ double lz_dpp_id(double x, double idv)
{
    if (ZERO_OK) return lz_dpp<CTRL>(x);
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(idv), __double2loint(x), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(idv), __double2hiint(x), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// DPP read under a row mask: rows outside the mask (and lanes without a source) keep `idv`
template <int CTRL, int RM>
__device__ __forceinline__ double lz_dpp_rm(double x, double idv)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(idv), __double2loint(x), CTRL, RM, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(idv), __double2hiint(x), CTRL, RM, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int LPS>
__device__ __forceinline__ double lz_prev(double x)          // previous lane (the group's first lane: unspecified, finite)
{
    if (LPS <= 16) return lz_dpp<0x111>(x);                  // row_shr:1
    return lz_dpp<0x138>(x);                                 // wave_shr:1
}
template <int LPS>
__device__ __forceinline__ double lz_next(double x)          // next lane (the group's last lane: unspecified, finite or as stored)
{
    if (LPS <= 16) return lz_dpp<0x101>(x);                  // row_shl:1
    return lz_dpp<0x130>(x);                                 // wave_shl:1
}
template <int LPS>
__device__ __forceinline__ int lz_previ(int x)
{
    if (LPS <= 16) return lz_dppi<0x111>(x);
    return lz_dppi<0x138>(x);
}

// Inclusive segmented prefix sums of (a, b) over the lanes of each group. k = 1.0 if no segment starts inside the lane,
// else 0.0 (the group's first lane: always 0.0). Needs no lane tests: see the header. (Groups of two rows: the first lanes
// of the second row have no source in the row steps — zeros for the sums are the identity, the flag reads 1.0 there.)
template <int LPS>
__device__ __forceinline__ void lz_seg_sum2(double k, double &a, double &b)
{
#define DOPF_LZ_STEP(CTRL)                                                           \
    {                                                                                \
        const double pa = lz_dpp<CTRL>(a), pb = lz_dpp<CTRL>(b), pk = lz_dpp_id<CTRL, (LPS <= 16)>(k, 1.0); \
        a = fma(pa, k, a); b = fma(pb, k, b); k *= pk;                                \
    }
    DOPF_LZ_STEP(0x111)
    if (LPS >= 4) DOPF_LZ_STEP(0x112)
    if (LPS >= 8) DOPF_LZ_STEP(0x114)
    if (LPS >= 16) DOPF_LZ_STEP(0x118)
    if (LPS == 32) DOPF_LZ_STEP(0x142)                       // row_bcast:15 (rows 0 and 2 start groups: their keep is 0 by now)
#undef DOPF_LZ_STEP
    if (LPS >= 64) {
        // one group per wave: rows 1 and 3 take the row before them (its last lane), then rows 2 and 3 take lane 31 — the rows
        // that are not meant read the identity (row masks; their old value is what they keep)
#define DOPF_LZ_STEP64(CTRL, RM)                                                     \
        {                                                                            \
            const double pa = lz_dpp_rm<CTRL, RM>(a, 0.0), pb = lz_dpp_rm<CTRL, RM>(b, 0.0), pk = lz_dpp_rm<CTRL, RM>(k, 1.0); \
            a = fma(pa, k, a); b = fma(pb, k, b); k *= pk;                            \
        }
        DOPF_LZ_STEP64(0x142, 0xA)
        DOPF_LZ_STEP64(0x143, 0xC)
#undef DOPF_LZ_STEP64
    }
}

// Suffix scans of the gated chains: lane composite (M, G) stands for x -> max(M, min(G, x)) (lower ends) resp.
// x -> min(M, max(G, x)) (upper ends), G in {+inf, -inf} ("passes what comes from the right" / "does not"). The group's
// last lane is blocked (the horizon), so nothing is read past it.
template <int LPS, bool LOWER>
__device__ __forceinline__ void lz_chain_scan(double &M, double &G, int lane)
{
#define DOPF_LZ_STEP(CTRL)                                                           \
    {                                                                                \
        const double pM = lz_dpp_id<CTRL, (LPS <= 16)>(M, LOWER ? -INFINITY : INFINITY);   \
        const double pG = lz_dpp_id<CTRL, (LPS <= 16)>(G, LOWER ? INFINITY : -INFINITY);   \
        if (LOWER) { M = lz_max(M, lz_min(G, pM)); G = lz_min(G, pG); }               \
        else { M = lz_min(M, lz_max(G, pM)); G = lz_max(G, pG); }                     \
    }
    // (groups of at most one row: a lane whose source is outside the row has taken in the group's blocked last lane by then and
    // ignores the zeros it reads; groups of two rows: the last lanes of the first row read the identity instead)
    DOPF_LZ_STEP(0x101)                                      // row_shl:1
    if (LPS >= 4) DOPF_LZ_STEP(0x102)
    if (LPS >= 8) DOPF_LZ_STEP(0x104)
    if (LPS >= 16) DOPF_LZ_STEP(0x108)
#undef DOPF_LZ_STEP
    if (LPS >= 32) {                                         // rows 0, 2 take the whole of the next row (its first lane)
        const double pM = __shfl(M, (lane | 15) + 1), pG = __shfl(G, (lane | 15) + 1);
        if ((lane & 16) == 0) {
            if (LOWER) { M = lz_max(M, lz_min(G, pM)); G = lz_min(G, pG); }
            else { M = lz_min(M, lz_max(G, pM)); G = lz_max(G, pG); }
        }
    }
    if (LPS >= 64) {                                         // rows 0, 1 take rows 2-3 (lane 32)
        const double pM = __shfl(M, 32), pG = __shfl(G, 32);
        if ((lane & 32) == 0) {
            if (LOWER) { M = lz_max(M, lz_min(G, pM)); G = lz_min(G, pG); }
            else { M = lz_min(M, lz_max(G, pM)); G = lz_max(G, pG); }
        }
    }
}

// inclusive segmented prefix (max of a, min of b) over the lanes of each group; K = +inf if no segment starts inside the
// lane, else -inf (the group's first lane: always -inf)
template <int LPS>
__device__ __forceinline__ void lz_seg_maxmin(double K, double &a, double &b)
{
#define DOPF_LZ_STEP(CTRL)                                                           \
    {                                                                                \
        const double pa = lz_dpp_id<CTRL, (LPS <= 16)>(a, -INFINITY), pb = lz_dpp_id<CTRL, (LPS <= 16)>(b, INFINITY); \
        const double pK = lz_dpp_id<CTRL, (LPS <= 16)>(K, INFINITY);                   \
        a = lz_max(a, lz_min(K, pa)); b = lz_min(b, lz_max(-K, pb)); K = lz_min(K, pK); \
    }
    DOPF_LZ_STEP(0x111)
    if (LPS >= 4) DOPF_LZ_STEP(0x112)
    if (LPS >= 8) DOPF_LZ_STEP(0x114)
    if (LPS >= 16) DOPF_LZ_STEP(0x118)
    if (LPS == 32) DOPF_LZ_STEP(0x142)
#undef DOPF_LZ_STEP
    if (LPS >= 64) {
#define DOPF_LZ_STEP64(CTRL, RM)                                                     \
        {                                                                            \
            const double pa = lz_dpp_rm<CTRL, RM>(a, -INFINITY), pb = lz_dpp_rm<CTRL, RM>(b, INFINITY), pK = lz_dpp_rm<CTRL, RM>(K, INFINITY); \
            a = lz_max(a, lz_min(K, pa)); b = lz_min(b, lz_max(-K, pb)); K = lz_min(K, pK); \
        }
        DOPF_LZ_STEP64(0x142, 0xA)
        DOPF_LZ_STEP64(0x143, 0xC)
#undef DOPF_LZ_STEP64
    }
}

// persistent iterations (agents_persist.h): Status::pseq = number of dual updates published inside launches (low 30 bits) | the halt
// word of the last of them (bit 31). One lane waits until the count has reached `want`; returns the word, or all ones on a time-out.
constexpr unsigned kPersistHaltBit = 0x80000000u, kPersistSeqMask = 0x3fffffffu;
__device__ __forceinline__ unsigned persist_wait_word(Status *st, const unsigned want)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned round = 1;; ++round) {
        const unsigned w = (unsigned)p_ldi(reinterpret_cast<const int *>(&st->pseq));
        if (((w - want) & kPersistSeqMask) < 0x20000000u) return w;          // (count >= want, wrap-safe)
        if ((round & 255u) == 0u && wall_clock64() - t0 > 200000000ull) return 0xffffffffu;      // 2 s of the 100 MHz wall clock
        __builtin_amdgcn_s_sleep(1);
    }
}

// returns the number of storages of the item left to the scan body (block-uniform); -1 = halted
// PERSIST (agents_persist.h: several iterations in one launch): the prices another block of this launch has just published are
// read past the caches, and the accumulator set is the caller's (`ppar`), not the status block's word.
// (PERSIST, pwant != 0: the wait for the previous iteration's prices happens INSIDE, behind the first pass's row loads — those
// do not depend on the prices; returns -1 when the wait found the halted state or timed out: nothing stored)
// LINES: an item of storages at a node of a network whose tables are EMPTY for every timestep (no kink of Psi inside the node's
// window: the settled state) — the copper plate's closed form with (Psi(0), slope) of the (node, timestep) in place of (theta, gamma);
// an item that meets a non-empty table is handed to the general body (sto_warm_body) as a whole.
// FULLT = false: the horizon ends inside the lane group (T < LPS * NCH): the steps behind it are inert (no charge, no slope, no
// contact) and sit in a segment of their own behind step T - 1, which always ends one.
// (the general body as a function of its own: inlined behind the table test, its registers crowd the lean body's)
template <int LPS, int NCH>
__device__ __attribute__((noinline)) int sto_warm_lines_call(const DevView *self, const int blk, const int halt)
{
    return sto_warm_body<LPS, NCH, true>(*self, blk, halt);
}

template <int LPS, int NCH, bool TAIL, bool PERSIST = false, bool LINES = false, bool FULLT = true>
__device__ __forceinline__ int sto_lean_body(const DevView &v, const int blk, const int halt, const int ppar = 0, const unsigned pwant = 0u)
{
    static_assert(LPS <= 64 && NCH <= 8, "4 pattern bits per step");
    static_assert(!(PERSIST && (LINES || !FULLT)) && !(TAIL && LINES), "persistent iterations / the tail in the launch: full-horizon copper plates");
    constexpr int NG = 256 / LPS, TP = LPS * NCH;
    const int T = FULLT ? TP : v.T;
    constexpr int MAXR = 16;                 // contact-set rounds per storage
    constexpr int MAXN = 40;                 // Newton iterations per round
    constexpr int BIG = 0x3fffffff;
    __shared__ double red[NG * TP];          // nuL: price of the segment that ends here
    __shared__ double fdL[NG * TP];          // flat segment: signed distance to the nearest kink ahead; release flags of the left-to-right pass
    const int tid = threadIdx.x, lane = tid & 63, li = tid & (LPS - 1), grp = tid / LPS;
    const int gbase = lane & ~(LPS - 1);
    Item it;
    if (v.stoChunk > 0) { it.a0 = blk * v.stoChunk; it.a1 = min(v.S, it.a0 + v.stoChunk); it.node = 0; }
    else it = v.sto_items[blk];
    const int N = v.N;
    const double w = v.w_prox, gam = v.gamma;
    const double a0 = w + gam, ia0 = v.cp_ia, idet0 = v.cp_idet, s20 = v.cp_s2;      // (host: the same expressions)
    const int tbase = li * NCH;
    double *nuL = red + grp * TP, *fd_ = fdL + grp * TP;
    const bool first = li == 0, last = li == LPS - 1;
    // step c of this lane: inside the horizon / the horizon's last step
#define LZ_IN(c) (FULLT || tbase + (c) < T)
#define LZ_TLAST(c) (FULLT ? (last && (c) == NCH - 1) : (tbase + (c) == T - 1))
    // box2 coefficients of step c: a = w + kappa, b = kappa with kappa = gamma on the copper plate, the slope of the step's Psi with lines
    double lp0[NCH], kapL[NCH], iaL[NCH], idetL[NCH], s2L[NCH];
#define LZ_KAP(c) (LINES ? kapL[c] : gam)
#define LZ_AA(c) (LINES ? w + kapL[c] : a0)
#define LZ_IA(c) (LINES ? iaL[c] : ia0)
#define LZ_IDET(c) (LINES ? idetL[c] : idet0)
#define LZ_S2(c) (LINES ? s2L[c] : s20)
    if (LINES) {
        int nonlin = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            lp0[c] = 0.0; kapL[c] = gam; iaL[c] = ia0; idetL[c] = idet0; s2L[c] = s20;
            if (LZ_IN(c)) {
                const size_t at = (size_t)it.node + (size_t)N * (tbase + c);
                const int m_ = v.tb_m[at];                   // (all three loaded at once: sto_warm_body)
                const double p0_ = v.tb_psi0[at], k_ = v.tb_slope[at * (v.M2 + 1)];
                nonlin |= m_ != 0;
                lp0[c] = p0_; kapL[c] = k_;
                lin_coef(w, 1.0 / w, k_, iaL[c], idetL[c], s2L[c]);
            }
        }
        // (block-uniform: the lanes of one group see every timestep of the item's node)
        if (__syncthreads_or(nonlin)) return sto_warm_lines_call<LPS, NCH>(v.self, blk, halt);
    }

    double accQ[NCH];
    double accCost = 0.0;
    int anyFail = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) accQ[c] = 0.0;
    // price + gamma * imbalance of the lane's steps: the same for every storage pass of the block. Parked in LDS (one read per
    // step and pass) instead of registers (the kernel sits at its register limit) or two more loads per step and pass, whose
    // addresses cost a v_readlane each once the solve has taken the scalar registers.
    __shared__ double th0L[TP];
    __shared__ int goL;
    if (!PERSIST && !LINES) {
        if (tid < LPS) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) th0L[tbase + c] = LZ_IN(c) ? v.price[it.node + N * (tbase + c)] + gam * v.s[tbase + c] : 0.0;
        }
        __syncthreads();
    }
#ifdef DOPF_STATS
    unsigned long long st_rounds = 0, st_newton = 0, st_short = 0;
#endif
    const int nRep = (it.a1 - it.a0 + NG - 1) / NG;
    for (int rep = 0; rep < nRep; ++rep) {
        const int s = it.a0 + rep * NG + grp;
        const bool live = s < it.a1;
        // The arrays' addresses are read from the view's copy in device memory where they are needed (scalar loads, issued
        // with the rows' loads) instead of being held in scalar registers across the solve: the solve keeps ~20 lane masks in
        // scalar register pairs, and every address held beside them was a v_writelane / v_readlane pair per pass.
#ifdef LZ_PV_LOAD
        const DevView *pv = v.self;
        asm volatile("" : "+s"(pv));
#else
        const DevView *pv = &v;
#endif
        // A lane group without a storage of its own (the item's last pass) solves the item's FIRST storage again and stores nothing
        // (gdone from the start): every load below is unconditional, no exec-mask regions around them. Element offsets are 32-bit
        // (dopf_create offers this body only while S * T * 8 < 4 GB): base register + offset addressing, no 64-bit address arithmetic.
        const unsigned sl = live ? (unsigned)s : (unsigned)it.a0;
        const double mc = pv->sto_mc[sl], pm = pv->sto_pmax[sl], em = pv->sto_emax[sl];
        const bool havenu = live && pv->nu_valid[sl] != 0;
        double A0[NCH], B0[NCH], nuv[NCH], dq[NCH];      // rD = A0 - nu, rC = B0 + nu
        double run = 0.0;
        double d0r[NCH], c0r[NCH], nur[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            const unsigned eb = (sl * (unsigned)T + (unsigned)(LZ_IN(c) ? t : 0)) * 8u;
            d0r[c] = *lz_at(pv->D, eb); c0r[c] = *lz_at(pv->C, eb);
            nur[c] = *lz_at(pv->nu_prev, eb);
            if (!LZ_IN(c)) { d0r[c] = 0.0; c0r[c] = 0.0; nur[c] = 0.0; }
        }
        if (PERSIST && rep == 0) {
            // the rows above are on their way; now the prices the launch's tail block publishes (agents_persist.h)
            if (tid == 0) {
                int go = 1;
                if (pwant != 0u) {
                    const unsigned w = persist_wait_word(v.st, pwant);
                    go = w == 0xffffffffu ? -1 : ((w & kPersistHaltBit) ? 0 : 1);
                }
                goL = go;
            }
            __syncthreads();
            if (goL <= 0) {
                if (goL < 0 && tid == 0) v.st->tail_timeout = 1;
                return -1;
            }
            if (tid < LPS) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) th0L[tbase + c] = p_ld(v.price + it.node + N * (tbase + c)) + gam * p_ld(v.s + tbase + c);
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            const double d0 = d0r[c], c0 = c0r[c], nu_st = nur[c];
            // copper plate: theta = price + gamma (imbalance - own injection); lines: Psi(0) of the node's linear piece - slope * own injection
            const double theta = LINES ? lp0[c] - kapL[c] * (d0 - c0) : th0L[t] - gam * (d0 - c0);
            dq[c] = c0 - d0;
            run += c0 - d0;
            A0[c] = w * d0 - mc - theta; B0[c] = w * c0 - mc + theta;
            // (copper plate: the warm start follows the price move — nu + theta is what is stored; lines: nu itself, as sto_warm_body)
            nuv[c] = havenu ? (LINES ? nu_st : nu_st - theta) : 0.0;
            if (!LZ_IN(c)) { A0[c] = -1e300; B0[c] = -1e300; nuv[c] = 0.0; }     // behind the horizon: D = C = 0 at every price, no slope
        }
        if (rep == 0 && halt) return -1;         // (uniform; the loads above are on their way, nothing has been stored)

        // previous level trajectory -> contacts
        const double tolc = 1e-9 * (1.0 + em), tolE = 1e-11 * (1.0 + em), tolr = 1e-12 * (1.0 + em);
        int kind[NCH];                       // 0 free, 1 empty, 2 full
        {
            double inclE = run, zero_ = 0.0;
            lz_seg_sum2<LPS>(first ? 0.0 : 1.0, inclE, zero_);          // (plain prefix sums: one segment per group)
            // (a DPP read is made by ALL lanes, then selected: inside a conditional its source lanes may be switched off, and a
            // switched-off source reads as 0)
            const double prevE = lz_prev<LPS>(inclE);
            double eo = first ? 0.0 : prevE;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                eo += dq[c];
                kind[c] = !LZ_IN(c) ? 0 : (eo <= tolc ? 1 : (eo >= em - tolc ? 2 : 0));
            }
        }

        bool gdone = !live, good = false;
        double Dv[NCH], Cv[NCH], nuc[NCH];
        for (int round = 0; round < MAXR; ++round) {
#ifdef DOPF_STATS
            if (first && !gdone) ++st_rounds;
#endif
            // ---- A. segments of the current contact set ------------------------------------------------------
            bool en[NCH];                        // the step ends a segment: a contact, or the last step
            double keep[NCH], pk[NCH], sb[NCH];
#define LZ_TGT(c) (kind[c] == 2 ? em : 0.0)
            int send[NCH];
            bool st0;
            {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    en[c] = kind[c] != 0 || LZ_TLAST(c);
                }
                // what the previous lane's last step ends: 0 nothing, 1 a segment at level 0 (an empty contact, or — ragged horizons — the
                // open last segment), 2 a segment at level emax. (Full horizons: that step is never step T - 1, so its kind says it all.)
                const int pkd = lz_previ<LPS>(FULLT ? kind[NCH - 1] : (en[NCH - 1] ? (kind[NCH - 1] == 2 ? 2 : 1) : 0));
                st0 = first || pkd != 0;
                double run_k = 1.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bool stc = c == 0 ? st0 : en[c > 0 ? c - 1 : 0];
                    keep[c] = stc ? 0.0 : 1.0;
                    run_k *= keep[c];
                    pk[c] = run_k;
                    // the level the segment starts from rides on its first step's net charge
                    sb[c] = !stc ? 0.0 : (c == 0 ? ((!first && pkd == 2) ? em : 0.0) : LZ_TGT(c > 0 ? c - 1 : 0));
                }
                int mfirst = BIG;
#pragma unroll
                for (int c = NCH - 1; c >= 0; --c)
                    if (en[c]) mfirst = tbase + c;
                const int incl = scan_min_rev_i<LPS>(mfirst, lane);
                const int nxt = next_lane_i<LPS>(incl);
                int carry = last ? BIG : nxt;                // first segment end in the lanes to the right
#pragma unroll
                for (int c = NCH - 1; c >= 0; --c) {
                    if (en[c]) carry = tbase + c;
                    send[c] = carry < TP ? carry : TP - 1;   // (steps behind the horizon, groups that are done: any valid slot)
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (en[c]) nuL[tbase + c] = kind[c] != 0 ? nuv[c] : 0.0;      // one price per segment; open last segment: 0
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < NCH; ++c) nuv[c] = LZ_IN(c) ? nuL[send[c]] : 0.0;
                __builtin_amdgcn_wave_barrier();
            }

            // ---- B. segmented Newton, bracketed --------------------------------------------------------------
            double px[NCH], ps[NCH];
            double blo[NCH], bhi[NCH];               // brackets of the segment prices, kept at the segment ends
#pragma unroll
            for (int c = 0; c < NCH; ++c) { blo[c] = -INFINITY; bhi[c] = INFINITY; }
            bool nconv = false, nfail = false;
            unsigned pat = 0u;                       // active set of every step of the lane at the last evaluation (4 bits per step)
            bool stepped = false;                    // the last iteration moved prices by plain Newton steps only (wave-uniform)
            for (int itn = 0; itn < MAXN; ++itn) {
#ifdef DOPF_STATS
                if (first && !gdone) ++st_newton;
#endif
                unsigned npat = 0u;
                double sl[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    double dd, cc, s1;
                    box2(LZ_AA(c), LZ_KAP(c), LZ_IA(c), LZ_IDET(c), LZ_S2(c), A0[c] - nuv[c], B0[c] + nuv[c], pm, dd, cc, s1);
                    Dv[c] = dd; Cv[c] = cc; sl[c] = s1;          // (of the last evaluation: the certified values when the round passes)
                    // the step's active set: each of D, C at 0 / inside / at pm (a step that jumps from one bound to the other
                    // has changed its piece although "inside or not" reads the same)
                    const unsigned cD = dd <= 0.0 ? 0u : (dd >= pm ? 2u : 1u), cC = cc <= 0.0 ? 0u : (cc >= pm ? 2u : 1u);
                    npat |= (cD | (cC << 2)) << (4 * c);
                }
                // A plain Newton step that left every step of the wave on its piece: the sums moved by slope x step exactly
#ifdef LZ_NO_SHORTCUT
                const bool same = false;
#else
                const bool same = stepped && __all(npat == pat || gdone);
#endif
                pat = npat;
                if (same) {
                    // px already holds the prediction (made where the prices were re-read). Rounding only: the step was exact;
                    // a residual that still shows is served by a full iteration.
                    bool q = false;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) q = q || (en[c] && kind[c] != 0 && fabs(px[c] - LZ_TGT(c)) > tolr);
                    if (!__any(q && !gdone)) {
#ifdef DOPF_STATS
                        if (first && !gdone) ++st_short;
#endif
                        nconv = !nfail;
                        break;
                    }
                }
                {
                    double rx = 0.0, rs = 0.0;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        rx = fma(rx, keep[c], (Cv[c] - Dv[c]) + sb[c]);
                        rs = fma(rs, keep[c], sl[c]);
                        px[c] = rx; ps[c] = rs;
                    }
                    double ax = rx, as = rs;
                    lz_seg_sum2<LPS>(pk[NCH - 1], ax, as);
                    const double cx = lz_prev<LPS>(ax), cs = lz_prev<LPS>(as);      // (first lane: pk = 0 everywhere)
#pragma unroll
                    for (int c = 0; c < NCH; ++c) { px[c] = fma(cx, pk[c], px[c]); ps[c] = fma(cs, pk[c], ps[c]); }
                }
                // residuals at the contacts
                double res[NCH];
                bool need[NCH], flatNeed = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    res[c] = px[c] - LZ_TGT(c);
                    need[c] = en[c] && kind[c] != 0 && fabs(res[c]) > tolr;
                    flatNeed = flatNeed || (need[c] && !(ps[c] > 0.0));
                }
                // does a segment that has to move sit on a flat piece? Then every step of it has D and C on bounds, its
                // net injection does not move with nu and the four prices at which D or C would leave a bound are closed
                // form: signed distance to the nearest one in the direction the residual asks for
                if (__any(flatNeed && !gdone)) {
                    double ru = -INFINITY, rd = INFINITY;        // (-min distance above, min distance below) so far
                    double fu[NCH], fn[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        double du = INFINITY, dn = INFINITY;
                        const double sh = LZ_KAP(c) * (Dv[c] - Cv[c]);
                        const double bD = A0[c] - sh, bC = -B0[c] - sh, wp = w * pm;
                        const double cand[4] = {bD, bD - wp, bC, bC + wp};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double d = cand[q] - nuv[c];
                            if (d > 0.0) du = lz_min(du, d);
                            if (d < 0.0) dn = lz_min(dn, -d);
                        }
                        if (keep[c] == 0.0) { ru = -INFINITY; rd = INFINITY; }
                        ru = lz_max(ru, -du); rd = lz_min(rd, dn);
                        fu[c] = ru; fn[c] = rd;
                    }
                    double au = ru, ad = rd;
                    lz_seg_maxmin<LPS>(pk[NCH - 1] != 0.0 ? INFINITY : -INFINITY, au, ad);
                    const double cu = lz_prev<LPS>(au), cd = lz_prev<LPS>(ad);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (pk[c] != 0.0) { fu[c] = lz_max(fu[c], cu); fn[c] = lz_min(fn[c], cd); }
                        if (en[c] && kind[c] != 0) fd_[tbase + c] = res[c] < 0.0 ? -fu[c] : -fn[c];
                    }
                }
                bool plain = true;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (need[c]) {
                        const int t = tbase + c;
                        const double r = res[c], nu = nuv[c];
                        if (r < 0.0) blo[c] = nu; else bhi[c] = nu;
                        const bool both = blo[c] > -INFINITY && bhi[c] < INFINITY;
                        double trial;
                        bool isPlain = true;
                        if (ps[c] > 0.0) {
                            trial = nu - r * rcp64(ps[c]);
                        } else {
                            const double sd = fd_[t];                  // signed; +-inf when no kink lies ahead
                            trial = nu + sd + copysign(1e-9 * (1.0 + fabs(nu) + fabs(sd)), sd);
                            isPlain = false;
                        }
                        if (!(trial > blo[c] && trial < bhi[c]) || (both && itn >= 8 && (itn & 1))) {
                            if (both) { trial = 0.5 * (blo[c] + bhi[c]); isPlain = false; }
                        }
                        if (trial > blo[c] && trial < bhi[c]) { nuL[t] = trial; plain = plain && isPlain; }
                        else if (both) need[c] = false;        // bracket is two adjacent doubles: this is the root
                        else nfail = true;                     // nothing ahead moves this segment: not a valid contact set
                    }
                }
                bool bad = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) bad = bad || need[c];
                nfail = group_bits<LPS>(nfail, gbase) != 0ull;
                nconv = group_bits<LPS>(bad, gbase) == 0ull && !nfail;
                // every group in the wave runs the same number of rounds (DPP scans need all lanes)
                if (__all(gdone || nconv || nfail)) break;
                stepped = __all(plain || gdone || nconv || nfail) != 0;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const double nn = LZ_IN(c) ? nuL[send[c]] : 0.0;
                    px[c] = fma(ps[c], nn - nuv[c], px[c]);          // where the step lands if every step of the wave stays on its piece
                    nuv[c] = nn;
                }
                __builtin_amdgcn_wave_barrier();
            }

            // ---- C. certificate: levels inside the band, price jumps have the right sign ----------------------
            // (the reasoning: sto_warm_body, section C)
            bool okk = true;
            int nkind[NCH];
            double mlo[NCH], mhi[NCH];
            {
                // corner intervals of the steps, intersected over each segment (inclusive prefix: complete at the segment's end)
                double rl = -INFINITY, rh = INFINITY;
                double slo[NCH], shi[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const double dd = Dv[c], cc = Cv[c];
                    // D = 0: nu >= A0 + gam C; D = pm: nu <= that - a pm; C = 0: nu <= -B0 - gam D; C = pm: nu >= that + a pm
                    const double lD = fma(LZ_KAP(c), cc, A0[c]), hC = -fma(LZ_KAP(c), dd, B0[c]), apm = LZ_AA(c) * pm;
                    double lo = dd <= 0.0 ? lD : -INFINITY;
                    double hi = (dd > 0.0 && dd >= pm) ? lD - apm : INFINITY;
                    hi = lz_min(hi, cc <= 0.0 ? hC : INFINITY);
                    lo = lz_max(lo, (cc > 0.0 && cc >= pm) ? hC + apm : -INFINITY);
                    const double K = keep[c] != 0.0 ? INFINITY : -INFINITY;
                    rl = lz_max(lo, lz_min(K, rl)); rh = lz_min(hi, lz_max(-K, rh));
                    slo[c] = rl; shi[c] = rh;
                }
                double al = rl, ah = rh;
                lz_seg_maxmin<LPS>(pk[NCH - 1] != 0.0 ? INFINITY : -INFINITY, al, ah);
                const double cl = lz_prev<LPS>(al), ch = lz_prev<LPS>(ah);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    nkind[c] = kind[c];
                    const bool contact = en[c] && kind[c] != 0;
                    if (!contact) {                              // (a contact's level is its Newton target)
                        if (px[c] < -tolE) { okk = false; nkind[c] = 1; }
                        else if (px[c] > em + tolE) { okk = false; nkind[c] = 2; }
                    }
                    const double Kp = pk[c] != 0.0 ? INFINITY : -INFINITY;
                    const double sl_ = lz_max(slo[c], lz_min(Kp, cl)), sh_ = lz_min(shi[c], lz_max(-Kp, ch));
                    const double pt = kind[c] != 0 ? nuv[c] : 0.0;
                    // flat segment (zero slope at its end = every step on a corner): the whole interval
                    const bool flat = contact && ps[c] == 0.0 && sl_ <= sh_ && nuv[c] >= sl_ - 1e-9 && nuv[c] <= sh_ + 1e-9;
                    mlo[c] = flat ? sl_ : pt; mhi[c] = flat ? sh_ : pt;
                }
            }
            // Right to left: flo_i = empty ? max(mlo_i, flo_{i+1}) : mlo_i, fhi_i = full ? min(mhi_i, fhi_{i+1}) : mhi_i; past the
            // horizon the price is 0. Steps that end no segment pass everything.
            {
                double Ml = -INFINITY, Gl = INFINITY, Mh = INFINITY, Gh = -INFINITY;      // this lane's composites
                // the horizon: a blocked element with value 0 behind the group's last step
                if (last) { Ml = 0.0; Gl = -INFINITY; Mh = 0.0; Gh = INFINITY; }
#define LZ_EL(c) (en[c] ? mlo[c] : -INFINITY)
#define LZ_GL(c) ((!en[c] || kind[c] == 1) ? INFINITY : -INFINITY)
#define LZ_EH(c) (en[c] ? mhi[c] : INFINITY)
#define LZ_GH(c) ((!en[c] || kind[c] == 2) ? -INFINITY : INFINITY)
#pragma unroll
                for (int c = NCH - 1; c >= 0; --c) {
                    Ml = lz_max(LZ_EL(c), lz_min(LZ_GL(c), Ml)); Gl = lz_min(LZ_GL(c), Gl);
                    Mh = lz_min(LZ_EH(c), lz_max(LZ_GH(c), Mh)); Gh = lz_max(LZ_GH(c), Gh);
                }
                lz_chain_scan<LPS, true>(Ml, Gl, lane);
                lz_chain_scan<LPS, false>(Mh, Gh, lane);
                // what arrives from the right of this lane (the suffix is blocked at the horizon: its value is M)
                const double nextMl = lz_next<LPS>(Ml), nextMh = lz_next<LPS>(Mh);       // (read by all lanes, then selected)
                double flo = last ? 0.0 : nextMl, fhi = last ? 0.0 : nextMh;
#pragma unroll
                for (int c = NCH - 1; c >= 0; --c) {
                    flo = lz_max(LZ_EL(c), lz_min(LZ_GL(c), flo));
                    fhi = lz_min(LZ_EH(c), lz_max(LZ_GH(c), fhi));
                    nuc[c] = nuv[c];
                    if (en[c] && kind[c] != 0) {
                        const double tn = 1e-10 * (1.0 + lz_minabs(flo, fhi));
                        if (flo > fhi + tn) { okk = false; nkind[c] = 0; }      // wrong sign: release the contact
                        nuc[c] = lz_min(lz_max(nuv[c], flo), lz_max(flo, fhi));
                    }
                }
#undef LZ_EL
#undef LZ_GL
#undef LZ_EH
#undef LZ_GH
            }
            // The same question from the LEFT, asked only when the right-to-left pass released a FEW contacts (1..4 of the
            // storage): sto_warm_body, section C.
            {
                int nrel = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) nrel += __popcll(group_bits<LPS>(nkind[c] == 0 && kind[c] != 0, gbase));
                const bool fwd = nrel >= 1 && nrel <= 4 && !gdone;          // (uniform over the lane group)
                if (__any(fwd)) {
                    double Llo = -INFINITY, Lhi = INFINITY, Ulo = -INFINITY, Uhi = INFINITY;     // this lane's composed maps
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (en[c]) {
                            const double l_lo = kind[c] == 2 ? mlo[c] : -INFINITY, l_hi = kind[c] == 2 ? INFINITY : -INFINITY;
                            const double u_lo = kind[c] == 1 ? -INFINITY : INFINITY, u_hi = kind[c] == 1 ? mhi[c] : INFINITY;
                            const double a1 = clampd(Llo, l_lo, l_hi), a2 = clampd(Lhi, l_lo, l_hi);
                            const double b1 = clampd(Ulo, u_lo, u_hi), b2 = clampd(Uhi, u_lo, u_hi);
                            Llo = a1; Lhi = a2; Ulo = b1; Uhi = b2;
                        }
                    }
                    scan_clamps_fwd<LPS>(Llo, Lhi, lane);
                    scan_clamps_fwd<LPS>(Ulo, Uhi, lane);
                    const double leftL = prev_lane<LPS>(Llo), leftU = prev_lane<LPS>(Uhi);
                    double pin = first ? -INFINITY : leftL, phin = first ? INFINITY : leftU;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (en[c]) {
                            const double glo = fmax(mlo[c], pin), ghi = fmin(mhi[c], phin);
                            const double tn = 1e-10 * (1.0 + fmin(fabs(glo), fabs(ghi)));
                            fd_[tbase + c] = glo > ghi + tn ? 1.0 : 0.0;
                            pin = kind[c] == 2 ? glo : -INFINITY;
                            phin = kind[c] == 1 ? ghi : INFINITY;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int sendNext = next_lane_i<LPS>(send[0]);        // the segment the next lane's first step belongs to
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (!LZ_TLAST(c) && en[c] && kind[c] != 0) {
                            const int sn = c + 1 < NCH ? send[c + 1 < NCH ? c + 1 : c] : sendNext;
                            if (fwd && fd_[sn] != 0.0) { okk = false; nkind[c] = 0; }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
#ifdef LZ_DEBUG_PRINT
            if (s == LZ_DEBUG_PRINT && v.st->iters_total == LZ_DEBUG_ITER) {
                for (int c = 0; c < NCH; ++c)
                    printf("r%d t%2d kind %d->%d en %d nuv %.6f D %.5f C %.5f px %.6f ps %.4f mlo %.6f mhi %.6f nuc %.6f okk %d nconv %d gdone %d A0 %.5f B0 %.5f\n", round, tbase + c, kind[c], nkind[c], (int)en[c], nuv[c], Dv[c], Cv[c], px[c], ps[c], mlo[c], mhi[c], nuc[c], (int)okk, (int)nconv, (int)gdone, A0[c], B0[c]);
            }
#endif
            const bool cert = nconv && group_bits<LPS>(!okk, gbase) == 0ull && !(v.debugLeave && s % 3 == 0);
            bool chg = false;
#pragma unroll
            for (int c = 0; c < NCH; ++c) chg = chg || nkind[c] != kind[c];
            const bool changed = group_bits<LPS>(chg, gbase) != 0ull;

            // ---- D. accept, repair the contact set, or give up --------------------------------------------------
            if (!gdone && cert) {
                int s_ = s;
                asm volatile("" : "+v"(s_));
#ifdef LZ_PV_STORE
                const DevView *pw = v.self;
                asm volatile("" : "+s"(pw));
#else
                const DevView *pw = &v;
#endif
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (!LZ_IN(c)) continue;
                    const unsigned eb = ((unsigned)s_ * (unsigned)T + (unsigned)(tbase + c)) * 8u;
                    *lz_at(pw->D, eb) = Dv[c];
                    *lz_at(pw->C, eb) = Cv[c];
                    // copper plate: nu + theta (theta back from the step's offsets); lines: nu
                    *lz_at(pw->nu_prev, eb) = LINES ? nuc[c] : nuc[c] + 0.5 * (B0[c] - A0[c] - w * dq[c]);
                    if (LINES && (pw->keepDeltas || pw->walk_any[tbase + c])) *lz_at(pw->dltS, eb) = (Dv[c] - Cv[c]) + dq[c];
                    accQ[c] += Dv[c] - Cv[c];
                    accCost += mc * (Dv[c] + Cv[c]);
                }
                good = true;
                gdone = true;
            }
            if (!gdone && (!nconv || !changed)) gdone = true;       // Newton stalled / nothing to repair: scan body
            if (__all(gdone)) break;
            if (!gdone) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) kind[c] = nkind[c];
            }
            __builtin_amdgcn_wave_barrier();
#undef LZ_TGT
        }

        {
#ifdef LZ_PV_STORE
            const DevView *pe = v.self;
            asm volatile("" : "+s"(pe));
#else
            const DevView *pe = &v;
#endif
            if (live && first) { pe->sto_fail[s] = good ? 0 : 1; if (good) pe->nu_valid[s] = 1; }
        }
        if (live && !good && first) anyFail += 1;
        __builtin_amdgcn_wave_barrier();
    }
#ifdef DOPF_STATS
    if (st_rounds) atomicAdd(&v.st->dbg_reason[0], st_rounds);
    if (st_newton) atomicAdd(&v.st->dbg_reason[1], st_newton);
    if (st_short) atomicAdd(&v.st->dbg_events, st_short);
    if (st_rounds) atomicMax(&v.st->dbg_reason[2], st_rounds);
    if (st_newton) atomicMax(&v.st->dbg_reason[3], st_newton);
#endif

    // The block's sums with ONE barrier.
    // The lane groups of a wave meet in a butterfly (lanes with the same timesteps sit LPS
    // apart: a fixed order), the four waves through LDS in wave order. (sto_warm_body: one lane group walks all NG rows in LDS.)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double x = accQ[c];
        if (LPS <= 8) x += lz_dpp<0x128>(x);                 // row_ror:8 (the lane 8 further on, modulo the row)
        if (LPS <= 16) x += __shfl_xor(x, 16);
        if (LPS <= 32) x += __shfl_xor(x, 32);
        accQ[c] = x;
    }
    __shared__ double wsumS[4][TP];
    __shared__ double wcostS[4];
    __shared__ int wfailS[4];
    {
        double cw = accCost;
        int fw = anyFail;
        for (int d = 32; d > 0; d >>= 1) { cw += __shfl_xor(cw, d); fw += __shfl_xor(fw, d); }
        if (lane == 0) { wcostS[tid >> 6] = cw; wfailS[tid >> 6] = fw; }
        if (lane < LPS) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) wsumS[tid >> 6][tbase + c] = accQ[c];
        }
    }
#ifndef LZ_NO_PV_FINAL
    const DevView *pf = v.self;                                  // (addresses from the view's copy in device memory: see the passes)
    asm volatile("" : "+s"(pf));
#else
    const DevView *pf = &v;
#endif
    TailView tv{};
    int tpar = 0;
    if (TAIL) { tv = *pf->tailDev; tpar = PERSIST ? ppar : pf->st->tail_par; }   // (TAIL: this launch adds into the accumulators; uniform scalar loads,
                                                                // in flight across the barrier. The per-launch fields of the view live in the kernel arguments only.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int blockFail = wfailS[0] + wfailS[1] + wfailS[2] + wfailS[3];
    if (tid < LPS) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            if (!LZ_IN(c)) continue;
            const double sum = ((wsumS[0][t] + wsumS[1][t]) + wsumS[2][t]) + wsumS[3][t];
            if (TAIL && blockFail == 0) acc_add(tv, tpar, t, sum, tv.scaleInj);
            else if (LINES) pf->part_T[(size_t)t * pf->rowsT + it.row + 1] = sum;
            else pf->part_sinj_w[(size_t)blk * T + t] = sum;
        }
    }
    if (tid == 0) {
        const double cw = ((wcostS[0] + wcostS[1]) + wcostS[2]) + wcostS[3];
        if (TAIL && blockFail == 0) acc_add(tv, tpar, T, cw, tv.scaleCost);
        else pf->part_scost_w[blk] = cw;
        pf->item_fail[blk] = blockFail;
    }
    if (blockFail != 0) __syncthreads();
    return blockFail;
#undef LZ_IN
#undef LZ_TLAST
#undef LZ_KAP
#undef LZ_AA
#undef LZ_IA
#undef LZ_IDET
#undef LZ_S2
}

}  // namespace dopf
