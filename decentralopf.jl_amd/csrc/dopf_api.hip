// dopf_api.hip — host side of libdopf_hip: the C ABI of include/dopf.h.
//
// One context = one GPU's shard of the agents + a replica of the O((N+L)T) consensus state.
// An ADMM iteration is a short kernel chain on one stream (with DOPF_F_OVERLAP_AGENTS the storage
// kernel is forked onto a side stream so that it overlaps the bandwidth-bound generator kernel):
//   [k_tables] -> k_gen_update -> k_sto_update -> [k_slack] -> k_reduce -> (all-reduce) -> k_dual -> k_price
// dopf_iterate replays it from a captured hipGraph (UNROLL iterations per graph launch) so that the
// loop runs without host round trips; convergence is tested on the device and freezes the state.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "dopf_ctx.h"

using namespace dopf;

namespace {

#ifndef DOPF_UNROLL
#define DOPF_UNROLL 16
#endif
constexpr int kUnroll = DOPF_UNROLL, kMid = 4;
constexpr int kCheckEvery = 512;
thread_local char g_create_err[512];

}  // namespace

namespace dopf {

int fail(dopf_ctx *c, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c ? c->err : g_create_err, 512, fmt, ap);
    va_end(ap);
    return code;
}

// a call that owns a temporary context (dopf_central_solve) hands the context's message to dopf_last_error(NULL)
// before the context goes away
void keep_error(const dopf_ctx *c)
{
    if (c && c->err[0]) { strncpy(g_create_err, c->err, 511); g_create_err[511] = 0; }
}

}  // namespace dopf

namespace {

template <class Tp>
int dev_alloc(dopf_ctx *c, Tp **out, size_t n, bool zero = true)
{
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(Tp);
    static const bool guard = getenv("DOPF_GUARD") != nullptr;
    if (guard) {
        // debugging aid: every array ends on a 2 MiB boundary of its own allocation, so that an access past its end leaves
        // the mapping at once (GPU memory fault at an address that names the array: the ranges are printed)
        const size_t two = 2u << 20, b16 = (bytes + 15) & ~(size_t)15, tot = (b16 + two - 1) / two * two;
        HIPCHK(c, hipMalloc(&p, tot));
        c->allocs.push_back(p);
        char *q = (char *)p + (tot - b16);
        static const bool chatty = atoi(getenv("DOPF_GUARD")) > 1;
        if (chatty) fprintf(stderr, "dopf guard: ctx %p alloc #%zu %zu bytes [%p, %p) base %p\n", (void *)c, c->allocs.size(), bytes, (void *)q, (void *)(q + b16), p);
        if (zero) HIPCHK(c, hipMemsetAsync(q, 0, bytes, c->main));
        *out = (Tp *)q;
        return DOPF_OK;
    }
    HIPCHK(c, hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    if (zero) HIPCHK(c, hipMemsetAsync(p, 0, bytes, c->main));
    *out = (Tp *)p;
    return DOPF_OK;
}

template <class Tp>
int dev_upload(dopf_ctx *c, const Tp **out, const std::vector<Tp> &h)
{
    Tp *p = nullptr;
    int rc = dev_alloc(c, &p, h.size(), false);
    if (rc) return rc;
    // blocking copy: the staging vector is usually a temporary that dies when this returns
    if (!h.empty()) HIPCHK(c, hipMemcpy(p, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
    *out = p;
    return DOPF_OK;
}

// split the node-sorted agent list into block work items that never cross a node boundary
void make_items(const std::vector<int> &node_sorted, int N, int chunk, std::vector<Item> &items,
                std::vector<int> &node_beg, std::vector<int> &node_item_beg)
{
    const int A = (int)node_sorted.size();
    node_beg.assign(N + 1, 0);
    for (int a = 0; a < A; ++a) node_beg[node_sorted[a] + 1]++;
    for (int n = 0; n < N; ++n) node_beg[n + 1] += node_beg[n];
    node_item_beg.assign(N + 1, 0);
    items.clear();
    for (int n = 0; n < N; ++n) {
        node_item_beg[n] = (int)items.size();
        for (int a = node_beg[n]; a < node_beg[n + 1]; a += chunk)
            items.push_back(Item{a, std::min(a + chunk, node_beg[n + 1]), n, 0});
    }
    node_item_beg[N] = (int)items.size();
}

}  // namespace

namespace dopf {

// single: the single-GPU dopf_iterate path (nothing reads cons between the reduce and the dual step)
// the quiet chain's dual/price kernel stages every partial row of its timestep in LDS: up to 96 KB of them
static bool quiet_rows_fit(int rows) { return rows > 0 && (size_t)rows * sizeof(double) <= 96 * 1024; }

static bool slice_dual(const DevView &v, bool single)
{
    const size_t NT = (size_t)v.N * v.T, LT = (size_t)v.L * v.T;
    return single && std::max(NT, LT) <= kSmallConsensus && NT <= 256;     // k_dual_price_small: 8 chunks of 32 entries
}

// the whole tail of the iteration rides in the x-update launch (k_agents / k_sto / the generator kernel): true single-GPU chain only
static bool tail_fused(const dopf_ctx *c, bool single)
{
    // no communicator, or a peer exchange that lives inside the tail block (copper plates; set up by dopf_xchg_init)
    return single && c->v.tailDev != nullptr && (c->comm == nullptr || c->tail_xchg);
}

// comm_quiet: a context on a peer exchange, networks whose dual step is the one-launch kernel, no line flagged at the last look
// (dopf_iterate): the single-GPU three-launch chain with the exchange of the node sums between k_slack and the dual/price
// kernel — no k_reduce launch (DevView::slackGlobal)
void enqueue_local(dopf_ctx *c, bool single, bool quiet, bool comm_quiet)
{
    DevView v = c->v;
    v.sliceDual = slice_dual(v, single) ? 1 : 0;
    v.tail = tail_fused(c, single) ? v.tailDev : nullptr;
    v.slackInDual = (single || comm_quiet) && v.slackDualOk;
    v.slackGlobal = comm_quiet ? 1 : 0;
    v.quiet = single && v.slackInDual && quiet;
    launch_tables(v, c->main);
    const bool fork = v.nGenItems > 0 && v.nStoItems > 0 && (c->q.flags & DOPF_F_OVERLAP_AGENTS);
    if (v.fuseAgents) {
        launch_agents_fused(v, c->lc, c->main);
    } else if (v.fuseNet) {
        launch_net_agents(v, c->lc, c->main);
    } else if (fork) {
        hipEventRecord(c->evFork, c->main);
        hipStreamWaitEvent(c->side, c->evFork, 0);
        launch_sto_update(v, c->lc, c->side);
        hipEventRecord(c->evJoin, c->side);
        launch_gen_update(v, c->main);
        hipStreamWaitEvent(c->main, c->evJoin, 0);
    } else {
        launch_gen_update(v, c->main);
        launch_sto_update(v, c->lc, c->main);
    }
    if (v.tail) return;                    // sums, dual step and stop test happened in the launch above
    if (!v.quiet) launch_slack(v, c->main);
    if (!v.slackInDual) launch_reduce(v, c->main);
}

void enqueue_apply(dopf_ctx *c, bool single, const XchgView *xd, bool quiet, bool comm_quiet)
{
    if (tail_fused(c, single)) return;
    DevView v = c->v;
    v.sliceDual = slice_dual(v, single) ? 1 : 0;
    v.slackInDual = (single || comm_quiet) && v.slackDualOk;
    v.slackGlobal = comm_quiet ? 1 : 0;
    v.quiet = single && v.slackInDual && quiet;
    launch_dual(v, c->main, xd);
}

void drop_graphs(dopf_ctx *c)
{
    if (c->graph1) hipGraphExecDestroy(c->graph1);
    if (c->graphM) hipGraphExecDestroy(c->graphM);
    if (c->graphU) hipGraphExecDestroy(c->graphU);
    if (c->graph1q) hipGraphExecDestroy(c->graph1q);
    if (c->graphMq) hipGraphExecDestroy(c->graphMq);
    if (c->graphUq) hipGraphExecDestroy(c->graphUq);
    c->graph1 = c->graphM = c->graphU = c->graph1q = c->graphMq = c->graphUq = nullptr;
    c->graphs_valid = c->graphs_q_valid = false;
}

int read_status(dopf_ctx *c)
{
    // (every dopf_iterate ends here: on short calls — 20 iterations of 30 us — the read-back is a visible share of the call)
    if (!c->host_pin && hipHostMalloc((void **)&c->host_pin, sizeof(Status), hipHostMallocDefault) != hipSuccess) {
        c->host_pin = nullptr;
        (void)hipGetLastError();
    }
    Status *dst = c->host_pin ? c->host_pin : &c->host_st;
    HIPCHK(c, hipMemcpyAsync(dst, c->v.st, sizeof(Status), hipMemcpyDeviceToHost, c->main));
    HIPCHK(c, hipStreamSynchronize(c->main));
    if (c->host_pin) c->host_st = *c->host_pin;
    if (c->host_st.res_set == 0 || c->host_st.res_set == 1)       // (k_dual_price_t1024 leaves the maxima as bit patterns, see Status)
        for (int k = 0; k < 3; ++k) memcpy(&c->host_st.res[k], &c->host_st.resbits2[c->host_st.res_set][k], sizeof(double));
    return DOPF_OK;
}

}  // namespace dopf

namespace {

// one iteration of the chain on the context's stream; a sharded context (dopf_comm_init) puts the all-reduce
// of the consensus buffer between the local sums and the dual step
int enqueue_iteration(dopf_ctx *c, bool quiet = false)
{
    const bool single = c->comm == nullptr;
    // copper plate + peer exchange: the one-block dual kernel exchanges the vector itself — the single-GPU chain, no extra launch
    const XchgView *xd = comm_xchg(c);
    if (xd && !(c->v.L == 0 && slice_dual(c->v, true)) && !c->tail_xchg) xd = nullptr;
    const bool like_single = single || xd != nullptr;        // (tail_xchg: the launch's tail block exchanges; nothing else is launched)
    // networks on a peer exchange while no line is flagged: k_slack's node sums are exchanged, the slack sums are formed behind
    // the exchange by the dual/price kernel — three launches + the exchange (DevView::slackGlobal)
    const XchgView *xn = comm_xchg(c);
    const bool comm_quiet = quiet && !like_single && xn != nullptr && c->comm_quiet_ok;
    enqueue_local(c, like_single, quiet && single, comm_quiet);
    if (comm_quiet) launch_xchg(c->v, *xn, c->main, true);
    else if (!like_single) { const int rc = comm_enqueue_allreduce(c); if (rc) return rc; }
    enqueue_apply(c, like_single, xd, quiet && single, comm_quiet);
    return DOPF_OK;
}

// several iterations in one launch (agents_persist.h): the single-GPU one-launch copper-plate chain only
bool persist_on(const dopf_ctx *c) { return c->v.persistOk && c->comm == nullptr && c->v.tailDev != nullptr; }

void enqueue_persist(dopf_ctx *c, int iters)
{
    DevView v = c->v;
    v.tail = v.tailDev;
    v.persistIters = iters;
    launch_agents_persist(v, c->lc, c->main);
}

int build_graph(dopf_ctx *c, int iters, hipGraphExec_t *out, bool quiet = false)
{
    hipGraph_t g = nullptr;
    // (a sharded context captures the RCCL collective with the kernels; thread-local mode keeps the capture
    // from tripping over what other host threads — other GPUs' drivers — do meanwhile)
    HIPCHK(c, hipStreamBeginCapture(c->main, c->comm ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed));
    int rc = DOPF_OK;
    if (persist_on(c)) enqueue_persist(c, iters);
    else for (int i = 0; i < iters && rc == DOPF_OK; ++i) rc = enqueue_iteration(c, quiet);
    const hipError_t ec = hipStreamEndCapture(c->main, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (ec != hipSuccess) return fail(c, DOPF_E_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(ec));
    hipError_t e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return fail(c, DOPF_E_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e));
    return DOPF_OK;
}

// a storage sub-problem that hit the root search's iteration cap leaves an unconverged row behind: report it
int check_solver(dopf_ctx *c)
{
    if (c->host_st.tail_timeout)
        return fail(c, DOPF_E_DEVICE, "the block sums of an iteration did not arrive in the launch's tail block in time; the state is not valid");
    if (c->host_st.xchg_timeout)
        return fail(c, DOPF_E_DEVICE, "peer exchange: a rank's part of the consensus sum did not arrive in time; the state is not valid");
    if (c->host_st.solver_fail > c->solver_fail_seen) {
        const unsigned long long n = c->host_st.solver_fail - c->solver_fail_seen;
        c->solver_fail_seen = c->host_st.solver_fail;
        return fail(c, DOPF_E_SOLVER, "%llu storage sub-problem(s) did not reach the root search's tolerance (%llu since creation)",
                    n, (unsigned long long)c->host_st.solver_fail);
    }
    return DOPF_OK;
}

}  // namespace

extern "C" {

const char *dopf_version(void) { return "libdopf_hip 0.1 (gfx950)"; }

void dopf_default_params(dopf_params *q)
{
    if (!q) return;
    memset(q, 0, sizeof *q);
    q->gamma = 0.3; q->w_flow = 10.0; q->w_prox = 1.0; q->eps = 1e-3; q->mask_thr = 1e-2;
    q->device = -1;
}

const char *dopf_last_error(const dopf_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

int dopf_create(dopf_ctx **out, const dopf_problem *p, const dopf_params *q)
{
    if (!out || !p || !q) return fail(nullptr, DOPF_E_INVALID, "null argument");
    *out = nullptr;
    if (p->N < 1 || p->T < 1 || p->L < 0 || p->G < 0 || p->S < 0)
        return fail(nullptr, DOPF_E_INVALID, "bad sizes N=%d L=%d T=%d G=%d S=%d", p->N, p->L, p->T, p->G, p->S);
    if (!(q->gamma > 0) || !(q->w_prox > 0) || !(q->w_flow > 0))
        return fail(nullptr, DOPF_E_INVALID, "gamma, w_prox, w_flow must be positive");
    if (!p->demand || (p->L > 0 && (!p->ptdf || !p->f_max)) || (p->G > 0 && (!p->gen_mc || !p->gen_pmax || !p->gen_node)) ||
        (p->S > 0 && (!p->sto_mc || !p->sto_pmax || !p->sto_emax || !p->sto_node)))
        return fail(nullptr, DOPF_E_INVALID, "null array in dopf_problem (every array with a positive extent must be given)");
    for (int g = 0; g < p->G; ++g)
        if (p->gen_node[g] < 0 || p->gen_node[g] >= p->N) return fail(nullptr, DOPF_E_INVALID, "gen_node[%d] out of range", g);
    for (int s = 0; s < p->S; ++s)
        if (p->sto_node[s] < 0 || p->sto_node[s] >= p->N) return fail(nullptr, DOPF_E_INVALID, "sto_node[%d] out of range", s);
    if ((int64_t)p->G * p->T > (int64_t)1 << 40) return fail(nullptr, DOPF_E_INVALID, "problem too large");
    Launch lc{};
    if (p->S > 0 && !sto_config_supported(p->T, &lc))
        return fail(nullptr, DOPF_E_UNSUPPORTED, "storage kernel supports T <= 512 (got %d)", p->T);
    if (2 * p->L > 4096) return fail(nullptr, DOPF_E_UNSUPPORTED, "table kernel supports L <= 2048 (got %d)", p->L);   // 4 * 2L doubles of LDS

    if (int rc1 = check_one_runtime(nullptr)) return rc1;
    int ndev = 0;
    hipError_t e0 = hipGetDeviceCount(&ndev);
    if (e0 != hipSuccess || ndev < 1)
        return fail(nullptr, DOPF_E_DEVICE, "no HIP device (%s); libdopf_hip has no CPU fallback", hipGetErrorString(e0));
    int dev = q->device;
    if (dev < 0) hipGetDevice(&dev);
    if (dev >= ndev) return fail(nullptr, DOPF_E_INVALID, "device %d of %d", dev, ndev);

    dopf_ctx *c = new (std::nothrow) dopf_ctx;
    if (!c) return fail(nullptr, DOPF_E_NOMEM, "out of host memory");
    c->device = dev;
    c->q = *q;
    c->lc = lc;
    DeviceGuard guard(dev);
    int rc = DOPF_OK;
    auto bail = [&](int code) { strncpy(g_create_err, c->err, 511); dopf_destroy(c); return code; };
#define TRY(x) do { rc = (x); if (rc) return bail(rc); } while (0)
#define HIPTRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(c, DOPF_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); return bail(DOPF_E_DEVICE); } } while (0)

    // (Round 3 ran big networks — configs[3] at full size — with the storage solve on a second stream by itself: 145 -> 136 us per
    // iteration. With generators and storages in one launch whose generator blocks work on both column halves at once
    // (k_net_agents) the one-stream form is ahead again, 117.8 vs 124.9 us; DOPF_F_OVERLAP_AGENTS stays for whoever asks.)
    if (q->stream) { c->main = (hipStream_t)q->stream; c->own_main = false; }
    else { HIPTRY(hipStreamCreateWithFlags(&c->main, hipStreamNonBlocking)); c->own_main = true; }
    // the side stream exists only when it is used: HIP maps streams onto a few hardware queues (4 by default), and a
    // process whose streams outnumber them pays barrier packets on every launch (measured: two contexts + PyTorch's
    // own streams made a 118-node iteration 4x slower)
    if (c->q.flags & DOPF_F_OVERLAP_AGENTS) {
        HIPTRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIPTRY(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
        HIPTRY(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
    }

    DevView &v = c->v;
    const int N = p->N, L = p->L, T = p->T, G = p->G, S = p->S;
    v.N = N; v.L = L; v.T = T; v.G = G; v.S = S; v.M2 = 2 * L;
    v.gamma = q->gamma; v.w_flow = q->w_flow; v.w_prox = q->w_prox; v.eps = q->eps; v.mask_thr = q->mask_thr;
    v.max_iters = q->max_iters;
    v.rootCap = (q->flags & DOPF_F_DEBUG_ROOT_CAP) ? 2 : 80;
    v.keepDeltas = (q->flags & DOPF_F_KEEP_DELTAS) ? 1 : 0;
    v.debugLeave = (q->flags & DOPF_F_DEBUG_LEAVE) ? 1 : 0;
    const int A = q->n_agents_global > 0 ? q->n_agents_global : G + S;
    v.invA = A > 0 ? 1.0 / (double)A : 0.0;
    v.nAgents = (double)A;
    {
        const double a0 = v.w_prox + v.gamma;
        v.cp_ia = 1.0 / a0; v.cp_idet = 1.0 / (a0 * a0 - v.gamma * v.gamma); v.cp_s2 = 2.0 / (a0 + v.gamma);
    }
    v.use_warm = (S > 0 && lc.stoNCH <= 3 && !(q->flags & DOPF_F_NO_WARM_START)) ? 1 : 0;
    // The lean active-set body (sto_lean.h): 32-bit element offsets; on a network only where the storage blocks outnumber the
    // chip's resident slots several times — its gain is instruction count, and a grid of one resident round is bound by one
    // block's latency chain, which is no shorter (configs[3]: 114 us against 120 at 100 k agents; its 12.5 k share 43.3 against 41.1).
    v.stoLean = ((q->flags & DOPF_F_STO_GENERAL) || (unsigned long long)S * T * sizeof(double) >= (1ull << 32) ||
                 (L > 0 && (long long)S * lc.stoLPS / 256 < 1024)) ? 0 : 1;
    v.genTT = std::min(T, 512);
    v.genR = 512 / v.genTT;
    v.genTT2 = (L == 0 && T % 2 == 0 && T / 2 <= 512) ? T / 2 : 0;
    v.fuseAgents = (v.genTT2 > 0 && v.genTT2 <= 256 && G > 0 && S > 0 && v.use_warm &&
                    !(q->flags & (DOPF_F_NO_FUSE | DOPF_F_OVERLAP_AGENTS))) ? 1 : 0;
    if (v.fuseAgents) {
        // one launch for all agents pays while its fixed cost matters and every storage block is resident from
        // the start (3 blocks of 256 per CU at the storage code's register count); see k_agents
        const int ng = 256 / lc.stoLPS;
        const int sch = (std::max(ng, (S + 2047) / 2048) + ng - 1) / ng * ng;
        const long long sto_blocks = (S + sch - 1) / sch + N - 1;
        if (sto_blocks > 3 * 256 || (long long)G * T > (8ll << 20)) v.fuseAgents = 0;
    }
    v.genR2 = v.genTT2 ? (v.fuseAgents ? 256 : 512) / v.genTT2 : 0;
    v.coldInWarm = (L > 0 && v.use_warm && !exp_env("DOPF_SPLIT_COLD")) ? 1 : 0;
    // networks: generators and storages in one launch (k_net_agents) unless the storages run on a stream of their own
    v.genTT256 = std::max(1, std::min((std::min(T, 512) + 1) / 2, 256 / v.genR));
    v.fuseNet = (L > 0 && G > 0 && S > 0 && v.coldInWarm && !(c->q.flags & (DOPF_F_NO_FUSE | DOPF_F_OVERLAP_AGENTS)) &&
                 v.genR * v.genTT256 <= 256 &&          // (T = 1: the 512-thread tiling has more agent lanes than such a block has threads)
                 !exp_env("DOPF_NO_NET_FUSE")) ? 1 : 0;

    // sort agents by node (stable), remember the permutation
    c->gen_perm.resize(G);
    c->sto_perm.resize(S);
    std::iota(c->gen_perm.begin(), c->gen_perm.end(), 0);
    std::iota(c->sto_perm.begin(), c->sto_perm.end(), 0);
    // generators: by node, then by marginal cost — a settled dispatch parks the cheap ones at pmax and the dear
    // ones at 0, so the rows the generator kernel may skip (and the ones it must stream) become contiguous
    std::stable_sort(c->gen_perm.begin(), c->gen_perm.end(), [&](int a, int b) {
        return p->gen_node[a] != p->gen_node[b] ? p->gen_node[a] < p->gen_node[b] : p->gen_mc[a] < p->gen_mc[b];
    });
    std::stable_sort(c->sto_perm.begin(), c->sto_perm.end(), [&](int a, int b) { return p->sto_node[a] < p->sto_node[b]; });
    std::vector<double> gmc(G), gpm(G), smc(S), spm(S), sem(S);
    std::vector<int> gnode(G), snode(S);
    for (int i = 0; i < G; ++i) { int a = c->gen_perm[i]; gmc[i] = p->gen_mc[a]; gpm[i] = p->gen_pmax[a]; gnode[i] = p->gen_node[a]; }
    for (int i = 0; i < S; ++i) { int a = c->sto_perm[i]; smc[i] = p->sto_mc[a]; spm[i] = p->sto_pmax[a]; sem[i] = p->sto_emax[a]; snode[i] = p->sto_node[a]; }
    for (int i = 0; i < G; ++i) if (!(gpm[i] >= 0)) { fail(c, DOPF_E_INVALID, "negative generator capacity"); return bail(DOPF_E_INVALID); }
    for (int i = 0; i < S; ++i) if (!(spm[i] >= 0) || !(sem[i] >= 0)) { fail(c, DOPF_E_INVALID, "negative storage capacity"); return bail(DOPF_E_INVALID); }

    TailView tvh{};
    bool tail_ok = false;
    {
        // One-launch iterations (kernels_agents.hip, "the tail of the iteration inside the x-update launch"): fixed-point scales
        // from the problem's bounds — |sum of net injections| <= sum of pmax (a storage's D - C lies in [-pmax, pmax]),
        // |cost| <= T * sum |mc| pmax (storages: 2 pmax) — so that no accumulator can overflow
        // (64 bits = sign + 53 value bits + the 10-bit arrival count).
        long double bi = 1.0L, bc = 1.0L;
        for (int i = 0; i < G; ++i) { bi += gpm[i]; bc += (long double)T * std::fabs(gmc[i]) * gpm[i]; }
        for (int i = 0; i < S; ++i) { bi += spm[i]; bc += (long double)T * std::fabs(smc[i]) * 2.0 * spm[i]; }
        const bool fin = std::isfinite((double)bi) && std::isfinite((double)bc);
        const int ki = fin ? 52 - (int)std::ceil(std::log2((double)bi)) : -1, kc = fin ? 52 - (int)std::ceil(std::log2((double)bc)) : -1;
        const bool chain = v.genTT2 > 0 && (S == 0 || v.use_warm) && G + S > 0;       // pair kernels / k_agents / k_sto
        tail_ok = (N == 1 && L == 0 && chain && ki >= 8 && kc >= 0 &&
                    !(q->flags & (DOPF_F_NO_TAIL_FUSE | DOPF_F_OVERLAP_AGENTS)) &&       // (two streams: the storage launch does not follow the generators')
                    !exp_env("DOPF_NO_TAIL_FUSE")) ? 1 : 0;
        tvh.accStride = (T + 1 + 15) / 16 * 16;                   // replicas on 128-byte lines of their own
        tvh.scaleInj = std::ldexp(1.0, std::max(0, std::min(ki, 60))); tvh.invInj = 1.0 / tvh.scaleInj;
        tvh.scaleCost = std::ldexp(1.0, std::max(0, std::min(kc, 60))); tvh.invCost = 1.0 / tvh.scaleCost;
    }
    std::vector<Item> gitems, sitems;
    int max_node_rows = 1;
    std::vector<int> ngb, nsb, ngib, nsib, row_of_pos, pos_of_row;
    {
        const int R = v.genTT2 ? v.genR2 : v.genR;
        // ~2048 blocks fill the chip several times over; in the fused launch the generator blocks share the wave slots with
        // the storage blocks and ~1536 somewhat larger ones come out ahead (measured on config2: 25.7 -> 24.0 us)
        // (with lines ~1024 blocks: the 118-node share 93.8 -> 87.3 us per iteration, config3 at full size 204 -> 203)
        // (networks, one launch for all agents: the generator blocks pass through the ~230 wave slots the storage blocks leave
        // free at that kernel's register count — ~512 larger ones: the 118-node share 51.6 -> 49.5 us per iteration)
        int target_items = v.fuseAgents ? 1536 : (L > 0 ? ((v.fuseNet && !(q->flags & DOPF_F_NET_SMALL_ITEMS)) ? 512 : 1024) : 2048);
        if (const char *e = exp_env("DOPF_GEN_TARGET_ITEMS")) target_items = std::max(1, atoi(e));     // (experiments)
        // streaming generator blocks (fused launch, one node): an item is ONE batch of loads, <= kGenStreamRows rows per lane
        const bool stream = v.fuseAgents && N == 1 && !exp_env("DOPF_NO_GEN_STREAM");
        if (stream) target_items = std::max(target_items, (G + kGenStreamRows * R - 1) / (kGenStreamRows * R));
        int chunk = std::max(R, (G + target_items - 1) / target_items);
        chunk = (chunk + R - 1) / R * R;
        if (stream) chunk = std::min(chunk, kGenStreamRows * R);
        make_items(gnode, N, chunk, gitems, ngb, ngib);
        v.genChunk = N == 1 ? chunk : 0;
        // (on short blocks the skip test costs more than the rows it saves: measured on config1/config2)
        v.genSkip = (v.genTT2 > 0 && chunk >= 8 * R && !(q->flags & DOPF_F_NO_ROW_SKIP)) ? 1 : 0;
        const int NG = S > 0 ? 256 / lc.stoLPS : 1;
        int sto_target = 2048;
        if (const char *e = exp_env("DOPF_STO_TARGET_ITEMS")) sto_target = std::max(1, atoi(e));     // (experiments)
        int schunk = std::max(NG, (S + sto_target - 1) / sto_target);
        schunk = (schunk + NG - 1) / NG * NG;
        // Big copper plates (the storage solve is a launch of its own: config4): as many passes per block as make the launch ONE
        // resident round — 3 blocks of 256 threads per CU at the storage code's register count — instead of several rounds of
        // shorter blocks (the blocks' fixed cost — entry, constants, the block's sums — is paid per block, and a round's last
        // blocks leave wave slots idle): config4 1 421 items of 2 passes -> 711 of 4: 78.7 -> 76.8 us per iteration. Only when that
        // round is well filled (a half-empty round of long blocks loses: 569 blocks of 5 passes 84.3 us).
        if (!v.fuseAgents && L == 0 && N == 1 && S > 0 && !exp_env("DOPF_STO_TARGET_ITEMS")) {
            int cus = 256;
            hipDeviceProp_t prop{};
            if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            const int slots = 3 * cus - 8, units = (S + NG - 1) / NG;
            const int passes = (units + slots - 1) / slots;
            if (passes >= 2 && (units + passes - 1) / passes >= (slots * 4) / 5) schunk = std::max(schunk, passes * NG);
        }
        make_items(snode, N, schunk, sitems, nsb, nsib);
        v.stoChunk = (N == 1 && !exp_env("DOPF_NO_STO_CHUNK")) ? schunk : 0;
    }
    v.maxNodeAgents = 0;
    for (int n = 0; n < N; ++n) v.maxNodeAgents = std::max(v.maxNodeAgents, (ngb[n + 1] - ngb[n]) + (nsb[n + 1] - nsb[n]));
    v.nGenItems = (int)gitems.size();
    v.nStoItems = (int)sitems.size();
    if (L > 0) {
        // rows of the transposed partial sums (DevView::part_T): node n owns rows [g0 + 2 s0, ...) — its generator items, then two per storage item
        // (node order: generator item i -> i + 2 s0, storage item k -> g0' + 2 k and the row behind it; then placed by the writer's XCD)
        v.rowsN = v.nGenItems + 2 * v.nStoItems;
        std::vector<int> xcd((size_t)v.rowsN, 0), cnt(8, 0), beg(9, 0);
        for (int i = 0; i < v.nGenItems; ++i) {
            gitems[i].row = i + 2 * nsib[gitems[i].node];
            xcd[(size_t)gitems[i].row] = ((v.fuseNet ? v.nStoItems : 0) + i) % 8;          // (k_net_agents: storage blocks in front)
        }
        for (int k = 0; k < v.nStoItems; ++k) {
            sitems[k].row = ngib[sitems[k].node + 1] + 2 * k;
            xcd[(size_t)sitems[k].row] = xcd[(size_t)sitems[k].row + 1] = k % 8;
        }
        for (int g = 0; g < v.rowsN; ++g) ++cnt[(size_t)xcd[(size_t)g]];
        for (int x = 0; x < 8; ++x) beg[(size_t)x + 1] = beg[(size_t)x] + (cnt[(size_t)x] + 15) / 16 * 16;
        v.rowsT = beg[8];
        row_of_pos.assign((size_t)std::max(v.rowsT, 1), -1); pos_of_row.assign((size_t)std::max(v.rowsN, 1), 0);
        std::vector<int> fillp(beg.begin(), beg.begin() + 8);
        for (int g = 0; g < v.rowsN; ++g) { const int p_ = fillp[(size_t)xcd[(size_t)g]]++; row_of_pos[(size_t)p_] = g; pos_of_row[(size_t)g] = p_; }
        for (int i = 0; i < v.nGenItems; ++i) gitems[i].row = pos_of_row[(size_t)gitems[i].row];
        for (int k = 0; k < v.nStoItems; ++k) sitems[k].row = pos_of_row[(size_t)sitems[k].row];       // (the warm-start row: the next position)
    }
    v.genBlocks = 0;
    if (v.fuseAgents && v.genChunk > 0 && !v.genSkip && v.genChunk <= kGenStreamRows * v.genR2 && !exp_env("DOPF_NO_GEN_STREAM")) {
        // as many generator blocks as find a wave slot next to the storage blocks (3 blocks of 256 per CU at the fused
        // kernel's register count): all resident from the start; at least a quarter of the chip
        int nb = 3 * 256 - v.nStoItems - (tail_ok ? 1 : 0);        // (tail in the launch: one slot for the tail block)
        if (const char *e = exp_env("DOPF_GEN_BLOCKS")) nb = atoi(e);          // (experiments)
        v.genBlocks = std::min(v.nGenItems, std::max(nb, 192));
    }
    v.genRows = v.genBlocks;
    {   // several iterations per launch: every block of the fused launch must be resident at once — 3 blocks of 256 threads per CU at
        // the kernel's register count, as many CUs as this device has (a grid that does not fit would time out, not hang)
        hipDeviceProp_t prop{};
        const bool known = hipGetDeviceProperties(&prop, c->device) == hipSuccess;
        const bool full = T == lc.stoLPS * lc.stoNCH && lc.stoLPS <= 32;
        v.persistOk = ((q->flags & DOPF_F_PERSIST) && tail_ok && v.fuseAgents && v.genBlocks > 0 && !v.genSkip && v.stoLean && full && S > 0 &&
                       known && v.nStoItems + v.genBlocks + 1 <= 3 * prop.multiProcessorCount) ? 1 : 0;
    }
    {
        // level-1 reduce blocks per node: ~16 items per block, at most 64 (and N*RB blocks in total)
        int max_items = 1;
        for (int n = 0; n < N; ++n)      // storage items: scan + warm rows; generators: items, or (one node, streaming) blocks
            max_items = std::max(max_items, (v.genRows > 0 ? v.genRows : ngib[n + 1] - ngib[n]) + 2 * (nsib[n + 1] - nsib[n]));
        max_node_rows = max_items;
        v.reduceRB = std::max(1, std::min(64, (max_items + 31) / 32));
        // Networks: nodes x timestep chunks already give hundreds of blocks, and more than one block per node means the
        // two-level sum — an agent-scope release (a write-back of the XCD's L2) and a ticket in EVERY block. configs[3] at full
        // size has a node with 36 partial rows: 2 blocks per node, 1 416 releases, k_reduce 38 us instead of 9. One block per
        // node walks up to 128 rows (four passes of its 8 x 4 loads in flight) before a second one is worth its ticket.
        if ((long long)N * ((T + 31) / 32) >= 128) v.reduceRB = std::max(1, std::min(64, (max_items + 127) / 128));
    }

    const size_t NT = (size_t)N * T, LT = (size_t)L * T;
    TRY(dev_upload(c, &v.demand, std::vector<double>(p->demand, p->demand + NT)));
    TRY(dev_upload(c, &v.ptdf, std::vector<double>(p->ptdf, p->ptdf + (size_t)L * N)));
    {
        std::vector<double> pt((size_t)L * N);
        for (int l = 0; l < L; ++l)
            for (int n = 0; n < N; ++n) pt[n + (size_t)N * l] = p->ptdf[l + (size_t)L * n];
        TRY(dev_upload(c, &v.ptdfT, pt));
    }
    TRY(dev_upload(c, &v.fmax, std::vector<double>(p->f_max, p->f_max + L)));
    TRY(dev_upload(c, &v.gen_mc, gmc)); TRY(dev_upload(c, &v.gen_pmax, gpm));
    {
        std::vector<double> mp(2 * (size_t)G);
        for (int i = 0; i < G; ++i) { mp[2 * (size_t)i] = gmc[i]; mp[2 * (size_t)i + 1] = gpm[i]; }
        const double *d = nullptr;
        TRY(dev_upload(c, &d, mp));
        v.gen_mp = reinterpret_cast<const double2 *>(d);
    }
    TRY(dev_upload(c, &v.sto_mc, smc)); TRY(dev_upload(c, &v.sto_pmax, spm)); TRY(dev_upload(c, &v.sto_emax, sem));
    TRY(dev_upload(c, &v.gen_items, gitems)); TRY(dev_upload(c, &v.sto_items, sitems));
    TRY(dev_upload(c, &v.node_gen_beg, ngb)); TRY(dev_upload(c, &v.node_sto_beg, nsb));
    {
        // a generator moves by at most pmax per iteration, a storage's net injection D - C by at most 2 pmax
        std::vector<double> win(N, 0.0);
        for (int i = 0; i < G; ++i) win[gnode[i]] = std::max(win[gnode[i]], gpm[i]);
        for (int i = 0; i < S; ++i) win[snode[i]] = std::max(win[snode[i]], 2.0 * spm[i]);
        for (int n = 0; n < N; ++n) win[n] = win[n] * (1.0 + 1e-9) + 1e-9;
        TRY(dev_upload(c, &v.node_win, win));
        // per line: the largest |kap| W_n over the nodes, kap = w2 h / (w2 + gamma) — if the line's slack offset clears it,
        // every agent of every node has that slack active (or inactive) and the sums are a dot product (k_reduce)
        const double w2 = 2.0 * q->w_flow, inv = 1.0 / (w2 + q->gamma);
        std::vector<double> reach(L, 0.0);
        for (int l = 0; l < L; ++l)
            for (int n = 0; n < N; ++n) reach[l] = std::max(reach[l], std::fabs(w2 * p->ptdf[l + (size_t)L * n] * inv) * win[n]);
        for (int l = 0; l < L; ++l) reach[l] = reach[l] * (1.0 + 1e-9) + 1e-300;      // strictly beyond every node's own reach
        TRY(dev_upload(c, &v.line_reach, reach));
    }
    TRY(dev_upload(c, &v.node_gitem_beg, ngib)); TRY(dev_upload(c, &v.node_sitem_beg, nsib));
    TRY(dev_alloc(c, &v.P, (size_t)G * T));
    TRY(dev_alloc(c, &v.gen_state, G));            // zero = "all zero", which is what P is now
    TRY(dev_alloc(c, &v.D, (size_t)S * T)); TRY(dev_alloc(c, &v.C, (size_t)S * T)); TRY(dev_alloc(c, &v.E, (size_t)S * T));
    if (L > 0) { TRY(dev_alloc(c, &v.dltG, (size_t)G * T)); TRY(dev_alloc(c, &v.dltS, (size_t)S * T)); }
    TRY(dev_alloc(c, &v.lam, T)); TRY(dev_alloc(c, &v.mu, LT)); TRY(dev_alloc(c, &v.rho, LT));
    TRY(dev_alloc(c, &v.lam_used, T)); TRY(dev_alloc(c, &v.mu_used, LT)); TRY(dev_alloc(c, &v.rho_used, LT));
    TRY(dev_alloc(c, &v.inj, NT)); TRY(dev_alloc(c, &v.s, T)); TRY(dev_alloc(c, &v.flow, LT));
    TRY(dev_alloc(c, &v.avgU, LT)); TRY(dev_alloc(c, &v.avgK, LT)); TRY(dev_alloc(c, &v.price, NT));
    TRY(dev_alloc(c, &v.s_used, T)); TRY(dev_alloc(c, &v.flow_used, LT)); TRY(dev_alloc(c, &v.avgU_used, LT)); TRY(dev_alloc(c, &v.avgK_used, LT));
    if (L > 0) {
        TRY(dev_alloc(c, &v.tb_beta, NT * v.M2)); TRY(dev_alloc(c, &v.tb_psi, NT * v.M2));
        TRY(dev_alloc(c, &v.tb_slope, NT * (v.M2 + 1))); TRY(dev_alloc(c, &v.tb_psi0, NT));
        TRY(dev_alloc(c, &v.tb_m, NT));
        TRY(dev_alloc(c, &v.part_U, NT * L)); TRY(dev_alloc(c, &v.part_K, NT * L)); TRY(dev_alloc(c, &v.node_dsum, NT));
        TRY(dev_alloc(c, &v.walk_flag, LT)); TRY(dev_alloc(c, &v.walk_any, T)); TRY(dev_alloc(c, &v.tab_skip, T));
    }
    TRY(dev_alloc(c, &v.part_ginj, (size_t)v.nGenItems * T)); TRY(dev_alloc(c, &v.part_gcost, v.nGenItems));
    TRY(dev_alloc(c, &v.part_sinj, (size_t)v.nStoItems * T)); TRY(dev_alloc(c, &v.part_scost, v.nStoItems));
    TRY(dev_alloc(c, &v.part_sinj_w, (size_t)v.nStoItems * T)); TRY(dev_alloc(c, &v.part_scost_w, v.nStoItems));
    if (L > 0) { TRY(dev_upload(c, &v.row_of_pos, row_of_pos)); TRY(dev_upload(c, &v.pos_of_row, pos_of_row)); }
    if (L > 0) TRY(dev_alloc(c, &v.part_T, (size_t)v.rowsT * T));      // (networks: the ADMM kernels' layout; the rows above serve dopf_central_solve)
    if (L > 0) TRY(dev_alloc(c, &v.prev_node, NT));
    TRY(dev_alloc(c, &v.nu_prev, (size_t)S * T)); TRY(dev_alloc(c, &v.nu_valid, S)); TRY(dev_alloc(c, &v.sto_fail, S));
    TRY(dev_alloc(c, &v.item_fail, v.nStoItems));
    TRY(dev_alloc(c, &v.part2, (size_t)N * v.reduceRB * T)); TRY(dev_alloc(c, &v.part2_cost, v.reduceRB));
    TRY(dev_alloc(c, &v.reduce_ticket, (size_t)N * ((T + 31) / 32)));
    TRY(dev_alloc(c, &v.dual_ticket, 1));
    if ((v.nGenItems + v.nStoItems) / kAccRep + 2 > 1000) tail_ok = false;   // (the 10-bit arrival count of a replica slot)
    if (tail_ok) {
        TRY(dev_alloc(c, &tvh.acc, (size_t)2 * kAccRep * tvh.accStride));
        tvh.expect = (v.fuseAgents && v.genBlocks > 0 && !v.genSkip ? v.genBlocks : v.nGenItems) + v.nStoItems;
        TailView *tvd = nullptr;
        TRY(dev_alloc(c, &tvd, 1, false));
        HIPTRY(hipMemcpy(tvd, &tvh, sizeof tvh, hipMemcpyHostToDevice));
        v.tailDev = tvd;
    }
    v.splitDual = exp_env("DOPF_SPLIT_DUAL") ? 1 : 0;
    {
        // networks whose dual step is the one-launch kernel (k_dual_price_t1024: <= 256 lines and nodes, consensus state beyond the
        // one-block kernel): it builds the tables too, with as many waves as find LDS scratch (<= 8) next to its own ~30 KB
        const size_t n1 = std::max(NT, LT);
        const size_t per_wave = (4 * (size_t)v.M2 + 1) * sizeof(double), own = 25 * 1024 + (4 * (size_t)N + 3 * (size_t)L) * sizeof(double);
        int tw = 0;
        if (L > 0 && L <= 256 && N <= 256 && n1 > kSmallConsensus && !v.splitDual && !exp_env("DOPF_TABLES_LAUNCH"))
            tw = (int)std::min<size_t>(8, (128 * 1024 - std::min<size_t>(own, 128 * 1024)) / per_wave);
        v.tablesInDual = tw;
        {   // its dynamic LDS: q[N] | d[L] | G[L] | S[L] | the rows of its timestep (quiet chain), later the tables' scratch | sd[N] win[N] na[N]
            const size_t scratch = (size_t)tw * (4 * (size_t)v.M2 + 1);
            v.dualRowsOff = N + 3 * L;
            v.dualSdOff = v.dualRowsOff + (int)std::max(scratch, (size_t)(quiet_rows_fit(v.rowsT) ? v.rowsN : 0));
            v.dualLdsBytes = (int)(((size_t)v.dualSdOff + 3 * (size_t)N) * sizeof(double));
        }
        // the same kernel forms the slack sums of its timestep (see DevView::slackInDual); DOPF_F_NO_TAIL_FUSE keeps the
        // k_reduce launch (the chain a sharded context runs: bitwise comparisons against it)
        v.slackDualOk = L > 0 && L <= 256 && N <= 256 && n1 > kSmallConsensus && !v.splitDual && !(q->flags & DOPF_F_NO_TAIL_FUSE) &&
                        !exp_env("DOPF_REDUCE_LAUNCH");
        // ... and, while no line is flagged, the node sums too (the quiet chain: no k_slack launch; DevView::quiet, dopf_iterate)
        // (up to 32 rows per node: one batch of the eight lanes' four loads. configs[3] at full size has 25 and is where the gain
        // ends — the node sums cost the dual kernel what k_slack and its boundary cost, 119.3 us per iteration either way)
        c->quiet_ok = v.slackDualOk && !(q->flags & DOPF_F_KEEP_DELTAS) && quiet_rows_fit(v.rowsT) &&
                      !(q->flags & DOPF_F_NO_QUIET);
        // the same chain on a peer exchange (k_slack stays: its node sums are what is exchanged): a function of the problem's shape and
        // the flags only — every rank decides alike
        c->comm_quiet_ok = v.slackDualOk && !(q->flags & (DOPF_F_KEEP_DELTAS | DOPF_F_NO_QUIET));
    }
    double *cons = nullptr;
    TRY(dev_alloc(c, &cons, NT + 2 * LT + 1));
    c->own_cons = cons;
    v.cons = cons;
    TRY(dev_alloc(c, &v.st, 1));
    Status st0{};
    st0.iteration = 1;                                      // admm.jl:29
    st0.res_set = -1;
    HIPTRY(hipMemcpyAsync(v.st, &st0, sizeof st0, hipMemcpyHostToDevice, c->main));
    // "no result yet" state: zeros everywhere, injection = -demand (helpers/results.jl:14-73)
    launch_derive(v, c->main, false);     // all-zero primal state: the consensus buffer is already zero
    HIPTRY(hipGetLastError());
    HIPTRY(hipStreamSynchronize(c->main));
    c->host_st = st0;
    {   // the view itself in device memory (non-inlined device functions take a pointer to it)
        DevView *dv = nullptr;
        TRY(dev_alloc(c, &dv, 1, false));
        v.self = dv;
        HIPTRY(hipMemcpy(dv, &v, sizeof(DevView), hipMemcpyHostToDevice));
    }
#undef TRY
#undef HIPTRY
    *out = c;
    return DOPF_OK;
}

void dopf_destroy(dopf_ctx *c)
{
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->main) hipStreamSynchronize(c->main);
    if (c->side) hipStreamSynchronize(c->side);
    drop_graphs(c);
    comm_release(c);
    for (void *p : c->allocs) hipFree(p);
    if (c->evFork) hipEventDestroy(c->evFork);
    if (c->evJoin) hipEventDestroy(c->evJoin);
    if (c->host_pin) hipHostFree(c->host_pin);
    if (c->evT0) hipEventDestroy(c->evT0);
    if (c->evT1) hipEventDestroy(c->evT1);
    if (c->side) hipStreamDestroy(c->side);
    if (c->own_main && c->main) hipStreamDestroy(c->main);
    delete c;
}

int dopf_iterate(dopf_ctx *c, int32_t n_iters, int32_t *iters_done, int32_t *converged)
{
    if (!c || n_iters < 0) return fail(c, DOPF_E_INVALID, "bad argument");
    DeviceGuard guard(c->device);
    const int before = c->host_st.iters_total;
    // with a communicator of more than one rank the chain is launched eagerly unless the caller opts into capturing
    // the collective (DOPF_F_COMM_GRAPH): either way nothing synchronises with the host inside the loop, and the
    // host enqueues an iteration's four launches faster than the GPU retires them
    bool eager = (c->q.flags & DOPF_F_NO_GRAPH) != 0 || (!comm_capturable(c) && !(c->q.flags & DOPF_F_COMM_GRAPH));
    if (!eager && !c->graphs_valid) {
        int rc = build_graph(c, 1, &c->graph1);
        if (rc == DOPF_OK) rc = build_graph(c, kMid, &c->graphM);
        if (rc == DOPF_OK) rc = build_graph(c, kUnroll, &c->graphU);
        if (rc) {
            if (!c->comm) return rc;
            // the collective refused to be captured: launch the same chain eagerly from now on (no host sync either way)
            drop_graphs(c);
            (void)hipGetLastError();
            c->q.flags |= DOPF_F_NO_GRAPH;
            eager = true;
        } else {
            c->graphs_valid = true;
        }
    }
    // Enqueue in slices and look at the device status word between slices, so that a converged (or capped)
    // run stops being fed no-op launches; one sync per kCheckEvery iterations costs nothing measurable.
    // DOPF_F_TIME_CALLS (measurement): an event in front of the call's first launch and one behind its last — the device-side
    // span of the call's iterations, without the host's launch latency in front and the status read-back behind
    const bool timed = (c->q.flags & DOPF_F_TIME_CALLS) != 0 && n_iters > 0;
    if (timed) {
        if (!c->evT0) { HIPCHK(c, hipEventCreate(&c->evT0)); HIPCHK(c, hipEventCreate(&c->evT1)); }
        HIPCHK(c, hipEventRecord(c->evT0, c->main));
    }
    int left = n_iters;
    while (left > 0) {
        int slice = std::min(left, kCheckEvery);
        left -= slice;
        // the quiet chain (networks, no line flagged at the last look: no k_slack launch) has graphs of its own, built when first used
        // (on a peer exchange: the chain without k_reduce, see enqueue_iteration — every rank takes the same decision from the same
        // replicated status word at the same iteration)
        bool quiet = c->quiet && (c->comm == nullptr ? c->quiet_ok : (c->comm_quiet_ok && comm_xchg(c) != nullptr && !c->tail_xchg));
        if (quiet && !eager && !c->graphs_q_valid) {
            int rc = build_graph(c, 1, &c->graph1q, true);
            if (rc == DOPF_OK) rc = build_graph(c, kMid, &c->graphMq, true);
            if (rc == DOPF_OK) rc = build_graph(c, kUnroll, &c->graphUq, true);
            if (rc) return rc;
            c->graphs_q_valid = true;
        }
        const int asked = slice, total_before = c->host_st.iters_total;
        if (eager && persist_on(c)) {
            for (; slice > 0; slice -= std::min(slice, kUnroll)) enqueue_persist(c, std::min(slice, kUnroll));
        } else if (eager) {
            for (int i = 0; i < slice; ++i) { const int rc = enqueue_iteration(c, quiet); if (rc) return rc; }
        } else {
            for (; slice >= kUnroll; slice -= kUnroll) HIPCHK(c, hipGraphLaunch(quiet ? c->graphUq : c->graphU, c->main));
            for (; slice >= kMid; slice -= kMid) HIPCHK(c, hipGraphLaunch(quiet ? c->graphMq : c->graphM, c->main));
            for (; slice > 0; --slice) HIPCHK(c, hipGraphLaunch(quiet ? c->graph1q : c->graph1, c->main));
        }
        HIPCHK(c, hipGetLastError());
        if (timed && left == 0) HIPCHK(c, hipEventRecord(c->evT1, c->main));
        const int rc = read_status(c);
        if (rc) return rc;
        if (c->host_st.halt == 2) {
            // the quiet chain parked itself: its last dual step flagged a line, the iterations behind it were no-ops. Release the
            // device, go back to the chain with k_slack and feed what is left of the slice again.
            HIPCHK(c, hipMemsetAsync(&c->v.st->halt, 0, sizeof(int), c->main));
            c->host_st.halt = 0;
            c->quiet = false;
            ++c->quiet_parked;
            left += asked - (c->host_st.iters_total - total_before);
            continue;
        }
        if (c->host_st.halt) break;
        c->quiet = (c->comm == nullptr ? c->quiet_ok : c->comm_quiet_ok) && c->host_st.walk_last == 0;
    }
    c->last_call_ms = -1.0;
    if (timed && left == 0) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->evT0, c->evT1) == hipSuccess) c->last_call_ms = (double)ms;
    }
    if (n_iters == 0) { const int rc = read_status(c); if (rc) return rc; }
    if (iters_done) *iters_done = c->host_st.iters_total - before;
    if (converged) *converged = c->host_st.converged;
    return check_solver(c);
}

double dopf_last_call_ms(const dopf_ctx *c) { return c ? c->last_call_ms : -1.0; }

int dopf_iterate_timed(dopf_ctx *c, int32_t n_iters, dopf_timing *out)
{
    if (!c || !out || n_iters < 1 || n_iters > 4096) return fail(c, DOPF_E_INVALID, "bad argument");
    DeviceGuard guard(c->device);
    DevView v = c->v;
    if (c->comm) return fail(c, DOPF_E_INVALID, "dopf_iterate_timed drives the single-GPU chain: not on a context joined to a communicator");
    v.sliceDual = slice_dual(v, true) ? 1 : 0;
    v.tail = tail_fused(c, true) ? v.tailDev : nullptr;
    v.slackInDual = v.slackDualOk;
    v.quiet = v.slackInDual && c->quiet_ok && c->quiet;        // (the chain dopf_iterate would launch now)
    enum { E_T0, E_T1, E_G0, E_G1, E_S0, E_S1, E_K0, E_K1, E_R1, E_D1, E_X0, E_X1, E_N };
    struct Events {                                     // destroyed on every way out
        std::vector<hipEvent_t> v;
        ~Events() { for (auto e : v) if (e) hipEventDestroy(e); }
    } evs;
    evs.v.assign((size_t)n_iters * E_N, nullptr);
    std::vector<hipEvent_t> &ev = evs.v;
    for (auto &e : ev) HIPCHK(c, hipEventCreate(&e));
    const bool fork = v.nGenItems > 0 && v.nStoItems > 0 && (c->q.flags & DOPF_F_OVERLAP_AGENTS);
    for (int i = 0; i < n_iters; ++i) {
        hipEvent_t *e = &ev[(size_t)i * E_N];
        hipEventRecord(e[E_T0], c->main);
        launch_tables(v, c->main);
        hipEventRecord(e[E_T1], c->main);
        hipStream_t ss = fork ? c->side : c->main;
        if (fork) { hipEventRecord(c->evFork, c->main); hipStreamWaitEvent(c->side, c->evFork, 0); }
        hipEventRecord(e[E_G0], c->main);
        if (v.fuseAgents) launch_agents_fused(v, c->lc, c->main);
        else if (v.fuseNet) launch_net_agents(v, c->lc, c->main);
        else launch_gen_update(v, c->main);
        hipEventRecord(e[E_G1], c->main);
        hipEventRecord(e[E_S0], ss);
        if (!v.fuseAgents && !v.fuseNet) launch_sto_update(v, c->lc, ss);
        hipEventRecord(e[E_S1], ss);
        if (fork) { hipEventRecord(c->evJoin, c->side); hipStreamWaitEvent(c->main, c->evJoin, 0); }
        hipEventRecord(e[E_K0], c->main);
        if (!v.tail && !v.quiet) launch_slack(v, c->main);
        hipEventRecord(e[E_K1], c->main);
        if (!v.tail && !v.slackInDual) launch_reduce(v, c->main);
        hipEventRecord(e[E_R1], c->main);
        if (!v.tail) launch_dual(v, c->main);
        hipEventRecord(e[E_D1], c->main);
        hipEventRecord(e[E_X0], c->main);
        hipEventRecord(e[E_X1], c->main);
    }
    HIPCHK(c, hipGetLastError());
    int rc = read_status(c);
    if (rc) return rc;
    if (c->host_st.halt == 2) {            // the quiet chain parked itself inside the timed iterations (the ones behind were no-ops)
        HIPCHK(c, hipMemsetAsync(&c->v.st->halt, 0, sizeof(int), c->main));
        c->host_st.halt = 0;
        c->quiet = false;
        ++c->quiet_parked;
    } else if (!c->host_st.halt) {
        c->quiet = c->quiet_ok && c->host_st.walk_last == 0;
    }
    if (c->side) HIPCHK(c, hipStreamSynchronize(c->side));
    memset(out, 0, sizeof *out);
    auto ms = [&](hipEvent_t a, hipEvent_t b) { float f = 0.f; hipEventElapsedTime(&f, a, b); return (double)f; };
    for (int i = 0; i < n_iters; ++i) {
        hipEvent_t *e = &ev[(size_t)i * E_N];
        out->tables_ms += ms(e[E_T0], e[E_T1]);
        out->gen_ms += ms(e[E_G0], e[E_G1]);
        out->sto_ms += ms(e[E_S0], e[E_S1]);
        out->slack_ms += ms(e[E_K0], e[E_K1]);
        out->reduce_ms += ms(e[E_K1], e[E_R1]);
        out->dual_ms += ms(e[E_R1], e[E_D1]);
        out->iter_ms += ms(e[E_T0], e[E_D1]);
        out->empty_ms += ms(e[E_X0], e[E_X1]);
    }
    const double inv = 1.0 / n_iters;
    out->tables_ms *= inv; out->gen_ms *= inv; out->sto_ms *= inv; out->slack_ms *= inv;
    out->reduce_ms *= inv; out->dual_ms *= inv; out->iter_ms *= inv; out->empty_ms *= inv;
    out->iters = n_iters;
    out->agents_fused = v.fuseAgents || v.fuseNet;
    out->tail_fused = v.tail ? 1 : 0;
    out->slack_in_dual = (!v.tail && v.slackInDual) ? 1 : 0;
    out->quiet = v.quiet ? 1 : 0;
    out->sto_lean = (v.stoLean && v.S > 0 && v.use_warm) ? 1 : 0;
    out->persist = persist_on(c) ? 1 : 0;
    return DOPF_OK;
}

int dopf_local_update(dopf_ctx *c)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    c->quiet = false;                      // (iterations driven from outside: the next dopf_iterate looks at the flags before it trusts them)
    enqueue_local(c, false);
    HIPCHK(c, hipGetLastError());
    return DOPF_OK;
}

int dopf_apply_consensus(dopf_ctx *c)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    c->quiet = false;
    enqueue_apply(c, false);
    HIPCHK(c, hipGetLastError());
    return DOPF_OK;
}

int64_t dopf_consensus_size(const dopf_ctx *c)
{
    return c ? (int64_t)c->v.N * c->v.T + 2 * (int64_t)c->v.L * c->v.T + 1 : 0;
}

void *dopf_consensus_ptr(dopf_ctx *c) { return c ? (void *)c->v.cons : nullptr; }

int dopf_bind_consensus(dopf_ctx *c, void *device_ptr)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->main));
    c->v.cons = device_ptr ? (double *)device_ptr : (double *)c->own_cons;
    HIPCHK(c, hipMemcpy(const_cast<DevView *>(c->v.self), &c->v, sizeof(DevView), hipMemcpyHostToDevice));
    drop_graphs(c);
    return DOPF_OK;
}

int dopf_sync(dopf_ctx *c, int32_t *iteration, int32_t *converged)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    int rc = read_status(c);
    if (rc) return rc;
    if (iteration) *iteration = c->host_st.iteration;
    if (converged) *converged = c->host_st.converged;
    return check_solver(c);
}

static int copy_out(dopf_ctx *c, double *dst, const double *src, size_t n)
{
    if (!dst || n == 0) return DOPF_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, c->main));
    return DOPF_OK;
}

int dopf_get_duals(dopf_ctx *c, double *lambda, double *mu, double *rho)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    const size_t LT = (size_t)c->v.L * c->v.T;
    int rc;
    if ((rc = copy_out(c, lambda, c->v.lam, c->v.T))) return rc;
    if ((rc = copy_out(c, mu, c->v.mu, LT))) return rc;
    if ((rc = copy_out(c, rho, c->v.rho, LT))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->main));
    return DOPF_OK;
}

int dopf_get_duals_used(dopf_ctx *c, double *lambda, double *mu, double *rho)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    const size_t LT = (size_t)c->v.L * c->v.T;
    int rc;
    if ((rc = copy_out(c, lambda, c->v.lam_used, c->v.T))) return rc;
    if ((rc = copy_out(c, mu, c->v.mu_used, LT))) return rc;
    if ((rc = copy_out(c, rho, c->v.rho_used, LT))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->main));
    return DOPF_OK;
}

// device rows are in node-sorted order; hand them back in the caller's agent order
static int get_rows(dopf_ctx *c, double *dst, const double *src, const std::vector<int> &perm)
{
    if (!dst || perm.empty()) return DOPF_OK;
    const int T = c->v.T;
    std::vector<double> tmp((size_t)perm.size() * T);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, c->main));
    HIPCHK(c, hipStreamSynchronize(c->main));
    for (size_t i = 0; i < perm.size(); ++i) memcpy(dst + (size_t)perm[i] * T, tmp.data() + i * T, sizeof(double) * T);
    return DOPF_OK;
}

static int set_rows(dopf_ctx *c, double *dst, const double *src, const std::vector<int> &perm)
{
    if (!src || perm.empty()) return DOPF_OK;
    const int T = c->v.T;
    std::vector<double> tmp((size_t)perm.size() * T);
    for (size_t i = 0; i < perm.size(); ++i) memcpy(tmp.data() + i * T, src + (size_t)perm[i] * T, sizeof(double) * T);
    HIPCHK(c, hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, c->main));
    HIPCHK(c, hipStreamSynchronize(c->main));
    return DOPF_OK;
}

int dopf_get_primal(dopf_ctx *c, double *P, double *D, double *C, double *E)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    int rc;
    if ((rc = get_rows(c, P, c->v.P, c->gen_perm))) return rc;
    if ((rc = get_rows(c, D, c->v.D, c->sto_perm))) return rc;
    if ((rc = get_rows(c, C, c->v.C, c->sto_perm))) return rc;
    if (E && c->level_from_primal) {         // the level is rebuilt from D and C (the iteration's kernels do not store it)
        launch_derive_level(c->v, c->main);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->main));
    }
    if ((rc = get_rows(c, E, c->v.E, c->sto_perm))) return rc;
    return DOPF_OK;
}

int dopf_get_consensus(dopf_ctx *c, double *injection, double *avg_U, double *avg_K, double *line_util, double *total_cost)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    const size_t NT = (size_t)c->v.N * c->v.T, LT = (size_t)c->v.L * c->v.T;
    int rc;
    if ((rc = copy_out(c, injection, c->v.inj, NT))) return rc;
    if ((rc = copy_out(c, avg_U, c->v.avgU, LT))) return rc;
    if ((rc = copy_out(c, avg_K, c->v.avgK, LT))) return rc;
    if ((rc = copy_out(c, line_util, c->v.flow, LT))) return rc;
    if ((rc = read_status(c))) return rc;
    if (total_cost) *total_cost = c->host_st.total_cost;
    return DOPF_OK;
}

int dopf_get_residuals(dopf_ctx *c, double *a, double *b, double *r, int32_t *iteration)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    int rc = read_status(c);
    if (rc) return rc;
    if (a) *a = c->host_st.res[0];
    if (b) *b = c->host_st.res[1];
    if (r) *r = c->host_st.res[2];
    if (iteration) *iteration = c->host_st.iteration;
    return DOPF_OK;
}

int dopf_get_nodal_price(dopf_ctx *c, int32_t which, double *out)
{
    if (!c || !out) return DOPF_E_INVALID;
    const int N = c->v.N, L = c->v.L, T = c->v.T;
    std::vector<double> lam(T), mu((size_t)L * T), rho((size_t)L * T), ptdf((size_t)L * N);
    int rc = which ? dopf_get_duals(c, lam.data(), mu.data(), rho.data())
                   : dopf_get_duals_used(c, lam.data(), mu.data(), rho.data());
    if (rc) return rc;
    DeviceGuard guard(c->device);
    if (L > 0) HIPCHK(c, hipMemcpy(ptdf.data(), c->v.ptdf, ptdf.size() * sizeof(double), hipMemcpyDeviceToHost));
    // lambda_t + sum_l (mu + rho)[l,t] * ptdf[l,n]   (helpers/network_elements.jl:16-25)
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < N; ++n) {
            double p = lam[t];
            for (int l = 0; l < L; ++l) p += (mu[l + (size_t)L * t] + rho[l + (size_t)L * t]) * ptdf[l + (size_t)L * n];
            out[n + (size_t)N * t] = p;
        }
    return DOPF_OK;
}

// scratch of the getters that reduce on the device: allocated once (an allocation per call would be a device-wide
// synchronisation per Result in a record-everything loop), freed with the context's other arrays
static int getter_scratch(dopf_ctx *c, double **out)
{
    if (!c->getter_scratch) {
        int rc = dev_alloc(c, &c->getter_scratch, 3 * (size_t)c->v.N * c->v.T, false);
        if (rc) return rc;
    }
    *out = c->getter_scratch;
    return DOPF_OK;
}

int dopf_get_node_results(dopf_ctx *c, double *generation, double *discharge, double *charge)
{
    if (!c) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    const size_t NT = (size_t)c->v.N * c->v.T;
    double *tmp = nullptr;
    int rc = getter_scratch(c, &tmp);
    if (rc) return rc;
    launch_node_results(c->v, tmp, tmp + NT, tmp + 2 * NT, c->main);
    HIPCHK(c, hipGetLastError());
    double *outs[3] = {generation, discharge, charge};
    for (int k = 0; k < 3; ++k)
        if (outs[k]) HIPCHK(c, hipMemcpyAsync(outs[k], tmp + k * NT, NT * sizeof(double), hipMemcpyDeviceToHost, c->main));
    HIPCHK(c, hipStreamSynchronize(c->main));
    return DOPF_OK;
}

// Result.penalty_term (src/structures/results.jl:66-70): the agents' three penalty vectors summed, one device pass
int dopf_get_penalty_sums(dopf_ctx *c, double *penalty)
{
    if (!c || !penalty) return DOPF_E_INVALID;
    const DevView &v = c->v;
    if (v.L == 0 || !v.keepDeltas)
        return fail(c, DOPF_E_UNSUPPORTED, "the agents' injection changes are kept on the device only with lines and DOPF_F_KEEP_DELTAS "
                                           "(otherwise: dopf_get_agent_penalty with the change passed in, agent by agent)");
    DeviceGuard guard(c->device);
    const size_t NT = (size_t)v.N * v.T;
    double *tmp = nullptr;
    int rc = getter_scratch(c, &tmp);
    if (rc) return rc;
    launch_penalty_sums(v, tmp, c->main);
    HIPCHK(c, hipGetLastError());
    std::vector<double> h(3 * NT);
    HIPCHK(c, hipMemcpyAsync(h.data(), tmp, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->main));
    HIPCHK(c, hipStreamSynchronize(c->main));
    for (int k = 0; k < 3; ++k)
        for (int t = 0; t < v.T; ++t) {
            double sum = 0.0;
            for (int n = 0; n < v.N; ++n) sum += h[k * NT + (size_t)n * v.T + t];       // (node order: fixed)
            penalty[(size_t)k * v.T + t] = sum;
        }
    return DOPF_OK;
}

int dopf_set_state(dopf_ctx *c, const double *P, const double *D, const double *C, const double *avg_U,
                   const double *avg_K, const double *lambda, const double *mu, const double *rho, int32_t iteration)
{
    if (!c || iteration < 1) return fail(c, DOPF_E_INVALID, "bad argument");
    DeviceGuard guard(c->device);
    DevView &v = c->v;
    const size_t LT = (size_t)v.L * v.T;
    HIPCHK(c, hipStreamSynchronize(c->main));
    int rc;
    if ((rc = set_rows(c, v.P, P, c->gen_perm))) return rc;
    if ((rc = set_rows(c, v.D, D, c->sto_perm))) return rc;
    if ((rc = set_rows(c, v.C, C, c->sto_perm))) return rc;
    auto up = [&](double *dst, const double *src, size_t n) -> int {
        if (!src || n == 0) return DOPF_OK;
        HIPCHK(c, hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice));
        return DOPF_OK;
    };
    if ((rc = up(v.avgU, avg_U, LT))) return rc;
    if ((rc = up(v.avgK, avg_K, LT))) return rc;
    if ((rc = up(v.lam, lambda, v.T))) return rc;
    if ((rc = up(v.mu, mu, LT))) return rc;
    if ((rc = up(v.rho, rho, LT))) return rc;
    if (v.S > 0) HIPCHK(c, hipMemset(v.nu_valid, 0, sizeof(int) * v.S));   // stored prices no longer match the state
    if (v.G > 0 && P) {                                                     // nor do the row summaries of P
        std::vector<int> mixed(v.G, 2);
        HIPCHK(c, hipMemcpy(v.gen_state, mixed.data(), sizeof(int) * v.G, hipMemcpyHostToDevice));
    }
    c->quiet = false;                      // (flags are formed anew from the state handed in)
    Status st{};
    HIPCHK(c, hipMemcpy(&st, v.st, sizeof st, hipMemcpyDeviceToHost));
    st.iteration = iteration;
    st.converged = 0;
    st.halt = (v.max_iters > 0 && iteration > v.max_iters);
    st.resbits[0] = st.resbits[1] = st.resbits[2] = 0;
    memset(st.resbits2, 0, sizeof st.resbits2);
    HIPCHK(c, hipMemcpy(v.st, &st, sizeof st, hipMemcpyHostToDevice));
    launch_derive(v, c->main, true);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->main));
    c->host_st = st;
    return DOPF_OK;
}

// Convergence.{lambda_res, mue_res, rho_res}[end] (src/structures/convergence.jl:5-12, src/optimization/convergence.jl:5-13):
// |dual after the last update - dual the last solve used|, entry by entry
int dopf_get_residual_vectors(dopf_ctx *c, double *lam_res, double *mu_res, double *rho_res)
{
    if (!c) return DOPF_E_INVALID;
    const size_t T = c->v.T, LT = (size_t)c->v.L * c->v.T;
    std::vector<double> a(std::max(T, LT)), b(std::max(T, LT));
    auto diff = [&](double *out, const double *now, const double *used, size_t n) -> int {
        if (!out || n == 0) return DOPF_OK;
        DeviceGuard guard(c->device);
        HIPCHK(c, hipMemcpyAsync(a.data(), now, n * sizeof(double), hipMemcpyDeviceToHost, c->main));
        HIPCHK(c, hipMemcpyAsync(b.data(), used, n * sizeof(double), hipMemcpyDeviceToHost, c->main));
        HIPCHK(c, hipStreamSynchronize(c->main));
        for (size_t i = 0; i < n; ++i) out[i] = fabs(a[i] - b[i]);
        return DOPF_OK;
    };
    int rc;
    if ((rc = diff(lam_res, c->v.lam, c->v.lam_used, T))) return rc;
    if ((rc = diff(mu_res, c->v.mu, c->v.mu_used, LT))) return rc;
    return diff(rho_res, c->v.rho, c->v.rho_used, LT);
}

// ResultGenerator/ResultStorage.{U, K, penalty_term} of the last solve (src/structures/results.jl:1-17,
// src/optimization/penalty_terms.jl:1-37): not stored per agent on the device — the slacks are eliminated in closed
// form — but recomputed here from the agent's injection change and the consensus state that solve read.
static int agent_result(dopf_ctx *c, int32_t agent, const double *delta_in, double *U, double *K, double *pen)
{
    const DevView &v = c->v;
    const int T = v.T, L = v.L, N = v.N, G = v.G, S = v.S;
    if (agent < 0 || agent >= G + S) return fail(c, DOPF_E_INVALID, "agent %d of %d", agent, G + S);
    if (!delta_in && L == 0) {
        if (pen) return fail(c, DOPF_E_UNSUPPORTED, "copper plate: the agent's injection change is not kept on the device (pass it in)");
        return DOPF_OK;         // no lines: U and K are empty
    }
    if (!delta_in && !v.keepDeltas)
        return fail(c, DOPF_E_UNSUPPORTED, "the agents' injection changes are kept on the device only with DOPF_F_KEEP_DELTAS (or pass the change in)");
    DeviceGuard guard(c->device);
    const bool is_gen = agent < G;
    const std::vector<int> &perm = is_gen ? c->gen_perm : c->sto_perm;
    const int want = is_gen ? agent : agent - G;
    int row = -1;
    for (size_t i = 0; i < perm.size(); ++i) if (perm[i] == want) { row = (int)i; break; }
    if (row < 0) return fail(c, DOPF_E_INVALID, "agent not found");
    HIPCHK(c, hipStreamSynchronize(c->main));
    // node of the agent: the item lists are per node; read it from the node ranges
    std::vector<int> beg(N + 1);
    HIPCHK(c, hipMemcpy(beg.data(), is_gen ? v.node_gen_beg : v.node_sto_beg, sizeof(int) * (N + 1), hipMemcpyDeviceToHost));
    int n = 0;
    while (n + 1 < N && row >= beg[n + 1]) ++n;
    std::vector<double> dl(T), s(T), f((size_t)L * T), aU((size_t)L * T), aK((size_t)L * T), h(L), F(L);
    if (delta_in) memcpy(dl.data(), delta_in, sizeof(double) * T);
    else HIPCHK(c, hipMemcpy(dl.data(), (is_gen ? v.dltG : v.dltS) + (size_t)row * T, sizeof(double) * T, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(s.data(), v.s_used, sizeof(double) * T, hipMemcpyDeviceToHost));
    if (L > 0) {
        const size_t LT = (size_t)L * T;
        HIPCHK(c, hipMemcpy(f.data(), v.flow_used, sizeof(double) * LT, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(aU.data(), v.avgU_used, sizeof(double) * LT, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(aK.data(), v.avgK_used, sizeof(double) * LT, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(h.data(), v.ptdf + (size_t)L * n, sizeof(double) * L, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(F.data(), v.fmax, sizeof(double) * L, hipMemcpyDeviceToHost));
    }
    const double w2 = 2.0 * v.w_flow, g = v.gamma;
    for (int t = 0; t < T; ++t) {
        double up = 0.0, lo = 0.0;
        for (int l = 0; l < L; ++l) {
            const size_t i = l + (size_t)L * t;
            const double fl = f[i] + h[l] * dl[t];
            const double u = std::max(0.0, (g * aU[i] - w2 * (fl - F[l])) / (w2 + g));
            const double k = std::max(0.0, (g * aK[i] + w2 * (fl + F[l])) / (w2 + g));
            if (U) U[i] = u;
            if (K) K[i] = k;
            up += (fl + u - F[l]) * (fl + u - F[l]);                                           // penalty_terms.jl:10-20
            lo += (k - fl - F[l]) * (k - fl - F[l]);                                           // :23-37
        }
        if (pen) { pen[t] = (s[t] + dl[t]) * (s[t] + dl[t]); pen[T + t] = up; pen[2 * T + t] = lo; }   // :3-7
    }
    return DOPF_OK;
}

int dopf_get_agent_slacks(dopf_ctx *c, int32_t agent, double *U, double *K)
{
    if (!c) return DOPF_E_INVALID;
    return agent_result(c, agent, nullptr, U, K, nullptr);
}

int dopf_get_agent_penalty(dopf_ctx *c, int32_t agent, const double *delta, double *penalty)
{
    if (!c || !penalty) return DOPF_E_INVALID;
    return agent_result(c, agent, delta, nullptr, nullptr, penalty);
}

// diagnostics: the breakpoint table of Psi_{n,t} as the last x-update saw it (L > 0 only)
int dopf_debug_table(dopf_ctx *c, int32_t n, int32_t t, double *beta, double *psi, double *slope, double *psi0, int32_t *m)
{
    if (!c || c->v.L == 0 || n < 0 || n >= c->v.N || t < 0 || t >= c->v.T) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    const DevView &v = c->v;
    const size_t at = (size_t)n + (size_t)v.N * t;
    HIPCHK(c, hipStreamSynchronize(c->main));
    HIPCHK(c, hipMemcpy(beta, v.tb_beta + at * v.M2, sizeof(double) * v.M2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(psi, v.tb_psi + at * v.M2, sizeof(double) * v.M2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(slope, v.tb_slope + at * (v.M2 + 1), sizeof(double) * (v.M2 + 1), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(psi0, v.tb_psi0 + at, sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(m, v.tb_m + at, sizeof(int), hipMemcpyDeviceToHost));
    return DOPF_OK;
}

// diagnostics (DOPF_STATS builds): cumulative storage-kernel counters {scans, wave loop trips, events}
int dopf_debug_stats(dopf_ctx *c, uint64_t *out3 /* 15 values */)
{
    if (!c || !out3) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    int rc = read_status(c);
    if (rc) return rc;
    out3[0] = c->host_st.dbg_scans; out3[1] = c->host_st.dbg_wave_loops; out3[2] = c->host_st.dbg_events;
    {   // warm-start kernel, LAST iteration: storages it solved / left to the scan kernel
        std::vector<int> f(c->v.nStoItems);
        if (!f.empty()) HIPCHK(c, hipMemcpy(f.data(), c->v.item_fail, f.size() * sizeof(int), hipMemcpyDeviceToHost));
        uint64_t nf = 0;
        for (int x : f) nf += (uint64_t)x;
        out3[4] = c->v.use_warm ? nf : (uint64_t)c->v.S;
        out3[3] = (uint64_t)c->v.S - out3[4];
    }
    for (int i = 0; i < 4; ++i) out3[5 + i] = c->host_st.dbg_reason[i];
    for (int i = 0; i < 6; ++i) out3[9 + i] = c->host_st.dbg_cyc[i];
    return DOPF_OK;
}

int dopf_debug_quiet(dopf_ctx *c, int64_t *out3)
{
    if (!c || !out3) return DOPF_E_INVALID;
    // (a context on a communicator: the chain without k_reduce on a peer exchange, DevView::slackGlobal)
    out3[0] = (c->comm ? (c->comm_quiet_ok && comm_xchg(c) != nullptr && !c->tail_xchg) : c->quiet_ok) ? 1 : 0;
    out3[1] = c->quiet ? 1 : 0; out3[2] = (int64_t)c->quiet_parked;
    return DOPF_OK;
}

int dopf_debug_timeline(dopf_ctx *c, uint64_t *out, int32_t n)
{
    if (!c || !out) return DOPF_E_INVALID;
    DeviceGuard guard(c->device);
    hipStreamSynchronize(c->main);
    return debug_timeline((unsigned long long *)out, n);
}

int64_t dopf_solver_failures(dopf_ctx *c)
{
    if (!c) return -1;
    DeviceGuard guard(c->device);
    if (read_status(c)) return -1;
    return (int64_t)c->host_st.solver_fail;
}

}  // extern "C"
