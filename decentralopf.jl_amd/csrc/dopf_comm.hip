// dopf_comm.hip — the consensus sum across GPUs, inside the library.
//
// The reference has no parallelism; what shards is the agent loop of
// optimize_all_subproblems! (src/optimization/subproblems.jl:1-17) and the only cross-agent data flow of an
// iteration is the agent sum of Result(...) (src/structures/results.jl:72-106): ONE all-reduce(sum) of the
// consensus buffer [N*T injections | L*T sum U | L*T sum K | cost] per iteration. Duals are then updated
// redundantly on every rank, so they stay identical without a second collective.
//
// Two ways to get there, both ending in the same per-context chain
//     k_agents ... k_reduce -> ncclAllReduce(consensus buffer, f64, sum, ctx stream) -> k_dual...
// which dopf_iterate launches without any host synchronisation per iteration: eagerly by default when the communicator
// has more than one rank, captured into the iteration hipGraphs with DOPF_F_COMM_GRAPH (RCCL collectives are capturable;
// if the capture is refused the chain falls back to eager launches). Capture has only been exercised at world size 1:
//   * one process per GPU (torch.distributed.run, MPI, Distributed.jl ...): every rank creates its own
//     context from its shard of the agents and calls dopf_comm_init(ctx, world, rank, id) with the 128-byte
//     id of dopf_comm_unique_id, carried from rank 0 to the others over any host channel;
//   * one process, n GPUs (a Julia `ccall` host needs no launcher): dopf_multi_create shards the agents
//     itself, creates one context per device, joins them with ncclCommInitAll and drives each from its own
//     host thread in dopf_multi_iterate.
// RCCL (librccl.so.1, the ROCm build of NCCL: rings / trees over xGMI) is loaded at run time, on the first
// call that needs it: a single-GPU user of libdopf_hip never touches it.
// DOPF_F_COMM_HOST replaces RCCL by a sum through host memory inside dopf_multi_iterate: a debugging
// transport that lets several shards share ONE device (RCCL refuses two ranks on a device), which is how
// the sharding logic of dopf_multi_* is tested on a one-GPU box.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dopf_ctx.h"

using namespace dopf;

struct dopf_comm_state {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool host_sum = false;      // DOPF_F_COMM_HOST: dopf_multi_iterate adds the buffers itself
    bool p2p = false;           // peer exchange (k_xchg) instead of an RCCL collective
    XchgView xv{};
    void *xbuf = nullptr;       // this rank's receive area (fine-grained device memory)
    size_t xbytes = 0;
    std::vector<void *> opened; // IPC mappings of other processes' areas
    XchgView *xdev = nullptr;   // the view in device memory (the tail block of a one-launch iteration reads it there)
};

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char err[512] = {0};
    int code = DOPF_E_DEVICE;      // what a caller returns with `err`: DOPF_E_UNSUPPORTED for "two copies mapped"
};

Rccl g_rccl;
std::once_flag g_rccl_once;

// What is mapped into this process: every librccl, every HIP runtime, and the directory a PyTorch build lives in.
struct Mapped {
    std::vector<std::string> rccl, hip;
    std::string torch_dir;
};

int scan_cb(struct dl_phdr_info *info, size_t, void *data)
{
    Mapped *m = static_cast<Mapped *>(data);
    const char *path = info->dlpi_name;
    if (!path || !*path) return 0;
    const char *base = strrchr(path, '/');
    base = base ? base + 1 : path;
    auto add = [](std::vector<std::string> &v, const char *p_) {
        for (const auto &x : v) if (x == p_) return;
        v.emplace_back(p_);
    };
    if (strncmp(base, "librccl.so", 10) == 0) add(m->rccl, path);
    if (strncmp(base, "libamdhip64.so", 14) == 0) add(m->hip, path);
    if (strncmp(base, "libtorch_hip.so", 15) == 0 || strncmp(base, "libc10_hip.so", 13) == 0) m->torch_dir.assign(path, (size_t)(base - path));
    return 0;
}

Mapped scan_mapped()
{
    Mapped m;
    dl_iterate_phdr(scan_cb, &m);
    return m;
}

// One RCCL per process, and the one that belongs to the HIP runtime the process runs on. PyTorch's wheels bundle their
// own librccl.so next to their own libamdhip64.so; a system RCCL (built against the system's runtime) bound to that
// bundled runtime, or two RCCLs side by side, end in heap corruption (round 2's abort inside dopf_create: the library
// had dlopen'ed the system's copy, PyTorch's arrived later). So: a copy that is mapped already is THE copy; otherwise
// the librccl.so next to the mapped HIP runtime or next to libtorch_hip.so (the copy PyTorch would load later); only a
// process without any of that (a C or Julia host on the system's ROCm) gets the system's library.
void load_rccl()
{
    const Mapped m = scan_mapped();
    if (m.rccl.size() > 1) {
        snprintf(g_rccl.err, sizeof g_rccl.err, "two copies of RCCL are mapped into this process (%s, %s): refusing to pick one",
                 m.rccl[0].c_str(), m.rccl[1].c_str());
        g_rccl.code = DOPF_E_UNSUPPORTED;
        return;
    }
    if (m.rccl.size() == 1) g_rccl.lib = dlopen(m.rccl[0].c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!g_rccl.lib) {
        std::vector<std::string> next_to;
        if (!m.torch_dir.empty()) next_to.push_back(m.torch_dir);
        for (const auto &h : m.hip) next_to.push_back(h.substr(0, h.rfind('/') + 1));
        for (const auto &d : next_to) {
            for (const char *n : {"librccl.so", "librccl.so.1"}) {
                if (g_rccl.lib) break;
                const std::string cand = d + n;
                if (access(cand.c_str(), R_OK) == 0) g_rccl.lib = dlopen(cand.c_str(), RTLD_NOW | RTLD_LOCAL);
            }
        }
    }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (g_rccl.lib) break;
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!g_rccl.lib) { snprintf(g_rccl.err, sizeof g_rccl.err, "cannot load librccl.so.1: %s", dlerror()); return; }
#define DOPF_SYM(field, name)                                                                   \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name));           \
    if (!g_rccl.field) { snprintf(g_rccl.err, sizeof g_rccl.err, "librccl: no symbol %s", name); g_rccl.lib = nullptr; return; }
    DOPF_SYM(GetUniqueId, "ncclGetUniqueId")
    DOPF_SYM(CommInitRank, "ncclCommInitRank")
    DOPF_SYM(CommInitAll, "ncclCommInitAll")
    DOPF_SYM(CommDestroy, "ncclCommDestroy")
    DOPF_SYM(AllReduce, "ncclAllReduce")
    DOPF_SYM(GetErrorString, "ncclGetErrorString")
#undef DOPF_SYM
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (g_rccl.lib) {
        // somebody may have loaded a second copy since (a PyTorch imported after the library had to load RCCL on a system
        // runtime): say so instead of running two collectives libraries in one address space
        const Mapped m = scan_mapped();
        if (m.rccl.size() > 1) {
            snprintf(g_rccl.err, sizeof g_rccl.err, "two copies of RCCL are mapped into this process (%s, %s): load PyTorch (or whatever brings "
                     "its own RCCL) before the first dopf_comm_* / dopf_multi_* call", m.rccl[0].c_str(), m.rccl[1].c_str());
            g_rccl.code = DOPF_E_UNSUPPORTED;
            return nullptr;
        }
    }
    return g_rccl.lib ? &g_rccl : nullptr;
}

}  // namespace

namespace dopf {

// dopf_create: more than one HIP runtime in the process means one of them cannot see the devices (and anything bound to
// the other one is an ABI mismatch): a clear message instead of "no device" or a crash later
int check_one_runtime(dopf_ctx *c)
{
    const Mapped m = scan_mapped();
    if (m.hip.size() > 1)
        return fail(c, DOPF_E_UNSUPPORTED, "two HIP runtimes are mapped into this process (%s, %s): load the one the process is to use "
                    "first (PyTorch bundles its own; decentralopf_jl_amd._capi.hip_api() pins it before libdopf_hip.so is loaded)",
                    m.hip[0].c_str(), m.hip[1].c_str());
    return DOPF_OK;
}

}  // namespace dopf

namespace {

}  // namespace

namespace dopf {

int comm_enqueue_allreduce(dopf_ctx *c)
{
    dopf_comm_state *cs = c->comm;
    if (!cs || cs->host_sum) return DOPF_OK;       // (host transport: dopf_multi_iterate adds)
    if (cs->p2p) {
        if (cs->xv.world < 1) return fail(c, DOPF_E_INVALID, "peer exchange exported but not initialised (dopf_xchg_init)");
        launch_xchg(c->v, cs->xv, c->main);
        return DOPF_OK;
    }
    const size_t n = (size_t)c->v.N * c->v.T + 2 * (size_t)c->v.L * c->v.T + 1;
    const ncclResult_t r = g_rccl.AllReduce(c->v.cons, c->v.cons, n, ncclDouble, ncclSum, cs->comm, c->main);
    if (r != ncclSuccess) return fail(c, DOPF_E_DEVICE, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    return DOPF_OK;
}

int comm_world(const dopf_ctx *c) { return c->comm ? c->comm->world : 1; }

const XchgView *comm_xchg(const dopf_ctx *c) { return (c->comm && c->comm->p2p && c->comm->xv.world > 0) ? &c->comm->xv : nullptr; }

// a chain with an RCCL collective of more than one rank is launched eagerly unless the caller opts into capturing it;
// the peer exchange is an ordinary kernel and is always captured
bool comm_capturable(const dopf_ctx *c) { return !c->comm || c->comm->p2p || c->comm->host_sum || c->comm->world == 1; }

void comm_release(dopf_ctx *c)
{
    if (!c->comm) return;
    if (c->comm->comm && g_rccl.lib) g_rccl.CommDestroy(c->comm->comm);
    for (void *p : c->comm->opened) hipIpcCloseMemHandle(p);
    if (c->comm->xbuf) hipFree(c->comm->xbuf);
    if (c->comm->xdev) hipFree(c->comm->xdev);
    delete c->comm;
    c->comm = nullptr;
}

}  // namespace dopf

// ------------------------------------------------------------------------------------------------
// one process, n GPUs
// ------------------------------------------------------------------------------------------------
struct dopf_multi {
    int n = 0;
    std::vector<dopf_ctx *> ctx;
    std::vector<int> g0, g1, s0, s1;        // shard i owns generators [g0,g1), storages [s0,s1) of the caller's lists
    int G = 0, S = 0, T = 0;
    bool host_sum = false;
    std::vector<double> hsum, hpart;
    char err[512] = {0};
};

namespace {

thread_local char g_multi_err[512];

int mfail(dopf_multi *m, int code, const char *msg)
{
    snprintf(m ? m->err : g_multi_err, 512, "%s", msg);
    return code;
}

// run f(i) for every shard, each on its own host thread (a context is driven by one thread at a time; the
// RCCL collectives of the n ranks must be in flight together)
template <class F>
int for_each_shard(dopf_multi *m, F f)
{
    std::vector<int> rc(m->n, DOPF_OK);
    if (m->n == 1) { rc[0] = f(0); }
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < m->n; ++i) th.emplace_back([&, i] { rc[i] = f(i); });
        for (auto &t : th) t.join();
    }
    for (int i = 0; i < m->n; ++i)
        if (rc[i]) { snprintf(m->err, 512, "shard %d: %s", i, dopf_last_error(m->ctx[i])); return rc[i]; }
    return DOPF_OK;
}

// ---- peer exchange set-up -------------------------------------------------------------------------------------
// layout of a rank's receive area: [world] hello words | [2][world][chunks] flags | [2][chunks] sum flags |
// (256-byte aligned) [2][world][n] doubles | [2][n] summed chunks
struct XchgLayout {
    size_t n, nchunks, flags_off, sflags_off, data_off, sum_off, bytes;
    XchgLayout(size_t n_, int world) : n(n_), nchunks((n_ + kXchgChunk - 1) / kXchgChunk)
    {
        flags_off = (size_t)kXchgMaxWorld * sizeof(unsigned long long);
        sflags_off = flags_off + 2 * (size_t)world * nchunks * sizeof(unsigned long long);
        data_off = (sflags_off + 2 * nchunks * sizeof(unsigned long long) + 255) / 256 * 256;
        sum_off = data_off + 2 * (size_t)world * n * sizeof(double);
        bytes = sum_off + 2 * n * sizeof(double);
    }
};

unsigned long long xchg_timeout_ticks()
{
    double ms = 20000.0;                            // generous: ranks reach their first iteration seconds apart
    if (const char *e = getenv("DOPF_XCHG_TIMEOUT_MS")) ms = atof(e);
    return (unsigned long long)(ms * 1e5);          // wall_clock64: 100 MHz
}

int xchg_alloc(dopf_ctx *c, dopf_comm_state *cs, int world)
{
    const XchgLayout lay((size_t)dopf_consensus_size(c), world);
    cs->xbytes = lay.bytes;
    // fine-grained: stores of other devices become visible to this device's loads without cache maintenance on this side
    HIPCHK(c, hipExtMallocWithFlags(&cs->xbuf, lay.bytes, hipDeviceMallocFinegrained));
    HIPCHK(c, hipMemset(cs->xbuf, 0, lay.bytes));
    HIPCHK(c, hipDeviceSynchronize());
    return DOPF_OK;
}

void xchg_fill_view(dopf_ctx *c, dopf_comm_state *cs, int world, int rank, void *const *areas)
{
    const XchgLayout lay((size_t)dopf_consensus_size(c), world);
    XchgView &x = cs->xv;
    x.world = world; x.me = rank; x.nchunks = (int)lay.nchunks; x.n = lay.n; x.timeout_ticks = xchg_timeout_ticks();
    // more than one chunk per rank: every chunk has an owner that adds the ranks' copies and hands the sum to everybody
    // (2 n doubles in and out per rank and iteration instead of world x n)
    x.rs = ((int)lay.nchunks > world && world > 1 && !(c->q.flags & DOPF_F_XCHG_ALLGATHER)) ? 1 : 0;
    if (c->q.flags & DOPF_F_XCHG_OWNER) x.rs = 1;                // (tests: the owner form on small vectors and at world 1)
    for (int r = 0; r < world; ++r) {
        x.flags[r] = reinterpret_cast<unsigned long long *>((char *)areas[r] + lay.flags_off);
        x.sflags[r] = reinterpret_cast<unsigned long long *>((char *)areas[r] + lay.sflags_off);
        x.data[r] = reinterpret_cast<double *>((char *)areas[r] + lay.data_off);
        x.sum[r] = reinterpret_cast<double *>((char *)areas[r] + lay.sum_off);
    }
    // Copper plate with one-launch iterations: the launch's tail block holds the rank's whole vector (T sums
    // and the cost) the moment the last block's adds have landed — it exchanges it itself (all-gather form, one chunk), so a
    // rank's iteration stays ONE launch. The view goes to device memory, the context's TailView points at it.
    c->tail_xchg = false;
    // (T + 1 <= 128: all slots in one pass of the tail block's lane pairs — the flags are published once, behind the whole vector)
    if (c->v.tailDev && c->v.L == 0 && c->v.N == 1 && lay.n == (size_t)c->v.T + 1 && lay.nchunks == 1 && c->v.T + 1 <= 128 &&
        !(c->q.flags & DOPF_F_NO_TAIL_XCHG)) {
        DeviceGuard guard(c->device);
        bool ok = cs->xdev || hipMalloc((void **)&cs->xdev, sizeof(XchgView)) == hipSuccess;
        ok = ok && hipMemcpy(cs->xdev, &x, sizeof(XchgView), hipMemcpyHostToDevice) == hipSuccess;
        const XchgView *dp = cs->xdev;
        ok = ok && hipMemcpy((char *)const_cast<TailView *>(c->v.tailDev) + offsetof(TailView, xchg), &dp, sizeof dp, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        c->tail_xchg = ok;
    }
}

// A rendezvous that failed after xchg_fill_view: the context goes back to its single-GPU chain — the tail block's pointer to
// the exchange's view is cleared on the DEVICE (the context's graphs read the TailView there), the view is marked unusable and
// the graphs are dropped. (Without this a later dopf_iterate made the tail block exchange with peers that never joined.)
void xchg_undo(dopf_ctx *c, dopf_comm_state *cs)
{
    if (c->tail_xchg && c->v.tailDev) {
        const XchgView *none = nullptr;
        (void)hipMemcpy((char *)const_cast<TailView *>(c->v.tailDev) + offsetof(TailView, xchg), &none, sizeof none, hipMemcpyHostToDevice);
    }
    c->tail_xchg = false;
    cs->xv.world = 0;
    drop_graphs(c);
}

}  // namespace

extern "C" {

int dopf_xchg_export(dopf_ctx *c, int32_t world, void *handle64)
{
    if (!c || !handle64 || world < 1 || world > kXchgMaxWorld) return fail(c, DOPF_E_INVALID, "bad argument (world <= %d)", kXchgMaxWorld);
    if (c->comm) return fail(c, DOPF_E_INVALID, "context already has a communicator");
    DeviceGuard guard(c->device);
    dopf_comm_state *cs = new (std::nothrow) dopf_comm_state;
    if (!cs) return fail(c, DOPF_E_NOMEM, "out of host memory");
    cs->world = world; cs->p2p = true;
    int rc = xchg_alloc(c, cs, world);
    if (rc == DOPF_OK) {
        hipIpcMemHandle_t h;
        static_assert(sizeof h == DOPF_XCHG_HANDLE_BYTES, "hipIpcMemHandle_t size");
        const hipError_t e = hipIpcGetMemHandle(&h, cs->xbuf);
        if (e != hipSuccess) rc = fail(c, DOPF_E_DEVICE, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
        else memcpy(handle64, &h, sizeof h);
    }
    if (rc) { if (cs->xbuf) hipFree(cs->xbuf); delete cs; return rc; }
    c->comm = cs;               // not usable before dopf_xchg_init (world of the view is 0 until then)
    return DOPF_OK;
}

int dopf_xchg_init(dopf_ctx *c, int32_t world, int32_t rank, const void *handles)
{
    if (!c || !handles || !c->comm || !c->comm->p2p || c->comm->world != world || rank < 0 || rank >= world || c->comm->xv.world)
        return fail(c, DOPF_E_INVALID, "bad argument (dopf_xchg_export first, same world)");
    DeviceGuard guard(c->device);
    dopf_comm_state *cs = c->comm;
    cs->rank = rank;
    void *areas[kXchgMaxWorld] = {nullptr};
    for (int r = 0; r < world; ++r) {
        if (r == rank) { areas[r] = cs->xbuf; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)r * DOPF_XCHG_HANDLE_BYTES, sizeof h);
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(c, DOPF_E_DEVICE, "hipIpcOpenMemHandle (rank %d): %s", r, hipGetErrorString(e));
        cs->opened.push_back(p);
        areas[r] = p;
    }
    xchg_fill_view(c, cs, world, rank, areas);
    // rendezvous: say hello in every peer's area, wait until every peer has said hello here — from then on all areas
    // are mapped everywhere and the ranks are at most a host call apart
    // The hello word carries the rank's iteration count (+1): the exchange's sequence numbers and slot parities are
    // derived from it, so ranks that join with different counts would read each other's other slot, whose older flag
    // already passes — refused here instead.
    if (int rcs = read_status(c)) { xchg_undo(c, cs); return rcs; }
    const unsigned long long one = (unsigned long long)c->host_st.iters_total + 1ull;
    constexpr unsigned long long kPhase2 = 1ull << 62;          // second rendezvous: the lines' reach has been handed round
    double wait_s = 120.0;
    if (const char *e = exp_env("DOPF_XCHG_HELLO_S")) wait_s = atof(e);
    const auto t0 = std::chrono::steady_clock::now();
    auto say = [&](unsigned long long word) -> int {
        for (int r = 0; r < world; ++r) {
            const hipError_t e = hipMemcpy((char *)areas[r] + (size_t)rank * sizeof word, &word, sizeof word, hipMemcpyHostToDevice);
            if (e != hipSuccess) { xchg_undo(c, cs); return fail(c, DOPF_E_DEVICE, "peer exchange: hello to rank %d: %s", r, hipGetErrorString(e)); }
        }
        return DOPF_OK;
    };
    auto wait_all = [&](unsigned long long need_bits) -> int {      // every rank's word is there (and carries need_bits)
        for (;;) {
            unsigned long long hello[kXchgMaxWorld];
            const hipError_t e = hipMemcpy(hello, cs->xbuf, sizeof hello, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { xchg_undo(c, cs); return fail(c, DOPF_E_DEVICE, "peer exchange: reading the hello words: %s", hipGetErrorString(e)); }
            int seen = 0;
            for (int r = 0; r < world; ++r) seen += hello[r] != 0ull && (hello[r] & need_bits) == need_bits;
            if (seen == world) {
                for (int r = 0; r < world; ++r)
                    if ((hello[r] & ~kPhase2) != one) {
                        xchg_undo(c, cs);
                        return fail(c, DOPF_E_INVALID, "peer exchange: rank %d joins after %llu iterations, this rank after %llu — the ranks "
                                    "must join with the same iteration count", r, (hello[r] & ~kPhase2) - 1ull, one - 1ull);
                    }
                return DOPF_OK;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) {
                xchg_undo(c, cs);
                return fail(c, DOPF_E_DEVICE, "peer exchange: %d of %d ranks showed up within %.0f s", seen, world, wait_s);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    };
    if (int rc = say(one)) return rc;
    if (int rc = wait_all(0ull)) return rc;
    // Networks: which lines' slack sums need the agents one by one ("flagged") is decided from each line's REACH — the largest
    // move any agent of any node can make, seen through the line — and a rank knows only its own agents. The ranks hand their
    // reaches round (through the receive areas just mapped, before any iteration uses them) and keep the elementwise maximum:
    // from here on the flags, and with them the choice between the chain with k_reduce and the one without
    // (DevView::slackGlobal), are the same replicated state on every rank as the duals they are computed from.
    if (c->v.L > 0) {
        const XchgLayout lay((size_t)dopf_consensus_size(c), world);
        const size_t L = (size_t)c->v.L;
        for (int r = 0; r < world; ++r) {
            const hipError_t e = hipMemcpy((char *)areas[r] + lay.data_off + (size_t)rank * lay.n * sizeof(double), c->v.line_reach,
                                           L * sizeof(double), hipMemcpyDeviceToDevice);
            if (e != hipSuccess) { xchg_undo(c, cs); return fail(c, DOPF_E_DEVICE, "peer exchange: reach to rank %d: %s", r, hipGetErrorString(e)); }
        }
        if (hipDeviceSynchronize() != hipSuccess) { xchg_undo(c, cs); return fail(c, DOPF_E_DEVICE, "peer exchange: reach hand-over did not complete"); }
        if (int rc = say(one | kPhase2)) return rc;
        if (int rc = wait_all(kPhase2)) return rc;
        std::vector<double> mx(L, 0.0), got(L);
        for (int r = 0; r < world; ++r) {
            const hipError_t e = hipMemcpy(got.data(), (char *)cs->xbuf + lay.data_off + (size_t)r * lay.n * sizeof(double), L * sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { xchg_undo(c, cs); return fail(c, DOPF_E_DEVICE, "peer exchange: reading rank %d's reach: %s", r, hipGetErrorString(e)); }
            for (size_t l = 0; l < L; ++l) mx[l] = std::max(mx[l], got[l]);
        }
        const hipError_t e = hipMemcpy(const_cast<double *>(c->v.line_reach), mx.data(), L * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { xchg_undo(c, cs); return fail(c, DOPF_E_DEVICE, "peer exchange: storing the common reach: %s", hipGetErrorString(e)); }
        // (the slots used here are overwritten by the first iteration's exchange; their flags were never set)
    }
    drop_graphs(c);
    return DOPF_OK;
}

int dopf_comm_unique_id(void *id128)
{
    if (!id128) return DOPF_E_INVALID;
    const Rccl *r = rccl();
    if (!r) return fail(nullptr, g_rccl.code, "%s", g_rccl.err);
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return fail(nullptr, DOPF_E_DEVICE, "ncclGetUniqueId: %s", r->GetErrorString(e));
    static_assert(sizeof id == DOPF_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof id);
    return DOPF_OK;
}

int dopf_comm_init(dopf_ctx *c, int32_t world, int32_t rank, const void *id128)
{
    if (!c || world < 1 || rank < 0 || rank >= world || !id128) return fail(c, DOPF_E_INVALID, "bad argument");
    if (c->comm) return fail(c, DOPF_E_INVALID, "context already has a communicator");
    const Rccl *r = rccl();
    if (!r) return fail(c, g_rccl.code, "%s", g_rccl.err);
    DeviceGuard guard(c->device);
    HIPCHK(c, hipStreamSynchronize(c->main));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    dopf_comm_state *cs = new (std::nothrow) dopf_comm_state;
    if (!cs) return fail(c, DOPF_E_NOMEM, "out of host memory");
    cs->world = world; cs->rank = rank;
    const ncclResult_t e = r->CommInitRank(&cs->comm, world, id, rank);
    if (e != ncclSuccess) { delete cs; return fail(c, DOPF_E_DEVICE, "ncclCommInitRank: %s", r->GetErrorString(e)); }
    c->comm = cs;
    drop_graphs(c);          // the chain changes: the next dopf_iterate captures it anew
    return DOPF_OK;
}

int dopf_comm_info(const dopf_ctx *c, int32_t *world, int32_t *rank, int32_t *in_graph)
{
    if (!c) return DOPF_E_INVALID;
    if (world) *world = c->comm ? c->comm->world : 1;
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    if (in_graph) *in_graph = (c->comm && c->graphs_valid) ? 1 : 0;
    return DOPF_OK;
}

const char *dopf_multi_last_error(const dopf_multi *m) { return m ? m->err : g_multi_err; }

void dopf_multi_destroy(dopf_multi *m)
{
    if (!m) return;
    for (dopf_ctx *c : m->ctx) dopf_destroy(c);
    delete m;
}

int dopf_multi_create(dopf_multi **out, const dopf_problem *p, const dopf_params *q, int32_t n_gpus, const int32_t *devices)
{
    if (!out || !p || !q || n_gpus < 1) return mfail(nullptr, DOPF_E_INVALID, "bad argument");
    *out = nullptr;
    const bool host_sum = (q->flags & DOPF_F_COMM_HOST) != 0;
    const bool p2p = !host_sum && (q->flags & DOPF_F_COMM_P2P) != 0;
    if (p2p && n_gpus > kXchgMaxWorld) return mfail(nullptr, DOPF_E_INVALID, "peer exchange: too many shards");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return mfail(nullptr, DOPF_E_DEVICE, "no HIP device; libdopf_hip has no CPU fallback");
    std::vector<int> dev(n_gpus);
    for (int i = 0; i < n_gpus; ++i) {
        dev[i] = devices ? devices[i] : (host_sum ? i % ndev : i);
        if (dev[i] < 0 || dev[i] >= ndev) return mfail(nullptr, DOPF_E_INVALID, "device ordinal out of range (n_gpus exceeds the visible devices?)");
        if (!host_sum && !p2p)
            for (int j = 0; j < i; ++j)
                if (dev[j] == dev[i]) return mfail(nullptr, DOPF_E_INVALID, "RCCL needs one distinct device per shard (DOPF_F_COMM_HOST lifts this for tests)");
    }
    const Rccl *r = nullptr;
    if (!host_sum && !p2p && n_gpus > 1) {
        r = rccl();
        if (!r) return mfail(nullptr, g_rccl.code, g_rccl.err);
    }
    dopf_multi *m = new (std::nothrow) dopf_multi;
    if (!m) return mfail(nullptr, DOPF_E_NOMEM, "out of host memory");
    m->n = n_gpus; m->G = p->G; m->S = p->S; m->T = p->T; m->host_sum = host_sum;
    m->ctx.assign(n_gpus, nullptr);
    m->g0.resize(n_gpus); m->g1.resize(n_gpus); m->s0.resize(n_gpus); m->s1.resize(n_gpus);
    // contiguous slices of the caller's agent lists (the consensus sum does not care which agents a shard holds;
    // each context sorts its own slice by node)
    auto cut = [&](int total, int i, int &lo, int &hi) {
        const int base = total / n_gpus, rem = total % n_gpus;
        lo = i * base + std::min(i, rem);
        hi = lo + base + (i < rem ? 1 : 0);
    };
    for (int i = 0; i < n_gpus; ++i) {
        cut(p->G, i, m->g0[i], m->g1[i]);
        cut(p->S, i, m->s0[i], m->s1[i]);
        dopf_problem pi = *p;
        pi.G = m->g1[i] - m->g0[i]; pi.S = m->s1[i] - m->s0[i];
        pi.gen_mc = p->gen_mc ? p->gen_mc + m->g0[i] : nullptr; pi.gen_pmax = p->gen_pmax ? p->gen_pmax + m->g0[i] : nullptr;
        pi.gen_node = p->gen_node ? p->gen_node + m->g0[i] : nullptr;
        pi.sto_mc = p->sto_mc ? p->sto_mc + m->s0[i] : nullptr; pi.sto_pmax = p->sto_pmax ? p->sto_pmax + m->s0[i] : nullptr;
        pi.sto_emax = p->sto_emax ? p->sto_emax + m->s0[i] : nullptr; pi.sto_node = p->sto_node ? p->sto_node + m->s0[i] : nullptr;
        dopf_params qi = *q;
        qi.device = dev[i];
        qi.stream = nullptr;
        qi.n_agents_global = q->n_agents_global > 0 ? q->n_agents_global : p->G + p->S;
        const int rc = dopf_create(&m->ctx[i], &pi, &qi);
        if (rc) {
            snprintf(g_multi_err, 512, "shard %d: %s", i, dopf_last_error(nullptr));
            dopf_multi_destroy(m);
            return rc;
        }
    }
    if (p2p) {
        // every shard's receive area is directly addressable from every other shard's device
        void *areas[kXchgMaxWorld] = {nullptr};
        for (int i = 0; i < n_gpus; ++i) {
            dopf_ctx *c = m->ctx[i];
            DeviceGuard guard(c->device);
            dopf_comm_state *cs = new dopf_comm_state;
            cs->world = n_gpus; cs->rank = i; cs->p2p = true;
            c->comm = cs;
            int rc = xchg_alloc(c, cs, n_gpus);
            for (int j = 0; j < n_gpus && rc == DOPF_OK; ++j) {
                if (dev[j] == dev[i]) continue;
                int can = 0;
                hipDeviceCanAccessPeer(&can, dev[i], dev[j]);
                if (!can) { rc = fail(c, DOPF_E_DEVICE, "device %d cannot address device %d", dev[i], dev[j]); break; }
                const hipError_t e = hipDeviceEnablePeerAccess(dev[j], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = fail(c, DOPF_E_DEVICE, "hipDeviceEnablePeerAccess: %s", hipGetErrorString(e));
                (void)hipGetLastError();
            }
            if (rc) {
                snprintf(g_multi_err, 512, "shard %d: %s", i, dopf_last_error(c));
                dopf_multi_destroy(m);
                return rc;
            }
            areas[i] = cs->xbuf;
        }
        for (int i = 0; i < n_gpus; ++i) xchg_fill_view(m->ctx[i], m->ctx[i]->comm, n_gpus, i, areas);
        // the lines' common reach (see dopf_xchg_init: the flags it decides must be the same replicated state on every shard)
        if (p->L > 0) {
            const size_t L = (size_t)p->L;
            std::vector<double> mx(L, 0.0), got(L);
            bool ok = true;
            for (int i = 0; i < n_gpus && ok; ++i) {
                DeviceGuard guard(m->ctx[i]->device);
                ok = hipMemcpy(got.data(), m->ctx[i]->v.line_reach, L * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
                for (size_t l = 0; l < L; ++l) mx[l] = std::max(mx[l], got[l]);
            }
            for (int i = 0; i < n_gpus && ok; ++i) {
                DeviceGuard guard(m->ctx[i]->device);
                ok = hipMemcpy(const_cast<double *>(m->ctx[i]->v.line_reach), mx.data(), L * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
            }
            if (!ok) {
                snprintf(g_multi_err, 512, "peer exchange: the lines' common reach could not be set");
                dopf_multi_destroy(m);
                return DOPF_E_DEVICE;
            }
        }
    } else if (n_gpus > 1 || host_sum) {
        std::vector<ncclComm_t> comms(n_gpus, nullptr);
        if (!host_sum) {
            const ncclResult_t e = r->CommInitAll(comms.data(), n_gpus, dev.data());
            if (e != ncclSuccess) {
                snprintf(g_multi_err, 512, "ncclCommInitAll: %s", r->GetErrorString(e));
                dopf_multi_destroy(m);
                return DOPF_E_DEVICE;
            }
        }
        for (int i = 0; i < n_gpus; ++i) {
            dopf_comm_state *cs = new dopf_comm_state;
            cs->comm = comms[i]; cs->world = n_gpus; cs->rank = i; cs->host_sum = host_sum;
            m->ctx[i]->comm = cs;
        }
        if (host_sum) {
            const size_t n = (size_t)dopf_consensus_size(m->ctx[0]);
            m->hsum.assign(n, 0.0); m->hpart.assign(n, 0.0);
        }
    }
    *out = m;
    return DOPF_OK;
}

int32_t dopf_multi_size(const dopf_multi *m) { return m ? m->n : 0; }

dopf_ctx *dopf_multi_ctx(dopf_multi *m, int32_t i) { return (m && i >= 0 && i < m->n) ? m->ctx[i] : nullptr; }

int dopf_multi_iterate(dopf_multi *m, int32_t n_iters, int32_t *iters_done, int32_t *converged)
{
    if (!m || n_iters < 0) return mfail(m, DOPF_E_INVALID, "bad argument");
    std::vector<int32_t> done(m->n, 0), conv(m->n, 0);
    int rc = DOPF_OK;
    if (!m->host_sum) {
        // every shard replays its own graph (kernels + its rank's part of the collective) from its own thread
        rc = for_each_shard(m, [&](int i) { return dopf_iterate(m->ctx[i], n_iters, &done[i], &conv[i]); });
    } else {
        // debugging transport: local sums -> host -> sum in shard order -> every shard -> dual step
        const size_t n = m->hsum.size();
        int32_t it = 0, cv = 0;
        rc = dopf_sync(m->ctx[0], &it, &cv);
        const int32_t before = m->ctx[0]->host_st.iters_total;
        for (int k = 0; k < n_iters && rc == DOPF_OK && !m->ctx[0]->host_st.halt; ++k) {
            std::fill(m->hsum.begin(), m->hsum.end(), 0.0);
            for (int i = 0; i < m->n && rc == DOPF_OK; ++i) {
                dopf_ctx *c = m->ctx[i];
                DeviceGuard guard(c->device);
                rc = dopf_local_update(c);
                if (rc) break;
                if (hipMemcpyAsync(m->hpart.data(), c->v.cons, n * sizeof(double), hipMemcpyDeviceToHost, c->main) != hipSuccess ||
                    hipStreamSynchronize(c->main) != hipSuccess) { rc = fail(c, DOPF_E_DEVICE, "consensus download failed"); break; }
                for (size_t j = 0; j < n; ++j) m->hsum[j] += m->hpart[j];
            }
            for (int i = 0; i < m->n && rc == DOPF_OK; ++i) {
                dopf_ctx *c = m->ctx[i];
                DeviceGuard guard(c->device);
                if (hipMemcpyAsync(c->v.cons, m->hsum.data(), n * sizeof(double), hipMemcpyHostToDevice, c->main) != hipSuccess) { rc = fail(c, DOPF_E_DEVICE, "consensus upload failed"); break; }
                rc = dopf_apply_consensus(c);
            }
            if (rc == DOPF_OK) rc = dopf_sync(m->ctx[0], &it, &cv);
        }
        for (int i = 0; i < m->n && rc == DOPF_OK; ++i) rc = dopf_sync(m->ctx[i], &it, &conv[i]);
        done[0] = m->ctx[0]->host_st.iters_total - before;
        if (rc) snprintf(m->err, 512, "%s", dopf_last_error(m->ctx[0]));
    }
    if (rc) return rc;
    // every shard computed the same stop test from the same sums
    for (int i = 1; i < m->n; ++i)
        if (conv[i] != conv[0] || m->ctx[i]->host_st.iteration != m->ctx[0]->host_st.iteration)
            return mfail(m, DOPF_E_DEVICE, "shards disagree on the iteration state (consensus sum not identical on all ranks)");
    if (iters_done) *iters_done = done[0];
    if (converged) *converged = conv[0];
    return DOPF_OK;
}

int dopf_multi_get_primal(dopf_multi *m, double *P, double *D, double *C, double *E)
{
    if (!m) return DOPF_E_INVALID;
    const size_t T = (size_t)m->T;
    for (int i = 0; i < m->n; ++i) {
        const int rc = dopf_get_primal(m->ctx[i], P ? P + T * m->g0[i] : nullptr, D ? D + T * m->s0[i] : nullptr,
                                       C ? C + T * m->s0[i] : nullptr, E ? E + T * m->s0[i] : nullptr);
        if (rc) { snprintf(m->err, 512, "shard %d: %s", i, dopf_last_error(m->ctx[i])); return rc; }
    }
    return DOPF_OK;
}

}  // extern "C"
