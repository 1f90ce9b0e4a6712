// dopf_ctx.h — the context behind the C ABI, shared by dopf_api.hip and dopf_comm.hip (internal).
#pragma once

#include <string>
#include <vector>

#include "dopf_internal.h"

struct dopf_comm_state;     // dopf_comm.hip: RCCL communicator of a sharded context

struct dopf_ctx {
    dopf::DevView v{};
    dopf::Launch lc{};
    dopf_params q{};
    int device = 0;
    hipStream_t main = nullptr, side = nullptr;
    bool own_main = false;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipEvent_t evT0 = nullptr, evT1 = nullptr;      // DOPF_F_TIME_CALLS: around the launches of the last dopf_iterate
    double last_call_ms = -1.0;
    hipGraphExec_t graph1 = nullptr, graphM = nullptr, graphU = nullptr;   // 1, kMid, kUnroll iterations per launch
    bool graphs_valid = false;
    // Networks on the three-launch chain: while no line is flagged (Status::walk_last == 0 at the last look) the "quiet" chain runs —
    // k_net_agents and the dual/price kernel, which forms the node sums itself; k_slack is not launched. Graphs of its own.
    hipGraphExec_t graph1q = nullptr, graphMq = nullptr, graphUq = nullptr;
    bool graphs_q_valid = false;
    bool quiet_ok = false;          // the problem and the flags allow the quiet chain
    bool comm_quiet_ok = false;     // ... and its form on a peer exchange (k_slack, exchange of the node sums, dual/price kernel: no k_reduce)
    bool quiet = false;             // the next launches may use it
    unsigned long long quiet_parked = 0;    // times the quiet chain parked itself (a line got flagged) and the host went back
    std::vector<void *> allocs;
    void *own_cons = nullptr;
    double *getter_scratch = nullptr;      // 3 * N * T doubles, allocated at the first getter that needs them (freed with the context)
    std::vector<int> gen_perm, sto_perm;   // sorted position -> caller's index
    dopf::Status host_st{};
    dopf::Status *host_pin = nullptr;       // page-locked landing area of the status read-back (a pageable target is staged: slower)
    unsigned long long solver_fail_seen = 0;   // failures already reported through DOPF_E_SOLVER
    dopf_comm_state *comm = nullptr;       // non-null: dopf_iterate runs local_update -> all-reduce -> apply_consensus
    bool tail_xchg = false;                // peer exchange inside the tail block of the one-launch iteration (copper plates)
    bool level_from_primal = true;         // dopf_get_primal rebuilds E = cumsum(C - D); false while a central solve's own levels are in v.E
    char err[512] = {0};
};

namespace dopf {

int fail(dopf_ctx *c, int code, const char *fmt, ...);
void keep_error(const dopf_ctx *c);          // the context's message becomes what dopf_last_error(NULL) returns
void enqueue_local(dopf_ctx *c, bool single, bool quiet = false, bool comm_quiet = false);
void enqueue_apply(dopf_ctx *c, bool single, const XchgView *xd = nullptr, bool quiet = false, bool comm_quiet = false);
void drop_graphs(dopf_ctx *c);
int read_status(dopf_ctx *c);
// dopf_comm.hip
int check_one_runtime(dopf_ctx *c);           // DOPF_E_UNSUPPORTED when two HIP runtimes are mapped into the process
int comm_enqueue_allreduce(dopf_ctx *c);      // sum of the consensus buffer over the ranks, on the context's stream
void comm_release(dopf_ctx *c);
int comm_world(const dopf_ctx *c);             // ranks of the context's communicator (1 without one)
const XchgView *comm_xchg(const dopf_ctx *c);   // the initialised peer exchange of the context, or null
bool comm_capturable(const dopf_ctx *c);      // the chain incl. its consensus sum may go into a hipGraph by default

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { hipGetDevice(&prev); if (dev != prev) hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return dopf::fail((c), DOPF_E_DEVICE, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

}  // namespace dopf
