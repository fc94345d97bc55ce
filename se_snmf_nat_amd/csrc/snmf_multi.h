// snmf_multi.h -- multi-GPU solves behind the C ABI (include/snmf.h: snmf_multi_*), included by snmf_api.hip.
//
// One PROCESS, several devices: the reference's host is a single MATLAB interpreter (run_basis_DNMF.m:36-55 calls
// sparse_nmf from one thread), so the MEX shim cannot bring a process-per-GPU launcher with it; this is the entry that
// puts the 8-GPU path behind "the same function signature" (SURVEY.md section 8b(1), 8e).  The process-per-GPU path over
// RCCL stays in se_snmf_nat_amd/dist.py (bench.py --gpus N).
//
// Frames are independent given W (src/sparse_nmf.m:189-208): rank g owns the contiguous column block
// [col[g], col[g+1]) of V and H on device devices[g] as an ordinary snmf_plan; W is replicated.  One host thread per
// rank issues that rank's launches.  Per iteration there is ONE exchange, the ONE-SHOT all-reduce SURVEY.md section 5 / 8e
// asks for instead of a ring (the message is ~0.5 MB: latency, not bytes, is the enemy on the fully connected xGMI
// mesh):
//     k_push_stats  every rank writes its fp64 statistics into slot g of EVERY rank's gather buffer (peer stores)
//     ordering      (see below)
//     k_sum_ranks   every rank adds the n slots of its own gather buffer in rank order
// so all ranks hold bit-identical sums, run the identical deterministic W epilogue and take identical stop decisions.
// Gather buffers (and events / flags) are double-buffered by iteration parity: rank g can only push iteration j+2 after
// it has seen every peer's push of j+1, which each peer issued after its own sum of iteration j.
// H-only solves exchange just the two cost scalars at the tail of the buffer.
//
// Ordering of push and sum, two modes (snmf_multi_set_exchange):
//   FLAGS  (round 3; the default when every rank has a device of its own): ordered ON THE DEVICES.  The last workgroup
//          of k_push_stats -- after a system-scope fence behind everybody's stores -- writes the exchange's sequence
//          number into this rank's arrival word on every peer; k_sum_ranks polls its n arrival words (system-scope
//          acquire loads, bounded by wall time: a lost peer raises the plan's fault word, never a hang) and only then
//          reads the slots.  Gather buffers and arrival words are fine-grained device memory, so a peer's stores are
//          visible without relying on what an L2 does at a kernel boundary.  A rank's host thread only ENQUEUES: two
//          launches per exchange, no host barrier, no event calls.
//   EVENTS (round 2; the default when ranks share a device, i.e. single-GPU testing): an event per rank and parity, a
//          host barrier so that every peer has ISSUED its record before anybody waits on it, n-1 cross-stream waits.
//          On a shared device FLAGS by itself could deadlock until its time-out: streams that land on the same hardware
//          queue run in submission order, and a polling kernel would then sit in front of the push it polls for (seen
//          once in round 3: the mapping of streams to queues changes from run to run).  FLAGS forced on a device list
//          with repeats (the single-GPU tests of the flag path) therefore keeps ONE host barrier per exchange between
//          the ranks' push and sum submissions: every push then precedes every sum in every queue.
#pragma once

#include <atomic>
#include <thread>
#include <system_error>

namespace snmf {

struct PushArgs {
    const double* src;
    double* dst[16];
    unsigned* flag[16];   // FLAGS mode: this rank's arrival word on every peer (nullptr: EVENTS mode)
    unsigned* done_ctr;   // FLAGS mode: workgroups of this launch that have finished (on this rank's device, zero at launch)
    size_t len;
    int n;
    unsigned seq;         // sequence number of the exchange (>= 1)
};
__global__ __launch_bounds__(256) void k_push_stats(PushArgs a) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.len; i += (size_t)gridDim.x * 256) {
        const double v = a.src[i];
        for (int q = 0; q < a.n; ++q) a.dst[q][i] = v;
    }
    if (a.done_ctr) {
        __threadfence_system();  // this thread's peer stores are performed before it reports
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1) {  // the last workgroup: every store of the launch is out -> announce on every peer
                __hip_atomic_store(a.done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                for (int q = 0; q < a.n; ++q) __hip_atomic_store(a.flag[q], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
// out[i] = slot_0[i] + slot_1[i] + ... in rank order (identical on every rank).
// flags != nullptr (FLAGS mode): first wait until every rank's arrival word has reached `seq`.  The wait is bounded by
// WALL time (s_memrealtime, 100 MHz): a peer that never arrives raises *fault and the sums are garbage the host rejects.
__global__ __launch_bounds__(256) void k_sum_ranks(const double* __restrict__ slots, int n, size_t len, double* __restrict__ out,
                                                   const unsigned* flags, unsigned seq, int* fault) {
    if (flags) {
        // (a time-out is sticky: once the fault word is up nobody waits again, so a lost peer costs one time-out, not one
        //  per remaining iteration, and the host finds the fault at its next look at the plan's state)
        if ((int)threadIdx.x < n && !__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while ((int)(__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 500000000ull) {  // 5 s
                    atomicExch(fault, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (system scope: nothing read below may come from a line cached before the arrival)
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256) {
        double s = slots[i];
        for (int q = 1; q < n; ++q) s += slots[(size_t)q * len + i];
        out[i] = s;
    }
}

}  // namespace snmf

// What a device list owns for as long as the process runs (round 5): the ranks' contexts (stream, transfer pipeline, cached device
// blocks) with a second context each (uploads under a running solve), the fine-grained gather buffers and arrival words, the
// events, the peer-access grants.  Creating these per CALL (round 4: every snmf_sparse_nmf_multi_* call, three times per
// run_basis_DNMF) cost milliseconds of hipExtMalloc / hipHostMalloc / stream creation per rank; a team is built once per
// device list, handed to one user at a time (team_acquire / team_release) and kept when released.  The handles of one user
// (the three solves of snmf_run_basis_dnmf_multi_*) may SHARE a team as long as they never run at the same time: exchange
// number and parity live in the team, so consecutive exchanges of different handles never reuse an arrival value.
struct MultiTeam {
    int n = 0;
    std::vector<int> dev;
    std::vector<snmf_ctx*> ctx;       // [n] (ctx[g]->aux: the second context of rank g's device)
    std::vector<double*> slots;       // [n] gather buffers, slots_cap doubles each (fine-grained device memory)
    size_t slots_cap = 0;
    std::vector<unsigned*> flags;     // [n] arrival words: [2 parities][n ranks], then the push launch's workgroup counter
    std::vector<hipEvent_t> ev[2];    // ev[parity][rank]: "rank has pushed"  (EVENTS mode)
    bool fine_grained = true;         // every gather buffer / arrival word is fine-grained (else only EVENTS ordering is safe)
    bool shared_dev = false;          // two ranks on one device
    unsigned xseq = 0;                // exchanges issued so far (the next one is number xseq + 1)
    int par = 0;                      // parity of the next exchange
    int refs = 0;                     // 0: idle in the cache
    // A run on this team ended in a failure (a rank's call failed, a rank left its loop early, a bounded device-side wait gave
    // up): exchange number, parity and the on-device arrival words may no longer agree between the ranks -- a peer may have
    // posted arrival values AHEAD of the number the failed rank recorded, and the next handle's exchanges would accept them as
    // their own and sum zeroed or stale slots.  A poisoned team is destroyed when its last user lets go, never cached.
    bool poisoned = false;
};

struct snmf_multi {
    int n = 0;
    snmf_params p{};
    MultiTeam* team = nullptr;
    bool team_owner = true;           // this handle took the team out of the cache (and gives it back)
    bool use_aux = false;             // the ranks' plans live on the second contexts
    std::vector<int> dev;
    std::vector<int64_t> col;         // n + 1 column offsets
    std::vector<snmf_ctx*> ctx;       // [n] borrowed from the team (ctx or aux)
    std::vector<snmf_plan*> plan;
    std::vector<double*> stats;       // [n] device statistics buffer of each rank
    int mode = 0;                     // SNMF_EXCHANGE_FLAGS / _EVENTS (resolved from AUTO at creation)
    bool shared_dev = false;          // two ranks on one device: submissions of push and sum are ordered by the host in every mode
    // every gather buffer and arrival word is fine-grained (coherent) device memory.  If the runtime refused one of them the
    // buffers are coarse-grained: a kernel that POLLS them while peers write (FLAGS) could see the flag and still read stale
    // slot lines from its L2, so only EVENTS ordering -- a kernel boundary between push and sum -- is allowed then.
    bool fine_grained = true;
    std::vector<uint8_t> w_ind, h_ind;
    size_t len = 0, xoff = 0, xlen = 0;  // statistics length; the exchanged part [xoff, xoff + xlen)
    bool upd_w = true, can_stop = false;
    int it = 0;                       // iterations issued
    bool inited = false, finalized = false, stopped = false;
    // host barrier of the rank threads (sense reversing)
    std::atomic<int> bar_count{0};
    std::atomic<int> bar_gen{0};
    std::atomic<int> failed_at{0x7fffffff};  // number of the first barrier some rank entered as failed
    std::vector<int> rc;
    std::vector<std::string> err;
};

// Host barrier of the rank threads that also AGREES on failure: a rank publishes "I have failed" only on its way INTO
// barrier number `seq` (as the smallest such number), and everybody reads the word only on the way OUT, comparing it
// with the number of the barrier just passed -- so a fast rank that fails and publishes at barrier seq+1 cannot make a
// slow rank, still reading after barrier seq, leave the loop one barrier early (it would never arrive at seq+1).
// Returns true when every rank was healthy up to this barrier.
static bool multi_barrier(snmf_multi* m, int seq, bool i_failed) {
    if (i_failed) {
        int cur = m->failed_at.load(std::memory_order_relaxed);
        while (seq < cur && !m->failed_at.compare_exchange_weak(cur, seq, std::memory_order_relaxed)) {}
    }
    const int gen = m->bar_gen.load(std::memory_order_acquire);
    if (m->bar_count.fetch_add(1, std::memory_order_acq_rel) == m->n - 1) {
        m->bar_count.store(0, std::memory_order_relaxed);
        m->bar_gen.store(gen + 1, std::memory_order_release);
    } else {
        while (m->bar_gen.load(std::memory_order_acquire) == gen) std::this_thread::yield();
    }
    return m->failed_at.load(std::memory_order_acquire) > seq;
}

// ---- teams ---------------------------------------------------------------------------------------------------------
static std::mutex g_team_mu;
static std::vector<MultiTeam*> g_teams;   // every team of the process (idle ones have refs == 0)
// Idle teams kept for the next handle on the same device list (SNMF_TEAM_CACHE overrides; 0 = none).  A rank of a team that has
// moved host arrays holds its contexts' bounce buffers (2 x 48 MiB pinned + as much device staging), so the cache is kept short.
static int max_idle_teams() {
    const char* e = getenv("SNMF_TEAM_CACHE");
    return e ? std::max(0, atoi(e)) : 2;
}

static void team_destroy(MultiTeam* t) {
    for (int g = 0; g < (int)t->ctx.size(); ++g) {
        if (!t->ctx[g]) continue;
        hipSetDevice(t->dev[g]);
        hipStreamSynchronize(t->ctx[g]->stream);
        for (int q = 0; q < 2; ++q)
            if (g < (int)t->ev[q].size() && t->ev[q][g]) hipEventDestroy(t->ev[q][g]);
        if (g < (int)t->slots.size() && t->slots[g]) hipFree(t->slots[g]);
        if (g < (int)t->flags.size() && t->flags[g]) hipFree(t->flags[g]);
        snmf_ctx_destroy(t->ctx[g]);
    }
    delete t;
}

static int team_create(const int32_t* devices, int n_dev, MultiTeam** out) {
    MultiTeam* t = new MultiTeam();
    t->n = n_dev;
    t->dev.assign(devices, devices + n_dev);
    t->ctx.assign(n_dev, nullptr);
    t->slots.assign(n_dev, nullptr);
    t->flags.assign(n_dev, nullptr);
    t->ev[0].assign(n_dev, nullptr);
    t->ev[1].assign(n_dev, nullptr);
    for (int g = 0; g < n_dev; ++g)
        for (int q = 0; q < g; ++q) t->shared_dev = t->shared_dev || devices[g] == devices[q];
    int s = SNMF_OK;
    for (int g = 0; g < n_dev && s == SNMF_OK; ++g) {
        s = snmf_ctx_create(&t->ctx[g], t->dev[g]);
        if (s == SNMF_OK) s = snmf_ctx_create(&t->ctx[g]->aux, t->dev[g]);
    }
    for (int g = 0; g < n_dev && s == SNMF_OK; ++g) {
        if (hipSetDevice(t->dev[g]) != hipSuccess) s = fail(SNMF_ERR_NO_DEVICE, "hipSetDevice(%d)", t->dev[g]);
        // arrival words are written by PEERS while this device may hold lines of them: fine-grained (coherent) device memory;
        // plain hipMalloc only if the runtime refuses (then kernel boundaries must do)
        const size_t fb = ((size_t)2 * n_dev + 1) * sizeof(unsigned);
        if (s == SNMF_OK && hipExtMallocWithFlags((void**)&t->flags[g], fb, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            t->fine_grained = false;
            s = dalloc(&t->flags[g], (size_t)2 * n_dev + 1);
        }
        if (s == SNMF_OK) hipMemset(t->flags[g], 0, fb);
        for (int q = 0; q < 2 && s == SNMF_OK; ++q)
            if (hipEventCreateWithFlags(&t->ev[q][g], hipEventDisableTiming) != hipSuccess)
                s = fail(SNMF_ERR_NO_DEVICE, "hipEventCreate failed on device %d", t->dev[g]);
        // peer stores into every other device's gather buffer
        for (int q = 0; q < n_dev && s == SNMF_OK; ++q) {
            if (t->dev[q] == t->dev[g]) continue;
            int can = 0;
            hipDeviceCanAccessPeer(&can, t->dev[g], t->dev[q]);
            if (!can) {
                s = fail(SNMF_ERR_UNSUPPORTED, "device %d cannot access device %d as a peer", t->dev[g], t->dev[q]);
                break;
            }
            const hipError_t e = hipDeviceEnablePeerAccess(t->dev[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                s = fail(SNMF_ERR_NO_DEVICE, "hipDeviceEnablePeerAccess(%d -> %d): %s", t->dev[g], t->dev[q], hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    if (s != SNMF_OK) {
        const std::string keep = g_err;
        team_destroy(t);
        g_err = keep;
        return s;
    }
    *out = t;
    return SNMF_OK;
}

// gather buffers for exchanges of xlen doubles per rank (grown, never shrunk; contents are zeroed by snmf_multi_init)
static int team_reserve(MultiTeam* t, size_t xlen) {
    const size_t need = (size_t)2 * t->n * xlen;
    if (need <= t->slots_cap) return SNMF_OK;
    for (int g = 0; g < t->n; ++g) {
        if (hipSetDevice(t->dev[g]) != hipSuccess) return fail(SNMF_ERR_NO_DEVICE, "hipSetDevice(%d)", t->dev[g]);
        hipStreamSynchronize(t->ctx[g]->stream);
        hipStreamSynchronize(t->ctx[g]->aux->stream);
        if (t->slots[g]) hipFree(t->slots[g]);
        t->slots[g] = nullptr;
        t->slots_cap = 0;  // (a failure below leaves some ranks without a buffer: the next handle starts over)
        if (hipExtMallocWithFlags((void**)&t->slots[g], need * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            t->fine_grained = false;  // (coarse-grained memory: only EVENTS ordering is safe, see snmf_multi_set_exchange)
            SN_TRY(dalloc(&t->slots[g], need));
        }
    }
    t->slots_cap = need;
    return SNMF_OK;
}

static int team_acquire(const int32_t* devices, int n_dev, MultiTeam** out) {
    {
        std::lock_guard<std::mutex> lk(g_team_mu);
        for (MultiTeam* t : g_teams)
            if (t->refs == 0 && t->n == n_dev && std::equal(t->dev.begin(), t->dev.end(), devices)) {
                t->refs = 1;
                *out = t;
                return SNMF_OK;
            }
    }
    MultiTeam* t = nullptr;
    SN_TRY(team_create(devices, n_dev, &t));
    t->refs = 1;
    std::lock_guard<std::mutex> lk(g_team_mu);
    g_teams.push_back(t);
    *out = t;
    return SNMF_OK;
}
static void team_release(MultiTeam* t) {
    MultiTeam* kill = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_team_mu);
        if (--t->refs > 0) return;
        // most recently used last; past the limit the LEAST recently used idle team goes (with a limit of 0: this one)
        g_teams.erase(std::find(g_teams.begin(), g_teams.end(), t));
        if (t->poisoned) {
            kill = t;  // (see MultiTeam::poisoned: fresh flags, slots and exchange numbers for the next handle on this list)
        } else {
        g_teams.push_back(t);
        int idle = 0;
        for (MultiTeam* q : g_teams) idle += q->refs == 0;
        if (idle > max_idle_teams()) {
            auto it = std::find_if(g_teams.begin(), g_teams.end(), [](MultiTeam* q) { return q->refs == 0; });
            kill = *it;
            g_teams.erase(it);
        }
        }
    }
    if (kill) team_destroy(kill);
}

// Give back what idle teams hold (per rank: two contexts with their pinned bounce buffers -- up to 2 x 48 MiB of host memory
// -- device staging, cached plan blocks, gather buffers): a long-lived host (MATLAB, a Python server) calls this when it is
// done with its device lists; a MEX file from its mexAtExit hook.  Teams in use are not touched.  Returns the number destroyed.
extern "C" int32_t snmf_multi_release_cache(void) {
    std::vector<MultiTeam*> kill;
    {
        std::lock_guard<std::mutex> lk(g_team_mu);
        for (auto it = g_teams.begin(); it != g_teams.end();)
            if ((*it)->refs == 0) {
                kill.push_back(*it);
                it = g_teams.erase(it);
            } else {
                ++it;
            }
    }
    int dev_before = -1;
    (void)hipGetDevice(&dev_before);
    for (MultiTeam* t : kill) team_destroy(t);
    if (dev_before >= 0) (void)hipSetDevice(dev_before);
    return (int32_t)kill.size();
}
// idle teams in the cache (tests)
extern "C" int32_t snmf_multi_cached_teams(void) {
    std::lock_guard<std::mutex> lk(g_team_mu);
    int n = 0;
    for (MultiTeam* t : g_teams) n += t->refs == 0;
    return n;
}

// Threads of the multi-device entries start through here: std::thread's constructor throws std::system_error when the
// process is out of threads, and these run inside extern "C" entries -- an exception must not unwind into MATLAB / ctypes
// (or reach std::terminate with earlier threads still joinable).  On failure the work runs on the calling thread instead.
template <typename Fn>
static void spawn_or_run(std::vector<std::thread>& th, Fn fn) {
    try {
        th.emplace_back(fn);
    } catch (const std::system_error&) {
        fn();
    }
}

extern "C" void snmf_multi_destroy(snmf_multi* m) {
    if (!m) return;
    for (int g = 0; g < (int)m->plan.size(); ++g) {
        if (g < (int)m->ctx.size() && m->ctx[g]) {
            hipSetDevice(m->dev[g]);
            hipStreamSynchronize(m->ctx[g]->stream);
        }
    }
    for (int g = 0; g < (int)m->plan.size(); ++g) {
        if (g >= (int)m->ctx.size() || !m->ctx[g]) continue;
        hipSetDevice(m->dev[g]);
        if (g < (int)m->stats.size() && m->stats[g]) hipFree(m->stats[g]);
        if (m->plan[g]) snmf_plan_destroy(m->plan[g]);
    }
    if (m->team && m->team_owner) team_release(m->team);
    delete m;
}

// the handle on a team the caller holds (team == nullptr: take one out of the cache for this handle's lifetime)
static int multi_create_on(MultiTeam* team, bool use_aux, const int32_t* devices, int32_t n_dev, const snmf_params* p,
                           const int64_t* col_begin, snmf_multi** out) {
    if (!devices || !p || !out) return fail(SNMF_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_dev < 1 || n_dev > 16) return fail(SNMF_ERR_INVALID, "n_dev = %d outside [1, 16]", n_dev);
    SN_TRY(validate_params(p));
    if (p->T < n_dev) return fail(SNMF_ERR_INVALID, "fewer frames (%d) than ranks (%d)", p->T, n_dev);
    int dev_before = -1;
    (void)hipGetDevice(&dev_before);  // the calling thread's current device is restored on the way out (torch / gpuArray callers)
    struct Restore {
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{dev_before};
    snmf_multi* m = new snmf_multi();
    m->n = n_dev;
    m->p = *p;
    m->dev.assign(devices, devices + n_dev);
    m->col.resize(n_dev + 1);
    for (int g = 0; g <= n_dev; ++g) m->col[g] = col_begin ? col_begin[g] : ((int64_t)p->T * g) / n_dev;  // balanced, contiguous
    if (m->col[0] != 0 || m->col[n_dev] != p->T) {
        delete m;
        return fail(SNMF_ERR_INVALID, "column ranges must start at 0 and end at T");
    }
    for (int g = 0; g < n_dev; ++g)
        if (m->col[g + 1] <= m->col[g]) {
            delete m;
            return fail(SNMF_ERR_INVALID, "rank %d owns no frames", g);
        }
    if (p->w_update_ind) m->w_ind.assign(p->w_update_ind, p->w_update_ind + p->r);
    if (p->h_update_ind) m->h_ind.assign(p->h_update_ind, p->h_update_ind + p->r);
    m->p.w_update_ind = m->w_ind.empty() ? nullptr : m->w_ind.data();
    m->p.h_update_ind = m->h_ind.empty() ? nullptr : m->h_ind.data();
    m->plan.assign(n_dev, nullptr);
    m->stats.assign(n_dev, nullptr);
    m->rc.assign(n_dev, SNMF_OK);
    m->err.assign(n_dev, std::string());
    m->use_aux = use_aux;
    int s = SNMF_OK;
    if (team) {
        m->team = team;
        m->team_owner = false;
    } else {
        s = team_acquire(devices, n_dev, &m->team);
        if (s != SNMF_OK) {
            delete m;
            return s;
        }
    }
    MultiTeam* t = m->team;
    m->ctx.assign(n_dev, nullptr);
    for (int g = 0; g < n_dev; ++g) m->ctx[g] = use_aux ? t->ctx[g]->aux : t->ctx[g];
    m->shared_dev = t->shared_dev;
    for (int g = 0; g < n_dev && s == SNMF_OK; ++g) {
        snmf_params pg = m->p;
        pg.T = (int32_t)(m->col[g + 1] - m->col[g]);
        s = snmf_plan_create(m->ctx[g], &pg, &m->plan[g]);
    }
    if (s == SNMF_OK) {
        m->upd_w = m->plan[0]->upd_w;
        m->can_stop = p->cost_check && p->conv_eps > 0.0;
        m->len = (size_t)snmf_plan_stats_len(m->plan[0]);
        m->xoff = m->upd_w ? 0 : m->len - 2;  // H-only solves exchange nothing but (div, sum S.*H)
        m->xlen = m->len - m->xoff;
        s = team_reserve(t, m->xlen);
    }
    for (int g = 0; g < n_dev && s == SNMF_OK; ++g) {
        if (hipSetDevice(m->dev[g]) != hipSuccess) s = fail(SNMF_ERR_NO_DEVICE, "hipSetDevice(%d)", m->dev[g]);
        if (s == SNMF_OK) s = dalloc(&m->stats[g], m->len);
        if (s == SNMF_OK) hipMemset(m->stats[g], 0, m->len * sizeof(double));
    }
    if (s != SNMF_OK) {
        const std::string keep = g_err;
        snmf_multi_destroy(m);
        g_err = keep;
        return s;
    }
    m->fine_grained = t->fine_grained;
    // AUTO: device-side ordering when every rank has a device of its own and the buffers are coherent
    m->mode = (!t->shared_dev && t->fine_grained) ? SNMF_EXCHANGE_FLAGS : SNMF_EXCHANGE_EVENTS;
    *out = m;
    return SNMF_OK;
}

extern "C" int snmf_multi_create(const int32_t* devices, int32_t n_dev, const snmf_params* p, const int64_t* col_begin,
                                 snmf_multi** out) {
    return multi_create_on(nullptr, false, devices, n_dev, p, col_begin, out);
}

extern "C" int snmf_multi_set_exchange(snmf_multi* m, int32_t mode) {
    if (!m) return fail(SNMF_ERR_INVALID, "multi handle is NULL");
    if (mode != SNMF_EXCHANGE_AUTO && mode != SNMF_EXCHANGE_FLAGS && mode != SNMF_EXCHANGE_EVENTS)
        return fail(SNMF_ERR_INVALID, "unknown exchange mode %d", mode);
    if (m->it != 0 && m->inited) return fail(SNMF_ERR_STATE, "the exchange mode can only change between solves (before snmf_multi_run)");
    if (mode == SNMF_EXCHANGE_AUTO) mode = (!m->shared_dev && m->fine_grained) ? SNMF_EXCHANGE_FLAGS : SNMF_EXCHANGE_EVENTS;
    if (mode == SNMF_EXCHANGE_FLAGS && !m->fine_grained)
        return fail(SNMF_ERR_UNSUPPORTED, "FLAGS ordering needs fine-grained gather buffers, which this runtime refused to allocate (EVENTS is in use)");
    m->mode = mode;
    return SNMF_OK;
}

#define MULTI_CHECK(m) \
    if (!(m)) return fail(SNMF_ERR_INVALID, "multi handle is NULL")

// Run fn(g) for every rank, ranks 1.. on threads of their own (each GPU has its own PCIe link, each context its own pinned
// pipeline; the host-side narrowing pool takes jobs from several threads at once): round 4 walked the ranks one after another
// from the calling thread.  Returns the first failure with its message.
template <typename Fn>
static int multi_for_ranks(snmf_multi* m, Fn fn) {
    std::vector<int> rc(m->n, SNMF_OK);
    std::vector<std::string> er(m->n);
    int dev_before = -1;
    (void)hipGetDevice(&dev_before);
    auto one = [&](int g) {
        if (hipSetDevice(m->dev[g]) != hipSuccess) rc[g] = fail(SNMF_ERR_NO_DEVICE, "hipSetDevice(%d) failed", m->dev[g]);
        else rc[g] = fn(g);
        if (rc[g] != SNMF_OK) er[g] = g_err;
    };
    std::vector<std::thread> th;
    for (int g = 1; g < m->n; ++g) spawn_or_run(th, [&one, g] { one(g); });
    one(0);
    for (auto& t : th) t.join();
    if (dev_before >= 0) (void)hipSetDevice(dev_before);
    for (int g = 0; g < m->n; ++g)
        if (rc[g] != SNMF_OK) return fail(rc[g], "rank %d (device %d): %s", g, m->dev[g], er[g].c_str());
    return SNMF_OK;
}

template <typename T>
static int multi_set_cols(snmf_multi* m, const T* A, int64_t ld, int which) {
    MULTI_CHECK(m);
    if (!A) return fail(SNMF_ERR_INVALID, "NULL matrix");
    SN_TRY(multi_for_ranks(m, [&](int g) {
        const T* Ag = A + (size_t)m->col[g] * ld;
        return which == 0 ? set_v<T>(m->plan[g], Ag, ld, 0) : set_h<T>(m->plan[g], Ag, ld, 0);
    }));
    m->inited = false;
    return SNMF_OK;
}
extern "C" int snmf_multi_set_v_f64(snmf_multi* m, const double* V, int64_t ld) { return multi_set_cols(m, V, ld, 0); }
extern "C" int snmf_multi_set_v_f32(snmf_multi* m, const float* V, int64_t ld) { return multi_set_cols(m, V, ld, 0); }
extern "C" int snmf_multi_set_h_f64(snmf_multi* m, const double* H, int64_t ld) { return multi_set_cols(m, H, ld, 1); }
extern "C" int snmf_multi_set_h_f32(snmf_multi* m, const float* H, int64_t ld) { return multi_set_cols(m, H, ld, 1); }
template <typename T>
static int multi_set_w(snmf_multi* m, const T* W, int64_t ld) {
    MULTI_CHECK(m);
    if (!W) return fail(SNMF_ERR_INVALID, "NULL matrix");
    SN_TRY(multi_for_ranks(m, [&](int g) { return set_w<T>(m->plan[g], W, ld, 0); }));
    m->inited = false;
    return SNMF_OK;
}
extern "C" int snmf_multi_set_w_f64(snmf_multi* m, const double* W, int64_t ld) { return multi_set_w(m, W, ld); }
extern "C" int snmf_multi_set_w_f32(snmf_multi* m, const float* W, int64_t ld) { return multi_set_w(m, W, ld); }
template <typename T>
static int multi_set_s(snmf_multi* m, const T* S) {
    MULTI_CHECK(m);
    if (!S) return fail(SNMF_ERR_INVALID, "NULL sparsity");
    for (int g = 0; g < m->n; ++g) {
        const T* Sg = m->p.sparsity_kind == SNMF_SPARSITY_FULL ? S + (size_t)m->col[g] * m->p.r : S;  // r x T, ld = r
        SN_TRY(set_s<T>(m->plan[g], Sg, 0));
    }
    m->inited = false;
    return SNMF_OK;
}
extern "C" int snmf_multi_set_sparsity_f64(snmf_multi* m, const double* S) { return multi_set_s(m, S); }
extern "C" int snmf_multi_set_sparsity_f32(snmf_multi* m, const float* S) { return multi_set_s(m, S); }

extern "C" int snmf_multi_init(snmf_multi* m) {
    MULTI_CHECK(m);
    // (the team's gather buffers may have been re-allocated for a later handle with longer statistics: the handle always goes
    //  through the team's pointers, and falls back to EVENTS ordering if that allocation came out coarse-grained)
    m->fine_grained = m->team->fine_grained;
    if (!m->fine_grained) m->mode = SNMF_EXCHANGE_EVENTS;
    for (int g = 0; g < m->n; ++g) SN_TRY(snmf_plan_init(m->plan[g]));
    // this handle's view of the team's gather buffers starts from zeros (the fused push writes the r real rows only: the pad
    // rows must be zero, and another handle of the team may have used the memory with another layout)
    for (int g = 0; g < m->n; ++g) {
        HIP_TRY(hipSetDevice(m->dev[g]));
        HIP_TRY(hipMemsetAsync(m->team->slots[g], 0, (size_t)2 * m->n * m->xlen * sizeof(double), m->ctx[g]->stream));
    }
    for (int g = 0; g < m->n; ++g) SN_TRY(snmf_ctx_sync(m->ctx[g]));
    m->it = 0;
    m->inited = true;
    m->finalized = false;
    m->stopped = false;
    return SNMF_OK;
}

// One exchange on rank g, number `xs` (>= 1), parity `par`.  `rc` is the rank's status so far.
// FLAGS mode: two launches, nothing else -- the devices order push and sum among themselves.
// EVENTS mode: the host barrier makes sure every peer has ISSUED its event record before anybody waits on it, and that
// nobody re-records an event a peer has not yet waited on; a failed rank still takes part in the barrier.
// Returns false when the ranks agreed to stop (some rank failed; EVENTS mode only -- in FLAGS mode a failed rank is
// found at the barrier behind the loop, and its peers' polls time out on the device).
// fused = the push already happened inside this rank's k_reduce launch (snmf_plan::xpush) and the sum will happen inside its
// k_wapply launch (snmf_plan::xgather): only the ORDERING between the two is left to do here (W updates; round 4: an
// iteration of a rank is four launches instead of six).
static bool multi_exchange(snmf_multi* m, int g, int par, unsigned xs, int& seq, int& rc, std::string& err, bool fused = false) {
    hipStream_t st = m->ctx[g]->stream;
    auto note = [&](int s) {
        if (rc == SNMF_OK && s != SNMF_OK) {
            rc = s;
            err = g_err;
        }
    };
    const bool flags = m->mode == SNMF_EXCHANGE_FLAGS;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((m->xlen + 255) / 256, 512));
    if (fused) {
        if (m->n > 1 && rc == SNMF_OK && !flags && hipEventRecord(m->team->ev[par][g], st) != hipSuccess) note(fail(SNMF_ERR_NO_DEVICE, "hipEventRecord failed"));
    } else if (m->n > 1 && rc == SNMF_OK) {
        PushArgs pa{};
        pa.src = m->stats[g] + m->xoff;
        pa.len = m->xlen;
        pa.n = m->n;
        pa.seq = xs;
        for (int q = 0; q < m->n; ++q) {
            pa.dst[q] = m->team->slots[q] + ((size_t)par * m->n + g) * m->xlen;
            pa.flag[q] = flags ? m->team->flags[q] + (size_t)par * m->n + g : nullptr;
        }
        pa.done_ctr = flags ? m->team->flags[g] + (size_t)2 * m->n : nullptr;
        hipLaunchKernelGGL(k_push_stats, dim3(grid), dim3(256), 0, st, pa);
        if (hipGetLastError() != hipSuccess) note(fail(SNMF_ERR_NO_DEVICE, "k_push_stats launch failed"));
        else if (!flags && hipEventRecord(m->team->ev[par][g], st) != hipSuccess) note(fail(SNMF_ERR_NO_DEVICE, "hipEventRecord failed"));
    }
    // every rank has recorded (EVENTS) / submitted its push (FLAGS on a shared device) -- or somebody failed: all leave
    if ((!flags || m->shared_dev) && !multi_barrier(m, seq++, rc != SNMF_OK)) return false;
    if (m->n > 1) {
        if (!flags)
            for (int q = 0; q < m->n; ++q)
                if (q != g && hipStreamWaitEvent(st, m->team->ev[par][q], 0) != hipSuccess) note(fail(SNMF_ERR_NO_DEVICE, "hipStreamWaitEvent failed"));
        if (rc == SNMF_OK && !fused) {
            hipLaunchKernelGGL(k_sum_ranks, dim3(grid), dim3(256), 0, st, (const double*)(m->team->slots[g] + (size_t)par * m->n * m->xlen),
                               m->n, m->xlen, m->stats[g] + m->xoff, flags ? (const unsigned*)(m->team->flags[g] + (size_t)par * m->n) : nullptr,
                               xs, &m->plan[g]->st->fault);
            if (hipGetLastError() != hipSuccess) note(fail(SNMF_ERR_NO_DEVICE, "k_sum_ranks launch failed"));
        }
    }
    return true;
}

// the loop of rank g: iterations [it0, target), then (optionally) the objective of the last iterate
static void multi_rank_loop(snmf_multi* m, int g, int it0, int target, bool finalize, int* stopped_out) {
    int rc = SNMF_OK, seq = 0;
    std::string err;
    auto step = [&](int st) {
        if (rc == SNMF_OK && st != SNMF_OK) {
            rc = st;
            err = g_err;
        }
    };
    step(hipSetDevice(m->dev[g]) == hipSuccess ? SNMF_OK : fail(SNMF_ERR_NO_DEVICE, "hipSetDevice(%d) failed", m->dev[g]));
    int par = m->team->par, since = 0, stopped = 0;
    unsigned xs = m->team->xseq;
    const bool flags = m->mode == SNMF_EXCHANGE_FLAGS;
    bool all_ok = true;
    // W updates: the exchange rides on the iteration's own launches (k_reduce pushes, k_wapply sums); SNMF_MULTI_UNFUSED=1
    // keeps the separate k_push_stats / k_sum_ranks launches (tests compare the two: bit-identical)
    const char* uf = getenv("SNMF_MULTI_UNFUSED");
    const bool fused = m->upd_w && m->n > 1 && m->xoff == 0 && !(uf && atoi(uf) != 0);
    // fault injection (tests only): SNMF_MULTI_TEST_FAIL="rank:iteration" makes that rank's H step of that iteration (0-based,
    // counted from the solve's start) report a failure, as a refused launch would
    int inj_rank = -1, inj_it = -1;
    if (const char* fi = getenv("SNMF_MULTI_TEST_FAIL")) (void)sscanf(fi, "%d:%d", &inj_rank, &inj_it);
    for (int it = it0; it < target && all_ok; ++it) {
        if (rc == SNMF_OK && g == inj_rank && it == inj_it) step(fail(SNMF_ERR_INTERNAL, "injected failure (SNMF_MULTI_TEST_FAIL)"));
        if (rc == SNMF_OK) step(snmf_plan_hstep(m->plan[g]));
        snmf_plan::Exchange X;
        ++xs;
        if (fused) {
            X.n = m->n;
            X.len = m->xlen;
            X.seq = xs;
            for (int q = 0; q < m->n; ++q) {
                X.push_dst[q] = m->team->slots[q] + ((size_t)par * m->n + g) * m->xlen;
                X.push_flag[q] = flags ? m->team->flags[q] + (size_t)par * m->n + g : nullptr;
            }
            X.push_done = flags ? m->team->flags[g] + (size_t)2 * m->n : nullptr;
            X.gather = m->team->slots[g] + (size_t)par * m->n * m->xlen;
            X.gflags = flags ? (const unsigned*)(m->team->flags[g] + (size_t)par * m->n) : nullptr;
            m->plan[g]->xpush = &X;
        }
        if (rc == SNMF_OK) step(snmf_plan_wstats(m->plan[g], m->stats[g]));
        m->plan[g]->xpush = nullptr;
        all_ok = multi_exchange(m, g, par, xs, seq, rc, err, fused);
        if (!all_ok) break;
        par ^= 1;
        if (fused) m->plan[g]->xgather = &X;
        if (rc == SNMF_OK) step(snmf_plan_wapply(m->plan[g], m->stats[g]));
        m->plan[g]->xgather = nullptr;
        if (m->can_stop && ++since >= 4) {  // same poll schedule and (bit-identical statistics) same answer on every rank
            since = 0;
            int32_t sflag = 0;
            if (rc == SNMF_OK) step(snmf_plan_stopped(m->plan[g], &sflag));
            // (FLAGS mode: no rendezvous -- every rank reads the same flag value: the statistics are bit-identical)
            if (!flags || m->shared_dev) all_ok = multi_barrier(m, seq++, rc != SNMF_OK);
            else if (rc != SNMF_OK) break;
            if (all_ok && sflag) {
                stopped = 1;
                break;
            }
        }
    }
    if (all_ok) all_ok = multi_barrier(m, seq++, rc != SNMF_OK);  // agree on the state the loop was left in
    if (all_ok && finalize && !stopped) {
        if (rc == SNMF_OK) step(snmf_plan_objstats(m->plan[g], m->stats[g]));
        all_ok = multi_exchange(m, g, par, ++xs, seq, rc, err);
        if (all_ok) {
            par ^= 1;
            if (rc == SNMF_OK) step(snmf_plan_objapply(m->plan[g], m->stats[g]));
        }
    }
    if (rc == SNMF_OK) step(snmf_ctx_sync(m->ctx[g]));
    m->rc[g] = rc;
    m->err[g] = err;
    if (g == 0) {
        m->team->par = par;
        m->team->xseq = xs;
        *stopped_out = stopped;
    }
}

extern "C" int snmf_multi_run(snmf_multi* m, int32_t n_iters, int32_t* iters_done) {
    MULTI_CHECK(m);
    if (!m->inited) return fail(SNMF_ERR_STATE, "snmf_multi_init must precede snmf_multi_run");
    const int target = std::min(m->p.max_iter, m->it + std::max(0, n_iters));
    const bool finalize = target >= m->p.max_iter && m->p.cost_check && !m->finalized && target > 0;
    int stopped = 0, dev_before = -1;
    (void)hipGetDevice(&dev_before);  // rank 0's loop runs on the calling thread and moves its current device
    if (!m->stopped && (m->it < target || finalize)) {
        m->failed_at.store(0x7fffffff);
        m->bar_count.store(0);
        std::vector<std::thread> th;
        // (the rank loops meet at host barriers, so a rank whose thread cannot be started cannot run inline behind the others:
        //  every thread is created first and held at a latch; if the process is out of threads nobody has begun, the ones that
        //  exist leave at once and the run is refused -- std::system_error must not unwind through this extern "C" entry)
        std::atomic<int> go{0};  // 0: hold, 1: run, -1: leave
        const int it0 = m->it;
        bool spawned = true;
        try {
            for (int g = 1; g < m->n; ++g)
                th.emplace_back([&, g] {
                    while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
                    if (go.load(std::memory_order_acquire) > 0) multi_rank_loop(m, g, it0, target, finalize, &stopped);
                });
        } catch (const std::system_error&) {
            spawned = false;
        }
        if (!spawned) {
            go.store(-1, std::memory_order_release);
            for (auto& t : th) t.join();
            return fail(SNMF_ERR_NOMEM, "snmf_multi_run: the process cannot start %d rank threads", m->n - 1);
        }
        go.store(1, std::memory_order_release);
        multi_rank_loop(m, 0, m->it, target, finalize, &stopped);
        for (auto& t : th) t.join();
        if (dev_before >= 0) (void)hipSetDevice(dev_before);
        // Any failure below leaves the team's exchange number / parity / arrival words in a state the ranks may not agree on
        // (MultiTeam::poisoned): the team is not cached again.
        for (int g = 0; g < m->n; ++g)
            if (m->rc[g] != SNMF_OK) {
                m->team->poisoned = true;
                return fail(m->rc[g], "rank %d (device %d): %s", g, m->dev[g], m->err[g].c_str());
            }
        if (m->failed_at.load() != 0x7fffffff) {
            m->team->poisoned = true;
            return fail(SNMF_ERR_INTERNAL, "a rank failed");
        }
        for (int g = 0; g < m->n; ++g) {  // a device-side wait that gave up (FLAGS mode: a peer never arrived) invalidates the run
            DevState hs{};
            if (hipSetDevice(m->dev[g]) != hipSuccess) {
                m->team->poisoned = true;
                return fail(SNMF_ERR_NO_DEVICE, "hipSetDevice(%d) failed", m->dev[g]);
            }
            const int rs = read_state(m->plan[g], &hs);
            if (dev_before >= 0) (void)hipSetDevice(dev_before);
            if (rs != SNMF_OK) {
                m->team->poisoned = true;
                return fail(rs, "rank %d (device %d): %s", g, m->dev[g], g_err.c_str());
            }
        }
        m->it = m->plan[0]->it_done;
        if (stopped) m->stopped = true;
        else if (finalize) m->finalized = true;
    }
    if (iters_done) {
        int32_t n_it = 0;
        SN_TRY(snmf_plan_get_objective(m->plan[0], nullptr, nullptr, &n_it));
        *iters_done = n_it;
    }
    return SNMF_OK;
}

extern "C" int snmf_multi_get_w_f64(snmf_multi* m, double* W, int64_t ld) {
    MULTI_CHECK(m);
    return snmf_plan_get_w_f64(m->plan[0], W, ld, 0);
}
extern "C" int snmf_multi_get_w_f32(snmf_multi* m, float* W, int64_t ld) {
    MULTI_CHECK(m);
    return snmf_plan_get_w_f32(m->plan[0], W, ld, 0);
}
// the W replica of one rank (tests: replicas must be bit-identical)
extern "C" int snmf_multi_get_w_rank_f64(snmf_multi* m, int32_t rank, double* W, int64_t ld) {
    MULTI_CHECK(m);
    if (rank < 0 || rank >= m->n) return fail(SNMF_ERR_INVALID, "rank out of range");
    return snmf_plan_get_w_f64(m->plan[rank], W, ld, 0);
}
extern "C" int snmf_multi_get_h_f64(snmf_multi* m, double* H, int64_t ld) {
    MULTI_CHECK(m);
    return multi_for_ranks(m, [&](int g) { return snmf_plan_get_h_f64(m->plan[g], H + (size_t)m->col[g] * ld, ld, 0); });
}
extern "C" int snmf_multi_get_h_f32(snmf_multi* m, float* H, int64_t ld) {
    MULTI_CHECK(m);
    return multi_for_ranks(m, [&](int g) { return snmf_plan_get_h_f32(m->plan[g], H + (size_t)m->col[g] * ld, ld, 0); });
}
extern "C" int snmf_multi_get_objective(snmf_multi* m, double* div_out, double* cost_out, int32_t* n_iter_out) {
    MULTI_CHECK(m);
    return snmf_plan_get_objective(m->plan[0], div_out, cost_out, n_iter_out);
}

template <typename T>
static int sparse_nmf_multi_impl(const int32_t* devices, int32_t n_dev, const snmf_params* p, const T* V, int64_t ldV, T* W,
                                 T* H, const T* sparsity, double* div_out, double* cost_out, int32_t* n_iter_out) {
    if (!V || !W || !H) return fail(SNMF_ERR_INVALID, "V, W and H must be non-NULL");
    if (p && p->T >= 1 && n_dev > p->T) n_dev = p->T;  // a short clip on a long device list: use as many ranks as there are frames
    snmf_multi* m = nullptr;
    SN_TRY(snmf_multi_create(devices, n_dev, p, nullptr, &m));
    int s = SNMF_OK;
    SN_STEP(s, multi_set_cols<T>(m, V, ldV, 0));
    SN_STEP(s, multi_set_w<T>(m, W, p->F));
    SN_STEP(s, multi_set_cols<T>(m, H, p->r, 1));
    if (p->sparsity_kind != SNMF_SPARSITY_SCALAR) {
        if (!sparsity) SN_STEP(s, fail(SNMF_ERR_INVALID, "sparsity array required for this sparsity_kind"));
        else SN_STEP(s, multi_set_s<T>(m, sparsity));
    }
    SN_STEP(s, snmf_multi_init(m));
    if (s == SNMF_OK) SN_STEP(s, snmf_multi_run(m, p->max_iter, nullptr));
    if (s == SNMF_OK) {
        if (sizeof(T) == 8) {
            SN_STEP(s, snmf_multi_get_w_f64(m, (double*)W, p->F));
            SN_STEP(s, snmf_multi_get_h_f64(m, (double*)H, p->r));
        } else {
            SN_STEP(s, snmf_multi_get_w_f32(m, (float*)W, p->F));
            SN_STEP(s, snmf_multi_get_h_f32(m, (float*)H, p->r));
        }
    }
    if (s == SNMF_OK) SN_STEP(s, snmf_multi_get_objective(m, div_out, cost_out, n_iter_out));
    const std::string keep = g_err;
    snmf_multi_destroy(m);
    g_err = keep;
    return s;
}
extern "C" int snmf_sparse_nmf_multi_f64(const int32_t* devices, int32_t n_dev, const snmf_params* p, const double* V, int64_t ldV,
                                         double* W, double* H, const double* sparsity, double* div_out, double* cost_out,
                                         int32_t* n_iter_out) {
    return sparse_nmf_multi_impl<double>(devices, n_dev, p, V, ldV, W, H, sparsity, div_out, cost_out, n_iter_out);
}
extern "C" int snmf_sparse_nmf_multi_f32(const int32_t* devices, int32_t n_dev, const snmf_params* p, const float* V, int64_t ldV,
                                         float* W, float* H, const float* sparsity, double* div_out, double* cost_out,
                                         int32_t* n_iter_out) {
    return sparse_nmf_multi_impl<float>(devices, n_dev, p, V, ldV, W, H, sparsity, div_out, cost_out, n_iter_out);
}

// ---- B_hat = run_basis_DNMF(x, d, B, p) over a device list, device-resident (round 5) ---------------------------------------
// run_basis_DNMF.m:36-55 with the frames of all three solves sharded over the ranks of ONE team: Y, X, D cross PCIe once each --
// every rank's shard through its own pinned pipeline, all ranks at once, X and D on the second contexts under solve 1 -- A_hat
// never leaves HBM (each rank hands ITS columns of solve 1's H to its plans of solves 2 / 3), only B_hat (and A_hat when asked
// for) comes back.  Round 4 ran this as three snmf_sparse_nmf_multi_* calls: three handle constructions, serial shard uploads
// from the calling thread, A_hat device -> host -> device twice.
template <typename T>
static int dnmf_multi_impl(const int32_t* devices, int32_t n_dev, const snmf_params* p, int R_x, int R_d, const T* Y, int64_t ldY,
                           const T* X, int64_t ldX, const T* D, int64_t ldD, const T* B, int64_t ldB, const T* H0, uint64_t seed,
                           T* B_hat, int64_t ldBh, T* A_hat, int64_t ldA, int32_t* n_iter3) {
    if (!devices || !p || !Y || !X || !D || !B || !B_hat) return fail(SNMF_ERR_INVALID, "devices, p, Y, X, D, B and B_hat must be non-NULL");
    if (R_x < 1 || R_d < 1 || p->r != R_x + R_d) return fail(SNMF_ERR_DIM, "params->r = %d must equal R_x + R_d = %d + %d", p->r, R_x, R_d);
    if (p->sparsity_kind != SNMF_SPARSITY_SCALAR)
        return fail(SNMF_ERR_DIM, "run_basis_DNMF needs a scalar p.sparsity (its W-only solves have R_x / R_d rows)");
    if (ldB < p->F || ldBh < p->F) return fail(SNMF_ERR_INVALID, "leading dimension of B / B_hat < F");
    if (A_hat && ldA < p->r) return fail(SNMF_ERR_INVALID, "leading dimension of A_hat < R_x + R_d");
    if (n_dev < 1 || n_dev > 16) return fail(SNMF_ERR_INVALID, "n_dev = %d outside [1, 16]", n_dev);
    if (p->T >= 1 && n_dev > p->T) n_dev = p->T;
    (void)hipGetLastError();
    int dev_before = -1;
    (void)hipGetDevice(&dev_before);
    struct Cleanup {
        MultiTeam* team = nullptr;
        snmf_multi *m1 = nullptr, *m2 = nullptr, *m3 = nullptr;
        int dev = -1;
        ~Cleanup() {
            const std::string keep = g_err;
            if (m3) snmf_multi_destroy(m3);
            if (m2) snmf_multi_destroy(m2);
            if (m1) snmf_multi_destroy(m1);
            if (team) team_release(team);
            if (dev >= 0) (void)hipSetDevice(dev);
            g_err = keep;
        }
    } C;
    C.dev = dev_before;
    SN_TRY(team_acquire(devices, n_dev, &C.team));
    std::vector<uint8_t> on(p->r, 1), off(p->r, 0);
    snmf_params q = *p;
    q.w_update_ind = off.data();  // run_basis_DNMF.m:37
    q.h_update_ind = on.data();   // :38
    SN_TRY(multi_create_on(C.team, false, devices, n_dev, &q, nullptr, &C.m1));
    q.r = R_x;
    q.w_update_ind = on.data();   // :43
    q.h_update_ind = off.data();  // :44
    SN_TRY(multi_create_on(C.team, true, devices, n_dev, &q, nullptr, &C.m2));
    q.r = R_d;                    // :49-50
    SN_TRY(multi_create_on(C.team, true, devices, n_dev, &q, nullptr, &C.m3));
    snmf_multi *m1 = C.m1, *m2 = C.m2, *m3 = C.m3;
    // solve 1's operands: every rank its shard of Y, B, and ITS columns of the initial activations
    SN_TRY(multi_for_ranks(m1, [&](int g) {
        snmf_plan* pl = m1->plan[g];
        SN_TRY(set_v<T>(pl, Y + (size_t)m1->col[g] * ldY, ldY, 0));
        SN_TRY(set_w<T>(pl, B, ldB, 0));  // p.init_w = B   (:39)
        if (H0) SN_TRY(set_h<T>(pl, H0 + (size_t)m1->col[g] * p->r, p->r, 0));
        else SN_TRY(rand_h(pl, seed, m1->col[g]));
        return (int)SNMF_OK;
    }));
    // X, D and the two halves of B go up on the second contexts while solve 1 runs
    int rc_up = SNMF_OK;
    std::string err_up;
    auto up_fn = [&] {
        rc_up = multi_for_ranks(m2, [&](int g) {
            SN_TRY(set_v<T>(m2->plan[g], X + (size_t)m2->col[g] * ldX, ldX, 0));
            SN_TRY(set_v<T>(m3->plan[g], D + (size_t)m3->col[g] * ldD, ldD, 0));
            SN_TRY(set_w<T>(m2->plan[g], B, ldB, 0));                      // p.init_w = B(:,1:R_x)            (:45)
            SN_TRY(set_w<T>(m3->plan[g], B + (size_t)R_x * ldB, ldB, 0));  // p.init_w = B(:,R_x+1:R_x+R_d)    (:51)
            return (int)SNMF_OK;
        });
        if (rc_up != SNMF_OK) err_up = g_err;
    };
    std::thread up;
    try {
        up = std::thread(up_fn);
    } catch (const std::system_error&) {
        up_fn();  // out of threads: the uploads run first instead of under solve 1
    }
    int32_t n1 = 0, n2 = 0, n3 = 0;
    int s = snmf_multi_init(m1);
    SN_STEP(s, snmf_multi_run(m1, p->max_iter, &n1));  // [~, A_hat] = sparse_nmf(Y, p)   (:40)
    if (up.joinable()) up.join();
    if (s != SNMF_OK) return s;
    if (rc_up != SNMF_OK) return fail(rc_up, "%s", err_up.c_str());
    // p.init_h = A_hat(1:R_x,:) / A_hat(R_x+1:end,:)   (:46, :52): rows of each rank's resident fp32 H, device to device
    SN_TRY(multi_for_ranks(m2, [&](int g) {
        int idx = 0;
        SN_TRY(result_h_index(m1->plan[g], &idx));
        const float* A = m1->plan[g]->H[idx];
        SN_TRY(set_h<float>(m2->plan[g], A, m1->plan[g]->rp, 1));
        SN_TRY(set_h<float>(m3->plan[g], A + R_x, m1->plan[g]->rp, 1));
        return (int)SNMF_OK;
    }));
    int rc_a = SNMF_OK;
    std::string err_a;
    std::thread dl;
    auto dl_fn = [&] {  // A_hat -> host under solves 2 / 3 (solve 1's contexts are idle now)
        rc_a = sizeof(T) == 8 ? snmf_multi_get_h_f64(m1, (double*)A_hat, ldA) : snmf_multi_get_h_f32(m1, (float*)A_hat, ldA);
        if (rc_a != SNMF_OK) err_a = g_err;
    };
    if (A_hat) {
        try {
            dl = std::thread(dl_fn);
        } catch (const std::system_error&) {
            dl_fn();  // out of threads: the download runs ahead of solves 2 / 3
        }
    }
    s = snmf_multi_init(m2);
    SN_STEP(s, snmf_multi_run(m2, p->max_iter, &n2));  // [B_hat_x, ~] = sparse_nmf(X, p)  (:47)
    SN_STEP(s, snmf_multi_init(m3));
    SN_STEP(s, snmf_multi_run(m3, p->max_iter, &n3));  // [B_hat_d, ~] = sparse_nmf(D, p)  (:53)
    if (sizeof(T) == 8) {
        SN_STEP(s, snmf_multi_get_w_f64(m2, (double*)B_hat, ldBh));  // B_hat = [B_hat_x, B_hat_d]   (:55)
        SN_STEP(s, snmf_multi_get_w_f64(m3, (double*)B_hat + (size_t)R_x * ldBh, ldBh));
    } else {
        SN_STEP(s, snmf_multi_get_w_f32(m2, (float*)B_hat, ldBh));
        SN_STEP(s, snmf_multi_get_w_f32(m3, (float*)B_hat + (size_t)R_x * ldBh, ldBh));
    }
    if (dl.joinable()) dl.join();
    if (s != SNMF_OK) return s;
    if (rc_a != SNMF_OK) return fail(rc_a, "%s", err_a.c_str());
    if (n_iter3) {
        n_iter3[0] = n1;
        n_iter3[1] = n2;
        n_iter3[2] = n3;
    }
    return SNMF_OK;
}
extern "C" int snmf_run_basis_dnmf_multi_f64(const int32_t* devices, int32_t n_dev, const snmf_params* p, int32_t R_x, int32_t R_d,
                                             const double* Y, int64_t ldY, const double* X, int64_t ldX, const double* D, int64_t ldD,
                                             const double* B, int64_t ldB, const double* H0, uint64_t seed, double* B_hat, int64_t ldBh,
                                             double* A_hat, int64_t ldA, int32_t* n_iter_out) {
    return dnmf_multi_impl<double>(devices, n_dev, p, R_x, R_d, Y, ldY, X, ldX, D, ldD, B, ldB, H0, seed, B_hat, ldBh, A_hat, ldA, n_iter_out);
}
extern "C" int snmf_run_basis_dnmf_multi_f32(const int32_t* devices, int32_t n_dev, const snmf_params* p, int32_t R_x, int32_t R_d,
                                             const float* Y, int64_t ldY, const float* X, int64_t ldX, const float* D, int64_t ldD,
                                             const float* B, int64_t ldB, const float* H0, uint64_t seed, float* B_hat, int64_t ldBh,
                                             float* A_hat, int64_t ldA, int32_t* n_iter_out) {
    return dnmf_multi_impl<float>(devices, n_dev, p, R_x, R_d, Y, ldY, X, ldX, D, ldD, B, ldB, H0, seed, B_hat, ldBh, A_hat, ldA, n_iter_out);
}
