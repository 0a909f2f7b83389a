// point_mass_sharded.cpp -- single-process, multi-GPU host of the MPPI engine: the C ABI of
// include/mppi_gpu_amd_sharded.h and the C++ class of include/point_mass_sharded.hpp.
//
// One shard engine (mppi_create_shard) per device and ONE HOST WORKER THREAD per engine: a thread
// binds its device once (hipSetDevice is per-thread state) and from then on runs the commands the
// controlling thread posts to all workers at once -- enqueueing a solve on 8 GPUs costs one
// hand-over, not 8 x (launches x 3 us) in a row.  Workers spin briefly for the next command before
// they sleep: a closed loop posts one every few hundred microseconds.
//
// Exchange of the T*A+2 floats per shard and solve:
//   COLLECTIVE  mppi_solve_local_async -> ncclAllGather on the shard's stream -> mppi_solve_finish_async.
//               RCCL is called natively (ncclCommInitAll over the devices of the controller, one
//               communicator per worker thread -- RCCL's one-thread-per-device mode).
//   DIRECT      mppi_solve_exchange_async: the combine kernels store into each other's inboxes
//               (raw pointers: same process; hipDeviceEnablePeerAccess between the devices).
//   COPY        mppi_solve_local_async -> event -> every shard pulls all partials with
//               hipMemcpyPeerAsync -> mppi_solve_finish_async.
// The reference's host loop this serves: src/main.cu:309-374.
#include "../../../include/mppi_gpu_amd_sharded.h"
#include "../../../include/point_mass_sharded.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

struct Shard;
using Job = std::function<int(Shard&)>;

struct Shard {
    int index = 0, device = 0, dup_rank = 0;      // dup_rank: ordinal among the shards of one device
    long long k_begin = 0, k_end = 0;
    mppi_engine* eng = nullptr;
    hipStream_t stream = nullptr;
    float* d_partial[2] = {nullptr, nullptr};     // by solve parity (COPY: a peer may still be
    hipEvent_t ev_partial[2] = {nullptr, nullptr};  //  pulling solve j while solve j+1 is written)
    float* d_gathered = nullptr;
    ncclComm_t comm = nullptr;
    void* inbox = nullptr;
    float act[4] = {0.f, 0.f, 0.f, 0.f};

    // command hand-over
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    Job job;
    std::atomic<unsigned> posted{0}, done{0};
    bool quit = false;
    int rc = 0;
    std::string err;
};

// a failing engine call inside a worker: keep its message (mppi_last_error is thread-local)
int eng_fail(Shard& s, int rc, const char* what)
{
    char buf[768];
    snprintf(buf, sizeof buf, "shard %d (device %d): %s: %s", s.index, s.device, what,
             mppi_last_error());
    s.err = buf;
    return rc;
}
int hip_fail(Shard& s, hipError_t e, const char* what)
{
    char buf[512];
    snprintf(buf, sizeof buf, "shard %d (device %d): %s: HIP error %d (%s)", s.index, s.device, what,
             (int)e, hipGetErrorString(e));
    s.err = buf;
    (void)hipGetLastError();
    return MPPI_EHIP;
}
int nccl_fail(Shard& s, ncclResult_t r, const char* what)
{
    char buf[512];
    snprintf(buf, sizeof buf, "shard %d (device %d): %s: RCCL error %d (%s)", s.index, s.device,
             what, (int)r, ncclGetErrorString(r));
    s.err = buf;
    return MPPI_EHIP;
}
#define ENG(s, call)                                            \
    do {                                                        \
        int rc__ = (call);                                      \
        if (rc__ != MPPI_OK) return eng_fail((s), rc__, #call); \
    } while (0)
#define HIPW(s, call)                                              \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return hip_fail((s), e__, #call);   \
    } while (0)

void worker_main(Shard* sp)
{
    Shard& s = *sp;
    (void)hipSetDevice(s.device);
    unsigned seen = 0;
    for (;;) {
        for (int i = 0; i < 20000 && s.posted.load(std::memory_order_acquire) == seen; ++i) cpu_relax();
        Job job;
        {
            std::unique_lock<std::mutex> lk(s.mu);
            s.cv.wait(lk, [&] { return s.quit || s.posted.load(std::memory_order_acquire) != seen; });
            if (s.quit) return;
            seen = s.posted.load(std::memory_order_acquire);
            job = s.job;
        }
        s.err.clear();
        s.rc = job(s);
        s.done.store(seen, std::memory_order_release);
    }
}

}  // namespace

struct mppi_sharded {
    int K = 0, T = 0, S = 0, A = 0, TA = 0, L = 0, n = 0, transport = MPPI_XPORT_COLLECTIVE;
    int verbose = 0;
    unsigned long long n_solves = 0;       // since set_data (parity of the COPY buffers)
    bool data_set = false;
    float x0[8] = {0};
    std::vector<Shard*> shards;
    std::vector<void*> inboxes;

    // run fn on every worker, wait for all; first failure wins (its message goes to g_err)
    int run_all(const Job& fn)
    {
        std::vector<unsigned> want(shards.size());
        for (size_t i = 0; i < shards.size(); ++i) {
            Shard& s = *shards[i];
            {
                std::lock_guard<std::mutex> lk(s.mu);
                s.job = fn;
                want[i] = s.posted.load(std::memory_order_relaxed) + 1;
                s.posted.store(want[i], std::memory_order_release);
            }
            s.cv.notify_one();
        }
        int rc = MPPI_OK;
        for (size_t i = 0; i < shards.size(); ++i) {
            Shard& s = *shards[i];
            int spins = 0;
            while (s.done.load(std::memory_order_acquire) != want[i]) {
                if (++spins < 4000) cpu_relax();
                else std::this_thread::yield();
            }
            if (s.rc != MPPI_OK && rc == MPPI_OK) {
                rc = s.rc;
                g_err = s.err.empty() ? "shard command failed" : s.err;
            }
        }
        return rc;
    }
};

namespace {

void split_range(long long K, int i, int n, long long& b, long long& e)
{   // contiguous and balanced: the first K % n shards hold one sample more (as sharded.py)
    const long long base = K / n, extra = K % n;
    b = i * base + (i < extra ? i : extra);
    e = b + base + (i < extra ? 1 : 0);
}

}  // namespace

extern "C" {

const char* mppi_sharded_last_error(void) { return g_err.c_str(); }

void mppi_sharded_destroy(mppi_sharded* c)
{
    if (!c) return;
    (void)c->run_all([c](Shard& s) {
        if (s.eng) {
            (void)mppi_sync_act(s.eng, nullptr);
            if (c->transport == MPPI_XPORT_DIRECT) (void)mppi_xchg_close(s.eng);
        }
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.comm) (void)ncclCommDestroy(s.comm);
        if (s.eng) mppi_destroy(s.eng);
        for (int p = 0; p < 2; ++p) {
            if (s.d_partial[p]) (void)hipFree(s.d_partial[p]);
            if (s.ev_partial[p]) (void)hipEventDestroy(s.ev_partial[p]);
        }
        if (s.d_gathered) (void)hipFree(s.d_gathered);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        s.eng = nullptr;
        return MPPI_OK;
    });
    for (Shard* s : c->shards) {
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->quit = true;
        }
        s->cv.notify_one();
        if (s->th.joinable()) s->th.join();
        delete s;
    }
    delete c;
}

int mppi_sharded_create(int K, int T, float dt, int S, int A, int verbose, int n_shards,
                        const int* devices, int transport, mppi_sharded** out)
{
    if (!out) return fail(MPPI_EINVAL, "out is null");
    *out = nullptr;
    if (transport < MPPI_XPORT_COLLECTIVE || transport > MPPI_XPORT_COPY)
        return fail(MPPI_EINVAL, "unknown transport %d", transport);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MPPI_ENODEV, "no HIP device: the engine has no CPU fallback");
    if (n_shards == 0) n_shards = ndev;
    if (n_shards < 1 || n_shards > 64) return fail(MPPI_EINVAL, "n_shards %d out of range (1..64)", n_shards);
    if (K < n_shards) return fail(MPPI_EINVAL, "fewer samples (%d) than shards (%d)", K, n_shards);
    if (A < 1 || A > 4 || S != 2 * A || T < 1 || !(dt > 0.f))
        return fail(MPPI_EINVAL, "bad problem dimensions (act_dim 1..4, state_dim = 2*act_dim, "
                    "steps >= 1, dt > 0)");
    std::vector<int> devs(n_shards);
    for (int i = 0; i < n_shards; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev)
            return fail(MPPI_EINVAL, "shard %d: device %d not visible (%d devices)", i, devs[i], ndev);
    }
    bool dup = false;
    for (int i = 0; i < n_shards; ++i)
        for (int j = 0; j < i; ++j) dup = dup || devs[i] == devs[j];
    if (dup && transport == MPPI_XPORT_COLLECTIVE)
        return fail(MPPI_EINVAL, "RCCL cannot place two ranks on one device: name every device once, "
                    "or use MPPI_XPORT_DIRECT / MPPI_XPORT_COPY for a rehearsal on one GPU");

    mppi_sharded* c = new mppi_sharded();
    c->K = K; c->T = T; c->S = S; c->A = A; c->TA = T * A; c->L = T * A + 2;
    c->n = n_shards; c->transport = transport; c->verbose = verbose;
    for (int i = 0; i < n_shards; ++i) {
        Shard* s = new Shard();
        s->index = i;
        s->device = devs[i];
        for (int j = 0; j < i; ++j) s->dup_rank += devs[j] == devs[i] ? 1 : 0;
        split_range(K, i, n_shards, s->k_begin, s->k_end);
        c->shards.push_back(s);
        s->th = std::thread(worker_main, s);
    }
    *out = c;       // a failure below still lets the caller destroy it

    // engines, streams, exchange buffers -- each on its own device, by its own thread
    int rc = c->run_all([c, dt, verbose, dup](Shard& s) {
        ENG(s, mppi_create_shard((int)(s.k_end - s.k_begin), s.k_begin, c->T, dt, c->S, c->A,
                                 verbose, &s.eng));
        if (dup) {
            // shards that share a device (rehearsal) wait for each other inside their kernels, so
            // their kernels must be able to run side by side: streams of different priority get
            // separate hardware queues
            int least = 0, greatest = 0;
            HIPW(s, hipDeviceGetStreamPriorityRange(&least, &greatest));
            int prio = greatest + s.dup_rank;
            if (prio > least) prio = least;
            HIPW(s, hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio));
        } else {
            HIPW(s, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        }
        for (int p = 0; p < 2; ++p) {
            HIPW(s, hipMalloc(&s.d_partial[p], (size_t)c->L * sizeof(float)));
            HIPW(s, hipMemset(s.d_partial[p], 0, (size_t)c->L * sizeof(float)));
            HIPW(s, hipEventCreateWithFlags(&s.ev_partial[p], hipEventDisableTiming));
        }
        HIPW(s, hipMalloc(&s.d_gathered, (size_t)c->L * c->n * sizeof(float)));
        HIPW(s, hipMemset(s.d_gathered, 0, (size_t)c->L * c->n * sizeof(float)));
        // peers: stores (DIRECT) and copies (COPY) cross devices
        for (Shard* o : c->shards) {
            if (o->device == s.device) continue;
            int can = 0;
            HIPW(s, hipDeviceCanAccessPeer(&can, s.device, o->device));
            if (!can) {
                if (c->transport == MPPI_XPORT_DIRECT) {
                    s.err = "device " + std::to_string(s.device) + " cannot access device " +
                            std::to_string(o->device) + ": the direct transport needs peer access";
                    return (int)MPPI_ENODEV;
                }
                continue;
            }
            const hipError_t pe = hipDeviceEnablePeerAccess(o->device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                return hip_fail(s, pe, "hipDeviceEnablePeerAccess");
            (void)hipGetLastError();
        }
        if (c->transport == MPPI_XPORT_DIRECT)
            ENG(s, mppi_xchg_open(s.eng, s.index, c->n, nullptr, &s.inbox));
        return (int)MPPI_OK;
    });
    if (rc) return rc;

    if (transport == MPPI_XPORT_DIRECT) {
        c->inboxes.resize(n_shards);
        for (int i = 0; i < n_shards; ++i) c->inboxes[i] = c->shards[i]->inbox;
        rc = c->run_all([c](Shard& s) {
            ENG(s, mppi_xchg_connect(s.eng, nullptr, c->inboxes.data()));
            return (int)MPPI_OK;
        });
        if (rc) return rc;
    }
    if (transport == MPPI_XPORT_COLLECTIVE) {
        std::vector<ncclComm_t> comms(n_shards);
        const ncclResult_t r = ncclCommInitAll(comms.data(), n_shards, devs.data());
        if (r != ncclSuccess)
            return fail(MPPI_ENODEV, "ncclCommInitAll over %d devices failed: %s", n_shards,
                        ncclGetErrorString(r));
        for (int i = 0; i < n_shards; ++i) c->shards[i]->comm = comms[i];
    }
    if (verbose)
        printf("mppi_gpu_amd: %d samples over %d shards, transport %d\n", K, n_shards, transport);
    return MPPI_OK;
}

int mppi_sharded_set_data(mppi_sharded* c, const float* x0, const float* u, const float* goal,
                          const float* w)
{
    if (!c || !x0 || !u || !goal || !w) return fail(MPPI_EINVAL, "null argument");
    int rc = c->run_all([=](Shard& s) {
        ENG(s, mppi_set_data(s.eng, x0, u, goal, w));
        return (int)MPPI_OK;
    });
    if (rc) return rc;
    for (int i = 0; i < c->S; ++i) c->x0[i] = x0[i];
    c->n_solves = 0;
    c->data_set = true;
    return MPPI_OK;
}

int mppi_sharded_set_x(mppi_sharded* c, const float* x0)
{
    if (!c || !x0) return fail(MPPI_EINVAL, "null argument");
    // host-only in the engine (the state travels by value with the next launch): no worker needed
    for (Shard* s : c->shards)
        if (mppi_set_x(s->eng, x0) != MPPI_OK) return fail(MPPI_EINVAL, "%s", mppi_last_error());
    for (int i = 0; i < c->S; ++i) c->x0[i] = x0[i];
    return MPPI_OK;
}

int mppi_sharded_get_x(mppi_sharded* c, float* x0)
{
    if (!c || !x0) return fail(MPPI_EINVAL, "null argument");
    for (int i = 0; i < c->S; ++i) x0[i] = c->x0[i];
    return MPPI_OK;
}

int mppi_sharded_get_u(mppi_sharded* c, float* u)
{
    if (!c || !u) return fail(MPPI_EINVAL, "null argument");
    int rc = mppi_sharded_sync_act(c, nullptr);
    if (rc) return rc;
    return c->run_all([=](Shard& s) {
        if (s.index == 0) ENG(s, mppi_get_u(s.eng, u));
        return (int)MPPI_OK;
    });
}

static int solve_async_impl(mppi_sharded* c, bool flush_too)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    if (!c->data_set) return fail(MPPI_ESTATE, "solve before mppi_sharded_set_data");
    const int par = (int)(c->n_solves & 1ull);
    int rc = MPPI_OK;
    switch (c->transport) {
    case MPPI_XPORT_DIRECT:
        rc = c->run_all([flush_too](Shard& s) {
            ENG(s, mppi_solve_exchange_async(s.eng, s.stream));
            // (a blocking get_act: launch the held-back exchange in the same hand-over -- every
            //  shard must have launched its own before anybody waits)
            if (flush_too) ENG(s, mppi_flush_async(s.eng));
            return (int)MPPI_OK;
        });
        break;
    case MPPI_XPORT_COLLECTIVE:
        rc = c->run_all([c, par](Shard& s) {
            ENG(s, mppi_solve_local_async(s.eng, s.d_partial[par], s.stream));
            const ncclResult_t r = ncclAllGather(s.d_partial[par], s.d_gathered, (size_t)c->L,
                                                 ncclFloat, s.comm, s.stream);
            if (r != ncclSuccess) return nccl_fail(s, r, "ncclAllGather");
            ENG(s, mppi_solve_finish_async(s.eng, s.d_gathered, c->n, s.stream));
            return (int)MPPI_OK;
        });
        break;
    default:
        // COPY, phase 1: rank-local part + an event behind it
        rc = c->run_all([par](Shard& s) {
            ENG(s, mppi_solve_local_async(s.eng, s.d_partial[par], s.stream));
            HIPW(s, hipEventRecord(s.ev_partial[par], s.stream));
            return (int)MPPI_OK;
        });
        if (rc) break;
        // phase 2 (every event is recorded by now): pull all partials, finish
        rc = c->run_all([c, par](Shard& s) {
            for (Shard* o : c->shards) {
                if (o != &s) HIPW(s, hipStreamWaitEvent(s.stream, o->ev_partial[par], 0));
                HIPW(s, hipMemcpyPeerAsync(s.d_gathered + (size_t)o->index * c->L, s.device,
                                           o->d_partial[par], o->device,
                                           (size_t)c->L * sizeof(float), s.stream));
            }
            ENG(s, mppi_solve_finish_async(s.eng, s.d_gathered, c->n, s.stream));
            return (int)MPPI_OK;
        });
        break;
    }
    if (rc) return rc;
    c->n_solves += 1;
    return MPPI_OK;
}

int mppi_sharded_solve_async(mppi_sharded* c) { return solve_async_impl(c, false); }

static int sync_act_impl(mppi_sharded* c, float* next_act, bool flushed)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    int rc;
    if (c->transport == MPPI_XPORT_DIRECT && !flushed) {
        // a held-back exchange waits for the peers' words: every shard launches its own before
        // anybody waits (mppi_flush_async)
        rc = c->run_all([](Shard& s) {
            ENG(s, mppi_flush_async(s.eng));
            return (int)MPPI_OK;
        });
        if (rc) return rc;
    }
    // a blocking get_act waits for the action words only (mppi_wait_act polls them: no
    // hipStreamSynchronize wake-up); everything else is a full synchronisation
    rc = c->run_all([flushed](Shard& s) {
        if (flushed) ENG(s, mppi_wait_act(s.eng, s.act));
        else ENG(s, mppi_sync_act(s.eng, s.act));
        return (int)MPPI_OK;
    });
    if (rc) return rc;
    if (c->n_solves > 0)
        for (Shard* s : c->shards)
            if (memcmp(s->act, c->shards[0]->act, (size_t)c->A * sizeof(float)) != 0)
                return fail(MPPI_ESTATE, "shard %d arrived at another action than shard 0", s->index);
    if (next_act) memcpy(next_act, c->shards[0]->act, (size_t)c->A * sizeof(float));
    return MPPI_OK;
}

int mppi_sharded_sync_act(mppi_sharded* c, float* next_act) { return sync_act_impl(c, next_act, false); }

int mppi_sharded_get_act(mppi_sharded* c, float* next_act)
{
    if (!c || !next_act) return fail(MPPI_EINVAL, "null argument");
    int rc = solve_async_impl(c, true);      // two hand-overs to the workers per call, not three
    if (rc) return rc;
    return sync_act_impl(c, next_act, true);
}

int mppi_sharded_get_inf(mppi_sharded* c, float* x_all, float* u, float* noise, float* cost,
                         float* beta, float* nabla, float* weight)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    int rc = mppi_sharded_sync_act(c, nullptr);
    if (rc) return rc;
    return c->run_all([=](Shard& s) {
        const size_t k0 = (size_t)s.k_begin;
        const bool first = s.index == 0;
        ENG(s, mppi_get_inf(s.eng, x_all ? x_all + k0 * (c->T + 1) * c->S : nullptr,
                            (first ? u : nullptr), noise ? noise + k0 * c->TA : nullptr,
                            cost ? cost + k0 : nullptr, first ? beta : nullptr,
                            first ? nabla : nullptr, weight ? weight + k0 : nullptr));
        return (int)MPPI_OK;
    });
}

int mppi_sharded_get_data(mppi_sharded* c, float* x_all, float* noise)
{
    return mppi_sharded_get_inf(c, x_all, nullptr, noise, nullptr, nullptr, nullptr, nullptr);
}

int mppi_sharded_set_params(mppi_sharded* c, float lambda, const float* sigma, const float* inv_s)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    int rc = mppi_sharded_sync_act(c, nullptr);
    if (rc) return rc;
    return c->run_all([=](Shard& s) {
        ENG(s, mppi_set_params(s.eng, lambda, sigma, inv_s));
        return (int)MPPI_OK;
    });
}

int mppi_sharded_set_seed(mppi_sharded* c, unsigned long long seed)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    return c->run_all([=](Shard& s) {
        ENG(s, mppi_set_seed(s.eng, seed));
        return (int)MPPI_OK;
    });
}

int mppi_sharded_set_noise(mppi_sharded* c, const float* noise)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    int rc = mppi_sharded_sync_act(c, nullptr);
    if (rc) return rc;
    return c->run_all([=](Shard& s) {
        ENG(s, mppi_set_noise(s.eng, noise ? noise + (size_t)s.k_begin * c->TA : nullptr));
        return (int)MPPI_OK;
    });
}

int mppi_sharded_set_action_limit(mppi_sharded* c, const float* max_a)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    int rc = mppi_sharded_sync_act(c, nullptr);
    if (rc) return rc;
    return c->run_all([=](Shard& s) {
        ENG(s, mppi_set_action_limit(s.eng, max_a));
        return (int)MPPI_OK;
    });
}

int mppi_sharded_set_timeout(mppi_sharded* c, double seconds)
{
    if (!c) return fail(MPPI_EINVAL, "null controller");
    if (c->transport != MPPI_XPORT_DIRECT) return MPPI_OK;
    return c->run_all([=](Shard& s) {
        ENG(s, mppi_xchg_set_timeout(s.eng, seconds));
        return (int)MPPI_OK;
    });
}

int mppi_sharded_n_shards(const mppi_sharded* c) { return c ? c->n : 0; }
int mppi_sharded_transport(const mppi_sharded* c) { return c ? c->transport : -1; }

int mppi_sharded_shard_info(const mppi_sharded* c, int i, long long out[3])
{
    if (!c || !out || i < 0 || i >= c->n) return fail(MPPI_EINVAL, "bad shard index");
    out[0] = c->shards[i]->k_begin;
    out[1] = c->shards[i]->k_end;
    out[2] = c->shards[i]->device;
    return MPPI_OK;
}

mppi_engine* mppi_sharded_engine(mppi_sharded* c, int i)
{
    return (c && i >= 0 && i < c->n) ? c->shards[i]->eng : nullptr;
}

}  // extern "C"

// ---- the C++ class --------------------------------------------------------------------------------
// reference include/mppi_utils.hpp:19-25 (CUDA_CALL_CONST): print file:line:code, exit(1)
#define MPPI_SH_CALL(x)                                                              \
    do {                                                                             \
        int err__ = (x);                                                             \
        if (err__ != MPPI_OK) {                                                      \
            printf("API error failed %s:%d Returned: %d (%s)\n", __FILE__, __LINE__, \
                   err__, mppi_sharded_last_error());                                \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

static int transport_by_name(const char* name)
{
    if (!name || !strcmp(name, "collective") || !strcmp(name, "rccl")) return MPPI_XPORT_COLLECTIVE;
    if (!strcmp(name, "direct")) return MPPI_XPORT_DIRECT;
    if (!strcmp(name, "copy")) return MPPI_XPORT_COPY;
    return -1;
}

ShardedPointMassModel::ShardedPointMassModel(int nb_sim, int steps, float dt, int state_dim,
                                             int act_dim, bool verbose, int n_gpus,
                                             const char* transport, const int* devices)
    : impl_(nullptr)
{
    std::cout << "Allocating Space... : " << std::flush;
    MPPI_SH_CALL(mppi_sharded_create(nb_sim, steps, dt, state_dim, act_dim, verbose ? 1 : 0, n_gpus,
                                     devices, transport_by_name(transport), &impl_));
    std::cout << "Done" << std::endl;
}

ShardedPointMassModel::~ShardedPointMassModel() { mppi_sharded_destroy(impl_); }

void ShardedPointMassModel::get_act(float* next_act) { MPPI_SH_CALL(mppi_sharded_get_act(impl_, next_act)); }
void ShardedPointMassModel::memcpy_set_data(float* x, float* u, float* goal, float* w)
{
    std::cout << "Setting inital state of the sims... : " << std::flush;
    MPPI_SH_CALL(mppi_sharded_set_data(impl_, x, u, goal, w));
    std::cout << "Done" << std::endl;
}
void ShardedPointMassModel::get_x(float* x) { MPPI_SH_CALL(mppi_sharded_get_x(impl_, x)); }
void ShardedPointMassModel::memcpy_get_data(float* x_all, float* e)
{
    MPPI_SH_CALL(mppi_sharded_get_data(impl_, x_all, e));
}
void ShardedPointMassModel::get_inf(float* x, float* u, float* e, float* cost, float* beta,
                                    float* nabla, float* weight)
{
    std::cout << "Collect informations: " << std::endl;
    MPPI_SH_CALL(mppi_sharded_get_inf(impl_, x, u, e, cost, beta, nabla, weight));
}
void ShardedPointMassModel::set_x(float* x) { MPPI_SH_CALL(mppi_sharded_set_x(impl_, x)); }
void ShardedPointMassModel::get_u(float* u) { MPPI_SH_CALL(mppi_sharded_get_u(impl_, u)); }
void ShardedPointMassModel::solve_async() { MPPI_SH_CALL(mppi_sharded_solve_async(impl_)); }
void ShardedPointMassModel::sync_act(float* next_act) { MPPI_SH_CALL(mppi_sharded_sync_act(impl_, next_act)); }
void ShardedPointMassModel::set_params(float lambda, const float* sigma, const float* inv_s)
{
    MPPI_SH_CALL(mppi_sharded_set_params(impl_, lambda, sigma, inv_s));
}
void ShardedPointMassModel::set_seed(unsigned long long seed) { MPPI_SH_CALL(mppi_sharded_set_seed(impl_, seed)); }
void ShardedPointMassModel::set_noise(const float* e) { MPPI_SH_CALL(mppi_sharded_set_noise(impl_, e)); }
void ShardedPointMassModel::set_action_limit(const float* max_a)
{
    MPPI_SH_CALL(mppi_sharded_set_action_limit(impl_, max_a));
}
int ShardedPointMassModel::n_shards() const { return mppi_sharded_n_shards(impl_); }
const char* ShardedPointMassModel::transport() const
{
    static const char* names[] = {"collective", "direct", "copy"};
    const int t = mppi_sharded_transport(impl_);
    return (t >= 0 && t <= 2) ? names[t] : "?";
}
