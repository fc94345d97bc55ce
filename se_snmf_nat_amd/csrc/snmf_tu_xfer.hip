// snmf_tu_xfer.hip -- host <-> device movement of the matrices that cross the drop-in boundary (snmf_internal.h).
//
// The reference's caller hands over MATLAB arrays: pageable host memory, fp64, column-major (src/sparse_nmf.m:71-72; SURVEY.md
// section 8b).  At BASELINE config 4 that is 410 MB per spectrogram, and one pageable hipMemcpy of it followed by a conversion
// kernel and a stream sync left the engine idle for 88 % of a run_basis_DNMF call (round 3).  Here a matrix moves as a
// PIPELINE of column chunks through three pinned bounce buffers:
//
//   host -> device   worker threads copy chunk i from the caller's array into pinned buffer i % 3, narrowing fp64 to fp32 on the
//                    way when the device copy is fp32 (the same round-to-nearest k_pack applied on the device: bit-identical
//                    results, half the PCIe bytes)  ||  the copy stream moves chunk i-1 to its device staging buffer  ||
//                    the engine's stream runs k_pack of chunk i-2 into the padded device layout (V floor of :169 included).
//                    The call returns when the last chunk has LEFT THE CALLER'S ARRAY (it is in pinned memory): the tail of
//                    the pipeline runs under whatever the caller does next, ordered on the engine's stream.
//   device -> host   k_unpack of chunk i into staging  ||  copy stream: staging -> pinned  ||  worker threads widen / copy chunk
//                    i-2 into the caller's array.  Returns when the array is complete.
//
// Staging is sized to the chunk (kChunkBytes), not to the matrix.  Counters (bytes, wall seconds, host-side copy seconds) are
// kept per context for scripts/bench_dropin.py (snmf_ctx_xfer_stats).
#include "snmf_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <thread>

namespace {

constexpr size_t kChunkBytes = (size_t)16 << 20;  // per bounce buffer (PCIe efficiency is flat above a few MB)
constexpr int kNB = 3;                            // bounce buffers: fill | DMA | pack
constexpr size_t kPoolMinBytes = (size_t)1 << 20; // below this a chunk is copied by the calling thread alone

// ---- a small process-wide worker pool for the host-side copies --------------------------------------------------------
// parallel_for may be called from several host threads at once (one per rank of the multi-device entry): every call is a
// job of its own in a shared queue; the caller works on its job too and returns when all of its items are done.
class Pool {
public:
    static Pool& get() {
        static Pool p;
        return p;
    }
    int threads() const { return (int)workers_.size() + 1; }
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (n == 1 || workers_.empty()) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->n = n;
        job->fn = &fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work_on(*job);
        std::unique_lock<std::mutex> lk(job->mu);
        job->cv.wait(lk, [&] { return job->done.load(std::memory_order_acquire) >= n; });
    }

private:
    struct Job {
        int n = 0;
        const std::function<void(int)>* fn = nullptr;
        std::atomic<int> next{0}, done{0};
        std::mutex mu;
        std::condition_variable cv;
    };
    Pool() {
        int n = (int)std::thread::hardware_concurrency();
        n = std::max(1, std::min(n, 16));  // a GPU's share of the host (the boxes give 16 cores per GPU)
        if (const char* e = getenv("SNMF_HOST_THREADS")) n = std::max(1, std::min(atoi(e), 64));
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    static void work_on(Job& j) {
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) return;
            (*j.fn)(i);
            if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 >= j.n) {
                std::lock_guard<std::mutex> lk(j.mu);
                j.cv.notify_all();
            }
        }
    }
    void loop() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return quit_ || !jobs_.empty(); });
                if (quit_) return;
                job = jobs_.front();
                if (job->next.load(std::memory_order_relaxed) >= job->n) {  // exhausted: retire it from the queue
                    jobs_.pop_front();
                    continue;
                }
            }
            work_on(*job);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> jobs_;
    std::vector<std::thread> workers_;
    bool quit_ = false;
};

template <typename TS, typename TD>
inline void copy_cvt(const TS* __restrict__ s, TD* __restrict__ d, size_t n) {
    if constexpr (std::is_same<TS, TD>::value) memcpy(d, s, n * sizeof(TS));
    else
        for (size_t i = 0; i < n; ++i) d[i] = (TD)s[i];
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct HostXfer {
    void* pin[kNB] = {nullptr, nullptr, nullptr};
    void* dev[kNB] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_dma[kNB] = {nullptr, nullptr, nullptr};   // DMA of the chunk in buffer b has completed
    hipEvent_t ev_krn[kNB] = {nullptr, nullptr, nullptr};   // k_pack / k_unpack of the chunk in buffer b has completed
    bool dma_pending[kNB] = {false, false, false}, krn_pending[kNB] = {false, false, false};
    hipStream_t cs = nullptr;  // copy stream
};

static int xfer_get(snmf_ctx* c, HostXfer** out) {
    if (c->xf) {
        *out = c->xf;
        return SNMF_OK;
    }
    HostXfer* x = new HostXfer();
    auto bail = [&](const char* what, hipError_t e) {
        const int rc = fail(e == hipErrorOutOfMemory ? SNMF_ERR_NOMEM : SNMF_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
        c->xf = x;
        xfer_destroy(c);
        return rc;
    };
    hipError_t e = hipStreamCreateWithFlags(&x->cs, hipStreamNonBlocking);
    if (e != hipSuccess) return bail("hipStreamCreate (copy stream)", e);
    for (int b = 0; b < kNB; ++b) {
        if ((e = hipHostMalloc(&x->pin[b], kChunkBytes, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc (bounce buffer)", e);
        if ((e = hipMalloc(&x->dev[b], kChunkBytes)) != hipSuccess) return bail("hipMalloc (chunk staging)", e);
        if ((e = hipEventCreateWithFlags(&x->ev_dma[b], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&x->ev_krn[b], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    c->xf = x;
    *out = x;
    return SNMF_OK;
}

void xfer_destroy(snmf_ctx* c) {
    HostXfer* x = c->xf;
    if (!x) return;
    if (x->cs) hipStreamSynchronize(x->cs);
    for (int b = 0; b < kNB; ++b) {
        if (x->pin[b]) hipHostFree(x->pin[b]);
        if (x->dev[b]) hipFree(x->dev[b]);
        if (x->ev_dma[b]) hipEventDestroy(x->ev_dma[b]);
        if (x->ev_krn[b]) hipEventDestroy(x->ev_krn[b]);
    }
    if (x->cs) hipStreamDestroy(x->cs);
    delete x;
    c->xf = nullptr;
}

int xfer_sync(snmf_ctx* c) {
    if (c->xf && c->xf->cs) HIP_TRY(hipStreamSynchronize(c->xf->cs));
    return SNMF_OK;
}

// columns [c0, c0 + nc) of a column-major host matrix <-> a tight [nc][rows] block of pinned memory
template <typename TH, typename TS>
static void host_to_pinned(const TH* src, int64_t ld, int rows, size_t c0, size_t nc, TS* pin) {
    const size_t bytes = nc * (size_t)rows * sizeof(TH);
    auto cols = [&](size_t a, size_t b) {
        if (ld == rows) copy_cvt(src + a * (size_t)ld + c0 * (size_t)ld, pin + (a) * (size_t)rows, (b - a) * (size_t)rows);
        else
            for (size_t c = a; c < b; ++c) copy_cvt(src + (c0 + c) * (size_t)ld, pin + c * (size_t)rows, (size_t)rows);
    };
    Pool& pool = Pool::get();
    if (bytes < kPoolMinBytes || pool.threads() == 1) {
        cols(0, nc);
        return;
    }
    const int parts = (int)std::min<size_t>(nc, (size_t)pool.threads() * 2);
    pool.parallel_for(parts, [&](int i) { cols(nc * (size_t)i / parts, nc * (size_t)(i + 1) / parts); });
}
template <typename TS, typename TH>
static void pinned_to_host(const TS* pin, int rows, size_t c0, size_t nc, TH* dst, int64_t ld) {
    const size_t bytes = nc * (size_t)rows * sizeof(TH);
    auto cols = [&](size_t a, size_t b) {
        if (ld == rows) copy_cvt(pin + a * (size_t)rows, dst + (c0 + a) * (size_t)ld, (b - a) * (size_t)rows);
        else
            for (size_t c = a; c < b; ++c) copy_cvt(pin + c * (size_t)rows, dst + (c0 + c) * (size_t)ld, (size_t)rows);
    };
    Pool& pool = Pool::get();
    if (bytes < kPoolMinBytes || pool.threads() == 1) {
        cols(0, nc);
        return;
    }
    const int parts = (int)std::min<size_t>(nc, (size_t)pool.threads() * 2);
    pool.parallel_for(parts, [&](int i) { cols(nc * (size_t)i / parts, nc * (size_t)(i + 1) / parts); });
}

// staged element type: fp32 whenever the device side is fp32 (the host narrows / widens), else the host's own type
template <typename TH, typename TD>
using staged_t = typename std::conditional<std::is_same<TD, float>::value, float, TH>::type;

template <typename TIn, typename TDst>
int xfer_pack_in(snmf_ctx* ctx, const TIn* src, int64_t ld, int rows, int cols, TDst* dst, int rowsP, int colsP, bool do_floor) {
    using TS = staged_t<TIn, TDst>;
    HostXfer* x = nullptr;
    SN_TRY(xfer_get(ctx, &x));
    hipStream_t st = ctx->stream;
    const double t0 = now_s();
    double t_host = 0.0;
    const size_t cpc = std::max<size_t>(1, kChunkBytes / ((size_t)rows * sizeof(TS)));  // columns per chunk
    if ((size_t)rows * sizeof(TS) > kChunkBytes) return fail(SNMF_ERR_UNSUPPORTED, "a column of %d rows does not fit a transfer chunk", rows);
    int i = 0;
    for (size_t c0 = 0; c0 < (size_t)cols; c0 += cpc, ++i) {
        const int b = i % kNB;
        const size_t nc = std::min(cpc, (size_t)cols - c0);
        if (x->dma_pending[b]) {  // the DMA that last read pinned buffer b
            HIP_TRY(hipEventSynchronize(x->ev_dma[b]));
            x->dma_pending[b] = false;
        }
        const double th = now_s();
        host_to_pinned<TIn, TS>(src, ld, rows, c0, nc, (TS*)x->pin[b]);
        t_host += now_s() - th;
        if (x->krn_pending[b]) HIP_TRY(hipStreamWaitEvent(x->cs, x->ev_krn[b], 0));  // the kernel that last used staging buffer b
        HIP_TRY(hipMemcpyAsync(x->dev[b], x->pin[b], nc * (size_t)rows * sizeof(TS), hipMemcpyHostToDevice, x->cs));
        HIP_TRY(hipEventRecord(x->ev_dma[b], x->cs));
        x->dma_pending[b] = true;
        HIP_TRY(hipStreamWaitEvent(st, x->ev_dma[b], 0));
        const size_t n = (size_t)rowsP * nc;
        hipLaunchKernelGGL((snmf::k_pack<TS, TDst>), dim3(grid_for(n)), dim3(256), 0, st, (const TS*)x->dev[b], (int64_t)rows, rows, (int)nc,
                           dst + c0 * (size_t)rowsP, rowsP, (int)nc, snmf::kFlr, do_floor ? 1 : 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(x->ev_krn[b], st));
        x->krn_pending[b] = true;
    }
    if (colsP > cols) HIP_TRY(hipMemsetAsync(dst + (size_t)cols * rowsP, 0, (size_t)(colsP - cols) * rowsP * sizeof(TDst), st));
    ctx->xs.h2d_bytes += (double)rows * cols * sizeof(TIn);
    ctx->xs.h2d_wall += now_s() - t0;
    ctx->xs.h2d_host += t_host;
    ctx->xs.h2d_calls += 1;
    return SNMF_OK;
}

template <typename TOut, typename TSrc>
int xfer_unpack_out(snmf_ctx* ctx, const TSrc* src, int rowsP, int rows, int cols, TOut* dst, int64_t ld) {
    // staged as fp32 when either side is fp32 (the device narrows an fp64 source, the host widens into an fp64 array)
    using TS = typename std::conditional<std::is_same<TSrc, float>::value || std::is_same<TOut, float>::value, float, double>::type;
    HostXfer* x = nullptr;
    SN_TRY(xfer_get(ctx, &x));
    hipStream_t st = ctx->stream;
    // the data must exist before it can move: waiting for the producer here (instead of inside the first chunk's wait) costs
    // nothing and keeps the transfer counters free of solve time
    HIP_TRY(hipStreamSynchronize(st));
    const double t0 = now_s();
    double t_host = 0.0;
    if ((size_t)rows * sizeof(TS) > kChunkBytes) return fail(SNMF_ERR_UNSUPPORTED, "a column of %d rows does not fit a transfer chunk", rows);
    const size_t cpc = std::max<size_t>(1, kChunkBytes / ((size_t)rows * sizeof(TS)));
    const int n_chunks = (int)(((size_t)cols + cpc - 1) / cpc);
    auto drain = [&](int i) -> int {  // chunk i: pinned -> the caller's array
        const int b = i % kNB;
        const size_t c0 = (size_t)i * cpc, nc = std::min(cpc, (size_t)cols - c0);
        HIP_TRY(hipEventSynchronize(x->ev_dma[b]));
        x->dma_pending[b] = false;
        const double th = now_s();
        pinned_to_host<TS, TOut>((const TS*)x->pin[b], rows, c0, nc, dst, ld);
        t_host += now_s() - th;
        return SNMF_OK;
    };
    for (int i = 0; i < n_chunks; ++i) {
        const int b = i % kNB;
        const size_t c0 = (size_t)i * cpc, nc = std::min(cpc, (size_t)cols - c0);
        if (i >= kNB) SN_TRY(drain(i - kNB));  // frees pinned buffer b (and its DMA has left staging buffer b)
        else if (x->dma_pending[b]) {          // an earlier transfer's DMA still owns the buffers
            HIP_TRY(hipEventSynchronize(x->ev_dma[b]));
            x->dma_pending[b] = false;
        }
        const size_t n = (size_t)rows * nc;
        hipLaunchKernelGGL((snmf::k_unpack<TS, TSrc>), dim3(grid_for(n)), dim3(256), 0, st, src + c0 * (size_t)rowsP, rowsP, rows, (int)nc,
                           (TS*)x->dev[b], (int64_t)rows);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(x->ev_krn[b], st));
        x->krn_pending[b] = true;
        HIP_TRY(hipStreamWaitEvent(x->cs, x->ev_krn[b], 0));
        HIP_TRY(hipMemcpyAsync(x->pin[b], x->dev[b], n * sizeof(TS), hipMemcpyDeviceToHost, x->cs));
        HIP_TRY(hipEventRecord(x->ev_dma[b], x->cs));
        x->dma_pending[b] = true;
    }
    for (int i = std::max(0, n_chunks - kNB); i < n_chunks; ++i) SN_TRY(drain(i));
    ctx->xs.d2h_bytes += (double)rows * cols * sizeof(TOut);
    ctx->xs.d2h_wall += now_s() - t0;
    ctx->xs.d2h_host += t_host;
    ctx->xs.d2h_calls += 1;
    return SNMF_OK;
}

// the combinations the library uses
#define XFER_IN(TI, TD) template int xfer_pack_in<TI, TD>(snmf_ctx*, const TI*, int64_t, int, int, TD*, int, int, bool)
XFER_IN(double, float);
XFER_IN(float, float);
XFER_IN(double, double);
XFER_IN(float, double);
#define XFER_OUT(TO, TS_) template int xfer_unpack_out<TO, TS_>(snmf_ctx*, const TS_*, int, int, int, TO*, int64_t)
XFER_OUT(double, float);
XFER_OUT(float, float);
XFER_OUT(double, double);
XFER_OUT(float, double);

// Transfer counters of a context since the last reset (scripts/bench_dropin.py): out[0..7] =
//   host->device: bytes of the callers' arrays, wall seconds inside the calls, seconds of host-side copying, calls;
//   device->host: the same four.
extern "C" int snmf_ctx_xfer_stats(snmf_ctx* c, double* out8, int reset) {
    if (!c) return fail(SNMF_ERR_INVALID, "ctx is NULL");
    if (out8) {
        out8[0] = c->xs.h2d_bytes; out8[1] = c->xs.h2d_wall; out8[2] = c->xs.h2d_host; out8[3] = c->xs.h2d_calls;
        out8[4] = c->xs.d2h_bytes; out8[5] = c->xs.d2h_wall; out8[6] = c->xs.d2h_host; out8[7] = c->xs.d2h_calls;
    }
    if (reset) c->xs = XferStats{};
    return SNMF_OK;
}
