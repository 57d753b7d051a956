// bk_comm.cpp -- multi-GPU behind the C ABI: row stripes + the frame reassembly over RCCL / xGMI.
//
// every output pixel is independent, so GPU r of N owns rows [H*r/N, H*(r+1)/N), builds and keeps
// only that stripe of the lensmap, holds a full replica of the globe, warps its stripe, and ONE exchange step
// reassembles the frame.  Two ways in:
//   * bk_comm_*   one rank (= one bk_ctx on one GPU); ranks may live in different processes (one process per
//                 GPU: bench.py under torchrun) or in one.  The transport is RCCL: ncclCommInitRank from a unique id the
//                 host ships to every rank, grouped ncclSend / ncclRecv for the stripes, ncclAllReduce(MAX) for display[].
//   * bk_multi_*  N ranks driven by ONE host thread (the engine: fisheye_hip.c stays a single C process):
//                 ncclCommInitAll, every exchange posted for all ranks inside one ncclGroup.  When the device list
//                 names a device twice (a one-GPU box; RCCL refuses duplicate devices) the same schedule runs over
//                 device-to-device copies instead - that is how the tests exercise the schedule on one GPU.
// librccl is resolved with dlopen/dlsym on first use (whichever RCCL the process already loaded wins, e.g.
// PyTorch's), so libblinkyhip.so itself has no link-time dependency on it; a missing RCCL is a loud error.
//
// The reference has no counterpart (fisheye.c is single-threaded CPU code); the stripes reassemble into exactly the
// frame render_lensmap (fisheye.c:2406-2424) writes.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <array>
#include <cstring>
#include <memory>
#include <thread>

#include "bk_internal.h"

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void *h = nullptr;
    if (!dlsym(RTLD_DEFAULT, "ncclSend")) {                 // not in the process yet: load the system RCCL
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { r.error = std::string("RCCL is not available: ") + dlerror(); return r; }
    }
    auto sym = [&](const char *n) -> void * {
        void *p = h ? dlsym(h, n) : dlsym(RTLD_DEFAULT, n);
        if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + n;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.error.empty();
    return r;
}

thread_local std::string g_comm_create_error;

}  // namespace

// One rank's share of an exchange: sends to / receives from peers, and the copy of its own rows.
struct BkOp {
    enum Kind { SEND, RECV, LOCAL } kind;
    int peer;
    const uint8_t *src;      // SEND, LOCAL
    uint8_t *dst;            // RECV, LOCAL
    size_t bytes;
};

struct bk_comm {
    bk_ctx *ctx = nullptr;
    int nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;       // null: a member of a copy-transport bk_multi, or nranks == 1
    bool owns_comm = true;
    int *d_flags = nullptr;          // [6] display flags on the device (all-reduce buffer)
    // exchanges run on their own stream so that the next batch's warp (context stream) overlaps the stripes' travel
    hipStream_t xstream = nullptr;
    hipEvent_t ev_ready = nullptr;                  // context stream: the stripe has been warped
    hipEvent_t ev_copied = nullptr;                 // copy transport: this rank's outgoing copies are done
    hipEvent_t ev_done[BK_COMM_SLOTS] = {};         // exchange stream: the exchange last posted with that slot has finished
    std::string err;

    // stripe bounds: rank r owns rows [bounds[r], bounds[r+1]); equal shares (multigpu.stripe_bounds) until bk_comm_rebalance /
    // bk_multi_rebalance cut them by work
    std::vector<int> bounds;

    int fail(int code, const std::string &m) { err = m; return code; }
    void equal_bounds() { bounds.resize((size_t)nranks + 1); for (int r = 0; r <= nranks; ++r) bounds[(size_t)r] = (int)((long long)ctx->H * r / nranks); }
    int row0(int r) const { return bounds[(size_t)r]; }
    int rows(int r) const { return row0(r + 1) - row0(r); }
};

#define BK_NCCL(c, expr)                                                                            \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return (c)->fail(BK_E_HIP, std::string(#expr " failed: ") + rccl().GetErrorString(r_)); \
    } while (0)
#define BK_CHIP(c, expr)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return (c)->fail(BK_E_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_)); \
    } while (0)

namespace {

// frame f of a batch: the rows of rank r inside a tight frame buffer [slot][H][W]
uint8_t *frame_rows(const bk_comm *c, void *frames, size_t frame_stride, int slot, int r)
{
    return (uint8_t *)frames + (size_t)slot * frame_stride + (size_t)c->row0(r) * c->ctx->W;
}

// every frame onto `root` (what a single display needs): frames_dev is read on the root only
void ops_gather(const bk_comm *c, const void *stripe, int nframes, int root, void *frames, size_t frame_stride, std::vector<BkOp> *ops)
{
    const size_t mine = (size_t)c->rows(c->rank) * c->ctx->W;
    for (int f = 0; f < nframes; ++f) {
        const uint8_t *src = (const uint8_t *)stripe + (size_t)f * mine;
        if (c->rank != root) { ops->push_back({BkOp::SEND, root, src, nullptr, mine}); continue; }
        ops->push_back({BkOp::LOCAL, root, src, frame_rows(c, frames, frame_stride, f, root), mine});
        for (int s = 0; s < c->nranks; ++s)
            if (s != root) ops->push_back({BkOp::RECV, s, nullptr, frame_rows(c, frames, frame_stride, f, s), (size_t)c->rows(s) * c->ctx->W});
    }
}

// batches: frame f is reassembled on rank f % N (a gather whose root rotates), slot f / N of that rank's buffer;
// all N*(N-1) links carry stripes at once and every GPU ends up holding whole frames
void ops_rotating(const bk_comm *c, const void *stripe, int nframes, void *frames, size_t frame_stride, std::vector<BkOp> *ops)
{
    const size_t mine = (size_t)c->rows(c->rank) * c->ctx->W;
    for (int f = 0; f < nframes; ++f) {
        const int owner = f % c->nranks, slot = f / c->nranks;
        const uint8_t *src = (const uint8_t *)stripe + (size_t)f * mine;
        if (owner != c->rank) { ops->push_back({BkOp::SEND, owner, src, nullptr, mine}); continue; }
        ops->push_back({BkOp::LOCAL, owner, src, frame_rows(c, frames, frame_stride, slot, owner), mine});
        for (int s = 0; s < c->nranks; ++s)
            if (s != owner) ops->push_back({BkOp::RECV, s, nullptr, frame_rows(c, frames, frame_stride, slot, s), (size_t)c->rows(s) * c->ctx->W});
    }
}

// post one rank's ops on its communicator and stream (inside the caller's ncclGroup when there are several ranks)
int post_rccl(bk_comm *c, const std::vector<BkOp> &ops)
{
    Rccl &R = rccl();
    for (const BkOp &o : ops) {
        if (!o.bytes) continue;
        if (o.kind == BkOp::LOCAL) BK_CHIP(c, hipMemcpyAsync(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice, c->xstream));
        else if (o.kind == BkOp::SEND) BK_NCCL(c, R.Send(o.src, o.bytes, ncclUint8, o.peer, c->comm, c->xstream));
        else BK_NCCL(c, R.Recv(o.dst, o.bytes, ncclUint8, o.peer, c->comm, c->xstream));
    }
    return BK_OK;
}

int check_args(bk_comm *c, const void *stripe, int nframes, int slot)
{
    if (!c || !c->ctx) return BK_E_INVALID;
    if (!stripe || nframes < 1 || slot < 0 || slot >= BK_COMM_SLOTS) return c->fail(BK_E_INVALID, "exchange: bad stripe pointer / frame count / slot");
    if (c->ctx->row0 != c->row0(c->rank) || c->ctx->row1 != c->row0(c->rank + 1))
        return c->fail(BK_E_STATE, "exchange: the context's rows are not this rank's stripe (bk_resize after bk_comm_create? call bk_comm_restripe)");
    return BK_OK;
}

}  // namespace

// ---- rank-level API ------------------------------------------------------------------------------------------

extern "C" int bk_comm_unique_id(uint8_t id[BK_COMM_ID_BYTES])
{
    Rccl &R = rccl();
    if (!R.ok) { g_comm_create_error = R.error; return BK_E_STATE; }
    static_assert(BK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    const ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) { g_comm_create_error = std::string("ncclGetUniqueId: ") + R.GetErrorString(r); return BK_E_HIP; }
    memcpy(id, u.internal, BK_COMM_ID_BYTES);
    return BK_OK;
}

static bk_comm *comm_shell(bk_ctx *ctx, int nranks, int rank)
{
    if (!ctx || ctx->device < 0 || nranks < 1 || rank < 0 || rank >= nranks) { g_comm_create_error = "bk_comm_create: bad context / rank"; return nullptr; }
    if (ctx->H <= 0) { g_comm_create_error = "bk_comm_create: call bk_resize first"; return nullptr; }
    if (nranks > ctx->H) { g_comm_create_error = "bk_comm_create: more ranks than output rows"; return nullptr; }
    std::unique_ptr<bk_comm> c(new bk_comm());
    c->ctx = ctx; c->nranks = nranks; c->rank = rank;
    c->equal_bounds();
    bool ok = hipSetDevice(ctx->device) == hipSuccess && hipMalloc((void **)&c->d_flags, BK_MAX_PLATES * sizeof(int)) == hipSuccess &&
              hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < BK_COMM_SLOTS; ++i) ok = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g_comm_create_error = "bk_comm_create: stream / event / buffer creation failed";
        bk_comm_destroy(c.release());
        return nullptr;
    }
    if (bk_set_rows(ctx, c->row0(rank), c->row0(rank + 1)) != BK_OK) { g_comm_create_error = bk_last_error(ctx); bk_comm_destroy(c.release()); return nullptr; }
    return c.release();
}

extern "C" bk_comm *bk_comm_create(bk_ctx *ctx, int nranks, int rank, const uint8_t id[BK_COMM_ID_BYTES])
{
    bk_comm *c = comm_shell(ctx, nranks, rank);
    if (!c) return nullptr;
    if (nranks == 1) return c;                                 // nothing to exchange with
    Rccl &R = rccl();
    if (!R.ok || !id) { g_comm_create_error = R.ok ? "bk_comm_create: no unique id" : R.error; bk_comm_destroy(c); return nullptr; }
    ncclUniqueId u;
    memcpy(u.internal, id, BK_COMM_ID_BYTES);
    // ncclCommInitRank blocks until EVERY rank has joined: should one of them never arrive (a rank that died, a node where RCCL's
    // bootstrap cannot reach its peers) the caller would hang for good.  The call runs on a helper thread and is given
    // BLINKY_HIP_COMM_TIMEOUT seconds (default 60); past that bk_comm_create fails with a message that says so - bench.py then
    // exchanges through torch.distributed and names that transport in config.parallelism - and the helper is left behind (it
    // owns nothing of the caller's: the communicator it may still produce is destroyed by it).
    struct Init { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr; };
    auto st = std::make_shared<Init>();
    const int device = ctx->device;
    std::thread([st, nranks, u, rank, device, &R]() {
        (void)hipSetDevice(device);
        ncclComm_t comm = nullptr;
        const ncclResult_t r = R.CommInitRank(&comm, nranks, u, rank);
        std::unique_lock<std::mutex> lock(st->m);
        if (st->abandoned) { lock.unlock(); if (r == ncclSuccess && comm) (void)R.CommDestroy(comm); return; }
        st->r = r; st->comm = comm; st->done = true;
        st->cv.notify_all();
    }).detach();
    double timeout_s = 60.0;
    if (const char *e = getenv("BLINKY_HIP_COMM_TIMEOUT")) { const double v = atof(e); if (v > 0) timeout_s = v; }
    {
        std::unique_lock<std::mutex> lock(st->m);
        if (!st->cv.wait_for(lock, std::chrono::duration<double>(timeout_s), [&] { return st->done; })) {
            st->abandoned = true;
            char msg[200];
            snprintf(msg, sizeof msg, "ncclCommInitRank did not return within %g s (BLINKY_HIP_COMM_TIMEOUT): rank %d of %d - did every rank join?", timeout_s, rank, nranks);
            g_comm_create_error = msg;
            lock.unlock();
            bk_comm_destroy(c);
            return nullptr;
        }
    }
    if (st->r != ncclSuccess) { g_comm_create_error = std::string("ncclCommInitRank: ") + R.GetErrorString(st->r); c->comm = nullptr; bk_comm_destroy(c); return nullptr; }
    c->comm = st->comm;
    return c;
}

extern "C" void bk_comm_destroy(bk_comm *c)
{
    if (!c) return;
    if (c->ctx && c->ctx->device >= 0) (void)hipSetDevice(c->ctx->device);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->comm && c->owns_comm) (void)rccl().CommDestroy(c->comm);
    (void)hipFree(c->d_flags);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
    for (hipEvent_t e : c->ev_done) if (e) (void)hipEventDestroy(e);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    delete c;
}

extern "C" const char *bk_comm_last_error(const bk_comm *c) { return c ? c->err.c_str() : g_comm_create_error.c_str(); }

extern "C" int bk_comm_stripe(const bk_comm *c, int rank, int *row0, int *row1)
{
    if (!c || rank < 0 || rank >= c->nranks) return BK_E_INVALID;
    if (row0) *row0 = c->row0(rank);
    if (row1) *row1 = c->row0(rank + 1);
    return BK_OK;
}

extern "C" int bk_comm_restripe(bk_comm *c)
{
    if (!c) return BK_E_INVALID;
    c->equal_bounds();                      // (a new size: equal shares again)
    if (int r = bk_set_rows(c->ctx, c->row0(c->rank), c->row0(c->rank + 1))) return c->fail(r, bk_last_error(c->ctx));
    return BK_OK;
}

extern "C" int bk_comm_or_display(bk_comm *c, int display[BK_MAX_PLATES])
{
    if (!c || !display) return BK_E_INVALID;
    if (c->nranks == 1 || !c->comm) return BK_OK;
    BK_CHIP(c, hipSetDevice(c->ctx->device));
    BK_CHIP(c, hipMemcpyAsync(c->d_flags, display, BK_MAX_PLATES * sizeof(int), hipMemcpyHostToDevice, c->ctx->stream));
    BK_NCCL(c, rccl().AllReduce(c->d_flags, c->d_flags, BK_MAX_PLATES, ncclInt32, ncclMax, c->comm, c->ctx->stream));
    BK_CHIP(c, hipMemcpyAsync(display, c->d_flags, BK_MAX_PLATES * sizeof(int), hipMemcpyDeviceToHost, c->ctx->stream));
    BK_CHIP(c, hipStreamSynchronize(c->ctx->stream));
    for (int i = 0; i < BK_MAX_PLATES; ++i) c->ctx->display[i] = display[i];
    return BK_OK;
}

static int run_rank(bk_comm *c, const std::vector<BkOp> &ops, int slot)
{
    BK_CHIP(c, hipSetDevice(c->ctx->device));
    if (c->nranks > 1 && !c->comm) return c->fail(BK_E_STATE, "this rank belongs to a copy-transport bk_multi: use the bk_multi_* exchange");
    BK_CHIP(c, hipEventRecord(c->ev_ready, c->ctx->stream));            // the stripe is warped ...
    BK_CHIP(c, hipStreamWaitEvent(c->xstream, c->ev_ready, 0));         // ... before it travels
    int rc;
    if (c->nranks == 1) rc = post_rccl(c, ops);                         // only LOCAL ops
    else {
        Rccl &R = rccl();
        BK_NCCL(c, R.GroupStart());
        rc = post_rccl(c, ops);
        BK_NCCL(c, R.GroupEnd());
    }
    BK_CHIP(c, hipEventRecord(c->ev_done[slot], c->xstream));
    return rc;
}

// the context's stream waits (on the device, not the host) for the exchange last posted with `slot`: call it before
// warping into the stripe buffer / reading the frame buffer that exchange used
extern "C" int bk_comm_wait(bk_comm *c, int slot)
{
    if (!c || slot < 0 || slot >= BK_COMM_SLOTS) return BK_E_INVALID;
    BK_CHIP(c, hipSetDevice(c->ctx->device));
    BK_CHIP(c, hipStreamWaitEvent(c->ctx->stream, c->ev_done[slot], 0));
    return BK_OK;
}

extern "C" int bk_comm_synchronize(bk_comm *c)
{
    if (!c) return BK_E_INVALID;
    BK_CHIP(c, hipSetDevice(c->ctx->device));
    BK_CHIP(c, hipStreamSynchronize(c->ctx->stream));
    BK_CHIP(c, hipStreamSynchronize(c->xstream));
    return BK_OK;
}

extern "C" int bk_comm_gather(bk_comm *c, const void *stripe_dev, int nframes, int root, void *frames_dev, size_t frame_stride, int slot)
{
    if (int r = check_args(c, stripe_dev, nframes, slot)) return r;
    if (root < 0 || root >= c->nranks || (c->rank == root && !frames_dev)) return c->fail(BK_E_INVALID, "bk_comm_gather: bad root / destination");
    std::vector<BkOp> ops;
    ops_gather(c, stripe_dev, nframes, root, frames_dev, frame_stride, &ops);
    return run_rank(c, ops, slot);
}

extern "C" int bk_comm_exchange_rotating(bk_comm *c, const void *stripe_dev, int nframes, void *frames_dev, size_t frame_stride, int slot)
{
    if (int r = check_args(c, stripe_dev, nframes, slot)) return r;
    if (!frames_dev && c->rank < nframes) return c->fail(BK_E_INVALID, "bk_comm_exchange_rotating: this rank owns frames but has no destination");
    std::vector<BkOp> ops;
    ops_rotating(c, stripe_dev, nframes, frames_dev, frame_stride, &ops);
    return run_rank(c, ops, slot);
}

// Stripes of equal WORK instead of equal height.  A lens that leaves part of the screen unmapped (hammer's ellipse,
// quincuncial under f_contain) gives the ranks that own the top and the bottom of the screen a fraction of the middle
// ranks' pixels; the frame is done when the slowest stripe is.  Row cost (bk_row_costs_device): for the staged apply the
// cost of the row's blocks in the block map - globe lines staged, mapped pixels, a constant per block: rows of equal
// mapped pixels differ by up to 5x in the lines they touch (cube corners against face centres) - and for the direct
// gather mapped pixels + W/32; either way > 0 for an empty row, so that those are dealt out too.  Bounds at equal shares
// of the prefix sum, multiples of 8 rows (the apply's smallest block height), at least 8 rows each.  Deterministic:
// every rank computes the same bounds from the same sums.
static std::vector<int> bounds_from_costs(const std::vector<uint32_t> &cost, int nranks)
{
    const int H = (int)cost.size();
    std::vector<uint64_t> pre((size_t)H + 1, 0);
    for (int y = 0; y < H; ++y) pre[(size_t)y + 1] = pre[(size_t)y] + cost[(size_t)y];
    if (pre[(size_t)H] == 0) for (int y = 0; y <= H; ++y) pre[(size_t)y] = (uint64_t)y;       // nothing to do anywhere: equal heights
    std::vector<int> b((size_t)nranks + 1, 0);
    b[(size_t)nranks] = H;
    const int q = H >= 16 * nranks ? 8 : 1;                 // (tiny frames: any row)
    for (int k = 1; k < nranks; ++k) {
        const uint64_t target = pre[(size_t)H] * (uint64_t)k / (uint64_t)nranks;
        int y = (int)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin());
        y = (y + q / 2) / q * q;
        const int lo = b[(size_t)k - 1] + q, hi = H - (nranks - k) * q;
        b[(size_t)k] = std::min(std::max(y, lo), hi);
    }
    return b;
}

#if BK_DEBUG_API
/* test hook (no device needed): the stripe bounds bk_comm_rebalance / bk_multi_rebalance derive from per-row costs */
extern "C" int bk_debug_stripe_bounds(const uint32_t *row_cost, int H, int W, int nranks, int *bounds_out)
{
    if (!row_cost || !bounds_out || H < 1 || W < 0 || nranks < 1 || nranks > H) return BK_E_INVALID;
    std::vector<uint32_t> cost(row_cost, row_cost + H);
    if (W > 0) for (uint32_t &c : cost) c += (uint32_t)std::max(1, W / 32);     // (as row_cost_kernel prices a row of mapped pixels)
    const std::vector<int> b = bounds_from_costs(cost, nranks);
    for (int i = 0; i <= nranks; ++i) bounds_out[i] = b[(size_t)i];
    return BK_OK;
}
#endif

// this rank's row costs, summed over the ranks, -> new bounds -> bk_set_rows.  The lensmap of the new stripe has to be
// built afterwards (bk_build; every rank), stripe buffers re-sized from bk_comm_stripe.  Collective: every rank calls it.
extern "C" int bk_comm_rebalance(bk_comm *c)
{
    if (!c || !c->ctx) return BK_E_INVALID;
    bk_ctx *ctx = c->ctx;
    if (c->nranks == 1) return BK_OK;
    if (!c->comm) return c->fail(BK_E_STATE, "this rank belongs to a copy-transport bk_multi: use bk_multi_rebalance");
    BK_CHIP(c, hipSetDevice(ctx->device));
    uint32_t *d = nullptr;
    BK_CHIP(c, hipMalloc((void **)&d, (size_t)ctx->H * sizeof(uint32_t)));
    std::vector<uint32_t> cost((size_t)ctx->H);
    int rc = bk_row_costs_device(ctx, d);
    if (rc != BK_OK) { (void)hipFree(d); return c->fail(rc, bk_last_error(ctx)); }
    const ncclResult_t nr = rccl().AllReduce(d, d, (size_t)ctx->H, ncclUint32, ncclSum, c->comm, ctx->stream);
    hipError_t e = nr == ncclSuccess ? hipMemcpyAsync(cost.data(), d, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (nr != ncclSuccess) return c->fail(BK_E_HIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
    if (e != hipSuccess) return c->fail(BK_E_HIP, std::string("bk_comm_rebalance: ") + hipGetErrorString(e));
    c->bounds = bounds_from_costs(cost, c->nranks);
    if (int r = bk_set_rows(ctx, c->row0(c->rank), c->row0(c->rank + 1))) return c->fail(r, bk_last_error(ctx));
    return BK_OK;
}

// ---- single-process group ------------------------------------------------------------------------------------

struct bk_multi {
    std::vector<bk_ctx *> ctx;
    std::vector<bk_comm *> comm;
    std::vector<hipStream_t> streams;       // one per rank, owned
    bool copy_transport = false;
    std::string err;
    int fail(int code, const std::string &m) { err = m; return code; }
};

extern "C" void bk_destroy_multi(bk_multi *m)
{
    if (!m) return;
    Rccl &R = rccl();
    for (size_t i = 0; i < m->comm.size(); ++i) {
        if (m->comm[i] && m->comm[i]->comm && R.ok) { (void)hipSetDevice(m->ctx[i]->device); (void)R.CommDestroy(m->comm[i]->comm); m->comm[i]->comm = nullptr; }
        bk_comm_destroy(m->comm[i]);
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (!m->ctx[i]) continue;
        (void)hipSetDevice(m->ctx[i]->device);
        (void)hipStreamSynchronize(m->ctx[i]->stream);
        hipStream_t s = i < m->streams.size() ? m->streams[i] : nullptr;
        bk_destroy(m->ctx[i]);
        if (s) (void)hipStreamDestroy(s);
    }
    delete m;
}

extern "C" bk_multi *bk_create_multi(int ndev, const int *devices)
{
    if (ndev < 1 || !devices) { g_comm_create_error = "bk_create_multi: bad device list"; return nullptr; }
    std::unique_ptr<bk_multi, void (*)(bk_multi *)> m(new bk_multi(), bk_destroy_multi);
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) m->copy_transport = true;       // RCCL refuses a device twice in one communicator
    if (const char *e = getenv("BLINKY_HIP_COMM")) if (!strcmp(e, "copy")) m->copy_transport = true;
    for (int i = 0; i < ndev; ++i) {
        bk_ctx *c = bk_create(devices[i]);
        if (!c) { g_comm_create_error = bk_last_error(nullptr); return nullptr; }
        m->ctx.push_back(c);
        hipStream_t s = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            g_comm_create_error = "bk_create_multi: stream creation failed";
            return nullptr;
        }
        m->streams.push_back(s);
        bk_set_stream(c, s);                                    // every rank its own stream: the stripes are warped concurrently
    }
    m->comm.assign((size_t)ndev, nullptr);
    return m.release();
}

extern "C" const char *bk_multi_last_error(const bk_multi *m) { return m ? m->err.c_str() : g_comm_create_error.c_str(); }
extern "C" int bk_multi_size(const bk_multi *m) { return m ? (int)m->ctx.size() : 0; }
extern "C" bk_ctx *bk_multi_ctx(bk_multi *m, int i) { return m && i >= 0 && i < (int)m->ctx.size() ? m->ctx[(size_t)i] : nullptr; }
extern "C" bk_comm *bk_multi_comm(bk_multi *m, int i) { return m && i >= 0 && i < (int)m->comm.size() ? m->comm[(size_t)i] : nullptr; }
extern "C" int bk_multi_uses_rccl(const bk_multi *m) { return m && !m->copy_transport && m->ctx.size() > 1 ? 1 : 0; }

#define BK_EACH(m, call)                                                                             \
    do {                                                                                             \
        for (size_t i_ = 0; i_ < (m)->ctx.size(); ++i_) {                                            \
            bk_ctx *c = (m)->ctx[i_];                                                                \
            if (int r_ = (call))  /* (script / zoom verdicts are the same on every device: their text goes out as it is) */ \
                return (m)->fail(r_, (r_ == BK_E_SCRIPT || r_ == BK_E_ZOOM ? std::string() : std::string("device ") + std::to_string(c->device) + ": ") + bk_last_error(c)); \
        }                                                                                            \
    } while (0)

extern "C" int bk_multi_load_globe(bk_multi *m, const char *src, size_t len, const char *chunkname) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_load_globe(c, src, len, chunkname)); return BK_OK; }
extern "C" int bk_multi_load_lens(bk_multi *m, const char *src, size_t len, const char *chunkname) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_load_lens(c, src, len, chunkname)); return BK_OK; }
extern "C" int bk_multi_clear_lens(bk_multi *m) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_clear_lens(c)); return BK_OK; }
extern "C" int bk_multi_clear_globe(bk_multi *m) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_clear_globe(c)); return BK_OK; }
extern "C" int bk_multi_set_frames(bk_multi *m, int nframes) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_set_frames(c, nframes)); return BK_OK; }
extern "C" int bk_multi_set_zoom(bk_multi *m, int zoom_type, int fov) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_set_zoom(c, zoom_type, fov)); return BK_OK; }
extern "C" int bk_multi_set_rubixgrid(bk_multi *m, int n, double cell, double pad) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_set_rubixgrid(c, n, cell, pad)); return BK_OK; }
extern "C" int bk_multi_upload_plate(bk_multi *m, int frame, int plate, const uint8_t *src, int pitch) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_upload_plate_async(c, frame, plate, src, pitch)); return BK_OK; }   /* N DMAs side by side */
extern "C" int bk_multi_fill_plate_lcg(bk_multi *m, int frame, int plate, uint32_t seed_frame) { if (!m) return BK_E_INVALID; BK_EACH(m, bk_fill_plate_lcg(c, frame, plate, seed_frame)); return BK_OK; }
extern "C" int bk_multi_synchronize(bk_multi *m)
{
    if (!m) return BK_E_INVALID;
    BK_EACH(m, bk_synchronize(c));
    for (bk_comm *c : m->comm) if (c) if (int r = bk_comm_synchronize(c)) return m->fail(r, c->err);
    return BK_OK;
}
extern "C" int bk_multi_wait(bk_multi *m, int slot)
{
    if (!m) return BK_E_INVALID;
    for (bk_comm *c : m->comm) if (c) if (int r = bk_comm_wait(c, slot)) return m->fail(r, c->err);
    return BK_OK;
}

// (re)size every stripe context and (re)build the communicators' stripes: rank i owns rows [H*i/N, H*(i+1)/N).
// Failure-atomic like bk_resize: if any device fails, EVERY context is left empty (W = H = 0) and no communicator is kept,
// so that a later bk_multi_resize starts from scratch.
extern "C" int bk_multi_resize(bk_multi *m, int width, int height)
{
    if (!m) return BK_E_INVALID;
    const int n = (int)m->ctx.size();
    if (height < n) return m->fail(BK_E_INVALID, "bk_multi_resize: fewer rows than devices");
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    auto undo = [&](int rc, const std::string &msg) {
        Rccl &R = rccl();
        for (int i = 0; i < n; ++i) {
            bk_comm *c = m->comm[(size_t)i];
            ncclComm_t raw = c ? c->comm : comms[(size_t)i];             // attached to a shell or still loose
            if (c) c->comm = nullptr;
            if (raw && R.ok) { (void)hipSetDevice(m->ctx[(size_t)i]->device); (void)R.CommDestroy(raw); }
            bk_comm_destroy(c);
            m->comm[(size_t)i] = nullptr;
            bk::empty_context(m->ctx[(size_t)i]);
        }
        return m->fail(rc, msg);
    };
    for (bk_ctx *c : m->ctx)
        if (int r = bk_resize(c, width, height)) return undo(r, std::string("device ") + std::to_string(c->device) + ": " + bk_last_error(c));
    bool have_all = true;
    for (bk_comm *c : m->comm) have_all = have_all && c;
    if (!have_all) {
        for (bk_comm *c : m->comm) if (c) return undo(BK_E_STATE, "bk_multi_resize: incomplete communicator set");     // (cannot happen: undo clears all)
        if (n > 1 && !m->copy_transport) {
            Rccl &R = rccl();
            if (!R.ok) return undo(BK_E_STATE, R.error);
            std::vector<int> devs;
            for (bk_ctx *c : m->ctx) devs.push_back(c->device);
            const ncclResult_t r = R.CommInitAll(comms.data(), n, devs.data());
            if (r != ncclSuccess) { std::fill(comms.begin(), comms.end(), nullptr); return undo(BK_E_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(r)); }
        }
        for (int i = 0; i < n; ++i) {
            m->comm[(size_t)i] = comm_shell(m->ctx[(size_t)i], n, i);
            if (!m->comm[(size_t)i]) return undo(BK_E_HIP, g_comm_create_error);
            m->comm[(size_t)i]->comm = comms[(size_t)i];
            comms[(size_t)i] = nullptr;
            m->comm[(size_t)i]->owns_comm = false;              // destroyed by bk_destroy_multi
        }
    } else {
        for (bk_comm *c : m->comm) if (int r = bk_comm_restripe(c)) return undo(r, c->err);
    }
    return BK_OK;
}

// 1 when every stripe context holds a valid lensmap for its current rows (bk_set_rows / bk_resize invalidate it)
extern "C" int bk_multi_lensmap_valid(const bk_multi *m)
{
    if (!m || m->ctx.empty()) return 0;
    for (const bk_ctx *c : m->ctx) if (!c->lensmap_valid) return 0;
    return 1;
}

// stripes of equal work (see bk_comm_rebalance): the row costs of every stripe's current lensmap, new bounds for all, and
// bk_set_rows on every context - call bk_multi_build again afterwards.  bounds_out (nullable) gets the N+1 row bounds.
extern "C" int bk_multi_rebalance(bk_multi *m, int *bounds_out)
{
    if (!m || m->ctx.empty() || !m->comm[0]) return BK_E_INVALID;
    const int n = (int)m->ctx.size(), H = m->ctx[0]->H;
    std::vector<uint32_t> cost((size_t)H, 0), part((size_t)H);
    for (int i = 0; i < n; ++i) {
        bk_ctx *c = m->ctx[(size_t)i];
        uint32_t *d = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void **)&d, (size_t)H * sizeof(uint32_t)) != hipSuccess)
            return m->fail(BK_E_HIP, "bk_multi_rebalance: allocation failed");
        int rc = bk_row_costs_device(c, d);
        hipError_t e = rc == BK_OK ? hipMemcpyAsync(part.data(), d, part.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) : hipSuccess;
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d);
        if (rc != BK_OK) return m->fail(rc, std::string("device ") + std::to_string(c->device) + ": " + bk_last_error(c));
        if (e != hipSuccess) return m->fail(BK_E_HIP, std::string("bk_multi_rebalance: ") + hipGetErrorString(e));
        for (int y = 0; y < H; ++y) cost[(size_t)y] += part[(size_t)y];
    }
    const std::vector<int> b = bounds_from_costs(cost, n);
    for (int i = 0; i < n; ++i) {
        m->comm[(size_t)i]->bounds = b;
        if (int r = bk_set_rows(m->ctx[(size_t)i], b[(size_t)i], b[(size_t)i + 1])) return m->fail(r, bk_last_error(m->ctx[(size_t)i]));
    }
    if (bounds_out) for (int i = 0; i <= n; ++i) bounds_out[i] = b[(size_t)i];
    return BK_OK;
}

// every stripe built concurrently (one host thread per device: emit + hiprtc + launch + fix-up run side by side),
// display[] OR-ed over the stripes (fisheye.c:1976 sets them while building)
extern "C" int bk_multi_build(bk_multi *m, int display_out[BK_MAX_PLATES], double *scale_out)
{
    if (!m) return BK_E_INVALID;
    const size_t n = m->ctx.size();
    // Asynchronous compilation: the generated source is the same for every stripe, so ask ONCE whether its module is there
    // and answer BK_PENDING before any stripe clears or rebuilds its map - otherwise the stripes whose module happened to
    // be cached would show the new lens next to stripes still showing the old one.  Once the code is in the process-wide
    // cache every context loads it from there without waiting.
    std::vector<char> was_async(n, 0);
    bool any_async = false;
    for (size_t i = 0; i < n; ++i) { was_async[i] = m->ctx[i]->async_compile; any_async = any_async || was_async[i]; }
    if (any_async) {
        for (size_t i = 0; i < n; ++i)
            if (was_async[i]) { if (bk::build_module_ready(m->ctx[i]) == BK_PENDING) return BK_PENDING; break; }
        for (size_t i = 0; i < n; ++i) m->ctx[i]->async_compile = false;      // (the code object is cached now: no stall)
    }
    std::vector<int> rc(n, BK_OK);
    std::vector<std::array<int, BK_MAX_PLATES>> disp(n);
    std::vector<double> scale(n, 0.0);
    std::vector<std::thread> pool;
    for (size_t i = 0; i < n; ++i)
        pool.emplace_back([&, i]() { rc[i] = bk_build(m->ctx[i], disp[i].data(), &scale[i]); });
    for (std::thread &t : pool) t.join();
    for (size_t i = 0; i < n; ++i) m->ctx[i]->async_compile = was_async[i] != 0;
    int all[BK_MAX_PLATES] = {0, 0, 0, 0, 0, 0};
    // a malformed callback result somewhere: the reference's scan stopped at the FIRST such pixel of the whole frame - every
    // stripe gives up what that scan had not reached (bk_build did so for the stripe's own first failing pixel only)
    unsigned int bad_key = 0;
    std::string bad_msg;
    for (size_t i = 0; i < n; ++i)
        if (rc[i] == BK_E_SCRIPT && bk_last_build_bad_key(m->ctx[i])) {
            if (bad_msg.empty()) bad_msg = bk_last_error(m->ctx[i]);
            bad_key = std::max(bad_key, bk_last_build_bad_key(m->ctx[i]));
            rc[i] = BK_OK;
        }
    for (size_t i = 0; i < n; ++i) {
        // (a zoom or script verdict is the same on every stripe and its text is the reference's console text: no device in front)
        if (rc[i] == BK_E_ZOOM || rc[i] == BK_E_SCRIPT) return m->fail(rc[i], bk_last_error(m->ctx[i]));
        if (rc[i] != BK_OK) return m->fail(rc[i], std::string("device ") + std::to_string(m->ctx[i]->device) + ": " + bk_last_error(m->ctx[i]));
        if (bad_key) if (int r = bk_truncate_build(m->ctx[i], bad_key, disp[i].data())) return m->fail(r, std::string("device ") + std::to_string(m->ctx[i]->device) + ": " + bk_last_error(m->ctx[i]));
        for (int p = 0; p < BK_MAX_PLATES; ++p) all[p] |= disp[i][(size_t)p];
    }
    for (size_t i = 0; i < n; ++i) for (int p = 0; p < BK_MAX_PLATES; ++p) m->ctx[i]->display[p] = all[p];
    if (display_out) for (int p = 0; p < BK_MAX_PLATES; ++p) display_out[p] = all[p];
    if (scale_out) *scale_out = scale[0];
    if (bad_key) return m->fail(BK_E_SCRIPT, bad_msg);
    return BK_OK;
}

// host destination: every device warps its stripe (all of them enqueued before any is waited for) and copies its rows
// straight into the caller's frame; no inter-GPU exchange is needed for a host frame
extern "C" int bk_multi_set_resident_apply(bk_multi *m, int on)
{
    if (!m) return BK_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        // contexts that share a device each get their own part of its CUs (a one-GPU box with the device named N times)
        int part = 0, parts = 0;
        for (size_t j = 0; j < m->ctx.size(); ++j)
            if (m->ctx[j]->device == m->ctx[i]->device) { if (j < i) ++part; ++parts; }
        if (int r = bk_set_resident_share(m->ctx[i], part, parts, m->ctx[i]->res_reserve)) return m->fail(r, bk_last_error(m->ctx[i]));
        if (int r = bk_set_resident_apply(m->ctx[i], on)) return m->fail(r, bk_last_error(m->ctx[i]));
    }
    return BK_OK;
}

extern "C" int bk_multi_apply(bk_multi *m, int frame, uint8_t *dst, int dst_pitch, int x0, int y0, int rubix_on,
                              const uint8_t pal[BK_MAX_PLATES][256])
{
    if (!m) return BK_E_INVALID;
    BK_EACH(m, bk_apply_begin(c, frame, rubix_on, pal));
    BK_EACH(m, bk_apply_end(c, dst, dst_pitch, x0, y0));
    return BK_OK;
}

// every device warps its stripe of `nframes` frames into its own tight stripe buffer stripes_dev[i] = [nframes][rows_i][W]
extern "C" int bk_multi_apply_stripes(bk_multi *m, int frame0, int nframes, void *const *stripes_dev, int rubix_on,
                                      const uint8_t pal[BK_MAX_PLATES][256])
{
    if (!m || !stripes_dev) return BK_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        bk_ctx *c = m->ctx[i];
        const int rows = c->rows();
        // bk_apply_device addresses pixel (0,0) of the WHOLE view and writes rows [row0,row1): a stripe buffer starts at row0
        uint8_t *origin = (uint8_t *)stripes_dev[i] - (size_t)c->row0 * c->W;
        if (int r = bk_apply_device(c, frame0, nframes, origin, c->W, (size_t)rows * c->W, 0, 0, rubix_on, pal))
            return m->fail(r, std::string("device ") + std::to_string(c->device) + ": " + bk_last_error(c));
    }
    return BK_OK;
}

// the exchange for all ranks at once: RCCL inside one group, or the same schedule over device-to-device copies
static int run_all(bk_multi *m, std::vector<std::vector<BkOp>> &ops, int slot)
{
    const size_t n = m->ctx.size();
#define BK_MHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return m->fail(BK_E_HIP, std::string(#expr " failed: ") + hipGetErrorString(e_)); } while (0)
    for (size_t i = 0; i < n; ++i) {                         // every stripe is warped before anything travels
        BK_MHIP(hipSetDevice(m->ctx[i]->device));
        BK_MHIP(hipEventRecord(m->comm[i]->ev_ready, m->ctx[i]->stream));
    }
    if (!m->copy_transport && n > 1) {
        Rccl &R = rccl();
        for (size_t i = 0; i < n; ++i) { BK_MHIP(hipSetDevice(m->ctx[i]->device)); BK_MHIP(hipStreamWaitEvent(m->comm[i]->xstream, m->comm[i]->ev_ready, 0)); }
        if (R.GroupStart() != ncclSuccess) return m->fail(BK_E_HIP, "ncclGroupStart failed");
        int rc = BK_OK;
        for (size_t i = 0; i < n && rc == BK_OK; ++i) {
            if (hipSetDevice(m->ctx[i]->device) != hipSuccess) rc = BK_E_HIP;
            else rc = post_rccl(m->comm[i], ops[i]);
            if (rc != BK_OK) m->err = m->comm[i]->err;
        }
        if (R.GroupEnd() != ncclSuccess && rc == BK_OK) rc = m->fail(BK_E_HIP, "ncclGroupEnd failed");
        for (size_t i = 0; i < n && rc == BK_OK; ++i) { BK_MHIP(hipSetDevice(m->ctx[i]->device)); BK_MHIP(hipEventRecord(m->comm[i]->ev_done[slot], m->comm[i]->xstream)); }
        return rc;
    }
    // copy transport: the k-th send of rank a to rank b pairs with the k-th receive of b from a (RCCL's own matching rule)
    std::vector<std::vector<size_t>> next_recv(n, std::vector<size_t>(n, 0));
    for (size_t a = 0; a < n; ++a) {
        BK_MHIP(hipSetDevice(m->ctx[a]->device));
        hipStream_t sa = m->comm[a]->xstream;
        for (size_t b = 0; b < n; ++b) BK_MHIP(hipStreamWaitEvent(sa, m->comm[b]->ev_ready, 0));      // b's stripe is warped, its frame buffer free
        for (const BkOp &o : ops[a]) {
            if (!o.bytes) continue;
            if (o.kind == BkOp::LOCAL) { BK_MHIP(hipMemcpyAsync(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice, sa)); continue; }
            if (o.kind != BkOp::SEND) continue;
            const size_t b = (size_t)o.peer;
            size_t &k = next_recv[b][a];
            const BkOp *rv = nullptr;
            for (; k < ops[b].size(); ++k)
                if (ops[b][k].kind == BkOp::RECV && ops[b][k].peer == (int)a) { rv = &ops[b][k++]; break; }
            if (!rv || rv->bytes != o.bytes) return m->fail(BK_E_STATE, "exchange schedule mismatch (send without a matching receive)");
            BK_MHIP(hipMemcpyAsync(rv->dst, o.src, o.bytes, hipMemcpyDeviceToDevice, sa));
        }
        BK_MHIP(hipEventRecord(m->comm[a]->ev_copied, sa));
    }
    for (size_t b = 0; b < n; ++b) {                         // rank b's exchange is done when every rank's copies have landed
        BK_MHIP(hipSetDevice(m->ctx[b]->device));
        for (size_t a = 0; a < n; ++a) if (a != b) BK_MHIP(hipStreamWaitEvent(m->comm[b]->xstream, m->comm[a]->ev_copied, 0));
        BK_MHIP(hipEventRecord(m->comm[b]->ev_done[slot], m->comm[b]->xstream));
    }
#undef BK_MHIP
    return BK_OK;
}

extern "C" int bk_multi_gather(bk_multi *m, void *const *stripes_dev, int nframes, int root, void *frames_dev, size_t frame_stride, int slot)
{
    if (!m || !stripes_dev || !m->comm[0]) return BK_E_INVALID;
    const size_t n = m->ctx.size();
    if (root < 0 || root >= (int)n || !frames_dev) return m->fail(BK_E_INVALID, "bk_multi_gather: bad root / destination");
    std::vector<std::vector<BkOp>> ops(n);
    for (size_t i = 0; i < n; ++i) {
        if (int r = check_args(m->comm[i], stripes_dev[i], nframes, slot)) return m->fail(r, m->comm[i]->err);
        ops_gather(m->comm[i], stripes_dev[i], nframes, root, frames_dev, frame_stride, &ops[i]);
    }
    return run_all(m, ops, slot);
}

extern "C" int bk_multi_exchange_rotating(bk_multi *m, void *const *stripes_dev, int nframes, void *const *frames_dev, size_t frame_stride, int slot)
{
    if (!m || !stripes_dev || !frames_dev || !m->comm[0]) return BK_E_INVALID;
    const size_t n = m->ctx.size();
    std::vector<std::vector<BkOp>> ops(n);
    for (size_t i = 0; i < n; ++i) {
        if (int r = check_args(m->comm[i], stripes_dev[i], nframes, slot)) return m->fail(r, m->comm[i]->err);
        ops_rotating(m->comm[i], stripes_dev[i], nframes, frames_dev[i], frame_stride, &ops[i]);
    }
    return run_all(m, ops, slot);
}
