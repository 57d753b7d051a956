"""Row-stripe sharding of the warp across ranks (one process per GPU, torch.distributed).

Every output pixel is independent: rank r owns rows [H*r//N, H*(r+1)//N), builds and
keeps only that stripe of the lensmap (no exchange, ever), holds a full replica of the globe, warps
its stripe, and the frame is reassembled by ONE grouped exchange per batch (RCCL over xGMI on GPUs;
gloo in the CPU tests):

* `gather_stripes`     - every frame onto one root (what a single display needs).  The root's xGMI
                         ingest (7 links) bounds it: 7/8 of every frame's bytes enter one GPU.
* `exchange_rotating`  - batches: frame f is reassembled on rank f % N (a gather whose root rotates),
                         one grouped send/recv per batch, so all N*(N-1) links carry stripes at once and
                         every GPU ends up holding whole frames (1/N of the batch each).

The only other exchange is a 6-int OR of the display[] flags after a build, so every rank knows
which plates the whole frame reads.

On GPUs the exchange itself lives behind the C ABI (include/blinky_hip.h, bk_comm_*: librccl called directly -
ncclCommInitRank, grouped ncclSend / ncclRecv, ncclAllReduce); `rccl_comm` below only ships the communicator's
unique id to the ranks through the process group the launcher set up.  The torch.distributed functions in this module
are the same schedule for host tensors (gloo), which is what the CPU tests and the one-GPU developer smoke use."""
import torch
import torch.distributed as dist


def rccl_comm(ctx, world, rank, device):
    """bk_comm for this rank: rank 0 draws the RCCL unique id, the process group broadcasts it, every rank joins
    (ncclCommInitRank inside libblinkyhip).  Also restricts `ctx` to this rank's stripe."""
    from . import ffi
    t = torch.zeros(129, dtype=torch.uint8, device=device)          # [0] = rank 0 has an id, [1:] = the id
    if rank == 0:
        try:
            uid = ffi.comm_unique_id()
            t[1:].copy_(torch.frombuffer(bytearray(uid), dtype=torch.uint8))
            t[0] = 1
        except ffi.BlinkyError as e:                                 # every rank must learn of it, or the others wait forever
            print(f"[blinky_amd.multigpu] rank 0: {e}", flush=True)
    if world > 1:
        dist.broadcast(t, src=0)
    h = t.cpu().numpy()
    if h[0] != 1:
        raise ffi.BlinkyError("rank 0 could not draw an RCCL unique id (bk_comm_unique_id)")
    return ffi.Comm(ctx, world, rank, bytes(h[1:].tobytes()))


def stripe_bounds(height, world):
    """[b0, b1, ..., bN]: rank r owns rows [b[r], b[r+1]) - as even as integer division allows."""
    return [height * r // world for r in range(world + 1)]


def gather_stripes(stripe, bounds, rank, world, dst=0, out_list=None):
    """stripe: uint8 tensor [..., rows_r, W] of this rank.  Returns the list of all stripes on `dst`
    (None elsewhere).  Collectives want equal shapes, so when H % N != 0 the shorter stripes travel
    padded to the tallest one (at most one extra row) and are trimmed on arrival."""
    if world == 1:
        return [stripe]
    heights = [bounds[r + 1] - bounds[r] for r in range(world)]
    hmax = max(heights)
    send = stripe
    if stripe.shape[-2] != hmax:
        shape = list(stripe.shape)
        shape[-2] = hmax
        send = torch.zeros(shape, dtype=stripe.dtype, device=stripe.device)
        send[..., : stripe.shape[-2], :] = stripe
    if rank == dst:
        if out_list is None or any(t.shape[-2] != hmax for t in out_list):
            shape = list(stripe.shape)
            shape[-2] = hmax
            out_list = [torch.empty(shape, dtype=stripe.dtype, device=stripe.device) for _ in range(world)]
        dist.gather(send.contiguous(), out_list, dst=dst)
        return [t[..., : heights[r], :] for r, t in enumerate(out_list)]
    dist.gather(send.contiguous(), None, dst=dst)
    return None


def owned_frames(nframes, rank, world):
    """frames of a batch that `exchange_rotating` assembles on `rank`: f = rank, rank + N, ..."""
    return list(range(rank, nframes, world))


def exchange_rotating(stripe, bounds, rank, world, out, wait=True):
    """stripe: uint8 [F, rows_r, W] - this rank's rows of F frames.  out: uint8 [ceil(F/N), H, W].
    Afterwards out[k] is the complete frame f = k*N + rank.  One grouped exchange (ncclGroup of
    sends/recvs under RCCL): rank r sends its stripe of frame f to rank f % N and receives, for each
    frame it owns, the other ranks' stripes straight into the rows they belong to (a row slice of a
    frame is contiguous, so no staging copy and uneven stripes need no padding).
    Returns the pending work handles when wait=False (the caller overlaps the next batch's warp)."""
    F = stripe.shape[0]
    r0, r1 = bounds[rank], bounds[rank + 1]
    for f in owned_frames(F, rank, world):
        out[f // world, r0:r1, :].copy_(stripe[f])                      # my own rows of my frames
    if world == 1:
        return []
    ops = []
    for f in range(F):                                                   # same order on every rank (gloo matches in order)
        owner = f % world
        if owner == rank:
            for src in range(world):
                if src != rank:
                    ops.append(dist.P2POp(dist.irecv, out[f // world, bounds[src]:bounds[src + 1], :], src))
        else:
            ops.append(dist.P2POp(dist.isend, stripe[f], owner))
    works = dist.batch_isend_irecv(ops) if ops else []
    if wait:
        for w in works:
            w.wait()
        return []
    return works


def assemble(stripes):
    """concatenate gathered stripes along the row axis -> the full frame(s)"""
    return torch.cat(stripes, dim=-2)


def or_display(display, world, device=None):
    """OR of the per-stripe display[] flags (fisheye.c:1976 sets them while building)."""
    if world == 1:
        return list(display)
    t = torch.tensor(list(display), dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.tolist()]
