"""Row-stripe sharding of the warp across ranks (one process per GPU, torch.distributed).

Every output pixel is independent: rank r owns rows [H*r//N, H*(r+1)//N), builds and
keeps only that stripe of the lensmap (no exchange, ever), holds a full replica of the globe, warps
its stripe, and the frame is reassembled by ONE collective: a gather of the stripes onto a root
(RCCL over xGMI on GPUs; gloo in the CPU tests).  The only other exchange is a 6-int OR of the
display[] flags after a build, so every rank knows which plates the whole frame reads."""
import torch
import torch.distributed as dist


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
