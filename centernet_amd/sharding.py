"""Multi-GPU layout of the path: images shard over ranks, weights replicate.

One process per GPU (``torch.distributed``, backend "nccl" == RCCL over xGMI).  The only
collective is ONE broadcast of the flattened parameter/buffer block at start-up
(the role ``DataParallel.replicate`` plays per forward in the reference's training code,
src/lib/models/data_parallel.py:70-75); the inference loop has no collectives.  With
7 point-to-point xGMI links per GPU the root feeds all peers concurrently, so a single
flat buffer (58 MB for resdcn_18) is the right granularity -- no bucketing, no ring.
"""
import torch


def shard_range(n_items, rank, world):
    """Contiguous static split of ``n_items`` (image indices) -- the same rule as
    ``torch.chunk``: the first ``n_items % world`` ranks get one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flatten_state(model):
    """All floating-point parameters and buffers as one flat fp32 tensor + the layout."""
    sd = model.state_dict()
    keys = [k for k in sd if sd[k].is_floating_point()]
    flat = torch.cat([sd[k].detach().reshape(-1).float() for k in keys]) if keys else torch.empty(0)
    return flat, [(k, tuple(sd[k].shape)) for k in keys]


def unflatten_state_(model, flat, layout):
    sd = model.state_dict()
    o = 0
    with torch.no_grad():
        for k, shape in layout:
            n = 1
            for s in shape:
                n *= s
            sd[k].copy_(flat[o:o + n].reshape(shape))
            o += n
    if hasattr(model, "invalidate_plans"):
        model.invalidate_plans()


def broadcast_weights(model, src=0, group=None):
    """ONE collective: rank ``src``'s weights to every rank (flat buffer)."""
    import torch.distributed as dist
    flat, layout = flatten_state(model)
    dev = next(model.parameters()).device
    flat = flat.to(dev).contiguous()
    dist.broadcast(flat, src=src, group=group)
    unflatten_state_(model, flat, layout)
    return flat.numel() * 4


def gather_detections(dets, group=None):
    """Optional: concatenate per-rank (B_local,K,D) detections in rank order on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    outs = [torch.empty_like(dets) for _ in range(world)]
    dist.all_gather(outs, dets.contiguous(), group=group)
    return torch.cat(outs, 0)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(script, argv, n_ranks, port=None, python=None):
    """The torchrun command line that starts ``n_ranks`` processes of ``script`` on this node
    (one per GPU; rendezvous on 127.0.0.1 -- the container hostname may not resolve)."""
    import sys
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(int(n_ranks)), "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script] + list(argv)


def launch_ranks(script, argv, n_ranks, env=None):
    """Start the ranks and wait; returns the launcher's exit status.  stdout / stderr are
    inherited, so rank 0's JSON line is this process's output.  The role of the reference's
    ``DataParallel`` device loop (src/lib/models/data_parallel.py:119-128) -- one replica per
    GPU -- played by processes instead of threads."""
    import os
    import subprocess
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC, required by RCCL here
    e.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(launch_command(script, argv, n_ranks), env=e)
