"""l-channel sharding over the GPUs of one node.  The channels are independent
(reference matrices.f90:242-248 touches only Uij(:,:,l) per iteration), so the data path needs no
collective; the only exchange is the final gather of the spectra (torch.distributed all_gather:
RCCL over xGMI with backend "nccl", gloo in the CPU tests)."""
import torch
import torch.distributed as dist

COLLECTIVE_CALLS = 0          # all_gather_into_tensor calls issued by gather_spectra in this process (bench.py reports it)


def channel_range(rank, world, lmax, per_rank=None):
    """Contiguous block of l-channels owned by `rank`: returns (l0, nl).

    per_rank=None: static block partition of 0..lmax (strong scaling of one problem);
    per_rank=c   : rank r owns l = r*c .. r*c + c - 1 (weak scaling, what bench.py uses)."""
    if per_rank is not None:
        return rank * per_rank, per_rank
    total = lmax + 1
    base, rem = divmod(total, world)
    nl = base + (1 if rank < rem else 0)
    l0 = rank * base + min(rank, rem)
    return l0, nl


def gather_spectra(E_local, nfun, counts, group=None):
    """All-gather per-rank spectra (nl_r x nfun, same dtype/device) into one (sum nl_r) x nfun tensor.
    `counts[r]` = number of channels of rank r.  Equal counts use one all_gather_into_tensor; ragged
    counts pad to the maximum (gather volume is (lmax+1)*nfun doubles: latency-bound either way)."""
    global COLLECTIVE_CALLS
    if not dist.is_initialized():          # a single process without a launcher: nothing to exchange with
        return E_local.reshape(-1, nfun)
    # with a process group the collective runs at every world size, 1 included: `torch.distributed.run --nproc-per-node 1`
    # is how the RCCL branch gets exercised on the one-GPU test box
    world = dist.get_world_size(group)
    nmax = max(counts)
    pad = torch.zeros(nmax * nfun, dtype=E_local.dtype, device=E_local.device)
    pad[: E_local.numel()] = E_local.reshape(-1)
    out = torch.empty(world * nmax * nfun, dtype=E_local.dtype, device=E_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    COLLECTIVE_CALLS += 1
    out = out.reshape(world, nmax, nfun)
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)
