"""Batch sharding of the rollout over the GPUs of one node.

Every env row is independent in every function on the path (DAM:118-427 are row-wise over the batch), so
the env axis is split contiguously — rank r owns rows [lo, hi) of obs / actions / ref_indexes and its own
handle — and there is NO data-path collective.  The only exchange is one all-gather per rollout of the
8-float episodic summary that eb_episode_summary produces for a shard (RCCL over xGMI on the GPU box:
backend "nccl"; the CPU tests run the same code over gloo).  The message is 32 bytes per rank:
latency-bound, off the critical path.

Host logic only — no arithmetic of the path lives here."""
import torch
import torch.distributed as dist

SUMMARY_LEN = 8   # EB_SUMMARY_LEN: [sum reward, sum punish_train, sum punish_real, #envs punished, sum |dy|, max |dy|, n_env, horizon]


def shard_range(n_env_total, rank, world):
    """Contiguous split of the env axis; the first (n_env_total % world) ranks take one extra row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError('bad rank/world: %r/%r' % (rank, world))
    base, extra = divmod(int(n_env_total), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_summaries(summary8, group=None):
    """All-gather of one shard summary -> [world, 8] on every rank (the only inter-GPU exchange)."""
    if summary8.numel() != SUMMARY_LEN:
        raise ValueError('summary must have %d floats' % SUMMARY_LEN)
    if not (dist.is_available() and dist.is_initialized()):
        return summary8.reshape(1, SUMMARY_LEN).clone()
    world = dist.get_world_size(group)
    out = torch.empty((world, SUMMARY_LEN), dtype=summary8.dtype, device=summary8.device)
    dist.all_gather_into_tensor(out.view(-1), summary8.reshape(-1).contiguous(), group=group)
    return out


class PendingGather:
    """An all-gather in flight (gather_summaries_async).  result() orders the caller's stream after the
    collective (no host wait on RCCL) and hands back the [world, 8] tensor."""
    def __init__(self, work, out, keep):
        self._work, self._out, self._keep = work, out, keep

    def result(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._out


def gather_summaries_async(summary8, group=None, copy=True):
    """gather_summaries without stalling the launch stream: the collective runs on the backend's own stream,
    ordered after the work already queued on the current stream, while the next rollout's kernels go on.
    The caller must not overwrite `summary8` before result() — keep one buffer per gather in flight.  Without a process group
    (one GPU) there is nothing to exchange: result() is a copy of `summary8` as [1, 8] made now (one 32-byte device copy), so that
    it stays what it was when the caller reuses the buffer — as the gathered tensor of an N-GPU job does.  copy=False: result()
    is `summary8` itself viewed as [1, 8] — no copy, no launch — for a caller that keeps the rule above on one GPU too."""
    if summary8.numel() != SUMMARY_LEN:
        raise ValueError('summary must have %d floats' % SUMMARY_LEN)
    if not (dist.is_available() and dist.is_initialized()):
        one = summary8.reshape(1, SUMMARY_LEN)
        return PendingGather(None, one.clone() if copy else one, None)
    world = dist.get_world_size(group)
    if summary8.is_cuda and dist.get_backend(group) != 'nccl':
        # a backend without device collectives (gloo in tests): through host memory, synchronously
        host = torch.empty((world, SUMMARY_LEN), dtype=summary8.dtype)
        dist.all_gather_into_tensor(host.view(-1), summary8.detach().reshape(-1).cpu(), group=group)
        return PendingGather(None, host.to(summary8.device), None)
    out = torch.empty((world, SUMMARY_LEN), dtype=summary8.dtype, device=summary8.device)
    src = summary8.reshape(-1).contiguous()
    work = dist.all_gather_into_tensor(out.view(-1), src, group=group, async_op=True)
    return PendingGather(work, out, src)


def combine_summaries(all8):
    """[world, 8] shard summaries -> the summary of the whole batch (sums add, max takes the max, the horizon
    is common).  Accumulated in float64, as eb_episode_summary does inside a shard."""
    a = all8.to(torch.float64)
    out = torch.empty((SUMMARY_LEN,), dtype=torch.float64, device=all8.device)
    out[0:5] = a[:, 0:5].sum(0)
    out[5] = a[:, 5].max()
    out[6] = a[:, 6].sum()
    out[7] = a[:, 7].max()
    return out.to(all8.dtype)
