"""Built-in demo controller of the reference (formation_gym/__init__.py:19-47
`ezpolicy`, :49-99 `get_action_BFS`), batched.

`get_action_BFS(ezpolicy, obs, per)` on a device tensor [B, N, 6N] - what a
rollout loop calls every step (test.py:23) - is ONE HIP launch
(`fg_policy_bfs`, csrc/fg_policy_kernels.hpp): the whole hierarchy of an env is
evaluated by one group of lanes from observation row 0.  There is no CPU
fallback for it: a missing library raises.

The reference's signature also admits an arbitrary `policy` callable and its
own calling convention (a list of N NumPy vectors, one env).  Those are host
orchestration around the caller's function: the hierarchy is then evaluated
level by level with tensor ops over all groups of a level at once, on whatever
device the observations live on, and NumPy goes out where NumPy came in.
"""
import math

import numpy as np
import torch

from . import _native


def _ez_batch(vel_unused, others, ideal, ivel):
    """ezpolicy on a batch.  others [M, n-1, 2] (positions of the other agents
    relative to me), ideal [M, n, 2], ivel [M, 2]  ->  action [M, 2]."""
    M, n = ideal.shape[0], ideal.shape[1]
    ideal = ideal - ideal.mean(1, keepdim=True)                      # :28
    cur = torch.cat((others, torch.zeros_like(others[:, :1])), 1)    # :31  me = last row
    cur = cur - cur.mean(1, keepdim=True)                            # :33
    me = cur[:, -1]
    d_me = (me[:, None, :] - ideal).norm(dim=-1)                     # [M, n]   :35
    order = torch.argsort(d_me, dim=1, stable=True)
    d_all = (cur[:, :, None, :] - ideal[:, None, :, :]).norm(dim=-1)  # [M, agent, mark]
    closest = d_all.argmin(dim=1)                                    # [M, mark]  :37
    mine = closest == (n - 1)
    mine_sorted = torch.gather(mine, 1, order)
    mine_sorted[:, -1] = True                                        # :38 fallback: last mark
    first = mine_sorted.to(torch.int8).argmax(dim=1)
    pick = torch.gather(order, 1, first[:, None])[:, 0]
    target = ideal[torch.arange(M, device=ideal.device), pick]
    act = torch.clamp(0.5 * (target - me), -1, 1)                    # :39
    done = (ideal - cur).flatten(1).norm(dim=1) < 0.01               # :42
    return act + torch.where(done[:, None], ivel, 0.3 * ivel)        # :43-46


def ezpolicy(obs):
    """Hand-written formation controller on one observation vector (len 6n) or a
    batch [..., 6n]."""
    as_numpy = not torch.is_tensor(obs)
    o = torch.as_tensor(np.asarray(obs, dtype=np.float64)) if as_numpy else obs
    flat = o.reshape(-1, o.shape[-1])
    n = flat.shape[-1] / 6
    assert float(n).is_integer(), n
    n = int(n)
    act = _ez_batch(flat[:, 0:2], flat[:, 2:2 * n].reshape(-1, n - 1, 2),
                    flat[:, 4 * n - 2:6 * n - 2].reshape(-1, n, 2), flat[:, -2:])
    act = act.reshape(o.shape[:-1] + (2,))
    return act.numpy() if as_numpy else act


def get_action_BFS(policy, obs, num_agents_per_layer, out=None):
    """Hierarchical expansion of `policy` over a `num_agents_per_layer`-ary tree.

    obs: list of N observation vectors (reference style, returns a list of N
    arrays of shape (2,)) or a tensor [B, N, 6N] (returns a tensor [B, N, 2]).
    `policy` must be batched when a tensor is given (`ezpolicy` is).
    out (batched extension): a float32 tensor [B, N, 2] to receive the actions."""
    per = int(num_agents_per_layer)
    ref_style = not torch.is_tensor(obs)
    if (not ref_style and policy is ezpolicy and obs.is_cuda and obs.dtype == torch.float32 and obs.dim() == 3
            and 2 <= per <= 8 and obs.shape[1] >= 3):
        return bfs_actions(obs, per, out=out)      # anything else (float64, wider hierarchies): the tensor path below
    o = torch.as_tensor(np.asarray(obs, dtype=np.float64))[None] if ref_style else obs
    B, N, D = o.shape
    layers = math.log(N) / math.log(per)
    assert abs(layers - round(layers)) < 1e-9, 'Observation shape error!'
    L = int(round(layers))
    # everything the hierarchy needs is in agent 0's row plus every agent's velocity
    vel = o[:, :, 0:2]                                               # [B, N, 2]
    rel0 = o[:, 0, 2:2 * N].reshape(B, N - 1, 2)
    pos = torch.cat((torch.zeros_like(rel0[:, :1]), rel0), 1)        # positions relative to agent 0
    ideal = o[:, 0, 4 * N - 2:6 * N - 2].reshape(B, N, 2)
    tgt_vel = o[:, :, -2:].clone()                                   # per-agent target velocity
    for lev in range(L, 0, -1):
        n_cur = per ** lev                                           # members per group
        n_sub = n_cur // per
        groups = N // n_cur
        # centroids of every subgroup: [B, groups, per, 2]
        cen = pos.reshape(B, groups, per, n_sub, 2).mean(3)
        tgt = ideal.reshape(B, groups, per, n_sub, 2).mean(3)
        lead_vel = vel.reshape(B, groups, per, n_sub, 2)[:, :, :, 0]     # leader = first of subgroup
        lead_tv = tgt_vel.reshape(B, groups, per, n_sub, 2)[:, :, :, 0]
        # for subgroup i: other subgroup centroids relative to its own, index order, i removed
        relc = cen[:, :, None, :, :] - cen[:, :, :, None, :]             # [B, g, i, k, 2]
        keep = ~torch.eye(per, dtype=torch.bool, device=o.device)
        others = relc[:, :, keep].reshape(B, groups, per, per - 1, 2)
        M = B * groups * per
        inp = torch.cat((lead_vel.reshape(M, 2), others.reshape(M, 2 * (per - 1)),
                         torch.zeros((M, 2 * (per - 1)), dtype=o.dtype, device=o.device),
                         tgt[:, :, None, :, :].expand(B, groups, per, per, 2).reshape(M, 2 * per),
                         lead_tv.reshape(M, 2)), 1)
        sub_vel = policy(inp) * float(lev)                               # :78-79
        sub_vel = torch.as_tensor(sub_vel).reshape(B, groups, per, 1, 2)
        tgt_vel = sub_vel.expand(B, groups, per, n_sub, 2).reshape(B, N, 2)
    act = tgt_vel
    if out is not None:
        out.copy_(act)
        act = out
    if ref_style:
        a = act[0].numpy()
        return [a[i] for i in range(N)]
    return act


def bfs_actions(obs, num_agents_per_layer, out=None):
    """`get_action_BFS(ezpolicy, obs, per)` for a device tensor obs [B, N, 6N] (float32; the env's own observation
    buffer, or any view of it whose envs are a constant stride apart): one `fg_policy_bfs` launch on the
    tensor's device and torch's current stream there.  Returns raw actions [B, N, 2] (`out` if given)."""
    if not (torch.is_tensor(obs) and obs.is_cuda and obs.dim() == 3 and obs.dtype == torch.float32):
        raise ValueError("bfs_actions needs a float32 CUDA tensor [B, N, 6N]")
    B, N, D = obs.shape
    if D != 6 * N:
        raise ValueError("observation rows of %d floats do not belong to %d agents" % (D, N))
    if B > 0 and (obs.stride(2) != 1 or obs.stride(1) != D):
        obs = obs.contiguous()
    stride = obs.stride(0) if B > 1 else D * N
    if out is None:
        out = torch.empty((B, N, 2), dtype=torch.float32, device=obs.device)
    elif tuple(out.shape) != (B, N, 2) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != obs.device:
        raise ValueError("out must be a contiguous float32 tensor [B, N, 2] on the observations' device")
    rc = _LIB()(B, N, int(num_agents_per_layer), obs.data_ptr(), int(stride), out.data_ptr(),
                _native.current_stream_fast(obs.device))
    if rc:
        _native.check(rc)
    return out


def _LIB(_cache=[]):
    """The bound `fg_policy_bfs` entry point (per-step path: looked up once)."""
    if not _cache:
        _cache.append(_native.load().fg_policy_bfs)
    return _cache[0]
