"""Fused Adam over fp32 master parameters (torch.optim.Adam semantics, the optimizer of
trainer.py:204-205) that refreshes the bf16 compute copies of the weights in the same pass.

One launch updates every flat parameter tensor: a table of (param, grad, exp_avg, exp_avg_sq, bf16 copy, n)
records and a table of (tensor, first element) chunks are uploaded each step (gradient buffers
may move between steps) and `vitmi_adam_step_multi` gives one block to each chunk.  Weight matrices whose transposed
bf16 copy is in use (the operand of the input-gradient products) go through `vitmi_adam_step_matrices`, a second
launch with one block per 64 x 64 tile that writes both copies in the same pass."""
import os

import numpy as np
import torch

from . import ops
from ._lib import check, lib

CHUNK = 65536  # elements per block
# weight matrices with a transposed copy: update and both copies in one pass (VITMI_TILE_ADAM=0: flat + transpose pass)
TILE_ADAM = os.environ.get("VITMI_TILE_ADAM", "1") != "0"


class FusedAdam:
    """params: an iterable of parameters or of {"params": [...], "lr": ..., ...} groups, as torch.optim.Adam takes
    them.  `param_groups` is a list of dicts with the usual keys, so a torch LR scheduler written against
    `optimizer.param_groups[i]["lr"]` drives it (trainer.py:206-211 steps CosineAnnealingLR; construct the scheduler
    on a torch optimizer or set the entries yourself: this class does not derive from torch.optim.Optimizer).
    Every parameter counts its own steps for the bias correction, as torch.optim.Adam does: one that skips a step
    (grad None) is not advanced.  Parameters must be contiguous float32 device tensors: the kernel reads and writes
    n fp32 elements at their address."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        groups = params if params and isinstance(params[0], dict) else [{"params": params}]
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g["params"] = [p for p in g["params"] if p.requires_grad]
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            g.setdefault("weight_decay", weight_decay)
            self.param_groups.append(g)
        self.params = [p for g in self.param_groups for p in g["params"]]
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.step_count = 0          # calls of step()
        self._steps = {}             # parameter -> updates it has taken
        self.state = {}
        self._tables = {}   # device -> (pinned host buffers, device buffer, copy-done events, parity)
        self._layouts = {}  # device -> (ids of the tensors, record constants, chunk table)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            st = (torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
            self.state[p] = st
        return st

    def _upload(self, device, words):
        """Stream-ordered upload of an int64 table through one of two pinned staging buffers."""
        n = len(words)
        ent = self._tables.get(device)
        if ent is None or ent[0][0].numel() < n:
            cap = max(2 * n, 1024)
            host = [torch.empty(cap, dtype=torch.int64).pin_memory() for _ in range(2)]
            dev = torch.empty(cap, dtype=torch.int64, device=device)
            ent = [host, dev, [None, None], 0]
            self._tables[device] = ent
        host, dev, events, parity = ent
        if events[parity] is not None:
            events[parity].synchronize()      # the copy that last read this staging buffer has finished
        host[parity][:n].copy_(torch.from_numpy(words))
        dev[:n].copy_(host[parity][:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        events[parity] = ev
        ent[3] = parity ^ 1
        return dev

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g["lr"] = value

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self.step_count += 1
        # the kernel writes the parameters through raw pointers (no version bump): operands derived from them that
        # it does not refresh itself (LayerNorm folded into a weight, for the inference entry points) are dropped
        ops.drop_folded()
        launches = {}                # (device, group index, step number) -> parameters
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam runs on ROCm device tensors only")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError(f"FusedAdam updates contiguous float32 parameters in place; got {p.dtype}, "
                                    f"contiguous={p.is_contiguous()}, shape {tuple(p.shape)} (keep fp32 master "
                                    "weights: the compute dtype is set with set_compute_dtype, not .bfloat16())")
                n = self._steps.get(p, 0) + 1
                self._steps[p] = n
                launches.setdefault((p.device, gi, n), []).append(p)
        for (device, gi, nstep), plist in launches.items():
            group = self.param_groups[gi]
            # weight matrices with a transposed bf16 copy take the tile form of the update, which writes that copy
            # in the same pass; everything else (and a matrix the tile form's alignment contract does not fit)
            # goes through the flat chunks, with a transpose pass afterwards for any such matrix
            keep, flat, mats, jobs = [], [], [], []
            for p in plist:
                g = p.grad.contiguous()
                if g.dtype != torch.float32:
                    g = g.float()
                keep.append(g)
                hit = ops._cache_lookup((id(p), torch.bfloat16), p)
                p16 = hit.data_ptr() if hit is not None else 0
                hit_t = ops._cache_lookup((id(p), torch.bfloat16, "T"), p)
                if hit_t is not None and p.dim() == 2:
                    m, v = self._state(p)
                    ptrs = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr())
                    if (TILE_ADAM and p.shape[1] % 4 == 0 and all(a % 16 == 0 for a in ptrs) and p16 % 8 == 0
                            and hit_t.data_ptr() % 8 == 0):
                        mats.append((p, g.data_ptr(), p16, hit_t.data_ptr()))
                        continue
                    jobs.append((p.data_ptr(), hit_t.data_ptr(), p.shape[0], p.shape[1]))
                else:
                    ops._cast_cache.pop((id(p), torch.bfloat16, "T"), None)
                flat.append((p, g.data_ptr(), p16))
            recs, chunks = self._layout((device, gi), [p for p, _, _ in flat])
            if flat:
                recs[:, 0] = [p.data_ptr() for p, _, _ in flat]
                recs[:, 1] = [g for _, g, _ in flat]
                recs[:, 4] = [c for _, _, c in flat]
            mrecs, starts, total_tiles = self._layout_tiles((device, gi), [p for p, _, _, _ in mats])
            if mats:
                mrecs[:, 0] = [p.data_ptr() for p, _, _, _ in mats]
                mrecs[:, 1] = [g for _, g, _, _ in mats]
                mrecs[:, 4] = [c for _, _, c, _ in mats]
                mrecs[:, 5] = [t for _, _, _, t in mats]
            words = np.concatenate([recs.reshape(-1), chunks, mrecs.reshape(-1), starts])
            hyper = (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"], nstep,
                     grad_scale)
            with torch.cuda.device(device):
                dev = self._upload(device, words)
                if flat:
                    check(lib.vitmi_adam_step_multi(dev.data_ptr(), len(flat), dev.data_ptr() + 8 * recs.size,
                                                    chunks.size // 2, CHUNK, *hyper, ops._stream()),
                          "vitmi_adam_step_multi")
                if mats:
                    base = dev.data_ptr() + 8 * (recs.size + chunks.size)
                    check(lib.vitmi_adam_step_matrices(base, base + 8 * mrecs.size, len(mats), total_tiles, *hyper,
                                                       ops._stream()), "vitmi_adam_step_matrices")
                if jobs:
                    tiles = [((r + 63) // 64) * ((c + 63) // 64) for _, _, r, c in jobs]
                    tstarts = np.concatenate([[0], np.cumsum(tiles)[:-1]]).astype(np.int64)
                    table = ops.upload(np.concatenate([np.asarray(jobs, dtype=np.int64).reshape(-1), tstarts]), device)
                    check(lib.vitmi_transpose_cast_multi(table.data_ptr(), table.data_ptr() + 32 * len(jobs), len(jobs),
                                                         int(sum(tiles)), ops._stream()), "vitmi_transpose_cast_multi")
            del keep

    def _layout_tiles(self, device, plist):
        """(records [len, 8] of vitmi_adam_matrix with the per-matrix constants filled in, first tile of each matrix,
        total tiles) for this set of weight matrices; cached like _layout."""
        key = ("tiles",) + tuple(id(p) for p in plist)
        ent = self._layouts.get((device, key))
        if ent is None:
            recs = np.zeros((len(plist), 8), dtype=np.int64)
            tiles = []
            for ti, p in enumerate(plist):
                m, v = self._state(p)
                recs[ti, 2], recs[ti, 3], recs[ti, 6], recs[ti, 7] = m.data_ptr(), v.data_ptr(), p.shape[0], p.shape[1]
                tiles.append(((p.shape[0] + 63) // 64) * ((p.shape[1] + 63) // 64))
            starts = np.concatenate([[0], np.cumsum(tiles)[:-1]]).astype(np.int64) if tiles else np.zeros(0, np.int64)
            ent = (recs, starts, int(sum(tiles)))
            self._layouts[(device, key)] = ent
        return ent

    def _layout(self, device, plist):
        """(records [len, 6] with the per-tensor constants filled in, chunk table) for this set of tensors; rebuilt
        only when the set changes (the state tensors never move, gradients and copies are filled in per step)."""
        # one entry per SET of tensors: parameters of a group whose step counts differ (one skipped a step) go out
        # as several launches per step, each with its own cached layout
        key = tuple(id(p) for p in plist)
        device = (device, key)
        ent = self._layouts.get(device)
        if ent is None:
            recs = np.zeros((len(plist), 6), dtype=np.int64)
            chunks = []
            for ti, p in enumerate(plist):
                m, v = self._state(p)
                n = p.numel()
                recs[ti, 2], recs[ti, 3], recs[ti, 5] = m.data_ptr(), v.data_ptr(), n
                firsts = np.arange(0, n, CHUNK, dtype=np.int64)
                chunks.append(np.stack([np.full_like(firsts, ti), firsts], axis=1).reshape(-1))
            ent = (key, recs, np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64))
            self._layouts[device] = ent
        return ent[1], ent[2]
