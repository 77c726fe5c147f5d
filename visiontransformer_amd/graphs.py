"""Forward + loss + backward of a drop-in model as one HIP graph.

At the 32x32-pixel geometries of the reference's configs (trainer.py with config/cifar10, config/cifar100) a
training step is some 350 launches of 5-20 us each, and the gaps between launches are a fifth of the step.
`GraphedForwardBackward` captures model(x[, unk_mask]) -> loss -> backward once (every kernel of this package is
launched on torch's current stream, so torch's stream capture records them) and replays it per step; the
optimizer runs outside the graph on the gradients the replay leaves in `param.grad`.

Shapes are fixed at capture.  Inputs are copied into static tensors before each replay; `unk_mask`, when the
model takes one, is a static device tensor refreshed the same way.  Code that decides anything on the host from
device values cannot be captured; the dropouts inside the blocks draw their seeds on the host, so a model with
block dropout > 0 would replay the same masks every step and is rejected here.

The graph reads the weights from the compute copies (bf16 and transposed bf16) that existed at capture.  Before every
replay those very buffers are brought up to date in place when a parameter has changed since (any optimizer that
modifies parameters through torch ops, load_state_dict, a cleared cache); the fused Adam kernel refreshes them itself.
"""
import torch

from . import ops


class GraphedForwardBackward:
    def __init__(self, model, loss_fn, example_inputs, example_targets, example_unk_mask=None, warmup=3):
        for name, m in model.named_modules():
            # the embedding dropout (`dropout` / `pos_drop`, torch's own graph-safe generator) is fine; the dropouts
            # inside the blocks draw their seeds on the host
            if (isinstance(m, torch.nn.Dropout) and m.p > 0 and model.training
                    and name not in ("dropout", "pos_drop", "head_drop")):
                raise ValueError("a model with dropout > 0 inside its blocks cannot be replayed from a graph "
                                 "(host-drawn seeds)")
        self.model, self.loss_fn = model, loss_fn
        self.x = example_inputs.clone()
        self.y = example_targets.clone()
        self.unk = example_unk_mask.clone() if example_unk_mask is not None else None
        self.params = [p for p in model.parameters() if p.requires_grad]
        side = torch.cuda.Stream(device=self.x.device)
        side.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(side):                    # warm-up off the default stream: lazy one-time set-up
            for _ in range(warmup):                      # (LDS attributes, workspaces, compute copies) happens here
                for p in self.params:
                    p.grad = None
                self._run().backward()
        torch.cuda.current_stream(self.x.device).wait_stream(side)
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._run()
            self.loss.backward()
        self.grads = [p.grad for p in self.params]       # static: every replay rewrites these tensors
        self._pinned = ops.pinned_compute_copies(list(model.parameters()))

    def _run(self):
        out = self.model(self.x, unk_mask=self.unk) if self.unk is not None else self.model(self.x)
        return self.loss_fn(out, self.y)

    def __call__(self, inputs, targets, unk_mask=None):
        """Runs forward + backward on the given batch; returns the (static) loss tensor.  Gradients are in
        `param.grad` afterwards (overwritten, not accumulated)."""
        self.x.copy_(inputs, non_blocking=True)
        self.y.copy_(targets, non_blocking=True)
        if self.unk is not None:
            self.unk.copy_(unk_mask, non_blocking=True)
        for p, g in zip(self.params, self.grads):
            p.grad = g
        ops.refresh_pinned_copies(self._pinned)
        self.graph.replay()
        return self.loss


class GraphedInference:
    """A forward pass without autograd graph (`model.eval()`, `torch.no_grad()`) as one HIP graph: for serving at small
    batches, where a forward pass of 12 blocks is about a hundred launches of a few microseconds each and the host
    cannot issue them as fast as the GPU retires them.  Shapes are fixed at capture; inputs are copied into a static
    tensor before each replay, and the returned logits are a static tensor that the next call overwrites (clone it to
    keep it).

    The graph reads the weights from the operands that existed at capture (bf16 compute copies and the weights with
    LayerNorm folded in).  This object keeps those tensors alive, so a replay never reads recycled memory, and it
    records what the parameters looked like at capture (version counter and address of each, and the count of
    raw-pointer updates such as FusedAdam.step, which moves no version counter): a call after any of them changed
    captures again first (`refresh()`; `auto_refresh=False` raises instead)."""

    def __init__(self, model, example_inputs, example_unk_mask=None, warmup=2, auto_refresh=True):
        self.model = model
        self.x = example_inputs.clone()
        self.unk = example_unk_mask.clone() if example_unk_mask is not None else None
        self.warmup = warmup
        self.auto_refresh = auto_refresh
        self.refresh()

    def _stamp(self):
        return (ops.param_epoch(),
                tuple((p._version, p.data_ptr()) for p in self.model.parameters()),
                tuple((b._version, b.data_ptr()) for b in self.model.buffers()))

    def _run(self):
        return self.model(self.x, unk_mask=self.unk) if self.unk is not None else self.model(self.x)

    def refresh(self):
        was_training = self.model.training
        self.model.eval()
        dev = self.x.device
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                # lazy one-time set-up happens outside the capture
                for _ in range(self.warmup):
                    self._run()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._run()
        # the operands the captured launches read: every cached compute copy and folded weight of this model's
        # parameters (made during the warm-up, outside the graph's memory pool); holding them keeps the caching
        # allocator from handing their memory to someone else when the caches drop them
        mine = {id(p) for p in self.model.parameters()}
        self._held = [ent[2] for key, ent in list(ops._cast_cache.items()) if key[0] in mine]
        self._held += [ent[2] for key, ent in list(ops._fold_cache.items()) if any(k in mine for k in key)]
        self._captured = self._stamp()
        self.model.train(was_training)

    def stale(self):
        """True when a parameter or buffer changed since the capture."""
        return self._stamp() != self._captured

    def __call__(self, inputs, unk_mask=None):
        if self.stale():
            if not self.auto_refresh:
                raise RuntimeError("GraphedInference: the model's parameters changed since the graph was captured; "
                                   "call refresh()")
            self.refresh()
        self.x.copy_(inputs, non_blocking=True)
        if self.unk is not None:
            self.unk.copy_(unk_mask, non_blocking=True)
        self.graph.replay()
        return self.out
