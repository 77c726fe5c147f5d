"""Batched DP-SGD gradient for the drop-in ViT modules.

defence_method/DPSGD.py:111-136 runs the batch one image at a time to obtain per-sample gradients, clips each
to `l2_norm_clip`, adds Gaussian noise of standard deviation `noise_multiplier * l2_norm_clip` and divides by
the batch size.  (As written that loop zeroes the gradient inside the per-sample loop and clips nothing; the
algorithm it intends is the one implemented here and by `literal_step`.)

`clipped_noised_gradients` produces the same gradient from ONE forward over the whole batch and two passes over
its graph:
  1. the backward of sum_i loss_i with every parameter-gradient product replaced by its per-image squared norm
     (`ops.per_sample_norms`: the weight-gradient MFMA kernel with its token slabs cut at image boundaries, each
     block reducing its tile to a sum of squares) -> ||g_i|| for every image, no per-sample gradient in memory;
  2. sum_i c_i g_i with c_i = min(1, C / (||g_i|| + 1e-6)).  The upstream gradient of image i under the loss
     sum_i c_i loss_i is c_i times its value in pass 1 (nothing in the network couples two images), so the
     input-gradient chain is NOT run again: pass 1 keeps each node's upstream gradients and per-image vector sums
     (`keep=True`, 0.7 GB per ViT-B/16 block at batch 256), and `ops.clipped_replay` walks the graph once more with
     every node forming only its parameter gradients -- the weight-gradient products with the rows of one operand
     scaled by c, biases and LayerNorm parameters as c-weighted sums of the per-image sums.  `reuse_norm_pass=False`
     runs the ordinary backward of sum_i c_i loss_i instead (the same gradient up to rounding).
"""
import torch

from . import ops


def per_sample_losses(logits, labels):
    """loss_i of nn.CrossEntropyLoss on a batch of one (DPSGD.py:119-120)."""
    return torch.nn.functional.cross_entropy(logits, labels, reduction="none")


def per_sample_grad_norms(model, losses, keep=False):
    """||g_i||_2 over all parameters of `model` for losses [B] computed from one batched forward.  Keeps the graph;
    keep=True also keeps, inside the graph's nodes, what `ops.clipped_replay` needs."""
    params = [p for p in model.parameters() if p.requires_grad]
    held = [p.grad for p in params]
    for p in params:
        p.grad = None
    with ops.per_sample_norms(losses.shape[0], losses.device, keep=keep) as ps:
        losses.sum().backward(retain_graph=True)
    leaked = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
    for p, g in zip(params, held):
        p.grad = g
    if leaked:
        raise RuntimeError("per-sample norms: parameters reached outside the HIP autograd nodes: " + ", ".join(leaked))
    return ps.sq.sqrt()


def clipped_noised_gradients(model, inputs, labels, l2_norm_clip=1.0, noise_multiplier=1.0, generator=None,
                             loss_fn=per_sample_losses, reducer=None, reuse_norm_pass=True, **forward_kwargs):
    """Leaves (sum_i clip(g_i) + N(0, (noise_multiplier * l2_norm_clip)^2)) / B in every parameter's .grad
    (DPSGD.py:122-127) and returns (mean loss, per-sample norms).

    Data parallel: pass the rank's `parallel.GradReducer(model, average=False)` as `reducer`.  The clipped sums of
    all ranks are then added (the bucketed all-reduce runs under the second backward pass), ONE noise draw is added
    to the global sum and the result is divided by the global batch; every rank must hand in a generator in the same
    state so that the replicas add the same noise and stay identical."""
    B = inputs.shape[0]
    logits = model(inputs, **forward_kwargs)
    losses = loss_fn(logits, labels)
    if losses.shape != (B,):
        raise ValueError("loss_fn must return one loss per sample")
    norms = per_sample_grad_norms(model, losses, keep=reuse_norm_pass)
    coef = torch.clamp(l2_norm_clip / (norms + 1e-6), max=1.0)
    for p in model.parameters():
        p.grad = None
    if reuse_norm_pass:
        with ops.clipped_replay(coef):
            losses.sum().backward()
    else:
        (losses * coef).sum().backward()
    total = B
    if reducer is not None:
        if reducer.average:
            raise ValueError("the reducer must sum (average=False): the division by the global batch happens here")
        reducer.finish()
        total = B * reducer.world
    std = noise_multiplier * l2_norm_clip
    with_grad = [p for p in model.parameters() if p.grad is not None]
    grads = [p.grad for p in with_grad]
    if std > 0 and grads:
        # ONE draw for all parameters (a flat buffer cut into views, in parameter order) and two multi-tensor launches,
        # instead of three small launches per parameter tensor
        by_kind = {}
        for g in grads:
            by_kind.setdefault((g.dtype, g.device), []).append(g)
        for (dtype, device), gs in by_kind.items():
            flat = torch.empty(sum(g.numel() for g in gs), dtype=dtype, device=device).normal_(0, std, generator=generator)
            views, off = [], 0
            for g in gs:
                views.append(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
            torch._foreach_add_(gs, views)
    if grads:
        torch._foreach_div_(grads, float(total))
    return losses.detach().mean(), norms


def literal_step_gradients(model, inputs, labels, l2_norm_clip=1.0, noise_multiplier=1.0, generator=None,
                           **forward_kwargs):
    """The same gradient by the reference's schedule: B passes of batch one (DPSGD.py:115-123)."""
    params = [p for p in model.parameters() if p.requires_grad]
    B = inputs.shape[0]
    acc = [torch.zeros_like(p) for p in params]
    norms = []
    for i in range(B):
        for p in params:
            p.grad = None
        torch.nn.functional.cross_entropy(model(inputs[i:i + 1], **forward_kwargs), labels[i:i + 1]).backward()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
        norms.append(norm)
        torch._foreach_add_(acc, torch._foreach_mul(grads, torch.clamp(l2_norm_clip / (norm + 1e-6), max=1.0)))
    std = noise_multiplier * l2_norm_clip
    for p, a in zip(params, acc):
        if std > 0:
            a = a + torch.empty_like(a).normal_(0, std, generator=generator)
        p.grad = a / B
    return torch.stack(norms)
