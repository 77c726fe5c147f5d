#!/usr/bin/env python3
"""Headline benchmark: train images/sec of ViT-B/16 224 px (ImageNet100-shaped synthetic data)
on N MI355X, bf16 MFMA kernels, fp32 master weights + Adam, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + CE loss + backward (+ bucketed gradient all-reduce when N > 1) + Adam step
on one batch of 256 images per GPU that is already resident in HBM.  Rank 0 prints one JSON
line.  `roofline` describes the dominant kernel family (per-launch HIP-event timing inside the
timed region); `cpu_baseline` is the CPU oracle timed on the host cores on a bounded sample.

`--workload` selects one of the reference's other configurations (same line format, each with
`roofline` and `cpu_baseline`):
    vit_b16_224         ViT-B/16 224 px, batch 256 per GPU, bf16 (the default, the headline)
    vit_b16_224_f32     the same geometry in the reference's own arithmetic: the library's exact fp32 mode
    cifar10_cpu         mymodel/ViT.py on the 32x32 / patch 4 geometry, batch 64, fp32: the reference's own
                        CPU-runnable case; the CPU oracle runs the whole batch-64 step beside the fp32 HIP path
    cifar100_export     the same geometry with 100 classes, bf16, batch 256: the training step, plus the forward
                        pass with the attention-probability export on (`export` object, HBM roofline on the
                        exported bytes)
    cifar_mask[_pbf]    ViT_mask / PEdrop masked training, a new mask every step (mask ratio 0.5)
    dpsgd_vit_b16       DP-SGD with per-sample clipping, batch 256
    tim_vit_small_mask  the timm-style ViT-S/16 with unk_mask that the reference's trainer builds
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak of one MI355X (vendor spec, no sparsity)
F32_PEAK_TFLOPS = 157.3        # fp32 vector / matrix peak (the exact-fp32 kernels are FMA loops)
HBM_PEAK_GBS = 8000.0
PROF_STRIDE = 5                # launch timers sample every 5th launch of the reported family

CIFAR = dict(image_size=32, patch_size=4, embed_dim=192, depth=12, heads=12, mlp_dim=512)
VITB = dict(image_size=224, patch_size=16, embed_dim=768, num_classes=100, depth=12, heads=12, mlp_dim=3072, dim_head=64)
WORKLOADS = {
    # name: (ctor args, kind, per-GPU batch, train GFLOP/img = 6 * fwd MACs)
    "vit_b16_224": (VITB, "plain", 256, 105.378822144),
    # BASELINE config 1: mymodel/ViT.py on config/cifar10 with the 32x32 / patch 4 overrides (UMAP.py:12-17), batch 64,
    # fp32 as trainer.py runs it
    "cifar10_cpu": (dict(CIFAR, num_classes=10), "plain_f32", 64, 1.327),
    # the metric's geometry in the reference's own arithmetic (trainer.py trains in fp32): the exact mode of the library
    "vit_b16_224_f32": (VITB, "plain_f32", 256, 105.378822144),
    # BASELINE config 2: CIFAR-100 geometry, bf16, with the attention-map export mia_attn.py consumes
    "cifar100_export": (dict(CIFAR, num_classes=100), "export", 256, 1.327),
    # BASELINE config 3: ViT_mask / PEdrop, a new BEiT block mask every step (32 of 64 patches)
    "cifar_mask": (dict(CIFAR, num_classes=10), "mask", 256, 1.327),
    # the same step with the "pbf" jigsaw augmentation in front (trainer.py:126): masked patches come from a
    # synthetic public bank [256, 48, 64] (dataloader.py:17-37 shape), images rebuilt by one HIP gather
    "cifar_mask_pbf": (dict(CIFAR, num_classes=10), "pbf", 256, 1.327),
    # BASELINE config 5: DP-SGD (defence_method/DPSGD.py:111-136) with per-sample clipping, as one batched forward, a
    # per-sample-norm backward and a clipped backward; clip 1.0, noise multiplier 1.0 (DPSGD.py:26,31)
    "dpsgd_vit_b16": (VITB, "dpsgd", 256, 105.378822144),
    # the timm-style ViT-S/16 with unk_mask that trainer.py:172-176 builds (dim_head = 384/6 = 64, mlp 1536)
    "tim_vit_small_mask": (dict(image_size=224, patch_size=16, embed_dim=384, num_classes=100, depth=12, heads=6,
                                mlp_dim=1536, dim_head=64), "timm", 256, 27.59),
}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(kw, kind, model, seconds_budget=14.0):
    """The CPU oracle (pure-torch restatement of the reference) on the host cores: a bounded sample of the same
    workload -- one warm-up step, then timed steps -- with all the cores this process may use and with one thread.
    224-px models: batch 8; the 32x32 geometry: the reference's own batch 64; DP-SGD: four images one at a time, each
    with its clip (the schedule of DPSGD.py:115-123)."""
    from oracle import vit_oracle
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    small = kw["image_size"] <= 64
    B = 64 if small else (4 if kind == "dpsgd" else 8)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, kw["image_size"], kw["image_size"], generator=g)
    y = torch.randint(0, kw["num_classes"], (B,), generator=g)
    what = f"oracle fwd+bwd fp32, batch {B}"
    if kind == "timm":
        from oracle import timm_vit_oracle
        params = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        um = torch.zeros(1, x.shape[-1] // kw["patch_size"] * (x.shape[-1] // kw["patch_size"]) + 1, dtype=torch.long)
        um[0, 1::2] = 1

        def one():
            timm_vit_oracle.loss_and_grads(params, x, y, kw["heads"], unk_mask=um)
        what = f"timm-style oracle fwd+bwd fp32 with unk_mask, batch {B}"
    else:
        masked = kind in ("mask", "pbf")
        params = vit_oracle.init_params(masked=masked, seed=1001, **kw)
        um = None
        if masked:
            um = torch.zeros(1, (kw["image_size"] // kw["patch_size"]) ** 2 + 1, dtype=torch.long)
            um[0, 1::2] = 1
        if kind == "dpsgd":
            def one():
                acc = None
                for i in range(B):
                    _, _, gr = vit_oracle.loss_and_grads(params, x[i:i + 1], y[i:i + 1], kw["heads"])
                    norm = torch.sqrt(sum((v.double() ** 2).sum() for v in gr.values()))
                    c = min(1.0, 1.0 / (float(norm) + 1e-6))
                    acc = {k: v * c for k, v in gr.items()} if acc is None else {k: acc[k] + v * c for k, v in gr.items()}
            what = f"oracle DP-SGD schedule: {B} x (batch 1 fwd+bwd fp32, clip, accumulate)"
        elif masked:
            def one():
                vit_oracle.loss_and_grads(params, x, y, kw["heads"], unk_mask=um)
            what += ", unk_mask on every other patch"
        else:
            def one():
                vit_oracle.loss_and_grads(params, x, y, kw["heads"])

    def timed(threads, budget, min_steps, max_steps):
        torch.set_num_threads(threads)
        one()
        t0 = time.perf_counter()
        steps = 0
        while steps < min_steps or (time.perf_counter() - t0 < budget and steps < max_steps):
            one()
            steps += 1
        return B * steps / (time.perf_counter() - t0), steps

    many, steps = timed(cores, seconds_budget, 3, 8)
    one_t, steps1 = timed(1, seconds_budget / 2, 1, 2)
    torch.set_num_threads(cores)
    return {"value": many, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{what}, {steps} timed steps after 1 warm-up",
            "one_thread": {"value": one_t, "unit": "images/s", "cores": 1,
                           "sample": f"the same, {steps1} timed step(s) after 1 warm-up"},
            "cpu_model": cpu_model_name(), "host_cpus": os.cpu_count()}


PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
KERNEL_SOURCES = ["common.h", "gemm_fast.hip", "gemm.hip", "attention.hip", "elementwise.hip", "blocks.hip"]
# timer family -> substrings of the names, in the rocprofv3 counter summary, of the kernels one call of the family
# launches: the first is the main kernel; a GEMM call also launches the small-tile kernel on the rows of a thin last
# round (nt) or the slab reduction (tn)
PMC_KERNELS_OF_FAMILY = {
    "gemm_nt_phased_kernel|gemm_nt_fast_kernel[plain->bf16]": ["gemm_nt_phased_kernelILi0EDF16b",
                                                               "gemm_nt_fast_kernelILi0EDF16bLb0ELi128ELi128"],
    "gemm_tn_batch_kernel|gemm_tn_phased_kernel|gemm_tn_fast_kernel[+reduce]": ["gemm_tn_batch_kernel", "tn_batch_reduce_kernel"],
}


def kernel_source_hash():
    """sha256 over the kernel sources the counters describe: a counter summary collected on other sources is stale."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "visiontransformer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(family):
    """HBM bytes per call of the dominant kernel family, from the committed rocprofv3 PMC summary of this same
    workload (counters cannot be read from inside the process): FETCH_SIZE x 2 + WRITE_SIZE, summed over the
    kernels a call launches and divided by the launches of the main one.  The summary records a hash of the kernel
    sources it was collected on; one collected on other sources is refused (traffic null, with the reason)."""
    keys = PMC_KERNELS_OF_FAMILY.get(family)
    if keys is None:
        return {"traffic": None, "traffic_source": f"no counter pass recorded for {family}"}
    if not os.path.exists(PMC_TRAFFIC_FILE):
        return {"traffic": None, "traffic_source": f"profiles/{os.path.basename(PMC_TRAFFIC_FILE)} is missing"}
    with open(PMC_TRAFFIC_FILE) as f:
        doc = json.load(f)
    have, want = doc.get("kernel_source_hash"), kernel_source_hash()
    if have != want:
        return {"traffic": None,
                "traffic_source": f"profiles/{os.path.basename(PMC_TRAFFIC_FILE)} was collected on kernel sources {have}, "
                                  f"this tree's are {want}: stale, not reported"}
    kernels = doc["kernels"]
    total, calls, parts = 0, 0, []
    for i, key in enumerate(keys):
        hits = [rec for name, rec in kernels.items() if key in name]
        if not hits and i == 0:
            return {"traffic": None, "traffic_source": f"no kernel matching {key!r} in the counter summary"}
        for rec in hits:
            total += rec["launches"] * rec["hbm_bytes_per_launch"]
            parts.append(f"{key}: {rec['launches']} x {rec['hbm_bytes_per_launch']}")
            if i == 0:
                calls += rec["launches"]
    return {"traffic": int(total / calls), "traffic_unit": "bytes/call",
            "traffic_parts": parts,
            "traffic_source": f"profiles/{os.path.basename(PMC_TRAFFIC_FILE)} (rocprofv3 --pmc FETCH_SIZE, "
                              f"WRITE_SIZE in separate passes; kernel sources {want})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="vit_b16_224", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--graph", action="store_true", default=None,
                    help="replay forward + loss + backward from one HIP graph (one GPU).  Default: on for the 32-pixel "
                         "workloads, whose step is ~350 launches of 5-30 us (the graph also lets the library run the "
                         "weight gradients on its second stream), off otherwise.  Launches inside a graph carry no "
                         "events: the roofline object then comes from eager warm-up steps")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--sweep-reserve", action="store_true",
                    help="N > 1: run the timed region three times, with 0, 8 and 16 compute units kept free of the "
                         "persistent kernels for the collective (and RCCL capped at as many channels), one JSON line each")
    args = ap.parse_args()

    if os.environ.get("VITMI_BENCH_WATCHDOG"):      # dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VITMI_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # one rank per GPU; more ranks than GPUs (a rehearsal of the N > 1 path on one card with the gloo
    # backend, VITMI_DIST_BACKEND=gloo) wraps around
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    backend = os.environ.get("VITMI_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    reserve_default = int(os.environ.get("VITMI_RESERVE_CUS", "16") or 0)

    def rccl_options():
        try:
            return dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            return None

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            if local_world > torch.cuda.device_count():
                raise SystemExit(f"bench.py: {local_world} ranks on this node but {torch.cuda.device_count()} GPU(s) visible; "
                                 "RCCL needs one GPU per rank (VITMI_DIST_BACKEND=gloo rehearses N > 1 on one card)")
            # The GEMMs and the attention backward keep every CU full (one workgroup with most of the register file
            # and LDS per CU), and a collective's workgroups cannot share a CU with them: every CU a collective
            # holds delays one block of the next persistent launch until the others have finished.  So the
            # collective gets as many channels (one workgroup each) as the kernels leave compute units free
            # (VITMI_RESERVE_CUS, default 16 = two per XCD; 343 MB of gradients per step need a small fraction of
            # the links' bandwidth), and runs on a high-priority stream.
            if reserve_default > 0:
                os.environ.setdefault("NCCL_MAX_NCHANNELS", str(reserve_default))
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=rccl_options())
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from visiontransformer_amd import ViT, ViT_mask, _lib, ops
    from visiontransformer_amd.optim import FusedAdam
    # A/B switches of the library (same box, one run each): VITMI_WGRAD_OVERLAP = -1 / 0 / 1, VITMI_SMALL_FUSED = mask
    for env, setter in (("VITMI_WGRAD_OVERLAP", _lib.lib.vitmi_debug_set_wgrad_overlap),
                        ("VITMI_SMALL_FUSED", _lib.lib.vitmi_debug_set_small_fused)):
        if os.environ.get(env, "") != "":
            _lib.check(setter(int(os.environ[env])))
    from visiontransformer_amd.parallel import GradReducer, broadcast_parameters

    kw, kind, batch, gflop_img = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    f32 = kind == "plain_f32"
    masked = kind in ("mask", "pbf", "timm")
    torch.manual_seed(1001)
    if kind == "timm":
        from visiontransformer_amd.vision_transformer import VisionTransformer
        model = VisionTransformer(img_size=kw["image_size"], patch_size=kw["patch_size"], num_classes=kw["num_classes"],
                                  embed_dim=kw["embed_dim"], depth=kw["depth"], num_heads=kw["heads"]).to(dev)
    else:
        model = (ViT_mask if kind in ("mask", "pbf") else ViT).VIT(**kw).to(dev)
    if kind in ("mask", "pbf"):
        model.train_method = "mask"
    if f32:
        model.set_compute_dtype(torch.float32)
    model.train()
    if world > 1:
        broadcast_parameters(model, src=0)
    opt = FusedAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=5e-5)

    torch.manual_seed(1234 + rank)
    x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], device=dev)
    y = torch.randint(0, kw["num_classes"], (batch,), device=dev)
    unk = None
    dp_gen = None
    if kind == "dpsgd":
        from visiontransformer_amd import dpsgd
        dp_gen = torch.Generator(device=dev).manual_seed(4321)      # same state on every rank: one noise draw
    if masked:
        import random
        import numpy as np
        from visiontransformer_amd.masking_generator import MaskingGenerator
        grid = kw["image_size"] // kw["patch_size"]
        gen = MaskingGenerator((grid, grid), grid * grid // 2, min_num_patches=4)
    jigsaw = None
    if kind == "pbf":
        from visiontransformer_amd.masking_generator import JigsawPuzzleMaskedRegion
        bank = torch.randn(256, 3 * kw["patch_size"] ** 2, grid * grid, generator=torch.Generator().manual_seed(101))
        jigsaw = JigsawPuzzleMaskedRegion(kw["image_size"], kw["patch_size"], grid * grid // 2, 4, mask_type="pbf",
                                          device=dev, pub_data=bank)

    state = {"reducer": None, "graphed": None}

    def step(i):
        nonlocal unk
        reducer = state["reducer"]
        if dp_gen is not None:
            loss, _ = dpsgd.clipped_noised_gradients(model, x, y, 1.0, 1.0, generator=dp_gen, reducer=reducer)
            opt.step()
            return loss
        xin = x
        if masked:
            random.seed(1234 + i), np.random.seed(1234 + i)
            if jigsaw is not None:
                xin, um = jigsaw(x)
            else:
                um = MaskingGenerator.unk_mask_from_flat(gen.masked_indexes()[0])
            unk = ops.upload(um.astype(np.int64), dev)      # pinned staging: the host does not wait for the stream
        if state["graphed"] is not None:
            loss = state["graphed"](xin, y, unk)
            opt.step()
            return loss
        opt.zero_grad(set_to_none=True)
        logits = model(xin, unk_mask=unk) if masked else model(xin)
        loss = torch.nn.functional.cross_entropy(logits, y)
        if reducer is not None:
            reducer.mark_backward_start()
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step(grad_scale=1.0 / world)
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def attach_reducer(reserve, group):
        # the 1/world factor rides on the Adam kernel.  VITMI_BUCKET_DTYPE=bf16 halves the bytes on the links,
        # VITMI_RESERVE_CUS=k keeps k compute units free of the persistent GEMMs for the collective's workgroups;
        # buckets of one encoder layer each (28 MB of fp32 gradients; a bucket closes at >= 24 MiB): what is still on
        # the links when backward ends is the first layer's bucket, not the last three layers' of a 64 MiB bucket
        red = GradReducer(model, group=group, bucket_mb=int(os.environ.get("VITMI_BUCKET_MB", "24")), average=False,
                          dtype=torch.bfloat16 if os.environ.get("VITMI_BUCKET_DTYPE", "fp32") == "bf16"
                          else torch.float32, reserve_cus=reserve)
        red.record_timeline = True
        return red

    def measure(reserve, group):
        """Warm-up, family survey, the timed region; returns the JSON line of rank 0 (None elsewhere)."""
        if world > 1:
            if state["reducer"] is not None:
                state["reducer"].remove()
            state["reducer"] = attach_reducer(reserve, group)
        reducer = state["reducer"]
        # Kernel timers (two HIP event records per launch) cost GPU time themselves: every family is timed over the
        # last warm-up steps, which yields the per-family shares and names the dominant family; inside the timed
        # region only that family is timed, on every PROF_STRIDE-th of its launches.
        use_graph = args.graph
        if use_graph is None:
            use_graph = kw["image_size"] <= 64 and world == 1 and dp_gen is None
        if use_graph and (world > 1 or dp_gen is not None):
            raise SystemExit("--graph: one GPU, and not the DP-SGD workload")
        timers = not args.no_kernel_timers
        survey_steps = min(args.warmup, 2) if timers else 0
        if use_graph:
            survey_steps = 2 if timers else 0            # eager steps in front of the capture
            step(0)                                      # one eager step: lazy set-up, and a mask to size the capture
        else:
            for i in range(args.warmup - survey_steps):
                step(i)
        shares, dominant, survey = [], -1, None
        if survey_steps:
            sync()
            _lib.check(_lib.lib.vitmi_prof_select(-1))
            _lib.check(_lib.lib.vitmi_prof_reset())
            _lib.check(_lib.lib.vitmi_prof_enable(1))
            for i in range(args.warmup - survey_steps, args.warmup):
                step(i)
            sync()
            _lib.check(_lib.lib.vitmi_prof_enable(0))
            for fam in range(_lib.FAM_COUNT):
                ms, work, n = _lib.prof_read(fam)
                if n:
                    shares.append((ms / survey_steps, fam))
                    if not survey or ms > survey[0]:
                        survey = (ms, work, n)
            shares.sort(reverse=True)
            dominant = shares[0][1] if shares else -1
        if use_graph and state["graphed"] is None:
            from visiontransformer_amd.graphs import GraphedForwardBackward
            state["graphed"] = GraphedForwardBackward(model, torch.nn.functional.cross_entropy, x, y, unk)
            for i in range(args.warmup):
                step(i)
        sync()
        if timers and dominant >= 0 and not use_graph:
            _lib.check(_lib.lib.vitmi_prof_reset())
            _lib.check(_lib.lib.vitmi_prof_select(dominant))
            # every 5th launch of the family (5 is coprime to the 6 shapes x 12 layers of its launch sequence, so every
            # shape is sampled equally over the steps): timing all of them costs 3 % of the step in event records
            _lib.check(_lib.lib.vitmi_prof_stride(PROF_STRIDE))
            _lib.check(_lib.lib.vitmi_prof_enable(1))
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(args.warmup + i)
        sync()
        dt = time.perf_counter() - t0
        if timers and not use_graph:
            _lib.check(_lib.lib.vitmi_prof_enable(0))
            _lib.check(_lib.lib.vitmi_prof_select(-1))
            _lib.check(_lib.lib.vitmi_prof_stride(1))
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        loss_val = float(loss.item())
        exchange = None
        if reducer is not None:
            # the slowest rank's figures: exposed communication is what the step pays beyond the backward it hides under
            tl = reducer.timeline() or {}
            pair = torch.tensor([tl.get("exposed_comm_ms") or 0.0, tl.get("backward_ms") or 0.0], dtype=torch.float64,
                                device=dev)
            dist.all_reduce(pair, op=dist.ReduceOp.MAX)
            exchange = dict(reducer.info(), rccl_ranks=dist.get_world_size() if "nccl" in str(dist.get_backend()).lower() else 0,
                            rccl_max_nchannels=os.environ.get("NCCL_MAX_NCHANNELS"),
                            exposed_comm_ms=float(pair[0].item()), backward_ms=float(pair[1].item()),
                            bucket_issue_done_ms_rank0=tl.get("buckets"),
                            timeline_of="the last timed step; max over ranks; milliseconds from the start of backward")
        if rank != 0:
            return None
        imgs = batch * world * args.steps
        value = imgs / dt
        roof = None
        if timers and dominant >= 0:
            # graph replays carry no events: the dominant family's figures are then those of the eager survey steps
            ms, work, n = survey if use_graph else _lib.prof_read(dominant)
            name = _lib.lib.vitmi_prof_name(dominant).decode()
            is_flop = name.startswith(("gemm", "attn"))
            achieved = work / (ms * 1e-3) / (1e12 if is_flop else 1e9)
            peak = (F32_PEAK_TFLOPS if f32 else MFMA_PEAK_TFLOPS) if is_flop else HBM_PEAK_GBS
            roof = {"bound": "mfma" if is_flop else "hbm", "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s" if is_flop else "GB/s", "frac": achieved / peak, "traffic": None,
                    "kernel": name, "launches": n, "avg_launch_ms": ms / n,
                    "timed": (f"every launch of this family in {survey_steps} eager steps before the capture (the timed "
                              "region replays a graph, whose launches carry no events)") if use_graph else
                             f"every {PROF_STRIDE}th launch of this family inside the timed region ({args.steps} steps)",
                    "time_share_ms_per_step": {_lib.lib.vitmi_prof_name(f).decode(): round(m, 3) for m, f in shares},
                    "time_share_from": f"all families timed over the last {survey_steps} warm-up steps"}
            if f32 and is_flop:
                roof["peak_note"] = "fp32 FMA peak: the exact-fp32 kernels do not use bf16 MFMA"
        if roof is not None and args.workload == "vit_b16_224" and batch == 256:
            roof.update(pmc_traffic(roof["kernel"]))
        step_tflops = value / world * gflop_img * 1e9 / 1e12
        return {
            "metric": "train images/sec (node) ViT-B/16 224px" if args.workload == "vit_b16_224"
            else f"train images/sec (node) {args.workload}",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if f32 else "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {kw}, per-GPU batch {batch}, CE loss, fused Adam "
                                   "(fp32 master weights), N(0,1) images resident in HBM",
                       "global_batch": batch * world, "seq_len": (kw["image_size"] // kw["patch_size"]) ** 2 + 1,
                       "parallelism": f"dp{world}"},
            "graph": bool(use_graph),
            "train_gflop_per_image": gflop_img,
            "step_mfma_frac_per_gpu": step_tflops / (F32_PEAK_TFLOPS if f32 else MFMA_PEAK_TFLOPS),
            "final_loss": loss_val,
            # what the data-parallel exchange was: ranks in the process group (RCCL when the backend is nccl),
            # buckets and bytes per step, and where the step's time went around it
            "exchange": exchange,
            "roofline": roof,
        }

    def export_forward():
        """BASELINE config 2's second half: the forward pass with the probability export on (hooks on every attn_drop,
        the way vit_rollout.py:85-93 collects them).  HBM roofline on the attention kernel that writes the maps:
        algorithmic bytes per launch = the exported [B,H,N,N] tensor + qkv read + output and log-sum-exp written."""
        model.eval()
        seen = []
        hooks = [mod.register_forward_hook(lambda mm, i, o: seen.append(o)) for name, mod in model.named_modules()
                 if name.endswith("attn_drop")]

        def fwd():
            seen.clear()
            with torch.no_grad():
                model(x)

        for _ in range(3):
            fwd()
        sync()
        _lib.check(_lib.lib.vitmi_prof_reset())
        _lib.check(_lib.lib.vitmi_prof_select(_lib.FAM_ATTN_FWD))
        _lib.check(_lib.lib.vitmi_prof_stride(1))
        _lib.check(_lib.lib.vitmi_prof_enable(1))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fwd()
        sync()
        dt = (time.perf_counter() - t0) / args.steps
        _lib.check(_lib.lib.vitmi_prof_enable(0))
        _lib.check(_lib.lib.vitmi_prof_select(-1))
        ms, _, n = _lib.prof_read(_lib.FAM_ATTN_FWD)
        for h in hooks:
            h.remove()
        model.train()
        N = (kw["image_size"] // kw["patch_size"]) ** 2 + 1
        H, dh = kw["heads"], kw.get("dim_head", 8)
        probs_bytes = batch * H * N * N * 2
        launch_bytes = probs_bytes + batch * N * 3 * H * dh * 2 + batch * N * H * dh * 2 + batch * H * N * 4
        exported = sum(t.numel() * t.element_size() for t in seen)
        gbs = launch_bytes / (ms / n * 1e-3) / 1e9
        return {"images_per_s": batch / dt, "ms_per_forward": dt * 1e3, "export_bytes_per_image": exported / batch,
                "maps": f"{len(seen)} x {list(seen[0].shape)} bf16",
                "roofline": {"bound": "hbm", "kernel": "attn_fwd_bf16_kernel (probability export)", "achieved": gbs,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                             "launches": n, "avg_launch_ms": ms / n, "algorithmic_bytes_per_launch": launch_bytes,
                             "timed": f"every launch of the family over {args.steps} forward passes"}}

    sweeps = [reserve_default]
    if args.sweep_reserve and world > 1 and backend == "nccl":
        sweeps = [0, 8, 16]
    for reserve in sweeps:
        group = None
        if len(sweeps) > 1:
            # RCCL reads its channel cap when a communicator is created: one process group per setting
            if reserve > 0:
                os.environ["NCCL_MAX_NCHANNELS"] = str(reserve)
            else:
                os.environ.pop("NCCL_MAX_NCHANNELS", None)
            group = dist.new_group(ranks=list(range(world)), backend="nccl", pg_options=rccl_options())
        out = measure(reserve if (world > 1 and backend == "nccl") else 0, group)
        if rank == 0:
            if kind == "export" and world == 1:
                out["export"] = export_forward()
            out["cpu_baseline"] = None
            if not args.no_cpu_baseline and world == 1:   # timed on rank 0 of the single-GPU run only
                out["cpu_baseline"] = cpu_baseline(kw, kind, model)
            print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
