"""DataManager / Pipeline / Trainer surface (samnerf/datamanager.py:35-117, samnerf/sam_pipeline.py,
nerfstudio/pipelines/base_pipeline.py:203-281, nerfstudio/engine/trainer.py:409-440).

The datamanager here is the SYNTHETIC one of SURVEY.md 8(d): seeded random rays and targets generated on the
device.  Disk formats (transforms.json, SAM .npy, ClipSeg .pt) are a 'next' row."""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Type

import torch

from . import distributed as D
from . import ops as ops_mod
from .engine import AdamOptimizerConfig, ExponentialDecaySchedulerConfig, Optimizers
from .model import (AFTER_TRAIN_ITERATION, BEFORE_TRAIN_ITERATION, InstantiateConfig, SAMModelConfig, SceneBox)
from .rays import RayBundle


@dataclass
class CameraOptimizerConfig:
    mode: str = "off"


@dataclass
class SAMDataManagerConfig(InstantiateConfig):
    """samnerf/datamanager.py (config fields used by the samnerf method configs)."""
    _target: Type = field(default_factory=lambda: SyntheticSAMDataManager)
    train_num_rays_per_batch: int = 4096 * 4
    eval_num_rays_per_batch: int = 4096 * 4
    camera_optimizer: CameraOptimizerConfig = field(default_factory=CameraOptimizerConfig)
    patch_size: int = 1
    distill_sam: bool = True
    use_dino_feature: bool = False
    use_clipseg_feature: bool = False
    num_train_images: int = 2
    seed: int = 0


class SyntheticSAMDataManager:
    """next_train(step) -> (RayBundle, batch) with batch keys image / indices / sam / clipseg
    (samnerf/datamanager.py:97-117).  Rays: origins U(-.5,.5)^3, unit-normal directions; targets random."""

    def __init__(self, config: SAMDataManagerConfig, device="cuda", world_size: int = 1, local_rank: int = 0, **kw):
        self.config = config
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(config.seed + local_rank)  # per-rank ray draws (samnerf/train.py:87)
        self.train_count = 0
        self.num_train_data = config.num_train_images

    def _bundle(self, R: int) -> RayBundle:
        o = torch.rand((R, 3), device=self.device, generator=self.gen) - 0.5
        d = torch.randn((R, 3), device=self.device, generator=self.gen)
        d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
        return RayBundle(origins=o, directions=d, pixel_area=torch.full((R, 1), 1e-6, device=self.device),
                         camera_indices=torch.zeros((R, 1), dtype=torch.long, device=self.device))

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        self.train_count += 1
        c = self.config
        R = c.train_num_rays_per_batch
        batch = {"image": torch.rand((R, 3), device=self.device, generator=self.gen),
                 "indices": torch.zeros((R, 3), dtype=torch.long, device=self.device)}
        if c.distill_sam:
            n_sam = R // (c.patch_size ** 2) if c.patch_size > 1 else R
            batch["sam"] = torch.randn((n_sam, 256), device=self.device, generator=self.gen)
            if c.use_clipseg_feature:
                batch["clipseg"] = torch.randn((R, 192), device=self.device, generator=self.gen)
        return self._bundle(R), batch

    def next_train_into(self, step: int, origins, directions, image, targets: Dict[str, torch.Tensor]) -> None:
        """next_train(step) written into caller-owned buffers (the static step schedule, step_program.py): the same draws
        from the same generator in the same order, so both paths see identical batches."""
        self.train_count += 1
        c = self.config
        torch.rand(image.shape, device=self.device, generator=self.gen, out=image)
        if c.distill_sam:
            torch.randn(targets["sam"].shape, device=self.device, generator=self.gen, out=targets["sam"])
            if c.use_clipseg_feature:
                torch.randn(targets["clipseg"].shape, device=self.device, generator=self.gen, out=targets["clipseg"])
        torch.rand(origins.shape, device=self.device, generator=self.gen, out=origins)
        origins.sub_(0.5)
        torch.randn(directions.shape, device=self.device, generator=self.gen, out=directions)
        directions.div_(torch.linalg.norm(directions, dim=-1, keepdim=True))

    def get_param_groups(self) -> Dict[str, List]:
        return {}


@dataclass
class SamPipelineConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: SamPipeline)
    datamanager: SAMDataManagerConfig = field(default_factory=SAMDataManagerConfig)
    model: SAMModelConfig = field(default_factory=SAMModelConfig)


class SamPipeline(torch.nn.Module):
    """VanillaPipeline.get_train_loss_dict (base_pipeline.py:256-281) around SAMModel."""

    def __init__(self, config: SamPipelineConfig, device="cuda", test_mode="val", world_size: int = 1, local_rank: int = 0):
        super().__init__()
        self.config = config
        self.datamanager = config.datamanager.setup(device=device, world_size=world_size, local_rank=local_rank)
        self._model = config.model.setup(scene_box=SceneBox(), num_train_data=self.datamanager.num_train_data,
                                         device=device)
        self.world_size = world_size

    @property
    def model(self):
        return self._model

    @property
    def device(self):
        return self.model.device

    def get_train_loss_dict(self, step: int):
        ray_bundle, batch = self.datamanager.next_train(step)
        model_outputs = self.model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict

    def get_param_groups(self):
        return {**self.datamanager.get_param_groups(), **self.model.get_param_groups()}

    def get_training_callbacks(self, attrs=None):
        return self.model.get_training_callbacks(attrs)


@dataclass
class TrainerConfig(InstantiateConfig):
    """nerfstudio/engine/trainer.py TrainerConfig / configs/experiment_config.py (fields the samnerf configs set)."""
    _target: Type = field(default_factory=lambda: Trainer)
    method_name: Optional[str] = None
    steps_per_eval_batch: int = 500
    steps_per_eval_image: int = 500
    steps_per_save: int = 2000
    max_num_iterations: int = 1000000
    mixed_precision: bool = False
    pipeline: SamPipelineConfig = field(default_factory=SamPipelineConfig)
    optimizers: Dict[str, Any] = field(default_factory=dict)
    vis: str = "viewer+wandb"
    wandb_name: Optional[str] = None
    seed: int = 42


class Trainer:
    """Trainer.setup / train_iteration (trainer.py:137-205,409-440): callbacks, forward, summed loss, backward,
    gradient mean across ranks, fused Adam, scheduler step.  mixed_precision is accepted and ignored: the kernels
    compute in fp32 (the parity mode of SURVEY.md 7.1), so no GradScaler is needed."""

    def __init__(self, config: TrainerConfig, local_rank: int = 0, world_size: int = 1, device="cuda") -> None:
        self.config = config
        self.local_rank, self.world_size = local_rank, world_size
        self.device = device
        self._start_step = 0
        # run the nerf / SAM-head / ClipSeg-head tasks on separate HIP streams (SNF_OVERLAP=0: one stream, for A/B runs)
        self.overlap = os.environ.get("SNF_OVERLAP", "1") == "1"
        self.enqueue_order = "heads_first"
        self.pipeline_steps = False  # True: do not join the head streams at the end of a step (see train_iteration)
        # "auto" | "sam" | "clipseg" | "own" | "main": the stream of the forward-time sorts (autotune_streams may pick "own")
        self.presort_host = os.environ.get("SNF_PRESORT_HOST", "auto")
        self._side = None
        # the step as a static launch schedule (step_program.py) instead of an autograd graph: same kernels, same arguments,
        # ~10x less host time per step, on one rank or many.  SNF_STATIC_STEP=0 keeps the eager autograd path.
        self.static_step = os.environ.get("SNF_STATIC_STEP", "1") == "1"
        self._program = None
        self._program_off = None  # reason the static schedule is not used
        # Optimizer semantics on steps where the proposal network gets no gradient (4 of 5 after warm-up,
        # ray_samplers.py:566-591).  The reference pins torch < 2 (requirements.txt:32): its zero_grad() zero-FILLS the
        # gradients, so torch.optim.Adam still steps the group -- moments decay, the step count advances, parameters move by
        # lr * m_hat / (sqrt(v_hat) + eps).  That is the default here.  SNF_TORCH2_NONE_GRADS=1 selects what torch >= 2 does
        # with the same script (grad = None: the group is skipped).
        self.zero_grad_adam = os.environ.get("SNF_TORCH2_NONE_GRADS", "0") != "1"

    def setup(self, test_mode="val") -> None:
        self.pipeline = self.config.pipeline.setup(device=self.device, test_mode=test_mode, world_size=self.world_size,
                                                   local_rank=self.local_rank)
        self.pipeline.model.train()
        arenas = self.pipeline.model.build_arenas(with_optimizer_state=True)
        self.optimizers = Optimizers(self.config.optimizers, arenas)
        if "sam_field" in arenas:  # the two feature heads are stepped (and sharded) as separate slices of one arena
            a = arenas["sam_field"]
            names = list(a.offsets)
            self.optimizers.shard_slices["sam_field"] = [
                (a.offsets[names[lo_i]][0], a.offsets[names[hi_i]][0] if hi_i < len(names) else a.numel)
                for lo_i, hi_i in self._head_param_ranges().values() if hi_i > lo_i]
        D.broadcast_parameters([a.param for a in arenas.values()])
        self.callbacks = self.pipeline.get_training_callbacks()

    # The feature heads see the nerfacto branch only through DETACHED quantities (sample positions sam_field.py:116,
    # sam_weights.detach() sam_model.py:260-277) and do not see each other, so a train step is three independent tasks
    # after the nerfacto forward:
    #   main stream     : nerf losses -> backward {fields, proposal_networks} -> gradient exchange -> Adam
    #   'sam' stream    : SAM head forward -> loss -> backward {sam grids, sam MLP, conv} -> exchange -> Adam (0.4 GB)
    #   'clipseg' stream: ClipSeg head forward -> loss -> backward {clipseg grids, MLP} -> exchange -> Adam (0.4 GB)
    # They are enqueued back to back and run concurrently on the GPU (autograd executes a node's backward on the stream
    # of its forward); the streams are joined at the end of the step.
    HEAD_LOSS = {"sam": "sam_loss", "clipseg": "clipseg_loss"}

    def _head_param_ranges(self):
        """{head: (first, last)} parameter index ranges inside the 'sam_field' arena (module registration order:
        clip_encs.*, sam_net, clipseg_encs.*, clipseg_net -- samnerf/sam_field.py:38-94)."""
        sf = self.pipeline.model.sam_field
        n_sam = len(list(sf.clip_encs.parameters())) + len(list(sf.sam_net.parameters()))
        n_all = len(list(sf.parameters()))
        return {"sam": (0, n_sam), "clipseg": (n_sam, n_all)}

    def train_iteration(self, step: int):
        for cb in self.callbacks:
            cb.run_callback_at_location(step, BEFORE_TRAIN_ITERATION)
        opt = self.optimizers
        model = self.pipeline.model
        if self.static_step and self._program_off is None:
            if self._program is None:
                from .step_program import StepProgram
                self._program_off = StepProgram.unsupported_reason(self)
                if self._program_off is None:
                    self._program = StepProgram(self)
            if self._program is not None:
                loss, loss_dict, metrics_dict = self._program.run(step)
                opt.scheduler_step_all(step)
                for cb in self.callbacks:
                    cb.run_callback_at_location(step, AFTER_TRAIN_ITERATION)
                return loss, loss_dict, metrics_dict
        opt.arm_fused_adam()
        use_side = self.overlap and torch.cuda.is_available() and "sam_field" in opt.arenas
        if use_side and self._side is None:
            from . import ops
            self._side = {"sam": ops.make_stream("sam"), "clipseg": ops.make_stream("clipseg")}
        if hasattr(model, "feature_streams"):
            model.feature_streams = self._side if use_side else None
        # the forward-time sorts ride on the SAM head's stream, which is idle until the nerfacto forward has produced the
        # weights: three streams in all, one hardware queue each ("own": a fourth stream -- ~1 % faster when it works, 15 %
        # slower in the runtime's slow mode)
        host = "sam" if self.presort_host == "auto" else self.presort_host
        ops_mod.PRESORT_HOST_STREAM = self._side[host] if (use_side and host in ("sam", "clipseg")) else None
        _, loss_dict, metrics_dict = self.pipeline.get_train_loss_dict(step=step)
        prop_updated = getattr(getattr(model, "proposal_sampler", None), "last_updated", True)
        head_losses = {h: loss_dict[k] for h, k in self.HEAD_LOSS.items() if k in loss_dict}
        rest = [v for k, v in loss_dict.items() if k not in self.HEAD_LOSS.values()]
        if use_side and len(head_losses) > 0:
            main = torch.cuda.current_stream()
            ranges = self._head_param_ranges()
            rest_groups = [g for g in opt.arenas if g not in ("sam_field", "conv")]
            loss_rest = sum(rest)
            loss = loss_rest.detach()

            def nerf_task():  # on the main stream
                loss_rest.backward()
                for g in rest_groups:
                    if g == "proposal_networks" and not prop_updated and not self.zero_grad_adam:
                        continue  # torch >= 2 semantics: grad is None on a non-update step, Adam skips the group
                    opt.exchange_and_step(g)

            # host enqueue order (the GPU runs the three tasks concurrently; a task cannot start before the host has
            # issued it): "heads_first" gets the two Adam-heavy head tasks going before the nerf backward is issued
            if self.enqueue_order != "heads_first":
                nerf_task()
            # one task per feature head on its own stream
            first_head = True
            for h, lv in head_losses.items():
                st = self._side[h]
                lo_i, hi_i = ranges[h]
                with torch.cuda.stream(st):
                    lv.backward()
                    opt.exchange_and_step("sam_field", lo_i, hi_i, count_step=first_head)
                    if h == "sam" and "conv" in opt.arenas:
                        opt.exchange_and_step("conv")
                first_head = False
            if self.enqueue_order == "heads_first":
                nerf_task()
            if self.pipeline_steps:
                # no join: the next step's nerfacto forward/backward (main stream) may run under the tails of this step's
                # head tasks.  Nothing is stale -- every parameter group is read and updated on one stream only -- it just
                # removes a false dependency.  `loss` then covers the main-stream terms; call synchronize() before
                # reading head losses on the host.
                pass
            else:
                for h in head_losses:
                    main.wait_stream(self._side[h])
                for lv in head_losses.values():
                    loss = loss + lv.detach()
        else:
            loss = sum(loss_dict.values())
            loss.backward()
            # same arena slices as the three-task schedule (the sharded optimizer's layout must not depend on the schedule)
            for g in opt.arenas:
                if g == "proposal_networks" and not prop_updated and not self.zero_grad_adam:
                    continue
                if g == "sam_field":
                    first = True
                    for lo_i, hi_i in self._head_param_ranges().values():
                        if hi_i > lo_i:
                            opt.exchange_and_step(g, lo_i, hi_i, count_step=first)
                            first = False
                else:
                    opt.exchange_and_step(g)
            loss = loss.detach()
        opt.disarm_fused_adam()
        opt.scheduler_step_all(step)
        for cb in self.callbacks:
            cb.run_callback_at_location(step, AFTER_TRAIN_ITERATION)
        return loss, loss_dict, metrics_dict

    def autotune_streams(self, steps: int = 10, warm: int = 3, verbose: bool = False) -> Dict[str, float]:
        """Pick the stream layout of the step on THIS device by timing it (like cudnn.benchmark picks an algorithm).

        How the runtime spreads HIP streams over its hardware queues differs between otherwise identical boxes and between
        runs: with the forward-time backward sorts on their own (fourth) stream a step is ~1 % faster most of the time and
        15 % slower some of the time.  Runs `warm + steps` real train iterations per candidate on the trainer's own data, keeps the
        fastest, and restores parameters, optimizer state, step counters and random generators, so training starts from the
        state it would have started from without the probe."""
        from . import ops
        if not torch.cuda.is_available():
            return {}
        opt, model, dm = self.optimizers, self.pipeline.model, self.pipeline.datamanager
        snap = {k: [t.clone() for t in (a.param, a.grad, a.exp_avg, a.exp_avg_sq)] for k, a in opt.arenas.items()}
        counts = (dict(opt.step_count), dict(opt.sched_step))
        ps = getattr(model, "proposal_sampler", None)
        ps_state = (ps._steps_since_update, ps._step, ps.last_updated) if ps is not None else None
        gens = [(g, g.get_state()) for g in (getattr(dm, "gen", None),) if g is not None]
        cuda_rng, cpu_rng = torch.cuda.get_rng_state(), torch.get_rng_state()
        dm_count = getattr(dm, "train_count", None)
        results = {}
        host0 = self.presort_host
        for name, host in (("sorts on the sam stream", "sam"), ("sorts on a fourth stream", "own")):
            self.presort_host = host
            for i in range(warm):
                self.train_iteration(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                self.train_iteration(warm + i)
            torch.cuda.synchronize()
            results[name] = (time.perf_counter() - t0) / steps * 1e3
        best = min(results, key=results.get)
        # a fourth stream has to win clearly: its slow mode comes and goes within a process
        self.presort_host = "own" if results["sorts on a fourth stream"] < 0.97 * results["sorts on the sam stream"] else "sam"
        del host0
        # back to the state before the probe
        for k, a in opt.arenas.items():
            for dst, src in zip((a.param, a.grad, a.exp_avg, a.exp_avg_sq), snap[k]):
                dst.copy_(src)
        opt.step_count.update(counts[0])
        opt.sched_step.update(counts[1])
        if ps is not None:
            ps._steps_since_update, ps._step, ps.last_updated = ps_state
        for g, st in gens:
            g.set_state(st)
        torch.cuda.set_rng_state(cuda_rng)
        torch.set_rng_state(cpu_rng)
        if dm_count is not None:
            dm.train_count = dm_count
        torch.cuda.synchronize()
        if verbose:
            print(f"[autotune_streams] {results} -> {best}", flush=True)
        return results

    def synchronize(self) -> None:
        """Join the task streams (needed before reading results of the head tasks when pipeline_steps is on)."""
        if self._side is not None and torch.cuda.is_available():
            main = torch.cuda.current_stream()
            for st in self._side.values():
                main.wait_stream(st)
        if self._program is not None and torch.cuda.is_available():
            # (without feature heads the proposal backward + Adam and the next prologue live on the schedule's own side stream)
            self._program.join_side_streams()

    def save_checkpoint(self, path: str, step: int) -> None:
        """trainer.py:379-406: {step, pipeline state_dict, optimizers}.  COLLECTIVE on a multi-rank run: every rank calls it
        (the table-parallel levels and the sharded Adam moments are gathered first), rank 0 writes the file -- the reference's
        `if self.local_rank == 0: save` pattern around this call would deadlock in consolidate_state."""
        self.synchronize()
        if self._program is not None:
            torch.cuda.synchronize()
            self._program.fold_guards()  # steps a non-finite loss vetoed do not count (GradScaler.step semantics)
        self.optimizers.consolidate_state()
        if not D.collectives_on() or torch.distributed.get_rank() == 0:
            torch.save({"step": step, "pipeline": self.pipeline.state_dict(), "optimizers": self.optimizers.state_dict()}, path)
        if D.collectives_on():
            torch.distributed.barrier()

    def load_checkpoint(self, path: str) -> int:
        st = torch.load(path, map_location=self.device)
        self.pipeline.load_state_dict(st["pipeline"])
        self.optimizers.load_optimizers(st["optimizers"])
        self._start_step = st["step"] + 1
        return self._start_step
