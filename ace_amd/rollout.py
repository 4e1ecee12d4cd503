"""RolloutEngine: the stepper loop (fme/ace/stepper/single_module.py:1124-1167) on static HBM
buffers with the per-step forward replayed from a hipGraph.

Per step s the engine enqueues
  1. ace_pack_normalize    gather state (previous output or the initial condition) and the forcing
                           slice (index s, or s+1 for next_step_forcing_names) -> packed, normalised input
  2. the SFNO forward      (`graph="step"`: the library's captured hipGraph; `graph="window"`: steps 1-3
                           of the whole window live in one torch-captured hipGraph; `graph=None`: eager launches)
  3. ace_unpack_denormalize  network output -> out[name][:, s] (denormalised), which is also step s+1's state
     (residual_prediction: the normalised prognostic inputs are added to the network output first; post-step hooks and
     prescribed prognostics then edit out[name][:, s] in place)
so the numbers are bit-identical to `Stepper.predict` (same kernels, same (x-mean)/std and y*std+mean roundings).
Nothing is allocated and the host never synchronises inside the window."""

import os
from typing import Dict, Mapping, Optional, Tuple

import torch

from . import _lib
from .stepper import Stepper


class RolloutEngine:
    # graph="window" with a noise-conditioned net keeps the conditioning fields of all T steps of a window in one static device
    # buffer (T x B x cond_channels x H x W fp32) that the captured steps read; beyond this size the engine refuses and names
    # graph="step" (ADVICE r05).  8 GiB = a 40-step window of ~290 conditioning channels at 1 degree.
    max_window_conditioning_bytes = 8 << 30

    def __init__(self, stepper: Stepper, batch: int, n_forward_steps: int, graph: Optional[str] = "step"):
        if graph not in (None, "step", "window"):
            raise ValueError("graph must be None, 'step' or 'window'")
        step = stepper._step_obj
        cfg = step.config
        # secondary decoder (fme/core/step/secondary_decoder.py): column-local diagnostics decoded from the network's normalised
        # output tensor - one more native module call on the static y buffer per step and a second fused unpack
        self._secondary = getattr(step, "secondary_decoder", None)
        if self._secondary is not None and graph == "window":
            raise NotImplementedError("graph='window' with a secondary decoder: its module call is not captured; use graph='step'")
        if getattr(stepper, "_masks", False):
            raise NotImplementedError("RolloutEngine: static spatial masking of the step inputs / outputs (input_masking, the dataset's "
                                      "mask provider) is applied by Stepper.predict")
        # multi-call diagnostics (fme/core/step/multi_call.py): per multiplier one more pack (the scaled forcing) / forward / unpack
        # into scratch planes / hooks per step, selected outputs copied under their suffixed names
        self._mc = getattr(stepper, "_multi_call_config", None)
        if self._mc is not None and graph == "window":
            raise NotImplementedError("graph='window' with multi-call diagnostics is not captured; use graph='step'")
        # post-step hooks (corrector, prescribed-SST ocean): torch ops on the static buffers between the fused unpack of
        # step s and the pack of step s + 1 - stream ordered, no host synchronisation.  The corrector state (dry-air
        # reference mass) is seeded by the first step after load() and survives continue_from_last().
        self._corrector = step._corrector
        self._ocean = step._ocean
        self._corrector_state = None
        # graph="window" captures the hooks with the steps.  The one piece of hook state - the dry-air reference mass the
        # corrector seeds on its first step (fme/core/corrector/state.py) - lives in a static device buffer with a device
        # flag, selected branch-free inside the captured region, so a replay neither re-seeds nor forgets it.
        self._ref_mass = None       # (B, 1, 1) fp64, allocated on first use
        self._have_ref = None       # device bool scalar
        self.stepper = stepper
        self.net = step.module.torch_module
        self._conditioned = hasattr(self.net, "draw_noise")
        self._labels = None          # (B, n_labels) tensor in the module's label encoding: set_labels()
        # seedable random source of a stochastic module (fme/core/random_state.py; StepperState.random_state): active around every
        # network call of a window, as Stepper.step activates it (single_module.py:1063-1068).  The draw happens on the host,
        # outside any captured region, and reaches the device as a plain copy.
        self._random_state = None
        # graph="window" with a noise-conditioned net: the draws of the whole window are made on the host BEFORE the replay, in step
        # order (the same generator consumption as the eager loop), into a static (T, B, cond, H, W) buffer the captured steps read
        self._cond_window = None
        self.B, self.T = batch, n_forward_steps
        self.H, self.W = step._img_shape
        self.HW = self.H * self.W
        self.in_names, self.out_names = list(cfg.in_names), list(cfg.out_names)
        self.prognostic = [n for n in self.out_names if n in self.in_names]
        self.forcing_names = [n for n in self.in_names if n not in self.out_names]
        self.next_step_forcing = set(cfg.next_step_forcing_names)
        self.graph_mode = graph
        dev = next(self.net.parameters()).device
        self.device = dev
        B, T, H, W = batch, n_forward_steps, self.H, self.W
        f32 = dict(dtype=torch.float32, device=dev)
        self.x = torch.zeros(B, len(self.in_names), H, W, **f32)    # packed normalised network input
        self.y = torch.zeros(B, len(self.out_names), H, W, **f32)   # network output (normalised)
        self.ic = {n: torch.zeros(B, 1, H, W, **f32) for n in self.prognostic}
        self.forcing = {n: torch.zeros(B, T + 1, H, W, **f32) for n in self.forcing_names}
        # next-step data the hooks read that is not a network forcing input (e.g. the prescribed SST: prognostic AND target)
        extra = set(cfg.ocean.forcing_names) if cfg.ocean is not None else set()
        extra |= set(cfg.prescribed_prognostic_names)
        self.prescribed = list(cfg.prescribed_prognostic_names)
        self.target_names = sorted(extra - set(self.forcing_names))
        self.target = {n: torch.zeros(B, T + 1, H, W, **f32) for n in self.target_names}
        self.out = {n: torch.zeros(B, T, H, W, **f32) for n in self.out_names}
        self.sec_names = list(cfg.secondary_decoder.secondary_diagnostic_names) if self._secondary is not None else []
        for n in self.sec_names:
            self.out[n] = torch.zeros(B, T, H, W, **f32)
        # The secondary decoder's diagnostics are unpacked straight into their output planes: they do not pass through the post-step
        # hooks here (the reference and Stepper.predict put them in the dict the corrector and the ocean see).  Refuse the
        # configurations where that difference would show instead of raising KeyError mid-rollout.
        # (the ocean's surface temperature cannot clash: OceanConfig.build requires it among the step's in_names AND out_names,
        # and a secondary diagnostic may be neither - step.py's configuration checks)
        hooked = set(getattr(self._corrector, "force_positive_names", []) or [])
        clash = sorted(hooked.intersection(self.sec_names))
        if clash:
            raise NotImplementedError(f"secondary-decoder diagnostics {clash} are also touched by the post-step hooks (force_positive / "
                                      "ocean): use Stepper.predict for this configuration")
        norm = step.normalizer
        # normalizer.py:212-242: NaNs become 0 in normalised space (inputs) / the variable's mean (outputs).  The fused pack / unpack
        # kernels do not replace NaNs: one in-place pass over the packed tensor on either side when the normaliser asks for it
        self._fill_in, self._fill_out = bool(norm.fill_nans_on_normalize), bool(norm.fill_nans_on_denormalize)
        self.in_mean = torch.stack([norm.means[n].to(dev) for n in self.in_names]).contiguous()
        self.in_std = torch.stack([norm.stds[n].to(dev) for n in self.in_names]).contiguous()
        self.out_mean = torch.stack([norm.means[n].to(dev) for n in self.out_names]).contiguous()
        self.out_std = torch.stack([norm.stds[n].to(dev) for n in self.out_names]).contiguous()
        # per-step pointer tables (device arrays of device pointers) and per-sample strides
        src_ptrs, src_strides, dst_ptrs = [], [], []
        for s in range(T):
            ptrs, strides = [], []
            for n in self.in_names:
                if n in self.ic:
                    if s == 0:
                        ptrs.append(self.ic[n].data_ptr()); strides.append(self.HW)
                    else:
                        ptrs.append(self.out[n].data_ptr() + 4 * (s - 1) * self.HW); strides.append(T * self.HW)
                else:
                    t = s + 1 if n in self.next_step_forcing else s
                    ptrs.append(self.forcing[n].data_ptr() + 4 * t * self.HW); strides.append((T + 1) * self.HW)
            src_ptrs.append(ptrs); src_strides.append(strides)
            dst_ptrs.append([self.out[n].data_ptr() + 4 * s * self.HW for n in self.out_names])
        i64 = dict(dtype=torch.int64, device=dev)
        self._src_ptrs = torch.tensor(src_ptrs, **i64)
        self._src_strides = torch.tensor(src_strides, **i64)
        self._dst_ptrs = torch.tensor(dst_ptrs, **i64)
        self._dst_strides = torch.full((len(self.out_names),), T * self.HW, **i64)
        # plain ints for the per-step launches (no tensor indexing on the hot host path)
        nin, nout = len(self.in_names), len(self.out_names)
        self._src_ptr_addr = [self._src_ptrs.data_ptr() + 8 * s * nin for s in range(T)]
        self._src_stride_addr = [self._src_strides.data_ptr() + 8 * s * nin for s in range(T)]
        self._dst_ptr_addr = [self._dst_ptrs.data_ptr() + 8 * s * nout for s in range(T)]
        if self.sec_names:
            self.sec_mean = torch.stack([norm.means[n].to(dev) for n in self.sec_names]).contiguous()
            self.sec_std = torch.stack([norm.stds[n].to(dev) for n in self.sec_names]).contiguous()
            self._sec_dst_ptrs = torch.tensor([[self.out[n].data_ptr() + 4 * s * self.HW for n in self.sec_names] for s in range(T)], **i64)
            self._sec_dst_strides = torch.full((len(self.sec_names),), T * self.HW, **i64)
            self._sec_dst_ptr_addr = [self._sec_dst_ptrs.data_ptr() + 8 * s * len(self.sec_names) for s in range(T)]
        self._mc_state = None
        if self._mc is not None:
            from .multi_call import get_multi_call_name
            mc = self._mc
            if mc.forcing_name not in self.forcing_names:
                raise NotImplementedError(f"RolloutEngine: the multi-call forcing '{mc.forcing_name}' must be an input-only network input")
            if any(n not in self.out_names for n in mc.output_names):
                raise NotImplementedError("RolloutEngine: multi-call outputs must be network outputs (not secondary-decoder diagnostics)")
            self._mc_factors = list(mc.forcing_multipliers.items())
            self._mc_copy = [[(n, get_multi_call_name(n, suffix)) for n in mc.output_names] for suffix, _ in self._mc_factors]
            for pairs in self._mc_copy:
                for _, new in pairs:
                    self.out[new] = torch.zeros(B, T, H, W, **f32)
            self.forcing_mc = [torch.zeros(B, T + 1, H, W, **f32) for _ in self._mc_factors]     # the scaled forcing, per multiplier
            self.mc_scratch = {n: torch.zeros(B, 1, H, W, **f32) for n in self.out_names}        # one step's outputs of a re-evaluation
            fi = self.in_names.index(mc.forcing_name)
            self._mc_src_ptrs = []
            for k in range(len(self._mc_factors)):
                tab = self._src_ptrs.clone()
                for st in range(T):
                    t = st + 1 if mc.forcing_name in self.next_step_forcing else st
                    tab[st, fi] = self.forcing_mc[k].data_ptr() + 4 * t * self.HW
                self._mc_src_ptrs.append(tab)
            self._mc_src_ptr_addr = [[tab.data_ptr() + 8 * st * nin for st in range(T)] for tab in self._mc_src_ptrs]
            self._mc_dst_ptrs = torch.tensor([self.mc_scratch[n].data_ptr() for n in self.out_names], **i64)
            self._mc_dst_strides = torch.full((len(self.out_names),), self.HW, **i64)
        # residual prediction (single_module.py:663-664): normalised prognostic inputs are added to the network output
        self._res_in = self._res_out = None
        if cfg.residual_prediction:
            self._res_in = torch.tensor([self.in_names.index(n) for n in self.prognostic], **i64)
            self._res_out = torch.tensor([self.out_names.index(n) for n in self.prognostic], **i64)
        # Post-step physics as four HIP kernels per step (ace_amd/physics.py, csrc/physics.hip) instead of captured torch ops
        # (round 2: +29 % step time).  ACE_NO_FUSED_PHYSICS=1 keeps the torch ops (A/B).
        self._physics = None
        slab = self._ocean is not None and getattr(self._ocean, "is_slab", False)   # the fused kernel knows the prescribed SST only
        if (self._corrector is not None or self._ocean is not None or self.prescribed) and not slab and not os.environ.get("ACE_NO_FUSED_PHYSICS"):
            self._physics = self._build_physics()
        self._window_graph = None
        self._window_graph_key = None   # (native handle, its weights generation) the window graph was captured against
        self.net._ensure_native(dev, B)
        self.net.sync_weights()

    def _build_physics(self):
        from .physics import FusedPhysics
        T, HW = self.T, self.HW

        def locate_gen(name, s):
            return (self.out[name].data_ptr() + 4 * s * HW, T * HW) if name in self.out else None

        def locate_in(name, s):
            if name in self.ic:
                return (self.ic[name].data_ptr(), HW) if s == 0 else (self.out[name].data_ptr() + 4 * (s - 1) * HW, T * HW)
            if name in self.forcing:
                t = s + 1 if name in self.next_step_forcing else s
                return (self.forcing[name].data_ptr() + 4 * t * HW, (T + 1) * HW)
            return None

        def locate_next(name, s):
            for src in (self.forcing, self.target):
                if name in src:
                    return (src[name].data_ptr() + 4 * (s + 1) * HW, (T + 1) * HW)
            return None

        return FusedPhysics(self._corrector, self._ocean, self.prescribed, self.B, (self.H, self.W), T,
                            gen_names=self.out_names, in_names=self.in_names,
                            next_names=self.forcing_names + self.target_names,
                            locate_gen=locate_gen, locate_in=locate_in, locate_next=locate_next, device=self.device)

    # -- one step, enqueued on the current stream
    def _enqueue_step(self, s: int, use_library_graph: bool):
        L = _lib.lib()
        stream = _lib.current_stream()
        nin, nout = len(self.in_names), len(self.out_names)
        _lib.check(L.ace_pack_normalize(self._src_ptr_addr[s], self._src_stride_addr[s],
                                        self.in_mean.data_ptr(), self.in_std.data_ptr(), self.x.data_ptr(),
                                        self.B, nin, self.HW, stream))
        if self._fill_in:
            torch.nan_to_num_(self.x, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        if self._conditioned:   # NoiseConditionedSFNO: fresh conditioning noise every step (stochastic_sfno.py:128-146), merged
            if self.graph_mode == "window":      # drawn for the whole window by _draw_window_conditioning(), static address per step
                noise = self._cond_window[s]
            else:
                noise = self.net.conditioning_field(self.B, self.device, labels=self._labels)   # with the label / positional context
            _lib.check(L.ace_sfno_forward_conditioned(self.net._native, self.x.data_ptr(), noise.data_ptr(), self.y.data_ptr(),
                                                      self.B, stream))
        else:
            fwd = L.ace_sfno_forward_graph if use_library_graph else L.ace_sfno_forward
            _lib.check(fwd(self.net._native, self.x.data_ptr(), self.y.data_ptr(), self.B, stream))
        if self._secondary is not None:      # decoded from the raw network output, before the residual is added (single_module.py:430-434)
            sec = self._secondary._module(self.y).contiguous()
            _lib.check(L.ace_unpack_denormalize(sec.data_ptr(), self.sec_mean.data_ptr(), self.sec_std.data_ptr(),
                                                self._sec_dst_ptr_addr[s], self._sec_dst_strides.data_ptr(),
                                                self.B, len(self.sec_names), self.HW, stream))
        if self._res_in is not None:
            self.y.index_add_(1, self._res_out, self.x.index_select(1, self._res_in))
        if self._fill_out:      # 0 in normalised space denormalises to the mean
            torch.nan_to_num_(self.y, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        _lib.check(L.ace_unpack_denormalize(self.y.data_ptr(), self.out_mean.data_ptr(), self.out_std.data_ptr(),
                                            self._dst_ptr_addr[s], self._dst_strides.data_ptr(),
                                            self.B, nout, self.HW, stream))
        if self._physics is not None:   # corrector -> ocean -> prescribed prognostics (single_module.py:669-716), four launches
            self._physics.apply(s, stream)
        else:
            if self._corrector is not None or self._ocean is not None:
                self._apply_hooks(s)
            for n in self.prescribed:     # after the ocean (single_module.py:700-716): overwritten from the data of step s + 1
                self.out[n][:, s].copy_(self.target[n][:, s + 1] if n in self.target else self.forcing[n][:, s + 1])
        if self._mc is not None:
            self._multi_call_step(s, use_library_graph)

    def _multi_call_step(self, s: int, use_library_graph: bool):
        """_multi_call.py:164-188 on the static buffers: the step of window position s evaluated again with the scaled forcing - same
        x / y buffers as the plain evaluation (whose outputs are already unpacked), outputs into scratch planes, the post-step hooks
        as torch ops on them (their corrector state is seeded like the plain path's and never fed back), selected fields copied."""
        L = _lib.lib()
        stream = _lib.current_stream()
        nin, nout = len(self.in_names), len(self.out_names)
        name = self._mc.forcing_name
        for k, (suffix, factor) in enumerate(self._mc_factors):
            _lib.check(L.ace_pack_normalize(self._mc_src_ptr_addr[k][s], self._src_stride_addr[s], self.in_mean.data_ptr(),
                                            self.in_std.data_ptr(), self.x.data_ptr(), self.B, nin, self.HW, stream))
            if self._fill_in:
                torch.nan_to_num_(self.x, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
            if self._conditioned:   # a fresh noise draw per evaluation, as the reference's repeated module call makes
                noise = self.net.conditioning_field(self.B, self.device, labels=self._labels)
                _lib.check(L.ace_sfno_forward_conditioned(self.net._native, self.x.data_ptr(), noise.data_ptr(), self.y.data_ptr(),
                                                          self.B, stream))
            else:
                fwd = L.ace_sfno_forward_graph if use_library_graph else L.ace_sfno_forward
                _lib.check(fwd(self.net._native, self.x.data_ptr(), self.y.data_ptr(), self.B, stream))
            if self._res_in is not None:
                self.y.index_add_(1, self._res_out, self.x.index_select(1, self._res_in))
            if self._fill_out:
                torch.nan_to_num_(self.y, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
            _lib.check(L.ace_unpack_denormalize(self.y.data_ptr(), self.out_mean.data_ptr(), self.out_std.data_ptr(),
                                                self._mc_dst_ptrs.data_ptr(), self._mc_dst_strides.data_ptr(), self.B, nout, self.HW, stream))
            gen = {n: self.mc_scratch[n][:, 0] for n in self.out_names}
            if self._corrector is not None or self._ocean is not None or self.prescribed:
                inp = {n: (self.ic[n][:, 0] if s == 0 else self.out[n][:, s - 1]) for n in self.prognostic}
                for n in self.forcing_names:
                    src = self.forcing_mc[k] if n == name else self.forcing[n]
                    inp[n] = src[:, s + 1 if n in self.next_step_forcing else s]
                nxt = {n: (self.forcing_mc[k] if n == name else self.forcing[n])[:, s + 1] for n in self.forcing_names}
                nxt.update({n: self.target[n][:, s + 1] for n in self.target_names})
                new = gen
                if self._corrector is not None:
                    new, state = self._corrector(inp, new, nxt, self._mc_state)
                    if self._mc_state is None:
                        self._mc_state = state
                if self._ocean is not None:
                    new = self._ocean(inp, new, nxt)
                for n in self.prescribed:
                    new = {**new, n: nxt[n]}
                gen = new
            for src_name, dst_name in self._mc_copy[k]:
                self.out[dst_name][:, s].copy_(gen[src_name])

    def _apply_hooks(self, s: int):
        """step_with_adjustments' tail (fme/core/step/single_module.py:669-716) on the window buffers of step s."""
        inp = {n: (self.ic[n][:, 0] if s == 0 else self.out[n][:, s - 1]) for n in self.prognostic}
        for n in self.forcing_names:
            inp[n] = self.forcing[n][:, s + 1 if n in self.next_step_forcing else s]
        gen = {n: self.out[n][:, s] for n in self.out_names}
        nxt = {n: self.forcing[n][:, s + 1] for n in self.forcing_names}
        nxt.update({n: self.target[n][:, s + 1] for n in self.target_names})
        new = gen
        if self._corrector is not None:
            if self.graph_mode == "window" and "conserve_dry_air" in self._corrector.corrections:
                from .corrector import CorrectorState, _seed_global_dry_air_mass
                c = self._corrector
                seeded = _seed_global_dry_air_mass(inp, None, c._mean, c._vcoord(self.device), torch.float64).global_dry_air_mass
                if self._ref_mass is None:
                    self._ref_mass = torch.zeros_like(seeded)
                    self._have_ref = torch.zeros((), dtype=torch.bool, device=self.device)
                    if self._corrector_state is not None and self._corrector_state.global_dry_air_mass is not None:
                        self._ref_mass.copy_(self._corrector_state.global_dry_air_mass)
                        self._have_ref.fill_(True)
                self._ref_mass.copy_(torch.where(self._have_ref, self._ref_mass, seeded))
                self._have_ref.fill_(True)
                self._corrector_state = CorrectorState(global_dry_air_mass=self._ref_mass)
            new, self._corrector_state = self._corrector(inp, new, nxt, self._corrector_state)
        if self._ocean is not None:
            new = self._ocean(inp, new, nxt)
        for n in self.out_names:
            if new[n] is not gen[n]:
                gen[n].copy_(new[n])

    def set_labels(self, labels) -> None:
        """Labels of a conditional module (fme/core/labels.py BatchLabels, or a (B, n_labels) tensor already in the module's
        encoding) for the following windows; conformed to the module's LabelEncoding as Module.__call__ does."""
        if labels is None:
            self._labels = None
            return
        enc = getattr(self.stepper._step_obj.module, "_label_encoding", None)
        if hasattr(labels, "conform_to_encoding"):
            if enc is None:
                raise TypeError("Labels are not allowed for unconditional models")
            labels = labels.conform_to_encoding(enc).tensor
        self._labels = labels.to(device=self.device, dtype=torch.float32).contiguous()
        if self._labels.shape[0] != self.B:
            raise ValueError(f"labels for {self._labels.shape[0]} samples, engine batch is {self.B}")

    def load(self, initial_condition: Mapping[str, torch.Tensor], forcing: Mapping[str, torch.Tensor]):
        for n in self.prognostic:
            self.ic[n].copy_(initial_condition[n].reshape(self.B, 1, self.H, self.W))
        for n in self.forcing_names:
            self.forcing[n].copy_(forcing[n][:, : self.T + 1])
        for n in self.target_names:
            self.target[n].copy_(forcing[n][:, : self.T + 1])
        self._corrector_state = None        # a new initial condition re-seeds the corrector
        self._mc_state = None
        if self._mc is not None:
            for k, (_, factor) in enumerate(self._mc_factors):
                torch.mul(self.forcing[self._mc.forcing_name], factor, out=self.forcing_mc[k])
        if self._have_ref is not None:
            self._have_ref.fill_(False)
        if self._physics is not None and self._physics.tracks_dry_air:
            self._physics.reset(_lib.current_stream())

    def set_random_state(self, random_state) -> None:
        """``ace_amd.rand.RandomState`` (or None: the global RNG) the conditioning noise of the following windows is drawn from."""
        self._random_state = random_state

    def _draw_window_conditioning(self):
        """graph="window": the conditioning fields of the T steps, drawn now (through fme.core.rand's contract: the active CPU
        generator of a seeded rollout, else the device RNG) in step order and copied into the static buffer the captured steps read"""
        for s in range(self.T):
            field = self.net.conditioning_field(self.B, self.device, labels=self._labels)
            if self._cond_window is None:
                nbytes = 4 * self.T * field.numel()
                if nbytes > self.max_window_conditioning_bytes:
                    raise RuntimeError(f"graph='window' would hold {nbytes / 2**30:.1f} GiB of conditioning fields ({self.T} steps x "
                                       f"{tuple(field.shape)} fp32) in device memory; use graph='step' (the same captured step, one "
                                       f"draw per replay) or raise RolloutEngine.max_window_conditioning_bytes")
                self._cond_window = torch.empty(self.T, *field.shape, dtype=torch.float32, device=self.device)
            self._cond_window[s].copy_(field)

    def run_window(self):
        """Enqueue the T steps of the window on the current stream (no host synchronisation)."""
        from .rand import use_generator
        with use_generator(None if self._random_state is None else self._random_state.generator):
            if self._conditioned and self.graph_mode == "window":
                self._draw_window_conditioning()
            self._run_window()

    def _run_window(self):
        # parameters changed since the last window (load_state_dict / stepper.load_state): upload them; the library drops
        # its captured per-step graphs itself, the window graph captured here is dropped too
        self.net.sync_weights()
        # The captured forwards hold scalars derived from the weights by value (ace_sfno_set_weight): whoever uploaded last -
        # this call, net.forward, Stepper.predict, another engine on the same net - the library's generation counter tells.
        key = (int(self.net._native.value or 0) if hasattr(self.net._native, "value") else int(self.net._native),
               int(_lib.lib().ace_sfno_weights_generation(self.net._native)))
        if key != self._window_graph_key:
            self._window_graph = None
        step = self.stepper._step_obj   # Stepper.replace_ocean / overrides after construction take effect here
        if step._ocean is not self._ocean or step._corrector is not self._corrector:
            # The engine's forcing / target name sets and static buffers were laid out for the ocean it was built with: a
            # replacement that reads other fields (a slab ocean's q_flux / mixed-layer depth, an SST the old configuration did not
            # prescribe) or that the fused kernels do not know cannot be swapped in - ask for a new engine instead of failing
            # mid-run on a missing buffer.
            new_ocean = step._ocean
            needed = set(getattr(new_ocean, "forcing_names", None) or [])      # ace_amd.ocean.Ocean.forcing_names, however it was swapped in
            missing = sorted(needed - set(self.forcing_names) - set(self.target_names))
            if missing or (self._physics is not None and new_ocean is not None and getattr(new_ocean, "is_slab", False)):
                raise RuntimeError("the stepper's ocean was replaced after this RolloutEngine was built and the new one needs fields / "
                                   f"a path the engine's static buffers were not laid out for ({missing or 'slab ocean with fused physics'}): "
                                   "build a new RolloutEngine for the modified stepper")
            self._ocean, self._corrector = step._ocean, step._corrector
            self._window_graph = None
            if self._physics is not None:
                self._physics = self._build_physics()
        if self.graph_mode == "window":
            if self._window_graph is None:
                # warm up outside capture (first-touch allocations inside torch), then capture once
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._enqueue_step(0, False)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for s in range(self.T):
                        self._enqueue_step(s, False)
                self._window_graph = g
                self._window_graph_key = key
            self._window_graph.replay()
            if self._have_ref is not None and self._physics is None:
                # host-side view of the hook state the captured region keeps in its static buffer (a replay does not run
                # the Python that set it at capture time); a copy, so that a state handed out survives the next load()
                from .corrector import CorrectorState
                self._corrector_state = CorrectorState(global_dry_air_mass=self._ref_mass.clone())
        else:
            for s in range(self.T):
                self._enqueue_step(s, self.graph_mode == "step")

    def predict(self, initial_condition, forcing, time=None) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        """Same contract as Stepper.predict for one window of n_forward_steps (a ``PrognosticState`` initial condition
        carries the corrector state of the previous window in; the returned state carries this window's out).  ``time``: the
        window's TimeAxis when the stepper derives forcings from it (or `forcing` is a ForcingWindow)."""
        from .step import StepperState
        from .stepper import PrognosticState
        deriver = self.stepper.forcing_deriver
        # ALWAYS computed - a data field of the same name in the forcing record is overwritten, as Insolation.compute and
        # Stepper.predict do (fme/core/step/derived_forcings: the derived field wins) - unless this very window has been through
        # the deriver already (EnginePredict derives in front of the engine and marks the window)
        if deriver.needs_time and not getattr(forcing, "derived", False):
            forcing = deriver(forcing, time, device=self.device)      # once per window, in front of the captured steps
        with torch.no_grad():
            self.load(initial_condition, forcing)
            carried = getattr(initial_condition, "stepper_state", None)
            carried_random = getattr(carried, "random_state", None)
            if carried_random is not None:      # a state that carries only the corrector's part keeps set_random_state()'s generator
                self._random_state = carried_random
            if carried is not None:
                self._mc_state = carried.corrector_state
                self._corrector_state = carried.corrector_state
                cs = carried.corrector_state
                if self._have_ref is not None and cs is not None and cs.global_dry_air_mass is not None:
                    self._ref_mass.copy_(cs.global_dry_air_mass)     # carried reference: the captured region keeps it
                    self._have_ref.fill_(True)
                if (self._physics is not None and self._physics.tracks_dry_air and cs is not None
                        and cs.global_dry_air_mass is not None):
                    self._physics.set_reference(cs.global_dry_air_mass, _lib.current_stream())
            self.run_window()
            if self._physics is not None and self._physics.tracks_dry_air:
                from .corrector import CorrectorState
                mass = self._physics.get_reference(_lib.current_stream())   # (synchronises: once per window, with the result)
                self._corrector_state = CorrectorState(global_dry_air_mass=mass) if mass is not None else None
        state = PrognosticState({n: self.out[n][:, -1:] for n in self.prognostic})
        if self._corrector_state is not None or self._random_state is not None:
            state.stepper_state = StepperState(corrector_state=self._corrector_state, random_state=self._random_state)
        return self.out, state

    def continue_from_last(self):
        """Carry the final prognostic state into the initial-condition slot (next window of a long rollout)."""
        for n in self.prognostic:
            self.ic[n].copy_(self.out[n][:, -1:])
