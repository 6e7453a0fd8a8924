"""MI355X execution engine for one DeeR-VLA control step.

Owns the device-resident weights (bf16, the LLM ones pre-packed into MFMA-fragment order), the static
workspace, the device-side exit control block, and the order in which the HIP kernels of
``libdeer_hip.so`` are enqueued.  A whole control step (2x ViT-L/14 -> 2x Perceiver -> MPT layers with gated
x-attn -> action-head evaluations at the exit layers) is enqueued WITHOUT any host synchronisation: the exit
criterion is evaluated on the device and later kernels return at entry once it fired, so the step can be
captured once into a HIP graph and replayed (``DeerEngine.step``); the host reads {exit_layer, action} once.

Reference path being replaced: ``ModelWrapper.step`` -> ``MPTFlamingo.forward``
(robot_flamingo/eval/eval_utils.py:279-480, robot_flamingo/models/flamingo_mpt.py:308-461).

PyTorch is plumbing here (device allocations, streams, graph capture, H2D/D2H copies); every FLOP of the step
runs in this repo's hand-written gfx950 kernels.
"""
from __future__ import annotations

import ctypes
import os
import time
from contextlib import contextmanager
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _abi as abi

import threading


class _StepGate:
    """Several env batches per GPU are stepped from several host threads (rollout.evaluate_policy_batched(groups=...)).  Steps of
    different engines run side by side (shared hold); a HIP-graph CAPTURE waits until no other thread is inside a step and keeps
    new steps out while it lasts (exclusive hold) - captures are rare (first use of an instruction length) and short."""

    def __init__(self):
        self.cv = threading.Condition()
        self.readers, self.writer = 0, False
        self.local = threading.local()

    @contextmanager
    def step(self):
        if getattr(self.local, "depth", 0):                      # nested (a step that calls another engine method): already inside
            self.local.depth += 1
            try:
                yield
            finally:
                self.local.depth -= 1
            return
        with self.cv:
            while self.writer:
                self.cv.wait()
            self.readers += 1
        self.local.depth = 1
        try:
            yield
        finally:
            self.local.depth = 0
            with self.cv:
                self.readers -= 1
                self.cv.notify_all()

    @contextmanager
    def exclusive(self):
        inside = bool(getattr(self.local, "depth", 0))
        with self.cv:
            if inside:
                self.readers -= 1                                # give up this thread's own shared hold while it waits
            while self.writer or self.readers > 0:
                self.cv.wait()
            self.writer = True
        try:
            yield
        finally:
            with self.cv:
                self.writer = False
                if inside:
                    self.readers += 1
                self.cv.notify_all()


_GATE = _StepGate()


@contextmanager
def _capture(graph):
    """HIP-graph capture while no other host thread is inside an engine step (see _StepGate); thread-local error mode so that other
    threads' allocations / copies outside a step do not invalidate it."""
    with _GATE.exclusive():
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            yield
from ._abi import config_to_c  # noqa: F401  (re-exported: tests / tools import it from here)
from .config import DeerConfig
from .synthetic import mlp_layer_indices

EPS = 1e-5


def _i32(v: int) -> int:
    """two's-complement wrap into int32 (halves of a pointer stored in an int32 tensor)"""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def _cur_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeerEngine:
    """Python host of the native spine (csrc/model.hip, include/deer_model.h): owns the two device allocations (weight arena,
    workspace) and the pinned host buffers, ingests the reference state dict by name, and feeds the step to the GPU as HIP-graph
    pieces.  The kernel ORDER of every piece lives in the C++ model object; nothing here touches a kernel directly."""
    LOOKAHEAD = 1        # trunk layers the host keeps in flight beyond an undecided exit check

    def __init__(self, cfg: DeerConfig, state_dict: Optional[Dict[str, torch.Tensor]], device="cuda", max_text_len: int = 32,
                 n_envs: int = 1, threshold_type: str = "L2", leq: bool = True, segmented: bool = True, precision: Optional[str] = None,
                 weights_from: Optional["DeerEngine"] = None):
        """n_envs: independent environments evaluated per control step (one "env batch" per rank).  They share every
        weight read: the ViT sees M = 514*n_envs rows, the LLM n_envs*T rows, each environment keeps its own LSTM state,
        thresholds are shared and every environment exits at its own layer (device side)."""
        if not torch.cuda.is_available():
            raise abi.DeerHipError("DeerEngine needs a HIP device (no CPU fallback exists in deer_vla_amd)")
        self.lib = abi.lib()
        self.cfg = cfg
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        assert 1 <= n_envs <= abi.MAX_ENVS
        self.B = n_envs
        self.n_cams = 2 * n_envs                       # images per step: (rgb, gripper) of every environment
        # trunk rows n_envs * T: up to 512 in the bf16 arithmetic (16 environments x the reference's max_length = 32, data.py:905-919; the
        # hi/lo-plane trunk GEMM runs them in blocks of 128), 128 in the fp32 arithmetic (one launch of deer_gemm_skinny)
        # precision (include/deer_model.h, _abi.PRECISIONS): "fp16" (default) / "bf16" = the product arithmetic on fp16 / bf16 operands,
        # "fp32" = the parity arithmetic.  Engines over one weight arena share it (the weights are stored in that format).
        if precision is None and weights_from is not None:
            precision = weights_from.precision
        precision = abi.resolve_precision(precision)
        if weights_from is not None and precision != weights_from.precision:
            raise ValueError("engines over one weight arena share the precision (the weights are stored in its format)")
        self.MAX_ROWS = abi.max_trunk_rows(cfg, precision)
        self.max_T = min(max_text_len, self.MAX_ROWS // n_envs)
        assert self.max_T >= 1, "n_envs * T must fit the trunk's LLM rows"      # load_inputs checks every instruction against max_T
        self._thr_type = abi.THR_TYPES[threshold_type]
        self._leq = 1 if leq else 0
        self._h = ctypes.c_void_p()
        # "fp32": fp32 activations everywhere (csrc/precise.hip) - the parity arithmetic of north_star's 1e-3 clause; single-stream
        # schedule, one graph per step.  "fp16" / "bf16": the product arithmetic (graph pieces, two vision chains, env batches to 16).
        self.precision = precision
        self.product = precision != "fp32"
        cc = config_to_c(cfg, n_envs, self.max_T, precision=precision)
        abi.check(self.lib.deer_model_create(ctypes.byref(cc), ctypes.byref(self._h)), "deer_model_create")
        with torch.cuda.device(self.dev):
            self.workspace = torch.zeros(self.lib.deer_model_workspace_bytes(self._h), dtype=torch.uint8, device=self.dev)
            if weights_from is None:
                self.arena = torch.zeros(self.lib.deer_model_arena_bytes(self._h), dtype=torch.uint8, device=self.dev)
                abi.check(self.lib.deer_model_bind(self._h, abi.ptr(self.arena), abi.ptr(self.workspace)), "deer_model_bind")
                self._load_weights(state_dict)
            else:                                      # ONE copy of the weights serves every engine built on the same config
                self.arena = weights_from.arena
                abi.check(self.lib.deer_model_bind(self._h, abi.ptr(self.arena), abi.ptr(self.workspace)), "deer_model_bind")
                abi.check(self.lib.deer_model_share_weights(self._h, weights_from._h), "deer_model_share_weights")
                self._weights_owner = weights_from     # keeps the owner (and its arena) alive
            self._make_views()
        self._siblings: Dict[int, "DeerEngine"] = {}
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self.segmented = segmented and self.product         # dynamic steps fed in per-layer graph pieces (see _step_segmented)
        # DEER_ONE_GRAPH=1: dynamic steps as ONE graph with branches - no host on the decision path at all, but every kernel behind
        # the exit still launches and returns at entry (see _step_one_graph; measured slower: 242 vs 352 steps/s at one
        # environment, 778 vs 862 at eight)
        self._one_graph = os.environ.get("DEER_ONE_GRAPH") == "1"
        self._side_stream = torch.cuda.Stream(device=self.dev)
        self._drain_events: Dict[torch.cuda.Stream, torch.cuda.Event] = {}
        self._plans = []                                                   # native step plans (own HIP events): freed with the graphs
        self._native_step = os.environ.get("DEER_NATIVE_STEP", "1") == "1"   # 0: the Python submission / polling loop (debugging)
        self._extra_streams: List[torch.cuda.Stream] = []
        self._use_side = os.environ.get("DEER_SIDE", "1") == "1"          # debugging knobs (README)
        self._lookahead = int(os.environ.get("DEER_LOOKAHEAD", self.LOOKAHEAD))
        self._seq = 0
        self._ids_tag = None
        self._trace = None                            # debugging aid: list of (label, event, host time) per piece
        self._time_stages = False                     # per-stage GPU times of pipelined steps -> last_stage_ms (eval_time hook)
        self.last_stage_ms: Dict[str, float] = {}
        # controller configuration (set by configure_exit)
        self.exit_ids = cfg.exit_ids()
        self.ctl_max_layer = self.exit_ids[-1]
        self._max_layer_arg = self.exit_ids[-1] + 1
        self.steps_per_stage = 1
        self._apply_controller()
        self.reset()

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                torch.cuda.synchronize(self.dev)
                for pl in getattr(self, "_plans", []):
                    self.lib.deer_step_plan_destroy(pl)
                self._plans = []
                self.lib.deer_model_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:                              # interpreter shutdown
            pass

    # ------------------------------------------------------------------------------------------ weights
    def _load_weights(self, sd: Dict[str, torch.Tensor]):
        """Every tensor of the reference state dict the model knows is uploaded and handed to ``deer_model_load_tensor``,
        which converts / re-lays it out on the device (bf16, MFMA-fragment packing, stacking: csrc/model.hip)."""
        st = _cur_stream()
        for name, t in sd.items():
            if not self.lib.deer_model_knows_tensor(self._h, name.encode()):
                continue
            t = t.detach()
            is_bf = t.dtype == torch.bfloat16
            src = t.to(device=self.dev, dtype=torch.bfloat16 if is_bf else torch.float32).contiguous()
            abi.check(self.lib.deer_model_load_tensor(self._h, name.encode(), abi.ptr(src), 1 if is_bf else 0, src.numel(), st),
                      f"deer_model_load_tensor({name}, shape {tuple(t.shape)})")
            torch.cuda.current_stream().synchronize()            # src may be freed once the conversion ran
        buf = ctypes.create_string_buffer(4096)
        n = self.lib.deer_model_missing_tensors(self._h, buf, len(buf))
        if n:
            raise abi.DeerHipError(f"{n} required parameters missing from the state dict, e.g. {buf.value.decode().split()[:4]}")

    def _buf(self, name: str, which: int = 1):
        off, nbytes = ctypes.c_long(), ctypes.c_long()
        abi.check(self.lib.deer_model_buffer(self._h, which, name.encode(), ctypes.byref(off), ctypes.byref(nbytes)), f"buffer {name}")
        base = self.workspace if which == 1 else self.arena
        return base[off.value: off.value + nbytes.value]

    def _make_views(self):
        cfg, B = self.cfg, self.B
        N, S, W, nl, d = self.n_cams, cfg.image_size, cfg.vit_width, cfg.perc_latents, cfg.d_model
        rows = min(B * self.max_T, self.MAX_ROWS)
        self.max_rows = rows
        v = lambda name, dt: self._buf(name).view(dt)
        # camera frames / media tokens / K|V / wte in the arithmetic's 16-bit format (fp16 keeps every 8-bit pixel level apart after CLIP
        # normalisation; bf16 merges some); f32 for the parity arithmetic
        self.img_dtype = abi.torch_dtype16(self.precision)
        self.img = v("img", self.img_dtype).view(N, 3, S, S)               # static input buffer (camera frames)
        self.vx = v("vx", torch.float32).view(N, cfg.n_patches + 1, W)     # ViT residual stream (fp32)
        self.media_dtype = torch.float16 if self.precision == "fp16" else torch.bfloat16
        # media tokens [rgb latents ; gripper latents] per env (post fusion; the buffers are sized for it); one set of latents per env with
        # fusion_mode="pre": the first B * n_media rows
        self.vis_x = v("vis_x", self.media_dtype).view(N * nl, W)[:B * cfg.n_media]
        self.vis_x_f32 = v("vis_x_f32", torch.float32).view(N * nl, W)[:B * cfg.n_media]
        self.kv_all = v("kv_all", self.media_dtype)
        self.ids = v("ids", torch.int64)
        self.key_mask = v("key_mask", torch.uint8)
        self.key_mask.fill_(1)
        self.text_time = v("text_time", torch.int32)
        self.x = v("x", torch.float32).view(rows, d)
        self.hidden = v("hidden", torch.float32).view(cfg.n_layers, rows, d)   # hidden_states[i] = output of layer i
        Lh, H = cfg.lstm_num_layers, cfg.head_hidden
        for n in ("h_state", "c_state", "h_tmp", "c_tmp", "h_shadow", "c_shadow"):   # LSTM state: [layer][env][H]
            setattr(self, n, v(n, torch.float32).view(Lh, B, H))
        self.pooled = v("pooled", torch.float32).view(B, d)
        if getattr(cfg, "use_state", False):              # robot state of the step: (arm pose robot_obs[:6], gripper opening robot_obs[-1], pad)
            self.state_in = v("state_in", torch.float32).view(B, 8)
        self.ctl = v("ctl", torch.int32)                                   # one control block per environment
        self.hold_dev = v("step_info", torch.int32)                        # step_info: {hold, seq, host mirror ptr lo, hi}
        self.thresholds = v("thresholds", torch.float32)
        self.thresholds.fill_(1e8)
        # outputs of the last head evaluation: [pose 6 A | gripper prob A | gripper logit A] per environment (A = multi_step_action)
        self.A = int(getattr(cfg, "multi_step_action", 1))
        self.action_dbg = v("action_dbg", torch.float32).view(B, 64)[:, :8 * self.A]
        self.act_ext = v("act_ext", torch.float32).view(B, 4, 64)         # multi_step_action > 1: previous | committed | ensemble action
        self.act_ext_host = torch.zeros(B, 4, 64, dtype=torch.float32).pin_memory()
        if getattr(cfg, "layerwise_exit_eval", False):                     # per-layer heads' LSTM state: [head][h, c][L][B][H]
            n_lw = len(cfg.layerwise_heads())
            self.lw_state = v("lw_state", torch.float32).view(n_lw, 2, Lh, B, H)
        self.wte = self._buf("lang_encoder.transformer.wte.weight", 0).view(torch.float32 if self.precision == "fp32" else self.media_dtype).view(cfg.vocab_size, d)
        self.ctl_host = torch.zeros(B * abi.CTL_WORDS, dtype=torch.int32).pin_memory()
        self._ctl_host_np = self.ctl_host.numpy()
        self.step_info_host = torch.zeros(8, 4, dtype=torch.int32).pin_memory()   # ring: an async upload may still be pending
        # host mirror of the verdicts (pinned => device-visible and system-coherent): see csrc/head.hip::check_done
        # (multi_step_action > 1: + [B][128] f32 behind the control blocks - the committed and the ensemble action of every environment)
        self.host_mirror = torch.zeros((1 + B) * abi.CTL_WORDS + (B * 128 if self.A > 1 else 0), dtype=torch.int32).pin_memory()
        self._hm = self.host_mirror.numpy()                                 # polled by the host between graph segments
        self.step_info_pinned = torch.zeros(4, dtype=torch.int32).pin_memory()   # read by the device in pipelined steps
        self._si_np = self.step_info_pinned.numpy()
        mptr = self.host_mirror.data_ptr()
        self._si_np[2], self._si_np[3] = _i32(mptr & 0xFFFFFFFF), _i32(mptr >> 32)

    # ----------------------------------------------------------------------------- pieces (thin calls into the spine)
    def enqueue_vision(self, part: str = "all", chain: int = -1, kv_all: bool = True):
        """ViT-L/14 on the camera frames (batched; the reference runs them separately: flamingo_mpt.py:626,633), Perceiver on
        each, concat -> vis_x, then K/V of every x-attn layer.  part: "all" | "head" | "tail"; chain >= 0: one chain of the
        multi-stream schedule."""
        abi.check(self.lib.deer_vision(self._h, chain, {"all": 0, "head": 1, "tail": 2}[part], 1 if kv_all else 0, _cur_stream()),
                  "deer_vision")

    def _enqueue_media_kv(self):
        abi.check(self.lib.deer_media_kv(self._h, _cur_stream()), "deer_media_kv")

    def enqueue_embed(self, T):
        abi.check(self.lib.deer_llm_embed(self._h, T, _cur_stream()), "deer_llm_embed")

    def enqueue_llm_layer(self, i, T, pending, use_mask: bool, finalize: bool, ctl=True):
        """FlamingoLayer.forward (flamingo_lm.py:46-83) on n_envs*T rows.  ``pending``: truthy when the previous layer left
        its last residual branch un-applied (it was not finalized); returns the same marker for this layer."""
        abi.check(self.lib.deer_llm_layer(self._h, i, T, 1 if use_mask else 0, 1 if pending else 0, 1 if finalize else 0,
                                          1 if ctl else 0, _cur_stream()), "deer_llm_layer")
        return None if finalize else True

    def enqueue_head(self, layer: int, T: int, kind: int, slot: int = -1, force: bool = False, use_ctl: bool = True,
                     feats: Optional[torch.Tensor] = None, shadow: bool = False, no_ctl_final: bool = False,
                     use_mask: bool = False):
        """One DeterministicDecoder evaluation on hidden_states[layer] (action_head.py:499-611) followed by the exit gate
        (value_net.py:120-133,277-297).  kind: PSEUDO / CHECK / COMMIT (include/deer_hip.h).  With feats=None inside a control step the
        LSTM's recurrent half comes from deer_begin_step's pre-pass; after a host write to h_state (_head_state_changed) the fused kernel
        reads h_state itself."""
        abi.check(self.lib.deer_head_eval(self._h, layer, T, kind, slot, 1 if force else 0, 1 if use_ctl else 0, 1 if shadow else 0,
                                          1 if no_ctl_final else 0, abi.ptr(feats), 1 if use_mask else 0, _cur_stream()),
                  "deer_head_eval")

    # ------------------------------------------------------------------------------------- step assembly
    def _apply_controller(self):
        ids = (ctypes.c_int * len(self.exit_ids))(*self.exit_ids)
        rc = self.lib.deer_model_configure_exit(self._h, ids, len(self.exit_ids), self._max_layer_arg, self._thr_type, self._leq)
        if rc != 0:
            raise ValueError(f"invalid exit configuration: exit ids {self.exit_ids}, max_layer {self._max_layer_arg} -> the deepest "
                             f"reachable layer {min(self._max_layer_arg - 1, self.exit_ids[-1])} must be one of the exits (otherwise no "
                             "exit check is ever forced; the reference raises KeyError on thresholds[i] there)")

    def set_compaction(self, on: bool):
        """Env batches: compaction of the rows of exited environments in the trunk (csrc/model.hip; on by default, DEER_COMPACT=0 turns it
        off at construction).  Per-environment results are bit-identical either way (tests/test_batch_parity.py)."""
        abi.check(self.lib.deer_model_set_compaction(self._h, 1 if on else 0), "deer_model_set_compaction")
        if self._graphs:
            torch.cuda.synchronize(self.dev)
        self._drop_graphs()

    def set_head_fused(self, on: bool):
        """One-environment control steps: every head evaluation (pseudo action / exit check) as ONE launch (csrc/head.hip:
        head_fused_kernel) instead of the eight separate kernels (default).  Same arithmetic per row (tests/test_engine_parity.py);
        measured SLOWER (65-71 us against 51 us per evaluation: every hand-off between resident workgroups costs 6-8 us, a kernel
        boundary 1.4 us + ramp) - an experiment like the persistent trunk layer, kept switchable (DEER_HEAD_FUSED=1)."""
        abi.check(self.lib.deer_model_set_head_fused(self._h, 1 if on else 0), "deer_model_set_head_fused")
        if self._graphs:
            torch.cuda.synchronize(self.dev)
        self._drop_graphs()

    def head_fused_error(self) -> int:
        """1 if a hand-off inside a one-launch head evaluation timed out (the step then ends without a verdict)"""
        return int(self._buf("head_fused_err").view(torch.int32)[0].item())

    def set_persistent_layer(self, on: bool):
        """N1 experiment (csrc/persistent_layer.hip): one persistent launch per trunk layer of a one-environment step.  Bit-identical to
        the twelve-launch layer and slower; needs ALL of its 256 workgroups resident, so only one engine per GPU may use it at a time."""
        if on and (self._siblings or self.B > 1):
            # the persistent launch needs ALL of its 256 workgroups resident at once and ignores ALL_EXITED: engines that run beside it on
            # the same GPU (sibling engines of a window / env batches in flight) cannot promise that
            raise abi.DeerHipError("set_persistent_layer(True): a stand-alone one-environment engine only (no sibling engines, no env batch)")
        abi.check(self.lib.deer_model_set_persistent_layer(self._h, 1 if on else 0), "deer_model_set_persistent_layer")
        self._persistent_on = bool(on)
        torch.cuda.synchronize(self.dev)
        self._drop_graphs()
        self._buf("pl_state").view(torch.int32)[:16 * 12 + 16].zero_()   # barrier counters (cumulative over launches) + error words

    def persistent_layer_error(self) -> int:
        """barrier number that timed out in a persistent layer launch (0 = none)"""
        return int(self._buf("pl_state").view(torch.int32)[16 * 12].item())

    def persistent_layer_trace(self, clear: bool = False):
        """int32 [64 layers][16 barriers][clock lo, clock hi (100 MHz), last workgroup to arrive, timeouts]; row 0 = entry of workgroup 0"""
        t = self._buf("pl_state").view(torch.int32)[16 * 12 + 16:16 * 12 + 16 + 64 * 64].view(64, 16, 4)
        out = t.cpu()
        if clear:
            t.zero_()
        return out

    def persistent_layer_error_detail(self):
        """[epoch, workgroup, generation word, top counter, 8 group counters] recorded by the first barrier that timed out"""
        return [int(v) for v in self._buf("pl_state").view(torch.int32)[16 * 12:16 * 12 + 12].tolist()]

    def configure_exit(self, exit_ids: Sequence[int], max_layer: int, steps_per_stage: int = 1):
        """``ExitController.__init__`` (value_net.py:164-173): max_layer = min(max_layer-1, last exit)."""
        old = (self.exit_ids, self._max_layer_arg)
        self.exit_ids = list(exit_ids)
        self._max_layer_arg = max_layer
        try:
            self._apply_controller()
        except ValueError:
            self.exit_ids, self._max_layer_arg = old
            raise
        self.ctl_max_layer = min(max_layer - 1, self.exit_ids[-1])
        self.steps_per_stage = steps_per_stage
        if self._graphs:
            torch.cuda.synchronize(self.dev)                      # pieces of the last step may still be replaying
        self._drop_graphs()

    @property
    def thr_type(self) -> int:
        return self._thr_type

    @thr_type.setter
    def thr_type(self, v: int):
        if v != self._thr_type:
            self._thr_type = int(v)
            self._apply_controller()
            self._drop_graphs()

    @property
    def leq(self) -> int:
        return self._leq

    @leq.setter
    def leq(self, v):
        if int(bool(v)) != self._leq:
            self._leq = int(bool(v))
            self._apply_controller()
            self._drop_graphs()

    @property
    def real_num_exit(self) -> int:
        return len([x for x in self.exit_ids if x <= self.ctl_max_layer])

    def set_thresholds(self, thresholds: Sequence[float]):
        """``ExitController._set_threshold_value`` (value_net.py:178-183)."""
        assert len(thresholds) == self.real_num_exit, (len(thresholds), self.real_num_exit)
        t = torch.full((16,), 1e8, dtype=torch.float32)
        t[: len(thresholds)] = torch.tensor([float(v) for v in thresholds], dtype=torch.float32)
        self.thresholds.copy_(t)

    def set_thresholds_device(self, row: torch.Tensor):
        """The same from a DEVICE row of 16 floats (unused slots 1e8): an asynchronous device-to-device copy on the caller's stream, for
        callers that change thresholds between steps of a timed loop (bench.py's scripted exit schedule)."""
        assert row.is_cuda and row.dtype == torch.float32 and row.numel() == 16
        self._drain_side_streams()                               # a speculative exit check of the last step may still read them
        self.thresholds.copy_(row, non_blocking=True)

    def reset(self):
        """Episode start: ``clear_all_exit_memory`` + controller state (eval_utils.py:252-277)."""
        # a speculative head evaluation of the last step may still be queued on the side stream; it returns at entry only
        # while that step's ALL_EXITED is set, so the control blocks / LSTM state must not be cleared underneath it
        self._drain_side_streams()
        self.h_state.zero_()
        self.c_state.zero_()
        if hasattr(self, "lw_state"):                              # clear_all_exit_memory clears EVERY head (flamingo_mpt.py:287-301)
            self.lw_state.zero_()
        self._head_state_changed()
        self.ctl.zero_()
        self._shadow_on = False
        self.cur_step = 0

    def reset_env(self, b: int):
        """Episode start of ONE environment of the batch (its sub-task changed): clears its LSTM state and control block; the
        other environments carry on."""
        self._drain_side_streams()
        self.h_state[:, b].zero_()
        self.c_state[:, b].zero_()
        if hasattr(self, "lw_state"):
            self.lw_state[:, :, :, b].zero_()
        self._head_state_changed()
        W = abi.CTL_WORDS
        keep = self.ctl[:W].clone() if b == 0 else None            # block 0 also holds the batch-global words (shadow flag, host ptr)
        self.ctl[b * W:(b + 1) * W].zero_()
        if keep is not None:
            for k in (abi.CTL_SHADOW, abi.CTL_HOST_PTR, abi.CTL_HOST_PTR + 1):
                self.ctl[k] = keep[k]

    def _head_state_changed(self):
        """The host wrote h_state: the recurrent half the last deer_begin_step computed from it (W_hh h + b_hh, read by the head
        evaluations of a control step) is stale until the next step begins; piece calls in between (enqueue_head with feats=None) fall
        back to the fused LSTM kernel (include/deer_model.h)."""
        self.lib.deer_model_head_state_changed(self._h)

    def dynamic_plan(self):
        """Per layer of the dynamic step: (layer, need_pseudo, is_exit, exit slot).  mosaic_gpt_3b.py:397-443; computed by
        the spine (deer_dynamic_plan)."""
        key = (tuple(self.exit_ids), self._max_layer_arg)
        if getattr(self, "_plan_key", None) != key:               # on the per-step path: ask the spine once per configuration
            n = self.cfg.n_layers
            a, b, c = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
            k = self.lib.deer_dynamic_plan(self._h, a, b, c, n)
            self._plan = [(i, bool(a[i]), bool(b[i]), int(c[i])) for i in range(k)]
            self._plan_key = key
        return self._plan

    def enqueue_dynamic_main(self, T, use_mask, i):
        """Trunk part of layer i of the dynamic step (layer 0 also embeds the tokens)."""
        plan = self.dynamic_plan()
        _, need_pseudo, is_exit, _ = plan[i]
        if i == 0:
            self.enqueue_embed(T)
        prev_open = i > 0 and not (plan[i - 1][1] or plan[i - 1][2])          # previous layer not finalized -> residual pending
        self.enqueue_llm_layer(i, T, prev_open, use_mask, finalize=(need_pseudo or is_exit))

    def enqueue_dynamic_heads(self, T, i, shadow: bool = False, use_mask: bool = False):
        """Head evaluations that read hidden_states[i]: the layer-0 pseudo action and/or the exit check."""
        _, need_pseudo, is_exit, slot = self.dynamic_plan()[i]
        pm = use_mask and self.B > 1          # rows of an env batch that are right-padding stay out of the token pool
        if need_pseudo:
            self.enqueue_head(i, T, abi.KIND_PSEUDO, use_mask=pm)
        if is_exit:
            self.enqueue_head(i, T, abi.KIND_CHECK, slot=slot, force=(i >= self.ctl_max_layer), shadow=shadow, use_mask=pm)

    def enqueue_llm_static(self, T, use_mask, exit_id):
        """exit_id given (flamingo_mpt.py:402-411,446-461): run layers 0..exit_id, committing head call."""
        if getattr(self.cfg, "use_state", False):
            abi.check(self.lib.deer_head_state(self._h, _cur_stream()), "deer_head_state")
        self.enqueue_embed(T)
        for i in range(exit_id + 1):
            self.enqueue_llm_layer(i, T, None, use_mask, finalize=True, ctl=False)
        self.enqueue_head(exit_id, T, abi.KIND_COMMIT, use_ctl=False, use_mask=use_mask and self.B > 1)

    def _drain_side_streams(self):
        # an unconditional stream-wait (event record + wait: ~6 us of host time, nothing on the device when the stream is idle);
        # `if not st.query()` in front of it cost 27 us per call on this stack (hipStreamQuery) - 53 us of every step during which
        # the GPU waited for the host (tools/host_gap.py, round 5)
        cur = torch.cuda.current_stream()
        evs = self._drain_events
        for st in (self._side_stream, *self._extra_streams):
            ev = evs.get(st)
            if ev is None:
                ev = evs[st] = torch.cuda.Event()                 # one event per side stream, re-recorded every step
            ev.record(st)
            cur.wait_event(ev)

    def _chain_stream(self, c: int):
        """stream of vision chain c >= 2 (chain 0: caller's stream, chain 1: the side stream)"""
        while len(self._extra_streams) < c - 1:
            self._extra_streams.append(torch.cuda.Stream(device=self.dev))
        return self._extra_streams[c - 2]

    @property
    def n_chains(self) -> int:
        return self.lib.deer_model_n_chains(self._h)

    def _enqueue_chain(self, c: int, part: str):
        """Vision tower of chain c of the two-stream schedule; chain 0's first piece also resets the control blocks (from
        the pinned step info)."""
        if c == 0 and part == "head":
            abi.check(self.lib.deer_begin_step(self._h, abi.ptr(self.step_info_pinned), _cur_stream()), "deer_begin_step")
        self.enqueue_vision(part, chain=c, kv_all=False)

    def _enqueue_step(self, T, use_mask, exit_id, shadow: bool = False):
        """the whole control step on the current stream (vision batched): one call into the spine"""
        abi.check(self.lib.deer_step_enqueue(self._h, T, 1 if use_mask else 0, -1 if exit_id is None else exit_id, 1 if shadow else 0,
                                             None, _cur_stream()), "deer_step_enqueue")

    # ----------------------------------------------------------------------------- in-situ profiler (bench.py roofline pass)
    def prof_begin(self):
        self.lib.deer_prof_enable(self._h, 1)

    def prof_end(self):
        """-> list of (kernel class = C-ABI entry point, microseconds, algorithmic flops, algorithmic bytes) per launch"""
        torch.cuda.synchronize(self.dev)
        self.lib.deer_prof_enable(self._h, 0)
        out = []
        name = ctypes.create_string_buffer(64)
        us, fl, by = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
        for i in range(self.lib.deer_prof_count(self._h)):
            abi.check(self.lib.deer_prof_get(self._h, i, name, 64, ctypes.byref(us), ctypes.byref(fl), ctypes.byref(by)), "deer_prof_get")
            out.append((name.value.decode(), us.value, fl.value, by.value))
        return out

    # ---------------------------------------------------------------------------------------- host API
    def load_inputs(self, rgb: torch.Tensor, gripper: torch.Tensor, ids: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """rgb / gripper: (B, ..., 3, S, S) one frame per environment (any singleton dims in between; B may be omitted
        when n_envs == 1); ids (B, T) int64 (right-padded to a common T); mask (B, T) bool."""
        S, B = self.cfg.image_size, self.B
        iv = getattr(self, "_img_views", None)
        if iv is None:
            if getattr(self.cfg, "sep_resampler", False):     # camera-major frames: every vision chain holds ONE camera (own Perceiver weights)
                img = self.img.view(2, B, 3, S, S)
                iv = self._img_views = (img[0], img[1])
            else:
                img = self.img.view(B, 2, 3, S, S)
                iv = self._img_views = (img[:, 0], img[:, 1])
        iv[0].copy_(rgb.view(B, 3, S, S) if rgb.is_contiguous() else rgb.reshape(B, 3, S, S), non_blocking=True)
        iv[1].copy_(gripper.view(B, 3, S, S) if gripper.is_contiguous() else gripper.reshape(B, 3, S, S), non_blocking=True)
        ids = ids.reshape(B, -1)
        T = ids.shape[1]
        assert 0 < T <= self.max_T and B * T <= self.max_rows
        # the instruction only changes between sub-tasks: skip the upload when the caller hands over the same tensor again
        tag = (ids.data_ptr(), ids._version, B * T)
        if tag != self._ids_tag:
            self.ids[:B * T].copy_(ids.reshape(-1), non_blocking=True)
            self._ids_tag = tag
            self._ids_keep = ids                              # keeps data_ptr from being recycled under the tag
        use_mask = False
        if mask is not None:
            # like the instruction, the mask only changes between sub-tasks: the "any padding?" question costs a device round trip
            # (bool() of a device tensor), so it is asked once per mask tensor
            mtag = (mask.data_ptr(), mask._version, B * T)
            if mtag != getattr(self, "_mask_tag", None):
                m = mask.reshape(B, T).to(torch.uint8)
                self._mask_any_pad = bool((m == 0).any())
                self.key_mask[:B * T].copy_(m.reshape(-1), non_blocking=True)
                self._mask_tag, self._mask_keep = mtag, mask
            use_mask = self._mask_any_pad
        else:
            self._mask_tag = None
        return T, use_mask

    def step(self, rgb, gripper, ids, mask=None, exit_id: Optional[int] = None, use_graph: bool = True, sync: bool = True,
             shadow: bool = False, state: Optional[torch.Tensor] = None, env_steps: Optional[Sequence[int]] = None):
        """state: robot_obs of every environment, (B, ..., 15) (eval_utils.py:324-332) - read only by a ``use_state`` model, whose
        head embeds (robot_obs[:6], robot_obs[-1]) into the pooled feature (action_head.py:524-536).
        env_steps: the step index of every environment inside ITS sub-task (what the reference hands to
        ``ExitController.set_timestep`` per rollout, eval_utils.py:662-663): with ``steps_per_stage`` > 1 environment b re-uses its
        previous exit while env_steps[b] % steps_per_stage != 0 (value_net.py:285-286).  Default: the engine's own step counter for
        every environment."""
        self._env_steps = None if env_steps is None else [int(v) for v in env_steps]
        if self._env_steps is not None:
            assert len(self._env_steps) == self.B, (len(self._env_steps), self.B)
        if getattr(self.cfg, "use_state", False):
            if exit_id is None:
                raise NotImplementedError("use_state has no dynamic exit: the reference's ActionValueNet calls the head without a state "
                                          "tensor and raises TypeError (value_net.py:122-129); pass a static exit_id")
            if state is None:
                raise ValueError("a use_state model needs the robot state of the step (state=robot_obs, (B, 15))")
            st = state.reshape(self.B, -1).to(torch.float32)
            packed = torch.zeros(self.B, 8, dtype=torch.float32)
            packed[:, :6] = st[:, :6].cpu()
            packed[:, 6] = st[:, -1].cpu()
            self.state_in.copy_(packed, non_blocking=False)
        with _GATE.step():
            return self._step_impl(rgb, gripper, ids, mask, exit_id, use_graph, sync, shadow)

    def _step_impl(self, rgb, gripper, ids, mask, exit_id, use_graph, sync, shadow):
        """One control step of all n_envs environments.  Returns (when sync) dict(pose (6,), gripper prob, gripper_logit,
        exit_layer, deltas) for n_envs == 1, else a list of such dicts (one per environment).
        shadow=True (calibration): every exit is evaluated and its delta recorded, the LSTM state / action are
        committed at the first exit whose criterion fires, but the step never terminates early."""
        # speculative head evaluations of the previous step may still be draining on the side stream (they return at entry,
        # but only while that step's ALL_EXITED is still set): nothing of this step may start before they are gone
        self._drain_side_streams()
        T, use_mask = self.load_inputs(rgb, gripper, ids, mask)
        if bool(shadow) != getattr(self, "_shadow_on", False):
            self.ctl[abi.CTL_SHADOW] = 1 if shadow else 0
            self._shadow_on = bool(shadow)
        if exit_id is not None and exit_id < 0:
            exit_id += self.cfg.n_layers
        if exit_id is None and self.exit_ids[0] == 0:
            # exit_interval = 1 makes layer 0 an exit; the reference's ActionValueNet has no action to compare its action with and asserts
            # (value_net.py:119, tests/golden/deer_forward_int1.npz) - the engine refuses instead of inventing a criterion
            raise NotImplementedError("dynamic exit with layer 0 in the exit list: the first layer similarity is not implemented "
                                      "(the reference asserts i > 0, value_net.py:119); use exit ids >= 1 or a static exit_id")
        hold = 0                                                  # bit b: environment b is inside a stage (csrc/head.hip: CTL_HOLD per environment)
        if exit_id is None and self.steps_per_stage != 1:
            steps = getattr(self, "_env_steps", None) or [self.cur_step] * self.B
            for b, t in enumerate(steps):
                if t % self.steps_per_stage != 0:
                    hold |= 1 << b
        seg_mode = self.segmented and use_graph and sync and exit_id is None and not shadow
        self._seq = (self._seq + 1) & 0xFFFFFF
        if self._seq == 0:                                        # wrap: stale mirror words would compare as "newer"
            torch.cuda.current_stream().synchronize()
            self._hm[:2] = 0
            self._seq = 1
        static_pieces = self.segmented and use_graph and sync and exit_id is not None and not shadow
        if seg_mode or static_pieces:
            # the step info is read by ctl_begin_step straight from pinned host memory (no upload launch); the host only
            # rewrites it after the previous step's verdict, i.e. after that step's ctl_begin_step ran
            si = self._si_np
            si[0], si[1] = hold, self._seq
            self.cur_step += 1
            if seg_mode and self._one_graph:
                return self._step_one_graph(T, use_mask)
            return self._step_segmented(T, use_mask) if seg_mode else self._step_static_pieces(T, use_mask, exit_id)
        si = self.step_info_host[self._seq & 7]
        si[0], si[1], si[2], si[3] = hold, self._seq, 0, 0
        self.hold_dev.copy_(si, non_blocking=True)
        key = (T, use_mask, exit_id, bool(shadow))
        if use_graph:
            g = self._graphs.get(key)
            if g is None:
                self._enqueue_step(T, use_mask, exit_id, shadow)  # eager warm-up (sets kernel attributes) - a real step
                torch.cuda.current_stream().synchronize()
                g = torch.cuda.CUDAGraph()
                with _capture(g):                         # capture does not execute kernels
                    self._enqueue_step(T, use_mask, exit_id, shadow)
                self._graphs[key] = g
            else:
                g.replay()
        else:
            self._enqueue_step(T, use_mask, exit_id, shadow)
        if shadow:
            self.h_state.copy_(self.h_shadow)
            self.c_state.copy_(self.c_shadow)
            self._head_state_changed()
        self.ctl_host.copy_(self.ctl, non_blocking=True)
        if self.A > 1:
            self.act_ext_host.copy_(self.act_ext, non_blocking=True)
        self.cur_step += 1
        if not sync:
            return None
        torch.cuda.current_stream().synchronize()
        return self.read_result()

    def _vision_chain_graphs(self):
        """Graphs of the vision tower as independent CHAINS of camera frames (chain 0 replays on the caller's stream, the
        others on side streams): each chain's launch boundaries and latency-bound kernels hide behind the other chain's
        work.  Every chain is two graphs, the first one SHORT (begin + patch embedding + a few ViT blocks): submitting a
        graph costs the host ~12 us + 0.3 us per node and the GPU idles until the first piece of a step is submitted.
        Captured once (the tower does not depend on the text length); the caller must have run the tower eagerly before."""
        C = self._graphs.get("chains")
        if C is None:
            def cap(fn):
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    fn()
                return g
            C = {"chain_head": [cap(lambda c=c: self._enqueue_chain(c, "head")) for c in range(self.n_chains)],
                 "chain_tail": [cap(lambda c=c: self._enqueue_chain(c, "tail")) for c in range(self.n_chains)],
                 "ev_in": torch.cuda.Event(), "ev_join": [torch.cuda.Event() for _ in range(self.n_chains)]}
            self._graphs["chains"] = C
        return C

    def _replay_vision_chains(self, main_st, side):
        """chain 0 on the main stream, the other chains on the side stream(s) (they start once the inputs are in place);
        the main stream continues after all of them (media tokens complete)."""
        C = self._graphs["chains"]
        nch = self.n_chains
        cst = [main_st] + [side if side is main_st or c == 1 else self._chain_stream(c) for c in range(1, nch)]
        if nch > 1:
            C["ev_in"].record(main_st)
            for c in range(1, nch):
                cst[c].wait_event(C["ev_in"])
        for part in ("chain_head", "chain_tail"):
            for c in range(nch):
                if c == 0:
                    C[part][0].replay()
                else:
                    with torch.cuda.stream(cst[c]):
                        C[part][c].replay()
        if nch > 1:
            if self._trace is not None:
                self._trace.append(("chain0 (main)", self._mark(main_st), time.perf_counter()))
                self._trace.append(("chain1 (side)", self._mark(cst[1]), time.perf_counter()))
            for c in range(1, nch):
                if cst[c] is not main_st:
                    ev = C["ev_join"][c]
                    ev.record(cst[c])
                    main_st.wait_event(ev)

    def _step_static_pieces(self, T, use_mask, exit_id):
        """Static ``exit_id`` step with the two-chain vision tower: chain graphs, then ONE graph for the media K/V GEMM,
        the embedding, layers 0..exit_id and the committing head call.  Same results as the single-graph schedule."""
        key = (T, use_mask, "static", exit_id)
        g = self._graphs.get(key)
        main_st = torch.cuda.current_stream()
        if g is None:
            self.hold_dev.copy_(self.step_info_pinned, non_blocking=True)
            self._enqueue_step(T, use_mask, exit_id)              # eager warm-up - a real step
            main_st.synchronize()
            self._vision_chain_graphs()
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                self._enqueue_media_kv()
                self.enqueue_llm_static(T, use_mask, exit_id)
            self._graphs[key] = g
        else:
            tm = self._time_stages
            if tm:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(main_st)
            self._replay_vision_chains(main_st, self._side_stream if self._use_side else main_st)
            if tm:
                ev[1].record(main_st)
            g.replay()
            if tm:
                ev[2].record(main_st)
        self.ctl_host.copy_(self.ctl, non_blocking=True)
        if self.A > 1:
            self.act_ext_host.copy_(self.act_ext, non_blocking=True)
        main_st.synchronize()
        if self._time_stages and "ev" in locals():
            self.last_stage_ms = {"vision": ev[0].elapsed_time(ev[1]), "llm_and_exit_checks": ev[1].elapsed_time(ev[2])}
        return self.read_result()

    def _step_one_graph(self, T, use_mask):
        """Dynamic step as ONE graph with parallel branches (opt-in, DEER_ONE_GRAPH=1): both vision chains, the trunk layers and -
        forked off after each exit layer - the head evaluations.  The host replays it and reads the verdict; nothing on the host
        sits between an exit decision and the end of the step.  Kernels behind the exit return at entry (~1.6 us each, ~150
        launches after an exit at layer 1), which is why the host-fed pieces of _step_segmented are the default: measured on
        MI355X 242 vs 352 steps/s with one environment and 778 vs 862 with eight."""
        key = (T, use_mask, "one_graph")
        g = self._graphs.get(key)
        main_st = torch.cuda.current_stream()
        if g is None:
            self.hold_dev.copy_(self.step_info_pinned, non_blocking=True)
            self._enqueue_step(T, use_mask, None)                 # eager warm-up of the whole step - a real step
            main_st.synchronize()
            plan = self.dynamic_plan()
            side = self._side_stream
            keep = []

            def fork(src, dst):
                e = torch.cuda.Event()
                e.record(src)
                dst.wait_event(e)
                keep.append(e)

            g = torch.cuda.CUDAGraph()
            with _capture(g):
                cap = torch.cuda.current_stream()
                fork(cap, side)
                with torch.cuda.stream(side):
                    for c in range(1, self.n_chains):
                        self._enqueue_chain(c, "head")
                        self._enqueue_chain(c, "tail")
                self._enqueue_chain(0, "head")
                self._enqueue_chain(0, "tail")
                fork(side, cap)
                self._enqueue_media_kv()
                for i, need_pseudo, is_exit, _ in plan:
                    if self.B > 1 and i >= 2 and plan[i - 2][2]:
                        fork(side, cap)                           # compaction layer: needs the verdict of the exit check of layer i - 2
                    self.enqueue_dynamic_main(T, use_mask, i)
                    if need_pseudo or is_exit:
                        fork(cap, side)
                        with torch.cuda.stream(side):
                            self.enqueue_dynamic_heads(T, i, use_mask=use_mask)
                fork(side, cap)
            self._graphs[key] = g
            self._graphs[(key, "events")] = keep
            self.ctl_host.copy_(self.ctl, non_blocking=True)
            if self.A > 1:
                self.act_ext_host.copy_(self.act_ext, non_blocking=True)
            main_st.synchronize()
            return self.read_result()
        g.replay()
        self.ctl_host.copy_(self.ctl, non_blocking=True)
        if self.A > 1:
            self.act_ext_host.copy_(self.act_ext, non_blocking=True)
        main_st.synchronize()
        return self.read_result()

    def _drop_graphs(self):
        """forget every captured graph (exit configuration changed) together with the native plans built over them"""
        for pl in self._plans:
            self.lib.deer_step_plan_destroy(pl)
        self._plans = []
        self._graphs.clear()

    def _make_step_plan(self, P, plan):
        """Hand the captured graph pieces to the native step driver (include/deer_model.h: deer_step_plan_*)."""
        C = self._graphs["chains"]
        nch = self.n_chains
        vp = ctypes.c_void_p

        def execs(gs):
            return (vp * len(gs))(*[vp(g.raw_cuda_graph_exec()) if g is not None else None for g in gs])

        side = self._side_stream if self._use_side else None
        cstreams = [self._side_stream if (c == 1 or not self._use_side) else self._chain_stream(c) for c in range(1, nch)]
        if not self._use_side:                                   # DEER_SIDE=0: everything on the caller's stream
            cstreams = [None] * (nch - 1)
        heads = [P["head"].get(i) for i, _, _, _ in plan]
        is_exit = (ctypes.c_int * len(plan))(*[1 if e else 0 for _, _, e, _ in plan])
        out = ctypes.c_void_p()
        abi.check(self.lib.deer_step_plan_create(nch, execs(C["chain_head"]), execs(C["chain_tail"]), len(plan), execs(P["main"]), execs(heads),
                                                 is_exit, self._lookahead, self.B, abi.ptr(self.step_info_pinned), abi.ptr(self.host_mirror),
                                                 ctypes.byref(out)), "deer_step_plan_create")
        self._plans.append(out)
        return {"plan": out, "chain_streams": (vp * max(nch - 1, 1))(*[vp(st.cuda_stream) if st is not None else None for st in cstreams] or [None]),
                "head_stream": vp(side.cuda_stream) if side is not None else None, "ctl_out": abi.ptr(self.ctl_host), "keep": (C, P)}

    def _step_segmented(self, T, use_mask):
        """Dynamic step, host-fed in PIECES: one graph per trunk layer (piece 0 = vision + embedding + layer 0) on the main
        stream, one graph per head evaluation on a side stream.

        * The head evaluation of exit k (8 launches, ~90 us) runs CONCURRENTLY with the next trunk layers: the trunk does
          not depend on it, only the decision to stop does.  The pseudo-action evaluation overlaps layer 1 the same way.
        * Verdicts reach the host through the pinned mirror (csrc/head.hip::check_done).  The host keeps at most LOOKAHEAD
          trunk layers in flight beyond an undecided exit check and stops enqueueing at the exit, so a few launches return
          at entry instead of every remaining one, and the device never waits for the host.
        * The action is on the host as soon as the exit check stored it.

        Results are identical to the single-stream schedule: no data flows from a head evaluation back into the trunk."""
        key = (T, use_mask, "pieces")
        P = self._graphs.get(key)
        plan = self.dynamic_plan()
        main_st = torch.cuda.current_stream()
        if P is None:
            self.hold_dev.copy_(self.step_info_pinned, non_blocking=True)
            self._enqueue_step(T, use_mask, None)                 # eager warm-up of the whole step - a real step
            main_st.synchronize()
            P = {"main": [], "head": {}, "ev": {}, "ev_done": {}}
            self._vision_chain_graphs()
            for i, need_pseudo, is_exit, _ in plan:
                g = torch.cuda.CUDAGraph()
                with _capture(g):
                    if i == 0:
                        self._enqueue_media_kv()
                    self.enqueue_dynamic_main(T, use_mask, i)
                P["main"].append(g)
                if need_pseudo or is_exit:
                    gh = torch.cuda.CUDAGraph()
                    with _capture(gh):
                        self.enqueue_dynamic_heads(T, i, use_mask=use_mask)
                    P["head"][i] = gh
                    P["ev"][i] = torch.cuda.Event()
                    P["ev_done"][i] = torch.cuda.Event()
            P["native"] = self._make_step_plan(P, plan)
            self._graphs[key] = P
            self.ctl_host.copy_(self.ctl, non_blocking=True)     # first call: verdict through the ordinary read-back
            if self.A > 1:
                self.act_ext_host.copy_(self.act_ext, non_blocking=True)
            main_st.synchronize()
            return self.read_result()

        if self._native_step and self._trace is None and not self._time_stages:
            # the native driver (csrc/step_driver.hip): the same submission / look-ahead / polling loop as below, one GIL-free call
            nat = P["native"]
            rc = self.lib.deer_step_plan_run(nat["plan"], int(self._si_np[0]), self._seq, ctypes.c_void_p(main_st.cuda_stream),
                                             nat["chain_streams"], nat["head_stream"], nat["ctl_out"], None)
            if rc == 3:
                raise abi.DeerHipError("no exit verdict from the device within 20 s" + (" (a hand-off of the one-launch head evaluation "
                                       "timed out)" if self.head_fused_error() else ""))
            abi.check(rc, "deer_step_plan_run")
            self._ext_from_mirror()
            return self.read_result()

        hm, seq, W = self._hm, self._seq, abi.CTL_WORDS
        if self._trace is not None:
            self._trace.append(("start", self._mark(main_st), time.perf_counter()))
        side = self._side_stream if self._use_side else main_st
        LOOK = self._lookahead
        exits = [i for i, _, is_exit, _ in plan if is_exit]      # exit k is decided by the check at layer exits[k]
        decided = 0                                              # number of exit checks whose verdict the host has seen
        done = False

        def poll(n_checks):
            """spin until `n_checks` checks of this step are decided or every environment exited"""
            want, spins, t_dead = seq * 64 + n_checks, 0, None
            while hm[abi.HOSTM_DONE] != seq and hm[abi.HOSTM_PROGRESS] < want:
                spins += 1
                if spins & 0xFFFF == 0:
                    t_dead = t_dead or time.monotonic() + 20.0
                    if time.monotonic() > t_dead:
                        raise abi.DeerHipError("no exit verdict from the device within 20 s (check %d)" % (n_checks - 1))
            return hm[abi.HOSTM_DONE] == seq

        tm = self._time_stages
        if tm:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(main_st)
        self._replay_vision_chains(main_st, side)
        if tm:
            ev[1].record(main_st)
        for i, need_pseudo, is_exit, _ in plan:
            # keep at most LOOKAHEAD trunk layers in flight beyond an undecided check
            while decided < len(exits) and exits[decided] + LOOK < i:
                done = poll(decided + 1)
                decided += 1
                if done:
                    break
            if done:
                break
            if self.B > 1 and side is not main_st and i >= 2 and (i - 2) in P["ev_done"]:
                main_st.wait_event(P["ev_done"][i - 2])           # compaction layer: the verdict of check i - 2 is complete (see csrc/step_driver.hip)
            P["main"][i].replay()
            if self._trace is not None:
                self._trace.append(("main%d" % i, self._mark(main_st), time.perf_counter()))
            if need_pseudo or is_exit:
                if side is main_st:
                    P["head"][i].replay()
                else:
                    hev = P["ev"][i]
                    hev.record(main_st)
                    side.wait_event(hev)
                    with torch.cuda.stream(side):
                        P["head"][i].replay()
                    if self.B > 1:
                        P["ev_done"][i].record(side)
                if self._trace is not None:
                    self._trace.append(("head%d" % i, self._mark(side), time.perf_counter()))
        if not done:
            done = poll(len(exits))
        if not done:                                             # the forced exit at the last check always fires
            raise abi.DeerHipError("dynamic step finished without an exit verdict")
        if tm:                                                   # per-stage GPU time of this step (reference hooks: eval_time)
            main_st.wait_stream(side)
            ev[2].record(main_st)
            ev[2].synchronize()
            self.last_stage_ms = {"vision": ev[0].elapsed_time(ev[1]), "llm_and_exit_checks": ev[1].elapsed_time(ev[2])}
        self._ctl_host_np[:] = hm[W:W * (1 + self.B)]            # keep the ordinary read-back buffer current (ctl_host users)
        self._ext_from_mirror()
        return self.read_result()

    # ------------------------------------------------------------------------- calibration (window mode)
    def sibling(self, n_envs: int, index: int = 0) -> "DeerEngine":
        """An engine for ``n_envs`` environments per step over the SAME weight arena (own workspace / LSTM state / controller).
        ``index`` > 0: a further one of the same size (two of them alternate in window mode, each on its own stream)."""
        if n_envs == self.B and index == 0:
            return self
        e = self._siblings.get((n_envs, index))
        if e is None:
            e = DeerEngine(self.cfg, None, device=self.dev, max_text_len=self.max_T, n_envs=n_envs, weights_from=self, segmented=self.segmented, precision=self.precision)
            self._siblings[(n_envs, index)] = e
        return e

    def window_hidden_states(self, rgb_seq: torch.Tensor, grip_seq: torch.Tensor, ids: torch.Tensor, mask=None, group: int = 8) -> torch.Tensor:
        """hidden_states of EVERY layer for each frame pair of calibration windows (the ``hidden_states`` the reference's window-mode
        forward hands to ``ActionValueNet(mode='generate')``: flamingo_mpt.py:463-517 on (bs*W) batch rows, value_net.py:375-386).
        rgb_seq / grip_seq: (F, 3, S, S) frames; ids (F, T) or (1, T) (one instruction for all frames); returns (F, n_layers, T, d)
        fp32 on the device.  The frames are BATCH ROWS here too: groups of ``group`` frames run through the env-batch sibling
        engine (ViT at M = 514*group rows, trunk at group*T rows), static full depth, every layer's output kept."""
        F = rgb_seq.shape[0]
        ids = ids.reshape(-1, ids.shape[-1])
        T = ids.shape[1]
        G = max(1, min(group, self.MAX_ROWS // T, F, abi.MAX_ENVS))
        last = self.cfg.n_layers - 1
        L, d = self.cfg.n_layers, self.cfg.d_model
        out = torch.empty(F, L, T, d, device=self.dev)
        # more than one group: two sibling engines alternate, each on its own stream - one group's MFMA-bound vision tower runs
        # beside the other's HBM-bound trunk (two groups of 6: 12.0 instead of 14.8 ms, tools/overlap_two_batches.py)
        n_groups = (F + G - 1) // G
        ws = [self.sibling(G)] + ([self.sibling(G, 1)] if n_groups > 1 else [])
        cur = torch.cuda.current_stream()
        sts = [cur] + ([self._side_stream] if n_groups > 1 else [])
        for w in ws:
            w._drain_side_streams()
        if n_groups > 1:
            self._side_stream.wait_stream(cur)                    # inputs produced on the caller's stream
        for gi, f0 in enumerate(range(0, F, G)):
            w, st = ws[gi % len(ws)], sts[gi % len(ws)]
            with torch.cuda.stream(st):
                idx = [min(f0 + i, F - 1) for i in range(G)]      # the last group is padded with its last frame
                ids_g = ids[idx] if ids.shape[0] == F else ids.expand(G, T)
                mask_g = None
                if mask is not None:
                    m2 = mask.reshape(-1, T)
                    mask_g = m2[idx] if m2.shape[0] == F else m2.expand(G, T)
                T_, use_mask = w.load_inputs(rgb_seq[idx], grip_seq[idx], ids_g.contiguous(), mask_g)
                w.hold_dev.zero_()
                w._enqueue_step(T_, use_mask, last)               # static full depth: every layer's output is kept
                n = min(G, F - f0)
                out[f0:f0 + n].copy_(w.hidden[:, :G * T].view(L, G, T, d).permute(1, 0, 2, 3)[:n])
        if n_groups > 1:
            cur.wait_stream(self._side_stream)
        return out

    def _head_eval(self, feats: torch.Tensor, commit: bool) -> torch.Tensor:
        """One DeterministicDecoder step of all n_envs sequences on ``feats`` (n_envs*T, d) from the current LSTM state; commit=True
        stores the new state (update_hidden_state protocol, action_head.py:548-558).  Returns the device rows [pose6, gripper prob,
        logit] per sequence, (n_envs, 8)."""
        self.enqueue_head(0, feats.shape[0] // self.B, abi.KIND_COMMIT, use_ctl=False, feats=feats, no_ctl_final=True)
        if commit:
            self.h_state.copy_(self.h_tmp)
            self.c_state.copy_(self.c_tmp)
            self._head_state_changed()
        return self.action_dbg.clone()

    def generate_values(self, hidden: torch.Tensor, rand_layers, threshold_type: str = "L2", group: int = 8) -> torch.Tensor:
        """``ActionValueNet.forward(mode='generate')`` (value_net.py:134-160): for the time steps seq_id in [W/2-1, W-1) the action of
        layer 0 and of every exit is predicted from [history = features of the random exit layers ``rand_layers[:seq_id]`` ; this
        layer's feature at seq_id] with the LSTM run from a zero state (window mode == carried-state steps); returns the deltas
        between consecutive exits.  hidden: (W, L, T, d) with rand_layers (W,) -> (n_exit, W - W/2), or a BATCH of windows
        (bs, W, L, T, d) with rand_layers (bs, W) -> (n_exit, bs * (W - W/2)) in the reference's order (window-major).  The windows
        of a group are the "environments" of one head evaluation (weights read once per group)."""
        single = hidden.dim() == 4
        if single:
            hidden = hidden.unsqueeze(0)
        rl = torch.as_tensor(rand_layers).reshape(hidden.shape[0], -1)
        bs, W, L, T, d = hidden.shape
        layers = [0] + list(self.exit_ids)
        NL = len(layers)
        NE = NL + 1
        Gw = min(group, bs, abi.MAX_ENVS // NE, self.MAX_ROWS // (NE * T))
        if Gw >= 1:
            # Round 5: the NL candidate layers of a time step AND the history feature are the "environments" of ONE head evaluation
            # (slot k < NL of a window: layer k's feature, slot NL: the random layer's; all from the same LSTM state, which every slot of
            # the window holds) - 11 evaluations per window instead of 53 (one per time step instead of NL + 1), the weights read once each
            w = self.sibling(Gw * NE)
            Lh, H = w.h_state.shape[0], w.h_state.shape[-1]
            per_window = []
            for b0 in range(0, bs, Gw):
                idx = [min(b0 + i, bs - 1) for i in range(Gw)]
                hg = hidden[idx]                                                   # (Gw, W, L, T, d)
                lay = torch.tensor([layers + [0]] * Gw)                            # (Gw, NE): the last slot is rewritten per time step
                w.h_state.zero_()
                w.c_state.zero_()
                acts = []
                gi = torch.arange(Gw).unsqueeze(1).expand(Gw, NE)
                for t in range(W - 1):
                    for g in range(Gw):
                        lay[g, NL] = int(rl[idx[g], t])
                    feats = hg[:, t][gi, lay]                                       # (Gw, NE, T, d)
                    out = w._head_eval(feats.reshape(Gw * NE * T, d).contiguous(), commit=False).view(Gw, NE, -1)
                    if t >= W // 2 - 1:
                        acts.append(out[:, :NL].transpose(0, 1))                   # (NL, Gw, 8 A)
                    # update_hidden_state=True for the history feature: its new state becomes the state of every slot of the window
                    ht, ct = w.h_tmp.view(Lh, Gw, NE, H), w.c_tmp.view(Lh, Gw, NE, H)
                    w.h_state.view(Lh, Gw, NE, H).copy_(ht[:, :, NL:NL + 1].expand(Lh, Gw, NE, H))
                    w.c_state.view(Lh, Gw, NE, H).copy_(ct[:, :, NL:NL + 1].expand(Lh, Gw, NE, H))
                    w._head_state_changed()
                a = torch.stack(acts, dim=2)[..., :6 * self.A]                     # (NL, Gw, W/2, 6 A)
                per_window.append(a[:, : min(Gw, bs - b0)])
            return self._values_from_actions(torch.cat(per_window, dim=1).cpu(), threshold_type)
        G = max(1, min(group, self.MAX_ROWS // T, bs, abi.MAX_ENVS))
        w = self.sibling(G)
        per_window = []
        for b0 in range(0, bs, G):
            idx = [min(b0 + i, bs - 1) for i in range(G)]
            hg = hidden[idx]                                                   # (G, W, L, T, d)
            w.h_state.zero_()
            w.c_state.zero_()
            acts = []
            for t in range(W - 1):
                if t >= W // 2 - 1:
                    acts.append(torch.stack([w._head_eval(hg[:, t, i].reshape(G * T, d).contiguous(), commit=False) for i in layers]))
                sel = torch.stack([hg[g, t, int(rl[idx[g], t])] for g in range(G)])                  # each window's own random layer
                w._head_eval(sel.reshape(G * T, d).contiguous(), commit=True)
            a = torch.stack(acts, dim=2)[..., :6 * self.A]                     # (n_exit+1, G, W/2, 6 A): the delta runs over all pose values
            per_window.append(a[:, : min(G, bs - b0)])
        return self._values_from_actions(torch.cat(per_window, dim=1).cpu(), threshold_type)

    @staticmethod
    def _values_from_actions(a: torch.Tensor, threshold_type: str) -> torch.Tensor:
        """a: (n_exit+1, bs, W/2, 6 A) actions of layer 0 and of every exit -> deltas between consecutive exits (value_net.py:105-117,160)"""
        prev, last = a[:-1], a[1:]
        dlt = (prev - last).abs()
        if threshold_type == "mean":
            v = dlt.mean(-1)
        elif threshold_type == "L2":
            v = dlt.pow(2).mean(-1).pow(0.5)
        elif threshold_type == "max":
            v = dlt.max(-1)[0]
        elif threshold_type == "cosine":
            v = 1 - torch.nn.functional.cosine_similarity(prev, last, dim=-1, eps=1e-5)
        else:
            raise NotImplementedError(threshold_type)
        return v.flatten(1, 2)                                                 # (n_exit, bs * W/2), window-major like value_net.py:160

    @staticmethod
    def _mark(stream):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def _ext_from_mirror(self):
        """multi_step_action > 1, pipelined steps: the committed / ensemble actions travel with the verdict behind the control blocks of
        the pinned mirror (csrc/head.hip) - bring them into the read-back buffer read_result decodes"""
        if self.A > 1:
            W = abi.CTL_WORDS
            ext = self._hm[W * (1 + self.B):].view(np.float32).reshape(self.B, 2, 64)
            self.act_ext_host.numpy()[:, 1:3] = ext

    def _check_device_errors(self):
        """the experimental persistent trunk layer reports a timed-out barrier in an error word: a step whose kernels gave up waiting ran
        on with undefined data, so the step raises instead of returning a result"""
        if getattr(self, "_persistent_on", False) and self.persistent_layer_error() != 0:
            raise abi.DeerHipError(f"persistent trunk layer: barrier timed out {self.persistent_layer_error_detail()}")

    def read_result(self, src=None):
        """Decode the per-environment control blocks (int32 numpy view; default: the pinned read-back buffer).  multi_step_action = A > 1:
        ``pose`` is (6 A,) = A consecutive 6-DoF actions and ``gripper`` / ``gripper_logit`` are (A,) tensors, the layout of the
        reference head's outputs (action_head.py:472-473, eval_utils.py:468-471)."""
        self._check_device_errors()
        W = abi.CTL_WORDS
        ci = (self._ctl_host_np if src is None else src).reshape(self.B, W)
        cf = ci.view(np.float32)
        out = []
        A = self.A
        for b in range(self.B):
            c, f = ci[b], cf[b]
            if int(c[abi.CTL_EXIT_LAYER]) < 0:
                raise abi.DeerHipError(f"environment {b}: the step ended without an exit verdict (no exit check was forced)")
            a = f[abi.CTL_OUT_ACTION: abi.CTL_OUT_ACTION + 8].copy()
            en = f[abi.CTL_ENS_ACTION: abi.CTL_ENS_ACTION + 8].copy()   # ActionValueNet.get_ensemble_action (dynamic steps only)
            r = dict(exit_layer=int(c[abi.CTL_EXIT_LAYER]), n_evals=int(c[abi.CTL_N_EVALS]),
                     pose=torch.from_numpy(a[:6]), gripper=float(a[6]), gripper_logit=float(a[7]),
                     ens_pose=torch.from_numpy(en[:6]), ens_gripper=float(en[6]), ens_count=int(en[7]),
                     deltas=torch.from_numpy(f[abi.CTL_DELTAS: abi.CTL_DELTAS + 16].copy()))
            if A > 1:
                x = self.act_ext_host[b].clone()                      # rows: previous | committed | ensemble
                r.update(pose=x[1, :6 * A], gripper=x[1, 6 * A:7 * A], gripper_logit=x[1, 7 * A:8 * A],
                         ens_pose=x[2, :6 * A], ens_gripper=x[2, 6 * A:7 * A])
            out.append(r)
        return out[0] if self.B == 1 else out

    # ------------------------------------------------------------------------- layerwise_exit_eval (flamingo_mpt.py:450-457)
    def layerwise_actions(self, exit_layers, T: int, use_mask: bool = False):
        """The action of every environment from the head of ITS exit layer (``lm_exits[k]`` / ``lm_head``) on hidden_states[k], each head
        running from its own LSTM state, which advances only for the environments that exited at its layer.  Returns a list of (8 A,)
        rows [pose | gripper prob | logit] per environment.  One head evaluation per DISTINCT exit layer of the batch (an ablation mode of
        the reference's harness, eval_calvin.py:330 - not a throughput path)."""
        rows = [None] * self.B
        for layer in sorted(set(int(v) for v in exit_layers)):
            head = self.lib.deer_model_layerwise_head(self._h, layer)
            if head < 1:
                raise abi.DeerHipError(f"layerwise_exit_eval: layer {layer} has no head of its own (exit layers: {self.cfg.exit_ids()})")
            abi.check(self.lib.deer_head_eval_layerwise(self._h, head, layer, T, 1 if (use_mask and self.B > 1) else 0, _cur_stream()),
                      "deer_head_eval_layerwise")
            out = self.action_dbg.clone()
            for b in range(self.B):
                if int(exit_layers[b]) == layer:
                    rows[b] = out[b].cpu()
                    self.lw_state[head - 1, 0, :, b].copy_(self.h_tmp[:, b])      # update_hidden_state=True for this head / environment
                    self.lw_state[head - 1, 1, :, b].copy_(self.c_tmp[:, b])
        return rows

    def weight_bytes(self) -> int:
        """device bytes of the weight arena (bf16 GEMM operands, f32 norms / biases / gates)"""
        return int(self.arena.numel())
