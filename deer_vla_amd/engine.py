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
from .config import DeerConfig
from .synthetic import mlp_layer_indices

EPS = 1e-5


def _pick_split(M: int, N: int, K: int, target_blocks: int = 320, max_split: int = 8) -> int:
    """Split-K factor for a projection whose 64x64 output tiles alone cannot fill 256 CUs: double while the grid stays
    under ~1.25 workgroups per CU (the f32 slabs cost HBM traffic: measured optimum 2,2 for ViT-L at 514 rows) and every slice keeps >= 2 K-steps of 64."""
    blocks = ((M + 63) // 64) * ((N + 63) // 64)
    S = 1
    while S * 2 <= max_split and blocks * S * 2 <= target_blocks and K % (S * 2 * 64) == 0 and K // (S * 2) >= 128:
        S *= 2
    return S


def _i32(v: int) -> int:
    """two's-complement wrap into int32 (halves of a pointer stored in an int32 tensor)"""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def _cur_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _ProfiledLib:
    """Proxy around the ctypes library: when ``engine._prof`` is a list every kernel-launching ABI call is bracketed
    by two HIP events on the current stream (bench.py's in-situ roofline pass).  Pass-through otherwise."""
    _HOST_ONLY = ("deer_skinny_splitk", "deer_hip_arch", "deer_hip_abi_version")

    def __init__(self, lib, eng):
        self._lib, self._eng = lib, eng

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self._HOST_ONLY:
            return fn
        eng = self._eng

        def call(*a):
            if eng._prof is None:
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            eng._prof.append((name, e0, e1) + eng._cost)
            return rc
        return call


class DeerEngine:
    LOOKAHEAD = 1        # trunk layers the host keeps in flight beyond an undecided exit check

    def __init__(self, cfg: DeerConfig, state_dict: Dict[str, torch.Tensor], device="cuda", max_text_len: int = 32,
                 n_envs: int = 1, threshold_type: str = "L2", leq: bool = True, segmented: bool = True):
        """n_envs: independent environments evaluated per control step (one "env batch" per rank).  They share every
        weight read: the ViT sees M = 514*n_envs rows, the LLM n_envs*T rows, each environment keeps its own LSTM state,
        thresholds are shared and every environment exits at its own layer (device side)."""
        if not torch.cuda.is_available():
            raise abi.DeerHipError("DeerEngine needs a HIP device (no CPU fallback exists in deer_vla_amd)")
        self._prof = None            # when a list: every launch is bracketed by HIP events (bench roofline pass)
        self._cost = (0.0, 0.0)
        self.lib = _ProfiledLib(abi.lib(), self)
        self.cfg = cfg
        self.dev = torch.device(device)
        assert 1 <= n_envs <= 8
        self.B = n_envs
        self.n_cams = 2 * n_envs                       # images per step: (rgb, gripper) of every environment
        self.max_T = max_text_len
        assert n_envs * 14 <= 64, "skinny GEMM handles <= 64 rows"
        self.thr_type = abi.THR_TYPES[threshold_type]
        self.leq = 1 if leq else 0
        assert cfg.vit_head_dim == 64 and cfg.perc_dim_head == 64 and cfg.xattn_dim_head == 64, "head_dim 64 kernels"
        assert cfg.head_dim <= 128 and cfg.d_model % 32 == 0
        assert cfg.n_media <= 128, "xattn_small kernel holds <=128 media tokens"
        self._keep: List[torch.Tensor] = []
        self._load_weights(state_dict)
        self._alloc_workspace()
        self._graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self.segmented = segmented                    # dynamic steps fed in per-layer graph pieces (see _step_segmented)
        self._side_stream = torch.cuda.Stream(device=self.dev)
        self._extra_streams: List[torch.cuda.Stream] = []
        self._use_side = os.environ.get("DEER_SIDE", "1") == "1"          # debugging knobs (README)
        self._lookahead = int(os.environ.get("DEER_LOOKAHEAD", self.LOOKAHEAD))
        self._seq = 0
        self._ids_tag = None
        self._trace = None                            # debugging aid: list of (label, event, host time) per piece
        # controller configuration (set by configure_exit)
        self.exit_ids = cfg.exit_ids()
        self.ctl_max_layer = self.exit_ids[-1]
        self.steps_per_stage = 1
        self.reset()

    # ------------------------------------------------------------------------------------------ weights
    def _dev(self, t: torch.Tensor, dtype) -> torch.Tensor:
        return t.detach().to(device=self.dev, dtype=dtype).contiguous()

    def _bf(self, t):
        return self._dev(t, torch.bfloat16)

    def _f32(self, t):
        return self._dev(t, torch.float32)

    def _packed(self, t: torch.Tensor) -> torch.Tensor:
        """row-major [N,K] -> MFMA-fragment order for the skinny GEMM (deer_pack_weight_mfma16)."""
        w = self._bf(t)
        N, K = w.shape
        out = torch.empty_like(w)
        abi.check(self.lib.deer_pack_weight_mfma16(abi.ptr(w), abi.ptr(out), N, K, _cur_stream()), "pack_weight")
        torch.cuda.current_stream().synchronize()
        return out

    def _load_weights(self, sd: Dict[str, torch.Tensor]):
        cfg = self.cfg
        W = cfg.vit_width
        g = lambda k: sd[k]
        self.vit = {}
        v = "vision_encoder.visual."
        kk = 3 * cfg.patch_size ** 2
        self.patch_kpad = (kk + 63) // 64 * 64
        conv = g(v + "conv1.weight").reshape(W, kk)
        convp = torch.zeros(W, self.patch_kpad, dtype=conv.dtype, device=conv.device)
        convp[:, :kk] = conv
        self.vit["conv"] = self._bf(convp)
        self.vit["cls"] = self._f32(g(v + "class_embedding"))
        self.vit["pos"] = self._f32(g(v + "positional_embedding"))
        self.vit["ln_pre_w"], self.vit["ln_pre_b"] = self._f32(g(v + "ln_pre.weight")), self._f32(g(v + "ln_pre.bias"))
        self.vit_layers = []
        for l in range(cfg.vit_layers):
            p = f"{v}transformer.resblocks.{l}."
            self.vit_layers.append(dict(
                ln1w=self._f32(g(p + "ln_1.weight")), ln1b=self._f32(g(p + "ln_1.bias")),
                wqkv=self._bf(g(p + "attn.in_proj_weight")), bqkv=self._f32(g(p + "attn.in_proj_bias")),
                wo=self._bf(g(p + "attn.out_proj.weight")), bo=self._f32(g(p + "attn.out_proj.bias")),
                ln2w=self._f32(g(p + "ln_2.weight")), ln2b=self._f32(g(p + "ln_2.bias")),
                wfc=self._bf(g(p + "mlp.c_fc.weight")), bfc=self._f32(g(p + "mlp.c_fc.bias")),
                wpr=self._bf(g(p + "mlp.c_proj.weight")), bpr=self._f32(g(p + "mlp.c_proj.bias"))))
        self.perc = dict(latents=self._f32(g("perceiver.latents")), normw=self._f32(g("perceiver.norm.weight")),
                         normb=self._f32(g("perceiver.norm.bias")))
        self.perc_layers = []
        for l in range(cfg.perc_depth):
            a, f = f"perceiver.layers.{l}.0.", f"perceiver.layers.{l}.1."
            self.perc_layers.append(dict(
                nlw=self._f32(g(a + "norm_latents.weight")), nlb=self._f32(g(a + "norm_latents.bias")),
                # latents are projected to q | k | v in one GEMM: [to_q ; to_kv] stacked on the output dimension
                wqkv=self._bf(torch.cat([g(a + "to_q.weight"), g(a + "to_kv.weight")], 0)), wo=self._bf(g(a + "to_out.weight")),
                fnw=self._f32(g(f + "0.weight")), fnb=self._f32(g(f + "0.bias")),
                w1=self._bf(g(f + "1.weight")), w2=self._bf(g(f + "3.weight"))))
        # the media tokens are the same in every Perceiver layer: norm_media / to_kv of ALL layers are applied up front
        pl = [f"perceiver.layers.{l}.0." for l in range(cfg.perc_depth)]
        self.perc_nm_w = self._f32(torch.stack([g(a + "norm_media.weight") for a in pl]))
        self.perc_nm_b = self._f32(torch.stack([g(a + "norm_media.bias") for a in pl]))
        self.perc_wkv_all = self._bf(torch.stack([g(a + "to_kv.weight") for a in pl]))        # [L][2*inner][W]
        # ---- LLM ----
        self.wte = self._bf(g("lang_encoder.transformer.wte.weight"))
        nine_b = cfg.llm_name == "mpt_9b"
        ln1, ln2 = ("norm_1", "norm_2") if nine_b else ("ln_1", "ln_2")
        up, down = ("ffn.up_proj", "ffn.down_proj") if nine_b else ("mlp.mlp_up", "mlp.mlp_down")
        self.llm_layers = []
        kv_rows = []
        for n in range(cfg.n_layers):
            blk = f"lang_encoder.transformer.blocks.{n}."
            L = {}
            if cfg.has_xattn(n):
                x = blk + "gated_cross_attn_layer."
                L["xa"] = dict(nw=self._f32(g(x + "attn.norm.weight")), nb=self._f32(g(x + "attn.norm.bias")),
                               wq=self._packed(g(x + "attn.to_q.weight")), wo=self._packed(g(x + "attn.to_out.weight")),
                               ag=self._f32(g(x + "attn_gate")), fg=self._f32(g(x + "ff_gate")),
                               fnw=self._f32(g(x + "ff.0.weight")), fnb=self._f32(g(x + "ff.0.bias")),
                               w1=self._packed(g(x + "ff.1.weight")), w2=self._packed(g(x + "ff.3.weight")),
                               kv_index=len(kv_rows))
                kv_rows.append(g(x + "attn.to_kv.weight"))
            m = blk + "decoder_layer."
            L["ln1w"] = self._f32(g(m + ln1 + ".weight"))
            L["ln1b"] = self._f32(sd[m + ln1 + ".bias"]) if (m + ln1 + ".bias") in sd else None
            L["wqkv"] = self._packed(g(m + "attn.Wqkv.weight"))
            L["qlnw"] = self._f32(g(m + "attn.q_ln.weight")) if cfg.attn_qk_ln else None
            L["klnw"] = self._f32(g(m + "attn.k_ln.weight")) if cfg.attn_qk_ln else None
            L["wo"] = self._packed(g(m + "attn.out_proj.weight"))
            L["ln2w"] = self._f32(g(m + ln2 + ".weight"))
            L["ln2b"] = self._f32(sd[m + ln2 + ".bias"]) if (m + ln2 + ".bias") in sd else None
            L["wup"] = self._packed(g(m + up + ".weight"))
            L["wdown"] = self._packed(g(m + down + ".weight"))
            self.llm_layers.append(L)
        # K/V projections of the media tokens for ALL x-attn layers in one GEMM (media is layer-invariant)
        self.n_xattn = len(kv_rows)
        self.xinner = cfg.xattn_heads * cfg.xattn_dim_head
        self.wkv_all = self._bf(torch.cat(kv_rows, dim=0)) if kv_rows else None
        # ---- head ----
        self.head = self._load_head(sd, "extra_exit.")

    def _load_head(self, sd, p):
        cfg = self.cfg
        H = {"lstm": [], "fc": []}
        for l in range(cfg.lstm_num_layers):
            if cfg.lstm_layernorm:
                r, sfx = f"{p}rnn.layers.{3 * l}.", "_l0"
            else:
                r, sfx = f"{p}rnn.", f"_l{l}"
            d = dict(wih=self._bf(sd[r + "weight_ih" + sfx]), whh=self._bf(sd[r + "weight_hh" + sfx]),
                     bih=self._f32(sd[r + "bias_ih" + sfx]), bhh=self._f32(sd[r + "bias_hh" + sfx]))
            if cfg.lstm_layernorm:
                d["lnw"] = self._f32(sd[f"{p}rnn.layers.{3 * l + 1}.weight"])
                d["lnb"] = self._f32(sd[f"{p}rnn.layers.{3 * l + 1}.bias"])
            H["lstm"].append(d)
        lin, ln, out = mlp_layer_indices(cfg.mlp_num_hidden_layers)
        for li, ni in zip(lin, ln):
            d = {}
            for gi, hname in enumerate(("actions", "gripper")):
                d[f"w{gi}"] = self._bf(sd[f"{p}{hname}.mlp.{li}.weight"])
                d[f"b{gi}"] = self._f32(sd[f"{p}{hname}.mlp.{li}.bias"])
                if cfg.mlp_layernorm:
                    d[f"lnw{gi}"] = self._f32(sd[f"{p}{hname}.mlp.{ni}.weight"])
                    d[f"lnb{gi}"] = self._f32(sd[f"{p}{hname}.mlp.{ni}.bias"])
            H["fc"].append(d)
        H["wa"], H["ba"] = self._bf(sd[f"{p}actions.mlp.{out}.weight"]), self._f32(sd[f"{p}actions.mlp.{out}.bias"])
        H["wg"], H["bg"] = self._bf(sd[f"{p}gripper.mlp.{out}.weight"]), self._f32(sd[f"{p}gripper.mlp.{out}.bias"])
        return H

    # ---------------------------------------------------------------------------------------- workspace
    def _vision_ws(self, n, img=None, vis_x=None, vis_x_f32=None, splits=None):
        """Activation buffers of the vision tower for n camera frames (SimpleNamespace; see enqueue_vision)."""
        cfg, dev = self.cfg, self.dev
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        bf = torch.bfloat16
        P, W, S = cfg.n_patches, cfg.vit_width, cfg.image_size
        nl, inner, Lp = cfg.perc_latents, cfg.perc_heads * cfg.perc_dim_head, cfg.perc_depth
        R = n * (P + 1)
        ws = SimpleNamespace(n=n)
        ws.img = z(n, 3, S, S, dt=bf) if img is None else img     # static input buffer (camera frames batched)
        ws.im2col = z(n * P, self.patch_kpad, dt=bf)
        ws.patch_out = z(n * P, W)
        ws.vx = z(n, P + 1, W)                                    # ViT residual stream (fp32)
        ws.v_ln = z(R, W, dt=bf)
        ws.v_qkv = z(R, 3 * W, dt=bf)
        ws.v_ao = z(R, W, dt=bf)
        ws.v_h = z(R, cfg.vit_mlp, dt=bf)
        # split-K factors of the residual projections (measured on MI355X, tools/bench_gemm.py; DEER_VIT_SPLIT overrides)
        ws.vit_split = (_pick_split(R, W, W), _pick_split(R, W, cfg.vit_mlp))
        ws.perc_split = (_pick_split(n * nl, W, inner), _pick_split(n * nl, W, cfg.perc_ff_mult * W))
        ov = os.environ.get("DEER_VIT_SPLIT")
        if ov:
            v = [int(t) for t in ov.split(",")]
            ws.vit_split, ws.perc_split = (v[0], v[1]), (v[2], v[3])
        if splits is not None:                                    # same summation order as the batched schedule: results of
            ws.vit_split, ws.perc_split = splits.vit_split, splits.perc_split   # all schedules are bit-identical
        ws.v_slab = z(max(max(ws.vit_split) * R, max(ws.perc_split) * n * nl) * W)
        ws.p_lat = z(n, nl, W)
        ws.p_mln = z(Lp, n * P, W, dt=bf)                         # norm_media_l(x) for every layer l
        ws.p_mkv = z(Lp, n * P, 2 * inner, dt=bf)                 # to_kv_l of it: media K | V of every layer
        ws.p_latln = z(n * nl, W, dt=bf)                          # norm_latents(latents) of the current layer
        ws.p_qkv = z(n * nl, 3 * inner, dt=bf)                    # q | k | v of the latents
        ws.p_ao = z(n * nl, inner, dt=bf)
        ws.p_ln = z(n * nl, W, dt=bf)
        ws.p_h = z(n * nl, cfg.perc_ff_mult * W, dt=bf)
        ws.vis_x = z(n * nl, W, dt=bf) if vis_x is None else vis_x            # media tokens [rgb latents ; gripper latents]
        ws.vis_x_f32 = z(n * nl, W) if vis_x_f32 is None else vis_x_f32
        return ws

    def _alloc_workspace(self):
        cfg, dev = self.cfg, self.dev
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        bf = torch.bfloat16
        N, P, W, S = self.n_cams, cfg.n_patches, cfg.vit_width, cfg.image_size
        nl, inner = cfg.perc_latents, cfg.perc_heads * cfg.perc_dim_head
        # vision workspace: one set for all camera frames batched (single-stream schedules), and one PRIVATE set per chain of
        # the two-stream schedule (the camera frames are independent until the media tokens are concatenated; chains share
        # only the input frames and the media-token output, as row ranges)
        self.vws = self._vision_ws(N)
        self.img, self.vx, self.vis_x, self.vis_x_f32 = self.vws.img, self.vws.vx, self.vws.vis_x, self.vws.vis_x_f32
        self.vchains = []
        n_ch = max(1, min(N, int(os.environ.get("DEER_CHAINS", "2"))))
        per = (N + n_ch - 1) // n_ch
        for c in range(n_ch):
            lo, hi = c * per, min(N, (c + 1) * per)
            self.vchains.append(self._vision_ws(hi - lo, img=self.img[lo:hi], vis_x=self.vis_x[lo * nl: hi * nl],
                                                vis_x_f32=self.vis_x_f32[lo * nl: hi * nl], splits=self.vws))
        self.kv_all = z(N * nl, max(self.n_xattn, 1) * 2 * self.xinner, dt=bf)
        d = cfg.d_model
        T = min(self.B * self.max_T, 64)                     # LLM rows = n_envs * text length
        self.max_rows = T
        self.ids = torch.zeros(T, dtype=torch.int64, device=dev)
        self.key_mask = torch.ones(T, dtype=torch.uint8, device=dev)
        self.text_time = torch.zeros(T, dtype=torch.int32, device=dev)
        self.x = z(T, d)
        self.xn = z(T, d)                                   # LN(x), fp32 (split into bf16 hi+lo inside the GEMM)
        self.ao = z(T, max(d, self.xinner))                 # attention outputs, fp32
        max_n = max(cfg.mlp_ratio * d, cfg.xattn_ff_mult * d, 3 * d)
        self.max_split = 32
        self.slab_a = z(self.max_split * 64 * d)             # outputs of width d (residual branches)
        self.slab_b = z(16 * 64 * max_n)                     # outputs of width 3d / 4d / inner
        self.qkv_ws = z(T, 3 * d)                            # reduced (+ q/k-normalised) qkv of the MPT attention
        self.hidden = z(cfg.n_layers, T, d)                  # hidden_states[i] = output of layer i
        Lh, H, B = cfg.lstm_num_layers, cfg.head_hidden, self.B
        self.h_state, self.c_state = z(Lh, B, H), z(Lh, B, H)     # LSTM state of every environment: [layer][env][H]
        self.h_tmp, self.c_tmp = z(Lh, B, H), z(Lh, B, H)
        self.h_shadow, self.c_shadow = z(Lh, B, H), z(Lh, B, H)   # commit target in shadow (calibration) mode
        dims = cfg.mlp_hidden_dims
        self.z_fc = [z(B, 2 * dm) for dm in dims]
        self.pooled = z(B, d)                                     # max/avg-pooled features of the current head evaluation
        self.ctl = torch.zeros(B * abi.CTL_WORDS, dtype=torch.int32, device=dev)   # one control block per environment
        self.ctl_host = torch.zeros(B * abi.CTL_WORDS, dtype=torch.int32).pin_memory()
        self._ctl_host_np = self.ctl_host.numpy()
        self.hold_dev = torch.zeros(4, dtype=torch.int32, device=dev)      # step_info: {hold, seq, host mirror ptr lo, hi}
        self.step_info_host = torch.zeros(8, 4, dtype=torch.int32).pin_memory()   # ring: an async upload may still be pending
        # host mirror of the verdicts (pinned => device-visible and system-coherent): see csrc/head.hip::check_done
        self.host_mirror = torch.zeros((1 + B) * abi.CTL_WORDS, dtype=torch.int32).pin_memory()
        self._hm = self.host_mirror.numpy()                                 # polled by the host between graph segments
        self.step_info_pinned = torch.zeros(4, dtype=torch.int32).pin_memory()   # read by the device in pipelined steps
        self._si_np = self.step_info_pinned.numpy()
        mptr = self.host_mirror.data_ptr()
        self._si_np[2], self._si_np[3] = _i32(mptr & 0xFFFFFFFF), _i32(mptr >> 32)
        self.thresholds = torch.full((16,), 1e8, dtype=torch.float32, device=dev)
        self.action_dbg = z(B, 8)

    # ------------------------------------------------------------------------------------- small helpers
    @contextmanager
    def _rec(self, name, flops=0.0, nbytes=0.0):
        """Attach algorithmic flops / bytes to the next launch (consumed by the profiling proxy, if active)."""
        self._cost = (float(flops), float(nbytes))
        yield
        self._cost = (0.0, 0.0)

    def _gemm(self, A, W, C, M, N, K, epi, bias=None, gate=None, lda=None, ldc=None, batch=1, strideA=0, strideC=0,
              tile=0, a_off=0, c_off=0):
        lda = K if lda is None else lda
        ldc = N if ldc is None else ldc
        out_b = 4 if epi in (abi.EPI_F32, abi.EPI_RESADD_F32) else 2
        with self._rec("gemm_tiled", 2.0 * M * N * K * batch, 2.0 * (M * K * batch + N * K) + out_b * M * N * batch):
            abi.check(self.lib.deer_gemm_bf16_nt(abi.ptr(A, a_off), lda, strideA, abi.ptr(W), K, abi.ptr(bias), abi.ptr(C, c_off),
                                                 ldc, strideC, M, N, K, batch, epi, abi.ptr(gate), tile, None, _cur_stream()),
                      "deer_gemm_bf16_nt")

    def _gemm_splitk(self, A, W, slab, M, N, K, S, tile=0):
        assert S * M * N <= slab.numel(), (S, M, N, slab.numel())
        with self._rec("gemm_tiled", 2.0 * M * N * K, 2.0 * (M * K + N * K) + 4.0 * S * M * N):
            abi.check(self.lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(W), K, abi.ptr(slab), M, N, K, S, tile, None,
                                                        _cur_stream()), "deer_gemm_bf16_nt_splitk")

    def _vresadd(self, x, slab, S, rows, C, bias=None, gamma=None, beta=None, out_bf=None, out_f32=None):
        """x += sum_s slab[s] + bias, then (optionally) LayerNorm -> bf16: closes a split-K projection of the vision tower."""
        abi.check(self.lib.deer_resadd_ln(abi.ptr(x), abi.ptr(slab), S, rows * C, None, abi.ptr(bias), abi.ptr(gamma), abi.ptr(beta),
                                          abi.ptr(out_bf) if gamma is not None else None,
                                          abi.ptr(out_f32) if gamma is not None else None, None, rows, C, EPS, None,
                                          _cur_stream()), "deer_resadd_ln")

    def _ln(self, x, gamma, beta, out_bf, rows, C, in_rstride=None, in_bstride=0, batch=1, out_rstride=None, out_bstride=0,
            x_off=0, out_off=0, out_f32=None):
        in_rstride = C if in_rstride is None else in_rstride
        out_rstride = C if out_rstride is None else out_rstride
        with self._rec("layernorm_rows", 8.0 * rows * batch * C, 6.0 * rows * batch * C):
            abi.check(self.lib.deer_layernorm_rows(abi.ptr(x, x_off), in_rstride, in_bstride, rows, batch, abi.ptr(gamma), abi.ptr(beta),
                                                   abi.ptr(out_bf, out_off), abi.ptr(out_f32), out_rstride, out_bstride, C, EPS,
                                                   _cur_stream()), "deer_layernorm_rows")

    # ------------------------------------------------------------------------------------------ vision
    VIT_HEAD_LAYERS = 3      # ViT blocks in the first graph piece of a step (short to submit; see _step_segmented)

    def enqueue_vision(self, part: str = "all", ws=None, kv_all: bool = True):
        """ViT-L/14 on both camera frames (batched, the reference runs them separately: flamingo_mpt.py:626,633),
        Perceiver on each, concat -> vis_x, then K/V of every x-attn layer.
        part: "all", or "head" (patch embedding + the first VIT_HEAD_LAYERS blocks) / "tail" (the rest)."""
        cfg, lib, st = self.cfg, self.lib, _cur_stream()
        ws = self.vws if ws is None else ws
        N, P, W = ws.n, cfg.n_patches, cfg.vit_width
        R = N * (P + 1)
        n_head = min(self.VIT_HEAD_LAYERS, len(self.vit_layers) - 1)
        lo, hi = {"all": (0, len(self.vit_layers)), "head": (0, n_head), "tail": (n_head, len(self.vit_layers))}[part]
        if part != "tail":
            self._enqueue_patch_embed(ws)
        H = cfg.vit_heads
        tok = P + 1
        So, Sp = ws.vit_split
        for li in range(lo, hi):
            L = self.vit_layers[li]
            nxt = self.vit_layers[li + 1] if li + 1 < len(self.vit_layers) else None
            self._gemm(ws.v_ln, L["wqkv"], ws.v_qkv, R, 3 * W, W, abi.EPI_BF16, bias=L["bqkv"])
            abi.check(lib.deer_attn_mfma_hd64(abi.ptr(ws.v_qkv), abi.ptr(ws.v_qkv, 2 * W), abi.ptr(ws.v_qkv, 4 * W),
                                              abi.ptr(ws.v_ao), N, H, tok, tok, 3 * W, 3 * W, 3 * W, W, tok * 3 * W, tok * 3 * W,
                                              tok * 3 * W, tok * W, 64 ** -0.5, st), "vit attn")
            self._gemm_splitk(ws.v_ao, L["wo"], ws.v_slab, R, W, W, So)
            self._vresadd(ws.vx, ws.v_slab, So, R, W, bias=L["bo"], gamma=L["ln2w"], beta=L["ln2b"], out_bf=ws.v_ln)
            self._gemm(ws.v_ln, L["wfc"], ws.v_h, R, cfg.vit_mlp, W, abi.EPI_QGELU_BF16, bias=L["bfc"])
            self._gemm_splitk(ws.v_h, L["wpr"], ws.v_slab, R, W, cfg.vit_mlp, Sp)
            if nxt is not None:
                self._vresadd(ws.vx, ws.v_slab, Sp, R, W, bias=L["bpr"], gamma=nxt["ln1w"], beta=nxt["ln1b"], out_bf=ws.v_ln)
            else:
                self._vresadd(ws.vx, ws.v_slab, Sp, R, W, bias=L["bpr"])
        if part != "head":
            self._enqueue_perceiver(ws)
            if kv_all:
                self._enqueue_media_kv()

    def _enqueue_patch_embed(self, ws):
        """conv1 as im2col + GEMM, class/positional embedding + ln_pre, ln_1 of the first block."""
        cfg, lib, st = self.cfg, self.lib, _cur_stream()
        N, P, W = ws.n, cfg.n_patches, cfg.vit_width
        R = N * (P + 1)
        abi.check(lib.deer_vit_im2col(abi.ptr(ws.img), 1, N, cfg.image_size, cfg.patch_size, abi.ptr(ws.im2col),
                                      self.patch_kpad, st), "im2col")
        self._gemm(ws.im2col, self.vit["conv"], ws.patch_out, N * P, W, self.patch_kpad, abi.EPI_F32)
        abi.check(lib.deer_vit_embed_lnpre(abi.ptr(ws.patch_out), abi.ptr(self.vit["cls"]), abi.ptr(self.vit["pos"]),
                                           abi.ptr(self.vit["ln_pre_w"]), abi.ptr(self.vit["ln_pre_b"]), abi.ptr(ws.vx), N, P, W,
                                           EPS, st), "vit_embed")
        # c_proj / out_proj run split-K (few output tiles, long K) into f32 slabs; the slab reduction, bias, residual add
        # and the NEXT LayerNorm are one launch (deer_resadd_ln), so a block is 7 launches and no projection leaves CUs idle.
        self._ln(ws.vx, self.vit_layers[0]["ln1w"], self.vit_layers[0]["ln1b"], ws.v_ln, R, W)

    def _enqueue_perceiver(self, ws):
        cfg, lib, st = self.cfg, self.lib, _cur_stream()
        N, P, W = ws.n, cfg.n_patches, cfg.vit_width
        tok = P + 1
        # ---- Perceiver (helpers.py:107-132) on the patch tokens x[:, 1:] of each camera ----
        # Media side once for all layers: one LayerNorm pass with every layer's norm_media affine, one batched GEMM with every
        # layer's to_kv.  Per layer only the 64 latents move: q|k|v projection, attention over [media K/V ; latent K/V] (two
        # segments, helpers.py:51 without the concat), to_out and the FF - each residual projection split-K, closed by the
        # reducer that also applies the NEXT LayerNorm (7 launches per layer).
        nl, inner, Lp = cfg.perc_latents, cfg.perc_heads * cfg.perc_dim_head, cfg.perc_depth
        abi.check(lib.deer_broadcast_rows(abi.ptr(self.perc["latents"]), abi.ptr(ws.p_lat), nl * W, N, st), "latents")
        with self._rec("layernorm_rows", 8.0 * N * P * W, (4.0 + 2.0 * Lp) * N * P * W):
            abi.check(lib.deer_layernorm_rows_multi(abi.ptr(ws.vx, W * 4), W, tok * W, P, N, abi.ptr(self.perc_nm_w),
                                                    abi.ptr(self.perc_nm_b), Lp, W, abi.ptr(ws.p_mln), N * P * W, W, P * W, W,
                                                    EPS, st), "norm_media (all layers)")
        with self._rec("gemm_tiled", 2.0 * Lp * N * P * 2 * inner * W, 2.0 * Lp * (N * P * W + 2 * inner * W + N * P * 2 * inner)):
            abi.check(lib.deer_gemm_bf16_nt_wbatch(abi.ptr(ws.p_mln), W, N * P * W, abi.ptr(self.perc_wkv_all), W, 2 * inner * W,
                                                   None, abi.ptr(ws.p_mkv), 2 * inner, N * P * 2 * inner, N * P, 2 * inner, W, Lp,
                                                   abi.EPI_BF16, 0, None, st), "to_kv (all layers)")
        self._ln(ws.p_lat, self.perc_layers[0]["nlw"], self.perc_layers[0]["nlb"], ws.p_latln, N * nl, W)
        Pa, Pf = ws.perc_split
        for li, L in enumerate(self.perc_layers):
            self._gemm(ws.p_latln, L["wqkv"], ws.p_qkv, N * nl, 3 * inner, W, abi.EPI_BF16)
            mkv = li * N * P * 2 * inner * 2                   # byte offset of this layer's media K | V
            abi.check(lib.deer_attn_mfma_hd64_2seg(abi.ptr(ws.p_qkv), abi.ptr(ws.p_mkv, mkv), abi.ptr(ws.p_mkv, mkv + inner * 2),
                                                   abi.ptr(ws.p_qkv, inner * 2), abi.ptr(ws.p_qkv, 2 * inner * 2), abi.ptr(ws.p_ao),
                                                   N, cfg.perc_heads, nl, P, nl, 3 * inner, 2 * inner, 3 * inner, inner,
                                                   nl * 3 * inner, P * 2 * inner, nl * 3 * inner, nl * inner,
                                                   cfg.perc_dim_head ** -0.5, st), "perc attn")
            self._gemm_splitk(ws.p_ao, L["wo"], ws.v_slab, N * nl, W, inner, Pa)
            self._vresadd(ws.p_lat, ws.v_slab, Pa, N * nl, W, gamma=L["fnw"], beta=L["fnb"], out_bf=ws.p_ln)
            self._gemm(ws.p_ln, L["w1"], ws.p_h, N * nl, cfg.perc_ff_mult * W, W, abi.EPI_GELU_BF16)
            self._gemm_splitk(ws.p_h, L["w2"], ws.v_slab, N * nl, W, cfg.perc_ff_mult * W, Pf)
            if li + 1 < Lp:
                nx = self.perc_layers[li + 1]
                self._vresadd(ws.p_lat, ws.v_slab, Pf, N * nl, W, gamma=nx["nlw"], beta=nx["nlb"], out_bf=ws.p_latln)
            else:                                              # closing perceiver.norm -> media tokens (bf16 for K/V, f32 kept)
                self._vresadd(ws.p_lat, ws.v_slab, Pf, N * nl, W, gamma=self.perc["normw"], beta=self.perc["normb"],
                              out_bf=ws.vis_x, out_f32=ws.vis_x_f32)

    def _enqueue_media_kv(self):
        """K | V of every gated x-attn layer from the media tokens of ALL frames, one GEMM (helpers.py:196-197)."""
        cfg = self.cfg
        if self.n_xattn:
            self._gemm(self.vis_x, self.wkv_all, self.kv_all, self.n_cams * cfg.perc_latents, self.n_xattn * 2 * self.xinner,
                       cfg.vit_width, abi.EPI_BF16)

    # --------------------------------------------------------------------------------------------- LLM
    def _skinny(self, Wp, N, K, T, out_slab, A=None, a_slab=None, s_in=0, a_mode=abi.A_F32, lda=None, ctl=True):
        S = self.lib.deer_skinny_splitk(T, N, K)
        mpad = 16 if T <= 16 else (32 if T <= 32 else 64)
        assert S * mpad * N <= out_slab.numel(), (S, mpad, N, out_slab.numel())
        with self._rec("gemm_skinny", 2.0 * T * N * K, 2.0 * N * K):      # algorithmic bytes = the bf16 weights, once
            abi.check(self.lib.deer_gemm_skinny(abi.ptr(A), (K if lda is None else lda), abi.ptr(a_slab), s_in,
                                                mpad * K, a_mode, abi.ptr(Wp), abi.ptr(out_slab), T, N, K, S,
                                                abi.ptr(self.ctl) if ctl else None, _cur_stream()), "deer_gemm_skinny")
        return S, mpad * N

    def _resadd(self, T, pending, gamma=None, beta=None, x_copy=None, ctl=True):
        slab, S, stride, gate = pending if pending is not None else (None, 0, 0, None)
        abi.check(self.lib.deer_resadd_ln(abi.ptr(self.x), abi.ptr(slab), S, stride, abi.ptr(gate), None, abi.ptr(gamma), abi.ptr(beta),
                                          None, abi.ptr(self.xn) if gamma is not None else None, abi.ptr(x_copy), T, self.cfg.d_model, EPS,
                                          abi.ptr(self.ctl) if ctl else None, _cur_stream()), "deer_resadd_ln")

    def enqueue_embed(self, T):
        cfg = self.cfg
        abi.check(self.lib.deer_embed_tokens(abi.ptr(self.ids), abi.ptr(self.wte), abi.ptr(self.x), abi.ptr(self.text_time), T,
                                             self.B, cfg.d_model, cfg.vocab_size, cfg.media_token_id, _cur_stream()), "embed_tokens")

    def enqueue_llm_layer(self, i, T, pending, use_mask: bool, finalize: bool, ctl=True):
        """FlamingoLayer.forward (flamingo_lm.py:46-83): gated x-attn (helpers.py:260-279) then the MPT block
         on R = n_envs*T rows.  ``pending`` = not-yet-applied residual branch of the previous op
        (split-K slabs + gate)."""
        cfg, L, st = self.cfg, self.llm_layers[i], _cur_stream()
        d, B = cfg.d_model, self.B
        R = B * T
        c = abi.ptr(self.ctl) if ctl else None
        if "xa" in L:
            X = L["xa"]
            self._resadd(R, pending, X["nw"], X["nb"], ctl=ctl)
            S, stride = self._skinny(X["wq"], self.xinner, d, R, self.slab_b, A=self.xn, ctl=ctl)
            kv_off = X["kv_index"] * 2 * self.xinner * 2          # bytes into a kv_all row
            abi.check(self.lib.deer_xattn_mfma(abi.ptr(self.slab_b), S, stride, self.xinner, abi.ptr(self.kv_all, kv_off),
                                                self.n_xattn * 2 * self.xinner, self.xinner, abi.ptr(self.text_time),
                                                cfg.n_media, abi.ptr(self.ao), 1, self.xinner, T, cfg.n_media,
                                                cfg.xattn_heads, B, cfg.xattn_dim_head ** -0.5, c, st), "deer_xattn_mfma")
            S, stride = self._skinny(X["wo"], d, self.xinner, R, self.slab_a, A=self.ao, lda=self.xinner, ctl=ctl)
            self._resadd(R, (self.slab_a, S, stride, X["ag"]), X["fnw"], X["fnb"], ctl=ctl)
            S, stride = self._skinny(X["w1"], cfg.xattn_ff_mult * d, d, R, self.slab_b, A=self.xn, ctl=ctl)
            S, stride = self._skinny(X["w2"], d, cfg.xattn_ff_mult * d, R, self.slab_a, a_slab=self.slab_b, s_in=S,
                                     a_mode=abi.A_SLABS_GELU, ctl=ctl)
            pending = (self.slab_a, S, stride, X["fg"])
        self._resadd(R, pending, L["ln1w"], L["ln1b"], ctl=ctl)
        S, stride = self._skinny(L["wqkv"], 3 * d, d, R, self.slab_b, A=self.xn, ctl=ctl)
        abi.check(self.lib.deer_mpt_attn_small(abi.ptr(self.slab_b), S, stride, d, cfg.n_heads, abi.ptr(L["qlnw"]), abi.ptr(L["klnw"]),
                                               EPS, abi.ptr(self.key_mask) if use_mask else None, float(cfg.alibi_bias_max),
                                               abi.ptr(self.qkv_ws), abi.ptr(self.ao), 1, d, T, B, c, st), "deer_mpt_attn_small")
        S, stride = self._skinny(L["wo"], d, d, R, self.slab_a, A=self.ao, lda=d, ctl=ctl)
        self._resadd(R, (self.slab_a, S, stride, None), L["ln2w"], L["ln2b"], ctl=ctl)
        S, stride = self._skinny(L["wup"], cfg.mlp_ratio * d, d, R, self.slab_b, A=self.xn, ctl=ctl)
        S, stride = self._skinny(L["wdown"], d, cfg.mlp_ratio * d, R, self.slab_a, a_slab=self.slab_b, s_in=S,
                                 a_mode=abi.A_SLABS_GELU, ctl=ctl)
        pending = (self.slab_a, S, stride, None)
        if finalize:                                            # hidden_states[i] = output of layer i (mosaic_gpt_3b.py:424-427)
            self._resadd(R, pending, None, None, x_copy=self.hidden[i], ctl=ctl)
            pending = None
        return pending

    # -------------------------------------------------------------------------------------------- head
    def enqueue_head(self, layer: int, T: int, kind: int, slot: int = -1, force: bool = False, use_ctl: bool = True,
                     feats: Optional[torch.Tensor] = None, h_prev=None, c_prev=None, shadow: bool = False,
                     no_ctl_final: bool = False):
        """One DeterministicDecoder evaluation on hidden_states[layer] (action_head.py:499-611) followed by the
        exit gate (value_net.py:120-133,277-297).  kind: PSEUDO (prev action from layer i-1, value_net.py:122-125),
        CHECK (delta <= threshold -> exit + commit LSTM state), COMMIT (static exit_id / committing call)."""
        cfg, Hd, lib, st = self.cfg, self.head, self.lib, _cur_stream()
        c = abi.ptr(self.ctl) if use_ctl else None
        H, d, B = cfg.head_hidden, cfg.d_model, self.B
        feats = self.hidden[layer] if feats is None else feats      # [B*T, d]: environment b owns rows b*T .. b*T+T-1
        h_prev = self.h_state if h_prev is None else h_prev
        c_prev = self.c_state if c_prev is None else c_prev
        abi.check(lib.deer_head_pool(abi.ptr(feats), abi.ptr(self.pooled), T, d, 0 if cfg.pooling == "max" else 1, B, c, kind, layer, st),
                  "deer_head_pool")
        for l, Lw in enumerate(Hd["lstm"]):
            if l == 0:
                src, bstride, mode, in_dim, lnw, lnb = self.pooled, d, abi.X_RAW, d, None, None
            else:
                prev = Hd["lstm"][l - 1]
                src, bstride, in_dim = self.h_tmp[l - 1], H, H
                mode, lnw, lnb = (abi.X_LN, prev["lnw"], prev["lnb"]) if cfg.lstm_layernorm else (abi.X_RAW, None, None)
            abi.check(lib.deer_head_lstm_layer(abi.ptr(src), bstride, mode, T, in_dim, abi.ptr(lnw), abi.ptr(lnb), abi.ptr(Lw["wih"]),
                                               abi.ptr(Lw["whh"]), abi.ptr(Lw["bih"]), abi.ptr(Lw["bhh"]), abi.ptr(h_prev[l]),
                                               abi.ptr(c_prev[l]), abi.ptr(self.h_tmp[l]), abi.ptr(self.c_tmp[l]), H, B, EPS, c, kind,
                                               layer, st), "deer_head_lstm_layer")
        src, in_dim, sstride = self.h_tmp[cfg.lstm_num_layers - 1], H, H
        last = Hd["lstm"][-1]
        pro, ln = (abi.PRO_LN, (last["lnw"], last["lnb"], None, None)) if cfg.lstm_layernorm else (abi.PRO_RAW, (None,) * 4)
        for fi, (Fw, dim) in enumerate(zip(Hd["fc"], cfg.mlp_hidden_dims)):
            abi.check(lib.deer_head_fc(abi.ptr(src), sstride, in_dim, pro, abi.ptr(ln[0]), abi.ptr(ln[1]), abi.ptr(ln[2]), abi.ptr(ln[3]),
                                       abi.ptr(Fw["w0"]), abi.ptr(Fw["b0"]), abi.ptr(Fw["w1"]), abi.ptr(Fw["b1"]), dim,
                                       abi.ptr(self.z_fc[fi]), B, EPS, c, kind, layer, st), "deer_head_fc")
            src, in_dim, sstride = self.z_fc[fi], dim, 2 * dim
            if cfg.mlp_layernorm:
                pro, ln = abi.PRO_GROUP_LN_RELU, (Fw["lnw0"], Fw["lnb0"], Fw["lnw1"], Fw["lnb1"])
            else:
                pro, ln = abi.PRO_GROUP_RELU, (None,) * 4
        abi.check(lib.deer_head_final(abi.ptr(src), sstride, in_dim, pro, abi.ptr(ln[0]), abi.ptr(ln[1]), abi.ptr(ln[2]), abi.ptr(ln[3]),
                                      abi.ptr(Hd["wa"]), abi.ptr(Hd["ba"]), abi.ptr(Hd["wg"]), abi.ptr(Hd["bg"]),
                                      None if no_ctl_final else abi.ptr(self.ctl), kind, layer, slot, abi.ptr(self.thresholds),
                                      1 if force else 0, self.thr_type, self.leq, abi.ptr(self.h_tmp), abi.ptr(self.c_tmp),
                                      abi.ptr(self.h_shadow if shadow else self.h_state),
                                      abi.ptr(self.c_shadow if shadow else self.c_state), cfg.lstm_num_layers, H, B,
                                      abi.ptr(self.action_dbg), EPS, st),
                  "deer_head_final")

    # ------------------------------------------------------------------------------------- step assembly
    def configure_exit(self, exit_ids: Sequence[int], max_layer: int, steps_per_stage: int = 1):
        """``ExitController.__init__`` (value_net.py:164-173): max_layer = min(max_layer-1, last exit)."""
        self.exit_ids = list(exit_ids)
        self.ctl_max_layer = min(max_layer - 1, self.exit_ids[-1])
        self.steps_per_stage = steps_per_stage
        if self._graphs:
            torch.cuda.synchronize(self.dev)                      # pieces of the last step may still be replaying
        self._graphs.clear()

    @property
    def real_num_exit(self) -> int:
        return len([x for x in self.exit_ids if x <= self.ctl_max_layer])

    def set_thresholds(self, thresholds: Sequence[float]):
        """``ExitController._set_threshold_value`` (value_net.py:178-183)."""
        assert len(thresholds) == self.real_num_exit, (len(thresholds), self.real_num_exit)
        t = torch.full((16,), 1e8, dtype=torch.float32)
        t[: len(thresholds)] = torch.tensor([float(v) for v in thresholds], dtype=torch.float32)
        self.thresholds.copy_(t)

    def reset(self):
        """Episode start: ``clear_all_exit_memory`` + controller state (eval_utils.py:252-277)."""
        # a speculative head evaluation of the last step may still be queued on the side stream; it returns at entry only
        # while that step's ALL_EXITED is set, so the control blocks / LSTM state must not be cleared underneath it
        self._drain_side_streams()
        self.h_state.zero_()
        self.c_state.zero_()
        self.ctl.zero_()
        self._shadow_on = False
        self.cur_step = 0

    def dynamic_plan(self):
        """Per layer of the dynamic step: (layer, need_pseudo, is_exit, exit slot).  mosaic_gpt_3b.py:397-443."""
        cfg, plan = self.cfg, []
        for i in range(cfg.n_layers):
            need_pseudo = ((i + 1) in self.exit_ids) and ((i + 1) - cfg.exit_interval < 0) and (i + 1) <= self.ctl_max_layer
            is_exit = (i in self.exit_ids) and i <= self.ctl_max_layer
            plan.append((i, need_pseudo, is_exit, self.exit_ids.index(i) if is_exit else -1))
            if i >= self.ctl_max_layer:
                break
        return plan

    def enqueue_dynamic_main(self, T, use_mask, i):
        """Trunk part of layer i of the dynamic step (layer 0 also embeds the tokens)."""
        _, need_pseudo, is_exit, _ = self.dynamic_plan()[i]
        if i == 0:
            self.enqueue_embed(T)
            self._pending = None
        self._pending = self.enqueue_llm_layer(i, T, self._pending, use_mask, finalize=(need_pseudo or is_exit))

    def enqueue_dynamic_heads(self, T, i, shadow: bool = False):
        """Head evaluations that read hidden_states[i]: the layer-0 pseudo action and/or the exit check."""
        _, need_pseudo, is_exit, slot = self.dynamic_plan()[i]
        if need_pseudo:
            self.enqueue_head(i, T, abi.KIND_PSEUDO)
        if is_exit:
            self.enqueue_head(i, T, abi.KIND_CHECK, slot=slot, force=(i >= self.ctl_max_layer), shadow=shadow)

    def enqueue_llm_dynamic(self, T, use_mask, shadow: bool = False):
        """MosaicGPT.forward loop with an exit controller (mosaic_gpt_3b.py:397-443), device-predicated, one stream."""
        for i, need_pseudo, is_exit, _ in self.dynamic_plan():
            self.enqueue_dynamic_main(T, use_mask, i)
            if need_pseudo or is_exit:
                self.enqueue_dynamic_heads(T, i, shadow)

    def enqueue_llm_static(self, T, use_mask, exit_id):
        """exit_id given (flamingo_mpt.py:402-411,446-461): run layers 0..exit_id, committing head call."""
        self.enqueue_embed(T)
        pending = None
        for i in range(exit_id + 1):
            pending = self.enqueue_llm_layer(i, T, pending, use_mask, finalize=True, ctl=False)
        self.enqueue_head(exit_id, T, abi.KIND_COMMIT, use_ctl=False)

    def _enqueue_front(self, part: str = "all", info=None):
        if part != "tail":
            info = self.hold_dev if info is None else info
            abi.check(self.lib.deer_ctl_begin_step(abi.ptr(self.ctl), abi.ptr(info), self.B, _cur_stream()), "ctl_begin_step")
        self.enqueue_vision(part)

    def _drain_side_streams(self):
        cur = torch.cuda.current_stream()
        cur.wait_stream(self._side_stream)
        for st in self._extra_streams:
            cur.wait_stream(st)

    def _chain_stream(self, c: int):
        """stream of vision chain c >= 2 (chain 0: caller's stream, chain 1: the side stream)"""
        while len(self._extra_streams) < c - 1:
            self._extra_streams.append(torch.cuda.Stream(device=self.dev))
        return self._extra_streams[c - 2]

    def _enqueue_chain(self, c: int, part: str):
        """Vision tower of chain c of the two-stream schedule; chain 0's first piece also resets the control blocks (from
        the pinned step info)."""
        if c == 0 and part == "head":
            abi.check(self.lib.deer_ctl_begin_step(abi.ptr(self.ctl), abi.ptr(self.step_info_pinned), self.B, _cur_stream()),
                      "ctl_begin_step")
        self.enqueue_vision(part, ws=self.vchains[c], kv_all=False)

    def _enqueue_step(self, T, use_mask, exit_id, shadow: bool = False):
        self._enqueue_front()
        if exit_id is None:
            self.enqueue_llm_dynamic(T, use_mask, shadow)
        else:
            self.enqueue_llm_static(T, use_mask, exit_id)

    # ---------------------------------------------------------------------------------------- host API
    def load_inputs(self, rgb: torch.Tensor, gripper: torch.Tensor, ids: torch.Tensor, mask: Optional[torch.Tensor] = None):
        """rgb / gripper: (B, ..., 3, S, S) one frame per environment (any singleton dims in between; B may be omitted
        when n_envs == 1); ids (B, T) int64 (right-padded to a common T); mask (B, T) bool."""
        S, B = self.cfg.image_size, self.B
        img = self.img.view(B, 2, 3, S, S)
        img[:, 0].copy_(rgb.reshape(B, 3, S, S), non_blocking=True)
        img[:, 1].copy_(gripper.reshape(B, 3, S, S), non_blocking=True)
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
            m = mask.reshape(B, T).to(torch.uint8)
            use_mask = bool((m == 0).any())
            self.key_mask[:B * T].copy_(m.reshape(-1), non_blocking=True)
        return T, use_mask

    def step(self, rgb, gripper, ids, mask=None, exit_id: Optional[int] = None, use_graph: bool = True, sync: bool = True,
             shadow: bool = False):
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
        hold = 1 if (exit_id is None and self.cur_step % self.steps_per_stage != 0) else 0
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
                with torch.cuda.graph(g):                         # capture does not execute kernels
                    self._enqueue_step(T, use_mask, exit_id, shadow)
                self._graphs[key] = g
            else:
                g.replay()
        else:
            self._enqueue_step(T, use_mask, exit_id, shadow)
        if shadow:
            self.h_state.copy_(self.h_shadow)
            self.c_state.copy_(self.c_shadow)
        self.ctl_host.copy_(self.ctl, non_blocking=True)
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
                with torch.cuda.graph(g):
                    fn()
                return g
            C = {"chain_head": [cap(lambda c=c: self._enqueue_chain(c, "head")) for c in range(len(self.vchains))],
                 "chain_tail": [cap(lambda c=c: self._enqueue_chain(c, "tail")) for c in range(len(self.vchains))],
                 "ev_in": torch.cuda.Event(), "ev_join": [torch.cuda.Event() for _ in self.vchains]}
            self._graphs["chains"] = C
        return C

    def _replay_vision_chains(self, main_st, side):
        """chain 0 on the main stream, the other chains on the side stream(s) (they start once the inputs are in place);
        the main stream continues after all of them (media tokens complete)."""
        C = self._graphs["chains"]
        nch = len(self.vchains)
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
            with torch.cuda.graph(g):
                self._enqueue_media_kv()
                self.enqueue_llm_static(T, use_mask, exit_id)
            self._graphs[key] = g
        else:
            self._replay_vision_chains(main_st, self._side_stream if self._use_side else main_st)
            g.replay()
        self.ctl_host.copy_(self.ctl, non_blocking=True)
        main_st.synchronize()
        return self.read_result()

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
            P = {"main": [], "head": {}, "ev": {}}
            self._vision_chain_graphs()
            for i, need_pseudo, is_exit, _ in plan:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if i == 0:
                        self._enqueue_media_kv()
                    self.enqueue_dynamic_main(T, use_mask, i)
                P["main"].append(g)
                if need_pseudo or is_exit:
                    gh = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gh):
                        self.enqueue_dynamic_heads(T, i)
                    P["head"][i] = gh
                    P["ev"][i] = torch.cuda.Event()
            self._graphs[key] = P
            self.ctl_host.copy_(self.ctl, non_blocking=True)     # first call: verdict through the ordinary read-back
            main_st.synchronize()
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

        self._replay_vision_chains(main_st, side)
        for i, need_pseudo, is_exit, _ in plan:
            # keep at most LOOKAHEAD trunk layers in flight beyond an undecided check
            while decided < len(exits) and exits[decided] + LOOK < i:
                done = poll(decided + 1)
                decided += 1
                if done:
                    break
            if done:
                break
            P["main"][i].replay()
            if self._trace is not None:
                self._trace.append(("main%d" % i, self._mark(main_st), time.perf_counter()))
            if need_pseudo or is_exit:
                if side is main_st:
                    P["head"][i].replay()
                else:
                    ev = P["ev"][i]
                    ev.record(main_st)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        P["head"][i].replay()
                if self._trace is not None:
                    self._trace.append(("head%d" % i, self._mark(side), time.perf_counter()))
        if not done:
            done = poll(len(exits))
        if not done:                                             # the forced exit at the last check always fires
            raise abi.DeerHipError("dynamic step finished without an exit verdict")
        self._ctl_host_np[:] = hm[W:]                            # keep the ordinary read-back buffer current (ctl_host users)
        return self.read_result()

    # ------------------------------------------------------------------------- calibration (window mode)
    def window_hidden_states(self, rgb_seq: torch.Tensor, grip_seq: torch.Tensor, ids: torch.Tensor, mask=None) -> torch.Tensor:
        """hidden_states of EVERY layer for each frame pair of a calibration window (the ``hidden_states`` the reference's
        forward hands to ``ActionValueNet(mode='generate')``, value_net.py:386).  rgb_seq / grip_seq: (W, 3, S, S); returns
        (W, n_layers, T, d) fp32 on the device.  n_envs must be 1."""
        assert self.B == 1
        W = rgb_seq.shape[0]
        last = self.cfg.n_layers - 1
        out = None
        self._drain_side_streams()
        for t in range(W):
            T, use_mask = self.load_inputs(rgb_seq[t], grip_seq[t], ids, mask)
            self.hold_dev.zero_()
            self._enqueue_step(T, use_mask, last)                 # static full depth: every layer's output is kept
            if out is None:
                out = torch.empty(W, self.cfg.n_layers, T, self.cfg.d_model, device=self.dev)
            out[t].copy_(self.hidden[:, :T])
        return out

    def _head_eval(self, feats: torch.Tensor, commit: bool) -> torch.Tensor:
        """One DeterministicDecoder step on ``feats`` (T, d) from the current LSTM state; commit=True stores the new state
        (update_hidden_state protocol, action_head.py:548-558).  Returns the device row [pose6, gripper prob, logit]."""
        self.enqueue_head(0, feats.shape[0], abi.KIND_COMMIT, use_ctl=False, feats=feats, no_ctl_final=True)
        if commit:
            self.h_state.copy_(self.h_tmp)
            self.c_state.copy_(self.c_tmp)
        return self.action_dbg[0].clone()

    def generate_values(self, hidden: torch.Tensor, rand_layers: Sequence[int], threshold_type: str = "L2") -> torch.Tensor:
        """``ActionValueNet.forward(mode='generate')`` (value_net.py:134-160) for ONE window: for the time steps seq_id in
        [W/2-1, W-1) the action of layer 0 and of every exit is predicted from [history = features of the random exit layers
        ``rand_layers[:seq_id]`` ; this layer's feature at seq_id] with the LSTM run from a zero state (window mode == the
        carried-state steps below); returns the deltas between consecutive exits, (n_exit, W - W/2) on the host."""
        W = hidden.shape[0]
        layers = [0] + list(self.exit_ids)
        self.h_state.zero_()
        self.c_state.zero_()
        acts = []
        for t in range(W - 1):
            if t >= W // 2 - 1:
                acts.append(torch.stack([self._head_eval(hidden[t, i].contiguous(), commit=False) for i in layers]))
            self._head_eval(hidden[t, int(rand_layers[t])].contiguous(), commit=True)
        a = torch.stack(acts, dim=1)[..., :6].cpu()                # (n_exit+1, W/2, 6)
        prev, last = a[:-1], a[1:]
        d = (prev - last).abs()
        if threshold_type == "mean":
            return d.mean(-1)
        if threshold_type == "L2":
            return d.pow(2).mean(-1).pow(0.5)
        if threshold_type == "max":
            return d.max(-1)[0]
        if threshold_type == "cosine":
            return 1 - torch.nn.functional.cosine_similarity(prev, last, dim=-1, eps=1e-5)
        raise NotImplementedError(threshold_type)

    @staticmethod
    def _mark(stream):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def read_result(self, src=None):
        """Decode the per-environment control blocks (int32 numpy view; default: the pinned read-back buffer)."""
        W = abi.CTL_WORDS
        ci = (self._ctl_host_np if src is None else src).reshape(self.B, W)
        cf = ci.view(np.float32)
        out = []
        for b in range(self.B):
            c, f = ci[b], cf[b]
            a = f[abi.CTL_OUT_ACTION: abi.CTL_OUT_ACTION + 8].copy()
            out.append(dict(exit_layer=int(c[abi.CTL_EXIT_LAYER]), n_evals=int(c[abi.CTL_N_EVALS]),
                            pose=torch.from_numpy(a[:6]), gripper=float(a[6]), gripper_logit=float(a[7]),
                            deltas=torch.from_numpy(f[abi.CTL_DELTAS: abi.CTL_DELTAS + 16].copy())))
        return out[0] if self.B == 1 else out

    def weight_bytes(self) -> int:
        tot = 0
        seen = set()

        def walk(o):
            nonlocal tot
            if torch.is_tensor(o):
                if o.data_ptr() not in seen:
                    seen.add(o.data_ptr())
                    tot += o.numel() * o.element_size()
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
        for o in (self.vit, self.vit_layers, self.perc, self.perc_layers, self.wte, self.llm_layers, self.wkv_all, self.head):
            walk(o)
        return tot
