"""Host-side mirror of ``DeterministicDecoder`` (robot_flamingo/models/action_head.py:408-611) on top of the
engine's head kernels (csrc/head.hip).

Only the inference behaviour DeeR uses is mirrored: max/avg pooling over the text tokens (:519-520), the 4-layer
(LayerNorm-)LSTM evaluated one time step at a time with the ``update_hidden_state`` commit/stash protocol
(:548-558) and the two MLP heads (:604-605).  The weights live in the engine (``DeerEngine.head``); this object is the
stateful handle the reference's harness and ``ActionValueNet`` poke at: ``hidden_state``, ``history_memory``,
``window_size``, ``clear_hidden_state()``, ``update_hidden_state()``, ``__call__(feat, update_hidden_state=...)``.

The fast path never goes through this class (the exit gate runs on the device, see engine.enqueue_head); it exists
so that a FOREIGN exit controller written against the reference protocol keeps working (slow host loop).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _abi as abi


class DeterministicDecoder:
    def __init__(self, engine, window_size: int = 12):
        self.engine = engine
        self.window_size = window_size
        self.history_memory: list = []
        self.tmp_hidden_state = None
        self.last_action = False
        self.is_extra_exit = True

    # ---- state (kept on the device inside the engine) -------------------------------------------------------
    @property
    def hidden_state(self):
        e = self.engine
        if not bool(e.h_state.any()) and not bool(e.c_state.any()):
            return None
        return (e.h_state.clone(), e.c_state.clone())           # (L, 1, H) like the reference (action_head.py:61-63)

    @hidden_state.setter
    def hidden_state(self, value):
        e = self.engine
        if value is None:
            e.h_state.zero_()
            e.c_state.zero_()
        else:
            e.h_state.copy_(value[0].reshape(e.h_state.shape))
            e.c_state.copy_(value[1].reshape(e.c_state.shape))
        e._head_state_changed()

    def clear_hidden_state(self) -> None:
        self.hidden_state = None

    def update_hidden_state(self):
        assert self.tmp_hidden_state is not None
        e = self.engine
        e.h_state.copy_(self.tmp_hidden_state[0])
        e.c_state.copy_(self.tmp_hidden_state[1])
        e._head_state_changed()
        self.tmp_hidden_state = None

    # ---- forward ---------------------------------------------------------------------------------------------
    def __call__(self, input_feature: torch.Tensor, h_0=None, state_tensor=None, return_feature=False,
                 return_aggregate_feature=False, with_gripper_logits=False, layer_indices=None,
                 update_hidden_state: bool = True):
        e = self.engine
        if input_feature.dim() != 3 or input_feature.shape[0] != 1:
            raise NotImplementedError("native head: step mode only, input (1, T, d) (window mode is a 'next' row, SURVEY §8f.1)")
        T = input_feature.shape[1]
        feats = input_feature.reshape(T, -1).to(device=e.dev, dtype=torch.float32).contiguous()
        self.history_memory.append(None)                      # the reference appends the pooled feature (unbounded leak)
        e.enqueue_head(0, T, abi.KIND_COMMIT, use_ctl=False, feats=feats, no_ctl_final=True)
        torch.cuda.current_stream().synchronize()
        a = e.action_dbg[0].clone()
        if update_hidden_state:
            e.h_state.copy_(e.h_tmp)
            e.c_state.copy_(e.c_tmp)
            e._head_state_changed()
        else:
            self.tmp_hidden_state = (e.h_tmp.clone(), e.c_tmp.clone())
        A = e.A                                              # multi_step_action: 6 A pose + A gripper outputs (action_head.py:472-473)
        pose = a[:6 * A].view(1, 1, 6 * A)
        grip = a[6 * A:7 * A].view(1, 1, A)
        if with_gripper_logits:
            return pose, (grip, a[7 * A:8 * A].view(1, 1, A))
        if return_feature:                                    # the pooled token feature (action_head.py:519-520: max or mean over the tokens)
            pooled = feats.amax(0, keepdim=True) if e.cfg.pooling == "max" else feats.mean(0, keepdim=True)
            return pose, grip, pooled
        return pose, grip
