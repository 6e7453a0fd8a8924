"""Closed-loop rollout harness around the engine: the reference's ``ModelWrapper`` and the sequence/rollout drivers of
``robot_flamingo/eval/eval_utils.py``, for the configuration DeeR evaluates (deterministic LSTM head, ``pad_length == -1``
step mode, ``fusion_mode == 'post'``, one action per step).

* ``ModelWrapper``                 (eval_utils.py:180-491)  obs dict + instruction -> 7-DoF action (float16 numpy, gripper in
                                    {-1, +1}), ``current_exit_layer``; same constructor arguments and ``reset()/step()`` protocol.
* ``preprocess_image`` / ``preprocess_text_calvin``          (robot_flamingo/data/data.py:898-919).
* ``DebugEnv``                     (eval_utils.py:152-175)  the reference's constant-observation stand-in for the CALVIN
                                    simulator; ``SyntheticEnv`` is a seeded variant whose frames change every step and whose
                                    sub-tasks "succeed" after a fixed number of steps, so the success/exit bookkeeping is exercised.
* ``rollout`` / ``evaluate_sequence`` / ``evaluate_policy_ddp`` (eval_utils.py:494-700): sequences sharded by rank exactly like
  the reference (``eval_sequences[rank*N/W:(rank+1)*N/W]``), results reduced with ONE packed all-reduce instead of a pickled
  ``gather_object`` (deer_vla_amd/distributed.py).

The CALVIN simulator, its task oracle and hydra configs are not part of the hot path (out of scope): the drivers
take the environment and a ``task_checker`` callable as arguments, so the real ones plug in unchanged.
"""
from __future__ import annotations

import functools
import time
from collections import Counter, deque
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import distributed as ddist

EP_LEN = 360            # eval_utils.py:44
NUM_SEQUENCES = 1000    # eval_utils.py:45 (overridden by --num_seq)


# ---------------------------------------------------------------------------------------------- preprocessing
def preprocess_image(sample, image_processor) -> torch.Tensor:
    """data.py:898-902: stack of processed frames, (n, 3, S, S).  A processor with ``on_device`` (factory.GpuImageProcessor) takes the
    raw uint8 frames of the whole list in one call and returns device tensors."""
    if getattr(image_processor, "on_device", False):
        if len(sample) == 1:                                       # the per-step case: no host-side copy of the frame
            return image_processor(np.asarray(sample[0])[None])
        return image_processor(np.stack([np.asarray(s) for s in sample]))
    return torch.cat([image_processor(s).unsqueeze(0) for s in sample], dim=0)


def preprocess_text_calvin(sample, tokenizer):
    """data.py:905-919: ``<image>{instruction}<|endofchunk|>{eos}``, right padded, max_length 32."""
    tokenizer.padding_side = "right"
    sample = [f"<image>{s.strip()}<|endofchunk|>{tokenizer.eos_token}" for s in sample]
    text = tokenizer(sample, max_length=32, padding="longest", truncation="only_first", return_tensors="pt")
    return text["input_ids"], text["attention_mask"]


# ---------------------------------------------------------------------------------------------- environments
class DebugEnv:
    """eval_utils.py:152-175 (constant frames)."""

    def get_random_obs(self):
        return {"rgb_obs": {"rgb_static": np.ones((200, 200, 3), dtype=np.uint8),
                            "rgb_gripper": np.ones((84, 84, 3), dtype=np.uint8)},
                "robot_obs": np.ones(15, dtype=np.float32)}

    def get_obs(self):
        return self.get_random_obs()

    def step(self, action):
        return self.get_random_obs(), 0.0, False, None

    def reset(self, **kwargs):
        return

    def get_info(self):
        return None


class SyntheticEnv(DebugEnv):
    """Seeded moving-texture frames; the observation depends on the step count and (weakly) on the actions applied, so
    the LSTM history, the exit layers and the actions vary over an episode.  ``info`` is the number of steps since the
    last reset - pair it with ``steps_task_checker``."""

    def __init__(self, seed: int = 0):
        self.rng = np.random.default_rng(seed)
        self.base_s = self.rng.integers(0, 255, size=(200, 200, 3), dtype=np.uint8)
        self.base_g = self.rng.integers(0, 255, size=(84, 84, 3), dtype=np.uint8)
        self.t = 0
        self.shift = 0

    def get_obs(self):
        s = np.roll(self.base_s, (self.t * 3 + self.shift) % 200, axis=1)
        g = np.roll(self.base_g, (self.t * 2 + self.shift) % 84, axis=0)
        return {"rgb_obs": {"rgb_static": s, "rgb_gripper": g}, "robot_obs": np.full(15, self.t, dtype=np.float32)}

    def step(self, action):
        self.t += 1
        self.shift = (self.shift + int(abs(float(np.asarray(action, dtype=np.float32)[0])) * 7)) % 200
        return self.get_obs(), 0.0, False, self.get_info()

    def reset(self, **kwargs):
        self.t = 0

    def get_info(self):
        return self.t


def steps_task_checker(n_steps: int) -> Callable:
    """Task oracle stand-in: a sub-task is solved once ``n_steps`` environment steps have passed since it started."""
    def check(start_info, current_info, subtask) -> bool:
        return (current_info - start_info) >= n_steps
    return check


# ---------------------------------------------------------------------------------------------- model wrapper
class ModelWrapper:
    """``ModelWrapper`` of the reference harness (eval_utils.py:180-491), DeeR configuration."""

    def __init__(self, model, tokenizer, image_processor, cast_dtype, use_diff=False, history_len=None, future_act_len=-1,
                 amp=False, exit_id=None, early_exit=False, exit_controller=None, multi_execution=1, use_action_ensemble=False):
        m = model.module
        if use_diff or m.head_type == "diffusion":
            raise NotImplementedError("diffusion head: not on the DeeR path (SURVEY §8, out of scope)")
        if use_action_ensemble and not (early_exit and exit_controller is not None
                                        and getattr(getattr(exit_controller, "module", exit_controller), "value_net", None) is not None):
            raise ValueError("use_action_ensemble needs the dynamic exit: the ensemble is ActionValueNet.get_ensemble_action() "
                             "(eval_utils.py:457-461)")
        if m.pad_length != -1:                                     # eval_utils.py:204
            raise AssertionError("Padding for multi-exit net is not implemented. It requires store feature_cache for each exit separately.")
        if m.fusion_mode in ("two_way", "vit_concat"):
            raise NotImplementedError("fusion_mode %r: the reference asserts out of it for dynamic exit (eval_utils.py:424)" % m.fusion_mode)
        if m.act_step != 1:                                        # eval_utils.py:217-220
            assert multi_execution <= m.act_step, (multi_execution, m.act_step)
            if use_action_ensemble:
                raise NotImplementedError("use_action_ensemble with multi-step action heads: the reference's harness takes the ensemble only "
                                          "when act_step == 1 (eval_utils.py:456-461)")
        self.model = model
        # this wrapper reads only the action and the exit layer: its OWN forward calls ask for host outputs (step()); the flag is
        # restored after every call, so other callers of the shared model keep eager device tensors
        self._host_outputs = hasattr(m, "host_outputs")
        self.replan = m.replan
        self.decoder_type = m.decoder_type
        self.cast_type = cast_dtype
        self.use_diff = False
        self.text_process_fn = functools.partial(preprocess_text_calvin, tokenizer=tokenizer)
        self.image_process_fn = functools.partial(preprocess_image, image_processor=image_processor)
        self.fusion_mode = m.fusion_mode
        self.amp = amp                                             # amp = fp16 autocast = the engine's default arithmetic, precision="fp16" 
        self.head_type = m.head_type
        self.exit_id = exit_id
        self.dynamic_early_exit = early_exit
        self.exit_controller = exit_controller
        self.multi_execution = multi_execution
        self.use_action_ensemble = bool(use_action_ensemble)
        self.current_exit_layer = exit_id if isinstance(exit_id, int) else -1
        self._text_cache: Tuple[Optional[str], Optional[torch.Tensor], Optional[torch.Tensor]] = (None, None, None)
        self.reset()

    def reset(self):
        """Called at the start of every sub-task rollout (eval_utils.py:252-277)."""
        hist = self.model.module.window_size
        self.img_queue, self.gripper_queue = deque(maxlen=hist), deque(maxlen=hist)
        self.mask_queue, self.text_queue = deque(maxlen=hist), deque(maxlen=hist)
        self.model.module.clear_all_exit_memory()
        if self.exit_controller is not None:
            vn = self.exit_controller.module.value_net
            if vn is not None:
                vn.hidden_state = None
                vn.history_memory = []

    def step(self, obs, goal, get_action=True):
        if not get_action:
            return None
        m = self.model.module
        image_x = self.image_process_fn([obs["rgb_obs"]["rgb_static"]]).unsqueeze(1).unsqueeze(1).to(dtype=self.cast_type)
        if goal != self._text_cache[0]:                            # one instruction per sub-task: tokenise / upload once
            ids, mask = self.text_process_fn([goal])
            self._text_cache = (goal, ids.cuda(), mask.cuda())
        _, text_x, mask = self._text_cache
        # step mode: the LSTM heads run with window_size 1 and carry their hidden state (eval_utils.py:313-320)
        window_size = m.set_all_exit_window_size(1)
        if self.exit_controller is not None and self.exit_controller.module.value_net is not None:
            self.exit_controller.module.value_net.window_size = 1
        gripper = None
        if m.use_gripper:
            gripper = self.image_process_fn([obs["rgb_obs"]["rgb_gripper"]]).unsqueeze(1).unsqueeze(1).to(dtype=self.cast_type)
        state = None
        if m.use_state or m.sep_lm_head:                            # eval_utils.py:324-332: robot_obs (15,) -> (1, 1, 1, 15) f32
            state = torch.from_numpy(np.stack([obs["robot_obs"]])).unsqueeze(1).unsqueeze(1).to(torch.float32)
        with torch.no_grad():
            image_x = image_x.cuda(non_blocking=True)
            gripper = gripper.cuda(non_blocking=True) if gripper is not None else None
            self.img_queue.append(image_x)
            self.gripper_queue.append(gripper)
            prev_ho = getattr(m, "host_outputs", False)
            if self._host_outputs:
                m.host_outputs = True
            try:
                out = self.model(vision_x=image_x, lang_x=text_x, attention_mask=mask, vision_gripper=gripper, state_tensor=state,
                                 return_feature=True, deterministic=True, exit_id=self.exit_id,
                                 dynamic_early_exit=self.dynamic_early_exit, exit_controller=self.exit_controller)
            finally:
                if self._host_outputs:
                    m.host_outputs = prev_ho
            if hasattr(out, "exit_layer"):
                self.current_exit_layer = out.exit_layer
            # eval_utils.py:454-462: [pose6, gripper > 0.5] of the last time step, gripper scaled to -1 / +1
            if m.act_step != 1:
                # eval_utils.py:466-475: the head predicted act_step actions; the first multi_execution of them are executed
                pose = out.logits[0].squeeze(0)[-1].view(m.act_step, -1)
                grip = (out.logits[1] > 0.5).to(pose.dtype).squeeze(0)[-1].view(m.act_step, -1)
                action = torch.cat([pose, grip], dim=-1)
                action[:, -1] = (action[:, -1] - 0.5) * 2
                action = action[:self.multi_execution]
            else:
                if not self.use_action_ensemble:
                    action = torch.concat((out.logits[0], (out.logits[1] > 0.5).to(out.logits[0].dtype)), dim=2).squeeze(0)[-1]
                else:                                               # mean of the last two exits' actions of this step
                    vn = self.exit_controller.module.value_net
                    pose, grip = vn.get_ensemble_action()
                    vn.reset_actions()
                    action = torch.concat((pose, (grip > 0.5).to(pose.dtype)), dim=2).squeeze(0)[-1]
                action[-1] = (action[-1] - 0.5) * 2
                action = torch.stack([action] * self.multi_execution, dim=0)
            action = action.cpu().detach().to(dtype=torch.float16).numpy()
        m.set_all_exit_window_size(window_size)
        if m.tcp_rel:
            raise NotImplementedError                               # eval_utils.py:481-482
        return action


# ---------------------------------------------------------------------------------------------- drivers
def count_success(results: Sequence[int]) -> List[float]:
    """eval_utils.py:53-60."""
    count = Counter(results)
    return [sum(count[j] for j in reversed(range(i, 6))) / len(results) for i in range(1, 6)]


def count_exit_ratio(exit_results: Sequence[int], n_layers: int) -> List[float]:
    """eval_utils.py:62-68."""
    count = Counter(exit_results)
    return [count[i] / max(len(exit_results), 1) for i in range(n_layers)]


def merge_multi_list(res):
    """eval_utils.py:47-51."""
    return [x for l in res for x in l]


def print_and_save(results, success_exit_results, fail_exit_results, step_results, success_llm_time_list, fail_llm_time_list,
                   sequences, log_dir, n_layer, epoch=None, return_data: bool = False):
    """The end-of-run report of the reference harness (eval_utils.py:71-118; pinned by tests/golden/metrics.json, which holds the
    output of the reference's own function): average successful sequence length, chain success rates, exit statistics of the
    steps of successful sub-tasks, per-task success counts.  Returns (avg_seq_len, avg exit layer + 1) like the reference
    (+ the ``data`` dict it builds when ``return_data``)."""
    print(f"Results for Epoch {epoch}:")
    steps = np.asarray(step_results)
    avg_seq_len = np.mean(results)
    chain_sr = {i + 1: sr for i, sr in enumerate(count_success(results))}
    print(f"Average successful sequence length: {avg_seq_len}")
    print("Success rates for i instructions in a row:")
    for i, sr in chain_sr.items():
        print(f"{i}: {sr * 100:.1f}%")
    ok_ratio = count_exit_ratio(success_exit_results, n_layer)
    avg_exit = np.mean(success_exit_results) + 1
    print(f"Early Exit (success tasks) | Total steps : {len(success_exit_results)} | VLM n_layer: {n_layer} | Average : {avg_exit:.1f} "
          f"| Min : {np.min(success_exit_results)+1} | Max : {np.max(success_exit_results)+1} | AVG LLM time: {np.mean(success_llm_time_list)*1000:.1f}ms")
    print(f"Total Successful steps: {np.sum(steps)} | Avg steps per successful subtask: {np.mean(steps):.1f} | Min: {np.min(steps)} | Max: {np.max(steps)}")
    print("Early exit rates for layer i in successful tasks:")
    for i, r in enumerate(ok_ratio):
        print(f"{i+1}: {r * 100:.1f}%")
    ok, bad = Counter(), Counter()
    for n_ok, (_, chain) in zip(results, sequences):
        ok.update(chain[:n_ok])
        if n_ok < len(chain):
            bad[chain[n_ok]] += 1                                  # the chain stops at its first failed sub-task
    total = ok + bad
    task_info = {}
    for task in sorted(total):
        task_info[task] = {"success": ok[task], "total": total[task]}
        print(f"{task}: {ok[task]} / {total[task]} |  SR: {ok[task] / total[task] * 100:.1f}%")
    data = {"avg_seq_len": avg_seq_len, "chain_sr": chain_sr, "task_info": task_info}
    return (avg_seq_len, avg_exit, data) if return_data else (avg_seq_len, avg_exit)


def pick_annotation(annotations, subtask: str, subtask_i: int, sequence_i: int) -> str:
    """eval_utils.py:638-644: ``new_playtable_validation`` style dict (task -> [instruction, ...]) or the enriched per-chain list
    of ``lang_annotation_cache.json`` (``diverse_inst``: annotations[sequence_i][subtask_i])."""
    ann = annotations[subtask][0] if isinstance(annotations, dict) else annotations[sequence_i][subtask_i]
    return ann.split("\n")[0]


def rollout(env, model: ModelWrapper, task_checker: Callable, subtask: str, lang_annotation: str, ep_len: int = EP_LEN):
    """One sub-task (one instruction): eval_utils.py:618-684.  Returns (success, exit_layers, n_steps, llm_times)."""
    planned_actions: List[np.ndarray] = []
    exit_layers, llm_times = [], []
    obs = env.get_obs()
    lang_annotation = lang_annotation.split("\n")[0]
    model.reset()
    start_info = env.get_info()
    step = -1
    for step in range(ep_len):
        if model.replan != -1 and step % model.replan == 0:
            model.reset()
        if model.exit_controller is not None:
            model.exit_controller.module.set_timestep(step)      # eval_utils.py:662-663
        action = model.step(obs, lang_annotation, len(planned_actions) == 0)
        exit_layers.append(model.current_exit_layer)
        llm_times.append(model.model.module.llm_inference_time)
        if len(planned_actions) == 0:
            if action.shape == (7,):
                planned_actions.append(action)
            else:
                planned_actions.extend([action[i] for i in range(action.shape[0])])
        action = planned_actions.pop(0)
        obs, _, _, current_info = env.step(action)
        if task_checker(start_info, current_info, subtask):
            return True, exit_layers, step + 1, llm_times
    return False, exit_layers, step + 1, llm_times


def evaluate_sequence(env, model: ModelWrapper, task_checker, eval_sequence: Sequence[str], annotations,
                      ep_len: int = EP_LEN, initial_state=None, sequence_i: int = -1):
    """A chain of up to five instructions; stops at the first failure (eval_utils.py:583-624).  Returns (n_ok, exit layers of the
    successful sub-tasks, of the failed one, steps per successful sub-task, llm times of successful / failed sub-tasks)."""
    env.reset() if initial_state is None else env.reset(**initial_state)
    n_ok, ok_exits, fail_exits, ok_steps, ok_llm, fail_llm = 0, [], [], [], [], []
    for subtask_i, subtask in enumerate(eval_sequence):
        success, exits, n_steps, llm = rollout(env, model, task_checker, subtask, pick_annotation(annotations, subtask, subtask_i, sequence_i), ep_len)
        if success:
            n_ok += 1
            ok_steps.append(n_steps)
            ok_exits.extend(exits)
            ok_llm.extend(llm)
        else:
            fail_exits.extend(exits)
            fail_llm.extend(llm)
            break
    return n_ok, ok_exits, fail_exits, ok_steps, ok_llm, fail_llm


def evaluate_policy_ddp(model: ModelWrapper, env, eval_sequences: Sequence[Tuple[object, Sequence[str]]],
                        annotations, task_checker, ep_len: int = EP_LEN, report: bool = False) -> Optional[dict]:
    """eval_utils.py:494-580: this rank's slice of the evaluation chains, then a reduction on rank 0.  Returns the result
    dict on rank 0 (avg successful sequence length, chain success rates, exit-layer histogram of the successful steps,
    steps/s of this job), None elsewhere.  ``annotations``: dict task -> [instruction] or the per-chain list of
    ``lang_annotation_cache.json`` (the reference's ``diverse_inst``).  ``report``: single-process runs print the reference's
    ``print_and_save`` report from the raw per-chain lists."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    mine = ddist.shard_sequences(list(eval_sequences), rank, world)
    base = rank * len(mine)
    n_layer = model.model.module.lang_encoder.config.n_layers
    results, ok_exits, fail_exits, ok_steps, ok_llm, fail_llm = [], [], [], [], [], []
    t0 = time.perf_counter()
    for local_i, (initial_state, seq) in enumerate(mine):
        n_ok, oe, fe, st, ol, fl = evaluate_sequence(env, model, task_checker, seq, annotations, ep_len, initial_state, base + local_i)
        results.append(n_ok)
        ok_exits.extend(oe)
        fail_exits.extend(fe)
        ok_steps.extend(st)
        ok_llm.extend(ol)
        fail_llm.extend(fl)
    wall = time.perf_counter() - t0
    # exit statistics over the steps of SUCCESSFUL sub-tasks, like print_and_save (eval_utils.py:83-91); the `llm_time` slot
    # carries this rank's wall time, so steps/s of the whole job = n_steps_all / (sum of walls / world)
    packed = ddist.pack_metrics(results, ok_exits, n_layer, wall)
    extra = torch.tensor([float(len(ok_exits) + len(fail_exits))], dtype=torch.float64)
    out = ddist.reduce_metrics(torch.cat([packed, extra]), n_extra=1)   # ONE all-reduce (SUM) over RCCL / gloo
    out["n_steps_all"] = int(out.pop("extra")[0])
    out["wall_s_mean"] = out.pop("llm_time") / world
    out["steps_per_s"] = out["n_steps_all"] / max(out["wall_s_mean"], 1e-9)
    out["avg_steps_per_success"] = float(np.mean(ok_steps)) if ok_steps else float("nan")   # this rank's sub-tasks
    if report and world == 1 and ok_exits:
        print_and_save(results, ok_exits, fail_exits, ok_steps, ok_llm, fail_llm, mine, None, n_layer, 0)
    return out if rank == 0 else None


# ---------------------------------------------------------------------------------------------- env batch per rank
class BatchedModelWrapper:
    """``ModelWrapper`` for ONE ENV BATCH PER RANK (BASELINE north_star): n_envs environments advance in lock step through one
    control step of the engine; every environment has its own instruction, LSTM history and exit layer.  ``step`` takes the
    observations / goals of all slots (None for an idle slot) and returns one (7,) float16 action per slot."""

    def __init__(self, model, tokenizer, image_processor, cast_dtype, exit_controller=None, exit_id=None, use_action_ensemble=False):
        m = model.module
        assert m.n_envs > 1, "build the model with n_envs > 1 (create_model_and_transforms(..., n_envs=B))"
        if use_action_ensemble and (exit_controller is None or exit_id is not None):
            raise ValueError("use_action_ensemble needs the dynamic exit (eval_utils.py:457-461)")
        if use_action_ensemble and m.act_step != 1:                # the same refusal as ModelWrapper
            raise NotImplementedError("use_action_ensemble with multi-step action heads: the reference's harness takes the ensemble only "
                                      "when act_step == 1 (eval_utils.py:456-461)")
        self.use_action_ensemble = bool(use_action_ensemble)
        self.model, self.B = model, m.n_envs
        self.cast_type = cast_dtype
        self.text_process_fn = functools.partial(preprocess_text_calvin, tokenizer=tokenizer)
        self.image_process_fn = functools.partial(preprocess_image, image_processor=image_processor)
        self.exit_controller, self.exit_id = exit_controller, exit_id
        # steps_per_stage > 1 (value_net.py:285-286, eval_calvin.py:340): the slots of an env batch are at different steps of their
        # sub-tasks, so the stage hold is per environment on the device (csrc/head.hip: CTL_HOLD of every control block) and `step` takes
        # every slot's own step index - the per-rollout `set_timestep(step)` of eval_utils.py:662-663
        self.replan = m.replan
        self.current_exit_layers = [-1] * self.B
        self._goals: List[Optional[str]] = [None] * self.B
        self._text = None
        self._blank = None

    def reset_env(self, b: int):
        """start of a sub-task in slot b (eval_utils.py:252-277 for that environment only)"""
        self.model.module.engine.reset_env(b)

    def check_instructions(self, goals):
        """Every instruction of an env batch is right-padded to the longest one and n_envs * T rows must fit the trunk's rows
        (engine.max_T = engine.MAX_ROWS // n_envs: 256 rows in the bf16 arithmetic -> the reference's full max_length = 32 at 8
        environments, data.py:905-919; 128 rows in the fp32 arithmetic).  Raises before a run starts instead of asserting in the middle of
        it."""
        max_T = self.model.module.engine.max_T
        ids, mask = self.text_process_fn(list(goals))
        worst = int(mask.sum(dim=1).max())
        if worst > max_T:
            bad = goals[int(mask.sum(dim=1).argmax())]
            rows = self.model.module.engine.MAX_ROWS
            raise ValueError(f"instruction {bad!r} tokenizes to {worst} tokens; an env batch of {self.B} takes at most {max_T} "
                             f"(n_envs * T <= {rows} trunk rows): build the model with n_envs <= {rows // worst}")

    def _frames(self, obs_list, key):
        """processed frames of all slots; an idle slot gets a blank frame of the SAME device / dtype as the live ones (a CPU fp32 blank
        cannot be stacked with the CUDA bf16 output of GpuImageProcessor: ADVICE r2)"""
        live = [self.image_process_fn([o["rgb_obs"][key]])[0] if o is not None else None for o in obs_list]
        ref = next((x for x in live if x is not None), None)
        if ref is None:
            S = self.model.module.cfg.image_size
            ref = torch.zeros(3, S, S)
        if self._blank is None or self._blank.device != ref.device or self._blank.dtype != ref.dtype:
            self._blank = torch.zeros_like(ref)
        return torch.stack([x if x is not None else self._blank for x in live])

    def step(self, obs_list, goals, env_steps=None):
        """env_steps: step index of every slot inside its sub-task (idle slots: 0); only read by a controller with steps_per_stage > 1."""
        rgb = self._frames(obs_list, "rgb_static")
        grip = self._frames(obs_list, "rgb_gripper")
        goals = [g if g is not None else "idle" for g in goals]
        if goals != self._goals:                                   # instructions change only between sub-tasks
            ids, mask = self.text_process_fn(goals)
            if ids.shape[1] > self.model.module.engine.max_T:
                self.check_instructions(goals)
            self._text = (ids.cuda(), mask.cuda())
            self._goals = list(goals)
        ids, mask = self._text
        with torch.no_grad():
            pose, g, exits = self.model.module.step_env_batch(rgb.to(self.cast_type).cuda(non_blocking=True), ids, mask,
                                                              grip.to(self.cast_type).cuda(non_blocking=True),
                                                              exit_controller=self.exit_controller, exit_id=self.exit_id,
                                                              ensemble=self.use_action_ensemble, env_steps=env_steps)
        self.current_exit_layers = exits
        A = self.model.module.act_step
        if A != 1:                                                 # eval_utils.py:466-475 per slot: (B, A, 7), the harness executes the first of them
            pose = pose.view(self.B, A, 6)
            act = torch.cat([pose, ((g > 0.5).to(pose.dtype).view(self.B, A, 1) - 0.5) * 2], dim=2)
            return act[:, 0].to(torch.float16).numpy()
        act = torch.cat([pose, ((g > 0.5).to(pose.dtype).unsqueeze(1) - 0.5) * 2], dim=1)      # eval_utils.py:454-464
        return act.to(torch.float16).numpy()


def evaluate_policy_batched(model: BatchedModelWrapper, envs: Sequence, eval_sequences, annotations, task_checker, ep_len: int = EP_LEN,
                            report: bool = False, groups: Sequence = ()) -> Optional[dict]:
    """``evaluate_policy_ddp`` with one env batch per rank: this rank's chains (same slicing, eval_utils.py:523-527) are dealt
    to the n_envs slots, which run them concurrently - per slot exactly the reference's chain / sub-task / step loop
    (eval_utils.py:583-684), per step ONE engine call for all slots.  Metrics reduce like evaluate_policy_ddp.

    ``groups``: further (BatchedModelWrapper, envs) pairs - env batches that run CONCURRENTLY with the first one, each in its own
    host thread on its own stream (build their models with ``MPTFlamingo.sibling()``: same weight arena, own workspace / LSTM state
    / controller).  All groups draw chains from the same queue.  While one batch is in its MFMA-bound vision tower another is in
    its HBM-bound trunk: four batches of eight reach ~1030 env-steps/s per MI355X against ~900 for one (bench.py `batched_groups`)."""
    import threading
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    mine = ddist.shard_sequences(list(eval_sequences), rank, world)
    base = rank * len(mine)
    n_layer = model.model.module.lang_encoder.config.n_layers
    queue = list(enumerate(mine))
    results: Dict[int, int] = {}
    ok_exits, fail_exits, ok_steps = [], [], []
    lock = threading.Lock()
    n_steps_box = [0]
    errors = []

    def worker(model, envs):
        B = model.B
        assert len(envs) == B
        slot = [None] * B                                          # per slot: dict(seq_i, chain, k, step, start, exits, n_ok)
        n_local = 0

        def start_subtask(b):
            st = slot[b]
            if st["k"] == 0:                                       # chain start: the chain's own initial condition (eval_utils.py:587-588)
                envs[b].reset() if st["init"] is None else envs[b].reset(**st["init"])
            model.reset_env(b)
            st["step"], st["exits"], st["start"] = 0, [], envs[b].get_info()
            st["goal"] = pick_annotation(annotations, st["chain"][st["k"]], st["k"], base + st["seq_i"])

        def next_chain(b):
            with lock:
                item = queue.pop(0) if queue else None
            if item is None:
                slot[b] = None
                return
            i, (initial_state, chain) = item
            slot[b] = dict(seq_i=i, chain=list(chain), k=0, n_ok=0, init=initial_state)
            start_subtask(b)

        for b in range(B):
            next_chain(b)
        while any(s is not None for s in slot):
            if model.exit_controller is not None:                  # eval_utils.py:662-663: every slot's own step index goes with the step
                model.exit_controller.module.set_timestep(0)
            obs = [envs[b].get_obs() if slot[b] is not None else None for b in range(B)]
            acts = model.step(obs, [slot[b]["goal"] if slot[b] is not None else None for b in range(B)],
                              env_steps=[slot[b]["step"] if slot[b] is not None else 0 for b in range(B)])
            for b in range(B):
                st = slot[b]
                if st is None:
                    continue
                n_local += 1
                st["exits"].append(model.current_exit_layers[b])
                st["step"] += 1
                _, _, _, info = envs[b].step(acts[b])
                done_ok = task_checker(st["start"], info, st["chain"][st["k"]])
                if done_ok or st["step"] >= ep_len:
                    if done_ok:
                        st["n_ok"] += 1
                        with lock:
                            ok_exits.extend(st["exits"])
                            ok_steps.append(st["step"])
                        st["k"] += 1
                        if st["k"] < len(st["chain"]):
                            start_subtask(b)
                            continue
                    else:
                        with lock:
                            fail_exits.extend(st["exits"])
                    with lock:
                        results[st["seq_i"]] = st["n_ok"]
                    next_chain(b)
        with lock:
            n_steps_box[0] += n_local

    members = [(model, envs)] + list(groups)
    # every instruction this rank will see must fit the env batch's rows: fail before the run, not in the middle of it
    all_goals = sorted({pick_annotation(annotations, sub, k, base + i) for i, (_, chain) in enumerate(mine) for k, sub in enumerate(chain)})
    if all_goals:
        for mw, _ in members:
            mw.check_instructions(all_goals)
    t0 = time.perf_counter()
    if len(members) == 1:
        worker(model, envs)
    else:
        def run(mw, ev, stream):
            try:
                with torch.cuda.stream(stream):
                    worker(mw, ev)
            except Exception as e:                                  # surfaced on the caller's thread below
                errors.append(e)
        threads = [threading.Thread(target=run, args=(mw, ev, torch.cuda.Stream())) for mw, ev in members]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        if errors:
            raise errors[0]
    n_steps_all = n_steps_box[0]
    B = sum(mw.B for mw, _ in members)
    wall = time.perf_counter() - t0
    res = [results[i] for i in range(len(mine))]
    packed = ddist.pack_metrics(res, ok_exits, n_layer, wall)
    extra = torch.tensor([float(n_steps_all)], dtype=torch.float64)
    out = ddist.reduce_metrics(torch.cat([packed, extra]), n_extra=1)
    out["n_steps_all"] = int(out.pop("extra")[0])
    out["wall_s_mean"] = out.pop("llm_time") / world
    out["steps_per_s"] = out["n_steps_all"] / max(out["wall_s_mean"], 1e-9)
    out["envs_per_rank"] = B
    if report and world == 1 and ok_exits:
        print_and_save(res, ok_exits, fail_exits, ok_steps, [0.0], [0.0], mine, None, n_layer, 0)
    return out if rank == 0 else None
