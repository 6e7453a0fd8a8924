#!/usr/bin/env python3
"""bench.py - action-steps/s of the DeeR-VLA early-exit forward path on MI355X (contract: see the task brief).

One "step" = one control step of one environment (what ``ModelWrapper.step`` -> ``MPTFlamingo.forward`` does in
the reference, robot_flamingo/eval/eval_utils.py:279-480): two 224x224 camera frames -> ViT-L/14 x2 -> Perceiver
x2 -> MPT-1B layers with gated x-attn until the exit criterion fires -> LSTM action head -> 7-DoF action on the
host.  Inputs are synthetic and already resident in HBM (SURVEY.md §8d protocol); weights are seeded random
tensors of the real architecture (no checkpoints/network in this environment).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Each rank owns a full model replica + its own environment stream (the path shards by environment):
no data-path collective; RCCL is used only for the barrier and the final metric reduction -> "scaling": "weak".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

EP_LEN = 360            # robot_flamingo/eval/eval_utils.py:44
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TF = 2500.0   # dense bf16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="deer_b", choices=["deer_b", "deer_s", "deer_9b", "tiny"],
                    help="deer_b: MPT-1B max_layer=12 exit_ratio 0.8 (the metric's config); deer_s: max_layer=4; tiny: reduced dims - "
                         "only for exercising the N>1 code path in tests (tests/test_distributed_cpu.py), never a result")
    ap.add_argument("--exit-ratio", type=float, default=0.8)
    ap.add_argument("--envs-per-gpu", type=int, default=1,
                    help="independent environments evaluated per control step on each GPU (one env batch per rank); "
                         "1 = the reference's one-environment-per-process latency mode")
    ap.add_argument("--batched-envs", type=int, default=8,
                    help="also report the env-batched throughput (this many environments per GPU) in the `batched` object; "
                         "0/1 disables")
    ap.add_argument("--scripted-steps", type=int, default=200,
                    help="also time this many steps of the scripted exit schedule (`scripted` object); 0 disables")
    ap.add_argument("--calib-steps", type=int, default=360,
                    help="calibration steps (one episode): deltas of every exit while the LSTM history follows a seeded random exit "
                         "layer per step - the reference's calibration protocol (flamingo_mpt.py:485-497, value_net.py:134-160)")
    ap.add_argument("--burn-in", type=int, default=120,
                    help="untimed steps from the episode start before warm-up: the timed window then sits mid-episode (LSTM in "
                         "steady state) whatever --steps is")
    ap.add_argument("--latency-reps", type=int, default=15, help="static-exit steps timed per exit for `latency_ms_by_exit` (0 disables)")
    ap.add_argument("--on-policy-steps", type=int, default=200,
                    help="timed steps of the `on_policy` leg (the real exit criterion with the calibrated thresholds, mid-episode); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU leg; 0 = pick the fastest of a measured sweep over 32 / 64 / 128 (capped at the host's cores)")
    ap.add_argument("--cpu-full-protocol", action="store_true",
                    help="CPU leg with BASELINE.md section 3's full 20 warm-up + 100 timed steps (minutes) instead of the bounded sample")
    ap.add_argument("--window-reps", type=int, default=10,
                    help="window-mode (threshold calibration) leg: windows of cfg.window_size frame pairs timed as batch rows (0 = skip)")
    ap.add_argument("--precision", choices=("fp16", "bf16", "fp32"), default="fp16",
                    help="fp16 (default): the product arithmetic on IEEE fp16 operands = the reference's evaluation arithmetic (fp32 weights under fp16 "
                         "autocast, eval_utils.py:333). bf16: the same kernels on bf16 operands (a --precision bf16 reference run; BASELINE's dtype floor). "
                         "fp32: the fp32-activation parity arithmetic (csrc/precise.hip) - a secondary figure for DESIGN.md, never the headline")
    ap.add_argument("--no-two-groups", action="store_true", help="skip the leg with several env batches in flight per GPU")
    ap.add_argument("--batched-groups", type=int, default=4,
                    help="env batches of --batched-envs environments in flight per GPU in the `batched_groups` leg (own stream and host thread each)")
    ap.add_argument("--surface-steps", type=int, default=200,
                    help="steps of the `surface` leg (ModelWrapper.step on raw uint8 frames, GPU preprocessing, numpy action); 0 disables")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--full-depth-only", type=int, default=0, metavar="K",
                    help="profiling aid: run K eager full-depth control steps (every kernel executes, no exit skipping) and "
                         "leave - the workload of the rocprofv3 --pmc passes (tools/profile_bench.sh)")
    return ap.parse_args()


# kernel class (C-ABI entry point) -> which roofline bounds it
SAME_KERNEL = {"deer_gemm_bf16_nt_splitk": "deer_gemm_bf16_nt", "deer_gemm_bf16_nt_wbatch": "deer_gemm_bf16_nt",
               "deer_attn_mfma_hd64_2seg": "deer_attn_mfma_hd64", "deer_layernorm_rows_multi": "deer_layernorm_rows"}
KERNEL_BOUND = {"deer_gemm_bf16_nt": "mfma", "deer_attn_mfma_hd64": "mfma", "deer_gemm_skinny": "hbm", "deer_gemm_skinny_hl": "hbm",
                "deer_trunk_wide_gemm": "hbm", "deer_gemm_f32_nt": "mfma"}
MFMA_PEAK_BY_CLASS = {"deer_gemm_f32_nt": 157.3}      # exact-f32 MFMA (v_mfma_f32_16x16x4_f32): 1/16 of the bf16 rate (MI355X_MICROARCH.md)


def measure_roofline(eng, cfg, frames, ids, n_pass: int = 6, traffic_key=None):
    """In-situ per-kernel timing with HIP events (on the launch stream) over full control steps of the SAME workload:
    every launch of every kernel class is bracketed by two events while the step runs eagerly behind a spin kernel
    (so the host is ahead of the GPU and brackets contain no host launch gaps).  The static full-depth schedule
    (exit at the last layer) is used so that every kernel really executes.  Returns the roofline object of the
    DOMINANT kernel class (largest share of GPU time) plus the per-class breakdown."""
    from deer_vla_amd import _abi as abi
    import ctypes
    lib = abi.lib()
    T = ids.shape[1]
    T = ids.reshape(eng.B, -1).shape[1]
    exit_id = eng.ctl_max_layer
    # event-bracket overhead, calibrated on a kernel of KNOWN duration (a 10 us spin): an empty bracket reads 4.6 us but a
    # bracket around a kernel adds only ~3.8 us (tools/event_overhead.py), and subtracting the empty-bracket figure made
    # every launch look ~0.8 us shorter than rocprofv3 reports it
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    lib.deer_spin_us(1000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for a, b in evs:
        a.record()
        lib.deer_spin_us(10, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        b.record()
    torch.cuda.synchronize()
    overhead_us = sorted(1e3 * a.elapsed_time(b) for a, b in evs)[len(evs) // 2] - 10.0
    passes = []
    for p in range(n_pass):
        rgb, grip = frames[p % len(frames)]
        eng.reset()
        eng.load_inputs(rgb, grip, ids, None)
        eng.hold_dev.fill_(0)
        torch.cuda.synchronize()
        eng.prof_begin()
        lib.deer_spin_us(12000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        eng._enqueue_step(T, False, exit_id)
        prof = eng.prof_end()
        if p == 0:
            continue                                      # first pass warms caches / clocks
        one = {}
        for name, us, fl, by in prof:
            name = SAME_KERNEL.get(name, name)                  # entry points that launch the same kernel
            d = one.setdefault(name, dict(us=0.0, n=0, flops=0.0, bytes=0.0))
            d["us"] += max(us - overhead_us, 0.0)
            d["n"] += 1
            d["flops"] += fl
            d["bytes"] += by
        passes.append(one)
    # the MEDIAN pass (by total bracketed time): one pass disturbed by a clock ramp or a neighbour on the box would otherwise
    # shift every class average (seen once: 12.5 us instead of 10.0 us per GEMM launch in one of four back-to-back runs)
    passes.sort(key=lambda one: sum(d["us"] for d in one.values()))
    agg = passes[len(passes) // 2]
    n_pass = 2                                            # the averages below divide by (n_pass - 1) passes
    total_us = sum(d["us"] for d in agg.values())
    # cross-check of the event brackets: the same full-depth step replayed as ONE graph, timed end to end
    rgb, grip = frames[0]
    for _ in range(3):
        eng.step(rgb, grip, ids, None, exit_id=exit_id, sync=False)
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(20):
        eng.step(rgb, grip, ids, None, exit_id=exit_id, sync=False)
    g1.record()
    torch.cuda.synchronize()
    graph_us = 1e3 * g0.elapsed_time(g1) / 20
    classes = {}
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        avg = d["us"] / d["n"]
        c = {"share": round(d["us"] / total_us, 4), "launches_per_step": d["n"] // (n_pass - 1), "avg_us": round(avg, 2)}
        if d["flops"]:
            c["TFLOP/s"] = round(d["flops"] / d["us"] / 1e6, 1)
        if d["bytes"]:
            c["GB/s"] = round(d["bytes"] / d["us"] / 1e3, 1)
        classes[name] = c
    dom = max((k for k in agg if k in KERNEL_BOUND), key=lambda k: agg[k]["us"])
    d = agg[dom]
    if KERNEL_BOUND[dom] == "mfma":
        achieved, peak, unit = d["flops"] / d["us"] / 1e6, MFMA_PEAK_BY_CLASS.get(dom, MFMA_PEAK_TF), "TFLOP/s"
    else:
        achieved, peak, unit = d["bytes"] / d["us"] / 1e3, HBM_PEAK_GBS, "GB/s"
    traffic, traffic_src = pmc_traffic(dom, traffic_key)
    # What actually binds the small-M GEMM class: not the matrix pipe but the L2 -> LDS operand fill.  A 64 x 64 output tile
    # pulls (64 + 64) x K x 2 B for 2 x 64 x 64 x K FLOP = 1 B per 32 FLOP through its CU, whatever the split-K; at M = 514 the tile cannot
    # grow without leaving CUs idle (432 tiles of in_proj on 256 CUs).  Reported beside the contract's MFMA fraction: the fill rate the
    # launch sustains end to end against the measured L2 bandwidth of the chip (34.5 TB/s, MI355X_MICROARCH.md).
    l2_fill = None
    if dom == "deer_gemm_bf16_nt" and traffic_key is not None and traffic_key.endswith("envs1"):
        fill_gbs = d["flops"] / 32.0 / d["us"] / 1e3
        l2_fill = {"bytes_per_flop": 1 / 32.0, "achieved_GBs": round(fill_gbs, 1), "peak_GBs": 34500.0, "frac": round(fill_gbs / 34500.0, 4),
                   "note": "whole-launch average incl. the ~3 us prologue / epilogue of a 10 us launch; inside the K loop the ring fills at 107 GB/s per CU "
                           "= 27 TB/s = 0.79 of the L2 peak (tools/ktrace_gemm.py)"}
    # the class names are the bf16 entry points (the profiler's labels); an fp16 engine launches their `_f16` twins - same kernel
    # templates, same arguments (include/deer_hip.h)
    entry = dom.replace("_bf16_", "_f16_") if getattr(eng, "precision", "") == "fp16" and "_bf16_" in dom else (dom + "_f16" if getattr(eng, "precision", "") == "fp16" else dom)
    return {"kernel": dom, "entry_point": entry, "bound": KERNEL_BOUND[dom], "l2_fill": l2_fill, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC)",
            "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(d["bytes"] / d["n"]),
            "avg_launch_us": round(d["us"] / d["n"], 2), "launches_per_step": d["n"] // (n_pass - 1),
            "event_overhead_us": round(overhead_us, 2), "gpu_us_per_full_depth_step": round(total_us / (n_pass - 1), 1),
            "graph_us_per_full_depth_step": round(graph_us, 1),
            "schedule_note": "per-kernel brackets need a serial stream: measured on the single-stream full-depth schedule (both "
                             "camera frames batched, M=514); the timed region replays the same kernels as two concurrent "
                             "per-frame chains + head evaluations on a side stream (DESIGN.md 4.1)",
            "classes": classes}


def window_leg(eng, cfg, frames, ids, reps):
    """Window mode (flamingo_mpt.py:485-497 + value_net.py:134-160,375-386): the W frame pairs of a calibration window run as BATCH ROWS
    (ViT at M = 514*G, trunk at G*T rows, every layer's output kept), then the head replays the window in sequence mode for every
    exit.  Timed: `reps` windows end to end, inputs resident in HBM.  Its own roofline: the dominant kernel class of the batched
    full-depth pass, measured in situ with the same event brackets as the step-mode roofline."""
    W = cfg.window_size
    from deer_vla_amd import _abi as abi
    T_ = ids.shape[-1]
    # frames per group: the whole window as ONE group where the engine takes it (round 5: 16 environments / 512 trunk rows per engine),
    # and the two-groups-on-two-streams form of rounds 3-4 (largest divisor of W up to 8 frame pairs); both are timed, the faster is reported
    cands = [g for g in (16, 12, 8, 7, 6, 5, 4, 3, 2, 1) if W % g == 0 and g <= abi.MAX_ENVS and g * T_ <= eng.MAX_ROWS]
    cands = [cands[0]] + [g for g in cands[1:] if g <= 8][:1]
    rgb = torch.cat([frames[i % len(frames)][0][:1] for i in range(W)])
    grip = torch.cat([frames[i % len(frames)][1][:1] for i in range(W)])
    ids1 = ids.reshape(-1, ids.shape[-1])[:1]
    gen = torch.Generator().manual_seed(7)
    rl = torch.randint(0, len(eng.exit_ids), (W,), generator=gen)
    rand_layers = torch.tensor([eng.exit_ids[int(k)] for k in rl])
    G = cands[0]
    def one(values):
        hid = eng.window_hidden_states(rgb, grip, ids1, None, group=G)
        return eng.generate_values(hid, rand_layers, group=G) if values else hid
    tried = {}
    for g in cands:
        G = g
        for _ in range(2):
            one(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(2, reps // 2)):
            one(True)
        torch.cuda.synchronize()
        tried[g] = round(1e3 * (time.perf_counter() - t0) / max(2, reps // 2), 3)
    G = min(tried, key=tried.get)
    out = {"ms_per_window_by_frames_per_group": {str(k): v for k, v in tried.items()}}
    for key, values in (("hidden_states_only", False), ("with_value_generation", True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            one(values)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[key] = {"windows_per_s": round(reps / dt, 2), "frame_pairs_per_s": round(reps * W / dt, 1), "ms_per_window": round(1e3 * dt / reps, 3)}
    w = eng.sibling(G)
    pool = [(torch.cat([frames[(p + i) % len(frames)][0][:1] for i in range(G)]), torch.cat([frames[(p + i) % len(frames)][1][:1] for i in range(G)]))
            for p in range(4)]
    ids_g = ids1.expand(G, -1).contiguous()
    w.configure_exit(eng.exit_ids, eng._max_layer_arg, 1)
    r = measure_roofline(w, cfg, pool, ids_g, traffic_key=None)      # no PMC pass at the window's row counts: traffic null
    r.pop("schedule_note", None)
    return {"window_size": W, "frames_per_group": G, "rows": {"vit": 514 * G, "trunk": int(G * ids1.shape[1])}, "reps": reps, **out, "roofline": r,
            "note": "frames of a window are batch rows of the env-batch engine (same kernels, same arena); value generation = "
                    "ActionValueNet(mode='generate') with the windows' time steps replayed through the LSTM head"}


def two_groups_leg(cfg, rb, B, steps, burn_in, local_rank, stagger=True, G=4):
    """G env batches of B environments per GPU, in flight together: one engine each over the SAME weight arena, each on its own
    stream, driven by its own host thread (the native step driver releases the GIL).  While one batch is in its MFMA-bound vision
    tower the other is in its HBM-bound trunk, so the GPU interleaves complementary work.  Same thresholds, dynamic exits, the
    host reads every batch's actions after every step - 2B environments per GPU advance, each at its own pace."""
    import threading
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.engine import DeerEngine
    e1 = rb["eng"]
    engs = [e1] + [DeerEngine(cfg, None, device=e1.dev, n_envs=B, weights_from=e1) for _ in range(G - 1)]
    ids = rb["ids"]
    pools = [rb["frames"]]
    for g in range(1, G):
        engs[g].configure_exit(e1.exit_ids, e1._max_layer_arg, 1)
        engs[g].set_thresholds(rb["thr"])
        per = []
        for s in range(len(rb["frames"])):
            pe = [syn.synthetic_step_inputs(cfg, s, rank=1000 * g + e, text_seed=7 + e) for e in range(B)]
            per.append((torch.stack([p[0] for p in pe]).to(e1.dev, e1.img_dtype), torch.stack([p[1] for p in pe]).to(e1.dev, e1.img_dtype)))
        pools.append(per)
    streams = [torch.cuda.Stream(device=e1.dev) for _ in range(G)]

    failures = []

    def episode_steps(g, lo, hi, acc, delay=0.0):
        try:
            _episode_steps(g, lo, hi, acc, delay)
        except Exception as e:                             # a worker thread must not die silently: the leg reports it
            failures.append(e)

    def _episode_steps(g, lo, hi, acc, delay=0.0):
        eng, pool = engs[g], pools[g]
        if delay:
            time.sleep(delay)                             # start half a step apart: vision of one batch beside the trunk of the other
        with torch.cuda.stream(streams[g]):
            for i in range(lo, hi):
                if i % EP_LEN == 0:
                    eng.reset()
                    eng.cur_step = 0
                r = eng.step(pool[i % len(pool)][0], pool[i % len(pool)][1], ids, None)
                acc[g] += sum(x["exit_layer"] + 1 for x in r)
    for g in range(G):                                    # graph capture and burn-in one engine at a time (capture is process-global)
        episode_steps(g, 0, burn_in, [0] * G)
    torch.cuda.synchronize()
    acc = [0] * G
    half = rb["t_max"] / max(steps, 1) / G if stagger else 0.0
    th = [threading.Thread(target=episode_steps, args=(g, burn_in, burn_in + steps, acc, g * half)) for g in range(G)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if failures:
        raise failures[0]
    n = G * B * steps
    return {"envs_per_gpu": G * B, "groups": G, "value": round(n / dt, 2), "unit": "action-steps/s", "steps_per_group": steps,
            "ms_per_env_step": round(1e3 * dt / n, 4), "avg_exit_layer": round(sum(acc) / n, 3),
            "note": "env batches in flight together over one weight arena, each on its own stream / host thread (the native step driver "
                    "releases the GIL): one batch's vision tower (MFMA-bound) overlaps another's trunk (HBM-bound)"}


def surface_leg(cfg, eng, thr, frames, ids, steps, warmup=20):
    """Steps/s through the DROP-IN SURFACE (a step = one ``ModelWrapper.step``, eval_utils.py:279-480): raw uint8 camera
    frames (200x200 static, 84x84 gripper) -> GPU preprocessing (csrc/preprocess.hip) -> ``MPTFlamingo.forward`` with the native exit
    controller -> float16 numpy action on the host, beside ``DeerEngine.step`` on the SAME frames preprocessed once (what `value` times).
    The model object shares this engine's weight arena."""
    import numpy as np
    from deer_vla_amd import rollout as ro
    from deer_vla_amd.action_head import DeterministicDecoder
    from deer_vla_amd.engine import DeerEngine
    from deer_vla_amd.factory import GpuImageProcessor, SyntheticTokenizer
    from deer_vla_amd.flamingo_mpt import MPTFlamingo
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    model = MPTFlamingo(cfg, None, window_size=cfg.window_size)
    model._engine = DeerEngine(cfg, None, device=eng.dev, n_envs=1, weights_from=eng)
    model.extra_exit = DeterministicDecoder(model._engine, cfg.window_size)
    model.lm_head = model.extra_exit
    vn = ActionValueNet(model.get_all_exit_idx(), None, cfg.exit_interval, cfg.window_size, "L2")
    ctl = ExitController(vn, model.get_all_exit_idx(), max_layer=eng._max_layer_arg)
    ctl._set_threshold_value(list(thr))
    proc = GpuImageProcessor(cfg.image_size, device=eng.dev, dtype=eng.img_dtype)
    w = ro.ModelWrapper(model, SyntheticTokenizer(cfg), proc, eng.img_dtype, early_exit=True, exit_controller=ctl)
    rng = np.random.default_rng(5)
    obs = [{"rgb_obs": {"rgb_static": rng.integers(0, 255, (200, 200, 3), dtype=np.uint8),
                        "rgb_gripper": rng.integers(0, 255, (84, 84, 3), dtype=np.uint8)}, "robot_obs": np.zeros(15, np.float32)} for _ in range(16)]
    goal = "lift the red block from the sliding cabinet"
    n = warmup + steps

    def run_surface():
        w.reset()
        exits = 0
        for i in range(n):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ctl.set_timestep(i)
            a = w.step(obs[i % len(obs)], goal)
            if i >= warmup:
                exits += w.current_exit_layer + 1
        torch.cuda.synchronize()
        assert a.shape == (7,) or a.shape == (1, 7)
        return steps / (time.perf_counter() - t0), exits / steps

    def run_engine():
        e = model.engine
        pre = [(proc(o["rgb_obs"]["rgb_static"]), proc(o["rgb_obs"]["rgb_gripper"])) for o in obs]      # preprocessed once, resident in HBM
        ids_t, _ = ro.preprocess_text_calvin([goal], SyntheticTokenizer(cfg))
        ids_t = ids_t.to(e.dev)
        e.configure_exit(ctl.exit_id_list, eng._max_layer_arg, 1)
        e.set_thresholds(list(thr))
        e.reset()
        e.cur_step = 0
        exits = 0
        for i in range(n):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            r = e.step(pre[i % len(pre)][0], pre[i % len(pre)][1], ids_t, None)
            if i >= warmup:
                exits += r["exit_layer"] + 1
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0), exits / steps

    def run_forward():
        """MPTFlamingo.forward alone (native controller, host_outputs) on the frames preprocessed once"""
        pre = [(proc(o["rgb_obs"]["rgb_static"]).unsqueeze(1).unsqueeze(1), proc(o["rgb_obs"]["rgb_gripper"]).unsqueeze(1).unsqueeze(1)) for o in obs]
        ids_t, mask_t = ro.preprocess_text_calvin([goal], SyntheticTokenizer(cfg))
        ids_t, mask_t = ids_t.to(eng.dev), mask_t.to(eng.dev)
        model.clear_all_exit_memory()
        exits = 0
        model.host_outputs = True                                  # what ModelWrapper.step asks for on its own calls
        try:
            for i in range(n):
                if i == warmup:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                ctl.set_timestep(i)
                o = model(vision_x=pre[i % len(pre)][0], lang_x=ids_t, attention_mask=mask_t, vision_gripper=pre[i % len(pre)][1],
                          dynamic_early_exit=True, exit_controller=ctl)
                if i >= warmup:
                    exits += o.exit_layer + 1
        finally:
            model.host_outputs = False
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0), exits / steps

    run_surface()                                                  # captures the graphs
    v_s, x_s = run_surface()
    v_f, x_f = run_forward()
    v_e, x_e = run_engine()
    return {"value": round(v_s, 2), "unit": "action-steps/s", "steps": steps, "avg_exit_layer": round(x_s, 3),
            "forward_same_frames": {"value": round(v_f, 2), "avg_exit_layer": round(x_f, 3)},
            "engine_step_same_frames": {"value": round(v_e, 2), "avg_exit_layer": round(x_e, 3)},
            "forward_over_engine": round(v_f / v_e, 4), "surface_over_engine": round(v_s / v_e, 4),
            "note": "ModelWrapper.step: raw uint8 frames -> deer_preprocess_frames (2 launches per camera) -> MPTFlamingo.forward "
                    "(native ExitController, host_outputs) -> float16 numpy action; forward_same_frames = MPTFlamingo.forward alone and "
                    "engine_step_same_frames = DeerEngine.step (the path `value` times) on the same frames preprocessed once; the "
                    "surface's extra time is the per-step preprocessing (~60 us: two uploads + four launches) and the wrapper's host code"}


def lib_hash():
    """sha256 (first 16 hex) of the HIP library the numbers were measured with"""
    import hashlib
    from deer_vla_amd import _abi as abi
    with open(abi.LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def pmc_traffic(kernel_class, workload_key):
    """HBM-side bytes per launch of a kernel class from the committed rocprofv3 --pmc summary (FETCH_SIZE and WRITE_SIZE
    are collected in separate passes by tools/profile_bench.sh over full-depth steps; bench.py cannot run PMC passes on
    itself).  The summary is keyed by WORKLOAD ("<workload>/envs<B>": the launches of a class move different bytes at
    another model size or row count) - a workload without its own pass gets None, never another shape's
    figure - and stamped with the hash of the kernel SOURCES it was collected on (csrc/*.hip, common.h): a stale stamp
    is reported, and the figure is then withheld (None)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or workload_key is None:
        return None, None
    with open(path) as fh:
        t = json.load(fh)
    wl = t.get("workloads", {}).get(workload_key)
    if not wl:
        return None, "no PMC pass of workload %s in profiles/pmc_traffic.json" % workload_key
    c = wl.get("classes", {}).get(kernel_class)
    if not c:
        return None, None
    stamp, now = wl.get("kernel_source_hash"), kernel_source_hash()
    if stamp != now:
        return None, "profiles/pmc_traffic.json is STALE (collected on kernel sources %s, now %s): re-run tools/profile_bench.sh" % (stamp, now)
    return c["hbm_bytes_per_launch"], "profiles/pmc_traffic.json[%s] (%s)" % (workload_key, t.get("note", ""))


def kernel_source_hash():
    """hash of the DEVICE code: csrc/*.h and every csrc/*.hip that defines a kernel (host-only files, e.g. the native step driver, do
    not change what a PMC pass measures), with `//` comments and blank lines stripped - rewording a comment does not make a profile
    stale, changing an instruction does"""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "deer_vla_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                src = fh.read()
            if f.endswith(".h") or b"__global__" in src:
                code = []
                for line in src.decode("utf-8", "replace").splitlines():
                    line = re.sub(r"//.*$", "", line).rstrip()
                    if line.strip():
                        code.append(line)
                h.update(f.encode() + b"\n" + "\n".join(code).encode())
    return h.hexdigest()[:16]


def target_schedule(n_exits, exit_ratio, n, seed=99):
    """Exit SLOTS (indices into the thresholded exits + the forced one) of n steps at the calibration target p_k ~ exit_ratio^k
    (value_net.py:216-217,238), STRATIFIED: slot k appears round(n * p_k) times (largest remainders), in a seeded shuffle - so the mix,
    and with it the average depth, is the target's for every n (20 steps: 5/4/4/3/2/2 -> 5.9 layers vs E = 5.74) instead of whatever n
    independent draws happen to give."""
    pk = [float(exit_ratio) ** k for k in range(1, n_exits + 1)]
    tot = sum(pk)
    want = [n * p / tot for p in pk]
    cnt = [int(w) for w in want]
    for k in sorted(range(n_exits), key=lambda k: -(want[k] - cnt[k]))[: n - sum(cnt)]:
        cnt[k] += 1
    slots = [k for k in range(n_exits) for _ in range(cnt[k])]
    perm = torch.randperm(len(slots), generator=torch.Generator().manual_seed(seed)).tolist()
    return [slots[i] for i in perm]


def forced_thresholds(slot, real):
    """thresholds that make the exit check of `slot` the first one to fire: the checks before it run and decline (delta > -1), this
    one accepts (delta <= 1e8) - the dynamic path does all of its work, the verdict is scripted"""
    return [-1.0] * slot + [1e8] * (real - slot)


def cpu_baseline(cfg, sd, ctl, slots, budget_s, threads, rank, full_protocol=False):
    """The CPU oracle (oracle/deer_oracle.py = pure-PyTorch fp32 restatement of the reference forward, pinned against
    the reference's own modules) timed on this box's host cores on a bounded sample of the SAME workload: same synthetic inputs and
    weights, the same dynamic-exit protocol, and the same scripted exit mix as the GPU leg (`slots`: stratified to the calibration
    target, so the CPU figure is MEASURED at the GPU leg's depth).  Thread count: the fastest of a measured sweep (BASELINE.md
    section 3 says "all host cores"; with all 256 logical cores PyTorch's intra-op pool oversubscribes - the sweep is in the JSON)."""
    from deer_vla_amd import synthetic as syn
    from oracle import deer_oracle as orc
    model = orc.OracleDeer(sd, cfg)
    model.set_all_exit_window_size(1)
    vn = orc.OracleValueNet(ctl.exit_id_list, model.extra_exit, cfg.exit_interval, 1, "L2")
    oc = orc.OracleExitController(vn, ctl.exit_id_list, steps_per_stage=1, max_layer=ctl.max_layer + 1)
    real = ctl.real_num_exit

    def one(s, slot):
        rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s, rank=rank)
        oc._set_threshold_value(forced_thresholds(slot, real))
        oc.set_timestep(s)
        o = model.forward(rgb, ids, mask, grip, dynamic_early_exit=True, exit_controller=oc)
        return o["exit_layer"]

    ncpu = os.cpu_count() or 1
    sweep = {}
    with torch.no_grad():
        cands = [threads] if threads > 0 else sorted({min(c, ncpu) for c in (32, 64, 128)})
        mid = real // 2
        for c in cands:
            torch.set_num_threads(c)
            one(0, mid)                                        # warm-up of this pool size
            t0 = time.perf_counter()
            one(1, mid)
            one(2, mid)
            sweep[c] = round(2 / (time.perf_counter() - t0), 4)
        cores = max(sweep, key=sweep.get)
        torch.set_num_threads(cores)
        warm = 1.0 / sweep[cores]
        # per-stage milliseconds: vision tower, one LLM layer (x-attn + block), one head evaluation
        rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 2, rank=rank)
        t0 = time.perf_counter()
        vis = model.encode_vision(rgb, grip)
        t_vis = time.perf_counter() - t0
        t0 = time.perf_counter()
        hid, _ = orc.llm_forward(sd, cfg, ids, mask.bool(), vis, exit_id=cfg.n_layers - 1)
        t_layer = (time.perf_counter() - t0) / cfg.n_layers
        t0 = time.perf_counter()
        model.extra_exit(hid[0], update_hidden_state=False)
        t_head = time.perf_counter() - t0
        model.clear_all_exit_memory()
        n_warm, n_steps = (20, 100) if full_protocol else (2, int(max(6, min(100, budget_s / max(warm, 1e-3)))))
        sched = target_schedule(real, _EXIT_RATIO[0], n_steps, seed=99)
        for s in range(n_warm):
            one(s, sched[s % len(sched)])
        exits = []
        t0 = time.perf_counter()
        for s in range(n_steps):
            exits.append(one(n_warm + s, sched[s]) + 1)
        dt = time.perf_counter() - t0
    return {"value": round(n_steps / dt, 4), "unit": "action-steps/s", "cores": cores, "kind": "port",
            "host_logical_cores": ncpu, "thread_sweep_steps_per_s": {str(k): v for k, v in sweep.items()},
            "per_stage_ms": {"vision_tower_2xViT_2xPerceiver": round(1e3 * t_vis, 1), "llm_layer": round(1e3 * t_layer, 2),
                             "head_evaluation": round(1e3 * t_head, 2)},
            "avg_exit_layer": round(sum(exits) / len(exits), 2),
            "sample": f"{n_warm} warm-up + {n_steps} timed control steps of the same workload at the GPU leg's scripted exit mix (fp32 oracle, "
                      f"dynamic-exit protocol, torch.set_num_threads({cores}) on a {ncpu}-logical-core host), {dt:.1f} s",
            "protocol": "BASELINE.md section 3 (20 + 100 steps)" if full_protocol else
                        "bounded sample (task contract: ~10-30 s of CPU work); --cpu-full-protocol runs BASELINE.md section 3's 20 + 100 "
                        "steps (profiles/ holds one such run per round)"}


_EXIT_RATIO = [0.8]     # set by main() from --exit-ratio (read by cpu_baseline's schedule)


def run_workload(args, cfg, sd_dev, B, rank, world, local_rank, dist, max_layer, timed_steps, warmup):
    """Build an engine for B environments per GPU, calibrate thresholds on-policy for --exit-ratio, and time
    `timed_steps` control steps (contract: barrier + synchronize on both sides, MAX over ranks)."""
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.engine import DeerEngine
    from deer_vla_amd.value_net import ExitController

    t0 = time.time()
    eng = DeerEngine(cfg, sd_dev, device=f"cuda:{local_rank}", n_envs=B, precision=args.precision)
    ctl = ExitController(None, cfg.exit_ids(), steps_per_stage=1, max_layer=max_layer)
    eng.configure_exit(ctl.exit_id_list, max_layer, 1)
    setup_s = time.time() - t0

    # ---- synthetic inputs, resident in HBM (bf16 frames) ----
    POOL = 32
    dev = eng.dev
    frames = []
    for s in range(POOL):
        per_env = [syn.synthetic_step_inputs(cfg, s, rank=rank * B + e, text_seed=7 + e) for e in range(B)]
        frames.append((torch.stack([p[0] for p in per_env]).to(dev, eng.img_dtype),      # resident in the engine's own frame format
                       torch.stack([p[1] for p in per_env]).to(dev, eng.img_dtype)))
    ids = torch.cat([p[2] for p in per_env]).to(dev)          # (B, T): one instruction per environment
    T = ids.shape[1]

    def run_step(i, use_graph=True, sync=True, shadow=False):
        if i % EP_LEN == 0:
            eng.reset()                                   # new episode: LSTM / controller state cleared
            eng.cur_step = 0
        rgb, grip = frames[i % POOL]
        return eng.step(rgb, grip, ids, None, use_graph=use_graph and not args.no_graph, sync=sync, shadow=shadow)

    if args.full_depth_only:
        for p in range(args.full_depth_only):
            rgb, grip = frames[p % POOL]
            eng.reset()
            eng.load_inputs(rgb, grip, ids, None)
            eng.hold_dev.fill_(0)
            eng._enqueue_step(T, False, eng.ctl_max_layer)
            torch.cuda.synchronize()
        return None

    # ---- threshold calibration for --exit-ratio (value_net.py:185-264 solver) -------------------------------
    # The reference's protocol, deterministic (no fixed point): the delta of EVERY exit is recorded at every step of one
    # episode (shadow mode) while the LSTM history follows a seeded random exit layer per step (flamingo_mpt.py:485-497 feeds
    # the head features of a random exit layer as history); `solve_thresholds` then turns the (n_exit, n_samples) matrix into
    # thresholds for the target distribution p_k ~ exit_ratio^k.  The committed oracle traces (tests/golden/episode_full.npz)
    # are made the same way.  In shadow mode the state commits at the first exit whose criterion fires: thresholds
    # [-1]*k + [1e8] make that exit k.
    real = ctl.real_num_exit
    gen = torch.Generator().manual_seed(4242)
    vals = []
    eng.reset()
    eng.cur_step = 0
    for i in range(args.calib_steps):
        k = int(torch.randint(0, real, (1,), generator=gen))
        eng.set_thresholds([-1.0] * k + [1e8] * (real - k))
        rgb, grip = frames[i % POOL]
        r = eng.step(rgb, grip, ids, None, shadow=True)
        for re in (r if B > 1 else [r]):
            vals.append(re["deltas"][:real].clone())
    values = torch.stack(vals, dim=1)                      # (n_exit, n_samples)
    ctl.set_threshold_from_values(values, args.exit_ratio, cfg.llm_name)
    thr = ctl.threshold_list()
    eng.set_thresholds(thr)

    real_n = ctl.real_num_exit

    def timed(n_steps, pre, slots=None):
        """`pre` untimed + `n_steps` timed control steps (contract: barrier + synchronize on both sides); slots: scripted verdicts
        (one exit slot per step, thresholds forced on the device before the step) or None = the calibrated thresholds (on-policy)."""
        thr_rows = None
        if slots is not None:
            thr_rows = torch.full((real_n, 16), 1e8, dtype=torch.float32)
            for k in range(real_n):
                thr_rows[k, :real_n] = torch.tensor(forced_thresholds(k, real_n))
            thr_rows = thr_rows.to(dev)
        else:
            eng.set_thresholds(thr)

        def one(i):
            if thr_rows is not None:
                eng.set_thresholds_device(thr_rows[slots[i % len(slots)]])
            return run_step(i)
        for i in range(pre):
            one(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        exit_sum, hist = 0, [0] * cfg.n_layers
        for i in range(pre, pre + n_steps):
            r = one(i)
            for re in (r if B > 1 else [r]):
                exit_sum += re["exit_layer"] + 1
                hist[re["exit_layer"]] += 1
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, exit_sum, hist

    # ---- on-policy leg (secondary): the REAL criterion with the calibrated thresholds, mid-episode (burn-in first: 20 steps right after
    # a reset averaged exit layer 9.8, 300 steps 4.45 in round 1).  Its depth is whatever the synthetic episode's own deltas give - with
    # random weights they are not the calibration deltas - which is why it is not the headline (318 <-> 357 <-> 387
    # steps/s between runs of the same code with the 4th digit of a threshold).
    on_policy = None
    n_on = args.on_policy_steps if B == 1 else timed_steps
    if n_on > 0:
        dt, xs, hist = timed(n_on, args.burn_in + warmup)
        on_policy = {"value": round(world * n_on * B / dt, 2), "unit": "action-steps/s", "steps": n_on, "avg_exit_layer": round(xs / (n_on * B), 3),
                     "exit_hist": hist, "thresholds": [round(x, 6) for x in thr],
                     "note": "dynamic exits decided by the calibrated thresholds on the synthetic episode's own deltas"}
        if dist is not None:
            # N > 1: the on-policy leg over ALL ranks - slowest rank's time, every rank's steps and exit layers -
            # so that the N-GPU line can be set beside the 1-GPU `on_policy` object; the per-rank min / max rate next to it
            ops = torch.tensor([dt, float(xs), float(n_on * B)], dtype=torch.float64,
                               device=dev if dist.get_backend() == "nccl" else "cpu")
            omax, omin = ops[:1].clone(), ops[:1].clone()
            dist.all_reduce(omax, op=dist.ReduceOp.MAX)
            dist.all_reduce(omin, op=dist.ReduceOp.MIN)
            dist.all_reduce(ops, op=dist.ReduceOp.SUM)
            on_policy["value"] = round(float(ops[2]) / float(omax[0]), 2)
            on_policy["avg_exit_layer"] = round(float(ops[1]) / float(ops[2]), 3)
            on_policy["per_rank_steps_per_s"] = {"min": round(n_on * B / float(omax[0]), 2), "max": round(n_on * B / float(omin[0]), 2)}
            on_policy["exit_hist_rank0"] = on_policy.pop("exit_hist")
    if B > 1:
        # env batches: every environment exits at its own layer by the real criterion (thresholds are shared by the batch, so a
        # scripted verdict would make all environments leave together) - this leg IS the on-policy one
        elapsed, exit_sum, hist = dt, xs, hist
        n_timed = n_on
    else:
        # ---- the timed region of `value`: `timed_steps` steps of the DYNAMIC pipeline (pseudo action, every exit check up to the exit,
        # device-side verdict, host-fed graph pieces) with the verdicts scripted to the calibration target mix, stratified - the depth
        # of the timed window is the target's whatever --steps is
        slots = target_schedule(real_n, args.exit_ratio, timed_steps, seed=99)
        eng.reset()
        eng.cur_step = 0
        elapsed, exit_sum, hist = timed(timed_steps, warmup, slots)
        eng.set_thresholds(thr)
        n_timed = timed_steps

    stats = torch.tensor([elapsed, float(exit_sum), float(n_timed * B)], dtype=torch.float64,
                         device=dev if (dist is None or dist.get_backend() == "nccl") else "cpu")
    rank_rate = None
    if dist is not None:
        tmax = stats[:1].clone()
        tmin = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)         # one tiny RCCL all-reduce: the only exchange of the path
        stats[0] = tmax[0]
        # every rank's own rate over its own timed region (stragglers show up on the first real N > 1 run: VERDICT r4 next-7)
        per_rank = n_timed * B
        rank_rate = {"min": round(per_rank / float(tmax[0]), 2), "max": round(per_rank / float(tmin[0]), 2), "unit": "action-steps/s per rank",
                     "slowest_over_fastest_time": round(float(tmax[0]) / max(float(tmin[0]), 1e-12), 4)}
    t_max, exits, n_steps = float(stats[0]), float(stats[1]), float(stats[2])
    return dict(eng=eng, ctl=ctl, frames=frames, ids=ids, T=T, t_max=t_max, value=n_steps / t_max, avg_exit=exits / n_steps, thr=thr,
                hist=hist, setup_s=setup_s, on_policy=on_policy, n_timed=n_timed, rank_rate=rank_rate)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    if os.environ.get("DEER_BENCH_SINGLE_DEVICE") == "1":   # test hook: all ranks on GPU 0 (with DEER_BENCH_BACKEND=gloo) to
        local_rank = 0                                        # exercise the N>1 code path on a one-GPU box
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DEER_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_3b

    max_layer = 4 if args.workload == "deer_s" else 12
    if args.workload == "tiny":
        from deer_vla_amd.config import deer_tiny
        cfg = deer_tiny()
        max_layer = cfg.early_exit_layer + 1
    elif args.workload == "deer_9b":                             # BASELINE configs[4]: OpenFlamingo-9B / MPT-7B trunk, max_layer 12
        from deer_vla_amd.config import deer_9b
        cfg = deer_9b(max_layer=max_layer)
    else:
        cfg = deer_3b(max_layer=max_layer)
    sd = syn.make_synthetic_state(cfg, 0, bf16_round=True) if args.workload == "tiny" else syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
    B = args.envs_per_gpu
    _EXIT_RATIO[0] = args.exit_ratio
    res = run_workload(args, cfg, sd, B, rank, world, local_rank, dist, max_layer, args.steps, args.warmup)
    if res is None:
        print(json.dumps({"full_depth_steps": args.full_depth_only}))
        return
    eng, ctl, T = res["eng"], res["ctl"], res["T"]
    eng_ctl_max = eng.ctl_max_layer
    t_max, value = res["t_max"], res["value"]

    out = {
        "metric": "action-steps/sec (whole job) + avg exit-layer, %s max_layer=%d, synthetic CALVIN-D-shaped inputs"
                  % ("MPT-7B" if args.workload == "deer_9b" else ("REDUCED-DIMS TEST MODEL (not a result)" if args.workload == "tiny" else "MPT-1B"), max_layer),
        "value": round(value, 2), "unit": "action-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * t_max / res["n_timed"], 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision if args.precision != "fp32" else "f32 activations, bf16-representable weights",
        "dtype_note": ("16-bit MFMA operands (%s weights, %s results in the vision tower, %s hi + lo activation planes in the trunk), f32 accumulation / "
                       "LayerNorm / softmax / LSTM state; fp16 = the reference's evaluation arithmetic (fp32 weights under fp16 autocast, eval_utils.py:333), "
                       "bf16 = a --precision bf16 reference run" % ((args.precision,) * 3)) if args.precision != "fp32" else "csrc/precise.hip",
        "data": "synthetic",
        "avg_exit_layer": round(res["avg_exit"], 3),
        # which verdicts the timed region of `value` ran on: "scripted" = the dynamic pipeline with the thresholds forced per step
        # to the stratified target mix (one environment: depth-stable); "on_policy" = the calibrated criterion on the episode's own deltas
        # (env batches: a scripted verdict would make all environments leave together).  The other flavour is the `on_policy` object.
        "verdicts": "scripted" if B == 1 else "on_policy",
        "config": {"workload": "%s DeeR-%s max_layer=%d exit_ratio=%.2f, step mode, %d env(s)/GPU per control "
                               "step, 2x224x224 frames + %d text tokens per env, LSTM history carried over %d-step episodes"
                               % ("OpenFlamingo-9B/MPT-7B" if args.workload == "deer_9b" else "OpenFlamingo-3B/MPT-1B",
                                  "B" if max_layer == 12 else "S", max_layer, args.exit_ratio, B, T, EP_LEN),
                   "avg_exit_layer": round(res["avg_exit"], 3),
                   "exit_schedule": ("dynamic pipeline (pseudo action + every exit check up to the exit, device-side verdict), verdicts "
                                     "scripted to the calibration target p_k ~ %.2f^k, stratified over the timed steps (depth-stable for "
                                     "every --steps); the real criterion on the synthetic episode is the `on_policy` object" % args.exit_ratio)
                                    if B == 1 else "real criterion, calibrated thresholds, every environment exits at its own layer",
                   "envs_per_gpu": B, "ms_per_env_step": round(1e3 * t_max / (res["n_timed"] * B), 4),
                   "exit_hist": res["hist"] if world == 1 else None, "per_gpu_steps_per_s": round(value / world, 2),
                   "graph": not args.no_graph, "weights_gb": round(eng.weight_bytes() / 1e9, 3),
                   "thresholds": [round(x, 6) for x in ctl.threshold_list()], "setup_s": round(res["setup_s"], 1),
                   "calibration": "reference protocol, deterministic: %d shadow steps, LSTM history on a seeded random exit layer per "
                                  "step, thresholds from solve_thresholds(exit_ratio=%.2f) (value_net.py:203-260)" % (args.calib_steps, args.exit_ratio),
                   "timed_window": "%d warm-up steps, then %d timed steps of %d-step episodes" % (args.warmup, res["n_timed"], EP_LEN)},
        "lib_sha256_16": lib_hash(), "kernel_source_hash": kernel_source_hash(),
    }
    if res.get("on_policy") is not None and B == 1:
        out["on_policy"] = res["on_policy"]
    # realised vs target exit distribution (p_k ~ exit_ratio^k over the thresholded exits + the forced one)
    if world == 1:
        xs_ = [e for e in cfg.exit_ids() if e <= eng.ctl_max_layer]
        pk_ = [args.exit_ratio ** k for k in range(1, len(xs_) + 1)]
        tot_ = max(sum(res["hist"]), 1)
        out["exit_distribution"] = {"exit_layers": xs_, "target": [round(p / sum(pk_), 3) for p in pk_],
                                    "realised": [round(res["hist"][e] / tot_, 3) for e in xs_],
                                    "target_avg_layers": round(sum((e + 1) * p for e, p in zip(xs_, pk_)) / sum(pk_), 2)}
    if world > 1:   # lets the driver verify that RCCL really saw N ranks on N distinct devices
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        uuid = str(getattr(props, "uuid", "")) or str(getattr(props, "pci_bus_id", ""))
        ub = uuid.encode()[:40].ljust(40, b" ")
        info = torch.tensor([rank, local_rank, torch.cuda.current_device()] + list(ub), dtype=torch.int64,
                            device=eng.dev if dist.get_backend() == "nccl" else "cpu")
        allinfo = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(allinfo, info)
        out["rccl_world"] = dist.get_world_size()
        out["backend"] = dist.get_backend()                  # the OBSERVED backend of the process group the reductions ran on
        out["ranks"] = [{"rank": int(t[0]), "local_rank": int(t[1]), "device": int(t[2]),
                         "device_uuid": bytes(int(x) for x in t[3:]).decode(errors="replace").strip()} for t in allinfo]
        out["distinct_devices"] = len({r["device_uuid"] or r["device"] for r in out["ranks"]})
        out["per_rank_steps_per_s"] = res.get("rank_rate")
    # step latency at a KNOWN depth: static exit at every exit layer (median of --latency-reps steps, action read on the host)
    if args.latency_reps > 0 and B == 1 and rank == 0:
        frames_, ids_l = res["frames"], res["ids"]
        lat = {}
        for e in [x for x in cfg.exit_ids() if x <= eng.ctl_max_layer]:
            for _ in range(3):
                eng.step(frames_[0][0], frames_[0][1], ids_l, None, exit_id=e)
            ts = []
            for i in range(args.latency_reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.step(frames_[i % len(frames_)][0], frames_[i % len(frames_)][1], ids_l, None, exit_id=e)
                ts.append(time.perf_counter() - t0)
            lat[str(e)] = round(1e3 * sorted(ts)[len(ts) // 2], 4)
        out["latency_ms_by_exit"] = lat
    # whole-step rooflines (the step is HBM-bound overall at one environment per GPU; both fractions reported):
    # F(e) = 347.1 + 2.68 e + 0.082 n_head GFLOP and Bytes(e) = 0.814 + 0.174 e GB for e = avg number of trunk layers run
    e_avg = res["avg_exit"]
    xs = [e for e in cfg.exit_ids() if e <= eng.ctl_max_layer]
    n_steps_h = max(sum(res["hist"]), 1)
    n_head = 2.0 + (sum((xs.index(l) + 1) * h for l, h in enumerate(res["hist"]) if h and l in xs) / n_steps_h if world == 1 else 2.0)
    gflop = 347.1 + 2.68 * e_avg + 0.082 * n_head
    gbyte = 0.814 + 0.174 * e_avg
    if args.workload not in ("deer_9b", "tiny"):               # the constants above are the 3B model's
        out["whole_step"] = {"algorithmic_gflop_per_step": round(gflop, 1), "algorithmic_gb_per_step": round(gbyte, 3),
                             "TFLOP/s": round(gflop * value / world / 1e3, 1),
                             "mfma_frac": round(gflop * value / world / 1e3 / MFMA_PEAK_TF, 4),
                             "GB/s": round(gbyte * value / world, 1), "hbm_frac": round(gbyte * value / world / HBM_PEAK_GBS, 4),
                             "note": "per GPU; weights counted once per step (bf16), SURVEY.md 8(d)"}
    # the same target mix with STATIC exit ids (no pseudo action, no exit checks: one committing head call per step) - beside `value`
    # it prices the exit checks of the dynamic pipeline
    if B == 1:
        out["value_at_target_depth"] = {"value": out["value"], "unit": "action-steps/s", "avg_exit_layer": out["avg_exit_layer"],
                                        "note": "= `value` since round 4: the timed region itself runs the target mix (kept for run-to-run "
                                                "comparison with rounds 1-3, where it was the static-exit schedule now under `scripted`)"}
    if args.scripted_steps > 0 and B == 1 and max_layer == 12:
        exits = [e for e in cfg.exit_ids() if e <= eng.ctl_max_layer]
        sched = [exits[k] for k in target_schedule(len(exits), args.exit_ratio, args.scripted_steps, seed=99)]
        frames, ids_ = res["frames"], res["ids"]
        eng.reset()
        for e in exits:                                           # capture one graph per exit id
            eng.step(frames[0][0], frames[0][1], ids_, None, exit_id=e)
            eng.step(frames[0][0], frames[0][1], ids_, None, exit_id=e)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for i, e in enumerate(sched):
            eng.step(frames[i % len(frames)][0], frames[i % len(frames)][1], ids_, None, exit_id=e)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        st = torch.tensor([dt], dtype=torch.float64, device=eng.dev if (dist is None or dist.get_backend() == "nccl") else "cpu")
        if dist is not None:
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        out["scripted"] = {"value": round(world * len(sched) / float(st[0]), 2), "unit": "action-steps/s", "steps": len(sched),
                           "avg_exit_layer": round(sum(e + 1 for e in sched) / len(sched), 3),
                           "note": "static exit_id per step at the stratified target mix; two-chain vision + one trunk graph per exit "
                                   "id, no exit checks, host reads the action after every step"}
    if os.environ.get("DEER_PERSISTENT_LAYER") == "1":     # N1 experiment: never the default; say so on the line
        out["experiment"] = {"persistent_layer": True, "barrier_error_word": eng.persistent_layer_error(), "first_timeout": eng.persistent_layer_error_detail()}
    if rank == 0 and world == 1 and B == 1 and args.surface_steps > 0 and args.precision != "fp32":
        try:                                               # auxiliary single-rank leg: its failure must not cost the bench line
            out["surface"] = surface_leg(cfg, eng, res["thr"], res["frames"], res["ids"], args.surface_steps)
        except Exception as e:
            out["surface"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and not args.no_roofline:
        out["roofline"] = measure_roofline(eng, cfg, res["frames"], res["ids"], traffic_key="%s/envs%d" % (args.workload, B))
    # ---- the same workload with one ENV BATCH per rank (north_star: "one env batch per rank"): every weight byte and
    #      every kernel boundary is shared by the environments of the batch.  Reported beside `value`, never as it. ----
    if args.batched_envs > 1 and B == 1:
        del eng
        res["eng"] = None
        torch.cuda.empty_cache()
        nb = max(args.steps // 3, 20)
        rb = run_workload(args, cfg, sd, args.batched_envs, rank, world, local_rank, dist, max_layer, nb, max(args.warmup // 3, 5))
        out["batched"] = {"envs_per_gpu": args.batched_envs, "value": round(rb["value"], 2), "unit": "action-steps/s",
                          "steps": nb, "ms_per_step": round(1e3 * rb["t_max"] / nb, 4), "exit_hist": rb["hist"],
                          "ms_per_env_step": round(1e3 * rb["t_max"] / (nb * args.batched_envs), 4),
                          "avg_exit_layer": round(rb["avg_exit"], 3),
                          "note": "all environments of a rank advance in lock step through the same graph pieces; same kernels, "
                                  "same thresholds solver, per-environment exit decisions on the device"}
        if rank == 0 and world == 1 and args.precision != "fp32" and not args.no_two_groups:
            try:                                           # auxiliary single-rank leg: its failure must not cost the bench line
                out["batched_groups"] = two_groups_leg(cfg, rb, args.batched_envs, nb, 60, local_rank, G=args.batched_groups)
            except Exception as e:
                out["batched_groups"] = {"error": f"{type(e).__name__}: {e}"}
        if args.window_reps > 0 and rank == 0 and world == 1 and rb.get("eng") is not None:
            try:
                out["window"] = window_leg(rb["eng"], cfg, rb["frames"], rb["ids"], args.window_reps)
            except Exception as e:
                out["window"] = {"error": f"{type(e).__name__}: {e}"}
        rb["eng"] = None
        rb = None
        # the largest env batch one engine takes (round 5: 16 environments, 512 trunk rows): weights streamed once per 16
        from deer_vla_amd import _abi as abi
        if args.batched_envs < abi.MAX_ENVS and rank == 0 and world == 1 and args.precision != "fp32" and not args.no_two_groups:
            try:
                torch.cuda.empty_cache()
                nb2 = max(nb // 2, 15)
                r16 = run_workload(args, cfg, sd, abi.MAX_ENVS, rank, world, local_rank, dist, max_layer, nb2, max(args.warmup // 3, 5))
                out["batched_max_envs"] = {"envs_per_gpu": abi.MAX_ENVS, "value": round(r16["value"], 2), "unit": "action-steps/s", "steps": nb2,
                                           "ms_per_step": round(1e3 * r16["t_max"] / nb2, 4),
                                           "ms_per_env_step": round(1e3 * r16["t_max"] / (nb2 * abi.MAX_ENVS), 4),
                                           "avg_exit_layer": round(r16["avg_exit"], 3)}
                r16["eng"] = None
            except Exception as e:
                out["batched_max_envs"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not (args.no_cpu_baseline or world > 1):
            cb = cpu_baseline(cfg, sd, ctl, None, args.cpu_budget_s, args.cpu_threads, rank, full_protocol=args.cpu_full_protocol)
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
