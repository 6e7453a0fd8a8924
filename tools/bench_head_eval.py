"""One head evaluation (exit check) of a one-environment control step by itself: the one-launch form (csrc/head.hip: head_fused_kernel)
against the eight separate kernels, full 3B size, graph replay of N evaluations in a row.  "hot": the 42 MB of head weights stay in the
Infinity Cache between evaluations; "cold": a 320 MB device copy between evaluations evicts them (what a trunk layer streaming 174 MB
beside the check does to them in the step).  Also DEER_HEAD_WGS sweeps.  usage: bench_head_eval.py [workgroups ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn, _abi as abi
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
wgs_list = [int(v) for v in sys.argv[1:]] or [128]
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
T = ids.shape[1]
big_a = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
big_b = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")


def run(eng, label, n=12):
    eng.configure_exit(cfg.exit_ids(), 12, 1)
    eng.set_thresholds([-1.0] * 6)                       # no check ever fires: every evaluation does all of its work
    eng.reset()
    eng.step(rgb, grip, ids, mask, exit_id=11, use_graph=False)      # hidden states of every layer; deer_begin_step ran (pre-pass valid)
    abi.check(eng.lib.deer_begin_step(eng._h, abi.ptr(eng.step_info_pinned), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "begin")
    torch.cuda.synchronize()
    for cold in (False, True):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            def body():
                for i in range(n):
                    if cold:
                        big_b.copy_(big_a)
                    eng.enqueue_head(1 + 2 * (i % 5), T, abi.KIND_CHECK, slot=i % 5)
            body()
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                body()
            gc = None
            if cold:                                      # the copies alone, to subtract
                gc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gc, stream=st):
                    for i in range(n):
                        big_b.copy_(big_a)
            def timed(gr):
                for _ in range(2):
                    gr.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(5):
                    gr.replay()
                e1.record(st)
                st.synchronize()
                return e0.elapsed_time(e1) * 1e3 / (5 * n)
            us = timed(g) - (timed(gc) if cold else 0.0)
        print(f"{label:34s} {'cold weights' if cold else 'hot weights '}: {us:7.1f} us per evaluation", flush=True)
    assert eng.head_fused_error() == 0
    if os.environ.get("DEER_HF_TRACE") == "1" and "launch" in label:
        eng.enqueue_head(5, T, abi.KIND_CHECK, slot=2)
        torch.cuda.synchronize()
        t = eng._buf("head_fused_trace").view(torch.int64).cpu().tolist()
        t = [v for v in t if v > 0]
        names = ["start", "pool"] + [x for l in range(4) for x in (f"L{l} gathered", f"L{l} normalised", f"L{l} published")] + ["lstm end"] + \
                [x for i in range(2) for x in (f"fc{i} published",)] + ["final gathered", "done"]
        print("   phase time stamps of workgroup 0 (us since start; 100 MHz clock):")
        print("   " + "  ".join(f"{names[i] if i < len(names) else i}={(v - t[0]) / 100.0:.1f}" for i, v in enumerate(t)))


base = DeerEngine(cfg, sd)
base.set_head_fused(False)
run(base, "eight separate kernels")
for w in wgs_list:
    os.environ["DEER_HEAD_WGS"] = str(w)
    e = DeerEngine(cfg, None, weights_from=base)
    e.set_head_fused(True)
    run(e, f"one launch, {w} workgroups")
