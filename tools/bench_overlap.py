"""Do a head-evaluation graph and a trunk-layer graph overlap when replayed on two streams? (MI355X, full-size MPT-1B)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd)
eng.configure_exit(cfg.exit_ids(), 12, 1)
eng.set_thresholds([-1.0] * 5 + [1e5])
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
rgb, grip, ids = rgb.cuda().bfloat16(), grip.cuda().bfloat16(), ids.cuda()
for _ in range(3):
    eng.step(rgb, grip, ids, None)
P = eng._graphs[(ids.shape[1], False, "pieces")]
eng.ctl.zero_()                                     # nothing exited: every kernel runs
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
M, H = P["main"][2], P["head"][1]


def timed(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def both():
    M.replay()
    with torch.cuda.stream(side):
        H.replay()
    main.wait_stream(side)


print("trunk layer graph alone   : %.1f us" % timed(M.replay))
print("head evaluation graph alone: %.1f us" % timed(H.replay))
print("both, two streams          : %.1f us" % timed(both))
print("both, one stream           : %.1f us" % timed(lambda: (M.replay(), H.replay())))

# the same pair as parallel branches INSIDE one graph (fork / join captured)
T = ids.shape[1]
eng._pending = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        eng.enqueue_dynamic_heads(T, 1)
    eng._pending = None
    eng.enqueue_llm_layer(2, T, None, False, finalize=False)
    cur.wait_stream(side)
print("both, one graph with a fork: %.1f us" % timed(g.replay))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    eng.enqueue_dynamic_heads(T, 1)
    eng.enqueue_llm_layer(2, T, None, False, finalize=False)
print("both, one graph, serial    : %.1f us" % timed(g2.replay))
