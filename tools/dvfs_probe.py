"""What clock does the chip sustain under the frame-tile GEMM, and is the 21 -> 30 us per 257x256x1024 tile between 64 and 256 busy
CUs a clock effect?  ("measure the 30 % chip-level loss you named".)

Sustained loops (graph replay, rotating cold weight copies, ~1.5 s each) of the c_fc frame tile at 4 / 8 / 12 / 16 camera frames (= 64 / 128 /
192 / 256 workgroups, one per CU) and of in_proj at 16 frames, on random and on zero-filled operands, while a host thread samples the
shader clock and the package power from the amdgpu hwmon / sysfs nodes (fallback: rocm-smi).  Prints one row per leg:
workgroups, us per launch, TFLOP/s, mean / min sclk (MHz), mean power (W) and "tile us x GHz" = shader cycles per tile - flat cycles with a
falling clock say DVFS, rising cycles at a flat clock say contention (L2 / fabric / LDS).

usage: dvfs_probe.py [seconds per leg]"""
import ctypes, glob, os, subprocess, sys, threading, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5


def _find_nodes():
    """(sclk node, power node) of the first amdgpu card that has them"""
    out = {"sclk": None, "power": None, "dpm": None}
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for n in ("freq1_input",):
            p = os.path.join(hw, n)
            if os.path.exists(p) and out["sclk"] is None:
                out["sclk"] = p
        for n in ("power1_average", "power1_input"):
            p = os.path.join(hw, n)
            if os.path.exists(p) and out["power"] is None:
                out["power"] = p
    for p in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        out["dpm"] = out["dpm"] or p
    return out


NODES = _find_nodes()


def _read_int(p):
    try:
        with open(p) as f:
            return int(f.read().strip())
    except Exception:
        return None


def _read_dpm(p):
    try:
        with open(p) as f:
            for ln in f:
                if "*" in ln:
                    return int(ln.split(":")[1].strip().split("M")[0])
    except Exception:
        pass
    return None


def _smi_sample():
    """fallback: one rocm-smi call (slow, ~0.3 s)"""
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        import json
        d = json.loads(o)
        card = d[sorted(d)[0]]
        sclk = pw = None
        for k, v in card.items():
            if k.lower().startswith("sclk clock level"):
                sclk = int(str(v).split("(")[1].split("M")[0])
            if "power" in k.lower() and "(w)" in k.lower():
                pw = float(v)
        return sclk, pw
    except Exception:
        return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop_flag = False
        self.sclk, self.power = [], []

    def run(self):
        while not self.stop_flag:
            s = p = None
            if NODES["sclk"]:
                v = _read_int(NODES["sclk"])
                s = v / 1e6 if v else None
            elif NODES["dpm"]:
                s = _read_dpm(NODES["dpm"])
            if NODES["power"]:
                v = _read_int(NODES["power"])
                p = v / 1e6 if v else None
            if s is None and p is None:
                s, p = _smi_sample()
            if s:
                self.sclk.append(s)
            if p:
                self.power.append(p)
            time.sleep(0.02)


def leg(name, frames, N, K, tile, epi, zero):
    M = 257 * frames
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16) if zero else torch.randn(M, K, device="cuda", generator=g).bfloat16()
    n_w = 24
    Ws = [torch.zeros(N, K, device="cuda", dtype=torch.bfloat16) if zero else (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
          for _ in range(n_w)]
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.Stream()
    reps = 48

    def run():
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for r in range(reps):
            rc = lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(Ws[r % n_w]), K, None, abi.ptr(C), N, 0, M, N, K, 1, epi, None, tile, None, s)
            assert rc == 0, rc

    with torch.cuda.stream(st):
        run()
        st.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            run()
        for _ in range(3):
            gr.replay()
        st.synchronize()
        smp = Sampler()
        smp.start()
        t0 = time.perf_counter()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        while time.perf_counter() - t0 < SECS:
            for _ in range(8):
                gr.replay()
            n += 8
            st.synchronize()
        e1.record(st)
        st.synchronize()
        smp.stop_flag = True
        smp.join()
    # the synchronize() every 8 replays leaves host gaps: time one replay burst separately, back to back
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(20):
            gr.replay()
        e1.record(st)
        st.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * reps)
    tf = 2.0 * M * N * K / us / 1e6
    wgs = frames * (N // (256 if tile == 63 else 192 if tile == 64 else 128))
    sc = smp.sclk
    pw = smp.power
    ms, mn = (sum(sc) / len(sc), min(sc)) if sc else (float("nan"), float("nan"))
    mp = sum(pw) / len(pw) if pw else float("nan")
    cyc = us * ms / 1e3 if sc else float("nan")           # thousands of shader cycles per launch (one tile per CU: per tile)
    print(f"{name:28s} {'zero' if zero else 'rand':4s} wgs {wgs:4d}  {us:7.2f} us  {tf:7.1f} TFLOP/s  sclk {ms:6.0f} / {mn:6.0f} MHz  power {mp:6.0f} W  "
          f"kcycles per launch {cyc:6.1f}  ({len(sc)} samples)", flush=True)


def main():
    print("nodes:", NODES, flush=True)
    print("idle sclk:", _read_int(NODES["sclk"]) if NODES["sclk"] else _read_dpm(NODES["dpm"]) if NODES["dpm"] else _smi_sample(), flush=True)
    for zero in (False, True):
        for frames in (4, 8, 12, 16):
            leg(f"c_fc 257x256 tile, {frames} frames", frames, 4096, 1024, 63, abi.EPI_QGELU_BF16, zero)
        leg("in_proj 257x192 tile, 16 fr", 16, 3072, 1024, 64, abi.EPI_BF16, zero)
    # the same 64-workgroup launch while the OTHER 192 CUs stream HBM (fabric / L2 contention without the matrix pipes' power)
    print("done")


if __name__ == "__main__":
    main()
