#!/usr/bin/env python3
"""bench.py -- single-stream decode of RWKV-6-World-7B Q4_0 (BASELINE.json's metric) through librwkv.so on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one decoded token: one pass of the hot path (embedding row, 32 x (time mixing + channel mixing), ln_out, head,
on-device argmax) with weights and recurrent state already resident in HBM.  Weights are synthetic (seeded random blocks
with the tensor names / shapes / dtypes of the real checkpoint; no checkpoint is available offline), the decode is greedy.

Prints ONE JSON line (rank 0) with the contract's keys plus
  "roofline":     the dominant kernel (quantised single-token projection) timed per launch with HIP events on its stream,
  "cpu_baseline": the CPU oracle (a port of the reference's ggml CPU algorithm) timed on the host cores, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches
MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 MFMA = 2x the bf16 rate (MI355X_MICROARCH.md: bf16 ~2.5 PF dense, I8 >= 3944 TOPS measured)
MFMA_F16_PEAK_TOPS = 2500.0


KERNEL_SOURCES = {2: ["rwkv.cpp_amd/csrc/ring_v6.hip", "rwkv.cpp_amd/csrc/ring_geom.h", "rwkv.cpp_amd/csrc/persist.h", "rwkv.cpp_amd/csrc/fused_blocks.h", "rwkv.cpp_amd/csrc/kdev.h"],
                  1: ["rwkv.cpp_amd/csrc/mega_v6.hip", "rwkv.cpp_amd/csrc/persist.h", "rwkv.cpp_amd/csrc/fused_blocks.h", "rwkv.cpp_amd/csrc/kdev.h"],
                  3: ["rwkv.cpp_amd/csrc/persist_v47.hip", "rwkv.cpp_amd/csrc/persist.h", "rwkv.cpp_amd/csrc/fused_blocks.h", "rwkv.cpp_amd/csrc/kdev.h"]}


def kernel_source_stamp(kind):
    """sha256 over the sources of the persistent kernel (kind 2: LDS-DMA ring, 1: register prefetch): a PMC quote is only valid for the build it
    was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES.get(kind, []):
        try:
            h.update(open(os.path.join(ROOT, f), "rb").read())
        except OSError:
            h.update(b"missing:" + f.encode())
    return h.hexdigest()[:16]


def pmc_traffic(path_id, args, kind=0):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written
    by tools/pmc_summary.py): the counters need their own profiler runs, so bench.py can only quote them, and only for the
    workload, kernel AND kernel build they were taken on (the entry carries a hash of the kernel's sources; a quote from another
    build is refused, not repeated)."""
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(f):
        return None, None
    try:
        d = json.load(open(f))
    except Exception:
        return None, None
    e = d.get(f"{args.config}:{args.dtype}:path{path_id}" + (f":kind{kind}" if path_id == 2 else ""))
    if not e:
        return None, None
    if path_id == 2 and e.get("kernel_source_stamp") != kernel_source_stamp(kind):
        return None, "stale: profiles/pmc_traffic.json was taken on another build of this kernel (source stamp differs); re-run tools/gpu_final.sh"
    return e.get("hbm_bytes_per_launch"), e.get("source")


def prefill_source_stamp():
    """sha256 over the COMPLETE sources of the sequence-mode GEMMs (prefill_fast.hip: the timed default; prefill.hip: the exact arm; their shared
    header; kdev.h): a matrix-pipe PMC quote is only valid for the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rwkv.cpp_amd/csrc/prefill_fast.hip", "rwkv.cpp_amd/csrc/prefill.hip", "rwkv.cpp_amd/csrc/prefill_mm.h", "rwkv.cpp_amd/csrc/kdev.h"):
        try:
            h.update(open(os.path.join(ROOT, f), "rb").read())
        except OSError:
            h.update(b"missing:" + f.encode())
    return h.hexdigest()[:16]


def mfma_busy(args):
    """Matrix-pipe utilisation of k_mmq_mfma from the committed rocprofv3 PMC pass (profiles/pmc_mfma.json): counters need their own profiler
    run, so the bench can only quote them, for the workload AND the build of prefill.hip they were taken on (source stamp, like pmc_traffic)."""
    f = os.path.join(ROOT, "profiles", "pmc_mfma.json")
    try:
        e = json.load(open(f)).get(f"{args.config}:{args.dtype}:prefill")
    except Exception:
        return None
    if not e:
        return None
    if e.get("prefill_source_stamp") != prefill_source_stamp():
        return {"stale": "profiles/pmc_mfma.json was taken on another build of prefill.hip (source stamp differs or missing); re-run tools/gpu_pmc_mfma.sh"}
    return e


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="rwkv6-7b", help="key of rwkv_cpp_amd.synth.CONFIGS")
    ap.add_argument("--dtype", default="Q4_0")
    ap.add_argument("--model-dir", default=os.environ.get("RWKV_BENCH_DIR", "/tmp"))
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-baseline leg (0 disables it)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event pass")
    ap.add_argument("--mode", default="decode", choices=["decode", "prefill"],
                    help="decode: BASELINE.json's headline metric (default). prefill: rwkv_eval_sequence over --seq-len tokens (BASELINE config 3)")
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--parity-tokens", type=int, default=64, help="greedy tokens compared GPU vs CPU oracle on the benchmarked file (0 disables)")
    ap.add_argument("--abi-tokens", type=int, default=24, help="tokens timed through the unmodified rwkv_eval ABI (host state in/out every call; 0 disables)")
    ap.add_argument("--chain", action="store_true",
                    help="with --gpus N in ONE process (no torch.distributed.run): the layer chain of RWKV_MI_DEVICES over N devices, decode loop in C++")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs block of the default line (BASELINE C2 / C4 / C3-decode measured beside the headline)")
    ap.add_argument("--chain-devices", default=None, help="device list of --chain (default 0-(N-1); e.g. 0,0 runs two stages on one GPU)")
    return ap.parse_args()


def ensure_model_file(args, synth, rank, barrier):
    spec = synth.CONFIGS[args.config]
    path = os.path.join(args.model_dir, f"synthetic-{args.config}-{args.dtype}-seed42.bin")
    if rank == 0:
        marker = path + ".ok"
        if not (os.path.exists(path) and os.path.exists(marker)):
            t = time.time()
            info = synth.write_model(path, spec, args.dtype, seed=42)
            with open(marker, "w") as f:
                f.write(json.dumps(info))
            print(f"[bench] wrote {path}: {info['bytes'] / 1e9:.2f} GB, {info['params'] / 1e9:.2f} B params in {time.time() - t:.1f}s", file=sys.stderr)
    barrier()
    return path, spec


def usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, capped (the port is memory-bound)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


DTYPE_DESC = {
    "Q4_0": "int8 x int4 dot, f32 accumulate (Q4_0 weights, Q8_0 activations); f16 head",
    "Q4_1": "int8 x uint4 dot, f32 accumulate (Q4_1 weights, Q8_1 activations); f16 head",
    "Q5_0": "int8 x int5 dot, f32 accumulate (Q5_0 weights, Q8_0 activations); f16 head",
    "Q5_1": "int8 x uint5 dot, f32 accumulate (Q5_1 weights, Q8_1 activations); f16 head",
    "Q8_0": "int8 x int8 dot, f32 accumulate (Q8_0 weights, Q8_0 activations); f16 head",
    "FP16": "f16 weights x f16-rounded activations, f32 accumulate",
    "FP32": "f32",
}


def abi_rate(model, first_token, n):
    """Tokens/s through the UNMODIFIED reference ABI: rwkv_eval with the state handed in and out through host memory on every
    call (rwkv_eval.inc:2-22) plus the logits download -- what a caller that does not know the rwkv_mi_* extensions sees."""
    import numpy as np
    state = model.init_state()
    logits = np.empty(model.n_vocab, dtype=np.float32)
    tok = first_token
    model.eval(tok, state, state, logits)  # warm
    tok = int(np.argmax(logits))
    t0 = time.perf_counter()
    for _ in range(n):
        model.eval(tok, state, state, logits)
        tok = int(np.argmax(logits))
    dt = time.perf_counter() - t0
    return {"tokens_per_s": n / dt, "ms_per_token": dt * 1e3 / n, "tokens": n,
            "note": "rwkv_eval(ctx, token, state, state, logits): host state in + out and logits out on every call, numpy argmax on the host"}


def cpu_baseline(path, first_token, budget_s, snapshot_at=64):
    """Times the CPU oracle (oracle/rwkv_oracle.c: ggml's CPU algorithm restated, OpenMP) on the same file and the same
    greedy decode. Bounded sample; reported, never the target. The greedy tokens are kept: main() compares them with the GPU's."""
    import numpy as np
    import oracle_lib
    cores = usable_cores()
    oracle_lib.lib().orc_set_threads(cores)
    oracle_lib.lib().orc_set_fast(1)   # AVX2 / AVX-512-VNNI row kernels (oracle/rwkv_oracle_fast.c): bit-identical to the scalar oracle
    simd = "AVX-512-VNNI" if oracle_lib.lib().orc_fast_uses_vnni() else "AVX2"
    t0 = time.time()
    om = oracle_lib.OracleModel(path)
    load_s = time.time() - t0
    state = om.init_state()
    tok, n, t_start = first_token, 0, time.time()
    toks = []
    final = None
    while True:
        logits, state = om.eval(tok, state)
        tok = int(np.argmax(logits))
        toks.append(tok)
        n += 1
        if n <= max(1, snapshot_at):     # logits and state behind the LAST token the parity leg compares (it compares the first min(n, snapshot_at))
            final = (np.array(logits, copy=True), np.array(state, copy=True))
        el = time.time() - t_start
        if n >= 2 and (el + el / n > budget_s or n >= 64):
            break
    om.free()
    return {"value": n / el, "unit": "tokens/s", "cores": cores, "kind": "port", "simd": simd,
            "sample": f"{n} greedy decode tokens of the same model file on the host CPU ({el:.1f}s, load {load_s:.1f}s); "
                      f"ggml's CPU algorithm restated with {simd} block dots, OpenMP over rows"}, toks, final


def load_leg(pkg, lib, path, spec):
    """File -> HBM with the file's pages dropped from the page cache first (fsync + POSIX_FADV_DONTNEED; a tmpfs cannot drop them and the
    figure then equals the warm one): parallel pread into pinned staging buffers, asynchronous copies, re-pack kernels (model.hip)."""
    if os.environ.get("RWKV_BENCH_NO_COLD") == "1":
        return {}
    try:
        fd = os.open(path, os.O_RDONLY)
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        os.close(fd)
        prev = os.environ.get("RWKV_MI_NO_AUTOTUNE")
        os.environ["RWKV_MI_NO_AUTOTUNE"] = "1"      # (this context only measures the load: no path calibration)
        try:
            m = pkg.RWKVModel(lib, path, thread_count=1, gpu_layer_count=spec.n_layer + 1)
            sec, nbytes = m.load_stats()
            m.free()
        finally:
            if prev is None:
                del os.environ["RWKV_MI_NO_AUTOTUNE"]
            else:
                os.environ["RWKV_MI_NO_AUTOTUNE"] = prev
        return {"cold_seconds": sec, "cold_GBps": nbytes / max(sec, 1e-9) / 1e9,
                "what": "payload of the model file -> HBM (reads + copies + re-pack); cold = after fsync + POSIX_FADV_DONTNEED on the file"}
    except Exception as e:   # noqa: BLE001
        return {"cold_error": repr(e)}


KERNEL_NAMES = {2: "k6_ring (persistent decode kernel: all layers of the stage in one launch, weights streamed through an LDS ring by LDS-DMA)",
                3: "k6_mega (persistent decode kernel: all layers of the stage in one launch, weights prefetched into registers)",
                4: "k47_persist (persistent RWKV-4 / RWKV-7 decode kernel: all layers of the stage in one launch, weights prefetched into registers a phase ahead)",
                1: "fused single-token layer kernels (quantised row phases)",
                0: "k_mvq_t1 (quantised single-token projection)"}


OTHER_CONFIGS = [("C2", "rwkv4-169m", "Q5_1"), ("C4", "rwkv7-2b9", "Q5_1"), ("C3-decode", "rwkv6-1b6", "Q4_0")]


def other_configs(args, pkg, lib, synth, torch):
    """The other single-GPU decode configurations of BASELINE.json beside the headline, in the driver's own run: tokens/s, the fraction of
    8 TB/s of the dominant kernel and of the whole token, in-run parity against the CPU oracle (tokens, last logits, whole state). Every leg
    is wrapped: a failure there is reported in its entry and cannot lose the headline line."""
    import argparse
    out = {}
    for tag, cfg, dtype in OTHER_CONFIGS:
        t0 = time.time()
        try:
            sub = argparse.Namespace(**vars(args))
            sub.config, sub.dtype, sub.steps, sub.warmup, sub.abi_tokens, sub.cpu_seconds, sub.parity_tokens, sub.quiet_fail = cfg, dtype, 128, 8, 0, 4.0, 32, True
            os.environ["RWKV_BENCH_NO_COLD"] = "1"
            path, spec = ensure_model_file(sub, synth, 0, lambda: None)
            r = bench_decode(sub, pkg, lib, path, spec, torch)
            roof = r.get("roofline", {})
            out[tag] = {"workload": r["config"]["workload"], "tokens_per_s": r["value"], "ms_per_step": r["ms_per_step"], "steps": sub.steps,
                        "decode_path": r["config"]["decode_path"], "persist_kind": r["config"]["persist_kind"],
                        "kernel": roof.get("kernel"), "kernel_frac_of_8TBps": roof.get("frac"), "kernel_avg_launch_us": roof.get("avg_launch_us"),
                        "kernel_traffic_bytes_per_launch": roof.get("traffic"), "kernel_bytes_per_launch": roof.get("avg_bytes_per_launch"),
                        "token_frac_of_8TBps": r["hbm"]["frac_of_8TBps"], "algorithmic_bytes_per_token": r["hbm"]["algorithmic_bytes_per_token"],
                        "parity": {k: r.get("parity", {}).get(k) for k in ("tokens_checked", "equal", "tokens_equal", "logits_equal", "state_equal", "crosses_tag_wrap")},
                        "cpu_tokens_per_s": r.get("cpu_baseline", {}).get("value"), "seconds": round(time.time() - t0, 1)}
        except (Exception, SystemExit) as e:   # noqa: BLE001
            out[tag] = {"error": repr(e), "seconds": round(time.time() - t0, 1)}
    return out


def bench_decode(args, pkg, lib, path, spec, torch):
    """One step = one decoded token (embedding row, every layer, ln_out, head, on-device argmax), state resident in HBM."""
    import numpy as np
    load = load_leg(pkg, lib, path, spec)
    t0 = time.time()
    model = pkg.RWKVModel(lib, path, thread_count=1, gpu_layer_count=spec.n_layer + 1)
    load_s = time.time() - t0
    sec, nbytes = model.load_stats()
    load.update({"warm_seconds": sec, "bytes": nbytes, "warm_GBps": nbytes / max(sec, 1e-9) / 1e9, "context_seconds": load_s})
    bpt = model.bytes_per_token()
    first = 1103515245 % spec.n_vocab
    model.state_load(None)
    if args.warmup > 0:
        model.decode_greedy(first, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, ev_ms = model.decode_greedy(first, args.steps)
    torch.cuda.synchronize()
    wall_s = time.perf_counter() - t0
    tok_s = args.steps / wall_s
    path_id = model.decode_path()
    result = {
        "metric": "tokens/sec single-stream decode", "value": tok_s, "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_s * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": DTYPE_DESC.get(args.dtype, args.dtype), "data": "synthetic",
        "config": {"workload": f"{spec.name} {args.dtype} single-stream greedy decode, state resident in HBM", "layers": spec.n_layer,
                   "n_embed": spec.n_embed, "n_vocab": spec.n_vocab, "parallelism": "1 GPU", "decode_path": path_id, "persist_kind": model.persist_kind()},
        "hbm": {"algorithmic_bytes_per_token": bpt, "achieved_GBps": bpt * tok_s / 1e9, "frac_of_8TBps": bpt * tok_s / 1e9 / HBM_PEAK_GBS,
                "hip_event_ms_per_token": ev_ms / args.steps},
        "load_seconds": load_s, "load": load,
    }
    if not args.no_profile:
        p = model.profile_decode(first, min(args.steps, 32))
        if p["launches"] > 0:
            ach = p["bytes"] / max(p["kernel_ms"], 1e-9) / 1e6
            traffic, traffic_src = pmc_traffic(path_id, args, model.persist_kind())
            kname = KERNEL_NAMES[(3 if model.persist_kind() == 1 else 4 if model.persist_kind() == 3 else 2) if path_id == 2 else path_id]
            result["roofline"] = {"bound": "hbm", "kernel": kname + f" [{args.dtype}]", "achieved": ach,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                  "launches": p["launches"], "avg_launch_us": p["kernel_ms"] * 1e3 / p["launches"],
                                  "avg_bytes_per_launch": p["bytes"] / p["launches"]}
    if args.abi_tokens > 0:
        result["abi"] = abi_rate(model, first, args.abi_tokens)
    if args.cpu_seconds <= 0:
        result["parity"] = {"skipped": "--cpu-seconds 0: the CPU oracle did not run"}
    if args.cpu_seconds > 0:
        # The CPU oracle (checker and reported baseline) decodes n greedy tokens of the same file within its budget; the GPU then decodes
        # exactly n tokens from a fresh state -- on the persistent path with the rolling hand-over tag preset so that they cross its
        # 16-bit wrap (it advances 8 per layer: every 256 tokens at 32 layers) -- and tokens, the last token's LOGITS and the whole
        # recurrent STATE must be equal bit for bit (argmax alone would hide low-order differences).
        result["cpu_baseline"], cpu_toks, (cpu_logits, cpu_state) = cpu_baseline(path, first, args.cpu_seconds, args.parity_tokens)
        n = min(len(cpu_toks), args.parity_tokens)
        if n <= 0:
            result["parity"] = {"skipped": "--parity-tokens 0"}
        else:
            wrap = False
            if path_id == 2:
                # place the n compared tokens across the 16-bit wrap of the hand-over tag: throw-away tokens until the generation
                # (rwkv_mi_decode_generation: +8 per layer and launch) is n / 2 tokens below it
                per = 8 * spec.n_layer
                left = (0x10000 - (model.decode_generation() & 0xFFFF)) // per     # whole tokens before the wrap
                burn = (left - max(1, n // 2)) % (0x10000 // per)
                if burn > 0:
                    model.decode_greedy(first, burn)
                g0 = model.decode_generation() & 0xFFFF
                wrap = g0 + n * per > 0x10000
            model.state_load(None)
            gpu_toks, _ = model.decode_greedy(first, n)
            gpu_logits, gpu_state = model.logits_store(), model.state_store()
            healthy = model.healthy()
            eq_t = list(gpu_toks[:n]) == list(cpu_toks[:n])
            eq_l, eq_s = bool(np.array_equal(gpu_logits, cpu_logits)), bool(np.array_equal(gpu_state, cpu_state))
            equal = bool(eq_t and eq_l and eq_s and healthy)
            result["parity"] = {"tokens_checked": n, "equal": equal, "tokens_equal": eq_t, "logits_equal": eq_l, "state_equal": eq_s,
                                "crosses_tag_wrap": bool(wrap),
                                "what": "n greedy tokens of rwkv_mi_decode_greedy on this file from a fresh state vs the CPU oracle's: token ids, the last "
                                        "token's logits and the whole state, np.array_equal"}
            if not equal:
                if not getattr(args, "quiet_fail", False):
                    print(json.dumps(result))
                raise SystemExit(f"[bench] PARITY FAILURE: GPU != CPU oracle after {n} greedy tokens (tokens {eq_t}, logits {eq_l}, state {eq_s}, healthy {healthy})")
    model.free()
    return result


class seq_arms:
    """Context manager: sequence mode on the named arms. "exact" (the product's default since round 6): F16 matrices on k_mvf in ggml's addition
    order, quantised matrices on the walk of k_mmq_mfma -- bit for bit the CPU oracle and the serial path. "fast" (opt-in): k_mmq_fast /
    k_mmf16_seq -- the same operands, the f32 additions in plain order."""
    ENV = {"exact": {"RWKV_MI_SEQ_F16": "valu", "RWKV_MI_SEQ_Q": "exact"}, "fast": {"RWKV_MI_SEQ_F16": "mfma", "RWKV_MI_SEQ_Q": "fast"}}

    def __init__(self, arm):
        self.vars = self.ENV[arm]

    def __enter__(self):
        self.prev = {k: os.environ.get(k) for k in self.vars}
        os.environ.update(self.vars)

    def __exit__(self, *a):
        for k, v in self.prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def exact_arms():
    return seq_arms("exact")


def fast_arm_gate(args, model, prompt, spec, np):
    """The opt-in arms against the default ones on the WHOLE benchmarked model (the analogue of the reference's own criterion for quantised
    formats, tests/logit_difference_validator.inc:68,83: |sum of logit differences| <= 1.05 x a recorded value): after the T-token pass the
    next 64 greedy tokens must be the exact arm's, and sum(logits_fast - logits_exact) over the vocabulary must sit inside 1.05 x the value
    recorded in tests/reference_constants.py for this (config, dtype, T)."""
    import reference_constants as R
    out = {}
    res = {}
    for arm in ("exact", "fast"):
        with seq_arms(arm):
            model.state_load(None)
            lg = model.eval_resident(prompt, want_logits=True)
        tok0 = int(np.argmax(lg))
        toks, _ = model.decode_greedy(tok0, 64)
        res[arm] = (lg.astype(np.float64), [tok0] + [int(t) for t in toks])
    sigma = float((res["fast"][0] - res["exact"][0]).sum())
    same = res["fast"][1] == res["exact"][1]
    key = (args.config, args.dtype, len(prompt))
    rec = getattr(R, "FAST_ARM_LOGIT_DIFFERENCE_SUM", {}).get(key)
    out = {"greedy_tokens_after_the_pass_equal": bool(same), "tokens_compared": 65, "logit_difference_sum": sigma,
           "max_abs_logit_diff": float(np.abs(res["fast"][0] - res["exact"][0]).max()), "recorded_sum": rec,
           "within_recorded_bound": (None if rec is None else bool(abs(sigma) <= 1.05 * abs(rec) + 1e-6))}
    want_same = getattr(R, "FAST_ARM_GREEDY_CONTINUATION_EQUAL", {}).get(key)
    out["greedy_continuation_equal_when_recorded"] = want_same
    if not same:
        a, b = res["fast"][1], res["exact"][1]
        out["first_divergence_at_token"] = next(i for i in range(len(a)) if a[i] != b[i])
    # fails when the sum leaves the recorded bound, or when a continuation that was equal when recorded is not any more
    out["ok"] = bool((rec is None or out["within_recorded_bound"]) and (same or want_same is not True))
    return out


def bench_prefill(args, pkg, lib, path, spec, torch):
    """One step = one rwkv_eval_sequence pass over --seq-len prompt tokens (BASELINE config 3: the prefill GEMMs), state resident."""
    import numpy as np
    t0 = time.time()
    model = pkg.RWKVModel(lib, path, thread_count=1, gpu_layer_count=spec.n_layer + 1)
    load_s = time.time() - t0
    T = args.seq_len
    prompt = [int((1103515245 * i + 12345) % spec.n_vocab) for i in range(T)]
    model.state_load(None)
    for _ in range(max(1, args.warmup)):
        model.eval_resident(prompt, want_logits=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = model.eval_resident(prompt, want_logits=True)
    torch.cuda.synchronize()
    wall_s = time.perf_counter() - t0
    tok_s = args.steps * T / wall_s
    flops = model.prefill_flops(T)
    result = {
        "metric": "tokens/sec prefill (rwkv_eval_sequence)", "value": tok_s, "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_s * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": DTYPE_DESC.get(args.dtype, args.dtype), "data": "synthetic",
        "config": {"workload": f"{spec.name} {args.dtype} prefill of {T} tokens per pass (rwkv_mi_eval_resident = rwkv_eval_sequence with the state resident)",
                   "layers": spec.n_layer, "n_embed": spec.n_embed, "n_vocab": spec.n_vocab, "seq_len": T, "parallelism": "1 GPU"},
        "load_seconds": load_s,
    }
    peak = MFMA_I8_PEAK_TOPS if args.dtype.startswith("Q") else MFMA_F16_PEAK_TOPS
    # the opt-in arms (RWKV_MI_SEQ_Q=fast RWKV_MI_SEQ_F16=mfma) timed the same way, as a side figure: `value` is the product's default
    with seq_arms("fast"):
        model.state_load(None)
        for _ in range(max(1, args.warmup)):
            model.eval_resident(prompt, want_logits=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.eval_resident(prompt, want_logits=True)
        torch.cuda.synchronize()
        fwall = time.perf_counter() - t0
        model.state_load(None)
        fpp = model.profile_prefill(prompt)
    result["fast_arms"] = {"tokens_per_s": args.steps * T / fwall, "ms_per_step": fwall * 1e3 / args.steps,
                           "gemm_TOPs_in_launches": (fpp["ops"] / max(fpp["kernel_ms"], 1e-9) / 1e9 if fpp["launches"] else 0.0),
                           "gemm_avg_launch_us": fpp["kernel_ms"] * 1e3 / max(fpp["launches"], 1), "env": seq_arms.ENV["fast"],
                           "note": "opt-in: k_mmq_fast (block sums in plain K order) and k_mmf16_seq (F16 matrices on the matrix cores); not bit-identical to "
                                   "the serial path, gated below (`parity.fast_arms`)"}
    # dominant kernel: the int8 MFMA GEMM (k_mmq_mfma), timed per launch with HIP events on the context's stream in a separate pass
    model.state_load(None)
    pp = model.profile_prefill(prompt)
    gemm = pp["ops"] / max(pp["kernel_ms"], 1e-9) / 1e9 if pp["launches"] else 0.0
    result["roofline"] = {"bound": "mfma", "kernel": f"k_mmq_mfma [{args.dtype}] (v_mfma_i32_32x32x32_i8; every quantised projection of the sequence pass, ggml's addition order)",
                          "achieved": gemm, "peak": peak, "unit": "TOP/s (int8)" if args.dtype.startswith("Q") else "TFLOP/s", "frac": gemm / peak,
                          "traffic": None, "launches": pp["launches"], "avg_launch_us": pp["kernel_ms"] * 1e3 / max(pp["launches"], 1),
                          "ops_per_pass_in_these_launches": pp["ops"], "whole_pass_TOPs": flops * args.steps / wall_s / 1e12, "flops_per_pass": flops,
                          "mfma_busy": mfma_busy(args),
                          "note": "achieved = 2*T*N*K of the launches / their HIP-event time; whole_pass_TOPs = 2*T*sum(2-D layer weights) + 2*V*D over the wall time "
                                  "of the pass (WKV, LayerNorm, mixes, quantiser included); the kernel is bound by the f32 fold per block (VALU), not by the matrix pipe"}
    if args.parity_tokens > 0 and args.cpu_seconds > 0:
        import oracle_lib
        oracle_lib.lib().orc_set_threads(usable_cores())
        oracle_lib.lib().orc_set_fast(1)
        om = oracle_lib.OracleModel(path)
        n = min(T, max(48, args.parity_tokens))   # --parity-tokens 1024: the whole benchmarked pass (the oracle needs ~20 s for it)
        t0 = time.time()
        ol, ost = om.eval_sequence(prompt[:n], om.init_state())
        cpu_s = time.time() - t0
        om.free()
        # The default arms must equal the oracle bit for bit on the whole model. The opt-in arms (same operands, other addition order) are
        # held to a tolerance where it means something -- a TWO-layer slice of the same geometry (a 24 - 32-layer network of RANDOM weights
        # amplifies a 1e-7 difference to 1e-2 over 32 layers x 128 tokens) -- and to the whole-model gate of fast_arm_gate().
        model.state_load(None)
        gl = model.eval_resident(prompt[:n], want_logits=True)
        gst = model.state_store()
        exact = bool(np.array_equal(gl, ol) and np.array_equal(gst, ost))
        with seq_arms("fast"):
            model.state_load(None)
            tl = model.eval_resident(prompt[:n], want_logits=True)
            tst = model.state_store()
        err_l, err_s = float(np.abs(tl - ol).max()), float(np.abs(tst - ost).max())
        timed = {"bit_identical": bool(err_l == 0.0 and err_s == 0.0), "max_abs_logit_diff_full_model": err_l, "max_abs_state_diff_full_model": err_s}
        timed_ok = True
        if not timed["bit_identical"]:
            from rwkv_cpp_amd import synth as synth_mod
            sp = os.path.join(args.model_dir, f"synthetic-{args.config}-{args.dtype}-slice2.bin")
            synth_mod.write_model(sp, spec, args.dtype, seed=42, limit_layers=2)
            som = oracle_lib.OracleModel(sp)
            sol, sost = som.eval_sequence(prompt[:n], som.init_state())
            som.free()
            sm = pkg.RWKVModel(lib, sp, thread_count=1, gpu_layer_count=3)
            with seq_arms("fast"):
                sm.state_load(None)
                sl = sm.eval_resident(prompt[:n], want_logits=True)
                sst = sm.state_store()
            sm.state_load(None)
            xl = sm.eval_resident(prompt[:n], want_logits=True)
            xst = sm.state_store()
            sm.free()
            os.remove(sp)
            # 1e-2 * (1 + max |oracle|) on logits and state: a product is inside 2e-6 relative (tests/test_gpu_prefill_fast.py, test_gpu_seq_f16.py);
            # two layers of exp / WKV accumulation amplify it (1.4e-3 measured on the 1.6B slice over 1024 tokens, 4e-3 on RWKV-7's recurrence
            # over 128); one-layer slices stay inside 1e-4 in the test suite
            rel = 1e-2
            tol_l, tol_s = rel * (1.0 + float(np.abs(sol).max())), rel * (1.0 + float(np.abs(sost).max()))
            e_l, e_s = float(np.abs(sl - sol).max()), float(np.abs(sst - sost).max())
            timed_ok = bool(e_l <= tol_l and e_s <= tol_s and np.array_equal(xl, sol) and np.array_equal(xst, sost))
            timed.update({"two_layer_slice": {"max_abs_logit_diff": e_l, "max_abs_state_diff": e_s, "tolerance_logits": tol_l, "tolerance_state": tol_s,
                                              "default_arm_bit_identical": bool(np.array_equal(xl, sol) and np.array_equal(xst, sost)), "within_tolerance": timed_ok}})
        gate = fast_arm_gate(args, model, prompt, spec, np)
        timed["whole_model_gate"] = gate
        timed_ok = timed_ok and gate["ok"]
        equal = exact and timed_ok
        result["parity"] = {"tokens_checked": n, "equal": equal, "default_arms_bit_identical": exact, "fast_arms": timed,
                            "what": "logits and state after the first tokens of the prompt, GPU sequence pass vs CPU oracle: bit for bit on the default arms "
                                    "(k_mmq_mfma / k_mvf: what `value` times); the opt-in arms (RWKV_MI_SEQ_Q=fast RWKV_MI_SEQ_F16=mfma) within 1e-2 * (1 + max |oracle|) "
                                    "on a two-layer slice of the same geometry, and on the whole model: the 64 greedy tokens behind the pass equal the default arms', "
                                    "|sum(logits_fast - logits_default)| <= 1.05 x the value recorded in tests/reference_constants.py"}
        result["cpu_baseline"] = {"value": n / cpu_s, "unit": "tokens/s", "cores": usable_cores(), "kind": "port",
                                  "sample": f"{n}-token sequence pass of the same file on the host CPU ({cpu_s:.1f}s)"}
        if not equal:
            print(json.dumps(result))
            raise SystemExit("[bench] PARITY FAILURE: GPU sequence pass differs from the CPU oracle")
    model.free()
    return result


def bench_chain(args, pkg, lib, path, spec, torch):
    """--chain: rwkv_init_from_file under RWKV_MI_DEVICES builds one stage per listed device inside this process; the greedy loop is the
    library's (runner.cpp): per-stage hipGraph replays, residual stream device to device, the chosen token back to the first stage -- no
    Python between tokens. Same metric as the N = 1 line (single-stream tokens/s, total work fixed = strong scaling); the aggregate of N
    decode streams in flight (what a layer pipeline is for) is reported beside it."""
    import numpy as np
    n = args.gpus
    devices = args.chain_devices or (f"0-{n - 1}" if n > 1 else "0")
    os.environ["RWKV_MI_DEVICES"] = devices
    try:
        front = pkg.RWKVModel(lib, path)
    finally:
        del os.environ["RWKV_MI_DEVICES"]
    first = 1103515245 % spec.n_vocab
    front.state_load(None)
    front.decode_greedy(first, args.warmup)
    front.state_load(None)
    toks, ms = front.decode_greedy(first, args.steps)
    single_tok_s = args.steps / (ms / 1e3)
    # N decode streams through the same chain
    streams = [front] + [front.clone() for _ in range(max(1, n) - 1)]
    firsts = [(1103515245 * (j + 1)) % spec.n_vocab for j in range(len(streams))]
    for m in streams:
        m.state_load(None)
    pkg.RWKVModel.decode_greedy_streams(streams, firsts, args.warmup)
    for m in streams:
        m.state_load(None)
    mtoks, mms = pkg.RWKVModel.decode_greedy_streams(streams, firsts, args.steps)
    result = {
        "metric": "tokens/sec single-stream decode", "value": single_tok_s, "unit": "tokens/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": DTYPE_DESC.get(args.dtype, args.dtype), "data": "synthetic",
        "config": {"workload": f"{spec.name} {args.dtype} single-stream greedy decode through a layer chain of {n} stage(s) in one process "
                               f"(RWKV_MI_DEVICES={devices}; C++ decode loop, one direct launch per stage, the residual stream stored by a stage's last layer in the next stage's buffer)",
                   "layers": spec.n_layer, "n_embed": spec.n_embed, "n_vocab": spec.n_vocab, "parallelism": f"pp{n} (one process)"},
        "multi_stream": {"streams": len(streams), "tokens_per_s_aggregate": len(streams) * args.steps / (mms / 1e3), "ms_per_step": mms / args.steps,
                         "note": "independent decode streams (clones) interleaved through the same chain: aggregate throughput"},
    }
    if args.parity_tokens > 0:
        # the chain against a one-device context of the same file (itself checked against the CPU oracle by the N = 1 line and tests/)
        k = min(args.parity_tokens, args.steps)
        ref = pkg.RWKVModel(lib, path)
        bpt = ref.bytes_per_token()            # the whole model's algorithmic bytes per token (the chain's stages together stream the same)
        result["roofline"] = {"bound": "hbm", "kernel": "the stages' single-token kernels, one stage active at a time", "achieved": bpt * single_tok_s / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bpt * single_tok_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                              "note": "algorithmic bytes per token x tokens/s of the whole chain (hops and launch gaps included); per-kernel launch times "
                                      "and PMC traffic are quoted by the N = 1 line"}
        ref.state_load(None)
        rt, _ = ref.decode_greedy(first, k)
        rs = ref.state_store()
        front.state_load(None)
        ct, _ = front.decode_greedy(first, k)
        cs = front.state_store()
        equal = bool(np.array_equal(rt, ct) and np.array_equal(rs, cs) and np.array_equal(toks[:k], rt) and np.array_equal(mtoks[0][:k], rt))
        result["parity"] = {"tokens_checked": int(k), "equal": equal,
                            "what": "tokens and recurrent state, stage chain vs a one-device context of the same file, bit for bit (transitive: the "
                                    "one-device context is what the N = 1 line and tests/ compare with the CPU oracle)"}
        ref.free()
        if args.cpu_seconds > 0:
            # ... and directly against the CPU oracle for the first tokens its budget allows
            _, cpu_toks, (cpu_logits, cpu_state) = cpu_baseline(path, first, args.cpu_seconds, k)
            kk = min(k, len(cpu_toks))
            front.state_load(None)
            ot, _ = front.decode_greedy(first, kk)
            o_eq = bool(list(ot[:kk]) == list(cpu_toks[:kk]) and np.array_equal(front.state_store(), cpu_state))
            result["parity"]["oracle"] = {"tokens_checked": int(kk), "equal": o_eq, "what": "tokens and recurrent state, stage chain vs the CPU oracle"}
            equal = equal and o_eq
        if not equal:
            print(json.dumps(result))
            raise SystemExit("[bench] PARITY FAILURE: the stage chain differs from the one-device context")
    for m in streams[1:]:
        m.free()
    front.free()
    return result


def main():
    args = parse_args()
    # watchdog: a device call that never returns must end the process (with the Python stacks on stderr), not hang the box
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("RWKV_BENCH_WATCHDOG_S", "1500")), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if world == 1 and args.gpus > 1 and not args.chain:
        # a bare `bench.py --gpus N` (no torch.distributed.run): the one-process layer chain needs no launcher. Never a 1-GPU line under an
        # N-GPU flag: with fewer than N devices visible the run stops here.
        have = torch.cuda.device_count()
        if have < args.gpus and not args.chain_devices:
            raise SystemExit(f"[bench] --gpus {args.gpus} but {have} device(s) visible and no torch.distributed.run world: nothing to measure "
                             f"(use --chain-devices 0,0,... to run the stages on fewer devices)")
        print(f"[bench] --gpus {args.gpus} without torch.distributed.run: running the one-process layer chain (RWKV_MI_DEVICES=0-{args.gpus - 1})", file=sys.stderr)
        args.chain = True
    elif world > 1 and world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}")

    backend = os.environ.get("RWKV_BENCH_BACKEND", "nccl")   # "gloo" only for smoke-testing the N > 1 path on a single GPU
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    import __graft_entry__ as graft
    pkg = graft.load_package()
    from rwkv_cpp_amd import synth
    if rank == 0 and not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    barrier()
    lib = pkg.load_rwkv_shared_library()

    path, spec = ensure_model_file(args, synth, rank, barrier)

    if args.chain and world == 1:
        result = bench_chain(args, pkg, lib, path, spec, torch)
    elif world > 1:
        from rwkv_cpp_amd import pipeline  # layer pipeline over RCCL send/recv
        result = pipeline.bench_pipeline(args, lib, path, spec, dist, rank, local_rank, world)
    elif args.mode == "prefill":
        result = bench_prefill(args, pkg, lib, path, spec, torch)
    else:
        result = bench_decode(args, pkg, lib, path, spec, torch)
        if args.config == "rwkv6-7b" and args.dtype == "Q4_0" and not args.no_other_configs:
            result["other_configs"] = other_configs(args, pkg, lib, synth, torch)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
