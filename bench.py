#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s vocoded by the V1 + ODConv + GRC-LoRA generator (BASELINE.json
configs[1]: B=32 clips x 32 mel frames -> 8192 samples, 80-mel, 22.05 kHz, inference), one process per
GPU.  Prints ONE JSON line on rank 0 (contract in the task description).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype fp32|bf16|fp16] [--no-cpu-baseline]

`--gpus N` (N > 1) without a torchrun environment starts `python -m torch.distributed.run` with N ranks as a CHILD
process before anything touches the GPU and relays its output; under torchrun (WORLD_SIZE set) each rank runs in place.

PRECISION.  north_star asks for <= 1e-3 relative L2 on the waveform.  tools/error_budget.py (a CPU simulation of every
rounding the 16-bit pipeline performs) shows that no pipeline with single 16-bit MFMA operands can meet that on this
network: fp16 everywhere lands at 1.5e-3 (22 kHz) / 4.8e-3 (48 kHz) from rounding alone, bf16 at 1.2e-2.  The headline
therefore runs the fastest mode that does meet it: fp32 storage with every MFMA operand split into a hi and a lo bf16
part (three products per MAC, fp32 accumulate; `--dtype fp32`).  The 16-bit storage modes are timed beside it under
"modes", each with its measured parity and `"parity_ok": false`.

A "step" is one generator forward over one batch of synthetic mels already resident in HBM.
Inference shards by minibatch with no data-path collective (replicas only -> weak scaling).
`roofline` is measured live with HIP events on the launch stream around the MRF ("ResBlock") stage,
the largest HBM mover of the path (64 channels in + 64 out per output sample = 134.2 MB
per block per batch of 32 in fp32 storage, 67.1 MB in 16-bit storage); `roofline.traffic` comes from the committed PMC
summary (profiles/, written by tools/profile_summary.py) and is null when that summary was taken from other kernel sources.
`cpu_baseline` times the CPU oracle (reference K-loop formulation, fp32) on the host cores of the same box for a bounded sample.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "a-modified-hifi-gan-vocoder-using-odconv-and-grc-for-expressive-voice-cloning-_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def cpu_baseline(sd, n_threads, budget_s=24.0, checks=None, embed_check=None, check48=None, check_v3=None):
    """The CPU leg - the only place bench.py touches oracle/ (as the checker and the reported baseline, never as the thing
    measured).  `checks`: {name: (gpu waveform on the host, (mel, spk, emo) fp32 host inputs)} -> rel-L2 of each against the
    oracle, returned under "parity".
    Oracle (reference formulation: K shared-weight convs + alpha-weighted sum) on the host cores, fp32.
    torch's CPU convolutions do not scale to every core of a 2-socket box, so a few thread counts share the time budget and
    the best one is reported as `value` (with its `cores`); the whole sweep, the single-thread figure SURVEY 8(d) asks for
    and the CPU model are kept beside it."""
    from oracle import vocoder_oracle as O
    torch.manual_seed(1)
    Tm = 32
    mel8, spk8, emo8 = torch.randn(8, 80, Tm), torch.randn(8, 192), torch.randn(8, 384)

    def run(threads, B, budget):
        torch.set_num_threads(threads)
        mel, spk, emo = mel8[:B], spk8[:B], emo8[:B]
        with torch.no_grad():
            O.generator_forward(mel, sd, "", spk, emo)  # warm-up
            t0, n = time.perf_counter(), 0
            while True:
                O.generator_forward(mel, sd, "", spk, emo)
                n += 1
                el = time.perf_counter() - t0
                if el > budget or n >= 50:
                    break
        return {"cores": threads, "batch": B, "value": round(B * Tm * n / el, 1), "forwards": n, "seconds": round(el, 1)}

    cands = sorted({1, min(16, n_threads), min(32, n_threads), n_threads})
    sweep = [run(t, 1 if t == 1 else 8, budget_s / len(cands)) for t in cands]
    best = max(sweep, key=lambda r: r["value"])
    out = {"value": best["value"], "unit": "mel-frames/s", "cores": best["cores"], "kind": "port",
           "sample": "%d forwards of B=%d x %d frames, fp32, oracle K-loop form, %.1f s (best of the thread sweep)"
                     % (best["forwards"], best["batch"], Tm, best["seconds"]),
           "value_1thread": sweep[0]["value"], "sweep": sweep}
    try:
        with open("/proc/cpuinfo") as fh:
            out["cpu_model"] = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
    except OSError:
        out["cpu_model"] = "unknown"
    if checks:
        torch.set_num_threads(best["cores"])
        out["parity"] = {}
        with torch.no_grad():
            for name, (wave, (m, sp, em)) in checks.items():
                out["parity"][name] = O.rel_l2(wave, O.generator_forward(m, sd, "", sp, em))
    if check_v3 is not None:         # configs[0]: the plain V3 generator on the CPU (restatement of the published architecture), B=1
        w3, m3, sd3 = check_v3
        torch.set_num_threads(best["cores"])
        with torch.no_grad():
            ref3 = O.plain_hifigan_forward(m3, sd3)
            out.setdefault("parity", {})["config_v3_plain"] = O.rel_l2(w3, ref3)
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < 2.0 and n < 10:
                O.plain_hifigan_forward(m3, sd3)
                n += 1
            out["config_v3_plain"] = {"value": round(344 * n / (time.perf_counter() - t0), 1), "unit": "mel-frames/s", "cores": best["cores"],
                                      "kind": "port", "sample": "%d forwards of B=1 x 344 frames, fp32" % n}
    if check48 is not None:          # 48 kHz leg: one B=2 oracle forward as the checker of every storage mode timed there
        w48, (mm, ss, ee), sd48 = check48
        torch.set_num_threads(best["cores"])
        with torch.no_grad():
            ref48 = O.generator_forward(mm, sd48, "", ss, ee, upsample_factors=(8, 8, 4, 2))
            for tag, w in w48.items():
                out.setdefault("parity", {})["config_48k_" + tag] = O.rel_l2(w, ref48)
    if embed_check is not None:      # conditioning producers: oracle/embed_oracle.py as checker and as the CPU figure (bounded: <= 3 s)
        from oracle import embed_oracle as E
        esd, emel, spk_gpu, emo_gpu = embed_check
        torch.set_num_threads(best["cores"])
        with torch.no_grad():
            spk_o, emo_o = E.embedding_extractor(emel, esd)
            out.setdefault("parity", {})["conditioning"] = {"speaker": O.rel_l2(spk_gpu, spk_o), "emotion": O.rel_l2(emo_gpu, emo_o)}
            m8 = torch.randn(8, 80, Tm)
            E.embedding_extractor(m8, esd)
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < 3.0 and n < 20:
                E.embedding_extractor(m8, esd)
                n += 1
            out["conditioning"] = {"value": round(8 * Tm * n / (time.perf_counter() - t0), 1), "unit": "mel-frames/s", "cores": best["cores"],
                                   "kind": "port", "sample": "%d forwards of B=8 x %d frames, fp32" % (n, Tm)}
    return out


def graph_time_ms(run, inner=20, outer=5):
    """Average duration (ms) of one `run()` with the launches captured in a HIP graph (inner copies per replay), timed with
    HIP events on the launch stream: eager Python issue can be slower than these short kernels, which would time the host."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(outer):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    t_graph = e0.elapsed_time(e1) / (inner * outer)
    # eager issue keeps the queue full when the host is faster than the kernels (then the events see back-to-back launches
    # without the graph's per-node dependency gaps); whichever way the host was not the bottleneck is the smaller figure
    e0.record()
    for _ in range(inner * outer):
        run()
    e1.record()
    torch.cuda.synchronize()
    t_eager = e0.elapsed_time(e1) / (inner * outer)
    return min(t_graph, t_eager)


PARITY_TOL = 1e-3           # north_star: waveform rel-L2 vs the reference CPU path
DT = {"bf16": "bfloat16", "fp16": "float16", "fp32": "float32", "mixed": "float32"}
MIXED_THROUGH = "up1"       # --dtype mixed: fp16 storage up to and including this stage, fp32 storage (split operands) behind it
MIXED_MRF_W16 = True        # ... and the three MRF blocks with two-product operands (hi + lo f16 activations x single f16 weights)


def spawn_ranks(n):
    """`bench.py --gpus N` outside torchrun: run the N ranks as a child `torch.distributed.run` (started before this process has
    made any GPU call - a process that initialised the GPU must never exec) and exit with its code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def _pmc_summaries(dtype_tag, files=None):
    """Committed PMC summaries (profiles/*_pmc_traffic.json, tools/profile_summary.py) taken at this dtype from the kernel sources
    this run is using (sha1 of csrc/* recorded in the summary); newest file name last."""
    pdir = os.path.join(ROOT, "profiles")
    out = []
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not f.endswith("_pmc_traffic.json"):
            continue
        try:
            j = json.load(open(os.path.join(pdir, f)))
        except (OSError, ValueError):
            continue
        if j.get("dtype") != dtype_tag:
            continue
        ok = True
        for src, sha in j.get("sources", {}).items():
            if files is not None and src not in files:
                continue                   # only the sources the quoted kernel is built from have to be unchanged
            path = os.path.join(PKG, "csrc", src)
            ok = ok and os.path.exists(path) and hashlib.sha1(open(path, "rb").read()).hexdigest() == sha
        if ok:
            out.append((f, j))
    return out


def pmc_traffic(kernel_substr, dtype_tag, per_launch_of=None, files=None, nblocks=1):
    """HBM bytes from the committed PMC summary, or None when no summary matches this dtype / these kernel sources.
    Sum over the kernels whose name contains `kernel_substr` of (fetch + write bytes); divided by the launch count of the kernel
    whose name contains `per_launch_of` (e.g. one MRF block = all mrf_kernel passes / launches of its once-per-block pass),
    or - without it - the plain per-launch average of the single matching kernel.  nblocks: the marker kernel runs once per chain
    of that many blocks (the streaming chain's F0 pass)."""
    best = None
    for f, j in _pmc_summaries(dtype_tag, files):
        tot, marker = 0.0, 0
        for name, k in j.get("kernels", {}).items():
            if kernel_substr in name:
                per = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
                tot += per * (k["launches"] if per_launch_of else 1)
            if per_launch_of and kernel_substr in name and per_launch_of in name:
                marker += k["launches"]
        if tot and (marker or not per_launch_of):
            best = {"bytes": int(tot / marker / nblocks) if per_launch_of else int(tot), "file": "profiles/" + f}
    return best


def timed_replays(replay, n):
    import torch
    for _ in range(20):
        replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="mixed", choices=["bf16", "fp16", "fp32", "mixed"],
                    help="storage of the headline run; fp32 = split bf16 MFMA operands everywhere; mixed = fp16 storage through "
                         "the second upsampler, fp32 behind it (both within north_star's 1e-3 on this configuration)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the bf16 / fp16 storage legs")
    ap.add_argument("--no-conditioning", action="store_true", help="skip the embedding-extractor leg")
    ap.add_argument("--no-v3", action="store_true", help="skip the plain HiFi-GAN V3 leg (BASELINE configs[0])")
    ap.add_argument("--no-48k", action="store_true", help="skip the 48 kHz leg (BASELINE configs[4] shapes)")
    ap.add_argument("--eager", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--train-steps", type=int, default=5, help="timed training steps (0 = skip the training metric)")
    ap.add_argument("--train-warmup", type=int, default=2)
    ap.add_argument("--train-dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--train-eager", action="store_true", help="issue the training step eagerly instead of replaying its HIP graph")
    ap.add_argument("--no-train-alt", action="store_true", help="skip the fp32 / fp16 activation legs of the training metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist
    # one process per GPU over RCCL ("nccl").  MV_DIST_BACKEND=gloo lets several ranks share one device for a rehearsal
    backend = os.environ.get("MV_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    import hifigan_modified as H
    from hifigan_modified import _native
    _native.lib()  # no fallback: raise if the HIP library is absent
    from hifigan_modified.graphs import GraphedVocoder

    dtype = getattr(torch, DT[args.dtype])
    torch.manual_seed(0)
    gen0 = H.ModifiedHiFiGANGenerator()
    sd_cpu = {k: v.detach().clone() for k, v in gen0.state_dict().items()}
    B, Tm = args.batch, args.frames
    torch.manual_seed(1 + rank)
    mel32 = torch.randn(B, 80, Tm, device=dev)
    spk32 = torch.randn(B, 192, device=dev)
    emo32 = torch.randn(B, 384, device=dev)

    def build(dt, mixed=False, w16=MIXED_MRF_W16):
        g = H.ModifiedHiFiGANGenerator()
        g.load_state_dict(sd_cpu)
        g = g.to(dev).to(dt).train(False)
        if mixed:
            g.set_mixed_precision(MIXED_THROUGH, torch.float16, mrf_weights="fp16" if w16 else None)
        return g, mel32.to(dt), spk32.to(dt), emo32.to(dt)

    gen, mel, spk, emo = build(dtype, mixed=(args.dtype == "mixed"))
    # the MRF blocks of the mixed mode are the fp32-storage chain, with two-product operands in its streaming form (mrf_stream.hip)
    mrf_tag = ("fp32w16" if MIXED_MRF_W16 else "fp32") if args.dtype == "mixed" else args.dtype

    # waveforms of 2 clips of this very configuration, checked against the oracle in the CPU leg (cpu_baseline)
    checks = {}
    host = lambda t: t.float().cpu()
    if rank == 0:
        with torch.no_grad():
            checks["headline"] = (host(gen(mel[:2], spk[:2], emo[:2])), (host(mel[:2]), host(spk[:2]), host(emo[:2])))

    if args.eager:
        def step():
            with torch.no_grad():
                return gen(mel, spk, emo)
    else:
        # one HIP graph per step: the ~20 launches of a forward are host-bound when issued eagerly
        graphed = GraphedVocoder(gen, mel, spk, emo)
        step = graphed.replay

    # untimed pre-warm (about half a second of replays): a fresh box starts with idle clocks and cold caches, which the
    # caller's W warmup steps (a few ms of GPU time) do not cover; then the W warmup steps of the contract
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # live roofline of the MRF ("ResBlock") stage: HIP events on the launch stream (torch's current stream) around
    # the fused channels-last block = 3 launches of mv::mrf_kernel<T,...,PASS=1|2|3> (GN5 stats, GN8 stats, output)
    from hifigan_modified import ops
    from hifigan_modified.fused import generator_fused_for

    def mrf_roofline(g, m, s, e, tag):
        """One MRF block as the generator really runs it.  fp32 storage: a third of the three-block chain's passes, timed WITHOUT a
        materialising pass (the generator forms the last block's output inside its output conv) - tile form (`fp32`: PASS 1 +
        3 x PASS 5 + 2 x PASS 4 of mv::mrf_kernel<float>) or streaming form (`fp32w16`: V0 + F0 + 2 x (A + F) of
        mv::mrf_stream_kernel).  16-bit storage: the per-block kernel (3 pass launches)."""
        fzz = generator_fused_for(g)
        w16 = tag == "fp32w16"
        with torch.no_grad():
            st_ = g(m, s, e, return_stages=True)
            x_cl = ops.nct_to_ntc(st_["up%d" % (len(g.upsample_layers) - 1)])
            nblk, chained = 1, False
            if fzz is not None and fzz.chain is not None and x_cl.dtype == torch.float32:
                x_in, pair = x_cl, False
                from hifigan_modified import fused as _fz
                if w16 and _fz._GEN_PAIR:     # (MV_GEN_PAIR=1: the generator hands the chain the last upsampler's pair rows - time that form)
                    from hifigan_modified import functional as _F2
                    li = len(g.upsample_layers) - 1
                    xin = ops.nct_to_ntc(st_["up%d" % (li - 1)]).float().contiguous()
                    x_in, pair = fzz.ups[li].forward_cl(xin, _F2._cache, pooled_in=xin.sum(dim=1).contiguous(), act=1, out_pair=True)
                run, nblk, chained = (lambda: fzz.chain.forward_cl(x_in, w16=w16, materialize=False, x_pair=pair)), len(fzz.mrfs), True
            elif fzz is not None:
                run = lambda: fzz.mrfs[0].forward_cl(x_cl)
            else:
                run = lambda: g.mrf_blocks[0](st_["up3"])
            ms = graph_time_ms(run) / nblk
        elt = x_cl.element_size()
        alg_bytes = 2 * x_cl.numel() * elt          # in + out of the block once, at the storage width
        alg_8d = 2 * x_cl.numel() * 2               # SURVEY 8(d)'s figure: 256 B per output sample (64 ch in + 64 ch out x 2 B)
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        # one block = every pass of this storage type / launches of the pass that runs once per block
        cname = {"fp32": "mrf_kernel<float", "fp32w16": "mrf_stream_kernel", "bf16": "mrf_kernel<__hip_bfloat16", "fp16": "mrf_kernel<_Float16"}[tag]
        per = "mrf_stream_kernel<1," if w16 else (", 5, " if chained else ", 3, ")
        tr = pmc_traffic(cname, tag, per_launch_of=per, nblocks=(nblk if w16 else 1),
                         files=(("mrf_stream.hip", "mrf_common.h") if w16 else ("mrf_fused.hip",)) + ("mfma.h", "common.h")) if (B == 32 and Tm == 32) else None
        how = ("streaming three-block chain / 3: 6 pass launches per 3 blocks (V0, F0, 2 x (A, F)), GroupNorm(8,64) deferred into the next "
               "block's pass A, residual stream between blocks as pre-split f16 hi/lo rows read by LDS-DMA" if w16 else
               "three-block chain / 3: 6 pass launches per 3 blocks, GroupNorm(8,64) deferred into the next block's load"
               if chained else "per-block kernel = 3 pass launches")
        r = {"bound": "hbm", "kernel": "mv::%s> (fused MRF block; %s)" % (cname, how), "achieved": round(achieved, 1),
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
             "frac_survey_8d": round(alg_8d / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "traffic": tr["bytes"] if tr else None, "traffic_source": tr["file"] if tr else None,
             "alg_bytes_per_launch": alg_bytes, "alg_bytes_survey_8d": alg_8d, "ms_per_launch": round(ms, 4),
             "note": "achieved / frac use the bytes of the storage type that meets north_star (fp32: 512 B per output sample); "
                     "frac_survey_8d prices the same time against SURVEY 8(d)'s 2-byte figure (256 B per output sample).  GroupNorm is global "
                     "in T, so a block is several passes over the stream: the chain moves 5 stream transfers per block (A: read f, read x, "
                     "write x'; F: read x', write f), i.e. 2.5x the storage-width algorithmic bytes by construction; in-kernel marks put the "
                     "passes at 3.9-5.4 TB/s of moved bytes (DESIGN.md section 4)"}
        return r, st_, fzz

    roof = od_roof = None
    if rank == 0:
        roof, st, fz = mrf_roofline(gen, mel, spk, emo, mrf_tag)
        # per-kernel HBM figures of the HBM-bound fused ODConvTranspose1d launches ((Cin/f + Cout) * es B per output sample)
        if fz is not None:
            od_roof = []
            from hifigan_modified import functional as _Fn
            elt = mel.element_size()
            names = {0: "odconv_kloop_kernel (banks as stored, weights-stationary over 2 samples)", 1: "odconv_sample_kernel (sample resident in LDS)",
                     2: "odconv_stream_kernel<128,4,fp16 in>", 3: "odconv_stream_kernel<64,8>"} if args.dtype == "mixed" else {}
            with torch.no_grad():
                for li in range(len(fz.ups)):
                    u = fz.ups[li]
                    # every layer exactly as the generator runs it: input in its producer's storage type, output in its own
                    xin = ops.nct_to_ntc(st["up%d" % (li - 1)] if li > 0 else st["film"]).contiguous()
                    sdt_ = st["up%d" % li].dtype
                    pooled = xin.float().sum(dim=1).contiguous()
                    run = lambda: u.forward_cl(xin, _Fn._cache, pooled_in=pooled, act=1, storage=sdt_)
                    y = run()
                    ms_u = graph_time_ms(run)
                    wb = u.mod.kernels.numel() * (2 if sdt_ != torch.float32 else 4)           # the K banks, read once
                    byts = xin.numel() * xin.element_size() + y.numel() * y.element_size() + wb
                    od_roof.append({"kernel": "mv::%s (upsample_layers.%d: %d->%d ch, x%d, %s -> %s)" % (
                                        names.get(li, "odconv_cl_*_kernel<%s>" % mrf_tag), li, u.mod.in_channels, u.mod.out_channels, u.mod.stride,
                                        str(xin.dtype).replace("torch.", ""), str(sdt_).replace("torch.", "")),
                                    "bound": "hbm", "achieved": round(byts / (ms_u * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(byts / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes_per_launch": byts,
                                    "ms_per_launch": round(ms_u, 4),
                                    "note": "algorithmic bytes = input + output once at their storage widths + the K kernel banks once (SURVEY 8(d)); "
                                            "the first two upsamplers are bound by the bank stream out of L2 and the per-sample kernel mix, not by HBM "
                                            "(DESIGN.md section 4)" if li < 2 else
                                            "algorithmic bytes = input + output once at their storage widths + the K kernel banks once (SURVEY 8(d))"})
        del st

    # ---------------------------------------------------------------- the 16-bit storage modes beside the headline (rank 0, N=1)
    modes = None
    if rank == 0 and world == 1 and not args.eager and not args.no_modes:
        modes = {}
        for tag in ("bf16", "fp16", "fp32", "mixed_3prod"):
            if tag == args.dtype or (tag == "mixed_3prod" and not (args.dtype == "mixed" and MIXED_MRF_W16)):
                continue
            if tag == "mixed_3prod":     # round 2's headline mode: the same storage mix with three-product (hi + lo bf16) MRF operands
                g2, m2, s2, e2 = build(torch.float32, mixed=True, w16=False)
            else:
                dt = getattr(torch, DT[tag])
                g2, m2, s2, e2 = build(dt)
            with torch.no_grad():
                checks["mode_" + tag] = (host(g2(m2[:2], s2[:2], e2[:2])), (host(m2[:2]), host(s2[:2]), host(e2[:2])))
            gv = GraphedVocoder(g2, m2, s2, e2)
            el = timed_replays(gv.replay, 200)
            modes[tag] = {"value": round(B * Tm / el, 1), "unit": "mel-frames/s", "ms_per_step": round(el * 1e3, 4),
                          "parity_rel_l2_vs_oracle": None, "parity_ok": None}
            if tag == "bf16":        # the BASELINE-named storage type: its MRF kernel against the same roofline
                modes[tag]["roofline"] = mrf_roofline(g2, m2, s2, e2, tag)[0]
            del gv, g2

    # ---------------------------------------------------------------- BASELINE configs[4]: 48 kHz variant, single-GPU share
    # 128-mel, upsample [8,8,4,2] (hop 512), 16 mel frames -> 8192 samples, B=32 per GPU: same kernels, other shapes (ups2 = k8 s4).
    # Timed in the parity-grade mode and in fp16 storage (the type configs[4] names).
    cfg48 = checks48 = None
    if rank == 0 and world == 1 and not args.eager and not args.no_48k:
        torch.manual_seed(0)
        g48c = H.ModifiedHiFiGANGenerator(mel_channels=128, upsample_factors=[8, 8, 4, 2])
        sd48 = {k: v.detach().clone() for k, v in g48c.state_dict().items()}
        torch.manual_seed(1)
        m48 = torch.randn(B, 128, 16, device=dev)
        cfg48 = {"workload": "configs[4] per-GPU share: 128-mel 48 kHz generator, upsample [8,8,4,2], B=%d x 16 mel frames -> 8192 samples" % B,
                 "note": "no sub-fp32 storage / operand mix is inside north_star's 1e-3 on this geometry (tools/error_budget.py, table in DESIGN.md "
                         "section 5: fp16 through input_proj 9.3e-4 simulated - the GPU adds 10-15 % -, through up0 1.25e-3, two-product MRF "
                         "1.54e-3), so the parity-grade figure here is the all-fp32 one; fp16 storage is listed with parity_ok false"}
        checks48 = {}
        for tag in ("fp32", "fp16"):
            dt = getattr(torch, DT[tag])
            g48 = H.ModifiedHiFiGANGenerator(mel_channels=128, upsample_factors=[8, 8, 4, 2])
            g48.load_state_dict(sd48)
            g48 = g48.to(dev).to(dt).train(False)
            mm, ss, ee = m48.to(dt), spk32.to(dt), emo32.to(dt)
            with torch.no_grad():
                checks48[tag] = host(g48(mm[:2], ss[:2], ee[:2]))
            gv48 = GraphedVocoder(g48, mm, ss, ee)
            el48 = timed_replays(gv48.replay, 200)
            cfg48[tag] = {"value": round(B * 16 / el48, 1), "unit": "mel-frames/s", "samples_per_s": round(B * 8192 / el48, 1),
                          "ms_per_step": round(el48 * 1e3, 4), "launch": "hipgraph", "parity_rel_l2_vs_oracle": None, "parity_ok": None}
            del gv48, g48
        checks48 = (checks48, (host(m48[:2]), host(spk32[:2]), host(emo32[:2])), sd48)

    # ---------------------------------------------------------------- BASELINE configs[0]: plain HiFi-GAN V3 (no ODConv), batch 1
    # the reference's own CPU-runnable case (fairseq's generator: parity unpinned); GPU figure + the CPU restatement timed
    cfg_v3, check_v3 = None, None
    if rank == 0 and world == 1 and not args.no_v3:
        from hifigan_modified.plain_hifigan import PlainHiFiGANGenerator
        torch.manual_seed(0)
        v3 = PlainHiFiGANGenerator()
        sd_v3 = {k: v.detach().clone() for k, v in v3.state_dict().items()}
        v3 = v3.to(dev).train(False)
        torch.manual_seed(1)
        mel_v3 = torch.randn(1, 80, 344, device=dev)
        with torch.no_grad():
            w_v3 = v3(mel_v3)
            launch = "eager"
            run_v3 = lambda: v3(mel_v3)
            if not args.eager and hasattr(v3, "graphed"):
                run_v3 = v3.graphed(mel_v3)
                launch = "hipgraph"
            # same protocol as the headline leg: this leg follows CPU-bound sections (idle clocks: the driver's box measured 2x the
            # builder's figure with 3 warm replays): ~0.3 s of untimed replays, then the timed ones
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < 0.3:
                for _ in range(20):
                    run_v3()
                torch.cuda.synchronize()
            n_v3 = 200
            el_v3 = timed_replays(run_v3, n_v3) * n_v3
        check_v3 = (w_v3.cpu(), mel_v3.cpu(), sd_v3)
        cfg_v3 = {"workload": "configs[0]: plain HiFi-GAN V3 generator, B=1 x 344 mel frames (4 s at 22.05 kHz), fp32",
                  "value": round(344 * n_v3 / el_v3, 1), "unit": "mel-frames/s", "ms_per_step": round(el_v3 / n_v3 * 1e3, 3), "dtype": "fp32",
                  "launch": launch, "parity_rel_l2_vs_cpu_restatement": None, "cpu": None,
                  "note": "parity unpinned: fairseq generator absent from the reference tree"}
        del v3

    # ---------------------------------------------------------------- conditioning producers
    # ECAPA-TDNN + Emotion2Vec on the same mel batch (what ModifiedHiFiGANVocoder.forward(extract_embeddings=True) runs in front
    # of the generator): throughput of the captured forward, the dominant kernel's roofline, parity against the CPU oracle.
    conditioning, embed_check = None, None
    if rank == 0 and world == 1 and not args.eager and not args.no_conditioning:
        from hifigan_modified.graphs import GraphedExtractor
        from hifigan_modified import _native as _N
        cdt = torch.bfloat16
        melc = mel32.to(cdt)
        torch.manual_seed(0)
        ex = H.EmbeddingExtractor().to(dev).train(False)
        with torch.no_grad():
            for bn in [m for m in ex.modules() if isinstance(m, torch.nn.BatchNorm1d)]:      # non-trivial running statistics
                bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
            s2, e2 = ex(melc[:2])
        embed_check = ({k: v.detach().float().cpu() for k, v in ex.state_dict().items()}, host(melc[:2]), host(s2), host(e2))
        ge = GraphedExtractor(ex, melc)
        elE = timed_replays(ge.replay, 200)
        # dominant kernel: the split-K GEMM of the 512 -> 512 layers (27 of ~80 launches per forward)
        Mr, Kd, Nd = B * Tm, 512, 512
        xg = torch.randn(Mr, Kd, device=dev).to(cdt)
        wg = ops.dconv_pack((torch.randn(Nd, Kd, 1, 1, device=dev) / Kd ** 0.5), cdt, 0)
        bg = torch.zeros(Nd, device=dev, dtype=cdt)
        yg = torch.empty(Mr, Nd, device=dev, dtype=cdt)
        run = lambda: _N.call("mv_gemm_cl_skinny", ops._p(xg), ops._p(wg), ops._p(bg), ops._p(yg), Mr, Kd, Nd, _N.ACT_LRELU, 0.0,
                              ops._dt(xg), ops._stream())
        msg = graph_time_ms(run)
        gb = (Mr * Kd + Nd * Kd + Mr * Nd) * 2
        conditioning = {"metric": "mel-frames/s embedded (ECAPA-TDNN speaker + Emotion2Vec emotion encoders)",
                        "value": round(B * Tm / elE, 1), "unit": "mel-frames/s", "ms_per_step": round(elE * 1e3, 4), "dtype": "bf16",
                        "launch": "hipgraph", "parity_rel_l2_vs_oracle": None,
                        "roofline": {"bound": "hbm", "kernel": "mv::gemm_skinny_kernel<bf16,2> (%dx%dx%d, split-K over 8 waves)" % (Mr, Nd, Kd),
                                     "achieved": round(gb / (msg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(gb / (msg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "alg_bytes_per_launch": gb,
                                     "ms_per_launch": round(msg, 4),
                                     "note": "launch/latency-bound at %d rows: ~80 dependent launches of <= 10 us per forward" % Mr}}
        del ge, ex

    # ---------------------------------------------------------------- training metric (BASELINE configs[2]/[3])
    # full two-optimizer step (complete_vocoder.py:199-233): G forward -> D step -> G step, + mel/STFT loss, + AdamW;
    # data parallel: 32 clips per GPU; each optimizer's gradient buckets are all-reduced (RCCL) from grad-ready hooks while
    # its backward is still running (parallel.OverlappedGradSync)
    train = None
    if args.train_steps > 0:
        from hifigan_modified.parallel import broadcast_parameters

        def train_leg(dtag, steps, warm, graph):
            """K timed steps of VocoderTrainer (G forward, D step, G step, mel/STFT loss, both AdamW) at activation type `dtag`."""
            tdt = getattr(torch, DT[dtag])
            torch.manual_seed(0)
            voc_ = H.ModifiedHiFiGANVocoder().to(dev)
            if world > 1:
                broadcast_parameters(voc_)
            tr_ = H.VocoderTrainer(voc_, device=dev, use_graph=graph)       # grad_sync defaults to "overlap" when world > 1
            torch.manual_seed(100 + rank)
            tmel = torch.randn(B, 80, Tm, device=dev).to(tdt)
            treal = torch.randn(B, 1, Tm * 256, device=dev).clamp(-1, 1).to(tdt)
            tspk, temo = torch.randn(B, 192, device=dev).to(tdt), torch.randn(B, 384, device=dev).to(tdt)
            torch.manual_seed(2 + rank)      # dropout stream of this rank
            for _ in range(warm + (tr_.graph_warmup + 1 if graph else 0)):   # (graph: eager warm-up steps, the capture, then `warm` replays)
                losses_ = tr_.train_step(tmel, treal, tspk, temo, return_tensors=True)
            ovs_ = [o for o in (tr_.overlap_sync(tr_.generator_optimizer), tr_.overlap_sync(tr_.discriminator_optimizer)) if o is not None]
            for o in ovs_:
                o.timing, o.exposed_ms, o.reduced_bytes, o._ev = True, 0.0, 0, []
            sync_all()
            t0_ = time.perf_counter()
            for _ in range(steps):
                losses_ = tr_.train_step(tmel, treal, tspk, temo, return_tensors=True)
            sync_all()
            tel_ = time.perf_counter() - t0_
            if world > 1:
                t_ = torch.tensor([tel_], device=dev, dtype=torch.float64)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                tel_ = float(t_.item())
            return tel_, tr_.to_floats(losses_), ovs_, voc_, tr_

        use_graph = world == 1 and not args.train_eager and not args.eager
        tdtype = getattr(torch, DT[args.train_dtype])
        graph_note = None
        try:
            tel, lf, ovs, voc, trainer = train_leg(args.train_dtype, args.train_steps, args.train_warmup, use_graph)
        except Exception as exc:        # a capture that fails must not cost the metric: time the step as issued
            if not use_graph:
                raise
            graph_note = "graph capture failed (%s: %s); timed eagerly" % (type(exc).__name__, str(exc)[:200])
            print("bench.py: " + graph_note, file=sys.stderr)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            use_graph = False
            tel, lf, ovs, voc, trainer = train_leg(args.train_dtype, args.train_steps, args.train_warmup, False)
        coll = ("RCCL (torch.distributed 'nccl' backend over xGMI)" if backend == "nccl" else "torch.distributed '%s' backend (a rehearsal: NOT RCCL)" % backend)
        train = {"metric": "train audio-samples/s (G fwd + D step + G step + mel/STFT loss + AdamW)",
                 "value": round(B * Tm * 256 * world * args.train_steps / tel, 1), "unit": "samples/s",
                 "ms_per_step": round(tel / args.train_steps * 1e3, 2), "steps": args.train_steps, "dtype": args.train_dtype,
                 "launch": "hipgraph (the whole step - ~1000 launches - captured after 2 eager steps and replayed)" if use_graph else "eager",
                 **({"launch_note": graph_note} if graph_note else {}),
                 "global_batch": B * world, "losses_finite": all(x == x and abs(x) != float("inf") for x in lf.values()),
                 "parallelism": ("dp%d, gradient buckets all-reduced under the backward over %s" % (world, coll)) if world > 1 else "single GPU"}
        if world > 1 and ovs:
            # achieved all-reduce bandwidth, measured standalone on the generator's flat gradient buffer (the step's largest exchange)
            buf = trainer.generator_optimizer.flat_g
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            ar_ms = e0.elapsed_time(e1) / 10
            train["allreduce_standalone"] = {"bytes": buf.numel() * 4, "ms": round(ar_ms, 4), "algbw_GBps": round(buf.numel() * 4 / ar_ms / 1e6, 1),
                                             "busbw_GBps": round(buf.numel() * 4 / ar_ms / 1e6 * 2 * (world - 1) / world, 1), "backend": backend}
        if rank == 0 and world == 1 and use_graph:
            # the same step issued eagerly, and the other activation types (fp32 = split-operand kernels, fp16), beside the headline
            del trainer
            torch.cuda.empty_cache()
            te, _, _, voc_e, tr_e = train_leg(args.train_dtype, max(2, args.train_steps // 2), 1, False)
            train["eager_ms_per_step"] = round(te / max(2, args.train_steps // 2) * 1e3, 2)
            del tr_e, voc_e
            if not args.no_train_alt:
                for alt in ("fp32", "fp16", "bf16"):
                    if alt == args.train_dtype:
                        continue
                    torch.cuda.empty_cache()
                    ta_, lfa, _, voc_a, tr_a = train_leg(alt, 3, 1, True)
                    train["train_" + alt] = {"value": round(B * Tm * 256 * 3 / ta_, 1), "unit": "samples/s", "ms_per_step": round(ta_ / 3 * 1e3, 2),
                                             "losses_finite": all(x == x and abs(x) != float("inf") for x in lfa.values())}
                    del tr_a, voc_a
                train["precision_note"] = ("activations in the named type, fp32 master weights / AdamW / weight gradients.  Small-model trajectories "
                                           "(tests/test_gpu_train_graph.py): fp16 follows the fp32 run within 0.7 % (G) / 0.05 % (D) over the first "
                                           "five steps and 6 % over twelve; bf16 3 % / 1 % and has left the fp32 trajectory by step 8")
            torch.cuda.empty_cache()
        if ovs:
            exposed = sum(o.collect_timing() for o in ovs) / args.train_steps
            nbytes = sum(o.reduced_bytes for o in ovs) / args.train_steps
            train["allreduce"] = {"bytes_per_step": int(nbytes), "buckets": sum(len(o.buckets) for o in ovs),
                                  "exposed_wait_ms_per_step": round(exposed, 3),
                                  "note": "exposed = time the compute stream waited for the collectives after the backward (what the overlap did not hide)"}
        if rank == 0 and tdtype != torch.float32:
            # MFMA roofline of the dominant discriminator conv (128 -> 256, 3x3, period-2 fold of the real+fake batch)
            import ctypes
            from hifigan_modified import _native as _N, disc_fused as _df
            conv = voc.discriminators.mpd.discriminators[0].conv_layers[6]
            Bd, Hd, Wd = 2 * B, 2, Tm * 256 // 2
            xin = torch.randn(Bd, Hd, Wd, 128, device=dev).to(tdtype)
            yout = torch.empty(Bd, Hd, Wd, 256, device=dev, dtype=tdtype)
            pk = _df._packs.get(conv.weight, tdtype, 0)   # keyed on the live parameter
            bias = conv.bias.detach().to(tdtype)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            def run():
                _N.call("mv_dconv_cl_fwd", P(xin), P(pk), P(bias), None, P(yout), Bd, Hd, Wd, 128, 256, 3, 3, 1, _N.ACT_LRELU, 0.1,
                        ops._dt(xin), ops._stream())
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            flops = 2.0 * Bd * Hd * Wd * 256 * 128 * 9
            ach = flops / (ms * 1e-3) / 1e12
            tname = {"bf16": "__hip_bfloat16", "fp16": "_Float16"}[args.train_dtype]
            # average over every launch of this instantiation in a training step (the 128->256 forward and the data-gradient launches it also serves)
            ttr = pmc_traffic("dconv_cl_wide_kernel<%s, 1, 4, 8, 64, 4>" % tname, "train_" + args.train_dtype,
                              files=("disc_fused.hip", "mfma.h", "common.h"))
            train["roofline"] = {"bound": "mfma", "kernel": "mv::dconv_cl_wide_kernel<%s,1,4,8,64,4> (Conv2d 128->256 3x3 + LeakyReLU, implicit GEMM, 4 waves: 256 rows x 128 positions, 2 workgroups per CU)" % args.train_dtype,
                                 "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
                                 "traffic": ttr["bytes"] if ttr else None, "flops_per_launch": flops, "ms_per_launch": round(ms, 4)}

    if rank == 0:
        frames = B * Tm * world * args.steps
        precision = {"mixed": "fp16 storage and MFMA operands through %s (prologue, input_proj + FiLM, the first upsamplers), then fp32 storage: the "
                              "last upsamplers and the output conv with every MFMA operand split into hi + lo bf16 (3 products per MAC), the three MRF "
                              "blocks with hi + lo f16 activation operands x %s; fp32 accumulate throughout"
                              % (MIXED_THROUGH, "single f16 weights (2 products per MAC)" if MIXED_MRF_W16 else "hi + lo bf16 weights (3 products)"),
                     "fp32": "fp32 storage; every MFMA operand split into hi + lo bf16 (3 products per MAC), fp32 accumulate",
                     "bf16": "bf16 storage and MFMA operands, fp32 accumulate", "fp16": "fp16 storage and MFMA operands, fp32 accumulate"}[args.dtype]
        out = {
            "metric": "mel-frames/s vocoded (V1 80-mel 22.05kHz generator, ODConv + GRC-LoRA" + ("; mixed fp16 + fp32 storage: the fastest "
                      "mode inside north_star's 1e-3 - BASELINE's bf16 storage is reported under bf16_storage with parity_ok false)" if args.dtype == "mixed" else ")"),
            "value": round(frames / elapsed, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16+fp32" if args.dtype == "mixed" else args.dtype, "precision": precision, "data": "synthetic",
            "config": {"workload": "configs[1]: V1 generator + ODConv1d + GRC-LoRA, B=%d x %d mel frames -> %d samples, inference"
                                   % (B, Tm, Tm * 256), "batch_per_gpu": B, "mel_frames": Tm, "n_mels": 80,
                       "parallelism": "replicas (batch-sharded, no collective)"},
            "samples_per_s": round(frames * 256 / elapsed, 1),
            "parity_rel_l2_vs_oracle": None, "parity_tol": PARITY_TOL, "parity_ok": None,
            "launch": "eager" if args.eager else "hipgraph",
            "roofline": roof,
            "roofline_odconv": od_roof,
            "modes": modes,
            "train": train,
            "conditioning": conditioning,
            "config_48k": cfg48,
            "config_v3_plain": cfg_v3,
        }
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(sd_cpu, max(1, (os.cpu_count() or 2) // 2), checks=checks, embed_check=embed_check,
                              check48=checks48, check_v3=check_v3)
            par = cb.pop("parity", {})
            ok = lambda v: None if v is None else bool(v <= PARITY_TOL)
            if cfg48 is not None:
                for tag in ("fp32", "fp16"):
                    cfg48[tag]["parity_rel_l2_vs_oracle"] = par.get("config_48k_" + tag)
                    cfg48[tag]["parity_ok"] = ok(par.get("config_48k_" + tag))
            if cfg_v3 is not None:
                cfg_v3["parity_rel_l2_vs_cpu_restatement"] = par.get("config_v3_plain")
                cfg_v3["cpu"] = cb.pop("config_v3_plain", None)
            if conditioning is not None:
                conditioning["parity_rel_l2_vs_oracle"] = par.get("conditioning")
                conditioning["cpu_baseline"] = cb.pop("conditioning", None)
            out["parity_rel_l2_vs_oracle"] = par.get("headline")
            out["parity_ok"] = ok(par.get("headline"))
            for tag in (modes or {}):
                modes[tag]["parity_rel_l2_vs_oracle"] = par.get("mode_" + tag)
                modes[tag]["parity_ok"] = ok(par.get("mode_" + tag))
            out["cpu_baseline"] = cb
        if modes and "bf16" in modes:      # the storage type BASELINE configs[1] names, first-class beside the headline
            out["bf16_storage"] = {k: modes["bf16"][k] for k in ("value", "unit", "ms_per_step", "parity_rel_l2_vs_oracle", "parity_ok")}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
