#!/usr/bin/env python3
"""Shader-chain throughput benchmark (driver contract: see the task description).

A "step" = one pass of the hot path (every pass of the preset) over one batch of synthetic
frames that are already resident in HBM.  One process per GPU; frames are independent, so N
GPUs shard the batch with no data-path collective (weak scaling: --batch frames per GPU per
step).  torch is used for device memory, the stream and the barrier / MAX-over-ranks reduce.

Launch: `python bench.py --gpus N ...` starts N rank processes itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set per child, before anything touches the GPU in the parent, which only relays rank 0's line);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the ranks already exist and
--gpus must equal WORLD_SIZE.  Either way a rank fails loudly if it has no device of its own.

Prints ONE JSON line on rank 0 with the BASELINE metric plus `roofline` (dominant kernel,
HIP-event timed on the launch stream), `cpu_baseline` (the oracle, timed on the host cores
on a bounded sample of the same workload; reported baseline only) and, for crt-royale, `mask_rendered`:
the same measurement with pass 6's unwritten varying read as 0, i.e. what GPU GL drivers render (the
default is what Mesa llvmpipe renders: that pass discards every fragment and the phosphor mask stays black).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); measured copy ~6300

WORKLOADS = {
    # key: (chain_specs preset key, source w, h, viewport w, h, description)
    "crt-royale": ("crt-royale", 1920, 1080, 1920, 1080, "crt/crt-royale.glslp 12-pass, 1920x1080 RGBA8 frames"),
    "crt-royale-fake-bloom": ("crt-royale-fake-bloom", 1920, 1080, 1920, 1080,
                              "crt/crt-royale-fake-bloom.glslp 9-pass, 1920x1080 RGBA8 frames"),
    "crt-hyllian-glow": ("crt-hyllian-glow", 1920, 1080, 1920, 1080,
                         "crt/crt-hyllian-glow.glslp 6-pass (the reference's smoke-test default), 1920x1080 RGBA8 frames"),
    "crt-easymode": ("crt-easymode", 1920, 1080, 1920, 1080, "crt/crt-easymode.glslp 1-pass, 1920x1080 RGBA8 frames"),
    "zfast-crt": ("zfast-crt", 1920, 1080, 1920, 1080, "crt/zfast-crt.glslp 1-pass, 1920x1080 RGBA8 frames"),
    "crt-pi": ("crt-pi", 1920, 1080, 1920, 1080, "crt/crt-pi.glslp 1-pass, 1920x1080 RGBA8 frames"),
    "ntsc": ("ntsc-256px-svideo", 1920, 1080, 1920, 1080,
             "ntsc/ntsc-256px-svideo.glslp 2-pass (RGBA32F 1024x1080 intermediate), 1920x1080 RGBA8 frames"),
    "xbr-lv3": ("xbr-lv3", 256, 224, 3840, 2160, "xbr/xbr-lv3.glslp 1-pass upscale 256x224 -> 3840x2160"),
    "xbr-lv2": ("xbr-lv2", 256, 224, 3840, 2160, "xbr/xbr-lv2.glslp 1-pass upscale 256x224 -> 3840x2160"),
    "scanline": ("scanline", 320, 240, 320, 240, "scanlines/shaders/scanline.glsl 1-pass, 320x240"),
    "crt-geom": ("crt-geom", 640, 480, 1920, 1440, "crt/crt-geom.glslp 1-pass (curvature, interlacing simulation on), 640x480 -> 1920x1440"),
    "scalefx": ("scalefx", 256, 224, 768, 672, "scalefx/scalefx.glslp 5-pass pixel-art upscale 256x224 -> 768x672"),
    "crt-lottes": ("crt-lottes", 640, 480, 1920, 1440, "crt/crt-lottes.glslp 1-pass (instruction-list kernel, ~2 500 operations / pixel), 640x480 -> 1920x1440"),
    "tvout": ("tvout+ntsc-256px-svideo", 256, 224, 1280, 960,
              "presets/tvout/tvout+ntsc-256px-svideo.glslp 4-pass (ntsc 3-phase, tvout-tweaks, image-adjustment), 256x224 -> 1280x960"),
    "lcd-grid-v2": ("lcd-grid-v2-gba-color-motionblur", 240, 160, 1920, 1280,
                    "handheld/lcd-grid-v2-gba-color-motionblur.glslp 3-pass (frame history: sequential frames), 240x160 -> 1920x1280"),
}


def shard_frames(total_frames, rank, world):
    """Contiguous block of frame indices owned by `rank` (frames are independent: no exchange)."""
    per = (total_frames + world - 1) // world
    return range(min(total_frames, rank * per), min(total_frames, (rank + 1) * per))


def aggregate(frames_per_rank, steps, seconds):
    """Whole-job throughput: frames all ranks processed / max-over-ranks time."""
    return sum(frames_per_rank) * steps / seconds


def default_workload():
    import chain_specs
    return "crt-royale" if "crt-royale" in chain_specs.PRESETS else "crt-pi"


def cpu_baseline(key, w, h, vw, vh, tree, budget_s=15.0, custom=None, luts=None, f16_targets=False):
    """Oracle (C restatement) on the host cores, rows of each pass split across threads."""
    import numpy as np
    import chain_specs
    import oracle_chain
    import oracle_lib
    from retrocapture_amd import engine as eng
    # the GPU box gives a one-GPU job 16 host cores; never start more workers than that
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
    passes = eng.preset_dump(tree[key])["passes"]
    rng = np.random.default_rng(123)
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    oracle_lib.set_threads(cores)
    kw = {"custom": custom or None, "luts": luts, "f16_targets": f16_targets}
    oracle_chain.run_chain(passes, frame[: max(8, h // 16)], vw, max(8, vh // 16), **kw)  # warm (page in)
    n, t0 = 0, time.perf_counter()
    while True:
        oracle_chain.run_chain(passes, frame, vw, vh, frame_count=n + 1, **kw)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 32:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d full frame(s) of the same workload through oracle/liboracle.so, %d threads by rows, %.1f s"
                      % (n, cores, dt)}


PMC_FILE = "r04_royale_pmc.csv"   # the committed counter summary the `traffic` / `valu` figures are read from (profiles/collect.sh) ...
PMC_FRAMES_PER_LAUNCH = 128.0     # ... and the launch shape it was collected at: the engine's default for 1080p chains
STATS_FILE = "r04_royale_kernel_stats.csv"   # the per-kernel durations of that collection (one lane): pmc_fresh


def pmc_traffic(kernel_name, frames_per_launch):
    """HBM-side bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary
    (profiles/r*_royale_pmc.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, KiB).
    Correction per /opt/skills/guides/MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the
    bytes of coalesced streaming reads (re-checked here on the frame-I/O kernels, whose byte counts
    are known: 4, 8, 12 and 16 B per lane all read exactly 1/2), WRITE_SIZE is exact.  Only valid
    for the launch shape it was collected at (PMC_FRAMES_PER_LAUNCH); otherwise None."""
    import csv
    import glob
    files = glob.glob(os.path.join(ROOT, "profiles", PMC_FILE))
    if not files or abs(frames_per_launch - PMC_FRAMES_PER_LAUNCH) > 1e-6:
        return None
    want = "k_" + kernel_name.replace("-", "_")
    alias = {"k_royale_scanlines_v": "k_royale_scan_v", "k_royale_scanlines_h": "k_royale_scan_h",
             "k_royale_bloom_horizontal": "k_royale_bloom_h", "k_royale_bloom_vertical": "k_royale_bloom_v"}
    want = alias.get(want, want)
    rows = {r["kernel"]: r for r in csv.DictReader(open(files[-1]))}
    for name in (want + "_quad", want + "_strip", want + "_tab", want + "2", want):      # the form the fast path launches
        r = rows.get(name)
        if r and r.get("FETCH_SIZE_avg") and r.get("WRITE_SIZE_avg"):
            return (2.0 * float(r["FETCH_SIZE_avg"]) + float(r["WRITE_SIZE_avg"])) * 1024.0
    return None


def pmc_fresh(kernel_name, avg_launch_ms):
    """The committed counters describe the kernel that is running now only while that kernel takes what it took when they
    were collected: the live launch duration against the committed rocprofv3 summary of the same collection
    (profiles/r04_royale_kernel_stats.csv), within 15 %.  Otherwise `traffic` / `valu` are withheld (null) rather than replayed."""
    import csv
    path = os.path.join(ROOT, "profiles", STATS_FILE)
    if not os.path.exists(path) or avg_launch_ms <= 0:
        return False
    want = {"royale-scanlines-v": "k_royale_scan_v", "royale-bloom-h": "k_royale_bloom_h"}.get(kernel_name, "k_" + kernel_name.replace("-", "_"))
    rows = {r["kernel"]: r for r in csv.DictReader(open(path))}
    for name in (want + "_quad", want + "_strip", want + "_tab", want + "2", want):
        r = rows.get(name)
        if r and r.get("avg_ns"):
            then_ms = float(r["avg_ns"]) * 1e-6
            if rows.get(want + "_fix", {}).get("avg_ns"):   # (the pass's launch duration covers its fix-up kernel too)
                then_ms += float(rows[want + "_fix"]["avg_ns"]) * 1e-6
            return abs(avg_launch_ms - then_ms) <= 0.15 * then_ms
    return False


def pmc_valu(kernel_name, frames_per_launch, avg_launch_ms):
    """The dominant kernel's real limiter when it is not HBM: VALU wave-instructions per launch from the same
    committed PMC summary (SQ_INSTS_VALU, SQ_INSTS_LDS) over the live launch time, per SIMD (256 CUs x 4), against the
    issue rate of this device's SLOW instruction class (4.4 cycles per wave instruction = 0.53 G/s/SIMD; fast-class
    instructions issue in 2.5 cycles, packed ones in 4.8, a ds_read_b128 occupies the CU's LDS pipe for 4.4, and LDS time
    ADDS to VALU time instead of overlapping: profiles/micro/*.hip)."""
    import csv
    import glob
    files = glob.glob(os.path.join(ROOT, "profiles", PMC_FILE))
    if not files or abs(frames_per_launch - PMC_FRAMES_PER_LAUNCH) > 1e-6 or avg_launch_ms <= 0:
        return None
    want = {"royale-scanlines-v": "k_royale_scan_v", "royale-bloom-h": "k_royale_bloom_h"}.get(kernel_name, "k_" + kernel_name.replace("-", "_"))
    rows = {r["kernel"]: r for r in csv.DictReader(open(files[-1]))}
    for name in (want + "_quad", want + "_strip", want + "_tab", want + "2", want):
        r = rows.get(name)
        if r and r.get("SQ_INSTS_VALU_avg"):
            insts = float(r["SQ_INSTS_VALU_avg"])
            rate = insts / (avg_launch_ms * 1e-3) / (256 * 4) / 1e9
            out = {"wave_insts_per_launch": insts, "G_wave_insts_per_s_per_simd": rate, "issue_ceiling_G_per_s_per_simd": 0.53,
                   "frac_of_issue_ceiling": rate / 0.53,
                   "issue_cost_cycles_measured": {"fast": 2.5, "slow": 4.4, "packed": 4.8, "transcendental": 8.1, "ds_read_b128_per_cu": 4.4},
                   "note": "not HBM-bound: VALU issue and LDS pipe time add up in this kernel (the GL's float arithmetic in the GL's order, "
                           "bit-exact); the ceiling quoted is the slow instruction class's, see DESIGN.md section 7"}
            if r.get("SQ_INSTS_LDS_avg"):
                out["lds_wave_insts_per_launch"] = float(r["SQ_INSTS_LDS_avg"])
            return out
    return None


def copy_ceiling(torch, mib=1024, reps=20):
    """Achieved stream-copy rates of this box (SURVEY.md section 8d asks for one beside the 8 TB/s vendor peak), read + write
    bytes over HIP-event time of a `mib` MiB device-to-device copy: (1) the library's own 16-byte-per-lane grid-stride copy
    kernel (rc_selftest_copy_rate: the shape the hardware guide measures at 6.3 TB/s) - the ceiling a kernel can reach; (2)
    the runtime's memcpy (torch Tensor.copy_ = __amd_rocclr_copyBuffer), which rounds 1-3 quoted and which is slower."""
    from retrocapture_amd import engine as eng
    kernel = eng.copy_rate(torch.cuda.current_device(), mib, reps)
    n = mib << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    a.fill_(7)
    for _ in range(3):
        b.copy_(a)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        b.copy_(a)
    t1.record()
    torch.cuda.synchronize()
    memcpy = 2.0 * n * reps / (t0.elapsed_time(t1) * 1e-3) / 1e9
    del a, b
    return kernel, memcpy


def io_measurements(e, w, h, batch, reps):
    """Side measurements: (1) the ingest / egress kernels alone, algorithmic bytes / event time;
    (2) the whole host-to-host frame path the reference runs per frame (FrameProcessor upload ->
    chain -> readback): pinned RGB24 in, H2D, ingest, chain, egress, D2H, pinned RGB24 out."""
    import torch
    from retrocapture_amd import engine as eng
    res = {}
    n = batch
    px = w * h * n
    src = {f: torch.randint(0, 256, (px * b,), dtype=torch.uint8, device="cuda")
           for f, b in (("rgb24", 3), ("bgra", 4), ("yuyv422", 2))}
    rgba = torch.empty(px * 4, dtype=torch.uint8, device="cuda")
    rgb = torch.empty(px * 3, dtype=torch.uint8, device="cuda")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    st = torch.cuda.current_stream().cuda_stream
    for f, b in (("rgb24", 3), ("bgra", 4), ("yuyv422", 2)):
        t = timed(lambda: eng.ingest(src[f], f, w, h, n, rgba, stream=st))
        res["ingest_" + f] = {"GB/s": px * (b + 4) / t / 1e9, "frac_of_8TBs": px * (b + 4) / t / 8e12, "frames": n, "w": w, "h": h}
    t = timed(lambda: eng.egress_rgb24(rgba, w, h, n, rgb, stream=st))
    res["egress_rgb24"] = {"GB/s": px * 7 / t / 1e9, "frac_of_8TBs": px * 7 / t / 8e12, "frames": n, "w": w, "h": h}

    # rc_present (OpenGLRenderer::renderTexture off-screen): algorithmic bytes = source once + target once
    dw, dh = w * 2 // 3, h * 2 // 3
    small = torch.empty(n * dw * dh * 4, dtype=torch.uint8, device="cuda")
    for name, kw, rd, wr in (
            ("present_bake_rgba8", dict(dst_w=w, dst_h=h, brightness=1.1, contrast=0.9), 4, px * 4),
            ("present_resize_bake_rgb24", dict(dst_w=dw, dst_h=dh, dst_kind="rgb24", bake=(1.1, 0.9), out_flip_rows=True), 4, n * dw * dh * 3),
            ("present_prepass_nearest_rgbx8", dict(dst_w=dw, dst_h=dh, dst_kind="rgbx8", src_rgb=True, src_linear=False,
                                                   viewport=eng.overscan_viewport(dw, dh, 5.0, 5.0)), 4, n * dw * dh * 4)):
        dst = rgb if kw.get("dst_kind") == "rgb24" else (small if kw["dst_w"] != w else torch.empty_like(rgba))
        kk = dict(kw)
        tw, th = kk.pop("dst_w"), kk.pop("dst_h")
        t = timed(lambda: eng.present(rgba, w, h, dst, tw, th, n_frames=n, stream=st, **kk))
        res[name] = {"GB/s": (px * rd + wr) / t / 1e9, "frac_of_8TBs": (px * rd + wr) / t / 8e12, "frames": n,
                     "src": [w, h], "dst": [tw, th], "us_per_frame": t / n * 1e6}

    # host-to-host: the chain's own output size may differ from the input's
    h_in = torch.randint(0, 256, (px * 3,), dtype=torch.uint8).pin_memory()
    d_in = torch.empty(px * 3, dtype=torch.uint8, device="cuda")
    e.applyShaderBatch(rgba.view(n, h, w, 4), n, w, h)
    ow, oh = e.getOutputWidth(), e.getOutputHeight()
    d_out = torch.empty(n * ow * oh * 3, dtype=torch.uint8, device="cuda")
    h_out = torch.empty(n * ow * oh * 3, dtype=torch.uint8).pin_memory()

    def frame_path():
        d_in.copy_(h_in, non_blocking=True)
        eng.ingest(d_in, "rgb24", w, h, n, rgba, stream=st)
        ptr, _, _ = e.applyShaderBatch(rgba, n, w, h)
        eng.egress_rgb24(ptr, ow, oh, n, d_out, stream=st)
        h_out.copy_(d_out, non_blocking=True)

    frame_path()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        frame_path()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # the same path through the library's own 3-stream ring (rc_pipeline_*): copies overlap kernels
    pipe = eng.FramePipeline(e, slots=4)
    h_frames = [h_in[k * w * h * 3:(k + 1) * w * h * 3].numpy() for k in range(n)]

    def piped():
        for fr in h_frames:
            while not pipe.submit(fr, "rgb24", w, h):
                if pipe.inFlight() == 0:      # not a full ring: a real failure (rc_last_error has the text)
                    raise SystemExit("rc_pipeline_submit failed with nothing in flight")
                pipe.receive(wait=True)
        while pipe.inFlight():
            pipe.receive(wait=True)

    piped()
    t0 = time.perf_counter()
    for _ in range(reps):
        piped()
    dtp = (time.perf_counter() - t0) / reps

    def piped_zero_copy():   # the producer writes straight into the pipeline's pinned staging
        for _k in range(n):
            buf = pipe.inputBuffer("rgb24", w, h)
            while buf is None:
                if pipe.inFlight() == 0:
                    raise SystemExit("rc_pipeline_input_buffer failed with nothing in flight")
                pipe.receive(wait=True)
                buf = pipe.inputBuffer("rgb24", w, h)
            pipe.submit(buf, "rgb24", w, h)
        while pipe.inFlight():
            pipe.receive(wait=True)

    piped_zero_copy()
    t0 = time.perf_counter()
    for _ in range(reps):
        piped_zero_copy()
    dtz = (time.perf_counter() - t0) / reps
    pipe.close()
    res["host_to_host_pipelined_pinned_input"] = {"frames_per_s": n / dtz, "slots": 4,
                                                  "note": "as above, frames produced directly in rc_pipeline_input_buffer"}
    res["host_to_host_pipelined"] = {"frames_per_s": n / dtp, "slots": 4,
                                     "note": "rc_pipeline_*: unpinned caller buffers -> pinned staging -> H2D / kernels / "
                                             "D2H on three streams, one frame per submit"}
    res["host_to_host"] = {"frames_per_s": n / dt, "bytes_in_per_frame": w * h * 3, "bytes_out_per_frame": ow * oh * 3,
                           "pcie_GB/s": n * (w * h * 3 + ow * oh * 3) / dt / 1e9,
                           "note": "one stream, copies not overlapped with compute; PCIe-inclusive, never `value`"}
    return res


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: one fresh child process per rank (this parent has not imported torch
    and never touches a GPU).  Rank 0's stdout (the JSON line) is relayed; any rank's failure fails the run."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting in a rendezvous or a barrier: stop everybody
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.kill()             # the exact children this process started
            break
        time.sleep(0.1)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write((out0[0] if out0 else b"").decode())
    sys.stdout.flush()
    if failed or any(codes):
        raise SystemExit("bench.py: a rank failed (exit codes %s): no result" % codes)
    return 0


def dry_run(args):
    """The multi-rank control flow on the CPU (gloo): shard the global batch, barrier, time, MAX over ranks, rank 0
    prints the line.  No engine, no GPU - what tests/test_sharding_gloo.py drives."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_frames(args.batch * world, rank, world)      # weak scaling: --batch frames per rank
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    done = 0
    for _ in range(args.steps):
        done += len(mine)                                      # stands in for applyShaderBatch over this rank's shard
        time.sleep(0.001 * (rank + 1))
    own = time.perf_counter() - t0                             # this rank's own work, before it waits for the others
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    counts = [len(mine)]
    per_rank = [own]
    if world > 1:
        pr = torch.zeros(world, dtype=torch.float64)
        pr[rank] = own
        dist.all_reduce(pr)
        per_rank = [float(v) for v in pr.tolist()]
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.zeros(world, dtype=torch.int64)
        c[rank] = len(mine)
        dist.all_reduce(c)
        counts = [int(v) for v in c]
        assert dist.get_world_size() == args.gpus
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": aggregate(counts, args.steps, dt), "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "frames_per_rank": counts, "scaling": "weak",
                          "ms_per_step": dt / args.steps * 1e3, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
                          "world_size_observed": dist.get_world_size() if world > 1 else 1,
                          "data": "none (dry run: launch / shard / reduce path only)"}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150, help="timed steps (default: 150 x 256 frames = about 2.4 s of GPU work per measured mode at 1080p)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (256 x 1080p = 2.1 GB in, 2.1 GB out)")
    ap.add_argument("--chunk", type=int, default=0, help="frames per kernel launch (0 = engine default)")
    ap.add_argument("--lanes", type=int, default=2, choices=[1, 2],
                    help="rc_engine_set_lanes: 2 (the engine's default) renders the second half of every batch on a second HIP stream")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of host work for the cpu_baseline sample")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE",
                    help="set a shader parameter before the run (side measurement, e.g. geom_mode_runtime=1: crt-royale's curved last pass)")
    ap.add_argument("--fp16-targets", action="store_true",
                    help="store float_framebuffer targets as binary16 (rc_engine_set_float_target_fp16; side measurement "
                         "for the ntsc workload: 28.2 MB instead of 45.9 MB of algorithmic bytes per frame, output within "
                         "one 8-bit step of the default fp32 path)")
    ap.add_argument("--io", action="store_true",
                    help="also time the ingest / egress kernels and the whole host-to-host frame path (side "
                         "measurements under \"io\"; never part of `value`)")
    ap.add_argument("--modes", default="both", choices=["both", "default", "mask"],
                    help="crt-royale: which behaviour of pass 6's unwritten varying to measure - both (the default line with its "
                         "`mask_rendered` object), `default` only, or `mask` only (mask rendered; for per-mode rocprofv3 kernel statistics: "
                         "then `value` IS the mask-rendered figure and config.mask_mode says so)")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the launch / sharding / reduction path without a GPU (gloo, no engine): for tests")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)          # the parent never imports torch or touches a GPU
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: launch one rank per GPU (python bench.py --gpus N, or "
                         "torch.distributed.run --nproc-per-node N bench.py --gpus N)" % (args.gpus, world_env))
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    import chain_specs
    from retrocapture_amd import ShaderEngine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal switch for a one-GPU box: RC_BENCH_REHEARSAL=1 runs every rank on GPU 0 with the gloo
    # backend (RCCL refuses two ranks on one device), to exercise the multi-rank control flow only.
    rehearsal = os.environ.get("RC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if not rehearsal and torch.cuda.device_count() < max(world, local + 1):
        # one rank per GPU of ONE node: every rank sees the same count, so every rank stops here (no half-started job)
        raise SystemExit("bench.py: rank %d of %d has no GPU of its own (%d device(s) visible): %d ranks requested, %d device(s)"
                         % (rank, world, torch.cuda.device_count(), world, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    wl = args.workload or default_workload()
    key, w, h, vw, vh, desc = WORKLOADS[wl]
    tmp = tempfile.TemporaryDirectory()
    tree = chain_specs.write_tree(tmp.name)

    stream = torch.cuda.current_stream()
    e = ShaderEngine()
    if not e.init(local, stream.cuda_stream):
        raise SystemExit("ShaderEngine.init failed (no HIP device; there is no CPU fallback)")
    e.setAllowMissingSources(True)
    e.setAsyncTableBuilds(False)   # the timed region must not start on the general forms while a table build is still running
    st = e.loadPresetStatus(tree[key])
    if st != 0:
        raise SystemExit("preset %s not fully supported (status %d)" % (key, st))
    e.setViewport(vw, vh)
    if args.fp16_targets:
        e.setFloatTargetFp16(True)
    for kv in args.param:
        name, value = kv.split("=")
        if not e.setShaderParameter(name, float(value)):
            raise SystemExit("unknown shader parameter " + name)
    if args.chunk:
        e.setChunkFrames(args.chunk)
    e.setLanes(args.lanes)
    if args.modes == "mask":
        e.setUndefinedVaryingZero(True)

    # synthetic frames, resident in HBM before the timed region: uniform noise, alpha 255
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    # this rank's contiguous block of the global batch (weak scaling: --batch frames per rank); no frame is exchanged
    mine = shard_frames(args.batch * world, rank, world)
    n_local = len(mine)
    frames = torch.randint(0, 256, (n_local, h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
    frames[..., 3] = 255

    def step():
        e.applyShaderBatch(frames, n_local, w, h)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0        # this rank's own work, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ranks(seconds):
        """every rank's own time (so that a straggler shows in the one line the driver keeps), and the world size the
        communicator itself reports"""
        if world == 1:
            return [seconds], 1
        t = torch.zeros(world, dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        t[rank] = seconds
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()], int(dist.get_world_size())

    per_rank_s, comm_world = gather_ranks(own)
    dt = max_over_ranks(dt)
    if world > 1:
        assert dist.get_world_size() == args.gpus

    # per-kernel timing with HIP events on the launch stream, over the same steps again
    e.setProfiling(True)
    for _ in range(args.steps):
        step()
    prof = [e.passProfile(i) for i in range(e.passCount())]
    infos = [e.passInfo(i) for i in range(e.passCount())]
    e.setProfiling(False)
    dom = max(range(len(prof)), key=lambda i: prof[i]["total_ms"])
    p = prof[dom]
    frames_per_launch = p["frames"] / max(1, p["launches"])
    bytes_per_launch = (p["read_bytes_per_frame"] + p["write_bytes_per_frame"]) * frames_per_launch
    avg_ms = p["total_ms"] / max(1, p["launches"])
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    chain_bytes = sum(q["read_bytes_per_frame"] + q["write_bytes_per_frame"] for q in prof)
    # ... of which the passes that were launched move (a pass folded into its consumers - crt-royale's pass 0, a byte map at 1:1 -
    # keeps its algorithmic bytes in `chain_bytes`, SURVEY section 8d's figure, but moves none)
    moved_bytes = sum(q["read_bytes_per_frame"] + q["write_bytes_per_frame"] for q in prof if not q.get("folded"))

    value = aggregate([args.batch] * world, args.steps, dt)
    # the same with pass 6's unwritten varying read as 0 (GPU GL drivers): the phosphor mask is rendered and passes 7-10
    # carry signal everywhere.  Same steps, same barriers; its own per-pass timing.
    mask_rendered = None
    if key.startswith("crt-royale") and args.modes == "both":
        e.setUndefinedVaryingZero(True)
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dtm = max_over_ranks(time.perf_counter() - t0)
        e.setProfiling(True)
        for _ in range(args.steps):
            step()
        profm = [e.passProfile(i) for i in range(e.passCount())]
        e.setProfiling(False)
        e.setUndefinedVaryingZero(False)
        bytes_m = sum(q["read_bytes_per_frame"] + q["write_bytes_per_frame"] for q in profm)
        vm = aggregate([args.batch] * world, args.steps, dtm)
        mask_rendered = {"value": vm, "unit": "frames/s", "ms_per_step": dtm / args.steps * 1e3,
                         "algorithmic_bytes_per_frame": bytes_m,
                         "hbm_roofline_frac_whole_chain": vm / world * bytes_m / (HBM_PEAK_GBS * 1e9),
                         "per_pass_ms_per_frame": [q["total_ms"] / max(1, q["frames"]) for q in profm],
                         "note": "rc_engine_set_undefined_varying_zero(1): what GL drivers that read 0 from an unwritten varying render"}
    # ... and on frames with the statistics of video rather than of a random number generator (smooth gradients, edges, mild
    # grain): uniform noise is the worst case for every table gather of the kernels (64 different bytes per wave access), so
    # this is reported beside `value`, never instead of it.  Mask rendered (the mode real GL drivers produce), fewer steps.
    natural = None
    if key.startswith("crt-royale") and args.modes == "both" and world == 1:
        yy = torch.arange(h, device="cuda", dtype=torch.float32).view(1, h, 1)
        xx = torch.arange(w, device="cuda", dtype=torch.float32).view(1, 1, w)
        ph = torch.arange(n_local, device="cuda", dtype=torch.float32).view(n_local, 1, 1)
        base = torch.stack([128 + 100 * torch.sin(xx / 97.0 + ph * 0.3) * torch.cos(yy / 61.0),
                            128 + 90 * torch.sin((xx + yy) / 143.0 + ph * 0.2),
                            128 + 110 * torch.cos(yy / 53.0 - xx / 211.0 + ph * 0.1)], -1)
        base = base + 40.0 * ((torch.floor(xx / 160.0) + torch.floor(yy / 120.0)) % 2.0).unsqueeze(-1)          # blocks: hard edges
        base = base + torch.randint(-3, 4, (n_local, h, w, 3), device="cuda", generator=g).to(torch.float32)   # grain
        nat = torch.empty((n_local, h, w, 4), dtype=torch.uint8, device="cuda")
        nat[..., :3] = base.clamp_(0, 255).to(torch.uint8)
        nat[..., 3] = 255
        del base
        saved = frames
        frames = nat
        e.setUndefinedVaryingZero(True)
        n_steps = max(1, args.steps // 4)
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()
        dtn = time.perf_counter() - t0
        e.setProfiling(True)
        for _ in range(n_steps):
            step()
        profn = [e.passProfile(i) for i in range(e.passCount())]
        e.setProfiling(False)
        e.setUndefinedVaryingZero(False)
        frames = saved
        del nat
        natural = {"value": aggregate([args.batch], n_steps, dtn), "unit": "frames/s", "steps": n_steps, "mask_mode": "rendered",
                   "per_pass_ms_per_frame": [q["total_ms"] / max(1, q["frames"]) for q in profn],
                   "note": "synthetic frames with video-like statistics (gradients, block edges, +-3 grain); `value` stays on uniform noise"}
    # ... and on ONE lane (rc_engine_set_lanes(1)): `value` is measured with the engine's default of two lanes - the second half of
    # the batch on a helper instance with its own HIP stream, kernels of the two halves overlapping -; the per-pass times above and
    # the roofline figures are one-lane figures (a kernel's own duration: profiled runs use one lane), and this is the whole chain
    # at that setting.  roofline.overlap_factor = one-lane sum of pass time / two-lane wall time of the same frames.
    one_lane = None
    if args.lanes == 2 and args.modes != "mask" and world == 1 and n_local >= 2:
        try:
            e.setLanes(1)
            n_steps = max(1, args.steps // 4)
            for _ in range(args.warmup):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                step()
            barrier()
            dt1 = time.perf_counter() - t0
            one_lane = {"value": aggregate([args.batch], n_steps, dt1), "unit": "frames/s", "steps": n_steps, "lanes": 1,
                        "note": "rc_engine_set_lanes(1): the whole batch on the engine's own stream, same bytes"}
        except Exception as ex:   # a side measurement never takes the line down
            one_lane = {"value": None, "error": repr(ex)}
        finally:
            e.setLanes(args.lanes)
    # The pixel-art upscalers on what they are made for: frames of flat-coloured tiles and sprites from a 16-colour palette
    # instead of uniform noise, on which every xbr rule fires at nearly every pixel (the worst case; `value` stays on it).
    pixel_art = None
    if wl in ("xbr-lv3", "xbr-lv2", "scalefx") and world == 1:
        pal = torch.randint(0, 256, (16, 3), dtype=torch.uint8, device="cuda", generator=g)
        tiles = torch.randint(0, 16, (n_local, (h + 7) // 8, (w + 7) // 8), device="cuda", generator=g)
        idx = tiles.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :h, :w]
        spr = torch.randint(0, 16, (n_local, h, w), device="cuda", generator=g)
        keep = torch.rand((n_local, h, w), device="cuda", generator=g) < 0.08          # 8 % of the pixels: sprite detail
        idx = torch.where(keep, spr, idx)
        art = torch.empty((n_local, h, w, 4), dtype=torch.uint8, device="cuda")
        art[..., :3] = pal[idx]
        art[..., 3] = 255
        saved = frames
        frames = art
        n_steps = max(1, args.steps // 4)
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        barrier()
        dta = time.perf_counter() - t0
        frames = saved
        del art, idx, spr, keep, tiles
        pixel_art = {"value": aggregate([args.batch], n_steps, dta), "unit": "frames/s", "steps": n_steps,
                     "note": "frames of 8x8 flat tiles from a 16-colour palette with 8 % sprite detail; `value` stays on uniform noise"}
    ceiling, memcpy_rate = copy_ceiling(torch)
    pass_ms_per_step = sum(q["total_ms"] / max(1, q["frames"]) for q in prof) * n_local
    # the committed counters are replayed only for the kernel they were collected on (same launch shape, same duration within 15 %)
    fresh = pmc_fresh(infos[dom]["kernel"], avg_ms)
    traffic = pmc_traffic(infos[dom]["kernel"], frames_per_launch) if fresh else None
    out = {
        # BASELINE.json's metric for the default workload; other --workload values are side measurements
        "metric": ("1080p frames/sec, crt-royale 12-pass, 1/2/4/8 MI355X; % HBM roofline" if wl == "crt-royale"
                   else "frames/sec, %s; %% HBM roofline" % wl),
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank_s],
        "world_size_observed": comm_world, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "float_target_storage": "f16" if args.fp16_targets else "f32",
        "config": {"workload": desc + (" [" + ", ".join(args.param) + "]" if args.param else ""), "frames_per_gpu_per_step": args.batch, "global_batch": args.batch * world,
                   "chunk_frames": args.chunk or "default", "parallelism": "frames sharded, no collective",
                   "mask_mode": "rendered (rc_engine_set_undefined_varying_zero)" if args.modes == "mask" else "llvmpipe (pass 6 discards)",
                   "algorithmic_bytes_per_frame": chain_bytes,
                   "bytes_moved_per_frame": moved_bytes,
                   "folded_passes": [i for i, q in enumerate(prof) if q.get("folded")],
                   "hbm_roofline_frac_whole_chain": value / world * chain_bytes / (HBM_PEAK_GBS * 1e9),
                   "lanes": args.lanes,
                   "stream_copy_ceiling_GBs": ceiling, "memcpy_GBs": memcpy_rate,
                   "copy_ceiling_frac_whole_chain": value / world * chain_bytes / (ceiling * 1e9)},
        "roofline": {"bound": "hbm", "kernel": infos[dom]["kernel"], "pass": dom, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": ("profiles/" + PMC_FILE) if traffic is not None else None,
                     "avg_launch_ms": avg_ms, "frames_per_launch": frames_per_launch,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "frac_of_copy_ceiling": achieved / ceiling,
                     "lanes_of_this_figure": 1,
                     "overlap_factor": pass_ms_per_step / (dt / args.steps * 1e3) if args.lanes == 2 else 1.0,
                     "valu": pmc_valu(infos[dom]["kernel"], frames_per_launch, avg_ms) if fresh else None},
        "per_pass_ms_per_frame": [q["total_ms"] / max(1, q["frames"]) for q in prof],
    }
    if mask_rendered is not None:
        out["mask_rendered"] = mask_rendered
    if natural is not None:
        out["natural_frames"] = natural
    if one_lane is not None:
        out["one_lane"] = one_lane
    if pixel_art is not None:
        out["pixel_art_frames"] = pixel_art
    if args.io and rank == 0:
        out["io"] = io_measurements(e, w, h, args.batch, max(3, args.steps))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(key, w, h, vw, vh, tree, args.cpu_budget,
                                               custom={kv.split("=")[0]: float(kv.split("=")[1]) for kv in args.param},
                                               f16_targets=args.fp16_targets)
        print(json.dumps(out))
    e.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
