#!/usr/bin/env python
"""bench.py -- frames/sec of PointFusion on B = 8 synthetic 640x480 RGB-D sequences (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1 without WORLD_SIZE: spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  The B (= --batch, default 8) independent sequences (seeds 0..B-1) are sharded over the ranks
(rank r owns multigpu.shard_sequences(B, N, r): 8 / 4 / 2 / 1 sequences per GPU at N = 1 / 2 / 4 / 8) and every rank
tracks ITS sequences as one batch with gradslam_amd.slam.PointFusion(odom="gradicp").step(): every kernel of a frame
serves all sequences of the rank at once (workgroup -> sequence), which is how one GPU is filled -- one 640x480
sequence alone is a chain of ~50 dependent launches.  A step = one frame of every sequence through the
whole hot path: back-projection / normals, target selection, 20 gradLM point-to-plane ICP iterations (40 exact 1-NN
searches), projective association, fuse + append.  Frames are resident in HBM before the timed region.  Total work
is fixed (B sequences x K frames) as N grows: "scaling": "strong".  value = B * K / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel (fused exact grid 1-NN + Gauss-Newton linearisation, batched over the
                sequences): algorithmic bytes / its mean launch duration, measured with HIP events on the launch
                stream inside the library in a second pass over the SAME frames from the same map state (event
                overhead never touches `value`); `traffic` = HBM bytes per launch from the committed PMC passes;
  roofline_hbm  the HBM-bound kernel groups (K1 frame maps, grid build, K5 association, K6 fuse) and ms per step;
  cpu_baseline  gradslam's OWN CPU path (the staged reference, oracle/_ref + oracle/run_reference.py, "kind": "reference")
                timed on this machine's host cores in the same run on a bounded sample of the same workload, rank 0,
                N = 1 only, with the CPU oracle (the C port, oracle/) as a second field; parity of the timed GPU poses
                against both and against the committed reference goldens (ate_vs_reference_live_m, ate_vs_oracle_m,
                ate_vs_reference_golden).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector peak (= f32 MFMA rate), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0      # HBM3E spec; ~6290 GB/s achievable (float4 copy)
KNN_FLOP_PER_PAIR = 8.0    # 3 sub + 3 mul/fma + compare + select
H, W = 480, 640


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="independent sequences in total (sharded over the GPUs)")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--odom", default="gradicp", choices=["gradicp", "icp", "gt"])
    ap.add_argument("--workload", default="c4", choices=["c4", "c5"],
                    help="c4: B x 640x480 PointFusion (BASELINE configs[1]/[3]); c5: one 1296x968 sequence, growing map")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --batch sequences PER GPU (8 N in total by default)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (configs 2, 3 and 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=6,
                    help="timed frames of the CPU baseline sample (reference and oracle port: ~10 + ~13 s of CPU work at the default)")
    ap.add_argument("--no-roofline-pass", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, RCCL rendezvous on
    127.0.0.1) and relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    sys.exit(max(rcs))


def _make(a):
    # (the generator is pure numpy: loaded by path so that a worker process does not import torch)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_gs_synthetic", os.path.join(REPO, "gradslam_amd", "datasets", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make_sequence(*a[:3], seed=a[3], first=a[4])


def make_sequences(seeds, L, Hh, Ww, chunk=None):
    """the seeded synthetic sequences of this rank, generated in parallel on the host cores (a single long sequence
    in chunks of frames: every frame is a function of its index, only the hole / colour streams are per chunk)"""
    import multiprocessing as mp
    if chunk is None:
        chunk = L if L <= 64 else 8   # up to 64 frames: exactly make_sequence(L, seed) (what the goldens were recorded on)
    jobs = [(min(chunk, L - f0), Hh, Ww, s, f0) for s in seeds for f0 in range(0, L, chunk)]
    # under rocprofv3 (its tool library is preloaded into forked workers and its SIGTERM handler can dead-lock a
    # terminating pool) the sequences are generated in this process
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or "ROCPROFILER_LIBRARY_CTOR" in os.environ
    if len(jobs) == 1 or (profiled and os.environ.get("GRADSLAM_BENCH_SERIAL_GEN") == "1"):
        parts = [_make(j) for j in jobs]
    else:
        # profiled runs: workers are SPAWNED (numpy only) from an environment without the profiler's variables, so the
        # tool library is never loaded into them (a long profiled run generated serially costs minutes of box time)
        scrub = {}
        if profiled:
            for k in list(os.environ):
                if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE") or k.startswith(("ROCP", "ROCPROF", "ROCTX")):
                    scrub[k] = os.environ.pop(k)
        try:
            pool = mp.get_context("spawn" if profiled else "fork").Pool(min(len(jobs), os.cpu_count() or 1, 96))
            try:
                parts = pool.map(_make, jobs)
            finally:
                pool.close()   # workers leave on their own: no SIGTERM (Pool.__exit__ terminates)
                pool.join()
        finally:
            os.environ.update(scrub)
    out, per = [], (L + chunk - 1) // chunk
    for i in range(len(seeds)):
        ps = parts[i * per:(i + 1) * per]
        out.append({"colors": np.concatenate([p["colors"] for p in ps]), "depths": np.concatenate([p["depths"] for p in ps]),
                    "intrinsics": ps[0]["intrinsics"], "poses": np.concatenate([p["poses"] for p in ps])})
    return out


def frames_on_device(gs, seqs, device):
    import torch
    st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).to(device)  # noqa: E731
    poses = st("poses")
    poses[:, 1:] = poses[:, :1]  # only the first pose is given; the rest is recovered by ICP
    return gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)


def run_steps(slam, pc, frames, prev, first, last, poses_out=None, after_step=None, events=None):
    for s in range(first, last):
        live = frames[:, s]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        if poses_out is not None:
            poses_out.append(pose[:, 0])
        prev = live
        if events is not None:   # one event per step on the launch stream: ms of every step without a host sync
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            events.append(ev)
        if after_step is not None:
            after_step(pc)
    return pc, prev


def timed_steps(gs, slam, frames, Wm, K, device, barrier, seg_every=0):
    """W warm-up steps (untimed; frame 0 only initialises the map), then exactly K timed steps bracketed by `barrier`.
    Returns a dict: elapsed / host times, per-step GPU ms (events), poses, map."""
    import torch
    from gradslam_amd.structures.pointclouds import _CountGroup
    pc = gs.Pointclouds(device=device)
    recovered = []
    pc, prev = run_steps(slam, pc, frames, None, 0, Wm, recovered)
    torch.cuda.synchronize(device)
    marks = []

    def mark(p):   # one sync every seg_every frames, to report ms per frame against the map size
        n = len(recovered) - Wm
        if n > 0 and n % seg_every == 0:
            torch.cuda.synchronize(device)
            marks.append((n, time.perf_counter(), max(p._count_of(b)[0] for b in range(len(p._n_host)))))

    ev0 = torch.cuda.Event(enable_timing=True)
    events = []
    barrier()
    w0 = _CountGroup.wait_s
    ev0.record()
    t0 = time.perf_counter()
    pc, prev = run_steps(slam, pc, frames, prev, Wm, Wm + K, recovered, after_step=mark if seg_every else None, events=events)
    t_enq = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    waits = _CountGroup.wait_s - w0
    step_ms = [a.elapsed_time(b) for a, b in zip([ev0] + events[:-1], events)]
    q = max(len(step_ms) // 4, 1)
    seg_out, pn, pt = None, 0, t0
    if marks:   # ms per frame of consecutive stretches of the timed region (growing map)
        seg_out = []
        for n, t, m in marks:
            seg_out.append({"frames": "%d-%d" % (pn, n), "ms_per_frame": (t - pt) / (n - pn) * 1e3, "map_bound_end": m})
            pn, pt = n, t
    return {"elapsed": elapsed, "t_enq": t_enq, "waits": waits, "step_ms": step_ms,
            "ms_first_quartile": sum(step_ms[:q]) / q, "ms_last_quartile": sum(step_ms[-q:]) / q,
            "poses": torch.stack(recovered, 1), "pc": pc, "segments": seg_out}


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the newest committed profiles/*_pmc_hbm_traffic.txt (produced by
    tools/collect_profiles.sh + tools/summarize_profiles.py from separate PMC passes); (None, None) if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_hbm_traffic.txt")))
    if not files:
        return None, None
    tot, calls = 0.0, 0
    for line in open(files[-1]):
        if line.startswith("void " + kernel_prefix) or line.startswith(kernel_prefix):
            f = line.split()
            c, fetch_kb, write_kb = int(f[-5]), float(f[-3]), float(f[-2])
            tot += c * (fetch_kb + write_kb) * 1024.0
            calls += c
    if not calls:
        return None, None
    import re
    head = "".join(open(files[-1]).readlines()[:4])
    cmd = re.search(r"`([^`]*)`", head)
    # the file and the command line of the run it came from: NOT this run (rocprofv3 cannot wrap a run from inside)
    return tot / calls, {"file": "profiles/" + os.path.basename(files[-1]), "command": cmd.group(1) if cmd else None,
                         "note": "separate rocprofv3 --pmc passes of that command (shorter run, smaller maps): valid for the "
                                 "ICP kernel, whose traffic does not depend on the map size"}


def read_profile(lib, kind):
    ms, n, work = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.gs_profile_read(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work))
    return ms.value, n.value, work.value


def cpu_reference(Hh, Ww, odom, frames=3, timeout_s=900):
    """gradslam's OWN CPU path on this machine's host cores, in the same run (north_star): the staged
    reference (oracle/_ref/gradslam_ref.zip, python -m oracle.stage_reference; travels like the built .so) through oracle/run_reference.py
    in a subprocess -- PointFusion(odom).step of the unmodified reference on sequence 0, frame 0 = map init, frame 1 =
    warm-up, the rest timed.  Returns (dict or None, poses (frames, 4, 4) or None)."""
    if not os.path.isfile(os.path.join(REPO, "oracle", "_ref", "gradslam_ref.zip")):
        return None, None
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="gs_ref_"), "ref.json")
    env = dict(os.environ)
    for k in list(env):   # the reference run is not what a profiler wrapped around bench.py is after
        if k in ("LD_PRELOAD", "HSA_TOOLS_LIB") or k.startswith(("ROCP", "ROCPROF")):
            env.pop(k)
    cmd = [sys.executable, "-m", "oracle.run_reference", "--frames", str(frames), "--height", str(Hh), "--width", str(Ww),
           "--seed", "0", "--odom", odom, "--out", out]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "reference run exceeded %d s" % timeout_s}, None
    if r.returncode != 0 or not os.path.exists(out):
        return {"error": (r.stderr or r.stdout)[-400:]}, None
    d = json.load(open(out))
    return {"value": d["frames_per_s"], "unit": "frames/s", "cores": d["cores"], "host_cores": d.get("host_cores"),
            "kind": "reference",
            "sample": "unmodified gradslam %s PointFusion(odom='%s').step (torch %s, torch.set_num_threads(%d)) on frames 2..%d "
                      "of sequence 0 of the same workload after an untimed map-init frame and one warm-up frame; %.1f s of "
                      "CPU work; chamferdist.knn_points = the oracle's OpenMP brute force on all host cores (stand-in KNN: "
                      "the package is not installed), everything else the reference's own PyTorch code with an intra-op "
                      "pool of `cores` threads -- more threads make its small tensor operations slower (oracle/_ref, "
                      "oracle/shims, oracle/run_reference.py)"
                      % (d["gradslam_version"], odom, d["torch"], d["cores"], frames - 1, sum(d["seconds_per_frame"][2:])),
            "seconds_per_frame": d["seconds_per_frame"], "surfels_per_frame": d["counts"],
            "total_s": time.perf_counter() - t0}, np.asarray(d["poses"], np.float32)


def cpu_baseline(seq, n_full_steps, odom):
    """Oracle (port of the reference's algorithm) on the host cores: frame 0 initialises the map
    (untimed, like the warm-up), the next `n_full_steps` frames are timed.  Returns (dict, oracle poses)."""
    from oracle import slam as oslam
    L = 1 + n_full_steps
    poses = seq["poses"][:L].copy()
    poses[1:] = poses[:1]
    marks = []
    t0 = time.perf_counter()
    _, op = oslam.run_sequence(seq["colors"][:L], seq["depths"][:L], seq["intrinsics"][0], poses, odom=odom,
                               per_frame=lambda s, m, p: marks.append(time.perf_counter()))
    dt = marks[-1] - marks[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    out = {"value": n_full_steps / dt, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "oracle/ (C restatement of gradslam's CPU path, OpenMP) on frames 1..%d of sequence 0 of the same "
                     "workload after an untimed map-init frame; %.1f s of CPU work" % (n_full_steps, dt),
           "total_s": time.perf_counter() - t0}
    rec = os.path.join(REPO, "tests", "golden", "cpu_ref_timing.json")
    if os.path.exists(rec):   # the unmodified reference, timed once in the build container (oracle/make_golden_640.py)
        r = json.load(open(rec))
        out["reference_recorded"] = {"value": r["frames_per_s_steady"], "unit": "frames/s", "cores": r["cores"],
                                     "what": r["what"], "machine": r["machine"],
                                     "source": "tests/golden/cpu_ref_timing.json"}
    return out, op


def secondary_measurements(gs, args, frames, device, barrier, Wm, K):
    """Driver-observable numbers for the other BASELINE configs, in the same process after the headline (rank 0,
    N = 1; bounded to a few tens of seconds):
      b1_640x480      configs[1] = what ONE GPU runs when the 8 sequences are sharded over 8 GPUs (--gpus 8): sequence 0
                      alone, K timed steps, with the ICP kernel's mean launch duration from a short profiled pass;
      c3_gradicp      configs[2]: point_to_plane_gradICP on the 640x480 lattice (ds = 4), 20 iterations, forward and
                      forward (taped) + backward through all iterations, ms per call;
      c5_1296x968     configs[4] shape: one 1296x968 sequence, 200 timed frames, ms per frame against the map size."""
    import torch
    from gradslam_amd import _C, ops
    lib = _C.lib()
    out = {}
    t_start = time.perf_counter()

    # ---- configs[1]: one 640x480 sequence per GPU
    slam = gs.slam.PointFusion(odom=args.odom, device=device)
    one = frames[0:1, :]
    r = timed_steps(gs, slam, one, Wm, K, device, barrier)
    b1 = {"frames_per_s": K / r["elapsed"], "ms_per_step": r["elapsed"] / K * 1e3,
          "ms_per_step_first_quartile": r["ms_first_quartile"], "ms_per_step_last_quartile": r["ms_last_quartile"],
          "host_enqueue_ms_per_step": (r["t_enq"] - r["waits"]) / K * 1e3, "steps": K, "warmup": Wm,
          "map_surfels_end": [int(p.shape[0]) for p in r["pc"].points_list],
          "what": "sequence 0 of the headline workload alone on the GPU (the per-GPU shard of --gpus 8)"}
    pc2, prev2 = run_steps(slam, gs.Pointclouds(device=device), one, None, 0, Wm)
    torch.cuda.synchronize(device)
    nprof = min(K, 6)
    _C.check(lib.gs_profile_begin(64 * nprof + 1024), "gs_profile_begin")
    run_steps(slam, pc2, one, prev2, Wm, Wm + nprof, after_step=lambda p: p._tighten_counts())
    _C.check(lib.gs_profile_end(), "gs_profile_end")
    ms, n, nbytes = read_profile(lib, 8)
    if n > 0:
        ns = float((one.depth_image[:, Wm:Wm + nprof, ::4, ::4] > 0).sum()) / nprof   # valid lattice pixels per frame
        gbs = 36.0 * ns / (ms / n * 1e-3) / 1e9   # SURVEY 8(d) K4 bytes, as in the headline's roofline
        b1["roofline"] = {"kernel": "gs_icp_half_batch_kernel (one sequence per launch)", "bound": "hbm", "achieved": gbs,
                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "launches": n,
                          "avg_launch_us": ms * 1e3 / n, "alg_bytes_per_launch": 36.0 * ns,
                          "impl_bytes_per_launch": nbytes / n, "frac_impl": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                          "profiled_steps": nprof}
    out["b1_640x480"] = b1
    del pc2, prev2, r

    # ---- configs[2]: gradICP forward + backward through 20 iterations on the 640x480 lattice
    try:
        seq = make_sequences([0], 3, 480, 640)[0]
        dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        Kc, pose = dv(seq["intrinsics"][0]), dv(seq["poses"][0])
        sets = []
        for f in (0, 2):
            d = dv(seq["depths"][f, ..., 0])
            v, nm, _, _ = ops.frame_maps(d, Kc)
            gv, gn = ops.global_maps(v, nm, d, pose)
            sets.append(ops.downsample_frame(gv, gn, None, d, 4)[:2])
        (tgt, tn), (src, _) = sets
        src = src.clone().requires_grad_(True)

        def fwd_bwd():
            T, _ = ops.grad_icp(src, tgt, tn, None, 20, 1e-8, None, 2.0, 1.0, 1.0, 200.0)
            T.sum().backward()

        def fwd():
            with torch.no_grad():
                ops.icp(src.detach(), tgt, tn, numiters=20, return_idx=False)

        def timeit(f, n=10):
            for _ in range(3):
                f()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize(device)
            return (time.perf_counter() - t0) / n * 1e3

        out["c3_gradicp_640x480"] = {"forward_ms": timeit(fwd), "forward_taped_plus_backward_ms": timeit(fwd_bwd),
                                     "src_points": int(src.shape[0]), "tgt_points": int(tgt.shape[0]), "iterations": 20,
                                     "calls_timed": 10,
                                     "what": "point_to_plane_gradICP between frames 0 and 2 of sequence 0 on the ds = 4 "
                                             "lattice; backward = d(sum T)/d(src) through all 20 iterations"}
        # configs[2] as SURVEY states it: depth.requires_grad_(); PointFusion(gradicp) over two frames through the driver;
        # recovered_poses.sum().backward() -> depth.grad (ICP, down-sampler, frame-map and map-append backward kernels)
        col, K4 = dv(seq["colors"][None, :2]), dv(seq["intrinsics"][None])
        pp = dv(seq["poses"][None, :2]).clone()
        pp[:, 1:] = pp[:, :1]
        dd = dv(seq["depths"][None, :2]).requires_grad_(True)
        slam3 = gs.slam.PointFusion(odom=args.odom, device=device)

        def chain_fwd():
            with torch.no_grad():
                slam3(gs.RGBDImages(col, dd.detach(), K4, pp))

        def chain():
            dd.grad = None
            _, rp = slam3(gs.RGBDImages(col, dd, K4, pp))
            rp.sum().backward()

        out["c3_gradicp_640x480"].update({
            "driver_forward_ms": timeit(chain_fwd), "driver_forward_plus_backward_ms": timeit(chain),
            "driver_what": "depth.requires_grad_(); PointFusion(odom).__call__ on 2 frames of 640x480 (frame maps, map "
                           "init, localisation, fusion); recovered_poses.sum().backward(); depth.grad of both frames",
            "driver_depth_grad_norm": [float(dd.grad[0, f].double().norm()) for f in (0, 1)]})
    except Exception as e:   # noqa: BLE001  (a secondary number must not take the headline down)
        out["c3_gradicp_640x480"] = {"error": repr(e)}

    # ---- the same workload with the frames STREAMED from pinned host memory (uint16 depth, uint8
    # colour, one asynchronous copy per sequence and modality on a copy stream + one conversion launch per frame,
    # overlapped with the previous step) next to the same quantised frames resident in HBM
    try:
        from gradslam_amd.datasets.streaming import FrameStreamer, quantize_sequences
        Bs = len(frames._rgb_image)
        seqs_s = make_sequences(list(range(Bs)), Wm + K, 480, 640)
        d16, c8 = quantize_sequences(seqs_s, 5000.0)
        Ks = torch.from_numpy(np.stack([s["intrinsics"] for s in seqs_s])).to(device)
        P0 = torch.from_numpy(np.stack([s["poses"][:1] for s in seqs_s])).to(device)

        def stream_run(resident):
            slam_s = gs.slam.PointFusion(odom=args.odom, device=device)
            if resident:
                dres = torch.empty((Wm + K, Bs, 480, 640, 1), dtype=torch.float32, device=device)   # time-major, as the raw frames
                cres = torch.empty((Wm + K, Bs, 480, 640, 3), dtype=torch.float32, device=device)
                ops.ingest_frames_native(d16.to(device), c8.to(device), dres, cres, 5000.0)
                fr = gs.RGBDImages(cres.transpose(0, 1), dres.transpose(0, 1), Ks, P0.repeat(1, Wm + K, 1, 1))
                get = lambda t: fr[:, t]   # noqa: E731
            else:
                st = FrameStreamer(d16, c8, Ks, P0, scale_div=5000.0, device=device)
                get = st.frame
            pc_s, prev_s, rec = gs.Pointclouds(device=device), None, []
            for t in range(Wm):
                live = get(t)
                pc_s, p = slam_s.step(pc_s, live, prev_s, inplace=True)
                prev_s = live
            barrier()
            t0 = time.perf_counter()
            for t in range(Wm, Wm + K):
                live = get(t)
                pc_s, p = slam_s.step(pc_s, live, prev_s, inplace=True)
                prev_s = live
                rec.append(p[:, 0])
            barrier()
            el = time.perf_counter() - t0
            import hashlib
            return el, hashlib.sha256(torch.stack(rec, 1).cpu().numpy().tobytes()).hexdigest()[:16]

        el_r, sha_r = stream_run(True)
        el_s, sha_s = stream_run(False)
        raw_mb = Bs * 480 * 640 * 5 / 1e6
        out["stream_b%d_640x480" % Bs] = {
            "frames_per_s_streamed": Bs * K / el_s, "frames_per_s_resident": Bs * K / el_r, "ratio": el_r / el_s,
            "ms_per_step_streamed": el_s / K * 1e3, "ms_per_step_resident": el_r / K * 1e3,
            "raw_megabytes_per_step": raw_mb, "pcie_gb_per_s_needed": raw_mb / 1e3 / (el_s / K),
            "poses_sha_streamed": sha_s, "poses_sha_resident": sha_r, "identical": sha_s == sha_r,
            "what": "uint16 depth (metres x 5000) + uint8 colour in pinned host memory -> hipMemcpyAsync on a copy stream -> "
                    "gs_ingest_frames_native_f32 -> PointFusion.step; frame t + 1 in flight while step t computes "
                    "(gradslam_amd/datasets/streaming.py); resident = the same quantised frames as float32 in HBM"}
        del d16, c8
    except Exception as e:   # noqa: BLE001
        out["stream_640x480"] = {"error": repr(e)}

    # ---- the headline workload over a LONG timed region (the default 20 steps are a growth-phase average):
    # 200 timed steps after 30 warm-up frames, maps growing from 1.2 M to ~3 M surfels per sequence and then saturating
    # (the synthetic camera turns back at frame 150)
    try:
        Bs = len(frames._rgb_image)
        Ls, Ws = 230, 30
        fr_l = frames_on_device(gs, make_sequences(list(range(Bs)), Ls, 480, 640), device)
        slam_l = gs.slam.PointFusion(odom=args.odom, device=device)
        r = timed_steps(gs, slam_l, fr_l, Ws, Ls - Ws, device, barrier, seg_every=50)
        out["steady_b%d_200_steps" % Bs] = {
            "frames_per_s": Bs * (Ls - Ws) / r["elapsed"], "ms_per_step": r["elapsed"] / (Ls - Ws) * 1e3,
            "ms_per_step_first_quartile": r["ms_first_quartile"], "ms_per_step_last_quartile": r["ms_last_quartile"],
            "segments": r["segments"], "steps": Ls - Ws, "warmup": Ws,
            "map_surfels_end": [int(p.shape[0]) for p in r["pc"].points_list],
            "what": "the headline workload (B sequences of 640x480, resident) over 200 timed steps"}
        del fr_l, r
    except Exception as e:   # noqa: BLE001
        out["steady_200_steps"] = {"error": repr(e)}

    # ---- configs[4] shape: 1296x968, 200 timed frames, map growing past 5 M surfels (the 60-frame leg of the
    # rounds before showed neither; the 205 frames are generated on the host cores in ~15 s; GRADSLAM_BENCH_C5_FRAMES overrides)
    try:
        Lc, Wc = 5 + int(os.environ.get("GRADSLAM_BENCH_C5_FRAMES", "200")), 5
        seqs = make_sequences([0], Lc, 968, 1296, chunk=5)
        fr5 = frames_on_device(gs, seqs, device)
        slam5 = gs.slam.PointFusion(odom=args.odom, device=device)
        r = timed_steps(gs, slam5, fr5, Wc, Lc - Wc, device, barrier, seg_every=50 if Lc > 100 else 15)
        from gradslam_amd.metrics import ate_rmse as ate_np
        out["c5_1296x968"] = {"frames_per_s": (Lc - Wc) / r["elapsed"], "ms_per_frame": r["elapsed"] / (Lc - Wc) * 1e3,
                              "ms_per_frame_first_quartile": r["ms_first_quartile"],
                              "ms_per_frame_last_quartile": r["ms_last_quartile"], "frames_timed": Lc - Wc, "warmup": Wc,
                              "segments": r["segments"], "map_surfels_end": int(r["pc"].points_list[0].shape[0]),
                              "ate_vs_ground_truth_m": ate_np(r["poses"][0].cpu().numpy(), seqs[0]["poses"]),
                              "what": "one 1296x968 sequence (BASELINE configs[4] shape, the first %d frames of the 500-frame "
                                      "workload of --workload c5), dynamic map growth" % Lc}
        del fr5, r
    except Exception as e:   # noqa: BLE001
        out["c5_1296x968"] = {"error": repr(e)}
    out["seconds"] = time.perf_counter() - t_start
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch

    import gradslam_amd as gs
    from gradslam_amd import _C, multigpu
    from gradslam_amd.metrics import ate_rmse as ate_np

    rank, world, local = multigpu.init_from_env()
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    share = os.environ.get("GRADSLAM_BENCH_SHARE_GPU") == "1"
    if share:   # rehearsal of the N-rank path on fewer GPUs (gloo only)
        local %= torch.cuda.device_count()
    elif local >= torch.cuda.device_count():
        # one process per GPU is the contract: a rank without a device of its own would silently share cuda:0 and the
        # line would read as an N-GPU measurement
        raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no GPU of its own (%d visible); set "
                         "GRADSLAM_BENCH_SHARE_GPU=1 (with GRADSLAM_DIST_BACKEND=gloo) for a shared-GPU rehearsal"
                         % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if args.weak:               # weak scaling: the per-GPU work is fixed (--batch sequences on every GPU)
        args.batch = args.batch * world
    if args.workload == "c5":   # BASELINE configs[4]: one ScanNet-resolution sequence, dynamic map growth
        args.batch, args.height, args.width = world, 968, 1296
    Hh, Ww = args.height, args.width

    K, Wm = args.steps, max(args.warmup, 1)  # frame 0 only initialises the map: it is always warm-up
    L = Wm + K
    mine = multigpu.shard_sequences(args.batch, world, rank)
    assert mine, "more GPUs than sequences"
    seqs = make_sequences(mine, L, Hh, Ww)
    frames = frames_on_device(gs, seqs, device)
    B_local = len(mine)
    slam = gs.slam.PointFusion(odom=args.odom, device=device)
    lib = _C.lib()

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # A full (generation-2) pass of Python's cyclic garbage collector over the ~10^6 objects `import torch`
    # leaves behind takes 30-40 ms; where it falls depends on the allocation count, i.e. on unrelated details.
    # Collect now and freeze the survivors: later passes only look at objects created from here on.
    import gc
    gc.collect()
    gc.freeze()

    # ---------------- warm-up (untimed: map init + first ICP frames, allocator, kernels), then exactly K timed steps
    run = timed_steps(gs, slam, frames, Wm, K, device, barrier, seg_every=25 if args.workload == "c5" else 0)
    pc, elapsed_local, t_enq = run["pc"], run["elapsed"], run["t_enq"]
    elapsed = elapsed_local
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    n_map = [int(p.shape[0]) for p in pc.points_list]
    poses_local = run["poses"]  # (B_local, L, 4, 4) recovered trajectories of this rank's sequences
    ate_gt = max(ate_np(poses_local[b].cpu().numpy(), seqs[b]["poses"]) for b in range(B_local))

    # ---------------- the only exchange step: final pose (+ map) gather over RCCL
    barrier()
    g0 = time.perf_counter()
    all_poses = multigpu.gather_poses(poses_local)
    barrier()
    g1 = time.perf_counter()
    all_maps = multigpu.gather_maps(pc, dst=0) if world > 1 else pc   # (the maps go to rank 0 only)
    barrier()
    gather_ms = (time.perf_counter() - g0) * 1e3
    gather_poses_ms, gather_maps_ms = (g1 - g0) * 1e3, (time.perf_counter() - g1) * 1e3
    gather_peak = int(multigpu.GATHER_PEAK_BYTES) if world > 1 else 0
    assert all_poses.shape[0] == args.batch and (rank != 0 or len(all_maps) == args.batch)
    # fingerprint of the gathered result: the same for every N (sequences do not interact; tests/test_hip_batch.py)
    import hashlib
    sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]  # noqa: E731
    poses_sha = sha(all_poses)
    poses_sha_by_sequence = [sha(all_poses[b:b + 1]) for b in range(all_poses.shape[0])]
    n_map_all = [int(p.shape[0]) for p in all_maps.points_list] if rank == 0 else None
    # what every rank saw: its own clock around the K steps, its sequences, the fingerprint of its poses
    props = torch.cuda.get_device_properties(device)
    mine_info = {"rank": rank, "sequences": mine, "elapsed_s": elapsed_local, "ms_per_step": elapsed_local / K * 1e3,
                 "host_enqueue_ms_per_step": (t_enq - run["waits"]) / K * 1e3, "poses_sha": sha(poses_local),
                 "poses_sha_by_sequence": [sha(poses_local[b:b + 1]) for b in range(B_local)],
                 "device_index": device.index, "device_uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
    ranks_info = [mine_info]
    if world > 1:
        ranks_info = [None] * world
        torch.distributed.all_gather_object(ranks_info, mine_info)
        ids = ["%s/%s" % (r["device_uuid"], r["device_index"]) for r in ranks_info]   # (one process per GPU of ONE node)
        if len(set(ids)) != world and not share:   # (two processes on one device: not an N-GPU measurement)
            raise SystemExit("bench.py: %d ranks landed on %d distinct GPUs: %s" % (world, len(set(ids)), ids))
    backend_seen = {"world_size_env": int(os.environ.get("WORLD_SIZE", "1")),
                    "dist_world_size": torch.distributed.get_world_size() if world > 1 else 1,
                    "dist_backend": torch.distributed.get_backend() if world > 1 else None,
                    "cuda_device_count_rank0": torch.cuda.device_count(),
                    "distinct_devices": len(set("%s/%s" % (r["device_uuid"], r["device_index"]) for r in ranks_info)),
                    "shared_gpu_rehearsal": share}

    # ---------------- roofline pass: same frames from the same map state, HIP events inside the library
    roofline, roofline_hbm = None, None
    if not args.no_roofline_pass and rank == 0:
        # a second pass over the same frames, from scratch (nothing of it exists while the timed region runs: a map
        # snapshot taken before the timed region cost it a few per cent): warm-up frames unprofiled, then the K steps
        pc2, prev2 = run_steps(slam, gs.Pointclouds(device=device), frames, None, 0, Wm)
        torch.cuda.synchronize(device)
        _C.check(lib.gs_profile_begin(64 * K + 1024), "gs_profile_begin")
        # exact counts on the host in this pass (read-backs after every step; nothing of it touches `value`): the sizes
        # SURVEY.md section 8(d) states its algorithmic bytes in -- N surfels before the update, Na of them in the frame,
        # Nm matched pixels, Nn appended pixels, Ns valid lattice pixels (ICP source points) -- per sequence and step
        from gradslam_amd import ops
        cnt = {"N": 0.0, "Na": 0.0, "Nm": 0.0, "Nn": 0.0, "Ns": 0.0, "P": 0.0}
        state = {"s": Wm, "n_old": [int(p.shape[0]) for p in pc2.points_list]}

        def count_step(p):
            p._tighten_counts()
            sidx = state["s"]
            depth = frames.depth_image[:, sidx]                    # (B, H, W, 1)
            nvalid = (depth > 0).flatten(1).sum(1).tolist()
            nsrc = (depth[:, ::4, ::4] > 0).flatten(1).sum(1).tolist()
            n_now = [int(x.shape[0]) for x in p.points_list]
            for b in range(B_local):
                n_old = state["n_old"][b]
                pix = ops.project_map(p.points_list[b][:n_old], live_pose(p, b), frames.intrinsics[b, 0], Hh, Ww)
                nn = n_now[b] - n_old
                cnt["N"] += n_old
                cnt["Na"] += int((pix >= 0).sum())   # (rows of the map as merged: in-frame decisions differ on ~1e-5 of them)
                cnt["Nn"] += nn
                cnt["Nm"] += nvalid[b] - nn
                cnt["Ns"] += nsrc[b]
                cnt["P"] += Hh * Ww
            state["n_old"], state["s"] = n_now, sidx + 1

        pose_box = {}

        def live_pose(p, b):
            return pose_box["pose"][b, 0]

        def run_counted():
            pc_, prev_ = pc2, prev2
            for sidx in range(Wm, L):
                live = frames[:, sidx]
                pc_, pose = slam.step(pc_, live, prev_, inplace=True)
                pose_box["pose"] = pose
                prev_ = live
                count_step(pc_)

        run_counted()
        _C.check(lib.gs_profile_end(), "gs_profile_end")
        per_step = {k: v / K for k, v in cnt.items()}   # sums over the sequences of this GPU, per step
        ms, n, nbytes = read_profile(lib, 8)
        if n > 0:  # dominant kernel by GPU time: the fused exact-NN search + Gauss-Newton linearisation
            launches_per_step = n / K
            alg = 36.0 * per_step["Ns"]                 # 12 B source point + 24 B matched point and normal per query
            gbs = alg / (ms / n * 1e-3) / 1e9
            gbs_impl = nbytes / (ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic("gs_icp_half_batch_kernel")
            roofline = {"kernel": "gs_icp_half_batch_kernel<FULL> (K3+K4 fused, %d sequences per launch: prologue = row sums "
                                  "+ 6x6 solve / LM update of the previous half-iteration, then exact grid 1-NN + GN rows + "
                                  "one partial row per block), %d launches per step" % (B_local, round(launches_per_step)),
                        "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "launches": n, "avg_launch_us": ms * 1e3 / n, "alg_bytes_per_launch": alg,
                        "impl_bytes_per_launch": nbytes / n, "achieved_impl": gbs_impl, "frac_impl": gbs_impl / PEAK_HBM_GBS,
                        "alg_le_traffic": None if traffic is None else bool(alg <= traffic),
                        "note": "latency-bound, not bandwidth-bound.  alg_bytes = SURVEY.md 8(d) K4: 36 B per valid lattice "
                                "pixel (12 B source point read + 24 B matched target point and normal) of every sequence "
                                "per launch, counted on the host in this pass; impl_bytes = what the launch has to move "
                                "as built (12 B per lattice slot read, 12 B cloud written in first halves, 24 B match, "
                                "partial rows, one 16 B pass over the binned targets; exact device counts).  `traffic` = "
                                "fabric-side bytes per launch (HBM or Infinity Cache: hits of the latter ARE counted) from "
                                "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2 and WRITE x1 as calibrated "
                                "on this chip for 12-byte rows, 16-byte rows and 4-byte columns "
                                "(profiles/r06_counter_calibration.txt): every launch re-fetches its working set, the per-XCD "
                                "L2s do not survive a kernel boundary (profiles/r05_a_icp_l2_vs_frame.txt).",
                        "traffic_calibration": {"file": "profiles/r06_counter_calibration.txt", "fetch_factor": 2.0,
                                                "write_factor": 1.0, "counts_infinity_cache_hits": True}}
        # HBM-bound groups: alg_bytes from SURVEY.md section 8(d) and the counts above, impl_bytes as booked by the library
        # (the implementation's own scratch tables included); frac is computed from alg_bytes
        c = per_step
        alg_by_kind = {
            2: ("K1 frame maps + update tables", 32.0 * c["P"],
                "8(d) K1: 4 B depth read + 28 B (vertex, normal, alpha) written per pixel; the update's per-pixel tables "
                "(12 B per pixel) and the grid-scratch clear ride in the same launch and are impl bytes"),
            6: ("K2/K3 lattice + projection + grid build", 28.0 * c["P"] / 16.0 + 12.0 * c["N"],
                "K2 (no figure in 8(d)): 28 B per lattice slot (vertex, normal read; source point written) + 12 B per surfel "
                "projected under the previous pose; the counting sort of the targets into the grid is impl bytes"),
            4: ("K5 association", 28.0 * c["N"] + 24.0 * c["Na"] + 8.0 * c["P"],
                "8(d) K5: 28 B per surfel + 24 B frame vertex and normal per in-frame surfel + 8 B key r/w per pixel"),
            5: ("K6 fuse+append", 120.0 * c["Nm"] + 80.0 * c["Nn"] + 72.0 * max(c["N"] - c["Nm"], 0.0),
                "8(d) K6: matched surfel r/w 2 x 40 B + frame 40 B; appended pixel 40 B read + 40 B written; parity mode "
                "(every row renormalised like the reference) + 2 x 36 B per UNMATCHED surfel (the matched ones are counted "
                "once)"),
        }
        groups = {}
        for kind, (name, alg_b, what) in alg_by_kind.items():
            gms, gn, gbytes = read_profile(lib, kind)
            if gn > 0:
                gbs = alg_b * K / (gms * 1e-3) / 1e9
                groups[name] = {"achieved": gbs, "frac": gbs / PEAK_HBM_GBS, "launches": gn, "ms_per_step": gms / K,
                                "alg_bytes_per_step": alg_b, "impl_bytes_per_step": gbytes / K,
                                "frac_impl": gbytes / (gms * 1e-3) / 1e9 / PEAK_HBM_GBS, "alg_bytes_what": what}
        tot = {k: read_profile(lib, k)[0] for k in range(9)}
        step_ms = {n_: tot[k] / K for k, n_ in ((8, "icp_search_linearise"), (7, "icp_finish"), (6, "prep_project_grid_build"),
                                                (2, "frame_maps_update_tables"), (4, "associate"), (5, "fuse"))}
        alg_step = sum(g["alg_bytes_per_step"] for g in groups.values()) + (roofline["alg_bytes_per_launch"] * n / K if roofline else 0.0)
        roofline_hbm = {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "groups": groups,
                        "gpu_ms_per_step_by_group": step_ms,
                        "counts_per_step_sum_over_sequences": per_step,
                        "non_icp_gpu_ms_per_step": sum(v for k_, v in step_ms.items() if not k_.startswith("icp")),
                        "hbm_frac_whole_step": alg_step / (elapsed / K) / 1e9 / PEAK_HBM_GBS,
                        "hbm_frac_whole_step_impl": (sum(read_profile(lib, k)[2] for k in (2, 4, 5, 6, 8)) / K) /
                                                    (elapsed / K) / 1e9 / PEAK_HBM_GBS,
                        "note": "counts: N surfels before the update, Na of them in the frame, Nm matched pixels, Nn appended, "
                                "Ns valid lattice pixels, P pixels; hbm_frac_whole_step = 8(d) algorithmic bytes of the whole "
                                "step / wall time of a step / 8 TB/s"}

    cpu, ate_oracle, ate_ref, ate_refs = None, None, None, {}
    if (Hh, Ww) == (480, 640) and args.odom == "gradicp" and L <= 64:
        # the REAL reference's poses (oracle/make_golden_640.py) for every sequence of this rank that has a golden: seed 0
        # over 60 frames (pf640_l60.npz), seeds 1..7 over 25 frames -- the goldens cover the TIMED window of the default
        # run (frames 5 .. 24), and the record says for every sequence how many warm-up and how many timed frames it
        # compared (seeds 1..7 had 5-frame goldens, i.e. warm-up frames only)
        for b, sd in enumerate(mine):
            names = ["pf640_l60.npz", "pf640.npz"] if sd == 0 else ["pf640_s%d.npz" % sd]
            gp = next((q for q in (os.path.join(REPO, "tests", "golden", n_) for n_ in names) if os.path.exists(q)), None)
            if gp is not None:
                g = np.load(gp)
                nfr = min(L, g["poses"].shape[0])
                mine_p = poses_local[b, :nfr].cpu().numpy()
                rec = {"value_m": ate_np(mine_p, g["poses"][:nfr]), "frames": nfr, "source": "tests/golden/" + os.path.basename(gp),
                       "warmup_frames": {"frames": min(Wm, nfr), "value_m": ate_np(mine_p[:min(Wm, nfr)], g["poses"][:min(Wm, nfr)])},
                       "timed_frames": {"first": Wm, "frames": max(nfr - Wm, 0), "of": K,
                                        "value_m": ate_np(mine_p[Wm:], g["poses"][Wm:nfr]) if nfr > Wm else None}}
                ate_refs[str(sd)] = rec
        if rank == 0 and "0" in ate_refs:
            ate_ref = dict(ate_refs["0"], source=ate_refs["0"]["source"] + " (unmodified gradslam PointFusion on the same sequence)")
    ate_ref_live = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        port, op = cpu_baseline(seqs[0], args.cpu_frames, args.odom)
        ate_oracle = ate_np(poses_local[0, :op.shape[0]].cpu().numpy(), op)
        # gradslam's own CPU path, timed here and now; the oracle port stays next to it as a second field
        ref, ref_poses = cpu_reference(Hh, Ww, args.odom, frames=2 + max(args.cpu_frames, 2)) if mine[0] == 0 else (None, None)   # map init + warm-up + the timed frames
        if ref is not None and ref_poses is not None:
            cpu = dict(ref, port=port)
            nfr = min(L, ref_poses.shape[0])
            ate_ref_live = {"value_m": ate_np(poses_local[0, :nfr].cpu().numpy(), ref_poses[:nfr]), "frames": nfr,
                            "source": "the cpu_baseline run of this bench invocation (oracle/run_reference.py)"}
        else:
            cpu = dict(port, reference_live=ref if ref is not None else
                       {"error": "oracle/_ref is not staged on this machine (python -m oracle.stage_reference)"})

    seg_out = run["segments"]
    second = None
    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "c4" and (Hh, Ww) == (480, 640):
        second = secondary_measurements(gs, args, frames, device, barrier, Wm, K)
    if rank == 0:
        value = args.batch * K / elapsed
        el = [r["elapsed_s"] for r in ranks_info]
        out = {
            "metric": "frames/sec PointFusion %dx%d RGB-D" % (Ww, Hh), "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "n_ranks_seen_by_backend": backend_seen,
            "scaling_note": "this line is ONE point of the curve; scaling efficiency is unmeasured until the driver has a "
                            "SCALE record for N = 1, 2, 4, 8 (no 8-GPU node has run this code: rounds 1-4 SCALE = skipped)",
            "config": {"workload": "PointFusion(odom=%s, dsratio=4, numiters=20) forward, B=%d independent %dx%d "
                                   "sequences sharded over %d GPU(s) (%d per GPU, batched: every kernel serves all "
                                   "sequences of the GPU), frames resident in HBM (BASELINE configs[%s])"
                                   % (args.odom, args.batch, Ww, Hh, world, B_local, "4" if args.workload == "c5" else "3"),
                       "sequences_total": args.batch, "sequences_per_gpu": B_local, "frames_timed_per_sequence": K,
                       "map_surfels_end_rank0": n_map, "map_surfels_all": n_map_all, "poses_sha": poses_sha,
                       "parity_mode": "renormalize_unmatched=True (reference-identical merge)",
                       "poses_sha_by_sequence": poses_sha_by_sequence,
                       "final_gather_ms": gather_ms, "gather_poses_ms": gather_poses_ms, "gather_maps_to_rank0_ms": gather_maps_ms,
                       "gather_peak_bytes_rank0": gather_peak, "api": "gradslam_amd.slam.PointFusion.step",
                       "host_enqueue_ms_per_step": (t_enq - run["waits"]) / K * 1e3,
                       "host_wait_ms_per_step": run["waits"] / K * 1e3,
                       "host_note": "host_enqueue = Python + launches of a step; host_wait = what the host spends blocked on "
                                    "purpose (it is kept at most ~6 frames ahead of the device, structures/pointclouds.py)",
                       "ms_per_step_first_quartile": run["ms_first_quartile"],
                       "ms_per_step_last_quartile": run["ms_last_quartile"],
                       "icp_engine": os.environ.get("GRADSLAM_HIP_ICP_ENGINE", "rows"),
                       "host_readbacks_per_frame": 0 if gs.ops.DEVICE_COUNTS else 3,
                       "ate_vs_ground_truth_m_rank0_max": ate_gt, "ate_vs_oracle_m": ate_oracle,
                       "ate_vs_reference_live_m": ate_ref_live,
                       "ate_vs_reference_golden": ate_ref, "ate_vs_reference_goldens_by_seed_rank0": ate_refs},
            "segments": seg_out,
            "ranks": {"elapsed_s_min": min(el), "elapsed_s_max": max(el), "elapsed_s_mean": sum(el) / len(el),
                      "per_rank": ranks_info},
            "secondary": second,
            "roofline": roofline, "roofline_hbm": roofline_hbm, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
